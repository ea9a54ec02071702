for rg in 0 1 2; do
ASR_SCONV16_RG=$rg python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-exact-f32 --no-other-configs 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rg',$rg, round(d['ms_per_step'],2), d['config']['stage_ms']['unet'], round(d['roofline']['frac'],4))"
done

cd $GRAFT_REPO_ROOT
bash scripts/make_profiles.sh r06 > gpurun_out/make_profiles.log 2>&1
python bench.py --steps 20 --warmup 3 > gpurun_out/bench_20.txt 2>&1
tail -1 gpurun_out/bench_20.txt | cut -c1-300

timeout 600 python -m pytest tests/test_gpu_conv16.py -x -q -m gpu > gpurun_out/t1.log 2>&1
tail -2 gpurun_out/t1.log
scripts/gpu_profile.sh r06c_10m --steps 2 --warmup 1 --no-cpu-baseline --no-pipelined --no-exact-f32 --no-other-configs --no-other-legs > /dev/null 2>&1
python scripts/layer_table.py gpurun_out/prof_r06c_10m_sconv_trace.csv > gpurun_out/r06c_layers.txt 2>&1
tail -3 gpurun_out/r06c_layers.txt
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --no-exact-f32 --no-other-legs > gpurun_out/bench_r06c.json 2> gpurun_out/bench_r06c.err
python -c "
import json
d=json.load(open('gpurun_out/bench_r06c.json'))
print(d['ms_per_step'], d['config']['stage_ms'])"

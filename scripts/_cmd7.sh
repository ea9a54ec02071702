timeout 900 python -m pytest tests/test_gpu_conv16.py tests/test_gpu_network.py -x -q -m gpu > gpurun_out/t1.log 2>&1
tail -3 gpurun_out/t1.log
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --no-exact-f32 > gpurun_out/bench_r06b.json 2> gpurun_out/bench_r06b.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_r06b.json"))
print(d["ms_per_step"], d["config"]["stage_ms"])
v=d["config"].get("untimed_f16x2_narrower_arithmetic"); print("f16x2", v and v["unet_ms"])
PY

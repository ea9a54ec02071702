mkdir -p gpurun_out
scripts/micro/mfma_round > gpurun_out/mfma_round.txt 2>&1
python scripts/split_error_study.py --out gpurun_out/split_err_base.json > gpurun_out/split_err_base.log 2>&1
python scripts/split_error_study.py --lib build_variants/libasr_hip_chain1.so --out gpurun_out/split_err_chain1.json > gpurun_out/split_err_chain1.log 2>&1
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs > gpurun_out/bench_r06a.json 2> gpurun_out/bench_r06a.err
timeout 900 python -m pytest tests/test_gpu_conv16.py tests/test_gpu_network.py tests/test_abi.py -x -q -m gpu > gpurun_out/t1.log 2>&1
tail -3 gpurun_out/t1.log; cat gpurun_out/mfma_round.txt; tail -5 gpurun_out/split_err_base.log

timeout 1200 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu -k "failing_rank or world_one or (library_sharded_forward_equals and 30000) or (library_sharded_geometry and 48000)" > gpurun_out/t_shard.log 2>&1
tail -15 gpurun_out/t_shard.log
for v in 2 3; do
python scripts/split_error_study.py --lib build_variants/libasr_hip_chain$v.so --out gpurun_out/split_err_chain$v.json > gpurun_out/split_err_chain$v.log 2>&1
done
grep -h "bf16x3_unet_ms_at" gpurun_out/split_err_chain*.json

timeout 1500 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu > gpurun_out/t_shard.log 2>&1
tail -5 gpurun_out/t_shard.log
( python scripts/shard_dry_run.py 10000000 8 0 1; python scripts/shard_dry_run.py 10000000 8 4 1; python scripts/shard_dry_run.py 10000000 8 7 1; python scripts/shard_dry_run.py 10000000 2 0 1; python scripts/shard_dry_run.py 10000000 4 2 1 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r06b_shard_dry_run.txt
cat gpurun_out/r06b_shard_dry_run.txt

for v in 2 3; do
python scripts/split_error_study.py --lib build_variants/libasr_hip_chain$v.so --out gpurun_out/split_err_chain$v.json > gpurun_out/split_err_chain$v.log 2>&1
done
grep -h "bf16x3_unet_ms_at" gpurun_out/split_err_chain*.json

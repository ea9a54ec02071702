python scripts/split_error_study.py --lib build_variants/libasr_hip_chain4.so --out gpurun_out/split_err_chain4.json > gpurun_out/split_err_chain4.log 2>&1
grep -h "bf16x3_unet_ms_at" gpurun_out/split_err_chain4.json

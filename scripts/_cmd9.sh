for i in 1 2; do
python scripts/unet_time.py --lib build_variants/libasr_v2.so 2>/dev/null
python scripts/unet_time.py --lib build_variants/libasr_nosp.so 2>/dev/null
done

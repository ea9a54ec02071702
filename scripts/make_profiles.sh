#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel stats of the default bench command plus separate PMC passes
# (FETCH_SIZE, WRITE_SIZE, SQ busy counters).  Summaries land in gpurun_out/; copy them to profiles/.
tag=${1:-r01}
scripts/gpu_profile.sh ${tag}_10m --steps 2 --warmup 1 --no-cpu-baseline --no-pipelined --no-exact-f32 --no-other-configs --no-other-legs | head -1
scripts/gpu_pmc.sh ${tag}_fetch "FETCH_SIZE" --steps 1 --warmup 0 --no-cpu-baseline --no-pipelined --no-exact-f32 --no-other-configs --no-other-legs > /dev/null
scripts/gpu_pmc.sh ${tag}_write "WRITE_SIZE" --steps 1 --warmup 0 --no-cpu-baseline --no-pipelined --no-exact-f32 --no-other-configs --no-other-legs > /dev/null
scripts/gpu_pmc.sh ${tag}_sq "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" --steps 1 --warmup 0 --no-cpu-baseline --no-pipelined --no-exact-f32 --no-other-configs --no-other-legs > /dev/null
scripts/gpu_pmc.sh ${tag}_grbm "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" --steps 1 --warmup 0 --no-cpu-baseline --no-pipelined --no-exact-f32 --no-other-configs --no-other-legs > /dev/null
ls -la gpurun_out | grep ${tag}_

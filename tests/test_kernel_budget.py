"""Occupancy budget of the sparse-conv instances the bench runs, read from the code-object notes of the built object (no GPU):
a change that pushes the 128-column instance over 80 registers, or the 64- / 32-column ones over 64 registers or 40 960 bytes of
LDS, costs a workgroup per CU and 5-10 % of the U-Net -- caught here, at build time."""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "adaptive-surface-reconstruction_amd", "csrc")
OBJ = os.path.join(CSRC, "asr_conv16.o")
OBJ_CCONV = os.path.join(CSRC, "asr_conv.o")
sys.path.insert(0, os.path.join(REPO, "scripts"))

# (instance of k_sconv_plan16<NT, KC, WAVES, MODE = bf16x3, IMP, DUAL, SPLIT>, max VGPRs, max LDS bytes): 8-wave workgroups,
# 512 registers and 160 KB of LDS per SIMD / CU -> 80 registers and 53 KB for three, 64 registers and 40 KB for four
BUDGET = [
    ("k_sconv_plan16<8, 32, 8, 2, false, false, false>", 80, 53248),
    ("k_sconv_plan16<4, 32, 8, 2, false, false, false>", 64, 40960),
    ("k_sconv_plan16<2, 32, 8, 2, false, false, false>", 64, 40960),
    ("k_sconv_plan16<8, 32, 8, 2, false, true, false>", 128, 81920),
    ("k_sconv_plan16<4, 32, 8, 2, false, true, false>", 128, 81920),
]


@pytest.mark.skipif(not os.path.exists(OBJ), reason="asr_conv16.o has not been built")
def test_bench_instances_keep_their_blocks_per_cu():
    import kernel_regs
    ks = {k["demangled"]: k for k in kernel_regs.kernels(OBJ)}
    for name, vgpr, lds in BUDGET:
        k = ks[name]
        assert k.get("vgpr_count", 0) + k.get("agpr_count", 0) <= vgpr, (name, k)
        assert k.get("group_segment_fixed_size", 0) <= lds, (name, k)
        assert k.get("private_segment_fixed_size", 0) == 0 and k.get("vgpr_spill_count", 0) == 0, (name, k)


@pytest.mark.skipif(not os.path.exists(OBJ_CCONV), reason="asr_conv.o has not been built")
def test_continuous_conv_keeps_sixteen_waves_per_cu():
    """k_cconv_mfma runs one 16-wave workgroup per CU: 128 registers and the CU's 160 KB of LDS are the limits"""
    import kernel_regs
    ks = {k["demangled"]: k for k in kernel_regs.kernels(OBJ_CCONV)}
    for name in ("k_cconv_mfma<true, 4>", "k_cconv_mfma<false, 4>"):
        k = ks[name]
        assert k.get("vgpr_count", 0) + k.get("agpr_count", 0) <= 128, (name, k)
        assert k.get("group_segment_fixed_size", 0) <= 160 * 1024, (name, k)
        assert k.get("private_segment_fixed_size", 0) == 0 and k.get("vgpr_spill_count", 0) == 0, (name, k)

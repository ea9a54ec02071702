"""GPU parity of EVERY k_sconv_mfma<NT,KC,IMP,WAVES,DUAL> instance the 10 M-point full-width bench
dispatches (row a12: SpecialSparseConv.forward, /root/reference/models/common_torch.py:95-148,133-146).

The instance is normally chosen from the problem size (asr_conv.hip: asr_conv_sparse), so small inputs
never reach the wide / 8-wave / two-bank instances on their own.  Here each instance is forced
(asr_sparse_conv_args.force_nt / force_waves, or the per-context launch thresholds) on the full-width
layer shapes of UNet5 default.yaml and compared with the oracle; the library's per-instance launch
counters prove which template instance actually ran.

BENCH_INSTANCES (tests/sconv_instances.py) is the set in the committed profiles/r*_10m_sconv_trace.csv;
tests/test_abi.py checks the list against the newest committed profile on the CPU side.
"""
import numpy as np
import pytest
import torch

import parity
from asr_hip import synth
from oracle import oracle as O

from sconv_instances import BENCH_INSTANCES

pytestmark = pytest.mark.gpu

_close = parity.assert_close  # north_star: within 1e-5 (fp32), plain absolute + relative
_close_scaled = parity.assert_close_scaled


def _t(a, gpu):
    return torch.from_numpy(np.ascontiguousarray(a)).to(gpu)


@pytest.fixture(scope="module")
def geo():
    p, q = synth.scan_cloud(20000, seed=21, device="cpu")
    pts, nrm = p.numpy(), q.numpy()
    rad = synth.knn_radii(pts, 24)
    bb = synth.bounding_box(pts, 0.1)
    item = parity.oracle_geometry(pts, rad, *bb)
    return item


def _csr(item, kind, level):
    """(idx, kidx, rs, num_inp) of the 55-slot lists / up lists / inverted up lists of a level"""
    if kind == "nb":
        rs = item["neighbors_row_splits%d" % level]
        return item["neighbors_index%d" % level], item["neighbors_kernel_index%d" % level], rs, len(rs) - 1
    if kind == "up":  # rows = level, inputs = level + 1
        rs = item["up_neighbors_row_splits%d" % level]
        return (item["up_neighbors_index%d" % level], item["up_neighbors_kernel_index%d" % level], rs,
                len(item["voxel_sizes%d" % (level + 1)]))
    # down: rows = level + 1, inputs = level
    n_coarse = len(item["voxel_sizes%d" % (level + 1)])
    idx, rs, attr = O.invert_neighbors_list(n_coarse, item["up_neighbors_index%d" % level],
                                            item["up_neighbors_row_splits%d" % level],
                                            item["up_neighbors_kernel_index%d" % level])
    return idx, attr, rs, len(item["voxel_sizes%d" % level])


# (instance, layer of UNet5 default.yaml that the bench runs on it, CSR kind, level, K, cin, cout_a, cout_b)
# cout_b > 0: conv1a + conv1b in one launch (two filter banks); widths from SURVEY 3.3
CASES = [
    ((4, 32, 1, 8, 1), "encblock0.conv1a+1b", "nb", 0, 55, 32, 56, 8),
    ((4, 32, 0, 8, 0), "encblock0.conv2", "nb", 0, 55, 64, 64, 0),
    ((4, 32, 0, 8, 0), "up0.conv1", "up", 0, 9, 128, 64, 0),
    ((8, 16, 1, 8, 1), "down1.conv1a+1b", "down", 0, 9, 64, 120, 8),
    ((8, 16, 1, 8, 1), "encblock1.conv1a+1b", "nb", 1, 55, 128, 120, 8),
    ((8, 16, 0, 8, 0), "encblock1.conv2", "nb", 1, 55, 128, 128, 0),
    ((8, 16, 0, 8, 0), "decblock1.conv1 (384 = up 256 | skip 128)", "nb", 1, 55, 384, 128, 0),
    ((8, 16, 1, 8, 1), "down2.conv1a+1b", "down", 1, 9, 128, 248, 8),
    ((8, 16, 1, 8, 1), "encblock2.conv1a+1b", "nb", 2, 55, 256, 248, 8),
    ((8, 16, 0, 8, 0), "encblock2.conv2", "nb", 2, 55, 256, 256, 0),
    ((8, 16, 0, 8, 0), "decblock2.conv1 (512)", "nb", 2, 55, 512, 256, 0),
    ((16, 16, 0, 8, 0), "up1.conv1", "up", 1, 9, 256, 256, 0),
    ((4, 32, 1, 4, 1), "down3.conv1a+1b / encblock3.conv1a+1b", "nb", 3, 55, 256, 248, 8),
    ((4, 32, 0, 4, 0), "encblock3.conv2", "nb", 3, 55, 256, 256, 0),
    ((4, 32, 0, 4, 0), "decblock3.conv1 (512)", "nb", 3, 55, 512, 256, 0),
    ((4, 32, 0, 4, 0), "up2.conv1", "up", 2, 9, 256, 256, 0),
    ((2, 64, 1, 4, 1), "down3 (3->4) / encblock4.conv1a+1b", "nb", 4, 55, 256, 248, 8),
    ((2, 64, 1, 4, 1), "down3 (3->4), K = 9", "down", 3, 9, 256, 248, 8),
    ((2, 64, 0, 4, 0), "encblock4.conv2", "nb", 4, 55, 256, 256, 0),
    ((2, 64, 0, 4, 0), "up3.conv1", "up", 3, 9, 256, 256, 0),
    ((2, 64, 0, 8, 0), "decblock0.conv1", "nb", 0, 55, 64, 32, 0),
    ((2, 32, 0, 8, 0), "decblock0.conv2", "nb", 0, 55, 32, 32, 0),
    # instances the launcher can also pick at other cloud sizes (1 M, 80 M per GPU): same check
    ((16, 16, 1, 8, 1), "encblock2.conv1a+1b at >= 4 M voxels on level 2", "nb", 2, 55, 256, 248, 8),
    ((16, 16, 0, 4, 0), "256-wide, 4 waves", "nb", 2, 55, 256, 256, 0),
    ((8, 16, 0, 4, 0), "128-wide, 4 waves", "nb", 1, 55, 128, 128, 0),
    ((1, 64, 0, 4, 0), "16-wide column tile", "nb", 2, 55, 64, 16, 0),
    ((1, 32, 1, 4, 0), "16-wide column tile, importance", "nb", 1, 55, 32, 16, 0),
]


def test_cases_cover_the_bench_instances():
    assert BENCH_INSTANCES <= {c[0] for c in CASES}


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%s-%s" % ("x".join(map(str, c[0])), c[1].split(" ")[0]))
def test_forced_instance_vs_oracle(geo, gpu, case):
    from asr_hip import ops
    (nt, kc, imp_flag, waves, dual), _, kind, level, K, cin, ca, cb = case
    idx, kidx, rs, num_inp = _csr(geo, kind, level)
    v = len(rs) - 1
    rng = np.random.default_rng(cin * 977 + ca + 13 * level)
    f = rng.standard_normal((num_inp, cin)).astype(np.float32)
    # variance preserving: ~8 of 55 (or 1 of 9) slots occupied per row
    occ = 8.0 if K == 55 else 1.0
    Wa = (rng.standard_normal((K, cin, ca)) * np.sqrt(2.0 / (occ * cin))).astype(np.float32)
    ba = (rng.standard_normal(ca) * 0.1).astype(np.float32)
    imp = rng.uniform(0.05, 1.0, size=num_inp).astype(np.float32)
    nimp = imp[idx.astype(np.int64)]
    perm = ops.row_groups(_t(kidx, gpu), _t(rs, gpu))  # the whole-path driver always passes a tiling order
    ctx = ops.context(gpu)
    ctx.sconv_variant_counts(reset=True)
    common = dict(row_perm=perm, force_nt=nt, force_waves=waves, algo=2)
    if dual:
        Wb = (rng.standard_normal((K, cin, cb)) * np.sqrt(2.0 / (occ * cin))).astype(np.float32)
        bb = (rng.standard_normal(cb) * 0.1).astype(np.float32)
        out, oimp = ops.sparse_conv(_t(Wa, gpu), _t(f, gpu), _t(idx, gpu), _t(kidx, gpu), _t(rs, gpu),
                                    inp_importance=_t(imp, gpu), normalize=True, bias=_t(ba, gpu), relu=True,
                                    return_importance=True, filters_b=_t(Wb, gpu), bias_b=_t(bb, gpu), **common)
        ref_a = np.maximum(O.sparse_conv(Wa, f, idx, kidx, None, rs, False) + ba, 0)      # conv1a: plain
        ref_b = np.maximum(O.sparse_conv(Wb, f, idx, kidx, nimp, rs, True) + bb, 0)       # conv1b: weighted
        got = out.cpu().numpy()
        _close(got[:, :ca], ref_a)
        _close(got[:, ca:], ref_b)
        _close(oimp.cpu().numpy(), O.reduce_subarrays_sum(nimp, rs))
    elif imp_flag:
        out, oimp = ops.sparse_conv(_t(Wa, gpu), _t(f, gpu), _t(idx, gpu), _t(kidx, gpu), _t(rs, gpu),
                                    inp_importance=_t(imp, gpu), normalize=True, bias=_t(ba, gpu), relu=True,
                                    return_importance=True, **common)
        _close(out.cpu().numpy(), np.maximum(O.sparse_conv(Wa, f, idx, kidx, nimp, rs, True) + ba, 0))
        _close(oimp.cpu().numpy(), O.reduce_subarrays_sum(nimp, rs))
    else:
        res = rng.standard_normal((v, ca)).astype(np.float32)
        out = ops.sparse_conv(_t(Wa, gpu), _t(f, gpu), _t(idx, gpu), _t(kidx, gpu), _t(rs, gpu), bias=_t(ba, gpu),
                              relu=True, residual=_t(res, gpu), **common)
        ref = np.maximum(O.sparse_conv(Wa, f, idx, kidx, None, rs, False) + ba, 0) + res
        _close(out.cpu().numpy(), ref)
    assert ctx.sconv_variant_counts() == {(nt, kc, imp_flag, waves, dual): 1}


def test_forced_instances_agree_bitwise_with_each_other(geo, gpu):
    """the tile shape only changes which block computes a row: every instance accumulates a row's
    slots in the same order, so all of them give identical bits"""
    from asr_hip import ops
    idx, kidx, rs, num_inp = _csr(geo, "nb", 1)
    rng = np.random.default_rng(5)
    f = rng.standard_normal((num_inp, 128)).astype(np.float32)
    W = (rng.standard_normal((55, 128, 128)) * 0.05).astype(np.float32)
    outs = []
    for nt, waves in ((8, 8), (8, 4), (4, 8), (4, 4), (2, 8), (2, 4), (1, 4), (16, 8)):
        outs.append(ops.sparse_conv(_t(W, gpu), _t(f, gpu), _t(idx, gpu), _t(kidx, gpu), _t(rs, gpu), algo=2,
                                    force_nt=nt, force_waves=waves))
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


def test_force_arguments_are_validated(geo, gpu):
    from asr_hip import _lib, ops
    idx, kidx, rs, num_inp = _csr(geo, "nb", 3)
    f = torch.zeros((num_inp, 32), device=gpu)
    W = torch.zeros((55, 32, 32), device=gpu)
    for bad in (dict(force_nt=3), dict(force_nt=1, force_waves=8), dict(force_waves=5)):
        with pytest.raises(_lib.AsrHipError):
            ops.sparse_conv(W, f, _t(idx, gpu), _t(kidx, gpu), _t(rs, gpu), algo=2, **bad)


@pytest.mark.parametrize("opts,expect", [
    # no minimum block count, 8-wave blocks always: the widest tile of every layer, as at >= 10 M points
    (dict(sconv_min_blocks=0, sconv_wide_min=0),
     {(4, 32, 1, 8, 1), (4, 32, 0, 8, 0), (8, 16, 1, 8, 1), (8, 16, 0, 8, 0), (16, 16, 1, 8, 1), (16, 16, 0, 8, 0),
      (2, 64, 0, 8, 0), (2, 32, 0, 8, 0)}),
    # everything narrowed to 32-column tiles and 4-wave blocks, as on the coarsest grids
    (dict(sconv_min_blocks=1 << 40, sconv_wide_min=1 << 40),
     {(2, 32, 1, 4, 1), (2, 64, 1, 4, 1), (2, 64, 0, 4, 0), (2, 32, 0, 4, 0)}),
])
def test_whole_path_full_width_with_bench_tile_shapes(gpu, opts, expect):
    """channel_div = 1 (the bench's widths) on a small cloud, with the launch thresholds set so that the
    launcher picks the tile shapes it picks at bench scale; values vs the oracle within 1e-5"""
    from asr_hip.pipeline import ImplicitPipeline
    p, q = synth.scan_cloud(6000, seed=31, device="cpu")
    pts, nrm = p.numpy(), q.numpy()
    rad = synth.knn_radii(pts, 24)
    bb = synth.bounding_box(pts, 0.1)
    weights = synth.make_weights(channel_div=1, seed=4)
    with O.precise():
        ref = parity.oracle_forward(pts, nrm, rad, bb[0], bb[1], weights)
    pipe = ImplicitPipeline(weights, device=gpu)
    for k, val in opts.items():
        pipe.ctx.set_option(k, val)
        assert pipe.ctx.get_option(k) == val
    pipe.ctx.sconv_variant_counts(reset=True)
    values = pipe.forward(_t(pts, gpu), _t(nrm, gpu), _t(rad, gpu), bb[0], bb[1])
    counts = pipe.ctx.sconv_variant_counts()
    assert sum(counts.values()) == 44  # 53 convs, conv1a + conv1b fused: 44 launches
    assert set(counts) == expect, counts
    _close(pipe.get("feats1").cpu().numpy(), ref["feats1"])
    _close_scaled(pipe.get("code").cpu().numpy(), ref["code"])
    _close(values.cpu().numpy(), ref["values"])

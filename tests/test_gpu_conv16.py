"""GPU parity of the 16-bit matrix-core variants of row a12 (asr_hip_sparse_conv_f16 / _bf16x3;
SpecialSparseConv.forward, /root/reference/models/common_torch.py:95-148).

bf16x3 ("fp32-class results from bf16 MFMAs"): compared with the oracle at the north_star tolerance,
1e-5, like the f32 kernel.
f16 (BASELINE config C5, "fp16 features"): the oracle runs in its precise mode on the SAME f16-rounded
activations and weights, so what is left is the accumulation order and, for f16 outputs, the final
rounding: tolerance 2e-3 relative to the tensor's range for f16 outputs (f16 has 11 significant bits:
half an ulp is 4.9e-4), 1e-5 for f32 outputs.
"""
import numpy as np
import pytest
import torch

import parity
from asr_hip import synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu

_close = parity.assert_close
_close_scaled = parity.assert_close_scaled


def _t(a, gpu):
    return torch.from_numpy(np.ascontiguousarray(a)).to(gpu)


def _h(a):
    """round to f16 and back: what an f16 buffer holds"""
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


@pytest.fixture(scope="module")
def geo():
    p, q = synth.scan_cloud(20000, seed=21, device="cpu")
    pts = p.numpy()
    rad = synth.knn_radii(pts, 24)
    bb = synth.bounding_box(pts, 0.1)
    return parity.oracle_geometry(pts, rad, *bb)


def _csr(item, kind, level):
    if kind == "nb":
        rs = item["neighbors_row_splits%d" % level]
        return item["neighbors_index%d" % level], item["neighbors_kernel_index%d" % level], rs, len(rs) - 1
    rs = item["up_neighbors_row_splits%d" % level]
    return (item["up_neighbors_index%d" % level], item["up_neighbors_kernel_index%d" % level], rs,
            len(item["voxel_sizes%d" % (level + 1)]))


# (kind, level, K, cin, cout_a, cout_b, force_nt, force_waves)
SHAPES = [
    ("nb", 0, 55, 32, 56, 8, 0, 0), ("nb", 0, 55, 64, 64, 0, 4, 8), ("nb", 0, 55, 64, 32, 0, 2, 4),
    ("nb", 1, 55, 128, 120, 8, 8, 8), ("nb", 1, 55, 128, 128, 0, 8, 4), ("nb", 1, 55, 384, 128, 0, 4, 8),
    ("nb", 2, 55, 256, 248, 8, 8, 4), ("nb", 2, 55, 512, 256, 0, 8, 8), ("up", 1, 9, 256, 256, 0, 0, 0),
    ("nb", 3, 55, 256, 256, 0, 2, 4), ("nb", 2, 55, 40, 16, 0, 1, 4), ("nb", 1, 55, 8, 24, 0, 0, 0),
    # the remaining instances of the 10 M-point bench (tests/sconv_instances.py BENCH_INSTANCES16)
    ("nb", 0, 55, 32, 56, 8, 4, 8), ("nb", 3, 55, 256, 248, 8, 4, 4), ("nb", 3, 55, 256, 256, 0, 4, 4),
    ("nb", 0, 55, 32, 32, 0, 2, 8), ("nb", 4, 55, 256, 248, 8, 2, 4),
]


def test_shapes_cover_the_bench_instances():
    from sconv_instances import BENCH_SHAPES16
    forced = {(s[6], 32, 0, s[7], int(s[5] > 0)) for s in SHAPES if s[6] and s[7] and s[3] % 32 == 0}
    assert BENCH_SHAPES16 <= forced, BENCH_SHAPES16 - forced  # every SHAPE runs in bf16x3 and in f16x2


@pytest.fixture
def plan_option(request, gpu):
    """runs a test with the plan-driven kernel (default) or with the table-driven one"""
    from asr_hip import ops
    ctx = ops.context(gpu)
    ctx.set_option("sconv_plan", int(request.param))
    yield bool(request.param)
    ctx.set_option("sconv_plan", 1)


@pytest.mark.parametrize("plan_option", [1, 0], ids=["plan", "table"], indirect=True)
@pytest.mark.parametrize("mode", ["bf16x3", "f16x2", "f16"])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "%s%d-%dx%d+%d-nt%dw%d" % (s[0], s[1], s[3], s[4], s[5], s[6], s[7]))
def test_sparse_conv16_vs_oracle(geo, gpu, mode, shape, plan_option):
    from asr_hip import ops
    kind, level, K, cin, ca, cb, nt, waves = shape
    # the plan-driven kernel takes cin that fills whole panels; everything else goes to the table-driven one
    kc = 64 if mode == "f16" and ((cin + 31) // 32 * 32) % 64 == 0 else 32
    expect_plan = int(plan_option and cin % kc == 0)
    if mode != "f16" and cin % 4:
        pytest.skip("f32 rows need cin % 4 == 0")
    idx, kidx, rs, num_inp = _csr(geo, kind, level)
    v = len(rs) - 1
    rng = np.random.default_rng(cin * 31 + ca + level)
    occ = 8.0 if K == 55 else 1.0
    f = rng.standard_normal((num_inp, cin)).astype(np.float32)
    Wa = (rng.standard_normal((K, cin, ca)) * np.sqrt(2.0 / (occ * cin))).astype(np.float32)
    ba = (rng.standard_normal(ca) * 0.1).astype(np.float32)
    imp = rng.uniform(0.05, 1.0, size=num_inp).astype(np.float32)
    nimp = imp[idx.astype(np.int64)]
    Wb = (rng.standard_normal((K, cin, cb)) * np.sqrt(2.0 / (occ * cin))).astype(np.float32) if cb else None
    bb = (rng.standard_normal(cb) * 0.1).astype(np.float32) if cb else None
    f16 = mode == "f16"
    act = torch.float16 if f16 else torch.float32
    # what the kernel sees: f16 mode rounds activations and weights to f16
    fq, Wa_q, Wb_q = (_h(f), _h(Wa), _h(Wb) if cb else None) if f16 else (f, Wa, Wb)
    packed = ops.pack_filters(_t(Wa, gpu), mode, _t(Wb, gpu) if cb else None)
    ctx = ops.context(gpu)
    ctx.sconv_variant_counts(reset=True)
    kw = dict(bias=_t(ba, gpu), relu=True, force_nt=nt, force_waves=waves)
    with O.precise():
        ref_a = np.maximum(O.sparse_conv(Wa_q, fq, idx, kidx, None, rs, False) + ba, 0)
        if cb:
            ref_b = np.maximum(O.sparse_conv(Wb_q, fq, idx, kidx, nimp, rs, True) + bb, 0)
            ref = np.concatenate([ref_a, ref_b], 1)
        else:
            ref = ref_a
        ref_imp = np.maximum(O.sparse_conv(Wa_q, fq, idx, kidx, nimp, rs, True) + ba, 0)
    # f32 output: accumulation order only
    if cb:
        out, oimp = ops.sparse_conv16(mode, packed, K, cin, ca, _t(f, gpu).to(act), _t(idx, gpu), _t(kidx, gpu),
                                      _t(rs, gpu), inp_importance=_t(imp, gpu), normalize=True, cout_b=cb,
                                      bias_b=_t(bb, gpu), return_importance=True, out_dtype=torch.float32, **kw)
        _close(oimp.cpu().numpy(), O.reduce_subarrays_sum(nimp, rs))
    else:
        out = ops.sparse_conv16(mode, packed, K, cin, ca, _t(f, gpu).to(act), _t(idx, gpu), _t(kidx, gpu),
                                _t(rs, gpu), out_dtype=torch.float32, **kw)
    assert out.dtype == torch.float32
    _close(out.cpu().numpy(), ref)
    key = list(ctx.sconv_variant_counts())
    assert len(key) == 1 and key[0][5] == {"f16": 1, "bf16x3": 2, "f16x2": 3}[mode] and (nt == 0 or key[0][0] == nt)
    assert key[0][6] == expect_plan, key
    if not cb:
        # importance weighted + normalised single bank (conv1b alone), residual after the activation
        res = rng.standard_normal((v, ca)).astype(np.float32)
        out2, oimp = ops.sparse_conv16(mode, packed, K, cin, ca, _t(f, gpu).to(act), _t(idx, gpu), _t(kidx, gpu),
                                       _t(rs, gpu), inp_importance=_t(imp, gpu), normalize=True,
                                       residual=_t(res, gpu).to(act), return_importance=True,
                                       out_dtype=torch.float32, **kw)
        _close(out2.cpu().numpy(), ref_imp + (_h(res) if f16 else res))
        _close(oimp.cpu().numpy(), O.reduce_subarrays_sum(nimp, rs))
    if f16:
        # f16 output (activations stay f16 in HBM): the f32 result rounded once
        o16 = ops.sparse_conv16(mode, packed, K, cin, ca, _t(f, gpu).to(act), _t(idx, gpu), _t(kidx, gpu), _t(rs, gpu),
                                inp_importance=_t(imp, gpu) if cb else None, normalize=bool(cb), cout_b=cb,
                                bias_b=_t(bb, gpu) if cb else None, **kw)
        assert o16.dtype == torch.float16
        got = o16.float().cpu().numpy()
        scale = max(1.0, float(np.abs(ref).max()))
        assert np.abs(got - ref).max() <= 2e-3 * scale
        # and bit-identical to rounding the kernel's own f32 result
        assert torch.equal(o16, out.to(torch.float16))


def test_bf16x3_split_is_exact(gpu):
    """a = a0 + a1 + a2 exactly for every finite f32 (a0 = rn(a), a1 = rn(a - a0), a2 = the rest): a 1 x 1
    'convolution' with one neighbour per row and an identity filter returns its input bit for bit"""
    from asr_hip import ops
    rng = np.random.default_rng(0)
    v, c = 4096, 32
    f = (rng.standard_normal((v, c)) * np.exp(rng.uniform(-20, 20, size=(v, c)))).astype(np.float32)
    W = np.zeros((1, c, c), np.float32)
    W[0] = np.eye(c, dtype=np.float32)
    idx = np.arange(v, dtype=np.int32)
    kidx = np.zeros(v, np.uint8)
    rs = np.arange(v + 1, dtype=np.int64)
    packed = ops.pack_filters(_t(W, gpu), "bf16x3")
    out = ops.sparse_conv16("bf16x3", packed, 1, c, c, _t(f, gpu), _t(idx, gpu), _t(kidx, gpu), _t(rs, gpu))
    assert np.array_equal(out.cpu().numpy().view(np.uint32), f.view(np.uint32))


@pytest.mark.parametrize("precision,tol", [("bf16x3", 1e-5), ("f16x2", 1e-5), ("f16", 5e-3)])
def test_whole_path_precisions(gpu, precision, tol):
    """asr_implicit_params.precision.  bf16x3 / f16x2: implicit values within 1e-5 of the range, like f32.
    f16 (config C5): f16 activations through 53 layers; bound 5e-3 of the range against the exact oracle
    (every layer rounds its output to 11 bits: 4.9e-4 per rounding; measured 1.9e-3), on the mixed-density
    cloud of that config."""
    from asr_hip.pipeline import ImplicitPipeline
    p, q = synth.scan_cloud(20000, seed=41, device="cpu", density_variance=10.0)
    pts, nrm = p.numpy(), q.numpy()
    rad = synth.knn_radii(pts, 24)
    bb = synth.bounding_box(pts, 0.1)
    weights = synth.make_weights(channel_div=1, seed=8)
    with O.precise():
        ref = parity.oracle_forward(pts, nrm, rad, bb[0], bb[1], weights)
    pipe = ImplicitPipeline(weights, device=gpu, precision=precision)
    pipe.ctx.sconv_variant_counts(reset=True)
    values = pipe.forward(_t(pts, gpu), _t(nrm, gpu), _t(rad, gpu), bb[0], bb[1])
    counts = pipe.ctx.sconv_variant_counts()
    # (an eighth field, 1, marks the slot-range split of the coarse grids' plain 55-slot layers)
    assert sum(counts.values()) == 44 and all(len(k) in (7, 8) and k[5] == {"f16": 1, "bf16x3": 2, "f16x2": 3}[precision] and k[6] == 1 for k in counts)
    assert np.array_equal(pipe.get("voxel_keys0").cpu().numpy().view(np.uint64), ref["voxel_keys0"])
    for name, got in (("code", pipe.get("code")), ("values", values)):
        scale = max(1.0, float(np.abs(ref[name]).max()))
        err = float(np.abs(got.cpu().numpy().astype(np.float64) - ref[name]).max())
        print("%s %s: max err %.3e, range %.3g" % (precision, name, err, scale))
        assert err <= tol * scale, (name, err, scale)
    # deterministic
    v2 = pipe.forward(_t(pts, gpu), _t(nrm, gpu), _t(rad, gpu), bb[0], bb[1])
    assert torch.equal(values, v2)


@pytest.mark.parametrize("mode", ["bf16x3", "f16x2"])
def test_conv_plan_reuse_and_row_lists(geo, gpu, mode):
    """One ConvPlan serves every convolution over its list (plain, two-bank, importance-weighted); a plan built
    for a row list (row_perm + num_rows) writes exactly those rows; per-pair importance ignores the plan."""
    from asr_hip import ops
    idx, kidx, rs, num_inp = _csr(geo, "nb", 1)
    v = len(rs) - 1
    rng = np.random.default_rng(5)
    K, cin, ca, cb = 55, 64, 56, 8
    f = rng.standard_normal((num_inp, cin)).astype(np.float32)
    Wa = (rng.standard_normal((K, cin, ca)) * np.sqrt(2.0 / (8 * cin))).astype(np.float32)
    Wb = (rng.standard_normal((K, cin, cb)) * np.sqrt(2.0 / (8 * cin))).astype(np.float32)
    imp = rng.uniform(0.05, 1.0, size=num_inp).astype(np.float32)
    nimp = imp[idx.astype(np.int64)]
    d_idx, d_k, d_rs = _t(idx, gpu), _t(kidx, gpu), _t(rs, gpu)
    perm = ops.row_groups(d_k, d_rs)
    plan = ops.ConvPlan(K, d_idx, d_k, d_rs, row_perm=perm)
    assert 0 < plan.nbytes() < 64 * idx.size + (1 << 21)  # about the size of the list, never 16x it
    ctx = ops.context(gpu)
    ctx.sconv_variant_counts(reset=True)
    pa = ops.pack_filters(_t(Wa, gpu), mode)
    pab = ops.pack_filters(_t(Wa, gpu), mode, _t(Wb, gpu))
    with O.precise():
        ref_a = O.sparse_conv(Wa, f, idx, kidx, None, rs, False)
        ref_b = O.sparse_conv(Wb, f, idx, kidx, nimp, rs, True)
        ref_i = O.sparse_conv(Wa, f, idx, kidx, nimp, rs, True)
    a = ops.sparse_conv16(mode, pa, K, cin, ca, _t(f, gpu), d_idx, d_k, d_rs, row_perm=perm, plan=plan)
    _close(a.cpu().numpy(), ref_a)
    ab, oi = ops.sparse_conv16(mode, pab, K, cin, ca, _t(f, gpu), d_idx, d_k, d_rs, row_perm=perm, plan=plan,
                               inp_importance=_t(imp, gpu), normalize=True, cout_b=cb, return_importance=True)
    _close(ab.cpu().numpy(), np.concatenate([ref_a, ref_b], 1))
    _close(oi.cpu().numpy(), O.reduce_subarrays_sum(nimp, rs))
    ai = ops.sparse_conv16(mode, pa, K, cin, ca, _t(f, gpu), d_idx, d_k, d_rs, row_perm=perm, plan=plan,
                           inp_importance=_t(imp, gpu), normalize=True)
    _close(ai.cpu().numpy(), ref_i)
    assert all(k[6] == 1 for k in ctx.sconv_variant_counts())
    # per-pair importance: table-driven kernel, same numbers
    ctx.sconv_variant_counts(reset=True)
    an = ops.sparse_conv16(mode, pa, K, cin, ca, _t(f, gpu), d_idx, d_k, d_rs, row_perm=perm, plan=plan,
                           neighbors_importance=_t(nimp, gpu), normalize=True)
    _close(an.cpu().numpy(), ref_i)
    assert all(k[6] == 0 for k in ctx.sconv_variant_counts())
    # row list: the first 1000 rows of a shuffled order; other rows of `out` stay untouched
    order = torch.from_numpy(rng.permutation(v).astype(np.int32)).to(gpu)
    n_rows = 1000
    lplan = ops.ConvPlan(K, d_idx, d_k, d_rs, row_perm=order, num_rows=n_rows)
    out = torch.full((v, ca), 7.0, dtype=torch.float32, device=gpu)
    ops.sparse_conv16(mode, pa, K, cin, ca, _t(f, gpu), d_idx, d_k, d_rs, row_perm=order, num_rows=n_rows,
                      plan=lplan, out=out)
    rows = order[:n_rows].long().cpu().numpy()
    got = out.cpu().numpy()
    _close(got[rows], ref_a[rows])
    rest = np.setdiff1d(np.arange(v), rows)
    assert np.all(got[rest] == 7.0)
    # a plan of another list is refused
    with pytest.raises(Exception):
        ops.sparse_conv16(mode, pa, K, cin, ca, _t(f, gpu), d_idx, d_k, d_rs, row_perm=perm, plan=lplan)


@pytest.mark.parametrize("mode", ["bf16x3", "f16x2"])
def test_slot_range_split_of_the_coarse_grids(geo, gpu, mode):
    """Plain 55-slot convolutions over a grid of sconv_split_min_rows .. sconv_split_rows rows run as one block per (tile,
    slot range) + a second pass that adds the partial sums in range order (asr_conv16.hip, split_range_mask): against the
    oracle, against the unsplit kernel, and -- what a sharded forward relies on -- bit-identical whatever rows share a tile."""
    from asr_hip import ops
    level = 0
    idx, kidx, rs, num_inp = _csr(geo, "nb", level)
    v = len(rs) - 1
    ctx = ops.context(gpu)
    assert ctx.get_option("sconv_split_min_rows") <= v <= ctx.get_option("sconv_split_rows"), v
    rng = np.random.default_rng(77)
    K, cin, ca = 55, 64, 128
    f = rng.standard_normal((num_inp, cin)).astype(np.float32)
    W = (rng.standard_normal((K, cin, ca)) * np.sqrt(2.0 / (8 * cin))).astype(np.float32)
    b = (rng.standard_normal(ca) * 0.1).astype(np.float32)
    res = rng.standard_normal((v, ca)).astype(np.float32)
    d_idx, d_k, d_rs = _t(idx, gpu), _t(kidx, gpu), _t(rs, gpu)
    pk = ops.pack_filters(_t(W, gpu), mode)
    with O.precise():
        ref = np.maximum(O.sparse_conv(W, f, idx, kidx, None, rs, False) + b, 0) + res
    outs = {}
    orders = {"regrouped": ops.row_groups(d_k, d_rs),
              "shuffled": torch.from_numpy(rng.permutation(v).astype(np.int32)).to(gpu),
              "natural": torch.arange(v, dtype=torch.int32, device=gpu)}
    for name, perm in orders.items():
        plan = ops.ConvPlan(K, d_idx, d_k, d_rs, row_perm=perm)
        ctx.sconv_variant_counts(reset=True)
        out = ops.sparse_conv16(mode, pk, K, cin, ca, _t(f, gpu), d_idx, d_k, d_rs, row_perm=perm, plan=plan,
                                bias=_t(b, gpu), relu=True, residual=_t(res, gpu))
        keys = list(ctx.sconv_variant_counts())
        assert len(keys) == 1 and len(keys[0]) == 8 and keys[0][7] == 1, keys  # the split ran
        outs[name] = out.cpu().numpy()
        _close(outs[name], ref)
        del plan
    assert np.array_equal(outs["regrouped"], outs["shuffled"]) and np.array_equal(outs["regrouped"], outs["natural"])
    # a row list (a rank's owned rows): the same bits for those rows, the others untouched
    n_rows = v // 3
    lplan = ops.ConvPlan(K, d_idx, d_k, d_rs, row_perm=orders["shuffled"], num_rows=n_rows)
    part = torch.full((v, ca), 7.0, dtype=torch.float32, device=gpu)
    ops.sparse_conv16(mode, pk, K, cin, ca, _t(f, gpu), d_idx, d_k, d_rs, row_perm=orders["shuffled"], num_rows=n_rows,
                      plan=lplan, out=part, bias=_t(b, gpu), relu=True, residual=_t(res, gpu))
    rows = orders["shuffled"][:n_rows].long().cpu().numpy()
    got = part.cpu().numpy()
    assert np.array_equal(got[rows], outs["regrouped"][rows])
    assert np.all(got[np.setdiff1d(np.arange(v), rows)] == 7.0)
    # the unsplit kernel: another summation order, the same tolerance
    ctx.set_option("sconv_split_rows", 0)
    try:
        plan = ops.ConvPlan(K, d_idx, d_k, d_rs, row_perm=orders["regrouped"])
        ctx.sconv_variant_counts(reset=True)
        whole = ops.sparse_conv16(mode, pk, K, cin, ca, _t(f, gpu), d_idx, d_k, d_rs, row_perm=orders["regrouped"],
                                  plan=plan, bias=_t(b, gpu), relu=True, residual=_t(res, gpu)).cpu().numpy()
        assert all(len(k) == 7 for k in ctx.sconv_variant_counts())
    finally:
        ctx.set_option("sconv_split_rows", 32768)
    _close(whole, ref)
    _close(whole, outs["regrouped"])


@pytest.mark.parametrize("mode", ["bf16x3", "f16x2"])
def test_feature_matrix_beyond_4gb(gpu, mode):
    """Feature matrices of 4 GB and more are out of reach of buffer addressing: the launcher hands them to the
    table-driven kernel, which switches to 64-bit pointers.  9 M rows x 128 f32 (4.6 GB), neighbours spread over the
    whole matrix, against a float64 gather-and-contract on the GPU."""
    from asr_hip import ops
    g = torch.Generator(device=gpu).manual_seed(11)
    num_inp, cin, cout, K, v, per_row = 9_000_000, 128, 32, 27, 2048, 6
    f = torch.randn((num_inp, cin), generator=g, device=gpu, dtype=torch.float32)
    assert f.numel() * 4 > 2**32
    W = torch.randn((K, cin, cout), generator=g, device=gpu) * (2.0 / (per_row * cin)) ** 0.5
    idx = torch.randint(0, num_inp, (v * per_row,), generator=g, device=gpu, dtype=torch.int64)
    idx[-per_row:] = torch.arange(num_inp - per_row, num_inp, device=gpu)  # the very last rows of the matrix
    slots = torch.stack([torch.randperm(K, generator=g, device=gpu)[:per_row].sort().values for _ in range(v)]).reshape(-1)
    rs = torch.arange(0, v * per_row + 1, per_row, device=gpu, dtype=torch.int64)
    ref = torch.einsum("pc,pco->po", f[idx].double(), W.double()[slots]).reshape(v, per_row, cout).sum(1)
    ctx = ops.context(gpu)
    ctx.sconv_variant_counts(reset=True)
    out = ops.sparse_conv16(mode, ops.pack_filters(W, mode), K, cin, cout, f, idx.to(torch.int32),
                            slots.to(torch.uint8), rs)
    counts = ctx.sconv_variant_counts()
    assert len(counts) == 1 and list(counts)[0][6] == 0, counts  # table-driven kernel
    err = float((out.double() - ref).abs().max())
    assert err <= 1e-5 * max(1.0, float(ref.abs().max())), err


@pytest.mark.parametrize("mode", ["bf16x3", "f16x2"])
def test_plan_kernel_equals_table_kernel_on_random_lists(gpu, mode):
    """Random neighbour lists (empty rows, rows with every slot, shuffled entry order inside a row, row lists, padded
    strides): the plan-driven kernel and the table-driven one run the same products in the same slot order, so plain
    convolutions agree bit for bit; with importance the row sums are formed in a different order (1e-6)."""
    from asr_hip import ops
    ctx = ops.context(gpu)
    g = torch.Generator(device="cpu").manual_seed(99)
    for case in range(24):
        K = [27, 55, 9][case % 3]
        cin = [32, 64, 96, 128][case % 4]
        cout = [16, 24, 40, 128, 56][case % 5]
        v, num_inp = int(torch.randint(1, 3000, (1,), generator=g)), int(torch.randint(1, 4000, (1,), generator=g))
        lens = torch.randint(0, min(K, 14) + 1, (v,), generator=g)
        lens[torch.randint(0, v, (max(1, v // 50),), generator=g)] = K  # some full rows
        slots = torch.cat([torch.randperm(K, generator=g)[:int(n)] for n in lens]) if int(lens.sum()) else torch.zeros(0, dtype=torch.int64)
        rs = torch.zeros(v + 1, dtype=torch.int64)
        rs[1:] = torch.cumsum(lens, 0)
        idx = torch.randint(0, num_inp, (int(rs[-1]),), generator=g)
        ld = cin + [0, 4, 32][case % 3]
        fbuf = torch.randn((num_inp, ld), generator=g)
        f = fbuf.to(gpu)[:, :cin]
        W = (torch.randn((K, cin, cout), generator=g) * 0.1).to(gpu)
        imp = torch.rand(num_inp, generator=g).to(gpu) if case % 2 else None
        use_rows = case % 4 == 3
        perm = torch.randperm(v, generator=g).to(torch.int32).to(gpu) if case % 3 else None
        n_rows = max(1, v // 2) if (use_rows and perm is not None) else None
        kw = dict(inp_importance=imp, normalize=imp is not None, relu=True, row_perm=perm, num_rows=n_rows)
        pk = ops.pack_filters(W, mode)
        # waves 0: the launcher's choice (4-wave blocks at these sizes: rows through registers); 8: the 8-wave instances, whose
        # rows travel HBM -> LDS by DMA in whole lines (round 6) -- absent neighbours, padded strides and partial last tiles
        # included
        for waves in (0, 8):
            outs = []
            for plan_on in (1, 0):
                ctx.set_option("sconv_plan", plan_on)
                ctx.sconv_variant_counts(reset=True)
                out = torch.full((v, cout), -3.0, device=gpu)
                ops.sparse_conv16(mode, pk, K, cin, cout, f, idx.to(torch.int32).to(gpu), slots.to(torch.uint8).to(gpu),
                                  rs.to(gpu), out=out, force_waves=waves, **kw)
                key = list(ctx.sconv_variant_counts())[0]
                assert key[6] == plan_on and (waves == 0 or key[3] == 8), key
                outs.append(out)
            ctx.set_option("sconv_plan", 1)
            if imp is None:
                assert torch.equal(outs[0], outs[1]), (case, waves)
            else:
                assert float((outs[0] - outs[1]).abs().max()) <= 1e-6 * max(1.0, float(outs[1].abs().max())), (case, waves)


def _rel(a, b, tol=2e-6):
    """|a - b| <= tol * max|b|: tensors whose magnitude is far from 1 (1e-5 absolute would be vacuous or hopeless)"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err, scale = np.abs(a - b).max(), np.abs(b).max()
    assert err <= tol * scale, "max err %.3e of %.3e (%.2e)" % (err, scale, err / scale)


@pytest.mark.parametrize("fscale,wscale", [(1.0, 1.0), (3e7, 1e-6), (2e-9, 5e4), (1e-20, 1.0), (1.0, 1e25), (1e-20, 1e25)])
def test_f16x2_range_and_running_maximum(geo, gpu, fscale, wscale):
    """f16x2 scales each tensor by a power of two before the f16 split: activations and weights far outside the f16
    range (65504 / 6e-8) give the same fp32-class result; a wide spread inside one tensor (1e-4 .. 1e4 per column)
    too.  The epilogue keeps the running maximum of what it writes (out_absmax), the next layer's inp_absmax: the
    same bits as a pass over the output, and a chained second layer agrees with the oracle."""
    from asr_hip import ops
    idx, kidx, rs, num_inp = _csr(geo, "nb", 1)
    v = len(rs) - 1
    cin, c1, c2 = 64, 64, 32
    rng = np.random.default_rng(77)
    spread = np.exp(rng.uniform(np.log(1e-4), np.log(1e4), size=cin)).astype(np.float32)
    f = (rng.standard_normal((num_inp, cin)) * spread * fscale).astype(np.float32)
    W1 = (rng.standard_normal((55, cin, c1)) * np.sqrt(2.0 / (8 * cin)) * wscale).astype(np.float32)
    W2 = (rng.standard_normal((55, c1, c2)) * np.sqrt(2.0 / (8 * c1))).astype(np.float32)
    b1 = (rng.standard_normal(c1) * 0.1 * fscale * wscale).astype(np.float32)
    p1 = ops.pack_filters(_t(W1, gpu), "f16x2")
    p2 = ops.pack_filters(_t(W2, gpu), "f16x2")
    with O.precise():
        r1 = np.maximum(O.sparse_conv(W1, f, idx, kidx, None, rs, False) + b1, 0)
        r2 = O.sparse_conv(W2, r1, idx, kidx, None, rs, False)
    g = [_t(a, gpu) for a in (idx, kidx, rs)]
    amax_in = ops.absmax(_t(f, gpu))
    assert int(amax_in.item()) == int(np.abs(f).max().view(np.int32))
    amax1 = torch.zeros(1, dtype=torch.int32, device=gpu)
    o1 = ops.sparse_conv16("f16x2", p1, 55, cin, c1, _t(f, gpu), *g, bias=_t(b1, gpu), relu=True, inp_absmax=amax_in,
                           out_absmax=amax1)
    _rel(o1.cpu().numpy(), r1)
    assert int(amax1.item()) == int(ops.absmax(o1).item()) == int(o1.abs().max().cpu().numpy().view(np.int32))
    o2 = ops.sparse_conv16("f16x2", p2, 55, c1, c2, o1, *g, inp_absmax=amax1)
    _rel(o2.cpu().numpy(), r2)
    # without inp_absmax the entry point makes the pass itself: same bits
    o2b = ops.sparse_conv16("f16x2", p2, 55, c1, c2, o1, *g)
    assert torch.equal(o2, o2b)


def test_f16x2_zero_input(geo, gpu):
    from asr_hip import ops
    idx, kidx, rs, num_inp = _csr(geo, "nb", 2)
    W = np.random.default_rng(1).standard_normal((55, 32, 32)).astype(np.float32)
    b = np.arange(32, dtype=np.float32)
    out = ops.sparse_conv16("f16x2", ops.pack_filters(_t(W, gpu), "f16x2"), 55, 32, 32,
                            torch.zeros(num_inp, 32, device=gpu), _t(idx, gpu), _t(kidx, gpu), _t(rs, gpu), bias=_t(b, gpu))
    assert torch.equal(out.cpu(), torch.from_numpy(b).expand(len(rs) - 1, 32))
    zero_w = ops.sparse_conv16("f16x2", ops.pack_filters(torch.zeros(55, 32, 32, device=gpu), "f16x2"), 55, 32, 32,
                               torch.ones(num_inp, 32, device=gpu), _t(idx, gpu), _t(kidx, gpu), _t(rs, gpu))
    assert float(zero_w.abs().max()) == 0.0


def test_duplicate_slot_in_a_row_is_refused(gpu):
    """the 16-bit kernels keep one neighbour per (row, slot); a list that names a slot twice in one row (legal for
    open3d::sparse_conv, never produced by the reference's grids, cpp/lib/grid.cpp:99-170) is refused when its plan is
    built instead of silently losing a contribution"""
    from asr_hip import ops
    from asr_hip._lib import AsrHipError
    rng = np.random.default_rng(1)
    v, c = 64, 32
    idx = np.repeat(np.arange(v, dtype=np.int32), 2)
    kidx = np.zeros(2 * v, np.uint8)
    kidx[1::2] = 1
    kidx[11] = 0  # row 5 lists slot 0 twice
    rs = np.arange(0, 2 * v + 1, 2, dtype=np.int64)
    W = rng.standard_normal((3, c, c)).astype(np.float32)
    f = rng.standard_normal((v, c)).astype(np.float32)
    packed = ops.pack_filters(_t(W, gpu), "bf16x3")
    with pytest.raises(AsrHipError, match="slot twice"):
        ops.sparse_conv16("bf16x3", packed, 3, c, c, _t(f, gpu), _t(idx, gpu), _t(kidx, gpu), _t(rs, gpu))
    kidx[11] = 1
    out = ops.sparse_conv16("bf16x3", packed, 3, c, c, _t(f, gpu), _t(idx, gpu), _t(kidx, gpu), _t(rs, gpu))
    _close(out.cpu().numpy(), O.sparse_conv(W, f, idx, kidx, None, rs, False))

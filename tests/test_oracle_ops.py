"""CPU tests of the oracle's conv arithmetic and of the model glue."""
import os

import numpy as np
import pytest

import parity
from asr_hip import synth
from oracle import oracle as O


def test_scale_compat_matches_python_reference(golden_dir):
    """golden vectors from the reference's models/common.py:18-44 (make_scale_compat.py)"""
    g = np.load(os.path.join(golden_dir, "scale_compat.npz"))
    c = O.scale_compat(g["query_scale"], g["radii"], g["idx"], g["rs"])
    assert np.array_equal(c, g["compat"])  # tolerance of the row is 1e-6; it is bit exact


def test_window_poly6():
    """models/common_torch.py:21-22 clamp((1-r)^3, 0, 1)"""
    r = np.array([0.0, 0.25, 1.0, 1.5, -0.5], np.float32)
    assert np.allclose(O.window_poly6(r), [1.0, 0.421875, 0.0, 0.0, 1.0])


def _random_csr(rng, rows, n_in, k, max_len):
    lens = rng.integers(0, max_len + 1, size=rows)
    lens[rng.integers(0, rows)] = 0
    rs = np.zeros(rows + 1, np.int64)
    rs[1:] = np.cumsum(lens)
    idx = rng.integers(0, n_in, size=rs[-1]).astype(np.int32)
    kidx = np.concatenate([np.sort(rng.choice(k, size=l, replace=False)) for l in lens] +
                          [np.zeros(0, int)]).astype(np.uint8)
    return idx, kidx, rs


def test_sparse_conv_against_dense_einsum():
    rng = np.random.default_rng(1)
    rows, n_in, cin, cout, K = 57, 40, 12, 9, 55
    idx, kidx, rs = _random_csr(rng, rows, n_in, K, 20)
    f = rng.standard_normal((n_in, cin)).astype(np.float32)
    W = rng.standard_normal((K, cin, cout)).astype(np.float32)
    imp = rng.uniform(0.1, 1, size=len(idx)).astype(np.float32)
    for nimp, normalize in ((None, False), (imp, True), (imp, False), (None, True)):
        out = O.sparse_conv(W, f, idx, kidx, nimp, rs, normalize)
        ref = np.zeros((rows, cout))
        for r in range(rows):
            s = slice(rs[r], rs[r + 1])
            w = nimp[s] if nimp is not None else np.ones(rs[r + 1] - rs[r])
            acc = np.einsum("p,pc,pco->o", w, f[idx[s]].astype(np.float64), W[kidx[s]].astype(np.float64))
            if normalize and w.sum() != 0:
                acc = acc / w.sum()
            ref[r] = acc
        assert np.abs(out - ref).max() < 1e-4


def test_continuous_conv_properties():
    rng = np.random.default_rng(2)
    n, v, cin, cout = 200, 30, 4, 6
    pos = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    feat = rng.standard_normal((n, cin)).astype(np.float32)
    out_pos = rng.uniform(-0.5, 0.5, size=(v, 3)).astype(np.float32)
    ext = rng.uniform(0.8, 1.5, size=v).astype(np.float32)
    lens = rng.integers(0, 9, size=v)
    rs = np.zeros(v + 1, np.int64)
    rs[1:] = np.cumsum(lens)
    idx = rng.integers(0, n, size=rs[-1]).astype(np.int32)
    imp = rng.uniform(0.1, 1, size=rs[-1]).astype(np.float32)
    # partition of unity: a filter that is constant over the 64 cells turns the conv into the
    # importance weighted mean of W^T f
    Wc = rng.standard_normal((cin, cout)).astype(np.float32)
    W = np.broadcast_to(Wc, (4, 4, 4, cin, cout)).copy()
    out = O.continuous_conv(W, out_pos, ext, pos, feat, idx, imp, rs, True)
    for q in range(v):
        s = slice(rs[q], rs[q + 1])
        if rs[q] == rs[q + 1]:
            assert np.all(out[q] == 0)
            continue
        ref = (imp[s, None] * (feat[idx[s]] @ Wc)).sum(0) / imp[s].sum()
        assert np.abs(out[q] - ref).max() < 1e-5
    # a neighbour at the output position addresses the centre of the 4^3 filter: u = 1.5
    W = rng.standard_normal((4, 4, 4, cin, cout)).astype(np.float32)
    out = O.continuous_conv(W, out_pos[:1], ext[:1], out_pos[:1], feat[:1], np.zeros(1, np.int32),
                            None, np.array([0, 1], np.int64), True)
    centre = W[1:3, 1:3, 1:3].reshape(8, cin, cout).mean(0)
    assert np.abs(out[0] - feat[0] @ centre).max() < 1e-5


def test_decode_matches_numpy():
    rng = np.random.default_rng(3)
    code = rng.standard_normal((50, 32)).astype(np.float32)
    w = synth.make_weights(1, seed=3)
    sizes = rng.uniform(0.01, 0.1, size=50).astype(np.float32)
    out = O.decode(code, w["dense_decoder1.weight"], w["dense_decoder1.bias"],
                   w["dense_decoder2.weight"], w["dense_decoder2.bias"], w["dense_decoder3.weight"], sizes)
    x = np.concatenate([np.zeros((50, 3), np.float32), code], 1).astype(np.float64)
    f1 = np.maximum(x @ w["dense_decoder1.weight"].T + w["dense_decoder1.bias"], 0)
    f2 = np.maximum(f1 @ w["dense_decoder2.weight"].T + w["dense_decoder2.bias"], 0)
    ref = f2 @ w["dense_decoder3.weight"].T
    ref[:, 0] *= sizes
    assert np.abs(out - ref).max() < 1e-5


@pytest.mark.parametrize("tag", ["d4_3k", "d1_2k"])
def test_network_restatement_matches_reference_model_fixture(golden_dir, tag):
    """tests/parity.py (numpy restatement of the unet wiring) reproduces what the reference's own
    net_definitions_torch.py computed over the same ops (fixture made by make_unet_fixture.py)"""
    fx = np.load(os.path.join(golden_dir, "unet_%s.npz" % tag))
    d = int(fx["channel_div"])
    weights = synth.make_weights(d, seed=int(fx["seed"]))
    item = {k[5:]: fx[k] for k in fx.files if k.startswith("geom_")}
    out = parity.oracle_network(item, fx["points"], fx["normals"], weights)
    assert np.array_equal(out["feats1"], fx["out_feats1"])
    assert np.array_equal(out["importance"], fx["out_importance"])
    assert np.abs(out["code"] - fx["out_code"]).max() < 1e-6
    if d == 1:
        assert np.abs(out["values"] - fx["out_values"]).max() < 1e-6
        assert fx["out_values"].std(0).min() > 0.05  # non-degenerate
    # geometry inside the fixture is what the oracle builds from the stored inputs
    geo = parity.oracle_geometry(fx["points"], fx["radii"], fx["bb_min"], fx["bb_max"])
    for k in item:
        assert np.array_equal(geo[k], item[k]), k


def test_every_layer_of_the_reference_graph_separately():
    """53 SpecialSparseConv calls recorded from the reference's own model code: the oracle reproduces each
    one from the call's own inputs (bit for bit: the fixture was made with the oracle's ops under the
    reference's glue, so this pins shapes, CSR choice, importance indexing and bias / ReLU per layer)"""
    import layer_fixture
    layers = layer_fixture.load()
    assert len(layers) == 53 and sum(l["name"] == "sparseconv_down3.conv1a" for l in layers) == 2
    for l in layers:
        idx, kidx, rs = l["csr"]
        nimp = l["imp"][idx.astype(np.int64)] if l["imp"] is not None else None
        out = O.sparse_conv(l["kernel"], l["inp"], idx, kidx, nimp, rs, l["normalize"])
        out = np.maximum(out + l["bias"], 0)
        assert np.array_equal(out, l["out"]), l["name"]
        if nimp is not None:
            assert np.array_equal(O.reduce_subarrays_sum(nimp, rs), l["oimp"]), l["name"]


def test_dense_open3d_style_evaluation_equals_the_pairwise_one():
    """O.dense(): the [32][K*cin] x [K*cin][cout] evaluation Open3D's CPU op performs (bench.py's cpu_baseline runs
    under it) gives the pairwise result up to the order of the fp32 sums"""
    rng = np.random.default_rng(1)
    v, cin, cout = 700, 12, 20
    lens = rng.integers(1, 12, size=v)
    rs = np.zeros(v + 1, np.int64)
    rs[1:] = np.cumsum(lens)
    idx = rng.integers(0, v, size=rs[-1]).astype(np.int32)
    kidx = np.concatenate([np.sort(rng.choice(55, size=l, replace=False)) for l in lens]).astype(np.uint8)
    f = rng.standard_normal((v, cin)).astype(np.float32)
    W = rng.standard_normal((55, cin, cout)).astype(np.float32)
    imp = rng.uniform(0.1, 1, size=rs[-1]).astype(np.float32)
    for nimp, normalize in ((None, False), (imp, True)):
        a = O.sparse_conv(W, f, idx, kidx, nimp, rs, normalize)
        with O.dense():
            b = O.sparse_conv(W, f, idx, kidx, nimp, rs, normalize)
        assert np.abs(a - b).max() <= 1e-5 * max(1.0, np.abs(a).max())


@pytest.mark.parametrize("tag,n", [("d4_3k", 3000), ("d1_2k", 2000), ("layers_d4_1k", 1000)])
def test_fixture_inputs_come_out_of_the_committed_generator(golden_dir, tag, n):
    """provenance chain of tests/golden/unet_*.npz: the stored inputs are what tests/golden/make_unet_fixture.py
    derives TODAY from (n, seed) -- scan_cloud, 24-NN radii, bounding box -- so a change of the generators after
    the fixtures were written (which would make the committed script stop reproducing them) fails here.  The
    outputs need /root/reference and are re-derived by running that script in the build container."""
    from asr_hip import synth
    fx = np.load(os.path.join(golden_dir, "unet_%s.npz" % tag))
    seed = int(fx["seed"])
    pts, nrm = synth.scan_cloud(n, seed=seed, device="cpu")
    assert np.array_equal(pts.numpy(), fx["points"]) and np.array_equal(nrm.numpy(), fx["normals"])
    assert np.array_equal(synth.knn_radii(fx["points"], 24), fx["radii"])
    bb_min, bb_max = synth.bounding_box(fx["points"], 0.1)
    assert np.array_equal(bb_min, fx["bb_min"]) and np.array_equal(bb_max, fx["bb_max"])

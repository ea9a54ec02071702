"""Oracle-side whole-path forward (numpy + oracle/libasr_oracle.so), the checker for the GPU
parity tests, smoke() and bench.py's cpu_baseline.  TEST INFRASTRUCTURE, never the product.

The reference's model file cannot travel to the GPU box, so the network wiring is restated here
following models/v0/net_definitions_torch.py (line numbers in the comments); the restatement is
itself pinned against tests/golden/unet_*.npz, which were produced by the reference's own model
code (tests/test_oracle_ops.py::test_network_restatement_matches_reference_model_fixture).
"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "adaptive-surface-reconstruction_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import oracle as O  # noqa: E402


def assert_close(a, b, atol=1e-5, rtol=1e-5):
    """the north_star tolerance, per element and without any magnitude scaling:
    |a - b| <= 1e-5 + 1e-5 |b|.  Used for the implicit values and for every single operator."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b)
    assert np.all(err <= atol + rtol * np.abs(b)), \
        "max err %.3e (ref max %.3e)" % (err.max() if err.size else 0.0, np.abs(b).max() if b.size else 0.0)


def assert_close_scaled(a, b, tol=1e-5):
    """|a - b| <= 1e-5 * max(1, max|b|): for whole-path outputs whose magnitude is not O(1) -- the deep
    feature map `code` (after 53 fp32 convolutions, activations reach 10..40 with the seeded test weights)
    and the `values` decoded from it on clouds of >= 10^4 points.  The rounding error of such a tensor is
    relative to the magnitude of the activations that were summed, not to the (possibly tiny) element it
    lands on: the fp32 oracle itself differs from the double-accumulating oracle (O.precise) by 4e-6 of
    the tensor's maximum on a 6 k-point cloud.  Every single operator, and the whole path on the
    reference-model fixtures, is checked with the plain assert_close."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(1.0, float(np.abs(b).max())) if b.size else 1.0
    err = np.abs(a - b)
    assert np.all(err <= tol * scale), "max err %.3e (ref max %.3e)" % (err.max() if err.size else 0.0, scale)


def pass_fraction(a, b, atol=1e-5, rtol=1e-5):
    """share of the elements that meet the north_star tolerance |a - b| <= 1e-5 + 1e-5 |b| one by one; printed by
    the whole-path tests whose assertion is range-scaled (assert_close_scaled), so that the claim is visible"""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if not b.size:
        return 1.0
    return float(np.mean(np.abs(a - b) <= atol + rtol * np.abs(b)))


def oracle_geometry(points, radii, bb_min, bb_max, radius_scale=1.0, max_depth=21, timings=None):
    """input_dict of cpp/lib/asr.cpp:143-312"""
    o = O.Oracle()
    t0 = time.time()
    o.build_octree(points, radii, bb_min, bb_max, radius_scale, max_depth)
    t1 = time.time()
    grids = o.create_grids(5)
    t2 = time.time()
    item = {"nodes": o.nodes}
    for i, g in enumerate(grids):
        for k, v in g.items():
            item[k + str(i)] = v
    idx, dist, rs, compat = o.radius_search(points, radii, grids[0]["voxel_centers"],
                                            grids[0]["voxel_sizes"])
    t3 = time.time()
    item["aggregation_neighbors_index"] = idx
    item["aggregation_neighbors_dist"] = dist
    item["aggregation_row_splits"] = rs
    item["aggregation_scale_compat"] = compat
    if timings is not None:
        timings.update(octree=t1 - t0, grids=t2 - t1, aggregation_search=t3 - t2)
    return item


def _conv(w, name, feats, nb, importance=None, normalize=False, residual=None):
    """SpecialSparseConv.forward, models/common_torch.py:95-148 (bias + ReLU included)"""
    idx, kidx, rs = nb
    nimp = None
    out_imp = None
    if importance is not None:
        nimp = importance[idx.astype(np.int64)]             # :124-126
        out_imp = O.reduce_subarrays_sum(nimp, rs)          # :127-128
    out = O.sparse_conv(w[name + ".kernel"], feats, idx, kidx, nimp, rs, normalize)  # :133-142
    out = out + w[name + ".bias"]                           # :144-145
    out = np.maximum(out, 0)                                # :146
    if residual is not None:
        out = out + residual
    return out.astype(np.float32), out_imp


def _enc_block(w, name, feats, nb, importance):
    """SparseConvBlock.forward, normalized_channels < output_channels branch
    (net_definitions_torch.py:278-287)"""
    a, _ = _conv(w, name + ".conv1a", feats, nb)
    b, out_imp = _conv(w, name + ".conv1b", feats, nb, importance, True)
    f = np.concatenate([a, b], axis=-1)
    for i in (2, 3, 4):
        f, _ = _conv(w, name + ".conv%d" % i, f, nb)
    return f, out_imp


def _dec_block(w, name, feats, nb):
    """SparseConvBlock.forward without importance (:296-302)"""
    f = feats
    for i in (1, 2, 3, 4):
        f, _ = _conv(w, name + ".conv%d" % i, f, nb)
    return f


def _down(w, name, feats, nb, importance):
    """SparseConvTransitionBlock.forward (:373-379)"""
    a, _ = _conv(w, name + ".conv1a", feats, nb)
    b, out_imp = _conv(w, name + ".conv1b", feats, nb, importance, True)
    return np.concatenate([a, b], axis=-1), out_imp


def oracle_network(item, points, normals, weights, scale_sdf=True, timings=None):
    w = weights
    t0 = time.time()
    # aggregate (:640-653, 72-120)
    feats = np.concatenate([normals, np.ones((len(points), 1), np.float32)], 1)  # asr.cpp:168-176
    imp_pairs = (item["aggregation_scale_compat"] *
                 O.window_poly6(item["aggregation_neighbors_dist"])).astype(np.float32)  # :107
    feats1 = O.continuous_conv(w["cconv_block_in.conv1.kernel"], item["voxel_centers0"],
                               item["voxel_sizes0"], points, feats,
                               item["aggregation_neighbors_index"], imp_pairs,
                               item["aggregation_row_splits"], True)
    feats1 = np.maximum(feats1 + w["cconv_block_in.conv1.bias"], 0).astype(np.float32)
    t1 = time.time()
    # unet (:535-638)
    nb = [(item["neighbors_index%d" % i], item["neighbors_kernel_index%d" % i],
           item["neighbors_row_splits%d" % i]) for i in range(5)]
    nb_up = [(item["up_neighbors_index%d" % i], item["up_neighbors_kernel_index%d" % i],
              item["up_neighbors_row_splits%d" % i]) for i in range(4)]
    nb_down = []
    for i in range(4):  # :548-559
        idx, rs, attr = O.invert_neighbors_list(len(item["voxel_sizes%d" % (i + 1)]), *nb_up[i][:1],
                                                nb_up[i][2], nb_up[i][1])
        nb_down.append((idx, attr, rs))
    out = {}
    # B.2: the per-pair importance array is indexed with voxel indices (:572-578)
    f2, imp = _enc_block(w, "sparseconv_encblock0", feats1, nb[0], imp_pairs)
    f3, imp = _down(w, "sparseconv_down1", f2, nb_down[0], imp)
    f4, imp = _enc_block(w, "sparseconv_encblock1", f3, nb[1], imp)
    f5, imp = _down(w, "sparseconv_down2", f4, nb_down[1], imp)
    f6, imp = _enc_block(w, "sparseconv_encblock2", f5, nb[2], imp)
    f7, imp = _down(w, "sparseconv_down3", f6, nb_down[2], imp)
    f8, imp = _enc_block(w, "sparseconv_encblock3", f7, nb[3], imp)
    f9, imp = _down(w, "sparseconv_down3", f8, nb_down[3], imp)  # down3 re-used (:596-598)
    f10, imp = _enc_block(w, "sparseconv_encblock4", f9, nb[4], imp)
    f11, _ = _conv(w, "sparseconv_up3.conv1", f10, nb_up[3])
    f13 = _dec_block(w, "sparseconv_decblock3", np.concatenate([f11, f8], -1), nb[3])  # :617-618
    f14, _ = _conv(w, "sparseconv_up2.conv1", f13, nb_up[2])
    f16 = _dec_block(w, "sparseconv_decblock2", np.concatenate([f14, f6], -1), nb[2])
    f17, _ = _conv(w, "sparseconv_up1.conv1", f16, nb_up[1])
    f19 = _dec_block(w, "sparseconv_decblock1", np.concatenate([f17, f4], -1), nb[1])
    f20, _ = _conv(w, "sparseconv_up0.conv1", f19, nb_up[0])
    f21 = (f20 + f2).astype(np.float32)  # residual skip (:631-633)
    code = _dec_block(w, "sparseconv_decblock0", f21, nb[0])
    t2 = time.time()
    # decode + sdf scale (:655-666, asr.cpp:324-336)
    values = O.decode(code, w["dense_decoder1.weight"], w["dense_decoder1.bias"],
                      w["dense_decoder2.weight"], w["dense_decoder2.bias"],
                      w["dense_decoder3.weight"], item["voxel_sizes0"] if scale_sdf else None)
    t3 = time.time()
    if timings is not None:
        timings.update(continuous_conv=t1 - t0, unet=t2 - t1, decode=t3 - t2)
    out.update(feats1=feats1, importance=imp_pairs, code=code, values=values, feats2=f2, feats10=f10)
    return out


def oracle_forward(points, normals, radii, bb_min, bb_max, weights, scale_sdf=True, timings=None):
    item = oracle_geometry(points, radii, bb_min, bb_max, timings=timings)
    out = oracle_network(item, points, normals, weights, scale_sdf, timings)
    out.update(item)
    return out


def save_torchscript_weights(weights, path):
    """a TorchScript archive whose state_dict carries the tensor names of the reference's model.pt
    (/root/reference/models/v0/convert_tf2torchscript.py:85-122, loaded at cpp/lib/asr.cpp:138-139):
    nested parameter-holder modules, scripted and saved"""
    import torch

    class Holder(torch.nn.Module):
        pass

    root = Holder()
    for name, arr in weights.items():
        parts = name.split(".")
        m = root
        for part in parts[:-1]:
            if not hasattr(m, part):
                m.add_module(part, Holder())
            m = getattr(m, part)
        m.register_parameter(parts[-1], torch.nn.Parameter(torch.from_numpy(np.array(arr, np.float32)),
                                                           requires_grad=False))
    torch.jit.script(root).save(path)


# ---- contouring test field ------------------------------------------------------------------------
SPHERE_MESH_PIN = (1113, 2258, "ce9b39826aeff585")


def sphere_field(n, seed=0, noise=0.0):
    """analytic values on the adaptive grid of a sphere cloud: [signed distance to the unit sphere,
    |sdf| / voxel size]; `noise` (in voxel sizes) roughens the field so that fans, quads and
    boundary edges all occur"""
    from asr_hip import synth
    pts, _ = synth.sphere_cloud(n, seed)
    rad = synth.knn_radii(pts, 24)
    bb = synth.bounding_box(pts, 0.1)
    o = O.Oracle()
    o.build_octree(pts, rad, *bb)
    g = o.create_grids(1)[0]
    du = o.create_dual_vertex_indices().astype(np.int64)
    c, vs = g["voxel_centers"], g["voxel_sizes"]
    sd = (np.linalg.norm(c, axis=1) - 1.0).astype(np.float32)
    if noise:
        sd = sd + np.random.default_rng(seed + 7).normal(0, noise, sd.shape).astype(np.float32) * vs
    values = np.stack([sd, np.abs(sd) / vs], 1).astype(np.float32)
    return g, du, values

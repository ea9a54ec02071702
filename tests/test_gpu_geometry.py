"""GPU parity, integer / geometry rows (a1-a8, a11): HIP path through the C ABI vs the oracle,
bit exact (indices, keys, centres, squared distances) -- compat within 1e-6."""
import ctypes

import numpy as np
import pytest
import torch

import parity
from asr_hip import _lib, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _u64(t):
    return t.cpu().numpy().view(np.uint64)


def _cloud(kind, n, seed):
    if kind == "sphere":
        pts, nrm = synth.sphere_cloud(n, seed)
    elif kind == "lattice":  # grid-sampled cloud: many exactly equal distances (ties at the k-th neighbour)
        m = int(round(n ** (1 / 3)))
        g = np.stack(np.meshgrid(*[np.arange(m)] * 3, indexing="ij"), -1).reshape(-1, 3)
        g = g[np.random.default_rng(seed).permutation(len(g))]
        pts = (g * np.float32(0.0625)).astype(np.float32)
        nrm = np.tile(np.float32([0, 0, 1]), (len(pts), 1))
        n = len(pts)
    else:
        p, q = synth.scan_cloud(n, seed=seed, device="cpu", density_variance=10.0 if kind == "mixed" else 1.0)
        pts, nrm = p.numpy(), q.numpy()
    rad = synth.knn_radii(pts, min(24, n))
    bb = synth.bounding_box(pts, 0.1)
    return pts, nrm, rad, bb


CASES = [("sphere", 50000, 0), ("scan", 20000, 1), ("mixed", 30000, 2), ("scan", 300, 3)]


@pytest.fixture(scope="module", params=CASES, ids=lambda c: "%s-%d" % (c[0], c[1]))
def case(request, gpu):
    from asr_hip import ops
    kind, n, seed = request.param
    pts, nrm, rad, bb = _cloud(kind, n, seed)
    o = O.Oracle()
    o.build_octree(pts, rad, *bb)
    grids = o.create_grids(5)
    frame = _lib.frame_init(*bb)
    d = dict(pts=pts, rad=rad, bb=bb, o=o, grids=grids, frame=frame, ops=ops,
             tpts=torch.from_numpy(pts).to(gpu), trad=torch.from_numpy(rad).to(gpu))
    return d


def test_point_keys_bit_exact(case):
    keys = case["ops"].point_keys(case["frame"], case["tpts"], case["trad"])
    assert np.array_equal(_u64(keys), case["o"].point_keys(case["pts"], case["rad"]))
    k5 = case["ops"].point_keys(case["frame"], case["tpts"], case["trad"], 1.5, 5)
    assert np.array_equal(_u64(k5), case["o"].point_keys(case["pts"], case["rad"], 1.5, 5))


def test_octree_nodes_and_leaves_bit_exact(case):
    nodes, leaves = case["ops"].octree_build(case["frame"], case["tpts"], case["trad"])
    assert np.array_equal(_u64(nodes), case["o"].nodes)
    assert np.array_equal(_u64(leaves), case["o"].leaves)


def test_grid_hierarchy_bit_exact(case, gpu):
    ops, grids = case["ops"], case["grids"]
    keys = torch.from_numpy(grids[0]["voxel_keys"].view(np.int64)).to(gpu)
    for i, g in enumerate(grids):
        assert np.array_equal(_u64(keys), g["voxel_keys"])
        centers, sizes = ops.voxel_info(case["frame"], keys)
        assert np.array_equal(centers.cpu().numpy(), g["voxel_centers"])
        assert np.array_equal(sizes.cpu().numpy(), g["voxel_sizes"])
        idx, kidx, rs = ops.grid_neighbors(keys)
        assert np.array_equal(rs.cpu().numpy(), g["neighbors_row_splits"])
        assert np.array_equal(idx.cpu().numpy(), g["neighbors_index"])
        assert np.array_equal(kidx.cpu().numpy(), g["neighbors_kernel_index"])
        if i < 4:
            nxt, up_idx, up_kidx, up_rs = ops.grid_coarsen(keys)
            assert np.array_equal(up_idx.cpu().numpy(), g["up_neighbors_index"])
            assert np.array_equal(up_kidx.cpu().numpy(), g["up_neighbors_kernel_index"])
            assert np.array_equal(up_rs.cpu().numpy(), g["up_neighbors_row_splits"])
            n_coarse = len(grids[i + 1]["voxel_keys"])
            d_idx, d_rs, d_attr = ops.invert_neighbors_list(n_coarse, up_idx, up_rs, up_kidx)
            o_idx, o_rs, o_attr = O.invert_neighbors_list(n_coarse, g["up_neighbors_index"],
                                                          g["up_neighbors_row_splits"],
                                                          g["up_neighbors_kernel_index"])
            assert np.array_equal(d_idx.cpu().numpy(), o_idx)
            assert np.array_equal(d_rs.cpu().numpy(), o_rs)
            assert np.array_equal(d_attr.cpu().numpy(), o_attr)
            keys = nxt


@pytest.mark.parametrize("hash_level", [-1, 0, 8], ids=["auto", "binary-search", "hash-to-8"])
def test_multi_radius_search(case, gpu, hash_level):
    """hash_level: finest level of the search's cell hash table; finer query levels find their cells by binary search in
    the Morton-sorted codes (auto: levels whose cells hold fewer than two points; 0: every level)"""
    g = case["grids"][0]
    ctx = case["ops"].context(gpu)
    ctx.set_option("search_hash_level", hash_level)
    try:
        idx, dist, rs, compat = case["ops"].multi_radius_search(
            case["frame"], case["tpts"], case["trad"], torch.from_numpy(g["voxel_centers"]).to(gpu),
            torch.from_numpy(g["voxel_sizes"]).to(gpu))
    finally:
        ctx.set_option("search_hash_level", -1)
    o_idx, o_dist, o_rs, o_compat = case["o"].radius_search(case["pts"], case["rad"],
                                                            g["voxel_centers"], g["voxel_sizes"])
    assert np.array_equal(rs.cpu().numpy(), o_rs)
    assert np.array_equal(idx.cpu().numpy(), o_idx)          # membership AND (distance, index) order
    assert np.array_equal(dist.cpu().numpy(), o_dist)        # squared distances, bit exact
    assert np.abs(compat.cpu().numpy() - o_compat).max() <= 1e-6


def test_reference_module_mirror(case):
    """adaptivesurfacereconstruction.create_octree / create_grids_from_octree
    (cpp/pybind/module.cpp:372-441) return what the oracle builds"""
    import adaptivesurfacereconstruction as asr
    tree = asr.create_octree(case["pts"], case["rad"], case["bb"][0], case["bb"][1], radius_scale=1,
                             grow_steps=0, max_depth=21)
    grids = asr.create_grids_from_octree(tree, 5, voxel_info_all_levels=True)
    assert len(grids) == 5
    for g, og in zip(grids, case["grids"]):
        assert set(g) == set(og)  # empty arrays omitted like the reference (up_* on grid 4)
        for k in og:
            assert g[k].dtype == og[k].dtype and np.array_equal(g[k], og[k]), k
    g2 = asr.create_grids_from_octree(tree, 2)
    assert "voxel_centers" in g2[0] and "voxel_centers" not in g2[1]
    with pytest.raises(ValueError):
        asr.create_octree(case["pts"][:, :2], case["rad"], case["bb"][0], case["bb"][1])
    tree = asr.create_octree(case["pts"], case["rad"], case["bb"][0], case["bb"][1])
    duals = asr.create_dual_vertex_indices(tree)
    ref = case["o"].create_dual_vertex_indices()
    assert duals.dtype == np.uint64 and np.array_equal(duals, ref.astype(np.uint64))


# ---- edge cases ---------------------------------------------------------------------------------
def test_edge_single_point_and_duplicates(gpu):
    from asr_hip import ops
    for pts, rad in (
            (np.array([[0.3, 0.2, 0.1]], np.float32), np.array([0.05], np.float32)),
            (np.tile(np.array([[0.3, 0.2, 0.1]], np.float32), (100, 1)), np.full(100, 0.01, np.float32)),
            (np.array([[0.1, 0.1, 0.1], [0.9, 0.9, 0.9]], np.float32), np.array([5.0, 1e-7], np.float32))):
        bb = (np.zeros(3, np.float32), np.ones(3, np.float32))
        o = O.Oracle()
        o.build_octree(pts, rad, *bb)
        frame = _lib.frame_init(*bb)
        nodes, leaves = ops.octree_build(frame, torch.from_numpy(pts).to(gpu), torch.from_numpy(rad).to(gpu))
        assert np.array_equal(_u64(nodes), o.nodes) and np.array_equal(_u64(leaves), o.leaves)
        grids = o.create_grids(5)
        idx, kidx, rs = ops.grid_neighbors(leaves)
        assert np.array_equal(idx.cpu().numpy(), grids[0]["neighbors_index"])


def test_edge_points_outside_bbox_and_on_max_face(gpu):
    """points outside the box are skipped (octree.cpp:248-251); a point on the max face maps to
    INVALID_KEY, which the reference inserts (UB) and this build skips"""
    from asr_hip import ops
    pts, nrm, rad, bb = _cloud("scan", 5000, 7)
    bb_small = (bb[0] + np.float32(0.8), bb[1] - np.float32(0.6))
    pts2 = np.concatenate([pts, bb_small[1][None, :]]).astype(np.float32)
    rad2 = np.concatenate([rad, [0.01]]).astype(np.float32)
    o = O.Oracle()
    o.build_octree(pts2, rad2, *bb_small)
    frame = _lib.frame_init(*bb_small)
    nodes, leaves = ops.octree_build(frame, torch.from_numpy(pts2).to(gpu), torch.from_numpy(rad2).to(gpu))
    assert np.array_equal(_u64(nodes), o.nodes) and np.array_equal(_u64(leaves), o.leaves)
    assert 0 not in set(_u64(nodes).tolist())
    # the radius search still sees every point, also those outside the root cube
    g = o.create_grids(1)[0]
    a = ops.multi_radius_search(frame, torch.from_numpy(pts2).to(gpu), torch.from_numpy(rad2).to(gpu),
                                torch.from_numpy(g["voxel_centers"]).to(gpu),
                                torch.from_numpy(g["voxel_sizes"]).to(gpu))
    b = o.radius_search(pts2, rad2, g["voxel_centers"], g["voxel_sizes"], brute=True)
    assert np.array_equal(a[0].cpu().numpy(), b[0]) and np.array_equal(a[2].cpu().numpy(), b[2])


def test_edge_max_depth_and_empty(gpu):
    from asr_hip import ops
    pts, nrm, rad, bb = _cloud("scan", 4000, 9)
    frame = _lib.frame_init(*bb)
    for depth in (0, 1, 3, 6):
        o = O.Oracle()
        o.build_octree(pts, rad, *bb, max_depth=depth)
        nodes, leaves = ops.octree_build(frame, torch.from_numpy(pts).to(gpu), torch.from_numpy(rad).to(gpu),
                                         1.0, depth)
        assert np.array_equal(_u64(leaves), o.leaves), depth
    # no point inside the box -> empty tree
    far = (pts + 100).astype(np.float32)
    nodes, leaves = ops.octree_build(frame, torch.from_numpy(far).to(gpu), torch.from_numpy(rad).to(gpu))
    assert nodes.numel() == 0 and leaves.numel() == 0
    # empty key list
    idx, kidx, rs = ops.grid_neighbors(torch.zeros(0, dtype=torch.int64, device=gpu))
    assert idx.numel() == 0 and rs.cpu().tolist() == [0]


def test_multi_radius_search_long_rows(gpu):
    """a few huge radii: rows with thousands of members, (distance, index) order incl. ties"""
    from asr_hip import ops
    rng = np.random.default_rng(4)
    pts = rng.uniform(0.05, 0.95, size=(20000, 3)).astype(np.float32)
    pts[100:200] = pts[0:100]  # exact duplicates -> distance ties, ordered by index
    rad = np.full(len(pts), 0.01, np.float32)
    bb = (np.zeros(3, np.float32), np.ones(3, np.float32))
    frame = _lib.frame_init(*bb)
    o = O.Oracle()
    o.build_octree(pts, rad, *bb)
    centers = np.array([[0.5, 0.5, 0.5], [0.25, 0.25, 0.25], [0.9, 0.1, 0.5], [0.5, 0.5, 0.5]], np.float32)
    sizes = np.array([0.5, 0.25, 0.03125, 1.0], np.float32)
    a = ops.multi_radius_search(frame, torch.from_numpy(pts).to(gpu), torch.from_numpy(rad).to(gpu),
                                torch.from_numpy(centers).to(gpu), torch.from_numpy(sizes).to(gpu))
    b = o.radius_search(pts, rad, centers, sizes, brute=True)
    assert b[2][-1] > 15000
    for x, y in zip(a[:3], b[:3]):
        assert np.array_equal(x.cpu().numpy(), y)


def test_kdtree_prefilter_queries(gpu):
    """KDTree mirror (cpp/pybind/module.cpp:455-489): exact k-th neighbour radius, inlier vote and
    radius neighbour counts vs brute force"""
    import adaptivesurfacereconstruction as asr
    for kind, n, seed in (("scan", 6000, 5), ("mixed", 4000, 6), ("sphere", 3000, 7), ("lattice", 1728, 8)):
        pts, nrm, rad, bb = _cloud(kind, n, seed)
        if kind == "sphere":
            pts[:50] = pts[50:100]  # duplicates: zero distances and ties
        tree = asr.KDTree(pts)
        for k in (1, 24, 60):
            r = tree.compute_k_radius(k)
            ref = O.knn_radius(pts, k)
            assert r.dtype == np.float32 and np.array_equal(r, ref), (kind, k, np.abs(r - ref).max())
        r24 = O.knn_radius(pts, 24)
        cnt = tree.compute_radius_neighbors(r24 * 1.3)
        assert cnt == O.radius_count(pts, (r24 * 1.3).astype(np.float32)).tolist()
        inl = tree.compute_inlier(r24, radius_fraction=0.9, k=24, outlier_threshold=3)
        # brute force vote over exactly the k nearest, ties at the k-th distance by ascending index
        # (cpp/lib/nsearch.cpp:62-80 votes over the k results of knnSearch)
        d2 = ((pts[:, None, :] - pts[None, :, :]) ** 2)
        d2 = (d2[..., 0] + d2[..., 1]) + d2[..., 2]
        order = np.argsort(d2, axis=1, kind="stable")[:, :24]  # stable: equal distances keep index order
        votes = (r24[order] < (r24 * np.float32(0.9))[:, None]).sum(1)
        assert np.array_equal(inl, votes < 3)
    with pytest.raises(ValueError):
        asr.KDTree(pts[:, :2])


def test_error_behaviour(gpu):
    from asr_hip import ops
    with pytest.raises(_lib.AsrHipError):
        ops.point_keys(_lib.frame_init([0, 0, 0], [1, 1, 1]), torch.zeros(4, 3), torch.zeros(4))  # CPU tensors
    ctx = ops.context()
    rc = ctx.lib.asr_hip_octree_build(ctx._h, None, None, None, ctypes.c_int64(3), ctypes.c_float(1), 21,
                                      None, None)
    assert rc == 1 and b"null" in ctx.lib.asr_hip_last_error(ctx._h)


# ---- "next" rows D.2 / D.3: contouring and component filter ---------------------------------------
def _gpu_mesh(values, du, centers, thr=1.0):
    from asr_hip import ops
    dev = torch.device("cuda:0")
    v, t = ops.contour(torch.from_numpy(values).to(dev), torch.from_numpy(du).to(dev),
                       torch.from_numpy(centers).to(dev), thr)
    return v.cpu().numpy(), t.cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("n,seed,noise", [(5000, 1, 0.0), (5000, 1, 0.3), (20000, 0, 0.0), (30000, 3, 1.0)])
def test_contour_bit_exact(n, seed, noise):
    """vertices (float bits) and triangle corner order equal to the serial restatement"""
    g, du, values = parity.sphere_field(n, seed=seed, noise=noise)
    want_v, want_t = O.create_triangle_mesh(values, du, g["voxel_centers"], 1.0)
    got_v, got_t = _gpu_mesh(values, du, g["voxel_centers"])
    assert got_v.shape == want_v.shape and got_t.shape == want_t.shape
    assert np.array_equal(got_v.view(np.uint32), want_v.view(np.uint32))
    assert np.array_equal(got_t, want_t)


@pytest.mark.gpu
def test_contour_golden_pin_and_edge_cases():
    import hashlib
    g, du, values = parity.sphere_field(5000, seed=1, noise=0.3)
    v, t = _gpu_mesh(values, du, g["voxel_centers"])
    h = hashlib.sha256(v.tobytes() + t.tobytes()).hexdigest()
    assert (v.shape[0], t.shape[0], h[:16]) == parity.SPHERE_MESH_PIN
    far = values.copy()
    far[:, 1] = 5.0
    v0, t0 = _gpu_mesh(far, du, g["voxel_centers"])
    assert v0.shape == (0, 3) and t0.shape == (0, 3)
    v1, t1 = _gpu_mesh(values, du[:0], g["voxel_centers"])
    assert v1.shape == (0, 3) and t1.shape == (0, 3)
    # threshold gate partially active
    want = O.create_triangle_mesh(values, du, g["voxel_centers"], 0.4)
    got = _gpu_mesh(values, du, g["voxel_centers"], 0.4)
    assert np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32)) and np.array_equal(got[1], want[1])


@pytest.mark.gpu
@pytest.mark.parametrize("keep_n,min_size", [(1, 3), (2, 3), (3, 1), (2**63 - 1, 3), (2**63 - 1, 10), (5, 40)])
def test_remove_components_bit_exact(keep_n, min_size):
    from asr_hip import ops
    g, du, values = parity.sphere_field(20000, seed=2, noise=1.0)  # noisy field: many small components
    v, t = O.create_triangle_mesh(values, du, g["voxel_centers"], 1.0)
    want_v, want_t = O.remove_connected_components(v, t, keep_n, min_size)
    dev = torch.device("cuda:0")
    got_v, got_t = ops.remove_components(torch.from_numpy(v).to(dev), torch.from_numpy(t).to(dev), keep_n, min_size)
    assert np.array_equal(got_v.cpu().numpy().view(np.uint32), want_v.view(np.uint32))
    assert np.array_equal(got_t.cpu().numpy(), want_t)


@pytest.mark.gpu
def test_remove_components_small_cases():
    import adaptivesurfacereconstruction as asr
    v = np.arange(33, dtype=np.float32).reshape(11, 3)
    t = np.array([[0, 1, 2], [3, 4, 5], [4, 5, 6], [7, 8, 9]], np.int32)
    for keep_n, min_size in [(1, 3), (2, 3), (2**63 - 1, 3), (2**63 - 1, 4), (2**63 - 1, 1), (0, 1)]:
        want = O.remove_connected_components(v, t, keep_n, min_size)
        got = asr.remove_connected_components(v, t, keep_n, min_size)
        assert np.array_equal(got["vertices"], want[0]) and np.array_equal(got["triangles"], want[1])
        assert got["vertices"].dtype == np.float32 and got["triangles"].dtype == np.int32
    got = asr.remove_connected_components(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32), 1)
    assert got["vertices"].shape == (0, 3) and got["triangles"].shape == (0, 3)
    with pytest.raises(ValueError):
        asr.remove_connected_components(np.zeros((4, 2), np.float32), t, 1)
    with pytest.raises(RuntimeError):
        asr.remove_connected_components(v[:5], t, 1)  # triangle index out of range


def test_non_finite_points_are_rejected(gpu):
    """the reference has undefined behaviour for inf / nan coordinates; here: a clean error (and no
    multi-minute neighbour search)"""
    import adaptivesurfacereconstruction as asr
    pts, nrm, rad, bb = _cloud("sphere", 3000, 9)
    bad = pts.copy()
    bad[17, 1] = np.inf
    with pytest.raises(RuntimeError):
        asr.create_octree(bad, rad, bb[0], bb[1])
    with pytest.raises(RuntimeError):
        asr.KDTree(np.where(np.isfinite(bad), bad, np.float32(np.nan))).compute_k_radius(8)


@pytest.mark.gpu
def test_knn_radius_cell_path_equals_the_wave_path():
    """asr_hip_knn_radius: the cell-parallel fast path (one wave per occupied cell, top-k in registers) plus its
    fallback list gives the radii of the wave-per-point kernel, bit for bit, on a cloud with duplicates, dense and
    sparse regions and isolated points; a slice is checked against the brute-force oracle."""
    from asr_hip import ops, synth
    dev = torch.device("cuda:0")
    p, _ = synth.scan_cloud(300000, seed=12, device="cpu", density_variance=10.0)
    pts = p.numpy().copy()
    rng = np.random.default_rng(3)
    pts[:2000] = pts[2000:4000]                                   # duplicates
    pts[4000:4200] = rng.uniform(-3, 3, size=(200, 3))            # isolated points far from the surface
    pts[10000:40000] = pts[9000] + rng.normal(0, 2e-5, size=(30000, 3)).astype(np.float32)  # a dense spot: one crowded cell
    d = torch.from_numpy(pts).to(dev)
    ctx = ops.context(dev)
    try:
        for k in (1, 8, 24, 32, 40):
            ctx.set_option("knn_cells", 1)
            fast = synth.knn_radii_gpu(d, k)
            ctx.set_option("knn_cells", 0)
            slow = synth.knn_radii_gpu(d, k)
            assert torch.equal(fast, slow), (k, float((fast - slow).abs().max()))
    finally:
        ctx.set_option("knn_cells", 1)
    small = pts[:20000]  # half of it is the dense spot: its points start their search on a finer level ...
    got = synth.knn_radii_gpu(torch.from_numpy(small).to(dev), 24).cpu().numpy()
    assert np.array_equal(got, O.knn_radius(small, 24))
    try:  # ... which must not change a bit
        ctx.set_option("knn_deep", 0)
        flat = synth.knn_radii_gpu(torch.from_numpy(small).to(dev), 24).cpu().numpy()
    finally:
        ctx.set_option("knn_deep", 1)
    assert np.array_equal(got, flat)


@pytest.mark.parametrize("shift", [0.0, 100.0], ids=["frame-at-origin", "frame-100-units-away"])
def test_aligned_search_finds_the_pairs_of_the_rounding_margin(gpu, shift):
    """The aggregation search of the whole path covers the ball of voxel (x, y, z, L) with the 4^3 cells of level
    L + 1 -- a cube without margin.  Points are planted at the poles of the balls (centre +- size along an axis, and
    the neighbouring floats): their cell may be the first layer BEYOND the cube although the float distance test of
    cpp/lib/nsearch.cpp:136-146 accepts them.  The result must equal the oracle's brute force (every point against
    every voxel) bit for bit, and the margin path must have been exercised.  The planted points have a root-level
    radius, so the octree is the one of the base cloud."""
    from asr_hip.pipeline import ImplicitPipeline
    from oracle import oracle as O
    pts, _ = synth.sphere_cloud(6000, seed=5)
    pts = (pts + np.float32(shift)).astype(np.float32)
    rad = synth.knn_radii(pts, 24)
    bb = synth.bounding_box(pts, 0.1)
    o = O.Oracle()
    o.build_octree(pts, rad, *bb)
    g0 = o.create_grids(1)[0]
    c, s = g0["voxel_centers"], g0["voxel_sizes"]
    rng = np.random.default_rng(11)
    pick = rng.choice(len(s), size=min(1500, len(s)), replace=False)
    planted = []
    for q in pick:
        for axis in range(3):
            for sign in (-1.0, 1.0):
                p0 = c[q].copy()
                p0[axis] = np.float32(c[q][axis] + np.float32(sign) * s[q])
                for step in range(-3, 2):  # the pole and the floats around it
                    p = p0.copy()
                    for _ in range(abs(step)):
                        p[axis] = np.nextafter(p[axis], np.float32(np.inf if step > 0 else -np.inf), dtype=np.float32)
                    planted.append(p)
    planted = np.array(planted, np.float32)
    inside = np.all((planted >= bb[0]) & (planted <= bb[1]), axis=1)
    planted = planted[inside]
    allp = np.concatenate([pts, planted]).astype(np.float32)
    edge = float((bb[1] - bb[0]).max())
    allr = np.concatenate([rad, np.full(len(planted), 0.6 * edge, np.float32)])  # level 0: no new octree node
    o2 = O.Oracle()
    o2.build_octree(allp, allr, *bb)
    assert np.array_equal(o2.leaves, o.leaves)
    ridx, rdist, rrs, rcompat = o2.radius_search(allp, allr, c, s, brute=True)
    pipe = ImplicitPipeline(synth.make_weights(4, seed=1), device=gpu)
    # half-size cells for the heavy rows only (the default), for every row, and 3^3 full-size cells throughout
    for half in (1, 2, 0):
        pipe.ctx.set_option("search_half", half)
        pipe.build(torch.from_numpy(allp).to(gpu), torch.from_numpy(allr).to(gpu), bb[0], bb[1])
        assert np.array_equal(pipe.get("voxel_keys0").cpu().numpy().view(np.uint64), g0["voxel_keys"])
        assert np.array_equal(pipe.get("aggregation_row_splits").cpu().numpy(), rrs), half
        assert np.array_equal(pipe.get("aggregation_neighbors_index").cpu().numpy(), ridx), half
        assert np.array_equal(pipe.get("aggregation_neighbors_dist").cpu().numpy(), rdist), half
        margin = pipe.ctx.get_option("last_search_margin_pairs")
        print("shift %g half %d: %d planted points, %d pairs, %d from the rounding margin" %
              (shift, half, len(planted), len(ridx), margin))
        if half:
            assert shift != 0.0 or margin > 0  # (far frame: the float spacing of the planted points is coarser)
    pipe.ctx.set_option("search_half", 1)


def test_six_grid_levels_vs_oracle(gpu):
    """asr::CreateGridsFromOctree(tree, num_levels = 6, ...) (cpp/lib/grid.cpp:245-314; the "6 levels" of BASELINE
    config C3 swept over the connectivity kernels): every array of all six grids equals the oracle's"""
    import adaptivesurfacereconstruction as asr
    from oracle import oracle as O
    p, _ = synth.scan_cloud(60000, seed=77, device="cpu")
    pts = p.numpy()
    rad = synth.knn_radii(pts, 24)
    bb = synth.bounding_box(pts, 0.1)
    o = O.Oracle()
    o.build_octree(pts, rad, *bb)
    want = o.create_grids(6)
    tree = asr.create_octree(pts, rad, bb[0], bb[1])
    got = asr.create_grids_from_octree(tree, 6, voxel_info_all_levels=True)
    assert len(got) == 6
    v = [len(w["voxel_keys"]) for w in want]
    assert v[0] > v[1] > v[2] > v[3] > v[4] >= v[5] >= 1
    for lvl, (g, w) in enumerate(zip(got, want)):
        for k in ("voxel_keys", "voxel_centers", "voxel_sizes", "neighbors_index", "neighbors_kernel_index",
                  "neighbors_row_splits"):
            assert np.array_equal(g[k], w[k]), (lvl, k)
        if lvl < 5:
            for k in ("up_neighbors_index", "up_neighbors_kernel_index", "up_neighbors_row_splits"):
                assert np.array_equal(g[k], w[k]), (lvl, k)
        else:
            assert "up_neighbors_index" not in g  # the coarsest grid has no up lists (quirk B.5)


def test_stage_banners_reach_the_print_callback(gpu):
    """the whole path announces its stages like asr::ReconstructSurface (cpp/lib/asr.cpp:144,264,314-323) through the
    callback registry (asr::SetPrintCallbackFunction, asr.cpp:34-47)"""
    from asr_hip import _lib
    from asr_hip.pipeline import ImplicitPipeline
    p, q = synth.scan_cloud(3000, seed=5, device="cpu")
    rad = torch.from_numpy(synth.knn_radii(p.numpy(), 24))
    bb = synth.bounding_box(p.numpy(), 0.1)
    got = []
    _lib.set_print_callback_function(got.append, ["INFO"])
    try:
        pipe = ImplicitPipeline(synth.make_weights(4, seed=1), device=gpu)
        pipe.forward(p.to(gpu), q.to(gpu), rad.to(gpu), bb[0], bb[1])
    finally:
        _lib.set_print_callback_function(None, ["INFO"])
    assert got == ["grid building\n", "aggregate\n", "network aggregate\n", "network unet\n", "network decode\n"]


@pytest.mark.parametrize("grow_steps", [1, 2, 3])
def test_create_octree_grow_steps_vs_oracle(gpu, grow_steps):
    """create_octree(..., grow_steps, ...) (cpp/pybind/module.cpp:144-161 -> Octree::Grow, cpp/lib/octree.cpp:44-108):
    nodes and leaves equal the oracle's on a one-point tree, a mixed-level scan and a sphere"""
    import adaptivesurfacereconstruction as asr
    from oracle import oracle as O
    clouds = [(np.array([[0.3, 0.3, 0.3]], np.float32), np.array([0.1], np.float32),
               (np.zeros(3, np.float32), np.ones(3, np.float32)))]
    for seed, n in ((3, 5000), (8, 40000)):
        p, _ = synth.scan_cloud(n, seed=seed, device="cpu")
        p = p.numpy()
        clouds.append((p, synth.knn_radii(p, 24), synth.bounding_box(p, 0.1)))
    p, _ = synth.sphere_cloud(20000, seed=2)
    clouds.append((p, synth.knn_radii(p, 24), synth.bounding_box(p, 0.1)))
    for pts, rad, bb in clouds:
        o = O.Oracle()
        o.build_octree(pts, rad, *bb, grow_steps=grow_steps)
        tree = asr.create_octree(pts, rad, bb[0], bb[1], grow_steps=grow_steps)
        assert np.array_equal(tree.nodes.cpu().numpy().view(np.uint64), o.nodes)
        assert np.array_equal(tree.leaves.cpu().numpy().view(np.uint64), o.leaves)
        o0 = O.Oracle()
        o0.build_octree(pts, rad, *bb)
        assert len(o.nodes) >= len(o0.nodes)


def test_neighbor_lists_of_a_row_subset(gpu):
    """asr_hip_grid_neighbors_rows_*: the 55-slot lists of a subset of the voxels equal the corresponding rows of the
    full lists (cpp/lib/grid.cpp:43-175); the other rows are empty"""
    from asr_hip import ops
    p, _ = synth.scan_cloud(30000, seed=12, device="cpu")
    pts = p.numpy()
    rad = synth.knn_radii(pts, 24)
    bb = synth.bounding_box(pts, 0.1)
    frame = _lib.frame_init(bb[0], bb[1])
    nodes, leaves = ops.octree_build(frame, torch.from_numpy(pts).to(gpu), torch.from_numpy(rad).to(gpu))
    idx, kidx, rs = (t.cpu().numpy() for t in ops.grid_neighbors(leaves))
    v = leaves.shape[0]
    rng = np.random.default_rng(3)
    for rows in (np.sort(rng.choice(v, size=v // 3, replace=False)), np.arange(v), np.zeros(0, np.int64), np.array([v - 1])):
        i2, k2, r2 = (t.cpu().numpy() for t in ops.grid_neighbors_rows(leaves, torch.from_numpy(rows.astype(np.int32)).to(gpu)))
        lens = np.zeros(v, np.int64)
        lens[rows] = (rs[1:] - rs[:-1])[rows]
        assert np.array_equal(r2, np.concatenate([[0], np.cumsum(lens)]))
        want_i = np.concatenate([idx[rs[q]:rs[q + 1]] for q in rows]) if len(rows) else np.zeros(0, np.int32)
        want_k = np.concatenate([kidx[rs[q]:rs[q + 1]] for q in rows]) if len(rows) else np.zeros(0, np.uint8)
        assert np.array_equal(i2, want_i) and np.array_equal(k2, want_k)


def test_octree_in_parts_equals_the_whole(gpu):
    """asr_hip_octree_build_parts: three shares of a cloud closed separately (no balancing), their node lists
    concatenated, closed and balanced as a whole -> the nodes and leaves of CreateOctreeFromPoints on the whole cloud
    (cpp/lib/octree.cpp:230-280)"""
    from asr_hip import ops
    p, _ = synth.scan_cloud(40000, seed=31, device="cpu", density_variance=10.0)
    pts = p.numpy()
    rad = synth.knn_radii(pts, 24)
    bb = synth.bounding_box(pts, 0.1)
    frame = _lib.frame_init(bb[0], bb[1])
    dp, dr = torch.from_numpy(pts).to(gpu), torch.from_numpy(rad).to(gpu)
    nodes, leaves = ops.octree_build(frame, dp, dr)
    cuts = [0, 9000, 9001, 40000]
    parts = [ops.octree_build_parts(frame, dp[a:b], dr[a:b], balance=False)[0] for a, b in zip(cuts[:-1], cuts[1:])]
    assert all(part.shape[0] < nodes.shape[0] for part in parts)
    n2, l2 = ops.octree_build_parts(frame, dp[:0], dr[:0], extra_keys=torch.cat(parts), balance=True)
    assert torch.equal(n2, nodes) and torch.equal(l2, leaves)
    o = O.Oracle()
    o.build_octree(pts, rad, *bb)
    assert np.array_equal(n2.cpu().numpy().view(np.uint64), o.nodes)

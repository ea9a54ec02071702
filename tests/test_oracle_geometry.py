"""CPU tests of the oracle's integer / geometry stages.

Pins available for these rows:
  * known-answer counts of the reference on the C1 cloud, recorded in SURVEY.md section 6
    (compiled reference, survey step);
  * development cross-check: sha256 of all grid arrays / dual indices of the same cloud.  The
    digests below equal the digests of the dumps the survey step's reference build left in
    /tmp/oracle_probe (that build used stand-in Eigen/libcuckoo headers, so it is NOT claimed
    as an oracle/_ref pin -- see DESIGN.md "Oracle status");
  * structural invariants of SURVEY A.5b.
"""
import hashlib

import numpy as np
import pytest

from asr_hip import synth
from oracle import oracle as O

GRID_KEYS = ("voxel_keys", "voxel_centers", "voxel_sizes", "neighbors_index",
             "neighbors_kernel_index", "neighbors_row_splits", "up_neighbors_index",
             "up_neighbors_kernel_index", "up_neighbors_row_splits")


@pytest.fixture(scope="module")
def c1():
    pts, _ = synth.sphere_cloud(50000, 0)
    rad = synth.knn_radii(pts, 24)
    bb_min, bb_max = synth.bounding_box(pts, 0.1)
    o = O.Oracle()
    o.build_octree(pts, rad, bb_min, bb_max)
    grids = o.create_grids(5)
    return pts, rad, o, grids


def test_morton_roundtrip_and_dilated_arithmetic():
    rng = np.random.default_rng(0)
    for _ in range(200):
        x, y, z = (int(v) for v in rng.integers(0, 1 << 21, size=3))
        m = O.morton3d(x, y, z)
        assert O.inverse_morton3d(m) == (x, y, z)
        a, b, c = (int(v) for v in rng.integers(0, 1 << 20, size=3))
        assert O.morton_add(m, O.morton3d(1, 0, 0)) == O.morton3d((x + 1) & 0x1FFFFF, y, z)
        big = O.morton3d(x | (1 << 20), y | (1 << 20), z | (1 << 20))
        small = O.morton3d(a, b, c)
        assert O.morton_sub(big, small) == O.morton3d((x | 1 << 20) - a, (y | 1 << 20) - b,
                                                      (z | 1 << 20) - c)
    # known answers: bit interleave x -> bit 0, y -> bit 1, z -> bit 2 (zindex.h:53)
    assert O.morton3d(1, 0, 0) == 1 and O.morton3d(0, 1, 0) == 2 and O.morton3d(0, 0, 1) == 4
    assert O.morton3d(3, 0, 0) == 0b1001 and O.morton3d((1 << 21) - 1, 0, 0) == 0x1249249249249249


def test_location_codes():
    # octreebase.h:59-65: key = morton | 1 << 3*lev ; invalid coordinates -> 0
    assert O.coord_key(0, 0, 0, 0) == 1
    assert O.coord_key(1, 1, 1, 1) == 8 + 7
    assert O.coord_key(2, 0, 0, 1) == 0 and O.coord_key(-1, 0, 0, 3) == 0
    for lev in (0, 1, 5, 21):
        lim = (1 << lev) - 1
        k = O.coord_key(lim, lim // 2, 0, lev)
        assert O.key_coord(k) == (lim, lim // 2, 0, lev)
    assert O.key_coord(O.coord_key(5, 6, 7, 21))[3] == 21


def test_known_answer_counts_c1(c1):
    """SURVEY.md section 6 [probe]: 50 k sphere -> 15 633 nodes / 13 679 leaves,
    V = 13 679 / 4 516 / 1 478 / 407 / 64, pairs 107 171 / 35 264 / 10 882 / 2 663 / 352,
    20 056 dual cells, leaf levels 3..8 at 1 M (here: subset)."""
    pts, rad, o, grids = c1
    assert len(o.nodes) == 15633 and len(o.leaves) == 13679
    assert [len(g["voxel_keys"]) for g in grids] == [13679, 4516, 1478, 407, 64]
    assert [len(g["neighbors_index"]) for g in grids] == [107171, 35264, 10882, 2663, 352]
    assert "up_neighbors_index" not in grids[4]
    assert o.create_dual_vertex_indices().shape == (20056, 8)


def test_development_crosscheck_digests_c1(c1):
    pts, rad, o, grids = c1
    h = hashlib.sha256()
    for g in grids:
        for k in GRID_KEYS:
            if k in g:
                h.update(np.ascontiguousarray(g[k]).tobytes())
    assert h.hexdigest() == "94c509ff76cee6345bf8262e5adfb6ed6348047302cb0b597b172d877ffadfc0"
    du = o.create_dual_vertex_indices().astype(np.uint64)
    assert hashlib.sha256(du.tobytes()).hexdigest() == \
        "397a829727f3796bcdee3669989ea4ebf5d628cbbe6e9087b7df6237311375ee"


def test_csr_invariants(c1):
    """SURVEY A.5b"""
    pts, rad, o, grids = c1
    for gi, g in enumerate(grids):
        keys = g["voxel_keys"]
        assert np.all(keys[1:] > keys[:-1])
        rs, idx, kidx = g["neighbors_row_splits"], g["neighbors_index"], g["neighbors_kernel_index"]
        assert rs[0] == 0 and rs[-1] == len(idx) and np.all(np.diff(rs) >= 1)
        assert np.array_equal(idx[rs[:-1]], np.arange(len(keys))) and np.all(kidx[rs[:-1]] == 0)
        row = np.repeat(np.arange(len(keys)), np.diff(rs))
        inner = np.ones(len(idx), bool)
        inner[rs[:-1]] = False
        assert np.all(np.diff(kidx.astype(int))[inner[1:]] > 0)  # strictly ascending slots
        assert kidx.max() < 55
        pairs = set(zip(row.tolist(), idx.tolist()))
        assert all((b, a) in pairs for a, b in pairs)  # symmetric relation
        if gi < 4:
            up_idx, up_k = g["up_neighbors_index"], g["up_neighbors_kernel_index"]
            assert np.array_equal(g["up_neighbors_row_splits"], np.arange(len(keys) + 1))
            nxt = grids[gi + 1]["voxel_keys"]
            merged = up_k < 8
            assert np.array_equal(nxt[up_idx[merged]], keys[merged] >> np.uint64(3))
            assert np.array_equal(up_k[merged], (keys[merged] & np.uint64(7)).astype(np.uint8))
            assert np.array_equal(nxt[up_idx[~merged]], keys[~merged])


def test_octree_closure_properties(c1):
    pts, rad, o, grids = c1
    nodes = set(o.nodes.tolist())
    assert 1 in nodes
    for k in o.nodes.tolist():
        if k != 1:
            assert (k >> 3) in nodes                      # ancestors
            assert all(((k & ~7) + j) in nodes for j in range(8))  # siblings
    leaves = set(o.leaves.tolist())
    assert leaves == {k for k in nodes if (k << 3) not in nodes}


def test_balance_sequential_equals_round_synchronous():
    """mode 1 restates the reference's sequential queue walk (octree.cpp:168-205) in ascending
    key order; on these clouds it must agree with the canonical round-synchronous statement"""
    for seed, n in ((0, 20000), (1, 5000)):
        pts, nrm = synth.scan_cloud(n, seed=seed, device="cpu")
        pts = pts.numpy()
        rad = synth.knn_radii(pts, 24)
        bb = synth.bounding_box(pts, 0.1)
        a, b = O.Oracle(), O.Oracle()
        la = a.build_octree(pts, rad, *bb, mode=0)
        lb = b.build_octree(pts, rad, *bb, mode=1)
        assert np.array_equal(la, lb) and np.array_equal(a.nodes, b.nodes)


def test_point_keys_edge_cases():
    o = O.Oracle()
    pts = np.array([[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [2, 0, 0], [1, 0.25, 0.75]], np.float32)
    rad = np.array([0.1, 0.1, 10.0, 0.1, 1e-9], np.float32)
    o.build_octree(pts, rad, np.zeros(3, np.float32), np.ones(3, np.float32))
    keys = o.point_keys(pts, rad)
    assert keys[3] == 0            # outside the box: skipped (octree.cpp:248-251)
    assert keys[1] == 0            # on the max face -> coordinate 2^lev -> INVALID_KEY (B.1)
    assert keys[2] == 1            # radius larger than the root cube -> level 0
    assert keys[4] == 0            # tiny radius -> level 21, x on the max face -> invalid
    assert O.key_coord(int(keys[0]))[:3] == (0, 0, 0) and keys[0] != 0
    # max_depth clamp (octree.cpp:253-254)
    k5 = o.point_keys(pts, rad, max_depth=5)
    assert O.key_coord(int(k5[0]))[3] <= 5


def test_radius_search_cells_equal_brute_force():
    pts, nrm = synth.scan_cloud(3000, seed=5, device="cpu")
    pts = pts.numpy()
    rad = synth.knn_radii(pts, 24)
    bb = synth.bounding_box(pts, 0.1)
    o = O.Oracle()
    o.build_octree(pts, rad, *bb)
    g = o.create_grids(1)[0]
    a = o.radius_search(pts, rad, g["voxel_centers"], g["voxel_sizes"], brute=True)
    b = o.radius_search(pts, rad, g["voxel_centers"], g["voxel_sizes"], brute=False)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    idx, dist, rs, compat = a
    # members: strict squared distance test; rows sorted by (distance, index)
    q = np.repeat(np.arange(len(rs) - 1), np.diff(rs))
    d = ((pts[idx] - g["voxel_centers"][q]) ** 2).sum(1)
    assert np.all(d < g["voxel_sizes"][q] ** 2 * (1 + 1e-6))
    for r in range(0, len(rs) - 1, 37):
        seg = dist[rs[r]:rs[r + 1]]
        assert np.all(np.diff(seg) >= 0)
    assert len(idx) >= len(g["voxel_sizes"])  # SURVEY B.2 precondition on such clouds


def test_invert_neighbors_list_roundtrip(c1):
    pts, rad, o, grids = c1
    g = grids[0]
    n_coarse = len(grids[1]["voxel_keys"])
    idx, rs, attr = O.invert_neighbors_list(n_coarse, g["up_neighbors_index"],
                                            g["up_neighbors_row_splits"],
                                            g["up_neighbors_kernel_index"])
    assert rs[-1] == len(idx) and set(np.diff(rs).tolist()) <= {1, 8}
    # rows hold the fine voxels in ascending order, attributes travel with them
    for r in range(0, n_coarse, 97):
        seg = idx[rs[r]:rs[r + 1]]
        assert np.all(np.diff(seg) > 0)
        assert np.all(g["up_neighbors_index"][seg] == r)
        assert np.array_equal(attr[rs[r]:rs[r + 1]], g["up_neighbors_kernel_index"][seg])
    # inverting twice returns the original list
    idx2, rs2, attr2 = O.invert_neighbors_list(len(g["voxel_keys"]), idx, rs, attr)
    assert np.array_equal(idx2, g["up_neighbors_index"]) and np.array_equal(attr2, g["up_neighbors_kernel_index"])


def test_grow_hand_worked_and_both_walks_agree():
    """Octree::Grow (/root/reference/cpp/lib/octree.cpp:44-108) in the oracle.  Hand-worked: one point
    (0.3, 0.3, 0.3), r 0.2 in [0,1]^3 -> level 2 (0.25 >= 0.2), cell (1,1,1), key 7 | 64 = 71.  One Grow
    iteration on the key set {71}: parent 8 = level-1 cell (0,0,0); configuration 7 - (71 & 7) = 0, offset
    (0,0,0), so the seven other cells of the block are parent + (1,0,0) .. (1,1,1) = keys 9..15 (none a node,
    none with a child): inserted (:73-91); siblings 64..70 inserted (:94-101).  A second iteration: every key 9..15
    (parent: the root, level 0) looks at level-0 cells next to the root: all outside the cube (INVALID_KEY);
    keys 64..70 find 9..15 present.  Closure afterwards adds 8 and the root: 17 nodes either way."""
    from oracle import oracle as O
    pts, rad = np.array([[0.3, 0.3, 0.3]], np.float32), np.array([0.2], np.float32)
    lo, hi = np.zeros(3, np.float32), np.ones(3, np.float32)
    for steps in (1, 2):
        for mode in (0, 1):
            o = O.Oracle()
            o.build_octree(pts, rad, lo, hi, mode=mode, grow_steps=steps)
            assert o.nodes.tolist() == [1] + list(range(8, 16)) + list(range(64, 72))
    # a point whose block reaches into neighbouring parents: (0.3,0.3,0.3) at level 3 -> cell (2,2,2), key
    # Morton(2,2,2) = 56 | 512 = 568, child 0 of parent 71 = level-2 cell (1,1,1): configuration 7, offset (1,1,1):
    # the block is parent + {-1,0}^3 without (0,0,0) = level-2 cells (0..1)^3 minus (1,1,1) = keys 64..70 (siblings of
    # 71: the closure would add them anyway).  BalanceFaces then works as without Grow: the leaf group 568.. wants the
    # face neighbours of its parent 71 = (1,1,1): (2,1,1) = 14|64 = 78, (1,2,1) = 21|64 = 85, (1,1,2) = 35|64 = 99 are
    # missing -> their sibling groups 72..79, 80..87, 96..103 are inserted (octree.cpp:191-201)
    o = O.Oracle()
    o.build_octree(pts, np.array([0.1], np.float32), lo, hi, grow_steps=1)
    assert o.nodes.tolist() == ([1] + list(range(8, 16)) + list(range(64, 88)) + list(range(96, 104)) +
                                list(range(568, 576)))
    # mixed-level scan clouds: the round-synchronous statement and the sequential walk give the same tree
    from asr_hip import synth
    for seed, n in ((3, 5000), (8, 20000)):
        p, _ = synth.scan_cloud(n, seed=seed, device="cpu")
        p = p.numpy()
        r = synth.knn_radii(p, 24)
        bb = synth.bounding_box(p, 0.1)
        sizes = []
        for steps in (0, 1, 2, 3):
            res = []
            for mode in (0, 1):
                o = O.Oracle()
                o.build_octree(p, r, *bb, mode=mode, grow_steps=steps)
                res.append(o.nodes)
            assert np.array_equal(res[0], res[1]), (seed, steps)
            sizes.append(len(res[0]))
        assert sizes[1] > sizes[0] and sizes[2] >= sizes[1]

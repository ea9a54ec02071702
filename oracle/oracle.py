"""ctypes wrapper around oracle/libasr_oracle.so (CPU restatement of the hot path).

TEST INFRASTRUCTURE ONLY -- see the header of oracle/asr_oracle.cpp.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile the oracle with g++ (recipe: oracle/Makefile)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libasr_oracle.so")
        src = os.path.join(_HERE, "asr_oracle.cpp")
        if not os.path.exists(path) or os.path.getmtime(src) > os.path.getmtime(path) + 1:
            build()  # never run with a library older than its source
        _LIB = ctypes.CDLL(path)
        _LIB.orc_create.restype = ctypes.c_void_p
        for name in ("orc_octree_build", "orc_octree_build_grow", "orc_num_nodes", "orc_leaf_neighbors", "orc_create_duals",
                     "orc_radius_search"):
            getattr(_LIB, name).restype = ctypes.c_int64
        for name in ("orc_morton3d", "orc_morton_add", "orc_morton_sub", "orc_coord_key"):
            getattr(_LIB, name).restype = ctypes.c_uint64
    return _LIB


def _p(a):
    if a is None:
        return ctypes.c_void_p(0)
    assert a.flags["C_CONTIGUOUS"]
    return ctypes.c_void_p(a.ctypes.data)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


u64 = ctypes.c_uint64
i64 = ctypes.c_int64


# ---- a1 / a2 -------------------------------------------------------------------------
def morton3d(x, y, z):
    return int(lib().orc_morton3d(u64(x), u64(y), u64(z)))


def inverse_morton3d(m):
    out = np.zeros(3, np.uint64)
    lib().orc_inverse_morton3d(u64(m), _p(out))
    return tuple(int(v) for v in out)


def morton_add(a, b):
    return int(lib().orc_morton_add(u64(a), u64(b)))


def morton_sub(a, b):
    return int(lib().orc_morton_sub(u64(a), u64(b)))


def coord_key(x, y, z, lev):
    return int(lib().orc_coord_key(int(x), int(y), int(z), int(lev)))


def key_coord(key):
    out = np.zeros(4, np.int32)
    lib().orc_key_coord(u64(key), _p(out))
    return tuple(int(v) for v in out)


class Oracle:
    """Owns one octree + grids + aggregation result."""

    def __init__(self):
        self._h = ctypes.c_void_p(lib().orc_create())

    def __del__(self):
        try:
            lib().orc_destroy(self._h)
        except Exception:
            pass

    # a3/a4 (cpp/lib/octree.cpp:230-280)
    def build_octree(self, points, radii, bb_min, bb_max, radius_scale=1.0, max_depth=21, mode=0, grow_steps=0):
        points = _f32(points)
        radii = _f32(radii)
        self.bb_min = _f32(bb_min)
        self.bb_max = _f32(bb_max)
        n = points.shape[0]
        nl = lib().orc_octree_build_grow(self._h, _p(points), i64(n), _p(radii), _p(self.bb_min),
                                         _p(self.bb_max), ctypes.c_float(radius_scale), int(max_depth),
                                         int(mode), int(grow_steps))
        self.leaves = np.zeros(nl, np.uint64)
        lib().orc_get_leaves(self._h, _p(self.leaves))
        nn = lib().orc_num_nodes(self._h)
        self.nodes = np.zeros(nn, np.uint64)
        lib().orc_get_nodes(self._h, _p(self.nodes))
        self.balance_rounds = lib().orc_balance_rounds(self._h)
        return self.leaves

    def frame(self):
        vs = np.zeros(22, np.float32)
        ivs = np.zeros(22, np.float32)
        off = np.zeros(3, np.int32)
        lib().orc_frame(self._h, _p(vs), _p(ivs), _p(off))
        return vs, ivs, off

    def point_keys(self, points, radii, radius_scale=1.0, max_depth=21):
        points = _f32(points)
        radii = _f32(radii)
        keys = np.zeros(points.shape[0], np.uint64)
        lib().orc_point_keys(self._h, _p(points), i64(points.shape[0]), _p(radii),
                             ctypes.c_float(radius_scale), int(max_depth), _p(keys))
        return keys

    # a5-a7 (cpp/lib/grid.cpp:245-314); returns list of dicts with the pybind key names
    def create_grids(self, num_levels=5):
        lib().orc_create_grids(self._h, int(num_levels))
        grids = []
        for lev in range(num_levels):
            sz = np.zeros(3, np.int64)
            lib().orc_grid_sizes(self._h, lev, _p(sz))
            V, P, U = (int(v) for v in sz)
            g = {
                "voxel_keys": np.zeros(V, np.uint64),
                "voxel_centers": np.zeros((V, 3), np.float32),
                "voxel_sizes": np.zeros(V, np.float32),
                "neighbors_index": np.zeros(P, np.int32),
                "neighbors_kernel_index": np.zeros(P, np.uint8),
                "neighbors_row_splits": np.zeros(V + 1, np.int64),
            }
            up = None
            if U:
                up = (np.zeros(U, np.int32), np.zeros(U, np.uint8), np.zeros(V + 1, np.int64))
                g["up_neighbors_index"], g["up_neighbors_kernel_index"], g["up_neighbors_row_splits"] = up
            lib().orc_grid_get(self._h, lev, _p(g["voxel_keys"]), _p(g["voxel_centers"]),
                               _p(g["voxel_sizes"]), _p(g["neighbors_index"]),
                               _p(g["neighbors_kernel_index"]), _p(g["neighbors_row_splits"]),
                               _p(up[0]) if up else _p(None), _p(up[1]) if up else _p(None),
                               _p(up[2]) if up else _p(None))
            grids.append(g)
        self.grids = grids
        return grids

    def leaf_neighbors(self, keys):
        keys = np.ascontiguousarray(keys, np.uint64)
        P = lib().orc_leaf_neighbors(self._h, _p(keys), i64(len(keys)))
        idx = np.zeros(P, np.int32)
        kidx = np.zeros(P, np.uint8)
        rs = np.zeros(len(keys) + 1, np.int64)
        lib().orc_grid_get(self._h, 0, _p(None), _p(None), _p(None), _p(idx), _p(kidx), _p(rs),
                           _p(None), _p(None), _p(None))
        return idx, kidx, rs

    def create_dual_vertex_indices(self):
        d = lib().orc_create_duals(self._h)
        out = np.zeros((d, 8), np.int64)
        lib().orc_get_duals(self._h, _p(out))
        return out

    # a8 (cpp/lib/nsearch.cpp:107-162)
    def radius_search(self, points, radii, centers, sizes, brute=False):
        points = _f32(points)
        radii = _f32(radii)
        centers = _f32(centers)
        sizes = _f32(sizes)
        P = lib().orc_radius_search(self._h, _p(points), i64(points.shape[0]), _p(radii),
                                    _p(centers), _p(sizes), i64(sizes.shape[0]), int(brute))
        idx = np.zeros(P, np.int32)
        dist = np.zeros(P, np.float32)
        rs = np.zeros(sizes.shape[0] + 1, np.int64)
        compat = np.zeros(P, np.float32)
        lib().orc_get_agg(self._h, _p(idx), _p(dist), _p(rs), _p(compat))
        return idx, dist, rs, compat


class dense:
    """context manager: sparse_conv evaluated the way Open3D's CPU op does (a dense [32][K*cin] matrix per block
    of 32 voxels times the filter matrix -- 55*cin deep for every voxel although ~8 slots are occupied).
    The cpu_baseline of bench.py runs under it: that is what the reference's CPU path costs."""

    def __enter__(self):
        self._old = lib().orc_get_dense()
        lib().orc_set_dense(1)
        return self

    def __exit__(self, *exc):
        lib().orc_set_dense(self._old)
        return False


class precise:
    """context manager: the floating point ops (continuous_conv, sparse_conv, decode) accumulate in
    double and round once -- the value every fp32 summation order approximates.  Checker for the
    large-cloud parity tests, where the max-norm over 10^6 outputs would otherwise also measure the
    rounding of the ORACLE's own fp32 pair-order sums."""

    def __enter__(self):
        self._old = lib().orc_get_precise()
        lib().orc_set_precise(1)
        return self

    def __exit__(self, *exc):
        lib().orc_set_precise(self._old)
        return False


# ---- Open3D ops ------------------------------------------------------
def continuous_conv(filters, out_positions, extents, inp_positions, inp_features, neighbors_index,
                    neighbors_importance, neighbors_row_splits, normalize=True):
    filters = _f32(filters)
    cin, cout = filters.shape[-2], filters.shape[-1]
    assert filters.shape[:3] == (4, 4, 4)
    out_positions = _f32(out_positions)
    v = out_positions.shape[0]
    extents = _f32(np.broadcast_to(np.asarray(extents, np.float32).reshape(-1), (v,)))
    inp_positions = _f32(inp_positions)
    inp_features = _f32(inp_features)
    nidx = np.ascontiguousarray(neighbors_index, np.int32)
    rs = np.ascontiguousarray(neighbors_row_splits, np.int64)
    nimp = None
    if neighbors_importance is not None and len(neighbors_importance):
        nimp = _f32(neighbors_importance)
    out = np.zeros((v, cout), np.float32)
    lib().orc_continuous_conv(_p(filters), _p(out_positions), _p(extents), _p(inp_positions),
                              _p(inp_features), _p(nidx), _p(nimp), _p(rs), i64(v), int(cin),
                              int(cout), int(bool(normalize)), _p(out))
    return out


def sparse_conv(filters, inp_features, neighbors_index, neighbors_kernel_index,
                neighbors_importance, neighbors_row_splits, normalize=False):
    filters = _f32(filters)
    _, cin, cout = filters.shape
    inp_features = _f32(inp_features)
    assert inp_features.shape[1] == cin
    nidx = np.ascontiguousarray(neighbors_index, np.int32)
    nk = np.ascontiguousarray(neighbors_kernel_index, np.uint8)
    rs = np.ascontiguousarray(neighbors_row_splits, np.int64)
    v = rs.shape[0] - 1
    nimp = None
    if neighbors_importance is not None and len(neighbors_importance):
        nimp = _f32(neighbors_importance)
    out = np.zeros((v, cout), np.float32)
    lib().orc_sparse_conv_k(_p(filters), int(filters.shape[0]), _p(inp_features), i64(cin), _p(nidx), _p(nk),
                            _p(nimp), _p(rs), i64(v), int(cin), int(cout), int(bool(normalize)), _p(out), i64(cout))
    return out


def knn_radius(points, k):
    """KDTree::ComputeKRadius (cpp/lib/nsearch.cpp:30-51), brute force"""
    points = _f32(points)
    out = np.zeros(points.shape[0], np.float32)
    lib().orc_knn_radius(_p(points), i64(points.shape[0]), int(k), _p(out))
    return out


def radius_count(points, radii):
    """KDTree::ComputeRadiusNeighbors (cpp/lib/nsearch.cpp:88-105), brute force"""
    points = _f32(points)
    radii = _f32(radii)
    out = np.zeros(points.shape[0], np.int32)
    lib().orc_radius_count(_p(points), i64(points.shape[0]), _p(radii), _p(out))
    return out


def scale_compat(voxel_sizes, radii, neighbors_index, row_splits):
    sizes = _f32(voxel_sizes)
    radii = _f32(radii)
    idx = np.ascontiguousarray(neighbors_index, np.int32)
    rs = np.ascontiguousarray(row_splits, np.int64)
    out = np.zeros(idx.shape[0], np.float32)
    lib().orc_scale_compat(_p(sizes), _p(radii), _p(idx), _p(rs), i64(rs.shape[0] - 1), _p(out))
    return out


def reduce_subarrays_sum(values, row_splits):
    values = _f32(values)
    rs = np.ascontiguousarray(row_splits, np.int64)
    out = np.zeros(rs.shape[0] - 1, np.float32)
    lib().orc_reduce_subarrays_sum(_p(values), _p(rs), i64(rs.shape[0] - 1), _p(out))
    return out


def invert_neighbors_list(num_points, inp_neighbors_index, inp_neighbors_row_splits,
                          inp_neighbors_attributes):
    idx = np.ascontiguousarray(inp_neighbors_index, np.int32)
    rs = np.ascontiguousarray(inp_neighbors_row_splits, np.int64)
    attr = None
    if inp_neighbors_attributes is not None and len(inp_neighbors_attributes):
        attr = np.ascontiguousarray(inp_neighbors_attributes, np.uint8)
    out_idx = np.zeros(idx.shape[0], np.int32)
    out_rs = np.zeros(num_points + 1, np.int64)
    out_attr = np.zeros(idx.shape[0] if attr is not None else 0, np.uint8)
    lib().orc_invert_neighbors_list(i64(num_points), _p(idx), _p(rs), i64(rs.shape[0] - 1),
                                    _p(attr), _p(out_idx), _p(out_rs), _p(out_attr))
    return out_idx, out_rs, out_attr


def decode(code, w1, b1, w2, b2, w3, voxel_sizes=None):
    code = _f32(code)
    v, c = code.shape
    w1, b1, w2, b2, w3 = (_f32(a) for a in (w1, b1, w2, b2, w3))
    sizes = _f32(voxel_sizes) if voxel_sizes is not None else None
    out = np.zeros((v, 2), np.float32)
    lib().orc_decode(_p(code), i64(v), int(c), _p(w1), _p(b1), int(w1.shape[0]), _p(w2), _p(b2),
                     int(w2.shape[0]), _p(w3), _p(sizes), _p(out))
    return out


def window_poly6(r_sqr):
    """models/common_torch.py:21-22"""
    r_sqr = np.asarray(r_sqr, np.float32)
    one = np.float32(1)
    t = one - r_sqr
    return np.clip(t * t * t, np.float32(0), one).astype(np.float32)


# ---- "next" rows D.2 / D.3: dual contouring and component filter --------------------------------
def create_triangle_mesh(values, dual_vertex_indices, node_positions, threshold=1.0):
    """asr::CreateTriangleMesh (cpp/lib/contouring.cpp:29-460) -> (vertices f32[M,3], triangles i32[T,3])"""
    values = _f32(values)
    duals = np.ascontiguousarray(dual_vertex_indices, np.int64)
    pos = _f32(node_positions)
    sz = np.zeros(2, np.int64)
    err = lib().orc_create_triangle_mesh(_p(values), i64(values.shape[0]), _p(duals), i64(duals.shape[0]),
                                         _p(pos), ctypes.c_float(threshold), _p(sz))
    if err:
        raise RuntimeError("this should not happen: cannot sort duals")
    v = np.zeros((int(sz[0]), 3), np.float32)
    t = np.zeros((int(sz[1]), 3), np.int32)
    lib().orc_get_mesh(_p(v), _p(t))
    return v, t


def remove_connected_components(vertices, triangles, keep_n_largest_components, minimum_component_size=3):
    """asr::RemoveConnectedComponents (cpp/lib/postprocess.cpp:141-176)"""
    v = _f32(vertices)
    t = np.ascontiguousarray(triangles, np.int32)
    lib().orc_set_mesh(_p(v), i64(v.shape[0]), _p(t), i64(t.shape[0]))
    sz = np.zeros(2, np.int64)
    lib().orc_remove_connected_components(i64(min(int(keep_n_largest_components), 2**62)),
                                          i64(int(minimum_component_size)), _p(sz))
    v2 = np.zeros((int(sz[0]), 3), np.float32)
    t2 = np.zeros((int(sz[1]), 3), np.int32)
    lib().orc_get_mesh(_p(v2), _p(t2))
    return v2, t2


def unordered_set_order(xs):
    xs = np.ascontiguousarray(xs, np.uint64)
    out = np.zeros(len(xs), np.uint64)
    lib().orc_unordered_set_order(_p(xs), int(len(xs)), _p(out))
    return out

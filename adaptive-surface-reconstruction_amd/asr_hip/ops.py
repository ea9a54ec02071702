"""Per-stage operators on GPU torch tensors, thin wrappers over the C ABI (include/asr_hip.h).

Names and argument meaning follow the reference interfaces they stand in for:
  create_octree / create_grids_from_octree  -> cpp/pybind/module.cpp:144-228
  multi_radius_search                        -> cpp/lib/nsearch.cpp:107-162
  continuous_conv / sparse_conv / invert_neighbors_list / reduce_subarrays_sum
                                             -> open3d.ml.torch.ops used by models/common_torch.py
"""
import ctypes

import torch

from . import _lib
from ._lib import AsrHipError, Context, ptr

_ctx = {}  # device index -> Context


def context(device=None):
    """the process-wide context of `device` (default: torch's current device), on torch's current
    stream of that device.  One context per GPU: a context is bound to the device it was created on."""
    if device is None:
        index = torch.cuda.current_device()
    else:
        device = torch.device(device)
        index = device.index if device.index is not None else torch.cuda.current_device()
    ctx = _ctx.get(index)
    if ctx is None:
        with torch.cuda.device(index):
            ctx = _ctx[index] = Context()
    ctx.set_stream(torch.cuda.current_stream(index))
    return ctx


def _dev(t, dtype):
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(t)
    if not t.is_cuda:
        raise AsrHipError("expected a GPU tensor: the HIP path has no CPU fallback")
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def _same_device(*tensors):
    """all tensors of one call must live on one GPU; returns that device"""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise AsrHipError("tensors on different devices (%s and %s) in one call" % (dev, t.device))
    return dev


i64 = ctypes.c_int64


def point_keys(frame, points, radii, radius_scale=1.0, max_depth=21):
    points = _dev(points, torch.float32)
    radii = _dev(radii, torch.float32)
    n = points.shape[0]
    keys = torch.empty(n, dtype=torch.int64, device=points.device)
    context(_same_device(points, radii)).call("asr_hip_point_keys", ctypes.byref(frame), ptr(points), ptr(radii), i64(n),
                   ctypes.c_float(radius_scale), int(max_depth), ptr(keys))
    return keys  # uint64 bit pattern in an int64 tensor


def octree_build(frame, points, radii, radius_scale=1.0, max_depth=21, grow_steps=0):
    """-> (nodes, leaves) sorted uint64 keys (as int64 tensors)"""
    points = _dev(points, torch.float32)
    radii = _dev(radii, torch.float32)
    if points.ndim != 2 or points.shape[1] != 3:
        raise ValueError("points must have shape [N,3]")
    if radii.ndim != 1 or radii.shape[0] != points.shape[0]:
        raise ValueError("radii must have shape [N]")
    nn, nl = i64(0), i64(0)
    ctx = context(_same_device(points, radii))
    ctx.call("asr_hip_octree_build_grow", ctypes.byref(frame), ptr(points), ptr(radii),
             i64(points.shape[0]), ctypes.c_float(radius_scale), int(grow_steps), int(max_depth), ctypes.byref(nn),
             ctypes.byref(nl))
    nodes = torch.empty(nn.value, dtype=torch.int64, device=points.device)
    leaves = torch.empty(nl.value, dtype=torch.int64, device=points.device)
    ctx.call("asr_hip_octree_get", ptr(nodes), ptr(leaves))
    return nodes, leaves


def octree_build_parts(frame, points, radii, radius_scale=1.0, max_depth=21, extra_keys=None, balance=True):
    """asr_hip_octree_build_parts: closure of the keys of `points` (may be empty) and of `extra_keys` (int64 tensor of
    node keys), balanced unless balance is False -> (nodes, leaves)"""
    points = _dev(points, torch.float32).reshape(-1, 3)
    radii = _dev(radii, torch.float32).reshape(-1)
    ek = _dev(extra_keys, torch.int64) if extra_keys is not None and extra_keys.numel() else None
    nn, nl = i64(0), i64(0)
    ctx = context(_same_device(points, radii, ek))
    ctx.call("asr_hip_octree_build_parts", ctypes.byref(frame), ptr(points) if points.shape[0] else ctypes.c_void_p(0),
             ptr(radii) if points.shape[0] else ctypes.c_void_p(0), i64(points.shape[0]), ctypes.c_float(radius_scale),
             int(max_depth), ptr(ek), i64(ek.shape[0] if ek is not None else 0), int(bool(balance)), ctypes.byref(nn),
             ctypes.byref(nl))
    nodes = torch.empty(nn.value, dtype=torch.int64, device=points.device)
    leaves = torch.empty(nl.value, dtype=torch.int64, device=points.device)
    ctx.call("asr_hip_octree_get", ptr(nodes), ptr(leaves))
    return nodes, leaves


def dual_cells(device, ctx=None, nodes=None, leaves=None):
    """dual_vertex_indices [D,8] (int64): of the octree built last on this context, or of the octree given by
    its sorted node / leaf key tensors (any tree that is still alive)"""
    ctx = ctx or context(device)
    d = i64(0)
    if nodes is not None:
        nodes, leaves = _dev(nodes, torch.int64), _dev(leaves, torch.int64)
        ctx.call("asr_hip_dual_cells_count_for", ptr(nodes), i64(nodes.shape[0]), ptr(leaves), i64(leaves.shape[0]),
                 ctypes.byref(d))
    else:
        ctx.call("asr_hip_dual_cells_count", ctypes.byref(d))
    out = torch.empty((d.value, 8), dtype=torch.int64, device=device)
    if d.value:
        ctx.call("asr_hip_dual_cells_fill", ptr(out))
    return out


def contour(values, dual_vertex_indices, node_positions, threshold=1.0, ctx=None):
    """asr::CreateTriangleMesh (cpp/lib/contouring.cpp:29-460) -> (vertices f32[M,3], triangles i32[T,3])"""
    values = _dev(values, torch.float32)
    duals = _dev(dual_vertex_indices, torch.int64)
    pos = _dev(node_positions, torch.float32)
    if values.dim() != 2 or values.shape[1] != 2:
        raise ValueError("values must have shape [V,2]")
    if duals.dim() != 2 or duals.shape[1] != 8:
        raise ValueError("dual_vertex_indices must have shape [D,8]")
    if pos.dim() != 2 or pos.shape[1] != 3 or pos.shape[0] != values.shape[0]:
        raise ValueError("node_positions must have shape [V,3]")
    ctx = ctx or context(_same_device(values, duals, pos))
    nv, nt = i64(0), i64(0)
    ctx.call("asr_hip_contour_count", ptr(values), i64(values.shape[0]), ptr(duals), i64(duals.shape[0]),
             ptr(pos), ctypes.c_float(threshold), ctypes.byref(nv), ctypes.byref(nt))
    vertices = torch.empty((nv.value, 3), dtype=torch.float32, device=values.device)
    triangles = torch.empty((nt.value, 3), dtype=torch.int32, device=values.device)
    ctx.call("asr_hip_contour_fill", ptr(vertices), ptr(triangles))
    return vertices, triangles


def remove_components(vertices, triangles, keep_n, min_size=3, ctx=None):
    """asr::RemoveConnectedComponents (cpp/lib/postprocess.cpp:141-176)"""
    vertices = _dev(vertices, torch.float32)
    triangles = _dev(triangles, torch.int32)
    if vertices.dim() != 2 or vertices.shape[1] != 3:
        raise ValueError("vertices must have shape [N,3]")
    if triangles.dim() != 2 or triangles.shape[1] != 3:
        raise ValueError("triangles must have shape [M,3]")
    ctx = ctx or context(_same_device(vertices, triangles))
    nv, nt = i64(0), i64(0)
    ctx.call("asr_hip_components_count", ptr(vertices), i64(vertices.shape[0]), ptr(triangles),
             i64(triangles.shape[0]), i64(min(int(keep_n), 2**62)), i64(int(min_size)), ctypes.byref(nv),
             ctypes.byref(nt))
    v2 = torch.empty((nv.value, 3), dtype=torch.float32, device=vertices.device)
    t2 = torch.empty((nt.value, 3), dtype=torch.int32, device=vertices.device)
    ctx.call("asr_hip_components_fill", ptr(v2), ptr(t2))
    return v2, t2


def grid_neighbors(keys):
    keys = _dev(keys, torch.int64)
    v = keys.shape[0]
    rs = torch.empty(v + 1, dtype=torch.int64, device=keys.device)
    p = i64(0)
    ctx = context(_same_device(keys))
    ctx.call("asr_hip_grid_neighbors_count", ptr(keys), i64(v), ptr(rs), ctypes.byref(p))
    idx = torch.empty(p.value, dtype=torch.int32, device=keys.device)
    kidx = torch.empty(p.value, dtype=torch.uint8, device=keys.device)
    ctx.call("asr_hip_grid_neighbors_fill", ptr(keys), i64(v), ptr(rs), ptr(idx), ptr(kidx))
    return idx, kidx, rs


def grid_neighbors_rows(keys, rows):
    """55-slot lists of the voxels `rows` (ascending int32 indices) only: -> (index, kernel_index, row_splits [V + 1]);
    the rows that are not listed are empty, the entries of the listed ones compact and in row order"""
    keys = _dev(keys, torch.int64)
    rows = _dev(rows, torch.int32)
    v = keys.shape[0]
    rs = torch.empty(v + 1, dtype=torch.int64, device=keys.device)
    p = i64(0)
    ctx = context(_same_device(keys, rows))
    ctx.call("asr_hip_grid_neighbors_rows_count", ptr(keys), i64(v), ptr(rows), i64(rows.shape[0]), ptr(rs), ctypes.byref(p))
    idx = torch.empty(p.value, dtype=torch.int32, device=keys.device)
    kidx = torch.empty(p.value, dtype=torch.uint8, device=keys.device)
    ctx.call("asr_hip_grid_neighbors_rows_fill", ptr(keys), i64(v), ptr(rows), i64(rows.shape[0]), ptr(rs), ptr(idx),
             ptr(kidx))
    return idx, kidx, rs


def grid_coarsen(keys):
    keys = _dev(keys, torch.int64)
    v = keys.shape[0]
    vo = i64(0)
    ctx = context(_same_device(keys))
    ctx.call("asr_hip_grid_coarsen_count", ptr(keys), i64(v), ctypes.byref(vo))
    out_keys = torch.empty(vo.value, dtype=torch.int64, device=keys.device)
    up_idx = torch.empty(v, dtype=torch.int32, device=keys.device)
    up_kidx = torch.empty(v, dtype=torch.uint8, device=keys.device)
    up_rs = torch.empty(v + 1, dtype=torch.int64, device=keys.device)
    ctx.call("asr_hip_grid_coarsen_fill", ptr(keys), i64(v), ptr(out_keys), vo, ptr(up_idx),
             ptr(up_kidx), ptr(up_rs))
    return out_keys, up_idx, up_kidx, up_rs


def voxel_info(frame, keys):
    keys = _dev(keys, torch.int64)
    v = keys.shape[0]
    centers = torch.empty((v, 3), dtype=torch.float32, device=keys.device)
    sizes = torch.empty(v, dtype=torch.float32, device=keys.device)
    context(_same_device(keys)).call("asr_hip_voxel_info", ctypes.byref(frame), ptr(keys), i64(v), ptr(centers),
                   ptr(sizes))
    return centers, sizes


def multi_radius_search(frame, points, radii, centers, sizes):
    """-> (index int32, squared dist f32, row_splits int64, scale_compat f32)"""
    points = _dev(points, torch.float32)
    radii = _dev(radii, torch.float32)
    centers = _dev(centers, torch.float32)
    sizes = _dev(sizes, torch.float32)
    n, v = points.shape[0], sizes.shape[0]
    rs = torch.empty(v + 1, dtype=torch.int64, device=points.device)
    p = i64(0)
    ctx = context(_same_device(points, radii, centers, sizes))
    ctx.call("asr_hip_multi_radius_search_count", ctypes.byref(frame), ptr(points), i64(n),
             ptr(centers), ptr(sizes), i64(v), ptr(rs), ctypes.byref(p))
    idx = torch.empty(p.value, dtype=torch.int32, device=points.device)
    dist = torch.empty(p.value, dtype=torch.float32, device=points.device)
    compat = torch.empty(p.value, dtype=torch.float32, device=points.device)
    ctx.call("asr_hip_multi_radius_search_fill", ptr(points), ptr(radii), i64(n), ptr(centers),
             ptr(sizes), i64(v), ptr(rs), ptr(idx), ptr(dist), ptr(compat))
    return idx, dist, rs, compat


def knn_radius(frame, points, k, radii=None, radius_fraction=0.5, outlier_threshold=1,
               want_inlier=False):
    """KDTree.compute_k_radius / compute_inlier (cpp/lib/nsearch.cpp:30-86) -> radii[, inlier]"""
    points = _dev(points, torch.float32)
    n = points.shape[0]
    out = torch.empty(n, dtype=torch.float32, device=points.device)
    rin = _dev(radii, torch.float32) if radii is not None else None
    inl = torch.empty(n, dtype=torch.uint8, device=points.device) if want_inlier else None
    context(_same_device(points, rin)).call("asr_hip_knn_radius", ctypes.byref(frame), ptr(points), i64(n), int(k), ptr(rin),
                   ctypes.c_float(radius_fraction), int(outlier_threshold), ptr(out), ptr(inl))
    return (out, inl.bool()) if want_inlier else out


def radius_neighbor_count(frame, points, radii):
    """KDTree.compute_radius_neighbors (cpp/lib/nsearch.cpp:88-105)"""
    points = _dev(points, torch.float32)
    radii = _dev(radii, torch.float32)
    out = torch.empty(points.shape[0], dtype=torch.int64, device=points.device)
    context(_same_device(points, radii)).call("asr_hip_radius_neighbor_count", ctypes.byref(frame), ptr(points), ptr(radii),
                   i64(points.shape[0]), ptr(out))
    return out


def aggregation_importance(compat, dist):
    compat = _dev(compat, torch.float32)
    dist = _dev(dist, torch.float32)
    out = torch.empty_like(compat)
    context(_same_device(compat, dist)).call("asr_hip_aggregation_importance", ptr(compat), ptr(dist), i64(compat.shape[0]),
                   ptr(out))
    return out


def continuous_conv(filters, out_positions, extents, inp_positions, inp_features, neighbors_index,
                    neighbors_importance, neighbors_row_splits, normalize=True, bias=None,
                    relu=False):
    filters = _dev(filters, torch.float32)
    if filters.ndim != 5 or tuple(filters.shape[:3]) != (4, 4, 4):
        raise RuntimeError("continuous_conv: only kernel_size [4,4,4] is implemented")
    cin, cout = filters.shape[3], filters.shape[4]
    out_positions = _dev(out_positions, torch.float32)
    v = out_positions.shape[0]
    extents = _dev(extents, torch.float32).reshape(-1)
    if extents.shape[0] == 1 and v != 1:
        extents = extents.expand(v).contiguous()
    if extents.shape[0] != v:
        raise RuntimeError("continuous_conv: extents must be a scalar or have one entry per output")
    inp_positions = _dev(inp_positions, torch.float32)
    inp_features = _dev(inp_features, torch.float32)
    if inp_features.shape[1] != cin:
        raise RuntimeError("continuous_conv: feature width does not match the filter")
    nidx = _dev(neighbors_index, torch.int32)
    rs = _dev(neighbors_row_splits, torch.int64)
    nimp = None
    if neighbors_importance is not None and neighbors_importance.numel():
        nimp = _dev(neighbors_importance, torch.float32)
    b = _dev(bias, torch.float32) if bias is not None else None
    out = torch.empty((v, cout), dtype=torch.float32, device=filters.device)
    dev = _same_device(filters, out_positions, extents, inp_positions, inp_features, nidx, rs, nimp, b)
    context(dev).call("asr_hip_continuous_conv_f32", ptr(filters), ptr(out_positions), ptr(extents),
                   ptr(inp_positions), ptr(inp_features), ptr(nidx), ptr(nimp), ptr(rs), i64(v),
                   int(cin), int(cout), int(bool(normalize)), ptr(b), int(bool(relu)), ptr(out))
    return out


def continuous_conv_basis(out_positions, extents, inp_positions, inp_features, neighbors_index,
                          neighbors_importance, neighbors_row_splits):
    """per-output interpolation matrices [V, 64*cin] and importance sums [V] of the continuous conv (cin = 4):
    the filter gradient is basis^T (grad / norm) (asr_hip_continuous_conv_basis_f32)"""
    out_positions = _dev(out_positions, torch.float32)
    v = out_positions.shape[0]
    extents = _dev(extents, torch.float32).reshape(-1)
    if extents.shape[0] == 1 and v != 1:
        extents = extents.expand(v).contiguous()
    inp_positions = _dev(inp_positions, torch.float32)
    inp_features = _dev(inp_features, torch.float32)
    cin = inp_features.shape[1]
    nidx = _dev(neighbors_index, torch.int32)
    rs = _dev(neighbors_row_splits, torch.int64)
    nimp = None
    if neighbors_importance is not None and neighbors_importance.numel():
        nimp = _dev(neighbors_importance, torch.float32)
    basis = torch.empty((v, 64 * cin), dtype=torch.float32, device=out_positions.device)
    norm = torch.empty(v, dtype=torch.float32, device=out_positions.device)
    dev = _same_device(out_positions, extents, inp_positions, inp_features, nidx, rs, nimp)
    context(dev).call("asr_hip_continuous_conv_basis_f32", ptr(out_positions), ptr(extents), ptr(inp_positions),
                      ptr(inp_features), ptr(nidx), ptr(nimp), ptr(rs), i64(v), int(cin), ptr(basis), ptr(norm))
    return basis, norm


def sparse_conv(filters, inp_features, neighbors_index, neighbors_kernel_index,
                neighbors_row_splits, inp_importance=None, normalize=False, bias=None, relu=False,
                residual=None, out=None, return_importance=False, algo=0,
                neighbors_importance=None, row_perm=None, filters_b=None, bias_b=None, force_nt=0,
                force_waves=0, num_rows=None):
    """SpecialSparseConv.forward (models/common_torch.py:95-148) in one launch.  filters_b / bias_b:
    optional second filter bank (conv1a + conv1b of a SparseConvBlock in one pass): output columns
    [cout, cout + cout_b); importance, normalize and the returned importance sum then belong to bank b.
    num_rows (with row_perm): compute only the rows row_perm[:num_rows] -- a rank of a sharded run computes the
    rows it owns and leaves the other rows of `out` untouched."""
    filters = _dev(filters, torch.float32)
    K, cin, cout = filters.shape
    inp_features = _dev(inp_features, torch.float32)
    nidx = _dev(neighbors_index, torch.int32)
    nk = _dev(neighbors_kernel_index, torch.uint8)
    rs = _dev(neighbors_row_splits, torch.int64)
    v = rs.shape[0] - 1
    if inp_features.shape[1] != cin:
        raise RuntimeError("sparse_conv: feature width does not match the filter")
    imp = _dev(inp_importance, torch.float32) if inp_importance is not None else None
    nimp = _dev(neighbors_importance, torch.float32) if neighbors_importance is not None else None
    b = _dev(bias, torch.float32) if bias is not None else None
    res = _dev(residual, torch.float32) if residual is not None else None
    fb = _dev(filters_b, torch.float32) if filters_b is not None else None
    bb = _dev(bias_b, torch.float32) if bias_b is not None else None
    if fb is not None and (fb.dim() != 3 or fb.shape[0] != K or fb.shape[1] != cin):
        raise RuntimeError("sparse_conv: second filter bank does not match the first")
    if out is None:
        out = torch.empty((v, cout + (fb.shape[2] if fb is not None else 0)), dtype=torch.float32,
                          device=filters.device)
    oimp = torch.empty(v, dtype=torch.float32, device=filters.device) if return_importance else None
    a = _lib.SparseConvArgs()
    a.filters = filters.data_ptr()
    a.inp_features = inp_features.data_ptr()
    a.inp_ld = inp_features.stride(0)
    a.inp_importance = imp.data_ptr() if imp is not None else None
    a.neighbors_importance = nimp.data_ptr() if nimp is not None else None
    a.neighbors_index = nidx.data_ptr()
    a.neighbors_kernel_index = nk.data_ptr()
    a.neighbors_row_splits = rs.data_ptr()
    if num_rows is not None:
        if row_perm is None or not 0 <= int(num_rows) <= row_perm.shape[0]:
            raise RuntimeError("sparse_conv: num_rows needs a row_perm with at least that many rows")
        a.num_out = int(num_rows)
    else:
        a.num_out = v
    a.num_inp = inp_features.shape[0]
    a.kernel_size = K
    a.cin = cin
    a.cout = cout
    a.normalize = int(bool(normalize))
    a.bias = b.data_ptr() if b is not None else None
    a.relu = int(bool(relu))
    a.residual = res.data_ptr() if res is not None else None
    a.residual_ld = res.stride(0) if res is not None else 0
    a.out = out.data_ptr()
    a.out_ld = out.stride(0)
    a.out_importance = oimp.data_ptr() if oimp is not None else None
    a.algo = int(algo)
    perm = _dev(row_perm, torch.int32) if row_perm is not None else None
    a.row_perm = perm.data_ptr() if perm is not None else None
    a.filters_b = fb.data_ptr() if fb is not None else None
    a.bias_b = bb.data_ptr() if bb is not None else None
    a.cout_b = fb.shape[2] if fb is not None else 0
    a.force_nt = int(force_nt)
    a.force_waves = int(force_waves)
    dev = _same_device(filters, inp_features, nidx, nk, rs, imp, nimp, b, res, fb, bb, out, perm)
    context(dev).call("asr_hip_sparse_conv_f32", ctypes.byref(a))
    if return_importance:
        return out, oimp
    return out


def pack_filters(filters, mode, filters_b=None):
    """re-packed 16-bit copy of a filter tensor [K, cin, cout] (+ second bank [K, cin, cout_b]) for
    sparse_conv16 (asr_hip_sparse_conv_pack); mode: "f16", "bf16x3" or "f16x2".  Pack once per weight tensor."""
    m = _lib.PRECISIONS[mode]
    filters = _dev(filters, torch.float32)
    fb = _dev(filters_b, torch.float32) if filters_b is not None else None
    K, cin, cout = filters.shape
    cb = fb.shape[2] if fb is not None else 0
    nbytes = _lib.load().asr_hip_sparse_conv_packed_bytes(m, int(K), int(cin), int(cout), int(cb))
    if nbytes == 0:
        raise RuntimeError("pack_filters: bad shape / mode")
    out = torch.empty(nbytes, dtype=torch.uint8, device=filters.device)
    context(_same_device(filters, fb)).call("asr_hip_sparse_conv_pack", m, ptr(filters), ptr(fb), int(K), int(cin),
                                            int(cout), int(cb), ptr(out))
    return out


class ConvPlan:
    """Row-group plan of one neighbour list (asr_hip_sparse_conv_plan_create): build once, pass as plan= to every
    sparse_conv16 over the same (neighbour arrays, row_perm, num_rows).  Keeps the arrays it was built from alive."""

    def __init__(self, kernel_size, neighbors_index, neighbors_kernel_index, neighbors_row_splits, row_perm=None,
                 num_rows=None):
        self.nidx = _dev(neighbors_index, torch.int32)
        self.nk = _dev(neighbors_kernel_index, torch.uint8)
        self.rs = _dev(neighbors_row_splits, torch.int64)
        self.perm = _dev(row_perm, torch.int32) if row_perm is not None else None
        self.num_rows = self.rs.shape[0] - 1 if num_rows is None else int(num_rows)
        self.kernel_size = int(kernel_size)
        self.ctx = context(_same_device(self.nidx, self.nk, self.rs, self.perm))
        h = ctypes.c_void_p(0)
        self.ctx.call("asr_hip_sparse_conv_plan_create", ptr(self.nidx), ptr(self.nk), ptr(self.rs),
                      ptr(self.perm) if self.perm is not None else ctypes.c_void_p(0), i64(self.num_rows),
                      ctypes.c_int(self.kernel_size), ctypes.byref(h))
        self.handle = h

    def nbytes(self):
        f = self.ctx.lib.asr_hip_sparse_conv_plan_bytes
        f.restype = ctypes.c_size_t
        return int(f(self.handle))

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            try:
                self.ctx.lib.asr_hip_sparse_conv_plan_destroy(h)
            except Exception:
                pass


def sparse_conv16(mode, packed, kernel_size, cin, cout, inp_features, neighbors_index, neighbors_kernel_index,
                  neighbors_row_splits, inp_importance=None, normalize=False, bias=None, relu=False, residual=None,
                  out=None, out_dtype=None, return_importance=False, neighbors_importance=None, row_perm=None,
                  num_rows=None, cout_b=0, bias_b=None, force_nt=0, force_waves=0, plan=None, inp_absmax=None,
                  out_absmax=None):
    """SpecialSparseConv.forward on the 16-bit matrix cores (asr_hip_sparse_conv_f16 / _bf16x3 / _f16x2).
    mode "f16": inp_features / residual are float16 tensors, out is float16 (default) or float32;
    mode "bf16x3" / "f16x2": float32 in and out, fp32-class result.  packed: pack_filters(filters, mode[, filters_b]).
    plan: ConvPlan of this list (one is built for the call otherwise).
    inp_absmax / out_absmax: int32 device scalars with the f32 bits of the largest |element| of the input (f16x2; None:
    computed by one pass) and the running maximum of what is written to out (new_absmax(); the caller zeroes it)."""
    m = _lib.PRECISIONS[mode]
    act = torch.float16 if mode == "f16" else torch.float32
    inp_features = _dev(inp_features, act)
    nidx = _dev(neighbors_index, torch.int32)
    nk = _dev(neighbors_kernel_index, torch.uint8)
    rs = _dev(neighbors_row_splits, torch.int64)
    v = rs.shape[0] - 1
    if inp_features.shape[1] != cin:
        raise RuntimeError("sparse_conv16: feature width does not match the filter")
    imp = _dev(inp_importance, torch.float32) if inp_importance is not None else None
    nimp = _dev(neighbors_importance, torch.float32) if neighbors_importance is not None else None
    b = _dev(bias, torch.float32) if bias is not None else None
    bb = _dev(bias_b, torch.float32) if bias_b is not None else None
    res = _dev(residual, act) if residual is not None else None
    if out_dtype is None:
        out_dtype = out.dtype if out is not None else act
    if out_dtype not in (torch.float32, act):
        raise RuntimeError("sparse_conv16: out must be float32 or the activation type")
    if out is None:
        out = torch.empty((v, cout + cout_b), dtype=out_dtype, device=inp_features.device)
    oimp = torch.empty(v, dtype=torch.float32, device=inp_features.device) if return_importance else None
    perm = _dev(row_perm, torch.int32) if row_perm is not None else None
    a = _lib.SparseConvArgs()
    a.inp_features = inp_features.data_ptr()
    a.inp_ld = inp_features.stride(0)
    a.inp_importance = imp.data_ptr() if imp is not None else None
    a.neighbors_importance = nimp.data_ptr() if nimp is not None else None
    a.neighbors_index = nidx.data_ptr()
    a.neighbors_kernel_index = nk.data_ptr()
    a.neighbors_row_splits = rs.data_ptr()
    a.num_out = v if num_rows is None else int(num_rows)
    a.num_inp = inp_features.shape[0]
    a.kernel_size, a.cin, a.cout, a.cout_b = int(kernel_size), int(cin), int(cout), int(cout_b)
    a.normalize = int(bool(normalize))
    a.bias = b.data_ptr() if b is not None else None
    a.bias_b = bb.data_ptr() if bb is not None else None
    a.relu = int(bool(relu))
    a.residual = res.data_ptr() if res is not None else None
    a.residual_ld = res.stride(0) if res is not None else 0
    a.out = out.data_ptr()
    a.out_ld = out.stride(0)
    a.out_importance = oimp.data_ptr() if oimp is not None else None
    a.row_perm = perm.data_ptr() if perm is not None else None
    a.force_nt, a.force_waves = int(force_nt), int(force_waves)
    if plan is not None:
        if plan.rs.data_ptr() != rs.data_ptr() or plan.nidx.data_ptr() != nidx.data_ptr() or \
                (plan.perm.data_ptr() if plan.perm is not None else None) != (perm.data_ptr() if perm is not None else None):
            raise RuntimeError("sparse_conv16: the plan belongs to other neighbour arrays")
        a.plan = plan.handle
    a.inp_absmax = inp_absmax.data_ptr() if inp_absmax is not None else None
    a.out_absmax = out_absmax.data_ptr() if out_absmax is not None else None
    ctx = context(_same_device(packed, inp_features, nidx, nk, rs, imp, nimp, b, bb, res, out, perm))
    if mode == "f16":
        ctx.call("asr_hip_sparse_conv_f16", ctypes.byref(a), ptr(packed), int(out.dtype == torch.float16))
    elif mode == "f16x2":
        ctx.call("asr_hip_sparse_conv_f16x2", ctypes.byref(a), ptr(packed))
    else:
        ctx.call("asr_hip_sparse_conv_bf16x3", ctypes.byref(a), ptr(packed))
    return (out, oimp) if return_importance else out


def absmax(x, out=None):
    """f32 bits of the largest |element| of a float32 matrix as an int32 device scalar (asr_hip_absmax_f32): the
    inp_absmax of sparse_conv16 in mode "f16x2" """
    x = _dev(x, torch.float32)
    if out is None:
        out = torch.zeros(1, dtype=torch.int32, device=x.device)
    context(x.device).call("asr_hip_absmax_f32", ptr(x), i64(x.shape[0]), ctypes.c_int(x.shape[1]), i64(x.stride(0)),
                           ptr(out))
    return out


def row_groups(neighbors_kernel_index, neighbors_row_splits, segment_rows=0):
    """MFMA tiling order of a CSR's rows (asr_hip_row_groups)"""
    nk = _dev(neighbors_kernel_index, torch.uint8)
    rs = _dev(neighbors_row_splits, torch.int64)
    v = rs.shape[0] - 1
    perm = torch.empty(v, dtype=torch.int32, device=rs.device)
    context(_same_device(nk, rs)).call("asr_hip_row_groups", ptr(nk), ptr(rs), i64(v), i64(segment_rows), ptr(perm))
    return perm


def invert_neighbors_list(num_points, inp_neighbors_index, inp_neighbors_row_splits,
                          inp_neighbors_attributes=None):
    idx = _dev(inp_neighbors_index, torch.int32)
    rs = _dev(inp_neighbors_row_splits, torch.int64)
    attr = None
    if inp_neighbors_attributes is not None and inp_neighbors_attributes.numel():
        attr = _dev(inp_neighbors_attributes, torch.uint8)
    p = idx.shape[0]
    out_idx = torch.empty(p, dtype=torch.int32, device=idx.device)
    out_rs = torch.empty(num_points + 1, dtype=torch.int64, device=idx.device)
    out_attr = torch.empty(p if attr is not None else 0, dtype=torch.uint8, device=idx.device)
    context(_same_device(idx, rs, attr)).call("asr_hip_invert_neighbors_list", i64(num_points), ptr(idx), ptr(rs),
                   i64(rs.shape[0] - 1), ptr(attr), ptr(out_idx), ptr(out_rs),
                   ptr(out_attr) if attr is not None else ctypes.c_void_p(0))
    return out_idx, out_rs, out_attr


def reduce_subarrays_sum(values, row_splits, gather_index=None):
    values = _dev(values, torch.float32)
    rs = _dev(row_splits, torch.int64)
    g = _dev(gather_index, torch.int32) if gather_index is not None else None
    out = torch.empty(rs.shape[0] - 1, dtype=torch.float32, device=values.device)
    context(_same_device(values, rs, g)).call("asr_hip_reduce_subarrays_sum", ptr(values), ptr(g), ptr(rs),
                   i64(rs.shape[0] - 1), ptr(out))
    return out


def decode_mlp(code, w1, b1, w2, b2, w3, voxel_sizes=None):
    code = _dev(code, torch.float32)
    w1, b1, w2, b2, w3 = (_dev(t, torch.float32) for t in (w1, b1, w2, b2, w3))
    sizes = _dev(voxel_sizes, torch.float32) if voxel_sizes is not None else None
    v, c = code.shape
    out = torch.empty((v, 2), dtype=torch.float32, device=code.device)
    dev = _same_device(code, w1, b1, w2, b2, w3, sizes)
    context(dev).call("asr_hip_decode_mlp", ptr(code), i64(v), int(c), ptr(w1), ptr(b1),
                   int(w1.shape[0]), ptr(w2), ptr(b2), int(w2.shape[0]), ptr(w3), ptr(sizes),
                   ptr(out))
    return out


def density_inlier(counts, density_percentile_threshold):
    """asr::ComputeInlierFromDensity (cpp/lib/preprocess.cpp:41-62) on host counts, literally
    (see asr_density_inlier in include/asr_hip.h)"""
    import numpy as np
    counts = np.ascontiguousarray(counts, np.int64)
    out = np.zeros(counts.shape[0], np.uint8)
    rc = _lib.load().asr_density_inlier(counts.ctypes.data_as(ctypes.c_void_p), i64(counts.shape[0]),
                                        ctypes.c_double(density_percentile_threshold),
                                        out.ctypes.data_as(ctypes.c_void_p))
    if rc:
        raise AsrHipError("asr_density_inlier failed (%d)" % rc)
    return out.astype(bool)

"""`adaptivesurfacereconstruction` drop-in for the hot-path part of the reference's pybind module
(cpp/pybind/module.cpp:279-491, re-exported by python/adaptivesurfacereconstruction/__init__.py).

Same function names, keyword defaults, numpy in / numpy out, ValueError for shape errors and
RuntimeError from the library.  Everything runs on the MI355X through libasr_hip.so; the rows of
SURVEY section 8 marked "next" that are not built yet (contouring, component filter) raise
NotImplementedError instead of silently falling back to a CPU path.
"""
import numpy as np
import torch

from asr_hip import _lib, ops as _ops
from asr_hip._lib import AsrHipError  # noqa: F401

__version__ = _lib.load().asr_hip_version().decode() if True else None


def get_version_str():
    """cpp/pybind/module.cpp:284-286"""
    return __version__


def _f32(a, name, shape_msg, ndim, last=None):
    a = np.ascontiguousarray(a, dtype=np.float32)  # forcecast, module.cpp:59-61
    if a.ndim != ndim or (last is not None and a.shape[-1] != last):
        raise ValueError("%s must have shape %s" % (name, shape_msg))
    return a


def _u64(t):
    return t.cpu().numpy().view(np.uint64)


class Octree:
    """Opaque handle returned by create_octree (module.cpp:282): sorted node and leaf keys stay on
    the GPU together with the octree frame."""

    _last = None

    def __init__(self, frame, nodes, leaves):
        Octree._last = self
        self.frame = frame
        self.nodes = nodes
        self.leaves = leaves

    def __repr__(self):
        return "<Octree nodes=%d leaves=%d>" % (self.nodes.shape[0], self.leaves.shape[0])


def create_octree(points, radii, bb_min, bb_max, radius_scale=1.0, grow_steps=0, max_depth=21):
    """module.cpp:144-161,372-400 -> asr::CreateOctreeFromPoints (cpp/lib/octree.cpp:230-280)"""
    points = _f32(points, "points", "[N,3]", 2, 3)
    radii = _f32(radii, "radii", "[N]", 1)
    if radii.shape[0] != points.shape[0]:
        raise ValueError("radii must have shape [N]")
    if grow_steps != 0:
        raise RuntimeError("grow_steps != 0 is never used on the reconstruction path "
                           "(cpp/lib/asr.cpp:151-153) and is not implemented")
    frame = _lib.frame_init(np.asarray(bb_min, np.float32), np.asarray(bb_max, np.float32))
    dev = torch.device("cuda")
    nodes, leaves = _ops.octree_build(frame, torch.from_numpy(points).to(dev),
                                      torch.from_numpy(radii).to(dev), radius_scale, max_depth)
    return Octree(frame, nodes, leaves)


def create_grids_from_octree(tree, num_levels, voxel_info_all_levels=False):
    """module.cpp:163-228,402-441 -> asr::CreateGridsFromOctree (cpp/lib/grid.cpp:245-314).
    Empty arrays are omitted from the dicts like in the reference."""
    result = []
    keys = tree.leaves
    for i in range(num_levels):
        d = {}
        up = None
        if i + 1 < num_levels:
            nxt, up_idx, up_kidx, up_rs = _ops.grid_coarsen(keys)
            up = (up_idx, up_kidx, up_rs)
        if i == 0 or voxel_info_all_levels:
            centers, sizes = _ops.voxel_info(tree.frame, keys)
            if keys.numel():
                d["voxel_keys"] = _u64(keys)
                d["voxel_centers"] = centers.cpu().numpy()
                d["voxel_sizes"] = sizes.cpu().numpy()
        idx, kidx, rs = _ops.grid_neighbors(keys)
        if idx.numel():
            d["neighbors_index"] = idx.cpu().numpy()
            d["neighbors_kernel_index"] = kidx.cpu().numpy()
        d["neighbors_row_splits"] = rs.cpu().numpy()
        if up is not None and up[0].numel():
            d["up_neighbors_index"] = up[0].cpu().numpy()
            d["up_neighbors_kernel_index"] = up[1].cpu().numpy()
            d["up_neighbors_row_splits"] = up[2].cpu().numpy()
        result.append(d)
        if up is not None:
            keys = nxt
    return result


def compute_aggregation_neighbors(tree, points, radii, voxel_centers, voxel_sizes):
    """asr::ComputeAggregationNeighborsAndScaleCompatibility (cpp/lib/nsearch.cpp:107-162); the
    reference keeps this internal to ReconstructSurface, models/v0/datareader.py:776-795 does the
    same with open3d.core.nns."""
    dev = torch.device("cuda")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)  # noqa: E731
    idx, dist, rs, compat = _ops.multi_radius_search(tree.frame, t(points), t(radii),
                                                     t(voxel_centers), t(voxel_sizes))
    return {"aggregation_neighbors_index": idx.cpu().numpy(),
            "aggregation_neighbors_dist": dist.cpu().numpy(),
            "aggregation_row_splits": rs.cpu().numpy(),
            "aggregation_scale_compat": compat.cpu().numpy()}


def _next_row(what, where):
    raise NotImplementedError("%s (%s) is a 'next' row of the hot-path scope table and is not "
                              "implemented on the MI355X path yet" % (what, where))


def create_dual_vertex_indices(tree):
    """module.cpp:230-235,443-453 -> asr::CreateDualVertexIndices (cpp/lib/grid.cpp:450-459):
    uint64 [D,8] indices into the leaves.  Must be called while `tree` is the octree built last on
    this process' context (the node set lives there)."""
    if tree is not Octree._last:
        raise RuntimeError("create_dual_vertex_indices needs the most recently created octree")
    return _ops.dual_cells(tree.leaves.device).cpu().numpy().astype(np.uint64)


def reconstruct_surface(points, normals, radii=np.empty((0,), np.float32), point_radius_scale=1.0,
                        density_percentile_threshold=10.0, point_radius_estimation_knn=24,
                        octree_max_depth=21, contouring_value_threshold=1.0,
                        keep_n_connected_components=2**63 - 1, minimum_component_size=3):
    _next_row("reconstruct_surface (pre-filter + contouring + component filter)",
              "cpp/lib/asr.cpp:116-135,340-346; use asr_hip.pipeline.ImplicitPipeline for the "
              "implicit values")


def remove_connected_components(vertices, triangles, keep_n_largest_components,
                                minimum_component_size=3):
    _next_row("remove_connected_components", "cpp/lib/postprocess.cpp:141")


class KDTree:
    """cpp/pybind/module.cpp:237-277,455-489 -> asr::KDTree (cpp/lib/nsearch.cpp:23-105).  The
    reference builds a nanoflann tree; here the points are Morton sorted on the GPU and all three
    queries are exact grid searches (asr_hip_knn_radius / asr_hip_radius_neighbor_count)."""

    def __init__(self, points):
        points = np.ascontiguousarray(points, dtype=np.float32)
        if points.ndim != 2 or points.shape[1] != 3:
            raise ValueError("points must have shape [N,3]")
        self._points = torch.from_numpy(points).to(torch.device("cuda"))
        lo, hi = points.min(0), points.max(0)
        m = max(1e-3, 1e-3 * float((hi - lo).max()))
        self._frame = _lib.frame_init(lo - np.float32(m), hi + np.float32(m))

    def compute_k_radius(self, k):
        return _ops.knn_radius(self._frame, self._points, k).cpu().numpy()

    def compute_inlier(self, radii, radius_fraction=0.5, k=24, outlier_threshold=1):
        radii = np.ascontiguousarray(radii, dtype=np.float32)
        if radii.ndim != 1 or radii.shape[0] != self._points.shape[0]:
            raise ValueError("radii must have shape [num_points]")
        _, inl = _ops.knn_radius(self._frame, self._points, k, torch.from_numpy(radii).to(self._points.device),
                                 radius_fraction, outlier_threshold, want_inlier=True)
        return inl.cpu().numpy()

    def compute_radius_neighbors(self, radii):
        radii = np.ascontiguousarray(radii, dtype=np.float32)
        cnt = _ops.radius_neighbor_count(self._frame, self._points,
                                         torch.from_numpy(radii).to(self._points.device))
        return [int(c) for c in cnt.cpu().tolist()]

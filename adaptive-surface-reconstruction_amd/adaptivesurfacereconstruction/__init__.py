"""`adaptivesurfacereconstruction` drop-in for the hot-path part of the reference's pybind module
(cpp/pybind/module.cpp:279-491, re-exported by python/adaptivesurfacereconstruction/__init__.py).

Same function names, keyword defaults, numpy in / numpy out, ValueError for shape errors and
RuntimeError from the library.  Everything runs on the MI355X through libasr_hip.so (no CPU
fallback), including the "next" rows of SURVEY section 8: pre-filter, dual cells, contouring and
component filter.
"""
import numpy as np
import torch

from asr_hip import _lib, ops as _ops
from asr_hip._lib import AsrHipError  # noqa: F401

__version__ = _lib.load().asr_hip_version().decode() if True else None


def get_version_str():
    """cpp/pybind/module.cpp:284-286"""
    return __version__


def set_print_callback_function(print_callback, levels=(0, 1, 2, 3)):
    """asr::SetPrintCallbackFunction (cpp/lib/asr.hpp:29-34; C++ API of the reference, not in its pybind module):
    stage banners and messages of the library go to `print_callback(str)`"""
    _lib.set_print_callback_function(print_callback, levels)


def _f32(a, name, shape_msg, ndim, last=None):
    a = np.ascontiguousarray(a, dtype=np.float32)  # forcecast, module.cpp:59-61
    if a.ndim != ndim or (last is not None and a.shape[-1] != last):
        raise ValueError("%s must have shape %s" % (name, shape_msg))
    return a


def _u64(t):
    return t.cpu().numpy().view(np.uint64)


def get_third_party_notices():
    """cpp/pybind/module.cpp:287-289.  The reference returns the licence texts of the libraries linked into its
    binary (Eigen, libcuckoo, nanoflann, TBB, Open3D, PyTorch); none of those is linked into libasr_hip.so."""
    return ("libasr_hip.so links the ROCm runtime (HIP) and uses the header-only rocPRIM (MIT licence, "
            "Copyright (c) Advanced Micro Devices, Inc.).  PyTorch (BSD-3-Clause) provides device memory and "
            "torch.distributed on the host side.  No code of the reference implementation or of its "
            "third-party dependencies is included.")


class Octree:
    """Opaque handle returned by create_octree (module.cpp:282): owns the sorted node and leaf keys (GPU
    tensors) and the octree frame, so it stays valid however many trees are built afterwards."""

    def __init__(self, frame, nodes, leaves):
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
    if grow_steps < 0:
        raise ValueError("grow_steps must be >= 0")
    frame = _lib.frame_init(np.asarray(bb_min, np.float32), np.asarray(bb_max, np.float32))
    dev = torch.device("cuda")
    nodes, leaves = _ops.octree_build(frame, torch.from_numpy(points).to(dev),
                                      torch.from_numpy(radii).to(dev), radius_scale, max_depth, int(grow_steps))
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


def create_dual_vertex_indices(tree):
    """module.cpp:230-235,443-453 -> asr::CreateDualVertexIndices (cpp/lib/grid.cpp:450-459):
    uint64 [D,8] indices into the leaves, for any live tree (models/v0/datareader.py:224-243 keeps several)."""
    return _ops.dual_cells(tree.leaves.device, nodes=tree.nodes, leaves=tree.leaves).cpu().numpy().astype(np.uint64)


def create_triangle_mesh(values, dual_vertex_indices, node_positions, contouring_value_threshold=1.0):
    """asr::CreateTriangleMesh (cpp/lib/contouring.cpp:29-460), internal to ReconstructSurface in the
    reference (cpp/lib/asr.cpp:340-342): values [V,2], dual_vertex_indices [D,8], node_positions
    [V,3] -> {'vertices': f32[M,3], 'triangles': i32[T,3]}"""
    dev = torch.device("cuda")
    values = _f32(values, "values", "[V,2]", 2, 2)
    pos = _f32(node_positions, "node_positions", "[V,3]", 2, 3)
    duals = np.ascontiguousarray(dual_vertex_indices)
    if duals.ndim != 2 or duals.shape[1] != 8:
        raise ValueError("dual_vertex_indices must have shape [D,8]")
    duals = duals.astype(np.int64, copy=False) if duals.dtype != np.uint64 else duals.view(np.int64)
    if duals.size and (duals.min() < 0 or duals.max() >= values.shape[0]):
        raise ValueError("dual_vertex_indices out of range")
    v, t = _ops.contour(torch.from_numpy(values).to(dev), torch.from_numpy(duals).to(dev),
                        torch.from_numpy(pos).to(dev), contouring_value_threshold)
    return {"vertices": v.cpu().numpy(), "triangles": t.cpu().numpy()}


def _load_weights(weights):
    """state dict of the network (names of UNet5.state_dict()).  The reference loads the TorchScript
    file <resource dir>/model.pt (cpp/lib/asr.cpp:138-139); here: a dict, an .npz, or a torch file
    holding a state dict or a TorchScript module, by argument or as
    $ASR_RESOURCE_DIR/{model_weights.npz, model_weights.pt, model.pt}."""
    import os
    if weights is None:
        base = os.environ.get("ASR_RESOURCE_DIR", "")
        for cand in ("model_weights.npz", "model_weights.pt", "model.pt"):
            if base and os.path.exists(os.path.join(base, cand)):
                weights = os.path.join(base, cand)
                break
        if weights is None:
            raise RuntimeError("no network weights: pass weights=... or set ASR_RESOURCE_DIR to a directory "
                               "with model_weights.npz / model_weights.pt")
    if isinstance(weights, str):
        if weights.endswith(".npz"):
            with np.load(weights) as z:
                return {k: z[k] for k in z.files}
        try:  # the reference ships a TorchScript archive (model.pt, cpp/lib/asr.cpp:138-139)
            sd = torch.jit.load(weights, map_location="cpu")
        except RuntimeError:  # not a TorchScript archive: a pickled state dict
            sd = torch.load(weights, map_location="cpu")
        sd = dict(sd.state_dict()) if hasattr(sd, "state_dict") else dict(sd)
        return {k: v for k, v in sd.items() if isinstance(v, torch.Tensor)}
    return weights


def reconstruct_surface(points, normals, radii=np.empty((0,), np.float32), point_radius_scale=1.0,
                        density_percentile_threshold=10.0, point_radius_estimation_knn=24,
                        octree_max_depth=21, contouring_value_threshold=1.0,
                        keep_n_connected_components=2**63 - 1, minimum_component_size=3, *, weights=None):
    """module.cpp:58-109,291-346 -> asr::ReconstructSurface (cpp/lib/asr.cpp:95-349): pre-filter,
    implicit values, dual contouring, component filter; every stage on the MI355X.
    `weights` (keyword only) replaces the reference's bundled model.pt, see _load_weights."""
    from asr_hip.pipeline import ImplicitPipeline
    points = _f32(points, "points", "[num_points,3]", 2, 3)
    normals = _f32(normals, "normals", "[num_points,3]", 2, 3)
    radii = np.ascontiguousarray(radii, dtype=np.float32)
    if normals.shape != points.shape:
        raise ValueError("normals must have shape [num_points,3]")
    if radii.ndim != 1 or radii.shape[0] not in (0, points.shape[0]):
        raise ValueError("radii must have shape [num_point3]")
    if points.shape[0] == 0:
        raise RuntimeError("points is null!\n")
    # preprocess (asr.cpp:116-135)
    _lib.library_print("preprocessing\n", _lib.PRINT_LEVELS["INFO"])  # asr.cpp:117
    tree = KDTree(points)
    if radii.shape[0]:
        counts = _ops.radius_neighbor_count(tree._frame, tree._points, torch.from_numpy(radii).to(tree._points.device))
        inlier = _ops.density_inlier(counts.cpu().numpy(), density_percentile_threshold)
    else:
        r = _ops.knn_radius(tree._frame, tree._points, point_radius_estimation_knn)
        _, inl = _ops.knn_radius(tree._frame, tree._points, point_radius_estimation_knn, r, 0.5, 1,
                                 want_inlier=True)
        radii = r.cpu().numpy()
        inlier = inl.cpu().numpy().astype(bool)
    points, normals, radii = points[inlier], normals[inlier], radii[inlier]
    if points.shape[0] == 0:
        raise RuntimeError("no points left after the pre-filter")
    dev = torch.device("cuda")
    pipe = ImplicitPipeline(_load_weights(weights), device="cuda:%d" % torch.cuda.current_device(),
                            point_radius_scale=point_radius_scale, octree_max_depth=octree_max_depth,
                            scale_sdf=True)
    # exact bounding box of the filtered points (asr.cpp:148-150; quirk B.1 applies)
    bb_min, bb_max = points.min(0), points.max(0)
    pipe.forward(torch.from_numpy(points).to(dev), torch.from_numpy(normals).to(dev),
                 torch.from_numpy(radii).to(dev), bb_min, bb_max)
    v, t = pipe.mesh(contouring_value_threshold, keep_n_connected_components, minimum_component_size)
    return {"vertices": v.cpu().numpy(), "triangles": t.cpu().numpy()}


def remove_connected_components(vertices, triangles, keep_n_largest_components,
                                minimum_component_size=3):
    """module.cpp:111-142,348-370 -> asr::RemoveConnectedComponents (cpp/lib/postprocess.cpp:141-176)"""
    vertices = _f32(vertices, "vertices", "[N,3]", 2, 3)
    triangles = np.ascontiguousarray(triangles, dtype=np.int32)
    if triangles.ndim != 2 or triangles.shape[1] != 3:
        raise ValueError("triangles must have shape [N,3]")
    dev = torch.device("cuda")
    v, t = _ops.remove_components(torch.from_numpy(vertices).to(dev), torch.from_numpy(triangles).to(dev),
                                  keep_n_largest_components, minimum_component_size)
    return {"vertices": v.cpu().numpy(), "triangles": t.cpu().numpy()}


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

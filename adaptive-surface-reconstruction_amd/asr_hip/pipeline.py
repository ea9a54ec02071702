"""Whole hot path: device-resident points / normals / radii -> values[V0, 2].

Host-side mirror of the middle of asr::ReconstructSurface (cpp/lib/asr.cpp:143-336) with the
network methods aggregate / unet / decode of models/v0/net_definitions_torch.py:535-666.  All
work happens inside libasr_hip.so (asr_hip_implicit_forward); this class only marshals pointers.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import AsrHipError, Context, ImplicitParams, ImplicitSizes, Weight, ptr

# dtype / trailing shape of the arrays asr_hip_implicit_get can return (input_dict keys of
# cpp/lib/asr.cpp:159-312)
_ARRAY_TYPES = {
    "values": (torch.float32, 2),
    "feats1": (torch.float32, None),
    "importance": (torch.float32, 0),
    "code": (torch.float32, None),
    "nodes": (torch.int64, 0),
    "voxel_keys": (torch.int64, 0),
    "voxel_centers": (torch.float32, 3),
    "voxel_sizes": (torch.float32, 0),
    "neighbors_index": (torch.int32, 0),
    "neighbors_kernel_index": (torch.uint8, 0),
    "neighbors_row_splits": (torch.int64, 0),
    "up_neighbors_index": (torch.int32, 0),
    "up_neighbors_kernel_index": (torch.uint8, 0),
    "up_neighbors_row_splits": (torch.int64, 0),
    "down_neighbors_index": (torch.int32, 0),
    "down_neighbors_kernel_index": (torch.uint8, 0),
    "down_neighbors_row_splits": (torch.int64, 0),
    "tiling": (torch.int32, 0),
    "tiling_up": (torch.int32, 0),
    "tiling_down": (torch.int32, 0),
    "aggregation_neighbors_index": (torch.int32, 0),
    "aggregation_neighbors_dist": (torch.float32, 0),
    "aggregation_row_splits": (torch.int64, 0),
    "aggregation_scale_compat": (torch.float32, 0),
}


class ImplicitPipeline:
    """weights: dict state_dict-name -> tensor (names of SURVEY A.6 / UNet5.state_dict())."""

    # grids and aggregation_search overlap (two streams) unless option "overlap" is 0; geometry_wall /
    # network_wall are the wall times of the two halves
    STAGES = ("octree", "grids", "aggregation_search", "continuous_conv", "unet", "decode",
              "geometry_wall", "network_wall")

    def __init__(self, weights, device="cuda:0", point_radius_scale=1.0, octree_max_depth=21,
                 scale_sdf=True, precision="f32"):
        """precision: arithmetic of the 53 sparse convs -- "f32" (exact f32 MFMA, the reference's type),
        "f16" (f16 activations and weights, f32 accumulate: BASELINE config C5), "bf16x3" (exact three-way
        bf16 split on the bf16 matrix cores, six MFMAs per product: fp32-class results) or "f16x2" (per-tensor
        power-of-two scaling + two-way f16 split, three MFMAs per product: fp32-class results, what bench.py times)"""
        if precision not in _lib.PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(_lib.PRECISIONS))
        self.precision = precision
        if not torch.cuda.is_available():
            raise AsrHipError("no GPU visible: the MI355X path cannot run (no CPU fallback)")
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        with torch.cuda.device(self.device):  # the context binds to the device current at creation
            self.ctx = Context()
        self.point_radius_scale = float(point_radius_scale)
        self.octree_max_depth = int(octree_max_depth)
        self.scale_sdf = bool(scale_sdf)
        self._weights = {}
        self._names = []
        for name, t in weights.items():
            if not isinstance(t, torch.Tensor):
                t = torch.from_numpy(np.ascontiguousarray(t))
            self._weights[name] = t.detach().to(self.device, torch.float32).contiguous()
        self._table = (Weight * len(self._weights))()
        for i, (name, t) in enumerate(self._weights.items()):
            self._names.append(name.encode())
            self._table[i].name = self._names[-1]
            self._table[i].data = t.data_ptr()
            self._table[i].ndim = t.ndim
            for d in range(t.ndim):
                self._table[i].shape[d] = t.shape[d]
        self.sizes = None

    def _params(self, bb_min, bb_max):
        p = ImplicitParams()
        p.point_radius_scale = self.point_radius_scale
        p.octree_max_depth = self.octree_max_depth
        for d in range(3):
            p.bb_min[d] = float(bb_min[d])
            p.bb_max[d] = float(bb_max[d])
        p.scale_sdf = int(self.scale_sdf)
        p.precision = _lib.PRECISIONS[self.precision]
        return p

    def _stream(self):
        """torch's current stream ON THE PIPELINE'S DEVICE (not on whatever device is current)"""
        self.ctx.set_stream(torch.cuda.current_stream(self.device))

    def _check(self, points, normals, radii):
        for t in (points, normals, radii):
            if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and
                    t.is_contiguous()):
                raise AsrHipError("inputs must be contiguous float32 GPU tensors")
            if t.device != self.device:
                raise AsrHipError("input on %s but the pipeline lives on %s" % (t.device, self.device))
        if points.ndim != 2 or points.shape[1] != 3:
            raise ValueError("points must have shape [num_points,3]")
        if normals.shape != points.shape:
            raise ValueError("normals must have shape [num_points,3]")
        if radii.ndim != 1 or radii.shape[0] != points.shape[0]:
            raise ValueError("radii must have shape [num_point3]")
        if points.shape[0] == 0:
            raise RuntimeError("points is null!\n")

    def forward(self, points, normals, radii, bb_min, bb_max):
        """Enqueues the whole path on torch's current stream; returns values[V0,2] (a view into
        the context arena, valid until the next forward)."""
        self._check(points, normals, radii)
        self._stream()
        p = self._params(bb_min, bb_max)
        sizes = ImplicitSizes()
        self.ctx.call("asr_hip_implicit_forward", ptr(points), ptr(normals), ptr(radii),
                      ctypes.c_int64(points.shape[0]), self._table, len(self._weights),
                      ctypes.byref(p), ctypes.byref(sizes))
        self.sizes = sizes
        return self.get("values")

    def forward_sharded(self, comm, points, normals, radii, bb_min, bb_max):
        """The whole path with the network half sharded over the ranks of `comm` (asr_hip.shardcomm: RcclComm over xGMI,
        HostStagedComm for tests) INSIDE the library (asr_hip_implicit_forward_sharded): every rank passes the whole
        cloud, computes the rows it owns and returns the complete values[V0, 2], bit-identical to forward().
        self.shard_stats: owned rows, halo rows, bytes and grouped exchanges of this forward."""
        self._check(points, normals, radii)
        self._stream()
        p = self._params(bb_min, bb_max)
        sizes = ImplicitSizes()
        stats = _lib.ShardStats()
        self.ctx.call("asr_hip_implicit_forward_sharded", comm.handle(), ptr(points), ptr(normals), ptr(radii),
                      ctypes.c_int64(points.shape[0]), self._table, len(self._weights), ctypes.byref(p),
                      ctypes.byref(sizes), ctypes.c_void_p(0), ctypes.byref(stats))
        self.sizes = sizes
        self.shard_stats = {"owned_rows": list(stats.owned_rows), "halo_rows_recv": list(stats.halo_rows_recv),
                            "bytes_sent": int(stats.bytes_sent), "bytes_received": int(stats.bytes_received),
                            "exchanges": int(stats.exchanges), "exchange_seconds": float(stats.exchange_seconds)}
        return self.get("values")

    def build(self, points, radii, bb_min, bb_max):
        """geometry half only (octree, grids, aggregation neighbours)"""
        self._check(points, points, radii)
        self._stream()
        p = self._params(bb_min, bb_max)
        sizes = ImplicitSizes()
        self.ctx.call("asr_hip_implicit_build", ptr(points), ptr(radii),
                      ctypes.c_int64(points.shape[0]), ctypes.byref(p), ctypes.byref(sizes))
        self.sizes = sizes
        return sizes

    def network(self, points, normals, bb_min, bb_max):
        """network half on the structures of the last build()"""
        self._check(points, normals, points[:, 0].contiguous())
        self._stream()
        p = self._params(bb_min, bb_max)
        self.ctx.call("asr_hip_implicit_network", ptr(points), ptr(normals),
                      ctypes.c_int64(points.shape[0]), self._table, len(self._weights),
                      ctypes.byref(p), ctypes.c_void_p(0))
        return self.get("values")

    def aggregate(self, points, normals, bb_min, bb_max):
        """UNet5.aggregate alone on the structures of the last build(): -> (feats1 [V0, C], importance [P_agg])"""
        self._check(points, normals, points[:, 0].contiguous())
        self._stream()
        p = self._params(bb_min, bb_max)
        self.ctx.call("asr_hip_implicit_aggregate", ptr(points), ptr(normals), ctypes.c_int64(points.shape[0]),
                      self._table, len(self._weights), ctypes.byref(p))
        return self.get("feats1"), self.get("importance")

    def get(self, name):
        """copy of one named array of the last forward (see include/asr_hip.h)"""
        base = name if name in _ARRAY_TYPES else name.rstrip("0123456789")
        if base not in _ARRAY_TYPES:
            raise KeyError(name)
        dtype, cols = _ARRAY_TYPES[base]
        nbytes = ctypes.c_size_t(0)
        self.ctx.call("asr_hip_implicit_get", name.encode(), ctypes.c_void_p(0), ctypes.byref(nbytes))
        item = torch.empty((), dtype=dtype).element_size()
        out = torch.empty(nbytes.value // item, dtype=dtype, device=self.device)
        self.ctx.call("asr_hip_implicit_get", name.encode(), ptr(out), ctypes.byref(nbytes))
        if cols is None:
            v0 = int(self.sizes.num_voxels[0])
            out = out.reshape(v0, -1)
        elif cols:
            out = out.reshape(-1, cols)
        return out

    def dual_cells(self):
        """dual_vertex_indices [D,8] of the octree of the last forward/build (cpp/lib/asr.cpp:154)"""
        from . import ops
        self._stream()
        return ops.dual_cells(self.device, ctx=self.ctx)

    def mesh(self, contouring_value_threshold=1.0, keep_n_connected_components=2**63 - 1,
             minimum_component_size=3, values=None):
        """contouring + component filter on the last forward (cpp/lib/asr.cpp:338-346) ->
        (vertices f32[M,3], triangles i32[T,3]) on the GPU"""
        from . import ops
        self._stream()
        if values is None:
            values = self.get("values")
        duals = ops.dual_cells(self.device, ctx=self.ctx)
        v, t = ops.contour(values, duals, self.get("voxel_centers0"), contouring_value_threshold, ctx=self.ctx)
        return ops.remove_components(v, t, keep_n_connected_components, minimum_component_size, ctx=self.ctx)

    def stage_ms(self):
        ms = (ctypes.c_float * 8)()
        self.ctx.call("asr_hip_implicit_stage_ms", ms)
        return dict(zip(self.STAGES, [float(x) for x in ms]))

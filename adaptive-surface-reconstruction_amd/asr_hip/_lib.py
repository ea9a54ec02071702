"""ctypes binding of libasr_hip.so (C ABI: include/asr_hip.h).

The product path has no CPU fallback: if the HIP library is missing or no GPU is present the
calls below raise.  Device memory comes from torch (plumbing only): every pointer handed to
the library is tensor.data_ptr() of a contiguous CUDA(HIP) tensor.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "libasr_hip.so")

ASR_MAX_LEVEL = 21
ASR_NUM_GRIDS = 5
CONV16_F16 = 1      # ASR_CONV16_F16: f16 activations + weights, f32 accumulate (config C5)
CONV16_BF16X3 = 2   # ASR_CONV16_BF16X3: exact three-way bf16 split, fp32-class results
CONV16_F16X2 = 3    # ASR_CONV16_F16X2: scaled two-way f16 split, fp32-class results from three MFMAs per product
PRECISIONS = {"f32": 0, "f16": CONV16_F16, "bf16x3": CONV16_BF16X3, "f16x2": CONV16_F16X2}


class AsrHipError(RuntimeError):
    pass


class OctreeFrame(ctypes.Structure):
    _fields_ = [
        ("voxel_size", ctypes.c_float * (ASR_MAX_LEVEL + 1)),
        ("inv_voxel_size", ctypes.c_float * (ASR_MAX_LEVEL + 1)),
        ("offset", ctypes.c_int32 * 3),
        ("bb_min", ctypes.c_float * 3),
        ("bb_max", ctypes.c_float * 3),
    ]


class SparseConvArgs(ctypes.Structure):
    _fields_ = [
        ("filters", ctypes.c_void_p),
        ("inp_features", ctypes.c_void_p),
        ("inp_ld", ctypes.c_int64),
        ("inp_importance", ctypes.c_void_p),
        ("neighbors_importance", ctypes.c_void_p),
        ("neighbors_index", ctypes.c_void_p),
        ("neighbors_kernel_index", ctypes.c_void_p),
        ("neighbors_row_splits", ctypes.c_void_p),
        ("num_out", ctypes.c_int64),
        ("num_inp", ctypes.c_int64),
        ("kernel_size", ctypes.c_int),
        ("cin", ctypes.c_int),
        ("cout", ctypes.c_int),
        ("normalize", ctypes.c_int),
        ("bias", ctypes.c_void_p),
        ("relu", ctypes.c_int),
        ("residual", ctypes.c_void_p),
        ("residual_ld", ctypes.c_int64),
        ("out", ctypes.c_void_p),
        ("out_ld", ctypes.c_int64),
        ("out_importance", ctypes.c_void_p),
        ("algo", ctypes.c_int),
        ("row_perm", ctypes.c_void_p),
        ("filters_b", ctypes.c_void_p),
        ("bias_b", ctypes.c_void_p),
        ("cout_b", ctypes.c_int),
        ("force_nt", ctypes.c_int),
        ("force_waves", ctypes.c_int),
        ("plan", ctypes.c_void_p),
        ("inp_absmax", ctypes.c_void_p),
        ("out_absmax", ctypes.c_void_p),
    ]


class Weight(ctypes.Structure):
    _fields_ = [
        ("name", ctypes.c_char_p),
        ("data", ctypes.c_void_p),
        ("ndim", ctypes.c_int32),
        ("shape", ctypes.c_int64 * 5),
    ]


class ImplicitParams(ctypes.Structure):
    _fields_ = [
        ("point_radius_scale", ctypes.c_float),
        ("octree_max_depth", ctypes.c_int),
        ("bb_min", ctypes.c_float * 3),
        ("bb_max", ctypes.c_float * 3),
        ("scale_sdf", ctypes.c_int),
        ("precision", ctypes.c_int),
    ]


class ImplicitSizes(ctypes.Structure):
    _fields_ = [
        ("num_points", ctypes.c_int64),
        ("num_nodes", ctypes.c_int64),
        ("num_voxels", ctypes.c_int64 * ASR_NUM_GRIDS),
        ("num_pairs", ctypes.c_int64 * ASR_NUM_GRIDS),
        ("num_agg_pairs", ctypes.c_int64),
    ]


# grouped point-to-point exchange / MAX all-reduce of asr_shard_comm (include/asr_hip.h)
SHARD_EXCHANGE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                     ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t), ctypes.c_int,
                                     ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_void_p),
                                     ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p)
SHARD_ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)
SHARD_EXCHANGE_MAX_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                         ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t), ctypes.c_int,
                                         ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_void_p),
                                         ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)


class ShardComm(ctypes.Structure):
    _fields_ = [
        ("user", ctypes.c_void_p),
        ("rank", ctypes.c_int),
        ("world", ctypes.c_int),
        ("exchange", SHARD_EXCHANGE_FN),
        ("allreduce_max_u32", SHARD_ALLREDUCE_FN),
        ("exchange_and_max", SHARD_EXCHANGE_MAX_FN),  # optional (NULL: the two above, one after the other)
    ]


class ShardStats(ctypes.Structure):
    _fields_ = [
        ("owned_rows", ctypes.c_int64 * ASR_NUM_GRIDS),
        ("halo_rows_recv", ctypes.c_int64 * ASR_NUM_GRIDS),
        ("bytes_sent", ctypes.c_int64),
        ("bytes_received", ctypes.c_int64),
        ("exchanges", ctypes.c_int64),
        ("exchange_seconds", ctypes.c_double),
    ]


# every symbol declared in include/asr_hip.h (tests/test_abi.py checks the list against the header)
EXPORTS = [
    "asr_hip_context_create", "asr_hip_context_destroy", "asr_hip_context_set_stream",
    "asr_hip_last_error", "asr_hip_version", "asr_hip_context_reserved_bytes", "asr_hip_set_print_callback", "asr_hip_print",
    "asr_hip_struct_size", "asr_hip_context_device", "asr_hip_context_weights_changed", "asr_hip_context_set_option", "asr_hip_context_get_option", "asr_hip_option_info",
    "asr_hip_sparse_conv_variant_counts", "asr_hip_sparse_conv_packed_bytes", "asr_hip_sparse_conv_pack",
    "asr_hip_sparse_conv_f16", "asr_hip_sparse_conv_bf16x3", "asr_hip_sparse_conv_f16x2",
    "asr_hip_absmax_f32", "asr_hip_convert_f16",
    "asr_hip_sparse_conv_plan_create", "asr_hip_sparse_conv_plan_destroy", "asr_hip_sparse_conv_plan_bytes", "asr_hip_context_plan_arena_reset",
    "asr_octree_frame_init", "asr_hip_point_keys", "asr_hip_octree_build", "asr_hip_octree_build_grow", "asr_hip_octree_build_parts", "asr_hip_octree_get", "asr_hip_dual_cells_count", "asr_hip_dual_cells_count_for", "asr_hip_dual_cells_fill",
    "asr_hip_contour_count", "asr_hip_contour_fill", "asr_hip_components_count", "asr_hip_components_fill",
    "asr_hip_unordered_set_order", "asr_density_inlier",
    "asr_hip_grid_neighbors_count", "asr_hip_grid_neighbors_fill", "asr_hip_grid_neighbors_rows_count", "asr_hip_grid_neighbors_rows_fill", "asr_hip_grid_coarsen_count",
    "asr_hip_grid_coarsen_fill", "asr_hip_voxel_info", "asr_hip_multi_radius_search_count",
    "asr_hip_multi_radius_search_fill", "asr_hip_knn_radius", "asr_hip_radius_neighbor_count", "asr_hip_continuous_conv_f32", "asr_hip_continuous_conv_basis_f32",
    "asr_hip_aggregation_importance", "asr_hip_sparse_conv_f32", "asr_hip_invert_neighbors_list", "asr_hip_row_groups",
    "asr_hip_reduce_subarrays_sum", "asr_hip_decode_mlp", "asr_hip_implicit_build",
    "asr_hip_implicit_network", "asr_hip_implicit_aggregate", "asr_hip_implicit_forward", "asr_hip_implicit_get",
    "asr_hip_implicit_stage_ms",
    "asr_hip_implicit_forward_sharded", "asr_hip_shard_comm_rccl_unique_id", "asr_hip_shard_comm_rccl_create",
    "asr_hip_shard_comm_rccl_destroy",
]

_lib = None


def load():
    """Loads libasr_hip.so; raises AsrHipError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AsrHipError(
                "libasr_hip.so not found at %s -- build it with __graft_entry__.build() or "
                "`make -C adaptive-surface-reconstruction_amd/csrc`; there is no CPU fallback" % LIB_PATH)
        # torch first: it ships its own HIP runtime, and a process that loads /opt/rocm's copy (through libasr_hip.so) BEFORE
        # torch's ends up with two runtimes -- torch.cuda.is_available() turns False and hipGetDeviceCount fails in the library
        # (ASR_HIP_ENODEV).  With torch loaded first the library binds to the runtime that is already there.
        import torch  # noqa: F401
        lib = ctypes.CDLL(LIB_PATH)
        lib.asr_hip_last_error.restype = ctypes.c_char_p
        lib.asr_hip_version.restype = ctypes.c_char_p
        lib.asr_hip_context_reserved_bytes.restype = ctypes.c_size_t
        lib.asr_hip_struct_size.restype = ctypes.c_size_t
        lib.asr_hip_sparse_conv_packed_bytes.restype = ctypes.c_size_t
        for name, cls in (("asr_octree_frame", OctreeFrame), ("asr_sparse_conv_args", SparseConvArgs),
                          ("asr_weight", Weight), ("asr_implicit_params", ImplicitParams),
                          ("asr_implicit_sizes", ImplicitSizes), ("asr_shard_comm", ShardComm),
                          ("asr_shard_stats", ShardStats)):
            if lib.asr_hip_struct_size(name.encode()) != ctypes.sizeof(cls):
                raise AsrHipError("ABI mismatch: struct %s has a different size in libasr_hip.so"
                                  % name)
        _lib = lib
    return _lib


PRINT_LEVELS = {"DEBUG": 0, "INFO": 1, "WARN": 2, "ERROR": 3}  # cpp/lib/asr.hpp:27
_PRINT_CB_TYPE = ctypes.CFUNCTYPE(None, ctypes.c_char_p, ctypes.c_void_p)
_print_keepalive = {}  # level -> ctypes thunk (must outlive its registration)


def set_print_callback_function(print_callback, levels):
    """asr::SetPrintCallbackFunction (cpp/lib/asr.hpp:29-34): `print_callback(str)` receives the library's messages of
    the given verbosity levels (ints 0..3 or the names DEBUG / INFO / WARN / ERROR); None removes it.  Raises
    RuntimeError("invalid verbosity level") like the reference (cpp/lib/asr.cpp:42-44).  No GPU needed."""
    lv = [PRINT_LEVELS[x] if isinstance(x, str) else int(x) for x in levels]
    thunk = None
    if print_callback is not None:
        thunk = _PRINT_CB_TYPE(lambda msg, user: print_callback(msg.decode() if msg else ""))
    arr = (ctypes.c_int * len(lv))(*lv)
    rc = load().asr_hip_set_print_callback(thunk if thunk is not None else _PRINT_CB_TYPE(), None, arr, len(lv))
    if rc != 0:
        raise RuntimeError("invalid verbosity level")
    for x in lv:
        _print_keepalive[x] = thunk


def library_print(msg, level=1):
    """asr::Print (cpp/lib/utils.h:26)"""
    f = load().asr_hip_print
    f.restype = None
    f(msg.encode(), int(level))


def ptr(t):
    """device pointer of a torch tensor (None -> NULL)"""
    if t is None:
        return ctypes.c_void_p(0)
    if not t.is_cuda:
        raise AsrHipError("expected a GPU tensor: the HIP path has no CPU fallback")
    if not t.is_contiguous():
        raise AsrHipError("expected a contiguous tensor")
    return ctypes.c_void_p(t.data_ptr())


class Context:
    """asr_hip_context bound to a torch stream."""

    def __init__(self, stream=None):
        import torch
        if not torch.cuda.is_available():
            raise AsrHipError("no GPU visible: the MI355X path cannot run (no CPU fallback)")
        self.lib = load()
        self._h = ctypes.c_void_p(0)
        s = stream if stream is not None else torch.cuda.current_stream()
        rc = self.lib.asr_hip_context_create(ctypes.byref(self._h), ctypes.c_void_p(s.cuda_stream))
        if rc != 0:
            raise AsrHipError("asr_hip_context_create failed with code %d" % rc)

    def set_stream(self, stream):
        self.lib.asr_hip_context_set_stream(self._h, ctypes.c_void_p(stream.cuda_stream))

    def check(self, rc):
        if rc != 0:
            msg = self.lib.asr_hip_last_error(self._h)
            raise AsrHipError("asr_hip error %d: %s" % (rc, msg.decode() if msg else "?"))

    def call(self, name, *args):
        self.check(getattr(self.lib, name)(self._h, *args))

    def set_option(self, name, value):
        """per-context tunable (include/asr_hip.h asr_hip_context_set_option)"""
        self.call("asr_hip_context_set_option", name.encode(), ctypes.c_int64(int(value)))

    def non_default_options(self):
        """{name: value} of every tunable of this context that differs from its built-in default (asr_hip_option_info)"""
        out, i = {}, 0
        name, dflt = ctypes.c_char_p(), ctypes.c_int64()
        while self.lib.asr_hip_option_info(i, ctypes.byref(name), ctypes.byref(dflt)) == 0:
            v = self.get_option(name.value.decode())
            if v != dflt.value:
                out[name.value.decode()] = v
            i += 1
        return out

    def get_option(self, name):
        v = ctypes.c_int64(0)
        self.call("asr_hip_context_get_option", name.encode(), ctypes.byref(v))
        return v.value

    @property
    def device_index(self):
        return int(self.lib.asr_hip_context_device(self._h))

    def sconv_variant_counts(self, reset=False):
        """{(NT, KC, IMP, WAVES, DUAL): launches} of k_sconv_mfma since the last reset; launches of the 16-bit
        kernels have a sixth field, the mode (1 = f16, 2 = bf16x3, 3 = f16x2), and a seventh: 1 = k_sconv_plan16 (plan-driven),
        0 = k_sconv_mfma16 (neighbour table in LDS)"""
        buf = ctypes.create_string_buffer(4096)
        self.call("asr_hip_sparse_conv_variant_counts", buf, ctypes.c_size_t(4096), int(bool(reset)))
        out = {}
        for item in buf.value.decode().split(";"):
            if item:
                k, c = item.split(":")
                out[tuple(int(x) for x in k.split(","))] = int(c)  # 6th field (16-bit kernels): mode
        return out

    def reserved_bytes(self):
        return int(self.lib.asr_hip_context_reserved_bytes(self._h))

    def close(self):
        if self._h:
            self.lib.asr_hip_context_destroy(self._h)
            self._h = ctypes.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def frame_init(bb_min, bb_max):
    f = OctreeFrame()
    mn = (ctypes.c_float * 3)(*[float(x) for x in bb_min])
    mx = (ctypes.c_float * 3)(*[float(x) for x in bb_max])
    rc = load().asr_octree_frame_init(ctypes.byref(f), mn, mx)
    if rc != 0:
        raise AsrHipError("asr_octree_frame_init failed (degenerate bounding box?)")
    return f

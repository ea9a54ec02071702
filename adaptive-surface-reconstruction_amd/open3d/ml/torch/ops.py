"""open3d.ml.torch.ops drop-in: registers the four torch ops the reference model calls
(models/common_torch.py:127,133; models/v0/net_definitions_torch.py:30) with the Open3D v0.14.1
schemas and a GPU implementation that calls libasr_hip.so.  CPU tensors have
no kernel here on purpose: torch raises NotImplementedError for them."""
from typing import NamedTuple

import torch

_lib = torch.library.Library("open3d", "DEF")
_lib.define(
    "invert_neighbors_list(int num_points, Tensor inp_neighbors_index, "
    "Tensor inp_neighbors_row_splits, Tensor inp_neighbors_attributes) -> "
    "(Tensor neighbors_index, Tensor neighbors_row_splits, Tensor neighbors_attributes)")
_lib.define("reduce_subarrays_sum(Tensor values, Tensor row_splits) -> Tensor")
_lib.define(
    "sparse_conv(Tensor filters, Tensor inp_features, Tensor inp_importance, "
    "Tensor neighbors_index, Tensor neighbors_kernel_index, Tensor neighbors_importance, "
    "Tensor neighbors_row_splits, bool normalize=False, int max_temp_mem_MB=64) -> Tensor")
_lib.define(
    "continuous_conv(Tensor filters, Tensor out_positions, Tensor extents, Tensor offset, "
    "Tensor inp_positions, Tensor inp_features, Tensor inp_importance, Tensor neighbors_index, "
    "Tensor neighbors_importance, Tensor neighbors_row_splits, bool align_corners=False, "
    "str coordinate_mapping=\"ball_to_cube_radial\", bool normalize=False, "
    "str interpolation=\"linear\", int max_temp_mem_MB=64) -> Tensor")


def _hip():
    from asr_hip import ops as hip_ops
    return hip_ops


def _sparse_conv_gpu(filters, inp_features, inp_importance, neighbors_index,
                     neighbors_kernel_index, neighbors_importance, neighbors_row_splits,
                     normalize=False, max_temp_mem_MB=64):
    if inp_importance.numel():
        raise RuntimeError("sparse_conv: per-point inp_importance is not supported")
    nimp = neighbors_importance.to(filters.device) if neighbors_importance.numel() else None
    return _hip().sparse_conv(filters, inp_features, neighbors_index, neighbors_kernel_index,
                              neighbors_row_splits, normalize=normalize,
                              neighbors_importance=nimp)


def _continuous_conv_gpu(filters, out_positions, extents, offset, inp_positions, inp_features,
                         inp_importance, neighbors_index, neighbors_importance,
                         neighbors_row_splits, align_corners=False,
                         coordinate_mapping="ball_to_cube_radial", normalize=False,
                         interpolation="linear", max_temp_mem_MB=64):
    if not align_corners or coordinate_mapping != "ball_to_cube_radial" or interpolation != "linear":
        raise RuntimeError("continuous_conv: only align_corners=True, ball_to_cube_radial, linear")
    if inp_importance.numel():
        raise RuntimeError("continuous_conv: per-point inp_importance is not supported")
    if offset.numel() and bool((offset != 0).any()):
        raise RuntimeError("continuous_conv: non-zero offset is not supported")
    nimp = neighbors_importance.to(filters.device) if neighbors_importance.numel() else None
    return _hip().continuous_conv(filters, out_positions, extents, inp_positions, inp_features,
                                  neighbors_index, nimp, neighbors_row_splits, normalize=normalize)


def _invert_gpu(num_points, inp_neighbors_index, inp_neighbors_row_splits,
                inp_neighbors_attributes):
    idx, rs, attr = _hip().invert_neighbors_list(num_points, inp_neighbors_index,
                                                 inp_neighbors_row_splits,
                                                 inp_neighbors_attributes)
    if inp_neighbors_attributes.numel() and attr.dtype != inp_neighbors_attributes.dtype:
        attr = attr.to(inp_neighbors_attributes.dtype)
    return idx.to(inp_neighbors_index.dtype), rs, attr


def _reduce_gpu(values, row_splits):
    return _hip().reduce_subarrays_sum(values, row_splits)


_lib.impl("sparse_conv", _sparse_conv_gpu, "CUDA")
_lib.impl("continuous_conv", _continuous_conv_gpu, "CUDA")
_lib.impl("invert_neighbors_list", _invert_gpu, "CUDA")
_lib.impl("reduce_subarrays_sum", _reduce_gpu, "CUDA")

sparse_conv = torch.ops.open3d.sparse_conv
continuous_conv = torch.ops.open3d.continuous_conv
reduce_subarrays_sum = torch.ops.open3d.reduce_subarrays_sum


class InvertNeighborsListResult(NamedTuple):
    neighbors_index: torch.Tensor
    neighbors_row_splits: torch.Tensor
    neighbors_attributes: torch.Tensor


def invert_neighbors_list(num_points: int, inp_neighbors_index: torch.Tensor,
                          inp_neighbors_row_splits: torch.Tensor,
                          inp_neighbors_attributes: torch.Tensor):
    a, b, c = torch.ops.open3d.invert_neighbors_list(num_points, inp_neighbors_index,
                                                     inp_neighbors_row_splits,
                                                     inp_neighbors_attributes)
    return InvertNeighborsListResult(a, b, c)

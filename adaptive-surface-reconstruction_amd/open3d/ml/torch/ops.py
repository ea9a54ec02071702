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

# ---- backward passes ("next" row f4): training on the MI355X ---------------------------------------------
# open3d::sparse_conv, out[q] = (1/n_q) sum_p imp_p W[k_p]^T f[idx_p]  (models/common_torch.py:133-142):
#   d f[i]  = sum over the pairs p that read input i of (imp_p / n_q(p)) W[k_p] g[q(p)]: the SAME gather-GEMM on the
#             transposed graph (inverted neighbour list, filters transposed per slot) -- run by the forward HIP kernel;
#   d W[k]  = F_k^T G_k with F_k / G_k the (importance scaled) input rows / output gradients of the pairs that use
#             slot k: 55 plain GEMMs (library GEMM through torch.matmul).
# Importance arrays come from the geometry (scale compatibility x window), not from learnable tensors: no gradient.
def _sparse_conv_setup(ctx, inputs, output):
    (filters, inp_features, _, neighbors_index, neighbors_kernel_index, neighbors_importance, neighbors_row_splits,
     normalize, _) = inputs
    ctx.save_for_backward(filters, inp_features, neighbors_index, neighbors_kernel_index, neighbors_importance,
                          neighbors_row_splits)
    ctx.normalize = bool(normalize)


def _sparse_conv_backward(ctx, grad):
    filters, f, idx, kidx, nimp, rs = ctx.saved_tensors
    K = filters.shape[0]
    v, n_in = rs.shape[0] - 1, f.shape[0]
    grad = grad.contiguous()
    has_imp = nimp.numel() > 0
    lens = rs[1:] - rs[:-1]
    row = torch.repeat_interleave(torch.arange(v, device=f.device), lens)      # output row of every pair
    if ctx.normalize:
        # the forward divides by the importance sum of the row, or by the neighbour COUNT when there is no importance
        norm = torch.ops.open3d.reduce_subarrays_sum(nimp, rs) if has_imp else lens.to(grad.dtype)
        grad = grad / torch.where(norm != 0, norm, torch.ones_like(norm))[:, None]
    g_filters = g_feats = None
    if ctx.needs_input_grad[1]:
        # transposed graph: rows = input points, entries in the order open3d::invert_neighbors_list produces
        order = torch.argsort(idx.long(), stable=True)
        inv_rs = torch.zeros(n_in + 1, dtype=torch.int64, device=f.device)
        inv_rs[1:] = torch.cumsum(torch.bincount(idx.long(), minlength=n_in), 0)
        g_feats = _hip().sparse_conv(filters.transpose(1, 2).contiguous(), grad, row[order].to(torch.int32),
                                     kidx[order], inv_rs, neighbors_importance=nimp[order] if has_imp else None)
    if ctx.needs_input_grad[0]:
        g_filters = torch.zeros_like(filters)
        src = f.index_select(0, idx.long())
        if has_imp:
            src = src * nimp[:, None]
        slot = kidx.long()
        for k in torch.unique(slot).tolist():
            sel = torch.nonzero(slot == k).reshape(-1)
            g_filters[k] = src.index_select(0, sel).t() @ grad.index_select(0, row.index_select(0, sel))
    return g_filters, g_feats, None, None, None, None, None, None, None


def _reduce_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[1])


def _reduce_backward(ctx, grad):
    (rs,) = ctx.saved_tensors
    return torch.repeat_interleave(grad, rs[1:] - rs[:-1]), None


# open3d::continuous_conv: gradient with respect to the filters (the aggregation block's inputs are data: normals and a
# constant, net_definitions_torch.py:640-653).  dW[cell][c][o] = sum_v B[v][cell][c] g[v][o] / norm[v] with the
# per-voxel interpolation matrices B from the HIP kernel (asr_hip_continuous_conv_basis_f32) and one library GEMM.
def _cconv_setup(ctx, inputs, output):
    (filters, out_positions, extents, _, inp_positions, inp_features, _, neighbors_index, neighbors_importance,
     neighbors_row_splits, _, _, normalize, _, _) = inputs
    ctx.save_for_backward(filters, out_positions, extents, inp_positions, inp_features, neighbors_index,
                          neighbors_importance, neighbors_row_splits)
    ctx.normalize = bool(normalize)


def _cconv_backward(ctx, grad):
    filters, out_positions, extents, inp_positions, inp_features, nidx, nimp, rs = ctx.saved_tensors
    if ctx.needs_input_grad[5] or ctx.needs_input_grad[4] or ctx.needs_input_grad[1]:
        raise RuntimeError("continuous_conv: only the gradient with respect to the filters is implemented")
    g_filters = None
    if ctx.needs_input_grad[0]:
        basis, norm = _hip().continuous_conv_basis(out_positions, extents, inp_positions, inp_features, nidx,
                                                   nimp if nimp.numel() else None, rs)
        g = grad.contiguous()
        if ctx.normalize:
            g = g / torch.where(norm != 0, norm, torch.ones_like(norm))[:, None]
        g_filters = (basis.t() @ g).reshape(filters.shape)
    return (g_filters,) + (None,) * 14


torch.library.register_autograd("open3d::continuous_conv", _cconv_backward, setup_context=_cconv_setup)
torch.library.register_autograd("open3d::sparse_conv", _sparse_conv_backward, setup_context=_sparse_conv_setup)
torch.library.register_autograd("open3d::reduce_subarrays_sum", _reduce_backward, setup_context=_reduce_setup)

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

"""open3d.ml.torch.layers.ContinuousConv as used by CConvAggregationBlock
(models/v0/net_definitions_torch.py:53-70,108-116): parameters `kernel` [*kernel_size, Cin, Cout]
and `bias`, forward with user supplied neighbours."""
import torch

from . import ops


class ContinuousConv(torch.nn.Module):

    def __init__(self, in_channels, filters, kernel_size, activation=None, use_bias=True,
                 kernel_initializer=lambda x: torch.nn.init.uniform_(x, -0.05, 0.05),
                 bias_initializer=torch.nn.init.zeros_, align_corners=True,
                 coordinate_mapping='ball_to_cube_radial', interpolation='linear',
                 normalize=True, radius_search_ignore_query_points=False,
                 radius_search_metric='L2', offset=None, window_function=None,
                 use_dense_layer_for_center=False, **kwargs):
        super().__init__()
        if window_function is not None or use_dense_layer_for_center:
            raise RuntimeError("ContinuousConv: window_function / dense centre layer unsupported")
        self.in_channels = in_channels
        self.filters = filters
        self.kernel_size = list(kernel_size)
        self.activation = activation if activation is not None else (lambda x: x)
        self.use_bias = use_bias
        self.align_corners = align_corners
        self.coordinate_mapping = coordinate_mapping
        self.interpolation = interpolation
        self.normalize = normalize
        self.register_buffer('offset', torch.zeros(3, dtype=torch.float32)
                             if offset is None else torch.as_tensor(offset, dtype=torch.float32))
        self.kernel = torch.nn.Parameter(torch.empty(*self.kernel_size, in_channels, filters))
        kernel_initializer(self.kernel)
        if use_bias:
            self.bias = torch.nn.Parameter(torch.empty(filters))
            bias_initializer(self.bias)

    def forward(self, inp_features, inp_positions, out_positions, extents, inp_importance=None,
                fixed_radius_search_hash_table=None, user_neighbors_index=None,
                user_neighbors_row_splits=None, user_neighbors_importance=None):
        if user_neighbors_index is None or user_neighbors_row_splits is None:
            raise RuntimeError("ContinuousConv: the built-in radius search is not part of the hot "
                               "path; pass user_neighbors_index / user_neighbors_row_splits")
        empty = torch.empty((0,), dtype=torch.float32, device=inp_features.device)
        if not isinstance(extents, torch.Tensor):
            extents = torch.tensor(extents, dtype=torch.float32, device=inp_features.device)
        nimp = user_neighbors_importance if user_neighbors_importance is not None else empty
        out = ops.continuous_conv(
            filters=self.kernel, out_positions=out_positions, extents=extents.reshape(-1),
            offset=self.offset.to(inp_features.device), inp_positions=inp_positions,
            inp_features=inp_features,
            inp_importance=inp_importance if inp_importance is not None else empty,
            neighbors_index=user_neighbors_index, neighbors_importance=nimp,
            neighbors_row_splits=user_neighbors_row_splits, align_corners=self.align_corners,
            coordinate_mapping=self.coordinate_mapping, normalize=self.normalize,
            interpolation=self.interpolation)
        if self.use_bias:
            out = out + self.bias
        return self.activation(out)

"""Arithmetic backend for asr_hip.sharding.ShardedNetwork built on the CPU oracle: lets the partition /
halo-exchange logic run under gloo on CPU tensors.  TEST INFRASTRUCTURE (the product backend is
asr_hip.sharding.HipBackend)."""
import numpy as np
import torch

from oracle import oracle as O


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


class OracleBackend:
    def _sub_csr(self, csr, rows):
        idx, kidx, rs = (_np(a) for a in csr)
        rows = _np(rows).astype(np.int64)
        lens = rs[rows + 1] - rs[rows]
        sub_rs = np.zeros(len(rows) + 1, np.int64)
        sub_rs[1:] = np.cumsum(lens)
        take = np.concatenate([np.arange(rs[r], rs[r + 1]) for r in rows]) if len(rows) else np.zeros(0, np.int64)
        return idx[take], kidx[take], sub_rs, rows

    def sparse_conv(self, kernel, bias, x, csr, rows, v_out, imp=None, normalize=False, residual=None):
        idx, kidx, sub_rs, rows = self._sub_csr(csr, rows)
        W, b, f = _np(kernel), _np(bias), _np(x)
        nimp = _np(imp)[idx.astype(np.int64)] if imp is not None else None   # common_torch.py:124-126
        out = O.sparse_conv(W, f, idx, kidx, nimp, sub_rs, normalize)
        out = np.maximum(out + b, 0)
        if residual is not None:
            out = out + _np(residual)[rows]
        full = torch.zeros((v_out, W.shape[2]), dtype=torch.float32)
        full[torch.from_numpy(rows)] = torch.from_numpy(out.astype(np.float32))
        if imp is None:
            return full
        oimp = torch.zeros(v_out, dtype=torch.float32)
        oimp[torch.from_numpy(rows)] = torch.from_numpy(O.reduce_subarrays_sum(nimp, sub_rs))
        return full, oimp

    def sparse_conv_ab(self, ka, ba, kb, bb, x, csr, rows, v_out, imp):
        a = self.sparse_conv(ka, ba, x, csr, rows, v_out)
        b, oimp = self.sparse_conv(kb, bb, x, csr, rows, v_out, imp, True)
        return torch.cat([a, b], 1), oimp

    def aggregate(self, points, normals, radii, bb, centers, sizes, kernel, bias):
        pts, nrm, rad = _np(points), _np(normals), _np(radii)
        o = O.Oracle()
        o.build_octree(np.zeros((0, 3), np.float32), np.zeros(0, np.float32), bb[0], bb[1])  # frame only
        idx, dist, rs, compat = o.radius_search(pts, rad, _np(centers), _np(sizes))
        imp = (compat * O.window_poly6(dist)).astype(np.float32)
        if kernel is None:
            return None, torch.from_numpy(imp)
        feats = np.concatenate([nrm, np.ones((len(pts), 1), np.float32)], 1)
        out = O.continuous_conv(_np(kernel), _np(centers), _np(sizes), pts, feats, idx, imp, rs, True)
        out = np.maximum(out + _np(bias), 0).astype(np.float32)
        return torch.from_numpy(out), torch.from_numpy(imp)

    def decode(self, code, w, sizes):
        v = O.decode(_np(code), _np(w["dense_decoder1.weight"]), _np(w["dense_decoder1.bias"]),
                     _np(w["dense_decoder2.weight"]), _np(w["dense_decoder2.bias"]),
                     _np(w["dense_decoder3.weight"]), _np(sizes) if sizes is not None else None)
        return torch.from_numpy(v)


def geometry_from_oracle(item):
    """parity.oracle_geometry dict -> torch tensors (uint64 keys as int64 bit patterns) + inverted up lists"""
    g = {}
    for k, v in item.items():
        if k == "nodes" or k.startswith("aggregation"):
            continue
        a = np.ascontiguousarray(v)
        if a.dtype == np.uint64:
            a = a.view(np.int64)
        g[k] = torch.from_numpy(a)
    for i in range(4):
        n_coarse = len(item["voxel_sizes%d" % (i + 1)])
        idx, rs, attr = O.invert_neighbors_list(n_coarse, item["up_neighbors_index%d" % i],
                                                item["up_neighbors_row_splits%d" % i],
                                                item["up_neighbors_kernel_index%d" % i])
        g["down_neighbors_index%d" % i] = torch.from_numpy(idx)
        g["down_neighbors_row_splits%d" % i] = torch.from_numpy(rs)
        g["down_neighbors_kernel_index%d" % i] = torch.from_numpy(attr)
    return g

"""Generates tests/golden/scale_compat.npz from the reference's own numpy implementation
(models/common.py:18-44) -- run in the build container only."""
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference")
from models.common import compute_scale_compatibility  # noqa: E402

rng = np.random.default_rng(7)
v, n = 300, 1000
lens = rng.integers(0, 12, size=v)
lens[5] = 0
rs = np.zeros(v + 1, np.int64)
rs[1:] = np.cumsum(lens)
idx = rng.integers(0, n, size=rs[-1]).astype(np.int32)
# voxel sizes are edge * 2^-l; radii log-uniform over two decades
query_scale = (np.float32(2.2) * np.float32(2.0) ** (-rng.integers(2, 9, size=v))).astype(np.float32)
radii = np.exp(rng.uniform(np.log(1e-3), np.log(0.3), size=n)).astype(np.float32)
compat = compute_scale_compatibility(query_scale, 2 * radii, idx, rs)
assert compat.dtype == np.float32
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scale_compat.npz")
np.savez_compressed(out, query_scale=query_scale, radii=radii, idx=idx, rs=rs, compat=compat)
print("wrote", out, compat.shape, compat.min(), compat.max())

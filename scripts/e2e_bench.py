"""End-to-end adaptivesurfacereconstruction.reconstruct_surface on a synthetic scan (host arrays in,
mesh out), with the analytic-weights caveat: random weights give an arbitrary surface."""
import sys
import time

import numpy as np
import torch

sys.path[:0] = ["adaptive-surface-reconstruction_amd"]
import adaptivesurfacereconstruction as asr  # noqa: E402
from asr_hip import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
p, q = synth.scan_cloud(n, seed=3, device="cuda:0")
pts, nrm = p.cpu().numpy(), q.cpu().numpy()
w = synth.make_weights(1, seed=0)
for rep in range(2):
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = asr.reconstruct_surface(pts, nrm, weights=w, keep_n_connected_components=8)
    torch.cuda.synchronize()
    print("reconstruct_surface(%d points, radii estimated): %.3f s -> %d vertices, %d triangles" %
          (n, time.perf_counter() - t, out["vertices"].shape[0], out["triangles"].shape[0]))

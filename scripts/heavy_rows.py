"""Lengths of the aggregation search's heavy rows (more than 128 hits) on the bench clouds: how many, how long.
usage: python scripts/heavy_rows.py [points]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "adaptive-surface-reconstruction_amd")]
import torch
from asr_hip import synth
from asr_hip.pipeline import ImplicitPipeline
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda:0")
for dv in (1.0, 10.0):
    pts, nrm = synth.scan_cloud(n, seed=1000, device=dev, density_variance=dv)
    radii = synth.knn_radii_gpu(pts, 24)
    bb = synth.bounding_box(pts, 0.1)
    pipe = ImplicitPipeline(synth.make_weights(4, seed=0), device=dev)
    pipe.build(pts, radii, bb[0], bb[1])
    rs = pipe.get("aggregation_row_splits")
    lens = rs[1:] - rs[:-1]
    h = lens[lens > 128].sort().values
    print("density variance %g: rows %d, pairs %d, heavy rows %d with %d pairs; longest %s; rows > 4096: %d, > 8192: %d, > 16384: %d"
          % (dv, lens.numel(), int(lens.sum()), h.numel(), int(h.sum()), h[-8:].tolist(), int((h > 4096).sum()), int((h > 8192).sum()),
             int((h > 16384).sum())), flush=True)
    del pipe

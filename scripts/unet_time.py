#!/usr/bin/env python3
"""U-Net stage time of the bench cloud on one build of the library: `--lib path` loads another libasr_hip.so (A/B runs of two
builds inside ONE gpurun call: boxes differ by a few tenths of a millisecond).  Prints per-level sums when --levels."""
import argparse
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "adaptive-surface-reconstruction_amd"), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--points", type=int, default=10_000_000)
ap.add_argument("--precision", default="bf16x3")
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--opt", action="append", default=[], help="name=value context options")
args = ap.parse_args()
from asr_hip import _lib
if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
from asr_hip import synth
from asr_hip.pipeline import ImplicitPipeline
dev = torch.device("cuda:0")
pts, nrm = synth.scan_cloud(args.points, seed=1000, device=dev)
radii = synth.knn_radii_gpu(pts, 24)
bb = synth.bounding_box(pts, 0.1)
pipe = ImplicitPipeline(synth.make_weights(1, seed=2), device=dev, precision=args.precision)
for o in args.opt:
    k, v = o.split("=")
    pipe.ctx.set_option(k, int(v))
ms, geo, cc = [], [], []
for i in range(2 + args.steps):
    pipe.forward(pts, nrm, radii, bb[0], bb[1])
    st = pipe.stage_ms()
    if i >= 2:
        ms.append(st["unet"])
        geo.append(st["geometry_wall"])
        cc.append(st["continuous_conv"])
ms.sort()
print("%-40s unet min %.3f med %.3f | geometry med %.3f cconv med %.3f" % (args.lib or "default", ms[0], ms[len(ms) // 2],
                                                                          sorted(geo)[len(geo) // 2], sorted(cc)[len(cc) // 2]))

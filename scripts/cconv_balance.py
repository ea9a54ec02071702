"""Load balance of k_cconv_mfma's static distribution: super-groups of 64 voxels go to the 2 048 waves round robin; cost model =
pairs of the rows the kernel accumulates (<= 256 pairs) + a constant per voxel.  usage: python scripts/cconv_balance.py"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "adaptive-surface-reconstruction_amd")]
import torch
from asr_hip import synth
from asr_hip.pipeline import ImplicitPipeline
dev = torch.device("cuda:0")
pts, nrm = synth.scan_cloud(10_000_000, seed=1000, device=dev)
radii = synth.knn_radii_gpu(pts, 24)
bb = synth.bounding_box(pts, 0.1)
pipe = ImplicitPipeline(synth.make_weights(4, seed=0), device=dev)
pipe.build(pts, radii, bb[0], bb[1])
rs = pipe.get("aggregation_row_splits")
cnt = (rs[1:] - rs[:-1]).float()
cnt = torch.where(cnt > 256, torch.zeros_like(cnt), cnt)
v = cnt.numel()
pad = (-v) % 64
c = torch.cat([cnt, cnt.new_zeros(pad)]).reshape(-1, 64)
for per_voxel in (0.0, 8.0, 20.0):
    cost = c.sum(1) + per_voxel * 64
    nsg = cost.numel()
    nw = 2048
    padw = (-nsg) % nw
    w = torch.cat([cost, cost.new_zeros(padw)]).reshape(-1, nw).sum(0)
    print("per-voxel %4.0f: super-groups %d, cost mean %.0f p99 %.0f max %.0f | per wave mean %.0f max %.0f (max/mean %.2f)" % (
        per_voxel, nsg, cost.mean().item(), torch.quantile(cost, 0.99).item(), cost.max().item(), w.mean().item(), w.max().item(),
        (w.max() / w.mean()).item()))
first = c.sum(1)[:2000]
print("pairs of the first super-groups:", [int(x) for x in first[:12].tolist()], "... of the last:", [int(x) for x in c.sum(1)[-6:].tolist()])
big = (c.sum(1) > 4 * c.sum(1).mean()).nonzero().flatten()
print("super-groups above 4x the mean: %d, indices %d .. %d" % (big.numel(), int(big.min()) if big.numel() else -1, int(big.max()) if big.numel() else -1))

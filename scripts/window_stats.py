"""How local are the gathers of a 128-row tile of the regrouped order?  For every tile: the natural-index span of its
rows, and the share of its (row, slot) pairs whose input row lies inside a window of W consecutive input rows placed
around the tile's rows -- what an LDS-staged window would serve.  usage: python scripts/window_stats.py [points]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "adaptive-surface-reconstruction_amd")]
import torch
from asr_hip import synth
from asr_hip.pipeline import ImplicitPipeline
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda:0")
pts, nrm = synth.scan_cloud(n, seed=1000, device=dev)
radii = synth.knn_radii_gpu(pts, 24)
bb = synth.bounding_box(pts, 0.1)
pipe = ImplicitPipeline(synth.make_weights(4, seed=0), device=dev)
pipe.build(pts, radii, bb[0], bb[1])
for lvl in range(3):
    rs = pipe.get("neighbors_row_splits%d" % lvl)
    idx = pipe.get("neighbors_index%d" % lvl).long()
    kidx = pipe.get("neighbors_kernel_index%d" % lvl).long()
    perm = pipe.get("tiling%d" % lvl).long()
    v = rs.numel() - 1
    rows = torch.repeat_interleave(torch.arange(v, device=dev), rs[1:] - rs[:-1])
    pos = torch.empty(v, dtype=torch.long, device=dev)
    pos[perm] = torch.arange(v, device=dev)          # position of a row in the tiling order
    tile = pos // 128
    nt = (v + 127) // 128
    big = torch.iinfo(torch.long).max
    tmin = torch.full((nt,), big, dtype=torch.long, device=dev).scatter_reduce(0, tile, torch.arange(v, device=dev), "amin")
    tmax = torch.zeros(nt, dtype=torch.long, device=dev).scatter_reduce(0, tile, torch.arange(v, device=dev), "amax")
    span = (tmax - tmin + 1).float()
    ptile = tile[rows]
    print("level %d: rows %d pairs %d tiles %d | row span of a tile: p25 %.0f p50 %.0f p75 %.0f p90 %.0f" % (
        lvl, v, idx.numel(), nt, *torch.quantile(span[:-1], torch.tensor([.25, .5, .75, .9], device=dev)).tolist()))
    for W in (256, 384, 512, 1024):
        # window of W input rows centred on the tile's row span
        mid = (tmin + tmax) // 2
        lo = (mid - W // 2)[ptile]
        inside = (idx >= lo) & (idx < lo + W)
        same = kidx < 7
        # tiles the window pays for: at least 70 % of their pairs inside
        per_tile_in = torch.zeros(nt, device=dev).scatter_add_(0, ptile, inside.float())
        per_tile_all = torch.zeros(nt, device=dev).scatter_add_(0, ptile, torch.ones_like(inside, dtype=torch.float))
        good = per_tile_in >= 0.7 * per_tile_all
        print("   W %4d: pairs inside %.3f (same-level pairs inside %.3f, cross-level %.3f) | tiles with >= 70%% inside: %.3f of the "
              "tiles holding %.3f of the pairs, %.3f of their pairs inside" % (
                  W, inside.float().mean().item(), inside[same].float().mean().item(), inside[~same].float().mean().item(),
                  good.float().mean().item(), (per_tile_all[good].sum() / per_tile_all.sum()).item(),
                  (per_tile_in[good].sum() / per_tile_all[good].sum()).item()), flush=True)

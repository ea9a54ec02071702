"""Step / wave activity statistics of the regrouped sparse-conv row tiles of the 10 M-point bench cloud: how
many (slot) steps a 128-row tile takes, how many of its 8 waves (16 rows each) hold the slot, and how full the
active waves are.  usage: python scripts/tile_stats.py [points]"""
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
for lvl in range(5):
    rs = pipe.get("neighbors_row_splits%d" % lvl)
    kidx = pipe.get("neighbors_kernel_index%d" % lvl).long()
    perm = pipe.get("tiling%d" % lvl).long()
    v = rs.numel() - 1
    rows = torch.repeat_interleave(torch.arange(v, device=dev), rs[1:] - rs[:-1])
    mask = torch.zeros(v, dtype=torch.int64, device=dev)
    mask.scatter_add_(0, rows, torch.ones_like(kidx) << kidx)  # each slot at most once per row
    m = mask[perm]  # tile order
    for tm in (128, 256):
        pad = (-v) % tm
        mp = torch.cat([m, m.new_zeros(pad)])
        def orr(x, g):  # bitwise or over groups of g consecutive rows
            x = x.reshape(-1, g)
            out = x[:, 0].clone()
            for j in range(1, g):
                out |= x[:, j]
            return out
        def popc(x):
            c = torch.zeros_like(x)
            for b in range(56):
                c += (x >> b) & 1
            return c
        w16 = orr(mp, 16)
        w32 = orr(mp, 32)
        tile = orr(mp, tm)
        steps = popc(tile).sum().item()
        act16 = popc(w16).sum().item()
        act32 = popc(w32).sum().item()
        nnz = kidx.numel()
        print("level %d rows %d nnz/row %.1f | TM %d: tiles %d slots/tile %.1f  wave16 activity %.2f fill %.2f | "
              "wave32 activity %.2f fill %.2f" % (
                  lvl, v, nnz / v, tm, tile.numel(), steps / tile.numel(), act16 / (steps * tm / 16),
                  nnz / (act16 * 16), act32 / (steps * tm / 32), nnz / (act32 * 32)), flush=True)

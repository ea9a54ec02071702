"""Serial depth of the sparse-conv row tiles of the 10 M-point bench cloud: the slot steps of a tile run one after
the other inside ONE block, so a launch can not end before its longest tile has.  Prints, per grid level and tile
height, the distribution of slot steps per tile and the share of the steps that sit in slot ranges (the fixed
split points a split-K launch would use).  usage: python scripts/tile_chain_stats.py [points]"""
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


def orr(x, g):
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


RANGES = [(0, 7), (7, 19), (19, 31), (31, 43), (43, 55)]
for lvl in range(5):
    rs = pipe.get("neighbors_row_splits%d" % lvl)
    kidx = pipe.get("neighbors_kernel_index%d" % lvl).long()
    perm = pipe.get("tiling%d" % lvl).long()
    v = rs.numel() - 1
    rows = torch.repeat_interleave(torch.arange(v, device=dev), rs[1:] - rs[:-1])
    mask = torch.zeros(v, dtype=torch.int64, device=dev)
    mask.scatter_add_(0, rows, torch.ones_like(kidx) << kidx)
    m = mask[perm]
    hist = torch.bincount(kidx, minlength=55).tolist()
    print("level %d rows %d pairs %d  pairs per slot range %s" % (
        lvl, v, kidx.numel(), [sum(hist[a:b]) for a, b in RANGES]), flush=True)
    for tm in (64, 128):
        pad = (-v) % tm
        mp = torch.cat([m, m.new_zeros(pad)])
        tile = orr(mp, tm)
        w16 = orr(mp, 16).reshape(-1, tm // 16)
        steps = popc(tile).float()
        act = popc(w16).float().sum(1)  # wave-steps with MFMA work
        q = torch.quantile(steps, torch.tensor([0.5, 0.9, 0.99], device=dev)).tolist()
        per_range = [popc(tile & (((1 << b) - 1) ^ ((1 << a) - 1))).float() for a, b in RANGES]
        print("  TM %3d: tiles %6d  steps/tile mean %.1f p50 %.0f p90 %.0f p99 %.0f max %.0f | sum %.0f  "
              "sum/(256 CUs) %.0f | active wave-steps / (steps * waves) %.2f" % (
                  tm, tile.numel(), steps.mean().item(), q[0], q[1], q[2], steps.max().item(), steps.sum().item(),
                  steps.sum().item() / 256, (act.sum() / (steps.sum() * (tm // 16))).item()), flush=True)
        print("          per slot range: max steps %s  mean %s" % (
            [int(p.max().item()) for p in per_range], ["%.1f" % p.mean().item() for p in per_range]), flush=True)

import os, sys
REPO = "/root/repo"
sys.path[:0] = [REPO, os.path.join(REPO, "adaptive-surface-reconstruction_amd")]
import torch
from asr_hip import synth
from asr_hip.pipeline import ImplicitPipeline
dev = torch.device("cuda:0")
for dv in (1.0, 10.0):
    pts, nrm = synth.scan_cloud(10_000_000, seed=1000, device=dev, density_variance=dv)
    radii = synth.knn_radii_gpu(pts, 24)
    bb = synth.bounding_box(pts, 0.1)
    pipe = ImplicitPipeline(synth.make_weights(4, seed=0), device=dev)
    pipe.build(pts, radii, bb[0], bb[1])
    rs = pipe.get("aggregation_row_splits")
    cnt = rs[1:] - rs[:-1]
    keys = pipe.get("voxel_keys0")
    lev = torch.zeros_like(keys)
    for l in range(1, 21):
        lev += (keys >= (1 << (3 * l))).to(keys.dtype)
    lev = torch.where(keys < 0, torch.full_like(keys, 21), lev)
    print("density variance", dv, "V0", keys.numel(), "pairs", int(rs[-1]))
    for l in sorted(set(lev.tolist())):
        m = lev == l
        c = cnt[m]
        print("  level %2d rows %8d  mean %.1f  max %6d  >128: %6d  >64: %7d  >256: %5d" % (l, int(m.sum()), float(c.float().mean()), int(c.max()), int((c > 128).sum()), int((c > 64).sum()), int((c > 256).sum())))
    del pipe

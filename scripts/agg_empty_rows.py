import sys, os
sys.path[:0] = ["/root/repo", "/root/repo/adaptive-surface-reconstruction_amd"]
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
ln = rs[1:] - rs[:-1]
print("rows", ln.numel(), "empty", float((ln == 0).float().mean()), "mean", float(ln.float().mean()), "<=4", float((ln <= 4).float().mean()),
      "<=64", float((ln <= 64).float().mean()), ">128", int((ln > 128).sum()))
keys = pipe.get("voxel_keys0")
lev = (torch.log2(keys.double()).floor().long()) // 3
for l in torch.unique(lev).tolist():
    m = lev == l
    print("level", l, "rows", int(m.sum()), "empty", round(float((ln[m] == 0).float().mean()), 3), "mean pairs", round(float(ln[m].float().mean()), 1))

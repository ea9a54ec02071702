import sys
import torch
sys.path[:0] = ["adaptive-surface-reconstruction_amd"]
from asr_hip import synth
from asr_hip.pipeline import ImplicitPipeline
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda:0")
pts, nrm = synth.scan_cloud(n, seed=1000, device=dev)
radii = synth.knn_radii_gpu(pts, 24)
bb = synth.bounding_box(pts, 0.1)
pipe = ImplicitPipeline(synth.make_weights(4, seed=0), device=dev)
pipe.build(pts, radii, bb[0], bb[1])
rs = pipe.get("aggregation_row_splits")
ln = rs[1:] - rs[:-1]
tot = int(ln.sum())
print("rows", ln.numel(), "pairs", tot, "max", int(ln.max()), "mean %.1f" % (tot / ln.numel()))
for t in (16, 32, 64, 128, 256, 512, 1024, 4096):
    m = ln > t
    print("rows > %4d: %8d (%.2f%%)  pairs in them %.1f%%" % (t, int(m.sum()), 100.0 * int(m.sum()) / ln.numel(), 100.0 * int(ln[m].sum()) / tot))

"""levels 3-4 of the 10 M-point bench cloud: time of one 256 x 256 f16x2 layer when only the first K slots of the
55-slot list are convolved (the serial depth of the longest tile shrinks with K, the total work with the pair share).
usage: python scripts/small_level_probe.py"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "adaptive-surface-reconstruction_amd")]
import torch
from asr_hip import synth, ops
from asr_hip.pipeline import ImplicitPipeline
dev = torch.device("cuda:0")
pts, nrm = synth.scan_cloud(10_000_000, seed=1000, device=dev)
radii = synth.knn_radii_gpu(pts, 24)
bb = synth.bounding_box(pts, 0.1)
pipe = ImplicitPipeline(synth.make_weights(4, seed=0), device=dev)
pipe.build(pts, radii, bb[0], bb[1])
g = torch.Generator(device=dev).manual_seed(1)
for lvl, cin, cout in ((4, 256, 256), (3, 256, 256), (2, 256, 256)):
    rs = pipe.get("neighbors_row_splits%d" % lvl)
    idx = pipe.get("neighbors_index%d" % lvl)
    kidx = pipe.get("neighbors_kernel_index%d" % lvl)
    perm = pipe.get("tiling%d" % lvl)
    v = rs.numel() - 1
    f = torch.randn((v, cin), generator=g, device=dev)
    plan = ops.ConvPlan(55, idx, kidx, rs, row_perm=perm)
    out = torch.empty((v, cout), device=dev)
    hist = torch.bincount(kidx.long(), minlength=55).cumsum(0).tolist()
    for K in (55, 43, 31, 19, 7, 1):
        W = torch.randn((K, cin, cout), generator=g, device=dev) * 0.02
        pk = ops.pack_filters(W, "f16x2")
        for _ in range(3):
            ops.sparse_conv16("f16x2", pk, K, cin, cout, f, idx, kidx, rs, row_perm=perm, plan=plan, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.sparse_conv16("f16x2", pk, K, cin, cout, f, idx, kidx, rs, row_perm=perm, plan=plan, out=out)
        e1.record()
        torch.cuda.synchronize()
        print("level %d rows %7d %dx%d K %2d (pairs %8d): %.1f us" % (lvl, v, cin, cout, K, hist[K - 1],
                                                                     e0.elapsed_time(e1) * 50), flush=True)

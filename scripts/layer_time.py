"""times single sparse-conv layers of the 10 M-point bench cloud's grids through the operator API (plan reused):
usage: python scripts/layer_time.py"""
import os, sys, time
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
for lvl, cin, cout in ((4, 256, 256), (3, 256, 256), (2, 256, 256), (1, 128, 128), (0, 64, 64)):
    rs = pipe.get("neighbors_row_splits%d" % lvl)
    idx = pipe.get("neighbors_index%d" % lvl)
    kidx = pipe.get("neighbors_kernel_index%d" % lvl)
    perm = pipe.get("tiling%d" % lvl)
    v = rs.numel() - 1
    f = torch.randn((v, cin), generator=g, device=dev)
    W = torch.randn((55, cin, cout), generator=g, device=dev) * 0.02
    pk = ops.pack_filters(W, "bf16x3")
    plan = ops.ConvPlan(55, idx, kidx, rs, row_perm=perm)
    out = torch.empty((v, cout), device=dev)
    for _ in range(3):
        ops.sparse_conv16("bf16x3", pk, 55, cin, cout, f, idx, kidx, rs, row_perm=perm, plan=plan, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.sparse_conv16("bf16x3", pk, 55, cin, cout, f, idx, kidx, rs, row_perm=perm, plan=plan, out=out)
    e1.record()
    torch.cuda.synchronize()
    print("level %d rows %8d %dx%d: %.1f us" % (lvl, v, cin, cout, e0.elapsed_time(e1) * 100), flush=True)

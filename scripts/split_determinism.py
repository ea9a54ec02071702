"""Operator-level repeatability of the slot-range split: the same layer 20 times on the level-2/3 grids of a 1 M-point cloud and on
level 4 of the 10 M-point cloud, f16x2 and bf16x3 -- every run must return the same bits, and stay within 1e-4 of the unsplit
kernel (another summation order).  usage: python scripts/split_determinism.py"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "adaptive-surface-reconstruction_amd")]
import torch
from asr_hip import synth, ops
from asr_hip.pipeline import ImplicitPipeline
dev = torch.device("cuda:0")
ctx = ops.context(dev)
g = torch.Generator(device=dev).manual_seed(1)
for n, levels in ((1_000_000, ((2, 256, 256), (3, 512, 256), (3, 256, 256))), (10_000_000, ((4, 256, 256), (4, 512, 256)))):
    pts, nrm = synth.scan_cloud(n, seed=5, device=dev)
    radii = synth.knn_radii_gpu(pts, 24)
    bb = synth.bounding_box(pts, 0.1)
    pipe = ImplicitPipeline(synth.make_weights(4, seed=0), device=dev)
    pipe.build(pts, radii, bb[0], bb[1])
    for lvl, cin, cout in levels:
        rs = pipe.get("neighbors_row_splits%d" % lvl); idx = pipe.get("neighbors_index%d" % lvl)
        kidx = pipe.get("neighbors_kernel_index%d" % lvl); perm = pipe.get("tiling%d" % lvl)
        v = rs.numel() - 1
        f = torch.randn((v, cin), generator=g, device=dev)
        W = torch.randn((55, cin, cout), generator=g, device=dev) * 0.02
        b = torch.randn((cout,), generator=g, device=dev) * 0.1
        plan = ops.ConvPlan(55, idx, kidx, rs, row_perm=perm)
        for mode in ("bf16x3", "f16x2"):
            pk = ops.pack_filters(W, mode)
            ctx.set_option("sconv_split_rows", 0)
            ref = ops.sparse_conv16(mode, pk, 55, cin, cout, f, idx, kidx, rs, row_perm=perm, plan=plan, bias=b, relu=True).clone()
            ctx.set_option("sconv_split_rows", 65536)
            ctx.sconv_variant_counts(reset=True)
            first = ops.sparse_conv16(mode, pk, 55, cin, cout, f, idx, kidx, rs, row_perm=perm, plan=plan, bias=b, relu=True).clone()
            keys = list(ctx.sconv_variant_counts())
            same = all(bool((ops.sparse_conv16(mode, pk, 55, cin, cout, f, idx, kidx, rs, row_perm=perm, plan=plan, bias=b,
                                               relu=True) == first).all()) for _ in range(20))
            print("%d points level %d rows %d %dx%d %s: split launched %s, 20 repeats identical %s, |split - unsplit| %.2e" % (
                n, lvl, v, cin, cout, mode, len(keys[0]) == 8, same, (first - ref).abs().max().item()), flush=True)
            assert same and (first - ref).abs().max().item() < 1e-4
    del pipe
ctx.set_option("sconv_split_rows", 32768)

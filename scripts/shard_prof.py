import os, sys, time
sys.path[:0] = ["/root/repo", "/root/repo/adaptive-surface-reconstruction_amd"]
import torch
from asr_hip import synth, sharding
from asr_hip.sharding import ShardedImplicitPipeline
dev = torch.device("cuda:0")
n = 10_000_000
pts, nrm = synth.scan_cloud(n, seed=1000, device=dev)
radii = synth.knn_radii_gpu(pts, 24)
bb = synth.bounding_box(pts, 0.1)
w = synth.make_weights(1, seed=0, init="reference")
sp = ShardedImplicitPipeline(w, dev, precision="f16x2")
for _ in range(2): sp.forward(pts, nrm, radii, bb[0], bb[1])
torch.cuda.synchronize()
def T():
    torch.cuda.synchronize(); return time.perf_counter()
pipe = sp.pipe
from asr_hip import _lib
for rep in range(2):
    t0 = T()
    pipe.ctx.set_option("build_search", 1)
    pipe.build(pts, radii, bb[0], bb[1]); t1 = T()
    f1, imp = pipe.aggregate(pts, nrm, bb[0], bb[1]); t2 = T()
    geom = sharding.geometry_from_pipeline(pipe); t3 = T()
    sp.backend.new_geometry()
    net = sharding.ShardedNetwork(sp.backend, geom, pipe._weights, 0, 1, None); t4 = T()
    vals, rows = net.forward(pts, nrm, radii, _lib.frame_init(bb[0], bb[1]), pipe.scale_sdf, feats1=f1, importance=imp); t5 = T()
    full = net.stitch(vals, rows); t6 = T()
    print("build %.2f aggregate %.2f geom_from_pipe %.2f net_init %.2f net_forward %.2f stitch %.2f total %.2f" % tuple(1e3*x for x in (t1-t0, t2-t1, t3-t2, t4-t3, t5-t4, t6-t5, t6-t0)))

"""geometry half only (asr_hip_implicit_build) of the 10 M-point bench cloud, N times: for rocprofv3 --kernel-trace.
usage: python scripts/prof_geom.py [points] [overlap 0/1]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "adaptive-surface-reconstruction_amd")]
import torch
from asr_hip import synth
from asr_hip.pipeline import ImplicitPipeline
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
overlap = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
pts, nrm = synth.scan_cloud(n, seed=1000, device=dev, density_variance=float(sys.argv[3]) if len(sys.argv) > 3 else 1.0)
radii = synth.knn_radii_gpu(pts, 24)
bb = synth.bounding_box(pts, 0.1)
pipe = ImplicitPipeline(synth.make_weights(4, seed=0), device=dev, precision="bf16x3")
pipe.ctx.set_option("overlap", overlap)
for i in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.build(pts, radii, bb[0], bb[1])
    torch.cuda.synchronize()
    print("build %d: %.2f ms" % (i, (time.perf_counter() - t0) * 1e3), flush=True)

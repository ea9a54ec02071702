"""aggregation search of the 10 M-point bench cloud, serial (overlap 0), per variant of the light-row pass:
wall time of the whole geometry build with one wave per voxel / quad queries / quad ablations (results of the ablations are invalid).
usage: python scripts/search_ablation.py [points]"""
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
pipe = ImplicitPipeline(synth.make_weights(4, seed=0), device=dev, precision="f16x2")
pipe.ctx.set_option("overlap", 0)
for name, opts in (("wave-per-voxel", {"search_quad": 0}), ("quad", {"search_quad": 1}),
                   ("quad natural order", {"search_quad": 1, "search_xcd_run": 0}),
                   ("quad xcd run 16", {"search_quad": 1, "search_xcd_run": 16}),
                   ("quad xcd run 256", {"search_quad": 1, "search_xcd_run": 256}),
                   ("quad xcd run 2048", {"search_quad": 1, "search_xcd_run": 2048}),
                   ("quad, look-ups only", {"search_quad": 1, "search_quad_stop": 1}),
                   ("quad, look-ups + walk", {"search_quad": 1, "search_quad_stop": 2})):
    for k, dflt in (("search_quad", 1), ("search_quad_stop", 0), ("search_xcd_run", 64)):
        pipe.ctx.set_option(k, opts.get(k, dflt))
    import time
    ms = []
    for i in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        try:
            pipe.build(pts, radii, bb[0], bb[1])
        except Exception as e:  # an ablation can trip consistency checks downstream
            print(name, "->", str(e)[:200])
            break
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t0) * 1e3)
    print("%-24s build ms (serial) %s" % (name, ["%.2f" % m for m in ms]), flush=True)

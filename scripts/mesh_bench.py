"""Times the contouring + component filter stage on the analytic field of the synthetic scan scene
(what a trained network approximates) on the grid of an n-point cloud.  GPU only."""
import sys
import time

import torch

sys.path[:0] = ["adaptive-surface-reconstruction_amd"]
from asr_hip import ops, synth  # noqa: E402
from asr_hip.pipeline import ImplicitPipeline  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda:0")
pts, nrm = synth.scan_cloud(n, seed=1000, device=dev)
radii = synth.knn_radii_gpu(pts, 24)
bb = synth.bounding_box(pts, 0.1)
pipe = ImplicitPipeline(synth.make_weights(4, seed=0), device=dev)
pipe.build(pts, radii, bb[0], bb[1])
centers, vs = pipe.get("voxel_centers0"), pipe.get("voxel_sizes0")
sdf = synth._scene_sdf(centers)
field = torch.stack([sdf, sdf.abs() / vs], 1).contiguous()


def timed(f, reps=3):
    f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3, r


t_dual, duals = timed(lambda: pipe.dual_cells())
t_cont, (v, t) = timed(lambda: ops.contour(field, duals, centers, 1.0, ctx=pipe.ctx))
t_comp, (v2, t2) = timed(lambda: ops.remove_components(v, t, 2**63 - 1, 3, ctx=pipe.ctx))
print("points %d voxels %d duals %d vertices %d triangles %d -> kept %d / %d" %
      (n, centers.shape[0], duals.shape[0], v.shape[0], t.shape[0], v2.shape[0], t2.shape[0]))
print("ms: dual cells %.2f  contour %.2f  components %.2f" % (t_dual, t_cont, t_comp))

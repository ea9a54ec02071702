import sys
import numpy as np
import torch
sys.path[:0] = ["adaptive-surface-reconstruction_amd", "."]
from asr_hip import ops, synth
from asr_hip.pipeline import ImplicitPipeline
from oracle import oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda:0")
pts, nrm = synth.scan_cloud(n, seed=31, device=dev)
radii = synth.knn_radii_gpu(pts, 24)
bb = synth.bounding_box(pts, 0.1)
pipe = ImplicitPipeline(synth.make_weights(4, seed=0), device=dev)
pipe.build(pts, radii, bb[0], bb[1])
centers, vs = pipe.get("voxel_centers0"), pipe.get("voxel_sizes0")
sdf = synth._scene_sdf(centers)
field = torch.stack([sdf, sdf.abs() / vs], 1).contiguous()
duals = pipe.dual_cells()
gv, gt = ops.contour(field, duals, centers, 1.0, ctx=pipe.ctx)
gv, gt = gv.cpu().numpy(), gt.cpu().numpy()
wv, wt = O.create_triangle_mesh(field.cpu().numpy(), duals.cpu().numpy(), centers.cpu().numpy(), 1.0)
print("shapes", gv.shape, wv.shape, gt.shape, wt.shape, "verts equal", np.array_equal(gv.view(np.uint32), wv.view(np.uint32)))
bad = np.nonzero((gt != wt).any(1))[0]
print("bad triangles", len(bad), bad[:20])
for b in bad[:10]:
    print(b, "gpu", gt[b - 1:b + 3].tolist(), "ref", wt[b - 1:b + 3].tolist())
if len(bad):
    b = bad[0]
    ids = [gt[b][0], gt[b][1], gt[b][2], gt[b + 1][2]]
    P = gv[ids]
    print("ids", ids)
    for r in P:
        print([float(x).hex() for x in r])
    f = np.float32
    def q(a, b, mode):
        d = (P[a] - P[b]).astype(f)
        s = d * d
        return (f(s[0] + s[1]) + s[2]) if mode == 0 else (s[0] + f(s[1] + s[2]))
    print("left assoc q02 %s q13 %s" % (q(0, 2, 0).item().hex(), q(1, 3, 0).item().hex()))
    print("eigen order q02 %s q13 %s" % (q(0, 2, 1).item().hex(), q(1, 3, 1).item().hex()))

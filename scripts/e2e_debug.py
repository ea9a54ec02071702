import sys, time
import numpy as np, torch
sys.path[:0] = ["adaptive-surface-reconstruction_amd"]
import adaptivesurfacereconstruction as asr
from asr_hip import synth, ops as _ops
from asr_hip.pipeline import ImplicitPipeline
n = int(sys.argv[1])
def T(msg, t0):
    torch.cuda.synchronize(); print("%-28s %.3f s" % (msg, time.perf_counter() - t0), flush=True); return time.perf_counter()
p, q = synth.scan_cloud(n, seed=3, device="cuda:0")
pts, nrm = p.cpu().numpy(), q.cpu().numpy()
w = synth.make_weights(1, seed=0)
t = time.perf_counter()
tree = asr.KDTree(pts); t = T("KDTree", t)
r = _ops.knn_radius(tree._frame, tree._points, 24); t = T("knn radius", t)
_, inl = _ops.knn_radius(tree._frame, tree._points, 24, r, 0.5, 1, want_inlier=True); t = T("inlier", t)
cnt = _ops.radius_neighbor_count(tree._frame, tree._points, r); t = T("radius neighbour count", t)
dens = _ops.density_inlier(cnt.cpu().numpy(), 10.0); t = T("density inlier (host) %d" % dens.sum(), t)
radii = r.cpu().numpy(); inlier = inl.cpu().numpy().astype(bool)
pts, nrm, radii = pts[inlier], nrm[inlier], radii[inlier]; t = T("filter (%d left)" % len(pts), t)
dev = torch.device("cuda")
pipe = ImplicitPipeline(w, device="cuda:0"); t = T("pipeline ctor", t)
pipe.forward(torch.from_numpy(pts).to(dev), torch.from_numpy(nrm).to(dev), torch.from_numpy(radii).to(dev), pts.min(0), pts.max(0)); t = T("forward", t)
print(pipe.stage_ms(), flush=True)
duals = pipe.dual_cells(); t = T("dual cells %d" % duals.shape[0], t)
v, tri = _ops.contour(pipe.get("values"), duals, pipe.get("voxel_centers0"), 1.0, ctx=pipe.ctx); t = T("contour %d %d" % (v.shape[0], tri.shape[0]), t)
v2, t2 = _ops.remove_components(v, tri, 8, 3, ctx=pipe.ctx); t = T("components %d %d" % (v2.shape[0], t2.shape[0]), t)

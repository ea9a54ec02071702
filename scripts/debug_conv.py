"""debug helper: run pieces of the path in isolation (each call in its own process under timeout)"""
import sys, os
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "adaptive-surface-reconstruction_amd"), REPO, os.path.join(REPO, "tests")]
from asr_hip import ops, synth
import parity
from oracle import oracle as O
what = sys.argv[1]
gpu = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu)
p, q = synth.scan_cloud(6000, seed=11, device="cpu")
pts, nrm = p.numpy(), q.numpy()
rad = synth.knn_radii(pts, 24); bb = synth.bounding_box(pts, 0.1)
item = parity.oracle_geometry(pts, rad, *bb)
idx, kidx, rs = item["neighbors_index0"], item["neighbors_kernel_index0"], item["neighbors_row_splits0"]
v = len(rs) - 1
rng = np.random.default_rng(0)
if what == "rowgroups":
    perm = ops.row_groups(t(kidx), t(rs), 256); torch.cuda.synchronize()
    print("perm ok", sorted(perm.cpu().tolist()) == list(range(v)))
elif what.startswith("conv"):
    cin, cout = (int(x) for x in what.split("_")[1:3])
    use_perm = what.endswith("perm")
    f = rng.standard_normal((v, cin)).astype(np.float32)
    W = (rng.standard_normal((55, cin, cout)) * 0.1).astype(np.float32)
    perm = ops.row_groups(t(kidx), t(rs), 256) if use_perm else None
    out = ops.sparse_conv(t(W), t(f), t(idx), t(kidx), t(rs), algo=2, row_perm=perm); torch.cuda.synchronize()
    ref = O.sparse_conv(W, f, idx, kidx, None, rs, False)
    print(what, "max err", np.abs(out.cpu().numpy() - ref).max())
elif what == "path":
    from asr_hip.pipeline import ImplicitPipeline
    w = synth.make_weights(4, seed=1)
    pipe = ImplicitPipeline(w, device=gpu)
    vals = pipe.forward(t(pts), t(nrm), t(rad), bb[0], bb[1]); torch.cuda.synchronize()
    ref = parity.oracle_forward(pts, nrm, rad, bb[0], bb[1], w)
    print("path max err", np.abs(vals.cpu().numpy() - ref["values"]).max())

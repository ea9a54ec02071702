import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "adaptive-surface-reconstruction_amd"), os.path.join(REPO, "tests")]
import torch
from asr_hip import sharding, synth, _lib
from asr_hip.pipeline import ImplicitPipeline
dev = torch.device("cuda:0")
pts, nrm = synth.scan_cloud(30000, seed=55, device=dev, density_variance=10.0)
rad = synth.knn_radii_gpu(pts, 24)
bb = synth.bounding_box(pts, 0.1)
weights = synth.make_weights(2, seed=6)
single = ImplicitPipeline(weights, device=dev)
v1 = single.forward(pts, nrm, rad, bb[0], bb[1]).clone()
f1 = single.get("feats1"); code1 = single.get("code"); imp1 = single.get("importance")
sp = sharding.ShardedImplicitPipeline(weights, dev)
# instrument
net_cls = sharding.ShardedNetwork
orig_block = net_cls._block
def dbg_block(self, name, x, i, imp, with_imp, imp_replicated=False):
    if name == "sparseconv_encblock0":
        print("feats1 equal:", torch.equal(x, f1), float((x - f1).abs().max()))
        print("imp prefix equal:", torch.equal(imp, imp1[:imp.shape[0]]))
    r = orig_block(self, name, x, i, imp, with_imp, imp_replicated)
    if name == "sparseconv_decblock0":
        print("code equal:", torch.equal(r[0], code1), float((r[0] - code1).abs().max()))
    return r
net_cls._block = dbg_block
v2 = sp.forward(pts, nrm, rad, bb[0], bb[1])
print("values equal:", torch.equal(v1, v2), float((v1 - v2).abs().max()))

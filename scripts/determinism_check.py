"""runs the 10 M-point bench cloud through the whole path several times and compares the values bit for bit
(a race in the kernels -- e.g. a panel DMA that is read too early -- would show up as run-to-run differences)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "adaptive-surface-reconstruction_amd")]
import torch
from asr_hip import synth
from asr_hip.pipeline import ImplicitPipeline
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
pts, nrm = synth.scan_cloud(n, seed=1000, device=dev)
radii = synth.knn_radii_gpu(pts, 24)
bb = synth.bounding_box(pts, 0.1)
for prec in ("f16x2", "bf16x3", "f16"):
    pipe = ImplicitPipeline(synth.make_weights(1, seed=0), device=dev, precision=prec)
    ref = pipe.forward(pts, nrm, radii, bb[0], bb[1]).clone()
    code = pipe.get("code").clone()
    for i in range(8):
        v = pipe.forward(pts, nrm, radii, bb[0], bb[1])
        assert torch.equal(v, ref), (prec, i, float((v - ref).abs().max()))
        assert torch.equal(pipe.get("code"), code), (prec, i)
    print(prec, "9 identical runs, V0 =", int(ref.shape[0]), flush=True)

"""times the stages of the 10 M-point forward with the library given in ASR_EXP_LIB (measurement builds whose numbers
may be garbage): usage: ASR_EXP_LIB=path python scripts/exp_time.py"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "adaptive-surface-reconstruction_amd")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from asr_hip import _lib
if os.environ.get("ASR_EXP_LIB"):
    _lib.LIB_PATH = os.environ["ASR_EXP_LIB"]
from asr_hip import synth
from asr_hip.pipeline import ImplicitPipeline
dev = torch.device("cuda:0")
pts, nrm = synth.scan_cloud(10_000_000, seed=1000, device=dev)
radii = synth.knn_radii_gpu(pts, 24)
bb = synth.bounding_box(pts, 0.1)
pipe = ImplicitPipeline(synth.make_weights(1, seed=0, init="reference"), device=dev, precision="f16x2")
acc = {}
for i in range(8):
    pipe.forward(pts, nrm, radii, bb[0], bb[1])
    torch.cuda.synchronize()
    if i >= 2:
        for k, v in pipe.stage_ms().items():
            acc[k] = acc.get(k, 0.0) + v / 6
print(os.environ.get("ASR_EXP_LIB", "default"), {k: round(v, 3) for k, v in acc.items()})

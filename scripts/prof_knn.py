"""exact 24-NN radii of the 10 M-point bench cloud, three times: for rocprofv3 --kernel-trace"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "adaptive-surface-reconstruction_amd")]
import torch
from asr_hip import synth
dev = torch.device("cuda:0")
pts, nrm = synth.scan_cloud(int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, seed=1000, device=dev)
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = synth.knn_radii_gpu(pts, 24)
    torch.cuda.synchronize(); print("knn %d: %.2f ms" % (i, (time.perf_counter() - t0) * 1e3), flush=True)

"""world size 1: the library's sharded driver against the monolithic one on the 10 M-point bench cloud, with the RCCL
communicator made BEFORE or AFTER the first forward (usage: python scripts/shard_world1.py [rccl_first 0/1])"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "adaptive-surface-reconstruction_amd")]
import torch
from asr_hip import synth, shardcomm
from asr_hip.pipeline import ImplicitPipeline
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev = torch.device("cuda:0")
pts, nrm = synth.scan_cloud(10_000_000, seed=1000, device=dev)
radii = synth.knn_radii_gpu(pts, 24)
bb = synth.bounding_box(pts, 0.1)
w = synth.make_weights(1, seed=0, init="reference")
pipe = ImplicitPipeline(w, device=dev, precision="f16x2")
print("affinity at start:", len(os.sched_getaffinity(0)), flush=True)
def run(name, f):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8 * 1e3
    print("%-28s %.2f ms  %s" % (name, dt, {k: round(v, 2) for k, v in pipe.stage_ms().items()}), flush=True)
rc = None
if first:
    rc = shardcomm.RcclComm(pipe.ctx)
    print("affinity after rccl init:", len(os.sched_getaffinity(0)), flush=True)
    run("sharded, rccl comm (first)", lambda: pipe.forward_sharded(rc, pts, nrm, radii, bb[0], bb[1]))
run("monolithic", lambda: pipe.forward(pts, nrm, radii, bb[0], bb[1]))
if rc is None:
    rc = shardcomm.RcclComm(pipe.ctx)
    print("affinity after rccl init:", len(os.sched_getaffinity(0)), flush=True)
run("sharded, rccl comm", lambda: pipe.forward_sharded(rc, pts, nrm, radii, bb[0], bb[1]))
run("monolithic again", lambda: pipe.forward(pts, nrm, radii, bb[0], bb[1]))

"""Experiment: two contexts on two streams, the geometry build of step i+1 overlaps the network of
step i.  Prints serial and pipelined ms per step (10 M points)."""
import sys
import time

import torch

sys.path[:0] = ["adaptive-surface-reconstruction_amd"]
from asr_hip import synth  # noqa: E402
from asr_hip.pipeline import ImplicitPipeline  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
pts, nrm = synth.scan_cloud(n, seed=1000, device=dev)
radii = synth.knn_radii_gpu(pts, 24)
bb = synth.bounding_box(pts, 0.1)
w = synth.make_weights(1, seed=0, init="reference")
pipes = [ImplicitPipeline(w, device=dev), ImplicitPipeline(w, device=dev)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
torch.cuda.synchronize()


def run(depth, steps):
    t0 = time.perf_counter()
    for i in range(steps):
        j = i % depth
        with torch.cuda.stream(streams[j]):
            v = pipes[j].forward(pts, nrm, radii, bb[0], bb[1])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, v


for depth in (1, 2):
    run(depth, 2)
    ms, v = run(depth, K)
    print("depth %d: %.2f ms/step  (values %s, finite %s)" % (depth, ms, tuple(v.shape), bool(torch.isfinite(v).all())))

# variant: builds on a high-priority stream, networks on a normal one
hi = [torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=-1)]
lo = [torch.cuda.Stream(), torch.cuda.Stream()]
net_done = [None, None]


def run_prio(steps):
    t0 = time.perf_counter()
    for i in range(steps):
        j = i % 2
        if net_done[j] is not None:
            hi[j].wait_event(net_done[j])
        with torch.cuda.stream(hi[j]):
            pipes[j].build(pts, radii, bb[0], bb[1])
            ev = hi[j].record_event()
        lo[j].wait_event(ev)
        with torch.cuda.stream(lo[j]):
            v = pipes[j].network(pts, nrm, bb[0], bb[1])
            net_done[j] = lo[j].record_event()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, v


run_prio(2)
ms, v = run_prio(K)
print("depth 2 + priorities: %.2f ms/step (finite %s)" % (ms, bool(torch.isfinite(v).all())))

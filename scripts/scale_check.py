"""Large-cloud check: the split-arithmetic paths (f16x2, bf16x3; (plan-driven kernel, falling back to the table-driven one where a feature
matrix exceeds the 4 GB buffer-addressing range)) against the exact f32 kernel on the same cloud.
usage: python scripts/scale_check.py [points]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "adaptive-surface-reconstruction_amd")]
import torch
from asr_hip import synth
from asr_hip.pipeline import ImplicitPipeline
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
dev = torch.device("cuda:0")
pts, nrm = synth.scan_cloud(n, seed=77, device=dev)
t0 = time.perf_counter()
radii = synth.knn_radii_gpu(pts, 24)
torch.cuda.synchronize()
print("knn %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
bb = synth.bounding_box(pts, 0.1)
w = synth.make_weights(1, seed=0)
out = {}
for prec in ("f32", "bf16x3", "f16x2"):
    pipe = ImplicitPipeline(w, device=dev, precision=prec)
    pipe.ctx.sconv_variant_counts(reset=True)
    for i in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        v = pipe.forward(pts, nrm, radii, bb[0], bb[1])
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out[prec] = v.clone()
    counts = pipe.ctx.sconv_variant_counts()
    print(prec, "%.1f ms/step, %.3g points/s" % (dt * 1e3, n / dt), "voxels", list(pipe.sizes.num_voxels)[:5],
          "reserved GB %.1f" % (pipe.ctx.reserved_bytes() / 2**30), flush=True)
    if prec != "f32":
        print("  launches plan/table/slot-range split:", sum(c for k, c in counts.items() if len(k) >= 7 and k[6] == 1) // 2,
              sum(c for k, c in counts.items() if len(k) >= 7 and k[6] == 0) // 2,
              sum(c for k, c in counts.items() if len(k) == 8) // 2, flush=True)
    del pipe
    torch.cuda.empty_cache()
a = out["f32"]
scale = float(a.abs().max())
for prec in ("bf16x3", "f16x2"):
    b = out[prec]
    assert bool(torch.isfinite(a).all()) and bool(torch.isfinite(b).all())
    err = float((a - b).abs().max())
    print("range %.3g, max |%s - f32| %.3e (%.2e of the range)" % (scale, prec, err, err / scale))
assert err <= 3e-5 * max(scale, 1.0), err

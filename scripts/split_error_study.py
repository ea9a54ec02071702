#!/usr/bin/env python3
"""Why does bf16x3 carry more error than f16x2 / a plain fp32 sum?  

Runs on the GPU box.  For one cloud (default 1 M points, full width, the bench's weights) the oracle's double-accumulating
network is the exact result; every arithmetic of the library (f32 MFMA, bf16x3, f16x2) is compared with it: max error / range,
rms error, share of elements within 1e-5 + 1e-5 |exact|, for `code` and `values`.  `--lib path` loads another build of
libasr_hip.so (e.g. one compiled with -DASR_BF16X3_CHAIN=1) -- one process per library; the exact result is cached in /tmp.

usage: python scripts/split_error_study.py [--points N] [--lib build_variants/libasr_hip_chain1.so] [--out gpurun_out/x.json]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "adaptive-surface-reconstruction_amd"), REPO, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--lib", default=None)
    ap.add_argument("--out", default=None)
    ap.add_argument("--steps", type=int, default=3, help="timed forwards per arithmetic")
    ap.add_argument("--time-points", type=int, default=10_000_000, help="U-Net time of bf16x3 on a cloud of this size (0: skip)")
    args = ap.parse_args()
    from asr_hip import _lib
    if args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)
    from asr_hip import synth
    from asr_hip.pipeline import ImplicitPipeline
    import parity
    from oracle import oracle as O

    gpu = torch.device("cuda:0")
    n = args.points
    pts, nrm = synth.scan_cloud(n, seed=1000, device=gpu)
    radii = torch.from_numpy(synth.knn_radii(pts.cpu().numpy(), 24)).to(gpu)
    bb = synth.bounding_box(pts, 0.1)
    weights = synth.make_weights(1, seed=2)
    cache = "/tmp/split_error_exact_%d.npz" % n
    if os.path.exists(cache):
        z = np.load(cache)
        exact = {k: z["exact_" + k] for k in ("code", "values")}
        ref32 = {k: z["ref32_" + k] for k in ("code", "values")}
    else:
        hp, hn = pts.cpu().numpy(), nrm.cpu().numpy()
        t0 = time.time()
        item = parity.oracle_geometry(hp, radii.cpu().numpy(), bb[0], bb[1])
        with O.precise():
            exact = parity.oracle_network(item, hp, hn, weights)
        ref32 = parity.oracle_network(item, hp, hn, weights)
        print("oracle: %.1f s" % (time.time() - t0), file=sys.stderr)
        np.savez(cache, **{"exact_" + k: exact[k] for k in ("code", "values")}, **{"ref32_" + k: ref32[k] for k in ("code", "values")})

    def stats(got, k):
        e = got.astype(np.float64) - exact[k]
        scale = max(1.0, float(np.abs(exact[k]).max()))
        return {"max_err_over_range": float(np.abs(e).max()) / scale, "rms_err_over_range": float(np.sqrt(np.mean(e * e))) / scale,
                "mean_err_over_range": float(e.mean()) / scale, "share": parity.pass_fraction(got, exact[k]), "range": scale}

    rec = {"points": n, "lib": args.lib or "default", "arithmetic": {}}
    rec["arithmetic"]["fp32 CPU oracle"] = {k: stats(ref32[k], k) for k in ("code", "values")}
    for precision in ("f32", "bf16x3", "f16x2"):
        pipe = ImplicitPipeline(weights, device=gpu, precision=precision)
        values = pipe.forward(pts, nrm, radii, bb[0], bb[1])
        r = {"code": stats(pipe.get("code").cpu().numpy(), "code"), "values": stats(values.cpu().numpy(), "values")}
        torch.cuda.synchronize()
        ms = []
        for _ in range(args.steps):
            pipe.forward(pts, nrm, radii, bb[0], bb[1])
            ms.append(pipe.stage_ms()["unet"])
        r["unet_ms"] = min(ms)
        rec["arithmetic"][precision] = r
        del pipe
    if args.time_points:
        del pts, nrm, radii
        torch.cuda.empty_cache()
        pts, nrm = synth.scan_cloud(args.time_points, seed=1000, device=gpu)
        radii = synth.knn_radii_gpu(pts, 24)
        bb = synth.bounding_box(pts, 0.1)
        pipe = ImplicitPipeline(weights, device=gpu, precision="bf16x3")
        ms = []
        for _ in range(2 + args.steps):
            pipe.forward(pts, nrm, radii, bb[0], bb[1])
            ms.append(pipe.stage_ms()["unet"])
        rec["bf16x3_unet_ms_at_%d" % args.time_points] = min(ms[2:])
    print(json.dumps(rec, indent=1))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(rec, f, indent=1)


if __name__ == "__main__":
    main()

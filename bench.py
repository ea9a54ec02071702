"""Benchmark of the hot path: input points/sec from device-resident points / normals / radii to
signed implicit values (BASELINE.json metric), one process per GPU.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A step is one pass of the whole path (octree -> 5 grids -> aggregation search -> continuous conv
-> 53 sparse convs -> decoder) over one synthetic 10 M-point scan-like cloud (config C3 of
BASELINE.json / SURVEY 8(d)) that is already resident in HBM.  The path shards by scan: every
rank owns one scan, there is no data-path collective ("scaling": "weak"); value = points all
ranks processed / max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(REPO, "adaptive-surface-reconstruction_amd"), REPO, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TFLOPS = 157.3  # f32-input MFMA (v_mfma_f32_16x16x4_f32) dense peak
MFMA_16BIT_PEAK_TFLOPS = 2500.0  # bf16 / f16 MFMA dense peak (same guide)
# dtype of the bench line, dense MFMA peak that prices the ALGORITHMIC flops of the sparse convs, kernel name
PRECISION_INFO = {
    "f32": ("f32", MFMA_F32_PEAK_TFLOPS, "k_sconv_mfma", "f32-input MFMA, exact"),
    # f32 in / f32 out with fp32-class error, computed as six bf16 MFMA products per algorithmic product: the
    # roofline of the ALGORITHMIC FLOP is the bf16 dense peak / 6 (so that the fraction is the share of the matrix
    # pipe's peak the executed instructions reach); the ratio to the f32-input MFMA peak is reported beside it
    "bf16x3": ("f32", MFMA_16BIT_PEAK_TFLOPS / 6.0, "k_sconv_plan16<bf16x3>",
               "bf16 dense MFMA peak (%.0f TFLOP/s) / 6: the kernel evaluates every f32 product as six bf16 MFMA "
               "products (exact three-way split, fp32-class result); f32-input MFMA peak %.1f TFLOP/s for comparison"
               % (MFMA_16BIT_PEAK_TFLOPS, MFMA_F32_PEAK_TFLOPS)),
    "f16": ("f16", MFMA_16BIT_PEAK_TFLOPS, "k_sconv_plan16<f16>", "f16 MFMA dense peak"),
}


def conv_flops(sizes, shapes):
    """algorithmic FLOP of the 53 sparse convs: 2 * P * Cin * Cout per conv with P the actual pair
    count; a transition has one pair per voxel of the finer grid.  conv1a + conv1b of a
    block run as ONE launch (second filter bank), so the 53 convs are 44 launches."""
    V = list(sizes.num_voxels)
    P = list(sizes.num_pairs)
    total, launches = 0.0, 0
    for name, shp in shapes.items():
        if not name.endswith(".kernel") or name.startswith("cconv"):
            continue
        k, cin, cout = shp
        blk = name.split(".")[0]
        lvl = int(blk[-1])
        if k == 55:
            pair_counts = [P[lvl]]
        elif blk.startswith("sparseconv_down"):
            pair_counts = [V[lvl - 1]] + ([V[3]] if lvl == 3 else [])  # down3 also runs 3->4
        else:
            pair_counts = [V[lvl]]
        for pairs in pair_counts:
            total += 2.0 * pairs * cin * cout
            launches += 0 if name.endswith(".conv1b.kernel") else 1
    return total, launches


def rank_seed(rank):
    """one scan per rank: the path shards by scan, no data moves between ranks"""
    return int(os.environ.get("ASR_BENCH_SEED_BASE", 1000)) + rank


def max_over_ranks(dt, world, device):
    """contract: the step time of the job is the slowest rank's"""
    if world <= 1:
        return dt
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def job_value(world, points_per_rank, steps, dt):
    """whole-job throughput: points processed by all ranks / max-over-ranks time"""
    return world * points_per_rank * steps / dt


def pmc_traffic(points, precision="f32"):
    """HBM bytes per sparse-conv launch from the committed rocprofv3 PMC passes of this very command
    (profiles/r*_pmc_traffic.json, produced by scripts/make_profiles.sh); PMC counters cannot be
    collected from inside the timed process.  None when no matching profile is committed."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_traffic.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("points", 10_000_000) == points and d.get("precision", "f32") == precision:
            best = (d["hbm_bytes_per_launch"], os.path.basename(path))
    return best


def cpu_baseline(n_sample, seed):
    """the oracle ("port" of the reference path incl. the Open3D op semantics) timed on the host
    cores on a bounded sample of the same workload generator"""
    import parity
    from asr_hip import synth
    from oracle import oracle as O
    O.lib()
    pts, nrm = synth.scan_cloud(n_sample, seed=seed, device="cpu")
    points, normals = pts.numpy(), nrm.numpy()
    radii = synth.knn_radii(points, 24)
    bb_min, bb_max = synth.bounding_box(points, 0.1)
    weights = synth.make_weights(1, seed=0, init="reference")
    timings = {}
    t0 = time.time()
    with O.dense():  # sparse convs evaluated like Open3D's CPU op: dense [32][55*cin] matrix per voxel block
        parity.oracle_forward(points, normals, radii, bb_min, bb_max, weights, timings=timings)
    dt = time.time() - t0
    return {"value": n_sample / dt, "unit": "points/s", "cores": os.cpu_count(), "kind": "port",
            "sample": "%d-point slice of the C3 scan generator, whole path, %.1f s, sparse convs in the dense "
                      "Open3D-style evaluation (55*cin deep per voxel); stage s: %s" %
                      (n_sample, dt, {k: round(v, 2) for k, v in timings.items()})}


def exact_f32_run(weights, dev, inputs, n, steps, shapes, values=None):
    """Outside the timed region and not part of `value`: the same cloud through the f32-input MFMA kernel
    (v_mfma_f32_16x16x4_f32, a bit-exact fmaf chain) -- the round-1 arithmetic, for comparison; `values`: the timed
    run's result, whose largest deviation from this kernel's is reported relative to the range of the values."""
    from asr_hip.pipeline import ImplicitPipeline
    pipe = ImplicitPipeline(weights, device=dev, precision="f32")
    ref = pipe.forward(*inputs)
    torch.cuda.synchronize()
    dev_rel = None
    if values is not None and values.shape == ref.shape:
        scale = float(ref.abs().max())
        dev_rel = {"max_abs_deviation": float((values.double() - ref.double()).abs().max()), "range_of_values": scale}
        dev_rel["deviation_over_range"] = dev_rel["max_abs_deviation"] / scale if scale > 0 else None
    unet = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe.forward(*inputs)
        unet += pipe.stage_ms()["unet"]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    flops, launches = conv_flops(pipe.sizes, shapes)
    tf = flops / (unet / steps * 1e-3) / 1e12
    return {"ms_per_step": dt / steps * 1e3, "points_per_s": n * steps / dt, "unet_ms": unet / steps,
            "kernel": "k_sconv_mfma", "achieved_tflops": tf, "frac_of_f32_mfma_peak": tf / MFMA_F32_PEAK_TFLOPS,
            "timed_values_vs_this_kernel": dev_rel}


def mesh_stage(pipe, synth):
    """Outside the timed region and not part of `value`: the stage after the path (dual cells, dual
    contouring, component filter) on the analytic signed distance of the synthetic scene -- random
    weights give no surface to contour."""
    centers, vs = pipe.get("voxel_centers0"), pipe.get("voxel_sizes0")
    sdf = synth._scene_sdf(centers)
    field = torch.stack([sdf, sdf.abs() / vs], 1).contiguous()
    pipe.mesh(values=field)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    v, t = pipe.mesh(values=field)
    torch.cuda.synchronize()
    return {"ms": round((time.perf_counter() - t0) * 1e3, 2), "vertices": int(v.shape[0]),
            "triangles": int(t.shape[0]), "field": "analytic scene sdf on grid 0"}


def pipelined_rate(pipe, weights, dev, inputs, n, steps, depth=2):
    """Outside the timed region and not part of `value`: throughput when consecutive clouds are pipelined
    over `depth` contexts, each with its own stream, arena and host thread, so that the geometry build of
    one cloud (latency-bound small kernels, host read-backs of sizes) overlaps the network of another.
    Same work per cloud as a serial step."""
    import threading
    from asr_hip.pipeline import ImplicitPipeline
    pipes = [pipe] + [ImplicitPipeline(weights, device=dev) for _ in range(depth - 1)]
    streams = [torch.cuda.Stream() for _ in range(depth)]
    for s_ in streams:
        s_.wait_stream(torch.cuda.current_stream())
    out = [None] * depth

    def worker(t, k):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(streams[t]):
            for _ in range(k):
                out[t] = pipes[t].forward(*inputs)
        streams[t].synchronize()

    def run(k):
        ths = [threading.Thread(target=worker, args=(t, k)) for t in range(depth)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        torch.cuda.synchronize()

    run(1)
    per = max(1, steps // depth)
    t0 = time.perf_counter()
    run(per)
    dt = time.perf_counter() - t0
    assert all(bool(torch.isfinite(v).all()) for v in out)
    clouds = per * depth
    return {"points_per_s": n * clouds / dt, "ms_per_cloud": dt / clouds * 1e3, "depth": depth, "clouds": clouds}


def one_scan_line(args, world, n, dt, sharded):
    """bench line of --shard one-scan: ONE cloud over all ranks, total work fixed ("strong")"""
    steps = max(args.steps, 1)
    net = sharded.net
    return {
        "metric": "input points/sec to signed implicit values",
        "value": n * steps / dt,
        "unit": "points/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "one %d-point scan-like synthetic cloud sharded over %d GPU(s) by Morton range; "
                               "grids replicated, aggregation + 53 sparse convs + decoder on owned rows, halo "
                               "exchange per convolution (RCCL send/recv), values stitched by all-reduce" % (n, world),
                   "points": n,
                   "voxels": net.v,
                   "owned_rows_rank0": [int(r.numel()) for r in net.rows],
                   "halo_rows_rank0": {"%s%d" % k: v for k, v in net.halo_rows().items()},
                   "parallelism": "spatial sharding, %d ranks" % world},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--points", type=int, default=int(os.environ.get("ASR_BENCH_POINTS", 10_000_000)))
    ap.add_argument("--cpu-sample", type=int, default=int(os.environ.get("ASR_BENCH_CPU_SAMPLE", 300_000)))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the informational two-context run (profiling)")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--precision", choices=["f32", "bf16x3", "f16"], default=os.environ.get("ASR_BENCH_PRECISION", "bf16x3"),
                    help="arithmetic of the 53 sparse convs: bf16x3 (default) = f32 in / f32 out, every operand split "
                         "exactly into three bf16 terms, six bf16 MFMAs per product, f32 accumulate (fp32-class "
                         "results, same parity bound as f32); f32 = f32-input MFMA, a bit-exact fmaf chain; "
                         "f16 = f16 activations and weights (config C5)")
    ap.add_argument("--no-exact-f32", action="store_true",
                    help="skip the informational re-run of the same cloud on the f32-input MFMA kernel (profiling)")
    ap.add_argument("--density-variance", type=float, default=1.0,
                    help="10 = the mixed-density cloud of BASELINE config C5")
    ap.add_argument("--shard", choices=["replicas", "one-scan"], default="replicas",
                    help="replicas (default): one scan per GPU, no collective, weak scaling.  one-scan: ONE cloud of "
                         "--points points sharded over the GPUs by Morton range with halo exchange (RCCL), strong scaling")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the MI355X path has no CPU fallback")
    # (ASR_BENCH_FORCE_DEVICE: run several ranks on one GPU with --backend gloo, to exercise the
    # multi-process logic on a single-GPU box)
    dev_index = int(os.environ.get("ASR_BENCH_FORCE_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, rank=rank, world_size=world,
                                device_id=dev if args.backend == "nccl" else None)

    from asr_hip import synth
    from asr_hip.pipeline import ImplicitPipeline

    # ---- inputs (untimed): one scan per rank, radii = exact 24-NN distance -------------------
    n = args.points
    one_scan = args.shard == "one-scan"
    # one-scan: every rank holds the same cloud (seed of rank 0); replicas: one scan per rank
    pts, nrm = synth.scan_cloud(n, seed=rank_seed(0 if one_scan else rank), device=dev,
                                density_variance=args.density_variance)
    t_knn = time.perf_counter()
    radii = synth.knn_radii_gpu(pts, 24)  # pre-filter row D.4 on the GPU, untimed input preparation
    torch.cuda.synchronize()
    t_knn = time.perf_counter() - t_knn
    bb_min, bb_max = synth.bounding_box(pts, 0.1)
    weights = synth.make_weights(1, seed=0, init="reference")  # released weights are not in the repo
    if one_scan:
        from asr_hip.sharding import ShardedImplicitPipeline
        sharded = ShardedImplicitPipeline(weights, dev, precision="bf16x3" if args.precision == "bf16x3" else "f32")
        pipe = sharded.pipe
    else:
        pipe = ImplicitPipeline(weights, device=dev, precision=args.precision)
    shapes = synth.unet5_param_shapes(1)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        if one_scan:
            return sharded.forward(pts, nrm, radii, bb_min, bb_max)
        return pipe.forward(pts, nrm, radii, bb_min, bb_max)

    for _ in range(args.warmup):
        step()
    barrier()
    stage_sum = dict.fromkeys(ImplicitPipeline.STAGES, 0.0)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        values = step()
        # stage times come from hip events recorded on the stream inside the library
        if not one_scan:
            for k, v in pipe.stage_ms().items():
                stage_sum[k] += v
    barrier()
    dt = time.perf_counter() - t0
    dt = max_over_ranks(dt, world, dev)
    assert values.shape[0] == pipe.sizes.num_voxels[0] and bool(torch.isfinite(values).all())
    values_timed = values.clone()  # `values` lives in the context arena until the next forward
    if one_scan:
        if rank == 0:
            print(json.dumps(one_scan_line(args, world, n, dt, sharded)))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    mesh_info = mesh_stage(pipe, synth) if rank == 0 else None
    pipelined = None
    if world == 1 and not args.no_pipelined:
        pipelined = [pipelined_rate(pipe, weights, dev, (pts, nrm, radii, bb_min, bb_max), n, max(6, 2 * args.steps), d)
                     for d in (2, 3)]  # contexts in flight

    # raw-scan rate (outside `value`: the metric takes radii as inputs): the exact 24-NN radius
    # estimate of the pre-filter (cpp/lib/preprocess.cpp:25-39) on the GPU, steady state (second call), + one step
    t_knn2 = time.perf_counter()
    synth.knn_radii_gpu(pts, 24)
    torch.cuda.synchronize()
    t_knn2 = time.perf_counter() - t_knn2
    exact = None
    if world == 1 and args.precision == "bf16x3" and not args.no_exact_f32:
        exact = exact_f32_run(weights, dev, (pts, nrm, radii, bb_min, bb_max), n, max(args.steps, 2), shapes,
                              values_timed)
    if rank == 0:
        steps = max(args.steps, 1)
        ms = dt / steps * 1e3
        stage_ms = {k: v / steps for k, v in stage_sum.items()}
        flops, launches = conv_flops(pipe.sizes, shapes)
        unet_s = stage_ms["unet"] * 1e-3
        achieved = flops / unet_s / 1e12 if unet_s > 0 else 0.0
        tr = pmc_traffic(n, args.precision) if args.density_variance == 1.0 else None
        dtype, peak, kname, peak_note = PRECISION_INFO[args.precision]
        out = {
            "metric": "input points/sec to signed implicit values",
            "value": job_value(world, n, steps, dt),
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": dtype,
            "data": "synthetic",
            "config": {"workload": "%s: %d-point scan-like synthetic cloud per GPU%s, radii = 24-NN, "
                                   "5 grid levels, UNet5 default.yaml widths, seeded random weights"
                                   % ("C5" if args.precision == "f16" and args.density_variance >= 10 else "C3", n,
                                      " (density variance %gx)" % args.density_variance
                                      if args.density_variance != 1.0 else ""),
                       "precision": args.precision,
                       "points_per_gpu": n,
                       "voxels": [int(v) for v in pipe.sizes.num_voxels],
                       "pairs": [int(v) for v in pipe.sizes.num_pairs],
                       "agg_pairs": int(pipe.sizes.num_agg_pairs),
                       "parallelism": "one scan per GPU, no collective on the data path",
                       "stage_ms": {k: round(v, 3) for k, v in stage_ms.items()},
                       "untimed_knn24_radii_ms": round(t_knn * 1e3, 1),
                       "untimed_raw_scan": {"knn24_radii_ms_steady": round(t_knn2 * 1e3, 1),
                                            "points_per_s_with_knn_radii": n / (t_knn2 + dt / steps),
                                            "note": "pre-filter radius estimate + one step; radii are inputs of the "
                                                    "metric, this is the rate from a raw scan"},
                       "untimed_mesh_stage": mesh_info,
                       "untimed_pipelined_two_contexts": pipelined,
                       "untimed_exact_f32_kernel": exact},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak,
                         "unit": "TFLOP/s", "frac": achieved / peak, "peak_note": peak_note,
                         "executed_bf16_mfma_tflops": 6 * achieved if args.precision == "bf16x3" else None,
                         "ratio_to_f32_input_mfma_peak": achieved / MFMA_F32_PEAK_TFLOPS
                         if args.precision == "bf16x3" else None,
                         "traffic": tr[0] if tr else None,
                         "traffic_note": ("HBM bytes per launch, rocprofv3 PMC passes in profiles/%s" % tr[1]) if tr else None,
                         "kernel": "%s (%d launches/step, %.3f ms avg, %.1f algorithmic "
                                   "GFLOP/step)" % (kname, launches, stage_ms["unet"] / launches, flops / 1e9)},
        }
        if not args.no_cpu_baseline and world == 1:  # reported baseline: rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, 1000)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()  # the other ranks wait for rank 0's cpu_baseline before tearing down
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

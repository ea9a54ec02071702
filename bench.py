"""Benchmark of the hot path: input points/sec from device-resident points / normals / radii to
signed implicit values (BASELINE.json metric), one process per GPU.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A step is one pass of the whole path (octree -> 5 grids -> aggregation search -> continuous conv
-> 53 sparse convs -> decoder) over one synthetic 10 M-point scan-like cloud (config C3 of
BASELINE.json / SURVEY 8(d)) that is already resident in HBM.

N = 1: the monolithic C++ driver (asr_hip_implicit_forward).
N > 1 (default --shard auto): BASELINE's metric is "10M-pt cloud, 1/2/4/8 GPU", so the headline is ONE 10 M-point
scan sharded over the N GPUs by Morton range with halo exchange per convolution over RCCL ("scaling": "strong",
asr_hip_implicit_forward_sharded).  The replica mode (one scan per GPU, no collective, "weak") is measured FIRST in the
same run and reported as config.replicas; --shard replicas makes it the headline.  The sharded path then runs under a
watchdog ($ASR_BENCH_ONE_SCAN_TIMEOUT, 420 s): should it fail or not finish on a node, the line falls back to the
replica measurement and says so.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# HIP maps streams to GPU_MAX_HW_QUEUES (default 4) hardware queues in creation order; with RCCL's own streams made first
# (torch.distributed's communicator at N > 1) the library's search stream can land on the main stream's queue and the two
# chains of the geometry build serialise (13.5 instead of 9.5 ms).  Must be set before the HIP runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

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
    "bf16x3": ("f32 via bf16x3 (24-bit operands, f32 accumulate)", MFMA_16BIT_PEAK_TFLOPS / 6.0, "k_sconv_plan16<bf16x3>",
               "bf16 dense MFMA peak (%.0f TFLOP/s) / 6: the kernel evaluates every f32 product as six bf16 MFMA "
               "products (exact three-way split, fp32-class result); f32-input MFMA peak %.1f TFLOP/s for comparison"
               % (MFMA_16BIT_PEAK_TFLOPS, MFMA_F32_PEAK_TFLOPS)),
    # the same with half the matrix instructions: per-tensor power-of-two scaling, two-way f16 split, three products
    "f16x2": ("f32 via f16x2 (22-bit operands, f32 accumulate)", MFMA_16BIT_PEAK_TFLOPS / 3.0, "k_sconv_plan16<f16x2>",
              "f16 dense MFMA peak (%.0f TFLOP/s) / 3: the kernel evaluates every f32 product as three f16 MFMA "
              "products (scaled two-way split, fp32-class result); f32-input MFMA peak %.1f TFLOP/s for comparison"
              % (MFMA_16BIT_PEAK_TFLOPS, MFMA_F32_PEAK_TFLOPS)),
    "f16": ("f16", MFMA_16BIT_PEAK_TFLOPS, "k_sconv_plan16<f16>", "f16 MFMA dense peak"),
}
SPLIT_PRODUCTS = {"bf16x3": 6, "f16x2": 3}
ARITHMETIC = {
    "f32": "f32 in / f32 out, f32-input MFMA (a bit-exact fmaf chain)",
    "bf16x3": "f32 in / f32 out; every operand split exactly into three bf16 terms (24 significant bits), six bf16 MFMA "
              "products per algorithmic product, f32 accumulate; dropped terms < 2^-24 |ab|",
    "f16x2": "f32 in / f32 out; every tensor scaled by a power of two and split into two f16 terms (11 + 11 = 22 significant "
             "bits per operand), three f16 MFMA products per algorithmic product, f32 accumulate; dropped terms <= 2^-21 |ab| "
             "-- narrower than an fp32 product, held to the 1e-5 contract against the oracle by the GPU tests",
    "f16": "f16 activations and weights in HBM, f32 accumulate (config C5)",
}


class quiet_stdout:
    """RCCL prints a version banner to C stdout when its first communicator is made; the contract is ONE json line on
    stdout.  Points fd 1 at stderr for the duration of the block and flushes C stdio before switching back."""

    def __enter__(self):
        import ctypes
        sys.stdout.flush()
        self._libc = ctypes.CDLL(None)
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        try:
            self._libc.fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


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


def conv_bytes(sizes, shapes, act_bytes=4, w_bytes=4):
    """HBM bytes of the 53 sparse convs per forward, two ways (conv1a + conv1b of a block share their gathers):
    algorithmic = SURVEY section 6 sheet, no reuse: every pair gathers its Cin-wide input row (P * Cin * act_bytes), every
    output row is written once (V_out * Cout * act_bytes), the filters are read once (K * Cin * Cout * w_bytes);
    compulsory = every input row read once (V_in * Cin * act_bytes) instead of once per pair."""
    V = list(sizes.num_voxels)
    P = list(sizes.num_pairs)
    alg = comp = 0.0
    for name, shp in shapes.items():
        if not name.endswith(".kernel") or name.startswith("cconv"):
            continue
        k, cin, cout = shp
        blk = name.split(".")[0]
        lvl = int(blk[-1])
        if k == 55:
            jobs = [(P[lvl], V[lvl], V[lvl])]                      # (pairs, input rows, output rows)
        elif blk.startswith("sparseconv_down"):
            jobs = [(V[lvl - 1], V[lvl - 1], V[lvl])] + ([(V[3], V[3], V[4])] if lvl == 3 else [])
        else:
            jobs = [(V[lvl], V[lvl + 1], V[lvl])]                  # up: one pair per fine voxel
        shared = name.endswith(".conv1b.kernel")                   # gathers of conv1a serve conv1b
        for pairs, vin, vout in jobs:
            w = k * cin * cout * w_bytes
            alg += (0 if shared else pairs * cin * act_bytes) + vout * cout * act_bytes + w
            comp += (0 if shared else vin * cin * act_bytes) + vout * cout * act_bytes + w
    return alg, comp


def geometry_bytes(sizes, n):
    """algorithmic HBM bytes of the geometry half: octree N*(12+4) read + 8 per node written;
    per grid V*8 keys + V*16 centres / sizes + P*5 + (V+1)*8 connectivity; up and down lists V*(4+1+8) each;
    aggregation search N*16 read + P_agg*(4 idx + 4 dist + 4 compat) + (V0+1)*8 written"""
    V = [int(v) for v in sizes.num_voxels]
    P = [int(p) for p in sizes.num_pairs]
    b = n * 16 + int(sizes.num_nodes) * 8
    for i in range(5):
        b += V[i] * 24 + P[i] * 5 + (V[i] + 1) * 8
        if i < 4:
            b += 2 * V[i] * 13
    b += n * 16 + int(sizes.num_agg_pairs) * 12 + (V[0] + 1) * 8
    return b


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


# integer stages of the REFERENCE's own code (cpp/lib/octree.cpp, grid.cpp compiled unmodified at survey time with
# stand-in headers, one host core of the build container) -- quoted, not re-measured: the
# reference's C++ cannot be built on the GPU box (no Eigen / libcuckoo)
REFERENCE_CODE_1_CORE = {"octree_build_ms": {"50k": 13, "1M": 642}, "grid_connectivity_5_levels_ms": {"50k": 27, "1M": 931},
                         "dual_cells_ms": {"50k": 34, "1M": 1702},
                         "note": "reference cpp/lib code, 1 host core of the build container, SURVEY.md section 6 [probe]"}


def cpu_baseline(n_sample, seed, budget_s=130.0):
    """the oracle ("port" of the reference path incl. the Open3D op semantics: sparse convs evaluated
    like Open3D's CPU op, a dense [32][55*cin] matrix per block of 32 voxels) timed on the host cores, stage by stage,
    on C1 (50 k-point sphere, whole path), C2 (1 M uniform sphere: octree + grids + a8 search + a10 continuous conv)
    and a slice of the C3 scan generator (whole path; 1 M points unless the time budget says otherwise).  `value` is
    the whole-path rate of the C3 slice.  Bounded: the legs stop being started once `budget_s` is used up."""
    import parity
    from asr_hip import synth
    from oracle import oracle as O
    O.lib()
    t_start = time.time()
    weights = synth.make_weights(1, seed=0, init="reference")
    out = {"unit": "points/s", "cores": os.cpu_count(), "kind": "port", "reference_code_1_core": REFERENCE_CODE_1_CORE}

    def whole_path(points, normals):
        radii = synth.knn_radii(points, 24)
        bb_min, bb_max = synth.bounding_box(points, 0.1)
        timings = {}
        t0 = time.time()
        with O.dense():
            parity.oracle_forward(points, normals, radii, bb_min, bb_max, weights, timings=timings)
        dt = time.time() - t0
        return dt, {k: round(v, 3) for k, v in timings.items()}

    # C1: BASELINE config 0, the reference's own CPU-runnable case
    p, q = synth.sphere_cloud(50_000, seed=0)
    dt, st = whole_path(p, q)
    out["c1_50k_sphere"] = {"points_per_s": 50_000 / dt, "seconds": round(dt, 2), "stage_s": st}
    # C3 slice first (it carries `value`), sized so that it fits what is left of the budget: the dense evaluation
    # costs ~5.4e-5 s per point on 256 threads and ~1e-3 s per point on 8
    # (the GPU box's 256 threads: 1.2e-4 s per point on C1, 7.4e-5 on the scan slice -> the full 1 M slice of SURVEY 8(d)
    # in ~75 s; an 8-core host gets ~10^5 points)
    per_point = dt / 50_000
    n3 = int(min(n_sample, max(100_000, budget_s / max(per_point, 1e-9))))
    pts, nrm = synth.scan_cloud(n3, seed=seed, device="cpu")
    dt3, st3 = whole_path(pts.numpy(), nrm.numpy())
    out["value"] = n3 / dt3
    out["c3_slice"] = {"points": n3, "points_per_s": n3 / dt3, "seconds": round(dt3, 2), "stage_s": st3}
    # C2: 1 M uniform-density points, single-scale continuous conv (a8 + a10) and the integer stages before it
    if time.time() - t_start < budget_s * 1.5:
        p, q = synth.sphere_cloud(1_000_000, seed=0)
        radii = synth.knn_radii(p, 24)
        bb_min, bb_max = synth.bounding_box(p, 0.1)
        tm = {}
        t0 = time.time()
        item = parity.oracle_geometry(p, radii, bb_min, bb_max, timings=tm)
        t1 = time.time()
        feats = np.concatenate([q, np.ones((len(p), 1), np.float32)], 1)
        imp = (item["aggregation_scale_compat"] * O.window_poly6(item["aggregation_neighbors_dist"])).astype(np.float32)
        O.continuous_conv(weights["cconv_block_in.conv1.kernel"], item["voxel_centers0"], item["voxel_sizes0"], p, feats,
                          item["aggregation_neighbors_index"], imp, item["aggregation_row_splits"], True)
        t2 = time.time()
        tm["continuous_conv"] = t2 - t1
        out["c2_1m_uniform"] = {"points_per_s_geometry_plus_cconv": 1_000_000 / (t2 - t0), "seconds": round(t2 - t0, 2),
                                "stage_s": {k: round(v, 3) for k, v in tm.items()}}
    out["sample"] = ("C3: %d-point slice of the scan generator, whole path, %.1f s (stage s: %s); C1 50 k sphere whole path "
                     "%.1f s; C2 1 M uniform sphere geometry + continuous conv; sparse convs in the dense Open3D-style "
                     "evaluation (55*cin deep per voxel); kNN radii untimed" % (n3, dt3, st3, dt))
    return out


def deviation_check(dev, inputs, precision, synth, weights=None):
    """Outside the timed region: an ARITHMETIC against the bit-exact f32-input MFMA kernel on the same cloud with
    VARIANCE-PRESERVING weights (the reference initialisers collapse the forward to a constant field of range 6e-8,
    SURVEY B.9 -- a deviation "of the range" of that says little).  The comparison with the ORACLE at this size is
    tests/test_gpu_scale.py (geometry bit for bit, the timed arithmetic end to end, full width)."""
    from asr_hip.pipeline import ImplicitPipeline
    w = weights if weights is not None else synth.make_weights(1, seed=2)
    out = {}
    for prec in ("f32", precision):
        pipe = ImplicitPipeline(w, device=dev, precision=prec)
        out[prec] = pipe.forward(*inputs).clone()
        del pipe
        torch.cuda.empty_cache()
    ref, got = out["f32"].double(), out[precision].double()
    scale = float(ref.abs().max())
    err = (got - ref).abs()
    return {"weights": "variance preserving (synth.make_weights(1, seed=2))", "range_of_values": scale,
            "max_abs_deviation": float(err.max()), "deviation_over_range": float(err.max()) / scale if scale > 0 else None,
            "fraction_within_1e-5_plus_1e-5_rel": float((err <= 1e-5 + 1e-5 * ref.abs()).double().mean())}


def timed_leg(weights, dev, inputs, n, steps, precision, shapes):
    """Outside the timed region and not part of `value`: the same cloud, `steps` steady-state steps of another
    (weights, arithmetic) pair -- same loop as the headline's"""
    from asr_hip.pipeline import ImplicitPipeline
    pipe = ImplicitPipeline(weights, device=dev, precision=precision)
    pipe.forward(*inputs)
    torch.cuda.synchronize()
    stage = dict.fromkeys(ImplicitPipeline.STAGES, 0.0)
    t0 = time.perf_counter()
    for _ in range(steps):
        v = pipe.forward(*inputs)
        for k, x in pipe.stage_ms().items():
            stage[k] += x / steps
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert bool(torch.isfinite(v).all())
    flops, launches = conv_flops(pipe.sizes, shapes)
    out = {"precision": precision, "ms_per_step": dt / steps * 1e3, "points_per_s": n * steps / dt,
           "unet_ms": stage["unet"], "geometry_wall_ms": stage["geometry_wall"],
           "unet_algorithmic_tflops": flops / (stage["unet"] * 1e-3) / 1e12 if stage["unet"] > 0 else None}
    del pipe
    torch.cuda.empty_cache()
    return out


def exact_f32_run(weights, dev, inputs, n, steps, shapes, values=None):
    """Outside the timed region and not part of `value`: the same cloud through the f32-input MFMA kernel
    (v_mfma_f32_16x16x4_f32, a bit-exact fmaf chain) -- the round-1 arithmetic, for comparison."""
    from asr_hip.pipeline import ImplicitPipeline
    pipe = ImplicitPipeline(weights, device=dev, precision="f32")
    ref = pipe.forward(*inputs)
    torch.cuda.synchronize()
    dev_rel = None
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
            "kernel": "k_sconv_mfma", "achieved_tflops": tf, "frac_of_f32_mfma_peak": tf / MFMA_F32_PEAK_TFLOPS}


def config_c2(dev, synth, steps=5):
    """BASELINE config 2 (outside `value`): single-scale continuous conv on 1 M uniform-density points, fp32 -- octree,
    aggregation search (a8) and continuous conv (a10) through the operator API, steady state"""
    from asr_hip import _lib, ops
    n = 1_000_000
    g = torch.Generator(device=dev)
    g.manual_seed(42)
    pts = (torch.rand((n, 3), generator=g, device=dev) * 2 - 1).contiguous()
    rad = torch.full((n,), 0.02, device=dev)
    feats = torch.randn((n, 4), generator=g, device=dev)
    W = torch.randn((4, 4, 4, 4, 32), generator=g, device=dev) * 0.5
    frame = _lib.frame_init(np.full(3, -1.05, np.float32), np.full(3, 1.05, np.float32))

    def once():
        nodes, leaves = ops.octree_build(frame, pts, rad, 1.0, 21)
        centers, sizes = ops.voxel_info(frame, leaves)
        idx, dist, rs, compat = ops.multi_radius_search(frame, pts, rad, centers, sizes)
        imp = ops.aggregation_importance(compat, dist)
        out = ops.continuous_conv(W, centers, sizes, pts, feats, idx, imp, rs, True)
        return leaves.shape[0], idx.shape[0], out

    once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        v, p, out = once()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    assert bool(torch.isfinite(out).all())
    return {"points": n, "voxels": int(v), "pairs": int(p), "ms": round(dt * 1e3, 3), "points_per_s": n / dt,
            "note": "octree + multi-radius search + continuous conv (operator API, fp32), uniform cube cloud, radius 0.02"}


def config_c5(dev, weights, synth, n, steps=3):
    """BASELINE config 5 (outside `value`): the mixed-density cloud (10x density variance) with f16 features"""
    from asr_hip.pipeline import ImplicitPipeline
    pts, nrm = synth.scan_cloud(n, seed=rank_seed(0), device=dev, density_variance=10.0)
    radii = synth.knn_radii_gpu(pts, 24)
    bb_min, bb_max = synth.bounding_box(pts, 0.1)
    pipe = ImplicitPipeline(weights, device=dev, precision="f16")
    pipe.forward(pts, nrm, radii, bb_min, bb_max)
    torch.cuda.synchronize()
    stage = dict.fromkeys(ImplicitPipeline.STAGES, 0.0)
    t0 = time.perf_counter()
    for _ in range(steps):
        v = pipe.forward(pts, nrm, radii, bb_min, bb_max)
        for k, x in pipe.stage_ms().items():
            stage[k] += x / steps
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    assert bool(torch.isfinite(v).all())
    out = {"points": n, "precision": "f16", "density_variance": 10.0, "ms_per_step": round(dt * 1e3, 3),
           "points_per_s": n / dt, "voxels": [int(x) for x in pipe.sizes.num_voxels],
           "agg_pairs": int(pipe.sizes.num_agg_pairs), "stage_ms": {k: round(x, 3) for k, x in stage.items()}}
    del pipe
    torch.cuda.empty_cache()
    return out


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


def one_scan_line(args, world, n, dt, sharded, extra):
    """bench line of the one-scan mode: ONE cloud over all ranks, total work fixed ("strong")"""
    steps = max(args.steps, 1)
    how = ("inside the library (asr_hip_implicit_forward_sharded, option shard_geometry = default): octree, voxel keys and "
           "up / down lists of the whole cloud on every rank; 55-slot neighbour lists, plans, aggregation search, continuous "
           "conv, 53 sparse convs and decoder on the owned rows only, one packed RCCL send/recv group per convolution on the "
           "library's stream, values all-gathered"
           if sharded.native else
           "Python reference driver (asr_hip/sharding.py): octree + grids replicated (from 4 ranks on: neighbour lists, "
           "tiling orders and aggregation for the owned voxels only), 53 sparse convs + decoder on owned rows, halo exchange "
           "per convolution (RCCL grouped send/recv), values stitched by all-reduce")
    cfg = {"workload": "C3 sharded: one %d-point scan-like synthetic cloud over %d GPU(s) by Morton range; %s"
                       % (n, world, how),
           "precision": args.precision,
           "sharded_driver": "library" if sharded.native else "python",
           "points": n,
           "voxels": sharded.num_voxels,
           "owned_rows_rank0": sharded.owned_rows,
           "halo_rows_rank0": sharded.halo_rows,
           "parallelism": "spatial sharding, %d ranks" % world}
    cfg.update(extra)
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
        "dtype": PRECISION_INFO[args.precision][0],
        "arithmetic": ARITHMETIC[args.precision],
        "data": "synthetic",
        "config": cfg,
    }


def run_one_scan(args, world, rank, dev, weights, n, barrier, synth, fused=0):
    """timed one-scan steps + an instrumented extra step (exchange bytes / time); returns the bench line (rank 0).
    fused > 0: config C4, `fused` scans of n points each in one cloud (weak scaling when fused == world)"""
    from asr_hip import sharding
    from asr_hip.sharding import ShardedImplicitPipeline
    if fused:
        pts, nrm = synth.fused_scan_cloud(fused, n, seed=rank_seed(0), device=dev)
        n = int(pts.shape[0])
    else:
        pts, nrm = synth.scan_cloud(n, seed=rank_seed(0), device=dev, density_variance=args.density_variance)
    radii = synth.knn_radii_gpu(pts, 24)
    bb_min, bb_max = synth.bounding_box(pts, 0.1)
    prec = args.precision if args.precision in SPLIT_PRODUCTS else "f32"
    sharded = ShardedImplicitPipeline(weights, dev, precision=prec)
    fallback = None
    if sharded.native:
        # The library's own driver over RCCL; should it fail on ANY rank (communicator, librccl.so), all ranks fall back
        # together to the Python reference driver of asr_hip/sharding.py
        ok = 1
        try:
            sharded.forward(pts, nrm, radii, bb_min, bb_max)
        except Exception as e:
            ok, fallback = 0, "%s: %s" % (type(e).__name__, e)
        if world > 1:
            t = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            ok = int(t.item())
        if not ok:
            del sharded
            torch.cuda.empty_cache()
            sharded = ShardedImplicitPipeline(weights, dev, precision=prec, native=False)
            fallback = fallback or "the library's sharded driver failed on another rank"
    for _ in range(args.warmup):
        sharded.forward(pts, nrm, radii, bb_min, bb_max)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        values = sharded.forward(pts, nrm, radii, bb_min, bb_max)
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0, world, dev)
    assert values.shape[0] == sharded.num_voxels[0] and bool(torch.isfinite(values).all())
    # one more step, instrumented: bytes and wall time of the halo exchanges
    st = sharded.instrumented_forward(pts, nrm, radii, bb_min, bb_max)
    extra = {"halo_bytes_per_step_rank0": {"sent": st["sent_bytes"], "received": st["recv_bytes"]},
             "exchanges_per_step": st["exchanges"],
             "exchange_ms_rank0": round(st["seconds"] * 1e3, 3),
             "exchange_note": "instrumented extra step (device synchronised around every exchange), outside the timed region"}
    if fallback:
        extra["library_driver_error"] = fallback
    try:
        extra["stage_ms_rank0_last_step"] = {k: round(v, 3) for k, v in sharded.pipe.stage_ms().items()}
    except Exception:
        pass
    line = one_scan_line(args, world, n, dt, sharded, extra) if rank == 0 else None
    if line is not None and fused:
        line["scaling"] = "weak"
        line["config"]["workload"] = ("C4: %d scan-like clouds fused into one %d-point cloud, sharded over %d GPU(s) by "
                                      "Morton range (as the C3 sharded line)" % (fused, n, world))
    del sharded
    torch.cuda.empty_cache()
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--points", type=int, default=int(os.environ.get("ASR_BENCH_POINTS", 10_000_000)))
    ap.add_argument("--cpu-sample", type=int, default=int(os.environ.get("ASR_BENCH_CPU_SAMPLE", 1_000_000)),
                    help="largest C3 slice of the cpu_baseline leg (shrunk to fit its time budget on slow hosts)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipelined", action="store_true", help="(default now; kept for the profiling scripts)")
    ap.add_argument("--pipelined", action="store_true",
                    help="also run the informational two- / three-context pipelined measurement (slower than the serial "
                         "step since round 2: the network leaves no room for a second stream)")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--precision", choices=["f32", "bf16x3", "f16x2", "f16"], default=os.environ.get("ASR_BENCH_PRECISION", "bf16x3"),
                    help="arithmetic of the 53 sparse convs: bf16x3 (default) = f32 in / f32 out, every operand split EXACTLY "
                         "into three bf16 terms (24 significant bits, nothing of an fp32 operand is dropped), six bf16 MFMAs per "
                         "product, f32 accumulate; f32 = f32-input MFMA, a bit-exact fmaf chain; f16x2 = two-way f16 split "
                         "with per-tensor power-of-two scaling, three MFMAs per product (22-bit operands: NARROWER than an "
                         "fp32 product, timed as a labelled sub-record only); f16 = f16 activations and weights (config C5)")
    ap.add_argument("--weights", choices=["variance", "reference"], default=os.environ.get("ASR_BENCH_WEIGHTS", "variance"),
                    help="weights of the TIMED run: variance (default) = variance-preserving seeded weights "
                         "(synth.make_weights(1, seed=2)): activations of O(1) magnitude in every layer, the chip under real "
                         "MFMA load; reference = the reference initialisers (models/common_torch.py:57-58), whose forward "
                         "collapses to a near-constant field (SURVEY B.9) and clocks ~10 %% higher -- timed as a sub-record")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the informational runs of BASELINE configs C2 (1 M single-scale continuous conv) and C5 "
                         "(mixed density, f16 features)")
    ap.add_argument("--no-other-legs", action="store_true",
                    help="skip the informational legs on the other weight set and on the f16x2 arithmetic (profiling)")
    ap.add_argument("--no-exact-f32", action="store_true",
                    help="skip the informational re-run of the same cloud on the f32-input MFMA kernel (profiling)")
    ap.add_argument("--density-variance", type=float, default=1.0,
                    help="10 = the mixed-density cloud of BASELINE config C5")
    ap.add_argument("--shard", choices=["auto", "replicas", "one-scan"], default="auto",
                    help="auto (default): the monolithic driver at 1 GPU; at N > 1 ONE cloud of --points points sharded "
                         "over the GPUs by Morton range with halo exchange (RCCL), strong scaling -- BASELINE's metric -- "
                         "with the replica measurement as a sub-record.  replicas: one scan per GPU, no collective, weak "
                         "scaling, as the headline.  one-scan: the sharded path even at 1 GPU")
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
        with quiet_stdout():  # (gloo and RCCL both announce themselves on stdout)
            dist.init_process_group(args.backend, rank=rank, world_size=world,
                                    device_id=dev if args.backend == "nccl" else None)
            dist.barrier()

    from asr_hip import synth
    from asr_hip.pipeline import ImplicitPipeline

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    n = args.points
    # released weights are not in the repo.  The timed run uses variance-preserving seeded weights: with the
    # reference initialisers the forward collapses to a near-constant field and the chip clocks higher than
    # under real MFMA load; that leg is timed as a sub-record.
    weights_ref = synth.make_weights(1, seed=0, init="reference")
    weights = synth.make_weights(1, seed=2) if args.weights == "variance" else weights_ref
    shapes = synth.unet5_param_shapes(1)
    mode = args.shard
    if mode == "auto":
        mode = "one-scan" if world > 1 else "replicas"
    if args.precision == "f16" and mode == "one-scan":
        mode = "replicas"  # the sharded network runs the f32-class kernels only

    # ---- --shard one-scan: the sharded path alone (also at 1 GPU) ----------------------------------
    one_scan_out, one_scan_error, c4_out = None, None, None
    if args.shard == "one-scan" and mode == "one-scan":
        one_scan_out = run_one_scan(args, world, rank, dev, weights, n, barrier, synth)
        if rank == 0:
            print(json.dumps(one_scan_out))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- one scan per rank: inputs (untimed), radii = exact 24-NN distance ------------------------
    pts, nrm = synth.scan_cloud(n, seed=rank_seed(rank), device=dev, density_variance=args.density_variance)
    t_knn = time.perf_counter()
    radii = synth.knn_radii_gpu(pts, 24)  # pre-filter row D.4 on the GPU, untimed input preparation
    torch.cuda.synchronize()
    t_knn = time.perf_counter() - t_knn
    bb_min, bb_max = synth.bounding_box(pts, 0.1)
    pipe = ImplicitPipeline(weights, device=dev, precision=args.precision)
    options_timed = pipe.ctx.non_default_options()

    def step():
        return pipe.forward(pts, nrm, radii, bb_min, bb_max)

    for _ in range(args.warmup):
        step()
    barrier()
    stage_sum = dict.fromkeys(ImplicitPipeline.STAGES, 0.0)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        values = step()
        # stage times come from hip events recorded on the stream inside the library
        for k, v in pipe.stage_ms().items():
            stage_sum[k] += v
    barrier()
    dt = time.perf_counter() - t0
    dt = max_over_ranks(dt, world, dev)
    assert values.shape[0] == pipe.sizes.num_voxels[0] and bool(torch.isfinite(values).all())
    values_timed = values.clone()  # `values` lives in the context arena until the next forward
    mesh_info = mesh_stage(pipe, synth) if rank == 0 else None
    pipelined = None
    if world == 1 and args.pipelined and not args.no_pipelined:
        pipelined = [pipelined_rate(pipe, weights, dev, (pts, nrm, radii, bb_min, bb_max), n, max(6, 2 * args.steps), d)
                     for d in (2, 3)]  # contexts in flight

    # raw-scan rate (outside `value`: the metric takes radii as inputs): the exact 24-NN radius
    # estimate of the pre-filter (cpp/lib/preprocess.cpp:25-39) on the GPU, steady state (second call), + one step
    t_knn2 = time.perf_counter()
    synth.knn_radii_gpu(pts, 24)
    torch.cuda.synchronize()
    t_knn2 = time.perf_counter() - t_knn2
    exact = other_weights = f16x2 = None
    inputs = (pts, nrm, radii, bb_min, bb_max)
    if world == 1 and args.precision in SPLIT_PRODUCTS and not args.no_exact_f32:
        exact = exact_f32_run(weights, dev, inputs, n, max(args.steps, 2), shapes, values_timed)
        exact["weights"] = args.weights
        exact["timed_arithmetic_vs_this_kernel"] = deviation_check(dev, inputs, args.precision, synth)
    if world == 1 and rank == 0 and not args.no_other_legs:
        # the other weight set on the timed arithmetic (bounds the clock / data effect of DESIGN 7.4), and the narrower
        # f16x2 arithmetic on the timed weights (a labelled sub-record: not creditable as an fp32 result)
        ow = weights_ref if args.weights == "variance" else synth.make_weights(1, seed=2)
        other_weights = timed_leg(ow, dev, inputs, n, max(args.steps // 2, 3), args.precision, shapes)
        other_weights["weights"] = ("reference initialisers uniform(-0.05, 0.05), zero bias (models/common_torch.py:57-58): the "
                                    "forward collapses to a near-constant field, SURVEY B.9" if args.weights == "variance"
                                    else "variance preserving (synth.make_weights(1, seed=2))")
        if args.precision != "f16x2" and args.precision in SPLIT_PRODUCTS:
            f16x2 = timed_leg(weights, dev, inputs, n, max(args.steps // 2, 3), "f16x2", shapes)
            f16x2["arithmetic"] = ARITHMETIC["f16x2"]
            f16x2["weights"] = args.weights
            f16x2["vs_exact_f32_kernel"] = deviation_check(dev, inputs, "f16x2", synth)
    c2 = c5 = None
    if world == 1 and rank == 0 and not args.no_other_configs and args.density_variance == 1.0:
        c2 = config_c2(dev, synth)
        if args.precision != "f16":
            c5 = config_c5(dev, weights, synth, n)
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
                                   "5 grid levels, UNet5 default.yaml widths, seeded random weights (%s)"
                                   % ("C5" if args.precision == "f16" and args.density_variance >= 10 else "C3", n,
                                      " (density variance %gx)" % args.density_variance
                                      if args.density_variance != 1.0 else "",
                                      "variance preserving, synth.make_weights(1, seed=2)" if args.weights == "variance"
                                      else "reference initialisers"),
                       "precision": args.precision,
                       # every tunable of the timed context that differs from its built-in default (environment seeds
                       # included); none of the library's options skips work
                       "non_default_options": options_timed,
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
                       "untimed_config_c2_single_scale_cconv_1m": c2,
                       "untimed_config_c5_mixed_density_f16": c5,
                       "untimed_pipelined_two_contexts": pipelined,
                       "untimed_other_weight_set": other_weights,
                       "untimed_f16x2_narrower_arithmetic": f16x2,
                       "untimed_exact_f32_kernel": exact},
            "roofline": None,
        }
        out["arithmetic"] = ARITHMETIC[args.precision]
        # the dominant kernel against BOTH roofs; `bound` = the one it sits closer to.  MFMA: algorithmic FLOP over the
        # dtype's dense peak (/ products per algorithmic product).  HBM: `achieved` = ALGORITHMIC bytes (SURVEY section 6
        # sheet, no reuse: every pair gathers its input row) over the same time; `traffic` = what the counters saw
        # (L2-miss bytes, Infinity-Cache hits included), `compulsory_bytes` = every tensor touched once.
        esz = 2 if args.precision == "f16" else 4
        wsz = {"f32": 4, "bf16x3": 6, "f16x2": 4, "f16": 2}[args.precision]
        alg_b, comp_b = conv_bytes(pipe.sizes, shapes, esz, wsz)
        r_mfma = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                  "peak_note": peak_note,
                  "executed_16bit_mfma_tflops": SPLIT_PRODUCTS[args.precision] * achieved
                  if args.precision in SPLIT_PRODUCTS else None,
                  "ratio_to_f32_input_mfma_peak": achieved / MFMA_F32_PEAK_TFLOPS
                  if args.precision in SPLIT_PRODUCTS else None}
        gbs = alg_b / unet_s / 1e9 if unet_s > 0 else 0.0
        r_hbm = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                 "algorithmic_bytes_per_launch": alg_b / launches, "compulsory_bytes_per_launch": comp_b / launches,
                 "frac_compulsory": comp_b / unet_s / 1e9 / HBM_PEAK_GBS if unet_s > 0 else 0.0,
                 "counter_traffic_GBs": (tr[0] * launches / unet_s / 1e9) if tr and unet_s > 0 else None,
                 "frac_counter_traffic": (tr[0] * launches / unet_s / 1e9 / HBM_PEAK_GBS) if tr and unet_s > 0 else None}
        head_r = dict(r_hbm if r_hbm["frac"] >= r_mfma["frac"] else r_mfma)
        head_r.update({"traffic": tr[0] if tr else None,
                       "traffic_from_profiles": ("HBM-side bytes per launch ((2*FETCH_SIZE + WRITE_SIZE)*1024, Infinity-Cache hits "
                                        "included), rocprofv3 PMC passes in profiles/%s" % tr[1]) if tr else None,
                       "kernel": "%s (%d launches/step, %.3f ms avg, %.1f algorithmic GFLOP and %.1f algorithmic / %.1f "
                                 "compulsory GB per step)" % (kname, launches, stage_ms["unet"] / launches, flops / 1e9,
                                                              alg_b / 1e9, comp_b / 1e9),
                       "mfma": r_mfma, "hbm": r_hbm})
        out["roofline"] = head_r
        gb = geometry_bytes(pipe.sizes, n)
        gms = stage_ms["geometry_wall"]
        out["roofline_geometry"] = {"bound": "hbm", "bytes": gb, "ms": round(gms, 3),
                                    "achieved": gb / (gms * 1e-3) / 1e9 if gms > 0 else 0.0, "peak": HBM_PEAK_GBS,
                                    "unit": "GB/s", "frac": gb / (gms * 1e-3) / 1e9 / HBM_PEAK_GBS if gms > 0 else 0.0,
                                    "note": "algorithmic bytes of octree + 5 grids + aggregation search (SURVEY 8(d) "
                                            "formulas, bench.geometry_bytes) over stage_ms.geometry_wall; the stage is a "
                                            "chain of latency-bound integer kernels, not a streaming pass"}
    else:
        out = None

    def final_line(head_scan, scan_error, c4):
        """rank 0: the line that is printed -- the one-scan measurement as the headline when there is one (the replicas ride
        along), else the replica line with the reason"""
        line = out
        if head_scan is not None:
            line = head_scan
            line["config"]["replicas"] = {k: out[k] for k in ("value", "ms_per_step", "scaling")}
            line["config"]["replicas"]["stage_ms"] = out["config"]["stage_ms"]
            line["config"]["replicas"]["note"] = "one %d-point scan per GPU, no collective (weak scaling), same run" % n
            line["config"]["non_default_options"] = out["config"]["non_default_options"]
            line["roofline"] = out["roofline"]
            line["roofline"]["note"] = "dominant kernel measured in the replica run of this job (one scan per GPU)"
            line["roofline_geometry"] = out["roofline_geometry"]
            if c4 is not None:
                line["config"]["c4_fused_scans"] = ({k: c4[k] for k in ("value", "ms_per_step", "scaling", "config")}
                                                    if "value" in c4 else c4)
        elif scan_error is not None:
            line["config"]["one_scan_error"] = scan_error
            line["config"]["note"] = "the sharded one-scan path failed on this node: replica measurement reported instead"
        return line

    # ---- ONE scan over all ranks (the headline at N > 1), AFTER the replica line exists: the sharded path has never run on
    # more than one physical GPU here, so each of its two runs has a watchdog of its own -- a hang inside a collective cannot
    # be recovered in-process: rank 0 prints what is finished by then (the replica line with a note if the headline run hangs,
    # the finished headline if only the C4 sub-record hangs) and every rank leaves.
    if mode == "one-scan":
        import threading
        printed = threading.Lock()  # whoever takes it prints the job's ONE line

        def watchdog_for(what, limit, head_scan):
            def give_up():
                if not printed.acquire(blocking=False):
                    return
                if rank == 0:
                    msg = "%s did not finish within %.0f s" % (what, limit)
                    if head_scan is None:
                        line = final_line(None, msg, None)
                    else:
                        line = final_line(head_scan, None, {"error": msg})
                    print(json.dumps(line), flush=True)
                sys.stdout.flush()
                os._exit(0)
            t = threading.Timer(limit, give_up)
            t.daemon = True
            t.start()
            return t

        limit = float(os.environ.get("ASR_BENCH_ONE_SCAN_TIMEOUT", 420))
        watchdog = watchdog_for("the sharded one-scan path", limit, None)
        del pipe
        torch.cuda.empty_cache()
        try:
            one_scan_out = run_one_scan(args, world, rank, dev, weights, n, barrier, synth)
        except Exception as e:  # the bench must still print a line: the replica measurement above
            one_scan_error = "%s: %s" % (type(e).__name__, e)
        # every rank must take the same decision about the next collective run: agree on "somebody failed" (still under the
        # first watchdog: a rank that died inside a collective leaves its peers waiting here)
        failed = torch.tensor([1 if one_scan_error is not None else 0], device=dev, dtype=torch.int32)
        if world > 1:
            if args.backend == "gloo":
                failed = failed.cpu()
            dist.all_reduce(failed, op=dist.ReduceOp.MAX)
        any_failed = bool(int(failed.item()))
        watchdog.cancel()
        if any_failed and one_scan_error is None:
            one_scan_error = "the sharded one-scan path failed on another rank"
            one_scan_out = None
        # config C4 (8 GPUs: eight fused scans, 80 M points, sharded): informational sub-record, its own watchdog
        if not any_failed and (world == 8 or os.environ.get("ASR_BENCH_C4")):
            limit4 = float(os.environ.get("ASR_BENCH_C4_TIMEOUT", 300))
            watchdog = watchdog_for("config C4 (fused scans, sharded)", limit4, one_scan_out)
            try:
                import copy
                a4 = copy.copy(args)
                a4.steps, a4.warmup = 1, 1
                c4_out = run_one_scan(a4, world, rank, dev, weights, n, barrier, synth, fused=world)
            except Exception as e:
                c4_out = {"error": "%s: %s" % (type(e).__name__, e)}
            watchdog.cancel()
        if not printed.acquire(blocking=False):  # a watchdog is printing: leave it to it
            time.sleep(30)
            os._exit(0)

    if rank == 0:
        out = final_line(one_scan_out, one_scan_error, c4_out)
        if not args.no_cpu_baseline and world == 1:  # reported baseline: rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, 1000)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()  # the other ranks wait for rank 0's cpu_baseline before tearing down
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

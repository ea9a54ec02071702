"""Multi-rank logic on CPU (gloo).  (1) bench.py's replica mode: per-rank scans, barrier, max-over-ranks
time, whole-job throughput.  (2) one scan sharded over the ranks (asr_hip.sharding): Morton-range
ownership, halo exchange per convolution, stitched values bit-identical to the single-rank result -- the
arithmetic comes from the oracle here (tests/sharding_oracle_backend.py), on the GPU from the C ABI."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path[:0] = [REPO, os.path.join(REPO, "adaptive-surface-reconstruction_amd")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from asr_hip import synth
    # every rank generates ITS scan; different seeds -> different clouds, same size
    pts, nrm = synth.scan_cloud(2000, seed=bench.rank_seed(rank), device="cpu")
    digest = float(pts.double().sum())
    dist.barrier()
    dt = 0.1 * (rank + 1)  # rank 1 is the slow one
    dt_job = bench.max_over_ranks(dt, world, torch.device("cpu"))
    gathered = [None] * world
    dist.all_gather_object(gathered, digest)
    # variable-length all-gather of the sharded octree build (node lists of the local octrees) and the send lists
    # derived from the symmetry of the neighbour relation
    from asr_hip import sharding
    mine = torch.arange(10 * rank, 10 * rank + 3 + 4 * rank, dtype=torch.int64)
    allk = sharding.all_gather_variable(mine)
    want = torch.cat([torch.arange(10 * r, 10 * r + 3 + 4 * r, dtype=torch.int64) for r in range(world)])
    assert torch.equal(allk, want)
    # a ring of 6 voxels, ranks own 0-2 / 3-5; rows of THIS rank only (sharded geometry): i <-> i +- 1
    owner = torch.tensor([0, 0, 0, 1, 1, 1])
    lens = torch.tensor([3 if owner[i] == rank else 0 for i in range(6)])
    rs = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(lens, 0)])
    idx = torch.tensor([j % 6 for i in range(6) if owner[i] == rank for j in (i, i - 1, i + 1)], dtype=torch.int32)
    plan = sharding.make_plan_owned(rank, world, idx, rs, owner)
    other = 1 - rank
    assert plan.recv[other].tolist() == ([3, 5] if rank == 0 else [0, 2])
    assert plan.send[other].tolist() == ([0, 2] if rank == 0 else [3, 5])
    if rank == 0:
        out.put((dt_job, bench.job_value(world, 2000, 3, dt_job), gathered))
    dist.destroy_process_group()


def test_two_rank_timing_and_sharding():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    dt_job, value, digests = out.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert abs(dt_job - 0.2) < 1e-12              # max over ranks
    assert abs(value - 2 * 2000 * 3 / 0.2) < 1e-6  # all ranks' points / that time
    assert digests[0] != digests[1]                # ranks really own different scans


def test_conv_flop_model_matches_survey():
    """bench.py's algorithmic FLOP count reproduces SURVEY section 6 (926 GFLOP at the C2 sizes)"""
    sys.path[:0] = [REPO, os.path.join(REPO, "adaptive-surface-reconstruction_amd")]
    import bench
    from asr_hip import synth

    class S:
        num_voxels = [422843, 129690, 36044, 9087, 2353]
        num_pairs = [3404263, 1012372, 273790, 67041, 16237]

    flops, launches = bench.conv_flops(S, synth.unet5_param_shapes(1))
    assert launches == 44  # 53 convs; conv1a + conv1b of the 9 blocks share a launch
    assert abs(flops / 1e9 - 926) < 1.0
    # bytes of the same sheet: 14 234 MB gathered + 1 959 MB written + 371.5 MB of filters (fp32, no reuse); compulsory
    # = every input row once instead of once per pair
    alg, comp = bench.conv_bytes(S, synth.unet5_param_shapes(1))
    assert abs(alg / 1e6 - 16564.5) < 0.01 * 16564.5, alg
    assert comp < 0.35 * alg and comp > 1959e6 + 371.5e6


def test_one_scan_bench_line_shape():
    """the line bench.py prints for N > 1 (one scan sharded over the GPUs): "strong" scaling, the halo traffic and
    exchange time of rank 0, which driver ran -- assembled from a stand-in for the sharded pipeline (no GPU here)"""
    sys.path[:0] = [REPO, os.path.join(REPO, "adaptive-surface-reconstruction_amd")]
    import argparse
    import json
    import bench

    class Sharded:
        native = True
        num_voxels = [1000, 300, 90, 30, 10]
        owned_rows = [500, 150, 45, 15, 5]
        halo_rows = {"nb0": 40, "nb1": 20, "nb2": 10, "nb3": 5, "nb4": 2}

    args = argparse.Namespace(steps=4, warmup=1, precision="f16x2")
    extra = {"halo_bytes_per_step_rank0": {"sent": 123, "received": 456}, "exchanges_per_step": 54, "exchange_ms_rank0": 1.5}
    line = json.loads(json.dumps(bench.one_scan_line(args, 2, 10000, 0.2, Sharded(), extra)))
    assert line["scaling"] == "strong" and line["n_gpus"] == 2 and line["unit"] == "points/s"
    assert abs(line["value"] - 10000 * 4 / 0.2) < 1e-6 and abs(line["ms_per_step"] - 50.0) < 1e-9
    cfg = line["config"]
    assert cfg["halo_bytes_per_step_rank0"] == {"sent": 123, "received": 456} and cfg["exchange_ms_rank0"] == 1.5
    assert cfg["sharded_driver"] == "library" and cfg["owned_rows_rank0"][0] == 500 and "workload" in cfg
    assert "f16x2" in line["dtype"] and "arithmetic" in line


# ---- one scan across ranks: partition + halo exchange ----------------------------------
def _shard_worker(rank, world, port, n_points, out):
    sys.path[:0] = [REPO, os.path.join(REPO, "adaptive-surface-reconstruction_amd"), os.path.join(REPO, "tests")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import parity
    from asr_hip import sharding, synth
    from sharding_oracle_backend import OracleBackend, geometry_from_oracle
    p, q = synth.scan_cloud(n_points, seed=77, device="cpu", density_variance=10.0)
    pts, nrm = p.numpy(), q.numpy()
    rad = synth.knn_radii(pts, 24)
    bb = synth.bounding_box(pts, 0.1)
    weights = synth.make_weights(4, seed=9)
    item = parity.oracle_geometry(pts, rad, *bb)
    geom = geometry_from_oracle(item)
    wt = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in weights.items()}
    net = sharding.ShardedNetwork(OracleBackend(), geom, wt, rank, world)
    values, rows = net.forward(torch.from_numpy(pts), torch.from_numpy(nrm), torch.from_numpy(rad), bb)
    full = net.stitch(values, rows)
    info = {"rank": rank, "owned": [int(r.numel()) for r in net.rows], "halo": net.halo_rows(),
            "owner0": net.owner[0].numpy().copy()}
    if rank == 0:
        ref = parity.oracle_network(item, pts, nrm, weights)["values"]
        info["equal"] = bool(np.array_equal(full.numpy(), ref))
        info["max_abs_diff"] = float(np.abs(full.numpy() - ref).max())
        info["v"] = [len(item["voxel_sizes%d" % i]) for i in range(5)]
        info["codes_sorted_by_owner"] = bool(np.all(np.diff(
            net.owner[0].numpy()[np.argsort(sharding.normalized_codes(geom["voxel_keys0"]).numpy())]) >= 0))
    out.put(info)
    dist.barrier()
    dist.destroy_process_group()


def _run_sharded(world, n_points):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, n_points, out)) for r in range(world)]
    for p in procs:
        p.start()
    infos = sorted([out.get(timeout=600) for _ in range(world)], key=lambda d: d["rank"])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return infos


def test_one_scan_sharded_over_two_ranks_equals_single_rank():
    infos = _run_sharded(2, 20000)
    r0 = infos[0]
    assert r0["equal"], r0["max_abs_diff"]            # bit-identical stitched values
    assert r0["codes_sorted_by_owner"]                # ownership = contiguous ranges of the Morton order
    for lvl in range(5):                              # the owned sets partition every level
        assert sum(i["owned"][lvl] for i in infos) == r0["v"][lvl]
    assert np.array_equal(infos[0]["owner0"], infos[1]["owner0"])  # every rank derives the same ownership
    # a real exchange happened on the fine levels, and it is a boundary, not a bulk transfer
    for i in infos:
        assert i["halo"]["nb", 0] > 0 and i["halo"]["nb", 0] < 0.25 * i["owned"][0]
    # a cut of the Morton order splits at most one sibling group per level: the transition halos are tiny
    assert sum(i["halo"]["down", 0] + i["halo"]["up", 0] for i in infos) <= 16
    # balanced by pair count: neither rank owns more than 60 % of the grid-0 voxels
    assert max(i["owned"][0] for i in infos) < 0.6 * r0["v"][0]


def test_one_scan_sharded_over_three_ranks_small_cloud():
    """more ranks than some coarse levels have voxels: empty ownership on coarse grids still works"""
    infos = _run_sharded(3, 3000)
    assert infos[0]["equal"], infos[0]["max_abs_diff"]

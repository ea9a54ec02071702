"""world_size-2 test of bench.py's multi-rank logic on CPU (gloo): per-rank scans, barrier,
max-over-ranks time, whole-job throughput.  The data path itself has no collective."""
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

"""GPU test of the one-scan sharding on ONE GPU: two processes share cuda:0, the
halo exchange goes over gloo (staged through the host; on a multi-GPU node the same code runs over
RCCL with `backend="nccl"`).  Every rank runs the HIP kernels through the C ABI on the rows it owns; the
stitched values must be bit-identical to the single-process ImplicitPipeline."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_points, channel_div, precision, out, fused=0, sharded_geometry=False):
    sys.path[:0] = [REPO, os.path.join(REPO, "adaptive-surface-reconstruction_amd"), os.path.join(REPO, "tests")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from asr_hip import sharding, synth
    from asr_hip.pipeline import ImplicitPipeline
    if fused:  # config C4: `fused` disjoint scans in one cloud
        pts, nrm = synth.fused_scan_cloud(fused, n_points // fused, seed=70, device=dev)
    else:
        pts, nrm = synth.scan_cloud(n_points, seed=55, device=dev, density_variance=10.0)
    rad = synth.knn_radii_gpu(pts, 24)
    bb = synth.bounding_box(pts, 0.1)
    weights = synth.make_weights(channel_div, seed=6)
    sp = sharding.ShardedImplicitPipeline(weights, dev, precision=precision, native=False)  # the Python reference driver
    sp.sharded_geometry = bool(sharded_geometry)
    full = sp.forward(pts, nrm, rad, bb[0], bb[1])
    info = {"rank": rank, "owned": [int(r.numel()) for r in sp.net.rows], "halo": sp.net.halo_rows()}
    if rank == 0:
        single = ImplicitPipeline(weights, device=dev, precision=precision).forward(pts, nrm, rad, bb[0], bb[1])
        info["equal"] = bool(torch.equal(full, single))
        info["max_abs_diff"] = float((full - single).abs().max())
        info["v0"] = int(single.shape[0])
    out.put(info)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_points,channel_div,precision",
                         [(2, 30000, 2, "f32"), (3, 8000, 1, "f32"), (2, 30000, 1, "bf16x3"), (3, 8000, 2, "bf16x3"),
                          (2, 30000, 1, "f16x2"), (3, 8000, 2, "f16x2")])
def test_sharded_values_equal_single_process(gpu, world, n_points, channel_div, precision):
    """precision: the arithmetic of the 53 sparse convs on both sides (exact f32 MFMA / plan-driven bf16x3 or f16x2 kernel
    with a plan per rank's row list): per row the same kernel arithmetic, so the stitched values are bit-identical either
    way (f16x2: the per-tensor scale comes from a MAX all-reduce of the ranks' running maxima = the monolithic one)"""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_points, channel_div, precision, out))
             for r in range(world)]
    for p in procs:
        p.start()
    infos = sorted([out.get(timeout=900) for _ in range(world)], key=lambda d: d["rank"])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    r0 = infos[0]
    assert r0["equal"], r0["max_abs_diff"]
    assert sum(i["owned"][0] for i in infos) == r0["v0"]
    assert all(i["halo"]["nb", 0] > 0 for i in infos)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_fused_multi_scan_cloud_equals_single_process(gpu, world):
    """BASELINE config C4 in small: eight disjoint scans fused into one cloud, cut into Morton ranges over `world`
    processes (the cuts fall between and inside scans), f16x2 arithmetic; the stitched values equal the monolithic
    driver's bit for bit"""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 64000, 2, "f16x2", out, 8)) for r in range(world)]
    for p in procs:
        p.start()
    infos = sorted([out.get(timeout=900) for _ in range(world)], key=lambda d: d["rank"])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    r0 = infos[0]
    assert r0["equal"], r0["max_abs_diff"]
    assert sum(i["owned"][0] for i in infos) == r0["v0"]
    owned = [i["owned"][0] for i in infos]
    assert max(owned) < 1.5 * min(owned)  # equal pair counts per rank give similar row counts


@pytest.mark.parametrize("world,fused,precision", [(2, 0, "bf16x3"), (3, 8, "f16x2"), (3, 0, "f32")])
def test_sharded_geometry_equals_single_process(gpu, world, fused, precision):
    """The geometry itself sharded (asr_hip.sharding.sharded_geometry): octree and voxel keys on every rank, 55-slot
    neighbour lists (asr_hip_grid_neighbors_rows_*), tiling orders, aggregation search and continuous conv only for the
    voxels a rank owns, send lists from the symmetry of the neighbour relation.  Stitched values equal the monolithic
    driver's bit for bit."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 48000, 2, precision, out, fused, True))
             for r in range(world)]
    for p in procs:
        p.start()
    infos = sorted([out.get(timeout=900) for _ in range(world)], key=lambda d: d["rank"])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    r0 = infos[0]
    assert r0["equal"], r0["max_abs_diff"]
    assert sum(i["owned"][0] for i in infos) == r0["v0"]
    assert all(i["halo"]["nb", 0] > 0 for i in infos)


# ---- the sharded forward INSIDE the library (asr_hip_implicit_forward_sharded, round 4) ------------------------------
def _native_worker(rank, world, port, n_points, channel_div, precision, out, fused=0, shard_geometry=0):
    sys.path[:0] = [REPO, os.path.join(REPO, "adaptive-surface-reconstruction_amd"), os.path.join(REPO, "tests")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from asr_hip import shardcomm, synth
    from asr_hip.pipeline import ImplicitPipeline
    if fused:
        pts, nrm = synth.fused_scan_cloud(fused, n_points // fused, seed=70, device=dev)
    else:
        pts, nrm = synth.scan_cloud(n_points, seed=55, device=dev, density_variance=10.0)
    rad = synth.knn_radii_gpu(pts, 24)
    bb = synth.bounding_box(pts, 0.1)
    weights = synth.make_weights(channel_div, seed=6)
    pipe = ImplicitPipeline(weights, device=dev, precision=precision)
    pipe.ctx.set_option("shard_geometry", shard_geometry)
    comm = shardcomm.HostStagedComm()
    full = pipe.forward_sharded(comm, pts, nrm, rad, bb[0], bb[1]).clone()
    again = pipe.forward_sharded(comm, pts, nrm, rad, bb[0], bb[1])  # a second forward on the same context
    info = {"rank": rank, "stats": pipe.shard_stats, "repeat_equal": bool(torch.equal(full, again)),
            "host_exchanges": comm.exchanges}
    if rank == 0:
        single = ImplicitPipeline(weights, device=dev, precision=precision).forward(pts, nrm, rad, bb[0], bb[1])
        info["equal"] = bool(torch.equal(full, single))
        info["max_abs_diff"] = float((full - single).abs().max())
        info["v0"] = int(single.shape[0])
    out.put(info)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_points,channel_div,precision,fused",
                         [(2, 30000, 2, "f32", 0), (3, 8000, 1, "f32", 0), (2, 30000, 1, "bf16x3", 0),
                          (2, 30000, 1, "f16x2", 0), (3, 8000, 2, "f16x2", 0), (3, 64000, 2, "f16x2", 8),
                          (2, 1000000, 1, "f16x2", 0)])
def test_library_sharded_forward_equals_single_process(gpu, world, n_points, channel_div, precision, fused):
    """asr_hip_implicit_forward_sharded with 2 - 3 processes on ONE GPU (transport: HostStagedComm over gloo; RCCL needs
    one GPU per rank): ownership, owned row lists + plans, halo lists, packing, the MAX all-reduce of the f16x2 maxima
    and the all-gather of the values all inside libasr_hip.so.  The complete values on every rank equal the
    single-process forward bit for bit; fused = 8: BASELINE config C4 in small."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_native_worker, args=(r, world, port, n_points, channel_div, precision, out, fused))
             for r in range(world)]
    for p in procs:
        p.start()
    infos = sorted([out.get(timeout=900) for _ in range(world)], key=lambda d: d["rank"])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    r0 = infos[0]
    assert r0["equal"], r0["max_abs_diff"]
    assert all(i["repeat_equal"] for i in infos)
    assert sum(i["stats"]["owned_rows"][0] for i in infos) == r0["v0"]
    assert all(i["stats"]["halo_rows_recv"][0] > 0 and i["stats"]["exchanges"] > 40 for i in infos)
    owned = [i["stats"]["owned_rows"][0] for i in infos]
    assert max(owned) < 1.6 * min(owned)  # equal pair counts per rank give similar row counts


@pytest.mark.parametrize("world,n_points,channel_div,precision,fused",
                         [(2, 30000, 2, "f32", 0), (3, 8000, 1, "f16x2", 0), (2, 48000, 2, "bf16x3", 0),
                          (3, 64000, 2, "f16x2", 8), (4, 200000, 1, "f16x2", 0), (2, 1000000, 1, "f16x2", 0)])
def test_library_sharded_geometry_equals_single_process(gpu, world, n_points, channel_div, precision, fused):
    """option shard_geometry = 1: octree, voxel keys and up / down lists on every rank; 55-slot neighbour lists, row-group
    plans, aggregation search and continuous conv for the owned voxels only (+ the importance prefix of SURVEY B.2), send
    lists from the symmetry of the neighbour relation -- all inside libasr_hip.so.  Values equal the single-process
    forward bit for bit."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_native_worker, args=(r, world, port, n_points, channel_div, precision, out, fused, 1))
             for r in range(world)]
    for p in procs:
        p.start()
    infos = sorted([out.get(timeout=900) for _ in range(world)], key=lambda d: d["rank"])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    r0 = infos[0]
    assert r0["equal"], r0["max_abs_diff"]
    assert all(i["repeat_equal"] for i in infos)
    assert sum(i["stats"]["owned_rows"][0] for i in infos) == r0["v0"]
    assert all(i["stats"]["halo_rows_recv"][0] > 0 for i in infos)
    owned = [i["stats"]["owned_rows"][0] for i in infos]
    assert max(owned) - min(owned) <= 1  # equal voxel counts per rank


@pytest.mark.parametrize("precision", ["f32", "f16x2"])
def test_library_sharded_forward_world_one_over_rccl(gpu, precision):
    """world size 1 through the RCCL transport (ncclCommInitRank of one rank inside the library): the sharded driver with
    every row owned equals the monolithic one bit for bit and adds no exchange"""
    from asr_hip import shardcomm, synth
    from asr_hip.pipeline import ImplicitPipeline
    pts, nrm = synth.scan_cloud(40000, seed=9, device=gpu)
    rad = synth.knn_radii_gpu(pts, 24)
    bb = synth.bounding_box(pts, 0.1)
    weights = synth.make_weights(2, seed=4)
    pipe = ImplicitPipeline(weights, device=gpu, precision=precision)
    single = pipe.forward(pts, nrm, rad, bb[0], bb[1]).clone()
    comm = shardcomm.RcclComm(pipe.ctx)
    full = pipe.forward_sharded(comm, pts, nrm, rad, bb[0], bb[1])
    assert torch.equal(full, single)
    assert pipe.shard_stats["exchanges"] == 0 and pipe.shard_stats["owned_rows"][0] == single.shape[0]
    comm.close()


def test_partial_build_of_a_sharded_forward_is_refused_by_the_standalone_entry_points(gpu):
    """After a sharded forward with per-rank geometry the context holds ONE RANK's neighbour lists, plans and aggregation
    rows; asr_hip_implicit_network / asr_hip_implicit_aggregate on that context would silently compute wrong values --
    they must fail until the next asr_hip_implicit_build (round-4 advisor finding).  One process plays rank 0 of a
    world of two with a transport that moves nothing (the halo rows keep stale values: only the refusal is checked)."""
    import ctypes
    sys.path[:0] = [REPO, os.path.join(REPO, "adaptive-surface-reconstruction_amd")]
    from asr_hip import _lib, synth
    from asr_hip._lib import AsrHipError
    from asr_hip.pipeline import ImplicitPipeline
    dev = torch.device("cuda:0")
    pts, nrm = synth.scan_cloud(6000, seed=5, device=dev)
    rad = synth.knn_radii_gpu(pts, 24)
    bb = synth.bounding_box(pts, 0.1)
    pipe = ImplicitPipeline(synth.make_weights(4, seed=1), device=dev, precision="bf16x3")
    ex = _lib.SHARD_EXCHANGE_FN(lambda *a: 0)
    ar = _lib.SHARD_ALLREDUCE_FN(lambda *a: 0)
    comm = _lib.ShardComm(None, 0, 2, ex, ar, _lib.SHARD_EXCHANGE_MAX_FN())

    class C:
        def handle(self):
            return ctypes.byref(comm)

    pipe.forward_sharded(C(), pts, nrm, rad, bb[0], bb[1])
    with pytest.raises(AsrHipError, match="sharded"):
        pipe.network(pts, nrm, bb[0], bb[1])
    with pytest.raises(AsrHipError, match="sharded"):
        pipe.aggregate(pts, nrm, bb[0], bb[1])
    # a fresh build clears the flag
    pipe.build(pts, rad, bb[0], bb[1])
    v = pipe.network(pts, nrm, bb[0], bb[1])
    ref = ImplicitPipeline(synth.make_weights(4, seed=1), device=dev, precision="bf16x3").forward(pts, nrm, rad, bb[0], bb[1])
    assert torch.equal(v, ref)


# ---- no rank fails alone (round 6): the ranks agree on the outcome of the build and of the network preparation ---------
def _failing_worker(rank, world, port, how, bad_rank, out):
    sys.path[:0] = [REPO, os.path.join(REPO, "adaptive-surface-reconstruction_amd"), os.path.join(REPO, "tests")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from asr_hip import shardcomm, synth
    from asr_hip._lib import AsrHipError
    from asr_hip.pipeline import ImplicitPipeline
    pts, nrm = synth.scan_cloud(30000, seed=55, device=dev)
    rad = synth.knn_radii_gpu(pts, 24)
    bb = synth.bounding_box(pts, 0.1)
    weights = synth.make_weights(2, seed=6)
    pipe = ImplicitPipeline(weights, device=dev, precision="bf16x3")
    comm = shardcomm.HostStagedComm()
    good = pipe.forward_sharded(comm, pts, nrm, rad, bb[0], bb[1]).clone()  # a healthy forward first
    if rank == bad_rank:
        if how == "arena":
            pipe.ctx.set_option("arena_cap_mb", 1)  # no arena may grow any more: the first new slab fails
        else:
            pipe.ctx.set_option("inject_failure", 1 if how == "build" else 2)
    if how == "arena":  # a cloud whose arrays do not fit the 256 MB slabs the first forward left behind
        pts, nrm = synth.scan_cloud(1500000, seed=56, device=dev)
        rad = synth.knn_radii_gpu(pts, 24)
        bb = synth.bounding_box(pts, 0.1)
    info = {"rank": rank, "error": None}
    try:
        pipe.forward_sharded(comm, pts, nrm, rad, bb[0], bb[1])
    except AsrHipError as e:
        info["error"] = str(e)
    # the context and the communicator survive: the next forward (fault removed) is healthy and equals the first one
    pipe.ctx.set_option("inject_failure", 0)
    pipe.ctx.set_option("arena_cap_mb", 0)
    if how != "arena":
        again = pipe.forward_sharded(comm, pts, nrm, rad, bb[0], bb[1])
        info["recovered"] = bool(torch.equal(again, good))
    else:
        again = pipe.forward_sharded(comm, pts, nrm, rad, bb[0], bb[1])
        info["recovered"] = bool(torch.isfinite(again).all())
    out.put(info)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("how,bad_rank", [("build", 1), ("network", 0), ("arena", 1)])
def test_library_sharded_forward_one_failing_rank_makes_every_rank_return(gpu, how, bad_rank):
    """A rank that fails between two collectives would leave its peers inside ncclSend / ncclRecv.  The sharded forward does
    everything that can fail locally BEFORE the first exchange -- the build, then the preparation pass of the network
    (weights, packing, every allocation, staging buffers) -- and the ranks agree on each part's outcome with one MAX
    all-reduce of a status word (asr_shard_agree).  Here one of two ranks fails in its build (injected), in its network
    preparation (injected) or by exhausting its arena budget (option arena_cap_mb on a cloud that needs new slabs): BOTH
    ranks return an error (the healthy one ASR_HIP_EPEER, code 6) instead of hanging, and the next forward works."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, how, bad_rank, out)) for r in range(2)]
    for p in procs:
        p.start()
    infos = sorted([out.get(timeout=600) for _ in range(2)], key=lambda d: d["rank"])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    bad, other = infos[bad_rank], infos[1 - bad_rank]
    assert bad["error"] is not None and other["error"] is not None, infos
    assert "error 6" in other["error"] and "another rank failed" in other["error"], other
    assert ("injected failure" in bad["error"]) if how != "arena" else ("allocation failed" in bad["error"]), bad
    assert bad["recovered"] and other["recovered"], infos

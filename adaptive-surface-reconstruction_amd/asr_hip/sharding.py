"""One scan across the GPUs of a node (BASELINE config C4): spatial sharding of the
network half of the hot path with halo exchange of boundary feature rows.

The reference has no multi-device path at all (cpp/lib/asr.cpp:161-163 creates CPU tensors); what
defines the halo is the stencil of its operators: one face ring per 55-slot convolution on the same /
child / parent level (cpp/lib/grid.cpp:99-170) and the parent <-> children coupling of the transitions
(cpp/lib/grid.cpp:206-242).

Scheme (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI):
  * the octree and the five grids are built on every rank (integer work, ~15 % of a 10 M-point step) so
    that every rank can derive ownership and all send / receive lists locally, without any negotiation;
  * grid-0 voxels are cut into `world` contiguous ranges of the Morton order of their cells (location
    codes sort level-major, so the keys are first normalised to level 21) with equal numbers of
    neighbour pairs; a coarser voxel belongs to the owner of its first child (carried voxels keep their
    owner), so every level is partitioned by the same space-filling curve;
  * a rank computes only the rows it owns: aggregation search + continuous conv for its grid-0 voxels,
    every sparse conv through the kernel's row list (asr_sparse_conv_args.row_perm / num_out), the
    decoder for its voxels.  Activations live in full-size buffers addressed by global row index;
  * before a convolution reads a buffer, the rows of that buffer that the rank's output rows reference
    but other ranks own (the halo, a few thousand boundary voxels) are exchanged point to point
    (batched isend / irecv: RCCL grouped send/recv) -- per convolution, no redundant ring compute;
  * the per-rank values are stitched with one all-reduce of a zero-initialised [V0, 2] array.
Every row is computed by the same kernel from the same inputs in the same summation order as on one
GPU, so the stitched result is bit-identical to the single-GPU result (tested).

This module is device agnostic (torch tensor algebra + a small `backend` object doing the arithmetic):
the GPU backend wraps the C ABI (HipBackend); the world-size-2 gloo test on CPU plugs in the oracle.
"""
import torch
import torch.distributed as dist

NUM_GRIDS = 5


# ---- ownership ---------------------------------------------------------------------------------------
def key_levels(keys):
    """level of each location code (cpp/lib/octreebase.h:41-57): keys are uint64 bit patterns held in
    int64; a level-21 key has bit 63 set and shows up negative"""
    lev = torch.zeros_like(keys)
    for l in range(1, 21):
        lev += (keys >= (1 << (3 * l))).to(keys.dtype)
    return torch.where(keys < 0, torch.full_like(keys, 21), lev)


def normalized_codes(keys):
    """Morton code of each voxel's minimum corner at level 21 (a space-filling order across levels;
    the location codes themselves sort level-major)"""
    lev = key_levels(keys)
    marker = torch.ones_like(keys) << (3 * lev)
    return (keys ^ marker) << (3 * (21 - lev))


def partition_level0(keys0, row_splits0, world):
    """owner[v] for the grid-0 voxels: contiguous ranges of the Morton order with equal pair counts"""
    order = torch.argsort(normalized_codes(keys0))
    weight = (row_splits0[1:] - row_splits0[:-1])[order].to(torch.float64)
    cum = torch.cumsum(weight, 0) - 0.5 * weight
    total = float(weight.sum())
    owner_sorted = torch.clamp((cum * (world / max(total, 1.0))).floor().to(torch.int64), 0, world - 1)
    owner = torch.empty_like(owner_sorted)
    owner[order] = owner_sorted
    return owner


def partition_by_count(keys0, world):
    """owner[v] for the grid-0 voxels without their neighbour lists: contiguous ranges of the Morton order with equal
    voxel counts (the sharded geometry build decides ownership BEFORE any list exists)"""
    order = torch.argsort(normalized_codes(keys0))
    v = keys0.shape[0]
    owner_sorted = torch.clamp((torch.arange(v, device=keys0.device, dtype=torch.int64) * world) // max(v, 1), 0, world - 1)
    owner = torch.empty_like(owner_sorted)
    owner[order] = owner_sorted
    return owner


def coarser_owner(owner, up_index, up_kernel_index, v_coarse):
    """a coarse voxel belongs to the owner of its first child (slot 0); a voxel carried to the coarser
    grid unchanged (slot 8, cpp/lib/grid.cpp:206-242) keeps its owner"""
    first = (up_kernel_index == 0) | (up_kernel_index == 8)
    out = torch.full((v_coarse,), -1, dtype=owner.dtype, device=owner.device)
    out[up_index[first].long()] = owner[first]
    return out


# ---- halo exchange -----------------------------------------------------------------------------------
class ExchangePlan:
    """rows of one input buffer to send to / receive from each peer before a consumer CSR is applied.
    send[d] / recv[s]: sorted int64 row indices (same order on both sides by construction)."""

    def __init__(self, send, recv):
        self.send = send
        self.recv = recv

    @property
    def num_recv(self):
        return sum(int(r.numel()) for r in self.recv.values())

    @property
    def num_send(self):
        return sum(int(r.numel()) for r in self.send.values())


def make_plan(rank, world, csr_index, csr_row_splits, owner_out, owner_in):
    """Halo of the consumer rows every rank owns: pairs (row, idx) whose two ends have different owners;
    the input row `idx` travels from owner_in[idx] to owner_out[row]."""
    if world == 1:
        return ExchangePlan({}, {})
    idx = csr_index.long()
    lens = csr_row_splits[1:] - csr_row_splits[:-1]
    row_owner = torch.repeat_interleave(owner_out, lens)
    src_owner = owner_in[idx]
    cross = row_owner != src_owner
    send, recv = {}, {}
    if bool(cross.any()):
        v_in = owner_in.shape[0]
        # unique (dst rank, input row): code = dst * v_in + row, sorted
        code = torch.unique(row_owner[cross] * v_in + idx[cross])
        dst = code // v_in
        row = code - dst * v_in
        src = owner_in[row]
        for peer in range(world):
            if peer == rank:
                continue
            m = (dst == rank) & (src == peer)
            if bool(m.any()):
                recv[peer] = row[m]
            m = (src == rank) & (dst == peer)
            if bool(m.any()):
                send[peer] = row[m]
    return ExchangePlan(send, recv)


# reporting: bytes this rank sent / received and the number of grouped exchanges since the last reset; with
# STATS["timed"] set every exchange is bracketed by device synchronisations and its wall time accumulated (an
# instrumented extra step of bench.py, never the timed region)
STATS = {"sent_bytes": 0, "recv_bytes": 0, "exchanges": 0, "seconds": 0.0, "timed": False}


def reset_stats(timed=False):
    STATS.update(sent_bytes=0, recv_bytes=0, exchanges=0, seconds=0.0, timed=bool(timed))


def make_plan_owned(rank, world, csr_index, csr_row_splits, owner):
    """The same for a 55-slot list of which only THIS rank's rows are filled (sharded geometry build).  What to receive
    follows from the rows as before; what to send follows from the symmetry of the neighbour relation (u is in the row
    of v exactly when v is in the row of u, cpp/lib/grid.cpp:99-170; checked by the parity tests): rank d needs my row v
    exactly when v has a neighbour owned by d."""
    if world == 1:
        return ExchangePlan({}, {})
    idx = csr_index.long()
    lens = csr_row_splits[1:] - csr_row_splits[:-1]
    row = torch.repeat_interleave(torch.arange(owner.shape[0], device=owner.device), lens)
    other = owner[idx]
    cross = other != rank
    send, recv = {}, {}
    if bool(cross.any()):
        v = owner.shape[0]
        peer_of = other[cross]
        need = torch.unique(peer_of * v + idx[cross])   # (owner of the input row, input row)
        give = torch.unique(peer_of * v + row[cross])   # (rank that needs my row, my row)
        for peer in range(world):
            if peer == rank:
                continue
            m = (need // v) == peer
            if bool(m.any()):
                recv[peer] = need[m] - peer * v
            m = (give // v) == peer
            if bool(m.any()):
                send[peer] = give[m] - peer * v
    return ExchangePlan(send, recv)


def all_gather_variable(t, group=None):
    """concatenation of every rank's 1-D tensor `t` (lengths differ), in rank order"""
    world = dist.get_world_size(group)
    via_host = dist.get_backend(group) == "gloo" and t.is_cuda
    work = t.cpu() if via_host else t
    n = torch.tensor([work.shape[0]], dtype=torch.int64, device=work.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    cap = max(sizes + [1])
    pad = torch.zeros(cap, dtype=work.dtype, device=work.device)
    pad[:work.shape[0]] = work
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)]).to(t.device)


def sharded_octree(frame, points, radii, rank, world, radius_scale=1.0, max_depth=21, group=None):
    """The octree of the whole cloud with the insertion spread over the ranks: every rank closes the keys of ITS share of
    the points under "all siblings, all ancestors" (no balancing: a group that is a leaf of a partial tree may be inner in
    the whole one), the node lists are all-gathered, and every rank closes and 2:1-balances the union
    (asr_hip_octree_build_parts).  Closure commutes with union, so the result is the tree of the whole cloud bit for bit."""
    from . import ops
    n = points.shape[0]
    lo, hi = rank * n // world, (rank + 1) * n // world
    local, _ = ops.octree_build_parts(frame, points[lo:hi], radii[lo:hi], radius_scale, max_depth, balance=False)
    union = all_gather_variable(local, group)
    return ops.octree_build_parts(frame, points[:0], radii[:0], radius_scale, max_depth, extra_keys=union, balance=True)


def sharded_geometry(frame, points, radii, rank, world, radius_scale=1.0, max_depth=21, group=None):
    """Geometry of one rank of the one-scan sharding through the operator API: the octree, the voxel keys of the five
    grids and the (one entry per voxel) up / down lists on every rank -- cheap integer work every rank needs to derive
    ownership -- and the expensive parts, the 55-slot neighbour lists and their MFMA tiling orders, for the OWNED voxels
    only (asr_hip_grid_neighbors_rows_*).  Returns the dict ShardedNetwork takes, with "owner<i>" and "owned_rows<i>"."""
    from . import ops
    g = {}
    if world > 1:
        nodes, leaves = sharded_octree(frame, points, radii, rank, world, radius_scale, max_depth, group)
    else:
        nodes, leaves = ops.octree_build(frame, points, radii, radius_scale, max_depth)
    keys = [leaves]
    for i in range(NUM_GRIDS - 1):
        nxt, up_idx, up_kidx, up_rs = ops.grid_coarsen(keys[i])
        keys.append(nxt)
        g["up_neighbors_index%d" % i], g["up_neighbors_kernel_index%d" % i], g["up_neighbors_row_splits%d" % i] = \
            up_idx, up_kidx, up_rs
        d_idx, d_rs, d_attr = ops.invert_neighbors_list(nxt.shape[0], up_idx, up_rs, up_kidx)
        g["down_neighbors_index%d" % i], g["down_neighbors_kernel_index%d" % i], g["down_neighbors_row_splits%d" % i] = \
            d_idx, d_attr, d_rs
    owner = [partition_by_count(keys[0], world)]
    for i in range(NUM_GRIDS - 1):
        owner.append(coarser_owner(owner[i], g["up_neighbors_index%d" % i], g["up_neighbors_kernel_index%d" % i],
                                   keys[i + 1].shape[0]))
    g["voxel_centers0"], g["voxel_sizes0"] = ops.voxel_info(frame, keys[0])
    for i in range(NUM_GRIDS):
        g["voxel_keys%d" % i] = keys[i]
        g["owner%d" % i] = owner[i]
        rows = torch.nonzero(owner[i] == rank).reshape(-1).to(torch.int32)
        idx, kidx, rs = ops.grid_neighbors_rows(keys[i], rows)
        g["neighbors_index%d" % i], g["neighbors_kernel_index%d" % i], g["neighbors_row_splits%d" % i] = idx, kidx, rs
        if rows.numel():
            # the listed rows' entries are compact and in row order: their local row splits are a gather
            rs_local = torch.cat([rs[rows.long()], rs[-1:]])
            perm_local = ops.row_groups(kidx, rs_local)
            g["owned_rows%d" % i] = rows.long()[perm_local.long()]
        else:
            g["owned_rows%d" % i] = rows.long()
    return g


def exchange(tensors, plan, group=None):
    """fills the halo rows of each tensor in `tensors` (same row space, e.g. features and importance) in
    place.  Point to point, batched: one grouped send/recv per call."""
    if not plan.send and not plan.recv:
        return
    t0 = None
    if STATS["timed"]:
        import time
        if tensors[0].is_cuda:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
    for t in tensors:
        row_bytes = t.element_size() * (t.shape[1] if t.dim() > 1 else 1)
        STATS["sent_bytes"] += row_bytes * plan.num_send
        STATS["recv_bytes"] += row_bytes * plan.num_recv
    STATS["exchanges"] += 1
    # gloo moves CPU tensors only: stage GPU tensors through the host (single-GPU multi-process tests)
    via_host = dist.get_backend(group) == "gloo" and tensors[0].is_cuda
    p2p, landing = [], []
    for t in tensors:
        for peer, rows in plan.send.items():
            buf = t.index_select(0, rows)
            p2p.append(dist.P2POp(dist.isend, buf.cpu() if via_host else buf.contiguous(), peer, group))
        for peer, rows in plan.recv.items():
            buf = torch.empty((rows.numel(),) + tuple(t.shape[1:]), dtype=t.dtype,
                              device="cpu" if via_host else t.device)
            p2p.append(dist.P2POp(dist.irecv, buf, peer, group))
            landing.append((t, rows, buf))
    for req in dist.batch_isend_irecv(p2p):
        req.wait()
    for t, rows, buf in landing:
        t.index_copy_(0, rows, buf.to(t.device))
    if t0 is not None:
        import time
        if tensors[0].is_cuda:
            torch.cuda.synchronize()
        STATS["seconds"] += time.perf_counter() - t0


# ---- the sharded forward ------------------------------------------------------------------------------
class ShardedNetwork:
    """aggregate / unet / decode (models/v0/net_definitions_torch.py:535-666) over the rows this rank
    owns.  geom: dict of torch tensors with the input_dict keys of cpp/lib/asr.cpp:159-312 for the five
    grids (plus "down_neighbors_*<i>", the inverted up lists); weights: dict state_dict name -> tensor."""

    def __init__(self, backend, geom, weights, rank, world, group=None):
        self.be, self.g, self.w = backend, geom, weights
        self.rank, self.world, self.group = rank, world, group
        self.v = [int(geom["voxel_keys%d" % i].shape[0]) for i in range(NUM_GRIDS)]
        g = geom
        self.owned_lists = "owner0" in g  # sharded_geometry(): 55-slot lists of this rank's rows only
        if self.owned_lists:
            owner = [g["owner%d" % i] for i in range(NUM_GRIDS)]
            self.rows = [g["owned_rows%d" % i] for i in range(NUM_GRIDS)]
        else:
            owner = [partition_level0(g["voxel_keys0"], g["neighbors_row_splits0"], world)]
            for i in range(NUM_GRIDS - 1):
                owner.append(coarser_owner(owner[i], g["up_neighbors_index%d" % i],
                                           g["up_neighbors_kernel_index%d" % i], self.v[i + 1]))
            self.rows = []   # owned rows per level, in the MFMA tiling order when the geometry provides one
            for i in range(NUM_GRIDS):
                mine = owner[i] == rank
                tiling = g.get("tiling%d" % i)
                if tiling is not None:
                    t = tiling.long()
                    self.rows.append(t[mine[t]])
                else:
                    self.rows.append(torch.nonzero(mine).reshape(-1))
        self.owner = owner
        self.plans = {}
        for i in range(NUM_GRIDS):
            if self.owned_lists:
                self.plans["nb", i] = make_plan_owned(rank, world, g["neighbors_index%d" % i],
                                                      g["neighbors_row_splits%d" % i], owner[i])
            else:
                self.plans["nb", i] = make_plan(rank, world, g["neighbors_index%d" % i],
                                                g["neighbors_row_splits%d" % i], owner[i], owner[i])
        for i in range(NUM_GRIDS - 1):
            # up lists: rows = grid i, inputs = grid i+1; down lists: rows = grid i+1, inputs = grid i
            self.plans["up", i] = make_plan(rank, world, g["up_neighbors_index%d" % i],
                                            g["up_neighbors_row_splits%d" % i], owner[i], owner[i + 1])
            self.plans["down", i] = make_plan(rank, world, g["down_neighbors_index%d" % i],
                                              g["down_neighbors_row_splits%d" % i], owner[i + 1], owner[i])

    # -- helpers
    def _csr(self, kind, i):
        pre = {"nb": "neighbors", "up": "up_neighbors", "down": "down_neighbors"}[kind]
        return (self.g["%s_index%d" % (pre, i)], self.g["%s_kernel_index%d" % (pre, i)],
                self.g["%s_row_splits%d" % (pre, i)])

    def _conv(self, name, x, kind, i, out_level, imp=None, normalize=False, residual=None, dual=False,
              imp_replicated=False, out=None):
        """one SpecialSparseConv (+ bias, ReLU; models/common_torch.py:95-148) on the owned rows of
        `out_level`; x: full-size input buffer whose owned rows are valid.  dual: conv1a + conv1b of a block
        (plain bank | importance weighted, normalised bank) -> (out, out_importance).  imp: importance of the
        INPUT rows (valid on the owned rows, its halo travels with the features) unless imp_replicated."""
        halo = [x] if imp is None or imp_replicated else [x, imp]
        exchange(halo, self.plans[kind, i], self.group)
        rows = self.rows[out_level]
        kw = {"out": out} if out is not None else {}  # (a column slice of a concat buffer: backends with supports_out)
        if dual:
            return self.be.sparse_conv_ab(self.w[name + ".conv1a.kernel"], self.w[name + ".conv1a.bias"],
                                          self.w[name + ".conv1b.kernel"], self.w[name + ".conv1b.bias"], x,
                                          self._csr(kind, i), rows, self.v[out_level], imp, **kw)
        return self.be.sparse_conv(self.w[name + ".kernel"], self.w[name + ".bias"], x, self._csr(kind, i), rows,
                                   self.v[out_level], imp, normalize, residual, **kw)

    def _block(self, name, x, i, imp, with_imp, imp_replicated=False, out=None):
        """SparseConvBlock (net_definitions_torch.py:253-302); out: where conv4 writes (e.g. the encoder half of the
        decoder's concat buffer)"""
        out_imp = None
        if with_imp:
            f, out_imp = self._conv(name, x, "nb", i, i, imp=imp, dual=True, imp_replicated=imp_replicated)
        else:
            f = self._conv(name + ".conv1", x, "nb", i, i)
        for k in (2, 3):
            f = self._conv(name + ".conv%d" % k, f, "nb", i, i)
        f = self._conv(name + ".conv4", f, "nb", i, i, out=out)
        return f, out_imp

    def forward(self, points, normals, radii, frame_or_bb, scale_sdf=True, feats1=None, importance=None):
        """-> (values_owned [n_owned, 2], owned_rows).  stitch() assembles the full array.
        feats1 [V0, C] / importance [P_agg]: the aggregation of the WHOLE cloud when the caller has it (the GPU
        pipeline runs its fast whole-cloud aggregation on every rank: cheaper than a generic search of the owned rows
        plus the importance prefix up to ~8 ranks); else the backend aggregates the rows this rank owns."""
        g, w, be = self.g, self.w, self.be
        V0 = self.v[0]
        own0 = self.rows[0]
        if hasattr(be, "begin_forward"):
            be.begin_forward()
        if feats1 is None:
            # ---- aggregate (net_definitions_torch.py:640-653): search + continuous conv for the owned voxels
            feats1_own, _ = be.aggregate(points, normals, radii, frame_or_bb, g["voxel_centers0"][own0],
                                         g["voxel_sizes0"][own0], w["cconv_block_in.conv1.kernel"],
                                         w["cconv_block_in.conv1.bias"])
            feats1 = torch.zeros((V0, feats1_own.shape[1]), dtype=feats1_own.dtype, device=feats1_own.device)
            feats1[own0] = feats1_own
            # SURVEY B.2 (net_definitions_torch.py:572-578 -> common_torch.py:125): encblock0 indexes the per-PAIR
            # importance array of the aggregation with grid-0 VOXEL indices, i.e. it reads the importance of the
            # first V0 pairs of the global CSR.  Those belong to the first few percent of the voxels in index
            # order; every rank searches that prefix itself (replicated, no communication).
            imp_prefix = self._importance_prefix(points, normals, radii, frame_or_bb, V0)
        else:
            if importance.shape[0] < V0:  # the reference would index out of range here (torch raises)
                raise RuntimeError("aggregation pairs (%d) < voxels (%d): reference indexing is out of range"
                                   % (importance.shape[0], V0))
            imp_prefix = importance[:V0]
        # decoder inputs [up_i | enc_i] (:617-618): both producers write into their half of one buffer
        direct = bool(getattr(be, "supports_out", False))
        c_enc = [int(w["sparseconv_encblock%d.conv4.kernel" % i].shape[2]) for i in range(NUM_GRIDS)]
        c_up = [int(w["sparseconv_up%d.conv1.kernel" % i].shape[2]) for i in range(NUM_GRIDS - 1)]
        cat = [None] * NUM_GRIDS
        if direct:
            for i in (1, 2, 3):
                cat[i] = torch.empty((self.v[i], c_up[i] + c_enc[i]), dtype=feats1.dtype, device=feats1.device)
        # ---- unet (net_definitions_torch.py:535-638)
        enc = [None] * NUM_GRIDS
        enc[0], imp = self._block("sparseconv_encblock0", feats1, 0, imp_prefix, True, imp_replicated=True)
        for i in range(1, NUM_GRIDS):
            dn = "sparseconv_down%d" % (i if i < 4 else 3)   # down3 is re-used for 3 -> 4 (:596-598)
            t, imp_d = self._conv(dn, enc[i - 1], "down", i - 1, i, imp=imp, dual=True)
            enc[i], imp = self._block("sparseconv_encblock%d" % i, t, i, imp_d, True,
                                      out=cat[i][:, c_up[i]:] if cat[i] is not None else None)
        cur = enc[4]
        for i in (3, 2, 1):
            if cat[i] is not None:
                self._conv("sparseconv_up%d.conv1" % i, cur, "up", i, i, out=cat[i][:, :c_up[i]])
                x = cat[i]
            else:
                up = self._conv("sparseconv_up%d.conv1" % i, cur, "up", i, i)
                x = torch.cat([up, enc[i]], 1)
            cur, _ = self._block("sparseconv_decblock%d" % i, x, i, None, False)
        f21 = self._conv("sparseconv_up0.conv1", cur, "up", 0, 0, residual=enc[0])  # :631-633
        code, _ = self._block("sparseconv_decblock0", f21, 0, None, False)
        # ---- decode + sdf scale (:655-666, cpp/lib/asr.cpp:324-336)
        values = be.decode(code[own0], w, g["voxel_sizes0"][own0] if scale_sdf else None)
        return values, own0

    def _importance_prefix(self, points, normals, radii, frame_or_bb, V0):
        g, be = self.g, self.be
        k = min(V0, max(1024, V0 // 8))
        while True:
            imp = be.aggregate(points, normals, radii, frame_or_bb, g["voxel_centers0"][:k], g["voxel_sizes0"][:k],
                               None, None)[1]
            if imp.shape[0] >= V0 or k >= V0:
                break
            k = min(V0, 2 * k)
        if imp.shape[0] < V0:  # the reference would index out of range here (torch raises)
            raise RuntimeError("aggregation pairs (%d) < voxels (%d): reference indexing is out of range"
                               % (imp.shape[0], V0))
        return imp[:V0].contiguous()

    def stitch(self, values_owned, owned_rows):
        """full [V0, 2] values on every rank: the owned row sets partition the rows, so a sum of
        zero-initialised arrays is exact"""
        full = torch.zeros((self.v[0], values_owned.shape[1]), dtype=values_owned.dtype,
                           device=values_owned.device)
        full[owned_rows] = values_owned
        if self.world > 1:
            via_host = dist.get_backend(self.group) == "gloo" and full.is_cuda
            buf = full.cpu() if via_host else full
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
            full = buf.to(values_owned.device)
        return full

    def halo_rows(self):
        """{(kind, level): rows received per application of that stencil} -- for reporting"""
        return {k: p.num_recv for k, p in self.plans.items()}


# ---- GPU backend: the C ABI through asr_hip.ops --------------------------------------------------------
class HipBackend:
    """precision "f32": the exact f32 MFMA kernel; "bf16x3" / "f16x2": the plan-driven 16-bit kernel (fp32-class).
    Either way a row is computed by the same kernel arithmetic as on one GPU, so sharded results equal the unsharded
    pipeline of the same precision bit for bit.  f16x2 scales every activation tensor by a power of two taken from its
    largest magnitude over ALL rows: each rank's convolution keeps the maximum of the rows it writes (out_absmax) and
    one MAX all-reduce of that scalar per convolution makes it the tensor's (group: the ranks that share the cloud)."""

    def __init__(self, device, precision="f32", group=None):
        if precision not in ("f32", "bf16x3", "f16x2"):
            raise ValueError("HipBackend: precision must be 'f32', 'bf16x3' or 'f16x2'")
        self.device = torch.device(device)
        self.precision = precision
        self.group = group
        self._amax = {}     # f16x2: storage of an activation buffer -> int32 scalar (f32 bits of its largest magnitude)
        self.supports_out = True  # sparse_conv(..., out=<column slice of a wider buffer>)
        self._packed = {}   # weight tensors (by identity) -> packed 16-bit copy; kept across forwards
        self._plans = {}    # (row splits, row list) -> (int32 row list, ConvPlan); one geometry
        self._rows32_cache = {}
        self._keep = []

    def new_geometry(self):
        """the neighbour lists changed: plans of the previous geometry are dropped (their memory, taken from the
        context's plan arena, is recycled in one go)"""
        from . import ops
        self._plans.clear()
        self._rows32_cache.clear()
        # Recycles every arena plan of this device's context (the library refuses such a plan afterwards: a second
        # backend on the same device that still holds plans gets an error, not recycled memory).  The option that
        # sends new plans to the arena is set only around this backend's own plan construction (_plan).
        ops.context(self.device).call("asr_hip_context_plan_arena_reset")

    def _pack(self, kernel, kernel_b=None):
        from . import ops
        key = (id(kernel), id(kernel_b))
        if key not in self._packed:
            self._keep.append((kernel, kernel_b))  # identities stay unique while cached
            self._packed[key] = ops.pack_filters(kernel, self.precision, kernel_b)
        return self._packed[key]

    # ---- f16x2: running maxima of the activation buffers of one forward
    def begin_forward(self):
        self._amax.clear()

    def _max_over_ranks(self, m):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            if dist.get_backend(self.group) == "gloo" and m.is_cuda:
                h = m.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.MAX, group=self.group)
                m.copy_(h)
            else:
                dist.all_reduce(m, op=dist.ReduceOp.MAX, group=self.group)
        return m

    def _amax_in(self, x):
        """scalar of an input buffer: kept by its producers, or (the aggregation's output: zero outside the rows this
        rank computed) one pass + the maximum over the ranks"""
        if self.precision != "f16x2":
            return None
        key = x.untyped_storage().data_ptr()
        if key not in self._amax:
            from . import ops
            self._amax[key] = self._max_over_ranks(ops.absmax(x))
        return self._amax[key]

    def _amax_out(self, out, fresh):
        """scalar the convolution that writes `out` updates; fresh: `out` is a new buffer (an address can be re-used
        within a forward), else a column slice of a concat buffer that another producer may have written already"""
        if self.precision != "f16x2":
            return None
        key = out.untyped_storage().data_ptr()
        if fresh or key not in self._amax:
            self._amax[key] = torch.zeros(1, dtype=torch.int32, device=self.device)
        return self._amax[key]

    def _plan(self, K, csr, rows):
        from . import ops
        idx, kidx, rs = csr
        key = (rs.data_ptr(), rows.data_ptr(), int(rows.numel()))
        if key not in self._plans:
            perm = rows.to(torch.int32).contiguous()
            ctx = ops.context(self.device)
            before = ctx.get_option("plan_arena")
            ctx.set_option("plan_arena", 1)  # this plan only: other users of the context keep their own setting
            try:
                plan = ops.ConvPlan(K, idx, kidx, rs, row_perm=perm, num_rows=rows.numel())
            finally:
                ctx.set_option("plan_arena", before)
            self._plans[key] = (perm, plan, rows)
        return self._plans[key][:2]

    def sparse_conv(self, kernel, bias, x, csr, rows, v_out, imp=None, normalize=False, residual=None, out=None):
        from . import ops
        idx, kidx, rs = csr
        fresh = out is None
        if out is None:  # only the owned rows are written; the others are never read before an exchange fills them
            out = self._new_out((v_out, kernel.shape[2]))
        if self.precision != "f32" and x.shape[1] % 4 == 0:
            perm, plan = self._plan(kernel.shape[0], csr, rows)
            m_in, m_out = self._amax_in(x), self._amax_out(out, fresh)
            r = ops.sparse_conv16(self.precision, self._pack(kernel), kernel.shape[0], kernel.shape[1], kernel.shape[2], x,
                                  idx, kidx, rs, inp_importance=imp, normalize=normalize, bias=bias, relu=True,
                                  residual=residual, out=out, return_importance=imp is not None, row_perm=perm,
                                  num_rows=rows.numel(), plan=plan, inp_absmax=m_in, out_absmax=m_out)
            if m_out is not None:
                self._max_over_ranks(m_out)
            return r
        # f32 kernel (also the fallback of the 16-bit modes for cin % 4 != 0): no running maximum is kept for `out`; a stale
        # entry of an earlier buffer at the same address must not survive (the consumer then takes one pass over the
        # buffer -- zero outside the owned rows, see _new_out -- and the maximum over the ranks)
        self._amax.pop(out.untyped_storage().data_ptr(), None)
        return ops.sparse_conv(kernel, x, idx, kidx, rs, inp_importance=imp, normalize=normalize, bias=bias,
                               relu=True, residual=residual, out=out, return_importance=imp is not None,
                               row_perm=self._rows32(rows), num_rows=rows.numel())

    def _new_out(self, shape):
        """output buffer of a convolution.  f16x2: zero-filled, because a consumer whose producer kept no running maximum
        (f32 fallback, non-fused conv1a / conv1b) scans ALL rows for the scale, and the rows this rank does not own would
        otherwise hold whatever the allocator left there (NaN / Inf would wreck the scale)"""
        if self.precision == "f16x2":
            return torch.zeros(shape, dtype=torch.float32, device=self.device)
        return torch.empty(shape, dtype=torch.float32, device=self.device)

    def _rows32(self, rows):
        key = (rows.data_ptr(), int(rows.numel()))
        if key not in self._rows32_cache:
            self._rows32_cache[key] = (rows.to(torch.int32).contiguous(), rows)
        return self._rows32_cache[key][0]

    def sparse_conv_ab(self, ka, ba, kb, bb, x, csr, rows, v_out, imp, out=None):
        """conv1a + conv1b in one launch when the fused kernel takes the widths, else two launches"""
        from . import ops
        idx, kidx, rs = csr
        ca, cb = ka.shape[2], kb.shape[2]
        fresh = out is None
        if out is None:
            out = self._new_out((v_out, ca + cb))
        fused = ca % 16 == 8 and cb == 8 and x.shape[1] % 4 == 0
        if fused and self.precision != "f32":
            perm, plan = self._plan(ka.shape[0], csr, rows)
            m_in, m_out = self._amax_in(x), self._amax_out(out, fresh)
            _, oimp = ops.sparse_conv16(self.precision, self._pack(ka, kb), ka.shape[0], ka.shape[1], ca, x, idx, kidx, rs,
                                        inp_importance=imp, normalize=True, bias=ba, relu=True, out=out,
                                        return_importance=True, row_perm=perm, num_rows=rows.numel(), cout_b=cb,
                                        bias_b=bb, plan=plan, inp_absmax=m_in, out_absmax=m_out)
            if m_out is not None:
                self._max_over_ranks(m_out)
            return out, oimp
        if fused:
            _, oimp = ops.sparse_conv(ka, x, idx, kidx, rs, inp_importance=imp, normalize=True, bias=ba, relu=True,
                                      out=out, return_importance=True, algo=2, row_perm=self._rows32(rows),
                                      num_rows=rows.numel(), filters_b=kb, bias_b=bb)
            return out, oimp
        a = self.sparse_conv(ka, ba, x, csr, rows, v_out)
        b, oimp = self.sparse_conv(kb, bb, x, csr, rows, v_out, imp, True)
        out[:, :ca] = a
        out[:, ca:] = b
        self._amax.pop(out.untyped_storage().data_ptr(), None)  # no running maximum for the assembled buffer
        return out, oimp

    def aggregate(self, points, normals, radii, frame, centers, sizes, kernel, bias):
        from . import ops
        centers, sizes = centers.contiguous(), sizes.contiguous()
        idx, dist_, rs, compat = ops.multi_radius_search(frame, points, radii, centers, sizes)
        imp = ops.aggregation_importance(compat, dist_)
        if kernel is None:
            return None, imp
        feats = torch.cat([normals, torch.ones((normals.shape[0], 1), dtype=normals.dtype, device=normals.device)], 1)
        out = ops.continuous_conv(kernel, centers, sizes, points, feats, idx, imp, rs, normalize=True, bias=bias,
                                  relu=True)
        return out, imp

    def decode(self, code, w, sizes):
        from . import ops
        return ops.decode_mlp(code.contiguous(), w["dense_decoder1.weight"], w["dense_decoder1.bias"],
                              w["dense_decoder2.weight"], w["dense_decoder2.bias"], w["dense_decoder3.weight"],
                              voxel_sizes=sizes.contiguous() if sizes is not None else None)


def geometry_from_pipeline(pipe):
    """the grid arrays of the last ImplicitPipeline.build() as the dict ShardedNetwork takes"""
    g = {}
    for i in range(NUM_GRIDS):
        for k in ("voxel_keys", "voxel_centers", "voxel_sizes", "neighbors_index", "neighbors_kernel_index",
                  "neighbors_row_splits", "tiling"):
            g["%s%d" % (k, i)] = pipe.get("%s%d" % (k, i))
        if i < NUM_GRIDS - 1:
            for k in ("up_neighbors_index", "up_neighbors_kernel_index", "up_neighbors_row_splits",
                      "down_neighbors_index", "down_neighbors_kernel_index", "down_neighbors_row_splits"):
                g["%s%d" % (k, i)] = pipe.get("%s%d" % (k, i))
    return g


class ShardedImplicitPipeline:
    """points / normals / radii (replicated on every rank) -> values[V0, 2] with the network half sharded
    over the ranks of `group`; same call signature as ImplicitPipeline.forward.

    native (default; ASR_SHARD_NATIVE=0 or native=False selects the Python reference implementation of this module):
    the sharded forward INSIDE the library (asr_hip_implicit_forward_sharded, csrc/asr_shard.hip) -- ownership, row lists,
    plans and halo lists on the device, packed point-to-point exchanges on the library's stream over RCCL
    (shardcomm.RcclComm when the group's backend is nccl or there is no group; shardcomm.HostStagedComm otherwise).  Both
    give the monolithic forward's values bit for bit (tests/test_gpu_sharded.py)."""

    def __init__(self, weights, device, group=None, point_radius_scale=1.0, octree_max_depth=21, scale_sdf=True,
                 precision="f32", native=None):
        import os
        from .pipeline import ImplicitPipeline
        if native is None:
            native = os.environ.get("ASR_SHARD_NATIVE", "1") != "0"
        self.native = bool(native)
        self.precision = precision
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.net = None
        self._comm = None
        if self.native:
            self.pipe = ImplicitPipeline(weights, device=device, point_radius_scale=point_radius_scale,
                                         octree_max_depth=octree_max_depth, scale_sdf=scale_sdf, precision=precision)
            return
        self.pipe = ImplicitPipeline(weights, device=device, point_radius_scale=point_radius_scale,
                                     octree_max_depth=octree_max_depth, scale_sdf=scale_sdf)
        self.backend = HipBackend(self.pipe.device, precision, group)
        # every rank runs the whole-cloud aggregation of the monolithic driver (search overlapped with the grid build,
        # Morton-ordered continuous conv): replicated work, but cheaper than a generic search of the owned rows plus
        # the importance prefix; whole_cloud_aggregation = False restores the owned-rows form
        self.whole_cloud_aggregation = True
        # sharded_geometry: octree + voxel keys on every rank, neighbour lists, tiling orders, aggregation search and
        # continuous conv for the owned voxels only (sharded_geometry()); None: from 4 ranks on
        env = os.environ.get("ASR_SHARDED_GEOMETRY")
        self.sharded_geometry = None if env is None else bool(int(env))

    # ---- reporting (both implementations) ----
    @property
    def num_voxels(self):
        return [int(v) for v in self.pipe.sizes.num_voxels] if self.native else list(self.net.v)

    @property
    def owned_rows(self):
        if self.native:
            return [int(x) for x in self.pipe.shard_stats["owned_rows"]]
        return [int(r.numel()) for r in self.net.rows]

    @property
    def halo_rows(self):
        """rows received per application of a stencil, by list"""
        if self.native:
            return {"nb%d" % i: int(x) for i, x in enumerate(self.pipe.shard_stats["halo_rows_recv"])}
        return {"%s%d" % k: v for k, v in self.net.halo_rows().items()}

    def _native_comm(self):
        if self._comm is None:
            from . import shardcomm
            backend = dist.get_backend(self.group) if dist.is_initialized() else None
            if backend in (None, "nccl"):
                self._comm = shardcomm.RcclComm(self.pipe.ctx, self.group)
            else:
                self._comm = shardcomm.HostStagedComm(self.group)
        return self._comm

    def instrumented_forward(self, points, normals, radii, bb_min, bb_max):
        """one more forward with every halo exchange bracketed by device synchronisations ->
        {"sent_bytes", "recv_bytes", "exchanges", "seconds"} of this rank (never part of a timed region)"""
        if self.native:
            self.pipe.ctx.set_option("shard_timing", 1)
            try:
                self.forward(points, normals, radii, bb_min, bb_max)
            finally:
                self.pipe.ctx.set_option("shard_timing", 0)
            st = self.pipe.shard_stats
            return {"sent_bytes": st["bytes_sent"], "recv_bytes": st["bytes_received"], "exchanges": st["exchanges"],
                    "seconds": st["exchange_seconds"]}
        reset_stats(timed=True)
        self.forward(points, normals, radii, bb_min, bb_max)
        st = dict(STATS)
        reset_stats()
        return {k: st[k] for k in ("sent_bytes", "recv_bytes", "exchanges", "seconds")}

    def forward(self, points, normals, radii, bb_min, bb_max):
        from . import _lib
        pipe = self.pipe
        if self.native:
            return pipe.forward_sharded(self._native_comm(), points, normals, radii, bb_min, bb_max)
        use_sg = self.sharded_geometry if self.sharded_geometry is not None else self.world >= 4
        if use_sg:
            frame = _lib.frame_init(bb_min, bb_max)
            geom = sharded_geometry(frame, points, radii, self.rank, self.world, pipe.point_radius_scale,
                                    pipe.octree_max_depth, self.group)
            self.backend.new_geometry()
            self.net = ShardedNetwork(self.backend, geom, pipe._weights, self.rank, self.world, self.group)
            values, rows = self.net.forward(points, normals, radii, frame, pipe.scale_sdf)
            return self.net.stitch(values, rows)
        pipe.ctx.set_option("build_search", 1 if self.whole_cloud_aggregation else 0)
        pipe.build(points, radii, bb_min, bb_max)
        feats1 = importance = None
        if self.whole_cloud_aggregation:
            feats1, importance = pipe.aggregate(points, normals, bb_min, bb_max)
        geom = geometry_from_pipeline(pipe)
        self.backend.new_geometry()
        self.net = ShardedNetwork(self.backend, geom, pipe._weights, self.rank, self.world, self.group)
        values, rows = self.net.forward(points, normals, radii, _lib.frame_init(bb_min, bb_max), pipe.scale_sdf,
                                        feats1=feats1, importance=importance)
        return self.net.stitch(values, rows)

// asr_shard.hip -- one scan over several GPUs inside the library (BASELINE config C4; include/asr_hip.h
// asr_hip_implicit_forward_sharded).  The reference has no multi-device path (cpp/lib/asr.cpp:161-163); what defines
// the halo is the stencil of its operators: one face ring per 55-slot convolution on the same / child / parent level
// (cpp/lib/grid.cpp:99-170), parent <-> children for the transitions (cpp/lib/grid.cpp:206-242).
//
// Every rank has built the geometry of the whole cloud (implicit_build), so ownership and all send / receive lists are
// derived locally, on the device, without negotiation:
//   * grid-0 voxels are cut into `world` contiguous ranges of the Morton order of their cells (location codes sort
//     level-major: keys are normalised to level 21 first) with equal numbers of neighbour pairs; a coarser voxel belongs
//     to the owner of its first child, a carried voxel keeps its owner -- every level is cut by the same curve;
//   * per neighbour list (5 x 55-slot, 4 up, 4 down): the owned output rows in the list's MFMA tiling order, a row-group
//     plan for exactly those rows, and per peer the input rows to send / to receive (ascending, same order on both sides);
//   * before a convolution the boundary rows of its input buffer (+ their importance) travel point to point, packed into
//     one message per peer; the f16x2 running maximum of an output buffer is MAX-reduced over the ranks; the owned values
//     are all-gathered at the end.
// Transport: asr_shard_comm (two primitives on device buffers).  The RCCL implementation loads librccl.so at run time.
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <chrono>
#include <map>
#include <vector>

#include <rocprim/rocprim.hpp>
#include <rccl/rccl.h>

#include "asr_common.h"
#include "asr_prim.h"

using namespace asr_prim;

namespace {
constexpr int BLK = 256;
constexpr int SHARD_MAX_WORLD = 64;

// level-21 Morton code of a voxel's minimum corner: a space-filling order across levels
__global__ void k_shard_codes(const u64* keys, i64 v, u64* codes, int32_t* ids) {
    const i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (i >= v) return;
    const u64 k = keys[i];
    const int lev = asr_key_level(k);
    codes[i] = (k ^ (u64(1) << (3 * lev))) << (3 * (ASR_MAX_LEVEL - lev));
    ids[i] = (int32_t)i;
}
__global__ void k_shard_weights(const int32_t* order, const i64* rs, i64 v, i64* w) {
    const i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (i >= v) return;
    const i64 q = order[i];
    w[i] = rs[q + 1] - rs[q];
}
// owner of the voxel at position i of the Morton order: equal pair counts per rank (midpoint rule)
__global__ void k_shard_owner0(const int32_t* order, const i64* w, const i64* cum, i64 v, int world, int32_t* owner) {
    const i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (i >= v) return;
    const i64 total = cum[v - 1];
    i64 o = total > 0 ? ((2 * cum[i] - w[i]) * world) / (2 * total) : 0;
    o = o < 0 ? 0 : (o > world - 1 ? world - 1 : o);
    owner[order[i]] = (int32_t)o;
}
// a coarse voxel belongs to the owner of its first child (slot 0); a carried voxel (slot 8) keeps its owner
__global__ void k_shard_coarser(const int32_t* owner_f, const int32_t* up_idx, const uint8_t* up_kidx, i64 v_f,
                                int32_t* owner_c) {
    const i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (i >= v_f) return;
    const int k = up_kidx[i];
    if (k == 0 || k == 8) owner_c[up_idx[i]] = owner_f[i];
}
__global__ void k_shard_flag_rows(const int32_t* perm, const int32_t* owner, int rank, i64 n, uint8_t* flags) {
    const i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (i >= n) return;
    flags[i] = owner[perm ? perm[i] : (int32_t)i] == rank ? 1 : 0;
}
__global__ void k_shard_owner_by_count(const int32_t* order, i64 v, int world, int32_t* owner) {
    const i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (i >= v) return;
    i64 o = (i * world) / v;
    owner[order[i]] = (int32_t)(o > world - 1 ? world - 1 : o);
}
// ---- all neighbour lists of the hierarchy in one pass each (a rank's forward paid ~330 tiny launches for the per-list form) ----
constexpr int SHARD_LISTS = 3 * ASR_NUM_GRIDS - 2;
struct ShardBatch {
    int n, world, me;
    i64 out_off[SHARD_LISTS + 1];  // prefix sums of the lists' row counts
    i64 in_off[SHARD_LISTS + 1];   // prefix sums of world * (input rows)
    const int32_t* perm[SHARD_LISTS];
    const int32_t* idx[SHARD_LISTS];
    const i64* rs[SHARD_LISTS];
    const int32_t* owner_out[SHARD_LISTS];
    const int32_t* owner_in[SHARD_LISTS];
    i64 v_in[SHARD_LISTS];
    int symmetric[SHARD_LISTS];
};
__device__ __forceinline__ int shard_list_of(const i64* off, int n, i64 t) {  // the j with off[j] <= t < off[j + 1]
    int j = 0;
    while (j + 1 < n && t >= off[j + 1]) ++j;
    return j;
}
// per output row of every list: is the row at this position of the tiling order mine (rflags), and the halo entries of row
// r: position in_off[j] + d * v_in + i in the SEND set = rank d needs my input row i, in_off[j] + s * v_in + i in the RECV
// set = I need input row i of rank s.  The sets are bit sets (first setter of a bit appends the position to the list; a
// rank's halo is a few percent of its rows, so the lists are short and sorted afterwards).
// symmetric (55-slot lists that hold this rank's rows only): u is in the row of v exactly when v is in the row of u
// (cpp/lib/grid.cpp:99-170), so rank s needs my row r exactly when r has a neighbour owned by s
__device__ __forceinline__ void shard_mark(unsigned* bits, i64 pos, int32_t* list, i64* n, i64 cap) {
    const unsigned bit = 1u << (pos & 31);
    unsigned* w = bits + (pos >> 5);
    if (*w & bit) return;  // (bits are only ever set: a cached one is the truth)
    if (atomicOr(w, bit) & bit) return;
    const i64 o = (i64)atomicAdd((unsigned long long*)n, 1ull);
    if (o < cap) list[o] = (int32_t)pos;
}
__global__ void k_shard_flags_batch(ShardBatch b, uint8_t* rflags, unsigned* F, unsigned* G, int32_t* send, int32_t* recv,
                                    i64* n_send, i64* n_recv, i64 cap) {
    const i64 t = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (t >= b.out_off[b.n]) return;
    const int j = shard_list_of(b.out_off, b.n, t);
    const i64 r = t - b.out_off[j];
    const int32_t* own = b.owner_out[j];
    rflags[t] = own[b.perm[j] ? b.perm[j][r] : (int32_t)r] == b.me ? 1 : 0;
    const int d = own[r];
    const i64 v_in = b.v_in[j];
    const i64 base = b.in_off[j];
    const int32_t* idx = b.idx[j];
    const int32_t* oin = b.owner_in[j];
    const int sym = b.symmetric[j];
    for (i64 p = b.rs[j][r], pe = b.rs[j][r + 1]; p < pe; ++p) {
        const i64 i = idx[p];
        const int s = oin[i];
        if (d == s) continue;
        if (d == b.me) {
            shard_mark(G, base + (i64)s * v_in + i, recv, n_recv, cap);
            if (sym) shard_mark(F, base + (i64)s * v_in + r, send, n_send, cap);
        }
        if (!sym && s == b.me) shard_mark(F, base + (i64)d * v_in + i, send, n_send, cap);
    }
}
__device__ __forceinline__ i64 shard_lower_bound(const int32_t* a, i64 n, i64 key) {
    i64 lo = 0, hi = n;
    while (lo < hi) {
        const i64 mid = (lo + hi) >> 1;
        if ((i64)(u32)a[mid] < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}
// boundaries of the lists (and, in the halo lists, of the peers) in the three selected-position arrays.  cnt, per list:
// [0] owned rows [1] send total [2] recv total [3] first owned row [4] first send entry [5] first recv entry
// [6 ..] world + 1 send offsets (relative), then world + 1 recv offsets
__global__ void k_shard_bounds_batch(ShardBatch b, const int32_t* sel_perm, const i64* n_perm, const int32_t* sel_send,
                                     const i64* n_send, const int32_t* sel_recv, const i64* n_recv, i64* cnt, int per) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int w1 = b.world + 1;
    const int each = 1 + 2 * w1;
    if (t >= b.n * each) return;
    const int j = t / each, k = t - j * each;
    i64* c = cnt + (size_t)j * per;
    if (k == 0) {
        const i64 a = shard_lower_bound(sel_perm, *n_perm, b.out_off[j]);
        const i64 e = shard_lower_bound(sel_perm, *n_perm, b.out_off[j + 1]);
        c[0] = e - a;
        c[3] = a;
        return;
    }
    const bool send = k <= w1;
    const int p = send ? k - 1 : k - 1 - w1;
    const int32_t* sel = send ? sel_send : sel_recv;
    const i64 n = send ? *n_send : *n_recv;
    const i64 a = shard_lower_bound(sel, n, b.in_off[j]);
    const i64 e = shard_lower_bound(sel, n, b.in_off[j] + (i64)p * b.v_in[j]);
    c[6 + (send ? 0 : w1) + p] = e - a;
    if (p == 0) c[send ? 4 : 5] = a;
    if (p == b.world) c[send ? 1 : 2] = e - a;
}
// selected positions -> row indices, in place: kind 0 = positions of the tiling orders, 1 = (peer, row) positions
__global__ void k_shard_rows_batch(ShardBatch b, int32_t* sel, const i64* count, int kind) {
    const i64 n = *count;
    for (i64 t = blockIdx.x * (i64)blockDim.x + threadIdx.x; t < n; t += (i64)gridDim.x * blockDim.x) {
        const i64 pos = (i64)(u32)sel[t];
        if (kind == 0) {
            const int j = shard_list_of(b.out_off, b.n, pos);
            const i64 r = pos - b.out_off[j];
            sel[t] = b.perm[j] ? b.perm[j][r] : (int32_t)r;
        } else {
            const int j = shard_list_of(b.in_off, b.n, pos);
            sel[t] = (int32_t)((pos - b.in_off[j]) % b.v_in[j]);
        }
    }
}
__global__ void k_shard_iota(int32_t* out, i64 n) {
    const i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)i;
}
__global__ void k_shard_flag_ge(const int32_t* rows, i64 n, int32_t k, uint8_t* flags) {
    const i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (i < n) flags[i] = rows[i] >= k ? 1 : 0;
}
__global__ void k_shard_bounds(const int32_t* sorted_owner, i64 v, int world, i64* offs) {
    const int t = threadIdx.x;
    if (t > world) return;
    i64 lo = 0, hi = v;
    while (lo < hi) {
        const i64 mid = (lo + hi) >> 1;
        if (sorted_owner[mid] < t)
            lo = mid + 1;
        else
            hi = mid;
    }
    offs[t] = lo;
}

// message layout of one peer: [n rows x row_dwords][n importance values (optional)]
struct PackDesc {
    int npeer;
    i64 row_first[SHARD_MAX_WORLD + 1];  // first row (in the concatenated row list) of each peer's message
    i64 msg_dword[SHARD_MAX_WORLD + 1];  // first dword of each peer's message in the staging buffer
};
__global__ void k_shard_pack(PackDesc d, const int32_t* rows, const unsigned* buf, i64 ld_dwords, int row_dwords,
                             const float* imp, unsigned* stage) {
    const i64 e = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    const i64 total = d.row_first[d.npeer];
    if (e >= total * (row_dwords + (imp ? 1 : 0))) return;
    const i64 nfeat = total * row_dwords;
    if (e < nfeat) {
        const i64 j = e / row_dwords;
        const int c = (int)(e % row_dwords);
        int p = 0;
        while (p + 1 < d.npeer && j >= d.row_first[p + 1]) ++p;
        stage[d.msg_dword[p] + (j - d.row_first[p]) * row_dwords + c] = buf[(i64)rows[j] * ld_dwords + c];
    } else {
        const i64 j = e - nfeat;
        int p = 0;
        while (p + 1 < d.npeer && j >= d.row_first[p + 1]) ++p;
        const i64 np = d.row_first[p + 1] - d.row_first[p];
        stage[d.msg_dword[p] + np * row_dwords + (j - d.row_first[p])] = __float_as_uint(imp[rows[j]]);
    }
}
__global__ void k_shard_unpack(PackDesc d, const int32_t* rows, unsigned* buf, i64 ld_dwords, int row_dwords, float* imp,
                               const unsigned* stage) {
    const i64 e = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    const i64 total = d.row_first[d.npeer];
    if (e >= total * (row_dwords + (imp ? 1 : 0))) return;
    const i64 nfeat = total * row_dwords;
    if (e < nfeat) {
        const i64 j = e / row_dwords;
        const int c = (int)(e % row_dwords);
        int p = 0;
        while (p + 1 < d.npeer && j >= d.row_first[p + 1]) ++p;
        buf[(i64)rows[j] * ld_dwords + c] = stage[d.msg_dword[p] + (j - d.row_first[p]) * row_dwords + c];
    } else {
        const i64 j = e - nfeat;
        int p = 0;
        while (p + 1 < d.npeer && j >= d.row_first[p + 1]) ++p;
        const i64 np = d.row_first[p + 1] - d.row_first[p];
        imp[rows[j]] = __uint_as_float(stage[d.msg_dword[p] + np * row_dwords + (j - d.row_first[p])]);
    }
}
}  // namespace

// out = the elements of `in` whose flag is set, in order; *count (device) = how many
template <class In>
static int shard_select(asr_hip_context* ctx, In in, const uint8_t* flags, int32_t* out, i64* count, size_t n) {
    size_t tb = 0;
    ASR_HIP_CHECK(ctx, rocprim::select(nullptr, tb, in, flags, out, count, n, ctx->stream));
    void* tmp = ctx->scratch.alloc(tb ? tb : 256);
    if (!tmp) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    ASR_HIP_CHECK(ctx, rocprim::select(tmp, tb, in, flags, out, count, n, ctx->stream));
    return ASR_HIP_OK;
}

static int shard_sort_u32(asr_hip_context* ctx, const int32_t* in, int32_t* out, i64 n, int end_bit) {
    size_t tb = 0;
    const unsigned* kin = reinterpret_cast<const unsigned*>(in);
    unsigned* kout = reinterpret_cast<unsigned*>(out);
    ASR_HIP_CHECK(ctx, rocprim::radix_sort_keys(nullptr, tb, kin, kout, (size_t)n, 0, end_bit, ctx->stream));
    void* tmp = ctx->scratch.alloc(tb ? tb : 256);
    if (!tmp) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    ASR_HIP_CHECK(ctx, rocprim::radix_sort_keys(tmp, tb, kin, kout, (size_t)n, 0, end_bit, ctx->stream));
    return ASR_HIP_OK;
}

// halo of one neighbour list: rows of the INPUT buffer, concatenated per peer (ascending peer, ascending row)
struct ShardCsr {
    const int32_t* perm = nullptr;  // owned output rows in the list's tiling order
    i64 num_out = 0;
    asr_conv_plan plan;
    bool has_plan = false;
    const int32_t* send_rows = nullptr;
    const int32_t* recv_rows = nullptr;
    std::vector<int> send_peer, recv_peer;  // peers with a non-empty message
    std::vector<i64> send_first, recv_first;  // first row of each of those messages, + total
};
struct asr_shard_state {
    asr_shard_comm comm;
    int rank = 0, world = 1;
    std::map<const void*, ShardCsr> csr;  // keyed by the list's row-splits pointer
    int32_t* owner[ASR_NUM_GRIDS] = {};
    bool owned_lists = false;  // sharded geometry: the 55-slot lists hold the owned rows only (halo by symmetry)
    int32_t* level_rows[ASR_NUM_GRIDS] = {};  // sharded geometry: owned rows per level, ascending
    i64 level_nrows[ASR_NUM_GRIDS] = {};
    int32_t* rows0 = nullptr;  // grid-0 rows grouped by owner (ascending rows within)
    i64 rows0_off[SHARD_MAX_WORLD + 1] = {};
    Arena* mem = nullptr;  // the context's shard arena: kept (and grown) across forwards, a state lives for one
    unsigned* stage_send = nullptr;  // (the context's staging buffers, grow-only as well)
    unsigned* stage_recv = nullptr;
    asr_shard_stats stats;
};

static int shard_stage(asr_hip_context* ctx, asr_shard_state* st, size_t dwords) {
    if (dwords > ctx->shard_stage_cap) {  // (a hipMalloc costs ~0.4 ms and a device synchronisation: never per forward)
        ASR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->shard_stage_send) (void)hipFree(ctx->shard_stage_send);
        if (ctx->shard_stage_recv) (void)hipFree(ctx->shard_stage_recv);
        ctx->shard_stage_send = ctx->shard_stage_recv = nullptr;
        ctx->shard_stage_cap = 0;
        const size_t cap = dwords + dwords / 4 + 1024;
        ASR_HIP_CHECK(ctx, hipMalloc((void**)&ctx->shard_stage_send, cap * 4));
        ASR_HIP_CHECK(ctx, hipMalloc((void**)&ctx->shard_stage_recv, cap * 4));
        ctx->shard_stage_cap = cap;
    }
    st->stage_send = ctx->shard_stage_send;
    st->stage_recv = ctx->shard_stage_recv;
    return ASR_HIP_OK;
}

void asr_shard_free(asr_shard_state* st) { delete st; }
void asr_shard_release(asr_hip_context* ctx) {
    if (ctx->shard_stage_send) (void)hipFree(ctx->shard_stage_send);
    if (ctx->shard_stage_recv) (void)hipFree(ctx->shard_stage_recv);
    ctx->shard_stage_send = ctx->shard_stage_recv = nullptr;
    ctx->shard_stage_cap = 0;
    ctx->shard_mem.release();
}

const asr_shard_stats* asr_shard_get_stats(const asr_shard_state* st) { return &st->stats; }

// the 13 neighbour lists of the hierarchy: (rows-level, input-level, arrays)
namespace {
struct ListRef {
    const int32_t* idx;
    const uint8_t* kidx;
    const i64* rs;
    const int32_t* perm;
    i64 v_out, v_in;
    int lvl_out, lvl_in, K;
};
std::vector<ListRef> shard_lists(asr_hip_context* ctx) {
    std::vector<ListRef> L;
    GridDev* g = ctx->grids;
    for (int i = 0; i < ASR_NUM_GRIDS; ++i) {
        L.push_back({g[i].nidx, g[i].nkidx, g[i].nrs, g[i].perm_nb, g[i].v, g[i].v, i, i, 55});
        if (i + 1 < ASR_NUM_GRIDS) {
            L.push_back({g[i].up_idx, g[i].up_kidx, g[i].up_rs, g[i].perm_up, g[i].v, g[i + 1].v, i, i + 1, 9});
            L.push_back({g[i].down_idx, g[i].down_kidx, g[i].down_rs, g[i].perm_down, g[i + 1].v, g[i].v, i + 1, i, 9});
        }
    }
    return L;
}
}  // namespace

// ownership of the voxels of the five grids.  by_pairs: equal numbers of neighbour pairs per rank (the 55-slot list of grid
// 0 exists); else equal voxel counts (sharded geometry: ownership is decided BEFORE any list exists), and the owned rows of
// every level are listed (ascending) for the builders that follow.
int asr_shard_ownership(asr_hip_context* ctx, const asr_shard_comm* comm, int by_pairs, asr_shard_state** out) {
    *out = nullptr;
    if (!comm || comm->world < 1 || comm->rank < 0 || comm->rank >= comm->world || comm->world > SHARD_MAX_WORLD)
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "sharded forward: bad communicator (world 1..%d)", SHARD_MAX_WORLD);
    if (comm->world > 1 && (!comm->exchange || !comm->allreduce_max_u32))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "sharded forward: the communicator lacks exchange / allreduce_max_u32");
    asr_shard_state* st = new asr_shard_state();
    struct Guard {
        asr_shard_state*& p;
        ~Guard() {
            if (p) asr_shard_free(p);
        }
    };
    asr_shard_state* guarded = st;
    Guard guard{guarded};
    st->comm = *comm;
    st->rank = comm->rank;
    st->world = comm->world;
    ctx->shard_mem.min_slab = size_t(64) << 20;
    ctx->shard_mem.reset();
    st->mem = &ctx->shard_mem;
    memset(&st->stats, 0, sizeof(st->stats));
    GridDev* g = ctx->grids;
    const int world = st->world, me = st->rank;
    (void)me;
    if (world == 1) {  // everything is mine
        guarded = nullptr;
        *out = st;
        return ASR_HIP_OK;
    }
    ASR_TRY(ensure_flags(ctx));
    ctx->scratch.reset();
    // ---- ownership ----
    const i64 V0 = g[0].v;
    for (int i = 0; i < (by_pairs ? ASR_NUM_GRIDS : 1); ++i) {
        st->owner[i] = arena_alloc<int32_t>((*st->mem), g[i].v);
        if (!st->owner[i]) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    }
    {
        u64* codes = arena_alloc<u64>(ctx->scratch, V0);
        u64* codes_s = arena_alloc<u64>(ctx->scratch, V0);
        int32_t* ids = arena_alloc<int32_t>(ctx->scratch, V0);
        int32_t* order = arena_alloc<int32_t>(ctx->scratch, V0);
        i64* w = arena_alloc<i64>(ctx->scratch, V0);
        i64* cum = arena_alloc<i64>(ctx->scratch, V0);
        if (!codes || !codes_s || !ids || !order || !w || !cum) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
        k_shard_codes<<<grid_for(V0, BLK), BLK, 0, ctx->stream>>>(g[0].keys, V0, codes, ids);
        ASR_CHECK_LAUNCH(ctx);
        // (no leaf is finer than leaf_lmax: the low 3 (21 - leaf_lmax) bits of every code are zero -- fewer radix passes)
        const int low = ctx->leaf_lmax >= 0 ? 3 * (ASR_MAX_LEVEL - std::min(ctx->leaf_lmax, ASR_MAX_LEVEL)) : 0;
        ASR_TRY((sort_pairs<u64, int32_t>(ctx, ctx->scratch, codes, codes_s, ids, order, V0, 63, low)));
        if (by_pairs) {
            k_shard_weights<<<grid_for(V0, BLK), BLK, 0, ctx->stream>>>(order, g[0].nrs, V0, w);
            ASR_CHECK_LAUNCH(ctx);
            size_t tb = 0;
            ASR_HIP_CHECK(ctx, rocprim::inclusive_scan(nullptr, tb, w, cum, (size_t)V0, rocprim::plus<i64>(), ctx->stream));
            void* tmp = ctx->scratch.alloc(tb ? tb : 256);
            if (!tmp) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
            ASR_HIP_CHECK(ctx, rocprim::inclusive_scan(tmp, tb, w, cum, (size_t)V0, rocprim::plus<i64>(), ctx->stream));
            k_shard_owner0<<<grid_for(V0, BLK), BLK, 0, ctx->stream>>>(order, w, cum, V0, world, st->owner[0]);
        } else {
            k_shard_owner_by_count<<<grid_for(V0, BLK), BLK, 0, ctx->stream>>>(order, V0, world, st->owner[0]);
        }
        ASR_CHECK_LAUNCH(ctx);
    }
    if (by_pairs) {
        for (int i = 0; i + 1 < ASR_NUM_GRIDS; ++i) {
            k_shard_coarser<<<grid_for(g[i].v, BLK), BLK, 0, ctx->stream>>>(st->owner[i], g[i].up_idx, g[i].up_kidx, g[i].v,
                                                                            st->owner[i + 1]);
            ASR_CHECK_LAUNCH(ctx);
        }
    } else {  // the owned rows of grid 0, ascending: what the aggregation search of this rank covers
        st->owned_lists = true;
        // position i of the Morton order belongs to rank floor(i * world / V0): the count needs no read-back
        auto first_of = [&](i64 r) { return (r * V0 + world - 1) / world; };
        st->level_nrows[0] = first_of(me + 1) - first_of(me);
        i64* d_n = arena_alloc<i64>(ctx->scratch, 1);
        uint8_t* fl = arena_alloc<uint8_t>(ctx->scratch, V0);
        st->level_rows[0] = arena_alloc<int32_t>((*st->mem), V0);
        if (!d_n || !fl || !st->level_rows[0]) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
        k_shard_flag_rows<<<grid_for(V0, BLK), BLK, 0, ctx->stream>>>(nullptr, st->owner[0], me, V0, fl);
        ASR_CHECK_LAUNCH(ctx);
        ASR_TRY(shard_select(ctx, rocprim::counting_iterator<int32_t>(0), fl, st->level_rows[0], d_n, (size_t)V0));
    }
    guarded = nullptr;
    *out = st;
    return ASR_HIP_OK;
}
// owners of the coarser grids (a merged voxel belongs to the owner of its first child), once those grids exist
int asr_shard_ownership_coarser(asr_hip_context* ctx, asr_shard_state* st) {
    if (st->world == 1 || st->owner[1]) return ASR_HIP_OK;
    GridDev* g = ctx->grids;
    for (int i = 0; i + 1 < ASR_NUM_GRIDS; ++i) {
        st->owner[i + 1] = arena_alloc<int32_t>((*st->mem), g[i + 1].v);
        if (!st->owner[i + 1]) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
        k_shard_coarser<<<grid_for(g[i].v, BLK), BLK, 0, ctx->stream>>>(st->owner[i], g[i].up_idx, g[i].up_kidx, g[i].v,
                                                                        st->owner[i + 1]);
        ASR_CHECK_LAUNCH(ctx);
    }
    return ASR_HIP_OK;
}
const int32_t* asr_shard_owner(const asr_shard_state* st, int level) { return st->owner[level]; }

// the rest of the shard state once the neighbour lists and their tiling orders exist: owned rows in tiling order, plans,
// halo lists, stitch lists
int asr_shard_lists(asr_hip_context* ctx, asr_shard_state* st, int want_plans) {
    GridDev* g = ctx->grids;
    const std::vector<ListRef> L = shard_lists(ctx);
    const int world = st->world, me = st->rank;
    const i64 V0 = g[0].v;
    if (world == 1) {  // everything is mine: the monolithic driver's own orders and plans
        for (const ListRef& l : L) {
            ShardCsr c;
            c.perm = l.perm;
            c.num_out = l.v_out;
            auto it = ctx->conv_plans.find(l.rs);
            if (it != ctx->conv_plans.end()) {
                c.plan = it->second;
                c.has_plan = true;
            }
            c.send_first.push_back(0);
            c.recv_first.push_back(0);
            st->csr[l.rs] = c;
        }
        for (int i = 0; i < ASR_NUM_GRIDS; ++i) st->stats.owned_rows[i] = g[i].v;
        return ASR_HIP_OK;
    }
    ctx->scratch.reset();
    // ---- per list: owned rows in tiling order, halo lists (device passes first, ONE read-back of all sizes) ----
    const int nl = (int)L.size();
    if (nl != SHARD_LISTS) ASR_FAIL(ctx, ASR_HIP_ELOGIC, "sharded forward: %d neighbour lists?", nl);
    // device counters per list: see k_shard_bounds_batch; + the stitch offsets
    const int per = 6 + 2 * (world + 1);
    i64* d_cnt = arena_alloc<i64>(ctx->scratch, (size_t)nl * per + world + 1 + 3);
    if (!d_cnt) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    ASR_HIP_CHECK(ctx, hipMemsetAsync(d_cnt, 0, ((size_t)nl * per + world + 1 + 3) * sizeof(i64), ctx->stream));
    i64* d_tot = d_cnt + (size_t)nl * per + world + 1;  // selected: owned rows, send entries, recv entries
    ShardBatch B;
    B.n = nl;
    B.world = world;
    B.me = me;
    B.out_off[0] = B.in_off[0] = 0;
    size_t cap = 0;  // a rank sends / receives an input row at most once per peer, and no more entries than the list has pairs
    for (int j = 0; j < nl; ++j) {
        const ListRef& l = L[j];
        B.out_off[j + 1] = B.out_off[j] + l.v_out;
        B.in_off[j + 1] = B.in_off[j] + (i64)world * l.v_in;
        B.perm[j] = l.perm;
        B.idx[j] = l.idx;
        B.rs[j] = l.rs;
        B.owner_out[j] = st->owner[l.lvl_out];
        B.owner_in[j] = st->owner[l.lvl_in];
        B.v_in[j] = l.v_in;
        B.symmetric[j] = st->owned_lists && l.K == 55 ? 1 : 0;
        cap += std::min<size_t>((size_t)world * l.v_in, (size_t)std::max<i64>(l.K == 55 ? g[l.lvl_out].p : g[std::min(l.lvl_out, l.lvl_in)].v, 1));
    }
    const i64 n_out = B.out_off[nl], n_in = B.in_off[nl];
    if (n_in >= (i64(1) << 31) || n_out >= (i64(1) << 31))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "sharded forward: world x voxels exceeds 2^31");
    const size_t words = ((size_t)n_in + 31) / 32;
    unsigned* F = arena_alloc<unsigned>(ctx->scratch, words);
    unsigned* G = arena_alloc<unsigned>(ctx->scratch, words);
    uint8_t* rflags = arena_alloc<uint8_t>(ctx->scratch, (size_t)n_out);
    int32_t* sel_perm = arena_alloc<int32_t>((*st->mem), (size_t)n_out);
    int32_t* raw_send = arena_alloc<int32_t>(ctx->scratch, cap);
    int32_t* raw_recv = arena_alloc<int32_t>(ctx->scratch, cap);
    if (!F || !G || !rflags || !sel_perm || !raw_send || !raw_recv) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    ASR_HIP_CHECK(ctx, hipMemsetAsync(F, 0, words * 4, ctx->stream));
    ASR_HIP_CHECK(ctx, hipMemsetAsync(G, 0, words * 4, ctx->stream));
    k_shard_flags_batch<<<grid_for(n_out, BLK), BLK, 0, ctx->stream>>>(B, rflags, F, G, raw_send, raw_recv, d_tot + 1, d_tot + 2, (i64)cap);
    ASR_CHECK_LAUNCH(ctx);
    ASR_TRY(shard_select(ctx, rocprim::counting_iterator<int32_t>(0), rflags, sel_perm, d_tot + 0, (size_t)n_out));
    i64 h_tot[3];
    ASR_HIP_CHECK(ctx, hipMemcpyAsync(h_tot, d_tot, sizeof(h_tot), hipMemcpyDeviceToHost, ctx->stream));
    ASR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    const i64 tot_send = h_tot[1], tot_recv = h_tot[2];
    if (tot_send > (i64)cap || tot_recv > (i64)cap) ASR_FAIL(ctx, ASR_HIP_ELOGIC, "sharded forward: halo list overflow");
    // the lists in ascending (list, peer, row) order, straight into the state's arena
    int32_t* sel_send = arena_alloc<int32_t>((*st->mem), (size_t)std::max<i64>(tot_send, 1));
    int32_t* sel_recv = arena_alloc<int32_t>((*st->mem), (size_t)std::max<i64>(tot_recv, 1));
    if (!sel_send || !sel_recv) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    if (tot_send > 0) ASR_TRY(shard_sort_u32(ctx, raw_send, sel_send, tot_send, bits_for(n_in)));
    if (tot_recv > 0) ASR_TRY(shard_sort_u32(ctx, raw_recv, sel_recv, tot_recv, bits_for(n_in)));
    k_shard_bounds_batch<<<grid_for((i64)nl * (1 + 2 * (world + 1)), 128), 128, 0, ctx->stream>>>(
            B, sel_perm, d_tot + 0, sel_send, d_tot + 1, sel_recv, d_tot + 2, d_cnt, per);
    ASR_CHECK_LAUNCH(ctx);
    k_shard_rows_batch<<<1024, BLK, 0, ctx->stream>>>(B, sel_perm, d_tot + 0, 0);
    k_shard_rows_batch<<<256, BLK, 0, ctx->stream>>>(B, sel_send, d_tot + 1, 1);
    k_shard_rows_batch<<<256, BLK, 0, ctx->stream>>>(B, sel_recv, d_tot + 2, 1);
    ASR_CHECK_LAUNCH(ctx);
    // stitch: grid-0 rows grouped by owner
    {
        int32_t* ids = arena_alloc<int32_t>(ctx->scratch, V0);
        int32_t* own_s = arena_alloc<int32_t>(ctx->scratch, V0);
        st->rows0 = arena_alloc<int32_t>((*st->mem), V0);
        if (!ids || !own_s || !st->rows0) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
        k_shard_iota<<<grid_for(V0, BLK), BLK, 0, ctx->stream>>>(ids, V0);
        ASR_CHECK_LAUNCH(ctx);
        ASR_TRY((sort_pairs<int32_t, int32_t>(ctx, ctx->scratch, st->owner[0], own_s, ids, st->rows0, V0, bits_for(world + 1))));
        k_shard_bounds<<<1, 128, 0, ctx->stream>>>(own_s, V0, world, d_cnt + (size_t)nl * per);
        ASR_CHECK_LAUNCH(ctx);
    }
    std::vector<i64> h_cnt((size_t)nl * per + world + 1 + 3);
    ASR_HIP_CHECK(ctx, hipMemcpyAsync(h_cnt.data(), d_cnt, h_cnt.size() * sizeof(i64), hipMemcpyDeviceToHost, ctx->stream));
    ASR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (int p = 0; p <= world; ++p) st->rows0_off[p] = h_cnt[(size_t)nl * per + p];
    // plans of the owned rows
    int32_t* send_all = sel_send;
    int32_t* recv_all = sel_recv;
    std::vector<asr_conv_plan> plans;
    std::vector<int> plan_of;
    for (int j = 0; j < nl; ++j) {
        const ListRef& l = L[j];
        const i64* c = &h_cnt[(size_t)j * per];
        ShardCsr cs;
        cs.perm = sel_perm + c[3];
        cs.num_out = c[0];
        const i64 ns = c[1], nr = c[2];
        if (ns > 0) cs.send_rows = send_all + c[4];
        if (nr > 0) cs.recv_rows = recv_all + c[5];
        for (int p = 0; p < world; ++p) {
            const i64 s0 = c[6 + p], s1 = c[6 + p + 1];
            if (s1 > s0) {
                cs.send_peer.push_back(p);
                cs.send_first.push_back(s0);
            }
            const i64 r0 = c[6 + world + 1 + p], r1 = c[6 + world + 1 + p + 1];
            if (r1 > r0) {
                cs.recv_peer.push_back(p);
                cs.recv_first.push_back(r0);
            }
        }
        cs.send_first.push_back(ns);
        cs.recv_first.push_back(nr);
        if (l.K == 55) st->stats.halo_rows_recv[l.lvl_out] = nr;
        if (l.K == 55) st->stats.owned_rows[l.lvl_out] = cs.num_out;
        st->csr[l.rs] = cs;
        if (want_plans && cs.num_out > 0) {
            asr_conv_plan pl;
            pl.nidx = l.idx;
            pl.kidx = l.kidx;
            pl.rs = l.rs;
            pl.perm = cs.perm;
            pl.num_out = cs.num_out;
            pl.K = l.K;
            plans.push_back(pl);
            plan_of.push_back(j);
        }
    }
    if (!plans.empty()) {
        ASR_TRY(asr_geom_conv_plan_batch(ctx, (*st->mem), plans.data(), (int)plans.size()));
        for (size_t q = 0; q < plans.size(); ++q) {
            ShardCsr& cs = st->csr[L[plan_of[q]].rs];
            cs.plan = plans[q];
            cs.has_plan = true;
        }
    }
    return ASR_HIP_OK;
}

// ownership by pair counts + lists: the geometry of the whole cloud is on every rank
int asr_shard_build(asr_hip_context* ctx, const asr_shard_comm* comm, int want_plans, asr_shard_state** out) {
    *out = nullptr;
    asr_shard_state* st = nullptr;
    ASR_TRY(asr_shard_ownership(ctx, comm, 1, &st));
    const int rc = asr_shard_lists(ctx, st, want_plans);
    if (rc != ASR_HIP_OK) {
        asr_shard_free(st);
        return rc;
    }
    *out = st;
    return ASR_HIP_OK;
}
const int32_t* asr_shard_level_rows(const asr_shard_state* st, int level, i64* n) {
    *n = st->level_nrows[level];
    return st->level_rows[level];
}
// rows the aggregation of a rank with sharded geometry covers: [0, prefix) (the importance of the first V0
// PAIRS is read by every rank) followed by the owned rows >= prefix, ascending
int asr_shard_query_rows(asr_hip_context* ctx, const asr_shard_state* st, i64 prefix, Arena& keep, int32_t** rows_out,
                         i64* n_out) {
    const i64 n_own = st->level_nrows[0];
    int32_t* q = arena_alloc<int32_t>(keep, prefix + n_own);
    uint8_t* fl = arena_alloc<uint8_t>(ctx->scratch, n_own > 0 ? n_own : 1);
    i64* d_n = arena_alloc<i64>(ctx->scratch, 1);
    if (!q || !fl || !d_n) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    if (prefix > 0) {
        k_shard_iota<<<grid_for(prefix, BLK), BLK, 0, ctx->stream>>>(q, prefix);
        ASR_CHECK_LAUNCH(ctx);
    }
    i64 h_n = 0;
    if (n_own > 0) {
        k_shard_flag_ge<<<grid_for(n_own, BLK), BLK, 0, ctx->stream>>>(st->level_rows[0], n_own, (int32_t)prefix, fl);
        ASR_CHECK_LAUNCH(ctx);
        ASR_TRY(shard_select(ctx, st->level_rows[0], fl, q + prefix, d_n, (size_t)n_own));
        ASR_HIP_CHECK(ctx, hipMemcpyAsync(&h_n, d_n, sizeof(i64), hipMemcpyDeviceToHost, ctx->stream));
        ASR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
    *rows_out = q;
    *n_out = prefix + h_n;
    return ASR_HIP_OK;
}
int asr_shard_rank(const asr_shard_state* st) { return st->rank; }
const int32_t* asr_shard_owned_rows0(const asr_shard_state* st, i64* n) {
    *n = st->rows0_off[st->rank + 1] - st->rows0_off[st->rank];
    return st->rows0 ? st->rows0 + st->rows0_off[st->rank] : nullptr;
}
int asr_shard_world(const asr_shard_state* st) { return st->world; }

// before a convolution over the list `rs`: the rows it computes (perm, num_out, plan), the halo exchange of its input and
// -- f16x2: in_amax, the running maximum the producers of the input buffer kept over the rows THEY wrote -- the maximum
// over the ranks (the tensor's, as on one GPU), in the same group as the exchange where the transport can
int asr_shard_before_conv(asr_hip_context* ctx, asr_shard_state* st, const void* rs, void* feat, i64 ld_bytes, i64 row_bytes,
                          float* imp, unsigned* in_amax, const int32_t** perm, i64* num_out, const asr_conv_plan** plan) {
    auto it = st->csr.find(rs);
    if (it == st->csr.end()) ASR_FAIL(ctx, ASR_HIP_ELOGIC, "sharded forward: convolution over an unknown neighbour list");
    ShardCsr& c = it->second;
    *perm = c.perm;
    *num_out = c.num_out;
    *plan = c.has_plan ? &c.plan : nullptr;
    const i64 ns = c.send_first.back(), nr = c.recv_first.back();
    if (st->world == 1) return ASR_HIP_OK;
    if (ns == 0 && nr == 0) {  // no halo here, but the maximum is a collective: every rank takes part
        if (ctx->dry_launch) return ASR_HIP_OK;
        if (in_amax && st->comm.allreduce_max_u32(st->comm.user, in_amax, 1, (void*)ctx->stream) != 0)
            ASR_FAIL(ctx, ASR_HIP_EHIP, "sharded forward: the MAX all-reduce failed");
        return ASR_HIP_OK;
    }
    if (ld_bytes % 4 || row_bytes % 4 || (uintptr_t)feat % 4)
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "sharded forward: feature rows must be multiples of 4 bytes");
    const int row_dwords = (int)(row_bytes / 4);
    const i64 per_row = row_dwords + (imp ? 1 : 0);
    ASR_TRY(shard_stage(ctx, st, (size_t)std::max(ns, nr) * per_row));
    if (ctx->dry_launch) return ASR_HIP_OK;  // preparation pass: the staging buffers are sized, nothing travels
    PackDesc ds, dr;
    ds.npeer = (int)c.send_peer.size();
    dr.npeer = (int)c.recv_peer.size();
    std::vector<const void*> sbuf(ds.npeer);
    std::vector<void*> rbuf(dr.npeer);
    std::vector<size_t> sbytes(ds.npeer), rbytes(dr.npeer);
    for (int p = 0; p <= ds.npeer; ++p) {
        ds.row_first[p] = c.send_first[p];
        ds.msg_dword[p] = c.send_first[p] * per_row;
    }
    for (int p = 0; p <= dr.npeer; ++p) {
        dr.row_first[p] = c.recv_first[p];
        dr.msg_dword[p] = c.recv_first[p] * per_row;
    }
    for (int p = 0; p < ds.npeer; ++p) {
        sbuf[p] = st->stage_send + ds.msg_dword[p];
        sbytes[p] = (size_t)(ds.msg_dword[p + 1] - ds.msg_dword[p]) * 4;
    }
    for (int p = 0; p < dr.npeer; ++p) {
        rbuf[p] = st->stage_recv + dr.msg_dword[p];
        rbytes[p] = (size_t)(dr.msg_dword[p + 1] - dr.msg_dword[p]) * 4;
    }
    if (ns > 0) {
        k_shard_pack<<<grid_for(ns * per_row, BLK), BLK, 0, ctx->stream>>>(ds, c.send_rows, (const unsigned*)feat, ld_bytes / 4,
                                                                          row_dwords, imp, st->stage_send);
        ASR_CHECK_LAUNCH(ctx);
    }
    const bool timed = ctx->opt.shard_timing != 0;
    std::chrono::steady_clock::time_point t0;
    if (timed) {
        ASR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        t0 = std::chrono::steady_clock::now();
    }
    int bad;
    if (in_amax && st->comm.exchange_and_max) {
        bad = st->comm.exchange_and_max(st->comm.user, ds.npeer, c.send_peer.data(), sbuf.data(), sbytes.data(), dr.npeer,
                                        c.recv_peer.data(), rbuf.data(), rbytes.data(), in_amax, 1, (void*)ctx->stream);
    } else {
        bad = in_amax ? st->comm.allreduce_max_u32(st->comm.user, in_amax, 1, (void*)ctx->stream) : 0;
        if (!bad)
            bad = st->comm.exchange(st->comm.user, ds.npeer, c.send_peer.data(), sbuf.data(), sbytes.data(), dr.npeer,
                                    c.recv_peer.data(), rbuf.data(), rbytes.data(), (void*)ctx->stream);
    }
    if (bad) ASR_FAIL(ctx, ASR_HIP_EHIP, "sharded forward: the halo exchange failed");
    if (timed) {
        ASR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        st->stats.exchange_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    if (nr > 0) {
        k_shard_unpack<<<grid_for(nr * per_row, BLK), BLK, 0, ctx->stream>>>(dr, c.recv_rows, (unsigned*)feat, ld_bytes / 4,
                                                                            row_dwords, imp, st->stage_recv);
        ASR_CHECK_LAUNCH(ctx);
    }
    st->stats.bytes_sent += ns * per_row * 4;
    st->stats.bytes_received += nr * per_row * 4;
    st->stats.exchanges += 1;
    return ASR_HIP_OK;
}

// A failure on ONE rank between collectives would leave its peers inside ncclSend / ncclRecv for ever.  The sharded forward
// therefore does everything that can fail for a reason of the rank's own -- the whole build (no collective in it), then the
// preparation pass of the network (argument and weight checks, weight packing, every allocation incl. the staging buffers) --
// BEFORE the first exchange, and the ranks agree on the outcome of each of the two parts with one MAX all-reduce of a
// status word: if any rank failed, every rank returns (the failed ones their own error, the others ASR_HIP_EPEER).
int asr_shard_agree(asr_hip_context* ctx, const asr_shard_comm* comm, int rc_local, const char* phase) {
    if (!comm || comm->world <= 1) return rc_local;
    const std::string own_err = ctx->err;  // (the calls below reuse the context's error text)
    if (!ctx->d_status && hipMalloc((void**)&ctx->d_status, 256) != hipSuccess) {
        ctx->d_status = nullptr;  // (this rank cannot take part: its peers wait -- nothing a rank without memory can do)
        if (rc_local != ASR_HIP_OK) return rc_local;
        ASR_FAIL(ctx, ASR_HIP_EHIP, "sharded forward: no memory for the status word");
    }
    unsigned* word = ctx->d_status;
    unsigned status = rc_local != ASR_HIP_OK ? 1u : 0u;
    bool ok = hipMemcpyAsync(word, &status, sizeof(status), hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
    ok = ok && comm->allreduce_max_u32(comm->user, word, 1, (void*)ctx->stream) == 0;
    unsigned any = 1;
    ok = ok && hipMemcpyAsync(&any, word, sizeof(any), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
    ok = ok && hipStreamSynchronize(ctx->stream) == hipSuccess;
    if (rc_local != ASR_HIP_OK) {
        ctx->err = own_err;
        return rc_local;
    }
    if (!ok) ASR_FAIL(ctx, ASR_HIP_EHIP, "sharded forward: the status all-reduce after the %s failed", phase);
    if (any) ASR_FAIL(ctx, ASR_HIP_EPEER, "sharded forward: another rank failed in its %s; no rank goes on", phase);
    return ASR_HIP_OK;
}

// values [V0, 2]: this rank's rows are valid; afterwards all rows are (all-gather of the owned rows)
int asr_shard_stitch(asr_hip_context* ctx, asr_shard_state* st, float* values) {
    if (st->world == 1) return ASR_HIP_OK;
    const int world = st->world, me = st->rank;
    const i64 V0 = st->rows0_off[world];
    const i64 mine = st->rows0_off[me + 1] - st->rows0_off[me];
    ASR_TRY(shard_stage(ctx, st, (size_t)V0 * 2));
    if (ctx->dry_launch) return ASR_HIP_OK;
    // my rows packed once, sent to every peer; the peers' rows land at their offsets of the grouped row list
    PackDesc dm;
    dm.npeer = 1;
    dm.row_first[0] = 0;
    dm.row_first[1] = mine;
    dm.msg_dword[0] = 0;
    dm.msg_dword[1] = mine * 2;
    if (mine > 0) {
        k_shard_pack<<<grid_for(mine * 2, BLK), BLK, 0, ctx->stream>>>(dm, st->rows0 + st->rows0_off[me], (const unsigned*)values,
                                                                      2, 2, nullptr, st->stage_send);
        ASR_CHECK_LAUNCH(ctx);
    }
    std::vector<int> speer, rpeer;
    std::vector<const void*> sbuf;
    std::vector<void*> rbuf;
    std::vector<size_t> sbytes, rbytes;
    for (int p = 0; p < world; ++p) {
        if (p == me) continue;
        if (mine > 0) {
            speer.push_back(p);
            sbuf.push_back(st->stage_send);
            sbytes.push_back((size_t)mine * 8);
        }
        const i64 np = st->rows0_off[p + 1] - st->rows0_off[p];
        if (np > 0) {
            rpeer.push_back(p);
            rbuf.push_back(st->stage_recv + st->rows0_off[p] * 2);
            rbytes.push_back((size_t)np * 8);
        }
    }
    if (st->comm.exchange(st->comm.user, (int)speer.size(), speer.data(), sbuf.data(), sbytes.data(), (int)rpeer.size(),
                          rpeer.data(), rbuf.data(), rbytes.data(), (void*)ctx->stream) != 0)
        ASR_FAIL(ctx, ASR_HIP_EHIP, "sharded forward: the all-gather of the values failed");
    for (int p = 0; p < world; ++p) {
        const i64 np = st->rows0_off[p + 1] - st->rows0_off[p];
        if (p == me || np == 0) continue;
        PackDesc dp;
        dp.npeer = 1;
        dp.row_first[0] = 0;
        dp.row_first[1] = np;
        dp.msg_dword[0] = st->rows0_off[p] * 2;
        dp.msg_dword[1] = st->rows0_off[p + 1] * 2;
        k_shard_unpack<<<grid_for(np * 2, BLK), BLK, 0, ctx->stream>>>(dp, st->rows0 + st->rows0_off[p], (unsigned*)values, 2, 2,
                                                                     nullptr, st->stage_recv);
        ASR_CHECK_LAUNCH(ctx);
        st->stats.bytes_received += np * 8;
    }
    st->stats.bytes_sent += mine * 8 * (world - 1);
    st->stats.exchanges += 1;
    return ASR_HIP_OK;
}

// ------------------------------------------------------------------------------------------
// RCCL transport (librccl.so loaded at run time: the library itself does not link against it)
// ------------------------------------------------------------------------------------------
namespace {
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    bool ok() const {
        return lib && GetUniqueId && CommInitRank && CommDestroy && GroupStart && GroupEnd && Send && Recv && AllReduce;
    }
};
RcclApi& rccl() {
    static RcclApi api = [] {
        RcclApi a;
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (a.lib) break;
        }
        if (a.lib) {
            a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.lib, "ncclGetUniqueId");
            a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.lib, "ncclCommInitRank");
            a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.lib, "ncclCommDestroy");
            a.GroupStart = (decltype(a.GroupStart))dlsym(a.lib, "ncclGroupStart");
            a.GroupEnd = (decltype(a.GroupEnd))dlsym(a.lib, "ncclGroupEnd");
            a.Send = (decltype(a.Send))dlsym(a.lib, "ncclSend");
            a.Recv = (decltype(a.Recv))dlsym(a.lib, "ncclRecv");
            a.AllReduce = (decltype(a.AllReduce))dlsym(a.lib, "ncclAllReduce");
        }
        return a;
    }();
    return api;
}
struct RcclComm {
    asr_shard_comm base;
    ncclComm_t comm = nullptr;
};
int rccl_exchange(void* user, int nsend, const int* send_peer, const void* const* send_buf, const size_t* send_bytes,
                  int nrecv, const int* recv_peer, void* const* recv_buf, const size_t* recv_bytes, void* stream) {
    RcclComm* c = (RcclComm*)user;
    RcclApi& a = rccl();
    if (nsend == 0 && nrecv == 0) return 0;
    if (a.GroupStart() != ncclSuccess) return 1;
    int bad = 0;
    for (int i = 0; i < nsend; ++i)
        bad |= a.Send(send_buf[i], send_bytes[i], ncclUint8, send_peer[i], c->comm, (hipStream_t)stream) != ncclSuccess;
    for (int i = 0; i < nrecv; ++i)
        bad |= a.Recv(recv_buf[i], recv_bytes[i], ncclUint8, recv_peer[i], c->comm, (hipStream_t)stream) != ncclSuccess;
    if (a.GroupEnd() != ncclSuccess) return 1;
    return bad;
}
int rccl_allreduce_max(void* user, uint32_t* buf, size_t n, void* stream) {
    RcclComm* c = (RcclComm*)user;
    return rccl().AllReduce(buf, buf, n, ncclUint32, ncclMax, c->comm, (hipStream_t)stream) != ncclSuccess;
}
int rccl_exchange_and_max(void* user, int nsend, const int* send_peer, const void* const* send_buf, const size_t* send_bytes,
                          int nrecv, const int* recv_peer, void* const* recv_buf, const size_t* recv_bytes, uint32_t* max_buf,
                          size_t max_n, void* stream) {
    RcclComm* c = (RcclComm*)user;
    RcclApi& a = rccl();
    if (a.GroupStart() != ncclSuccess) return 1;
    int bad = a.AllReduce(max_buf, max_buf, max_n, ncclUint32, ncclMax, c->comm, (hipStream_t)stream) != ncclSuccess;
    for (int i = 0; i < nsend; ++i)
        bad |= a.Send(send_buf[i], send_bytes[i], ncclUint8, send_peer[i], c->comm, (hipStream_t)stream) != ncclSuccess;
    for (int i = 0; i < nrecv; ++i)
        bad |= a.Recv(recv_buf[i], recv_bytes[i], ncclUint8, recv_peer[i], c->comm, (hipStream_t)stream) != ncclSuccess;
    if (a.GroupEnd() != ncclSuccess) return 1;
    return bad;
}
}  // namespace

extern "C" {
int asr_hip_shard_comm_rccl_unique_id(asr_hip_context* ctx, void* unique_id_out) {
    if (!ctx) return ASR_HIP_EINVAL;
    if (!unique_id_out) ASR_FAIL(ctx, ASR_HIP_EINVAL, "rccl_unique_id: null output");
    if (!rccl().ok()) ASR_FAIL(ctx, ASR_HIP_ENODEV, "librccl.so could not be loaded");
    ncclUniqueId id;
    if (rccl().GetUniqueId(&id) != ncclSuccess) ASR_FAIL(ctx, ASR_HIP_EHIP, "ncclGetUniqueId failed");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    memcpy(unique_id_out, &id, sizeof(id));
    return ASR_HIP_OK;
}
int asr_hip_shard_comm_rccl_create(asr_hip_context* ctx, const void* unique_id, int rank, int world,
                                   asr_shard_comm** comm_out) {
    if (!ctx) return ASR_HIP_EINVAL;
    if (!unique_id || !comm_out || world < 1 || rank < 0 || rank >= world)
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "rccl_create: bad argument");
    *comm_out = nullptr;
    if (!rccl().ok()) ASR_FAIL(ctx, ASR_HIP_ENODEV, "librccl.so could not be loaded");
    ASR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ASR_TRY(asr_ctx_ensure_aux(ctx));  // the search's stream before RCCL makes its own (hardware queues, see there)
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    RcclComm* c = new RcclComm();
    if (rccl().CommInitRank(&c->comm, world, id, rank) != ncclSuccess) {
        delete c;
        ASR_FAIL(ctx, ASR_HIP_EHIP, "ncclCommInitRank failed (rank %d of %d)", rank, world);
    }
    c->base.user = c;
    c->base.rank = rank;
    c->base.world = world;
    c->base.exchange = rccl_exchange;
    c->base.allreduce_max_u32 = rccl_allreduce_max;
    c->base.exchange_and_max = rccl_exchange_and_max;
    *comm_out = &c->base;
    return ASR_HIP_OK;
}
void asr_hip_shard_comm_rccl_destroy(asr_shard_comm* comm) {
    if (!comm) return;
    RcclComm* c = (RcclComm*)comm->user;
    if (c && c->comm && rccl().ok()) (void)rccl().CommDestroy(c->comm);
    delete c;
}
}  // extern "C"

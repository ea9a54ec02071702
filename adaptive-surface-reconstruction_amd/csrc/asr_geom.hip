// asr_geom.hip -- integer / geometry half of the hot path on gfx950:
//   a3 point location codes, a4 octree closure + 2:1 face balance, a5 55-slot neighbour CSR,
//   a6 sibling coarsening, a7 voxel info, a8 multi-radius search + scale compatibility,
//   a11 CSR inversion.
// All kernels are HBM / latency bound integer work: open-addressing hash tables in HBM
// (64-bit CAS), rocPRIM radix sort / scan for ordering, one thread per voxel / point / pair.
// Compiled with -ffp-contract=off: the float expressions below must round exactly like the
// reference's.
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <rocprim/rocprim.hpp>

#include "asr_common.h"
#include "asr_prim.h"

namespace {
using namespace asr_prim;

constexpr int BLK = 256;

// ------------------------------------------------------------------------------------------
// hash set / map on u64 keys (0 = empty)
// ------------------------------------------------------------------------------------------
struct HashTab {
    u64* keys;
    int32_t* vals;  // may be null (set)
    u64 mask;
};

// Probe sequence of a key.  Locality preserving: the low 6 bits of a location code (two levels of child index = the
// position inside a 4 x 4 x 4 block of cells) are kept, only the block id is scrambled.  The keys of one block land in
// one 64-slot bucket (512 B) at distinct slots, and the probes a voxel makes for its face neighbours -- or a wave of
// Morton-ordered voxels makes together -- stay in a few cache lines instead of one random line per probe.  A
// collision (another block hashed to the same bucket) moves to the SAME position of another bucket (double
// hashing over buckets): stepping to the next slot instead would run through the neighbouring keys of both
// blocks (measured: 5x slower than plain hashing).  cap is a power of two >= 64, the bucket step is odd.
// Keys whose low 6 bits are concentrated on a few values (points on a lattice) fill "their" position of the buckets
// before the table is half full: the builders report the overflow and the hosts rebuild the table four times the size.
struct TabProbe {
    u64 bucket, step, low, bmask;
    __device__ inline u64 slot() const { return ((bucket & bmask) << 6) | low; }
    __device__ inline void next() { bucket += step; }
};
__device__ inline TabProbe tab_probe(const HashTab& t, u64 key) {
    const u64 h = asr_hash64(key >> 6);
    return TabProbe{h, (h >> 32) | 1, key & 63, t.mask >> 6};
}

__device__ inline u64 ld_agent(const u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// returns 1 = inserted, 0 = existed, -1 = table full; slot_out = slot of the key
__device__ inline int tab_insert(const HashTab& t, u64 key, u64* slot_out = nullptr) {
    TabProbe pr = tab_probe(t, key);
    for (u64 probe = 0; probe <= pr.bmask; ++probe, pr.next()) {
        const u64 slot = pr.slot();
        u64 cur = ld_agent(&t.keys[slot]);
        if (cur == 0) {
            cur = atomicCAS((unsigned long long*)&t.keys[slot], 0ull, (unsigned long long)key);
            if (cur == 0) {
                if (slot_out) *slot_out = slot;
                return 1;
            }
        }
        if (cur == key) {
            if (slot_out) *slot_out = slot;
            return 0;
        }
    }
    return -1;
}
// the same for key sets with FEW distinct values among many insertions (the slot masks of the neighbour lists: most rows
// of a grid share the interior mask, the rows of other ranks of a sharded cloud the empty one): an agent-scope load of one
// address by every wave of the launch serialises at that address's memory channel, so the probe reads through the caches
// first.  A key never changes once written: a cached non-zero value is the truth, a cached zero is checked again.
__device__ inline int tab_insert_shared(const HashTab& t, u64 key) {
    TabProbe pr = tab_probe(t, key);
    for (u64 probe = 0; probe <= pr.bmask; ++probe, pr.next()) {
        const u64 slot = pr.slot();
        u64 cur = t.keys[slot];
        if (cur == 0) cur = ld_agent(&t.keys[slot]);
        if (cur == 0) {
            cur = atomicCAS((unsigned long long*)&t.keys[slot], 0ull, (unsigned long long)key);
            if (cur == 0) return 1;
        }
        if (cur == key) return 0;
    }
    return -1;
}
// lookup after the building kernel finished (plain loads)
__device__ inline bool tab_contains(const HashTab& t, u64 key) {
    TabProbe pr = tab_probe(t, key);
    for (u64 probe = 0; probe <= pr.bmask; ++probe, pr.next()) {
        const u64 slot = pr.slot();
        const u64 cur = t.keys[slot];
        if (cur == key) return true;
        if (cur == 0) return false;
    }
    return false;
}
__device__ inline int tab_find(const HashTab& t, u64 key) {
    TabProbe pr = tab_probe(t, key);
    for (u64 probe = 0; probe <= pr.bmask; ++probe, pr.next()) {
        const u64 slot = pr.slot();
        const u64 cur = t.keys[slot];
        if (cur == key) return t.vals[slot];
        if (cur == 0) return -1;
    }
    return -1;
}
__device__ inline i64 tab_find_slot(const HashTab& t, u64 key) {
    TabProbe pr = tab_probe(t, key);
    for (u64 probe = 0; probe <= pr.bmask; ++probe, pr.next()) {
        const u64 slot = pr.slot();
        const u64 cur = t.keys[slot];
        if (cur == key) return (i64)slot;
        if (cur == 0) return -1;
    }
    return -1;
}

// append with ONE atomic per 256-thread block; must be reached by every thread of the block (4 waves).
// Same-address atomics retire at ~88 / us: one per wave is 0.5 ms for 2.6 M voxels.
__device__ inline int block_append(bool pred, int* counter) {
    __shared__ int s_cnt[4];
    __shared__ int s_base;
    const unsigned long long m = __ballot(pred);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_cnt[wave] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        s_base = total ? atomicAdd(counter, total) : 0;
    }
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; ++w) off += s_cnt[w];
    return off + __popcll(m & ((1ull << lane) - 1));
}

u64 next_pow2(u64 x) {
    u64 p = 1;
    while (p < x) p <<= 1;
    return p;
}

// ------------------------------------------------------------------------------------------
// a3: point -> (level, key)   octree.h:42-68
// ------------------------------------------------------------------------------------------
__device__ inline int level_from_scale(const asr_octree_frame& f, float scale) {
    for (int level = 0; level <= ASR_MAX_LEVEL; ++level)
        if (f.voxel_size[level] < scale) return level - 1 > 0 ? level - 1 : 0;
    return ASR_MAX_LEVEL;
}
__device__ inline void frame_coord(const asr_octree_frame& f, float px, float py, float pz,
                                   int level, int& x, int& y, int& z) {
    float inv = f.inv_voxel_size[ASR_MAX_LEVEL];
    float tx = px * inv, ty = py * inv, tz = pz * inv;
    x = (int)floorf(tx) + f.offset[0];
    y = (int)floorf(ty) + f.offset[1];
    z = (int)floorf(tz) + f.offset[2];
    int s = ASR_MAX_LEVEL - level;
    x >>= s;
    y >>= s;
    z >>= s;
}
__device__ inline u64 point_key(const asr_octree_frame& f, const float* pts, const float* radii,
                                i64 i, float radius_scale, int max_depth) {
    float px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
    if (px < f.bb_min[0] || py < f.bb_min[1] || pz < f.bb_min[2] || px > f.bb_max[0] ||
        py > f.bb_max[1] || pz > f.bb_max[2])
        return 0;
    int level = level_from_scale(f, radius_scale * radii[i]);
    level = level < max_depth ? level : max_depth;
    int x, y, z;
    frame_coord(f, px, py, pz, level, x, y, z);
    return asr_coord_key(x, y, z, level);
}

__global__ void k_point_keys(asr_octree_frame f, const float* pts, const float* radii, i64 n,
                             float radius_scale, int max_depth, u64* keys) {
    i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (i < n) keys[i] = point_key(f, pts, radii, i, radius_scale, max_depth);
}

// ------------------------------------------------------------------------------------------
// a4: octree node set.  Nodes live in an open-addressing hash set (membership) AND in an append
// only list of sibling groups (8 consecutive keys, first sibling first): the list gives the
// BalanceFaces frontiers and the final node array without ever scanning the (mostly empty) table.
// cnt[0] = list entries (multiple of 8), cnt[1] = overflow flag, cnt[9] = root inserted.
// ------------------------------------------------------------------------------------------
// insert the sibling group of `a` and all missing ancestor groups (+ root).
// The thread that wins the CAS on a group's first sibling creates the other seven and
// continues upwards; losers stop (CreateAncestorsAndSiblings, octree.cpp:110-150).
__device__ inline void insert_with_ancestors(const HashTab& t, u64 a, int* cnt, u64* list,
                                             int list_cap) {
    const int lane = threadIdx.x & 63;
    while (true) {
        if (a == 1) {
            int r = tab_insert(t, 1);
            if (r == 1) cnt[9] = 1;
            if (r < 0) cnt[1] = 1;
            return;
        }
        u64 first = a & ~u64(7);
        int r = tab_insert(t, first);
        if (r < 0) {
            cnt[1] = 1;
            return;
        }
        // One list append per WAVE iteration: the lanes of the wave that are in this iteration of the loop and created
        // a group share one atomic on the list counter (same-address atomics are the bottleneck of this kernel: one
        // per new group was 0.37 M of them at 10 M points).  __ballot / __shfl act on the lanes that execute them.
        const bool created = r == 1;
        const unsigned long long m = __ballot(created);
        if (!created) return;
        for (int j = 1; j < 8; ++j)
            if (tab_insert(t, first + j) < 0) cnt[1] = 1;
        const int leader = __builtin_ctzll(m);
        int base = 0;
        if (lane == leader) base = atomicAdd(&cnt[0], 8 * __popcll(m));
        base = __shfl(base, leader, 64);
        const int pos = base + 8 * __popcll(m & ((1ull << lane) - 1));
        if (pos + 8 <= list_cap) {
            for (int j = 0; j < 8; ++j) list[pos + j] = first + j;
        } else {
            cnt[1] = 1;
        }
        a >>= 3;
    }
}

// cnt[13] = number of significant bits of the longest key in the node list (cnt[0] entries; the list is radix-sorted on
// those bits only).  A pass of its own over the list: the same test inside the insertion kernels, next to their atomics
// on cnt[0], made them 2.6x slower.
__global__ void k_list_key_bits(const u64* list, int list_cap, int* cnt) {
    const int n = min(cnt[0], list_cap);
    int bits = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const u64 k = list[i];
        bits = max(bits, k ? 64 - __clzll((long long)k) : 0);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bits = max(bits, __shfl_xor(bits, o, 64));
    if ((threadIdx.x & 63) == 0 && bits) atomicMax(&cnt[13], bits);
}

__global__ void k_octree_insert_points(asr_octree_frame f, const float* pts, const float* radii,
                                       i64 n, float radius_scale, int max_depth, HashTab t,
                                       int* cnt, u64* list, int list_cap) {
    i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    u64 key = 0;
    if (i < n) {
        if (!(isfinite(pts[3 * i]) && isfinite(pts[3 * i + 1]) && isfinite(pts[3 * i + 2]) && isfinite(radii[i])))
            cnt[11] = 1;  // rejected by the host: the reference has undefined behaviour on such input
        else
            key = point_key(f, pts, radii, i, radius_scale, max_depth);
    }
    // consecutive points of a scan mostly fall into the same cell: a lane whose left neighbour carries the same key
    // leaves the insertion (a probe of the table at least) to it
    const u64 left = __shfl_up(key, 1, 64);
    if ((threadIdx.x & 63) != 0 && left == key) return;
    if (key == 0) return;  // the reference inserts key 0 here and then hits UB
    insert_with_ancestors(t, key, cnt, list, list_cap);
}

// ------------------------------------------------------------------------------------------
// Octree::Grow (octree.cpp:44-108), on the raw point keys: a second hash set + key list.  One iteration = two
// kernels: the candidates of condition :87 ("not a node, has no child") are evaluated against the set as it is at
// the START of the iteration, then candidates and siblings are inserted (what is new forms the next frontier).
// ------------------------------------------------------------------------------------------
__global__ void k_grow_init(asr_octree_frame f, const float* pts, const float* radii, i64 n, float radius_scale,
                            int max_depth, HashTab t, int* cnt, u64* list, int list_cap) {
    i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (!(isfinite(pts[3 * i]) && isfinite(pts[3 * i + 1]) && isfinite(pts[3 * i + 2]) && isfinite(radii[i]))) {
        cnt[11] = 1;
        return;
    }
    const u64 key = point_key(f, pts, radii, i, radius_scale, max_depth);
    if (key == 0) return;
    const int r = tab_insert(t, key);
    if (r < 0) cnt[1] = 1;
    if (r == 1) {
        const int pos = atomicAdd(&cnt[0], 1);
        if (pos < list_cap)
            list[pos] = key;
        else
            cnt[1] = 1;
    }
}
__device__ inline bool tab_has_child(const HashTab& t, u64 key) {
    if (__clzll((long long)key) <= 1) return false;
    for (int k = 0; k < 8; ++k)
        if (tab_contains(t, (key << 3) + k)) return true;
    return false;
}
// one thread per (frontier key, j): cand[7 i + jj] = the parent-level cell to add, or 0
__global__ void k_grow_candidates(HashTab t, const u64* list, int lo, int count, u64* cand) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count * 7) return;
    const u64 cur = list[lo + e / 7];
    u64 out = 0;
    if (cur != 1) {
        const int cfg = 7 - (int)(cur & 7);
        int j = e % 7;
        if (j >= cfg) ++j;  // the eight cells of the block without the parent itself
        int x, y, z, lev;
        asr_key_coord(cur >> 3, x, y, z, lev);
        const u64 key = asr_coord_key(x + (j & 1) - (cfg & 1), y + ((j >> 1) & 1) - ((cfg >> 1) & 1),
                                      z + ((j >> 2) & 1) - ((cfg >> 2) & 1), lev);
        if (key != 0 && !tab_contains(t, key) && !tab_has_child(t, key)) out = key;
    }
    cand[e] = out;
}
__global__ void k_grow_insert(HashTab t, u64* list, int lo, int count, const u64* cand, int* cnt, int list_cap) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count * 14) return;
    const u64 cur = list[lo + e / 14];
    if (cur == 1) return;
    const int j = e % 14;
    u64 key;
    if (j < 7) {
        key = cand[(e / 14) * 7 + j];
        if (key == 0) return;
    } else {  // siblings (:94-101)
        const u64 first = cur & ~u64(7);
        int sj = j - 7;
        if (first + sj >= cur) ++sj;
        key = first + sj;
    }
    const int r = tab_insert(t, key);
    if (r < 0) cnt[1] = 1;
    if (r == 1) {
        const int pos = atomicAdd(&cnt[0], 1);
        if (pos < list_cap)
            list[pos] = key;
        else
            cnt[1] = 1;
    }
}
// closure over an explicit key list (the grown set) instead of the points
__global__ void k_octree_insert_keys(const u64* keys, i64 n, HashTab t, int* cnt, u64* list, int list_cap) {
    i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (i >= n) return;
    insert_with_ancestors(t, keys[i], cnt, list, list_cap);
}

// BalanceFaces (octree.cpp:152-206), one round: the frontier is the list range [lo, hi) of sibling
// groups.  Leaf test against the node set at round start (octree.cpp:175) ...
__global__ void k_balance_classify(HashTab t, const u64* list, int lo, int ngroups, uint8_t* flag) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ngroups) return;
    u64 k = list[lo + 8 * i];
    bool has_child = (__clzll((long long)k) > 1) && tab_contains(t, k << 3);
    flag[i] = has_child ? 0 : 1;
}
// ... then octree.cpp:177-203 for all leaf groups: the 6 face neighbours of the parent must exist
__global__ void k_balance_insert(HashTab t, u64* list, int lo, int ngroups, const uint8_t* flag,
                                 int* cnt, int list_cap) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ngroups * 6) return;
    if (!flag[i / 6]) return;
    u64 k = list[lo + 8 * (i / 6)];
    int j = i % 6;
    const int ox = (j == 0) ? -1 : (j == 3) ? 1 : 0;
    const int oy = (j == 1) ? -1 : (j == 4) ? 1 : 0;
    const int oz = (j == 2) ? -1 : (j == 5) ? 1 : 0;
    int x, y, z, lev;
    asr_key_coord(k >> 3, x, y, z, lev);
    u64 key = asr_coord_key(x + ox, y + oy, z + oz, lev);
    if (key == 0) return;
    insert_with_ancestors(t, key, cnt, list, list_cap);
}
// leaves = nodes without first child (InitAttributesAndLeaves, octree.cpp:208-228): flags over the SORTED node
// array, compacted in order (the leaves come out sorted: no second radix sort)
__global__ void k_leaf_flags(HashTab t, const u64* nodes, i64 n, uint8_t* flags) {
    i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 k = nodes[i];
    flags[i] = (k != 0 && !((__clzll((long long)k) > 1) && tab_contains(t, k << 3))) ? 1 : 0;
}

// number of leaves + first / last leaf key in one read-back (the levels of the leaves are the query levels of the
// aggregation search)
__global__ void k_leaf_tail(const u64* leaves, const i64* num, i64* out3) {
    const i64 n = *num;
    out3[0] = n;
    out3[1] = n > 0 ? (i64)leaves[0] : 0;
    out3[2] = n > 0 ? (i64)leaves[n - 1] : 0;
}

// ------------------------------------------------------------------------------------------
// a5: neighbour CSR.  grid.cpp:43-175
// ------------------------------------------------------------------------------------------
__global__ void k_map_build(const u64* keys, i64 v, HashTab t, int* cnt) {
    i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (i >= v) return;
    u64 slot;
    int r = tab_insert(t, keys[i], &slot);
    if (r < 0)
        cnt[1] = 1;
    else
        t.vals[slot] = (int32_t)i;
}

__constant__ int8_t c_child_off[24][3] = {
        {-1, 0, 0}, {-1, 1, 0}, {-1, 0, 1}, {-1, 1, 1}, {2, 0, 0},  {2, 1, 0},  {2, 0, 1},  {2, 1, 1},
        {0, -1, 0}, {1, -1, 0}, {0, -1, 1}, {1, -1, 1}, {0, 2, 0},  {1, 2, 0},  {0, 2, 1},  {1, 2, 1},
        {0, 0, -1}, {1, 0, -1}, {0, 1, -1}, {1, 1, -1}, {0, 0, 2},  {1, 0, 2},  {0, 1, 2},  {1, 1, 2}};
__constant__ int8_t c_nb_off[6][3] = {{-1, 0, 0}, {1, 0, 0}, {0, -1, 0},
                                      {0, 1, 0},  {0, 0, -1}, {0, 0, 1}};
__constant__ int8_t c_parent_koff[6][8] = {
        {-1, 0, -1, 1, -1, 2, -1, 3}, {0, -1, 1, -1, 2, -1, 3, -1}, {-1, -1, 0, 1, -1, -1, 2, 3},
        {0, 1, -1, -1, 2, 3, -1, -1}, {-1, -1, -1, -1, 0, 1, 2, 3}, {0, 1, 2, 3, -1, -1, -1, -1}};

// candidate c in [0,36): 0..5 same level, 6..29 child level, 30..35 parent level.
// returns neighbour index or -1; slot = kernel index of the hit.
__device__ inline int neighbor_candidate(const HashTab& t, u64 key, int x, int y, int z, int lev,
                                         int c, int& slot) {
    if (c < 6) {
        u64 nk = asr_coord_key(x + c_nb_off[c][0], y + c_nb_off[c][1], z + c_nb_off[c][2], lev);
        slot = 1 + c;
        return nk ? tab_find(t, nk) : -1;
    } else if (c < 30) {
        if (lev >= ASR_MAX_LEVEL) return -1;
        int j = c - 6;
        u64 nk = asr_coord_key(2 * x + c_child_off[j][0], 2 * y + c_child_off[j][1],
                               2 * z + c_child_off[j][2], lev + 1);
        slot = 7 + j;
        return nk ? tab_find(t, nk) : -1;
    } else {
        if (lev <= 0) return -1;
        int j = c - 30;
        u64 nk = asr_coord_key(x + c_nb_off[j][0], y + c_nb_off[j][1], z + c_nb_off[j][2], lev);
        if (!nk) return -1;
        int koff = c_parent_koff[j][nk & 7];
        if (koff < 0) return -1;  // sibling: its parent is this voxel's parent, never in the grid
        slot = 31 + 4 * j + koff;
        return tab_find(t, nk >> 3);
    }
}

// The hits of the counting pass are parked (neighbour index + slot, NB_STAGE per row) so that the fill pass is a
// copy for all but the few rows with more neighbours: the 8..13 hash probes per voxel are not repeated.
constexpr int NB_STAGE = 16;
__global__ void k_neighbors_count(const u64* keys, i64 v, HashTab t, i64* counts, u64* masks, int32_t* stage_idx,
                                  uint8_t* stage_slot) {
    i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (i > v) return;
    if (i == v) {
        counts[v] = 0;
        return;
    }
    u64 key = keys[i];
    int x, y, z, lev;
    asr_key_coord(key, x, y, z, lev);
    u64 m = 0;
    int n = 1;
#define ASR_NB_HIT(c_, idx_, slot_)                                   \
    {                                                                 \
        m |= u64(1) << (c_);                                          \
        if (stage_idx && n - 1 < NB_STAGE) {                          \
            stage_idx[i * NB_STAGE + (n - 1)] = (idx_);               \
            stage_slot[i * NB_STAGE + (n - 1)] = (uint8_t)(slot_);    \
        }                                                             \
        ++n;                                                          \
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        int slot = 0;
        const int idx = neighbor_candidate(t, key, x, y, z, lev, c, slot);
        if (idx >= 0) ASR_NB_HIT(c, idx, slot)
    }
    // The voxels of a grid are disjoint cells: where the same-level neighbour across a face exists,
    // neither its four children nor its parent can, so those five probes are skipped (36 -> ~13 probes
    // per voxel on a scan; the resulting mask is the same).
#pragma unroll 6
    for (int c = 6; c < 36; ++c) {
        const int face = c < 30 ? (c - 6) >> 2 : c - 30;
        if ((m >> face) & 1) continue;
        int slot = 0;
        const int idx = neighbor_candidate(t, key, x, y, z, lev, c, slot);
        if (idx >= 0) ASR_NB_HIT(c, idx, slot)
    }
#undef ASR_NB_HIT
    counts[i] = n;
    if (masks) masks[i] = m;
}
__global__ void k_neighbors_fill(const u64* keys, i64 v, HashTab t, const i64* rs, const u64* masks,
                                 const int32_t* stage_idx, const uint8_t* stage_slot, int32_t* nidx, uint8_t* nkidx) {
    i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (i >= v) return;
    i64 o = rs[i];
    const int n = (int)(rs[i + 1] - o);
    nidx[o] = (int32_t)i;
    nkidx[o] = 0;
    ++o;
    if (stage_idx && n - 1 <= NB_STAGE) {  // the candidate order of the counting pass is ascending slot order
        for (int j = 0; j < n - 1; ++j) {
            nidx[o + j] = stage_idx[i * NB_STAGE + j];
            nkidx[o + j] = stage_slot[i * NB_STAGE + j];
        }
        return;
    }
    u64 key = keys[i];
    int x, y, z, lev;
    asr_key_coord(key, x, y, z, lev);
    u64 m = masks ? masks[i] : ~u64(0);
    for (int c = 0; c < 36; ++c) {
        if (!((m >> c) & 1)) continue;
        int slot;
        int idx = neighbor_candidate(t, key, x, y, z, lev, c, slot);
        if (idx >= 0) {
            nidx[o] = idx;
            nkidx[o] = (uint8_t)slot;
            ++o;
        }
    }
}

// The same for a LIST of rows (a rank of the one-scan sharding builds the lists of the voxels it owns): counts of
// the other rows stay 0, so the row splits keep their full length and the entries of the listed rows are compact.
__device__ inline int neighbors_of_row(const u64* keys, i64 i, const HashTab& t, int32_t* nidx, uint8_t* nkidx) {
    const u64 key = keys[i];
    int x, y, z, lev;
    asr_key_coord(key, x, y, z, lev);
    u64 m = 0;
    int n = 0;
    if (nidx) {
        nidx[0] = (int32_t)i;
        nkidx[0] = 0;
    }
    ++n;
    for (int c = 0; c < 36; ++c) {
        if (c >= 6) {  // a same-level neighbour across a face rules out its children and its parent (see above)
            const int face = c < 30 ? (c - 6) >> 2 : c - 30;
            if ((m >> face) & 1) continue;
        }
        int slot = 0;
        const int idx = neighbor_candidate(t, key, x, y, z, lev, c, slot);
        if (idx < 0) continue;
        m |= u64(1) << c;
        if (nidx) {
            nidx[n] = idx;
            nkidx[n] = (uint8_t)slot;
        }
        ++n;
    }
    return n;
}
// public row lists / key lists are caller data: a bad entry must become an error, not an out-of-bounds write
__global__ void k_check_rows(const int32_t* rows, i64 nrows, i64 v, int* cnt) {
    const i64 r = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    const i64 q = rows[r];
    if (q < 0 || q >= v || (r > 0 && rows[r - 1] >= q)) cnt[3] = 1;  // in range, strictly ascending
}
__global__ void k_check_keys(const u64* keys, i64 n, int* cnt) {
    const i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 k = keys[i];
    // a location code: leading bit at position 3 * level, level <= ASR_MAX_LEVEL (0 is the tables' empty sentinel)
    if (k == 0 || (63 - __clzll((long long)k)) % 3 != 0) cnt[3] = 1;
}
__global__ void k_neighbors_count_rows(const u64* keys, HashTab t, const int32_t* rows, i64 nrows, i64* counts) {
    const i64 r = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    const i64 i = rows[r];
    counts[i] = neighbors_of_row(keys, i, t, nullptr, nullptr);
}
__global__ void k_neighbors_fill_rows(const u64* keys, HashTab t, const int32_t* rows, i64 nrows, const i64* rs,
                                      int32_t* nidx, uint8_t* nkidx) {
    const i64 r = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    const i64 i = rows[r];
    const i64 o = rs[i];
    (void)neighbors_of_row(keys, i, t, nidx + o, nkidx + o);
}

// ------------------------------------------------------------------------------------------
// The same for ALL grids of a hierarchy in one launch each (asr_geom_neighbors_build_batch): the coarse grids hold a few
// thousand voxels, whose 13-odd dependent probes cost a level-0 kernel's latency each when launched one by one.  Rows of
// all jobs share one index space, every job followed by one terminator entry, so that ONE exclusive scan gives every
// job's row splits (shifted by the job's first value) and the pair counts at the job boundaries.
// ------------------------------------------------------------------------------------------
constexpr int NB_MAX_JOBS = 8;
struct NbBatch {
    int n;
    i64 base[NB_MAX_JOBS + 1];  // first entry of each job in the shared index space; base[n] = total
    i64 v[NB_MAX_JOBS];
    const u64* keys[NB_MAX_JOBS];
    HashTab tab[NB_MAX_JOBS];
    i64* rs[NB_MAX_JOBS];
    int32_t* idx[NB_MAX_JOBS];
    uint8_t* kidx[NB_MAX_JOBS];
    const int32_t* owner[NB_MAX_JOBS];  // optional row filter (one rank of a sharded cloud builds its own rows)
    int me[NB_MAX_JOBS];
};
__device__ inline int nb_job_of(const NbBatch& b, i64 e) {
    int j = 0;
    while (j + 1 < b.n && e >= b.base[j + 1]) ++j;
    return j;
}
__global__ void k_map_build_batch(NbBatch b, int* cnt) {
    const i64 e = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (e >= b.base[b.n]) return;
    const int j = nb_job_of(b, e);
    const i64 i = e - b.base[j];
    if (i >= b.v[j]) return;
    u64 slot;
    if (tab_insert(b.tab[j], b.keys[j][i], &slot) < 0)
        cnt[1] = 1;
    else
        b.tab[j].vals[slot] = (int32_t)i;
}
__global__ void k_neighbors_count_batch(NbBatch b, i64* counts, u64* masks, int32_t* stage_idx, uint8_t* stage_slot) {
    const i64 e = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (e >= b.base[b.n]) return;
    const int j = nb_job_of(b, e);
    const i64 i = e - b.base[j];
    if (i >= b.v[j]) {  // the job's terminator
        counts[e] = 0;
        return;
    }
    if (b.owner[j] && b.owner[j][i] != b.me[j]) {
        counts[e] = 0;
        masks[e] = 0;
        return;
    }
    const HashTab t = b.tab[j];
    const u64 key = b.keys[j][i];
    int x, y, z, lev;
    asr_key_coord(key, x, y, z, lev);
    u64 m = 0;
    int n = 1;
#pragma unroll 6
    for (int c = 0; c < 36; ++c) {
        if (c >= 6) {  // a same-level neighbour across a face rules out its children and its parent (k_neighbors_count)
            const int face = c < 30 ? (c - 6) >> 2 : c - 30;
            if ((m >> face) & 1) continue;
        }
        int slot = 0;
        const int idx = neighbor_candidate(t, key, x, y, z, lev, c, slot);
        if (idx < 0) continue;
        m |= u64(1) << c;
        if (n - 1 < NB_STAGE) {
            stage_idx[e * NB_STAGE + (n - 1)] = idx;
            stage_slot[e * NB_STAGE + (n - 1)] = (uint8_t)slot;
        }
        ++n;
    }
    counts[e] = n;
    masks[e] = m;
}
// totals[j] = pairs of job j (scan value at the next job's first entry minus the job's own)
__global__ void k_nb_totals(NbBatch b, const i64* scan, i64* totals) {
    const int j = threadIdx.x;
    if (j < b.n) totals[j] = scan[b.base[j] + b.v[j]] - scan[b.base[j]];
}
__global__ void k_neighbors_fill_batch(NbBatch b, const i64* scan, const u64* masks, const int32_t* stage_idx,
                                       const uint8_t* stage_slot) {
    const i64 e = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (e >= b.base[b.n]) return;
    const int j = nb_job_of(b, e);
    const i64 i = e - b.base[j];
    const i64 first = scan[b.base[j]];
    i64 o = scan[e] - first;
    b.rs[j][i] = o;  // (the terminator writes rs[v] = pairs of the job)
    if (i >= b.v[j]) return;
    const int n = (int)(scan[e + 1] - first - o);
    if (n == 0) return;  // a row of another rank
    int32_t* nidx = b.idx[j];
    uint8_t* nkidx = b.kidx[j];
    nidx[o] = (int32_t)i;
    nkidx[o] = 0;
    ++o;
    if (n - 1 <= NB_STAGE) {  // the candidate order of the counting pass is ascending slot order
        for (int q = 0; q < n - 1; ++q) {
            nidx[o + q] = stage_idx[e * NB_STAGE + q];
            nkidx[o + q] = stage_slot[e * NB_STAGE + q];
        }
        return;
    }
    const HashTab t = b.tab[j];
    const u64 key = b.keys[j][i];
    int x, y, z, lev;
    asr_key_coord(key, x, y, z, lev);
    const u64 m = masks[e];
    for (int c = 0; c < 36; ++c) {
        if (!((m >> c) & 1)) continue;
        int slot;
        const int idx = neighbor_candidate(t, key, x, y, z, lev, c, slot);
        if (idx >= 0) {
            nidx[o] = idx;
            nkidx[o] = (uint8_t)slot;
            ++o;
        }
    }
}

// ------------------------------------------------------------------------------------------
// a6: CombineSiblings.  grid.cpp:177-243
// ------------------------------------------------------------------------------------------
__device__ inline bool merged_head(const u64* keys, i64 v, i64 i) {
    u64 k = keys[i];
    return (k & 7) == 0 && i + 7 < v && keys[i + 7] == (k | 7);
}
// state: 0 = carried, 1 = merged head, 2 = merged member (dropped)
__device__ inline int coarsen_state(const u64* keys, i64 v, i64 i) {
    u64 k = keys[i];
    if ((k & 7) == 0) return merged_head(keys, v, i) ? 1 : 0;
    i64 h = i - (i64)(k & 7);
    if (h >= 0 && keys[h] == (k & ~u64(7)) && merged_head(keys, v, h)) return 2;
    return 0;
}
__global__ void k_coarsen_count(const u64* keys, i64 v, int* cnt) {
    i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    bool keep = i < v && coarsen_state(keys, v, i) != 2;
    (void)block_append(keep, &cnt[5]);
}
__global__ void k_coarsen_emit(const u64* keys, i64 v, u64* out_keys, int32_t* out_src, int* cnt) {
    i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    int st = i < v ? coarsen_state(keys, v, i) : 2;
    int pos = block_append(st != 2, &cnt[5]);
    if (st == 2) return;
    out_keys[pos] = st == 1 ? keys[i] >> 3 : keys[i];
    out_src[pos] = (int32_t)i;
}
// CombineSiblings without a sort (round 4).  The coarse key set is the kept keys plus one parent per merged group; both
// sub-sequences are already sorted in input order, and location codes sort level-major, so an emitted key's position in
// the sorted output is  (#kept keys smaller) + (#parents smaller):
//   kept key e at input index i:        kept_before[i] + heads_before[lower_bound(keys, e << 3)]
//   parent e = keys[i] >> 3 of a head:  heads_before[i] + kept_before[lower_bound(keys, e)]
// (every key of a lower level is smaller; a head with key < 8 e has a parent < e).  One flag pass, one exclusive scan of
// the packed (kept | heads << 32) counts, one placement pass with a binary search -- instead of a multi-pass radix sort
// whose dozen launches per level dominate the four small coarsening steps.
__global__ void k_coarsen_flags(const u64* keys, i64 v, u64* packed) {
    const i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (i > v) return;
    u64 f = 0;
    if (i < v) {
        const int st = coarsen_state(keys, v, i);
        f = st == 0 ? u64(1) : (st == 1 ? (u64(1) << 32) : 0);
    }
    packed[i] = f;
}
__global__ void k_coarsen_place(const u64* keys, i64 v, const u64* pre, u64* out_keys, int32_t* out_src) {
    const i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (i >= v) return;
    const u64 mine = pre[i + 1] - pre[i];  // 1: kept, 1 << 32: head, 0: merged member
    if (!mine) return;
    const bool head = (mine >> 32) != 0;
    const u64 key = keys[i];
    const u64 e = head ? key >> 3 : key;
    const u64 probe = head ? e : key << 3;
    i64 lo = 0, hi = v;  // first input key >= probe (a level-21 key has no children: its probe overflows to "beyond the end")
    if (!head && asr_key_level(key) >= ASR_MAX_LEVEL) {
        lo = v;
    } else {
        while (lo < hi) {
            const i64 mid = (lo + hi) >> 1;
            if (keys[mid] < probe)
                lo = mid + 1;
            else
                hi = mid;
        }
    }
    const u64 at = pre[lo];
    const i64 pos = head ? (i64)(pre[i] >> 32) + (i64)(at & 0xffffffffu) : (i64)(pre[i] & 0xffffffffu) + (i64)(at >> 32);
    out_keys[pos] = e;
    out_src[pos] = (int32_t)i;
}
__global__ void k_coarsen_up(const u64* keys, i64 v, const int32_t* sorted_src, i64 v_out,
                             int32_t* up_idx, uint8_t* up_kidx) {
    i64 p = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (p >= v_out) return;
    i64 i = sorted_src[p];
    if (merged_head(keys, v, i)) {
        for (int j = 0; j < 8; ++j) {
            up_idx[i + j] = (int32_t)p;
            up_kidx[i + j] = (uint8_t)j;
        }
    } else {
        up_idx[i] = (int32_t)p;
        up_kidx[i] = 8;
    }
}
// The inverted up lists ("down" lists, net_definitions_torch.py:548-559) follow from the same facts without the
// generic inversion (histogram + scan + stable sort): coarse voxel p owns either its eight consecutive children
// (slots 0..7) or the one carried voxel (slot 8), in ascending fine index -- the order of
// open3d::invert_neighbors_list.  cnt[p] feeds the exclusive scan that gives the row splits.
__global__ void k_coarsen_down_count(const u64* keys, i64 v, const int32_t* sorted_src, i64 v_out, i64* cnt) {
    i64 p = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (p > v_out) return;
    cnt[p] = p < v_out ? (merged_head(keys, v, sorted_src[p]) ? 8 : 1) : 0;
}
__global__ void k_coarsen_down_fill(const u64* keys, i64 v, const int32_t* sorted_src, i64 v_out, const i64* rs,
                                    int32_t* down_idx, uint8_t* down_kidx) {
    i64 p = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (p >= v_out) return;
    const i64 i = sorted_src[p], o = rs[p];
    if (merged_head(keys, v, i)) {
        for (int j = 0; j < 8; ++j) {
            down_idx[o + j] = (int32_t)(i + j);
            down_kidx[o + j] = (uint8_t)j;
        }
    } else {
        down_idx[o] = (int32_t)i;
        down_kidx[o] = 8;
    }
}
__global__ void k_iota64(i64* out, i64 n) {
    i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (i < n) out[i] = i;
}

// ------------------------------------------------------------------------------------------
// a7: voxel centres / sizes.  grid.cpp:251-268, octree.h:78-93 (double arithmetic)
// ------------------------------------------------------------------------------------------
__global__ void k_voxel_info(asr_octree_frame f, const u64* keys, i64 v, float* centers,
                             float* sizes) {
    i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (i >= v) return;
    int x, y, z, lev;
    asr_key_coord(keys[i], x, y, z, lev);
    int s = ASR_MAX_LEVEL - lev;
    double half = 0.5 * (double)(u64(1) << s);
    double vs = (double)f.voxel_size[ASR_MAX_LEVEL];
    if (centers) {
        centers[3 * i + 0] = (float)((((x << s) - f.offset[0]) + half) * vs);
        centers[3 * i + 1] = (float)((((y << s) - f.offset[1]) + half) * vs);
        centers[3 * i + 2] = (float)((((z << s) - f.offset[2]) + half) * vs);
    }
    if (sizes) sizes[i] = f.voxel_size[lev];
}

// ------------------------------------------------------------------------------------------
// a8: multi radius search.  Points are sorted by their level-21 Morton code inside the octree
// frame; for every level used by a query a hash table maps cell -> [start,end) in that order;
// a query of radius r looks at the 3^3 cells of the deepest level whose cell size is >= r.
// ------------------------------------------------------------------------------------------
__global__ void k_point_codes(asr_octree_frame f, const float* pts, i64 n, u64* codes,
                              int32_t* ids, int* cnt) {
    i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (!(isfinite(pts[3 * i]) && isfinite(pts[3 * i + 1]) && isfinite(pts[3 * i + 2]))) cnt[11] = 1;
    int x, y, z;
    frame_coord(f, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], ASR_MAX_LEVEL, x, y, z);
    const int lim = (1 << ASR_MAX_LEVEL) - 1;
    x = min(max(x, 0), lim);
    y = min(max(y, 0), lim);
    z = min(max(z, 0), lim);
    codes[i] = asr_morton3d((u64)x, (u64)y, (u64)z);
    ids[i] = (int32_t)i;
}
__global__ void k_gather_points(const float* pts, const int32_t* ids, i64 n, float4* sorted, int32_t* rank,
                                const float* radii = nullptr, float* srad = nullptr) {
    i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t id = ids[i];
    sorted[i] = make_float4(pts[3 * (i64)id], pts[3 * (i64)id + 1], pts[3 * (i64)id + 2],
                            __int_as_float(id));
    if (rank) rank[id] = (int32_t)i;  // original index -> position in Morton order
    if (srad) srad[i] = radii[id];    // per-point radii in Morton order (k_gather_radii in the same pass)
}
// per-point radii in Morton order (the compat factor of a pair then reads next to its neighbours)
__global__ void k_gather_radii(const float* radii, const int32_t* ids, i64 n, float* out) {
    i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (i < n) out[i] = radii[ids[i]];
}
__device__ inline int query_level(const asr_octree_frame& f, float r) {
    int lev = 0;
    while (lev < ASR_MAX_LEVEL && f.voxel_size[lev + 1] >= r) ++lev;
    return lev;
}
// grid-stride + block reduction: one atomic pair per BLOCK (same-address atomics retire at ~88 / us, one
// per wave was 0.9 ms for 2.6 M voxels)
__global__ __launch_bounds__(256) void k_query_levels(asr_octree_frame f, const float* sizes, i64 v, int* cnt) {
    __shared__ int s_lo[4], s_hi[4];
    int lo = ASR_MAX_LEVEL, hi = 0;
    for (i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x; i < v; i += (i64)gridDim.x * blockDim.x) {
        const int l = query_level(f, sizes[i]);
        lo = min(lo, l);
        hi = max(hi, l);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, __shfl_xor(lo, o, 64));
        hi = max(hi, __shfl_xor(hi, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        s_lo[threadIdx.x >> 6] = lo;
        s_hi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMin(&cnt[6], min(min(s_lo[0], s_lo[1]), min(s_lo[2], s_lo[3])));
        atomicMax(&cnt[7], max(max(s_hi[0], s_hi[1]), max(s_hi[2], s_hi[3])));
    }
}
// boundaries of the sorted code array: cell (prefix,level) starts / ends at i
template <bool COUNT_ONLY>
__global__ __launch_bounds__(256) void k_cell_bounds(const u64* codes, i64 n, int lmin, int lmax, HashTab t,
                                                     int32_t* start, int32_t* end, int* cnt, int* level_cnt) {
    __shared__ int s_sum[4];
    __shared__ int s_lvl[ASR_MAX_LEVEL + 1];  // COUNT_ONLY: cells per level (the host picks the hashed levels)
    if (COUNT_ONLY) {
        if (threadIdx.x <= ASR_MAX_LEVEL) s_lvl[threadIdx.x] = 0;
        __syncthreads();
    }
    int local = 0;
    for (i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x; i <= n; i += (i64)gridDim.x * blockDim.x) {
        u64 a = i > 0 ? codes[i - 1] : 0, b = i < n ? codes[i] : 0;
        for (int l = lmin; l <= lmax; ++l) {
            int s = 3 * (ASR_MAX_LEVEL - l);
            u64 pa = (s >= 63) ? 0 : (a >> s), pb = (s >= 63) ? 0 : (b >> s);
            bool has_a = i > 0, has_b = i < n;
            if (has_a && has_b && pa == pb) continue;
            u64 marker = u64(1) << (3 * l);
            if (COUNT_ONLY) {
                if (has_b) {
                    ++local;
                    atomicAdd(&s_lvl[l], 1);
                }
            } else {
                u64 slot;
                if (has_a) {
                    if (tab_insert(t, pa | marker, &slot) < 0)
                        cnt[1] = 1;
                    else
                        end[slot] = (int32_t)i;
                }
                if (has_b) {
                    if (tab_insert(t, pb | marker, &slot) < 0)
                        cnt[1] = 1;
                    else
                        start[slot] = (int32_t)i;
                }
            }
        }
    }
    if (COUNT_ONLY) {  // one atomic per block: 10^7 same-address atomics cost 2 ms
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o, 64);
        if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = local;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(&cnt[8], s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]);
        if (level_cnt && threadIdx.x <= ASR_MAX_LEVEL && s_lvl[threadIdx.x]) atomicAdd(&level_cnt[threadIdx.x], s_lvl[threadIdx.x]);
    }
}

__device__ inline float sqdist3(float ax, float ay, float az, float bx, float by, float bz) {
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    float s = dx * dx;
    s = s + dy * dy;
    s = s + dz * dz;
    return s;
}

// One WAVE per query: lanes 0..26 look up the 3^3 cells, the candidate points of all cells are
// then walked 64 at a time (flattened through a prefix of the cell counts).  Row lengths range
// from 0 to tens of thousands (a coarse voxel near the surface sees every point within one voxel
// size), so a thread-per-query loop is dominated by its longest row.
// MODE 0: count only (KDTree::ComputeRadiusNeighbors).
// MODE 2: single pass of the aggregation search.  Hits are collected in LDS; a row of at most
// RADIUS_LIGHT hits is ranked inside the wave (rank = number of smaller keys, keys are unique) and
// written, sorted, to the row's fixed slot tmp[q * RADIUS_LIGHT ..].  Rows with more hits, and rows
// whose 27 cells hold more than RADIUS_GIANT candidates (a coarse voxel next to a dense region: up
// to 10^5 candidates, which one wave would walk for milliseconds), go to the heavy list and are
// processed by k_radius_heavy with RADIUS_SPLIT blocks per row.
constexpr int RADIUS_LIGHT = 128;
constexpr int RADIUS_GIANT = 4096;
constexpr int RADIUS_SPLIT = 64;

// lanes 0..26 look up the 3^3 cells around the query; returns the candidate total, fills the
// per-wave prefix / begin tables
// Point ranges of cells.  Levels up to `lhash` are in the hash table; finer levels -- where a cloud with very dense
// spots would put nearly every point into a cell of its own, ten million entries per level -- are looked up by two
// binary searches in the Morton-sorted level-21 codes (the points of a cell are contiguous in that order).
struct CellIndex {
    HashTab tab;
    const int32_t* start;
    const int32_t* end;
    const u64* codes;  // sorted (on the bits of the finest indexed level)
    int n;
    int lhash;
};
__device__ inline void cell_range(const CellIndex& ci, u64 cell, int lev, int& b, int& cnt) {
    b = 0;
    cnt = 0;
    if (lev <= ci.lhash) {
        const i64 slot = tab_find_slot(ci.tab, cell | (u64(1) << (3 * lev)));
        if (slot >= 0) {
            b = ci.start[slot];
            cnt = ci.end[slot] - b;
        }
        return;
    }
    const int s = 3 * (ASR_MAX_LEVEL - lev);
    int lo = 0, hi = ci.n;  // first point whose level-`lev` cell is >= cell
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((ci.codes[mid] >> s) < cell)
            lo = mid + 1;
        else
            hi = mid;
    }
    int lo2 = lo, hi2 = ci.n;  // first point whose cell is > cell
    while (lo2 < hi2) {
        const int mid = (lo2 + hi2) >> 1;
        if ((ci.codes[mid] >> s) <= cell)
            lo2 = mid + 1;
        else
            hi2 = mid;
    }
    b = lo;
    cnt = lo2 - lo;
}

template <int NCELL>
__device__ inline int cells_prefix(int b, int n, int lane, int* s_pref, int* s_beg) {
    int pre = n;  // inclusive prefix of n over lanes 0..NCELL-1
#pragma unroll
    for (int o = 1; o < (NCELL > 32 ? 64 : 32); o <<= 1) {
        int up = __shfl_up(pre, o, 64);
        if (lane >= o) pre += up;
    }
    if (lane < NCELL) {
        s_pref[lane + 1] = pre;
        s_beg[lane] = b;
    }
    if (lane == 0) s_pref[0] = 0;
    const int total = __shfl(pre, NCELL - 1, 64);
    __builtin_amdgcn_wave_barrier();
    return total;
}
__device__ inline int radius_cells(const asr_octree_frame& f, const CellIndex& ci, float cx, float cy, float cz,
                                   float r, int lane, int* s_pref, int* s_beg) {
    const int lev = query_level(f, r);
    int x, y, z;
    frame_coord(f, cx, cy, cz, lev, x, y, z);
    const int lim = (1 << lev) - 1;
    int b = 0, n = 0;
    if (lane < 27) {
        int xx = x + lane % 3 - 1, yy = y + (lane / 3) % 3 - 1, zz = z + lane / 9 - 1;
        if (xx >= 0 && yy >= 0 && zz >= 0 && xx <= lim && yy <= lim && zz <= lim)
            cell_range(ci, asr_morton3d((u64)xx, (u64)yy, (u64)zz), lev, b, n);
    }
    return cells_prefix<27>(b, n, lane, s_pref, s_beg);
}

// ------------------------------------------------------------------------------------------
// ALIGNED queries (the aggregation search proper): query q is the centre of the grid voxel `keys[q]` = cell
// (x, y, z) of level L with radius = that level's cell size s.  The ball is then covered EXACTLY by the 4 x 4 x 4
// half-size cells (level L + 1) [2x-1, 2x+2]^3 -- a cube of side 2 s holding 1.9x the hits instead of the 6.4x of
// the 3 x 3 x 3 full-size cells.  "Exactly" has no margin, and the cell of a point (floor of a float product) and
// the centre of a voxel (a double rounded to float) carry rounding errors of up to 3 M 2^-24 level-21 units
// (M = largest coordinate magnitude in those units): a hit may sit in one of the first T = 2 + floor(3 M 2^-24)
// level-21 layers beyond a face of the cube -- then within sqrt(2 S T) + 2 T units of the axis through the voxel
// centre (S = voxel size in units), i.e. the "pole" of the ball.  k_radius_extras finds exactly those (point,
// voxel) pairs -- integer tests on the point codes, a handful of candidates per million points -- and the query
// kernels take them as additional candidates.  Levels whose half cell is narrower than 16 T units (and level 21)
// keep the 3 x 3 x 3 block of full-size cells, whose margin is half a cell.
// ------------------------------------------------------------------------------------------
constexpr int EXTRA_CAP = 1 << 16;
struct AlignedQ {
    const u64* keys;     // sorted voxel keys, one per query
    const int2* extras;  // (query, position in Morton order) pairs beyond the cube
    const int* extras_cnt;
    int lhalf_max;       // finest voxel level that uses half cells
};
__device__ inline int aligned_cells(const CellIndex& ci, const AlignedQ& aq, u64 key, int lane, int* s_pref, int* s_beg) {
    int x, y, z, lev;
    asr_key_coord(key, x, y, z, lev);
    int b = 0, n = 0;
    if (lev <= aq.lhalf_max) {
        const int lim = (2 << lev) - 1;
        const int xx = 2 * x - 1 + (lane & 3), yy = 2 * y - 1 + ((lane >> 2) & 3), zz = 2 * z - 1 + (lane >> 4);
        if (xx >= 0 && yy >= 0 && zz >= 0 && xx <= lim && yy <= lim && zz <= lim)
            cell_range(ci, asr_morton3d((u64)xx, (u64)yy, (u64)zz), lev + 1, b, n);
    } else if (lane < 27) {
        const int lim = (1 << lev) - 1;
        const int xx = x + lane % 3 - 1, yy = y + (lane / 3) % 3 - 1, zz = z + lane / 9 - 1;
        if (xx >= 0 && yy >= 0 && zz >= 0 && xx <= lim && yy <= lim && zz <= lim)
            cell_range(ci, asr_morton3d((u64)xx, (u64)yy, (u64)zz), lev, b, n);
    }
    return cells_prefix<64>(b, n, lane, s_pref, s_beg);
}
// candidate i of the concatenated cell ranges
template <int NCELL = 27>
__device__ inline float4 radius_candidate(const float4* sorted, const int* s_pref, const int* s_beg, int i,
                                          int* pos_out = nullptr) {
    int lo = 0, hi = NCELL;  // cell c with pref[c] <= i < pref[c+1]
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (s_pref[mid] <= i)
            lo = mid;
        else
            hi = mid;
    }
    const int pos = s_beg[lo] + (i - s_pref[lo]);  // position in Morton order
    if (pos_out) *pos_out = pos;
    return sorted[pos];
}

// the (voxel, point) pairs of the rounding margin, see above.  One thread per point (position s in Morton order).
struct ExtraParams {
    int lmin, lmax;  // voxel levels to test (<= lhalf_max)
    int T;
    int rho[ASR_MAX_LEVEL + 1];  // pole radius in level-21 units per voxel level
};
__global__ void k_radius_extras(const u64* codes, const float4* sorted, i64 n, const u64* keys, i64 v, const float* centers,
                                const float* sizes, ExtraParams ep, int2* out, int* out_cnt) {
    const i64 s = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (s >= n) return;
    const u64 code = codes[s];
    const int c[3] = {(int)asr_compact21(code), (int)asr_compact21(code >> 1), (int)asr_compact21(code >> 2)};
    for (int lev = ep.lmin; lev <= ep.lmax; ++lev) {
        const int sh = ASR_MAX_LEVEL - lev, H = 1 << (sh - 1), S = 1 << sh;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int A = c[a] + H, m = A & (S - 1);
            int xa;
            if (m < ep.T)
                xa = (A >> sh) - 2;  // beyond the + face of voxel xa
            else if (m >= S - ep.T)
                xa = (A >> sh) + 1;  // beyond the - face
            else
                continue;
            const int b0 = (a + 1) % 3, b1 = (a + 2) % 3;
            const int y0 = c[b0] >> sh, y1 = c[b1] >> sh;
            if (abs(c[b0] - (y0 * S + H)) > ep.rho[lev] || abs(c[b1] - (y1 * S + H)) > ep.rho[lev]) continue;
            int xyz[3];
            xyz[a] = xa;
            xyz[b0] = y0;
            xyz[b1] = y1;
            const u64 key = asr_coord_key(xyz[0], xyz[1], xyz[2], lev);
            if (!key) continue;
            i64 lo = 0, hi = v;  // first voxel key >= key
            while (lo < hi) {
                const i64 mid = (lo + hi) >> 1;
                if (keys[mid] < key)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            if (lo >= v || keys[lo] != key) continue;
            const float4 pt = sorted[s];
            const float r = sizes[lo];
            if (sqdist3(pt.x, pt.y, pt.z, centers[3 * lo], centers[3 * lo + 1], centers[3 * lo + 2]) < r * r) {
                const int o = atomicAdd(out_cnt, 1);
                if (o < EXTRA_CAP) out[o] = make_int2((int)lo, (int)s);
            }
        }
    }
}

// end of a row of the aggregation search (MODE 2): the wave holds `found` hits, the first RADIUS_LIGHT of them as
// (distance bits | index) keys + Morton positions in LDS.  Heavy rows go to the list of k_radius_heavy; the others
// are ranked inside the wave (rank = number of smaller keys, keys are unique) and written, sorted, to the row's fixed
// slot tmp[q * RADIUS_LIGHT ..] as (distance, position in Morton order): k_radius_place turns the position back into
// the index with a clustered read.
__device__ __forceinline__ void radius_row_out(i64 q, i64 found, bool heavy, int lane, const u64* s_keys, const int* s_pos,
                                      i64* counts, u64* tmp, int32_t* heavy_out, int* heavy_cnt, uint8_t* is_heavy) {
    if (heavy) {  // counted and written by k_radius_heavy
        if (lane == 0) {
            counts[q] = 0;
            is_heavy[q] = 1;
            heavy_out[atomicAdd(heavy_cnt, 1)] = (int32_t)q;
        }
        return;
    }
    if (lane == 0) {
        counts[q] = found;
        is_heavy[q] = 0;
    }
    __builtin_amdgcn_wave_barrier();
    const int h = (int)found;
    u64 mine[RADIUS_LIGHT / 64];
    int mypos[RADIUS_LIGHT / 64];
    int rank[RADIUS_LIGHT / 64];
#pragma unroll
    for (int u = 0; u < RADIUS_LIGHT / 64; ++u) {
        mine[u] = lane + 64 * u < h ? s_keys[lane + 64 * u] : 0;
        mypos[u] = lane + 64 * u < h ? s_pos[lane + 64 * u] : 0;
        rank[u] = 0;
    }
    if (h <= 64) {  // (wave uniform; 19 of 20 rows)
        for (int j = 0; j < h; ++j) rank[0] += s_keys[j] < mine[0];  // LDS broadcast reads
    } else {
        for (int j = 0; j < h; ++j) {
            const u64 kj = s_keys[j];
#pragma unroll
            for (int u = 0; u < RADIUS_LIGHT / 64; ++u) rank[u] += kj < mine[u];
        }
    }
#pragma unroll
    for (int u = 0; u < RADIUS_LIGHT / 64; ++u)
        if (lane + 64 * u < h) tmp[q * RADIUS_LIGHT + rank[u]] = (mine[u] & 0xffffffff00000000ull) | (u32)mypos[u];
}

// one query of k_radius_query by one wave; s_*: the wave's LDS rows
template <int MODE, bool ALIGNED>
__device__ __forceinline__ void radius_query_one(const asr_octree_frame& f, const float4* sorted, const float* centers,
                                                 const float* sizes, i64 q, const CellIndex& ci, const AlignedQ& aq,
                                                 i64* counts, u64* tmp, int32_t* heavy_out, int* heavy_cnt, uint8_t* is_heavy,
                                                 int lane, int* s_pref, int* s_beg, u64* s_keys, int* s_pos) {
    constexpr int NCELL = ALIGNED ? 64 : 27;
    const float cx = centers[3 * q], cy = centers[3 * q + 1], cz = centers[3 * q + 2];
    const float r = sizes[q];
    const float r2 = r * r;
    const int total = ALIGNED ? aligned_cells(ci, aq, aq.keys[q], lane, s_pref, s_beg)
                              : radius_cells(f, ci, cx, cy, cz, r, lane, s_pref, s_beg);
    i64 found = 0;
    bool heavy = MODE == 2 && total > RADIUS_GIANT;
    for (int i0 = 0; i0 < total && !heavy; i0 += 64) {
        const int i = i0 + lane;
        bool hit = false;
        float d = 0.f;
        int id = 0, pos = 0;
        if (i < total) {
            const float4 pt = radius_candidate<NCELL>(sorted, s_pref, s_beg, i, &pos);
            d = sqdist3(pt.x, pt.y, pt.z, cx, cy, cz);
            hit = d < r2;
            id = __float_as_int(pt.w);
        }
        const unsigned long long m = __ballot(hit);
        if (MODE == 2 && hit) {
            const i64 o = found + __popcll(m & ((1ull << lane) - 1));
            if (o < RADIUS_LIGHT) {
                s_keys[o] = ((u64)__float_as_uint(d) << 32) | (u32)id;
                s_pos[o] = pos;
            }
        }
        found += __popcll(m);
        if (MODE == 2 && found > RADIUS_LIGHT) heavy = true;
    }
    if (ALIGNED) {  // the pairs of the rounding margin (k_radius_extras): normally none at all
        const int nex = min(*aq.extras_cnt, EXTRA_CAP);
        for (int e0 = 0; e0 < nex && !heavy; e0 += 64) {
            const int e = e0 + lane;
            bool hit = false;
            float d = 0.f;
            int id = 0, pos = 0;
            if (e < nex) {
                const int2 it = aq.extras[e];
                if (it.x == (int)q) {
                    pos = it.y;
                    const float4 pt = sorted[pos];
                    d = sqdist3(pt.x, pt.y, pt.z, cx, cy, cz);
                    hit = d < r2;
                    id = __float_as_int(pt.w);
                }
            }
            const unsigned long long m = __ballot(hit);
            if (MODE == 2 && hit) {
                const i64 o = found + __popcll(m & ((1ull << lane) - 1));
                if (o < RADIUS_LIGHT) {
                    s_keys[o] = ((u64)__float_as_uint(d) << 32) | (u32)id;
                    s_pos[o] = pos;
                }
            }
            found += __popcll(m);
            if (MODE == 2 && found > RADIUS_LIGHT) heavy = true;
        }
    }
    if (MODE == 0) {
        if (lane == 0) counts[q] = found;
        return;
    }
    radius_row_out(q, found, heavy, lane, s_keys, s_pos, counts, tmp, heavy_out, heavy_cnt, is_heavy);
}

// all queries 0..v (launch of ceil((v + 1) / 4) blocks), one query per wave
template <int MODE, bool ALIGNED>
__global__ __launch_bounds__(256) void k_radius_query(asr_octree_frame f, const float4* sorted,
                                                      const float* centers, const float* sizes,
                                                      i64 v, CellIndex ci, AlignedQ aq, i64* counts, u64* tmp,
                                                      int32_t* heavy_out, int* heavy_cnt,
                                                      uint8_t* is_heavy) {
    constexpr int NCELL = ALIGNED ? 64 : 27;
    __shared__ int s_pref[4][NCELL + 1];
    __shared__ int s_beg[4][NCELL];
    __shared__ u64 s_keys[MODE == 2 ? 4 : 1][RADIUS_LIGHT];
    __shared__ int s_pos[MODE == 2 ? 4 : 1][RADIUS_LIGHT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kw = MODE == 2 ? wave : 0;
    const i64 q = blockIdx.x * (i64)4 + wave;
    if (q == v && lane == 0) counts[v] = 0;
    if (q >= v) return;
    radius_query_one<MODE, ALIGNED>(f, sorted, centers, sizes, q, ci, aq, counts, tmp, heavy_out, heavy_cnt, is_heavy, lane,
                                    s_pref[wave], s_beg[wave], s_keys[kw], s_pos[kw]);
}

// Cell ranges of the heavy rows, looked up ONCE per row (one wave each) and kept for the RADIUS_SPLIT x 4 waves of the
// counting pass and of the filling pass, which used to repeat the 27 / 64 table look-ups each (2 x 2 x 10^8 probes at
// 10 M points, three times the light rows' own)
constexpr int HC_LD = 65;  // ints per row in hc_pref (prefix, NCELL + 1 used) and hc_beg (NCELL used)
template <bool ALIGNED>
__global__ __launch_bounds__(256) void k_radius_heavy_cells(asr_octree_frame f, const float* centers, const float* sizes,
                                                            const int32_t* heavy, int nh, CellIndex ci, AlignedQ aq,
                                                            int* hc_pref, int* hc_beg) {
    constexpr int NCELL = ALIGNED ? 64 : 27;
    __shared__ int s_pref[4][NCELL + 1];
    __shared__ int s_beg[4][NCELL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x * 4 + wave;
    if (j >= nh) return;
    const i64 q = heavy[j];
    if (ALIGNED)
        aligned_cells(ci, aq, aq.keys[q], lane, s_pref[wave], s_beg[wave]);
    else
        radius_cells(f, ci, centers[3 * q], centers[3 * q + 1], centers[3 * q + 2], sizes[q], lane, s_pref[wave], s_beg[wave]);
    __builtin_amdgcn_wave_barrier();
    if (lane <= NCELL) hc_pref[(i64)j * HC_LD + lane] = s_pref[wave][lane];
    if (NCELL == 64 && lane == 0) hc_pref[(i64)j * HC_LD + 64] = s_pref[wave][64];
    if (lane < NCELL) hc_beg[(i64)j * HC_LD + lane] = s_beg[wave][lane];
}

// Heavy rows: RADIUS_SPLIT blocks per row, each block walks 1/RADIUS_SPLIT of the candidates.
// FILL = false: accumulates the hit count into counts[q]; FILL = true: writes the (distance, index)
// keys at hoff[j] + cursor (any order: the rows are sorted by a segmented sort afterwards).
template <bool FILL, bool ALIGNED>
__global__ __launch_bounds__(256) void k_radius_heavy(asr_octree_frame f, const float4* sorted, const float* centers,
                                                      const float* sizes, const int32_t* heavy, CellIndex ci, AlignedQ aq,
                                                      i64* counts, const i64* hoff, int* cursor, u64* keys_out,
                                                      int32_t* row_out, const int* hc_pref, const int* hc_beg,
                                                      int idx_bits) {
    // idx_bits > 0 (FILL): ONE sortable key per hit, (row j, squared distance bits, index) = row_bits + 31 + idx_bits
    // <= 64 bits (distances are >= 0: the sign bit is dropped) -- the rows are then ordered by a single keys-only
    // radix sort; idx_bits == 0: (distance bits << 32 | index) + the row in row_out, two stable sorts
    constexpr int NCELL = ALIGNED ? 64 : 27;
    __shared__ int s_pref[4][NCELL + 1];
    __shared__ int s_beg[4][NCELL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x;
    const int total = hc_pref[(i64)j * HC_LD + NCELL];  // (k_radius_heavy_cells)
    int per = (total + RADIUS_SPLIT - 1) / RADIUS_SPLIT;
    per = (per + 255) & ~255;  // whole 4-wave rounds
    const int lo = blockIdx.y * per;
    const int hi = min(total, lo + per);
    if (lo >= total && !(ALIGNED && blockIdx.y == 0)) return;  // (block 0 also walks the pairs of the rounding margin)
    const i64 q = heavy[j];
    const float cx = centers[3 * q], cy = centers[3 * q + 1], cz = centers[3 * q + 2];
    const float r = sizes[q];
    const float r2 = r * r;
    if (lane <= NCELL) s_pref[wave][lane] = hc_pref[(i64)j * HC_LD + lane];
    if (NCELL == 64 && lane == 0) s_pref[wave][64] = total;
    if (lane < NCELL) s_beg[wave][lane] = hc_beg[(i64)j * HC_LD + lane];
    __builtin_amdgcn_wave_barrier();
    unsigned long long found = 0;
    auto consume = [&](bool hit, float d, int id) {
        const unsigned long long m = __ballot(hit);
        if (FILL && m) {
            int o0 = 0;
            if (lane == 0) o0 = atomicAdd(&cursor[j], __popcll(m));
            o0 = __shfl(o0, 0, 64);
            if (hit) {
                const i64 o = hoff[j] + o0 + __popcll(m & ((1ull << lane) - 1));
                if (idx_bits > 0) {
                    keys_out[o] = ((u64)j << (31 + idx_bits)) | ((u64)__float_as_uint(d) << idx_bits) | (u32)id;
                } else {
                    keys_out[o] = ((u64)__float_as_uint(d) << 32) | (u32)id;
                    row_out[o] = j;
                }
            }
        }
        found += __popcll(m);
    };
    for (int i0 = lo + wave * 64; i0 < hi; i0 += 256) {
        const int i = i0 + lane;
        bool hit = false;
        float d = 0.f;
        int id = 0;
        if (i < hi) {
            const float4 pt = radius_candidate<NCELL>(sorted, s_pref[wave], s_beg[wave], i);
            d = sqdist3(pt.x, pt.y, pt.z, cx, cy, cz);
            hit = d < r2;
            id = __float_as_int(pt.w);
        }
        consume(hit, d, id);
    }
    if (ALIGNED && blockIdx.y == 0 && wave == 0) {  // pairs of the rounding margin, once per row
        const int nex = min(*aq.extras_cnt, EXTRA_CAP);
        for (int e0 = 0; e0 < nex; e0 += 64) {
            const int e = e0 + lane;
            bool hit = false;
            float d = 0.f;
            int id = 0;
            if (e < nex) {
                const int2 it = aq.extras[e];
                if (it.x == (int)q) {
                    const float4 pt = sorted[it.y];
                    d = sqdist3(pt.x, pt.y, pt.z, cx, cy, cz);
                    hit = d < r2;
                    id = __float_as_int(pt.w);
                }
            }
            if (__ballot(hit)) consume(hit, d, id);
        }
    }
    if (!FILL && lane == 0 && found) atomicAdd((unsigned long long*)&counts[q], found);
}
// copies the light rows from their fixed slots to the CSR positions: 16 lanes per row.  The slots hold
// (squared distance, position in Morton order); index and radius come from the Morton-ordered arrays
// (neighbours of one voxel are close in that order: clustered reads instead of one line per pair)
__global__ void k_radius_place(const u64* tmp, const i64* rs, const uint8_t* is_heavy, i64 v, const float* sizes,
                               const int32_t* ids, const float* srad, int32_t* idx, int32_t* spos, float* dist,
                               float* compat) {
    const i64 q = (blockIdx.x * (i64)blockDim.x + threadIdx.x) >> 4;
    const int l = threadIdx.x & 15;
    if (q >= v) return;
    const i64 b = rs[q];
    const int cnt = (int)(rs[q + 1] - b);
    if (is_heavy[q]) return;  // written by k_radius_unpack_heavy
    const float a = sizes[q];
    for (int i = l; i < cnt; i += 16) {
        const u64 k = tmp[q * RADIUS_LIGHT + i];
        const int32_t pos = (int32_t)(k & 0xffffffffu);
        idx[b + i] = ids[pos];
        if (spos) spos[b + i] = pos;
        dist[b + i] = __uint_as_float((unsigned)(k >> 32));
        if (compat) {
            float bb = 2 * srad[pos];
            float ratio = fminf(a, bb) / fmaxf(a, bb);
            compat[b + i] = ratio * ratio;
        }
    }
}
__global__ void k_heavy_counts(const int32_t* heavy, i64 nh, const i64* rs, i64* cnt) {
    i64 j = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (j > nh) return;
    cnt[j] = j < nh ? rs[heavy[j] + 1] - rs[heavy[j]] : 0;
}
__global__ void k_radius_unpack_heavy(const u64* keys, const int32_t* hrow, i64 num_pairs, const int32_t* heavy,
                                      const i64* hoff, const i64* rs, const float* sizes, const float* radii,
                                      const int32_t* rank, int32_t* idx, int32_t* spos, float* dist, float* compat,
                                      int idx_bits) {
    i64 p = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (p >= num_pairs) return;
    u64 k = keys[p];
    int j;
    if (idx_bits > 0) {  // (row, distance bits, index) in one key: see k_radius_heavy
        j = (int)(k >> (31 + idx_bits));
        k = (((k >> idx_bits) & 0x7fffffffull) << 32) | (k & ((u64(1) << idx_bits) - 1));
    } else {
        j = hrow[p];
    }
    const i64 q = heavy[j];
    const i64 o = rs[q] + (p - hoff[j]);
    const int32_t id = (int32_t)(k & 0xffffffffu);
    idx[o] = id;
    if (spos) spos[o] = rank[id];
    dist[o] = __uint_as_float((unsigned)(k >> 32));
    if (compat) {
        float a = sizes[q];
        float bb = 2 * radii[id];
        float ratio = fminf(a, bb) / fmaxf(a, bb);
        compat[o] = ratio * ratio;
    }
}
// Exact k-th nearest neighbour distance, one wave per point (in Morton order).  Level by level,
// finest first: lanes 0..26 look up the 3^3 cells around the point's cell; if they hold >= k points
// the k-th smallest squared distance d_k among them is selected; it is exact as soon as
// d_k <= (cell size)^2, because every point closer than one cell size lies inside the 3^3 block.
// Selection: binary search on the bit pattern of the (non-negative) squared distances with wave
// ballots; up to 512 candidates are held in registers, larger sets are streamed from memory.
constexpr int KNN_CROWD = 1024;  // points of a finest-level cell above which a point starts its search on a finer level
__global__ __launch_bounds__(256) void k_knn(asr_octree_frame f, const float4* sorted, i64 n, CellIndex ci, int lfine,
                                             int ldeep, int k, const float* radii_in, float radius_fraction,
                                             int outlier_threshold, float* radii_out,
                                             uint8_t* inlier_out, const int32_t* list, i64 nlist) {
    constexpr int CMAX = 8;
    __shared__ int s_pref[4][28];
    __shared__ int s_beg[4][28];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    i64 s = blockIdx.x * (i64)4 + wave;
    if (list) {  // only the points (positions in Morton order) that k_knn_cells could not settle
        if (s >= nlist) return;
        s = list[s];
    }
    if (s >= n) return;
    const float4 me = sorted[s];
    const int kk = (int)(n < k ? n : k);
    u32 kth_bits = 0;
    u32 bound_bits = 0x7f800000u;  // upper bound of the answer (+inf until a level has been evaluated)
    int total = 0;
    // Start level: the finest table level, or -- for a point in a crowded cell of a cloud with dense spots (ldeep >
    // lfine: the codes are sorted that deep) -- the level at which its own cell holds at most KNN_CROWD points; cells
    // below the table's finest level are found by binary search in the sorted codes (cell_range).
    int lev0 = lfine;
    if (ldeep > lfine) {
        const u64 code = ci.codes[s];
        int b0, pop;
        cell_range(ci, code >> (3 * (ASR_MAX_LEVEL - lfine)), lfine, b0, pop);
        while (pop > KNN_CROWD && lev0 < ldeep) {
            ++lev0;
            cell_range(ci, code >> (3 * (ASR_MAX_LEVEL - lev0)), lev0, b0, pop);
        }
    }
    for (int lev = lev0; lev >= 0; --lev) {
        int x, y, z;
        if (lev > lfine) {  // the point's cell as the sorted codes have it
            const u64 cell = ci.codes[s] >> (3 * (ASR_MAX_LEVEL - lev));
            x = (int)asr_compact21(cell);
            y = (int)asr_compact21(cell >> 1);
            z = (int)asr_compact21(cell >> 2);
        } else {
            frame_coord(f, me.x, me.y, me.z, lev, x, y, z);
        }
        const int lim = (1 << lev) - 1;
        x = min(max(x, 0), lim);  // points outside the root cube were clamped into it
        y = min(max(y, 0), lim);
        z = min(max(z, 0), lim);
        int b = 0, cnt = 0;
        if (lane < 27) {
            int xx = x + lane % 3 - 1, yy = y + (lane / 3) % 3 - 1, zz = z + lane / 9 - 1;
            if (xx >= 0 && yy >= 0 && zz >= 0 && xx <= lim && yy <= lim && zz <= lim)
                cell_range(ci, asr_morton3d((u64)xx, (u64)yy, (u64)zz), lev, b, cnt);
        }
        int pre = cnt;
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
            int up = __shfl_up(pre, o, 64);
            if (lane >= o) pre += up;
        }
        total = __shfl(pre, 26, 64);
        if (total < kk && lev > 0) continue;
        __builtin_amdgcn_wave_barrier();
        if (lane < 27) {
            s_pref[wave][lane + 1] = pre;
            s_beg[wave][lane] = b;
        }
        if (lane == 0) s_pref[wave][0] = 0;
        __builtin_amdgcn_wave_barrier();
        auto cand = [&](int i) -> float {
            int lo = 0, hi = 27;
            while (hi - lo > 1) {
                int mid = (lo + hi) >> 1;
                if (s_pref[wave][mid] <= i)
                    lo = mid;
                else
                    hi = mid;
            }
            const float4 pt = sorted[s_beg[wave][lo] + (i - s_pref[wave][lo])];
            return sqdist3(pt.x, pt.y, pt.z, me.x, me.y, me.z);
        };
        const int need = total < kk ? total : kk;  // lev == 0 with fewer than k points overall
        // Selection of the need-th smallest squared distance: one pass over the candidates that keeps a
        // pool of 8 values per lane.  Only values <= bound enter, where bound is an upper bound of the
        // answer: the result of the finer level (k-th among a subset of these candidates) and, once a
        // lane's slots are full, the need-th smallest of the pool itself.  (The previous version ran a
        // 31-pass bit search over ALL candidates of every level it tried: an isolated point whose
        // 3^3 block is a coarse cell next to a dense region walked 10^6 candidates 31 times.)
        u32 pool[CMAX];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) pool[c] = 0x7f800000u;
        int nl = 0;  // used slots of this lane
        bool overflow = false;
        // need-th smallest of the pool = largest bit pattern t with #{pool < t} < need
        auto select = [&]() {
            u32 lo_bits = 0;
            for (int bit = 30; bit >= 0; --bit) {
                const u32 trial = lo_bits | (1u << bit);
                int c_lt = 0;
#pragma unroll
                for (int c = 0; c < CMAX; ++c) c_lt += __popcll(__ballot(pool[c] < trial));
                if (c_lt < need) lo_bits = trial;
            }
            return lo_bits;
        };
        for (int i0 = 0; i0 < total && !overflow; i0 += 64) {
            const int i = i0 + lane;
            const u32 db = i < total ? __float_as_uint(cand(i)) : 0x7f800001u;
            if (__ballot(db <= bound_bits && nl == CMAX)) {
                // a lane is full: tighten the bound to the pool's need-th smallest and drop the rest
                int have = 0;
#pragma unroll
                for (int c = 0; c < CMAX; ++c) have += __popcll(__ballot(pool[c] != 0x7f800000u));
                if (have >= need) {
                    const u32 nb = select();
                    if (nb < bound_bits) bound_bits = nb;
                }
                u32 keep[CMAX];
                int m = 0;
#pragma unroll
                for (int c = 0; c < CMAX; ++c) keep[c] = 0x7f800000u;
#pragma unroll
                for (int c = 0; c < CMAX; ++c) {
                    const bool k2 = pool[c] <= bound_bits && pool[c] != 0x7f800000u;
#pragma unroll
                    for (int e = 0; e < CMAX; ++e)
                        if (k2 && e == m) keep[e] = pool[c];
                    m += k2;
                }
#pragma unroll
                for (int c = 0; c < CMAX; ++c) pool[c] = keep[c];
                nl = m;
                if (__ballot(db <= bound_bits && nl == CMAX)) overflow = true;  // > 8 ties on one lane
            }
            if (!overflow && db <= bound_bits) {
#pragma unroll
                for (int c = 0; c < CMAX; ++c)
                    if (c == nl) pool[c] = db;
                ++nl;
            }
        }
        u32 lo_bits = 0;
        if (!overflow) {
            lo_bits = select();
        } else {  // pathological (hundreds of equal distances): bit search over the stream
            for (int bit = 30; bit >= 0; --bit) {
                const u32 trial = lo_bits | (1u << bit);
                int c_lt = 0;
                for (int i0 = 0; i0 < total; i0 += 64) {
                    const int i = i0 + lane;
                    const bool lt = i < total && __float_as_uint(cand(i)) < trial;
                    c_lt += __popcll(__ballot(lt));
                }
                if (c_lt < need) lo_bits = trial;
            }
        }
        bound_bits = lo_bits;  // k-th among a subset of the next (coarser) level's candidates
        kth_bits = lo_bits;
        const float cs = f.voxel_size[lev];
        if (lev == 0 || __uint_as_float(kth_bits) <= cs * cs) {
            // exact.  Optional inlier vote (KDTree::ComputeInlier): neighbours among the k nearest
            // whose radius is below radius_fraction * radius_i
            if (inlier_out) {
                // The vote is over exactly `need` neighbours, like the k results of a kNN query
                // (nsearch.cpp:62-80): all candidates strictly closer than the k-th distance, plus as
                // many of the candidates AT that distance as are needed to reach k, taken in ascending
                // point index (which tie a k-d tree returns is traversal dependent; this is the
                // deterministic rule, and the one the brute-force check in the tests uses).
                const float thr = radii_in[__float_as_int(me.w)] * radius_fraction;
                auto cand_pt = [&](int i) -> float4 {
                    int lo = 0, hi = 27;
                    while (hi - lo > 1) {
                        int mid = (lo + hi) >> 1;
                        if (s_pref[wave][mid] <= i)
                            lo = mid;
                        else
                            hi = mid;
                    }
                    return sorted[s_beg[wave][lo] + (i - s_pref[wave][lo])];
                };
                int c_lt = 0, n_eq = 0, votes_lt = 0, votes_eq = 0;
                for (int i0 = 0; i0 < total; i0 += 64) {
                    const int i = i0 + lane;
                    bool lt = false, eq = false, voter = false;
                    if (i < total) {
                        const float4 pt = cand_pt(i);
                        const u32 db = __float_as_uint(sqdist3(pt.x, pt.y, pt.z, me.x, me.y, me.z));
                        lt = db < kth_bits;
                        eq = db == kth_bits;
                        voter = (lt || eq) && radii_in[__float_as_int(pt.w)] < thr;
                    }
                    c_lt += __popcll(__ballot(lt));
                    n_eq += __popcll(__ballot(eq));
                    votes_lt += __popcll(__ballot(lt && voter));
                    votes_eq += __popcll(__ballot(eq && voter));
                }
                const int admit = need - c_lt;  // >= 1: the k-th neighbour itself sits at kth_bits
                if (n_eq > admit) {
                    // more ties than places: the `admit` smallest indices among them (bit search)
                    u32 idx_hi = 0;  // largest t with #{ties with index < t} < admit == the admit-th smallest index
                    for (int bit = 30; bit >= 0; --bit) {
                        const u32 trial = idx_hi | (1u << bit);
                        int below = 0;
                        for (int i0 = 0; i0 < total; i0 += 64) {
                            const int i = i0 + lane;
                            bool b2 = false;
                            if (i < total) {
                                const float4 pt = cand_pt(i);
                                b2 = __float_as_uint(sqdist3(pt.x, pt.y, pt.z, me.x, me.y, me.z)) == kth_bits &&
                                     (u32)__float_as_int(pt.w) < trial;
                            }
                            below += __popcll(__ballot(b2));
                        }
                        if (below < admit) idx_hi = trial;
                    }
                    votes_eq = 0;
                    for (int i0 = 0; i0 < total; i0 += 64) {
                        const int i = i0 + lane;
                        bool v = false;
                        if (i < total) {
                            const float4 pt = cand_pt(i);
                            v = __float_as_uint(sqdist3(pt.x, pt.y, pt.z, me.x, me.y, me.z)) == kth_bits &&
                                (u32)__float_as_int(pt.w) <= idx_hi && radii_in[__float_as_int(pt.w)] < thr;
                        }
                        votes_eq += __popcll(__ballot(v));
                    }
                }
                const int votes = votes_lt + votes_eq;
                if (lane == 0) inlier_out[__float_as_int(me.w)] = votes < outlier_threshold ? 1 : 0;
            }
            break;
        }
    }
    if (lane == 0 && radii_out) radii_out[__float_as_int(me.w)] = sqrtf(__uint_as_float(kth_bits));
}


// ------------------------------------------------------------------------------------------
// kNN radius, fast path: ONE WAVE PER OCCUPIED CELL of the finest level.  All points of a cell share the 3^3
// block of candidate cells, so the wave stages the candidates in LDS 64 at a time and every lane keeps the K
// smallest squared distances of ITS query point in registers (ascending; inserting d and dropping the largest is
// a[c] = med3(a[c-1], d, a[c]) from the top down: one v_med3_f32 per slot, no divergence; a chunk no lane can use
// is skipped with one ballot).  The wave-per-point kernel above spends 27 table probes, a candidate walk and two
// 31-step bit searches on every point.  A result is exact when the k-th distance does not exceed the cell size
// (the k nearest lie inside the block); the other points -- sparse regions, cells with fewer than k candidates
// -- go to a list that k_knn settles level by level as before.  Same numbers as k_knn (tested).  Measured and not
// kept: cells one level coarser (43 vs 26 ms), one wave per parent cell with the 4^3 block of finest cells as
// candidates (fuller waves, but 2.3x the candidates per query: 20.5 vs 16.7 ms).
// ------------------------------------------------------------------------------------------
// largest population of a level-`level` cell (table slots hold [start, end) per (cell, level))
__global__ void k_max_cell_pop(HashTab t, const int32_t* start, const int32_t* end, int level, int* out) {
    const u64 slot = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (slot > t.mask) return;
    const u64 key = t.keys[slot];
    if (key == 0 || asr_key_level(key) != level) return;
    const int pop = end[slot] - start[slot];
    if (pop > KNN_CROWD) atomicMax(out, pop);
}

// first point of every run of equal level-`level` cells in the sorted codes.  4096 points per block, ONE atomic per
// block for the block's share of the list (same-address atomics retire at ~88 / us).
__global__ __launch_bounds__(256) void k_cell_list(const u64* codes, i64 n, int level, int32_t* list, int* cnt) {
    __shared__ int s_cnt[256];
    __shared__ int s_base;
    const int s = 3 * (ASR_MAX_LEVEL - level);
    const i64 base = blockIdx.x * (i64)4096;
    int mine = 0;
    unsigned flags = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const i64 i = base + j * 256 + threadIdx.x;
        if (i < n && (i == 0 || (codes[i - 1] >> s) != (codes[i] >> s))) {
            flags |= 1u << j;
            ++mine;
        }
    }
    s_cnt[threadIdx.x] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int t = 0; t < 256; ++t) {
            const int c = s_cnt[t];
            s_cnt[t] = tot;
            tot += c;
        }
        s_base = tot ? atomicAdd(cnt, tot) : 0;
    }
    __syncthreads();
    int o = s_base + s_cnt[threadIdx.x];
#pragma unroll
    for (int j = 0; j < 16; ++j)
        if (flags & (1u << j)) list[o++] = (int32_t)(base + j * 256 + threadIdx.x);
}

constexpr int KNN_CELL_MAX_WORK = 12288;  // passes of 64 queries x candidates of the 3^3 block a wave may take on (~0.5 ms)
template <int K>
__global__ __launch_bounds__(256) void k_knn_cells(asr_octree_frame f, const float4* sorted, i64 n, HashTab t,
                                                   const int32_t* start, const int32_t* end, int lfine, int k,
                                                   const int32_t* cells, i64 ncells, float* radii_out,
                                                   int32_t* fallback, int* fallback_cnt) {
    __shared__ int s_pref[4][28];
    __shared__ int s_beg[4][28];
    __shared__ __attribute__((aligned(16))) float4 s_cand[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const i64 ci = blockIdx.x * (i64)4 + wave;
    if (ci >= ncells) return;
    const int first = cells[ci];
    const float4 p0 = sorted[first];
    int x, y, z;
    frame_coord(f, p0.x, p0.y, p0.z, lfine, x, y, z);
    const int lim = (1 << lfine) - 1;
    x = min(max(x, 0), lim);
    y = min(max(y, 0), lim);
    z = min(max(z, 0), lim);
    int b = 0, cnt = 0;
    if (lane < 27) {
        // candidate cells nearest first (the cell itself, 6 face, 12 edge, 8 corner neighbours): the k-th distance is
        // tight after the first few, and chunks of the far cells are skipped by the ballot test below
        constexpr unsigned char order[27] = {13, 4, 10, 12, 14, 16, 22, 1, 3, 5, 7, 9, 11, 15, 17, 19, 21, 23, 25,
                                             0, 2, 6, 8, 18, 20, 24, 26};
        const int c = order[lane];
        int xx = x + c % 3 - 1, yy = y + (c / 3) % 3 - 1, zz = z + c / 9 - 1;
        if (xx >= 0 && yy >= 0 && zz >= 0 && xx <= lim && yy <= lim && zz <= lim) {
            i64 slot = tab_find_slot(t, asr_morton3d((u64)xx, (u64)yy, (u64)zz) | (u64(1) << (3 * lfine)));
            if (slot >= 0) {
                b = start[slot];
                cnt = end[slot] - b;
            }
        }
    }
    int pre = cnt;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
        int up = __shfl_up(pre, o, 64);
        if (lane >= o) pre += up;
    }
    const int total = __shfl(pre, 26, 64);
    const int qb = __shfl(b, 0, 64), qn = __shfl(cnt, 0, 64);  // the cell itself: its points are the queries
    if (lane < 27) {
        s_pref[wave][lane + 1] = pre;
        s_beg[wave][lane] = b;
    }
    if (lane == 0) s_pref[wave][0] = 0;
    __builtin_amdgcn_wave_barrier();
    const float cs = f.voxel_size[lfine];
    const float cs2 = cs * cs;
    // Crowded cells go to the wave-per-point kernel: this wave would walk points x candidates alone (a dense spot of
    // 10^5 points in one cell: 10^10 distance evaluations on one SIMD; such a cloud took 312 s before this line).
    if (total < k || (i64)((qn + 63) / 64) * total > KNN_CELL_MAX_WORK) {
        for (int q0 = lane; q0 < qn; q0 += 64) fallback[atomicAdd(fallback_cnt, 1)] = qb + q0;
        return;
    }
    for (int q0 = 0; q0 < qn; q0 += 64) {
        const bool active = q0 + lane < qn;
        const float4 me = sorted[qb + (active ? q0 + lane : 0)];
        float a[K];  // squared distances are >= 0 and finite: float order, +inf = empty
#pragma unroll
        for (int c = 0; c < K; ++c) a[c] = __uint_as_float(0x7f800000u);
        if (total >= k) {
            for (int c0 = 0; c0 < total; c0 += 64) {
                const int m = min(64, total - c0);
                __builtin_amdgcn_wave_barrier();  // the previous chunk has been read
                if (lane < m) s_cand[wave][lane] = radius_candidate(sorted, s_pref[wave], s_beg[wave], c0 + lane);
                __builtin_amdgcn_wave_barrier();
                for (int j = 0; j < m; ++j) {
                    const float4 pt = s_cand[wave][j];
                    const float d = sqdist3(pt.x, pt.y, pt.z, me.x, me.y, me.z);
                    if (__ballot(d < a[K - 1]) == 0) continue;
#pragma unroll
                    for (int c = K - 1; c > 0; --c) a[c] = __builtin_amdgcn_fmed3f(a[c - 1], d, a[c]);
                    a[0] = fminf(a[0], d);
                }
            }
        }
        float kth = a[K - 1];
#pragma unroll
        for (int c = 0; c < K; ++c)
            if (c == k - 1) kth = a[c];
        if (active) {
            if (total >= k && kth <= cs2)
                radii_out[__float_as_int(me.w)] = sqrtf(kth);
            else
                fallback[atomicAdd(fallback_cnt, 1)] = qb + q0 + lane;
        }
    }
}

// rows are sorted by a segmented radix sort on (distance bits, index) keys; unpack + compat
// ------------------------------------------------------------------------------------------
// a11: CSR inversion
// ------------------------------------------------------------------------------------------
__global__ void k_invert_hist(const int32_t* idx, i64 p, i64* counts, i64 num_points) {
    i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (i < p) atomicAdd((unsigned long long*)&counts[idx[i]], 1ull);
    (void)num_points;
}
__global__ void k_iota32(int32_t* out, i64 n) {
    i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)i;
}
__global__ void k_invert_fill(const int32_t* sorted_src, i64 p, const i64* rs, i64 num_rows,
                              const uint8_t* attr, int32_t* out_idx, uint8_t* out_attr) {
    i64 d = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (d >= p) return;
    i64 src = sorted_src[d];
    // row of pair src: largest q with rs[q] <= src
    i64 lo = 0, hi = num_rows;
    while (hi - lo > 1) {
        i64 mid = (lo + hi) >> 1;
        if (rs[mid] <= src)
            lo = mid;
        else
            hi = mid;
    }
    out_idx[d] = (int32_t)lo;
    if (attr) out_attr[d] = attr[src];
}

// ------------------------------------------------------------------------------------------
// row regrouping for the MFMA sparse conv: rows of one CSR are reordered inside segments of
// `seg` consecutive rows so that rows with the same set of kernel slots sit in the same 16-row
// MFMA tile (ascending 55-bit slot mask).  Purely a tiling order: results do not depend on it.
// ------------------------------------------------------------------------------------------
__global__ void k_row_masks(const uint8_t* kidx, const i64* rs, i64 v, u64* masks, int32_t* ids) {
    i64 q = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (q >= v) return;
    u64 m = 0;
    for (i64 p = rs[q]; p < rs[q + 1]; ++p) m |= u64(1) << (kidx[p] & 63);
    masks[q] = m;
    ids[q] = (int32_t)q;
}
// Longest-processing-time order of the 128-row chunks inside each segment: a block's cost is the
// number of slots in the union of its rows' masks (one weight panel walk + barrier per slot), so
// heavy tiles start first and the tail of a launch is made of light ones.
__global__ void k_tile_cost(const int32_t* perm, const u64* masks, i64 v, int32_t* cost) {
    const i64 tile = (blockIdx.x * (i64)blockDim.x + threadIdx.x) >> 6;  // one wave per 64-row tile
    const int lane = threadIdx.x & 63;
    if (tile * 64 >= v) return;
    const i64 r = tile * 64 + lane;
    u64 m = r < v ? masks[perm[r]] : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m |= __shfl_xor(m, o, 64);
    if (lane == 0) cost[tile] = __popcll(m);
}
__global__ void k_chunk_key(const int32_t* tile_cost, i64 v, i64 seg, int32_t* key, int32_t* ids) {
    i64 c = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    i64 r0 = c * 128;
    if (r0 >= v) return;
    int cost = tile_cost[2 * c] + (r0 + 64 < v ? tile_cost[2 * c + 1] : 0);
    if (r0 + 128 > v) cost = 0;  // the partial chunk stays last
    key[c] = (int32_t)((r0 / seg) * 128 + (127 - cost));
    ids[c] = (int32_t)c;
}
__global__ void k_chunk_apply(const int32_t* perm, const int32_t* order, i64 v, int32_t* out) {
    i64 r = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (r >= v) return;
    i64 src = (i64)order[r >> 7] * 128 + (r & 127);
    out[r] = perm[src];
}
__global__ void k_segment_of(const int32_t* rows, i64 v, i64 seg, int32_t* out) {
    i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (i < v) out[i] = (int32_t)(rows[i] / seg);
}

// ------------------------------------------------------------------------------------------
// batched row regrouping: the 13 CSRs of a grid hierarchy (5 neighbour lists, 4 up lists, 4 inverted up lists)
// in ONE radix sort + one chunk sort instead of 13 x (2..3 sorts) -- on the small levels each rocPRIM sort is
// a dozen launches of a few microseconds.  Rows of all jobs live in one index space, every job padded to a
// multiple of 128 rows; key = job (4 bits) | segment (6 bits) | slot mask without bit 0 (54 bits): the stable
// sort leaves each job's rows contiguous, ordered by (segment, mask, row) exactly like asr_geom_row_groups.
// ------------------------------------------------------------------------------------------
constexpr int RG_MAX_JOBS = 16;
struct RowGroupBatch {
    int n;
    i64 base[RG_MAX_JOBS + 1];  // first padded row of each job (multiples of 128); base[n] = total
    i64 v[RG_MAX_JOBS];
    const uint8_t* kidx[RG_MAX_JOBS];
    const i64* rs[RG_MAX_JOBS];
    int32_t* out[RG_MAX_JOBS];
};
__device__ inline int rg_job_of(const RowGroupBatch& b, i64 e) {
    int j = 0;
    while (j + 1 < b.n && e >= b.base[j + 1]) ++j;
    return j;
}
// key = (job, segment, slot mask without slot 0) in 4 + 6 + mask_bits bits
__global__ void k_rg_keys(RowGroupBatch b, i64 seg, u64* keys, u64* masks, int32_t* ids, int mask_bits) {
    const i64 e = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (e >= b.base[b.n]) return;
    const int j = rg_job_of(b, e);
    const i64 r = e - b.base[j];
    u64 m = 0, key;
    if (r < b.v[j]) {
        const i64* rs = b.rs[j];
        const uint8_t* kidx = b.kidx[j];
        for (i64 p = rs[r]; p < rs[r + 1]; ++p) m |= u64(1) << (kidx[p] & 63);
        key = ((u64)j << (6 + mask_bits)) | ((u64)(r / seg) << mask_bits) | ((m >> 1) & ((u64(1) << mask_bits) - 1));
    } else {
        key = ((u64)j << (6 + mask_bits)) | (u64(63) << mask_bits) | ((u64(1) << mask_bits) - 1);  // padding: last inside the job
    }
    keys[e] = key;
    masks[e] = m;
    ids[e] = (int32_t)e;
}
// 128-row chunks of the sorted order: cost = slots in the union of the rows' masks (as k_tile_cost / k_chunk_key)
__global__ void k_rg_chunk_keys(RowGroupBatch b, i64 seg, const int32_t* sorted_ids, const u64* masks, int32_t* ckey,
                                int32_t* cid) {
    const i64 c = (blockIdx.x * (i64)blockDim.x + threadIdx.x) >> 6;  // one wave per chunk
    const int lane = threadIdx.x & 63;
    if (c * 128 >= b.base[b.n]) return;
    const int j = rg_job_of(b, c * 128);
    const i64 r0 = c * 128 - b.base[j];
    u64 m0 = masks[sorted_ids[c * 128 + lane]], m1 = masks[sorted_ids[c * 128 + 64 + lane]];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        m0 |= __shfl_xor(m0, o, 64);
        m1 |= __shfl_xor(m1, o, 64);
    }
    if (lane == 0) {
        int cost = __popcll(m0) + __popcll(m1);  // the two 64-row tiles of the chunk, like k_chunk_key
        if (r0 + 128 > b.v[j]) cost = 0;         // the partial chunk stays last
        ckey[c] = (int32_t)(((i64)j << 13) | ((r0 / seg) << 7) | (127 - cost));
        cid[c] = (int32_t)c;
    }
}
__global__ void k_rg_apply(RowGroupBatch b, const int32_t* sorted_ids, const int32_t* order, int lpt) {
    const i64 e = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (e >= b.base[b.n]) return;
    const int j = rg_job_of(b, e);
    const i64 r = e - b.base[j];
    if (r >= b.v[j]) return;
    const i64 src = lpt ? (i64)order[e >> 7] * 128 + (e & 127) : e;
    b.out[j][r] = (int32_t)(sorted_ids[src] - b.base[j]);
}

// Ranked keys (round 4): a hierarchy has a few thousand DISTINCT slot masks, so the 54 mask bits of the sort key are replaced
// by the mask's rank among the distinct masks (same order, same permutation): (job, segment, rank) is 22 bits -- three digit
// passes over all 13 lists in one sort instead of three passes over the 9-slot lists plus eight over the 55-slot ones.
// Distinct masks: a small hash set (k_rg_masks: all but the first insertion of a mask are read-only probes), collected into a
// dense list (k_rg_mask_collect) and ranked by counting (k_rg_mask_ranks), rank written back as the set's value.  More than
// RG_MAX_MASKS distinct masks: the caller falls back to the full-mask keys.
constexpr int RG_MAX_MASKS = 16384;  // (C3: ~1 500 distinct masks, mixed-density C5: more than 4 096)
constexpr int RG_MASK_TAB = 65536;
__global__ void k_rg_masks(RowGroupBatch b, u64* masks, HashTab t, int* cnt) {
    const i64 e = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (e >= b.base[b.n]) return;
    const int j = rg_job_of(b, e);
    const i64 r = e - b.base[j];
    u64 m = 0;
    if (r < b.v[j]) {
        const i64* rs = b.rs[j];
        const uint8_t* kidx = b.kidx[j];
        for (i64 p = rs[r]; p < rs[r + 1]; ++p) m |= u64(1) << (kidx[p] & 63);
        if (tab_insert_shared(t, (m >> 1) + 1) < 0) cnt[1] = 1;
    }
    masks[e] = m;
}
// the set's keys as a dense list (any order) with their slots; cnt[2] = how many
__global__ void k_rg_mask_collect(HashTab t, u64* list, int32_t* slot_of, int* cnt) {
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    const u64 k = i <= t.mask ? t.keys[i] : 0;
    const int o = block_append(k != 0, cnt + 2);
    if (k && o < RG_MAX_MASKS) {
        list[o] = k;
        slot_of[o] = (int32_t)i;
    }
}
// rank of every distinct mask = number of smaller ones (the masks are distinct): one thread per mask, the list read through
// LDS tiles.  (Round 4: this was a bitonic sort by ONE workgroup, 0.5 ms alone and 1.3 ms beside the search's query kernel,
// whose waves share that CU's issue slots -- on the critical chain of the build.)
__global__ __launch_bounds__(256) void k_rg_mask_ranks(HashTab t, const u64* list, const int32_t* slot_of, const int* cnt) {
    __shared__ u64 s_key[2048];
    const int n = cnt[2];
    if (n > RG_MAX_MASKS) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x * blockDim.x >= n) return;
    const u64 mine = i < n ? list[i] : 0;
    int rank = 0;
    for (int base = 0; base < n; base += 2048) {
        __syncthreads();
        for (int e = threadIdx.x; e < 2048; e += blockDim.x) s_key[e] = base + e < n ? list[base + e] : ~u64(0);
        __syncthreads();
        const int m = min(2048, n - base);
        for (int e = 0; e < m; ++e) rank += s_key[e] < mine;  // (LDS broadcast reads)
    }
    if (i < n) t.vals[slot_of[i]] = rank;
}
// key = (job, segment, rank of the slot mask) in 4 + 6 + rank_bits bits
__global__ void k_rg_keys_ranked(RowGroupBatch b, i64 seg, const u64* masks, HashTab t, unsigned* keys, int32_t* ids,
                                 int rank_bits) {
    const i64 e = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (e >= b.base[b.n]) return;
    const int j = rg_job_of(b, e);
    const i64 r = e - b.base[j];
    unsigned key;
    if (r < b.v[j]) {
        const int rank = tab_find(t, (masks[e] >> 1) + 1);
        key = ((unsigned)j << (6 + rank_bits)) | ((unsigned)(r / seg) << rank_bits) | (unsigned)rank;
    } else {
        key = ((unsigned)j << (6 + rank_bits)) | (63u << rank_bits) | ((1u << rank_bits) - 1u);  // padding: last inside the job
    }
    keys[e] = key;
    ids[e] = (int32_t)e;
}

// ------------------------------------------------------------------------------------------
// dual cells ("next" row D.1): CreateDualVertexIndices, cpp/lib/grid.cpp:316-459 with the vertex /
// adjacent-node key algebra of cpp/lib/octreebase.h:86-118.  One thread per leaf, 8 corners each.
// ------------------------------------------------------------------------------------------
__device__ inline u64 dev_morton_add(u64 a, u64 b) {
    const u64 M = 0x9249249249249249ull;
    u64 c = ((a | ~M) + (b & M)) & M;
    c |= ((a | ~(M << 1)) + (b & (M << 1))) & (M << 1);
    c |= ((a | ~(M << 2)) + (b & (M << 2))) & (M << 2);
    return c;
}
__device__ inline u64 dev_morton_sub(u64 a, u64 b) {
    const u64 M = 0x9249249249249249ull;
    u64 c = ((a & M) - (b & M)) & M;
    c |= ((a & (M << 1)) - (b & (M << 1))) & (M << 1);
    c |= ((a & (M << 2)) - (b & (M << 2))) & (M << 2);
    return c;
}
// map: node key -> leaf index, or -1 for inner nodes; returns -2 when the key is no node
__device__ inline int node_lookup(const HashTab& t, u64 key) {
    TabProbe pr = tab_probe(t, key);
    for (u64 probe = 0; probe <= pr.bmask; ++probe, pr.next()) {
        const u64 slot = pr.slot();
        const u64 cur = t.keys[slot];
        if (cur == key) return t.vals[slot];
        if (cur == 0) return -2;
    }
    return -2;
}
// does leaf `key` own the dual cell of its corner i? (grid.cpp:334-360)
__device__ inline bool dual_owned(const HashTab& t, u64 key, int i, u64& vertex_key) {
    const u64 IM = 0x9249249249249249ull;
    const int lev = asr_key_level(key);
    const u64 min_lev_key = u64(1) << (3 * lev);
    const u64 vk = dev_morton_add(key, (u64)i);
    const u64 vk_ = vk - min_lev_key;
    if (vk >= (min_lev_key << 1) || !(vk_ & IM) || !(vk_ & (IM << 1)) || !(vk_ & (IM << 2))) return false;
    vertex_key = vk;
    for (int j = 0; j < 8; ++j) {
        if (j == i) continue;
        const u64 ak = dev_morton_sub(vk, (u64)j);
        const int r = node_lookup(t, ak);
        if (r == -2) continue;        // no node there: this leaf is the deeper one
        if (r == -1) return false;    // inner node: one of its children owns the vertex
        if (ak < key) return false;   // same-level leaf with the smaller key owns it
    }
    return true;
}
__global__ void k_dual_count(const u64* leaves, i64 v, HashTab t, i64* counts) {
    i64 q = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (q > v) return;
    if (q == v) {
        counts[v] = 0;
        return;
    }
    const u64 key = leaves[q];
    int n = 0;
    u64 vk;
    for (int i = 0; i < 8; ++i) n += dual_owned(t, key, i, vk) ? 1 : 0;
    counts[q] = n;
}
__global__ void k_dual_fill(const u64* leaves, i64 v, HashTab t, const i64* offsets, i64* out, int* cnt) {
    i64 q = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (q >= v) return;
    const u64 key = leaves[q];
    i64 o = offsets[q];
    for (int i = 0; i < 8; ++i) {
        u64 vk;
        if (!dual_owned(t, key, i, vk)) continue;
        for (int j = 0; j < 8; ++j) {
            u64 k = dev_morton_sub(vk, (u64)j);
            int r = -2;
            while (k != 0 && (r = node_lookup(t, k)) == -2) k >>= 3;  // grid.cpp:429-441
            if (k == 0 || r < 0) {
                cnt[1] = 1;  // "invalid key after searching for node" / "found node is not a leaf"
                r = -1;
            }
            out[o * 8 + j] = r;
        }
        ++o;
    }
}

// ------------------------------------------------------------------------------------------
// host helpers
// ------------------------------------------------------------------------------------------
int make_table(asr_hip_context* ctx, Arena& arena, u64 cap, bool with_vals, HashTab& t) {
    t.keys = arena_alloc<u64>(arena, cap);
    t.vals = with_vals ? arena_alloc<int32_t>(arena, cap) : nullptr;
    t.mask = cap - 1;
    if (!t.keys || (with_vals && !t.vals)) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    ASR_HIP_CHECK(ctx, hipMemsetAsync(t.keys, 0, cap * sizeof(u64), ctx->stream));
    return ASR_HIP_OK;
}

}  // namespace

// ==========================================================================================
// entry points used by asr_api.hip
// ==========================================================================================
int asr_geom_point_keys(asr_hip_context* ctx, const asr_octree_frame* frame, const float* pts,
                        const float* radii, i64 n, float radius_scale, int max_depth, u64* keys) {
    if (n <= 0) return ASR_HIP_OK;
    k_point_keys<<<grid_for(n, BLK), BLK, 0, ctx->stream>>>(*frame, pts, radii, n, radius_scale,
                                                            max_depth, keys);
    ASR_CHECK_LAUNCH(ctx);
    return ASR_HIP_OK;
}

// Result: ctx->nodes / ctx->leaves (sorted) in the persist arena.
// Octree::Grow: the grown key set of `grow_steps` iterations as a device list in `arena` (see k_grow_*)
static int grow_keys(asr_hip_context* ctx, Arena& arena, const asr_octree_frame* frame, const float* pts, const float* radii,
                     i64 n, float radius_scale, int max_depth, int grow_steps, u64** keys_out, i64* count_out) {
    int host[16];
    u64 cap = next_pow2((u64)std::max<i64>(i64(1) << 16, 8 * n));
    for (int attempt = 0; attempt < 6; ++attempt, cap <<= 2) {
        HashTab t;
        ASR_TRY(make_table(ctx, arena, cap, false, t));
        const int lcap = (int)std::min<u64>(cap / 2, u64(1) << 30);
        u64* list = arena_alloc<u64>(arena, (size_t)lcap + 8);
        if (!list) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
        ASR_TRY(fresh_flags(ctx));
        k_grow_init<<<grid_for(n, BLK), BLK, 0, ctx->stream>>>(*frame, pts, radii, n, radius_scale, max_depth, t,
                                                             ctx->d_flags, list, lcap);
        ASR_CHECK_LAUNCH(ctx);
        ASR_TRY(read_flags(ctx, host));
        if (host[3]) ASR_FAIL(ctx, ASR_HIP_EINVAL, "octree: extra_keys holds a value that is no location code (0 or a leading "
                                                   "bit that is not at 3 * level)");
        if (host[11]) ASR_FAIL(ctx, ASR_HIP_EINVAL, "octree: points / radii contain non-finite values");
        bool overflow = host[1] != 0;
        int lo = 0, hi = host[0];
        for (int it = 0; it < grow_steps && !overflow && hi > lo; ++it) {
            const int count = hi - lo;
            if ((i64)count * 14 >= (i64(1) << 31) || (u64)(hi + (i64)count * 14) > (u64)lcap) {
                overflow = true;  // worst case of this iteration does not fit: larger table
                break;
            }
            u64* cand = arena_alloc<u64>(arena, (size_t)count * 7);
            if (!cand) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
            k_grow_candidates<<<grid_for((i64)count * 7, BLK), BLK, 0, ctx->stream>>>(t, list, lo, count, cand);
            ASR_CHECK_LAUNCH(ctx);
            k_grow_insert<<<grid_for((i64)count * 14, BLK), BLK, 0, ctx->stream>>>(t, list, lo, count, cand, ctx->d_flags,
                                                                                  lcap);
            ASR_CHECK_LAUNCH(ctx);
            ASR_TRY(read_flags(ctx, host));
            overflow = host[1] != 0;
            lo = hi;
            hi = host[0];
        }
        if (overflow) continue;
        *keys_out = list;
        *count_out = host[0];
        return ASR_HIP_OK;
    }
    ASR_FAIL(ctx, ASR_HIP_ELOGIC, "octree grow: hash table overflow");
}

int asr_geom_octree_build(asr_hip_context* ctx, const asr_octree_frame* frame, const float* pts,
                          const float* radii, i64 n, float radius_scale, int max_depth, int grow_steps,
                          const u64* extra_keys, i64 num_extra, bool balance) {
    ASR_TRY(ensure_flags(ctx));
    if (max_depth > ASR_MAX_LEVEL) max_depth = ASR_MAX_LEVEL;
    if (max_depth < 0) ASR_FAIL(ctx, ASR_HIP_EINVAL, "max_depth must be >= 0");
    if (grow_steps < 0) ASR_FAIL(ctx, ASR_HIP_EINVAL, "grow_steps must be >= 0");
    // Octree::Grow (octree.cpp:266) works on the raw point keys, before ancestors and siblings exist: its result
    // replaces the points as the input of the closure below
    u64* grown = nullptr;
    i64 num_grown = 0;
    Arena grow_arena;
    grow_arena.min_slab = size_t(16) << 20;
    struct ArenaGuard {
        Arena& a;
        ~ArenaGuard() { a.release(); }
    } grow_guard{grow_arena};
    if (grow_steps > 0 && n > 0)
        ASR_TRY(grow_keys(ctx, grow_arena, frame, pts, radii, n, radius_scale, max_depth, grow_steps, &grown, &num_grown));
    // nodes of a scan are ~0.3 n; the table must stay at most half full (cap / 2 list entries), else retry 4x larger
    u64 cap = next_pow2((u64)std::max<i64>(i64(1) << 16, 2 * std::max(std::max(n, 4 * num_grown), num_extra)));
    int host[16];
    for (int attempt = 0; attempt < 7; ++attempt, cap <<= 2) {
        ctx->scratch.reset();
        HashTab t;
        ASR_TRY(make_table(ctx, ctx->scratch, cap, false, t));
        // the table is kept at most half full, so the node list never exceeds cap / 2 entries
        int lcap = (int)std::min<u64>(cap / 2, u64(1) << 30);
        u64* list = arena_alloc<u64>(ctx->scratch, (size_t)lcap + 8);
        uint8_t* flag = arena_alloc<uint8_t>(ctx->scratch, (size_t)lcap / 8 + 8);
        if (!list || !flag) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
        ASR_TRY(fresh_flags(ctx));
        if (grown) {
            if (num_grown > 0)
                k_octree_insert_keys<<<grid_for(num_grown, BLK), BLK, 0, ctx->stream>>>(grown, num_grown, t, ctx->d_flags,
                                                                                       list, lcap);
            ASR_CHECK_LAUNCH(ctx);
        } else if (n > 0) {
            k_octree_insert_points<<<grid_for(n, BLK), BLK, 0, ctx->stream>>>(*frame, pts, radii, n, radius_scale, max_depth,
                                                                              t, ctx->d_flags, list, lcap);
            ASR_CHECK_LAUNCH(ctx);
        }
        if (num_extra > 0) {  // node keys of other builds (the local octrees of the other ranks): same closure
            k_check_keys<<<grid_for(num_extra, BLK), BLK, 0, ctx->stream>>>(extra_keys, num_extra, ctx->d_flags);
            ASR_CHECK_LAUNCH(ctx);
            k_octree_insert_keys<<<grid_for(num_extra, BLK), BLK, 0, ctx->stream>>>(extra_keys, num_extra, t, ctx->d_flags,
                                                                                   list, lcap);
            ASR_CHECK_LAUNCH(ctx);
        }
        k_list_key_bits<<<256, BLK, 0, ctx->stream>>>(list, lcap, ctx->d_flags);  // (balancing adds no longer keys)
        ASR_CHECK_LAUNCH(ctx);
        ASR_TRY(read_flags(ctx, host));
        if (host[11]) ASR_FAIL(ctx, ASR_HIP_EINVAL, "octree: points / radii contain non-finite values");
        bool overflow = host[1] != 0 || (u64)host[0] * 2 > cap;
        int lo = 0, hi = host[0];
        while (balance && !overflow && hi > lo) {
            int ngroups = (hi - lo) / 8;
            k_balance_classify<<<grid_for(ngroups, BLK), BLK, 0, ctx->stream>>>(t, list, lo, ngroups, flag);
            ASR_CHECK_LAUNCH(ctx);
            k_balance_insert<<<grid_for((i64)ngroups * 6, BLK), BLK, 0, ctx->stream>>>(
                    t, list, lo, ngroups, flag, ctx->d_flags, lcap);
            ASR_CHECK_LAUNCH(ctx);
            ASR_TRY(read_flags(ctx, host));
            overflow = host[1] != 0 || (u64)host[0] * 2 > cap;
            lo = hi;
            hi = host[0];
        }
        if (overflow) continue;
        i64 num_nodes = (i64)host[0] + (host[9] ? 1 : 0);
        ctx->leaf_lmin = ctx->leaf_lmax = -1;
        if (num_nodes == 0) {
            ctx->num_nodes = ctx->num_leaves = 0;
            ctx->nodes = ctx->leaves = nullptr;
            return ASR_HIP_OK;
        }
        if (host[9]) {  // the root closes the list
            const u64 one = 1;
            ASR_HIP_CHECK(ctx, hipMemcpyAsync(list + host[0], &one, sizeof(u64), hipMemcpyHostToDevice,
                                              ctx->stream));
        }
        ctx->nodes = arena_alloc<u64>(ctx->persist, num_nodes);
        u64* leaves_tmp = arena_alloc<u64>(ctx->scratch, num_nodes);
        uint8_t* lflag = arena_alloc<uint8_t>(ctx->scratch, num_nodes);
        i64* d_num = arena_alloc<i64>(ctx->scratch, 1);
        if (!ctx->nodes || !leaves_tmp || !lflag || !d_num) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
        // (every node key is a prefix of an inserted key or a sibling / face neighbour of one: never longer)
        ASR_TRY(sort_keys(ctx, ctx->scratch, list, ctx->nodes, num_nodes, host[13] > 0 && host[13] <= 64 ? host[13] : 64));
        k_leaf_flags<<<grid_for(num_nodes, BLK), BLK, 0, ctx->stream>>>(t, ctx->nodes, num_nodes, lflag);
        ASR_CHECK_LAUNCH(ctx);
        {
            size_t tb = 0;
            ASR_HIP_CHECK(ctx, rocprim::select(nullptr, tb, ctx->nodes, lflag, leaves_tmp, d_num, (size_t)num_nodes,
                                               ctx->stream));
            void* tmp = ctx->scratch.alloc(tb ? tb : 256);
            if (!tmp) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
            ASR_HIP_CHECK(ctx, rocprim::select(tmp, tb, ctx->nodes, lflag, leaves_tmp, d_num, (size_t)num_nodes,
                                               ctx->stream));
        }
        i64* d_tail = arena_alloc<i64>(ctx->scratch, 3);
        if (!d_tail) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
        k_leaf_tail<<<1, 1, 0, ctx->stream>>>(leaves_tmp, d_num, d_tail);
        ASR_CHECK_LAUNCH(ctx);
        i64 tail[3] = {0, 0, 0};
        ASR_HIP_CHECK(ctx, hipMemcpyAsync(tail, d_tail, sizeof(tail), hipMemcpyDeviceToHost, ctx->stream));
        ASR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        const i64 num_leaves = tail[0];
        auto level_of = [](u64 k) { return k ? (63 - __builtin_clzll(k)) / 3 : 0; };
        ctx->leaf_lmin = level_of((u64)tail[1]);
        ctx->leaf_lmax = level_of((u64)tail[2]);
        ctx->leaves = arena_alloc<u64>(ctx->persist, num_leaves);
        if (!ctx->leaves) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
        ASR_HIP_CHECK(ctx, hipMemcpyAsync(ctx->leaves, leaves_tmp, 8 * num_leaves, hipMemcpyDeviceToDevice, ctx->stream));
        ctx->num_nodes = num_nodes;
        ctx->num_leaves = num_leaves;
        return ASR_HIP_OK;
    }
    ASR_FAIL(ctx, ASR_HIP_ELOGIC, "octree hash table overflow after 6 growth attempts");
}

static int build_key_map(asr_hip_context* ctx, const u64* keys, i64 v, HashTab& t, int grow = 0) {
    ASR_TRY(ensure_flags(ctx));
    u64 cap = next_pow2((u64)std::max<i64>(1024, 2 * v)) << grow;
    ASR_TRY(make_table(ctx, ctx->scratch, cap, true, t));
    ASR_TRY(fresh_flags(ctx));
    k_map_build<<<grid_for(v, BLK), BLK, 0, ctx->stream>>>(keys, v, t, ctx->d_flags);
    ASR_CHECK_LAUNCH(ctx);
    return ASR_HIP_OK;
}

// The same for the stand-alone operators: the overflow flag is read back (one synchronisation) and a map that lost keys
// -- lattice-like key sets fill "their" positions of the 64-slot buckets early -- is rebuilt four times the size, so that
// every entry point either sees all keys or fails; none drops neighbours silently.
static int build_key_map_complete(asr_hip_context* ctx, const u64* keys, i64 v, HashTab& t) {
    for (int grow = 0;; grow += 2) {
        ASR_TRY(build_key_map(ctx, keys, v, t, grow));
        int host[16];
        ASR_HIP_CHECK(ctx, hipMemcpyAsync(host, ctx->d_flags, 16 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        ASR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (!host[1]) return ASR_HIP_OK;
        if (grow >= 6) ASR_FAIL(ctx, ASR_HIP_ELOGIC, "voxel key map overflow");
    }
}

static int check_row_list(asr_hip_context* ctx, const int32_t* rows, i64 nrows, i64 v) {
    ASR_TRY(ensure_flags(ctx));
    ASR_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_flags + 3, 0, sizeof(int), ctx->stream));
    k_check_rows<<<grid_for(nrows, BLK), BLK, 0, ctx->stream>>>(rows, nrows, v, ctx->d_flags);
    ASR_CHECK_LAUNCH(ctx);
    int host[16];
    ASR_TRY(read_flags(ctx, host));
    if (host[3]) ASR_FAIL(ctx, ASR_HIP_EINVAL, "neighbour rows: the row list must be strictly ascending indices in [0, %lld)", (long long)v);
    return ASR_HIP_OK;
}

int asr_geom_neighbors_count(asr_hip_context* ctx, const u64* keys, i64 v, i64* rs, i64* num_pairs) {
    if (v <= 0) {
        *num_pairs = 0;
        if (rs) ASR_HIP_CHECK(ctx, hipMemsetAsync(rs, 0, sizeof(i64), ctx->stream));
        return ASR_HIP_OK;
    }
    HashTab t;
    ASR_TRY(build_key_map_complete(ctx, keys, v, t));
    i64* counts = arena_alloc<i64>(ctx->scratch, v + 1);
    if (!counts) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    k_neighbors_count<<<grid_for(v + 1, BLK), BLK, 0, ctx->stream>>>(keys, v, t, counts, nullptr, nullptr, nullptr);
    ASR_CHECK_LAUNCH(ctx);
    ASR_TRY(scan_counts(ctx, ctx->scratch, counts, rs, v + 1));
    ASR_TRY(read_i64(ctx, rs + v, num_pairs));
    return ASR_HIP_OK;
}
int asr_geom_neighbors_fill(asr_hip_context* ctx, const u64* keys, i64 v, const i64* rs,
                            int32_t* idx, uint8_t* kidx) {
    if (v <= 0) return ASR_HIP_OK;
    HashTab t;
    ASR_TRY(build_key_map_complete(ctx, keys, v, t));
    k_neighbors_fill<<<grid_for(v, BLK), BLK, 0, ctx->stream>>>(keys, v, t, rs, nullptr, nullptr, nullptr, idx, kidx);
    ASR_CHECK_LAUNCH(ctx);
    return ASR_HIP_OK;
}
int asr_geom_neighbors_rows_count(asr_hip_context* ctx, const u64* keys, i64 v, const int32_t* rows, i64 nrows, i64* rs,
                                  i64* num_pairs) {
    *num_pairs = 0;
    if (v <= 0) {
        if (rs) ASR_HIP_CHECK(ctx, hipMemsetAsync(rs, 0, sizeof(i64), ctx->stream));
        return ASR_HIP_OK;
    }
    HashTab t;
    ASR_TRY(build_key_map_complete(ctx, keys, v, t));
    i64* counts = arena_alloc<i64>(ctx->scratch, v + 1);
    if (!counts) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    ASR_HIP_CHECK(ctx, hipMemsetAsync(counts, 0, (v + 1) * sizeof(i64), ctx->stream));
    if (nrows > 0) {
        ASR_TRY(check_row_list(ctx, rows, nrows, v));
        k_neighbors_count_rows<<<grid_for(nrows, BLK), BLK, 0, ctx->stream>>>(keys, t, rows, nrows, counts);
        ASR_CHECK_LAUNCH(ctx);
    }
    ASR_TRY(scan_counts(ctx, ctx->scratch, counts, rs, v + 1));
    ASR_TRY(read_i64(ctx, rs + v, num_pairs));
    return ASR_HIP_OK;
}
int asr_geom_neighbors_rows_fill(asr_hip_context* ctx, const u64* keys, i64 v, const int32_t* rows, i64 nrows,
                                 const i64* rs, int32_t* idx, uint8_t* kidx) {
    if (v <= 0 || nrows <= 0) return ASR_HIP_OK;
    HashTab t;
    ASR_TRY(build_key_map_complete(ctx, keys, v, t));
    ASR_TRY(check_row_list(ctx, rows, nrows, v));
    k_neighbors_fill_rows<<<grid_for(nrows, BLK), BLK, 0, ctx->stream>>>(keys, t, rows, nrows, rs, idx, kidx);
    ASR_CHECK_LAUNCH(ctx);
    return ASR_HIP_OK;
}
// fused variant for the whole-path driver: one map build, masks carried from count to fill.
int asr_geom_neighbors_build(asr_hip_context* ctx, Arena& out_arena, const u64* keys, i64 v,
                             i64** rs_out, int32_t** idx_out, uint8_t** kidx_out, i64* num_pairs) {
    i64* rs = arena_alloc<i64>(out_arena, v + 1);
    if (!rs) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    for (int grow = 0;; grow += 2) {  // a key map that overflowed (see TabProbe) is rebuilt four times the size
        HashTab t;
        ASR_TRY(build_key_map(ctx, keys, v, t, grow));
        i64* counts = arena_alloc<i64>(ctx->scratch, v + 1);
        u64* masks = arena_alloc<u64>(ctx->scratch, v);
        int32_t* st_idx = arena_alloc<int32_t>(ctx->scratch, (size_t)v * NB_STAGE);
        uint8_t* st_slot = arena_alloc<uint8_t>(ctx->scratch, (size_t)v * NB_STAGE);
        if (!counts || !masks || !st_idx || !st_slot) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
        k_neighbors_count<<<grid_for(v + 1, BLK), BLK, 0, ctx->stream>>>(keys, v, t, counts, masks, st_idx, st_slot);
        ASR_CHECK_LAUNCH(ctx);
        ASR_TRY(scan_counts(ctx, ctx->scratch, counts, rs, v + 1));
        int host[16];
        ASR_HIP_CHECK(ctx, hipMemcpyAsync(host, ctx->d_flags, 16 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        ASR_TRY(read_i64(ctx, rs + v, num_pairs));  // synchronises: the flags have arrived as well
        if (host[1]) {
            if (grow >= 6) ASR_FAIL(ctx, ASR_HIP_ELOGIC, "voxel key map overflow");
            continue;
        }
        int32_t* idx = arena_alloc<int32_t>(out_arena, *num_pairs);
        uint8_t* kidx = arena_alloc<uint8_t>(out_arena, *num_pairs);
        if (!idx || !kidx) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
        k_neighbors_fill<<<grid_for(v, BLK), BLK, 0, ctx->stream>>>(keys, v, t, rs, masks, st_idx, st_slot, idx, kidx);
        ASR_CHECK_LAUNCH(ctx);
        *rs_out = rs;
        *idx_out = idx;
        *kidx_out = kidx;
        return ASR_HIP_OK;
    }
}

int asr_geom_neighbors_build_batch(asr_hip_context* ctx, Arena& out_arena, asr_nb_job* jobs, int n) {
    if (n < 1 || n > NB_MAX_JOBS) ASR_FAIL(ctx, ASR_HIP_EINVAL, "neighbors_build_batch: 1..%d jobs", NB_MAX_JOBS);
    ASR_TRY(ensure_flags(ctx));
    NbBatch b;
    b.n = n;
    i64 total = 0;
    for (int j = 0; j < n; ++j) {
        if (jobs[j].v <= 0 || !jobs[j].keys) ASR_FAIL(ctx, ASR_HIP_EINVAL, "neighbors_build_batch: empty grid");
        b.base[j] = total;
        b.v[j] = jobs[j].v;
        b.keys[j] = jobs[j].keys;
        total += jobs[j].v + 1;
        jobs[j].rs = arena_alloc<i64>(out_arena, jobs[j].v + 1);
        if (!jobs[j].rs) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
        b.rs[j] = jobs[j].rs;
        b.idx[j] = nullptr;
        b.kidx[j] = nullptr;
        b.owner[j] = jobs[j].owner;
        b.me[j] = jobs[j].me;
    }
    b.base[n] = total;
    for (int grow = 0;; grow += 2) {  // key maps that overflowed (see TabProbe) are rebuilt four times the size
        u64 caps[NB_MAX_JOBS], cap_sum = 0;
        for (int j = 0; j < n; ++j) {
            caps[j] = next_pow2((u64)std::max<i64>(1024, 2 * jobs[j].v)) << grow;
            cap_sum += caps[j];
        }
        u64* tkeys = arena_alloc<u64>(ctx->scratch, cap_sum);  // one allocation, one memset for all tables
        int32_t* tvals = arena_alloc<int32_t>(ctx->scratch, cap_sum);
        i64* counts = arena_alloc<i64>(ctx->scratch, total + 1);
        i64* scan = arena_alloc<i64>(ctx->scratch, total + 1);
        u64* masks = arena_alloc<u64>(ctx->scratch, total);
        int32_t* st_idx = arena_alloc<int32_t>(ctx->scratch, (size_t)total * NB_STAGE);
        uint8_t* st_slot = arena_alloc<uint8_t>(ctx->scratch, (size_t)total * NB_STAGE);
        i64* totals = arena_alloc<i64>(ctx->scratch, NB_MAX_JOBS);
        if (!tkeys || !tvals || !counts || !scan || !masks || !st_idx || !st_slot || !totals)
            ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
        u64 off = 0;
        for (int j = 0; j < n; ++j) {
            b.tab[j] = HashTab{tkeys + off, tvals + off, caps[j] - 1};
            off += caps[j];
        }
        ASR_HIP_CHECK(ctx, hipMemsetAsync(tkeys, 0, cap_sum * sizeof(u64), ctx->stream));
        ASR_TRY(fresh_flags(ctx));
        ASR_HIP_CHECK(ctx, hipMemsetAsync(counts + total, 0, sizeof(i64), ctx->stream));
        k_map_build_batch<<<grid_for(total, BLK), BLK, 0, ctx->stream>>>(b, ctx->d_flags);
        ASR_CHECK_LAUNCH(ctx);
        k_neighbors_count_batch<<<grid_for(total, BLK), BLK, 0, ctx->stream>>>(b, counts, masks, st_idx, st_slot);
        ASR_CHECK_LAUNCH(ctx);
        ASR_TRY(scan_counts(ctx, ctx->scratch, counts, scan, total + 1));
        k_nb_totals<<<1, 64, 0, ctx->stream>>>(b, scan, totals);
        ASR_CHECK_LAUNCH(ctx);
        int host_flags[16];
        i64 host_totals[NB_MAX_JOBS];
        ASR_HIP_CHECK(ctx, hipMemcpyAsync(host_flags, ctx->d_flags, 16 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        ASR_HIP_CHECK(ctx, hipMemcpyAsync(host_totals, totals, n * sizeof(i64), hipMemcpyDeviceToHost, ctx->stream));
        ASR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (host_flags[1]) {
            if (grow >= 6) ASR_FAIL(ctx, ASR_HIP_ELOGIC, "voxel key map overflow");
            continue;
        }
        for (int j = 0; j < n; ++j) {
            jobs[j].p = host_totals[j];
            jobs[j].idx = arena_alloc<int32_t>(out_arena, jobs[j].p);
            jobs[j].kidx = arena_alloc<uint8_t>(out_arena, jobs[j].p);
            if (!jobs[j].idx || !jobs[j].kidx) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
            b.idx[j] = jobs[j].idx;
            b.kidx[j] = jobs[j].kidx;
        }
        k_neighbors_fill_batch<<<grid_for(total, BLK), BLK, 0, ctx->stream>>>(b, scan, masks, st_idx, st_slot);
        ASR_CHECK_LAUNCH(ctx);
        return ASR_HIP_OK;
    }
}

int asr_geom_row_groups(asr_hip_context* ctx, const uint8_t* kidx, const i64* rs, i64 v, i64 seg,
                        int32_t* perm_out, int kbits) {
    if (kbits < 1 || kbits > 56) kbits = 56;  // slot masks only use bits below the kernel size
    if (v <= 0) return ASR_HIP_OK;
    if (seg < 16) seg = 16;
    if (v >= (i64(1) << 31)) ASR_FAIL(ctx, ASR_HIP_EINVAL, "row_groups: too many rows");
    u64* masks = arena_alloc<u64>(ctx->scratch, v);
    u64* masks_s = arena_alloc<u64>(ctx->scratch, v);
    int32_t* ids = arena_alloc<int32_t>(ctx->scratch, v);
    if (!masks || !masks_s || !ids) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    k_row_masks<<<grid_for(v, BLK), BLK, 0, ctx->stream>>>(kidx, rs, v, masks, ids);
    ASR_CHECK_LAUNCH(ctx);
    // order by (segment, mask) with two stable LSD radix sorts: by mask, then by segment id
    const bool lpt = ctx->opt.row_lpt != 0;
    int32_t* perm_m = perm_out;  // (segment, mask) order; re-ordered by chunk cost below
    if (lpt && v > 128) {
        perm_m = arena_alloc<int32_t>(ctx->scratch, v);
        if (!perm_m) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    }
    if (seg >= v) {
        ASR_TRY((sort_pairs<u64, int32_t>(ctx, ctx->scratch, masks, masks_s, ids, perm_m, v, kbits)));
    } else {
        int32_t* ids_m = arena_alloc<int32_t>(ctx->scratch, v);
        int32_t* segk = arena_alloc<int32_t>(ctx->scratch, v);
        int32_t* segk_s = arena_alloc<int32_t>(ctx->scratch, v);
        if (!ids_m || !segk || !segk_s) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
        ASR_TRY((sort_pairs<u64, int32_t>(ctx, ctx->scratch, masks, masks_s, ids, ids_m, v, kbits)));
        k_segment_of<<<grid_for(v, BLK), BLK, 0, ctx->stream>>>(ids_m, v, seg, segk);
        ASR_CHECK_LAUNCH(ctx);
        ASR_TRY((sort_pairs<int32_t, int32_t>(ctx, ctx->scratch, segk, segk_s, ids_m, perm_m, v,
                                              bits_for((v + seg - 1) / seg + 1))));
    }
    if (perm_m != perm_out) {
        const i64 nc = (v + 127) / 128;
        int32_t* ckey = arena_alloc<int32_t>(ctx->scratch, nc);
        int32_t* ckey_s = arena_alloc<int32_t>(ctx->scratch, nc);
        int32_t* cid = arena_alloc<int32_t>(ctx->scratch, nc);
        int32_t* cid_s = arena_alloc<int32_t>(ctx->scratch, nc);
        if (!ckey || !ckey_s || !cid || !cid_s) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
        int32_t* tcost = arena_alloc<int32_t>(ctx->scratch, 2 * nc);
        if (!tcost) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
        k_tile_cost<<<grid_for(((v + 63) / 64) * 64, BLK), BLK, 0, ctx->stream>>>(perm_m, masks, v, tcost);
        ASR_CHECK_LAUNCH(ctx);
        k_chunk_key<<<grid_for(nc, BLK), BLK, 0, ctx->stream>>>(tcost, v, seg, ckey, cid);
        ASR_CHECK_LAUNCH(ctx);
        ASR_TRY((sort_pairs<int32_t, int32_t>(ctx, ctx->scratch, ckey, ckey_s, cid, cid_s, nc,
                                              7 + bits_for((v + seg - 1) / seg + 1))));
        k_chunk_apply<<<grid_for(v, BLK), BLK, 0, ctx->stream>>>(perm_m, cid_s, v, perm_out);
        ASR_CHECK_LAUNCH(ctx);
    }
    return ASR_HIP_OK;
}

// all CSRs of one hierarchy in one pass (see k_rg_keys); falls back to the per-CSR routine for shapes the
// packed key cannot hold
static int rg_batch_run(asr_hip_context* ctx, const asr_row_group_job* jobs, int n, i64 seg, int mask_bits);
static int rg_batch_run_ranked(asr_hip_context* ctx, const asr_row_group_job* jobs, int n, i64 seg, bool* done);
int asr_geom_row_groups_batch(asr_hip_context* ctx, const asr_row_group_job* jobs, int n, i64 seg) {
    if (seg < 128) seg = 128;
    bool fits = n <= RG_MAX_JOBS && seg % 128 == 0;
    for (int j = 0; j < n && fits; ++j) fits = jobs[j].v / seg < 63 && jobs[j].v < (i64(1) << 31);
    if (!fits) {
        for (int j = 0; j < n; ++j)
            ASR_TRY(asr_geom_row_groups(ctx, jobs[j].kidx, jobs[j].rs, jobs[j].v, seg, jobs[j].perm_out, jobs[j].kbits));
        return ASR_HIP_OK;
    }
    // The radix sort of the keys is what this routine costs.  Lists with few slots (the 9-slot up / down lists: more than
    // half of all rows of a hierarchy) go through it with 19-bit keys, three digit passes instead of eight.
    if (ctx->opt.row_ranked) {  // all lists in one sort on (job, segment, rank of the mask) keys
        bool done = false;
        ASR_TRY(rg_batch_run_ranked(ctx, jobs, n, seg, &done));
        if (done) return ASR_HIP_OK;
    }
    std::vector<asr_row_group_job> small_jobs, wide_jobs;
    for (int j = 0; j < n; ++j) (jobs[j].kbits <= 10 ? small_jobs : wide_jobs).push_back(jobs[j]);
    if (!small_jobs.empty()) ASR_TRY(rg_batch_run(ctx, small_jobs.data(), (int)small_jobs.size(), seg, 9));
    if (!wide_jobs.empty()) ASR_TRY(rg_batch_run(ctx, wide_jobs.data(), (int)wide_jobs.size(), seg, 54));
    return ASR_HIP_OK;
}
// *done = false: more than RG_MAX_MASKS distinct masks (or a full mask set): nothing written, the caller sorts on full masks
static int rg_batch_run_ranked(asr_hip_context* ctx, const asr_row_group_job* jobs, int n, i64 seg, bool* done) {
    *done = false;
    ASR_TRY(ensure_flags(ctx));
    RowGroupBatch b;
    b.n = 0;
    i64 total = 0;
    for (int j = 0; j < n; ++j) {
        if (jobs[j].v <= 0) continue;
        b.base[b.n] = total;
        b.v[b.n] = jobs[j].v;
        b.kidx[b.n] = jobs[j].kidx;
        b.rs[b.n] = jobs[j].rs;
        b.out[b.n] = jobs[j].perm_out;
        total += (jobs[j].v + 127) / 128 * 128;
        ++b.n;
    }
    if (b.n == 0) {
        *done = true;
        return ASR_HIP_OK;
    }
    b.base[b.n] = total;
    if (total >= (i64(1) << 31)) ASR_FAIL(ctx, ASR_HIP_EINVAL, "row_groups: too many rows");
    u64* masks = arena_alloc<u64>(ctx->scratch, total);
    unsigned* keys = arena_alloc<unsigned>(ctx->scratch, total);
    unsigned* keys_s = arena_alloc<unsigned>(ctx->scratch, total);
    int32_t* ids = arena_alloc<int32_t>(ctx->scratch, total);
    int32_t* ids_s = arena_alloc<int32_t>(ctx->scratch, total);
    const i64 nc = total / 128;
    int32_t* ckey = arena_alloc<int32_t>(ctx->scratch, nc);
    int32_t* ckey_s = arena_alloc<int32_t>(ctx->scratch, nc);
    int32_t* cid = arena_alloc<int32_t>(ctx->scratch, nc);
    int32_t* cid_s = arena_alloc<int32_t>(ctx->scratch, nc);
    if (!masks || !keys || !keys_s || !ids || !ids_s || !ckey || !ckey_s || !cid || !cid_s)
        ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    HashTab t;
    ASR_TRY(make_table(ctx, ctx->scratch, RG_MASK_TAB, true, t));
    ASR_TRY(fresh_flags(ctx));
    k_rg_masks<<<grid_for(total, BLK), BLK, 0, ctx->stream>>>(b, masks, t, ctx->d_flags);
    ASR_CHECK_LAUNCH(ctx);
    u64* mlist = arena_alloc<u64>(ctx->scratch, RG_MAX_MASKS);
    int32_t* mslot = arena_alloc<int32_t>(ctx->scratch, RG_MAX_MASKS);
    if (!mlist || !mslot) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    k_rg_mask_collect<<<grid_for((i64)t.mask + 1, BLK), BLK, 0, ctx->stream>>>(t, mlist, mslot, ctx->d_flags);
    ASR_CHECK_LAUNCH(ctx);
    k_rg_mask_ranks<<<RG_MAX_MASKS / 256, 256, 0, ctx->stream>>>(t, mlist, mslot, ctx->d_flags);
    ASR_CHECK_LAUNCH(ctx);
    int host[16];
    ASR_TRY(read_flags(ctx, host));
    if (host[1] || host[2] > RG_MAX_MASKS) return ASR_HIP_OK;  // too many distinct masks for the ranked keys
    const int rank_bits = bits_for(std::max(host[2], 2));
    k_rg_keys_ranked<<<grid_for(total, BLK), BLK, 0, ctx->stream>>>(b, seg, masks, t, keys, ids, rank_bits);
    ASR_CHECK_LAUNCH(ctx);
    ASR_TRY((sort_pairs<unsigned, int32_t>(ctx, ctx->scratch, keys, keys_s, ids, ids_s, total, 4 + 6 + rank_bits)));
    const int lpt = ctx->opt.row_lpt != 0;
    if (lpt) {
        k_rg_chunk_keys<<<grid_for(nc * 64, BLK), BLK, 0, ctx->stream>>>(b, seg, ids_s, masks, ckey, cid);
        ASR_CHECK_LAUNCH(ctx);
        ASR_TRY((sort_pairs<int32_t, int32_t>(ctx, ctx->scratch, ckey, ckey_s, cid, cid_s, nc, 17)));
    }
    k_rg_apply<<<grid_for(total, BLK), BLK, 0, ctx->stream>>>(b, ids_s, cid_s, lpt);
    ASR_CHECK_LAUNCH(ctx);
    *done = true;
    return ASR_HIP_OK;
}
static int rg_batch_run(asr_hip_context* ctx, const asr_row_group_job* jobs, int n, i64 seg, int mask_bits) {
    RowGroupBatch b;
    b.n = 0;
    i64 total = 0;
    for (int j = 0; j < n; ++j) {
        if (jobs[j].v <= 0) continue;
        b.base[b.n] = total;
        b.v[b.n] = jobs[j].v;
        b.kidx[b.n] = jobs[j].kidx;
        b.rs[b.n] = jobs[j].rs;
        b.out[b.n] = jobs[j].perm_out;
        total += (jobs[j].v + 127) / 128 * 128;
        ++b.n;
    }
    if (b.n == 0) return ASR_HIP_OK;
    b.base[b.n] = total;
    if (total >= (i64(1) << 31)) ASR_FAIL(ctx, ASR_HIP_EINVAL, "row_groups: too many rows");
    u64* keys = arena_alloc<u64>(ctx->scratch, total);
    u64* keys_s = arena_alloc<u64>(ctx->scratch, total);
    u64* masks = arena_alloc<u64>(ctx->scratch, total);
    int32_t* ids = arena_alloc<int32_t>(ctx->scratch, total);
    int32_t* ids_s = arena_alloc<int32_t>(ctx->scratch, total);
    const i64 nc = total / 128;
    int32_t* ckey = arena_alloc<int32_t>(ctx->scratch, nc);
    int32_t* ckey_s = arena_alloc<int32_t>(ctx->scratch, nc);
    int32_t* cid = arena_alloc<int32_t>(ctx->scratch, nc);
    int32_t* cid_s = arena_alloc<int32_t>(ctx->scratch, nc);
    if (!keys || !keys_s || !masks || !ids || !ids_s || !ckey || !ckey_s || !cid || !cid_s)
        ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    k_rg_keys<<<grid_for(total, BLK), BLK, 0, ctx->stream>>>(b, seg, keys, masks, ids, mask_bits);
    ASR_CHECK_LAUNCH(ctx);
    ASR_TRY((sort_pairs<u64, int32_t>(ctx, ctx->scratch, keys, keys_s, ids, ids_s, total, 4 + 6 + mask_bits)));
    const int lpt = ctx->opt.row_lpt != 0;
    if (lpt) {
        k_rg_chunk_keys<<<grid_for(nc * 64, BLK), BLK, 0, ctx->stream>>>(b, seg, ids_s, masks, ckey, cid);
        ASR_CHECK_LAUNCH(ctx);
        ASR_TRY((sort_pairs<int32_t, int32_t>(ctx, ctx->scratch, ckey, ckey_s, cid, cid_s, nc, 17)));
    }
    k_rg_apply<<<grid_for(total, BLK), BLK, 0, ctx->stream>>>(b, ids_s, cid_s, lpt);
    ASR_CHECK_LAUNCH(ctx);
    return ASR_HIP_OK;
}

int asr_geom_coarsen_count(asr_hip_context* ctx, const u64* keys, i64 v, i64* v_out) {
    ASR_TRY(ensure_flags(ctx));
    if (v <= 0) {
        *v_out = 0;
        return ASR_HIP_OK;
    }
    int host[16];
    ASR_TRY(fresh_flags(ctx));
    k_coarsen_count<<<grid_for(v, BLK), BLK, 0, ctx->stream>>>(keys, v, ctx->d_flags);
    ASR_CHECK_LAUNCH(ctx);
    ASR_TRY(read_flags(ctx, host));
    *v_out = host[5];
    return ASR_HIP_OK;
}
int asr_geom_coarsen_fill(asr_hip_context* ctx, const u64* keys, i64 v, u64* out_keys, i64 v_out,
                          int32_t* up_idx, uint8_t* up_kidx, i64* up_rs, int32_t* down_idx, uint8_t* down_kidx,
                          i64* down_rs) {
    ASR_TRY(ensure_flags(ctx));
    if (v <= 0) return ASR_HIP_OK;
    u64* k_u = arena_alloc<u64>(ctx->scratch, v_out);
    int32_t* s_u = arena_alloc<int32_t>(ctx->scratch, v_out);
    int32_t* s_s = arena_alloc<int32_t>(ctx->scratch, v_out);
    if (!k_u || !s_u || !s_s) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    ASR_TRY(fresh_flags(ctx));
    k_coarsen_emit<<<grid_for(v, BLK), BLK, 0, ctx->stream>>>(keys, v, k_u, s_u, ctx->d_flags);
    ASR_CHECK_LAUNCH(ctx);
    ASR_TRY((sort_pairs<u64, int32_t>(ctx, ctx->scratch, k_u, out_keys, s_u, s_s, v_out, 64)));
    k_coarsen_up<<<grid_for(v_out, BLK), BLK, 0, ctx->stream>>>(keys, v, s_s, v_out, up_idx, up_kidx);
    ASR_CHECK_LAUNCH(ctx);
    k_iota64<<<grid_for(v + 1, BLK), BLK, 0, ctx->stream>>>(up_rs, v + 1);
    ASR_CHECK_LAUNCH(ctx);
    if (down_rs) {  // the inverted lists, rows = coarse voxels
        i64* cnt = arena_alloc<i64>(ctx->scratch, v_out + 1);
        if (!cnt) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
        k_coarsen_down_count<<<grid_for(v_out + 1, BLK), BLK, 0, ctx->stream>>>(keys, v, s_s, v_out, cnt);
        ASR_CHECK_LAUNCH(ctx);
        ASR_TRY(scan_counts(ctx, ctx->scratch, cnt, down_rs, v_out + 1));
        k_coarsen_down_fill<<<grid_for(v_out, BLK), BLK, 0, ctx->stream>>>(keys, v, s_s, v_out, down_rs, down_idx,
                                                                          down_kidx);
        ASR_CHECK_LAUNCH(ctx);
    }
    return ASR_HIP_OK;
}

// Whole-path variant: ONE pass over the fine keys (the emit kernel counts while it appends: no separate counting
// kernel), one size read-back, outputs allocated from `keep`.
int asr_geom_coarsen_build(asr_hip_context* ctx, Arena& keep, const u64* keys, i64 v, u64** out_keys, i64* v_out,
                           int32_t** up_idx, uint8_t** up_kidx, i64** up_rs, int32_t** down_idx, uint8_t** down_kidx,
                           i64** down_rs, int key_bits) {
    ASR_TRY(ensure_flags(ctx));
    *v_out = 0;
    if (v <= 0) return ASR_HIP_OK;
    (void)key_bits;
    // flags -> exclusive scan of the packed counts (kept | heads << 32; both below 2^31) -> one read-back of the totals
    u64* packed = arena_alloc<u64>(ctx->scratch, v + 1);
    u64* pre = arena_alloc<u64>(ctx->scratch, v + 2);
    int32_t* s_s = arena_alloc<int32_t>(ctx->scratch, v);
    if (!packed || !pre || !s_s) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    k_coarsen_flags<<<grid_for(v + 1, BLK), BLK, 0, ctx->stream>>>(keys, v, packed);
    ASR_CHECK_LAUNCH(ctx);
    {
        size_t tb = 0;
        ASR_HIP_CHECK(ctx, rocprim::exclusive_scan(nullptr, tb, packed, pre, u64(0), (size_t)(v + 1), rocprim::plus<u64>(),
                                                   ctx->stream));
        void* tmp = ctx->scratch.alloc(tb ? tb : 256);
        if (!tmp) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
        ASR_HIP_CHECK(ctx, rocprim::exclusive_scan(tmp, tb, packed, pre, u64(0), (size_t)(v + 1), rocprim::plus<u64>(),
                                                   ctx->stream));
    }
    u64 totals = 0;
    ASR_HIP_CHECK(ctx, hipMemcpyAsync(&totals, pre + v, sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    ASR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    const i64 vo = (i64)(totals & 0xffffffffu) + (i64)(totals >> 32);
    *v_out = vo;
    *out_keys = arena_alloc<u64>(keep, vo);
    *up_idx = arena_alloc<int32_t>(keep, v);
    *up_kidx = arena_alloc<uint8_t>(keep, v);
    *up_rs = arena_alloc<i64>(keep, v + 1);
    *down_idx = arena_alloc<int32_t>(keep, v);
    *down_kidx = arena_alloc<uint8_t>(keep, v);
    *down_rs = arena_alloc<i64>(keep, vo + 1);
    i64* cnt = arena_alloc<i64>(ctx->scratch, vo + 1);
    if (!*out_keys || !*up_idx || !*up_kidx || !*up_rs || !*down_idx || !*down_kidx || !*down_rs || !cnt)
        ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    k_coarsen_place<<<grid_for(v, BLK), BLK, 0, ctx->stream>>>(keys, v, pre, *out_keys, s_s);
    ASR_CHECK_LAUNCH(ctx);
    k_coarsen_up<<<grid_for(vo, BLK), BLK, 0, ctx->stream>>>(keys, v, s_s, vo, *up_idx, *up_kidx);
    ASR_CHECK_LAUNCH(ctx);
    k_iota64<<<grid_for(v + 1, BLK), BLK, 0, ctx->stream>>>(*up_rs, v + 1);
    ASR_CHECK_LAUNCH(ctx);
    k_coarsen_down_count<<<grid_for(vo + 1, BLK), BLK, 0, ctx->stream>>>(keys, v, s_s, vo, cnt);
    ASR_CHECK_LAUNCH(ctx);
    ASR_TRY(scan_counts(ctx, ctx->scratch, cnt, *down_rs, vo + 1));
    k_coarsen_down_fill<<<grid_for(vo, BLK), BLK, 0, ctx->stream>>>(keys, v, s_s, vo, *down_rs, *down_idx, *down_kidx);
    ASR_CHECK_LAUNCH(ctx);
    return ASR_HIP_OK;
}

int asr_geom_voxel_info(asr_hip_context* ctx, const asr_octree_frame* frame, const u64* keys,
                        i64 v, float* centers, float* sizes) {
    if (v <= 0) return ASR_HIP_OK;
    k_voxel_info<<<grid_for(v, BLK), BLK, 0, ctx->stream>>>(*frame, keys, v, centers, sizes);
    ASR_CHECK_LAUNCH(ctx);
    return ASR_HIP_OK;
}

// state kept in the scratch arena between _count and _fill
struct RadiusState {
    bool valid = false;
    asr_octree_frame frame;
    float4* sorted = nullptr;  // points in Morton order, w = original index
    int32_t* ids = nullptr;    // original index of the point at each Morton position
    int32_t* rank = nullptr;   // inverse of ids (only when the caller asked for it)
    const u64* codes = nullptr;  // sorted level-21 codes (scratch arena: valid until the next reset)
    u64* codes_w = nullptr;      // the same, writable, and the unsorted inputs of the sort (deepen_point_index)
    const u64* codes_u = nullptr;
    const int32_t* ids_u = nullptr;
    int32_t* ids_w = nullptr;
    int lhash = ASR_MAX_LEVEL;   // finest level in the hash table (finer ones: binary search in codes)
    CellIndex index() const { return {tab, start, end, codes, (int)n, lhash}; }
    HashTab tab;
    int32_t* start = nullptr;
    int32_t* end = nullptr;
    i64 n = 0, v = 0;
    u64* tmp = nullptr;        // [v][RADIUS_LIGHT] sorted keys of the light rows
    int32_t* heavy = nullptr;  // rows with more than RADIUS_LIGHT hits or RADIUS_GIANT candidates
    uint8_t* is_heavy = nullptr;
    i64 num_heavy = 0;
    int* hc_pref = nullptr;    // cell ranges of the heavy rows (k_radius_heavy_cells)
    int* hc_beg = nullptr;
    const u64* dual_leaves = nullptr;  // leaf array of the pending dual-cell count / fill pair
    i64 dual_nl = 0;
    float* srad = nullptr;     // radii in Morton order (when the count call was given them)
    const float* radii_src = nullptr;
    int lsort = 0;             // the points are sorted on the code bits down to this level
    int cell_grow = 0;         // log2 of the extra capacity of the cell table (sticky, raised after an overflow)
    bool aligned = false;      // queries are voxel centres of `aq.keys`: half-size cells + margin pairs (heavy rows)
    bool aligned_main = false; // ... in the per-voxel kernel as well (option search_half = 2)
    AlignedQ aq = {nullptr, nullptr, nullptr, -1};
};
static RadiusState& rstate(asr_hip_context* ctx) {
    if (!ctx->radius_state) ctx->radius_state = new RadiusState();
    return *(RadiusState*)ctx->radius_state;
}
void asr_geom_release(asr_hip_context* ctx) {
    delete (RadiusState*)ctx->radius_state;
    ctx->radius_state = nullptr;
}

// points sorted by level-21 Morton code (sort_points) + hash map (cell, level) -> [start, end) for levels
// lmin..lmax (build_cell_table)
// `keep`: arena for the arrays that outlive the search (Morton-ordered points, ids, rank); the scratch
// arena when null.  `want_rank`: also build the original index -> Morton position table.  `radii`: gathered into
// Morton order as well.  `lsort`: the points are grouped by their level-lsort cell (order inside such a cell:
// input order); `keep_all`: the sorted codes go to `keep` too (asr_geom_presort: they outlive this call).
static int sort_points(asr_hip_context* ctx, const asr_octree_frame* frame, const float* pts, i64 n, int lsort,
                       RadiusState& st, Arena* keep, bool want_rank, const float* radii, bool keep_all = false) {
    st.frame = *frame;
    st.n = n;
    st.lsort = lsort;
    Arena& ka = keep ? *keep : ctx->scratch;
    u64* codes_u = arena_alloc<u64>(ctx->scratch, n + 1);
    u64* codes = arena_alloc<u64>(keep_all ? ka : ctx->scratch, n + 1);
    int32_t* ids_u = arena_alloc<int32_t>(ctx->scratch, n + 1);
    int32_t* ids = arena_alloc<int32_t>(ka, n + 1);
    st.sorted = arena_alloc<float4>(ka, n + 1);
    st.ids = ids;
    st.codes = codes;
    st.codes_w = codes;
    st.codes_u = codes_u;
    st.ids_u = ids_u;
    st.ids_w = ids;
    st.rank = want_rank ? arena_alloc<int32_t>(ka, n + 1) : nullptr;
    st.srad = radii ? arena_alloc<float>(keep_all ? ka : ctx->scratch, n + 1) : nullptr;
    st.radii_src = radii;
    if (!codes_u || !codes || !ids_u || !ids || !st.sorted || (want_rank && !st.rank) || (radii && !st.srad))
        ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    if (n > 0) {
        k_point_codes<<<grid_for(n, BLK), BLK, 0, ctx->stream>>>(*frame, pts, n, codes_u, ids_u, ctx->d_flags);
        ASR_CHECK_LAUNCH(ctx);
        // the cell tables only cover levels up to lsort: sort the top 3*lsort bits of the 63-bit code only
        ASR_TRY((sort_pairs<u64, int32_t>(ctx, ctx->scratch, codes_u, codes, ids_u, ids, n, 63,
                                          3 * (ASR_MAX_LEVEL - std::max(lsort, 1)))));
        k_gather_points<<<grid_for(n, BLK), BLK, 0, ctx->stream>>>(pts, ids, n, st.sorted, st.rank, radii, st.srad);
        ASR_CHECK_LAUNCH(ctx);
    }
    return ASR_HIP_OK;
}
static int build_cell_table(asr_hip_context* ctx, i64 n, int lmin, int lmax, RadiusState& st, bool hash_all,
                            Arena* where = nullptr);
static int build_point_index(asr_hip_context* ctx, const asr_octree_frame* frame, const float* pts,
                             i64 n, int lmin, int lmax, RadiusState& st, Arena* keep = nullptr,
                             bool want_rank = false, bool hash_all = true, const float* radii = nullptr) {
    ASR_TRY(fresh_flags(ctx));
    ASR_TRY(sort_points(ctx, frame, pts, n, lmax, st, keep, want_rank, radii));
    return build_cell_table(ctx, n, lmin, lmax, st, hash_all);
}
static int build_cell_table(asr_hip_context* ctx, i64 n, int lmin, int lmax, RadiusState& st, bool hash_all, Arena* where) {
    Arena& arena = where ? *where : ctx->scratch;
    int host[16];
    const u64* codes = st.codes;
    HashTab dummy{nullptr, nullptr, 0};
    int* level_cnt = ctx->d_flags + 32;  // cells per level
    int host_lvl[ASR_MAX_LEVEL + 1] = {0};
    if (n > 0) {
        ASR_HIP_CHECK(ctx, hipMemsetAsync(level_cnt, 0, (ASR_MAX_LEVEL + 1) * sizeof(int), ctx->stream));
        k_cell_bounds<true><<<std::min<unsigned>(grid_for(n + 1, BLK), 4096u), BLK, 0, ctx->stream>>>(
                codes, n, lmin, lmax, dummy, nullptr, nullptr, ctx->d_flags, level_cnt);
        ASR_CHECK_LAUNCH(ctx);
        ASR_HIP_CHECK(ctx, hipMemcpyAsync(host_lvl, level_cnt, sizeof(host_lvl), hipMemcpyDeviceToHost, ctx->stream));
    }
    ASR_TRY(read_flags(ctx, host));
    if (host[11]) ASR_FAIL(ctx, ASR_HIP_EINVAL, "points contain non-finite values");
    // Levels whose cells hold fewer than two points on average stay out of the hash table (a cloud with very dense
    // spots has query levels at which every point is a cell: 10^7 insertions per level); they are served by binary
    // search (cell_range).  Option search_hash_level forces the finest hashed level (tests).
    int lhash = lmax;
    if (!hash_all) {
        lhash = lmin;
        while (lhash < lmax && (i64)host_lvl[lhash + 1] * 2 <= n) ++lhash;
        if (ctx->opt.search_hash_level >= 0) lhash = (int)std::min<i64>(lmax, std::max<i64>(lmin - 1, ctx->opt.search_hash_level));
    }
    st.lhash = lhash;
    i64 cells = 0;
    for (int l = lmin; l <= lhash; ++l) cells += host_lvl[l];
    // cell_grow: enlarged after an overflow (keys whose low bits are concentrated on few values -- points on a
    // lattice -- crowd a few positions of the buckets, see TabProbe)
    u64 cap = next_pow2((u64)std::max<i64>(1024, 2 * cells)) << st.cell_grow;
    ASR_TRY(make_table(ctx, arena, cap, false, st.tab));
    st.start = arena_alloc<int32_t>(arena, cap);
    st.end = arena_alloc<int32_t>(arena, cap);
    if (!st.start || !st.end) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    if (n > 0) {
        k_cell_bounds<false><<<grid_for(n + 1, BLK), BLK, 0, ctx->stream>>>(
                codes, n, lmin, lhash, st.tab, st.start, st.end, ctx->d_flags, nullptr);
        ASR_CHECK_LAUNCH(ctx);
    }
    return ASR_HIP_OK;
}

constexpr int PRESORT_LEVEL = 13;  // the least depth of the early sort (five 8-bit passes either way)
// finest level a point can be inserted on (point_key: level_from_scale of its radius, capped at max_depth) = an upper bound
// of the octree's finest leaf level, before the octree exists
__global__ __launch_bounds__(256) void k_finest_point_level(asr_octree_frame f, const float* __restrict__ radii, i64 n,
                                                            float radius_scale, int max_depth, int* __restrict__ out) {
    int m = 0;
    for (i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        int l = level_from_scale(f, radius_scale * radii[i]);
        l = l < max_depth ? l : max_depth;
        m = max(m, l);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(out, m);
}
// radius_scale > 0: the sort goes one level below the finest level a point can be inserted on (capped at the deepest), so that
// the aggregation search of a cloud with very dense spots (leaves finer than level 12: BASELINE's mixed-density config) finds
// its cells in THIS order too instead of sorting the points again after the octree (3 ms at 10 M points)
int asr_geom_presort(asr_hip_context* ctx, Arena& keep, const asr_octree_frame* frame, const float* pts,
                     const float* radii, i64 n, float radius_scale, int max_depth) {
    ASR_TRY(ensure_flags(ctx));
    AsrPointIndex& pi = ctx->pindex;
    pi.valid = false;
    if (n <= 0 || n >= (i64(1) << 31)) return ASR_HIP_OK;
    int lsort = PRESORT_LEVEL;
    if (radius_scale > 0.f && radii) {
        int host[16];
        ASR_TRY(fresh_flags(ctx));
        k_finest_point_level<<<std::min<unsigned>(grid_for(n, BLK), 2048u), BLK, 0, ctx->stream>>>(
                *frame, radii, n, radius_scale, std::min(max_depth, ASR_MAX_LEVEL), ctx->d_flags + 8);
        ASR_CHECK_LAUNCH(ctx);
        ASR_TRY(read_flags(ctx, host));
        lsort = std::max(PRESORT_LEVEL, std::min(ASR_MAX_LEVEL, host[8] + 1));
    }
    RadiusState st;
    ASR_TRY(sort_points(ctx, frame, pts, n, lsort, st, &keep, true, radii, true));
    pi.pts = pts;
    pi.radii = radii;
    pi.n = n;
    pi.lsort = lsort;
    pi.sorted = st.sorted;
    pi.ids = st.ids;
    pi.rank = st.rank;
    pi.srad = st.srad;
    pi.codes = st.codes_w;
    pi.valid = true;
    return ASR_HIP_OK;
}

// The aggregation search's cell table needs nothing from the octree either, only the range of levels its leaves can
// have: [1, PRESORT_LEVEL] covers every octree whose leaves are not finer than the sort (a root-only octree or a deeper
// one makes asr_geom_radius_count build its own table).  Host round trips: call it from the search's own thread.
int asr_geom_precells(asr_hip_context* ctx, Arena& keep) {
    AsrPointIndex& pi = ctx->pindex;
    pi.tab_keys = nullptr;
    if (!pi.valid || pi.n <= 0) return ASR_HIP_OK;
    RadiusState st;
    st.codes = st.codes_w = pi.codes;
    st.n = pi.n;
    st.cell_grow = rstate(ctx).cell_grow;
    ASR_TRY(fresh_flags(ctx));
    ASR_TRY(build_cell_table(ctx, pi.n, 1, pi.lsort, st, false, &keep));
    int host[16];
    ASR_TRY(read_flags(ctx, host));
    if (host[1]) return ASR_HIP_OK;  // overflow: the query builds (and grows) its own table
    pi.tab_grow = st.cell_grow;
    pi.tab_keys = st.tab.keys;
    pi.tab_mask = st.tab.mask;
    pi.tab_start = st.start;
    pi.tab_end = st.end;
    pi.tab_lmin = 1;
    pi.tab_lmax = pi.lsort;
    pi.tab_lhash = st.lhash;
    return ASR_HIP_OK;
}

static int query_level_range(asr_hip_context* ctx, const asr_octree_frame* frame, const float* sizes,
                             i64 v, int* lmin, int* lmax) {
    int host[16];
    ASR_TRY(fresh_flags(ctx));
    int init[2] = {ASR_MAX_LEVEL, 0};
    ASR_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_flags + 6, init, 2 * sizeof(int),
                                      hipMemcpyHostToDevice, ctx->stream));
    k_query_levels<<<std::min<unsigned>(grid_for(v, BLK), 2048u), BLK, 0, ctx->stream>>>(*frame, sizes, v, ctx->d_flags);
    ASR_CHECK_LAUNCH(ctx);
    ASR_TRY(read_flags(ctx, host));
    *lmin = host[6];
    *lmax = host[7];
    return ASR_HIP_OK;
}

int asr_geom_radius_count(asr_hip_context* ctx, const asr_octree_frame* frame, const float* pts,
                          i64 n, const float* centers, const float* sizes, i64 v, i64* rs,
                          i64* num_pairs, Arena* keep, const float* radii, const u64* voxel_keys, int lmin_hint,
                          int lmax_hint, const AsrPointIndex* pre) {
    ASR_TRY(ensure_flags(ctx));
    RadiusState& st = rstate(ctx);
    st.valid = false;
    if (v <= 0) {
        *num_pairs = 0;
        if (rs) ASR_HIP_CHECK(ctx, hipMemsetAsync(rs, 0, sizeof(i64), ctx->stream));
        return ASR_HIP_OK;
    }
    if (n >= (i64(1) << 31)) ASR_FAIL(ctx, ASR_HIP_EINVAL, "too many points for int32 indices");
    int host[16];
    st.v = v;
    int lmin = lmin_hint, lmax = lmax_hint;
    if (lmin < 0 || lmax < lmin) ASR_TRY(query_level_range(ctx, frame, sizes, v, &lmin, &lmax));
    // aligned queries (voxel_keys): half-size cells of level L + 1 for the voxel levels L <= lhalf_max (see AlignedQ)
    // search_half 1 (default): half-size cells for the HEAVY rows only -- they walk 3.4x fewer candidates (0.90 -> 0.21 ms
    // at 10 M points) -- while the per-voxel kernel keeps the 27 full-size cells: its 64 look-ups with half-size cells
    // cost more (2.49 vs 1.23 ms) than the shorter candidate walk saves (0.45 vs 1.08 ms).  2: half-size cells everywhere.
    st.aligned = voxel_keys != nullptr && ctx->opt.search_half != 0;
    st.aligned_main = st.aligned && ctx->opt.search_half >= 2;
    ExtraParams ep;
    memset(&ep, 0, sizeof(ep));
    int lhalf_max = -1;
    if (st.aligned) {
        double M = 1;
        for (int d = 0; d < 3; ++d)
            M = std::max(M, std::max(std::fabs((double)frame->offset[d]), std::fabs((double)(1 << ASR_MAX_LEVEL) - frame->offset[d])));
        const int T = 2 + (int)std::floor(3.0 * (M + 1) / 16777216.0);
        lhalf_max = ASR_MAX_LEVEL - 1;
        while (lhalf_max >= 0 && (1 << (ASR_MAX_LEVEL - 1 - lhalf_max)) < 16 * T) --lhalf_max;
        ep.T = T;
        ep.lmin = lmin;
        ep.lmax = std::min(lmax, lhalf_max);
        for (int l = 0; l <= ASR_MAX_LEVEL; ++l)
            ep.rho[l] = (int)std::ceil(std::sqrt(2.0 * (double)(1 << (ASR_MAX_LEVEL - l)) * T)) + 2 * T + 1;
    }
    const int ltab = st.aligned ? std::min(ASR_MAX_LEVEL, std::min(lmax, lhalf_max) + 1) : lmax;
    if (pre && pre->valid && pre->pts == pts && pre->n == n && pre->lsort >= std::max(lmax, ltab) && (!radii || pre->srad)) {
        st.frame = *frame;  // the points are already in Morton order (asr_geom_presort)
        st.n = n;
        st.lsort = pre->lsort;
        st.sorted = pre->sorted;
        st.ids = pre->ids;
        st.rank = pre->rank;
        st.srad = radii ? pre->srad : nullptr;
        st.radii_src = radii;
        st.codes = st.codes_w = pre->codes;
        st.codes_u = nullptr;
        st.ids_u = nullptr;
        st.ids_w = pre->ids;
        ASR_TRY(fresh_flags(ctx));
        if (pre->tab_keys && pre->tab_grow == st.cell_grow && pre->tab_lmin <= lmin && pre->tab_lmax >= std::max(lmax, ltab)) {
            st.tab = HashTab{pre->tab_keys, nullptr, pre->tab_mask};  // built beside the octree (asr_geom_precells)
            st.start = pre->tab_start;
            st.end = pre->tab_end;
            st.lhash = pre->tab_lhash;
        } else {
            ASR_TRY(build_cell_table(ctx, n, lmin, std::max(lmax, ltab), st, false));
        }
    } else {
        ASR_TRY(build_point_index(ctx, frame, pts, n, lmin, std::max(lmax, ltab), st, keep, true, false, radii));
    }
    i64* counts = arena_alloc<i64>(ctx->scratch, v + 1);
    st.tmp = arena_alloc<u64>(ctx->scratch, (size_t)v * RADIUS_LIGHT);
    st.heavy = arena_alloc<int32_t>(ctx->scratch, v);
    st.is_heavy = arena_alloc<uint8_t>(ctx->scratch, v);
    int2* extras = arena_alloc<int2>(ctx->scratch, EXTRA_CAP);
    if (!counts || !st.tmp || !st.heavy || !st.is_heavy || !extras) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    ASR_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_flags + 10, 0, sizeof(int), ctx->stream));
    ASR_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_flags + 15, 0, sizeof(int), ctx->stream));
    st.aq = AlignedQ{voxel_keys, extras, ctx->d_flags + 15, lhalf_max};
    // one pass: counts, the sorted light rows (fixed slots) and the list of heavy rows
    if (st.aligned && ep.lmax >= ep.lmin && n > 0) {
        k_radius_extras<<<grid_for(n, BLK), BLK, 0, ctx->stream>>>(st.codes, st.sorted, n, voxel_keys, v, centers, sizes,
                                                                  ep, extras, ctx->d_flags + 15);
        ASR_CHECK_LAUNCH(ctx);
    }
    const unsigned qgrid = grid_for(v + 1, 4);
    if (st.aligned_main)
        k_radius_query<2, true><<<qgrid, BLK, 0, ctx->stream>>>(*frame, st.sorted, centers, sizes, v, st.index(), st.aq, counts,
                                                                st.tmp, st.heavy, ctx->d_flags + 10, st.is_heavy);
    else
        k_radius_query<2, false><<<qgrid, BLK, 0, ctx->stream>>>(*frame, st.sorted, centers, sizes, v, st.index(), st.aq, counts,
                                                                 st.tmp, st.heavy, ctx->d_flags + 10, st.is_heavy);
    ASR_CHECK_LAUNCH(ctx);
    ASR_TRY(read_flags(ctx, host));
    if (host[1]) {  // retry with a table four times the size
        if (st.cell_grow >= 6) ASR_FAIL(ctx, ASR_HIP_ELOGIC, "radius search cell table overflow");
        st.cell_grow += 2;
        return asr_geom_radius_count(ctx, frame, pts, n, centers, sizes, v, rs, num_pairs, keep, radii, voxel_keys,
                                     lmin_hint, lmax_hint, pre);
    }
    if (host[15] > EXTRA_CAP) ASR_FAIL(ctx, ASR_HIP_ELOGIC, "radius search: %d pairs in the rounding margin (cap %d)", host[15], EXTRA_CAP);
    ctx->search_extras = host[15];
    st.num_heavy = host[10];
    if (st.num_heavy > 0) {
        st.hc_pref = arena_alloc<int>(ctx->scratch, (size_t)st.num_heavy * HC_LD);
        st.hc_beg = arena_alloc<int>(ctx->scratch, (size_t)st.num_heavy * HC_LD);
        if (!st.hc_pref || !st.hc_beg) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
        const unsigned cgrid = grid_for(st.num_heavy, 4);
        if (st.aligned) {
            k_radius_heavy_cells<true><<<cgrid, BLK, 0, ctx->stream>>>(*frame, centers, sizes, st.heavy, (int)st.num_heavy,
                                                                       st.index(), st.aq, st.hc_pref, st.hc_beg);
            k_radius_heavy<false, true><<<dim3((unsigned)st.num_heavy, RADIUS_SPLIT), BLK, 0, ctx->stream>>>(
                    *frame, st.sorted, centers, sizes, st.heavy, st.index(), st.aq, counts, nullptr, nullptr, nullptr, nullptr,
                    st.hc_pref, st.hc_beg, 0);
        } else {
            k_radius_heavy_cells<false><<<cgrid, BLK, 0, ctx->stream>>>(*frame, centers, sizes, st.heavy, (int)st.num_heavy,
                                                                        st.index(), st.aq, st.hc_pref, st.hc_beg);
            k_radius_heavy<false, false><<<dim3((unsigned)st.num_heavy, RADIUS_SPLIT), BLK, 0, ctx->stream>>>(
                    *frame, st.sorted, centers, sizes, st.heavy, st.index(), st.aq, counts, nullptr, nullptr, nullptr, nullptr,
                    st.hc_pref, st.hc_beg, 0);
        }
        ASR_CHECK_LAUNCH(ctx);
    }
    ASR_TRY(scan_counts(ctx, ctx->scratch, counts, rs, v + 1));
    ASR_TRY(read_i64(ctx, rs + v, num_pairs));
    st.valid = true;
    return ASR_HIP_OK;
}

// KDTree::ComputeRadiusNeighbors (cpp/lib/nsearch.cpp:88-105): per point the number of points with
// squared distance < radius_i^2 (the point itself included)
int asr_geom_radius_neighbor_count(asr_hip_context* ctx, const asr_octree_frame* frame, const float* pts,
                                   const float* radii, i64 n, i64* counts_out) {
    ASR_TRY(ensure_flags(ctx));
    if (n <= 0) return ASR_HIP_OK;
    if (n >= (i64(1) << 31)) ASR_FAIL(ctx, ASR_HIP_EINVAL, "too many points for int32 indices");
    RadiusState st;
    int host[16], lmin, lmax;
    ASR_TRY(query_level_range(ctx, frame, radii, n, &lmin, &lmax));
    ASR_TRY(build_point_index(ctx, frame, pts, n, lmin, lmax, st, nullptr, false, false));
    i64* counts = arena_alloc<i64>(ctx->scratch, n + 1);
    if (!counts) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    k_radius_query<0, false><<<grid_for(n + 1, 4), BLK, 0, ctx->stream>>>(
            *frame, st.sorted, pts, radii, n, st.index(), st.aq, counts, nullptr, nullptr, nullptr, nullptr);
    ASR_CHECK_LAUNCH(ctx);
    ASR_HIP_CHECK(ctx, hipMemcpyAsync(counts_out, counts, n * sizeof(i64), hipMemcpyDeviceToDevice,
                                      ctx->stream));
    ASR_TRY(read_flags(ctx, host));
    if (host[1]) ASR_FAIL(ctx, ASR_HIP_ELOGIC, "cell table overflow");
    return ASR_HIP_OK;
}

// Re-sorts the points on the code bits down to level `ldeep` (they were sorted down to the table's finest level): the
// cell ranges of the table stay what they are (a stable sort on more bits only reorders points inside those cells), and
// cells of finer levels become contiguous, i.e. searchable (cell_range).
static int deepen_point_index(asr_hip_context* ctx, const float* pts, RadiusState& st, int ldeep) {
    ASR_TRY((sort_pairs<u64, int32_t>(ctx, ctx->scratch, st.codes_u, st.codes_w, st.ids_u, st.ids_w, st.n, 63,
                                      3 * (ASR_MAX_LEVEL - ldeep))));
    k_gather_points<<<grid_for(st.n, BLK), BLK, 0, ctx->stream>>>(pts, st.ids_w, st.n, st.sorted, st.rank);
    ASR_CHECK_LAUNCH(ctx);
    return ASR_HIP_OK;
}

// KDTree::ComputeKRadius / ComputeInlier (cpp/lib/nsearch.cpp:30-86)
int asr_geom_knn(asr_hip_context* ctx, const asr_octree_frame* frame, const float* pts, i64 n, int k,
                 const float* radii_in, float radius_fraction, int outlier_threshold, float* radii_out,
                 uint8_t* inlier_out) {
    ASR_TRY(ensure_flags(ctx));
    if (n <= 0) return ASR_HIP_OK;
    if (k < 1) ASR_FAIL(ctx, ASR_HIP_EINVAL, "knn: k must be >= 1");
    if (n >= (i64(1) << 31)) ASR_FAIL(ctx, ASR_HIP_EINVAL, "too many points for int32 indices");
    // finest level: about one point per cell if the cloud filled the cube
    int lfine = 1;
    while (lfine < ASR_MAX_LEVEL - 1 && (i64(1) << (3 * lfine)) < n) ++lfine;
    RadiusState st;
    int host[16];
    ASR_TRY(build_point_index(ctx, frame, pts, n, 0, lfine, st));
    st.lhash = lfine;
    // dense spots: a finest-level cell with thousands of points makes every one of them walk thousands of candidates.
    // Such clouds get their points sorted eight levels deeper, and the points of crowded cells start on a finer level.
    int ldeep = lfine;
    {
        ASR_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_flags + 14, 0, sizeof(int), ctx->stream));
        k_max_cell_pop<<<grid_for((i64)st.tab.mask + 1, BLK), BLK, 0, ctx->stream>>>(st.tab, st.start, st.end, lfine,
                                                                                   ctx->d_flags + 14);
        ASR_CHECK_LAUNCH(ctx);
        ASR_TRY(read_flags(ctx, host));
        if (host[14] > KNN_CROWD && ctx->opt.knn_deep) {
            ldeep = std::min<int>(ASR_MAX_LEVEL, lfine + 8);
            ASR_TRY(deepen_point_index(ctx, pts, st, ldeep));
        }
    }
    const bool fast = !inlier_out && radii_out && k <= 32 && n >= k && ctx->opt.knn_cells;
    if (fast) {
        // cell-parallel pass, then the wave-per-point kernel for what it could not certify
        int32_t* cells = arena_alloc<int32_t>(ctx->scratch, n + 1);
        int32_t* fallback = arena_alloc<int32_t>(ctx->scratch, n + 1);
        if (!cells || !fallback) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
        ASR_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_flags + 12, 0, 2 * sizeof(int), ctx->stream));
        k_cell_list<<<grid_for(n, 4096), BLK, 0, ctx->stream>>>(st.codes, n, lfine, cells, ctx->d_flags + 12);
        ASR_CHECK_LAUNCH(ctx);
        ASR_TRY(read_flags(ctx, host));
        const i64 ncells = host[12];
#define ASR_KNN_CELLS(K_)                                                                                           \
    k_knn_cells<K_><<<grid_for(ncells, 4), BLK, 0, ctx->stream>>>(*frame, st.sorted, n, st.tab, st.start, st.end, lfine, \
                                                                   k, cells, ncells, radii_out, fallback,            \
                                                                   ctx->d_flags + 13)
        if (k <= 8)
            ASR_KNN_CELLS(8);
        else if (k <= 16)
            ASR_KNN_CELLS(16);
        else if (k <= 24)
            ASR_KNN_CELLS(24);
        else
            ASR_KNN_CELLS(32);
#undef ASR_KNN_CELLS
        ASR_CHECK_LAUNCH(ctx);
        ASR_TRY(read_flags(ctx, host));
        const i64 nfb = host[13];
        if (nfb > 0) {
            k_knn<<<grid_for(nfb, 4), BLK, 0, ctx->stream>>>(*frame, st.sorted, n, st.index(), lfine, ldeep, k, radii_in,
                                                             radius_fraction, outlier_threshold, radii_out, nullptr,
                                                             fallback, nfb);
            ASR_CHECK_LAUNCH(ctx);
        }
    } else {
        k_knn<<<grid_for(n, 4), BLK, 0, ctx->stream>>>(*frame, st.sorted, n, st.index(), lfine, ldeep, k, radii_in,
                                                       radius_fraction, outlier_threshold, radii_out, inlier_out,
                                                       nullptr, 0);
        ASR_CHECK_LAUNCH(ctx);
    }
    ASR_TRY(read_flags(ctx, host));
    if (host[1]) ASR_FAIL(ctx, ASR_HIP_ELOGIC, "cell table overflow");
    return ASR_HIP_OK;
}

int asr_geom_radius_fill(asr_hip_context* ctx, const float* pts, const float* radii, i64 n,
                         const float* centers, const float* sizes, i64 v, const i64* rs,
                         int32_t* idx, float* dist, float* compat, int32_t* spos, const float4** sorted_out) {
    RadiusState& st = rstate(ctx);
    (void)pts;
    if (sorted_out) *sorted_out = nullptr;
    if (v <= 0) return ASR_HIP_OK;
    if (!st.valid || st.n != n || st.v != v)
        ASR_FAIL(ctx, ASR_HIP_EINVAL,
                 "asr_hip_multi_radius_search_fill must follow the matching _count call");
    if (sorted_out) *sorted_out = st.sorted;
    float* srad = st.srad;  // gathered with the points when the count call had the radii
    if (compat && !srad) {
        srad = arena_alloc<float>(ctx->scratch, n + 1);
        if (!srad) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
        if (n > 0) {
            k_gather_radii<<<grid_for(n, BLK), BLK, 0, ctx->stream>>>(radii, st.ids, n, srad);
            ASR_CHECK_LAUNCH(ctx);
        }
    }
    // light rows: already sorted in their fixed slots, copy them to the CSR positions.  (Round 4: on a side stream beside the
    // heavy rows' fill / sort / unpack below -- measured, no gain: the build's two chains already fill the GPU, what one
    // kernel gains the ones beside it lose.)
    k_radius_place<<<grid_for(v * 16, BLK), BLK, 0, ctx->stream>>>(st.tmp, rs, st.is_heavy, v, sizes, st.ids, srad, idx,
                                                                   spos, dist, compat);
    ASR_CHECK_LAUNCH(ctx);
    const i64 nh = st.num_heavy;
    if (nh > 0) {
        // heavy rows (> RADIUS_LIGHT hits): gather keys per row, segmented sort by (squared distance,
        // index) -- distances are >= 0 so their float bits order like unsigned integers
        i64* hcnt = arena_alloc<i64>(ctx->scratch, nh + 1);
        i64* hoff = arena_alloc<i64>(ctx->scratch, nh + 1);
        if (!hcnt || !hoff) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
        k_heavy_counts<<<grid_for(nh + 1, BLK), BLK, 0, ctx->stream>>>(st.heavy, nh, rs, hcnt);
        ASR_CHECK_LAUNCH(ctx);
        ASR_TRY(scan_counts(ctx, ctx->scratch, hcnt, hoff, nh + 1));
        i64 hp = 0;
        ASR_TRY(read_i64(ctx, hoff + nh, &hp));
        if (hp >= (i64(1) << 32)) ASR_FAIL(ctx, ASR_HIP_EINVAL, "too many aggregation pairs");
        // One key per hit when (row, 31 distance bits, index) fits 64 bits: a single keys-only radix sort over exactly those
        // bits orders all heavy rows by (squared distance, index) -- distances are >= 0, their float bits order like
        // unsigned integers.  Otherwise two stable sorts: by the (distance, index) key, then by the row (a segmented sort
        // spends 1.6 ms on the few 10^4-entry rows).
        const int row_bits = bits_for(nh + 1);
        int idx_bits = bits_for(std::max<i64>(n, 2));
        if (row_bits + 31 + idx_bits > 64) idx_bits = 0;
        u64* k_u = arena_alloc<u64>(ctx->scratch, hp);
        u64* k_s = arena_alloc<u64>(ctx->scratch, hp);
        int32_t* t_row = idx_bits ? nullptr : arena_alloc<int32_t>(ctx->scratch, hp);
        if (!k_u || !k_s || (!idx_bits && !t_row)) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
        int* cursor = arena_alloc<int>(ctx->scratch, nh);
        if (!cursor) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
        ASR_HIP_CHECK(ctx, hipMemsetAsync(cursor, 0, (size_t)nh * sizeof(int), ctx->stream));
        if (st.aligned)
            k_radius_heavy<true, true><<<dim3((unsigned)nh, RADIUS_SPLIT), BLK, 0, ctx->stream>>>(
                    st.frame, st.sorted, centers, sizes, st.heavy, st.index(), st.aq, nullptr, hoff, cursor, k_u, t_row,
                    st.hc_pref, st.hc_beg, idx_bits);
        else
            k_radius_heavy<true, false><<<dim3((unsigned)nh, RADIUS_SPLIT), BLK, 0, ctx->stream>>>(
                    st.frame, st.sorted, centers, sizes, st.heavy, st.index(), st.aq, nullptr, hoff, cursor, k_u, t_row,
                    st.hc_pref, st.hc_beg, idx_bits);
        ASR_CHECK_LAUNCH(ctx);
        if (idx_bits) {
            ASR_TRY(sort_keys(ctx, ctx->scratch, k_u, k_s, hp, row_bits + 31 + idx_bits));
        } else {
            int32_t* t_row_s = arena_alloc<int32_t>(ctx->scratch, hp);
            if (!t_row_s) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
            ASR_TRY((sort_pairs<u64, int32_t>(ctx, ctx->scratch, k_u, k_s, t_row, t_row_s, hp, 64)));
            ASR_TRY((sort_pairs<int32_t, u64>(ctx, ctx->scratch, t_row_s, t_row, k_s, k_u, hp, bits_for(nh + 1))));
            k_s = k_u;  // sorted keys; t_row holds the sorted rows
        }
        k_radius_unpack_heavy<<<grid_for(hp, BLK), BLK, 0, ctx->stream>>>(k_s, t_row, hp, st.heavy, hoff, rs, sizes,
                                                                        radii, st.rank, idx, spos, dist, compat, idx_bits);
        ASR_CHECK_LAUNCH(ctx);
    }
    st.valid = false;
    return ASR_HIP_OK;
}

// Dual cells of the LAST octree build (ctx->nodes / ctx->leaves).  count: number of cells; fill:
// [D,8] leaf indices, leaf order x corner order.
int asr_geom_dual_count(asr_hip_context* ctx, const u64* nodes, i64 nn, const u64* leaves, i64 nl, i64* num_cells) {
    ASR_TRY(ensure_flags(ctx));
    *num_cells = 0;
    RadiusState& st0 = rstate(ctx);
    st0.dual_leaves = leaves;
    st0.dual_nl = nl;
    if (nl <= 0) return ASR_HIP_OK;
    ctx->scratch.reset();
    HashTab t;
    u64 cap = next_pow2((u64)std::max<i64>(1024, 2 * nn));
    ASR_TRY(make_table(ctx, ctx->scratch, cap, true, t));
    ASR_HIP_CHECK(ctx, hipMemsetAsync(t.vals, 0xFF, cap * sizeof(int32_t), ctx->stream));  // -1 = inner
    ASR_TRY(fresh_flags(ctx));
    int32_t* keep = t.vals;
    t.vals = arena_alloc<int32_t>(ctx->scratch, cap);  // throw-away values for the node pass
    if (!t.vals) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    k_map_build<<<grid_for(nn, BLK), BLK, 0, ctx->stream>>>(nodes, nn, t, ctx->d_flags);
    ASR_CHECK_LAUNCH(ctx);
    t.vals = keep;
    k_map_build<<<grid_for(nl, BLK), BLK, 0, ctx->stream>>>(leaves, nl, t, ctx->d_flags);
    ASR_CHECK_LAUNCH(ctx);
    i64* counts = arena_alloc<i64>(ctx->scratch, nl + 1);
    i64* offsets = arena_alloc<i64>(ctx->scratch, nl + 1);
    if (!counts || !offsets) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    k_dual_count<<<grid_for(nl + 1, BLK), BLK, 0, ctx->stream>>>(leaves, nl, t, counts);
    ASR_CHECK_LAUNCH(ctx);
    ASR_TRY(scan_counts(ctx, ctx->scratch, counts, offsets, nl + 1));
    ASR_TRY(read_i64(ctx, offsets + nl, num_cells));
    RadiusState& st = rstate(ctx);  // park the table + offsets for the fill call
    st.valid = false;
    st.tab = t;
    st.start = (int32_t*)offsets;
    st.v = -nl;  // marks "dual state"
    return ASR_HIP_OK;
}
int asr_geom_dual_fill(asr_hip_context* ctx, i64* out) {
    RadiusState& st = rstate(ctx);
    const i64 nl = st.dual_nl;
    if (nl <= 0) return ASR_HIP_OK;
    if (st.v != -nl) ASR_FAIL(ctx, ASR_HIP_EINVAL, "dual_fill must follow the matching dual_count call");
    int host[16];
    k_dual_fill<<<grid_for(nl, BLK), BLK, 0, ctx->stream>>>(st.dual_leaves, nl, st.tab, (const i64*)st.start, out,
                                                            ctx->d_flags);
    ASR_CHECK_LAUNCH(ctx);
    ASR_TRY(read_flags(ctx, host));
    st.v = 0;
    if (host[1]) ASR_FAIL(ctx, ASR_HIP_ELOGIC, "invalid key after searching for node (cpp/lib/grid.cpp:435-440)");
    return ASR_HIP_OK;
}

int asr_geom_invert(asr_hip_context* ctx, i64 num_points, const int32_t* idx, const i64* rs,
                    i64 num_rows, const uint8_t* attr, int32_t* out_idx, i64* out_rs,
                    uint8_t* out_attr, i64 known_pairs) {
    i64 p = known_pairs;  // >= 0: the caller knows rs[num_rows] (no read-back)
    if (p < 0) {
        p = 0;
        if (num_rows > 0) ASR_TRY(read_i64(ctx, rs + num_rows, &p));
    }
    i64* counts = arena_alloc<i64>(ctx->scratch, num_points + 1);
    if (!counts) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    ASR_HIP_CHECK(ctx, hipMemsetAsync(counts, 0, (num_points + 1) * sizeof(i64), ctx->stream));
    if (p > 0) {
        k_invert_hist<<<grid_for(p, BLK), BLK, 0, ctx->stream>>>(idx, p, counts, num_points);
        ASR_CHECK_LAUNCH(ctx);
    }
    ASR_TRY(scan_counts(ctx, ctx->scratch, counts, out_rs, num_points + 1));
    if (p > 0) {
        int32_t* src = arena_alloc<int32_t>(ctx->scratch, p);
        int32_t* src_s = arena_alloc<int32_t>(ctx->scratch, p);
        int32_t* key_s = arena_alloc<int32_t>(ctx->scratch, p);
        if (!src || !src_s || !key_s) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
        k_iota32<<<grid_for(p, BLK), BLK, 0, ctx->stream>>>(src, p);
        ASR_CHECK_LAUNCH(ctx);
        // stable LSD radix sort by target row == "fill in query order" 
        ASR_TRY((sort_pairs<int32_t, int32_t>(ctx, ctx->scratch, idx, key_s, src, src_s, p,
                                              bits_for(num_points + 1))));
        k_invert_fill<<<grid_for(p, BLK), BLK, 0, ctx->stream>>>(src_s, p, rs, num_rows, attr,
                                                                out_idx, out_attr);
        ASR_CHECK_LAUNCH(ctx);
    }
    return ASR_HIP_OK;
}

// ==========================================================================================
// row-group plans for the plan-driven sparse conv (layout: asr_common.h, consumer: asr_conv16.hip)
// ==========================================================================================
namespace {
constexpr i64 PLAN_DUP_FLAG = i64(1) << 40;
__global__ void k_plan_masks(const uint8_t* __restrict__ kidx, const i64* __restrict__ rs,
                             const int32_t* __restrict__ perm, i64 num_out, int K, i64 groups_pad,
                             uint4* __restrict__ hdr, i64* __restrict__ counts) {
    const i64 row = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    unsigned long long m = 0;
    int dup = 0;  // a slot that occurs twice in a row: the plan holds ONE neighbour per (row, slot)
    if (row < num_out) {
        const i64 q = perm ? perm[row] : row;
        int n = 0;
        for (i64 p = rs[q], pe = rs[q + 1]; p < pe; ++p) {
            const int k = kidx[p];
            if (k < K) {
                m |= 1ull << k;
                ++n;
            }
        }
        dup = n != __popcll(m);
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
        m |= __shfl_xor(m, o, 64);
        dup |= __shfl_xor(dup, o, 64);
    }
    const i64 grp = row >> 4;
    if ((threadIdx.x & 15) == 0 && grp < groups_pad) {
        hdr[grp] = make_uint4((unsigned)m, (unsigned)(m >> 32), 0u, 0u);
        counts[grp] = (i64)__popcll(m) + (dup ? PLAN_DUP_FLAG : 0);  // the host sees the flag in the block total
    }
}

__global__ void k_plan_fill(const int32_t* __restrict__ nidx, const uint8_t* __restrict__ kidx,
                            const i64* __restrict__ rs, const int32_t* __restrict__ perm, i64 num_out, int K,
                            i64 groups_pad, const i64* __restrict__ offs, uint4* __restrict__ hdr, int32_t* __restrict__ pool) {
    const i64 row = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    const i64 grp = row >> 4;
    if (grp >= groups_pad) return;
    const unsigned off = (unsigned)offs[grp];
    if ((row & 15) == 0) hdr[grp].z = off;
    if (row >= num_out) return;
    const uint2 h = *reinterpret_cast<const uint2*>(&hdr[grp]);
    const unsigned long long m = (unsigned long long)h.x | ((unsigned long long)h.y << 32);
    const i64 q = perm ? perm[row] : row;
    for (i64 p = rs[q], pe = rs[q + 1]; p < pe; ++p) {
        const int k = kidx[p];
        if (k >= K) continue;
        const int j = __popcll(m & ((1ull << k) - 1));
        pool[((i64)off + j) * 16 + (row & 15)] = nidx[p];
    }
}


}  // namespace

int asr_geom_conv_plan_count(asr_hip_context* ctx, Arena& keep, const int32_t* nidx, const uint8_t* kidx, const i64* rs,
                             const int32_t* perm, i64 num_out, int K, asr_conv_plan* plan) {
    *plan = asr_conv_plan();
    plan->nidx = nidx;
    plan->kidx = kidx;
    plan->rs = rs;
    plan->perm = perm;
    plan->num_out = num_out;
    plan->K = K;
    if (num_out <= 0) return ASR_HIP_OK;
    plan->groups = (num_out + 15) / 16;
    plan->groups_pad = (plan->groups + 15) / 16 * 16;  // whole 256-row tiles
    plan->hdr = (uint4*)keep.alloc(plan->groups_pad * sizeof(uint4));
    plan->offs = (i64*)keep.alloc((plan->groups_pad + 1) * sizeof(i64));
    i64* counts = (i64*)ctx->scratch.alloc((plan->groups_pad + 1) * sizeof(i64));
    if (!plan->hdr || !plan->offs || !counts) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    ASR_HIP_CHECK(ctx, hipMemsetAsync(counts + plan->groups_pad, 0, sizeof(i64), ctx->stream));
    k_plan_masks<<<grid_for(plan->groups_pad * 16, BLK), BLK, 0, ctx->stream>>>(kidx, rs, perm, num_out, K,
                                                                              plan->groups_pad, plan->hdr, counts);
    ASR_CHECK_LAUNCH(ctx);
    return asr_prim::scan_counts(ctx, ctx->scratch, counts, plan->offs, plan->groups_pad + 1);
}

int asr_geom_conv_plan_fill(asr_hip_context* ctx, Arena& keep, asr_conv_plan* plan, i64 blocks) {
    if (plan->num_out <= 0) return ASR_HIP_OK;
    plan->blocks = blocks;
    plan->pool = (int32_t*)keep.alloc((size_t)(blocks > 0 ? blocks : 1) * 64);
    if (!plan->pool) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    ASR_HIP_CHECK(ctx, hipMemsetAsync(plan->pool, 0xFF, (size_t)(blocks > 0 ? blocks : 1) * 64, ctx->stream));
    k_plan_fill<<<grid_for(plan->groups_pad * 16, BLK), BLK, 0, ctx->stream>>>(
            plan->nidx, plan->kidx, plan->rs, plan->perm, plan->num_out, plan->K, plan->groups_pad, plan->offs,
            plan->hdr, plan->pool);
    ASR_CHECK_LAUNCH(ctx);
    return ASR_HIP_OK;
}

// All lists of a build in one pass: one slot-set kernel, one scan, one read-back, one fill over the concatenated
// row ranges.  The plans share one pool (their headers hold positions in it).
namespace {
constexpr int PLAN_MAX_JOBS = 16;
struct PlanBatch {
    int n;
    i64 base[PLAN_MAX_JOBS + 1];  // first (padded) row of each job, multiples of 256; base[n] = total
    i64 rows[PLAN_MAX_JOBS];
    int K[PLAN_MAX_JOBS];
    const int32_t* nidx[PLAN_MAX_JOBS];
    const uint8_t* kidx[PLAN_MAX_JOBS];
    const i64* rs[PLAN_MAX_JOBS];
    const int32_t* perm[PLAN_MAX_JOBS];
};
__device__ inline int plan_job_of(const PlanBatch& b, i64 row) {
    int j = 0;
    while (j + 1 < b.n && row >= b.base[j + 1]) ++j;
    return j;
}
__global__ void k_plan_masks_batch(PlanBatch b, uint4* __restrict__ hdr, i64* __restrict__ counts) {
    const i64 grow = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (grow >= b.base[b.n]) return;
    const int j = plan_job_of(b, grow);
    const i64 row = grow - b.base[j];
    unsigned long long m = 0;
    if (row < b.rows[j]) {
        const i64 q = b.perm[j] ? b.perm[j][row] : row;
        const uint8_t* kidx = b.kidx[j];
        for (i64 p = b.rs[j][q], pe = b.rs[j][q + 1]; p < pe; ++p) {
            const int k = kidx[p];
            if (k < b.K[j]) m |= 1ull << k;
        }
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) m |= __shfl_xor(m, o, 64);
    if ((threadIdx.x & 15) == 0) {
        hdr[grow >> 4] = make_uint4((unsigned)m, (unsigned)(m >> 32), 0u, 0u);
        counts[grow >> 4] = (i64)__popcll(m);
    }
}
// Thread r of a 16-thread group owns column r (its row) of every 64-byte line of the group: it walks the group's slot set
// and its own row together (rows list their slots in ascending order, cpp/lib/grid.cpp:99-170,229-240) and writes the
// neighbour or -1, so the 16 threads complete one line per store instruction and no line needs an initialising memset.
__global__ void k_plan_fill_batch(PlanBatch b, const i64* __restrict__ offs, uint4* __restrict__ hdr,
                                  int32_t* __restrict__ pool) {
    const i64 grow = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (grow >= b.base[b.n]) return;
    const i64 grp = grow >> 4;
    const unsigned off = (unsigned)offs[grp];
    if ((grow & 15) == 0) hdr[grp].z = off;
    const int j = plan_job_of(b, grow);
    const i64 row = grow - b.base[j];
    const uint2 h = *reinterpret_cast<const uint2*>(&hdr[grp]);
    unsigned long long m = (unsigned long long)h.x | ((unsigned long long)h.y << 32);
    i64 p = 0, pe = 0;
    const uint8_t* kidx = b.kidx[j];
    const int32_t* nidx = b.nidx[j];
    if (row < b.rows[j]) {
        const i64 q = b.perm[j] ? b.perm[j][row] : row;
        p = b.rs[j][q];
        pe = b.rs[j][q + 1];
    }
    int32_t* line = pool + (i64)off * 16 + (grow & 15);
    while (m) {
        const int k = __builtin_ctzll(m);
        m &= m - 1;
        while (p < pe && kidx[p] < k) ++p;
        int32_t v = -1;
        if (p < pe && kidx[p] == k) v = nidx[p++];
        *line = v;
        line += 16;
    }
}
}  // namespace

int asr_geom_conv_plan_batch(asr_hip_context* ctx, Arena& keep, asr_conv_plan* plans, int n) {
    if (n > PLAN_MAX_JOBS) ASR_FAIL(ctx, ASR_HIP_EINVAL, "conv_plan_batch: too many lists");
    PlanBatch b;
    memset(&b, 0, sizeof(b));
    b.n = 0;
    i64 total = 0;
    int which[PLAN_MAX_JOBS];
    for (int j = 0; j < n; ++j) {
        asr_conv_plan& pl = plans[j];
        pl.hdr = nullptr;
        pl.pool = nullptr;
        pl.offs = nullptr;
        pl.groups = pl.groups_pad = pl.blocks = 0;
        if (pl.num_out <= 0) continue;
        pl.groups = (pl.num_out + 15) / 16;
        pl.groups_pad = (pl.groups + 15) / 16 * 16;
        which[b.n] = j;
        b.base[b.n] = total;
        b.rows[b.n] = pl.num_out;
        b.K[b.n] = pl.K;
        b.nidx[b.n] = pl.nidx;
        b.kidx[b.n] = pl.kidx;
        b.rs[b.n] = pl.rs;
        b.perm[b.n] = pl.perm;
        total += pl.groups_pad * 16;
        ++b.n;
    }
    b.base[b.n] = total;
    if (b.n == 0) return ASR_HIP_OK;
    const i64 groups = total / 16;
    uint4* hdr = (uint4*)keep.alloc(groups * sizeof(uint4));
    i64* offs = (i64*)keep.alloc((groups + 1) * sizeof(i64));
    i64* counts = (i64*)ctx->scratch.alloc((groups + 1) * sizeof(i64));
    if (!hdr || !offs || !counts) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    ASR_HIP_CHECK(ctx, hipMemsetAsync(counts + groups, 0, sizeof(i64), ctx->stream));
    k_plan_masks_batch<<<grid_for(total, BLK), BLK, 0, ctx->stream>>>(b, hdr, counts);
    ASR_CHECK_LAUNCH(ctx);
    ASR_TRY(asr_prim::scan_counts(ctx, ctx->scratch, counts, offs, groups + 1));
    i64 blocks = 0;
    ASR_TRY(asr_prim::read_i64(ctx, offs + groups, &blocks));
    int32_t* pool = (int32_t*)keep.alloc((size_t)(blocks > 0 ? blocks : 1) * 64);
    if (!pool) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    k_plan_fill_batch<<<grid_for(total, BLK), BLK, 0, ctx->stream>>>(b, offs, hdr, pool);
    ASR_CHECK_LAUNCH(ctx);
    for (int t = 0; t < b.n; ++t) {
        asr_conv_plan& pl = plans[which[t]];
        pl.hdr = hdr + b.base[t] / 16;
        pl.offs = offs + b.base[t] / 16;
        pl.pool = pool;
        pl.blocks = blocks;
    }
    return ASR_HIP_OK;
}

int asr_geom_conv_plan_build(asr_hip_context* ctx, Arena& keep, const int32_t* nidx, const uint8_t* kidx, const i64* rs,
                             const int32_t* perm, i64 num_out, int K, asr_conv_plan* plan) {
    ASR_TRY(asr_geom_conv_plan_count(ctx, keep, nidx, kidx, rs, perm, num_out, K, plan));
    if (num_out <= 0) return ASR_HIP_OK;
    i64 blocks = 0;
    ASR_TRY(asr_prim::read_i64(ctx, plan->offs + plan->groups_pad, &blocks));
    if (blocks >= PLAN_DUP_FLAG)
        ASR_FAIL(ctx, ASR_HIP_EINVAL,
                 "sparse_conv plan: a row lists a kernel slot twice; the 16-bit kernels take one neighbour per (row, slot) "
                 "(use asr_hip_sparse_conv_f32 with algo = 1 for such lists)");
    return asr_geom_conv_plan_fill(ctx, keep, plan, blocks);
}

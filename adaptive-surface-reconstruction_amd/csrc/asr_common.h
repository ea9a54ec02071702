// asr_common.h -- context, device arena and shared device helpers (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/asr_hip.h"

typedef uint64_t u64;
typedef int64_t i64;
typedef uint32_t u32;

// ---------------------------------------------------------------------------------------
// Device arena: a list of hipMalloc'd slabs with bump allocation.  reset() rewinds the bump
// pointers but keeps the slabs, so steady-state forwards do not call hipMalloc.
// ---------------------------------------------------------------------------------------
struct Arena {
    struct Slab {
        char* base;
        size_t size, used;
    };
    std::vector<Slab> slabs;
    size_t min_slab = size_t(256) << 20;
    size_t cap = 0;  // option "arena_cap_mb": the arena never reserves more than this many bytes (0: no limit)

    void* alloc(size_t bytes) {
        bytes = (bytes + 255) & ~size_t(255);
        if (bytes == 0) bytes = 256;
        for (auto& s : slabs)
            if (s.size - s.used >= bytes) {
                void* p = s.base + s.used;
                s.used += bytes;
                return p;
            }
        size_t sz = bytes > min_slab ? bytes : min_slab;
        if (cap && reserved() + sz > cap) return nullptr;
        void* p = nullptr;
        if (hipMalloc(&p, sz) != hipSuccess) return nullptr;
        slabs.push_back({(char*)p, sz, bytes});
        return p;
    }
    void reset() {
        for (auto& s : slabs) s.used = 0;
    }
    size_t reserved() const {
        size_t t = 0;
        for (auto& s : slabs) t += s.size;
        return t;
    }
    void release() {
        for (auto& s : slabs) (void)hipFree(s.base);
        slabs.clear();
    }
};

// mark/rewind: remembers the bump pointers of an arena
struct ArenaMark {
    std::vector<size_t> used;
};
static inline void arena_mark(const Arena& a, ArenaMark& m) {
    m.used.clear();
    for (auto& s : a.slabs) m.used.push_back(s.used);
}
static inline void arena_rewind(Arena& a, const ArenaMark& m) {
    for (size_t i = 0; i < a.slabs.size(); ++i) a.slabs[i].used = i < m.used.size() ? m.used[i] : 0;
}

struct GridDev {
    i64 v = 0, p = 0;
    u64* keys = nullptr;
    float* centers = nullptr;
    float* sizes = nullptr;
    int32_t* nidx = nullptr;
    uint8_t* nkidx = nullptr;
    i64* nrs = nullptr;
    int32_t* up_idx = nullptr;
    uint8_t* up_kidx = nullptr;
    i64* up_rs = nullptr;
    // inverted up lists (rows = next coarser grid)
    int32_t* down_idx = nullptr;
    uint8_t* down_kidx = nullptr;
    i64* down_rs = nullptr;
    // MFMA tiling orders (asr_geom_row_groups) of the three CSRs whose rows live on this grid
    int32_t* perm_nb = nullptr;    // rows = this grid (55-slot lists)
    int32_t* perm_up = nullptr;    // rows = this grid (up lists, inputs from the coarser grid)
    int32_t* perm_down = nullptr;  // rows = next coarser grid (inverted up lists)
};

// Tunables of one context (asr_hip_context_set_option); the defaults can also be seeded from the
// environment (ASR_SCONV_MIN_BLOCKS, ...) when the context is created, for experiments.
struct AsrOptions {
    i64 sconv_min_blocks = 2816;  // narrow the column tile until a launch has this many blocks (11 per CU)
    i64 sconv16_min_blocks = 1024;  // the same for the 16-bit kernels (4 per CU: every column chunk of a tile gathers the
                                    // features again, and these kernels are bound by the load path on the small grids)
    i64 sconv_wide_min = 2048;    // 8-wave (128-row) blocks when they still give this many blocks
    i64 row_segment = 524288;     // rows are regrouped inside segments of this many consecutive rows
    i64 search_hash_level = -1;   // >= 0: finest level of the search's cell hash table (finer: binary search); tests
    i64 knn_deep = 1;             // kNN radius: finer start levels for the points of crowded cells (0: off)
    i64 knn_cells = 1;            // kNN radius: cell-parallel fast path (0: wave per point only)
    i64 plan_arena = 0;           // asr_hip_sparse_conv_plan_create: 1 = memory from the context's plan arena (no hipMalloc /
                                  // hipFree per plan; all such plans die with asr_hip_context_plan_arena_reset)
    i64 sconv_plan = 1;           // 16-bit sparse conv: plan-driven kernel where it applies (0: table-driven)
    i64 row_ranked = 1;           // row regrouping: sort on (job, segment, rank of the slot mask) keys, all lists in one sort
    i64 row_lpt = 1;              // longest-first order of the 128-row chunks of a segment
    i64 overlap = 1;              // aggregation search on the auxiliary stream, overlapped with the grids
    i64 cconv_valu = 0;           // 1: whole-path continuous conv with the VALU contraction (k_cconv) instead of k_cconv_mfma
    i64 build_search = 1;         // 0: implicit_build stops after the grids (sharded runs search their own rows)
    i64 shard_geometry = -1;      // sharded forward: 1 = lists / plans / search for the owned voxels only, 0 = whole cloud on every rank, -1 = 1 whenever world > 1
    i64 shard_timing = 0;         // sharded forward: synchronise around every halo exchange and accumulate its wall time
    i64 search_priority = 2;      // priority of the search's stream (set before the first build): 0 lowest, 1 middle, 2 highest
                                  // (round 4: the search is the longer of the two chains; 9.6 -> 8.8 ms on its stream)
    i64 search_half = 1;          // aggregation search: 4^3 half-size cells per voxel (0: 3^3 full-size cells)
    i64 early_cells = 1;            // ... and its cell table, on the search thread
    i64 early_sort = 1;           // overlapped search: its point sort starts on the auxiliary stream beside the octree build
    // 16-bit sparse conv, 55-slot lists whose INPUT grid has between these many rows (the coarse levels: a handful of
    // tiles, the longest of which runs its 30 slots x cin / 32 steps alone on its CU): the slots are cut into five fixed
    // ranges, every (tile, range) is a block of its own, a second kernel adds the partial sums in range order.  The
    // choice depends on the input grid's size only, so that one rank of a sharded cloud makes the same one.
    i64 arena_cap_mb = 0;         // > 0: each device arena of the context (results, scratch, shard state) stops growing at this
                                  // many MiB and the call fails with "arena allocation failed" instead (memory budget per rank)
    i64 inject_failure = 0;       // fault injection for the tests of the sharded forward's error agreement: 1 = the build of this
                                  // rank fails, 2 = its network preparation fails (the call RETURNS AN ERROR, nothing is skipped)
    i64 sconv_split_min_rows = 2048;
    i64 sconv_split_rows = 32768;  // 0: never
};

// Row-group plan of a neighbour list for the plan-driven 16-bit sparse conv (asr_conv16.hip): per 16
// consecutive rows (row_perm order) the set of kernel slots in use and, slot by slot, the 16 neighbour indices.
struct asr_conv_plan_view {
    const uint4* hdr;      // [groups_pad] {slot mask lo, hi, first pool block, 0}
    const int32_t* pool;   // [blocks][16], -1: no neighbour
    i64 groups;            // ceil(num_out / 16)
    unsigned pool_bytes;
};
struct asr_conv_plan {
    uint4* hdr = nullptr;
    int32_t* pool = nullptr;
    i64 groups = 0, groups_pad = 0, blocks = 0;
    i64* offs = nullptr;  // [groups_pad + 1] first pool block of every group; offs[groups_pad] = blocks (device)
    // the list it was built from
    const int32_t* nidx = nullptr;
    const uint8_t* kidx = nullptr;
    const i64* rs = nullptr;
    const int32_t* perm = nullptr;
    i64 num_out = 0;
    int K = 0;
    bool usable() const { return hdr && blocks * 64 < (i64(1) << 32) - 65536; }
    asr_conv_plan_view view() const { return {hdr, pool, groups, (unsigned)(blocks * 64)}; }
};

// Points of the last implicit_build in Morton order (asr_geom_presort): shared by the octree insertion (neighbouring
// lanes then carry the same key: one table probe per run) and by the aggregation search.
struct AsrPointIndex {
    bool valid = false;
    const float* pts = nullptr;
    const float* radii = nullptr;
    i64 n = 0;
    int lsort = 0;               // sorted on the code bits down to this level
    float4* sorted = nullptr;    // (x, y, z, original index bits)
    int32_t* ids = nullptr;      // original index per Morton position
    int32_t* rank = nullptr;     // Morton position per original index
    float* srad = nullptr;       // radii in Morton order
    u64* codes = nullptr;        // sorted level-21 codes
    // cell table of the levels [tab_lmin, tab_lmax] built ahead of the query (asr_geom_precells), or tab_keys == nullptr
    u64* tab_keys = nullptr;
    u64 tab_mask = 0;
    int32_t* tab_start = nullptr;
    int32_t* tab_end = nullptr;
    int tab_lmin = 0, tab_lmax = -1, tab_lhash = 0, tab_grow = 0;
};

struct asr_hip_context {
    hipStream_t stream = nullptr;
    std::string err;
    AsrOptions opt;
    std::map<const void*, asr_conv_plan> conv_plans;  // row splits pointer -> plan of that list (implicit_build)
    std::map<std::string, i64> sconv_launches;  // "NT,KC,IMP,WAVES,DUAL" -> launches (asr_hip_sparse_conv_variant_counts)
    ArenaMark build_mark;                         // persist arena right after implicit_build
    bool build_mark_ok = false;
    Arena persist;  // results that outlive a call (octree, grids, values)
    Arena scratch;  // temporaries
    Arena plan_arena;  // row-group plans created with option "plan_arena" (asr_hip_context_plan_arena_reset)
    uint64_t plan_arena_epoch = 0;  // generation of plan_arena: plans of an earlier generation are refused
    // last octree
    u64* nodes = nullptr;
    u64* leaves = nullptr;
    i64 num_nodes = 0, num_leaves = 0;
    // last implicit build
    asr_octree_frame frame;
    asr_implicit_sizes sizes;
    GridDev grids[ASR_NUM_GRIDS];
    int32_t* agg_idx = nullptr;
    float* agg_dist = nullptr;
    float* agg_compat = nullptr;
    i64* agg_rs = nullptr;
    int32_t* agg_spos = nullptr;         // neighbour of each aggregation pair as a position in Morton order
    const float4* agg_sorted = nullptr;  // points in Morton order (x, y, z, original index bits)
    bool has_search = false;             // the last implicit_build ran the aggregation search
    // the last implicit_build was ONE RANK's build of a sharded cloud (asr_hip_implicit_forward_sharded, option shard_geometry):
    // 55-slot lists, plans and the aggregation CSR hold the owned rows only -- the stand-alone network entry points refuse it
    bool build_sharded = false;
    // sharded geometry: the search covered agg_nq (> 0) rows of grid 0 only -- agg_rows lists them, the CSR above and
    // the query centres / sizes below are compact over that list
    const int32_t* agg_rows = nullptr;
    i64 agg_nq = 0;
    const float* agg_qcenters = nullptr;
    const float* agg_qsizes = nullptr;
    AsrPointIndex pindex;                // asr_geom_presort
    i64 shard_prefix_hint = 0;           // rows of the importance prefix the last sharded build needed
    int search_extras = 0;               // pairs the last aligned search took from the rounding margin (diagnostic)
    int leaf_lmin = -1, leaf_lmax = -1;  // levels of the first / last leaf of the last octree
    float* values = nullptr;
    float* feats1 = nullptr;
    int feats1_width = 0;
    float* importance = nullptr;
    float* code = nullptr;
    float stage_ms[6] = {0, 0, 0, 0, 0, 0};
    hipEvent_t ev[8] = {};
    bool ev_ok = false;
    int* d_flags = nullptr;  // the CURRENT block of 64 device counters: a window into d_flags_base (asr_prim.h fresh_flags)
    int* d_flags_base = nullptr;  // pool of zeroed counter blocks, re-zeroed with ONE memset when it is used up
    int flags_next = 0;
    float* d_zeros = nullptr;  // 4 KB of zeros: target of masked-out loads
    float* split_part = nullptr;   // partial sums of the slot-range split of the 16-bit sparse conv (grown on demand)
    size_t split_part_bytes = 0;
    unsigned* d_status = nullptr;  // sharded forward: the status word of asr_shard_agree
    unsigned* d_absmax = nullptr;  // f16x2: running maxima of the network's activation buffers; [255]: one-off inputs
    void* radius_state = nullptr;  // RadiusState of asr_geom.hip (between _count and _fill)
    void* mesh_state = nullptr;    // MeshState of asr_mesh.hip (between _count and _fill)
    std::map<std::string, std::pair<const void*, size_t>> named;  // asr_hip_implicit_get
    int device = 0;                 // HIP device of this context (helper threads must select it)
    asr_hip_context* aux = nullptr;  // second stream + arenas: the aggregation search runs there, overlapped
                                    // with the grid hierarchy build (asr_api.hip implicit_build)
    bool aux_stream_owned = false;
    hipEvent_t aux_ev = nullptr;
    hipEvent_t aux_t0 = nullptr, aux_t1 = nullptr;  // search start / end on the auxiliary stream
    bool search_overlapped = false;
    // packed 16-bit copies of the weight tensors of the whole-path driver, keyed by (pointers of the f32 tensors
    // of both banks, their shapes, mode).  A weight table is immutable while a context holds packed copies of it:
    // callers that update weights in place call asr_hip_context_weights_changed (the copies are made again).
    typedef std::tuple<const void*, const void*, i64, i64, i64, i64, int> PackedKey;
    std::map<PackedKey, void*> packed_weights;
    Arena shard_mem;  // ownership, halo lists and plans of the sharded forward (reset per forward, kept across)
    unsigned* shard_stage_send = nullptr;  // message staging of the halo exchanges (grow-only)
    unsigned* shard_stage_recv = nullptr;
    size_t shard_stage_cap = 0;  // dwords each
    struct asr_shard_state* shard = nullptr;  // set while asr_hip_implicit_forward_sharded runs its network half
    // preparation pass of the sharded network (asr_api.hip implicit_network, phase 1): every argument check, weight packing
    // and allocation of a convolution happens, the kernel launch and the halo exchange do not
    bool dry_launch = false;
};

#define ASR_FAIL(ctx, code, ...)                         \
    do {                                                 \
        char _b[512];                                    \
        snprintf(_b, sizeof(_b), __VA_ARGS__);           \
        (ctx)->err = _b;                                 \
        return (code);                                   \
    } while (0)

#define ASR_HIP_CHECK(ctx, expr)                                                            \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess)                                                               \
            ASR_FAIL(ctx, ASR_HIP_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                     __FILE__, __LINE__);                                                   \
    } while (0)

#define ASR_CHECK_LAUNCH(ctx) ASR_HIP_CHECK(ctx, hipGetLastError())

#define ASR_TRY(expr)               \
    do {                            \
        int _r = (expr);            \
        if (_r != ASR_HIP_OK) return _r; \
    } while (0)

static inline int asr_ctx_zeros(asr_hip_context* ctx, const float** out) {
    if (!ctx->d_zeros) {
        ASR_HIP_CHECK(ctx, hipMalloc((void**)&ctx->d_zeros, 4096));
        ASR_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_zeros, 0, 4096, ctx->stream));
    }
    *out = ctx->d_zeros;
    return ASR_HIP_OK;
}

template <class T>
static inline T* arena_alloc(Arena& a, size_t count) {
    return (T*)a.alloc(count * sizeof(T));
}

static inline unsigned grid_for(i64 n, int block) {
    i64 g = (n + block - 1) / block;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// ---------------------------------------------------------------------------------------
// device helpers: Morton / location codes (cpp/lib/zindex.h, octreebase.h)
// ---------------------------------------------------------------------------------------
__host__ __device__ static inline u64 asr_dilate21(u64 x) {
    x = (x | (x << 32)) & 0x001F00000000FFFFull;
    x = (x | (x << 16)) & 0x00FF0000FF0000FFull;
    x = (x | (x << 8)) & 0xF00F00F00F00F00Full;
    x = (x | (x << 4)) & 0x30C30C30C30C30C3ull;
    x = (x | (x << 2)) & 0x9249249249249249ull;
    return x;
}
__host__ __device__ static inline u64 asr_compact21(u64 x) {
    x &= 0x1249249249249249ull;
    x = ((x >> 2) | x) & 0x30C30C30C30C30C3ull;
    x = ((x >> 4) | x) & 0xF00F00F00F00F00Full;
    x = ((x >> 8) | x) & 0x00FF0000FF0000FFull;
    x = ((x >> 16) | x) & 0x001F00000000FFFFull;
    x = ((x >> 32) | x) & 0x00000000001FFFFFull;
    return x;
}
__host__ __device__ static inline u64 asr_morton3d(u64 x, u64 y, u64 z) {
    return asr_dilate21(x) | (asr_dilate21(y) << 1) | (asr_dilate21(z) << 2);
}
__device__ static inline int asr_key_level(u64 key) { return (63 - __clzll((long long)key)) / 3; }
// key of (x,y,z,lev) or 0 if outside the level's cube (octreebase.h:59-65,120-128)
__device__ static inline u64 asr_coord_key(int x, int y, int z, int lev) {
    int lim = 1 << lev;
    if (lev > ASR_MAX_LEVEL || x < 0 || x >= lim || y < 0 || y >= lim || z < 0 || z >= lim)
        return 0;
    return asr_morton3d((u64)x, (u64)y, (u64)z) | (u64(1) << (3 * lev));
}
__device__ static inline void asr_key_coord(u64 key, int& x, int& y, int& z, int& lev) {
    lev = asr_key_level(key);
    u64 k = key & ~(u64(1) << (3 * lev));
    x = (int)asr_compact21(k);
    y = (int)asr_compact21(k >> 1);
    z = (int)asr_compact21(k >> 2);
}

// 64-bit finaliser (murmur3 fmix64) used by all device hash tables
__device__ static inline u64 asr_hash64(u64 k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return k;
}

int asr_ctx_ensure_aux(asr_hip_context* ctx);  // the search's auxiliary context + stream (asr_api.hip)
// one scan over several GPUs (asr_shard.hip): ownership, owned row lists / plans, halo lists of the 13 neighbour lists
// of the last implicit_build; hooks of the network driver
struct asr_shard_state;
int asr_shard_build(asr_hip_context* ctx, const asr_shard_comm* comm, int want_plans, asr_shard_state** out);
// the same in two steps for a rank that builds the 55-slot lists of its own rows only (by_pairs 0: equal voxel counts)
int asr_shard_ownership(asr_hip_context* ctx, const asr_shard_comm* comm, int by_pairs, asr_shard_state** out);
int asr_shard_ownership_coarser(asr_hip_context* ctx, asr_shard_state* st);
const int32_t* asr_shard_owner(const asr_shard_state* st, int level);
int asr_shard_lists(asr_hip_context* ctx, asr_shard_state* st, int want_plans);
const int32_t* asr_shard_level_rows(const asr_shard_state* st, int level, i64* n);
int asr_shard_query_rows(asr_hip_context* ctx, const asr_shard_state* st, i64 prefix, Arena& keep, int32_t** rows_out,
                         i64* n_out);
int asr_shard_world(const asr_shard_state* st);
int asr_shard_rank(const asr_shard_state* st);
void asr_shard_free(asr_shard_state* st);
void asr_shard_release(asr_hip_context* ctx);  // the context's shard arena and staging buffers
const asr_shard_stats* asr_shard_get_stats(const asr_shard_state* st);
// all ranks learn whether ANY rank failed so far (one MAX all-reduce of a status word): rc_local when this rank failed,
// ASR_HIP_EPEER when only others did, ASR_HIP_OK when nobody did
int asr_shard_agree(asr_hip_context* ctx, const asr_shard_comm* comm, int rc_local, const char* phase);
int asr_shard_before_conv(asr_hip_context* ctx, asr_shard_state* st, const void* rs, void* feat, i64 ld_bytes, i64 row_bytes,
                          float* imp, unsigned* in_amax, const int32_t** perm, i64* num_out, const asr_conv_plan** plan);
int asr_shard_stitch(asr_hip_context* ctx, asr_shard_state* st, float* values);
const int32_t* asr_shard_owned_rows0(const asr_shard_state* st, i64* n);  // grid-0 rows of this rank, ascending (world > 1)

// internal entry points shared between translation units
int asr_geom_point_keys(asr_hip_context* ctx, const asr_octree_frame* frame, const float* pts,
                        const float* radii, i64 n, float radius_scale, int max_depth, u64* keys);
int asr_geom_octree_build(asr_hip_context* ctx, const asr_octree_frame* frame, const float* pts,
                          const float* radii, i64 n, float radius_scale, int max_depth, int grow_steps = 0,
                          const u64* extra_keys = nullptr, i64 num_extra = 0, bool balance = true);
// cell table of ctx->pindex (after asr_geom_presort) for every level a leaf can have; blocks on ctx->stream
int asr_geom_precells(asr_hip_context* ctx, Arena& keep);
// Morton order of the points (+ radii) into ctx->pindex, arrays in `keep`
int asr_geom_presort(asr_hip_context* ctx, Arena& keep, const asr_octree_frame* frame, const float* pts,
                     const float* radii, i64 n, float radius_scale = 0.f, int max_depth = 21);
int asr_geom_neighbors_count(asr_hip_context* ctx, const u64* keys, i64 v, i64* rs, i64* num_pairs);
int asr_geom_neighbors_fill(asr_hip_context* ctx, const u64* keys, i64 v, const i64* rs,
                            int32_t* idx, uint8_t* kidx);
int asr_geom_neighbors_rows_count(asr_hip_context* ctx, const u64* keys, i64 v, const int32_t* rows, i64 nrows, i64* rs,
                                  i64* num_pairs);
int asr_geom_neighbors_rows_fill(asr_hip_context* ctx, const u64* keys, i64 v, const int32_t* rows, i64 nrows,
                                 const i64* rs, int32_t* idx, uint8_t* kidx);
int asr_geom_neighbors_build(asr_hip_context* ctx, Arena& out_arena, const u64* keys, i64 v,
                             i64** rs_out, int32_t** idx_out, uint8_t** kidx_out, i64* num_pairs);
// the neighbour lists of all grids of a hierarchy in ONE pass: one launch each for the key maps, the counting pass and
// the filling pass, one scan, one read-back of the pair counts (round 4; was five times count / scan / read-back / fill)
struct asr_nb_job {
    const u64* keys;  // in: sorted voxel keys
    i64 v;
    i64* rs;          // out (arrays in out_arena)
    int32_t* idx;
    uint8_t* kidx;
    i64 p;
    const int32_t* owner = nullptr;  // optional: lists for the rows with owner[row] == me only (the others stay empty)
    int me = 0;
};
int asr_geom_neighbors_build_batch(asr_hip_context* ctx, Arena& out_arena, asr_nb_job* jobs, int n);
int asr_geom_row_groups(asr_hip_context* ctx, const uint8_t* kidx, const i64* rs, i64 v, i64 seg,
                        int32_t* perm_out, int kbits);
struct asr_row_group_job {
    const uint8_t* kidx;
    const i64* rs;
    i64 v;
    int kbits;
    int32_t* perm_out;
};
int asr_geom_row_groups_batch(asr_hip_context* ctx, const asr_row_group_job* jobs, int n, i64 seg);
int asr_geom_coarsen_count(asr_hip_context* ctx, const u64* keys, i64 v, i64* v_out);
// down_*: optional inverted up lists (rows = coarse voxels; what open3d::invert_neighbors_list returns for the up lists)
int asr_geom_coarsen_fill(asr_hip_context* ctx, const u64* keys, i64 v, u64* out_keys, i64 v_out,
                          int32_t* up_idx, uint8_t* up_kidx, i64* up_rs, int32_t* down_idx = nullptr,
                          uint8_t* down_kidx = nullptr, i64* down_rs = nullptr);
int asr_geom_coarsen_build(asr_hip_context* ctx, Arena& keep, const u64* keys, i64 v, u64** out_keys, i64* v_out,
                           int32_t** up_idx, uint8_t** up_kidx, i64** up_rs, int32_t** down_idx, uint8_t** down_kidx,
                           i64** down_rs, int key_bits = 64);  // key_bits: significant bits of the largest key
int asr_geom_voxel_info(asr_hip_context* ctx, const asr_octree_frame* frame, const u64* keys,
                        i64 v, float* centers, float* sizes);
// keep: arena for the Morton-ordered point arrays (scratch when null); fill: spos (optional) receives
// each pair's neighbour as a position in Morton order, sorted_out the Morton-ordered points (x, y, z,
// original index bits) they refer to
// radii: gathered into Morton order together with the points (the fill call then needs no pass of its own);
// voxel_keys: the queries are the centres of these grid voxels with radius = voxel size ("aligned": half-size search
// cells, see AlignedQ in asr_geom.hip); lmin / lmax: levels of the queries when the caller knows them (else -1)
int asr_geom_radius_count(asr_hip_context* ctx, const asr_octree_frame* frame, const float* pts,
                          i64 n, const float* centers, const float* sizes, i64 v, i64* rs,
                          i64* num_pairs, Arena* keep = nullptr, const float* radii = nullptr,
                          const u64* voxel_keys = nullptr, int lmin_hint = -1, int lmax_hint = -1,
                          const AsrPointIndex* pre = nullptr);

int asr_geom_radius_fill(asr_hip_context* ctx, const float* pts, const float* radii, i64 n,
                         const float* centers, const float* sizes, i64 v, const i64* rs,
                         int32_t* idx, float* dist, float* compat, int32_t* spos = nullptr,
                         const float4** sorted_out = nullptr);
int asr_geom_radius_neighbor_count(asr_hip_context* ctx, const asr_octree_frame* frame, const float* pts,
                                   const float* radii, i64 n, i64* counts_out);
int asr_geom_knn(asr_hip_context* ctx, const asr_octree_frame* frame, const float* pts, i64 n, int k,
                 const float* radii_in, float radius_fraction, int outlier_threshold, float* radii_out,
                 uint8_t* inlier_out);
int asr_geom_dual_count(asr_hip_context* ctx, const u64* nodes, i64 num_nodes, const u64* leaves, i64 num_leaves,
                        i64* num_cells);
int asr_geom_dual_fill(asr_hip_context* ctx, i64* out);
int asr_geom_invert(asr_hip_context* ctx, i64 num_points, const int32_t* idx, const i64* rs,
                    i64 num_rows, const uint8_t* attr, int32_t* out_idx, i64* out_rs,
                    uint8_t* out_attr, i64 known_pairs = -1);
void asr_geom_release(asr_hip_context* ctx);

int asr_mesh_contour_count(asr_hip_context* ctx, const float* values, i64 num_values, const i64* duals,
                           i64 num_duals, const float* positions, float threshold, i64* num_vertices,
                           i64* num_triangles);
int asr_mesh_contour_fill(asr_hip_context* ctx, float* vertices, int32_t* triangles);
int asr_mesh_components_count(asr_hip_context* ctx, const float* vertices, i64 nv, const int32_t* triangles,
                              i64 nt, i64 keep_n, i64 min_size, i64* nv_out, i64* nt_out);
int asr_mesh_components_fill(asr_hip_context* ctx, float* vertices_out, int32_t* triangles_out);
void asr_mesh_release(asr_hip_context* ctx);

int asr_conv_agg_importance(asr_hip_context* ctx, const float* compat, const float* dist, i64 n,
                            float* out);
int asr_conv_cconv(asr_hip_context* ctx, const float* filters, const float* out_pos,
                   const float* extents, const float* inp_pos, const float* inp_feat,
                   const int32_t* nidx, const float* nimp, const i64* rs, i64 num_out, int cin,
                   int cout, int normalize, const float* bias, int relu, float* out, int sorted4 = 0,
                   unsigned* out_absmax = nullptr);  // != null: max(its value, f32 bits of the largest |out|), kept by the kernels
int asr_conv_cconv_basis(asr_hip_context* ctx, const float* out_pos, const float* extents, const float* inp_pos,
                         const float* inp_feat, const int32_t* nidx, const float* nimp, const i64* rs, i64 num_out,
                         float* basis_out, float* norm_out);
int asr_conv_sparse(asr_hip_context* ctx, const asr_sparse_conv_args* args);
// asr_conv16.hip: 16-bit matrix-core variants (f16 activations / exact bf16x3 split)
// two phases so that callers with several lists pay one host read-back: count enqueues the slot-set pass and
// the scan (the block total lands in plan->offs[groups_pad]); fill allocates the pool and writes it
int asr_geom_conv_plan_count(asr_hip_context* ctx, Arena& keep, const int32_t* nidx, const uint8_t* kidx, const i64* rs,
                             const int32_t* perm, i64 num_out, int K, asr_conv_plan* plan);
int asr_geom_conv_plan_fill(asr_hip_context* ctx, Arena& keep, asr_conv_plan* plan, i64 blocks);
// plans[j] with nidx/kidx/rs/perm/num_out/K set: all built in one pass over the concatenated lists, one pool
int asr_geom_conv_plan_batch(asr_hip_context* ctx, Arena& keep, asr_conv_plan* plans, int n);
int asr_geom_conv_plan_build(asr_hip_context* ctx, Arena& keep, const int32_t* nidx, const uint8_t* kidx, const i64* rs,
                             const int32_t* perm, i64 num_out, int K, asr_conv_plan* plan);  // count + read-back + fill

size_t asr_conv16_packed_bytes(int mode, int K, int cin, int cout, int cout_b);
int asr_conv16_pack(asr_hip_context* ctx, int mode, const float* wa, const float* wb, int K, int cin, int ca, int cb,
                    void* out);
int asr_conv16_convert(asr_hip_context* ctx, const void* in, i64 n, void* out, int to_f16);
int asr_conv16_absmax(asr_hip_context* ctx, const float* x, i64 rows, int c, i64 ld, unsigned* out);
int asr_conv_sparse16(asr_hip_context* ctx, const asr_sparse_conv_args* args, const void* packed, int mode,
                      int out_f16, const asr_conv_plan* plan);
int asr_conv_reduce(asr_hip_context* ctx, const float* values, const int32_t* gidx, const i64* rs,
                    i64 rows, float* out);
int asr_conv_decode(asr_hip_context* ctx, const float* code, i64 v, int c, const float* w1,
                    const float* b1, int h1, const float* w2, const float* b2, int h2,
                    const float* w3, const float* sizes, float* out, const int32_t* rows = nullptr);  // rows: v listed rows

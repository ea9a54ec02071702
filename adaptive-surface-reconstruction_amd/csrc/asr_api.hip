// asr_api.hip -- the C ABI (include/asr_hip.h) and the whole-path driver that replaces the
// section of asr::ReconstructSurface between pre-filter and contouring
// (cpp/lib/asr.cpp:143-336) plus UNet5.aggregate/unet/decode
// (models/v0/net_definitions_torch.py:535-666).
#include <cmath>
#include <cstdlib>
#include <cstring>

#include <algorithm>
#include <functional>
#include <future>
#include <mutex>
#include <thread>

#include "asr_common.h"
#include "asr_uset.h"

// rows are regrouped inside segments of this many consecutive rows (option "row_segment")
#define ASR_ROW_GROUP_SEGMENT (ctx->opt.row_segment > 0 ? ctx->opt.row_segment : (i64)524288)
// every entry point selects the context's device on the calling thread: a context is bound to the
// device that was current when it was created, whatever the caller's current device is now
#define CTX_GUARD(ctx)                                                  \
    if (!(ctx)) return ASR_HIP_EINVAL;                                  \
    (ctx)->err.clear();                                                 \
    if (hipSetDevice((ctx)->device) != hipSuccess) {                    \
        (ctx)->err = "hipSetDevice failed for the context's device";   \
        return ASR_HIP_EHIP;                                            \
    }

namespace {
struct OptEntry {
    const char* name;
    const char* env;
    i64 AsrOptions::*field;
};
const OptEntry k_options[] = {
        {"sconv_min_blocks", "ASR_SCONV_MIN_BLOCKS", &AsrOptions::sconv_min_blocks},
        {"sconv_wide_min", "ASR_SCONV_WIDE_MIN", &AsrOptions::sconv_wide_min},
        {"row_segment", "ASR_ROW_SEGMENT", &AsrOptions::row_segment},
        {"row_lpt", "ASR_ROW_LPT", &AsrOptions::row_lpt},
        {"row_ranked", "ASR_ROW_RANKED", &AsrOptions::row_ranked},
        {"sconv_plan", "ASR_SCONV_PLAN", &AsrOptions::sconv_plan},
        {"plan_arena", "ASR_PLAN_ARENA", &AsrOptions::plan_arena},
        {"sconv16_min_blocks", "ASR_SCONV16_MIN_BLOCKS", &AsrOptions::sconv16_min_blocks},
        {"knn_cells", "ASR_KNN_CELLS", &AsrOptions::knn_cells},
        {"knn_deep", "ASR_KNN_DEEP", &AsrOptions::knn_deep},
        {"search_hash_level", "ASR_SEARCH_HASH_LEVEL", &AsrOptions::search_hash_level},
        {"overlap", "ASR_OVERLAP", &AsrOptions::overlap},
        {"build_search", "ASR_BUILD_SEARCH", &AsrOptions::build_search},
        {"cconv_valu", "ASR_CCONV_VALU", &AsrOptions::cconv_valu},
        {"search_half", "ASR_SEARCH_HALF", &AsrOptions::search_half},
        {"search_priority", "ASR_SEARCH_PRIORITY", &AsrOptions::search_priority},
        {"shard_timing", "ASR_SHARD_TIMING", &AsrOptions::shard_timing},
        {"shard_geometry", "ASR_SHARD_GEOMETRY", &AsrOptions::shard_geometry},
        {"early_sort", "ASR_EARLY_SORT", &AsrOptions::early_sort},
        {"early_cells", "ASR_EARLY_CELLS", &AsrOptions::early_cells},
        {"sconv_split_rows", "ASR_SCONV_SPLIT_ROWS", &AsrOptions::sconv_split_rows},
        {"sconv_split_min_rows", "ASR_SCONV_SPLIT_MIN_ROWS", &AsrOptions::sconv_split_min_rows},
        {"arena_cap_mb", "ASR_ARENA_CAP_MB", &AsrOptions::arena_cap_mb},
        {"inject_failure", "ASR_INJECT_FAILURE", &AsrOptions::inject_failure},
};
// option "arena_cap_mb" -> the arenas of the context and of its auxiliary (search) context
void apply_arena_cap(asr_hip_context* ctx) {
    const size_t cap = ctx->opt.arena_cap_mb > 0 ? (size_t)ctx->opt.arena_cap_mb << 20 : 0;
    for (asr_hip_context* c : {ctx, ctx->aux})
        if (c) c->persist.cap = c->scratch.cap = c->shard_mem.cap = cap;
}

// asr::GetPrintCallbackFunction (cpp/lib/asr.cpp:34-37): one callback per verbosity level, process wide
struct PrintSlot {
    asr_hip_print_callback cb = nullptr;
    void* user = nullptr;
};
std::mutex g_print_mu;
PrintSlot g_print[4];
}  // namespace

extern "C" {

const char* asr_hip_version(void) { return "0.2.0+mi355x.r6"; }

int asr_hip_set_print_callback(asr_hip_print_callback callback, void* user, const int* levels, int num_levels) {
    if (num_levels < 0 || (num_levels > 0 && !levels)) return ASR_HIP_EINVAL;
    for (int i = 0; i < num_levels; ++i)
        if (levels[i] < ASR_HIP_DEBUG || levels[i] > ASR_HIP_ERROR) return ASR_HIP_EINVAL;  // "invalid verbosity level"
    std::lock_guard<std::mutex> lock(g_print_mu);
    for (int i = 0; i < num_levels; ++i) g_print[levels[i]] = PrintSlot{callback, callback ? user : nullptr};
    return ASR_HIP_OK;
}
void asr_hip_print(const char* msg, int level) {
    if (!msg || level < ASR_HIP_DEBUG || level > ASR_HIP_ERROR) return;
    PrintSlot s;
    {
        std::lock_guard<std::mutex> lock(g_print_mu);
        s = g_print[level];
    }
    if (s.cb) s.cb(msg, s.user);
}

size_t asr_hip_struct_size(const char* name) {
    if (!name) return 0;
    if (!strcmp(name, "asr_octree_frame")) return sizeof(asr_octree_frame);
    if (!strcmp(name, "asr_sparse_conv_args")) return sizeof(asr_sparse_conv_args);
    if (!strcmp(name, "asr_weight")) return sizeof(asr_weight);
    if (!strcmp(name, "asr_implicit_params")) return sizeof(asr_implicit_params);
    if (!strcmp(name, "asr_implicit_sizes")) return sizeof(asr_implicit_sizes);
    if (!strcmp(name, "asr_shard_comm")) return sizeof(asr_shard_comm);
    if (!strcmp(name, "asr_shard_stats")) return sizeof(asr_shard_stats);
    return 0;
}

int asr_hip_context_create(asr_hip_context** out, void* stream) {
    if (!out) return ASR_HIP_EINVAL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ASR_HIP_ENODEV;
    asr_hip_context* ctx = new asr_hip_context();
    ctx->stream = (hipStream_t)stream;
    (void)hipGetDevice(&ctx->device);
    memset(&ctx->sizes, 0, sizeof(ctx->sizes));
    for (const OptEntry& o : k_options)  // experiment defaults from the environment, per context
        if (const char* e = getenv(o.env)) ctx->opt.*(o.field) = atoll(e);
    apply_arena_cap(ctx);
    *out = ctx;
    return ASR_HIP_OK;
}
int asr_hip_context_set_option(asr_hip_context* ctx, const char* name, int64_t value) {
    if (!ctx || !name) return ASR_HIP_EINVAL;
    for (const OptEntry& o : k_options)
        if (!strcmp(name, o.name)) {
            ctx->opt.*(o.field) = value;
            if (ctx->aux) ctx->aux->opt.*(o.field) = value;
            apply_arena_cap(ctx);
            return ASR_HIP_OK;
        }
    ASR_FAIL(ctx, ASR_HIP_EINVAL, "unknown option '%s'", name);
}
int asr_hip_option_info(int index, const char** name, int64_t* default_value) {
    const int n = (int)(sizeof(k_options) / sizeof(k_options[0]));
    if (index < 0 || index >= n || !name || !default_value) return ASR_HIP_EINVAL;
    static const AsrOptions defaults;
    *name = k_options[index].name;
    *default_value = defaults.*(k_options[index].field);
    return ASR_HIP_OK;
}
int asr_hip_context_get_option(asr_hip_context* ctx, const char* name, int64_t* value) {
    if (!ctx || !name || !value) return ASR_HIP_EINVAL;
    if (!strcmp(name, "last_search_margin_pairs")) {  // read-only diagnostic of the last aggregation search
        *value = ctx->search_overlapped && ctx->aux ? ctx->aux->search_extras : ctx->search_extras;
        return ASR_HIP_OK;
    }
    for (const OptEntry& o : k_options)
        if (!strcmp(name, o.name)) {
            *value = ctx->opt.*(o.field);
            return ASR_HIP_OK;
        }
    ASR_FAIL(ctx, ASR_HIP_EINVAL, "unknown option '%s'", name);
}
int asr_hip_context_device(const asr_hip_context* ctx) { return ctx ? ctx->device : -1; }
int asr_hip_context_weights_changed(asr_hip_context* ctx) {
    CTX_GUARD(ctx);
    ASR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // a forward that reads the copies may still be running
    for (auto& kv : ctx->packed_weights) (void)hipFree(kv.second);
    ctx->packed_weights.clear();
    return ASR_HIP_OK;
}
int asr_hip_sparse_conv_variant_counts(asr_hip_context* ctx, char* buf, size_t cap, int reset) {
    if (!ctx || (!buf && cap)) return ASR_HIP_EINVAL;
    std::string s;
    for (auto& kv : ctx->sconv_launches) s += kv.first + ":" + std::to_string(kv.second) + ";";
    if (buf && cap) {
        if (s.size() + 1 > cap) ASR_FAIL(ctx, ASR_HIP_EINVAL, "variant_counts: buffer too small (%zu needed)", s.size() + 1);
        memcpy(buf, s.c_str(), s.size() + 1);
    }
    if (reset) ctx->sconv_launches.clear();
    return ASR_HIP_OK;
}
static void release_members(asr_hip_context* ctx) {
    for (auto& kv : ctx->packed_weights) (void)hipFree(kv.second);
    ctx->packed_weights.clear();
    asr_geom_release(ctx);
    asr_mesh_release(ctx);
    asr_shard_release(ctx);
    ctx->persist.release();
    ctx->scratch.release();
    ctx->plan_arena.release();
    if (ctx->d_flags_base) (void)hipFree(ctx->d_flags_base);
    if (ctx->d_zeros) (void)hipFree(ctx->d_zeros);
    if (ctx->d_absmax) (void)hipFree(ctx->d_absmax);
    if (ctx->d_status) (void)hipFree(ctx->d_status);
    if (ctx->split_part) (void)hipFree(ctx->split_part);
    if (ctx->ev_ok)
        for (auto& e : ctx->ev) (void)hipEventDestroy(e);
}
void asr_hip_context_destroy(asr_hip_context* ctx) {
    if (!ctx) return;
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->aux) {
        (void)hipStreamSynchronize(ctx->aux->stream);
        release_members(ctx->aux);
        if (ctx->aux_stream_owned) (void)hipStreamDestroy(ctx->aux->stream);
        if (ctx->aux_ev) (void)hipEventDestroy(ctx->aux_ev);
        if (ctx->aux_t0) (void)hipEventDestroy(ctx->aux_t0);
        if (ctx->aux_t1) (void)hipEventDestroy(ctx->aux_t1);
        delete ctx->aux;
    }
    release_members(ctx);
    delete ctx;
}
void asr_hip_context_set_stream(asr_hip_context* ctx, void* stream) {
    if (ctx) ctx->stream = (hipStream_t)stream;
}
const char* asr_hip_last_error(const asr_hip_context* ctx) { return ctx ? ctx->err.c_str() : "null context"; }
size_t asr_hip_context_reserved_bytes(const asr_hip_context* ctx) {
    if (!ctx) return 0;
    size_t t = ctx->persist.reserved() + ctx->scratch.reserved() + ctx->shard_mem.reserved() + 8 * ctx->shard_stage_cap;
    if (ctx->aux) t += ctx->aux->persist.reserved() + ctx->aux->scratch.reserved();
    return t;
}

// cpp/lib/octree.cpp:20-42 (float / double operation order is part of the contract, A.7)
int asr_octree_frame_init(asr_octree_frame* f, const float bb_min[3], const float bb_max[3]) {
    if (!f || !bb_min || !bb_max) return ASR_HIP_EINVAL;
    float center[3];
    for (int d = 0; d < 3; ++d) {
        f->bb_min[d] = bb_min[d];
        f->bb_max[d] = bb_max[d];
        center[d] = 0.5f * (bb_max[d] + bb_min[d]);
    }
    float edge = bb_max[0] - bb_min[0];
    edge = std::fmax(edge, bb_max[1] - bb_min[1]);
    edge = std::fmax(edge, bb_max[2] - bb_min[2]);
    if (!(edge > 0.f)) return ASR_HIP_EINVAL;
    f->voxel_size[0] = edge;
    f->inv_voxel_size[0] = 1 / edge;
    for (int i = 1; i <= ASR_MAX_LEVEL; ++i) {
        double tmp = edge * (1.0 / std::pow(2, i));
        f->voxel_size[i] = (float)tmp;
        f->inv_voxel_size[i] = (float)(1.0 / tmp);
    }
    for (int d = 0; d < 3; ++d) {
        volatile float new_min = center[d] - 0.5f * edge;
        volatile float t = new_min * f->inv_voxel_size[ASR_MAX_LEVEL];
        f->offset[d] = (int)(-std::floor(t));
    }
    return ASR_HIP_OK;
}

int asr_hip_point_keys(asr_hip_context* ctx, const asr_octree_frame* frame, const float* points,
                       const float* radii, int64_t n, float radius_scale, int max_depth,
                       uint64_t* keys_out) {
    CTX_GUARD(ctx);
    if (!frame || n < 0 || (n > 0 && (!points || !radii || !keys_out)))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "point_keys: null argument");
    return asr_geom_point_keys(ctx, frame, points, radii, n, radius_scale, max_depth, keys_out);
}

int asr_hip_octree_build(asr_hip_context* ctx, const asr_octree_frame* frame, const float* points,
                         const float* radii, int64_t n, float radius_scale, int max_depth,
                         int64_t* num_nodes, int64_t* num_leaves) {
    CTX_GUARD(ctx);
    if (!frame || n < 0 || (n > 0 && (!points || !radii)))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "octree_build: null argument");
    ctx->persist.reset();
    ctx->named.clear();
    ASR_TRY(asr_geom_octree_build(ctx, frame, points, radii, n, radius_scale, max_depth));
    if (num_nodes) *num_nodes = ctx->num_nodes;
    if (num_leaves) *num_leaves = ctx->num_leaves;
    return ASR_HIP_OK;
}
int asr_hip_octree_build_grow(asr_hip_context* ctx, const asr_octree_frame* frame, const float* points,
                              const float* radii, int64_t n, float radius_scale, int grow_steps, int max_depth,
                              int64_t* num_nodes, int64_t* num_leaves) {
    CTX_GUARD(ctx);
    if (!frame || n < 0 || (n > 0 && (!points || !radii)))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "octree_build: null argument");
    ctx->persist.reset();
    ctx->named.clear();
    ASR_TRY(asr_geom_octree_build(ctx, frame, points, radii, n, radius_scale, max_depth, grow_steps));
    if (num_nodes) *num_nodes = ctx->num_nodes;
    if (num_leaves) *num_leaves = ctx->num_leaves;
    return ASR_HIP_OK;
}
int asr_hip_octree_build_parts(asr_hip_context* ctx, const asr_octree_frame* frame, const float* points,
                               const float* radii, int64_t n, float radius_scale, int max_depth,
                               const uint64_t* extra_keys, int64_t num_extra, int balance, int64_t* num_nodes,
                               int64_t* num_leaves) {
    CTX_GUARD(ctx);
    if (!frame || n < 0 || num_extra < 0 || (n > 0 && (!points || !radii)) || (num_extra > 0 && !extra_keys))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "octree_build_parts: null argument");
    ctx->persist.reset();
    ctx->named.clear();
    ASR_TRY(asr_geom_octree_build(ctx, frame, points, radii, n, radius_scale, max_depth, 0, extra_keys, num_extra,
                                  balance != 0));
    if (num_nodes) *num_nodes = ctx->num_nodes;
    if (num_leaves) *num_leaves = ctx->num_leaves;
    return ASR_HIP_OK;
}
int asr_hip_octree_get(asr_hip_context* ctx, uint64_t* nodes_out, uint64_t* leaves_out) {
    CTX_GUARD(ctx);
    if (nodes_out && ctx->num_nodes)
        ASR_HIP_CHECK(ctx, hipMemcpyAsync(nodes_out, ctx->nodes, 8 * ctx->num_nodes,
                                          hipMemcpyDeviceToDevice, ctx->stream));
    if (leaves_out && ctx->num_leaves)
        ASR_HIP_CHECK(ctx, hipMemcpyAsync(leaves_out, ctx->leaves, 8 * ctx->num_leaves,
                                          hipMemcpyDeviceToDevice, ctx->stream));
    return ASR_HIP_OK;
}

int asr_hip_contour_count(asr_hip_context* ctx, const float* values, int64_t num_values, const int64_t* duals,
                          int64_t num_duals, const float* positions, float threshold, int64_t* num_vertices,
                          int64_t* num_triangles) {
    CTX_GUARD(ctx);
    if (!num_vertices || !num_triangles || num_values < 0 || num_duals < 0 ||
        (num_duals > 0 && (!values || !duals || !positions)))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "contour_count: null argument");
    return asr_mesh_contour_count(ctx, values, num_values, duals, num_duals, positions, threshold, num_vertices,
                                  num_triangles);
}
int asr_hip_contour_fill(asr_hip_context* ctx, float* vertices, int32_t* triangles) {
    CTX_GUARD(ctx);
    return asr_mesh_contour_fill(ctx, vertices, triangles);
}
int asr_hip_components_count(asr_hip_context* ctx, const float* vertices, int64_t nv, const int32_t* triangles,
                             int64_t nt, int64_t keep_n, int64_t min_size, int64_t* nv_out, int64_t* nt_out) {
    CTX_GUARD(ctx);
    if (!nv_out || !nt_out || nv < 0 || nt < 0 || (nv > 0 && !vertices) || (nt > 0 && !triangles))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "components_count: null argument");
    return asr_mesh_components_count(ctx, vertices, nv, triangles, nt, keep_n, min_size, nv_out, nt_out);
}
int asr_hip_components_fill(asr_hip_context* ctx, float* vertices_out, int32_t* triangles_out) {
    CTX_GUARD(ctx);
    return asr_mesh_components_fill(ctx, vertices_out, triangles_out);
}
int asr_density_inlier(const int64_t* counts, int64_t n, double density_percentile_threshold, uint8_t* inlier) {
    if (n < 0 || (n > 0 && (!counts || !inlier))) return ASR_HIP_EINVAL;
    if (n == 0) return ASR_HIP_OK;
    std::vector<int> c((size_t)n);
    for (int64_t i = 0; i < n; ++i) c[(size_t)i] = (int)counts[i];
    size_t middle = (size_t)((density_percentile_threshold / 100) * (double)n);
    middle = std::min<size_t>((size_t)n, std::max<size_t>(1, middle));
    std::partial_sort(c.begin(), c.begin() + middle, c.end());
    const int threshold = c[middle - 1];
    for (int64_t i = 0; i < n; ++i) inlier[i] = c[(size_t)i] > threshold;
    return ASR_HIP_OK;
}
int asr_hip_unordered_set_order(const uint32_t* xs, int n, uint32_t* out) {
    if (!xs || !out || n < 0 || n > ASR_USET_CAP) return ASR_HIP_EINVAL;
    asr_uset_order(xs, n, out);
    return ASR_HIP_OK;
}
int asr_hip_dual_cells_count(asr_hip_context* ctx, int64_t* num_cells) {
    CTX_GUARD(ctx);
    if (!num_cells) ASR_FAIL(ctx, ASR_HIP_EINVAL, "dual_cells_count: null argument");
    return asr_geom_dual_count(ctx, ctx->nodes, ctx->num_nodes, ctx->leaves, ctx->num_leaves, num_cells);
}
int asr_hip_dual_cells_count_for(asr_hip_context* ctx, const uint64_t* nodes, int64_t num_nodes,
                                 const uint64_t* leaves, int64_t num_leaves, int64_t* num_cells) {
    CTX_GUARD(ctx);
    if (!num_cells || num_nodes < 0 || num_leaves < 0 || (num_leaves > 0 && (!nodes || !leaves)))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "dual_cells_count_for: null argument");
    return asr_geom_dual_count(ctx, nodes, num_nodes, leaves, num_leaves, num_cells);
}
int asr_hip_dual_cells_fill(asr_hip_context* ctx, int64_t* out) {
    CTX_GUARD(ctx);
    if (!out) ASR_FAIL(ctx, ASR_HIP_EINVAL, "dual_cells_fill: null argument");
    return asr_geom_dual_fill(ctx, out);
}

int asr_hip_grid_neighbors_count(asr_hip_context* ctx, const uint64_t* keys, int64_t v,
                                 int64_t* row_splits_out, int64_t* num_pairs) {
    CTX_GUARD(ctx);
    if (v < 0 || !row_splits_out || !num_pairs || (v > 0 && !keys))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "grid_neighbors_count: null argument");
    ctx->scratch.reset();
    return asr_geom_neighbors_count(ctx, keys, v, row_splits_out, num_pairs);
}
int asr_hip_grid_neighbors_fill(asr_hip_context* ctx, const uint64_t* keys, int64_t v,
                                const int64_t* row_splits, int32_t* index_out,
                                uint8_t* kernel_index_out) {
    CTX_GUARD(ctx);
    if (v < 0 || (v > 0 && (!keys || !row_splits || !index_out || !kernel_index_out)))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "grid_neighbors_fill: null argument");
    ctx->scratch.reset();
    return asr_geom_neighbors_fill(ctx, keys, v, row_splits, index_out, kernel_index_out);
}
int asr_hip_grid_neighbors_rows_count(asr_hip_context* ctx, const uint64_t* keys, int64_t v, const int32_t* rows,
                                      int64_t num_rows, int64_t* row_splits_out, int64_t* num_pairs) {
    CTX_GUARD(ctx);
    if (v < 0 || num_rows < 0 || !row_splits_out || !num_pairs || (v > 0 && !keys) || (num_rows > 0 && !rows))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "grid_neighbors_rows_count: null argument");
    ctx->scratch.reset();
    return asr_geom_neighbors_rows_count(ctx, keys, v, rows, num_rows, row_splits_out, num_pairs);
}
int asr_hip_grid_neighbors_rows_fill(asr_hip_context* ctx, const uint64_t* keys, int64_t v, const int32_t* rows,
                                     int64_t num_rows, const int64_t* row_splits, int32_t* index_out,
                                     uint8_t* kernel_index_out) {
    CTX_GUARD(ctx);
    if (v < 0 || num_rows < 0 || (num_rows > 0 && (!keys || !rows || !row_splits || !index_out || !kernel_index_out)))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "grid_neighbors_rows_fill: null argument");
    ctx->scratch.reset();
    return asr_geom_neighbors_rows_fill(ctx, keys, v, rows, num_rows, row_splits, index_out, kernel_index_out);
}
int asr_hip_grid_coarsen_count(asr_hip_context* ctx, const uint64_t* keys, int64_t v,
                               int64_t* v_out) {
    CTX_GUARD(ctx);
    if (v < 0 || !v_out || (v > 0 && !keys))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "grid_coarsen_count: null argument");
    return asr_geom_coarsen_count(ctx, keys, v, v_out);
}
int asr_hip_grid_coarsen_fill(asr_hip_context* ctx, const uint64_t* keys, int64_t v,
                              uint64_t* out_keys, int64_t v_out, int32_t* up_index_out,
                              uint8_t* up_kernel_index_out, int64_t* up_row_splits_out) {
    CTX_GUARD(ctx);
    if (v < 0 || (v > 0 && (!keys || !out_keys || !up_index_out || !up_kernel_index_out ||
                            !up_row_splits_out)))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "grid_coarsen_fill: null argument");
    ctx->scratch.reset();
    return asr_geom_coarsen_fill(ctx, keys, v, out_keys, v_out, up_index_out, up_kernel_index_out,
                                 up_row_splits_out);
}
int asr_hip_voxel_info(asr_hip_context* ctx, const asr_octree_frame* frame, const uint64_t* keys,
                       int64_t v, float* centers_out, float* sizes_out) {
    CTX_GUARD(ctx);
    if (!frame || v < 0 || (v > 0 && !keys)) ASR_FAIL(ctx, ASR_HIP_EINVAL, "voxel_info: null argument");
    return asr_geom_voxel_info(ctx, frame, keys, v, centers_out, sizes_out);
}
int asr_hip_multi_radius_search_count(asr_hip_context* ctx, const asr_octree_frame* frame,
                                      const float* points, int64_t n, const float* centers,
                                      const float* sizes, int64_t v, int64_t* row_splits_out,
                                      int64_t* num_pairs) {
    CTX_GUARD(ctx);
    if (!frame || n < 0 || v < 0 || !row_splits_out || !num_pairs || (n > 0 && !points) ||
        (v > 0 && (!centers || !sizes)))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "multi_radius_search_count: null argument");
    ctx->scratch.reset();
    return asr_geom_radius_count(ctx, frame, points, n, centers, sizes, v, row_splits_out, num_pairs);
}
int asr_hip_multi_radius_search_fill(asr_hip_context* ctx, const float* points, const float* radii,
                                     int64_t n, const float* centers, const float* sizes,
                                     int64_t v, const int64_t* row_splits, int32_t* index_out,
                                     float* dist_out, float* compat_out) {
    CTX_GUARD(ctx);
    if (v > 0 && (!row_splits || !index_out || !dist_out || (compat_out && !radii)))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "multi_radius_search_fill: null argument");
    return asr_geom_radius_fill(ctx, points, radii, n, centers, sizes, v, row_splits, index_out,
                                dist_out, compat_out);
}

int asr_hip_knn_radius(asr_hip_context* ctx, const asr_octree_frame* frame, const float* points,
                       int64_t n, int k, const float* radii_in, float radius_fraction,
                       int outlier_threshold, float* radii_out, uint8_t* inlier_out) {
    CTX_GUARD(ctx);
    if (!frame || n < 0 || (n > 0 && !points) || (inlier_out && !radii_in))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "knn_radius: null argument");
    ctx->scratch.reset();
    return asr_geom_knn(ctx, frame, points, n, k, radii_in, radius_fraction, outlier_threshold, radii_out,
                        inlier_out);
}
int asr_hip_radius_neighbor_count(asr_hip_context* ctx, const asr_octree_frame* frame,
                                  const float* points, const float* radii, int64_t n,
                                  int64_t* counts_out) {
    CTX_GUARD(ctx);
    if (!frame || n < 0 || (n > 0 && (!points || !radii || !counts_out)))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "radius_neighbor_count: null argument");
    ctx->scratch.reset();
    return asr_geom_radius_neighbor_count(ctx, frame, points, radii, n, counts_out);
}

int asr_hip_continuous_conv_f32(asr_hip_context* ctx, const float* filters, const float* out_pos,
                                const float* extents, const float* inp_pos, const float* inp_feat,
                                const int32_t* nidx, const float* nimp, const int64_t* rs,
                                int64_t num_out, int cin, int cout, int normalize,
                                const float* bias, int relu, float* out) {
    CTX_GUARD(ctx);
    if (num_out > 0 && (!filters || !out_pos || !extents || !inp_pos || !inp_feat || !rs || !out))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "continuous_conv: null argument");
    ctx->scratch.reset();
    return asr_conv_cconv(ctx, filters, out_pos, extents, inp_pos, inp_feat, nidx, nimp, rs, num_out,
                          cin, cout, normalize, bias, relu, out);
}
int asr_hip_continuous_conv_basis_f32(asr_hip_context* ctx, const float* out_pos, const float* extents,
                                      const float* inp_pos, const float* inp_feat, const int32_t* nidx,
                                      const float* nimp, const int64_t* rs, int64_t num_out, int cin, float* basis_out,
                                      float* norm_out) {
    CTX_GUARD(ctx);
    if (cin != 4) ASR_FAIL(ctx, ASR_HIP_EINVAL, "continuous_conv_basis: cin must be 4");
    if (num_out > 0 && (!out_pos || !extents || !inp_pos || !inp_feat || !rs || !basis_out || !norm_out))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "continuous_conv_basis: null argument");
    return asr_conv_cconv_basis(ctx, out_pos, extents, inp_pos, inp_feat, nidx, nimp, rs, num_out, basis_out, norm_out);
}
int asr_hip_aggregation_importance(asr_hip_context* ctx, const float* compat, const float* dist,
                                   int64_t n, float* out) {
    CTX_GUARD(ctx);
    if (n > 0 && (!compat || !dist || !out))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "aggregation_importance: null argument");
    return asr_conv_agg_importance(ctx, compat, dist, n, out);
}
int asr_hip_sparse_conv_f32(asr_hip_context* ctx, const asr_sparse_conv_args* a) {
    CTX_GUARD(ctx);
    if (!a) ASR_FAIL(ctx, ASR_HIP_EINVAL, "sparse_conv: null args");
    if (a->num_out > 0 && (!a->filters || !a->inp_features || !a->neighbors_index ||
                           !a->neighbors_kernel_index || !a->neighbors_row_splits || !a->out))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "sparse_conv: null argument");
    return asr_conv_sparse(ctx, a);
}
size_t asr_hip_sparse_conv_packed_bytes(int mode, int K, int cin, int cout, int cout_b) {
    if ((mode != ASR_CONV16_F16 && mode != ASR_CONV16_BF16X3 && mode != ASR_CONV16_F16X2) || K < 1 || cin < 1 || cout < 1 ||
        cout_b < 0)
        return 0;
    return asr_conv16_packed_bytes(mode, K, cin, cout, cout_b);
}
int asr_hip_sparse_conv_pack(asr_hip_context* ctx, int mode, const float* filters, const float* filters_b, int K,
                             int cin, int cout, int cout_b, void* packed_out) {
    CTX_GUARD(ctx);
    return asr_conv16_pack(ctx, mode, filters, filters_b, K, cin, cout, cout_b, packed_out);
}
static int check_conv16_args(asr_hip_context* ctx, const asr_sparse_conv_args* a, const void* packed) {
    if (!a || !packed) ASR_FAIL(ctx, ASR_HIP_EINVAL, "sparse_conv16: null args");
    if (a->num_out > 0 && (!a->inp_features || !a->neighbors_index || !a->neighbors_kernel_index ||
                           !a->neighbors_row_splits || !a->out))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "sparse_conv16: null argument");
    return ASR_HIP_OK;
}
struct asr_hip_conv_plan {
    Arena mem;  // empty when the plan lives in the context's plan arena
    asr_conv_plan p;
    // plans from the context arena die with asr_hip_context_plan_arena_reset: they remember the arena generation they
    // were taken from and the 16-bit entry points refuse them afterwards (instead of convolving with recycled memory)
    const asr_hip_context* arena_ctx = nullptr;
    uint64_t arena_epoch = 0;
};
int asr_hip_sparse_conv_plan_create(asr_hip_context* ctx, const int32_t* nidx, const uint8_t* kidx, const int64_t* rs,
                                    const int32_t* row_perm, int64_t num_out, int kernel_size,
                                    asr_hip_conv_plan** plan_out) {
    CTX_GUARD(ctx);
    if (!plan_out) ASR_FAIL(ctx, ASR_HIP_EINVAL, "sparse_conv_plan_create: null plan_out");
    *plan_out = nullptr;
    if (num_out < 0 || kernel_size < 1 || kernel_size > 56 || (num_out > 0 && (!nidx || !kidx || !rs)))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "sparse_conv_plan_create: bad argument");
    asr_hip_conv_plan* pl = new asr_hip_conv_plan();
    pl->mem.min_slab = size_t(1) << 20;
    ctx->scratch.reset();
    Arena& where = ctx->opt.plan_arena ? ctx->plan_arena : pl->mem;
    if (ctx->opt.plan_arena) {
        pl->arena_ctx = ctx;
        pl->arena_epoch = ctx->plan_arena_epoch;
    }
    const int rc = asr_geom_conv_plan_build(ctx, where, nidx, kidx, rs, row_perm, num_out, kernel_size, &pl->p);
    if (rc != ASR_HIP_OK) {
        pl->mem.release();
        delete pl;
        return rc;
    }
    *plan_out = pl;
    return ASR_HIP_OK;
}
void asr_hip_sparse_conv_plan_destroy(asr_hip_conv_plan* plan) {
    if (!plan) return;
    plan->mem.release();
    delete plan;
}
size_t asr_hip_sparse_conv_plan_bytes(const asr_hip_conv_plan* plan) { return plan ? plan->mem.reserved() : 0; }
int asr_hip_context_plan_arena_reset(asr_hip_context* ctx) {
    CTX_GUARD(ctx);
    ctx->plan_arena.reset();
    ++ctx->plan_arena_epoch;
    return ASR_HIP_OK;
}

// 16-bit entry points: the caller's plan, or a temporary one in the scratch arena
static int conv16_entry(asr_hip_context* ctx, const asr_sparse_conv_args* a, const void* packed, int mode, int out_f16) {
    ASR_TRY(check_conv16_args(ctx, a, packed));
    if (a->plan && a->plan->arena_ctx && (a->plan->arena_ctx != ctx || a->plan->arena_epoch != ctx->plan_arena_epoch))
        ASR_FAIL(ctx, ASR_HIP_EINVAL,
                 "sparse_conv16: stale plan -- it was taken from a context plan arena that has been reset since "
                 "(asr_hip_context_plan_arena_reset) or belongs to another context");
    if (a->plan) return asr_conv_sparse16(ctx, a, packed, mode, out_f16, &a->plan->p);
    asr_conv_plan tmp;
    const bool want = ctx->opt.sconv_plan && !a->neighbors_importance && a->num_out > 0 && a->algo != 1;
    if (want) {
        ctx->scratch.reset();
        ASR_TRY(asr_geom_conv_plan_build(ctx, ctx->scratch, a->neighbors_index, a->neighbors_kernel_index,
                                         a->neighbors_row_splits, a->row_perm, a->num_out, a->kernel_size, &tmp));
    }
    return asr_conv_sparse16(ctx, a, packed, mode, out_f16, want ? &tmp : nullptr);
}
int asr_hip_sparse_conv_f16(asr_hip_context* ctx, const asr_sparse_conv_args* a, const void* packed, int out_is_f16) {
    CTX_GUARD(ctx);
    return conv16_entry(ctx, a, packed, ASR_CONV16_F16, out_is_f16 ? 1 : 0);
}
int asr_hip_sparse_conv_bf16x3(asr_hip_context* ctx, const asr_sparse_conv_args* a, const void* packed) {
    CTX_GUARD(ctx);
    return conv16_entry(ctx, a, packed, ASR_CONV16_BF16X3, 0);
}
int asr_hip_sparse_conv_f16x2(asr_hip_context* ctx, const asr_sparse_conv_args* a, const void* packed) {
    CTX_GUARD(ctx);
    return conv16_entry(ctx, a, packed, ASR_CONV16_F16X2, 0);
}
int asr_hip_absmax_f32(asr_hip_context* ctx, const float* x, int64_t rows, int cols, int64_t ld, uint32_t* out) {
    CTX_GUARD(ctx);
    if (!out || rows < 0 || cols < 0 || ld < cols || (rows > 0 && cols > 0 && !x))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "absmax_f32: bad argument");
    return asr_conv16_absmax(ctx, x, rows, cols, ld, out);
}
int asr_hip_convert_f16(asr_hip_context* ctx, const void* in, int64_t n, void* out, int to_f16) {
    CTX_GUARD(ctx);
    if (n > 0 && (!in || !out)) ASR_FAIL(ctx, ASR_HIP_EINVAL, "convert_f16: null argument");
    return asr_conv16_convert(ctx, in, n, out, to_f16);
}
int asr_hip_row_groups(asr_hip_context* ctx, const uint8_t* kidx, const int64_t* rs, int64_t v,
                       int64_t seg, int32_t* perm_out) {
    CTX_GUARD(ctx);
    if (v < 0 || (v > 0 && (!kidx || !rs || !perm_out)))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "row_groups: null argument");
    ctx->scratch.reset();
    return asr_geom_row_groups(ctx, kidx, rs, v, seg > 0 ? seg : ASR_ROW_GROUP_SEGMENT, perm_out, 56);
}
int asr_hip_invert_neighbors_list(asr_hip_context* ctx, int64_t num_points, const int32_t* idx,
                                  const int64_t* rs, int64_t num_rows, const uint8_t* attr,
                                  int32_t* out_idx, int64_t* out_rs, uint8_t* out_attr) {
    CTX_GUARD(ctx);
    if (num_points < 0 || num_rows < 0 || !out_rs || (num_rows > 0 && !rs))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "invert_neighbors_list: null argument");
    ctx->scratch.reset();
    return asr_geom_invert(ctx, num_points, idx, rs, num_rows, attr, out_idx, out_rs, out_attr);
}
int asr_hip_reduce_subarrays_sum(asr_hip_context* ctx, const float* values, const int32_t* gidx,
                                 const int64_t* rs, int64_t rows, float* out) {
    CTX_GUARD(ctx);
    if (rows > 0 && (!values || !rs || !out))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "reduce_subarrays_sum: null argument");
    return asr_conv_reduce(ctx, values, gidx, rs, rows, out);
}
int asr_hip_decode_mlp(asr_hip_context* ctx, const float* code, int64_t v, int c, const float* w1,
                       const float* b1, int h1, const float* w2, const float* b2, int h2,
                       const float* w3, const float* sizes, float* out) {
    CTX_GUARD(ctx);
    if (v > 0 && (!code || !w1 || !b1 || !w2 || !b2 || !w3 || !out))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "decode_mlp: null argument");
    return asr_conv_decode(ctx, code, v, c, w1, b1, h1, w2, b2, h2, w3, sizes, out);
}

}  // extern "C"

// ==========================================================================================
// whole path
// ==========================================================================================
namespace {

// cpp/lib/asr.cpp:168-176: feats = [nx, ny, nz, 1].  Written as 32-byte records {x, y, z, -, nx, ny, nz, 1} in
// Morton order (sorted[s].w = original index of the point at position s): the continuous conv reads position
// and features of a pair from one cache line, and the neighbours of a voxel from adjacent records
__global__ void k_make_feats(const float* normals, const float4* sorted, i64 n, float* rec) {
    i64 s = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (s >= n) return;
    const float4 p = sorted[s];
    const i64 i = __float_as_int(p.w);
    reinterpret_cast<float4*>(rec)[2 * s] = p;
    reinterpret_cast<float4*>(rec)[2 * s + 1] =
            make_float4(normals[3 * i], normals[3 * i + 1], normals[3 * i + 2], 1.f);
}

int ensure_events(asr_hip_context* ctx) {
    if (!ctx->ev_ok) {
        for (auto& e : ctx->ev) ASR_HIP_CHECK(ctx, hipEventCreate(&e));
        ctx->ev_ok = true;
    }
    return ASR_HIP_OK;
}

void name_it(asr_hip_context* ctx, const std::string& name, const void* p, size_t bytes) {
    ctx->named[name] = {p, bytes};
}

struct WeightTable {
    const asr_weight* w;
    int n;
    const asr_weight* find(const std::string& name) const {
        for (int i = 0; i < n; ++i)
            if (w[i].name && name == w[i].name) return &w[i];
        return nullptr;
    }
};

struct Feat {
    float* p;  // f16 data when the network runs with ASR_CONV16_F16 activations (see Net::esz)
    i64 ld;    // row stride in elements
    int c;
    unsigned* amax = nullptr;  // ASR_CONV16_F16X2: device scalar with the f32 bits of the buffer's largest magnitude,
                               // kept by the epilogues of the convolutions that write the buffer (Net::new_amax)
};

struct Net {
    asr_hip_context* ctx;
    WeightTable wt;
    int precision = 0;  // 0 = exact f32 MFMA, ASR_CONV16_F16, ASR_CONV16_BF16X3, ASR_CONV16_F16X2 (asr_implicit_params.precision)
    int num_amax = 0;
    // sharded forward (ctx->shard): importance arrays are valid on the owned rows only and travel with the halo of the
    // features -- except this one, the aggregation's per-pair array, which every rank holds in full (quirk B.2)
    const float* replicated_imp = nullptr;

    // rows / plan of a convolution over the list `rs` and the halo exchange of its input (no-op on one GPU)
    int shard_hook(const i64* rs, Feat in, const float* imp, const int32_t** perm, i64* num_out,
                   const asr_conv_plan** plan) {
        *plan = nullptr;
        if (!ctx->shard) return ASR_HIP_OK;
        float* imp_x = (imp && imp != replicated_imp) ? const_cast<float*>(imp) : nullptr;
        return asr_shard_before_conv(ctx, ctx->shard, rs, in.p, in.ld * (i64)esz(), in.c * (i64)esz(), imp_x, in.amax, perm,
                                     num_out, plan);
    }

    // f16x2: a zeroed device scalar for the running maximum of one activation buffer (nullptr in the other modes)
    unsigned* new_amax() {
        if (precision != ASR_CONV16_F16X2 || !ctx->d_absmax || num_amax >= 255) return nullptr;
        return ctx->d_absmax + num_amax++;
    }
    int begin_amax() {
        num_amax = 0;
        if (precision != ASR_CONV16_F16X2) return ASR_HIP_OK;
        if (!ctx->d_absmax) ASR_HIP_CHECK(ctx, hipMalloc((void**)&ctx->d_absmax, 256 * sizeof(unsigned)));
        ASR_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_absmax, 0, 256 * sizeof(unsigned), ctx->stream));
        return ASR_HIP_OK;
    }

    size_t esz() const { return precision == ASR_CONV16_F16 ? 2 : 4; }  // bytes per activation element
    // activation buffer [rows, c] in the scratch arena (element type per precision)
    float* act(i64 rows, int c) { return (float*)ctx->scratch.alloc((size_t)rows * c * esz()); }
    float* col(float* base, int c) const { return (float*)((char*)base + (size_t)c * esz()); }  // column offset

    // packed 16-bit copy of a (two-bank) filter tensor, made once per weight tensor and mode
    int packed(const asr_weight* ka, const asr_weight* kb, const void** out) {
        const int cb = kb ? (int)kb->shape[2] : 0;
        const asr_hip_context::PackedKey key((const void*)ka->data, kb ? (const void*)kb->data : nullptr, ka->shape[0],
                                             ka->shape[1], ka->shape[2], (i64)cb, precision);
        auto it = ctx->packed_weights.find(key);
        if (it == ctx->packed_weights.end()) {
            const size_t bytes = asr_conv16_packed_bytes(precision, (int)ka->shape[0], (int)ka->shape[1],
                                                         (int)ka->shape[2], cb);
            void* p = nullptr;
            ASR_HIP_CHECK(ctx, hipMalloc(&p, bytes));
            const int rc = asr_conv16_pack(ctx, precision, ka->data, kb ? kb->data : nullptr, (int)ka->shape[0],
                                           (int)ka->shape[1], (int)ka->shape[2], cb, p);
            if (rc != ASR_HIP_OK) {  // never cache a buffer that was not filled
                (void)hipFree(p);
                return rc;
            }
            it = ctx->packed_weights.emplace(key, p).first;
        }
        *out = it->second;
        return ASR_HIP_OK;
    }
    // dispatch of one assembled argument block: f32 kernel or a 16-bit variant; out_f32: the layer's output
    // feeds the decoder (f32 `code`)
    int launch(asr_sparse_conv_args& a, const asr_weight* ka, const asr_weight* kb, bool out_f32,
               const asr_conv_plan* shard_plan = nullptr) {
        if (precision == 0) return asr_conv_sparse(ctx, &a);
        const void* pk = nullptr;
        ASR_TRY(packed(ka, kb, &pk));
        const asr_conv_plan* plan = shard_plan;
        if (!ctx->shard) {
            auto pl = ctx->conv_plans.find(a.neighbors_row_splits);
            plan = pl == ctx->conv_plans.end() ? nullptr : &pl->second;
        }
        return asr_conv_sparse16(ctx, &a, pk, precision, precision == ASR_CONV16_F16 && !out_f32, plan);
    }

    int get(const std::string& name, int ndim, const asr_weight** out) {
        const asr_weight* w = wt.find(name);
        if (!w || !w->data) ASR_FAIL(ctx, ASR_HIP_EWEIGHT, "missing weight '%s'", name.c_str());
        if (w->ndim != ndim)
            ASR_FAIL(ctx, ASR_HIP_EWEIGHT, "weight '%s' has rank %d, expected %d", name.c_str(),
                     w->ndim, ndim);
        *out = w;
        return ASR_HIP_OK;
    }

    // one SpecialSparseConv (+bias, ReLU): models/common_torch.py:95-148
    int conv(const std::string& prefix, int K, Feat in, const int32_t* nidx, const uint8_t* nk,
             const i64* rs, const int32_t* perm, i64 num_out, i64 num_inp, const float* imp,
             int normalize, float* out, i64 out_ld, int expect_cout, float* out_imp,
             const float* residual, i64 residual_ld, bool out_f32 = false, unsigned* out_amax = nullptr) {
        const asr_weight *k, *b;
        ASR_TRY(get(prefix + ".kernel", 3, &k));
        ASR_TRY(get(prefix + ".bias", 1, &b));
        if (k->shape[0] != K || k->shape[1] != in.c || (expect_cout > 0 && k->shape[2] != expect_cout) ||
            b->shape[0] != k->shape[2])
            ASR_FAIL(ctx, ASR_HIP_EWEIGHT, "weight '%s.kernel' has shape [%lld,%lld,%lld], expected [%d,%d,%d]",
                     prefix.c_str(), (long long)k->shape[0], (long long)k->shape[1],
                     (long long)k->shape[2], K, in.c, expect_cout);
        const asr_conv_plan* shard_plan = nullptr;
        ASR_TRY(shard_hook(rs, in, imp, &perm, &num_out, &shard_plan));
        asr_sparse_conv_args a;
        memset(&a, 0, sizeof(a));
        a.filters = k->data;
        a.inp_features = in.p;
        a.inp_ld = in.ld;
        a.inp_importance = imp;
        a.neighbors_index = nidx;
        a.neighbors_kernel_index = nk;
        a.neighbors_row_splits = rs;
        a.num_out = num_out;
        a.num_inp = num_inp;
        a.kernel_size = K;
        a.cin = in.c;
        a.cout = (int)k->shape[2];
        a.normalize = normalize;
        a.bias = b->data;
        a.relu = 1;
        a.residual = residual;
        a.residual_ld = residual_ld;
        a.out = out;
        a.out_ld = out_ld;
        a.out_importance = out_imp;
        a.row_perm = perm;
        a.inp_absmax = in.amax;
        a.out_absmax = out_amax;
        return launch(a, k, nullptr, out_f32, shard_plan);
    }
    // conv1a + conv1b of a block in one launch (second filter bank of asr_sparse_conv_args): same
    // gather, one extra column tile.  Falls back to two launches for widths the fused kernel does not take.
    int conv_ab(const std::string& name, int K, Feat in, const int32_t* nidx, const uint8_t* nk, const i64* rs,
                const int32_t* perm, i64 num_out, i64 num_inp, const float* imp, float* out, i64 out_ld, int ca,
                int cb, float* out_imp, unsigned* out_amax) {
        const asr_weight *ka, *ba, *kb, *bb;
        ASR_TRY(get(name + ".conv1a.kernel", 3, &ka));
        ASR_TRY(get(name + ".conv1a.bias", 1, &ba));
        ASR_TRY(get(name + ".conv1b.kernel", 3, &kb));
        ASR_TRY(get(name + ".conv1b.bias", 1, &bb));
        const bool fused = ca % 16 == 8 && cb == 8 && in.c % 8 == 0 && in.ld % 8 == 0 && ka->shape[0] == K &&
                           kb->shape[0] == K && ka->shape[1] == in.c && kb->shape[1] == in.c &&
                           ka->shape[2] == ca && kb->shape[2] == cb && ba->shape[0] == ca && bb->shape[0] == cb &&
                           (uintptr_t)kb->data % 16 == 0 && (uintptr_t)ka->data % 16 == 0 &&
                           (uintptr_t)in.p % 16 == 0;
        if (!fused) {
            ASR_TRY(conv(name + ".conv1a", K, in, nidx, nk, rs, perm, num_out, num_inp, nullptr, 0, out, out_ld, ca,
                         nullptr, nullptr, 0, false, out_amax));
            return conv(name + ".conv1b", K, in, nidx, nk, rs, perm, num_out, num_inp, imp, 1, col(out, ca), out_ld,
                        cb, out_imp, nullptr, 0, false, out_amax);
        }
        const asr_conv_plan* shard_plan = nullptr;
        ASR_TRY(shard_hook(rs, in, imp, &perm, &num_out, &shard_plan));
        asr_sparse_conv_args a;
        memset(&a, 0, sizeof(a));
        a.filters = ka->data;
        a.filters_b = kb->data;
        a.inp_features = in.p;
        a.inp_ld = in.ld;
        a.inp_importance = imp;
        a.neighbors_index = nidx;
        a.neighbors_kernel_index = nk;
        a.neighbors_row_splits = rs;
        a.num_out = num_out;
        a.num_inp = num_inp;
        a.kernel_size = K;
        a.cin = in.c;
        a.cout = ca;
        a.cout_b = cb;
        a.normalize = 1;
        a.bias = ba->data;
        a.bias_b = bb->data;
        a.relu = 1;
        a.out = out;
        a.out_ld = out_ld;
        a.out_importance = out_imp;
        a.row_perm = perm;
        a.algo = 2;
        a.inp_absmax = in.amax;
        a.out_absmax = out_amax;
        return launch(a, ka, kb, false, shard_plan);
    }
    int cout_of(const std::string& prefix, int* c) {
        const asr_weight* k;
        ASR_TRY(get(prefix + ".kernel", 3, &k));
        *c = (int)k->shape[2];
        return ASR_HIP_OK;
    }

    // SparseConvBlock with conv1a/conv1b (normalized_channels < output_channels) or plain
    // conv1 (decoder blocks): net_definitions_torch.py:253-302.  `out` receives conv4.
    int block(const std::string& name, Feat in, const GridDev& g, const float* imp, bool with_imp,
              Feat out, float** out_imp, bool out_f32 = false) {
        asr_hip_context* c = ctx;
        int C = out.c;
        float* t1 = act(g.v, C);
        float* t2 = act(g.v, C);
        if (!t1 || !t2) ASR_FAIL(c, ASR_HIP_EHIP, "arena allocation failed");
        int ca_ = 0, cb_ = 0;
        float* oi_ = nullptr;
        if (with_imp) {
            ASR_TRY(cout_of(name + ".conv1a", &ca_));
            ASR_TRY(cout_of(name + ".conv1b", &cb_));
            if (ca_ + cb_ != C) ASR_FAIL(c, ASR_HIP_EWEIGHT, "%s: conv1a+conv1b != block width", name.c_str());
            oi_ = arena_alloc<float>(c->scratch, g.v);
            if (!oi_) ASR_FAIL(c, ASR_HIP_EHIP, "arena allocation failed");
        }
        // t1 is written twice (conv1, conv3): one running maximum per use
        Feat f1{t1, C, C, new_amax()}, f2{t2, C, C, new_amax()}, f3{t1, C, C, new_amax()};
        if (with_imp) {
            ASR_TRY(conv_ab(name, 55, in, g.nidx, g.nkidx, g.nrs, g.perm_nb, g.v, g.v, imp, t1, C, ca_, cb_, oi_, f1.amax));
            *out_imp = oi_;
        } else {
            ASR_TRY(conv(name + ".conv1", 55, in, g.nidx, g.nkidx, g.nrs, g.perm_nb, g.v, g.v, nullptr, 0, t1, C,
                         C, nullptr, nullptr, 0, false, f1.amax));
        }
        ASR_TRY(conv(name + ".conv2", 55, f1, g.nidx, g.nkidx, g.nrs, g.perm_nb, g.v, g.v, nullptr, 0, t2, C, C,
                     nullptr, nullptr, 0, false, f2.amax));
        ASR_TRY(conv(name + ".conv3", 55, f2, g.nidx, g.nkidx, g.nrs, g.perm_nb, g.v, g.v, nullptr, 0, t1, C, C,
                     nullptr, nullptr, 0, false, f3.amax));
        ASR_TRY(conv(name + ".conv4", 55, f3, g.nidx, g.nkidx, g.nrs, g.perm_nb, g.v, g.v, nullptr, 0, out.p,
                     out.ld, C, nullptr, nullptr, 0, out_f32, out.amax));
        return ASR_HIP_OK;
    }
    // SparseConvTransitionBlock down (conv1a/conv1b): net_definitions_torch.py:357-387
    int down(const std::string& name, Feat in, const GridDev& fine, const GridDev& coarse,
             const float* imp, Feat out, float** out_imp) {
        int ca, cb;
        ASR_TRY(cout_of(name + ".conv1a", &ca));
        ASR_TRY(cout_of(name + ".conv1b", &cb));
        if (ca + cb != out.c) ASR_FAIL(ctx, ASR_HIP_EWEIGHT, "%s: conv1a+conv1b != width", name.c_str());
        float* oi = arena_alloc<float>(ctx->scratch, coarse.v);
        if (!oi) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
        ASR_TRY(conv_ab(name, 9, in, fine.down_idx, fine.down_kidx, fine.down_rs, fine.perm_down, coarse.v, fine.v,
                        imp, out.p, out.ld, ca, cb, oi, out.amax));
        *out_imp = oi;
        return ASR_HIP_OK;
    }
    // up transition (conv1): rows = fine voxels, inputs from the coarse grid
    int up(const std::string& name, Feat in, const GridDev& fine, const GridDev& coarse, Feat out,
           const float* residual, i64 residual_ld) {
        return conv(name + ".conv1", 9, in, fine.up_idx, fine.up_kidx, fine.up_rs, fine.perm_up, fine.v, coarse.v,
                    nullptr, 0, out.p, out.ld, out.c, nullptr, residual, residual_ld, false, out.amax);
    }
};

}  // namespace
// The auxiliary context of the aggregation search: its own stream, arenas, counters.  Stream priority (option
// "search_priority", fixed when the stream is made): highest by default -- the search is the longer of the two chains of the
// geometry build (round 4: 9.6 -> 8.8 ms on its stream); it used to be lowest so that its large grids would not starve the
// small kernels of the main chain.  HIP maps streams to a small number of hardware queues (GPU_MAX_HW_QUEUES, default 4)
// in creation order: a stream made after a library such as RCCL has made its own may share the main stream's queue, and
// the two chains then run one after the other (geometry 13.5 instead of 9.5 ms at 10 M points).  asr_hip_shard_comm_rccl_create
// therefore makes this stream first; processes that initialise RCCL before they create a context should raise
// GPU_MAX_HW_QUEUES to 8 (bench.py does).
int asr_ctx_ensure_aux(asr_hip_context* ctx) {
    if (ctx->aux) return ASR_HIP_OK;
    ASR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ctx->aux = new asr_hip_context();
    ctx->aux->device = ctx->device;
    memset(&ctx->aux->sizes, 0, sizeof(ctx->aux->sizes));
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    const int prio = ctx->opt.search_priority >= 2 ? prio_greatest
                     : (ctx->opt.search_priority == 1 ? (prio_least + prio_greatest) / 2 : prio_least);
    ASR_HIP_CHECK(ctx, hipStreamCreateWithPriority(&ctx->aux->stream, hipStreamNonBlocking, prio));
    ctx->aux_stream_owned = true;
    ASR_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->aux_ev, hipEventDisableTiming));
    ASR_HIP_CHECK(ctx, hipEventCreate(&ctx->aux_t0));
    ASR_HIP_CHECK(ctx, hipEventCreate(&ctx->aux_t1));
    ctx->aux->opt = ctx->opt;
    apply_arena_cap(ctx);
    return ASR_HIP_OK;
}

namespace {
// MFMA tiling orders of the 13 neighbour lists (one batched regrouping) and, for the 16-bit kernels, their row-group plans.
// `on`: the context whose stream / scratch arena / flags do the work (the grids' own context, or its auxiliary context when
// the step runs beside the continuous conv); plans land in `on`'s persist arena, the orders in the arrays implicit_build
// allocated.
int build_tilings_and_plans(asr_hip_context* ctx, asr_hip_context* on, int precision) {
    std::vector<asr_row_group_job> rg_jobs;
    for (int i = 0; i + 1 < ASR_NUM_GRIDS; ++i) {
        GridDev& g = ctx->grids[i];
        rg_jobs.push_back({g.up_kidx, g.up_rs, g.v, 9, g.perm_up});
        rg_jobs.push_back({g.down_kidx, g.down_rs, ctx->grids[i + 1].v, 9, g.perm_down});
    }
    for (int i = 0; i < ASR_NUM_GRIDS; ++i) {
        GridDev& g = ctx->grids[i];
        rg_jobs.push_back({g.nkidx, g.nrs, g.v, 55, g.perm_nb});
    }
    on->scratch.reset();
    if (asr_geom_row_groups_batch(on, rg_jobs.data(), (int)rg_jobs.size(), ASR_ROW_GROUP_SEGMENT) != ASR_HIP_OK) {
        if (on != ctx) ctx->err = on->err;
        return ASR_HIP_EHIP;
    }
    if (precision != 0 && ctx->opt.sconv_plan) {
        struct Job {
            const int32_t* idx;
            const uint8_t* kidx;
            const i64* rs;
            const int32_t* perm;
            i64 rows;
            int K;
        };
        std::vector<Job> jobs;
        for (int i = 0; i < ASR_NUM_GRIDS; ++i) {
            GridDev& g = ctx->grids[i];
            jobs.push_back({g.nidx, g.nkidx, g.nrs, g.perm_nb, g.v, 55});
            if (i + 1 < ASR_NUM_GRIDS) {
                GridDev& c = ctx->grids[i + 1];
                jobs.push_back({g.up_idx, g.up_kidx, g.up_rs, g.perm_up, g.v, 9});
                jobs.push_back({g.down_idx, g.down_kidx, g.down_rs, g.perm_down, c.v, 9});
            }
        }
        std::vector<asr_conv_plan> plans(jobs.size());
        for (size_t j = 0; j < jobs.size(); ++j) {
            plans[j].nidx = jobs[j].idx;
            plans[j].kidx = jobs[j].kidx;
            plans[j].rs = jobs[j].rs;
            plans[j].perm = jobs[j].perm;
            plans[j].num_out = jobs[j].rows;
            plans[j].K = jobs[j].K;
        }
        const int rc = asr_geom_conv_plan_batch(on, on->persist, plans.data(), (int)plans.size());
        if (rc != ASR_HIP_OK) {
            if (on != ctx) ctx->err = on->err;
            return rc;
        }
        for (size_t j = 0; j < jobs.size(); ++j) ctx->conv_plans[jobs[j].rs] = plans[j];
    }
    return ASR_HIP_OK;
}

// gathers of the compact query list of a rank with sharded geometry, and the scatter of its aggregated rows
__global__ void k_gather_queries(const int32_t* rows, i64 nq, const float* centers, const float* sizes, const u64* keys,
                                 float* qc, float* qs, u64* qk) {
    const i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const i64 r = rows[i];
    qc[3 * i] = centers[3 * r];
    qc[3 * i + 1] = centers[3 * r + 1];
    qc[3 * i + 2] = centers[3 * r + 2];
    qs[i] = sizes[r];
    qk[i] = keys[r];
}
__global__ void k_scatter_rows(const float* src, const int32_t* rows, i64 nq, int c, float* dst) {
    const i64 t = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (t >= nq * c) return;
    const i64 i = t / c;
    dst[(i64)rows[i] * c + (t - i * c)] = src[t];
}

// what every build starts with: nothing of the previous build survives
int build_begin(asr_hip_context* ctx, i64 n, const asr_implicit_params* prm) {
    ASR_TRY(ensure_events(ctx));
    ctx->persist.reset();
    ctx->scratch.reset();
    ctx->named.clear();
    ctx->values = ctx->feats1 = ctx->importance = ctx->code = nullptr;
    ctx->agg_rs = nullptr;
    ctx->agg_idx = ctx->agg_spos = nullptr;
    ctx->agg_dist = ctx->agg_compat = nullptr;
    ctx->agg_sorted = nullptr;
    ctx->agg_rows = nullptr;
    ctx->agg_nq = 0;
    ctx->agg_qcenters = ctx->agg_qsizes = nullptr;
    ctx->has_search = false;
    ctx->build_sharded = false;
    ctx->build_mark_ok = false;
    ctx->conv_plans.clear();
    memset(&ctx->sizes, 0, sizeof(ctx->sizes));
    ctx->sizes.num_points = n;
    if (asr_octree_frame_init(&ctx->frame, prm->bb_min, prm->bb_max) != ASR_HIP_OK)
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "degenerate bounding box");
    asr_hip_print("grid building\n", ASR_HIP_INFO);  // cpp/lib/asr.cpp:144
    ASR_HIP_CHECK(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    return ASR_HIP_OK;
}
// centres / sizes of the level-0 voxels (= the octree's leaves)
int build_level0(asr_hip_context* ctx) {
    GridDev& g0 = ctx->grids[0];
    g0 = GridDev();
    g0.v = ctx->num_leaves;
    g0.keys = ctx->leaves;
    g0.centers = arena_alloc<float>(ctx->persist, 3 * g0.v);
    g0.sizes = arena_alloc<float>(ctx->persist, g0.v);
    if (!g0.centers || !g0.sizes) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    return asr_geom_voxel_info(ctx, &ctx->frame, g0.keys, g0.v, g0.centers, g0.sizes);
}
// key sets of the coarser grids with their up / down lists: a chain of four small coarsening steps
// (cpp/lib/grid.cpp:245-314; CombineSiblings + neighbors_down = invert(up lists), net_definitions_torch.py:548-559)
int build_coarse_grids(asr_hip_context* ctx) {
    for (int i = 1; i < ASR_NUM_GRIDS; ++i) {
        GridDev& g = ctx->grids[i];
        g = GridDev();
        ctx->scratch.reset();
        GridDev& prev = ctx->grids[i - 1];
        // (the keys of every grid are location codes of at most the deepest leaf's level: the sort skips the rest)
        ASR_TRY(asr_geom_coarsen_build(ctx, ctx->persist, prev.keys, prev.v, &g.keys, &g.v, &prev.up_idx,
                                       &prev.up_kidx, &prev.up_rs, &prev.down_idx, &prev.down_kidx, &prev.down_rs,
                                       ctx->leaf_lmax >= 0 ? 3 * ctx->leaf_lmax + 1 : 64));
        prev.perm_up = arena_alloc<int32_t>(ctx->persist, prev.v);
        prev.perm_down = arena_alloc<int32_t>(ctx->persist, g.v);
        if (!prev.perm_up || !prev.perm_down) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
        std::string s = std::to_string(i - 1);
        name_it(ctx, "up_neighbors_index" + s, prev.up_idx, 4 * prev.v);
        name_it(ctx, "up_neighbors_kernel_index" + s, prev.up_kidx, prev.v);
        name_it(ctx, "up_neighbors_row_splits" + s, prev.up_rs, 8 * (prev.v + 1));
        name_it(ctx, "down_neighbors_index" + s, prev.down_idx, 4 * prev.v);
        name_it(ctx, "down_neighbors_kernel_index" + s, prev.down_kidx, prev.v);
        name_it(ctx, "down_neighbors_row_splits" + s, prev.down_rs, 8 * (g.v + 1));
        name_it(ctx, "tiling_up" + s, prev.perm_up, 4 * prev.v);
        name_it(ctx, "tiling_down" + s, prev.perm_down, 4 * g.v);
        g.centers = arena_alloc<float>(ctx->persist, 3 * g.v);
        g.sizes = arena_alloc<float>(ctx->persist, g.v);
        if (!g.centers || !g.sizes) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
        ASR_TRY(asr_geom_voxel_info(ctx, &ctx->frame, g.keys, g.v, g.centers, g.sizes));
    }
    return ASR_HIP_OK;
}
// tiling-order arrays + the names of the per-grid arrays, once the 55-slot lists exist
int build_grid_names(asr_hip_context* ctx) {
    for (int i = 0; i < ASR_NUM_GRIDS; ++i) {
        GridDev& g = ctx->grids[i];
        g.perm_nb = arena_alloc<int32_t>(ctx->persist, g.v);
        if (!g.perm_nb) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
        ctx->sizes.num_voxels[i] = g.v;
        ctx->sizes.num_pairs[i] = g.p;
        std::string s = std::to_string(i);
        name_it(ctx, "voxel_keys" + s, g.keys, 8 * g.v);
        name_it(ctx, "voxel_centers" + s, g.centers, 12 * g.v);
        name_it(ctx, "voxel_sizes" + s, g.sizes, 4 * g.v);
        name_it(ctx, "neighbors_index" + s, g.nidx, 4 * g.p);
        name_it(ctx, "neighbors_kernel_index" + s, g.nkidx, g.p);
        name_it(ctx, "neighbors_row_splits" + s, g.nrs, 8 * (g.v + 1));
        name_it(ctx, "tiling" + s, g.perm_nb, 4 * g.v);
    }
    return ASR_HIP_OK;
}

// comm / st_out (both or neither): the build of ONE RANK of a cloud cut into Morton ranges (option
// shard_geometry): octree, voxel keys and the one-entry-per-voxel up / down lists of all five grids on every rank -- cheap
// integer work that ownership is derived from -- and the expensive parts for the voxels this rank owns only: 55-slot
// neighbour lists, row-group plans, the aggregation search (+ the importance prefix, see the search below).
int implicit_build(asr_hip_context* ctx, const float* points, const float* radii, i64 n,
                   const asr_implicit_params* prm, const asr_shard_comm* comm = nullptr,
                   asr_shard_state** st_out = nullptr) {
    asr_shard_state* st = nullptr;
    struct ShardGuard {  // the state is the caller's only after a complete build
        asr_shard_state*& st;
        bool keep = false;
        ~ShardGuard() {
            if (!keep && st) asr_shard_free(st);
        }
    } shard_guard{st};
    ASR_TRY(build_begin(ctx, n, prm));
    // Aggregation neighbours (cpp/lib/asr.cpp:266-273) on the auxiliary context: its own stream, arenas,
    // counters and host thread, overlapped with the grid hierarchy below.  Both are chains of
    // latency-bound kernels with host round trips for the data-dependent sizes; neither fills the GPU.
    const bool want_search = ctx->opt.build_search != 0;
    const bool overlap = want_search && ctx->opt.overlap != 0;
    asr_hip_context* sc = ctx;  // context the search runs on
    if (overlap) {
        ASR_TRY(asr_ctx_ensure_aux(ctx));
        sc = ctx->aux;
        sc->persist.reset();
        sc->scratch.reset();
        sc->err.clear();
        sc->pindex.valid = false;
        ASR_HIP_CHECK(ctx, hipEventRecord(ctx->aux_ev, ctx->stream));  // the inputs are ready
        ASR_HIP_CHECK(ctx, hipStreamWaitEvent(sc->stream, ctx->aux_ev, 0));
        ASR_HIP_CHECK(ctx, hipEventRecord(ctx->aux_t0, sc->stream));  // the search is timed on ITS stream
        // The search's point sort needs nothing from the octree (it sorts deep enough for any realistic leaf level,
        // PRESORT_LEVEL): enqueued on the auxiliary stream now, it runs beside the octree construction.
        // (the sort itself is enqueued by the search thread below: it reads one number back first -- the finest level a point
        // can be inserted on, which sets the depth of the sort -- and this thread goes on to the octree meanwhile)
    }
    // The search thread starts here: first the cell table of the sorted points (host round trips of its own, beside
    // the octree construction on this thread), then it waits for the level-0 voxels.
    std::promise<bool> level0_ready;
    std::future<bool> level0_future = level0_ready.get_future();
    struct Release {  // an early return of this function must not leave the thread waiting
        std::promise<bool>& p;
        bool done = false;
        void set(bool v) {
            if (!done) p.set_value(v);
            done = true;
        }
        ~Release() { set(false); }
    };
    i64 agg_pairs = 0;
    int search_rc = ASR_HIP_OK;
    std::function<int()> search;  // defined below, once the names it uses exist
    std::thread worker;
    struct Joiner {  // the code below returns early on errors: never leave the thread running
        std::thread& t;
        ~Joiner() {
            if (t.joinable()) t.join();
        }
    } joiner{worker};
    Release release{level0_ready};  // (destroyed before the joiner: the thread is released, then joined)
    if (overlap) {
        worker = std::thread([&] {
            if (hipSetDevice(ctx->device) != hipSuccess) {
                sc->err = "hipSetDevice failed in the search thread";
                search_rc = ASR_HIP_EHIP;
                (void)level0_future.get();
                return;
            }
            int rc = ASR_HIP_OK;
            if (ctx->opt.early_sort)
                rc = asr_geom_presort(sc, sc->persist, &ctx->frame, points, radii, n, prm->point_radius_scale, prm->octree_max_depth);
            if (rc == ASR_HIP_OK && ctx->opt.early_sort && ctx->opt.early_cells && sc->pindex.valid)
                rc = asr_geom_precells(sc, sc->persist);
            const bool go = level0_future.get();
            if (rc != ASR_HIP_OK || !go) {
                search_rc = rc;
                return;
            }
            search_rc = search();
        });
    }
    ctx->pindex.valid = false;
    ASR_TRY(asr_geom_octree_build(ctx, &ctx->frame, points, radii, n, prm->point_radius_scale, prm->octree_max_depth));
    ctx->sizes.num_nodes = ctx->num_nodes;
    name_it(ctx, "nodes", ctx->nodes, 8 * ctx->num_nodes);
    ASR_HIP_CHECK(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    if (ctx->num_leaves == 0) ASR_FAIL(ctx, ASR_HIP_EINVAL, "no point inside the bounding box");

    // level-0 voxel centres / sizes first: the aggregation search only needs those
    ASR_TRY(build_level0(ctx));
    if (comm) ASR_TRY(asr_shard_ownership(ctx, comm, 0, &st));  // grid 0 by voxel count: no list exists yet

    if (want_search) asr_hip_print("aggregate\n", ASR_HIP_INFO);  // cpp/lib/asr.cpp:264 (here: concurrent with the grids)
    if (overlap) ASR_HIP_CHECK(ctx, hipEventRecord(ctx->aux_ev, ctx->stream));  // the level-0 voxel centres / sizes are ready
    ctx->search_overlapped = overlap;
    search = [&]() -> int {
        if (sc != ctx) {
            ASR_HIP_CHECK(sc, hipStreamWaitEvent(sc->stream, ctx->aux_ev, 0));
        } else {
            ctx->scratch.reset();
        }
        GridDev& g0 = ctx->grids[0];
        const float* qc = g0.centers;
        const float* qs = g0.sizes;
        const u64* qk = g0.keys;
        i64 nq = g0.v;
        if (st && asr_shard_world(st) > 1) {
            // One rank of a sharded cloud searches [0, prefix) + the rows it owns.  SURVEY B.2: encblock0 reads the
            // importance of the first V0 PAIRS of the whole cloud's CSR; those belong to the first few percent of the
            // voxels in index order, which every rank therefore searches itself (no communication).
            ArenaMark before;
            arena_mark(sc->persist, before);
            // (the first V0 pairs sit in the first few thousand rows -- the coarsest voxels have the longest rows; the number
            // that sufficed last time is the next build's first guess)
            i64 prefix = std::min<i64>(g0.v, ctx->shard_prefix_hint > 0 ? ctx->shard_prefix_hint : std::max<i64>(1024, g0.v / 256));
            for (;;) {
                arena_rewind(sc->persist, before);
                int32_t* qrows = nullptr;
                ASR_TRY(asr_shard_query_rows(sc, st, prefix, sc->persist, &qrows, &nq));
                float* c = arena_alloc<float>(sc->persist, 3 * nq);
                float* z = arena_alloc<float>(sc->persist, nq);
                u64* k = arena_alloc<u64>(sc->persist, nq);
                ctx->agg_rs = arena_alloc<i64>(sc->persist, nq + 1);
                if (!c || !z || !k || !ctx->agg_rs) ASR_FAIL(sc, ASR_HIP_EHIP, "arena allocation failed");
                k_gather_queries<<<grid_for(nq, 256), 256, 0, sc->stream>>>(qrows, nq, g0.centers, g0.sizes, g0.keys, c, z, k);
                ASR_CHECK_LAUNCH(sc);
                ASR_TRY(asr_geom_radius_count(sc, &ctx->frame, points, n, c, z, nq, ctx->agg_rs, &agg_pairs, &sc->persist,
                                              radii, k, ctx->leaf_lmin, ctx->leaf_lmax, sc->pindex.valid ? &sc->pindex : nullptr));
                i64 prefix_pairs = 0;
                ASR_HIP_CHECK(sc, hipMemcpyAsync(&prefix_pairs, ctx->agg_rs + prefix, sizeof(i64), hipMemcpyDeviceToHost, sc->stream));
                ASR_HIP_CHECK(sc, hipStreamSynchronize(sc->stream));
                if (prefix_pairs < g0.v && prefix < g0.v) {
                    prefix = std::min<i64>(g0.v, 4 * prefix);
                    continue;
                }
                ctx->shard_prefix_hint = prefix;
                // (prefix == V0 and fewer pairs than voxels: implicit_network reports the reference's out-of-range indexing)
                ctx->agg_rows = qrows;
                ctx->agg_nq = nq;
                ctx->agg_qcenters = qc = c;
                ctx->agg_qsizes = qs = z;
                break;
            }
        } else {
            ctx->agg_rs = arena_alloc<i64>(sc->persist, g0.v + 1);
            if (!ctx->agg_rs) ASR_FAIL(sc, ASR_HIP_EHIP, "arena allocation failed");
            ASR_TRY(asr_geom_radius_count(sc, &ctx->frame, points, n, g0.centers, g0.sizes, g0.v, ctx->agg_rs,
                                          &agg_pairs, &sc->persist, radii, g0.keys, ctx->leaf_lmin, ctx->leaf_lmax,
                                          sc->pindex.valid ? &sc->pindex : nullptr));
        }
        ctx->agg_idx = arena_alloc<int32_t>(sc->persist, agg_pairs);
        ctx->agg_dist = arena_alloc<float>(sc->persist, agg_pairs);
        ctx->agg_compat = arena_alloc<float>(sc->persist, agg_pairs);
        ctx->agg_spos = arena_alloc<int32_t>(sc->persist, agg_pairs);
        if (!ctx->agg_idx || !ctx->agg_dist || !ctx->agg_compat || !ctx->agg_spos)
            ASR_FAIL(sc, ASR_HIP_EHIP, "arena allocation failed");
        ASR_TRY(asr_geom_radius_fill(sc, points, radii, n, qc, qs, nq, ctx->agg_rs, ctx->agg_idx,
                                     ctx->agg_dist, ctx->agg_compat, ctx->agg_spos, &ctx->agg_sorted));
        (void)qk;
        if (sc != ctx) {
            ASR_HIP_CHECK(sc, hipEventRecord(ctx->aux_t1, sc->stream));
            ASR_HIP_CHECK(sc, hipStreamSynchronize(sc->stream));
        }
        return ASR_HIP_OK;
    };
    release.set(true);  // the search thread goes on

    // grids (cpp/lib/grid.cpp:245-314).  First the key sets of the coarser grids with their up / down lists (a chain of
    // four small coarsening steps), then the 55-slot neighbour lists of ALL five grids in one batch
    // (asr_geom_neighbors_build_batch: one key-map launch, one counting pass, one scan, one read-back, one filling pass
    // -- the coarse grids used to pay a level-0 kernel's latency each, five times over); the MFMA tiling orders of all
    // 13 CSRs are computed in one batch at the end.
    ASR_TRY(build_coarse_grids(ctx));
    if (st) ASR_TRY(asr_shard_ownership_coarser(ctx, st));
    {
        ctx->scratch.reset();
        asr_nb_job nb[ASR_NUM_GRIDS];
        for (int i = 0; i < ASR_NUM_GRIDS; ++i) {
            nb[i] = asr_nb_job{ctx->grids[i].keys, ctx->grids[i].v, nullptr, nullptr, nullptr, 0};
            if (st && asr_shard_world(st) > 1) {  // this rank's rows only
                nb[i].owner = asr_shard_owner(st, i);
                nb[i].me = asr_shard_rank(st);
            }
        }
        ASR_TRY(asr_geom_neighbors_build_batch(ctx, ctx->persist, nb, ASR_NUM_GRIDS));
        for (int i = 0; i < ASR_NUM_GRIDS; ++i) {
            GridDev& g = ctx->grids[i];
            g.nrs = nb[i].rs;
            g.nidx = nb[i].idx;
            g.nkidx = nb[i].kidx;
            g.p = nb[i].p;
        }
    }
    ASR_TRY(build_grid_names(ctx));
    // MFMA tiling orders + row-group plans.  (Deferring them to the auxiliary stream beside the continuous conv of the network
    // half -- only the U-Net needs them -- was measured again in round 4: geometry wall 9.5 -> 8.5-8.9 ms, network wall + 1.2 ms.)
    if (st) {  // tiling orders of all rows; the plans are this rank's (asr_shard_lists)
        ASR_TRY(build_tilings_and_plans(ctx, ctx, 0));
        ASR_TRY(asr_shard_lists(ctx, st, prm->precision != 0 && ctx->opt.sconv_plan));
    } else {
        ASR_TRY(build_tilings_and_plans(ctx, ctx, prm->precision));
    }
    ASR_HIP_CHECK(ctx, hipEventRecord(ctx->ev[2], ctx->stream));

    if (overlap)
        worker.join();
    else if (want_search)
        search_rc = search();
    if (search_rc != ASR_HIP_OK) {
        if (sc != ctx) ctx->err = sc->err;
        return search_rc;
    }
    if (want_search) {
        GridDev& g0 = ctx->grids[0];
        ctx->has_search = true;
        ctx->sizes.num_agg_pairs = agg_pairs;
        name_it(ctx, "aggregation_neighbors_index", ctx->agg_idx, 4 * agg_pairs);
        name_it(ctx, "aggregation_neighbors_dist", ctx->agg_dist, 4 * agg_pairs);
        name_it(ctx, "aggregation_scale_compat", ctx->agg_compat, 4 * agg_pairs);
        name_it(ctx, "aggregation_row_splits", ctx->agg_rs, 8 * ((ctx->agg_nq > 0 ? ctx->agg_nq : g0.v) + 1));
        if (ctx->agg_nq > 0) name_it(ctx, "aggregation_rows", const_cast<int32_t*>(ctx->agg_rows), 4 * ctx->agg_nq);
    }
    ASR_HIP_CHECK(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
    // everything implicit_network allocates from the persist arena goes above this mark, so that
    // repeated network() calls on one build do not grow the arena
    arena_mark(ctx->persist, ctx->build_mark);
    ctx->build_mark_ok = true;
    if (st_out) {
        *st_out = st;
        shard_guard.keep = true;
    }
    ctx->build_sharded = st && asr_shard_world(st) > 1;
    return ASR_HIP_OK;
}

// aggregate (net_definitions_torch.py:640-653, 72-120): feats1 [V0, C0] and the per-pair importance of the last build
// feats_amax (f16x2 network): zeroed device scalar that receives the f32 bits of the largest |feats1| (kept by the continuous
// conv's kernels instead of a pass over its 330 MB output)
int implicit_aggregate(asr_hip_context* ctx, const float* points, const float* normals, i64 n, Net& net,
                       unsigned* feats_amax = nullptr) {
    (void)points;
    if (n != ctx->sizes.num_points || ctx->sizes.num_voxels[0] == 0)
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "implicit_network: no matching implicit_build");
    if (!ctx->has_search || !ctx->agg_rs || !ctx->agg_sorted)  // what the LAST BUILD did, not the option's value now
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "implicit_network: the build skipped the aggregation search (option build_search)");
    ctx->scratch.reset();
    if (ctx->build_mark_ok) arena_rewind(ctx->persist, ctx->build_mark);
    if (net.precision != 0 && net.precision != ASR_CONV16_F16 && net.precision != ASR_CONV16_BF16X3 &&
        net.precision != ASR_CONV16_F16X2)
        ASR_FAIL(ctx, ASR_HIP_EINVAL,
                 "implicit_network: precision must be 0, ASR_CONV16_F16, ASR_CONV16_BF16X3 or ASR_CONV16_F16X2");
    GridDev* g = ctx->grids;
    const i64 V0 = g[0].v;
    const i64 P = ctx->sizes.num_agg_pairs;
    ASR_HIP_CHECK(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
    asr_hip_print("network aggregate\n", ASR_HIP_INFO);  // cpp/lib/asr.cpp:314
    const asr_weight *ck, *cb;
    ASR_TRY(net.get("cconv_block_in.conv1.kernel", 5, &ck));
    ASR_TRY(net.get("cconv_block_in.conv1.bias", 1, &cb));
    if (ck->shape[0] != 4 || ck->shape[1] != 4 || ck->shape[2] != 4 || ck->shape[3] != 4)
        ASR_FAIL(ctx, ASR_HIP_EWEIGHT, "cconv_block_in.conv1.kernel must be [4,4,4,4,C]");
    const int C0 = (int)ck->shape[4];
    float* feats = arena_alloc<float>(ctx->scratch, 8 * (size_t)n);
    float* imp_pairs = arena_alloc<float>(ctx->persist, P > V0 ? P : V0);
    float* feats1 = arena_alloc<float>(ctx->persist, (size_t)V0 * C0);
    if (!feats || !imp_pairs || !feats1) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    k_make_feats<<<grid_for(n, 256), 256, 0, ctx->stream>>>(normals, ctx->agg_sorted, n, feats);
    ASR_CHECK_LAUNCH(ctx);
    ASR_TRY(asr_conv_agg_importance(ctx, ctx->agg_compat, ctx->agg_dist, P, imp_pairs));
    // positions, features and pair indices all in Morton order (the search's own point order)
    if (ctx->agg_nq > 0) {  // sharded geometry: the listed rows only; the others stay zero (f16x2: neutral for the maximum)
        float* part = arena_alloc<float>(ctx->scratch, (size_t)ctx->agg_nq * C0);
        if (!part) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
        ASR_TRY(asr_conv_cconv(ctx, ck->data, ctx->agg_qcenters, ctx->agg_qsizes, feats, feats, ctx->agg_spos, imp_pairs,
                               ctx->agg_rs, ctx->agg_nq, 4, C0, 1, cb->data, 1, part, 1, feats_amax));
        ASR_HIP_CHECK(ctx, hipMemsetAsync(feats1, 0, sizeof(float) * (size_t)V0 * C0, ctx->stream));
        k_scatter_rows<<<grid_for(ctx->agg_nq * C0, 256), 256, 0, ctx->stream>>>(part, ctx->agg_rows, ctx->agg_nq, C0, feats1);
        ASR_CHECK_LAUNCH(ctx);
    } else {
        ASR_TRY(asr_conv_cconv(ctx, ck->data, g[0].centers, g[0].sizes, feats, feats,
                               ctx->agg_spos, imp_pairs, ctx->agg_rs, V0, 4, C0, 1, cb->data, 1, feats1, 1, feats_amax));
    }
    ctx->feats1 = feats1;
    ctx->importance = imp_pairs;
    ctx->feats1_width = C0;
    name_it(ctx, "feats1", feats1, 4 * (size_t)V0 * C0);
    name_it(ctx, "importance", imp_pairs, 4 * (size_t)P);
    ASR_HIP_CHECK(ctx, hipEventRecord(ctx->ev[5], ctx->stream));
    return ASR_HIP_OK;
}

// U-Net + decoder over the feats1 / importance of implicit_aggregate (the part of the network that exchanges halos in a
// sharded forward).  With ctx->dry_launch set this is the PREPARATION pass: every weight look-up and shape check, every
// allocation (activations, packed weights, partial-sum and staging buffers) happens, no kernel is launched and nothing travels.
int network_unet_decode(asr_hip_context* ctx, Net& net, const asr_implicit_params* prm, float* values_out, unsigned* feats_amax) {
    GridDev* g = ctx->grids;
    const i64 V0 = g[0].v;
    const i64 P = ctx->sizes.num_agg_pairs;
    float* feats1 = ctx->feats1;
    float* imp_pairs = ctx->importance;
    const int C0 = ctx->feats1_width;
    net.replicated_imp = imp_pairs;

    // the per-PAIR importance array is indexed with grid-0 VOXEL indices
    // (net_definitions_torch.py:572-578 -> common_torch.py:125).  torch would raise on P < V0.
    if (P < V0)
        ASR_FAIL(ctx, ASR_HIP_EINVAL,
                 "aggregation pairs (%lld) < voxels (%lld): reference indexing is out of range",
                 (long long)P, (long long)V0);

    // ---- unet (net_definitions_torch.py:535-638) ----
    asr_hip_print("network unet\n", ASR_HIP_INFO);  // cpp/lib/asr.cpp:319
    int c_enc[5], c_down[5], c_up[4], c_dec[4];
    {
        int a, b;
        for (int i = 0; i < 5; ++i) {
            std::string nm = "sparseconv_encblock" + std::to_string(i);
            ASR_TRY(net.cout_of(nm + ".conv1a", &a));
            ASR_TRY(net.cout_of(nm + ".conv1b", &b));
            c_enc[i] = a + b;
        }
        for (int i = 1; i <= 3; ++i) {
            std::string nm = "sparseconv_down" + std::to_string(i);
            ASR_TRY(net.cout_of(nm + ".conv1a", &a));
            ASR_TRY(net.cout_of(nm + ".conv1b", &b));
            c_down[i] = a + b;
        }
        c_down[4] = c_down[3];  // sparseconv_down3 is reused for 3->4 (B.3)
        for (int i = 0; i < 4; ++i) {
            ASR_TRY(net.cout_of("sparseconv_up" + std::to_string(i) + ".conv1", &c_up[i]));
            ASR_TRY(net.cout_of("sparseconv_decblock" + std::to_string(i) + ".conv4", &c_dec[i]));
        }
    }
    auto buf = [&](i64 rows, int c) { return net.act(rows, c); };  // element type per precision
    // concat buffers [up_i | enc_i] for levels 1..3; level 0 uses a residual add
    float* cat[4] = {nullptr, nullptr, nullptr, nullptr};
    int cat_ld[4] = {0, 0, 0, 0};
    for (int i = 1; i <= 3; ++i) {
        cat_ld[i] = c_up[i] + c_enc[i];
        cat[i] = buf(g[i].v, cat_ld[i]);
        if (!cat[i]) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    }
    float* f2 = buf(V0, c_enc[0]);
    float* f10 = buf(g[4].v, c_enc[4]);
    if (!f2 || !f10) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");

    unsigned* cat_amax[4] = {nullptr, net.new_amax(), net.new_amax(), net.new_amax()};
    Feat enc_out[5];
    enc_out[0] = Feat{f2, c_enc[0], c_enc[0], net.new_amax()};
    for (int i = 1; i <= 3; ++i) enc_out[i] = Feat{net.col(cat[i], c_up[i]), cat_ld[i], c_enc[i], cat_amax[i]};
    enc_out[4] = Feat{f10, c_enc[4], c_enc[4], net.new_amax()};

    float* imp = nullptr;
    float* feats1_in = feats1;
    if (net.precision == ASR_CONV16_F16) {  // f16 activations: the continuous conv's f32 output is converted once
        feats1_in = buf(V0, C0);
        if (!feats1_in) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
        ASR_TRY(asr_conv16_convert(ctx, feats1, V0 * (i64)C0, feats1_in, 1));
    }
    Feat f_in{feats1_in, C0, C0, feats_amax};
    ASR_TRY(net.block("sparseconv_encblock0", f_in, g[0], imp_pairs, true, enc_out[0], &imp));
    for (int i = 1; i <= 4; ++i) {
        std::string dn = "sparseconv_down" + std::to_string(i < 4 ? i : 3);
        float* t = buf(g[i].v, c_down[i]);
        if (!t) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
        float* imp_d = nullptr;
        const Feat ft{t, c_down[i], c_down[i], net.new_amax()};
        ASR_TRY(net.down(dn, enc_out[i - 1], g[i - 1], g[i], imp, ft, &imp_d));
        float* imp_e = nullptr;
        ASR_TRY(net.block("sparseconv_encblock" + std::to_string(i), ft, g[i], imp_d, true, enc_out[i], &imp_e));
        imp = imp_e;
    }
    // decoder
    Feat cur = enc_out[4];
    for (int i = 3; i >= 1; --i) {
        ASR_TRY(net.up("sparseconv_up" + std::to_string(i), cur, g[i], g[i + 1],
                       Feat{cat[i], cat_ld[i], c_up[i], cat_amax[i]}, nullptr, 0));
        float* d = buf(g[i].v, c_dec[i]);
        if (!d) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
        float* dummy = nullptr;
        const Feat fd{d, c_dec[i], c_dec[i], net.new_amax()};
        ASR_TRY(net.block("sparseconv_decblock" + std::to_string(i), Feat{cat[i], cat_ld[i], cat_ld[i], cat_amax[i]},
                          g[i], nullptr, false, fd, &dummy));
        cur = fd;
    }
    if (c_up[0] != c_enc[0])
        ASR_FAIL(ctx, ASR_HIP_EWEIGHT, "residual skip needs up0 width == encblock0 width");
    float* f21 = buf(V0, c_up[0]);
    float* code = arena_alloc<float>(ctx->persist, (size_t)V0 * c_dec[0]);
    if (!f21 || !code) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    // feats21 = relu(up0(feats19)) + feats2   (net_definitions_torch.py:631-633)
    const Feat ff21{f21, c_up[0], c_up[0], net.new_amax()};
    ASR_TRY(net.up("sparseconv_up0", cur, g[0], g[1], ff21, f2, c_enc[0]));
    {
        float* dummy = nullptr;
        ASR_TRY(net.block("sparseconv_decblock0", ff21, g[0], nullptr, false,
                          Feat{code, c_dec[0], c_dec[0]}, &dummy, true));  // `code` is f32 in every mode
    }
    ctx->code = code;
    name_it(ctx, "code", code, 4 * (size_t)V0 * c_dec[0]);
    ASR_HIP_CHECK(ctx, hipEventRecord(ctx->ev[6], ctx->stream));

    // ---- decode + sdf scale (net_definitions_torch.py:655-666, asr.cpp:324-336) ----
    asr_hip_print("network decode\n", ASR_HIP_INFO);  // cpp/lib/asr.cpp:323
    const asr_weight *w1, *b1, *w2, *b2, *w3;
    ASR_TRY(net.get("dense_decoder1.weight", 2, &w1));
    ASR_TRY(net.get("dense_decoder1.bias", 1, &b1));
    ASR_TRY(net.get("dense_decoder2.weight", 2, &w2));
    ASR_TRY(net.get("dense_decoder2.bias", 1, &b2));
    ASR_TRY(net.get("dense_decoder3.weight", 2, &w3));
    if (w1->shape[1] != 3 + c_dec[0] || w2->shape[1] != w1->shape[0] || w3->shape[1] != w2->shape[0] ||
        w3->shape[0] != 2)
        ASR_FAIL(ctx, ASR_HIP_EWEIGHT, "dense_decoder shapes do not chain");
    float* values = values_out;
    if (!values) {
        values = arena_alloc<float>(ctx->persist, 2 * (size_t)V0);
        if (!values) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    }
    // sharded forward: the decoder runs over the rows this rank owns; the owners' rows are gathered
    const int32_t* drows = nullptr;
    i64 dn = V0;
    if (ctx->shard && asr_shard_world(ctx->shard) > 1) {
        drows = asr_shard_owned_rows0(ctx->shard, &dn);
        if (!drows) dn = V0;
    }
    ASR_TRY(asr_conv_decode(ctx, code, dn, c_dec[0], w1->data, b1->data, (int)w1->shape[0], w2->data,
                            b2->data, (int)w2->shape[0], w3->data,
                            prm->scale_sdf ? g[0].sizes : nullptr, values, drows));
    if (ctx->shard) ASR_TRY(asr_shard_stitch(ctx, ctx->shard, values));
    ctx->values = values;
    name_it(ctx, "values", values, 8 * (size_t)V0);
    ASR_HIP_CHECK(ctx, hipEventRecord(ctx->ev[7], ctx->stream));
    return ASR_HIP_OK;
}

// comm != null (asr_hip_implicit_forward_sharded, ctx->shard set): the aggregation runs (it has no collective), then the
// U-Net and the decoder go through their preparation pass, the ranks agree that nobody failed, and only then the pass
// that launches and exchanges runs -- from the same arena marks, so that it repeats the allocations it has already made.
int implicit_network(asr_hip_context* ctx, const float* points, const float* normals, i64 n,
                     const asr_weight* weights, int num_weights, const asr_implicit_params* prm,
                     float* values_out, const asr_shard_comm* comm = nullptr) {
    ASR_TRY(ensure_events(ctx));
    Net net{ctx, {weights, num_weights}};
    net.precision = prm->precision;
    // f16x2: every activation buffer has a running maximum, kept by the kernels that write it (a concat buffer by both of its
    // producers) and read by the ones that consume it -- feats1's by the continuous conv
    unsigned* feats_amax = nullptr;
    int rc = net.begin_amax();
    if (rc == ASR_HIP_OK) {
        feats_amax = net.new_amax();
        rc = implicit_aggregate(ctx, points, normals, n, net, feats_amax);
    }
    if (!comm || comm->world <= 1) {
        ASR_TRY(rc);
        return network_unet_decode(ctx, net, prm, values_out, feats_amax);
    }
    ArenaMark m_scratch, m_persist;
    arena_mark(ctx->scratch, m_scratch);
    arena_mark(ctx->persist, m_persist);
    const int amax_mark = net.num_amax;
    if (rc == ASR_HIP_OK) {
        ctx->dry_launch = true;
        rc = network_unet_decode(ctx, net, prm, values_out, feats_amax);
        ctx->dry_launch = false;
    }
    if (rc == ASR_HIP_OK && ctx->opt.inject_failure == 2) {
        ctx->err = "injected failure (option inject_failure = 2)";
        rc = ASR_HIP_ELOGIC;
    }
    ASR_TRY(asr_shard_agree(ctx, comm, rc, "network preparation"));
    arena_rewind(ctx->scratch, m_scratch);
    arena_rewind(ctx->persist, m_persist);
    net.num_amax = amax_mark;
    return network_unet_decode(ctx, net, prm, values_out, feats_amax);
}

}  // namespace

extern "C" {

int asr_hip_implicit_build(asr_hip_context* ctx, const float* points, const float* radii, int64_t n,
                           const asr_implicit_params* prm, asr_implicit_sizes* sizes) {
    CTX_GUARD(ctx);
    if (!points || !radii || n <= 0 || !prm)
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "implicit_build: points is null!");  // cpp/lib/asr.cpp:101-103
    ASR_TRY(implicit_build(ctx, points, radii, n, prm));
    if (sizes) *sizes = ctx->sizes;
    return ASR_HIP_OK;
}
int asr_hip_implicit_network(asr_hip_context* ctx, const float* points, const float* normals,
                             int64_t n, const asr_weight* weights, int num_weights,
                             const asr_implicit_params* prm, float* values_out) {
    CTX_GUARD(ctx);
    if (!points || !normals || n <= 0 || !weights || !prm)
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "implicit_network: null argument");
    if (ctx->build_sharded)  // (the sharded forward calls the internal function with ctx->shard set)
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "implicit_network: the last build was one rank's build of a sharded cloud (partial neighbour "
                                      "lists, plans and aggregation rows): run asr_hip_implicit_build again");
    return implicit_network(ctx, points, normals, n, weights, num_weights, prm, values_out);
}
int asr_hip_implicit_aggregate(asr_hip_context* ctx, const float* points, const float* normals, int64_t n,
                               const asr_weight* weights, int num_weights, const asr_implicit_params* prm) {
    CTX_GUARD(ctx);
    if (!points || !normals || n <= 0 || !weights || !prm)
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "implicit_aggregate: null argument");
    if (ctx->build_sharded)
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "implicit_aggregate: the last build was one rank's build of a sharded cloud (the "
                                      "aggregation rows of the owned voxels only): run asr_hip_implicit_build again");
    ASR_TRY(ensure_events(ctx));
    Net net{ctx, {weights, num_weights}};
    net.precision = prm->precision;
    return implicit_aggregate(ctx, points, normals, n, net);
}
int asr_hip_implicit_forward(asr_hip_context* ctx, const float* points, const float* normals,
                             const float* radii, int64_t n, const asr_weight* weights,
                             int num_weights, const asr_implicit_params* prm,
                             asr_implicit_sizes* sizes) {
    CTX_GUARD(ctx);
    if (!points || !normals || !radii || n <= 0 || !weights || !prm)
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "implicit_forward: points is null!");
    ASR_TRY(implicit_build(ctx, points, radii, n, prm));
    ASR_TRY(implicit_network(ctx, points, normals, n, weights, num_weights, prm, nullptr));
    if (sizes) *sizes = ctx->sizes;
    return ASR_HIP_OK;
}
int asr_hip_implicit_forward_sharded(asr_hip_context* ctx, const asr_shard_comm* comm, const float* points,
                                     const float* normals, const float* radii, int64_t n, const asr_weight* weights,
                                     int num_weights, const asr_implicit_params* prm, asr_implicit_sizes* sizes,
                                     float* values_out, asr_shard_stats* stats) {
    CTX_GUARD(ctx);
    if (!comm || !prm || (num_weights > 0 && !weights)) ASR_FAIL(ctx, ASR_HIP_EINVAL, "implicit_forward_sharded: null argument");
    if (n > 0 && (!points || !normals || !radii))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "implicit_forward_sharded: points is null!");  // cpp/lib/asr.cpp:101-103
    if (prm->precision == ASR_CONV16_F16)
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "implicit_forward_sharded: the f16-activation network is not sharded (precision 0, "
                                      "ASR_CONV16_BF16X3 or ASR_CONV16_F16X2)");
    if (!ctx->opt.build_search) ASR_FAIL(ctx, ASR_HIP_EINVAL, "implicit_forward_sharded needs option build_search");
    // option shard_geometry: 0 = the whole cloud's geometry on every rank (overlapped search, as on one GPU), 1 = lists,
    // plans and search for the owned voxels only, -1 (default) = 1 whenever there is more than one rank
    const bool own_geometry = comm->world > 1 && (ctx->opt.shard_geometry > 0 || (ctx->opt.shard_geometry < 0 && comm->world >= 2));
    // No rank may fail alone between two collectives (its peers would wait in ncclSend / ncclRecv for ever): the build has no
    // collective, the ranks agree on its outcome, and the network prepares everything that can fail before its first
    // exchange and agrees again (implicit_network, asr_shard_agree).
    asr_shard_state* st = nullptr;
    int rc;
    if (own_geometry) {
        rc = implicit_build(ctx, points, radii, n, prm, comm, &st);
    } else {
        rc = implicit_build(ctx, points, radii, n, prm);
        if (rc == ASR_HIP_OK) rc = asr_shard_build(ctx, comm, prm->precision != 0 && ctx->opt.sconv_plan, &st);
    }
    if (rc == ASR_HIP_OK && ctx->opt.inject_failure == 1) {
        ctx->err = "injected failure (option inject_failure = 1)";
        rc = ASR_HIP_ELOGIC;
    }
    rc = asr_shard_agree(ctx, comm, rc, "geometry build");
    if (rc != ASR_HIP_OK) {
        if (st) asr_shard_free(st);
        return rc;
    }
    if (sizes) *sizes = ctx->sizes;
    ctx->shard = st;
    rc = implicit_network(ctx, points, normals, n, weights, num_weights, prm, values_out, comm);
    ctx->shard = nullptr;
    ctx->dry_launch = false;
    if (stats) *stats = *asr_shard_get_stats(st);
    asr_shard_free(st);
    return rc;
}
int asr_hip_implicit_get(asr_hip_context* ctx, const char* name, void* dst, size_t* nbytes) {
    CTX_GUARD(ctx);
    if (!name) ASR_FAIL(ctx, ASR_HIP_EINVAL, "implicit_get: null name");
    auto it = ctx->named.find(name);
    if (it == ctx->named.end()) ASR_FAIL(ctx, ASR_HIP_EINVAL, "implicit_get: unknown array '%s'", name);
    if (nbytes) *nbytes = it->second.second;
    if (dst && it->second.second)
        ASR_HIP_CHECK(ctx, hipMemcpyAsync(dst, it->second.first, it->second.second,
                                          hipMemcpyDeviceToDevice, ctx->stream));
    return ASR_HIP_OK;
}
int asr_hip_implicit_stage_ms(asr_hip_context* ctx, float out_ms[8]) {
    CTX_GUARD(ctx);
    if (!ctx->ev_ok || !out_ms) ASR_FAIL(ctx, ASR_HIP_EINVAL, "stage_ms: nothing recorded");
    ASR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    // [0] octree [1] grids [2] aggregation search [3] continuous conv [4] unet [5] decode
    // [6] geometry wall (octree start .. search joined) [7] network wall.  With the search on the
    // auxiliary stream (option "overlap", default) [1] and [2] run concurrently: [2] is then measured
    // on the auxiliary stream and [0] + [1] + [2] > [6].
    const int pairs[8][2] = {{0, 1}, {1, 2}, {2, 3}, {4, 5}, {5, 6}, {6, 7}, {0, 3}, {4, 7}};
    for (int i = 0; i < 8; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->ev[pairs[i][0]], ctx->ev[pairs[i][1]]) != hipSuccess) ms = -1.f;
        out_ms[i] = ms;
    }
    if (ctx->search_overlapped && ctx->aux_t0 && ctx->aux_t1) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->aux_t0, ctx->aux_t1) != hipSuccess) ms = -1.f;
        out_ms[2] = ms;
    }
    return ASR_HIP_OK;
}

}  // extern "C"

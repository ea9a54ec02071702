/*
 * asr_hip.h -- C ABI of the MI355X (gfx950) implementation of the adaptive-surface-
 * reconstruction hot path: oriented points -> signed/unsigned implicit values on the
 * adaptive octree grid.
 *
 * Conventions
 *   - every pointer documented as "dev" is a device (HBM) pointer, everything else is host
 *   - all kernels are enqueued on the context's hipStream_t; functions that return a
 *     data-dependent size synchronise that stream before returning
 *   - int return value: 0 = ok, otherwise an ASR_HIP_E* code; asr_hip_last_error() gives text
 *   - no ownership transfer: the caller allocates every output after the *_count call
 *   - dtypes follow the reference tensors (cpp/lib/asr.cpp:179-312): indices int32, kernel
 *     indices uint8, row splits int64, keys uint64, features float32
 *
 * The reference has no C ABI (the ASR_API_EXTERN_C macros of cpp/lib/asr_config.h:18-28 are
 * unused); each entry point cites the reference interface it stands in for.
 */
#ifndef ASR_HIP_H
#define ASR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASR_HIP_OK 0
#define ASR_HIP_EINVAL 1   /* bad argument (shape / null / unsupported size)            */
#define ASR_HIP_EHIP 2     /* a HIP runtime call failed                                 */
#define ASR_HIP_ENODEV 3   /* no gfx950 device / extension not usable                   */
#define ASR_HIP_ELOGIC 4   /* internal invariant violated (e.g. hash table overflow)    */
#define ASR_HIP_EWEIGHT 5  /* weight table incomplete / wrong shape                     */
#define ASR_HIP_EPEER 6    /* sharded forward: ANOTHER rank failed; every rank returns  */

#define ASR_MAX_LEVEL 21      /* cpp/lib/octreebase.h:42                                  */
#define ASR_NUM_GRIDS 5       /* cpp/lib/asr.cpp:156, models/v0/net_definitions_torch.py:403 */

typedef struct asr_hip_context asr_hip_context;

/* Octree frame: per-level voxel sizes and the integer offset of the root cube.
 * Replaces asr::Octree::Octree (cpp/lib/octree.cpp:20-42); plain host POD, passed by value
 * into kernels. */
typedef struct asr_octree_frame {
    float voxel_size[ASR_MAX_LEVEL + 1];
    float inv_voxel_size[ASR_MAX_LEVEL + 1];
    int32_t offset[3];
    float bb_min[3];
    float bb_max[3];
} asr_octree_frame;

/* ---- context ---------------------------------------------------------------------- */
/* stream: a hipStream_t (NULL = the null stream). The context owns a device arena that is
 * reused across calls. */
int asr_hip_context_create(asr_hip_context** ctx, void* stream);
void asr_hip_context_destroy(asr_hip_context* ctx);
void asr_hip_context_set_stream(asr_hip_context* ctx, void* stream);
const char* asr_hip_last_error(const asr_hip_context* ctx);
const char* asr_hip_version(void); /* asr::GetVersionStr, cpp/lib/asr.hpp:29 */
/* sizeof() of an ABI struct by name ("asr_octree_frame", "asr_sparse_conv_args", "asr_weight",
 * "asr_implicit_params", "asr_implicit_sizes"); 0 for unknown names. Lets FFI bindings verify
 * their struct layouts. */
size_t asr_hip_struct_size(const char* name);
/* bytes currently reserved by the arena */
size_t asr_hip_context_reserved_bytes(const asr_hip_context* ctx);
/* HIP device the context is bound to (the device current when it was created); every entry point
 * selects it on the calling thread.  -1 for a null context. */
int asr_hip_context_device(const asr_hip_context* ctx);
/* The whole-path driver keeps packed 16-bit copies of the weight tensors it was given (keyed by the tensors'
 * device pointers and shapes).  Call this after the contents of a weight tensor changed in place -- or when a new
 * table may reuse the addresses of an old one -- so that the copies are made again by the next forward. */
int asr_hip_context_weights_changed(asr_hip_context* ctx);
/* Per-context tunables (no process-wide state).  Names: "sconv_min_blocks" (2816) and
 * "sconv_wide_min" (2048): launch-size thresholds that pick the sparse-conv tile shape;
 * "row_segment" (524288), "row_lpt" (1): MFMA row regrouping; "overlap" (1): aggregation search on a
 * second stream; "build_search" (1): 0 makes asr_hip_implicit_build stop
 * after the grid hierarchy (a rank of a sharded run searches only the rows it owns).  Results never depend
 * on the tuning options: none of them skips work.
 * "search_half" (1): the aggregation search of asr_hip_implicit_build covers a voxel's ball with 4^3 half-size
 * cells for the rows with many candidates (2: for every voxel, 0: 3^3 full-size cells throughout).  get_option also answers the read-only "last_search_margin_pairs": pairs the last
 * such search found in the rounding margin of the half-size cells. */
int asr_hip_context_set_option(asr_hip_context* ctx, const char* name, int64_t value);
int asr_hip_context_get_option(asr_hip_context* ctx, const char* name, int64_t* value);
/* The table of tunables: name and built-in default of option `index` (0, 1, ... until ASR_HIP_EINVAL).  A context's defaults can
 * be seeded from the environment (ASR_<NAME>) when it is created; bench.py prints every option that differs from its built-in
 * default into the line's `config` (the reference has no tunables: cpp/lib/asr.hpp:27-101). */
int asr_hip_option_info(int index, const char** name, int64_t* default_value);

/* ---- print callbacks: asr::SetPrintCallbackFunction (cpp/lib/asr.hpp:25-34, asr.cpp:34-47) ------------- */
/* Process-wide like the reference's (a static table of four callbacks).  Levels: cpp/lib/asr.hpp:27.  The whole-path
 * entry points announce their stages at ASR_HIP_INFO with the reference's texts (cpp/lib/asr.cpp:117,144,264,
 * 314-323): "grid building\n", "aggregate\n", "network aggregate\n", "network unet\n", "network decode\n"
 * ("preprocessing\n" comes from the host side that runs the pre-filter).  Without a callback nothing is printed. */
#define ASR_HIP_DEBUG 0
#define ASR_HIP_INFO 1
#define ASR_HIP_WARN 2
#define ASR_HIP_ERROR 3
typedef void (*asr_hip_print_callback)(const char* msg, void* user);
/* installs `callback` (NULL: removes) for each of levels[0..num_levels); ASR_HIP_EINVAL for a level outside
 * DEBUG..ERROR ("invalid verbosity level", asr.cpp:42-44) -- nothing is changed in that case */
int asr_hip_set_print_callback(asr_hip_print_callback callback, void* user, const int* levels, int num_levels);
/* asr::Print (cpp/lib/utils.h:26): hands msg to the callback of `level`, if one is installed */
void asr_hip_print(const char* msg, int level);

/* ---- a3: octree frame (cpp/lib/octree.cpp:20-42), host only --------------------------- */
int asr_octree_frame_init(asr_octree_frame* frame, const float bb_min[3], const float bb_max[3]);

/* ---- a1/a2/a3: per-point location codes (cpp/lib/octree.h:42-68, octreebase.h:59-65) --- */
/* keys_out[i] = key of point i at the level chosen from radius_scale*radii[i] (clamped to
 * max_depth); 0 for points outside [bb_min,bb_max] or with an invalid coordinate. */
int asr_hip_point_keys(asr_hip_context* ctx, const asr_octree_frame* frame,
                       const float* points_dev, const float* radii_dev, int64_t n,
                       float radius_scale, int max_depth, uint64_t* keys_out_dev);

/* ---- a4: octree construction (asr::CreateOctreeFromPoints, cpp/lib/octree.cpp:230-280;
 *      pybind create_octree, cpp/pybind/module.cpp:144-161) ----------------------------- */
/* Builds the balanced node set inside the context and reports its sizes (grow_steps = 0, the value of the
 * reconstruction path, cpp/lib/asr.cpp:151-153). */
int asr_hip_octree_build(asr_hip_context* ctx, const asr_octree_frame* frame,
                         const float* points_dev, const float* radii_dev, int64_t n,
                         float radius_scale, int max_depth, int64_t* num_nodes,
                         int64_t* num_leaves);
/* the same with Octree::Grow (cpp/lib/octree.cpp:44-108) applied `grow_steps` times to the point keys before the
 * closure, as create_octree(..., grow_steps, ...) of the pybind module does (cpp/pybind/module.cpp:144-161).  The
 * candidate test of an iteration (:87) is evaluated against the key set at the start of the iteration; the reference
 * walks its hash map sequentially, which only matters for grow_steps >= 2 on trees that mix levels. */
int asr_hip_octree_build_grow(asr_hip_context* ctx, const asr_octree_frame* frame,
                              const float* points_dev, const float* radii_dev, int64_t n,
                              float radius_scale, int grow_steps, int max_depth, int64_t* num_nodes,
                              int64_t* num_leaves);
/* An octree in parts, for builds that are spread over several processes: the node set is the closure (all siblings,
 * all ancestors; cpp/lib/octree.cpp:110-150) of the keys of `points` AND of `extra_keys` (node keys of other builds),
 * then 2:1 balanced (:152-206) unless balance == 0.  Closure commutes with union, so every rank can close the keys of its
 * share of the points (balance = 0), the ranks exchange their node lists, and each rank closes and balances the union
 * (n = 0, extra_keys = all lists): the tree of the whole cloud, bit for bit. */
int asr_hip_octree_build_parts(asr_hip_context* ctx, const asr_octree_frame* frame, const float* points_dev,
                               const float* radii_dev, int64_t n, float radius_scale, int max_depth,
                               const uint64_t* extra_keys_dev, int64_t num_extra, int balance,
                               int64_t* num_nodes, int64_t* num_leaves);
/* copies the sorted node keys / sorted leaf keys (tree.leaves) of the last build */
int asr_hip_octree_get(asr_hip_context* ctx, uint64_t* nodes_out_dev, uint64_t* leaves_out_dev);

/* ---- dual cells ("next" row D.1): asr::CreateDualVertexIndices (cpp/lib/grid.cpp:316-459) ------ */
/* For the octree of the last asr_hip_octree_build / asr_hip_implicit_build of this context.
 * count returns the number of dual cells D; fill writes [D,8] leaf indices (leaf order x corner
 * order), the `dual_vertex_indices` consumed by CreateTriangleMesh (cpp/lib/asr.cpp:154,340). */
int asr_hip_dual_cells_count(asr_hip_context* ctx, int64_t* num_cells);
/* the same for ANY octree given by its sorted node / leaf keys (asr_hip_octree_get): the pybind Octree handle
 * (cpp/pybind/module.cpp:230-235) stays usable however many trees were built since.  The arrays must stay valid
 * until the fill call. */
int asr_hip_dual_cells_count_for(asr_hip_context* ctx, const uint64_t* nodes_dev, int64_t num_nodes,
                                 const uint64_t* leaves_dev, int64_t num_leaves, int64_t* num_cells);
int asr_hip_dual_cells_fill(asr_hip_context* ctx, int64_t* dual_vertex_indices_out_dev);

/* ---- dual contouring ("next" row D.2): asr::CreateTriangleMesh (cpp/lib/contouring.cpp:29-460) ---- */
/* values [V,2] (signed, unsigned), dual_vertex_indices [D,8] int64, node_positions [V,3]; all
 * device pointers that must stay valid until the fill call.  count returns the mesh sizes, fill
 * writes vertices f32[M,3] (one per active dual cell in dual order, then the fan centres in
 * emission order) and triangles i32[T,3] in the reference's serial emission order, including the
 * corner order it derives from libstdc++'s unordered_set iteration (see csrc/asr_uset.h). */
int asr_hip_contour_count(asr_hip_context* ctx, const float* values_dev, int64_t num_values,
                          const int64_t* dual_vertex_indices_dev, int64_t num_duals,
                          const float* node_positions_dev, float value_threshold,
                          int64_t* num_vertices, int64_t* num_triangles);
int asr_hip_contour_fill(asr_hip_context* ctx, float* vertices_out_dev, int32_t* triangles_out_dev);

/* ---- component filter ("next" row D.3): asr::RemoveConnectedComponents
 * (cpp/lib/postprocess.cpp:141-176).  Keeps the keep_n largest components (size in vertices,
 * ties -> the component found later first, like std::greater on (size, label)) that have at least
 * min_size vertices; compacts vertices and re-indexes triangles, order preserving. */
int asr_hip_components_count(asr_hip_context* ctx, const float* vertices_dev, int64_t num_vertices,
                             const int32_t* triangles_dev, int64_t num_triangles, int64_t keep_n,
                             int64_t min_size, int64_t* num_vertices_out, int64_t* num_triangles_out);
int asr_hip_components_fill(asr_hip_context* ctx, float* vertices_out_dev, int32_t* triangles_out_dev);

/* Host-only: asr::ComputeInlierFromDensity (cpp/lib/preprocess.cpp:41-62) on the neighbour counts of
 * asr_hip_radius_neighbor_count (host array).  Reproduces the reference literally, including that
 * it compares the counts AFTER std::partial_sort has permuted them in place (:53-60), so inlier[i]
 * is not a function of point i's own count (quirk B.11 in DESIGN.md). */
int asr_density_inlier(const int64_t* counts_host, int64_t n, double density_percentile_threshold,
                       uint8_t* inlier_out_host);

/* Host-only helper (no GPU): iteration order of a libstdc++ std::unordered_set<size_t> filled with
 * xs[0..n) in that order, n <= 29 -- the rule the contouring kernel replays (csrc/asr_uset.h). */
int asr_hip_unordered_set_order(const uint32_t* xs, int n, uint32_t* out);

/* ---- a5: CreateLeafNeighborInformation (cpp/lib/grid.cpp:43-175) ---------------------- */
/* keys: sorted unique voxel keys. count: fills row_splits[V+1] and returns the pair count;
 * fill: writes the CSR entries in ascending kernel-slot order. */
int asr_hip_grid_neighbors_count(asr_hip_context* ctx, const uint64_t* keys_dev, int64_t v,
                                 int64_t* row_splits_out_dev, int64_t* num_pairs);
int asr_hip_grid_neighbors_fill(asr_hip_context* ctx, const uint64_t* keys_dev, int64_t v,
                                const int64_t* row_splits_dev, int32_t* index_out_dev,
                                uint8_t* kernel_index_out_dev);

/* The same for a list of rows (ascending voxel indices, int32): row_splits keeps its full length V + 1 with empty
 * rows for the voxels that are not listed, index / kernel_index hold the listed rows' entries only.  A rank of the
 * one-scan sharding builds the lists of the voxels it owns. */
int asr_hip_grid_neighbors_rows_count(asr_hip_context* ctx, const uint64_t* keys_dev, int64_t v,
                                      const int32_t* rows_dev, int64_t num_rows, int64_t* row_splits_out_dev,
                                      int64_t* num_pairs);
int asr_hip_grid_neighbors_rows_fill(asr_hip_context* ctx, const uint64_t* keys_dev, int64_t v,
                                     const int32_t* rows_dev, int64_t num_rows, const int64_t* row_splits_dev,
                                     int32_t* index_out_dev, uint8_t* kernel_index_out_dev);

/* ---- a6: CombineSiblings (cpp/lib/grid.cpp:177-243) ----------------------------------- */
int asr_hip_grid_coarsen_count(asr_hip_context* ctx, const uint64_t* keys_dev, int64_t v,
                               int64_t* v_out);
int asr_hip_grid_coarsen_fill(asr_hip_context* ctx, const uint64_t* keys_dev, int64_t v,
                              uint64_t* out_keys_dev, int64_t v_out, int32_t* up_index_out_dev,
                              uint8_t* up_kernel_index_out_dev, int64_t* up_row_splits_out_dev);

/* ---- a7: InitGridVoxelInfo (cpp/lib/grid.cpp:251-268) --------------------------------- */
int asr_hip_voxel_info(asr_hip_context* ctx, const asr_octree_frame* frame,
                       const uint64_t* keys_dev, int64_t v, float* centers_out_dev,
                       float* sizes_out_dev);

/* ---- a8: ComputeAggregationNeighborsAndScaleCompatibility (cpp/lib/nsearch.cpp:107-162) - */
/* Members of row q: points with ((dx*dx+dy*dy)+dz*dz) < size[q]^2, ordered by (distance,
 * index); dist = squared distance; compat = (min(size, 2 r)/max(size, 2 r))^2. */
int asr_hip_multi_radius_search_count(asr_hip_context* ctx, const asr_octree_frame* frame,
                                      const float* points_dev, int64_t n,
                                      const float* centers_dev, const float* sizes_dev,
                                      int64_t v, int64_t* row_splits_out_dev,
                                      int64_t* num_pairs);
int asr_hip_multi_radius_search_fill(asr_hip_context* ctx, const float* points_dev,
                                     const float* radii_dev, int64_t n,
                                     const float* centers_dev, const float* sizes_dev,
                                     int64_t v, const int64_t* row_splits_dev,
                                     int32_t* index_out_dev, float* dist_out_dev,
                                     float* compat_out_dev);

/* ---- pre-filter ("next" row D.4): asr::KDTree (cpp/lib/nsearch.cpp:23-105) -------------------- */
/* The frame only defines the acceleration grid; any box that contains the points works.
 * radii_out[i] = distance to the k-th nearest neighbour, the point itself included (:30-51).
 * If inlier_out is given, radii_in is required: inlier iff fewer than outlier_threshold of the k
 * nearest neighbours have radius < radius_fraction * radii_in[i] (:54-86). */
int asr_hip_knn_radius(asr_hip_context* ctx, const asr_octree_frame* frame, const float* points_dev,
                       int64_t n, int k, const float* radii_in_dev, float radius_fraction,
                       int outlier_threshold, float* radii_out_dev, uint8_t* inlier_out_dev);
/* counts_out[i] = number of points with squared distance < radii[i]^2 (:88-105) */
int asr_hip_radius_neighbor_count(asr_hip_context* ctx, const asr_octree_frame* frame,
                                  const float* points_dev, const float* radii_dev, int64_t n,
                                  int64_t* counts_out_dev);

/* ---- a10: open3d::continuous_conv as used by CConvAggregationBlock
 *      (models/v0/net_definitions_torch.py:53-70,107-116): kernel 4x4x4, align_corners,
 *      linear, ball_to_cube_radial, per-output extent, per-neighbour importance ------------ */
/* filters [4,4,4,cin,cout]; out = conv (+bias if bias_dev) (relu if relu). */
int asr_hip_continuous_conv_f32(asr_hip_context* ctx, const float* filters_dev,
                                const float* out_positions_dev, const float* extents_dev,
                                const float* inp_positions_dev, const float* inp_features_dev,
                                const int32_t* neighbors_index_dev,
                                const float* neighbors_importance_dev,
                                const int64_t* neighbors_row_splits_dev, int64_t num_out,
                                int cin, int cout, int normalize, const float* bias_dev,
                                int relu, float* out_dev);
/* Backward support ("next" row f4): per-output interpolation matrices basis_out [num_out, 4*4*4*cin] (filter cell
 * major, un-normalised: sum_p importance_p * trilinear_weight(cell, p) * feature_p[c]) and importance sums norm_out
 * [num_out]; the filter gradient is dW = basis^T (g / norm).  cin must be 4. */
int asr_hip_continuous_conv_basis_f32(asr_hip_context* ctx, const float* out_positions_dev, const float* extents_dev,
                                      const float* inp_positions_dev, const float* inp_features_dev,
                                      const int32_t* neighbors_index_dev, const float* neighbors_importance_dev,
                                      const int64_t* neighbors_row_splits_dev, int64_t num_out, int cin,
                                      float* basis_out_dev, float* norm_out_dev);
/* neighbors_importance = compat * clamp((1-d)^3,0,1) (models/common_torch.py:21-22,
 * net_definitions_torch.py:107) */
int asr_hip_aggregation_importance(asr_hip_context* ctx, const float* compat_dev,
                                   const float* dist_dev, int64_t num_pairs, float* out_dev);

/* ---- a12: SpecialSparseConv.forward / open3d::sparse_conv
 *      (models/common_torch.py:95-148) ------------------------------------------------- */
typedef struct asr_sparse_conv_args {
    const float* filters;              /* dev [K, cin, cout] row-major                       */
    const float* inp_features;         /* dev, row i at inp_features + i*inp_ld              */
    int64_t inp_ld;                    /* row stride in floats (>= cin)                      */
    const float* inp_importance;       /* dev [num_inp] or NULL: neighbour importance =
                                          inp_importance[neighbors_index] (:124-126)         */
    const float* neighbors_importance; /* dev [P] or NULL: per-pair importance as taken by
                                          open3d::sparse_conv; wins over inp_importance      */
    const int32_t* neighbors_index;    /* dev [P]                                            */
    const uint8_t* neighbors_kernel_index; /* dev [P]                                        */
    const int64_t* neighbors_row_splits;   /* dev [num_out+1]                                */
    int64_t num_out;
    int64_t num_inp;
    int kernel_size;                   /* 55 or 9                                            */
    int cin, cout;
    int normalize;                     /* divide by the importance sum (if != 0)             */
    const float* bias;                 /* dev [cout] or NULL (:144-145)                      */
    int relu;                          /* (:146)                                             */
    const float* residual;             /* dev or NULL: added after the activation
                                          (net_definitions_torch.py:633)                     */
    int64_t residual_ld;
    float* out;                        /* dev, row i at out + i*out_ld                       */
    int64_t out_ld;
    float* out_importance;             /* dev [num_out] or NULL: sum of neighbour importance
                                          (reduce_subarrays_sum, :127-128)                   */
    int algo;                          /* 0 = auto, 1 = scalar reference kernel, 2 = MFMA    */
    const int32_t* row_perm;           /* dev [num_out] or NULL: order in which the MFMA kernel
                                          tiles the output rows (asr_hip_row_groups); results
                                          do not depend on it                                */
    /* optional second filter bank, same gather, same launch (SparseConvBlock conv1a + conv1b,
     * models/v0/net_definitions_torch.py:262-281): output columns [cout, cout + cout_b).  With
     * filters_b != NULL the importance arrays, `normalize` and `out_importance` belong to bank b
     * only and bank a is the plain convolution.  MFMA path: cout % 16 == 8, cout_b == 8. */
    const float* filters_b;            /* dev [K, cin, cout_b] or NULL                       */
    const float* bias_b;               /* dev [cout_b] or NULL                               */
    int cout_b;
    /* 0 = chosen from the problem size.  Otherwise the MFMA kernel instance is forced: column tile
     * width force_nt * 16 (1, 2, 4, 8, 16) and force_waves * 16 rows per block (4 or 8) -- lets a
     * small input run the instances that large inputs select (parity tests). */
    int force_nt;
    int force_waves;
    /* 16-bit entry points only: row-group plan of this neighbour list (asr_hip_sparse_conv_plan_create), or NULL
     * to have a temporary one built for the call.  A plan is tied to (neighbour arrays, row_perm, num_out). */
    const struct asr_hip_conv_plan* plan;
    /* ASR_CONV16_F16X2 only: largest |element| of inp_features as f32 bits (device scalar), or NULL to have it
     * computed by one pass over the input. */
    const uint32_t* inp_absmax;
    /* 16-bit entry points: device scalar that receives max(its value, f32 bits of the largest |element| written to
     * `out`) -- the inp_absmax of the convolutions that read `out`; the caller zeroes it.  NULL: not kept. */
    uint32_t* out_absmax;
} asr_sparse_conv_args;
int asr_hip_sparse_conv_f32(asr_hip_context* ctx, const asr_sparse_conv_args* args);
/* Launch statistics of the MFMA sparse conv since the last reset: text "NT,KC,IMP,WAVES,DUAL:count;..."
 * (template instance k_sconv_mfma<NT,KC,IMP,WAVES,DUAL> -> number of launches), NUL terminated.  The 16-bit
 * kernels count as "NT,KC,IMP,WAVES,DUAL,MODE,PLAN" (PLAN 1: k_sconv_plan16, 0: k_sconv_mfma16). */
int asr_hip_sparse_conv_variant_counts(asr_hip_context* ctx, char* buf, size_t cap, int reset);

/* ---- a12 on the 16-bit matrix cores (v_mfma_f32_16x16x32_{f16,bf16}) --------------------------------
 * ASR_CONV16_F16 (BASELINE config C5, "fp16 features"): activations in HBM are f16, weights are rounded to
 *   f16 once, products accumulate in f32.  In args, inp_features / residual / out point to f16 data (out: f32
 *   when out_is_f16 == 0), leading dimensions count elements; importance, bias and out_importance stay f32.
 * ASR_CONV16_BF16X3: f32 activations and weights; every operand is split exactly into three bf16 terms and
 *   the product is evaluated with six bf16 MFMAs, f32 accumulate -- fp32-class results (error of the dropped
 *   terms < 2^-23 per product) at 2.7x the f32 matrix peak.  args as for asr_hip_sparse_conv_f32.
 * ASR_CONV16_F16X2: f32 activations and weights, fp32-class results from HALF the MFMAs of bf16x3: each tensor is
 *   scaled by a power of two (its largest magnitude to [2^14, 2^15): exact, no f16 overflow or underflow worth
 *   speaking of) and split into two f16 terms, a*b = a0*b0 + a1*b0 + a0*b1 (three f16 MFMAs, f32 accumulate).
 *   Error per product < 2^-21 for elements within 2^-17 of their tensor's maximum, < 2^-38 of the product of the
 *   maxima otherwise.  The activations' scale comes from args->inp_absmax, which the producing convolution keeps
 *   through args->out_absmax (asr_hip_absmax_f32 for other producers).  args as for asr_hip_sparse_conv_f32.
 * All take the filters re-packed by asr_hip_sparse_conv_pack (16-bit, [plane][K][cin panel][cout padded to
 * 16][panel depth] in the kernels' LDS order; bank b appended as columns); args->filters / filters_b are
 * ignored, cout_b > 0 selects the two-bank form.  cin and the row strides must be multiples of 8 (f16) / 4 (f32) elements.
 * A row may name every kernel slot at most once (true for all lists of the reference's grids, cpp/lib/grid.cpp:99-170,
 * 229-240); a list that does not is refused with ASR_HIP_EINVAL when its plan is built. */
#define ASR_CONV16_F16 1
#define ASR_CONV16_BF16X3 2
#define ASR_CONV16_F16X2 3
size_t asr_hip_sparse_conv_packed_bytes(int mode, int kernel_size, int cin, int cout, int cout_b);
int asr_hip_sparse_conv_pack(asr_hip_context* ctx, int mode, const float* filters_dev, const float* filters_b_dev,
                             int kernel_size, int cin, int cout, int cout_b, void* packed_out_dev);
int asr_hip_sparse_conv_f16(asr_hip_context* ctx, const asr_sparse_conv_args* args, const void* packed_dev,
                            int out_is_f16);
int asr_hip_sparse_conv_bf16x3(asr_hip_context* ctx, const asr_sparse_conv_args* args, const void* packed_dev);
int asr_hip_sparse_conv_f16x2(asr_hip_context* ctx, const asr_sparse_conv_args* args, const void* packed_dev);
/* f32 bits of the largest |element| of a [rows, cols] matrix with row stride ld (floats) into *out_dev */
int asr_hip_absmax_f32(asr_hip_context* ctx, const float* x_dev, int64_t rows, int cols, int64_t ld, uint32_t* out_dev);
/* Row-group plan: the neighbour list re-laid in the order the 16-bit kernels stream it -- per 16 consecutive
 * rows (row_perm order) the set of kernel slots in use and, slot by slot, the 16 neighbour indices.  Built once
 * per list and reused by every convolution over it (the U-Net runs ~10 per grid level); the kernels then need no
 * neighbour table in LDS and do no CSR parsing.  The plan owns its device memory; the arrays it was built from
 * must keep their contents while it is in use.  Per-pair importance (neighbors_importance) and neighbour-count
 * normalisation are served by the table-driven kernel, which ignores the plan. */
typedef struct asr_hip_conv_plan asr_hip_conv_plan;
int asr_hip_sparse_conv_plan_create(asr_hip_context* ctx, const int32_t* neighbors_index_dev,
                                    const uint8_t* neighbors_kernel_index_dev, const int64_t* neighbors_row_splits_dev,
                                    const int32_t* row_perm_dev, int64_t num_out, int kernel_size,
                                    asr_hip_conv_plan** plan_out);
void asr_hip_sparse_conv_plan_destroy(asr_hip_conv_plan* plan);
/* With context option "plan_arena" = 1 the plans created afterwards take their memory from an arena of the context
 * instead of owning it (no device allocation per plan); this call invalidates ALL of them at once and recycles the
 * memory (hosts that rebuild every plan per geometry, e.g. the one-scan sharding). */
int asr_hip_context_plan_arena_reset(asr_hip_context* ctx);
/* bytes of device memory a plan holds */
size_t asr_hip_sparse_conv_plan_bytes(const asr_hip_conv_plan* plan);
/* f32 <-> f16 conversion of an activation buffer (n elements) */
int asr_hip_convert_f16(asr_hip_context* ctx, const void* in_dev, int64_t n, void* out_dev, int to_f16);

/* MFMA tiling order for asr_hip_sparse_conv_f32: reorders the rows of a CSR inside segments of
 * `segment_rows` consecutive rows (0 = default) by their set of kernel slots, so that 16-row MFMA
 * tiles see few distinct slots. perm_out_dev [num_rows] goes into asr_sparse_conv_args.row_perm. */
int asr_hip_row_groups(asr_hip_context* ctx, const uint8_t* kernel_index_dev,
                       const int64_t* row_splits_dev, int64_t num_rows, int64_t segment_rows,
                       int32_t* perm_out_dev);

/* ---- a11: open3d::invert_neighbors_list (net_definitions_torch.py:22-36,548-559) -------- */
int asr_hip_invert_neighbors_list(asr_hip_context* ctx, int64_t num_points,
                                  const int32_t* inp_index_dev,
                                  const int64_t* inp_row_splits_dev, int64_t num_rows,
                                  const uint8_t* inp_attributes_dev, int32_t* out_index_dev,
                                  int64_t* out_row_splits_dev, uint8_t* out_attributes_dev);

/* ---- open3d::reduce_subarrays_sum (models/common_torch.py:127) -------------------------- */
/* gather_index may be NULL; otherwise out[i] = sum values[gather_index[p]] */
int asr_hip_reduce_subarrays_sum(asr_hip_context* ctx, const float* values_dev,
                                 const int32_t* gather_index_dev,
                                 const int64_t* row_splits_dev, int64_t num_rows,
                                 float* out_dev);

/* ---- a14: UNet5.decode with zero shifts + sdf scale
 *      (net_definitions_torch.py:655-666, cpp/lib/asr.cpp:324-336) ------------------------ */
/* w1 [h1, 3+c], b1 [h1], w2 [h2,h1], b2 [h2], w3 [2,h2] (torch Linear layout);
 * sizes may be NULL (no sdf scale). out [v,2]. */
int asr_hip_decode_mlp(asr_hip_context* ctx, const float* code_dev, int64_t v, int c,
                       const float* w1_dev, const float* b1_dev, int h1, const float* w2_dev,
                       const float* b2_dev, int h2, const float* w3_dev,
                       const float* sizes_dev, float* out_dev);

/* ---- whole path: the section of asr::ReconstructSurface between the pre-filter and the
 *      contouring (cpp/lib/asr.cpp:143-336) ---------------------------------------------- */
typedef struct asr_weight {
    const char* name;   /* state_dict name of UNet5, e.g. "sparseconv_encblock0.conv1a.kernel" */
    const float* data;  /* dev                                                                */
    int32_t ndim;
    int64_t shape[5];
} asr_weight;

typedef struct asr_implicit_params {
    float point_radius_scale; /* asr.hpp:55, default 1                                      */
    int octree_max_depth;     /* asr.hpp:63, default 21                                     */
    float bb_min[3];          /* bounding box handed to CreateOctreeFromPoints              */
    float bb_max[3];
    int scale_sdf;            /* 1: values[:,0] *= voxel_size (asr.cpp:334-336)             */
    int precision;            /* arithmetic of the 53 sparse convs: 0 = exact f32 MFMA (default),
                                 ASR_CONV16_F16 = f16 activations + weights (config C5),
                                 ASR_CONV16_BF16X3 = fp32-class result on the bf16 matrix cores (six MFMAs per
                                 product), ASR_CONV16_F16X2 = fp32-class result on the f16 matrix cores
                                 (three MFMAs per product, per-tensor power-of-two scaling)             */
} asr_implicit_params;

/* sizes of the structures built by the last asr_hip_implicit_* call */
typedef struct asr_implicit_sizes {
    int64_t num_points;
    int64_t num_nodes;
    int64_t num_voxels[ASR_NUM_GRIDS];
    int64_t num_pairs[ASR_NUM_GRIDS];
    int64_t num_agg_pairs;
} asr_implicit_sizes;

/* geometry half: octree, 5 grids, aggregation neighbours. Results stay in the context. */
int asr_hip_implicit_build(asr_hip_context* ctx, const float* points_dev,
                           const float* radii_dev, int64_t n,
                           const asr_implicit_params* params, asr_implicit_sizes* sizes);
/* network half: aggregate + unet + decode on the structures of the last build.
 * values_out_dev [num_voxels[0], 2]. */
int asr_hip_implicit_network(asr_hip_context* ctx, const float* points_dev,
                             const float* normals_dev, int64_t n, const asr_weight* weights,
                             int num_weights, const asr_implicit_params* params,
                             float* values_out_dev);
/* first stage of the network half alone (UNet5.aggregate, net_definitions_torch.py:640-653): "feats1" [V0, C] and the
 * per-pair "importance" of the last build, readable with asr_hip_implicit_get.  For hosts that run the U-Net themselves
 * (the one-scan sharding: every rank aggregates the whole cloud, then convolves the rows it owns). */
int asr_hip_implicit_aggregate(asr_hip_context* ctx, const float* points_dev, const float* normals_dev, int64_t n,
                               const asr_weight* weights, int num_weights, const asr_implicit_params* params);
/* both halves; values live in the context afterwards (asr_hip_implicit_get "values") */
int asr_hip_implicit_forward(asr_hip_context* ctx, const float* points_dev,
                             const float* normals_dev, const float* radii_dev, int64_t n,
                             const asr_weight* weights, int num_weights,
                             const asr_implicit_params* params, asr_implicit_sizes* sizes);
/* ---- one scan over several GPUs, inside the library (round 4; SURVEY 8(e), BASELINE config C4) ----------------------
 * The reference has no multi-device path (cpp/lib/asr.cpp:161-163 creates CPU tensors); what defines the halo is the
 * stencil of its operators: one face ring per 55-slot convolution on the same / child / parent level
 * (cpp/lib/grid.cpp:99-170) and parent <-> children for the transitions (cpp/lib/grid.cpp:206-242).
 *
 * asr_hip_implicit_forward_sharded: every rank holds the whole cloud.  Option "shard_geometry" (default -1 = 1 whenever
 * world > 1): 1 = octree, voxel keys and the one-entry-per-voxel up / down lists of the whole cloud on every rank (the
 * integer work ownership is derived from), 55-slot neighbour lists, row-group plans, aggregation search and continuous
 * conv for the voxels the rank owns only; 0 = geometry + aggregation of the whole cloud on every rank (replicated).
 * After a forward with per-rank geometry the context holds a PARTIAL build ("neighbors_*", "aggregation_*" of
 * asr_hip_implicit_get list the owned rows, sizes.num_pairs / num_agg_pairs are the rank's): asr_hip_implicit_network
 * and asr_hip_implicit_aggregate refuse it (ASR_HIP_EINVAL) until the next asr_hip_implicit_build.
 * The 53 sparse convolutions and the decoder run on the rows the rank OWNS (grid-0 voxels
 * cut into `world` contiguous ranges of their level-21 Morton order with equal pair counts, a coarser voxel belongs to
 * the owner of its first child), with one point-to-point exchange of the boundary rows of the input buffer (+ the
 * importance of those rows) before each convolution -- grouped with the MAX all-reduce of the f16x2 running maximum of that
 * buffer --, and an all-gather of the owned values at the end.  Per row the same kernel, plan-group arithmetic and summation order as on one GPU:
 * the values equal asr_hip_implicit_forward's bit for bit.
 *
 * Transport: a table of two collective primitives on DEVICE buffers, enqueued on (or synchronised with) `stream`.
 * asr_hip_shard_comm_rccl_* provides it over RCCL (librccl.so is loaded at run time); tests plug in a host-staged one.
 */
typedef struct asr_shard_comm {
    void* user;
    int rank, world;
    /* grouped point-to-point exchange: message i goes to / comes from peer[i]; all of one call may proceed
     * concurrently (ncclGroupStart .. ncclGroupEnd).  Buffers are device memory.  0 = ok. */
    int (*exchange)(void* user, int nsend, const int* send_peer, const void* const* send_buf, const size_t* send_bytes,
                    int nrecv, const int* recv_peer, void* const* recv_buf, const size_t* recv_bytes, void* stream);
    /* in-place MAX over the ranks of n uint32 values (f16x2 running maxima: non-negative f32 bit patterns) */
    int (*allreduce_max_u32)(void* user, uint32_t* buf_dev, size_t n, void* stream);
    /* optional (NULL: the two calls above, one after the other): both in ONE group -- the maximum of a convolution's
     * input tensor and the halo of that tensor are needed at the same moment, before the convolution starts */
    int (*exchange_and_max)(void* user, int nsend, const int* send_peer, const void* const* send_buf,
                            const size_t* send_bytes, int nrecv, const int* recv_peer, void* const* recv_buf,
                            const size_t* recv_bytes, uint32_t* max_buf_dev, size_t max_n, void* stream);
} asr_shard_comm;

/* RCCL transport.  unique_id_out / unique_id: the 128 bytes of ncclUniqueId (rank 0 creates, the host broadcasts). */
int asr_hip_shard_comm_rccl_unique_id(asr_hip_context* ctx, void* unique_id_out);
int asr_hip_shard_comm_rccl_create(asr_hip_context* ctx, const void* unique_id, int rank, int world,
                                   asr_shard_comm** comm_out);
void asr_hip_shard_comm_rccl_destroy(asr_shard_comm* comm);

typedef struct asr_shard_stats {
    int64_t owned_rows[ASR_NUM_GRIDS];
    int64_t halo_rows_recv[ASR_NUM_GRIDS]; /* 55-slot lists: rows received per application of the level's stencil */
    int64_t bytes_sent, bytes_received;    /* halo exchanges + stitch of the last forward                       */
    int64_t exchanges;                     /* grouped exchanges of the last forward                              */
    double exchange_seconds;               /* option "shard_timing": wall time of the exchanges, each bracketed by stream
                                              synchronisations (instrumented runs only; 0 otherwise)             */
} asr_shard_stats;

/* values_out_dev [num_voxels[0], 2] complete on every rank (NULL: stays in the context, "values") */
int asr_hip_implicit_forward_sharded(asr_hip_context* ctx, const asr_shard_comm* comm, const float* points_dev,
                                     const float* normals_dev, const float* radii_dev, int64_t n,
                                     const asr_weight* weights, int num_weights, const asr_implicit_params* params,
                                     asr_implicit_sizes* sizes, float* values_out_dev, asr_shard_stats* stats);
/* copies one of the arrays of the last build/forward into dst_dev (device to device).
 * name: "values", "feats1", "importance", "code",
 *       "voxel_keys<i>", "voxel_centers<i>", "voxel_sizes<i>", "neighbors_index<i>",
 *       "neighbors_kernel_index<i>", "neighbors_row_splits<i>", "up_neighbors_index<i>",
 *       "up_neighbors_kernel_index<i>", "up_neighbors_row_splits<i>",
 *       "aggregation_neighbors_index", "aggregation_neighbors_dist",
 *       "aggregation_row_splits", "aggregation_scale_compat", "nodes",
 *       "down_neighbors_{index,kernel_index,row_splits}<i>" (inverted up lists, rows = grid i+1),
 *       "tiling<i>", "tiling_up<i>", "tiling_down<i>" (int32 MFMA tiling orders of the three CSRs)
 * (the input_dict keys of cpp/lib/asr.cpp:159-312). nbytes returns the byte size; dst_dev may
 * be NULL to query only. */
int asr_hip_implicit_get(asr_hip_context* ctx, const char* name, void* dst_dev,
                         size_t* nbytes);
/* per-stage times (ms, hip events) of the last forward: [0] octree, [1] grids, [2] aggregation
 * search, [3] continuous conv, [4] unet, [5] decode, [6] geometry wall (octree start .. search
 * joined), [7] network wall.  With option "overlap" (default) the search runs on a second stream
 * concurrently with the grids: [2] is measured on that stream and [0]+[1]+[2] exceeds [6]. */
int asr_hip_implicit_stage_ms(asr_hip_context* ctx, float out_ms[8]);

#ifdef __cplusplus
}
#endif
#endif /* ASR_HIP_H */

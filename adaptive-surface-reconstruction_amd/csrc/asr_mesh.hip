// asr_mesh.hip -- dual contouring and component filter on the GPU ("next" rows D.2 / D.3).
//
// Replaces asr::CreateTriangleMesh (cpp/lib/contouring.cpp:29-460) and
// asr::RemoveConnectedComponents (cpp/lib/postprocess.cpp:27-201).  The reference walks the dual
// cells serially; its output order is a pure function of the inputs (vertices in dual order, fan
// centres appended in emission order, triangles in (dual, owned edge) order), so every stage here
// is a data-parallel map + exclusive scan that lands each item at the index the serial loop gives.
// Compiled with -ffp-contract=off: the crossing points are double sums that must round like the
// reference's.
#include <cstring>
#include <algorithm>

#include "asr_common.h"
#include "asr_prim.h"
#include "asr_uset.h"

namespace {
using namespace asr_prim;

constexpr int BLK = 256;

// contouring.cpp:53-79
__constant__ int c_cube_edges[12][2] = {{0, 1}, {1, 3}, {3, 2}, {2, 0}, {4, 5}, {5, 7},
                                        {7, 6}, {6, 4}, {0, 4}, {1, 5}, {3, 7}, {2, 6}};
__constant__ int c_cube_faces[6][4] = {{0, 1, 3, 2}, {4, 6, 7, 5}, {1, 5, 7, 3},
                                       {2, 3, 7, 6}, {0, 2, 6, 4}, {0, 4, 5, 1}};
__constant__ int c_owned_edges[3][2] = {{0, 1}, {1, 3}, {1, 5}};  // edge subset {0,1,9}

struct MeshState {
    int kind = 0;  // 1 = contour, 2 = components
    // contour
    const float* values = nullptr;
    const i64* duals = nullptr;
    i64 num_values = 0, num_duals = 0, num_active = 0, num_extra = 0, num_tri = 0;
    float thr = 0;
    int32_t* active = nullptr;  // vertex -> dual
    float* vtx = nullptr;       // [num_active,3]
    i64* adj_rs = nullptr;      // voxel -> active duals (vertex indices, ascending)
    int32_t* adj = nullptr;
    i64* tri_off = nullptr;  // per (vertex, owned edge)
    i64* extra_off = nullptr;
    // components
    const float* in_vtx = nullptr;
    const int32_t* in_tri = nullptr;
    i64 nv = 0, nt = 0, nv_out = 0, nt_out = 0;
    i64* v_off = nullptr;  // exclusive scan of the vertex keep flags (nv+1)
    i64* t_off = nullptr;  // same for triangles
};

__device__ inline bool crossing(const float* values, float thr, i64 a, i64 b) {  // :81-111
    const float2 va = ((const float2*)values)[a], vb = ((const float2*)values)[b];
    if (va.y > thr && vb.y > thr) return false;
    return (va.x < 0 && vb.x > 0) || (va.x > 0 && vb.x < 0);
}

// one thread per dual cell: active flag and number of distinct corner voxels
__global__ void k_contour_active(const float* values, const i64* duals, i64 nd, float thr, i64* flag,
                                 i64* npairs) {
    i64 d = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (d > nd) return;
    if (d == nd) {
        flag[d] = 0;
        npairs[d] = 0;
        return;
    }
    i64 c[8];
    for (int k = 0; k < 8; ++k) c[k] = duals[d * 8 + k];
    bool act = false;
    for (int e = 0; e < 12; ++e) act |= crossing(values, thr, c[c_cube_edges[e][0]], c[c_cube_edges[e][1]]);
    int distinct = 0;
    for (int k = 0; k < 8; ++k) {
        bool first = true;
        for (int j = 0; j < k; ++j) first &= c[j] != c[k];
        distinct += first;
    }
    flag[d] = act ? 1 : 0;
    npairs[d] = act ? distinct : 0;
}

// vertex of an active dual (:114-142) + its (voxel, vertex) adjacency pairs
__global__ void k_contour_vertices(const float* values, const i64* duals, i64 nd, const float* pos, float thr,
                                   const i64* voff, const i64* poff, int32_t* active, float* vtx, u64* pairs) {
    i64 d = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= nd) return;
    const i64 vid = voff[d];
    if (voff[d + 1] == vid) return;
    active[vid] = (int32_t)d;
    i64 c[8];
    for (int k = 0; k < 8; ++k) c[k] = duals[d * 8 + k];
    double px = 0, py = 0, pz = 0;
    int count = 0;
    for (int e = 0; e < 12; ++e) {
        const i64 a = c[c_cube_edges[e][0]], b = c[c_cube_edges[e][1]];
        const float2 va = ((const float2*)values)[a], vb = ((const float2*)values)[b];
        if (va.y > thr && vb.y > thr) continue;
        const double v1 = va.x, v2 = vb.x;
        if ((v1 < 0 && v2 > 0) || (v1 > 0 && v2 < 0)) {
            double t = -v1 / (v2 - v1);
            if (!isfinite(t) || t < 0 || t > 1) t = 0.5;
            px += (1 - t) * (double)pos[a * 3 + 0] + t * (double)pos[b * 3 + 0];
            py += (1 - t) * (double)pos[a * 3 + 1] + t * (double)pos[b * 3 + 1];
            pz += (1 - t) * (double)pos[a * 3 + 2] + t * (double)pos[b * 3 + 2];
            ++count;
        }
    }
    vtx[vid * 3 + 0] = (float)(px / count);
    vtx[vid * 3 + 1] = (float)(py / count);
    vtx[vid * 3 + 2] = (float)(pz / count);
    i64 o = poff[d];
    for (int k = 0; k < 8; ++k) {
        bool first = true;
        for (int j = 0; j < k; ++j) first &= c[j] != c[k];
        if (first) pairs[o++] = ((u64)c[k] << 32) | (u64)vid;
    }
}

// row splits of the sorted (voxel, vertex) pairs by binary search + payload extraction
__global__ void k_adj_splits(const u64* pairs, i64 np, i64 nv, i64* rs) {
    i64 v = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (v > nv) return;
    const u64 key = (u64)v << 32;
    i64 lo = 0, hi = np;
    while (lo < hi) {
        i64 mid = (lo + hi) >> 1;
        if (pairs[mid] < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    rs[v] = lo;
}
__global__ void k_adj_payload(const u64* pairs, i64 np, int32_t* adj) {
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < np) adj[i] = (int32_t)(u32)pairs[i];
}

// active duals containing both voxels (:202-213): intersection of two ascending lists, in
// ascending order (== the insertion order into the reference's unordered_set)
__device__ inline int edge_duals(const i64* rs, const int32_t* adj, i64 a, i64 b, u32* out, int cap) {
    i64 i = rs[a], ie = rs[a + 1], j = rs[b], je = rs[b + 1];
    int n = 0;
    while (i < ie && j < je) {
        const int32_t x = adj[i], y = adj[j];
        if (x == y) {
            if (n < cap) out[n] = (u32)x;
            ++n;
            ++i;
            ++j;
        } else if (x < y)
            ++i;
        else
            ++j;
    }
    return n;
}

struct Face4 {  // smallset.h: sorted, duplicate free
    i64 d[4];
    int n;
};
__device__ inline void face_insert(Face4& f, i64 v) {
    int i = 0;
    while (i < f.n && f.d[i] < v) ++i;
    if (i < f.n && f.d[i] == v) return;
    for (int j = f.n; j > i; --j) f.d[j] = f.d[j - 1];
    f.d[i] = v;
    ++f.n;
}
__device__ inline Face4 face_of(const i64* c, int fi) {
    Face4 f;
    f.n = 0;
    for (int k = 0; k < 4; ++k) face_insert(f, c[c_cube_faces[fi][k]]);
    return f;
}
__device__ inline bool face_eq(const Face4& a, const Face4& b) {
    if (a.n != b.n) return false;
    for (int i = 0; i < a.n; ++i)
        if (a.d[i] != b.d[i]) return false;
    return true;
}
// :216-232
__device__ inline Face4 face_with_oriented_edge(const i64* c, i64 e0, i64 e1) {
    for (int fi = 0; fi < 6; ++fi)
        for (int j = 0; j < 4; ++j)
            if (c[c_cube_faces[fi][j]] == e0 && c[c_cube_faces[fi][(j + 1) & 3]] == e1) {
                Face4 f = face_of(c, fi);
                if (f.n >= 3) return f;
            }
    Face4 none;
    none.n = 0;
    return none;
}
__device__ inline bool dual_has_face(const i64* c, const Face4& face) {  // :235-244
    for (int fi = 0; fi < 6; ++fi)
        if (face_eq(face_of(c, fi), face)) return true;
    return false;
}
__device__ inline void load_dual(const i64* duals, const int32_t* active, u32 vid, i64* c) {
    const i64 d = active[vid];
    for (int k = 0; k < 8; ++k) c[k] = duals[d * 8 + k];
}

// One thread per (active dual, owned edge).  COUNT: number of triangles / fan centres it emits.
// FILL: orders the duals around the edge (:247-299) and writes the triangles (:364-451).
template <bool FILL>
__global__ __launch_bounds__(BLK) void k_contour_edges(const float* values, const i64* duals, float thr,
                                                       const int32_t* active, i64 na, const i64* rs,
                                                       const int32_t* adj, i64* tri_cnt, i64* extra_cnt,
                                                       const i64* tri_off, const i64* extra_off, float* vtx,
                                                       int32_t* tri, int* flags) {
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (!FILL && t == na * 3) {
        tri_cnt[t] = 0;
        extra_cnt[t] = 0;
    }
    if (t >= na * 3) return;
    const i64 vid = t / 3;
    const int e = (int)(t - vid * 3);
    const i64 d = active[vid];
    i64 a = duals[d * 8 + c_owned_edges[e][0]], b = duals[d * 8 + c_owned_edges[e][1]];
    u32 xs[ASR_USET_CAP];
    int n = 0;
    if (a != b && crossing(values, thr, a, b)) n = edge_duals(rs, adj, a, b, xs, ASR_USET_CAP);
    if (!FILL) {
        tri_cnt[t] = n == 3 ? 1 : (n == 4 ? 2 : (n > 4 ? n : 0));
        extra_cnt[t] = n > 4 ? 1 : 0;
        if (n > ASR_USET_CAP) atomicOr(&flags[2], 1);
        return;
    }
    if (n < 3 || n > ASR_USET_CAP) return;
    if (values[a * 2] > values[b * 2]) {  // :357-358 orient from the lower to the higher value
        i64 s = a;
        a = b;
        b = s;
    }
    u32 rest[ASR_USET_CAP], sorted[ASR_USET_CAP];
    asr_uset_order(xs, n, rest);
    int nrest = n - 1, ns = 1;
    sorted[0] = rest[n - 1];
    bool reverse_again = false;
    for (int it = 0; it < n * n && nrest > 0; ++it) {
        i64 c1[8];
        load_dual(duals, active, sorted[ns - 1], c1);
        const Face4 face = face_with_oriented_edge(c1, a, b);
        int found = -1;
        for (int j = 0; j < nrest && found < 0; ++j) {
            i64 c2[8];
            load_dual(duals, active, rest[j], c2);
            if (dual_has_face(c2, face)) found = j;
        }
        if (found >= 0) {
            sorted[ns++] = rest[found];
            for (int j = found; j + 1 < nrest; ++j) rest[j] = rest[j + 1];
            --nrest;
        } else {
            for (int j = 0; j < ns / 2; ++j) {
                u32 s = sorted[j];
                sorted[j] = sorted[ns - 1 - j];
                sorted[ns - 1 - j] = s;
            }
            i64 s = a;
            a = b;
            b = s;
            reverse_again = !reverse_again;
        }
    }
    if (reverse_again)
        for (int j = 0; j < ns / 2; ++j) {
            u32 s = sorted[j];
            sorted[j] = sorted[ns - 1 - j];
            sorted[ns - 1 - j] = s;
        }
    if (ns != n) {  // "this should not happen: cannot sort duals" (:366-370)
        atomicOr(&flags[3], 1);
        return;
    }
    int32_t* out = tri + tri_off[t] * 3;
    if (n == 3) {
        out[0] = (int32_t)sorted[0];
        out[1] = (int32_t)sorted[1];
        out[2] = (int32_t)sorted[2];
    } else if (n == 4) {
        float p[4][3];
        for (int i = 0; i < 4; ++i)
            for (int q = 0; q < 3; ++q) p[i][q] = vtx[(i64)sorted[i] * 3 + q];
        // Eigen's unrolled sum for a 3-vector: x*x + (y*y + z*z)
        const float d0x = p[0][0] - p[2][0], d0y = p[0][1] - p[2][1], d0z = p[0][2] - p[2][2];
        const float d1x = p[1][0] - p[3][0], d1y = p[1][1] - p[3][1], d1z = p[1][2] - p[3][2];
        const float q02 = d0x * d0x + (d0y * d0y + d0z * d0z);
        const float q13 = d1x * d1x + (d1y * d1y + d1z * d1z);
        if (q02 > q13) {
            out[0] = (int32_t)sorted[0]; out[1] = (int32_t)sorted[1]; out[2] = (int32_t)sorted[3];
            out[3] = (int32_t)sorted[1]; out[4] = (int32_t)sorted[2]; out[5] = (int32_t)sorted[3];
        } else {
            out[0] = (int32_t)sorted[0]; out[1] = (int32_t)sorted[1]; out[2] = (int32_t)sorted[2];
            out[3] = (int32_t)sorted[0]; out[4] = (int32_t)sorted[2]; out[5] = (int32_t)sorted[3];
        }
    } else {
        float cx = 0, cy = 0, cz = 0;
        for (int i = 0; i < n; ++i) {
            cx += vtx[(i64)sorted[i] * 3 + 0];
            cy += vtx[(i64)sorted[i] * 3 + 1];
            cz += vtx[(i64)sorted[i] * 3 + 2];
        }
        const i64 ci = na + extra_off[t];
        vtx[ci * 3 + 0] = cx / (float)n;
        vtx[ci * 3 + 1] = cy / (float)n;
        vtx[ci * 3 + 2] = cz / (float)n;
        for (int i = 0; i < n; ++i) {
            out[i * 3 + 0] = (int32_t)sorted[i];
            out[i * 3 + 1] = (int32_t)sorted[(i + 1) % n];
            out[i * 3 + 2] = (int32_t)ci;
        }
    }
}

// ------------------------------------------------------------------------------------------
// connected components: lock-free union-find, the root of a set is its smallest vertex, so the
// rank of a root among the roots is the label the reference's DFS over i = 0..nv-1 assigns
// (postprocess.cpp:57-80)
// ------------------------------------------------------------------------------------------
__device__ inline int uf_find(int* parent, int x) {
    while (true) {
        int p = __atomic_load_n(&parent[x], __ATOMIC_RELAXED);
        if (p == x) return x;
        int gp = __atomic_load_n(&parent[p], __ATOMIC_RELAXED);
        if (gp != p) __atomic_store_n(&parent[x], gp, __ATOMIC_RELAXED);  // path halving
        x = p;
    }
}
__device__ inline void uf_union(int* parent, int a, int b) {
    while (true) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a < b) {
            int s = a;
            a = b;
            b = s;
        }
        if (atomicCAS(&parent[a], a, b) == a) return;  // hang the larger root below the smaller
    }
}
__global__ void k_uf_init(int* parent, i64 n) {
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) parent[i] = (int)i;
}
__global__ void k_uf_link(const int32_t* tri, i64 nt, i64 nv, int* parent, int* flags) {
    i64 f = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nt) return;
    const int a = tri[f * 3], b = tri[f * 3 + 1], c = tri[f * 3 + 2];
    if (a < 0 || b < 0 || c < 0 || a >= nv || b >= nv || c >= nv) {
        atomicOr(&flags[4], 1);
        return;
    }
    uf_union(parent, a, b);
    uf_union(parent, b, c);
}
__global__ void k_uf_roots(int* parent, i64 n, i64* is_root) {
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) {
        is_root[i] = 0;
        return;
    }
    const int r = uf_find(parent, (int)i);
    is_root[i] = r == (int)i ? 1 : 0;
}
// after k_uf_roots all paths are short; flatten and count members per component label
__global__ void k_comp_sizes(int* parent, i64 n, const i64* label_of_root, int32_t* comp, int* sizes) {
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;  // (exited lanes are inactive below: readfirstlane / ballot see active lanes only)
    const int r = uf_find(parent, (int)i);
    const int c = (int)label_of_root[r];
    comp[i] = c;
    // one atomic per distinct label in the wave (a mesh is mostly one component: 10^6 atomics on one
    // address would take 12 ms)
    bool todo = true;
    while (todo) {
        const int lead = __builtin_amdgcn_readfirstlane(c);
        const unsigned long long same = __ballot(c == lead);
        if (c == lead) {
            if ((int)(threadIdx.x & 63) == __ffsll((long long)same) - 1) atomicAdd(&sizes[lead], __popcll(same));
            todo = false;
        }
    }
}
__global__ void k_comp_keys(const int* sizes, i64 nc, u64* keys) {
    i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < nc) keys[c] = ((u64)(u32)sizes[c] << 32) | (u64)c;  // std::greater on (size, label)
}
__global__ void k_comp_keep(const u64* sorted_desc, i64 nc, i64 keep_n, i64 min_size, uint8_t* keep) {
    i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nc) return;
    const u64 k = sorted_desc[r];
    keep[(u32)k] = (r < keep_n && (i64)(k >> 32) >= min_size) ? 1 : 0;
}
__global__ void k_vertex_keep(const int32_t* comp, const uint8_t* keep, i64 nv, i64* flag) {
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > nv) return;
    flag[i] = (i < nv && keep[comp[i]]) ? 1 : 0;
}
__global__ void k_tri_keep(const int32_t* tri, i64 nt, const i64* voff, i64* flag) {
    i64 f = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (f > nt) return;
    bool k = false;
    if (f < nt) {
        k = true;
        for (int q = 0; q < 3; ++q) {
            const i64 v = tri[f * 3 + q];
            k &= voff[v + 1] > voff[v];
        }
    }
    flag[f] = k ? 1 : 0;
}
__global__ void k_compact_vertices(const float* in, i64 nv, const i64* voff, float* out) {
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv || voff[i + 1] == voff[i]) return;
    const i64 o = voff[i];
    out[o * 3 + 0] = in[i * 3 + 0];
    out[o * 3 + 1] = in[i * 3 + 1];
    out[o * 3 + 2] = in[i * 3 + 2];
}
__global__ void k_compact_triangles(const int32_t* in, i64 nt, const i64* voff, const i64* toff, int32_t* out) {
    i64 f = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nt || toff[f + 1] == toff[f]) return;
    const i64 o = toff[f];
    for (int q = 0; q < 3; ++q) out[o * 3 + q] = (int32_t)voff[in[f * 3 + q]];
}

MeshState& mstate(asr_hip_context* ctx) {
    if (!ctx->mesh_state) ctx->mesh_state = new MeshState();
    return *(MeshState*)ctx->mesh_state;
}

#define MESH_ALLOC(var, T, count)                                                    \
    T* var = arena_alloc<T>(ctx->scratch, (size_t)(count));                          \
    if (!var) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed")

}  // namespace

void asr_mesh_release(asr_hip_context* ctx) {
    delete (MeshState*)ctx->mesh_state;
    ctx->mesh_state = nullptr;
}

int asr_mesh_contour_count(asr_hip_context* ctx, const float* values, i64 num_values, const i64* duals,
                           i64 num_duals, const float* positions, float threshold, i64* num_vertices,
                           i64* num_triangles) {
    MeshState& st = mstate(ctx);
    st = MeshState();
    *num_vertices = 0;
    *num_triangles = 0;
    if (num_duals <= 0 || num_values <= 0) {
        st.kind = 1;
        return ASR_HIP_OK;
    }
    if (num_values >= (i64(1) << 31) || num_duals >= (i64(1) << 31))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "contour: more than 2^31 voxels or dual cells");
    ASR_TRY(ensure_flags(ctx));
    ctx->scratch.reset();
    hipStream_t s = ctx->stream;
    ASR_TRY(fresh_flags(ctx));
    MESH_ALLOC(flag, i64, num_duals + 1);
    MESH_ALLOC(npairs, i64, num_duals + 1);
    MESH_ALLOC(voff, i64, num_duals + 1);
    MESH_ALLOC(poff, i64, num_duals + 1);
    k_contour_active<<<grid_for(num_duals + 1, BLK), BLK, 0, s>>>(values, duals, num_duals, threshold, flag,
                                                                   npairs);
    ASR_CHECK_LAUNCH(ctx);
    ASR_TRY(scan_counts(ctx, ctx->scratch, flag, voff, num_duals + 1));
    ASR_TRY(scan_counts(ctx, ctx->scratch, npairs, poff, num_duals + 1));
    i64 na = 0, np = 0;
    ASR_TRY(read_i64(ctx, voff + num_duals, &na));
    ASR_TRY(read_i64(ctx, poff + num_duals, &np));
    st.kind = 1;
    st.values = values;
    st.duals = duals;
    st.num_values = num_values;
    st.num_duals = num_duals;
    st.thr = threshold;
    st.num_active = na;
    if (na == 0) return ASR_HIP_OK;
    MESH_ALLOC(active, int32_t, na);
    MESH_ALLOC(pairs, u64, np);
    MESH_ALLOC(pairs_sorted, u64, np);
    MESH_ALLOC(adj_rs, i64, num_values + 1);
    MESH_ALLOC(adj, int32_t, np);
    MESH_ALLOC(tri_cnt, i64, na * 3 + 1);
    MESH_ALLOC(extra_cnt, i64, na * 3 + 1);
    MESH_ALLOC(tri_off, i64, na * 3 + 1);
    MESH_ALLOC(extra_off, i64, na * 3 + 1);
    // vertices: room for one fan centre per (dual, edge) is far too much; sized after the count
    MESH_ALLOC(vtx0, float, na * 3);
    k_contour_vertices<<<grid_for(num_duals, BLK), BLK, 0, s>>>(values, duals, num_duals, positions, threshold,
                                                                voff, poff, active, vtx0, pairs);
    ASR_CHECK_LAUNCH(ctx);
    ASR_TRY(sort_keys(ctx, ctx->scratch, pairs, pairs_sorted, np, 32 + bits_for(num_values + 1)));
    k_adj_splits<<<grid_for(num_values + 1, BLK), BLK, 0, s>>>(pairs_sorted, np, num_values, adj_rs);
    ASR_CHECK_LAUNCH(ctx);
    k_adj_payload<<<grid_for(np, BLK), BLK, 0, s>>>(pairs_sorted, np, adj);
    ASR_CHECK_LAUNCH(ctx);
    k_contour_edges<false><<<grid_for(na * 3 + 1, BLK), BLK, 0, s>>>(values, duals, threshold, active, na, adj_rs,
                                                                     adj, tri_cnt, extra_cnt, nullptr, nullptr,
                                                                     nullptr, nullptr, ctx->d_flags);
    ASR_CHECK_LAUNCH(ctx);
    ASR_TRY(scan_counts(ctx, ctx->scratch, tri_cnt, tri_off, na * 3 + 1));
    ASR_TRY(scan_counts(ctx, ctx->scratch, extra_cnt, extra_off, na * 3 + 1));
    ASR_TRY(read_i64(ctx, tri_off + na * 3, &st.num_tri));
    ASR_TRY(read_i64(ctx, extra_off + na * 3, &st.num_extra));
    int host[16];
    ASR_TRY(read_flags(ctx, host));
    if (host[2])
        ASR_FAIL(ctx, ASR_HIP_ELOGIC, "contour: more than %d dual cells around one edge", ASR_USET_CAP);
    if (na + st.num_extra >= (i64(1) << 31) || st.num_tri >= (i64(1) << 31) / 3)
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "contour: mesh does not fit 32-bit indices");
    st.active = active;
    st.vtx = vtx0;
    st.adj_rs = adj_rs;
    st.adj = adj;
    st.tri_off = tri_off;
    st.extra_off = extra_off;
    *num_vertices = na + st.num_extra;
    *num_triangles = st.num_tri;
    return ASR_HIP_OK;
}

int asr_mesh_contour_fill(asr_hip_context* ctx, float* vertices, int32_t* triangles) {
    MeshState& st = mstate(ctx);
    if (st.kind != 1) ASR_FAIL(ctx, ASR_HIP_EINVAL, "contour_fill must follow the matching contour_count call");
    st.kind = 0;
    const i64 na = st.num_active;
    if (na == 0) return ASR_HIP_OK;
    hipStream_t s = ctx->stream;
    ASR_HIP_CHECK(ctx, hipMemcpyAsync(vertices, st.vtx, (size_t)na * 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
    k_contour_edges<true><<<grid_for(na * 3, BLK), BLK, 0, s>>>(st.values, st.duals, st.thr, st.active, na,
                                                                st.adj_rs, st.adj, nullptr, nullptr, st.tri_off,
                                                                st.extra_off, vertices, triangles, ctx->d_flags);
    ASR_CHECK_LAUNCH(ctx);
    int host[16];
    ASR_TRY(read_flags(ctx, host));
    if (host[3]) ASR_FAIL(ctx, ASR_HIP_ELOGIC, "this should not happen: cannot sort duals (cpp/lib/contouring.cpp:366-370)");
    return ASR_HIP_OK;
}

int asr_mesh_components_count(asr_hip_context* ctx, const float* vertices, i64 nv, const int32_t* triangles,
                              i64 nt, i64 keep_n, i64 min_size, i64* nv_out, i64* nt_out) {
    MeshState& st = mstate(ctx);
    st = MeshState();
    st.kind = 2;
    *nv_out = 0;
    *nt_out = 0;
    if (nv <= 0) return ASR_HIP_OK;
    if (nv >= (i64(1) << 31) || nt >= (i64(1) << 31) / 3)
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "components: mesh does not fit 32-bit indices");
    ASR_TRY(ensure_flags(ctx));
    ctx->scratch.reset();
    hipStream_t s = ctx->stream;
    ASR_TRY(fresh_flags(ctx));
    MESH_ALLOC(parent, int, nv);
    MESH_ALLOC(is_root, i64, nv + 1);
    MESH_ALLOC(label, i64, nv + 1);
    MESH_ALLOC(comp, int32_t, nv);
    k_uf_init<<<grid_for(nv, BLK), BLK, 0, s>>>(parent, nv);
    ASR_CHECK_LAUNCH(ctx);
    if (nt > 0) {
        k_uf_link<<<grid_for(nt, BLK), BLK, 0, s>>>(triangles, nt, nv, parent, ctx->d_flags);
        ASR_CHECK_LAUNCH(ctx);
    }
    k_uf_roots<<<grid_for(nv + 1, BLK), BLK, 0, s>>>(parent, nv, is_root);
    ASR_CHECK_LAUNCH(ctx);
    ASR_TRY(scan_counts(ctx, ctx->scratch, is_root, label, nv + 1));
    i64 nc = 0;
    ASR_TRY(read_i64(ctx, label + nv, &nc));
    int host[16];
    ASR_TRY(read_flags(ctx, host));
    if (host[4]) ASR_FAIL(ctx, ASR_HIP_EINVAL, "components: triangle index out of range");
    MESH_ALLOC(sizes, int, nc);
    MESH_ALLOC(keys, u64, nc);
    MESH_ALLOC(keys_sorted, u64, nc);
    MESH_ALLOC(keep, uint8_t, nc);
    ASR_HIP_CHECK(ctx, hipMemsetAsync(sizes, 0, (size_t)nc * sizeof(int), s));
    k_comp_sizes<<<grid_for(nv, BLK), BLK, 0, s>>>(parent, nv, label, comp, sizes);
    ASR_CHECK_LAUNCH(ctx);
    k_comp_keys<<<grid_for(nc, BLK), BLK, 0, s>>>(sizes, nc, keys);
    ASR_CHECK_LAUNCH(ctx);
    {
        size_t tb = 0;
        ASR_HIP_CHECK(ctx, rocprim::radix_sort_keys_desc(nullptr, tb, keys, keys_sorted, (size_t)nc, 0, 64, s));
        void* tmp = ctx->scratch.alloc(tb ? tb : 256);
        if (!tmp) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
        ASR_HIP_CHECK(ctx, rocprim::radix_sort_keys_desc(tmp, tb, keys, keys_sorted, (size_t)nc, 0, 64, s));
    }
    k_comp_keep<<<grid_for(nc, BLK), BLK, 0, s>>>(keys_sorted, nc, keep_n, min_size, keep);
    ASR_CHECK_LAUNCH(ctx);
    MESH_ALLOC(vflag, i64, nv + 1);
    MESH_ALLOC(voff, i64, nv + 1);
    MESH_ALLOC(tflag, i64, nt + 1);
    MESH_ALLOC(toff, i64, nt + 1);
    k_vertex_keep<<<grid_for(nv + 1, BLK), BLK, 0, s>>>(comp, keep, nv, vflag);
    ASR_CHECK_LAUNCH(ctx);
    ASR_TRY(scan_counts(ctx, ctx->scratch, vflag, voff, nv + 1));
    k_tri_keep<<<grid_for(nt + 1, BLK), BLK, 0, s>>>(triangles, nt, voff, tflag);
    ASR_CHECK_LAUNCH(ctx);
    ASR_TRY(scan_counts(ctx, ctx->scratch, tflag, toff, nt + 1));
    ASR_TRY(read_i64(ctx, voff + nv, &st.nv_out));
    ASR_TRY(read_i64(ctx, toff + nt, &st.nt_out));
    st.in_vtx = vertices;
    st.in_tri = triangles;
    st.nv = nv;
    st.nt = nt;
    st.v_off = voff;
    st.t_off = toff;
    *nv_out = st.nv_out;
    *nt_out = st.nt_out;
    return ASR_HIP_OK;
}

int asr_mesh_components_fill(asr_hip_context* ctx, float* vertices_out, int32_t* triangles_out) {
    MeshState& st = mstate(ctx);
    if (st.kind != 2)
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "components_fill must follow the matching components_count call");
    st.kind = 0;
    hipStream_t s = ctx->stream;
    if (st.nv > 0 && st.nv_out > 0) {
        k_compact_vertices<<<grid_for(st.nv, BLK), BLK, 0, s>>>(st.in_vtx, st.nv, st.v_off, vertices_out);
        ASR_CHECK_LAUNCH(ctx);
    }
    if (st.nt > 0 && st.nt_out > 0) {
        k_compact_triangles<<<grid_for(st.nt, BLK), BLK, 0, s>>>(st.in_tri, st.nt, st.v_off, st.t_off,
                                                                 triangles_out);
        ASR_CHECK_LAUNCH(ctx);
    }
    ASR_HIP_CHECK(ctx, hipStreamSynchronize(s));
    return ASR_HIP_OK;
}

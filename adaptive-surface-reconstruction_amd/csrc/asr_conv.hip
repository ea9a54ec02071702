// asr_conv.hip -- floating point half of the hot path on gfx950 (fp32 throughout):
//   a10 continuous conv (wave per output voxel, lane = one of the 4x4x4 filter cells),
//   a12 sparse conv (gather -> f32 MFMA 16x16x4, slot-skipping row tiles; scalar reference
//       kernel for odd shapes and cross-checking),
//   importance sums, a14 decoder MLP.
#include <cstdlib>

#include <type_traits>

#include "asr_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------
// neighbour importance of the aggregation block: compat * clamp((1-d)^3, 0, 1)
// (models/common_torch.py:21-22, net_definitions_torch.py:107)
// ------------------------------------------------------------------------------------------
__global__ void k_agg_importance(const float* compat, const float* dist, i64 n, float* out) {
    i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (i >= n) return;
    float t = 1.f - dist[i];
    float w = t * t * t;
    w = fminf(fmaxf(w, 0.f), 1.f);
    out[i] = compat[i] * w;
}

// ------------------------------------------------------------------------------------------
// a10: continuous conv, kernel 4x4x4 (64 cells == one wavefront).
// One wave per output voxel.  Lane l owns filter cell l = (z*4+y)*4+x and accumulates
// B[cell][0..3] (four input channels per pass) over the voxel's neighbours with its own
// trilinear weight; the 256-deep contraction with W[cell][cin][cout] is a per-lane partial
// dot product followed by a wave reduce-scatter.  HBM traffic: neighbour index / importance /
// point rows (gather) + one output row; the 32 KB filter stays in L1/L2.
// ------------------------------------------------------------------------------------------
__device__ inline float wave_reduce_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// per-pair filter coordinates u in [0,3]^3: ball -> cube (radial) -> 4^3 grid, align_corners
// 
__device__ inline void cconv_pair_coords(float dx, float dy, float dz, float& ux, float& uy, float& uz) {
    float r = sqrtf(dx * dx + dy * dy + dz * dz);
    float m = fmaxf(fabsf(dx), fmaxf(fabsf(dy), fabsf(dz)));
    if (m < 1e-8f) {
        dx = dy = dz = 0.f;
    } else {
        float s = 0.5f * r / m;
        dx *= s;
        dy *= s;
        dz *= s;
    }
    ux = fminf(fmaxf((dx + 0.5f) * 3.f, 0.f), 3.f);
    uy = fminf(fmaxf((dy + 0.5f) * 3.f, 0.f), 3.f);
    uz = fminf(fmaxf((dz + 0.5f) * 3.f, 0.f), 3.f);
}

// wave-wide sum on the DPP network (row shifts + row broadcasts), result in every lane.  __shfl_xor would
// go through ds_bpermute: an LDS round trip and a bounds select per step.
__device__ inline float wave_sum_dpp(float v) {
#define ASR_DPP_ADD(ctrl_, rmask_)                                                                              \
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl_, rmask_, 0xf, false));
    ASR_DPP_ADD(0x111, 0xf)  // row_shr:1
    ASR_DPP_ADD(0x112, 0xf)  // row_shr:2
    ASR_DPP_ADD(0x114, 0xf)  // row_shr:4
    ASR_DPP_ADD(0x118, 0xf)  // row_shr:8   -> lane 15 of each row holds the row sum
    ASR_DPP_ADD(0x142, 0xa)  // row_bcast:15 into rows 1 and 3
    ASR_DPP_ADD(0x143, 0xc)  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave sum
#undef ASR_DPP_ADD
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Accumulates pairs [p0, p0+cnt) (cnt <= 64, wave uniform) of one output voxel into this lane's filter
// cell.  Lane j loads pair j (index, position, importance, 4 features), computes its filter
// coordinates and parks (ux, uy, uz, f0) / (f1, f2, f3, -) in the wave's LDS slab; the wave then walks the
// batch reading each pair with two uniform-address ds_read_b128 (LDS broadcasts, issued ahead of their
// use: they do not take VALU issue slots, where the seven v_readlane per pair of the previous version
// were a third of the kernel's VALU instructions).
// The trilinear weight of cell c along an axis is the hat function clamp(1 - |u - c|, 0, 1), which
// equals the (1-a, a) corner weights of linear interpolation with border clamping.
// SORTED: inp_pos is an array of 32-byte records {x, y, z, -, f0, f1, f2, f3} in Morton order (inp_feat unused) and
// nidx holds positions in that order: the neighbours of one voxel (and of the next voxels, which follow in
// Morton order too) share cache lines, and position + features of a pair come from ONE line, where the AoS
// gathers at original indices fetch two lines per pair.
template <bool SORTED>
__device__ inline void cconv_batch(const float* __restrict__ inp_pos, const float* __restrict__ inp_feat,
                                   const int32_t* __restrict__ nidx, const float* __restrict__ nimp,
                                   i64 p0, int cnt, int lane, int cin, int c0, float ox, float oy,
                                   float oz, float sc2, float cxf, float cyf, float czf, float& B0,
                                   float& B1, float& B2, float& B3, float& norm_lane, float4* s_pair) {
    float ux = 0.f, uy = 0.f, uz = 0.f, f0 = 0.f, f1 = 0.f, f2 = 0.f, f3 = 0.f;
    if (lane < cnt) {
        const i64 p = p0 + lane;
        const int32_t i = nidx[p];
        const float w = nimp ? nimp[p] : 1.f;
        norm_lane += w;
        if (SORTED) {
            const float4 P = reinterpret_cast<const float4*>(inp_pos)[2 * (i64)i];
            const float4 F = reinterpret_cast<const float4*>(inp_pos)[2 * (i64)i + 1];
            cconv_pair_coords((P.x - ox) * sc2, (P.y - oy) * sc2, (P.z - oz) * sc2, ux, uy, uz);
            f0 = w * F.x;
            f1 = w * F.y;
            f2 = w * F.z;
            f3 = w * F.w;
        } else {
            cconv_pair_coords((inp_pos[3 * (i64)i] - ox) * sc2, (inp_pos[3 * (i64)i + 1] - oy) * sc2,
                              (inp_pos[3 * (i64)i + 2] - oz) * sc2, ux, uy, uz);
            const float* f = inp_feat + (i64)i * cin + c0;
            f0 = w * f[0];
            if (c0 + 1 < cin) f1 = w * f[1];
            if (c0 + 2 < cin) f2 = w * f[2];
            if (c0 + 3 < cin) f3 = w * f[3];
        }
    }
    __builtin_amdgcn_wave_barrier();  // the previous batch's reads are done (LDS ops of a wave complete in order)
    s_pair[2 * lane] = make_float4(ux, uy, uz, f0);
    s_pair[2 * lane + 1] = make_float4(f1, f2, f3, 0.f);
    __builtin_amdgcn_wave_barrier();
#pragma unroll 4
    for (int j = 0; j < cnt; ++j) {
        const float4 a = s_pair[2 * j], b = s_pair[2 * j + 1];
        const float wx = fminf(fmaxf(1.f - fabsf(a.x - cxf), 0.f), 1.f);
        const float wy = fminf(fmaxf(1.f - fabsf(a.y - cyf), 0.f), 1.f);
        const float wz = fminf(fmaxf(1.f - fabsf(a.z - czf), 0.f), 1.f);
        const float wt = wx * wy * wz;
        B0 += wt * a.w;
        B1 += wt * b.x;
        B2 += wt * b.y;
        B3 += wt * b.z;
    }
}

// The same batch on the matrix cores (cin == 4): the 64 cells x 4 features of the voxel are one 16 x 16 accumulator
// tile D[i = cy + 4 cz][j = c + 4 cx] (k = 16 i + j of the cell-major basis), four pairs per v_mfma_f32_16x16x4_f32;
// lane (n = lane & 15, g = lane >> 4) supplies A = hy(cy) hz(cz) and B = hx(cx) w f_c of pair g (see k_cconv_mfma).
template <bool SORTED>
__device__ inline void cconv_batch_mma(const float* __restrict__ inp_pos, const float* __restrict__ inp_feat,
                                       const int32_t* __restrict__ nidx, const float* __restrict__ nimp, i64 p0, int cnt,
                                       int lane, float ox, float oy, float oz, float sc2, f32x4& D, float& norm_lane,
                                       float4* s_pair) {
    const int n = lane & 15, g = lane >> 4;
    const float a_cy = (float)(n & 3), a_cz = (float)(n >> 2), b_cx = (float)(n >> 2);
    const int b_c = n & 3;
    float ux = 0.f, uy = 0.f, uz = 0.f;
    float4 F = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < cnt) {
        const i64 p = p0 + lane;
        const int32_t i = nidx[p];
        const float w = nimp ? nimp[p] : 1.f;
        norm_lane += w;
        float4 P;
        if (SORTED) {
            P = reinterpret_cast<const float4*>(inp_pos)[2 * (i64)i];
            F = reinterpret_cast<const float4*>(inp_pos)[2 * (i64)i + 1];
        } else {
            P = make_float4(inp_pos[3 * (i64)i], inp_pos[3 * (i64)i + 1], inp_pos[3 * (i64)i + 2], 0.f);
            F = make_float4(inp_feat[4 * (i64)i], inp_feat[4 * (i64)i + 1], inp_feat[4 * (i64)i + 2], inp_feat[4 * (i64)i + 3]);
        }
        cconv_pair_coords((P.x - ox) * sc2, (P.y - oy) * sc2, (P.z - oz) * sc2, ux, uy, uz);
        F = make_float4(w * F.x, w * F.y, w * F.z, w * F.w);
    }
    __builtin_amdgcn_wave_barrier();
    s_pair[2 * lane] = make_float4(ux, uy, uz, 0.f);
    s_pair[2 * lane + 1] = F;  // zero beyond cnt: those products vanish
    __builtin_amdgcn_wave_barrier();
    const float* sp = reinterpret_cast<const float*>(s_pair);
    for (int j = 0; j < cnt; j += 4) {
        const float4 a = s_pair[2 * (j + g)];
        const float fc = sp[8 * (j + g) + 4 + b_c];
        const float wx = fminf(fmaxf(1.f - fabsf(a.x - b_cx), 0.f), 1.f);
        const float wy = fminf(fmaxf(1.f - fabsf(a.y - a_cy), 0.f), 1.f);
        const float wz = fminf(fmaxf(1.f - fabsf(a.z - a_cz), 0.f), 1.f);
        D = __builtin_amdgcn_mfma_f32_16x16x4f32(wy * wz, wx * fc, D, 0, 0, 0);
    }
}

// One wave per output voxel, persistent blocks (grid-stride over voxels).  The 4-channel slice
// of the filter is staged in LDS once per block as [cell][cout][4] (a guarded global load per
// filter element costs a vmcnt(0) round trip each: 128 per voxel).
// Contraction out[o] = sum_cell sum_c W[cell][c][o] * B[cell][c]: the wave writes its B (one
// float4 per lane = cell) to LDS, then lane (part, o) with part = lane / COUT_MAX sums the cells of
// its part (broadcast B reads, contiguous W reads) and log2(64 / COUT_MAX) shuffles join the parts.
// (A per-lane partial product followed by a 32-value reduce-scatter butterfly was measured at
// 6.4 ms of this kernel's 9.8 ms on the 10 M cloud.)
template <int COUT_MAX, bool SORTED>
__global__ __launch_bounds__(1024, 8) void k_cconv(const float* __restrict__ filters,
                                               const float* __restrict__ out_pos,
                                               const float* __restrict__ extents,
                                               const float* __restrict__ inp_pos,
                                               const float* __restrict__ inp_feat,
                                               const int32_t* __restrict__ nidx,
                                               const float* __restrict__ nimp,
                                               const i64* __restrict__ rs, i64 num_out, int cin,
                                               int cout, int normalize,
                                               const float* __restrict__ bias, int relu,
                                               float* __restrict__ out, i64 heavy_rows) {
    constexpr int PARTS = 64 / COUT_MAX;      // lane = (part, o)
    constexpr int CELLS = 64 / PARTS;         // cells summed by one lane
    __shared__ __attribute__((aligned(16))) float4 s_f[64 * COUT_MAX];  // [cell][o] -> 4 channels
    __shared__ __attribute__((aligned(16))) float4 s_b[16][64];          // per wave: B[cell]
    __shared__ __attribute__((aligned(16))) float4 s_pair[16][128];      // per wave: the current batch of pairs
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const float cxf = (float)(lane & 3), cyf = (float)((lane >> 2) & 3), czf = (float)(lane >> 4);
    const i64 wave0 = (blockIdx.x * (i64)blockDim.x + threadIdx.x) >> 6;
    const i64 nwaves = ((i64)gridDim.x * blockDim.x) >> 6;
    const int my_o = lane % COUT_MAX, part = lane / COUT_MAX;
    const bool writer = part == 0 && my_o < cout;
    const float bias_o = (bias && my_o < cout) ? bias[my_o] : 0.f;

    for (int c0 = 0; c0 < cin; c0 += 4) {
        __syncthreads();
        for (int e = threadIdx.x; e < 64 * COUT_MAX; e += blockDim.x) {
            const int cell = e / COUT_MAX, o = e % COUT_MAX;
            float w[4];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                w[c] = (c0 + c < cin && o < cout) ? filters[((i64)cell * cin + c0 + c) * cout + o] : 0.f;
            s_f[e] = make_float4(w[0], w[1], w[2], w[3]);
        }
        __syncthreads();
        for (i64 q = wave0; q < num_out; q += nwaves) {
            // q is wave uniform: row bounds through scalar registers, so that the pair loop is a scalar loop
            const i64 b = (i64)(((u64)(u32)__builtin_amdgcn_readfirstlane((int)(rs[q] >> 32)) << 32) |
                                (u32)__builtin_amdgcn_readfirstlane((int)rs[q]));
            const i64 e = (i64)(((u64)(u32)__builtin_amdgcn_readfirstlane((int)(rs[q + 1] >> 32)) << 32) |
                                (u32)__builtin_amdgcn_readfirstlane((int)rs[q + 1]));
            if (e - b > heavy_rows) continue;  // long rows go to k_cconv_heavy (16 waves per row)
            float* orow = out + q * cout;
            if (e == b) {
                // no neighbour: conv = 0 -> activation(bias)
                if (c0 == 0 && writer) orow[my_o] = relu ? fmaxf(bias_o, 0.f) : bias_o;
                continue;
            }
            const float ox = out_pos[3 * q], oy = out_pos[3 * q + 1], oz = out_pos[3 * q + 2];
            const float sc2 = 2.f * (1.f / extents[q]);
            float B0 = 0.f, B1 = 0.f, B2 = 0.f, B3 = 0.f, norm_lane = 0.f;
            for (i64 p0 = b; p0 < e; p0 += 64)
                cconv_batch<SORTED>(inp_pos, inp_feat, nidx, nimp, p0, (int)((e - p0) < 64 ? (e - p0) : 64), lane,
                                    cin, c0, ox, oy, oz, sc2, cxf, cyf, czf, B0, B1, B2, B3, norm_lane,
                                    s_pair[wib]);
            s_b[wib][lane] = make_float4(B0, B1, B2, B3);
            const float norm = wave_sum_dpp(norm_lane);
            __builtin_amdgcn_wave_barrier();  // LDS ops of one wave complete in order
            float r = 0.f;
#pragma unroll 8
            for (int i = 0; i < CELLS; ++i) {
                const int cell = part * CELLS + i;
                const float4 bb = s_b[wib][cell];
                const float4 ww = s_f[cell * COUT_MAX + my_o];
                r += ww.x * bb.x + ww.y * bb.y + ww.z * bb.z + ww.w * bb.w;
            }
#pragma unroll
            for (int m = COUT_MAX; m < 64; m <<= 1) r += __shfl_xor(r, m, 64);
            __builtin_amdgcn_wave_barrier();
            if (writer) {
                if (c0 > 0) r += orow[my_o];  // wider inputs: accumulate the raw sums over chunks
                if (c0 + 4 >= cin) {
                    if (normalize && norm != 0.f) r = r / norm;
                    r += bias_o;
                    if (relu) r = fmaxf(r, 0.f);
                }
                orow[my_o] = r;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// a10 with pair loop and contraction on the matrix cores (whole-path layout: Morton-ordered 32-byte records, cin = 4,
// cout <= 32).
//   * A wave owns SUPER-GROUPS of 64 consecutive voxels (lane u holds the row bounds / centre / extent of voxel u) and
//     streams their pairs in CHUNKS of at most 64 pairs: a chunk is a run of whole voxels (the mean row has 18 pairs, so
//     3-4 voxels share one batch of 64 lanes: one index load, one record load, one coordinate transform and one LDS
//     store per chunk instead of per voxel); a row of more than 64 pairs gets chunks of its own, cut at multiples of 64
//     from the row start.  The index loads of chunk c+2 and the record loads of chunk c+1 are in flight while chunk c
//     is accumulated (the row_splits -> index -> record chain is the latency that bounds a wave; the pipeline gives the
//     depth that the two to four waves per SIMD alone do not).
//   * Per voxel the 64 cells x 4 features are ONE 16 x 16 accumulator tile D[i = cy + 4 cz][j = c + 4 cx], four pairs
//     per v_mfma_f32_16x16x4_f32, quads counted from the ROW start (a lane whose pair lies beyond the row end feeds a
//     zero B operand): a row's result does not depend on which rows share its chunk.
//   * Every CCG voxels the CCG x 256 matrix parked in LDS is contracted with the 256 x cout filter matrix on the f32 matrix
//     cores (exact f32 products, f32 accumulate); the two forms of the contraction are described at the template below.
//     CCG == 16: MFMA step (j, t) of k-lane g contracts k = 16 j + 4 g + t on both operands, so the A fragment of four
//     steps is one ds_read_b128; the LDS rows are 264 floats apart (conflict-free for that read, and for CCG == 4's).
// ------------------------------------------------------------------------------------------
constexpr int CCG_LD = 264;    // floats per LDS row of the group matrix
#ifndef ASR_CCONV_GROUP
#define ASR_CCONV_GROUP 4
#endif
constexpr int CCS = 64;        // voxels per super-group (one per lane)
constexpr int CC_PAIR_LD = 160;  // float4 per wave: 64 pairs x 2 + 16 zero pairs (quads are read four at a time, up to 15 beyond a batch)
// SORTED: neighbours are positions into the 32-byte records `rec`; else `rec` is unused and positions / features
// come from the AoS arrays inp_pos [N,3] / inp_feat [N,4] at original indices (the generic operator boundary):
// same arithmetic, so both layouts give identical bits.
// CCG = voxels per contraction group.
//   16 (rounds 2-5): the group matrix is the A operand of v_mfma_f32_16x16x4_f32, the filter fragments live in 128 registers:
//      8 waves per CU (135 KB of group matrices, 200 registers).
//   4 (round 6): the 16 blocks of v_mfma_f32_4x4x1_16B_f32 are (8 groups of four output channels) x (two halves of k): block
//      (og, kh) accumulates out[voxel i][4 og + j] over k = 128 kh + s -- the same multiply-adds per cycle as the 16 x 16 x 4
//      form with 4 voxels per group instead of 16.  The filter leaves the registers for 32 KB of LDS shared by the block (one
//      ds_read_b128 per four steps and lane, as for the group matrix); the two halves of k are added with one lane exchange
//      per output.  16 waves per CU (4 per SIMD) instead of 8: the kernel is bound by the latency of its dependent chains
//      (one wave per SIMD: 2.48 ms, two: 1.56 ms).
constexpr int cconv_waves(int ccg) { return ccg == 16 ? 8 : 16; }
template <bool SORTED, int CCG>
__global__ __launch_bounds__(cconv_waves(CCG) * 64) void k_cconv_mfma(
        const float* __restrict__ filters, const float* __restrict__ out_pos, const float* __restrict__ extents,
        const float4* __restrict__ rec, const float* __restrict__ inp_pos, const float* __restrict__ inp_feat,
        const int32_t* __restrict__ nidx, const float* __restrict__ nimp,
        const i64* __restrict__ rs, i64 num_out, int cout, int normalize, const float* __restrict__ bias, int relu,
        float* __restrict__ out, i64 heavy_rows, float* __restrict__ basis_out, float* __restrict__ norm_out,
        unsigned* __restrict__ out_absmax, int* __restrict__ sg_counter) {
    // basis_out != null ("next" row f4, filter gradient): only the per-voxel matrices B[v][256] and the importance sums
    // are written; the contraction happens in the caller (dW = B^T g)
    // out_absmax != null: the running maximum of |out| for the f16x2 sparse conv that reads it (one atomic per wave)
    unsigned amax = 0;
    constexpr int NW = cconv_waves(CCG);
    __shared__ __attribute__((aligned(16))) float s_bt[NW][CCG][CCG_LD];
    __shared__ __attribute__((aligned(16))) float4 s_pair[NW][CC_PAIR_LD];
    __shared__ float s_norm[NW][CCG];
    // CCG == 4: s_w[u][lane = 4 (og + 8 kh) + j] = W[k = 128 kh + 4 u + t][o = 4 og + j], t = 0..3
    __shared__ __attribute__((aligned(16))) float4 s_w[CCG == 16 ? 1 : 32 * 64];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n = lane & 15, g = lane >> 4;
    // pair loop on the matrix cores: D[i = cy + 4 cz][j = c + 4 cx] = sum over pairs of (hy hz)[i] * (hx w f_c)[j], i.e.
    // k = 16 i + j of the group matrix; lane (i = n, pair g) supplies A = hy(cy) hz(cz), lane (j = n, pair g)
    // B = hx(cx) f_c -- three hat functions and two products per lane for four pairs.
    const float a_cy = (float)(n & 3), a_cz = (float)(n >> 2);  // A operand: row i = n
    const float b_cx = (float)(n >> 2);                         // B operand: column j = n -> cx = j / 4, channel j % 4
    const int b_c = n & 3;
    // filter fragments: wreg[T][j][t] = W[k = 16 j + 4 g + t][o = 16 T + n], W = filters viewed as [256][cout]
    float wreg[CCG == 16 ? 2 : 1][CCG == 16 ? 16 : 1][4];
    float bias_o[2];
    if constexpr (CCG == 16) {
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int j = 0; j < 16; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int o = 16 * T + n;
                    wreg[T][j][t] = o < cout ? filters[(i64)(16 * j + 4 * g + t) * cout + o] : 0.f;
                }
#pragma unroll
        for (int T = 0; T < 2; ++T) bias_o[T] = (bias && 16 * T + n < cout) ? bias[16 * T + n] : 0.f;
    } else {
        for (int e = threadIdx.x; e < 32 * 64; e += NW * 64) {
            const int u = e >> 6, l = e & 63, o = l & 31, k0 = 128 * (l >> 5) + 4 * u;
            float w[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) w[t] = (o < cout && !basis_out) ? filters[(i64)(k0 + t) * cout + o] : 0.f;
            s_w[e] = make_float4(w[0], w[1], w[2], w[3]);
        }
        bias_o[0] = (bias && (lane & 31) < cout) ? bias[lane & 31] : 0.f;  // lane = output channel (both halves of k)
        bias_o[1] = 0.f;
        __syncthreads();
    }
    if (lane < CC_PAIR_LD - 128) s_pair[wib][128 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);  // never written again

    float4* const sp4 = &s_pair[wib][0];
    const float* const sp = reinterpret_cast<const float*>(sp4);
    // Super-groups are handed out by an atomic counter (zeroed by the launcher): with the static round robin the slowest of the
    // 2 048 waves carried 1.13-1.27 x the mean number of pairs (scripts/cconv_balance.py) and set the kernel's time.  The
    // next ticket is drawn while the current super-group is processed.
    const i64 sgroups = (num_out + CCS - 1) / CCS;
    auto draw = [&]() __attribute__((always_inline)) -> int {
        int t = 0;
        if (lane == 0) t = atomicAdd(sg_counter, 1);
        return t;
    };
    int ticket = draw();
    for (;;) {
        const i64 sg = __builtin_amdgcn_readfirstlane(ticket);
        if (sg >= sgroups) break;
        ticket = draw();
        const i64 q0 = sg * CCS;
        const int nvox = (int)(num_out - q0 < CCS ? num_out - q0 : CCS);
        // lane u: voxel u of the super-group.  mcnt = pairs this kernel accumulates (0 for a row of k_cconv_heavy)
        int mb = 0, mcnt = 0, mall = 0;
        float mox = 0.f, moy = 0.f, moz = 0.f, msc = 0.f;
        const i64 pbase = rs[q0];
        if (lane < nvox) {
            const i64 b = rs[q0 + lane], e = rs[q0 + lane + 1];
            mb = (int)(b - pbase);
            mall = (int)(e - b < 0x7fffffff ? e - b : 0x7fffffff);
            mcnt = (e - b) > heavy_rows ? 0 : mall;
            mox = out_pos[3 * (q0 + lane)];
            moy = out_pos[3 * (q0 + lane) + 1];
            moz = out_pos[3 * (q0 + lane) + 2];
            msc = 2.f * (1.f / extents[q0 + lane]);
        }
        // ---- chunk sequence (scalar state): next pair to hand out is pair `cso` of voxel `cu` ----
        int cu = 0, cso = 0;
        struct Chunk {
            int u0, so, u1, total;  // voxels [u0, u1), `so` pairs of u0 consumed by earlier chunks, pairs in the chunk (-1: none)
        };
        auto next_chunk = [&]() __attribute__((always_inline)) -> Chunk {
            Chunk c = {0, 0, 0, -1};
            if (cu >= nvox) return c;
            c.u0 = cu;
            c.so = cso;
            const int left = __builtin_amdgcn_readlane(mcnt, cu) - cso;
            if (left > 64) {  // the row goes on in the next chunk
                c.u1 = cu + 1;
                c.total = 64;
                cso += 64;
                return c;
            }
            // whole rows while their quads fit: every row starts on a quad boundary of the LDS batch (zero filled
            // up to it), so that the quad loop needs no end-of-row test
            int total = left, padded = (left + 3) & ~3;
            ++cu;
            cso = 0;
            while (cu < nvox) {
                const int c2 = __builtin_amdgcn_readlane(mcnt, cu);
                if (padded + c2 > 64) break;
                total += c2;
                padded = (padded + c2 + 3) & ~3;
                ++cu;
            }
            c.u1 = cu;
            c.total = total;
            return c;
        };
        // stage A: which voxel a lane's pair belongs to, its index and importance
        // (lpos: where the pair goes in the LDS batch = the lane moved up by the zero fill in front of its row)
        auto stage_a = [&](const Chunk& c, int& vox, int& lpos, int& idx, float& w) __attribute__((always_inline)) {
            vox = c.u0;
            lpos = lane;
            idx = 0;
            w = 0.f;
            if (c.total < 0) return;
            int vb = __builtin_amdgcn_readlane(mb, c.u0) + c.so, voff = 0, fill = 0;
            int off = __builtin_amdgcn_readlane(mcnt, c.u0) - c.so;
            off = off > 64 ? 64 : off;
            int poff = (off + 3) & ~3;
            for (int u = c.u0 + 1; u < c.u1; ++u) {
                const int cn = __builtin_amdgcn_readlane(mcnt, u), bu = __builtin_amdgcn_readlane(mb, u);
                if (lane >= off) {
                    vox = u;
                    vb = bu;
                    voff = off;
                    fill = poff - off;
                }
                off += cn;
                poff = (poff + cn + 3) & ~3;
            }
            lpos = lane + fill;
            if (lane < c.total) {
                const i64 p = pbase + vb + (lane - voff);
                idx = nidx[p];
                w = nimp ? nimp[p] : 1.f;
            }
        };
        // stage B: position + features of the pair
        auto stage_b = [&](int idx, float4& P, float4& F) __attribute__((always_inline)) {
            if (SORTED) {
                P = rec[2 * (i64)idx];
                F = rec[2 * (i64)idx + 1];
            } else {
                P = make_float4(inp_pos[3 * (i64)idx], inp_pos[3 * (i64)idx + 1], inp_pos[3 * (i64)idx + 2], 0.f);
                F = make_float4(inp_feat[4 * (i64)idx], inp_feat[4 * (i64)idx + 1], inp_feat[4 * (i64)idx + 2],
                                inp_feat[4 * (i64)idx + 3]);
            }
        };
        f32x4 D = {0.f, 0.f, 0.f, 0.f};  // the voxel being accumulated (carried over chunks for rows of more than 64 pairs)
        float nl = 0.f;                  // its importance, summed per lane over the row's batches
        // contraction of the CCG voxels q0 + CCG k .. parked in s_bt
        auto contract = [&](int k) __attribute__((always_inline)) {
            __builtin_amdgcn_wave_barrier();
            if constexpr (CCG == 4) {
                // lane 4 b + i supplies A = group matrix[voxel i][128 kh + s], lane 4 b + j supplies B = W[128 kh + s][4 og + j]
                // (b = og + 8 kh); four independent chains (s mod 4) so that consecutive instructions do not wait for each other
                f32x4 acc[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const float* arow = &s_bt[wib][lane & 3][128 * (lane >> 5)];
#pragma unroll  // (all 64 reads up front: -4 % against groups of eight)
                for (int u = 0; u < 32; ++u) {
                    const float4 a4 = *reinterpret_cast<const float4*>(arow + 4 * u);
                    const float4 b4 = s_w[u * 64 + lane];
                    acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4.x, b4.x, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4.y, b4.y, acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4.z, b4.z, acc[2], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a4.w, b4.w, acc[3], 0, 0, 0);
                }
                // acc[.][r] = partial out[voxel r][o = lane & 31] of this lane's half of k
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = (acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r]);
                    v += __shfl_xor(v, 32, 64);
                    const i64 q = q0 + 4 * k + r;
                    const float norm = s_norm[wib][r];
                    if (q >= num_out || norm < 0.f) continue;
                    if (normalize && norm != 0.f) v = v / norm;
                    v += bias_o[0];
                    if (relu) v = fmaxf(v, 0.f);
                    if (lane < cout && lane < 32) {
                        out[q * cout + lane] = v;
                        amax = max(amax, __float_as_uint(v) & 0x7fffffffu);
                    }
                }
            } else {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float4 a4 = *reinterpret_cast<const float4*>(&s_bt[wib][n][16 * j + 4 * g]);
                const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], wreg[0][j][t], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], wreg[1][j][t], acc1, 0, 0, 0);
                }
            }
            // acc[r] = out[voxel 4 g + r][o = 16 T + n]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int vx = 4 * g + r;
                const i64 q = q0 + 16 * k + vx;
                const float norm = s_norm[wib][vx];
                if (q >= num_out || norm < 0.f) continue;
#pragma unroll
                for (int T = 0; T < 2; ++T) {
                    const int o = 16 * T + n;
                    float v = T == 0 ? acc0[r] : acc1[r];
                    if (normalize && norm != 0.f) v = v / norm;
                    v += bias_o[T];
                    if (relu) v = fmaxf(v, 0.f);
                    if (o < cout) {
                        out[q * cout + o] = v;
                        amax = max(amax, __float_as_uint(v) & 0x7fffffffu);
                    }
                }
            }
            }
            __builtin_amdgcn_wave_barrier();  // the group matrix is re-used by the next group
        };
        // stage C: coordinates of the chunk's pairs into LDS, then voxel by voxel through the matrix cores
        auto stage_c = [&](const Chunk& c, int vox, int lpos, float w, const float4& P, const float4& F) __attribute__((always_inline)) {
            const float ox = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * vox, __float_as_int(mox)));
            const float oy = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * vox, __float_as_int(moy)));
            const float oz = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * vox, __float_as_int(moz)));
            const float sc2 = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * vox, __float_as_int(msc)));
            float ux, uy, uz;
            cconv_pair_coords((P.x - ox) * sc2, (P.y - oy) * sc2, (P.z - oz) * sc2, ux, uy, uz);
            __builtin_amdgcn_wave_barrier();
            // zero fill first (LDS operations of a wave complete in order), then the pairs at their rows' positions
            sp4[lane] = make_float4(0.f, 0.f, 0.f, 0.f);
            sp4[64 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (lane < c.total) {
                sp4[2 * lpos] = make_float4(ux, uy, uz, w);
                sp4[2 * lpos + 1] = make_float4(w * F.x, w * F.y, w * F.z, w * F.w);
            }
            __builtin_amdgcn_wave_barrier();
            int off = 0;
            for (int u = c.u0; u < c.u1; ++u) {
                const int full = __builtin_amdgcn_readlane(mcnt, u);
                const int so = u == c.u0 ? c.so : 0;
                int cn = full - so;
                cn = cn > 64 ? 64 : cn;
                if (so == 0) {
                    D = (f32x4){0.f, 0.f, 0.f, 0.f};
                    nl = 0.f;
                }
                if (cn > 0) {
                    // sixteen pairs per round: the eight LDS reads of four quads are issued together (one exposed LDS
                    // latency per round instead of one per quad; reads beyond the batch hit the zero pad)
                    for (int j = 0; j < cn; j += 16) {
                        float4 a[4];
                        float fc[4];
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            a[t] = sp4[2 * (off + j + 4 * t + g)];
                            fc[t] = sp[8 * (off + j + 4 * t + g) + 4 + b_c];
                        }
                        // (pins the eight reads here, before the wave-uniform branches below: the compiler would sink every
                        // read into the branch that consumes it, which puts one LDS round trip in front of every MFMA)
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            asm volatile("" : "+v"(a[t].x), "+v"(a[t].y), "+v"(a[t].z), "+v"(fc[t]));
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            if (t > 0 && j + 4 * t >= cn) break;  // wave uniform (the last quad's tail is zero fill)
                            const float wx = fminf(fmaxf(1.f - fabsf(a[t].x - b_cx), 0.f), 1.f);
                            const float wy = fminf(fmaxf(1.f - fabsf(a[t].y - a_cy), 0.f), 1.f);
                            const float wz = fminf(fmaxf(1.f - fabsf(a[t].z - a_cz), 0.f), 1.f);
                            D = __builtin_amdgcn_mfma_f32_16x16x4f32(wy * wz, wx * fc[t], D, 0, 0, 0);
                        }
                    }
                    if (lane < cn) nl += sp[8 * (off + lane) + 3];
                }
                off += (cn + 3) & ~3;
                if (so + cn < full) continue;  // the row goes on in the next chunk (it is the chunk's only voxel)
                // D[r] = cell sums k = 16 (4 g + r) + n
                const int vx = u & (CCG - 1);
#pragma unroll
                for (int r = 0; r < 4; ++r) s_bt[wib][vx][16 * (4 * g + r) + n] = D[r];
                const float norm = wave_sum_dpp(nl);
                const bool heavy = __builtin_amdgcn_readlane(mall, u) > full;  // written by k_cconv_heavy
                if (lane == 0) s_norm[wib][vx] = heavy ? -1.f : norm;
                if (basis_out) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) basis_out[(q0 + u) * 256 + 16 * (4 * g + r) + n] = D[r];
                    if (lane == 0) norm_out[q0 + u] = norm;
                } else if (vx == CCG - 1 || u == nvox - 1) {
                    contract(u / CCG);
                }
            }
        };

        // ---- the pipeline: chunk c in stage C, c+1 in stage B (records in flight), c+2 in stage A (indices in flight) ----
        Chunk c0 = next_chunk(), c1 = next_chunk();
        int vox0, idx0, lpos0, vox1, idx1, lpos1;
        float w0, w1;
        float4 P0, F0;
        stage_a(c0, vox0, lpos0, idx0, w0);
        stage_b(idx0, P0, F0);
        stage_a(c1, vox1, lpos1, idx1, w1);
#pragma unroll 1
        while (c0.total >= 0) {
            const float4 Pc = P0, Fc = F0;
            const float wc = w0;
            const int voxc = vox0, lposc = lpos0;
            const Chunk cc = c0;
            // records of the next chunk, indices of the one after it
            c0 = c1;
            vox0 = vox1;
            lpos0 = lpos1;
            w0 = w1;
            if (c0.total >= 0) stage_b(idx1, P0, F0);
            c1 = next_chunk();
            stage_a(c1, vox1, lpos1, idx1, w1);
            stage_c(cc, voxc, lposc, wc, Pc, Fc);
        }
    }
    if (out_absmax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = max(amax, (unsigned)__shfl_xor((int)amax, o, 64));
        if (lane == 0 && amax > __hip_atomic_load(out_absmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out_absmax, amax);
    }
}

// ------------------------------------------------------------------------------------------
// a10, long rows: a coarse voxel next to the surface can have tens of thousands of neighbours
// while the mean is ~15.  Rows above CCONV_HEAVY pairs are collected and handled by one
// 1024-thread block each: the 16 waves take interleaved 64-pair batches, partial B sums are
// combined through LDS in wave order (deterministic), wave 0 does the contraction.
// ------------------------------------------------------------------------------------------
constexpr i64 CCONV_HEAVY = 256;
__global__ void k_cconv_absmax(const float* __restrict__ x, i64 n, unsigned* __restrict__ out) {
    unsigned m = 0;
    for (i64 e = blockIdx.x * (i64)blockDim.x + threadIdx.x; e < n; e += (i64)gridDim.x * blockDim.x)
        m = max(m, __float_as_uint(x[e]) & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
__global__ void k_cconv_heavy_list(const i64* rs, i64 num_out, i64 thr, int32_t* list, int* count) {
    i64 q = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (q >= num_out) return;
    if (rs[q + 1] - rs[q] > thr) list[atomicAdd(count, 1)] = (int32_t)q;
}

template <int COUT_MAX, bool SORTED>
__global__ __launch_bounds__(1024) void k_cconv_heavy(
        const float* __restrict__ filters, const float* __restrict__ out_pos,
        const float* __restrict__ extents, const float* __restrict__ inp_pos,
        const float* __restrict__ inp_feat, const int32_t* __restrict__ nidx,
        const float* __restrict__ nimp, const i64* __restrict__ rs, const int32_t* __restrict__ list,
        const int* __restrict__ list_count, int cin, int cout, int normalize, const float* __restrict__ bias,
        int relu, float* __restrict__ out, unsigned* __restrict__ out_absmax) {
    unsigned amax = 0;  // (wave 0 writes the rows)
    __shared__ float s_part[16][64][4];
    __shared__ float s_norm[16];
    __shared__ __attribute__((aligned(16))) float4 s_pair[16][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float cxf = (float)(lane & 3), cyf = (float)((lane >> 2) & 3), czf = (float)(lane >> 4);
    // the number of long rows stays on the device (no host round trip in the middle of the network):
    // a fixed grid strides over the list
    const int n_list = *list_count;
    for (int li = blockIdx.x; li < n_list; li += gridDim.x) {
        __syncthreads();  // the previous row's LDS partials are consumed
        const i64 q = list[li];
        const i64 b = rs[q], e = rs[q + 1];
        const float ox = out_pos[3 * q], oy = out_pos[3 * q + 1], oz = out_pos[3 * q + 2];
        const float sc2 = 2.f * (1.f / extents[q]);
        float acc[COUT_MAX];
    #pragma unroll
        for (int o = 0; o < COUT_MAX; ++o) acc[o] = 0.f;
        float norm_total = 0.f;
        for (int c0 = 0; c0 < cin; c0 += 4) {
            float B0 = 0.f, B1 = 0.f, B2 = 0.f, B3 = 0.f, norm_lane = 0.f;
            f32x4 D = {0.f, 0.f, 0.f, 0.f};
            const bool mma = cin == 4;  // the whole path: pair loop on the matrix cores
            for (i64 p0 = b + 64 * (i64)wave; p0 < e; p0 += 64 * 16) {
                const int cnt = (int)((e - p0) < 64 ? (e - p0) : 64);
                if (mma)
                    cconv_batch_mma<SORTED>(inp_pos, inp_feat, nidx, nimp, p0, cnt, lane, ox, oy, oz, sc2, D, norm_lane,
                                            s_pair[wave]);
                else
                    cconv_batch<SORTED>(inp_pos, inp_feat, nidx, nimp, p0, cnt, lane, cin, c0, ox, oy, oz, sc2, cxf, cyf,
                                        czf, B0, B1, B2, B3, norm_lane, s_pair[wave]);
            }
            const float norm = wave_sum_dpp(norm_lane);
            __syncthreads();  // previous chunk's partials consumed
            if (mma) {  // D[r] = basis element k = 16 (4 g + r) + n = 4 cell + channel
                float* flat = &s_part[wave][0][0];
#pragma unroll
                for (int r = 0; r < 4; ++r) flat[16 * (4 * (lane >> 4) + r) + (lane & 15)] = D[r];
            } else {
                s_part[wave][lane][0] = B0;
                s_part[wave][lane][1] = B1;
                s_part[wave][lane][2] = B2;
                s_part[wave][lane][3] = B3;
            }
            if (lane == 0) s_norm[wave] = norm;
            __syncthreads();
            if (wave == 0) {
                B0 = B1 = B2 = B3 = 0.f;
                float nt = 0.f;
                for (int w2 = 0; w2 < 16; ++w2) {
                    B0 += s_part[w2][lane][0];
                    B1 += s_part[w2][lane][1];
                    B2 += s_part[w2][lane][2];
                    B3 += s_part[w2][lane][3];
                    nt += s_norm[w2];
                }
                if (c0 == 0) norm_total = nt;
                const float* wrow = filters + ((i64)lane * cin + c0) * cout;
    #pragma unroll
                for (int o = 0; o < COUT_MAX; ++o) {
                    if (o < cout) {
                        float s = wrow[o] * B0;
                        if (c0 + 1 < cin) s += wrow[cout + o] * B1;
                        if (c0 + 2 < cin) s += wrow[2 * cout + o] * B2;
                        if (c0 + 3 < cin) s += wrow[3 * cout + o] * B3;
                        acc[o] += s;
                    }
                }
            }
        }
        if (wave != 0) continue;
        float mine = 0.f;
    #pragma unroll
        for (int o = 0; o < COUT_MAX; ++o) {
            if (o < cout) {
                float s = wave_reduce_sum(acc[o]);
                if (lane == o) mine = s;
            }
        }
        if (lane < cout) {
            float r = mine;
            if (normalize && norm_total != 0.f) r = r / norm_total;
            if (bias) r += bias[lane];
            if (relu) r = fmaxf(r, 0.f);
            out[q * cout + lane] = r;
            amax = max(amax, __float_as_uint(r) & 0x7fffffffu);
        }
    }
    if (out_absmax && wave == 0) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = max(amax, (unsigned)__shfl_xor((int)amax, o, 64));
        if (lane == 0 && amax) atomicMax(out_absmax, amax);
    }
}

// ------------------------------------------------------------------------------------------
// Long rows of the whole-path layout (cin == 4), cut into segments.  At 10 M points 135 rows are long and 60 of them hold
// 35 000 pairs on average (a coarse voxel beside the surface): one block per ROW walked such a row for 0.18 ms on an otherwise
// idle chip.  Here a work item is (row, segment of CCH_SEG pairs); its block leaves the 64 x 4 cell sums + the importance sum of
// the segment in a partial slot and k_cconv_heavy4_finish adds a row's segments IN ORDER and contracts; a row of one segment
// is finished by its block.  k_cconv_heavy4_items lays the items out on the device (the number of long rows never reaches
// the host): item_first[li] = first item of long row li, part_first[li] = its first partial slot, or -1 for a row of one
// item -- also EVERY row when the segments of all long rows together would not fit the CCH_CAP slots (one decision per list).
// ------------------------------------------------------------------------------------------
constexpr int CCH_SEG = 4096;
constexpr int CCH_CAP = 16384;  // partial slots of 260 floats
constexpr int CCH_LD = 260;

__global__ __launch_bounds__(1024) void k_cconv_heavy4_items(const i64* __restrict__ rs, const int32_t* __restrict__ list,
                                                             int* __restrict__ cnt, int* __restrict__ item_first,
                                                             int* __restrict__ part_first) {
    __shared__ int s_a[1024], s_b[1024];
    __shared__ int s_carry[2];
    const int n = cnt[0], t = threadIdx.x;
    // The list comes out of atomics in no particular order, so whether a row is cut must not depend on its position in it
    // (a cut row is summed segment by segment: other bits than the same row kept whole): ALL rows of more than one segment are
    // cut when their slots fit together, none otherwise -- one decision per LAUNCH, the same on every run.  (A rank of a
    // sharded cloud lists its own rows only: its decision equals the one-GPU run's as long as the long rows of the WHOLE
    // cloud fit the CCH_CAP slots, i.e. below CCH_CAP x CCH_SEG = 6.7e7 pairs in rows of more than one segment -- 2.1e6 at
    // 10 M points, 1.7e7 at 80 M.  Beyond that the one-GPU run keeps the rows whole while a rank may still cut them, and
    // the two differ in summation order, within the tolerance, not in bits.)
    __shared__ long long s_total;
    if (t == 0) {
        s_carry[0] = s_carry[1] = 0;
        s_total = 0;
    }
    __syncthreads();
    {
        long long mine = 0;
        for (int li = t; li < n; li += 1024) {
            const i64 q = list[li];
            const i64 seg = (rs[q + 1] - rs[q] + CCH_SEG - 1) / CCH_SEG;
            if (seg > 1) mine += seg;
        }
        if (mine) atomicAdd((unsigned long long*)&s_total, (unsigned long long)mine);
    }
    __syncthreads();
    const bool cut_all = s_total <= CCH_CAP;
    for (int base = 0; base < n; base += 1024) {
        const int li = base + t;
        int seg = 0;
        if (li < n) {
            const i64 q = list[li];
            seg = (int)((rs[q + 1] - rs[q] + CCH_SEG - 1) / CCH_SEG);
        }
        const int multi = seg > 1 ? seg : 0;
        // inclusive scans of the partial slots, then (with the rows that do not fit kept whole) of the items
        s_a[t] = multi;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int x = t >= o ? s_a[t - o] : 0;
            __syncthreads();
            s_a[t] += x;
            __syncthreads();
        }
        const int pf = s_carry[0] + s_a[t] - multi;
        const bool cut = multi && cut_all;
        const int items = li < n ? (cut ? seg : 1) : 0;
        s_b[t] = items;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int x = t >= o ? s_b[t - o] : 0;
            __syncthreads();
            s_b[t] += x;
            __syncthreads();
        }
        if (li < n) {
            item_first[li] = s_carry[1] + s_b[t] - items;
            part_first[li] = cut ? pf : -1;
        }
        __syncthreads();
        if (t == 1023) {
            s_carry[0] += s_a[1023];
            s_carry[1] += s_b[1023];
        }
        __syncthreads();
    }
    if (t == 0) {
        item_first[n] = s_carry[1];
        cnt[1] = s_carry[1];
    }
}

// contraction of one row's cell sums (lane = cell: B0..B3 its four channels) with the filter, epilogue, store; every lane of the
// wave takes part.  Returns the f32 bits of the largest |value| written (for out_absmax).
template <int COUT_MAX>
__device__ inline unsigned cconv_heavy4_finish_row(const float* __restrict__ filters, float B0, float B1, float B2, float B3,
                                                   float norm_total, int cout, int normalize, const float* __restrict__ bias,
                                                   int relu, float* __restrict__ orow, int lane) {
    const float* wrow = filters + (i64)lane * 4 * cout;
    float mine = 0.f;
#pragma unroll
    for (int o = 0; o < COUT_MAX; ++o) {
        if (o < cout) {
            float a = wrow[o] * B0;
            a += wrow[cout + o] * B1;
            a += wrow[2 * cout + o] * B2;
            a += wrow[3 * cout + o] * B3;
            const float t = wave_reduce_sum(a);
            if (lane == o) mine = t;
        }
    }
    unsigned amax = 0;
    if (lane < cout) {
        float r = mine;
        if (normalize && norm_total != 0.f) r = r / norm_total;
        if (bias) r += bias[lane];
        if (relu) r = fmaxf(r, 0.f);
        orow[lane] = r;
        amax = __float_as_uint(r) & 0x7fffffffu;
    }
    return amax;
}

template <int COUT_MAX, bool SORTED>
__global__ __launch_bounds__(1024) void k_cconv_heavy4(
        const float* __restrict__ filters, const float* __restrict__ out_pos, const float* __restrict__ extents,
        const float* __restrict__ inp_pos, const float* __restrict__ inp_feat, const int32_t* __restrict__ nidx,
        const float* __restrict__ nimp, const i64* __restrict__ rs, const int32_t* __restrict__ list,
        const int* __restrict__ cnt, const int* __restrict__ item_first, const int* __restrict__ part_first, int cout,
        int normalize, const float* __restrict__ bias, int relu, float* __restrict__ out, float* __restrict__ part,
        unsigned* __restrict__ out_absmax) {
    __shared__ float s_part[16][64][4];
    __shared__ float s_norm[16];
    __shared__ __attribute__((aligned(16))) float4 s_pair[16][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_list = cnt[0], n_items = cnt[1];
    unsigned amax = 0;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        __syncthreads();  // the previous item's LDS partials are consumed
        int lo = 0, hi = n_list;  // long row li with item_first[li] <= item < item_first[li + 1]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (item_first[mid] <= item)
                lo = mid;
            else
                hi = mid;
        }
        const int li = lo, seg = item - item_first[li], pf = part_first[li];
        const i64 q = list[li];
        i64 b = rs[q], e = rs[q + 1];
        if (pf >= 0) {
            b += (i64)seg * CCH_SEG;
            e = e < b + CCH_SEG ? e : b + CCH_SEG;
        }
        const float ox = out_pos[3 * q], oy = out_pos[3 * q + 1], oz = out_pos[3 * q + 2];
        const float sc2 = 2.f * (1.f / extents[q]);
        f32x4 D = {0.f, 0.f, 0.f, 0.f};
        float norm_lane = 0.f;
        for (i64 p0 = b + 64 * (i64)wave; p0 < e; p0 += 64 * 16) {
            const int c = (int)((e - p0) < 64 ? (e - p0) : 64);
            cconv_batch_mma<SORTED>(inp_pos, inp_feat, nidx, nimp, p0, c, lane, ox, oy, oz, sc2, D, norm_lane, s_pair[wave]);
        }
        const float norm = wave_sum_dpp(norm_lane);
        {  // D[r] = basis element k = 16 (4 g + r) + n = 4 cell + channel
            float* flat = &s_part[wave][0][0];
#pragma unroll
            for (int r = 0; r < 4; ++r) flat[16 * (4 * (lane >> 4) + r) + (lane & 15)] = D[r];
        }
        if (lane == 0) s_norm[wave] = norm;
        __syncthreads();
        if (wave != 0) continue;
        float B0 = 0.f, B1 = 0.f, B2 = 0.f, B3 = 0.f, nt = 0.f;
        for (int w2 = 0; w2 < 16; ++w2) {  // wave order: deterministic
            B0 += s_part[w2][lane][0];
            B1 += s_part[w2][lane][1];
            B2 += s_part[w2][lane][2];
            B3 += s_part[w2][lane][3];
            nt += s_norm[w2];
        }
        if (pf >= 0) {  // one segment of several: the partial slot
            float* ps = part + (i64)(pf + seg) * CCH_LD;
            *reinterpret_cast<float4*>(ps + 4 * lane) = make_float4(B0, B1, B2, B3);
            if (lane == 0) ps[256] = nt;
        } else {
            amax = max(amax, cconv_heavy4_finish_row<COUT_MAX>(filters, B0, B1, B2, B3, nt, cout, normalize, bias, relu,
                                                               out + q * cout, lane));
        }
    }
    if (out_absmax && wave == 0) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = max(amax, (unsigned)__shfl_xor((int)amax, o, 64));
        if (lane == 0 && amax) atomicMax(out_absmax, amax);
    }
}

// the rows of several segments: one wave per long row, segments added in order
template <int COUT_MAX>
__global__ __launch_bounds__(256) void k_cconv_heavy4_finish(const float* __restrict__ filters, const i64* __restrict__ rs,
                                                             const int32_t* __restrict__ list, const int* __restrict__ cnt,
                                                             const int* __restrict__ item_first,
                                                             const int* __restrict__ part_first, int cout, int normalize,
                                                             const float* __restrict__ bias, int relu, float* __restrict__ out,
                                                             const float* __restrict__ part, unsigned* __restrict__ out_absmax) {
    const int lane = threadIdx.x & 63;
    const int n_list = cnt[0];
    unsigned amax = 0;
    for (int li = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6); li < n_list; li += (int)((gridDim.x * blockDim.x) >> 6)) {
        const int pf = part_first[li];
        if (pf < 0) continue;  // (wave uniform) finished by its one block
        const int nseg = item_first[li + 1] - item_first[li];
        float B0 = 0.f, B1 = 0.f, B2 = 0.f, B3 = 0.f, nt = 0.f;
        for (int sgm = 0; sgm < nseg; ++sgm) {
            const float* ps = part + (i64)(pf + sgm) * CCH_LD;
            const float4 v = *reinterpret_cast<const float4*>(ps + 4 * lane);
            B0 += v.x;
            B1 += v.y;
            B2 += v.z;
            B3 += v.w;
            nt += ps[256];
        }
        const i64 q = list[li];
        amax = max(amax, cconv_heavy4_finish_row<COUT_MAX>(filters, B0, B1, B2, B3, nt, cout, normalize, bias, relu,
                                                           out + q * cout, lane));
    }
    if (out_absmax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = max(amax, (unsigned)__shfl_xor((int)amax, o, 64));
        if (lane == 0 && amax) atomicMax(out_absmax, amax);
    }
}

// ------------------------------------------------------------------------------------------
// a12 scalar reference kernel: one thread per (row, out channel).  Any cin / cout / strides.
// Used for shapes the MFMA kernel does not take and as an in-library cross-check (algo=1).
// ------------------------------------------------------------------------------------------
__global__ void k_sconv_scalar(asr_sparse_conv_args a) {
    i64 t = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    i64 q = t / a.cout;
    int oc = (int)(t % a.cout);
    if (q >= a.num_out) return;
    if (a.row_perm) q = a.row_perm[q];  // row list (global row indices)
    float acc = 0.f, norm = 0.f;
    for (i64 p = a.neighbors_row_splits[q]; p < a.neighbors_row_splits[q + 1]; ++p) {
        int32_t i = a.neighbors_index[p];
        float w = a.neighbors_importance ? a.neighbors_importance[p]
                                         : (a.inp_importance ? a.inp_importance[i] : 1.f);
        norm += w;
        const float* f = a.inp_features + (i64)i * a.inp_ld;
        const float* W = a.filters + ((i64)a.neighbors_kernel_index[p] * a.cin) * a.cout + oc;
        for (int ic = 0; ic < a.cin; ++ic) acc += W[(i64)ic * a.cout] * (w * f[ic]);
    }
    if (a.normalize && norm != 0.f) acc /= norm;
    if (a.bias) acc += a.bias[oc];
    if (a.relu) acc = fmaxf(acc, 0.f);
    if (a.residual) acc += a.residual[q * a.residual_ld + oc];
    a.out[q * a.out_ld + oc] = acc;
    if (a.out_importance && oc == 0) a.out_importance[q] = norm;
}

// ------------------------------------------------------------------------------------------
// a12 MFMA kernel (v2).
// Block = 256 threads = 4 waves; tile = 64 output rows (16 per wave) x NCOL = NT*16 output
// columns (several column chunks per row tile share an XCD, see the block index mapping).  Rows come from an optional permutation that
// groups rows with equal kernel-slot signatures (asr_geom_row_groups): ~8 of 55 slots are
// occupied per voxel and without regrouping a 16-row MFMA tile touches ~23 distinct slots.
// A dense slot table nbr[64][55] is built in LDS from the CSR rows.  The block walks the slots
// present in ANY of its rows; per slot and KC-deep cin chunk the W[k] panel [KC x NCOL] is staged
// through LDS (double buffered, global loads issued before the MFMAs of the previous panel,
// LDS write after them), shared by the four waves; a wave whose 16 rows lack the slot skips the
// MFMAs.  Lane (r = l&15, g = l>>4) gathers the float4 f[idx(r,k)][c0+16j+4g .. +3]; MFMA step t
// contracts cin index c0+16j+4g+t (the four k-lanes of v_mfma_f32_16x16x4_f32 are mapped to a
// permuted cin order so the gather is 16 B per lane).  No branch sits between a load and its
// consumer (a guarded load costs a vmcnt(0) per MFMA).  Epilogue: normalise, bias, ReLU, residual,
// strided store (zero-copy concat).
// ------------------------------------------------------------------------------------------
constexpr int NBR_LD = 57;  // odd stride: conflict-free column reads

// DUAL (with IMP): two filter banks in one pass (SparseConvBlock conv1a + conv1b): columns
// [0, cout) are the plain convolution, columns [cout, cout + cout_b) are weighted by the neighbour
// importance and normalised.  cout % 16 == 8 and cout_b == 8, so bank b is the upper half of the last
// column tile; that tile gets a second accumulator fed with the importance-scaled A operand.
template <int NT, int KC, bool IMP, int WAVES, bool DUAL = false>
__global__ __launch_bounds__(WAVES * 64, WAVES == 8 ? 2 : (NT >= 16 ? 3 : 4)) void k_sconv_mfma(asr_sparse_conv_args a,
                                                    const float* __restrict__ zeros) {
    constexpr int TM = WAVES * 16;   // rows per block: more rows share one staged weight panel
    constexpr int NTHR = WAVES * 64;
    constexpr int NCOL = NT * 16;
    constexpr int BLD = NCOL + 4;            // (4g+t)*BLD mod 32 separates the two 16-lane halves
    constexpr int NJ = KC / 16;              // float4 gathers per lane per panel
    constexpr int PV = KC * NCOL / 4;               // float4 per panel
    constexpr int SV = (PV + NTHR - 1) / NTHR;      // float4 staged per thread per panel
    constexpr bool SV_EXACT = PV % NTHR == 0;       // else the last threads stage nothing
    __shared__ int s_nbr[TM * NBR_LD];
    __shared__ float s_w[IMP ? TM * NBR_LD : 1];
    __shared__ float s_norm[TM];
    __shared__ int s_row[TM];
    __shared__ unsigned long long s_mask[TM];
    __shared__ __attribute__((aligned(16))) float s_B[2][KC * BLD];

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    // (an XCD-contiguous tile order was measured and is slower: tiles are sorted by slot mask, so a
    // contiguous range per XCD is unbalanced -- 59 -> 70 ms for the 10 M U-Net)
    // 1-D grid.  With several column chunks per row tile the chunks of one tile get consecutive slots
    // on ONE XCD (workgroups go round-robin over the 8 XCDs: id % 8), so that the repeated feature
    // gathers of chunks 1.. hit that XCD's L2 instead of HBM: id = ((tile / 8) * nY + y) * 8 + tile % 8.
    const int nY = (a.cout + (DUAL ? a.cout_b : 0) + NCOL - 1) / NCOL;
    i64 tile = blockIdx.x;
    int ychunk = 0;
    if (nY > 1) {
        const i64 r8 = blockIdx.x >> 3;
        ychunk = (int)(r8 % nY);
        tile = (r8 / nY) * 8 + (blockIdx.x & 7);
    }
    const i64 row0 = tile * TM;
    if (row0 >= a.num_out) return;  // grid padding (whole block, before any barrier)
    const int n0 = ychunk * NCOL;
    const int K = a.kernel_size;
    const int cin = a.cin;
    const int ca = a.cout;                              // bank a width
    const int cout = a.cout + (DUAL ? a.cout_b : 0);    // all output columns of this launch
    const bool has_b = DUAL && ychunk == nY - 1;  // this block holds the bank-b half tile

    for (int i = tid; i < TM * NBR_LD; i += NTHR) s_nbr[i] = -1;
    __syncthreads();
    {
        // slot table: TPR = 4 threads walk one CSR row (entries p, p+4, ...) -- the row walk is a chain
        // of dependent global loads and sits in front of every tile
        constexpr int TPR = NTHR / TM;
        const int prow = tid / TPR, pj = tid % TPR;
        // entry row0 + prow of the row list; with a row list the rows are GLOBAL indices that may exceed
        // num_out (a rank of a sharded run lists only the rows it owns)
        i64 q = row0 + prow;
        const bool valid = q < a.num_out;
        if (valid && a.row_perm) q = a.row_perm[q];
        unsigned long long m = 0;
        float norm = 0.f;
        if (valid) {
            const i64 pe = a.neighbors_row_splits[q + 1];
            for (i64 p = a.neighbors_row_splits[q] + pj; p < pe; p += TPR) {
                int k = a.neighbors_kernel_index[p];
                int32_t i = a.neighbors_index[p];
                if (k >= K) continue;  // malformed input: slot outside the filter
                s_nbr[prow * NBR_LD + k] = i;
                m |= 1ull << k;
                float w = 1.f;
                if (IMP) {
                    w = a.neighbors_importance ? a.neighbors_importance[p] : a.inp_importance[i];
                    s_w[prow * NBR_LD + k] = w;
                }
                norm += w;
            }
        } else {
            q = -1;
        }
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) {  // the TPR lanes of a row are adjacent
            m |= __shfl_xor(m, o, 64);
            norm += __shfl_xor(norm, o, 64);
        }
        if (pj == 0) {
            s_row[prow] = (int)q;
            s_mask[prow] = m;
            s_norm[prow] = norm;
        }
    }
    __syncthreads();

    const int r = lane & 15, g = lane >> 4;
    const int lrow = wave * 16 + r;
    // slot masks: this wave's 16 rows and the whole block (both wave uniform)
    unsigned long long wmask = s_mask[lrow];
    unsigned long long bmask = s_mask[lane];
    if (TM > 64) bmask |= s_mask[64 + lane];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bmask |= __shfl_xor(bmask, o, 64);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) wmask |= __shfl_xor(wmask, o, 64);
    // readfirstlane returns a signed int: go through unsigned or bit 31 smears into bits 32..63
    wmask = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)wmask) |
            ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(wmask >> 32)) << 32);
    bmask = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)bmask) |
            ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(bmask >> 32)) << 32);
    bmask &= (1ull << K) - 1;  // K <= 56

    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc_b = {0.f, 0.f, 0.f, 0.f};

    const int ncol = lane & 15;
    const int npanel = (cin + KC - 1) / KC;
    // staging geometry: thread -> (panel row, float4 column) for SV float4 each

    // ---- software pipeline over (slot, panel) steps ------------------------------------------
    // Prefetch distance is TWO steps, held in registers: at step i the weight panel and the feature
    // gather of step i+2 are issued, the panel of step i+1 (loaded during step i-1) is written to
    // the other LDS buffer after the MFMAs.  One step of MFMAs (0.4 us for the narrow layers) is
    // shorter than an L2 / HBM round trip, two are not.  Slots alternate by step parity (the loop is
    // unrolled by two) so that no in-flight register is ever moved, and every load is unconditional
    // (finished sequences read the zero line): a guarded or moved load would turn the counted
    // s_waitcnt vmcnt(n) into vmcnt(0).
    // sequence state (wave uniform scalars): remaining slot mask, current slot (-1 = finished), panel
#define ASR_SEQ_ADVANCE(todo, k, p)                          \
    if ((k) >= 0 && ++(p) == npanel) {                       \
        (p) = 0;                                             \
        (todo) &= (todo)-1;                                  \
        (k) = (todo) ? __builtin_ctzll(todo) : -1;           \
    }
    f32x4 stage0[SV], stage1[SV];  // separate objects (not stage[2][SV]): they must stay in registers
    f32x4 a_q0[NJ], a_q1[NJ];
    float imp_q0 = 1.f, imp_q1 = 1.f;

    auto load_panel = [&](const int qk, const int qp, f32x4 (&st)[SV]) __attribute__((always_inline)) {
        const float* Wk = a.filters + (i64)(qk < 0 ? 0 : qk) * cin * ca;
        const float* Wkb = DUAL ? a.filters_b + (i64)(qk < 0 ? 0 : qk) * cin * a.cout_b : nullptr;
#pragma unroll
        for (int s = 0; s < SV; ++s) {
            int e = tid + s * NTHR;            // float4 index inside the panel
            int pr = e / (NCOL / 4);          // panel row
            int pc = (e % (NCOL / 4)) * 4;    // column
            int ci = qp * KC + pr, col = n0 + pc;
            // cout % 4 == 0 and col % 4 == 0: a float4 is entirely inside or outside the row.
            // Out-of-range entries read a zero line instead of being masked after the load: a
            // select on the loaded value would force the vmcnt wait in front of the MFMAs.
            bool ok = qk >= 0 && ci < cin && col < cout && (SV_EXACT || e < PV);
            const float* src = ok ? Wk + (i64)ci * ca + col : zeros;
            if (DUAL && ok && col >= ca) src = Wkb + (i64)ci * a.cout_b + (col - ca);
            st[s] = *reinterpret_cast<const f32x4*>(src);
        }
    };
    auto store_panel = [&](int buf, const f32x4 (&st)[SV]) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < SV; ++s) {
            int e = tid + s * NTHR;
            int pr = e / (NCOL / 4);
            int pc = (e % (NCOL / 4)) * 4;
            if (SV_EXACT || e < PV) *reinterpret_cast<f32x4*>(&s_B[buf][pr * BLD + pc]) = st[s];
        }
    };
    // The neighbour of (row, slot) is looked up in the LDS slot table once per SLOT, not once per panel:
    // that ds_read sits at the head of every step's dependency chain (ds_read -> address -> global load)
    // right after the barrier, when no wave of the block has MFMAs to issue.
    int cache_k = -2;
    const float* cache_row = zeros;
    bool cache_valid = false;
    float cache_imp = 0.f;
    auto gather_a = [&](const int qk, const int qp, f32x4 (&aq)[NJ], float& imp) __attribute__((always_inline)) {
        if (qk != cache_k) {  // wave uniform
            cache_k = qk;
            const int idx = qk < 0 ? -1 : s_nbr[lrow * NBR_LD + qk];
            cache_valid = idx >= 0;
            cache_row = a.inp_features + (i64)(cache_valid ? idx : 0) * a.inp_ld;
            if (IMP) cache_imp = (cache_valid && qk >= 0) ? s_w[lrow * NBR_LD + qk] : 0.f;
        }
        if (IMP) imp = cache_imp;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            int c = qp * KC + 16 * j + 4 * g;
            const float* src = (cache_valid && c < cin) ? cache_row + c : zeros;
            aq[j] = *reinterpret_cast<const f32x4*>(src);
        }
    };

    unsigned long long todo1 = bmask;  // the sequence position one step ahead of the current one
    int k_cur = bmask ? __builtin_ctzll(bmask) : -1, p_cur = 0;
    int k1 = k_cur, p1 = 0;
    ASR_SEQ_ADVANCE(todo1, k1, p1)
    if (k_cur >= 0) {
        load_panel(k_cur, p_cur, stage0);
        gather_a(k_cur, p_cur, a_q0, imp_q0);
        store_panel(0, stage0);
        load_panel(k1, p1, stage1);
        gather_a(k1, p1, a_q1, imp_q1);
    }
    __syncthreads();
    int buf = 0;
    // one step: aq = gather of this step, st_next = panel of the next step (in flight), st_free free
    auto step = [&](f32x4 (&aq)[NJ], float& imp, f32x4 (&st_free)[SV], f32x4 (&st_next)[SV]) __attribute__((always_inline)) {
        int k2 = k1, p2 = p1;
        ASR_SEQ_ADVANCE(todo1, k2, p2)   // todo1 now belongs to the position two steps ahead
        f32x4 a_cur[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) a_cur[j] = aq[j];
        const float imp_cur = imp;
        // panel of the NEXT step (loaded one step ago) goes to the other LDS buffer right away: that
        // buffer was released by the barrier that ended the previous step, and the ds_write + its wait
        // no longer sit between the last MFMA and the barrier
        store_panel(buf ^ 1, st_next);
        load_panel(k2, p2, st_free);    // two steps ahead; stays in flight across the barrier
        gather_a(k2, p2, aq, imp);
        // a wave whose 16 rows lack slot k_cur skips the MFMAs (PMC: executing them
        // unconditionally doubles the MFMA work)
        if ((wmask >> k_cur) & 1) {
            const float* sb = &s_B[buf][(4 * g) * BLD + ncol];
            float bv[2][NT];
#pragma unroll
            for (int nb = 0; nb < NT; ++nb) bv[0][nb] = sb[nb * 16];
            __builtin_amdgcn_s_setprio(1);  // a wave with MFMAs to issue wins over the ones issuing loads (1 %)
#pragma unroll
            for (int jt = 0; jt < NJ * 4; ++jt) {
                const int j = jt >> 2, t = jt & 3;
                if (jt + 1 < NJ * 4) {
                    const int j1 = (jt + 1) >> 2, t1 = (jt + 1) & 3;
#pragma unroll
                    for (int nb = 0; nb < NT; ++nb)
                        bv[(jt + 1) & 1][nb] = sb[(16 * j1 + t1) * BLD + nb * 16];
                }
                float av = t == 0 ? a_cur[j].x : t == 1 ? a_cur[j].y : t == 2 ? a_cur[j].z : a_cur[j].w;
                if (IMP && !DUAL) av *= imp_cur;
#pragma unroll
                for (int nb = 0; nb < NT; ++nb)
                    acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[jt & 1][nb], acc[nb], 0, 0, 0);
                if (DUAL && has_b)
                    acc_b = __builtin_amdgcn_mfma_f32_16x16x4f32(av * imp_cur, bv[jt & 1][NT - 1], acc_b, 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
        buf ^= 1;
        k_cur = k1;
        p_cur = p1;
        k1 = k2;
        p1 = p2;
    };
    while (k_cur >= 0) {
        step(a_q0, imp_q0, stage0, stage1);
        if (k_cur < 0) break;
        step(a_q1, imp_q1, stage1, stage0);
    }
#undef ASR_SEQ_ADVANCE

    // epilogue: acc[nb][i] is C[row = 4*g + i][col = ncol] of the wave's 16 x 16 block.
    // All loads (bias, residual) are unconditional (absent -> zero line) and issued before the
    // stores: a guarded load per element would cost one memory round trip per store.
    float bv[NT];
#pragma unroll
    for (int nb = 0; nb < NT; ++nb) {
        const int col = n0 + nb * 16 + ncol;
        const float* bp = (a.bias && col < ca) ? a.bias + col : zeros;
        if (DUAL && a.bias_b && col >= ca && col < cout) bp = a.bias_b + (col - ca);
        bv[nb] = *bp;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int lr = wave * 16 + 4 * g + i;
        const i64 q = s_row[lr];
        const bool rowok = q >= 0;
        const float norm = s_norm[lr];
        const bool do_norm = a.normalize && norm != 0.f;
        float res[NT];
#pragma unroll
        for (int nb = 0; nb < NT; ++nb) {
            const int col = n0 + nb * 16 + ncol;
            const float* rp = (a.residual && rowok && col < cout) ? a.residual + q * a.residual_ld + col : zeros;
            res[nb] = *rp;
        }
#pragma unroll
        for (int nb = 0; nb < NT; ++nb) {
            const int col = n0 + nb * 16 + ncol;
            float v = acc[nb][i];
            if (DUAL) {  // bank b: importance weighted + normalised; bank a: plain
                const bool colb = col >= ca;
                if (nb == NT - 1 && has_b && colb) v = acc_b[i];
                v = (colb && do_norm) ? v / norm : v;
            } else {
                v = do_norm ? v / norm : v;
            }
            v += bv[nb];
            if (a.relu) v = fmaxf(v, 0.f);
            v += res[nb];
            if (rowok && col < cout) a.out[q * a.out_ld + col] = v;
        }
    }
    if (a.out_importance && ychunk == 0 && tid < TM && s_row[tid] >= 0)
        a.out_importance[s_row[tid]] = s_norm[tid];
}

// ------------------------------------------------------------------------------------------
// reduce_subarrays_sum (with optional gather)
// ------------------------------------------------------------------------------------------
__global__ void k_reduce_rows(const float* values, const int32_t* gidx, const i64* rs, i64 rows,
                              float* out) {
    i64 q = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (q >= rows) return;
    float s = 0.f;
    for (i64 p = rs[q]; p < rs[q + 1]; ++p) s += gidx ? values[gidx[p]] : values[p];
    out[q] = s;
}

// ------------------------------------------------------------------------------------------
// a14 decoder MLP (3+c -> h1 -> h2 -> 2, zero shifts) + sdf scale; thread per voxel,
// weights staged in LDS.
// ------------------------------------------------------------------------------------------
constexpr int DEC_MAX = 64;
// CT/H1T/H2T > 0: compile-time widths (registers, fully unrolled); 0: runtime widths (generic)
template <int CT, int H1T, int H2T>
__global__ __launch_bounds__(256) void k_decode(const float* __restrict__ code, i64 v, int c_,
                                                const float* __restrict__ w1,
                                                const float* __restrict__ b1, int h1_,
                                                const float* __restrict__ w2,
                                                const float* __restrict__ b2, int h2_,
                                                const float* __restrict__ w3,
                                                const float* __restrict__ sizes,
                                                float* __restrict__ out, const int32_t* __restrict__ rows = nullptr) {
    const int c = CT ? CT : c_, h1 = H1T ? H1T : h1_, h2 = H2T ? H2T : h2_;
    extern __shared__ float s_w[];
    float* sw1 = s_w;                 // [h1][c]   (shift columns dropped: shifts are zero)
    float* sb1 = sw1 + h1 * c;        // [h1]
    float* sw2 = sb1 + h1;            // [h2][h1]
    float* sb2 = sw2 + h2 * h1;       // [h2]
    float* sw3 = sb2 + h2;            // [2][h2]
    for (int i = threadIdx.x; i < h1 * c; i += blockDim.x) sw1[i] = w1[(i / c) * (3 + c) + 3 + i % c];
    for (int i = threadIdx.x; i < h1; i += blockDim.x) sb1[i] = b1[i];
    for (int i = threadIdx.x; i < h2 * h1; i += blockDim.x) sw2[i] = w2[i];
    for (int i = threadIdx.x; i < h2; i += blockDim.x) sb2[i] = b2[i];
    for (int i = threadIdx.x; i < 2 * h2; i += blockDim.x) sw3[i] = w3[i];
    __syncthreads();
    i64 q = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (q >= v) return;
    if (rows) q = rows[q];  // a row list instead of rows 0 .. v
    float x[CT ? CT : DEC_MAX], f1[H1T ? H1T : DEC_MAX];
    if (CT) {
#pragma unroll
        for (int k = 0; k < (CT ? CT : 1); k += 4) {
            float4 t = *reinterpret_cast<const float4*>(code + q * c + k);
            x[k] = t.x;
            x[k + 1] = t.y;
            x[k + 2] = t.z;
            x[k + 3] = t.w;
        }
#pragma unroll
        for (int j = 0; j < (H1T ? H1T : 1); ++j) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < (CT ? CT : 1); ++k) s += x[k] * sw1[j * c + k];
            s += sb1[j];
            f1[j] = fmaxf(s, 0.f);
        }
    } else {
        for (int k = 0; k < c; ++k) x[k] = code[q * c + k];
        for (int j = 0; j < h1; ++j) {
            float s = 0.f;
            for (int k = 0; k < c; ++k) s += x[k] * sw1[j * c + k];
            s += sb1[j];
            f1[j] = fmaxf(s, 0.f);
        }
    }
    float o0 = 0.f, o1 = 0.f;
    if (CT) {
#pragma unroll 4
        for (int j = 0; j < (H2T ? H2T : 1); ++j) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < (H1T ? H1T : 1); ++k) s += f1[k] * sw2[j * h1 + k];
            s += sb2[j];
            s = fmaxf(s, 0.f);
            o0 += s * sw3[j];
            o1 += s * sw3[h2 + j];
        }
    } else {
        for (int j = 0; j < h2; ++j) {
            float s = 0.f;
            for (int k = 0; k < h1; ++k) s += f1[k] * sw2[j * h1 + k];
            s += sb2[j];
            s = fmaxf(s, 0.f);
            o0 += s * sw3[j];
            o1 += s * sw3[h2 + j];
        }
    }
    if (sizes) o0 *= sizes[q];
    out[2 * q] = o0;
    out[2 * q + 1] = o1;
}

// The same decoder for the widths of the released model (c = h1 = h2 = 32) on the f32 matrix cores: a wave owns groups of 16
// voxels, every layer is out[16 voxels][32] = act[16][32] . W^T as 2 x 8 v_mfma_f32_16x16x4_f32 (exact f32 products, f32
// accumulate), the third layer one tile of which two columns are real.  The thread-per-voxel kernel above issues one LDS
// broadcast read per FMA (2 112 of each per voxel: 0.36 ms at 10 M points); here the weights stay in registers as B fragments
// and a layer's result goes through a 2 KB LDS tile per wave from the accumulator layout to the A layout.  k order: MFMA step s of
// k-lane kk contracts k = 8 kk + s on both operands, so that a lane's eight A values are 32 contiguous bytes.
__global__ __launch_bounds__(256) void k_decode_mfma(const float* __restrict__ code, i64 v, const float* __restrict__ w1,
                                                     const float* __restrict__ b1, const float* __restrict__ w2,
                                                     const float* __restrict__ b2, const float* __restrict__ w3,
                                                     const float* __restrict__ sizes, float* __restrict__ out,
                                                     const int32_t* __restrict__ rows) {  // rows: a row list instead of 0 .. v
    constexpr int C = 32, LD = 36;
    __shared__ __attribute__((aligned(16))) float s_t[4][16][LD];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n = lane & 15, kk = lane >> 4;
    float B1[8][2], B2[8][2], B3[8], bias1[2], bias2[2];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
#pragma unroll
        for (int T = 0; T < 2; ++T) {
            B1[s][T] = w1[(i64)(16 * T + n) * (3 + C) + 3 + 8 * kk + s];  // (the three shift columns are zero inputs)
            B2[s][T] = w2[(i64)(16 * T + n) * C + 8 * kk + s];
        }
        B3[s] = n < 2 ? w3[n * C + 8 * kk + s] : 0.f;
    }
#pragma unroll
    for (int T = 0; T < 2; ++T) {
        bias1[T] = b1[16 * T + n];
        bias2[T] = b2[16 * T + n];
    }
    const i64 groups = (v + 15) / 16;
    const i64 wave0 = (i64)blockIdx.x * 4 + wave, nwaves = (i64)gridDim.x * 4;
    float (*st)[LD] = s_t[wave];
    auto load_rows = [&](i64 grp, float4& t0, float4& t1) __attribute__((always_inline)) {
        i64 q = grp * 16 + n;
        t0 = make_float4(0.f, 0.f, 0.f, 0.f);
        t1 = t0;
        if (grp < groups && q < v) {
            if (rows) q = rows[q];
            t0 = *reinterpret_cast<const float4*>(code + q * C + 8 * kk);
            t1 = *reinterpret_cast<const float4*>(code + q * C + 8 * kk + 4);
        }
    };
    float4 n0, n1;  // the next group's rows, in flight while this group runs through the layers
    load_rows(wave0, n0, n1);
    for (i64 grp = wave0; grp < groups; grp += nwaves) {
        float x[8];
        x[0] = n0.x; x[1] = n0.y; x[2] = n0.z; x[3] = n0.w;
        x[4] = n1.x; x[5] = n1.y; x[6] = n1.z; x[7] = n1.w;
        load_rows(grp + nwaves, n0, n1);
#pragma unroll
        for (int layer = 0; layer < 2; ++layer) {
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[s], layer == 0 ? B1[s][0] : B2[s][0], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[s], layer == 0 ? B1[s][1] : B2[s][1], a1, 0, 0, 0);
            }
            // a_T[i] = act[voxel 4 kk + i][column 16 T + n]: bias, ReLU, then through LDS into the A layout
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                st[4 * kk + i][n] = fmaxf(a0[i] + (layer == 0 ? bias1[0] : bias2[0]), 0.f);
                st[4 * kk + i][16 + n] = fmaxf(a1[i] + (layer == 0 ? bias1[1] : bias2[1]), 0.f);
            }
            __builtin_amdgcn_wave_barrier();
            const float4 t0 = *reinterpret_cast<const float4*>(&st[n][8 * kk]);
            const float4 t1 = *reinterpret_cast<const float4*>(&st[n][8 * kk + 4]);
            x[0] = t0.x; x[1] = t0.y; x[2] = t0.z; x[3] = t0.w;
            x[4] = t1.x; x[5] = t1.y; x[6] = t1.z; x[7] = t1.w;
        }
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 8; ++s) o = __builtin_amdgcn_mfma_f32_16x16x4f32(x[s], B3[s], o, 0, 0, 0);
        if (n < 2) {  // o[i] = out[voxel 4 kk + i][n]
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                i64 vox = grp * 16 + 4 * kk + i;
                if (vox < v) {
                    if (rows) vox = rows[vox];
                    out[2 * vox + n] = (n == 0 && sizes) ? o[i] * sizes[vox] : o[i];
                }
            }
        }
    }
}

}  // namespace

// ==========================================================================================
int asr_conv_agg_importance(asr_hip_context* ctx, const float* compat, const float* dist, i64 n,
                            float* out) {
    if (n <= 0) return ASR_HIP_OK;
    k_agg_importance<<<grid_for(n, 256), 256, 0, ctx->stream>>>(compat, dist, n, out);
    ASR_CHECK_LAUNCH(ctx);
    return ASR_HIP_OK;
}

int asr_conv_cconv(asr_hip_context* ctx, const float* filters, const float* out_pos,
                   const float* extents, const float* inp_pos, const float* inp_feat,
                   const int32_t* nidx, const float* nimp, const i64* rs, i64 num_out, int cin,
                   int cout, int normalize, const float* bias, int relu, float* out, int sorted4, unsigned* out_absmax) {
    if (num_out <= 0) return ASR_HIP_OK;
    if (sorted4 && cin != 4) ASR_FAIL(ctx, ASR_HIP_EINVAL, "continuous_conv: the Morton-ordered layout needs cin == 4");
    if (cout < 1 || cout > 64) ASR_FAIL(ctx, ASR_HIP_EINVAL, "continuous_conv: cout must be 1..64");
    if (cin < 1) ASR_FAIL(ctx, ASR_HIP_EINVAL, "continuous_conv: cin must be >= 1");
    // 1024-thread blocks (80 KB of LDS: filter slice + per-wave slabs), 2 per CU: 32 waves per CU hide the
    // dependent row_splits -> index -> position load chain of every voxel
    unsigned blocks = grid_for(num_out * 64, 1024);
    if (blocks > 256 * 2) blocks = 256 * 2;  // persistent: the filter slice is staged once per block
    const bool mfma_path = cin == 4 && cout <= 32 && !ctx->opt.cconv_valu;  // (these kernels keep out_absmax themselves)
    // long rows: collect, then one 1024-thread block per row
    int32_t* heavy = arena_alloc<int32_t>(ctx->scratch, (size_t)num_out);
    int* d_count = arena_alloc<int>(ctx->scratch, 4);
    if (!heavy || !d_count) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    ASR_HIP_CHECK(ctx, hipMemsetAsync(d_count, 0, 4 * sizeof(int), ctx->stream));  // [0] long rows, [1] their items, [2] tickets
    k_cconv_heavy_list<<<grid_for(num_out, 256), 256, 0, ctx->stream>>>(rs, num_out, CCONV_HEAVY, heavy,
                                                                      d_count);
    ASR_CHECK_LAUNCH(ctx);
#define ASR_LAUNCH_CCONV_S(C_, S_)                                                                       \
    k_cconv<C_, S_><<<blocks, 1024, 0, ctx->stream>>>(filters, out_pos, extents, inp_pos, inp_feat, nidx,   \
                                                     nimp, rs, num_out, cin, cout, normalize, bias, relu, \
                                                     out, CCONV_HEAVY);
#define ASR_LAUNCH_CCONV(C_)        \
    if (sorted4)                    \
        ASR_LAUNCH_CCONV_S(C_, true) \
    else                            \
        ASR_LAUNCH_CCONV_S(C_, false)
#define ASR_LAUNCH_CCONV_HEAVY_S(C_, S_)                                                                  \
    k_cconv_heavy<C_, S_><<<512, 1024, 0, ctx->stream>>>(filters, out_pos, extents, inp_pos, inp_feat, nidx, \
                                                         nimp, rs, heavy, d_count, cin, cout, normalize,   \
                                                         bias, relu, out, mfma_path ? out_absmax : nullptr);
#define ASR_LAUNCH_CCONV_HEAVY(C_)        \
    if (sorted4)                          \
        ASR_LAUNCH_CCONV_HEAVY_S(C_, true) \
    else                                  \
        ASR_LAUNCH_CCONV_HEAVY_S(C_, false)
    if (mfma_path) {
        // contraction on the matrix cores: one persistent 16-wave block per CU (138 KB of LDS)
        if (sorted4)
            k_cconv_mfma<true, ASR_CCONV_GROUP><<<256, cconv_waves(ASR_CCONV_GROUP) * 64, 0, ctx->stream>>>(filters, out_pos, extents, (const float4*)inp_pos, nullptr,
                                                             nullptr, nidx, nimp, rs, num_out, cout, normalize, bias,
                                                             relu, out, CCONV_HEAVY, nullptr, nullptr, out_absmax, d_count + 2);
        else
            k_cconv_mfma<false, ASR_CCONV_GROUP><<<256, cconv_waves(ASR_CCONV_GROUP) * 64, 0, ctx->stream>>>(filters, out_pos, extents, nullptr, inp_pos, inp_feat,
                                                              nidx, nimp, rs, num_out, cout, normalize, bias, relu,
                                                              out, CCONV_HEAVY, nullptr, nullptr, out_absmax, d_count + 2);
    } else if (cout <= 8)
        ASR_LAUNCH_CCONV(8)
    else if (cout <= 32)
        ASR_LAUNCH_CCONV(32)
    else
        ASR_LAUNCH_CCONV(64)
    ASR_CHECK_LAUNCH(ctx);
    if (mfma_path) {  // long rows cut into segments (k_cconv_heavy4)
        int* item_first = arena_alloc<int>(ctx->scratch, (size_t)num_out + 2);
        int* part_first = arena_alloc<int>(ctx->scratch, (size_t)num_out + 2);
        float* part = arena_alloc<float>(ctx->scratch, (size_t)CCH_CAP * CCH_LD);
        if (!item_first || !part_first || !part) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
        k_cconv_heavy4_items<<<1, 1024, 0, ctx->stream>>>(rs, heavy, d_count, item_first, part_first);
        ASR_CHECK_LAUNCH(ctx);
        if (sorted4)
            k_cconv_heavy4<32, true><<<1024, 1024, 0, ctx->stream>>>(filters, out_pos, extents, inp_pos, inp_feat, nidx, nimp, rs,
                                                                    heavy, d_count, item_first, part_first, cout, normalize,
                                                                    bias, relu, out, part, out_absmax);
        else
            k_cconv_heavy4<32, false><<<1024, 1024, 0, ctx->stream>>>(filters, out_pos, extents, inp_pos, inp_feat, nidx, nimp, rs,
                                                                     heavy, d_count, item_first, part_first, cout, normalize,
                                                                     bias, relu, out, part, out_absmax);
        ASR_CHECK_LAUNCH(ctx);
        k_cconv_heavy4_finish<32><<<64, 256, 0, ctx->stream>>>(filters, rs, heavy, d_count, item_first, part_first, cout, normalize,
                                                               bias, relu, out, part, out_absmax);
    } else if (cout <= 8)
        ASR_LAUNCH_CCONV_HEAVY(8)
    else if (cout <= 32)
        ASR_LAUNCH_CCONV_HEAVY(32)
    else
        ASR_LAUNCH_CCONV_HEAVY(64)
#undef ASR_LAUNCH_CCONV
#undef ASR_LAUNCH_CCONV_HEAVY
#undef ASR_LAUNCH_CCONV_S
#undef ASR_LAUNCH_CCONV_HEAVY_S
    ASR_CHECK_LAUNCH(ctx);
    if (out_absmax && !mfma_path) {  // the general kernels do not keep it: one pass over the output
        k_cconv_absmax<<<(unsigned)std::min<i64>((num_out * cout + 255) / 256, 8192), 256, 0, ctx->stream>>>(out, num_out * (i64)cout,
                                                                                                           out_absmax);
        ASR_CHECK_LAUNCH(ctx);
    }
    return ASR_HIP_OK;
}

// per-voxel interpolation matrices B[v][64 cells][4 channels] (un-normalised) and importance sums of the continuous
// conv: what the filter gradient contracts with (dW[cell][c][o] = sum_v B[v][cell][c] g[v][o] / norm[v])
int asr_conv_cconv_basis(asr_hip_context* ctx, const float* out_pos, const float* extents, const float* inp_pos,
                         const float* inp_feat, const int32_t* nidx, const float* nimp, const i64* rs, i64 num_out,
                         float* basis_out, float* norm_out) {
    if (num_out <= 0) return ASR_HIP_OK;
    const float* zeros = nullptr;
    ASR_TRY(asr_ctx_zeros(ctx, &zeros));  // stands in for the (unused) filter matrix
    int* tickets = arena_alloc<int>(ctx->scratch, 4);
    if (!tickets) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    ASR_HIP_CHECK(ctx, hipMemsetAsync(tickets, 0, 4 * sizeof(int), ctx->stream));
    k_cconv_mfma<false, ASR_CCONV_GROUP><<<256, cconv_waves(ASR_CCONV_GROUP) * 64, 0, ctx->stream>>>(zeros, out_pos, extents, nullptr, inp_pos, inp_feat, nidx, nimp,
                                                      rs, num_out, 0, 0, nullptr, 0, nullptr,
                                                      (i64)0x7fffffff, basis_out, norm_out, nullptr, tickets);
    ASR_CHECK_LAUNCH(ctx);
    return ASR_HIP_OK;
}

int asr_conv_sparse(asr_hip_context* ctx, const asr_sparse_conv_args* pa) {
    asr_sparse_conv_args a = *pa;
    if (a.num_out <= 0) return ASR_HIP_OK;
    if (a.kernel_size < 1 || a.kernel_size > 56)
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "sparse_conv: kernel_size must be 1..56");
    if (a.cin < 1 || a.cout < 1) ASR_FAIL(ctx, ASR_HIP_EINVAL, "sparse_conv: bad channel count");
    if (a.filters_b && a.cout_b < 1) ASR_FAIL(ctx, ASR_HIP_EINVAL, "sparse_conv: bad channel count");
    if (a.inp_ld < a.cin || a.out_ld < a.cout + (a.filters_b ? a.cout_b : 0))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "sparse_conv: row stride smaller than channel count");
    bool mfma_ok = (a.cin % 4 == 0) && (a.cout % 4 == 0) && (a.inp_ld % 4 == 0) &&
                   ((uintptr_t)a.filters % 16 == 0) &&
                   ((uintptr_t)a.inp_features % 16 == 0);
    int algo = a.algo;
    if (algo == 0) algo = mfma_ok ? 2 : 1;
    if (algo == 2 && !mfma_ok)
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "sparse_conv: MFMA path needs cin %% 4 == 0 and 16 B rows");
    if (a.filters_b && algo != 2)
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "sparse_conv: the second filter bank needs the MFMA path");
    if (algo == 1) {
        i64 total = a.num_out * a.cout;
        k_sconv_scalar<<<grid_for(total, 256), 256, 0, ctx->stream>>>(a);
        ASR_CHECK_LAUNCH(ctx);
        return ASR_HIP_OK;
    }
    if (a.cout % 4 != 0)
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "sparse_conv: MFMA path needs cout %% 4 == 0");
    const float* zeros = nullptr;
    ASR_TRY(asr_ctx_zeros(ctx, &zeros));
    if (ctx->dry_launch) return ASR_HIP_OK;  // preparation pass of the sharded network: checked and allocated, not launched
    const bool imp = a.inp_importance || a.neighbors_importance;
    const bool dual = a.filters_b != nullptr;
    const int ctot = a.cout + (dual ? a.cout_b : 0);
    if (dual && (!imp || a.cout % 16 != 8 || a.cout_b != 8 || a.residual || (uintptr_t)a.filters_b % 16 != 0))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "sparse_conv: second filter bank needs importance, cout %% 16 == 8, cout_b == 8");
    // widest column tile that fits cout, narrowed while the launch has too few blocks to fill
    // 256 CUs (coarse grids have only a few thousand rows; the gather is then repeated per
    // column chunk, which those levels can afford).  128-row blocks (8 waves) halve the weight
    // panel traffic per MFMA and are used whenever they still give enough blocks.
    // force_nt / force_waves (tests): pick the instance regardless of the problem size.
    int nt = ctot > 128 ? 16 : ctot > 64 ? 8 : ctot > 32 ? 4 : ctot > 16 ? 2 : 1;
    const i64 tiles64 = (a.num_out + 63) / 64;
    if (a.force_nt) {
        if (a.force_nt != 1 && a.force_nt != 2 && a.force_nt != 4 && a.force_nt != 8 && a.force_nt != 16)
            ASR_FAIL(ctx, ASR_HIP_EINVAL, "sparse_conv: force_nt must be 1, 2, 4, 8 or 16");
        nt = a.force_nt;
    } else {
        while (nt > 2 && tiles64 * ((ctot + nt * 16 - 1) / (nt * 16)) < ctx->opt.sconv_min_blocks) nt >>= 1;
    }
    while (dual && ctot % (nt * 16) != 0) nt >>= 1;  // bank b must be the last tile of the last column chunk
    if (a.force_nt && nt != a.force_nt)
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "sparse_conv: force_nt %d does not divide the two-bank width %d", a.force_nt, ctot);
    const i64 tiles128 = (a.num_out + 127) / 128;
    bool wide = nt >= 2 && tiles128 * ((ctot + nt * 16 - 1) / (nt * 16)) >= ctx->opt.sconv_wide_min;
    if (a.force_waves) {
        if ((a.force_waves != 4 && a.force_waves != 8) || (a.force_waves == 8 && nt < 2))
            ASR_FAIL(ctx, ASR_HIP_EINVAL, "sparse_conv: force_waves must be 4 or 8 (8 needs nt >= 2)");
        wide = a.force_waves == 8;
    }
    int kc_used = 0, waves_used = 0;
#define ASR_LAUNCH_SCONV(NT_, KC_, W_)                                                           \
    {                                                                                            \
        const i64 tiles_ = (a.num_out + W_ * 16 - 1) / (W_ * 16);                                \
        const i64 ny_ = (ctot + NT_ * 16 - 1) / (NT_ * 16);                                      \
        dim3 grid((unsigned)(ny_ > 1 ? ((tiles_ + 7) / 8) * 8 * ny_ : tiles_));                   \
        kc_used = KC_;                                                                           \
        waves_used = W_;                                                                         \
        if (dual)                                                                                \
            k_sconv_mfma<NT_, KC_, true, W_, true><<<grid, dim3(W_ * 64), 0, ctx->stream>>>(a, zeros); \
        else if (imp)                                                                            \
            k_sconv_mfma<NT_, KC_, true, W_><<<grid, dim3(W_ * 64), 0, ctx->stream>>>(a, zeros);   \
        else                                                                                     \
            k_sconv_mfma<NT_, KC_, false, W_><<<grid, dim3(W_ * 64), 0, ctx->stream>>>(a, zeros);  \
    }
#define ASR_LAUNCH_SCONV_W(NT_, KC_)   \
    if (wide)                          \
        ASR_LAUNCH_SCONV(NT_, KC_, 8)  \
    else                               \
        ASR_LAUNCH_SCONV(NT_, KC_, 4)
    // KC: deeper panels (8/32, 4/64) measured neutral; 2 and 1 column tiles take 32-deep panels when
    // cin <= 32 so that no zero-padded k step is executed (decblock0: 0.97 -> 0.6 ms per layer)
    switch (nt) {
        case 16: ASR_LAUNCH_SCONV_W(16, 16) break;
        case 8: ASR_LAUNCH_SCONV_W(8, 16) break;
        case 4: ASR_LAUNCH_SCONV_W(4, 32) break;
        case 2:
            if (a.cin <= 32) ASR_LAUNCH_SCONV_W(2, 32) else ASR_LAUNCH_SCONV_W(2, 64)  // no zero-padded k steps
            break;
        default:
            if (a.cin <= 32) ASR_LAUNCH_SCONV(1, 32, 4) else ASR_LAUNCH_SCONV(1, 64, 4)
            break;
    }
    {
        char key[48];
        snprintf(key, sizeof(key), "%d,%d,%d,%d,%d", nt, kc_used, imp ? 1 : 0, waves_used, dual ? 1 : 0);
        ++ctx->sconv_launches[key];
    }
#undef ASR_LAUNCH_SCONV_W
#undef ASR_LAUNCH_SCONV
    ASR_CHECK_LAUNCH(ctx);
    return ASR_HIP_OK;
}

int asr_conv_reduce(asr_hip_context* ctx, const float* values, const int32_t* gidx, const i64* rs,
                    i64 rows, float* out) {
    if (rows <= 0) return ASR_HIP_OK;
    k_reduce_rows<<<grid_for(rows, 256), 256, 0, ctx->stream>>>(values, gidx, rs, rows, out);
    ASR_CHECK_LAUNCH(ctx);
    return ASR_HIP_OK;
}

int asr_conv_decode(asr_hip_context* ctx, const float* code, i64 v, int c, const float* w1,
                    const float* b1, int h1, const float* w2, const float* b2, int h2,
                    const float* w3, const float* sizes, float* out, const int32_t* rows) {
    if (v <= 0) return ASR_HIP_OK;
    if (c > DEC_MAX || h1 > DEC_MAX || h2 > DEC_MAX || c < 1 || h1 < 1 || h2 < 1)
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "decode_mlp: layer widths must be 1..64");
    size_t lds = sizeof(float) * (size_t)(h1 * c + h1 + h2 * h1 + h2 + 2 * h2);
    if (ctx->dry_launch) return ASR_HIP_OK;
    if (c == 32 && h1 == 32 && h2 == 32 && ((uintptr_t)code % 16 == 0))
        k_decode_mfma<<<(unsigned)std::min<i64>((v + 63) / 64, 2048), 256, 0, ctx->stream>>>(code, v, w1, b1, w2, b2, w3, sizes, out, rows);
    else
        k_decode<0, 0, 0><<<grid_for(v, 256), 256, lds, ctx->stream>>>(code, v, c, w1, b1, h1, w2, b2, h2, w3,
                                                                       sizes, out, rows);
    ASR_CHECK_LAUNCH(ctx);
    return ASR_HIP_OK;
}

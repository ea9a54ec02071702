// asr_conv16.hip -- a12 on the 16-bit matrix cores of gfx950 (v_mfma_f32_16x16x32_{f16,bf16}, 16x the rate
// of the f32-input MFMA), two arithmetic modes behind one gather-GEMM kernel:
//
//   ASR_CONV16_F16     config C5 of BASELINE.json: activations live in HBM as f16 (half the gather bytes that
//                      bound the level-0 layers), weights are rounded to f16 once, products accumulate in f32.
//   ASR_CONV16_BF16X3  fp32-class results from bf16 MFMAs: every f32 operand is split EXACTLY into three bf16
//                      terms (a = a0 + a1 + a2: a0 = rn(a), a1 = rn(a - a0), a2 = the rest; 8 mantissa bits
//                      each), and a*b is evaluated as the six products a_i*b_j with i + j <= 2, accumulated in
//                      f32.  The dropped terms are below 2^-24 |a*b| and signed, i.e. less than the error of
//                      one fp32 rounding.  6 MFMAs at 16x the f32-MFMA rate = 2.7x the f32 matrix peak for
//                      the same algorithmic FLOP.
//
//   ASR_CONV16_F16X2   fp32-class results from HALF the MFMAs of bf16x3: both operands are scaled by a power of two
//                      (per tensor: the largest magnitude goes to [2^14, 2^15), exact) and split into two f16
//                      terms (a = a0 + a1, 11 mantissa bits each, round to nearest), a*b = a0*b0 + a1*b0 + a0*b1.
//                      What is dropped (a1*b1 and the rounding of the two low terms) stays below 2^-21 |a*b| for
//                      every element within 2^-17 of its tensor's maximum and below 2^-38 max|a| max|b| for the
//                      rest -- the f16 range never matters because of the scaling.  The scale of the activations
//                      comes from a device-side running maximum that the PRODUCING kernel's epilogue maintains
//                      (asr_sparse_conv_args.out_absmax -> the consumer's inp_absmax); nothing is read back.
//
// Weights are re-packed once per weight tensor (asr_conv16_pack): [plane][slot k][column][cin] with cin
// contiguous and padded to 32, so that a B fragment (8 consecutive k of one column) is one 16-byte piece in
// HBM, in the LDS panel and in the register.  Same tiling / slot-skipping / two-filter-bank scheme as
// k_sconv_mfma (asr_conv.hip); reference semantics: models/common_torch.py:95-148.
#include <type_traits>

#include "asr_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint16_t u16;

constexpr int NBR_LD = 57;

__device__ inline u16 f32_to_bf16_bits(float x) {  // round to nearest even
    const f32x2 v = {x, 0.f};
    return (u16)(__builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)) & 0xffffu);
}

__device__ inline u16 f32_to_f16_bits(float x) {
    _Float16 h = (_Float16)x;  // round to nearest even
    return __builtin_bit_cast(u16, h);
}

// ------------------------------------------------------------------------------------------
// weight packing: filters [K, cin, cout] (+ bank b [K, cin, cout_b], appended as columns) ->
// packed [planes][K][ctot_pad][cin_pad] 16-bit, zero padded
// ------------------------------------------------------------------------------------------
template <int KC>
__device__ inline int swz(int col, int slot) {  // 16-byte piece index inside a panel row, swizzled
    if (KC == 32) return slot ^ ((0x6C >> (2 * ((col >> 2) & 3))) & 3);
    return slot ^ ((col >> 1) & 7);
}

// kc = panel depth the kernels will use for this tensor (asr_conv16_panel_depth); element (k, column, c) of
// plane pl goes to  pl * plane + ((k * npanel + c / kc) * ctot_pad + column) * kc + swizzled(c % kc)
// Exponent s of the power-of-two scale 2^s that takes a tensor with the given largest magnitude (f32 bits) to
// [2^14, 2^15): the rounded-up f16 of the largest element stays finite.  0, denormal, inf and nan: 0.  |s| <= 100, so
// that 2^s, 2^-s and the halves of the sum of two such exponents are normal f32 numbers (tensors below 2^-86 or above
// 2^114 lose accuracy).
__device__ inline int f16x2_scale_exp(unsigned absmax_bits) {
    const int e = (int)((absmax_bits >> 23) & 0xffu);
    if (e == 0 || e == 255) return 0;
    const int s = 14 - (e - 127);
    return s < -100 ? -100 : (s > 100 ? 100 : s);
}
__device__ inline float f16x2_pow2(int s) { return __uint_as_float((unsigned)(127 + s) << 23); }  // |s| <= 126

// largest |x| of a strided matrix as f32 bits (non-negative floats order like their bit patterns)
__global__ void k_absmax(const float* __restrict__ x, i64 rows, int c, i64 ld, unsigned* __restrict__ out) {
    unsigned m = 0;
    const i64 total = rows * c;
    for (i64 e = blockIdx.x * (i64)blockDim.x + threadIdx.x; e < total; e += (i64)gridDim.x * blockDim.x)
        m = max(m, __float_as_uint(x[(e / c) * ld + e % c]) & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, m);
}

__global__ void k_pack_filters(const float* __restrict__ wa, const float* __restrict__ wb, int K, int cin, int ca,
                               int cb, int cin_pad, int ctot_pad, int mode, int kc, u16* __restrict__ out,
                               const unsigned* __restrict__ w_absmax) {
    const i64 total = (i64)K * ctot_pad * cin_pad;
    const int npanel = cin_pad / kc;
    float wscale = 1.f;
    if (mode == ASR_CONV16_F16X2) {  // trailer: what the conv kernel multiplies its sums with
        const int sw = f16x2_scale_exp(*w_absmax);
        wscale = f16x2_pow2(sw);
        if (blockIdx.x == 0 && threadIdx.x == 0) *(int*)(out + 2 * total) = sw;
    }
    for (i64 e = blockIdx.x * (i64)blockDim.x + threadIdx.x; e < total; e += (i64)gridDim.x * blockDim.x) {
        const int c = (int)(e % cin_pad);
        const int col = (int)((e / cin_pad) % ctot_pad);
        const int k = (int)(e / ((i64)cin_pad * ctot_pad));
        float w = 0.f;
        if (c < cin) {
            if (col < ca)
                w = wa[((i64)k * cin + c) * ca + col];
            else if (col < ca + cb)
                w = wb[((i64)k * cin + c) * cb + (col - ca)];
        }
        const int pn = c / kc, cc = c % kc;
        const int piece = kc == 32 ? swz<32>(col & 127, cc >> 3) : swz<64>(col & 127, cc >> 3);
        const i64 o = (((i64)k * npanel + pn) * ctot_pad + col) * kc + piece * 8 + (cc & 7);
        if (mode == ASR_CONV16_F16) {
            out[o] = f32_to_f16_bits(w);
        } else if (mode == ASR_CONV16_F16X2) {
            const float ws = w * wscale;
            const u16 h = f32_to_f16_bits(ws);
            out[o] = h;
            out[total + o] = f32_to_f16_bits(ws - (float)__builtin_bit_cast(_Float16, h));
        } else {  // exact three-way bf16 split, round to nearest (w = b0 + b1 + b2)
            const u16 b0 = f32_to_bf16_bits(w);
            const float r1 = w - __uint_as_float((unsigned)b0 << 16);
            const u16 b1 = f32_to_bf16_bits(r1);
            const float r2 = r1 - __uint_as_float((unsigned)b1 << 16);
            out[o] = b0;
            out[total + o] = b1;
            out[2 * total + o] = f32_to_bf16_bits(r2);
        }
    }
}

__global__ void k_f32_to_f16(const float* __restrict__ in, i64 n, u16* __restrict__ out) {
    i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (i < n) out[i] = f32_to_f16_bits(in[i]);
}
__global__ void k_f16_to_f32(const u16* __restrict__ in, i64 n, float* __restrict__ out) {
    i64 i = blockIdx.x * (i64)blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)__builtin_bit_cast(_Float16, in[i]);
}

// f16x2: the 8 gathered f32 of a lane (two 16-byte pieces), scaled, as two f16 fragments hi = rn(x), lo = rn(x - hi)
// (v_cvt_pk_f16_f32: two values per instruction, round to nearest even)
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void sconv16_split_f16x2(const u32x4& q0, const u32x4& q1, float scale, u32x4& hi, u32x4& lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const u32x4& q = i < 2 ? q0 : q1;
        const f32x2 v = {__uint_as_float(q[2 * (i & 1)]) * scale, __uint_as_float(q[2 * (i & 1) + 1]) * scale};
        const f16x2 hv = __builtin_convertvector(v, f16x2);
        const f32x2 r = {v.x - (float)hv.x, v.y - (float)hv.y};
        h[i] = __builtin_bit_cast(unsigned, hv);
        l[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
    }
    hi = (u32x4){h[0], h[1], h[2], h[3]};
    lo = (u32x4){l[0], l[1], l[2], l[3]};
}

// ------------------------------------------------------------------------------------------
// Products of one step, shared by the two kernels below: fa = the wave's A fragments of the step (f16: one per 32
// k; bf16x3: the three exact bf16 terms of the gathered f32), sb = the step's weight panel in LDS.  bf16x3 evaluates
// a*b as the six products a_i*b_j with i + j <= 2, one weight plane at a time (a B fragment lives for at most three
// MFMAs) and the small terms first.  IMP: everything goes to the per-slot accumulators tacc; DUAL: the products of the last
// column tile (bank a's last eight columns and bank b's eight) go to tacc[0] ONLY; at the end of the slot it is added to
// acc[NT - 1] as it is (bank a) and, scaled per row by the importance, to acc_b (bank b) -- not a second chain of products.
// ------------------------------------------------------------------------------------------
// -DASR_BF16X3_CHAIN=4 (build-time, off): bf16x3 with a SECOND accumulator for the five small products of a step, added once in
// the epilogue.  The default chain rounds the large running sum six times per (slot, panel) step, which is where bf16x3's
// error comes from (scripts/split_error_study.py: rms error of `values` 2.1e-7 of the range -> 6.3e-8, below the
// exact-f32 kernel's 1.9e-7); it costs 32 registers on the 128-column instances = two blocks per CU instead of three,
// U-Net 26.4 -> 27.5 ms at 10 M points, so the default stays the single chain.
#ifndef ASR_BF16X3_CHAIN
#define ASR_BF16X3_CHAIN 0
#endif
template <int NT, int KC, int MODE, bool IMP, bool DUAL, int PLANES, int NJ>
__device__ inline void sconv16_products(const u32x4 (&fa)[NJ][PLANES], const u32x4* __restrict__ sb, f32x4 (&acc)[NT],
                                        f32x4 (&tacc)[IMP ? NT : 1], bool has_b, int ncol, int g, f32x4 (&lo)[NT]) {
    constexpr int SLOTS = KC / 8;
    constexpr int PLANE_PIECES = NT * 16 * SLOTS;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        if constexpr (MODE == ASR_CONV16_F16) {
            const f16x8 af = __builtin_bit_cast(f16x8, fa[j][0]);
#pragma unroll
            for (int nb = 0; nb < NT; ++nb) {
                const int col = nb * 16 + ncol;
                const f16x8 bf = __builtin_bit_cast(f16x8, sb[col * SLOTS + swz<KC>(col, 4 * j + g)]);
                if (IMP) {
                    tacc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, tacc[nb], 0, 0, 0);
                } else if (DUAL && has_b && nb == NT - 1) {
                    tacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, tacc[0], 0, 0, 0);  // (see the head comment)
                } else {
                    acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, acc[nb], 0, 0, 0);
                }
            }
        } else if constexpr (MODE == ASR_CONV16_F16X2) {
            const f16x8 a0 = __builtin_bit_cast(f16x8, fa[j][0]);
            const f16x8 a1 = __builtin_bit_cast(f16x8, fa[j][PLANES > 1 ? 1 : 0]);
#pragma unroll
            for (int nb = 0; nb < NT; ++nb) {
                const int col = nb * 16 + ncol;
                const int piece = col * SLOTS + swz<KC>(col, 4 * j + g);
#define ASR_THREE(ACC_)                                                                                 \
    {                                                                                                   \
        const f16x8 b1 = __builtin_bit_cast(f16x8, sb[PLANE_PIECES + piece]);                           \
        ACC_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, ACC_, 0, 0, 0);                           \
        const f16x8 b0 = __builtin_bit_cast(f16x8, sb[piece]);                                          \
        ACC_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, ACC_, 0, 0, 0);                           \
        ACC_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, ACC_, 0, 0, 0);                           \
    }
                if (IMP) {
                    ASR_THREE(tacc[nb])
                } else if (DUAL && has_b && nb == NT - 1) {
                    ASR_THREE(tacc[0])
                } else {
                    ASR_THREE(acc[nb])
                }
#undef ASR_THREE
            }
        } else {
            const bf16x8 a0 = __builtin_bit_cast(bf16x8, fa[j][0]);
            const bf16x8 a1 = __builtin_bit_cast(bf16x8, fa[j][PLANES > 1 ? 1 : 0]);
            const bf16x8 a2 = __builtin_bit_cast(bf16x8, fa[j][PLANES > 2 ? 2 : 0]);
#pragma unroll
            for (int nb = 0; nb < NT; ++nb) {
                const int col = nb * 16 + ncol;
                const int piece = col * SLOTS + swz<KC>(col, 4 * j + g);
#define ASR_SIX(ACC_)                                                                                   \
    {                                                                                                   \
        const bf16x8 b2 = __builtin_bit_cast(bf16x8, sb[2 * PLANE_PIECES + piece]);                     \
        ACC_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b2, ACC_, 0, 0, 0);                          \
        const bf16x8 b1 = __builtin_bit_cast(bf16x8, sb[PLANE_PIECES + piece]);                         \
        ACC_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, ACC_, 0, 0, 0);                          \
        ACC_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b1, ACC_, 0, 0, 0);                          \
        const bf16x8 b0 = __builtin_bit_cast(bf16x8, sb[piece]);                                        \
        ACC_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b0, ACC_, 0, 0, 0);                          \
        ACC_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b0, ACC_, 0, 0, 0);                          \
        ACC_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, ACC_, 0, 0, 0);                          \
    }
                if (IMP) {
                    ASR_SIX(tacc[nb])
                } else {
#if ASR_BF16X3_CHAIN == 4
                    if (DUAL && has_b && nb == NT - 1) {
                        ASR_SIX(tacc[0])
                    } else
                    // the five small products (2^-8 and 2^-16 of the leading one) go to a second accumulator that is added
                    // ONCE, in the epilogue: one fp32 rounding at the running sum's magnitude per step instead of six
                    {
                        const bf16x8 b2 = __builtin_bit_cast(bf16x8, sb[2 * PLANE_PIECES + piece]);
                        lo[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b2, lo[nb], 0, 0, 0);
                        const bf16x8 b1 = __builtin_bit_cast(bf16x8, sb[PLANE_PIECES + piece]);
                        lo[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, lo[nb], 0, 0, 0);
                        lo[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b1, lo[nb], 0, 0, 0);
                        const bf16x8 b0 = __builtin_bit_cast(bf16x8, sb[piece]);
                        lo[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b0, lo[nb], 0, 0, 0);
                        lo[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b0, lo[nb], 0, 0, 0);
                        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, acc[nb], 0, 0, 0);
                    }
#else
                    if (DUAL && has_b && nb == NT - 1) {
                        ASR_SIX(tacc[0])
                    } else {
                        ASR_SIX(acc[nb])
                    }
#endif
                }
#undef ASR_SIX
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Epilogue shared by the two kernels below: acc[nb][i] is C[row = 4 g + i][col = ncol] of the wave's 16 x 16 blocks.
// rows[i] / norms[i]: output row (-1: none) and importance sum of accumulator row i; bank b (DUAL) sits in the upper
// half of the last column tile (acc_b) and is the only part that is normalised there.  Normalise, bias, ReLU, residual
// (activation type), strided store as f32 or f16.
// ------------------------------------------------------------------------------------------
template <int NT, int MODE, bool DUAL>
__device__ inline void sconv16_epilogue(const asr_sparse_conv_args& a, const f32x4 (&acc)[NT], const f32x4& acc_b,
                                        const int (&rows)[4], const float (&norms)[4], int n0, int ncol, int ca, int cout,
                                        bool has_b, int out_f16, const float* __restrict__ zeros, const float (&unscale)[2]) {
    float bv[NT];
    unsigned amax = 0;  // largest |output| of this lane (f32 bits), for the next layer's f16x2 scale
#pragma unroll
    for (int nb = 0; nb < NT; ++nb) {
        const int col = n0 + nb * 16 + ncol;
        const float* bp = (a.bias && col < ca) ? a.bias + col : zeros;
        if (DUAL && a.bias_b && col >= ca && col < cout) bp = a.bias_b + (col - ca);
        bv[nb] = *bp;
    }
    constexpr bool res16 = MODE == ASR_CONV16_F16;  // the residual has the activations' type
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const i64 q = rows[i];
        const bool rowok = q >= 0;
        const float norm = norms[i];
        const bool do_norm = a.normalize && norm != 0.f;
        float res[NT];
#pragma unroll
        for (int nb = 0; nb < NT; ++nb) {
            const int col = n0 + nb * 16 + ncol;
            const bool ok = a.residual && rowok && col < cout;
            if (res16) {
                const u16* rp = ok ? (const u16*)a.residual + q * a.residual_ld + col : (const u16*)zeros;
                res[nb] = (float)__builtin_bit_cast(_Float16, *rp);
            } else {
                const float* rp = ok ? a.residual + q * a.residual_ld + col : zeros;
                res[nb] = *rp;
            }
        }
#pragma unroll
        for (int nb = 0; nb < NT; ++nb) {
            const int col = n0 + nb * 16 + ncol;
            float v = acc[nb][i];
            if (DUAL) {
                const bool colb = col >= ca;
                if (nb == NT - 1 && has_b && colb) v = acc_b[i];
                if (MODE == ASR_CONV16_F16X2) v = v * unscale[0] * unscale[1];  // powers of two: exact
                v = (colb && do_norm) ? v / norm : v;
            } else {
                if (MODE == ASR_CONV16_F16X2) v = v * unscale[0] * unscale[1];
                v = do_norm ? v / norm : v;
            }
            v += bv[nb];
            if (a.relu) v = fmaxf(v, 0.f);
            v += res[nb];
            if (rowok && col < cout) {
                if (out_f16)
                    ((u16*)a.out)[q * a.out_ld + col] = f32_to_f16_bits(v);
                else
                    a.out[q * a.out_ld + col] = v;
                amax = max(amax, __float_as_uint(v) & 0x7fffffffu);
            }
        }
    }
    if (a.out_absmax) {
        // One candidate per wave; the value only grows, so a wave whose candidate is not above what it reads (possibly
        // stale, i.e. smaller) has nothing to add -- after the first waves hardly any atomic reaches the L2.
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = max(amax, (unsigned)__shfl_xor((int)amax, o, 64));
        if ((threadIdx.x & 63) == 0 && amax > __hip_atomic_load(a.out_absmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(a.out_absmax, amax);
    }
}

// ------------------------------------------------------------------------------------------
// the kernel.  Block = WAVES waves, tile = WAVES*16 output rows x NT*16 columns; per (slot, KC-deep cin panel)
// step the weight panel [planes][NCOL][KC] goes global -> registers -> LDS (XOR-swizzled 16-byte pieces,
// conflict-free ds_read_b128), double buffered, with the same two-step register prefetch as k_sconv_mfma.
// Lane (r = l & 15, g = l >> 4) gathers the 8 consecutive cin values c0 + 32 j + 8 g .. + 7 of its row's
// neighbour: 16 bytes of f16, or 32 bytes of f32 that are split into three bf16 fragments in registers.
// Importance (conv1b) is applied per (row, slot) on the accumulator side: the slot's products go to a
// temporary accumulator that is scaled and added when the slot is finished -- exact f32 scaling in both
// modes, no second A operand.
// ------------------------------------------------------------------------------------------
template <int NT, int KC, int WAVES, int MODE, bool IMP, bool DUAL>
__global__ __launch_bounds__(WAVES * 64, WAVES == 8 ? 2 : 3) void k_sconv_mfma16(
        asr_sparse_conv_args a, const u16* __restrict__ packed, int cin_pad, int ctot_pad, int out_f16,
        const float* __restrict__ zeros) {
    constexpr int TM = WAVES * 16;
    constexpr int NTHR = WAVES * 64;
    constexpr int NCOL = NT * 16;
    constexpr int PLANES = MODE == ASR_CONV16_BF16X3 ? 3 : (MODE == ASR_CONV16_F16X2 ? 2 : 1);
    constexpr int SLOTS = KC / 8;                 // 16-byte pieces per panel row
    constexpr int NJ = KC / 32;                   // MFMA k-chunks per panel
    constexpr int PV = PLANES * NCOL * SLOTS;     // 16-byte pieces per panel
    constexpr int SV = (PV + NTHR - 1) / NTHR;
    constexpr bool SV_EXACT = PV % NTHR == 0;
    constexpr int PLANE_PIECES = NCOL * SLOTS;
    constexpr int AW = MODE == ASR_CONV16_F16 ? 1 : 2;  // 16-byte loads per lane per k-chunk
    __shared__ int s_nbr[TM * NBR_LD];
    // per (row, slot) importance: only the single-bank IMP form keeps it in LDS.  The two-bank form reads
    // inp_importance[neighbour] from global memory at the head of a slot's last step (80 KB of LDS per 8-wave
    // block = two blocks per CU; with the table it was 108 KB and one block)
    __shared__ float s_w[IMP ? TM * NBR_LD : 1];
    __shared__ float s_norm[TM];
    __shared__ int s_row[TM];
    __shared__ unsigned long long s_mask[TM];
    __shared__ __attribute__((aligned(16))) u32x4 s_B[2][PV];

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int nY = ctot_pad / NCOL;  // the launcher picks NT so that NCOL divides the padded width
    i64 tile = blockIdx.x;
    int ychunk = 0;
    if (nY > 1) {  // column chunks of one row tile on one XCD, consecutive dispatch slots (see k_sconv_mfma)
        const i64 r8 = blockIdx.x >> 3;
        ychunk = (int)(r8 % nY);
        tile = (r8 / nY) * 8 + (blockIdx.x & 7);
    }
    const i64 row0 = tile * TM;
    if (row0 >= a.num_out) return;
    const int n0 = ychunk * NCOL;
    const int K = a.kernel_size;
    const int cin = a.cin;
    const int ca = a.cout;
    const int cout = a.cout + (DUAL ? a.cout_b : 0);
    const bool has_b = DUAL && ychunk == nY - 1;

    for (int i = tid; i < TM * NBR_LD; i += NTHR) {
        s_nbr[i] = -1;
        if (IMP) s_w[i] = 0.f;
    }
    __syncthreads();
    {
        constexpr int TPR = NTHR / TM;
        const int prow = tid / TPR, pj = tid % TPR;
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
                if (k >= K) continue;
                s_nbr[prow * NBR_LD + k] = i;
                m |= 1ull << k;
                float w = 1.f;
                if (IMP || DUAL) {
                    w = a.neighbors_importance ? a.neighbors_importance[p] : a.inp_importance[i];
                    if (IMP) s_w[prow * NBR_LD + k] = w;
                }
                norm += w;
            }
        } else {
            q = -1;
        }
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) {
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
    unsigned long long wmask = s_mask[lrow];
    unsigned long long bmask = s_mask[lane];
    if (TM > 64) bmask |= s_mask[64 + lane];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bmask |= __shfl_xor(bmask, o, 64);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) wmask |= __shfl_xor(wmask, o, 64);
    wmask = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)wmask) |
            ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(wmask >> 32)) << 32);
    bmask = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)bmask) |
            ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(bmask >> 32)) << 32);
    bmask &= (1ull << K) - 1;

    f32x4 acc[NT];
    f32x4 lo[NT];  // (ASR_BF16X3_CHAIN == 4: the small products' accumulator; unused and removed otherwise)
#pragma unroll
    for (int t = 0; t < NT; ++t) lo[t] = {0.f, 0.f, 0.f, 0.f};
    f32x4 tacc[IMP ? NT : 1];  // per-slot accumulators of the importance-weighted bank
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < (IMP ? NT : 1); ++t) tacc[t] = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc_b = {0.f, 0.f, 0.f, 0.f};  // DUAL: bank b of the last column tile

    const int npanel = (cin + KC - 1) / KC;
    const i64 plane_stride = (i64)K * ctot_pad * cin_pad;  // elements per weight plane
    // f16x2: activations * a_scale before the split, sums * 2^-(sa + sw) after (in two factors: the sum can reach 200)
    float a_scale = 1.f, unscale[2] = {1.f, 1.f};
    if constexpr (MODE == ASR_CONV16_F16X2) {
        const int sa = f16x2_scale_exp(*a.inp_absmax);
        const int un = -(sa + *(const int*)(packed + PLANES * plane_stride));
        a_scale = f16x2_pow2(sa);
        unscale[0] = f16x2_pow2(un / 2);
        unscale[1] = f16x2_pow2(un - un / 2);
    }

#define ASR_SEQ_ADVANCE(todo, k, p)                          \
    if ((k) >= 0 && ++(p) == npanel) {                       \
        (p) = 0;                                             \
        (todo) &= (todo)-1;                                  \
        (k) = (todo) ? __builtin_ctzll(todo) : -1;           \
    }
    u32x4 stage0[SV], stage1[SV];
    u32x4 a_q0[NJ * AW], a_q1[NJ * AW];

    // Both operand streams go through raw buffer loads: the address of a piece is a per-thread byte offset
    // that does not change from step to step (VGPR), plus a wave-uniform offset of the step (SGPR), plus an
    // immediate -- no per-step vector address arithmetic next to the MFMAs.  Offsets beyond the buffer read as
    // zero, which is how absent neighbours (OOB_OFF) and unused staging pieces load their zeros.
    constexpr unsigned OOB_OFF = 0xFFFFE000u;
    constexpr int RSRC_FLAGS = 0x00020000;
    const __amdgpu_buffer_rsrc_t rs_w =
            __builtin_amdgcn_make_buffer_rsrc((void*)packed, 0, (int)(PLANES * plane_stride * 2), RSRC_FLAGS);
    unsigned w_off[SV];  // byte offset of this thread's pieces inside a (slot 0, panel 0) weight panel
    int w_lds[SV];       // their swizzled position in the LDS panel
#pragma unroll
    for (int s = 0; s < SV; ++s) {
        const int e = tid + s * NTHR;  // piece index: (plane, column, slot)
        const int pl = e / PLANE_PIECES;
        const int rem = e % PLANE_PIECES;
        const int col = rem / SLOTS, slot = rem % SLOTS;
        const bool ok = SV_EXACT || e < PV;
        // packed panels are stored in LDS order (swizzle included): memory piece = LDS piece
        w_off[s] = ok ? (unsigned)((pl * plane_stride + (i64)(n0 + col) * KC + 8 * slot) * 2) : OOB_OFF;
        w_lds[s] = ok ? e : -1;
    }
    auto load_panel = [&](const int qk, const int qp, u32x4 (&st)[SV]) __attribute__((always_inline)) {
        const int soff = (((qk < 0 ? 0 : qk) * npanel + qp) * ctot_pad * KC) * 2;
#pragma unroll
        for (int s = 0; s < SV; ++s) st[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)w_off[s], soff, 0);
    };
    auto store_panel = [&](int buf, const u32x4 (&st)[SV]) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < SV; ++s)
            if (SV_EXACT || w_lds[s] >= 0) s_B[buf][w_lds[s]] = st[s];
    };
    constexpr int ESZ = MODE == ASR_CONV16_F16 ? 2 : 4;  // bytes per activation element
    // feature rows: buffer addressing when the matrix spans less than 4 GB, 64-bit pointers otherwise
    const i64 a_span = a.num_inp > 0 ? ((a.num_inp - 1) * a.inp_ld + cin) * ESZ : 0;
    const bool a_big = a_span > (i64)(OOB_OFF - 4096);
    const __amdgpu_buffer_rsrc_t rs_a =
            __builtin_amdgcn_make_buffer_rsrc((void*)a.inp_features, 0, a_big ? 0 : (int)(unsigned)a_span, RSRC_FLAGS);
    const bool cin_tail = cin % KC != 0;  // the last panel reads pieces beyond the row: they must be zero
    int cache_k = -2;
    unsigned cache_off = OOB_OFF;                   // byte offset of the lane's neighbour row + its 8 g columns
    const char* cache_row = (const char*)zeros;     // a_big only
    bool cache_valid = false;
    auto gather_a = [&](const int qk, const int qp, u32x4 (&aq)[NJ * AW]) __attribute__((always_inline)) {
        if (qk != cache_k) {
            cache_k = qk;
            const int idx = qk < 0 ? -1 : s_nbr[lrow * NBR_LD + qk];
            cache_valid = idx >= 0;
            cache_off = cache_valid ? (unsigned)idx * (unsigned)(a.inp_ld * ESZ) + (unsigned)(8 * g * ESZ) : OOB_OFF;
            if (a_big) cache_row = (const char*)a.inp_features + (i64)(cache_valid ? idx : 0) * a.inp_ld * ESZ;
        }
        if (!a_big && !cin_tail) {
            const int soff = qp * KC * ESZ;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int h = 0; h < AW; ++h)
                    aq[j * AW + h] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, (int)cache_off + (32 * j + 4 * h) * ESZ,
                                                                           soff, 0);
            return;
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = qp * KC + 32 * j + 8 * g;
#pragma unroll
            for (int h = 0; h < AW; ++h) {
                // cin % 8 == 0 (f16) / cin % 4 == 0 (f32): a 16-byte piece is inside or outside the row
                const int cc = c + 4 * h;
                if (a_big) {
                    const void* src = (cache_valid && cc < cin) ? (const void*)(cache_row + (i64)cc * ESZ) : (const void*)zeros;
                    aq[j * AW + h] = *reinterpret_cast<const u32x4*>(src);
                } else {
                    const unsigned vo = cc < cin ? cache_off + (unsigned)((32 * j + 4 * h) * ESZ) : OOB_OFF;
                    aq[j * AW + h] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, (int)vo, qp * KC * ESZ, 0);
                }
            }
        }
    };

    unsigned long long todo1 = bmask;
    int k_cur = bmask ? __builtin_ctzll(bmask) : -1, p_cur = 0;
    int k1 = k_cur, p1 = 0;
    ASR_SEQ_ADVANCE(todo1, k1, p1)
    if (k_cur >= 0) {
        load_panel(k_cur, p_cur, stage0);
        gather_a(k_cur, p_cur, a_q0);
        store_panel(0, stage0);
        load_panel(k1, p1, stage1);
        gather_a(k1, p1, a_q1);
    }
    __syncthreads();
    int buf = 0;
    const int ncol = lane & 15;
    auto step = [&](u32x4 (&aq)[NJ * AW], u32x4 (&st_free)[SV], u32x4 (&st_next)[SV]) __attribute__((always_inline)) {
        int k2 = k1, p2 = p1;
        ASR_SEQ_ADVANCE(todo1, k2, p2)
        store_panel(buf ^ 1, st_next);
        load_panel(k2, p2, st_free);
        const bool active = (wmask >> k_cur) & 1;
        float w4[4] = {0.f, 0.f, 0.f, 0.f};  // importance of this lane's four accumulator rows for slot k_cur
        if (DUAL && has_b && p_cur == npanel - 1 && active) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {  // issued here, consumed after the MFMAs of this step
                const int idx = s_nbr[(wave * 16 + 4 * g + i) * NBR_LD + k_cur];
                w4[i] = *(idx >= 0 ? a.inp_importance + idx : zeros);
            }
        }
        // The A fragments of this step are taken out of aq BEFORE the gather of step + 2 is issued into the
        // same registers (no register rotation, the gather stays two steps ahead).
        u32x4 fa[NJ][PLANES];
        {  // (unconditional: a wave without the slot splits the zeros its gathers returned -- no branch, no merge copies)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if constexpr (MODE == ASR_CONV16_F16) {
                    fa[j][0] = aq[j];
                } else if constexpr (MODE == ASR_CONV16_F16X2) {
                    sconv16_split_f16x2(aq[j * 2], aq[j * 2 + 1], a_scale, fa[j][0], fa[j][PLANES > 1 ? 1 : 0]);
                } else {
                    // exact split of the 8 gathered f32 into three bf16 fragments: a0 = rn(a), a1 = rn(a - a0),
                    // a2 = a - a0 - a1 (exactly representable: 24 = 8 + 8 + 8 mantissa bits).  Round to nearest
                    // (v_cvt_pk_bf16_f32, two values per instruction) keeps the residuals signed, so the dropped
                    // product terms do not add up to a bias.
                    unsigned p0[4], p1[4], p2[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const f32x2 v = {__uint_as_float(aq[j * 2 + (i >> 1)][2 * (i & 1)]),
                                         __uint_as_float(aq[j * 2 + (i >> 1)][2 * (i & 1) + 1])};
                        p0[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
                        const f32x2 r1 = {v.x - __uint_as_float(p0[i] << 16), v.y - __uint_as_float(p0[i] & 0xffff0000u)};
                        p1[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
                        const f32x2 r2 = {r1.x - __uint_as_float(p1[i] << 16), r1.y - __uint_as_float(p1[i] & 0xffff0000u)};
                        p2[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
                    }
                    fa[j][0] = (u32x4){p0[0], p0[1], p0[2], p0[3]};
                    fa[j][PLANES > 1 ? 1 : 0] = (u32x4){p1[0], p1[1], p1[2], p1[3]};
                    fa[j][PLANES > 2 ? 2 : 0] = (u32x4){p2[0], p2[1], p2[2], p2[3]};
                }
            }
        }
        gather_a(k2, p2, aq);
        if (active) {
            const u32x4* sb = s_B[buf];
            __builtin_amdgcn_s_setprio(1);
            sconv16_products<NT, KC, MODE, IMP, DUAL, PLANES, NJ>(fa, sb, acc, tacc, has_b, ncol, g, lo);
            __builtin_amdgcn_s_setprio(0);
            // end of a slot: scale the slot's importance-weighted products per row and fold them in
            if ((IMP || DUAL) && p_cur == npanel - 1) {
                if (IMP) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) w4[i] = s_w[(wave * 16 + 4 * g + i) * NBR_LD + k_cur];
#pragma unroll
                    for (int nb = 0; nb < NT; ++nb) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[nb][i] += w4[i] * tacc[nb][i];
                        tacc[nb] = {0.f, 0.f, 0.f, 0.f};
                    }
                } else if (has_b) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc_b[i] += w4[i] * tacc[0][i];
                    acc[NT - 1] += tacc[0];  // bank a's columns of the shared tile: the slot's plain sums
                    tacc[0] = {0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        __syncthreads();
        buf ^= 1;
        k_cur = k1;
        p_cur = p1;
        k1 = k2;
        p1 = p2;
    };
    while (k_cur >= 0) {
        step(a_q0, stage0, stage1);
        if (k_cur < 0) break;
        step(a_q1, stage1, stage0);
    }
#undef ASR_SEQ_ADVANCE

    int rows4[4];
    float norms4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        rows4[i] = s_row[wave * 16 + 4 * g + i];
        norms4[i] = s_norm[wave * 16 + 4 * g + i];
    }
#if ASR_BF16X3_CHAIN == 4
#pragma unroll
    for (int nb = 0; nb < NT; ++nb) acc[nb] += lo[nb];
#endif
    sconv16_epilogue<NT, MODE, DUAL>(a, acc, acc_b, rows4, norms4, n0, ncol, ca, cout, has_b, out_f16, zeros, unscale);
    if (a.out_importance && ychunk == 0 && tid < TM && s_row[tid] >= 0)
        a.out_importance[s_row[tid]] = s_norm[tid];
}


// ------------------------------------------------------------------------------------------
// Row-group plan of a neighbour list (asr_conv16_plan_*): the CSR is re-laid once per list into the order the
// kernel below streams it.  16 consecutive rows (in row_perm order) form a group; its header holds the set of
// kernel slots any of its rows uses and the position of its first block in the pool; the pool holds, for every
// slot of the set in ascending order, the 16 neighbour indices (-1: this row has no neighbour in the slot).
// Row regrouping makes the groups nearly homogeneous, so the pool is about as large as the CSR itself.
// ------------------------------------------------------------------------------------------

// ------------------------------------------------------------------------------------------
// Slot-range split of the 55-slot lists (coarse grids).  A tile's slot steps run one after the other inside one block;
// on the coarse grids a launch has fewer tiles than the chip has CUs and ends with its longest tile (up to 31 slots x
// cin / 32 steps by ONE wave per SIMD, whose dependent MFMAs nothing overlaps).  With SPLIT the slots are cut into
// SPLIT_RANGES FIXED ranges -- the 7 same-level slots, then the 48 cross-level slots in four blocks of 12 -- and every
// (tile, range) is a block of its own (blockIdx.y = range).  A tile whose slots all lie in one range is finished by
// that block as usual; otherwise each block stores its partial sums part[range][row position][column] and
// k_sconv_split_reduce adds them IN RANGE ORDER and applies bias / ReLU / residual.  A row's result is
// ((P0 + P1) + P2) + ... over the ranges its tile has, a range the row itself has no slot in contributes exact zeros:
// the result does not depend on which rows share the tile (one rank of a sharded cloud computes the same bits).
// ------------------------------------------------------------------------------------------
constexpr int SPLIT_RANGES = 5;
__host__ __device__ inline unsigned long long split_range_mask(int s) {
    const int lo = s == 0 ? 0 : 7 + 12 * (s - 1), hi = s == 0 ? 7 : 19 + 12 * (s - 1);
    return ((1ull << hi) - 1) ^ ((1ull << lo) - 1);
}
// blockIdx.y -> range: the ranges with the longest chains (twelve cross-level slots) are dispatched first
__device__ inline int split_range_of_y(int y) { return y == 0 ? 1 : (y == 1 ? 2 : (y == 2 ? 0 : y)); }
struct asr_split_args {
    float* part;   // [SPLIT_RANGES][tiles * TM][ctot_pad] f32 partial sums (already unscaled)
    i64 stride;    // elements per range
};

// ------------------------------------------------------------------------------------------
// The plan-driven kernel: same tiles, panels, arithmetic and epilogue as k_sconv_mfma16, but
//   * no neighbour table in LDS and no CSR parsing per block: a wave reads its group header with scalar loads
//     and the 16 indices of a slot with one 64-byte load, prefetched one slot ahead;
//   * weight panels go global -> LDS directly (buffer_load ... lds, 1 KB per wave instruction; the packed
//     tensor is stored in LDS order), no staging registers;
//   * LDS holds only the two panel buffers, so three 8-wave blocks fit a CU.
// Row weights (conv1b: importance of the neighbour) are read per (row, slot) through the plan as well.
// ------------------------------------------------------------------------------------------
#ifndef ASR_PLAN_LINEGATHER
#define ASR_PLAN_LINEGATHER 1
#endif
#ifndef ASR_PLAN_LINEDMA
#define ASR_PLAN_LINEDMA 1
#endif
#ifndef ASR_PLAN_LDMA_DUAL
#define ASR_PLAN_LDMA_DUAL 1
#endif
#ifndef ASR_PLAN_LDMA_F16
#define ASR_PLAN_LDMA_F16 8  // widest plain f16 instance (column tiles) with LDS-DMA gathers; 0: none
#endif
// which instances of k_sconv_plan16 gather whole lines (see gather_a): by LDS DMA (gathers one step ahead) or through registers
// and an LDS transpose
template <int NT, int KC, int WAVES, int MODE, bool IMP, bool DUAL>
constexpr bool plan_ldma() {
    // (f16 activations, config C5: 64-deep panels of 2-byte elements are the same 128 bytes per row and step; the f16 panel
    // buffers leave room for the stage at every width)
    if (MODE == ASR_CONV16_F16)
        return ASR_PLAN_LINEGATHER && ASR_PLAN_LINEDMA && ASR_PLAN_LDMA_F16 && KC == 64 && WAVES == 8 && (IMP || DUAL || NT <= ASR_PLAN_LDMA_F16);
    return ASR_PLAN_LINEGATHER && ASR_PLAN_LINEDMA && KC == 32 && WAVES == 8 &&
           ((!IMP && !DUAL && NT <= 4) || (ASR_PLAN_LDMA_DUAL && (IMP || DUAL)));
}
template <int NT, int KC, int WAVES, int MODE, bool IMP, bool DUAL>
constexpr bool plan_line() {
    return plan_ldma<NT, KC, WAVES, MODE, IMP, DUAL>() ||
           (ASR_PLAN_LINEGATHER && MODE != ASR_CONV16_F16 && KC == 32 && (NT <= 2 || (NT == 4 && WAVES == 8 && (IMP || DUAL))));
}
template <int NT, int KC, int WAVES, int MODE, bool IMP, bool DUAL, bool SPLIT = false>
__global__ __launch_bounds__(WAVES * 64, WAVES == 8 ? (NT >= 8 && (IMP || DUAL) ? 2 : 3) : 4) void k_sconv_plan16(
        asr_sparse_conv_args a, asr_conv_plan_view plan, const u16* __restrict__ packed, int cin_pad, int ctot_pad, int out_f16,
        const float* __restrict__ zeros, asr_split_args sp) {
    constexpr int TM = WAVES * 16;
    constexpr int NCOL = NT * 16;
    constexpr int PLANES = MODE == ASR_CONV16_BF16X3 ? 3 : (MODE == ASR_CONV16_F16X2 ? 2 : 1);
    constexpr int SLOTS = KC / 8;
    constexpr int NJ = KC / 32;
    constexpr int PV = PLANES * NCOL * SLOTS;  // 16-byte pieces per panel
    constexpr int PLANE_PIECES = NCOL * SLOTS;
    constexpr int NCHUNK = PV / 64;            // 1 KB pieces of a panel, one per wave instruction
    constexpr int CHUNKS_PER_PLANE = PLANE_PIECES / 64;
    constexpr int SV = (NCHUNK + WAVES - 1) / WAVES;
    constexpr int AW = MODE == ASR_CONV16_F16 ? 1 : 2;
    constexpr int ESZ = MODE == ASR_CONV16_F16 ? 2 : 4;
    constexpr bool ROWW = IMP || DUAL;  // per (row, slot) weights
    // two separate arrays: the compiler then knows that a panel DMA into one does not feed reads of the other
    __shared__ __attribute__((aligned(16))) u32x4 s_B0[PV];
    __shared__ __attribute__((aligned(16))) u32x4 s_B1[PV];
    // (LDMA: the eight mask words live in the tail of the second panel buffer, which no DMA writes before the barrier that ends
    // the prologue -- 4 blocks x 40 960 bytes are exactly the CU's 160 KB)
    constexpr bool LDMA_ = plan_ldma<NT, KC, WAVES, MODE, IMP, DUAL>();
    __shared__ unsigned long long s_wm_own[LDMA_ ? 1 : WAVES];
    unsigned long long* const s_wm = LDMA_ ? reinterpret_cast<unsigned long long*>(&s_B1[PV - 4]) : s_wm_own;
    // LINE (whole-line gathers, see gather_a): the rows arrive as (row L >> 3, piece L & 7) and leave as operand fragments
    // (row r, pieces 2 g, 2 g + 1) through 2 KB of LDS per wave; piece p of row q sits at column p ^ (q & 7), which makes
    // both the ds_write_b128 and the ds_read_b128 conflict free
    __shared__ __attribute__((aligned(16))) u32x4 s_stage[plan_line<NT, KC, WAVES, MODE, IMP, DUAL>() ? WAVES * 128 : 1];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int nY = ctot_pad / NCOL;
    i64 tile = blockIdx.x;
    int ychunk = 0;
    if (nY > 1) {  // column chunks of one row tile on one XCD, consecutive dispatch slots (see k_sconv_mfma)
        const i64 r8 = blockIdx.x >> 3;
        ychunk = (int)(r8 % nY);
        tile = (r8 / nY) * 8 + (blockIdx.x & 7);
    }
    const i64 row0 = tile * TM;
    if (row0 >= a.num_out) return;
    const int n0 = ychunk * NCOL;
    const int K = a.kernel_size;
    const int cin = a.cin;
    const int ca = a.cout;
    const int cout = a.cout + (DUAL ? a.cout_b : 0);
    const bool has_b = DUAL && ychunk == nY - 1;
    const bool roww = IMP || has_b;
    const int r = lane & 15, g = lane >> 4;
    const int ncol = r;

    // group header: wave-uniform
    const i64 grp = tile * WAVES + wave;
    uint4 h = make_uint4(0, 0, 0, 0);
    if (grp < plan.groups) h = plan.hdr[grp];
    // wmask_all: the group's slots (the pool holds one block per slot, in ascending order); wmask: those this block convolves
    const unsigned long long wmask_all =
            ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)h.x) |
             ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)h.y) << 32)) & ((1ull << K) - 1);
    const unsigned woff = (unsigned)__builtin_amdgcn_readfirstlane((int)h.z);
    if (lane == 0) s_wm[wave] = wmask_all;
    __syncthreads();
    unsigned long long bmask = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) bmask |= s_wm[w];
    bmask = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)bmask) |
            ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(bmask >> 32)) << 32);
    unsigned long long wmask = wmask_all;
    bool partial = false;  // SPLIT: the tile has slots in several ranges, this block stores partial sums
    if constexpr (SPLIT) {
        const unsigned long long rm = split_range_mask(split_range_of_y((int)blockIdx.y));
        if ((bmask & rm) == 0) return;  // (block uniform)
        partial = (bmask & ~rm) != 0;
        wmask &= rm;
        bmask &= rm;
    }

    f32x4 acc[NT];
    f32x4 lo[NT];  // (ASR_BF16X3_CHAIN == 4: the small products' accumulator; unused and removed otherwise)
#pragma unroll
    for (int t = 0; t < NT; ++t) lo[t] = {0.f, 0.f, 0.f, 0.f};
    f32x4 tacc[IMP ? NT : 1];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < (IMP ? NT : 1); ++t) tacc[t] = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc_b = {0.f, 0.f, 0.f, 0.f};
    float norm4[4] = {0.f, 0.f, 0.f, 0.f};

    const int npanel = (cin + KC - 1) / KC;
    const i64 plane_stride = (i64)K * ctot_pad * cin_pad;
    // f16x2: activations * a_scale before the split, sums * 2^-(sa + sw) after (in two factors: the sum can reach 200)
    float a_scale = 1.f, unscale[2] = {1.f, 1.f};
    if constexpr (MODE == ASR_CONV16_F16X2) {
        const int sa = f16x2_scale_exp(*a.inp_absmax);
        const int un = -(sa + *(const int*)(packed + PLANES * plane_stride));
        a_scale = f16x2_pow2(sa);
        unscale[0] = f16x2_pow2(un / 2);
        unscale[1] = f16x2_pow2(un - un / 2);
    }

    // The block walks its slots in ascending order.  Position i of that sequence ("ordinal") is what the loop carries;
    // two per-wave registers map it back: lane i of v_seq holds the byte offset of the weight block of the i-th slot, lane i
    // of v_off the byte offset of this wave's pool block for that slot, or OOB_OFF when the wave has no row in the slot (read
    // with v_readlane -- the 64-bit mask arithmetic this replaces was a third of the loop's scalar instructions, and the
    // scalar unit is what the 64- and 32-column instances of level 0 are short of; U-Net 26.4 -> 25.9 ms.  A third
    // register would cost the 128- and 64-column instances a block per CU: 80 and 64 registers are the limits).
    constexpr unsigned OOB_OFF = 0xFFFFE000u;
    const int nslots = __popcll(bmask);
    const int panel_bytes = ctot_pad * KC * 2, slot_bytes = npanel * panel_bytes;
    int v_seq, v_off;
    {
        const bool mine = lane < K && ((bmask >> lane) & 1);               // lane = slot number
        const int ord = __popcll(bmask & ((1ull << lane) - 1));            // its position in the sequence
        const bool has = (wmask >> lane) & 1;
        const unsigned off = has ? (woff + (unsigned)__popcll(wmask_all & ((1ull << lane) - 1))) * 64u : OOB_OFF;
        const int dst = (mine ? ord : 63) * 4;                            // (nslots <= 56: lane 63 is never an ordinal)
        v_seq = __builtin_amdgcn_ds_permute(dst, lane * slot_bytes);
        v_off = __builtin_amdgcn_ds_permute(dst, (int)off);
        if (lane >= nslots) {  // beyond the sequence: a dummy weight block (never read), no pool block
            v_seq = 0;
            v_off = (int)OOB_OFF;
        }
    }
    // next (ordinal, panel); runs on beyond the end of the sequence (at most two steps: the prefetch distance), where the
    // tables answer "no slot"
#define ASR_SEQ_ADVANCE(i, p)                   \
    {                                           \
        const int nx_ = (p) + 1;                \
        const int wrap_ = nx_ == npanel ? 1 : 0; \
        (p) = wrap_ ? 0 : nx_;                  \
        (i) += wrap_;                           \
    }
    constexpr int RSRC_FLAGS = 0x00020000;
    const __amdgpu_buffer_rsrc_t rs_w =
            __builtin_amdgcn_make_buffer_rsrc((void*)packed, 0, (int)(PLANES * plane_stride * 2), RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rs_p =
            __builtin_amdgcn_make_buffer_rsrc((void*)plan.pool, 0, (int)plan.pool_bytes, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rs_i = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(ROWW ? a.inp_importance : zeros), 0, ROWW ? (int)(unsigned)(a.num_inp * 4) : 4, RSRC_FLAGS);
    // weight panel -> LDS: chunk c = s * WAVES + wave (1 KB) of the panel, lane l moves its 16-byte piece; the
    // chunk position is wave-uniform and travels in the scalar offset
    const int lane16 = lane * 16;
    unsigned w_soff[SV];
#pragma unroll
    for (int s = 0; s < SV; ++s) {
        const int c = s * WAVES + wave;
        const int pl = c / CHUNKS_PER_PLANE, ci = c % CHUNKS_PER_PLANE;
        w_soff[s] = (unsigned)((pl * plane_stride + (i64)n0 * KC) * 2 + ci * 1024);
    }
    auto dma_panel = [&](const int qi, const int qp, auto bufc) __attribute__((always_inline)) {
        const int soff = __builtin_amdgcn_readlane(v_seq, qi & 63) + qp * panel_bytes;
        u32x4* dst = decltype(bufc)::value ? s_B1 : s_B0;
#pragma unroll
        for (int s = 0; s < SV; ++s) {
            const int c = s * WAVES + wave;
            if (NCHUNK % WAVES == 0 || c < NCHUNK)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)&dst[c * 64], 16,
                                                         lane16, soff + (int)w_soff[s], 0, 0);
        }
    };
    // neighbour index of this lane's row for slot k (wave-uniform k): one 64-byte block of the pool.  The load
    // is issued whether or not the wave has the slot (offset beyond the pool -> 0), so that every step has the
    // same sequence of memory instructions; has_slot() tells the two apart.
    // ordinal i (any value: lanes beyond the sequence hold OOB_OFF) -> the wave's pool block; the range check of a raw
    // buffer looks at the VECTOR offset only, so a wave without the slot puts OOB_OFF there and gets zeros
    // (a wave without the slot: OOB_OFF in the vector offset -> zeros, whatever the scalar offset)
    auto lacks = [&](const int i) __attribute__((always_inline)) -> int {
        return (unsigned)__builtin_amdgcn_readlane(v_off, i & 63) == OOB_OFF ? (int)OOB_OFF : 0;
    };
    auto load_idx = [&](const int i) __attribute__((always_inline)) -> int {
        return (int)__builtin_amdgcn_raw_buffer_load_b32(rs_p, r * 4 + lacks(i), __builtin_amdgcn_readlane(v_off, i & 63), 0);
    };
    auto load_idx4 = [&](const int i) __attribute__((always_inline)) -> u32x4 {  // rows 4 g .. 4 g + 3
        return __builtin_amdgcn_raw_buffer_load_b128(rs_p, g * 16 + lacks(i), __builtin_amdgcn_readlane(v_off, i & 63), 0);
    };
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
            (void*)a.inp_features, 0, (int)(unsigned)(a.num_inp > 0 ? ((a.num_inp - 1) * a.inp_ld + cin) * ESZ : 0),
            RSRC_FLAGS);
    int cache_i = -2;
    unsigned cache_off = OOB_OFF;
    int pref_idx = -1;  // index for the slot after the one being gathered, in flight
    // LINE: whole-line gathers (round 6).  Lane L loads piece (L & 7) of rows (L >> 3) and 8 + (L >> 3): eight adjacent lanes
    // cover one 128-byte line, one tag look-up per line instead of two half-line ones (the operand layout has lane (r, g) on
    // bytes [32 g, 32 g + 32) of row r: the four lanes of a row take 64 of its line's 128 bytes per instruction), and the rows
    // go through 2 KB of LDS per wave into the operand layout (s_stage).  The gather-bound instances of level 0 are bound by
    // the L1's REQUEST rate, not by bytes or latency: a 4-byte touch of the next slot's rows made them 11-18 % slower, the
    // whole-line addressing alone (wrong operands, no transpose) 16 % faster.  Same-box A/B, level 0 at 10 M points: 64-column
    // layers 3 885 -> 3 224 us (four launches), 32-column 2 367 -> 1 954 us, two-bank 64-column 912 -> 810 us; the two-bank
    // 128-column instances of levels 1-2 -1 %.  Not for the plain 128-column instances: 64 KB of LDS = two blocks per CU instead
    // of three on MFMA-bound layers (+8 %).
    constexpr bool LDMA = plan_ldma<NT, KC, WAVES, MODE, IMP, DUAL>();
    constexpr bool LINE = plan_line<NT, KC, WAVES, MODE, IMP, DUAL>();
    // LDMA (plain 8-wave instances, gathers one step ahead): the two whole-line gathers are buffer_load ... lds, i.e. the rows
    // go HBM -> LDS without passing through registers (lane L fetches piece (L & 7) ^ (row & 7) so that it lands in the
    // swizzled column), eight registers fewer than the register form: the 64-column instance keeps four blocks per CU
    unsigned cache_off1 = OOB_OFF;
    int pref_idx1 = -1;
    const int lrow = lane >> 3, lpiece = lane & 7;
    auto load_idx_row = [&](const int i, const int row) __attribute__((always_inline)) -> int {
        return (int)__builtin_amdgcn_raw_buffer_load_b32(rs_p, row * 4 + lacks(i), __builtin_amdgcn_readlane(v_off, i & 63), 0);
    };
    // (the launcher sends matrices of 4 GB and more and cin that is not a multiple of KC to k_sconv_mfma16)
    auto gather_a = [&](const int qi, const int qp, u32x4 (&aq)[NJ * AW], const bool first) __attribute__((always_inline)) {
        const bool sw = qi != cache_i;
        if constexpr (LINE) {
            int idx0 = pref_idx, idx1 = pref_idx1;
            if (first) {
                idx0 = load_idx_row(qi, lrow);
                idx1 = load_idx_row(qi, 8 + lrow);
            }
            if (sw) {
                cache_i = qi;
                const bool has = lacks(qi) == 0;
                const int pc = LDMA ? (lpiece ^ (lrow & 7)) : lpiece;  // (LDMA: the swizzle is applied on the source side)
                cache_off = has && idx0 >= 0 ? (unsigned)idx0 * (unsigned)(a.inp_ld * ESZ) + (unsigned)(pc * 16) : OOB_OFF;
                cache_off1 = has && idx1 >= 0 ? (unsigned)idx1 * (unsigned)(a.inp_ld * ESZ) + (unsigned)(pc * 16) : OOB_OFF;
                pref_idx = load_idx_row(qi + 1, lrow);
                pref_idx1 = load_idx_row(qi + 1, 8 + lrow);
            }
            const int soff = qp * KC * ESZ;
            if constexpr (LDMA) {  // lane L's 16 bytes land at stage + 16 L: rows 0 .. 7, then rows 8 .. 15
                u32x4* st = s_stage + wave * 128;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)st, 16, (int)cache_off, soff, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(st + 64), 16, (int)cache_off1, soff, 0, 0);
            } else {
                aq[0] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, (int)cache_off, soff, 0);
                aq[1] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, (int)cache_off1, soff, 0);
            }
            return;
        }
        int idx = pref_idx;
        if (first) idx = load_idx(qi);
        if (sw) {  // once per slot: take the prefetched index, prefetch the one of the slot after it
            cache_i = qi;
            const bool valid = lacks(qi) == 0 && idx >= 0;
            cache_off = valid ? (unsigned)idx * (unsigned)(a.inp_ld * ESZ) + (unsigned)(8 * g * ESZ) : OOB_OFF;
            pref_idx = load_idx(qi + 1);
        }
        const int soff = qp * KC * ESZ;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int hh = 0; hh < AW; ++hh)
                aq[j * AW + hh] =
                        __builtin_amdgcn_raw_buffer_load_b128(rs_a, (int)cache_off + (32 * j + 4 * hh) * ESZ, soff, 0);
    };

    // Feature gathers run two steps ahead, or one step ahead where the eight registers that saves buy another
    // block per CU (plain 8-wave instances: 80 registers -> three blocks of NT = 8; measured per layer)
    constexpr int DEPTH = ((WAVES == 8 && !IMP && !DUAL) || LDMA) ? 1 : 2;
    u32x4 a_q0[NJ * AW], a_q1[DEPTH == 2 ? NJ * AW : 1];
    // (k_cur, k1, k2: ORDINALS of the slot of this step, the next one and the one after it; p_*: their panels)
    int k_cur = 0, p_cur = 0;
    int k1 = 0, p1 = 0;
    ASR_SEQ_ADVANCE(k1, p1)
    u32x4 idx4_cur = {~0u, ~0u, ~0u, ~0u}, idx4_next = {~0u, ~0u, ~0u, ~0u};
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    if (k_cur < nslots) {
        dma_panel(k_cur, p_cur, B0());
        if (ROWW && roww) idx4_next = load_idx4(k_cur);
        gather_a(k_cur, p_cur, a_q0, true);
        if constexpr (DEPTH == 2) gather_a(k1, p1, a_q1, false);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    auto step = [&](u32x4 (&aq)[NJ * AW], auto bufc) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value;
        int k2 = k1, p2 = p1;
        ASR_SEQ_ADVANCE(k2, p2)
        if constexpr (LDMA) {
            // the rows of THIS step were the last two loads of the previous one: they have landed when nothing is in flight;
            // they leave the stage (into aq) before the next step's rows are sent into it
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const u32x4* st = s_stage + wave * 128;
            // f32 rows: the lane's eight values are pieces 2 g, 2 g + 1; f16 rows (two 32-deep halves): pieces g and 4 + g
            aq[0] = st[r * 8 + ((MODE == ASR_CONV16_F16 ? g : 2 * g) ^ (r & 7))];
            aq[1] = st[r * 8 + ((MODE == ASR_CONV16_F16 ? 4 + g : 2 * g + 1) ^ (r & 7))];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        // panel of the next step straight into the other LDS buffer (every wave is past the barrier that ended
        // its reads).  Issued before every other load of this step: the wait at the end of the step counts on it.
        dma_panel(k1, p1, std::integral_constant<int, BUF ^ 1>());
        // (into the stage: the registers of this step's rows are untouched.  The two-bank instances issue it after their
        // importance loads, which this step still waits for: the gathers must be younger than those)
        if constexpr (LDMA && !ROWW) gather_a(k1, p1, aq, false);
        __builtin_amdgcn_sched_barrier(0);
        const bool active = lacks(k_cur) == 0;
        float w4[4] = {0.f, 0.f, 0.f, 0.f};
        const bool slot_end = p_cur == npanel - 1;
        if (ROWW && roww) {
            // only where they are needed (wave-uniform branches): the wait at the end of the step needs at least the
            // NJ*AW gathers after the panel DMA, more loads in between only make it wait for them as well
            if (p_cur == 0) idx4_cur = idx4_next;
            if (p1 == 0) idx4_next = load_idx4(k1);  // first step of the next slot is the next step
            if (slot_end && active) {
#pragma unroll
                for (int i = 0; i < 4; ++i)  // idx -1 -> beyond the buffer -> 0
                    w4[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_i, (int)(idx4_cur[i] * 4u), 0, 0));
            }
        }
        u32x4 fa[NJ][PLANES];
        if (active) {
            if constexpr (LINE && !LDMA) {
                u32x4* st = s_stage + wave * 128;
                const int wcol = lpiece ^ (lrow & 7);
                st[lrow * 8 + wcol] = aq[0];
                st[(8 + lrow) * 8 + wcol] = aq[1];
                __builtin_amdgcn_wave_barrier();  // (LDS operations of a wave complete in order)
                aq[0] = st[r * 8 + ((2 * g) ^ (r & 7))];
                aq[1] = st[r * 8 + ((2 * g + 1) ^ (r & 7))];
                __builtin_amdgcn_wave_barrier();
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if constexpr (MODE == ASR_CONV16_F16) {
                    fa[j][0] = aq[j];
                } else if constexpr (MODE == ASR_CONV16_F16X2) {
                    sconv16_split_f16x2(aq[j * 2], aq[j * 2 + 1], a_scale, fa[j][0], fa[j][PLANES > 1 ? 1 : 0]);
                } else {
                    unsigned p0[4], p1v[4], p2v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const f32x2 v = {__uint_as_float(aq[j * 2 + (i >> 1)][2 * (i & 1)]),
                                         __uint_as_float(aq[j * 2 + (i >> 1)][2 * (i & 1) + 1])};
                        p0[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
                        const f32x2 r1 = {v.x - __uint_as_float(p0[i] << 16), v.y - __uint_as_float(p0[i] & 0xffff0000u)};
                        p1v[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
                        const f32x2 r2 = {r1.x - __uint_as_float(p1v[i] << 16), r1.y - __uint_as_float(p1v[i] & 0xffff0000u)};
                        p2v[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
                    }
                    fa[j][0] = (u32x4){p0[0], p0[1], p0[2], p0[3]};
                    fa[j][PLANES > 1 ? 1 : 0] = (u32x4){p1v[0], p1v[1], p1v[2], p1v[3]};
                    fa[j][PLANES > 2 ? 2 : 0] = (u32x4){p2v[0], p2v[1], p2v[2], p2v[3]};
                }
            }
        }
        if constexpr (DEPTH == 2)
            gather_a(k2, p2, aq, false);
        else if constexpr (!LDMA || ROWW)
            gather_a(k1, p1, aq, false);
        if (active) {
            const u32x4* sb = BUF ? s_B1 : s_B0;
            __builtin_amdgcn_s_setprio(1);
            sconv16_products<NT, KC, MODE, IMP, DUAL, PLANES, NJ>(fa, sb, acc, tacc, has_b, ncol, g, lo);
            __builtin_amdgcn_s_setprio(0);
            if (ROWW && roww && slot_end) {
#pragma unroll
                for (int i = 0; i < 4; ++i) norm4[i] += w4[i];
                if (IMP) {
#pragma unroll
                    for (int nb = 0; nb < NT; ++nb) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[nb][i] += w4[i] * tacc[nb][i];
                        tacc[nb] = {0.f, 0.f, 0.f, 0.f};
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc_b[i] += w4[i] * tacc[0][i];
                    acc[NT - 1] += tacc[0];  // bank a's columns of the shared tile: the slot's plain sums
                    tacc[0] = {0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        // the panel DMA of this step has landed once at most the NJ*AW gather loads issued after it are in flight
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NJ * AW) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        k_cur = k1;
        p_cur = p1;
        k1 = k2;
        p1 = p2;
    };
    while (k_cur < nslots) {
        step(a_q0, B0());
        if (k_cur >= nslots) break;
        if constexpr (DEPTH == 2)
            step(a_q1, B1());
        else
            step(a_q0, B1());
    }
#undef ASR_SEQ_ADVANCE

#if ASR_BF16X3_CHAIN == 4
#pragma unroll
    for (int nb = 0; nb < NT; ++nb) acc[nb] += lo[nb];
#endif
    // output rows of this lane's four accumulator rows (-1: beyond the list)
    int q4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const i64 lr = row0 + wave * 16 + 4 * g + i;
        q4[i] = lr < a.num_out ? (a.row_perm ? a.row_perm[lr] : (int)lr) : -1;
    }
    if constexpr (SPLIT) {
        if (partial) {  // part[range][row position][column], unscaled (powers of two: exact)
            float* pr = sp.part + (i64)split_range_of_y((int)blockIdx.y) * sp.stride;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const i64 lr = row0 + wave * 16 + 4 * g + i;
#pragma unroll
                for (int nb = 0; nb < NT; ++nb)
                    pr[lr * ctot_pad + n0 + nb * 16 + ncol] = acc[nb][i] * unscale[0] * unscale[1];
            }
            return;
        }
    }
    sconv16_epilogue<NT, MODE, DUAL>(a, acc, acc_b, q4, norm4, n0, ncol, ca, cout, has_b, out_f16, zeros, unscale);
    if (ROWW && a.out_importance && ncol == 0 && (DUAL ? has_b : ychunk == 0)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (q4[i] >= 0) a.out_importance[q4[i]] = norm4[i];
    }
}

// Second pass of the slot-range split: one 256-thread block per 16-row group of a tile (a tile's rows are 16 x ctot_pad / 4
// float4 pieces per group).  Tiles with a single range were finished by the first pass.
template <int WAVES>
__global__ __launch_bounds__(256) void k_sconv_split_reduce(asr_sparse_conv_args a, asr_conv_plan_view plan, asr_split_args sp,
                                                            int ctot_pad, const float* __restrict__ zeros) {
    constexpr int TM = WAVES * 16;
    const i64 tile = blockIdx.x / WAVES;
    const int grp_in_tile = (int)(blockIdx.x % WAVES);
    unsigned long long bmask = 0;
    for (int w = 0; w < WAVES; ++w) {  // (uniform loads)
        const i64 grp = tile * WAVES + w;
        if (grp < plan.groups) {
            const uint4 h = plan.hdr[grp];
            bmask |= (unsigned long long)h.x | ((unsigned long long)h.y << 32);
        }
    }
    bmask &= (1ull << a.kernel_size) - 1;
    int ranges = 0;
#pragma unroll
    for (int s = 0; s < SPLIT_RANGES; ++s) ranges += (bmask & split_range_mask(s)) != 0;
    if (ranges < 2) return;
    const int c4n = ctot_pad / 4;
    unsigned amax = 0;
    const bool res_v4 = a.residual && a.residual_ld % 4 == 0 && ((uintptr_t)a.residual & 15) == 0;
    const bool out_v4 = a.out_ld % 4 == 0 && ((uintptr_t)a.out & 15) == 0;
    for (int e = threadIdx.x; e < 16 * c4n; e += blockDim.x) {
        const int r = e / c4n, c = 4 * (e % c4n);
        const i64 lr = tile * TM + grp_in_tile * 16 + r;
        if (lr >= a.num_out) break;
        const i64 q = a.row_perm ? a.row_perm[lr] : lr;
        float4 pv[SPLIT_RANGES];
#pragma unroll
        for (int s = 0; s < SPLIT_RANGES; ++s)  // all loads first (wave-uniform predicates)
            if (bmask & split_range_mask(s)) pv[s] = *reinterpret_cast<const float4*>(sp.part + s * sp.stride + lr * ctot_pad + c);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        bool first = true;
#pragma unroll
        for (int s = 0; s < SPLIT_RANGES; ++s) {
            if (!(bmask & split_range_mask(s))) continue;
            if (first) {
                v = pv[s];
                first = false;
            } else {
                v.x += pv[s].x;
                v.y += pv[s].y;
                v.z += pv[s].z;
                v.w += pv[s].w;
            }
        }
        float o[4] = {v.x, v.y, v.z, v.w};
        float4 res = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool full4 = c + 3 < a.cout;
        if (res_v4 && full4) res = *reinterpret_cast<const float4*>(a.residual + q * a.residual_ld + c);
        const float rr[4] = {res.x, res.y, res.z, res.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int col = c + i;
            if (col >= a.cout) continue;
            float x = o[i] + (a.bias ? a.bias[col] : 0.f);
            if (a.relu) x = fmaxf(x, 0.f);
            if (a.residual) x += (res_v4 && full4) ? rr[i] : a.residual[q * a.residual_ld + col];
            o[i] = x;
            amax = max(amax, __float_as_uint(x) & 0x7fffffffu);
        }
        if (full4 && out_v4) {
            *reinterpret_cast<float4*>(a.out + q * a.out_ld + c) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (c + i < a.cout) a.out[q * a.out_ld + c + i] = o[i];
        }
    }
    if (a.out_absmax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = max(amax, (unsigned)__shfl_xor((int)amax, o, 64));
        if ((threadIdx.x & 63) == 0 && amax > __hip_atomic_load(a.out_absmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(a.out_absmax, amax);
    }
}


}  // namespace

// ==========================================================================================
static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// f16 tensors whose padded cin is a multiple of 64 use 64-deep panels (bf16x3 panels carry three planes)
static inline int panel_depth(int mode, int cin) { return mode == ASR_CONV16_F16 && round_up(cin, 32) % 64 == 0 ? 64 : 32; }

static inline bool mode_ok(int mode) { return mode == ASR_CONV16_F16 || mode == ASR_CONV16_BF16X3 || mode == ASR_CONV16_F16X2; }

// f16x2: two planes + a 16-byte trailer: [0] the exponent of the weights' power-of-two scale, [1] their largest magnitude
size_t asr_conv16_packed_bytes(int mode, int K, int cin, int cout, int cout_b) {
    const size_t planes = mode == ASR_CONV16_BF16X3 ? 3 : (mode == ASR_CONV16_F16X2 ? 2 : 1);
    return planes * (size_t)K * round_up(cout + cout_b, 16) * round_up(cin, 32) * sizeof(u16) +
           (mode == ASR_CONV16_F16X2 ? 16 : 0);
}

// largest |x| of a [rows, c] matrix with row stride ld, as f32 bits, into *out (device)
int asr_conv16_absmax(asr_hip_context* ctx, const float* x, i64 rows, int c, i64 ld, unsigned* out) {
    ASR_HIP_CHECK(ctx, hipMemsetAsync(out, 0, sizeof(unsigned), ctx->stream));
    const i64 total = rows * c;
    if (total <= 0) return ASR_HIP_OK;
    k_absmax<<<(unsigned)std::min<i64>((total + 255) / 256, 8192), 256, 0, ctx->stream>>>(x, rows, c, ld, out);
    ASR_CHECK_LAUNCH(ctx);
    return ASR_HIP_OK;
}

int asr_conv16_pack(asr_hip_context* ctx, int mode, const float* wa, const float* wb, int K, int cin, int ca, int cb,
                    void* out) {
    if (!mode_ok(mode))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "sparse_conv_pack: mode must be ASR_CONV16_F16, ASR_CONV16_BF16X3 or ASR_CONV16_F16X2");
    if (!wa || !out || K < 1 || K > 56 || cin < 1 || ca < 1 || cb < 0 || (cb > 0 && !wb))
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "sparse_conv_pack: bad argument");
    const int cin_pad = round_up(cin, 32), ctot_pad = round_up(ca + cb, 16);
    const i64 total = (i64)K * ctot_pad * cin_pad;
    unsigned* w_absmax = nullptr;
    if (mode == ASR_CONV16_F16X2) {  // one scale for both banks; the trailer's spare bytes hold the maximum
        w_absmax = (unsigned*)((u16*)out + 2 * total) + 1;
        ASR_TRY(asr_conv16_absmax(ctx, wa, (i64)K * cin, ca, ca, w_absmax));
        if (cb > 0) {
            k_absmax<<<(unsigned)std::min<i64>(((i64)K * cin * cb + 255) / 256, 8192), 256, 0, ctx->stream>>>(
                    wb, (i64)K * cin, cb, cb, w_absmax);
            ASR_CHECK_LAUNCH(ctx);
        }
    }
    k_pack_filters<<<(unsigned)std::min<i64>((total + 255) / 256, 65535), 256, 0, ctx->stream>>>(
            wa, wb, K, cin, ca, cb, cin_pad, ctot_pad, mode, panel_depth(mode, cin), (u16*)out, w_absmax);
    ASR_CHECK_LAUNCH(ctx);
    return ASR_HIP_OK;
}

int asr_conv16_convert(asr_hip_context* ctx, const void* in, i64 n, void* out, int to_f16) {
    if (n <= 0) return ASR_HIP_OK;
    if (to_f16)
        k_f32_to_f16<<<grid_for(n, 256), 256, 0, ctx->stream>>>((const float*)in, n, (u16*)out);
    else
        k_f16_to_f32<<<grid_for(n, 256), 256, 0, ctx->stream>>>((const u16*)in, n, (float*)out);
    ASR_CHECK_LAUNCH(ctx);
    return ASR_HIP_OK;
}

// a: shapes, CSR, bias, flags as for asr_conv_sparse; in ASR_CONV16_F16 mode inp_features / residual (and out
// when out_f16) point to f16 data, leading dimensions count elements.
int asr_conv_sparse16(asr_hip_context* ctx, const asr_sparse_conv_args* pa, const void* packed, int mode,
                      int out_f16, const asr_conv_plan* plan) {
    asr_sparse_conv_args a = *pa;
    if (a.num_out <= 0) return ASR_HIP_OK;
    if (!mode_ok(mode)) ASR_FAIL(ctx, ASR_HIP_EINVAL, "sparse_conv16: unknown mode");
    if (mode != ASR_CONV16_F16 && out_f16) ASR_FAIL(ctx, ASR_HIP_EINVAL, "sparse_conv16: bf16x3 and f16x2 write f32");
    if (a.kernel_size < 1 || a.kernel_size > 56) ASR_FAIL(ctx, ASR_HIP_EINVAL, "sparse_conv16: kernel_size must be 1..56");
    const bool dual = a.cout_b > 0;
    const bool imp = a.inp_importance || a.neighbors_importance;
    const int esz = mode == ASR_CONV16_F16 ? 2 : 4;
    const int gran = 16 / esz;  // elements per 16-byte gather piece
    if (a.cin % gran != 0 || a.inp_ld % gran != 0 || (uintptr_t)a.inp_features % 16 != 0 || (uintptr_t)packed % 16 != 0)
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "sparse_conv16: cin and the row stride must be multiples of %d, 16-byte rows", gran);
    if (dual && (!a.inp_importance || a.neighbors_importance || a.cout % 16 != 8 || a.cout_b != 8 || a.residual))
        ASR_FAIL(ctx, ASR_HIP_EINVAL,
                 "sparse_conv16: second filter bank needs inp_importance (per input row), cout %% 16 == 8, cout_b == 8");
    if (a.inp_ld < a.cin || a.out_ld < a.cout + a.cout_b)
        ASR_FAIL(ctx, ASR_HIP_EINVAL, "sparse_conv16: row stride smaller than channel count");
    const float* zeros = nullptr;
    ASR_TRY(asr_ctx_zeros(ctx, &zeros));
    if (mode == ASR_CONV16_F16X2 && !a.inp_absmax) {  // nobody kept the running maximum of this input: one pass over it
        if (!ctx->d_absmax) ASR_HIP_CHECK(ctx, hipMalloc((void**)&ctx->d_absmax, 256 * sizeof(unsigned)));
        unsigned* m = ctx->d_absmax + 255;
        ASR_TRY(asr_conv16_absmax(ctx, a.inp_features, a.num_inp, a.cin, a.inp_ld, m));
        a.inp_absmax = m;
    }
    const int cin_pad = round_up(a.cin, 32), ctot_pad = round_up(a.cout + a.cout_b, 16);
    // column tile: the widest of 128 / 64 / 32 / 16 that divides the padded width, narrowed while the launch
    // has too few blocks (as asr_conv_sparse)
    int nt = 8;
    while (nt > 1 && ctot_pad % (nt * 16) != 0) nt >>= 1;
    const i64 tiles64 = (a.num_out + 63) / 64;
    if (a.force_nt) {
        if ((a.force_nt != 1 && a.force_nt != 2 && a.force_nt != 4 && a.force_nt != 8) || ctot_pad % (a.force_nt * 16) != 0)
            ASR_FAIL(ctx, ASR_HIP_EINVAL, "sparse_conv16: force_nt must be 1, 2, 4 or 8 and divide the padded width");
        nt = a.force_nt;
    } else {
        while (nt > 2 && tiles64 * (ctot_pad / (nt * 16)) < ctx->opt.sconv16_min_blocks) nt >>= 1;
    }
    const i64 tiles128 = (a.num_out + 127) / 128;
    bool wide = tiles128 * (ctot_pad / (nt * 16)) >= ctx->opt.sconv_wide_min;
    if (a.force_waves) {
        if (a.force_waves != 4 && a.force_waves != 8) ASR_FAIL(ctx, ASR_HIP_EINVAL, "sparse_conv16: force_waves must be 4 or 8");
        wide = a.force_waves == 8;
    }
    const bool kc64 = panel_depth(mode, a.cin) == 64;
    // the plan-driven kernel covers everything but per-entry importance and neighbour-count normalisation
    const bool use_plan = plan && plan->usable() && ctx->opt.sconv_plan && !a.neighbors_importance &&
                          !(a.normalize && !imp) && !(a.out_importance && !imp) &&
                          a.num_inp * a.inp_ld * esz < (i64(1) << 32) - 65536 && a.cin % (kc64 ? 64 : 32) == 0;
    asr_conv_plan_view pv = {nullptr, nullptr, 0, 0};
    if (use_plan) {
        if (plan->num_out != a.num_out || plan->K < a.kernel_size || plan->perm != a.row_perm)
            ASR_FAIL(ctx, ASR_HIP_EINVAL, "sparse_conv16: the plan was built for another list");
        pv = plan->view();
    }
    // slot-range split (see split_range_mask): plain 55-slot convolutions over a coarse grid.  Decided from the INPUT grid's
    // row count, which a rank of a sharded cloud shares with the one-GPU run.
    const bool split = use_plan && !dual && !imp && mode != ASR_CONV16_F16 && a.kernel_size == 55 && !a.force_nt &&
                       !a.force_waves && ctot_pad % 64 == 0 && ctx->opt.sconv_split_rows > 0 &&
                       a.num_inp <= ctx->opt.sconv_split_rows && a.num_inp >= ctx->opt.sconv_split_min_rows;
    asr_split_args spa = {nullptr, 0};
    if (split) {
        nt = ctot_pad % 128 == 0 ? 8 : 4;
        wide = true;
        const i64 tiles = (a.num_out + 127) / 128;
        spa.stride = tiles * 128 * ctot_pad;
        const size_t need = (size_t)SPLIT_RANGES * spa.stride * sizeof(float);
        if (need > ctx->split_part_bytes) {  // grown on demand, kept for the context's life
            if (ctx->split_part) {
                ASR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
                (void)hipFree(ctx->split_part);
                ctx->split_part = nullptr;
                ctx->split_part_bytes = 0;
            }
            ASR_HIP_CHECK(ctx, hipMalloc((void**)&ctx->split_part, need + need / 4));
            ctx->split_part_bytes = need + need / 4;
        }
        spa.part = ctx->split_part;
    }
    if (ctx->dry_launch) return ASR_HIP_OK;  // preparation pass of the sharded network: checked and allocated, not launched
#define ASR_L16(NT_, KC_, W_, M_, I_, D_)                                                                        \
    {                                                                                                            \
        const i64 tiles_ = (a.num_out + W_ * 16 - 1) / (W_ * 16);                                                \
        const i64 ny_ = ctot_pad / (NT_ * 16);                                                                   \
        dim3 grid((unsigned)(ny_ > 1 ? ((tiles_ + 7) / 8) * 8 * ny_ : tiles_));                                  \
        if (use_plan)                                                                                            \
            k_sconv_plan16<NT_, KC_, W_, M_, I_, D_><<<grid, dim3(W_ * 64), 0, ctx->stream>>>(                   \
                    a, pv, (const u16*)packed, cin_pad, ctot_pad, out_f16, zeros, spa);                          \
        else                                                                                                     \
            k_sconv_mfma16<NT_, KC_, W_, M_, I_, D_><<<grid, dim3(W_ * 64), 0, ctx->stream>>>(                   \
                    a, (const u16*)packed, cin_pad, ctot_pad, out_f16, zeros);                                   \
    }
#define ASR_L16_ID(NT_, KC_, W_, M_) \
    if (dual)                        \
        ASR_L16(NT_, KC_, W_, M_, false, true) \
    else if (imp)                    \
        ASR_L16(NT_, KC_, W_, M_, true, false) \
    else                             \
        ASR_L16(NT_, KC_, W_, M_, false, false)
#define ASR_L16_W(NT_, KC_, M_) \
    if (wide)                   \
        ASR_L16_ID(NT_, KC_, 8, M_) \
    else                        \
        ASR_L16_ID(NT_, KC_, 4, M_)
#define ASR_L16_NT(KC_, M_)                 \
    switch (nt) {                           \
        case 8: ASR_L16_W(8, KC_, M_) break; \
        case 4: ASR_L16_W(4, KC_, M_) break; \
        case 2: ASR_L16_W(2, KC_, M_) break; \
        default: ASR_L16_W(1, KC_, M_) break; \
    }
#define ASR_L16_SPLIT(NT_, M_)                                                                                   \
    {                                                                                                            \
        const i64 tiles_ = (a.num_out + 127) / 128;                                                              \
        const i64 ny_ = ctot_pad / (NT_ * 16);                                                                   \
        dim3 grid((unsigned)(ny_ > 1 ? ((tiles_ + 7) / 8) * 8 * ny_ : tiles_), SPLIT_RANGES);                    \
        k_sconv_plan16<NT_, 32, 8, M_, false, false, true><<<grid, dim3(512), 0, ctx->stream>>>(                 \
                a, pv, (const u16*)packed, cin_pad, ctot_pad, out_f16, zeros, spa);                              \
        ASR_CHECK_LAUNCH(ctx);                                                                                   \
        k_sconv_split_reduce<8><<<dim3((unsigned)(tiles_ * 8)), dim3(256), 0, ctx->stream>>>(a, pv, spa, ctot_pad, zeros); \
    }
    if (split) {
        if (mode == ASR_CONV16_F16X2) {
            if (nt == 8)
                ASR_L16_SPLIT(8, ASR_CONV16_F16X2)
            else
                ASR_L16_SPLIT(4, ASR_CONV16_F16X2)
        } else {
            if (nt == 8)
                ASR_L16_SPLIT(8, ASR_CONV16_BF16X3)
            else
                ASR_L16_SPLIT(4, ASR_CONV16_BF16X3)
        }
    } else if (mode == ASR_CONV16_F16) {
        if (kc64)
            ASR_L16_NT(64, ASR_CONV16_F16)
        else
            ASR_L16_NT(32, ASR_CONV16_F16)
    } else if (mode == ASR_CONV16_F16X2) {
        ASR_L16_NT(32, ASR_CONV16_F16X2)
    } else {
        ASR_L16_NT(32, ASR_CONV16_BF16X3)
    }
#undef ASR_L16_SPLIT
#undef ASR_L16_NT
#undef ASR_L16_W
#undef ASR_L16_ID
#undef ASR_L16
    ASR_CHECK_LAUNCH(ctx);
    {
        char key[64];  // NT,KC,IMP,WAVES,DUAL,MODE,PLAN[,1 = slot-range split] (k_sconv_mfma16 / k_sconv_plan16 instance)
        snprintf(key, sizeof(key), "%d,%d,%d,%d,%d,%d,%d%s", nt, kc64 ? 64 : 32, imp && !dual ? 1 : 0, wide ? 8 : 4,
                 dual ? 1 : 0, mode, use_plan ? 1 : 0, split ? ",1" : "");
        ++ctx->sconv_launches[key];
    }
    return ASR_HIP_OK;
}

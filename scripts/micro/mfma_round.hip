// How does v_mfma_f32_16x16x32_bf16 (gfx950) round?  One wave, one instruction per case; prints the accumulator bits.
// Build: hipcc --offload-arch=gfx950 -O2 scripts/micro/mfma_round.hip -o scripts/micro/mfma_round
// Cases (A row 0 x B column 0, the other rows / columns zero; C = c everywhere):
//   1  one product p = 1.5 * 2^-24 (0.75 ulp of 1), C = 1:        RNE -> 1 + 2^-23, truncation -> 1
//   2  one product p = 0.5 * 2^-23 exactly half an ulp, C = 1:    RNE (ties to even) -> 1
//   3  32 products of 2^-26 each (sum 2^-21 = 4 ulp), C = 1:      exact sum -> 1 + 2^-21; products dropped one by one -> 1
//   4  32 products of 2^-28 each (sum 2^-23 = 1 ulp), C = 1:      exact -> 1 + 2^-23
//   5  one product 1 + 2^-8 times 1 + 2^-8 (needs 17 bits), C = 0: the product itself exact? -> 1 + 2^-7 + 2^-16
//   6  two products 1 and 2^-30, C = -1:                           exact cancellation keeps 2^-30?
//   7  case 1 with C = -1 and p negative (sign symmetry)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline unsigned short bf(float x) { return (unsigned short)(__float_as_uint(x) >> 16); }  // x exactly representable

struct Case {
    float a[32], b[32], c;
};

__global__ void k(const Case* cs, int n, float* out) {
    const int lane = threadIdx.x, row = lane & 15, g = lane >> 4;
    for (int t = 0; t < n; ++t) {
        unsigned short av[8], bv[8];
        for (int i = 0; i < 8; ++i) {  // lane (row, g) holds k = 8 g + i of its row (A) / column (B)
            av[i] = row == 0 ? bf(cs[t].a[8 * g + i]) : 0;
            bv[i] = row == 0 ? bf(cs[t].b[8 * g + i]) : 0;
        }
        bf16x8 A, B;
        memcpy(&A, av, 16);
        memcpy(&B, bv, 16);
        f32x4 C = {cs[t].c, cs[t].c, cs[t].c, cs[t].c};
        C = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, C, 0, 0, 0);
        if (lane == 0) out[t] = C[0];  // D[row 0][col 0]
    }
}

int main() {
    Case h[7];
    memset(h, 0, sizeof(h));
    auto p2 = [](int e) { return (float)ldexp(1.0, e); };
    h[0].a[0] = 1.5f; h[0].b[0] = p2(-24); h[0].c = 1.f;
    h[1].a[0] = 1.0f; h[1].b[0] = p2(-24); h[1].c = 1.f;
    for (int i = 0; i < 32; ++i) { h[2].a[i] = 1.f; h[2].b[i] = p2(-26); }
    h[2].c = 1.f;
    for (int i = 0; i < 32; ++i) { h[3].a[i] = 1.f; h[3].b[i] = p2(-28); }
    h[3].c = 1.f;
    h[4].a[0] = 1.f + p2(-7); h[4].b[0] = 1.f + p2(-7); h[4].c = 0.f;  // bf16 has 8 significant bits: 1 + 2^-7
    h[5].a[0] = 1.f; h[5].b[0] = 1.f; h[5].a[1] = 1.f; h[5].b[1] = p2(-30); h[5].c = -1.f;
    h[6].a[0] = -1.5f; h[6].b[0] = p2(-24); h[6].c = -1.f;
    Case* d;
    float* o;
    hipMalloc(&d, sizeof(h));
    hipMalloc(&o, 7 * sizeof(float));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, 7, o);
    float r[7];
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    const char* what[7] = {"C=1 + 0.75 ulp (RNE: 1+2^-23 = 3f800001)", "C=1 + exactly 0.5 ulp (ties-even: 3f800000)",
                           "C=1 + 32 x 2^-26 (exact: 3f800004)", "C=1 + 32 x 2^-28 (exact: 3f800001)",
                           "(1+2^-7)^2, C=0 (exact: 1+2^-6+2^-14 = 3f820200)", "1 + 2^-30 - 1 (exact: 2^-30 = 30800000)",
                           "C=-1 - 0.75 ulp (RNE: bf800001)"};
    for (int t = 0; t < 7; ++t) {
        uint32_t u;
        memcpy(&u, &r[t], 4);
        printf("case %d  %-55s -> %08x  %.10g\n", t + 1, what[t], u, r[t]);
    }
    return 0;
}

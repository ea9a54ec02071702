// Lane layout of v_mfma_f32_4x4x1_16B_f32 (gfx950): 16 independent 4x4 blocks, D_b = A_b (4x1) B_b (1x4) + C_b.
// Expected: A_b[i] in lane 4 b + i, B_b[j] in lane 4 b + j, D_b[i][j] in register i of lane 4 b + j.
// Build: hipcc --offload-arch=gfx950 -O2 scripts/micro/mfma_4x4.hip -o scripts/micro/mfma_4x4
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k(float* out) {
    const int lane = threadIdx.x;
    const float a = 1.f + lane;          // A_b[i] = 1 + 4 b + i
    const float b = 100.f * (1 + lane);  // B_b[j] = 100 (1 + 4 b + j)
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
}

int main() {
    float* d;
    hipMalloc(&d, 256 * sizeof(float));
    k<<<1, 64>>>(d);
    float h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 4; ++r) {
            const int blk = lane / 4, j = lane % 4;
            const float want = (1.f + 4 * blk + r) * 100.f * (1 + 4 * blk + j);
            if (h[lane * 4 + r] != want) {
                if (bad < 8) printf("lane %d reg %d: got %g want %g\n", lane, r, h[lane * 4 + r], want);
                ++bad;
            }
        }
    printf(bad ? "layout differs in %d places\n" : "layout as expected (%d)\n", bad);
    return 0;
}

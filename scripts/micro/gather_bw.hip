// Microbenchmark: per-CU throughput of the sparse-conv gather pattern (16 rows x 128 B per wave-instruction pair,
// 16 B per lane) as a function of loads in flight and of the footprint the rows come from.
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/micro/gather_bw scripts/micro/gather_bw.hip ; run on the GPU box: scripts/micro/gather_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH, int MAP>
__global__ __launch_bounds__(512) void k_gather(const float* __restrict__ feat, unsigned bytes, const int* __restrict__ idx,
                                                int n_idx, int iters, int ld_bytes, unsigned* out) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)feat, 0, (int)bytes, 0x00020000);
    const int lane = threadIdx.x & 63;
    const int r = MAP == 0 ? (lane & 15) : MAP == 1 ? (lane >> 2) : (lane >> 3);
    const int g = MAP == 0 ? (lane >> 4) : MAP == 1 ? (lane & 3) : (lane & 7);
    const unsigned lane_off = MAP == 0 ? 32u * g : 16u * g;
    const unsigned second = MAP == 0 ? 16u : MAP == 1 ? 64u : 0u;  // MAP 2: second load = rows + 8 (other idx)
    const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    u32x4 q[DEPTH][2];
    unsigned acc = 0;
    int pos = (wave_global * 16 + r) % n_idx;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        const unsigned off = (unsigned)idx[pos] * (unsigned)ld_bytes + lane_off;
        const unsigned off2 = MAP == 2 ? (unsigned)idx[(pos + 8) % n_idx] * (unsigned)ld_bytes + lane_off : off + second;
        q[d][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
        q[d][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off2, 0, 0);
        pos = (pos + 7919) % n_idx;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            acc += q[d][0][0] ^ q[d][1][3];
            const unsigned off = (unsigned)idx[pos] * (unsigned)ld_bytes + lane_off;
            const unsigned off2 = MAP == 2 ? (unsigned)idx[(pos + 8) % n_idx] * (unsigned)ld_bytes + lane_off : off + second;
            q[d][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
            q[d][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off2, 0, 0);
            pos = (pos + 7919) % n_idx;
        }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) acc += q[d][0][1] ^ q[d][1][2];
    if (acc == 0x12345678u) out[0] = acc;
}

// MAP 3: the same 16 rows x 128 B per pair of wave instructions, but global -> LDS by DMA (buffer_load ... lds): lane l moves
// piece l & 7 of row l >> 3 (eight adjacent lanes = one 128-byte line) into LDS slot l; READ: the wave then reads its two
// 16-byte operand pieces (row l & 15, pieces 2 (l >> 4), + 1) back with ds_read_b128
template <int DEPTH, bool READ>
__global__ __launch_bounds__(512) void k_gather_lds(const float* __restrict__ feat, unsigned bytes, const int* __restrict__ idx,
                                                    int n_idx, int iters, int ld_bytes, unsigned* out) {
    __shared__ __attribute__((aligned(16))) u32x4 s_a[8 * DEPTH * 128];  // per wave and slot: 16 rows x 8 pieces
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)feat, 0, (int)bytes, 0x00020000);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r8 = lane >> 3, g8 = lane & 7;
    const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    unsigned acc = 0;
    int pos = (wave_global * 16 + r8) % n_idx;
    u32x4* base = s_a + wave * DEPTH * 128;
    auto issue = [&](int d) __attribute__((always_inline)) {
        const unsigned off = (unsigned)idx[pos] * (unsigned)ld_bytes + 16u * g8;
        const unsigned off2 = (unsigned)idx[(pos + 8) % n_idx] * (unsigned)ld_bytes + 16u * g8;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)&base[d * 128], 16, (int)off, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)&base[d * 128 + 64], 16, (int)off2, 0, 0, 0);
        pos = (pos + 7919) % n_idx;
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue(d);
    const int rr = lane & 15, gg = lane >> 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (DEPTH - 1)) : "memory");
            if (READ) {
                const u32x4 a = base[d * 128 + rr * 8 + ((2 * gg) ^ (rr & 7))];
                const u32x4 b = base[d * 128 + rr * 8 + ((2 * gg + 1) ^ (rr & 7))];
                acc += a[0] ^ b[3];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            issue(d);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const int ld = 256;  // 64 channels f32
    const size_t rows_max = 2500000;
    float* feat;
    hipMalloc(&feat, rows_max * ld);
    hipMemset(feat, 0, rows_max * ld);
    unsigned* out;
    hipMalloc(&out, 4);
    const int n_idx = 1 << 22;
    int* d_idx;
    hipMalloc(&d_idx, n_idx * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t foot[] = {64, 8192, 65536, 2500000};  // rows: 16 KB (L1), 2 MB (L2), 16 MB (MALL), 640 MB (HBM)
    for (size_t rows : foot) {
        std::vector<int> h(n_idx);
        srand(1);
        for (int i = 0; i < n_idx; ++i) h[i] = (int)(((size_t)rand() * 32768 + rand()) % rows);
        hipMemcpy(d_idx, h.data(), n_idx * 4, hipMemcpyHostToDevice);
        for (int map : {0, 1, 2, 3, 4})
        for (int blocks_per_cu : {2}) {
            for (int depth : {2}) {
                const int iters = 2000 / depth;
                const int grid = 256 * blocks_per_cu;
                auto launch = [&]() {
                    if (map == 0) k_gather<2, 0><<<grid, 512>>>(feat, (unsigned)(rows * ld), d_idx, n_idx, iters, ld, out);
                    if (map == 1) k_gather<2, 1><<<grid, 512>>>(feat, (unsigned)(rows * ld), d_idx, n_idx, iters, ld, out);
                    if (map == 2) k_gather<2, 2><<<grid, 512>>>(feat, (unsigned)(rows * ld), d_idx, n_idx, iters, ld, out);
                    if (map == 3) k_gather_lds<2, false><<<grid, 512>>>(feat, (unsigned)(rows * ld), d_idx, n_idx, iters, ld, out);
                    if (map == 4) k_gather_lds<2, true><<<grid, 512>>>(feat, (unsigned)(rows * ld), d_idx, n_idx, iters, ld, out);
                };
                launch();
                hipDeviceSynchronize();
                hipEventRecord(e0);
                launch();
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                const double bytes = (double)grid * 8 * (iters * depth) * 2048.0;
                printf("map %d rows %8zu  blocks/CU %d depth %d : %7.2f TB/s  (%5.1f B/clk/CU at 2.4 GHz)\n", map, rows, blocks_per_cu,
                       depth, bytes / ms / 1e9, bytes / ms / 1e9 * 1e12 / 256 / 2.4e9 / 1e0 / 1e0 * 1.0);
            }
        }
    }
    return 0;
}

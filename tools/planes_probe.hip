// Experiment (tools only, not part of the library): what does a pointwise split-operand contraction reach when BOTH
// operands already exist as bf16 planes in memory — weights as today (ws[chunk16][plane][row][16 bf16]), activations as
// [plane][k-octet][pixel][8 bf16] — so that a wavefront's stream is loads + MFMAs only: no LDS, no barrier, no staging
// arithmetic?  (three bf16 planes are a lossless image of an fp32 value, 6 bytes for 4.)
// Wavefront w of a 256-thread workgroup owns all 128 rows of the tile and pixels 32 w .. 32 w + 31: per 16-channel chunk
// 12 A fragments (shared by the four wavefronts through L1) + 3 B fragments (its own) feed 24 MFMAs.
//
//   hipcc -O3 --offload-arch=gfx950 -o tools/_bin/planes_probe tools/planes_probe.hip && tools/_bin/planes_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ bf16x8 bf(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

__device__ __forceinline__ f32x16 mfma6(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a[2]), bf(b[0]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a[0]), bf(b[2]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a[1]), bf(b[1]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a[1]), bf(b[0]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a[0]), bf(b[1]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(a[0]), bf(b[0]), c, 0, 0, 0);
    return c;
}

// A: [nchunk][3][M][16 bf16] (u32x4 index ((ch*3 + p)*M + row)*2 + h);  B: [3][K/8][N][8 bf16] (u32x4 index (p*K/8 + o)*N + px)
__global__ __launch_bounds__(256, 2) void planes_kernel(const u32x4* __restrict__ A, const u32x4* __restrict__ B,
                                                        float* __restrict__ C, int M, int K, int N) {
    const int mt = M / 128;
    const int tile = blockIdx.x, i0 = (tile % mt) * 128, j0 = (tile / mt) * 128;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lh = lane >> 5;
    const int nchunk = K / 16, no = K / 8;
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const u32x4* pa = A + (size_t)(i0 + l31) * 2 + lh;
    const u32x4* pb = B + (size_t)lh * N + j0 + wave * 32 + l31;
    u32x4 af[2][4][3], bfr[2][3];
    auto load = [&](int ch, int set) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            bfr[set][p] = pb[((size_t)p * no + 2 * ch) * N];
#pragma unroll
            for (int a = 0; a < 4; ++a) af[set][a][p] = pa[((size_t)(ch * 3 + p) * M + a * 32) * 2];
        }
    };
    load(0, 0);
    for (int ch = 0; ch < nchunk; ch += 2) {
        if (ch + 1 < nchunk) load(ch + 1, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a] = mfma6(af[0][a], bfr[0], acc[a]);
        __builtin_amdgcn_sched_barrier(0);
        if (ch + 2 < nchunk) load(ch + 2, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (ch + 1 < nchunk) {
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a] = mfma6(af[1][a], bfr[1], acc[a]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const int col = j0 + wave * 32 + l31;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = i0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            C[(size_t)row * N + col] = acc[a][r];
        }
}

static void run(int M, int K, int N, const u32x4* A, const u32x4* B, float* C) {
    const int grid = (M / 128) * (N / 128);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(planes_kernel, dim3(grid), dim3(256), 0, 0, A, B, C, M, K, N);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 10;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(planes_kernel, dim3(grid), dim3(256), 0, 0, A, B, C, M, K, N);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double flops = 2.0 * M * K * (double)N;
    printf("M %4d K %4d N %6d: %7.1f us  %6.1f TF (fp32-equivalent; 6 bf16 MFMA terms)  %4d tiles  planes read %.0f MB + C %.0f MB\n", M,
           K, N, ms * 1e3, flops / (ms * 1e-3) / 1e12, grid, (6.0 * K * N + 6.0 * M * K) / 1e6, 4.0 * M * N / 1e6);
}

int main() {
    const size_t maxA = (size_t)2048 * 2048 * 6, maxB = (size_t)301056 * 64 * 6 > (size_t)75264 * 512 * 6 ? (size_t)301056 * 64 * 6 : (size_t)75264 * 512 * 6;
    const size_t maxC = (size_t)256 * 301056 * 4 > (size_t)1024 * 18816 * 4 ? (size_t)256 * 301056 * 4 : (size_t)1024 * 18816 * 4;
    std::vector<uint32_t> h(1 << 22);
    uint32_t s = 12345;
    for (auto& v : h) {
        uint32_t w = 0;
        for (int k = 0; k < 2; ++k) {
            s = s * 1664525u + 1013904223u;
            const uint32_t r = s >> 8;
            w |= (((r & 1) << 15) | ((120 + ((r >> 1) & 7)) << 7) | ((r >> 4) & 0x7f)) << (16 * k);
        }
        v = w;
    }
    u32x4 *A, *B;
    float* C;
    CHECK(hipMalloc(&A, maxA));
    CHECK(hipMalloc(&B, maxB));
    CHECK(hipMalloc(&C, maxC));
    for (size_t off = 0; off < maxA; off += h.size() * 4) CHECK(hipMemcpy((char*)A + off, h.data(), std::min(h.size() * 4, maxA - off), hipMemcpyHostToDevice));
    for (size_t off = 0; off < maxB; off += h.size() * 4) CHECK(hipMemcpy((char*)B + off, h.data(), std::min(h.size() * 4, maxB - off), hipMemcpyHostToDevice));
    run(256, 512, 75264, A, B, C);      // 512 -> 256 @28x28, batch 96   (split kernel today: 112 us, 176 TF)
    run(1024, 256, 18816, A, B, C);     // 256 -> 1024 @14x14             (65 us, 150 TF)
    run(128, 512, 75264, A, B, C);      // 512 -> 128 @28x28              (69 us, 143 TF)
    run(256, 64, 301056, A, B, C);      // 64 -> 256 @56x56               (118 us, 83 TF)
    run(512, 2048, 4608, A, B, C);      // 2048 -> 512 @7x7 (4704 px)     (103 us, 95 TF)
    return 0;
}

// Calibration probe (tools only, not part of the library): what rate does the bf16 matrix pipe of an MI355X actually
// deliver for the instruction stream of the split-operand kernels' consumer loop — six dependent
// v_mfma_f32_32x32x16_bf16 per accumulator tile, operands re-read from LDS by ds_read_b128 — on RANDOM data, and at
// which clock?  Variants: registers only / LDS fragment reads / + one s_barrier per 48 MFMAs / 16x16x32 shape;
// 1, 2 or 3 wavefronts per SIMD.  Prints TFLOP/s (dense bf16), the fraction of 2.5 PF, the share of cycles the pipe
// was busy at the in-kernel clock (s_memtime / s_memrealtime, MI355X_MICROARCH.md 'DVFS give-back' item 6).
//
//   hipcc -O3 --offload-arch=gfx950 -o tools/_bin/mfma_probe tools/mfma_probe.hip && tools/_bin/mfma_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

struct Stamp { unsigned long long cyc, rt; };

__device__ __forceinline__ bf16x8 as_bf(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// MODE 0: registers only   1: B fragments from LDS (3 x ds_read_b128 per group, one group ahead)
// 2: mode 1 + __syncthreads() every 8 groups (48 MFMAs)      SHAPE 0: 32x32x16, 1: 16x16x32 (4 tiles per group)
template <int MODE, int SHAPE>
__global__ __launch_bounds__(256) void probe(const u32x4* __restrict__ src, float* __restrict__ out, Stamp* stamps,
                                             int iters) {
    extern __shared__ __align__(16) u32x4 lds[];          // 24 KB: [3 planes][4 octets][128 px]
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 1536; i += 256) lds[i] = src[(blockIdx.x * 1536 + i) % (1 << 16)];
    __syncthreads();
    u32x4 a[3], b[2][3];
    for (int p = 0; p < 3; ++p) a[p] = src[(tid * 3 + p + 7) % (1 << 16)];
    for (int p = 0; p < 3; ++p) b[0][p] = b[1][p] = lds[p * 512 + lane];
    f32x16 acc[4];
    f32x4 acc4[4][4];
    for (int t = 0; t < 4; ++t) {
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        for (int u = 0; u < 4; ++u) acc4[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int l31 = lane & 31, lh = lane >> 5;
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int t = g & 3, cur = g & 1, nxt = cur ^ 1;
            if constexpr (MODE >= 1) {
                const u32x4* p = lds + ((g >> 2) * 2 + lh) * 128 + ((g + 1) & 3) * 32 + l31;
#pragma unroll
                for (int q = 0; q < 3; ++q) b[nxt][q] = p[q * 512];
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (SHAPE == 0) {
                f32x16 c = acc[t];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a[2]), as_bf(b[cur][0]), c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a[0]), as_bf(b[cur][2]), c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a[1]), as_bf(b[cur][1]), c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a[1]), as_bf(b[cur][0]), c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a[0]), as_bf(b[cur][1]), c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a[0]), as_bf(b[cur][0]), c, 0, 0, 0);
                acc[t] = c;
            } else {
                // the same FLOPs as one 32x32x16 group: 12 x 16x16x32 (half the work each) on 4 accumulator tiles
#pragma unroll
                for (int rep = 0; rep < 1; ++rep)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        f32x4 c = acc4[t][u];
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(a[(rep + u) % 3]), as_bf(b[cur][0]), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(a[(rep + u + 1) % 3]), as_bf(b[cur][1]), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(a[(rep + u + 2) % 3]), as_bf(b[cur][2]), c, 0, 0, 0);
                        acc4[t][u] = c;
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (MODE == 2) __syncthreads();
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int t = 0; t < 4; ++t) {
        for (int r = 0; r < 16; ++r) s += acc[t][r];
        for (int u = 0; u < 4; ++u) s += acc4[t][u][0] + acc4[t][u][3];
    }
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) stamps[blockIdx.x] = Stamp{c1 - c0, r1 - r0};
}

template <int MODE, int SHAPE>
static void run(const char* name, int wg_per_cu, const u32x4* src, float* out, Stamp* stamps, int iters) {
    const int grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto k = probe<MODE, SHAPE>;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 24576, 0, src, out, stamps, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 24576, 0, src, out, stamps, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    std::vector<Stamp> h(grid);
    CHECK(hipMemcpy(h.data(), stamps, grid * sizeof(Stamp), hipMemcpyDeviceToHost));
    std::vector<double> ghz;
    for (auto& s : h) ghz.push_back((double)s.cyc / ((double)s.rt * 10.0) );   // realtime ticks at 100 MHz -> ns x 10
    std::sort(ghz.begin(), ghz.end());
    const double clk = ghz[ghz.size() / 2];
    // FLOPs: per group 6 x 32x32x16 (or the equivalent 24 x 16x16x32 / 2...): count instructions
    const double flop_per_group = SHAPE == 0 ? 6.0 * 2 * 32 * 32 * 16 : 12.0 * 2 * 16 * 16 * 32;
    const double flops = (double)grid * 4 * iters * 8 * flop_per_group;
    const double tf = flops / (ms * 1e-3) / 1e12;
    const double busy_cycles_per_simd = (double)wg_per_cu * iters * 8 * (SHAPE == 0 ? 6 * 32.0 : 12 * 16.0);
    printf("%-34s %d wave/SIMD  %8.3f ms  %7.1f TF  %.3f of 2.5PF  clock %.2f GHz  pipe busy %.3f\n", name, wg_per_cu,
           ms, tf, tf / 2500.0, clk, busy_cycles_per_simd / (ms * 1e-3 * clk * 1e9));
}

int main() {
    const int N = 1 << 16;
    std::vector<uint32_t> h(N * 4);
    uint32_t s = 12345;
    for (auto& v : h) {                       // random bf16 pairs in [-2, 2): random sign, exponent 120..127, mantissa
        uint32_t w = 0;
        for (int k = 0; k < 2; ++k) {
            s = s * 1664525u + 1013904223u;
            const uint32_t r = s >> 8;
            const uint32_t b = ((r & 1) << 15) | ((120 + ((r >> 1) & 7)) << 7) | ((r >> 4) & 0x7f);
            w |= b << (16 * k);
        }
        v = w;
    }
    u32x4* src;
    float* out;
    Stamp* stamps;
    CHECK(hipMalloc(&src, N * 16));
    CHECK(hipMalloc(&out, 256 * 4 * 256 * 4));
    CHECK(hipMalloc(&stamps, 256 * 4 * sizeof(Stamp)));
    CHECK(hipMemcpy(src, h.data(), N * 16, hipMemcpyHostToDevice));
    const int iters = 4000;     // 32k groups per wave ~ 6 M MFMA-cycles ~ 3 ms
    for (int w = 1; w <= 3; ++w) {
        run<0, 0>("32x32x16 registers only", w, src, out, stamps, iters);
        run<1, 0>("32x32x16 + LDS fragment reads", w, src, out, stamps, iters);
        run<2, 0>("32x32x16 + LDS + barrier/48", w, src, out, stamps, iters);
        run<0, 1>("16x16x32 registers only", w, src, out, stamps, iters);
        run<1, 1>("16x16x32 + LDS fragment reads", w, src, out, stamps, iters);
    }
    return 0;
}

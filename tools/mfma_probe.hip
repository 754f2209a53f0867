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


// Two kinds of wavefronts on every SIMD, as in the producer/consumer kernels: waves 0-3 run the MFMA + ds_read_b128
// consumer loop above, waves 4-7 run what a producer does for the same 48 MFMAs — PM bit 0: the three-way split of 16
// floats per lane (+ scale/shift/max as the fused BatchNorm + ReLU), bit 1: the ds_write_b64 of the planes, bit 2: the
// 16 floats come from global memory (a 64 MB window, so L2 and HBM traffic as in the kernels) instead of registers.
// Reported: time of the MFMA wavefronts (their own s_memtime stamps) and of the whole kernel.
__device__ __forceinline__ unsigned pk(float a, float b) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f2{a, b}, bf2));
}
template <int PM, int NPW = 4, bool MF = true, int PRIO = 0, bool AG = false, bool S16 = false>
__global__ __launch_bounds__(256 + 64 * NPW) void probe_pc(const u32x4* __restrict__ src, const float* __restrict__ big,
                                                float* __restrict__ out, Stamp* stamps, int iters) {
    extern __shared__ __align__(16) u32x4 lds[];          // 24 KB fragments + 3 x 24 KB stage buffers
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 1536; i += 256 + 64 * NPW) lds[i] = src[(blockIdx.x * 1536 + i) % (1 << 16)];
    __syncthreads();
    if (wave < 4) {
        if constexpr (PRIO < 0) __builtin_amdgcn_s_setprio(-PRIO);
        u32x4 a[3], b[2][3];
        for (int p = 0; p < 3; ++p) a[p] = src[(tid * 3 + p + 7) % (1 << 16)];
        for (int p = 0; p < 3; ++p) b[0][p] = b[1][p] = lds[p * 512 + lane];
        f32x16 acc[4];
        for (int t = 0; t < 4; ++t)
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        const int l31 = lane & 31, lh = lane >> 5;
        unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < (MF ? iters : 1); ++it) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int t = g & 3, cur = g & 1, nxt = cur ^ 1;
                const u32x4* p = lds + ((g >> 2) * 2 + lh) * 128 + ((g + 1) & 3) * 32 + l31;
#pragma unroll
                for (int q = 0; q < 3; ++q) b[nxt][q] = p[q * 512];
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (S16) {
                    // same FLOPs per group on 16x16x32: 12 instructions of 16 cycles, accumulators f32x4 carved out of acc[t]
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        f32x4 c4 = f32x4{acc[t][4 * u], acc[t][4 * u + 1], acc[t][4 * u + 2], acc[t][4 * u + 3]};
                        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(a[u % 3]), as_bf(b[cur][0]), c4, 0, 0, 0);
                        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(a[(u + 1) % 3]), as_bf(b[cur][1]), c4, 0, 0, 0);
                        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(a[(u + 2) % 3]), as_bf(b[cur][2]), c4, 0, 0, 0);
                        acc[t][4 * u] = c4[0]; acc[t][4 * u + 1] = c4[1]; acc[t][4 * u + 2] = c4[2]; acc[t][4 * u + 3] = c4[3];
                    }
                } else if constexpr (AG) {
#define MF_AG(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(A), "v"(B))
                    MF_AG(a[2], b[cur][0]); MF_AG(a[0], b[cur][2]); MF_AG(a[1], b[cur][1]);
                    MF_AG(a[1], b[cur][0]); MF_AG(a[0], b[cur][1]); MF_AG(a[0], b[cur][0]);
                } else {
                f32x16 c = acc[t];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a[2]), as_bf(b[cur][0]), c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a[0]), as_bf(b[cur][2]), c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a[1]), as_bf(b[cur][1]), c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a[1]), as_bf(b[cur][0]), c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a[0]), as_bf(b[cur][1]), c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a[0]), as_bf(b[cur][0]), c, 0, 0, 0);
                acc[t] = c;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        float s = 0.f;
        for (int t = 0; t < 4; ++t)
            for (int r = 0; r < 16; ++r) s += acc[t][r];
        out[blockIdx.x * 256 + tid] = s;
        if (tid == 0) stamps[blockIdx.x] = Stamp{c1 - c0, r1 - r0};
    } else {
        if constexpr (PRIO > 0) __builtin_amdgcn_s_setprio(PRIO);
        if constexpr (PRIO < 0) __builtin_amdgcn_s_setprio(0);
        const int pt = (tid - 256) & 255, half = (tid - 256) >> 8;     // NPW = 8: two sets of producers, half the stages each
        float x[16];
        for (int j = 0; j < 16; ++j) x[j] = __uint_as_float(src[(pt + j * 256) % (1 << 16)][j & 3]);
        const float sc = 1.0001f, sh = 0.01f;
        unsigned keep = 0;
        // 64 MB window: 16 M floats; a workgroup's stage = 4096 consecutive floats (16 per lane as 4 x b128)
        const f32x4* g4 = reinterpret_cast<const f32x4*>(big);
        unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = half; it < iters; it += NPW / 4) {
            if constexpr ((PM & 4) != 0) {
                const unsigned base = ((unsigned)(blockIdx.x * 977 + it) * 1024u) & ((1u << 22) - 1);   // in f32x4 units
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = g4[base + q * 256 + pt];
                    x[4 * q] = v[0]; x[4 * q + 1] = v[1]; x[4 * q + 2] = v[2]; x[4 * q + 3] = v[3];
                }
            }
            unsigned pl[3][8];
            if constexpr ((PM & 1) != 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float a = fmaxf(x[2 * j] * sc + sh, 0.f), b = fmaxf(x[2 * j + 1] * sc + sh, 0.f);
                    unsigned h = pk(a, b);
                    a -= __uint_as_float(h << 16); b -= __uint_as_float(h & 0xffff0000u);
                    unsigned m = pk(a, b);
                    a -= __uint_as_float(m << 16); b -= __uint_as_float(m & 0xffff0000u);
                    unsigned l = pk(a, b);
                    pl[0][j] = h; pl[1][j] = m; pl[2][j] = l;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) { pl[0][j] = __float_as_uint(x[2 * j]); pl[1][j] = __float_as_uint(x[2 * j + 1]); pl[2][j] = pl[0][j] ^ pl[1][j]; }
            }
            if constexpr ((PM & 2) != 0) {
                u32x4* dst = lds + 1536 + (it % 3) * 1536;
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    dst[p * 512 + pt] = u32x4{pl[p][0], pl[p][1], pl[p][2], pl[p][3]};
                    dst[p * 512 + 256 + pt] = u32x4{pl[p][4], pl[p][5], pl[p][6], pl[p][7]};
                }
            } else {
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int j = 0; j < 8; ++j) keep ^= pl[p][j];
            }
            if constexpr ((PM & 4) == 0) {
#pragma unroll
                for (int j = 0; j < 16; ++j) x[j] = x[j] * 1.0000001f + __uint_as_float(0x33000000u | (keep & 0xff));
            }
        }
        unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
        if (keep == 0x12345u) out[tid] = 1.f;
        if (tid == 256) stamps[1024 + blockIdx.x] = Stamp{0, r1 - r0};
    }
}

template <int PM, int NPW = 4, bool MF = true, int PRIO = 0, bool AG = false, bool S16 = false>
static void run_pc(const char* name, const u32x4* src, const float* big, float* out, Stamp* stamps, int iters) {
    const int grid = 256;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto k = probe_pc<PM, NPW, MF, PRIO, AG, S16>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(256 + 64 * NPW), 98304, 0, src, big, out, stamps, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 3;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(256 + 64 * NPW), 98304, 0, src, big, out, stamps, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    std::vector<Stamp> h(1024 + grid);
    CHECK(hipMemcpy(h.data(), stamps, (1024 + grid) * sizeof(Stamp), hipMemcpyDeviceToHost));
    std::vector<double> tm, tp, ghz;
    for (int i = 0; i < grid; ++i) {
        tm.push_back(h[i].rt * 10e-6);              // ms
        tp.push_back(h[1024 + i].rt * 10e-6);
        ghz.push_back((double)h[i].cyc / ((double)h[i].rt * 10.0));
    }
    std::sort(tm.begin(), tm.end()); std::sort(tp.begin(), tp.end()); std::sort(ghz.begin(), ghz.end());
    const double flops = (double)grid * 4 * iters * 8 * 6.0 * 2 * 32 * 32 * 16;
    const double tmm = tm[grid / 2];
    printf("%-44s kernel %7.3f ms | mfma waves %7.3f ms = %7.1f TF (%.3f of 2.5PF) clock %.2f GHz | producers %7.3f ms\n", name, ms,
           tmm, flops / (tmm * 1e-3) / 1e12, flops / (tmm * 1e-3) / 1e12 / 2500.0, ghz[grid / 2], tp[grid / 2]);
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
    CHECK(hipMalloc(&stamps, 2048 * sizeof(Stamp)));
    CHECK(hipMemcpy(src, h.data(), N * 16, hipMemcpyHostToDevice));
    float* big;
    CHECK(hipMalloc(&big, 64u << 20));
    CHECK(hipMemset(big, 0x3c, 64u << 20));
    if (getenv("PROBE_PC")) {
        const int it = 3000;
        run_pc<0>("pc: producers idle loop", src, big, out, stamps, it);
        run_pc<1>("pc: + split (BN, ReLU, 3 planes)", src, big, out, stamps, it);
        run_pc<2>("pc: + LDS writes only", src, big, out, stamps, it);
        run_pc<3>("pc: + split + LDS writes", src, big, out, stamps, it);
        run_pc<4>("pc: + global loads only", src, big, out, stamps, it);
        run_pc<6>("pc: + global loads + LDS writes", src, big, out, stamps, it);
        run_pc<7>("pc: + global loads + split + LDS writes", src, big, out, stamps, it);
        run_pc<0, 4, true, 0, true>("pc: AccVGPR accumulators, producers idle loop", src, big, out, stamps, it);
        run_pc<1, 4, true, 0, true>("pc: AccVGPR accumulators, split", src, big, out, stamps, it);
        run_pc<7, 4, true, 0, true>("pc: AccVGPR accumulators, loads+split+LDS", src, big, out, stamps, it);
        run_pc<0, 4, true, 0, false, true>("pc: 16x16x32, producers idle loop", src, big, out, stamps, it);
        run_pc<1, 4, true, 0, false, true>("pc: 16x16x32, split", src, big, out, stamps, it);
        run_pc<7, 4, true, 0, false, true>("pc: 16x16x32, loads+split+LDS", src, big, out, stamps, it);
        run_pc<1, 4, true, 1>("pc: split, producers s_setprio 1", src, big, out, stamps, it);
        run_pc<1, 4, true, 3>("pc: split, producers s_setprio 3", src, big, out, stamps, it);
        run_pc<7, 4, true, 1>("pc: loads+split+LDS, producers s_setprio 1", src, big, out, stamps, it);
        run_pc<7, 4, true, 3>("pc: loads+split+LDS, producers s_setprio 3", src, big, out, stamps, it);
        run_pc<7, 8, true, 2>("pc: loads+split+LDS, 8 producers prio 2", src, big, out, stamps, it);
        run_pc<7, 4, true, -2>("pc: loads+split+LDS, MFMA waves prio 2", src, big, out, stamps, it);
        run_pc<1, 8>("pc: split, 8 producer waves", src, big, out, stamps, it);
        run_pc<7, 8>("pc: loads+split+LDS, 8 producer waves", src, big, out, stamps, it);
        run_pc<1, 4, false>("pc: split, MFMA waves idle", src, big, out, stamps, it);
        run_pc<7, 4, false>("pc: loads+split+LDS, MFMA waves idle", src, big, out, stamps, it);
        run_pc<7, 8, false>("pc: loads+split+LDS, 8 producers, MFMA idle", src, big, out, stamps, it);
        return 0;
    }
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

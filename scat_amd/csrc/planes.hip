// Activations that exist as pre-split bf16 planes ("P8" layout) and the pointwise contraction that reads them by LDS-DMA.
//
// The split-operand kernels (conv1x1.hip, conv3x3.hip, conv_wgrad_*.hip) form an fp32 product from the three bf16
// terms of each operand (split.h).  Weights are split once per step (w_prep_batch_kernel); activations used to be
// split by EVERY tile that reads them, on its way into LDS: 5.5 vector instructions per element, Cout / 128 times per
// element forward and again in both gradient kernels, issued by the same SIMDs that issue the MFMAs (on this
// part MFMA time and staging time add).  Here the split happens ONCE, in an HBM-bound pass whose vector unit is idle
// (planes_from_f32_kernel, and the BatchNorm passes of norm.hip that emit planes beside or instead of fp32), and the
// contraction's activation path has no vector instruction at all:
//
//   P8 layout:  planes[p][n][c / 8][pixel][c % 8]  bf16,  p = 0 (hi), 1 (mid), 2 (lo); plane stride = B*C*HW*2 bytes.
//   A (plane, k-octet) row of 64 consecutive pixels is 1 KiB contiguous = ONE `buffer_load_dwordx4 ... lds` per
//   wavefront (lane = pixel, 16 bytes = the pixel's 8 channels = one MFMA operand fragment), landing in the LDS image
//   [plane][k-octet][pixel][8 bf16] that conv1x1_split_kernel builds with ds_write — so a B fragment stays one
//   ds_read_b128 and the MFMA stream is unchanged: the results are bit-identical to the in-kernel split.
//   hi + mid + lo is EXACTLY the fp32 value (3 x 8 significand bits), so a tensor kept as planes loses nothing.
//
// Reference semantics replaced: nn.Conv2d(k=1) forward and its autograd data gradient at models/resnet.py:65-72,84-92
// (conv1 / conv3 of every Bottleneck), fp32.
#include "conv_common.h"
#include "split.h"

namespace scat {

typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int PL_KS = 32;      // channels per stage (= PW_KS of conv1x1.hip: the LDS image is the same)

struct PlDesc {
    const uint16_t* src;   // P8 planes of the activations
    const void* w;         // prepared weights ws[chunk][plane][row][16 bf16] (wprep_octet, conv1x1.hip)
    int C, M, HW, npix;
    FastDiv dHW;
    uint32_t pstride;      // bytes between planes = B * C * HW * 2
    uint32_t src_bytes;    // 3 * pstride
    uint32_t w_bytes;
};

// ---- hand-counted vector-memory operations (cdna_hip_programming.md 5.7): hipcc neither counts these nor waits for them,
// so the loop below owns every s_waitcnt vmcnt.  All vector-memory instructions of the main loop are of these two kinds.
__device__ __forceinline__ i32x4 make_srd(const void* p, uint32_t bytes) {
    const uint64_t a = (uint64_t)p;
    return i32x4{(int)(uint32_t)a, (int)((uint32_t)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
// 64 lanes x 16 bytes from base + voff + soff into LDS at lds_addr + 16 * lane
// (M0 holds the LDS destination; it is compiler-reserved, so the statement saves and restores it: cdna_hip_programming.md 5.7)
__device__ __forceinline__ void dma16(const i32x4 rs, const uint32_t lds_addr, const int voff, const int soff) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ void aload16(u32x4& dst, const i32x4 rs, const int voff, const int soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rs), "s"(soff) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

// One 32*WM x BN output tile.  NB LDS stage buffers, LA = NB - 1 stages of lookahead: at the top of stage s (behind its
// barrier) a wavefront issues, in this order, the 6 weight loads (both 16-channel sub-chunks) and its PPW activation
// pieces of stage s + LA; the wait in front of the barrier of stage s is therefore vmcnt((LA - 1) * (6 + PPW)):
// everything of stage s has landed, LA - 1 younger stages stay in flight.  Loads past the last stage are issued with an
// out-of-range offset (the bounds check drops them) so that the counts never change.
// DIAG (tools build only, results wrong): 1 = no activation DMA, 2 = no weight loads, 4 = no MFMAs, 8 = no output stores,
// 16 = no stage barrier, 32 = one LDS fragment read per stage instead of 24; 64 (results RIGHT) = the loads spread over the stage
template <int WM, int BN, int NB, int DIAG = 0>
__global__ __launch_bounds__(NT, (NB == 2 ? 3 : 2)) void pw_planes_kernel(PlDesc d, OutDesc dc) {
    constexpr int BM = 32 * WM, WN = 4 / WM, NI = BN / (32 * WN);
    constexpr int H = BN / 64;             // 64-pixel pieces per (plane, octet) row
    constexpr int PPW = 3 * H;             // pieces per wavefront per stage (12 rows x H pieces, 4 wavefronts)
    constexpr int LA = NB - 1;
    constexpr int BUF = 12 * BN;           // u32x4 per stage buffer: [3 planes][4 k-octets][BN]
    extern __shared__ __align__(16) float lds[];
    u32x4* const B0 = (u32x4*)lds;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds;

    const int mt = (d.M + BM - 1) / BM, nt = (d.npix + BN - 1) / BN;
    const int tile = xcd_remap(blockIdx.x, mt * nt);
    const int i0 = (tile % mt) * BM, j0 = (tile / mt) * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lh = lane >> 5;
    const int nstage = d.C / PL_KS;

    const i32x4 rs_b = make_srd(d.src, d.src_bytes), rs_a = make_srd(d.w, d.w_bytes);

    // ---- activation pieces of this wavefront: rows (plane, octet) = pair 3*wave + i / H, 64-pixel piece i % H
    int voff[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        int j = j0 + h * 64 + lane;
        j = j < d.npix ? j : d.npix - 1;                       // a dead column is never stored: any valid address will do
        const uint32_t n = d.dHW.div((uint32_t)j);
        voff[h] = (int)((n * (uint32_t)(d.C >> 3) * (uint32_t)d.HW + ((uint32_t)j - n * (uint32_t)d.HW)) * 16u);
    }
    const int stage_soff = 4 * d.HW * 16;                      // 4 k-octets further
    int psoff[PPW];                                            // scalar: plane and octet of piece i
    uint32_t plds[PPW];                                        // scalar: its LDS byte offset inside a stage buffer
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int pair = 3 * wave + i / H, plane = pair >> 2, oct = pair & 3;
        psoff[i] = (int)((uint32_t)plane * d.pstride) + oct * d.HW * 16;
        plds[i] = (uint32_t)(((plane * 4 + oct) * BN + (i % H) * 64) * 16);
    }
    auto issue_b1 = [&](int st, int buf, auto i_tag) {        // piece i of stage st
        constexpr int i = decltype(i_tag)::value;
        const uint32_t la = lds0 + (uint32_t)buf * (BUF * 16) + plds[i];
        const int vo = st < nstage ? voff[i % H] : OOB, so = psoff[i] + st * stage_soff;
        if constexpr (DIAG & 1) asm volatile("" :: "s"(la), "v"(vo), "s"(so));
        else dma16(rs_b, la, vo, so);
    };
    auto issue_b = [&](int st, int buf) {
        static_for<PPW>([&](auto i_tag) { issue_b1(st, buf, i_tag); });
    };

    // ---- weights: lane (row, h) takes 16 bytes per plane and 16-channel sub-chunk
    const int row = i0 + wm * 32 + l31;
    const int aoff = row < d.M ? row * 32 + lh * 16 : OOB;
    const int aplane = d.M * 32;
    auto issue_a1 = [&](u32x4 (&dst)[2][3], int st, auto k_tag) {      // load k = 3 t + p of stage st
        constexpr int t = decltype(k_tag)::value / 3, p = decltype(k_tag)::value % 3;
        const int vo = st < nstage ? aoff : OOB;
        if constexpr (DIAG & 2) dst[t][p] = u32x4{(uint32_t)vo, 0x3f803f80u, (uint32_t)(st + t + p), 0x3f803f80u};
        else aload16(dst[t][p], rs_a, vo, ((2 * st + t) * 3 + p) * aplane);
    };
    auto issue_a = [&](u32x4 (&dst)[2][3], int st) {
        static_for<6>([&](auto k_tag) { issue_a1(dst, st, k_tag); });
    };

    f32x16 acc[1][NI];
#pragma unroll
    for (int b = 0; b < NI; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][b][r] = 0.f;

    const int b_frag = lh * BN + wn * (BN / WN) + l31;          // u32x4 index inside (plane 0, octet pair 0)
    auto read_b = [&](u32x4 (&dst)[3], const u32x4* buf, int t, int b) {
        const u32x4* p = buf + 2 * t * BN + b_frag + b * 32;
#pragma unroll
        for (int q = 0; q < 3; ++q) dst[q] = p[q * 4 * BN];
    };

    u32x4 areg[NB][2][3];
    u32x4 bfr[2][3];

    static_for<LA>([&](auto st_tag) {
        constexpr int ST = decltype(st_tag)::value;
        issue_a(areg[ST], ST);
        issue_b(ST, ST);
    });

    auto stage = [&](int s, auto cur_tag) {
        constexpr int CUR = decltype(cur_tag)::value, NXT = (CUR + LA) % NB;
        wait_vm<(LA - 1) * (((DIAG & 2) ? 0 : 6) + ((DIAG & 1) ? 0 : PPW))>();
        // the values the asm loads produced are only now what the registers hold: nothing may be scheduled across
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int p = 0; p < 3; ++p) asm volatile("" : "+v"(areg[CUR][t][p]));
        if constexpr (!(DIAG & 16)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(DIAG & 64)) {
            issue_a(areg[NXT], s + LA);
            issue_b(s + LA, NXT);
        }
        const u32x4* bcur = B0 + CUR * BUF;
        read_b(bfr[0], bcur, 0, 0);
        static_for<2 * NI>([&](auto i_tag) {
            constexpr int I = decltype(i_tag)::value, t = I / NI, b = I % NI;
            constexpr int fcur = I & 1, fnxt = fcur ^ 1;
            if constexpr (DIAG & 64) {
                // the same 6 + PPW vector-memory instructions in the same order, spread over the stage's MFMA groups
                // instead of issued as one burst behind the barrier
                constexpr int NG = 2 * NI, TOT = 6 + PPW;
                constexpr int lo = I * TOT / NG, hi = (I + 1) * TOT / NG;
                static_for<hi - lo>([&](auto q_tag) {
                    constexpr int q = lo + decltype(q_tag)::value;
                    if constexpr (q < 6) issue_a1(areg[NXT], s + LA, std::integral_constant<int, q>{});
                    else issue_b1(s + LA, NXT, std::integral_constant<int, q - 6>{});
                });
            }
            if constexpr (DIAG & 32) {
#pragma unroll
                for (int q = 0; q < 3; ++q) bfr[fnxt][q] = bfr[fcur][q];
            } else {
                if constexpr (b + 1 < NI) read_b(bfr[fnxt], bcur, t, b + 1);
                else if constexpr (t == 0) read_b(bfr[fnxt], bcur, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (DIAG & 4) {
#pragma unroll
                for (int q = 0; q < 3; ++q) asm volatile("" :: "v"(areg[CUR][t][q]), "v"(bfr[fcur][q]));
            } else {
                acc[0][b] = mfma_split(areg[CUR][t], bfr[fcur], acc[0][b]);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    for (int s = 0; s < nstage; s += NB) {
        static_for<NB>([&](auto k_tag) {
            constexpr int K = decltype(k_tag)::value;
            if (s + K < nstage) stage(s + K, k_tag);
        });
    }
    wait_vm<0>();                                               // (dropped out-of-range loads of the last LA stages)
    asm volatile("" ::: "memory");
    if constexpr (DIAG & 8) {
        if (d.npix != 0x7fffffff) return;                       // (never false: the stores below keep the accumulators live)
    }
    store_tile<1, NI, BM, BN, WM, WN>(acc, dc, d.M, d.npix, i0, j0, 0);
}

// ---------------------------------------------------------------- fp32 NCHW -> P8 planes (optionally through the
// fused BatchNorm + ReLU of the producing layer): 4 B read + 6 B written per element, the split on an idle vector unit.
// V4: a thread owns one k-octet x 4 consecutive pixels (eight 16-byte loads along pixels, 3 x 64 contiguous bytes out).
template <bool TF, bool V4>
__global__ __launch_bounds__(256) void planes_from_f32_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ shift, int relu, int C, int HW,
                                                              int64_t nitems, int64_t plane_u4) {
    constexpr int PV = V4 ? 4 : 1;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= nitems) return;
    const int HWq = HW / PV, C8 = C >> 3;
    const int q = (int)(e % HWq);
    const int64_t no = e / HWq;                   // n * C8 + o
    const int o = (int)(no % C8);
    const int64_t n = no / C8;
    const float* s = src + ((n * C + 8 * o) * (int64_t)HW + (int64_t)q * PV);
    float x[8][PV];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        if constexpr (V4) {
            const f32x4 v = *(const f32x4*)(s + (int64_t)m * HW);
            x[m][0] = v[0]; x[m][1] = v[1]; x[m][2] = v[2]; x[m][3] = v[3];
        } else {
            x[m][0] = s[(int64_t)m * HW];
        }
    }
    if constexpr (TF) {
        const float lo = relu ? 0.f : -__builtin_inff();
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const float sc = scale[8 * o + m], sh = shift[8 * o + m];
#pragma unroll
            for (int k = 0; k < PV; ++k) x[m][k] = fmaxf(fmaf(x[m][k], sc, sh), lo);
        }
    }
    u32x4* out = (u32x4*)dst + (no * HW + (int64_t)q * PV);
#pragma unroll
    for (int k = 0; k < PV; ++k) {
        float v[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m] = x[m][k];
        u32x4 hi, mid, lo;
        split3x8(v, hi, mid, lo);
        out[k] = hi;
        out[plane_u4 + k] = mid;
        out[2 * plane_u4 + k] = lo;
    }
}

}  // namespace scat

using namespace scat;

extern "C" int64_t scat_planes_bytes(int B, int C, int HW) { return (int64_t)3 * B * C * HW * 2; }

extern "C" int scat_planes_from_f32(const float* src, void* planes, int B, int C, int HW, const float* in_scale,
                                    const float* in_shift, int in_relu, void* stream) {
    SCAT_REQUIRE(src && planes, SCAT_E_ARG, "scat_planes_from_f32: null pointer");
    SCAT_REQUIRE(B > 0 && C > 0 && HW > 0 && C % 8 == 0, SCAT_E_SHAPE, "scat_planes_from_f32: C must be a multiple of 8");
    SCAT_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), SCAT_E_ARG, "scat_planes_from_f32: scale/shift pair");
    SCAT_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)planes & 15) == 0, SCAT_E_ARG,
                 "scat_planes_from_f32: 16-B alignment");
    SCAT_REQUIRE(fits_i32(scat_planes_bytes(B, C, HW)), SCAT_E_SHAPE, "scat_planes_from_f32: planes exceed 2 GiB");
    hipStream_t st = (hipStream_t)stream;
    const bool v4 = HW % 4 == 0;
    const int64_t nitems = (int64_t)B * (C / 8) * (HW / (v4 ? 4 : 1));
    const int64_t plane_u4 = (int64_t)B * (C / 8) * HW;
    const dim3 grid((unsigned)((nitems + 255) / 256));
    uint16_t* dst = (uint16_t*)planes;
    if (in_scale) {
        if (v4) hipLaunchKernelGGL((planes_from_f32_kernel<true, true>), grid, dim3(256), 0, st, src, dst, in_scale, in_shift, in_relu, C, HW, nitems, plane_u4);
        else hipLaunchKernelGGL((planes_from_f32_kernel<true, false>), grid, dim3(256), 0, st, src, dst, in_scale, in_shift, in_relu, C, HW, nitems, plane_u4);
    } else {
        if (v4) hipLaunchKernelGGL((planes_from_f32_kernel<false, true>), grid, dim3(256), 0, st, src, dst, in_scale, in_shift, 0, C, HW, nitems, plane_u4);
        else hipLaunchKernelGGL((planes_from_f32_kernel<false, false>), grid, dim3(256), 0, st, src, dst, in_scale, in_shift, 0, C, HW, nitems, plane_u4);
    }
    SCAT_LAUNCH_CHECK("scat_planes_from_f32");
    return SCAT_OK;
}

template <int WM, int BN, int NB, int DIAG = 0>
static void launch_pw_planes(const PlDesc& d, const OutDesc& dc_in, hipStream_t st) {
    constexpr int BM = 32 * WM;
    const int mt = cdiv(d.M, BM), nt = cdiv(d.npix, BN);
    constexpr size_t lds_bytes = (size_t)NB * 12 * BN * 16;
    OutDesc dc = dc_in;
    if (!dc.accumulate && !dc.bias) {                 // a forward convolution: its BatchNorm's sums ride in the epilogue
        dc.sg = nt * (4 / WM);
        dc.stats = epi_stats_take(d.M, dc.sg, &dc.stats_shift);
    }
    auto kern = pw_planes_kernel<WM, BN, NB, DIAG>;
    if constexpr (lds_bytes > 64 * 1024) {
        static bool once = (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                (int)lds_bytes) == hipSuccess);
        (void)once;
    }
    hipLaunchKernelGGL(kern, dim3(mt * nt), dim3(NT), lds_bytes, st, d, dc);
}

extern "C" int scat_conv1x1_planes(const void* planes, const float* w, float* dst, int B, int C, int HW, int M,
                                   int transposed, const float* bias, int accumulate, void* ws, int64_t ws_bytes,
                                   int w_ready, int lds_stages, void* stream) {
    SCAT_REQUIRE(planes && w && dst, SCAT_E_ARG, "scat_conv1x1_planes: null pointer");
    SCAT_REQUIRE(math_mode() == 1, SCAT_E_ARG, "scat_conv1x1_planes: needs the split-operand product mode");
    SCAT_REQUIRE(B > 0 && C > 0 && HW > 0 && M > 0 && C % 32 == 0, SCAT_E_SHAPE,
                 "scat_conv1x1_planes: channels must be a multiple of 32");
    SCAT_REQUIRE(ws && ws_bytes >= scat_conv1x1_s1_ws(M, C) && ((uintptr_t)ws & 15) == 0, SCAT_E_WORKSPACE,
                 "scat_conv1x1_planes: workspace too small / unaligned");
    SCAT_REQUIRE(((uintptr_t)planes & 15) == 0, SCAT_E_ARG, "scat_conv1x1_planes: 16-B alignment");
    SCAT_REQUIRE(fits_i32(scat_planes_bytes(B, C, HW)) && fits_i32((int64_t)B * M * HW * 4) &&
                     fits_i32(scat_conv1x1_s1_ws(M, C)),
                 SCAT_E_SHAPE, "scat_conv1x1_planes: tensor exceeds 32-bit byte offsets");
    hipStream_t st = (hipStream_t)stream;
    if (!w_ready) wprep_launch(wprep_job(w, ws, M, C, transposed, 1, 1, 1, 1, 0, 0, 1), st);
    PlDesc d{};
    d.src = (const uint16_t*)planes; d.w = ws; d.C = C; d.M = M; d.HW = HW; d.npix = B * HW;
    d.dHW = FastDiv::make(HW);
    d.pstride = (uint32_t)((int64_t)B * C * HW * 2);
    d.src_bytes = 3u * d.pstride;
    d.w_bytes = (uint32_t)scat_conv1x1_s1_ws(M, C);
    OutDesc dc{};
    dc.p = dst; dc.mode = 1; dc.I = M; dc.J = d.npix; dc.C = M; dc.HW = HW; dc.dHW = FastDiv::make(HW);
    dc.bias = bias; dc.bias_mode = bias ? 1 : 0; dc.accumulate = accumulate; dc.n = (int64_t)B * M * HW;
#ifdef SCAT_DIAG
    if (lds_stages >= 100 && M > 64) {       // tools build: timing ablations of the 128 x 128, two-stage kernel (wrong results)
        set_kernel_label("conv1x1_planes_128x128x32_diag%d", lds_stages - 100);
        switch (lds_stages - 100) {
        case 1: launch_pw_planes<4, 128, 2, 1>(d, dc, st); break;
        case 2: launch_pw_planes<4, 128, 2, 2>(d, dc, st); break;
        case 3: launch_pw_planes<4, 128, 2, 3>(d, dc, st); break;
        case 4: launch_pw_planes<4, 128, 2, 4>(d, dc, st); break;
        case 8: launch_pw_planes<4, 128, 2, 8>(d, dc, st); break;
        case 11: launch_pw_planes<4, 128, 2, 11>(d, dc, st); break;
        case 16: launch_pw_planes<4, 128, 2, 16>(d, dc, st); break;
        case 32: launch_pw_planes<4, 128, 2, 32>(d, dc, st); break;
        case 59: launch_pw_planes<4, 128, 2, 59>(d, dc, st); break;
        case 64: launch_pw_planes<4, 128, 2, 64>(d, dc, st); break;
        default: launch_pw_planes<4, 128, 2, 0>(d, dc, st);
        }
        SCAT_LAUNCH_CHECK("scat_conv1x1_planes");
        return SCAT_OK;
    }
#endif
    SCAT_REQUIRE(lds_stages == 0 || lds_stages == 2 || lds_stages == 3, SCAT_E_ARG,
                 "scat_conv1x1_planes: lds_stages must be 0 (library default), 2 or 3");
    const int nb = lds_stages ? lds_stages : 2;
    const int cfg = M > 64 ? 0 : 1;
    set_kernel_label("conv1x1_planes_%sx32_nb%d", cfg == 0 ? "128x128" : "64x128", nb);
    if (cfg == 0) {
        if (nb == 3) launch_pw_planes<4, 128, 3>(d, dc, st); else launch_pw_planes<4, 128, 2>(d, dc, st);
    } else {
        if (nb == 3) launch_pw_planes<2, 128, 3>(d, dc, st); else launch_pw_planes<2, 128, 2>(d, dc, st);
    }
    SCAT_LAUNCH_CHECK("scat_conv1x1_planes");
    return SCAT_OK;
}

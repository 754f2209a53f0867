// Weight gradient of a pointwise (1x1, stride 1) convolution, second generation:
//     dW[co][ci] = sum over pixels  dy[n][co][r] * f(x[n][ci][r]),      f = relu(x*scale+shift) or identity
// (the autograd weight gradient of nn.Conv2d(k=1) at models/resnet.py:65-67,70-72 and of the packed shortcuts :127-132).
//
// What the first kernel (conv_wgrad_split.hip, wgrad_pc_kernel) measured at batch 96: 108-130 TF, matrix pipe 27 % busy,
// 54 % of the wave cycles waiting for memory (profiles/r02_pmc_util.txt).  Its staging thread owns (row, 8 pixels): a
// wave-wide 16-byte load touches 32 rows x 32 bytes — 32 cache lines for 1 KB — and a 128-byte line is fetched by four
// different instructions of two different stages; loads run two 16-pixel stages ahead.  Here:
//   * a stage is 32 pixels = one 128-byte line per row.  EIGHT ADJACENT LANES read one row's line (16 bytes each): a
//     wave-wide load is 8 whole lines, every line is fetched exactly once.
//   * a lane's 4 pixels are half an MFMA k-octet: it splits them into the three bf16 planes and writes 8 bytes per plane
//     into the operand image [plane][octet][row][8 bf16] (its neighbour writes the other half).
//   * tiles of 256 x 128 (or 128 x 256) outputs on eight consumer wavefronts + four producer wavefronts, ONE workgroup
//     per CU with 150 KB of LDS (two stage buffers): a quarter less staging per MFMA than 128 x 128, one barrier per
//     48 MFMAs per wavefront, and the loads of stage s+2 are in flight for a whole stage (~1.7 us) before they are
//     needed.  128 x 128 / 256 x 64 / 64 x 256 tiles on four consumers for the small layers and for the folded
//     BatchNorm backward (two source tensors: twice the staging registers).
//   * images are walked in octets as before (ceil(HW/8) per image, the ragged last one masked), deterministic split-K
//     over stages into fp32 slabs + the fixed-order reduce.
#include "conv_common.h"
#include "split.h"

namespace scat {

struct WgPwDesc {
    const float* dy;      // [B][Cout][HW]
    const float* x;       // [B][Cin][HW]
    const float* scale;   // optional fused input transform on x, per Cin
    const float* shift;
    const float* dy2;     // DSA: dy = ca[co]*dy + cb[co]*dy2 + cc[co]
    const float* coef;    // [3][Cout]
    int relu;
    int Cout, Cin, HW;
    int NO, U;            // octets per image, octets in total
    int NS;               // stages (4 octets each) in total
    int spz;              // stages per split-K slice
    int nsplit, ngroup;   // split-K slices; runs a slice's tiles are cut into (splitk_xcd_map)
    FastDiv dNO;
    int64_t ndy, nx;
};

typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

// WA x WB consumer wavefronts, 64 x 64 outputs each; 4 producer wavefronts.  RAG: HW % 4 != 0 (7 x 7 planes): a quad of
// pixels may straddle the end of an image, loads go pixel by pixel.
// DIAG (diag build only, results are wrong): 1 = no global loads, 2 = no split arithmetic, 4 = no MFMAs, 8 = no LDS
// fragment reads, 16 = no LDS writes
// NP producer wavefronts (4 or 8): a producer's stream is latency-bound when it is alone on its SIMD (the kernel with the
// MFMAs removed takes as long as with them), two per SIMD hide each other's waits.
template <int WA, int WB, bool TF, bool DSA, bool RAG, int DIAG = 0, int NP = 4>
__global__ __launch_bounds__(64 * (WA * WB + NP)) void wgrad_pw_kernel(WgPwDesc d, OutDesc dc) {
    constexpr int NC = WA * WB, RA = 64 * WA, RB = 64 * WB;
    constexpr int OSA = RA + 8, OSB = RB + 8;          // u32x4 per (plane, octet) slab; +128 B: octet o lands 32 banks on
    constexpr int BUF = 12 * (OSA + OSB);              // u32x4 per stage buffer: [A: 3 planes x 4 octets | B: same]
    constexpr int PR = 8 * NP;                         // rows covered by the producers per item (8 lanes per row)
    constexpr int IA = RA / PR, IB = RB / PR;          // staging items (row, 4 pixels) per producer thread and stage
    extern __shared__ __align__(16) float lds[];
    u32x4* const L0 = (u32x4*)lds;

    // The tiles of one split-K slice read the SAME pixels of dy and x (each tile its own rows): they must share an L2.
    // Hardware deals block b to XCD b % 8, so slice z = 8 * (k / tiles) + b % 8 and tile = k % tiles with k = b / 8: the
    // tiles of a slice are consecutive blocks of one XCD and every operand panel comes from HBM once, not once per tile
    // (512->256 @28, four tiles: 462 -> 231 MB per launch).  The grid is padded to a multiple of 8 slices.
    // With few slices (not a multiple of 8) the tiles of a slice are cut into `ngroup` runs and a (slice, run) pair is
    // the unit dealt to an XCD, so that all eight XCDs have work (splitk_xcd_map, conv_common.h).
    const int mt = (d.Cout + RA - 1) / RA, nt = (d.Cin + RB - 1) / RB;
    int tile, z;
    if (!splitk_xcd_map(blockIdx.x, mt * nt, d.nsplit, d.ngroup, tile, z)) return;   // (the whole workgroup: before any barrier)
    const int i0 = (tile % mt) * RA, j0 = (tile / mt) * RB;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int sbeg = z * d.spz, send = min(sbeg + d.spz, d.NS);
    const int nst = send > sbeg ? send - sbeg : 0;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;

    if (wave >= NC) {
        // ------------------------------------------------------------ producers
        const int p = threadIdx.x - 64 * NC;           // 0 .. 64 NP - 1
        const int q = p & 7, rb = p >> 3;              // pixel quad of the stage, first row
        const int oct = q >> 1, half = q & 1;
        const __amdgpu_buffer_rsrc_t rsa = make_rsrc(d.dy, d.ndy), rsb = make_rsrc(d.x, d.nx);
        const __amdgpu_buffer_rsrc_t rsa2 = make_rsrc(DSA ? d.dy2 : d.dy, DSA ? d.ndy : 0);
        int arow_off[IA], brow_off[IB];                // byte offset of (image 0, row, pixel 0), or OOB for rows past the end
        float aca[DSA ? IA : 1], acb[DSA ? IA : 1], acc_[DSA ? IA : 1];
        float bsc[TF ? IB : 1], bsh[TF ? IB : 1];
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            const int row = i0 + rb + PR * i;
            arow_off[i] = row < d.Cout ? row * d.HW * 4 : OOB;
            if constexpr (DSA) {
                const int r = row < d.Cout ? row : 0;
                aca[i] = d.coef[r]; acb[i] = d.coef[d.Cout + r]; acc_[i] = d.coef[2 * d.Cout + r];
            }
        }
#pragma unroll
        for (int i = 0; i < IB; ++i) {
            const int row = j0 + rb + PR * i;
            brow_off[i] = row < d.Cin ? row * d.HW * 4 : OOB;
            if constexpr (TF) {
                const int r = row < d.Cin ? row : 0;
                bsc[i] = d.scale[r]; bsh[i] = d.shift[r];
            }
        }
        const int aimg = d.Cout * d.HW * 4, bimg = d.Cin * d.HW * 4;     // bytes per image
        float araw[2][IA][4], braw[2][IB][4];
        float araw2[DSA ? 2 : 1][DSA ? IA : 1][4];
        int cnt[2] = {0, 0};                           // live pixels of this thread's quad, per register set
        auto load_stage = [&](int s, auto set_tag) {
            constexpr int Q = decltype(set_tag)::value;
            const int u = 4 * s + oct;
            const bool in = s < send && u < d.U;
            const uint32_t uu = in ? (uint32_t)u : 0u;
            const uint32_t n = d.dNO.div(uu);
            const int r0 = 8 * (int)(uu - n * (uint32_t)d.NO) + 4 * half;
            const int c = in ? max(0, min(4, d.HW - r0)) : 0;
            cnt[Q] = c;
            const int pix = r0 * 4;
            auto quad = [&](__amdgpu_buffer_rsrc_t rs, int row_off, int img, float (&v)[4]) {
                if constexpr (DIAG & 1) {
                    v[0] = __int_as_float(row_off + c); v[1] = 1.f; v[2] = __int_as_float(pix); v[3] = 2.f;
                } else if constexpr (!RAG) {
                    const int off = (c > 0 && row_off != OOB) ? (int)n * img + row_off + pix : OOB;
                    const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
                    v[0] = __uint_as_float(t.x); v[1] = __uint_as_float(t.y);
                    v[2] = __uint_as_float(t.z); v[3] = __uint_as_float(t.w);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int off = (e < c && row_off != OOB) ? (int)n * img + row_off + pix + 4 * e : OOB;
                        v[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
                    }
                }
            };
#pragma unroll
            for (int i = 0; i < IA; ++i) {
                quad(rsa, arow_off[i], aimg, araw[Q][i]);
                if constexpr (DSA) quad(rsa2, arow_off[i], aimg, araw2[Q][i]);
            }
#pragma unroll
            for (int i = 0; i < IB; ++i) quad(rsb, brow_off[i], bimg, braw[Q][i]);
        };
        const float relu_lo = d.relu ? 0.f : -__builtin_inff();      // ReLU as a lower bound: one v_max, no select
        // [plane][octet][row] x 16 bytes; this thread's 8 bytes are half `half` of (octet `oct`, row)
        auto store_stage = [&](u32x4* buf, auto set_tag) {
            constexpr int Q = decltype(set_tag)::value;
            u32x2_t* const A2 = (u32x2_t*)(buf + oct * OSA) + half;
            u32x2_t* const B2 = (u32x2_t*)(buf + 12 * OSA + oct * OSB) + half;
            auto body = [&](auto whole_tag) {
                constexpr bool WHOLE = decltype(whole_tag)::value;
                const int c = cnt[Q];
#pragma unroll
                for (int i = 0; i < IA; ++i) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = araw[Q][i][e];
                        if constexpr (DSA) t = fmaf(aca[i], t, fmaf(acb[i], araw2[Q][i][e], acc_[i]));
                        v[e] = (WHOLE || !DSA || e < c) ? t : 0.f;      // (without DSA a dead pixel was loaded as 0)
                    }
                    if constexpr (DSA) {
                        if (arow_off[i] == OOB) { v[0] = v[1] = v[2] = v[3] = 0.f; }
                    }
                    uint32_t h0, m0, l0, h1, m1, l1;
                    if constexpr (DIAG & 2) {
                        h0 = __float_as_uint(v[0]); m0 = __float_as_uint(v[1]); l0 = h0 ^ m0;
                        h1 = __float_as_uint(v[2]); m1 = __float_as_uint(v[3]); l1 = h1 ^ m1;
                    } else {
                        split3(v[0], v[1], h0, m0, l0);
                        split3(v[2], v[3], h1, m1, l1);
                    }
                    u32x2_t* o = A2 + (rb + PR * i) * 2;
                    if constexpr (DIAG & 16) {
                        if (h0 == 0x12345u && m1 == 0x54321u) o[0] = u32x2_t{l0, l1};
                    } else {
                    o[0] = u32x2_t{h0, h1};
                    o[4 * OSA * 2] = u32x2_t{m0, m1};
                    o[8 * OSA * 2] = u32x2_t{l0, l1};
                    }
                }
#pragma unroll
                for (int i = 0; i < IB; ++i) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = braw[Q][i][e];
                        if constexpr (TF) t = fmaxf(fmaf(t, bsc[i], bsh[i]), relu_lo);
                        v[e] = (WHOLE || !TF || e < c) ? t : 0.f;
                    }
                    if constexpr (TF) {
                        if (brow_off[i] == OOB) { v[0] = v[1] = v[2] = v[3] = 0.f; }
                    }
                    uint32_t h0, m0, l0, h1, m1, l1;
                    if constexpr (DIAG & 2) {
                        h0 = __float_as_uint(v[0]); m0 = __float_as_uint(v[1]); l0 = h0 ^ m0;
                        h1 = __float_as_uint(v[2]); m1 = __float_as_uint(v[3]); l1 = h1 ^ m1;
                    } else {
                        split3(v[0], v[1], h0, m0, l0);
                        split3(v[2], v[3], h1, m1, l1);
                    }
                    u32x2_t* o = B2 + (rb + PR * i) * 2;
                    if constexpr (DIAG & 16) {
                        if (h0 == 0x12345u && m1 == 0x54321u) o[0] = u32x2_t{l0, l1};
                    } else {
                    o[0] = u32x2_t{h0, h1};
                    o[4 * OSB * 2] = u32x2_t{m0, m1};
                    o[8 * OSB * 2] = u32x2_t{l0, l1};
                    }
                }
            };
            // almost every stage is whole (4 live pixels in every quad): a wave-uniform test picks the body without selects
            if (__builtin_amdgcn_ballot_w64(cnt[Q] != 4) == 0) body(std::true_type{});
            else body(std::false_type{});
        };
        // stage sbeg + k lives in register set k & 1 and LDS buffer k & 1
        load_stage(sbeg, S0{});
        load_stage(sbeg + 1, S1{});
        store_stage(L0, S0{});
        __syncthreads();
        unsigned long long tw = 0, tb0 = 0;                    // DIAG & 32: cycles parked at the stage barrier / loop start
        auto iter = [&](int k, auto par_tag) {
            constexpr int P = decltype(par_tag)::value;        // k & 1
            load_stage(sbeg + k + 2, par_tag);                 // its set was written to LDS an iteration ago
            store_stage(L0 + (P ^ 1) * BUF, std::integral_constant<int, P ^ 1>{});   // stage k+1, loaded an iteration ago
            if constexpr (DIAG & 32) {
                const unsigned long long t0 = __builtin_amdgcn_s_memtime();
                __syncthreads();
                tw += __builtin_amdgcn_s_memtime() - t0;
            } else {
                __syncthreads();
            }
        };
        if constexpr (DIAG & 32) tb0 = __builtin_amdgcn_s_memtime();
        for (int k = 0; k < nst; k += 2) {
            iter(k, S0{});
            if (k + 1 < nst) iter(k + 1, S1{});
        }
        if constexpr (DIAG & 32) {
            const unsigned long long tot = __builtin_amdgcn_s_memtime() - tb0;
            __syncthreads();                                   // (pairs with the consumers' barrier behind their epilogue)
            if (lane == 0) {
                unsigned long long* o = (unsigned long long*)(dc.p + (int64_t)z * dc.sz + (int64_t)(i0 + wave) * dc.si + j0);
                o[0] = tot; o[1] = tw;
            }
        }
        return;
    }

    // ---------------------------------------------------------------- consumers: wavefront (wa, wb), 64 x 64 outputs
    const int wa = wave / WB, wb = wave % WB;
    const int l31 = lane & 31, lh = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int afrag = lh * OSA + wa * 64 + l31, bfrag = 12 * OSA + lh * OSB + wb * 64 + l31;
    auto lda = [&](u32x4 (&f)[3], const u32x4* cur, int t, int blk) {
        if constexpr (DIAG & 8) {
#pragma unroll
            for (int p = 0; p < 3; ++p) f[p] = u32x4{(uint32_t)(t + blk), 0x3f803f80u, (uint32_t)p, 0x3f803f80u};
            return;
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) f[p] = cur[afrag + (p * 4 + 2 * t) * OSA + 32 * blk];
    };
    auto ldb = [&](u32x4 (&f)[3], const u32x4* cur, int t, int blk) {
        if constexpr (DIAG & 8) {
#pragma unroll
            for (int p = 0; p < 3; ++p) f[p] = u32x4{(uint32_t)(t * 2 + blk), 0x3f803f80u, (uint32_t)p, 0x3f803f80u};
            return;
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) f[p] = cur[bfrag + (p * 4 + 2 * t) * OSB + 32 * blk];
    };
    auto mm = [&](const u32x4 (&a)[3], const u32x4 (&b)[3], f32x16 c) {
        if constexpr (DIAG & 4) {
            c[0] += __uint_as_float(a[0].x ^ b[0].y ^ a[1].z ^ b[1].w ^ a[2].x ^ b[2].y);
            return c;
        } else {
            return mfma_split(a, b, c);
        }
    };
    __syncthreads();
    u32x4 a0[3], a1[3], b0[3], b1[3];
    unsigned long long tw = 0, tb0 = 0;
    if constexpr (DIAG & 32) tb0 = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < nst; ++k) {
        const u32x4* cur = L0 + (k & 1) * BUF;
        lda(a0, cur, 0, 0);
        ldb(b0, cur, 0, 0);
        ldb(b1, cur, 0, 1);
        lda(a1, cur, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = mm(a0, b0, acc[0][0]);
        acc[0][1] = mm(a0, b1, acc[0][1]);
        __builtin_amdgcn_sched_barrier(0);
        lda(a0, cur, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc[1][0] = mm(a1, b0, acc[1][0]);
        __builtin_amdgcn_sched_barrier(0);
        ldb(b0, cur, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc[1][1] = mm(a1, b1, acc[1][1]);
        __builtin_amdgcn_sched_barrier(0);
        ldb(b1, cur, 1, 1);
        lda(a1, cur, 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = mm(a0, b0, acc[0][0]);
        acc[0][1] = mm(a0, b1, acc[0][1]);
        acc[1][0] = mm(a1, b0, acc[1][0]);
        acc[1][1] = mm(a1, b1, acc[1][1]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (DIAG & 32) {
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            __syncthreads();
            tw += __builtin_amdgcn_s_memtime() - t0;
        } else {
            __syncthreads();
        }
    }
    store_tile<2, 2, RA, RB, WA, WB>(acc, dc, d.Cout, d.Cin, i0, j0, z);
    if constexpr (DIAG & 32) {      // every wavefront leaves (loop cycles, cycles parked at the barrier) in row i0 + wave
        const unsigned long long tot = __builtin_amdgcn_s_memtime() - tb0;
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (lane == 0) {
            unsigned long long* o = (unsigned long long*)(dc.p + (int64_t)z * dc.sz + (int64_t)(i0 + wave) * dc.si + j0);
            o[0] = tot; o[1] = tw;
        }
    }
}

static int wg_pw_np() {        // SCAT_WGPW_NP=8: eight producer wavefronts with the eight-consumer tiles (1024 threads)
    static const int m = diag_env_int("SCAT_WGPW_NP", 4);
    return m;
}

template <int WA, int WB, bool TF, bool DSA, bool RAG, int DIAG = 0, int NP = 4>
static void launch_wg_pw(const WgPwDesc& d, const OutDesc& dc, int splits, hipStream_t st) {
    if constexpr (NP == 4 && DIAG == 0 && WA * WB == 8 && !DSA) {
        if (wg_pw_np() == 8) return launch_wg_pw<WA, WB, TF, DSA, RAG, 0, 8>(d, dc, splits, st);
    }
    constexpr int RA = 64 * WA, RB = 64 * WB;
    const int mt = cdiv(d.Cout, RA), nt = cdiv(d.Cin, RB);
    constexpr size_t lds_bytes = (size_t)2 * 12 * (RA + 8 + RB + 8) * 16;
#ifdef SCAT_DIAG
    if constexpr (DIAG == 0 && !TF && !DSA && !RAG && WA * WB == 8) {     // ablation variants: SCAT_TUNE = 200 + DIAG
        switch (tuning() - 200) {
        case 1: return launch_wg_pw<WA, WB, TF, DSA, RAG, 1>(d, dc, splits, st);
        case 2: return launch_wg_pw<WA, WB, TF, DSA, RAG, 2>(d, dc, splits, st);
        case 3: return launch_wg_pw<WA, WB, TF, DSA, RAG, 3>(d, dc, splits, st);
        case 4: return launch_wg_pw<WA, WB, TF, DSA, RAG, 4>(d, dc, splits, st);
        case 8: return launch_wg_pw<WA, WB, TF, DSA, RAG, 8>(d, dc, splits, st);
        case 12: return launch_wg_pw<WA, WB, TF, DSA, RAG, 12>(d, dc, splits, st);
        case 16: return launch_wg_pw<WA, WB, TF, DSA, RAG, 16>(d, dc, splits, st);
        case 19: return launch_wg_pw<WA, WB, TF, DSA, RAG, 19>(d, dc, splits, st);
        case 32: return launch_wg_pw<WA, WB, TF, DSA, RAG, 32>(d, dc, splits, st);
        default: break;
        }
    }
#endif
    auto kern = wgrad_pw_kernel<WA, WB, TF, DSA, RAG, DIAG, NP>;
    static bool once = (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)lds_bytes) == hipSuccess);
    (void)once;
    hipLaunchKernelGGL(kern, dim3(splitk_xcd_grid(mt * nt, splits, d.ngroup)), dim3(64 * (WA * WB + NP)), lds_bytes, st, d, dc);
}

template <int WA, int WB, bool DSA>
static void launch_wg_pw_v(bool tf, bool rag, const WgPwDesc& d, const OutDesc& dc, int splits, hipStream_t st) {
    if constexpr (DSA) {           // the folded BatchNorm backward exists for H >= 28 planes only: never ragged
        if (tf) launch_wg_pw<WA, WB, true, true, false>(d, dc, splits, st);
        else launch_wg_pw<WA, WB, false, true, false>(d, dc, splits, st);
    } else {
        if (tf) { if (rag) launch_wg_pw<WA, WB, true, false, true>(d, dc, splits, st); else launch_wg_pw<WA, WB, true, false, false>(d, dc, splits, st); }
        else { if (rag) launch_wg_pw<WA, WB, false, false, true>(d, dc, splits, st); else launch_wg_pw<WA, WB, false, false, false>(d, dc, splits, st); }
    }
}

static int wg_pw_mode() {       // SCAT_WG_PW=0: the first-generation kernels everywhere (A/B runs)
    static const int m = diag_env_int("SCAT_WG_PW", 1);
    return m;
}

WgPwPlan wgrad_pw_plan(int B, int Cin, int Cout, int HW, bool dsa, const void* dy, const void* x) {
    WgPwPlan p{};
    p.ok = false;
    if (!wg_pw_mode() || Cout % 64 || Cin % 64 || (int64_t)Cout * Cin < 128 * 128) return p;
    if (((uintptr_t)dy & 15) || ((uintptr_t)x & 15)) return p;
    if (dsa && HW % 4) return p;
    // tile shape: eight consumers on 256 x 128 / 128 x 256 where the layer has that many rows and columns; the folded
    // BatchNorm backward stages two tensors for its rows and stays on four consumers (register budget)
    if (!dsa && Cout >= 256 && Cin >= 128 && (Cout >= Cin || Cin < 256)) { p.wa = 4; p.wb = 2; }
    else if (!dsa && Cin >= 256 && Cout >= 128) { p.wa = 2; p.wb = 4; }
    else if (Cout >= 128 && Cin >= 128) { p.wa = 2; p.wb = 2; }
    else if (Cout >= 256) { p.wa = 4; p.wb = 1; }
    else if (Cin >= 256) { p.wa = 1; p.wb = 4; }
    else return p;
    const int NO = (HW + 7) / 8;
    p.stages = (B * NO + 3) / 4;
    const int tiles = cdiv(Cout, 64 * p.wa) * cdiv(Cin, 64 * p.wb);
    static const int forced = diag_env_int("SCAT_WGPW_TARGET", 0);
    const int target = forced > 0 ? forced : 256;      // one workgroup per CU
    int s = cdiv(target, tiles);
    const int smax = p.stages / 8 > 0 ? p.stages / 8 : 1;      // >= 8 stages (256 pixels) per slice
    if (s > smax) s = smax;
    if (s < 1) s = 1;
    p.spz = cdiv(p.stages, s);
    p.splits = cdiv(p.stages, p.spz);
    p.ok = true;
    return p;
}

void wgrad_pw_launch(const WgPwPlan& p, const float* dy, const float* x, float* out, int B, int Cin, int HW, int Cout,
                     const float* in_scale, const float* in_shift, int in_relu, hipStream_t st, const float* dy2,
                     const float* coef3) {
    WgPwDesc d{};
    d.dy = dy; d.x = x; d.scale = in_scale; d.shift = in_shift; d.dy2 = dy2; d.coef = coef3;
    d.relu = in_scale ? in_relu : 0;
    d.Cout = Cout; d.Cin = Cin; d.HW = HW;
    d.NO = (HW + 7) / 8; d.U = B * d.NO; d.NS = p.stages; d.spz = p.spz; d.nsplit = p.splits;
    d.ngroup = splitk_xcd_groups(p.splits);
    d.dNO = FastDiv::make(d.NO);
    d.ndy = (int64_t)B * Cout * HW; d.nx = (int64_t)B * Cin * HW;
    OutDesc dc{};
    dc.p = out; dc.mode = 0; dc.si = Cin; dc.sj = 1; dc.sz = (int64_t)Cout * Cin; dc.I = Cout; dc.J = Cin;
    dc.n = (int64_t)Cout * Cin;
    const bool tf = in_scale != nullptr, rag = HW % 4 != 0;
    set_kernel_label("wgrad1x1_pw_%dx%dx32%s%s%s_split%d", 64 * p.wa, 64 * p.wb, tf ? "_tf" : "", dy2 ? "_bnb" : "",
                     rag ? "_rag" : "", p.splits);
    if (dy2) {
        if (p.wa == 2 && p.wb == 2) launch_wg_pw_v<2, 2, true>(tf, false, d, dc, p.splits, st);
        else if (p.wa == 4) launch_wg_pw_v<4, 1, true>(tf, false, d, dc, p.splits, st);
        else launch_wg_pw_v<1, 4, true>(tf, false, d, dc, p.splits, st);
        return;
    }
    if (p.wa == 4 && p.wb == 2) launch_wg_pw_v<4, 2, false>(tf, rag, d, dc, p.splits, st);
    else if (p.wa == 2 && p.wb == 4) launch_wg_pw_v<2, 4, false>(tf, rag, d, dc, p.splits, st);
    else if (p.wa == 2 && p.wb == 2) launch_wg_pw_v<2, 2, false>(tf, rag, d, dc, p.splits, st);
    else if (p.wa == 4) launch_wg_pw_v<4, 1, false>(tf, rag, d, dc, p.splits, st);
    else launch_wg_pw_v<1, 4, false>(tf, rag, d, dc, p.splits, st);
}

}  // namespace scat

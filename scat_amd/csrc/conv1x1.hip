// Pointwise (1x1, stride 1) convolution forward / data-gradient:  dst[n][m][r] = sum_c A[m][c] * f(src[n][c][r]).
//
// Same operand split as conv3x3.hip, which measured ~15 % faster than staging both operands through LDS:
//  * A (weights [M][K], K contiguous, L2-resident) goes straight from global memory to the MFMA operand
//    registers.  The contraction order inside a 16-channel sub-chunk is free, so lane (row, h) takes
//    k = 8h .. 8h+7: 32 contiguous bytes, two 16-B loads, prefetched one sub-chunk ahead.
//  * B (activations) is staged through LDS 32 channels at a time in k-pair-interleaved rows
//    [k/2][pixel][k&1]: a staging thread owns a channel PAIR x 4 pixels (two 16-B global loads -> two
//    ds_write_b128) and a lane reads its k pair with one ds_read_b64.  The fused BatchNorm+ReLU of the
//    producing layer is applied once per element on the way into LDS.
//  * one barrier per 32 channels (64 MFMAs per wave at 128x128); fragments of the second sub-chunk are
//    read while the first one computes.
//
// Replaces nn.Conv2d(k=1) and its input gradient as called at models/resnet.py:65-67,70-72,84-92 of the
// reference (conv1/conv3 of every Bottleneck) and the 1x1 reductions of models/hand_net.py, fp32.
#include <atomic>

#include "conv_common.h"
#include "split.h"

namespace scat {

struct PwDesc {
    const float* src;     // [Nimg][C][HW]
    const float* w;       // [M][C]
    const float* scale;   // optional fused input transform, per source channel
    const float* shift;
    int relu;
    int C, M, HW, npix;   // HW: source plane; npix: columns of the contraction = B * OHW
    FastDiv dHW;
    int64_t nsrc, nw;
    // split kernel only — taps: source (y, x) of column pixel (oy, ox) and tap (th, tw) is
    // (oy*a + th*tb + c0y, ox*a + tw*tb + c0x); a 1x1/stride-1 conv has ntap = 1, a = 1, OHW = HW
    int ntap, KWt, a, tb, c0y, c0x, H, W, OW, OHW;
    FastDiv dOHW, dOW;
    int variant;          // SCAT_TUNE pass-through for kernel-variant experiments
    const float* src2;    // dual-source (DS) kernels: second tensor and the [3][C] coefficient table
    const float* coef;
};

constexpr int PW_KS = 32;     // channels per LDS stage (two 16-channel sub-chunks)


template <int BM, int BN, bool V4, bool TF>
__global__ __launch_bounds__(NT, (BM * BN >= 128 * 128 ? 2 : 3)) void conv1x1_kernel(PwDesc d, OutDesc dc) {
    constexpr int MI = BM / 64, NI = BN / 64;
    constexpr int PV = V4 ? 4 : 1;                    // pixels per staging item
    constexpr int PT = BN / PV;                       // staging threads along the pixel dimension
    constexpr int NIT = 16 * PT / NT;                 // staging items (channel pair x PV pixels) per thread
    static_assert(NIT >= 1, "tile too small for 256 staging threads");
    extern __shared__ __align__(16) float lds[];      // B[2][16 k-pairs][BN][2]
    auto Bs = [&](int buf) -> float* { return lds + buf * (PW_KS * BN); };

    const int mt = (d.M + BM - 1) / BM, nt = (d.npix + BN - 1) / BN;
    const int tile = xcd_remap(blockIdx.x, mt * nt);
    const int i0 = (tile % mt) * BM, j0 = (tile / mt) * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const int nstage = (d.C + PW_KS - 1) / PW_KS;

    // ---- activation staging: item r of this thread = channel pair p = tid/PT + r*(NT/PT), pixels (tid%PT)*PV ..
    const __amdgpu_buffer_rsrc_t rsrc_b = make_rsrc(d.src, d.nsrc);
    const __amdgpu_buffer_rsrc_t rsrc_sc = make_rsrc(TF ? d.scale : d.src, TF ? d.C : 0);
    const __amdgpu_buffer_rsrc_t rsrc_sh = make_rsrc(TF ? d.shift : d.src, TF ? d.C : 0);
    const int pcol = (tid % PT) * PV, prow = tid / PT;
    const int chw4 = d.HW * 4;
    int boff;                                          // byte offset of (n, channel 2*prow, r) or OOB
    bool bok;
    {
        const int j = j0 + pcol;
        bok = j < d.npix;
        const uint32_t jj = bok ? (uint32_t)j : 0u;
        const uint32_t n = d.dHW.div(jj);
        boff = bok ? (int)((n * (uint32_t)d.C * (uint32_t)d.HW + (jj - n * (uint32_t)d.HW)) * 4u) + 2 * prow * chw4 : OOB;
    }
    float bst[NIT][2][PV];
    f32x2 tsc[TF ? NIT : 1], tsh[TF ? NIT : 1];
    auto load_b = [&](int c0) {                        // channels >= C read 0 (also the prefetch past the end)
#pragma unroll
        for (int r = 0; r < NIT; ++r) {
            const int c = c0 + 2 * (prow + r * (NT / PT));
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int vo = c + u < d.C ? boff : OOB;
                const int so = (c0 + 2 * r * (NT / PT) + u) * chw4;
                if constexpr (V4) {
                    u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, vo, so, 0);
                    bst[r][u][0] = __uint_as_float(t.x); bst[r][u][1] = __uint_as_float(t.y);
                    bst[r][u][2] = __uint_as_float(t.z); bst[r][u][3] = __uint_as_float(t.w);
                } else {
                    bst[r][u][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc_b, vo, so, 0));
                }
            }
            if constexpr (TF) {                        // scale/shift of the pair (c even, C even: 8-B load)
                const int vo = c < d.C ? c * 4 : OOB;
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                u32x2 a = __builtin_amdgcn_raw_buffer_load_b64(rsrc_sc, vo, 0, 0);
                u32x2 b = __builtin_amdgcn_raw_buffer_load_b64(rsrc_sh, vo, 0, 0);
                tsc[r] = f32x2{__uint_as_float(a.x), __uint_as_float(a.y)};
                tsh[r] = f32x2{__uint_as_float(b.x), __uint_as_float(b.y)};
            }
        }
    };
    auto store_b = [&](float* dst) {
#pragma unroll
        for (int r = 0; r < NIT; ++r) {
            float x[2][PV];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int q = 0; q < PV; ++q) {
                    float t = bst[r][u][q];
                    if constexpr (TF) {
                        t = fmaf(t, tsc[r][u], tsh[r][u]);
                        t = d.relu ? fmaxf(t, 0.f) : t;
                        t = bok ? t : 0.f;             // out-of-range pixels / channels (scale, shift read 0 there)
                    }
                    x[u][q] = t;
                }
            float* p = dst + ((prow + r * (NT / PT)) * BN + pcol) * 2;
            if constexpr (V4) {
                *(f32x4*)p = f32x4{x[0][0], x[1][0], x[0][1], x[1][1]};
                *(f32x4*)(p + 4) = f32x4{x[0][2], x[1][2], x[0][3], x[1][3]};
            } else {
                *(f32x2*)p = f32x2{x[0][0], x[1][0]};
            }
        }
    };

    // ---- weights: lane (row, h) holds k = 8h .. 8h+7 of its rows, straight from global memory
    const __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc(d.w, d.nw);
    int aoff[MI];
#pragma unroll
    for (int a = 0; a < MI; ++a) {
        const int row = i0 + wm * (BM / 2) + a * 32 + l31;
        aoff[a] = row < d.M ? (row * d.C + lh * 8) * 4 : OOB;
    }
    auto load_a = [&](float (&dst)[MI][8], int c) {   // c >= C: zeros
#pragma unroll
        for (int a = 0; a < MI; ++a) {
            const int vo = c < d.C ? aoff[a] : OOB;
            u32x4 t0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, vo, c * 4, 0);
            u32x4 t1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, vo + 16, c * 4, 0);
            dst[a][0] = __uint_as_float(t0.x); dst[a][1] = __uint_as_float(t0.y);
            dst[a][2] = __uint_as_float(t0.z); dst[a][3] = __uint_as_float(t0.w);
            dst[a][4] = __uint_as_float(t1.x); dst[a][5] = __uint_as_float(t1.y);
            dst[a][6] = __uint_as_float(t1.z); dst[a][7] = __uint_as_float(t1.w);
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // B fragments of sub-chunk t: NI pixels x 4 k-pairs, 8 bytes each (rows t*8 + 4h + jp)
    const int b_frag = lh * 4 * BN + wn * (BN / 2) + l31;
    auto read_b = [&](f32x2 (&dst)[NI][4], const float* buf, int t) {
        const f32x2* p = (const f32x2*)buf + t * 8 * BN + b_frag;
#pragma unroll
        for (int b = 0; b < NI; ++b)
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) dst[b][jp] = p[jp * BN + b * 32];
    };
    float areg[2][MI][8];
    f32x2 breg[2][NI][4];
    auto mfmas = [&](int set) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int a = 0; a < MI; ++a)
#pragma unroll
                for (int b = 0; b < NI; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[set][a][j], breg[set][b][j >> 1][j & 1],
                                                                     acc[a][b], 0, 0, 0);
    };

    load_b(0);
    load_a(areg[0], 0);
    store_b(Bs(0));
    __syncthreads();
    read_b(breg[0], Bs(0), 0);

    for (int s = 0; s < nstage; ++s) {
        const int c0 = s * PW_KS;
        const float* bcur = Bs(s & 1);
        float* bnext = Bs((s + 1) & 1);
        // sub-chunk 0: start the next stage's activations and this stage's second half
        load_b(c0 + PW_KS);
        load_a(areg[1], c0 + 16);
        read_b(breg[1], bcur, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0);
        __builtin_amdgcn_sched_barrier(0);
        // sub-chunk 1
        load_a(areg[0], c0 + PW_KS);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(1);
        __builtin_amdgcn_sched_barrier(0);
        store_b(bnext);
        __syncthreads();
        read_b(breg[0], bnext, 0);
    }
    store_tile<MI, NI, BM, BN, 2, 2>(acc, dc, d.M, d.npix, i0, j0, 0);
}

// ---------------------------------------------------------------- split-operand variant (see conv3x3.hip)
//
// fp32 products as six bf16 MFMA terms of three-way split operands, fp32 accumulation.  The activations are split
// once per element on the way into LDS ([plane][k-octet][pixel][8 bf16], one ds_read_b128 per plane and
// fragment); the weights are loaded as fp32 straight into registers (lane (row, h): k = 8h..8h+7, 32 bytes) and
// split there, in the shadow of the previous chunk's MFMAs.
// DS ("dual source"): the operand is a BatchNorm-backward output that was never materialised:
//   value = ca[c]*src + cb[c]*src2 + cc[c]   (src = masked incoming gradient g, src2 = the layer's raw conv output,
//   coef = [ca | cb | cc] from scat_bn_bwd_pre).  One register set, activations one stage ahead.
// STEM: the 7x7 / stride-2 / pad-3 convolution of a 3-channel image (models/resnet.py:105, the first layer) on the same
// loop.  The contraction index is (kh, c, kw) with kw padded 7 -> 8: k-octet o = 3 kh + c holds the 8 CONSECUTIVE input
// pixels (2 oy + kh - 3, 2 ox - 3 .. 2 ox + 4) of channel c, the weight of the eighth is zero; 21 octets padded to 24
// (d.C = 192): 168 / 147 = 1.14 x the MFMA work of the exact contraction instead of 16 / 3 = 5.3 x for a channel-padded
// taps layout.  Only the staging addresses differ from the pointwise case.
// PL ("plain": one tap, C % 32 == 0 — every pointwise layer of the networks): no per-load channel bound, and a pixel
// column past the end of the tensor is left to hold whatever the transform makes of a zero (columns are independent and
// the epilogue never stores it), so the staging VALU carries no masks at all.  Beside a busy matrix pipe a SIMD issues
// about one vector instruction per MFMA (tools/mfma_probe.hip, PROBE_PC=1): every instruction taken out of the staging
// is time given back to the MFMAs.
//
// pw_split_tile: the contraction stages [s_begin, s_end) of one output tile (a whole tile for the one-tile-per-workgroup
// kernel; a K-segment of it for the persistent stream-K kernel below).  `post(acc)` runs on the finished accumulators
// and says whether the epilogue (store_tile) follows.
template <int WM, int BN, bool TF, bool DS, bool STEM, bool PL, class Post, bool BNB = false>
__device__ __forceinline__ void pw_split_tile(const PwDesc& d, const OutDesc& dc, const int tile, const int s_begin,
                                              const int s_end, Post post) {
    static_assert(!(TF && DS), "one input transform at a time");
    static_assert(!(PL && STEM), "the stem has its own staging");
    static_assert(!STEM || (!TF && !DS), "the stem reads the raw image");
    constexpr int BM = 32 * WM, WN = 4 / WM, NI = BN / (32 * WN);
    constexpr int NIT = 4 * BN / NT;                  // k-octets staged per thread per 32-channel stage
    static_assert(NIT >= 1, "tile too small");
    extern __shared__ __align__(16) float lds[];      // B[2][3 planes][4 k-octets][BN] x 16 bytes
    auto Bs = [&](int buf) -> u32x4* { return (u32x4*)lds + buf * (12 * BN); };

    const int mt = (d.M + BM - 1) / BM;
    const int i0 = (tile % mt) * BM, j0 = (tile / mt) * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lh = lane >> 5;
    const int nsc = (d.C + PW_KS - 1) / PW_KS;       // 32-channel stages per tap
    const int nstage = s_end;                          // contraction order: (tap, channel); loads past s_end read zeros

    // ---- activation staging: this thread's pixel, k-octets g0 + r*(NT/BN)
    const __amdgpu_buffer_rsrc_t rsrc_b = make_rsrc(d.src, d.nsrc);
    const int pcol = tid % BN, g0 = tid / BN;
    const int chw4 = d.HW * 4;
    int boff;                                          // byte offset of (n, channel 8*g0, tap (0,0)); only used under pmask
    uint32_t pmask = 0;                                // taps that fall inside the source plane for this pixel
    bool bok;
    {
        const int j = j0 + pcol;
        bok = j < d.npix;
        const uint32_t jj = bok ? (uint32_t)j : 0u;
        const uint32_t n = d.dOHW.div(jj);
        const uint32_t r = jj - n * (uint32_t)d.OHW;
        const int oy = (int)d.dOW.div(r), ox = (int)r - oy * d.OW;
        const int sy0 = oy * d.a + d.c0y, sx0 = ox * d.a + d.c0x;
        boff = ((int)n * d.C * d.HW + sy0 * d.W + sx0) * 4 + 8 * g0 * chw4;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int th = t / d.KWt, tw = t - th * d.KWt;
            const bool ok = bok && t < d.ntap && (unsigned)(sy0 + th * d.tb) < (unsigned)d.H &&
                            (unsigned)(sx0 + tw * d.tb) < (unsigned)d.W;
            pmask |= (ok ? 1u : 0u) << t;
        }
    }
    // two register sets: the activations of stage s+2 are requested at the top of stage s and written to LDS at the
    // end of stage s+1 (one stage of cover is not enough for an HBM miss at 2 waves per SIMD)
    float bst[DS ? 1 : 2][NIT][8];
    float bst2[DS ? NIT : 1][8];
    const __amdgpu_buffer_rsrc_t rsrc_b2 = make_rsrc(DS ? d.src2 : d.src, DS ? d.nsrc : 0);
    // STEM: (image, oy, ox) of this thread's pixel
    int st_n = 0, st_oy = 0, st_ox = 0;
    if constexpr (STEM) {
        const uint32_t jj = bok ? (uint32_t)(j0 + pcol) : 0u;
        const uint32_t n = d.dOHW.div(jj);
        const uint32_t r = jj - n * (uint32_t)d.OHW;
        st_n = (int)n; st_oy = (int)d.dOW.div(r); st_ox = (int)r - st_oy * d.OW;
    }
    auto load_b = [&](int st, auto set_tag) {         // st >= nstage: every lane reads 0
        constexpr int Q = decltype(set_tag)::value;
        if constexpr (STEM) {
#pragma unroll
            for (int r = 0; r < NIT; ++r) {
                const int o = 4 * st + g0 + r * (NT / BN);          // k-octet = (kh, c)
                const int kh = o / 3, c = o - 3 * kh;
                const int iy = 2 * st_oy + kh - 3, ix0 = 2 * st_ox - 3;
                const bool rowok = bok && o < 21 && st < nstage && (unsigned)iy < (unsigned)d.H;
                const int base = (((st_n * 3 + c) * d.H + iy) * d.W + ix0) * 4;
#pragma unroll
                for (int m = 0; m < 8; ++m)
                    bst[Q][r][m] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                        rsrc_b, rowok && (unsigned)(ix0 + m) < (unsigned)d.W ? base + 4 * m : OOB, 0, 0));
            }
            return;
        }
        const int tap = st / nsc, c0 = (st - tap * nsc) * PW_KS;
        const int th = tap / d.KWt, tw = tap - th * d.KWt;
        const int vbase = ((pmask >> (tap < 9 ? tap : 9)) & 1u) && st < nstage ? boff + (th * d.W + tw) * d.tb * 4 : OOB;
#pragma unroll
        for (int r = 0; r < NIT; ++r) {
            const int c = c0 + 8 * (g0 + r * (NT / BN));
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                bst[Q][r][m] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                    rsrc_b, PL || c + m < d.C ? vbase : OOB, (c0 + 8 * r * (NT / BN) + m) * chw4, 0));
                if constexpr (DS)
                    bst2[r][m] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                        rsrc_b2, PL || c + m < d.C ? vbase : OOB, (c0 + 8 * r * (NT / BN) + m) * chw4, 0));
            }
        }
    };
    // a staging thread's k-octet is the same for its whole wavefront (BN >= 64): the fused transform's constants are
    // scalar loads
    const int g0u = __builtin_amdgcn_readfirstlane(g0);
    const float relu_lo = d.relu ? 0.f : -__builtin_inff();     // ReLU as a lower bound: one v_max, no select
    auto store_b = [&](int st, u32x4* dst, auto set_tag) {
        constexpr int Q = decltype(set_tag)::value;
        const int tap = st / nsc, c0 = (st - tap * nsc) * PW_KS;
        const bool live = (pmask >> (tap < 9 ? tap : 9)) & 1u;   // padding must stay zero AFTER the transform
#pragma unroll
        for (int r = 0; r < NIT; ++r) {
            u32x4 hi, mid, lo;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float x[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int m = 2 * q + u;
                    float t = bst[Q][r][m];
                    if constexpr (TF) {
                        const int c = c0 + 8 * (g0u + r * (NT / BN)) + m;
                        const int cc = PL || c < d.C ? c : 0;
                        t = fmaxf(fmaf(t, d.scale[cc], d.shift[cc]), relu_lo);
                        if constexpr (!PL) t = live ? t : 0.f;
                    }
                    if constexpr (DS) {
                        const int c = c0 + 8 * (g0u + r * (NT / BN)) + m;
                        const int cc = PL || c < d.C ? c : 0;
                        t = fmaf(d.coef[cc], t, fmaf(d.coef[d.C + cc], bst2[r][m], d.coef[2 * d.C + cc]));
                        if constexpr (!PL) t = (live && c < d.C) ? t : 0.f;
                    }
                    x[u] = t;
                }
                uint32_t h, mm, l;
                split3(x[0], x[1], h, mm, l);
                hi[q] = h; mid[q] = mm; lo[q] = l;
            }
            const int g = g0 + r * (NT / BN);
            dst[(0 * 4 + g) * BN + pcol] = hi;
            dst[(1 * 4 + g) * BN + pcol] = mid;
            dst[(2 * 4 + g) * BN + pcol] = lo;
        }
    };

    // the same work in 2*NI slices, one per MFMA group of a stage: slice i splits PP element pairs and writes an
    // octet's three fragments once its fourth pair is done (fragments live in shi/smid/slo across slices)
    constexpr int PP = NIT * 4 / (2 * NI) > 0 ? NIT * 4 / (2 * NI) : 1;
    uint32_t shi[2], smid[2], slo[2];                  // one half octet: written as 8 bytes per plane
    auto store_slice = [&](int st, u32x4* dst, auto set_tag, auto idx_tag) {
        constexpr int Q = decltype(set_tag)::value, IDX = decltype(idx_tag)::value;
        const int tap = st / nsc, c0 = (st - tap * nsc) * PW_KS;
        const bool live = (pmask >> (tap < 9 ? tap : 9)) & 1u;
        static_for<PP>([&](auto j_tag) {
            constexpr int pp = IDX * PP + decltype(j_tag)::value;
            if constexpr (pp < NIT * 4) {
            constexpr int r = pp >> 2, q = pp & 3;
            float x[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int m = 2 * q + u;
                float t = bst[Q][r][m];
                if constexpr (TF) {
                    const int c = c0 + 8 * (g0u + r * (NT / BN)) + m;
                    const int cc = PL || c < d.C ? c : 0;
                    t = fmaxf(fmaf(t, d.scale[cc], d.shift[cc]), relu_lo);
                    if constexpr (!PL) t = live ? t : 0.f;
                }
                x[u] = t;
            }
            uint32_t h, mm, l;
            split3(x[0], x[1], h, mm, l);
            shi[q & 1] = h; smid[q & 1] = mm; slo[q & 1] = l;
            if constexpr ((q & 1) == 1) {
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                const int g = g0 + r * (NT / BN);
                u32x2* p = (u32x2*)(dst + (0 * 4 + g) * BN + pcol) + (q >> 1);
                p[0] = u32x2{shi[0], shi[1]};
                p[(size_t)4 * BN * 2] = u32x2{smid[0], smid[1]};
                p[(size_t)8 * BN * 2] = u32x2{slo[0], slo[1]};
            }
            }
        });
    };
    // ---- weights: pre-split planes ws[chunk][plane][row][16 bf16]; lane (row, h) takes 16 bytes per plane
    const __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc(d.w, d.nw);
    const int row = i0 + wm * 32 + l31;
    const int aoff = row < d.M ? row * 32 + lh * 16 : OOB;
    const int aplane = d.M * 32;                       // bytes per (chunk, plane) slab
    const int nchunk = (d.C + 15) / 16;
    // sub-chunk q = 2*stage + t of the (tap, channel) order -> slab (tap*nchunk + chunk); a stage's second sub-chunk
    // may lie past the channels (C % 32 == 16), the sub-chunk after the last stage always does: zeros
    auto load_a = [&](u32x4 (&dst)[3], int q) {
        const int st = q >> 1, tap = st / nsc, ch = (st - tap * nsc) * 2 + (q & 1);
        const int vo = (ch < nchunk && st < nstage) ? aoff : OOB;
#pragma unroll
        for (int p = 0; p < 3; ++p)
            dst[p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, vo, ((tap * nchunk + ch) * 3 + p) * aplane, 0);
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

    u32x4 areg[2][3];
    u32x4 bfr[2][3];

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    // diag build only (SCAT_TUNE=77, tools/pw_stamp.py): thread 0 overwrites the tile's first 8 outputs with time stamps
    const bool stamp = kDiag && d.variant == 77;
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
    if (stamp) ts0 = __builtin_amdgcn_s_memrealtime();
    if constexpr (DS) {
        load_b(s_begin, S0{});
        load_a(areg[0], 2 * s_begin);
        store_b(s_begin, Bs(0), S0{});
        __syncthreads();
        read_b(bfr[0], Bs(0), 0, 0);
        for (int s = s_begin; s < nstage; ++s) {
            const u32x4* bcur = Bs((s - s_begin) & 1);
            u32x4* bnext = Bs((s - s_begin + 1) & 1);
            load_b(s + 1, S0{});
            static_for<2 * NI>([&](auto i_tag) {
                constexpr int I = decltype(i_tag)::value, t = I / NI, b = I % NI;
                if constexpr (b == 0) load_a(areg[t ^ 1], 2 * s + t + 1);
                constexpr int fcur = I & 1, fnxt = fcur ^ 1;
                if constexpr (b + 1 < NI) read_b(bfr[fnxt], bcur, t, b + 1);
                else if constexpr (t == 0) read_b(bfr[fnxt], bcur, 1, 0);
                __builtin_amdgcn_sched_barrier(0);
                acc[0][b] = mfma_split(areg[t], bfr[fcur], acc[0][b]);
                __builtin_amdgcn_sched_barrier(0);
            });
            store_b(s + 1, bnext, S0{});
            __syncthreads();
            read_b(bfr[0], bnext, 0, 0);
        }
    } else {
        load_b(s_begin, S0{});
        load_b(s_begin + 1, S1{});
        load_a(areg[0], 2 * s_begin);
        store_b(s_begin, Bs(0), S0{});
        __syncthreads();
        read_b(bfr[0], Bs(0), 0, 0);
        if (stamp) ts1 = __builtin_amdgcn_s_memrealtime();

        // stage s: LDS buffer s & 1, register set s & 1 is free again (its data went to LDS one stage ago)
        auto stage = [&](int s, auto cur_tag) {
            constexpr int CUR = decltype(cur_tag)::value;
            const u32x4* bcur = Bs(CUR);
            u32x4* bnext = Bs(CUR ^ 1);
            load_b(s + 2, std::integral_constant<int, CUR>{});
                // the split of stage s+1 (its data landed a stage ago, its LDS buffer has been free since the last
                // barrier) rides in the shadow of this stage's MFMAs: one slice per MFMA group
                static_for<2 * NI>([&](auto i_tag) {
                    constexpr int I = decltype(i_tag)::value, t = I / NI, b = I % NI;
                    if constexpr (b == 0) load_a(areg[t ^ 1], 2 * s + t + 1);
                    constexpr int fcur = I & 1, fnxt = fcur ^ 1;
                    if constexpr (b + 1 < NI) read_b(bfr[fnxt], bcur, t, b + 1);
                    else if constexpr (t == 0) read_b(bfr[fnxt], bcur, 1, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    store_slice(s + 1, bnext, std::integral_constant<int, CUR ^ 1>{}, i_tag);
                    acc[0][b] = mfma_split(areg[t], bfr[fcur], acc[0][b]);
                    __builtin_amdgcn_sched_barrier(0);
                });
            __syncthreads();
            read_b(bfr[0], bnext, 0, 0);
        };
        for (int s = s_begin; s < nstage; s += 2) {
            stage(s, S0{});
            if (s + 1 < nstage) stage(s + 1, S1{});
        }
    }
    if (stamp) ts2 = __builtin_amdgcn_s_memrealtime();
    if (!post(acc)) return;
    store_tile<1, NI, BM, BN, WM, WN, BNB>(acc, dc, d.M, d.npix, i0, j0, 0);
    if (stamp) {
        __syncthreads();
        if (tid == 0 && dc.mode == 1) {
            __builtin_amdgcn_s_waitcnt(0);
            const uint32_t n = dc.dHW.div((uint32_t)j0);
            const int hw = j0 - (int)n * dc.HW;
            if (hw + 8 <= dc.HW && ((hw & 1) == 0)) {
                unsigned long long* o = (unsigned long long*)(dc.p + ((int64_t)n * dc.C + i0) * dc.HW + hw);
                o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = __builtin_amdgcn_s_memrealtime();
            }
        }
    }
}

// one tile per workgroup, XCD-aware tile order
template <int WM, int BN, bool TF, bool DS = false, bool STEM = false, bool PL = false, bool BNB = false>
__global__ __launch_bounds__(NT, 3) void conv1x1_split_kernel(PwDesc d, OutDesc dc) {
    constexpr int BM = 32 * WM;
    const int mt = (d.M + BM - 1) / BM, nt = (d.npix + BN - 1) / BN;
    const int tile = xcd_remap(blockIdx.x, mt * nt);
    const int nstage = d.ntap * ((d.C + PW_KS - 1) / PW_KS);
    if constexpr (kDiag) {
        // tools build, SCAT_TUNE = 300 + t: the workgroups of the second / third residency slot of a CU (block index / 256)
        // start t / 2t microseconds late, so that the co-resident tiles' prologues and epilogues stop coinciding
        if (d.variant >= 300 && d.variant < 400) {
            const int slot = (blockIdx.x >> 8) % 3;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();     // 100 MHz
            const unsigned long long wait = (unsigned long long)slot * (d.variant - 300) * 100ull;
            while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(32);
        }
    }
    auto all = [](auto&) { return true; };
    pw_split_tile<WM, BN, TF, DS, STEM, PL, decltype(all), BNB>(d, dc, tile, 0, nstage, all);
}

// ---------------------------------------------------------------- persistent stream-K schedule
//
// At batch 96 the tile counts of this network are multiples of 147 (96 * 49 * 2^k / 128): 588 or 1 176 tiles on the
// 768 workgroup slots of the chip (256 CUs x 3) — the last round of a one-tile-per-workgroup grid is half empty, and
// in-kernel time stamps (tools/pw_stamp.py) show the stage loop itself already saturates the matrix pipe.  Here the grid
// IS the slot count: the tiles x stages of a launch are one list of (tile, stage) units, cut into equal contiguous
// ranges, one per workgroup (Osama et al., "Stream-K", PPoPP'23).  A range is at most: the tail of a tile, whole tiles,
// the head of a tile.
//   * XCD-aware: hardware deals workgroup b to XCD b % 8.  Each XCD gets a contiguous chunk of whole tiles (as xcd_remap
//     does) and its G / 8 workgroups stream over that chunk only, so a split tile's two halves, its weights and its
//     activation panel stay in one L2.
//   * A workgroup walks its range from the END: the head segment of its last tile comes first — its accumulators go to
//     the workgroup's slot of the scratch buffer and a flag is raised — and the tail segment of its first tile comes
//     last: the workgroup that holds a tile's tail owns the tile, adds the partial sums of its predecessors (in
//     descending workgroup order: the result depends on the shape only, bit for bit reproducible) and runs the epilogue
//     (bias / accumulate / BatchNorm sums unchanged: they see the complete accumulators).
//   * An owner only ever waits for workgroups with a LOWER block index on the same XCD, whose partial was the first
//     thing they did: hardware dispatches blocks in index order, so the wait can not deadlock whatever else occupies
//     the GPU, and in practice never spins.  The spin is bounded all the same; exhaustion raises a sticky device-side
//     error word (scat_device_error) instead of hanging the queue.
//   * Partials and flags are written and read with agent-coherent (sc1) accesses: correct even if the two workgroups
//     did not share an L2, and no L2 write-back fence (buffer_wbl2) anywhere.
struct SkDesc {
    float* part;          // [G][BM * BN] accumulator images in fragment order
    uint32_t* flags;      // [G]: == id once the workgroup's partial is complete
    uint32_t* err;        // sticky error word
    uint32_t id;          // unique per launch (the scratch buffer is recycled between launches)
    int G;                // workgroups = slots, multiple of 8
    int coh;              // 1: partials through agent-coherent (sc1, write-through) accesses; 0: through the XCD's L2
};

template <int WM, int BN, bool TF, bool DS = false, bool PL = true>
__global__ __launch_bounds__(NT, 3) void conv1x1_sk_kernel(PwDesc d, OutDesc dc, SkDesc sk) {
    constexpr int BM = 32 * WM, WN = 4 / WM, NI = BN / (32 * WN);
    const int mt = (d.M + BM - 1) / BM, nt = (d.npix + BN - 1) / BN;
    const int T = mt * nt;
    const int nstage = d.ntap * ((d.C + PW_KS - 1) / PW_KS);
    const int x = blockIdx.x & 7, s = blockIdx.x >> 3, Gx = sk.G >> 3;
    const int q = T >> 3, r = T & 7;
    const int t0 = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;      // this XCD's tiles: [t0, t0 + tcount)
    const int tcount = q + (x < r ? 1 : 0);
    const int64_t U = (int64_t)tcount * nstage;
    auto ubeg = [&](int w) { return (int)((int64_t)w * U / Gx); };      // unit range of workgroup w of this XCD
    const int u0 = ubeg(s), u1 = ubeg(s + 1);
    const int tid = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rp = make_rsrc(sk.part, (int64_t)sk.G * BM * BN);
    constexpr int SC1 = 16;                                                // agent-coherent cache policy (gfx940+)

    for (int u = u1; u > u0;) {
        const int tl = (u - 1) / nstage;                                   // local tile of the range's last unit
        const int ua = tl * nstage;
        const int a = (u0 > ua ? u0 : ua) - ua, e = u - ua;                // stages [a, e) of that tile
        u = ua + a;
        pw_split_tile<WM, BN, TF, DS, false, PL>(d, dc, t0 + tl, a, e, [&](f32x16 (&acc)[1][NI]) {
            if (a == 0 && e == nstage) return true;                        // a whole tile: nothing to exchange
            if (e != nstage) {
                // head (or middle) segment: publish the partial sums
                const int slot = (x * Gx + s) * (BM * BN * 4);
#pragma unroll
                for (int b = 0; b < NI; ++b)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        u32x4 v = {__float_as_uint(acc[0][b][4 * k]), __float_as_uint(acc[0][b][4 * k + 1]),
                                   __float_as_uint(acc[0][b][4 * k + 2]), __float_as_uint(acc[0][b][4 * k + 3])};
                        if (sk.coh) __builtin_amdgcn_raw_buffer_store_b128(v, rp, ((b * 4 + k) * NT + tid) * 16, slot, SC1);
                        else __builtin_amdgcn_raw_buffer_store_b128(v, rp, ((b * 4 + k) * NT + tid) * 16, slot, 0);
                    }
                __builtin_amdgcn_s_waitcnt(0);                             // this thread's stores have been acknowledged
                __syncthreads();                                           // ... every thread's
                if (tid == 0) __hip_atomic_store(sk.flags + x * Gx + s, sk.id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
            // tail segment: this workgroup owns the tile — add the predecessors' partial sums, nearest first
            for (int w = s - 1; w >= 0; --w) {
                if (ubeg(w) == ubeg(w + 1)) continue;                      // an empty range publishes nothing
                if (tid == 0) {
                    int spins = 0;
                    while (__hip_atomic_load(sk.flags + x * Gx + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != sk.id) {
                        __builtin_amdgcn_s_sleep(8);
                        if (++spins > (1 << 22)) {                         // seconds: something is badly wrong
                            __hip_atomic_store(sk.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                    }
                }
                __syncthreads();
                const int slot = (x * Gx + w) * (BM * BN * 4);
#pragma unroll
                for (int b = 0; b < NI; ++b) {
                    u32x4 v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        v[k] = sk.coh ? __builtin_amdgcn_raw_buffer_load_b128(rp, ((b * 4 + k) * NT + tid) * 16, slot, SC1)
                                      : __builtin_amdgcn_raw_buffer_load_b128(rp, ((b * 4 + k) * NT + tid) * 16, slot, 1);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        acc[0][b][4 * k] += __uint_as_float(v[k].x);
                        acc[0][b][4 * k + 1] += __uint_as_float(v[k].y);
                        acc[0][b][4 * k + 2] += __uint_as_float(v[k].z);
                        acc[0][b][4 * k + 3] += __uint_as_float(v[k].w);
                    }
                }
                if (ubeg(w) <= ua) break;                                  // that one started the tile
            }
            return true;
        });
    }
}

// ---------------------------------------------------------------- producer / consumer wave specialisation
//
// The split kernel above makes every wavefront do everything: request activations, apply the fused transform, split
// them into bf16 terms, write LDS, read fragments, issue MFMAs.  Its instruction stream per 32-channel stage is
// ~230 VALU + ~250 SALU + 24 LDS reads + 22 loads around 48 MFMAs, and a wavefront issues in order: the matrix pipe
// idles while its owner splits (PMC: 0.34 busy at 2.3 wavefronts per SIMD, half of every wavefront's life spent
// waiting to issue).  Here a 512-thread workgroup has two kinds of wavefront:
//   * wavefronts 4-7, producers: global loads (three stages ahead) -> fused BatchNorm+ReLU -> three-way split ->
//     LDS (two stages ahead of the readers, three buffers).  No MFMA, no LDS reads.
//   * wavefronts 0-3, consumers: weights straight from L2 into operand registers (pre-split planes), activation
//     fragments from LDS, MFMAs.  No staging VALU; the first fragment of the next stage is read before the barrier
//     because that stage was published a barrier earlier.
// One s_barrier per stage joins the two streams.  A SIMD holds one consumer and one producer per workgroup, two
// workgroups per CU at MI = 1: the matrix pipe of a SIMD is fed by two wavefronts whose streams are MFMA + LDS
// reads only, the VALU by two whose streams never wait for the matrix pipe.
// DIAG (timing experiments only, results are wrong): 1 = no activation loads, 2 = no weight loads, 4 = no MFMAs,
// 8 = no output stores, 16 = no split / LDS writes
// W4 (pointwise stride-1 layers with HW % 4 == 0): column (b, c) of the tile is pixel 4c + b instead of 32b + c.  A
// producer lane then owns FOUR CONSECUTIVE pixels of four channels — four 16-byte loads per stage instead of sixteen
// 4-byte ones, LDS writes of 8 bytes that consecutive lanes place in consecutive slots — and a consumer lane's four
// accumulator columns are four consecutive pixels of one row: the epilogue stores 16 bytes per instruction instead of
// 4.  The MFMA does not care which pixel a column is; only the two ends of the kernel know.
// NCW consumer wavefronts (32 MI rows each) + 4 producers.  NCW = 8: a 256 x 128 tile staged once for twice the rows —
// half the activation traffic and producer work per MFMA, two consumers per SIMD inside ONE workgroup per CU.
// S16: the consumers issue v_mfma_f32_16x16x32_bf16 (a whole 32-channel stage per instruction, 16 x 16 tiles) instead of
// 32x32x16: the same fragments, planes and LDS image, addressed as lane (l & 15, k-octet l >> 4); the chip holds a
// higher clock on this shape (tools/mfma_probe.hip: +5 % with the LDS reads).
// RING = R > 0: no workgroup barrier in the main loop.  The stage buffers form a ring of R slots with two LDS words per
// slot: FULL (each of the 4 producer wavefronts adds 1 once its part of the stage is written) and FREE (each of the NCW
// consumers adds 1 once it has read the stage).  A producer runs ahead until the ring is full, a consumer waits only for
// data that is not there: the two streams are coupled by data, not by a rendezvous every stage (with a barrier per
// stage, a late load stalls the MFMA wavefronts even when the fragments they need next are already in LDS).  Spins are
// bounded (the kernel ends with wrong results rather than hanging should a count be wrong).
template <int MI, bool TF, int DIAG = 0, bool W4 = false, int NCW = 4, bool S16 = false, int RING = 0>
__global__ __launch_bounds__((NCW + 4) * 64, (NCW == 8 ? 3 : (MI == 1 ? 4 : 2))) void conv1x1_pc_kernel(PwDesc d, OutDesc dc) {
    constexpr int BN = 128, BM = 32 * NCW * MI, NI = 4, PT = 256;
    static_assert(RING == 0 || (!W4 && !S16 && RING >= 3), "ring form: plain columns, 32x32x16");
    constexpr int NIT = 4 * BN / PT;                  // k-octets per producer thread per 32-channel stage
    constexpr int BUF = 12 * BN;                      // u32x4 per buffer: [3 planes][4 k-octets][BN]
    extern __shared__ __align__(16) float lds[];
    u32x4* const B0 = (u32x4*)lds;
    // ring form: FULL[R] | FREE[R] behind the R stage buffers
    unsigned* const ring_full = (unsigned*)(B0 + (RING > 0 ? RING : 3) * BUF);
    unsigned* const ring_free = ring_full + (RING > 0 ? RING : 0);
    if constexpr (RING > 0) {
        if (threadIdx.x < 2 * RING) ring_full[threadIdx.x] = 0u;
        __syncthreads();
    }
    // (relaxed LDS accesses + compiler barriers: the LDS serves one wavefront's requests in order, so "writes, then
    // lgkmcnt(0), then the count" and "poll, then reads" need no hardware fence; a release/acquire pair would make hipcc
    // drain vmcnt too and with it the producers' prefetched global loads)
    auto wait_ge = [&](unsigned* p, unsigned target) {
        for (int it = 0; it < (1 << 16); ++it) {
            if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= target) break;
            __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
    };
    auto signal = [&](unsigned* p) {
        asm volatile("" ::: "memory");
        if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };

    const int mt = (d.M + BM - 1) / BM, nt = (d.npix + BN - 1) / BN;
    const int tile = xcd_remap(blockIdx.x, mt * nt);
    const int i0 = (tile % mt) * BM, j0 = (tile / mt) * BN;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int nsc = (d.C + PW_KS - 1) / PW_KS;
    const int nstage = d.ntap * nsc;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;

    if constexpr (W4) {
      if (wave >= NCW) {
        // ------------------------------------------------------------ producer, wide form: wavefront = k-octet,
        // lane = (pixel quad p, channel half h)
        const __amdgpu_buffer_rsrc_t rsrc_b = make_rsrc(d.src, d.nsrc);
        const __amdgpu_buffer_rsrc_t rsrc_sc = make_rsrc(TF ? d.scale : d.src, TF ? d.C : 0);
        const __amdgpu_buffer_rsrc_t rsrc_sh = make_rsrc(TF ? d.shift : d.src, TF ? d.C : 0);
        const int g = wave - NCW, p = lane >> 1, h = lane & 1;
        const int chw4 = d.HW * 4;
        int boff;                                      // byte offset of (image, channel 8g + 4h, first pixel of the quad)
        {
            const int j = j0 + 4 * p;
            const bool bok = j < d.npix;
            const uint32_t jj = bok ? (uint32_t)j : 0u;
            const uint32_t n = d.dHW.div(jj);
            boff = bok ? (int)((n * (uint32_t)d.C * (uint32_t)d.HW + (jj - n * (uint32_t)d.HW)) * 4u) +
                             (8 * g + 4 * h) * chw4 : OOB;
        }
        u32x4 bst[2][4];                               // [register set][channel m] = 4 pixels
        f32x4 tsc[2], tsh[2];
        auto load_b = [&](int st, auto set_tag) {     // st >= nstage: zeros
            constexpr int Q = decltype(set_tag)::value;
            const int c0 = st * PW_KS;
            const int vo = (c0 + 8 * g < d.C && st < nstage) ? boff : OOB;      // C % 16 == 0: whole octets
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if constexpr (DIAG & 1) bst[Q][m] = u32x4{(uint32_t)vo, 0x3f800000u, (uint32_t)m, 0x40000000u};
                else bst[Q][m] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, vo, (c0 + m) * chw4, 0);
            }
            if constexpr (TF) {
                const int co = (c0 + 8 * g + 4 * h < d.C && st < nstage) ? (c0 + 8 * g + 4 * h) * 4 : OOB;
                tsc[Q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_sc, co, 0, 0));
                tsh[Q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_sh, co, 0, 0));
            }
        };
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        auto store_b = [&](int st, u32x4* dst, auto set_tag) {
            constexpr int Q = decltype(set_tag)::value;
            u32x2* base = (u32x2*)(dst + g * BN + p) + h;             // plane 0, position p (pixel 4p), half h
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float x[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    float t = __uint_as_float(bst[Q][m][k]);
                    if constexpr (TF) {
                        t = fmaf(t, tsc[Q][m], tsh[Q][m]);          // (channels >= C: scale = shift = 0 -> 0)
                        t = d.relu ? fmaxf(t, 0.f) : t;
                    }
                    x[m] = t;
                }
                if constexpr (DIAG & 16) {
                    asm volatile("" ::"v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]));
                } else {
                    uint32_t h0, m0, l0, h1, m1, l1;
                    split3(x[0], x[1], h0, m0, l0);
                    split3(x[2], x[3], h1, m1, l1);
                    u32x2* q = base + (size_t)k * 32 * 2;              // position k*32 + p
                    q[0] = u32x2{h0, h1};
                    q[(size_t)4 * BN * 2] = u32x2{m0, m1};
                    q[(size_t)8 * BN * 2] = u32x2{l0, l1};
                }
            }
        };
        load_b(0, S0{});
        load_b(1, S1{});
        store_b(0, B0, S0{});
        load_b(2, S0{});
        store_b(1, B0 + BUF, S1{});
        __syncthreads();
        int wb = 2;
        auto iter = [&](int s, auto par_tag) {
            constexpr int P = decltype(par_tag)::value;
            load_b(s + 3, std::integral_constant<int, P ^ 1>{});
            store_b(s + 2, B0 + wb * BUF, par_tag);
            wb = wb == 2 ? 0 : wb + 1;
            __syncthreads();
        };
        for (int s = 0; s < nstage; s += 2) {
            iter(s, S0{});
            if (s + 1 < nstage) iter(s + 1, S1{});
        }
        return;
      }
    } else
    if (wave >= NCW) {
        // ------------------------------------------------------------ producer
        const int ptid = threadIdx.x - 64 * NCW;
        const __amdgpu_buffer_rsrc_t rsrc_b = make_rsrc(d.src, d.nsrc);
        const int pcol = ptid % BN, g0 = ptid / BN;
        const int g0u = __builtin_amdgcn_readfirstlane(g0);
        const int chw4 = d.HW * 4;
        int boff;
        uint32_t pmask = 0;
        {
            const int j = j0 + pcol;
            const bool bok = j < d.npix;
            const uint32_t jj = bok ? (uint32_t)j : 0u;
            const uint32_t n = d.dOHW.div(jj);
            const uint32_t r = jj - n * (uint32_t)d.OHW;
            const int oy = (int)d.dOW.div(r), ox = (int)r - oy * d.OW;
            const int sy0 = oy * d.a + d.c0y, sx0 = ox * d.a + d.c0x;
            boff = ((int)n * d.C * d.HW + sy0 * d.W + sx0) * 4 + 8 * g0 * chw4;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int th = t / d.KWt, tw = t - th * d.KWt;
                const bool ok = bok && t < d.ntap && (unsigned)(sy0 + th * d.tb) < (unsigned)d.H &&
                                (unsigned)(sx0 + tw * d.tb) < (unsigned)d.W;
                pmask |= (ok ? 1u : 0u) << t;
            }
        }
        float bst[2][NIT][8];
        auto load_b = [&](int st, auto set_tag) {     // st >= nstage: every lane reads 0
            constexpr int Q = decltype(set_tag)::value;
            const int tap = st / nsc, c0 = (st - tap * nsc) * PW_KS;
            const int th = tap / d.KWt, tw = tap - th * d.KWt;
            const int vbase = ((pmask >> (tap < 9 ? tap : 9)) & 1u) && st < nstage ? boff + (th * d.W + tw) * d.tb * 4 : OOB;
#pragma unroll
            for (int r = 0; r < NIT; ++r) {
                const int cu = c0 + 8 * (g0u + r * (PT / BN));          // wave-uniform; C % 16 == 0: whole octets
                const int vo = cu < d.C ? vbase : OOB;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    if constexpr (DIAG & 1) bst[Q][r][m] = __int_as_float(vo + m);
                    else
                    bst[Q][r][m] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                        rsrc_b, vo, (c0 + 8 * r * (PT / BN) + m) * chw4, 0));
                }
            }
        };
        auto store_b = [&](int st, u32x4* dst, auto set_tag) {
            constexpr int Q = decltype(set_tag)::value;
            const int tap = st / nsc, c0 = (st - tap * nsc) * PW_KS;
            const bool live = (pmask >> (tap < 9 ? tap : 9)) & 1u;       // padding must stay zero AFTER the transform
            if constexpr (DIAG & 16) {
#pragma unroll
                for (int r = 0; r < NIT; ++r)
#pragma unroll
                    for (int m = 0; m < 8; ++m) asm volatile("" ::"v"(bst[Q][r][m]));
                return;
            }
#pragma unroll
            for (int r = 0; r < NIT; ++r) {
                float x[8];
                const int cu = c0 + 8 * (g0u + r * (PT / BN));
                const int cc = cu < d.C ? cu : 0;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    float t = bst[Q][r][m];
                    if constexpr (TF) {
                        t = fmaf(t, d.scale[cc + m], d.shift[cc + m]);
                        t = d.relu ? fmaxf(t, 0.f) : t;
                        t = (live && cu < d.C) ? t : 0.f;
                    }
                    x[m] = t;
                }
                u32x4 hi, mid, lo;
                split3x8(x, hi, mid, lo);
                const int g = g0 + r * (PT / BN);
                dst[(0 * 4 + g) * BN + pcol] = hi;
                dst[(1 * 4 + g) * BN + pcol] = mid;
                dst[(2 * 4 + g) * BN + pcol] = lo;
            }
        };
        if constexpr (RING > 0) {
            load_b(0, S0{});
            load_b(1, S1{});
            int slot = 0, use = 0;
            auto put = [&](int ps, auto par_tag) {
                if (use > 0) wait_ge(ring_free + slot, (unsigned)(NCW * use));     // every consumer has read its last use
                store_b(ps, B0 + slot * BUF, par_tag);
                load_b(ps + 2, par_tag);                                           // (the set is free again)
                __builtin_amdgcn_s_waitcnt(0xc07f);                                // lgkmcnt(0): the writes have landed
                signal(ring_full + slot);
                if (++slot == RING) { slot = 0; ++use; }
            };
            for (int ps = 0; ps < nstage; ps += 2) {
                put(ps, S0{});
                if (ps + 1 < nstage) put(ps + 1, S1{});
            }
            return;
        }
        load_b(0, S0{});
        load_b(1, S1{});
        store_b(0, B0, S0{});
        load_b(2, S0{});
        store_b(1, B0 + BUF, S1{});
        __syncthreads();
        int wb = 2;                                    // buffer of stage s + 2
        auto iter = [&](int s, auto par_tag) {
            constexpr int P = decltype(par_tag)::value;
            load_b(s + 3, std::integral_constant<int, P ^ 1>{});
            store_b(s + 2, B0 + wb * BUF, par_tag);
            wb = wb == 2 ? 0 : wb + 1;
            __syncthreads();
        };
        for (int s = 0; s < nstage; s += 2) {
            iter(s, S0{});
            if (s + 1 < nstage) iter(s + 1, S1{});
        }
        return;
    }

    if constexpr (S16) {
        // ------------------------------------------------------------ consumer, 16x16x32 form (MI = 1, plain columns)
        typedef float f32x4v __attribute__((ext_vector_type(4)));
        const int l15 = lane & 15, kq = lane >> 4;
        const __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc(d.w, d.nw);
        const int aplane = d.M * 32;
        const int nchunk = (d.C + 15) / 16;
        int aoff[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int row = i0 + wave * 32 + rb * 16 + l15;
            aoff[rb] = row < d.M ? row * 32 + (kq & 1) * 16 + (kq >> 1) * 3 * aplane : OOB;
        }
        auto load_a = [&](u32x4 (&dst)[3], int st, int rb) {     // the 16 rows rb of stage st: 32 channels per lane group
            const int tap = st / nsc, ch = (st - tap * nsc) * 2;
            const bool ok = st < nstage && ch + (kq >> 1) < nchunk;
#pragma unroll
            for (int p = 0; p < 3; ++p)
                dst[p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, ok ? aoff[rb] : OOB,
                                                               ((tap * nchunk + ch) * 3 + p) * aplane, 0);
        };
        f32x4v acc[2][8];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) acc[rb][cb] = f32x4v{0.f, 0.f, 0.f, 0.f};
        const int b_frag = kq * BN + l15;
        auto read_b = [&](u32x4 (&dst)[3], const u32x4* buf, int cb) {
            const u32x4* p = buf + b_frag + cb * 16;
#pragma unroll
            for (int q = 0; q < 3; ++q) dst[q] = p[q * 4 * BN];
        };
        auto mfma6 = [&](const u32x4 (&a)[3], const u32x4 (&b)[3], f32x4v c) {
            const bf16x8 ah = __builtin_bit_cast(bf16x8, a[0]), am = __builtin_bit_cast(bf16x8, a[1]),
                         al = __builtin_bit_cast(bf16x8, a[2]);
            const bf16x8 bh = __builtin_bit_cast(bf16x8, b[0]), bm = __builtin_bit_cast(bf16x8, b[1]),
                         bl = __builtin_bit_cast(bf16x8, b[2]);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
            return c;
        };
        // the two 16-row halves of a stage run one after the other (fragments re-read per half: 48 reads per stage),
        // so a half's weights are dead after its 8 column blocks and the next stage's load over them has half a stage
        // of cover: one register set per half
        u32x4 areg[2][3];
        u32x4 bfr[2][3];
        load_a(areg[0], 0, 0);
        load_a(areg[1], 0, 1);
        __syncthreads();
        read_b(bfr[0], B0, 0);
        int rb_ = 0;
        for (int s = 0; s < nstage; ++s) {
            const u32x4* bcur = B0 + rb_ * BUF;
            rb_ = rb_ == 2 ? 0 : rb_ + 1;
            const u32x4* bnxt = B0 + rb_ * BUF;
            static_for<16>([&](auto i_tag) {
                constexpr int I = decltype(i_tag)::value, rb = I / 8, cb = I % 8;
                constexpr int fcur = I & 1, fnxt = fcur ^ 1;
                if constexpr (I == 8) load_a(areg[0], s + 1, 0);       // half 0 is done with its weights
                if constexpr (I < 15) read_b(bfr[fnxt], bcur, (I + 1) % 8);
                else read_b(bfr[fnxt], bnxt, 0);                        // published one barrier ago
                __builtin_amdgcn_sched_barrier(0);
                acc[rb][cb] = mfma6(areg[rb], bfr[fcur], acc[rb][cb]);
                __builtin_amdgcn_sched_barrier(0);
            });
            load_a(areg[1], s + 1, 1);
            __syncthreads();
        }
        // C/D map of the 16x16 MFMA: col = lane & 15, row = 4 (lane >> 4) + reg
        const __amdgpu_buffer_rsrc_t rc = make_rsrc(dc.p, dc.n);
        const bool biasi = dc.bias && dc.bias_mode == 1;
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) {
            const int j = j0 + cb * 16 + l15;
            const bool colok = j < d.npix;
            const uint32_t jj = colok ? (uint32_t)j : 0u;
            const uint32_t n = dc.dHW.div(jj);
            const int coloff = (int)n * dc.C * dc.HW + (int)(jj - n * (uint32_t)dc.HW);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = i0 + wave * 32 + rb * 16 + kq * 4 + r;
                    const int vo = (colok && i < d.M) ? (coloff + i * dc.HW) * 4 : OOB;
                    float v = acc[rb][cb][r] + (biasi ? dc.bias[i < d.M ? i : 0] : 0.f);
                    if (dc.accumulate) v += bload(rc, vo);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rc, vo, 0, 0);
                }
        }
        return;
    }
    // ---------------------------------------------------------------- consumer: rows i0 + wave*32*MI ..
    const int l31 = lane & 31, lh = lane >> 5;
    const __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc(d.w, d.nw);
    int aoff[MI];
#pragma unroll
    for (int a = 0; a < MI; ++a) {
        const int row = i0 + (wave * MI + a) * 32 + l31;
        aoff[a] = row < d.M ? row * 32 + lh * 16 : OOB;
    }
    const int aplane = d.M * 32;
    const int nchunk = (d.C + 15) / 16;
    auto load_a = [&](u32x4 (&dst)[MI][3], int q) {
        const int st = q >> 1, tap = st / nsc, ch = (st - tap * nsc) * 2 + (q & 1);
        const bool ok = ch < nchunk && st < nstage;
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                if constexpr (DIAG & 2) { dst[a][p] = u32x4{(uint32_t)q, 0x3f803f80u, (uint32_t)aoff[a], 0x3f803f80u}; }
                else
                dst[a][p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, ok ? aoff[a] : OOB,
                                                                  ((tap * nchunk + ch) * 3 + p) * aplane, 0);
            }
    };
    f32x16 acc[MI][NI];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int b_frag = lh * BN + l31;
    auto read_b = [&](u32x4 (&dst)[3], const u32x4* buf, int t, int b) {
        const u32x4* p = buf + 2 * t * BN + b_frag + b * 32;
#pragma unroll
        for (int q = 0; q < 3; ++q) dst[q] = p[q * 4 * BN];
    };
    u32x4 areg[2][MI][3];
    u32x4 bfr[2][3];
    load_a(areg[0], 0);
    if constexpr (RING > 0) wait_ge(ring_full, 4u);
    else __syncthreads();
    read_b(bfr[0], B0, 0, 0);
    int rb = 0, ruse = 0;
    constexpr int NSLOT = RING > 0 ? RING : 3;
    for (int s = 0; s < nstage; ++s) {
        const u32x4* bcur = B0 + rb * BUF;
        unsigned* const fcur_free = ring_free + rb;
        if (++rb == NSLOT) { rb = 0; ++ruse; }
        const u32x4* bnxt = B0 + rb * BUF;
        static_for<2 * NI>([&](auto i_tag) {
            constexpr int I = decltype(i_tag)::value, t = I / NI, b = I % NI;
            if constexpr (b == 0) load_a(areg[t ^ 1], 2 * s + t + 1);
            constexpr int fcur = I & 1, fnxt = fcur ^ 1;
            if constexpr (RING > 0 && I == 2 * NI - 1) {
                if (s + 1 < nstage) wait_ge(ring_full + rb, 4u * (unsigned)(ruse + 1));      // the next stage is in LDS
            }
            if constexpr (b + 1 < NI) read_b(bfr[fnxt], bcur, t, b + 1);
            else if constexpr (t == 0) read_b(bfr[fnxt], bcur, 1, 0);
            else read_b(bfr[fnxt], bnxt, 0, 0);          // published one barrier ago
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < MI; ++a) {
                if constexpr (DIAG & 4) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        asm volatile("" ::"v"(areg[t][a][p]));
                        asm volatile("" ::"v"(bfr[fcur][p]));
                    }
                } else
                acc[a][b] = mfma_split(areg[t][a], bfr[fcur], acc[a][b]);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (RING > 0) signal(fcur_free);       // (LDS serves a wavefront's requests in order: its reads of
        else __syncthreads();                             //  this slot are behind it)
    }
    if constexpr (DIAG & 8) {
        if (d.variant != 0x7fffffff) return;          // (never true: keeps the accumulators live)
    }
    if constexpr (W4) {
        // lane (l31, lh): rows (reg & 3) + 8 (reg >> 2) + 4 lh of its 32-row block, pixels j0 + 4 l31 .. + 3 = the four
        // accumulator tiles' column l31: one 16-byte store per row (NCHW, whole quads inside one image)
        const __amdgpu_buffer_rsrc_t rc = make_rsrc(dc.p, dc.n);
        const int j = j0 + 4 * l31;
        const bool colok = j < d.npix;
        const uint32_t jj = colok ? (uint32_t)j : 0u;
        const uint32_t n = dc.dHW.div(jj);
        const int coloff = (int)n * dc.C * dc.HW + (int)(jj - n * (uint32_t)dc.HW);
        const bool biasi = dc.bias && dc.bias_mode == 1;
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + (wave * MI + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const bool ok = colok && i < d.M;
                const int vo = ok ? (coloff + i * dc.HW) * 4 : OOB;
                const float bi = biasi ? dc.bias[i < d.M ? i : 0] : 0.f;
                f32x4 v = f32x4{acc[a][0][r] + bi, acc[a][1][r] + bi, acc[a][2][r] + bi, acc[a][3][r] + bi};
                if (dc.accumulate) v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rc, vo, 0, 0));
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rc, vo, 0, 0);
            }
        return;
    }
    store_tile<MI, NI, BM, BN, NCW, 1>(acc, dc, d.M, d.npix, i0, j0, 0);
}

// ws[tap][ch][plane][i][16 bf16]: the three bf16 terms of element (i, c, tap).  w is the conv weight
// [Cout][Cin][KH][KW]; forward: (i, c) = (co, ci); transposed (data gradient): (i, c) = (ci, co).  Tap t of the ntap
// listed ones is (kh0 + ts*(t / KWt), kw0 + ts*(t % KWt)).  (WPrepJob in conv_common.h.)
// One work item = one k-octet of one row: 8 source weights -> three 16-byte stores (hi | mid | lo planes), the stores of
// a wavefront contiguous (item index = (r, i, half) with half fastest).  The first version made one element per thread
// with three 2-byte stores: 207 us per step for the 106 convolutions of ResNet-50 against ~60 us of HBM time.
__device__ __forceinline__ void wprep_octet(const WPrepJob& j, int64_t e) {
    const int nchunk = (j.C + 15) / 16;
    const int half = e & 1;
    int64_t r = e >> 1;
    const int i = r % j.M;
    r /= j.M;                                            // r = tap * nchunk + ch
    const int ch = r % nchunk, t = r / nchunk;
    const int kh = j.kh0 + j.ts * (t / j.KWt), kw = j.kw0 + j.ts * (t % j.KWt);
    const int c0 = ch * 16 + half * 8;
    const int64_t tap = (int64_t)kh * j.KW + kw, kk = (int64_t)j.KH * j.KW;
    float v[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int c = c0 + m;
        const int64_t pair = j.transposed ? (int64_t)c * j.M + i : (int64_t)i * j.C + c;
        v[m] = c < j.C ? j.w[pair * kk + tap] : 0.f;
    }
    u32x4 hi, mid, lo;
    split3x8(v, hi, mid, lo);
    u32x4* dst = (u32x4*)j.dst + ((r * 3) * j.M + i) * 2 + half;      // 16-byte units
    dst[0] = hi;
    dst[(int64_t)j.M * 2] = mid;
    dst[(int64_t)j.M * 4] = lo;
}
static inline int64_t wprep_elements(const WPrepJob& j) { return (int64_t)j.ntap * ((j.C + 15) / 16) * j.M * 2; }

__global__ __launch_bounds__(256) void w_taps_split_kernel(WPrepJob j) {
    const int64_t n = (int64_t)j.ntap * ((j.C + 15) / 16) * j.M * 2;
    for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < n; e += gridDim.x * 256ll) wprep_octet(j, e);
}

// the same for a table of jobs (device memory, sorted by blk0): block b belongs to the last job with blk0 <= b
__global__ __launch_bounds__(256) void w_prep_batch_kernel(const WPrepJob* __restrict__ jobs, int njobs) {
    int lo = 0, hi = njobs - 1;
    const int b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].blk0 <= b) lo = mid;
        else hi = mid - 1;
    }
    const WPrepJob j = jobs[lo];
    const int64_t n = (int64_t)j.ntap * ((j.C + 15) / 16) * j.M * 2;
    const int64_t e = (int64_t)(b - j.blk0) * 256 + threadIdx.x;
    if (e < n) wprep_octet(j, e);
}

WPrepJob wprep_job(const float* w, void* dst, int M, int C, int transposed, int KH, int KW, int ntap, int KWt, int kh0,
                   int kw0, int ts) {
    WPrepJob j{};
    j.w = w; j.dst = (uint16_t*)dst; j.M = M; j.C = C; j.transposed = transposed; j.KH = KH; j.KW = KW; j.ntap = ntap;
    j.KWt = KWt; j.kh0 = kh0; j.kw0 = kw0; j.ts = ts;
    j.blk0 = 0;
    j.nblk = (int)((wprep_elements(j) + 255) / 256);
    return j;
}

void wprep_batch_launch(const WPrepJob* jobs_dev, int njobs, int nblocks, hipStream_t st) {
    hipLaunchKernelGGL(w_prep_batch_kernel, dim3(nblocks), dim3(256), 0, st, jobs_dev, njobs);
}

void wprep_launch(const WPrepJob& j, hipStream_t st) {
    hipLaunchKernelGGL(w_taps_split_kernel, dim3(j.nblk < 2048 ? j.nblk : 2048), dim3(256), 0, st, j);
}

// fp32 transpose for the fp32-MFMA twin: ws[i][c] = w[c][i]
__global__ void w1x1_t_kernel(const float* __restrict__ w, float* __restrict__ ws, int M, int C) {
    const int64_t n = (int64_t)M * C;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = e % C, i = e / C;
        ws[e] = w[(int64_t)c * M + i];
    }
}

static int pw_plain_mode() {
    static const int m = diag_env_int("SCAT_PW_PLAIN", 1);
    return m;
}

// stream-K scratch armed by the caller for the next pointwise launch of this host thread (scat_streamk_arm)
struct SkScratch { void* buf; int64_t bytes; };
static thread_local SkScratch g_sk = {nullptr, 0};
constexpr int SK_G = 768;                             // 256 CUs x 3 workgroups (48 KB of LDS, <= 170 VGPRs each)
constexpr int64_t SK_PART = (int64_t)SK_G * 128 * 128 * 4;
constexpr int64_t SK_BYTES = SK_PART + SK_G * 4 + 64;
static std::atomic<uint32_t> g_sk_id{1};
static int sk_mode() { return 1; }      // (the caller decides by arming a scratch buffer or not: ops.STREAMK)
struct SkDisarm { ~SkDisarm() { g_sk.buf = nullptr; g_sk.bytes = 0; } };
static bool sk_take(SkDesc& sk) {
    void* b = g_sk.buf;
    const int64_t n = g_sk.bytes;
    g_sk.buf = nullptr;
    g_sk.bytes = 0;
    if (!b || n < SK_BYTES || ((uintptr_t)b & 15)) return false;
    sk.part = (float*)b;
    sk.flags = (uint32_t*)((char*)b + SK_PART);
    sk.err = sk.flags + SK_G;
    sk.id = g_sk_id.fetch_add(1, std::memory_order_relaxed);
    if (sk.id == 0) sk.id = g_sk_id.fetch_add(1, std::memory_order_relaxed);   // 0 is what a fresh buffer holds
    sk.G = SK_G;
    static const int coh = diag_env_int("SCAT_SK_COH", 1);
    sk.coh = coh;
    return true;
}

template <int WM, int BN, bool TF, bool DS = false, bool STEM = false>
static void launch_pw_split(const PwDesc& d, const OutDesc& dc_in, hipStream_t st) {
    constexpr int BM = 32 * WM;
    const int mt = cdiv(d.M, BM), nt = cdiv(d.npix, BN);
    constexpr size_t lds_bytes = (size_t)2 * 12 * BN * 16;
    OutDesc dc = dc_in;
    if (!dc.accumulate && !dc.bias) {                 // a forward convolution: its BatchNorm's sums ride in the epilogue
        dc.sg = nt * (4 / WM);
        dc.stats = epi_stats_take(d.M, dc.sg, &dc.stats_shift);
    }
    if (!dc.accumulate) dc.st_aux = store_policy(dc.n * 4);
    if constexpr (!STEM && !DS && !TF && WM == 4 && BN == 128) {
        // a data gradient that completes the gradient of a block output (C += ...): the BatchNorm backward's reduction
        // of the block that produced it rides in the epilogue when the caller armed it (scat_epilogue_bnb_arm)
        EpiBnb eb;
        if (dc.accumulate && !dc.bias && dc.mode == 1 && d.ntap == 1 && d.C % 32 == 0 && pw_plain_mode() &&
            epi_bnb_take(d.M, nt * (4 / WM), dc.n, &eb)) {
            dc.bnb_x = eb.x; dc.bnb_mask = eb.mask; dc.bnb_mean = eb.mean; dc.bnb_part = eb.part;
            dc.bnb_sg = nt * (4 / WM);
            hipLaunchKernelGGL((conv1x1_split_kernel<WM, BN, TF, DS, false, true, true>), dim3(mt * nt), dim3(NT), lds_bytes,
                               st, d, dc);
            append_kernel_label("_epibn");
            return;
        }
    }
    if constexpr (!STEM && BN == 128) {
        // persistent stream-K grid when the caller armed a scratch buffer and the launch has enough tiles to share out
        SkDesc sk{};
        if (d.ntap == 1 && d.C % 32 == 0 && pw_plain_mode() && sk_mode() && (int64_t)mt * nt >= 256 && sk_take(sk)) {
            hipLaunchKernelGGL((conv1x1_sk_kernel<WM, BN, TF, DS>), dim3(SK_G), dim3(NT), lds_bytes, st, d, dc, sk);
            append_kernel_label("_sk");
            return;
        }
    }
    if constexpr (!STEM) {
        if (d.ntap == 1 && d.C % 32 == 0 && pw_plain_mode()) {
            hipLaunchKernelGGL((conv1x1_split_kernel<WM, BN, TF, DS, false, true>), dim3(mt * nt), dim3(NT), lds_bytes, st,
                               d, dc);
            return;
        }
    }
    hipLaunchKernelGGL((conv1x1_split_kernel<WM, BN, TF, DS, STEM>), dim3(mt * nt), dim3(NT), lds_bytes, st, d, dc);
}

// ws[chunk = o / 2][plane][co][16 bf16], k16 = 8 (o & 1) + kw: the three bf16 terms of w[co][c][kh][kw], o = 3 kh + c;
// kw = 7 and o >= 21 are zero
__global__ __launch_bounds__(256) void stem_wprep_kernel(const float* __restrict__ w, uint16_t* __restrict__ dst, int M) {
    const int n = 12 * M * 16;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
        const int k16 = e & 15, co = (e >> 4) % M, ch = (e >> 4) / M;
        const int o = 2 * ch + (k16 >> 3), kw = k16 & 7;
        const int kh = o / 3, c = o - 3 * kh;
        const float v = (o < 21 && kw < 7) ? w[((co * 3 + c) * 7 + kh) * 7 + kw] : 0.f;
        uint32_t hi, mid, lo;
        split3(v, 0.f, hi, mid, lo);
        const int base = ((ch * 3) * M + co) * 16 + k16;
        dst[base] = (uint16_t)hi;
        dst[base + M * 16] = (uint16_t)mid;
        dst[base + 2 * M * 16] = (uint16_t)lo;
    }
}

// SCAT_PC: unset = per-layer choice (see scat_conv1x1_s1), 0 = every wavefront stages and multiplies
// (conv1x1_split_kernel), 5 / 6 = eight consumer wavefronts on a 256 x 128 tile (plain / wide form), 1 = producer/consumer wavefronts with
// 128-row tiles, 2 = 256-row tiles where the layer has them, 3 / 4 = the same with the wide (pixel-quad) form where the
// plane allows it
static int pc_mode() {
    static const int m = [] { const char* e = getenv("SCAT_PC"); return e ? atoi(e) : -1; }();
    return m;
}

template <int MI, bool TF, int DIAG = 0, bool W4 = false, int NCW = 4, bool S16 = false, int RING = 0>
static void launch_pw_pc(const PwDesc& d, const OutDesc& dc, hipStream_t st) {
    constexpr int BM = 32 * NCW * MI, BN = 128;
    const int mt = cdiv(d.M, BM), nt = cdiv(d.npix, BN);
    constexpr size_t lds_bytes = (size_t)(RING > 0 ? RING : 3) * 12 * BN * 16 + (RING > 0 ? 64 : 0);
    auto kern = conv1x1_pc_kernel<MI, TF, DIAG, W4, NCW, S16, RING>;
    static bool once = (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)lds_bytes) == hipSuccess);
    (void)once;
    hipLaunchKernelGGL(kern, dim3(mt * nt), dim3((NCW + 4) * 64), lds_bytes, st, d, dc);
}

template <int BM, int BN, bool V4, bool TF>
static void launch_pw(const PwDesc& d, const OutDesc& dc, hipStream_t st) {
    const int mt = cdiv(d.M, BM), nt = cdiv(d.npix, BN);
    constexpr size_t lds_bytes = sizeof(float) * 2 * PW_KS * BN;
    hipLaunchKernelGGL((conv1x1_kernel<BM, BN, V4, TF>), dim3(mt * nt), dim3(NT), lds_bytes, st, d, dc);
}

template <bool V4, bool TF>
static void launch_pw_cfg(int cfg, const PwDesc& d, const OutDesc& dc, hipStream_t st) {
    if (cfg == 0) launch_pw<128, 128, V4, TF>(d, dc, st);
    else if (cfg == 1) launch_pw<64, 128, V4, TF>(d, dc, st);
    else launch_pw<64, 64, V4, TF>(d, dc, st);
}

// General entry of the split kernel: dst (+)= sum over the listed taps and C source channels.
//   source pixel of column (oy, ox), tap (th, tw): (oy*a + th*tb + c0y, ox*a + tw*tb + c0x) of src[B][C][H][W]
//   weights: taps (kh0 + ts*th, kw0 + ts*tw) of the conv weight w[Cout][Cin][KH][KW], (row, channel) = (co, ci) or,
//   transposed, (ci, co).  ws >= taps_split_ws(M, C, KHt*KWt) bytes.
int64_t taps_split_ws(int M, int C, int ntap) { return (int64_t)ntap * M * ((C + 15) / 16 * 16) * 6; }

void taps_split_launch(const TapsGeom& g, const float* src, const float* w, const OutDesc& dc, int B, int C, int M,
                       const float* in_scale, const float* in_shift, int in_relu, void* ws, const char* label,
                       hipStream_t st, bool w_ready) {
    PwDesc d{};
    d.src = src; d.scale = in_scale; d.shift = in_shift; d.relu = in_scale ? in_relu : 0;
    d.C = C; d.M = M; d.HW = g.H * g.W; d.npix = B * g.OH * g.OW; d.dHW = FastDiv::make(d.HW);
    d.ntap = g.KHt * g.KWt; d.KWt = g.KWt; d.a = g.a; d.tb = g.tb; d.c0y = g.c0y; d.c0x = g.c0x; d.H = g.H; d.W = g.W;
    d.OW = g.OW; d.OHW = g.OH * g.OW; d.dOHW = FastDiv::make(d.OHW); d.dOW = FastDiv::make(g.OW);
    d.nsrc = (int64_t)B * C * d.HW;
    d.variant = tuning();
    const int64_t nel = (int64_t)d.ntap * M * ((C + 15) / 16 * 16);
    if (!w_ready)
        wprep_launch(wprep_job(w, ws, M, C, g.transposed, g.KH, g.KW, d.ntap, g.KWt, g.kh0, g.kw0, g.ts), st);
    d.w = (const float*)ws;
    d.nw = (nel * 6 + 3) / 4;
    int cfg = M > 64 ? 0 : 1;
    if ((int64_t)cdiv(M, 128) * cdiv(d.npix, 128) < 256 && (int64_t)cdiv(M, 64) * cdiv(d.npix, 64) >= 256) cfg = 2;
    if (tuning() >= 1 && tuning() <= 3) cfg = tuning() - 1;
    static const char* const names[] = {"128x128", "64x128", "64x64"};
    set_kernel_label("%s_split_%sx32%s", label, names[cfg], in_scale ? "_tf" : "");
    if (in_scale) {
        if (cfg == 0) launch_pw_split<4, 128, true>(d, dc, st);
        else if (cfg == 1) launch_pw_split<2, 128, true>(d, dc, st);
        else launch_pw_split<2, 64, true>(d, dc, st);
    } else {
        if (cfg == 0) launch_pw_split<4, 128, false>(d, dc, st);
        else if (cfg == 1) launch_pw_split<2, 128, false>(d, dc, st);
        else launch_pw_split<2, 64, false>(d, dc, st);
    }
}

}  // namespace scat

using namespace scat;

extern "C" int64_t scat_conv2d_fwd_split_ws(int Cout, int Cin, int KH, int KW) {
    return taps_split_ws(Cout, Cin, KH * KW);
}

// Forward convolution (1x1 or 3x3, stride 1 or 2) on the taps kernel: contraction ordered (tap, channel), one
// 32-channel activation stage of one tap per barrier.  Split-operand products only (scat_get_math_mode() == 1);
// Cin % 16 == 0.  Used for the stride-2 convolutions (3x3/s1 has the halo kernel, 1x1/s1 scat_conv1x1_s1).
extern "C" int scat_conv2d_fwd_split(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int H,
                                     int W, int Cout, int KH, int KW, int stride, int pad, const float* in_scale,
                                     const float* in_shift, int in_relu, void* ws, int64_t ws_bytes, int w_ready,
                                     void* stream) {
    int OH, OW;
    if (int e = check_geom("scat_conv2d_fwd_split", B, Cin, H, W, Cout, KH, KW, stride, pad, &OH, &OW)) return e;
    SCAT_REQUIRE(x && w && y, SCAT_E_ARG, "scat_conv2d_fwd_split: null pointer");
    SCAT_REQUIRE(math_mode() == 1, SCAT_E_ARG, "scat_conv2d_fwd_split: needs the split-operand product mode");
    SCAT_REQUIRE(KH <= 3 && Cin % 16 == 0, SCAT_E_SHAPE, "scat_conv2d_fwd_split: 1x1/3x3 and Cin % 16 == 0 only");
    SCAT_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), SCAT_E_ARG, "scat_conv2d_fwd_split: scale/shift pair");
    SCAT_REQUIRE(ws && ws_bytes >= scat_conv2d_fwd_split_ws(Cout, Cin, KH, KW) && ((uintptr_t)ws & 15) == 0,
                 SCAT_E_WORKSPACE, "scat_conv2d_fwd_split: workspace too small / unaligned");
    TapsGeom g{};
    g.H = H; g.W = W; g.OH = OH; g.OW = OW; g.a = stride; g.tb = 1; g.c0y = -pad; g.c0x = -pad;
    g.KHt = KH; g.KWt = KW; g.KH = KH; g.KW = KW; g.kh0 = 0; g.kw0 = 0; g.ts = 1; g.transposed = 0;
    OutDesc dc{};
    dc.p = y; dc.mode = 1; dc.I = Cout; dc.J = B * OH * OW; dc.C = Cout; dc.HW = OH * OW;
    dc.dHW = FastDiv::make(OH * OW); dc.bias = bias; dc.bias_mode = bias ? 1 : 0; dc.n = (int64_t)B * Cout * OH * OW;
    char label[32];
    snprintf(label, sizeof label, "conv%dx%d_s%d", KH, KW, stride);
    taps_split_launch(g, x, w, dc, B, Cin, Cout, in_scale, in_shift, in_relu, ws, label, (hipStream_t)stream,
                      w_ready != 0);
    SCAT_LAUNCH_CHECK("scat_conv2d_fwd_split");
    return SCAT_OK;
}

extern "C" int64_t scat_conv7x7_s2_fwd_split_ws(int Cout) { return (int64_t)12 * 3 * Cout * 32; }

// The ResNet stem (models/resnet.py:105: Conv2d(3, 64, 7, stride 2, padding 3, bias=False)) on split-operand products:
// y[B,Cout,OH,OW] = conv(x[B,3,H,W], w[Cout,3,7,7]).  ws: scat_conv7x7_s2_fwd_split_ws(Cout) bytes.
extern "C" int scat_conv7x7_s2_fwd_split(const float* x, const float* w, float* y, int B, int H, int W, int Cout,
                                         void* ws, int64_t ws_bytes, void* stream) {
    SCAT_REQUIRE(x && w && y, SCAT_E_ARG, "scat_conv7x7_s2_fwd_split: null pointer");
    SCAT_REQUIRE(math_mode() == 1, SCAT_E_ARG, "scat_conv7x7_s2_fwd_split: needs the split-operand product mode");
    SCAT_REQUIRE(B > 0 && H > 6 && W > 6 && Cout > 0, SCAT_E_SHAPE, "scat_conv7x7_s2_fwd_split: bad dimension");
    SCAT_REQUIRE(ws && ws_bytes >= scat_conv7x7_s2_fwd_split_ws(Cout) && ((uintptr_t)ws & 15) == 0, SCAT_E_WORKSPACE,
                 "scat_conv7x7_s2_fwd_split: workspace too small / unaligned");
    const int OH = (H + 6 - 7) / 2 + 1, OW = (W + 6 - 7) / 2 + 1;
    SCAT_REQUIRE(fits_i32((int64_t)B * 3 * H * W * 4) && fits_i32((int64_t)B * Cout * OH * OW * 4), SCAT_E_SHAPE,
                 "scat_conv7x7_s2_fwd_split: tensor exceeds 32-bit byte offsets");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(stem_wprep_kernel, dim3(cdiv(12 * Cout * 16, 256)), dim3(256), 0, st, w, (uint16_t*)ws, Cout);
    PwDesc d{};
    d.src = x; d.w = (const float*)ws; d.C = 192; d.M = Cout; d.HW = H * W; d.npix = B * OH * OW;
    d.dHW = FastDiv::make(d.HW);
    d.ntap = 1; d.KWt = 1; d.a = 1; d.tb = 1; d.c0y = 0; d.c0x = 0; d.H = H; d.W = W; d.OW = OW; d.OHW = OH * OW;
    d.dOHW = FastDiv::make(d.OHW); d.dOW = FastDiv::make(OW);
    d.nsrc = (int64_t)B * 3 * H * W; d.nw = ((int64_t)12 * 3 * Cout * 32 + 3) / 4;
    d.variant = tuning();
    OutDesc dc{};
    dc.p = y; dc.mode = 1; dc.I = Cout; dc.J = d.npix; dc.C = Cout; dc.HW = OH * OW; dc.dHW = FastDiv::make(OH * OW);
    dc.n = (int64_t)B * Cout * OH * OW;
    set_kernel_label("conv7x7_s2_split_%dx128x32", Cout > 64 ? 128 : 64);
    if (Cout > 64) launch_pw_split<4, 128, false, false, true>(d, dc, st);
    else launch_pw_split<2, 128, false, false, true>(d, dc, st);
    SCAT_LAUNCH_CHECK("scat_conv7x7_s2_fwd_split");
    return SCAT_OK;
}

// Data gradient of a 1x1/stride-1 convolution whose output gradient is a BatchNorm backward that was never
// materialised: dx[B,Cin,HW] (+)= w^T . (ca*g + cb*z + cc), g = masked incoming gradient [B,Cout,HW] (written by
// scat_bn_bwd_pre), z = the conv's raw output [B,Cout,HW], coef3 = [ca | cb | cc] per output channel.
// Split-operand products only; Cout % 16 == 0.  ws: scat_conv1x1_s1_ws(Cin, Cout) bytes.
extern "C" int scat_conv1x1_s1_bnb(const float* g, const float* z, const float* coef3, const float* w, float* dx, int B,
                                   int Cin, int HW, int Cout, int accumulate, void* ws, int64_t ws_bytes,
                                   int w_ready, void* stream) {
    SkDisarm sk_guard;
    SCAT_REQUIRE(g && z && coef3 && w && dx, SCAT_E_ARG, "scat_conv1x1_s1_bnb: null pointer");
    SCAT_REQUIRE(math_mode() == 1, SCAT_E_ARG, "scat_conv1x1_s1_bnb: needs the split-operand product mode");
    SCAT_REQUIRE(B > 0 && Cin > 0 && HW > 0 && Cout > 0 && Cout % 16 == 0, SCAT_E_SHAPE,
                 "scat_conv1x1_s1_bnb: bad dimension (Cout must be a multiple of 16)");
    SCAT_REQUIRE(ws && ws_bytes >= taps_split_ws(Cin, Cout, 1) && ((uintptr_t)ws & 15) == 0, SCAT_E_WORKSPACE,
                 "scat_conv1x1_s1_bnb: workspace too small / unaligned");
    SCAT_REQUIRE(fits_i32((int64_t)B * Cout * HW * 4) && fits_i32((int64_t)B * Cin * HW * 4), SCAT_E_SHAPE,
                 "scat_conv1x1_s1_bnb: tensor exceeds 32-bit byte offsets");
    hipStream_t st = (hipStream_t)stream;
    const int C = Cout, M = Cin;
    PwDesc d{};
    d.src = g; d.src2 = z; d.coef = coef3;
    d.C = C; d.M = M; d.HW = HW; d.npix = B * HW; d.dHW = FastDiv::make(HW);
    d.ntap = 1; d.KWt = 1; d.a = 1; d.tb = 1; d.c0y = 0; d.c0x = 0; d.H = 1; d.W = HW; d.OW = HW; d.OHW = HW;
    d.dOHW = FastDiv::make(HW); d.dOW = FastDiv::make(HW);
    d.nsrc = (int64_t)B * C * HW; d.variant = tuning();
    const int64_t nel = (int64_t)M * ((C + 15) / 16 * 16);
    const int rblocks = (int)((nel + 255) / 256 < 2048 ? (nel + 255) / 256 : 2048);
    if (!w_ready) wprep_launch(wprep_job(w, ws, M, C, 1, 1, 1, 1, 1, 0, 0, 1), st);
    d.w = (const float*)ws;
    d.nw = (nel * 6 + 3) / 4;
    OutDesc dc{};
    dc.p = dx; dc.mode = 1; dc.I = M; dc.J = d.npix; dc.C = M; dc.HW = HW; dc.dHW = FastDiv::make(HW);
    dc.accumulate = accumulate; dc.n = (int64_t)B * M * HW;
    const int cfg = M > 64 ? 0 : 1;
    set_kernel_label("conv1x1_split_%sx32_bnb", cfg == 0 ? "128x128" : "64x128");
    if (cfg == 0) launch_pw_split<4, 128, false, true>(d, dc, st);
    else launch_pw_split<2, 128, false, true>(d, dc, st);
    SCAT_LAUNCH_CHECK("scat_conv1x1_s1_bnb");
    return SCAT_OK;
}

extern "C" int64_t scat_conv1x1_s1_ws(int M, int C) { return (int64_t)M * ((C + 15) / 16 * 16) * 6; }

// dst[B,M,HW] (+)= A[M,C] . f(src[B,C,HW]),  f = relu(x*scale+shift) when scale is given.
//   transposed = 0, forward:        w = [M][C]  (M = Cout, C = Cin)
//   transposed = 1, data gradient:  w = [C][M]  (the forward weight; M = Cin, C = Cout), src = dy
// ws: scat_conv1x1_s1_ws(M, C) bytes (the weights' bf16 terms, or their fp32 transpose).
// Needs C % 16 == 0 and 16-B aligned w/src; callers fall back to scat_conv2d_fwd / scat_conv2d_dgrad otherwise.
extern "C" int scat_conv1x1_s1(const float* src, const float* w, float* dst, int B, int C, int HW, int M,
                               int transposed, const float* bias, const float* in_scale, const float* in_shift,
                               int in_relu, int accumulate, void* ws, int64_t ws_bytes, int w_ready, void* stream) {
    SkDisarm sk_guard;      // an armed stream-K scratch is this call's or nobody's
    SCAT_REQUIRE(src && w && dst, SCAT_E_ARG, "scat_conv1x1_s1: null pointer");
    SCAT_REQUIRE(!w_ready || math_mode() == 1, SCAT_E_ARG, "scat_conv1x1_s1: prepared weights exist for split products only");
    SCAT_REQUIRE(ws && ws_bytes >= scat_conv1x1_s1_ws(M, C), SCAT_E_WORKSPACE, "scat_conv1x1_s1: workspace too small");
    const float* a = w;
    SCAT_REQUIRE(B > 0 && C > 0 && HW > 0 && M > 0, SCAT_E_SHAPE, "scat_conv1x1_s1: non-positive dimension");
    SCAT_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), SCAT_E_ARG, "scat_conv1x1_s1: scale/shift pair");
    SCAT_REQUIRE(C % 16 == 0, SCAT_E_SHAPE, "scat_conv1x1_s1: channels must be a multiple of 16");
    SCAT_REQUIRE(((uintptr_t)w & 15) == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)ws & 15) == 0, SCAT_E_ARG,
                 "scat_conv1x1_s1: 16-B alignment");
    SCAT_REQUIRE(!in_scale || (((uintptr_t)in_scale & 15) == 0 && ((uintptr_t)in_shift & 15) == 0), SCAT_E_ARG,
                 "scat_conv1x1_s1: scale/shift must be 16-B aligned");
    SCAT_REQUIRE(fits_i32((int64_t)B * C * HW * 4) && fits_i32((int64_t)B * M * HW * 4) && fits_i32((int64_t)M * C * 4),
                 SCAT_E_SHAPE, "scat_conv1x1_s1: tensor exceeds 32-bit byte offsets");
    hipStream_t st = (hipStream_t)stream;
    PwDesc d{};
    d.src = src; d.w = a; d.scale = in_scale; d.shift = in_shift; d.relu = in_scale ? in_relu : 0;
    d.C = C; d.M = M; d.HW = HW; d.npix = B * HW; d.dHW = FastDiv::make(HW);
    d.ntap = 1; d.KWt = 1; d.a = 1; d.tb = 1; d.c0y = 0; d.c0x = 0; d.H = 1; d.W = HW; d.OW = HW; d.OHW = HW;
    d.dOHW = FastDiv::make(HW); d.dOW = FastDiv::make(HW);
    d.nsrc = (int64_t)B * C * HW; d.nw = (int64_t)M * C;
    d.variant = tuning();
    OutDesc dc{};
    dc.p = dst; dc.mode = 1; dc.I = M; dc.J = d.npix; dc.C = M; dc.HW = HW; dc.dHW = FastDiv::make(HW);
    dc.bias = bias; dc.bias_mode = bias ? 1 : 0; dc.accumulate = accumulate; dc.n = (int64_t)B * M * HW;
    auto tiles = [&](int bm, int bn) { return (int64_t)cdiv(M, bm) * cdiv(d.npix, bn); };
    int cfg = (M > 64 && tiles(128, 128) >= 1024) ? 0 : 2;   // measured: 64x128 never wins at batch 96
    if (tuning() >= 1 && tuning() <= 3) cfg = tuning() - 1;
    static const char* const names[] = {"128x128", "64x128", "64x64"};
    const int64_t nel = (int64_t)M * ((C + 15) / 16 * 16);
    const int rblocks = (int)((nel + 255) / 256 < 2048 ? (nel + 255) / 256 : 2048);
    if (math_mode() == 1) {
        if (!w_ready) wprep_launch(wprep_job(w, ws, M, C, transposed, 1, 1, 1, 1, 0, 0, 1), st);
        d.w = (const float*)ws;
        d.nw = (nel * 6 + 3) / 4;
        if (!(tuning() >= 1 && tuning() <= 3)) {
            static const int thin = diag_env_int("SCAT_PW_THIN", 0);
            cfg = M > 64 ? 0 : 1;   // measured at batch 96: 64x64 never wins
            if (cfg == 0 && tiles(128, 128) < thin) cfg = 1;          // (SCAT_PW_THIN: 64x128 below that many tiles)
        }
        int pc = pc_mode();
        // default (SCAT_PC unset): the 8-consumer producer/consumer form for the few layers whose 128x128 tile count is
        // just over one tile per CU (294-296 tiles at batch 96: 1024->256 @14x14 forward, 256->1024 @14x14 and
        // 1024->2048 @7x7 data gradients) — there the round-1 kernel runs 38 CUs with two workgroups and 218 with one,
        // and one 256x128 workgroup per CU is 10-15 % faster (profiles/r02_conv_shapes.txt); everywhere else the
        // round-1 form wins.  The 64-wide tile counts are batch dependent, the rule is not.
        if (pc < 0) pc = (cfg == 0 && M >= 256 && tiles(128, 128) >= 257 && tiles(128, 128) < 330) ? 6 : 0;
        // wide form: pointwise, whole pixel quads inside one image, 16-byte aligned planes, NCHW output
        const bool w4ok = HW % 4 == 0 && ((uintptr_t)dst & 15) == 0 && (!bias || dc.bias_mode == 1);
#ifdef SCAT_DIAG     // negative results kept for A/B runs in the tools build only: barrier-free ring, 16x16x32 consumers
        if (cfg == 0 && pc == 8 && M >= 256) {
            set_kernel_label("conv1x1_split_pcring_256x128x32%s", in_scale ? "_tf" : "");
            if (in_scale) launch_pw_pc<1, true, 0, false, 8, false, 6>(d, dc, st);
            else launch_pw_pc<1, false, 0, false, 8, false, 6>(d, dc, st);
            SCAT_LAUNCH_CHECK("scat_conv1x1_s1");
            return SCAT_OK;
        }
        if (cfg == 0 && pc == 7 && dc.mode == 1 && (!bias || dc.bias_mode == 1)) {
            set_kernel_label("conv1x1_split_pc16_128x128x32%s", in_scale ? "_tf" : "");
            if (in_scale) launch_pw_pc<1, true, 0, false, 4, true>(d, dc, st);
            else launch_pw_pc<1, false, 0, false, 4, true>(d, dc, st);
            SCAT_LAUNCH_CHECK("scat_conv1x1_s1");
            return SCAT_OK;
        }
#else
        if (pc != 5 && pc != 6) pc = 0;      // the shipped library has the two forms the default rule picks, nothing else
#endif
        if (cfg == 0 && (pc == 5 || pc == 6) && M >= 256) {
            const bool wide = pc == 6 && w4ok;
            set_kernel_label("conv1x1_split_pc%s256x128x32%s", wide ? "4_" : "8w_", in_scale ? "_tf" : "");
            if (wide) {
                if (in_scale) launch_pw_pc<1, true, 0, true, 8>(d, dc, st); else launch_pw_pc<1, false, 0, true, 8>(d, dc, st);
            } else {
                if (in_scale) launch_pw_pc<1, true, 0, false, 8>(d, dc, st); else launch_pw_pc<1, false, 0, false, 8>(d, dc, st);
            }
            SCAT_LAUNCH_CHECK("scat_conv1x1_s1");
            return SCAT_OK;
        }
#ifdef SCAT_DIAG     // four-consumer producer/consumer forms: no gain over the plain kernel, tools build only
        if (cfg == 0 && pc >= 3 && pc <= 4 && w4ok) {
            const bool big = pc == 4 && M >= 256;
            set_kernel_label("conv1x1_split_pc4_%dx128x32%s", big ? 256 : 128, in_scale ? "_tf" : "");
#ifdef SCAT_DIAG
            if (!in_scale && !big && tuning() >= 100) {      // ablation variants (SCAT_TUNE=100+DIAG), diag build only
                switch (tuning() - 100) {
                case 1: launch_pw_pc<1, false, 1, true>(d, dc, st); break;
                case 2: launch_pw_pc<1, false, 2, true>(d, dc, st); break;
                case 4: launch_pw_pc<1, false, 4, true>(d, dc, st); break;
                case 8: launch_pw_pc<1, false, 8, true>(d, dc, st); break;
                case 11: launch_pw_pc<1, false, 11, true>(d, dc, st); break;
                case 20: launch_pw_pc<1, false, 20, true>(d, dc, st); break;
                default: launch_pw_pc<1, false, 0, true>(d, dc, st);
                }
            } else
#endif
            if (in_scale) {
                if (big) launch_pw_pc<2, true, 0, true>(d, dc, st); else launch_pw_pc<1, true, 0, true>(d, dc, st);
            } else {
                if (big) launch_pw_pc<2, false, 0, true>(d, dc, st); else launch_pw_pc<1, false, 0, true>(d, dc, st);
            }
            SCAT_LAUNCH_CHECK("scat_conv1x1_s1");
            return SCAT_OK;
        }
        if (cfg == 0 && pc && pc <= 4) {
            const bool big = (pc == 2 || pc == 4) && M >= 256;
            set_kernel_label("conv1x1_split_pc%dx128x32%s", big ? 256 : 128, in_scale ? "_tf" : "");
#ifdef SCAT_DIAG
            if (!in_scale && !big && tuning() >= 100) {      // ablation variants (SCAT_TUNE=100+DIAG), diag build only
                switch (tuning() - 100) {
                case 1: launch_pw_pc<1, false, 1>(d, dc, st); break;
                case 2: launch_pw_pc<1, false, 2>(d, dc, st); break;
                case 4: launch_pw_pc<1, false, 4>(d, dc, st); break;
                case 8: launch_pw_pc<1, false, 8>(d, dc, st); break;
                case 16: launch_pw_pc<1, false, 16>(d, dc, st); break;
                case 17: launch_pw_pc<1, false, 17>(d, dc, st); break;
                case 3: launch_pw_pc<1, false, 3>(d, dc, st); break;
                case 11: launch_pw_pc<1, false, 11>(d, dc, st); break;
                case 20: launch_pw_pc<1, false, 20>(d, dc, st); break;
                default: launch_pw_pc<1, false>(d, dc, st);
                }
                SCAT_LAUNCH_CHECK("scat_conv1x1_s1");
                return SCAT_OK;
            }
#endif
            if (in_scale) { if (big) launch_pw_pc<2, true>(d, dc, st); else launch_pw_pc<1, true>(d, dc, st); }
            else { if (big) launch_pw_pc<2, false>(d, dc, st); else launch_pw_pc<1, false>(d, dc, st); }
            SCAT_LAUNCH_CHECK("scat_conv1x1_s1");
            return SCAT_OK;
        }
#endif
        set_kernel_label("conv1x1_split_%sx32%s", names[cfg], in_scale ? "_tf" : "");
        if (in_scale) {
            if (cfg == 0) launch_pw_split<4, 128, true>(d, dc, st);
            else if (cfg == 1) launch_pw_split<2, 128, true>(d, dc, st);
            else launch_pw_split<2, 64, true>(d, dc, st);
        } else {
            if (cfg == 0) launch_pw_split<4, 128, false>(d, dc, st);
            else if (cfg == 1) launch_pw_split<2, 128, false>(d, dc, st);
            else launch_pw_split<2, 64, false>(d, dc, st);
        }
        SCAT_LAUNCH_CHECK("scat_conv1x1_s1");
        return SCAT_OK;
    }
    if (transposed) {
        hipLaunchKernelGGL(w1x1_t_kernel, dim3(rblocks), dim3(256), 0, st, w, (float*)ws, M, C);
        d.w = (const float*)ws;
    }
    const bool v4 = HW % 4 == 0;
    set_kernel_label("conv1x1_pw_%sx32%s%s", names[cfg], v4 ? "_b4" : "", in_scale ? "_tf" : "");
    if (v4) {
        if (in_scale) launch_pw_cfg<true, true>(cfg, d, dc, st);
        else launch_pw_cfg<true, false>(cfg, d, dc, st);
    } else {
        if (in_scale) launch_pw_cfg<false, true>(cfg, d, dc, st);
        else launch_pw_cfg<false, false>(cfg, d, dc, st);
    }
    SCAT_LAUNCH_CHECK("scat_conv1x1_s1");
    return SCAT_OK;
}

// ---------------------------------------------------------------- dense GEMM on the pointwise split kernel
//
// C[M][N] (+)= op(A)[M][K] . B[K][N] (+ bias[N]); B and C row-major, A either [M][K] (a_transposed = 0) or stored
// [K][M] (a_transposed = 1).  This IS the pointwise convolution above with one image: A plays the weights (split once
// into MFMA operand planes by the re-layout launch: ws), B the activations [K channels][N pixels] that are split on
// their way into LDS, C the NCHW output [1][M][N].  It puts the ViT projections (vision_transformer.py:46-79:
// x.Wqkv^T, attn.Wout^T and their gradients, M = 2016 tokens at batch 96) on split-operand products:
//   forward   y  = x . W^T      A = x [M][K],             B = W^T [K][N] (scat_transpose2d of the weight)
//   dgrad     dx = dy . W       A = dy [M][N'],           B = W [N'][K'] as stored
//   wgrad     dW = dy^T . x     A = dy, a_transposed = 1, B = x [tokens][K] as stored
// K need not be a multiple of 16 (the re-layout pads A, the activation loads mask the ragged channels).
extern "C" int64_t scat_gemm_split_ws(int M, int K) { return taps_split_ws(M, K, 1); }

extern "C" int scat_gemm_split(const float* a, int a_transposed, const float* b, float* c, int M, int N, int K,
                               const float* bias_n, int accumulate, void* ws, int64_t ws_bytes, void* stream) {
    SCAT_REQUIRE(a && b && c, SCAT_E_ARG, "scat_gemm_split: null pointer");
    SCAT_REQUIRE(M > 0 && N > 0 && K > 0, SCAT_E_SHAPE, "scat_gemm_split: non-positive dimension");
    SCAT_REQUIRE(math_mode() == 1, SCAT_E_ARG, "scat_gemm_split: needs the split-operand product mode");
    SCAT_REQUIRE(ws && ws_bytes >= scat_gemm_split_ws(M, K) && ((uintptr_t)ws & 15) == 0, SCAT_E_WORKSPACE,
                 "scat_gemm_split: workspace too small / unaligned");
    SCAT_REQUIRE(fits_i32((int64_t)K * N * 4) && fits_i32((int64_t)M * N * 4) && fits_i32((int64_t)M * K * 4),
                 SCAT_E_SHAPE, "scat_gemm_split: operand exceeds 32-bit byte offsets");
    hipStream_t st = (hipStream_t)stream;
    wprep_launch(wprep_job(a, ws, M, K, a_transposed, 1, 1, 1, 1, 0, 0, 1), st);
    PwDesc d{};
    d.src = b; d.C = K; d.M = M; d.HW = N; d.npix = N; d.dHW = FastDiv::make(N);
    d.ntap = 1; d.KWt = 1; d.a = 1; d.tb = 1; d.c0y = 0; d.c0x = 0; d.H = 1; d.W = N; d.OW = N; d.OHW = N;
    d.dOHW = FastDiv::make(N); d.dOW = FastDiv::make(N);
    d.nsrc = (int64_t)K * N; d.variant = tuning();
    const int64_t nel = (int64_t)M * ((K + 15) / 16 * 16);
    d.w = (const float*)ws;
    d.nw = (nel * 6 + 3) / 4;
    OutDesc dc{};
    dc.p = c; dc.mode = 1; dc.I = M; dc.J = N; dc.C = M; dc.HW = N; dc.dHW = FastDiv::make(N);
    dc.bias = bias_n; dc.bias_mode = bias_n ? 2 : 0; dc.accumulate = accumulate; dc.n = (int64_t)M * N;
    // few, small problems (2016 x 1536 x 784 is the largest): the tile is chosen for workgroup count first
    auto tiles = [&](int bm, int bn) { return (int64_t)cdiv(M, bm) * cdiv(N, bn); };
    int cfg = tiles(128, 128) >= 512 ? 0 : (tiles(64, 128) >= 512 ? 1 : 2);
    if (tuning() >= 1 && tuning() <= 3) cfg = tuning() - 1;
    static const char* const names[] = {"128x128", "64x128", "64x64"};
    set_kernel_label("gemm_split_%sx32", names[cfg]);
    if (cfg == 0) launch_pw_split<4, 128, false>(d, dc, st);
    else if (cfg == 1) launch_pw_split<2, 128, false>(d, dc, st);
    else launch_pw_split<2, 64, false>(d, dc, st);
    SCAT_LAUNCH_CHECK("scat_gemm_split");
    return SCAT_OK;
}

// dst[C][R] = src[R][C] (row-major): the weight transpose of the forward projection above
__global__ __launch_bounds__(256) void transpose2d_kernel(const float* __restrict__ src, float* __restrict__ dst, int R,
                                                          int C) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, cc = c0 + tx;
        tile[ty + 8 * k][tx] = (r < R && cc < C) ? src[(int64_t)r * C + cc] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int cc = c0 + ty + 8 * k, r = r0 + tx;
        if (cc < C && r < R) dst[(int64_t)cc * R + r] = tile[tx][ty + 8 * k];
    }
}

extern "C" int scat_transpose2d(const float* src, float* dst, int R, int C, void* stream) {
    SCAT_REQUIRE(src && dst && R > 0 && C > 0, SCAT_E_ARG, "scat_transpose2d: bad argument");
    hipLaunchKernelGGL(transpose2d_kernel, dim3(cdiv(C, 32), cdiv(R, 32)), dim3(256), 0, (hipStream_t)stream, src, dst,
                       R, C);
    SCAT_LAUNCH_CHECK("scat_transpose2d");
    return SCAT_OK;
}

/* Stream-K scratch for the pointwise kernels (include/scat_hip.h).  Consumed by the next qualifying launch of this host
 * thread; a launch that does not qualify (few tiles, taps, scratch too small) disarms it and runs one tile per
 * workgroup. */
extern "C" int64_t scat_streamk_bytes(void) { return scat::SK_BYTES; }
extern "C" int scat_streamk_arm(void* buf, int64_t bytes) {
    scat::g_sk.buf = buf;
    scat::g_sk.bytes = buf ? bytes : 0;
    return SCAT_OK;
}
extern "C" int scat_streamk_error(const void* buf, int64_t bytes, void* stream) {
    SCAT_REQUIRE(buf && bytes >= scat::SK_BYTES, SCAT_E_ARG, "scat_streamk_error: not a stream-K scratch buffer");
    uint32_t v = 0;
    hipError_t e = hipMemcpyAsync(&v, (const char*)buf + scat::SK_PART + scat::SK_G * 4, 4, hipMemcpyDeviceToHost,
                                  (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    SCAT_REQUIRE(e == hipSuccess, SCAT_E_LAUNCH, "scat_streamk_error: %s", hipGetErrorString(e));
    SCAT_REQUIRE(v == 0, SCAT_E_LAUNCH, "a stream-K workgroup gave up waiting for a partial tile (device error word %u)", v);
    return SCAT_OK;
}

// 3x3 / stride 1 / pad 1 convolution (forward and data-gradient) with an LDS-resident halo.
//
// The generic gather (gemm_engine.h) re-reads every input element once per tap (9x) and spends vector
// ALU on tap decode for each of those loads.  Here the contraction is ordered (channel chunk, tap, channel):
//  * B operand: a chunk of 16 source channels of the tile's flat pixels PLUS a halo of W+1 pixels on either
//    side is staged into LDS once (fused BatchNorm+ReLU applied once per element); each of the 9 taps is an
//    MFMA pass over the same rows at a shifted LDS address — in the flat [n][y][x] pixel order of an NCHW
//    plane the tap (kh,kw) of pixel g is element g + (kh-1)*W + (kw-1).  Taps that fall on padding (or
//    wrap a row / an image) are zeroed by a per-lane 9-bit mask when the fragment is read.
//  * A operand (weights, L2-resident): never touches LDS.  The 32x32x2 MFMA wants lane (i, h) to hold
//    A[i][k(j,h)] at its j-th issue; the contraction order inside a chunk is free, so k(j,h) = 8h + j and a
//    lane's 8 operands are 32 contiguous bytes of the re-laid weights wt[tap][chunk][row][16]: two 16-B
//    loads per 32-row block per tap, prefetched one tap ahead into registers.
//  * so the only LDS traffic is the halo (written once per chunk, read 9 times with ds_read_b64: rows are
//    stored in k pairs), and the only barrier is one per chunk = per 9 taps = per 288 MFMAs per wave.
//
// Replaces nn.Conv2d(k=3, s=1, p=1) and its input gradient as called at models/resnet.py:68-69,86-88 of
// the reference (the conv2 of every Bottleneck), fp32.
#include "conv_common.h"
#include "split.h"

namespace scat {

struct HaloDesc {
    const float* src;     // [Nimg][C][H][W]
    const float* wt;      // [9][nchunk][M][16]  (channels zero-padded to a multiple of 16)
    const float* scale;   // optional fused input transform, per source channel
    const float* shift;
    int relu;
    int C, M, H, W, HW, npix;
    int sgn;              // +1 forward: tap reads (y+kh-1, x+kw-1);  -1 data-gradient: (y+1-kh, x+1-kw)
    int RS;               // halo row length (floats) = tile pixels + 2*(W+1)
    FastDiv dHW, dW;
    int64_t nsrc, nwt;
};

constexpr int HB_K = 16;      // source channels per chunk
constexpr int HB_TLOAD = 6;   // tap at which the next chunk's halo loads are issued (written to LDS after tap 8)


template <int BM, int HB_N, bool TF>
__global__ __launch_bounds__(NT, (BM * HB_N >= 128 * 128 ? 2 : 3)) void conv3x3_halo_kernel(HaloDesc d, OutDesc dc) {
    constexpr int MI = BM / 64, NI = HB_N / 64;
    extern __shared__ __align__(16) float lds[];      // B[2][8 k-pairs][RS][2]
    const int RS = d.RS;
    auto Bs = [&](int buf) -> float* { return lds + buf * (HB_K * RS); };

    const int mt = (d.M + BM - 1) / BM, nt = (d.npix + HB_N - 1) / HB_N;
    const int tile = xcd_remap(blockIdx.x, mt * nt);
    const int i0 = (tile % mt) * BM, j0 = (tile / mt) * HB_N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const int halo = d.W + 1;
    const int nchunk = (d.C + HB_K - 1) / HB_K;

    // ---- halo staging: thread e owns LDS column e of all 16 channel rows
    const __amdgpu_buffer_rsrc_t rsrc_b = make_rsrc(d.src, d.nsrc);
    int boff;                                          // byte offset of (n, channel 0, r) or OOB
    bool bok;
    {
        const int g = j0 - halo + tid;
        bok = tid < RS && g >= 0 && g < d.npix;
        const uint32_t gg = bok ? (uint32_t)g : 0u;
        const uint32_t n = d.dHW.div(gg);
        boff = bok ? (int)((n * (uint32_t)d.C * (uint32_t)d.HW + (gg - n * (uint32_t)d.HW)) * 4u) : OOB;
    }
    const int chw4 = d.HW * 4;
    float bst[HB_K];
    auto load_b = [&](int c0) {                        // channels >= C read 0 (also the prefetch past the end)
#pragma unroll
        for (int m = 0; m < HB_K; ++m)
            bst[m] = __uint_as_float(
                __builtin_amdgcn_raw_buffer_load_b32(rsrc_b, c0 + m < d.C ? boff : OOB, (c0 + m) * chw4, 0));
    };
    // channel m = 8h + j of the chunk lives at [(4h + j/2)][column][j & 1]: lane half h reads its k pair as 8 bytes
    auto store_b = [&](int c0, float* dst) {
        if (tid < RS) {
#pragma unroll
            for (int m = 0; m < HB_K; ++m) {
                float x = bst[m];
                if constexpr (TF) {
                    const int c = c0 + m < d.C ? c0 + m : 0;
                    x = fmaf(x, d.scale[c], d.shift[c]);
                    x = d.relu ? fmaxf(x, 0.f) : x;
                    x = bok ? x : 0.f;
                }
                dst[(((m >> 3) * 4 + ((m & 7) >> 1)) * RS + tid) * 2 + (m & 1)] = x;
            }
        }
    };

    // ---- weights: lane (row, h) holds k = 8h .. 8h+7 of its rows, straight from global memory
    const __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc(d.wt, d.nwt);
    int aoff[MI];
#pragma unroll
    for (int a = 0; a < MI; ++a) {
        const int row = i0 + wm * (BM / 2) + a * 32 + l31;
        aoff[a] = row < d.M ? (row * 16 + lh * 8) * 4 : OOB;
    }
    const int achunk4 = d.M * 64;                      // bytes per (tap, chunk) slab
    auto load_a = [&](float (&dst)[MI][8], int tap, int ch) {   // ch >= nchunk: zeros
#pragma unroll
        for (int a = 0; a < MI; ++a) {
            const int vo = ch < nchunk ? aoff[a] : OOB;
            const int so = (tap * nchunk + ch) * achunk4;
            u32x4 t0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, vo, so, 0);
            u32x4 t1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, vo + 16, so, 0);
            dst[a][0] = __uint_as_float(t0.x); dst[a][1] = __uint_as_float(t0.y);
            dst[a][2] = __uint_as_float(t0.z); dst[a][3] = __uint_as_float(t0.w);
            dst[a][4] = __uint_as_float(t1.x); dst[a][5] = __uint_as_float(t1.y);
            dst[a][6] = __uint_as_float(t1.z); dst[a][7] = __uint_as_float(t1.w);
        }
    };

    // ---- per-lane tap-validity masks of the NI pixels this lane feeds to the MFMA B operand
    uint32_t pm[NI];
#pragma unroll
    for (int b = 0; b < NI; ++b) {
        const int j = j0 + wn * (HB_N / 2) + b * 32 + l31;
        uint32_t m = 0;
        if (j < d.npix) {
            const uint32_t n = d.dHW.div((uint32_t)j);
            const uint32_t r = (uint32_t)j - n * (uint32_t)d.HW;
            const int y = (int)d.dW.div(r), x = (int)r - y * d.W;
            uint32_t rowm = 0, colm = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                rowm |= ((unsigned)(y + d.sgn * (k - 1)) < (unsigned)d.H ? 1u : 0u) << k;
                colm |= ((unsigned)(x + d.sgn * (k - 1)) < (unsigned)d.W ? 1u : 0u) << k;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if ((rowm >> k) & 1u) m |= colm << (3 * k);
        }
        pm[b] = m;
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // float2 index of this lane's (pixel, k-pair 0) in a halo buffer
    const int b_frag = lh * 4 * RS + halo + wn * (HB_N / 2) + l31;

    // B fragments of one tap: NI pixels x 4 k-pairs, 8 bytes each
    auto read_b = [&](f32x2 (&dst)[NI][4], const float* buf, int tap) {
        const int sh = d.sgn * ((tap / 3 - 1) * d.W + (tap % 3 - 1));
        const f32x2* p = (const f32x2*)buf + b_frag + sh;
#pragma unroll
        for (int b = 0; b < NI; ++b)
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) dst[b][jp] = p[jp * RS + b * 32];
    };

    float areg[2][MI][8];
    f32x2 breg[2][NI][4];

    // prologue: chunk 0 halo, tap 0 weights and fragments
    load_b(0);
    load_a(areg[0], 0, 0);
    store_b(0, Bs(0));
    __syncthreads();
    read_b(breg[0], Bs(0), 0);

    // one chunk = 9 taps.  Register sets alternate per tap; 9 is odd, so two chunk bodies of opposite parity
    // are unrolled back to back and every register index stays compile-time.
    auto chunk = [&](int ch, auto parity_tag) {
        constexpr int P = decltype(parity_tag)::value;
        const int c0 = ch * HB_K;
        const float* bcur = Bs(ch & 1);
        float* bnext = Bs((ch + 1) & 1);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int cur = (tap + P) & 1, nxt = cur ^ 1;
            // one tap ahead: weights of (tap+1, ch) or (0, ch+1); the halo of chunk ch+1 at tap HB_TLOAD
            if (tap < 8) load_a(areg[nxt], tap + 1, ch);
            else load_a(areg[nxt], 0, ch + 1);
            if (tap == HB_TLOAD) load_b(c0 + HB_K);
            if (tap < 8) read_b(breg[nxt], bcur, tap + 1);
            __builtin_amdgcn_sched_barrier(0);

            bool okb[NI];
#pragma unroll
            for (int b = 0; b < NI; ++b) okb[b] = (pm[b] >> tap) & 1u;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float bv[NI];
#pragma unroll
                for (int b = 0; b < NI; ++b) bv[b] = okb[b] ? breg[cur][b][j >> 1][j & 1] : 0.f;
#pragma unroll
                for (int a = 0; a < MI; ++a)
#pragma unroll
                    for (int b = 0; b < NI; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[cur][a][j], bv[b], acc[a][b], 0, 0, 0);
            }
            if (tap == 8) {
                __builtin_amdgcn_sched_barrier(0);
                store_b(c0 + HB_K, bnext);
                __syncthreads();
                read_b(breg[nxt], bnext, 0);
            }
        }
    };
    for (int ch = 0; ch < nchunk; ch += 2) {
        chunk(ch, std::integral_constant<int, 0>{});
        if (ch + 1 < nchunk) chunk(ch + 1, std::integral_constant<int, 1>{});
    }
    store_tile<MI, NI, BM, HB_N, 2, 2>(acc, dc, d.M, d.npix, i0, j0, 0);
}

// wt[t][ch][i][16]: k = channel within chunk ch (zero beyond the source channel count).
// element (t, i, c) = transposed ? w[c][i][t] : w[i][c][t]      (w: [Cout][Cin][9])
__global__ void wt3x3_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout, int Cin, int transposed) {
    const int I = transposed ? Cin : Cout, K = transposed ? Cout : Cin;
    const int nchunk = (K + 15) / 16;
    const int64_t n = (int64_t)9 * nchunk * I * 16;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int k16 = e & 15;
        int64_t r = e >> 4;
        const int i = r % I;
        r /= I;
        const int ch = r % nchunk, t = r / nchunk;
        const int c = ch * 16 + k16;
        const int co = transposed ? c : i, ci = transposed ? i : c;
        wt[e] = c < K ? w[((int64_t)co * Cin + ci) * 9 + t] : 0.f;
    }
}

// ---------------------------------------------------------------- split-operand variant
//
// fp32 products on the bf16 matrix pipe (16x the fp32 MFMA rate).  Every fp32 operand is written as
//   a = hi + mid + lo,   hi = bf16_rn(a), mid = bf16_rn(a - hi), lo = bf16_rn(a - hi - mid)
// (|mid| <= 2^-9 |a|, |lo| <= 2^-18 |a|, residual <= 2^-27 |a|) and a product is the six bf16 MFMA terms
//   hi.hi + hi.mid + mid.hi + hi.lo + mid.mid + lo.hi,
// each exact in the fp32 accumulator; the dropped terms (mid.lo, lo.mid, lo.lo, residuals) are <= 2^-25 |a b|,
// i.e. below one fp32 rounding of the product.  Accumulation is fp32, as in the fp32 MFMA.
// One v_mfma_f32_32x32x16_bf16 covers the whole 16-channel chunk, so a (32x32 tile, tap) costs 6 x 32 cycles
// instead of 8 x 64.  Operand maps: lane (r, h) holds k = 8h..8h+7 as 8 bf16 = 16 bytes:
//   A: planes of the re-laid weights wt[tap][chunk][plane][row][16 bf16], one 16-B load per plane;
//   B: LDS halo rows [plane][h][pixel][8 bf16]: one ds_read_b128 per plane at the tap-shifted pixel.
// The split of the activations happens once per element on the way into LDS.
// WM waves along the output rows (32 rows each), 4/WM along the pixels; BM = 32*WM.
template <int WM, int HB_N, bool TF>
__global__ __launch_bounds__(NT, 2) void conv3x3_split_kernel(HaloDesc d, OutDesc dc) {
    constexpr int BM = 32 * WM, WN = 4 / WM, NI = HB_N / (32 * WN);
    extern __shared__ __align__(16) float lds[];      // B[2][3 planes][2 k-octets][RS] x 16 bytes
    const int RS = d.RS;
    auto Bs = [&](int buf) -> u32x4* { return (u32x4*)lds + buf * (6 * RS); };

    const int mt = (d.M + BM - 1) / BM, nt = (d.npix + HB_N - 1) / HB_N;
    const int tile = xcd_remap(blockIdx.x, mt * nt);
    const int i0 = (tile % mt) * BM, j0 = (tile / mt) * HB_N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lh = lane >> 5;
    const int halo = d.W + 1;
    const int nchunk = (d.C + HB_K - 1) / HB_K;

    // ---- halo staging: thread e owns LDS column e of all 16 channel rows
    const __amdgpu_buffer_rsrc_t rsrc_b = make_rsrc(d.src, d.nsrc);
    int boff;
    bool bok;
    {
        const int g = j0 - halo + tid;
        bok = tid < RS && g >= 0 && g < d.npix;
        const uint32_t gg = bok ? (uint32_t)g : 0u;
        const uint32_t n = d.dHW.div(gg);
        boff = bok ? (int)((n * (uint32_t)d.C * (uint32_t)d.HW + (gg - n * (uint32_t)d.HW)) * 4u) : OOB;
    }
    const int chw4 = d.HW * 4;
    float bst[HB_K];
    auto load_b = [&](int c0) {
#pragma unroll
        for (int m = 0; m < HB_K; ++m)
            bst[m] = __uint_as_float(
                __builtin_amdgcn_raw_buffer_load_b32(rsrc_b, c0 + m < d.C ? boff : OOB, (c0 + m) * chw4, 0));
    };
    auto store_b = [&](int c0, u32x4* dst) {
        if (tid < RS) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                u32x4 hi, mid, lo;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float x[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int m = 8 * g + 2 * q + u;
                        float t = bst[m];
                        if constexpr (TF) {
                            const int c = c0 + m < d.C ? c0 + m : 0;
                            t = fmaf(t, d.scale[c], d.shift[c]);
                            t = d.relu ? fmaxf(t, 0.f) : t;
                            t = bok ? t : 0.f;
                        }
                        x[u] = t;
                    }
                    uint32_t h, mm, l;
                    split3(x[0], x[1], h, mm, l);
                    hi[q] = h; mid[q] = mm; lo[q] = l;
                }
                dst[(0 * 2 + g) * RS + tid] = hi;
                dst[(1 * 2 + g) * RS + tid] = mid;
                dst[(2 * 2 + g) * RS + tid] = lo;
            }
        }
    };

    // ---- weights: lane (row, h) holds k = 8h..8h+7 of its row, 16 bytes per plane
    const __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc(d.wt, d.nwt);
    const int row = i0 + wm * 32 + l31;
    const int aoff = row < d.M ? row * 32 + lh * 16 : OOB;
    const int aplane = d.M * 32;                       // bytes per (tap, chunk, plane) slab
    auto load_a = [&](u32x4 (&dst)[3], int tap, int ch) {
        const int vo = ch < nchunk ? aoff : OOB;
#pragma unroll
        for (int p = 0; p < 3; ++p)
            dst[p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, vo, ((tap * nchunk + ch) * 3 + p) * aplane, 0);
    };

    // ---- per-lane tap-validity masks of the NI pixels this lane feeds to the MFMA B operand
    uint32_t pm[NI];
#pragma unroll
    for (int b = 0; b < NI; ++b) {
        const int j = j0 + wn * (HB_N / WN) + b * 32 + l31;
        uint32_t m = 0;
        if (j < d.npix) {
            const uint32_t n = d.dHW.div((uint32_t)j);
            const uint32_t r = (uint32_t)j - n * (uint32_t)d.HW;
            const int y = (int)d.dW.div(r), x = (int)r - y * d.W;
            uint32_t rowm = 0, colm = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                rowm |= ((unsigned)(y + d.sgn * (k - 1)) < (unsigned)d.H ? 1u : 0u) << k;
                colm |= ((unsigned)(x + d.sgn * (k - 1)) < (unsigned)d.W ? 1u : 0u) << k;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if ((rowm >> k) & 1u) m |= colm << (3 * k);
        }
        pm[b] = m;
    }

    f32x16 acc[1][NI];
#pragma unroll
    for (int b = 0; b < NI; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][b][r] = 0.f;

    const int b_frag = lh * RS + halo + wn * (HB_N / WN) + l31;   // u32x4 index inside plane 0
    auto read_b = [&](u32x4 (&dst)[3], const u32x4* buf, int tap, int b) {
        const int sh = d.sgn * ((tap / 3 - 1) * d.W + (tap % 3 - 1));
        const u32x4* p = buf + b_frag + sh + b * 32;
#pragma unroll
        for (int q = 0; q < 3; ++q) dst[q] = p[q * 2 * RS];
    };

    u32x4 areg[2][3];
    u32x4 bfr[2][3];

    load_b(0);
    load_a(areg[0], 0, 0);
    store_b(0, Bs(0));
    __syncthreads();
    read_b(bfr[0], Bs(0), 0, 0);

    auto chunk = [&](int ch, auto parity_tag) {
        constexpr int P = decltype(parity_tag)::value;
        const int c0 = ch * HB_K;
        const u32x4* bcur = Bs(ch & 1);
        u32x4* bnext = Bs((ch + 1) & 1);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int acur = (tap + P) & 1, anxt = acur ^ 1;
            if (tap < 8) load_a(areg[anxt], tap + 1, ch);
            else load_a(areg[anxt], 0, ch + 1);
            if (tap == HB_TLOAD) load_b(c0 + HB_K);
#pragma unroll
            for (int b = 0; b < NI; ++b) {
                const int fcur = (P * 9 * NI + tap * NI + b) & 1, fnxt = fcur ^ 1;
                // next fragment: (tap, b+1) or (tap+1, 0); the one after the chunk's last is read behind the barrier
                if (b + 1 < NI) read_b(bfr[fnxt], bcur, tap, b + 1);
                else if (tap < 8) read_b(bfr[fnxt], bcur, tap + 1, 0);
                __builtin_amdgcn_sched_barrier(0);
                const bool ok = (pm[b] >> tap) & 1u;
                bf16x8 bh, bm_, bl;
                {
                    u32x4 t0 = bfr[fcur][0], t1 = bfr[fcur][1], t2 = bfr[fcur][2];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        t0[q] = ok ? t0[q] : 0u;
                        t1[q] = ok ? t1[q] : 0u;
                        t2[q] = ok ? t2[q] : 0u;
                    }
                    bh = __builtin_bit_cast(bf16x8, t0);
                    bm_ = __builtin_bit_cast(bf16x8, t1);
                    bl = __builtin_bit_cast(bf16x8, t2);
                }
                const bf16x8 ah = __builtin_bit_cast(bf16x8, areg[acur][0]);
                const bf16x8 am = __builtin_bit_cast(bf16x8, areg[acur][1]);
                const bf16x8 al = __builtin_bit_cast(bf16x8, areg[acur][2]);
                f32x16 c = acc[0][b];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);   // small terms first
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm_, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm_, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
                acc[0][b] = c;
                __builtin_amdgcn_sched_barrier(0);
            }
            if (tap == 8) {
                store_b(c0 + HB_K, bnext);
                __syncthreads();
                read_b(bfr[((P + 1) * 9 * NI) & 1], bnext, 0, 0);
            }
        }
    };
    for (int ch = 0; ch < nchunk; ch += 2) {
        chunk(ch, std::integral_constant<int, 0>{});
        if (ch + 1 < nchunk) chunk(ch + 1, std::integral_constant<int, 1>{});
    }
    store_tile<1, NI, BM, HB_N, WM, WN>(acc, dc, d.M, d.npix, i0, j0, 0);
}

// (split-operand weights: wt[t][ch][plane][i][16 bf16], the shared re-layout of conv_common.h's WPrepJob with 9 taps)

template <int WM, int HB_N, bool TF>
static void launch_split(const HaloDesc& d, const OutDesc& dc_in, hipStream_t st) {
    constexpr int BM = 32 * WM;
    const int mt = cdiv(d.M, BM), nt = cdiv(d.npix, HB_N);
    const size_t lds_bytes = (size_t)2 * 6 * d.RS * 16;
    OutDesc dc = dc_in;
    if (!dc.accumulate && !dc.bias) {                 // a forward convolution: its BatchNorm's sums ride in the epilogue
        dc.sg = nt * (4 / WM);
        dc.stats = epi_stats_take(d.M, dc.sg, &dc.stats_shift);
    }
    if (!dc.accumulate) dc.st_aux = store_policy(dc.n * 4);
    auto kern = conv3x3_split_kernel<WM, HB_N, TF>;
    hipLaunchKernelGGL(kern, dim3(mt * nt), dim3(NT), lds_bytes, st, d, dc);
}

template <int BM, int HB_N, bool TF>
static void launch_halo(const HaloDesc& d, const OutDesc& dc, hipStream_t st) {
    const int mt = cdiv(d.M, BM), nt = cdiv(d.npix, HB_N);
    const size_t lds_bytes = sizeof(float) * 2 * HB_K * d.RS;
    auto kern = conv3x3_halo_kernel<BM, HB_N, TF>;
    hipLaunchKernelGGL(kern, dim3(mt * nt), dim3(NT), lds_bytes, st, d, dc);
}

}  // namespace scat

using namespace scat;

// re-laid weights: the contraction side (Cin forward, Cout data-gradient) is padded to a multiple of 16
extern "C" int64_t scat_conv3x3_s1_ws(int Cout, int Cin) {
    return (int64_t)9 * ((Cout + 15) / 16 * 16) * ((Cin + 15) / 16 * 16) * 6;   // 3 bf16 terms (>= one fp32)
}

// transposed = 0: dst[B,Cout,H,W] = conv3x3(src[B,Cin,H,W], w)            (Csrc = Cin,  Cdst = Cout)
// transposed = 1: dst[B,Cin,H,W]  = data gradient of that conv from src = dy[B,Cout,H,W]  (Csrc = Cout, Cdst = Cin)
// w is always the forward weight [Cout][Cin][3][3].
extern "C" int scat_conv3x3_s1(const float* src, const float* w, float* dst, int B, int Cin, int H, int W, int Cout,
                               int transposed, const float* in_scale, const float* in_shift, int in_relu,
                               int accumulate, void* ws, int64_t ws_bytes, int w_ready, void* stream) {
    int OH, OW;
    if (int e = check_geom("scat_conv3x3_s1", B, Cin, H, W, Cout, 3, 3, 1, 1, &OH, &OW)) return e;
    SCAT_REQUIRE(src && w && dst, SCAT_E_ARG, "scat_conv3x3_s1: null pointer");
    SCAT_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), SCAT_E_ARG, "scat_conv3x3_s1: scale/shift pair");
    SCAT_REQUIRE(!(transposed && in_scale), SCAT_E_ARG, "scat_conv3x3_s1: no input transform on the data gradient");
    const int Csrc = transposed ? Cout : Cin, Cdst = transposed ? Cin : Cout;
    SCAT_REQUIRE(W + 1 <= 64, SCAT_E_SHAPE, "scat_conv3x3_s1: width > 63 (halo row would not fit one thread per column)");
    SCAT_REQUIRE(ws && ws_bytes >= scat_conv3x3_s1_ws(Cout, Cin), SCAT_E_WORKSPACE, "scat_conv3x3_s1: workspace too small");
    SCAT_REQUIRE(fits_i32((int64_t)B * Csrc * H * W * 4) && fits_i32((int64_t)B * Cdst * H * W * 4) &&
                     fits_i32((int64_t)Cout * Cin * 9 * 4),
                 SCAT_E_SHAPE, "scat_conv3x3_s1: tensor exceeds 32-bit byte offsets");
    hipStream_t st = (hipStream_t)stream;
    const bool split = math_mode() == 1;
    const int64_t nw = (int64_t)9 * ((Csrc + 15) / 16) * Cdst * 16;
    const int blocks = (int)((nw + 255) / 256 < 2048 ? (nw + 255) / 256 : 2048);
    SCAT_REQUIRE(!w_ready || split, SCAT_E_ARG, "scat_conv3x3_s1: prepared weights exist for split products only");
    if (split) {
        if (!w_ready) wprep_launch(wprep_job(w, ws, Cdst, Csrc, transposed, 3, 3, 9, 3, 0, 0, 1), st);
    } else
        hipLaunchKernelGGL(wt3x3_kernel, dim3(blocks), dim3(256), 0, st, w, (float*)ws, Cout, Cin, transposed);

    HaloDesc d{};
    d.src = src; d.wt = (const float*)ws; d.scale = in_scale; d.shift = in_shift; d.relu = in_scale ? in_relu : 0;
    d.C = Csrc; d.M = Cdst; d.H = H; d.W = W; d.HW = H * W; d.npix = B * H * W;
    d.sgn = transposed ? -1 : 1;
    d.dHW = FastDiv::make(H * W); d.dW = FastDiv::make(W);
    d.nsrc = (int64_t)B * Csrc * H * W;
    d.nwt = split ? (nw * 6 + 3) / 4 : nw;            // buffer bounds in floats
    OutDesc dc{};
    dc.p = dst; dc.mode = 1; dc.I = Cdst; dc.J = d.npix; dc.C = Cdst; dc.HW = H * W; dc.dHW = FastDiv::make(H * W);
    dc.accumulate = accumulate; dc.n = (int64_t)B * Cdst * H * W;
    // tile: the largest that still gives every CU several workgroups (see pick_cfg in conv.hip); SCAT_TUNE 1..3
    // forces 128x128 / 64x128 / 64x64 for measurements
    auto tiles = [&](int bm, int bn) { return (int64_t)cdiv(Cdst, bm) * cdiv(d.npix, bn); };
    int cfg = (Cdst > 64 && tiles(128, 128) >= 1024) ? 0 : (tiles(64, 128) >= 1024 ? 1 : 2);
    if (tuning() >= 1 && tuning() <= 3) cfg = tuning() - 1;
    const int bm = cfg == 0 ? 128 : 64, bn = cfg == 2 ? 64 : 128;
    d.RS = bn + 2 * (W + 1);
    if (split) {
        if (!(tuning() >= 1 && tuning() <= 3)) {         // measured at batch 96: 64x128 unless the grid gets thin
            static const int thin = diag_env_int("SCAT_C3_THIN", 512);
            cfg = tiles(64, 128) >= thin ? 1 : 2;
            d.RS = (cfg == 2 ? 64 : 128) + 2 * (W + 1);
        }
        // 32 output channels (HRNet's highest-resolution branch, models/hrnet.py:79-144): a 64-row tile would spend half
        // of its MFMAs on rows that do not exist — four wavefronts side by side on one 32-row block instead
        static const int thin32 = diag_env_int("SCAT_C3_M32", 1);
        if (Cdst <= 32 && thin32 && !(tuning() >= 1 && tuning() <= 3)) {
            d.RS = 128 + 2 * (W + 1);
            set_kernel_label("conv3x3_split_32x128x16%s%s", transposed ? "_dgrad" : "", in_scale ? "_tf" : "");
            if (in_scale) launch_split<1, 128, true>(d, dc, st);
            else launch_split<1, 128, false>(d, dc, st);
            SCAT_LAUNCH_CHECK("scat_conv3x3_s1");
            return SCAT_OK;
        }
        set_kernel_label("conv3x3_split_%dx%dx16%s%s", cfg == 0 ? 128 : 64, cfg == 2 ? 64 : 128, transposed ? "_dgrad" : "", in_scale ? "_tf" : "");
        if (in_scale) {
            if (cfg == 0) launch_split<4, 128, true>(d, dc, st);
            else if (cfg == 1) launch_split<2, 128, true>(d, dc, st);
            else launch_split<2, 64, true>(d, dc, st);
        } else {
            if (cfg == 0) launch_split<4, 128, false>(d, dc, st);
            else if (cfg == 1) launch_split<2, 128, false>(d, dc, st);
            else launch_split<2, 64, false>(d, dc, st);
        }
        SCAT_LAUNCH_CHECK("scat_conv3x3_s1");
        return SCAT_OK;
    }
    set_kernel_label("conv3x3_halo_%dx%dx16%s%s", bm, bn, transposed ? "_dgrad" : "", in_scale ? "_tf" : "");
    if (in_scale) {
        if (cfg == 0) launch_halo<128, 128, true>(d, dc, st);
        else if (cfg == 1) launch_halo<64, 128, true>(d, dc, st);
        else launch_halo<64, 64, true>(d, dc, st);
    } else {
        if (cfg == 0) launch_halo<128, 128, false>(d, dc, st);
        else if (cfg == 1) launch_halo<64, 128, false>(d, dc, st);
        else launch_halo<64, 64, false>(d, dc, st);
    }
    SCAT_LAUNCH_CHECK("scat_conv3x3_s1");
    return SCAT_OK;
}

// fp32 MFMA contraction engine for gfx950 (MI355X).
//
// One workgroup = 256 threads = 4 wavefronts (64 lanes) computing a BM x BN tile of
//     C[i][j] = sum_k A(i,k) * B(k,j)
// with v_mfma_f32_32x32x2_f32 (exact fp32 fma chain, 64 FLOP/clk/SIMD — the chip's fp32
// peak; there is no xf32/TF32 on gfx950).  Operands are staged HBM -> registers -> LDS as
// k-major tiles  T[k][i]  (row stride BI+4 floats), so the MFMA operand fetch
// "lane l wants X[i = l&31][k = l>>5]" is one conflict-free ds_read_b32 per 32-lane half.
// Staging is register double-buffered: the global loads of K-step t+1 are in flight while
// the MFMAs of K-step t issue; one s_barrier per K-step.
//
// The two operands come from "loaders":
//   MatLoader     - a strided matrix (weights, Linear activations, split-K slabs)
//   GatherLoader  - an NCHW feature map seen through conv index arithmetic
//                   (im2col on the fly; zero padding; optional fused per-channel
//                   scale/shift/ReLU = the previous BatchNorm applied in the consumer's
//                   prologue).  The same loader serves forward (pixel = tile column),
//                   data-gradient (transposed-conv arithmetic, stride 1 or 2) and
//                   weight-gradient (pixel = contraction index).
//
// Reference semantics replaced: torch.nn.Conv2d / nn.Linear as called from
// models/resnet.py:62-98,103-162 and models/vision_transformer.py:28-79 (fp32).
#pragma once
#include <type_traits>

#include "common.h"

namespace scat {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NT = 256;   // threads per workgroup
constexpr int LPAD = 4;   // LDS row padding (floats): k-fast writes become <=2-way, rows stay 16-B aligned

// ---------------------------------------------------------------- descriptors

struct MatDesc {          // elem(i,k) = p[z*sz + i*si + k*sk]
    const float* p;
    int64_t si, sk, sz;
    int I, K;
    int64_t n;            // floats addressable from p + z*sz (buffer bounds)
};

struct GatherDesc {       // elem(pix, ct) of an NCHW tensor through conv arithmetic
    const float* p;
    const float* scale;   // optional per-source-channel fused transform (nullptr = identity)
    const float* shift;
    int relu;
    int C, H, W;          // source tensor: [Nimg][C][H][W]
    int PH, PW;           // pixel grid the GEMM index runs over
    int a, b, c0;         // source row  t = y*a + kh*b + c0
    int c0x;              // source col  t = x*a + kw*b + c0x
    int npix;             // Nimg*PH*PW
    int nct;              // C*KH*KW
    FastDiv dPHW, dPW;
    int64_t n;            // floats in the source tensor (buffer bounds)
};

struct OutDesc {
    float* p;
    int mode;             // 0: C[z*sz + i*si + j*sj]   1: NCHW, j = pixel: [(j/HW)][i][j%HW]
                          // 2: NCHW, j = pixel of a strided sub-grid [QH][QW] -> (qy*sub_s + sub_y, qx*sub_s + sub_x)
    int W, QW, sub_s, sub_y, sub_x;
    FastDiv dQHW, dQW;
    int64_t si, sj, sz;
    int I, J;
    int C, HW;
    FastDiv dHW;
    const float* bias;    // optional
    int bias_mode;        // 1: bias[i]  2: bias[j]
    int accumulate;       // C += result
    int64_t n;            // floats addressable from p (+ z*sz in mode 0): buffer bounds
    float* stats;         // optional: per-row (sum, sum of squares) of every wavefront's live columns of the tile,
    int sg;               //   stats[(row * sg + group) * 2 + {0, 1}], group = tile column * WN + wn  (BatchNorm statistics
                          //   of a convolution's output without reading it back: scat_epilogue_stats_arm)
    int st_aux;                 // cache policy of the epilogue's stores (store_tile): 0 default, 16 sc1, 2 nt
    const float* stats_shift;   // optional per-row reference c[row]: the sums are of (x - c) and (x - c)^2 — fp32 partials of
                                // x^2 cancel catastrophically in E[x^2] - mean^2 when |mean| >> sigma (scat_epilogue_stats_arm_shift)
    // BatchNorm-backward epilogue (store_tile<..., BNB = true>, accumulate only; scat_epilogue_bnb_arm): the tensor being
    // completed is the gradient of  relu(bn(x) + residual)  — mask it with the output's sign bits (bit e % 4 of byte
    // e / 4), store the MASKED gradient g, and leave per row (= channel) and column group the sums of g and
    // g * (x - bnb_mean[row]) in bnb_part[(row * bnb_sg + group) * 2 + {0, 1}]: the reduction pass of that BatchNorm's
    // backward without reading the gradient back
    const float* bnb_x;
    const uint8_t* bnb_mask;
    const float* bnb_mean;
    float* bnb_part;
    int bnb_sg;
};

// ---------------------------------------------------------------- loaders
//
// All global reads are raw buffer loads through a wave-uniform SRSRC descriptor: an element that
// must read as zero (tile edge, conv padding, wrong stride-2 parity) gets voffset = 2^31, which the
// hardware bounds check turns into 0 — no exec-mask branches, no per-load s_waitcnt, every load of a
// K-step is in flight together.  The fused BatchNorm+ReLU transform is applied when the registers are
// written to LDS (after the MFMAs of the previous K-step), masked so padding stays zero.

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int OOB = (int)0x80000000;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, int64_t nfloats) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)(nfloats * 4), 0x00020000);
}
__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}
__device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t r, int byte_off) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// VEC = 1: one float per load.  VEC = 4 (KFAST only): 16-B loads along k; needs sk == 1, K % 4 == 0,
// 16-B aligned rows (host checks).
template <int BI, int BK, bool KFAST, int VEC = 1>
struct MatLoader {
    static constexpr int NE = BI * BK / NT;
    static constexpr int NV = NE / VEC;
    static constexpr int TPR = BK / VEC;          // threads per row along k (KFAST)
    using Desc = MatDesc;
    float v[NE];
    int i_t, k_t;
    __amdgpu_buffer_rsrc_t rs;

    __device__ __forceinline__ void init(const MatDesc& d, int i0, int z) {
        const int tid = threadIdx.x;
        if (KFAST) { k_t = (tid % TPR) * VEC; i_t = tid / TPR; }
        else       { i_t = tid % BI; k_t = tid / BI; }
        rs = make_rsrc(d.p + (int64_t)z * d.sz, d.n);
        i_t += i0;
#pragma unroll
        for (int r = 0; r < NE; ++r) v[r] = 0.f;
    }
    __device__ __forceinline__ void load(const MatDesc& d, int k0, int kend) {
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            int i = i_t + (KFAST ? r * (NT / TPR) : 0);
            int k = k0 + k_t + (KFAST ? 0 : r * (NT / BI));
            bool ok = (i < d.I) && (k < kend);
            int off = ok ? (i * (int)d.si + k * (int)d.sk) * 4 : OOB;
            if (VEC == 4) {
                float4 t = bload4(rs, off);
                v[4 * r] = t.x; v[4 * r + 1] = t.y; v[4 * r + 2] = t.z; v[4 * r + 3] = t.w;
            } else {
                v[r] = bload(rs, off);
            }
        }
    }
    __device__ __forceinline__ void store(const MatDesc&, float* lds) const {
        const int tid = threadIdx.x;
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            int il = KFAST ? tid / TPR + r * (NT / TPR) : tid % BI;
            int kl = KFAST ? (tid % TPR) * VEC : tid / BI + r * (NT / BI);
#pragma unroll
            for (int q = 0; q < VEC; ++q) lds[(kl + q) * (BI + LPAD) + il] = v[VEC * r + q];
        }
    }
};

// Conv-arithmetic loader. KH,KW compile-time so tap decode is mul-shift. D2 = divisor 2 (data-gradient
// of a stride-2 conv: only taps of matching parity contribute).  PIXK: the pixel index is the
// contraction index (weight gradient).  V4: 16-B loads along 4 consecutive pixels of the flat NCHW
// plane — stride-1 arithmetic only (a == 1), (PH*PW) % 4 == 0.  For KH*KW > 1 the vector is shifted by
// the tap offset (4-B aligned dwordx4) and each of the 4 pixels carries its own tap-validity mask.
// TF: the fused per-channel scale/shift(+ReLU) input transform is compiled in.
template <int BI, int BK, int KH, int KW, bool D2, bool PIXK, bool V4 = false, bool TF = false>
struct GatherLoader {
    static constexpr int KK = KH * KW;
    static constexpr int NE = BI * BK / NT;
    static constexpr int NV = V4 ? NE / 4 : NE;
    static_assert(!V4 || (KK <= 9 && !D2), "V4 is a stride-1 path for up to 3x3 taps");
    static constexpr bool V4M = V4 && KK > 1;     // multi-tap vector path: per-pixel masks
    using Desc = GatherDesc;
    float v[NE];
    unsigned okbits;                              // V4M: 4 bits per element (one per pixel)
    int pixoff;
    uint64_t mask;
    uint32_t pm[V4M ? 4 : 1];                     // V4M: tap mask of each of the 4 pixels
    // PIXK: ct-side state fixed per tile (offset, tap, fused-transform constants)
    int ctoff[PIXK ? NV : 1];
    int cttap[PIXK ? NV : 1];
    float csc[PIXK ? NV : 1], csh[PIXK ? NV : 1];
    // !PIXK: per-K-step channel of each element row, wave-uniform
    float rsc[PIXK ? 1 : NV], rsh[PIXK ? 1 : NV];
    __amdgpu_buffer_rsrc_t rs;

    static __device__ __forceinline__ int tap_off(const GatherDesc& d, int tap) {
        int kh = tap / KW, kw = tap - kh * KW;
        if (D2) return -((kh >> 1) * d.W + (kw >> 1));
        return (kh * d.W + kw) * d.b;
    }

    __device__ __forceinline__ void decode_pix(const GatherDesc& d, int pix) {
        mask = 0;
        pixoff = 0;
        if (pix >= d.npix) return;
        uint32_t n = d.dPHW.div((uint32_t)pix);
        uint32_t r = pix - n * (d.PH * d.PW);
        uint32_t y = d.dPW.div(r);
        uint32_t x = r - y * d.PW;
        int ty0 = (int)y * d.a + d.c0, tx0 = (int)x * d.a + d.c0x;
        uint32_t rowm = 0, colm = 0;
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
            int t = ty0 + kh * d.b;
            bool ok = D2 ? (t >= 0 && !(t & 1) && (t >> 1) < d.H) : ((unsigned)t < (unsigned)d.H);
            rowm |= (ok ? 1u : 0u) << kh;
        }
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) {
            int t = tx0 + kw * d.b;
            bool ok = D2 ? (t >= 0 && !(t & 1) && (t >> 1) < d.W) : ((unsigned)t < (unsigned)d.W);
            colm |= (ok ? 1u : 0u) << kw;
        }
#pragma unroll
        for (int kh = 0; kh < KH; ++kh)
            if ((rowm >> kh) & 1u) mask |= (uint64_t)colm << (kh * KW);
        pixoff = (int)n * d.C * d.H * d.W + (D2 ? ((ty0 >> 1) * d.W + (tx0 >> 1)) : (ty0 * d.W + tx0));
    }

    // V4M: tap masks of 4 consecutive pixels starting at pix (same image: (PH*PW) % 4 == 0, pix % 4 == 0)
    __device__ __forceinline__ void decode4(const GatherDesc& d, int pix) {
        pm[0] = pm[V4M ? 1 : 0] = pm[V4M ? 2 : 0] = pm[V4M ? 3 : 0] = 0;
        pixoff = 0;
        if (pix >= d.npix) return;
        uint32_t n = d.dPHW.div((uint32_t)pix);
        uint32_t r = pix - n * (d.PH * d.PW);
        int y = (int)d.dPW.div(r);
        int x = (int)r - y * d.PW;
        pixoff = (int)n * d.C * d.H * d.W + (y + d.c0) * d.W + (x + d.c0x);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t rowm = 0, colm = 0, m = 0;
#pragma unroll
            for (int kh = 0; kh < KH; ++kh) rowm |= ((unsigned)(y + d.c0 + kh * d.b) < (unsigned)d.H ? 1u : 0u) << kh;
#pragma unroll
            for (int kw = 0; kw < KW; ++kw) colm |= ((unsigned)(x + d.c0x + kw * d.b) < (unsigned)d.W ? 1u : 0u) << kw;
#pragma unroll
            for (int kh = 0; kh < KH; ++kh)
                if ((rowm >> kh) & 1u) m |= colm << (kh * KW);
            pm[V4M ? q : 0] = m;
            if (++x == d.PW) { x = 0; ++y; }
        }
    }
    // one shifted 16-B load; vectors that would start before / end after the tensor (first and last pixel
    // group of the whole tensor only) fall back to 4 bounds-checked scalar loads
    __device__ __forceinline__ void vload_shifted(const GatherDesc& d, int r, int elem_off, uint32_t bits4) {
        const int boff = elem_off * 4;
        if (bits4 == 0) {
            v[4 * r] = v[4 * r + 1] = v[4 * r + 2] = v[4 * r + 3] = 0.f;
        } else if (boff >= 0 && (int64_t)boff + 16 <= d.n * 4) {
            float4 t = bload4(rs, boff);
            v[4 * r] = t.x; v[4 * r + 1] = t.y; v[4 * r + 2] = t.z; v[4 * r + 3] = t.w;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[4 * r + q] = bload(rs, ((bits4 >> q) & 1u) ? boff + 4 * q : OOB);
        }
    }
    __device__ __forceinline__ uint32_t bits_for_tap(int tap) const {
        return ((pm[0] >> tap) & 1u) | (((pm[V4M ? 1 : 0] >> tap) & 1u) << 1) | (((pm[V4M ? 2 : 0] >> tap) & 1u) << 2) |
               (((pm[V4M ? 3 : 0] >> tap) & 1u) << 3);
    }

    // thread -> (pixel-side index, ct-side index) inside the tile, element r
    static constexpr int PV = V4 ? 4 : 1;                                   // pixels per load
    static constexpr int PT = (PIXK ? BK : BI) / PV;                        // threads along the pixel dim
    __device__ __forceinline__ int pix_l() const { return (threadIdx.x % PT) * PV; }
    __device__ __forceinline__ int ct_l(int r) const { return threadIdx.x / PT + r * (NT / PT); }

    __device__ __forceinline__ void init(const GatherDesc& d, int i0, int /*z*/) {
        rs = make_rsrc(d.p, d.n);
        okbits = 0;
#pragma unroll
        for (int r = 0; r < NE; ++r) v[r] = 0.f;
        if (!PIXK) {
            if constexpr (V4M) decode4(d, i0 + pix_l());
            else decode_pix(d, i0 + pix_l());
        } else {
#pragma unroll
            for (int r = 0; r < NV; ++r) {
                int ct = i0 + ct_l(r);
                int c = ct / KK, tap = ct - c * KK;
                bool ok = ct < d.nct;
                cttap[r] = ok ? tap : -1;
                ctoff[r] = c * d.H * d.W + tap_off(d, tap);
                csc[r] = (TF && ok) ? d.scale[c] : 1.f;
                csh[r] = (TF && ok) ? d.shift[c] : 0.f;
            }
        }
    }

    __device__ __forceinline__ void load(const GatherDesc& d, int k0, int kend) {
        okbits = 0;
        if (!PIXK) {
#pragma unroll
            for (int r = 0; r < NV; ++r) {
                // with >= 64 threads along the pixel dim the ct row is the same for the whole wavefront:
                // make that visible so the tap decode and the fused-transform constants go scalar
                int ct = k0 + (PT >= 64 ? __builtin_amdgcn_readfirstlane(ct_l(r)) : ct_l(r));
                int c = ct / KK, tap = ct - c * KK;
                if constexpr (V4M) {
                    const uint32_t bits4 = ct < kend ? bits_for_tap(tap) : 0u;
                    vload_shifted(d, r, pixoff + c * d.H * d.W + tap_off(d, tap), bits4);
                    okbits |= bits4 << (4 * r);
                    if constexpr (TF) {
                        int cc = ct < kend ? c : 0;
                        rsc[r] = d.scale[cc];
                        rsh[r] = d.shift[cc];
                    }
                    continue;
                }
                bool ok = (ct < kend) && ((mask >> tap) & 1ull);
                int off = ok ? (pixoff + c * d.H * d.W + tap_off(d, tap)) * 4 : OOB;
                if (V4) {
                    float4 t = bload4(rs, off);
                    v[4 * r] = t.x; v[4 * r + 1] = t.y; v[4 * r + 2] = t.z; v[4 * r + 3] = t.w;
                } else {
                    v[r] = bload(rs, off);
                }
                okbits |= (ok ? 1u : 0u) << r;
                if constexpr (TF) {
                    int cc = ct < kend ? c : 0;
                    rsc[r] = d.scale[cc];
                    rsh[r] = d.shift[cc];
                }
            }
        } else {
            int pix = k0 + pix_l();
            if constexpr (V4M) {
                decode4(d, pix < kend ? pix : d.npix);
#pragma unroll
                for (int r = 0; r < NV; ++r) {
                    const uint32_t bits4 = cttap[r] >= 0 ? bits_for_tap(cttap[r] & 15) : 0u;
                    vload_shifted(d, r, pixoff + ctoff[r], bits4);
                    okbits |= bits4 << (4 * r);
                }
                return;
            }
            decode_pix(d, pix < kend ? pix : d.npix);
#pragma unroll
            for (int r = 0; r < NV; ++r) {
                bool ok = (cttap[r] >= 0) && ((mask >> (cttap[r] & 63)) & 1ull);
                int off = ok ? (pixoff + ctoff[r]) * 4 : OOB;
                if (V4) {
                    float4 t = bload4(rs, off);
                    v[4 * r] = t.x; v[4 * r + 1] = t.y; v[4 * r + 2] = t.z; v[4 * r + 3] = t.w;
                } else {
                    v[r] = bload(rs, off);
                }
                okbits |= (ok ? 1u : 0u) << r;
            }
        }
    }

    __device__ __forceinline__ void store(const GatherDesc& d, float* lds) const {
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            float t[PV];
#pragma unroll
            for (int q = 0; q < PV; ++q) {
                const bool ok = V4M ? (okbits >> (4 * r + q)) & 1u : (okbits >> r) & 1u;
                float x = v[PV * r + q];
                if constexpr (TF) {
                    x = fmaf(x, PIXK ? csc[r] : rsc[r], PIXK ? csh[r] : rsh[r]);
                    x = d.relu ? fmaxf(x, 0.f) : x;
                }
                // a shifted vector reads real neighbours where the conv wants padding: mask per pixel;
                // otherwise the bounds-checked load already returned 0 (only the transform needs re-masking)
                if constexpr (TF || V4M) x = ok ? x : 0.f;
                t[q] = x;
            }
            const int pl = pix_l(), cl = ct_l(r);
            if (!PIXK) {          // tile[k = ct][i = pixel]: pixels contiguous
                float* dst = lds + cl * (BI + LPAD) + pl;
                if (V4) *(float4*)dst = make_float4(t[0], t[1], t[2], t[3]);
                else dst[0] = t[0];
            } else {              // tile[k = pixel][i = ct]
#pragma unroll
                for (int q = 0; q < PV; ++q) lds[(pl + q) * (BI + LPAD) + cl] = t[q];
            }
        }
    }
};

// ---------------------------------------------------------------- kernel

// XCD-aware, bijective remap of the linear workgroup id: the 8 XCDs (private L2 each) get
// contiguous chunks of the tile list, so workgroups that share an operand panel hit one L2.
__device__ __forceinline__ int xcd_remap(int id, int n) {
    int q = n >> 3, r = n & 7, x = id & 7, s = id >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s;
}

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N-1>{})
template <int I, int N, class F>
__device__ __forceinline__ void static_for_impl(F& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_impl<I + 1, N>(f);
    }
}
template <int N, class F>
__device__ __forceinline__ void static_for(F f) { static_for_impl<0, N>(f); }

// ---------------------------------------------------------------- epilogue
// C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
// Branch-free: 32-bit element offsets, raw buffer stores whose out-of-range lanes (tile edge) are
// dropped by the hardware bounds check; the (accumulate, bias) variants are separate straight-line
// copies selected once by uniform branches.
template <int MI, int NI, int BM, int BN, int WM, int WN, bool BNB = false>
__device__ __forceinline__ void store_tile(f32x16 (&acc)[MI][NI], const OutDesc& dc, int M, int N, int i0, int j0,
                                           int z) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lh = lane >> 5;
    const __amdgpu_buffer_rsrc_t rc = make_rsrc(dc.p + (dc.mode != 0 ? 0 : (int64_t)z * dc.sz), dc.n);
    const int rstride = dc.mode != 0 ? dc.HW : (int)dc.si;
    int coloff[NI];
    bool colok[NI];
    float bj[NI];
#pragma unroll
    for (int b = 0; b < NI; ++b) {
        const int j = j0 + wn * (BN / WN) + b * 32 + l31;
        colok[b] = j < N;
        const int jc = colok[b] ? j : 0;
        if (dc.mode == 1) {
            uint32_t n = dc.dHW.div((uint32_t)jc);
            coloff[b] = (int)n * dc.C * dc.HW + (jc - (int)n * dc.HW);
        } else if (dc.mode == 2) {
            uint32_t n = dc.dQHW.div((uint32_t)jc);
            uint32_t r = jc - n * dc.dQHW.d;
            uint32_t qy = dc.dQW.div(r), qx = r - qy * dc.QW;
            coloff[b] = (int)n * dc.C * dc.HW + ((int)qy * dc.sub_s + dc.sub_y) * dc.W + (int)qx * dc.sub_s + dc.sub_x;
        } else {
            coloff[b] = jc * (int)dc.sj;
        }
        bj[b] = (dc.bias && dc.bias_mode == 2) ? dc.bias[jc] : 0.f;
    }
    const int ibase = i0 + wm * (BM / WM) + 4 * lh;
    if constexpr (BNB) {
        // BatchNorm-backward epilogue (OutDesc::bnb_*): C = mask(C_old + acc), row sums of g and g * (x - mean) per column
        // group.  Two rows at a time (registers: the caller's 64 accumulators stay live), a row quad's sums are reduced
        // over the 32 lanes of the row and written before the next quad.
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(dc.bnb_x, dc.n);
        const __amdgpu_buffer_rsrc_t rm =
            __builtin_amdgcn_make_buffer_rsrc((void*)dc.bnb_mask, 0, (int)((dc.n + 3) / 4), 0x00020000);
        const int grp = (j0 / BN) * WN + wn;
#pragma unroll
        for (int a = 0; a < MI; ++a) {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                float s4[4] = {0.f, 0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f};
                static_for<2>([&](auto h_tag) {
                    constexpr int H2 = 2 * decltype(h_tag)::value;
                    int voff[2][NI];
                    float val[2][NI], old[2][NI], xv[2][NI], mu[2];
                    uint32_t mb[2][NI];
#pragma unroll
                    for (int r2 = 0; r2 < 2; ++r2) {
                        const int i = ibase + a * 32 + H2 + r2 + 8 * rq;
                        const bool rowok = i < M;
                        mu[r2] = dc.bnb_mean[rowok ? i : 0];
#pragma unroll
                        for (int b = 0; b < NI; ++b) {
                            voff[r2][b] = (rowok && colok[b]) ? (coloff[b] + i * rstride) * 4 : OOB;
                            val[r2][b] = acc[a][b][rq * 4 + H2 + r2];
                        }
                    }
#pragma unroll
                    for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                        for (int b = 0; b < NI; ++b) {
                            old[r2][b] = bload(rc, voff[r2][b]);
                            xv[r2][b] = bload(rx, voff[r2][b]);
                            mb[r2][b] = __builtin_amdgcn_raw_buffer_load_b8(rm, voff[r2][b] == OOB ? OOB : (voff[r2][b] >> 4), 0, 0);
                        }
#pragma unroll
                    for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                        for (int b = 0; b < NI; ++b) {
                            const bool live = voff[r2][b] != OOB && ((mb[r2][b] >> ((voff[r2][b] >> 2) & 3)) & 1u);
                            const float g = live ? val[r2][b] + old[r2][b] : 0.f;
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(g), rc, voff[r2][b], 0, 0);
                            s4[H2 + r2] += g;
                            q4[H2 + r2] = fmaf(g, xv[r2][b] - mu[r2], q4[H2 + r2]);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                });
                // reduce-scatter over the 32 lanes of a row: lane bit 4 keeps rows {2, 3} or {0, 1}, bit 3 the odd or the even
                // one, then a butterfly over the eight lanes that hold the same row
                {
                    const bool up = (lane & 16) != 0;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const float ks = up ? s4[k + 2] : s4[k], xs = up ? s4[k] : s4[k + 2];
                        const float kq = up ? q4[k + 2] : q4[k], xq = up ? q4[k] : q4[k + 2];
                        s4[k] = ks + __shfl_xor(xs, 16);
                        q4[k] = kq + __shfl_xor(xq, 16);
                    }
                }
                {
                    const bool up = (lane & 8) != 0;
                    const float ks = up ? s4[1] : s4[0], xs = up ? s4[0] : s4[1];
                    const float kq = up ? q4[1] : q4[0], xq = up ? q4[0] : q4[1];
                    s4[0] = ks + __shfl_xor(xs, 8);
                    q4[0] = kq + __shfl_xor(xq, 8);
                }
#pragma unroll
                for (int bit = 4; bit >= 1; bit >>= 1) {
                    s4[0] += __shfl_xor(s4[0], bit);
                    q4[0] += __shfl_xor(q4[0], bit);
                }
                const int rr = ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1);
                const int i = ibase + a * 32 + rr + 8 * rq;
                if ((lane & 7) == 0 && i < M) {
                    float* o = dc.bnb_part + ((int64_t)i * dc.bnb_sg + grp) * 2;
                    o[0] = s4[0];
                    o[1] = q4[0];
                }
            }
        }
        return;
    }
    auto emit = [&](auto acc_tag, auto bias_tag) {
        constexpr bool ACC = decltype(acc_tag)::value;
        constexpr bool BIASI = decltype(bias_tag)::value;
#pragma unroll
        for (int a = 0; a < MI; ++a) {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {          // 4 registers = 4 consecutive rows
                int voff[4][NI];
                float val[4][NI];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int i = ibase + a * 32 + rr + 8 * rq;
                    const bool rowok = i < M;
                    const float bi = BIASI ? dc.bias[rowok ? i : 0] : 0.f;
#pragma unroll
                    for (int b = 0; b < NI; ++b) {
                        voff[rr][b] = (rowok && colok[b]) ? (coloff[b] + i * rstride) * 4 : OOB;
                        val[rr][b] = acc[a][b][rq * 4 + rr] + bj[b] + bi;
                    }
                }
                if constexpr (ACC) {
                    float old[4][NI];
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                        for (int b = 0; b < NI; ++b) old[rr][b] = bload(rc, voff[rr][b]);
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                        for (int b = 0; b < NI; ++b) val[rr][b] += old[rr][b];
                }
                // cache policy of the output stores (OutDesc::st_aux): 0 = default (the line stays in the XCD's L2),
                // 16 = sc1 (written through, the line is dropped from L2), 2 = nt
                if (dc.st_aux == 16) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                        for (int b = 0; b < NI; ++b)
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val[rr][b]), rc, voff[rr][b], 0, 16);
                } else if (dc.st_aux == 2) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                        for (int b = 0; b < NI; ++b)
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val[rr][b]), rc, voff[rr][b], 0, 2);
                } else {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                        for (int b = 0; b < NI; ++b)
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val[rr][b]), rc, voff[rr][b], 0, 0);
                }
            }
        }
    };
    const bool biasi = dc.bias && dc.bias_mode == 1;
    if (dc.accumulate) {
        if (biasi) emit(std::true_type{}, std::true_type{});
        else emit(std::true_type{}, std::false_type{});
    } else {
        if (biasi) emit(std::false_type{}, std::true_type{});
        else emit(std::false_type{}, std::false_type{});
    }
    if (dc.stats) {
        // Row sums of the tile while it is still in registers.  A row's columns sit in NI registers x 32 lanes: add the
        // NI registers, then a reduce-scatter over the 32 lanes (step with lane bit k: a lane keeps the half of its
        // registers whose index bit matches and adds the partner's) — 16 + 8 + 4 + 2 + 1 exchanges per quantity instead
        // of 16 x 5; lane l ends with the total of accumulator register (l >> 1) & 15.
        const int grp = (j0 / BN) * WN + wn;
#pragma unroll
        for (int a = 0; a < MI; ++a) {
            float s[16], q[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float cref = 0.f;
                if (dc.stats_shift) {
                    const int ir = ibase + a * 32 + (r & 3) + 8 * (r >> 2);
                    cref = finite_or_zero(dc.stats_shift[ir < M ? ir : 0]);
                }
                float ss = 0.f, qq = 0.f;
#pragma unroll
                for (int b = 0; b < NI; ++b) {
                    const float v = colok[b] ? acc[a][b][r] - cref : 0.f;
                    ss += v;
                    qq = fmaf(v, v, qq);
                }
                s[r] = ss; q[r] = qq;
            }
            static_for<4>([&](auto st_tag) {
                constexpr int ST = decltype(st_tag)::value, H = 8 >> ST, BIT = 16 >> ST;
                const bool up = (lane & BIT) != 0;
#pragma unroll
                for (int k = 0; k < H; ++k) {
                    const float ks = up ? s[k + H] : s[k], xs = up ? s[k] : s[k + H];
                    const float kq = up ? q[k + H] : q[k], xq = up ? q[k] : q[k + H];
                    s[k] = ks + __shfl_xor(xs, BIT);
                    q[k] = kq + __shfl_xor(xq, BIT);
                }
            });
            s[0] += __shfl_xor(s[0], 1);
            q[0] += __shfl_xor(q[0], 1);
            const int r = (lane >> 1) & 15;
            const int i = ibase + a * 32 + (r & 3) + 8 * (r >> 2);
            if ((lane & 1) == 0 && i < M) {
                float* o = dc.stats + ((int64_t)i * dc.sg + grp) * 2;
                o[0] = s[0];
                o[1] = q[0];
            }
        }
    }
}

// PD = prefetch distance in K-steps.  PD = 1: the loads of step t+1 fly under the MFMAs of step t.
// PD = 2: a second register set keeps the loads of step t+2 in flight as well (twice the bytes in flight per
// CU — what the HBM-bound short-K layers need, MI355X wants >= 64 KiB in flight per CU to hide an HBM miss).
// one (tile, K-slice) of a contraction: the body of gemm_kernel, and of the grouped launch in linear.hip
template <class LA, class LB, int BM, int BN, int BK, int WM, int WN, int PD = 1>
__device__ __forceinline__ void gemm_tile(const typename LA::Desc& da, const typename LB::Desc& db, const OutDesc& dc,
                                          const int M, const int N, const int K, const int kchunk, const int tile,
                                          const int z) {
    constexpr int MI = BM / WM / 32, NI = BN / WN / 32;
    static_assert(WM * WN == 4 && MI >= 1 && NI >= 1, "wave layout");
    static_assert(PD == 1 || PD == 2, "prefetch distance");
    constexpr int SA = BM + LPAD, SB = BN + LPAD;
    extern __shared__ __align__(16) float lds[];   // 2*BK*(SA+SB) floats: A buffers 0,1 then B buffers 0,1
    auto As = [&](int buf) -> float* { return lds + buf * (BK * SA); };
    auto Bs = [&](int buf) -> float* { return lds + 2 * BK * SA + buf * (BK * SB); };

    const int mt = (M + BM - 1) / BM;
    const int i0 = (tile % mt) * BM, j0 = (tile / mt) * BN;   // m fastest: neighbours share the B panel
    const int kbeg = z * kchunk, kend = min(K, kbeg + kchunk);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lh = lane >> 5;

    LA la[PD];
    LB lb[PD];
#pragma unroll
    for (int s = 0; s < PD; ++s) {
        la[s].init(da, i0, z);
        lb[s].init(db, j0, z);
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nk = (kend - kbeg + BK - 1) / BK;
    // the MFMAs of one K-step on LDS buffer `cur`
    auto mfmas = [&](int cur) {
        const float* as = As(cur) + wm * (BM / WM) + l31 + lh * SA;
        const float* bs = Bs(cur) + wn * (BN / WN) + l31 + lh * SB;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float av[MI], bv[NI];
#pragma unroll
            for (int a = 0; a < MI; ++a) av[a] = as[kk * SA + a * 32];
#pragma unroll
            for (int b = 0; b < NI; ++b) bv[b] = bs[kk * SB + b * 32];
#pragma unroll
            for (int a = 0; a < MI; ++a)
#pragma unroll
                for (int b = 0; b < NI; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
    };
    // K-step t: [issue loads of step t+PD into register set LS] -> MFMAs on LDS[t&1] -> [write step t+1 from
    // register set SS to LDS[(t+1)&1]] -> barrier.  LOAD/STORE are compile-time so the body is branch-free.
    auto kstep = [&](int t, auto ls_tag, auto ss_tag, auto load_tag, auto store_tag) {
        constexpr int LS = decltype(ls_tag)::value, SS = decltype(ss_tag)::value;
        if constexpr (decltype(load_tag)::value) {
            la[LS].load(da, kbeg + (t + PD) * BK, kend);
            lb[LS].load(db, kbeg + (t + PD) * BK, kend);
            // hipcc otherwise sinks the buffer loads below the MFMAs, next to the ds_writes that consume them
            // (seen in the ISA): the loads would then be waited for as soon as they are issued
            __builtin_amdgcn_sched_barrier(0);
        }
        mfmas(t & 1);
        if constexpr (decltype(store_tag)::value) {
            la[SS].store(da, As((t + 1) & 1));
            lb[SS].store(db, Bs((t + 1) & 1));
        }
        __syncthreads();
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, PD - 1>;
    using T_ = std::true_type;
    using F_ = std::false_type;
    if (nk > 0) {
        la[0].load(da, kbeg, kend);
        lb[0].load(db, kbeg, kend);
        if constexpr (PD == 2) {
            if (nk > 1) {
                la[1].load(da, kbeg + BK, kend);
                lb[1].load(db, kbeg + BK, kend);
            }
        }
        la[0].store(da, As(0));
        lb[0].store(db, Bs(0));
    }
    __syncthreads();
    if constexpr (PD == 1) {
        int t = 0;
        for (; t + 1 < nk; ++t) kstep(t, I0{}, I0{}, T_{}, T_{});
        if (nk > 0) kstep(t, I0{}, I0{}, F_{}, F_{});
    } else {
        // step t (even) consumes LDS[0]; its successor t+1 is in register set 1, step t+2 goes to set 0
        int t = 0;
        for (; t + 3 < nk; t += 2) {
            kstep(t, I0{}, I1{}, T_{}, T_{});
            kstep(t + 1, I1{}, I0{}, T_{}, T_{});
        }
        for (; t < nk; ++t) {   // tail (<= 3 steps): generic parity, loads/stores only while tiles remain
            const bool ld = t + 2 < nk, st = t + 1 < nk;
            if (t & 1) {
                if (ld) kstep(t, I1{}, I0{}, T_{}, T_{});
                else if (st) kstep(t, I1{}, I0{}, F_{}, T_{});
                else kstep(t, I1{}, I0{}, F_{}, F_{});
            } else {
                if (ld) kstep(t, I0{}, I1{}, T_{}, T_{});
                else if (st) kstep(t, I0{}, I1{}, F_{}, T_{});
                else kstep(t, I0{}, I1{}, F_{}, F_{});
            }
        }
    }

    store_tile<MI, NI, BM, BN, WM, WN>(acc, dc, M, N, i0, j0, z);
}

template <class LA, class LB, int BM, int BN, int BK, int WM, int WN, int PD = 1>
__global__ __launch_bounds__(NT) void gemm_kernel(typename LA::Desc da, typename LB::Desc db, OutDesc dc,
                                                  int M, int N, int K, int kchunk) {
    const int mt = (M + BM - 1) / BM, nt = (N + BN - 1) / BN;
    gemm_tile<LA, LB, BM, BN, BK, WM, WN, PD>(da, db, dc, M, N, K, kchunk, xcd_remap(blockIdx.x, mt * nt), blockIdx.z);
}

// deterministic split-K combine: out[e] (+)= sum_z slab[z][e], fixed order
__global__ void splitk_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out, int64_t n, int splits,
                                     int accumulate);

template <class LA, class LB, int BM, int BN, int BK, int WM, int WN, int PD = 1>
static inline void launch_gemm(const typename LA::Desc& da, const typename LB::Desc& db, const OutDesc& dc, int M,
                               int N, int K, int splits, hipStream_t st) {
    int mt = cdiv(M, BM), nt = cdiv(N, BN);
    int kchunk = cdiv(cdiv(K, splits), BK) * BK;
    dim3 grid(mt * nt, 1, splits);
    constexpr size_t lds_bytes = sizeof(float) * 2 * BK * (BM + BN + 2 * LPAD);
    auto kern = gemm_kernel<LA, LB, BM, BN, BK, WM, WN, PD>;
    if constexpr (lds_bytes > 64 * 1024) {
        static bool once = (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                (int)lds_bytes) == hipSuccess);
        (void)once;
    }
    hipLaunchKernelGGL(kern, grid, dim3(NT), lds_bytes, st, da, db, dc, M, N, K, kchunk);
}

int tuning();   // SCAT_TUNE environment knob for kernel-variant experiments (0 = shipped default)
// 0: products on the fp32 MFMA (v_mfma_f32_32x32x2_f32).  1: fp32 operands split into three bf16 terms, six bf16
// MFMA products per fp32 product, fp32 accumulation (see conv3x3.hip).  SCAT_MATH=f32|bf16x3, scat_set_math_mode().
int math_mode();

}  // namespace scat

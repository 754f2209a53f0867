// Convolution weight gradient (1x1 and 3x3/pad 1, stride 1) with split-operand products (split.h):
//   dW[co][(ci,tap)] = sum over pixels  dy[n][co][r] * f(x[n][ci][r + shift(tap)]),   f = relu(x*scale+shift)
// an "NT" contraction whose two operands are both contiguous along the contraction index (pixels):
//  * a staging thread owns one (row, pixel octet): 8 consecutive pixels of a dy row / of an x row shifted by the
//    tap = 32 contiguous bytes = two 16-B loads; it applies the fused transform and the per-pixel padding mask,
//    splits the 8 values into the three bf16 fragments and writes three ds_write_b128 — the LDS image
//    [plane][octet][row][8 bf16] is exactly the MFMA operand layout (lane (row, h) <- octet h).
//  * images are walked in octets (ceil(HW/8) per image, the ragged last one masked), so an octet never straddles
//    two images and 7x7 / 14x14 planes need no special path.
//  * 16 pixels per stage, LDS double-buffered, global loads for the next stage in flight under the MFMAs.
//  * deterministic split-K over pixel stages into fp32 slabs, reduced in a fixed order (splitk_reduce_kernel).
// Replaces the autograd weight gradient of nn.Conv2d at models/resnet.py:65-72 (conv1, conv2, conv3, downsample).
#include "conv_common.h"
#include "split.h"

namespace scat {

struct WgDesc {
    const float* dy;      // [B][Cout][HW]
    const float* x;       // [B][Cin][HW]
    const float* scale;   // optional fused input transform on x, per Cin
    const float* shift;
    int relu;
    int Cout, Cin, H, W, HW;   // x plane
    int OW, OHW;          // dy plane (== x plane for stride 1)
    int stride, pad;
    int NO;               // octets per image = ceil(OHW / 8)
    int U;                // octets in total = B * NO
    int N;                // columns = Cin * KK
    int spz;              // stages (2 octets each) per split-K slice
    FastDiv dNO, dW, dKK;
    int64_t ndy, nx;
    const float* dy2;     // DSA kernels: dy = ca[co]*dy + cb[co]*dy2 + cc[co] (BatchNorm backward folded into the load)
    const float* coef;    // [3][Cout]
};

// 8 consecutive floats at byte offset off (OOB: zeros).  Two 16-B loads; a vector that would start before or end
// after the tensor (first / last rows of the whole tensor only) is fetched as 8 bounds-checked dwords instead,
// so nothing depends on how the hardware range-checks a partly out-of-range vector.
__device__ __forceinline__ void load8(__amdgpu_buffer_rsrc_t rs, int off, uint32_t mask, int64_t nfloats,
                                      float (&v)[8]) {
    if (off != OOB && (off < 0 || (int64_t)off + 32 > nfloats * 4)) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int o = off + 4 * e;
            const bool ok = ((mask >> e) & 1u) && o >= 0 && (int64_t)o + 4 <= nfloats * 4;
            v[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, ok ? o : OOB, 0, 0));
        }
        return;
    }
    const u32x4 t0 = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
    const u32x4 t1 = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 16, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) { v[q] = __uint_as_float(t0[q]); v[4 + q] = __uint_as_float(t1[q]); }
}

// S2: stride-2 convolution — the 8 pixels of an octet are 8 strided source pixels, fetched as 8 dwords
template <int KK, int MI, int NI, bool TF, bool S2, bool DSA = false>
__global__ __launch_bounds__(NT, 3) void wgrad_split_kernel(WgDesc d, OutDesc dc) {
    constexpr int BM = 64 * MI, BN = 64 * NI;
    constexpr int OSA = BM + 8, OSB = BN + 8;          // u32x4 per (plane, octet) slab; +128 B keeps the two octets of
                                                       // a row pair on different banks for the staging writes
    extern __shared__ __align__(16) float lds[];       // [2 buffers][A: 6 slabs of OSA | B: 6 slabs of OSB] x 16 B
    auto As = [&](int buf) -> u32x4* { return (u32x4*)lds + buf * 6 * (OSA + OSB); };
    auto Bs = [&](int buf) -> u32x4* { return (u32x4*)lds + buf * 6 * (OSA + OSB) + 6 * OSA; };

    const int mt = (d.Cout + BM - 1) / BM, nt = (d.N + BN - 1) / BN;
    const int tile = xcd_remap(blockIdx.x, mt * nt);
    const int i0 = (tile % mt) * BM, j0 = (tile / mt) * BN;
    const int z = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const int sbeg = z * d.spz, send = min(sbeg + d.spz, (d.U + 1) >> 1);

    // ---- staging items: (row, octet) = (tid >> 1, tid & 1) of each operand
    const __amdgpu_buffer_rsrc_t rsa = make_rsrc(d.dy, d.ndy), rsb = make_rsrc(d.x, d.nx);
    const int srow = tid >> 1, so = tid & 1;
    const bool a_item = srow < BM, b_item = srow < BN;
    const int arow = i0 + srow;
    const bool a_ok = a_item && arow < d.Cout;
    const int col = j0 + srow;
    const bool b_ok = b_item && col < d.N;
    int ci = 0, kh = 1, kw = 1;
    if (KK == 9) {
        const uint32_t c = d.dKK.div((uint32_t)(b_ok ? col : 0));
        const int tap = (b_ok ? col : 0) - (int)c * 9;
        ci = (int)c; kh = tap / 3; kw = tap - 3 * kh;
    } else {
        ci = b_ok ? col : 0;
    }
    if (S2 && KK == 1) { kh = 0; kw = 0; }             // (the stride-1 path folds pad 1 into kh - 1, kw - 1)
    const int bshift = (kh - 1) * d.W + (kw - 1);
    float bsc = 1.f, bsh = 0.f;
    if (TF && b_ok) { bsc = d.scale[ci]; bsh = d.shift[ci]; }
    const __amdgpu_buffer_rsrc_t rsa2 = make_rsrc(DSA ? d.dy2 : d.dy, DSA ? d.ndy : 0);
    float aca = 1.f, acb = 0.f, acc_ = 0.f;
    if (DSA && a_ok) { aca = d.coef[arow]; acb = d.coef[d.Cout + arow]; acc_ = d.coef[2 * d.Cout + arow]; }

    // two register sets: the loads of stage s+2 are issued at the top of stage s and written to LDS at the end of
    // stage s+1, i.e. ~2 x (MI*NI*6) MFMAs of cover for an HBM miss
    float araw[2][8], braw[2][8];
    float araw2[DSA ? 8 : 1];                          // DSA runs one stage ahead on a single register set
    uint32_t amask[2] = {0, 0}, bmask[2] = {0, 0};     // validity of the 8 pixels
    auto load_stage = [&](int s, auto set_tag) {
        constexpr int Q = decltype(set_tag)::value;
        const int u = 2 * s + so;
        const bool in = s < send && u < d.U;
        const uint32_t uu = in ? (uint32_t)u : 0u;
        const uint32_t n = d.dNO.div(uu);
        const int r0 = 8 * (int)(uu - n * (uint32_t)d.NO);
        const int cnt = in ? min(8, d.OHW - r0) : 0;
        const uint32_t live = (1u << cnt) - 1u;
        // A: dy[n][arow][r0 .. r0+7]
        {
            const int off = (a_ok && cnt > 0) ? (((int)n * d.Cout + arow) * d.OHW + r0) * 4 : OOB;
            load8(rsa, off, a_ok ? live : 0u, d.ndy, araw[Q]);
            if constexpr (DSA) {
                float tmp[8];
                load8(rsa2, off, a_ok ? live : 0u, d.ndy, tmp);
#pragma unroll
                for (int e = 0; e < 8; ++e) araw2[e] = tmp[e];
            }
            amask[Q] = a_ok ? live : 0u;
        }
        // B: x[n][ci][r0 + shift .. +7], per-pixel padding mask
        if constexpr (!S2) {
            uint32_t m = live;
            if (KK == 9) {
                const int y0 = (int)d.dW.div((uint32_t)r0), x0 = r0 - y0 * d.W;
                uint32_t v = 0;
                int y = y0 + kh - 1, x = x0 + kw - 1;      // source row / column of pixel e
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    // x runs past the row end after the wrap: the source column restarts at kw-1
                    const bool ok = (unsigned)y < (unsigned)d.H && (unsigned)x < (unsigned)d.W;
                    v |= (ok ? 1u : 0u) << e;
                    ++x;
                    if (x == d.W + kw - 1) { x = kw - 1; ++y; }
                }
                m &= v;
            }
            m = b_ok ? m : 0u;
            const int off = m ? (((int)n * d.Cin + ci) * d.HW + r0 + bshift) * 4 : OOB;
            load8(rsb, off, m, d.nx, braw[Q]);
            bmask[Q] = m;
        } else {
            // pixel e of the octet is output pixel (oy, ox): source (2*oy + kh - pad, 2*ox + kw - pad)
            int oy = (int)d.dW.div((uint32_t)r0), ox = r0 - oy * d.OW;     // dW divides by OW here
            const int base = ((int)n * d.Cin + ci) * d.HW;
            uint32_t m = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int sy = d.stride * oy + kh - d.pad, sx = d.stride * ox + kw - d.pad;
                const bool ok = b_ok && ((live >> e) & 1u) && (unsigned)sy < (unsigned)d.H && (unsigned)sx < (unsigned)d.W;
                braw[Q][e] = __uint_as_float(
                    __builtin_amdgcn_raw_buffer_load_b32(rsb, ok ? (base + sy * d.W + sx) * 4 : OOB, 0, 0));
                m |= (ok ? 1u : 0u) << e;
                if (++ox == d.OW) { ox = 0; ++oy; }
            }
            bmask[Q] = m;
        }
    };
    const float relu_lo = d.relu ? 0.f : -__builtin_inff();      // ReLU as a lower bound: one v_max, no select
    // a whole stage (every staged row has its 8 live pixels — almost all of them) takes the body without the
    // per-element selects: wave-uniform choice, see wgrad_pc_kernel
    auto store_stage = [&](int buf, auto set_tag) {
        constexpr int Q = decltype(set_tag)::value;
        auto body_a = [&](auto whole_tag) {
            constexpr bool WHOLE = decltype(whole_tag)::value;
            if (a_item) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float t = araw[Q][e];
                    if constexpr (DSA) t = fmaf(aca, t, fmaf(acb, araw2[e], acc_));
                    v[e] = WHOLE || ((amask[Q] >> e) & 1u) ? t : 0.f;
                }
                u32x4 hi, mid, lo;
                split3x8(v, hi, mid, lo);
                u32x4* p = As(buf) + so * OSA + srow;
                p[0] = hi; p[2 * OSA] = mid; p[4 * OSA] = lo;
            }
        };
        auto body_b = [&](auto whole_tag) {
            constexpr bool WHOLE = decltype(whole_tag)::value;
            if (b_item) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float t = braw[Q][e];
                    if constexpr (TF) t = fmaxf(fmaf(t, bsc, bsh), relu_lo);
                    v[e] = WHOLE || ((bmask[Q] >> e) & 1u) ? t : 0.f;
                }
                u32x4 hi, mid, lo;
                split3x8(v, hi, mid, lo);
                u32x4* p = Bs(buf) + so * OSB + srow;
                p[0] = hi; p[2 * OSB] = mid; p[4 * OSB] = lo;
            }
        };
        if (__builtin_amdgcn_ballot_w64(a_item && amask[Q] != 0xffu) == 0) body_a(std::true_type{});
        else body_a(std::false_type{});
        if (__builtin_amdgcn_ballot_w64(b_item && bmask[Q] != 0xffu) == 0) body_b(std::true_type{});
        else body_b(std::false_type{});
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int afrag = lh * OSA + wm * (BM / 2) + l31, bfrag = lh * OSB + wn * (BN / 2) + l31;

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    // stage sbeg + t lives in LDS buffer t & 1 and came through register set t & 1
    auto stage = [&](int s, auto cur_tag) {
        constexpr int CUR = decltype(cur_tag)::value;
        if constexpr (DSA) load_stage(s + 1, S0{});
        else load_stage(s + 2, std::integral_constant<int, CUR>{});      // past the end: every lane reads 0
        __builtin_amdgcn_sched_barrier(0);
        u32x4 af[MI][3], bf[NI][3];
        const u32x4* pa = As(CUR) + afrag;
        const u32x4* pb = Bs(CUR) + bfrag;
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
            for (int p = 0; p < 3; ++p) af[a][p] = pa[p * 2 * OSA + a * 32];
#pragma unroll
        for (int b = 0; b < NI; ++b)
#pragma unroll
            for (int p = 0; p < 3; ++p) bf[b][p] = pb[p * 2 * OSB + b * 32];
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
            for (int b = 0; b < NI; ++b) acc[a][b] = mfma_split(af[a], bf[b], acc[a][b]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (DSA) store_stage(CUR ^ 1, S0{});
        else store_stage(CUR ^ 1, std::integral_constant<int, CUR ^ 1>{});   // stage s+1, loaded one stage ago
        __syncthreads();
    };
    if (sbeg < send) {
        load_stage(sbeg, S0{});
        if constexpr (!DSA) load_stage(sbeg + 1, S1{});
        store_stage(0, S0{});
    }
    __syncthreads();
    for (int s = sbeg; s < send; s += 2) {
        stage(s, S0{});
        if (s + 1 < send) stage(s + 1, S1{});
    }
    store_tile<MI, NI, BM, BN, 2, 2>(acc, dc, d.Cout, d.N, i0, j0, z);
}

// ---------------------------------------------------------------- producer / consumer wave specialisation
//
// The weight gradient splits BOTH operands on their way into LDS: 16 elements of staging (load, mask, fused transform,
// three-way split, three 16-byte LDS writes: ~130 VALU) per thread for every 24 MFMAs, and its split-K grids are sized
// for ~1.5 workgroups per CU (wgrad_split_plan) — with one or two in-order wavefronts per SIMD the matrix pipe idles
// while its wavefront splits (measured: half of what the same MFMA stream reaches alone, tools/mfma_probe.hip).  Here
// a 512-thread workgroup has four consumer wavefronts (fragments from LDS, MFMAs: nothing else) and four producer
// wavefronts (everything else, two register sets, LDS three stages deep); one barrier per 16-pixel stage.
template <int KK, bool TF, bool S2, bool DSA>
__global__ __launch_bounds__(512, 4) void wgrad_pc_kernel(WgDesc d, OutDesc dc) {
    constexpr int MI = 2, NI = 2, BM = 128, BN = 128;
    constexpr int OSA = BM + 8, OSB = BN + 8;
    constexpr int BUF = 6 * (OSA + OSB);               // u32x4 per stage buffer
    extern __shared__ __align__(16) float lds[];       // [3 buffers][A: 6 slabs of OSA | B: 6 slabs of OSB] x 16 B
    u32x4* const L0 = (u32x4*)lds;

    const int mt = (d.Cout + BM - 1) / BM, nt = (d.N + BN - 1) / BN;
    const int tile = xcd_remap(blockIdx.x, mt * nt);
    const int i0 = (tile % mt) * BM, j0 = (tile / mt) * BN;
    const int z = blockIdx.z;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int sbeg = z * d.spz, send = min(sbeg + d.spz, (d.U + 1) >> 1);
    const int nst = send > sbeg ? send - sbeg : 0;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;

    if (wave >= 4) {
        // ------------------------------------------------------------ producer: (row, octet) = (ptid >> 1, ptid & 1)
        const int tid = threadIdx.x - 256;
        const __amdgpu_buffer_rsrc_t rsa = make_rsrc(d.dy, d.ndy), rsb = make_rsrc(d.x, d.nx);
        const int srow = tid >> 1, so = tid & 1;
        const int arow = i0 + srow;
        const bool a_ok = arow < d.Cout;
        const int col = j0 + srow;
        const bool b_ok = col < d.N;
        int ci = 0, kh = 1, kw = 1;
        if (KK == 9) {
            const uint32_t c = d.dKK.div((uint32_t)(b_ok ? col : 0));
            const int tap = (b_ok ? col : 0) - (int)c * 9;
            ci = (int)c; kh = tap / 3; kw = tap - 3 * kh;
        } else {
            ci = b_ok ? col : 0;
        }
        if (S2 && KK == 1) { kh = 0; kw = 0; }
        const int bshift = (kh - 1) * d.W + (kw - 1);
        float bsc = 1.f, bsh = 0.f;
        if (TF && b_ok) { bsc = d.scale[ci]; bsh = d.shift[ci]; }
        const __amdgpu_buffer_rsrc_t rsa2 = make_rsrc(DSA ? d.dy2 : d.dy, DSA ? d.ndy : 0);
        float aca = 1.f, acb = 0.f, acc_ = 0.f;
        if (DSA && a_ok) { aca = d.coef[arow]; acb = d.coef[d.Cout + arow]; acc_ = d.coef[2 * d.Cout + arow]; }
        float araw[2][8], braw[2][8];
        float araw2[DSA ? 2 : 1][8];
        uint32_t amask[2] = {0, 0}, bmask[2] = {0, 0};
        auto load_stage = [&](int s, auto set_tag) {
            constexpr int Q = decltype(set_tag)::value;
            const int u = 2 * s + so;
            const bool in = s < send && u < d.U;
            const uint32_t uu = in ? (uint32_t)u : 0u;
            const uint32_t n = d.dNO.div(uu);
            const int r0 = 8 * (int)(uu - n * (uint32_t)d.NO);
            const int cnt = in ? min(8, d.OHW - r0) : 0;
            const uint32_t live = (1u << cnt) - 1u;
            {
                const int off = (a_ok && cnt > 0) ? (((int)n * d.Cout + arow) * d.OHW + r0) * 4 : OOB;
                load8(rsa, off, a_ok ? live : 0u, d.ndy, araw[Q]);
                if constexpr (DSA) load8(rsa2, off, a_ok ? live : 0u, d.ndy, araw2[Q]);
                amask[Q] = a_ok ? live : 0u;
            }
            if constexpr (!S2) {
                uint32_t m = live;
                if (KK == 9) {
                    const int y0 = (int)d.dW.div((uint32_t)r0), x0 = r0 - y0 * d.W;
                    uint32_t v = 0;
                    int y = y0 + kh - 1, x = x0 + kw - 1;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const bool ok = (unsigned)y < (unsigned)d.H && (unsigned)x < (unsigned)d.W;
                        v |= (ok ? 1u : 0u) << e;
                        ++x;
                        if (x == d.W + kw - 1) { x = kw - 1; ++y; }
                    }
                    m &= v;
                }
                m = b_ok ? m : 0u;
                const int off = m ? (((int)n * d.Cin + ci) * d.HW + r0 + bshift) * 4 : OOB;
                load8(rsb, off, m, d.nx, braw[Q]);
                bmask[Q] = m;
            } else {
                int oy = (int)d.dW.div((uint32_t)r0), ox = r0 - oy * d.OW;
                const int base = ((int)n * d.Cin + ci) * d.HW;
                uint32_t m = 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int sy = d.stride * oy + kh - d.pad, sx = d.stride * ox + kw - d.pad;
                    const bool ok = b_ok && ((live >> e) & 1u) && (unsigned)sy < (unsigned)d.H && (unsigned)sx < (unsigned)d.W;
                    braw[Q][e] = __uint_as_float(
                        __builtin_amdgcn_raw_buffer_load_b32(rsb, ok ? (base + sy * d.W + sx) * 4 : OOB, 0, 0));
                    m |= (ok ? 1u : 0u) << e;
                    if (++ox == d.OW) { ox = 0; ++oy; }
                }
                bmask[Q] = m;
            }
        };
        const float relu_lo = d.relu ? 0.f : -__builtin_inff();      // ReLU as a lower bound: one v_max, no select
        // Almost every stage is whole (8 live pixels in every row of the tile): a wave-uniform test picks a body without
        // the per-element selects — beside a busy matrix pipe the SIMD issues about one vector instruction per MFMA
        // (tools/mfma_probe.hip, PROBE_PC=1), so the three instructions per element of the masking are worth a branch.
        auto store_stage = [&](u32x4* buf, auto set_tag) {
            constexpr int Q = decltype(set_tag)::value;
            auto body_a = [&](auto whole_tag) {
                constexpr bool WHOLE = decltype(whole_tag)::value;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float t = araw[Q][e];
                    if constexpr (DSA) t = fmaf(aca, t, fmaf(acb, araw2[Q][e], acc_));
                    v[e] = WHOLE || ((amask[Q] >> e) & 1u) ? t : 0.f;
                }
                u32x4 hi, mid, lo;
                split3x8(v, hi, mid, lo);
                u32x4* p = buf + so * OSA + srow;
                p[0] = hi; p[2 * OSA] = mid; p[4 * OSA] = lo;
            };
            auto body_b = [&](auto whole_tag) {
                constexpr bool WHOLE = decltype(whole_tag)::value;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float t = braw[Q][e];
                    if constexpr (TF) t = fmaxf(fmaf(t, bsc, bsh), relu_lo);
                    v[e] = WHOLE || ((bmask[Q] >> e) & 1u) ? t : 0.f;
                }
                u32x4 hi, mid, lo;
                split3x8(v, hi, mid, lo);
                u32x4* p = buf + 6 * OSA + so * OSB + srow;
                p[0] = hi; p[2 * OSB] = mid; p[4 * OSB] = lo;
            };
            if (__builtin_amdgcn_ballot_w64(amask[Q] != 0xffu) == 0) body_a(std::true_type{});
            else body_a(std::false_type{});
            if (__builtin_amdgcn_ballot_w64(bmask[Q] != 0xffu) == 0) body_b(std::true_type{});
            else body_b(std::false_type{});
        };
        // stage sbeg + k: register set k & 1, LDS buffer k % 3
        load_stage(sbeg, S0{});
        load_stage(sbeg + 1, S1{});
        store_stage(L0, S0{});
        load_stage(sbeg + 2, S0{});
        store_stage(L0 + BUF, S1{});
        __syncthreads();
        int wb = 2;
        auto iter = [&](int k, auto par_tag) {
            constexpr int P = decltype(par_tag)::value;
            load_stage(sbeg + k + 3, std::integral_constant<int, P ^ 1>{});
            store_stage(L0 + wb * BUF, par_tag);
            wb = wb == 2 ? 0 : wb + 1;
            __syncthreads();
        };
        for (int k = 0; k < nst; k += 2) {
            iter(k, S0{});
            if (k + 1 < nst) iter(k + 1, S1{});
        }
        return;
    }

    // ---------------------------------------------------------------- consumer (2 x 2 wavefronts, 64 x 64 each)
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    f32x16 acc[MI][NI];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int afrag = lh * OSA + wm * (BM / 2) + l31, bfrag = 6 * OSA + lh * OSB + wn * (BN / 2) + l31;
    __syncthreads();
    // fragments (a = 0) and (b = 0) of a stage are read during the previous stage's last MFMA group: the stage was
    // published a barrier earlier (three buffers), so the first group of every stage starts right after the barrier
    u32x4 a0[3], b0[3], a1[3], b1[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) { a0[p] = L0[afrag + p * 2 * OSA]; b0[p] = L0[bfrag + p * 2 * OSB]; }
    int rb = 0;
    for (int k = 0; k < nst; ++k) {
        const u32x4* cur = L0 + rb * BUF;
        rb = rb == 2 ? 0 : rb + 1;
        const u32x4* nxt = L0 + rb * BUF;
#pragma unroll
        for (int p = 0; p < 3; ++p) b1[p] = cur[bfrag + p * 2 * OSB + 32];
#pragma unroll
        for (int p = 0; p < 3; ++p) a1[p] = cur[afrag + p * 2 * OSA + 32];
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = mfma_split(a0, b0, acc[0][0]);
        acc[0][1] = mfma_split(a0, b1, acc[0][1]);
        acc[1][0] = mfma_split(a1, b0, acc[1][0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < 3; ++p) { a0[p] = nxt[afrag + p * 2 * OSA]; b0[p] = nxt[bfrag + p * 2 * OSB]; }
        __builtin_amdgcn_sched_barrier(0);
        acc[1][1] = mfma_split(a1, b1, acc[1][1]);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
    store_tile<MI, NI, BM, BN, 2, 2>(acc, dc, d.Cout, d.N, i0, j0, z);
}

template <int KK, bool TF, bool S2, bool DSA>
static void launch_wg_pc(const WgDesc& d, const OutDesc& dc, int splits, hipStream_t st) {
    const int mt = cdiv(d.Cout, 128), nt = cdiv(d.N, 128);
    constexpr size_t lds_bytes = (size_t)3 * 6 * (128 + 8 + 128 + 8) * 16;
    auto kern = wgrad_pc_kernel<KK, TF, S2, DSA>;
    static bool once = (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)lds_bytes) == hipSuccess);
    (void)once;
    hipLaunchKernelGGL(kern, dim3(mt * nt, 1, splits), dim3(512), lds_bytes, st, d, dc);
}

// SCAT_WG_PC=0: every wavefront stages and multiplies; 1: producer / consumer wavefronts for the 128 x 128 tiles
static int wg_pc_mode() {
    static const int m = diag_env_int("SCAT_WG_PC", 1);
    return m;
}

template <int KK, int MI, int NI, bool TF, bool S2, bool DSA = false>
static void launch_wg(const WgDesc& d, const OutDesc& dc, int splits, hipStream_t st) {
    constexpr int BM = 64 * MI, BN = 64 * NI;
    const int mt = cdiv(d.Cout, BM), nt = cdiv(d.N, BN);
    constexpr size_t lds_bytes = (size_t)2 * 6 * (BM + 8 + BN + 8) * 16;
    hipLaunchKernelGGL((wgrad_split_kernel<KK, MI, NI, TF, S2, DSA>), dim3(mt * nt, 1, splits), dim3(NT), lds_bytes, st,
                       d, dc);
}

template <int KK, bool TF, bool S2>
static void launch_wg_tile(int mi, int ni, const WgDesc& d, const OutDesc& dc, int splits, hipStream_t st) {
    if (mi == 2 && ni == 2) launch_wg<KK, 2, 2, TF, S2>(d, dc, splits, st);
    else if (mi == 2) launch_wg<KK, 2, 1, TF, S2>(d, dc, splits, st);
    else if (ni == 2) launch_wg<KK, 1, 2, TF, S2>(d, dc, splits, st);
    else launch_wg<KK, 1, 1, TF, S2>(d, dc, splits, st);
}

WgSplitPlan wgrad_split_plan(int B, int Cin, int Cout, int KK, int HW) {
    WgSplitPlan p;
    p.M = Cout;
    p.N = Cin * KK;
    p.mi = Cout > 64 ? 2 : 1;
    p.ni = p.N > 64 ? 2 : 1;
    const int NO = (HW + 7) / 8;
    p.stages = (B * NO + 1) / 2;
    const int tiles = cdiv(p.M, 64 * p.mi) * cdiv(p.N, 64 * p.ni);
    // measured at batch 96 (SCAT_WG_TARGET / _TARGET1 / _TARGET9 sweeps of the whole train step, where these
    // kernels share the GPU with the data-gradient chain): ~1.5 workgroups per CU for 1x1 (384), ~3 for 3x3 (768).  (Alone on the
    // GPU the 3x3 kernel prefers ~6 — its staging is VALU-bound and more co-resident waves help — but in the step
    // the extra slab traffic and reduce work cost more: 1536 -> 768 is +1.0 % on the step.)
    static const int forced = diag_env_int("SCAT_WG_TARGET", 0);
    static const int forced1 = diag_env_int("SCAT_WG_TARGET1", 0);
    static const int forced9 = diag_env_int("SCAT_WG_TARGET9", 0);
    const int f = KK == 1 ? (forced1 > 0 ? forced1 : forced) : (forced9 > 0 ? forced9 : forced);
    const int target = f > 0 ? f : (KK == 1 ? 384 : 768);
    int s = cdiv(target, tiles);
    static const int minst = diag_env_int("SCAT_WG_MINSTAGES", 24);
    const int smax = p.stages / minst > 0 ? p.stages / minst : 1;   // >= 24 stages (384 pixels) per slice
    if (s > smax) s = smax;
    if (s > 2048) s = 2048;
    if (s < 1) s = 1;
    p.spz = cdiv(p.stages, s);
    p.splits = cdiv(p.stages, p.spz);
    return p;
}

template <bool TF>
static void launch_wg_dsa(int mi, int ni, const WgDesc& d, const OutDesc& dc, int splits, hipStream_t st) {
    if (mi == 2 && ni == 2) launch_wg<1, 2, 2, TF, false, true>(d, dc, splits, st);
    else if (mi == 2) launch_wg<1, 2, 1, TF, false, true>(d, dc, splits, st);
    else if (ni == 2) launch_wg<1, 1, 2, TF, false, true>(d, dc, splits, st);
    else launch_wg<1, 1, 1, TF, false, true>(d, dc, splits, st);
}

void wgrad_split_launch(const WgSplitPlan& p, const float* dy, const float* x, float* out, int B, int Cin, int H, int W,
                        int Cout, int KK, int stride, const float* in_scale, const float* in_shift, int in_relu,
                        hipStream_t st, const float* dy2, const float* coef3) {
    WgDesc d{};
    d.dy2 = dy2; d.coef = coef3;
    const int pad = KK == 9 ? 1 : 0, k = KK == 9 ? 3 : 1;
    const int OH = (H + 2 * pad - k) / stride + 1, OW = (W + 2 * pad - k) / stride + 1;
    d.dy = dy; d.x = x; d.scale = in_scale; d.shift = in_shift; d.relu = in_scale ? in_relu : 0;
    d.Cout = Cout; d.Cin = Cin; d.H = H; d.W = W; d.HW = H * W; d.OW = OW; d.OHW = OH * OW; d.stride = stride; d.pad = pad;
    d.NO = (d.OHW + 7) / 8; d.U = B * d.NO; d.N = p.N; d.spz = p.spz;
    d.dNO = FastDiv::make(d.NO); d.dW = FastDiv::make(OW); d.dKK = FastDiv::make(KK);
    d.ndy = (int64_t)B * Cout * d.OHW; d.nx = (int64_t)B * Cin * d.HW;
    OutDesc dc{};
    dc.p = out; dc.mode = 0; dc.si = p.N; dc.sj = 1; dc.sz = (int64_t)p.M * p.N; dc.I = p.M; dc.J = p.N;
    dc.n = (int64_t)p.M * p.N;
    set_kernel_label("wgrad%s%s_split_%dx%dx16%s%s_split%d", KK == 9 ? "3x3" : "1x1", stride == 2 ? "_s2" : "", 64 * p.mi,
                     64 * p.ni, in_scale ? "_tf" : "", dy2 ? "_bnb" : "", p.splits);
    if (wg_pc_mode() && p.mi == 2 && p.ni == 2) {
        const bool tf = in_scale != nullptr;
        set_kernel_label("wgrad%s%s_split_pc128x128x16%s%s_split%d", KK == 9 ? "3x3" : "1x1", stride == 2 ? "_s2" : "",
                         in_scale ? "_tf" : "", dy2 ? "_bnb" : "", p.splits);
        if (dy2) {
            if (tf) launch_wg_pc<1, true, false, true>(d, dc, p.splits, st);
            else launch_wg_pc<1, false, false, true>(d, dc, p.splits, st);
        } else if (stride == 2) {
            if (KK == 9) { if (tf) launch_wg_pc<9, true, true, false>(d, dc, p.splits, st); else launch_wg_pc<9, false, true, false>(d, dc, p.splits, st); }
            else { if (tf) launch_wg_pc<1, true, true, false>(d, dc, p.splits, st); else launch_wg_pc<1, false, true, false>(d, dc, p.splits, st); }
        } else if (KK == 9) {
            if (tf) launch_wg_pc<9, true, false, false>(d, dc, p.splits, st); else launch_wg_pc<9, false, false, false>(d, dc, p.splits, st);
        } else {
            if (tf) launch_wg_pc<1, true, false, false>(d, dc, p.splits, st); else launch_wg_pc<1, false, false, false>(d, dc, p.splits, st);
        }
        return;
    }
    if (dy2) {      // 1x1 / stride 1 only (checked by the caller)
        if (in_scale) launch_wg_dsa<true>(p.mi, p.ni, d, dc, p.splits, st);
        else launch_wg_dsa<false>(p.mi, p.ni, d, dc, p.splits, st);
    } else if (stride == 2) {
        if (KK == 9) {
            if (in_scale) launch_wg_tile<9, true, true>(p.mi, p.ni, d, dc, p.splits, st);
            else launch_wg_tile<9, false, true>(p.mi, p.ni, d, dc, p.splits, st);
        } else {
            if (in_scale) launch_wg_tile<1, true, true>(p.mi, p.ni, d, dc, p.splits, st);
            else launch_wg_tile<1, false, true>(p.mi, p.ni, d, dc, p.splits, st);
        }
    } else if (KK == 9) {
        if (in_scale) launch_wg_tile<9, true, false>(p.mi, p.ni, d, dc, p.splits, st);
        else launch_wg_tile<9, false, false>(p.mi, p.ni, d, dc, p.splits, st);
    } else {
        if (in_scale) launch_wg_tile<1, true, false>(p.mi, p.ni, d, dc, p.splits, st);
        else launch_wg_tile<1, false, false>(p.mi, p.ni, d, dc, p.splits, st);
    }
}

}  // namespace scat

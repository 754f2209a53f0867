// Weight gradient of a 3x3 / stride-1 / pad-1 convolution with FEW channels (Cin, Cout multiples of 32 up to 64:
// HRNet's 32- and 64-channel branches, models/hrnet.py:38-63, and ResNet layer1's conv2, models/resnet.py:68) on
// split-operand products:
//     dW[co][ci][kh][kw] = sum over (n, y, xx)  dy[n][co][y][xx - (kw-1)] * x[n][ci][y + kh - 1][xx]
// a 32 x 288 output per (Cout block, Cin block) with a contraction over 300 k pixels at batch 96 and 56 x 56.  The
// general kernel (conv_wgrad_split.hip) gathers x once per tap and splits every element of both operands once per
// 128-column tile: 9x (x) and 3x (dy) the necessary staging work around 12 MFMAs per wavefront and stage, and beside
// a busy matrix pipe a SIMD issues about one vector instruction per MFMA (tools/mfma_probe.hip, PROBE_PC=1) — it runs
// at a tenth of the pipe.  Here a workgroup walks whole image rows:
//  * each x row (32 channels) is split ONCE into three bf16 planes in LDS and stays there for the three output rows
//    that read it (ring of four rows); the row above / below an image is simply not multiplied;
//  * each dy row is split once and written three times, shifted by -1 / 0 / +1 pixels (the kw taps), so every
//    fragment of every tap is one aligned ds_read_b128: the shift costs four v_alignbit per plane in the producer,
//    nothing in the consumers;
//  * consumer wavefront q (0..3) owns the 16-pixel step q of every row and all nine taps: 54 MFMAs on 18 fragment
//    reads per row, its 32 x 288 partial tile summed with the other three through LDS at the end; four producer
//    wavefronts load, transform (the fused BatchNorm + ReLU of the layer input), split and write one row ahead.
// Deterministic split over rows into fp32 slabs [splits][Cout][Cin*9], fixed-order reduce (launch_splitk_reduce).
#include "conv_common.h"
#include "split.h"

namespace scat {

struct RowWgDesc {
    const float* dy;      // [B][Cout][H][W]
    const float* x;       // [B][Cin][H][W]
    float* slab;          // [splits][Cout][Cin * 9]
    const float* scale;   // fused input transform relu(x * scale[ci] + shift[ci]) or nullptr
    const float* shift;
    int relu;
    int B, Cin, Cout, H, W;
    int rows;             // B * H output rows in total
    int rpw;              // output rows per workgroup
    int ncb;              // Cin / 32
    int noct;             // pixel octets per row held in LDS (2 per 16-pixel step)
    int nov;              // octets that hold pixels: ceil(W / 8)
    FastDiv dH;
    int64_t ndy, nx;
    int ntile, nsplit, ngroup;   // XCD-local split-K mapping
    int stamp;            // diag build only (SCAT_WG_ROWS_STAMP, tools/rows_stamp.py): results are overwritten by time stamps
};

constexpr int RW_NT = 512;      // wavefronts 0-3: consumers (16-pixel step q = wavefront, all nine taps); 4-7: producers

__global__ __launch_bounds__(RW_NT) void wgrad3x3_rows_kernel(RowWgDesc d) {
    extern __shared__ __align__(16) float lds[];
    u32x4* const L = (u32x4*)lds;
    const int XS = 3 * d.noct * 32;                 // u32x4 per x row slot:  [plane][octet][ci]
    const int DS = 9 * d.noct * 32;                 // u32x4 per dy buffer:   [shift][plane][octet][co]
    u32x4* const XR = L;                            // ring of 4 x rows
    u32x4* const DY = L + 4 * XS;                   // 2 dy buffers
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    // the tiles of one row-range slice read the same rows of dy and x: one XCD (splitk_xcd_map, conv_common.h)
    int tile_, z;
    if (!splitk_xcd_map(blockIdx.x, d.ntile, d.nsplit, d.ngroup, tile_, z)) return;
    const int cb = tile_ % d.ncb, mb = tile_ / d.ncb;
    const int g0 = z * d.rpw, g1 = min(g0 + d.rpw, d.rows);
    const int N = d.Cin * 9;

    unsigned long long t0 = 0, t1 = 0, t2 = 0;
    if (kDiag && d.stamp) t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = tid; i < 4 * XS + 2 * DS; i += RW_NT) L[i] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    if (g0 >= g1) return;

    if (wave >= 4) {
        // ------------------------------------------------------------------ producers: item (row r, octet o)
        const int pt = tid - 256, r = pt & 31, o = pt >> 5;
        const bool active = o < d.nov;
        const __amdgpu_buffer_rsrc_t rsx = make_rsrc(d.x, d.nx), rsy = make_rsrc(d.dy, d.ndy);
        const int ci = cb * 32 + r, co = mb * 32 + r;
        float sc = 1.f, sh = 0.f;
        if (d.scale) { sc = d.scale[ci]; sh = d.shift[ci]; }
        const float relu_lo = d.relu ? 0.f : -__builtin_inff();
        const bool tf = d.scale != nullptr;
        const int nlive = min(8, d.W - 8 * o);                 // pixels of this octet inside the row (W % 4 == 0)
        auto load_x = [&](int gi, float (&v)[8]) {
            const bool ok = active && gi >= 0 && gi < d.rows;
            const uint32_t gg = ok ? (uint32_t)gi : 0u;
            const uint32_t n = d.dH.div(gg);
            const int y = (int)(gg - n * (uint32_t)d.H);
            const int off = ((((int)n * d.Cin + ci) * d.H + y) * d.W + 8 * o) * 4;
            const u32x4 t0 = __builtin_amdgcn_raw_buffer_load_b128(rsx, ok ? off : OOB, 0, 0);
            const u32x4 t1 = __builtin_amdgcn_raw_buffer_load_b128(rsx, ok && nlive > 4 ? off + 16 : OOB, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) { v[q] = __uint_as_float(t0[q]); v[4 + q] = __uint_as_float(t1[q]); }
        };
        auto store_x = [&](int gi, const float (&v)[8]) {
            if (!(active && gi >= 0 && gi < d.rows)) return;
            float t[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                t[e] = v[e];
                if (tf) {
                    t[e] = fmaxf(fmaf(t[e], sc, sh), relu_lo);
                    t[e] = e < nlive ? t[e] : 0.f;              // padding stays zero after the transform
                }
            }
            u32x4 hi, mid, lo;
            split3x8(t, hi, mid, lo);
            u32x4* p = XR + (gi & 3) * XS + o * 32 + r;
            p[0] = hi; p[d.noct * 32] = mid; p[2 * d.noct * 32] = lo;
        };
        // dy: pixels 8o-1 .. 8o+8 of the row (the two neighbours feed the shifted copies)
        auto load_dy = [&](int g, float (&v)[10]) {
            const bool ok = active && g < g1;
            const uint32_t gg = ok ? (uint32_t)g : 0u;
            const uint32_t n = d.dH.div(gg);
            const int y = (int)(gg - n * (uint32_t)d.H);
            const int off = ((((int)n * d.Cout + co) * d.H + y) * d.W + 8 * o) * 4;
            const u32x4 t0 = __builtin_amdgcn_raw_buffer_load_b128(rsy, ok ? off : OOB, 0, 0);
            const u32x4 t1 = __builtin_amdgcn_raw_buffer_load_b128(rsy, ok && nlive > 4 ? off + 16 : OOB, 0, 0);
            v[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsy, ok && o > 0 ? off - 4 : OOB, 0, 0));
            v[9] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsy, ok && nlive == 8 && 8 * o + 8 < d.W ? off + 32 : OOB, 0, 0));
#pragma unroll
            for (int q = 0; q < 4; ++q) { v[1 + q] = __uint_as_float(t0[q]); v[5 + q] = __uint_as_float(t1[q]); }
        };
        auto store_dy = [&](int g, const float (&v)[10]) {
            if (!(active && g < g1)) return;
            // pairs (e-1,e0) (e1,e2) (e3,e4) (e5,e6) (e7,e8): dwords 0..3 are the copy for kw = 2 (dy[xx-1]), dwords 1..4 the
            // copy for kw = 0 (dy[xx+1]); kw = 1 pairs (e0,e1).. = the same halves one bf16 further on
            uint32_t pa[3][5];
#pragma unroll
            for (int j = 0; j < 5; ++j) split3(v[2 * j], v[2 * j + 1], pa[0][j], pa[1][j], pa[2][j]);
            u32x4* base = DY + (g & 1) * DS + o * 32 + r;
            const int ps = d.noct * 32;                          // plane stride
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                base[(0 * 3 + p) * ps] = u32x4{pa[p][1], pa[p][2], pa[p][3], pa[p][4]};
                u32x4 mid;
#pragma unroll
                for (int j = 0; j < 4; ++j) mid[j] = __builtin_amdgcn_alignbit(pa[p][j + 1], pa[p][j], 16);
                base[(1 * 3 + p) * ps] = mid;
                base[(2 * 3 + p) * ps] = u32x4{pa[p][0], pa[p][1], pa[p][2], pa[p][3]};
            }
        };
        float xr[8], dr[10];
        {
            float xa[8], xb[8], xc[8], da[10];
            load_x(g0 - 1, xa); load_x(g0, xb); load_x(g0 + 1, xc); load_dy(g0, da);
            load_x(g0 + 2, xr); load_dy(g0 + 1, dr);
            store_x(g0 - 1, xa); store_x(g0, xb); store_x(g0 + 1, xc); store_dy(g0, da);
        }
        __syncthreads();
        for (int g = g0; g < g1; ++g) {
            store_x(g + 2, xr);
            store_dy(g + 1, dr);
            load_x(g + 3, xr);
            load_dy(g + 2, dr);
            __syncthreads();
        }
        __syncthreads();      // the consumers' two reduction rounds
        __syncthreads();
        __syncthreads();
        __syncthreads();
        return;
    }

    // ---------------------------------------------------------------------- consumers: 16-pixel step q = wavefront
    // (W > 48: four steps per row, one per wavefront; narrower rows leave wavefronts idle), all nine taps each:
    // 9 dy + 9 x fragment reads and 54 MFMAs per row; the four partial 32 x 288 tiles are summed through LDS at the end
    const int l31 = lane & 31, lh = lane >> 5;
    const int ks = d.noct >> 1;
    const int ps = d.noct * 32;
    f32x16 acc[3][3];                                            // [kh][kw]
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][s][i] = 0.f;
    int y;
    {
        const uint32_t n = d.dH.div((uint32_t)g0);
        y = g0 - (int)n * d.H;
    }
    const int frag = lh * 32 + l31;
    __syncthreads();
    if (kDiag && d.stamp) t1 = __builtin_amdgcn_s_memrealtime();
    for (int g = g0; g < g1; ++g) {
        for (int q = wave; q < ks; q += 4) {
            const u32x4* ds = DY + (g & 1) * DS + frag + q * 64;
            u32x4 a[3][3];
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int p = 0; p < 3; ++p) a[s][p] = ds[(s * 3 + p) * ps];
            static_for<3>([&](auto kh_tag) {
                constexpr int KH = decltype(kh_tag)::value;
                const int yi = y + KH - 1;
                if (yi >= 0 && yi < d.H) {
                    const u32x4* xs = XR + ((g + KH + 3) & 3) * XS + frag + q * 64;
                    u32x4 b[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) b[p] = xs[p * ps];
#pragma unroll
                    for (int s = 0; s < 3; ++s) acc[KH][s] = mfma_split(a[s], b, acc[KH][s]);
                }
            });
        }
        if (++y == d.H) y = 0;
        __syncthreads();
    }
    if (kDiag && d.stamp) t2 = __builtin_amdgcn_s_memrealtime();
    // sum of the four wavefronts' tiles: 2 + 3 -> LDS -> 0 + 1, then 1 -> LDS -> 0 (9 tiles x 16 registers x 64 lanes)
    float* const R = lds;
    auto put = [&](int slot) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int i = 0; i < 16; ++i) R[((slot * 9 + a * 3 + s) * 16 + i) * 64 + lane] = acc[a][s][i];
    };
    auto add = [&](int slot) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[a][s][i] += R[((slot * 9 + a * 3 + s) * 16 + i) * 64 + lane];
    };
    if (wave >= 2) put(wave - 2);
    __syncthreads();
    if (wave < 2) add(wave);
    __syncthreads();
    if (wave == 1) put(0);
    __syncthreads();
    if (wave == 0) {
        add(0);
        // rows = output channels, columns = input channels of this block; column index of dW: ci * 9 + kh * 3 + kw
        float* out = d.slab + ((int64_t)z * d.Cout + mb * 32) * N + (cb * 32 + l31) * 9;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = (i & 3) + 8 * (i >> 2) + 4 * lh;
                    out[(int64_t)row * N + a * 3 + s] = acc[a][s][i];
                }
    }
    __syncthreads();
    if (kDiag && d.stamp && tid == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        unsigned long long* o64 = (unsigned long long*)(d.slab + ((int64_t)z * d.Cout + mb * 32) * N + cb * 288);
        o64[0] = t0; o64[1] = t1; o64[2] = t2; o64[3] = __builtin_amdgcn_s_memrealtime();
    }
}

// ---------------------------------------------------------------- 64 x 64 blocks, rows of at most 32 pixels
//
// The same walk for the narrower feature maps (W <= 32: 28 x 28, 14 x 14, 7 x 7) of layers with 64 .. 512 channels: a
// workgroup owns a 64 (Cout) x 64 (Cin) block of dW — four 32 x 32 sub-blocks, one per consumer wavefront (mi, ci), nine
// taps each, no reduction between wavefronts — and RPI image rows per barrier interval (two for W <= 16, where a row is a
// single 16-pixel step).  x rows live in a ring of 4 RPI slots, dy rows (three shifted copies) in 2 RPI buffers; producer
// lane (r, o, j): channel r of the block, pixel octet o, row j of the group.  Every element of x is split Cout / 64
// times and every element of dy Cin / 64 times in total, instead of 9 Cout / 128 and Cin 9 / 128.
struct RowWg64Desc {
    const float* dy;
    const float* x;
    float* slab;
    const float* scale;
    const float* shift;
    int relu;
    int B, Cin, Cout, H, W;
    int rows, rpw, ncb, noct, nov, rpi;
    int ntile, nsplit, ngroup;   // XCD-local split-K mapping
    FastDiv dH;
    int64_t ndy, nx;
};

template <int RPI>
__global__ __launch_bounds__(512) void wgrad3x3_rows64_kernel(RowWg64Desc d) {
    constexpr int NX = 4 * RPI, ND = 2 * RPI, NG = RPI == 1 ? 3 : 2;     // ring sizes (rows); x row groups a consumer needs
    extern __shared__ __align__(16) float lds[];
    u32x4* const L = (u32x4*)lds;
    const int XS = 3 * d.noct * 64;                 // u32x4 per x row slot:  [plane][octet][ci 64]
    const int DS = 9 * d.noct * 64;                 // u32x4 per dy row:      [shift][plane][octet][co 64]
    u32x4* const XR = L;
    u32x4* const DY = L + NX * XS;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    // the tiles of one row-range slice read the same rows of dy and x: one XCD (splitk_xcd_map, conv_common.h)
    int tile_, z;
    if (!splitk_xcd_map(blockIdx.x, d.ntile, d.nsplit, d.ngroup, tile_, z)) return;
    const int cb = tile_ % d.ncb, mb = tile_ / d.ncb;
    const int g0 = z * d.rpw, g1 = min(g0 + d.rpw, d.rows);
    const int N = d.Cin * 9;
    const int nit = (g1 - g0 + RPI - 1) / RPI;

    for (int i = tid; i < NX * XS + ND * DS; i += 512) L[i] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    if (g0 >= g1) return;

    if (wave >= 4) {
        // ------------------------------------------------------------------ producers
        const int pt = tid - 256, r = pt & 63;
        const int o = (pt >> 6) % d.nov, j = (pt >> 6) / d.nov;           // octet, row of the group
        const bool active = j < RPI;
        const __amdgpu_buffer_rsrc_t rsx = make_rsrc(d.x, d.nx), rsy = make_rsrc(d.dy, d.ndy);
        const int ci = cb * 64 + r, co = mb * 64 + r;
        float sc = 1.f, sh = 0.f;
        if (d.scale) { sc = d.scale[ci]; sh = d.shift[ci]; }
        const float relu_lo = d.relu ? 0.f : -__builtin_inff();
        const bool tf = d.scale != nullptr;
        const int nlive = min(8, d.W - 8 * o);
        const bool al4 = (d.W & 3) == 0;
        const int ps = d.noct * 64;
        // x row group k = rows g0 - 1 + k RPI + j; dy row group k = rows g0 + k RPI + j
        auto load_x = [&](int k, float (&v)[8]) {
            const int gi = g0 - 1 + k * RPI + j;
            const bool ok = active && gi >= 0 && gi < d.rows;
            const uint32_t gg = ok ? (uint32_t)gi : 0u;
            const uint32_t n = d.dH.div(gg);
            const int y = (int)(gg - n * (uint32_t)d.H);
            const int off = ((((int)n * d.Cin + ci) * d.H + y) * d.W + 8 * o) * 4;
            if (al4) {
                const u32x4 t0 = __builtin_amdgcn_raw_buffer_load_b128(rsx, ok ? off : OOB, 0, 0);
                const u32x4 t1 = __builtin_amdgcn_raw_buffer_load_b128(rsx, ok && nlive > 4 ? off + 16 : OOB, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) { v[q] = __uint_as_float(t0[q]); v[4 + q] = __uint_as_float(t1[q]); }
            } else {                                            // rows that are not 16-byte multiples (14, 7 pixels)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    v[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsx, ok && e < nlive ? off + 4 * e : OOB, 0, 0));
            }
        };
        auto store_x = [&](int k, const float (&v)[8]) {
            const int gi = g0 - 1 + k * RPI + j;
            if (!(active && gi >= 0 && gi < d.rows)) return;
            float t[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                t[e] = v[e];
                if (tf) {
                    t[e] = fmaxf(fmaf(t[e], sc, sh), relu_lo);
                    t[e] = e < nlive ? t[e] : 0.f;
                }
            }
            u32x4 hi, mid, lo;
            split3x8(t, hi, mid, lo);
            u32x4* p = XR + (gi & (NX - 1)) * XS + o * 64 + r;
            p[0] = hi; p[ps] = mid; p[2 * ps] = lo;
        };
        auto load_dy = [&](int k, float (&v)[10]) {
            const int g = g0 + k * RPI + j;
            const bool ok = active && g < g1;
            const uint32_t gg = ok ? (uint32_t)g : 0u;
            const uint32_t n = d.dH.div(gg);
            const int y = (int)(gg - n * (uint32_t)d.H);
            const int off = ((((int)n * d.Cout + co) * d.H + y) * d.W + 8 * o) * 4;
            v[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsy, ok && o > 0 ? off - 4 : OOB, 0, 0));
            v[9] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsy, ok && nlive == 8 && 8 * o + 8 < d.W ? off + 32 : OOB, 0, 0));
            if (al4) {
                const u32x4 t0 = __builtin_amdgcn_raw_buffer_load_b128(rsy, ok ? off : OOB, 0, 0);
                const u32x4 t1 = __builtin_amdgcn_raw_buffer_load_b128(rsy, ok && nlive > 4 ? off + 16 : OOB, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) { v[1 + q] = __uint_as_float(t0[q]); v[5 + q] = __uint_as_float(t1[q]); }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    v[1 + e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsy, ok && e < nlive ? off + 4 * e : OOB, 0, 0));
            }
        };
        auto store_dy = [&](int k, const float (&v)[10]) {
            const int g = g0 + k * RPI + j;
            if (!(active && g < g1)) return;
            uint32_t pa[3][5];
#pragma unroll
            for (int q = 0; q < 5; ++q) split3(v[2 * q], v[2 * q + 1], pa[0][q], pa[1][q], pa[2][q]);
            u32x4* base = DY + (g & (ND - 1)) * DS + o * 64 + r;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                base[(0 * 3 + p) * ps] = u32x4{pa[p][1], pa[p][2], pa[p][3], pa[p][4]};
                u32x4 mid;
#pragma unroll
                for (int q = 0; q < 4; ++q) mid[q] = __builtin_amdgcn_alignbit(pa[p][q + 1], pa[p][q], 16);
                base[(1 * 3 + p) * ps] = mid;
                base[(2 * 3 + p) * ps] = u32x4{pa[p][0], pa[p][1], pa[p][2], pa[p][3]};
            }
        };
        float xr[8], dr[10];
        {
            float xa[NG][8], da[10];
#pragma unroll
            for (int k = 0; k < NG; ++k) load_x(k, xa[k]);
            load_dy(0, da);
            load_x(NG, xr);
            load_dy(1, dr);
#pragma unroll
            for (int k = 0; k < NG; ++k) store_x(k, xa[k]);
            store_dy(0, da);
        }
        __syncthreads();
        for (int it = 0; it < nit; ++it) {
            store_x(it + NG, xr);
            store_dy(it + 1, dr);
            load_x(it + NG + 1, xr);
            load_dy(it + 2, dr);
            __syncthreads();
        }
        return;
    }

    // ---------------------------------------------------------------------- consumers: sub-block (mi, ci), nine taps
    const int mi = wave >> 1, cw = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const int ks = d.noct >> 1;
    const int ps = d.noct * 64;
    f32x16 acc[3][3];                                            // [kh][kw]
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][s][i] = 0.f;
    int y;
    {
        const uint32_t n = d.dH.div((uint32_t)g0);
        y = g0 - (int)n * d.H;
    }
    const int fa = lh * 64 + mi * 32 + l31, fb = lh * 64 + cw * 32 + l31;
    __syncthreads();
    for (int it = 0; it < nit; ++it) {
        // the RPI rows of the group are RPI * ks steps of one loop: step u = (row jr, 16-pixel step q)
        for (int u = 0; u < RPI * ks; ++u) {
            const int jr = RPI == 1 ? 0 : u / ks, q = RPI == 1 ? u : u - jr * ks;
            const int g = g0 + it * RPI + jr;
            int yr = y + jr;
            if (yr >= d.H) yr -= d.H;
            if (g < g1) {
                const u32x4* ds = DY + (g & (ND - 1)) * DS + fa + q * 128;
                u32x4 a[3][3];
#pragma unroll
                for (int s = 0; s < 3; ++s)
#pragma unroll
                    for (int p = 0; p < 3; ++p) a[s][p] = ds[(s * 3 + p) * ps];
                static_for<3>([&](auto kh_tag) {
                    constexpr int KH = decltype(kh_tag)::value;
                    const int yi = yr + KH - 1;
                    if (yi >= 0 && yi < d.H) {
                        const u32x4* xs = XR + ((g + KH - 1 + NX) & (NX - 1)) * XS + fb + q * 128;
                        u32x4 b[3];
#pragma unroll
                        for (int p = 0; p < 3; ++p) b[p] = xs[p * ps];
#pragma unroll
                        for (int s = 0; s < 3; ++s) acc[KH][s] = mfma_split(a[s], b, acc[KH][s]);
                    }
                });
            }
        }
        y += RPI;
        if (y >= d.H) y -= d.H;
        __syncthreads();
    }
    float* out = d.slab + ((int64_t)z * d.Cout + mb * 64 + mi * 32) * N + (cb * 64 + cw * 32 + l31) * 9;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * lh;
                out[(int64_t)row * N + a * 3 + s] = acc[a][s][i];
            }
}

static int wg_rows_mode() {
    static const int m = [] { const char* e = getenv("SCAT_WG_ROWS"); return e ? atoi(e) : 1; }();
    return m;
}

static bool rows64_shape(int Cin, int H, int W, int Cout) {
    return Cin % 64 == 0 && Cout % 64 == 0 && W >= 7 && W <= 32 && H >= 2;
}

bool wgrad_rows_ok(int B, int Cin, int H, int W, int Cout, int KH, int stride, int pad, const void* dy, const void* x) {
    if (!(wg_rows_mode() && KH == 3 && stride == 1 && pad == 1 && (((uintptr_t)dy | (uintptr_t)x) & 15) == 0)) return false;
    if (Cin % 32 == 0 && Cout % 32 == 0 && Cin <= 64 && Cout <= 64 && W % 4 == 0 && W > 32 && W <= 64 && H >= 2) return true;
    if (!rows64_shape(Cin, H, W, Cout)) return false;
    if (wg_rows_mode() & 2) return true;                       // SCAT_WG_ROWS=3: wherever the kernel can run (tests)
    // measured at batch 96 (tools/conv_bench.py): ahead of the 128 x 128 producer/consumer kernel where a workgroup has
    // enough rows to amortise its prologue and epilogue — 128 -> 128 @28 188 -> 157 us, 256 -> 256 @14 209 -> 178 us —
    // behind it on 7 x 7 maps (7 of 16 pixels of a step are live) and on HRNet's small 64 / 128 / 256-channel branches
    const int64_t cc = (int64_t)Cin * Cout;
    return (W > 16 && cc >= 128 * 128) || (W > 8 && W <= 16 && cc >= 256 * 256);
}

static void rows_plan(int B, int Cin, int H, int W, int Cout, int& rpw, int& splits) {
    const bool big = W <= 32;                                 // 64 x 64 blocks
    const int rows = B * H, combos = big ? (Cin / 64) * (Cout / 64) : (Cin / 32) * (Cout / 32);
    static const int tgt = diag_env_int("SCAT_WG_ROWS_TARGET", 0);
    const int target = tgt > 0 ? tgt : 256;                   // one workgroup per CU (LDS)
    int s = (target + combos - 1) / combos;
    rpw = (rows + s - 1) / s;
    if (rpw < 8) rpw = 8;
    if (big && W <= 16) rpw = (rpw + 1) & ~1;                 // whole two-row groups
    if (rpw > rows) rpw = rows;
    splits = (rows + rpw - 1) / rpw;
}

int64_t wgrad_rows_ws(int B, int Cin, int H, int W, int Cout) {
    int rpw, splits;
    rows_plan(B, Cin, H, W, Cout, rpw, splits);
    return (int64_t)splits * Cout * Cin * 9 * sizeof(float);
}

// slab -> dw by launch_splitk_reduce (caller); returns the number of slabs
int wgrad_rows_launch(const float* dy, const float* x, float* slab, int B, int Cin, int H, int W, int Cout,
                      const float* in_scale, const float* in_shift, int in_relu, hipStream_t st) {
    int splits, rpw;
    rows_plan(B, Cin, H, W, Cout, rpw, splits);
    if (W <= 32) {
        RowWg64Desc d{};
        d.dy = dy; d.x = x; d.slab = slab; d.scale = in_scale; d.shift = in_shift; d.relu = in_scale ? in_relu : 0;
        d.B = B; d.Cin = Cin; d.Cout = Cout; d.H = H; d.W = W; d.rows = B * H; d.rpw = rpw; d.ncb = Cin / 64;
        d.noct = 2 * ((W + 15) / 16); d.nov = (W + 7) / 8;
        d.dH = FastDiv::make(H);
        d.ndy = (int64_t)B * Cout * H * W; d.nx = (int64_t)B * Cin * H * W;
        const int rpi = W <= 16 ? 2 : 1;
        d.rpi = rpi;
        const size_t lds_bytes = (size_t)(4 * rpi * 3 + 2 * rpi * 9) * d.noct * 64 * 16;
        static bool once1 = (hipFuncSetAttribute((const void*)wgrad3x3_rows64_kernel<1>,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 30 * 4 * 64 * 16) == hipSuccess);
        static bool once2 = (hipFuncSetAttribute((const void*)wgrad3x3_rows64_kernel<2>,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 60 * 2 * 64 * 16) == hipSuccess);
        (void)once1; (void)once2;
        set_kernel_label("wgrad3x3_rows_64x576x16%s_r%d_split%d", in_scale ? "_tf" : "", rpi, splits);
        d.ntile = d.ncb * (Cout / 64); d.nsplit = splits; d.ngroup = splitk_xcd_groups(splits);
        const dim3 grid(splitk_xcd_grid(d.ntile, splits, d.ngroup));
        if (rpi == 2) hipLaunchKernelGGL(wgrad3x3_rows64_kernel<2>, grid, dim3(512), lds_bytes, st, d);
        else hipLaunchKernelGGL(wgrad3x3_rows64_kernel<1>, grid, dim3(512), lds_bytes, st, d);
        return splits;
    }
    RowWgDesc d{};
    d.dy = dy; d.x = x; d.slab = slab; d.scale = in_scale; d.shift = in_shift; d.relu = in_scale ? in_relu : 0;
    d.B = B; d.Cin = Cin; d.Cout = Cout; d.H = H; d.W = W; d.rows = B * H; d.ncb = Cin / 32;
    d.noct = 2 * ((W + 15) / 16); d.nov = (W + 7) / 8;
    d.dH = FastDiv::make(H);
    d.ndy = (int64_t)B * Cout * H * W; d.nx = (int64_t)B * Cin * H * W;
    d.rpw = rpw;
    static const int stamp = diag_env_int("SCAT_WG_ROWS_STAMP", 0);
    d.stamp = kDiag ? stamp : 0;
    const size_t lds_bytes = (size_t)(4 * 3 + 2 * 9) * d.noct * 32 * 16;
    static bool once = (hipFuncSetAttribute((const void*)wgrad3x3_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)((4 * 3 + 2 * 9) * 8 * 32 * 16)) == hipSuccess);
    (void)once;
    set_kernel_label("wgrad3x3_rows_32x288x16%s_split%d", in_scale ? "_tf" : "", splits);
    d.ntile = d.ncb * (Cout / 32); d.nsplit = splits; d.ngroup = splitk_xcd_groups(splits);
    hipLaunchKernelGGL(wgrad3x3_rows_kernel, dim3(splitk_xcd_grid(d.ntile, splits, d.ngroup)), dim3(RW_NT), lds_bytes, st, d);
    return splits;
}

}  // namespace scat

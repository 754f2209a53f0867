// Weight gradient of the ResNet stem, Conv2d(3, 64, 7, stride 2, padding 3) (models/resnet.py:105), on split-operand
// products:   dW[co][(c, kh, kw)] = sum over (n, oy, ox)  dy[n][co][oy][ox] * x[n][c][2 oy + kh - 3][2 ox + kw - 3]
// a 64 x 147 output with a contraction over 1.2 M pixels at batch 96.  The general weight-gradient kernels stage both
// operands through LDS as split planes; here the second operand is a stride-2 gather of 147 rows (8 dwords per row and
// pixel octet through the texture path: 438 us on the fp32 engine, the un-overlapped tail of the backward).  Instead:
//  * one workgroup walks whole OUTPUT ROWS.  The 7 input rows x 3 channels an output row touches (4.9 k floats) are
//    copied once into LDS as they are (zero padding included), double-buffered against the previous row's MFMAs;
//  * the B fragment of column (c, kh, kw) and pixel octet h is 8 stride-2 LDS reads from that patch, split in registers;
//    the A fragment (dy) is 8 consecutive pixels of one channel row straight from global memory, split in registers —
//    no split planes ever exist in LDS or HBM;
//  * the 64 x 160 accumulator (147 columns padded to 5 blocks of 32) lives in the four wavefronts (2 row blocks x
//    {3, 2} column blocks); deterministic split over output rows into fp32 slabs, fixed-order reduce.
#include "conv_common.h"
#include "split.h"

namespace scat {

struct StemWgDesc {
    const float* dy;      // [B][64][OH][OW]
    const float* x;       // [B][3][H][W]
    float* slab;          // [splits][64][147]
    int B, H, W, OH, OW;
    int rows;             // B * OH output rows in total
    int rpw;              // output rows per workgroup
    int64_t ndy, nx;
};

constexpr int SW_PW = 232;                       // patch row pitch in floats (>= 2*OW + 5 for OW <= 112, 16-B multiple)
constexpr int SW_PATCH = 3 * 7 * SW_PW;          // floats per patch

__global__ __launch_bounds__(256) void stem_wgrad_split_kernel(StemWgDesc d) {
    extern __shared__ __align__(16) float lds[];           // [2 patches][3][7][SW_PW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int rb = wave & 1;                                // 32-row block of the 64 output channels
    const int cb0 = (wave >> 1) * 3, ncb = (wave >> 1) ? 2 : 3;      // column blocks {0,1,2} / {3,4}
    const int r_beg = blockIdx.x * d.rpw, r_end = min(r_beg + d.rpw, d.rows);

    const __amdgpu_buffer_rsrc_t rsx = make_rsrc(d.x, d.nx), rsy = make_rsrc(d.dy, d.ndy);

    // ---- patch staging: thread item i -> (plane q = c*7 + r, column v): value x[n][c][2 oy + r - 3][v - 3]
    constexpr int NP = (3 * 7 * SW_PW + 255) / 256;         // items per thread
    float pst[NP];
    auto load_patch = [&](int row) {
        const bool live = row < r_end;
        const int n = live ? row / d.OH : 0, oy = live ? row - n * d.OH : 0;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int it = tid + 256 * i;
            const int q = it / SW_PW, v = it - q * SW_PW;
            const int c = q / 7, r = q - 7 * c;
            const int iy = 2 * oy + r - 3, ix = v - 3;
            const bool ok = live && it < SW_PATCH && (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
            pst[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                rsx, ok ? (((n * 3 + c) * d.H + iy) * d.W + ix) * 4 : OOB, 0, 0));
        }
    };
    auto store_patch = [&](float* dst) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int it = tid + 256 * i;
            if (it < SW_PATCH) dst[it] = pst[i];
        }
    };

    // ---- this lane's columns: n = 32 cb + l31 -> (c, kh, kw); the patch offset of pixel 0, element 0
    int boff[3];
    bool bok[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int n = (cb0 + t) * 32 + l31;
        bok[t] = t < ncb && n < 147;
        const int nn = bok[t] ? n : 0;
        const int c = nn / 49, kh = (nn - 49 * c) / 7, kw = nn - 49 * c - 7 * kh;
        boff[t] = (c * 7 + kh) * SW_PW + kw + 16 * lh;      // + 2 * (pixel in the row) : octet lh starts 8 pixels = 16 floats in
    }

    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int nsteps = d.OW / 16;                           // k16-steps per output row (OW % 16 == 0)
    float araw[2][8];
    auto load_a = [&](int row, int step, auto set_tag) {
        constexpr int Q = decltype(set_tag)::value;
        const bool live = row < r_end;
        const int n = live ? row / d.OH : 0, oy = live ? row - n * d.OH : 0;
        const int off = live ? ((((n * 64 + rb * 32 + l31) * d.OH + oy) * d.OW) + 16 * step + 8 * lh) * 4 : OOB;
        const u32x4 t0 = __builtin_amdgcn_raw_buffer_load_b128(rsy, off, 0, 0);
        const u32x4 t1 = __builtin_amdgcn_raw_buffer_load_b128(rsy, off, 16, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) { araw[Q][e] = __uint_as_float(t0[e]); araw[Q][4 + e] = __uint_as_float(t1[e]); }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;

    load_patch(r_beg);
    store_patch(lds);
    load_a(r_beg, 0, S0{});
    __syncthreads();
    int pbuf = 0;
    // one output row; P = the register set that holds its first A fragment (an odd step count flips it every row)
    auto row_body = [&](int row, auto p_tag) {
        constexpr int P = decltype(p_tag)::value;
        const float* patch = lds + pbuf * SW_PATCH;
        load_patch(row + 1);                                 // next row's patch rides under this row's MFMAs
        auto step = [&](int s, auto cur_tag) {
            constexpr int CUR = decltype(cur_tag)::value;
            // next A fragment: the next step of this row, or step 0 of the next row
            if (s + 1 < nsteps) load_a(row, s + 1, std::integral_constant<int, CUR ^ 1>{});
            else load_a(row + 1, 0, std::integral_constant<int, CUR ^ 1>{});
            u32x4 a[3];
            split3x8(araw[CUR], a[0], a[1], a[2]);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                if (t < ncb) {
                    float v[8];
                    const float* p = patch + boff[t] + 32 * s;          // 16 pixels per step = 32 floats
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = bok[t] ? p[2 * e] : 0.f;
                    u32x4 b[3];
                    split3x8(v, b[0], b[1], b[2]);
                    acc[t] = mfma_split(a, b, acc[t]);
                }
            }
        };
        for (int s = 0; s < nsteps; s += 2) {
            step(s, std::integral_constant<int, P>{});
            if (s + 1 < nsteps) step(s + 1, std::integral_constant<int, P ^ 1>{});
        }
        store_patch(lds + (pbuf ^ 1) * SW_PATCH);
        pbuf ^= 1;
        __syncthreads();
    };
    int par = 0;
    for (int row = r_beg; row < r_end; ++row) {
        if (par == 0) row_body(row, S0{});
        else row_body(row, S1{});
        par ^= nsteps & 1;
    }

    // ---- slab store: C/D map col = l31, row = (r & 3) + 8 (r >> 2) + 4 lh
    float* out = d.slab + (int64_t)blockIdx.x * 64 * 147;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        if (t < ncb) {
            const int n = (cb0 + t) * 32 + l31;
            if (n < 147) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    out[co * 147 + n] = acc[t][r];
                }
            }
        }
    }
}

}  // namespace scat

using namespace scat;

static int stem_wg_rows_per_wg(int rows) {
    int rpw = (rows + 767) / 768;                          // ~3 workgroups per CU
    if (rpw < 4) rpw = 4;
    return rpw;
}

extern "C" int64_t scat_conv7x7_s2_wgrad_split_ws(int B, int H, int W) {
    const int OH = (H + 6 - 7) / 2 + 1;
    const int rows = B * OH, rpw = stem_wg_rows_per_wg(rows);
    return (int64_t)((rows + rpw - 1) / rpw) * 64 * 147 * 4;
}

// dw[64,3,7,7] = weight gradient of Conv2d(3, 64, 7, stride 2, padding 3) from dy[B,64,OH,OW] and x[B,3,H,W].
// OW % 16 == 0, OW <= 112 (the reference geometry: 224 -> 112).  ws: scat_conv7x7_s2_wgrad_split_ws(B, H, W) bytes.
extern "C" int scat_conv7x7_s2_wgrad_split(const float* dy, const float* x, float* dw, int B, int H, int W, int Cout,
                                           void* ws, int64_t ws_bytes, void* stream) {
    SCAT_REQUIRE(dy && x && dw, SCAT_E_ARG, "scat_conv7x7_s2_wgrad_split: null pointer");
    SCAT_REQUIRE(math_mode() == 1, SCAT_E_ARG, "scat_conv7x7_s2_wgrad_split: needs the split-operand product mode");
    const int OH = (H + 6 - 7) / 2 + 1, OW = (W + 6 - 7) / 2 + 1;
    SCAT_REQUIRE(B > 0 && Cout == 64 && OW % 16 == 0 && OW <= 112 && 2 * OW + 5 <= SW_PW, SCAT_E_SHAPE,
                 "scat_conv7x7_s2_wgrad_split: Cout = 64, output width a multiple of 16 up to 112 (got Cout %d OW %d)", Cout, OW);
    SCAT_REQUIRE(ws && ws_bytes >= scat_conv7x7_s2_wgrad_split_ws(B, H, W) && ((uintptr_t)ws & 15) == 0 &&
                     ((uintptr_t)dy & 15) == 0,
                 SCAT_E_WORKSPACE, "scat_conv7x7_s2_wgrad_split: workspace too small / unaligned");
    SCAT_REQUIRE(fits_i32((int64_t)B * 64 * OH * OW * 4) && fits_i32((int64_t)B * 3 * H * W * 4), SCAT_E_SHAPE,
                 "scat_conv7x7_s2_wgrad_split: tensor exceeds 32-bit byte offsets");
    hipStream_t st = (hipStream_t)stream;
    StemWgDesc d{};
    d.dy = dy; d.x = x; d.slab = (float*)ws; d.B = B; d.H = H; d.W = W; d.OH = OH; d.OW = OW;
    d.rows = B * OH; d.rpw = stem_wg_rows_per_wg(d.rows);
    d.ndy = (int64_t)B * 64 * OH * OW; d.nx = (int64_t)B * 3 * H * W;
    const int splits = (d.rows + d.rpw - 1) / d.rpw;
    set_kernel_label("wgrad7x7_s2_split_64x160x16_split%d", splits);
    hipLaunchKernelGGL(stem_wgrad_split_kernel, dim3(splits), dim3(256), (size_t)2 * SW_PATCH * 4, st, d);
    SCAT_LAUNCH_CHECK("scat_conv7x7_s2_wgrad_split");
    launch_splitk_reduce((const float*)ws, dw, (int64_t)64 * 147, splits, 0, st);
    SCAT_LAUNCH_CHECK("scat_conv7x7_s2_wgrad_split(reduce)");
    return SCAT_OK;
}

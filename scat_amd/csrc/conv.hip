// Convolution forward / data-gradient / weight-gradient on the fp32 MFMA engine.
// Replaces nn.Conv2d (+ its autograd) as called at models/resnet.py:65-72,105,129 and
// models/hand_net.py:329 of the reference.
#include "gemm_engine.h"

namespace scat {

enum Cfg { C128x128, C128x64, C64x128, C64x64 };

static Cfg pick_cfg(int M, int N) {
    // largest tile that still gives the 256 CUs >= ~1.5 workgroups each
    auto tiles = [&](int bm, int bn) { return (int64_t)cdiv(M, bm) * cdiv(N, bn); };
    if (M > 64 && tiles(128, 128) >= 384) return C128x128;
    if (M > 64 && tiles(128, 64) >= 384) return C128x64;
    if (M <= 64) return tiles(64, 128) >= 384 ? C64x128 : C64x64;
    if (tiles(128, 64) >= 256) return C128x64;
    return C64x64;
}

// forward / data-gradient share one kernel family: A = weights (K contiguous), B = gather, pixel = column
template <int KH, int KW, bool D2>
static void conv_gemm(const MatDesc& da, const GatherDesc& db, const OutDesc& dc, int M, int N, int K,
                      hipStream_t st) {
    const Cfg cfg = pick_cfg(M, N);
    static const char* const names[] = {"128x128", "128x64", "64x128", "64x64"};
    set_kernel_label("conv_gather%dx%d%s_%sx16", KH, KW, D2 ? "_d2" : "", names[cfg]);
    switch (cfg) {
        case C128x128:
            launch_gemm<MatLoader<128, 16, true>, GatherLoader<128, 16, KH, KW, D2, false>, 128, 128, 16, 2, 2>(
                da, db, dc, M, N, K, 1, st);
            break;
        case C128x64:
            launch_gemm<MatLoader<128, 16, true>, GatherLoader<64, 16, KH, KW, D2, false>, 128, 64, 16, 2, 2>(
                da, db, dc, M, N, K, 1, st);
            break;
        case C64x128:
            launch_gemm<MatLoader<64, 16, true>, GatherLoader<128, 16, KH, KW, D2, false>, 64, 128, 16, 2, 2>(
                da, db, dc, M, N, K, 1, st);
            break;
        default:
            launch_gemm<MatLoader<64, 16, true>, GatherLoader<64, 16, KH, KW, D2, false>, 64, 64, 16, 2, 2>(
                da, db, dc, M, N, K, 1, st);
    }
}

static int check_geom(const char* who, int B, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad,
                      int* OH, int* OW) {
    SCAT_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, SCAT_E_SHAPE, "%s: non-positive dimension", who);
    SCAT_REQUIRE(KH == KW && (KH == 1 || KH == 3 || KH == 7), SCAT_E_SHAPE, "%s: kernel %dx%d unsupported", who, KH,
                 KW);
    SCAT_REQUIRE(stride == 1 || stride == 2, SCAT_E_SHAPE, "%s: stride %d unsupported", who, stride);
    SCAT_REQUIRE(pad >= 0 && pad < KH + (KH == 1), SCAT_E_SHAPE, "%s: pad %d unsupported", who, pad);
    *OH = (H + 2 * pad - KH) / stride + 1;
    *OW = (W + 2 * pad - KW) / stride + 1;
    SCAT_REQUIRE(*OH > 0 && *OW > 0, SCAT_E_SHAPE, "%s: empty output", who);
    SCAT_REQUIRE(fits_i32((int64_t)B * Cin * H * W) && fits_i32((int64_t)B * Cout * *OH * *OW) &&
                     fits_i32((int64_t)Cout * Cin * KH * KW),
                 SCAT_E_SHAPE, "%s: tensor exceeds 2^31 elements", who);
    return SCAT_OK;
}

__global__ void wt_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout, int Cin, int KK) {
    // wt[ci][co*KK + t] = w[co][ci][t]
    int64_t n = (int64_t)Cout * Cin * KK;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        int t = e % KK;
        int64_t r = e / KK;
        int co = r % Cout, ci = r / Cout;
        wt[e] = w[((int64_t)co * Cin + ci) * KK + t];
    }
}

__global__ void splitk_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out, int64_t n, int splits,
                                     int accumulate) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        float s = accumulate ? out[e] : 0.f;
        for (int z = 0; z < splits; ++z) s += slab[(int64_t)z * n + e];
        out[e] = s;
    }
}

static int wgrad_splits(int M, int N, int K, int bm, int bn) {
    int tiles = cdiv(M, bm) * cdiv(N, bn);
    int s = cdiv(1024, tiles);                       // aim for ~4 workgroups per CU
    int smax = K / 512 > 0 ? K / 512 : 1;            // keep >= 512 contraction steps per slice
    if (s > smax) s = smax;
    if (s > 256) s = 256;
    return s < 1 ? 1 : s;
}

struct WgradPlan {
    int M, N, K, bm, bn, splits;
};
static WgradPlan wgrad_plan(int B, int Cin, int Cout, int KK, int OH, int OW) {
    WgradPlan p;
    p.M = Cout;
    p.N = Cin * KK;
    p.K = B * OH * OW;
    p.bm = Cout <= 64 ? 64 : 128;
    p.bn = 64;
    p.splits = wgrad_splits(p.M, p.N, p.K, p.bm, p.bn);
    return p;
}

template <int KH, int KW>
static void wgrad_gemm(const WgradPlan& p, const GatherDesc& da, const GatherDesc& db, const OutDesc& dc,
                       hipStream_t st) {
    set_kernel_label("wgrad_gather%dx%d_%dx64x32_split%d", KH, KW, p.bm, p.splits);
    if (p.bm == 64)
        launch_gemm<GatherLoader<64, 32, 1, 1, false, true>, GatherLoader<64, 32, KH, KW, false, true>, 64, 64, 32, 2,
                    2>(da, db, dc, p.M, p.N, p.K, p.splits, st);
    else
        launch_gemm<GatherLoader<128, 32, 1, 1, false, true>, GatherLoader<64, 32, KH, KW, false, true>, 128, 64, 32,
                    2, 2>(da, db, dc, p.M, p.N, p.K, p.splits, st);
}

}  // namespace scat

using namespace scat;

extern "C" int scat_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int H,
                               int W, int Cout, int KH, int KW, int stride, int pad, const float* in_scale,
                               const float* in_shift, int in_relu, void* stream) {
    int OH, OW;
    if (int e = check_geom("scat_conv2d_fwd", B, Cin, H, W, Cout, KH, KW, stride, pad, &OH, &OW)) return e;
    SCAT_REQUIRE(x && w && y, SCAT_E_ARG, "scat_conv2d_fwd: null pointer");
    SCAT_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), SCAT_E_ARG, "scat_conv2d_fwd: scale/shift pair");
    const int KK = KH * KW, K = Cin * KK, N = B * OH * OW;
    MatDesc da{w, K, 1, 0, Cout, K};
    GatherDesc db{x, in_scale, in_shift, in_relu, Cin, H, W, OH, OW, stride, 1, -pad, N, K,
                  FastDiv::make(OH * OW), FastDiv::make(OW)};
    OutDesc dc{};
    dc.p = y; dc.mode = 1; dc.I = Cout; dc.J = N; dc.C = Cout; dc.HW = OH * OW; dc.dHW = FastDiv::make(OH * OW);
    dc.bias = bias; dc.bias_mode = bias ? 1 : 0; dc.accumulate = 0;
    hipStream_t st = (hipStream_t)stream;
    if (KH == 1) conv_gemm<1, 1, false>(da, db, dc, Cout, N, K, st);
    else if (KH == 3) conv_gemm<3, 3, false>(da, db, dc, Cout, N, K, st);
    else conv_gemm<7, 7, false>(da, db, dc, Cout, N, K, st);
    SCAT_LAUNCH_CHECK("scat_conv2d_fwd");
    return SCAT_OK;
}

extern "C" int scat_conv2d_wt(const float* w, float* wt, int Cout, int Cin, int KH, int KW, void* stream) {
    SCAT_REQUIRE(w && wt && Cout > 0 && Cin > 0 && KH > 0 && KW > 0, SCAT_E_ARG, "scat_conv2d_wt: bad argument");
    int64_t n = (int64_t)Cout * Cin * KH * KW;
    int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(wt_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wt, Cout, Cin, KH * KW);
    SCAT_LAUNCH_CHECK("scat_conv2d_wt");
    return SCAT_OK;
}

extern "C" int scat_conv2d_dgrad(const float* dy, const float* wt, float* dx, int B, int Cin, int H, int W, int Cout,
                                 int KH, int KW, int stride, int pad, int accumulate, void* stream) {
    int OH, OW;
    if (int e = check_geom("scat_conv2d_dgrad", B, Cin, H, W, Cout, KH, KW, stride, pad, &OH, &OW)) return e;
    SCAT_REQUIRE(dy && wt && dx, SCAT_E_ARG, "scat_conv2d_dgrad: null pointer");
    SCAT_REQUIRE(KH != 7, SCAT_E_SHAPE, "scat_conv2d_dgrad: 7x7 not needed on this path (stem input has no grad)");
    const int KK = KH * KW, K = Cout * KK, N = B * H * W;
    MatDesc da{wt, K, 1, 0, Cin, K};
    // source = dy[B,Cout,OH,OW]; pixel grid = input pixels; t = y*1 + kh*(-1) + pad, divisor = stride
    GatherDesc db{dy, nullptr, nullptr, 0, Cout, OH, OW, H, W, 1, -1, pad, N, K, FastDiv::make(H * W),
                  FastDiv::make(W)};
    OutDesc dc{};
    dc.p = dx; dc.mode = 1; dc.I = Cin; dc.J = N; dc.C = Cin; dc.HW = H * W; dc.dHW = FastDiv::make(H * W);
    dc.accumulate = accumulate;
    hipStream_t st = (hipStream_t)stream;
    if (stride == 1) {
        if (KH == 1) conv_gemm<1, 1, false>(da, db, dc, Cin, N, K, st);
        else conv_gemm<3, 3, false>(da, db, dc, Cin, N, K, st);
    } else {
        if (KH == 1) conv_gemm<1, 1, true>(da, db, dc, Cin, N, K, st);
        else conv_gemm<3, 3, true>(da, db, dc, Cin, N, K, st);
    }
    SCAT_LAUNCH_CHECK("scat_conv2d_dgrad");
    return SCAT_OK;
}

extern "C" int64_t scat_conv2d_wgrad_ws(int B, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad) {
    int OH, OW;
    if (check_geom("scat_conv2d_wgrad_ws", B, Cin, H, W, Cout, KH, KW, stride, pad, &OH, &OW)) return -1;
    WgradPlan p = wgrad_plan(B, Cin, Cout, KH * KW, OH, OW);
    return p.splits > 1 ? (int64_t)p.splits * p.M * p.N * sizeof(float) : 0;
}

extern "C" int scat_conv2d_wgrad(const float* dy, const float* x, float* dw, int B, int Cin, int H, int W, int Cout,
                                 int KH, int KW, int stride, int pad, const float* in_scale, const float* in_shift,
                                 int in_relu, void* ws, int64_t ws_bytes, void* stream) {
    int OH, OW;
    if (int e = check_geom("scat_conv2d_wgrad", B, Cin, H, W, Cout, KH, KW, stride, pad, &OH, &OW)) return e;
    SCAT_REQUIRE(dy && x && dw, SCAT_E_ARG, "scat_conv2d_wgrad: null pointer");
    SCAT_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), SCAT_E_ARG, "scat_conv2d_wgrad: scale/shift pair");
    WgradPlan p = wgrad_plan(B, Cin, Cout, KH * KW, OH, OW);
    int64_t need = p.splits > 1 ? (int64_t)p.splits * p.M * p.N * sizeof(float) : 0;
    SCAT_REQUIRE(ws_bytes >= need && (need == 0 || ws), SCAT_E_WORKSPACE,
                 "scat_conv2d_wgrad: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)need);
    const int npix = B * OH * OW;
    // A: dy as [Cout][pixel]; B: x through the forward conv arithmetic as [pixel][(ci,kh,kw)]
    GatherDesc da{dy, nullptr, nullptr, 0, Cout, OH, OW, OH, OW, 1, 0, 0, npix, Cout, FastDiv::make(OH * OW),
                  FastDiv::make(OW)};
    GatherDesc db{x, in_scale, in_shift, in_relu, Cin, H, W, OH, OW, stride, 1, -pad, npix, Cin * KH * KW,
                  FastDiv::make(OH * OW), FastDiv::make(OW)};
    OutDesc dc{};
    dc.p = p.splits > 1 ? (float*)ws : dw;
    dc.mode = 0; dc.si = p.N; dc.sj = 1; dc.sz = (int64_t)p.M * p.N; dc.I = p.M; dc.J = p.N;
    hipStream_t st = (hipStream_t)stream;
    if (KH == 1) wgrad_gemm<1, 1>(p, da, db, dc, st);
    else if (KH == 3) wgrad_gemm<3, 3>(p, da, db, dc, st);
    else wgrad_gemm<7, 7>(p, da, db, dc, st);
    SCAT_LAUNCH_CHECK("scat_conv2d_wgrad");
    if (p.splits > 1) {
        int64_t n = (int64_t)p.M * p.N;
        int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)ws, dw, n, p.splits, 0);
        SCAT_LAUNCH_CHECK("scat_conv2d_wgrad(reduce)");
    }
    return SCAT_OK;
}

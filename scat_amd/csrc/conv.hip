// Convolution forward / data-gradient on the fp32 MFMA engine (weight-gradient: conv_wgrad.hip).
// Replaces nn.Conv2d (+ its autograd) as called at models/resnet.py:65-72,105,129 and
// models/hand_net.py:329 of the reference.
#include <string.h>

#include "conv_common.h"

namespace scat {

enum Cfg { C128x128, C128x64, C64x128, C64x64 };

static Cfg pick_cfg(int M, int N) {
    // largest tile that still gives the 256 CUs >= ~1.5 workgroups each
    auto tiles = [&](int bm, int bn) { return (int64_t)cdiv(M, bm) * cdiv(N, bn); };
    if (M > 64 && tiles(128, 128) >= 384) return C128x128;
    if (M > 64 && tiles(128, 64) >= 384) return C128x64;
    if (M <= 64) return tiles(64, 128) >= 384 ? C64x128 : C64x64;
    if (tiles(128, 64) >= 400) return C128x64;
    return C64x64;   // under-filled grids (layer4: N = 4704): more, smaller workgroups
}

// forward / data-gradient share one kernel family: A = weights (K contiguous), B = gather, pixel = column
template <int KH, int KW, bool D2, int AV, bool BV4, bool TF>
static void conv_gemm_cfg(Cfg cfg, const MatDesc& da, const GatherDesc& db, const OutDesc& dc, int M, int N, int K,
                          hipStream_t st) {
    static const char* const names[] = {"128x128", "128x64", "64x128", "64x64"};
    set_kernel_label("conv%dx%d%s_%sx16_a%d%s%s", KH, KW, D2 ? "_d2" : "", names[cfg], AV, BV4 ? "_b4" : "",
                     TF ? "_tf" : "");
    switch (cfg) {
        case C128x128:
            launch_gemm<MatLoader<128, 16, true, AV>, GatherLoader<128, 16, KH, KW, D2, false, BV4, TF>, 128, 128, 16,
                        2, 2>(da, db, dc, M, N, K, 1, st);
            break;
        case C128x64:
            launch_gemm<MatLoader<128, 16, true, AV>, GatherLoader<64, 16, KH, KW, D2, false, BV4, TF>, 128, 64, 16, 2,
                        2>(da, db, dc, M, N, K, 1, st);
            break;
        case C64x128:
            launch_gemm<MatLoader<64, 16, true, AV>, GatherLoader<128, 16, KH, KW, D2, false, BV4, TF>, 64, 128, 16, 2,
                        2>(da, db, dc, M, N, K, 1, st);
            break;
        default:
            launch_gemm<MatLoader<64, 16, true, AV>, GatherLoader<64, 16, KH, KW, D2, false, BV4, TF>, 64, 64, 16, 2,
                        2>(da, db, dc, M, N, K, 1, st);
    }
}

template <int KH, int KW, bool D2, bool TF>
static void conv_gemm_tf(const MatDesc& da, const GatherDesc& db, const OutDesc& dc, int M, int N, int K, bool bv4,
                         hipStream_t st) {
    const bool av4 = (K % 4 == 0) && (((uintptr_t)da.p & 15) == 0);
    if (!av4) {   // odd contraction length (the 7x7 stem: K = 147): scalar weight loads
        conv_gemm_cfg<KH, KW, D2, 1, false, TF>(KH == 7 ? pick_cfg(M, N) : C64x64, da, db, dc, M, N, K, st);
        return;
    }
    if constexpr (KH * KW <= 9 && !D2) {
        if (bv4) {
            conv_gemm_cfg<KH, KW, false, 4, true, TF>(pick_cfg(M, N), da, db, dc, M, N, K, st);
            return;
        }
    }
    if constexpr (KH != 7) conv_gemm_cfg<KH, KW, D2, 4, false, TF>(pick_cfg(M, N), da, db, dc, M, N, K, st);
    else conv_gemm_cfg<KH, KW, D2, 1, false, TF>(pick_cfg(M, N), da, db, dc, M, N, K, st);
}

template <int KH, int KW, bool D2>
static void conv_gemm(const MatDesc& da, const GatherDesc& db, const OutDesc& dc, int M, int N, int K, bool bv4,
                      hipStream_t st) {
    if constexpr (KH != 7 && !D2) {   // the fused BatchNorm+ReLU input transform exists on forward 1x1 / 3x3 only
        if (db.scale) {
            conv_gemm_tf<KH, KW, D2, true>(da, db, dc, M, N, K, bv4, st);
            return;
        }
    }
    conv_gemm_tf<KH, KW, D2, false>(da, db, dc, M, N, K, bv4, st);
}

// stride-2 data-gradient, decomposed by input-pixel parity: class (py,px) is a stride-1 gather over dy with the
// KHc x KWc taps of matching parity (1..4 of the 9), so no MFMA work is spent on structurally-zero taps.
__global__ void wt_class_kernel(const float* __restrict__ w, float* __restrict__ wtc, int Cout, int Cin, int KH,
                                int KW, int kh0, int kw0, int KHc, int KWc) {
    // wtc[ci][(co*KHc + kky)*KWc + kkx] = w[co][ci][kh0 + 2*kky][kw0 + 2*kkx]
    const int KKc = KHc * KWc;
    int64_t n = (int64_t)Cin * Cout * KKc;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        int t = e % KKc;
        int64_t r = e / KKc;
        int co = r % Cout, ci = r / Cout;
        int kky = t / KWc, kkx = t % KWc;
        wtc[e] = w[(((int64_t)co * Cin + ci) * KH + kh0 + 2 * kky) * KW + kw0 + 2 * kkx];
    }
}

template <int KHc, int KWc>
static void dgrad_class_gemm(const MatDesc& da, const GatherDesc& db, const OutDesc& dc, int M, int N, int K, bool bv4,
                             hipStream_t st) {
    const Cfg cfg = pick_cfg(M, N);
    static const char* const names[] = {"128x128", "128x64", "64x128", "64x64"};
    set_kernel_label("dgrad_s2_class%dx%d_%sx16%s", KHc, KWc, names[cfg], bv4 ? "_b4" : "");
    if (bv4) {
        conv_gemm_cfg<KHc, KWc, false, 4, true, false>(cfg, da, db, dc, M, N, K, st);
        set_kernel_label("dgrad_s2_class%dx%d_%sx16_b4", KHc, KWc, names[cfg]);
        return;
    }
    conv_gemm_cfg<KHc, KWc, false, 4, false, false>(cfg, da, db, dc, M, N, K, st);
    set_kernel_label("dgrad_s2_class%dx%d_%sx16", KHc, KWc, names[cfg]);
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

}  // namespace scat

using namespace scat;

extern "C" int scat_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int H,
                               int W, int Cout, int KH, int KW, int stride, int pad, const float* in_scale,
                               const float* in_shift, int in_relu, void* stream) {
    int OH, OW;
    if (int e = check_geom("scat_conv2d_fwd", B, Cin, H, W, Cout, KH, KW, stride, pad, &OH, &OW)) return e;
    SCAT_REQUIRE(x && w && y, SCAT_E_ARG, "scat_conv2d_fwd: null pointer");
    SCAT_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), SCAT_E_ARG, "scat_conv2d_fwd: scale/shift pair");
    SCAT_REQUIRE(!(in_scale && KH == 7), SCAT_E_SHAPE, "scat_conv2d_fwd: fused input transform not built for 7x7");
    if (!in_scale) in_relu = 0;
    const int KK = KH * KW, K = Cin * KK, N = B * OH * OW;
    MatDesc da{w, K, 1, 0, Cout, K, (int64_t)Cout * K};
    GatherDesc db{x, in_scale, in_shift, in_relu, Cin, H, W, OH, OW, stride, 1, -pad, -pad, N, K,
                  FastDiv::make(OH * OW), FastDiv::make(OW), (int64_t)B * Cin * H * W};
    OutDesc dc{};
    dc.p = y; dc.mode = 1; dc.I = Cout; dc.J = N; dc.C = Cout; dc.HW = OH * OW; dc.dHW = FastDiv::make(OH * OW);
    dc.bias = bias; dc.bias_mode = bias ? 1 : 0; dc.accumulate = 0; dc.n = (int64_t)B * Cout * OH * OW;
    hipStream_t st = (hipStream_t)stream;
    // 16-B pixel vectors: stride 1, output grid == input grid (1x1 pad 0 / 3x3 pad 1), plane a multiple of 4
    const bool bv4 = stride == 1 && OH == H && OW == W && (H * W) % 4 == 0 && W >= 4 && ((uintptr_t)x & 15) == 0;
    if (KH == 1) conv_gemm<1, 1, false>(da, db, dc, Cout, N, K, bv4, st);
    else if (KH == 3) conv_gemm<3, 3, false>(da, db, dc, Cout, N, K, bv4, st);
    else conv_gemm<7, 7, false>(da, db, dc, Cout, N, K, false, st);
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
    MatDesc da{wt, K, 1, 0, Cin, K, (int64_t)Cin * K};
    // source = dy[B,Cout,OH,OW]; pixel grid = input pixels; t = y*1 + kh*(-1) + pad, divisor = stride
    GatherDesc db{dy, nullptr, nullptr, 0, Cout, OH, OW, H, W, 1, -1, pad, pad, N, K, FastDiv::make(H * W),
                  FastDiv::make(W), (int64_t)B * Cout * OH * OW};
    OutDesc dc{};
    dc.p = dx; dc.mode = 1; dc.I = Cin; dc.J = N; dc.C = Cin; dc.HW = H * W; dc.dHW = FastDiv::make(H * W);
    dc.accumulate = accumulate; dc.n = (int64_t)B * Cin * H * W;
    hipStream_t st = (hipStream_t)stream;
    const bool bv4 = stride == 1 && OH == H && OW == W && (H * W) % 4 == 0 && W >= 4 && ((uintptr_t)dy & 15) == 0;
    if (stride == 1) {
        if (KH == 1) conv_gemm<1, 1, false>(da, db, dc, Cin, N, K, bv4, st);
        else conv_gemm<3, 3, false>(da, db, dc, Cin, N, K, bv4, st);
    } else {
        if (KH == 1) conv_gemm<1, 1, true>(da, db, dc, Cin, N, K, false, st);
        else conv_gemm<3, 3, true>(da, db, dc, Cin, N, K, false, st);
    }
    SCAT_LAUNCH_CHECK("scat_conv2d_dgrad");
    return SCAT_OK;
}

extern "C" int64_t scat_conv2d_dgrad_s2_ws(int Cin, int Cout, int KH, int KW) {
    const int64_t f32 = (int64_t)Cin * Cout * KH * KW * sizeof(float);
    const int64_t split = taps_split_ws(Cin, Cout, KH * KW);           // every parity class has its own slice
    return f32 > split ? f32 : split;
}

// the parity classes of the stride-2 data gradient, in launch order: taps (kh0 + 2*th, kw0 + 2*tw) of w
struct S2Class { int py, px, kh0, kw0, KHc, KWc; int64_t ws_off; };
static int dgrad_s2_classes(int Cin, int Cout, int KH, int KW, int pad, S2Class* out) {
    int n = 0;
    int64_t off = 0;
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            const int kh0 = (py + pad) & 1, kw0 = (px + pad) & 1;
            const int KHc = (KH - kh0 + 1) / 2, KWc = (KW - kw0 + 1) / 2;   // taps of matching parity
            if (KHc <= 0 || KWc <= 0) continue;                             // 1x1: only class (0,0) has a tap
            out[n++] = S2Class{py, px, kh0, kw0, KHc, KWc, off};
            off += taps_split_ws(Cin, Cout, KHc * KWc);
        }
    return n;
}

namespace scat {
int dgrad_s2_wprep_jobs(const float* w, void* ws, int Cin, int Cout, int KH, int KW, int pad, WPrepJob* out) {
    S2Class cls[4];
    const int n = dgrad_s2_classes(Cin, Cout, KH, KW, pad, cls);
    for (int k = 0; k < n; ++k)
        out[k] = wprep_job(w, (char*)ws + cls[k].ws_off, Cin, Cout, 1, KH, KW, cls[k].KHc * cls[k].KWc, cls[k].KWc,
                           cls[k].kh0, cls[k].kw0, 2);
    return n;
}
}  // namespace scat

extern "C" int scat_conv2d_dgrad_s2(const float* dy, const float* w, float* dx, int B, int Cin, int H, int W,
                                    int Cout, int KH, int KW, int pad, int accumulate, void* ws, int64_t ws_bytes,
                                    int w_ready, void* stream) {
    int OH, OW;
    if (int e = check_geom("scat_conv2d_dgrad_s2", B, Cin, H, W, Cout, KH, KW, 2, pad, &OH, &OW)) return e;
    SCAT_REQUIRE(dy && w && dx, SCAT_E_ARG, "scat_conv2d_dgrad_s2: null pointer");
    SCAT_REQUIRE((KH == 1 && pad == 0) || (KH == 3 && pad == 1), SCAT_E_SHAPE,
                 "scat_conv2d_dgrad_s2: only 1x1/pad0 and 3x3/pad1 (use scat_conv2d_dgrad otherwise)");
    SCAT_REQUIRE((Cout * KH * KW) % 4 == 0 && Cout % 4 == 0, SCAT_E_SHAPE, "scat_conv2d_dgrad_s2: Cout % 4 != 0");
    SCAT_REQUIRE(ws && ws_bytes >= scat_conv2d_dgrad_s2_ws(Cin, Cout, KH, KW), SCAT_E_WORKSPACE,
                 "scat_conv2d_dgrad_s2: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (!accumulate && KH == 1) {   // 1x1: only even pixels receive gradient, the rest of dx is zero
        if (hipMemsetAsync(dx, 0, (size_t)B * Cin * H * W * sizeof(float), st) != hipSuccess) {
            set_error("scat_conv2d_dgrad_s2: memset failed");
            return SCAT_E_LAUNCH;
        }
    }
    float* wtc = (float*)ws;
    const bool split_ok = math_mode() == 1 && Cout % 16 == 0 && ((uintptr_t)ws & 15) == 0;
    SCAT_REQUIRE(!w_ready || split_ok, SCAT_E_ARG, "scat_conv2d_dgrad_s2: prepared weights exist for split products only");
    S2Class cls[4];
    const int ncls = dgrad_s2_classes(Cin, Cout, KH, KW, pad, cls);
    for (int k = 0; k < ncls; ++k) {
        {
            const int py = cls[k].py, px = cls[k].px, kh0 = cls[k].kh0, kw0 = cls[k].kw0;
            const int KHc = cls[k].KHc, KWc = cls[k].KWc;
            const int QH = (H - py + 1) / 2, QW = (W - px + 1) / 2;         // input pixels in this class
            if (QH <= 0 || QW <= 0) continue;
            const int KKc = KHc * KWc, K = Cout * KKc, N = B * QH * QW;
            if (split_ok) {
                // split-operand taps kernel: class pixel (qy, qx), tap (kky, kkx) reads dy(qy + (py+pad)/2 - kky, ...)
                TapsGeom g{};
                g.H = OH; g.W = OW; g.OH = QH; g.OW = QW; g.a = 1; g.tb = -1;
                g.c0y = (py + pad) >> 1; g.c0x = (px + pad) >> 1;
                g.KHt = KHc; g.KWt = KWc; g.KH = KH; g.KW = KW; g.kh0 = kh0; g.kw0 = kw0; g.ts = 2; g.transposed = 1;
                OutDesc dc{};
                dc.p = dx; dc.mode = 2; dc.I = Cin; dc.J = N; dc.C = Cin; dc.HW = H * W; dc.W = W; dc.QW = QW;
                dc.sub_s = 2; dc.sub_y = py; dc.sub_x = px; dc.dQHW = FastDiv::make(QH * QW);
                dc.dQW = FastDiv::make(QW); dc.accumulate = accumulate; dc.n = (int64_t)B * Cin * H * W;
                char label[40];
                snprintf(label, sizeof label, "dgrad_s2_class%dx%d", KHc, KWc);
                taps_split_launch(g, dy, w, dc, B, Cout, Cin, nullptr, nullptr, 0, (char*)ws + cls[k].ws_off, label, st,
                                  w_ready != 0);
                continue;
            }
            int64_t nw = (int64_t)Cin * K;
            int blocks = (int)((nw + 255) / 256 < 2048 ? (nw + 255) / 256 : 2048);
            hipLaunchKernelGGL(wt_class_kernel, dim3(blocks), dim3(256), 0, st, w, wtc, Cout, Cin, KH, KW, kh0, kw0,
                               KHc, KWc);
            MatDesc da{wtc, K, 1, 0, Cin, K, nw};
            // oy = q + floor((py+pad)/2) - kky  over the class grid [QH][QW]
            GatherDesc db{dy, nullptr, nullptr, 0, Cout, OH, OW, QH, QW, 1, -1, (py + pad) >> 1, (px + pad) >> 1, N, K,
                          FastDiv::make(QH * QW), FastDiv::make(QW), (int64_t)B * Cout * OH * OW};
            OutDesc dc{};
            dc.p = dx; dc.mode = 2; dc.I = Cin; dc.J = N; dc.C = Cin; dc.HW = H * W; dc.W = W; dc.QW = QW;
            dc.sub_s = 2; dc.sub_y = py; dc.sub_x = px; dc.dQHW = FastDiv::make(QH * QW); dc.dQW = FastDiv::make(QW);
            dc.accumulate = accumulate; dc.n = (int64_t)B * Cin * H * W;
            // class grid == dy grid (even H, W): flat pixel index of the class maps 1:1 onto dy's plane
            const bool bv4 = QH == OH && QW == OW && (OH * OW) % 4 == 0 && OW >= 4 && ((uintptr_t)dy & 15) == 0;
            if (KHc == 1 && KWc == 1) dgrad_class_gemm<1, 1>(da, db, dc, Cin, N, K, bv4, st);
            else if (KHc == 1) dgrad_class_gemm<1, 2>(da, db, dc, Cin, N, K, bv4, st);
            else if (KWc == 1) dgrad_class_gemm<2, 1>(da, db, dc, Cin, N, K, bv4, st);
            else dgrad_class_gemm<2, 2>(da, db, dc, Cin, N, K, bv4, st);
        }
    }
    SCAT_LAUNCH_CHECK("scat_conv2d_dgrad_s2");
    return SCAT_OK;
}

// ---------------------------------------------------------------- prepared weights
//
// Every split-operand convolution re-lays its weights (three bf16 planes, MFMA operand order) before its main
// kernel: one small launch per convolution and direction, 114 per ResNet-50 step.  A caller that keeps one
// persistent workspace per (weight, direction) can instead describe all of them once (scat_wprep_jobs, host side),
// upload the table, re-lay the whole network with ONE launch after each weight update (scat_wprep_run) and pass
// w_ready = 1 to the convolution entry points.  The library keeps no state: table and workspaces are the caller's.
extern "C" int64_t scat_wprep_job_bytes(void) { return (int64_t)sizeof(WPrepJob); }

extern "C" int64_t scat_wprep_jobs(int kind, const float* w, void* ws, int64_t ws_bytes, int Cout, int Cin, int KH,
                                   int KW, int pad, int64_t blk0, void* jobs_out, int max_jobs, int* njobs_out) {
    if (!w || !ws || !jobs_out || !njobs_out || Cout <= 0 || Cin <= 0 || max_jobs < 4 || ((uintptr_t)ws & 15)) {
        set_error("scat_wprep_jobs: bad argument");
        return SCAT_E_ARG;
    }
    WPrepJob jobs[4];
    int n = 1;
    int64_t need = 0;
    switch (kind) {
        case SCAT_WPREP_CONV1X1_FWD:
            jobs[0] = wprep_job(w, ws, Cout, Cin, 0, 1, 1, 1, 1, 0, 0, 1);
            need = taps_split_ws(Cout, Cin, 1);
            break;
        case SCAT_WPREP_CONV1X1_DGRAD:
            jobs[0] = wprep_job(w, ws, Cin, Cout, 1, 1, 1, 1, 1, 0, 0, 1);
            need = taps_split_ws(Cin, Cout, 1);
            break;
        case SCAT_WPREP_CONV3X3_FWD:
            jobs[0] = wprep_job(w, ws, Cout, Cin, 0, 3, 3, 9, 3, 0, 0, 1);
            need = taps_split_ws(Cout, Cin, 9);
            break;
        case SCAT_WPREP_CONV3X3_DGRAD:
            jobs[0] = wprep_job(w, ws, Cin, Cout, 1, 3, 3, 9, 3, 0, 0, 1);
            need = taps_split_ws(Cin, Cout, 9);
            break;
        case SCAT_WPREP_FWD_SPLIT:
            jobs[0] = wprep_job(w, ws, Cout, Cin, 0, KH, KW, KH * KW, KW, 0, 0, 1);
            need = taps_split_ws(Cout, Cin, KH * KW);
            break;
        case SCAT_WPREP_DGRAD_S2:
            n = dgrad_s2_wprep_jobs(w, ws, Cin, Cout, KH, KW, pad, jobs);
            need = taps_split_ws(Cin, Cout, KH * KW);
            break;
        default:
            set_error("scat_wprep_jobs: unknown kind %d", kind);
            return SCAT_E_ARG;
    }
    if (ws_bytes < need) {
        set_error("scat_wprep_jobs: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)need);
        return SCAT_E_WORKSPACE;
    }
    int64_t b = blk0;
    for (int k = 0; k < n; ++k) {
        if (b + jobs[k].nblk > 0x7fffffffll) {
            set_error("scat_wprep_jobs: table exceeds 2^31 blocks");
            return SCAT_E_SHAPE;
        }
        jobs[k].blk0 = (int)b;
        b += jobs[k].nblk;
    }
    memcpy(jobs_out, jobs, sizeof(WPrepJob) * n);
    *njobs_out = n;
    return b;      // first free block after these jobs
}

extern "C" int scat_wprep_run(const void* jobs_dev, int njobs, int64_t nblocks, void* stream) {
    SCAT_REQUIRE(jobs_dev && njobs > 0 && nblocks > 0 && nblocks <= 0x7fffffffll, SCAT_E_ARG,
                 "scat_wprep_run: bad argument");
    wprep_batch_launch((const WPrepJob*)jobs_dev, njobs, (int)nblocks, (hipStream_t)stream);
    SCAT_LAUNCH_CHECK("scat_wprep_run");
    return SCAT_OK;
}

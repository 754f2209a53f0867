// Convolution weight-gradient: dW[Cout][Cin*KH*KW] = dY[Cout][pixels] * im2col(X)[pixels][Cin*KH*KW],
// contraction over B*OH*OW pixels, deterministic two-stage split-K (slabs + fixed-order sum).
#include <vector>

#include "conv_common.h"

namespace scat {

// out[e] (+)= sum_z slab[z][e]: 8 independent loads in flight per thread, combined in a fixed order
__global__ void splitk_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out, int64_t n, int splits,
                                     int accumulate) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        float s = accumulate ? out[e] : 0.f;
        int z = 0;
        for (; z + 8 <= splits; z += 8) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = slab[(int64_t)(z + q) * n + e];
            s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        for (; z < splits; ++z) s += slab[(int64_t)z * n + e];
        out[e] = s;
    }
}

// The same sum for n % 4 == 0, 16 bytes per lane.  A workgroup owns 256 consecutive elements; its WAVES wavefronts
// take consecutive ranges of the slabs (a small output with hundreds of slabs — layer1's 64x64 — would otherwise be a
// few wavefronts walking the whole stack at one memory latency per 8 slabs) and their partial sums are combined
// through LDS in wavefront order: the result depends on (splits, WAVES) only, never on timing.
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void splitk_reduce4_kernel(const float4* __restrict__ slab,
                                                                    float4* __restrict__ out, int64_t n4, int splits,
                                                                    int accumulate) {
    __shared__ float4 part[WAVES > 1 ? WAVES : 1][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t e = blockIdx.x * 64ll + lane;
    const bool live = e < n4;
    const int per = (splits + WAVES - 1) / WAVES;
    const int z0 = wave * per, z1 = min(z0 + per, splits);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        int z = z0;
        for (; z + 8 <= z1; z += 8) {
            float4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = slab[(int64_t)(z + q) * n4 + e];
            s.x += ((v[0].x + v[1].x) + (v[2].x + v[3].x)) + ((v[4].x + v[5].x) + (v[6].x + v[7].x));
            s.y += ((v[0].y + v[1].y) + (v[2].y + v[3].y)) + ((v[4].y + v[5].y) + (v[6].y + v[7].y));
            s.z += ((v[0].z + v[1].z) + (v[2].z + v[3].z)) + ((v[4].z + v[5].z) + (v[6].z + v[7].z));
            s.w += ((v[0].w + v[1].w) + (v[2].w + v[3].w)) + ((v[4].w + v[5].w) + (v[6].w + v[7].w));
        }
        for (; z < z1; ++z) {
            const float4 v = slab[(int64_t)z * n4 + e];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    if constexpr (WAVES > 1) {
        part[wave][lane] = s;
        __syncthreads();
        if (wave != 0) return;
        s = part[0][lane];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) {
            const float4 v = part[w][lane];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    if (!live) return;
    if (accumulate) {
        const float4 o = out[e];
        s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
    }
    out[e] = s;
}

// ---- deferred reduces: ONE grouped launch for the fixed-order sums of many weight gradients
// A ResNet-50 backward issues 54 of the launches above, 6-12 us each, every one behind its contraction on the weight-
// gradient stream.  With scat_splitk_defer(1) in force (per host thread) a reduce is not launched but recorded; the caller
// keeps every recorded slab intact (its own workspace per contraction) until scat_splitk_reduce_flush() sums them all
// with one launch: a workgroup finds its job in a table passed in the kernel arguments (<= 48 jobs per launch) and runs
// the four-wavefront body above — the result depends on (splits, 4) only, bit-reproducible like the single launches.
struct ReduceJob {
    const float4* slab;
    float4* out;
    int64_t n4;
    int splits, accumulate, blk0, pad;
};
constexpr int RG_MAX = 48;
struct ReduceTable {
    ReduceJob j[RG_MAX];
    int njobs;
};

__global__ __launch_bounds__(256) void splitk_reduce_group_kernel(ReduceTable t) {
    __shared__ float4 part[4][64];
    int lo = 0, hi = t.njobs - 1;
    const int b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (t.j[mid].blk0 <= b) lo = mid;
        else hi = mid - 1;
    }
    const float4* __restrict__ slab = t.j[lo].slab;
    float4* __restrict__ out = t.j[lo].out;
    const int64_t n4 = t.j[lo].n4;
    const int splits = t.j[lo].splits, accumulate = t.j[lo].accumulate;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t e = (int64_t)(b - t.j[lo].blk0) * 64 + lane;
    const bool live = e < n4;
    const int per = (splits + 3) / 4;
    const int z0 = wave * per, z1 = min(z0 + per, splits);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        int z = z0;
        for (; z + 8 <= z1; z += 8) {
            float4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = slab[(int64_t)(z + q) * n4 + e];
            s.x += ((v[0].x + v[1].x) + (v[2].x + v[3].x)) + ((v[4].x + v[5].x) + (v[6].x + v[7].x));
            s.y += ((v[0].y + v[1].y) + (v[2].y + v[3].y)) + ((v[4].y + v[5].y) + (v[6].y + v[7].y));
            s.z += ((v[0].z + v[1].z) + (v[2].z + v[3].z)) + ((v[4].z + v[5].z) + (v[6].z + v[7].z));
            s.w += ((v[0].w + v[1].w) + (v[2].w + v[3].w)) + ((v[4].w + v[5].w) + (v[6].w + v[7].w));
        }
        for (; z < z1; ++z) {
            const float4 v = slab[(int64_t)z * n4 + e];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave != 0 || !live) return;
    s = part[0][lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        const float4 v = part[w][lane];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (accumulate) {
        const float4 o = out[e];
        s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
    }
    out[e] = s;
}

struct ReduceDefer {
    bool on = false;
    std::vector<ReduceJob> jobs;
};
static thread_local ReduceDefer g_rdefer;

void launch_splitk_reduce(const float* slab, float* out, int64_t n, int splits, int accumulate, hipStream_t st) {
    static const int vec = diag_env_int("SCAT_REDUCE_VEC", 1);
    if (g_rdefer.on && n % 4 == 0 && (((uintptr_t)slab | (uintptr_t)out) & 15) == 0 && n / 4 < (1ll << 30)) {
        g_rdefer.jobs.push_back(ReduceJob{(const float4*)slab, (float4*)out, n / 4, splits, accumulate, 0, 0});
        append_kernel_label("_rdefer");
        return;
    }
    if (vec && n % 4 == 0 && (((uintptr_t)slab | (uintptr_t)out) & 15) == 0) {
        const int64_t n4 = n / 4;
        const dim3 grid((unsigned)((n4 + 63) / 64));
        const float4* s4 = (const float4*)slab;
        float4* o4 = (float4*)out;
        // enough wavefronts to cover the chip (~2048) before giving a wavefront more than 8 slabs
        const int64_t waves1 = (n4 + 63) / 64;
        if (splits >= 64 && waves1 < 512)
            hipLaunchKernelGGL(splitk_reduce4_kernel<16>, grid, dim3(1024), 0, st, s4, o4, n4, splits, accumulate);
        else if (splits >= 16 && waves1 < 2048)
            hipLaunchKernelGGL(splitk_reduce4_kernel<4>, grid, dim3(256), 0, st, s4, o4, n4, splits, accumulate);
        else
            hipLaunchKernelGGL(splitk_reduce4_kernel<1>, grid, dim3(64), 0, st, s4, o4, n4, splits, accumulate);
        return;
    }
    const int blocks = (int)((n + 63) / 64 < 4096 ? (n + 63) / 64 : 4096);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(64), 0, st, slab, out, n, splits, accumulate);
}

static int wgrad_splits(int M, int N, int K, int bm, int bn) {
    int tiles = cdiv(M, bm) * cdiv(N, bn);
    static const int target = diag_env_int("SCAT_WG_F32_TARGET", 1024);
    int s = cdiv(target, tiles);                     // aim for ~4 workgroups per CU
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

template <int KH, int KW, bool AV4, bool BV4, bool TF>
static void wgrad_gemm_tf(const WgradPlan& p, const GatherDesc& da, const GatherDesc& db, const OutDesc& dc,
                          hipStream_t st) {
    set_kernel_label("wgrad%dx%d_%dx64x32%s%s%s_split%d", KH, KW, p.bm, AV4 ? "_a4" : "", BV4 ? "_b4" : "",
                     TF ? "_tf" : "", p.splits);
    if (p.bm == 64)
        launch_gemm<GatherLoader<64, 32, 1, 1, false, true, AV4, false>,
                    GatherLoader<64, 32, KH, KW, false, true, BV4, TF>, 64, 64, 32, 2, 2>(da, db, dc, p.M, p.N, p.K,
                                                                                          p.splits, st);
    else
        launch_gemm<GatherLoader<128, 32, 1, 1, false, true, AV4, false>,
                    GatherLoader<64, 32, KH, KW, false, true, BV4, TF>, 128, 64, 32, 2, 2>(da, db, dc, p.M, p.N, p.K,
                                                                                           p.splits, st);
}

template <int KH, int KW, bool AV4, bool BV4>
static void wgrad_gemm(const WgradPlan& p, const GatherDesc& da, const GatherDesc& db, const OutDesc& dc,
                       hipStream_t st) {
    if constexpr (KH != 7) {
        if (db.scale) {
            wgrad_gemm_tf<KH, KW, AV4, BV4, true>(p, da, db, dc, st);
            return;
        }
    }
    wgrad_gemm_tf<KH, KW, AV4, BV4, false>(p, da, db, dc, st);
}

// measured at batch 96: the split kernel wins or ties everywhere it applies (tiny outputs with huge contractions —
// ResNet layer1's 64x64 1x1, HRNet's 32-channel 3x3 — are staging-bound on both engines; SCAT_WG_MINMN sets a floor
// on Cout*Cin*k*k below which the fp32 engine is used instead)
static bool wgrad_split_ok(int KH, int stride, int pad, int Cout, int Cin) {
    // (1x1/stride 2: the strided 8-dword gather makes the split kernel staging-bound, 315-380 us vs 240-250 us)
    static const int64_t minmn = diag_env_int("SCAT_WG_MINMN", 0ll);
    return ((KH == 1 && pad == 0 && stride == 1) || (KH == 3 && pad == 1)) && (int64_t)Cout * Cin * KH * KH > minmn;
}

}  // namespace scat

using namespace scat;

extern "C" int64_t scat_conv2d_wgrad_ws(int B, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad) {
    int OH, OW;
    if (check_geom("scat_conv2d_wgrad_ws", B, Cin, H, W, Cout, KH, KW, stride, pad, &OH, &OW)) return -1;
    WgradPlan p = wgrad_plan(B, Cin, Cout, KH * KW, OH, OW);
    int64_t need = p.splits > 1 ? (int64_t)p.splits * p.M * p.N * sizeof(float) : 0;
    if (wgrad_split_ok(KH, stride, pad, Cout, Cin)) {            // the math mode may change between this query and the call
        WgSplitPlan q = wgrad_split_plan(B, Cin, Cout, KH * KW, OH * OW);
        int64_t n2 = q.splits > 1 ? (int64_t)q.splits * q.M * q.N * sizeof(float) : 0;
        if (n2 > need) need = n2;
    }
    if (wgrad_rows_ok(B, Cin, H, W, Cout, KH, stride, pad, nullptr, nullptr)) {
        const int64_t n3 = wgrad_rows_ws(B, Cin, H, W, Cout);
        if (n3 > need) need = n3;
    }
    if (KH == 1 && stride == 1 && pad == 0) {
        const WgPwPlan w = wgrad_pw_plan(B, Cin, Cout, OH * OW, false, nullptr, nullptr);
        const int64_t n4 = w.ok && w.splits > 1 ? (int64_t)w.splits * Cout * Cin * sizeof(float) : 0;
        if (n4 > need) need = n4;
    }
    return need;
}

// Weight gradient of a 1x1/stride-1 convolution from a BatchNorm backward that was never materialised:
// dy = ca*g + cb*z + cc per output channel (see scat_bn_bwd_pre / scat_conv1x1_s1_bnb).  Split products only.
extern "C" int scat_conv1x1_wgrad_bnb(const float* g, const float* z, const float* coef3, const float* x, float* dw,
                                      int B, int Cin, int HW, int Cout, const float* in_scale, const float* in_shift,
                                      int in_relu, void* ws, int64_t ws_bytes, void* stream) {
    SCAT_REQUIRE(g && z && coef3 && x && dw, SCAT_E_ARG, "scat_conv1x1_wgrad_bnb: null pointer");
    SCAT_REQUIRE(math_mode() == 1, SCAT_E_ARG, "scat_conv1x1_wgrad_bnb: needs the split-operand product mode");
    SCAT_REQUIRE(B > 0 && Cin > 0 && HW > 0 && Cout > 0, SCAT_E_SHAPE, "scat_conv1x1_wgrad_bnb: non-positive dimension");
    SCAT_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), SCAT_E_ARG, "scat_conv1x1_wgrad_bnb: scale/shift pair");
    SCAT_REQUIRE(fits_i32((int64_t)B * Cout * HW * 4) && fits_i32((int64_t)B * Cin * HW * 4), SCAT_E_SHAPE,
                 "scat_conv1x1_wgrad_bnb: tensor exceeds 32-bit byte offsets");
    hipStream_t st = (hipStream_t)stream;
    const WgPwPlan w = wgrad_pw_plan(B, Cin, Cout, HW, true, g, x);
    if (w.ok && ((uintptr_t)z & 15) == 0) {
        const int64_t needw = w.splits > 1 ? (int64_t)w.splits * Cout * Cin * sizeof(float) : 0;
        SCAT_REQUIRE(ws_bytes >= needw && (needw == 0 || ws), SCAT_E_WORKSPACE,
                     "scat_conv1x1_wgrad_bnb: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)needw);
        wgrad_pw_launch(w, g, x, w.splits > 1 ? (float*)ws : dw, B, Cin, HW, Cout, in_scale, in_shift,
                        in_scale ? in_relu : 0, st, z, coef3);
        SCAT_LAUNCH_CHECK("scat_conv1x1_wgrad_bnb(pw)");
        if (w.splits > 1) {
            launch_splitk_reduce((const float*)ws, dw, (int64_t)Cout * Cin, w.splits, 0, st);
            SCAT_LAUNCH_CHECK("scat_conv1x1_wgrad_bnb(reduce)");
        }
        return SCAT_OK;
    }
    const WgSplitPlan q = wgrad_split_plan(B, Cin, Cout, 1, HW);
    const int64_t need = q.splits > 1 ? (int64_t)q.splits * q.M * q.N * sizeof(float) : 0;
    SCAT_REQUIRE(ws_bytes >= need && (need == 0 || ws), SCAT_E_WORKSPACE,
                 "scat_conv1x1_wgrad_bnb: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)need);
    wgrad_split_launch(q, g, x, q.splits > 1 ? (float*)ws : dw, B, Cin, 1, HW, Cout, 1, 1, in_scale, in_shift,
                       in_scale ? in_relu : 0, st, z, coef3);
    SCAT_LAUNCH_CHECK("scat_conv1x1_wgrad_bnb");
    if (q.splits > 1) {
        int64_t n = (int64_t)q.M * q.N;
        launch_splitk_reduce((const float*)ws, dw, n, q.splits, 0, st);
        SCAT_LAUNCH_CHECK("scat_conv1x1_wgrad_bnb(reduce)");
    }
    return SCAT_OK;
}
extern "C" int64_t scat_conv1x1_wgrad_bnb_ws(int B, int Cin, int HW, int Cout) {
    const WgSplitPlan q = wgrad_split_plan(B, Cin, Cout, 1, HW);
    int64_t need = q.splits > 1 ? (int64_t)q.splits * q.M * q.N * sizeof(float) : 0;
    const WgPwPlan w = wgrad_pw_plan(B, Cin, Cout, HW, true, nullptr, nullptr);
    const int64_t needw = w.ok && w.splits > 1 ? (int64_t)w.splits * Cout * Cin * sizeof(float) : 0;
    return needw > need ? needw : need;
}

extern "C" int scat_conv2d_wgrad(const float* dy, const float* x, float* dw, int B, int Cin, int H, int W, int Cout,
                                 int KH, int KW, int stride, int pad, const float* in_scale, const float* in_shift,
                                 int in_relu, void* ws, int64_t ws_bytes, void* stream) {
    int OH, OW;
    if (int e = check_geom("scat_conv2d_wgrad", B, Cin, H, W, Cout, KH, KW, stride, pad, &OH, &OW)) return e;
    SCAT_REQUIRE(dy && x && dw, SCAT_E_ARG, "scat_conv2d_wgrad: null pointer");
    SCAT_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), SCAT_E_ARG, "scat_conv2d_wgrad: scale/shift pair");
    SCAT_REQUIRE(!(in_scale && KH == 7), SCAT_E_SHAPE, "scat_conv2d_wgrad: fused input transform not built for 7x7");
    if (!in_scale) in_relu = 0;
    if (math_mode() == 1 && KH == KW && wgrad_rows_ok(B, Cin, H, W, Cout, KH, stride, pad, dy, x) &&
        fits_i32((int64_t)B * Cout * H * W * 4) && fits_i32((int64_t)B * Cin * H * W * 4)) {
        const int64_t need3 = wgrad_rows_ws(B, Cin, H, W, Cout);
        SCAT_REQUIRE(ws && ws_bytes >= need3, SCAT_E_WORKSPACE, "scat_conv2d_wgrad: workspace %lld < %lld bytes",
                     (long long)ws_bytes, (long long)need3);
        hipStream_t st3 = (hipStream_t)stream;
        const int splits = wgrad_rows_launch(dy, x, (float*)ws, B, Cin, H, W, Cout, in_scale, in_shift, in_relu, st3);
        SCAT_LAUNCH_CHECK("scat_conv2d_wgrad(rows)");
        launch_splitk_reduce((const float*)ws, dw, (int64_t)Cout * Cin * 9, splits, 0, st3);
        SCAT_LAUNCH_CHECK("scat_conv2d_wgrad(reduce)");
        return SCAT_OK;
    }
    if (math_mode() == 1 && KH == 1 && KW == 1 && stride == 1 && pad == 0 &&
        fits_i32((int64_t)B * Cout * H * W * 4) && fits_i32((int64_t)B * Cin * H * W * 4)) {
        const WgPwPlan w = wgrad_pw_plan(B, Cin, Cout, H * W, false, dy, x);
        if (w.ok) {
            const int64_t need4 = w.splits > 1 ? (int64_t)w.splits * Cout * Cin * sizeof(float) : 0;
            SCAT_REQUIRE(ws_bytes >= need4 && (need4 == 0 || ws), SCAT_E_WORKSPACE,
                         "scat_conv2d_wgrad: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)need4);
            hipStream_t st4 = (hipStream_t)stream;
            wgrad_pw_launch(w, dy, x, w.splits > 1 ? (float*)ws : dw, B, Cin, H * W, Cout, in_scale, in_shift, in_relu, st4);
            SCAT_LAUNCH_CHECK("scat_conv2d_wgrad(pw)");
            if (w.splits > 1) {
                launch_splitk_reduce((const float*)ws, dw, (int64_t)Cout * Cin, w.splits, 0, st4);
                SCAT_LAUNCH_CHECK("scat_conv2d_wgrad(reduce)");
            }
            return SCAT_OK;
        }
    }
    if (math_mode() == 1 && wgrad_split_ok(KH, stride, pad, Cout, Cin)) {
        const WgSplitPlan q = wgrad_split_plan(B, Cin, Cout, KH * KW, OH * OW);
        const int64_t need2 = q.splits > 1 ? (int64_t)q.splits * q.M * q.N * sizeof(float) : 0;
        SCAT_REQUIRE(ws_bytes >= need2 && (need2 == 0 || ws), SCAT_E_WORKSPACE,
                     "scat_conv2d_wgrad: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)need2);
        hipStream_t st2 = (hipStream_t)stream;
        wgrad_split_launch(q, dy, x, q.splits > 1 ? (float*)ws : dw, B, Cin, H, W, Cout, KH * KW, stride, in_scale,
                           in_shift, in_relu, st2);
        SCAT_LAUNCH_CHECK("scat_conv2d_wgrad");
        if (q.splits > 1) {
            int64_t n = (int64_t)q.M * q.N;
            launch_splitk_reduce((const float*)ws, dw, n, q.splits, 0, st2);
            SCAT_LAUNCH_CHECK("scat_conv2d_wgrad(reduce)");
        }
        return SCAT_OK;
    }
    WgradPlan p = wgrad_plan(B, Cin, Cout, KH * KW, OH, OW);
    int64_t need = p.splits > 1 ? (int64_t)p.splits * p.M * p.N * sizeof(float) : 0;
    SCAT_REQUIRE(ws_bytes >= need && (need == 0 || ws), SCAT_E_WORKSPACE,
                 "scat_conv2d_wgrad: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)need);
    const int npix = B * OH * OW;
    // A: dy as [Cout][pixel]; B: x through the forward conv arithmetic as [pixel][(ci,kh,kw)]
    GatherDesc da{dy, nullptr, nullptr, 0, Cout, OH, OW, OH, OW, 1, 0, 0, 0, npix, Cout, FastDiv::make(OH * OW),
                  FastDiv::make(OW), (int64_t)B * Cout * OH * OW};
    GatherDesc db{x, in_scale, in_shift, in_relu, Cin, H, W, OH, OW, stride, 1, -pad, -pad, npix, Cin * KH * KW,
                  FastDiv::make(OH * OW), FastDiv::make(OW), (int64_t)B * Cin * H * W};
    OutDesc dc{};
    dc.p = p.splits > 1 ? (float*)ws : dw;
    dc.mode = 0; dc.si = p.N; dc.sj = 1; dc.sz = (int64_t)p.M * p.N; dc.I = p.M; dc.J = p.N; dc.n = (int64_t)p.M * p.N;
    hipStream_t st = (hipStream_t)stream;
    const bool av4 = (OH * OW) % 4 == 0 && ((uintptr_t)dy & 15) == 0;
    const bool bv4 = KH <= 3 && stride == 1 && OH == H && OW == W && (H * W) % 4 == 0 && W >= 4 &&
                     ((uintptr_t)x & 15) == 0;
    if (KH == 1) {
        if (av4 && bv4) wgrad_gemm<1, 1, true, true>(p, da, db, dc, st);
        else if (av4) wgrad_gemm<1, 1, true, false>(p, da, db, dc, st);
        else wgrad_gemm<1, 1, false, false>(p, da, db, dc, st);
    } else if (KH == 3) {
        if (av4 && bv4) wgrad_gemm<3, 3, true, true>(p, da, db, dc, st);
        else if (av4) wgrad_gemm<3, 3, true, false>(p, da, db, dc, st);
        else wgrad_gemm<3, 3, false, false>(p, da, db, dc, st);
    } else {
        if (av4) wgrad_gemm<7, 7, true, false>(p, da, db, dc, st);
        else wgrad_gemm<7, 7, false, false>(p, da, db, dc, st);
    }
    SCAT_LAUNCH_CHECK("scat_conv2d_wgrad");
    if (p.splits > 1) {
        int64_t n = (int64_t)p.M * p.N;
        launch_splitk_reduce((const float*)ws, dw, n, p.splits, 0, st);
        SCAT_LAUNCH_CHECK("scat_conv2d_wgrad(reduce)");
    }
    return SCAT_OK;
}


/* Deferred split-K reduces (include/scat_hip.h). */
extern "C" int scat_splitk_defer(int on) {
    scat::g_rdefer.on = on != 0;
    return SCAT_OK;
}
extern "C" int scat_splitk_reduce_pending(void) { return (int)scat::g_rdefer.jobs.size(); }
extern "C" int scat_splitk_reduce_discard(void) {
    scat::g_rdefer.jobs.clear();
    return SCAT_OK;
}
extern "C" int scat_splitk_reduce_flush(void* stream) {
    auto& jobs = scat::g_rdefer.jobs;
    hipStream_t st = (hipStream_t)stream;
    for (size_t i = 0; i < jobs.size(); i += scat::RG_MAX) {
        scat::ReduceTable t{};
        int blk = 0;
        t.njobs = (int)(jobs.size() - i < (size_t)scat::RG_MAX ? jobs.size() - i : (size_t)scat::RG_MAX);
        for (int k = 0; k < t.njobs; ++k) {
            t.j[k] = jobs[i + k];
            t.j[k].blk0 = blk;
            blk += (int)((t.j[k].n4 + 63) / 64);
        }
        hipLaunchKernelGGL(scat::splitk_reduce_group_kernel, dim3(blk), dim3(256), 0, st, t);
    }
    jobs.clear();
    SCAT_LAUNCH_CHECK("scat_splitk_reduce_flush");
    return SCAT_OK;
}

// FAVOR+ (performer) linear attention core: models/vision_performer.py:34-53 of the reference.
//   kp = exp(k·wᵀ − |k|²/2)/√m,  qp likewise;  D = qp·Σ_t kp;  kptv = vᵀ·kp;  y = (qp·kptvᵀ)/D
// All heads of a block share one kqv Linear (vision_performer.py:17) and one frozen w[m,e] (:32); the
// reference loops heads in Python (:59-60) — here every (batch, head) pair is a workgroup of one launch.
// Work is O(T·m·e) per head (2.4e5 MAC at T=21,e=49,m=24): latency-bound, so the kernels are plain
// thread-per-output loops with operands served from L1/L2; no n×n score matrix exists.
#include "common.h"

namespace scat {

// zp[b,h,t,i] for z = k (slot 0) and q (slot 1); one wave per (b,t,h) row
__global__ __launch_bounds__(256) void prm_exp_kernel(const float* __restrict__ kqv, const float* __restrict__ w,
                                                      float* __restrict__ kp, float* __restrict__ qp, int B, int T,
                                                      int H, int e, int m) {
    const int lane = threadIdx.x & 63;
    const int64_t row = blockIdx.x * 4ll + (threadIdx.x >> 6);   // (b,t,h)
    if (row >= (int64_t)B * T * H) return;
    const int h = row % H;
    const int64_t bt = row / H;
    const int t = bt % T, b = bt / T;
    const float rs = rsqrtf((float)m);
    for (int slot = 0; slot < 2; ++slot) {
        const float* z = kqv + row * 3 * e + slot * e;
        float n2 = 0.f;
        for (int d = lane; d < e; d += 64) n2 += z[d] * z[d];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) n2 += __shfl_xor(n2, o, 64);
        float* out = (slot == 0 ? kp : qp) + (((int64_t)b * H + h) * T + t) * m;
        for (int i = lane; i < m; i += 64) {
            float s = 0.f;
            for (int d = 0; d < e; ++d) s = fmaf(z[d], w[i * e + d], s);
            out[i] = expf(s - 0.5f * n2) * rs;
        }
    }
}

// per (b,h): ksum[i] = Σ_t kp[t][i];  kptv[n][i] = Σ_t v[t][n] kp[t][i]
__global__ __launch_bounds__(256) void kptv_kernel(const float* __restrict__ kqv, const float* __restrict__ kp,
                                                   float* __restrict__ kptv, float* __restrict__ ksum, int T, int H,
                                                   int e, int m) {
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const float* kpb = kp + (int64_t)blockIdx.x * T * m;
    for (int p = threadIdx.x; p < e * m + m; p += 256) {
        float s = 0.f;
        if (p < e * m) {
            const int n = p / m, i = p - n * m;
            for (int t = 0; t < T; ++t)
                s = fmaf(kqv[(((int64_t)b * T + t) * H + h) * 3 * e + 2 * e + n], kpb[t * m + i], s);
            kptv[(int64_t)blockIdx.x * e * m + p] = s;
        } else {
            const int i = p - e * m;
            for (int t = 0; t < T; ++t) s += kpb[t * m + i];
            ksum[(int64_t)blockIdx.x * m + i] = s;
        }
    }
}

// y[b,t,h*e+n] = Σ_i qp[t][i] kptv[n][i] / D[t],  D[t] = Σ_i qp[t][i] ksum[i]; one wave per (b,h,t)
__global__ __launch_bounds__(256) void performer_out_kernel(const float* __restrict__ qp,
                                                            const float* __restrict__ kptv,
                                                            const float* __restrict__ ksum, float* __restrict__ y,
                                                            float* __restrict__ Dout, int B, int T, int H, int e,
                                                            int m) {
    const int lane = threadIdx.x & 63;
    const int64_t row = blockIdx.x * 4ll + (threadIdx.x >> 6);   // (b,h,t)
    if (row >= (int64_t)B * H * T) return;
    const int t = row % T;
    const int64_t bh = row / T;
    const int h = bh % H, b = bh / H;
    const float* q = qp + row * m;
    float d = 0.f;
    for (int i = lane; i < m; i += 64) d += q[i] * ksum[bh * m + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
    if (lane == 0) Dout[row] = d;
    for (int n = lane; n < e; n += 64) {
        float s = 0.f;
        for (int i = 0; i < m; ++i) s = fmaf(q[i], kptv[(bh * e + n) * m + i], s);
        y[(((int64_t)b * T + t) * H + h) * e + n] = s / d;
    }
}

// backward 1: per (b,h,t): dnum = dy/D, dD = -Σ_n dy·y / D, dqp[i] = Σ_n dnum[n] kptv[n][i] + dD ksum[i]
__global__ __launch_bounds__(256) void performer_bwd1_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                             const float* __restrict__ D,
                                                             const float* __restrict__ kptv,
                                                             const float* __restrict__ ksum, float* __restrict__ dnum,
                                                             float* __restrict__ dD, float* __restrict__ dqp, int B,
                                                             int T, int H, int e, int m) {
    const int lane = threadIdx.x & 63;
    const int64_t row = blockIdx.x * 4ll + (threadIdx.x >> 6);   // (b,h,t)
    if (row >= (int64_t)B * H * T) return;
    const int t = row % T;
    const int64_t bh = row / T;
    const int h = bh % H, b = bh / H;
    const int64_t yo = (((int64_t)b * T + t) * H + h) * e;
    const float d = D[row];
    float acc = 0.f;
    for (int n = lane; n < e; n += 64) {
        float g = dy[yo + n];
        dnum[row * e + n] = g / d;
        acc += g * y[yo + n];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    const float dd = -acc / d;
    if (lane == 0) dD[row] = dd;
    for (int i = lane; i < m; i += 64) {
        float s = dd * ksum[bh * m + i];
        for (int n = 0; n < e; ++n) s = fmaf(dy[yo + n] / d, kptv[(bh * e + n) * m + i], s);
        dqp[row * m + i] = s;
    }
}

// backward 2: per (b,h): dkptv[n][i] = Σ_t dnum[t][n] qp[t][i];  dksum[i] = Σ_t dD[t] qp[t][i]
__global__ __launch_bounds__(256) void performer_bwd2_kernel(const float* __restrict__ dnum,
                                                             const float* __restrict__ dD,
                                                             const float* __restrict__ qp, float* __restrict__ dkptv,
                                                             float* __restrict__ dksum, int T, int e, int m) {
    const int64_t bh = blockIdx.x;
    for (int p = threadIdx.x; p < e * m + m; p += 256) {
        float s = 0.f;
        if (p < e * m) {
            const int n = p / m, i = p - n * m;
            for (int t = 0; t < T; ++t) s = fmaf(dnum[(bh * T + t) * e + n], qp[(bh * T + t) * m + i], s);
            dkptv[bh * e * m + p] = s;
        } else {
            const int i = p - e * m;
            for (int t = 0; t < T; ++t) s = fmaf(dD[bh * T + t], qp[(bh * T + t) * m + i], s);
            dksum[bh * m + i] = s;
        }
    }
}

// backward 3: one wave per (b,h,t): dv, dkp -> dk, dqp -> dq, written to dkqv[b,t,h,3e] (k|q|v)
__global__ __launch_bounds__(256) void performer_bwd3_kernel(const float* __restrict__ kqv, const float* __restrict__ w,
                                                             const float* __restrict__ kp, const float* __restrict__ qp,
                                                             const float* __restrict__ dqp,
                                                             const float* __restrict__ dkptv,
                                                             const float* __restrict__ dksum, float* __restrict__ dkqv,
                                                             int B, int T, int H, int e, int m) {
    __shared__ float sh[4][2][128];   // per wave: dkp·kp and dqp·qp (m <= 128)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t row = blockIdx.x * 4ll + wv;   // (b,h,t)
    const bool live = row < (int64_t)B * H * T;
    const int t = live ? row % T : 0;
    const int64_t bh = live ? row / T : 0;
    const int h = bh % H, b = bh / H;
    const int64_t zo = (((int64_t)b * T + t) * H + h) * 3 * e;
    if (live) {
        for (int i = lane; i < m; i += 64) {
            float s = dksum[bh * m + i];
            for (int n = 0; n < e; ++n) s = fmaf(dkptv[(bh * e + n) * m + i], kqv[zo + 2 * e + n], s);
            sh[wv][0][i] = s * kp[row * m + i];
            sh[wv][1][i] = dqp[row * m + i] * qp[row * m + i];
        }
    }
    __syncthreads();
    if (!live) return;
    float sk = 0.f, sq = 0.f;
    for (int i = 0; i < m; ++i) { sk += sh[wv][0][i]; sq += sh[wv][1][i]; }
    for (int d = lane; d < e; d += 64) {
        float ak = 0.f, aq = 0.f, av = 0.f;
        for (int i = 0; i < m; ++i) {
            ak = fmaf(sh[wv][0][i], w[i * e + d], ak);
            aq = fmaf(sh[wv][1][i], w[i * e + d], aq);
            av = fmaf(dkptv[(bh * e + d) * m + i], kp[row * m + i], av);
        }
        dkqv[zo + d] = ak - sk * kqv[zo + d];                   // Σ_i dkp·kp (w[i][d] − k[d])
        dkqv[zo + e + d] = aq - sq * kqv[zo + e + d];
        dkqv[zo + 2 * e + d] = av;
    }
}

}  // namespace scat

using namespace scat;

static int perf_check(const char* who, int B, int T, int H, int e, int m) {
    SCAT_REQUIRE(B > 0 && T > 0 && H > 0 && e > 0 && m > 0 && m <= 128, SCAT_E_SHAPE, "%s: need m <= 128", who);
    return SCAT_OK;
}

extern "C" int scat_performer_fwd(const float* kqv, const float* w, float* y, float* kp, float* qp, float* kptv,
                                  float* ksum, float* D, int B, int T, int heads, int e, int m, void* stream) {
    if (int r = perf_check("scat_performer_fwd", B, T, heads, e, m)) return r;
    SCAT_REQUIRE(kqv && w && y && kp && qp && kptv && ksum && D, SCAT_E_ARG, "scat_performer_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int64_t rows = (int64_t)B * T * heads;
    hipLaunchKernelGGL(prm_exp_kernel, dim3((int)((rows + 3) / 4)), dim3(256), 0, st, kqv, w, kp, qp, B, T, heads, e, m);
    hipLaunchKernelGGL(kptv_kernel, dim3(B * heads), dim3(256), 0, st, kqv, (const float*)kp, kptv, ksum, T, heads, e, m);
    hipLaunchKernelGGL(performer_out_kernel, dim3((int)((rows + 3) / 4)), dim3(256), 0, st, (const float*)qp,
                       (const float*)kptv, (const float*)ksum, y, D, B, T, heads, e, m);
    SCAT_LAUNCH_CHECK("scat_performer_fwd");
    return SCAT_OK;
}

extern "C" int64_t scat_performer_bwd_ws(int B, int T, int heads, int e, int m) {
    const int64_t bh = (int64_t)B * heads;
    return (bh * T * e + bh * T + bh * T * m + bh * e * m + bh * m) * (int64_t)sizeof(float);
}

extern "C" int scat_performer_bwd(const float* dy, const float* kqv, const float* w, const float* y, const float* kp,
                                  const float* qp, const float* kptv, const float* ksum, const float* D, float* dkqv,
                                  int B, int T, int heads, int e, int m, void* ws, int64_t ws_bytes, void* stream) {
    if (int r = perf_check("scat_performer_bwd", B, T, heads, e, m)) return r;
    SCAT_REQUIRE(dy && kqv && w && y && kp && qp && kptv && ksum && D && dkqv, SCAT_E_ARG,
                 "scat_performer_bwd: null pointer");
    SCAT_REQUIRE(ws && ws_bytes >= scat_performer_bwd_ws(B, T, heads, e, m), SCAT_E_WORKSPACE,
                 "scat_performer_bwd: workspace too small");
    const int64_t bh = (int64_t)B * heads, rows = bh * T;
    float* dnum = (float*)ws;
    float* dD = dnum + bh * T * e;
    float* dqp = dD + bh * T;
    float* dkptv = dqp + bh * T * m;
    float* dksum = dkptv + bh * e * m;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(performer_bwd1_kernel, dim3((int)((rows + 3) / 4)), dim3(256), 0, st, dy, y, D, kptv, ksum, dnum,
                       dD, dqp, B, T, heads, e, m);
    hipLaunchKernelGGL(performer_bwd2_kernel, dim3((int)bh), dim3(256), 0, st, (const float*)dnum, (const float*)dD, qp,
                       dkptv, dksum, T, e, m);
    hipLaunchKernelGGL(performer_bwd3_kernel, dim3((int)((rows + 3) / 4)), dim3(256), 0, st, kqv, w, kp, qp,
                       (const float*)dqp, (const float*)dkptv, (const float*)dksum, dkqv, B, T, heads, e, m);
    SCAT_LAUNCH_CHECK("scat_performer_bwd");
    return SCAT_OK;
}

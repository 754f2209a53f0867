// Element-wise passes, token glue, the iterative regressor head, the train.py loss and Adam.
// All HBM-bound (or tiny): grid-stride, 16-B accesses where alignment allows.
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace scat {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static thread_local char g_label[128] = "";

void set_kernel_label(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_label, sizeof(g_label), fmt, ap);
    va_end(ap);
}

void append_kernel_label(const char* suffix) {
    const size_t n = strlen(g_label);
    if (n + strlen(suffix) + 1 <= sizeof(g_label)) strcpy(g_label + n, suffix);
}

struct EpiStats { float* buf; int64_t cap; int groups; const float* shift; };
static thread_local EpiStats g_epi = {nullptr, 0, 0, nullptr};
float* epi_stats_take(int rows, int groups, const float** shift) {
    float* b = g_epi.buf;
    g_epi.buf = nullptr;
    if (shift) *shift = nullptr;
    if (!b || (int64_t)rows * groups * 2 > g_epi.cap) return nullptr;
    g_epi.groups = groups;
    if (shift) *shift = g_epi.shift;
    return b;
}

static thread_local EpiBnb g_bnb = {nullptr, nullptr, nullptr, nullptr, 0, 0, 0};
bool epi_bnb_take(int rows, int groups, int64_t n, EpiBnb* out) {
    const EpiBnb b = g_bnb;
    g_bnb.part = nullptr;
    g_bnb.groups = 0;
    if (!b.part || !b.x || !b.mask || !b.mean || b.n != n || (int64_t)rows * groups * 2 > b.cap) return false;
    g_bnb.groups = groups;
    *out = b;
    return true;
}

static int g_math_mode = -1;
int math_mode() {
    if (g_math_mode < 0) {
        const char* e = getenv("SCAT_MATH");
        g_math_mode = (e && (!strcmp(e, "f32") || !strcmp(e, "0"))) ? 0 : 1;
    }
    return g_math_mode;
}

// Bulk output stores of tensors of 200 MiB and more (at batch 96: the 256-channel 56 x 56 maps and the stem's output,
// 294 MiB each) are non-temporal: such a tensor cannot stay in the 256 MiB Infinity Cache anyway, and written with the
// default policy it evicts what the next kernels would have found there.  Measured (profiles/r04_store_policy.txt,
// r04_ab_store_nt.txt): 64->256 @56x56 forward 108 -> 79 us alone; train step 23.89 -> 23.71 ms with the threshold at
// 100 or 200 MiB, no gain with nt everywhere (smaller outputs ARE re-read from the cache), sc1 0.6 % slower.
// (SCAT_STORE_AUX / SCAT_STORE_NT_MIN_MB: tools build only.)
int store_policy(int64_t out_bytes) {
    static const int aux = (int)diag_env_int("SCAT_STORE_AUX", 2);
    static const int64_t min_b = diag_env_int("SCAT_STORE_NT_MIN_MB", 200) << 20;
    return out_bytes >= min_b ? aux : 0;
}

int tuning() {
    static int v = diag_env_int("SCAT_TUNE", 0);
    return v;
}

static inline int grid_for(int64_t n) { return (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192); }

#define GRID_STRIDE(e, n) \
    for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < (n); e += gridDim.x * 256ll)

__global__ void gelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    GRID_STRIDE(e, n) {
        float v = x[e];
        y[e] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
    }
}
__global__ void gelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx,
                                int64_t n) {
    GRID_STRIDE(e, n) {
        float v = x[e];
        float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752f));
        float pdf = 0.3989422804014327f * expf(-0.5f * v * v);
        dx[e] = dy[e] * (cdf + v * pdf);
    }
}
__global__ void relu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    GRID_STRIDE(e, n) y[e] = fmaxf(x[e], 0.f);
}
__global__ void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx,
                                int64_t n) {
    GRID_STRIDE(e, n) dx[e] = y[e] > 0.f ? dy[e] : 0.f;
}
__global__ void axpy_kernel(const float* __restrict__ a, const float* __restrict__ b, float alpha,
                            float* __restrict__ y, int64_t n) {
    GRID_STRIDE(e, n) y[e] = a[e] + alpha * b[e];
}

// inverted dropout with a counter-hash mask (regenerated in backward from the same seed): y = x*keep/(1-p)
__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, float p, float scale,
                               uint64_t seed) {
    GRID_STRIDE(e, n) {
        uint64_t z = (uint64_t)e * 0xD1342543DE82EF95ull + seed;
        z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
        float u = (float)(z >> 40) * (1.0f / 16777216.0f);
        y[e] = u >= p ? x[e] * scale : 0.f;
    }
}

__global__ void tokens_fwd_kernel(const float* __restrict__ x, const float* __restrict__ pe,
                                  const float* __restrict__ mask_token, const int32_t* __restrict__ masked,
                                  int nmasked, float* __restrict__ y, int64_t total, int T, int D) {
    GRID_STRIDE(e, total) {
        int d = e % D;
        int t = (e / D) % T;
        bool m = false;
        for (int k = 0; k < nmasked; ++k) m |= (masked[k] == t);
        y[e] = m ? mask_token[d] : x[e] + (pe ? pe[t * D + d] : 0.f);
    }
}
// dx = dy outside masked rows, 0 inside; dmask_token[d] = sum over batch and masked rows (fixed order)
__global__ void tokens_bwd_kernel(const float* __restrict__ dy, const int32_t* __restrict__ masked, int nmasked,
                                  float* __restrict__ dx, int64_t total, int T, int D) {
    GRID_STRIDE(e, total) {
        int t = (e / D) % T;
        bool m = false;
        for (int k = 0; k < nmasked; ++k) m |= (masked[k] == t);
        dx[e] = m ? 0.f : dy[e];
    }
}
// d(mask_token)[d] = sum over images and masked rows of dy.  A workgroup owns 64 consecutive d; its 16 wavefronts
// take the images b = w, w + 16, ... and their partial sums are combined in wavefront order (fixed order, and
// B x nmasked loads per thread become B/16 x nmasked).
__global__ __launch_bounds__(1024) void tokens_dmask_kernel(const float* __restrict__ dy,
                                                            const int32_t* __restrict__ masked, int nmasked,
                                                            float* __restrict__ dmask, int B, int T, int D) {
    __shared__ float part[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int d = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (d < D)
        for (int b = wave; b < B; b += 16)
            for (int k = 0; k < nmasked; ++k) s += dy[((int64_t)b * T + masked[k]) * D + d];
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && d < D) {
        float t = part[0][lane];
#pragma unroll
        for (int w = 1; w < 16; ++w) t += part[w][lane];
        dmask[d] = t;
    }
}

// ---------------------------------------------------------------- regressor head
// One workgroup per sample.  base_j = bias_j + W[j,:F]·feat is loop invariant.

__global__ __launch_bounds__(256) void regressor_fwd_kernel(const float* __restrict__ feat,
                                                            const float* __restrict__ feat_out,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ preds,
                                                            float* __restrict__ out, int B, int F, int P, int iters,
                                                            int root_rel) {
    __shared__ float base[128], pred[128], upd[128];
    __shared__ float fsh[2048];                        // this sample's feature row (F <= 2048), read by every output
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ldw = F + P;
    const float* fb = feat + (int64_t)b * F;
    const bool staged = F <= 2048;
    if (staged) {
        for (int f = tid; f < F; f += 256) fsh[f] = fb[f];
        __syncthreads();
    }
    // four outputs per pass: four times the weight loads in flight per wavefront (the loop is latency-bound)
    for (int j = wave; j < P; j += 16) {
        const float* wr[4];
        bool has[4];
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            has[q] = j + 4 * q < P;
            wr[q] = w + (int64_t)(has[q] ? j + 4 * q : j) * ldw;
        }
        for (int f = lane; f < F; f += 64) {
            const float x = staged ? fsh[f] : fb[f];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = fmaf(wr[q][f], x, acc[q]);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] += __shfl_xor(acc[q], o, 64);
        if (lane == 0)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (has[q]) base[j + 4 * q] = acc[q] + bias[j + 4 * q];
    }
    if (tid < P) {
        float v = mean[tid];
        if (feat_out && tid >= 3) v += feat_out[(int64_t)b * (P - 3) + tid - 3];
        pred[tid] = v;
        preds[((int64_t)0 * B + b) * P + tid] = v;
    }
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        if (tid < P) {
            float s = base[tid];
            for (int p = 0; p < P; ++p) s = fmaf(w[(int64_t)tid * ldw + F + p], pred[p], s);
            upd[tid] = s;
        }
        __syncthreads();
        if (tid < P) {
            pred[tid] += upd[tid];
            preds[((int64_t)(it + 1) * B + b) * P + tid] = pred[tid];
        }
        __syncthreads();
    }
    if (tid < P) {
        float v = pred[tid];
        if (root_rel && tid >= 3) v -= pred[3 + 3 + (tid - 3) % 3];   // root joint 1 (hand_net.py:389-391)
        out[(int64_t)b * P + tid] = v;
    }
}

// per sample: deltas[t][b][:] = grad wrt the t-th linear output, dsum[b][:] = sum_t, dfeat_out
__global__ __launch_bounds__(128) void regressor_bwd_delta_kernel(const float* __restrict__ dout,
                                                                  const float* __restrict__ w,
                                                                  float* __restrict__ deltas,
                                                                  float* __restrict__ dsum,
                                                                  float* __restrict__ dfeat_out, int B, int F, int P,
                                                                  int iters, int root_rel) {
    __shared__ float g[128], gn[128];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int ldw = F + P;
    if (tid < P) g[tid] = dout[(int64_t)b * P + tid];
    __syncthreads();
    // root-relative backward: joint 1 receives minus the sum over joints, per coordinate
    float adj = 0.f;
    if (root_rel && tid >= 6 && tid < 9) {
        for (int k = 0; k < (P - 3) / 3; ++k) adj += g[3 + 3 * k + (tid - 6)];
    }
    __syncthreads();
    if (root_rel && tid >= 6 && tid < 9) g[tid] -= adj;
    __syncthreads();
    float ds = 0.f;
    for (int it = iters - 1; it >= 0; --it) {
        if (tid < P) {
            float d = g[tid];   // delta_t = g_{t+1}
            deltas[((int64_t)it * B + b) * P + tid] = d;
            ds += d;
            float s = d;
            for (int j = 0; j < P; ++j) s = fmaf(g[j], w[(int64_t)j * ldw + F + tid], s);   // g_t = g + W2^T g
            gn[tid] = s;
        }
        __syncthreads();
        if (tid < P) g[tid] = gn[tid];
        __syncthreads();
    }
    if (tid < P) {
        dsum[(int64_t)b * P + tid] = ds;
        if (dfeat_out && tid >= 3) dfeat_out[(int64_t)b * (P - 3) + tid - 3] = g[tid];
    }
}

// dW[j][c], dbias[j]: thread per (j, c); fixed summation order over (t, b)
__global__ __launch_bounds__(256) void regressor_bwd_w_kernel(const float* __restrict__ feat,
                                                              const float* __restrict__ preds,
                                                              const float* __restrict__ deltas,
                                                              const float* __restrict__ dsum, float* __restrict__ dw,
                                                              float* __restrict__ dbias, int B, int F, int P,
                                                              int iters) {
    const int ldw = F + P;
    int64_t e = blockIdx.x * 256ll + threadIdx.x;
    if (e >= (int64_t)P * ldw) return;
    int j = e / ldw, c = e % ldw;
    float s = 0.f;
    if (c < F) {
        for (int b = 0; b < B; ++b) s = fmaf(dsum[(int64_t)b * P + j], feat[(int64_t)b * F + c], s);
    } else {
        int p = c - F;
        for (int t = 0; t < iters; ++t)
            for (int b = 0; b < B; ++b)
                s = fmaf(deltas[((int64_t)t * B + b) * P + j], preds[((int64_t)t * B + b) * P + p], s);
    }
    dw[e] = s;
    if (c == 0) {
        float sb = 0.f;
        for (int b = 0; b < B; ++b) sb += dsum[(int64_t)b * P + j];
        dbias[j] = sb;
    }
}

__global__ __launch_bounds__(256) void regressor_bwd_feat_kernel(const float* __restrict__ dsum,
                                                                 const float* __restrict__ w,
                                                                 float* __restrict__ dfeat, int B, int F, int P) {
    const int ldw = F + P;
    int64_t e = blockIdx.x * 256ll + threadIdx.x;
    if (e >= (int64_t)B * F) return;
    int b = e / F, f = e % F;
    float s = 0.f;
    for (int j = 0; j < P; ++j) s = fmaf(dsum[(int64_t)b * P + j], w[(int64_t)j * ldw + f], s);
    dfeat[e] = s;
}

// ---------------------------------------------------------------- input pipeline (dataset/load_STB.py:48-67)
// uint8 image (HWC as decoded, or CHW) -> /127.5 - 1 -> bilinear resize (align_corners = False) -> fp32 NCHW.
// One pass: 3 B/pixel read, 12 B/pixel written; replaces ToTensor + Normalize(.5,.5) + Resize(224) + H2D of fp32.
__global__ void preprocess_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int64_t total, int SH,
                                  int SW, int OH, int OW, int hwc, float ry, float rx) {
    GRID_STRIDE(e, total) {
        int ox = e % OW;
        int64_t r = e / OW;
        int oy = r % OH;
        r /= OH;
        int c = r % 3;
        int64_t b = r / 3;
        float fy = fmaxf((oy + 0.5f) * ry - 0.5f, 0.f), fx = fmaxf((ox + 0.5f) * rx - 0.5f, 0.f);
        int y0 = (int)fy, x0 = (int)fx;
        int y1 = min(y0 + 1, SH - 1), x1 = min(x0 + 1, SW - 1);
        float wy = fy - y0, wx = fx - x0;
        auto at = [&](int y, int x) -> float {
            int64_t i = hwc ? ((b * SH + y) * SW + x) * 3 + c : ((b * 3 + c) * SH + y) * (int64_t)SW + x;
            return (float)src[i] * (1.0f / 127.5f) - 1.0f;
        };
        float top = at(y0, x0) + wx * (at(y0, x1) - at(y0, x0));
        float bot = at(y1, x0) + wx * (at(y1, x1) - at(y1, x0));
        dst[e] = top + wy * (bot - top);
    }
}

// ---------------------------------------------------------------- nearest upsample (hrnet.py:107) / token mean

__global__ void upsample_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t total, int H, int W,
                                    int f) {
    const int OW = W * f, OH = H * f;
    GRID_STRIDE(e, total) {
        int ox = e % OW;
        int64_t r = e / OW;
        int oy = r % OH;
        int64_t nc = r / OH;
        y[e] = x[(nc * H + oy / f) * W + ox / f];
    }
}
__global__ void upsample_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int64_t total, int H, int W,
                                    int f) {
    const int OW = W * f;
    GRID_STRIDE(e, total) {
        int x = e % W;
        int64_t r = e / W;
        int y = r % H;
        int64_t nc = r / H;
        const float* p = dy + (nc * H * f + (int64_t)y * f) * OW + x * f;
        float s = 0.f;
        for (int a = 0; a < f; ++a)
            for (int b = 0; b < f; ++b) s += p[a * OW + b];
        dx[e] = s;
    }
}
// y[b,d] = mean_t x[b,t,d]
__global__ void tokmean_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t total, int T, int D) {
    GRID_STRIDE(e, total) {
        int d = e % D;
        int64_t b = e / D;
        float s = 0.f;
        for (int t = 0; t < T; ++t) s += x[(b * T + t) * D + d];
        y[e] = s / T;
    }
}
__global__ void tokmean_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int64_t total, int T, int D) {
    GRID_STRIDE(e, total) {
        int d = e % D;
        int64_t b = e / ((int64_t)T * D);
        dx[e] = dy[b * D + d] / T;
    }
}

// ---------------------------------------------------------------- loss (train.py:165-203)

__global__ __launch_bounds__(256) void loss_kernel(const float* __restrict__ out, const float* __restrict__ gt3d,
                                                   const float* __restrict__ gt2d, int ld_gt, float w3d, float w2d,
                                                   float* __restrict__ losses, float* __restrict__ dout, int B) {
    __shared__ double red[2][4];
    const int tid = threadIdx.x;
    double s3 = 0, s2 = 0;
    const float i3 = 1.0f / (B * 63.0f), i2 = 1.0f / (B * 42.0f);
    for (int b = tid; b < B; b += 256) {
        const float* o = out + (int64_t)b * 66;
        float* d = dout + (int64_t)b * 66;
        const float c0 = o[0], c1 = o[1], c2 = o[2];
        float dc0 = 0.f, dc1 = 0.f, dc2 = 0.f;
        for (int k = 0; k < 21; ++k) {
            float dj[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float diff = o[3 + 3 * k + c] - gt3d[(int64_t)b * ld_gt + 3 * k + c];
                s3 += (double)diff * diff;
                dj[c] = w3d * 2.0f * diff * i3;
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float xt = o[3 + 3 * k + c] + (c == 0 ? c1 : c2);
                float j2 = (c0 * xt) * 112.0f + 112.0f;
                float diff = j2 - gt2d[(int64_t)b * ld_gt + 2 * k + c];
                s2 += fabs((double)diff);
                float g = w2d * i2 * (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f)) * 112.0f;
                dc0 += g * xt;
                if (c == 0) dc1 += g * c0; else dc2 += g * c0;
                dj[c] += g * c0;
            }
            d[3 + 3 * k + 0] = dj[0];
            d[3 + 3 * k + 1] = dj[1];
            d[3 + 3 * k + 2] = dj[2];
        }
        d[0] = dc0; d[1] = dc1; d[2] = dc2;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s3 += __shfl_down(s3, o, 64); s2 += __shfl_down(s2, o, 64); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = s3; red[1][tid >> 6] = s2; }
    __syncthreads();
    if (tid == 0) {
        double l3 = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (B * 63.0);
        double l2 = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / (B * 42.0);
        losses[0] = (float)(w3d * l3 + w2d * l2);
        losses[1] = (float)l3;
        losses[2] = (float)l2;
    }
}

// ---------------------------------------------------------------- pose-length term (train.py:178-183)
//
//   pl_lengths[b] = sqrt( mean_c sum_{h,w} pl_term[b,c,h,w]^2 );  pl_mean = 0.01 * mean_b pl_lengths;
//   l_pl = mean_b (pl_lengths[b] - pl_mean)^2
// Two launches instead of a dozen torch reductions: one workgroup per image (fp64 partial sums, fixed order), then one
// wavefront-sized finish.  pl_term has no graph (hand_net.py:396), so there is no backward.
__global__ __launch_bounds__(256) void pl_lengths_kernel(const float* __restrict__ pl, float* __restrict__ lens, int C,
                                                         int HW) {
    __shared__ double red[4];
    const int tid = threadIdx.x;
    const int64_t n = (int64_t)C * HW;
    const float* p = pl + blockIdx.x * n;
    double s = 0;
    if ((n & 3) == 0 && (((uintptr_t)p) & 15) == 0) {
        for (int64_t e = tid * 4ll; e < n; e += 1024) {
            const float4 v = *(const float4*)(p + e);
            s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
        }
    } else {
        for (int64_t e = tid; e < n; e += 256) s += (double)p[e] * p[e];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) lens[blockIdx.x] = (float)sqrt((red[0] + red[1] + red[2] + red[3]) / C);
}

__global__ __launch_bounds__(64) void pl_finish_kernel(const float* __restrict__ lens, float* __restrict__ out, int B) {
    const int tid = threadIdx.x;
    double s = 0;
    for (int b = tid; b < B; b += 64) s += lens[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float pl_mean = 0.01f * (float)(s / B);
    double q = 0;
    for (int b = tid; b < B; b += 64) {
        const float dlt = lens[b] - pl_mean;
        q += (double)dlt * dlt;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    if (tid == 0) out[0] = (float)(q / B);
}

// ---------------------------------------------------------------- Adam

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                                   float b1, float b2, float eps, float bc1, float rsbc2,
                                                   float gscale) {
    GRID_STRIDE(e, n) {
        float gr = g[e] * gscale;
        float mm = b1 * m[e] + (1.f - b1) * gr;
        float vv = b2 * v[e] + (1.f - b2) * gr * gr;
        m[e] = mm;
        v[e] = vv;
        float denom = sqrtf(vv) * rsbc2 + eps;
        p[e] -= (lr / bc1) * (mm / denom);
    }
}

}  // namespace scat

using namespace scat;

extern "C" int scat_version(void) { return 100; }
extern "C" const char* scat_last_error(void) { return g_err; }
extern "C" const char* scat_last_kernel(void) { return g_label; }

extern "C" int scat_get_math_mode(void) { return scat::math_mode(); }
extern "C" int scat_set_math_mode(int mode) {
    SCAT_REQUIRE(mode == 0 || mode == 1, SCAT_E_ARG, "scat_set_math_mode: 0 (fp32 MFMA) or 1 (bf16x3 split)");
    scat::math_mode();
    scat::g_math_mode = mode;
    return SCAT_OK;
}

extern "C" int scat_check_device(void) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        set_error("scat_check_device: no HIP device");
        return SCAT_E_ARCH;
    }
    SCAT_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, SCAT_E_ARCH, "scat_check_device: %s is not gfx950",
                 prop.gcnArchName);
    return SCAT_OK;
}

#define EW_ENTRY(name, kernel, ...)                                                                 \
    SCAT_REQUIRE(n >= 0, SCAT_E_SHAPE, #name ": negative size");                                    \
    if (n == 0) return SCAT_OK;                                                                     \
    hipLaunchKernelGGL(kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);  \
    SCAT_LAUNCH_CHECK(#name);                                                                       \
    return SCAT_OK;

extern "C" int scat_gelu_fwd(const float* x, float* y, int64_t n, void* stream) {
    SCAT_REQUIRE(x && y, SCAT_E_ARG, "scat_gelu_fwd: null pointer");
    EW_ENTRY(scat_gelu_fwd, gelu_fwd_kernel, x, y, n)
}
extern "C" int scat_gelu_bwd(const float* dy, const float* x, float* dx, int64_t n, void* stream) {
    SCAT_REQUIRE(dy && x && dx, SCAT_E_ARG, "scat_gelu_bwd: null pointer");
    EW_ENTRY(scat_gelu_bwd, gelu_bwd_kernel, dy, x, dx, n)
}
extern "C" int scat_relu_fwd(const float* x, float* y, int64_t n, void* stream) {
    SCAT_REQUIRE(x && y, SCAT_E_ARG, "scat_relu_fwd: null pointer");
    EW_ENTRY(scat_relu_fwd, relu_fwd_kernel, x, y, n)
}
extern "C" int scat_relu_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream) {
    SCAT_REQUIRE(dy && y && dx, SCAT_E_ARG, "scat_relu_bwd: null pointer");
    EW_ENTRY(scat_relu_bwd, relu_bwd_kernel, dy, y, dx, n)
}
extern "C" int scat_axpy(const float* a, const float* b, float alpha, float* y, int64_t n, void* stream) {
    SCAT_REQUIRE(a && b && y, SCAT_E_ARG, "scat_axpy: null pointer");
    EW_ENTRY(scat_axpy, axpy_kernel, a, b, alpha, y, n)
}

extern "C" int scat_preprocess_u8(const uint8_t* src, float* dst, int B, int SH, int SW, int OH, int OW, int hwc,
                                  void* stream) {
    SCAT_REQUIRE(src && dst && B > 0 && SH > 0 && SW > 0 && OH > 0 && OW > 0, SCAT_E_ARG,
                 "scat_preprocess_u8: bad argument");
    int64_t n = (int64_t)B * 3 * OH * OW;
    hipLaunchKernelGGL(preprocess_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, src, dst, n, SH, SW, OH,
                       OW, hwc, (float)SH / OH, (float)SW / OW);
    SCAT_LAUNCH_CHECK("scat_preprocess_u8");
    return SCAT_OK;
}

// ---------------------------------------------------------------- exchange-unit sum (hrnet.py:117-144)
//
// One output of a StageModule's exchange is relu(sum_j f_ij(x_j)): the branch's own map, 1x1 convolutions of the
// lower-resolution branches (BatchNorm, nearest upsample by 2^k) and strided 3x3 chains of the higher-resolution ones
// (BatchNorm).  One pass: every term is read once at ITS resolution, normalised by its BatchNorm's (scale, shift) if it has
// one, and added in the order given (the reference's order of additions); the ReLU rides along.  Replaces, per term, a
// BatchNorm apply pass, an upsample pass and an add pass, and the final ReLU pass.
struct FuseSumDesc {
    const float* in[4];
    const float* sc[4];       // nullptr: the term is taken as it is
    const float* sh[4];
    int k[4];                 // log2 of the nearest-upsample factor of the term
    int n, C, H, W, relu;
};

template <bool V4>
__global__ __launch_bounds__(256) void fuse_sum_kernel(FuseSumDesc d, float* __restrict__ out, int64_t total) {
    constexpr int V = V4 ? 4 : 1;
    const int Wv = d.W / V;
    GRID_STRIDE(e, total) {
        const int xv = (int)(e % Wv);
        const int64_t r = e / Wv;
        const int y = (int)(r % d.H);
        const int64_t nc = r / d.H;
        const int c = (int)(nc % d.C);
        float acc[V];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j >= d.n) break;
            const int k = d.k[j];
            const float sc = d.sc[j] ? d.sc[j][c] : 1.f, sh = d.sc[j] ? d.sh[j][c] : 0.f;
            float v[V];
            if (k == 0) {
                if constexpr (V4) {
                    const float4 t = *reinterpret_cast<const float4*>(d.in[j] + e * 4);
                    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
                } else {
                    v[0] = d.in[j][e];
                }
            } else {
                const float* p = d.in[j] + (nc * (d.H >> k) + (y >> k)) * (int64_t)(d.W >> k);
#pragma unroll
                for (int i = 0; i < V; ++i) v[i] = p[(xv * V + i) >> k];
            }
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const float t = d.sc[j] ? fmaf(v[i], sc, sh) : v[i];
                acc[i] = j == 0 ? t : acc[i] + t;
            }
        }
        if (d.relu) {
#pragma unroll
            for (int i = 0; i < V; ++i) acc[i] = fmaxf(acc[i], 0.f);
        }
        if constexpr (V4) *reinterpret_cast<float4*>(out + e * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        else out[e] = acc[0];
    }
}

extern "C" int scat_upsample_nearest_fwd(const float* x, float* y, int B, int C, int H, int W, int factor,
                                         void* stream) {
    SCAT_REQUIRE(x && y && B > 0 && C > 0 && H > 0 && W > 0 && factor >= 1, SCAT_E_ARG,
                 "scat_upsample_nearest_fwd: bad argument");
    int64_t n = (int64_t)B * C * H * W * factor * factor;
    hipLaunchKernelGGL(upsample_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, H, W, factor);
    SCAT_LAUNCH_CHECK("scat_upsample_nearest_fwd");
    return SCAT_OK;
}
extern "C" int scat_upsample_nearest_bwd(const float* dy, float* dx, int B, int C, int H, int W, int factor,
                                         void* stream) {
    SCAT_REQUIRE(dy && dx && B > 0 && C > 0 && H > 0 && W > 0 && factor >= 1, SCAT_E_ARG,
                 "scat_upsample_nearest_bwd: bad argument");
    int64_t n = (int64_t)B * C * H * W;
    hipLaunchKernelGGL(upsample_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dy, dx, n, H, W, factor);
    SCAT_LAUNCH_CHECK("scat_upsample_nearest_bwd");
    return SCAT_OK;
}
extern "C" int scat_fuse_sum(const float* in0, const float* in1, const float* in2, const float* in3, const float* sc0,
                             const float* sc1, const float* sc2, const float* sc3, const float* sh0, const float* sh1,
                             const float* sh2, const float* sh3, int k0, int k1, int k2, int k3, int n, float* out, int B,
                             int C, int H, int W, int relu, void* stream) {
    SCAT_REQUIRE(out && n >= 1 && n <= 4 && B > 0 && C > 0 && H > 0 && W > 0, SCAT_E_ARG, "scat_fuse_sum: bad argument");
    FuseSumDesc d{};
    const float* in[4] = {in0, in1, in2, in3};
    const float* sc[4] = {sc0, sc1, sc2, sc3};
    const float* sh[4] = {sh0, sh1, sh2, sh3};
    const int k[4] = {k0, k1, k2, k3};
    bool v4 = W % 4 == 0 && ((uintptr_t)out & 15) == 0;
    for (int j = 0; j < n; ++j) {
        SCAT_REQUIRE(in[j] && (sc[j] == nullptr) == (sh[j] == nullptr), SCAT_E_ARG, "scat_fuse_sum: null term / scale without shift");
        SCAT_REQUIRE(k[j] >= 0 && k[j] < 8 && H % (1 << k[j]) == 0 && W % (1 << k[j]) == 0, SCAT_E_SHAPE,
                     "scat_fuse_sum: the map is not a multiple of a term's upsample factor");
        d.in[j] = in[j]; d.sc[j] = sc[j]; d.sh[j] = sh[j]; d.k[j] = k[j];
        if (k[j] == 0 && ((uintptr_t)in[j] & 15)) v4 = false;
    }
    d.n = n; d.C = C; d.H = H; d.W = W; d.relu = relu;
    const int64_t total = (int64_t)B * C * H * W / (v4 ? 4 : 1);
    if (v4) hipLaunchKernelGGL(fuse_sum_kernel<true>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, d, out, total);
    else hipLaunchKernelGGL(fuse_sum_kernel<false>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, d, out, total);
    SCAT_LAUNCH_CHECK("scat_fuse_sum");
    return SCAT_OK;
}
extern "C" int scat_token_mean_fwd(const float* x, float* y, int B, int T, int D, void* stream) {
    SCAT_REQUIRE(x && y && B > 0 && T > 0 && D > 0, SCAT_E_ARG, "scat_token_mean_fwd: bad argument");
    int64_t n = (int64_t)B * D;
    hipLaunchKernelGGL(tokmean_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, T, D);
    SCAT_LAUNCH_CHECK("scat_token_mean_fwd");
    return SCAT_OK;
}
extern "C" int scat_token_mean_bwd(const float* dy, float* dx, int B, int T, int D, void* stream) {
    SCAT_REQUIRE(dy && dx && B > 0 && T > 0 && D > 0, SCAT_E_ARG, "scat_token_mean_bwd: bad argument");
    int64_t n = (int64_t)B * T * D;
    hipLaunchKernelGGL(tokmean_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dy, dx, n, T, D);
    SCAT_LAUNCH_CHECK("scat_token_mean_bwd");
    return SCAT_OK;
}

extern "C" int scat_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, void* stream) {
    SCAT_REQUIRE(x && y && p >= 0.f && p < 1.f, SCAT_E_ARG, "scat_dropout: bad argument");
    EW_ENTRY(scat_dropout, dropout_kernel, x, y, n, p, 1.0f / (1.0f - p), seed)
}

extern "C" int scat_tokens_fwd(const float* x, const float* pe, const float* mask_token, const int32_t* masked,
                               int nmasked, float* y, int B, int T, int D, void* stream) {
    SCAT_REQUIRE(x && y && B > 0 && T > 0 && D > 0, SCAT_E_ARG, "scat_tokens_fwd: bad argument");
    SCAT_REQUIRE(nmasked == 0 || (masked && mask_token), SCAT_E_ARG, "scat_tokens_fwd: mask arguments");
    int64_t n = (int64_t)B * T * D;
    hipLaunchKernelGGL(tokens_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, pe, mask_token,
                       masked, nmasked, y, n, T, D);
    SCAT_LAUNCH_CHECK("scat_tokens_fwd");
    return SCAT_OK;
}

extern "C" int scat_tokens_bwd(const float* dy, const int32_t* masked, int nmasked, float* dx, float* dmask_token,
                               int B, int T, int D, void* stream) {
    SCAT_REQUIRE(dy && dx && B > 0 && T > 0 && D > 0, SCAT_E_ARG, "scat_tokens_bwd: bad argument");
    SCAT_REQUIRE(nmasked == 0 || masked, SCAT_E_ARG, "scat_tokens_bwd: mask arguments");
    int64_t n = (int64_t)B * T * D;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(tokens_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, st, dy, masked, nmasked, dx, n, T, D);
    if (dmask_token)
        hipLaunchKernelGGL(tokens_dmask_kernel, dim3(cdiv(D, 64)), dim3(1024), 0, st, dy, masked, nmasked, dmask_token,
                           B, T, D);
    SCAT_LAUNCH_CHECK("scat_tokens_bwd");
    return SCAT_OK;
}

extern "C" int scat_regressor_fwd(const float* feat, const float* feat_out, const float* mean, const float* w,
                                  const float* bias, float* preds, float* out, int B, int F, int P, int iters,
                                  int root_relative, void* stream) {
    SCAT_REQUIRE(feat && mean && w && bias && preds && out, SCAT_E_ARG, "scat_regressor_fwd: null pointer");
    SCAT_REQUIRE(B > 0 && F > 0 && P >= 9 && P <= 128 && iters >= 0, SCAT_E_SHAPE, "scat_regressor_fwd: need 9 <= P <= 128");
    SCAT_REQUIRE(!root_relative || (P - 3) % 3 == 0, SCAT_E_SHAPE, "scat_regressor_fwd: root-relative needs P = 3 + 3J");
    hipLaunchKernelGGL(regressor_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, feat, feat_out, mean, w, bias,
                       preds, out, B, F, P, iters, root_relative);
    SCAT_LAUNCH_CHECK("scat_regressor_fwd");
    return SCAT_OK;
}

extern "C" int64_t scat_regressor_bwd_ws(int B, int F, int P, int iters) {
    return (int64_t)((int64_t)(iters > 0 ? iters : 1) * B * P + (int64_t)B * P) * sizeof(float);
}

extern "C" int scat_regressor_bwd(const float* dout, const float* feat, const float* preds, const float* w,
                                  float* dfeat, float* dfeat_out, float* dw, float* dbias, int B, int F, int P,
                                  int iters, int root_relative, void* ws, int64_t ws_bytes, void* stream) {
    SCAT_REQUIRE(dout && feat && preds && w && dfeat && dw && dbias, SCAT_E_ARG, "scat_regressor_bwd: null pointer");
    SCAT_REQUIRE(B > 0 && F > 0 && P >= 9 && P <= 128 && iters >= 0, SCAT_E_SHAPE, "scat_regressor_bwd: need 9 <= P <= 128");
    SCAT_REQUIRE(ws && ws_bytes >= scat_regressor_bwd_ws(B, F, P, iters), SCAT_E_WORKSPACE,
                 "scat_regressor_bwd: workspace too small");
    float* deltas = (float*)ws;
    float* dsum = deltas + (int64_t)(iters > 0 ? iters : 1) * B * P;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(regressor_bwd_delta_kernel, dim3(B), dim3(128), 0, st, dout, w, deltas, dsum, dfeat_out, B, F,
                       P, iters, root_relative);
    int64_t nw = (int64_t)P * (F + P);
    if (iters > 0) {
        // dW = [dsum^T . feat | sum_t deltas_t^T . preds_t], dbias = column sums of dsum: three small contractions on
        // the general engine instead of a 96..288-long dependent FMA chain per element
        const int ldw = F + P;
        if (int e = scat_gemm(dsum, 1, P, feat, F, 1, dw, ldw, 1, P, F, B, nullptr, 0, 0, nullptr, 0, stream)) return e;
        if (int e = scat_gemm(deltas, 1, P, preds, P, 1, dw + F, ldw, 1, P, P, iters * B, nullptr, 0, 0, nullptr, 0, stream))
            return e;
        if (int e = scat_colsum(dsum, dbias, B, P, 0, stream)) return e;
    } else {
        hipLaunchKernelGGL(regressor_bwd_w_kernel, dim3((int)((nw + 255) / 256)), dim3(256), 0, st, feat, preds,
                           (const float*)deltas, (const float*)dsum, dw, dbias, B, F, P, iters);
    }
    int64_t nf = (int64_t)B * F;
    hipLaunchKernelGGL(regressor_bwd_feat_kernel, dim3((int)((nf + 255) / 256)), dim3(256), 0, st,
                       (const float*)dsum, w, dfeat, B, F, P);
    SCAT_LAUNCH_CHECK("scat_regressor_bwd");
    return SCAT_OK;
}

extern "C" int scat_loss_fwd_bwd(const float* out, const float* gt3d, const float* gt2d, int ld_gt, float w3d,
                                 float w2d, float* losses, float* dout, int B, void* stream) {
    SCAT_REQUIRE(out && gt3d && gt2d && losses && dout && B > 0 && ld_gt > 0, SCAT_E_ARG,
                 "scat_loss_fwd_bwd: bad argument");
    hipLaunchKernelGGL(loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, out, gt3d, gt2d, ld_gt, w3d, w2d,
                       losses, dout, B);
    SCAT_LAUNCH_CHECK("scat_loss_fwd_bwd");
    return SCAT_OK;
}

extern "C" int scat_pose_length_term(const float* pl_term, float* lens, float* l_pl, int B, int C, int HW, void* stream) {
    SCAT_REQUIRE(pl_term && lens && l_pl && B > 0 && C > 0 && HW > 0, SCAT_E_ARG, "scat_pose_length_term: bad argument");
    hipLaunchKernelGGL(pl_lengths_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, pl_term, lens, C, HW);
    hipLaunchKernelGGL(pl_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, lens, l_pl, B);
    SCAT_LAUNCH_CHECK("scat_pose_length_term");
    return SCAT_OK;
}

extern "C" int scat_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                         float eps, int step, float grad_scale, void* stream) {
    SCAT_REQUIRE(p && g && m && v, SCAT_E_ARG, "scat_adam: null pointer");
    SCAT_REQUIRE(n >= 0 && step >= 1, SCAT_E_SHAPE, "scat_adam: bad size/step");
    if (n == 0) return SCAT_OK;
    double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1,
                       beta2, eps, (float)bc1, (float)(1.0 / sqrt(bc2)), grad_scale);
    SCAT_LAUNCH_CHECK("scat_adam");
    return SCAT_OK;
}

extern "C" int scat_epilogue_stats_arm(float* buf, int64_t bytes) {
    scat::g_epi.buf = buf;
    scat::g_epi.cap = buf ? bytes / 4 : 0;
    scat::g_epi.groups = 0;
    scat::g_epi.shift = nullptr;
    return SCAT_OK;
}
extern "C" int scat_epilogue_stats_arm_shift(float* buf, int64_t bytes, const float* shift) {
    scat_epilogue_stats_arm(buf, bytes);
    scat::g_epi.shift = buf ? shift : nullptr;
    return SCAT_OK;
}
extern "C" int scat_epilogue_bnb_arm(const float* x, const uint8_t* mask, const float* mean, int64_t n, float* part,
                                     int64_t part_bytes) {
    scat::g_bnb = scat::EpiBnb{x, mask, mean, (x && mask && mean) ? part : nullptr, part ? part_bytes / 4 : 0, n, 0};
    return SCAT_OK;
}
extern "C" int scat_epilogue_bnb_groups(void) {
    const int g = scat::g_bnb.groups;
    scat::g_bnb.part = nullptr;
    scat::g_bnb.groups = 0;
    return g;
}
extern "C" int scat_epilogue_stats_groups(void) {
    const int g = scat::g_epi.groups;
    scat::g_epi.buf = nullptr;
    scat::g_epi.groups = 0;
    return g;
}

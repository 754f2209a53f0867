// BatchNorm2d (train/eval) and LayerNorm, forward + backward, NCHW fp32.
// HBM-bound passes: 16-B loads where HW % 4 == 0, per-channel sums accumulated in fp64 in a
// fixed order (two-stage, no atomics) so results are bitwise reproducible run to run.
// Reference: nn.BatchNorm2d at models/resnet.py:68-73,108,131; nn.LayerNorm at
// models/vision_transformer.py:20-26.
#include <atomic>
#include <chrono>

#include "common.h"

namespace scat {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum of two doubles (256 threads); result valid in thread 0
__device__ __forceinline__ void block_sum2(double& a, double& b, double* sh) {
    a = wave_sum(a);
    b = wave_sum(b);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { sh[w] = a; sh[4 + w] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = sh[0] + sh[1] + sh[2] + sh[3];
        b = sh[4] + sh[5] + sh[6] + sh[7];
    }
}

// ---------------------------------------------------------------- finalize by the last arriver, no second launch
//
// A per-channel reduction over S workgroups followed by a per-channel finalize used to be two launches, the second a few
// microseconds of work behind a full launch boundary on the critical path (22 of them in a ResNet-50 backward, 512 in an
// HRNet-W32 step).  Now every workgroup publishes its partial {s1, s2, tag} with agent-coherent (sc1, write-through)
// stores — data, wait, tag, wait — and then looks at the tags of all S slots of its channel with sc1 loads: the workgroup
// whose publication completed last necessarily sees them all (MI355X guide, Guideline 16: sc1 stores need no release
// fence, sc1 loads no acquire; no buffer_wbl2 anywhere — the device-scope fence of a first attempt at this cost 2.7 ms
// per step).  Several workgroups may see a complete channel; an atomic exchange on a claim word picks exactly one (the
// running statistics are updated in place, so the finish must run once).  `tag` is unique per launch: slots and claim
// words live in a recycled workspace and are never cleared.  The sum runs over the slots in index order: same bits as the
// two-launch form.
struct BnSlot {
    double s1, s2;
    unsigned long long tag, pad;
};
// Launch tags: a bijective 64-bit mix (splitmix64's finaliser) of a salted counter — unique per launch of this process, and
// a recycled workspace holds one by accident with probability 2^-64 per slot (a plain counter would collide with any
// small integer a previous owner of the memory left there).
static std::atomic<unsigned long long> g_bn_tag{1};
static unsigned long long bn_next_tag() {
    static const unsigned long long salt =
        (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count() * 0x9E3779B97F4A7C15ull;
    unsigned long long z = g_bn_tag.fetch_add(1, std::memory_order_relaxed) + salt;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Called by ALL 256 threads of workgroup (c, s) with the workgroup's sums in thread 0: publish, then return true in
// thread 0 of exactly one workgroup of the channel, with the channel totals in (s1, s2).  The S tags are checked by S
// threads at once (S <= 256, bn_splits): one round trip, not S.
__device__ __forceinline__ bool bn_publish(BnSlot* __restrict__ slots, unsigned long long* __restrict__ claim, int s,
                                           int S, unsigned long long tag, double& s1, double& s2) {
    __shared__ double shp[512];
    __shared__ int won;
    const int t = threadIdx.x;
    if (t == 0) {
        __hip_atomic_store(&slots[s].s1, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&slots[s].s2, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);
        __hip_atomic_store(&slots[s].tag, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);
    }
    __syncthreads();
    const bool here = t >= S || __hip_atomic_load(&slots[t < S ? t : 0].tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tag;
    if (!__syncthreads_and(here)) return false;
    if (t == 0) won = __hip_atomic_exchange(claim, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag;
    __syncthreads();
    if (!won) return false;
    if (t < S) {
        shp[2 * t] = __hip_atomic_load(&slots[t].s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        shp[2 * t + 1] = __hip_atomic_load(&slots[t].s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (t != 0) return false;
    double t1 = 0, t2 = 0;
    for (int q = 0; q < S; ++q) { t1 += shp[2 * q]; t2 += shp[2 * q + 1]; }      // slot order: the two-launch form's bits
    s1 = t1; s2 = t2;
    return true;
}

struct BnFin {                // what a last arriver needs to finish a channel (nullptr slots: two-launch form)
    BnSlot* slots;            // [C][S]
    unsigned long long* claim;   // [C]
    unsigned long long tag;
    double count;
    // forward statistics
    const float* gamma; const float* beta; float* rmean; float* rvar; float momentum, eps;
    float* save_mean; float* save_invstd; float* scale; float* shift;
    // backward
    float* dgamma; float* dbeta; float* coef; const float* mean; const float* invstd; int coef3;
};

static inline int ew_grid(int64_t n) { return (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384); }

static int bn_splits(int B, int C) {
    static const int target = diag_env_int("SCAT_BN_BLOCKS", 2048);
    int s = cdiv(target, C);
    if (s > B) s = B;
    if (s > 256) s = 256;      // (one thread per slot in bn_publish)
    return s < 1 ? 1 : s;
}

// ---------------------------------------------------------------- BN forward statistics

__device__ __forceinline__ void bn_finish(int c, double s1, double s2, double count, const float* __restrict__ gamma,
                                          const float* __restrict__ beta, float* __restrict__ rmean,
                                          float* __restrict__ rvar, float momentum, float eps,
                                          float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                          float* __restrict__ scale, float* __restrict__ shift);

// Each block reduces channel c over images n = s, s+S, ...; the (image, pixel) pair is flattened so small
// feature maps (7x7) still use every lane.  V = 4: 16-B loads (HW % 4 == 0).
template <int V>
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, int B, int C, int HW, int S,
                                                       FastDiv dHWv, double* __restrict__ part, BnFin f) {
    const int c = blockIdx.x, s = blockIdx.y;
    const int nimg = (B - s + S - 1) / S, hwv = HW / V;
    double s1 = 0, s2 = 0;
    for (int idx = threadIdx.x; idx < nimg * hwv; idx += 256) {
        const int nl = (int)dHWv.div((uint32_t)idx), i = idx - nl * hwv;
        const float* p = x + ((int64_t)(s + nl * S) * C + c) * HW + i * V;
        if (V == 4) {
            float4 v = *(const float4*)p;
            s1 += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
            s2 += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
        } else {
            double v = *p;
            s1 += v;
            s2 += v * v;
        }
    }
    __shared__ double sh[8];
    block_sum2(s1, s2, sh);
    if (f.slots) {
        if (bn_publish(f.slots + (int64_t)c * S, f.claim + c, s, S, f.tag, s1, s2))
            bn_finish(c, s1, s2, f.count, f.gamma, f.beta, f.rmean, f.rvar, f.momentum, f.eps, f.save_mean, f.save_invstd,
                      f.scale, f.shift);
    } else if (threadIdx.x == 0) {
        part[((int64_t)c * S + s) * 2 + 0] = s1;
        part[((int64_t)c * S + s) * 2 + 1] = s2;
    }
}

// statistics -> what forward and backward need of them (+ running-stat update, unbiased variance)
__device__ __forceinline__ void bn_finish_mv(int c, double mean, double var, double count,
                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                             float* __restrict__ rmean, float* __restrict__ rvar, float momentum,
                                             float eps, float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                             float* __restrict__ scale, float* __restrict__ shift);
__device__ __forceinline__ void bn_finish(int c, double s1, double s2, double count, const float* __restrict__ gamma,
                                          const float* __restrict__ beta, float* __restrict__ rmean,
                                          float* __restrict__ rvar, float momentum, float eps,
                                          float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                          float* __restrict__ scale, float* __restrict__ shift) {
    const double mean = s1 / count;
    bn_finish_mv(c, mean, s2 / count - mean * mean, count, gamma, beta, rmean, rvar, momentum, eps, save_mean, save_invstd,
                 scale, shift);
}
__device__ __forceinline__ void bn_finish_mv(int c, double mean, double var, double count,
                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                             float* __restrict__ rmean, float* __restrict__ rvar, float momentum,
                                             float eps, float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                             float* __restrict__ scale, float* __restrict__ shift) {
    if (var < 0) var = 0;
    double invstd = 1.0 / sqrt(var + (double)eps);
    save_mean[c] = (float)mean;
    save_invstd[c] = (float)invstd;
    float sc = gamma[c] * (float)invstd;
    scale[c] = sc;
    shift[c] = beta[c] - (float)mean * sc;
    if (rmean) {
        double unb = count > 1 ? var * count / (count - 1) : var;
        rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * mean);
        rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * unb);
    }
}

__global__ void bn_finalize_kernel(const double* __restrict__ part, int C, int S, double count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ rmean, float* __restrict__ rvar, float momentum, float eps,
                                   float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                   float* __restrict__ scale, float* __restrict__ shift) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1 = 0, s2 = 0;
    for (int s = 0; s < S; ++s) {
        s1 += part[((int64_t)c * S + s) * 2];
        s2 += part[((int64_t)c * S + s) * 2 + 1];
    }
    bn_finish(c, s1, s2, count, gamma, beta, rmean, rvar, momentum, eps, save_mean, save_invstd, scale, shift);
}

// Statistics and finalize in ONE launch when the channels alone fill the GPU (C >= 256: one 1024-thread workgroup per
// channel, 16 wavefronts each) — the forward is a strict dependency chain, so every launch saved is latency saved.
template <int V>
__global__ __launch_bounds__(1024) void bn_stats_fin_kernel(const float* __restrict__ x, int B, int C, int HW,
                                                            FastDiv dHWv, double count,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ rmean,
                                                            float* __restrict__ rvar, float momentum, float eps,
                                                            float* __restrict__ save_mean,
                                                            float* __restrict__ save_invstd, float* __restrict__ scale,
                                                            float* __restrict__ shift) {
    const int c = blockIdx.x;
    const int hwv = HW / V;
    double s1 = 0, s2 = 0;
    for (int idx = threadIdx.x; idx < B * hwv; idx += 1024) {
        const int n = (int)dHWv.div((uint32_t)idx), i = idx - n * hwv;
        const float* p = x + ((int64_t)n * C + c) * HW + i * V;
        if (V == 4) {
            float4 v = *(const float4*)p;
            s1 += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
            s2 += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
        } else {
            double v = *p;
            s1 += v;
            s2 += v * v;
        }
    }
    __shared__ double sh[32];
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { sh[w] = s1; sh[16 + w] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0, b = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { a += sh[k]; b += sh[16 + k]; }
        bn_finish(c, a, b, count, gamma, beta, rmean, rvar, momentum, eps, save_mean, save_invstd, scale, shift);
    }
}

// Statistics from the row sums a convolution's epilogue left behind (OutDesc::stats, gemm_engine.h store_tile):
// part[c][G][2] = (sum, sum of squares) of channel c over column group g, fp32; summed here in fp64, fixed order.
// (stat_shift: the partials are sums of (x - c) and (x - c)^2 about a per-channel reference c — mean = c + S1/N,
// var = S2/N - (S1/N)^2, the same finish)
__global__ __launch_bounds__(256) void bn_partials_fin_kernel(const float* __restrict__ part, int G, double count,
                                                               const float* __restrict__ stat_shift,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ rmean,
                                                               float* __restrict__ rvar, float momentum, float eps,
                                                               float* __restrict__ save_mean,
                                                               float* __restrict__ save_invstd, float* __restrict__ scale,
                                                               float* __restrict__ shift) {
    const int c = blockIdx.x;
    const float2* p = (const float2*)part + (int64_t)c * G;
    double s1 = 0, s2 = 0;
    for (int g = threadIdx.x; g < G; g += 256) {
        const float2 v = p[g];
        s1 += (double)v.x;
        s2 += (double)v.y;
    }
    __shared__ double sh[8];
    block_sum2(s1, s2, sh);
    if (threadIdx.x == 0) {
        if (stat_shift) {
            const double d1 = s1 / count;
            bn_finish_mv(c, (double)finite_or_zero(stat_shift[c]) + d1, s2 / count - d1 * d1, count, gamma, beta, rmean, rvar, momentum, eps,
                         save_mean, save_invstd, scale, shift);
        } else {
            bn_finish(c, s1, s2, count, gamma, beta, rmean, rvar, momentum, eps, save_mean, save_invstd, scale, shift);
        }
    }
}

__global__ void bn_eval_fold_kernel(const float* gamma, const float* beta, const float* rmean, const float* rvar,
                                    float eps, int C, float* scale, float* shift) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float invstd = 1.0f / sqrtf(rvar[c] + eps);
    float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - rmean[c] * sc;
}

// ---------------------------------------------------------------- BN apply

typedef float v4f_t __attribute__((ext_vector_type(4)));
// The forward tensor a BatchNorm pass streams through once (x: a raw convolution output) is read non-temporally: it is not
// re-read before the caches have turned over, and loaded with the default policy it evicts what is (same-box A/B of the train
// step, profiles/r04_ab_ldnt.txt: 22.92 / 22.92 -> 22.75 / 22.87 ms).
__device__ __forceinline__ float4 ld4x(const float* p) {
    const v4f_t v = __builtin_nontemporal_load((const v4f_t*)p);
    return make_float4(v[0], v[1], v[2], v[3]);
}
// a 16-byte store with the cache policy of store_policy(): nt for bulk outputs that nothing re-reads soon
__device__ __forceinline__ void st4(float4* p, const float4 v, const int nt) {
    if (nt) __builtin_nontemporal_store(v4f_t{v.x, v.y, v.z, v.w}, (v4f_t*)p);
    else *p = v;
}

// flat element-wise pass; channel of element e = (e / HW) % C by mul-shift division
template <int V>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                       const float* __restrict__ shift,
                                                       const float* __restrict__ res,
                                                       const float* __restrict__ rscale,
                                                       const float* __restrict__ rshift, int relu,
                                                       float* __restrict__ y, uint8_t* __restrict__ mask,
                                                       int64_t nvec, FastDiv dHWv, FastDiv dC, int nt) {
    for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < nvec; e += gridDim.x * 256ll) {
        const uint32_t row = dHWv.div((uint32_t)e);
        const int c = (int)(row - dC.div(row) * dC.d);
        const float sc = scale[c], sh = shift[c];
        // the residual may itself be a raw convolution output with its own BatchNorm (the shortcut branch)
        const float rs = rscale ? rscale[c] : 1.f, rh = rscale ? rshift[c] : 0.f;
        if (V == 4) {
            float4 v = ld4x(x + 4 * e);
            v.x = fmaf(v.x, sc, sh); v.y = fmaf(v.y, sc, sh); v.z = fmaf(v.z, sc, sh); v.w = fmaf(v.w, sc, sh);
            if (res) {
                float4 r = ((const float4*)res)[e];
                if (rscale) { r.x = fmaf(r.x, rs, rh); r.y = fmaf(r.y, rs, rh); r.z = fmaf(r.z, rs, rh); r.w = fmaf(r.w, rs, rh); }
                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
            }
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            st4((float4*)y + e, v, nt);
            // sign bits of the output, one byte per float4: what the backward needs of y (32x fewer bytes)
            if (mask) mask[e] = (uint8_t)((v.x > 0.f) | ((v.y > 0.f) << 1) | ((v.z > 0.f) << 2) | ((v.w > 0.f) << 3));
        } else {
            float v = fmaf(x[e], sc, sh);
            if (res) v += rscale ? fmaf(res[e], rs, rh) : res[e];
            if (relu) v = fmaxf(v, 0.f);
            y[e] = v;
        }
    }
}

// ---------------------------------------------------------------- BN backward

__device__ __forceinline__ float bn_mask(float dy, float x, float yv, bool has_y, int relu, float sc, float sh) {
    if (has_y) return yv > 0.f ? dy : 0.f;      // (a sign-mask bit is passed as yv = 1 / 0)
    if (relu) return fmaf(x, sc, sh) > 0.f ? dy : 0.f;
    return dy;
}

// per-channel constants of a BatchNorm backward from the two sums (the bodies of bn_bwd_finalize_kernel / _finalize3)
__device__ __forceinline__ void bn_bwd_finish(int c, int C, double s1, double s2, const BnFin& f) {
    f.dbeta[c] = (float)s1;
    f.dgamma[c] = (float)s2;
    if (!f.coef3) {
        f.coef[2 * c] = (float)(s1 / f.count);
        f.coef[2 * c + 1] = (float)(s2 / f.count);
    } else {
        const float k1 = (float)(s1 / f.count), k2 = (float)(s2 / f.count);
        const float ca = f.gamma[c] * f.invstd[c], cb = -ca * f.invstd[c] * k2;
        f.coef[c] = ca;
        f.coef[C + c] = cb;
        f.coef[2 * C + c] = -ca * k1 - cb * f.mean[c];
    }
}

template <int V>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ yout,
                                                            const uint8_t* __restrict__ ymask, int relu,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, int B, int C, int HW,
                                                            int S, FastDiv dHWv, double* __restrict__ part, BnFin f) {
    const int c = blockIdx.x, s = blockIdx.y;
    const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
    const int nimg = (B - s + S - 1) / S, hwv = HW / V;
    const bool has_m = V == 4 && ymask != nullptr;
    const bool has_y = yout != nullptr || has_m;
    double s1 = 0, s2 = 0;
    for (int idx = threadIdx.x; idx < nimg * hwv; idx += 256) {
        const int nl = (int)dHWv.div((uint32_t)idx), i = idx - nl * hwv;
        const int64_t off = ((int64_t)(s + nl * S) * C + c) * HW + i * V;
        float xv[V], gv[V], yv[V];
        if (V == 4) {
            float4 t = ld4x(x + off); xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
            t = *(const float4*)(dy + off); gv[0] = t.x; gv[1] = t.y; gv[2] = t.z; gv[3] = t.w;
            if (has_m) {
                const uint32_t m = ymask[off >> 2];
                yv[0] = (float)(m & 1u); yv[1] = (float)((m >> 1) & 1u); yv[2] = (float)((m >> 2) & 1u); yv[3] = (float)((m >> 3) & 1u);
            } else if (has_y) { t = *(const float4*)(yout + off); yv[0] = t.x; yv[1] = t.y; yv[2] = t.z; yv[3] = t.w; }
        } else {
            xv[0] = x[off]; gv[0] = dy[off];
            if (has_y) yv[0] = yout[off];
        }
#pragma unroll
        for (int q = 0; q < V; ++q) {
            float g = bn_mask(gv[q], xv[q], has_y ? yv[q] : 0.f, has_y, relu, sc, sh);
            s1 += g;
            s2 += (double)g * ((xv[q] - mu) * is);
        }
    }
    __shared__ double shm[8];
    block_sum2(s1, s2, shm);
    if (f.slots) {
        if (bn_publish(f.slots + (int64_t)c * S, f.claim + c, s, S, f.tag, s1, s2)) bn_bwd_finish(c, C, s1, s2, f);
    } else if (threadIdx.x == 0) {
        part[((int64_t)c * S + s) * 2 + 0] = s1;
        part[((int64_t)c * S + s) * 2 + 1] = s2;
    }
}

// ---------------------------------------------------------------- BatchNorm backward straight from a max-pool's gradient
//
// The stem (models/resnet.py:108-112: conv1 -> bn1 -> relu -> maxpool 3x3 / 2) used to scatter the pooled gradient into
// the full-resolution map (write 308 MB at batch 96), reduce over that map and x (read 616 MB) and apply (read 616 MB,
// write 308 MB).  The gradient of bn1's output is non-zero only at arg-max positions, so:
//   * the two sums run over the POOLED grid: window q contributes g_q at its arg-max pixel p(q) — mask and x-hat taken
//     from x[p(q)] (a gather inside a 3 x 3 window: the same cache lines the neighbours touch);
//   * the apply pass rebuilds the scattered gradient of a 2 x 2 input quad from the six windows that see it (the
//     max-pool backward's own arithmetic, pool.hip) and writes dx directly.
// 96 + 96 + 308 MB read and 308 MB written instead of 1.9 GB.  Even H, W with W % 4 == 0 (the stem: 112 x 112 -> 56 x 56).
__global__ __launch_bounds__(256) void bn_bwd_pool_reduce_kernel(const float* __restrict__ dyp, const int8_t* __restrict__ idx,
                                                                 const float* __restrict__ x, int relu,
                                                                 const float* __restrict__ scale,
                                                                 const float* __restrict__ shift,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, int B, int C, int H, int W,
                                                                 int S, double* __restrict__ part, BnFin f) {
    const int c = blockIdx.x, s = blockIdx.y;
    const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
    const int OH = H / 2, OW = W / 2, nimg = (B - s + S - 1) / S, ohw = OH * OW;
    double s1 = 0, s2 = 0;
    for (int e = threadIdx.x; e < nimg * ohw; e += 256) {
        const int nl = e / ohw, q = e - nl * ohw;
        const int a = q / OW, b = q - a * OW;
        const int64_t nc = (int64_t)(s + nl * S) * C + c;
        const float g0 = dyp[nc * ohw + q];
        // (a tap written by scat_maxpool3x3s2_fwd is 0..8 and points inside the map; anything else — an index buffer nobody
        //  filled — is clamped: a wrong number, never an access outside x)
        const int t = min(max((int)idx[nc * ohw + q], 0), 8);
        const int iy = min(max(2 * a - 1 + t / 3, 0), H - 1), ix = min(max(2 * b - 1 + t % 3, 0), W - 1);
        const float xv = x[(nc * H + iy) * W + ix];
        const float g = bn_mask(g0, xv, 0.f, false, relu, sc, sh);
        s1 += g;
        s2 += (double)g * ((xv - mu) * is);
    }
    __shared__ double shm[8];
    block_sum2(s1, s2, shm);
    if (f.slots) {
        if (bn_publish(f.slots + (int64_t)c * S, f.claim + c, s, S, f.tag, s1, s2)) bn_bwd_finish(c, C, s1, s2, f);
    } else if (threadIdx.x == 0) {
        part[((int64_t)c * S + s) * 2 + 0] = s1;
        part[((int64_t)c * S + s) * 2 + 1] = s2;
    }
}

// one thread = two horizontally adjacent 2 x 2 input quads (pool.hip maxpool_bwd_quad_kernel), then the BatchNorm backward
__global__ __launch_bounds__(256) void bn_bwd_pool_apply_kernel(const float* __restrict__ dyp, const int8_t* __restrict__ idx,
                                                                const float* __restrict__ x, int relu,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ coef, float* __restrict__ dx,
                                                                int64_t total, int C, int W, int OH, int OW) {
    const int QW = OW / 2;
    for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < total; e += gridDim.x * 256ll) {
        const int b2 = e % QW;
        const int64_t r = e / QW;
        const int a = r % OH;
        const int64_t nc = r / OH;
        const int c = (int)(nc % C);
        const float* g = dyp + nc * OH * OW;
        const int8_t* id = idx + nc * OH * OW;
        const int b = 2 * b2;
        float gv[2][3];
        int iv[2][3];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const bool ok = (a + u < OH) && (b + v < OW);
                const int q = ok ? (a + u) * OW + b + v : 0;
                gv[u][v] = ok ? g[q] : 0.f;
                iv[u][v] = ok ? (int)id[q] : -1;
            }
        float top[4], bot[4];
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            top[2 * v] = iv[0][v] == 4 ? gv[0][v] : 0.f;
            top[2 * v + 1] = (iv[0][v] == 5 ? gv[0][v] : 0.f) + (iv[0][v + 1] == 3 ? gv[0][v + 1] : 0.f);
            bot[2 * v] = (iv[0][v] == 7 ? gv[0][v] : 0.f) + (iv[1][v] == 1 ? gv[1][v] : 0.f);
            bot[2 * v + 1] = (iv[0][v] == 8 ? gv[0][v] : 0.f) + (iv[0][v + 1] == 6 ? gv[0][v + 1] : 0.f) +
                             (iv[1][v] == 2 ? gv[1][v] : 0.f) + (iv[1][v + 1] == 0 ? gv[1][v + 1] : 0.f);
        }
        const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
        const float k1 = coef[2 * c], k2 = coef[2 * c + 1], gi = gamma[c] * is;
        const int64_t off = (nc * 2 * OH + 2 * a) * W + 2 * b;
        const float4 x0 = *(const float4*)(x + off), x1 = *(const float4*)(x + off + W);
        const float xt[4] = {x0.x, x0.y, x0.z, x0.w}, xb[4] = {x1.x, x1.y, x1.z, x1.w};
        float ot[4], ob[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float gt = bn_mask(top[q], xt[q], 0.f, false, relu, sc, sh), gb = bn_mask(bot[q], xb[q], 0.f, false, relu, sc, sh);
            ot[q] = gi * (gt - k1 - (xt[q] - mu) * is * k2);
            ob[q] = gi * (gb - k1 - (xb[q] - mu) * is * k2);
        }
        *(float4*)(dx + off) = make_float4(ot[0], ot[1], ot[2], ot[3]);
        *(float4*)(dx + off + W) = make_float4(ob[0], ob[1], ob[2], ob[3]);
    }
}

// The same reduction with its finalize in ONE launch when the channels alone fill the GPU (one 1024-thread workgroup
// per channel): the reduce -> finalize -> apply chain of every BatchNorm backward sits on the data-gradient critical path.
template <int V>
__global__ __launch_bounds__(1024) void bn_bwd_reduce_fin_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                 const float* __restrict__ yout,
                                                                 const uint8_t* __restrict__ ymask, int relu,
                                                                 const float* __restrict__ scale,
                                                                 const float* __restrict__ shift,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, int B, int C, int HW,
                                                                 FastDiv dHWv, double count, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta, float* __restrict__ coef) {
    const int c = blockIdx.x;
    const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
    const int hwv = HW / V;
    const bool has_m = V == 4 && ymask != nullptr;
    const bool has_y = yout != nullptr || has_m;
    double s1 = 0, s2 = 0;
    for (int idx = threadIdx.x; idx < B * hwv; idx += 1024) {
        const int n = (int)dHWv.div((uint32_t)idx), i = idx - n * hwv;
        const int64_t off = ((int64_t)n * C + c) * HW + i * V;
        float xv[V], gv[V], yv[V];
        if (V == 4) {
            float4 t = ld4x(x + off); xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
            t = *(const float4*)(dy + off); gv[0] = t.x; gv[1] = t.y; gv[2] = t.z; gv[3] = t.w;
            if (has_m) {
                const uint32_t m = ymask[off >> 2];
                yv[0] = (float)(m & 1u); yv[1] = (float)((m >> 1) & 1u); yv[2] = (float)((m >> 2) & 1u); yv[3] = (float)((m >> 3) & 1u);
            } else if (has_y) { t = *(const float4*)(yout + off); yv[0] = t.x; yv[1] = t.y; yv[2] = t.z; yv[3] = t.w; }
        } else {
            xv[0] = x[off]; gv[0] = dy[off];
            if (has_y) yv[0] = yout[off];
        }
#pragma unroll
        for (int q = 0; q < V; ++q) {
            float g = bn_mask(gv[q], xv[q], has_y ? yv[q] : 0.f, has_y, relu, sc, sh);
            s1 += g;
            s2 += (double)g * ((xv[q] - mu) * is);
        }
    }
    __shared__ double shm[32];
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { shm[w] = s1; shm[16 + w] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0, b = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { a += shm[k]; b += shm[16 + k]; }
        dbeta[c] = (float)a;
        dgamma[c] = (float)b;
        coef[2 * c] = (float)(a / count);
        coef[2 * c + 1] = (float)(b / count);
    }
}

// Reduce AND apply in one launch when a channel's elements fit the registers of its workgroup (B * HW <= 1024 * NIT * V:
// the 14 x 14 and 7 x 7 planes at batch 96): the masked gradient and x are read ONCE, held while the two sums go through
// the same reduction as above (same thread -> element map, same tree: the same bits as reduce + apply), and dx / dres are
// written from the registers — 3 tensor passes instead of 5 and one launch instead of two.
template <int V, int NIT>
__global__ __launch_bounds__(1024) void bn_bwd_onepass_kernel(const float* dy, const float* __restrict__ x,
                                                              const float* __restrict__ yout,
                                                              const uint8_t* __restrict__ ymask, int relu,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ shift,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ invstd,
                                                              const float* __restrict__ gamma, int B, int C, int HW,
                                                              FastDiv dHWv, double count, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, float* __restrict__ coef, float* dx,
                                                              float* dres, int dres_acc) {
    const int c = blockIdx.x;
    const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
    const int hwv = HW / V, nv = B * hwv;
    const bool has_m = V == 4 && ymask != nullptr;
    const bool has_y = yout != nullptr || has_m;
    float gk[NIT][V], xk[NIT][V];
    double s1 = 0, s2 = 0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = threadIdx.x + it * 1024;
        const bool live = idx < nv;
        const int n = live ? (int)dHWv.div((uint32_t)idx) : 0, i = live ? idx - n * hwv : 0;
        const int64_t off = ((int64_t)n * C + c) * HW + i * V;
        float xv[V], gv[V], yv[V];
#pragma unroll
        for (int q = 0; q < V; ++q) { xv[q] = gv[q] = yv[q] = 0.f; }
        if (live) {
            if (V == 4) {
                float4 t = ld4x(x + off); xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
                t = *(const float4*)(dy + off); gv[0] = t.x; gv[1] = t.y; gv[2] = t.z; gv[3] = t.w;
                if (has_m) {
                    const uint32_t m = ymask[off >> 2];
                    yv[0] = (float)(m & 1u); yv[1] = (float)((m >> 1) & 1u); yv[2] = (float)((m >> 2) & 1u); yv[3] = (float)((m >> 3) & 1u);
                } else if (has_y) { t = *(const float4*)(yout + off); yv[0] = t.x; yv[1] = t.y; yv[2] = t.z; yv[3] = t.w; }
            } else {
                xv[0] = x[off]; gv[0] = dy[off];
                if (has_y) yv[0] = yout[off];
            }
        }
#pragma unroll
        for (int q = 0; q < V; ++q) {
            const float g = live ? bn_mask(gv[q], xv[q], has_y ? yv[q] : 0.f, has_y, relu, sc, sh) : 0.f;
            gk[it][q] = g; xk[it][q] = xv[q];
            if (live) {
                s1 += g;
                s2 += (double)g * ((xv[q] - mu) * is);
            }
        }
    }
    __shared__ double shm[34];
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { shm[w] = s1; shm[16 + w] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0, b = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { a += shm[k]; b += shm[16 + k]; }
        dbeta[c] = (float)a;
        dgamma[c] = (float)b;
        const float k1 = (float)(a / count), k2 = (float)(b / count);
        coef[2 * c] = k1;
        coef[2 * c + 1] = k2;
        shm[32] = k1; shm[33] = k2;
    }
    __syncthreads();
    const float k1 = (float)shm[32], k2 = (float)shm[33], gi = gamma[c] * is;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = threadIdx.x + it * 1024;
        if (idx >= nv) break;
        const int n = (int)dHWv.div((uint32_t)idx), i = idx - n * hwv;
        const int64_t off = ((int64_t)n * C + c) * HW + i * V;
        float ov[V], rv[V];
#pragma unroll
        for (int q = 0; q < V; ++q) rv[q] = 0.f;
        if (dres && dres_acc) {
            if (V == 4) { const float4 t = *(const float4*)(dres + off); rv[0] = t.x; rv[1] = t.y; rv[2] = t.z; rv[3] = t.w; }
            else rv[0] = dres[off];
        }
#pragma unroll
        for (int q = 0; q < V; ++q) {
            const float xh = (xk[it][q] - mu) * is;
            ov[q] = gi * (gk[it][q] - k1 - xh * k2);
            rv[q] += gk[it][q];
        }
        if (V == 4) {
            *(float4*)(dx + off) = make_float4(ov[0], ov[1], ov[2], ov[3]);
            if (dres) *(float4*)(dres + off) = make_float4(rv[0], rv[1], rv[2], rv[3]);
        } else {
            dx[off] = ov[0];
            if (dres) dres[off] = rv[0];
        }
    }
}

__global__ void bn_bwd_finalize_kernel(const double* __restrict__ part, int C, int S, double count,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       float* __restrict__ coef) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1 = 0, s2 = 0;
    for (int s = 0; s < S; ++s) {
        s1 += part[((int64_t)c * S + s) * 2];
        s2 += part[((int64_t)c * S + s) * 2 + 1];
    }
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
    coef[2 * c] = (float)(s1 / count);
    coef[2 * c + 1] = (float)(s2 / count);
}

// First half of a BatchNorm backward whose second half is folded into the consumers of dx: the masked gradient g
// replaces dy IN PLACE (it is also the residual branch's gradient), per-channel sums are reduced as above, and the
// finalize emits the three per-channel constants of   dx = ca*g + cb*x + cc   (from dx = gamma*invstd*(g - mean(g)
// - xhat*mean(g*xhat)), xhat = (x - mu)*invstd): ca = gamma*invstd, cb = -ca*invstd*mean(g*xhat), cc = -ca*mean(g) - cb*mu.
__global__ __launch_bounds__(256) void bn_bwd_reduce_g_kernel(float* __restrict__ dy, const float* __restrict__ dy2,
                                                              const float* __restrict__ x,
                                                              const float* __restrict__ yout,
                                                              const uint8_t* __restrict__ ymask, int relu,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ shift,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, int B, int C, int HW,
                                                              int S, FastDiv dHWv, double* __restrict__ part, BnFin f) {
    const int c = blockIdx.x, s = blockIdx.y;
    const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
    const int nimg = (B - s + S - 1) / S, hwv = HW / 4;
    const bool has_m = ymask != nullptr;
    const bool has_y = yout != nullptr || has_m;
    double s1 = 0, s2 = 0;
    for (int idx = threadIdx.x; idx < nimg * hwv; idx += 256) {
        const int nl = (int)dHWv.div((uint32_t)idx), i = idx - nl * hwv;
        const int64_t off = ((int64_t)(s + nl * S) * C + c) * HW + i * 4;
        float xv[4], gv[4], yv[4];
        float4 t = ld4x(x + off); xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
        t = *(const float4*)(dy + off); gv[0] = t.x; gv[1] = t.y; gv[2] = t.z; gv[3] = t.w;
        if (dy2) {      // a second contribution to the incoming gradient, added on the way in
            t = *(const float4*)(dy2 + off); gv[0] += t.x; gv[1] += t.y; gv[2] += t.z; gv[3] += t.w;
        }
        if (has_m) {
            const uint32_t m = ymask[off >> 2];
            yv[0] = (float)(m & 1u); yv[1] = (float)((m >> 1) & 1u); yv[2] = (float)((m >> 2) & 1u); yv[3] = (float)((m >> 3) & 1u);
        } else if (has_y) { t = *(const float4*)(yout + off); yv[0] = t.x; yv[1] = t.y; yv[2] = t.z; yv[3] = t.w; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float g = bn_mask(gv[q], xv[q], has_y ? yv[q] : 0.f, has_y, relu, sc, sh);
            gv[q] = g;
            s1 += g;
            s2 += (double)g * ((xv[q] - mu) * is);
        }
        *(float4*)(dy + off) = make_float4(gv[0], gv[1], gv[2], gv[3]);
    }
    __shared__ double shm[8];
    block_sum2(s1, s2, shm);
    if (f.slots) {
        if (bn_publish(f.slots + (int64_t)c * S, f.claim + c, s, S, f.tag, s1, s2)) bn_bwd_finish(c, C, s1, s2, f);
    } else if (threadIdx.x == 0) {
        part[((int64_t)c * S + s) * 2 + 0] = s1;
        part[((int64_t)c * S + s) * 2 + 1] = s2;
    }
}

__global__ void bn_bwd_finalize3_kernel(const double* __restrict__ part, int C, int S, double count,
                                        const float* __restrict__ gamma, const float* __restrict__ mean,
                                        const float* __restrict__ invstd, float* __restrict__ dgamma,
                                        float* __restrict__ dbeta, float* __restrict__ coef3) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1 = 0, s2 = 0;
    for (int s = 0; s < S; ++s) {
        s1 += part[((int64_t)c * S + s) * 2];
        s2 += part[((int64_t)c * S + s) * 2 + 1];
    }
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
    const float k1 = (float)(s1 / count), k2 = (float)(s2 / count);
    const float ca = gamma[c] * invstd[c], cb = -ca * invstd[c] * k2;
    coef3[c] = ca;
    coef3[C + c] = cb;
    coef3[2 * C + c] = -ca * k1 - cb * mean[c];
}

// The same finish from the sums a data-gradient kernel's epilogue left behind (OutDesc::bnb_part, gemm_engine.h store_tile):
// part[c][G][2] = (sum of g, sum of g * (x - mean[c])) of channel c over column group g, fp32; summed here in fp64, fixed order.
__global__ __launch_bounds__(256) void bn_bwd_partials_fin3_kernel(const float* __restrict__ part, int G, int C, double count,
                                                                   const float* __restrict__ gamma,
                                                                   const float* __restrict__ mean,
                                                                   const float* __restrict__ invstd,
                                                                   float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                   float* __restrict__ coef3) {
    const int c = blockIdx.x;
    const float2* p = (const float2*)part + (int64_t)c * G;
    double s1 = 0, s2 = 0;
    for (int g = threadIdx.x; g < G; g += 256) {
        const float2 v = p[g];
        s1 += (double)v.x;
        s2 += (double)v.y;
    }
    __shared__ double sh[8];
    block_sum2(s1, s2, sh);
    if (threadIdx.x == 0) {
        s2 *= (double)invstd[c];                       // sum of g * xhat
        dbeta[c] = (float)s1;
        dgamma[c] = (float)s2;
        const float k1 = (float)(s1 / count), k2 = (float)(s2 / count);
        const float ca = gamma[c] * invstd[c], cb = -ca * invstd[c] * k2;
        coef3[c] = ca;
        coef3[C + c] = cb;
        coef3[2 * C + c] = -ca * k1 - cb * mean[c];
    }
}

// dx may alias dy, dres may alias dy: every element is read before it is written by the same thread
template <int V>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* dy, const float* __restrict__ x,
                                                           const float* __restrict__ yout,
                                                           const uint8_t* __restrict__ ymask, int relu,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ coef, float* dx,
                                                           float* dres, int dres_acc, int64_t nvec, FastDiv dHWv,
                                                           FastDiv dC, int nt) {
    const bool has_m = V == 4 && ymask != nullptr;
    const bool has_y = yout != nullptr || has_m;
    for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < nvec; e += gridDim.x * 256ll) {
        const uint32_t row = dHWv.div((uint32_t)e);
        const int c = (int)(row - dC.div(row) * dC.d);
        const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
        const float k1 = coef[2 * c], k2 = coef[2 * c + 1], gi = gamma[c] * is;
        float xv[V], gv[V], yv[V], rv[V], ov[V];
        if (V == 4) {
            float4 t = ld4x(x + 4 * e); xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
            t = ((const float4*)dy)[e]; gv[0] = t.x; gv[1] = t.y; gv[2] = t.z; gv[3] = t.w;
            if (has_m) {
                const uint32_t m = ymask[e];
                yv[0] = (float)(m & 1u); yv[1] = (float)((m >> 1) & 1u); yv[2] = (float)((m >> 2) & 1u); yv[3] = (float)((m >> 3) & 1u);
            } else if (has_y) { t = ((const float4*)yout)[e]; yv[0] = t.x; yv[1] = t.y; yv[2] = t.z; yv[3] = t.w; }
            if (dres && dres_acc) { t = ((const float4*)dres)[e]; rv[0] = t.x; rv[1] = t.y; rv[2] = t.z; rv[3] = t.w; }
        } else {
            xv[0] = x[e]; gv[0] = dy[e];
            if (has_y) yv[0] = yout[e];
            if (dres && dres_acc) rv[0] = dres[e];
        }
#pragma unroll
        for (int q = 0; q < V; ++q) {
            float g = bn_mask(gv[q], xv[q], has_y ? yv[q] : 0.f, has_y, relu, sc, sh);
            float xh = (xv[q] - mu) * is;
            ov[q] = gi * (g - k1 - xh * k2);
            rv[q] = ((dres && dres_acc) ? rv[q] : 0.f) + g;
        }
        if (V == 4) {
            st4((float4*)dx + e, make_float4(ov[0], ov[1], ov[2], ov[3]), nt);
            if (dres) ((float4*)dres)[e] = make_float4(rv[0], rv[1], rv[2], rv[3]);
        } else {
            dx[e] = ov[0];
            if (dres) dres[e] = rv[0];
        }
    }
}

// ---------------------------------------------------------------- LayerNorm

__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int rows,
                                                     int dim, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (int64_t)row * dim;
    float s = 0.f;
    for (int i = lane; i < dim; i += 64) s += xr[i];
    const float mu = wave_sum_f(s) / dim;
    float q = 0.f;
    for (int i = lane; i < dim; i += 64) { float d = xr[i] - mu; q += d * d; }
    const float rs = 1.0f / sqrtf(wave_sum_f(q) / dim + eps);
    float* yr = y + (int64_t)row * dim;
    for (int i = lane; i < dim; i += 64) yr[i] = (xr[i] - mu) * rs * gamma[i] + beta[i];
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// dx per row; t = dy * xhat written for the dgamma column sum (t == nullptr: the input gradient only)
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, float* __restrict__ dx,
                                                     float* __restrict__ t, int rows, int dim) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t off = (int64_t)row * dim;
    const float mu = mean[row], rs = rstd[row];
    float a = 0.f, b = 0.f;
    for (int i = lane; i < dim; i += 64) {
        float g = dy[off + i] * gamma[i], xh = (x[off + i] - mu) * rs;
        a += g;
        b += g * xh;
    }
    a = wave_sum_f(a) / dim;
    b = wave_sum_f(b) / dim;
    for (int i = lane; i < dim; i += 64) {
        float d = dy[off + i], xh = (x[off + i] - mu) * rs;
        dx[off + i] = rs * (d * gamma[i] - a - xh * b);
        if (t) t[off + i] = d * xh;
    }
}

// out[j] (+)= sum_i x[i*cols + j]; block = 16 columns x 64 row lanes (many blocks, short loops, 8 loads in
// flight per thread), fixed summation order
__device__ __forceinline__ void colsum_block(const float* __restrict__ x, float* __restrict__ out, int rows, int cols,
                                             int accumulate, int bx) {
    __shared__ float sh[64][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int j = bx * 16 + tx;
    float s = 0.f;
    if (j < cols) {
        int i = ty;
        for (; i + 7 * 64 < rows; i += 8 * 64) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = x[(int64_t)(i + q * 64) * cols + j];
#pragma unroll
            for (int q = 0; q < 8; ++q) s += v[q];
        }
        for (; i < rows; i += 64) s += x[(int64_t)i * cols + j];
    }
    sh[ty][tx] = s;
    __syncthreads();
    for (int half = 32; half > 0; half >>= 1) {
        if (ty < half) sh[ty][tx] += sh[ty + half][tx];
        __syncthreads();
    }
    if (ty == 0 && j < cols) out[j] = (accumulate ? out[j] : 0.f) + sh[0][tx];
}
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, int rows,
                                                      int cols, int accumulate) {
    colsum_block(x, out, rows, cols, accumulate, blockIdx.x);
}
// several independent column sums (the bias gradients of a token mixer's backward, LayerNorm's d-gamma / d-beta pair) as ONE
// launch: blockIdx.y = the job, each job summed exactly as colsum_kernel sums it
constexpr int CSG_MAX = 16;
struct ColsumJobs {
    const float* x[CSG_MAX];
    float* out[CSG_MAX];
    int rows[CSG_MAX], cols[CSG_MAX], acc[CSG_MAX];
};
__global__ __launch_bounds__(1024) void colsum_group_kernel(ColsumJobs J) {
    const int q = blockIdx.y;
    const int cols = J.cols[q];
    if ((int)blockIdx.x * 16 >= cols) return;
    colsum_block(J.x[q], J.out[q], J.rows[q], cols, J.acc[q], blockIdx.x);
}

}  // namespace scat

using namespace scat;

// [C][S] slots of 32 bytes (two-launch form: the first 16 of each pair of doubles) | [C] claim words | [2 C] floats
extern "C" int64_t scat_bn_ws(int B, int C, int HW) {
    if (B <= 0 || C <= 0) return 0;
    return (int64_t)C * bn_splits(B, C) * sizeof(BnSlot) + (int64_t)C * 8 + (int64_t)C * 2 * sizeof(float);
}
static int bn_last_arriver() {     // SCAT_BN_LASTBLOCK=0: the reduce and its finalize as two launches (A/B runs)
    static const int m = [] { const char* e = getenv("SCAT_BN_LASTBLOCK"); return e ? atoi(e) : 1; }();
    return m;
}
static BnFin bn_fin_base(void* ws, int C, int S, double count) {
    BnFin f{};
    if (bn_last_arriver()) {
        f.slots = (BnSlot*)ws;
        f.claim = (unsigned long long*)((char*)ws + (int64_t)C * S * sizeof(BnSlot));
        f.tag = bn_next_tag();
    }
    f.count = count;
    return f;
}
static float* bn_ws_coef(void* ws, int C, int S) { return (float*)((char*)ws + (int64_t)C * S * sizeof(BnSlot) + (int64_t)C * 8); }

extern "C" int scat_bn_train_stats(const float* x, int B, int C, int HW, const float* gamma, const float* beta,
                                   float* running_mean, float* running_var, float momentum, float eps,
                                   float* save_mean, float* save_invstd, float* scale, float* shift, void* ws,
                                   int64_t ws_bytes, void* stream) {
    SCAT_REQUIRE(x && gamma && beta && save_mean && save_invstd && scale && shift, SCAT_E_ARG,
                 "scat_bn_train_stats: null pointer");
    SCAT_REQUIRE(B > 0 && C > 0 && HW > 0, SCAT_E_SHAPE, "scat_bn_train_stats: non-positive dimension");
    SCAT_REQUIRE((running_mean == nullptr) == (running_var == nullptr), SCAT_E_ARG, "scat_bn_train_stats: running pair");
    SCAT_REQUIRE(ws && ws_bytes >= scat_bn_ws(B, C, HW), SCAT_E_WORKSPACE, "scat_bn_train_stats: workspace too small");
    const int S = bn_splits(B, C);
    hipStream_t st = (hipStream_t)stream;
    static const int fused_min_c = diag_env_int("SCAT_BN_FUSED_MIN_C", 256);
    if (C >= fused_min_c) {
        const double count = (double)B * HW;
        if ((HW & 3) == 0 && ((uintptr_t)x & 15) == 0)
            hipLaunchKernelGGL(bn_stats_fin_kernel<4>, dim3(C), dim3(1024), 0, st, x, B, C, HW, FastDiv::make(HW / 4),
                               count, gamma, beta, running_mean, running_var, momentum, eps, save_mean, save_invstd,
                               scale, shift);
        else
            hipLaunchKernelGGL(bn_stats_fin_kernel<1>, dim3(C), dim3(1024), 0, st, x, B, C, HW, FastDiv::make(HW),
                               count, gamma, beta, running_mean, running_var, momentum, eps, save_mean, save_invstd,
                               scale, shift);
        SCAT_LAUNCH_CHECK("scat_bn_train_stats");
        return SCAT_OK;
    }
    BnFin f = bn_fin_base(ws, C, S, (double)B * HW);
    f.gamma = gamma; f.beta = beta; f.rmean = running_mean; f.rvar = running_var; f.momentum = momentum; f.eps = eps;
    f.save_mean = save_mean; f.save_invstd = save_invstd; f.scale = scale; f.shift = shift;
    if ((HW & 3) == 0 && ((uintptr_t)x & 15) == 0)
        hipLaunchKernelGGL(bn_stats_kernel<4>, dim3(C, S), dim3(256), 0, st, x, B, C, HW, S, FastDiv::make(HW / 4),
                           (double*)ws, f);
    else
        hipLaunchKernelGGL(bn_stats_kernel<1>, dim3(C, S), dim3(256), 0, st, x, B, C, HW, S, FastDiv::make(HW),
                           (double*)ws, f);
    if (!f.slots)
        hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 128)), dim3(128), 0, st, (const double*)ws, C, S,
                           (double)B * HW, gamma, beta, running_mean, running_var, momentum, eps, save_mean, save_invstd,
                           scale, shift);
    SCAT_LAUNCH_CHECK("scat_bn_train_stats");
    return SCAT_OK;
}

extern "C" int scat_bn_train_stats_partials(const float* partials, int groups, int B, int C, int HW, const float* gamma,
                                            const float* beta, float* running_mean, float* running_var, float momentum,
                                            float eps, float* save_mean, float* save_invstd, float* scale, float* shift,
                                            void* stream) {
    SCAT_REQUIRE(partials && gamma && beta && save_mean && save_invstd && scale && shift, SCAT_E_ARG,
                 "scat_bn_train_stats_partials: null pointer");
    SCAT_REQUIRE(B > 0 && C > 0 && HW > 0 && groups > 0, SCAT_E_SHAPE, "scat_bn_train_stats_partials: non-positive dimension");
    SCAT_REQUIRE((running_mean == nullptr) == (running_var == nullptr), SCAT_E_ARG,
                 "scat_bn_train_stats_partials: running pair");
    hipLaunchKernelGGL(bn_partials_fin_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, partials, groups,
                       (double)B * HW, (const float*)nullptr, gamma, beta, running_mean, running_var, momentum, eps,
                       save_mean, save_invstd, scale, shift);
    SCAT_LAUNCH_CHECK("scat_bn_train_stats_partials");
    return SCAT_OK;
}

extern "C" int scat_bn_train_stats_partials_shifted(const float* partials, int groups, const float* stat_shift, int B,
                                                    int C, int HW, const float* gamma, const float* beta,
                                                    float* running_mean, float* running_var, float momentum, float eps,
                                                    float* save_mean, float* save_invstd, float* scale, float* shift,
                                                    void* stream) {
    SCAT_REQUIRE(partials && stat_shift && gamma && beta && save_mean && save_invstd && scale && shift, SCAT_E_ARG,
                 "scat_bn_train_stats_partials_shifted: null pointer");
    SCAT_REQUIRE(groups > 0 && B > 0 && C > 0 && HW > 0, SCAT_E_SHAPE,
                 "scat_bn_train_stats_partials_shifted: non-positive dimension");
    SCAT_REQUIRE((running_mean == nullptr) == (running_var == nullptr), SCAT_E_ARG,
                 "scat_bn_train_stats_partials_shifted: running pair");
    SCAT_REQUIRE(stat_shift != save_mean, SCAT_E_ARG,
                 "scat_bn_train_stats_partials_shifted: the reference must not be the buffer the new mean is written to");
    hipLaunchKernelGGL(bn_partials_fin_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, partials, groups,
                       (double)B * HW, stat_shift, gamma, beta, running_mean, running_var, momentum, eps, save_mean,
                       save_invstd, scale, shift);
    SCAT_LAUNCH_CHECK("scat_bn_train_stats_partials_shifted");
    return SCAT_OK;
}

extern "C" int scat_bn_eval_fold(const float* gamma, const float* beta, const float* running_mean,
                                 const float* running_var, float eps, int C, float* scale, float* shift,
                                 void* stream) {
    SCAT_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && C > 0, SCAT_E_ARG,
                 "scat_bn_eval_fold: bad argument");
    hipLaunchKernelGGL(bn_eval_fold_kernel, dim3(cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, gamma, beta,
                       running_mean, running_var, eps, C, scale, shift);
    SCAT_LAUNCH_CHECK("scat_bn_eval_fold");
    return SCAT_OK;
}

extern "C" int scat_bn_apply(const float* x, const float* scale, const float* shift, const float* residual,
                             const float* res_scale, const float* res_shift, int relu, float* y, uint8_t* mask_out,
                             int B, int C, int HW, void* stream) {
    SCAT_REQUIRE(x && scale && shift && y, SCAT_E_ARG, "scat_bn_apply: null pointer");
    SCAT_REQUIRE((res_scale == nullptr) == (res_shift == nullptr) && (!res_scale || residual), SCAT_E_ARG,
                 "scat_bn_apply: residual scale/shift pair");
    SCAT_REQUIRE(B > 0 && C > 0 && HW > 0, SCAT_E_SHAPE, "scat_bn_apply: non-positive dimension");
    const int64_t total = (int64_t)B * C * HW;
    SCAT_REQUIRE(fits_i32(total), SCAT_E_SHAPE, "scat_bn_apply: tensor exceeds 2^31 elements");
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (HW & 3) == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual) & 15) == 0;
    if (vec) {
        const int64_t nv = total / 4;
        hipLaunchKernelGGL(bn_apply_kernel<4>, dim3(ew_grid(nv)), dim3(256), 0, st, x, scale, shift, residual, res_scale,
                           res_shift, relu, y, mask_out, nv, FastDiv::make(HW / 4), FastDiv::make(C), store_policy(total * 4) == 2);
    } else {
        SCAT_REQUIRE(!mask_out, SCAT_E_SHAPE, "scat_bn_apply: the sign mask needs HW % 4 == 0 and 16-B aligned tensors");
        hipLaunchKernelGGL(bn_apply_kernel<1>, dim3(ew_grid(total)), dim3(256), 0, st, x, scale, shift, residual, res_scale,
                           res_shift, relu, y, nullptr, total, FastDiv::make(HW), FastDiv::make(C), 0);
    }
    SCAT_LAUNCH_CHECK("scat_bn_apply");
    return SCAT_OK;
}

extern "C" int scat_bn_bwd(const float* dy, const float* x, const float* y_out, const uint8_t* y_mask, int relu,
                           const float* scale,
                           const float* shift, const float* save_mean, const float* save_invstd, const float* gamma,
                           float* dgamma, float* dbeta, float* dx, float* dres, int dres_accumulate, int B, int C,
                           int HW, void* ws, int64_t ws_bytes, void* stream) {
    SCAT_REQUIRE(dy && x && scale && shift && save_mean && save_invstd && gamma && dgamma && dbeta && dx, SCAT_E_ARG,
                 "scat_bn_bwd: null pointer");
    SCAT_REQUIRE(B > 0 && C > 0 && HW > 0, SCAT_E_SHAPE, "scat_bn_bwd: non-positive dimension");
    SCAT_REQUIRE(ws && ws_bytes >= scat_bn_ws(B, C, HW), SCAT_E_WORKSPACE, "scat_bn_bwd: workspace too small");
    const int S = bn_splits(B, C);
    double* part = (double*)ws;
    float* coef = bn_ws_coef(ws, C, S);
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = (int64_t)B * C * HW;
    SCAT_REQUIRE(fits_i32(total), SCAT_E_SHAPE, "scat_bn_bwd: tensor exceeds 2^31 elements");
    const bool vec = (HW & 3) == 0 &&
                     (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)y_out | (uintptr_t)dx | (uintptr_t)dres) & 15) == 0;
    SCAT_REQUIRE(!y_mask || vec, SCAT_E_SHAPE, "scat_bn_bwd: the sign mask needs HW % 4 == 0 and 16-B aligned tensors");
    SCAT_REQUIRE(!(y_mask && y_out), SCAT_E_ARG, "scat_bn_bwd: pass the output OR its sign mask");
    static const int fused_min_c = diag_env_int("SCAT_BN_FUSED_MIN_C", 256);
    // small planes: one workgroup per channel holds its elements in registers (reduce + apply in one launch)
    static const int onepass = [] { const char* e = getenv("SCAT_BN_ONEPASS"); return e ? atoi(e) : 1; }();
    const int64_t nvc = (int64_t)B * HW / (vec ? 4 : 1);
    if (onepass && C >= 64 && nvc <= 1024 * 8) {
        const FastDiv dv = FastDiv::make(vec ? HW / 4 : HW);
        const double cnt = (double)B * HW;
#define SCAT_BN_ONEPASS(V, NIT)                                                                                              \
        hipLaunchKernelGGL((bn_bwd_onepass_kernel<V, NIT>), dim3(C), dim3(1024), 0, st, dy, x, y_out, vec ? y_mask : nullptr, \
                           relu, scale, shift, save_mean, save_invstd, gamma, B, C, HW, dv, cnt, dgamma, dbeta, coef, dx,     \
                           dres, dres_accumulate)
        if (vec) { if (nvc <= 1024 * 2) SCAT_BN_ONEPASS(4, 2); else if (nvc <= 1024 * 5) SCAT_BN_ONEPASS(4, 5); else SCAT_BN_ONEPASS(4, 8); }
        else { if (nvc <= 1024 * 2) SCAT_BN_ONEPASS(1, 2); else if (nvc <= 1024 * 5) SCAT_BN_ONEPASS(1, 5); else SCAT_BN_ONEPASS(1, 8); }
#undef SCAT_BN_ONEPASS
        SCAT_LAUNCH_CHECK("scat_bn_bwd");
        return SCAT_OK;
    }
    if (C >= fused_min_c) {
        if (vec)
            hipLaunchKernelGGL(bn_bwd_reduce_fin_kernel<4>, dim3(C), dim3(1024), 0, st, dy, x, y_out, y_mask, relu, scale,
                               shift, save_mean, save_invstd, B, C, HW, FastDiv::make(HW / 4), (double)B * HW, dgamma,
                               dbeta, coef);
        else
            hipLaunchKernelGGL(bn_bwd_reduce_fin_kernel<1>, dim3(C), dim3(1024), 0, st, dy, x, y_out, nullptr, relu, scale,
                               shift, save_mean, save_invstd, B, C, HW, FastDiv::make(HW), (double)B * HW, dgamma, dbeta,
                               coef);
    } else {
        BnFin f = bn_fin_base(ws, C, S, (double)B * HW);
        f.dgamma = dgamma; f.dbeta = dbeta; f.coef = coef; f.coef3 = 0;
        if (vec)
            hipLaunchKernelGGL(bn_bwd_reduce_kernel<4>, dim3(C, S), dim3(256), 0, st, dy, x, y_out, y_mask, relu, scale,
                               shift, save_mean, save_invstd, B, C, HW, S, FastDiv::make(HW / 4), part, f);
        else
            hipLaunchKernelGGL(bn_bwd_reduce_kernel<1>, dim3(C, S), dim3(256), 0, st, dy, x, y_out, nullptr, relu, scale,
                               shift, save_mean, save_invstd, B, C, HW, S, FastDiv::make(HW), part, f);
        if (!f.slots)
            hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 128)), dim3(128), 0, st, (const double*)part, C, S,
                               (double)B * HW, dgamma, dbeta, coef);
    }
    if (vec) {
        const int64_t nv = total / 4;
        hipLaunchKernelGGL(bn_bwd_apply_kernel<4>, dim3(ew_grid(nv)), dim3(256), 0, st, dy, x, y_out, y_mask, relu,
                           scale, shift, save_mean, save_invstd, gamma, (const float*)coef, dx, dres, dres_accumulate, nv,
                           FastDiv::make(HW / 4), FastDiv::make(C), store_policy(total * 4) == 2);
    } else {
        hipLaunchKernelGGL(bn_bwd_apply_kernel<1>, dim3(ew_grid(total)), dim3(256), 0, st, dy, x, y_out, nullptr, relu,
                           scale, shift, save_mean, save_invstd, gamma, (const float*)coef, dx, dres, dres_accumulate, total,
                           FastDiv::make(HW), FastDiv::make(C), 0);
    }
    SCAT_LAUNCH_CHECK("scat_bn_bwd");
    return SCAT_OK;
}

// dx[B,C,H,W] = BatchNorm(+ReLU) backward of the gradient that a 3x3 / stride-2 / pad-1 max-pool's backward would scatter from
// dy_pooled[B,C,H/2,W/2] with the arg-max taps idx (scat_maxpool3x3s2_fwd), without materialising that gradient.
extern "C" int scat_bn_bwd_maxpool(const float* dy_pooled, const int8_t* idx, const float* x, int relu, const float* scale,
                                   const float* shift, const float* save_mean, const float* save_invstd,
                                   const float* gamma, float* dgamma, float* dbeta, float* dx, int B, int C, int H, int W,
                                   void* ws, int64_t ws_bytes, void* stream) {
    SCAT_REQUIRE(dy_pooled && idx && x && scale && shift && save_mean && save_invstd && gamma && dgamma && dbeta && dx,
                 SCAT_E_ARG, "scat_bn_bwd_maxpool: null pointer");
    SCAT_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 4 == 0, SCAT_E_SHAPE,
                 "scat_bn_bwd_maxpool: needs even H and W % 4 == 0");
    SCAT_REQUIRE((((uintptr_t)x | (uintptr_t)dx) & 15) == 0, SCAT_E_SHAPE, "scat_bn_bwd_maxpool: 16-B aligned tensors");
    SCAT_REQUIRE(ws && ws_bytes >= scat_bn_ws(B, C, H * W), SCAT_E_WORKSPACE, "scat_bn_bwd_maxpool: workspace too small");
    SCAT_REQUIRE(fits_i32((int64_t)B * C * H * W), SCAT_E_SHAPE, "scat_bn_bwd_maxpool: tensor exceeds 2^31 elements");
    const int S = bn_splits(B, C);
    double* part = (double*)ws;
    float* coef = bn_ws_coef(ws, C, S);
    hipStream_t st = (hipStream_t)stream;
    BnFin f = bn_fin_base(ws, C, S, (double)B * H * W);
    f.dgamma = dgamma; f.dbeta = dbeta; f.coef = coef; f.coef3 = 0;
    hipLaunchKernelGGL(bn_bwd_pool_reduce_kernel, dim3(C, S), dim3(256), 0, st, dy_pooled, idx, x, relu, scale, shift, save_mean,
                       save_invstd, B, C, H, W, S, part, f);
    if (!f.slots)
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 128)), dim3(128), 0, st, (const double*)part, C, S,
                           (double)B * H * W, dgamma, dbeta, coef);
    const int OH = H / 2, OW = W / 2;
    const int64_t total = (int64_t)B * C * OH * (OW / 2);
    hipLaunchKernelGGL(bn_bwd_pool_apply_kernel, dim3(ew_grid(total)), dim3(256), 0, st, dy_pooled, idx, x, relu, scale, shift,
                       save_mean, save_invstd, gamma, (const float*)coef, dx, total, C, W, OH, OW);
    SCAT_LAUNCH_CHECK("scat_bn_bwd_maxpool");
    return SCAT_OK;
}

// g (masked dy) overwrites dy; coef3[3*C] = (ca, cb, cc) with dx = ca*g + cb*x + cc left to the consumers
// (scat_conv1x1_s1_bnb, scat_conv2d_wgrad_bnb).  HW % 4 == 0 and 16-B aligned tensors only.
extern "C" int scat_bn_bwd_pre(float* dy_g, const float* dy_add, const float* x, const float* y_out, const uint8_t* y_mask, int relu,
                               const float* scale, const float* shift, const float* save_mean,
                               const float* save_invstd, const float* gamma, float* dgamma, float* dbeta,
                               float* coef3, int B, int C, int HW, void* ws, int64_t ws_bytes, void* stream) {
    SCAT_REQUIRE(dy_g && x && scale && shift && save_mean && save_invstd && gamma && dgamma && dbeta && coef3, SCAT_E_ARG,
                 "scat_bn_bwd_pre: null pointer");
    SCAT_REQUIRE(B > 0 && C > 0 && HW > 0, SCAT_E_SHAPE, "scat_bn_bwd_pre: non-positive dimension");
    SCAT_REQUIRE(ws && ws_bytes >= scat_bn_ws(B, C, HW), SCAT_E_WORKSPACE, "scat_bn_bwd_pre: workspace too small");
    SCAT_REQUIRE(!(y_mask && y_out), SCAT_E_ARG, "scat_bn_bwd_pre: pass the output OR its sign mask");
    SCAT_REQUIRE((HW & 3) == 0 && (((uintptr_t)dy_g | (uintptr_t)dy_add | (uintptr_t)x | (uintptr_t)y_out) & 15) == 0, SCAT_E_SHAPE,
                 "scat_bn_bwd_pre: needs HW % 4 == 0 and 16-B aligned tensors");
    SCAT_REQUIRE(fits_i32((int64_t)B * C * HW), SCAT_E_SHAPE, "scat_bn_bwd_pre: tensor exceeds 2^31 elements");
    const int S = bn_splits(B, C);
    double* part = (double*)ws;
    hipStream_t st = (hipStream_t)stream;
    BnFin f = bn_fin_base(ws, C, S, (double)B * HW);
    f.dgamma = dgamma; f.dbeta = dbeta; f.coef = coef3; f.coef3 = 1; f.gamma = gamma; f.mean = save_mean; f.invstd = save_invstd;
    hipLaunchKernelGGL(bn_bwd_reduce_g_kernel, dim3(C, S), dim3(256), 0, st, dy_g, dy_add, x, y_out, y_mask, relu, scale, shift,
                       save_mean, save_invstd, B, C, HW, S, FastDiv::make(HW / 4), part, f);
    if (!f.slots)
        hipLaunchKernelGGL(bn_bwd_finalize3_kernel, dim3(cdiv(C, 128)), dim3(128), 0, st, (const double*)part, C, S,
                           (double)B * HW, gamma, save_mean, save_invstd, dgamma, dbeta, coef3);
    SCAT_LAUNCH_CHECK("scat_bn_bwd_pre");
    return SCAT_OK;
}

// coef3 / dgamma / dbeta of scat_bn_bwd_pre from the partial sums a data-gradient epilogue wrote (scat_epilogue_bnb_arm):
// the masked gradient is already in place, this is only the finish
extern "C" int scat_bn_bwd_pre_partials(const float* partials, int groups, int B, int C, int HW, const float* save_mean,
                                        const float* save_invstd, const float* gamma, float* dgamma, float* dbeta,
                                        float* coef3, void* stream) {
    SCAT_REQUIRE(partials && save_mean && save_invstd && gamma && dgamma && dbeta && coef3, SCAT_E_ARG,
                 "scat_bn_bwd_pre_partials: null pointer");
    SCAT_REQUIRE(groups > 0 && B > 0 && C > 0 && HW > 0, SCAT_E_SHAPE, "scat_bn_bwd_pre_partials: non-positive dimension");
    hipLaunchKernelGGL(bn_bwd_partials_fin3_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, partials, groups, C,
                       (double)B * HW, gamma, save_mean, save_invstd, dgamma, dbeta, coef3);
    SCAT_LAUNCH_CHECK("scat_bn_bwd_pre_partials");
    return SCAT_OK;
}

extern "C" int scat_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                  float* rstd, int rows, int dim, float eps, void* stream) {
    SCAT_REQUIRE(x && gamma && beta && y && mean && rstd, SCAT_E_ARG, "scat_layernorm_fwd: null pointer");
    SCAT_REQUIRE(rows > 0 && dim > 0, SCAT_E_SHAPE, "scat_layernorm_fwd: non-positive dimension");
    hipLaunchKernelGGL(ln_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, mean,
                       rstd, rows, dim, eps);
    SCAT_LAUNCH_CHECK("scat_layernorm_fwd");
    return SCAT_OK;
}

// Tall matrices (HRNet's token mixer: 196 608 rows x 196 columns): cols / 16 workgroups cannot pull 150 MB fast enough
// (13 workgroups: 80 us per sum).  Two sums (a and b, same shape) in row slices — grid (cols / 16, S), partial sums
// part[which][slice][col] — and a finish that adds the slices in index order: fixed summation order, no atomics.
template <bool TWO>
__global__ __launch_bounds__(1024) void colsum2_part_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            float* __restrict__ part, int rows, int cols, int rps) {
    __shared__ float sh[2][64][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int j = blockIdx.x * 16 + tx, S = gridDim.y, sl = blockIdx.y;
    const int r0 = sl * rps, r1 = min(r0 + rps, rows);
    float s0 = 0.f, s1 = 0.f;
    if (j < cols) {
        int i = r0 + ty;
        for (; i + 3 * 64 < r1; i += 4 * 64) {
            float u[4], v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                u[q] = a[(int64_t)(i + q * 64) * cols + j];
                v[q] = TWO ? b[(int64_t)(i + q * 64) * cols + j] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) { s0 += u[q]; s1 += v[q]; }
        }
        for (; i < r1; i += 64) { s0 += a[(int64_t)i * cols + j]; if (TWO) s1 += b[(int64_t)i * cols + j]; }
    }
    sh[0][ty][tx] = s0; sh[1][ty][tx] = s1;
    __syncthreads();
    for (int half = 32; half > 0; half >>= 1) {
        if (ty < half) { sh[0][ty][tx] += sh[0][ty + half][tx]; sh[1][ty][tx] += sh[1][ty + half][tx]; }
        __syncthreads();
    }
    if (ty == 0 && j < cols) {
        part[((int64_t)0 * S + sl) * cols + j] = sh[0][0][tx];
        if (TWO) part[((int64_t)1 * S + sl) * cols + j] = sh[1][0][tx];
    }
}
__global__ __launch_bounds__(256) void colsum2_fin_kernel(const float* __restrict__ part, float* __restrict__ oa,
                                                          float* __restrict__ ob, int S, int cols, int accumulate) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= cols) return;
    float s0 = 0.f, s1 = 0.f;
    for (int q = 0; q < S; ++q) {
        s0 += part[(int64_t)q * cols + j];
        if (ob) s1 += part[((int64_t)S + q) * cols + j];
    }
    oa[j] = (accumulate ? oa[j] : 0.f) + s0;
    if (ob) ob[j] = s1;
}
static int colsum2_slices(int rows, int cols) {
    if ((int64_t)rows * cols < (1 << 22)) return 1;            // small: the one-launch-per-sum form
    const int want = cdiv(1024, cdiv(cols, 16));                // ~1024 workgroups
    return max(1, min(want, rows / 256));
}

extern "C" int64_t scat_layernorm_bwd_ws(int rows, int dim) {
    if (!(rows > 0 && dim > 0)) return 0;
    return (int64_t)rows * dim * sizeof(float) + (int64_t)2 * colsum2_slices(rows, dim) * dim * sizeof(float);
}

extern "C" int scat_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean,
                                  const float* rstd, float* dx, float* dgamma, float* dbeta, int rows, int dim,
                                  void* ws, int64_t ws_bytes, void* stream) {
    SCAT_REQUIRE(dy && x && gamma && mean && rstd && dx && (!dgamma == !dbeta), SCAT_E_ARG,
                 "scat_layernorm_bwd: null pointer");
    SCAT_REQUIRE(rows > 0 && dim > 0, SCAT_E_SHAPE, "scat_layernorm_bwd: non-positive dimension");
    hipStream_t st = (hipStream_t)stream;
    if (!dgamma) {      // the input gradient only (the pose-length term's replay of the tape): no parameter sums, no workspace
        hipLaunchKernelGGL(ln_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, dy, x, gamma, mean, rstd, dx, (float*)nullptr,
                           rows, dim);
        SCAT_LAUNCH_CHECK("scat_layernorm_bwd");
        return SCAT_OK;
    }
    SCAT_REQUIRE(ws && ws_bytes >= scat_layernorm_bwd_ws(rows, dim), SCAT_E_WORKSPACE,
                 "scat_layernorm_bwd: workspace too small");
    float* t = (float*)ws;
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, dy, x, gamma, mean, rstd, dx, t, rows,
                       dim);
    const int S = colsum2_slices(rows, dim);
    if (S > 1) {
        float* part = t + (int64_t)rows * dim;
        hipLaunchKernelGGL(colsum2_part_kernel<true>, dim3(cdiv(dim, 16), S), dim3(1024), 0, st, (const float*)t, dy, part, rows,
                           dim, cdiv(rows, S));
        hipLaunchKernelGGL(colsum2_fin_kernel, dim3(cdiv(dim, 256)), dim3(256), 0, st, (const float*)part, dgamma, dbeta, S, dim, 0);
    } else {
        ColsumJobs J = {};
        J.x[0] = t; J.out[0] = dgamma; J.x[1] = dy; J.out[1] = dbeta;
        J.rows[0] = J.rows[1] = rows; J.cols[0] = J.cols[1] = dim;
        hipLaunchKernelGGL(colsum_group_kernel, dim3(cdiv(dim, 16), 2), dim3(1024), 0, st, J);
    }
    SCAT_LAUNCH_CHECK("scat_layernorm_bwd");
    return SCAT_OK;
}

extern "C" int scat_colsum(const float* x, float* out, int rows, int cols, int accumulate, void* stream);
extern "C" int64_t scat_colsum_ws(int rows, int cols) {
    const int S = rows > 0 && cols > 0 ? colsum2_slices(rows, cols) : 1;
    return S > 1 ? (int64_t)S * cols * sizeof(float) : 0;
}

// the same sum with a caller-owned scratch of scat_colsum_ws(rows, cols) bytes: tall matrices are summed in row slices by
// ~1024 workgroups (scat_colsum alone: cols / 16 workgroups), slices added in index order
extern "C" int scat_colsum_sliced(const float* x, float* out, int rows, int cols, int accumulate, void* ws, int64_t ws_bytes,
                                  void* stream) {
    SCAT_REQUIRE(x && out && rows > 0 && cols > 0, SCAT_E_ARG, "scat_colsum_sliced: bad argument");
    const int S = colsum2_slices(rows, cols);
    if (S <= 1) return scat_colsum(x, out, rows, cols, accumulate, stream);
    SCAT_REQUIRE(ws && ws_bytes >= scat_colsum_ws(rows, cols), SCAT_E_WORKSPACE, "scat_colsum_sliced: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum2_part_kernel<false>, dim3(cdiv(cols, 16), S), dim3(1024), 0, st, x, (const float*)nullptr,
                       (float*)ws, rows, cols, cdiv(rows, S));
    hipLaunchKernelGGL(colsum2_fin_kernel, dim3(cdiv(cols, 256)), dim3(256), 0, st, (const float*)ws, out, (float*)nullptr, S,
                       cols, accumulate);
    SCAT_LAUNCH_CHECK("scat_colsum_sliced");
    return SCAT_OK;
}

extern "C" int scat_colsum(const float* x, float* out, int rows, int cols, int accumulate, void* stream) {
    SCAT_REQUIRE(x && out && rows > 0 && cols > 0, SCAT_E_ARG, "scat_colsum: bad argument");
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(cols, 16)), dim3(1024), 0, (hipStream_t)stream, x, out, rows, cols,
                       accumulate);
    SCAT_LAUNCH_CHECK("scat_colsum");
    return SCAT_OK;
}

// n <= 16 independent column sums, one launch; each is summed in scat_colsum's order (bit-identical to n calls)
extern "C" int scat_colsum_group(const ScatColsumJob* jobs, int n, void* stream) {
    SCAT_REQUIRE(jobs && n > 0 && n <= CSG_MAX, SCAT_E_ARG, "scat_colsum_group: 1..16 jobs");
    ColsumJobs J = {};
    int maxc = 0;
    for (int q = 0; q < n; ++q) {
        SCAT_REQUIRE(jobs[q].x && jobs[q].out && jobs[q].rows > 0 && jobs[q].cols > 0, SCAT_E_ARG, "scat_colsum_group: bad job");
        J.x[q] = jobs[q].x; J.out[q] = jobs[q].out; J.rows[q] = jobs[q].rows; J.cols[q] = jobs[q].cols;
        J.acc[q] = jobs[q].accumulate;
        maxc = max(maxc, jobs[q].cols);
    }
    hipLaunchKernelGGL(colsum_group_kernel, dim3(cdiv(maxc, 16), n), dim3(1024), 0, (hipStream_t)stream, J);
    SCAT_LAUNCH_CHECK("scat_colsum_group");
    return SCAT_OK;
}

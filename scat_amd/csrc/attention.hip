// Softmax attention core for short sequences (n <= 128 tokens, head dim 64): one workgroup
// per (batch, head) keeps Q, K, V and the n x n score matrix in LDS (160 KB/CU) — no score
// matrix ever reaches HBM except the saved probabilities the backward needs.
// Reference: models/vision_transformer.py:61-76 (scale dim_head^-0.5), models/vit.py:51-66
// (scale dim^-0.5); 'b n (h d) -> b h n d' layout handled by indexing, no permute copies.
#include "common.h"

namespace scat {

constexpr int HD = 64;        // head dim
constexpr int HS = HD + 1;    // padded LDS row
typedef float f32x16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                       float* __restrict__ attn, int n, int heads, float scale) {
    extern __shared__ __align__(16) float smem[];
    float* Qs = smem;                 // [n][HS]   (later V as [n][HD])
    float* Ks = Qs + n * HS;          // [n][HS]
    float* Ps = Ks + n * HS;          // [n][n+1]
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int inner = heads * HD, ld = 3 * inner;
    const float* base = qkv + (int64_t)b * n * ld + h * HD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int PS = n + 1;

    for (int e = tid; e < n * HD; e += 256) {
        int i = e >> 6, d = e & 63;
        Qs[i * HS + d] = base[(int64_t)i * ld + d];
        Ks[i * HS + d] = base[(int64_t)i * ld + inner + d];
    }
    __syncthreads();
    for (int p = tid; p < n * n; p += 256) {
        int i = p / n, j = p - i * n;
        float s = 0.f;
#pragma unroll 16
        for (int d = 0; d < HD; ++d) s = fmaf(Qs[i * HS + d], Ks[j * HS + d], s);
        Ps[i * PS + j] = s * scale;
    }
    __syncthreads();
    for (int i = wave; i < n; i += 4) {
        float v0 = lane < n ? Ps[i * PS + lane] : -INFINITY;
        float v1 = lane + 64 < n ? Ps[i * PS + lane + 64] : -INFINITY;
        float m = wmax(fmaxf(v0, v1));
        float e0 = lane < n ? expf(v0 - m) : 0.f;
        float e1 = lane + 64 < n ? expf(v1 - m) : 0.f;
        float inv = 1.0f / wsum(e0 + e1);
        float* arow = attn + (((int64_t)b * heads + h) * n + i) * n;
        if (lane < n) { Ps[i * PS + lane] = e0 * inv; arow[lane] = e0 * inv; }
        if (lane + 64 < n) { Ps[i * PS + lane + 64] = e1 * inv; arow[lane + 64] = e1 * inv; }
    }
    __syncthreads();
    float* Vs = Qs;   // Q,K are dead: [n][HD] fits in their space
    for (int e = tid; e < n * HD; e += 256) {
        int i = e >> 6, d = e & 63;
        Vs[i * HD + d] = base[(int64_t)i * ld + 2 * inner + d];
    }
    __syncthreads();
    for (int e = tid; e < n * HD; e += 256) {
        int i = e >> 6, d = e & 63;
        float s = 0.f;
        for (int j = 0; j < n; ++j) s = fmaf(Ps[i * PS + j], Vs[j * HD + d], s);
        out[((int64_t)b * n + i) * inner + h * HD + d] = s;
    }
}

__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ qkv,
                                                       const float* __restrict__ attn, float* __restrict__ dqkv,
                                                       int n, int heads, float scale) {
    extern __shared__ __align__(16) float smem[];
    float* Ds = smem;                 // [n][n+1]  P, then dS
    float* Xa = Ds + n * (n + 1);     // [n][HS]   dO, then K
    float* Xb = Xa + n * HS;          // [n][HS]   V,  then Q
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int inner = heads * HD, ld = 3 * inner;
    const float* base = qkv + (int64_t)b * n * ld + h * HD;
    float* dbase = dqkv + (int64_t)b * n * ld + h * HD;
    const float* arow0 = attn + ((int64_t)b * heads + h) * n * n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int PS = n + 1;

    for (int p = tid; p < n * n; p += 256) {
        int i = p / n, j = p - i * n;
        Ds[i * PS + j] = arow0[p];
    }
    for (int e = tid; e < n * HD; e += 256) {
        int i = e >> 6, d = e & 63;
        Xa[i * HS + d] = dout[((int64_t)b * n + i) * inner + h * HD + d];
        Xb[i * HS + d] = base[(int64_t)i * ld + 2 * inner + d];
    }
    __syncthreads();
    // dV = P^T dO
    for (int e = tid; e < n * HD; e += 256) {
        int j = e >> 6, d = e & 63;
        float s = 0.f;
        for (int i = 0; i < n; ++i) s = fmaf(Ds[i * PS + j], Xa[i * HS + d], s);
        dbase[(int64_t)j * ld + 2 * inner + d] = s;
    }
    __syncthreads();
    // dS = P * (dP - rowsum(P*dP)) * scale, dP = dO V^T
    for (int i = wave; i < n; i += 4) {
        float dp0 = 0.f, dp1 = 0.f;
        if (lane < n) {
#pragma unroll 16
            for (int d = 0; d < HD; ++d) dp0 = fmaf(Xa[i * HS + d], Xb[lane * HS + d], dp0);
        }
        if (lane + 64 < n) {
#pragma unroll 16
            for (int d = 0; d < HD; ++d) dp1 = fmaf(Xa[i * HS + d], Xb[(lane + 64) * HS + d], dp1);
        }
        float p0 = lane < n ? Ds[i * PS + lane] : 0.f;
        float p1 = lane + 64 < n ? Ds[i * PS + lane + 64] : 0.f;
        float dot = wsum(p0 * dp0 + p1 * dp1);
        if (lane < n) Ds[i * PS + lane] = p0 * (dp0 - dot) * scale;
        if (lane + 64 < n) Ds[i * PS + lane + 64] = p1 * (dp1 - dot) * scale;
    }
    __syncthreads();
    for (int e = tid; e < n * HD; e += 256) {
        int i = e >> 6, d = e & 63;
        Xa[i * HS + d] = base[(int64_t)i * ld + inner + d];   // K
        Xb[i * HS + d] = base[(int64_t)i * ld + d];           // Q
    }
    __syncthreads();
    for (int e = tid; e < n * HD; e += 256) {
        int i = e >> 6, d = e & 63;
        float sq = 0.f, sk = 0.f;
        for (int j = 0; j < n; ++j) {
            sq = fmaf(Ds[i * PS + j], Xa[j * HS + d], sq);   // dQ[i] = sum_j dS[i][j] K[j]
            sk = fmaf(Ds[j * PS + i], Xb[j * HS + d], sk);   // dK[i] = sum_j dS[j][i] Q[j]
        }
        dbase[(int64_t)i * ld + d] = sq;
        dbase[(int64_t)i * ld + inner + d] = sk;
    }
}

static int set_lds(const void* fn, size_t bytes) {
    if (bytes > 64 * 1024)
        return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess ? 0 : -1;
    return 0;
}

// ---------------------------------------------------------------- the same core on the matrix pipe, n = 64 / 96 / 128
//
// models/vit.py:51-66 on HRNet's 128 tokens (hand_net.py:161): per (batch, head) 128 x 128 scores — 4.2 MFLOP forward,
// 10.5 backward, 768 heads per batch-96 step and layer: the lane-per-output loops above take 332 + 619 us per layer at
// this size (profiles/r03_hrnet_kernel_summary_serialized.txt).  Here every product is v_mfma_f32_32x32x2_f32 (exact fp32
// fma chains, like the loops it replaces) and the softmax never leaves the registers:
//   * a wavefront owns 32 queries i and ALL keys j.  It forms the TRANSPOSED score tiles S^T[j][i] = K . Q^T (A = K rows
//     from LDS, B = Q rows from LDS): in the 32x32 accumulator layout the column sits on the lane, so lane (i, half h)
//     holds S[i][j] for j = 32 t + (r & 3) + 8 (r >> 2) + 4 h over its registers r and tiles t — a query's whole row lives
//     in ONE lane pair, and row max / row sum are register loops plus one exchange with the partner half (no LDS, no
//     cross-lane tree; cdna_hip_programming.md T12's "swapped QK^T").
//   * those registers ARE the A operand of the next product: v_mfma_f32_32x32x2_f32 wants A[i = lane & 31][k = lane >> 5]
//     and the order of k is free, so register r of tile t is the k-pair (j, j + 4) and the B operand (V, or K for dQ) is
//     read from LDS at exactly those two rows.  P / dS never go through LDS for O = P V and dQ = dS K.
//   * the products that sum over queries (dV = P^T dO, dK = dS^T Q) read P / dS from an LDS image [i][j] written once.
// LDS rows of Q / K / V / dO are padded to 65 floats, score rows to n + 1: every ds_read_b32 below is conflict-free within
// its 32-lane half.
template <int NTI>     // n / 32
__global__ __launch_bounds__(256) void attn_mfma_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                            float* __restrict__ attn, int heads, float scale) {
    constexpr int n = 32 * NTI;
    extern __shared__ __align__(16) float smem[];
    float* Qs = smem;                 // [n][HS]
    float* Ks = Qs + n * HS;          // [n][HS]
    float* Vs = Ks + n * HS;          // [n][HS]
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int inner = heads * HD, ld = 3 * inner;
    const float* base = qkv + (int64_t)b * n * ld + h * HD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (int e = tid; e < n * 16; e += 256) {                // 16 float4 per 64-float row
        const int i = e >> 4, c = (e & 15) * 4;
        const f4 q = *(const f4*)(base + (int64_t)i * ld + c);
        const f4 k = *(const f4*)(base + (int64_t)i * ld + inner + c);
        const f4 v = *(const f4*)(base + (int64_t)i * ld + 2 * inner + c);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            Qs[i * HS + c + u] = q[u];
            Ks[i * HS + c + u] = k[u];
            Vs[i * HS + c + u] = v[u];
        }
    }
    __syncthreads();
    for (int it = wave; it < NTI; it += 4) {                 // query strips of this wavefront
        f32x16v acc[NTI];
#pragma unroll
        for (int t = 0; t < NTI; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        const float* qrow = Qs + (32 * it + l31) * HS + lh;
        const float* krow = Ks + l31 * HS + lh;
#pragma unroll 4
        for (int k0 = 0; k0 < HD; k0 += 2) {
            const float bq = qrow[k0];
#pragma unroll
            for (int t = 0; t < NTI; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(krow[32 * t * HS + k0], bq, acc[t], 0, 0, 0);
        }
        // softmax over j of scale * S[i][j], i = this lane's query
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < NTI; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[t][r] *= scale;
                m = fmaxf(m, acc[t][r]);
            }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NTI; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[t][r] = expf(acc[t][r] - m);
                sum += acc[t][r];
            }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        float* arow = attn + (((int64_t)b * heads + h) * n + 32 * it + l31) * n + 4 * lh;
#pragma unroll
        for (int t = 0; t < NTI; ++t)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                f4 pv;
#pragma unroll
                for (int u = 0; u < 4; ++u) pv[u] = acc[t][4 * q4 + u] = acc[t][4 * q4 + u] * inv;
                *(f4*)(arow + 32 * t + 8 * q4) = pv;         // j = 32 t + 8 q4 + 4 h + u
            }
        // O[i][d] = sum_j P[i][j] V[j][d]: register r of tile t is the k-pair (j, j + 4)
        f32x16v o[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
        for (int t = 0; t < NTI; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* vrow = Vs + (32 * t + (r & 3) + 8 * (r >> 2) + 4 * lh) * HS + l31;
                o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[t][r], vrow[0], o[0], 0, 0, 0);
                o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[t][r], vrow[32], o[1], 0, 0, 0);
            }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = 32 * it + (r & 3) + 8 * (r >> 2) + 4 * lh;
                out[((int64_t)b * n + i) * inner + h * HD + 32 * dt + l31] = o[dt][r];
            }
    }
}

template <int NTI>
__global__ __launch_bounds__(256) void attn_mfma_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ qkv,
                                                            const float* __restrict__ attn, float* __restrict__ dqkv,
                                                            int heads, float scale) {
    constexpr int n = 32 * NTI, PS = n + 1;
    extern __shared__ __align__(16) float smem[];
    float* Ds = smem;                 // [n][n+1]  P, then dS
    float* Xa = Ds + n * PS;          // [n][HS]   dO, then K
    float* Xb = Xa + n * HS;          // [n][HS]   V,  then Q
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int inner = heads * HD, ld = 3 * inner;
    const float* base = qkv + (int64_t)b * n * ld + h * HD;
    float* dbase = dqkv + (int64_t)b * n * ld + h * HD;
    const float* arow0 = attn + ((int64_t)b * heads + h) * n * n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    typedef float f4 __attribute__((ext_vector_type(4)));
    static_assert(NTI <= 4, "one query strip per wavefront");
    const bool live = wave < NTI;                             // (n = 64 / 96: the last wavefronts only help with the copies)
    const int it = live ? wave : 0;

    for (int e = tid; e < n * 16; e += 256) {
        const int i = e >> 4, c = (e & 15) * 4;
        const f4 g = *(const f4*)(dout + ((int64_t)b * n + i) * inner + h * HD + c);
        const f4 v = *(const f4*)(base + (int64_t)i * ld + 2 * inner + c);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            Xa[i * HS + c + u] = g[u];
            Xb[i * HS + c + u] = v[u];
        }
    }
    // P of this wavefront's queries, in the transposed accumulator layout (lane = query), and into the LDS image [i][j]
    f32x16v p[NTI];
    if (live) {
        const float* arow = arow0 + (int64_t)(32 * it + l31) * n + 4 * lh;
#pragma unroll
        for (int t = 0; t < NTI; ++t)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f4 pv = *(const f4*)(arow + 32 * t + 8 * q4);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    p[t][4 * q4 + u] = pv[u];
                    Ds[(32 * it + l31) * PS + 32 * t + 8 * q4 + 4 * lh + u] = pv[u];
                }
            }
    }
    __syncthreads();
    f32x16v ds[NTI];
    if (live) {
        // dP^T[j][i] = sum_d V[j][d] dO[i][d]
#pragma unroll
        for (int t = 0; t < NTI; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) ds[t][r] = 0.f;
        const float* grow = Xa + (32 * it + l31) * HS + lh;
        const float* vrow = Xb + l31 * HS + lh;
#pragma unroll 4
        for (int k0 = 0; k0 < HD; k0 += 2) {
            const float bg = grow[k0];
#pragma unroll
            for (int t = 0; t < NTI; ++t)
                ds[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[32 * t * HS + k0], bg, ds[t], 0, 0, 0);
        }
        float dot = 0.f;
#pragma unroll
        for (int t = 0; t < NTI; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) dot = fmaf(p[t][r], ds[t][r], dot);
        dot += __shfl_xor(dot, 32, 64);
#pragma unroll
        for (int t = 0; t < NTI; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) ds[t][r] = p[t][r] * (ds[t][r] - dot) * scale;
        // dV[j][d] = sum_i P[i][j] dO[i][d], j = this wavefront's 32 keys
        f32x16v o[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
        const float* pcol = Ds + lh * PS + 32 * it + l31;
        const float* gcol = Xa + lh * HS + l31;
#pragma unroll 4
        for (int i0 = 0; i0 < n; i0 += 2) {
            const float a = pcol[i0 * PS];
            o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, gcol[i0 * HS], o[0], 0, 0, 0);
            o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, gcol[i0 * HS + 32], o[1], 0, 0, 0);
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = 32 * it + (r & 3) + 8 * (r >> 2) + 4 * lh;
                dbase[(int64_t)j * ld + 2 * inner + 32 * dt + l31] = o[dt][r];
            }
    }
    __syncthreads();                  // P, dO and V have been read by everyone
    if (live) {
#pragma unroll
        for (int t = 0; t < NTI; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Ds[(32 * it + l31) * PS + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * lh] = ds[t][r];
    }
    for (int e = tid; e < n * 16; e += 256) {
        const int i = e >> 4, c = (e & 15) * 4;
        const f4 k = *(const f4*)(base + (int64_t)i * ld + inner + c);
        const f4 q = *(const f4*)(base + (int64_t)i * ld + c);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            Xa[i * HS + c + u] = k[u];
            Xb[i * HS + c + u] = q[u];
        }
    }
    __syncthreads();
    if (live) {
        // dQ[i][d] = sum_j dS[i][j] K[j][d] (A from the registers); dK[j][d] = sum_i dS[i][j] Q[i][d] (A from the LDS image)
        f32x16v oq[2], ok[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oq[dt][r] = ok[dt][r] = 0.f;
#pragma unroll
        for (int t = 0; t < NTI; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* krow = Xa + (32 * t + (r & 3) + 8 * (r >> 2) + 4 * lh) * HS + l31;
                oq[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds[t][r], krow[0], oq[0], 0, 0, 0);
                oq[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds[t][r], krow[32], oq[1], 0, 0, 0);
            }
        const float* dcol = Ds + lh * PS + 32 * it + l31;
        const float* qcol = Xb + lh * HS + l31;
#pragma unroll 4
        for (int i0 = 0; i0 < n; i0 += 2) {
            const float a = dcol[i0 * PS];
            ok[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, qcol[i0 * HS], ok[0], 0, 0, 0);
            ok[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, qcol[i0 * HS + 32], ok[1], 0, 0, 0);
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = 32 * it + (r & 3) + 8 * (r >> 2) + 4 * lh;
                dbase[(int64_t)i * ld + 32 * dt + l31] = oq[dt][r];
                dbase[(int64_t)i * ld + inner + 32 * dt + l31] = ok[dt][r];
            }
    }
}

template <int NTI>
static int launch_attn_mfma(bool bwd, const float* a0, const float* a1, const float* a2, float* o0, float* o1, int B, int heads,
                            float scale, hipStream_t st) {
    constexpr int n = 32 * NTI;
    if (!bwd) {
        const size_t lds = (size_t)3 * n * HS * sizeof(float);
        if (set_lds((const void*)attn_mfma_fwd_kernel<NTI>, lds) != 0) return -1;
        hipLaunchKernelGGL((attn_mfma_fwd_kernel<NTI>), dim3(B * heads), dim3(256), lds, st, a0, o0, o1, heads, scale);
    } else {
        const size_t lds = (size_t)(n * (n + 1) + 2 * n * HS) * sizeof(float);
        if (set_lds((const void*)attn_mfma_bwd_kernel<NTI>, lds) != 0) return -1;
        hipLaunchKernelGGL((attn_mfma_bwd_kernel<NTI>), dim3(B * heads), dim3(256), lds, st, a0, a1, a2, o0, heads, scale);
    }
    return 0;
}

}  // namespace scat

using namespace scat;

extern "C" int scat_attention_fwd(const float* qkv, float* out, float* attn, int B, int n, int heads, int dim_head,
                                  float scale, void* stream) {
    SCAT_REQUIRE(qkv && out && attn, SCAT_E_ARG, "scat_attention_fwd: null pointer");
    SCAT_REQUIRE(dim_head == HD, SCAT_E_SHAPE, "scat_attention_fwd: dim_head must be 64 (got %d)", dim_head);
    SCAT_REQUIRE(B > 0 && heads > 0 && n > 0 && n <= 128, SCAT_E_SHAPE, "scat_attention_fwd: need 1 <= n <= 128");
    if (n >= 64 && n % 32 == 0 && (((uintptr_t)qkv | (uintptr_t)attn) & 15) == 0 && heads * HD % 4 == 0) {
        // 64 / 96 / 128 tokens: the matrix-pipe form (same fp32 products, softmax in registers)
        int rc = n == 64 ? launch_attn_mfma<2>(false, qkv, nullptr, nullptr, out, attn, B, heads, scale, (hipStream_t)stream)
               : n == 96 ? launch_attn_mfma<3>(false, qkv, nullptr, nullptr, out, attn, B, heads, scale, (hipStream_t)stream)
                         : launch_attn_mfma<4>(false, qkv, nullptr, nullptr, out, attn, B, heads, scale, (hipStream_t)stream);
        SCAT_REQUIRE(rc == 0, SCAT_E_LAUNCH, "scat_attention_fwd: LDS attribute");
        SCAT_LAUNCH_CHECK("scat_attention_fwd");
        return SCAT_OK;
    }
    size_t lds = (size_t)(2 * n * HS + n * (n + 1)) * sizeof(float);
    SCAT_REQUIRE(set_lds((const void*)attn_fwd_kernel, lds) == 0, SCAT_E_LAUNCH, "scat_attention_fwd: LDS attribute");
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(B * heads), dim3(256), lds, (hipStream_t)stream, qkv, out, attn, n, heads,
                       scale);
    SCAT_LAUNCH_CHECK("scat_attention_fwd");
    return SCAT_OK;
}

extern "C" int scat_attention_bwd(const float* dout, const float* qkv, const float* attn, float* dqkv, int B, int n,
                                  int heads, int dim_head, float scale, void* stream) {
    SCAT_REQUIRE(dout && qkv && attn && dqkv, SCAT_E_ARG, "scat_attention_bwd: null pointer");
    SCAT_REQUIRE(dim_head == HD, SCAT_E_SHAPE, "scat_attention_bwd: dim_head must be 64 (got %d)", dim_head);
    SCAT_REQUIRE(B > 0 && heads > 0 && n > 0 && n <= 128, SCAT_E_SHAPE, "scat_attention_bwd: need 1 <= n <= 128");
    if (n >= 64 && n % 32 == 0 && heads * HD % 4 == 0 &&
        (((uintptr_t)qkv | (uintptr_t)attn | (uintptr_t)dout) & 15) == 0) {
        hipStream_t st = (hipStream_t)stream;
        int rc = n == 64 ? launch_attn_mfma<2>(true, dout, qkv, attn, dqkv, nullptr, B, heads, scale, st)
               : n == 96 ? launch_attn_mfma<3>(true, dout, qkv, attn, dqkv, nullptr, B, heads, scale, st)
                         : launch_attn_mfma<4>(true, dout, qkv, attn, dqkv, nullptr, B, heads, scale, st);
        SCAT_REQUIRE(rc == 0, SCAT_E_LAUNCH, "scat_attention_bwd: LDS attribute");
        SCAT_LAUNCH_CHECK("scat_attention_bwd");
        return SCAT_OK;
    }
    size_t lds = (size_t)(2 * n * HS + n * (n + 1)) * sizeof(float);
    SCAT_REQUIRE(set_lds((const void*)attn_bwd_kernel, lds) == 0, SCAT_E_LAUNCH, "scat_attention_bwd: LDS attribute");
    hipLaunchKernelGGL(attn_bwd_kernel, dim3(B * heads), dim3(256), lds, (hipStream_t)stream, dout, qkv, attn, dqkv, n,
                       heads, scale);
    SCAT_LAUNCH_CHECK("scat_attention_bwd");
    return SCAT_OK;
}

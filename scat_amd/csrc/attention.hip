// Softmax attention core for short sequences (n <= 128 tokens, head dim 64): one workgroup
// per (batch, head) keeps Q, K, V and the n x n score matrix in LDS (160 KB/CU) — no score
// matrix ever reaches HBM except the saved probabilities the backward needs.
// Reference: models/vision_transformer.py:61-76 (scale dim_head^-0.5), models/vit.py:51-66
// (scale dim^-0.5); 'b n (h d) -> b h n d' layout handled by indexing, no permute copies.
#include "common.h"

namespace scat {

constexpr int HD = 64;        // head dim
constexpr int HS = HD + 1;    // padded LDS row

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

}  // namespace scat

using namespace scat;

extern "C" int scat_attention_fwd(const float* qkv, float* out, float* attn, int B, int n, int heads, int dim_head,
                                  float scale, void* stream) {
    SCAT_REQUIRE(qkv && out && attn, SCAT_E_ARG, "scat_attention_fwd: null pointer");
    SCAT_REQUIRE(dim_head == HD, SCAT_E_SHAPE, "scat_attention_fwd: dim_head must be 64 (got %d)", dim_head);
    SCAT_REQUIRE(B > 0 && heads > 0 && n > 0 && n <= 128, SCAT_E_SHAPE, "scat_attention_fwd: need 1 <= n <= 128");
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
    size_t lds = (size_t)(2 * n * HS + n * (n + 1)) * sizeof(float);
    SCAT_REQUIRE(set_lds((const void*)attn_bwd_kernel, lds) == 0, SCAT_E_LAUNCH, "scat_attention_bwd: LDS attribute");
    hipLaunchKernelGGL(attn_bwd_kernel, dim3(B * heads), dim3(256), lds, (hipStream_t)stream, dout, qkv, attn, dqkv, n,
                       heads, scale);
    SCAT_LAUNCH_CHECK("scat_attention_bwd");
    return SCAT_OK;
}

// Strided fp32 GEMM on the MFMA engine: nn.Linear forward / input-gradient / weight-gradient
// (models/vision_transformer.py:33-35,55-57; models/resnet.py:116; models/hand_net.py:353).
#include "gemm_engine.h"

namespace scat {

struct GemmPlan {
    int bm, splits;
};

static GemmPlan gemm_plan(int M, int N, int K) {
    GemmPlan p;
    int64_t t128 = (int64_t)cdiv(M, 128) * cdiv(N, 64);
    p.bm = (M > 64 && t128 >= 256) ? 128 : 64;
    int tiles = cdiv(M, p.bm) * cdiv(N, 64);
    int s = 1;
    if (tiles < 256) {
        s = cdiv(512, tiles);
        int smax = K / 256 > 0 ? K / 256 : 1;
        if (s > smax) s = smax;
        if (s > 64) s = 64;
    }
    p.splits = s < 1 ? 1 : s;
    return p;
}

template <bool AK, bool BK_, int AV, int BV>
static void gemm_dispatch(const GemmPlan& p, const MatDesc& da, const MatDesc& db, const OutDesc& dc, int M, int N,
                          int K, hipStream_t st) {
    set_kernel_label("gemm_%c%d%c%d_%dx64x16_split%d", AK ? 'k' : 'i', AV, BK_ ? 'k' : 'j', BV, p.bm, p.splits);
    if (p.bm == 128)
        launch_gemm<MatLoader<128, 16, AK, AV>, MatLoader<64, 16, BK_, BV>, 128, 64, 16, 2, 2>(da, db, dc, M, N, K,
                                                                                                p.splits, st);
    else
        launch_gemm<MatLoader<64, 16, AK, AV>, MatLoader<64, 16, BK_, BV>, 64, 64, 16, 2, 2>(da, db, dc, M, N, K,
                                                                                              p.splits, st);
}

__global__ void splitk_reduce_bias_kernel(const float* __restrict__ slab, float* __restrict__ c, int64_t c_si,
                                          int64_t c_sj, int M, int N, int splits, const float* __restrict__ bias,
                                          int bias_mode, int accumulate) {
    int64_t n = (int64_t)M * N;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        int i = e / N, j = e % N;
        float* dst = c + i * c_si + j * c_sj;
        float s = accumulate ? *dst : 0.f;
        for (int z = 0; z < splits; ++z) s += slab[(int64_t)z * n + e];
        if (bias_mode == 1) s += bias[i];
        else if (bias_mode == 2) s += bias[j];
        *dst = s;
    }
}

}  // namespace scat

using namespace scat;

extern "C" int64_t scat_gemm_ws(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    GemmPlan p = gemm_plan(M, N, K);
    return p.splits > 1 ? (int64_t)p.splits * M * N * sizeof(float) : 0;
}

extern "C" int scat_gemm(const float* a, int64_t a_si, int64_t a_sk, const float* b, int64_t b_sk, int64_t b_sj,
                         float* c, int64_t c_si, int64_t c_sj, int M, int N, int K, const float* bias, int bias_mode,
                         int accumulate, void* ws, int64_t ws_bytes, void* stream) {
    SCAT_REQUIRE(a && b && c, SCAT_E_ARG, "scat_gemm: null pointer");
    SCAT_REQUIRE(M > 0 && N > 0 && K > 0, SCAT_E_SHAPE, "scat_gemm: non-positive dimension");
    SCAT_REQUIRE(a_si == 1 || a_sk == 1, SCAT_E_SHAPE, "scat_gemm: A needs a unit stride");
    SCAT_REQUIRE(b_sk == 1 || b_sj == 1, SCAT_E_SHAPE, "scat_gemm: B needs a unit stride");
    SCAT_REQUIRE(bias_mode >= 0 && bias_mode <= 2 && (bias_mode == 0 || bias), SCAT_E_ARG, "scat_gemm: bias");
    GemmPlan p = gemm_plan(M, N, K);
    int64_t need = p.splits > 1 ? (int64_t)p.splits * M * N * sizeof(float) : 0;
    if (need > ws_bytes || (need && !ws)) { p.splits = 1; need = 0; }   // no workspace: single pass
    SCAT_REQUIRE(fits_i32(((int64_t)(M - 1) * a_si + (int64_t)(K - 1) * a_sk + 1) * 4) &&
                     fits_i32(((int64_t)(N - 1) * b_sj + (int64_t)(K - 1) * b_sk + 1) * 4),
                 SCAT_E_SHAPE, "scat_gemm: operand exceeds 2 GiB");
    SCAT_REQUIRE(fits_i32(((int64_t)(M - 1) * c_si + (int64_t)(N - 1) * c_sj + 1) * 4), SCAT_E_SHAPE,
                 "scat_gemm: output exceeds 2 GiB");
    MatDesc da{a, a_si, a_sk, 0, M, K, (int64_t)(M - 1) * a_si + (int64_t)(K - 1) * a_sk + 1};
    MatDesc db{b, b_sj, b_sk, 0, N, K, (int64_t)(N - 1) * b_sj + (int64_t)(K - 1) * b_sk + 1};
    OutDesc dc{};
    hipStream_t st = (hipStream_t)stream;
    if (p.splits > 1) {
        dc.p = (float*)ws; dc.mode = 0; dc.si = N; dc.sj = 1; dc.sz = (int64_t)M * N; dc.I = M; dc.J = N; dc.n = (int64_t)M * N;
    } else {
        dc.p = c; dc.mode = 0; dc.si = c_si; dc.sj = c_sj; dc.sz = 0; dc.I = M; dc.J = N;
        dc.bias = bias; dc.bias_mode = bias_mode; dc.accumulate = accumulate;
        dc.n = (int64_t)(M - 1) * c_si + (int64_t)(N - 1) * c_sj + 1;
    }
    const bool ak = (a_sk == 1), bk = (b_sk == 1);
    // 16-B loads along k where the contraction dim is contiguous, a multiple of 4, and rows stay 16-B aligned
    const bool av = ak && K % 4 == 0 && a_si % 4 == 0 && ((uintptr_t)a & 15) == 0;
    const bool bv = bk && K % 4 == 0 && b_sj % 4 == 0 && ((uintptr_t)b & 15) == 0;
    if (ak && bk) {
        if (av && bv) gemm_dispatch<true, true, 4, 4>(p, da, db, dc, M, N, K, st);
        else gemm_dispatch<true, true, 1, 1>(p, da, db, dc, M, N, K, st);
    } else if (ak) {
        if (av) gemm_dispatch<true, false, 4, 1>(p, da, db, dc, M, N, K, st);
        else gemm_dispatch<true, false, 1, 1>(p, da, db, dc, M, N, K, st);
    } else if (bk) {
        if (bv) gemm_dispatch<false, true, 1, 4>(p, da, db, dc, M, N, K, st);
        else gemm_dispatch<false, true, 1, 1>(p, da, db, dc, M, N, K, st);
    } else {
        gemm_dispatch<false, false, 1, 1>(p, da, db, dc, M, N, K, st);
    }
    SCAT_LAUNCH_CHECK("scat_gemm");
    if (p.splits > 1) {
        int64_t n = (int64_t)M * N;
        int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
        hipLaunchKernelGGL(splitk_reduce_bias_kernel, dim3(blocks), dim3(256), 0, st, (const float*)ws, c, c_si, c_sj,
                           M, N, p.splits, bias, bias_mode, accumulate);
        SCAT_LAUNCH_CHECK("scat_gemm(reduce)");
    }
    return SCAT_OK;
}

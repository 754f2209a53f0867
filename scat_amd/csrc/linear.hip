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

// ---------------------------------------------------------------- grouped launch
//
// The twelve weight-gradient contractions of the token mixer's backward (to_qkv, to_out, FeedForward x 3 layers,
// models/vision_transformer.py:33-35,52-57) are independent of each other and each too small for the chip (3 .. 312
// tiles of 64 x 64 on 256 CUs, 17-97 us apiece at 14-50 TF).  One grid whose unit list spans all of them: a workgroup
// finds its problem from a table in the kernel arguments, then runs the ordinary tile body.  Split-K slabs of every
// problem live in one workspace and are combined by one reduce launch, fixed order.
constexpr int GROUP_MAX = 16;
struct GroupItem {
    const float* a;
    const float* b;
    float* c;
    float* slab;          // ws slice of this problem (splits > 1), else nullptr
    int a_si, a_sk, b_sj, b_sk, c_si, c_sj;
    int M, N, K;
    int first;            // first unit (tile x K-slice) of this problem in the grid
    int tiles;
    int rfirst;           // first 256-element block of this problem in the reduce grid
};
struct GroupArgs {
    int n, splits, kchunk_steps;      // the same number of K-slices for every problem (K is the token count for all)
    GroupItem it[GROUP_MAX];
};

template <class LA, class LB, int BM>
__global__ __launch_bounds__(NT) void gemm_group_kernel(GroupArgs g) {
    int p = 0;
#pragma unroll 1
    for (int q = 1; q < g.n; ++q)
        if ((int)blockIdx.x >= g.it[q].first) p = q;
    const GroupItem& t = g.it[p];
    const int local = blockIdx.x - t.first;
    const int z = local / t.tiles, tile = local - z * t.tiles;
    MatDesc da{t.a, t.a_si, t.a_sk, 0, t.M, t.K, (int64_t)(t.M - 1) * t.a_si + (int64_t)(t.K - 1) * t.a_sk + 1};
    MatDesc db{t.b, t.b_sj, t.b_sk, 0, t.N, t.K, (int64_t)(t.N - 1) * t.b_sj + (int64_t)(t.K - 1) * t.b_sk + 1};
    OutDesc dc{};
    dc.mode = 0; dc.I = t.M; dc.J = t.N;
    if (g.splits > 1) {
        dc.p = t.slab; dc.si = t.N; dc.sj = 1; dc.sz = (int64_t)t.M * t.N; dc.n = (int64_t)t.M * t.N;
    } else {
        dc.p = t.c; dc.si = t.c_si; dc.sj = t.c_sj; dc.sz = 0; dc.n = (int64_t)(t.M - 1) * t.c_si + (int64_t)(t.N - 1) * t.c_sj + 1;
    }
    gemm_tile<LA, LB, BM, 64, 16, 2, 2>(da, db, dc, t.M, t.N, t.K, g.kchunk_steps * 16, tile, z);
}

__global__ __launch_bounds__(256) void gemm_group_reduce_kernel(GroupArgs g) {
    int p = 0;
#pragma unroll 1
    for (int q = 1; q < g.n; ++q)
        if ((int)blockIdx.x >= g.it[q].rfirst) p = q;
    const GroupItem& t = g.it[p];
    const int64_t n = (int64_t)t.M * t.N;
    const int64_t e = (int64_t)(blockIdx.x - t.rfirst) * 256 + threadIdx.x;
    if (e >= n) return;
    float s = 0.f;
    for (int z = 0; z < g.splits; ++z) s += t.slab[(int64_t)z * n + e];
    const int i = (int)(e / t.N), j = (int)(e - (int64_t)i * t.N);
    t.c[(int64_t)i * t.c_si + (int64_t)j * t.c_sj] = s;
}

// rows of a tile: 64.  (128-row tiles — twice the MFMAs per staged B element, half the tiles — measured 3 % slower on
// the token mixer's twelve problems at batch 96, tools/vit_group_bench.py; SCAT_GROUP_BM=128 for A/B runs.)
static int group_bm(const ScatGemmProblem*, int) {
    static const int forced = diag_env_int("SCAT_GROUP_BM", 0);
    return forced == 128 ? 128 : 64;
}

static int group_splits(const ScatGemmProblem* pr, int n, int64_t* tiles_out) {
    int64_t tiles = 0;
    int kmin = 1 << 30;
    const int bm = group_bm(pr, n);
    for (int q = 0; q < n; ++q) {
        tiles += (int64_t)cdiv(pr[q].M, bm) * cdiv(pr[q].N, 64);
        if (pr[q].K < kmin) kmin = pr[q].K;
    }
    if (tiles_out) *tiles_out = tiles;
    int s = cdiv(2048, tiles > 0 ? tiles : 1);         // ~8 workgroups per CU in flight over the launch
    const int smax = kmin / 256 > 0 ? kmin / 256 : 1;  // >= 256 deep per slice
    if (s > smax) s = smax;
    if (s > 16) s = 16;
    return s < 1 ? 1 : s;
}

}  // namespace scat

using namespace scat;

extern "C" int64_t scat_gemm_group_ws(const ScatGemmProblem* problems, int n) {
    if (!problems || n <= 0 || n > GROUP_MAX) return 0;
    const int s = group_splits(problems, n, nullptr);
    if (s <= 1) return 0;
    int64_t need = 0;
    for (int q = 0; q < n; ++q) need += (int64_t)s * problems[q].M * problems[q].N * sizeof(float);
    return need;
}

// c_q[M_q, N_q] = A_q . B_q for n <= 16 independent problems in ONE launch (+ one reduce launch).  Every problem must
// have the operand layout of a weight gradient: A and B contiguous along their OUTPUT index (a_si == 1, b_sj == 1) and
// the same contraction length K.
extern "C" int scat_gemm_group(const ScatGemmProblem* problems, int n, void* ws, int64_t ws_bytes, void* stream) {
    SCAT_REQUIRE(problems && n > 0 && n <= GROUP_MAX, SCAT_E_ARG, "scat_gemm_group: 1..%d problems", GROUP_MAX);
    GroupArgs g{};
    g.n = n;
    int64_t tiles = 0;
    g.splits = group_splits(problems, n, &tiles);
    const int K = problems[0].K;
    g.kchunk_steps = cdiv(cdiv(K, g.splits), 16);
    g.splits = cdiv(K, g.kchunk_steps * 16);
    const int64_t need = scat_gemm_group_ws(problems, n);
    SCAT_REQUIRE(g.splits == 1 || (ws && ws_bytes >= need && ((uintptr_t)ws & 15) == 0), SCAT_E_WORKSPACE,
                 "scat_gemm_group: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)need);
    int first = 0, rfirst = 0;
    float* slab = (float*)ws;
    const int bm = group_bm(problems, n);
    for (int q = 0; q < n; ++q) {
        const ScatGemmProblem& pr = problems[q];
        SCAT_REQUIRE(pr.a && pr.b && pr.c && pr.M > 0 && pr.N > 0 && pr.K == K, SCAT_E_SHAPE,
                     "scat_gemm_group: problem %d: null pointer, empty, or a different contraction length", q);
        SCAT_REQUIRE(pr.a_si == 1 && pr.b_sj == 1, SCAT_E_SHAPE,
                     "scat_gemm_group: problem %d: operands must be contiguous along their output index", q);
        SCAT_REQUIRE(fits_i32(((int64_t)(pr.M - 1) * pr.a_si + (int64_t)(K - 1) * pr.a_sk + 1) * 4) &&
                         fits_i32(((int64_t)(pr.N - 1) * pr.b_sj + (int64_t)(K - 1) * pr.b_sk + 1) * 4) &&
                         fits_i32(((int64_t)(pr.M - 1) * pr.c_si + (int64_t)(pr.N - 1) * pr.c_sj + 1) * 4),
                     SCAT_E_SHAPE, "scat_gemm_group: problem %d exceeds 2 GiB", q);
        GroupItem& t = g.it[q];
        t.a = pr.a; t.b = pr.b; t.c = pr.c;
        t.a_si = (int)pr.a_si; t.a_sk = (int)pr.a_sk; t.b_sj = (int)pr.b_sj; t.b_sk = (int)pr.b_sk;
        t.c_si = (int)pr.c_si; t.c_sj = (int)pr.c_sj;
        t.M = pr.M; t.N = pr.N; t.K = K;
        t.tiles = cdiv(pr.M, bm) * cdiv(pr.N, 64);
        t.first = first;
        first += t.tiles * g.splits;
        t.rfirst = rfirst;
        rfirst += (int)(((int64_t)pr.M * pr.N + 255) / 256);
        t.slab = g.splits > 1 ? slab : nullptr;
        if (g.splits > 1) slab += (int64_t)g.splits * pr.M * pr.N;
    }
    hipStream_t st = (hipStream_t)stream;
    set_kernel_label("gemm_group%d_i1j1_%dx64x16_split%d", n, bm, g.splits);
    if (bm == 128) {
        constexpr size_t lds_bytes = sizeof(float) * 2 * 16 * (128 + 64 + 2 * LPAD);
        hipLaunchKernelGGL((gemm_group_kernel<MatLoader<128, 16, false, 1>, MatLoader<64, 16, false, 1>, 128>), dim3(first),
                           dim3(NT), lds_bytes, st, g);
    } else {
        constexpr size_t lds_bytes = sizeof(float) * 2 * 16 * (64 + 64 + 2 * LPAD);
        hipLaunchKernelGGL((gemm_group_kernel<MatLoader<64, 16, false, 1>, MatLoader<64, 16, false, 1>, 64>), dim3(first),
                           dim3(NT), lds_bytes, st, g);
    }
    SCAT_LAUNCH_CHECK("scat_gemm_group");
    if (g.splits > 1) {
        hipLaunchKernelGGL(gemm_group_reduce_kernel, dim3(rfirst), dim3(256), 0, st, g);
        SCAT_LAUNCH_CHECK("scat_gemm_group(reduce)");
    }
    return SCAT_OK;
}

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

// Fused qkv projection + softmax attention of one transformer layer (models/vision_transformer.py:61-76):
//     qkv = h . Wqkv^T   (no bias)   ->   per (image, head):  P = softmax(scale . Q K^T),  O = P V
// for short sequences (n <= 32 tokens: the 21 joint tokens of hand_net.py:364), head dim 64.
//
// One workgroup = one HEAD of one BLOCK OF IMAGES: IB = 128 / n images (6 x 21 = 126 token rows) x the 192 columns
// of Wqkv that belong to the head (q_h | k_h | v_h).  The projection is a 128 x 192 x dim contraction on split-operand
// products (split.h): a wavefront owns 32 token rows — its A fragments come straight from global memory (lane (row, h)
// takes 8 consecutive features = 32 bytes, split in registers) — and the head's weight columns, which all four
// wavefronts share, are staged through LDS as the pre-split planes the weight re-layout wrote (a plain copy, no
// arithmetic).  The 126 x 192 result never leaves the CU for the attention: it is parked in LDS (over the weight
// buffers), every image's 21 x 21 scores, softmax and P.V are computed there, and only what the backward needs goes to
// HBM (qkv, the probabilities) besides the output.  blockIdx % heads is the head: with round-robin placement every
// XCD keeps ONE head's 0.9 MB weight slice in its L2 while the token blocks stream through.
//
// Why blocks of images and not one workgroup per image: a 21-row tile would re-read the layer's weights once per
// image (96 x 7.2 MB from L2 at batch 96) and fill 21 of 32 MFMA rows; why not one workgroup per 128 tokens over all
// heads: 16 workgroups for 256 CUs.  128 workgroups of 7056 MFMAs each is what 2016 tokens offer.
#include "conv_common.h"
#include "split.h"

namespace scat {

struct VitDesc {
    const float* h;        // [Mtok][dim] tokens (LayerNorm already applied by the caller's ln_fwd, which the backward needs)
    const float* wq;       // pre-split planes of Wqkv: [chunk16][plane][3*inner][16 bf16]  (wprep_job, M = 3*inner, C = dim)
    float* qkv;            // [Mtok][3*inner]
    float* attn;           // [B][heads][n][n]
    float* ao;             // [Mtok][inner]
    int B, n, dim, heads, Mtok, IB;
    float scale;
    int64_t nh, nw;
};

constexpr int VF_COLS = 192;                 // q_h | k_h | v_h
constexpr int VF_TS = VF_COLS + 1;           // padded row of the parked projection tile (floats)

// 512 threads: wavefronts 0-3 own columns 0..95 of their 32 token rows, wavefronts 4-7 columns 96..191 (two wavefronts
// per SIMD: one's fragment reads and waits are covered by the other's MFMAs)
// RW = token rows per workgroup (128: six 21-token images, 64: three).  Threads = 2 x (RW / 32) wavefronts: one per
// (32-row block, column half).  64-row blocks give 32 x 8 = 256 workgroups at batch 96 — every CU one — at the price of
// reading each head's weight slice twice as often.
template <int RW>
__global__ __launch_bounds__(RW * 4) void vit_qkv_attn_kernel(VitDesc d) {
    constexpr int VF_NT = RW * 4, VF_NI = (2304 + VF_NT - 1) / VF_NT;     // staging items per thread (2304 per stage)
    extern __shared__ __align__(16) float lds[];
    // during the projection: three weight stage buffers [2 chunks][3 planes][2 k-octets][192 cols] x 16 B = 36,864 B each;
    // afterwards the same memory holds the projection tile [128][193] floats and the scores [IB][n][n + 1]
    u32x4* const W0 = (u32x4*)lds;
    constexpr int WBUF = 2 * 3 * 2 * VF_COLS;                      // u32x4 per stage buffer
    const int head = blockIdx.x % d.heads, blk = blockIdx.x / d.heads;
    const int inner = d.heads * 64, ld = 3 * inner;
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) % (RW / 32), wcol = (tid >> 6) / (RW / 32);
    const int l31 = lane & 31, lh = lane >> 5;
    const int rows_blk = d.IB * d.n;
    const int r0 = blk * rows_blk;
    const int nrows = min(rows_blk, d.Mtok - r0);
    const int nchunk = (d.dim + 15) / 16, nstage = (nchunk + 1) / 2;

    // ---- weight staging: item i of this thread = (chunk t, plane p, octet hh, col c); 2304 items of 16 bytes per stage
    const __amdgpu_buffer_rsrc_t rsw = make_rsrc(d.wq, d.nw);
    int woff[VF_NI], wlds[VF_NI];
#pragma unroll
    for (int i = 0; i < VF_NI; ++i) {
        const int it = min(tid + VF_NT * i, 2303);
        const int c = it % VF_COLS, q = it / VF_COLS;             // q = (t*3 + p)*2 + hh
        const int hh = q & 1, tp = q >> 1, p = tp % 3, t = tp / 3;
        const int row = (c >> 6) * inner + head * 64 + (c & 63);  // row of Wqkv
        woff[i] = ((t * 3 + p) * ld + row) * 32 + hh * 16;        // + stage * 2 * 3 * ld * 32
        wlds[i] = q * VF_COLS + c;
    }
    u32x4 wst[2][VF_NI];                               // two register sets: loads run two stages ahead
    auto load_w = [&](int s, auto set_tag) {
        constexpr int Q = decltype(set_tag)::value;
#pragma unroll
        for (int i = 0; i < VF_NI; ++i) {
            const int t = min(tid + VF_NT * i, 2303) / VF_COLS / 6;
            const bool ok = s < nstage && 2 * s + t < nchunk && tid + VF_NT * i < 2304;
            wst[Q][i] = __builtin_amdgcn_raw_buffer_load_b128(rsw, ok ? woff[i] : OOB, s * (2 * 3 * ld * 32), 0);
        }
    };
    auto store_w = [&](u32x4* buf, auto set_tag) {
        constexpr int Q = decltype(set_tag)::value;
#pragma unroll
        for (int i = 0; i < VF_NI; ++i)
            if (tid + VF_NT * i < 2304) buf[wlds[i]] = wst[Q][i];
    };

    // ---- tokens: lane (row, lh) takes features 16 ch + 8 lh .. + 7 of its row, split in registers
    const __amdgpu_buffer_rsrc_t rsh = make_rsrc(d.h, d.nh);
    const int arow = wave * 32 + l31;
    const int abase = arow < nrows ? (r0 + arow) * d.dim * 4 : OOB;
    float araw3[2][2][8];
    auto load_a = [&](int s, auto set_tag) {
        constexpr int Q = decltype(set_tag)::value;
        float (&araw)[2][8] = araw3[Q];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int k0 = (2 * s + t) * 16 + 8 * lh;
#pragma unroll
            for (int v = 0; v < 2; ++v) {       // dim % 4 == 0: whole 16-byte pieces inside the row or outside
                const bool ok = abase != OOB && s < nstage && k0 + 4 * v < d.dim;
                const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rsh, ok ? abase + (k0 + 4 * v) * 4 : OOB, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) araw[t][4 * v + e] = __uint_as_float(x[e]);
            }
        }
    };

    f32x16 acc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    // weights: global loads three stages ahead (two register sets), LDS two stages ahead (three buffers), so a stage's
    // first fragment can be read before the barrier that ends the previous stage; tokens: two register sets
    load_w(0, S0{});
    load_w(1, S1{});
    load_a(0, S0{});
    load_a(1, S1{});
    store_w(W0, S0{});
    load_w(2, S0{});
    store_w(W0 + WBUF, S1{});
    __syncthreads();
    u32x4 bf[2][3];
    auto read_b = [&](u32x4 (&dst)[3], const u32x4* buf, int t, int c) {
#pragma unroll
        for (int p = 0; p < 3; ++p) dst[p] = buf[((t * 3 + p) * 2 + lh) * VF_COLS + (3 * wcol + c) * 32 + l31];
    };
    read_b(bf[0], W0, 0, 0);
    int rb = 0, wb = 2;
    auto stage = [&](int s, auto cur_tag) {
        constexpr int CUR = decltype(cur_tag)::value;
        const u32x4* cur = W0 + rb * WBUF;
        rb = rb == 2 ? 0 : rb + 1;
        const u32x4* nxt = W0 + rb * WBUF;
        u32x4 a[2][3];
#pragma unroll
        for (int t = 0; t < 2; ++t) split3x8(araw3[CUR][t], a[t][0], a[t][1], a[t][2]);
        load_a(s + 2, cur_tag);                                        // (set CUR is free again: just split)
        load_w(s + 3, std::integral_constant<int, CUR ^ 1>{});        // (set CUR^1 went to LDS a stage ago)
        static_for<6>([&](auto g_tag) {
            constexpr int G = decltype(g_tag)::value, t = G / 3, c = G % 3;
            constexpr int fcur = G & 1, fnxt = fcur ^ 1;
            if constexpr (G < 5) read_b(bf[fnxt], cur, (G + 1) / 3, (G + 1) % 3);
            else read_b(bf[fnxt], nxt, 0, 0);                          // published one barrier ago
            __builtin_amdgcn_sched_barrier(0);
            acc[c] = mfma_split(a[t], bf[fcur], acc[c]);
            __builtin_amdgcn_sched_barrier(0);
        });
        store_w(W0 + wb * WBUF, cur_tag);                              // stage s + 2 (loaded two stages ago)
        wb = wb == 2 ? 0 : wb + 1;
        __syncthreads();
    };
    for (int s = 0; s < nstage; s += 2) {
        stage(s, S0{});
        if (s + 1 < nstage) stage(s + 1, S1{});
    }

    // ---- park the projection tile in LDS (the weight buffers are dead) and write qkv
    float* const T = lds;                                   // [128][VF_TS]
    float* const S = lds + RW * VF_TS;                      // [IB][n][n + 1]
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, col = (3 * wcol + c) * 32 + l31;
            T[row * VF_TS + col] = acc[c][r];
            if (row < nrows)
                d.qkv[(int64_t)(r0 + row) * ld + (col >> 6) * inner + head * 64 + (col & 63)] = acc[c][r];
        }
    __syncthreads();
    const int n = d.n, PS = n + 1;
    const int nimg = nrows / n;
    // scores: S[i][a][b] = scale * <Q[a], K[b]>
    for (int e = tid; e < nimg * n * n; e += VF_NT) {
        const int i = e / (n * n), ab = e - i * n * n, a = ab / n, b = ab - a * n;
        const float* q = T + (i * n + a) * VF_TS, *k = T + (i * n + b) * VF_TS + 64;
        float s = 0.f;
#pragma unroll 16
        for (int x = 0; x < 64; ++x) s = fmaf(q[x], k[x], s);
        S[(i * n + a) * PS + b] = s * d.scale;
    }
    __syncthreads();
    // softmax over b: half a wavefront per row (n <= 32)
    for (int row = 2 * (tid >> 6) + lh; row < nimg * n; row += 2 * (VF_NT / 64)) {
        const float v = l31 < n ? S[row * PS + l31] : -INFINITY;
        float m = v;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        const float ex = l31 < n ? expf(v - m) : 0.f;
        float sum = ex;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float p = ex / sum;
        if (l31 < n) {
            S[row * PS + l31] = p;
            const int img = blk * d.IB + row / n, a = row % n;
            d.attn[(((int64_t)img * d.heads + head) * n + a) * n + l31] = p;
        }
    }
    __syncthreads();
    // O = P V
    for (int e = tid; e < nimg * n * 64; e += VF_NT) {
        const int row = e >> 6, x = e & 63;
        const int i = row / n;
        const float* p = S + row * PS;
        const float* v = T + (i * n) * VF_TS + 128 + x;
        float s = 0.f;
        for (int b = 0; b < n; ++b) s = fmaf(p[b], v[b * VF_TS], s);
        d.ao[(int64_t)(r0 + row) * inner + head * 64 + x] = s;
    }
}

}  // namespace scat

using namespace scat;

extern "C" int64_t scat_vit_qkv_attn_fwd_ws(int dim, int heads) {
    return (int64_t)((dim + 15) / 16) * 3 * (3 * heads * 64) * 32;
}

// qkv[B*n, 3*heads*64] = h[B*n, dim] . wqkv[3*heads*64, dim]^T; attn[B,heads,n,n] = softmax(scale . q k^T);
// ao[B*n, heads*64] = attn . v, heads laid out 'b n (h d)' (models/vision_transformer.py:62-76).
// n <= 32, dim % 4 == 0, head dim 64, split-operand products.  ws: scat_vit_qkv_attn_fwd_ws(dim, heads) bytes.
extern "C" int scat_vit_qkv_attn_fwd(const float* h, const float* wqkv, float* qkv, float* attn, float* ao, int B, int n,
                                     int dim, int heads, float scale, void* ws, int64_t ws_bytes, void* stream) {
    SCAT_REQUIRE(h && wqkv && qkv && attn && ao, SCAT_E_ARG, "scat_vit_qkv_attn_fwd: null pointer");
    SCAT_REQUIRE(math_mode() == 1, SCAT_E_ARG, "scat_vit_qkv_attn_fwd: needs the split-operand product mode");
    SCAT_REQUIRE(B > 0 && n > 0 && n <= 32 && dim > 0 && dim % 4 == 0 && heads > 0, SCAT_E_SHAPE,
                 "scat_vit_qkv_attn_fwd: n <= 32, dim %% 4 == 0 required (n=%d dim=%d)", n, dim);
    SCAT_REQUIRE(ws && ws_bytes >= scat_vit_qkv_attn_fwd_ws(dim, heads) && ((uintptr_t)ws & 15) == 0 &&
                     ((uintptr_t)h & 15) == 0,
                 SCAT_E_WORKSPACE, "scat_vit_qkv_attn_fwd: workspace too small / unaligned");
    const int inner = heads * 64;
    SCAT_REQUIRE(fits_i32((int64_t)B * n * dim * 4) && fits_i32((int64_t)((dim + 15) / 16) * 3 * 3 * inner * 32),
                 SCAT_E_SHAPE, "scat_vit_qkv_attn_fwd: operand exceeds 32-bit byte offsets");
    hipStream_t st = (hipStream_t)stream;
    wprep_launch(wprep_job(wqkv, ws, 3 * inner, dim, 0, 1, 1, 1, 1, 0, 0, 1), st);
    VitDesc d{};
    d.h = h; d.wq = (const float*)ws; d.qkv = qkv; d.attn = attn; d.ao = ao;
    // 64-row blocks when 128-row ones would leave CUs without a workgroup
    const bool small = n <= 64 && (int64_t)cdiv(B, 128 / n) * heads < 200 && B > 128 / n;
    const int RW = small ? 64 : 128;
    d.B = B; d.n = n; d.dim = dim; d.heads = heads; d.Mtok = B * n; d.IB = RW / n; d.scale = scale;
    d.nh = (int64_t)B * n * dim;
    d.nw = ((int64_t)((dim + 15) / 16) * 3 * 3 * inner * 32 + 3) / 4;
    const int nblk = cdiv(B, d.IB);
    // LDS: max(three weight stage buffers, projection tile + scores)
    constexpr size_t wbytes = (size_t)3 * 2 * 3 * 2 * VF_COLS * 16;
    auto launch = [&](auto rw_tag) {
        constexpr int R = decltype(rw_tag)::value;
        constexpr size_t tbytes = sizeof(float) * (R * VF_TS + R * 33);
        constexpr size_t lds_bytes = tbytes > wbytes ? tbytes : wbytes;
        auto kern = vit_qkv_attn_kernel<R>;
        static bool once = (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                (int)lds_bytes) == hipSuccess);
        (void)once;
        hipLaunchKernelGGL(kern, dim3(nblk * heads), dim3(R * 4), lds_bytes, st, d);
    };
    set_kernel_label("vit_qkv_attn_fused_%dx192x32", RW);
    if (small) launch(std::integral_constant<int, 64>{});
    else launch(std::integral_constant<int, 128>{});
    SCAT_LAUNCH_CHECK("scat_vit_qkv_attn_fwd");
    return SCAT_OK;
}

#pragma once
#include "gemm_engine.h"

namespace scat {

static inline int check_geom(const char* who, int B, int Cin, int H, int W, int Cout, int KH, int KW, int stride,
                             int pad, int* OH, int* OW) {
    SCAT_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, SCAT_E_SHAPE, "%s: non-positive dimension", who);
    SCAT_REQUIRE(KH == KW && (KH == 1 || KH == 3 || KH == 7), SCAT_E_SHAPE, "%s: kernel %dx%d unsupported", who, KH,
                 KW);
    SCAT_REQUIRE(stride == 1 || stride == 2, SCAT_E_SHAPE, "%s: stride %d unsupported", who, stride);
    SCAT_REQUIRE(pad >= 0 && pad < KH + (KH == 1), SCAT_E_SHAPE, "%s: pad %d unsupported", who, pad);
    *OH = (H + 2 * pad - KH) / stride + 1;
    *OW = (W + 2 * pad - KW) / stride + 1;
    SCAT_REQUIRE(*OH > 0 && *OW > 0, SCAT_E_SHAPE, "%s: empty output", who);
    // byte offsets are 32-bit buffer-load offsets: every tensor must stay below 2 GiB
    SCAT_REQUIRE(fits_i32((int64_t)B * Cin * H * W * 4) && fits_i32((int64_t)B * Cout * *OH * *OW * 4) &&
                     fits_i32((int64_t)Cout * Cin * KH * KW * 4),
                 SCAT_E_SHAPE, "%s: tensor exceeds 2 GiB", who);
    return SCAT_OK;
}

// split-operand taps kernel (conv1x1.hip): see taps_split_launch
struct TapsGeom {
    int H, W;             // source plane
    int OH, OW;           // pixel grid of the contraction's columns
    int a, tb, c0y, c0x;  // source (y, x) = (oy*a + th*tb + c0y, ox*a + tw*tb + c0x)
    int KHt, KWt;         // taps used
    int KH, KW, kh0, kw0, ts, transposed;   // which weights: (kh0 + ts*th, kw0 + ts*tw) of w[Cout][Cin][KH][KW]
};
int64_t taps_split_ws(int M, int C, int ntap);
// w_ready: ws already holds this call's re-laid weights (scat_wprep_run), skip the re-layout launch
void taps_split_launch(const TapsGeom& g, const float* src, const float* w, const OutDesc& dc, int B, int C, int M,
                       const float* in_scale, const float* in_shift, int in_relu, void* ws, const char* label,
                       hipStream_t st, bool w_ready = false);

// One weight re-layout for the split-operand kernels: dst[tap][chunk][plane][i][16 bf16] = the three bf16 terms of
// element (i, c) of tap (kh0 + ts*(t / KWt), kw0 + ts*(t % KWt)) of w[Cout][Cin][KH][KW]; (i, c) = (co, ci), or
// (ci, co) when transposed.  Every split kernel's weight operand is one or more of these (wprep_launch: one job, one
// launch; scat_wprep_run: a table of jobs, one launch for a whole network).
struct WPrepJob {
    const float* w;
    uint16_t* dst;
    int M, C, transposed, KH, KW, ntap, KWt, kh0, kw0, ts;
    int blk0, nblk;    // 256-element blocks [blk0, blk0 + nblk) of the batched launch
};
WPrepJob wprep_job(const float* w, void* dst, int M, int C, int transposed, int KH, int KW, int ntap, int KWt, int kh0,
                   int kw0, int ts);
void wprep_launch(const WPrepJob& j, hipStream_t st);
void wprep_batch_launch(const WPrepJob* jobs_dev, int njobs, int nblocks, hipStream_t st);
// jobs of scat_conv2d_dgrad_s2's parity classes (each class has its own slice of ws); returns their number
int dgrad_s2_wprep_jobs(const float* w, void* ws, int Cin, int Cout, int KH, int KW, int pad, WPrepJob* out);

// split-operand weight gradient (conv_wgrad_split.hip): 1x1/pad 0 and 3x3/pad 1, stride 1
struct WgSplitPlan {
    int M, N, mi, ni, stages, spz, splits;
};
WgSplitPlan wgrad_split_plan(int B, int Cin, int Cout, int KK, int HW);
void wgrad_split_launch(const WgSplitPlan& p, const float* dy, const float* x, float* out, int B, int Cin, int H, int W,
                        int Cout, int KK, int stride, const float* in_scale, const float* in_shift, int in_relu,
                        hipStream_t st, const float* dy2 = nullptr, const float* coef3 = nullptr);
// XCD-local split-K: the tiles of one split-K slice of a weight gradient read the same pixels of both operands, so they
// should run on ONE XCD (one L2) at about the same time.  Hardware deals block b to XCD b % 8 (MI355X_MICROARCH.md,
// workgroup dispatch).  A slice's `tiles` tiles are cut into `ngroup` contiguous runs of tg = ceil(tiles / ngroup); the
// unit dealt to an XCD is a (slice, run) pair, numbered v = slice * ngroup + run: block b -> XCD x = b % 8, k = b / 8,
// position in the run k % tg, v = (k / tg) * 8 + x.  ngroup = 1 (whole slices, grid padded to a multiple of 8 slices)
// whenever there are many slices; 8 / gcd(8, splits) for the few-slice launches so that all eight XCDs get work.
static inline int splitk_xcd_groups(int splits) {
    static const int on = diag_env_int("SCAT_WG_XCD", 1);
    if (!on) return 0;         // A/B switch: 0 = tiles in XCD chunks, slices in launch order (the round-2 mapping)
    if (splits >= 32 || splits % 8 == 0) return 1;
    int g = 8;
    while (g > 1 && (splits % g)) g >>= 1;      // gcd(8, splits)
    return 8 / g;
}
static inline int splitk_xcd_grid(int tiles, int splits, int ngroup) {
    if (ngroup == 0) return tiles * splits;
    const int tg = (tiles + ngroup - 1) / ngroup;
    const int nv = (splits * ngroup + 7) / 8 * 8;
    return tg * nv;
}
__device__ __forceinline__ bool splitk_xcd_map(int b, int tiles, int splits, int ngroup, int& tile, int& z) {
    if (ngroup == 0) {
        z = b / tiles;
        const int id = b - z * tiles, q = tiles >> 3, r = tiles & 7, x = id & 7, s = id >> 3;
        tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s;      // xcd_remap of gemm_engine.h
        return true;
    }
    const int tg = (tiles + ngroup - 1) / ngroup;
    const int k = b >> 3;
    const int v = (k / tg) * 8 + (b & 7);
    z = v / ngroup;
    tile = (v - z * ngroup) * tg + k % tg;
    return z < splits && tile < tiles;
}

// pointwise weight gradient, second generation (conv_wgrad_pw.hip): 32-pixel stages read a cache line per row, 256 x 128
// tiles on eight consumer + four producer wavefronts.  ok = false: the shape stays on the kernels above.
struct WgPwPlan {
    bool ok;
    int wa, wb, stages, spz, splits;
};
WgPwPlan wgrad_pw_plan(int B, int Cin, int Cout, int HW, bool dsa, const void* dy, const void* x);
void wgrad_pw_launch(const WgPwPlan& p, const float* dy, const float* x, float* out, int B, int Cin, int HW, int Cout,
                     const float* in_scale, const float* in_shift, int in_relu, hipStream_t st,
                     const float* dy2 = nullptr, const float* coef3 = nullptr);
// row-walking 3x3 / stride-1 weight gradient for 32- and 64-channel layers (conv_wgrad_rows.hip)
bool wgrad_rows_ok(int B, int Cin, int H, int W, int Cout, int KH, int stride, int pad, const void* dy, const void* x);
int64_t wgrad_rows_ws(int B, int Cin, int H, int W, int Cout);
int wgrad_rows_launch(const float* dy, const float* x, float* slab, int B, int Cin, int H, int W, int Cout,
                      const float* in_scale, const float* in_shift, int in_relu, hipStream_t st);
__global__ void splitk_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out, int64_t n, int splits,
                                     int accumulate);
// out[e] (+)= sum over slabs, fixed order (vectorised / slab-parallel when n % 4 == 0)
void launch_splitk_reduce(const float* slab, float* out, int64_t n, int splits, int accumulate, hipStream_t st);

}  // namespace scat

// MaxPool2d(3,2,1) with the stem BatchNorm+ReLU fused into the load, and the 7x7 global
// average pool.  Reference: models/resnet.py:110,146-147 and :115,155-157.
#include "common.h"

namespace scat {

__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int relu,
                                                          float* __restrict__ y, int8_t* __restrict__ idx,
                                                          int64_t total, int C, int H, int W, int OH, int OW) {
    for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < total; e += gridDim.x * 256ll) {
        int ox = e % OW;
        int64_t r = e / OW;
        int oy = r % OH;
        int64_t nc = r / OH;
        int c = nc % C;
        const float* p = x + nc * H * W;
        float sc = scale ? scale[c] : 1.f, sh = scale ? shift[c] : 0.f;
        float best = -INFINITY;
        int bi = -1;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            int iy = oy * 2 - 1 + kh;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                int ix = ox * 2 - 1 + kw;
                if ((unsigned)ix >= (unsigned)W) continue;
                float v = p[iy * W + ix];
                if (scale) v = fmaf(v, sc, sh);
                if (relu) v = fmaxf(v, 0.f);
                if (v > best || v != v) { best = v; bi = kh * 3 + kw; }   // first maximum wins, like ATen
            }
        }
        y[e] = best;
        idx[e] = (int8_t)bi;
    }
}

// even H, W % 4 == 0 (the stem: 112x112 -> 56x56): a thread owns the output pair (oy, 2q), (oy, 2q+1) and reads, per
// window row, the aligned float4 of columns 4q..4q+3 plus column 4q-1 (the neighbouring lane's .w when that lane holds
// the previous quad of the same row).  Same scan order and tie/NaN rule as the scalar kernel: bit-identical results.
__global__ __launch_bounds__(256) void maxpool_fwd_pair_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ scale,
                                                               const float* __restrict__ shift, int relu,
                                                               float* __restrict__ y, int8_t* __restrict__ idx,
                                                               int64_t total, int C, int H, int W, int OH, int OW) {
    const int QW = OW / 2;                                // pairs per output row
    const int lane = threadIdx.x & 63;
    const int64_t e0 = blockIdx.x * 256ll + threadIdx.x;
    const int64_t stride = gridDim.x * 256ll;
    // every lane of a wavefront runs the same number of iterations (the shuffle below needs its neighbour alive)
    const int64_t wave0 = e0 - lane;
    for (int64_t base = wave0; base < total; base += stride) {
        const int64_t e = base + lane;
        const bool live = e < total;
        const int64_t ee = live ? e : 0;
        const int q = (int)(ee % QW);
        const int64_t r = ee / QW;
        const int oy = (int)(r % OH);
        const int64_t nc = r / OH;
        const int c = (int)(nc % C);
        const float* p = x + nc * H * W;
        const float sc = scale ? scale[c] : 1.f, sh = scale ? shift[c] : 0.f;
        float best0 = -INFINITY, best1 = -INFINITY;
        int b0 = -1, b1 = -1;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int iy = oy * 2 - 1 + kh;
            const bool rowok = (unsigned)iy < (unsigned)H;          // (uniform per output row, not per wavefront)
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rowok) v = *reinterpret_cast<const float4*>(p + iy * W + 4 * q);
            // column 4q-1: the previous lane's .w when it holds quad q-1 of the same (plane, row)
            float left = __shfl_up(v.w, 1);
            if (q > 0 && lane == 0 && rowok) left = p[iy * W + 4 * q - 1];
            if (!rowok) continue;
            float t[5] = {left, v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                if (scale) t[j] = fmaf(t[j], sc, sh);
                if (relu) t[j] = fmaxf(t[j], 0.f);
            }
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                if (!(kw == 0 && q == 0)) {                       // column -1 is padding
                    const float a = t[kw];
                    if (a > best0 || a != a) { best0 = a; b0 = kh * 3 + kw; }
                }
                const float b = t[2 + kw];
                if (b > best1 || b != b) { best1 = b; b1 = kh * 3 + kw; }
            }
        }
        if (live) {
            *reinterpret_cast<float2*>(y + 2 * e) = make_float2(best0, best1);
            *reinterpret_cast<char2*>(idx + 2 * e) = make_char2((signed char)b0, (signed char)b1);
        }
    }
}

// gather form (no atomics): an input pixel sums the windows whose arg-max it is.
// generic sizes: grid.y = (n,c) plane, threads sweep the plane.
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy,
                                                          const int8_t* __restrict__ idx, float* __restrict__ dx,
                                                          int H, int W, int OH, int OW) {
    const int64_t nc = blockIdx.y;
    const float* g = dy + nc * OH * OW;
    const int8_t* id = idx + nc * OH * OW;
    float* o = dx + nc * H * W;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < H * W; e += gridDim.x * 256) {
        const int iy = e / W, ix = e - iy * W;
        float s = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            int t = iy + 1 - kh;
            if (t < 0 || (t & 1) || (t >> 1) >= OH) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                int u = ix + 1 - kw;
                if (u < 0 || (u & 1) || (u >> 1) >= OW) continue;
                int q = (t >> 1) * OW + (u >> 1);
                if (id[q] == kh * 3 + kw) s += g[q];
            }
        }
        o[e] = s;
    }
}

// even H, W with W % 4 == 0 (the stem: 112x112 -> 56x56).  Input row 2a / column 2b is seen by one window
// row / column (tap 1), the odd ones by two (tap 2 of window a, tap 0 of window a+1), so a 2x2 input quad
// depends on the 2x2 windows (a..a+1, b..b+1) only.  One thread = two horizontally adjacent quads:
// 6 window reads (shared with the neighbours through L1), two 16-B stores.
__global__ __launch_bounds__(256) void maxpool_bwd_quad_kernel(const float* __restrict__ dy,
                                                               const int8_t* __restrict__ idx,
                                                               float* __restrict__ dx, int64_t total, int W,
                                                               int OH, int OW) {
    const int QW = OW / 2;
    for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < total; e += gridDim.x * 256ll) {
        const int b2 = e % QW;
        const int64_t r = e / QW;
        const int a = r % OH;
        const int64_t nc = r / OH;
        const float* g = dy + nc * OH * OW;
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
        for (int v = 0; v < 2; ++v) {   // quad v uses window columns v, v+1
            top[2 * v] = iv[0][v] == 4 ? gv[0][v] : 0.f;
            top[2 * v + 1] = (iv[0][v] == 5 ? gv[0][v] : 0.f) + (iv[0][v + 1] == 3 ? gv[0][v + 1] : 0.f);
            bot[2 * v] = (iv[0][v] == 7 ? gv[0][v] : 0.f) + (iv[1][v] == 1 ? gv[1][v] : 0.f);
            bot[2 * v + 1] = (iv[0][v] == 8 ? gv[0][v] : 0.f) + (iv[0][v + 1] == 6 ? gv[0][v + 1] : 0.f) +
                             (iv[1][v] == 2 ? gv[1][v] : 0.f) + (iv[1][v + 1] == 0 ? gv[1][v + 1] : 0.f);
        }
        float* o = dx + (nc * 2 * OH + 2 * a) * W + 2 * b;
        *(float4*)o = make_float4(top[0], top[1], top[2], top[3]);
        *(float4*)(o + W) = make_float4(bot[0], bot[1], bot[2], bot[3]);
    }
}

// 16 lanes per (n,c) row
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          int64_t rows, int HW, int relu) {
    int64_t row = blockIdx.x * 16ll + (threadIdx.x >> 4);
    int l = threadIdx.x & 15;
    float s = 0.f;
    if (row < rows)
        for (int i = l; i < HW; i += 16) s += x[row * HW + i];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 16);
    if (row < rows && l == 0) {
        s /= HW;
        y[row] = relu ? fmaxf(s, 0.f) : s;
    }
}

__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                          int relu, float* __restrict__ dx, int64_t total, int HW,
                                                          int accumulate) {
    for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < total; e += gridDim.x * 256ll) {
        int64_t row = e / HW;
        float g = dy[row] / HW;
        if (relu && !(y[row] > 0.f)) g = 0.f;
        dx[e] = accumulate ? dx[e] + g : g;
    }
}

// y[b,c,oy,ox] = x[b,c,2*oy,2*ox]: the pixels a 1x1/stride-2 convolution reads, packed so that its forward and weight
// gradient run as stride-1 pointwise convolutions (resnet.py:127-132 shortcut).  P = outputs per thread.
template <int P>
__global__ __launch_bounds__(256) void subsample2_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         int64_t total, int H, int W, int OH, int OW) {
    const int OWP = OW / P;
    for (int64_t e = blockIdx.x * 256ll + threadIdx.x; e < total; e += gridDim.x * 256ll) {
        const int64_t row = e / OWP;
        const int q = (int)(e - row * OWP);
        const int64_t bc = row / OH;
        const int oy = (int)(row - bc * OH);
        const float* src = x + (bc * H + 2 * oy) * W + 2 * P * q;
        float* dst = y + row * OW + P * q;
        if constexpr (P == 2) {
            const float4 v = *reinterpret_cast<const float4*>(src);
            *reinterpret_cast<float2*>(dst) = make_float2(v.x, v.z);
        } else {
            dst[0] = src[0];
        }
    }
}

static inline int grid_for(int64_t n) { return (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192); }

}  // namespace scat

using namespace scat;

extern "C" int scat_maxpool3x3s2_fwd(const float* x, const float* scale, const float* shift, int relu, float* y,
                                     int8_t* idx, int B, int C, int H, int W, void* stream) {
    SCAT_REQUIRE(x && y && idx, SCAT_E_ARG, "scat_maxpool3x3s2_fwd: null pointer");
    SCAT_REQUIRE((scale == nullptr) == (shift == nullptr), SCAT_E_ARG, "scat_maxpool3x3s2_fwd: scale/shift pair");
    SCAT_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, SCAT_E_SHAPE, "scat_maxpool3x3s2_fwd: non-positive dimension");
    int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    int64_t total = (int64_t)B * C * OH * OW;
    if (H % 2 == 0 && W % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 7) == 0 && ((uintptr_t)idx & 1) == 0) {
        hipLaunchKernelGGL(maxpool_fwd_pair_kernel, dim3(grid_for(total / 2)), dim3(256), 0, (hipStream_t)stream, x,
                           scale, shift, relu, y, idx, total / 2, C, H, W, OH, OW);
        SCAT_LAUNCH_CHECK("scat_maxpool3x3s2_fwd");
        return SCAT_OK;
    }
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, scale, shift,
                       relu, y, idx, total, C, H, W, OH, OW);
    SCAT_LAUNCH_CHECK("scat_maxpool3x3s2_fwd");
    return SCAT_OK;
}

extern "C" int scat_maxpool3x3s2_bwd(const float* dy, const int8_t* idx, float* dx, int B, int C, int H, int W,
                                     void* stream) {
    SCAT_REQUIRE(dy && idx && dx, SCAT_E_ARG, "scat_maxpool3x3s2_bwd: null pointer");
    SCAT_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, SCAT_E_SHAPE, "scat_maxpool3x3s2_bwd: non-positive dimension");
    int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    if (H % 2 == 0 && W % 4 == 0 && ((uintptr_t)dx & 15) == 0) {
        const int64_t total = (int64_t)B * C * OH * (OW / 2);
        hipLaunchKernelGGL(maxpool_bwd_quad_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dy, idx,
                           dx, total, W, OH, OW);
        SCAT_LAUNCH_CHECK("scat_maxpool3x3s2_bwd");
        return SCAT_OK;
    }
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(cdiv((int64_t)H * W, 256 * 4), B * C), dim3(256), 0,
                       (hipStream_t)stream, dy, idx, dx, H, W, OH, OW);
    SCAT_LAUNCH_CHECK("scat_maxpool3x3s2_bwd");
    return SCAT_OK;
}

extern "C" int scat_avgpool_fwd(const float* x, float* y, int B, int C, int HW, int relu, void* stream) {
    SCAT_REQUIRE(x && y && B > 0 && C > 0 && HW > 0, SCAT_E_ARG, "scat_avgpool_fwd: bad argument");
    int64_t rows = (int64_t)B * C;
    hipLaunchKernelGGL(avgpool_fwd_kernel, dim3((int)((rows + 15) / 16)), dim3(256), 0, (hipStream_t)stream, x, y,
                       rows, HW, relu);
    SCAT_LAUNCH_CHECK("scat_avgpool_fwd");
    return SCAT_OK;
}

extern "C" int scat_avgpool_bwd(const float* dy, const float* y, int relu, float* dx, int B, int C, int HW,
                                int accumulate, void* stream) {
    SCAT_REQUIRE(dy && y && dx && B > 0 && C > 0 && HW > 0, SCAT_E_ARG, "scat_avgpool_bwd: bad argument");
    int64_t total = (int64_t)B * C * HW;
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dy, y, relu, dx,
                       total, HW, accumulate);
    SCAT_LAUNCH_CHECK("scat_avgpool_bwd");
    return SCAT_OK;
}

extern "C" int scat_subsample2(const float* x, float* y, int B, int C, int H, int W, void* stream) {
    SCAT_REQUIRE(x && y && B > 0 && C > 0 && H > 0 && W > 0, SCAT_E_ARG, "scat_subsample2: bad argument");
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const int64_t rows = (int64_t)B * C * OH;
    if (W % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 7) == 0) {
        const int64_t total = rows * (OW / 2);
        hipLaunchKernelGGL(subsample2_kernel<2>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, total, H,
                           W, OH, OW);
    } else {
        const int64_t total = rows * OW;
        hipLaunchKernelGGL(subsample2_kernel<1>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, total, H,
                           W, OH, OW);
    }
    SCAT_LAUNCH_CHECK("scat_subsample2");
    return SCAT_OK;
}

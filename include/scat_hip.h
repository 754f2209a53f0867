/* libscat_hip — C ABI of the MI355X (gfx950) SCAT hot path.
 *
 * Every function: plain pointers to DEVICE memory owned by the caller (PyTorch-ROCm
 * allocations, passed as tensor.data_ptr()), explicit int sizes, a caller-provided
 * workspace where one is needed, and a hipStream_t passed as void*.  Stream-ordered,
 * never synchronises, never allocates, retains no pointer.  Returns 0 or a negative
 * SCAT_E_* code; scat_last_error() gives the thread-local message.  fp32 everywhere.
 *
 * Each entry point names the reference call it stands in for (paths relative to the
 * reference checkout tomguluson92/SCAT).  The reference has no FFI of its own (it is pure
 * torch.nn); the binding a maintainer adds is the ctypes stub in INTEGRATION.md.
 */
#ifndef SCAT_HIP_H
#define SCAT_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SCAT_OK 0
#define SCAT_E_SHAPE (-1)     /* bad / unsupported dimensions */
#define SCAT_E_ARG (-2)       /* null pointer, bad flag */
#define SCAT_E_WORKSPACE (-3) /* workspace too small */
#define SCAT_E_LAUNCH (-4)    /* hipError at launch (message carries hipGetErrorString) */
#define SCAT_E_ARCH (-5)      /* not a gfx950 device */

int scat_version(void);
const char* scat_last_error(void);
/* label of the contraction-engine instantiation the last conv/gemm call on this thread launched
 * (tile shape + loaders), for per-kernel roofline accounting in bench.py */
const char* scat_last_kernel(void);
/* 0 if the current device is gfx950, else SCAT_E_ARCH. */
int scat_check_device(void);

/* ---- convolution: nn.Conv2d(bias=False) at models/resnet.py:65-72,105,129; hand_net.py:329 ----
 * x[B,Cin,H,W] NCHW, w[Cout,Cin,KH,KW], y[B,Cout,OH,OW]; KH=KW in {1,3,7}, stride in {1,2}.
 * in_scale/in_shift (per input channel, nullable) + in_relu fuse the PREVIOUS BatchNorm(+ReLU)
 * into the operand load: the conv sees relu(x*scale+shift); zero padding stays zero. */
int scat_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int H, int W,
                    int Cout, int KH, int KW, int stride, int pad, const float* in_scale, const float* in_shift,
                    int in_relu, void* stream);
/* dx[B,Cin,H,W] (+)= conv_transpose(dy[B,Cout,OH,OW], w).  wt = scat_conv2d_wt() output. */
int scat_conv2d_dgrad(const float* dy, const float* wt, float* dx, int B, int Cin, int H, int W, int Cout, int KH,
                      int KW, int stride, int pad, int accumulate, void* stream);
/* Stride-2 data-gradient decomposed by input-pixel parity (1x1/pad 0 and 3x3/pad 1): four stride-1
 * contractions over only the taps that can contribute (1+2+2+4 of 9), i.e. no MFMA work on structural
 * zeros.  Takes the ORIGINAL weights w[Cout,Cin,KH,KW]; ws: scat_conv2d_dgrad_s2_ws() bytes. */
int64_t scat_conv2d_dgrad_s2_ws(int Cin, int Cout, int KH, int KW);
int scat_conv2d_dgrad_s2(const float* dy, const float* w, float* dx, int B, int Cin, int H, int W, int Cout, int KH,
                         int KW, int pad, int accumulate, void* ws, int64_t ws_bytes, int w_ready, void* stream);
/* How the contraction kernels that support it form fp32 products (scat_conv3x3_s1 today):
 *   0  v_mfma_f32_32x32x2_f32 — the fp32 matrix instruction (64 FLOP/clk/SIMD);
 *   1  each fp32 operand split into three bf16 terms (round-to-nearest), six v_mfma_f32_32x32x16_bf16 products
 *      per fp32 product (hi.hi, hi.mid, mid.hi, hi.lo, mid.mid, lo.hi), fp32 accumulation: the dropped terms are
 *      <= 2^-25 of the product, below one fp32 rounding.  Default; SCAT_MATH=f32 selects 0 at load time. */
int scat_get_math_mode(void);
int scat_set_math_mode(int mode);
/* 3x3 / stride 1 / pad 1 with an LDS-resident halo (every input element is fetched once per tile instead of
 * once per tap): transposed = 0 -> dst[B,Cout,H,W] = conv(relu(src*scale+shift), w), src[B,Cin,H,W];
 * transposed = 1 -> dst[B,Cin,H,W] (+)= data gradient from src = dy[B,Cout,H,W].  w is the forward weight
 * [Cout,Cin,3,3]; ws: scat_conv3x3_s1_ws() bytes (re-laid weights).  Needs W <= 63.
 * Replaces nn.Conv2d(3, padding=1) at models/resnet.py:68-69 and its autograd. */
int64_t scat_conv3x3_s1_ws(int Cout, int Cin);
int scat_conv3x3_s1(const float* src, const float* w, float* dst, int B, int Cin, int H, int W, int Cout,
                    int transposed, const float* in_scale, const float* in_shift, int in_relu, int accumulate,
                    void* ws, int64_t ws_bytes, int w_ready, void* stream);
/* Forward conv (1x1 / 3x3, stride 1 or 2) on the split-operand taps kernel: contraction ordered (tap, channel),
 * weights re-laid and split per call into ws (scat_conv2d_fwd_split_ws bytes).  Needs scat_get_math_mode() == 1 and
 * Cin % 16 == 0.  The library's path for the stride-2 convolutions (models/resnet.py:68,131-135). */
int64_t scat_conv2d_fwd_split_ws(int Cout, int Cin, int KH, int KW);
int scat_conv2d_fwd_split(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int H, int W,
                          int Cout, int KH, int KW, int stride, int pad, const float* in_scale, const float* in_shift,
                          int in_relu, void* ws, int64_t ws_bytes, int w_ready, void* stream);
/* One transformer layer's qkv projection + softmax attention in one launch (models/vision_transformer.py:61-76):
 * qkv[B*n,3*heads*64] = h[B*n,dim] . wqkv^T; attn[B,heads,n,n] = softmax(scale q k^T); ao[B*n,heads*64] = attn . v
 * ('b n (h d)').  One workgroup per (block of 128/n images, head): the block's projection stays in LDS for its
 * attention.  n <= 32, dim % 4 == 0, head dim 64, scat_get_math_mode() == 1.  ws: scat_vit_qkv_attn_fwd_ws bytes. */
int64_t scat_vit_qkv_attn_fwd_ws(int dim, int heads);
int scat_vit_qkv_attn_fwd(const float* h, const float* wqkv, float* qkv, float* attn, float* ao, int B, int n, int dim,
                          int heads, float scale, void* ws, int64_t ws_bytes, void* stream);
/* The ResNet stem, Conv2d(3, Cout, 7, stride 2, padding 3, bias=False) (models/resnet.py:105), on split-operand
 * products: contraction over (kh, c, kw) with kw padded to 8, so a k-octet is 8 consecutive input pixels.
 * y[B,Cout,OH,OW]; ws: scat_conv7x7_s2_fwd_split_ws(Cout) bytes.  Needs scat_get_math_mode() == 1. */
int64_t scat_conv7x7_s2_fwd_split_ws(int Cout);
int scat_conv7x7_s2_fwd_split(const float* x, const float* w, float* y, int B, int H, int W, int Cout, void* ws,
                              int64_t ws_bytes, void* stream);
/* Weight gradient of that stem convolution: dw[64,3,7,7] from dy[B,64,OH,OW] and x[B,3,H,W]; one workgroup walks whole
 * output rows with the 7 x 3 input rows they touch in LDS, both MFMA operands split in registers, deterministic split
 * over output rows + fixed-order reduce.  Cout = 64, OW % 16 == 0, OW <= 112.  ws: ..._ws(B, H, W) bytes. */
int64_t scat_conv7x7_s2_wgrad_split_ws(int B, int H, int W);
int scat_conv7x7_s2_wgrad_split(const float* dy, const float* x, float* dw, int B, int H, int W, int Cout, void* ws,
                                int64_t ws_bytes, void* stream);
/* Pointwise (1x1, stride 1, pad 0) conv: dst[B,M,HW] (+)= A[M,C] . relu(src[B,C,HW]*scale+shift) (+ bias[M]).
 * Weights go straight from L2 to the MFMA operand registers, activations through LDS 32 channels per barrier.
 * transposed = 0 (forward): w = [M,C] = the conv weight [Cout,Cin];  transposed = 1 (data gradient): w = [C,M] is
 * still the forward weight (M = Cin, C = Cout) and src = dy.  ws: scat_conv1x1_s1_ws(M, C) bytes for the re-laid
 * weights.  Needs C % 16 == 0 and 16-B aligned w/src/ws.  Replaces nn.Conv2d(k=1) at models/resnet.py:65-72 and
 * its autograd. */
int64_t scat_conv1x1_s1_ws(int M, int C);
int scat_conv1x1_s1(const float* src, const float* w, float* dst, int B, int C, int HW, int M, int transposed,
                    const float* bias, const float* in_scale, const float* in_shift, int in_relu, int accumulate,
                    void* ws, int64_t ws_bytes, int w_ready, void* stream);
/* ---- activations as pre-split bf16 planes ("P8" layout) ----
 * The split-operand kernels form an fp32 product from the three bf16 terms hi + mid + lo of each operand; an activation
 * tensor can be handed over already split:  planes[p][n][c / 8][pixel][c % 8]  (bf16; p = hi, mid, lo; C % 8 == 0;
 * scat_planes_bytes(B, C, HW) bytes, 16-B aligned).  hi + mid + lo is exactly the fp32 value.
 * scat_planes_from_f32: planes = split(relu?(src[B,C,HW] * scale + shift)) (scale / shift optional, per channel) — the
 * fused BatchNorm + ReLU of models/resnet.py:85-86,89-90 applied once instead of in every consumer's operand load.
 * scat_conv1x1_planes: scat_conv1x1_s1 with the activations given as planes (no input transform: it is already in
 * them); the activation path is LDS-DMA only (buffer_load ... lds), bit-identical results.  C % 32 == 0.
 * lds_stages: depth of the LDS ring the activations land in (2: 48 KB, three workgroups per CU; 3: 72 KB, two, one
 * more stage in flight); 0 = the library's choice.
 * Replaces nn.Conv2d(k=1) forward / input gradient at models/resnet.py:65-72,84-92. */
int64_t scat_planes_bytes(int B, int C, int HW);
int scat_planes_from_f32(const float* src, void* planes, int B, int C, int HW, const float* in_scale,
                         const float* in_shift, int in_relu, void* stream);
int scat_conv1x1_planes(const void* planes, const float* w, float* dst, int B, int C, int HW, int M, int transposed,
                        const float* bias, int accumulate, void* ws, int64_t ws_bytes, int w_ready, int lds_stages,
                        void* stream);
/* ---- persistent stream-K schedule for the pointwise kernels ----
 * One tile per workgroup leaves the last round of a launch half empty at batch 96 (588 / 1 176 tiles of 128 x 128 on the
 * 768 workgroup slots of the chip).  scat_streamk_arm(buf, bytes) before a call of scat_conv1x1_s1 / scat_conv1x1_s1_bnb
 * lends its kernel a scratch buffer (scat_streamk_bytes(), 16-B aligned, zero-filled when first handed over, used by one
 * stream at a time): the launch then runs as one persistent grid whose workgroups share the (tile, 32-channel stage)
 * list equally and exchange the partial sums of split tiles through that buffer — deterministic (a tile's partial sums
 * are added in a fixed order that depends on the shape only), same epilogues.  The arming is per host thread and is
 * consumed by the next pointwise call whether or not it qualifies (>= 256 tiles, no taps, split products).
 * scat_streamk_error(buf, bytes, stream) SYNCHRONISES the stream and reports whether a workgroup ever gave up waiting
 * for a partial tile (a diagnostic for tests; the wait is bounded so a fault can not hang the queue).  No reference
 * counterpart: scheduling of nn.Conv2d(k=1), models/resnet.py:65-72. */
int64_t scat_streamk_bytes(void);
int scat_streamk_arm(void* buf, int64_t bytes);
int scat_streamk_error(const void* buf, int64_t bytes, void* stream);
/* ---- prepared weights (split-operand products) ----
 * The five entry points above that take `w_ready` re-lay their weights into ws (three bf16 planes in MFMA operand
 * order) before their main kernel: one small launch per convolution and direction, 114 per ResNet-50 train step
 * (53 nn.Conv2d of models/resnet.py x forward + data gradient).  With w_ready = 1 the caller promises that ws
 * already holds that re-layout and the launch is skipped.  scat_wprep_jobs() writes the description of one
 * (weight, kind) re-layout — 1 job, or up to 4 for the parity classes of the stride-2 data gradient — into HOST
 * memory at jobs_out (scat_wprep_job_bytes() each), numbered from block blk0, and returns the first free block (< 0:
 * SCAT_E_*); the caller concatenates the jobs of a whole network, uploads the table once and runs it with ONE
 * launch after each weight update (scat_wprep_run).  ws must be the buffer later passed to the entry point
 * (16-B aligned, the entry's *_ws() bytes); Cout/Cin/KH/KW/pad are those of w[Cout,Cin,KH,KW].  The library
 * keeps no state: the table and the workspaces belong to the caller. */
#define SCAT_WPREP_CONV1X1_FWD 0   /* scat_conv1x1_s1, transposed = 0 */
#define SCAT_WPREP_CONV1X1_DGRAD 1 /* scat_conv1x1_s1, transposed = 1; scat_conv1x1_s1_bnb */
#define SCAT_WPREP_CONV3X3_FWD 2   /* scat_conv3x3_s1, transposed = 0 */
#define SCAT_WPREP_CONV3X3_DGRAD 3 /* scat_conv3x3_s1, transposed = 1 */
#define SCAT_WPREP_FWD_SPLIT 4     /* scat_conv2d_fwd_split */
#define SCAT_WPREP_DGRAD_S2 5      /* scat_conv2d_dgrad_s2 */
int64_t scat_wprep_job_bytes(void);
int64_t scat_wprep_jobs(int kind, const float* w, void* ws, int64_t ws_bytes, int Cout, int Cin, int KH, int KW, int pad,
                        int64_t blk0, void* jobs_out, int max_jobs, int* njobs_out);
int scat_wprep_run(const void* jobs_dev, int njobs, int64_t nblocks, void* stream);
/* wt[Cin][Cout*KH*KW] = w[Cout][Cin][KH][KW] re-laid for the data-gradient contraction. */
int scat_conv2d_wt(const float* w, float* wt, int Cout, int Cin, int KH, int KW, void* stream);
/* dw[Cout,Cin,KH,KW] = sum over pixels dy * relu(x*scale+shift).  Deterministic two-stage
 * split-K (no float atomics).  ws: scat_conv2d_wgrad_ws() bytes. */
int64_t scat_conv2d_wgrad_ws(int B, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad);
/* ---- deferred split-K reduces ----
 * Every weight-gradient entry point ends with the fixed-order sum of its split-K slabs: 54 launches of 6-12 us in a
 * ResNet-50 backward.  After scat_splitk_defer(1) (per host thread) those sums are recorded instead of launched — the
 * caller then owes every recorded contraction a workspace of its own until scat_splitk_reduce_flush(stream) performs
 * them all with one grouped launch (per 48 jobs) on that stream, which must be ordered after the contractions.
 * scat_splitk_reduce_pending() = recorded and not yet flushed; scat_splitk_reduce_discard() forgets them (start of a
 * new backward after an aborted one).  Deterministic: a job's result depends on its split count only.  No reference
 * counterpart (autograd of nn.Conv2d, models/resnet.py:65-72). */
int scat_splitk_defer(int on);
int scat_splitk_reduce_pending(void);
int scat_splitk_reduce_discard(void);
int scat_splitk_reduce_flush(void* stream);

int scat_conv2d_wgrad(const float* dy, const float* x, float* dw, int B, int Cin, int H, int W, int Cout, int KH,
                      int KW, int stride, int pad, const float* in_scale, const float* in_shift, int in_relu,
                      void* ws, int64_t ws_bytes, void* stream);

/* ---- generic fp32 GEMM: nn.Linear / einsum at models/vision_transformer.py:33-35,55-57,61,75;
 *      models/resnet.py:116; hand_net.py:353 ----
 * C[M,N] (+)= op(A)[M,K] * op(B)[K,N] (+ bias).  A(i,k) = a[i*a_si + k*a_sk], B(k,j) = b[k*b_sk + j*b_sj],
 * C(i,j) = c[i*c_si + j*c_sj]; one of each stride pair must be 1.  bias_mode 0 none, 1 bias[i], 2 bias[j].
 * ws may be NULL (no split-K). */
int64_t scat_gemm_ws(int M, int N, int K);
int scat_gemm(const float* a, int64_t a_si, int64_t a_sk, const float* b, int64_t b_sk, int64_t b_sj, float* c,
              int64_t c_si, int64_t c_sj, int M, int N, int K, const float* bias, int bias_mode, int accumulate,
              void* ws, int64_t ws_bytes, void* stream);
/* n <= 16 independent contractions c_q[M_q,N_q] = A_q . B_q in ONE launch (+ one fixed-order split-K reduce): the twelve
 * weight gradients of the token mixer's backward (to_qkv / to_out / FeedForward of the three layers,
 * models/vision_transformer.py:33-35,52-57 through autograd), each too small to fill the chip on its own.  Operand
 * layout of a weight gradient only: A and B contiguous along their output index (a_si == 1, b_sj == 1), the same
 * contraction length K for every problem.  `problems` is a HOST array; ws: scat_gemm_group_ws() bytes, 16-B aligned. */
typedef struct ScatGemmProblem {
    const float* a;
    int64_t a_si, a_sk;
    const float* b;
    int64_t b_sk, b_sj;
    float* c;
    int64_t c_si, c_sj;
    int M, N, K;
} ScatGemmProblem;
int64_t scat_gemm_group_ws(const ScatGemmProblem* problems, int n);
int scat_gemm_group(const ScatGemmProblem* problems, int n, void* ws, int64_t ws_bytes, void* stream);

/* The same contraction on split-operand products (scat_get_math_mode() == 1), for the dense projections of the ViT
 * blocks (models/vision_transformer.py:52,57,76: to_qkv, to_out and their gradients; FeedForward :33-35):
 * c[M,N] (+)= op(a)[M,K] . b[K,N] (+ bias_n[N]); b and c row-major; a is [M,K] row-major, or stored [K,M] when
 * a_transposed.  ws: scat_gemm_split_ws(M, K) bytes.  scat_transpose2d: dst[C,R] = src[R,C] (the W^T a forward
 * projection needs as its b operand). */
int64_t scat_gemm_split_ws(int M, int K);
int scat_gemm_split(const float* a, int a_transposed, const float* b, float* c, int M, int N, int K, const float* bias_n,
                    int accumulate, void* ws, int64_t ws_bytes, void* stream);
int scat_transpose2d(const float* src, float* dst, int R, int C, void* stream);

/* ---- BatchNorm2d, training + inference: models/resnet.py:68-73,108,131 (eps 1e-5, momentum .1) ----
 * stats: per-channel batch mean / biased variance (fp64 accumulation, fixed reduction order), folded
 * into scale = gamma*invstd, shift = beta - mean*scale; running stats updated (unbiased var).
 * ws: scat_bn_ws(C) bytes. save_mean / save_invstd [C] are kept for backward. */
int64_t scat_bn_ws(int B, int C, int HW);
int scat_bn_train_stats(const float* x, int B, int C, int HW, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, float momentum, float eps, float* save_mean,
                        float* save_invstd, float* scale, float* shift, void* ws, int64_t ws_bytes, void* stream);
/* The same statistics without reading the convolution output back (the reference's nn.BatchNorm2d after nn.Conv2d,
 * models/resnet.py:65-73): scat_epilogue_stats_arm(buf, bytes) before a forward convolution call makes its kernel, if it
 * is one of the split-operand kernels, leave per-tile row sums (sum, sum of squares; fp32 over <= 128 pixels) in buf
 * ([C][groups][2] floats; bytes >= C * (ceil(B*OH*OW / 32) + 4) * 8 always suffices); scat_epilogue_stats_groups() right after
 * the call returns `groups` (0: that kernel does not write them — use scat_bn_train_stats) and disarms.  The state is per
 * host thread.  scat_bn_train_stats_partials sums the partials in fp64 in a fixed order and finishes as above. */
int scat_epilogue_stats_arm(float* buf, int64_t bytes);
int scat_epilogue_stats_groups(void);
/* The same with a per-channel reference c[C] (device pointer, e.g. the previous step's batch mean): the kernel leaves the
 * sums of (x - c) and (x - c)^2 instead — fp32 sums of x^2 over 32..128 pixels cancel catastrophically in
 * E[x^2] - mean^2 when |mean| >> sigma — and scat_bn_train_stats_partials_shifted(partials, groups, c, ...) finishes
 * mean = c + S1/N, var = S2/N - (S1/N)^2.  c must stay unchanged until that call has run. */
int scat_epilogue_stats_arm_shift(float* buf, int64_t bytes, const float* shift);
int scat_bn_train_stats_partials(const float* partials, int groups, int B, int C, int HW, const float* gamma,
                                 const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                 float* save_mean, float* save_invstd, float* scale, float* shift, void* stream);
int scat_bn_train_stats_partials_shifted(const float* partials, int groups, const float* stat_shift, int B, int C, int HW,
                                         const float* gamma, const float* beta, float* running_mean, float* running_var,
                                         float momentum, float eps, float* save_mean, float* save_invstd, float* scale,
                                         float* shift, void* stream);
/* inference: scale/shift from running stats */
int scat_bn_eval_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                      float eps, int C, float* scale, float* shift, void* stream);
/* y = [relu]( x*scale[c] + shift[c] [+ residual] ), or with res_scale/res_shift [+ residual*res_scale[c] +
 * res_shift[c]]: the shortcut branch's BatchNorm (models/resnet.py:92-96) applied while adding, its output never
 * materialised.  mask_out (optional, HW % 4 == 0): B*C*HW/4 bytes, bit q of byte e = (y[4e+q] > 0) — all the
 * backward needs of y, at 1/32 of its bytes. */
int scat_bn_apply(const float* x, const float* scale, const float* shift, const float* residual, const float* res_scale,
                  const float* res_shift, int relu, float* y, uint8_t* mask_out, int B, int C, int HW, void* stream);
/* backward of y = [relu](bn(x) [+res]):  g = dy * (mask);  mask from y_out>0 (if y_out), from the sign mask of
 * scat_bn_apply (if y_mask), else from x*scale+shift>0 (if relu), else 1.  Produces dgamma, dbeta, dx and (if dres)
 * dres (+)= g. */
int scat_bn_bwd(const float* dy, const float* x, const float* y_out, const uint8_t* y_mask, int relu, const float* scale,
                const float* shift, const float* save_mean, const float* save_invstd, const float* gamma,
                float* dgamma, float* dbeta, float* dx, float* dres, int dres_accumulate, int B, int C, int HW,
                void* ws, int64_t ws_bytes, void* stream);

/* the same backward for the stem (models/resnet.py:108-112 conv1 -> bn1 -> relu -> maxpool): dy is the gradient of the
 * 3x3 / stride-2 / pad-1 max-pool's OUTPUT [B,C,H/2,W/2] with its arg-max taps idx (scat_maxpool3x3s2_fwd); the scattered
 * full-resolution gradient is never written.  dx[B,C,H,W].  Even H, W % 4 == 0.  ws: scat_bn_ws(B, C, H*W). */
int scat_bn_bwd_maxpool(const float* dy_pooled, const int8_t* idx, const float* x, int relu, const float* scale,
                        const float* shift, const float* save_mean, const float* save_invstd, const float* gamma,
                        float* dgamma, float* dbeta, float* dx, int B, int C, int H, int W, void* ws, int64_t ws_bytes,
                        void* stream);
/* BatchNorm backward split in two so that its second half never touches memory: this call masks dy IN PLACE
 * (g = dy * mask; g is also the residual branch's gradient), reduces the per-channel sums and emits
 * coef3[3*C] = (ca | cb | cc) with  dx = ca*g + cb*x + cc;  the consumers of dx apply that while loading their
 * operand (scat_conv1x1_s1_bnb, scat_conv1x1_wgrad_bnb).  dy_add (nullable): a second contribution to the incoming
 * gradient, summed while loading (dy_g <- mask * (dy_g + dy_add)).  Needs HW % 4 == 0 and 16-B aligned tensors. */
int scat_bn_bwd_pre(float* dy_g, const float* dy_add, const float* x, const float* y_out, const uint8_t* y_mask, int relu, const float* scale,
                    const float* shift, const float* save_mean, const float* save_invstd, const float* gamma,
                    float* dgamma, float* dbeta, float* coef3, int B, int C, int HW, void* ws, int64_t ws_bytes,
                    void* stream);
/* The reduction of that first half inside the kernel that COMPLETES the gradient (models/resnet.py:93-96: the gradient of a
 * block output is the next block's conv1 data gradient accumulated onto the shortcut's gradient): armed before an
 * accumulating scat_conv1x1_s1 call, the kernel — if it is one that can (128-row tiles on split products; others ignore it)
 * — masks the completed gradient with the output's sign bits `mask` (scat_bn_apply's mask_out), stores the MASKED gradient
 * and leaves per channel and column group the sums of g and g * (x - mean[c]) in `part` ([C][groups][2] floats; bytes >=
 * C * (ceil(B*HW / 32) + 4) * 8 always suffices).  x: the raw output of the convolution in front of that BatchNorm, n floats
 * (= the size of the gradient tensor).  scat_epilogue_bnb_groups() right after the call returns the number of groups written
 * (0: not done, use scat_bn_bwd_pre) and disarms; scat_bn_bwd_pre_partials finishes coef3 / dgamma / dbeta from them.  Same
 * host-thread, one-shot protocol as scat_epilogue_stats_arm. */
int scat_epilogue_bnb_arm(const float* x, const uint8_t* mask, const float* mean, int64_t n, float* part, int64_t part_bytes);
int scat_epilogue_bnb_groups(void);
int scat_bn_bwd_pre_partials(const float* partials, int groups, int B, int C, int HW, const float* save_mean,
                             const float* save_invstd, const float* gamma, float* dgamma, float* dbeta, float* coef3,
                             void* stream);
/* dx[B,Cin,HW] (+)= w^T . (ca*g + cb*z + cc): data gradient of a 1x1 conv whose output gradient is the BatchNorm
 * backward above (g, coef3 from scat_bn_bwd_pre; z = the conv's raw output).  ws: scat_conv1x1_s1_ws(Cin, Cout). */
int scat_conv1x1_s1_bnb(const float* g, const float* z, const float* coef3, const float* w, float* dx, int B, int Cin,
                        int HW, int Cout, int accumulate, void* ws, int64_t ws_bytes, int w_ready, void* stream);
/* dw[Cout,Cin] = sum over pixels (ca*g + cb*z + cc)[co] * relu(x*scale+shift)[ci]: the same conv's weight gradient. */
int64_t scat_conv1x1_wgrad_bnb_ws(int B, int Cin, int HW, int Cout);
int scat_conv1x1_wgrad_bnb(const float* g, const float* z, const float* coef3, const float* x, float* dw, int B, int Cin,
                           int HW, int Cout, const float* in_scale, const float* in_shift, int in_relu, void* ws,
                           int64_t ws_bytes, void* stream);

/* ---- pooling: models/resnet.py:110 (MaxPool2d(3,2,1)), :115 (AvgPool2d(7)) ----
 * max-pool reads relu(x*scale+shift) when scale != NULL (stem BN fused); idx = argmax tap (int8). */
int scat_maxpool3x3s2_fwd(const float* x, const float* scale, const float* shift, int relu, float* y, int8_t* idx,
                          int B, int C, int H, int W, void* stream);
int scat_maxpool3x3s2_bwd(const float* dy, const int8_t* idx, float* dx, int B, int C, int H, int W, void* stream);
/* global average over HW then relu: y[B,C] */
int scat_avgpool_fwd(const float* x, float* y, int B, int C, int HW, int relu, void* stream);
int scat_avgpool_bwd(const float* dy, const float* y, int relu, float* dx, int B, int C, int HW, int accumulate,
                     void* stream);

/* y[B,C,ceil(H/2),ceil(W/2)] = x[:,:,::2,::2] — the pixels the 1x1/stride-2 shortcut convolution reads
 * (models/resnet.py:127-132), packed once so that its forward and weight gradient run at stride 1. */
int scat_subsample2(const float* x, float* y, int B, int C, int H, int W, void* stream);

/* ---- LayerNorm over the last dim (eps 1e-5): models/vision_transformer.py:20-26 ---- */
int scat_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                       int rows, int dim, float eps, void* stream);
int64_t scat_layernorm_bwd_ws(int rows, int dim);
/* dgamma == dbeta == NULL: the input gradient only (no parameter sums, ws unused) — the pose-length term's replay,
 * models/hand_net.py:396 */
int scat_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                       float* dx, float* dgamma, float* dbeta, int rows, int dim, void* ws, int64_t ws_bytes,
                       void* stream);

/* ---- attention core: models/vision_transformer.py:61-76 / models/vit.py:51-66 ----
 * qkv[B,n,3*h*d] (q|k|v, each 'b n (h d)'), out[B,n,h*d], attn[B,h,n,n] (saved softmax). n <= 128, d = 64. */
int scat_attention_fwd(const float* qkv, float* out, float* attn, int B, int n, int heads, int dim_head, float scale,
                       void* stream);
int scat_attention_bwd(const float* dout, const float* qkv, const float* attn, float* dqkv, int B, int n, int heads,
                       int dim_head, float scale, void* stream);

/* ---- performer (FAVOR+) linear attention core: models/vision_performer.py:34-53 ----
 * kqv[B,T,heads,3e] (k|q|v per head, one shared Linear, :17,47), w[m,e] frozen random features (:32);
 * y[B,T,heads*e].  Saved for backward: kp,qp[B,heads,T,m], kptv[B,heads,e,m], ksum[B,heads,m], D[B,heads,T]. */
int scat_performer_fwd(const float* kqv, const float* w, float* y, float* kp, float* qp, float* kptv, float* ksum,
                       float* D, int B, int T, int heads, int e, int m, void* stream);
int64_t scat_performer_bwd_ws(int B, int T, int heads, int e, int m);
int scat_performer_bwd(const float* dy, const float* kqv, const float* w, const float* y, const float* kp,
                       const float* qp, const float* kptv, const float* ksum, const float* D, float* dkqv, int B, int T,
                       int heads, int e, int m, void* ws, int64_t ws_bytes, void* stream);

/* ---- elementwise ---- */
/* exact-erf GELU (nn.GELU default), models/vision_transformer.py:34 */
int scat_gelu_fwd(const float* x, float* y, int64_t n, void* stream);
int scat_gelu_bwd(const float* dy, const float* x, float* dx, int64_t n, void* stream);
int scat_relu_fwd(const float* x, float* y, int64_t n, void* stream);
int scat_relu_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream);
/* inverted dropout, mask = counter hash of (seed, index): the same call with dy regenerates the mask for the
 * backward (vision_performer.py:18,28 Dropout(0.1), active in train mode) */
int scat_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, void* stream);
/* y = a + alpha*b */
int scat_axpy(const float* a, const float* b, float alpha, float* y, int64_t n, void* stream);
/* column sums: out[j] (+)= sum_i x[i*cols + j]  (bias gradients), fixed order */
int scat_colsum(const float* x, float* out, int rows, int cols, int accumulate, void* stream);
/* the same with a caller-owned scratch of scat_colsum_ws(rows, cols) bytes (0 for small inputs): tall matrices are summed in
 * row slices by ~1024 workgroups instead of cols / 16, slices added in index order (bias gradients of the token mixers:
 * models/vit.py:40-47 at HRNet's 24 672 x 196 tokens) */
int64_t scat_colsum_ws(int rows, int cols);
int scat_colsum_sliced(const float* x, float* out, int rows, int cols, int accumulate, void* ws, int64_t ws_bytes, void* stream);
/* n <= 16 independent column sums as ONE launch, each summed in scat_colsum's order (bit-identical to n calls): the bias
 * gradients of a token mixer's backward (to_out and the two FeedForward Linears of every layer,
 * models/vision_transformer.py:33-35,57).  `jobs` is a HOST array. */
typedef struct ScatColsumJob {
    const float* x;
    float* out;
    int rows, cols, accumulate;
} ScatColsumJob;
int scat_colsum_group(const ScatColsumJob* jobs, int n, void* stream);
/* tokens: y[B,T,D] = x[B,T,D] + pe[T,D], then rows t in masked[] <- mask_token[D]   (hand_net.py:366-373) */
int scat_tokens_fwd(const float* x, const float* pe, const float* mask_token, const int32_t* masked, int nmasked,
                    float* y, int B, int T, int D, void* stream);
int scat_tokens_bwd(const float* dy, const int32_t* masked, int nmasked, float* dx, float* dmask_token, int B, int T,
                    int D, void* stream);

/* input pipeline (dataset/load_STB.py:48-67: Resize(224), ToTensor, Normalize(.5,.5)): uint8 RGB image batch
 * (hwc = 1: [B,SH,SW,3] as decoded; 0: [B,3,SH,SW]) -> x/127.5-1 -> bilinear (align_corners=False) -> fp32
 * [B,3,OH,OW] in one pass */
int scat_preprocess_u8(const uint8_t* src, float* dst, int B, int SH, int SW, int OH, int OW, int hwc, void* stream);
/* nearest-neighbour upsample by an integer factor (models/hrnet.py:107) and mean over tokens
 * (hand_net.py:203 feat.mean(dim=1); vision_performer.py:108) */
int scat_upsample_nearest_fwd(const float* x, float* y, int B, int C, int H, int W, int factor, void* stream);
int scat_upsample_nearest_bwd(const float* dy, float* dx, int B, int C, int H, int W, int factor, void* stream);
/* one output of an HRNet exchange unit (models/hrnet.py:117-144: fuse_layers[i] summed, then ReLU) in one pass:
 * out[B,C,H,W] = relu?( sum_{j<n} term_j ),  term_j = sc_j[c] * in_j[b, c, y >> k_j, x >> k_j] + sh_j[c]  (sc_j = sh_j =
 * NULL: in_j as it is), in_j of shape [B, C, H >> k_j, W >> k_j]; terms are added in the order given.  n <= 4. */
int scat_fuse_sum(const float* in0, const float* in1, const float* in2, const float* in3, const float* sc0, const float* sc1,
                  const float* sc2, const float* sc3, const float* sh0, const float* sh1, const float* sh2, const float* sh3,
                  int k0, int k1, int k2, int k3, int n, float* out, int B, int C, int H, int W, int relu, void* stream);
int scat_token_mean_fwd(const float* x, float* y, int B, int T, int D, void* stream);
int scat_token_mean_bwd(const float* dy, float* dx, int B, int T, int D, void* stream);

/* ---- head: regressor loop + root-relative (hand_net.py:379-393) ----
 * pred0[b] = mean[P]; pred0[:,3:] += feat_out[b,P-3] (if feat_out); iter x: pred += [feat,pred]·W^T + bias;
 * root_relative: joints -= joint1.  preds[(iters+1),B,P] keeps every iterate for backward; out[B,P].
 * The same loop is the HRNet wrapper's head (hand_net.py:206-211: F=196, P=61, no offsets, not root-relative)
 * and ViP's (vision_performer.py:112-115). */
int scat_regressor_fwd(const float* feat, const float* feat_out, const float* mean, const float* w, const float* bias,
                       float* preds, float* out, int B, int F, int P, int iters, int root_relative, void* stream);
int scat_regressor_bwd(const float* dout, const float* feat, const float* preds, const float* w, float* dfeat,
                       float* dfeat_out, float* dw, float* dbias, int B, int F, int P, int iters, int root_relative,
                       void* ws, int64_t ws_bytes, void* stream);
int64_t scat_regressor_bwd_ws(int B, int F, int P, int iters);

/* ---- loss: train.py:165-203 (orthographic projection *112+112, MSE 3-D, L1 2-D) ----
 * out[B,66], gt3d[B,63] / gt2d[B,42] with row stride ld_gt; losses[3] = {loss, l3d, l2d}; dout[B,66]. */
int scat_loss_fwd_bwd(const float* out, const float* gt3d, const float* gt2d, int ld_gt, float w3d, float w2d,
                      float* losses, float* dout, int B, void* stream);
/* The pose-length regulariser as train.py:178-183 computes it from the network's third output pl_term[B,C,H,W]
 * (hand_net.py:395-396; it has no graph, so no backward):  len[b] = sqrt(mean_c sum_hw pl^2), l_pl = mean_b (len[b] -
 * 0.01 mean_b len)^2.  lens: B floats of scratch; l_pl: one float. */
int scat_pose_length_term(const float* pl_term, float* lens, float* l_pl, int B, int C, int HW, void* stream);

/* ---- Adam (torch.optim.Adam defaults, train.py:60): one launch over a flat parameter bucket ---- */
int scat_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
              int step, float grad_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif

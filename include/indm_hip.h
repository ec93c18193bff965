/* indm_hip.h -- C ABI of libindm_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the INDM hot path.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 data unless stated; tensors are dense/contiguous
 *   - activations are NHWC ([N,H,W,C]); convolution weights are RSCK ([KH,KW,Cin,Cout]); 1x1 weights [Cin,Cout]
 *   - the caller owns all memory (outputs and `ws` scratch); nothing is allocated or freed here
 *   - `stream` is a hipStream_t (NULL = default stream); kernels are only enqueued, never synchronised
 *   - return value: 0 = ok, INDM_E_SHAPE (-1), INDM_E_DTYPE (-2), INDM_E_LAUNCH (-3), INDM_E_ARG (-4)
 *   - no global state: every entry point is re-entrant per stream
 *
 * Each entry point names the reference interface (byeonghu-na/INDM, file:line) whose work it replaces.
 * The reference binds its two native ops through pybind11 (op/upfirdn2d.cpp:21-23, op/fused_bias_act.cpp:19-21)
 * and reaches everything else through ATen/cuDNN; see INTEGRATION.md for the reference-side bindings.
 */
#ifndef INDM_HIP_H
#define INDM_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define INDM_OK 0
#define INDM_E_SHAPE (-1)
#define INDM_E_DTYPE (-2)
#define INDM_E_LAUNCH (-3)
#define INDM_E_ARG (-4)

/* ---- reference native ops ------------------------------------------------------------------------------ */

/* up-FIR-down resampling.  Replaces upfirdn2d_op.upfirdn2d(input, kernel, up_x, up_y, down_x, down_y,
 * pad_x0, pad_x1, pad_y0, pad_y1) -- op/upfirdn2d.cpp:12-23, op/upfirdn2d_kernel.cu:107-369; semantics pinned
 * by upfirdn2d_native, op/upfirdn2d.py:159-200.  layout 0: x is [N,C,H,W] (the reference's [major,H,W,1] view),
 * layout 1: x is [N,H,W,C].  Output spatial size (in*up + pad0 + pad1 - k)/down + 1. */
int indm_upfirdn2d_f32(const float* x, const float* kernel, float* y, int N, int C, int in_h, int in_w, int kh,
                       int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0,
                       int pad_y1, int layout, void* stream);
/* the same op on bf16 tensors (the reference kernel's half instantiation, op/upfirdn2d_kernel.cu:311): fp32 taps and
 * accumulation, one rounding at the store */
int indm_upfirdn2d_bf16(const void* x, const float* kernel, void* y, int N, int C, int in_h, int in_w, int kh,
                       int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0,
                       int pad_y1, int layout, void* stream);

/* y = act(x + bias[(i/step_b) % size_b]) * scale.  Replaces fused.fused_bias_act(input, bias, refer, act, grad,
 * alpha, scale) -- op/fused_bias_act.cpp:11-21, op/fused_bias_act_kernel.cu:18-99 (act 1 linear, 3 leaky-ReLU;
 * grad 0/1/2).  bias / ref may be NULL. */
int indm_fused_bias_act_f32(const float* x, const float* bias, const float* ref, float* y, long n, long step_b,
                            int size_b, int act, int grad, float alpha, float scale, void* stream);

/* ---- dense contractions (fp32 MFMA implicit-GEMM engine) ----------------------------------------------- */

/* C[b] = alpha*(opA(A[b] .* gateA[b]) opB(B[b]) + bias[n] + bias2[m/rows_per_group][n]) + beta*R[b].
 * gateA (optional, A's layout) is multiplied into A inside the operand loader: the cos(2 pi p) gates of the explicit
 * vector-Jacobian chain (iresblock.py:253-273) never make a round trip through HBM.
 * Replaces ATen matmul/linear/einsum call sites: NIN models/layers.py:546-556, Dense models/layerspp.py:239-261,
 * attention einsums models/layerspp.py:95-99, LopCondConv2d 1x1 lipschitz.py:429-435, NICEMLPBlock blocks.py:36-43.
 * transA: A stored [K,M]; transB: B stored [N,K].  ws/ws_floats: optional split-K scratch (NULL = no split-K).
 * Size limits (INDM_E_SHAPE beyond them; they also hold for indm_conv2d_*): the epilogue addresses the residual and the
 * per-group bias with 32-bit element offsets -- M * ldr < 2^32 when R is given, (M / rows_per_group + 1) * N < 2^32 when bias2 is. */
int indm_gemm_f32(const float* A, const float* gateA, const float* B, float* C, int M, int N, int K, int transA,
                  int transB, long lda,
                  long ldb, long ldc, int batch, long strideA, long strideB, long strideC, float alpha,
                  const float* bias, const float* bias2, int rows_per_group, const float* R, long ldr,
                  long strideR, float beta, float* ws, long ws_floats, void* stream);

/* y = alpha*(conv(x,w) + bias[co] + bias2[n][co]) + beta*R.  Replaces F.conv2d call sites: ddpm_conv3x3
 * models/layers.py:113-124, LopConv2d.forward lipschitz.py:361-363, encoder convs resnet_batchnorm.py:8-10,
 * conv_downsample_2d up_or_down_sampling.py:178.  in_dil > 1 treats x as zero-inserted. */
int indm_conv2d_fwd_f32(const float* x, const float* w, float* y, int Nimg, int H, int W, int Cin, int Cout, int KH,
                        int KW, int stride, int pad, int in_dil, int OH, int OW, float alpha, const float* bias,
                        const float* bias2, const float* R, float beta, float* ws, long ws_floats, void* stream);

/* dx = alpha * conv_transpose(dy, w) + beta*R: the data gradient of indm_conv2d_fwd_f32 (autograd of F.conv2d in
 * the reference) and, as a forward op, the explicit vector-Jacobian product of the residual-flow g-net that
 * replaces torch.autograd.grad(g, x, v) at iresblock.py:257,269-271.  pad_t = K-1-pad_fwd, in_dil = fwd stride. */
int indm_conv2d_dgrad_f32(const float* dy, const float* w, float* dx, int Nimg, int H, int W, int Cin, int Cout,
                          int KH, int KW, int pad_t, int in_dil, int OH, int OW, float alpha, const float* R,
                          float beta, float* ws, long ws_floats, void* stream);

/* (a, d) = Sin(A B^T + bias[n] + bias2[m / rows_per_group][n]) in one launch: the second level of the residual flow's g-net
 * (1x1 LopConv2d / LopLinear -> Sin: lipschitz.py:321-369, layers/base/activations.py:7-12), the pre-activation never written.
 * A [M, K] row-major (lda), B stored [N, K] (ldb), a and d [M, N] (ldc).  Bit-identical to indm_gemm_f32 (transB = 1) followed by
 * indm_sincos_act_fwd_f32.  Returns 1 (not an error) when the problem is not the direct-to-LDS kernel's (M >= 1024, N >= 64,
 * K % 16 == 0, at least 224 tiles of 256 x 128, 16-byte aligned operands): the caller then runs the two steps. */
int indm_gemm_sin_f32(const float* A, const float* B, float* a, float* d, int M, int N, int K, long lda, long ldb, long ldc,
                      const float* bias, const float* bias2, int rows_per_group, void* stream);

/* One Sin level of the hand-written iResBlock backward (the reverse of lipschitz.py:321-369 -> layers/base/activations.py:7-12
 * inside iresblock.py:60-76) together with the GEMM that produces its incoming gradient: da = A B^T is NOT written,
 *   dp = da*d - 4 pi^2 * a * (du*t),  dt = du*d (dt may be NULL),  seg[s][n] = sum of dp over the rows of segment s (seg may be
 * NULL; else rows_per_seg % 64 == 0, M % rows_per_seg == 0 and ws >= indm_ires_bwd_act_ws_floats(M, N) floats).
 * A [M, K] (lda), B stored [N, K] (ldb), du, d, a, t, dp, dt [M, N] contiguous.  Bit-identical to indm_gemm_f32 (transB = 1)
 * followed by indm_ires_bwd_act_f32, segment sums included.  Returns 1 (not an error) when the direct-to-LDS kernel does not
 * take the problem (see indm_gemm_sin_f32; also M % 64 == 0): the caller then runs the two steps. */
int indm_gemm_ires_bwd_act_f32(const float* A, const float* B, const float* du, const float* d, const float* a, const float* t,
                               float* dp, float* dt, float* seg, int M, int N, int K, long lda, long ldb, long rows_per_seg,
                               float* ws, void* stream);

/* The same Sin level when the GEMM produces du (the chain's cotangent) instead of da: du = (A .* gateA) B is written, and
 *   dp = da*d - 4 pi^2 * a * (du*t),  seg[s][n] = per-segment sums of dp (seg may be NULL)
 * come out of the same launch (iresblock.py:60-76 reversed; activations.py:7-12).  A, gateA [M, K] (lda), B [K, N] row-major (ldb),
 * the other tensors [M, N] contiguous.  Bit-identical to indm_gemm_f32 with a gate followed by indm_ires_bwd_act_f32.  Returns 1
 * (not an error) when the gated direct-to-LDS kernel does not take the problem. */
int indm_gemm_gated_ires_bwd_act_f32(const float* A, const float* gateA, const float* B, const float* da, const float* d,
                                     const float* a, const float* t, float* du, float* dp, float* seg, int M, int N, int K,
                                     long lda, long ldb, long rows_per_seg, float* ws, void* stream);

/* dw[kh,kw,ci,co] = alpha * sum_pixels x * dy (+ beta*R): weight gradient of the convolution (autograd of
 * F.conv2d); the pixel reduction is split-K into ws. */
int indm_conv2d_wgrad_f32(const float* x, const float* dy, float* dw, int Nimg, int H, int W, int Cin, int Cout,
                          int KH, int KW, int stride, int pad, int OH, int OW, float alpha, const float* R,
                          float beta, float* ws, long ws_floats, void* stream);

/* 3x3 / pad 1 / stride 1 convolution whose INPUT has C <= 3 (or C <= 12, C % 4 == 0) channels -- LopConv2d at the
 * flow's image side, lipschitz.py:321-369 -- with the im2col gather inside the thin-K GEMM's operand loader:
 * y = alpha*(conv_sign(x, w[9C, Cout]) + bias[co] + bias2[n][co]) + beta*R; sign = -1 forward, +1 transposed conv.
 * Returns 1 (not an error) when the geometry is not covered; the caller then uses indm_tap_spread_f32 + indm_gemm_f32. */
int indm_conv3x3_thin_f32(const float* x, const float* w, float* y, int Nimg, int H, int W, int C, int Cout, int sign,
                          float alpha, const float* bias, const float* bias2, const float* R, float beta, void* stream);
int indm_conv3x3_thin_bf16(const float* x, const float* w, float* y, int Nimg, int H, int W, int C, int Cout, int sign,
                          float alpha, const float* bias, const float* bias2, const float* R, float beta, void* stream);

/* The same convolution with the flow's Sin activation (the g-net's first level, LopConv2d -> Sin: lipschitz.py:321-369,
 * layers/base/activations.py:7-12) as its epilogue:
 * a = sin(2 pi v) / (2 pi), d = cos(2 pi v), v = conv_sign(x, w) + bias[co] + bias2[n][co] (not written).  Bit-identical to
 * indm_conv3x3_thin_* followed by indm_sincos_act_fwd_f32.  Returns 1 (not an error) when the geometry is not covered
 * (needs C = 3 or 12, W = 16 or 32, Cout % 128 == 0, pixels % 32 == 0): the caller then runs the two steps. */
int indm_conv3x3_thin_act_f32(const float* x, const float* w, float* a, float* d, int Nimg, int H, int W, int C, int Cout,
                              int sign, const float* bias, const float* bias2, void* stream);
int indm_conv3x3_thin_act_bf16(const float* x, const float* w, float* a, float* d, int Nimg, int H, int W, int C, int Cout,
                               int sign, const float* bias, const float* bias2, void* stream);

/* Name of the kernel variant this thread's last dense launch (gemm / conv2d_*) was dispatched to, e.g.
 * "glds256x128,gated", "big256x256", "thin128x128", "128x32": bench.py labels its live per-kernel timings with it. */
const char* indm_igemm_last_variant(void);
/* "<abi major>.<minor> gfx950": major changes when an existing entry point changes, minor when entry points are added
 * (the pybind modules of op/upfirdn2d.cpp:22-23 / op/fused_bias_act.cpp:20-21 carry no version; a binding can check this one) */
const char* indm_version(void);

/* "bf16 compute / fp32 accumulate" variants (BASELINE configs #3-#5): identical contracts and fp32 tensors, but the
 * MFMA inputs are rounded to bf16 (round-to-nearest-even) and multiplied with v_mfma_f32_32x32x16_bf16. */
int indm_gemm_bf16(const float* A, const float* gateA, const float* B, float* C, int M, int N, int K, int transA,
                   int transB, long lda, long ldb, long ldc, int batch, long strideA, long strideB, long strideC,
                   float alpha, const float* bias, const float* bias2, int rows_per_group, const float* R, long ldr,
                   long strideR, float beta, float* ws, long ws_floats, void* stream);
int indm_conv2d_fwd_bf16(const float* x, const float* w, float* y, int Nimg, int H, int W, int Cin, int Cout, int KH,
                         int KW, int stride, int pad, int in_dil, int OH, int OW, float alpha, const float* bias,
                         const float* bias2, const float* R, float beta, float* ws, long ws_floats, void* stream);
int indm_conv2d_dgrad_bf16(const float* dy, const float* w, float* dx, int Nimg, int H, int W, int Cin, int Cout,
                           int KH, int KW, int pad_t, int in_dil, int OH, int OW, float alpha, const float* R,
                           float beta, float* ws, long ws_floats, void* stream);
int indm_conv2d_wgrad_bf16(const float* x, const float* dy, float* dw, int Nimg, int H, int W, int Cin, int Cout,
                           int KH, int KW, int stride, int pad, int OH, int OW, float alpha, const float* R,
                           float beta, float* ws, long ws_floats, void* stream);

/* ---- normalisation / softmax / reductions --------------------------------------------------------------- */

/* GroupNorm(G, C, eps) [+SiLU] [+Philox dropout], fused.  Replaces nn.GroupNorm + nn.SiLU + nn.Dropout at
 * models/layerspp.py:232,244,256,277-278, models/layers.py:38-39.  act 0 none / 1 SiLU.
 * The Philox key of the dropout mask is seed + (seed_dev ? *seed_dev : 0): seed_dev (may be NULL) points at a uint64 in DEVICE
 * memory, read when the kernel runs -- a launch captured into a hipGraph draws a fresh mask on every replay. */
long indm_groupnorm_ws_floats(int N, int HW, int C, int G);
int indm_groupnorm_fused(int HW, int C, int G); /* 1: single-pass register-resident kernels for this geometry, 0: stats / finalize / apply */
int indm_groupnorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                           float* rstd, int N, int HW, int C, int G, float eps, int act, float drop_p,
                           unsigned long long seed, const unsigned long long* seed_dev, float* ws, void* stream);
int indm_groupnorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* beta,
                           const float* mean, const float* rstd, float* dx, float* dgamma, float* dbeta, int N,
                           int HW, int C, int G, int act, float drop_p, unsigned long long seed,
                           const unsigned long long* seed_dev, float* ws, void* stream);

/* Training-mode BatchNorm2d (+ELU) of the flow encoder: nnet/resnets/resnet_batchnorm.py:23-76. */
int indm_bn_stats_f32(const float* x, float* mean, float* var, float* rstd, long R, int C, float eps, float* ws,
                      void* stream);
int indm_bn_apply_f32(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                      float* y, long R, int C, int act, void* stream);
int indm_bn_bwd_f32(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                    const float* beta, float* dx, float* dgamma, float* dbeta, long R, int C, int act, float* ws,
                    void* stream);

/* softmax(scale*x) over the last dim and its gradient: F.softmax at models/layerspp.py:97. */
int indm_softmax_fwd_f32(const float* x, float* y, long rows, int T, float scale, void* stream);
int indm_softmax_bwd_f32(const float* dy, const float* y, float* dx, long rows, int T, float scale, void* stream);

/* out[c] = alpha * sum_r f(x[r,c], y[r,c]) + beta*out[c]  (mode 0: x, 1: x*y, 2: x*x): bias gradients. */
long indm_colreduce_ws_floats(long R, int C);
int indm_colreduce_f32(const float* x, const float* y, float* out, long R, int C, int mode, float alpha, float beta,
                       float* ws, void* stream);

/* Segmented column sums out[s][c] = alpha * sum_{r < rows_per_seg} x[s*rows_per_seg + r][c]: gradient of the per-image
 * bias (the `+ Dense(temb)[:, :, None, None]` of layerspp.py:262-263 and the flow's conditioning bias lipschitz.py:433). */
long indm_segreduce_ws_floats(long nseg, long rows_per_seg, int C);
int indm_segreduce_f32(const float* x, float* out, long nseg, long rows_per_seg, int C, float alpha, float* ws,
                       void* stream);

/* out[b] = alpha * sum_i f(x[b,i], y[b,i])  (mode 0: x*y, 1: x*x, 2: x): Hutchinson traces
 * (iresblock.py:258,272; likelihood.py:37,206-213), prior_logp (sde_lib.py:165-169), Langevin norms. */
long indm_rowreduce_ws_floats(long B, long D);
int indm_rowreduce_f32(const float* x, const float* y, float* out, long B, long D, int mode, float alpha, float* ws,
                       void* stream);

/* ---- elementwise / layout -------------------------------------------------------------------------------- */

/* Sin activation of the residual flow and its derivative in one pass: a = sin(2 pi p)/(2 pi), d = cos(2 pi p)
 * (layers/base/activations.py:7-12); bwd: dp = da*d - 4 pi^2 * a * dd. */
int indm_sincos_act_fwd_f32(const float* p, float* a, float* d, long n, void* stream);
int indm_sincos_act_bwd_f32(const float* da, const float* dd, const float* a, const float* d, float* dp, long n,
                            void* stream);
/* One Sin level of the hand-written backward of an iResBlock (replaces autograd over iresblock.py:253-273's double
 * backward): with u = t .* d a link of the explicit VJP chain and a = Sin(p) the activation of g,
 *   dp = da*d - 4 pi^2 * a * (du*t);  dt_out = du*d (optional);  seg_out[s][c] = per-segment column sums of dp (optional:
 * the per-image conditioning-bias gradient; rows_per_seg % 64 == 0).  All tensors [R, C]; ws: indm_ires_bwd_act_ws_floats. */
/* Fused single-head attention of AttnBlockpp (models/layerspp.py:88-104): h = softmax(q k^T * scale) v with the
 * [T, T] logits kept in registers (q, k, v, h: [B, T, C]; lse: [B, T] log-sum-exp per query, saved for the backward).
 * bwd: Dv[B, T] = <dh[q], h[q]>.  Return 1 = geometry not covered (T not in {64,128,256} or C % 128): unfused path. */
int indm_attn_fwd_f32(const float* q, const float* k, const float* v, float* h, float* lse, int B, int T, int C,
                      float scale, void* stream);
int indm_attn_bwd_f32(const float* q, const float* k, const float* v, const float* dh, const float* lse,
                      const float* Dv, float* dq, float* dk, float* dv, int B, int T, int C, float scale, void* stream);
/* Back-to-back contraction of one link pair of the VJP chain: P[M, J] = ((A .* G) W2T .* D1) W1p[:, :J] with the
 * 512-wide intermediate kept on chip (A, G: [M, K]; W2T: [K, 512] row-major; D1: [M, 512]; W1p: [512, 32 if J <= 32 else 128]
 * zero-padded; J = 9*C <= 128; T2 optional [M, 512]: also stores the intermediate (A .* G) W2T, which the
 * gradient-carrying chain needs in its backward).  Returns 1 if the geometry is not covered (M % 64, K % 16, N != 512): the caller then
 * runs indm_gemm_f32 twice.  Replaces two torch.autograd.grad links of iresblock.py:257,269. */
int indm_ires_chain_f32(const float* A, const float* G, const float* W2T, const float* D1, const float* W1p, float* P,
                        float* T2, int M, int K, int N, int J, void* stream);
int indm_ires_chain_bf16(const float* A, const float* G, const float* W2T, const float* D1, const float* W1p, float* P,
                         float* T2, int M, int K, int N, int J, void* stream);
/* bf16 compute with W2T pre-packed once per flow call into bf16 MFMA-fragment order
 *   W2Tp[s][b][lane][e] = bf16(W2T[16 s + 8 (e >> 2) + 4 (lane >> 5) + (e & 3)][32 b + (lane & 31)]),  s < K/16, b < 16, e < 8
 * (512 KB, read straight into registers; A and the gate rotate through three 8 KB LDS buffers).  Same results as
 * indm_ires_chain_bf16. */
int indm_ires_chain_bf16w(const float* A, const float* G, const void* W2Tp, const float* D1, const float* W1p, float* P,
                          float* T2, int M, int K, int N, int J, void* stream);
long indm_ires_bwd_act_ws_floats(long R, int C);
int indm_ires_bwd_act_f32(const float* da, const float* du, const float* d, const float* a, const float* t, float* dp,
                          float* dt_out, float* seg_out, long R, int C, long rows_per_seg, float* ws, void* stream);
/* ---- bf16 STORAGE of the flow's [pixels, 512] tensors (BASELINE configs #3-#5; csrc/flow_f16.hip) ---------------
 * "F16" = bf16 in MFMA-fragment order: X[M, N] (M, N multiples of 32) is cut into 32 x 32 blocks (mt = m / 32, nb = n / 32),
 * a block into two 1 KB chunks s; chunk (mt, nb, s) holds for lane = 32 lhi + m % 32 and e < 8 the element
 * X[m][32 nb + 16 s + 8 (e >> 2) + 4 lhi + (e & 3)] at bf16 index (((mt N/32 + nb) 2 + s) 64 + lane) 8 + e -- the accumulator
 * layout of v_mfma_f32_32x32x16_bf16 and, unchanged, its input fragment for a contraction over n.  Packed weights:
 * Wp[ks][nblk][lane][e] = bf16(W[16 ks + 8 (e >> 2) + 4 (lane >> 5) + (e & 3)][32 nblk + (lane & 31)]) (the layout of
 * indm_ires_chain_bf16w's W2Tp) for a row-major W[K][Nw], Nw a multiple of 32 (zero-padded panels included).
 * The tensors of one iResBlock (iresblock.py:60-76 g-net, :253-273 estimators) never exist as fp32 in this mode. */
/* fp32 row-major [M, N] <-> F16 */
int indm_f16_pack_f32(const float* x, void* y, long M, int N, void* stream);
int indm_f16_unpack_f32(const void* x, float* y, long M, int N, void* stream);
/* 3x3 / pad 1 / stride 1 convolution of a skinny fp32 NHWC tensor (C <= 3, or C <= 12 with C % 4 == 0) -> F16
 * [Nimg H W, Cout] (LopConv2d at the flow's image side, lipschitz.py:321-369; sign -1 conv, +1 transposed conv with w
 * pre-packed as [(tap, co), ci]); w: [9 C, Cout] fp32.  epilogue 0: out1 = conv + bias; 1: (out1, out2) = (sin(2 pi p)/(2 pi),
 * cos(2 pi p)) of p = conv + bias (the Sin activation, activations.py:7-12: the pre-activation is never stored);
 * 2: out1 = (conv + bias) .* gate (F16), out2 (optional) = conv + bias.  Returns 1 if the geometry is not covered. */
int indm_f16_thin_conv(const float* x, const float* w, const float* bias, const void* gate, void* out1, void* out2,
                       int Nimg, int H, int W, int C, int Cout, int sign, int epilogue, void* stream);
/* The im2col image of a skinny fp32 NHWC tensor (C % 4 == 0, 9 C <= 512) as an F16 [Nimg H W, 512] operand:
 * out[pixel][(tap, c)] = x[pixel shifted by -sign (tap - 1)][c], zero columns behind 9 C.  The 48-channel scale of the 64 x 64
 * flows (CelebA, 9 C = 432): its 3x3 convolutions with the 512-wide side (LopConv2d, lipschitz.py:321-369) then run as
 * indm_f16_rows modes 0 / 1 / 4 with the weight's rows zero-padded to 512, their weight gradients as indm_f16_wgrad_gemm.
 * Returns 1 if not covered. */
int indm_f16_tap_spread(const float* x, void* out, int Nimg, int H, int W, int C, int sign, void* stream);
/* A workgroup owns 64 rows x 512 columns of an F16 A[M, 512] (M % 64 == 0):
 * mode 0  out1 = A Wp + bias[n] + bias2[m / rows_per_group][n]                        (F16; LopCondConv2d 1x1 lipschitz.py:429-435 and its data gradients)
 * mode 1  (out1, out2) = Sin activation of that product (a, cos)
 * mode 2  P[M, J] = ((A Wp) .* D1) Pp[:, :J]; out1 (optional) = A Wp                (the back-to-back VJP link of indm_ires_chain_*, iresblock.py:257,269)
 * mode 3  P[M, J] = A Pp[:, :J]                                                       (3x3 conv with a skinny OUTPUT before its tap gather)
 * mode 4  out1 = (A Wp + bias) .* D1, out2 (optional) = A Wp + bias                   (mode 0 with the gate epilogue of indm_f16_thin_conv)
 * mode 5  mode 1, and P[M, J] = out1 Pp[:, :J] in the same launch (J <= 128)           (the g-net's Sin -> skinny-output 3x3 conv: modes 1 + 3)
 * Wp: packed [512, 512]; Pp: packed [512, 32 | 128 | 448] panel (J <= 32 | J <= 128 | J <= 448), P fp32 row-major.
 * Returns 1 if not covered. */
int indm_f16_rows(int mode, const void* A, const void* Wp, const void* D1, const void* Pp, float* P, void* out1, void* out2,
                  const float* bias, const float* bias2, int M, int J, int rows_per_group, void* stream);

/* One series iteration of the log-det estimators (iresblock.py:257, 269-271: v -> v^T dg/dx up to the skinny tail) as ONE launch:
 *   P[M, J] = (((C3^T x) .* gate) W2Tp .* D1) Pp[:, :J],  M = Nimg H W, J = 9 C
 * = indm_f16_thin_conv(x, ., gate, epilogue 2) followed by indm_f16_rows mode 2 without the F16 [M, 512] tensor between them: the
 * workgroup builds its A tile from x and the d2 gate itself (bit-identical to the two launches).  x: fp32 NHWC [Nimg,H,W,C] with
 * C <= 3 or C in {4, 8, 12}; Wc: the transposed convolution's weight w[(tap, co), ci] (rows zero-padded to 16 S, S = ceil(9 C / 16))
 * packed as Wc[s][nb][lane][e] = bf16(w[16 s + 8 (lane >> 5) + e][32 nb + (lane & 31)]); gate, D1: F16 [M, 512]; W2Tp, Pp: as for
 * indm_f16_rows.  t2_out (optional, F16): (C3^T x .* gate) W2Tp, the tensor mode 2 stores through out1; t3_out (optional, needs
 * t2_out): the ungated C3^T x -- the training step's last link saves both for the backward.  Returns 1 if not covered. */
int indm_f16_chain_conv(const float* x, const void* Wc, const void* gate, const void* W2Tp, const void* D1, const void* Pp, float* P,
                        void* t3_out, void* t2_out, int Nimg, int H, int W, int C, int sign, int J, void* stream);

/* The same tile construction followed by the plain 64-row GEMM (the backward's reverse sweep, replaces autograd through
 * iresblock.py:264-273):  raw_out = conv(x, Wc), gated_out = raw_out .* gate, out = gated_out Wp  (all F16 [M, 512]; raw_out and
 * gated_out optional) = indm_f16_thin_conv(epilogue 2) + indm_f16_rows mode 0 in one launch, bit-identical.  sign as for
 * indm_f16_thin_conv; Wc as for indm_f16_chain_conv.  Returns 1 if not covered. */
int indm_f16_conv_gemm(const float* x, const void* Wc, const void* gate, const void* Wp, void* out, void* raw_out, void* gated_out,
                       int Nimg, int H, int W, int C, int sign, void* stream);

/* The g-net of an iResBlock (iresblock.py:60-76 behind the optional first Sin) as ONE launch, bit-identical to
 * indm_f16_thin_conv(epilogue 1) + indm_f16_rows mode 5:
 *   (a1, d1) = Sin(conv(x, W1) + b1);  (a2, d2) = Sin(a1 W2 + b2 + cb[m / rows_per_group]);  P[M, 9 C] = a2 W3p
 * x fp32 NHWC (C <= 3 or C in {4, 8, 12}); W1c: W1 as [(tap, c), 512] in indm_f16_chain_conv's Wc form; a1, d1, a2, d2: F16 [M, 512]
 * outputs (all four are saved for the backward); needs M >= 32768 (64-row tiles).  Returns 1 if not covered. */
int indm_f16_gnet(const float* x, const void* W1c, const float* b1, void* a1, void* d1, const void* W2p, const float* b2, const float* cb,
                  int rows_per_group, void* a2, void* d2, const void* W3p, float* P, int Nimg, int H, int W, int C, void* stream);

/* One Sin level of the hand-written iResBlock backward on F16 tensors (the F16 form of indm_ires_bwd_act_f32; replaces
 * autograd over iresblock.py:253-273's double backward): dp = da .* d - 4 pi^2 a .* (du .* t); seg (optional, fp32
 * [R / rows_per_seg][512]) = per-image column sums of dp.  rows_per_seg % 128 == 0 (indm_f16_bwd_act_ok), else returns 1. */
long indm_f16_bwd_act_ok(long R, long rows_per_seg);
int indm_f16_bwd_act(const void* da, const void* du, const void* d, const void* a, const void* t, void* dp, float* seg,
                     long R, long rows_per_seg, void* stream);
/* Weight gradients = contractions over the ROWS of F16 [M, 512] tensors (autograd of F.conv2d / the 1x1 convolutions of
 * lipschitz.py:361-363,429-435 w.r.t. their weights), deterministic split over the rows into ws (indm_f16_wgrad_ws_floats(M, T),
 * T = 0 for the GEMM form):
 *   thin:  out[t][n] = alpha sum_m thin[m][t] (wide .* gate)[m][n]   thin fp32 [M, T] (T <= 128: the tap-spread skinny side),
 *          wide / gate F16; transpose = 1 writes out[n][t]; ldo = row stride of out
 *   gemm:  out[n1][n2] = alpha sum_m (X .* gateX)[m][n1] Y[m][n2]     (transpose = 1: out[n2][n1])
 * M % 32 == 0; returns 1 if not covered. */
long indm_f16_wgrad_ws_floats(long M, int T);
int indm_f16_thin_wgrad(const float* thin, const void* wide, const void* gate, float* out, long M, int T, float alpha,
                        int transpose, long ldo, float* ws, long ws_floats, void* stream);
int indm_f16_wgrad_gemm(const void* X, const void* gateX, const void* Y, float* out, long M, float alpha, int transpose,
                        float* ws, long ws_floats, void* stream);

/* ---- bf16 STORAGE of the score network's activations ("H16", BASELINE configs #3-#5; csrc/unet_h16.hip) -------------
 * NHWC rows of bf16 (two values per dword); accumulators, biases, per-image embeddings, weight gradients and the fp32
 * master weights stay fp32.  The workloads: ResnetBlockBigGANpp's 3x3 / 1x1 convolutions (models/layerspp.py:225-287,
 * models/layers.py:88-124), NIN (models/layers.py:546-556), the NCSN++ graph of models/ncsnpp.py:251-414. */
/* Packed bf16 copies of the convolution weights, one launch for the whole network.  jobs: DEVICE array of
 * struct indm_h16_pack_job {const float* src; void* fwd; void* dg; int T, Cin, Cout, tile0, dg_ld, pad;}:
 *   src  fp32 RSCK [T][Cin][Cout] (1x1 / NIN: T = 1)
 *   fwd  bf16 [Cout][T][Cin]        -- B operand of the forward convolution (may be NULL)
 *   dg   bf16 [Cin][T][Cout], taps flipped: dg[ci][t][co] = src[T-1-t][ci][co] -- B operand of the data gradient (may be NULL)
 *   tile0 = running sum of T * ceil(Cin/32) * ceil(Cout/32) over the preceding jobs; ntiles = the total;
 *   dg_ld = row stride of dg in elements (0: T * Cout; wider: several weights side by side in one [Cin][sum Cout] pack, the
 *   q | k | v projections of AttnBlockpp, models/layerspp.py:88-93, as ONE data-gradient operand). */
struct indm_h16_pack_job { const float* src; void* fwd; void* dg; int T, Cin, Cout, tile0, dg_ld, pad; };
int indm_h16_pack_weights(const void* jobs, int njobs, int ntiles, void* stream);
/* y[n,oh,ow,co] = alpha * (sum x[n, oh s + kh - pad, ow s + kw - pad, ci] wp[co][(kh,kw,ci)] + bias[co] + bias2[n][co]) + beta * R
 * x bf16 NHWC, wp packed `fwd` form, y bf16 (out_f32 != 0: fp32, no R), R bf16 of y's shape (ddpm_conv3x3 / conv1x1 / NIN,
 * models/layers.py:88-124,546-556, with the temb bias, residual and 1/sqrt(2) of layerspp.py:262-287 in the epilogue).  The data
 * gradient of a stride-1 convolution is the same call on dy with the `dg` pack and pad' = K - 1 - pad; KH = KW = 1 is a plain
 * NT GEMM.  ws: fp32 split-K workspace; tile_counters: n_counters int32 words in device memory, ZERO at rest (the kernels put
 * them back), through which the last-arriving workgroup of a tile combines the K-slices inside the launch (NULL: a separate
 * reduce launch) -- one array per stream that runs these kernels concurrently.  Returns 1 (not an error) when the geometry is not covered (Cin % 32, Cout % 2,
 * OW / OH*OW not powers of two, >= 2 GB tensors, unaligned pointers). */
int indm_h16_conv2d(const void* x, const void* wp, void* y, int Nimg, int H, int W, int Cin, int Cout, int KH, int KW,
                    int stride, int pad, int OH, int OW, float alpha, const float* bias, const float* bias2,
                    const void* R, float beta, int out_f32, float* ws, long ws_floats, int* tile_counters, int n_counters,
                    void* stream);
/* dw[kh,kw,ci,co] = alpha * sum_{n,oh,ow} x[n, oh s + kh - pad, ow s + kw - pad, ci] dy[n,oh,ow,co]  (fp32 RSCK; autograd of
 * F.conv2d w.r.t. the weight): x, dy bf16 NHWC; deterministic split over the pixels into ws (>= KH KW Cin Cout floats).
 * Returns 1 when not covered (Cin % 8, Cout % 8, pixels % 32, OW / OH*OW not powers of two, >= 2 GB tensors). */
int indm_h16_conv2d_wgrad(const void* x, const void* dy, float* dw, int Nimg, int H, int W, int Cin, int Cout, int KH,
                          int KW, int stride, int pad, int OH, int OW, float alpha, float* ws, long ws_floats,
                          int* tile_counters, int n_counters, void* stream);
/* fp32 <-> bf16 (round to nearest even); n even, 16-byte aligned pointers; to_f32 scales by alpha */
int indm_h16_from_f32(const float* x, void* y, long n, void* stream);
int indm_h16_to_f32(const void* x, float* y, long n, float alpha, void* stream);

/* AttnBlockpp's single-head attention (models/layerspp.py:88-104) on bf16 q / k / v: h[B T, C] (bf16) = softmax(q k^T scale) v,
 * lse[B, T] (fp32) = log-sum-exp of the scaled logits per query.  q, k, v: bf16 matrices of row stride ld elements (ld = 3 C: the
 * three column blocks of a fused QKV product; ld = C: separate tensors); logits and probabilities never leave the registers.
 * Backward: dq, dk, dv (bf16, row stride ldg) from dh [B T, C], lse and D[B, T] = <dh[q], h[q]> (indm_h16_rowdot).
 * Return 1 when not covered (T not in {64, 128, 256}, C % 128, ld % 8, alignment): the caller takes the fp32-storage path. */
int indm_h16_attn_fwd(const void* q, const void* k, const void* v, long ld, void* h, float* lse, int B, int T, int C,
                      float scale, void* stream);
int indm_h16_attn_bwd(const void* q, const void* k, const void* v, long ld, const void* dh, const float* lse, const float* D,
                      void* dq, void* dk, void* dv, long ldg, int B, int T, int C, float scale, void* stream);
/* out[r] = <a[r], b[r]> for bf16 [R, C] (C even), fp32 out */
int indm_h16_rowdot(const void* a, const void* b, float* out, long R, int C, void* stream);
/* GroupNorm(min(C/4,32), C, eps) [+ SiLU] [+ dropout] on bf16 NHWC (models/layerspp.py:232-279, models/layers.py:38-39): x, y, dy,
 * dx bf16 [N, HW, C]; gamma, beta, mean, rstd [N, G], dgamma, dbeta fp32; same Philox dropout mask as indm_groupnorm_*_f32 for
 * the same seed.  indm_h16_groupnorm_ok: 1 when covered (C % 8 == 0, (C / G) % 4 == 0, a block of whole groups of <= 32 channels
 * fits one workgroup's registers); fwd / bwd return 1 otherwise.  ws (bwd): indm_h16_groupnorm_ws_floats(N, C) floats.
 * bwd: add (optional, bf16, x's shape) -- dx += add_scale * add in the same pass: the gradient that reaches x along the skip
 * connection of its residual block (layerspp.py:283-287), instead of a scale launch and an accumulate launch. */
int indm_h16_groupnorm_ok(int HW, int C, int G);
long indm_h16_groupnorm_ws_floats(int N, int C);
int indm_h16_groupnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                           int N, int HW, int C, int G, float eps, int act, float drop_p, unsigned long long seed,
                           const unsigned long long* seed_dev, void* stream);
int indm_h16_groupnorm_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean,
                           const float* rstd, void* dx, float* dgamma, float* dbeta, int N, int HW, int C, int G, int act,
                           float drop_p, unsigned long long seed, const unsigned long long* seed_dev, const void* add,
                           float add_scale, float* ws, void* stream);
/* out[s][c] = alpha * sum_{r < rps} x[(s rps + r) C + c], x bf16 [nseg rps, C], out fp32 [nseg, C]: the bias and per-image
 * temb-bias gradients of the convolutions (layerspp.py:262-263); C % 8 == 0, C <= 2048; ws: indm_h16_segreduce_ws_floats floats */
long indm_h16_segreduce_ws_floats(long nseg, long rps, int C);
int indm_h16_segreduce(const void* x, float* out, long nseg, long rps, int C, float alpha, float* ws, void* stream);
/* y = a x (bf16), n % 8 == 0;  out (bf16) = a x (fp32) + b y (bf16): an fp32-computed edge joining the bf16 trunk (the input
 * pyramid's skinny convolution, models/ncsnpp.py:339-347) */
int indm_h16_scale(const void* x, void* y, long n, float a, void* stream);
int indm_h16_axpby_mixed(const float* x, const void* y, void* out, long n, float a, float b, void* stream);
/* naive_upsample_2d / naive_downsample_2d (models/up_or_down_sampling.py:59-69) on bf16 NHWC, C % 8 == 0 */
int indm_h16_upsample2x(const void* x, void* y, int N, int H, int W, int C, float alpha, void* stream);
int indm_h16_downsample2x(const void* x, void* y, int N, int H, int W, int C, float alpha, void* stream);

/* act 1 SiLU (models/layers.py:38-39), 2 ELU (resnet_batchnorm.py:27) */
int indm_act_fwd_f32(const float* x, float* y, long n, int act, void* stream);
int indm_act_bwd_f32(const float* dy, const float* x, float* dx, long n, int act, void* stream);
int indm_axpby_f32(const float* x, const float* y, float* out, float a, float b, float c, long n, void* stream);
int indm_mul_f32(const float* x, const float* y, float* out, float alpha, long n, void* stream);
/* out1 = x*y1, out2 = x*y2 in one pass: the backward of a gated contraction (d(v.*g) -> dv = d.*g, dg = d.*v). */
int indm_mul2_f32(const float* x, const float* y1, const float* y2, float* out1, float* out2, long n, void* stream);
int indm_scale_rows_f32(const float* x, const float* s, float* out, long B, long D, void* stream);
/* naive_upsample_2d / naive_downsample_2d, models/up_or_down_sampling.py:59-69 (NHWC) */
int indm_upsample2x_f32(const float* x, float* y, int N, int H, int W, int C, float alpha, void* stream);
int indm_downsample2x_f32(const float* x, float* y, int N, int H, int W, int C, float alpha, void* stream);
/* squeeze / unsqueeze (pixel_shuffle), flows/resflow/layers/squeeze.py:27-45, NHWC with channel = c*4+dy*2+dx */
int indm_squeeze2x_f32(const float* x, float* y, int N, int H, int W, int C, void* stream);
int indm_unsqueeze2x_f32(const float* x, float* y, int N, int H, int W, int C4, void* stream);
/* NCHW <-> NHWC at the API edge (dir 0: NCHW->NHWC, 1: NHWC->NCHW) */
int indm_nchw_nhwc_f32(const float* x, float* y, int N, int C, int HW, int dir, void* stream);
/* Column blocks of x[R, Ctot] <-> one flat buffer holding block b as a contiguous [R, C_b] matrix at offset R*offs[b]
 * (offs: device int32[nblk+1]; dir 0: x -> flat, 1: flat -> x): the per-block slices of the batched Dense_0(act(temb))
 * product, models/layerspp.py:262-263 */
int indm_split_cols_f32(const float* x, float* y, const int* offs, int nblk, long R, int Ctot, int dir, void* stream);
/* nn.utils.weight_norm(Linear): w[o,:] = v[o,:] g[o] / ||v[o,:]||, and its gradient (nnet/weight_norm.py:8-40) */
int indm_weightnorm_fwd_f32(const float* v, const float* g, float* w, float* nrm, int O, int I, void* stream);
int indm_weightnorm_bwd_f32(const float* dw, const float* v, const float* g, const float* nrm, float* dv, float* dg,
                            int O, int I, void* stream);
/* Lop weight normalisation w / max(1, ||w[:,co]||_1 / coeff) and its gradient: LopConv2d.compute_weight,
 * layers/base/lipschitz.py:350-359 (domain = codomain = inf) */
int indm_lop_normalize_fwd_f32(const float* w, float* wn, float* l1, int R, int Co, float coeff, void* stream);
/* dst_i[0..n_i) = src_i[0..n_i) for a DEVICE table of count rows of three int64 {src, dst, n}: gathers the parameter
 * gradients autograd produced into the flat gradient buffer in one launch (replaces one AccumulateGrad add per
 * parameter; the flat buffer is what the single all-reduce and the fused optimiser of losses.py:51-60 read). */
int indm_multi_copy_f32(const void* table, int count, long max_n, void* stream);

/* The same for MANY weights in two launches each way (all 96 LopConv weights of the 16-16 flow are normalised once
 * per flow call).  table: DEVICE array of count rows of six int64 {w, aux, out, red, R, Co} -- forward: out = wn,
 * red = l1 (written); backward: aux = dwn, out = dw, red = scratch[Co], and l1_table[i] = the forward's l1. */
int indm_lop_normalize_multi_fwd_f32(const void* table, int count, int max_cols, long max_elems, float coeff,
                                     void* stream);
int indm_lop_normalize_multi_bwd_f32(const void* table, const void* l1_table, int count, int max_cols, long max_elems,
                                     float coeff, void* stream);
int indm_lop_normalize_bwd_f32(const float* dwn, const float* w, const float* l1, float* dw, float* dot, int R,
                               int Co, float coeff, void* stream);
/* 3x3 convolutions with a skinny channel side (flow edges 512 <-> 3/12, resflow_.py:442-476) are computed as ONE
 * MFMA GEMM against the tap-major weight ([pixels,wide] x [wide, 9*C]) followed by this gather over the 9 shifted
 * taps (sign +1 forward conv, -1 transposed conv); tap_spread is its adjoint (used for the weight gradient). */
int indm_tap_gather_f32(const float* P, const float* bias, float* y, int N, int H, int W, int C, int sign,
                        float alpha, void* stream);
/* the same gather with the series bookkeeping of neumann_logdet_estimator (iresblock.py:264-273) in the same pass:
 * y = alpha*(gather + bias) .* mulp (the block's input gate cos(2 pi x), optional); acc_out = acc_in + coeff*y (optional). */
int indm_tap_gather_acc_f32(const float* P, const float* bias, const float* mulp, float* y, const float* acc_in,
                            float* acc_out, float coeff, int N, int H, int W, int C, int sign, float alpha, void* stream);
int indm_tap_spread_f32(const float* y, float* P, int N, int H, int W, int C, int sign, float alpha, void* stream);
/* stop test of the fixed-point inverse: count of elements with (x-xp)^2/(atol+|y|rtol) >= 1, iresblock.py:81-82 */
int indm_fixed_point_check_f32(const float* x, const float* xp, const float* y, long n, float atol, float rtol,
                               float* out, void* stream);
/* log|det W| and W^-1, n <= 64: torch.slogdet(weight.to('cpu')), flows/permutation.py:107-108,131-132 */
int indm_slogdet_inv_f32(const float* W, int n, float* logdet, float* inv, void* stream);

/* ---- SDE / loss / sampler / optimiser --------------------------------------------------------------------- */

/* Affine coupling of the 64-d prior flow (NICE1d + Affine: flows/couplings/coupling.py:13-177, transform.py:49-81): split,
 * transform (scale = sigmoid(ls + 2) + 1e-3; inverse divides by scale + 1e-12), recombination and log-det row sum in one
 * launch.  x, y, prm = (mu | ls): [B, 2F]; logdet: [B]; split 0 continuous / 1 skip; up, inverse: 0/1. */
int indm_coupling_fwd_f32(const float* x, const float* prm, float* y, float* logdet, int B, int F, int split, int up,
                          int inverse, void* stream);
int indm_coupling_bwd_f32(const float* dy, const float* dlogdet, const float* x, const float* prm, float* dx, float* dprm,
                          int B, int F, int split, int up, int inverse, void* stream);
/* Standard-normal noise on the device (Philox4x32-10 + Box-Muller); replaces the torch.randn / randn_like calls of
 * losses.py:103, sampling.py:196,209,283 and iresblock.py:137.  `state` points to {seed, offset} (two uint64) in device
 * memory; the launch advances offset by ceil(n/4), so a captured hipGraph of a sampler step draws fresh noise per replay. */
int indm_philox_randn_f32(float* out, long n, void* state, void* stream);
/* out = mc[b]*x + sd[b]*noise: sde.marginal_prob + perturbation, sde_lib.py:153-157,284-287; losses.py:103-105 */
int indm_sde_perturb_f32(const float* x, const float* noise, const float* mc, const float* sd, float* out, long B,
                         long D, void* stream);
/* per-image DSM loss from the raw network output and its gradient: losses.py:106-118 with
 * score = -out/std (kind 0, models/utils.py:175-177) or out (kind 1) */
long indm_dsm_ws_floats(long B, long D);
int indm_dsm_loss_fwd_f32(const float* out, const float* z, const float* std, const float* w, float* loss, long B,
                          long D, int kind, int mode, int reduce_mean, float* ws, void* stream);
int indm_dsm_loss_bwd_f32(const float* gl, const float* out, const float* z, const float* std, const float* w,
                          float* dout, long B, long D, int kind, int mode, int reduce_mean, void* stream);
/* x_mean = a[b]*x + b[b]*s ; x_new = x_mean + c[b]*z: EulerMaruyama / ReverseDiffusion / Langevin updates,
 * sampling.py:191-197,205-210,289-290 */
int indm_affine_update_f32(const float* x, const float* s, const float* z, const float* a, const float* b,
                           const float* c, float* x_new, float* x_mean, long B, long D, void* stream);
/* Langevin step size from batch-mean norms, sampling.py:284-288 */
int indm_langevin_step_f32(const float* grad_sq, const float* noise_sq, const float* alpha, float* step,
                           float* noise_coef, int B, float snr, void* stream);
/* ---- block-level sampler / likelihood entry points (csrc/pc_step.hip): one call = one reference update_fn after the
 * score network ran.  x, net_out, z, x_new, x_mean: [B, D] fp32, D % 4 == 0, 16-byte aligned; f_lin, g|G, k, alpha: [B].
 * Noise: explicit z XOR the device-resident Philox state {seed, offset} of indm_philox_randn_f32 (drawn inside the update
 * kernel with the same counter -> value map and advanced by B D / 4); both NULL only for a probability-flow step. */
/* EulerMaruyamaPredictor.update_fn sampling.py:191-197 with RSDE.sde sde_lib.py:96-103; forward drift f = f_lin x,
 * diffusion g, score = k net_out (models/utils.py:167-177), dt = -1 / N */
int indm_pc_step_em_f32(const float* x, const float* net_out, const float* f_lin, const float* g, const float* k,
                        double dt, int probability_flow, const float* z, void* philox, float* x_new, float* x_mean,
                        long B, long D, void* stream);
/* ReverseDiffusionPredictor.update_fn sampling.py:205-210 with RSDE.discretize sde_lib.py:105-118; discretised forward
 * step f = f_lin x, G */
int indm_pc_step_rd_f32(const float* x, const float* net_out, const float* f_lin, const float* G, const float* k,
                        int probability_flow, const float* z, void* philox, float* x_new, float* x_mean, long B, long D,
                        void* stream);
/* LangevinCorrector.update_fn, one iteration, sampling.py:279-290 (batch-mean norms :286-288) */
long indm_pc_step_ws_floats(long B, long D);
int indm_pc_step_langevin_f32(const float* x, const float* net_out, const float* k, const float* alpha, float snr,
                              const float* z, void* philox, float* x_new, float* x_mean, float* ws, long B, long D,
                              void* stream);
/* get_div_fn likelihood.py:27-38: div[b] = sum_i grad_fn_eps[b,i] eps[b,i] (Hutchinson) */
long indm_hutch_div_ws_floats(long B, long D);
int indm_hutch_div_f32(const float* grad_fn_eps, const float* eps, float* div, long B, long D, float* ws, void* stream);
/* sum g^2 -> device scalar (torch.nn.utils.clip_grad_norm_, losses.py:58-59) */
long indm_sqnorm_ws_floats(void);
int indm_sqnorm_f32(const float* g, long n, float* out, float* ws, void* stream);
/* fused clip_grad_norm_ + AdamW + EMA on flat buffers: losses.py:30-62, models/ema.py:32-51 */
int indm_adamw_ema_f32(float* p, const float* g, float* m, float* v, float* ema, long n, float lr, float beta1,
                       float beta2, float eps, float weight_decay, int step, float clip, const float* gnorm_sq,
                       float gscale, float ema_one_minus_decay, void* stream);
int indm_ema_update_f32(float* s, const float* p, long n, float one_minus_decay, void* stream);

/* ---- block-level drivers: ONE call per residual-flow block ------------------------------------------------- */

/* The iResBlock of the residual flow, flows/resflow/layers/iresblock.py:14-174, as ONE call per direction: the driver
 * enqueues the whole launch sequence of the block (g-net: Sin -> thin 3x3 -> Sin -> conditioned 1x1 GEMM -> Sin -> thin
 * 3x3; then the K = n + 2 (training, neumann_logdet_estimator :264-273) or n + 20 (eval, basic_logdet_estimator
 * :253-261) vector-Jacobian iterations on the back-to-back chain kernel) on `stream` -- ~25-40 kernels -- instead of one
 * host call per kernel.  The kernels are the ones the op-level entry points above launch; only the SEQUENCING lives
 * here, so results are bit-identical to the op-level composition.
 * Geometry covered (return 1 otherwise, nothing enqueued; the caller composes the op-level calls): C <= 3 with
 * W | 128 and H*W % 128 == 0, or C % 4 == 0 with 9C <= 128 (bf16 = 2: C <= 48); idim == 512; N*H*W % 64 == 0 and >= 1024.
 * All tensors NHWC fp32; weights are the Lop-NORMALISED ones (lipschitz.py:350-359), in the layouts named below.
 * `coeffs` is a HOST array.  ws: device scratch of indm_iresblock_ws_floats() floats; splitk_ws: the dense engine's
 * split-K scratch (may be shared with other calls on the same stream). */
typedef struct {
  int N, H, W, C, idim; /* block input [N,H,W,C]; width of the g-net (512) */
  int preact;           /* the block starts with a Sin activation (every block but the flow's first) */
  int training;         /* 1: Neumann surrogate + tensors saved for indm_iresblock_bwd; 0: power series, nothing saved */
  int bf16;             /* 0: exact fp32 MFMA; 1: MFMA inputs rounded to bf16 (fp32 tensors, fp32 accumulate);
                         * 2: bf16 STORAGE as well -- a1, d1, a2, d2, t3, t2 are F16 buffers (bf16 in MFMA-fragment order,
                         * N H W idim elements each; see the F16 section above), the kernels are csrc/flow_f16.hip's, and the
                         * packed weights W2T_bf16w, W2_bf16w, panel_bf16w, W3_bf16w must be given */
  const float* x;       /* [N,H,W,C] */
  const float* cb;      /* [N, idim] folded conditioning bias Linear(h) W2 (lipschitz.py:429-435) */
  const float* W1;      /* [3,3,C,idim] RSCK == [(tap,c), idim] */
  const float* W2T;     /* [idim, idim] = W2^T stored row-major */
  const void* W2T_bf16w; /* bf16 mode: W2T in indm_ires_chain_bf16w's fragment order, or NULL */
  const float* W3_ci_tapco; /* [idim, (tap, c)] : W3 [3,3,idim,C] re-laid-out for the skinny-output conv */
  const float* W3_tapco_ci; /* [(tap, c), idim] : W3 re-laid-out for the transposed conv */
  const float* panel;   /* [idim, 32 | 128] zero-padded W1 as [idim, (tap,c)]: the chain kernel's second stage */
  const float *b1, *b2, *b3;
  const float* eps;     /* [N,H,W,C] Hutchinson probe (iresblock.py:137) */
  const float* coeffs;  /* HOST: training: (-1)^k c_k, k = 1..K; eval: (-1)^(k+1) c_k / k */
  int ncoeff;           /* K */
  float* y;             /* [N,H,W,C] = x + g(x) */
  float* logdet;        /* [N] */
  /* training = 1: saved for the backward, caller-allocated ([N,H,W,idim] unless stated); eval: ignored */
  float *a0, *d0;       /* [N,H,W,C], preact only */
  float *a1, *d1, *a2, *d2, *t3, *t2;
  float* c1;            /* [N,H,W,C], preact only */
  float* w;             /* [N,H,W,C] eps + sum_k coeffs[k] v_k */
  float* ws; long ws_floats;
  float* splitk_ws; long splitk_ws_floats;
  void* timer;          /* optional indm_timer handle: HIP events around the chain-kernel launches (bench roofline) */
  /* bf16 = 2 only: weights in the packed bf16 fragment order (F16 section above) */
  const void* W2_bf16w;    /* W2 [idim, idim] (= [Cin, Cout]) */
  const void* panel_bf16w; /* `panel` */
  const void* W3_bf16w;    /* W3_ci_tapco zero-padded to [idim, 32 | 128 | 448] */
  /* bf16 = 2 and 12 < C <= 48 only (9 C > 128 taps: the convolutions run as 64-row GEMMs on the F16 im2col operand) */
  const void* W1_bf16w;    /* W1 as [(tap,c), idim], rows zero-padded to 512, packed */
  const void* W3t_bf16w;   /* W3_tapco_ci, rows zero-padded to 512, packed */
  /* bf16 = 2 and C <= 12, optional: W3_tapco_ci in indm_f16_chain_conv's Wc form -- the series iterations then run as one launch each */
  const void* W3t_thin_bf16w;
  const void* W1_thin_bf16w;   /* optional, same form for W1 as [(tap,c), idim]: the g-net then runs as one launch (indm_f16_gnet) */
} indm_iresblock_fwd_args;

/* Backward of the training-mode block (replaces autograd's double backward through iresblock.py:264-273 and g's own
 * backward): gradients w.r.t. x, cb, the normalised weights and the biases.  dW layouts: dW1 [3,3,C,idim],
 * dW2 [idim,idim], dW3 [3,3,idim,C], dW2T [idim,idim] (gradient w.r.t. the stored W2^T; the caller adds its transpose
 * to dW2), db* [idim|C], dcb [N, idim]. */
typedef struct {
  int N, H, W, C, idim, preact, bf16;
  const float* gy;      /* [N,H,W,C] cotangent of y */
  const float* gld;     /* [N] cotangent of logdet */
  const float *a0, *d0, *a1, *d1, *a2, *d2, *t3, *t2, *c1, *w, *eps; /* as saved by the forward (a0 = x without preact) */
  const float *W1, *W2, *W2T, *W3_tapco_ci; /* W2 [idim, idim] row-major (= [Cin, Cout]) */
  float *dx, *dcb, *dW1, *dW2, *dW3, *dW2T, *db1, *db2, *db3;
  float* ws; long ws_floats;
  float* splitk_ws; long splitk_ws_floats;
  /* bf16 = 2 only (a1, d1, a2, d2, t3, t2 are F16 buffers): packed weights */
  const void *W2_bf16w, *W2T_bf16w, *panel_bf16w;
  const void *W1_bf16w, *W3t_bf16w;   /* 12 < C <= 48 only, as in the forward block */
  /* optional (bf16 = 0 / 1): a second stream for the block's weight / bias gradients, which do not feed its data path, with a
   * split-K scratch of its own; the driver orders the two streams with events and joins them before it returns.  NULL: one
   * stream, the launch order of rounds 3 - 5.  Same kernels and results either way. */
  void* side_stream;
  float* side_splitk_ws; long side_splitk_ws_floats;
  /* bf16 = 2 and C <= 12, optional: W1 as [(tap,c), idim] in indm_f16_chain_conv's Wc form -- q1, q1 .* d1 and du2 then come out of
   * one launch (indm_f16_conv_gemm) */
  const void* W1_thin_bf16w;
} indm_iresblock_bwd_args;

/* Fixed-point inverse x <- y - g(x) (iresblock.py:78-88): iterates until every element passes
 * (dx)^2 / (atol + |y| rtol) < 1 or 1000 iterations; reads ONE device scalar per iteration (the reference's torch.all).
 * Uses the forward argument block: x = y (input), y = x_out (output); eps / coeffs / logdet / saved tensors unused. */
long indm_iresblock_ws_floats(int N, int H, int W, int C, int idim, int which /* 0 fwd train, 1 fwd eval, 2 bwd, 3 inverse; +16: the bf16 = 2 form */);
int indm_iresblock_fwd_logdet(const indm_iresblock_fwd_args* a, void* stream);
int indm_iresblock_bwd(const indm_iresblock_bwd_args* a, void* stream);
int indm_iresblock_inverse(const indm_iresblock_fwd_args* a, float atol, float rtol, int* iterations, void* stream);

/* Launch timer used by the block drivers (they own the launches bench.py's roofline line prices): HIP events on the
 * launch stream around every chain-kernel launch whose class bit is set in cls_mask (bit 0: J <= 32; 1: J <= 32 + stored
 * intermediate; 2: J <= 128; 3: J <= 128 + stored intermediate).  create/destroy allocate and free the events. */
int indm_timer_create(int capacity, void** handle);
int indm_timer_destroy(void* handle);
int indm_timer_reset(void* handle, int cls_mask);
int indm_timer_read(void* handle, float* total_ms, int* launches, double* flops);

#ifdef __cplusplus
}
#endif
#endif /* INDM_HIP_H */

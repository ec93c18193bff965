// Block-level drivers of the residual flow: ONE C-ABI call per iResBlock and direction (include/indm_hip.h,
// "block-level drivers").  Nothing here is a kernel of its own except one tiny re-layout: the drivers enqueue the
// launch sequence that indm_amd/flow.py::_IResTrain composes from the op-level entry points -- same kernels, same
// order, same arguments, hence bit-identical results -- so that a training step crosses the Python / ctypes boundary
// 64 times for the flow instead of ~1 800 times (the 16- to 64-image per-rank shards of BASELINE configs #2-#5 are
// launch-bound on the host otherwise).
//
// Reference: flows/resflow/layers/iresblock.py:14-174 (forward :60-76, _logdetgrad :90-174, estimators :253-273,
// _inverse_fixed_point :78-88).
#include <cstdlib>
#include "indm_common.h"

#pragma GCC visibility push(default)
#include "../../include/indm_hip.h"
#pragma GCC visibility pop

#include <vector>

namespace {

struct Bump {
  float* base;
  long cap, off;
  float* get(long n) {
    const long a = (n + 63) & ~63L;  // 256-byte granules: every tensor stays 16-byte aligned for the DMA kernels
    if (!base || off + a > cap) return nullptr;
    float* p = base + off;
    off += a;
    return p;
  }
};
static long pad64(long n) { return (n + 63) & ~63L; }

// ---- launch timer -------------------------------------------------------------------------------------------------
struct Timer {
  std::vector<hipEvent_t> ev;
  int used = 0, mask = 0;
  double flops = 0.0;
};

static bool covered(int N, int H, int W, int C, int idim) {
  const long M = (long)N * H * W;
  if (idim != 512 || C < 1 || M < 1024 || (M % 64) || ((long)H * W) % 64) return false;
  if (M * 9 * C >= (1L << 31) || M * idim >= (1L << 40)) return false;
  const bool flat = C <= 3 && W <= 128 && (128 % W) == 0 && ((H * W) % 128) == 0 && (M % 128) == 0;
  const bool k128 = !(C & 3) && 9 * C <= 128;
  return flat || k128;
}

#define TRY(call)            \
  do {                       \
    int st_ = (call);        \
    if (st_ != 0) return st_ < 0 ? st_ : INDM_E_SHAPE; /* a sub-launch that declines a covered geometry is a bug */ \
  } while (0)

// INDM_GPASS_FUSED (A/B switch, read once): bit 0 conv + Sin, bit 1 GEMM + Sin (g-pass), bit 2 GEMM + Sin-level backward (level 1),
// bit 3 gated GEMM + Sin-level backward (level 2) in one launch each; default 15, 0 = the two-step forms
static int gpass_fused() {
  static const int v = getenv("INDM_GPASS_FUSED") ? atoi(getenv("INDM_GPASS_FUSED")) : 15;
  return v;
}

struct Ctx {
  int N, H, W, C, Cw, bf16;
  long M;
  float* sk;
  long sk_floats;
  void* st;
  Timer* timer;

  int gemm(const float* A, const float* G, const float* B, float* Cout, int M_, int N_, int K_, int ta, int tb, long lda,
           long ldb, const float* bias, const float* bias2, int rpg) const {
    auto f = bf16 ? indm_gemm_bf16 : indm_gemm_f32;
    return f(A, G, B, Cout, M_, N_, K_, ta, tb, lda, ldb, N_, 1, 0, 0, (long)M_ * N_, 1.0f, bias, bias2, rpg, nullptr, N_,
             0, 1.0f, sk, sk_floats, st);
  }
  // 3x3 / pad 1 conv of a skinny NHWC tensor x [N,H,W,C] with w2d [9C, Cw]; sign -1 conv, +1 transposed conv
  int thin(const float* x, const float* w2d, float* y, int sign, const float* bias) const {
    auto f = bf16 ? indm_conv3x3_thin_bf16 : indm_conv3x3_thin_f32;
    return f(x, w2d, y, N, H, W, C, Cw, sign, 1.0f, bias, nullptr, nullptr, 1.0f, st);
  }
  // conv + Sin in one launch; 1 = geometry not covered (or no gate output wanted)
  int thin_act(const float* x, const float* w2d, float* a, float* d, const float* bias) const {
    if (!d || !(gpass_fused() & 1) || (C > 3 && M < 8192)) return 1;   // (12 channels at 16 images: 21.7 us fused against 13.5 + 6.2)
    auto f = bf16 ? indm_conv3x3_thin_act_bf16 : indm_conv3x3_thin_act_f32;
    return f(x, w2d, a, d, N, H, W, C, Cw, -1, bias, nullptr, st);
  }
  int chain(const float* t3, const float* d2, const float* W2T, const void* W2Tp, const float* d1, const float* panel,
            float* P, float* t2) const {
    const int J = 9 * C;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timer) {
      const int cls = (J <= 32 ? 0 : 2) + (t2 ? 1 : 0);
      if (((timer->mask >> cls) & 1) && timer->used + 2 <= (int)timer->ev.size()) {
        e0 = timer->ev[timer->used];
        e1 = timer->ev[timer->used + 1];
      }
    }
    if (e0) (void)hipEventRecord(e0, (hipStream_t)st);
    int r;
    if (bf16 && W2Tp) r = indm_ires_chain_bf16w(t3, d2, W2Tp, d1, panel, P, t2, (int)M, Cw, Cw, J, st);
    else r = (bf16 ? indm_ires_chain_bf16 : indm_ires_chain_f32)(t3, d2, W2T, d1, panel, P, t2, (int)M, Cw, Cw, J, st);
    if (e0) {
      (void)hipEventRecord(e1, (hipStream_t)st);
      timer->used += 2;
      timer->flops += 2.0 * (double)M * Cw * (Cw + J);
    }
    return r;
  }
};

// g(x, h) of the block (iresblock.py:60-76 / nnet): fills a0,d0 (preact), a1,d1,a2,d2 and returns g in `g`
// (p = [M, Cw] scratch for the two pre-activations, Pn = [M, 9C] scratch).  want_d = 0: gates are not produced.
static int g_pass(const Ctx& c, int preact, const float* x, const float* cb, const float* W1, const float* W2T,
                  const float* W3ct, const float* b1, const float* b2, const float* b3, float* a0, float* d0, float* a1,
                  float* d1, float* a2, float* d2, float* p, float* Pn, float* g, float* y_res = nullptr) {
  const long nC = c.M * c.C, nW = c.M * c.Cw;
  const float* in = x;
  if (preact) {
    TRY(indm_sincos_act_fwd_f32(x, a0, d0, nC, c.st));
    in = a0;
  }
  // (a1, d1) = Sin(conv(a0, W1) + b1): one kernel where the row-fragment thin convolution covers the geometry (p1 never
  // written), else the convolution and the activation pass; same values either way
  const int fused = c.thin_act(in, W1, a1, d1, b1);
  if (fused < 0) return fused;
  if (fused == 1) {
    TRY(c.thin(in, W1, p, -1, b1));                                                 // p1 = conv(a0, W1) + b1
    TRY(indm_sincos_act_fwd_f32(p, a1, d1, nW, c.st));
  }
  // (a2, d2) = Sin(a1 W2 + b2 + cb[n]): the GEMM's epilogue where the direct-to-LDS kernel takes the problem (fp32, gates wanted)
  int fused2 = 1;
  if (d2 && !c.bf16 && (gpass_fused() & 2))
    fused2 = indm_gemm_sin_f32(a1, W2T, a2, d2, (int)c.M, c.Cw, c.Cw, c.Cw, c.Cw, c.Cw, b2, cb, c.H * c.W, c.st);
  if (fused2 < 0) return fused2;
  if (fused2 == 1) {
    TRY(c.gemm(a1, nullptr, W2T, p, (int)c.M, c.Cw, c.Cw, 0, 1, c.Cw, c.Cw, b2, cb, c.H * c.W));   // p2 = a1 W2 + b2 + cb[n]
    TRY(indm_sincos_act_fwd_f32(p, a2, d2, nW, c.st));
  }
  TRY(c.gemm(a2, nullptr, W3ct, Pn, (int)c.M, 9 * c.C, c.Cw, 0, 0, c.Cw, 9 * c.C, nullptr, nullptr, 1));   // skinny-output conv: GEMM + 9-tap gather
  if (y_res)   // the block's output y = x + g in the gather's own pass (x + 1.0f * g: the same sum axpby would form)
    TRY(indm_tap_gather_acc_f32(Pn, b3, nullptr, g, x, y_res, 1.0f, c.N, c.H, c.W, c.C, 1, 1.0f, c.st));
  else
    TRY(indm_tap_gather_f32(Pn, b3, g, c.N, c.H, c.W, c.C, 1, 1.0f, c.st));
  return 0;
}

// out[b][a][c] = x[a][b][c] (+ x2[a][b][c]) (+ y[b][a][c]) : re-layout (and sum) of the small weight-gradient GEMM results
// (ldx = row stride of x / x2 over a: B C when dense, 512 when they are the padded [512, 512] results of the F16 row contraction)
__global__ __launch_bounds__(256) void permute021_add_kernel(const float* __restrict__ x, const float* __restrict__ x2,
                                                             const float* __restrict__ y, float* __restrict__ out, int A,
                                                             int B, int C, int ldx) {
  const int total = A * B * C;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int c = i % C, ab = i / C;
    const int a = ab % A, b = ab / A;        // i indexes out[b][a][c]
    const int src = a * ldx + b * C + c;
    float v = x[src];
    if (x2) v = v + x2[src];
    if (y) v = v + y[i];
    out[i] = v;
  }
}
static int permute021_add(const float* x, const float* x2, const float* y, float* out, int A, int B, int C, void* st, int ldx = 0) {
  const int total = A * B * C;
  hipLaunchKernelGGL(permute021_add_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, (hipStream_t)st, x, x2, y, out, A, B, C,
                     ldx > 0 ? ldx : B * C);
  INDM_CHECK_LAUNCH();
  return 0;
}

}  // namespace

INDM_API long indm_iresblock_ws_floats(int N, int H, int W, int C, int idim, int which) {
  const long M = (long)N * H * W, nC = pad64(M * C), nW = pad64(M * idim), nP = pad64(M * 9 * C);
  const long red = pad64(indm_rowreduce_ws_floats(N, (long)H * W * C)) + pad64(N);
  const long nH = pad64(M * idim / 2);   // one F16 [M, idim] tensor, in floats
  switch (which) {
    // (C > 12, the wide form: + the F16 im2col operand; the weight-gradient results are padded [512][512])
    case 16:  // forward, training, bf16 storage: u2 (F16) + Pn, g, v, wtmp + reductions
      return nH + nP + 4 * nC + red + (C > 12 ? nH : 0) + 64;
    case 17:  // forward, eval, bf16 storage: + a0, d0 (fp32) and a1, d1, a2, d2 (F16) live here
      return nH + nP + 3 * nC + red + 2 * nC + 4 * nH + (C > 12 ? nH : 0) + 64;
    case 18: {  // backward, bf16 storage: five F16 [M, idim] buffers + the fp32 skinny side
      const long small = 6 * pad64((long)idim * (C > 12 ? 512 : 9 * C)) + pad64((long)N * idim) + pad64(idim);
      const long redb = pad64(indm_colreduce_ws_floats(M, C)) + pad64(indm_colreduce_ws_floats(N, idim));
      return 6 * nH + 2 * nP + 5 * nC + small + redb + (C > 12 ? 2 * nH : 0) + 64;   // (+ dp1's own buffer / the side stream's spread operand)
    }
    case 0:  // forward, training: p, Pn, g, v, wtmp, vj + reductions
      return nW + nP + 4 * nC + red + 64;
    case 1:  // forward, eval: + a0,d0,a1,d1,a2,d2,t3 live here
      return nW + nP + 3 * nC + red + 2 * nC + 5 * nW + 64;
    case 2: {  // backward
      const long small = 6 * pad64((long)idim * 9 * C) + pad64((long)N * idim) + pad64(idim);
      const long redb = pad64(indm_colreduce_ws_floats(M, idim)) + pad64(indm_ires_bwd_act_ws_floats(M, idim)) +
                        pad64(indm_colreduce_ws_floats(M, C)) + pad64(indm_colreduce_ws_floats(N, idim));
      return 6 * nW + 2 * nP + 5 * nC + small + redb + 64;   // (two more [M, idim] buffers: the two-stream form keeps du2 / dp2 alive)
    }
    default:  // inverse: a0, a1, a2, p, Pn, g, x ping-pong, check scalar
      return 3 * nW + nP + 4 * nC + 128;
  }
}

// ---- bf16 storage (args.bf16 == 2): the same block on the F16 kernels of flow_f16.hip ---------------------------------
// The [M, 512] tensors live in HBM as bf16 in MFMA-fragment order; pre-activations are never stored (the Sin activation is
// the epilogue of the contraction that produces them), the d2 gate of a chain link is applied by the thin transposed
// convolution that produces t3 (so the chain kernel's main loop multiplies fragments as they come out of LDS), the d1 gate
// by the chain kernel on its accumulators.  Per series iteration: 134 MB written + 134 MB read by the thin convolution,
// 268 MB read by the chain kernel at batch 128 (fp32 tensors: 268 written, 805 read).
namespace {
enum { F16_PLAIN = 0, F16_ACT = 1, F16_GATE = 2, F16R_GEMM = 0, F16R_GEMM_ACT = 1, F16R_CHAIN = 2, F16R_THINN = 3, F16R_GEMM_GATE = 4, F16R_GEMM_ACT_THINN = 5 };

// 12 < C (the 48-channel scale of the 64 x 64 flows, 9 C = 432 taps): no gathering thin kernel -- the skinny side's im2col image
// becomes an F16 [M, 512] operand (indm_f16_tap_spread into `colF`, zero columns behind 9 C) and the convolution the 64-row
// GEMM on it with the K-padded packed weights (`wp`), epilogues unchanged; weight gradients: the row contraction of two F16
// tensors (indm_f16_wgrad_gemm) instead of the thin form.
struct Ctx16 {
  int N, H, W, C, Cw;
  long M;
  void* st;
  Timer* timer;
  void* colF;   // wide only
  bool wide() const { return C > 12; }
  int thin(const float* x, const float* w2d, const void* wp, const float* bias, const void* gate, void* o1, void* o2, int sign,
           int epi) const {
    if (!wide()) return indm_f16_thin_conv(x, w2d, bias, gate, o1, o2, N, H, W, C, Cw, sign, epi, st);
    int r = indm_f16_tap_spread(x, colF, N, H, W, C, sign, st);
    if (r != 0) return r;
    const int mode = epi == F16_PLAIN ? F16R_GEMM : epi == F16_ACT ? F16R_GEMM_ACT : F16R_GEMM_GATE;
    return indm_f16_rows(mode, colF, wp, gate, nullptr, nullptr, o1, o2, bias, nullptr, (int)M, 0, 1, st);
  }
  // weight gradient of a 3x3 convolution between the skinny side `xs` (fp32 NHWC, spread with `sign`) and the F16 tensor `wide16`
  // (.* gate): tr = 1: out[n][(tap, c)] (row stride ld_out()), tr = 0: out[(tap, c)][n]
  int ld_out() const { return wide() ? 512 : 9 * C; }
  int wgrad(const float* xs, int sign, float* col, const void* wide16, const void* gate, float* out, int tr, float* sk,
            long skn) const {
    const int J = 9 * C;
    if (!wide()) {
      int r = indm_tap_spread_f32(xs, col, N, H, W, C, sign, 1.0f, st);
      if (r != 0) return r;
      return indm_f16_thin_wgrad(col, wide16, gate, out, M, J, 1.0f, tr, tr ? J : Cw, sk, skn, st);
    }
    int r = indm_f16_tap_spread(xs, colF, N, H, W, C, sign, st);
    if (r != 0) return r;
    if (tr) return indm_f16_wgrad_gemm(wide16, gate, colF, out, M, 1.0f, 0, sk, skn, st);   // [Cw][512]: columns (tap, c), zeros behind
    return gate ? INDM_E_ARG : indm_f16_wgrad_gemm(colF, nullptr, wide16, out, M, 1.0f, 0, sk, skn, st);   // [512][Cw]: the first 9 C rows
  }
  // one series iteration: u2 = (C3^T v) .* d2, P = ((u2 W2T) .* d1) panel -- ONE launch (indm_f16_chain_conv) when the thin-order
  // pack of W3^T is given (C <= 12), else the thin convolution + the chain link
  int series(const float* vin, const float* w3t, const void* w3t_wide, const void* w3t_thin, const void* d2, void* u2,
             const void* W2Tp, const void* d1, const void* panelp, float* P, void* t3 = nullptr, void* t2 = nullptr) const {
    static const int fused = getenv("INDM_F16_CHAIN_CONV") ? atoi(getenv("INDM_F16_CHAIN_CONV")) : 1;
    if (fused && w3t_thin && !wide()) {
      const int J = 9 * C;
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (timer) {
        const int cls = (J <= 32 ? 0 : 2) + (t2 ? 1 : 0);
        if (((timer->mask >> cls) & 1) && timer->used + 2 <= (int)timer->ev.size()) {
          e0 = timer->ev[timer->used];
          e1 = timer->ev[timer->used + 1];
        }
      }
      if (e0) (void)hipEventRecord(e0, (hipStream_t)st);
      const int r = indm_f16_chain_conv(vin, w3t_thin, d2, W2Tp, d1, panelp, P, t3, t2, N, H, W, C, 1, J, st);
      if (r != 1) {
        if (e0) {
          (void)hipEventRecord(e1, (hipStream_t)st);
          timer->used += 2;
          timer->flops += 2.0 * (double)M * Cw * (Cw + 2 * J);
        }
        return r;
      }
    }
    int r = thin(vin, w3t, w3t_wide, nullptr, d2, u2, t3, 1, F16_GATE);
    if (r != 0) return r;
    return chain(u2, W2Tp, d1, panelp, P, t2);
  }
  int chain(const void* u2, const void* W2Tp, const void* d1, const void* panelp, float* P, void* t2) const {
    const int J = 9 * C;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timer) {
      const int cls = (J <= 32 ? 0 : 2) + (t2 ? 1 : 0);
      if (((timer->mask >> cls) & 1) && timer->used + 2 <= (int)timer->ev.size()) {
        e0 = timer->ev[timer->used];
        e1 = timer->ev[timer->used + 1];
      }
    }
    if (e0) (void)hipEventRecord(e0, (hipStream_t)st);
    const int r = indm_f16_rows(F16R_CHAIN, u2, W2Tp, d1, panelp, P, t2, nullptr, nullptr, nullptr, (int)M, J, 1, st);
    if (e0) {
      (void)hipEventRecord(e1, (hipStream_t)st);
      timer->used += 2;
      timer->flops += 2.0 * (double)M * Cw * (Cw + J);
    }
    return r;
  }
};

static bool covered16(int N, int H, int W, int C, int idim) {
  const long M = (long)N * H * W;
  return idim == 512 && (C <= 3 || (C <= 48 && !(C & 3))) && M >= 1024 && !(M % 128) && !(((long)H * W) % 64) &&
         M * 9 * C < (1L << 31);
}

// g(x, h) on F16 tensors: fills a0, d0 (preact, fp32), a1, d1, a2, d2 (F16) and g (fp32) [+ y = x + g]
static int g_pass16(const Ctx16& c, int preact, const float* x, const float* cb, const float* W1, const void* W1p, const void* W1thin,
                    const void* W2p, const void* W3p, const float* b1, const float* b2, const float* b3, float* a0, float* d0, void* a1, void* d1,
                    void* a2, void* d2, float* Pn, float* g, float* y_res) {
  const long nC = c.M * c.C;
  const float* in = x;
  if (preact) {
    TRY(indm_sincos_act_fwd_f32(x, a0, d0, nC, c.st));
    in = a0;
  }
  // round 6: the whole g-net -- conv + Sin, GEMM + Sin, thin-N -- as ONE launch where the geometry allows (indm_f16_gnet)
  static const int fuse_gnet = getenv("INDM_F16_GNET") ? atoi(getenv("INDM_F16_GNET")) : 1;
  int fg = 1;
  if (fuse_gnet && W1thin && !c.wide())
    fg = indm_f16_gnet(in, W1thin, b1, a1, d1, W2p, b2, cb, c.H * c.W, a2, d2, W3p, Pn, c.N, c.H, c.W, c.C, c.st);
  if (fg < 0) return fg;
  if (fg == 0) {
    if (y_res) TRY(indm_tap_gather_acc_f32(Pn, b3, nullptr, g, x, y_res, 1.0f, c.N, c.H, c.W, c.C, 1, 1.0f, c.st));
    else TRY(indm_tap_gather_f32(Pn, b3, g, c.N, c.H, c.W, c.C, 1, 1.0f, c.st));
    return 0;
  }
  TRY(c.thin(in, W1, W1p, b1, nullptr, a1, d1, -1, F16_ACT));                                // (a1, d1) = Sin(conv(a0, W1) + b1)
  // (a2, d2) = Sin(a1 W2 + b2 + cb[n]) and Pn = a2 W3 (the skinny-output convolution before its tap gather): one launch where the
  // panel has <= 128 columns (round 6: the thin-N product runs on the activation's accumulators), else two
  static const int fuse_thinn = getenv("INDM_F16_ACT_THINN") ? atoi(getenv("INDM_F16_ACT_THINN")) : 1;
  int fz = 1;
  if (fuse_thinn && 9 * c.C <= 128 && c.M / 64 >= 512)
    fz = indm_f16_rows(F16R_GEMM_ACT_THINN, a1, W2p, nullptr, W3p, Pn, a2, d2, b2, cb, (int)c.M, 9 * c.C, c.H * c.W, c.st);
  if (fz < 0) return fz;
  if (fz == 1) {
    TRY(indm_f16_rows(F16R_GEMM_ACT, a1, W2p, nullptr, nullptr, nullptr, a2, d2, b2, cb, (int)c.M, 0, c.H * c.W, c.st));   // Sin(a1 W2 + b2 + cb[n])
    TRY(indm_f16_rows(F16R_THINN, a2, nullptr, nullptr, W3p, Pn, nullptr, nullptr, nullptr, nullptr, (int)c.M, 9 * c.C, 1, c.st));
  }
  if (y_res) TRY(indm_tap_gather_acc_f32(Pn, b3, nullptr, g, x, y_res, 1.0f, c.N, c.H, c.W, c.C, 1, 1.0f, c.st));
  else TRY(indm_tap_gather_f32(Pn, b3, g, c.N, c.H, c.W, c.C, 1, 1.0f, c.st));
  return 0;
}

static int fwd_logdet_f16(const indm_iresblock_fwd_args* a, void* stream) {
  if (!covered16(a->N, a->H, a->W, a->C, a->idim) || !a->panel) return 1;
  if (!a->x || !a->cb || !a->W1 || !a->W3_tapco_ci || !a->b1 || !a->b2 || !a->b3 || !a->eps || !a->y || !a->logdet ||
      !a->W2T_bf16w || !a->W2_bf16w || !a->panel_bf16w || !a->W3_bf16w || (a->ncoeff > 0 && !a->coeffs) || a->ncoeff < 0 ||
      (a->C > 12 && (!a->W1_bf16w || !a->W3t_bf16w)))
    return INDM_E_ARG;
  Ctx16 c = {a->N, a->H, a->W, a->C, a->idim, (long)a->N * a->H * a->W, stream, (Timer*)a->timer, nullptr};
  const long nC = c.M * c.C, nP = c.M * 9 * c.C, nH = c.M * c.Cw / 2;
  Bump ws = {a->ws, a->ws_floats, 0};
  float *a0 = a->a0, *d0 = a->d0;
  void *a1 = a->a1, *d1 = a->d1, *a2 = a->a2, *d2 = a->d2;
  if (a->training) {
    if (!a1 || !d1 || !a2 || !d2 || !a->t3 || !a->t2 || !a->w || (a->preact && (!a0 || !d0 || !a->c1))) return INDM_E_ARG;
  } else {
    if (a->preact) { a0 = ws.get(nC); d0 = ws.get(nC); }
    a1 = ws.get(nH); d1 = ws.get(nH); a2 = ws.get(nH); d2 = ws.get(nH);
  }
  void* u2 = ws.get(nH);
  float *Pn = ws.get(nP), *g = ws.get(nC), *v = ws.get(nC);
  float* rws = ws.get(indm_rowreduce_ws_floats(c.N, (long)c.H * c.W * c.C));
  if (c.wide()) c.colF = ws.get(nH);
  if (!u2 || !Pn || !g || !v || !rws || (c.wide() && !c.colF) || (!a->training && (!a1 || !d2 || (a->preact && !d0))))
    return INDM_E_ARG;
  const int K = a->ncoeff;
  TRY(g_pass16(c, a->preact, a->x, a->cb, a->W1, a->W1_bf16w, a->W1_thin_bf16w, a->W2_bf16w, a->W3_bf16w, a->b1, a->b2, a->b3, a0, d0, a1, d1, a2, d2,
               Pn, g, a->y));
  const float* gate0 = a->preact ? d0 : nullptr;
  if (a->training) {
    float* wtmp = ws.get(nC);
    if (!wtmp) return INDM_E_ARG;
    const float *vin = a->eps, *win = a->eps;
    for (int k = 0; k < K; ++k) {
      float* wout = ((K - 1 - k) & 1) ? wtmp : a->w;
      TRY(c.series(vin, a->W3_tapco_ci, a->W3t_bf16w, a->W3t_thin_bf16w, d2, u2, a->W2T_bf16w, d1, a->panel_bf16w, Pn));   // u2 = (C3^T v) .* d2; the chain link
      TRY(indm_tap_gather_acc_f32(Pn, nullptr, gate0, v, win, wout, a->coeffs[k], c.N, c.H, c.W, c.C, -1, 1.0f, stream));
      vin = v;
      win = wout;
    }
    if (K == 0 && hipMemcpyAsync(a->w, a->eps, nC * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
      return INDM_E_LAUNCH;
    TRY(c.series(a->w, a->W3_tapco_ci, a->W3t_bf16w, a->W3t_thin_bf16w, d2, u2, a->W2T_bf16w, d1, a->panel_bf16w, Pn, a->t3, a->t2));   // ... and t3, t2 saved
    if (a->preact) {
      TRY(indm_tap_gather_f32(Pn, nullptr, a->c1, c.N, c.H, c.W, c.C, -1, 1.0f, stream));
      TRY(indm_mul_f32(a->c1, d0, g, 1.0f, nC, stream));
    } else {
      TRY(indm_tap_gather_f32(Pn, nullptr, g, c.N, c.H, c.W, c.C, -1, 1.0f, stream));
    }
    TRY(indm_rowreduce_f32(g, a->eps, a->logdet, c.N, (long)c.H * c.W * c.C, 0, 1.0f, rws, stream));
    return INDM_OK;
  }
  float* tr = ws.get(c.N);
  if (!tr) return INDM_E_ARG;
  const float* vin = a->eps;
  for (int k = 0; k < K; ++k) {
    TRY(c.series(vin, a->W3_tapco_ci, a->W3t_bf16w, a->W3t_thin_bf16w, d2, u2, a->W2T_bf16w, d1, a->panel_bf16w, Pn));
    TRY(indm_tap_gather_acc_f32(Pn, nullptr, gate0, v, nullptr, nullptr, 0.0f, c.N, c.H, c.W, c.C, -1, 1.0f, stream));
    vin = v;
    TRY(indm_rowreduce_f32(v, a->eps, k == 0 ? a->logdet : tr, c.N, (long)c.H * c.W * c.C, 0, a->coeffs[k], rws, stream));
    if (k > 0) TRY(indm_axpby_f32(a->logdet, tr, a->logdet, 1.0f, 1.0f, 0.0f, c.N, stream));
  }
  if (K == 0 && hipMemsetAsync(a->logdet, 0, c.N * sizeof(float), (hipStream_t)stream) != hipSuccess) return INDM_E_LAUNCH;
  return INDM_OK;
}
}  // namespace

INDM_API int indm_iresblock_fwd_logdet(const indm_iresblock_fwd_args* a, void* stream) {
  if (!a) return INDM_E_ARG;
  if (a->N <= 0) return INDM_OK;
  if (a->bf16 == 2) return fwd_logdet_f16(a, stream);
  if (!covered(a->N, a->H, a->W, a->C, a->idim) || !a->panel) return 1;
  if (!a->x || !a->cb || !a->W1 || !a->W2T || !a->W3_ci_tapco || !a->W3_tapco_ci || !a->b1 || !a->b2 || !a->b3 ||
      !a->eps || !a->y || !a->logdet || (a->ncoeff > 0 && !a->coeffs) || a->ncoeff < 0)
    return INDM_E_ARG;
  Ctx c = {a->N, a->H, a->W, a->C, a->idim, a->bf16, (long)a->N * a->H * a->W, a->splitk_ws, a->splitk_ws_floats, stream,
           (Timer*)a->timer};
  const long nC = c.M * c.C, nW = c.M * c.Cw, nP = c.M * 9 * c.C;
  Bump ws = {a->ws, a->ws_floats, 0};
  float *a0 = a->a0, *d0 = a->d0, *a1 = a->a1, *d1 = a->d1, *a2 = a->a2, *d2 = a->d2, *t3 = a->t3;
  if (a->training) {
    if (!a1 || !d1 || !a2 || !d2 || !t3 || !a->t2 || !a->w || (a->preact && (!a0 || !d0 || !a->c1))) return INDM_E_ARG;
  } else {
    if (a->preact) { a0 = ws.get(nC); d0 = ws.get(nC); }
    a1 = ws.get(nW); d1 = ws.get(nW); a2 = ws.get(nW); d2 = ws.get(nW); t3 = ws.get(nW);
  }
  float *p = ws.get(nW), *Pn = ws.get(nP), *g = ws.get(nC), *v = ws.get(nC);
  float* rws = ws.get(indm_rowreduce_ws_floats(c.N, (long)c.H * c.W * c.C));
  if (!p || !Pn || !g || !v || !rws || (!a->training && (!a1 || !t3 || (a->preact && !d0)))) return INDM_E_ARG;
  const int K = a->ncoeff;

  TRY(g_pass(c, a->preact, a->x, a->cb, a->W1, a->W2T, a->W3_ci_tapco, a->b1, a->b2, a->b3, a0, d0, a1, d1, a2, d2, p, Pn, g,
             a->y));                                                                // ... and y = x + g
  const float* gate0 = a->preact ? d0 : nullptr;

  if (a->training) {
    // neumann_logdet_estimator (iresblock.py:264-273): v_k = J^T v_{k-1}, w = eps + sum (-1)^k c_k v_k, no graph
    float* wtmp = ws.get(nC);
    if (!wtmp) return INDM_E_ARG;
    const float *vin = a->eps, *win = a->eps;
    for (int k = 0; k < K; ++k) {
      float* wout = ((K - 1 - k) & 1) ? wtmp : a->w;          // the last iteration lands in the saved tensor
      TRY(c.thin(vin, a->W3_tapco_ci, t3, 1, nullptr));       // t3 = C3^T v
      TRY(c.chain(t3, d2, a->W2T, a->W2T_bf16w, d1, a->panel, Pn, nullptr));
      TRY(indm_tap_gather_acc_f32(Pn, nullptr, gate0, v, win, wout, a->coeffs[k], c.N, c.H, c.W, c.C, -1, 1.0f, stream));
      vin = v;
      win = wout;
    }
    if (K == 0 && hipMemcpyAsync(a->w, a->eps, nC * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
      return INDM_E_LAUNCH;
    // the gradient-carrying last link: vj = d0 .* C1^T(d1 .* ((d2 .* C3^T w) W2^T)), logdet = <vj, eps>
    TRY(c.thin(a->w, a->W3_tapco_ci, t3, 1, nullptr));
    TRY(c.chain(t3, d2, a->W2T, a->W2T_bf16w, d1, a->panel, Pn, a->t2));
    const float* vj;
    if (a->preact) {
      TRY(indm_tap_gather_f32(Pn, nullptr, a->c1, c.N, c.H, c.W, c.C, -1, 1.0f, stream));
      TRY(indm_mul_f32(a->c1, d0, g, 1.0f, nC, stream));     // g is dead: reuse it for vj
      vj = g;
    } else {
      TRY(indm_tap_gather_f32(Pn, nullptr, g, c.N, c.H, c.W, c.C, -1, 1.0f, stream));
      vj = g;
    }
    TRY(indm_rowreduce_f32(vj, a->eps, a->logdet, c.N, (long)c.H * c.W * c.C, 0, 1.0f, rws, stream));
    return INDM_OK;
  }
  // basic_logdet_estimator (iresblock.py:253-261): logdet = sum_k (-1)^(k+1)/k c_k <J^k-fold vjp, eps>
  float* tr = ws.get(c.N);
  if (!tr) return INDM_E_ARG;
  const float* vin = a->eps;
  for (int k = 0; k < K; ++k) {
    TRY(c.thin(vin, a->W3_tapco_ci, t3, 1, nullptr));
    TRY(c.chain(t3, d2, a->W2T, a->W2T_bf16w, d1, a->panel, Pn, nullptr));
    TRY(indm_tap_gather_acc_f32(Pn, nullptr, gate0, v, nullptr, nullptr, 0.0f, c.N, c.H, c.W, c.C, -1, 1.0f, stream));
    vin = v;
    TRY(indm_rowreduce_f32(v, a->eps, k == 0 ? a->logdet : tr, c.N, (long)c.H * c.W * c.C, 0, a->coeffs[k], rws, stream));
    if (k > 0) TRY(indm_axpby_f32(a->logdet, tr, a->logdet, 1.0f, 1.0f, 0.0f, c.N, stream));
  }
  if (K == 0 && hipMemsetAsync(a->logdet, 0, c.N * sizeof(float), (hipStream_t)stream) != hipSuccess) return INDM_E_LAUNCH;
  return INDM_OK;
}

namespace {
// Round 6, the "two-stream" form of the fp32 backward: the block's weight / bias gradients do not feed its data path, so they
// are enqueued on a SIDE stream (args.side_stream, with its own split-K scratch) behind events recorded where their operands are
// complete -- the HBM-bound thin weight gradients then run next to the MFMA-bound fused GEMMs of the data path, and the
// MFMA-bound [512, 512] weight gradients next to the HBM-bound thin convolutions (fp32 MFMAs and memory streaming use
// different parts of the CU; inside ONE stream every kernel waits for the previous one).  Same kernels, same arguments, same
// results; the main stream waits for the side stream before the call returns (the scratch is the caller's).
struct SideEvents {
  hipEvent_t ev[7];
  bool ok = false;
  SideEvents() {
    ok = true;
    for (auto& e : ev) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
  }
};
static SideEvents* side_events() {
  static SideEvents* s = new SideEvents();   // (never destroyed: the HIP runtime may be gone at static-destructor time)
  return s->ok ? s : nullptr;
}
}  // namespace

namespace {
// backward of the training-mode block on F16 tensors (args.bf16 == 2): the launch sequence of indm_iresblock_bwd with the
// F16 kernels.  Differences in HOW: the thin convolution that produces q1 also writes q1 .* d1 (= dt2, the cotangent the
// fp32 path takes from the Sin-level kernel), so the contraction behind it needs no gate; pre-gated operands replace
// gates on GEMM loaders, the weight-gradient kernels gate while they transpose.
static int bwd_f16(const indm_iresblock_bwd_args* a, void* stream) {
  if (!covered16(a->N, a->H, a->W, a->C, a->idim) || !a->gld || !indm_f16_bwd_act_ok((long)a->N * a->H * a->W, (long)a->H * a->W))
    return 1;
  if (!a->gy || !a->a0 || !a->a1 || !a->d1 || !a->a2 || !a->d2 || !a->t3 || !a->t2 || !a->w || !a->eps || !a->W1 ||
      !a->W3_tapco_ci || !a->W2_bf16w || !a->W2T_bf16w || !a->panel_bf16w || (a->preact && (!a->d0 || !a->c1)) || !a->dx ||
      !a->dcb || !a->dW1 || !a->dW2 || !a->dW3 || !a->dW2T || !a->db1 || !a->db2 || !a->db3 || !a->splitk_ws)
    return INDM_E_ARG;
  Ctx16 c = {a->N, a->H, a->W, a->C, a->idim, (long)a->N * a->H * a->W, stream, nullptr, nullptr};
  const int C = c.C, Cw = c.Cw, Mi = (int)c.M, rps = c.H * c.W, J = 9 * C;
  if (c.wide() && (!a->W1_bf16w || !a->W3t_bf16w)) return INDM_E_ARG;
  const long nC = c.M * C, nP = c.M * 9 * C, nH = c.M * Cw / 2;
  const long nQ = (long)Cw * (c.wide() ? 512 : J);   // one weight-gradient result: [Cw][9 C], or the padded [512][512] of the row contraction
  const int ldq = c.ld_out();
  Bump ws = {a->ws, a->ws_floats, 0};
  void *bufA = ws.get(nH), *bufB = ws.get(nH), *bufC = ws.get(nH), *bufD = ws.get(nH), *bufE = ws.get(nH);
  float *col = ws.get(nP), *Pn = ws.get(nP);
  float *dvj = ws.get(nC), *q0 = ws.get(nC), *dd0 = ws.get(nC), *da0 = ws.get(nC), *tmpc = ws.get(nC);
  float *qa = ws.get(nQ), *qb = ws.get(nQ), *qc = ws.get(nQ), *qd = ws.get(nQ);
  float* seg1 = ws.get((long)c.N * Cw);
  float *cr_c = ws.get(indm_colreduce_ws_floats(c.M, C)), *cr_n = ws.get(indm_colreduce_ws_floats(c.N, Cw));
  if (c.wide()) c.colF = ws.get(nH);
  if (!bufA || !bufB || !bufC || !bufD || !bufE || !col || !Pn || !dvj || !q0 || !dd0 || !da0 || !tmpc || !qa || !qb || !qc ||
      !qd || !seg1 || !cr_c || !cr_n || (c.wide() && !c.colF))
    return INDM_E_ARG;
  // ---- the side stream of the weight / bias gradients (SideEvents above): at the 16 x 16 scale and at the per-rank shards these
  // launches are short and latency-bound -- next to the data path's they fill the chip instead of queueing behind it
  SideEvents* se = (a->side_stream && a->side_splitk_ws && a->side_stream != stream) ? side_events() : nullptr;
  Ctx16 w = c;
  hipStream_t ms = (hipStream_t)stream, ss = ms;
  float* sk = a->splitk_ws;
  long skn = a->splitk_ws_floats;
  void* bufF = bufD;   // dp1: du2's buffer when everything runs in one stream
  if (se) {
    w.st = a->side_stream;
    ss = (hipStream_t)a->side_stream;
    sk = a->side_splitk_ws;
    skn = a->side_splitk_ws_floats;
    bufF = ws.get(nH);                       // du2 is still being read by the side stream's weight gradient when dp1 is written
    if (c.wide()) w.colF = ws.get(nH);       // the spread operand of the side stream's weight gradients
    if (!bufF || (c.wide() && !w.colF)) return INDM_E_ARG;
  }
  auto fork = [&](int i) -> int {
    if (!se) return 0;
    if (hipEventRecord(se->ev[i], ms) != hipSuccess || hipStreamWaitEvent(ss, se->ev[i], 0) != hipSuccess) return INDM_E_LAUNCH;
    return 0;
  };
  auto join = [&]() -> int {
    if (!se) return 0;
    if (hipEventRecord(se->ev[6], ss) != hipSuccess || hipStreamWaitEvent(ms, se->ev[6], 0) != hipSuccess) return INDM_E_LAUNCH;
    return 0;
  };
#define TRYJ(call)                                              \
  do {                                                          \
    int st_ = (call);                                           \
    if (st_ != 0) {                                             \
      (void)join();                                             \
      return st_ < 0 ? st_ : INDM_E_SHAPE;                      \
    }                                                           \
  } while (0)
  void* sst = w.st;
  TRYJ(fork(0));   // the inputs of this call are complete on the main stream
  // ---- g path, level 3:  g = conv(a2, W3) + b3 ----
  void* da2 = bufA;
  TRYJ(c.thin(a->gy, a->W3_tapco_ci, a->W3t_bf16w, nullptr, nullptr, da2, nullptr, 1, F16_PLAIN));         // da2 = C3^T gy
  TRYJ(w.wgrad(a->gy, 1, col, a->a2, nullptr, qa, 1, sk, skn));                                            // a2^T spread(gy): [Cw, 9C]
  TRYJ(indm_colreduce_f32(a->gy, nullptr, a->db3, c.M, C, 0, 1.0f, 0.0f, cr_c, sst));
  // ---- reverse sweep of the last VJP chain ----
  TRYJ(indm_scale_rows_f32(a->eps, a->gld, dvj, c.N, (long)rps * C, stream));
  const float* q0p = dvj;
  if (a->preact) {
    TRYJ(indm_mul2_f32(dvj, a->d0, a->c1, q0, dd0, nC, stream));
    q0p = q0;
  }
  TRYJ(fork(1));   // q0
  void *q1 = bufB, *q1g = bufC;
  // q1 = conv(q0, W1); q1g = q1 .* d1 (= dt2); du2 = q1g W2 -- ONE launch (indm_f16_conv_gemm) when W1's thin-order pack is given
  void* du2 = bufD;
  static const int fuse_cg = getenv("INDM_F16_CONV_GEMM") ? atoi(getenv("INDM_F16_CONV_GEMM")) : 1;
  int fcg = 1;
  if (fuse_cg && a->W1_thin_bf16w && !c.wide())
    fcg = indm_f16_conv_gemm(q0p, a->W1_thin_bf16w, a->d1, a->W2_bf16w, du2, q1, q1g, c.N, c.H, c.W, C, -1, stream);
  if (fcg < 0) { (void)join(); return fcg; }
  if (fcg == 1) TRYJ(c.thin(q0p, a->W1, a->W1_bf16w, nullptr, a->d1, q1g, q1, -1, F16_GATE));
  TRYJ(w.wgrad(q0p, -1, col, a->t2, a->d1, qb, 1, sk, skn));                                               // (t2 .* d1)^T spread(q0)
  if (fcg == 1)
    TRYJ(indm_f16_rows(F16R_GEMM, q1g, a->W2_bf16w, nullptr, nullptr, nullptr, du2, nullptr, nullptr, nullptr, Mi, 0, 1, stream));   // du2 = (q1 .* d1) W2
  TRYJ(fork(2));   // q1g, du2
  TRYJ(w.wgrad(a->w, 1, col, du2, a->d2, qc, 1, sk, skn));                                                 // (du2 .* d2)^T spread(w)
  TRYJ(permute021_add(qa, qc, nullptr, a->dW3, Cw, 9, C, sst, ldq));
  TRYJ(indm_f16_wgrad_gemm(a->t3, a->d2, q1g, a->dW2T, c.M, 1.0f, 0, sk, skn, sst));                        // dW2T = (t3 .* d2)^T (q1 .* d1)
  void* dp2 = bufE;
  TRYJ(indm_f16_bwd_act(da2, du2, a->d2, a->a2, a->t3, dp2, a->dcb, c.M, rps, stream));
  TRYJ(fork(3));   // dp2, dcb
  TRYJ(indm_colreduce_f32(a->dcb, nullptr, a->db2, c.N, Cw, 0, 1.0f, 0.0f, cr_n, sst));
  TRYJ(indm_f16_wgrad_gemm(a->a1, nullptr, dp2, a->dW2, c.M, 1.0f, 0, sk, skn, sst));                       // dW2 = a1^T dp2
  // ---- level 2:  p2 = a1 W2 + b2 + cb[image] ----
  void* da1 = bufA;
  TRYJ(indm_f16_rows(F16R_GEMM, dp2, a->W2T_bf16w, nullptr, nullptr, nullptr, da1, nullptr, nullptr, nullptr, Mi, 0, 1, stream));  // da1 = dp2 W2^T
  void* dp1 = bufF;
  TRYJ(indm_f16_bwd_act(da1, q1, a->d1, a->a1, a->t2, dp1, seg1, c.M, rps, stream));
  TRYJ(fork(4));   // dp1, seg1
  TRYJ(indm_colreduce_f32(seg1, nullptr, a->db1, c.N, Cw, 0, 1.0f, 0.0f, cr_n, sst));
  // ---- level 1:  p1 = conv(a0, W1) + b1 ----
  TRYJ(indm_f16_rows(F16R_THINN, dp1, nullptr, nullptr, a->panel_bf16w, Pn, nullptr, nullptr, nullptr, nullptr, Mi, J, 1, stream));
  TRYJ(indm_tap_gather_f32(Pn, nullptr, da0, c.N, c.H, c.W, C, -1, 1.0f, stream));
  TRYJ(w.wgrad(a->a0, -1, col, dp1, nullptr, qd, 0, sk, skn));                                             // spread(a0)^T dp1: [9C, Cw]
  TRYJ(permute021_add(qb, nullptr, qd, a->dW1, Cw, J, 1, sst, ldq));
  if (a->preact) {
    TRYJ(indm_sincos_act_bwd_f32(da0, dd0, a->a0, a->d0, tmpc, nC, stream));
    TRYJ(indm_axpby_f32(a->gy, tmpc, a->dx, 1.0f, 1.0f, 0.0f, nC, stream));
  } else {
    TRYJ(indm_axpby_f32(a->gy, da0, a->dx, 1.0f, 1.0f, 0.0f, nC, stream));
  }
#undef TRYJ
  return join() == 0 ? INDM_OK : INDM_E_LAUNCH;
}
}  // namespace

INDM_API int indm_iresblock_bwd(const indm_iresblock_bwd_args* a, void* stream) {
  if (!a) return INDM_E_ARG;
  if (a->N <= 0) return INDM_OK;
  if (a->bf16 == 2) return bwd_f16(a, stream);
  if (!covered(a->N, a->H, a->W, a->C, a->idim) || !a->gld) return 1;
  if (!a->gy || !a->a0 || !a->a1 || !a->d1 || !a->a2 || !a->d2 || !a->t3 || !a->t2 || !a->w || !a->eps || !a->W1 ||
      !a->W2 || !a->W2T || !a->W3_tapco_ci || (a->preact && (!a->d0 || !a->c1)) || !a->dx || !a->dcb || !a->dW1 ||
      !a->dW2 || !a->dW3 || !a->dW2T || !a->db1 || !a->db2 || !a->db3)
    return INDM_E_ARG;
  Ctx c = {a->N, a->H, a->W, a->C, a->idim, a->bf16, (long)a->N * a->H * a->W, a->splitk_ws, a->splitk_ws_floats, stream,
           nullptr};
  const int C = c.C, Cw = c.Cw, Mi = (int)c.M, rps = c.H * c.W;
  const long nC = c.M * C, nW = c.M * Cw, nP = c.M * 9 * C;
  Bump ws = {a->ws, a->ws_floats, 0};
  float *bufA = ws.get(nW), *bufB = ws.get(nW), *bufC = ws.get(nW), *bufD = ws.get(nW), *bufE = ws.get(nW), *bufF = ws.get(nW);
  float *col = ws.get(nP), *Pn = ws.get(nP);
  float *dvj = ws.get(nC), *q0 = ws.get(nC), *dd0 = ws.get(nC), *da0 = ws.get(nC), *tmpc = ws.get(nC);
  float *qa = ws.get((long)Cw * 9 * C), *qb = ws.get((long)Cw * 9 * C), *qc = ws.get((long)Cw * 9 * C),
        *qd = ws.get((long)Cw * 9 * C);
  float *seg1 = ws.get((long)c.N * Cw);
  float *cr_w = ws.get(indm_colreduce_ws_floats(c.M, Cw)), *act_w = ws.get(indm_ires_bwd_act_ws_floats(c.M, Cw)),
        *cr_c = ws.get(indm_colreduce_ws_floats(c.M, C)), *cr_n = ws.get(indm_colreduce_ws_floats(c.N, Cw));
  if (!bufA || !bufB || !bufC || !bufD || !bufE || !bufF || !col || !Pn || !dvj || !q0 || !dd0 || !da0 || !tmpc || !qa || !qb ||
      !qc || !qd || !seg1 || !cr_w || !act_w || !cr_c || !cr_n)
    return INDM_E_ARG;

  // ---- the side stream of the weight / bias gradients (see SideEvents above); without one, `w` IS the main context and the
  // launches below run in the order of rounds 3 - 5
  SideEvents* se = (a->side_stream && a->side_splitk_ws && a->side_stream != stream) ? side_events() : nullptr;
  Ctx w = c;
  hipStream_t ms = (hipStream_t)stream, ss = ms;
  if (se) {
    w.st = a->side_stream;
    w.sk = a->side_splitk_ws;
    w.sk_floats = a->side_splitk_ws_floats;
    ss = (hipStream_t)a->side_stream;
  }
  // side waits for what the main stream has done up to here (event i) / main waits for the side stream's last launch
  auto fork = [&](int i) -> int {
    if (!se) return 0;
    if (hipEventRecord(se->ev[i], ms) != hipSuccess || hipStreamWaitEvent(ss, se->ev[i], 0) != hipSuccess) return INDM_E_LAUNCH;
    return 0;
  };
  auto join = [&]() -> int {
    if (!se) return 0;
    if (hipEventRecord(se->ev[4], ss) != hipSuccess || hipStreamWaitEvent(ms, se->ev[4], 0) != hipSuccess) return INDM_E_LAUNCH;
    return 0;
  };
  // any error return below leaves the side stream un-joined: join first, then report
#define TRYJ(call)                                              \
  do {                                                          \
    int st_ = (call);                                           \
    if (st_ != 0) {                                             \
      (void)join();                                             \
      return st_ < 0 ? st_ : INDM_E_SHAPE;                      \
    }                                                           \
  } while (0)

  TRYJ(fork(0));   // the inputs of this call (gy, the saved tensors) are complete on the main stream
  // ---- g path, level 3:  g = conv(a2, W3) + b3 ----
  float* da2 = bufA;
  TRYJ(c.thin(a->gy, a->W3_tapco_ci, da2, 1, nullptr));                                          // da2 = C3^T gy
  TRYJ(indm_tap_spread_f32(a->gy, col, c.N, c.H, c.W, C, 1, 1.0f, w.st));                       // dW3 (g part) = a2^T spread(gy)
  TRYJ(w.gemm(a->a2, nullptr, col, qa, Cw, 9 * C, Mi, 1, 0, Cw, 9 * C, nullptr, nullptr, 1));
  TRYJ(indm_colreduce_f32(a->gy, nullptr, a->db3, c.M, C, 0, 1.0f, 0.0f, cr_c, w.st));
  // ---- reverse sweep of the last VJP chain: logdet = <d0 .* C1^T(d1 .* t2), eps>, t2 = (t3 .* d2) W2^T, t3 = C3^T w ----
  TRYJ(indm_scale_rows_f32(a->eps, a->gld, dvj, c.N, (long)rps * C, stream));
  const float* q0p = dvj;
  if (a->preact) {
    TRYJ(indm_mul2_f32(dvj, a->d0, a->c1, q0, dd0, nC, stream));
    q0p = q0;
  }
  TRYJ(fork(1));   // q0
  float* q1 = bufB;
  TRYJ(c.thin(q0p, a->W1, q1, -1, nullptr));                                                     // q1 = conv(q0, W1) = dL/d(t2 .* d1)
  TRYJ(indm_tap_spread_f32(q0p, col, c.N, c.H, c.W, C, -1, 1.0f, w.st));                        // dW1 (chain part)^T = (t2 .* d1)^T spread(q0)
  TRYJ(w.gemm(a->t2, a->d1, col, qb, Cw, 9 * C, Mi, 1, 0, Cw, 9 * C, nullptr, nullptr, 1));
  // du2 = (q1 .* d1) W2 = dL/d(t3 .* d2), and -- where the gated direct-to-LDS kernel takes the GEMM (fp32) -- the Sin level that
  // du2 feeds as the same launch's epilogue: dp2 = da2 d2 - 4 pi^2 a2 (du2 t3) and its per-image sums dcb
  float *du2 = bufC, *dp2 = bufD;
  int fusedc = 1;
  if (!c.bf16 && (gpass_fused() & 8))
    fusedc = indm_gemm_gated_ires_bwd_act_f32(q1, a->d1, a->W2, da2, a->d2, a->a2, a->t3, du2, dp2, a->dcb, Mi, Cw, Cw, Cw, Cw, rps,
                                              act_w, stream);
  if (fusedc < 0) { (void)join(); return fusedc; }
  if (fusedc == 1) {
    TRYJ(c.gemm(q1, a->d1, a->W2, du2, Mi, Cw, Cw, 0, 0, Cw, Cw, nullptr, nullptr, 1));
    TRYJ(indm_ires_bwd_act_f32(da2, du2, a->d2, a->a2, a->t3, dp2, nullptr, a->dcb, c.M, Cw, rps, act_w, stream));
  }
  TRYJ(fork(2));   // du2, dp2, dcb
  TRYJ(indm_tap_spread_f32(a->w, col, c.N, c.H, c.W, C, 1, 1.0f, w.st));                        // dW3 (chain part) = (du2 .* d2)^T spread(w)
  TRYJ(w.gemm(du2, a->d2, col, qc, Cw, 9 * C, Mi, 1, 0, Cw, 9 * C, nullptr, nullptr, 1));
  TRYJ(permute021_add(qa, qc, nullptr, a->dW3, Cw, 9, C, w.st));                                // [Cw, tap, c] -> [tap, Cw, c], both parts summed
  TRYJ(indm_colreduce_f32(a->dcb, nullptr, a->db2, c.N, Cw, 0, 1.0f, 0.0f, cr_n, w.st));
  TRYJ(w.gemm(a->a1, nullptr, dp2, a->dW2, Cw, Cw, Mi, 1, 0, Cw, Cw, nullptr, nullptr, 1));     // dW2 = a1^T dp2
  // ---- level 2:  p2 = a1 W2 + b2 + cb[image] ----
  // da1 = dp2 W2^T and the Sin level it feeds -- dp1 = da1 d1 - 4 pi^2 a1 (q1 t2), dt2 = q1 d1, per-image sums of dp1 -- as ONE
  // launch where the direct-to-LDS kernel takes the GEMM (fp32): da1 is never written, and the pass's 1.6 GB of HBM traffic
  // runs under the other workgroup's MFMAs.  (dp1, dt2 in buffers of their own: du2 and dp2 are still being read by the
  // weight gradients above when those run on the side stream.)
  float *dp1 = bufE, *dt2 = bufF;
  int fusedb = 1;
  if (!c.bf16 && (gpass_fused() & 4))
    fusedb = indm_gemm_ires_bwd_act_f32(dp2, a->W2, q1, a->d1, a->a1, a->t2, dp1, dt2, seg1, Mi, Cw, Cw, Cw, Cw, rps, act_w, stream);
  if (fusedb < 0) { (void)join(); return fusedb; }
  if (fusedb == 1) {
    float* da1 = bufA;                                                                             // da2 is dead
    TRYJ(c.gemm(dp2, nullptr, a->W2, da1, Mi, Cw, Cw, 0, 1, Cw, Cw, nullptr, nullptr, 1));      // da1 = dp2 W2^T
    TRYJ(indm_ires_bwd_act_f32(da1, q1, a->d1, a->a1, a->t2, dp1, dt2, seg1, c.M, Cw, rps, act_w, stream));
  }
  TRYJ(fork(3));   // dp1, dt2, seg1
  TRYJ(w.gemm(a->t3, a->d2, dt2, a->dW2T, Cw, Cw, Mi, 1, 0, Cw, Cw, nullptr, nullptr, 1));      // dW2T = (t3 .* d2)^T (q1 .* d1)
  TRYJ(indm_colreduce_f32(seg1, nullptr, a->db1, c.N, Cw, 0, 1.0f, 0.0f, cr_n, w.st));
  // ---- level 1:  p1 = conv(a0, W1) + b1 ----
  TRYJ(c.gemm(dp1, nullptr, a->W1, Pn, Mi, 9 * C, Cw, 0, 1, Cw, Cw, nullptr, nullptr, 1));       // da0 = C1^T dp1: GEMM + gather
  TRYJ(indm_tap_gather_f32(Pn, nullptr, da0, c.N, c.H, c.W, C, -1, 1.0f, stream));
  TRYJ(indm_tap_spread_f32(a->a0, col, c.N, c.H, c.W, C, -1, 1.0f, w.st));                      // dW1 (g part) = spread(a0)^T dp1
  TRYJ(w.gemm(col, nullptr, dp1, qd, 9 * C, Cw, Mi, 1, 0, 9 * C, Cw, nullptr, nullptr, 1));
  TRYJ(permute021_add(qb, nullptr, qd, a->dW1, Cw, 9 * C, 1, w.st));                             // qb^T + qd
  if (a->preact) {
    TRYJ(indm_sincos_act_bwd_f32(da0, dd0, a->a0, a->d0, tmpc, nC, stream));
    TRYJ(indm_axpby_f32(a->gy, tmpc, a->dx, 1.0f, 1.0f, 0.0f, nC, stream));
  } else {
    TRYJ(indm_axpby_f32(a->gy, da0, a->dx, 1.0f, 1.0f, 0.0f, nC, stream));
  }
#undef TRYJ
  return join() == 0 ? INDM_OK : INDM_E_LAUNCH;
}

INDM_API int indm_iresblock_inverse(const indm_iresblock_fwd_args* a, float atol, float rtol, int* iterations,
                                    void* stream) {
  if (!a) return INDM_E_ARG;
  if (a->N <= 0) return INDM_OK;
  if (!covered(a->N, a->H, a->W, a->C, a->idim)) return 1;
  if (!a->x || !a->cb || !a->W1 || !a->W2T || !a->W3_ci_tapco || !a->b1 || !a->b2 || !a->b3 || !a->y) return INDM_E_ARG;
  Ctx c = {a->N, a->H, a->W, a->C, a->idim, a->bf16, (long)a->N * a->H * a->W, a->splitk_ws, a->splitk_ws_floats, stream,
           nullptr};
  const long nC = c.M * c.C, nW = c.M * c.Cw, nP = c.M * 9 * c.C;
  Bump ws = {a->ws, a->ws_floats, 0};
  float *a0 = ws.get(nC), *a1 = ws.get(nW), *a2 = ws.get(nW), *p = ws.get(nW), *Pn = ws.get(nP), *g = ws.get(nC);
  float *xa = ws.get(nC), *xb = ws.get(nC), *cnt = ws.get(64);
  if (!a0 || !a1 || !a2 || !p || !Pn || !g || !xa || !xb || !cnt) return INDM_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const float* y = a->x;   // the block's OUTPUT is the input of the inverse
  // x <- y - g(x), starting from x = y (iresblock.py:78-88)
  TRY(g_pass(c, a->preact, y, a->cb, a->W1, a->W2T, a->W3_ci_tapco, a->b1, a->b2, a->b3, a0, nullptr, a1, nullptr, a2,
             nullptr, p, Pn, g));
  float *x = xa, *other = xb;
  TRY(indm_axpby_f32(y, g, x, 1.0f, -1.0f, 0.0f, nC, stream));
  const float* x_prev = y;
  int i = 0;
  for (;;) {
    TRY(indm_fixed_point_check_f32(x, x_prev, y, nC, atol, rtol, cnt, stream));
    float h = 0.f;
    if (hipMemcpyAsync(&h, cnt, sizeof(float), hipMemcpyDeviceToHost, st) != hipSuccess) return INDM_E_LAUNCH;
    if (hipStreamSynchronize(st) != hipSuccess) return INDM_E_LAUNCH;
    if (h == 0.f) break;
    TRY(g_pass(c, a->preact, x, a->cb, a->W1, a->W2T, a->W3_ci_tapco, a->b1, a->b2, a->b3, a0, nullptr, a1, nullptr, a2,
               nullptr, p, Pn, g));
    TRY(indm_axpby_f32(y, g, other, 1.0f, -1.0f, 0.0f, nC, stream));
    x_prev = x;
    float* t = x; x = other; other = t;
    if (++i > 1000) break;
  }
  if (hipMemcpyAsync(a->y, x, nC * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return INDM_E_LAUNCH;
  if (iterations) *iterations = i;
  return INDM_OK;
}

// ---- launch timer ------------------------------------------------------------------------------------------------
INDM_API int indm_timer_create(int capacity, void** handle) {
  if (!handle || capacity <= 0) return INDM_E_ARG;
  *handle = nullptr;
  Timer* t = new Timer();
  t->ev.reserve(2 * (size_t)capacity);
  for (size_t i = 0; i < 2 * (size_t)capacity; ++i) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) {   // give back what was created: the caller gets no handle to destroy
      for (auto& d : t->ev) (void)hipEventDestroy(d);
      delete t;
      return INDM_E_LAUNCH;
    }
    t->ev.push_back(e);
  }
  *handle = t;
  return INDM_OK;
}
INDM_API int indm_timer_destroy(void* handle) {
  Timer* t = (Timer*)handle;
  if (!t) return INDM_OK;
  for (auto& e : t->ev) (void)hipEventDestroy(e);
  delete t;
  return INDM_OK;
}
INDM_API int indm_timer_reset(void* handle, int cls_mask) {
  Timer* t = (Timer*)handle;
  if (!t) return INDM_E_ARG;
  t->used = 0;
  t->mask = cls_mask;
  t->flops = 0.0;
  return INDM_OK;
}
INDM_API int indm_timer_read(void* handle, float* total_ms, int* launches, double* flops) {
  Timer* t = (Timer*)handle;
  if (!t || !total_ms || !launches) return INDM_E_ARG;
  float sum = 0.f;
  for (int i = 0; i + 1 < t->used; i += 2) {
    if (hipEventSynchronize(t->ev[i + 1]) != hipSuccess) return INDM_E_LAUNCH;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, t->ev[i], t->ev[i + 1]) != hipSuccess) return INDM_E_LAUNCH;
    sum += ms;
  }
  *total_ms = sum;
  *launches = t->used / 2;
  if (flops) *flops = t->flops;
  return INDM_OK;
}

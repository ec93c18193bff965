// HBM-bound elementwise / permutation kernels (fp32, float4-vectorised, grid-stride).
//   * Sin activation of the residual-flow g-net and its derivative in ONE pass
//     (flow_models/wolf/flows/resflow/layers/base/activations.py:7-12)
//   * SiLU / ELU, axpby, mul, per-image dot products (Hutchinson traces, iresblock.py:253-273)
//   * nearest x2 upsample / 2x2 mean downsample (models/up_or_down_sampling.py:59-69)
//   * squeeze / unsqueeze space-to-depth (flows/resflow/layers/squeeze.py:27-45) in NHWC
//   * NCHW <-> NHWC layout conversion at the API edge
//   * Lop weight normalisation (layers/base/lipschitz.py:350-359) on RSCK weights
#include "indm_common.h"

namespace {

inline bool al16(const void* q) { return ((uintptr_t)q & 15) == 0; }

constexpr float TWO_PI = 6.283185307179586f;
constexpr float INV_TWO_PI = 0.15915494309189535f;

template <typename F>
__global__ __launch_bounds__(256) void ew_kernel(long n, F f) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) f(i);
}
template <typename F>
int ew_launch(long n, hipStream_t st, F f) {
  if (n <= 0) return INDM_OK;
  hipLaunchKernelGGL(ew_kernel<F>, dim3(ew_grid(n, 256)), dim3(256), 0, st, n, f);
  if (hipGetLastError() != hipSuccess) return INDM_E_LAUNCH;
  return INDM_OK;
}

// per-image reduction: out[b] = alpha * sum_i f(i) over D contiguous elements; one block per (b, chunk)
template <int MODE>
__global__ __launch_bounds__(256) void rowreduce_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                        float* __restrict__ part, long D, int chunks) {
  __shared__ float sh[16];
  const int b = blockIdx.y, c = blockIdx.x;
  const long per = ((D + chunks - 1) / chunks + 3) & ~3L;
  const long i0 = (long)c * per, i1 = min(D, i0 + per);
  const float* xb = x + (long)b * D;
  const float* yb = y ? y + (long)b * D : nullptr;
  float s = 0.f;
  for (long i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
    const float a = xb[i];
    if (MODE == 0) s += a * yb[i];
    else if (MODE == 1) s += a * a;
    else s += a;
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) part[(long)b * chunks + c] = s;
}
__global__ void rowreduce_final_kernel(const float* __restrict__ part, float* __restrict__ out, int B, int chunks,
                                       float alpha) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float s = 0.f;
  for (int c = 0; c < chunks; ++c) s += part[(long)b * chunks + c];
  out[b] = alpha * s;
}

// Lop normalisation over RSCK weight viewed as [R, Co]: per-column L1 norm.  grid (Co/64, RB): each workgroup
// owns 64 columns x one row-block (4 row lanes inside).  The hosts below launch ONE row-block (gridDim.y == 1): the add
// onto the zeroed column is then exact and the result has a fixed summation order (the training path normalises all
// weights through lop_multi_*; this single-weight form serves stand-alone InducedNormConv2d calls).
__global__ __launch_bounds__(256) void lop_colnorm_kernel(const float* __restrict__ w, float* __restrict__ l1, int R,
                                                          int Co) {
  __shared__ float sh[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const int rows = (R + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * rows, r1 = min(R, r0 + rows);
  float s = 0.f;
  if (c < Co)
    for (int r = r0 + rl; r < r1; r += 4) s += fabsf(w[(long)r * Co + c]);
  sh[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < Co)
    atomicAdd(&l1[c], sh[0][threadIdx.x & 63] + sh[1][threadIdx.x & 63] + sh[2][threadIdx.x & 63] + sh[3][threadIdx.x & 63]);
}
// dot[c] = sum_r dwn[r,c]*w[r,c]
__global__ __launch_bounds__(256) void lop_coldot_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ out, int R, int Co) {
  __shared__ float sh[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const int rows = (R + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * rows, r1 = min(R, r0 + rows);
  float s = 0.f;
  if (c < Co)
    for (int r = r0 + rl; r < r1; r += 4) s += a[(long)r * Co + c] * b[(long)r * Co + c];
  sh[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < Co)
    atomicAdd(&out[c], sh[0][threadIdx.x & 63] + sh[1][threadIdx.x & 63] + sh[2][threadIdx.x & 63] + sh[3][threadIdx.x & 63]);
}

// counts elements violating (x-xp)^2 / (atol + |y|*rtol) < 1  (iresblock.py:81-82)
__global__ __launch_bounds__(256) void fp_check_kernel(const float* __restrict__ x, const float* __restrict__ xp,
                                                       const float* __restrict__ y, long n, float atol, float rtol,
                                                       float* __restrict__ out) {
  __shared__ float sh[16];
  float c = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float d = x[i] - xp[i];
    const float tol = atol + fabsf(y[i]) * rtol;
    c += (d * d / tol < 1.f) ? 0.f : 1.f;
  }
  c = block_sum(c, sh);
  if (threadIdx.x == 0 && c != 0.f) atomicAdd(out, c);
}

// Tap gather for 3x3 convolutions with a skinny channel side (flow edges C = 3 / 12):
//   y[n,h,w,c] = alpha * ( sum_{kh,kw} P[n, h + s*(kh-1), w + s*(kw-1), (kh*3+kw)*C + c] + bias[c] )
// P is the [pixels, 9*C] product of the wide side with the tap-major weight (an MFMA GEMM with N = 9C instead
// of nine N = C ones).  s = +1: forward conv; s = -1: data-grad / transposed conv.
// optional tail (the per-iteration bookkeeping of the residual flow's series, iresblock.py:264-273, in the same pass):
//   y = alpha*(gather + bias) * mulp   and   acc_out = acc_in + coeff * y
__global__ __launch_bounds__(256) void tap_gather_kernel(const float* __restrict__ P, const float* __restrict__ bias,
                                                         float* __restrict__ y, int N, int H, int W, int C, int s,
                                                         float alpha, const float* __restrict__ mulp,
                                                         const float* __restrict__ acc_in, float* __restrict__ acc_out,
                                                         float coeff) {
  const long total = (long)N * H * W * C;
  const int C9 = 9 * C;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {
    const unsigned pix = i / (unsigned)C;
    const int c = (int)(i - pix * (unsigned)C);
    const unsigned row = pix / (unsigned)W;
    const int w = (int)(pix - row * (unsigned)W);
    const int n = (int)(row / (unsigned)H);
    const int h = (int)(row - (unsigned)n * (unsigned)H);
    // all nine taps requested before the first is added (round 5: with `if (outside) continue` around each load hipcc
    // waited for every tap with vmcnt(0) before it requested the next -- nine memory latencies in sequence per element,
    // 8 us per launch, 220 launches per step); an outside tap reads the pixel's own panel entry and adds 0.0f instead:
    // the same sum in the same order
    float t[9];
    bool ok[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int hh = h + s * (kh - 1);
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ww = w + s * (kw - 1);
        const int tp = kh * 3 + kw;
        ok[tp] = hh >= 0 && hh < H && ww >= 0 && ww < W;
        t[tp] = P[(((long)n * H + (ok[tp] ? hh : h)) * W + (ok[tp] ? ww : w)) * C9 + tp * C + c];
      }
    }
    float acc = bias ? bias[c] : 0.f;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) acc += ok[tp] ? t[tp] : 0.f;
    float v = alpha * acc;
    if (mulp) v *= mulp[i];
    y[i] = v;
    if (acc_out) acc_out[i] = acc_in[i] + coeff * v;
  }
}
// adjoint of the gather: Pout[n,h',w',(tap,c)] = alpha * y[n, h' - s*(kh-1), w' - s*(kw-1), c]  (0 outside)
__global__ __launch_bounds__(256) void tap_spread_kernel(const float* __restrict__ y, float* __restrict__ P, int N,
                                                         int H, int W, int C, int s, float alpha) {
  // one thread per (pixel, tap): the C (3 or 12) channels of a tap are contiguous in both tensors, so the index
  // arithmetic (32-bit; the host checks the sizes) is paid once per tap instead of once per element
  const unsigned total = (unsigned)N * H * W * 9u;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned pix = i / 9u;
    const int tap = (int)(i - pix * 9u);
    const unsigned row = pix / (unsigned)W;
    const int w = (int)(pix - row * (unsigned)W);
    const int n = (int)(row / (unsigned)H);
    const int h = (int)(row - (unsigned)n * (unsigned)H);
    const int kh = tap / 3, kw = tap - kh * 3;
    const int hh = h - s * (kh - 1), ww = w - s * (kw - 1);
    float* dst = P + (long)i * C;
    if (hh >= 0 && hh < H && ww >= 0 && ww < W) {
      const float* src = y + (((long)n * H + hh) * W + ww) * C;
      for (int c = 0; c < C; ++c) dst[c] = alpha * src[c];
    } else {
      for (int c = 0; c < C; ++c) dst[c] = 0.f;
    }
  }
}

}  // namespace

// a = sin(2*pi*p)/(2*pi), d = cos(2*pi*p)   (d may be null)
INDM_API int indm_sincos_act_fwd_f32(const float* p, float* a, float* d, long n, void* stream) {
  return ew_launch(n, (hipStream_t)stream, [=] __device__(long i) {
    float s, c;
    indm_sincos(TWO_PI * p[i], s, c);
    a[i] = s * INV_TWO_PI;
    if (d) d[i] = c;
  });
}
// dp = da * d + dd * (-4*pi^2 * a)   (da / dd may be null)
INDM_API int indm_sincos_act_bwd_f32(const float* da, const float* dd, const float* a, const float* d, float* dp,
                                     long n, void* stream) {
  return ew_launch(n, (hipStream_t)stream, [=] __device__(long i) {
    float g = 0.f;
    if (da) g += da[i] * d[i];
    if (dd) g -= dd[i] * (TWO_PI * TWO_PI) * a[i];
    dp[i] = g;
  });
}

// ---- hand-written backward of the residual-flow block (flow.py::_IResTrain) --------------------------------------
// One pass replaces autograd's mul2 + sincos_act_bwd (+ the segmented bias reduction) at one Sin level of the block:
//   u = t .* d  is a link of the explicit VJP chain (d = cos(2 pi p) the gate, t the incoming vector) and
//   a = sin(2 pi p)/(2 pi) the activation of g itself, so with  da = dL/da (through g)  and  du = dL/du (through the
//   chain):   dp = da*d - 4 pi^2 * a * (du*t)          (dd = du*t is never written)
//             dt = du*d                                  (optional second output: the chain's next cotangent)
//             seg[s][c] = sum over the rows of segment s of dp[., c]   (optional: per-image bias gradient)
// Rows are [R, C] with C contiguous; a workgroup owns 64 consecutive rows (all columns), so the column sums of dp come
// for free in registers: partial sums per (64-row chunk, column) go to `ws`, a second tiny kernel folds the chunks of
// each segment.  Reads 5 tensors, writes 1 (2): 6 (7) passes instead of the 10 of mul2 + sincos_act_bwd.
namespace {
constexpr int IRES_CH = 64;
template <bool HAS_DT, bool HAS_SEG>
__global__ __launch_bounds__(256) void ires_bwd_act_kernel(const float* __restrict__ da, const float* __restrict__ du,
                                                           const float* __restrict__ d, const float* __restrict__ a,
                                                           const float* __restrict__ t, float* __restrict__ dp,
                                                           float* __restrict__ dt, float* __restrict__ part, long R,
                                                           int C) {
  const float K = TWO_PI * TWO_PI;
  for (long chunk = blockIdx.x; chunk * IRES_CH < R; chunk += gridDim.x) {
    const long r0 = chunk * IRES_CH;
    const int nr = (int)min((long)IRES_CH, R - r0);
    for (int c = threadIdx.x; c < C; c += 256) {
      float s = 0.f;
      long i = r0 * C + c;
#pragma unroll 4
      for (int r = 0; r < nr; ++r, i += C) {
        const float uu = du[i], dd = d[i];
        const float g = da[i] * dd - K * a[i] * (uu * t[i]);
        dp[i] = g;
        if (HAS_DT) dt[i] = uu * dd;
        if (HAS_SEG) s += g;
      }
      if (HAS_SEG) part[chunk * C + c] = s;
    }
  }
}
__global__ __launch_bounds__(256) void ires_seg_fold_kernel(const float* __restrict__ part, float* __restrict__ seg,
                                                            long nseg, int cps, int C) {
  const long total = nseg * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long sgm = i / C;
    const int c = (int)(i - sgm * C);
    const float* q = part + (sgm * cps) * C + c;
    float s = 0.f;
    for (int k = 0; k < cps; ++k) s += q[(long)k * C];
    seg[i] = s;
  }
}
}  // namespace

INDM_API long indm_ires_bwd_act_ws_floats(long R, int C) { return ((R + IRES_CH - 1) / IRES_CH) * (long)C; }

// the fold of the 64-row partial sums, for the GEMM that carries the pass as its epilogue (igemm.hip, not exported)
int indm_ires_seg_fold(const float* part, float* seg, long nseg, int cps, int C, hipStream_t st) {
  hipLaunchKernelGGL(ires_seg_fold_kernel, dim3(ew_grid(nseg * C, 256)), dim3(256), 0, st, part, seg, nseg, cps, C);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}

// da, du, d, a, t, dp (and dt_out if non-null): [R, C]; seg_out (if non-null): [R / rows_per_seg, C], which needs
// rows_per_seg % 64 == 0 and ws of indm_ires_bwd_act_ws_floats(R, C) floats.
INDM_API int indm_ires_bwd_act_f32(const float* da, const float* du, const float* d, const float* a, const float* t,
                                   float* dp, float* dt_out, float* seg_out, long R, int C, long rows_per_seg,
                                   float* ws, void* stream) {
  if (R < 0 || C <= 0) return INDM_E_SHAPE;
  if (R == 0) return INDM_OK;
  if (seg_out && (rows_per_seg <= 0 || rows_per_seg % IRES_CH || R % rows_per_seg || !ws)) return INDM_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const long chunks = (R + IRES_CH - 1) / IRES_CH;
  dim3 grid((unsigned)(chunks > 4096 ? 4096 : chunks)), block(256);
  if (dt_out && seg_out) hipLaunchKernelGGL((ires_bwd_act_kernel<true, true>), grid, block, 0, st, da, du, d, a, t, dp, dt_out, ws, R, C);
  else if (dt_out) hipLaunchKernelGGL((ires_bwd_act_kernel<true, false>), grid, block, 0, st, da, du, d, a, t, dp, dt_out, ws, R, C);
  else if (seg_out) hipLaunchKernelGGL((ires_bwd_act_kernel<false, true>), grid, block, 0, st, da, du, d, a, t, dp, dt_out, ws, R, C);
  else hipLaunchKernelGGL((ires_bwd_act_kernel<false, false>), grid, block, 0, st, da, du, d, a, t, dp, dt_out, ws, R, C);
  INDM_CHECK_LAUNCH();
  if (seg_out) {
    const long nseg = R / rows_per_seg;
    hipLaunchKernelGGL(ires_seg_fold_kernel, dim3(ew_grid(nseg * C, 256)), block, 0, st, ws, seg_out, nseg,
                       (int)(rows_per_seg / IRES_CH), C);
    INDM_CHECK_LAUNCH();
  }
  return INDM_OK;
}

// act: 1 = SiLU, 2 = ELU(alpha=1)
INDM_API int indm_act_fwd_f32(const float* x, float* y, long n, int act, void* stream) {
  return ew_launch(n, (hipStream_t)stream, [=] __device__(long i) {
    const float u = x[i];
    y[i] = act == 1 ? u / (1.f + expf(-u)) : (u > 0.f ? u : expm1f(u));
  });
}
INDM_API int indm_act_bwd_f32(const float* dy, const float* x, float* dx, long n, int act, void* stream) {
  return ew_launch(n, (hipStream_t)stream, [=] __device__(long i) {
    const float u = x[i];
    float g;
    if (act == 1) {
      const float s = 1.f / (1.f + expf(-u));
      g = s * (1.f + u * (1.f - s));
    } else {
      g = u > 0.f ? 1.f : expf(u);
    }
    dx[i] = dy[i] * g;
  });
}

// out = a*x + b*y + c   (y may be null)
INDM_API int indm_axpby_f32(const float* x, const float* y, float* out, float a, float b, float c, long n,
                            void* stream) {
  if ((n & 3) == 0 && (((uintptr_t)x | (uintptr_t)out | (uintptr_t)y) & 15) == 0) {
    const float4* x4 = (const float4*)x;
    const float4* y4 = (const float4*)y;
    float4* o4 = (float4*)out;
    return ew_launch(n >> 2, (hipStream_t)stream, [=] __device__(long i) {
      float4 v = x4[i];
      v.x = a * v.x; v.y = a * v.y; v.z = a * v.z; v.w = a * v.w;
      if (y4) {
        const float4 w = y4[i];
        v.x += b * w.x; v.y += b * w.y; v.z += b * w.z; v.w += b * w.w;
      }
      v.x += c; v.y += c; v.z += c; v.w += c;
      o4[i] = v;
    });
  }
  return ew_launch(n, (hipStream_t)stream, [=] __device__(long i) { out[i] = a * x[i] + (y ? b * y[i] : 0.f) + c; });
}

// out = alpha * x * y
INDM_API int indm_mul_f32(const float* x, const float* y, float* out, float alpha, long n, void* stream) {
  return ew_launch(n, (hipStream_t)stream, [=] __device__(long i) { out[i] = alpha * x[i] * y[i]; });
}

// out1 = x * y1, out2 = x * y2: the two products of a gated contraction's backward (d(a.*g) -> da, dg) in one pass
INDM_API int indm_mul2_f32(const float* x, const float* y1, const float* y2, float* out1, float* out2, long n,
                           void* stream) {
  // (a float4 version of these streaming kernels measured 5.3 TB/s against 5.6 TB/s for one float per lane)
  return ew_launch(n, (hipStream_t)stream, [=] __device__(long i) {
    const float a = x[i];
    out1[i] = a * y1[i];
    out2[i] = a * y2[i];
  });
}

// out[b, j] = x[b, j] * s[b] (+ bias)
INDM_API int indm_scale_rows_f32(const float* x, const float* s, float* out, long B, long D, void* stream) {
  return ew_launch(B * D, (hipStream_t)stream, [=] __device__(long i) { out[i] = x[i] * s[i / D]; });
}

INDM_API long indm_rowreduce_ws_floats(long B, long D) {
  int chunks = (int)((D + 8191) / 8192);
  if (chunks < 1) chunks = 1;
  return B * chunks;
}
// out[b] = alpha * reduce_i f(x[b,i], y[b,i]); mode 0: x*y, 1: x*x, 2: x
INDM_API int indm_rowreduce_f32(const float* x, const float* y, float* out, long B, long D, int mode, float alpha,
                                float* ws, void* stream) {
  if (B <= 0) return INDM_OK;
  hipStream_t st = (hipStream_t)stream;
  int chunks = (int)((D + 8191) / 8192);
  if (chunks < 1) chunks = 1;
  dim3 grid(chunks, (unsigned)B);
  if (mode == 0) hipLaunchKernelGGL(rowreduce_kernel<0>, grid, dim3(256), 0, st, x, y, ws, D, chunks);
  else if (mode == 1) hipLaunchKernelGGL(rowreduce_kernel<1>, grid, dim3(256), 0, st, x, y, ws, D, chunks);
  else hipLaunchKernelGGL(rowreduce_kernel<2>, grid, dim3(256), 0, st, x, y, ws, D, chunks);
  INDM_CHECK_LAUNCH();
  hipLaunchKernelGGL(rowreduce_final_kernel, dim3(cdiv(B, 256)), dim3(256), 0, st, ws, out, (int)B, chunks, alpha);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}

// nearest-neighbour x2 upsample, NHWC: y[n,2h+a,2w+b,c] = alpha * x[n,h,w,c]
INDM_API int indm_upsample2x_f32(const float* x, float* y, int N, int H, int W, int C, float alpha, void* stream) {
  const long total = (long)N * 4 * H * W * C;
  return ew_launch(total, (hipStream_t)stream, [=] __device__(long i) {
    const int c = (int)(i % C);
    long r = i / C;
    const int ow = (int)(r % (2 * W));
    r /= 2 * W;
    const int oh = (int)(r % (2 * H));
    const int n = (int)(r / (2 * H));
    y[i] = alpha * x[(((long)n * H + (oh >> 1)) * W + (ow >> 1)) * C + c];
  });
}
// 2x2 sum-pool scaled by alpha (alpha = 0.25 -> mean), NHWC
INDM_API int indm_downsample2x_f32(const float* x, float* y, int N, int H, int W, int C, float alpha, void* stream) {
  const int OH = H / 2, OW = W / 2;
  const long total = (long)N * OH * OW * C;
  return ew_launch(total, (hipStream_t)stream, [=] __device__(long i) {
    const int c = (int)(i % C);
    long r = i / C;
    const int ow = (int)(r % OW);
    r /= OW;
    const int oh = (int)(r % OH);
    const int n = (int)(r / OH);
    const float* b = x + (((long)n * H + 2 * oh) * W + 2 * ow) * C + c;
    y[i] = alpha * (b[0] + b[C] + b[(long)W * C] + b[(long)W * C + C]);
  });
}

// squeeze (space-to-depth x2) in NHWC with the reference's channel order c*4 + dy*2 + dx:
//   y[n, h, w, c*4+dy*2+dx] = x[n, 2h+dy, 2w+dx, c];   x: [N,H,W,C] -> y: [N,H/2,W/2,4C]
INDM_API int indm_squeeze2x_f32(const float* x, float* y, int N, int H, int W, int C, void* stream) {
  const long total = (long)N * H * W * C;
  const int OH = H / 2, OW = W / 2;
  return ew_launch(total, (hipStream_t)stream, [=] __device__(long i) {  // i indexes y
    const int co = (int)(i % (4 * C));
    long r = i / (4 * C);
    const int w = (int)(r % OW);
    r /= OW;
    const int h = (int)(r % OH);
    const int n = (int)(r / OH);
    const int c = co >> 2, dy = (co >> 1) & 1, dx = co & 1;
    y[i] = x[(((long)n * H + 2 * h + dy) * W + 2 * w + dx) * C + c];
  });
}
// inverse of the above: x: [N,H,W,4C] -> y: [N,2H,2W,C]
INDM_API int indm_unsqueeze2x_f32(const float* x, float* y, int N, int H, int W, int C4, void* stream) {
  const int C = C4 / 4;
  const long total = (long)N * H * W * C4;
  return ew_launch(total, (hipStream_t)stream, [=] __device__(long i) {  // i indexes y [N,2H,2W,C]
    const int c = (int)(i % C);
    long r = i / C;
    const int ow = (int)(r % (2 * W));
    r /= 2 * W;
    const int oh = (int)(r % (2 * H));
    const int n = (int)(r / (2 * H));
    y[i] = x[(((long)n * H + (oh >> 1)) * W + (ow >> 1)) * C4 + c * 4 + (oh & 1) * 2 + (ow & 1)];
  });
}

// layout conversion at the API edge: NCHW -> NHWC (dir 0) or NHWC -> NCHW (dir 1)
INDM_API int indm_nchw_nhwc_f32(const float* x, float* y, int N, int C, int HW, int dir, void* stream) {
  const long total = (long)N * C * HW;
  return ew_launch(total, (hipStream_t)stream, [=] __device__(long i) {
    if (dir == 0) {  // i indexes y [N,HW,C]
      const int c = (int)(i % C);
      const long r = i / C;
      const int p = (int)(r % HW);
      const int n = (int)(r / HW);
      y[i] = x[((long)n * C + c) * HW + p];
    } else {  // i indexes y [N,C,HW]
      const int p = (int)(i % HW);
      const long r = i / HW;
      const int c = (int)(r % C);
      const int n = (int)(r / C);
      y[i] = x[((long)n * HW + p) * C + c];
    }
  });
}

// Column blocks of x[R, Ctot] <-> one flat buffer in which block b is a CONTIGUOUS [R, C_b] matrix at offset
// R * offs[b] (C_b = offs[b+1] - offs[b]; offs: device int32[nblk + 1]).  dir 0: x -> flat, dir 1: flat -> x.
// (The score net computes the time-embedding bias of all of its residual blocks in one GEMM; every block's conv
// epilogue wants its own columns as a dense [images, C_b] matrix.)
INDM_API int indm_split_cols_f32(const float* x, float* y, const int* offs, int nblk, long R, int Ctot, int dir,
                                 void* stream) {
  if (nblk <= 0 || R < 0 || Ctot <= 0 || (dir != 0 && dir != 1)) return INDM_E_ARG;
  return ew_launch(R * Ctot, (hipStream_t)stream, [=] __device__(long i) {  // i indexes the [R, Ctot] side
    const long r = i / Ctot;
    const int c = (int)(i - r * Ctot);
    int lo = 0, hi = nblk;  // block of column c: offs[lo] <= c < offs[lo + 1]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (offs[mid] <= c) lo = mid; else hi = mid;
    }
    const int o = offs[lo], cb = offs[lo + 1] - o;
    const long j = R * o + r * cb + (c - o);
    if (dir == 0) y[j] = x[i];
    else y[i] = x[j];
  });
}

// weight-normalised linear layer (nn.utils.weight_norm): w[o,:] = v[o,:] * g[o] / ||v[o,:]||_2 ; one wave per row.
namespace {
__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__global__ __launch_bounds__(256) void weightnorm_fwd_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                             float* __restrict__ w, float* __restrict__ nrm, int O, int I) {
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (o >= O) return;
  const float* vr = v + (long)o * I;
  float s = 0.f;
  for (int i = lane; i < I; i += 64) s += vr[i] * vr[i];
  const float n = sqrtf(wave_sum(s));
  const float sc = g[o] / n;
  for (int i = lane; i < I; i += 64) w[(long)o * I + i] = vr[i] * sc;
  if (lane == 0) nrm[o] = n;
}
__global__ __launch_bounds__(256) void weightnorm_bwd_kernel(const float* __restrict__ dw, const float* __restrict__ v,
                                                             const float* __restrict__ g, const float* __restrict__ nrm,
                                                             float* __restrict__ dv, float* __restrict__ dg, int O, int I) {
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (o >= O) return;
  const float* vr = v + (long)o * I;
  const float* dr = dw + (long)o * I;
  float s = 0.f;
  for (int i = lane; i < I; i += 64) s += dr[i] * vr[i];
  const float dot = wave_sum(s), n = nrm[o], gg = g[o];
  const float a = gg / n, b = dot * gg / (n * n * n);
  for (int i = lane; i < I; i += 64) dv[(long)o * I + i] = dr[i] * a - vr[i] * b;
  if (lane == 0) dg[o] = dot / n;
}
}  // namespace
// v, w: [O, I]; g, nrm: [O]
INDM_API int indm_weightnorm_fwd_f32(const float* v, const float* g, float* w, float* nrm, int O, int I, void* stream) {
  if (O <= 0 || I <= 0) return INDM_E_SHAPE;
  hipLaunchKernelGGL(weightnorm_fwd_kernel, dim3(cdiv(O, 4)), dim3(256), 0, (hipStream_t)stream, v, g, w, nrm, O, I);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}
INDM_API int indm_weightnorm_bwd_f32(const float* dw, const float* v, const float* g, const float* nrm, float* dv,
                                     float* dg, int O, int I, void* stream) {
  if (O <= 0 || I <= 0) return INDM_E_SHAPE;
  hipLaunchKernelGGL(weightnorm_bwd_kernel, dim3(cdiv(O, 4)), dim3(256), 0, (hipStream_t)stream, dw, v, g, nrm, dv, dg, O, I);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}

// Lop normalisation forward: wn[r,c] = w[r,c] / max(1, l1[c]/coeff);  l1[c] = sum_r |w[r,c]|
INDM_API int indm_lop_normalize_fwd_f32(const float* w, float* wn, float* l1, int R, int Co, float coeff,
                                        void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int RB = 1;  // one row-block per column group: fixed summation order
  if (hipMemsetAsync(l1, 0, sizeof(float) * Co, st) != hipSuccess) return INDM_E_LAUNCH;
  hipLaunchKernelGGL(lop_colnorm_kernel, dim3(cdiv(Co, 64), RB), dim3(256), 0, st, w, l1, R, Co);
  INDM_CHECK_LAUNCH();
  return ew_launch((long)R * Co, st, [=] __device__(long i) {
    const int c = (int)(i % Co);
    wn[i] = w[i] / fmaxf(1.f, l1[c] / coeff);
  });
}
// backward: dw = dwn/f - [l1/coeff > 1] * sign(w) * dot[c] / (f^2 * coeff), dot[c] = sum_r dwn*w
INDM_API int indm_lop_normalize_bwd_f32(const float* dwn, const float* w, const float* l1, float* dw, float* dot,
                                        int R, int Co, float coeff, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int RB = 1;
  if (hipMemsetAsync(dot, 0, sizeof(float) * Co, st) != hipSuccess) return INDM_E_LAUNCH;
  hipLaunchKernelGGL(lop_coldot_kernel, dim3(cdiv(Co, 64), RB), dim3(256), 0, st, dwn, w, dot, R, Co);
  INDM_CHECK_LAUNCH();
  return ew_launch((long)R * Co, st, [=] __device__(long i) {
    const int c = (int)(i % Co);
    const float s = l1[c] / coeff;
    const float f = fmaxf(1.f, s);
    float g = dwn[i] / f;
    if (s > 1.f) {
      const float wv = w[i];
      const float sg = wv > 0.f ? 1.f : (wv < 0.f ? -1.f : 0.f);
      g -= sg * dot[c] / (f * f * coeff);
    }
    dw[i] = g;
  });
}

// out[0] = number of elements where NOT ((x-xp)^2 / (atol + |y|*rtol) < 1)  -- fixed-point stop test
INDM_API int indm_fixed_point_check_f32(const float* x, const float* xp, const float* y, long n, float atol,
                                        float rtol, float* out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(out, 0, sizeof(float), st) != hipSuccess) return INDM_E_LAUNCH;
  if (n <= 0) return INDM_OK;
  hipLaunchKernelGGL(fp_check_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, st, x, xp, y, n, atol, rtol, out);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}

INDM_API int indm_tap_gather_f32(const float* P, const float* bias, float* y, int N, int H, int W, int C, int sign,
                                 float alpha, void* stream) {
  const long total = (long)N * H * W * C;
  if (total <= 0) return INDM_OK;
  if (total >= (1L << 31)) return INDM_E_SHAPE;
  hipLaunchKernelGGL(tap_gather_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, P, bias, y, N, H,
                     W, C, sign, alpha, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, 0.f);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}
// tap gather with the series bookkeeping fused: y = alpha*(gather + bias) .* mulp (mulp optional);
// acc_out = acc_in + coeff*y (acc_in / acc_out optional, together)
INDM_API int indm_tap_gather_acc_f32(const float* P, const float* bias, const float* mulp, float* y, const float* acc_in,
                                     float* acc_out, float coeff, int N, int H, int W, int C, int sign, float alpha,
                                     void* stream) {
  const long total = (long)N * H * W * C;
  if (total <= 0) return INDM_OK;
  if (total >= (1L << 31)) return INDM_E_SHAPE;
  if ((acc_in == nullptr) != (acc_out == nullptr)) return INDM_E_ARG;
  hipLaunchKernelGGL(tap_gather_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, P, bias, y, N, H,
                     W, C, sign, alpha, mulp, acc_in, acc_out, coeff);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}
INDM_API int indm_tap_spread_f32(const float* y, float* P, int N, int H, int W, int C, int sign, float alpha,
                                 void* stream) {
  const long total = (long)N * H * W * C * 9;
  if (total <= 0) return INDM_OK;
  if (total >= (1L << 31)) return INDM_E_SHAPE;
  hipLaunchKernelGGL(tap_spread_kernel, dim3(ew_grid(total / C, 256)), dim3(256), 0, (hipStream_t)stream, y, P, N, H, W, C,
                     sign, alpha);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}

// ---- multi-tensor Lop normalisation: all LopConv weights of the flow in two launches (fwd) / two launches (bwd) ------
// `table` is a DEVICE array of `count` rows of 6 int64: {w, aux, out, red, R, Co}
//   forward : aux unused, out = wn, red = l1  (written here)
//   backward: aux = dwn,  out = dw, red = scratch dot[Co]; l1 is read from table2 = {l1}
namespace {
struct LopRow {
  const float* w;
  const float* aux;
  float* out;
  float* red;
  long R, Co;
};

// column reduction  red[c] = sum_r f(w[r,c], aux[r,c]);  one block per (64- or fewer-column group, tensor); rows are spread
// over 1024/cols lanes with 4 independent accumulators and combined through LDS in a fixed order (deterministic)
template <int MODE>
__global__ __launch_bounds__(1024) void lop_multi_reduce_kernel(const LopRow* __restrict__ table) {
  __shared__ float sh[1024];
  const LopRow t = table[blockIdx.y];
  const int Co = (int)t.Co, R = (int)t.R;
  int cpb = 64;
  while (cpb > 1 && (cpb >> 1) >= Co) cpb >>= 1;  // columns per block: 64, or the next power of two >= Co
  const int c0 = blockIdx.x * cpb;
  if (c0 >= Co) return;
  const int lanes = 1024 / cpb;
  const int tc = threadIdx.x % cpb, l = threadIdx.x / cpb;
  const int c = c0 + tc;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < Co) {
    auto f = [&](int r) {
      const long i = (long)r * Co + c;
      return MODE == 0 ? fabsf(t.w[i]) : t.w[i] * t.aux[i];
    };
    int r = l;
    for (; r + 3 * lanes < R; r += 4 * lanes) {
      s0 += f(r);
      s1 += f(r + lanes);
      s2 += f(r + 2 * lanes);
      s3 += f(r + 3 * lanes);
    }
    for (; r < R; r += lanes) s0 += f(r);
  }
  sh[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  // fixed-order tree over the row lanes (lanes is a power of two)
  for (int step = lanes >> 1; step >= 1; step >>= 1) {
    if (l < step) sh[l * cpb + tc] += sh[(l + step) * cpb + tc];
    __syncthreads();
  }
  if (l == 0 && c < Co) t.red[c] = sh[tc];
}

template <int MODE>
__global__ __launch_bounds__(256) void lop_multi_apply_kernel(const LopRow* __restrict__ table,
                                                              const float* const* __restrict__ l1tab, float coeff) {
  const LopRow t = table[blockIdx.y];
  const unsigned n = (unsigned)(t.R * t.Co), Co = (unsigned)t.Co;
  const float* l1 = MODE == 0 ? t.red : l1tab[blockIdx.y];
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const unsigned c = i % Co;
    const float s = l1[c] / coeff;
    const float fct = fmaxf(1.f, s);
    if (MODE == 0) {
      t.out[i] = t.w[i] / fct;
    } else {
      float g = t.aux[i] / fct;
      if (s > 1.f) {
        const float wv = t.w[i];
        const float sg = wv > 0.f ? 1.f : (wv < 0.f ? -1.f : 0.f);
        g -= sg * t.red[c] / (fct * fct * coeff);
      }
      t.out[i] = g;
    }
  }
}
}  // namespace

// max_cols = largest Co, max_elems = largest R*Co over the table (grid sizing only)
INDM_API int indm_lop_normalize_multi_fwd_f32(const void* table, int count, int max_cols, long max_elems, float coeff,
                                              void* stream) {
  if (count <= 0) return INDM_OK;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(lop_multi_reduce_kernel<0>, dim3(cdiv(max_cols, 64) > 0 ? cdiv(max_cols, 64) : 1, count),
                     dim3(1024), 0, st, (const LopRow*)table);
  INDM_CHECK_LAUNCH();
  hipLaunchKernelGGL(lop_multi_apply_kernel<0>, dim3(ew_grid(max_elems, 256) > 64 ? 64 : ew_grid(max_elems, 256), count),
                     dim3(256), 0, st, (const LopRow*)table, (const float* const*)nullptr, coeff);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}

INDM_API int indm_lop_normalize_multi_bwd_f32(const void* table, const void* l1_table, int count, int max_cols,
                                              long max_elems, float coeff, void* stream) {
  if (count <= 0) return INDM_OK;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(lop_multi_reduce_kernel<1>, dim3(cdiv(max_cols, 64) > 0 ? cdiv(max_cols, 64) : 1, count),
                     dim3(1024), 0, st, (const LopRow*)table);
  INDM_CHECK_LAUNCH();
  hipLaunchKernelGGL(lop_multi_apply_kernel<1>, dim3(ew_grid(max_elems, 256) > 64 ? 64 : ew_grid(max_elems, 256), count),
                     dim3(256), 0, st, (const LopRow*)table, (const float* const*)l1_table, coeff);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}

// ---- multi-tensor copy: dst_i[0..n_i) = src_i[0..n_i) for a DEVICE table of count rows {src, dst, n} (int64) -----------
// Gathers every parameter gradient autograd produced into the flat gradient buffer in ONE launch (optim.py), instead of
// one accumulate kernel per parameter.
namespace {
struct CopyRow {
  const float* src;
  float* dst;
  long n;
};
__global__ __launch_bounds__(256) void multi_copy_kernel(const CopyRow* __restrict__ table) {
  const CopyRow t = table[blockIdx.y];
  const long n4 = (((uintptr_t)t.src | (uintptr_t)t.dst) & 15) == 0 ? t.n >> 2 : 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
    ((float4*)t.dst)[i] = ((const float4*)t.src)[i];
  for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < t.n; i += (long)gridDim.x * blockDim.x)
    t.dst[i] = t.src[i];
}
}  // namespace

INDM_API int indm_multi_copy_f32(const void* table, int count, long max_n, void* stream) {
  if (count <= 0) return INDM_OK;
  int gx = ew_grid((max_n + 3) / 4, 256);
  if (gx > 32) gx = 32;
  hipLaunchKernelGGL(multi_copy_kernel, dim3(gx, count), dim3(256), 0, (hipStream_t)stream, (const CopyRow*)table);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}

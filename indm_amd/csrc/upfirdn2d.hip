// up-FIR-down resampling and fused bias+activation for gfx950.
// Behavioural contract (NOT a translation) of the reference's native ops:
//   upfirdn2d      : op/upfirdn2d.cpp:12-23, op/upfirdn2d_kernel.cu:107-207, CPU oracle op/upfirdn2d.py:159-200
//   fused_bias_act : op/fused_bias_act.cpp:11-21, op/fused_bias_act_kernel.cu:18-49
//
// MI355X design: the tensor is addressed through generic strides so the SAME kernel serves the
// reference's [major=N*C, H, W, minor=1] view (planar NCHW) and this library's native NHWC
// activations, where the channel dimension rides on the lanes: a wavefront reads 64 consecutive
// channels (256 B) of one pixel per load, the FIR taps live in LDS/SGPRs, and each work-item
// walks the <= 4x4 taps of one output pixel-channel.  Zero insertion (up) and decimation (down)
// are index arithmetic; nothing is materialised.
#include "indm_common.h"

namespace {

struct UfdParams {
  const float* x;
  const float* k;  // [kh, kw]
  float* y;
  int N, C, in_h, in_w, out_h, out_w;
  int up_x, up_y, down_x, down_y, pad_x0, pad_y0, kh, kw;
  long xs_n, xs_c, xs_h, xs_w;  // input strides (elements)
  long ys_n, ys_c, ys_h, ys_w;  // output strides
  int c_fast;                   // 1: channel is the fastest output index (NHWC)
};

__device__ __forceinline__ int floor_div(int a, int b) {
  int q = a / b;
  if ((a % b != 0) && ((a < 0) != (b < 0))) --q;
  return q;
}

__global__ __launch_bounds__(256) void upfirdn2d_kernel(UfdParams p) {
  __shared__ float sk[64];
  const int ktaps = p.kh * p.kw;
  for (int i = threadIdx.x; i < ktaps; i += blockDim.x) {
    // flipped kernel: out = sum_{ky,kx} xpad[oy*down+ky, ox*down+kx] * k[kh-1-ky, kw-1-kx]
    const int ky = i / p.kw, kx = i - ky * p.kw;
    sk[i] = p.k[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)];
  }
  __syncthreads();
  const long total = (long)p.N * p.C * p.out_h * p.out_w;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int n, c, oy, ox;
    long r = i;
    if (p.c_fast) {
      c = (int)(r % p.C); r /= p.C;
      ox = (int)(r % p.out_w); r /= p.out_w;
      oy = (int)(r % p.out_h);
      n = (int)(r / p.out_h);
    } else {
      ox = (int)(r % p.out_w); r /= p.out_w;
      oy = (int)(r % p.out_h); r /= p.out_h;
      c = (int)(r % p.C);
      n = (int)(r / p.C);
    }
    const float* xb = p.x + n * p.xs_n + c * p.xs_c;
    // position in the zero-inserted, padded signal: u = o*down + k - pad ; source = u/up when divisible
    const int uy0 = oy * p.down_y - p.pad_y0, ux0 = ox * p.down_x - p.pad_x0;
    float acc = 0.f;
    for (int ky = 0; ky < p.kh; ++ky) {
      const int uy = uy0 + ky;
      if (uy < 0) continue;
      const int iy = uy / p.up_y;
      if (iy * p.up_y != uy || iy >= p.in_h) continue;
      for (int kx = 0; kx < p.kw; ++kx) {
        const int ux = ux0 + kx;
        if (ux < 0) continue;
        const int ix = ux / p.up_x;
        if (ix * p.up_x != ux || ix >= p.in_w) continue;
        acc += xb[iy * p.xs_h + ix * p.xs_w] * sk[ky * p.kw + kx];
      }
    }
    p.y[n * p.ys_n + c * p.ys_c + oy * p.ys_h + ox * p.ys_w] = acc;
  }
}

__global__ __launch_bounds__(256) void fused_bias_act_kernel(const float* __restrict__ x, const float* __restrict__ b,
                                                              const float* __restrict__ ref, float* __restrict__ y,
                                                              long n, int act, int grad, float alpha, float scale,
                                                              long step_b, int size_b, int use_bias, int use_ref) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = x[i];
    if (use_bias) v += b[(i / step_b) % size_b];
    const float r = use_ref ? ref[i] : 0.f;
    float o;
    // act*10 + grad, same dispatch table as the reference kernel
    switch (act * 10 + grad) {
      default:
      case 10: o = v; break;
      case 11: o = v; break;
      case 12: o = 0.f; break;
      case 30: o = v > 0.f ? v : v * alpha; break;
      case 31: o = r > 0.f ? v : v * alpha; break;
      case 32: o = 0.f; break;
    }
    y[i] = o * scale;
  }
}

}  // namespace

// Generic-stride upfirdn2d.  out_h = (in_h*up_y + pad_y0 + pad_y1 - kh)/down_y + 1 (floor), same for w;
// pads may be negative (crop).  layout: 0 = planar [N,C,H,W] (reference view), 1 = NHWC.
INDM_API int indm_upfirdn2d_f32(const float* x, const float* kernel, float* y, int N, int C, int in_h, int in_w,
                                int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1,
                                int pad_y0, int pad_y1, int layout, void* stream) {
  if (N < 0 || C <= 0 || in_h <= 0 || in_w <= 0 || kh <= 0 || kw <= 0 || kh * kw > 64) return INDM_E_SHAPE;
  if (up_x < 1 || up_y < 1 || down_x < 1 || down_y < 1) return INDM_E_ARG;
  UfdParams p;
  p.x = x; p.k = kernel; p.y = y; p.N = N; p.C = C; p.in_h = in_h; p.in_w = in_w;
  p.out_h = (in_h * up_y + pad_y0 + pad_y1 - kh) / down_y + 1;
  p.out_w = (in_w * up_x + pad_x0 + pad_x1 - kw) / down_x + 1;
  if (p.out_h <= 0 || p.out_w <= 0) return INDM_E_SHAPE;
  p.up_x = up_x; p.up_y = up_y; p.down_x = down_x; p.down_y = down_y;
  p.pad_x0 = pad_x0; p.pad_y0 = pad_y0; p.kh = kh; p.kw = kw;
  if (layout == 1) {
    p.xs_n = (long)in_h * in_w * C; p.xs_h = (long)in_w * C; p.xs_w = C; p.xs_c = 1;
    p.ys_n = (long)p.out_h * p.out_w * C; p.ys_h = (long)p.out_w * C; p.ys_w = C; p.ys_c = 1;
    p.c_fast = 1;
  } else {
    p.xs_n = (long)C * in_h * in_w; p.xs_c = (long)in_h * in_w; p.xs_h = in_w; p.xs_w = 1;
    p.ys_n = (long)C * p.out_h * p.out_w; p.ys_c = (long)p.out_h * p.out_w; p.ys_h = p.out_w; p.ys_w = 1;
    p.c_fast = 0;
  }
  const long total = (long)N * C * p.out_h * p.out_w;
  if (total == 0) return INDM_OK;
  hipLaunchKernelGGL(upfirdn2d_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, p);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}

// y = act(x + bias[(i/step_b) % size_b]) * scale ; act 1 = linear, 3 = leaky-ReLU(alpha);
// grad 0 = forward, 1 = first derivative gated by the sign of `ref`, 2 = zero (second derivative).
INDM_API int indm_fused_bias_act_f32(const float* x, const float* bias, const float* ref, float* y, long n,
                                     long step_b, int size_b, int act, int grad, float alpha, float scale,
                                     void* stream) {
  if (n <= 0) return INDM_OK;
  if (!(act == 1 || act == 3) || grad < 0 || grad > 2) return INDM_E_ARG;
  const int use_bias = bias != nullptr && size_b > 0, use_ref = ref != nullptr;
  hipLaunchKernelGGL(fused_bias_act_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, x, bias, ref, y,
                     n, act, grad, alpha, scale, step_b > 0 ? step_b : 1, size_b > 0 ? size_b : 1, use_bias, use_ref);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}

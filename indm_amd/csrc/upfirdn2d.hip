// up-FIR-down resampling and fused bias+activation for gfx950.
// Behavioural contract (NOT a translation) of the reference's native ops:
//   upfirdn2d      : op/upfirdn2d.cpp:12-23, op/upfirdn2d_kernel.cu:107-207, CPU oracle op/upfirdn2d.py:159-200
//   fused_bias_act : op/fused_bias_act.cpp:11-21, op/fused_bias_act_kernel.cu:18-49
//
// MI355X design, two kernels:
//   * upfirdn2d_tile_kernel (NHWC, C % 4 == 0: every call of the NCSN++ on this path): a workgroup owns an 8 x 8 output
//     tile x 32 channels of one image and stages the input patch it needs (<= 19 x 19 pixels x 32 channels, 46 KB) in LDS
//     with coalesced float4 loads along C -- every input element leaves HBM/L2 once per tile instead of once per tap;
//     each work-item then walks the <= 4 x 4 taps of one output pixel for four channels out of LDS.  Zero insertion (up)
//     and decimation (down) are index arithmetic; nothing is materialised.  (Round 1's one-thread-per-element kernel
//     ran at 0.35-0.84 TB/s on the VE config's shapes: four 64-bit div/mod per element and scalar tap loads.)
//   * upfirdn2d_kernel: generic strides, one work-item per output element -- the reference's [major=N*C, H, W, minor=1]
//     view (planar NCHW), odd channel counts (the 3-channel image pyramid) and kernels larger than 8 x 8.
#include "indm_common.h"

namespace {

// I/O element types: float, or bf16 (the reference kernel's half instantiation, op/upfirdn2d_kernel.cu:311, in the storage type
// of this library's bf16 mode): taps and accumulation are fp32 either way, the LDS patch holds fp32
struct bf16_t { unsigned short v; };
__device__ __forceinline__ float ld1(const float* q) { return *q; }
__device__ __forceinline__ float ld1(const bf16_t* q) { return __uint_as_float((unsigned)q->v << 16); }
__device__ __forceinline__ void st1(float* q, float v) { *q = v; }
__device__ __forceinline__ void st1(bf16_t* q, float v) {
  const __bf16 h = (__bf16)v;
  q->v = *(const unsigned short*)&h;
}
__device__ __forceinline__ float4 ld4(const float* q) { return *(const float4*)q; }
__device__ __forceinline__ float4 ld4(const bf16_t* q) {
  const uint2 w = *(const uint2*)q;
  return make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16),
                     __uint_as_float(w.y & 0xffff0000u));
}
__device__ __forceinline__ void st4(float* q, const float4& v) { *(float4*)q = v; }
__device__ __forceinline__ void st4(bf16_t* q, const float4& v) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t a = {v.x, v.y}, b = {v.z, v.w};
  const bf16x2_t ha = __builtin_convertvector(a, bf16x2_t), hb = __builtin_convertvector(b, bf16x2_t);
  *(uint2*)q = make_uint2(*(const unsigned*)&ha, *(const unsigned*)&hb);
}

template <class T>
struct UfdParams {
  const T* x;
  const float* k;  // [kh, kw]
  T* y;
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

template <class T>
__global__ __launch_bounds__(256) void upfirdn2d_kernel(UfdParams<T> p) {
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
    const T* xb = p.x + n * p.xs_n + c * p.xs_c;
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
        acc += ld1(xb + iy * p.xs_h + ix * p.xs_w) * sk[ky * p.kw + kx];
      }
    }
    st1(p.y + n * p.ys_n + c * p.ys_c + oy * p.ys_h + ox * p.ys_w, acc);
  }
}

// ---- LDS-tiled NHWC kernel ----
constexpr int UC = 32;  // channels per tile
__device__ __forceinline__ int pos_mod(int a, int b) {
  const int r = a % b;
  return r < 0 ? r + b : r;
}
// ut_sh = log2(output tile edge): 16 x 16 outputs when the patch stays small (up-sampling / same-size filtering), 8 x 8
// for decimation (the patch is 19 x 19 pixels there)
template <class T>
__global__ __launch_bounds__(256) void upfirdn2d_tile_kernel(UfdParams<T> p, int tiles_x, int tiles_y, int cchunks, int IH, int IW,
                                                             int ut_sh) {
  extern __shared__ __attribute__((aligned(16))) float patch[];  // [IH][IW][UC]
  __shared__ float sk[64];
  const int ktaps = p.kh * p.kw;
  for (int i = threadIdx.x; i < ktaps; i += blockDim.x) {
    const int ky = i / p.kw, kx = i - ky * p.kw;
    sk[i] = p.k[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)];
  }
  const int UT = 1 << ut_sh;
  int b = blockIdx.x;
  const int cc = b % cchunks; b /= cchunks;
  const int tx = b % tiles_x; b /= tiles_x;
  const int ty = b % tiles_y;
  const int n = b / tiles_y;
  const int oy0 = ty * UT, ox0 = tx * UT, c0 = cc * UC;
  const int nc4 = min(UC, p.C - c0) >> 2;                           // float4 groups of this chunk (C % 4 == 0)
  const int iy0 = floor_div(oy0 * p.down_y - p.pad_y0, p.up_y), ix0 = floor_div(ox0 * p.down_x - p.pad_x0, p.up_x);
  const T* xb = p.x + (long)n * p.xs_n + c0;
  for (int i = threadIdx.x; i < IH * IW * (UC / 4); i += 256) {
    const int c4 = i & (UC / 4 - 1), pix = i >> 3;
    const int r = pix / IW, c = pix - r * IW;
    const int iy = iy0 + r, ix = ix0 + c;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < nc4 && iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w)
      v = ld4(xb + (long)iy * p.xs_h + (long)ix * p.xs_w + 4 * c4);
    *(float4*)(patch + (long)pix * UC + 4 * c4) = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < UT * UT * (UC / 4); i += 256) {
    const int c4 = i & (UC / 4 - 1), pix = i >> 3;
    const int oy = oy0 + (pix >> ut_sh), ox = ox0 + (pix & (UT - 1));
    if (c4 >= nc4 || oy >= p.out_h || ox >= p.out_w) continue;
    // taps that land on a real (non-inserted) sample: ky = ky0, ky0 + up, ... with source row iy, iy + 1, ...
    const int uy0 = oy * p.down_y - p.pad_y0, ux0 = ox * p.down_x - p.pad_x0;
    const int ky0 = pos_mod(-uy0, p.up_y), kx0 = pos_mod(-ux0, p.up_x);
    const int iyb = floor_div(uy0 + ky0, p.up_y), ixb = floor_div(ux0 + kx0, p.up_x);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int iy = iyb;
    for (int ky = ky0; ky < p.kh; ky += p.up_y, ++iy) {
      if (iy < 0 || iy >= p.in_h) continue;
      const float* prow = patch + ((iy - iy0) * IW - ix0) * UC + 4 * c4;
      int ix = ixb;
      for (int kx = kx0; kx < p.kw; kx += p.up_x, ++ix) {
        if (ix < 0 || ix >= p.in_w) continue;
        const float w = sk[ky * p.kw + kx];
        const float4 v = *(const float4*)(prow + ix * UC);
        acc.x += v.x * w; acc.y += v.y * w; acc.z += v.z * w; acc.w += v.w * w;
      }
    }
    st4(p.y + (long)n * p.ys_n + (long)oy * p.ys_h + (long)ox * p.ys_w + c0 + 4 * c4, acc);
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
template <class T>
static int upfirdn2d_impl(const T* x, const float* kernel, T* y, int N, int C, int in_h, int in_w,
                          int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1,
                          int pad_y0, int pad_y1, int layout, void* stream) {
  if (N < 0 || C <= 0 || in_h <= 0 || in_w <= 0 || kh <= 0 || kw <= 0 || kh * kw > 64) return INDM_E_SHAPE;
  if (up_x < 1 || up_y < 1 || down_x < 1 || down_y < 1) return INDM_E_ARG;
  UfdParams<T> p;
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
  if (layout == 1 && (C & 3) == 0 && (((uintptr_t)x | (uintptr_t)y) & (4 * sizeof(T) - 1)) == 0) {
    // input rows / columns an output tile touches (the accumulation order over taps equals the generic kernel's)
    int ut_sh = 4;
    int IH = (((1 << ut_sh) - 1) * down_y + kh - 1) / up_y + 2, IW = (((1 << ut_sh) - 1) * down_x + kw - 1) / up_x + 2;
    if ((size_t)IH * IW * UC * sizeof(float) > 48 * 1024) {
      ut_sh = 3;
      IH = (((1 << ut_sh) - 1) * down_y + kh - 1) / up_y + 2;
      IW = (((1 << ut_sh) - 1) * down_x + kw - 1) / up_x + 2;
    }
    const size_t bytes = (size_t)IH * IW * UC * sizeof(float);
    if (bytes <= 64 * 1024) {
      const int UT = 1 << ut_sh;
      const int tiles_x = cdiv(p.out_w, UT), tiles_y = cdiv(p.out_h, UT), cchunks = cdiv(C, UC);
      const long blocks = (long)N * tiles_y * tiles_x * cchunks;
      if (blocks < (1L << 31)) {
        hipLaunchKernelGGL(upfirdn2d_tile_kernel<T>, dim3((unsigned)blocks), dim3(256), bytes, (hipStream_t)stream, p, tiles_x,
                           tiles_y, cchunks, IH, IW, ut_sh);
        INDM_CHECK_LAUNCH();
        return INDM_OK;
      }
    }
  }
  hipLaunchKernelGGL(upfirdn2d_kernel<T>, dim3(ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, p);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}

INDM_API int indm_upfirdn2d_f32(const float* x, const float* kernel, float* y, int N, int C, int in_h, int in_w,
                                int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1,
                                int pad_y0, int pad_y1, int layout, void* stream) {
  return upfirdn2d_impl<float>(x, kernel, y, N, C, in_h, in_w, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0,
                               pad_y1, layout, stream);
}
// the same op on bf16 tensors (x, y: bf16; taps fp32; fp32 accumulation, one rounding at the store)
INDM_API int indm_upfirdn2d_bf16(const void* x, const float* kernel, void* y, int N, int C, int in_h, int in_w,
                                 int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1,
                                 int pad_y0, int pad_y1, int layout, void* stream) {
  return upfirdn2d_impl<bf16_t>((const bf16_t*)x, kernel, (bf16_t*)y, N, C, in_h, in_w, kh, kw, up_x, up_y, down_x, down_y,
                                pad_x0, pad_x1, pad_y0, pad_y1, layout, stream);
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

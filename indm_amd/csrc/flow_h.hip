// bf16-STORAGE variants of the residual flow's HBM-bound kernels (BASELINE configs #3-#5: "bf16 compute / fp32 accum").
//
// With bf16 MFMA inputs every contraction of the flow is an HBM streamer; what it streams are the eight
// 512-channel tensors of an iResBlock (p1, a1, d1, p2, a2, d2 and the chain's t3, t2) and their backward counterparts.
// In storage mode those tensors live in HBM as bf16 (RNE of the fp32 value the producing kernel holds in registers):
// half the bytes for the streamers, the thin kernels and the elementwise passes.  Everything skinny (3-/12-channel
// images, per-image vectors), every parameter, gradient, optimiser state, loss and the accumulators stay fp32.
//   indm_sincos_act_fwd_bf16 : Sin activation + derivative, bf16 in -> bf16 a, d
//   indm_ires_bwd_act_bf16   : one Sin level of the hand-written iResBlock backward (elementwise.hip), bf16 in / out
//   indm_cast_f32_bf16 / indm_cast_bf16_f32 : layout-preserving casts at the boundary of the storage region
#include "indm_common.h"

namespace {

constexpr float TWO_PI = 6.283185307179586f;
constexpr float INV_TWO_PI = 0.15915494309189535f;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <typename F>
__global__ __launch_bounds__(256) void ewh_kernel(long n, F f) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) f(i);
}
template <typename F>
int ewh_launch(long n, hipStream_t st, F f) {
  if (n <= 0) return INDM_OK;
  hipLaunchKernelGGL(ewh_kernel<F>, dim3(ew_grid(n, 256)), dim3(256), 0, st, n, f);
  return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
}

constexpr int IRES_CH = 64;
// rows [R, C] bf16, C even; a thread owns two adjacent columns (4-byte accesses: 256 B per wave-instruction)
template <bool HAS_DT, bool HAS_SEG>
__global__ __launch_bounds__(256) void ires_bwd_act_h_kernel(const bf16x2* __restrict__ da, const bf16x2* __restrict__ du,
                                                             const bf16x2* __restrict__ d, const bf16x2* __restrict__ a,
                                                             const bf16x2* __restrict__ t, bf16x2* __restrict__ dp,
                                                             bf16x2* __restrict__ dt, float* __restrict__ part, long R,
                                                             int C2) {
  const float K = TWO_PI * TWO_PI;
  for (long chunk = blockIdx.x; chunk * IRES_CH < R; chunk += gridDim.x) {
    const long r0 = chunk * IRES_CH;
    const int nr = (int)min((long)IRES_CH, R - r0);
    for (int c = threadIdx.x; c < C2; c += 256) {
      float s0 = 0.f, s1 = 0.f;
      long i = r0 * C2 + c;
#pragma unroll 4
      for (int r = 0; r < nr; ++r, i += C2) {
        const bf16x2 vu = du[i], vd = d[i], va = da[i], aa = a[i], tt = t[i];
        const float g0 = (float)va[0] * (float)vd[0] - K * (float)aa[0] * ((float)vu[0] * (float)tt[0]);
        const float g1 = (float)va[1] * (float)vd[1] - K * (float)aa[1] * ((float)vu[1] * (float)tt[1]);
        bf16x2 o;
        o[0] = (__bf16)g0; o[1] = (__bf16)g1;
        dp[i] = o;
        if (HAS_DT) {
          bf16x2 q;
          q[0] = (__bf16)((float)vu[0] * (float)vd[0]);
          q[1] = (__bf16)((float)vu[1] * (float)vd[1]);
          dt[i] = q;
        }
        if (HAS_SEG) { s0 += g0; s1 += g1; }
      }
      if (HAS_SEG) {
        part[chunk * (2 * C2) + 2 * c] = s0;
        part[chunk * (2 * C2) + 2 * c + 1] = s1;
      }
    }
  }
}
__global__ __launch_bounds__(256) void ires_seg_fold_h_kernel(const float* __restrict__ part, float* __restrict__ seg,
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

// a = sin(2 pi p)/(2 pi), d = cos(2 pi p); p, a, d: bf16 [n], n % 4 == 0, 8-byte aligned
INDM_API int indm_sincos_act_fwd_bf16(const void* p, void* a, void* d, long n, void* stream) {
  if (n & 3) return INDM_E_SHAPE;
  const bf16x4* pp = (const bf16x4*)p;
  bf16x4* pa = (bf16x4*)a;
  bf16x4* pd = (bf16x4*)d;
  return ewh_launch(n >> 2, (hipStream_t)stream, [=] __device__(long i) {
    const bf16x4 v = pp[i];
    bf16x4 oa, od;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float s, c;
      sincosf(TWO_PI * (float)v[k], &s, &c);
      oa[k] = (__bf16)(s * INV_TWO_PI);
      od[k] = (__bf16)c;
    }
    pa[i] = oa;
    pd[i] = od;
  });
}

// bf16 version of indm_ires_bwd_act_f32: dp = da*d - 4 pi^2 a (du*t) [, dt = du*d] in bf16; seg sums (fp32) of the fp32 dp
INDM_API int indm_ires_bwd_act_bf16(const void* da, const void* du, const void* d, const void* a, const void* t, void* dp,
                                    void* dt_out, float* seg_out, long R, int C, long rows_per_seg, float* ws,
                                    void* stream) {
  if (R < 0 || C <= 0 || (C & 1)) return INDM_E_SHAPE;
  if (R == 0) return INDM_OK;
  if (seg_out && (rows_per_seg <= 0 || rows_per_seg % IRES_CH || R % rows_per_seg || !ws)) return INDM_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const long chunks = (R + IRES_CH - 1) / IRES_CH;
  dim3 grid((unsigned)(chunks > 4096 ? 4096 : chunks)), block(256);
  const bf16x2 *pda = (const bf16x2*)da, *pdu = (const bf16x2*)du, *pd = (const bf16x2*)d, *pa = (const bf16x2*)a,
               *pt = (const bf16x2*)t;
  bf16x2 *pdp = (bf16x2*)dp, *pdt = (bf16x2*)dt_out;
  const int C2 = C / 2;
  if (dt_out && seg_out) hipLaunchKernelGGL((ires_bwd_act_h_kernel<true, true>), grid, block, 0, st, pda, pdu, pd, pa, pt, pdp, pdt, ws, R, C2);
  else if (dt_out) hipLaunchKernelGGL((ires_bwd_act_h_kernel<true, false>), grid, block, 0, st, pda, pdu, pd, pa, pt, pdp, pdt, ws, R, C2);
  else if (seg_out) hipLaunchKernelGGL((ires_bwd_act_h_kernel<false, true>), grid, block, 0, st, pda, pdu, pd, pa, pt, pdp, pdt, ws, R, C2);
  else hipLaunchKernelGGL((ires_bwd_act_h_kernel<false, false>), grid, block, 0, st, pda, pdu, pd, pa, pt, pdp, pdt, ws, R, C2);
  INDM_CHECK_LAUNCH();
  if (seg_out) {
    const long nseg = R / rows_per_seg;
    hipLaunchKernelGGL(ires_seg_fold_h_kernel, dim3(ew_grid(nseg * C, 256)), block, 0, st, ws, seg_out, nseg,
                       (int)(rows_per_seg / IRES_CH), C);
    INDM_CHECK_LAUNCH();
  }
  return INDM_OK;
}

INDM_API int indm_cast_f32_bf16(const float* x, void* y, long n, void* stream) {
  __bf16* py = (__bf16*)y;
  return ewh_launch(n, (hipStream_t)stream, [=] __device__(long i) { py[i] = (__bf16)x[i]; });
}
INDM_API int indm_cast_bf16_f32(const void* x, float* y, long n, void* stream) {
  const __bf16* px = (const __bf16*)x;
  return ewh_launch(n, (hipStream_t)stream, [=] __device__(long i) { y[i] = (float)px[i]; });
}

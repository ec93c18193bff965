// Block-level entry points of the predictor-corrector sampler and of the likelihood's divergence estimator
// (indm_pc_step_{em,rd,langevin}, indm_hutch_div).  One call = one reference update_fn AFTER the score
// network ran (the network stays a composition of the op-level entry points):
//   EulerMaruyamaPredictor.update_fn   sampling.py:191-197  with RSDE.sde        sde_lib.py:96-103
//   ReverseDiffusionPredictor.update_fn sampling.py:205-210 with RSDE.discretize sde_lib.py:105-118
//   LangevinCorrector.update_fn (one iteration) sampling.py:279-290
//   get_div_fn                          likelihood.py:27-38
// The noise of an update is either an explicit tensor z (parity replays of recorded noise) or drawn INSIDE the update
// kernel from the device-resident Philox state {seed, offset} -- the same counters -> values map as
// indm_philox_randn_f32 (sde_optim.hip), so "draw, then update" and the fused call give identical bits; the noise
// tensor (12 KB per image) is then never written or read.  HBM-bound: x, net_out read once, x_new / x_mean written once.
#include "indm_common.h"

#include <math.h>

namespace {

struct Noise {
  const float* z;                    // explicit noise, or
  const unsigned long long* state;   // device {seed, offset}
};

// the four normals of counter group i (elements 4 i .. 4 i + 3): identical to philox_randn_kernel
__device__ __forceinline__ float4 normal4(unsigned long long seed, unsigned long long off, long i) {
  const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
  const unsigned long long c = off + (unsigned long long)i;
  const uint4 r = philox4x32(key, make_uint4((uint32_t)c, (uint32_t)(c >> 32), 0x6e646d69u, 0u));
  const float r0 = sqrtf(-2.f * logf(u32_to_unit(r.x))), r1 = sqrtf(-2.f * logf(u32_to_unit(r.z)));
  float s0, c0, s1, c1;
  sincosf(6.28318530717958647692f * u32_to_unit(r.y), &s0, &c0);
  sincosf(6.28318530717958647692f * u32_to_unit(r.w), &s1, &c1);
  return make_float4(r0 * c0, r0 * s0, r1 * c1, r1 * s1);
}

__device__ __forceinline__ float4 noise4(const Noise& nz, unsigned long long seed, unsigned long long off, long i) {
  return nz.z ? *(const float4*)(nz.z + (i << 2)) : normal4(seed, off, i);
}

// MODE 0: Euler-Maruyama, 1: reverse diffusion, 2: Langevin (step / noise coefficient per image in `p0` / `p1`)
//   x_mean = a x + b net_out,  x_new = x_mean + c z        with per-image a, b, c from the reference's quantities:
//   EM : a = 1 + f_lin dt         b = -(g^2) k dt [* 1/2]   c = g sqrt(-dt)  [0 for the probability flow]
//   RD : a = 1 - f_lin            b = (G^2) k [* 1/2]       c = G            [0]
//   LV : a = 1                    b = step k                c = sqrt(2 step)       (grad = k net_out)
template <int MODE>
__global__ __launch_bounds__(256) void pc_update_kernel(const float* __restrict__ x, const float* __restrict__ out,
                                                         const float* __restrict__ p0, const float* __restrict__ p1,
                                                         const float* __restrict__ k, float dt, float sdt, int pf, Noise nz,
                                                         float* __restrict__ x_new, float* __restrict__ x_mean, long B,
                                                         long D4) {
  unsigned long long seed = 0, off = 0;
  if (!nz.z && nz.state) { seed = nz.state[0]; off = nz.state[1]; }
  const bool noisy = (nz.z || nz.state) && !(pf && MODE != 2);
  const long n4 = B * D4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const long b = i / D4;
    float a, bb, c;
    if (MODE == 0) {
      const float g = p1[b];
      a = 1.f + p0[b] * dt;
      bb = -(g * g) * k[b] * dt;
      if (pf) bb *= 0.5f;
      c = g * sdt;
    } else if (MODE == 1) {
      const float G = p1[b];
      a = 1.f - p0[b];
      bb = (G * G) * k[b];
      if (pf) bb *= 0.5f;
      c = G;
    } else {
      a = 1.f;
      bb = p0[b];
      c = p1[b];
    }
    const float4 xv = *(const float4*)(x + (i << 2)), ov = *(const float4*)(out + (i << 2));
    float xi[4] = {xv.x, xv.y, xv.z, xv.w}, oi[4] = {ov.x, ov.y, ov.z, ov.w}, m[4], r[4];
    if (MODE == 2) {
      const float kk = k[b];
#pragma unroll
      for (int j = 0; j < 4; ++j) oi[j] *= kk;   // grad = k * net_out, rounded as the composed path rounds it
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      m[j] = a * xi[j];
      m[j] += bb * oi[j];
      r[j] = m[j];
    }
    if (noisy) {
      const float4 zv = noise4(nz, seed, off, i);
      r[0] = m[0] + c * zv.x; r[1] = m[1] + c * zv.y; r[2] = m[2] + c * zv.z; r[3] = m[3] + c * zv.w;
    }
    if (x_mean) *(float4*)(x_mean + (i << 2)) = make_float4(m[0], m[1], m[2], m[3]);
    if (x_new) *(float4*)(x_new + (i << 2)) = make_float4(r[0], r[1], r[2], r[3]);
  }
}

// Langevin: per (image, chunk) partial sums of |k out|^2 and |z|^2 -> part[(b * chunks + c) * 2 + {0, 1}]
__global__ __launch_bounds__(256) void langevin_part_kernel(const float* __restrict__ out, const float* __restrict__ k,
                                                             Noise nz, float* __restrict__ part, long D4, int chunks) {
  __shared__ float sh[16];
  unsigned long long seed = 0, off = 0;
  if (!nz.z) { seed = nz.state[0]; off = nz.state[1]; }
  const int b = blockIdx.y, c = blockIdx.x;
  const long per = (D4 + chunks - 1) / chunks;
  const long i0 = (long)c * per, i1 = min(D4, i0 + per);
  const float kk = k[b];
  float gs = 0.f, ns = 0.f;
  for (long i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
    const long gi = (long)b * D4 + i;
    const float4 ov = *(const float4*)(out + (gi << 2));
    const float4 zv = noise4(nz, seed, off, gi);
    const float g0 = ov.x * kk, g1 = ov.y * kk, g2 = ov.z * kk, g3 = ov.w * kk;
    gs += (g0 * g0 + g1 * g1) + (g2 * g2 + g3 * g3);
    ns += (zv.x * zv.x + zv.y * zv.y) + (zv.z * zv.z + zv.w * zv.w);
  }
  gs = block_sum(gs, sh);
  ns = block_sum(ns, sh);
  if (threadIdx.x == 0) {
    part[((long)b * chunks + c) * 2] = gs;
    part[((long)b * chunks + c) * 2 + 1] = ns;
  }
}
// batch-mean norms -> per-image step size and noise coefficient (sampling.py:284-290): one workgroup, fixed order
__global__ __launch_bounds__(256) void langevin_coef_kernel(const float* __restrict__ part, const float* __restrict__ alpha,
                                                             float* __restrict__ step, float* __restrict__ ncoef, int B,
                                                             int chunks, float snr) {
  __shared__ float sh[16];
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    float g = 0.f, n = 0.f;
    for (int c = 0; c < chunks; ++c) {
      g += part[((long)i * chunks + c) * 2];
      n += part[((long)i * chunks + c) * 2 + 1];
    }
    a += sqrtf(g);
    b += sqrtf(n);
  }
  a = block_sum(a, sh);
  b = block_sum(b, sh);
  const float r = snr * (b / B) / (a / B);
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    const float s = r * r * 2.f * alpha[i];
    step[i] = s;
    ncoef[i] = sqrtf(s * 2.f);
  }
}

__global__ void philox_skip_kernel(unsigned long long* state, unsigned long long by) { state[1] += by; }

// div[b] partials: sum over a chunk of g * e
__global__ __launch_bounds__(256) void hutch_part_kernel(const float* __restrict__ g, const float* __restrict__ e,
                                                          float* __restrict__ part, long D, int chunks) {
  __shared__ float sh[16];
  const int b = blockIdx.y, c = blockIdx.x;
  const long per = ((D + chunks - 1) / chunks + 3) & ~3L;
  const long i0 = (long)c * per, i1 = min(D, i0 + per);
  const float* gp = g + (long)b * D;
  const float* ep = e + (long)b * D;
  float s = 0.f;
  for (long i = i0 + threadIdx.x; i < i1; i += blockDim.x) s += gp[i] * ep[i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) part[(long)b * chunks + c] = s;
}
__global__ void hutch_final_kernel(const float* __restrict__ part, float* __restrict__ div, int B, int chunks) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float s = 0.f;
  for (int c = 0; c < chunks; ++c) s += part[(long)b * chunks + c];
  div[b] = s;
}

int chunks_of(long D) {
  int c = (int)((D + 8191) / 8192);
  return c < 1 ? 1 : c;
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

template <int MODE>
int predictor_step(const float* x, const float* net_out, const float* p0, const float* p1, const float* k, double dt, int pf,
                   const float* z, void* philox, float* x_new, float* x_mean, long B, long D, void* stream) {
  if (B < 0 || D <= 0 || (D & 3)) return INDM_E_SHAPE;
  if (B == 0) return INDM_OK;
  if (!x || !net_out || !p0 || !p1 || !k || (!x_new && !x_mean)) return INDM_E_ARG;
  if (z && philox) return INDM_E_ARG;                      // one noise source
  if (!pf && x_new && !z && !philox) return INDM_E_ARG;    // a stochastic update needs one
  if (!aligned16(x) || !aligned16(net_out) || !aligned16(z) || !aligned16(x_new) || !aligned16(x_mean)) return INDM_E_ARG;
  if (MODE == 0 && !(dt < 0.0)) return INDM_E_ARG;         // reverse time: dt = -1 / N
  hipStream_t st = (hipStream_t)stream;
  const Noise nz = {z, (const unsigned long long*)philox};
  const long n4 = B * (D >> 2);
  hipLaunchKernelGGL(pc_update_kernel<MODE>, dim3(ew_grid(n4, 256)), dim3(256), 0, st, x, net_out, p0, p1, k, (float)dt,
                     MODE == 0 ? (float)sqrt(-dt) : 0.f, pf, nz, x_new, x_mean, B, D >> 2);
  INDM_CHECK_LAUNCH();
  if (philox) {   // the draw consumes ceil(B D / 4) counters, exactly as indm_philox_randn_f32 would have -- also for a
                  // probability-flow step, whose reference update draws z and multiplies it by G = 0 (sampling.py:207-209)
    hipLaunchKernelGGL(philox_skip_kernel, dim3(1), dim3(1), 0, st, (unsigned long long*)philox, (unsigned long long)n4);
    INDM_CHECK_LAUNCH();
  }
  return INDM_OK;
}

}  // namespace

// x, net_out, z, x_new, x_mean: [B, D] fp32 (any image layout: the update is elementwise), D % 4 == 0, 16-byte aligned.
// f_lin, g, k: [B] device vectors -- the forward SDE's drift f(x, t) = f_lin x (VP: -beta_t / 2, VE: 0), its diffusion
// g(t), and the factor that turns the raw network output into the score (models/utils.py:167-177: -1 / std for
// ddpm_score, else 1).  dt < 0 (-1 / N).  probability_flow: drift halves the score term and the noise is dropped.
// Noise: z (explicit) XOR philox (device {seed, offset}, advanced by B D / 4 -- also by a probability-flow step, which
// does not use the values); both may be NULL only for the probability flow.
// x_new or x_mean may be NULL.
INDM_API int indm_pc_step_em_f32(const float* x, const float* net_out, const float* f_lin, const float* g, const float* k,
                                 double dt, int probability_flow, const float* z, void* philox, float* x_new,
                                 float* x_mean, long B, long D, void* stream) {
  return predictor_step<0>(x, net_out, f_lin, g, k, dt, probability_flow, z, philox, x_new, x_mean, B, D, stream);
}

// f_lin, G: the DISCRETISED forward step f = f_lin x, G (VP: sqrt(1 - beta_i) - 1, sqrt(beta_i); VE: 0,
// sqrt(sigma_i^2 - sigma_{i-1}^2)): x_mean = x - (f - G^2 score [/ 2]), x_new = x_mean + G z.
INDM_API int indm_pc_step_rd_f32(const float* x, const float* net_out, const float* f_lin, const float* G, const float* k,
                                 int probability_flow, const float* z, void* philox, float* x_new, float* x_mean, long B,
                                 long D, void* stream) {
  return predictor_step<1>(x, net_out, f_lin, G, k, -1.0, probability_flow, z, philox, x_new, x_mean, B, D, stream);
}

INDM_API long indm_pc_step_ws_floats(long B, long D) { return 2 * B * chunks_of(D) + 2 * B; }

// One Langevin iteration: grad = k net_out; step = 2 alpha (snr mean_b|z| / mean_b|grad|)^2 (BATCH-mean norms,
// sampling.py:286-288); x_mean = x + step grad; x_new = x_mean + sqrt(2 step) z.  ws: indm_pc_step_ws_floats(B, D) floats.
INDM_API int indm_pc_step_langevin_f32(const float* x, const float* net_out, const float* k, const float* alpha, float snr,
                                       const float* z, void* philox, float* x_new, float* x_mean, float* ws, long B,
                                       long D, void* stream) {
  if (B < 0 || D <= 0 || (D & 3)) return INDM_E_SHAPE;
  if (B == 0) return INDM_OK;
  if (!x || !net_out || !k || !alpha || !ws || (!x_new && !x_mean) || (!z == !philox)) return INDM_E_ARG;
  if (!aligned16(x) || !aligned16(net_out) || !aligned16(z) || !aligned16(x_new) || !aligned16(x_mean)) return INDM_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const Noise nz = {z, (const unsigned long long*)philox};
  const int chunks = chunks_of(D);
  float* part = ws;
  float* step = ws + 2 * B * chunks;
  float* ncoef = step + B;
  hipLaunchKernelGGL(langevin_part_kernel, dim3(chunks, (unsigned)B), dim3(256), 0, st, net_out, k, nz, part, D >> 2, chunks);
  INDM_CHECK_LAUNCH();
  hipLaunchKernelGGL(langevin_coef_kernel, dim3(1), dim3(256), 0, st, part, alpha, step, ncoef, (int)B, chunks, snr);
  INDM_CHECK_LAUNCH();
  const long n4 = B * (D >> 2);
  hipLaunchKernelGGL(pc_update_kernel<2>, dim3(ew_grid(n4, 256)), dim3(256), 0, st, x, net_out, step, ncoef, k, 0.f, 0.f, 0,
                     nz, x_new, x_mean, B, D >> 2);
  INDM_CHECK_LAUNCH();
  if (philox) {
    hipLaunchKernelGGL(philox_skip_kernel, dim3(1), dim3(1), 0, st, (unsigned long long*)philox, (unsigned long long)n4);
    INDM_CHECK_LAUNCH();
  }
  return INDM_OK;
}

INDM_API long indm_hutch_div_ws_floats(long B, long D) { return B * chunks_of(D); }

// Hutchinson's estimator of the drift's divergence (likelihood.py:27-38): div[b] = sum_i grad_fn_eps[b, i] eps[b, i],
// grad_fn_eps = d(sum(fn(x, t) eps))/dx from the score network's backward pass.  Fixed summation order.
INDM_API int indm_hutch_div_f32(const float* grad_fn_eps, const float* eps, float* div, long B, long D, float* ws,
                                void* stream) {
  if (B < 0 || D <= 0) return INDM_E_SHAPE;
  if (B == 0) return INDM_OK;
  if (!grad_fn_eps || !eps || !div || !ws) return INDM_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int chunks = chunks_of(D);
  hipLaunchKernelGGL(hutch_part_kernel, dim3(chunks, (unsigned)B), dim3(256), 0, st, grad_fn_eps, eps, ws, D, chunks);
  INDM_CHECK_LAUNCH();
  hipLaunchKernelGGL(hutch_final_kernel, dim3(cdiv(B, 256)), dim3(256), 0, st, ws, div, (int)B, chunks);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}

// Per-step fused kernels around the networks (all HBM-bound, fp32):
//   * SDE perturbation  x_t = mean_coef[b]*x + std[b]*eps            (sde_lib.py:153-157, losses.py:103-105)
//   * denoising-score-matching loss + its gradient                    (losses.py:106-118, models/utils.py:175-177)
//   * predictor / corrector affine update  x' = a[b]*x + b[b]*s + c[b]*z  (sampling.py:186-210, 263-292)
//   * Langevin step-size from batch-mean norms                         (sampling.py:284-288)
//   * global grad-norm, clip + AdamW + EMA in one pass over flat buffers (losses.py:30-62, models/ema.py:32-51)
// Per-image scalars (mean_coef, std, g, ...) are [B] device arrays produced by the host-side
// SDE mirror (fp32, same op order as the reference) -- see DESIGN.md "SDE scalars".
#include "indm_common.h"

namespace {

template <typename F>
__global__ __launch_bounds__(256) void ew2_kernel(long n, F f) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) f(i);
}
template <typename F>
int ew2_launch(long n, hipStream_t st, F f) {
  if (n <= 0) return INDM_OK;
  hipLaunchKernelGGL(ew2_kernel<F>, dim3(ew_grid(n, 256)), dim3(256), 0, st, n, f);
  if (hipGetLastError() != hipSuccess) return INDM_E_LAUNCH;
  return INDM_OK;
}

// score from raw network output: kind 0 -> -out/std (VP, ddpm_score); kind 1 -> out (VE)
__device__ __forceinline__ float score_of(float o, float sd, int kind) { return kind == 0 ? -o / sd : o; }

// one block per (image, chunk): partial sums of the per-element DSM term
__global__ __launch_bounds__(256) void dsm_loss_kernel(const float* __restrict__ out, const float* __restrict__ z,
                                                        const float* __restrict__ std, float* __restrict__ part, long D,
                                                        int chunks, int kind, int mode) {
  __shared__ float sh[16];
  const int b = blockIdx.y, c = blockIdx.x;
  const long per = ((D + chunks - 1) / chunks + 3) & ~3L;
  const long i0 = (long)c * per, i1 = min(D, i0 + per);
  const float sd = std[b];
  float s = 0.f;
  for (long i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
    const float sc = score_of(out[(long)b * D + i], sd, kind);
    const float e = mode == 0 ? sc * sd + z[(long)b * D + i] : sc + z[(long)b * D + i] / sd;
    s += e * e;
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) part[(long)b * chunks + c] = s;
}
__global__ void dsm_final_kernel(const float* __restrict__ part, const float* __restrict__ w, float* __restrict__ loss,
                                 int B, int chunks, float red_scale) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float s = 0.f;
  for (int c = 0; c < chunks; ++c) s += part[(long)b * chunks + c];
  loss[b] = w[b] * (s * red_scale);
}

__global__ __launch_bounds__(256) void sqnorm_part_kernel(const float* __restrict__ g, long n, float* __restrict__ part) {
  __shared__ float sh[16];
  float s = 0.f;
  const long n4 = n >> 2;
  const float4* g4 = (const float4*)g;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = g4[i];
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0)
    for (long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) s += g[i] * g[i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void sqnorm_final_kernel(const float* __restrict__ part, int P, float* __restrict__ out) {
  __shared__ float sh[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < P; i += blockDim.x) s += part[i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) out[0] = s;
}

struct AdamArgs {
  float lr, beta1, beta2, eps, wd, bc1, bc2s, clip, ema_omd, gscale;
};

__global__ __launch_bounds__(256) void adamw_ema_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v,
                                                         float* __restrict__ ema, long n, AdamArgs a,
                                                         const float* __restrict__ gnorm_sq) {
  // torch.nn.utils.clip_grad_norm_: coef = clip / (norm + 1e-6), clamped to 1
  float coef = a.gscale;
  if (a.clip >= 0.f && gnorm_sq) {
    const float nrm = sqrtf(gnorm_sq[0]) * a.gscale;
    const float cc = a.clip / (nrm + 1e-6f);
    coef *= cc < 1.f ? cc : 1.f;
  }
  const long n4 = n >> 2;
  float4* p4 = (float4*)p;
  const float4* g4 = (const float4*)g;
  float4* m4 = (float4*)m;
  float4* v4 = (float4*)v;
  float4* e4 = (float4*)ema;
  auto upd = [&](float& pw, float gw, float& mw, float& vw, float* ew) {
    gw *= coef;
    pw *= 1.f - a.lr * a.wd;
    mw = a.beta1 * mw + (1.f - a.beta1) * gw;
    vw = a.beta2 * vw + (1.f - a.beta2) * gw * gw;
    const float denom = sqrtf(vw) / a.bc2s + a.eps;
    pw -= (a.lr / a.bc1) * (mw / denom);
    if (ew) *ew -= a.ema_omd * (*ew - pw);
  };
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 pp = p4[i], mm = m4[i], vv = v4[i];
    const float4 gg = g4[i];
    float4 ee = ema ? e4[i] : make_float4(0, 0, 0, 0);
    upd(pp.x, gg.x, mm.x, vv.x, ema ? &ee.x : nullptr);
    upd(pp.y, gg.y, mm.y, vv.y, ema ? &ee.y : nullptr);
    upd(pp.z, gg.z, mm.z, vv.z, ema ? &ee.z : nullptr);
    upd(pp.w, gg.w, mm.w, vv.w, ema ? &ee.w : nullptr);
    p4[i] = pp; m4[i] = mm; v4[i] = vv;
    if (ema) e4[i] = ee;
  }
  if (blockIdx.x == 0) {
    for (long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) {
      float pp = p[i], mm = m[i], vv = v[i], ee = ema ? ema[i] : 0.f;
      upd(pp, g[i], mm, vv, ema ? &ee : nullptr);
      p[i] = pp; m[i] = mm; v[i] = vv;
      if (ema) ema[i] = ee;
    }
  }
}

// step[b] = (snr * mean_b||noise|| / mean_b||grad||)^2 * 2 * alpha[b]
__global__ __launch_bounds__(256) void langevin_step_kernel(const float* __restrict__ gsq, const float* __restrict__ nsq,
                                                             const float* __restrict__ alpha, float* __restrict__ step,
                                                             float* __restrict__ noise_coef, int B, float snr) {
  __shared__ float sh[16];
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    a += sqrtf(gsq[i]);
    b += sqrtf(nsq[i]);
  }
  a = block_sum(a, sh);
  b = block_sum(b, sh);
  const float gn = a / B, nn = b / B;
  const float r = snr * nn / gn;
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    const float s = r * r * 2.f * alpha[i];
    step[i] = s;
    noise_coef[i] = sqrtf(s * 2.f);
  }
}

// log|det W| and W^-1 of one n x n (n <= 64) matrix: Gauss-Jordan with partial pivoting; the augmented [n][2n] system lives
// in LDS.  Round 5: 512 threads (64 rows x SLD_PARTS = 8 parts) -- thread (row = tid % 64, part = tid / 64) owns 16 columns of
// its row -- instead of one
// wavefront walking all 128 columns of every row per pivot (350 us per call, twice per training step, with the GPU otherwise
// idle).  Wave 0 searches the pivot exactly as before and publishes it through LDS; every element still sees the same
// operations in the same order (a[r][j] /= piv; a[r][j] -= f * a[k][j]): bit-identical results.
constexpr int SLD_PARTS = 8, SLD_COLS = 128 / SLD_PARTS;
__global__ __launch_bounds__(64 * SLD_PARTS) void slogdet_inv_kernel(const float* __restrict__ W, int n, float* __restrict__ logdet,
                                                                     float* __restrict__ inv) {
  __shared__ float a[64][129];
  __shared__ int piv_row;
  const int row = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int n2 = 2 * n;
  const int j0 = part * SLD_COLS, j1 = min(n2, j0 + SLD_COLS);
  if (row < n)
    for (int j = j0; j < j1; ++j) a[row][j] = j < n ? W[row * n + j] : (j - n == row ? 1.f : 0.f);
  __syncthreads();
  float ld = 0.f;
  for (int k = 0; k < n; ++k) {
    if (part == 0) {   // pivot search over rows >= k (one wavefront; column k was last written before the previous barrier)
      float v = (row >= k && row < n) ? fabsf(a[row][k]) : -1.f;
      int idx = row;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
      }
      if (row == 0) piv_row = idx;
    }
    __syncthreads();                       // B1: the pivot row is known; last step's column fix is done
    const int p = piv_row;
    const float piv = a[p][k];             // (= a[k][k] after the swap)
    ld += logf(fabsf(piv));
    __syncthreads();                       // B2: everybody has read the pivot before its column is rewritten
    if (row == 0) {                        // swap rows k and p and scale the new row k: one thread per part, its columns
      for (int j = j0; j < j1; ++j) {
        const float tk = a[k][j], tp = a[p][j];
        if (p != k) a[p][j] = tk;
        a[k][j] = tp / piv;
      }
    }
    __syncthreads();                       // B3
    float f = 0.f;
    if (row < n && row != k) {
      f = a[row][k];
      if (f != 0.f)
        for (int j = j0; j < j1; ++j)
          if (j != k) a[row][j] -= f * a[k][j];
    }
    __syncthreads();                       // B4: every part of a row has read its f
    // column k of the eliminated rows (a[row][k] -= f * a[k][k], a[k][k] == piv / piv): by the part that owns column k, after
    // the barrier, so that no part reads an already-cleared f.  The next pivot search reads column k + 1: no conflict.
    if (f != 0.f && k >= j0 && k < j1) a[row][k] = f - f * a[k][k];
  }
  __syncthreads();
  if (threadIdx.x == 0) logdet[0] = ld;
  if (row < n)
    for (int j = max(j0, n); j < j1; ++j) inv[row * n + (j - n)] = a[row][j];
}

}  // namespace

// out[b,:] = mc[b]*x[b,:] + sd[b]*noise[b,:]   (noise / sd may be null -> pure scaling)
// ---- affine coupling of the 64-d prior flow (NICE1d + Affine: flows/couplings/coupling.py:13-177, transform.py:49-81) ----
// x [B, D] is split into (z, zp): 'continuous' = halves, 'skip' = even / odd features; `up` picks which half conditions
// the other.  params [B, 2F] = (mu | log-scale) come from the MLP on z (three small GEMMs of the caller).
//   scale = sigmoid(ls + 2) + 1e-3
//   forward : zp' = scale*zp + mu,            logdet[b] =  sum_f log scale          (transform.py:57-66)
//   inverse : zp' = (zp - mu)/(scale + 1e-12), logdet[b] = -sum_f log scale          (transform.py:68-76)
// One wavefront per row: split, transform, recombination and the log-det row sum in one launch (round 1: ~10 ATen
// elementwise kernels per coupling and direction).
namespace {
__device__ __forceinline__ void coupling_idx(int f, int F, int split, int up, int& iz, int& izp) {
  if (split == 0) {  // continuous: z1 = x[:F], z2 = x[F:]
    izp = up ? F + f : f;
    iz = up ? f : F + f;
  } else {           // skip: z1 = even, z2 = odd
    izp = up ? 2 * f + 1 : 2 * f;
    iz = up ? 2 * f : 2 * f + 1;
  }
}
__global__ __launch_bounds__(64) void coupling_fwd_kernel(const float* __restrict__ x, const float* __restrict__ prm,
                                                          float* __restrict__ y, float* __restrict__ logdet, int B, int F,
                                                          int split, int up, int inverse) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float ld = 0.f;
  for (int f = lane; f < F; f += 64) {
    int iz, izp;
    coupling_idx(f, F, split, up, iz, izp);
    const float mu = prm[(long)b * 2 * F + f], ls = prm[(long)b * 2 * F + F + f];
    const float sc = 1.f / (1.f + expf(-(ls + 2.f))) + 1e-3f;
    const float zp = x[(long)b * 2 * F + izp];
    y[(long)b * 2 * F + izp] = inverse ? (zp - mu) / (sc + 1e-12f) : sc * zp + mu;
    y[(long)b * 2 * F + iz] = x[(long)b * 2 * F + iz];
    ld += logf(sc);
  }
  ld = wave_sum(ld);
  if (lane == 0) logdet[b] = inverse ? -ld : ld;
}
__global__ __launch_bounds__(64) void coupling_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ dld,
                                                          const float* __restrict__ x, const float* __restrict__ prm,
                                                          float* __restrict__ dx, float* __restrict__ dprm, int B, int F,
                                                          int split, int up, int inverse) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float gl = dld ? dld[b] : 0.f;
  for (int f = lane; f < F; f += 64) {
    int iz, izp;
    coupling_idx(f, F, split, up, iz, izp);
    const float mu = prm[(long)b * 2 * F + f], ls = prm[(long)b * 2 * F + F + f];
    const float sg = 1.f / (1.f + expf(-(ls + 2.f)));
    const float sc = sg + 1e-3f;
    const float zp = x[(long)b * 2 * F + izp];
    const float g = dy[(long)b * 2 * F + izp];
    float dzp, dmu, dsc;
    if (!inverse) {
      dzp = g * sc;
      dmu = g;
      dsc = g * zp + gl / sc;
    } else {
      const float q = 1.f / (sc + 1e-12f);
      dzp = g * q;
      dmu = -g * q;
      dsc = -g * (zp - mu) * q * q - gl / sc;
    }
    dx[(long)b * 2 * F + izp] = dzp;
    dx[(long)b * 2 * F + iz] = dy[(long)b * 2 * F + iz];
    dprm[(long)b * 2 * F + f] = dmu;
    dprm[(long)b * 2 * F + F + f] = dsc * sg * (1.f - sg);
  }
}
}  // namespace

// x, y: [B, 2F]; prm: [B, 2F] = (mu | ls); logdet: [B].  split 0 continuous / 1 skip; up 0/1; inverse 0/1.
INDM_API int indm_coupling_fwd_f32(const float* x, const float* prm, float* y, float* logdet, int B, int F, int split,
                                   int up, int inverse, void* stream) {
  if (B < 0 || F <= 0) return INDM_E_SHAPE;
  if (B == 0) return INDM_OK;
  hipLaunchKernelGGL(coupling_fwd_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, x, prm, y, logdet, B, F, split, up,
                     inverse);
  return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
}
// gradient of (y, logdet) w.r.t. (x, prm): dy [B, 2F], dlogdet [B] (may be NULL) -> dx [B, 2F] (the conditioning half
// passes dy through; the MLP's contribution arrives through dprm), dprm [B, 2F].
INDM_API int indm_coupling_bwd_f32(const float* dy, const float* dlogdet, const float* x, const float* prm, float* dx,
                                   float* dprm, int B, int F, int split, int up, int inverse, void* stream) {
  if (B < 0 || F <= 0) return INDM_E_SHAPE;
  if (B == 0) return INDM_OK;
  hipLaunchKernelGGL(coupling_bwd_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, dy, dlogdet, x, prm, dx, dprm, B, F,
                     split, up, inverse);
  return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
}

// Standard-normal noise from a counter-based generator (Philox4x32-10 + Box-Muller): out[4i..4i+3] come from counter
// (offset + i) under key `seed`.  `state` = {seed, offset} lives in DEVICE memory and the offset is advanced by the
// launch itself (a second, one-thread kernel in stream order), so a captured hipGraph of a sampler step draws fresh
// noise on every replay.  Replaces torch.randn on the step (the reference draws from the global torch generator:
// losses.py:103, sampling.py:196,209,283; iresblock.py:137).
namespace {
__global__ __launch_bounds__(256) void philox_randn_kernel(float* __restrict__ out, long n,
                                                           const unsigned long long* __restrict__ state) {
  const unsigned long long seed = state[0], off = state[1];
  const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
  const long groups = (n + 3) >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < groups; i += (long)gridDim.x * blockDim.x) {
    const unsigned long long c = off + (unsigned long long)i;
    const uint4 r = philox4x32(key, make_uint4((uint32_t)c, (uint32_t)(c >> 32), 0x6e646d69u, 0u));
    const float r0 = sqrtf(-2.f * logf(u32_to_unit(r.x))), r1 = sqrtf(-2.f * logf(u32_to_unit(r.z)));
    float s0, c0, s1, c1;
    sincosf(6.28318530717958647692f * u32_to_unit(r.y), &s0, &c0);
    sincosf(6.28318530717958647692f * u32_to_unit(r.w), &s1, &c1);
    const float v[4] = {r0 * c0, r0 * s0, r1 * c1, r1 * s1};
    const long b = i << 2;
    if (b + 3 < n) {
      *(float4*)(out + b) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      for (int k = 0; k < 4 && b + k < n; ++k) out[b + k] = v[k];
    }
  }
}
__global__ void philox_advance_kernel(unsigned long long* state, unsigned long long by) { state[1] += by; }
}  // namespace

// out: n floats (16-byte aligned); state: device pointer to {seed, offset} (two uint64), advanced by ceil(n/4).
INDM_API int indm_philox_randn_f32(float* out, long n, void* state, void* stream) {
  if (n < 0 || !state) return INDM_E_ARG;
  if (n == 0) return INDM_OK;
  if ((uintptr_t)out & 15) return INDM_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const long groups = (n + 3) >> 2;
  hipLaunchKernelGGL(philox_randn_kernel, dim3(ew_grid(groups, 256)), dim3(256), 0, st, out, n,
                     (const unsigned long long*)state);
  hipLaunchKernelGGL(philox_advance_kernel, dim3(1), dim3(1), 0, st, (unsigned long long*)state,
                     (unsigned long long)groups);
  return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
}

INDM_API int indm_sde_perturb_f32(const float* x, const float* noise, const float* mc, const float* sd, float* out,
                                  long B, long D, void* stream) {
  return ew2_launch(B * D, (hipStream_t)stream, [=] __device__(long i) {
    const long b = i / D;
    float v = mc[b] * x[i];
    if (noise) v += sd[b] * noise[i];
    out[i] = v;
  });
}

INDM_API long indm_dsm_ws_floats(long B, long D) {
  int chunks = (int)((D + 8191) / 8192);
  if (chunks < 1) chunks = 1;
  return B * chunks;
}
// loss[b] = w[b] * red( e^2 ),  e = score*std + z (mode 0)  |  score + z/std (mode 1)
// score = -out/std (kind 0) | out (kind 1);  red = mean (reduce_mean=1) or sum over D
INDM_API int indm_dsm_loss_fwd_f32(const float* out, const float* z, const float* std, const float* w, float* loss,
                                   long B, long D, int kind, int mode, int reduce_mean, float* ws, void* stream) {
  if (B <= 0) return INDM_OK;
  hipStream_t st = (hipStream_t)stream;
  int chunks = (int)((D + 8191) / 8192);
  if (chunks < 1) chunks = 1;
  hipLaunchKernelGGL(dsm_loss_kernel, dim3(chunks, (unsigned)B), dim3(256), 0, st, out, z, std, ws, D, chunks, kind, mode);
  INDM_CHECK_LAUNCH();
  hipLaunchKernelGGL(dsm_final_kernel, dim3(cdiv(B, 256)), dim3(256), 0, st, ws, w, loss, (int)B, chunks,
                     reduce_mean ? 1.f / (float)D : 1.f);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}
// dout[b,i] = gl[b]*w[b]*red_scale * 2*e * de/dout
INDM_API int indm_dsm_loss_bwd_f32(const float* gl, const float* out, const float* z, const float* std, const float* w,
                                   float* dout, long B, long D, int kind, int mode, int reduce_mean, void* stream) {
  const float rs = reduce_mean ? 1.f / (float)D : 1.f;
  return ew2_launch(B * D, (hipStream_t)stream, [=] __device__(long i) {
    const long b = i / D;
    const float sd = std[b];
    const float sc = score_of(out[i], sd, kind);
    const float e = mode == 0 ? sc * sd + z[i] : sc + z[i] / sd;
    const float dsc = kind == 0 ? -1.f / sd : 1.f;  // dscore/dout
    const float de = mode == 0 ? dsc * sd : dsc;
    dout[i] = gl[b] * w[b] * rs * 2.f * e * de;
  });
}

// x_mean = a[b]*x + b[b]*s ;  x_new = x_mean + c[b]*z      (s, z, x_mean may be null)
INDM_API int indm_affine_update_f32(const float* x, const float* s, const float* z, const float* a, const float* b,
                                    const float* c, float* x_new, float* x_mean, long B, long D, void* stream) {
  return ew2_launch(B * D, (hipStream_t)stream, [=] __device__(long i) {
    const long r = i / D;
    float m = a[r] * x[i];
    if (s) m += b[r] * s[i];
    if (x_mean) x_mean[i] = m;
    if (x_new) x_new[i] = z ? m + c[r] * z[i] : m;
  });
}

INDM_API int indm_langevin_step_f32(const float* grad_sq, const float* noise_sq, const float* alpha, float* step,
                                    float* noise_coef, int B, float snr, void* stream) {
  if (B <= 0) return INDM_OK;
  hipLaunchKernelGGL(langevin_step_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, grad_sq, noise_sq, alpha, step,
                     noise_coef, B, snr);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}

#define INDM_SQNORM_BLOCKS 1024
INDM_API long indm_sqnorm_ws_floats(void) { return INDM_SQNORM_BLOCKS; }
// out[0] = sum g^2   (device scalar; no host sync)
INDM_API int indm_sqnorm_f32(const float* g, long n, float* out, float* ws, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (((uintptr_t)g & 15) != 0) return INDM_E_ARG;
  int blocks = ew_grid(n >> 2, 256);
  if (blocks > INDM_SQNORM_BLOCKS) blocks = INDM_SQNORM_BLOCKS;
  hipLaunchKernelGGL(sqnorm_part_kernel, dim3(blocks), dim3(256), 0, st, g, n, ws);
  INDM_CHECK_LAUNCH();
  hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(256), 0, st, ws, blocks, out);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}

// Fused  [grad *= gscale] -> clip_grad_norm_(clip) -> AdamW(step t) -> EMA  over flat fp32 buffers.
// gnorm_sq: device scalar holding sum(g^2) of the UNSCALED grads (from indm_sqnorm_f32); clip < 0 disables.
// ema may be null.  ema_one_minus_decay = 1-d for the already warmed-up decay d: s -= (1-d)*(s-p_new).
INDM_API int indm_adamw_ema_f32(float* p, const float* g, float* m, float* v, float* ema, long n, float lr,
                                float beta1, float beta2, float eps, float weight_decay, int step, float clip,
                                const float* gnorm_sq, float gscale, float ema_one_minus_decay, void* stream) {
  if (n <= 0) return INDM_OK;
  if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)ema) & 15) != 0) return INDM_E_ARG;
  AdamArgs a;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay;
  a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  a.bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  a.clip = clip; a.ema_omd = ema_one_minus_decay; a.gscale = gscale;
  hipLaunchKernelGGL(adamw_ema_kernel, dim3(ew_grid(n >> 2, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, ema,
                     n, a, gnorm_sq);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}

// s -= omd * (s - p)   (models/ema.py:49-51)
INDM_API int indm_ema_update_f32(float* s, const float* p, long n, float one_minus_decay, void* stream) {
  return ew2_launch(n, (hipStream_t)stream, [=] __device__(long i) { s[i] -= one_minus_decay * (s[i] - p[i]); });
}

// logdet[0] = log|det W|, inv = W^-1 for an n x n matrix, n <= 64 (replaces the reference's host-side
// torch.slogdet(weight.to('cpu')), flows/permutation.py:107-108, and provides d logdet/dW = inv^T)
INDM_API int indm_slogdet_inv_f32(const float* W, int n, float* logdet, float* inv, void* stream) {
  if (n <= 0 || n > 64) return INDM_E_SHAPE;
  hipLaunchKernelGGL(slogdet_inv_kernel, dim3(1), dim3(64 * SLD_PARTS), 0, (hipStream_t)stream, W, n, logdet, inv);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}

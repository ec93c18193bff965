// HBM-bound passes of the score network on bf16-stored activations ("H16", see unet_h16.hip): GroupNorm (+ SiLU, + dropout)
// forward / backward, the bias-gradient column sums, nearest / mean resampling and the small elementwise glue.
// A thread owns 8 consecutive channels of a pixel = one 16-byte access; statistics, sums and the arithmetic are fp32.
// Reference: nn.GroupNorm(min(C/4, 32), C, eps=1e-6) + SiLU + Dropout, models/layerspp.py:232-279, models/layers.py:38-39;
// naive_upsample_2d / naive_downsample_2d, models/up_or_down_sampling.py:59-69.
#include "indm_common.h"
#include <stdlib.h>

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  const f32x2 v = {lo, hi};
  const bf16x2 h = __builtin_convertvector(v, bf16x2);
  return *(const unsigned*)&h;
}
__device__ __forceinline__ void unpack8(const i32x4& w, float (&v)[8]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[2 * k] = bf16_lo((unsigned)w[k]);
    v[2 * k + 1] = bf16_hi((unsigned)w[k]);
  }
}
__device__ __forceinline__ i32x4 pack8(const float (&v)[8]) {
  i32x4 w;
#pragma unroll
  for (int k = 0; k < 4; ++k) w[k] = (int)pack_bf16x2(v[2 * k], v[2 * k + 1]);
  return w;
}
// make the compiler forget what it knows about a raw chunk: the second pass then unpacks it again (8 shifts / ands) instead of
// keeping the first pass's 8 floats per chunk alive across the reduction (which spills)
__device__ __forceinline__ void forget(i32x4& w) {
  asm volatile("" : "+v"(w));
}
__device__ __forceinline__ float silu_f(float u) { return u / (1.f + __expf(-u)); }
__device__ __forceinline__ float silu_grad(float u) {
  const float s = 1.f / (1.f + __expf(-u));
  return s * (1.f + u * (1.f - s));
}
// dropout factors of the 4 elements of flat quad idx4 (the mask of norm_act.hip's dropout_keep: same seed, same mask)
__device__ __forceinline__ void dropout4(uint64_t seed, uint64_t idx4, float p, float (&f)[4]) {
  const uint4 r = philox4x32(make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)),
                             make_uint4((uint32_t)idx4, (uint32_t)(idx4 >> 32), 0u, 0u));
  const float keep = 1.f / (1.f - p);
  f[0] = u32_to_unit(r.x) > p ? keep : 0.f;
  f[1] = u32_to_unit(r.y) > p ? keep : 0.f;
  f[2] = u32_to_unit(r.z) > p ? keep : 0.f;
  f[3] = u32_to_unit(r.w) > p ? keep : 0.f;
}

__device__ __forceinline__ void gn_decode(int b, int GB, int N, int& n, int& gb) {   // group blocks of one image on one XCD
  const int full = (N / 8) * 8 * GB;
  if (b < full) {
    const int xcd = b & 7, t = b >> 3;
    gb = t % GB;
    n = (t / GB) * 8 + xcd;
  } else {
    const int r = b - full;
    n = (N / 8) * 8 + r / GB;
    gb = r % GB;
  }
}

constexpr int GN_MAXW = 16;    // waves per workgroup
constexpr int GN_MAXQ = 8;     // channel quads per slab (CS <= 32 channels)

// Sum `NV` values over the lanes of a wave that own the same slab position q = tid % TPL and leave them in lanes 0 .. TPL-1
// (lane == q).  TPL (lanes per pixel) is a power of two: strided xor butterfly (the lanes of one q are TPL apart).  A 24-channel
// slab (TP = 3 chunks per pixel: the 384-channel maps) runs with TPL = 4 and an idle fourth lane instead of a masked-sum path.
template <int NV>
__device__ __forceinline__ void sum_same_q(float (&v)[NV], int TPL) {
  for (int off = TPL; off < 64; off <<= 1) {
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] += __shfl_xor(v[j], off, 64);
  }
}

// One workgroup per (image, block of GS groups = CS channels, CS % 8 == 0, CS <= 32): the slab [HW][CS] lives in registers as
// packed bf16 (NCH 16-byte chunks per thread), is reduced to the group statistics and written back normalised.
template <int NCH, int MAXT>
__global__ __launch_bounds__(MAXT) void h16_gn_fwd_kernel(const i32x4* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, i32x4* __restrict__ y,
                                                           float* __restrict__ mean, float* __restrict__ rstd, int N, int HW,
                                                           int C, int G, int GS, float eps, int act, float drop_p,
                                                           uint64_t seed, const uint64_t* __restrict__ seed_dev) {
  if (seed_dev) seed += *seed_dev;   // graph-captured launches: the per-call seed lives in device memory (indm_amd/graphed.py)
  __shared__ float shw[GN_MAXW * 4 * 4];    // [wave][q][s1 lo, s2 lo, s1 hi, s2 hi]
  __shared__ float shq[GN_MAXQ * 2];        // [quad][s1, s2]
  __shared__ float shg[GN_MAXQ * 2];        // [group][mean, rstd]
  const int bs = blockDim.x, tid = threadIdx.x;
  int n, gb;
  gn_decode(blockIdx.x, G / GS, N, n, gb);
  const int cpg = C / G, CS = GS * cpg, TP = CS >> 3, C8 = C >> 3;
  const int TPL = TP == 3 ? 4 : TP;         // lanes per pixel
  const int ql = tid % TPL;
  const bool live = ql < TP;
  const int q = live ? ql : 0;
  const int total = HW * TPL;
  const long img8 = (long)n * HW * C8;      // 16-byte index of the image
  const int c8_0 = gb * TP;
  i32x4 raw[NCH];
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int pix = min(tid + k * bs, total - 1) / TPL;
    raw[k] = x[img8 + (long)pix * C8 + c8_0 + q];
  }
  __builtin_amdgcn_sched_barrier(0);
  float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    if (live && tid + k * bs < total) {
      float v[8];
      unpack8(raw[k], v);
      s[0] += (v[0] + v[1]) + (v[2] + v[3]);
      s[1] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
      s[2] += (v[4] + v[5]) + (v[6] + v[7]);
      s[3] += (v[4] * v[4] + v[5] * v[5]) + (v[6] * v[6] + v[7] * v[7]);
    }
  }
  sum_same_q<4>(s, TPL);
  const int wave = tid >> 6, nw = bs >> 6, lane = tid & 63;
  if (lane < TP) {
#pragma unroll
    for (int j = 0; j < 4; ++j) shw[(wave * 4 + lane) * 4 + j] = s[j];
  }
  __syncthreads();
  if (tid < 2 * TP) {   // quad tid = (q = tid >> 1, half = tid & 1): sum over the waves in wave order
    float a = 0.f, b = 0.f;
    for (int w = 0; w < nw; ++w) {
      a += shw[(w * 4 + (tid >> 1)) * 4 + 2 * (tid & 1)];
      b += shw[(w * 4 + (tid >> 1)) * 4 + 2 * (tid & 1) + 1];
    }
    shq[2 * tid] = a;
    shq[2 * tid + 1] = b;
  }
  __syncthreads();
  if (tid < GS) {
    const int qpg = cpg >> 2;
    float a = 0.f, b = 0.f;
    for (int k = tid * qpg; k < (tid + 1) * qpg; ++k) { a += shq[2 * k]; b += shq[2 * k + 1]; }
    const float inv_cnt = 1.f / ((float)HW * (float)cpg);
    const float m = a * inv_cnt;
    float var = b * inv_cnt - m * m;
    var = fmaxf(var, 0.f);
    const float r = rsqrtf(var + eps);
    shg[2 * tid] = m;
    shg[2 * tid + 1] = r;
    mean[n * G + gb * GS + tid] = m;
    rstd[n * G + gb * GS + tid] = r;
  }
  __syncthreads();
  const int glo = (8 * q) / cpg, ghi = (8 * q + 4) / cpg;
  const float mlo = shg[2 * glo], rlo = shg[2 * glo + 1], mhi = shg[2 * ghi], rhi = shg[2 * ghi + 1];
  float gm[8], bt[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    gm[j] = gamma[gb * CS + 8 * q + j];
    bt[j] = beta[gb * CS + 8 * q + j];
  }
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int f = tid + k * bs;
    const int pix = min(f, total - 1) / TPL;
    const long i8 = img8 + (long)pix * C8 + c8_0 + q;
    float v[8], o[8];
    forget(raw[k]);
    unpack8(raw[k], v);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float u = (v[j] - (j < 4 ? mlo : mhi)) * (j < 4 ? rlo : rhi) * gm[j] + bt[j];
      if (act == 1) u = silu_f(u);
      o[j] = u;
    }
    if (drop_p > 0.f) {
      float f0[4], f1[4];
      dropout4(seed, (uint64_t)i8 * 2, drop_p, f0);
      dropout4(seed, (uint64_t)i8 * 2 + 1, drop_p, f1);
#pragma unroll
      for (int j = 0; j < 4; ++j) { o[j] *= f0[j]; o[4 + j] *= f1[j]; }
    }
    if (live && f < total) y[i8] = pack8(o);
    __builtin_amdgcn_sched_barrier(0);   // one chunk at a time (interleaved, the unrolled chunks spill)
  }
}

// Backward of the fused GroupNorm (+ SiLU, + dropout): dx, and the per-image partial sums of dbeta / dgamma
// (cpart[n][0][C], cpart[n][1][C]; h16_gn_cpart_kernel folds them over the images).
template <int NCH, int MAXT>
__global__ __launch_bounds__(MAXT) void h16_gn_bwd_kernel(const i32x4* __restrict__ dy, const i32x4* __restrict__ x,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           i32x4* __restrict__ dx, float* __restrict__ cpart, int N, int HW,
                                                           int C, int G, int GS, int act, float drop_p, uint64_t seed,
                                                           const uint64_t* __restrict__ seed_dev, const i32x4* __restrict__ add,
                                                           float add_scale) {
  if (seed_dev) seed += *seed_dev;   // graph-captured launches: the per-call seed lives in device memory (indm_amd/graphed.py)
  __shared__ float shw[GN_MAXW * 4 * 16];   // [wave][q][8: sum d, 8: sum d * xh]
  __shared__ float chs[2 * 32];             // channel sums of the slab
  __shared__ float gs_[2 * GN_MAXQ];        // group sums
  const int bs = blockDim.x, tid = threadIdx.x;
  int n, gb;
  gn_decode(blockIdx.x, G / GS, N, n, gb);
  const int cpg = C / G, CS = GS * cpg, TP = CS >> 3, C8 = C >> 3;
  const int TPL = TP == 3 ? 4 : TP;
  const int ql = tid % TPL;
  const bool alive = ql < TP;
  const int q = alive ? ql : 0;
  const int total = HW * TPL;
  const long img8 = (long)n * HW * C8;
  const int c8_0 = gb * TP;
  const int glo = (8 * q) / cpg, ghi = (8 * q + 4) / cpg;
  const float mlo = mean[n * G + gb * GS + glo], rlo = rstd[n * G + gb * GS + glo];
  const float mhi = mean[n * G + gb * GS + ghi], rhi = rstd[n * G + gb * GS + ghi];
  float gm[8], bt[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    gm[j] = gamma[gb * CS + 8 * q + j];
    bt[j] = beta[gb * CS + 8 * q + j];
  }
  i32x4 xr[NCH], dr[NCH];
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int pix = min(tid + k * bs, total - 1) / TPL;
    const long i8 = img8 + (long)pix * C8 + c8_0 + q;
    xr[k] = x[i8];
    dr[k] = dy[i8];
  }
  __builtin_amdgcn_sched_barrier(0);
  // dL/du of a chunk (u = the affine output, before SiLU / dropout) is computed twice -- here for the sums, below for dx --
  // from the raw bf16 words, which stay in registers (8 per chunk; the fp32 values would be 16); the dropout mask is kept as bits
  unsigned keep[NCH];
  float a[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) a[j] = 0.f;
  const float kscale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int f = tid + k * bs;
    const int pix = min(f, total - 1) / TPL;
    const long i8 = img8 + (long)pix * C8 + c8_0 + q;
    float xv[8], dv[8];
    unpack8(xr[k], xv);
    unpack8(dr[k], dv);
    unsigned kb = 0xffu;
    if (drop_p > 0.f) {
      float f0[4], f1[4];
      dropout4(seed, (uint64_t)i8 * 2, drop_p, f0);
      dropout4(seed, (uint64_t)i8 * 2 + 1, drop_p, f1);
      kb = 0u;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        kb |= f0[j] != 0.f ? (1u << j) : 0u;
        kb |= f1[j] != 0.f ? (1u << (4 + j)) : 0u;
      }
    }
    if (!alive || f >= total) kb = 0u;
    keep[k] = kb;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float h = (xv[j] - (j < 4 ? mlo : mhi)) * (j < 4 ? rlo : rhi);
      float d = ((kb >> j) & 1u) ? dv[j] * kscale : 0.f;
      if (act == 1) d *= silu_grad(h * gm[j] + bt[j]);
      a[j] += d;
      a[8 + j] += d * h;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  sum_same_q<16>(a, TPL);
  const int wave = tid >> 6, nw = bs >> 6, lane = tid & 63;
  if (lane < TP) {
#pragma unroll
    for (int j = 0; j < 16; ++j) shw[(wave * 4 + lane) * 16 + j] = a[j];
  }
  __syncthreads();
  if (tid < 2 * CS) {   // (which = tid / CS: 0 sum d, 1 sum d * xh; channel c = tid % CS = 8 q' + j)
    const int which = tid / CS, c = tid - which * CS;
    float s = 0.f;
    for (int w = 0; w < nw; ++w) s += shw[(w * 4 + (c >> 3)) * 16 + 8 * which + (c & 7)];
    chs[32 * which + c] = s;
    cpart[((long)n * 2 + which) * C + gb * CS + c] = s;
  }
  __syncthreads();
  if (tid < GS) {
    float s1 = 0.f, s2 = 0.f;
    for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) {
      const float g = gamma[gb * CS + c];
      s1 += chs[c] * g;
      s2 += chs[32 + c] * g;
    }
    gs_[tid] = s1;
    gs_[GN_MAXQ + tid] = s2;
  }
  __syncthreads();
  const float inv_cnt = 1.f / ((float)HW * (float)cpg);
  const float s1lo = gs_[glo] * inv_cnt, s2lo = gs_[GN_MAXQ + glo] * inv_cnt;
  const float s1hi = gs_[ghi] * inv_cnt, s2hi = gs_[GN_MAXQ + ghi] * inv_cnt;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int f = tid + k * bs;
    const int pix = min(f, total - 1) / TPL;
    const long i8 = img8 + (long)pix * C8 + c8_0 + q;
    float xv[8], dv[8], o[8];
    forget(xr[k]);
    forget(dr[k]);
    unpack8(xr[k], xv);
    unpack8(dr[k], dv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float r = j < 4 ? rlo : rhi;
      const float h = (xv[j] - (j < 4 ? mlo : mhi)) * r;
      float d = ((keep[k] >> j) & 1u) ? dv[j] * kscale : 0.f;
      if (act == 1) d *= silu_grad(h * gm[j] + bt[j]);
      o[j] = r * (gm[j] * d - (j < 4 ? s1lo : s1hi) - h * (j < 4 ? s2lo : s2hi));
    }
    if (add) {   // the gradient that reaches x along its OTHER use (the skip connection of the residual block): dx += add_scale * add
      float av[8];
      unpack8(add[i8], av);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += add_scale * av[j];
    }
    if (alive && f < total) dx[i8] = pack8(o);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// dbeta[c] = sum_n cpart[n][0][c], dgamma[c] = sum_n cpart[n][1][c]: one wavefront per channel, fixed order
__global__ __launch_bounds__(256) void h16_gn_cpart_kernel(const float* __restrict__ cpart, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int N, int C) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  float db = 0.f, dg = 0.f;
  for (int n = lane; n < N; n += 64) {
    db += cpart[((long)n * 2) * C + c];
    dg += cpart[((long)n * 2 + 1) * C + c];
  }
  db = wave_sum(db);
  dg = wave_sum(dg);
  if (lane == 0) {
    dbeta[c] = db;
    dgamma[c] = dg;
  }
}

struct GnGeom { int gs, bs, nch; };
// slab = the widest block of whole groups of at most 32 channels whose chunks fit `maxch` per thread of a <= maxbs-thread workgroup
// (forward: 16 chunks x 1024 threads; backward: 4 x 512 -- it keeps x AND dy and spills beyond that)
static GnGeom gn_geom(int HW, int C, int G, int maxch, int maxbs) {
  GnGeom g = {0, 0, 0};
  if (G <= 0 || C % G || (C & 7)) return g;
  const int cpg = C / G;
  if (cpg & 3) return g;
  for (int cs = 32; cs >= 8; cs -= 8) {
    if (cs % cpg || C % cs) continue;
    const int gs = cs / cpg, tp = cs / 8;
    if (G % gs || gs > GN_MAXQ) continue;
    const long total = (long)HW * (tp == 3 ? 4 : tp);   // (24-channel slabs: four lanes per pixel, the fourth idle)
    int bs = 256;
    if (total <= 64) bs = 64;
    else if (total <= 128) bs = 128;
    while ((total + bs - 1) / bs > maxch && bs * 2 <= maxbs) bs *= 2;
    const int need = (int)((total + bs - 1) / bs);
    if (need > maxch) continue;
    g.gs = gs; g.bs = bs;
    g.nch = need <= 1 ? 1 : need <= 2 ? 2 : need <= 4 ? 4 : need <= 8 ? 8 : 16;
    return g;
  }
  return g;
}

// ---- column sums of a bf16 [R][C] matrix over row segments: out[s][c] = alpha * sum_{r < rps} x[(s rps + r) C + c] (fp32) ----
// stage 1: part[p][c] over rows [p rpp, (p + 1) rpp); a thread owns 8 channels, the workgroup's row lanes are folded through LDS
__global__ __launch_bounds__(256) void h16_colpart_kernel(const i32x4* __restrict__ x, float* __restrict__ part, long R, int C,
                                                           long rpp) {
  __shared__ float sh[256 * 8];
  const int C8 = C >> 3;
  const int tp = min(C8, 256);                    // threads per row (C <= 2048)
  const int rows = 256 / tp;                      // rows in flight
  const int tid = threadIdx.x, q = tid % tp, rl = tid / tp;
  const long r0 = (long)blockIdx.x * rpp, r1 = min(R, r0 + rpp);
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (rl < rows && q < C8) {
    for (long r = r0 + rl; r < r1; r += rows) {
      float v[8];
      unpack8(x[r * C8 + q], v);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) sh[tid * 8 + j] = s[j];
  __syncthreads();
  if (tid < tp && q < C8) {
    float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l < rows; ++l)
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] += sh[(l * tp + tid) * 8 + j];
#pragma unroll
    for (int j = 0; j < 8; ++j) part[(long)blockIdx.x * C + 8 * tid + j] = t[j];
  }
}
// stage 2: out[s][c] = alpha * sum of the segment's pps partial rows.  One workgroup per (segment, 64 columns): four row lanes
// each add every fourth partial row (eight loads in flight), LDS folds the four -- a fixed order, deterministic
__global__ __launch_bounds__(256) void h16_colfinal_kernel(const float* __restrict__ part, float* __restrict__ out, long nseg,
                                                            int pps, int C, float alpha) {
  __shared__ float sh[256];
  const int cb = (C + 63) / 64;
  const long s = blockIdx.x / cb;
  const int c = (blockIdx.x % cb) * 64 + (threadIdx.x & 63), ty = threadIdx.x >> 6;
  float a = 0.f;
  if (c < C) {
    const float* q = part + s * pps * (long)C + c;
    int p = ty;
    for (; p + 28 < pps; p += 32) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = q[(long)(p + 4 * j) * C];
#pragma unroll
      for (int j = 0; j < 8; ++j) a += v[j];
    }
    for (; p < pps; p += 4) a += q[(long)p * C];
  }
  sh[threadIdx.x] = a;
  __syncthreads();
  if (ty == 0 && c < C) out[s * C + c] = alpha * ((sh[threadIdx.x] + sh[64 + threadIdx.x]) + (sh[128 + threadIdx.x] + sh[192 + threadIdx.x]));
}
static long seg_parts(long nseg, long rps, int C) {   // partial rows per segment (a power of two dividing rps)
  long S = 1;
  while (S < 256 && nseg * S < 1024 && rps % (2 * S) == 0 && rps / (2 * S) >= 8) S *= 2;
  return S;
}

// ---- elementwise glue (16 bytes per thread) ----
__global__ __launch_bounds__(256) void h16_scale_kernel(const i32x4* __restrict__ x, i32x4* __restrict__ y, long n8, float a) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float v[8];
    unpack8(x[i], v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= a;
    y[i] = pack8(v);
  }
}
// out (bf16) = a * x (fp32) + b * y (bf16)
__global__ __launch_bounds__(256) void h16_axpby_mixed_kernel(const f32x4* __restrict__ x, const i32x4* __restrict__ y,
                                                               i32x4* __restrict__ out, long n8, float a, float b) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const f32x4 x0 = x[2 * i], x1 = x[2 * i + 1];
    float v[8];
    unpack8(y[i], v);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[j] = a * x0[j] + b * v[j];
      v[4 + j] = a * x1[j] + b * v[4 + j];
    }
    out[i] = pack8(v);
  }
}
// nearest x2: y[n, 2h + a, 2w + b, :] = alpha * x[n, h, w, :]; one thread per INPUT chunk writes its four copies
__global__ __launch_bounds__(256) void h16_upsample2x_kernel(const i32x4* __restrict__ x, i32x4* __restrict__ y, long n8, int H,
                                                              int W, int C8, float alpha) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C8);
    long r = i / C8;
    const int w = (int)(r % W);
    r /= W;
    const int h = (int)(r % H);
    const long n = r / H;
    i32x4 o = x[i];
    if (alpha != 1.f) {
      float v[8];
      unpack8(o, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] *= alpha;
      o = pack8(v);
    }
    i32x4* d = y + (((n * 2 * H + 2 * h) * 2 * W) + 2 * w) * C8 + c;
    d[0] = o;
    d[C8] = o;
    d[(long)2 * W * C8] = o;
    d[(long)2 * W * C8 + C8] = o;
  }
}
// 2 x 2 sum scaled by alpha (0.25: mean); one thread per OUTPUT chunk
__global__ __launch_bounds__(256) void h16_downsample2x_kernel(const i32x4* __restrict__ x, i32x4* __restrict__ y, long n8, int H,
                                                                int W, int C8, float alpha) {
  const int OH = H / 2, OW = W / 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C8);
    long r = i / C8;
    const int ow = (int)(r % OW);
    r /= OW;
    const int oh = (int)(r % OH);
    const long n = r / OH;
    const i32x4* b = x + ((n * H + 2 * oh) * W + 2 * ow) * C8 + c;
    float v0[8], v1[8], v2[8], v3[8], o[8];
    unpack8(b[0], v0);
    unpack8(b[C8], v1);
    unpack8(b[(long)W * C8], v2);
    unpack8(b[(long)W * C8 + C8], v3);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = alpha * (v0[j] + v1[j] + v2[j] + v3[j]);
    y[i] = pack8(o);
  }
}

inline bool al16(const void* q) { return ((uintptr_t)q & 15) == 0; }

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// C ABI (include/indm_hip.h)
// ---------------------------------------------------------------------------------------------------------------------

// 1 when indm_h16_groupnorm_fwd / _bwd cover the geometry (C % 8 == 0, (C / G) % 4 == 0, a block of whole groups of <= 32 channels
// whose slab fits the registers of one workgroup), else 0: the caller converts and takes indm_groupnorm_*_f32
INDM_API int indm_h16_groupnorm_ok(int HW, int C, int G) {
  return (gn_geom(HW, C, G, 16, 1024).nch && (gn_geom(HW, C, G, 4, 512).nch || gn_geom(HW, C, G, 4, 1024).nch)) ? 1 : 0;
}
INDM_API long indm_h16_groupnorm_ws_floats(int N, int C) { return 2L * N * C + 16; }

// y = dropout(act(GroupNorm(x)));  x, y bf16 [N, HW, C];  mean, rstd fp32 [N, G] (saved for the backward)
INDM_API int indm_h16_groupnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                    int N, int HW, int C, int G, float eps, int act, float drop_p, unsigned long long seed,
                                    const unsigned long long* seed_dev_, void* stream) {
  const uint64_t* seed_dev = (const uint64_t*)seed_dev_;
  if (N <= 0) return INDM_OK;
  const GnGeom g = gn_geom(HW, C, G, 16, 1024);
  if (!g.nch) return 1;
  if (!al16(x) || !al16(y)) return INDM_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((G / g.gs) * N), block(g.bs);
#define H16_GN_FWD(NC)                                                                                                          \
  do {                                                                                                                          \
    if (g.bs <= 256) hipLaunchKernelGGL((h16_gn_fwd_kernel<NC, 256>), grid, block, 0, st, (const i32x4*)x, gamma, beta, (i32x4*)y, \
                                        mean, rstd, N, HW, C, G, g.gs, eps, act, drop_p, (uint64_t)seed, seed_dev);                        \
    else if (g.bs <= 512) hipLaunchKernelGGL((h16_gn_fwd_kernel<NC, 512>), grid, block, 0, st, (const i32x4*)x, gamma, beta,      \
                                             (i32x4*)y, mean, rstd, N, HW, C, G, g.gs, eps, act, drop_p, (uint64_t)seed, seed_dev);        \
    else hipLaunchKernelGGL((h16_gn_fwd_kernel<NC, 1024>), grid, block, 0, st, (const i32x4*)x, gamma, beta, (i32x4*)y, mean,     \
                            rstd, N, HW, C, G, g.gs, eps, act, drop_p, (uint64_t)seed, seed_dev);                                          \
  } while (0)
  switch (g.nch) {
    case 1: H16_GN_FWD(1); break;
    case 2: H16_GN_FWD(2); break;
    case 4: H16_GN_FWD(4); break;
    case 8: H16_GN_FWD(8); break;
    default: H16_GN_FWD(16); break;
  }
#undef H16_GN_FWD
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}

// dx (bf16), dgamma, dbeta (fp32) of the above; ws >= indm_h16_groupnorm_ws_floats(N, C) floats
INDM_API int indm_h16_groupnorm_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean,
                                    const float* rstd, void* dx, float* dgamma, float* dbeta, int N, int HW, int C, int G,
                                    int act, float drop_p, unsigned long long seed, const unsigned long long* seed_dev_,
                                    const void* add, float add_scale, float* ws, void* stream) {
  const uint64_t* seed_dev = (const uint64_t*)seed_dev_;
  if (N <= 0) return INDM_OK;
  static const int t_ch = getenv("INDM_H16_GNB_CH") ? atoi(getenv("INDM_H16_GNB_CH")) : 4;      // tuning hooks (tools/h16_gn_bench.py)
  static const int t_bs = getenv("INDM_H16_GNB_BS") ? atoi(getenv("INDM_H16_GNB_BS")) : 512;
  GnGeom g = gn_geom(HW, C, G, t_ch, t_bs);
  if (!g.nch) g = gn_geom(HW, C, G, 4, 512);
  if (!g.nch) g = gn_geom(HW, C, G, 4, 1024);   // 64 x 64 maps: 1024 threads x 4 chunks (this variant spills 15 registers)
  if (!g.nch) return 1;
  if (!al16(x) || !al16(dy) || !al16(dx) || !al16(add) || !ws) return INDM_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((G / g.gs) * N), block(g.bs);
#define H16_GN_BWD(NC)                                                                                                          \
  do {                                                                                                                          \
    if (g.bs <= 256) hipLaunchKernelGGL((h16_gn_bwd_kernel<NC, 256>), grid, block, 0, st, (const i32x4*)dy, (const i32x4*)x, gamma, \
                                        beta, mean, rstd, (i32x4*)dx, ws, N, HW, C, G, g.gs, act, drop_p, (uint64_t)seed, seed_dev, (const i32x4*)add, add_scale); \
    else if (g.bs <= 512) hipLaunchKernelGGL((h16_gn_bwd_kernel<NC, 512>), grid, block, 0, st, (const i32x4*)dy, (const i32x4*)x,  \
                                             gamma, beta, mean, rstd, (i32x4*)dx, ws, N, HW, C, G, g.gs, act, drop_p,            \
                                             (uint64_t)seed, seed_dev, (const i32x4*)add, add_scale);                           \
    else hipLaunchKernelGGL((h16_gn_bwd_kernel<NC, 1024>), grid, block, 0, st, (const i32x4*)dy, (const i32x4*)x, gamma, beta,     \
                            mean, rstd, (i32x4*)dx, ws, N, HW, C, G, g.gs, act, drop_p, (uint64_t)seed, seed_dev,                  \
                            (const i32x4*)add, add_scale);                                                                       \
  } while (0)
  switch (g.nch) {
    case 1: H16_GN_BWD(1); break;
    case 2: H16_GN_BWD(2); break;
    case 4: H16_GN_BWD(4); break;
    default: H16_GN_BWD(8); break;
  }
#undef H16_GN_BWD
  INDM_CHECK_LAUNCH();
  hipLaunchKernelGGL(h16_gn_cpart_kernel, dim3(cdiv(C, 4)), dim3(256), 0, st, ws, dgamma, dbeta, N, C);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}

// out[s][c] = alpha * sum_{r < rps} x[(s rps + r) C + c]: per-image (rps = pixels) or whole-tensor (nseg = 1) column sums of a
// bf16 matrix -- the bias / temb-bias gradients of the convolutions (layerspp.py:262-263).  C % 8 == 0, C <= 2048.
INDM_API long indm_h16_segreduce_ws_floats(long nseg, long rps, int C) { return nseg * seg_parts(nseg, rps, C) * C; }
INDM_API int indm_h16_segreduce(const void* x, float* out, long nseg, long rps, int C, float alpha, float* ws, void* stream) {
  if (nseg <= 0 || C <= 0) return INDM_OK;
  if (rps <= 0 || (C & 7) || C > 2048) return INDM_E_SHAPE;
  if (!al16(x) || !ws) return INDM_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const long S = seg_parts(nseg, rps, C);
  hipLaunchKernelGGL(h16_colpart_kernel, dim3((unsigned)(nseg * S)), dim3(256), 0, st, (const i32x4*)x, ws, nseg * rps, C, rps / S);
  INDM_CHECK_LAUNCH();
  hipLaunchKernelGGL(h16_colfinal_kernel, dim3((unsigned)(nseg * ((C + 63) / 64))), dim3(256), 0, st, ws, out, nseg, (int)S, C, alpha);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}

// y = a * x (bf16 -> bf16), n % 8 == 0
INDM_API int indm_h16_scale(const void* x, void* y, long n, float a, void* stream) {
  if (n < 0 || (n & 7)) return INDM_E_SHAPE;
  if (n == 0) return INDM_OK;
  if (!al16(x) || !al16(y)) return INDM_E_ARG;
  hipLaunchKernelGGL(h16_scale_kernel, dim3(ew_grid(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const i32x4*)x, (i32x4*)y, n / 8, a);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}
// out (bf16) = a * x (fp32) + b * y (bf16), n % 8 == 0: an fp32-computed edge (the input pyramid's skinny convolution,
// ncsnpp.py:339-347) joining the bf16 trunk
INDM_API int indm_h16_axpby_mixed(const float* x, const void* y, void* out, long n, float a, float b, void* stream) {
  if (n < 0 || (n & 7)) return INDM_E_SHAPE;
  if (n == 0) return INDM_OK;
  if (!al16(x) || !al16(y) || !al16(out)) return INDM_E_ARG;
  hipLaunchKernelGGL(h16_axpby_mixed_kernel, dim3(ew_grid(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)x,
                     (const i32x4*)y, (i32x4*)out, n / 8, a, b);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}
// naive_upsample_2d / naive_downsample_2d (models/up_or_down_sampling.py:59-69) on bf16 NHWC, C % 8 == 0
INDM_API int indm_h16_upsample2x(const void* x, void* y, int N, int H, int W, int C, float alpha, void* stream) {
  if (N < 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7)) return INDM_E_SHAPE;
  if (N == 0) return INDM_OK;
  if (!al16(x) || !al16(y)) return INDM_E_ARG;
  const long n8 = (long)N * H * W * (C / 8);
  hipLaunchKernelGGL(h16_upsample2x_kernel, dim3(ew_grid(n8, 256)), dim3(256), 0, (hipStream_t)stream, (const i32x4*)x, (i32x4*)y,
                     n8, H, W, C / 8, alpha);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}
INDM_API int indm_h16_downsample2x(const void* x, void* y, int N, int H, int W, int C, float alpha, void* stream) {
  if (N < 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || (H & 1) || (W & 1)) return INDM_E_SHAPE;
  if (N == 0) return INDM_OK;
  if (!al16(x) || !al16(y)) return INDM_E_ARG;
  const long n8 = (long)N * (H / 2) * (W / 2) * (C / 8);
  hipLaunchKernelGGL(h16_downsample2x_kernel, dim3(ew_grid(n8, 256)), dim3(256), 0, (hipStream_t)stream, (const i32x4*)x,
                     (i32x4*)y, n8, H, W, C / 8, alpha);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}

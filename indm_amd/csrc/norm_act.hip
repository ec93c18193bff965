// HBM-bound normalisation / activation kernels (NHWC fp32) for gfx950.
//   * GroupNorm (+ optional SiLU, + optional Philox dropout) forward / backward
//     - reference: nn.GroupNorm(min(C/4,32), C, eps=1e-6) + nn.SiLU + nn.Dropout,
//       models/layerspp.py:232-279, models/layers.py:38-39
//   * row softmax forward / backward (attention logits, models/layerspp.py:95-99)
//   * column reductions (bias grads, BatchNorm statistics)
// Layout: activations are [N, HW, C] with C contiguous, so every wave reads whole 256 B+
// rows (float4 per lane); group statistics use wavefront shuffles + one LDS hop.
#include "indm_common.h"
#include <stdlib.h>

namespace {

constexpr int GN_SPLIT_TARGET = 1024;  // aim for >= this many workgroups

__device__ __forceinline__ float silu_f(float u) { return u / (1.f + __expf(-u)); }
__device__ __forceinline__ float silu_grad(float u) {
  const float s = 1.f / (1.f + __expf(-u));
  return s * (1.f + u * (1.f - s));
}

__device__ __forceinline__ float dropout_keep(uint64_t seed, uint64_t idx4, int lane_in4, float p) {
  // one Philox call yields the 4 uniforms of an aligned float4 group
  uint4 r = philox4x32(make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)),
                       make_uint4((uint32_t)idx4, (uint32_t)(idx4 >> 32), 0u, 0u));
  const uint32_t v = lane_in4 == 0 ? r.x : lane_in4 == 1 ? r.y : lane_in4 == 2 ? r.z : r.w;
  return u32_to_unit(v) > p ? 1.f / (1.f - p) : 0.f;
}

// ---- pass 1: partial sums per (n, split, group) ---------------------------------
// grid (S, N); each block owns HW/S pixels of one image; thread owns channel-quad
// c4 = tid % (C/4) and strides over pixels.
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, float* __restrict__ part,
                                                        int HW, int C, int G, int S) {
  extern __shared__ float sh[];  // [2][pstride][C] per-thread partials, reduced in a fixed order
  const int n = blockIdx.y, s = blockIdx.x;
  const int C4 = C >> 2;
  const int cpg = C / G;
  const int p_per = (HW + S - 1) / S;
  const int p0 = s * p_per, p1 = min(HW, p0 + p_per);
  const float* xn = x + (long)n * HW * C;
  const int c4 = threadIdx.x % C4;
  const int prow = threadIdx.x / C4, pstride = blockDim.x / C4;
  float* sh1 = sh;
  float* sh2 = sh + pstride * C;
  float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  if (prow < pstride) {
    for (int p = p0 + prow; p < p1; p += pstride) {
      const float4 v = *(const float4*)(xn + (long)p * C + c4 * 4);
      s1[0] += v.x; s2[0] += v.x * v.x;
      s1[1] += v.y; s2[1] += v.y * v.y;
      s1[2] += v.z; s2[2] += v.z * v.z;
      s1[3] += v.w; s2[3] += v.w * v.w;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      sh1[prow * C + c4 * 4 + i] = s1[i];
      sh2[prow * C + c4 * 4 + i] = s2[i];
    }
  }
  __syncthreads();
  for (int g = threadIdx.x; g < G; g += blockDim.x) {
    float a = 0.f, b = 0.f;
    for (int r = 0; r < pstride; ++r)
      for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
        a += sh1[r * C + c];
        b += sh2[r * C + c];
      }
    part[((long)(n * S + s)) * 2 * G + g] = a;
    part[((long)(n * S + s)) * 2 * G + G + g] = b;
  }
}

// ---- finalize: mean / rstd per (n, g) ---------------------------------------------
__global__ void gn_finalize_kernel(const float* __restrict__ part, float* __restrict__ mean,
                                   float* __restrict__ rstd, int NG, int G, int S, float inv_cnt, float eps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NG) return;
  const int n = i / G, g = i - n * G;
  float a = 0.f, b = 0.f;
  for (int s = 0; s < S; ++s) {
    a += part[((long)(n * S + s)) * 2 * G + g];
    b += part[((long)(n * S + s)) * 2 * G + G + g];
  }
  const float m = a * inv_cnt;
  float var = b * inv_cnt - m * m;
  var = fmaxf(var, 0.f);
  mean[i] = m;
  rstd[i] = rsqrtf(var + eps);
}

// ---- pass 2: normalise + affine (+SiLU) (+dropout) --------------------------------
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, float* __restrict__ y,
                                                        long total4, int HW, int C, int G, int act, float drop_p,
                                                        uint64_t seed, const uint64_t* __restrict__ seed_dev) {
  if (seed_dev) seed += *seed_dev;   // graph-captured launches: the per-call seed lives in device memory (indm_amd/graphed.py)
  const int C4 = C >> 2;
  const int cpg = C / G;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    const long pix = i / C4;
    const int n = (int)(pix / HW);
    const float4 v = *(const float4*)(x + i * 4);
    const float4 gm = *(const float4*)(gamma + c4 * 4);
    const float4 bt = *(const float4*)(beta + c4 * 4);
    float in[4] = {v.x, v.y, v.z, v.w};
    const float gmv[4] = {gm.x, gm.y, gm.z, gm.w};
    const float btv[4] = {bt.x, bt.y, bt.z, bt.w};
    float out[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int g = (c4 * 4 + k) / cpg;
      const float m = mean[n * G + g], r = rstd[n * G + g];
      float u = (in[k] - m) * r * gmv[k] + btv[k];
      if (act == 1) u = silu_f(u);
      if (drop_p > 0.f) u *= dropout_keep(seed, (uint64_t)i, k, drop_p);
      out[k] = u;
    }
    *(float4*)(y + i * 4) = make_float4(out[0], out[1], out[2], out[3]);
  }
}

// ---- backward pass 1: per (n,split): group sums + per-channel dgamma/dbeta partials ----
__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            float* __restrict__ gpart, float* __restrict__ cpart, int HW,
                                                            int C, int G, int S, int act, float drop_p, uint64_t seed,
                                                            const uint64_t* __restrict__ seed_dev) {
  if (seed_dev) seed += *seed_dev;   // graph-captured launches: the per-call seed lives in device memory (indm_amd/graphed.py)
  extern __shared__ float sh[];  // [2][pstride][C] partials, then [2][C] channel sums
  const int n = blockIdx.y, s = blockIdx.x;
  const int C4 = C >> 2, cpg = C / G;
  const int p_per = (HW + S - 1) / S;
  const int p0 = s * p_per, p1 = min(HW, p0 + p_per);
  const long base = (long)n * HW * C;
  const int c4 = threadIdx.x % C4;
  const int prow = threadIdx.x / C4, pstride = blockDim.x / C4;
  float* sh1 = sh;
  float* sh2 = sh + pstride * C;
  float* ch = sh + 2 * pstride * C;  // [2][C]
  if (prow < pstride) {
    const float4 gm = *(const float4*)(gamma + c4 * 4);
    const float4 bt = *(const float4*)(beta + c4 * 4);
    const float gmv[4] = {gm.x, gm.y, gm.z, gm.w};
    const float btv[4] = {bt.x, bt.y, bt.z, bt.w};
    float mg[4], rg[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int g = (c4 * 4 + k) / cpg;
      mg[k] = mean[n * G + g];
      rg[k] = rstd[n * G + g];
    }
    float a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};  // sum dyu, sum dyu*xhat (per channel)
    for (int p = p0 + prow; p < p1; p += pstride) {
      const long off = base + (long)p * C + c4 * 4;
      const float4 xv = *(const float4*)(x + off);
      const float4 dv = *(const float4*)(dy + off);
      const float xi[4] = {xv.x, xv.y, xv.z, xv.w};
      const float di[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float xh = (xi[k] - mg[k]) * rg[k];
        float d = di[k];
        if (drop_p > 0.f) d *= dropout_keep(seed, (uint64_t)(off >> 2), k, drop_p);
        if (act == 1) d *= silu_grad(xh * gmv[k] + btv[k]);
        a1[k] += d;
        a2[k] += d * xh;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      sh1[prow * C + c4 * 4 + k] = a1[k];
      sh2[prow * C + c4 * 4 + k] = a2[k];
    }
  }
  __syncthreads();
  const long blk = (long)(n * S + s);
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float a = 0.f, b = 0.f;
    for (int r = 0; r < pstride; ++r) {
      a += sh1[r * C + c];
      b += sh2[r * C + c];
    }
    ch[c] = a;
    ch[C + c] = b;
    cpart[blk * 2 * C + c] = a;      // dbeta partial
    cpart[blk * 2 * C + C + c] = b;  // dgamma partial
  }
  __syncthreads();
  for (int g = threadIdx.x; g < G; g += blockDim.x) {
    float a = 0.f, b = 0.f;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
      a += ch[c] * gamma[c];      // sum gamma*dyu
      b += ch[C + c] * gamma[c];  // sum gamma*dyu*xhat
    }
    gpart[blk * 2 * G + g] = a;
    gpart[blk * 2 * G + G + g] = b;
  }
}

// one workgroup per output element: blocks [0, N*G) reduce the group sums over the S splits, blocks
// [N*G, N*G + C) reduce dbeta/dgamma over all N*S partial rows (fixed order -> deterministic)
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const float* __restrict__ gpart, const float* __restrict__ cpart,
                                                               float* __restrict__ gsum, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, int N, int C, int G, int S) {
  __shared__ float sh[16];
  const int b = blockIdx.x;
  if (b < N * G) {
    if (threadIdx.x == 0) {
      const int n = b / G, g = b - n * G;
      float a0 = 0.f, a1 = 0.f;
      for (int s = 0; s < S; ++s) {
        a0 += gpart[((long)(n * S + s)) * 2 * G + g];
        a1 += gpart[((long)(n * S + s)) * 2 * G + G + g];
      }
      gsum[2 * b] = a0;
      gsum[2 * b + 1] = a1;
    }
    return;
  }
  const int c = b - N * G;
  float db = 0.f, dg = 0.f;
  for (int r = threadIdx.x; r < N * S; r += blockDim.x) {
    db += cpart[(long)r * 2 * C + c];
    dg += cpart[(long)r * 2 * C + C + c];
  }
  db = block_sum(db, sh);
  dg = block_sum(dg, sh);
  if (threadIdx.x == 0) {
    dbeta[c] = db;
    dgamma[c] = dg;
  }
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gsum, float* __restrict__ dx,
                                                            long total4, int HW, int C, int G, int act, float drop_p,
                                                            uint64_t seed, float inv_cnt, const uint64_t* __restrict__ seed_dev) {
  if (seed_dev) seed += *seed_dev;   // graph-captured launches: the per-call seed lives in device memory (indm_amd/graphed.py)
  const int C4 = C >> 2, cpg = C / G;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    const long pix = i / C4;
    const int n = (int)(pix / HW);
    const float4 xv = *(const float4*)(x + i * 4);
    const float4 dv = *(const float4*)(dy + i * 4);
    const float4 gm = *(const float4*)(gamma + c4 * 4);
    const float4 bt = *(const float4*)(beta + c4 * 4);
    const float xi[4] = {xv.x, xv.y, xv.z, xv.w};
    const float di[4] = {dv.x, dv.y, dv.z, dv.w};
    const float gmv[4] = {gm.x, gm.y, gm.z, gm.w};
    const float btv[4] = {bt.x, bt.y, bt.z, bt.w};
    float out[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int g = (c4 * 4 + k) / cpg;
      const float m = mean[n * G + g], r = rstd[n * G + g];
      const float xh = (xi[k] - m) * r;
      float d = di[k];
      if (drop_p > 0.f) d *= dropout_keep(seed, (uint64_t)i, k, drop_p);
      if (act == 1) d *= silu_grad(xh * gmv[k] + btv[k]);
      const float s1 = gsum[2 * (n * G + g)] * inv_cnt, s2 = gsum[2 * (n * G + g) + 1] * inv_cnt;
      out[k] = r * (gmv[k] * d - s1 - xh * s2);
    }
    *(float4*)(dx + i * 4) = make_float4(out[0], out[1], out[2], out[3]);
  }
}

// ---- fused single-pass GroupNorm (round 3) ---------------------------------------------------------------------------
// The three-kernel passes above read x twice (stats, apply) and, in the backward, dy and x twice: 12 / 20 bytes per
// element.  When one image's slab of a few adjacent groups -- CS = 12 or 16 channels x HW pixels -- fits the registers
// of one workgroup (<= 16 float4 per thread: every map of the 32 x 32 configurations), ONE workgroup per (image, group
// block) keeps the slab in registers between the reduction and the apply: 8 / 12 bytes per element, one launch
// forward, one launch + a [N, C] column reduce backward.  Threads walk the slab in 16-byte chunks (a pixel's CS
// channels are one 48- or 64-byte segment); blockDim is a multiple of CS4 = CS / 4, so a thread keeps its channel quad.
// Same formulas as the three-kernel path (sum / sum-of-squares statistics; dropout mask from (seed, flat float4 index)).
constexpr int GNF_MAXG = 8;   // groups per workgroup (GS <= 4 in practice)

// 1-D grid, XCD-aware: the hardware deals workgroup b to XCD b % 8; the group blocks of ONE image share 128-byte lines
// (a pixel's C channels are contiguous), so all of them go to the same XCD's L2: b -> (n, gb) with n % 8 == b % 8
__device__ __forceinline__ void gnf_decode(int b, int GB, int N, int& n, int& gb) {
  const int full = (N / 8) * 8 * GB;          // images beyond the last multiple of 8 keep the plain order
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

template <int NF4>
__global__ __launch_bounds__(256) void gn_fused_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd, int N, int HW,
                                                            int C, int G, int GS, float eps, int act, float drop_p,
                                                            uint64_t seed, const uint64_t* __restrict__ seed_dev) {
  if (seed_dev) seed += *seed_dev;   // graph-captured launches: the per-call seed lives in device memory (indm_amd/graphed.py)
  __shared__ float sh[4 * GNF_MAXG * 2];   // [wave][group][sum, sum of squares]
  // every kernel argument is loaded HERE, not where it is first used (hipcc otherwise issues the scalar loads of gamma /
  // beta / y / mean / rstd after the statistics phase; see tools/late_kernarg_scan.py and DESIGN.md section 6)
  INDM_PIN_ARGS9(gamma, beta, y, mean, rstd, eps, act, drop_p, seed);
  const int bs = blockDim.x, tid = threadIdx.x;
  int n, gb;
  gnf_decode(blockIdx.x, G / GS, N, n, gb);
  const int cpg = C / G, CS = GS * cpg, CS4 = CS >> 2, C4 = C >> 2;
  const int q = tid % CS4;                       // this thread's channel quad inside the slab (bs % CS4 == 0)
  const int gl = (q * 4) / cpg;                  // ... and its group inside the block
  const int total = HW * CS4;
  const long img4 = (long)n * HW * C4;           // float4 index of the image
  const int c4_0 = gb * CS4;
  // Round 5: ALL of the thread's chunks are requested before the first is used (clamped addresses, out-of-range chunks
  // zeroed afterwards).  Written as `if (f < total) { v = load; s += v }` every load sat behind a branch with
  // `s_waitcnt vmcnt(0)` -- sixteen memory latencies in sequence per workgroup -- and the stores below were FLAT stores
  // (the argument pin hides the pointer's address space) waited for one by one with `vmcnt(0) lgkmcnt(0)`: 3.8 TB/s on
  // a kernel that only streams its tensor once in and once out.  Same sums in the same order.
  typedef __attribute__((address_space(1))) float gfloat;
  typedef __attribute__((address_space(1))) f32x4 gfloat4;   // (the ext-vector type: HIP's float4 class has no address-space-1 copy)
  gfloat* yg = (gfloat*)y;
  float4 v[NF4];
#pragma unroll
  for (int k = 0; k < NF4; ++k) {
    const int pix = min(tid + k * bs, total - 1) / CS4;
    v[k] = *(const float4*)(x + (img4 + (long)pix * C4 + c4_0 + q) * 4);
  }
  __builtin_amdgcn_sched_barrier(0);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < NF4; ++k) {
    if (tid + k * bs >= total) v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    s1 += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    s2 += (v[k].x * v[k].x + v[k].y * v[k].y) + (v[k].z * v[k].z + v[k].w * v[k].w);
  }
  // statistics of the slab's groups: per-wave butterfly sums (fixed order) -> one LDS word per (wave, group) -> every thread
  // adds the waves' partials of its own group in wave order.  (Round 3 folded 256 per-thread partials serially on GS threads
  // behind a barrier; see DESIGN.md on the reproducibility flake.)  blockDim is a multiple of 64: all lanes take part.
  const int wave = tid >> 6, nw = bs >> 6;
#pragma unroll
  for (int g = 0; g < GNF_MAXG; ++g) {
    if (g < GS) {
      const float a = wave_sum(gl == g ? s1 : 0.f), b = wave_sum(gl == g ? s2 : 0.f);
      if ((tid & 63) == 0) {
        sh[(wave * GNF_MAXG + g) * 2] = a;
        sh[(wave * GNF_MAXG + g) * 2 + 1] = b;
      }
    }
  }
  __syncthreads();
  float m, r;
  {
    float a = 0.f, b = 0.f;
    for (int w = 0; w < nw; ++w) {
      a += sh[(w * GNF_MAXG + gl) * 2];
      b += sh[(w * GNF_MAXG + gl) * 2 + 1];
    }
    const float inv_cnt = 1.f / ((float)HW * (float)cpg);
    m = a * inv_cnt;
    float var = b * inv_cnt - m * m;
    var = fmaxf(var, 0.f);
    r = rsqrtf(var + eps);
  }
  if (tid < GS) {   // thread tid's own group may differ from group tid: fold that one for the saved statistics
    float a = 0.f, b = 0.f;
    for (int w = 0; w < nw; ++w) {
      a += sh[(w * GNF_MAXG + tid) * 2];
      b += sh[(w * GNF_MAXG + tid) * 2 + 1];
    }
    const float inv_cnt = 1.f / ((float)HW * (float)cpg);
    const float mm = a * inv_cnt;
    float var = b * inv_cnt - mm * mm;
    var = fmaxf(var, 0.f);
    ((gfloat*)mean)[n * G + gb * GS + tid] = mm;
    ((gfloat*)rstd)[n * G + gb * GS + tid] = rsqrtf(var + eps);
  }
  const f32x4 gm = *(const gfloat4*)((const gfloat*)gamma + (c4_0 + q) * 4);
  const f32x4 bt = *(const gfloat4*)((const gfloat*)beta + (c4_0 + q) * 4);
  const float gmv[4] = {gm[0], gm[1], gm[2], gm[3]}, btv[4] = {bt[0], bt[1], bt[2], bt[3]};
  // (a workgroup-uniform flag instead of a per-chunk branch for the whole slab: with a branch per chunk hipcc waits for the
  // previous chunk's store -- `vmcnt(0)` at every merge -- before it computes the next)
  const bool whole = total == NF4 * bs;
#pragma unroll
  for (int k = 0; k < NF4; ++k) {
    const int f = tid + k * bs;
    const int pix = min(f, total - 1) / CS4;
    const long i4 = img4 + (long)pix * C4 + c4_0 + q;
    const float in[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
    float out[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float u = (in[j] - m) * r * gmv[j] + btv[j];
      if (act == 1) u = silu_f(u);
      if (drop_p > 0.f) u *= dropout_keep(seed, (uint64_t)i4, j, drop_p);
      out[j] = u;
    }
    if (whole || f < total) *(gfloat4*)(yg + i4 * 4) = f32x4{out[0], out[1], out[2], out[3]};
  }
}

// Round 3's form of the forward kernel, kept ONLY as the stress reproducer of the reproducibility flake (INDM_GN_FUSED=2,
// tools/flake_pattern.py): 256 per-thread partials in LDS, one barrier, GS threads fold them serially while the other waves
// wait at the next barrier.  On a GPU shared by more processes than it has queue slots this kernel occasionally returned
// statistics with part of the partials missing; the butterfly form above has not.
template <int NF4, bool PIN>
__global__ __launch_bounds__(256) void gn_fused_fwd_serial_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd, int N, int HW,
                                                            int C, int G, int GS, float eps, int act, float drop_p,
                                                            uint64_t seed, const uint64_t* __restrict__ seed_dev) {
  if (seed_dev) seed += *seed_dev;   // graph-captured launches: the per-call seed lives in device memory (indm_amd/graphed.py)
  if constexpr (PIN) INDM_PIN_ARGS9(gamma, beta, y, mean, rstd, eps, act, drop_p, seed);   // INDM_GN_FUSED=3: the A/B arm "serial fold, arguments loaded at the start"
  __shared__ float sh[2 * 256];
  __shared__ float st[2 * GNF_MAXG];
  const int bs = blockDim.x, tid = threadIdx.x;
  int n, gb;
  gnf_decode(blockIdx.x, G / GS, N, n, gb);
  const int cpg = C / G, CS = GS * cpg, CS4 = CS >> 2, C4 = C >> 2;
  const int q = tid % CS4;                       // this thread's channel quad inside the slab (bs % CS4 == 0)
  const int gl = (q * 4) / cpg;                  // ... and its group inside the block
  const int total = HW * CS4;
  const long img4 = (long)n * HW * C4;           // float4 index of the image
  const int c4_0 = gb * CS4;
  float4 v[NF4];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < NF4; ++k) {
    const int f = tid + k * bs;
    v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (f < total) {
      const int pix = f / CS4;
      v[k] = *(const float4*)(x + (img4 + (long)pix * C4 + c4_0 + q) * 4);
      s1 += (v[k].x + v[k].y) + (v[k].z + v[k].w);
      s2 += (v[k].x * v[k].x + v[k].y * v[k].y) + (v[k].z * v[k].z + v[k].w * v[k].w);
    }
  }
  sh[tid] = s1;
  sh[256 + tid] = s2;
  __syncthreads();
  if (tid < GS) {   // fixed-order sum over the threads of this group (its cpg / 4 channel quads): deterministic
    float a = 0.f, b = 0.f;
    for (int qq = tid * (cpg >> 2); qq < (tid + 1) * (cpg >> 2); ++qq)
      for (int t = qq; t < bs; t += CS4) { a += sh[t]; b += sh[256 + t]; }
    const float inv_cnt = 1.f / ((float)HW * (float)cpg);
    const float m = a * inv_cnt;
    float var = b * inv_cnt - m * m;
    var = fmaxf(var, 0.f);
    const float r = rsqrtf(var + eps);
    st[tid] = m;
    st[GNF_MAXG + tid] = r;
    mean[n * G + gb * GS + tid] = m;
    rstd[n * G + gb * GS + tid] = r;
  }
  __syncthreads();
  const float m = st[gl], r = st[GNF_MAXG + gl];
  const float4 gm = *(const float4*)(gamma + (c4_0 + q) * 4);
  const float4 bt = *(const float4*)(beta + (c4_0 + q) * 4);
  const float gmv[4] = {gm.x, gm.y, gm.z, gm.w}, btv[4] = {bt.x, bt.y, bt.z, bt.w};
#pragma unroll
  for (int k = 0; k < NF4; ++k) {
    const int f = tid + k * bs;
    if (f < total) {
      const int pix = f / CS4;
      const long i4 = img4 + (long)pix * C4 + c4_0 + q;
      const float in[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
      float out[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float u = (in[j] - m) * r * gmv[j] + btv[j];
        if (act == 1) u = silu_f(u);
        if (drop_p > 0.f) u *= dropout_keep(seed, (uint64_t)i4, j, drop_p);
        out[j] = u;
      }
      *(float4*)(y + i4 * 4) = make_float4(out[0], out[1], out[2], out[3]);
    }
  }
}

template <int NF4>
__global__ __launch_bounds__(256) void gn_fused_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            float* __restrict__ dx, float* __restrict__ cpart, int N, int HW,
                                                            int C, int G, int GS, int act, float drop_p, uint64_t seed,
                                                            const uint64_t* __restrict__ seed_dev) {
  if (seed_dev) seed += *seed_dev;   // graph-captured launches: the per-call seed lives in device memory (indm_amd/graphed.py)
  __shared__ float sh[4 * 8 * 8];         // [wave][quad][4: sum d, 4: sum d*xh]
  INDM_PIN_ARGS9(gamma, beta, mean, rstd, dx, cpart, act, drop_p, seed);
  __shared__ float chs[2 * 32];           // channel sums of the slab (CS <= 32)
  __shared__ float gs_[2 * GNF_MAXG];     // group sums
  const int bs = blockDim.x, tid = threadIdx.x;
  int n, gb;
  gnf_decode(blockIdx.x, G / GS, N, n, gb);
  const int cpg = C / G, CS = GS * cpg, CS4 = CS >> 2, C4 = C >> 2;
  const int q = tid % CS4, gl = (q * 4) / cpg;
  const int total = HW * CS4;
  const long img4 = (long)n * HW * C4;
  const int c4_0 = gb * CS4;
  const float4 gm = *(const float4*)(gamma + (c4_0 + q) * 4);
  const float4 bt = *(const float4*)(beta + (c4_0 + q) * 4);
  const float gmv[4] = {gm.x, gm.y, gm.z, gm.w}, btv[4] = {bt.x, bt.y, bt.z, bt.w};
  const float m = mean[n * G + gb * GS + gl], r = rstd[n * G + gb * GS + gl];
  float4 xh[NF4], dd[NF4];                 // normalised input and dL/du of this thread's chunks
  float a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < NF4; ++k) {
    const int f = tid + k * bs;
    xh[k] = dd[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (f < total) {
      const int pix = f / CS4;
      const long i4 = img4 + (long)pix * C4 + c4_0 + q;
      const float4 xv = *(const float4*)(x + i4 * 4);
      const float4 dv = *(const float4*)(dy + i4 * 4);
      const float xi[4] = {xv.x, xv.y, xv.z, xv.w}, di[4] = {dv.x, dv.y, dv.z, dv.w};
      float h[4], d[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        h[j] = (xi[j] - m) * r;
        d[j] = di[j];
        if (drop_p > 0.f) d[j] *= dropout_keep(seed, (uint64_t)i4, j, drop_p);
        if (act == 1) d[j] *= silu_grad(h[j] * gmv[j] + btv[j]);
        a1[j] += d[j];
        a2[j] += d[j] * h[j];
      }
      xh[k] = make_float4(h[0], h[1], h[2], h[3]);
      dd[k] = make_float4(d[0], d[1], d[2], d[3]);
    }
  }
  // channel sums of the slab: per-wave butterfly sums over the lanes that own the same channel quad (fixed order), one LDS
  // row per (wave, quad), then the waves' rows are added in wave order
  const int wave = tid >> 6, nw = bs >> 6;
  if ((CS4 & (CS4 - 1)) == 0) {
    // 1, 2, 4 or 8 quads: the lanes that own one quad are CS4 apart (64 % CS4 == 0) -> strided butterfly, log2(64 / CS4)
    // steps per value; afterwards every lane holds its own quad's wave sum and lanes 0 .. CS4-1 (quad = lane) publish it
    float v8[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { v8[j] = a1[j]; v8[4 + j] = a2[j]; }
    for (int off = CS4; off < 64; off <<= 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v8[j] += __shfl_xor(v8[j], off, 64);
    }
    if ((tid & 63) < CS4) {
#pragma unroll
      for (int j = 0; j < 8; ++j) sh[(wave * 8 + (tid & 63)) * 8 + j] = v8[j];
    }
  } else {
#pragma unroll
    for (int qq = 0; qq < 8; ++qq) {   // 3 or 6 quads (bs = 192): masked full-wave sums
      if (qq < CS4) {
        float v8[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v8[j] = wave_sum(q == qq ? a1[j] : 0.f);
          v8[4 + j] = wave_sum(q == qq ? a2[j] : 0.f);
        }
        if ((tid & 63) == 0) {
#pragma unroll
          for (int j = 0; j < 8; ++j) sh[(wave * 8 + qq) * 8 + j] = v8[j];
        }
      }
    }
  }
  __syncthreads();
  if (tid < CS) {   // channel tid = quad cq, component cj
    const int cq = tid >> 2, cj = tid & 3;
    float a = 0.f, b = 0.f;
    for (int w = 0; w < nw; ++w) {
      a += sh[(w * 8 + cq) * 8 + cj];
      b += sh[(w * 8 + cq) * 8 + 4 + cj];
    }
    chs[tid] = a;
    chs[32 + tid] = b;
    cpart[(long)n * 2 * C + gb * CS + tid] = a;          // dbeta partial of image n
    cpart[(long)n * 2 * C + C + gb * CS + tid] = b;      // dgamma partial
  }
  __syncthreads();
  if (tid < GS) {
    float a = 0.f, b = 0.f;
    for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) {
      const float g = gamma[gb * CS + c];
      a += chs[c] * g;
      b += chs[32 + c] * g;
    }
    gs_[tid] = a;
    gs_[GNF_MAXG + tid] = b;
  }
  __syncthreads();
  const float inv_cnt = 1.f / ((float)HW * (float)cpg);
  const float s1 = gs_[gl] * inv_cnt, s2 = gs_[GNF_MAXG + gl] * inv_cnt;
#pragma unroll
  for (int k = 0; k < NF4; ++k) {
    const int f = tid + k * bs;
    if (f < total) {
      const int pix = f / CS4;
      const long i4 = img4 + (long)pix * C4 + c4_0 + q;
      const float h[4] = {xh[k].x, xh[k].y, xh[k].z, xh[k].w}, d[4] = {dd[k].x, dd[k].y, dd[k].z, dd[k].w};
      float out[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) out[j] = r * (gmv[j] * d[j] - s1 - h[j] * s2);
      *(float4*)(dx + i4 * 4) = make_float4(out[0], out[1], out[2], out[3]);
    }
  }
}

// dbeta[c] = sum_n cpart[n][c], dgamma[c] = sum_n cpart[n][C + c]: one wavefront per channel (fixed order -> deterministic)
__global__ __launch_bounds__(256) void gn_cpart_reduce_kernel(const float* __restrict__ cpart, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, int N, int C) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  float db = 0.f, dg = 0.f;
  for (int n = lane; n < N; n += 64) {
    db += cpart[(long)n * 2 * C + c];
    dg += cpart[(long)n * 2 * C + C + c];
  }
  db = wave_sum(db);
  dg = wave_sum(dg);
  if (lane == 0) {
    dbeta[c] = db;
    dgamma[c] = dg;
  }
}

// geometry of the fused path: groups per workgroup, threads, float4 chunks per thread (0 = not covered)
struct GnFused { int gs, bs, nf4; };
static GnFused gn_fused_geom(int HW, int C, int G) {
  GnFused f = {0, 0, 0};
  static const int mode = getenv("INDM_GN_FUSED") ? atoi(getenv("INDM_GN_FUSED")) : 1;   // 0: three-kernel passes (probes)
  if (!mode) return f;
  const int cpg = C / G;
  if (cpg < 4 || (cpg & 3)) return f;
  int gs = cpg >= 12 ? 1 : (16 / cpg);
  while (gs > 1 && (G % gs)) gs >>= 1;
  // small maps (8 x 8, 4 x 4): wider group blocks -- up to 32 channels = one 128-byte line per pixel -- so that a workgroup
  // has at least ~1024 chunks of work and the grid is not thousands of near-empty workgroups
  while ((long)HW * (gs * cpg / 4) < 1024 && 2 * gs * cpg <= 32 && 2 * gs <= GNF_MAXG && (G % (2 * gs)) == 0) gs *= 2;
  // (round 5 probe: 32-channel slabs = whole 128-byte lines per pixel wherever 16 chunks per thread still hold the slab measured
  // the same -- 16 x 16 x 256: 19.9 -> 21.6 us forward, 25.3 -> 24.7 us backward -- so the 16-channel blocks stay)
  const int cs4 = gs * cpg / 4;
  if (gs > GNF_MAXG || gs * cpg > 32 || (cs4 != 1 && cs4 != 2 && cs4 != 3 && cs4 != 4 && cs4 != 6 && cs4 != 8)) return f;
  const long total = (long)HW * cs4;
  const bool three = (cs4 % 3) == 0;                   // blockDim must be a multiple of cs4
  int bs = three ? 192 : 256;                          // whole wavefronts (the reductions are wave butterflies)
  if (!three && total <= 64) bs = 64;                  // 4 x 4 maps: one wavefront
  else if (!three && total <= 128) bs = 128;
  const int need = (int)((total + bs - 1) / bs);
  const int nf4 = need <= 1 ? 1 : need <= 2 ? 2 : need <= 4 ? 4 : need <= 8 ? 8 : need <= 16 ? 16 : 0;
  if (!nf4) return f;
  f.gs = gs; f.bs = bs; f.nf4 = nf4;
  return f;
}

// ---- softmax over the last dim (T <= 1024), one wavefront per row -------------------
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           long rows, int T, float scale) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * T;
  float* yr = y + row * T;
  float v[16];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int c = lane + j * 64;
    v[j] = c < T ? xr[c] * scale : -INFINITY;
    mx = fmaxf(mx, v[j]);
  }
  mx = wave_max(mx);
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int c = lane + j * 64;
    v[j] = c < T ? expf(v[j] - mx) : 0.f;
    s += v[j];
  }
  s = wave_sum(s);
  const float inv = 1.f / s;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int c = lane + j * 64;
    if (c < T) yr[c] = v[j] * inv;
  }
}

// dx = scale * y * (dy - sum(dy*y))
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                           float* __restrict__ dx, long rows, int T, float scale) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* dr = dy + row * T;
  const float* yr = y + row * T;
  float a[16], b[16];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int c = lane + j * 64;
    a[j] = c < T ? dr[c] : 0.f;
    b[j] = c < T ? yr[c] : 0.f;
    s += a[j] * b[j];
  }
  s = wave_sum(s);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int c = lane + j * 64;
    if (c < T) dx[row * T + c] = scale * b[j] * (a[j] - s);
  }
}

// ---- column reductions over a [R, C] matrix ------------------------------------------
// mode 0: sum x ; mode 1: sum x*y ; mode 2: sum x^2 ; mode 3: sum |x|.  Two-stage, deterministic.
__global__ __launch_bounds__(256) void colreduce_part_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                              float* __restrict__ part, long R, int C, int mode,
                                                              long rows_per_block) {
  // thread = (column-quad, row-lane): float4 loads along C, 4 independent row streams per thread, LDS combine
  __shared__ float4 sh[256];
  const long r0 = (long)blockIdx.y * rows_per_block, r1 = min(R, r0 + rows_per_block);
  const int C4 = (C + 3) >> 2;
  const int qpb = min(C4, 64);             // column quads per block
  const int lanes = 256 / qpb;             // row lanes per block
  const int q = blockIdx.x * qpb + (threadIdx.x % qpb);
  const int rl = threadIdx.x / qpb;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool vec = (C & 3) == 0;
  if (q < C4 && rl < lanes) {
    for (long r = r0 + rl; r < r1; r += lanes) {
      float4 v, w = make_float4(0.f, 0.f, 0.f, 0.f);
      const long off = r * C + (long)q * 4;
      if (vec) {
        v = *(const float4*)(x + off);
        if (mode == 1) w = *(const float4*)(y + off);
      } else {
        const int c = q * 4;
        v.x = x[off]; v.y = c + 1 < C ? x[off + 1] : 0.f; v.z = c + 2 < C ? x[off + 2] : 0.f; v.w = c + 3 < C ? x[off + 3] : 0.f;
        if (mode == 1) { w.x = y[off]; w.y = c + 1 < C ? y[off + 1] : 0.f; w.z = c + 2 < C ? y[off + 2] : 0.f; w.w = c + 3 < C ? y[off + 3] : 0.f; }
      }
      if (mode == 0) { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
      else if (mode == 1) { acc.x += v.x * w.x; acc.y += v.y * w.y; acc.z += v.z * w.z; acc.w += v.w * w.w; }
      else if (mode == 2) { acc.x += v.x * v.x; acc.y += v.y * v.y; acc.z += v.z * v.z; acc.w += v.w * v.w; }
      else { acc.x += fabsf(v.x); acc.y += fabsf(v.y); acc.z += fabsf(v.z); acc.w += fabsf(v.w); }
    }
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  if (rl == 0 && q < C4) {
    float4 s = sh[threadIdx.x];
    for (int l = 1; l < lanes; ++l) {
      const float4 t = sh[threadIdx.x + l * qpb];
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    const int c = q * 4;
    float* o = part + (long)blockIdx.y * C + c;
    o[0] = s.x;
    if (c + 1 < C) o[1] = s.y;
    if (c + 2 < C) o[2] = s.z;
    if (c + 3 < C) o[3] = s.w;
  }
}
// block = 64 columns x 4 partial-lanes; fixed combination order -> deterministic
// 64 columns x 16 row-lanes per block: every thread adds <= 16 partial rows with 4 independent accumulators, then a
// fixed-order LDS combine (deterministic).  The old 4-lane version spent ~11 us per call in a 64-deep dependent chain.
__global__ __launch_bounds__(1024) void colreduce_final_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                                int C, int P, float alpha, float beta) {
  __shared__ float sh[16][64];
  const int t = threadIdx.x & 63, l = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + t;
  part += (long)blockIdx.y * P * C;  // segmented form (indm_segreduce_f32): one output row per blockIdx.y
  out += (long)blockIdx.y * C;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < C) {
    int p = l;
    for (; p + 48 < P; p += 64) {
      s0 += part[(long)p * C + c];
      s1 += part[(long)(p + 16) * C + c];
      s2 += part[(long)(p + 32) * C + c];
      s3 += part[(long)(p + 48) * C + c];
    }
    for (; p < P; p += 16) s0 += part[(long)p * C + c];
  }
  sh[l][t] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (l == 0 && c < C) {
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) r += sh[k][t];
    out[c] = alpha * r + (beta != 0.f ? beta * out[c] : 0.f);
  }
}

// ---- BatchNorm2d (training-mode batch statistics) + optional ELU, rows = N*H*W, NHWC ----------
// y = act((x - mean[c]) * rstd[c] * gamma[c] + beta[c])
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y, long n,
                                                        int C, int act) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    float u = (x[i] - mean[c]) * rstd[c] * gamma[c] + beta[c];
    if (act == 2) u = u > 0.f ? u : expm1f(u);
    y[i] = u;
  }
}
// dyu = dy * act'(u)
__global__ __launch_bounds__(256) void bn_dyu_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* __restrict__ dyu, long n, int C, int act) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    float d = dy[i];
    if (act == 2) {
      const float u = (x[i] - mean[c]) * rstd[c] * gamma[c] + beta[c];
      d *= u > 0.f ? 1.f : expf(u);
    }
    dyu[i] = d;
  }
}
// dx = gamma*rstd * (dyu - sdy[c]/R - xhat * sdyx[c]/R);  sdy = sum dyu, sdyx = sum dyu*xhat
__global__ __launch_bounds__(256) void bn_dx_kernel(const float* __restrict__ dyu, const float* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ sdy,
                                                     const float* __restrict__ sdyx, float* __restrict__ dx, long n,
                                                     int C, float invR) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const float xh = (x[i] - mean[c]) * rstd[c];
    dx[i] = gamma[c] * rstd[c] * (dyu[i] - sdy[c] * invR - xh * sdyx[c] * invR);
  }
}
// mean/var from column sums; also emits dgamma = rstd*(sum(dyu*x) - mean*sum(dyu)) for the backward
__global__ void bn_stats_final_kernel(const float* __restrict__ s1, const float* __restrict__ s2, float* __restrict__ mean,
                                      float* __restrict__ var, float* __restrict__ rstd, int C, float invR, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float m = s1[c] * invR;
  float v = s2[c] * invR - m * m;
  v = fmaxf(v, 0.f);
  mean[c] = m;
  var[c] = v;
  rstd[c] = rsqrtf(v + eps);
}
__global__ void bn_dgamma_kernel(const float* __restrict__ sdy, const float* __restrict__ sdyx_raw,
                                 const float* __restrict__ mean, const float* __restrict__ rstd,
                                 float* __restrict__ dgamma, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  dgamma[c] = rstd[c] * (sdyx_raw[c] - mean[c] * sdy[c]);
}

int gn_split(int N, int HW) {
  int S = (GN_SPLIT_TARGET + N - 1) / N;
  if (S > HW / 4) S = HW / 4;
  if (S < 1) S = 1;
  if (S > 64) S = 64;
  return S;
}

}  // namespace

// Workspace (floats) needed by the GroupNorm entry points.
INDM_API long indm_groupnorm_ws_floats(int N, int HW, int C, int G) {
  const int S = gn_split(N, HW);
  return (long)N * S * (2 * G + 2 * C) + 2L * N * G + 16;
}

// 1 if indm_groupnorm_fwd/bwd_f32 run the single-pass register-resident kernels for this geometry, 0 if the three-kernel passes
INDM_API int indm_groupnorm_fused(int HW, int C, int G) {
  if (C % 4 != 0 || G <= 0 || C % G != 0) return 0;
  return gn_fused_geom(HW, C, G).nf4 ? 1 : 0;
}

// y = dropout(act(GroupNorm(x)));  x,y: [N, HW, C];  mean,rstd: [N, G] (saved for backward)
INDM_API int indm_groupnorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y,
                                    float* mean, float* rstd, int N, int HW, int C, int G, float eps, int act,
                                    float drop_p, unsigned long long seed, const unsigned long long* seed_dev_, float* ws,
                                    void* stream) {
  if (N <= 0) return INDM_OK;
  if (C % 4 != 0 || G <= 0 || C % G != 0 || (C / 4) > 256) return INDM_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const uint64_t* seed_dev = (const uint64_t*)seed_dev_;
  const GnFused ff = gn_fused_geom(HW, C, G);
  if (ff.nf4) {   // single pass: one workgroup per (image, group block) keeps its slab in registers
    const dim3 grid((G / ff.gs) * N), block(ff.bs);
    static const int serial = getenv("INDM_GN_FUSED") ? atoi(getenv("INDM_GN_FUSED")) : 1;   // 2 / 3: the round-3 reproducer
#define GNF_FWD(NF)                                                                                                          \
  do {                                                                                                                       \
    if (serial == 2) hipLaunchKernelGGL((gn_fused_fwd_serial_kernel<NF, false>), grid, block, 0, st, x, gamma, beta, y, mean,  \
                                        rstd, N, HW, C, G, ff.gs, eps, act, drop_p, (uint64_t)seed, seed_dev);                         \
    else if (serial == 3) hipLaunchKernelGGL((gn_fused_fwd_serial_kernel<NF, true>), grid, block, 0, st, x, gamma, beta, y,   \
                                             mean, rstd, N, HW, C, G, ff.gs, eps, act, drop_p, (uint64_t)seed, seed_dev);              \
    else hipLaunchKernelGGL(gn_fused_fwd_kernel<NF>, grid, block, 0, st, x, gamma, beta, y, mean, rstd, N, HW, C, G, ff.gs,  \
                            eps, act, drop_p, (uint64_t)seed, seed_dev);                                                               \
  } while (0)
    switch (ff.nf4) {
      case 1: GNF_FWD(1); break;
      case 2: GNF_FWD(2); break;
      case 4: GNF_FWD(4); break;
      case 8: GNF_FWD(8); break;
      default: GNF_FWD(16); break;
    }
#undef GNF_FWD
    INDM_CHECK_LAUNCH();
    return INDM_OK;
  }
  const int S = gn_split(N, HW);
  hipLaunchKernelGGL(gn_stats_kernel, dim3(S, N), dim3(256), 2 * (256 / (C / 4)) * C * sizeof(float), st, x, ws, HW, C, G, S);
  INDM_CHECK_LAUNCH();
  const float inv_cnt = 1.f / ((float)HW * (float)(C / G));
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(cdiv(N * G, 256)), dim3(256), 0, st, ws, mean, rstd, N * G, G, S,
                     inv_cnt, eps);
  INDM_CHECK_LAUNCH();
  const long total4 = (long)N * HW * C / 4;
  hipLaunchKernelGGL(gn_apply_kernel, dim3(ew_grid(total4, 256)), dim3(256), 0, st, x, gamma, beta, mean, rstd, y,
                     total4, HW, C, G, act, drop_p, (uint64_t)seed, seed_dev);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}

INDM_API int indm_groupnorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* beta,
                                    const float* mean, const float* rstd, float* dx, float* dgamma, float* dbeta,
                                    int N, int HW, int C, int G, int act, float drop_p, unsigned long long seed,
                                    const unsigned long long* seed_dev_, float* ws, void* stream) {
  if (N <= 0) return INDM_OK;
  if (C % 4 != 0 || G <= 0 || C % G != 0 || (C / 4) > 256) return INDM_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const uint64_t* seed_dev = (const uint64_t*)seed_dev_;
  const GnFused ff = gn_fused_geom(HW, C, G);
  if (ff.nf4) {   // single pass over dy and x + a [N, C] column reduce for dgamma / dbeta
    const dim3 grid((G / ff.gs) * N), block(ff.bs);
#define GNF_BWD(NF) hipLaunchKernelGGL(gn_fused_bwd_kernel<NF>, grid, block, 0, st, dy, x, gamma, beta, mean, rstd, dx, ws, N, HW, \
                                       C, G, ff.gs, act, drop_p, (uint64_t)seed, seed_dev)
    switch (ff.nf4) {
      case 1: GNF_BWD(1); break;
      case 2: GNF_BWD(2); break;
      case 4: GNF_BWD(4); break;
      case 8: GNF_BWD(8); break;
      default: GNF_BWD(16); break;
    }
#undef GNF_BWD
    INDM_CHECK_LAUNCH();
    hipLaunchKernelGGL(gn_cpart_reduce_kernel, dim3(cdiv(C, 4)), dim3(256), 0, st, ws, dgamma, dbeta, N, C);
    INDM_CHECK_LAUNCH();
    return INDM_OK;
  }
  const int S = gn_split(N, HW);
  float* gpart = ws;
  float* cpart = gpart + (long)N * S * 2 * G;
  float* gsum = cpart + (long)N * S * 2 * C;
  hipLaunchKernelGGL(gn_bwd_stats_kernel, dim3(S, N), dim3(256), (2 * (256 / (C / 4)) * C + 2 * C) * sizeof(float), st, dy, x, gamma,
                     beta, mean, rstd, gpart, cpart, HW, C, G, S, act, drop_p, (uint64_t)seed, seed_dev);
  INDM_CHECK_LAUNCH();
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(N * G + C), dim3(256), 0, st, gpart, cpart, gsum, dgamma, dbeta, N, C,
                     G, S);
  INDM_CHECK_LAUNCH();
  const long total4 = (long)N * HW * C / 4;
  const float inv_cnt = 1.f / ((float)HW * (float)(C / G));
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(ew_grid(total4, 256)), dim3(256), 0, st, dy, x, gamma, beta, mean,
                     rstd, gsum, dx, total4, HW, C, G, act, drop_p, (uint64_t)seed, inv_cnt, seed_dev);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}

// y[r,:] = softmax(scale * x[r,:]), T <= 1024
INDM_API int indm_softmax_fwd_f32(const float* x, float* y, long rows, int T, float scale, void* stream) {
  if (rows <= 0) return INDM_OK;
  if (T <= 0 || T > 1024) return INDM_E_SHAPE;
  hipLaunchKernelGGL(softmax_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, y, rows, T, scale);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}
INDM_API int indm_softmax_bwd_f32(const float* dy, const float* y, float* dx, long rows, int T, float scale,
                                  void* stream) {
  if (rows <= 0) return INDM_OK;
  if (T <= 0 || T > 1024) return INDM_E_SHAPE;
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, rows, T,
                     scale);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}

static long colreduce_parts(long R, int C) {
  // enough row-blocks to fill the chip (~2048 workgroups with the column blocks), at most 256 partial rows
  const int C4 = (C + 3) / 4;
  const int cb = (C4 + 63) / 64;
  long P = 2048 / cb;
  if (P > 256) P = 256;
  const long maxp = (R + 63) / 64;
  if (P > maxp) P = maxp;
  if (P < 1) P = 1;
  return P;
}
INDM_API long indm_colreduce_ws_floats(long R, int C) { return colreduce_parts(R, C) * C; }
// out[c] = alpha * sum_r f(x[r,c], y[r,c]) + beta*out[c]; mode 0: x, 1: x*y, 2: x*x
INDM_API int indm_colreduce_f32(const float* x, const float* y, float* out, long R, int C, int mode, float alpha,
                                float beta, float* ws, void* stream) {
  if (C <= 0) return INDM_OK;
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0 && R <= 1024) {
    // a short matrix (the [images, C] per-image bias gradients -> the bias gradient): the second stage alone, with x
    // as its partial rows -- one launch
    hipLaunchKernelGGL(colreduce_final_kernel, dim3(cdiv(C, 64)), dim3(1024), 0, st, x, out, C, (int)R, alpha, beta);
    INDM_CHECK_LAUNCH();
    return INDM_OK;
  }
  const long P = colreduce_parts(R, C);
  const long rpb = (R + P - 1) / P;
  const int C4 = (C + 3) / 4;
  hipLaunchKernelGGL(colreduce_part_kernel, dim3(cdiv(C4, 64), (int)P), dim3(256), 0, st, x, y, ws, R, C, mode, rpb);
  INDM_CHECK_LAUNCH();
  hipLaunchKernelGGL(colreduce_final_kernel, dim3(cdiv(C, 64)), dim3(1024), 0, st, ws, out, C, (int)P, alpha, beta);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}

// Segmented column sums: out[s][c] = alpha * sum_{r < rps} x[(s*rps + r)*C + c]  -- the per-image bias gradients
// (bias2 of the GEMM / conv epilogues: losses w.r.t. the flow's conditioning bias and the score net's temb bias).
// Each segment is cut into S row blocks (S | rps) so that ~2048 workgroups stream the tensor; deterministic.
static long segreduce_split(long nseg, long rps, int C) {
  const int cb = ((C + 3) / 4 + 63) / 64;
  long S = 1;
  while (S < 64 && nseg * S * cb < 2048 && rps % (2 * S) == 0 && rps / (2 * S) >= 16) S *= 2;
  return S;
}
INDM_API long indm_segreduce_ws_floats(long nseg, long rps, int C) { return nseg * segreduce_split(nseg, rps, C) * C; }
INDM_API int indm_segreduce_f32(const float* x, float* out, long nseg, long rps, int C, float alpha, float* ws,
                                void* stream) {
  if (C <= 0 || nseg <= 0) return INDM_OK;
  if (rps <= 0) return INDM_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const long S = segreduce_split(nseg, rps, C);
  if (nseg * S > 65535) return INDM_E_SHAPE;
  const int C4 = (C + 3) / 4;
  hipLaunchKernelGGL(colreduce_part_kernel, dim3(cdiv(C4, 64), (int)(nseg * S)), dim3(256), 0, st, x, (const float*)nullptr,
                     ws, nseg * rps, C, 0, rps / S);
  INDM_CHECK_LAUNCH();
  hipLaunchKernelGGL(colreduce_final_kernel, dim3(cdiv(C, 64), (int)nseg), dim3(1024), 0, st, ws, out, C, (int)S, alpha, 0.f);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}

// BatchNorm statistics over rows: mean[c], var[c] (biased), rstd[c] = rsqrt(var+eps).  ws >= colreduce ws + 2C.
INDM_API int indm_bn_stats_f32(const float* x, float* mean, float* var, float* rstd, long R, int C, float eps,
                               float* ws, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  float* s1 = ws;
  float* s2 = ws + C;
  float* cws = ws + 2 * C;
  int rc = indm_colreduce_f32(x, nullptr, s1, R, C, 0, 1.f, 0.f, cws, stream);
  if (rc) return rc;
  rc = indm_colreduce_f32(x, nullptr, s2, R, C, 2, 1.f, 0.f, cws, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(bn_stats_final_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, s1, s2, mean, var, rstd, C,
                     1.f / (float)R, eps);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}
// y = act(BN(x)) with given statistics; act: 0 none, 2 ELU
INDM_API int indm_bn_apply_f32(const float* x, const float* mean, const float* rstd, const float* gamma,
                               const float* beta, float* y, long R, int C, int act, void* stream) {
  const long n = R * C;
  if (n <= 0) return INDM_OK;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, x, mean, rstd, gamma,
                     beta, y, n, C, act);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}
// Backward of y = act(BN_batchstats(x)): dx, dgamma, dbeta.  ws >= colreduce ws + R*C + 2C floats.
INDM_API int indm_bn_bwd_f32(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                             const float* beta, float* dx, float* dgamma, float* dbeta, long R, int C, int act,
                             float* ws, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const long n = R * C;
  if (n <= 0) return INDM_OK;
  float* dyu = ws;
  float* sdyx = ws + n;
  float* cws = sdyx + 2 * C;
  hipLaunchKernelGGL(bn_dyu_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, st, dy, x, mean, rstd, gamma, beta, dyu, n, C,
                     act);
  INDM_CHECK_LAUNCH();
  int rc = indm_colreduce_f32(dyu, nullptr, dbeta, R, C, 0, 1.f, 0.f, cws, stream);
  if (rc) return rc;
  rc = indm_colreduce_f32(dyu, x, sdyx, R, C, 1, 1.f, 0.f, cws, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(bn_dgamma_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, dbeta, sdyx, mean, rstd, dgamma, C);
  INDM_CHECK_LAUNCH();
  hipLaunchKernelGGL(bn_dx_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, st, dyu, x, mean, rstd, gamma, dbeta, dgamma,
                     dx, n, C, 1.f / (float)R);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}

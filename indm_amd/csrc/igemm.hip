// fp32 MFMA implicit-GEMM engine for gfx950 (MI355X).
//
// One kernel template covers every dense contraction on the INDM step:
//   * plain / transposed / batched GEMM (1x1 conv, NIN, Linear, attention bmm's)
//   * NHWC convolution as implicit GEMM (forward, data-grad, weight-grad)
// C[M,N] = epilogue( sum_k Aeff[m,k] * Beff[k,n] ), where Aeff / Beff are index maps
// over the caller's tensors (no im2col buffer is ever materialised in HBM).
//
// Hardware mapping (see DESIGN.md "igemm"):
//   * v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD = 157 TF chip peak)
//   * 256 threads = 4 waves per workgroup; block tiles 128x128 (2x2 waves of 64x64),
//     128x32 and 32x128 (for the 3-/12-/27-wide flow edges)
//   * both operands staged through LDS in [k][row] order so every fragment read is a
//     conflict-free ds_read_b32 of 32 consecutive floats per half-wave
//   * register prefetch of the next K-slab + double-buffered LDS: one barrier per slab
//   * XCD-aware tile order: the N-tiles that share an A row-panel land on one XCD's L2
//   * optional split-K into a caller-provided workspace (deterministic reduce kernel)
#include "indm_common.h"
#include <cstdlib>
#include <stdlib.h>

namespace {

#ifndef INDM_BK
#define INDM_BK 16
#endif
constexpr int BK = INDM_BK;

enum { A_ROW = 0, A_COL = 1, A_CONV = 2, A_WGRAD = 3 };
enum { B_ROW = 0, B_COL = 1, B_DGRAD = 2 };

struct IgemmParams {
  const float* A;
  const float* G;  // optional elementwise gate on A (same layout): Aeff = A .* G (VJP chains of the flow)
  const float* B;
  float* C;
  int M, N, K;
  long lda, ldb, ldc;
  int batch;
  long sA, sB, sC;
  // convolution geometry (A_CONV / A_WGRAD / B_DGRAD)
  int H, W, Cin, OH, OW, KH, KW, stride, pad, dil;
  int Cout;  // B_DGRAD: inner (reduction) width per tap of the packed weight
  // epilogue: C = alpha*(acc + bias[n] + bias2[m/rpg][n]) + beta*R[m][n]
  float alpha, beta;
  const float* bias;
  const float* bias2;
  int rpg;
  const float* R;
  long ldr, sR;
  // split-K
  int splitk, kps;
  float* ws;
  int a_vec, b_vec;
  int a_tap_uniform, b_tap_uniform;  // reduction slab of BK never straddles a tap (Cin % BK == 0)
};

struct PixDec {
  int img, oh, ow;
};

__device__ __forceinline__ PixDec decode_pix(int m, int OH, int OW) {
  PixDec d;
  const int hw = OH * OW;
  d.img = m / hw;
  const int r = m - d.img * hw;
  d.oh = r / OW;
  d.ow = r - d.oh * OW;
  return d;
}

// ---- scalar element maps (slow path + semantic definition of each mode) ----
template <int AM>
__device__ __forceinline__ float a_elem(const IgemmParams& p, const float* A, int m, int k, int kend) {
  if (m >= p.M || k >= kend) return 0.f;
  if constexpr (AM == A_ROW) {
    return A[(long)m * p.lda + k];
  } else if constexpr (AM == A_COL) {
    return A[(long)k * p.lda + m];
  } else if constexpr (AM == A_CONV) {
    const PixDec d = decode_pix(m, p.OH, p.OW);
    const int tap = k / p.Cin, ci = k - tap * p.Cin;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    int ih = d.oh * p.stride + kh - p.pad, iw = d.ow * p.stride + kw - p.pad;
    if (ih < 0 || iw < 0) return 0.f;
    if (p.dil > 1) {
      if ((ih % p.dil) | (iw % p.dil)) return 0.f;
      ih /= p.dil;
      iw /= p.dil;
    }
    if (ih >= p.H || iw >= p.W) return 0.f;
    return A[(((long)d.img * p.H + ih) * p.W + iw) * p.Cin + ci];
  } else {  // A_WGRAD: row m = (tap, ci), reduction k = output pixel
    const PixDec d = decode_pix(k, p.OH, p.OW);
    const int tap = m / p.Cin, ci = m - tap * p.Cin;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const int ih = d.oh * p.stride + kh - p.pad, iw = d.ow * p.stride + kw - p.pad;
    if (ih < 0 || iw < 0 || ih >= p.H || iw >= p.W) return 0.f;
    return A[(((long)d.img * p.H + ih) * p.W + iw) * p.Cin + ci];
  }
}

template <int BMODE>
__device__ __forceinline__ float b_elem(const IgemmParams& p, const float* B, int k, int n, int kend) {
  if (n >= p.N || k >= kend) return 0.f;
  if constexpr (BMODE == B_ROW) {
    return B[(long)k * p.ldb + n];
  } else if constexpr (BMODE == B_COL) {
    return B[(long)n * p.ldb + k];
  } else {  // B_DGRAD: k = (tap', co) ; n = ci of the packed weight [T][N][Cout]
    const int tap = k / p.Cout, co = k - tap * p.Cout;
    const int T = p.KH * p.KW;
    return B[((long)(T - 1 - tap) * p.N + n) * p.Cout + co];
  }
}

// ---- vectorised group loads -------------------------------------------------
// K-contiguous group: (row, k..k+3).  Row-contiguous group: (row..row+3, k).
template <int AM>
__device__ __forceinline__ float4 a_group(const IgemmParams& p, const float* A, const float* G, int row, int k,
                                          int kend, const PixDec& rowdec, const int3& tapdec, int kbase) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (AM == A_ROW) {
    if (p.a_vec && row < p.M && k + 3 < kend) {
      v = *(const float4*)(A + (long)row * p.lda + k);
      if (G) {
        const float4 g = *(const float4*)(G + (long)row * p.lda + k);
        v.x *= g.x; v.y *= g.y; v.z *= g.z; v.w *= g.w;
      }
      return v;
    }
    v.x = a_elem<AM>(p, A, row, k, kend);
    v.y = a_elem<AM>(p, A, row, k + 1, kend);
    v.z = a_elem<AM>(p, A, row, k + 2, kend);
    v.w = a_elem<AM>(p, A, row, k + 3, kend);
    if (G) {
      v.x *= a_elem<AM>(p, G, row, k, kend);
      v.y *= a_elem<AM>(p, G, row, k + 1, kend);
      v.z *= a_elem<AM>(p, G, row, k + 2, kend);
      v.w *= a_elem<AM>(p, G, row, k + 3, kend);
    }
  } else if constexpr (AM == A_COL) {
    if (p.a_vec && k < kend && row + 3 < p.M) {
      v = *(const float4*)(A + (long)k * p.lda + row);
      if (G) {
        const float4 g = *(const float4*)(G + (long)k * p.lda + row);
        v.x *= g.x; v.y *= g.y; v.z *= g.z; v.w *= g.w;
      }
      return v;
    }
    v.x = a_elem<AM>(p, A, row, k, kend);
    v.y = a_elem<AM>(p, A, row + 1, k, kend);
    v.z = a_elem<AM>(p, A, row + 2, k, kend);
    v.w = a_elem<AM>(p, A, row + 3, k, kend);
    if (G) {
      v.x *= a_elem<AM>(p, G, row, k, kend);
      v.y *= a_elem<AM>(p, G, row + 1, k, kend);
      v.z *= a_elem<AM>(p, G, row + 2, k, kend);
      v.w *= a_elem<AM>(p, G, row + 3, k, kend);
    }
  } else if constexpr (AM == A_CONV) {
    if (p.a_vec) {
      if (row >= p.M || k >= kend) return v;
      // rowdec = (img, oh, ow) hoisted out of the K loop; tapdec = (kh, kw, ci) uniform per slab when aligned
      int kh, kw, ci;
      if (p.a_tap_uniform) {
        kh = tapdec.x; kw = tapdec.y; ci = tapdec.z + (k - kbase);
      } else {
        const int tap = k / p.Cin;
        ci = k - tap * p.Cin;
        kh = tap / p.KW; kw = tap - kh * p.KW;
      }
      int ih = rowdec.oh * p.stride + kh - p.pad, iw = rowdec.ow * p.stride + kw - p.pad;
      if (ih < 0 || iw < 0) return v;
      if (p.dil > 1) {
        if ((ih % p.dil) | (iw % p.dil)) return v;
        ih /= p.dil;
        iw /= p.dil;
      }
      if (ih >= p.H || iw >= p.W) return v;
      return *(const float4*)(A + (((long)rowdec.img * p.H + ih) * p.W + iw) * p.Cin + ci);
    }
    v.x = a_elem<AM>(p, A, row, k, kend);
    v.y = a_elem<AM>(p, A, row, k + 1, kend);
    v.z = a_elem<AM>(p, A, row, k + 2, kend);
    v.w = a_elem<AM>(p, A, row, k + 3, kend);
  } else {  // A_WGRAD, row-contiguous over ci; rowdec = (kh, kw, ci) of this row group (hoisted)
    if (p.a_vec) {
      if (row >= p.M || k >= kend) return v;
      const PixDec d = decode_pix(k, p.OH, p.OW);
      const int ih = d.oh * p.stride + rowdec.img - p.pad, iw = d.ow * p.stride + rowdec.oh - p.pad;
      if (ih < 0 || iw < 0 || ih >= p.H || iw >= p.W) return v;
      return *(const float4*)(A + (((long)d.img * p.H + ih) * p.W + iw) * p.Cin + rowdec.ow);
    }
    v.x = a_elem<AM>(p, A, row, k, kend);
    v.y = a_elem<AM>(p, A, row + 1, k, kend);
    v.z = a_elem<AM>(p, A, row + 2, k, kend);
    v.w = a_elem<AM>(p, A, row + 3, k, kend);
  }
  return v;
}

template <int BMODE>
__device__ __forceinline__ float4 b_group(const IgemmParams& p, const float* B, int row, int k, int kend,
                                          const int2& btap, int kbase) {
  float4 v;
  if constexpr (BMODE == B_ROW) {
    if (p.b_vec && k < kend && row + 3 < p.N) return *(const float4*)(B + (long)k * p.ldb + row);
    v.x = b_elem<BMODE>(p, B, k, row, kend);
    v.y = b_elem<BMODE>(p, B, k, row + 1, kend);
    v.z = b_elem<BMODE>(p, B, k, row + 2, kend);
    v.w = b_elem<BMODE>(p, B, k, row + 3, kend);
  } else if constexpr (BMODE == B_COL) {
    if (p.b_vec && row < p.N && k + 3 < kend) return *(const float4*)(B + (long)row * p.ldb + k);
    v.x = b_elem<BMODE>(p, B, k, row, kend);
    v.y = b_elem<BMODE>(p, B, k + 1, row, kend);
    v.z = b_elem<BMODE>(p, B, k + 2, row, kend);
    v.w = b_elem<BMODE>(p, B, k + 3, row, kend);
  } else {
    if (p.b_vec && row < p.N && k + 3 < kend) {
      int tap, co;
      if (p.b_tap_uniform) {
        tap = btap.x; co = btap.y + (k - kbase);
      } else {
        tap = k / p.Cout; co = k - tap * p.Cout;
      }
      const int T = p.KH * p.KW;
      return *(const float4*)(B + ((long)(T - 1 - tap) * p.N + row) * p.Cout + co);
    }
    v.x = b_elem<BMODE>(p, B, k, row, kend);
    v.y = b_elem<BMODE>(p, B, k + 1, row, kend);
    v.z = b_elem<BMODE>(p, B, k + 2, row, kend);
    v.w = b_elem<BMODE>(p, B, k + 3, row, kend);
  }
  return v;
}

template <int AM>
constexpr bool a_kcontig() {
  return AM == A_ROW || AM == A_CONV;
}
template <int BMODE>
constexpr bool b_kcontig() {
  return BMODE == B_COL || BMODE == B_DGRAD;
}

// Bijective XCD-aware remap: hardware places workgroup b on XCD b%8; give each XCD a
// contiguous run of logical tile ids so neighbouring tiles share one L2.
__device__ __forceinline__ int xcd_remap(int b, int n) {
  const int q = n >> 3, r = n & 7;
  const int xcd = b & 7, idx = b >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

template <int AM, int BMODE, int TM, int TN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, (WAVES_M * WAVES_N == 4 ? 3 : 4)) void igemm_kernel(IgemmParams p) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int RM = TM / WAVES_M / 32, RN = TN / WAVES_N / 32;
  // row-contiguous operands: +4 padding keeps every [k][row] slab row 16-byte aligned -> one ds_write_b128 per
  // float4; k-contiguous operands keep +2 (conflict-free transposing ds_write_b32)
  constexpr int LDA_S = TM + (a_kcontig<AM>() ? 2 : 4), LDB_S = TN + (b_kcontig<BMODE>() ? 2 : 4);
  constexpr int KG = BK / 4;  // float4 groups along K per row
  constexpr int GA = TM * KG, GB = TN * KG;  // float4 groups per operand slab
  constexpr int NGA = (GA + NT - 1) / NT, NGB = (GB + NT - 1) / NT;
  __shared__ __attribute__((aligned(16))) float smem[2 * BK * (LDA_S + LDB_S)];
  float* As = smem;
  float* Bs = smem + 2 * BK * LDA_S;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_m = (p.M + TM - 1) / TM, tiles_n = (p.N + TN - 1) / TN;
  const int ntiles = tiles_m * tiles_n;
  const int lt = xcd_remap(blockIdx.x, ntiles);
  const int tm = lt / tiles_n, tn = lt - tm * tiles_n;
  const int m0 = tm * TM, n0 = tn * TN;
  const int z = blockIdx.y;
  const int bz = z / p.splitk, kz = z - bz * p.splitk;
  const float* A = p.A + (long)bz * p.sA;
  const float* G = p.G ? p.G + (long)bz * p.sA : nullptr;
  const float* B = p.B + (long)bz * p.sB;
  const int kbeg = kz * p.kps;
  const int kend = min(p.K, kbeg + p.kps);
  const int nk = (kend - kbeg + BK - 1) / BK;

  f32x16 acc[RM][RN];
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra0[NGA], rb0[NGB], ra1[NGA], rb1[NGB];  // two register sets: slab kt+1 (to be stored) and kt+2 (in flight)

  // loop-invariant per-thread coordinates of the A row groups
  PixDec rowdec[NGA];
#pragma unroll
  for (int j = 0; j < NGA; ++j) {
    const int g = tid + j * NT;
    rowdec[j].img = rowdec[j].oh = rowdec[j].ow = 0;
    if constexpr (AM == A_CONV) {
      const int row = m0 + g / KG;
      if (g < GA && row < p.M) rowdec[j] = decode_pix(row, p.OH, p.OW);
    } else if constexpr (AM == A_WGRAD) {
      constexpr int Q = TM / 4;
      const int row = m0 + (g % Q) * 4;
      if (g < GA && row < p.M) {
        const int tap = row / p.Cin;
        rowdec[j].ow = row - tap * p.Cin;       // ci
        rowdec[j].img = tap / p.KW;             // kh
        rowdec[j].oh = tap - rowdec[j].img * p.KW;  // kw
      }
    }
  }

  auto gload = [&](int kt, float4 (&ra)[NGA], float4 (&rb)[NGB]) {
    const int k0 = kbeg + kt * BK;
    int3 tapdec = make_int3(0, 0, 0);
    int2 btap = make_int2(0, 0);
    if constexpr (AM == A_CONV) {
      if (p.a_tap_uniform) {
        const int tap = k0 / p.Cin;
        tapdec.z = k0 - tap * p.Cin;
        tapdec.x = tap / p.KW;
        tapdec.y = tap - tapdec.x * p.KW;
      }
    }
    if constexpr (BMODE == B_DGRAD) {
      if (p.b_tap_uniform) {
        btap.x = k0 / p.Cout;
        btap.y = k0 - btap.x * p.Cout;
      }
    }
#pragma unroll
    for (int j = 0; j < NGA; ++j) {
      const int g = tid + j * NT;
      if (g < GA) {
        if constexpr (a_kcontig<AM>()) {
          ra[j] = a_group<AM>(p, A, G, m0 + g / KG, k0 + (g % KG) * 4, kend, rowdec[j], tapdec, k0);
        } else {
          constexpr int Q = TM / 4;
          ra[j] = a_group<AM>(p, A, G, m0 + (g % Q) * 4, k0 + g / Q, kend, rowdec[j], tapdec, k0);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NGB; ++j) {
      const int g = tid + j * NT;
      if (g < GB) {
        if constexpr (b_kcontig<BMODE>()) {
          rb[j] = b_group<BMODE>(p, B, n0 + g / KG, k0 + (g % KG) * 4, kend, btap, k0);
        } else {
          constexpr int Q = TN / 4;
          rb[j] = b_group<BMODE>(p, B, n0 + (g % Q) * 4, k0 + g / Q, kend, btap, k0);
        }
      }
    }
  };
  auto sstore = [&](int buf, const float4 (&ra)[NGA], const float4 (&rb)[NGB]) {
    float* as = As + buf * BK * LDA_S;
    float* bs = Bs + buf * BK * LDB_S;
#pragma unroll
    for (int j = 0; j < NGA; ++j) {
      const int g = tid + j * NT;
      if (g < GA) {
        if constexpr (a_kcontig<AM>()) {
          const int row = g / KG, k4 = (g % KG) * 4;
          as[(k4 + 0) * LDA_S + row] = ra[j].x;
          as[(k4 + 1) * LDA_S + row] = ra[j].y;
          as[(k4 + 2) * LDA_S + row] = ra[j].z;
          as[(k4 + 3) * LDA_S + row] = ra[j].w;
        } else {
          constexpr int Q = TM / 4;
          const int k = g / Q, row = (g % Q) * 4;
          *(float4*)(as + k * LDA_S + row) = ra[j];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NGB; ++j) {
      const int g = tid + j * NT;
      if (g < GB) {
        if constexpr (b_kcontig<BMODE>()) {
          const int row = g / KG, k4 = (g % KG) * 4;
          bs[(k4 + 0) * LDB_S + row] = rb[j].x;
          bs[(k4 + 1) * LDB_S + row] = rb[j].y;
          bs[(k4 + 2) * LDB_S + row] = rb[j].z;
          bs[(k4 + 3) * LDB_S + row] = rb[j].w;
        } else {
          constexpr int Q = TN / 4;
          const int k = g / Q, row = (g % Q) * 4;
          *(float4*)(bs + k * LDB_S + row) = rb[j];
        }
      }
    }
  };

  const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
  const int mw = wm * (TM / WAVES_M), nw = wn * (TN / WAVES_N);
  const int l31 = lane & 31, lhi = lane >> 5;

  // One K-slab of MFMAs out of LDS buffer `buf`.  `mid` runs after half of the k-pairs: it stores the NEXT slab
  // (already in registers) into the other LDS buffer, so the LDS-write issue and the vmcnt wait sit inside this
  // wave's own MFMA stream instead of in front of the barrier.
  auto compute = [&](int buf, auto&& mid) {
    const float* as = As + buf * BK * LDA_S + mw + l31;
    const float* bs = Bs + buf * BK * LDB_S + nw + l31;
    float a[2][RM], b[2][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i) a[0][i] = as[lhi * LDA_S + i * 32];
#pragma unroll
    for (int j = 0; j < RN; ++j) b[0][j] = bs[lhi * LDB_S + j * 32];
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const int cur = kk & 1, nxt = cur ^ 1;
      if (kk + 1 < BK / 2) {
        const int k = (kk + 1) * 2 + lhi;
#pragma unroll
        for (int i = 0; i < RM; ++i) a[nxt][i] = as[k * LDA_S + i * 32];
#pragma unroll
        for (int j = 0; j < RN; ++j) b[nxt][j] = bs[k * LDB_S + j * 32];
      }
#pragma unroll
      for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
      // pin the interleave: next k-pair's LDS reads first, then this k-pair's MFMAs (the LLVM scheduler otherwise
      // sinks the reads to just before their use and exposes the LDS latency every 4 MFMAs)
      if (kk + 1 < BK / 2) __builtin_amdgcn_sched_group_barrier(0x100, RM + RN, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, RM * RN, 0);
      if (kk == BK / 4 - 1) mid();
    }
  };

  // prologue: slab 0 -> LDS buffer 0, slab 1 -> register set 1
  if (nk > 0) {
    gload(0, ra0, rb0);
    sstore(0, ra0, rb0);
  }
  if (nk > 1) gload(1, ra1, rb1);
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    // even slab: MFMAs on buffer 0; set 1 holds slab kt+1 (stored mid-way); set 0 receives slab kt+2
    if (kt + 2 < nk) gload(kt + 2, ra0, rb0);
    compute(0, [&] { if (kt + 1 < nk) sstore(1, ra1, rb1); });
    __syncthreads();
    if (kt + 1 < nk) {
      if (kt + 3 < nk) gload(kt + 3, ra1, rb1);
      compute(1, [&] { if (kt + 2 < nk) sstore(0, ra0, rb0); });
      __syncthreads();
    }
  }

  // ---- epilogue ----
  const bool partial = p.splitk > 1;
  float* C = partial ? p.ws + ((long)z * p.M) * p.N : p.C + (long)bz * p.sC;
  const long ldc = partial ? p.N : p.ldc;
  const float* R = p.R ? p.R + (long)bz * p.sR : nullptr;
#pragma unroll
  for (int i = 0; i < RM; ++i) {
#pragma unroll
    for (int j = 0; j < RN; ++j) {
      const int n = n0 + nw + j * 32 + l31;
      if (n >= p.N) continue;
      float bn = 0.f;
      if (!partial && p.bias) bn = p.bias[n];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (m >= p.M) continue;
        float v = acc[i][j][r];
        if (!partial) {
          v += bn;
          if (p.bias2) v += p.bias2[(long)(m / p.rpg) * p.N + n];
          v *= p.alpha;
          if (R) v += p.beta * R[(long)m * p.ldr + n];
        }
        C[(long)m * ldc + n] = v;
      }
    }
  }
}

// Deterministic split-K combine + epilogue.
__global__ void igemm_splitk_reduce(IgemmParams p) {
  const long total = (long)p.batch * p.M * p.N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long mn = (long)p.M * p.N;
    const int bz = (int)(i / mn);
    const long r = i - (long)bz * mn;
    const int m = (int)(r / p.N), n = (int)(r - (long)m * p.N);
    float s = 0.f;
    for (int kz = 0; kz < p.splitk; ++kz) s += p.ws[((long)(bz * p.splitk + kz) * p.M + m) * p.N + n];
    float v = s;
    if (p.bias) v += p.bias[n];
    if (p.bias2) v += p.bias2[(long)(m / p.rpg) * p.N + n];
    v *= p.alpha;
    if (p.R) v += p.beta * p.R[(long)bz * p.sR + (long)m * p.ldr + n];
    p.C[(long)bz * p.sC + (long)m * p.ldc + n] = v;
  }
}

template <int AM>
inline bool use_tall_tile(const IgemmParams& p) {
  if constexpr (AM == A_ROW || AM == A_CONV) {
    return p.splitk == 1 && p.N >= 128 && (long)cdiv(p.M, 256) * cdiv(p.N, 128) * p.batch >= 256;
  } else {
    return false;
  }
}

template <int AM, int BMODE>
int launch_tiles(const IgemmParams& p, hipStream_t st) {
  // tile choice: skinny N (flow 3/12-channel edges, last score conv) -> 128x32;
  // skinny M (27-/108-row weight grads) -> 32x128; everything else 128x128.
  dim3 block(256);
  if (p.N <= 32 && p.M > 32) {
    dim3 grid(cdiv(p.M, 128) * cdiv(p.N, 32), p.batch * p.splitk);
    hipLaunchKernelGGL((igemm_kernel<AM, BMODE, 128, 32, 4, 1>), grid, block, 0, st, p);
  } else if (p.M <= 128 && cdiv(p.M, 32) * cdiv(p.N, 128) * p.batch * p.splitk <= 512) {
    // short-M GEMMs (per-image vectors x weights): 32-row tiles so more than a handful of CUs get work
    dim3 grid(cdiv(p.M, 32) * cdiv(p.N, 128), p.batch * p.splitk);
    hipLaunchKernelGGL((igemm_kernel<AM, BMODE, 32, 128, 1, 4>), grid, block, 0, st, p);
  } else if (use_tall_tile<AM>(p)) {
    // 8-wave 256x128 tile: +8-10 % on the long-M forward / data-grad contractions (measured)
    dim3 grid(cdiv(p.M, 256) * cdiv(p.N, 128), p.batch * p.splitk);
    hipLaunchKernelGGL((igemm_kernel<AM, BMODE, 256, 128, 4, 2>), grid, dim3(512), 0, st, p);
  } else {
    dim3 grid(cdiv(p.M, 128) * cdiv(p.N, 128), p.batch * p.splitk);
    hipLaunchKernelGGL((igemm_kernel<AM, BMODE, 128, 128, 2, 2>), grid, block, 0, st, p);
  }
  if (hipGetLastError() != hipSuccess) return INDM_E_LAUNCH;
  if (p.splitk > 1) {
    const long total = (long)p.batch * p.M * p.N;
    hipLaunchKernelGGL(igemm_splitk_reduce, dim3(ew_grid(total, 256)), dim3(256), 0, st, p);
    if (hipGetLastError() != hipSuccess) return INDM_E_LAUNCH;
  }
  return INDM_OK;
}

inline bool al16(const void* q) { return ((uintptr_t)q & 15) == 0; }

// Pick split-K so that tiles*splits fills the chip (>= ~2 waves of 256 CUs) when the
// output is small and the reduction long (weight grads: K = B*H*W pixels).
int choose_splitk(int M, int N, int K, int batch, long ws_floats) {
  long tiles = (long)cdiv(M, (N <= 32 && M > 32) ? 128 : (M <= 128 ? 32 : 128)) *
               cdiv(N, (N <= 32 && M > 32) ? 32 : 128) * batch;
  if (tiles >= 256 || K < 4 * BK * 8) return 1;
  static const long target = getenv("INDM_SPLITK_TARGET") ? atol(getenv("INDM_SPLITK_TARGET")) : 1024;
  long want = (target + tiles - 1) / tiles;
  long maxs = K / (BK * 8);
  if (want > maxs) want = maxs;
  static const long cap = getenv("INDM_SPLITK_MAX") ? atol(getenv("INDM_SPLITK_MAX")) : 256;
  if (want > cap) want = cap;
  while (want > 1 && (long)want * batch * M * N > ws_floats) --want;
  return want < 1 ? 1 : (int)want;
}

int run(IgemmParams& p, int am, int bm, float* ws, long ws_floats, hipStream_t st) {
  if (p.M <= 0 || p.N <= 0 || p.batch <= 0) return INDM_OK;
  if (p.K < 0) return INDM_E_SHAPE;
  p.splitk = ws ? choose_splitk(p.M, p.N, p.K, p.batch, ws_floats) : 1;
  p.ws = ws;
  int kps = cdiv(p.K > 0 ? p.K : 1, p.splitk);
  kps = cdiv(kps, BK) * BK;
  p.kps = kps;
  p.splitk = cdiv(p.K > 0 ? p.K : 1, kps);
#define DISPATCH(AMV, BMV) \
  if (am == AMV && bm == BMV) return launch_tiles<AMV, BMV>(p, st);
  DISPATCH(A_ROW, B_ROW)
  DISPATCH(A_ROW, B_COL)
  DISPATCH(A_COL, B_ROW)
  DISPATCH(A_COL, B_COL)
  DISPATCH(A_CONV, B_ROW)
  DISPATCH(A_CONV, B_DGRAD)
  DISPATCH(A_WGRAD, B_ROW)
#undef DISPATCH
  return INDM_E_ARG;
}

}  // namespace

// ---------------------------------------------------------------------------
// C ABI (declared in include/indm_hip.h)
// ---------------------------------------------------------------------------

// C[b] = alpha * (opA(A[b] .* gateA[b]) * opB(B[b]) + bias[n] + bias2[m/rpg][n]) + beta * R[b]   (gateA optional)
//   transA == 0: A is [M,K] row-major (lda);  transA == 1: A is stored [K,M] (lda)
//   transB == 0: B is [K,N] row-major (ldb);  transB == 1: B is stored [N,K] (ldb)
INDM_API int indm_gemm_f32(const float* A, const float* gateA, const float* B, float* C, int M, int N, int K, int transA,
                           int transB, long lda, long ldb, long ldc, int batch, long strideA,
                           long strideB, long strideC, float alpha, const float* bias,
                           const float* bias2, int rows_per_group, const float* R, long ldr,
                           long strideR, float beta, float* ws, long ws_floats, void* stream) {
  if (M < 0 || N < 0 || K < 0 || batch < 0) return INDM_E_SHAPE;
  IgemmParams p = {};
  p.A = A; p.G = gateA; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.batch = batch; p.sA = strideA; p.sB = strideB; p.sC = strideC;
  p.alpha = alpha; p.beta = beta; p.bias = bias; p.bias2 = bias2;
  p.rpg = rows_per_group > 0 ? rows_per_group : 1;
  p.R = R; p.ldr = ldr; p.sR = strideR;
  p.a_vec = al16(A) && al16(gateA) && (lda % 4 == 0) && (strideA % 4 == 0);
  p.b_vec = al16(B) && (ldb % 4 == 0) && (strideB % 4 == 0);
  return run(p, transA ? A_COL : A_ROW, transB ? B_COL : B_ROW, ws, ws_floats, (hipStream_t)stream);
}

// NHWC convolution forward:  y[n,oh,ow,co] = sum x[n, oh*s+kh-p, ow*s+kw-p, ci] * w[kh,kw,ci,co]
// (cross-correlation, same as torch conv2d). `in_dil` > 1 treats x as zero-inserted
// (used for the data-grad of strided convolutions). Weight layout is RSCK = [KH,KW,Cin,Cout].
INDM_API int indm_conv2d_fwd_f32(const float* x, const float* w, float* y, int Nimg, int H, int W,
                                 int Cin, int Cout, int KH, int KW, int stride, int pad, int in_dil,
                                 int OH, int OW, float alpha, const float* bias, const float* bias2,
                                 const float* R, float beta, float* ws, long ws_floats, void* stream) {
  if (Nimg < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || OH <= 0 || OW <= 0) return INDM_E_SHAPE;
  IgemmParams p = {};
  p.A = x; p.B = w; p.C = y;
  p.M = Nimg * OH * OW; p.N = Cout; p.K = KH * KW * Cin;
  p.ldb = Cout; p.ldc = Cout; p.batch = 1;
  p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.KH = KH; p.KW = KW;
  p.stride = stride; p.pad = pad; p.dil = in_dil > 0 ? in_dil : 1;
  p.alpha = alpha; p.beta = beta; p.bias = bias; p.bias2 = bias2; p.rpg = OH * OW;
  p.R = R; p.ldr = Cout;
  p.a_vec = al16(x) && (Cin % 4 == 0);
  p.b_vec = al16(w) && (Cout % 4 == 0);
  p.a_tap_uniform = (Cin % BK == 0);
  return run(p, A_CONV, B_ROW, ws, ws_floats, (hipStream_t)stream);
}

// Data-grad of a stride-1 convolution (and, with in_dil = forward stride, of a strided one):
//   dx[n,ih,iw,ci] = sum_{kh,kw,co} dy[n, ih+pad'-..] * w[KH-1-kh, KW-1-kw, ci, co]
// dy: [Nimg,H,W,Cout] (H,W = dy spatial), dx: [Nimg,OH,OW,Cin].  pad' = K-1-pad_fwd.
INDM_API int indm_conv2d_dgrad_f32(const float* dy, const float* w, float* dx, int Nimg, int H, int W,
                                   int Cin, int Cout, int KH, int KW, int pad_t, int in_dil, int OH,
                                   int OW, float alpha, const float* R, float beta, float* ws, long ws_floats,
                                   void* stream) {
  if (Nimg < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || OH <= 0 || OW <= 0) return INDM_E_SHAPE;
  IgemmParams p = {};
  p.A = dy; p.B = w; p.C = dx;
  p.M = Nimg * OH * OW; p.N = Cin; p.K = KH * KW * Cout;
  p.ldc = Cin; p.batch = 1;
  p.H = H; p.W = W; p.Cin = Cout; p.Cout = Cout; p.OH = OH; p.OW = OW; p.KH = KH; p.KW = KW;
  p.stride = 1; p.pad = pad_t; p.dil = in_dil > 0 ? in_dil : 1;
  p.alpha = alpha; p.beta = beta; p.rpg = 1; p.R = R; p.ldr = Cin;
  p.a_vec = al16(dy) && (Cout % 4 == 0);
  p.b_vec = al16(w) && (Cout % 4 == 0);
  p.a_tap_uniform = p.b_tap_uniform = (Cout % BK == 0);
  return run(p, A_CONV, B_DGRAD, ws, ws_floats, (hipStream_t)stream);
}

// Weight-grad: dw[kh,kw,ci,co] = alpha * sum_{n,oh,ow} x[n, oh*s+kh-p, ow*s+kw-p, ci] * dy[n,oh,ow,co]
//              (+ beta * R, same layout).  Reduction over pixels is split-K into `ws`.
INDM_API int indm_conv2d_wgrad_f32(const float* x, const float* dy, float* dw, int Nimg, int H, int W,
                                   int Cin, int Cout, int KH, int KW, int stride, int pad, int OH,
                                   int OW, float alpha, const float* R, float beta, float* ws,
                                   long ws_floats, void* stream) {
  if (Nimg < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || OH <= 0 || OW <= 0) return INDM_E_SHAPE;
  IgemmParams p = {};
  p.A = x; p.B = dy; p.C = dw;
  p.M = KH * KW * Cin; p.N = Cout; p.K = Nimg * OH * OW;
  p.ldb = Cout; p.ldc = Cout; p.batch = 1;
  p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.KH = KH; p.KW = KW;
  p.stride = stride; p.pad = pad; p.dil = 1;
  p.alpha = alpha; p.beta = beta; p.rpg = 1; p.R = R; p.ldr = Cout;
  p.a_vec = al16(x) && (Cin % 4 == 0);
  p.b_vec = al16(dy) && (Cout % 4 == 0);
  return run(p, A_WGRAD, B_ROW, ws, ws_floats, (hipStream_t)stream);
}

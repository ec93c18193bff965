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
// the direct-to-LDS copy helpers write M0 from inline asm and say so in their clobber lists (M0 is a reserved register:
// hipcc warns about the clobber, which is exactly the declaration its M0-initialisation merging needs to see)
#pragma clang diagnostic ignored "-Winline-asm"
#include "igemm_thin.h"
#include <type_traits>
#include <cstdlib>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

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
  int wg_inc;                        // A_WGRAD: pixel decode advanced incrementally (needs OH*OW >= BK)
  int ow_sh, ohw_sh;                 // log2(OW), log2(OH*OW) when both are powers of two, else -1
  int a_buf;                         // A_CONV on the direct-to-LDS kernel: buffer-addressed copies (dma_buffer)
  int bf16;                          // MFMA inputs rounded to bf16 (RNE), fp32 accumulate (BASELINE configs #3-#5)
  float* act_d;                      // Sin epilogue (indm_gemm_sin_f32): C receives sin(2 pi v) / (2 pi), act_d cos(2 pi v) ([M, ldc])
  // Sin-level backward as the epilogue (indm_gemm_ires_bwd_act_f32; all [M, ldc]): with da = this GEMM's result (never written)
  //   C = dp = da * d - 4 pi^2 * a * (du * t),   e_dt = du * d (optional),
  //   e_part[(m / 64) * N + n] = sum of dp over the 64-row chunk in row order (optional; indm_ires_bwd_act_f32's partial sums)
  const float *e_du, *e_d, *e_a, *e_t;
  float *e_dt, *e_part;
  // the same pass on the GEMM that produces du instead (indm_gemm_gated_ires_bwd_act_f32): C = du = this GEMM's result (written:
  // other consumers need it), e_da = the incoming da (read), e_dp = dp (written); e_d, e_a, e_t, e_part as above
  const float* e_da;
  float* e_dp;
};
// (the literals of elementwise.hip: the fused Sin epilogue must round like indm_sincos_act_fwd_f32)
constexpr float SIN_TWO_PI = 6.283185307179586f;
constexpr float SIN_INV_TWO_PI = 0.15915494309189535f;

// name of the kernel variant the last launch of this thread went to (bench.py labels its live timings with it)
thread_local const char* last_variant = "";

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
      // the caller tracks (img, oh, ow) of this thread's reduction pixel incrementally (tapdec); see gload
      PixDec d;
      if (p.wg_inc) {
        d.img = tapdec.x; d.oh = tapdec.y; d.ow = tapdec.z;
      } else {
        d = decode_pix(k, p.OH, p.OW);
      }
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

// BF16: the "bf16 compute / fp32 accumulate" mode of BASELINE configs #3-#5.  Tensors stay fp32 in HBM and in LDS;
// a fragment is 8 k-values per lane (k = 8*(lane/32) + t), rounded to bf16 with v_cvt_pk_bf16_f32 after the LDS read,
// and ONE v_mfma_f32_32x32x16_bf16 per 16-deep slab replaces eight fp32 MFMAs (16x the MFMA rate; the kernel is then
// bound by its operand traffic, not by the matrix pipe).
// General epilogue of one 32 x 32 accumulator block (residual R and / or per-image bias2 present, or a ragged tile): the 16 bias2
// and the 16 residual values of the lane are requested TOGETHER, from clamped addresses, before the first of them is used.
// Written as `if (m < M) { v += bias2[..]; v += beta * R[..]; store }` per element, every one of those loads became a branch
// with `s_waitcnt vmcnt(0)` behind it (profiles/r04_isa_serial_load_census.txt): 32 memory latencies per block, one after the
// other.  Same operations in the same order per element (R may alias C: a lane reads exactly the elements it then writes).
template <class Acc, bool ACT = false>
__device__ __forceinline__ void epilogue_block(const IgemmParams& p, const Acc& acc, float* C, long ldc, const float* R,
                                               bool partial, int mbase, int n, int lhi) {
  __builtin_amdgcn_sched_barrier(0);   // one block at a time: hipcc otherwise hoists every block's loads to the top and spills
  const bool nok = n < p.N;
  float v[16];
  unsigned ok = 0;                     // bit r: element r of this lane is inside the matrix
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    v[r] = acc[r];
    ok |= (nok && mbase + (r & 3) + 8 * (r >> 2) + 4 * lhi < p.M) ? (1u << r) : 0u;
  }
  if (!partial) {
    float bn = 0.f;
    if (p.bias) bn = p.bias[nok ? n : 0];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] += bn;
    if (p.bias2) {   // 32-bit element offsets from the uniform base (scalar-base addressing: one register per address)
      float t[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mbase + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        t[r] = p.bias2[((ok >> r) & 1) ? (unsigned)(m / p.rpg) * (unsigned)p.N + (unsigned)n : 0u];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] += t[r];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] *= p.alpha;
    if (R) {
      float t[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mbase + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        t[r] = R[((ok >> r) & 1) ? (unsigned)m * (unsigned)p.ldr + (unsigned)n : 0u];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] += p.beta * t[r];
    }
  }
  if constexpr (ACT) {   // (a, d) = Sin(v): v is what the plain epilogue would have written, so both outputs equal the two-step form's
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = mbase + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      float sn, cs;
      indm_sincos(SIN_TWO_PI * v[r], sn, cs);
      if ((ok >> r) & 1) {
        C[(long)m * ldc + n] = sn * SIN_INV_TWO_PI;
        p.act_d[(long)m * ldc + n] = cs;
      }
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = mbase + (r & 3) + 8 * (r >> 2) + 4 * lhi;
    if ((ok >> r) & 1) C[(long)m * ldc + n] = v[r];
  }
}

// One 32 x 32 block of indm_gemm_ires_bwd_act_f32's epilogue: `da` = the block's GEMM result (lane: column n, rows
// mbase + 8 q + 4 lhi + e), the four operand tensors read at the same elements (16 loads in flight per quarter), dp / dt
// written, and `run` = the column's running sum of dp over the rows so far IN ROW ORDER -- rows 8 q .. 8 q + 3 live in lane
// half 0, 8 q + 4 .. 8 q + 7 in half 1, so the sum crosses the halves twice per quarter (ds_bpermute): the same additions
// in the same order as ires_bwd_act_kernel's `s += g` over its 64-row chunk (bit-identical partial sums).
// DU_ACC: the block is du (stored to C), da is read from p.e_da and dp goes to p.e_dp -- same expressions.
template <bool DU_ACC>
__device__ __forceinline__ void epilogue_bwd_act_block(const IgemmParams& p, const f32x16& da, int mbase, int n, int lane,
                                                       float& run) {
  const float K = SIN_TWO_PI * SIN_TWO_PI;
  const int lhi = lane >> 5;
  const bool nok = n < p.N;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    __builtin_amdgcn_sched_barrier(0);   // one quarter's 16 loads at a time
    float uu[4], dd[4], aa[4], tt[4], g[4];
    bool ok[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int m = mbase + 8 * q + 4 * lhi + e;
      ok[e] = nok && m < p.M;
      const unsigned o = ok[e] ? (unsigned)m * (unsigned)p.ldc + (unsigned)n : 0u;
      uu[e] = DU_ACC ? p.e_da[o] : p.e_du[o];   // (DU_ACC: this slot holds da)
      dd[e] = p.e_d[o]; aa[e] = p.e_a[o]; tt[e] = p.e_t[o];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int m = mbase + 8 * q + 4 * lhi + e;
      const float dav = DU_ACC ? uu[e] : da[4 * q + e], duv = DU_ACC ? da[4 * q + e] : uu[e];
      g[e] = dav * dd[e] - K * aa[e] * (duv * tt[e]);
      if (ok[e]) {
        const unsigned o = (unsigned)m * (unsigned)p.ldc + (unsigned)n;
        if constexpr (DU_ACC) {
          p.C[o] = duv;
          p.e_dp[o] = g[e];
        } else {
          p.C[o] = g[e];
          if (p.e_dt) p.e_dt[o] = duv * dd[e];
        }
      }
    }
    if (p.e_part) {
      float sa = run;                                    // meaningful in half 0: rows 8 q .. 8 q + 3
#pragma unroll
      for (int e = 0; e < 4; ++e) sa = ok[e] ? sa + g[e] : sa;
      const float from0 = __shfl_xor(sa, 32, 64);
      float sb = lhi ? from0 : 0.f;                      // half 1 continues from half 0's sum: rows 8 q + 4 .. 8 q + 7
#pragma unroll
      for (int e = 0; e < 4; ++e) sb = ok[e] ? sb + g[e] : sb;
      const float from1 = __shfl_xor(sb, 32, 64);
      run = lhi ? sb : from1;                            // both halves now hold the sum over rows .. 8 q + 7
    }
  }
}

template <int AM, int BMODE, int TM, int TN, int WAVES_M, int WAVES_N, bool BF16 = false>
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

  // A_WGRAD: the reduction index is an output pixel; its (img, oh, ow) decode costs two integer divisions per
  // load.  gload() is called for consecutive slabs, so each thread keeps the decode of its pixel and steps it by
  // BK pixels per call (BK = sa*OW + sb; at most one carry each because OH*OW >= BK).
  PixDec kdec[NGA];
  const int wg_sa = (AM == A_WGRAD) ? BK / p.OW : 0, wg_sb = (AM == A_WGRAD) ? BK % p.OW : 0;
  if constexpr (AM == A_WGRAD) {
#pragma unroll
    for (int j = 0; j < NGA; ++j) kdec[j] = decode_pix(kbeg + (tid + j * NT) / (TM / 4), p.OH, p.OW);
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
          if constexpr (AM == A_WGRAD) {
            tapdec = make_int3(kdec[j].img, kdec[j].oh, kdec[j].ow);
            kdec[j].ow += wg_sb;
            kdec[j].oh += wg_sa;
            if (kdec[j].ow >= p.OW) { kdec[j].ow -= p.OW; ++kdec[j].oh; }
            if (kdec[j].oh >= p.OH) { kdec[j].oh -= p.OH; ++kdec[j].img; }
          }
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
    if constexpr (BF16) {
      static_assert(BK == 16, "one bf16 MFMA k-step per slab");
      bf16x8 fa[RM], fb[RN];
#pragma unroll
      for (int i = 0; i < RM; ++i)
        {
          const float* q = as + 8 * lhi * LDA_S + i * 32;
          fa[i] = indm_pack_bf16x8(q[0], q[LDA_S], q[2 * LDA_S], q[3 * LDA_S], q[4 * LDA_S], q[5 * LDA_S], q[6 * LDA_S], q[7 * LDA_S]);
        }
#pragma unroll
      for (int j = 0; j < RN; ++j)
        {
          const float* q = bs + 8 * lhi * LDB_S + j * 32;
          fb[j] = indm_pack_bf16x8(q[0], q[LDB_S], q[2 * LDB_S], q[3 * LDB_S], q[4 * LDB_S], q[5 * LDB_S], q[6 * LDB_S], q[7 * LDB_S]);
        }
      mid();
#pragma unroll
      for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      return;
    }
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
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j)
      epilogue_block(p, acc[i][j], C, ldc, R, partial, m0 + mw + i * 32, n0 + nw + j * 32 + l31, lhi);
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// ---- loaders of the big-tile kernel ---------------------------------------------------------------------------
// Preconditions (checked on the host, big_tile_n): a_vec && b_vec, K % 16 == 0, conv reduction slabs never straddle
// a tap.  Rows / columns beyond M / N are CLAMPED, not zeroed: they only feed accumulators whose results the
// epilogue never stores, so the GEMM modes load unconditionally (no select that would drag the vmcnt wait up to
// the load).  Only the zero padding of a convolution needs a predicate; it is an exec-masked load.
template <int AM>
__device__ __forceinline__ long a_off(const IgemmParams& p, int row, int k, const PixDec& rd, const int3& tapdec,
                                      int kbase, bool& ok) {
  ok = true;
  if constexpr (AM == A_ROW) {
    return (long)min(row, p.M - 1) * p.lda + k;
  } else if constexpr (AM == A_COL) {
    return (long)k * p.lda + min(row, p.M - 4);
  } else if constexpr (AM == A_CONV) {
    int ih = rd.oh * p.stride + tapdec.x - p.pad, iw = rd.ow * p.stride + tapdec.y - p.pad;
    ok = ih >= 0 && iw >= 0;
    if (p.dil > 1) {
      ok = ok && ((ih % p.dil) | (iw % p.dil)) == 0;
      ih /= p.dil;
      iw /= p.dil;
    }
    ok = ok && ih < p.H && iw < p.W;
    return (((long)rd.img * p.H + ih) * p.W + iw) * p.Cin + tapdec.z + (k - kbase);
  } else {
    PixDec d;
    if (p.ow_sh >= 0) {  // OW and OH*OW powers of two (every map of the INDM nets): shifts instead of two divisions
      d.img = k >> p.ohw_sh;
      const int r = k & ((1 << p.ohw_sh) - 1);
      d.oh = r >> p.ow_sh;
      d.ow = r & ((1 << p.ow_sh) - 1);
    } else {
      d = decode_pix(k, p.OH, p.OW);
    }
    const int ih = d.oh * p.stride + rd.img - p.pad, iw = d.ow * p.stride + rd.oh - p.pad;
    ok = ih >= 0 && iw >= 0 && ih < p.H && iw < p.W;
    return (((long)d.img * p.H + ih) * p.W + iw) * p.Cin + rd.ow;
  }
}

template <int BMODE>
__device__ __forceinline__ float4 b_fast(const IgemmParams& p, const float* B, int row, int k, const int2& btap,
                                         int kbase) {
  long off;
  if constexpr (BMODE == B_ROW) {
    off = (long)k * p.ldb + min(row, p.N - 4);
  } else if constexpr (BMODE == B_COL) {
    off = (long)min(row, p.N - 1) * p.ldb + k;
  } else {
    off = ((long)(p.KH * p.KW - 1 - btap.x) * p.N + min(row, p.N - 1)) * p.Cout + btap.y + (k - kbase);
  }
  return *(const float4*)(B + off);
}

// ---------------------------------------------------------------------------------------------------------------
// "Big tile" kernel for the long contractions (M*N large, no split-K): ONE wavefront per SIMD owning a 128 x 128
// (or 128 x 64) register tile -- 256 (128) accumulator registers out of the 512-entry unified VGPR/AGPR file --
// so one pass over a 16-deep LDS slab feeds 128 (64) MFMAs per wave between barriers instead of 32.  Four waves
// (2 x 2) make a 256 x 256 (256 x 128) block tile; operand traffic per FLOP drops 1.33x vs the 8-wave 256 x 128
// kernel and the LDS-read / barrier / global-load instruction counts per MFMA drop 2-4x.
//
// LDS images (double-buffered, every store is one ds_write_b128, every layout conflict-free):
//   * k-contiguous operand (A_ROW, A_CONV, B_COL, B_DGRAD): [row][16 k], the four 16-byte k-chunks of a row
//     XOR-swizzled with (row>>2)&3.  Fragments are read with ds_read_b128: lane (r, h) takes k-chunk 2*sg+h of
//     row r and feeds component t of it to the t-th MFMA of k-group sg.
//   * row-contiguous operand (A_COL, A_WGRAD, B_ROW): [k][rows+4], fragments by ds_read_b32 of row
//     8*sg + 4*h + t -- the SAME k as the k-contiguous operand supplies in that MFMA.
// The MFMA only sums over k, so any bijection lane-half/step -> k is valid as long as A and B agree; that frees
// the k-contiguous operands from transposing LDS writes.
// BF16 = true ("bf16 compute / fp32 accumulate"): same staging; per slab both 4-deep fragments of a lane are rounded
// and packed into one v_mfma_f32_32x32x16_bf16 per 32 x 32 block.  The kernel is then HBM-bound, so the slab is issued
// in plain order (fetch s+2, read fragments, RM*RN MFMAs, store s+1, barrier) instead of the fp32 weave.
template <int AM, int BMODE, int TN, bool HAS_G, bool BF16 = false>
__global__ __launch_bounds__(256, (TN == 256 ? 1 : 2)) void igemm_big_kernel(IgemmParams p) {
  static_assert(BK == 16, "big-tile kernel is written for 16-deep slabs");
  constexpr int TM = 256, NT = 256;
  constexpr int WN = TN / 2;
  constexpr int RM = 4, RN = WN / 32;
  constexpr bool AK = a_kcontig<AM>(), BKC = b_kcontig<BMODE>();
  constexpr int LDA_R = TM + 4, LDB_R = TN + 4;
  constexpr int A_SZ = AK ? TM * BK : BK * LDA_R;
  constexpr int B_SZ = BKC ? TN * BK : BK * LDB_R;
  constexpr int NGA = TM * 4 / NT, NGB = TN * 4 / NT;  // float4 groups per thread per slab
  constexpr int NMFMA = 4 * RM * RN;                   // MFMAs of one 8-deep k-group
  constexpr int NITEMS = RM + RN + NGA + NGB;          // side-work items woven into one k-group
  constexpr int STEP = NMFMA / NITEMS;
  static_assert(STEP >= 1, "not enough MFMAs to hide the side work");
  __shared__ __attribute__((aligned(16))) float smem[2 * (A_SZ + B_SZ)];
  float* As = smem;
  float* Bs = smem + 2 * A_SZ;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_m = (p.M + TM - 1) / TM, tiles_n = (p.N + TN - 1) / TN;
  const int lt = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int tm = lt / tiles_n, tn = lt - tm * tiles_n;
  const int m0 = tm * TM, n0 = tn * TN;
  const int z = blockIdx.y;
  const int bz = z / p.splitk, kz = z - bz * p.splitk;
  const float* A = p.A + (long)bz * p.sA;
  const float* G = HAS_G ? p.G + (long)bz * p.sA : nullptr;
  const float* B = p.B + (long)bz * p.sB;
  const int kbeg = kz * p.kps;                         // split-K: this block reduces over [kbeg, kbeg + kps)
  const int nk = (min(p.K, kbeg + p.kps) - kbeg) / BK;

  f32x16 acc[RM][RN];
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[2][NGA], rb[2][NGB], rg[2][HAS_G ? NGA : 1];  // two prefetch sets: slab s lives in set s&1
  float4 fa[2][RM], fb[2][RN];                             // fragments of k-group 0 / 1
  unsigned okmask[2] = {0u, 0u};                           // conv zero-padding predicates of the two prefetch sets

  PixDec rowdec[NGA];
#pragma unroll
  for (int j = 0; j < NGA; ++j) {
    const int g = tid + j * NT;
    rowdec[j].img = rowdec[j].oh = rowdec[j].ow = 0;
    if constexpr (AM == A_CONV) {
      const int row = m0 + g / 4;
      if (row < p.M) rowdec[j] = decode_pix(row, p.OH, p.OW);
    } else if constexpr (AM == A_WGRAD) {
      const int row = m0 + (g % (TM / 4)) * 4;
      if (row < p.M) {
        const int tap = row / p.Cin;
        rowdec[j].ow = row - tap * p.Cin;
        rowdec[j].img = tap / p.KW;
        rowdec[j].oh = tap - rowdec[j].img * p.KW;
      }
    }
  }

  // ---- side-work items: each is one memory instruction plus its address arithmetic --------------------------
  auto gl_a = [&](auto SET, int j, int kt) {  // global -> register set SET: A group j of slab kt
    const int k0 = kbeg + kt * BK;
    int3 tapdec = make_int3(0, 0, 0);
    if constexpr (AM == A_CONV) {
      const int tap = k0 / p.Cin;
      tapdec.z = k0 - tap * p.Cin;
      tapdec.x = tap / p.KW;
      tapdec.y = tap - tapdec.x * p.KW;
    }
    const int g = tid + j * NT;
    bool ok;
    long off;
    if constexpr (AK) {
      off = a_off<AM>(p, m0 + g / 4, k0 + (g % 4) * 4, rowdec[j], tapdec, k0, ok);
    } else {
      constexpr int Q = TM / 4;
      off = a_off<AM>(p, m0 + (g % Q) * 4, k0 + g / Q, rowdec[j], tapdec, k0, ok);
    }
    if constexpr (AM == A_CONV || AM == A_WGRAD) {
      // zero padding: fetch offset 0 instead and remember the predicate; the select happens at the LDS store, so no
      // branch splits the MFMA stream and no wait is pulled up to the load
      off = ok ? off : 0;
      okmask[SET] = (okmask[SET] & ~(1u << j)) | ((ok ? 1u : 0u) << j);
    }
    ra[SET][j] = *(const float4*)(A + off);
    if constexpr (HAS_G) rg[SET][j] = *(const float4*)(G + off);
  };
  auto gl_b = [&](auto SET, int j, int kt) {
    const int k0 = kbeg + kt * BK;
    int2 btap = make_int2(0, 0);
    if constexpr (BMODE == B_DGRAD) {
      btap.x = k0 / p.Cout;
      btap.y = k0 - btap.x * p.Cout;
    }
    const int g = tid + j * NT;
    if constexpr (BKC) {
      rb[SET][j] = b_fast<BMODE>(p, B, n0 + g / 4, k0 + (g % 4) * 4, btap, k0);
    } else {
      constexpr int Q = TN / 4;
      rb[SET][j] = b_fast<BMODE>(p, B, n0 + (g % Q) * 4, k0 + g / Q, btap, k0);
    }
  };
  auto st_a = [&](auto BUF, auto SET, int j) {  // register set SET -> LDS buffer BUF (gate multiplied in here)
    float* as = As + BUF * A_SZ;
    const int g = tid + j * NT;
    float4 v = ra[SET][j];
    if constexpr (AM == A_CONV || AM == A_WGRAD) {
      if (!((okmask[SET] >> j) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if constexpr (HAS_G) {
      const float4 q = rg[SET][j];
      v.x *= q.x; v.y *= q.y; v.z *= q.z; v.w *= q.w;
    }
    if constexpr (AK) {
      const int row = g / 4, q = g % 4;
      *(float4*)(as + row * BK + ((q ^ ((row >> 2) & 3)) << 2)) = v;
    } else {
      constexpr int Q = TM / 4;
      *(float4*)(as + (g / Q) * LDA_R + (g % Q) * 4) = v;
    }
  };
  auto st_b = [&](auto BUF, auto SET, int j) {
    float* bs = Bs + BUF * B_SZ;
    const int g = tid + j * NT;
    if constexpr (BKC) {
      const int row = g / 4, q = g % 4;
      *(float4*)(bs + row * BK + ((q ^ ((row >> 2) & 3)) << 2)) = rb[SET][j];
    } else {
      constexpr int Q = TN / 4;
      *(float4*)(bs + (g / Q) * LDB_R + (g % Q) * 4) = rb[SET][j];
    }
  };

  const int wm = wave >> 1, wn = wave & 1;
  const int mw = wm * 128, nw = wn * WN;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int swz = (l31 >> 2) & 3;  // (row>>2)&3 of every fragment row this lane reads (tile bases are multiples of 32)

  // fragments of the 8-deep k-group SG of LDS buffer BUF: component t feeds the t-th MFMA of the group
  auto fr_a = [&](auto BUF, auto SG, int i) {
    const float* as = As + BUF * A_SZ;
    if constexpr (AK) {
      fa[SG][i] = *(const float4*)(as + (mw + i * 32 + l31) * BK + (((SG * 2 + lhi) ^ swz) << 2));
    } else {
      const float* q = as + (SG * 8 + lhi * 4) * LDA_R + mw + i * 32 + l31;
      fa[SG][i] = make_float4(q[0], q[LDA_R], q[2 * LDA_R], q[3 * LDA_R]);
    }
  };
  auto fr_b = [&](auto BUF, auto SG, int j) {
    const float* bs = Bs + BUF * B_SZ;
    if constexpr (BKC) {
      fb[SG][j] = *(const float4*)(bs + (nw + j * 32 + l31) * BK + (((SG * 2 + lhi) ^ swz) << 2));
    } else {
      const float* q = bs + (SG * 8 + lhi * 4) * LDB_R + nw + j * 32 + l31;
      fb[SG][j] = make_float4(q[0], q[LDB_R], q[2 * LDB_R], q[3 * LDB_R]);
    }
  };
  auto comp = [](const float4& v, int t) { return t == 0 ? v.x : t == 1 ? v.y : t == 2 ? v.z : v.w; };

  // One k-group of MFMAs on fragment set SG with `side(n)`, n = 0..NITEMS-1, woven in every STEP MFMAs.  With one
  // wave per SIMD nobody else fills issue slots, so memory instructions must sit BETWEEN MFMAs (15 free issue
  // slots under each 64-cycle MFMA), never in a burst after them; sched_barrier(0) pins the source order.
  auto kgroup = [&](auto SG, auto&& side) {
    static_for<0, NMFMA>([&](auto Q) {
      constexpr int q = decltype(Q)::value;
      constexpr int t = q / (RM * RN), i = (q / RN) % RM, j = q % RN;
      const float a = comp(fa[SG][i], t), b = comp(fb[SG][j], t);
      if constexpr (TN == 256) {
        // 256 accumulators only fit the AGPR half of the register file: pin them there (hipcc would otherwise pick
        // the VGPR-destination MFMA form and shuttle every accumulator through v_accvgpr_* each slab)
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(a), "v"(b));
      } else {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0);
      }
      if constexpr ((q + 1) % STEP == 0 && (q + 1) / STEP <= NITEMS) {
        side(std::integral_constant<int, (q + 1) / STEP - 1>{});
        __builtin_amdgcn_sched_barrier(0);
      }
    });
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // Slab s (LDS buffer PAR = s&1):
  //   k-group A: MFMAs on k-group 1 of slab s-1 | reads k-group 0 of slab s, fetches slab s+2 into set PAR
  //   k-group B: MFMAs on k-group 0 of slab s   | reads k-group 1 of slab s, stores slab s+1 (set 1-PAR) to LDS
  //   barrier
  // Past the end of K the fetches are clamped to the last slab (valid memory, the copy is stored to the LDS buffer
  // nobody reads any more), so the loop body has no tail variants.
  auto slab = [&](auto PAR, int s) {
    using OTH = std::integral_constant<int, 1 - decltype(PAR)::value>;
    const int kt = min(s + 2, nk - 1);
    kgroup(I1{}, [&](auto NC) {
      constexpr int n = decltype(NC)::value;
      if constexpr (n < RM) fr_a(PAR, I0{}, n);
      else if constexpr (n < RM + RN) fr_b(PAR, I0{}, n - RM);
      else if constexpr (n < RM + RN + NGA) gl_a(PAR, n - RM - RN, kt);
      else gl_b(PAR, n - RM - RN - NGA, kt);
    });
    kgroup(I0{}, [&](auto NC) {
      constexpr int n = decltype(NC)::value;
      if constexpr (n < RM) fr_a(PAR, I1{}, n);
      else if constexpr (n < RM + RN) fr_b(PAR, I1{}, n - RM);
      else if constexpr (n < RM + RN + NGA) st_a(OTH{}, OTH{}, n - RM - RN);
      else st_b(OTH{}, OTH{}, n - RM - RN - NGA);
    });
    __syncthreads();
  };

  if constexpr (BF16) {
    auto pack = [](const float4& lo, const float4& hi) {
      return indm_pack_bf16x8(lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w);
    };
    auto slab16 = [&](auto PAR, int s) {
      using OTH = std::integral_constant<int, 1 - decltype(PAR)::value>;
      const int kt = min(s + 2, nk - 1);
#pragma unroll
      for (int j = 0; j < NGA; ++j) gl_a(PAR, j, kt);   // slab s+2 -> register set PAR (its old content is in LDS)
#pragma unroll
      for (int j = 0; j < NGB; ++j) gl_b(PAR, j, kt);
#pragma unroll
      for (int i = 0; i < RM; ++i) { fr_a(PAR, I0{}, i); fr_a(PAR, I1{}, i); }
#pragma unroll
      for (int j = 0; j < RN; ++j) { fr_b(PAR, I0{}, j); fr_b(PAR, I1{}, j); }
      bf16x8 hb[RN];
#pragma unroll
      for (int j = 0; j < RN; ++j) hb[j] = pack(fb[0][j], fb[1][j]);
#pragma unroll
      for (int i = 0; i < RM; ++i) {
        const bf16x8 ha = pack(fa[0][i], fa[1][i]);
#pragma unroll
        for (int j = 0; j < RN; ++j) {
          // (the builtin, not inline asm: the hazard recogniser must see the converts that feed these MFMAs)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb[j], acc[i][j], 0, 0, 0);
        }
      }
#pragma unroll
      for (int j = 0; j < NGA; ++j) st_a(OTH{}, OTH{}, j);  // slab s+1 (set 1-PAR) -> LDS buffer 1-PAR
#pragma unroll
      for (int j = 0; j < NGB; ++j) st_b(OTH{}, OTH{}, j);
      __syncthreads();
    };
#pragma unroll
    for (int j = 0; j < NGA; ++j) gl_a(I0{}, j, 0);
#pragma unroll
    for (int j = 0; j < NGB; ++j) gl_b(I0{}, j, 0);
#pragma unroll
    for (int j = 0; j < NGA; ++j) st_a(I0{}, I0{}, j);
#pragma unroll
    for (int j = 0; j < NGB; ++j) st_b(I0{}, I0{}, j);
    {
      const int k1 = min(1, nk - 1);
#pragma unroll
      for (int j = 0; j < NGA; ++j) gl_a(I1{}, j, k1);
#pragma unroll
      for (int j = 0; j < NGB; ++j) gl_b(I1{}, j, k1);
    }
    __syncthreads();
    for (int s = 0; s < nk; s += 2) {
      slab16(I0{}, s);
      if (s + 1 < nk) slab16(I1{}, s + 1);
    }
  } else {
  // prologue: slab 0 -> LDS buffer 0, slab 1 -> set 1, slab 2 -> set 0; then "k-group B" of slab 0
#pragma unroll
  for (int j = 0; j < NGA; ++j) gl_a(I0{}, j, 0);
#pragma unroll
  for (int j = 0; j < NGB; ++j) gl_b(I0{}, j, 0);
#pragma unroll
  for (int j = 0; j < NGA; ++j) st_a(I0{}, I0{}, j);
#pragma unroll
  for (int j = 0; j < NGB; ++j) st_b(I0{}, I0{}, j);
  {
    const int k1 = min(1, nk - 1);
#pragma unroll
    for (int j = 0; j < NGA; ++j) gl_a(I1{}, j, k1);
#pragma unroll
    for (int j = 0; j < NGB; ++j) gl_b(I1{}, j, k1);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < RM; ++i) fr_a(I0{}, I0{}, i);
#pragma unroll
  for (int j = 0; j < RN; ++j) fr_b(I0{}, I0{}, j);
  {
    const int k2 = min(2, nk - 1);
#pragma unroll
    for (int j = 0; j < NGA; ++j) gl_a(I0{}, j, k2);
#pragma unroll
    for (int j = 0; j < NGB; ++j) gl_b(I0{}, j, k2);
  }
  kgroup(I0{}, [&](auto NC) {
    constexpr int n = decltype(NC)::value;
    if constexpr (n < RM) fr_a(I0{}, I1{}, n);
    else if constexpr (n < RM + RN) fr_b(I0{}, I1{}, n - RM);
    else if constexpr (n < RM + RN + NGA) st_a(I1{}, I1{}, n - RM - RN);
    else st_b(I1{}, I1{}, n - RM - RN - NGA);
  });
  __syncthreads();

  for (int s = 1; s < nk; s += 2) {
    slab(I1{}, s);
    if (s + 1 < nk) slab(I0{}, s + 1);
  }
  kgroup(I1{}, [&](auto) {});  // k-group 1 of the last slab
  }
  if constexpr (TN == 256 && !BF16) {
    // the accumulators were written by MFMAs the compiler cannot see: cover the XDL-write -> VALU-read hazard
    // (18 wait states for a 16-pass MFMA) by hand before the epilogue reads them
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
      for (int j = 0; j < RN; ++j) asm volatile("s_nop 15\n\ts_nop 7" : "+a"(acc[i][j]));
  }

  if (p.splitk > 1) {  // partial sums -> workspace slab z; igemm_splitk_reduce applies the epilogue
    float* W = p.ws + ((long)z * p.M) * p.N;
#pragma unroll
    for (int i = 0; i < RM; ++i) {
#pragma unroll
      for (int j = 0; j < RN; ++j) {
        const int n = n0 + nw + j * 32 + l31;
        if (n >= p.N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          if (m < p.M) W[(long)m * p.N + n] = acc[i][j][r];
        }
      }
    }
    return;
  }
  float* C = p.C + (long)bz * p.sC;
  const float* R = p.R ? p.R + (long)bz * p.sR : nullptr;
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j)
      epilogue_block(p, acc[i][j], C, p.ldc, R, false, m0 + mw + i * 32, n0 + nw + j * 32 + l31, lhi);
}

// ---------------------------------------------------------------------------------------------------------------
// Gated plain GEMM with DIRECT global -> LDS staging (global_load_lds_dwordx4, gfx950).  The flow's dominant
// contraction (v .* cos) W^T keeps two streams in flight for its A operand (the tensor and its gate); with register
// staging that overflows the 128 x 64 register tile's budget (igemm_big_kernel) -- here nothing is staged in VGPRs:
// every wave DMA-copies its share of the next slab (A, gate, B) straight into the other LDS buffer while the MFMAs
// of the current slab run, and the gate is multiplied in when the fragments are read.
//   * LDS images: A / gate / k-contiguous B as [row][16 k] with the XOR chunk swizzle applied on the SOURCE address
//     (the DMA writes lane-linear: LDS slot = lane), row-contiguous B as plain [16 k][128 n]
//   * 256 x 128 block tile, 4 waves of 128 x 64, two blocks per CU; 80 KB of LDS per block when gated
typedef float f32x2 __attribute__((ext_vector_type(2)));
// (an inline-asm v_pk_mul_f32 here is unsafe: the hazard recogniser does not see its VGPR writes next to in-flight MFMAs)
__device__ __forceinline__ f32x2 pk_mul_f32(f32x2 a, f32x2 b) { return a * b; }

// One direct-to-LDS copy (64 lanes x 16 bytes) in the `voffset, s[base:base+1]` form: a wave-uniform base in an SGPR pair
// that the SCALAR unit advances from slab to slab + a per-lane 32-bit byte offset that never changes.  hipcc lowers
// __builtin_amdgcn_global_load_lds on a loop-carried pointer to a 64-bit VALU add per copy, and non-MFMA VALU
// instructions do not overlap with the MFMAs on this part: tools/probe/mfma_model.hip prices that arithmetic at
// 9-11 % of a 32-MFMA slab (125 -> 141 TF in the model).  Opaque to the compiler: the callers order it with their own
// s_waitcnt vmcnt + barriers, exactly as they order the builtin.
__device__ __forceinline__ void dma_saddr(unsigned lds_byte_addr, unsigned voff, const void* sbase) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
               :: "s"(lds_byte_addr), "v"(voff), "s"(sbase) : "memory", "m0");
}
// The same copy through a raw BUFFER resource (`buffer_load_dwordx4 voffset, s[rsrc], soffset offen lds`, gfx950):
// address = resource base + soffset (SGPR: the slab's uniform tap offset) + the lane's fixed 32-bit offset, and a lane
// whose offset is out of range (bit 31 set) copies ZEROS -- a padded tap of an implicit-GEMM convolution costs two VALU
// instructions (shift + and-or of a per-lane tap mask) instead of two compares, a 64-bit add and a two-register
// select of a zero page's address (tools/probe/buffer_lds.hip checks the semantics on the part).
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma_buffer(unsigned lds_byte_addr, unsigned voff, i32x4 rsrc, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_byte_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_byte_addr(const float* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}

// 16 bytes of zeros in device memory: the source of a padded tap's DMA copy
__device__ __attribute__((aligned(16))) float indm_zero16[4] = {0.f, 0.f, 0.f, 0.f};

// BF16 = true: the "bf16 compute / fp32 accumulate" mode of BASELINE configs #3-#5 on the same staging: the fp32 LDS
// images are unchanged (the DMA cannot convert), the two 4-deep fragments a lane reads per slab are rounded to bf16
// (v_cvt_pk_bf16_f32, RNE; gated operands round the fp32 product) and packed into ONE v_mfma_f32_32x32x16_bf16 per
// 32 x 32 block and slab -- 8x fewer matrix cycles, so the kernel turns from MFMA-bound into an HBM streamer.  (Any
// bijection (lane-half, position) -> k is valid for an MFMA as long as both operands use the same one.)
// LDS slab buffers (see the BF16 loop).  Three where it measured faster (bf16, ungated: implicit-GEMM convolution forward
// + 5 .. 28 %, data-grad + 3 .. 9 %, NT GEMM + 5 %, TN split-K + 10 %), two where it did not (the A_WGRAD weight-grad
// image - 4 .. 8 %, NN GEMM - 1 .. 5 %) and for the gated forms (3 x 40 KB would leave one workgroup per CU).
template <int AM, int BMODE, bool HAS_G, bool BF16>
constexpr int GLDS_NBUF = (BF16 && !HAS_G && (AM == A_CONV || AM == A_COL || (AM == A_ROW && BMODE == B_COL))) ? 3 : 2;
template <int C>
__device__ __forceinline__ void glds_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C) : "memory"); }
// Workgroup barrier that orders LDS accesses for the COMPILER too (workgroup-scope release / acquire fences around
// s_barrier: what __syncthreads() is made of) without waiting for the asm-issued copies, which hipcc does not count: a bare
// s_barrier intrinsic is IntrNoMem -- nothing but the scheduler's habits keeps the LDS reads of the next slab below it.
// (ISA check: no extra vmcnt wait appears in the loop, the counted waits stay vmcnt(NDMA) / vmcnt(0).)
__device__ __forceinline__ void glds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// ACT (round 5): the flow's Sin activation as the epilogue (second level of the g-net, indm_gemm_sin_f32): every tile takes the
// general epilogue, which ends in sincosf instead of the store of v.
// ACT = 2: the Sin-level BACKWARD of the flow as the epilogue (indm_gemm_ires_bwd_act_f32, epilogue_bwd_act_block above): the
// HBM-bound pass (four tensors read, two written per element) runs while the CU's other workgroup multiplies.
template <int AM, int BMODE, bool HAS_G, int NW, bool BF16 = false, int ACT = 0>
__global__ __launch_bounds__(NW * 64, (NW == 8 ? 4 : 2)) void igemm_glds_kernel(IgemmParams p) {
  static_assert(AM == A_ROW || AM == A_CONV || AM == A_COL || AM == A_WGRAD,
                "A_ROW / A_CONV: k-contiguous; A_COL / A_WGRAD: row-contiguous ([16 k][256 m] image)");
  constexpr bool A_RC = (AM == A_COL || AM == A_WGRAD);   // row-contiguous A image
  static_assert(BK == 16, "written for 16-deep slabs");
  // NW = 4: waves 2 x 2 of 128 x 64;  NW = 8: waves 4 x 2 of 64 x 64 (four waves per SIMD with two blocks per CU)
  constexpr int TM = 256, TN = 128, WM = (NW == 8 ? 64 : 128), WN = 64, RM = WM / 32, RN = 2;
  constexpr bool BKC = (BMODE == B_COL || BMODE == B_DGRAD);
  constexpr int A_SZ = TM * BK, G_SZ = HAS_G ? TM * BK : 0, B_SZ = TN * BK, BUF = A_SZ + G_SZ + B_SZ;
  extern __shared__ __attribute__((aligned(16))) float glds_smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: LDS destinations of the DMA stay in SGPRs
  const int tiles_m = (p.M + TM - 1) / TM, tiles_n = (p.N + TN - 1) / TN;
  // 1-D grid over (tile, batch x split-K slice); the XCD-aware remap runs over the COMBINED index so that the tiles of
  // one K-slice -- which share their A / B panels -- are neighbours on one XCD's L2 (a 2-D grid put the 8 tiles of
  // a slice on 8 different XCDs: every panel streamed from HBM once per tile, 4x / 2x the traffic of a bf16 wgrad)
  const int ntile = tiles_m * tiles_n;
  const int lg = xcd_remap(blockIdx.x, ntile * p.batch * p.splitk);
  const int z = lg / ntile, lt = lg - z * ntile;
  const int tm = lt / tiles_n, tn = lt - tm * tiles_n;
  const int m0 = tm * TM, n0 = tn * TN;
  const int bz = z / p.splitk, kz = z - bz * p.splitk;
  const float* A = p.A + (long)bz * p.sA;
  const float* G = HAS_G ? p.G + (long)bz * p.sA : nullptr;
  const float* B = p.B + (long)bz * p.sB;
  const int kbeg = kz * p.kps;                         // split-K: this block reduces over [kbeg, kbeg + kps)
  const int nk = (min(p.K, kbeg + p.kps) - kbeg) / BK;

  f32x16 acc[RM][RN];
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto dma = [&](const float* src, float* dst_wave_base) {
    if constexpr (BF16) {
      // opaque to the compiler like dma_saddr: hipcc puts an s_waitcnt vmcnt(0) in front of every LDS read that follows a
      // BUILTIN copy (its LDS-DMA alias tracking), which would serialise the three-buffer loop even when this path never runs
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                   :: "s"(lds_byte_addr(dst_wave_base)), "v"(src) : "memory", "m0");
    } else {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst_wave_base, 16, 0, 0);
    }
  };
  // implicit-GEMM A operand: the output pixel of each of this lane's LDS rows (fixed over the K loop)
  PixDec pix[16 / NW];
  // undilated input (everything but the data-grad of a strided conv): top-left input coordinate and element offset of
  // each of this lane's rows, fixed over the K loop -> per slab only the (uniform) tap offset is added and two
  // unsigned compares test the padding (no integer multiplies in the loop: v_mul_lo_u32 is quarter rate and the
  // loop's VALU work does not overlap with the MFMAs)
  int ih0[16 / NW], iw0[16 / NW];
  long pbase[16 / NW];
  // buffer form (p.a_buf: undilated input smaller than 2 GB, at most 31 taps): the resource starts `shift` elements
  // BEFORE the tensor so that the offset of a pixel's top-left tap is never negative; cvoff = the lane's fixed byte
  // offset, cbad = bit t set if tap t falls into the padding for this lane's pixel
  unsigned cvoff[16 / NW], cbad[16 / NW];
  i32x4 arsrc = {0, 0, 0, 0};
  if constexpr (AM == A_CONV) {
    const long shift = ((long)p.pad * p.W + p.pad) * p.Cin;
    if (p.a_buf) {
      const unsigned long long ab = (unsigned long long)(A - shift);
      arsrc[0] = (int)(unsigned)(ab & 0xffffffffu);
      arsrc[1] = (int)(unsigned)((ab >> 32) & 0xffffu);                 // stride 0: raw buffer
      arsrc[2] = (int)(((long)(p.M / (p.OH * p.OW)) * p.H * p.W * p.Cin + shift) * 4);
      arsrc[3] = 0x00020000;                                            // DATA_FORMAT_32 (gfx90a / gfx94x / gfx950)
#pragma unroll
      for (int i = 0; i < 4; ++i) arsrc[i] = __builtin_amdgcn_readfirstlane(arsrc[i]);
    }
#pragma unroll
    for (int q = 0; q < 16 / NW; ++q) {
      const int row = wave * (256 / NW) + q * 16 + (lane >> 2);
      pix[q] = decode_pix(min(m0 + row, p.M - 1), p.OH, p.OW);
      ih0[q] = pix[q].oh * p.stride - p.pad;
      iw0[q] = pix[q].ow * p.stride - p.pad;
      const int c = (lane & 3) ^ ((row >> 2) & 3);
      pbase[q] = (((long)pix[q].img * p.H + ih0[q]) * p.W + iw0[q]) * p.Cin + 4 * c;
      cvoff[q] = (unsigned)((pbase[q] + shift) * 4);
      unsigned bad = 0;
      if (p.a_buf) {
        for (int t = 0; t < p.KH * p.KW; ++t) {
          const int kh = t / p.KW, kw = t - kh * p.KW;
          const bool ok = (unsigned)(ih0[q] + kh) < (unsigned)p.H && (unsigned)(iw0[q] + kw) < (unsigned)p.W;
          bad |= ok ? 0u : (1u << t);
        }
      }
      cbad[q] = bad;
    }
  }
  // A_ROW / B_COL / B_ROW: every DMA of this lane reads the same row (or k-row) in every slab -> per-lane source
  // pointers advanced by one slab per stage() call (one v_lshl_add_u64 each) instead of re-deriving 64-bit offsets
  // from k0 (the non-MFMA VALU work of the loop does not overlap with the MFMAs: PMC SQ_VALU_MFMA_COEXEC_CYCLES = 0)
  // (round 3: wave-uniform running bases in SGPRs + fixed per-lane 32-bit byte offsets, dma_saddr above)
  const char *ua = nullptr, *ug = nullptr, *ub = nullptr;
  unsigned aoff[16 / NW], boff[8 / NW];
  if constexpr (AM == A_ROW) {
    ua = (const char*)(A + (long)m0 * p.lda + kbeg);
    ug = HAS_G ? (const char*)(G + (long)m0 * p.lda + kbeg) : nullptr;
#pragma unroll
    for (int q = 0; q < 16 / NW; ++q) {
      const int row = wave * (256 / NW) + q * 16 + (lane >> 2);
      const int c = (lane & 3) ^ ((row >> 2) & 3);
      aoff[q] = ((unsigned)(min(m0 + row, p.M - 1) - m0) * (unsigned)p.lda + 4u * c) * 4u;
    }
  }
  if constexpr (AM == A_COL) {  // A stored [K][M]: LDS image [16 k][256 m], one k-row (1 KB) per wave-instruction
    ua = (const char*)(A + (long)kbeg * p.lda + m0);
    ug = HAS_G ? (const char*)(G + (long)kbeg * p.lda + m0) : nullptr;
#pragma unroll
    for (int q = 0; q < 16 / NW; ++q)
      aoff[q] = ((unsigned)(wave * (16 / NW) + q) * (unsigned)p.lda + (unsigned)(min(m0 + 4 * lane, p.M - 4) - m0)) * 4u;
  }
  // A_WGRAD (round 3): weight-gradient gather A[m = (tap, ci)][k = output pixel] = x[pixel shifted by tap][ci] as a
  // [16 k][256 m] image.  One k-row is one copy instruction: the pixel's top-left input offset is wave-uniform (soffset of a
  // buffer copy, scalar arithmetic), the lane's (tap, ci) offset never changes, and a lane whose tap falls into the padding
  // for this pixel sets bit 31 of its offset and copies zeros (see dma_buffer).  Needs OW and OH*OW powers of two, Cin % 4 == 0.
  unsigned wg_voff = 0;
  int wg_kh = 0, wg_kw = 0;
  bool wg_dead = false;
  i32x4 wrsrc = {0, 0, 0, 0};
  if constexpr (AM == A_WGRAD) {
    const long shift = ((long)p.pad * p.W + p.pad) * p.Cin;
    const unsigned long long ab = (unsigned long long)(A - shift);
    wrsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(ab & 0xffffffffu));
    wrsrc[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((ab >> 32) & 0xffffu));
    wrsrc[2] = __builtin_amdgcn_readfirstlane((int)(((long)(p.K >> p.ohw_sh) * p.H * p.W * p.Cin + shift) * 4));
    wrsrc[3] = 0x00020000;
    const int m = m0 + 4 * lane;                      // this lane's four rows (tap, ci .. ci + 3): one 16-byte chunk
    wg_dead = m >= p.M;
    const int tap = min(m, p.M - 4) / p.Cin, ci = min(m, p.M - 4) - tap * p.Cin;
    wg_kh = tap / p.KW;
    wg_kw = tap - wg_kh * p.KW;
    wg_voff = (unsigned)((((long)wg_kh * p.W + wg_kw) * p.Cin + ci) * 4);
  }
  if constexpr (BMODE == B_COL) {
    ub = (const char*)(B + (long)n0 * p.ldb + kbeg);
#pragma unroll
    for (int q = 0; q < 8 / NW; ++q) {
      const int row = wave * (128 / NW) + q * 16 + (lane >> 2);
      const int c = (lane & 3) ^ ((row >> 2) & 3);
      boff[q] = ((unsigned)(min(n0 + row, p.N - 1) - n0) * (unsigned)p.ldb + 4u * c) * 4u;
    }
  } else if constexpr (BMODE == B_ROW) {
    ub = (const char*)(B + (long)kbeg * p.ldb + n0);
#pragma unroll
    for (int q = 0; q < 8 / NW; ++q)
      boff[q] = ((unsigned)(wave * (16 / NW) + q * 2 + (lane >> 5)) * (unsigned)p.ldb +
                 (unsigned)(min(n0 + 4 * (lane & 31), p.N - 4) - n0)) * 4u;
  }
  const long b_step = (BMODE == B_ROW ? (long)BK * p.ldb : BK) * (long)sizeof(float);   // bytes per slab
  const long a_step = (AM == A_COL ? (long)BK * p.lda : BK) * (long)sizeof(float);
  long bdg[8 / NW];  // B_DGRAD: this lane's row of the packed weight (fixed over the K loop)
  if constexpr (BMODE == B_DGRAD) {
#pragma unroll
    for (int q = 0; q < 8 / NW; ++q) {
      const int row = wave * (128 / NW) + q * 16 + (lane >> 2);
      const int c = (lane & 3) ^ ((row >> 2) & 3);
      bdg[q] = (long)min(n0 + row, p.N - 1) * p.Cout + 4 * c;
    }
  }
  // this wave's share of slab kt -> LDS buffer buf (called for kt = 0, 1, 2, ... in order)
  auto stage = [&](int buf, int kt) {
    float* base = glds_smem + buf * BUF;
    const int k0 = kbeg + kt * BK;
    int kh = 0, kw = 0, ci0 = 0, tap_idx = 0;
    long tap_off = 0;
    if constexpr (AM == A_CONV) {  // one tap per slab (host checks Cin % 16 == 0)
      const int tap = k0 / p.Cin;
      tap_idx = tap;
      ci0 = k0 - tap * p.Cin;
      kh = tap / p.KW;
      kw = tap - kh * p.KW;
      tap_off = ((long)kh * p.W + kw) * p.Cin + ci0;  // uniform
    }
#pragma unroll
    for (int q = 0; q < 16 / NW; ++q) {  // A (+ gate): 16 rows per instruction, 256 / NW rows per wave
      const int r0 = wave * (256 / NW) + q * 16;
      const int row = r0 + (lane >> 2);
      const int c = (lane & 3) ^ ((row >> 2) & 3);  // the k-chunk that belongs in LDS slot (row, lane&3)
      if constexpr (AM == A_ROW) {
        dma_saddr(lds_byte_addr(base + r0 * BK), aoff[q], ua);
        if constexpr (HAS_G) dma_saddr(lds_byte_addr(base + A_SZ + r0 * BK), aoff[q], ug);
      } else if constexpr (AM == A_COL) {
        const int kr = wave * (16 / NW) + q;
        dma_saddr(lds_byte_addr(base + kr * TM), aoff[q], ua);
        if constexpr (HAS_G) dma_saddr(lds_byte_addr(base + A_SZ + kr * TM), aoff[q], ug);
      } else if constexpr (AM == A_WGRAD) {
        const int kr = wave * (16 / NW) + q;
        const int kp = k0 + kr;                                        // output pixel of this k-row (wave-uniform)
        const int img = kp >> p.ohw_sh, rem = kp & ((1 << p.ohw_sh) - 1);
        const int ih0w = (rem >> p.ow_sh) * p.stride - p.pad, iw0w = (rem & ((1 << p.ow_sh) - 1)) * p.stride - p.pad;
        // (the host checked that the whole input + the shift stays below 2 GB: 32-bit scalar arithmetic)
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(
            (((img * p.H + ih0w) * p.W + iw0w) * p.Cin + (p.pad * p.W + p.pad) * p.Cin) * 4);
        const bool bad = wg_dead || (unsigned)(ih0w + wg_kh) >= (unsigned)p.H || (unsigned)(iw0w + wg_kw) >= (unsigned)p.W;
        dma_buffer(lds_byte_addr(base + kr * TM), bad ? 0x80000000u : wg_voff, wrsrc, soff);
      } else if (p.a_buf) {     // block-uniform branch: buffer form, see dma_buffer
        const unsigned oob = (cbad[q] << (31 - tap_idx)) & 0x80000000u;
        dma_buffer(lds_byte_addr(base + r0 * BK), cvoff[q] | oob, arsrc, (unsigned)(tap_off * 4));
      } else if (p.dil == 1) {  // block-uniform branch
        const bool ok = (unsigned)(ih0[q] + kh) < (unsigned)p.H && (unsigned)(iw0[q] + kw) < (unsigned)p.W;
        const float* src = A + pbase[q] + tap_off;
        dma(ok ? src : indm_zero16, base + r0 * BK);  // zero padding: the DMA copies from a 16-byte zero page
      } else {
        int ih = pix[q].oh * p.stride + kh - p.pad, iw = pix[q].ow * p.stride + kw - p.pad;
        bool ok = ih >= 0 && iw >= 0;
        ok = ok && ((ih % p.dil) | (iw % p.dil)) == 0;
        ih /= p.dil;
        iw /= p.dil;
        ok = ok && ih < p.H && iw < p.W;
        const float* src = A + (((long)pix[q].img * p.H + ih) * p.W + iw) * p.Cin + ci0 + 4 * c;
        dma(ok ? src : indm_zero16, base + r0 * BK);
      }
    }
    if constexpr (AM == A_ROW || AM == A_COL) {   // the uniform bases move on by one slab (scalar adds)
      ua += a_step;
      if constexpr (HAS_G) ug += a_step;
    }
    float* bb = base + A_SZ + G_SZ;
#pragma unroll
    for (int q = 0; q < 8 / NW; ++q) {
      if constexpr (BKC) {  // [128 n-rows][16 k]: 128 / NW rows per wave
        const int r0 = wave * (128 / NW) + q * 16;
        const int row = r0 + (lane >> 2);
        const int c = (lane & 3) ^ ((row >> 2) & 3);
        if constexpr (BMODE == B_COL) {
          dma_saddr(lds_byte_addr(bb + r0 * BK), boff[q], ub);
        } else {  // B_DGRAD: packed weight [T][N][Cout], taps flipped; one tap per slab
          const int tap = k0 / p.Cout, co0 = k0 - tap * p.Cout;
          const long toff = (long)(p.KH * p.KW - 1 - tap) * p.N * p.Cout + co0;  // uniform
          dma_saddr(lds_byte_addr(bb + r0 * BK), (unsigned)(bdg[q] * 4), B + toff);
        }
      } else {              // [16 k][128 n]: two k-rows per instruction, 16 / NW k-rows per wave
        const int kr0 = wave * (16 / NW) + q * 2;
        dma_saddr(lds_byte_addr(bb + kr0 * TN), boff[q], ub);
      }
    }
    if constexpr (BMODE == B_COL || BMODE == B_ROW) ub += b_step;
  };

  const int wm = wave >> 1, wn = wave & 1;
  const int mw = wm * WM, nw = wn * WN;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int swz = (l31 >> 2) & 3;

  auto frags = [&](int buf, int sg, float4 (&fa)[RM], float4 (&fb)[RN]) {
    const float* as = glds_smem + buf * BUF;
    const float* bs = as + A_SZ + G_SZ;
    const int ch = ((sg * 2 + lhi) ^ swz) << 2;
#pragma unroll
    for (int i = 0; i < RM; ++i) {
      if constexpr (A_RC) {  // [16 k][256 m]: k = 8 sg + 4 lhi + t, rows of TM floats
        const float* q = as + (sg * 8 + lhi * 4) * TM + mw + i * 32 + l31;
        float4 v = make_float4(q[0], q[TM], q[2 * TM], q[3 * TM]);
        if constexpr (HAS_G) {
          const float* g = q + A_SZ;
          v.x *= g[0]; v.y *= g[TM]; v.z *= g[2 * TM]; v.w *= g[3 * TM];
        }
        fa[i] = v;
      } else {
        const int o = (mw + i * 32 + l31) * BK + ch;
        if constexpr (HAS_G) {  // gate multiplied as two packed pairs (v_pk_mul_f32: half the VALU issue slots)
          const f32x4 av = *(const f32x4*)(as + o), gv = *(const f32x4*)(as + A_SZ + o);
          const f32x2 lo = pk_mul_f32(av.lo, gv.lo), hi = pk_mul_f32(av.hi, gv.hi);
          fa[i] = make_float4(lo.x, lo.y, hi.x, hi.y);
        } else {
          fa[i] = *(const float4*)(as + o);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < RN; ++j) {
      if constexpr (BKC) {
        fb[j] = *(const float4*)(bs + (nw + j * 32 + l31) * BK + ch);
      } else {
        const float* q = bs + (sg * 8 + lhi * 4) * TN + nw + j * 32 + l31;
        fb[j] = make_float4(q[0], q[TN], q[2 * TN], q[3 * TN]);
      }
    }
  };
  auto mfmas = [&](const float4 (&fa)[RM], const float4 (&fb)[RN]) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j) {
          const float a = t == 0 ? fa[i].x : t == 1 ? fa[i].y : t == 2 ? fa[i].z : fa[i].w;
          const float b = t == 0 ? fb[j].x : t == 1 ? fb[j].y : t == 2 ? fb[j].z : fb[j].w;
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0);
        }
  };

  // Skewed by half a slab (as in igemm_big_kernel): after each barrier the fragments of the new slab are requested
  // first, then the MFMAs of the PREVIOUS slab's second k-group run (their operands are already in registers), so the
  // LDS latency right after the barrier is covered.
  float4 fa0[RM], fb0[RN], fa1[RM], fb1[RN];
  if constexpr (BF16) {
    auto pack = [](const float4& lo, const float4& hi) {
      return indm_pack_bf16x8(lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w);
    };
    // GLDS_NBUF = 3: two slabs in flight per workgroup (round 5).  A 16-deep slab holds 4 bf16 MFMAs per wave --
    // ~0.05 us of arithmetic between a copy wait and a workgroup barrier -- so with one slab in flight the kernel ran at the
    // copies' latency (885 MB staged in 102 us by the 3 x 3 convolution at batch 128: 1.4 us per slab and workgroup).  The
    // loop waits with a counted vmcnt for the OLDER slab only and synchronises with a bare s_barrier (__syncthreads()
    // would drain the copies), as ires_chain_pw_kernel does.  Conv forward 16 x 16, 512 -> 256 at batch 128: 280 -> 201 us.
    constexpr int D = GLDS_NBUF<AM, BMODE, HAS_G, true> - 1;
    constexpr int NDMA = (16 / NW) * (HAS_G ? 2 : 1) + 8 / NW;   // copy instructions per stage() and wave
    {
      const int pre = nk < D ? nk : D;
      for (int d = 0; d < pre; ++d) stage(d, d);
      if (nk >= D) glds_wait_vm<NDMA * (D - 1)>();   // slab 0 has landed, the rest of the prologue is on its way
      else glds_wait_vm<0>();
      glds_barrier();
    }
    int buf = 0, sbuf = D % (D + 1);   // buffer of slab s / of slab s + D
    for (int s = 0; s < nk; ++s) {
      if (s + D < nk) stage(sbuf, s + D);   // that buffer's previous contents (slab s - 1) were read before the last barrier
      // operand by operand: read the two 4-deep fragments, (gate,) round and pack at once -- short live ranges (the
      // generic frags() + pack kept 2 x (RM + RN) float4 alive and spilled an accumulator in the gated A_COL variant)
      bf16x8 ha[RM], hb[RN];
      {
#pragma unroll
        for (int i = 0; i < RM; ++i) {
          const float* as = glds_smem + buf * BUF;
          float4 lo, hi;
          if constexpr (A_RC) {
            const float* q = as + (lhi * 4) * TM + mw + i * 32 + l31;
            lo = make_float4(q[0], q[TM], q[2 * TM], q[3 * TM]);
            hi = make_float4(q[8 * TM], q[9 * TM], q[10 * TM], q[11 * TM]);
            if constexpr (HAS_G) {
              const float* g = q + A_SZ;
              lo.x *= g[0]; lo.y *= g[TM]; lo.z *= g[2 * TM]; lo.w *= g[3 * TM];
              hi.x *= g[8 * TM]; hi.y *= g[9 * TM]; hi.z *= g[10 * TM]; hi.w *= g[11 * TM];
            }
          } else {
            const int o = (mw + i * 32 + l31) * BK;
            const int c0 = (lhi ^ swz) << 2, c1 = ((2 + lhi) ^ swz) << 2;
            lo = *(const float4*)(as + o + c0);
            hi = *(const float4*)(as + o + c1);
            if constexpr (HAS_G) {
              const float4 g0 = *(const float4*)(as + A_SZ + o + c0), g1 = *(const float4*)(as + A_SZ + o + c1);
              lo.x *= g0.x; lo.y *= g0.y; lo.z *= g0.z; lo.w *= g0.w;
              hi.x *= g1.x; hi.y *= g1.y; hi.z *= g1.z; hi.w *= g1.w;
            }
          }
          ha[i] = pack(lo, hi);
          if constexpr (AM == A_COL && HAS_G) __builtin_amdgcn_sched_barrier(0);  // one operand's reads at a time: <= 128 VGPRs
        }
#pragma unroll
        for (int j = 0; j < RN; ++j) {
          const float* bs = glds_smem + buf * BUF + A_SZ + G_SZ;
          float4 lo, hi;
          if constexpr (BKC) {
            const int o = (nw + j * 32 + l31) * BK;
            lo = *(const float4*)(bs + o + ((lhi ^ swz) << 2));
            hi = *(const float4*)(bs + o + (((2 + lhi) ^ swz) << 2));
          } else {
            const float* q = bs + (lhi * 4) * TN + nw + j * 32 + l31;
            lo = make_float4(q[0], q[TN], q[2 * TN], q[3 * TN]);
            hi = make_float4(q[8 * TN], q[9 * TN], q[10 * TN], q[11 * TN]);
          }
          hb[j] = pack(lo, hi);
        }
      }
#pragma unroll
      for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha[i], hb[j], acc[i][j], 0, 0, 0);
      // keep the MFMAs on this side of the barrier: hipcc otherwise sinks three of the four below it and then
      // rotates the 64 accumulator registers with v_mov every iteration (and spills in the gated A_COL variant)
      __builtin_amdgcn_sched_barrier(0);
      // slab s + 1 must have landed; the younger one (if any) stays in flight
      if constexpr (D == 1) {   // two buffers: the round-2 loop tail (the weight-grad image measured 3 % slower on the bare barrier)
        glds_wait_vm<0>();
        __syncthreads();
        buf ^= 1;
        sbuf ^= 1;
      } else {
        if (s + 2 < nk) glds_wait_vm<NDMA>();
        else glds_wait_vm<0>();
        glds_barrier();
        buf = buf == D ? 0 : buf + 1;
        sbuf = sbuf == D ? 0 : sbuf + 1;
      }
    }
  } else {
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (nk > 1) stage(1, 1);
  frags(0, 0, fa0, fb0);
  frags(0, 1, fa1, fb1);
  mfmas(fa0, fb0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA has landed ...
  __syncthreads();                                    // ... and so has everybody else's
  for (int s = 1; s < nk; ++s) {
    const int buf = s & 1;
    if (s + 1 < nk) stage(buf ^ 1, s + 1);  // lands while this slab is multiplied; that buffer was last read in slab s-1
    frags(buf, 0, fa0, fb0);
    mfmas(fa1, fb1);                        // k-group 1 of slab s-1
    frags(buf, 1, fa1, fb1);
    mfmas(fa0, fb0);                        // k-group 0 of slab s
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  mfmas(fa1, fb1);
  }

  const bool partial = p.splitk > 1;  // partial sums -> workspace slab z; igemm_splitk_reduce applies the epilogue
  float* C = partial ? p.ws + ((long)z * p.M) * p.N : p.C + (long)bz * p.sC;
  const long ldc = partial ? p.N : p.ldc;
  const float* R = (!partial && p.R) ? p.R + (long)bz * p.sR : nullptr;
  if (ACT == 0 && m0 + TM <= p.M && n0 + TN <= p.N && !R && (partial || !p.bias2)) {
    // full tile, no per-element reads: no bounds tests, one running row pointer per 32 x 32 block (the epilogue's
    // integer work is ~40 % of the kernel's non-MFMA VALU instructions, none of which overlap with the MFMAs)
    const float alpha = partial ? 1.f : p.alpha;
#pragma unroll
    for (int i = 0; i < RM; ++i) {
#pragma unroll
      for (int j = 0; j < RN; ++j) {
        const int n = n0 + nw + j * 32 + l31;
        const float bn = (!partial && p.bias) ? p.bias[n] : 0.f;
        float* cp = C + (long)(m0 + mw + i * 32 + 4 * lhi) * ldc + n;
#pragma unroll
        for (int r = 0; r < 16; ++r) cp[(long)((r & 3) + 8 * (r >> 2)) * ldc] = (acc[i][j][r] + bn) * alpha;
      }
    }
    return;
  }
  if constexpr (ACT == 2 || ACT == 3) {   // 3: the block is du (gated NN GEMM), see IgemmParams::e_da
    static_assert(RM == 2 && RN == 2, "written for the 8-wave tile: a wave owns one 64-row chunk x 64 columns");
    float run0 = 0.f, run1 = 0.f;   // running column sums of the two 32-column blocks over the wave's 64 rows
#pragma unroll 1
    for (int b = 0; b < 4; ++b) {   // (i = b >> 1, j = b & 1): rows 0..31 of both column blocks, then rows 32..63
      f32x16 v;
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = b == 0 ? acc[0][0][r] : b == 1 ? acc[0][1][r] : b == 2 ? acc[1][0][r] : acc[1][1][r];
      float run = (b & 1) ? run1 : run0;
      epilogue_bwd_act_block<ACT == 3>(p, v, m0 + mw + (b >> 1) * 32, n0 + nw + (b & 1) * 32 + l31, lane, run);
      if (b & 1) run1 = run; else run0 = run;
    }
    if (p.e_part && lhi == 0 && m0 + mw < p.M) {
      const long chunk = (m0 + mw) >> 6;
      const int n = n0 + nw + l31;
      if (n < p.N) p.e_part[chunk * p.N + n] = run0;
      if (n + 32 < p.N) p.e_part[chunk * p.N + n + 32] = run1;
    }
    return;
  }
  if constexpr (ACT == 1) {
    // ONE copy of the 16 inlined sincosf per kernel: the four 32 x 32 blocks go through a ROLLED loop, each selected into a
    // fixed register block with uniform conditions.  (Left to `#pragma unroll`, hipcc rolled the j loop on its own, indexed
    // acc[i][j] dynamically and moved all 64 accumulators to scratch -- written back every slab of the main loop.)
    static_assert(RM == 2 && RN == 2, "written for the 8-wave tile");
#pragma unroll 1
    for (int b = 0; b < 4; ++b) {
      f32x16 v;
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = b == 0 ? acc[0][0][r] : b == 1 ? acc[0][1][r] : b == 2 ? acc[1][0][r] : acc[1][1][r];
      epilogue_block<f32x16, true>(p, v, C, ldc, R, partial, m0 + mw + (b >> 1) * 32, n0 + nw + (b & 1) * 32 + l31, lhi);
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j)
      epilogue_block(p, acc[i][j], C, ldc, R, partial, m0 + mw + i * 32, n0 + nw + j * 32 + l31, lhi);
}

__global__ void igemm_splitk_reduce(IgemmParams p);

template <int AM, int BMODE, bool HAS_G, int NW, bool BF16>
int launch_glds_t(const IgemmParams& p, hipStream_t st) {
  constexpr int BUF = 256 * BK + (HAS_G ? 256 * BK : 0) + 128 * BK;
  constexpr size_t bytes = GLDS_NBUF<AM, BMODE, HAS_G, BF16> * BUF * sizeof(float);
  // > 64 KB of LDS needs the opt-in once per kernel and device
  if (indm_lds_optin<igemm_glds_kernel<AM, BMODE, HAS_G, NW, BF16>>(bytes) != INDM_OK) return INDM_E_LAUNCH;
  dim3 grid(cdiv(p.M, 256) * cdiv(p.N, 128) * p.batch * p.splitk);
  if (BF16)
    last_variant = HAS_G ? (p.splitk > 1 ? "glds256x128,splitk,gated,bf16" : "glds256x128,gated,bf16")
                         : (p.splitk > 1 ? "glds256x128,splitk,bf16" : "glds256x128,bf16");
  else
    last_variant = HAS_G ? (p.splitk > 1 ? "glds256x128,splitk,gated" : "glds256x128,gated")
                         : (p.splitk > 1 ? "glds256x128,splitk" : "glds256x128");
  hipLaunchKernelGGL((igemm_glds_kernel<AM, BMODE, HAS_G, NW, BF16>), grid, dim3(NW * 64), bytes, st, p);
  if (hipGetLastError() != hipSuccess) return INDM_E_LAUNCH;
  if (p.splitk > 1) {
    hipLaunchKernelGGL(igemm_splitk_reduce, dim3(ew_grid((long)p.batch * p.M * p.N, 256)), dim3(256), 0, st, p);
    if (hipGetLastError() != hipSuccess) return INDM_E_LAUNCH;
  }
  return INDM_OK;
}

template <int AM, int BMODE, bool HAS_G, int NW>
int launch_glds(const IgemmParams& p, hipStream_t st) {
  if constexpr (AM == A_ROW && BMODE == B_COL && !HAS_G && NW == 8) {
    if (p.e_du) {   // (fp32, no split-K, no bias / residual: checked by the caller)
      constexpr size_t bytes = 2 * (256 * BK + 128 * BK) * sizeof(float);
      if (indm_lds_optin<igemm_glds_kernel<A_ROW, B_COL, false, 8, false, 2>>(bytes) != INDM_OK) return INDM_E_LAUNCH;
      last_variant = "glds256x128,ires_bwd_act";
      hipLaunchKernelGGL((igemm_glds_kernel<A_ROW, B_COL, false, 8, false, 2>), dim3(cdiv(p.M, 256) * cdiv(p.N, 128) * p.batch),
                         dim3(512), bytes, st, p);
      return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
    }
    if (p.act_d) {   // (fp32, no split-K: checked by the caller)
      constexpr size_t bytes = 2 * (256 * BK + 128 * BK) * sizeof(float);
      if (indm_lds_optin<igemm_glds_kernel<A_ROW, B_COL, false, 8, false, 1>>(bytes) != INDM_OK) return INDM_E_LAUNCH;
      last_variant = "glds256x128,sin";
      hipLaunchKernelGGL((igemm_glds_kernel<A_ROW, B_COL, false, 8, false, 1>), dim3(cdiv(p.M, 256) * cdiv(p.N, 128) * p.batch),
                         dim3(512), bytes, st, p);
      return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
    }
  }
  if constexpr (AM == A_ROW && BMODE == B_ROW && HAS_G && NW == 8) {
    if (p.e_da) {   // (fp32, no split-K, no bias / residual: checked by the caller)
      constexpr size_t bytes = 2 * (2 * 256 * BK + 128 * BK) * sizeof(float);
      if (indm_lds_optin<igemm_glds_kernel<A_ROW, B_ROW, true, 8, false, 3>>(bytes) != INDM_OK) return INDM_E_LAUNCH;
      last_variant = "glds256x128,gated,ires_bwd_act";
      hipLaunchKernelGGL((igemm_glds_kernel<A_ROW, B_ROW, true, 8, false, 3>), dim3(cdiv(p.M, 256) * cdiv(p.N, 128) * p.batch),
                         dim3(512), bytes, st, p);
      return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
    }
  }
  return p.bf16 ? launch_glds_t<AM, BMODE, HAS_G, NW, true>(p, st) : launch_glds_t<AM, BMODE, HAS_G, NW, false>(p, st);
}

// Deterministic split-K combine + epilogue.
__global__ void igemm_splitk_reduce(IgemmParams p) {
  const long total = (long)p.batch * p.M * p.N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long mn = (long)p.M * p.N;
    const int bz = (int)(i / mn);
    const long r = i - (long)bz * mn;
    const int m = (int)(r / p.N), n = (int)(r - (long)m * p.N);
    // four partial sums in a fixed order (deterministic); round 5: sixteen slab values are requested before the first is added
    // (four at a time meant splitk / 4 memory latencies in sequence: 12 us per launch on average, 343 launches per step) -- the
    // association of the sums is unchanged, so are the results
    const float* w = p.ws + ((long)bz * p.splitk * p.M + m) * p.N + n;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int kz = 0;
    for (; kz + 15 < p.splitk; kz += 16) {
      float v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = w[(long)(kz + j) * mn];
#pragma unroll
      for (int j = 0; j < 16; j += 4) {
        s0 += v[j];
        s1 += v[j + 1];
        s2 += v[j + 2];
        s3 += v[j + 3];
      }
    }
    for (; kz + 3 < p.splitk; kz += 4) {
      s0 += w[(long)kz * mn];
      s1 += w[(long)(kz + 1) * mn];
      s2 += w[(long)(kz + 2) * mn];
      s3 += w[(long)(kz + 3) * mn];
    }
    for (; kz < p.splitk; ++kz) s0 += w[(long)kz * mn];
    float v = (s0 + s1) + (s2 + s3);
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

// Big-tile kernel eligibility: no split-K and enough 256-row tiles to give every CU work.  Returns the block tile
// width (256 / 128) or 0.
inline int big_tile_n(const IgemmParams& p) {
  if (p.splitk != 1 || p.N < 64 || p.M < 1024 || !p.a_vec || !p.b_vec || (p.K & 15)) return 0;
  int tn = (cdiv(p.N, 256) * 256 <= cdiv(p.N, 128) * 128) ? 256 : 128;
  // 256-wide tiles leave one block per CU, so nothing overlaps a tile's prologue / epilogue: only worth it when the
  // K loop is long (measured: K = 512 -> 106 TF at 256x256 vs 119 TF at 256x128; K = 4096 -> 142 vs 138)
  if (tn == 256 && p.K < 1024) tn = 128;
  if ((long)cdiv(p.M, 256) * cdiv(p.N, tn) * p.batch >= 224) return tn;
  if (tn == 256 && (long)cdiv(p.M, 256) * cdiv(p.N, 128) * p.batch >= 224) return 128;
  return 0;
}

static bool wgrad_glds_enabled() {
  static const int v = getenv("INDM_WGRAD_GLDS") ? atoi(getenv("INDM_WGRAD_GLDS")) : 1;
  return v != 0;
}

template <int AM, int BMODE>
int launch_tiles(const IgemmParams& p, hipStream_t st) {
  // tile choice: skinny N (flow 3/12-channel edges, last score conv) -> 128x32;
  // skinny M (27-/108-row weight grads) -> 32x128; everything else 128x128.
  dim3 block(256);
  if constexpr (AM == A_ROW && (BMODE == B_ROW || BMODE == B_COL)) {
    // long plain GEMMs, gated or not: direct-to-LDS staging kernel (igemm_glds_kernel), 8 waves
    // (plain ones with a long K loop and 256-wide N stay on the 256 x 256 big tile: 142 vs 138 TF at K = 4096)
    const bool big256 = !p.G && p.K >= 1024 && big_tile_n(p) == 256;
    if ((!big256 || p.bf16) && p.splitk == 1 && p.a_vec && p.b_vec && !(p.K & 15) && p.M >= 1024 &&
        p.N >= 64 && (BMODE == B_COL || !(p.N & 3)) && (long)cdiv(p.M, 256) * cdiv(p.N, 128) * p.batch >= 224) {
      if ((p.act_d || p.e_du) && (p.G || BMODE != B_COL)) return 1;
      if (p.e_da && (!p.G || BMODE != B_ROW)) return 1;
      if (p.G) return launch_glds<A_ROW, BMODE, true, 8>(p, st);
      return launch_glds<A_ROW, BMODE, false, 8>(p, st);
    }
  }
  if (p.act_d || p.e_du || p.e_da) return 1;   // the fused epilogues exist on the direct-to-LDS kernel only: the caller runs the two steps
  if constexpr (AM == A_CONV && (BMODE == B_ROW || BMODE == B_DGRAD)) {
    // long convolutions (forward and data-grad): the same direct-to-LDS kernel on the implicit-GEMM A operand.
    // (Round 2 kept the data-grads that fill 256 x 256 tiles on the big-tile kernel, 118-123 TF; with the buffer-addressed
    // copies of round 3 the 8-wave kernel is ahead there too: 16x16 512->256 630 -> 576 us, 32x32 256->128 684 -> 605 us.)
    if (!p.G && p.splitk == 1 && p.a_vec && p.b_vec && p.a_tap_uniform &&
        (BMODE == B_ROW ? !(p.N & 3) : p.b_tap_uniform != 0) && p.M >= 1024 && p.N >= 64 &&
        (long)cdiv(p.M, 256) * cdiv(p.N, 128) * p.batch >= 224)
      return launch_glds<A_CONV, BMODE, false, 8>(p, st);
  }
  if constexpr (AM == A_ROW || (AM == A_CONV && BMODE == B_DGRAD)) {  // the long forward / data-grad contractions
    // measured on MI355X (tools/dense_bench.py): the big tile wins on ungated plain GEMMs (+7-8 %) and on data-grad
    // convolutions that fill 256x256 tiles (+7-14 %); gated GEMMs (two operand streams in flight do not fit the
    // register budget) and B_ROW-weight convolutions stay on the 8-wave kernel.
    const bool eligible = !p.G && (AM == A_ROW || (BMODE == B_DGRAD && p.a_tap_uniform && p.b_tap_uniform)) &&
                          (BMODE != B_ROW || (p.N & 3) == 0);
    int btn = eligible ? big_tile_n(p) : 0;
    if (AM == A_CONV && btn != 256) btn = 0;
    if (btn) {
      // one wave per SIMD with a 128x128 / 128x64 register tile (see igemm_big_kernel)
      dim3 grid(cdiv(p.M, 256) * cdiv(p.N, btn), p.batch);
      last_variant = btn == 256 ? "big256x256" : "big256x128";
      if (p.bf16) {  // (the 256-wide register tile spills with the bf16 fragments: bf16 mode stays on 256 x 128)
        last_variant = "big256x128,bf16";
        dim3 g16(cdiv(p.M, 256) * cdiv(p.N, 128), p.batch);
        hipLaunchKernelGGL((igemm_big_kernel<AM, BMODE, 128, false, true>), g16, block, 0, st, p);
      } else if (btn == 256) hipLaunchKernelGGL((igemm_big_kernel<AM, BMODE, 256, false>), grid, block, 0, st, p);
      else hipLaunchKernelGGL((igemm_big_kernel<AM, BMODE, 128, false>), grid, block, 0, st, p);
      return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
    }
  }
#define INDM_LAUNCH(TMV, TNV, WMV, WNV, GRID, BLOCK)                                                         \
  do {                                                                                                       \
    if (p.bf16) hipLaunchKernelGGL((igemm_kernel<AM, BMODE, TMV, TNV, WMV, WNV, true>), GRID, BLOCK, 0, st, p);  \
    else hipLaunchKernelGGL((igemm_kernel<AM, BMODE, TMV, TNV, WMV, WNV, false>), GRID, BLOCK, 0, st, p);      \
  } while (0)
  if constexpr (AM == A_ROW && (BMODE == B_ROW || BMODE == B_COL)) {
    // long-M GEMMs with N <= 128 and only a few hundred 128-row tiles (the flow's scale-1 edge [32768, 512] x [512, 108]):
    // 64-row tiles double the block count instead of splitting K
    if (p.N > 32 && p.N <= 128 && p.splitk == 1 && p.M >= 8192 && cdiv(p.M, 128) * p.batch < 512) {
      dim3 grid(cdiv(p.M, 64) * cdiv(p.N, 128), p.batch);
      last_variant = p.bf16 ? "64x128,bf16" : "64x128";
      INDM_LAUNCH(64, 128, 2, 2, grid, block);
      return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
    }
  }
  if (p.N <= 32 && p.M > 32) {
    dim3 grid(cdiv(p.M, 128) * cdiv(p.N, 32), p.batch * p.splitk);
    last_variant = p.bf16 ? "128x32,bf16" : "128x32";
    INDM_LAUNCH(128, 32, 4, 1, grid, block);
  } else if (p.M <= 128 && cdiv(p.M, 32) * cdiv(p.N, 128) * p.batch * p.splitk <= 512) {
    // short-M GEMMs (per-image vectors x weights): 32-row tiles so more than a handful of CUs get work
    dim3 grid(cdiv(p.M, 32) * cdiv(p.N, 128), p.batch * p.splitk);
    last_variant = p.bf16 ? "32x128,bf16" : "32x128";
    INDM_LAUNCH(32, 128, 1, 4, grid, block);
  } else if (use_tall_tile<AM>(p)) {
    // 8-wave 256x128 tile: +8-10 % on the long-M forward / data-grad contractions (measured)
    dim3 grid(cdiv(p.M, 256) * cdiv(p.N, 128), p.batch * p.splitk);
    last_variant = p.bf16 ? "256x128,bf16" : "256x128";
    INDM_LAUNCH(256, 128, 4, 2, grid, dim3(512));
  } else {
    dim3 grid(cdiv(p.M, 128) * cdiv(p.N, 128), p.batch * p.splitk);
    last_variant = p.bf16 ? "128x128,bf16" : "128x128";
    INDM_LAUNCH(128, 128, 2, 2, grid, block);
  }
#undef INDM_LAUNCH
  if (hipGetLastError() != hipSuccess) return INDM_E_LAUNCH;
  if (p.splitk > 1) {
    const long total = (long)p.batch * p.M * p.N;
    hipLaunchKernelGGL(igemm_splitk_reduce, dim3(ew_grid(total, 256)), dim3(256), 0, st, p);
    if (hipGetLastError() != hipSuccess) return INDM_E_LAUNCH;
  }
  return INDM_OK;
}

inline bool al16(const void* q) { return ((uintptr_t)q & 15) == 0; }

// Tuning hook of tools/splitk_sweep.py: with INDM_SPLITK_TUNE set, every split-K decision below takes its slice count from
// INDM_SPLITK (read per call; 0 / unset = the rule's own choice).  The clamps to the K length and the workspace still apply.
static long tuned(long want) {
  static const bool on = getenv("INDM_SPLITK_TUNE") != nullptr;
  if (!on) return want;
  const char* v = getenv("INDM_SPLITK");
  const long f = v ? atol(v) : 0;
  return f > 0 ? f : want;
}

// Pick split-K so that tiles*splits fills the chip (>= ~2 waves of 256 CUs) when the
// output is small and the reduction long (weight grads: K = B*H*W pixels).
int choose_splitk(int M, int N, int K, int batch, long ws_floats) {
  long tiles = (long)cdiv(M, (N <= 32 && M > 32) ? 128 : (M <= 128 ? 32 : 128)) *
               cdiv(N, (N <= 32 && M > 32) ? 32 : 128) * batch;
  if (tiles >= 256 || K < 4 * BK * 8) return 1;
  long want = tuned((1024 + tiles - 1) / tiles);  // 512...1536 measured within noise of each other
  long maxs = K / (BK * 8);
  if (want > maxs) want = maxs;
  if (want > 256) want = 256;
  while (want > 1 && (long)want * batch * M * N > ws_floats) --want;
  return want < 1 ? 1 : (int)want;
}

int run(IgemmParams& p, int am, int bm, float* ws, long ws_floats, hipStream_t st) {
  if (p.M <= 0 || p.N <= 0 || p.batch <= 0) return INDM_OK;
  if (p.K < 0) return INDM_E_SHAPE;
  // the general epilogue addresses the residual and the per-image bias with 32-bit element offsets (epilogue_block)
  if ((p.R && (long)p.M * p.ldr >= (1L << 32)) || (p.bias2 && ((long)p.M / (p.rpg > 0 ? p.rpg : 1) + 1) * p.N >= (1L << 32)))
    return INDM_E_SHAPE;
  p.ws = ws;
  if (!p.act_d && !p.e_du && !p.e_da && (am == A_ROW || am == A_COL) && (bm == B_ROW || bm == B_COL)) {
    // thin contractions of the flow's 3-/12-channel edges: HBM-streaming kernels of igemm_thin.hip (bf16 mode: operands
    // rounded at the fragment read, exact fp32 MFMA)
    ThinGemmArgs t = {p.A, p.G, p.B, p.C, p.M, p.N, p.K, am == A_COL, bm == B_COL, p.lda, p.ldb, p.ldc, p.batch,
                      p.alpha, p.beta, p.bias, p.bias2, p.rpg, p.R, p.ldr, ws, ws_floats};
    t.bf16 = p.bf16;
    const char* name = "";
    const int r = indm_thin_gemm(t, st, &name);
    if (r != THIN_NA) {
      last_variant = name;
      return r;
    }
  }
  if (am == A_COL && bm == B_ROW && ws && p.bf16 && p.batch == 1 && p.a_vec && p.b_vec && !(p.K & 15) && p.K >= 8192 &&
      p.M >= 128 && p.N >= 128 && !(p.M & 3) && !(p.N & 3)) {
    // bf16 mode: the weight-grad GEMMs are HBM streamers -- direct-to-LDS staging of the row-contiguous A (and its
    // gate) keeps 16 waves per CU in flight without a register-staged prefetch (the gated big tile spills there)
    const long tiles = (long)cdiv(p.M, 256) * cdiv(p.N, 128);
    long want = tuned(512 / tiles);
    while (want > 1 && (p.K / want < 16 * BK || want * p.M * p.N > ws_floats)) --want;
    if (want > 1) {
      p.kps = cdiv(cdiv(p.K, (int)want), BK) * BK;
      p.splitk = cdiv(p.K, p.kps);
      return p.G ? launch_glds<A_COL, B_ROW, true, 8>(p, st) : launch_glds<A_COL, B_ROW, false, 8>(p, st);
    }
  }
  if (am == A_COL && bm == B_ROW && ws) {
    // weight-grad GEMMs (tiny output, K = pixels): 256x256 big tiles, one block per CU, split-K sized to 256 blocks
    // (bf16 mode: 256 x 128 tiles, two blocks per CU -- the 256-wide register tile spills with the bf16 fragments)
    // 256 x 128 tiles for N <= 256 outputs with M >= 512 ([32768,512]^T x [32768,256]: 140 -> 84 us)
    const int tnw = (p.bf16 || (p.N <= 256 && p.M >= 512 && !p.G)) ? 128 : 256;
    const long tiles = (long)cdiv(p.M, 256) * cdiv(p.N, tnw) * p.batch;
    if (p.M >= 256 && p.N >= 256 && !(p.M & 3) && !(p.N & 3) && !(p.K & 15) && p.a_vec && p.b_vec &&
        tiles <= (tnw == 128 ? 128 : 64) && p.K >= 8192 && !(tnw == 128 && !p.bf16 && p.G)) {
      long want = tuned((tnw == 128 ? 512 : 256) / tiles);
      while (want > 1 && (p.K / want < 512 || want * p.batch * p.M * p.N > ws_floats)) --want;
      // fewer than 128 blocks (e.g. the 256 x 256 NIN weight grads at K = 32768: 64 blocks, 31 TF) leave most of the
      // chip idle: those go to the 128 x 128 split-K path below
      if (want > 1 && tiles * want >= 128) {
        int kps = cdiv(cdiv(p.K, (int)want), BK) * BK;
        p.kps = kps;
        p.splitk = cdiv(p.K, kps);
        last_variant = p.bf16 ? "big256x128,bf16" : "big256x256";
        dim3 grid((unsigned)(cdiv(p.M, 256) * cdiv(p.N, tnw)), (unsigned)(p.batch * p.splitk));
        if (p.bf16) {
          if (p.G) hipLaunchKernelGGL((igemm_big_kernel<A_COL, B_ROW, 128, true, true>), grid, dim3(256), 0, st, p);
          else hipLaunchKernelGGL((igemm_big_kernel<A_COL, B_ROW, 128, false, true>), grid, dim3(256), 0, st, p);
        } else if (tnw == 128) {
          last_variant = "big256x128";
          hipLaunchKernelGGL((igemm_big_kernel<A_COL, B_ROW, 128, false>), grid, dim3(256), 0, st, p);
        } else if (p.G) hipLaunchKernelGGL((igemm_big_kernel<A_COL, B_ROW, 256, true>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((igemm_big_kernel<A_COL, B_ROW, 256, false>), grid, dim3(256), 0, st, p);
        if (hipGetLastError() != hipSuccess) return INDM_E_LAUNCH;
        const long total = (long)p.batch * p.M * p.N;
        hipLaunchKernelGGL(igemm_splitk_reduce, dim3(ew_grid(total, 256)), dim3(256), 0, st, p);
        return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
      }
    }
  }
  if (am == A_COL && bm == B_ROW && ws && !p.bf16 && p.batch == 1 && p.a_vec && p.b_vec && !(p.K & 15) && p.K >= 16384 &&
      p.M >= 256 && p.N >= 128 && !(p.M & 3) && !(p.N & 3) && wgrad_glds_enabled()) {
    // fp32 weight-grad GEMMs that are too small for the big tile (e.g. the attention NIN grads [256, 32768]^T x [32768, 256]):
    // the direct-to-LDS kernel's row-contiguous A image with split-K instead of the generic 128 x 128 tile (round 3;
    // 72.9 -> 63.4 us; shorter K or 128-row outputs measured slower and stay on the generic tile)
    const long tiles = (long)cdiv(p.M, 256) * cdiv(p.N, 128);
    long want = tuned(512 / tiles);
    if (want < 1) want = 1;
    while (want > 1 && (p.K / want < 8 * BK || want * p.M * p.N > ws_floats)) --want;
    p.kps = cdiv(cdiv(p.K, (int)want), BK) * BK;
    p.splitk = cdiv(p.K, p.kps);
    return p.G ? launch_glds<A_COL, B_ROW, true, 8>(p, st) : launch_glds<A_COL, B_ROW, false, 8>(p, st);
  }
  if (am == A_WGRAD && bm == B_ROW && ws && p.a_vec && p.b_vec && !(p.K & 15) && p.N >= 64 && !(p.N & 3) && p.M >= 256 &&
      !(p.Cin & 3) && p.ow_sh >= 0 && p.ohw_sh >= 0 && p.dil == 1 && p.K >= 1024 &&
      ((long)(p.K >> p.ohw_sh) * p.H * p.W + (long)p.pad * (p.W + 1)) * p.Cin * 4 < (1L << 31) && wgrad_glds_enabled()) {
    // convolution weight grads on the direct-to-LDS kernel (round 3, A_WGRAD image): 256 x 128 tiles, split-K over the
    // output pixels so that every CU gets its two workgroups, chunks of at least 8 slabs
    const long tiles = (long)cdiv(p.M, 256) * cdiv(p.N, 128);
    long want = tuned(512 / tiles);
    if (want < 1) want = 1;
    while (want > 1 && (p.K / want < 8 * BK || want * p.M * p.N > ws_floats)) --want;
    p.kps = cdiv(cdiv(p.K, (int)want), BK) * BK;
    p.splitk = cdiv(p.K, p.kps);
    return launch_glds<A_WGRAD, B_ROW, false, 8>(p, st);
  }
  if (am == A_WGRAD && bm == B_ROW && ws && p.a_vec && p.b_vec && !(p.K & 15) && p.N >= 128 && p.M >= 256 &&
      (p.K >= 16384 || (p.K >= 8192 && p.M >= 4096))) {  // (8x8 maps: 512-channel inputs only -- measured 245 -> 192 us)
    // convolution weight grads: same recipe on the implicit-GEMM A operand (256 x 256|128 tiles)
    const int tn = (p.N % 256 == 0 && !p.bf16) ? 256 : 128;
    const long tiles = (long)cdiv(p.M, 256) * cdiv(p.N, tn);
    long want = tuned((tn == 256 ? 256 : 512) / tiles);
    while (want > 1 && (p.K / want < 512 || want * p.M * p.N > ws_floats)) --want;
    if (tiles <= 128 && want > 1) {
      int kps = cdiv(cdiv(p.K, (int)want), BK) * BK;
      p.kps = kps;
      p.splitk = cdiv(p.K, kps);
      dim3 grid((unsigned)tiles, (unsigned)p.splitk);
      last_variant = tn == 256 ? "big256x256" : "big256x128";
      if (p.bf16) {
        last_variant = "big256x128,bf16";
        hipLaunchKernelGGL((igemm_big_kernel<A_WGRAD, B_ROW, 128, false, true>), grid, dim3(256), 0, st, p);
      } else if (tn == 256) hipLaunchKernelGGL((igemm_big_kernel<A_WGRAD, B_ROW, 256, false>), grid, dim3(256), 0, st, p);
      else hipLaunchKernelGGL((igemm_big_kernel<A_WGRAD, B_ROW, 128, false>), grid, dim3(256), 0, st, p);
      if (hipGetLastError() != hipSuccess) return INDM_E_LAUNCH;
      hipLaunchKernelGGL(igemm_splitk_reduce, dim3(ew_grid((long)p.M * p.N, 256)), dim3(256), 0, st, p);
      return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
    }
  }
  if (am == A_CONV && (bm == B_ROW || bm == B_DGRAD) && ws && !p.G && p.a_vec && p.b_vec &&
      p.a_tap_uniform && (bm == B_ROW ? !(p.N & 3) : p.b_tap_uniform != 0) && p.M >= 1024 && p.N >= 64) {
    // mid-size convolutions (8x8 / 16x16 maps): too few 256 x 128 tiles to fill the chip -> split-K on the
    // direct-to-LDS kernel, chunks of at least 8 slabs
    const long tiles = (long)cdiv(p.M, 256) * cdiv(p.N, 128) * p.batch;
    if (tiles < 224 && tiles >= 16) {
      long want = tuned(448 / tiles);
      while (want > 1 && (p.K / want < 8 * BK || want * p.batch * p.M * p.N > ws_floats)) --want;
      if (want > 1) {
        p.kps = cdiv(cdiv(p.K, (int)want), BK) * BK;
        p.splitk = cdiv(p.K, p.kps);
        return bm == B_ROW ? launch_glds<A_CONV, B_ROW, false, 8>(p, st) : launch_glds<A_CONV, B_DGRAD, false, 8>(p, st);
      }
    }
  }
  if (am == A_ROW && (bm == B_ROW || bm == B_COL) && ws && p.batch == 1 && p.a_vec && p.b_vec && !(p.K & 15) &&
      p.K >= 512 && p.M >= 1024 && p.N >= 64 && (bm == B_COL || !(p.N & 3))) {
    // plain GEMMs with too few 256 x 128 tiles to fill the chip (the flow's scale-1 edge [32768, 512] x [512, 108],
    // gated or not: 128 tiles, one 4-wave block per CU on the 8-wave engine = 40 TF): split-K on the direct-to-LDS kernel
    const long tiles = (long)cdiv(p.M, 256) * cdiv(p.N, 128);
    // (N <= 128 with a long M goes to the 64 x 128 tiles of launch_tiles instead: 66 vs 76 us on [32768,512]x[512,108])
    if (tiles < 224 && tiles >= 16 && !(p.N <= 128 && p.M >= 8192)) {
      long want = tuned(448 / tiles);
      while (want > 1 && (p.K / want < 8 * BK || want * p.M * p.N > ws_floats)) --want;
      if (want > 1) {
        p.kps = cdiv(cdiv(p.K, (int)want), BK) * BK;
        p.splitk = cdiv(p.K, p.kps);
        if (bm == B_ROW) return p.G ? launch_glds<A_ROW, B_ROW, true, 8>(p, st) : launch_glds<A_ROW, B_ROW, false, 8>(p, st);
        return p.G ? launch_glds<A_ROW, B_COL, true, 8>(p, st) : launch_glds<A_ROW, B_COL, false, 8>(p, st);
      }
    }
  }
  p.splitk = ws ? choose_splitk(p.M, p.N, p.K, p.batch, ws_floats) : 1;
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

// Kernel variant chosen by this thread's last dense launch ("glds256x128,gated", "big256x256", "thin128x128", ...).
INDM_API const char* indm_igemm_last_variant(void) { return last_variant; }
// Library version: "<major>.<minor> gfx950" -- major = the build round that last changed the C ABI's existing entry points,
// minor = the round that last added entry points (the reference's native ops carry no version: op/upfirdn2d.cpp:22-23)
INDM_API const char* indm_version(void) { return "6.6 gfx950"; }

// 3x3 / pad 1 / stride 1 convolution with a SKINNY input (the flow's 3-/12-channel edges) as a thin-K GEMM whose operand
// loader gathers the 9 taps itself (no [pixels, 9C] panel in HBM):
//   y[p, co] = alpha*(sum_{tap,c} x[n, h - sign*(kh-1), w - sign*(kw-1), c] * w[tap*C + c, co] + bias[co] + bias2[n][co]) + beta*R
// sign = -1: cross-correlation (conv forward); sign = +1: the transposed convolution (w pre-packed by the caller).
// Returns 1 when the geometry is not covered (the caller then takes the tap_spread + GEMM route).
static int conv3x3_thin_impl(int bf16, const float* x, const float* w, float* y, int Nimg, int H, int W, int C, int Cout,
                             int sign, float alpha, const float* bias, const float* bias2, const float* R,
                             float beta, void* stream, float* act_d = nullptr) {
  if (Nimg < 0 || H <= 0 || W <= 0 || C <= 0 || Cout <= 0) return INDM_E_SHAPE;
  if (Nimg == 0) return INDM_OK;
  ThinGemmArgs t = {};
  t.bf16 = bf16;
  t.act_d = act_d;
  t.B = w; t.C = y; t.M = Nimg * H * W; t.N = Cout; t.K = 9 * C;
  t.ldb = Cout; t.ldc = Cout; t.batch = 1; t.alpha = alpha; t.beta = beta; t.bias = bias; t.bias2 = bias2; t.rpg = H * W;
  t.R = R; t.ldr = Cout; t.X = x; t.cH = H; t.cW = W; t.cC = C; t.csign = sign;
  const int r = indm_thin_conv3x3(t, (hipStream_t)stream);
  if (r == THIN_NA) return 1;
  last_variant = C <= 3 ? (bf16 ? "thin_conv3x3_flat,bf16" : "thin_conv3x3_flat") : (bf16 ? "thin_conv3x3_k128,bf16" : "thin_conv3x3_k128");
  return r;
}
INDM_API int indm_conv3x3_thin_f32(const float* x, const float* w, float* y, int Nimg, int H, int W, int C, int Cout,
                                   int sign, float alpha, const float* bias, const float* bias2, const float* R,
                                   float beta, void* stream) {
  return conv3x3_thin_impl(0, x, w, y, Nimg, H, W, C, Cout, sign, alpha, bias, bias2, R, beta, stream);
}
// same with both operands rounded to bf16 (the "bf16 compute / fp32 accumulate" mode)
INDM_API int indm_conv3x3_thin_bf16(const float* x, const float* w, float* y, int Nimg, int H, int W, int C, int Cout,
                                    int sign, float alpha, const float* bias, const float* bias2, const float* R,
                                    float beta, void* stream) {
  return conv3x3_thin_impl(1, x, w, y, Nimg, H, W, C, Cout, sign, alpha, bias, bias2, R, beta, stream);
}

// the same convolution with the flow's Sin activation as its epilogue (iresblock.py's g-net: conv1 -> Sin): a = sin(2 pi v) / (2 pi),
// d = cos(2 pi v), v = conv_sign(x, w) + bias[co] + bias2[n][co] -- v itself is not written.  Bit-identical to
// indm_conv3x3_thin_* followed by indm_sincos_act_fwd_f32.  Returns 1 (not an error) when the geometry is not covered by the
// row-fragment kernel (C = 3 / 12, W = 16 / 32, Cout % 128 == 0, pixels % 32 == 0): the caller then runs the two steps.
INDM_API int indm_conv3x3_thin_act_f32(const float* x, const float* w, float* a, float* d, int Nimg, int H, int W, int C, int Cout,
                                       int sign, const float* bias, const float* bias2, void* stream) {
  if (!a || !d) return INDM_E_ARG;
  return conv3x3_thin_impl(0, x, w, a, Nimg, H, W, C, Cout, sign, 1.0f, bias, bias2, nullptr, 0.0f, stream, d);
}
INDM_API int indm_conv3x3_thin_act_bf16(const float* x, const float* w, float* a, float* d, int Nimg, int H, int W, int C, int Cout,
                                        int sign, const float* bias, const float* bias2, void* stream) {
  if (!a || !d) return INDM_E_ARG;
  return conv3x3_thin_impl(1, x, w, a, Nimg, H, W, C, Cout, sign, 1.0f, bias, bias2, nullptr, 0.0f, stream, d);
}

// C[b] = alpha * (opA(A[b] .* gateA[b]) * opB(B[b]) + bias[n] + bias2[m/rpg][n]) + beta * R[b]   (gateA optional)
//   transA == 0: A is [M,K] row-major (lda);  transA == 1: A is stored [K,M] (lda)
//   transB == 0: B is [K,N] row-major (ldb);  transB == 1: B is stored [N,K] (ldb)
static int gemm_f32_impl(int bf16, const float* A, const float* gateA, const float* B, float* C, int M, int N, int K, int transA,
                           int transB, long lda, long ldb, long ldc, int batch, long strideA,
                           long strideB, long strideC, float alpha, const float* bias,
                           const float* bias2, int rows_per_group, const float* R, long ldr,
                           long strideR, float beta, float* ws, long ws_floats, void* stream) {
  if (M < 0 || N < 0 || K < 0 || batch < 0) return INDM_E_SHAPE;
  IgemmParams p = {};
  p.bf16 = bf16;
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
INDM_API int indm_gemm_f32(const float* A, const float* gateA, const float* B, float* C, int M, int N, int K, int transA,
                           int transB, long lda, long ldb, long ldc, int batch, long strideA,
                           long strideB, long strideC, float alpha, const float* bias,
                           const float* bias2, int rows_per_group, const float* R, long ldr,
                           long strideR, float beta, float* ws, long ws_floats, void* stream) {
  return gemm_f32_impl(0, A, gateA, B, C, M, N, K, transA, transB, lda, ldb, ldc, batch, strideA, strideB, strideC, alpha, bias, bias2, rows_per_group, R, ldr, strideR, beta, ws, ws_floats, stream);
}
// (a, d) = Sin(A B^T + bias[n] + bias2[m / rpg][n]): the g-net's second level (iresblock.py: Linear/1x1 LopConv -> Sin) in ONE
// launch -- the pre-activation is never written.  A [M, K] row-major (lda), B stored [N, K] (ldb), a / d [M, N] (ldc).
// Bit-identical to indm_gemm_f32 (transB = 1) followed by indm_sincos_act_fwd_f32.  Returns 1 (not an error) when the problem
// is not the direct-to-LDS kernel's (M >= 1024, N >= 64, K % 16 == 0, >= 224 tiles of 256 x 128, 16-byte aligned operands).
INDM_API int indm_gemm_sin_f32(const float* A, const float* B, float* a, float* d, int M, int N, int K, long lda, long ldb, long ldc,
                               const float* bias, const float* bias2, int rows_per_group, void* stream) {
  if (M < 0 || N < 0 || K < 0) return INDM_E_SHAPE;
  if (!a || !d) return INDM_E_ARG;
  if ((long)M * ldc >= (1L << 32)) return 1;
  IgemmParams p = {};
  p.A = A; p.B = B; p.C = a; p.act_d = d; p.M = M; p.N = N; p.K = K;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.batch = 1;
  p.alpha = 1.0f; p.beta = 0.0f; p.bias = bias; p.bias2 = bias2;
  p.rpg = rows_per_group > 0 ? rows_per_group : 1;
  p.a_vec = al16(A) && (lda % 4 == 0);
  p.b_vec = al16(B) && (ldb % 4 == 0);
  return run(p, A_ROW, B_COL, nullptr, 0, (hipStream_t)stream);
}
// One Sin level of the hand-written iResBlock backward with its producing GEMM: da = A B^T is NOT written,
//   dp = da * d - 4 pi^2 * a * (du * t),  dt = du * d (dt may be NULL),  seg[s][n] = sum of dp over the rows of segment s
// (seg may be NULL; else rows_per_seg % 64 == 0 and ws >= indm_ires_bwd_act_ws_floats(M, N) floats).  A [M, K] (lda), B stored
// [N, K] (ldb), every other tensor [M, N] contiguous.  Bit-identical to indm_gemm_f32 (transB = 1) + indm_ires_bwd_act_f32,
// segment sums included.  Returns 1 (not an error) when the direct-to-LDS kernel does not take the problem.
int indm_ires_seg_fold(const float* part, float* seg, long nseg, int cps, int C, hipStream_t st);   // elementwise.hip
INDM_API int indm_gemm_ires_bwd_act_f32(const float* A, const float* B, const float* du, const float* d, const float* a,
                                        const float* t, float* dp, float* dt, float* seg, int M, int N, int K, long lda, long ldb,
                                        long rows_per_seg, float* ws, void* stream) {
  if (M < 0 || N < 0 || K < 0) return INDM_E_SHAPE;
  if (!du || !d || !a || !t || !dp) return INDM_E_ARG;
  if (seg && (rows_per_seg <= 0 || rows_per_seg % 64 || M % rows_per_seg || !ws)) return INDM_E_ARG;
  if ((long)M * N >= (1L << 32) || (M & 63)) return 1;
  IgemmParams p = {};
  p.A = A; p.B = B; p.C = dp; p.M = M; p.N = N; p.K = K;
  p.lda = lda; p.ldb = ldb; p.ldc = N; p.batch = 1;
  p.alpha = 1.0f; p.beta = 0.0f; p.rpg = 1;
  p.e_du = du; p.e_d = d; p.e_a = a; p.e_t = t; p.e_dt = dt; p.e_part = seg ? ws : nullptr;
  p.a_vec = al16(A) && (lda % 4 == 0);
  p.b_vec = al16(B) && (ldb % 4 == 0);
  const int r = run(p, A_ROW, B_COL, nullptr, 0, (hipStream_t)stream);
  if (r != INDM_OK) return r;
  if (seg) return indm_ires_seg_fold(ws, seg, M / rows_per_seg, (int)(rows_per_seg / 64), N, (hipStream_t)stream);
  return INDM_OK;
}
// The same level when the GEMM produces du instead: du = (A .* gateA) B is written (its other consumers need it) and
//   dp = da * d - 4 pi^2 * a * (du * t),  seg[s][n] = per-segment sums of dp (optional)
// come out of the same launch.  A, gateA [M, K] (lda), B [K, N] row-major (ldb).  Bit-identical to indm_gemm_f32 (gated) followed
// by indm_ires_bwd_act_f32.  Returns 1 when the gated direct-to-LDS kernel does not take the problem.
INDM_API int indm_gemm_gated_ires_bwd_act_f32(const float* A, const float* gateA, const float* B, const float* da, const float* d,
                                              const float* a, const float* t, float* du, float* dp, float* seg, int M, int N,
                                              int K, long lda, long ldb, long rows_per_seg, float* ws, void* stream) {
  if (M < 0 || N < 0 || K < 0) return INDM_E_SHAPE;
  if (!gateA || !da || !d || !a || !t || !du || !dp) return INDM_E_ARG;
  if (seg && (rows_per_seg <= 0 || rows_per_seg % 64 || M % rows_per_seg || !ws)) return INDM_E_ARG;
  if ((long)M * N >= (1L << 32) || (M & 63)) return 1;
  IgemmParams p = {};
  p.A = A; p.G = gateA; p.B = B; p.C = du; p.M = M; p.N = N; p.K = K;
  p.lda = lda; p.ldb = ldb; p.ldc = N; p.batch = 1;
  p.alpha = 1.0f; p.beta = 0.0f; p.rpg = 1;
  p.e_da = da; p.e_d = d; p.e_a = a; p.e_t = t; p.e_dp = dp; p.e_part = seg ? ws : nullptr;
  p.a_vec = al16(A) && al16(gateA) && (lda % 4 == 0);
  p.b_vec = al16(B) && (ldb % 4 == 0);
  const int r = run(p, A_ROW, B_ROW, nullptr, 0, (hipStream_t)stream);
  if (r != INDM_OK) return r;
  if (seg) return indm_ires_seg_fold(ws, seg, M / rows_per_seg, (int)(rows_per_seg / 64), N, (hipStream_t)stream);
  return INDM_OK;
}
// same contraction with the MFMA inputs rounded to bf16 (fp32 tensors, fp32 accumulate)
INDM_API int indm_gemm_bf16(const float* A, const float* gateA, const float* B, float* C, int M, int N, int K, int transA,
                           int transB, long lda, long ldb, long ldc, int batch, long strideA,
                           long strideB, long strideC, float alpha, const float* bias,
                           const float* bias2, int rows_per_group, const float* R, long ldr,
                           long strideR, float beta, float* ws, long ws_floats, void* stream) {
  return gemm_f32_impl(1, A, gateA, B, C, M, N, K, transA, transB, lda, ldb, ldc, batch, strideA, strideB, strideC, alpha, bias, bias2, rows_per_group, R, ldr, strideR, beta, ws, ws_floats, stream);
}

// NHWC convolution forward:  y[n,oh,ow,co] = sum x[n, oh*s+kh-p, ow*s+kw-p, ci] * w[kh,kw,ci,co]
// (cross-correlation, same as torch conv2d). `in_dil` > 1 treats x as zero-inserted
// (used for the data-grad of strided convolutions). Weight layout is RSCK = [KH,KW,Cin,Cout].
static int conv2d_fwd_f32_impl(int bf16, const float* x, const float* w, float* y, int Nimg, int H, int W,
                                 int Cin, int Cout, int KH, int KW, int stride, int pad, int in_dil,
                                 int OH, int OW, float alpha, const float* bias, const float* bias2,
                                 const float* R, float beta, float* ws, long ws_floats, void* stream) {
  if (Nimg < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || OH <= 0 || OW <= 0) return INDM_E_SHAPE;
  IgemmParams p = {};
  p.bf16 = bf16;
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
  p.a_buf = p.dil == 1 && pad >= 0 && KH * KW <= 31 && ((long)Nimg * H * W + (long)pad * (W + 1)) * Cin * 4 < (1L << 31);
  return run(p, A_CONV, B_ROW, ws, ws_floats, (hipStream_t)stream);
}
INDM_API int indm_conv2d_fwd_f32(const float* x, const float* w, float* y, int Nimg, int H, int W,
                                 int Cin, int Cout, int KH, int KW, int stride, int pad, int in_dil,
                                 int OH, int OW, float alpha, const float* bias, const float* bias2,
                                 const float* R, float beta, float* ws, long ws_floats, void* stream) {
  return conv2d_fwd_f32_impl(0, x, w, y, Nimg, H, W, Cin, Cout, KH, KW, stride, pad, in_dil, OH, OW, alpha, bias, bias2, R, beta, ws, ws_floats, stream);
}
// same contraction with the MFMA inputs rounded to bf16 (fp32 tensors, fp32 accumulate)
INDM_API int indm_conv2d_fwd_bf16(const float* x, const float* w, float* y, int Nimg, int H, int W,
                                 int Cin, int Cout, int KH, int KW, int stride, int pad, int in_dil,
                                 int OH, int OW, float alpha, const float* bias, const float* bias2,
                                 const float* R, float beta, float* ws, long ws_floats, void* stream) {
  return conv2d_fwd_f32_impl(1, x, w, y, Nimg, H, W, Cin, Cout, KH, KW, stride, pad, in_dil, OH, OW, alpha, bias, bias2, R, beta, ws, ws_floats, stream);
}

// Data-grad of a stride-1 convolution (and, with in_dil = forward stride, of a strided one):
//   dx[n,ih,iw,ci] = sum_{kh,kw,co} dy[n, ih+pad'-..] * w[KH-1-kh, KW-1-kw, ci, co]
// dy: [Nimg,H,W,Cout] (H,W = dy spatial), dx: [Nimg,OH,OW,Cin].  pad' = K-1-pad_fwd.
static int conv2d_dgrad_f32_impl(int bf16, const float* dy, const float* w, float* dx, int Nimg, int H, int W,
                                   int Cin, int Cout, int KH, int KW, int pad_t, int in_dil, int OH,
                                   int OW, float alpha, const float* R, float beta, float* ws, long ws_floats,
                                   void* stream) {
  if (Nimg < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || OH <= 0 || OW <= 0) return INDM_E_SHAPE;
  IgemmParams p = {};
  p.bf16 = bf16;
  p.A = dy; p.B = w; p.C = dx;
  p.M = Nimg * OH * OW; p.N = Cin; p.K = KH * KW * Cout;
  p.ldc = Cin; p.batch = 1;
  p.H = H; p.W = W; p.Cin = Cout; p.Cout = Cout; p.OH = OH; p.OW = OW; p.KH = KH; p.KW = KW;
  p.stride = 1; p.pad = pad_t; p.dil = in_dil > 0 ? in_dil : 1;
  p.alpha = alpha; p.beta = beta; p.rpg = 1; p.R = R; p.ldr = Cin;
  p.a_vec = al16(dy) && (Cout % 4 == 0);
  p.b_vec = al16(w) && (Cout % 4 == 0);
  p.a_tap_uniform = p.b_tap_uniform = (Cout % BK == 0);
  p.a_buf = p.dil == 1 && pad_t >= 0 && KH * KW <= 31 && ((long)Nimg * H * W + (long)pad_t * (W + 1)) * Cout * 4 < (1L << 31);
  return run(p, A_CONV, B_DGRAD, ws, ws_floats, (hipStream_t)stream);
}
INDM_API int indm_conv2d_dgrad_f32(const float* dy, const float* w, float* dx, int Nimg, int H, int W,
                                   int Cin, int Cout, int KH, int KW, int pad_t, int in_dil, int OH,
                                   int OW, float alpha, const float* R, float beta, float* ws, long ws_floats,
                                   void* stream) {
  return conv2d_dgrad_f32_impl(0, dy, w, dx, Nimg, H, W, Cin, Cout, KH, KW, pad_t, in_dil, OH, OW, alpha, R, beta, ws, ws_floats, stream);
}
// same contraction with the MFMA inputs rounded to bf16 (fp32 tensors, fp32 accumulate)
INDM_API int indm_conv2d_dgrad_bf16(const float* dy, const float* w, float* dx, int Nimg, int H, int W,
                                   int Cin, int Cout, int KH, int KW, int pad_t, int in_dil, int OH,
                                   int OW, float alpha, const float* R, float beta, float* ws, long ws_floats,
                                   void* stream) {
  return conv2d_dgrad_f32_impl(1, dy, w, dx, Nimg, H, W, Cin, Cout, KH, KW, pad_t, in_dil, OH, OW, alpha, R, beta, ws, ws_floats, stream);
}

// Weight-grad: dw[kh,kw,ci,co] = alpha * sum_{n,oh,ow} x[n, oh*s+kh-p, ow*s+kw-p, ci] * dy[n,oh,ow,co]
//              (+ beta * R, same layout).  Reduction over pixels is split-K into `ws`.
static int conv2d_wgrad_f32_impl(int bf16, const float* x, const float* dy, float* dw, int Nimg, int H, int W,
                                   int Cin, int Cout, int KH, int KW, int stride, int pad, int OH,
                                   int OW, float alpha, const float* R, float beta, float* ws,
                                   long ws_floats, void* stream) {
  if (Nimg < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || OH <= 0 || OW <= 0) return INDM_E_SHAPE;
  IgemmParams p = {};
  p.bf16 = bf16;
  p.A = x; p.B = dy; p.C = dw;
  p.M = KH * KW * Cin; p.N = Cout; p.K = Nimg * OH * OW;
  p.ldb = Cout; p.ldc = Cout; p.batch = 1;
  p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.KH = KH; p.KW = KW;
  p.stride = stride; p.pad = pad; p.dil = 1;
  p.alpha = alpha; p.beta = beta; p.rpg = 1; p.R = R; p.ldr = Cout;
  p.a_vec = al16(x) && (Cin % 4 == 0);
  p.b_vec = al16(dy) && (Cout % 4 == 0);
  p.wg_inc = (OH * OW >= BK);
  p.ow_sh = p.ohw_sh = -1;
  if (!(OW & (OW - 1)) && !((OH * OW) & (OH * OW - 1))) {
    p.ow_sh = __builtin_ctz((unsigned)OW);
    p.ohw_sh = __builtin_ctz((unsigned)(OH * OW));
  }
  return run(p, A_WGRAD, B_ROW, ws, ws_floats, (hipStream_t)stream);
}
INDM_API int indm_conv2d_wgrad_f32(const float* x, const float* dy, float* dw, int Nimg, int H, int W,
                                   int Cin, int Cout, int KH, int KW, int stride, int pad, int OH,
                                   int OW, float alpha, const float* R, float beta, float* ws,
                                   long ws_floats, void* stream) {
  return conv2d_wgrad_f32_impl(0, x, dy, dw, Nimg, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW, alpha, R, beta, ws, ws_floats, stream);
}
// same contraction with the MFMA inputs rounded to bf16 (fp32 tensors, fp32 accumulate)
INDM_API int indm_conv2d_wgrad_bf16(const float* x, const float* dy, float* dw, int Nimg, int H, int W,
                                   int Cin, int Cout, int KH, int KW, int stride, int pad, int OH,
                                   int OW, float alpha, const float* R, float beta, float* ws,
                                   long ws_floats, void* stream) {
  return conv2d_wgrad_f32_impl(1, x, dy, dw, Nimg, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW, alpha, R, beta, ws, ws_floats, stream);
}

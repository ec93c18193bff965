// Fused single-head attention of NCSN++'s AttnBlockpp (models/layerspp.py:88-104) on bf16-STORED q / k / v ("H16", see
// unet_h16.hip): the bf16 form of attention.hip.
//
//   w = softmax(q k^T / sqrt(C)),  h = w v        q, k, v: bf16 rows of one [B T, ld] matrix (ld = 3 C: the fused QKV product),
//                                                 h, dh: bf16 [B T, C];  lse, D: fp32 [B, T]
// A wavefront owns 32 tokens and keeps its 32 x T strip of logits in registers from the first MFMA to the last:
//   phase 1   S^T[y][x] = sum_c Y[y][c] X[x][c] over 32-deep bf16 slabs ([row][64 B] LDS images filled by direct-to-LDS copies,
//             XOR-swizzled on the source address; a fragment is one ds_read_b128): lane (l31, lhi) ends up with
//             S[x = l31][y = 32 yb + 8 a + b + 4 lhi] in accumulator register 4 a + b of block yb
//   softmax   in registers (a lane owns half of its token's row, the other half sits in lane ^ 32), fp32
//   phase 2   out[x][c] = sum_y W[x][y] Z[y][c]: registers 8 a2 .. 8 a2 + 7 of a block, rounded to bf16, ARE the A operand of
//             one v_mfma_f32_32x32x16_bf16 (k position t <-> y = 32 yb + 16 a2 + 8 (t >> 2) + 4 lhi + (t & 3)); the matching B
//             operand comes out of the channel-contiguous Z tile ([64 y][128 c], memory order) through two transposing LDS
//             reads (ds_read_b64_tr_b16) whose four row addresses are exactly those y
// Three instantiations:  FWD (owner = queries) -> h, lse;  DQ (owner = queries) -> dq;  DKV (owner = keys) -> dv, dk, with
//   P = exp(s / sqrt(C) - lse[query]),  dS = P (dP - D[query]) / sqrt(C),  D[q] = <dh[q], h[q]>   (nothing T x T is stored).
#include "indm_common.h"
#pragma clang diagnostic ignored "-Winline-asm"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;

enum { FWD = 0, DQ = 1, DKV = 2 };

struct AttnArgs16 {
  const u16* Q; const u16* K; const u16* V;   // rows of stride ld
  const u16* dO;                              // [B T, C]
  const float* lse_in; const float* Dv;       // [B, T] (backward)
  u16* out0; u16* out1;                       // FWD: out0 = h (stride C);  DQ: out0 = dq;  DKV: out0 = dv, out1 = dk (stride ldg)
  float* lse_out;
  int T, C;
  long ld, ldg;
  float scale;
};

__device__ __forceinline__ void dma16(const void* src, void* dst_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)dst_wave_base, 16, 0, 0);
}
__device__ __forceinline__ float swap_pair(float v) {
  const int i = __float_as_int(v);
  return __int_as_float(__builtin_amdgcn_update_dpp(i, i, 0xB1, 0xF, 0xF, false));
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  const f32x2 v = {lo, hi};
  const bf16x2 h = __builtin_convertvector(v, bf16x2);
  return *(const unsigned*)&h;
}

// NB = T / 32 token blocks on the streamed side; WV waves per workgroup (each owns 32 tokens)
template <int MODE, int NB, int WV>
__global__ __launch_bounds__(WV * 64, 1) void attn16_kernel(AttnArgs16 p) {
  constexpr int T = NB * 32, OWN = WV * 32;
  constexpr bool BWD = MODE != FWD;
  constexpr int NARR = BWD ? 2 : 1;
  constexpr int ROWB = 64;                                       // bytes per row and slab (32 bf16)
  constexpr int Y_SZ = T * ROWB, X_SZ = OWN * ROWB;
  constexpr int BUF1 = NARR * (Y_SZ + X_SZ);                     // bytes per phase-1 buffer
  constexpr int ZT = 64, ZC = 128, ZROWB = ZC * 2, Z_SZ = ZT * ZROWB;   // phase-2 tile: [64 y][128 c] bf16
  extern __shared__ __attribute__((aligned(16))) char smem16[];
  float* lse_s = (float*)(smem16 + (2 * BUF1 > 2 * Z_SZ ? 2 * BUF1 : 2 * Z_SZ));   // DKV: lse / D of all queries
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5, swz = (l31 >> 2) & 3;
  const int b = blockIdx.y, x0 = blockIdx.x * OWN;
  const int C = p.C;
  const long ld = p.ld;
  const long img = (long)b * T * ld, imgo = (long)b * T * C;
  // phase-1 operands: S = Yq . Xq ; dP = Yd . Xd   (row strides: q / k / v ld, dO C)
  const u16* Yq = (MODE == DKV ? p.Q : p.K) + img;
  const u16* Xq = (MODE == DKV ? p.K : p.Q) + img + (long)x0 * ld;
  const u16* Yd = MODE == DKV ? p.dO + imgo : p.V + img;
  const u16* Xd = MODE == DKV ? p.V + img + (long)x0 * ld : p.dO + imgo + (long)x0 * C;
  const long ldYd = MODE == DKV ? C : ld, ldXd = MODE == DKV ? ld : C;

  if constexpr (MODE == DKV) {
    for (int i = tid; i < T; i += WV * 64) {
      lse_s[i] = p.lse_in[(long)b * T + i];
      lse_s[T + i] = p.Dv[(long)b * T + i];
    }
  }

  f32x16 S[NB], dP[BWD ? NB : 1];
#pragma unroll
  for (int y = 0; y < NB; ++y)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      S[y][r] = 0.f;
      if constexpr (BWD) dP[y][r] = 0.f;
    }

  // ---- phase 1: one 32-deep c-slab per stage (all waves share the copies: 16 rows per wave-instruction) ----
  constexpr int RB_Y = T / 16, RB_X = OWN / 16, RB_ALL = NARR * (RB_Y + RB_X);
  auto stage1 = [&](int buf, int kt) {
    char* base = smem16 + buf * BUF1;
#pragma unroll
    for (int n = 0; n < (RB_ALL + WV - 1) / WV; ++n) {
      const int ii = wave + n * WV;
      if (ii < RB_ALL) {
        const int arr = ii / (RB_Y + RB_X), rr = ii % (RB_Y + RB_X);
        const bool isx = rr >= RB_Y;
        const int rb = isx ? rr - RB_Y : rr;
        const int row = rb * 16 + (lane >> 2);
        const int ch = (lane & 3) ^ ((row >> 2) & 3);
        const u16* m = arr == 0 ? (isx ? Xq : Yq) : (isx ? Xd : Yd);
        const long lds = arr == 0 ? ld : (isx ? ldXd : ldYd);
        const u16* src = m + (long)row * lds + kt * 32 + 8 * ch;
        char* dst = base + arr * (Y_SZ + X_SZ) + (isx ? Y_SZ : 0) + rb * 16 * ROWB;
        dma16(src, dst);
      }
    }
  };
  auto compute1 = [&](int buf) {
    const char* base = smem16 + buf * BUF1;
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
      const int ch = ((sg * 2 + lhi) ^ swz) << 4;
      const bf16x8 xq = *(const bf16x8*)(base + Y_SZ + (wave * 32 + l31) * ROWB + ch);
      bf16x8 xd = xq;
      if constexpr (BWD) xd = *(const bf16x8*)(base + (Y_SZ + X_SZ) + Y_SZ + (wave * 32 + l31) * ROWB + ch);
#pragma unroll
      for (int y = 0; y < NB; ++y) {
        const bf16x8 yq = *(const bf16x8*)(base + (y * 32 + l31) * ROWB + ch);
        S[y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yq, xq, S[y], 0, 0, 0);
        if constexpr (BWD) {
          const bf16x8 yd = *(const bf16x8*)(base + (Y_SZ + X_SZ) + (y * 32 + l31) * ROWB + ch);
          dP[y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yd, xd, dP[y], 0, 0, 0);
          // (two y blocks of fragment reads in flight at most: hoisted further, the 16 + 16 fragments of a slab spill next
          // to the 256 accumulator registers)
          if constexpr (NB > 4) { if (y & 1) __builtin_amdgcn_sched_barrier(0); }
        }
      }
    }
  };
  const int nk = C / 32;
  stage1(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) stage1((kt + 1) & 1, kt + 1);
    compute1(kt & 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // lane (l31, lhi) now holds S[x = l31][y = 32 yb + 8 (r >> 2) + (r & 3) + 4 lhi]  (dP alike)

  // ---- softmax / dS in registers ----
  const float sc = p.scale;
  const int xg = x0 + wave * 32 + l31;  // this lane's owner token
  if constexpr (MODE == FWD) {
    float m = -INFINITY;
#pragma unroll
    for (int y = 0; y < NB; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, S[y][r]);
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int y = 0; y < NB; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        S[y][r] = __expf(sc * (S[y][r] - m));
        sum += S[y][r];
      }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
#pragma unroll
    for (int y = 0; y < NB; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) S[y][r] *= inv;
    if (lhi == 0) p.lse_out[(long)b * T + xg] = sc * m + __logf(sum);
  } else if constexpr (MODE == DQ) {
    const float lse = p.lse_in[(long)b * T + xg], Dx = p.Dv[(long)b * T + xg];
#pragma unroll
    for (int y = 0; y < NB; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pr = __expf(sc * S[y][r] - lse);
        dP[y][r] = pr * (dP[y][r] - Dx) * sc;  // dS
      }
  } else {
#pragma unroll
    for (int y = 0; y < NB; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = y * 32 + 8 * (r >> 2) + (r & 3) + 4 * lhi;
        const float pr = __expf(sc * S[y][r] - lse_s[q]);
        S[y][r] = pr;                                   // P   (-> dV)
        dP[y][r] = pr * (dP[y][r] - lse_s[T + q]) * sc;  // dS  (-> dK)
      }
  }

  // ---- phase 2: out[x][c] = sum_y W[x][y] Z[y][c], Z streamed in [64 y][128 c] bf16 tiles, double-buffered ----
  const int jj = (lane & 15) >> 2, g1 = (lane >> 4) & 1, c4 = lane & 3;
  auto phase2 = [&](const f32x16 (&W)[NB], const u16* Z, long ldz, u16* out, long ldout) {
    constexpr int TPC = NB / 2;  // tiles per 128-column chunk
    const int ncc = C / ZC;
    auto stage2 = [&](int par, int cc, int yt) {
      char* base = smem16 + par * Z_SZ;
#pragma unroll
      for (int n = 0; n < 16 / WV; ++n) {
        const int rp = wave + n * WV;                      // four rows (1 KB) per wave-instruction
        const int row = rp * 4 + (lane >> 4);
        const int csrc = (lane & 15) ^ ((row & 3) << 2);   // 64-byte units swizzled with (row & 3): see the transposing reads
        const u16* src = Z + (long)(yt * 64 + row) * ldz + cc * ZC + 8 * csrc;
        dma16(src, base + rp * 4 * ZROWB);
      }
    };
    __syncthreads();  // every wave is done with the phase-1 buffers / the previous pass's tiles
    stage2(0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int par = 0;
    for (int cc = 0; cc < ncc; ++cc) {
      f32x16 acc[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
      for (int yt = 0; yt < TPC; ++yt) {
        if (yt + 1 < TPC) stage2(par ^ 1, cc, yt + 1);
        else if (cc + 1 < ncc) stage2(par ^ 1, cc + 1, 0);
        const char* zb = smem16 + par * Z_SZ;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int y = yt * 2 + h;
#pragma unroll
          for (int a2 = 0; a2 < 2; ++a2) {
            const bf16x8 wa = indm_pack_bf16x8(W[y][8 * a2], W[y][8 * a2 + 1], W[y][8 * a2 + 2], W[y][8 * a2 + 3],
                                               W[y][8 * a2 + 4], W[y][8 * a2 + 5], W[y][8 * a2 + 6], W[y][8 * a2 + 7]);
            const char* zr = zb + (h * 32 + 16 * a2 + 4 * lhi + jj) * ZROWB;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const unsigned bo = ((unsigned)(j * 32 + 16 * g1 + 4 * c4) * 2u) ^ ((unsigned)jj << 6);
              const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(zr + bo));
              const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(zr + 8 * ZROWB + bo));
              const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
              acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, __builtin_bit_cast(bf16x8, v), acc[j], 0, 0, 0);
            }
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        par ^= 1;
      }
      // D'[x = 8 (r>>2) + (r&3) + 4 lhi][c = l31]: bf16 rows written as dwords (lanes c / c + 1 trade one value per row pair)
      const bool odd = lane & 1;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float give = odd ? acc[j][2 * q] : acc[j][2 * q + 1];
          const float got = swap_pair(give);
          const unsigned w = odd ? pack_bf16x2(got, acc[j][2 * q + 1]) : pack_bf16x2(acc[j][2 * q], got);
          const int r = 2 * q + (odd ? 1 : 0);
          const int x = x0 + wave * 32 + 8 * (r >> 2) + (r & 3) + 4 * lhi;
          *(unsigned*)(out + (long)x * ldout + cc * ZC + j * 32 + l31 - (odd ? 1 : 0)) = w;
        }
    }
  };
  if constexpr (MODE == FWD) {
    phase2(S, p.V + img, ld, p.out0 + imgo, C);
  } else if constexpr (MODE == DQ) {
    phase2(dP, p.K + img, ld, p.out0 + (long)b * T * p.ldg, p.ldg);
  } else {
    phase2(S, p.dO + imgo, C, p.out0 + (long)b * T * p.ldg, p.ldg);    // dV = P^T dO
    phase2(dP, p.Q + img, ld, p.out1 + (long)b * T * p.ldg, p.ldg);    // dK = dS^T Q
  }
}

template <int MODE, int NB, int WV>
int launch_attn16(const AttnArgs16& p, int B, hipStream_t st) {
  constexpr int T = NB * 32, OWN = WV * 32;
  constexpr int NARR = MODE == FWD ? 1 : 2;
  constexpr int BUF1 = NARR * (T + OWN) * 64;
  constexpr int main_b = (2 * BUF1 > 2 * 64 * 256 ? 2 * BUF1 : 2 * 64 * 256);
  constexpr size_t bytes = (size_t)main_b + (MODE == DKV ? 2 * T * sizeof(float) : 0);
  if (indm_lds_optin<attn16_kernel<MODE, NB, WV>>(bytes) != INDM_OK) return INDM_E_LAUNCH;
  hipLaunchKernelGGL((attn16_kernel<MODE, NB, WV>), dim3(T / OWN, B), dim3(WV * 64), bytes, st, p);
  return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
}

template <int MODE>
int dispatch_attn16(const AttnArgs16& p, int B, hipStream_t st) {
  switch (p.T) {
    case 256: return launch_attn16<MODE, 8, 4>(p, B, st);
    case 128: return launch_attn16<MODE, 4, 4>(p, B, st);
    case 64: return launch_attn16<MODE, 2, 2>(p, B, st);
    default: return 1;
  }
}

inline bool attn16_covered(int T, int C, long ld, const void* a) {
  return (T == 256 || T == 128 || T == 64) && C >= 128 && C % 128 == 0 && (ld % 8) == 0 && (((uintptr_t)a) & 15) == 0;
}

// D[r] = <a[r], b[r]> over C bf16 values per row: one wavefront per row (the query-wise <dh, h> of the attention backward)
__global__ __launch_bounds__(256) void h16_rowdot_kernel(const u16* __restrict__ a, const u16* __restrict__ b, float* __restrict__ out,
                                                          long R, int C) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  float s = 0.f;
  for (int c = lane * 2; c < C; c += 128) {
    const unsigned wa = *(const unsigned*)(a + row * C + c), wb = *(const unsigned*)(b + row * C + c);
    s += __uint_as_float(wa << 16) * __uint_as_float(wb << 16) + __uint_as_float(wa & 0xffff0000u) * __uint_as_float(wb & 0xffff0000u);
  }
  s = wave_sum(s);
  if (lane == 0) out[row] = s;
}

}  // namespace

// h[B T, C] (bf16) = softmax(q k^T * scale) v, lse[B, T] (fp32) = log-sum-exp of the scaled logits per query (saved for the
// backward).  q, k, v: bf16 matrices of row stride ld elements (the three column blocks of the fused QKV product: ld = 3 C, or
// separate tensors: ld = C).  Returns 1 when the geometry is not covered (T not in {64, 128, 256}, C % 128, ld % 8, alignment).
INDM_API int indm_h16_attn_fwd(const void* q, const void* k, const void* v, long ld, void* h, float* lse, int B, int T, int C,
                               float scale, void* stream) {
  if (B < 0 || T <= 0 || C <= 0) return INDM_E_SHAPE;
  if (B == 0) return INDM_OK;
  if (!attn16_covered(T, C, ld, q) || (((uintptr_t)k | (uintptr_t)v | (uintptr_t)h) & 15)) return 1;
  AttnArgs16 p = {};
  p.Q = (const u16*)q; p.K = (const u16*)k; p.V = (const u16*)v; p.out0 = (u16*)h; p.lse_out = lse; p.T = T; p.C = C; p.ld = ld;
  p.scale = scale;
  return dispatch_attn16<FWD>(p, B, (hipStream_t)stream);
}

// Backward: dq, dk, dv (bf16, row stride ldg) from q, k, v (stride ld), dh [B T, C], lse and D[B, T] = <dh[q], h[q]>
// (indm_h16_rowdot).  Logits and probabilities are recomputed in registers.
INDM_API int indm_h16_attn_bwd(const void* q, const void* k, const void* v, long ld, const void* dh, const float* lse,
                               const float* D, void* dq, void* dk, void* dv, long ldg, int B, int T, int C, float scale,
                               void* stream) {
  if (B < 0 || T <= 0 || C <= 0) return INDM_E_SHAPE;
  if (B == 0) return INDM_OK;
  if (!attn16_covered(T, C, ld, q) || (ldg % 8) ||
      (((uintptr_t)k | (uintptr_t)v | (uintptr_t)dh | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15)) return 1;
  AttnArgs16 p = {};
  p.Q = (const u16*)q; p.K = (const u16*)k; p.V = (const u16*)v; p.dO = (const u16*)dh; p.lse_in = lse; p.Dv = D;
  p.T = T; p.C = C; p.ld = ld; p.ldg = ldg; p.scale = scale;
  p.out0 = (u16*)dq;
  int r = dispatch_attn16<DQ>(p, B, (hipStream_t)stream);
  if (r != INDM_OK) return r;
  p.out0 = (u16*)dv; p.out1 = (u16*)dk;
  return dispatch_attn16<DKV>(p, B, (hipStream_t)stream);
}

// out[r] = <a[r], b[r]> for bf16 [R, C] (C even), fp32 out
INDM_API int indm_h16_rowdot(const void* a, const void* b, float* out, long R, int C, void* stream) {
  if (R < 0 || C <= 0 || (C & 1)) return INDM_E_SHAPE;
  if (R == 0) return INDM_OK;
  hipLaunchKernelGGL(h16_rowdot_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const u16*)a,
                     (const u16*)b, out, R, C);
  return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
}

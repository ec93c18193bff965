// Fused single-head attention of NCSN++'s AttnBlockpp (models/layerspp.py:88-104) for gfx950, fp32 MFMA.
//
//   w = softmax(q k^T / sqrt(C)),  h = w v        q, k, v, h: [B, T, C]   (T = H*W tokens, one head of width C)
//
// The reference (and round 1 of this repo) materialises the [B, T, T] logits and their softmax in HBM (33.5 MB per
// block at batch 128, again for the backward).  Here a wavefront OWNS 32 tokens and keeps its 32 x T strip of
// logits in registers from the first MFMA to the last:
//   phase 1   S^T[y][x] = sum_c Y[y][c] X[x][c]  over 16-wide slabs of c staged by global_load_lds into XOR-swizzled
//             [row][16] LDS images; the operand order is chosen so that lane (l31, lhi) ends up holding
//             S[x = l31][y = 8a + b + 4 lhi (+32 per block)] -- the A-operand layout of the next MFMA, so the
//             probabilities feed phase 2 straight from registers (no LDS transpose, no HBM round trip)
//   softmax   in registers: a lane owns half of its token's row, the other half sits in lane ^ 32
//   phase 2   out[x][c] = sum_y W[x][y] Z[y][c]  with Z streamed through LDS in [64 y][128 c] tiles
// Three instantiations of one template:
//   FWD   owner = queries:  S = K.Q,            W = P             Z = V   -> h, logsumexp per query
//   DQ    owner = queries:  S = K.Q, dP = V.dO, W = dS            Z = K   -> dq
//   DKV   owner = keys:     S = Q.K, dP = dO.V, W = P | dS        Z = dO | Q -> dv, dk
// with P = exp(s/sqrt(C) - lse[query]),  dS = P (dP - D[query]) / sqrt(C),  D[q] = <dO[q], h[q]>  (the usual
// recompute-in-backward formulation: nothing of size T x T is ever stored).
#include "indm_common.h"

namespace {

constexpr int BK = 16;
enum { FWD = 0, DQ = 1, DKV = 2 };

struct AttnArgs {
  const float* Q; const float* K; const float* V; const float* dO;
  const float* lse_in; const float* Dv;       // [B, T] (backward)
  float* out0; float* out1; float* lse_out;   // FWD: out0 = h;  DQ: out0 = dq;  DKV: out0 = dv, out1 = dk
  int T, C;
  float scale;
};

__device__ __forceinline__ void dma16(const float* src, float* dst_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)dst_wave_base, 16, 0, 0);
}

// NB = T / 32 token blocks on the streamed side; WV waves per workgroup (each owns 32 tokens)
template <int MODE, int NB, int WV>
__global__ __launch_bounds__(WV * 64, 1) void attn_kernel(AttnArgs p) {
  constexpr int T = NB * 32, OWN = WV * 32;
  constexpr bool BWD = MODE != FWD;
  constexpr int NARR = BWD ? 2 : 1;                              // operand pairs of phase 1
  constexpr int Y_SZ = T * BK, X_SZ = OWN * BK;
  constexpr int BUF1 = NARR * (Y_SZ + X_SZ);                     // floats per phase-1 buffer
  constexpr int ZT = 64, ZC = 128, Z_SZ = ZT * ZC;               // phase-2 tile
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* lse_s = smem + (2 * BUF1 > 2 * Z_SZ ? 2 * BUF1 : 2 * Z_SZ);  // DKV: lse / D of all queries
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5, swz = (l31 >> 2) & 3;
  const int b = blockIdx.y, x0 = blockIdx.x * OWN;
  const int C = p.C;
  const long img = (long)b * T * C;
  // phase-1 operands: S = Yq . Xq ; dP = Yd . Xd
  const float* Yq = (MODE == DKV ? p.Q : p.K) + img;
  const float* Xq = (MODE == DKV ? p.K : p.Q) + img + (long)x0 * C;
  const float* Yd = (MODE == DKV ? p.dO : p.V) + img;
  const float* Xd = (MODE == DKV ? p.V : p.dO) + img + (long)x0 * C;

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

  // ---- phase 1: DMA staging of one 16-wide c-slab (all waves share the work: 16 rows per wave-instruction) ----
  constexpr int RB_Y = T / 16, RB_X = OWN / 16, RB_ALL = NARR * (RB_Y + RB_X);
  auto stage1 = [&](int buf, int kt) {
    float* base = smem + buf * BUF1;
#pragma unroll
    for (int n = 0; n < (RB_ALL + WV - 1) / WV; ++n) {
      const int ii = wave + n * WV;
      if (ii < RB_ALL) {
        const int arr = ii / (RB_Y + RB_X), rr = ii % (RB_Y + RB_X);
        const bool isx = rr >= RB_Y;
        const int rb = isx ? rr - RB_Y : rr;
        const int row = rb * 16 + (lane >> 2);
        const int ch = (lane & 3) ^ ((row >> 2) & 3);
        const float* src = (arr == 0 ? (isx ? Xq : Yq) : (isx ? Xd : Yd)) + (long)row * C + kt * BK + 4 * ch;
        float* dst = base + arr * (Y_SZ + X_SZ) + (isx ? Y_SZ : 0) + rb * 16 * BK;
        dma16(src, dst);
      }
    }
  };
  auto compute1 = [&](int buf) {
    const float* base = smem + buf * BUF1;
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
      const int ch = ((sg * 2 + lhi) ^ swz) << 2;
      const float4 xq = *(const float4*)(base + Y_SZ + (wave * 32 + l31) * BK + ch);
      float4 xd = xq;
      if constexpr (BWD) xd = *(const float4*)(base + (Y_SZ + X_SZ) + Y_SZ + (wave * 32 + l31) * BK + ch);
#pragma unroll
      for (int y = 0; y < NB; ++y) {
        const float4 yq = *(const float4*)(base + (y * 32 + l31) * BK + ch);
        S[y] = __builtin_amdgcn_mfma_f32_32x32x2f32(yq.x, xq.x, S[y], 0, 0, 0);
        S[y] = __builtin_amdgcn_mfma_f32_32x32x2f32(yq.y, xq.y, S[y], 0, 0, 0);
        S[y] = __builtin_amdgcn_mfma_f32_32x32x2f32(yq.z, xq.z, S[y], 0, 0, 0);
        S[y] = __builtin_amdgcn_mfma_f32_32x32x2f32(yq.w, xq.w, S[y], 0, 0, 0);
        if constexpr (BWD) {
          const float4 yd = *(const float4*)(base + (Y_SZ + X_SZ) + (y * 32 + l31) * BK + ch);
          dP[y] = __builtin_amdgcn_mfma_f32_32x32x2f32(yd.x, xd.x, dP[y], 0, 0, 0);
          dP[y] = __builtin_amdgcn_mfma_f32_32x32x2f32(yd.y, xd.y, dP[y], 0, 0, 0);
          dP[y] = __builtin_amdgcn_mfma_f32_32x32x2f32(yd.z, xd.z, dP[y], 0, 0, 0);
          dP[y] = __builtin_amdgcn_mfma_f32_32x32x2f32(yd.w, xd.w, dP[y], 0, 0, 0);
        }
      }
    }
  };
  const int nk = C / BK;
  stage1(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) stage1((kt + 1) & 1, kt + 1);
    compute1(kt & 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // lane (l31, lhi) now holds S[x = l31][y = 32 yb + 8 (r>>2) + (r&3) + 4 lhi]  (dP alike)

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
        S[y][r] = expf(sc * (S[y][r] - m));
        sum += S[y][r];
      }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
#pragma unroll
    for (int y = 0; y < NB; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) S[y][r] *= inv;
    if (lhi == 0) p.lse_out[(long)b * T + xg] = sc * m + logf(sum);
  } else if constexpr (MODE == DQ) {
    const float lse = p.lse_in[(long)b * T + xg], Dx = p.Dv[(long)b * T + xg];
#pragma unroll
    for (int y = 0; y < NB; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pr = expf(sc * S[y][r] - lse);
        dP[y][r] = pr * (dP[y][r] - Dx) * sc;  // dS
      }
  } else {
#pragma unroll
    for (int y = 0; y < NB; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = y * 32 + 8 * (r >> 2) + (r & 3) + 4 * lhi;
        const float pr = expf(sc * S[y][r] - lse_s[q]);
        S[y][r] = pr;                                   // P   (-> dV)
        dP[y][r] = pr * (dP[y][r] - lse_s[T + q]) * sc;  // dS  (-> dK)
      }
  }

  // ---- phase 2: out[x][c] = sum_y W[x][y] Z[y][c], Z streamed in [64 y][128 c] tiles, double-buffered ----
  // tile sequence (cc, yt) row-major; tile n lives in LDS buffer n & 1; the DMA of tile n+1 overlaps the MFMAs of tile n
  auto phase2 = [&](const f32x16 (&W)[NB], const float* Z, float* out) {
    constexpr int TPC = NB / 2;  // tiles per 128-column chunk
    const int ncc = C / ZC;
    auto stage2 = [&](int par, int cc, int yt) {
      float* base = smem + par * Z_SZ;
#pragma unroll
      for (int n = 0; n < 32 / WV; ++n) {
        const int rp = wave + n * WV;  // row pair: one wave-instruction = 2 rows of 512 B
        const float* src = Z + (long)(yt * 64 + rp * 2 + (lane >> 5)) * C + cc * ZC + 4 * (lane & 31);
        dma16(src, base + rp * 2 * ZC);
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
        const float* zb = smem + par * Z_SZ;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int y = yt * 2 + h;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float* zr = zb + (h * 32 + 8 * (r >> 2) + (r & 3) + 4 * lhi) * ZC + l31;
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(W[y][r], zr[j * 32], acc[j], 0, 0, 0);
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        par ^= 1;
      }
      // D'[x = 8 (r>>2) + (r&3) + 4 lhi][c = l31]
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int x = x0 + wave * 32 + 8 * (r >> 2) + (r & 3) + 4 * lhi;
          out[(long)x * C + cc * ZC + j * 32 + l31] = acc[j][r];
        }
    }
  };
  if constexpr (MODE == FWD) {
    phase2(S, p.V + img, p.out0 + img);
  } else if constexpr (MODE == DQ) {
    phase2(dP, p.K + img, p.out0 + img);
  } else {
    phase2(S, p.dO + img, p.out0 + img);   // dV = P^T dO
    phase2(dP, p.Q + img, p.out1 + img);   // dK = dS^T Q
  }
}

template <int MODE, int NB, int WV>
int launch_attn(const AttnArgs& p, int B, hipStream_t st) {
  constexpr int T = NB * 32, OWN = WV * 32;
  constexpr int NARR = MODE == FWD ? 1 : 2;
  constexpr int BUF1 = NARR * (T + OWN) * BK;
  constexpr int main_f = (2 * BUF1 > 2 * 64 * 128 ? 2 * BUF1 : 2 * 64 * 128);
  constexpr size_t bytes = (size_t)(main_f + (MODE == DKV ? 2 * T : 0)) * sizeof(float);
  if (indm_lds_optin<attn_kernel<MODE, NB, WV>>(bytes) != INDM_OK) return INDM_E_LAUNCH;
  hipLaunchKernelGGL((attn_kernel<MODE, NB, WV>), dim3(T / OWN, B), dim3(WV * 64), bytes, st, p);
  return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
}

template <int MODE>
int dispatch_attn(const AttnArgs& p, int B, hipStream_t st) {
  switch (p.T) {
    case 256: return launch_attn<MODE, 8, 4>(p, B, st);
    case 128: return launch_attn<MODE, 4, 4>(p, B, st);
    case 64: return launch_attn<MODE, 2, 2>(p, B, st);
    default: return 1;
  }
}

inline bool attn_covered(int T, int C, const void* a, const void* b, const void* c) {
  return (T == 256 || T == 128 || T == 64) && C >= 128 && C % 128 == 0 &&
         ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) == 0);
}

}  // namespace

// h[B,T,C] = softmax(q k^T * scale) v ; lse[B,T] = log-sum-exp of the scaled logits per query (saved for the backward).
// Returns 1 when the geometry is not covered (T not in {64,128,256} or C % 128 != 0): the caller uses the unfused path.
INDM_API int indm_attn_fwd_f32(const float* q, const float* k, const float* v, float* h, float* lse, int B, int T, int C,
                               float scale, void* stream) {
  if (B < 0 || T <= 0 || C <= 0) return INDM_E_SHAPE;
  if (B == 0) return INDM_OK;
  if (!attn_covered(T, C, q, k, v) || ((uintptr_t)h & 15)) return 1;
  AttnArgs p = {};
  p.Q = q; p.K = k; p.V = v; p.out0 = h; p.lse_out = lse; p.T = T; p.C = C; p.scale = scale;
  return dispatch_attn<FWD>(p, B, (hipStream_t)stream);
}

// Backward of indm_attn_fwd_f32.  Dv[B,T] = <dh[q], h[q]> per query (row dots, computed by the caller with
// indm_rowreduce_f32).  dq, dk, dv: [B,T,C].
INDM_API int indm_attn_bwd_f32(const float* q, const float* k, const float* v, const float* dh, const float* lse,
                               const float* Dv, float* dq, float* dk, float* dv, int B, int T, int C, float scale,
                               void* stream) {
  if (B < 0 || T <= 0 || C <= 0) return INDM_E_SHAPE;
  if (B == 0) return INDM_OK;
  if (!attn_covered(T, C, q, k, v) || (((uintptr_t)dh | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15)) return 1;
  AttnArgs p = {};
  p.Q = q; p.K = k; p.V = v; p.dO = dh; p.lse_in = lse; p.Dv = Dv; p.T = T; p.C = C; p.scale = scale;
  p.out0 = dq;
  int r = dispatch_attn<DQ>(p, B, (hipStream_t)stream);
  if (r != INDM_OK) return r;
  p.out0 = dv; p.out1 = dk;
  return dispatch_attn<DKV>(p, B, (hipStream_t)stream);
}

// bf16-STORAGE contractions of the residual flow (BASELINE configs #3-#5), direct-to-LDS staging:
//
//   C[M, N] = alpha * ( (A .* G)[M, K] . W[N, K]^T + bias[n] + bias2[m / rpg][n] )
//
// A, G (optional gate): bf16 [M, K] row-major; W: bf16 [N, K] row-major (k-contiguous: the weight copy the host converts
// once per flow call); C: bf16 or fp32 [M, N]; bias / bias2: fp32.  Same structure as igemm_glds_kernel (igemm.hip):
// 256 x 128 block tile, 8 waves of 64 x 64, global_load_lds_dwordx4 into XOR-swizzled [row][64 B] LDS images, double
// buffered, one barrier per slab -- but a 64-byte row slab now holds 32 bf16 k-values, so a lane's 16-byte fragment
// read IS the 8-deep operand of v_mfma_f32_32x32x16_bf16: no conversion in the loop (the gate costs an unpack / multiply /
// repack of 8 values).  The kernel streams half the bytes of the fp32-storage bf16 path of igemm.hip.
#include "indm_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int TM = 256, TN = 128, NW = 8, KB = 32;                 // KB: bf16 k-values per slab (64 B per row)
constexpr int A_B = TM * 64, B_B = TN * 64;                         // bytes per LDS image of one slab

struct HgemmArgs {
  const __bf16* A; const __bf16* G; const __bf16* W;
  void* C;
  int M, N, K;
  float alpha;
  const float* bias; const float* bias2;
  int rpg;
};

__device__ __forceinline__ bf16x8 gate8(bf16x8 a, bf16x8 g) {
  bf16x8 o;
#pragma unroll
  for (int t = 0; t < 8; ++t) o[t] = (__bf16)((float)a[t] * (float)g[t]);
  return o;
}

template <bool HAS_G, bool OUT_H>
__global__ __launch_bounds__(NW * 64, 4) void hgemm_kernel(HgemmArgs p) {
  constexpr int BUF = A_B + (HAS_G ? A_B : 0) + B_B;                // bytes per buffer: 40 KB gated, 24 KB plain
  extern __shared__ __attribute__((aligned(16))) char hsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (p.N + TN - 1) / TN;
  // XCD-aware order: the N-tiles that share an A row-panel are neighbours on one XCD's L2 (as in igemm.hip)
  const int ntile = ((p.M + TM - 1) / TM) * tiles_n;
  int lt;
  {
    const int q = ntile >> 3, r = ntile & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    lt = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = lt / tiles_n, tn = lt - tm * tiles_n;
  const int m0 = tm * TM, n0 = tn * TN;
  const int nk = p.K / KB;
  const int l31 = lane & 31, lhi = lane >> 5, swz = (l31 >> 2) & 3;
  const int wm = wave >> 1, wn = wave & 1, mw = wm * 64, nw = wn * 64;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto dma = [&](const char* src, char* dst_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst_wave_base, 16, 0, 0);
  };
  // per-lane running source pointers (bytes): a wave stages 32 rows of A (2 instructions of 16 rows), the same rows of the
  // gate, and 16 rows of W; lane -> (row = lane >> 2, 16-byte slot = lane & 3) holding k-chunk slot ^ ((row >> 2) & 3)
  const char* pa[2];
  const char* pg[2];
  const char* pb;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int row = wave * 32 + q * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((row >> 2) & 3);
    const long off = ((long)min(m0 + row, p.M - 1) * p.K) * 2 + 16 * c;
    pa[q] = (const char*)p.A + off;
    pg[q] = HAS_G ? (const char*)p.G + off : nullptr;
  }
  {
    const int row = wave * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((row >> 2) & 3);
    pb = (const char*)p.W + ((long)min(n0 + row, p.N - 1) * p.K) * 2 + 16 * c;
  }
  auto stage = [&](int buf) {
    char* base = hsm + buf * BUF;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int r0 = wave * 32 + q * 16;
      dma(pa[q], base + r0 * 64);
      pa[q] += 64;
      if constexpr (HAS_G) {
        dma(pg[q], base + A_B + r0 * 64);
        pg[q] += 64;
      }
    }
    dma(pb, base + A_B + (HAS_G ? A_B : 0) + wave * 16 * 64);
    pb += 64;
  };

  stage(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int s = 0; s < nk; ++s) {
    const int buf = s & 1;
    if (s + 1 < nk) stage(buf ^ 1);
    const char* as = hsm + buf * BUF;
    const char* bs = as + A_B + (HAS_G ? A_B : 0);
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const int ch = ((st * 2 + lhi) ^ swz) * 16;
      bf16x8 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int o = (mw + i * 32 + l31) * 64 + ch;
        fa[i] = *(const bf16x8*)(as + o);
        if constexpr (HAS_G) fa[i] = gate8(fa[i], *(const bf16x8*)(as + A_B + o));
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = *(const bf16x8*)(bs + (nw + j * 32 + l31) * 64 + ch);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue: D[m = 8 (r>>2) + (r&3) + 4 lhi][n = l31] per 32 x 32 block ----
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + nw + j * 32 + l31;
      if (n >= p.N) continue;
      const float bn = p.bias ? p.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (m >= p.M) continue;
        float v = acc[i][j][r] + bn;
        if (p.bias2) v += p.bias2[(long)(m / p.rpg) * p.N + n];
        v *= p.alpha;
        if constexpr (OUT_H) ((__bf16*)p.C)[(long)m * p.N + n] = (__bf16)v;
        else ((float*)p.C)[(long)m * p.N + n] = v;
      }
    }
  }
}

template <bool HAS_G, bool OUT_H>
int launch_hgemm(const HgemmArgs& p, hipStream_t st) {
  constexpr int BUF = A_B + (HAS_G ? A_B : 0) + B_B;
  constexpr size_t bytes = 2 * BUF;
  static bool ready = false;
  if (!ready) {
    if (hipFuncSetAttribute((const void*)hgemm_kernel<HAS_G, OUT_H>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)bytes) != hipSuccess)
      return INDM_E_LAUNCH;
    ready = true;
  }
  const long tiles = (long)cdiv(p.M, TM) * cdiv(p.N, TN);
  hipLaunchKernelGGL((hgemm_kernel<HAS_G, OUT_H>), dim3((unsigned)tiles), dim3(NW * 64), bytes, st, p);
  return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
}

}  // namespace

// bf16-storage GEMM of the flow: C = alpha * ((A .* G) W^T + bias + bias2[m / rpg]).  A, G: bf16 [M, K]; W: bf16 [N, K];
// C: bf16 (out_bf16 = 1) or fp32 [M, N]; K % 32 == 0; 16-byte aligned pointers.  fp32 accumulation.
INDM_API int indm_hgemm_bf16(const void* A, const void* G, const void* W, void* C, int M, int N, int K, float alpha,
                             const float* bias, const float* bias2, int rows_per_group, int out_bf16, void* stream) {
  if (M < 0 || N <= 0 || K <= 0) return INDM_E_SHAPE;
  if (M == 0) return INDM_OK;
  if ((K % KB) || (((uintptr_t)A | (uintptr_t)G | (uintptr_t)W | (uintptr_t)C) & 15)) return INDM_E_ARG;
  HgemmArgs p = {(const __bf16*)A, (const __bf16*)G, (const __bf16*)W, C, M, N, K, alpha, bias, bias2,
                 rows_per_group > 0 ? rows_per_group : 1};
  hipStream_t st = (hipStream_t)stream;
  if (G) return out_bf16 ? launch_hgemm<true, true>(p, st) : launch_hgemm<true, false>(p, st);
  return out_bf16 ? launch_hgemm<false, true>(p, st) : launch_hgemm<false, false>(p, st);
}

// ---------------------------------------------------------------------------------------------------------------
// bf16-storage version of the back-to-back VJP-chain kernel (ires_chain.hip):
//   P[M, J] = ((A .* G) W^T .* D1) W1p[:, :J],  optional T2[M, 512] = (A .* G) W^T   (bf16)
// A, G: bf16 [M, K]; W: bf16 [512, K] (k-contiguous: the [Cin, Cout] weight itself); D1: bf16 [M, 512]; W1p: fp32
// [512, 32 | 128]; P: fp32.  64 x 512 block tile, stage 1 with swapped MFMA operands (accumulator = A-operand layout of
// stage 2), gate and stage 2 exactly as in the fp32-storage kernel, operands rounded to bf16.
namespace {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int CM = 64, CN = 512;
constexpr int CA_B = CM * 64, CW_B = CN * 64, CBUF = 2 * CA_B + CW_B;   // 4 + 4 + 32 KB

struct HchainArgs {
  const __bf16* A; const __bf16* G; const __bf16* W; const __bf16* D1;
  const float* W1p;
  float* P;
  __bf16* T2;
  int M, K, J;
};

__device__ __forceinline__ float bfr(float x) { return (float)(__bf16)x; }

template <int JB>
__global__ __launch_bounds__(NW * 64, 4) void hchain_kernel(HchainArgs p) {
  extern __shared__ __attribute__((aligned(16))) char hsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * CM;
  const int nk = p.K / KB;
  const int l31 = lane & 31, lhi = lane >> 5, swz = (l31 >> 2) & 3;
  const int nw = wave * 64;

  f32x16 acc[2][2];  // [n-block][m-block]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto dma = [&](const char* src, char* dst_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst_wave_base, 16, 0, 0);
  };
  // waves 0-3: 16 rows of A each; waves 4-7: the same rows of the gate; every wave: 64 rows of W (4 instructions)
  const int arow = (wave & 3) * 16 + (lane >> 2);
  const char* pa = (const char*)(wave < 4 ? p.A : p.G) + ((long)(m0 + arow) * p.K) * 2 + 16 * ((lane & 3) ^ ((arow >> 2) & 3));
  const int a_dst = (wave < 4 ? 0 : CA_B) + (wave & 3) * 16 * 64;
  const char* pw[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = wave * 64 + q * 16 + (lane >> 2);
    pw[q] = (const char*)p.W + ((long)row * p.K) * 2 + 16 * ((lane & 3) ^ ((row >> 2) & 3));
  }
  auto stage = [&](int buf) {
    char* base = hsm + buf * CBUF;
    dma(pa, base + a_dst);
    pa += 64;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      dma(pw[q], base + 2 * CA_B + (wave * 64 + q * 16) * 64);
      pw[q] += 64;
    }
  };

  stage(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int s = 0; s < nk; ++s) {
    const int buf = s & 1;
    if (s + 1 < nk) stage(buf ^ 1);
    const char* as = hsm + buf * CBUF;
    const char* ws = as + 2 * CA_B;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const int ch = ((st * 2 + lhi) ^ swz) * 16;
      bf16x8 fm[2], fn[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int o = (j * 32 + l31) * 64 + ch;
        fm[j] = gate8(*(const bf16x8*)(as + o), *(const bf16x8*)(as + CA_B + o));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) fn[i] = *(const bf16x8*)(ws + (nw + i * 32 + l31) * 64 + ch);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fn[i], fm[j], acc[i][j], 0, 0, 0);  // D[n][m]
      __builtin_amdgcn_sched_barrier(0);  // one k-step's fragments at a time: <= 128 VGPRs, no accumulator rotation
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // acc[i][j][r] = t2[m = m0 + 32 j + l31][n = nw + 32 i + 8 (r>>2) + (r&3) + 4 lhi]
  if (p.T2) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      __bf16* trow = p.T2 + (long)(m0 + j * 32 + l31) * CN + nw + 4 * lhi;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          bf16x4 v;
#pragma unroll
          for (int b = 0; b < 4; ++b) v[b] = (__bf16)acc[i][j][4 * a + b];
          *(bf16x4*)(trow + i * 32 + 8 * a) = v;
        }
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const __bf16* drow = p.D1 + (long)(m0 + j * 32 + l31) * CN + nw + 4 * lhi;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      bf16x4 g[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) g[a] = *(const bf16x4*)(drow + i * 32 + 8 * a);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[i][j][4 * a + b] = bfr(bfr(acc[i][j][4 * a + b]) * (float)g[a][b]);  // t2 as stored (bf16), gated, rounded
    }
  }

  const int JP = 32 * JB;
  float* red = (float*)hsm;  // [8 waves][64 m][32 j] = 64 KB (the operand buffers are dead)
#pragma unroll 1
  for (int jb = 0; jb < JB; ++jb) {
    f32x16 out[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) out[j][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float w[16];
#pragma unroll
      for (int r = 0; r < 16; ++r)
        w[r] = bfr(p.W1p[(long)(nw + i * 32 + 8 * (r >> 2) + (r & 3) + 4 * lhi) * JP + jb * 32 + l31]);
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          out[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[i][j][r], w[r], out[j], 0, 0, 0);
    }
    if (jb > 0) __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = j * 32 + 8 * (r >> 2) + (r & 3) + 4 * lhi;
        red[(wave * 64 + m) * 32 + l31] = out[j][r];
      }
    __syncthreads();
    {
      const int m = tid >> 3, j0 = (tid & 7) * 4;
      float4 sm = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int wv = 0; wv < NW; ++wv) {
        const float4 v = *(const float4*)(red + (wv * 64 + m) * 32 + j0);
        sm.x += v.x; sm.y += v.y; sm.z += v.z; sm.w += v.w;
      }
      float* dst = p.P + (long)(m0 + m) * p.J + jb * 32 + j0;
      const int left = p.J - (jb * 32 + j0);
      if (left > 0) dst[0] = sm.x;
      if (left > 1) dst[1] = sm.y;
      if (left > 2) dst[2] = sm.z;
      if (left > 3) dst[3] = sm.w;
    }
  }
}

template <int JB>
int launch_hchain(const HchainArgs& p, hipStream_t st) {
  constexpr size_t bytes = 2 * CBUF;
  static bool ready = false;
  if (!ready) {
    if (hipFuncSetAttribute((const void*)hchain_kernel<JB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) !=
        hipSuccess)
      return INDM_E_LAUNCH;
    ready = true;
  }
  hipLaunchKernelGGL(hchain_kernel<JB>, dim3(p.M / CM), dim3(NW * 64), bytes, st, p);
  return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
}

}  // namespace

// bf16-storage back-to-back chain.  Requirements: M % 64 == 0, K % 32 == 0, K >= 64, N == 512, J <= 128.  Returns 1 when
// the geometry is not covered.
INDM_API int indm_hchain_bf16(const void* A, const void* G, const void* W, const void* D1, const float* W1p, float* P,
                              void* T2, int M, int K, int N, int J, void* stream) {
  if (M < 0 || K <= 0 || N <= 0 || J <= 0) return INDM_E_SHAPE;
  if (M == 0) return INDM_OK;
  if (N != CN || (M % CM) || (K % KB) || K < 2 * KB || J > 128) return 1;
  if (((uintptr_t)A | (uintptr_t)G | (uintptr_t)W | (uintptr_t)D1 | (uintptr_t)W1p | (uintptr_t)T2) & 15) return 1;
  HchainArgs p = {(const __bf16*)A, (const __bf16*)G, (const __bf16*)W, (const __bf16*)D1, W1p, P, (__bf16*)T2, M, K, J};
  return J <= 32 ? launch_hchain<1>(p, (hipStream_t)stream) : launch_hchain<4>(p, (hipStream_t)stream);
}

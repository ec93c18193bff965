// Thin GEMMs (see igemm_thin.h): HBM-streaming contractions of the flow's 3-/12-channel edges on gfx950.
//
//   thin-K   C[M,N] = epi(A[M,K] B[K,N]),  K <= 128, N >= 128     (3x3 conv with a skinny INPUT after tap_spread:
//            K = 27 / 108).  Write-bound: 268 MB of C per launch at M = 131072.  The generic engine spends its time on
//            scalar operand loads (lda = 27 is not float4-aligned) and on 64 dword-store instructions per wave; here the
//            A tile is fetched as ONE contiguous run, the whole reduction sits in LDS at once, and the accumulators go
//            back through LDS so that every store instruction writes two full 512-byte rows (float4 per lane).
#include "igemm_thin.h"
#pragma clang diagnostic ignored "-Winline-asm"   // M0 clobber declared on the copy helper (see igemm.hip)

namespace {
// (the literals of elementwise.hip: the fused Sin epilogue must round like indm_sincos_act_fwd_f32)
constexpr float TWO_PI = 6.283185307179586f;
constexpr float INV_TWO_PI = 0.15915494309189535f;

template <bool ON>
__device__ __forceinline__ float rb16(float x) {
  if constexpr (ON) return (float)(__bf16)x;
  return x;
}
}  // namespace

namespace {

// Bijective XCD-aware remap (same as igemm.hip): workgroup b runs on XCD b % 8; give each XCD a contiguous run of tiles.
__device__ __forceinline__ int xcd_remap(int b, int n) {
  const int q = n >> 3, r = n & 7;
  const int xcd = b & 7, idx = b >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

inline bool al16(const void* q) { return ((uintptr_t)q & 15) == 0; }

// ---------------------------------------------------------------------------------------------------------------
// thin-K kernel.  128 x 128 tile, 4 waves, wave w owns rows [32w, 32w+32) x 128 columns (1 x 4 MFMA 32x32 tiles).
// FLAT: lda == K <= 32 -> the tile's A rows are one contiguous run of 128*K floats (float4 loads whatever K is).
// otherwise K % 4 == 0 and lda % 4 == 0, the reduction walks in chunks of 32.
constexpr int TK_TM = 128, TK_TN = 128, TK_KC = 32;
constexpr int TK_LDA = TK_TM + 2;   // [k][row], transposing ds_write_b32: 2*(k4) + row distinct over a half-wave
constexpr int TK_LDB = TK_TN + 4;   // [k][col], 16-byte aligned rows: one ds_write_b128 per float4
constexpr int TK_LDS = TK_TN + 8;   // staging rows for the epilogue (16-byte aligned; 4 rows apart = 32 banks apart)
constexpr int TK_OPER = TK_KC * (TK_LDA + TK_LDB);
constexpr int TK_STAGE = 4 * 16 * TK_LDS;
constexpr int TK_SMEM = TK_OPER > TK_STAGE ? TK_OPER : TK_STAGE;

enum { TK_STRIDED = 0, TK_FLAT = 1, TK_IM2COL = 2 };

template <int MODE, bool BF16 = false>
__global__ __launch_bounds__(256, 4) void igemm_thin_k_kernel(ThinGemmArgs p) {
  constexpr bool FLAT = MODE == TK_FLAT;
  __shared__ __attribute__((aligned(16))) float smem[TK_SMEM];
  float* as = smem;
  float* bs = smem + TK_KC * TK_LDA;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int tiles_n = (p.N + TK_TN - 1) / TK_TN, tiles_m = (p.M + TK_TM - 1) / TK_TM;
  const int lt = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int tm = lt / tiles_n, tn = lt - tm * tiles_n;
  const int m0 = tm * TK_TM, n0 = tn * TK_TN;

  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  for (int k0 = 0; k0 < p.K; k0 += TK_KC) {
    const int kc = min(TK_KC, p.K - k0);
    if (k0) __syncthreads();  // previous chunk's fragments are consumed
    // ---- B chunk: [kc x 128] float4 row groups (zero beyond K / N)
    // Two-phase staging: ALL loads of the chunk (B and A), from clamped addresses, then all LDS stores.  Written as
    // `if (inside) v = load; store v` per group, every load was a branch with `s_waitcnt vmcnt(0)` behind it (8 memory
    // latencies per chunk, one after the other); unconditional loads alone are not enough either -- hipcc keeps each
    // load -> LDS-store pair together -- so the loads go into a register array first and a scheduling barrier ends the phase.
    float4 vb[4], va[4];
    bool okb[4], oka[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int g = tid + j * 256;
      const int k = g >> 5, c4 = (g & 31) * 4;
      okb[j] = k < kc && n0 + c4 < p.N;
      vb[j] = *(const float4*)(p.B + (okb[j] ? (long)(k0 + k) * p.ldb + n0 + c4 : 0L));
    }
    // ---- A chunk
    if constexpr (FLAT) {
      const int rows = min(TK_TM, p.M - m0);
      const int total = rows * p.K;
      const float* src = p.A + (long)m0 * p.K;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e0 = (tid + j * 256) * 4;
        if (e0 < total) {
          float v[4] = {0.f, 0.f, 0.f, 0.f};
          if (e0 + 3 < total) {
            const float4 q = *(const float4*)(src + e0);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
          } else {
            for (int t = 0; t < 4; ++t)
              if (e0 + t < total) v[t] = src[e0 + t];
          }
          int row = e0 / p.K, k = e0 - row * p.K;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            if (e0 + t < total) as[k * TK_LDA + row] = v[t];
            if (++k == p.K) { k = 0; ++row; }
          }
        }
      }
      if ((p.K & 1) && tid < TK_TM) as[p.K * TK_LDA + tid] = 0.f;  // odd K: the last k-pair's second half
    } else if constexpr (MODE == TK_IM2COL) {
      // A[p, tap*C + c] gathered from the NHWC image: four consecutive k stay inside one tap (C % 4 == 0)
      const int C = p.cC, W = p.cW, H = p.cH;
      const float rC = 1.0f / (float)C;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int g = tid + j * 256;
        const int row = g >> 3, k4 = (g & 7) * 4;
        const int pix = m0 + row, kk = k0 + k4;
        const int tap = (int)(((float)kk + 0.5f) * rC);  // exact for kk < 128
        const int c = kk - tap * C;
        const int kh = (tap * 11) >> 5, kw = tap - 3 * kh;  // tap / 3 for tap < 9
        const int hw = H * W;
        const int n = pix / hw, r = pix - n * hw;
        const int h = r / W, w = r - h * W;
        const int hh = h - p.csign * (kh - 1), ww = w - p.csign * (kw - 1);
        oka[j] = pix < p.M && k4 < kc && hh >= 0 && hh < H && ww >= 0 && ww < W;
        va[j] = *(const float4*)(p.X + (oka[j] ? (((long)n * H + hh) * W + ww) * C + c : 0L));
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int g = tid + j * 256;
        const int row = g >> 3, k4 = (g & 7) * 4;
        oka[j] = m0 + row < p.M && k4 < kc;
        va[j] = *(const float4*)(p.A + (oka[j] ? (long)(m0 + row) * p.lda + k0 + k4 : 0L));
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // end of the load phase
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int g = tid + j * 256;
      const int k = g >> 5, c4 = (g & 31) * 4;
      *(float4*)(bs + k * TK_LDB + c4) = okb[j] ? vb[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if constexpr (!FLAT) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int g = tid + j * 256;
        const int row = g >> 3, k4 = (g & 7) * 4;
        const float4 v = oka[j] ? va[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        as[(k4 + 0) * TK_LDA + row] = v.x;
        as[(k4 + 1) * TK_LDA + row] = v.y;
        as[(k4 + 2) * TK_LDA + row] = v.z;
        as[(k4 + 3) * TK_LDA + row] = v.w;
      }
    }
    __syncthreads();
    // ---- MFMAs: k-pairs of this chunk
    const float* af = as + wave * 32 + l31;
    const float* bf = bs + l31;
    const int npairs = (kc + 1) >> 1;
#pragma unroll 2
    for (int kk = 0; kk < npairs; ++kk) {
      const int k = 2 * kk + lhi;
      const float a = rb16<BF16>(af[k * TK_LDA]);
      float b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = rb16<BF16>(bf[k * TK_LDB + j * 32]);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[j], acc[j], 0, 0, 0);
    }
  }

  // ---- epilogue: accumulators -> LDS (row-major, per wave, 16 rows at a time) -> float4 rows -> HBM
  float* stg = smem + wave * 16 * TK_LDS;
  const int mw = m0 + wave * 32;
  int q0 = 0, rem0 = 0;
  if (p.bias2) {  // image index of the wave's first row; later rows carry at most once when rpg >= 32
    q0 = mw / p.rpg;
    rem0 = mw - q0 * p.rpg;
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    __syncthreads();  // operand buffers (h = 0) / the previous half's staging rows are no longer read
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int r = 8 * h + rr;
        const int lr = (r & 3) + 8 * ((r >> 2) & 1) + 4 * lhi;
        stg[lr * TK_LDS + j * 32 + l31] = acc[j][r];
      }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int lr = it * 2 + lhi;
      const int c4 = l31 * 4;
      const int d = 16 * h + lr;
      const int m = mw + d, n = n0 + c4;
      if (m < p.M && n < p.N) {
        float4 v = *(const float4*)(stg + lr * TK_LDS + c4);
        if (p.bias) {
          const float4 b = *(const float4*)(p.bias + n);
          v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        }
        if (p.bias2) {
          const int img = (p.rpg >= 32) ? q0 + (rem0 + d >= p.rpg ? 1 : 0) : m / p.rpg;
          const float4 b = *(const float4*)(p.bias2 + (long)img * p.N + n);
          v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        }
        v.x *= p.alpha; v.y *= p.alpha; v.z *= p.alpha; v.w *= p.alpha;
        if (p.R) {
          const float4 rv = *(const float4*)(p.R + (long)m * p.ldr + n);
          v.x += p.beta * rv.x; v.y += p.beta * rv.y; v.z += p.beta * rv.z; v.w += p.beta * rv.w;
        }
        *(float4*)(p.C + (long)m * p.ldc + n) = v;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// thin-K, flat form (lda == K <= 32), persistent: a block keeps ONE 128-column B tile in LDS and walks down the rows.
// Every wave of this kernel has the same timing, so the waves of the chip run in lock-step and phases that wait for
// memory do NOT overlap across waves (measured on the first version: MFMA 31 us + stores 39 us + loads/LDS 20 us
// simply added up to 89 us).  The overlap therefore has to happen inside each wave:
//   * the next A tile (one contiguous run of 128*K floats) is requested right before the MFMA loop with inline-asm
//     global_load_dwordx4 (hipcc sinks a C++ load below the loop and then waits for it with vmcnt(0)) and is awaited
//     only after the loop, where it is written to LDS for the next iteration;
//   * the stores of a tile are issued last and nothing waits for them until the next tile's MFMAs are done
//     (vmcnt counts in order: waiting for the next loads is the first point that needs them retired);
//     weaving the previous tile's store passes INTO the MFMA loop (second accumulator set, 2 waves/SIMD) measured
//     the same 71 us and was dropped;
//   * the accumulators leave through per-wave staging rows (8 rows x 512 B per pass): no block barrier in the epilogue.
constexpr int TF_LDS = TK_TN + 8;
constexpr int TF_AS = TK_KC * TK_LDA, TF_BS = TK_KC * TK_LDB, TF_STG = 4 * 8 * TF_LDS;

__device__ float thin_zero_word[4];  // padded taps of the implicit im2col load from here (zero-initialised)

__device__ __forceinline__ void async_load1(float& dst, const float* src) {
  asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(src) : "memory");
}

// IM2COL: the A tile is gathered from a 3-channel NHWC image (K = 27) instead of being read as a flat run: 128
// consecutive pixels are whole image rows (128 % W == 0, H*W % 128 == 0, host-checked), so the (row, tap) -> source
// offset table of a thread's 16 elements is the same for every tile and only the row bound depends on the tile.
template <bool IM2COL, bool BF16 = false>
__global__ __launch_bounds__(256, (IM2COL ? 2 : 3)) void igemm_thin_kflat_kernel(ThinGemmArgs p) {
  __shared__ __attribute__((aligned(16))) float smem[TF_AS + TF_BS + TF_STG];
  float* as = smem;
  float* bs = smem + TF_AS;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  float* stg = smem + TF_AS + TF_BS + wave * 8 * TF_LDS;
  const int tiles_n = (p.N + TK_TN - 1) / TK_TN, tiles_m = (p.M + TK_TM - 1) / TK_TM;
  const int lb = xcd_remap(blockIdx.x, gridDim.x);  // the tiles_n blocks that share A tiles sit on one XCD
  const int tn = lb % tiles_n, mstep = gridDim.x / tiles_n;
  const int n0 = tn * TK_TN;
  const int K = p.K;
  const int tile_elems = TK_TM * K;

  // (row, k) and the clamped source offset of this thread's four float4 groups inside an A tile (same for every tile;
  // groups past the tile re-read its last float4 and are never stored)
  int grow[4], gk[4], goff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e0 = (tid + j * 256) * 4;
    grow[j] = e0 / K;
    gk[j] = e0 - grow[j] * K;
    goff[j] = min(e0, tile_elems - 4);
  }
  // The flat form prefetches the next tile with PLAIN loads (`ra`; hipcc places their wait, possibly after sinking them
  // below the MFMA loop -- this form is off the hot path).  Round 3 issued them as inline-asm global_load_dwordx4 awaited
  // by a later asm `s_waitcnt`: where the asynchronous and the synchronous (ragged / first tile) definitions of the
  // register set merged, hipcc copied the destination registers BEFORE the wait (v_mov_b64 of registers whose load was
  // still in flight: stale data whenever memory was slower than the MFMA loop in between).  The IM2COL form below keeps
  // asm loads into registers with a single definition; tools/asm_async_scan.py checks the ISA for such reads
  // (tests/test_isa_cpu.py).
  f32x4 ra[4], rs[4];
  // A tile -> LDS [k][row]; `total` = valid elements of the tile (ragged last tile: the rest stays stale, never stored)
  auto sstore_a = [&](const f32x4 (&ra)[4], int total) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e0 = (tid + j * 256) * 4;
      if (e0 < total) {
        int row = grow[j], k = gk[j];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (e0 + t < total) as[k * TK_LDA + row] = ra[j][t];
          if (++k == K) { k = 0; ++row; }
        }
      }
    }
  };
  // the last tile may be ragged (fewer than 128 rows): its loads are clamped element-wise in C++ (cold path)
  auto load_tile_sync = [&](int tm) {
    const int m0 = tm * TK_TM;
    const int total = min(TK_TM, p.M - m0) * K;
    const float* src = p.A + (long)m0 * K;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e0 = (tid + j * 256) * 4;
#pragma unroll
      for (int t = 0; t < 4; ++t) rs[j][t] = e0 + t < total ? src[e0 + t] : 0.f;
    }
    return total;
  };

  int tm = lb / tiles_n;
  // B tile: [K x 128] once (zero beyond K / N)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int g = tid + j * 256;
    const int k = g >> 5, c4 = (g & 31) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < K && n0 + c4 < p.N) v = *(const float4*)(p.B + (long)k * p.ldb + n0 + c4);
    *(float4*)(bs + k * TK_LDB + c4) = v;
  }
  if ((K & 1) && tid < TK_TM) as[K * TK_LDA + tid] = 0.f;  // odd K: second half of the last k-pair
  if constexpr (!IM2COL) {
    if (tm < tiles_m) sstore_a(rs, load_tile_sync(tm));
  }
  const int npairs = (K + 1) >> 1;
  const int c4 = l31 * 4, n = n0 + c4;  // this lane's four output columns: the same for every row it stores
  const bool nok = n < p.N;
  float4 bn = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias && nok) bn = *(const float4*)(p.bias + n);
  const float* af = as + wave * 32 + l31;
  const float* bf = bs + l31;

  // ---- implicit im2col: per-thread element table (tile-invariant) and the gather / store helpers
  int eoff[IM2COL ? 16 : 1], emeta[IM2COL ? 16 : 1];
  float rv[IM2COL ? 16 : 1];
  if constexpr (IM2COL) {
    const int C = p.cC, W = p.cW;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int e = tid + j * 256;
      eoff[j] = 0;
      emeta[j] = 0;
      if (e < tile_elems) {
        const int row = e / K, k = e - row * K;
        const int tap = k / C, c = k - tap * C;
        const int kh = tap / 3, kw = tap - 3 * kh;
        const int rh = row / W, rw = row - rh * W;
        const int hoff = rh - p.csign * (kh - 1), ww = rw - p.csign * (kw - 1);
        eoff[j] = (hoff * W + ww) * C + c;
        // bits 0-15: LDS slot, 16-23: row offset + 64, 24: column in range, 25: element exists
        emeta[j] = (k * TK_LDA + row) | ((hoff + 64) << 16) | ((ww >= 0 && ww < W) ? (1 << 24) : 0) | (1 << 25);
      }
    }
  }
  auto gather = [&](int tmg) {   // requests tile tmg's elements (asynchronous: see wait_gather)
    if constexpr (IM2COL) {
      const int m0 = tmg * TK_TM, hw = p.cH * p.cW;
      const int img = m0 / hw, h0 = (m0 - img * hw) / p.cW;
      const float* base = p.X + ((long)img * p.cH + h0) * p.cW * p.cC;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int hh = h0 + ((emeta[j] >> 16) & 0xff) - 64;
        const bool ok = ((emeta[j] >> 24) & 1) && hh >= 0 && hh < p.cH;
        async_load1(rv[j], ok ? base + eoff[j] : thin_zero_word);
      }
    }
  };
  // `tile_bias` != nullptr: bias[n] + bias2[image of the tile][n] already in registers (the im2col form prefetches it with the
  // next tile's gather: a load here would sit between the tile's stores and make hipcc drain them with vmcnt(0))
  auto epilogue_pass = [&](const f32x16 (&src)[4], int q, int tme, const float4* tile_bias) {
    const int mw = tme * TK_TM + wave * 32;
    const bool full = (tme + 1) * TK_TM <= p.M && n0 + TK_TN <= p.N;  // block-uniform
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) stg[(rr + 4 * lhi) * TF_LDS + j * 32 + l31] = src[j][4 * q + rr];
    __builtin_amdgcn_wave_barrier();
    float4 bq = bn;  // bias[n] + bias2[image of these 8 rows][n]: one load per pass (rpg % 8 == 0, host-checked)
    if (tile_bias) {
      bq = *tile_bias;
    } else if (p.bias2 && nok && mw + 8 * q < p.M) {  // (rows of a ragged last tile past M have no image: M % 8 == 0 here)
      const int img = (mw + 8 * q) / p.rpg;
      const float4 b = *(const float4*)(p.bias2 + (long)img * p.N + n);
      bq.x += b.x; bq.y += b.y; bq.z += b.z; bq.w += b.w;
    }
    float4 v[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      v[it] = *(const float4*)(stg + (it * 2 + lhi) * TF_LDS + c4);
      v[it].x = (v[it].x + bq.x) * p.alpha; v[it].y = (v[it].y + bq.y) * p.alpha;
      v[it].z = (v[it].z + bq.z) * p.alpha; v[it].w = (v[it].w + bq.w) * p.alpha;
    }
    if (p.R) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int m = mw + 8 * q + it * 2 + lhi;
        if (m < p.M && nok) {
          const float4 rv = *(const float4*)(p.R + (long)m * p.ldr + n);
          v[it].x += p.beta * rv.x; v[it].y += p.beta * rv.y; v[it].z += p.beta * rv.z; v[it].w += p.beta * rv.w;
        }
      }
    }
    float* crow = p.C + (long)(mw + 8 * q + lhi) * p.ldc + n;
    if (full && ((p.tune >> 8) & 1)) {
#pragma unroll
      for (int it = 0; it < 4; ++it) __builtin_nontemporal_store(*(const f32x4*)&v[it], (f32x4*)(crow + (long)(2 * it) * p.ldc));
    } else if (full) {
#pragma unroll
      for (int it = 0; it < 4; ++it) *(float4*)(crow + (long)(2 * it) * p.ldc) = v[it];
    } else {
#pragma unroll
      for (int it = 0; it < 4; ++it)
        if (mw + 8 * q + it * 2 + lhi < p.M && nok) *(float4*)(crow + (long)(2 * it) * p.ldc) = v[it];
    }
    __builtin_amdgcn_wave_barrier();
  };

  const int npairs_ = npairs;
  auto mfma_tile = [&](f32x16 (&acc)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // fragments of k-pair kk+1 are requested before the MFMAs of kk
    float a0 = rb16<BF16>(af[lhi * TK_LDA]), b0[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b0[j] = rb16<BF16>(bf[lhi * TK_LDB + j * 32]);
    for (int kk = 0; kk < npairs_; ++kk) {
      const int k = 2 * min(kk + 1, npairs_ - 1) + lhi;
      const float a1 = rb16<BF16>(af[k * TK_LDA]);
      float b1[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b1[j] = rb16<BF16>(bf[k * TK_LDB + j * 32]);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0[j], acc[j], 0, 0, 0);
      a0 = a1;
#pragma unroll
      for (int j = 0; j < 4; ++j) b0[j] = b1[j];
    }
  };

  if constexpr (IM2COL) {
    // Round 5: the stores of a tile are never waited for inside the iteration that issues them.  vmcnt retires in order, so
    // the OLD loop (gather(i+1) -> MFMAs(i) -> wait for the gather -> stores(i)) drained the stores of tile i-1 at every wait
    // for a gather and then left the write path idle while it scattered the tile and staged the next epilogue: 9 us per
    // round of 512 tiles against 5.3 us of HBM write time.  Here the prefetch group of tile i+2 (its 16 gathers + the
    // per-image bias row) is issued BEFORE the 16 stores of tile i, and the wait in front of scatter(i+2) is the counted
    // `vmcnt(16)`: it retires the prefetch group and everything older -- the stores of tile i-1 -- and leaves the 16 stores
    // of tile i in flight for one whole iteration more.  (A whole tile lies inside one image -- H W % 128 == 0, host-checked --
    // so its bias2 row is ONE 16-byte load per lane; all tiles are full: M % 128 == 0, N % 128 == 0 checked by the launcher
    // for the counted form, anything else waits with vmcnt(0).)
    // ONE static site per asynchronous load and per register hand-over (software-pipelined loop whose first two rounds only
    // prefetch): with a gather in a prologue AND in the loop the two definitions of `rv` merge at the loop header and hipcc
    // copies the destination registers there -- while the loads are in flight (tools/asm_async_scan.py, round 4's bug class).
    float4 bq_cur = bn, bq_next = bn;
    f32x4 bq_nxt;
    const bool exact = (p.M % TK_TM) == 0 && (p.N % TK_TN) == 0 && !p.R && !((p.tune >> 9) & 1);   // 16 unconditional stores per wave and tile
    bool counted = false;   // the first wait has no stores behind its prefetch group
    if ((p.tune & 255) && ((lb / tiles_n) & 1))
      for (int i = 0; i < (p.tune & 255); ++i) __builtin_amdgcn_s_sleep(16);
    // round r: compute + store tile t0 + r mstep (r >= 0), scatter tile t0 + (r + 1) mstep, request tile t0 + (r + 2) mstep
    for (int r = -2;; ++r) {
      const int tc = tm + r * mstep, tsc = tc + mstep, tg = tc + 2 * mstep;
      if (r >= 0 && tc >= tiles_m) break;
      f32x16 acc[4];
      if (r >= 0) {
        __syncthreads();  // this tile's A (written in the previous round) is visible
        mfma_tile(acc);
        __syncthreads();  // every wave is done with this tile's A fragments
      }
      if (r >= -1 && tsc < tiles_m) {
        if (counted && exact) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile(""
                     : "+v"(rv[0]), "+v"(rv[1]), "+v"(rv[2]), "+v"(rv[3]), "+v"(rv[4]), "+v"(rv[5]), "+v"(rv[6]), "+v"(rv[7]),
                       "+v"(rv[8]), "+v"(rv[9]), "+v"(rv[10]), "+v"(rv[11]), "+v"(rv[12]), "+v"(rv[13]), "+v"(rv[14]),
                       "+v"(rv[15]), "+v"(bq_nxt)::"memory");
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if ((emeta[j] >> 25) & 1) as[emeta[j] & 0xffff] = rv[j];
        bq_next = make_float4(bn.x + bq_nxt[0], bn.y + bq_nxt[1], bn.z + bq_nxt[2], bn.w + bq_nxt[3]);
      }
      if (tg < tiles_m) {
        gather(tg);
        const float* src = thin_zero_word;   // the tile's bias2 row: asynchronous like the gathers, awaited by the same s_waitcnt
        if (p.bias2 && nok) src = p.bias2 + (long)((tg * TK_TM) / p.rpg) * p.N + n;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bq_nxt) : "v"(src) : "memory");
      }
      if (r >= 0) {
        // stores last: they stay in flight until the wait of the round after the next
#pragma unroll
        for (int q = 0; q < 4; ++q) epilogue_pass(acc, q, tc, &bq_cur);
        counted = true;
      }
      bq_cur = bq_next;
    }
  } else {
    for (; tm < tiles_m; tm += mstep) {
      __syncthreads();  // this tile's A (written at the end of the previous iteration) is visible
      const int tnext = tm + mstep;
      const bool next_full = (tnext + 1) * TK_TM <= p.M;  // block-uniform
      if (next_full) {
        const float* src = p.A + (long)tnext * TK_TM * K;
#pragma unroll
        for (int j = 0; j < 4; ++j) ra[j] = *(const f32x4*)(src + goff[j]);   // plain loads: hipcc owns their wait (see `ra`)
      }
      f32x16 acc[4];
      mfma_tile(acc);
      __syncthreads();  // every wave is done with this tile's A fragments
      if (next_full) {
        sstore_a(ra, tile_elems);
      } else if (tnext < tiles_m) {
        sstore_a(rs, load_tile_sync(tnext));
      }
      // stores last: nothing waits for them until the next tile's MFMAs are done
#pragma unroll
      for (int q = 0; q < 4; ++q) epilogue_pass(acc, q, tm, nullptr);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Round 5: 3x3 convolution of a <= 3-channel NHWC image (K = 9 C <= 27, W = 32) without the LDS round trip of its im2col
// operand and without workgroup barriers.  igemm_thin_kflat_kernel<true, *> above staged the 128 x 27 im2col tile through
// LDS (16 gathers + 16 ds_write_b32 per thread, two workgroup barriers per tile) and its accumulators through LDS again (64
// ds_write_b32 + 16 ds_read_b128 per wave and tile): 72 us per launch at M = 131072 against 24 us of MFMA time and a 40 us HBM
// write floor, and tools/kflat_probe.py showed that nothing the memory system offers moves it (non-temporal stores, stores left
// in flight across the next tile with a counted vmcnt, de-phased workgroups: all within +-4 us; one workgroup per CU instead
// of two: 96 us) -- the launch is bound by the length of ONE wave's dependent chain per tile at two waves per SIMD.
// Here (the design of the bf16-storage thin convolution, flow_f16.hip):
//   * a wave owns 32 consecutive pixels (one image row) x 128 output columns; lane (l31, lhi) gathers exactly the 14 im2col
//     values it multiplies, k = 2 kk + lhi of pixel l31 (unconditional loads from clamped offsets + select: no branch, all
//     14 in flight), one tile ahead of the MFMAs;
//   * the MFMA operands are swapped, D[n][m]: the lane ends up with 16 x 4 CONSECUTIVE output columns of ITS pixel and stores
//     them as float4 straight from the accumulators (a 128-byte row segment is completed by four stores of the same wave, as
//     in the chain kernel's t2 store): no staging, no wave barrier;
//   * the weight slice [28 x 128] sits in LDS for the whole launch (one barrier per workgroup, before the loop); the waves of
//     a workgroup then run independently, four per SIMD (<= 128 VGPRs), so one wave's stores overlap another's MFMAs.
// Same arithmetic as the kernel it replaces (k pairs in the same order on the same fp32 MFMA, (acc + (bias + bias2)) * alpha):
// bit-identical results.
constexpr int TR_LDW = TK_TN + 4;   // weight slice rows: [28 VW][128 + 4]

// VW = 1: C = 3 (K = 27 = 27 units of one value); VW = 4: C = 12 (K = 108 = 27 units of four consecutive channels, fetched
// with one 16-byte load).  Either way a lane gathers 14 units per tile -- unit u = 2 kk + lhi, tap = u / 3 -- and multiplies
// them in the natural k order (k = VW u + i), so the weight rows are read as they lie in memory.  NWV waves per workgroup
// share one weight slice ([28 VW][128] floats: 15 KB / 59 KB) and nothing else.
// UNIT: alpha == 1 (x * 1.0f == x: the multiply is dropped; every call of the flow); JB: 32-column blocks per MFMA group
// ACT (round 5): the Sin activation of the flow's g-net as this kernel's epilogue -- C receives a = sin(2 pi v) / (2 pi),
// p.act_d receives d = cos(2 pi v), v = the convolution + bias (never written): the same `sincosf(TWO_PI * v)` /
// `s * INV_TWO_PI` expressions as indm_sincos_act_fwd_f32 (bit-identical), one 268 MB write and one 268 MB read less per
// g-pass at batch 128.  The conversion work (~60 VALU per element) runs next to the stores of a write-bound kernel.
template <int VW, int NWV, bool BF16, bool UNIT, int JB, bool COAL = false, bool ACT = false>
__global__ __launch_bounds__(NWV * 64, (VW == 1 ? 4 : 2)) void thin_conv_rows_kernel(ThinGemmArgs p) {   // (round 6: indm_sincos fits the ACT epilogue into 128 registers; libm sincosf spilled 14)
  __shared__ __attribute__((aligned(16))) float wsm[28 * VW * TR_LDW];
  __shared__ __attribute__((aligned(16))) float bsm[NWV * TK_TN];   // per wave: bias[n] + bias2[image][n] of its current image
  // COAL: a 32 x 32 block leaves through a per-wave LDS transpose (four ds_write_b128 + four ds_read_b128, no workgroup
  // barrier) so that every store instruction writes eight whole 128-byte row segments instead of 32-byte pieces of 32 rows
  __shared__ __attribute__((aligned(16))) float tsm[COAL ? NWV * 32 * 36 : 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int tiles_n = p.N / TK_TN;
  const int lb = xcd_remap(blockIdx.x, gridDim.x);   // the tiles_n workgroups of one row group sit on one XCD
  const int tn = lb % tiles_n, grp = lb / tiles_n, ngroups = gridDim.x / tiles_n;
  const int n0 = tn * TK_TN;
  const int K = p.K, C = p.cC, H = p.cH, W = p.cW;
  // weight slice -> LDS (rows >= K are zero: the pad of the last pair)
  for (int g = tid; g < 28 * VW * 32; g += NWV * 64) {
    const int k = g >> 5, c4 = (g & 31) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < K) v = *(const float4*)(p.B + (long)k * p.ldb + n0 + c4);
    *(float4*)(wsm + k * TR_LDW + c4) = v;
  }
  // Instruction count is what bounds this kernel once the barriers are gone: fp32 MFMAs do not co-execute with other waves'
  // VALU work (SQ_VALU_MFMA_COEXEC_CYCLES = 0, profiles/r05_pmc_thin_conv_rows.txt), so every VALU instruction of a tile is
  // added to its MFMA cycles.  The gathers therefore go through a raw BUFFER resource over the image: a lane whose tap
  // falls outside the image ORs bit 31 into its byte offset and the load returns 0.0 by itself (out of range) -- no
  // clamped address, no select behind the load: add + shift + and-or per unit.
  // per-lane tables of its 14 units: offset relative to its own pixel in units of VW floats (packed 10-bit fields:
  // |off| <= 3 (W + 1) + 2 = 101 for W = 32), and for every tap row kh the mask of the units whose tap is valid in that row
  // (W = 32: a tile is one image row; W = 16: two -- lane l31 is column l31 % W of row l31 / W of the tile)
  const int wl = l31 & (W - 1), dr = l31 / W, rpt = 32 / W;
  unsigned offp[5] = {0u, 0u, 0u, 0u, 0u};   // three fields per register (offset + 128)
  unsigned km[3] = {0u, 0u, 0u};             // bit kk of km[kh]: unit kk belongs to tap row kh, exists, column inside the image
#pragma unroll
  for (int kk = 0; kk < 14; ++kk) {
    const int u = 2 * kk + lhi;
    const int tap = u / 3, cu = u - tap * 3;
    const int kh = tap / 3, kw = tap - 3 * kh;
    const int dh = -p.csign * (kh - 1), dw = -p.csign * (kw - 1);
    const bool ok = u < 27 && wl + dw >= 0 && wl + dw < W;
    offp[kk / 3] |= (unsigned)((ok ? (dh * W + dw) * 3 + cu : 0) + 128) << (10 * (kk % 3));
    if (ok) km[kh < 3 ? kh : 0] |= 1u << kk;
  }
  float* brow = bsm + wave * TK_TN;
  const int rblocks = p.M >> 5;
  // a wave walks CONSECUTIVE image rows (its own run of `per` tiles): the image -- and with it the bias row -- changes at most
  // once or twice per wave (strided over the launch's waves, every tile was another image: a refresh with its drained loads per tile)
  const int per = (rblocks + ngroups * NWV - 1) / (ngroups * NWV);
  const int rb_beg = (grp * NWV + wave) * per, rb_end = min(rblocks, rb_beg + per);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, p.M * C * 4, 0x00020000);
  float* __restrict__ Cout = p.C;
  typedef float gvec __attribute__((ext_vector_type(VW)));
  gvec ga[14];
  auto gather = [&](int rbg, gvec (&g)[14]) {
    const int r0 = rbg * rpt;
    const int h = r0 - (r0 / H) * H + dr;   // this lane's image row
    // taps of row kh read image row h - csign (kh - 1): kh = 0 needs h + csign inside, kh = 2 needs h - csign inside
    const unsigned ok = (h + p.csign >= 0 && h + p.csign < H ? km[0] : 0u) | km[1] | (h - p.csign >= 0 && h - p.csign < H ? km[2] : 0u);
    const unsigned bad = ~ok;
    const unsigned pix = (unsigned)((rbg * 32 + l31) * 3 - 128) * (4u * VW);   // (the -128 undoes the bias of the packed offsets)
#pragma unroll
    for (int kk = 0; kk < 14; ++kk) {
      const unsigned o = ((offp[kk / 3] >> (10 * (kk % 3))) & 1023u) * (4u * VW);
      const unsigned oob = (bad << (31 - kk)) & 0x80000000u;
      if constexpr (VW == 1) g[kk] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, (pix + o) | oob, 0, 0));
      else g[kk] = __builtin_bit_cast(gvec, __builtin_amdgcn_raw_buffer_load_b128(xr, (pix + o) | oob, 0, 0));
    }
  };
  __syncthreads();
  int cur_img = -1;
  gvec gn[VW == 1 ? 14 : 1];   // VW = 1: the next tile's units, requested one tile ahead (56 more registers for VW = 4: not there)
  if constexpr (VW == 1) {
    if (rb_beg < rb_end) gather(rb_beg, gn);
  }
  for (int rb = rb_beg; rb < rb_end; ++rb) {
    if constexpr (VW == 1) {
#pragma unroll
      for (int kk = 0; kk < 14; ++kk) ga[kk] = gn[kk];
      // the NEXT tile's gathers go out first: they are the oldest requests in flight when this tile's stores are issued
      // (vmcnt retires in order), and this tile's MFMAs cover their latency
      if (rb + 1 < rb_end) gather(rb + 1, gn);
    } else {
      gather(rb, ga);   // the other waves of the SIMD cover the latency
    }
    if constexpr (BF16) {
#pragma unroll
      for (int kk = 0; kk < 14; ++kk)
#pragma unroll
        for (int i = 0; i < VW; ++i) ga[kk][i] = rb16<true>(ga[kk][i]);
    }
    const int img = (rb * 32) / p.rpg;
    if (img != cur_img) {   // wave-uniform: refresh this wave's bias row (two columns per lane; zeros without a bias:
      cur_img = img;        // the sum below is unconditional, like the `(v + bq) * alpha` of the kernel this one replaces)
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int c = lane + 64 * t;
        float v = p.bias ? *(const float*)((const char*)p.bias + (unsigned)((n0 + c) * 4)) : 0.f;
        if (p.bias2) v += *(const float*)((const char*)p.bias2 + (unsigned)((img * p.N + n0 + c) * 4));
        brow[c] = v;
      }
      __builtin_amdgcn_wave_barrier();
    }
    // one 64-bit row pointer per tile; the 16 stores differ by immediates (j * 128 + a * 32 bytes)
    float* crow = (float*)((char*)Cout + ((unsigned)(rb * 32 + l31) * (unsigned)p.ldc + (unsigned)(n0 + 4 * lhi)) * 4u);
    const float* bl = brow + 4 * lhi;
    // JB 32-column blocks at a time: their MFMAs (JB independent accumulator chains), then their 4 JB stores -- the stores
    // of a tile are spread over its arithmetic instead of following it (the first store of a launch leaves after a
    // fraction of a tile) and only 16 JB accumulator registers are live
#pragma unroll
    for (int j0 = 0; j0 < 4; j0 += JB) {
      __builtin_amdgcn_sched_barrier(0);   // one group at a time: hipcc otherwise hoists every group's LDS reads to the top and spills
      f32x16 acc[JB];
#pragma unroll
      for (int j = 0; j < JB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < 14; ++kk) {
        if constexpr (VW == 4) {
          if ((kk & 1) == 0) __builtin_amdgcn_sched_barrier(0);   // (at most eight weight reads ahead: 56 hoisted ones spill)
        }
#pragma unroll
        for (int i = 0; i < VW; ++i)
#pragma unroll
          for (int j = 0; j < JB; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(rb16<BF16>(wsm[(VW * (2 * kk + lhi) + i) * TR_LDW + l31 + (j0 + j) * 32]),
                                                          ga[kk][i], acc[j], 0, 0, 0);   // D[n][m]
      }
      __builtin_amdgcn_sched_barrier(0);
      // lane (l31, lhi) holds C[m = 32 rb + l31][n0 + 32 j + 8 a + 4 lhi + (0..3)] in acc[j][4a .. 4a+3]
#pragma unroll
      for (int j = 0; j < JB; ++j)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const float4 bq = *(const float4*)(bl + (j0 + j) * 32 + 8 * a);
          float4 v = make_float4(acc[j][4 * a] + bq.x, acc[j][4 * a + 1] + bq.y, acc[j][4 * a + 2] + bq.z, acc[j][4 * a + 3] + bq.w);
          if constexpr (!UNIT) { v.x *= p.alpha; v.y *= p.alpha; v.z *= p.alpha; v.w *= p.alpha; }
          if constexpr (ACT) {   // (a, d) = Sin(v); the cosines wait in the accumulator registers for their turn to leave
            float sn, cs;
            indm_sincos(TWO_PI * v.x, sn, cs); v.x = sn * INV_TWO_PI; acc[j][4 * a] = cs;
            indm_sincos(TWO_PI * v.y, sn, cs); v.y = sn * INV_TWO_PI; acc[j][4 * a + 1] = cs;
            indm_sincos(TWO_PI * v.z, sn, cs); v.z = sn * INV_TWO_PI; acc[j][4 * a + 2] = cs;
            indm_sincos(TWO_PI * v.w, sn, cs); v.w = sn * INV_TWO_PI; acc[j][4 * a + 3] = cs;
          }
          if constexpr (COAL) *(float4*)(tsm + (wave * 32 + l31) * 36 + 8 * a + 4 * lhi) = v;
          else *(float4*)(crow + (j0 + j) * 32 + 8 * a) = v;
        }
      if constexpr (COAL) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < JB; ++j)   // (JB = 1 only)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int row = (lane >> 3) + 8 * t;
            const float4 v = *(const float4*)(tsm + (wave * 32 + row) * 36 + 4 * (lane & 7));
            *(float4*)((char*)Cout + ((unsigned)(rb * 32 + row) * (unsigned)p.ldc + (unsigned)(n0 + (j0 + j) * 32 + 4 * (lane & 7))) * 4u) = v;
          }
        __builtin_amdgcn_wave_barrier();
      }
      if constexpr (ACT) {   // the cosine block, the same way
        float* __restrict__ Dout = p.act_d;
#pragma unroll
        for (int j = 0; j < JB; ++j)
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            const float4 v = make_float4(acc[j][4 * a], acc[j][4 * a + 1], acc[j][4 * a + 2], acc[j][4 * a + 3]);
            if constexpr (COAL) *(float4*)(tsm + (wave * 32 + l31) * 36 + 8 * a + 4 * lhi) = v;
            else *(float4*)((char*)Dout + ((char*)crow - (char*)Cout) + ((j0 + j) * 32 + 8 * a) * 4) = v;
          }
        if constexpr (COAL) {
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int j = 0; j < JB; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const int row = (lane >> 3) + 8 * t;
              const float4 v = *(const float4*)(tsm + (wave * 32 + row) * 36 + 4 * (lane & 7));
              *(float4*)((char*)Dout + ((unsigned)(rb * 32 + row) * (unsigned)p.ldc + (unsigned)(n0 + (j0 + j) * 32 + 4 * (lane & 7))) * 4u) = v;
            }
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
  }
}

template <int VW, int NWV>
int launch_thin_conv_rows(const ThinGemmArgs& a, int groups, int tiles_n, hipStream_t st) {
  const dim3 grid(groups * tiles_n), blk(NWV * 64);
  const bool unit = a.alpha == 1.0f;
  const int jb = ((a.tune >> 13) & 3);   // probe: 0 -> 1 block per group (default: measured best), 1 -> 2
  // stores through the per-wave LDS transpose (whole 128-byte row segments per store instruction) for the 3-channel form:
  // 59.4 -> 49.3 us at M = 131072 (5.45 TB/s); the 12-channel form is MFMA-bound and measured the same either way (42.6 / 43.2 us
  // at 16 x 16, 153 / 150 us at 32 x 32): it keeps the direct stores.  INDM_THIN_TUNE bit 15 flips the choice (tools/kflat_probe.py).
  constexpr bool CO = VW == 1;
  const bool flip = (a.tune >> 15) & 1;
  if (a.act_d) {   // (unit alpha: checked by the caller)
    if (a.bf16) hipLaunchKernelGGL((thin_conv_rows_kernel<VW, NWV, true, true, 1, CO, true>), grid, blk, 0, st, a);
    else hipLaunchKernelGGL((thin_conv_rows_kernel<VW, NWV, false, true, 1, CO, true>), grid, blk, 0, st, a);
    return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
  }
  if (a.bf16) {
    if (unit) hipLaunchKernelGGL((thin_conv_rows_kernel<VW, NWV, true, true, 1, CO>), grid, blk, 0, st, a);
    else hipLaunchKernelGGL((thin_conv_rows_kernel<VW, NWV, true, false, 1, CO>), grid, blk, 0, st, a);
  } else if (!unit) hipLaunchKernelGGL((thin_conv_rows_kernel<VW, NWV, false, false, 1, CO>), grid, blk, 0, st, a);
  else if (flip) hipLaunchKernelGGL((thin_conv_rows_kernel<VW, NWV, false, true, 1, !CO>), grid, blk, 0, st, a);
  else if (jb == 1) hipLaunchKernelGGL((thin_conv_rows_kernel<VW, NWV, false, true, 2, false>), grid, blk, 0, st, a);
  else hipLaunchKernelGGL((thin_conv_rows_kernel<VW, NWV, false, true, 1, CO>), grid, blk, 0, st, a);
  return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
}

int launch_thin_k(const ThinGemmArgs& a, hipStream_t st) {
  const bool flat = a.lda == a.K && a.K <= TK_KC;
  if (!flat && ((a.lda & 3) || (a.K & 3))) return THIN_NA;
  const int tiles = cdiv(a.M, TK_TM) * cdiv(a.N, TK_TN);
  if (flat && (!a.bias2 || !(a.rpg & 7))) {
    // persistent: three blocks per CU, each with its own B tile; a multiple of tiles_n (and of 8 for the XCD remap)
    const int tiles_n = cdiv(a.N, TK_TN), tiles_m = cdiv(a.M, TK_TM);
    const int target = 768;
    int groups = target / tiles_n;
    if (groups > tiles_m) groups = tiles_m;
    if (groups < 1) groups = 1;
    if (a.bf16) hipLaunchKernelGGL((igemm_thin_kflat_kernel<false, true>), dim3(groups * tiles_n), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((igemm_thin_kflat_kernel<false, false>), dim3(groups * tiles_n), dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
  }
  if (a.bf16) {
    if (flat) hipLaunchKernelGGL((igemm_thin_k_kernel<TK_FLAT, true>), dim3(tiles), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((igemm_thin_k_kernel<TK_STRIDED, true>), dim3(tiles), dim3(256), 0, st, a);
  } else if (flat) hipLaunchKernelGGL((igemm_thin_k_kernel<TK_FLAT, false>), dim3(tiles), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((igemm_thin_k_kernel<TK_STRIDED, false>), dim3(tiles), dim3(256), 0, st, a);
  return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
}


// ---------------------------------------------------------------------------------------------------------------
// thin weight grads:  D[t, c] = sum_p thin[p, t] * (wide[p, c] * gate[p, c]),   t < T <= 128,  c < N (512),  p < P.
// (3x3 convs with a skinny side: T = 27 / 108 taps x channels, P = pixels.)  One pass over the wide tensor (+ gate):
// HBM-bound for T = 27, MFMA-bound for T = 108.  Both operands are stored [pixel][column] = the [k][row] LDS layout,
// so slabs of 32 pixels are copied with float4 loads / ds_write_b128 (the thin one is a contiguous run of 32*T floats).
// grid = (N / 128 column tiles) x (S pixel chunks); block tile TP x 128 (TP = 32: one 32x32 MFMA tile per wave;
// TP = 128: 2x2 waves of 64x64); partial sums -> ws[s][t][c]; thin_wgrad_reduce adds them in a fixed order
// (deterministic) and writes D or its transpose (the form whose OUTPUT is [wide column][t]).
constexpr int TW_BK = 32, TW_TN = 128, TW_LDB = TW_TN + 4;

template <int TP, bool HAS_G, bool BF16 = false>
__global__ __launch_bounds__(256, (TP == 32 ? 3 : 2)) void thin_wgrad_kernel(const float* __restrict__ thin,
                                                                             const float* __restrict__ wide,
                                                                             const float* __restrict__ gate, float* __restrict__ ws,
                                                                             int T, int N, long ldw, int chunk, int P) {
  constexpr int LDA = TP + 4;
  constexpr int RM = TP == 32 ? 1 : 2, RN = TP == 32 ? 1 : 2;
  __shared__ __attribute__((aligned(16))) float smem[2 * TW_BK * (LDA + TW_LDB)];
  float* As = smem;
  float* Bs = smem + 2 * TW_BK * LDA;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int n0 = blockIdx.x * TW_TN;
  const int p_beg = blockIdx.y * chunk, p_end = min(P, p_beg + chunk);
  const int nslab = (p_end - p_beg) / TW_BK;
  const int mw = TP == 32 ? 0 : (wave >> 1) * 64, nw = TP == 32 ? wave * 32 : (wave & 1) * 64;

  // zero the pad columns [T, TP) of both A buffers once (slabs only write t < T)
  for (int i = tid; i < 2 * TW_BK * LDA; i += 256) As[i] = 0.f;
  __syncthreads();

  // thin slab = 32*T contiguous floats; this thread's float4 groups and their (pixel, t) inside the slab
  const int tgroups = (TW_BK * T) >> 2;
  int tpx[4], tt[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e0 = (tid + j * 256) * 4;
    tpx[j] = e0 / T;
    tt[j] = e0 - tpx[j] * T;
  }
  float4 rt[4], rw[4], rg[4];
  auto gload = [&](int sl) {
    const long p0 = p_beg + (long)sl * TW_BK;
    const float* tsrc = thin + p0 * T;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (tid + j * 256 < tgroups) rt[j] = *(const float4*)(tsrc + (tid + j * 256) * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int g = tid + j * 256;
      const int px = g >> 5, c4 = (g & 31) * 4;
      rw[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n0 + c4 < N) {
        rw[j] = *(const float4*)(wide + (p0 + px) * ldw + n0 + c4);
        if constexpr (HAS_G) rg[j] = *(const float4*)(gate + (p0 + px) * ldw + n0 + c4);
      } else if constexpr (HAS_G) {
        rg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto sstore = [&](int buf) {
    float* as = As + buf * TW_BK * LDA;
    float* bs = Bs + buf * TW_BK * TW_LDB;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (tid + j * 256 < tgroups) {
        int px = tpx[j], t = tt[j];
        const float v[4] = {rt[j].x, rt[j].y, rt[j].z, rt[j].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          as[px * LDA + t] = v[q];
          if (++t == T) { t = 0; ++px; }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int g = tid + j * 256;
      const int px = g >> 5, c4 = (g & 31) * 4;
      float4 v = rw[j];
      if constexpr (HAS_G) { v.x *= rg[j].x; v.y *= rg[j].y; v.z *= rg[j].z; v.w *= rg[j].w; }
      *(float4*)(bs + px * TW_LDB + c4) = v;
    }
  };

  f32x16 acc[RM][RN];
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nslab > 0) {
    gload(0);
    sstore(0);
  }
  __syncthreads();
  for (int sl = 0; sl < nslab; ++sl) {
    const int buf = sl & 1;
    if (sl + 1 < nslab) gload(sl + 1);
    const float* af = As + buf * TW_BK * LDA + mw + l31;
    const float* bf = Bs + buf * TW_BK * TW_LDB + nw + l31;
#pragma unroll
    for (int kk = 0; kk < TW_BK / 2; ++kk) {
      const int k = 2 * kk + lhi;
      float a[RM], b[RN];
#pragma unroll
      for (int i = 0; i < RM; ++i) a[i] = rb16<BF16>(af[k * LDA + i * 32]);
#pragma unroll
      for (int j = 0; j < RN; ++j) b[j] = rb16<BF16>(bf[k * TW_LDB + j * 32]);
#pragma unroll
      for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (sl + 1 < nslab) sstore(buf ^ 1);
    __syncthreads();
  }

  float* out = ws + (long)blockIdx.y * T * N;
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j) {
      const int c = n0 + nw + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int t = mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (t < T && c < N) out[(long)t * N + c] = acc[i][j][r];
      }
    }
}

// out = alpha * sum_s ws[s][t][c]  (+ beta * R, same layout as out), written as D[t][c] (ldo) or transposed [c][t] (ldo)
__global__ void thin_wgrad_reduce(const float* __restrict__ ws, float* __restrict__ out, int S, int T, int N, float alpha,
                                  int transpose, long ldo) {
  const int total = T * N;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int t = i / N, c = i - t * N;
  // Round 5: SIXTEEN slab values requested before the first is added (the loop used to fetch four, wait, add: S / 4 memory
  // latencies one after the other -- 15 us for S = 192 slabs of 55 KB, two thirds of what the weight-grad kernel itself takes).
  // The sums keep their association (s_j collects the slabs s = j mod 4 in order): bit-identical results.
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const float* w = ws + i;
  int s = 0;
  for (; s + 15 < S; s += 16) {
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = w[(long)(s + j) * total];
#pragma unroll
    for (int j = 0; j < 16; j += 4) {
      s0 += v[j];
      s1 += v[j + 1];
      s2 += v[j + 2];
      s3 += v[j + 3];
    }
  }
  for (; s + 3 < S; s += 4) {
    s0 += w[(long)s * total];
    s1 += w[(long)(s + 1) * total];
    s2 += w[(long)(s + 2) * total];
    s3 += w[(long)(s + 3) * total];
  }
  for (; s < S; ++s) s0 += w[(long)s * total];
  const float v = alpha * ((s0 + s1) + (s2 + s3));
  if (transpose) out[(long)c * ldo + t] = v;
  else out[(long)t * ldo + c] = v;
}

// ---- direct-to-LDS form of the thin weight grad (round 3) ---------------------------------------------------------
// thin_wgrad_kernel above stages its operands through registers: per 32-pixel slab a thread stores 16 thin elements with
// scalar LDS writes and carried (pixel, t) counters, multiplies the gate in, writes the wide rows -- about five non-MFMA
// VALU instructions per MFMA on a part where those do not overlap with the MFMAs: 4.0 - 4.3 TB/s on a
// launch that only has to stream 268 / 536 MB.  Here every operand goes global -> LDS by `global_load_lds_dwordx4
// voffset, s[base]` copies: the wide (and gate) rows as a [32 pixels][128 columns] image, the thin slab as the flat
// 128 T bytes it is in memory ([pixel][T], row stride T: fragment reads at t >= T pick up the next pixel's values, which
// only reach output rows t >= T that are never stored).  The gate is multiplied at the fragment read.
constexpr int TD_BK = 32;
__device__ __forceinline__ void td_dma(unsigned lds_byte_addr, unsigned voff, const void* sbase) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
               :: "s"(lds_byte_addr), "v"(voff), "s"(sbase) : "memory", "m0");
}
__device__ __forceinline__ unsigned td_lds(const float* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}

template <int TP, bool HAS_G, bool BF16 = false>
__global__ __launch_bounds__(256, 2) void thin_wgrad_dma_kernel(const float* __restrict__ thin, const float* __restrict__ wide,
                                                                 const float* __restrict__ gate, float* __restrict__ ws, int T,
                                                                 int N, long ldw, int chunk, int P) {
  constexpr int RM = TP == 32 ? 1 : 2, RN = TP == 32 ? 1 : 2;
  extern __shared__ __attribute__((aligned(16))) float td_smem[];
  // per buffer: wide [32][128], gate [32][128] (gated), thin [32 T] (+ TP floats of slack for the t >= T reads)
  const int thin_floats = ((TD_BK * T + TP + 3) & ~3);
  const int buf_floats = TD_BK * 128 * (HAS_G ? 2 : 1) + thin_floats;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int n0 = blockIdx.x * 128;
  const int p_beg = blockIdx.y * chunk, p_end = min(P, p_beg + chunk);
  const int nslab = (p_end - p_beg) / TD_BK;
  const int mw = TP == 32 ? 0 : (wave >> 1) * 64, nw = TP == 32 ? wave * 32 : (wave & 1) * 64;

  // copies of one slab: wide / gate rows 8 w .. 8 w + 7 of the slab (two 512-byte row pieces per instruction), and this
  // wave's share of the thin slab's 16-byte chunks
  const unsigned w_voff = (unsigned)((lane >> 5) * ldw * 4 + (lane & 31) * 16);
  const char* w_ub = (const char*)(wide + (long)(p_beg + 8 * wave) * ldw + n0);
  const char* g_ub = HAS_G ? (const char*)(gate + (long)(p_beg + 8 * wave) * ldw + n0) : nullptr;
  const long w_step = (long)TD_BK * ldw * 4, w_pair = 2 * ldw * 4;
  const int tchunks = (TD_BK * T) >> 2;                       // 16-byte chunks of a thin slab (128 T bytes)
  const int tinstr = (tchunks + 63) >> 6;                     // wave-instructions to copy it
  const char* t_ub = (const char*)(thin + (long)p_beg * T);
  const long t_step = (long)TD_BK * T * 4;
  auto stage = [&](int buf) {
    float* base = td_smem + buf * buf_floats;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      td_dma(__builtin_amdgcn_readfirstlane(td_lds(base + (8 * wave + 2 * i) * 128)), w_voff, w_ub + i * w_pair);
      if constexpr (HAS_G)
        td_dma(__builtin_amdgcn_readfirstlane(td_lds(base + TD_BK * 128 + (8 * wave + 2 * i) * 128)), w_voff, g_ub + i * w_pair);
    }
    float* tb = base + TD_BK * 128 * (HAS_G ? 2 : 1);
    for (int i = wave; i < tinstr; i += 4) {
      const int c = i * 64 + lane;
      const unsigned dst = __builtin_amdgcn_readfirstlane(td_lds(tb + i * 256));   // wave-uniform: keep it scalar for M0
      if (c < tchunks) td_dma(dst, (unsigned)(c * 16), t_ub);
    }
    w_ub += w_step;
    if constexpr (HAS_G) g_ub += w_step;
    t_ub += t_step;
  };

  f32x16 acc[RM][RN];
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nslab > 0) stage(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int sl = 0; sl < nslab; ++sl) {
    const int buf = sl & 1;
    if (sl + 1 < nslab) stage(buf ^ 1);
    const float* bw = td_smem + buf * buf_floats + nw + l31;
    const float* af = td_smem + buf * buf_floats + TD_BK * 128 * (HAS_G ? 2 : 1) + mw + l31;
#pragma unroll
    for (int kk = 0; kk < TD_BK / 2; ++kk) {
      const int k = 2 * kk + lhi;
      float a[RM], b[RN];
#pragma unroll
      for (int i = 0; i < RM; ++i) a[i] = rb16<BF16>(af[k * T + i * 32]);
#pragma unroll
      for (int j = 0; j < RN; ++j) {
        float v = bw[k * 128 + j * 32];
        if constexpr (HAS_G) v *= bw[TD_BK * 128 + k * 128 + j * 32];
        b[j] = rb16<BF16>(v);
      }
#pragma unroll
      for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  float* out = ws + (long)blockIdx.y * T * N;
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j) {
      const int c = n0 + nw + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int t = mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (t < T && c < N) out[(long)t * N + c] = acc[i][j][r];
      }
    }
}

// ---- register-streamed form of the thin weight grad, T <= 32 (round 5) ---------------------------------------------------
// Both forms above copy a 32-pixel slab to LDS, wait for it with vmcnt(0), synchronise the workgroup and only then multiply:
// one slab (16 / 32 KB) in flight per workgroup: 3.9 - 4.5 TB/s on launches that only have to read 268 / 536 MB once.  The
// contraction needs no LDS at all: with the MFMA's B operand taken as `wide[pixel 2 kk + lhi][128 w + 4 l31 + i]` -- the four
// floats of ONE 16-byte load per lane, a half-wave reading the 512 contiguous bytes of its pixel's row piece -- MFMA i (i = 0..3)
// accumulates output columns 128 w + 4 l31 + i: a permutation of the columns, undone for free by the float4 stores of the
// epilogue.  The A operand thin[pixel 2 kk + lhi][t = l31] is one dword load (lanes t >= T read on into the next pixel: they
// feed output rows that are never stored; a raw buffer resource returns 0 past the end of the tensor).  A workgroup owns a run
// of PIXELS and all N columns -- wave w the 128-column tile w, so the workgroup reads whole contiguous rows -- and its waves
// never meet: no barrier, no cross-wave sum, two stages of four pixel pairs in flight per wave (24 loads, counted vmcnt waits).
// Per pixel pair and 128 columns: two 16-byte loads (wide, gate), one dword load, two packed multiplies, four MFMAs.
// Measured at M = 131072 (tools/thin_bench.py, kernel + slab reduce; split = workgroups = partial slabs): ungated 61.4 us staged
// (192 slabs) -> 56.5 us at 256, 73 at 512, 96 at 1024; gated 106.3 -> 105.2 / 114 / 141: the launch is at the part's mixed-read
// rate either way (4.7 - 5.1 TB/s) and the reduce of the partial slabs (55 KB each) is what a finer split costs, so ONE
// workgroup per CU (256) is the default (INDM_THIN_WGRAD_STREAM = split, 0 = the staged forms above).
// TT = 1 (T <= 32): wave w of a workgroup owns the 128-column tile 4 blockIdx.x + w.  TT = 4 (T <= 128: the 12-channel scale,
// T = 108): the workgroup owns ONE 128-column tile and wave w the 32 thin rows [32 w, 32 w + 32) -- the four waves request the
// same wide / gate rows (one HBM fetch, four L1 / L2 reads) and 4 x the MFMAs run on them: MFMA-bound there.
template <bool HAS_G, bool BF16, int TT>
__global__ __launch_bounds__(256, 2) void thin_wgrad_stream_kernel(const float* __restrict__ thin, const float* __restrict__ wide,
                                                                    const float* __restrict__ gate, float* __restrict__ ws, int T,
                                                                    int N, long ldw, int chunk, int P) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int n0 = TT == 1 ? (blockIdx.x * 4 + wave) * 128 : blockIdx.x * 128;
  const int t0 = TT == 1 ? 0 : wave * 32;   // first thin row of this wave
  if (n0 >= N || t0 >= T) return;
  const int w_beg = blockIdx.y * chunk, w_end = min(P, w_beg + chunk);   // a multiple of 16 pixels (host-checked)
  const __amdgpu_buffer_rsrc_t tr = __builtin_amdgcn_make_buffer_rsrc((void*)thin, 0, (int)((long)P * T * 4), 0x00020000);
  const unsigned w_off = (unsigned)(lhi * ldw + 4 * l31) * 4u;        // bytes from the pair's row base
  const unsigned t_off = (unsigned)(lhi * T + t0 + l31) * 4u;
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  constexpr int U = 4;   // pixel pairs per stage
  f32x4 wv[2][U], gv[2][U];
  float tv[2][U];
  auto request = [&](int px, int st) {   // the pixel pairs (px + 2u, px + 2u + 1) -> stage st
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long row = (long)(px + 2 * u) * ldw + n0;
      wv[st][u] = *(const f32x4*)((const char*)(wide + row) + w_off);
      if constexpr (HAS_G) gv[st][u] = *(const f32x4*)((const char*)(gate + row) + w_off);
      tv[st][u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(tr, (unsigned)((px + 2 * u) * T) * 4u + t_off, 0, 0));
    }
  };
  auto consume = [&](int st) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      f32x4 b = wv[st][u];
      if constexpr (HAS_G) b = b * gv[st][u];
      const float a = rb16<BF16>(tv[st][u]);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, rb16<BF16>(b[i]), acc[i], 0, 0, 0);
    }
  };
  int px = w_beg;
  if (px < w_end) request(px, 0);
  for (; px + 4 * U <= w_end; px += 4 * U) {   // two stages per trip: no register rotation
    request(px + 2 * U, 1);
    __builtin_amdgcn_sched_barrier(0);   // (hipcc otherwise sinks the requests below the MFMAs they are meant to run under)
    consume(0);
    __builtin_amdgcn_sched_barrier(0);
    request(min(px + 4 * U, w_end - 2 * U), 0);   // unconditional (the last trip re-reads its own pixels): a branch here makes hipcc
    __builtin_amdgcn_sched_barrier(0);            // wait for the merged state of both paths, i.e. drain the requests it has just issued
    consume(1);
    __builtin_amdgcn_sched_barrier(0);
  }
  // lane (l31, lhi), register r of accumulator i: out[t = t0 + (r & 3) + 8 (r >> 2) + 4 lhi][n0 + 4 l31 + i]
  float* out = ws + (long)blockIdx.y * T * N + n0 + 4 * l31;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int t = t0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
    if (t < T) *(float4*)(out + (long)t * N) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
  }
}

template <int TP, bool HAS_G, bool BF16>
int launch_thin_wgrad_dma(const float* thin, const float* wide, const float* gate, float* ws, int T, int N, long ldw, int chunk,
                          int P, dim3 grid, hipStream_t st) {
  const int thin_floats = ((TD_BK * T + TP + 3) & ~3);
  const size_t bytes = 2 * (size_t)(TD_BK * 128 * (HAS_G ? 2 : 1) + thin_floats) * sizeof(float);
  // > 64 KB of LDS needs the opt-in (per instantiation and device; T changes the size)
  if (indm_lds_optin<thin_wgrad_dma_kernel<TP, HAS_G, BF16>>(bytes) != INDM_OK) return INDM_E_LAUNCH;
  hipLaunchKernelGGL((thin_wgrad_dma_kernel<TP, HAS_G, BF16>), grid, dim3(256), bytes, st, thin, wide, gate, ws, T, N, ldw, chunk, P);
  return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
}

// thin: [P, T] contiguous; wide (+gate): [P, N] with row stride ldw; out: [T, N] (ldo) or, transposed, [N, T] (ldo)
int launch_thin_wgrad(const float* thin, const float* wide, const float* gate, float* out, int T, int N, long P, long ldw,
                      float alpha, int transpose, long ldo, float* ws, long ws_floats, hipStream_t st, int bf16) {
  if (T < 1 || T > 128 || (N & 3) || (ldw & 3) || (P % TW_BK) || P < 4096 || P > (1L << 30)) return THIN_NA;
  if (!al16(thin) || !al16(wide) || !al16(gate) || !ws) return THIN_NA;
  const int tiles_n = cdiv(N, TW_TN);
  long S = (T <= 32 ? 768 : 512) / tiles_n;  // ~3 (TP = 32) / 2 (TP = 128) blocks per CU
  if (S < 1) S = 1;
  int chunk = cdiv(cdiv(P, S), TW_BK) * TW_BK;
  S = cdiv(P, chunk);
  if (S * T * N > ws_floats || S > 65535) return THIN_NA;
  dim3 grid(tiles_n, (unsigned)S);
  static const int stream_s = [] { const char* e = getenv("INDM_THIN_WGRAD_STREAM"); return e ? atoi(e) : 256; }();   // 0: staged forms
  static const int stream_wide = [] { const char* e = getenv("INDM_THIN_WGRAD_STREAM_T128"); return e ? atoi(e) : 1; }();
  // (T > 32 only for the GATED launches -- measured at M = 32768, T = 108, kernel + reduce: gated 61.3 -> 52.7 us, ungated
  // 50.8 -> 56.8 us: the direct-to-LDS form keeps those)
  if ((T <= 32 || (T <= 128 && stream_wide && gate)) && !(N & 127) && !(ldw & 3) && (long)P * T * 4 < (1L << 31) && stream_s > 0) {
    // register-streamed form.  T <= 32: one workgroup per pixel run (all N columns, 128 per wave), one workgroup per CU.
    // T <= 128: a workgroup per (pixel run, 128-column tile), wave = 32 thin rows; the same number of workgroups per CU.
    const bool wide_t = T > 32;
    long Ss = wide_t ? cdiv(stream_s, tiles_n) * 2 : stream_s;
    int ch = cdiv(cdiv(P, Ss), TW_BK) * TW_BK;   // a multiple of 32 pixels (P % 32 == 0 checked above): whole 16-pixel trips
    Ss = cdiv(P, ch);
    if (Ss * T * N <= ws_floats && Ss <= 65535) {
      const dim3 g2(wide_t ? tiles_n : cdiv(tiles_n, 4), (unsigned)Ss);
#define INDM_TWS(G, B, TTV) hipLaunchKernelGGL((thin_wgrad_stream_kernel<G, B, TTV>), g2, dim3(256), 0, st, thin, wide, gate, ws, T, N, ldw, ch, (int)P)
      if (wide_t) {
        if (gate) { if (bf16) INDM_TWS(true, true, 4); else INDM_TWS(true, false, 4); }
        else { if (bf16) INDM_TWS(false, true, 4); else INDM_TWS(false, false, 4); }
      } else {
        if (gate) { if (bf16) INDM_TWS(true, true, 1); else INDM_TWS(true, false, 1); }
        else { if (bf16) INDM_TWS(false, true, 1); else INDM_TWS(false, false, 1); }
      }
#undef INDM_TWS
      if (hipGetLastError() != hipSuccess) return INDM_E_LAUNCH;
      hipLaunchKernelGGL(thin_wgrad_reduce, dim3(cdiv((long)T * N, 256)), dim3(256), 0, st, ws, out, (int)Ss, T, N, alpha, transpose, ldo);
      return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
    }
  }
  // direct-to-LDS form for the UNGATED launches: full 128-column tiles, 16-byte aligned rows (every production shape).
  // Measured at M = 131072 / 32768 (kernel + reduce): 77 -> 69 us and 65 -> 57 us ungated; the gated form reads three LDS
  // words per MFMA at two workgroups per CU and ran 119 / 73 us against 114 / 68 us register-staged, so it stays there.
  if (!gate && !(N & 127) && !(ldw & 3) && (long)P * ldw * 4 < (1L << 32) && (long)P * T * 4 < (1L << 32)) {
    int r;
    if (T <= 32) r = gate ? (bf16 ? launch_thin_wgrad_dma<32, true, true>(thin, wide, gate, ws, T, N, ldw, chunk, (int)P, grid, st)
                                  : launch_thin_wgrad_dma<32, true, false>(thin, wide, gate, ws, T, N, ldw, chunk, (int)P, grid, st))
                         : (bf16 ? launch_thin_wgrad_dma<32, false, true>(thin, wide, gate, ws, T, N, ldw, chunk, (int)P, grid, st)
                                  : launch_thin_wgrad_dma<32, false, false>(thin, wide, gate, ws, T, N, ldw, chunk, (int)P, grid, st));
    else r = gate ? (bf16 ? launch_thin_wgrad_dma<128, true, true>(thin, wide, gate, ws, T, N, ldw, chunk, (int)P, grid, st)
                          : launch_thin_wgrad_dma<128, true, false>(thin, wide, gate, ws, T, N, ldw, chunk, (int)P, grid, st))
                  : (bf16 ? launch_thin_wgrad_dma<128, false, true>(thin, wide, gate, ws, T, N, ldw, chunk, (int)P, grid, st)
                          : launch_thin_wgrad_dma<128, false, false>(thin, wide, gate, ws, T, N, ldw, chunk, (int)P, grid, st));
    if (r != INDM_OK) return r;
    hipLaunchKernelGGL(thin_wgrad_reduce, dim3(cdiv((long)T * N, 256)), dim3(256), 0, st, ws, out, (int)S, T, N, alpha, transpose, ldo);
    return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
  }
  if (T <= 32) {
    if (gate) { if (bf16) hipLaunchKernelGGL((thin_wgrad_kernel<32, true, true>), grid, dim3(256), 0, st, thin, wide, gate, ws, T, N, ldw, chunk, (int)P);
      else hipLaunchKernelGGL((thin_wgrad_kernel<32, true, false>), grid, dim3(256), 0, st, thin, wide, gate, ws, T, N, ldw, chunk, (int)P); }
    else { if (bf16) hipLaunchKernelGGL((thin_wgrad_kernel<32, false, true>), grid, dim3(256), 0, st, thin, wide, gate, ws, T, N, ldw, chunk, (int)P);
      else hipLaunchKernelGGL((thin_wgrad_kernel<32, false, false>), grid, dim3(256), 0, st, thin, wide, gate, ws, T, N, ldw, chunk, (int)P); }
  } else {
    if (gate) { if (bf16) hipLaunchKernelGGL((thin_wgrad_kernel<128, true, true>), grid, dim3(256), 0, st, thin, wide, gate, ws, T, N, ldw, chunk, (int)P);
      else hipLaunchKernelGGL((thin_wgrad_kernel<128, true, false>), grid, dim3(256), 0, st, thin, wide, gate, ws, T, N, ldw, chunk, (int)P); }
    else { if (bf16) hipLaunchKernelGGL((thin_wgrad_kernel<128, false, true>), grid, dim3(256), 0, st, thin, wide, gate, ws, T, N, ldw, chunk, (int)P);
      else hipLaunchKernelGGL((thin_wgrad_kernel<128, false, false>), grid, dim3(256), 0, st, thin, wide, gate, ws, T, N, ldw, chunk, (int)P); }
  }
  if (hipGetLastError() != hipSuccess) return INDM_E_LAUNCH;
  hipLaunchKernelGGL(thin_wgrad_reduce, dim3(cdiv((long)T * N, 256)), dim3(256), 0, st, ws, out, (int)S, T, N, alpha, transpose, ldo);
  return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
}

}  // namespace

int indm_thin_conv3x3(const ThinGemmArgs& a_in, hipStream_t st) {
  ThinGemmArgs a = a_in;
  {
    static const int tune = [] {   // experiment knobs, read once per process (tools/kflat_probe.py starts one process per value)
      const char* e = getenv("INDM_THIN_TUNE");
      return e ? (int)strtol(e, nullptr, 0) : 0;
    }();
    a.tune = tune;
  }
  const int C = a.cC, H = a.cH, W = a.cW;
  if (!a.X || C < 1 || a.K != 9 * C || a.batch != 1 || a.G || a.M < 1024 || a.N < 128 || (a.N & 3) || (a.ldb & 3) ||
      (a.ldc & 3) || (a.R && (a.ldr & 3)) || !al16(a.X) || !al16(a.B) || !al16(a.C) || !al16(a.bias) || !al16(a.bias2) ||
      !al16(a.R) || (long)a.M * 9 * C >= (1L << 31))
    return THIN_NA;
  const int tiles_n = cdiv(a.N, TK_TN), tiles_m = cdiv(a.M, TK_TM);
  // bit 12 of INDM_THIN_TUNE: the round-4 staged kernels (A/B, tools/kflat_probe.py)
  if ((C == 3 || C == 12) && (W == 32 || W == 16) && !(H % (32 / W)) && !(a.N % TK_TN) && !a.R && !(a.M % 32) && (!a.bias2 || !(a.rpg % 32)) && !((a.tune >> 12) & 1) &&
      (long)a.M * a.ldc * 4 < (1L << 32) && (!a.bias2 || ((long)a.M / a.rpg + 1) * a.N * 4 < (1L << 32)) && (long)a.M * C * 4 < (1L << 31)) {
    int per_cu = (a.tune >> 10) & 3;
    const int nwv = 4;
    // C = 3: 88 VGPRs, 17 KB of LDS -> four workgroups per CU; C = 12: the 14 float4 units of a lane are 56 registers (~160 in
    // all; forced under 128 hipcc spills 13 - 90 of them depending on the loop shape) and the weight slice is 59 KB -> two
    if (!per_cu) per_cu = C == 3 ? (a.act_d ? 3 : 4) : 2;   // (the Sin epilogue: 144 VGPRs)
    int groups = 256 * per_cu / tiles_n;    // each workgroup walks its share of the image rows
    const int rgroups = cdiv(a.M / 32, nwv);
    if (groups > rgroups) groups = rgroups;
    if (groups < 1) groups = 1;
    if (a.act_d && (a.alpha != 1.0f || !al16(a.act_d))) return THIN_NA;
    return C == 3 ? launch_thin_conv_rows<1, 4>(a, groups, tiles_n, st) : launch_thin_conv_rows<4, 4>(a, groups, tiles_n, st);
  }
  if (a.act_d) return THIN_NA;   // only the row-fragment kernel has the activation epilogue
  if (C <= 3 && W <= 128 && (128 % W) == 0 && ((H * W) % 128) == 0 && (a.M % 128) == 0 && (!a.bias2 || !(a.rpg & 7))) {
    int groups = 512 / tiles_n;  // two blocks per CU (the element tables cost registers)
    if ((a.tune >> 10) & 3) groups = 256 * ((a.tune >> 10) & 3) / tiles_n;
    if (groups > tiles_m) groups = tiles_m;
    if (groups < 1) groups = 1;
    if (a.bf16) hipLaunchKernelGGL((igemm_thin_kflat_kernel<true, true>), dim3(groups * tiles_n), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((igemm_thin_kflat_kernel<true, false>), dim3(groups * tiles_n), dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
  }
  if (!(C & 3) && 9 * C <= 128) {
    if (a.bf16) hipLaunchKernelGGL((igemm_thin_k_kernel<TK_IM2COL, true>), dim3(tiles_m * tiles_n), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((igemm_thin_k_kernel<TK_IM2COL, false>), dim3(tiles_m * tiles_n), dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
  }
  return THIN_NA;
}

int indm_thin_gemm(const ThinGemmArgs& a, hipStream_t st, const char** variant) {
  if (a.batch != 1) return THIN_NA;
  if (!a.transA && !a.transB && !a.G && a.K >= 1 && a.K <= 128 && a.N >= 128 && !(a.N & 3) && a.M >= 1024 &&
      !(a.ldb & 3) && !(a.ldc & 3) && al16(a.A) && al16(a.B) && al16(a.C) && al16(a.bias) && al16(a.bias2) &&
      al16(a.R) && !(a.ldr & 3)) {
    *variant = "thin_k128x128";
    return launch_thin_k(a, st);
  }
  // weight-grad forms: reduction over pixels (K), A stored [K, M]
  if (a.transA && !a.transB && !a.bias && !a.bias2 && !a.R && a.K >= 4096) {
    // thin^T x wide -> [T, N]  (A = thin, contiguous [K, M]; B = wide)
    *variant = (a.M <= 128 ? a.M : a.N) <= 32 ? "thin_wgrad32x128" : "thin_wgrad128x128";
    if (a.M <= 128 && a.N >= 256 && !a.G && a.lda == a.M)
      return launch_thin_wgrad(a.A, a.B, nullptr, a.C, a.M, a.N, a.K, a.ldb, a.alpha, 0, a.ldc, a.ws, a.ws_floats, st, a.bf16);
    // (wide .* gate)^T x thin -> [N_wide, T]  (A = wide (+ gate), B = thin contiguous [K, N])
    if (a.N <= 128 && a.M >= 256 && a.ldb == a.N)
      return launch_thin_wgrad(a.B, a.A, a.G, a.C, a.N, a.M, a.K, a.lda, a.alpha, 1, a.ldc, a.ws, a.ws_floats, st, a.bf16);
  }
  return THIN_NA;
}

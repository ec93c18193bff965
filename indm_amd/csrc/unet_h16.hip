// bf16 STORAGE of the score network (NCSN++) -- BASELINE configs #3-#5 ("bf16 compute / fp32 accumulate"), round 6.
//
// Round 2-5's bf16 mode rounded the MFMA inputs of every contraction but kept the U-Net's activations fp32 in HBM: its
// convolutions streamed fp32 slabs through LDS and converted per fragment (0.16 of the bf16 MFMA peak, latency-bound on
// the copies).  Here the activations, their gradients and a per-call packed copy of the weights are bf16 in HBM
// ("H16": NHWC rows of bf16, two values per dword); accumulators, biases, the time embedding, statistics, weight
// gradients and the fp32 master weights stay fp32.
//   * h16_conv_kernel   implicit-GEMM convolution / plain NT GEMM: A = bf16 NHWC activations gathered by tap
//                       (buffer-addressed direct-to-LDS copies, a padded tap copies zeros), B = bf16 weights packed
//                       k-contiguous [N][K].  A 32-deep bf16 slab is the same 64 bytes per row as the fp32 kernel's
//                       16-deep slab, so the LDS image ([row][4 x 16 B], XOR-swizzled on the SOURCE address) and the
//                       three-buffer counted-vmcnt loop of igemm_glds_kernel carry over -- but a fragment is ONE
//                       ds_read_b128 with no conversion and a slab holds eight v_mfma_f32_32x32x16_bf16 per wave.
//                       Forward, data-grad (= the same kernel on the tap-flipped packed weight), 1x1 / NIN.
//   * h16_wgrad_kernel  weight gradient dW[(tap,ci)][co] = sum_pixels x[pixel + tap][ci] dy[pixel][co]: both operands
//                       are stored channel-contiguous while the MFMA wants 8 consecutive PIXELS per lane -- the LDS
//                       images stay in memory order ([pixel][channel], filled by the same DMA) and the fragments come
//                       out of ds_read_b64_tr_b16 (gfx950's transposing LDS read).  fp32 split-K partials + reduce.
//   * pack / convert / reduce helpers.
// Reference workloads: models/layerspp.py:225-287 (ResnetBlockBigGANpp), models/ncsnpp.py:251-414, models/layers.py:88-124.
#include "indm_common.h"
#pragma clang diagnostic ignored "-Winline-asm"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;

struct H16Args {
  const void* A;       // bf16 activations, NHWC [Nimg, H, W, Cin]
  const void* B;       // conv: bf16 packed weight [N][K] (k = (tap, ci));  wgrad: bf16 dy [K = pixels][N]
  void* C;             // conv: [M][N] bf16 (fp32 when out_f32);  wgrad: fp32 [M = taps * Cin][N]
  int M, N, K;
  int Nimg, H, W, Cin, OH, OW, KH, KW, stride, pad;
  int ow_sh, ohw_sh;   // log2(OW), log2(OH * OW) (both powers of two)
  float alpha, beta;
  const float* bias;   // [N]
  const float* bias2;  // [M / rpg][N]
  int rpg;
  const void* R;       // bf16 residual [M][N]
  int out_f32;
  int splitk, kps;     // kps: reduction length per split (multiple of 32)
  float* ws;           // fp32 partials [splitk][M][N]
  int* cnt;            // per-tile arrival counters (zero at rest) for the in-kernel split-K combine, or NULL: separate reduce launch
};

__device__ __forceinline__ int xcd_remap(int b, int n) {
  const int q = n >> 3, r = n & 7;
  const int xcd = b & 7, idx = b >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
// direct-to-LDS copy through a raw buffer resource (see igemm.hip: dma_buffer): 64 lanes x 16 bytes land lane-linear at
// M0; a lane whose offset is out of range (bit 31 set / beyond num_records) copies zeros
__device__ __forceinline__ void dma_buffer(unsigned lds_byte_addr, unsigned voff, i32x4 rsrc, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_byte_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
template <int C>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C) : "memory"); }
__device__ __forceinline__ void lds_barrier() {   // orders LDS for the compiler, does not wait for the asm-issued copies
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ i32x4 make_rsrc(const void* base, long bytes) {
  const unsigned long long ab = (unsigned long long)base;
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(ab & 0xffffffffu));
  r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((ab >> 32) & 0xffffu));   // stride 0: raw buffer
  r[2] = __builtin_amdgcn_readfirstlane((int)(unsigned)(bytes > 0x7fffffffL ? 0x7fffffffL : bytes));
  r[3] = 0x00020000;                                                               // DATA_FORMAT_32
  return r;
}
__device__ __forceinline__ float bf16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
// value of the neighbouring lane (lane ^ 1): one DPP quad_perm [1,0,3,2] move, no LDS crossbar
__device__ __forceinline__ float swap_pair(float v) {
  const int i = __float_as_int(v);
  return __int_as_float(__builtin_amdgcn_update_dpp(i, i, 0xB1, 0xF, 0xF, false));
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {   // RNE, one v_cvt_pk_bf16_f32
  const f32x2 v = {lo, hi};
  const bf16x2 h = __builtin_convertvector(v, bf16x2);
  return *(const unsigned*)&h;
}

// ---------------------------------------------------------------------------------------------------------------------
// Epilogue of one 32 x 32 accumulator block (lane: column n = n0 + l31, rows mbase + (r & 3) + 8 (r >> 2) + 4 lhi):
//   v = alpha * (acc + bias[n] + bias2[m / rpg][n]) + beta * R[m][n]   -> bf16 C[m][n]
// bf16 rows are written as dwords: lanes n / n + 1 trade one value per row pair (DPP), the even lane stores
// (C[m][n], C[m][n+1]) of the even row, the odd lane (C[m+1][n-1], C[m+1][n]) of the odd row -- 8 stores of 64 bytes per
// half-wave instead of 16 two-byte stores; the residual comes in the same way.
template <bool F32OUT>
__device__ __forceinline__ void h16_epilogue_block(const H16Args& p, const f32x16& acc, int mbase, int n, int lhi, bool odd) {
  __builtin_amdgcn_sched_barrier(0);
  const bool nok = n < p.N;
  float v[16];
  const float bn = (p.bias && nok) ? p.bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = acc[r] + bn;
  if (p.bias2) {
    float t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = mbase + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      t[r] = p.bias2[(nok && m < p.M) ? (unsigned)(m / p.rpg) * (unsigned)p.N + (unsigned)n : 0u];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] += t[r];
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] *= p.alpha;
  if constexpr (F32OUT) {
    float* C = (float*)p.C;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = mbase + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      if (nok && m < p.M) C[(long)m * p.N + n] = v[r];
    }
    return;
  }
  const unsigned* R = (const unsigned*)p.R;
  unsigned* C = (unsigned*)p.C;
  // this lane's dword of row pair j (rows r = 2 j, 2 j + 1): even lane -> row 2 j at column n, odd lane -> row 2 j + 1 at n - 1
  unsigned off[8];
  bool ok[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int r = 2 * j + (odd ? 1 : 0);
    const int m = mbase + (r & 3) + 8 * (r >> 2) + 4 * lhi;
    ok[j] = nok && m < p.M;
    off[j] = ok[j] ? ((unsigned)m * (unsigned)p.N + (unsigned)(n - (odd ? 1 : 0))) >> 1 : 0u;
  }
  if (R) {
    unsigned w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = R[off[j]];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // even lane holds (R[m][n], R[m][n+1]) of the even row, odd lane (R[m+1][n-1], R[m+1][n]) of the odd row
      const float mine = odd ? bf16_hi(w[j]) : bf16_lo(w[j]);          // the row this lane loaded, its own column
      const float give = odd ? bf16_lo(w[j]) : bf16_hi(w[j]);          // the partner's column of that row
      const float got = swap_pair(give);                                // the OTHER row, this lane's column
      v[2 * j] += p.beta * (odd ? got : mine);
      v[2 * j + 1] += p.beta * (odd ? mine : got);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float give = odd ? v[2 * j] : v[2 * j + 1];
    const float got = swap_pair(give);
    const unsigned w = odd ? pack_bf16x2(got, v[2 * j + 1]) : pack_bf16x2(v[2 * j], got);
    if (ok[j]) C[off[j]] = w;
  }
}

// One column pair (m, n .. n + 1) of the split-K combine + epilogue: the partial slabs in slab order (fixed: deterministic),
// eight loads in flight.  Shared by h16_reduce_kernel and the in-kernel combine of the last-arriving workgroup of a tile.
__device__ __forceinline__ void h16_reduce_pair(const H16Args& p, int m, int n) {
  const long mn = (long)p.M * p.N;
  const long e = (long)m * p.N + n;
  float s0 = 0.f, s1 = 0.f;
  int kz = 0;
  for (; kz + 7 < p.splitk; kz += 8) {
    f32x2 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = *(const f32x2*)(p.ws + (long)(kz + j) * mn + e);
#pragma unroll
    for (int j = 0; j < 8; ++j) { s0 += v[j][0]; s1 += v[j][1]; }
  }
  for (; kz < p.splitk; ++kz) {
    const f32x2 v = *(const f32x2*)(p.ws + (long)kz * mn + e);
    s0 += v[0]; s1 += v[1];
  }
  if (p.bias) { s0 += p.bias[n]; s1 += p.bias[n + 1]; }
  if (p.bias2) {
    const long o = (long)(m / p.rpg) * p.N + n;
    s0 += p.bias2[o]; s1 += p.bias2[o + 1];
  }
  s0 *= p.alpha; s1 *= p.alpha;
  if (p.out_f32) {
    if (p.R) {   // (fp32 outputs carry an fp32 residual: the weight-grad accumulate form)
      s0 += p.beta * ((const float*)p.R)[e];
      s1 += p.beta * ((const float*)p.R)[e + 1];
    }
    *(f32x2*)((float*)p.C + e) = f32x2{s0, s1};
  } else {
    if (p.R) {
      const unsigned w = ((const unsigned*)p.R)[e >> 1];
      s0 += p.beta * bf16_lo(w);
      s1 += p.beta * bf16_hi(w);
    }
    ((unsigned*)p.C)[e >> 1] = pack_bf16x2(s0, s1);
  }
}

// Split-K seam inside the launch (cdna guide, Guideline 16 / the split-K recipe): every K-slice workgroup of a tile has written
// its fp32 slab; each drains its stores, one lane releases at agent scope and draws a ticket from the tile's counter; the
// workgroup that draws the last ticket acquires and combines the tile's slabs -- in slab order, so the result does not depend
// on who arrived last -- instead of a second launch per convolution (232 reduce launches of ~9 us per bf16 training step).
// MEASURED SLOWER and therefore OFF by default (INDM_H16_FUSED_REDUCE=1 enables it; profiles/r06_h16_fused_reduce_ab.txt): the
// combine of a tile runs on ONE workgroup (a weight gradient has 5 - 36 tiles of up to 100 slabs: 67 -> 267 us at 32 x 32,
// 128 -> 128 channels) and every K-slice workgroup pays an agent-scope release (an L2 write-back); the separate reduce launch
// spreads the same reads over the chip and costs one ~2 us boundary inside the captured graph.
// The counter is zero at rest: the last arriver puts it back.  `flag` is a word of the (idle) dynamic LDS region.
template <int NT, int TM>
__device__ __forceinline__ void h16_splitk_seam(const H16Args& p, int tile, int m0, int n0, int* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int t = __hip_atomic_fetch_add(p.cnt + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = t == p.splitk - 1;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(p.cnt + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    *flag = last;
  }
  __syncthreads();
  if (!*flag) return;
  const int cp = threadIdx.x & 63;                 // column pair of the 128-column tile
  const int n = n0 + 2 * cp;
  if (n >= p.N) return;
  for (int r = threadIdx.x >> 6; r < TM; r += NT / 64) {
    const int m = m0 + r;
    if (m < p.M) h16_reduce_pair(p, m, n);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// NW = 8: 256 x 128 block tile, waves 4 x 2 of 64 x 64;  NW = 4: 128 x 128, waves 2 x 2.  32-deep bf16 slabs, three LDS
// buffers (two slabs in flight), one barrier per slab.
template <int NW>
__global__ __launch_bounds__(NW * 64, 4) void h16_conv_kernel(H16Args p) {
  constexpr int TM = NW * 32, TN = 128, RM = 2, RN = 2;
  constexpr int ROWB = 64;                                   // bytes per row and slab (32 bf16)
  constexpr int A_SZ = TM * ROWB, B_SZ = TN * ROWB, BUF = A_SZ + B_SZ;
  constexpr int NBUF = 3, D = NBUF - 1;
  constexpr int QA = 2, QB = 128 / NW / 16;                  // copy instructions per wave and slab (A: 32 rows, B: 128 / NW rows)
  constexpr int NDMA = QA + QB;
  extern __shared__ __attribute__((aligned(16))) char h16_smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_m = (p.M + TM - 1) / TM, tiles_n = (p.N + TN - 1) / TN;
  const int ntile = tiles_m * tiles_n;
  const int lg = xcd_remap(blockIdx.x, ntile * p.splitk);
  const int kz = lg / ntile, lt = lg - kz * ntile;
  const int tm = lt / tiles_n, tn = lt - tm * tiles_n;
  const int m0 = tm * TM, n0 = tn * TN;
  const int kbeg = kz * p.kps;
  const int nk = (min(p.K, kbeg + p.kps) - kbeg) >> 5;

  f32x16 acc[RM][RN];
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // A: the resource starts `shift` elements before the tensor so that a pixel's top-left tap never has a negative offset
  const long shift = ((long)p.pad * p.W + p.pad) * p.Cin;
  const i32x4 arsrc = make_rsrc((const char*)p.A - shift * 2, ((long)p.Nimg * p.H * p.W * p.Cin + 2 * shift) * 2);
  const i32x4 brsrc = make_rsrc(p.B, (long)p.N * p.K * 2);
  unsigned cvoff[QA], cbad[QA], bvoff[QB];
#pragma unroll
  for (int q = 0; q < QA; ++q) {
    const int row = wave * 32 + q * 16 + (lane >> 2);
    const int m = m0 + row;
    const int mm = min(m, p.M - 1);
    const int img = mm >> p.ohw_sh, rem = mm & ((1 << p.ohw_sh) - 1);
    const int oh = rem >> p.ow_sh, ow = rem & ((1 << p.ow_sh) - 1);
    const int ih0 = oh * p.stride - p.pad, iw0 = ow * p.stride - p.pad;
    const int c = (lane & 3) ^ ((row >> 2) & 3);               // the k-chunk that belongs in LDS slot (row, lane & 3)
    cvoff[q] = (unsigned)(((((long)img * p.H + ih0) * p.W + iw0) * p.Cin + shift) * 2 + 16 * c);
    unsigned bad = 0;
    for (int t = 0; t < p.KH * p.KW; ++t) {
      const int kh = t / p.KW, kw = t - kh * p.KW;
      const bool ok = (unsigned)(ih0 + kh) < (unsigned)p.H && (unsigned)(iw0 + kw) < (unsigned)p.W;
      bad |= ok ? 0u : (1u << t);
    }
    cbad[q] = m < p.M ? bad : 0xffffffffu;
  }
#pragma unroll
  for (int q = 0; q < QB; ++q) {
    const int row = wave * (128 / NW) + q * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((row >> 2) & 3);
    const int n = n0 + row;
    bvoff[q] = n < p.N ? (unsigned)((long)n * p.K * 2 + 16 * c) : 0x80000000u;
  }
  // running tap decomposition of the slab index (one tap per slab: Cin % 32 == 0)
  const int slabs_per_tap = p.Cin >> 5;
  int s_tap = (kbeg >> 5) / slabs_per_tap, s_ci = (kbeg >> 5) - s_tap * slabs_per_tap;   // tap, slab inside the tap
  int s_kh = s_tap / p.KW, s_kw = s_tap - s_kh * p.KW;
  unsigned s_k0 = (unsigned)kbeg * 2u;                                                     // byte offset of the slab in a B row

  auto stage = [&](int buf) {   // called for slabs kbeg/32 + 0, 1, 2, ... in order
    char* base = h16_smem + buf * BUF;
    const unsigned soff = (unsigned)((((long)s_kh * p.W + s_kw) * p.Cin + s_ci * 32) * 2);
#pragma unroll
    for (int q = 0; q < QA; ++q) {
      const unsigned oob = (cbad[q] << (31 - s_tap)) & 0x80000000u;
      dma_buffer(lds_addr(base + (wave * 32 + q * 16) * ROWB), cvoff[q] | oob, arsrc, soff);
    }
#pragma unroll
    for (int q = 0; q < QB; ++q)
      dma_buffer(lds_addr(base + A_SZ + (wave * (128 / NW) + q * 16) * ROWB), bvoff[q], brsrc, s_k0);
    s_k0 += 64u;
    if (++s_ci == slabs_per_tap) {
      s_ci = 0;
      ++s_tap;
      if (++s_kw == p.KW) { s_kw = 0; ++s_kh; }
    }
  };

  const int wm = wave >> 1, wn = wave & 1;
  const int mw = wm * 64, nw = wn * 64;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int swz = (l31 >> 2) & 3;

  {
    const int pre = nk < D ? nk : D;
    for (int d = 0; d < pre; ++d) stage(d);
    if (nk >= D) wait_vm<NDMA * (D - 1)>();
    else wait_vm<0>();
    lds_barrier();
  }
  int buf = 0, sbuf = D;
  for (int s = 0; s < nk; ++s) {
    if (s + D < nk) stage(sbuf);
    const char* as = h16_smem + buf * BUF;
    const char* bs = as + A_SZ;
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
      const int ch = ((sg * 2 + lhi) ^ swz) << 4;
      bf16x8 ha[RM], hb[RN];
#pragma unroll
      for (int i = 0; i < RM; ++i) ha[i] = *(const bf16x8*)(as + (mw + i * 32 + l31) * ROWB + ch);
#pragma unroll
      for (int j = 0; j < RN; ++j) hb[j] = *(const bf16x8*)(bs + (nw + j * 32 + l31) * ROWB + ch);
#pragma unroll
      for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha[i], hb[j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (s + 2 < nk) wait_vm<NDMA>();
    else wait_vm<0>();
    lds_barrier();
    buf = buf == D ? 0 : buf + 1;
    sbuf = sbuf == D ? 0 : sbuf + 1;
  }

  if (p.splitk > 1) {   // fp32 partial sums; h16_reduce_kernel applies the epilogue
    float* W = p.ws + (long)kz * p.M * p.N;
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
      for (int j = 0; j < RN; ++j) {
        const int n = n0 + nw + j * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          if (n < p.N && m < p.M) W[(long)m * p.N + n] = acc[i][j][r];
        }
      }
    if (p.cnt) h16_splitk_seam<NW * 64, TM>(p, lt, m0, n0, (int*)h16_smem);
    return;
  }
  const bool odd = lane & 1;
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j) {
      if (p.out_f32) h16_epilogue_block<true>(p, acc[i][j], m0 + mw + i * 32, n0 + nw + j * 32 + l31, lhi, odd);
      else h16_epilogue_block<false>(p, acc[i][j], m0 + mw + i * 32, n0 + nw + j * 32 + l31, lhi, odd);
    }
}

// split-K combine + epilogue (fixed summation order: deterministic), bf16 or fp32 output
__global__ __launch_bounds__(256) void h16_reduce_kernel(H16Args p) {
  const long total2 = (long)p.M * p.N / 2;   // column pairs (N is even)
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total2; i += (long)gridDim.x * blockDim.x) {
    const long e = 2 * i;
    const int m = (int)(e / p.N), n = (int)(e - (long)m * p.N);
    h16_reduce_pair(p, m, n);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient.  256 x 128 tile of dW[(tap, ci)][co], 8 waves of 64 x 64, slabs of 32 output pixels.
//   LDS images in memory order: A [32 px][256 m] (512-byte rows), B [32 px][128 n] (256-byte rows); the 64-byte units of a row
//   are XOR-swizzled with (px & 3) on the source address so that the four pixel rows one ds_read_b64_tr_b16 touches per
//   half-wave fall into four different bank quarters.
//   A fragment (lane: row m = l31 of a 32-row block, k = 8 lhi + t pixels) = two transposing reads: within a 16-lane
//   group, lane i supplies the address of 4 consecutive m of pixel row (i >> 2) and receives column i of the 4 x 16 block.
__global__ __launch_bounds__(512, 4) void h16_wgrad_kernel(H16Args p) {
  constexpr int TM = 256, TN = 128, RM = 2, RN = 2;
  constexpr int A_ROWB = TM * 2, B_ROWB = TN * 2;
  constexpr int A_SZ = 32 * A_ROWB, B_SZ = 32 * B_ROWB, BUF = A_SZ + B_SZ;
  constexpr int NBUF = 3, D = NBUF - 1;
  constexpr int NDMA = 3;   // per wave and slab: A 2 (two pixel rows each), B 1 (four pixel rows)
  extern __shared__ __attribute__((aligned(16))) char h16_smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_m = (p.M + TM - 1) / TM, tiles_n = (p.N + TN - 1) / TN;
  const int ntile = tiles_m * tiles_n;
  const int lg = xcd_remap(blockIdx.x, ntile * p.splitk);
  const int kz = lg / ntile, lt = lg - kz * ntile;
  const int tm = lt / tiles_n, tn = lt - tm * tiles_n;
  const int m0 = tm * TM, n0 = tn * TN;
  const int kbeg = kz * p.kps;
  const int nk = (min(p.K, kbeg + p.kps) - kbeg) >> 5;

  f32x16 acc[RM][RN];
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const i32x4 arsrc = make_rsrc(p.A, (long)p.Nimg * p.H * p.W * p.Cin * 2);
  const i32x4 brsrc = make_rsrc(p.B, (long)p.K * p.N * 2);
  // A: copy q of this wave covers pixel rows 4 wave + 2 q + (lane >> 5); the lane's LDS slot is 16-byte chunk (lane & 31),
  // which holds source chunk (lane & 31) ^ ((px & 3) << 2): eight consecutive m = (tap, ci .. ci + 7)
  int a_kh[2], a_kw[2];
  unsigned a_fix[2];      // ((kh * W + kw) * Cin + ci) * 2, or bit 31 when the chunk lies beyond M
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int px = 4 * wave + 2 * q + (lane >> 5);
    const int csrc = (lane & 31) ^ ((px & 3) << 2);
    const int m = m0 + 8 * csrc;
    const int mm = min(m, p.M - 8);
    const int tap = mm / p.Cin, ci = mm - tap * p.Cin;
    a_kh[q] = tap / p.KW;
    a_kw[q] = tap - a_kh[q] * p.KW;
    a_fix[q] = m < p.M ? (unsigned)((((long)a_kh[q] * p.W + a_kw[q]) * p.Cin + ci) * 2) : 0x80000000u;
  }
  // B: the wave's copy covers pixel rows 4 wave + (lane >> 4); slot chunk (lane & 15) holds source chunk (lane & 15) ^ ((px & 3) << 2)
  unsigned b_fix;
  {
    const int px = 4 * wave + (lane >> 4);
    const int csrc = (lane & 15) ^ ((px & 3) << 2);
    const int n = n0 + 8 * csrc;
    b_fix = n < p.N ? (unsigned)(((long)px * p.N + n) * 2) : 0x80000000u;
  }
  int s_p0 = kbeg;   // first output pixel of the next slab to stage

  auto stage = [&](int buf) {
    char* base = h16_smem + buf * BUF;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int pix = s_p0 + 4 * wave + 2 * q + (lane >> 5);
      const int img = pix >> p.ohw_sh, rem = pix & ((1 << p.ohw_sh) - 1);
      const int ih = (rem >> p.ow_sh) * p.stride - p.pad + a_kh[q];
      const int iw = (rem & ((1 << p.ow_sh) - 1)) * p.stride - p.pad + a_kw[q];
      const bool bad = (unsigned)ih >= (unsigned)p.H || (unsigned)iw >= (unsigned)p.W;
      // (top-left offset of the pixel; the tap's (kh, kw, ci) part is in a_fix) -- 32-bit arithmetic: the host checked < 2 GB
      const unsigned off = (unsigned)(((img * p.H + (ih - a_kh[q])) * p.W + (iw - a_kw[q])) * p.Cin * 2) + a_fix[q];
      dma_buffer(lds_addr(base + (4 * wave + 2 * q) * A_ROWB), bad ? 0x80000000u : off, arsrc, 0u);
    }
    dma_buffer(lds_addr(base + A_SZ + (4 * wave) * B_ROWB), b_fix, brsrc, (unsigned)s_p0 * (unsigned)p.N * 2u);
    s_p0 += 32;
  };

  const int wm = wave >> 1, wn = wave & 1;
  const int mw = wm * 64, nw = wn * 64;
  const int l31 = lane & 31, lhi = lane >> 5;
  // transposing reads: lane supplies (pixel row jj = (lane & 15) >> 2, 4 consecutive rows/cols starting at 16 g1 + 4 (lane & 3))
  const int jj = (lane & 15) >> 2, g1 = (lane >> 4) & 1, c4 = lane & 3;
  // byte offset inside a pixel row of the 8-byte piece this lane addresses, before the swizzle; the 64-byte unit index is
  // XORed with (px & 3) = jj (the slab's pixel rows 16 sg + 8 lhi + 4 h + jj keep px & 3 == jj)
  unsigned a_off[RM], b_off[RN];
#pragma unroll
  for (int i = 0; i < RM; ++i) {
    const unsigned bo = (unsigned)(mw + i * 32 + 16 * g1 + 4 * c4) * 2u;
    a_off[i] = (bo ^ ((unsigned)jj << 6)) + (unsigned)(8 * lhi + jj) * A_ROWB;
  }
#pragma unroll
  for (int j = 0; j < RN; ++j) {
    const unsigned bo = (unsigned)(nw + j * 32 + 16 * g1 + 4 * c4) * 2u;
    b_off[j] = (bo ^ ((unsigned)jj << 6)) + (unsigned)(8 * lhi + jj) * B_ROWB;
  }
  auto tr_read = [](const char* q) -> s16x4 {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)q);
  };
  auto frag = [&](const char* img, unsigned off, int rowb, int sg) -> bf16x8 {
    const char* q = img + off + (16 * sg) * rowb;
    const s16x4 lo = tr_read(q), hi = tr_read(q + 4 * rowb);
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
  };

  {
    const int pre = nk < D ? nk : D;
    for (int d = 0; d < pre; ++d) stage(d);
    if (nk >= D) wait_vm<NDMA * (D - 1)>();
    else wait_vm<0>();
    lds_barrier();
  }
  int buf = 0, sbuf = D;
  for (int s = 0; s < nk; ++s) {
    if (s + D < nk) stage(sbuf);
    const char* as = h16_smem + buf * BUF;
    const char* bs = as + A_SZ;
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
      bf16x8 ha[RM], hb[RN];
#pragma unroll
      for (int i = 0; i < RM; ++i) ha[i] = frag(as, a_off[i], A_ROWB, sg);
#pragma unroll
      for (int j = 0; j < RN; ++j) hb[j] = frag(bs, b_off[j], B_ROWB, sg);
#pragma unroll
      for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha[i], hb[j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (s + 2 < nk) wait_vm<NDMA>();
    else wait_vm<0>();
    lds_barrier();
    buf = buf == D ? 0 : buf + 1;
    sbuf = sbuf == D ? 0 : sbuf + 1;
  }
  float* Wp = p.ws + (long)kz * p.M * p.N;   // always through the partial slabs (splitk >= 1) + h16_reduce_kernel
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j) {
      const int n = n0 + nw + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (n < p.N && m < p.M) Wp[(long)m * p.N + n] = acc[i][j][r];
      }
    }
  if (p.cnt) h16_splitk_seam<512, TM>(p, lt, m0, n0, (int*)h16_smem);
}

// ---------------------------------------------------------------------------------------------------------------------
// Packed bf16 copies of the weights, all convolutions of the network in ONE launch (a job table in device memory):
//   src  fp32 RSCK [T][Cin][Cout]  (1x1 / NIN: T = 1, [Cin][Cout])
//   fwd  bf16 [Cout][T][Cin]        B operand of the forward convolution (k = (tap, ci) contiguous)
//   dg   bf16 [Cin][T][Cout], taps flipped: dg[ci][t][co] = src[T - 1 - t][ci][co]   B operand of the data-grad
struct PackJob {
  const float* src;
  void* fwd;
  void* dg;
  int T, Cin, Cout;
  int tile0;   // first workgroup of this job: one workgroup per (tap, 32 x 32 tile of [Cin][Cout])
  int dg_ld;   // row stride of dg in elements (0: T * Cout).  Wider: several weights side by side in one [Cin][sum Cout] pack --
  int pad_;    // the q | k | v projections of an attention block as ONE data-grad operand
};
__global__ __launch_bounds__(256) void h16_pack_kernel(const PackJob* __restrict__ jobs, int njobs) {
  __shared__ float tile[32][33];
  int lo = 0, hi = njobs - 1;
  const int b = blockIdx.x;
  while (lo < hi) {   // last job whose tile0 <= b
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].tile0 <= b) lo = mid; else hi = mid - 1;
  }
  const PackJob J = jobs[lo];
  const int tci = (J.Cin + 31) >> 5, tco = (J.Cout + 31) >> 5;
  int r = b - J.tile0;
  const int t = r / (tci * tco);
  r -= t * tci * tco;
  const int ci0 = (r / tco) * 32, co0 = (r % tco) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  u16* fwd = (u16*)J.fwd;
  u16* dg = (u16*)J.dg;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int ci = ci0 + ty + 8 * k, co = co0 + tx;
    float v = 0.f;
    if (ci < J.Cin && co < J.Cout) v = J.src[((long)t * J.Cin + ci) * J.Cout + co];
    tile[ty + 8 * k][tx] = v;
    if (dg && ci < J.Cin && co < J.Cout) {
      const __bf16 h = (__bf16)v;
      dg[(long)ci * (J.dg_ld ? J.dg_ld : J.T * J.Cout) + (long)(J.T - 1 - t) * J.Cout + co] = *(const u16*)&h;
    }
  }
  __syncthreads();
  if (fwd) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int co = co0 + ty + 8 * k, ci = ci0 + tx;
      if (ci < J.Cin && co < J.Cout) {
        const __bf16 h = (__bf16)tile[tx][ty + 8 * k];
        fwd[((long)co * J.T + t) * J.Cin + ci] = *(const u16*)&h;
      }
    }
  }
}

__global__ __launch_bounds__(256) void h16_from_f32_kernel(const float* __restrict__ x, unsigned* __restrict__ y, long n2) {
  // n2 = pairs; 4 pairs (8 values) per thread and iteration where they exist
  const long n8 = n2 >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const f32x4 a = *(const f32x4*)(x + 8 * i), b = *(const f32x4*)(x + 8 * i + 4);
    i32x4 o;
    o[0] = (int)pack_bf16x2(a[0], a[1]); o[1] = (int)pack_bf16x2(a[2], a[3]);
    o[2] = (int)pack_bf16x2(b[0], b[1]); o[3] = (int)pack_bf16x2(b[2], b[3]);
    *(i32x4*)(y + 4 * i) = o;
  }
  for (long i = (n8 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x)
    y[i] = pack_bf16x2(x[2 * i], x[2 * i + 1]);
}
__global__ __launch_bounds__(256) void h16_to_f32_kernel(const unsigned* __restrict__ x, float* __restrict__ y, long n2, float alpha) {
  const long n8 = n2 >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const i32x4 w = *(const i32x4*)(x + 4 * i);
    f32x4 a, b;
    a[0] = alpha * bf16_lo((unsigned)w[0]); a[1] = alpha * bf16_hi((unsigned)w[0]);
    a[2] = alpha * bf16_lo((unsigned)w[1]); a[3] = alpha * bf16_hi((unsigned)w[1]);
    b[0] = alpha * bf16_lo((unsigned)w[2]); b[1] = alpha * bf16_hi((unsigned)w[2]);
    b[2] = alpha * bf16_lo((unsigned)w[3]); b[3] = alpha * bf16_hi((unsigned)w[3]);
    *(f32x4*)(y + 8 * i) = a;
    *(f32x4*)(y + 8 * i + 4) = b;
  }
  for (long i = (n8 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) {
    const unsigned w = x[i];
    y[2 * i] = alpha * bf16_lo(w);
    y[2 * i + 1] = alpha * bf16_hi(w);
  }
}

inline bool al16(const void* q) { return ((uintptr_t)q & 15) == 0; }
inline int ilog2_exact(int v) { return (v > 0 && !(v & (v - 1))) ? __builtin_ctz((unsigned)v) : -1; }

static long h16_tuned(const char* name, long dflt) {
  const char* v = getenv(name);
  return v ? atol(v) : dflt;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// C ABI (include/indm_hip.h)
// ---------------------------------------------------------------------------------------------------------------------

// y[n, oh, ow, co] = alpha * (sum x[n, oh s + kh - pad, ow s + kw - pad, ci] wp[co][(kh, kw, ci)] + bias[co] + bias2[n][co]) + beta * R
//   x bf16 NHWC, wp bf16 packed [Cout][KH*KW*Cin] (indm_h16_pack_weights), y bf16 (out_f32 != 0: fp32), R bf16 (y's shape).
// The data-grad of a stride-1 convolution is the same call on dy with the tap-flipped pack [Cin][KH*KW*Cout] and pad' = K - 1 - pad;
// a plain NT GEMM (NIN, 1x1) is KH = KW = 1.  Returns 1 (not an error) when the kernel does not cover the geometry
// (Cin % 32, Cout % 2, OH*OW / OW not powers of two, tensors >= 2 GB, unaligned pointers): the caller converts and takes the fp32-storage path.
INDM_API int indm_h16_conv2d(const void* x, const void* wp, void* y, int Nimg, int H, int W, int Cin, int Cout, int KH, int KW,
                             int stride, int pad, int OH, int OW, float alpha, const float* bias, const float* bias2,
                             const void* R, float beta, int out_f32, float* ws, long ws_floats, int* tile_counters,
                             int n_counters, void* stream) {
  if (Nimg < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || OH <= 0 || OW <= 0 || KH <= 0 || KW <= 0) return INDM_E_SHAPE;
  if (Nimg == 0) return INDM_OK;
  const int ow_sh = ilog2_exact(OW), ohw_sh = ilog2_exact(OH * OW);
  if ((Cin & 31) || (Cout & 1) || ow_sh < 0 || ohw_sh < 0 || KH * KW > 31 || pad < 0 || stride < 1) return 1;
  if (!al16(x) || !al16(wp) || ((uintptr_t)y & 3) || ((uintptr_t)R & 3)) return 1;
  const long in_bytes = ((long)Nimg * H * W + 2L * pad * (W + 1)) * Cin * 2;
  const long M = (long)Nimg * OH * OW;
  if (in_bytes >= (1L << 31) || M * Cout >= (1L << 31) || (long)Cout * KH * KW * Cin * 2 >= (1L << 31)) return 1;
  if (out_f32 && R) return 1;
  H16Args p = {};
  p.A = x; p.B = wp; p.C = y;
  p.M = (int)M; p.N = Cout; p.K = KH * KW * Cin;
  p.Nimg = Nimg; p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad;
  p.ow_sh = ow_sh; p.ohw_sh = ohw_sh;
  p.alpha = alpha; p.beta = beta; p.bias = bias; p.bias2 = bias2; p.rpg = OH * OW; p.R = R; p.out_f32 = out_f32;
  p.ws = ws;
  hipStream_t st = (hipStream_t)stream;
  // tile / split choice: 256-row tiles when they give every CU its two workgroups, else 128-row tiles; split-K (fp32 partial
  // slabs + reduce) when even those leave most of the chip idle and the reduction is long
  const long t256 = (long)cdiv(p.M, 256) * cdiv(p.N, 128), t128 = (long)cdiv(p.M, 128) * cdiv(p.N, 128);
  const long big_min = h16_tuned("INDM_H16_BIG_MIN", 512);
  const bool big = t256 >= big_min;
  const long tiles = big ? t256 : t128;
  long want = 1;
  if (tiles < 384 && ws) {
    want = h16_tuned("INDM_H16_SPLIT_TARGET", 768) / tiles;
    const long maxs = p.K / (32 * 8);   // at least 8 slabs per split
    if (want > maxs) want = maxs;
    while (want > 1 && want * M * Cout > ws_floats) --want;
    if (want < 1) want = 1;
  }
  p.kps = cdiv(cdiv(p.K, (int)want), 32) * 32;
  p.splitk = cdiv(p.K, p.kps);
  p.cnt = (p.splitk > 1 && tile_counters && tiles <= n_counters && h16_tuned("INDM_H16_FUSED_REDUCE", 0)) ? tile_counters : nullptr;
  const dim3 grid((unsigned)(tiles * p.splitk));
  if (big) {
    constexpr size_t bytes = 3 * (256 + 128) * 64;
    if (indm_lds_optin<h16_conv_kernel<8>>(bytes) != INDM_OK) return INDM_E_LAUNCH;
    hipLaunchKernelGGL(h16_conv_kernel<8>, grid, dim3(512), bytes, st, p);
  } else {
    constexpr size_t bytes = 3 * (128 + 128) * 64;
    if (indm_lds_optin<h16_conv_kernel<4>>(bytes) != INDM_OK) return INDM_E_LAUNCH;
    hipLaunchKernelGGL(h16_conv_kernel<4>, grid, dim3(256), bytes, st, p);
  }
  if (hipGetLastError() != hipSuccess) return INDM_E_LAUNCH;
  if (p.splitk > 1 && !p.cnt) {
    hipLaunchKernelGGL(h16_reduce_kernel, dim3(ew_grid(M * Cout / 2, 256)), dim3(256), 0, st, p);
    if (hipGetLastError() != hipSuccess) return INDM_E_LAUNCH;
  }
  return INDM_OK;
}

// dw[kh, kw, ci, co] = alpha * sum_{n, oh, ow} x[n, oh s + kh - pad, ow s + kw - pad, ci] dy[n, oh, ow, co]   (fp32, RSCK)
//   x bf16 NHWC [Nimg, H, W, Cin], dy bf16 [Nimg, OH, OW, Cout]; split-K over the output pixels into ws (fp32 partial slabs,
//   at least KH*KW*Cin*Cout floats), fixed-order reduce.  Returns 1 when the geometry is not covered (Cin % 8, Cout % 8,
//   pixels % 32, OH*OW / OW not powers of two, tensors >= 2 GB).
INDM_API int indm_h16_conv2d_wgrad(const void* x, const void* dy, float* dw, int Nimg, int H, int W, int Cin, int Cout, int KH,
                                   int KW, int stride, int pad, int OH, int OW, float alpha, float* ws, long ws_floats,
                                   int* tile_counters, int n_counters, void* stream) {
  if (Nimg < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || OH <= 0 || OW <= 0 || KH <= 0 || KW <= 0) return INDM_E_SHAPE;
  if (!dw || !ws) return INDM_E_ARG;
  const int ow_sh = ilog2_exact(OW), ohw_sh = ilog2_exact(OH * OW);
  const long P = (long)Nimg * OH * OW, M = (long)KH * KW * Cin;
  if ((Cin & 7) || (Cout & 7) || (P & 31) || P <= 0 || ow_sh < 0 || ohw_sh < 0 || pad < 0 || stride < 1) return 1;
  if (!al16(x) || !al16(dy) || ((uintptr_t)dw & 7)) return 1;
  if ((long)Nimg * H * W * Cin * 2 >= (1L << 31) || P * Cout * 2 >= (1L << 31) || M * Cout > ws_floats) return 1;
  H16Args p = {};
  p.A = x; p.B = dy; p.C = dw;
  p.M = (int)M; p.N = Cout; p.K = (int)P;
  p.Nimg = Nimg; p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad;
  p.ow_sh = ow_sh; p.ohw_sh = ohw_sh;
  p.alpha = alpha; p.beta = 0.f; p.rpg = 1; p.out_f32 = 1; p.ws = ws;
  const long tiles = (long)cdiv(p.M, 256) * cdiv(p.N, 128);
  long want = h16_tuned("INDM_H16_WGRAD_TARGET", 512) / tiles;
  const long maxs = p.K / (32 * 4);   // at least 4 slabs per split
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  while (want > 1 && want * M * Cout > ws_floats) --want;
  p.kps = cdiv(cdiv(p.K, (int)want), 32) * 32;
  p.splitk = cdiv(p.K, p.kps);
  p.cnt = (tile_counters && tiles <= n_counters && h16_tuned("INDM_H16_FUSED_REDUCE", 0)) ? tile_counters : nullptr;
  hipStream_t st = (hipStream_t)stream;
  constexpr size_t bytes = 3 * (256 + 128) * 64;
  if (indm_lds_optin<h16_wgrad_kernel>(bytes) != INDM_OK) return INDM_E_LAUNCH;
  hipLaunchKernelGGL(h16_wgrad_kernel, dim3((unsigned)(tiles * p.splitk)), dim3(512), bytes, st, p);
  if (hipGetLastError() != hipSuccess) return INDM_E_LAUNCH;
  if (p.cnt) return INDM_OK;
  hipLaunchKernelGGL(h16_reduce_kernel, dim3(ew_grid(M * Cout / 2, 256)), dim3(256), 0, st, p);
  return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
}

// jobs: device array of njobs records {src fp32 RSCK, fwd bf16 [Cout][T][Cin] or NULL, dg bf16 [Cin][T][Cout] taps flipped or
// NULL, T, Cin, Cout, tile0, dg_ld, pad} (struct indm_h16_pack_job, include/indm_hip.h); ntiles = total workgroups = sum over the jobs of
// T * ceil(Cin / 32) * ceil(Cout / 32), tile0 = the running sum.
INDM_API int indm_h16_pack_weights(const void* jobs, int njobs, int ntiles, void* stream) {
  if (njobs < 0 || ntiles < 0) return INDM_E_SHAPE;
  if (njobs == 0 || ntiles == 0) return INDM_OK;
  if (!jobs) return INDM_E_ARG;
  hipLaunchKernelGGL(h16_pack_kernel, dim3((unsigned)ntiles), dim3(256), 0, (hipStream_t)stream, (const PackJob*)jobs, njobs);
  return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
}

// fp32 <-> bf16 (RNE) converters, n even, 4-byte aligned bf16 / 8-byte aligned fp32 (16 / 32 bytes for the vector path)
INDM_API int indm_h16_from_f32(const float* x, void* y, long n, void* stream) {
  if (n < 0 || (n & 1)) return INDM_E_SHAPE;
  if (n == 0) return INDM_OK;
  if (!al16(x) || !al16(y)) return INDM_E_ARG;
  hipLaunchKernelGGL(h16_from_f32_kernel, dim3(ew_grid(n / 8 + 1, 256)), dim3(256), 0, (hipStream_t)stream, x, (unsigned*)y, n / 2);
  return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
}
INDM_API int indm_h16_to_f32(const void* x, float* y, long n, float alpha, void* stream) {
  if (n < 0 || (n & 1)) return INDM_E_SHAPE;
  if (n == 0) return INDM_OK;
  if (!al16(x) || !al16(y)) return INDM_E_ARG;
  hipLaunchKernelGGL(h16_to_f32_kernel, dim3(ew_grid(n / 8 + 1, 256)), dim3(256), 0, (hipStream_t)stream, (const unsigned*)x, y, n / 2, alpha);
  return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
}

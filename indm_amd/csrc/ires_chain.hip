// Back-to-back contraction of the residual flow's vector-Jacobian chain (iresblock.py:253-273).
//
// One Neumann / power-series iteration multiplies a 3-/12-channel vector through the transposed g-net:
//     t3 = C3^T v            [M, 512]   (thin transposed 3x3 conv, igemm_thin.hip)
//     t2 = (t3 .* d2) W2^T   [M, 512]   <- the dominant fp32 MFMA contraction of the step
//     P  = (t2 .* d1) W1f^T  [M, 9C]    <- HBM-bound thin-N GEMM (9C = 27 / 108), then a 9-tap gather
// Run as separate launches, t2 (268 MB at batch 128) is written by the second and re-read -- together with its gate
// d1 -- by the third.  This kernel keeps the 512-wide row of t2 ON CHIP: a workgroup owns 64 rows of M and ALL 512
// columns of t2, so the second contraction (over those 512 columns) finishes inside the workgroup:
//   stage 1   acc = W2T^T-major product: D[n][m] = sum_k W2T[k][n] * (t3 .* d2)[m][k]   (fp32 MFMA 32x32x2, operands swapped so that
//             the accumulator of lane (l31, lhi) holds t2[m = l31][n = 8a + b + 4 lhi], exactly the A-operand layout
//             of the next MFMA: no LDS transpose between the two GEMMs)
//   gate      acc *= d1[m][n]            (16-byte loads, 128-byte rows completed across 4 instructions)
//   stage 2   out[m][j] += sum_n acc[n][m] * W1p[n][j]   per wave over its 64 n's, then an 8-way reduction over
//             the waves through LDS and a [64, 9C] store.
// Stage 1 is the direct-to-LDS kernel of igemm.hip (global_load_lds_dwordx4, [row][16 k] images XOR-swizzled on the
// source address, half-slab skew) on a 64 x 512 block tile: the same 40 KB of operand traffic per 1 MFLOP slab as its
// 256 x 128 tile (A + gate rows + B rows = 640 either way), A and its gate are read exactly once from HBM, W2T
// (1 MB) stays in L2.  Saves, per iteration at batch 128: the 268 MB write, the 536 MB gated re-read and one launch.
#include "indm_common.h"
// the direct-to-LDS copy helpers write M0 from inline asm and say so in their clobber lists (M0 is a reserved register:
// hipcc warns about the clobber, which is exactly the declaration its M0-initialisation merging needs to see)
#pragma clang diagnostic ignored "-Winline-asm"

namespace {

constexpr int BK = 16, TM = 64, TN = 512, NW = 8;
constexpr int A_SZ = TM * BK, G_SZ = TM * BK, B_SZ = BK * TN, BUF = A_SZ + G_SZ + B_SZ;  // floats per LDS buffer
// (ires_chain_kernel is templated on its row count TMV = 64 | 32: the 32-row form doubles the workgroup count when M is
// too small to give every CU its two 64-row workgroups -- the per-rank shards of BASELINE configs #2 - #5; the LDS
// layout keeps the 64-row offsets, the upper halves of the A / gate images simply stay unused)

struct ChainArgs {
  const float* A;    // t3  [M, K]
  const float* G;    // d2  [M, K]
  const float* B;    // W2T [K, 512] row-major
  const float* D1;   // d1  [M, 512]
  const float* W1p;  // [512, 32*JB] row-major, zero-padded beyond J
  float* P;          // [M, J]
  int M, K, J;
  float* T2;         // optional [M, 512]: the intermediate (t3 .* d2) W2T BEFORE the d1 gate (the gradient-carrying
                     // chain needs it in its backward)
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// One direct-to-LDS copy (64 lanes x 16 bytes) in the `voffset, s[base:base+1] offset:imm` form: the wave-uniform base
// lives in an SGPR pair that the SCALAR unit advances from slab to slab, the per-lane 32-bit byte offset never changes,
// and the immediate is added to the global AND the LDS address (M0).  hipcc lowers __builtin_amdgcn_global_load_lds on a
// loop-carried pointer to a 64-bit VALU add per copy; tools/probe/mfma_model.hip prices that arithmetic at 9-11 % of
// a 32-MFMA slab (non-MFMA VALU instructions do not overlap with the MFMAs on this part): 125 -> 141 TF in the model.
// The asm is opaque to the compiler: the callers order it with their own s_waitcnt vmcnt + barriers, as before.
template <int IMM>
__device__ __forceinline__ void dma_s(unsigned lds_byte_addr, unsigned voff, const void* sbase) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3"
               :: "s"(lds_byte_addr), "v"(voff), "s"(sbase), "n"(IMM) : "memory", "m0");
}
// four copies 1 KB apart in global memory and in LDS: one M0 value, one address, four immediates (ONE asm statement: nothing
// can come between the M0 write and its readers)
__device__ __forceinline__ void dma_s_x4(unsigned lds_byte_addr, unsigned voff, const void* sbase) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, %2 offset:0\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
               "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072"
               :: "s"(lds_byte_addr), "v"(voff), "s"(sbase) : "memory", "m0");
}
__device__ __forceinline__ void dma_s_x2(unsigned lds_byte_addr, unsigned voff, const void* sbase) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, %2 offset:0\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024"
               :: "s"(lds_byte_addr), "v"(voff), "s"(sbase) : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_addr(const float* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ float bf16_round(float x) { return (float)(__bf16)x; }

// Everything after the first contraction, shared by the kernels below: optional store of the intermediate, the d1
// gate, the second contraction against the W1 panel and the 8-way cross-wave reduction (see the header comment).
// acc[i][j][r] = t2[m = m0 + 32j + l31][n = nw + 32i + 8(r>>2) + (r&3) + 4 lhi] on entry; every wave has passed a
// workgroup barrier after its last read of the operand buffers (`red` overlays them).
template <int JB, bool BF16, bool T2OUT, int TMV = 64>
__device__ __forceinline__ void chain_epilogue(const ChainArgs& p, f32x16 (&acc)[2][TMV / 32], float* smem, int tid, int wave,
                                               int m0, int nw, int l31, int lhi) {
  constexpr int MB = TMV / 32;   // 32-row m-blocks per workgroup
  if constexpr (T2OUT) {  // 16-byte pieces, a 128-byte row segment is completed by four stores of the same wave
#pragma unroll
    for (int j = 0; j < MB; ++j) {
      float* trow = p.T2 + (long)(m0 + j * 32 + l31) * TN + nw + 4 * lhi;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int a = 0; a < 4; ++a)
          *(float4*)(trow + i * 32 + 8 * a) =
              make_float4(acc[i][j][4 * a], acc[i][j][4 * a + 1], acc[i][j][4 * a + 2], acc[i][j][4 * a + 3]);
    }
  }
  // ---- gate: acc[i][j][r] = t2[m = m0 + 32j + l31][n = nw + 32i + 8(r>>2) + (r&3) + 4 lhi]  *=  d1[m][n] ----
#pragma unroll
  for (int j = 0; j < MB; ++j) {
    const float* drow = p.D1 + (long)(m0 + j * 32 + l31) * TN + nw + 4 * lhi;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float4 g[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) g[a] = *(const float4*)(drow + i * 32 + 8 * a);
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        acc[i][j][4 * a + 0] *= g[a].x;
        acc[i][j][4 * a + 1] *= g[a].y;
        acc[i][j][4 * a + 2] *= g[a].z;
        acc[i][j][4 * a + 3] *= g[a].w;
      }
      if constexpr (BF16) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = bf16_round(acc[i][j][r]);
      }
    }
  }

  // ---- stage 2: out[m][j] = sum over this wave's 64 n's of u1[m][n] * W1p[n][j]; 8-way reduce through LDS ----
  const int JP = 32 * JB;
  float* red = smem;  // [8 waves][TMV m][32 j] <= 64 KB (the operand buffers are dead)
#pragma unroll 1
  for (int jb = 0; jb < JB; ++jb) {
    f32x16 out[MB];
#pragma unroll
    for (int j = 0; j < MB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) out[j][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float w[16];
#pragma unroll
      for (int r = 0; r < 16; ++r)
        w[r] = p.W1p[(long)(nw + i * 32 + 8 * (r >> 2) + (r & 3) + 4 * lhi) * JP + jb * 32 + l31];
      if constexpr (BF16) {
#pragma unroll
        for (int r = 0; r < 16; ++r) w[r] = bf16_round(w[r]);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int j = 0; j < MB; ++j)
          out[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[i][j][r], w[r], out[j], 0, 0, 0);  // D'[m][j]
    }
    if (jb > 0) __syncthreads();  // the previous round's reduction has read `red`
#pragma unroll
    for (int j = 0; j < MB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = j * 32 + 8 * (r >> 2) + (r & 3) + 4 * lhi;
        red[(wave * TMV + m) * 32 + l31] = out[j][r];
      }
    __syncthreads();
    // TMV x 32 outputs, 4 per thread: thread -> (m = tid / 8, j = (tid % 8) * 4 .. +3)
    if (tid < TMV * 8) {
      const int m = tid >> 3, j0 = (tid & 7) * 4;
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int wv = 0; wv < NW; ++wv) {
        const float4 v = *(const float4*)(red + (wv * TMV + m) * 32 + j0);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      float* dst = p.P + (long)(m0 + m) * p.J + jb * 32 + j0;
      const int left = p.J - (jb * 32 + j0);
      if (left > 0) dst[0] = s.x;
      if (left > 1) dst[1] = s.y;
      if (left > 2) dst[2] = s.z;
      if (left > 3) dst[3] = s.w;
    }
  }
}

// BF16: MFMA inputs rounded to bf16 (RNE), fp32 accumulate -- stage 1 packs both 4-deep fragments of a slab into one
// v_mfma_f32_32x32x16_bf16 (as igemm_glds_kernel<.., BF16>); stage 2 rounds its operands to bf16 and keeps the fp32
// MFMA (products of bf16 values are exact in fp32: identical to a bf16 MFMA).
// T2OUT: the launch also stores the intermediate (its own instantiation: a distinct kernel name in profiles, and no
// run-time branch around the store).
template <int JB, bool BF16, bool T2OUT, int TMV = 64>
__global__ __launch_bounds__(NW * 64, 4) void ires_chain_kernel(ChainArgs p) {
  constexpr int MB = TMV / 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * TMV;
  const int nk = p.K / BK;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int swz = (l31 >> 2) & 3;
  const int nw = wave * 64;

  f32x16 acc[2][MB];  // [n-block][m-block]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < MB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto dma = [&](const float* src, float* dst_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst_wave_base, 16, 0, 0);
  };
  // per-lane running source pointers: waves 0-3 stage 16 rows of A each, waves 4-7 the same rows of the gate;
  // every wave stages two k-rows of B (2 KB each = two wave-instructions)
  const int arow = (wave & 3) * 16 + (lane >> 2);
  const int achunk = (lane & 3) ^ ((arow >> 2) & 3);
  // source addresses = wave-uniform running base (SGPR pair, advanced by scalar adds) + a per-lane 32-bit byte offset
  // that never changes: the copies then use the `voffset, s[base]` form and the loop issues NO vector address arithmetic
  const char* pa = (const char*)((wave < 4 ? p.A : p.G) + (long)m0 * p.K);
  const unsigned a_off = ((unsigned)arow * (unsigned)p.K + 4u * achunk) * 4u;
  const int a_dst = (wave < 4 ? 0 : A_SZ) + (wave & 3) * 16 * BK;
  const char* pb = (const char*)(p.B + (long)(wave * 2) * TN);
  const unsigned b_off = 16u * lane;
  // The wave's four B copies (k-row 2w: [0,256) [256,512); k-row 2w+1: ...) are 1 KB apart in global memory AND in LDS:
  // ONE address register and ONE M0 value, the rest is the instruction's immediate offset (it is added to both
  // addresses).  A loop model (tools/probe/mfma_model.hip) priced the per-copy address arithmetic -- a 64-bit VALU add,
  // an M0 write and its hazard wait each -- at 9 % of this loop: non-MFMA VALU work does NOT overlap with the MFMAs.
  auto stage = [&](int buf) {
    float* base = smem + buf * BUF;
    if ((wave & 3) < TMV / 16) dma_s<0>(lds_addr(base + a_dst), a_off, pa);   // 16 rows of A / of the gate per wave
    pa += BK * sizeof(float);
    dma_s_x4(lds_addr(base + A_SZ + G_SZ + wave * 2 * TN), b_off, pb);
    pb += (long)BK * TN * sizeof(float);
  };

  auto frags = [&](int buf, int sg, float4 (&fm)[MB], float4 (&fn)[2]) {
    const float* as = smem + buf * BUF;
    const float* bs = as + A_SZ + G_SZ;
    const int ch = ((sg * 2 + lhi) ^ swz) << 2;
#pragma unroll
    for (int j = 0; j < MB; ++j) {  // m side: (t3 .* d2)[m = 32j + l31][k = 8sg + 4lhi + t]
      const int o = (j * 32 + l31) * BK + ch;
      const f32x4 av = *(const f32x4*)(as + o), gv = *(const f32x4*)(as + A_SZ + o);
      const f32x4 u = av * gv;
      fm[j] = make_float4(u.x, u.y, u.z, u.w);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {  // n side: W2T[k][n = nw + 32i + l31]
      const float* q = bs + (sg * 8 + lhi * 4) * TN + nw + i * 32 + l31;
      fn[i] = make_float4(q[0], q[TN], q[2 * TN], q[3 * TN]);
    }
  };
  auto mfmas = [&](const float4 (&fm)[MB], const float4 (&fn)[2]) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < MB; ++j) {
          const float a = t == 0 ? fn[i].x : t == 1 ? fn[i].y : t == 2 ? fn[i].z : fn[i].w;
          const float b = t == 0 ? fm[j].x : t == 1 ? fm[j].y : t == 2 ? fm[j].z : fm[j].w;
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0);  // D[n][m]
        }
  };

  float4 fm0[MB], fn0[2], fm1[MB], fn1[2];
  if constexpr (BF16) {
    auto pack = [](const float4& lo, const float4& hi) {
      return indm_pack_bf16x8(lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w);
    };
    stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int s = 0; s < nk; ++s) {
      const int buf = s & 1;
      if (s + 1 < nk) stage(buf ^ 1);
      frags(buf, 0, fm0, fn0);
      frags(buf, 1, fm1, fn1);
      bf16x8 hm[MB], hn[2];
#pragma unroll
      for (int j = 0; j < MB; ++j) hm[j] = pack(fm0[j], fm1[j]);
#pragma unroll
      for (int i = 0; i < 2; ++i) hn[i] = pack(fn0[i], fn1[i]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < MB; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hn[i], hm[j], acc[i][j], 0, 0, 0);  // D[n][m]
      __builtin_amdgcn_sched_barrier(0);  // (MFMAs stay above the barrier: no accumulator rotation, see igemm.hip)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  } else {
  stage(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (nk > 1) stage(1);
  frags(0, 0, fm0, fn0);
  frags(0, 1, fm1, fn1);
  mfmas(fm0, fn0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int s = 1; s < nk; ++s) {
    const int buf = s & 1;
    if (s + 1 < nk) stage(buf ^ 1);
    frags(buf, 0, fm0, fn0);
    mfmas(fm1, fn1);
    frags(buf, 1, fm1, fn1);
    mfmas(fm0, fn0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  mfmas(fm1, fn1);
  }

  chain_epilogue<JB, BF16, T2OUT, TMV>(p, acc, smem, tid, wave, m0, nw, l31, lhi);
}

// ---------------------------------------------------------------------------------------------------------------
// bf16 compute with PRE-PACKED weights (round 2).  With bf16 MFMAs (8x fewer matrix cycles) the kernel above is no
// longer MFMA-bound and not yet HBM-bound either: a 16-deep slab holds ~0.3 us of arithmetic between a copy wait and a
// workgroup barrier (PMC, profiles/r02_pmc_chain_bf16.json: MFMA pipe 25 % busy, LDS index unit 29 %, 45 % of wave time
// parked at waitcnt / barrier, 805 MB in 240-300 us), every wave fetches its W2T fragments as eight ds_read_b32 and
// the 32 KB fp32 image of the W2T slab is re-staged for every slab.  Here the host packs W2T once per flow call into
// bf16 MFMA-fragment order (`Bp[slab][32-column block][lane][8]`, 512 KB, L2-resident): its slab image shrinks to
// 16 KB (copied by the waves that use it, one ds_read_b128 per fragment: a third fewer LDS and VALU instructions), a
// buffer to 24 KB, three buffers rotate (two slabs in flight per workgroup, two workgroups per CU), the loop waits with
// vmcnt(3) for the older slab only and synchronises with a bare s_barrier (no fence: __syncthreads() would drain the
// copies).  231 us / 82 us at the two scales (302 / 92 before; 204 vs 240 us under the profiler).  Measured
// alternatives: six buffers with one 144 KB workgroup per CU 254 us (the epilogue no longer overlaps another
// workgroup's main loop); W2T fragments fetched straight into registers 224 us but vmcnt's in-order retirement then
// ties the copies to the register loads.  Still latency-bound per slab (MFMA 29 %, LDS 23 %, 42 % parked): the next
// step is more arithmetic per barrier -- 32- or 64-deep slabs, which need bf16 A / gate images to fit two workgroups.
// Same rounding points (RNE to bf16 of W, of the gated product, of the stage-2 operands) and the same epilogue as
// ires_chain_kernel<., true, .>: bit-identical results.
// k order inside a lane's 8-element fragment (both operands): e -> 8 (e >> 2) + 4 lhi + (e & 3).
struct ChainArgsPW {
  ChainArgs c;
  const void* Bp;  // packed bf16 W2T: [K / 16][512 / 32][64 lanes][8]
};
constexpr int PW_NBUF = 3, PW_D = PW_NBUF - 1, PW_B = 16 * 64 * 4, PW_BUF = A_SZ + G_SZ + PW_B;  // floats

template <int C>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C) : "memory"); }

template <int JB, bool T2OUT>
__global__ __launch_bounds__(NW * 64, 4) void ires_chain_pw_kernel(ChainArgsPW q) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const ChainArgs& p = q.c;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * TM;
  const int nk = p.K / BK;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int swz = (l31 >> 2) & 3;
  const int nw = wave * 64;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // three copies per wave and slab: 16 rows of A (waves 0-3) or of the gate (waves 4-7), and the wave's own two
  // 32-column blocks of the packed W2T slab (1 KB each: lane l of the copy lands at byte 16 l, fragment order)
  const int arow = (wave & 3) * 16 + (lane >> 2);
  const int achunk = (lane & 3) ^ ((arow >> 2) & 3);
  const char* pa = (const char*)((wave < 4 ? p.A : p.G) + (long)m0 * p.K);        // wave-uniform running bases (SGPRs) ...
  const unsigned a_off = ((unsigned)arow * (unsigned)p.K + 4u * achunk) * 4u;      // ... + fixed per-lane byte offsets
  const int a_dst = (wave < 4 ? 0 : A_SZ) + (wave & 3) * 16 * BK;
  const char* pb = (const char*)((const float*)q.Bp + (long)(wave * 2) * 64 * 4);
  const unsigned b_off = 16u * lane;
  const int b_dst = A_SZ + G_SZ + wave * 2 * 256;
  auto stage = [&](int buf) {
    float* base = smem + buf * PW_BUF;
    dma_s<0>(lds_addr(base + a_dst), a_off, pa);
    dma_s_x2(lds_addr(base + b_dst), b_off, pb);   // the second block: +1 KB in global memory and in LDS
    pa += BK * sizeof(float);
    pb += 16 * 256 * sizeof(float);
  };

  const int pre = nk < PW_D ? nk : PW_D;
  for (int d = 0; d < pre; ++d) stage(d);
  if (nk >= PW_D) wait_vm<3 * (PW_D - 1)>();   // slab 0 has landed, the rest of the prologue is on its way
  else wait_vm<0>();
  __builtin_amdgcn_s_barrier();
  int buf = 0, sbuf = PW_D % PW_NBUF;   // buffer of slab s / of slab s + PW_D
  for (int s = 0; s < nk; ++s) {
    if (s + PW_D < nk) stage(sbuf);     // its previous contents (slab s - 1) were read before the last barrier
    const float* as = smem + buf * PW_BUF;
    bf16x8 hm[2], hn[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) hn[i] = *(const bf16x8*)(as + b_dst + i * 256 + lane * 4);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int o0 = (j * 32 + l31) * BK + ((lhi ^ swz) << 2), o1 = (j * 32 + l31) * BK + (((2 + lhi) ^ swz) << 2);
      const f32x4 u = *(const f32x4*)(as + o0) * *(const f32x4*)(as + A_SZ + o0);
      const f32x4 w = *(const f32x4*)(as + o1) * *(const f32x4*)(as + A_SZ + o1);
      hm[j] = indm_pack_bf16x8(u.x, u.y, u.z, u.w, w.x, w.y, w.z, w.w);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hn[i], hm[j], acc[i][j], 0, 0, 0);  // D[n][m]
    __builtin_amdgcn_sched_barrier(0);
    // slab s + 1 must have landed; the younger ones (PW_D - 1 at most) stay in flight: 3 copies per slab
    const int younger = min(PW_D - 1, nk - 2 - s);
    if (younger >= 4) wait_vm<12>();
    else if (younger == 3) wait_vm<9>();
    else if (younger == 2) wait_vm<6>();
    else if (younger == 1) wait_vm<3>();
    else wait_vm<0>();   // (PW_D - 1 bounds `younger`: the deeper cases fold away for a short rotation)
    __builtin_amdgcn_s_barrier();
    buf = buf + 1 == PW_NBUF ? 0 : buf + 1;
    sbuf = sbuf + 1 == PW_NBUF ? 0 : sbuf + 1;
  }
  chain_epilogue<JB, true, T2OUT>(p, acc, smem, tid, wave, m0, nw, l31, lhi);
}

template <int JB, bool T2OUT>
int launch_chain_pw_t(const ChainArgsPW& q, hipStream_t st) {
  constexpr size_t bytes = PW_NBUF * PW_BUF * sizeof(float);  // 72 KB (the 64 KB stage-2 reduction buffer overlays it)
  if (indm_lds_optin<ires_chain_pw_kernel<JB, T2OUT>>(bytes) != INDM_OK) return INDM_E_LAUNCH;
  hipLaunchKernelGGL((ires_chain_pw_kernel<JB, T2OUT>), dim3(q.c.M / TM), dim3(NW * 64), bytes, st, q);
  return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
}

template <int JB, bool BF16, bool T2OUT, int TMV>
int launch_chain_tm(const ChainArgs& p, hipStream_t st) {
  constexpr size_t bytes = 2 * BUF * sizeof(float);
  if (indm_lds_optin<ires_chain_kernel<JB, BF16, T2OUT, TMV>>(bytes) != INDM_OK) return INDM_E_LAUNCH;
  hipLaunchKernelGGL((ires_chain_kernel<JB, BF16, T2OUT, TMV>), dim3(p.M / TMV), dim3(NW * 64), bytes, st, p);
  return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
}
template <int JB, bool BF16, bool T2OUT>
int launch_chain_t(const ChainArgs& p, hipStream_t st) {
  // fewer than two 64-row workgroups per CU (M < 32768 on this part): 32-row workgroups, twice as many
  if (p.M / TM < 512) return launch_chain_tm<JB, BF16, T2OUT, 32>(p, st);
  return launch_chain_tm<JB, BF16, T2OUT, 64>(p, st);
}
template <int JB, bool BF16>
int launch_chain(const ChainArgs& p, hipStream_t st) {
  return p.T2 ? launch_chain_t<JB, BF16, true>(p, st) : launch_chain_t<JB, BF16, false>(p, st);
}

}  // namespace

// P[M, J] = ((A .* G) W2T .* D1) W1p[:, :J]  (and, if T2 != NULL, T2[M, 512] = (A .* G) W2T)  with A, G: [M, K]; W2T: [K, 512]; D1: [M, 512]; W1p: [512, 32 if J <= 32 else 128]
// (zero-padded columns).  Requirements: M % 64 == 0, K % 16 == 0, K >= 32, J <= 128, 16-byte aligned pointers.
// Returns 1 when the geometry is not covered (the caller then runs the two contractions separately).
static int ires_chain_impl(bool bf16, const float* A, const float* G, const float* W2T, const float* D1, const float* W1p,
                           float* P, float* T2, int M, int K, int N, int J, void* stream) {
  if (M < 0 || K <= 0 || N <= 0 || J <= 0) return INDM_E_SHAPE;
  if (M == 0) return INDM_OK;
  if (N != TN || (M % TM) || (K % BK) || K < 2 * BK || J > 128) return 1;
  if (((uintptr_t)A | (uintptr_t)G | (uintptr_t)W2T | (uintptr_t)D1 | (uintptr_t)W1p | (uintptr_t)T2) & 15) return 1;
  ChainArgs p = {A, G, W2T, D1, W1p, P, M, K, J, T2};
  hipStream_t st = (hipStream_t)stream;
  // two instantiations: J <= 32 (the 3-channel scale: 27) and J <= 128 (the 12- / 48-channel scales: 108); W1p is
  // padded to 32 / 128 columns accordingly
  if (bf16) return J <= 32 ? launch_chain<1, true>(p, st) : launch_chain<4, true>(p, st);
  return J <= 32 ? launch_chain<1, false>(p, st) : launch_chain<4, false>(p, st);
}
INDM_API int indm_ires_chain_f32(const float* A, const float* G, const float* W2T, const float* D1, const float* W1p,
                                 float* P, float* T2, int M, int K, int N, int J, void* stream) {
  return ires_chain_impl(false, A, G, W2T, D1, W1p, P, T2, M, K, N, J, stream);
}
// same contraction with the MFMA inputs rounded to bf16 (fp32 tensors, fp32 accumulate)
INDM_API int indm_ires_chain_bf16(const float* A, const float* G, const float* W2T, const float* D1, const float* W1p,
                                  float* P, float* T2, int M, int K, int N, int J, void* stream) {
  return ires_chain_impl(true, A, G, W2T, D1, W1p, P, T2, M, K, N, J, stream);
}

// bf16 compute with W2T pre-packed by the caller (indm_pack_w2t_bf16 layout): see ires_chain_pw_kernel
INDM_API int indm_ires_chain_bf16w(const float* A, const float* G, const void* W2Tp, const float* D1, const float* W1p,
                                   float* P, float* T2, int M, int K, int N, int J, void* stream) {
  if (M < 0 || K <= 0 || N <= 0 || J <= 0) return INDM_E_SHAPE;
  if (M == 0) return INDM_OK;
  if (N != TN || (M % TM) || (K % BK) || K < 2 * BK || J > 128) return 1;
  if (((uintptr_t)A | (uintptr_t)G | (uintptr_t)W2Tp | (uintptr_t)D1 | (uintptr_t)W1p | (uintptr_t)T2) & 15) return 1;
  ChainArgsPW q = {{A, G, nullptr, D1, W1p, P, M, K, J, T2}, W2Tp};
  hipStream_t st = (hipStream_t)stream;
  if (J <= 32) return T2 ? launch_chain_pw_t<1, true>(q, st) : launch_chain_pw_t<1, false>(q, st);
  return T2 ? launch_chain_pw_t<4, true>(q, st) : launch_chain_pw_t<4, false>(q, st);
}

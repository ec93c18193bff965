// bf16 STORAGE of the residual flow's 512-wide tensors (BASELINE configs #3-#5: "bf16 compute / fp32 accum").
//
// Round 2/3's bf16 mode rounded the MFMA inputs but kept every [pixels, 512] tensor of an iResBlock (a1, d1, a2, d2, t3,
// t2, ...) as fp32 in HBM: its kernels turned into streamers of fp32 data at 3.5 - 5 TB/s.  Here those
// tensors live in HBM as bf16 in MFMA-FRAGMENT order ("F16" below), the layout a v_mfma_f32_32x32x16_bf16 produces and
// consumes:
//
//   X[M, N] (M % 32 == 0, N % 32 == 0) is cut into 32 x 32 blocks (mt = m / 32, nb = n / 32); a block is two 1 KB chunks
//   s = 0, 1; chunk (mt, nb, s) holds, for lane = 32 lhi + (m % 32) and e < 8, the element
//       X[m][32 nb + 16 s + 8 (e >> 2) + 4 lhi + (e & 3)]
//   at bf16 index (((mt * N/32 + nb) * 2 + s) * 64 + lane) * 8 + e.
//
//   * it IS the accumulator layout of a 32x32 MFMA whose B operand carried the rows m (lane (m, lhi) holds the 16
//     columns (r & 3) + 8 (r >> 2) + 4 lhi, r = 8 s + e): a producer stores 16 bytes per lane, 1 KB contiguous per wave
//     instruction, no LDS staging and no partial cache lines;
//   * it IS an MFMA input fragment (8 k-values per lane for row m) for a contraction over n: a consumer copies chunks
//     global -> LDS with lane-linear direct-to-LDS copies and reads them back with one conflict-free ds_read_b128 per
//     fragment -- no swizzle, no convert, no pack in the loop.  The weights are packed once per flow call into the same k
//     order (hip.pack_w2t_bf16: Wp[ks][nblk][lane][e] = W[16 ks + 8 (e >> 2) + 4 lhi + (e & 3)][32 nblk + l31]);
//   * elementwise kernels do not care (all [M, 512] tensors of a block share the layout), reductions over rows sum over
//     lanes.
//
// Kernels (all gfx950, hand-written):
//   f16_thin_conv_kernel   3x3 convolution of a skinny (C <= 12 channel) fp32 NHWC tensor -> F16 [pixels, 512]:
//                          im2col gather + bf16 MFMA (K = 9C padded to 16s), epilogues: plain | Sin activation (writes a
//                          and cos, never the pre-activation) | gate (u = out .* G, optionally the raw output too)
//   f16_rows_kernel        a workgroup owns 64 rows x all 512 columns: the A tile (64 KB, ONE contiguous run of F16 memory)
//                          is copied to LDS once, the packed weights stream from L2 straight into registers, no barrier
//                          in the main loop.  Modes: GEMM (-> F16), GEMM + Sin activation (-> a, cos), CHAIN (the
//                          back-to-back VJP link of ires_chain.hip: x W2T, [store t2], .* d1, x W1 panel -> [M, 9C] fp32),
//                          THIN_N ([M,512] x [512, 9C] panel -> fp32)
//   converters             fp32 row-major <-> F16 (tests, the boundary of kernels that have no F16 form yet)
//
// Reference: flows/resflow/layers/iresblock.py:60-174,253-273 (what is computed); DESIGN.md section 3.1.2 (why).
#include "indm_common.h"
#include <stdlib.h>
#pragma clang diagnostic ignored "-Winline-asm"   // M0 clobber declared on the copy helpers (see igemm.hip)

#pragma GCC visibility push(default)
#include "../../include/indm_hip.h"
#pragma GCC visibility pop

namespace {

typedef unsigned short u16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr float TWO_PI_F = 6.283185307179586f, INV_TWO_PI_F = 0.15915494309189535f;

__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
// four direct-to-LDS copies (64 lanes x 16 bytes each) 1 KB apart in global memory and in LDS: one M0 value, one address
// register, four immediates (the immediate is added to both addresses)
__device__ __forceinline__ void dma_x4(unsigned lds_byte_addr, unsigned voff, const void* sbase) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, %2 offset:0\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
               "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072"
               :: "s"(lds_byte_addr), "v"(voff), "s"(sbase) : "memory", "m0");
}

__device__ __forceinline__ u32x4 pack8(const float* v) {   // 8 floats -> 8 bf16 (RNE), element e at bits 16 (e & 1) of word e >> 1
  return __builtin_bit_cast(u32x4, indm_pack_bf16x8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]));
}
__device__ __forceinline__ void unpack8(const u32x4& w, float* v) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    v[2 * q] = __uint_as_float(w[q] << 16);
    v[2 * q + 1] = __uint_as_float(w[q] & 0xffff0000u);
  }
}
// a = sin(2 pi p) / (2 pi), d = cos(2 pi p) on the hardware's revolution-based sine / cosine (v_sin_f32 / v_cos_f32 take
// their argument in revolutions): the results are rounded to bf16 (2^-9 relative) right after, far above the
// instructions' error.  layers/base/activations.py:7-12.
__device__ __forceinline__ void sin_act(float p, float& a, float& d) {
  const float f = __builtin_amdgcn_fractf(p);
  a = __builtin_amdgcn_sinf(f) * INV_TWO_PI_F;
  d = __builtin_amdgcn_cosf(f);
}

// ---------------------------------------------------------------------------------------------------------------
// converters
__global__ __launch_bounds__(256) void f16_pack_kernel(const float* __restrict__ x, u16* __restrict__ y, long M, int N) {
  // one thread per 16-byte chunk element group: (mt, nb, s, lane)
  const long groups = M * N / 8;
  for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < groups; g += (long)gridDim.x * 256) {
    const int lane = (int)(g & 63);
    const long c = g >> 6;
    const int s = (int)(c & 1);
    const long blk = c >> 1;
    const int nbn = N >> 5;
    const long mt = blk / nbn;
    const int nb = (int)(blk - mt * nbn);
    const int l31 = lane & 31, lhi = lane >> 5;
    const float* row = x + (mt * 32 + l31) * N + nb * 32 + 16 * s + 4 * lhi;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = row[8 * (e >> 2) + (e & 3)];
    *(u32x4*)(y + g * 8) = pack8(v);
  }
}
__global__ __launch_bounds__(256) void f16_unpack_kernel(const u16* __restrict__ x, float* __restrict__ y, long M, int N) {
  const long groups = M * N / 8;
  for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < groups; g += (long)gridDim.x * 256) {
    const int lane = (int)(g & 63);
    const long c = g >> 6;
    const int s = (int)(c & 1);
    const long blk = c >> 1;
    const int nbn = N >> 5;
    const long mt = blk / nbn;
    const int nb = (int)(blk - mt * nbn);
    const int l31 = lane & 31, lhi = lane >> 5;
    float v[8];
    unpack8(*(const u32x4*)(x + g * 8), v);
    float* row = y + (mt * 32 + l31) * N + nb * 32 + 16 * s + 4 * lhi;
    *(float4*)(row) = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)(row + 8) = make_float4(v[4], v[5], v[6], v[7]);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// f16_rows_kernel
enum { R_GEMM = 0, R_GEMM_ACT = 1, R_CHAIN = 2, R_THINN = 3, R_GEMM_GATE = 4, R_GEMM_ACT_THINN = 5 };
constexpr int R_TM = 64, R_N = 512, R_NW = 8;
constexpr int R_LDS_BYTES = R_TM * R_N * 2;   // the A tile; the stage-2 reduction buffer [8][64][32] fp32 overlays it

struct RowsArgs {
  const u16* A;      // F16 [M, 512]
  const u16* Bp;     // packed [32][16][64][8] (stage 1)
  const u16* D1;     // F16 [M, 512]: gate of the chain epilogue / of GEMM_GATE
  const u16* Pp;     // packed panel [32][JB][64][8] (stage 2 / THIN_N)
  float* P;          // [M, J] fp32 row-major
  u16* O1;           // GEMM: C; GEMM_ACT: a; CHAIN: t2 (optional); GEMM_GATE: C .* D1
  u16* O2;           // GEMM_ACT: cos; GEMM_GATE: C (optional)
  const float* bias;   // [512] (GEMM_ACT / GEMM, optional)
  const float* bias2;  // [M / rpg][512] (optional)
  int M, J, rpg;
};

// One tile of the 64-row kernel family: everything after the A tile has landed in `smem` (the caller has synchronised).
// D = k-steps of weight fragments in flight (L2 latency / MFMA time of a k-step round).  Barriers executed: GEMM modes 0,
// CHAIN / THIN_N 2 JB.
template <int MODE, int JB, bool T2OUT, int TMV, int D>
__device__ __forceinline__ void rows_tile(const RowsArgs& p, char* smem, const long tile, const int tid, const int lane,
                                          const int wave, const int l31, const int lhi) {
  constexpr int MB = TMV / 32;   // 32-row blocks per workgroup
  const long m0 = tile * TMV;
  f32x16 acc[2][MB];
  bf16x8 hm2[2][MB][2];   // stage-2 operand fragments [i][j][s] (CHAIN: gated accumulators; THIN_N: A fragments)
  if constexpr (MODE != R_THINN) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < MB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // weights of this wave's two 32-column blocks, k-step ks: Bp[(ks * 16 + 2 wave + i) * 64 + lane] (16 bytes)
    const u32x4* bp = (const u32x4*)p.Bp + (long)(2 * wave) * 64 + lane;
    u32x4 q[D][2];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
      for (int i = 0; i < 2; ++i) q[d][i] = bp[(long)d * 1024 + i * 64];
    // The 32 k-steps are unrolled in full (ring slot ks % D, every condition a compile-time constant):
    //  * the MFMAs read q[d] in place and the fetch of k-step ks + D is issued AFTER them into the same registers.  Round 4's
    //    first form (a rolled loop that copied q[d] to a temporary and fetched before the MFMAs) made hipcc rotate the ring
    //    with v_mov_b64 of just-loaded registers behind `s_waitcnt vmcnt(0)`: every k-step waited for the loads it had just
    //    issued (PMC: MFMA pipe 36 - 44 % busy, half of the wave time in s_waitcnt);
    //  * a scheduling barrier closes every k-step: without it hipcc's scheduler sinks the fetches down to their use D steps
    //    later (the ISA then waits with vmcnt(0 / 1) again); with it the waits are vmcnt(2 D - 2 .. 2 D - 1).
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) {
      const int d = ks % D;
      bf16x8 hm[MB];
#pragma unroll
      for (int j = 0; j < MB; ++j) hm[j] = *(const bf16x8*)(smem + (j * 32 + ks) * 1024 + lane * 16);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < MB; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, q[d][i]), hm[j], acc[i][j], 0, 0, 0);  // D[n][m]
      if (ks + D < 32) {
#pragma unroll
        for (int i = 0; i < 2; ++i) q[d][i] = bp[(long)(ks + D) * 1024 + i * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < MB; ++j)
#pragma unroll
        for (int s = 0; s < 2; ++s)
          hm2[i][j][s] = *(const bf16x8*)(smem + (j * 32 + 2 * (2 * wave + i) + s) * 1024 + lane * 16);
  }

  // chunk (m-block j, column block 2 wave + i, half s) of an F16 [M, 512] tensor, this lane's 16 bytes
  auto chunk = [&](int i, int j, int s) -> long {
    return ((((tile * MB + j) * 16 + 2 * wave + i) * 2 + s) * 64 + lane) * 8;   // bf16 index
  };

  if constexpr (MODE == R_GEMM || MODE == R_GEMM_ACT || MODE == R_GEMM_GATE || MODE == R_GEMM_ACT_THINN) {
    // acc[i][j][r] = C[m = m0 + 32 j + l31][n = 64 wave + 32 i + (r & 3) + 8 (r >> 2) + 4 lhi]
    const float* b2 = p.bias2 ? p.bias2 + (m0 / p.rpg) * R_N : nullptr;   // rpg % 64 == 0: one image per tile
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float bv[16];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int n = 64 * wave + 32 * i + 8 * a + 4 * lhi;
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias) b = *(const float4*)(p.bias + n);
        if (b2) {
          const float4 c = *(const float4*)(b2 + n);
          b.x += c.x; b.y += c.y; b.z += c.z; b.w += c.w;
        }
        bv[4 * a] = b.x; bv[4 * a + 1] = b.y; bv[4 * a + 2] = b.z; bv[4 * a + 3] = b.w;
      }
#pragma unroll
      for (int j = 0; j < MB; ++j)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          float v[8], w[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = acc[i][j][8 * s + e] + bv[8 * s + e];
          if constexpr (MODE == R_GEMM_ACT || MODE == R_GEMM_ACT_THINN) {
#pragma unroll
            for (int e = 0; e < 8; ++e) sin_act(v[e], v[e], w[e]);
            *(u32x4*)(p.O2 + chunk(i, j, s)) = pack8(w);
          }
          if constexpr (MODE == R_GEMM_GATE) {
            if (p.O2) *(u32x4*)(p.O2 + chunk(i, j, s)) = pack8(v);
            unpack8(*(const u32x4*)(p.D1 + chunk(i, j, s)), w);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= w[e];
          }
          if constexpr (MODE == R_GEMM_ACT_THINN) {
            // (round 6) the activation a, rounded to bf16, IS the operand fragment of the thin-N product that follows it in the
            // g-net (a2 W3 before the tap gather): stage 2 below runs on it -- the separate thin-N launch re-read a2 from HBM
            const u32x4 pk = pack8(v);
            hm2[i][j][s] = __builtin_bit_cast(bf16x8, pk);
            *(u32x4*)(p.O1 + chunk(i, j, s)) = pk;
          } else {
            *(u32x4*)(p.O1 + chunk(i, j, s)) = pack8(v);
          }
        }
    }
    if constexpr (MODE != R_GEMM_ACT_THINN) return;
    else __syncthreads();   // every wave is done with the A tile: the reduction buffer overlays it
  }

  if constexpr (MODE == R_CHAIN) {
    // [store t2]; gate by d1; the gated accumulators, rounded to bf16, ARE the next contraction's operand fragments
    // (lane (m, lhi) holds 8 k = n values per half s: the F16 fragment again)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < MB; ++j)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          float v[8], g[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = acc[i][j][8 * s + e];
          if constexpr (T2OUT) *(u32x4*)(p.O1 + chunk(i, j, s)) = pack8(v);
          unpack8(*(const u32x4*)(p.D1 + chunk(i, j, s)), g);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= g[e];
          hm2[i][j][s] = __builtin_bit_cast(bf16x8, pack8(v));
          if (j == MB - 1 && s == 1) __builtin_amdgcn_sched_barrier(0);   // one column block's gate chunks in flight at a time (registers)
        }
    __syncthreads();   // every wave is done with the A tile: the reduction buffer overlays it
  }
  if constexpr (MODE == R_THINN) __syncthreads();

  // ---- stage 2: out[m][jp] = sum over this wave's 64 k of u[m][k] Wp[k][jp]; 8-way reduction over the waves through LDS
  float* red = (float*)smem;   // [8 waves][TMV m][32 jp]
  const u32x4* pp = (const u32x4*)p.Pp + lane;
#pragma unroll 1
  for (int jb = 0; jb < JB; ++jb) {
    f32x16 out[MB];
#pragma unroll
    for (int j = 0; j < MB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) out[j][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 hp = __builtin_bit_cast(bf16x8, pp[((long)(2 * (2 * wave + i) + s) * JB + jb) * 64]);
#pragma unroll
        for (int j = 0; j < MB; ++j)
          out[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hm2[i][j][s], hp, out[j], 0, 0, 0);   // D[m][jp]
      }
    if (jb > 0) __syncthreads();   // the previous round's reduction has read `red`
#pragma unroll
    for (int j = 0; j < MB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = j * 32 + 8 * (r >> 2) + (r & 3) + 4 * lhi;
        red[(wave * TMV + m) * 32 + l31] = out[j][r];
      }
    __syncthreads();
    if (tid < TMV * 8) {  // TMV x 32 outputs, 4 per thread: thread -> (m = tid / 8, jp = (tid % 8) * 4 .. +3)
      const int m = tid >> 3, j0 = (tid & 7) * 4;
      float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int wv = 0; wv < R_NW; ++wv) {
        const float4 v = *(const float4*)(red + (wv * TMV + m) * 32 + j0);
        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
      }
      float* dst = p.P + (m0 + m) * p.J + jb * 32 + j0;
      const int left = p.J - (jb * 32 + j0);
      if (left > 0) dst[0] = sum.x;
      if (left > 1) dst[1] = sum.y;
      if (left > 2) dst[2] = sum.z;
      if (left > 3) dst[3] = sum.w;
    }
  }
}

// TMV = 64 rows per workgroup, or 32 when M is too small to give every CU its two 64-row workgroups (per-rank shards)
template <int MODE, int JB, bool T2OUT, int TMV>
__global__ __launch_bounds__(R_NW * 64, 4) void f16_rows_kernel(RowsArgs p) {
  constexpr int MB = TMV / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const long tile = blockIdx.x;
  // ---- the A tile: TMV rows x 512 columns of F16 memory = 32 MB consecutive 1 KB chunks; chunk c = 32 j + ks is fragment
  // (m-block j, k-step ks).  4 MB chunks per wave.
  {
    const char* src = (const char*)p.A + tile * (long)(TMV * R_N * 2) + (long)wave * (MB * 4096);
    const unsigned voff = 16u * lane;
#pragma unroll
    for (int g = 0; g < MB; ++g) dma_x4(lds_addr(smem + wave * (MB * 4096) + g * 4096), voff, src + g * 4096);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's copies have landed ...
  __syncthreads();                                    // ... and everybody else's
  rows_tile<MODE, JB, T2OUT, TMV, 4>(p, smem, tile, tid, lane, wave, l31, lhi);
}

template <int MODE, int JB, bool T2OUT, int TMV>
int launch_rows_tm(const RowsArgs& a, hipStream_t st) {
  constexpr int bytes = TMV * R_N * 2;
  if (indm_lds_optin<f16_rows_kernel<MODE, JB, T2OUT, TMV>>(bytes) != INDM_OK) return INDM_E_LAUNCH;
  hipLaunchKernelGGL((f16_rows_kernel<MODE, JB, T2OUT, TMV>), dim3(a.M / TMV), dim3(R_NW * 64), bytes, st, a);
  return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
}
template <int MODE, int JB, bool T2OUT>
int launch_rows(const RowsArgs& a, hipStream_t st) {
  // fewer than two 64-row workgroups per CU (M < 32768): 32-row workgroups, twice as many.  The thin-N mode has no main
  // loop to amortise a tile's load latency over: always 32-row tiles (32 KB of LDS: four workgroups per CU overlap).
  // Measured and dropped (round 4, M = 131072): 32-row tiles for the other modes (chain 104 -> 131 us: twice the weight
  // traffic out of L2), and a persistent form -- one workgroup per CU, two 64 KB tile buffers, a ninth wave that only
  // copies the next tile -- 112 us with four k-steps of weight fragments in flight, 140 us with eight (spills at the
  // 168-register cap of a nine-wave workgroup): with two waves per SIMD the weight-fragment latency is exposed; the
  // two-workgroups-per-CU form hides it behind four waves per SIMD.  (Round 5: SIX k-steps of weight fragments in flight in THIS
  // form -- 128 VGPRs, no spills -- measured the same: GEMM 90.4 / 91.5 us, chain 103.7 / 103.8 us.  Starting a tile when the FIRST
  // half of its A image has landed does not survive the compiler: its own `s_waitcnt vmcnt(6 / 7)` in front of every k-step count
  // only the fragment loads it knows, vmcnt retires in order, so the first counted wait drains the older asm-issued copies too.)  Starting the second workgroup of every CU 6 - 18 us late
  // (so that one streams while the other multiplies) only added the delay: 101 -> 105 / 114 / 126 us.  What the numbers
  // point at: every workgroup pulls the whole 512 KB of packed weights out of L2 per tile -- 1 GB per launch at 64-row tiles
  // (10 TB/s of L2 -> CU traffic next to 2.6 TB/s of HBM traffic), 2 GB at 32-row tiles (15.6 TB/s, and slower).
  if (a.M / R_TM < 512 || MODE == R_THINN) return launch_rows_tm<MODE, JB, T2OUT, 32>(a, st);
  return launch_rows_tm<MODE, JB, T2OUT, 64>(a, st);
}

// ---------------------------------------------------------------------------------------------------------------
// f16_thin_conv_kernel: out[pixel, co] = sum_{tap, c} x[pixel shifted by tap, c] * w[(tap, c), co], co < 512, C <= 12.
// 128 pixels x 128 columns per workgroup, 4 waves (wave w: pixels 32 w .. 32 w + 31, four 32-column blocks); the column
// tile's weights are packed into LDS by the workgroup.  K = 9 C is padded to S = ceil(K / 16)
// k-steps; the operand fragments use consecutive k per lane (k = 16 s + 8 lhi + e) on both sides.  A lane GATHERS the
// fragment it multiplies (lane l31 = its pixel, its 8 k per step): the im2col values never pass through LDS.
enum { TC_PLAIN = 0, TC_ACT = 1, TC_GATE = 2 };
constexpr int TC_TM = 128, TC_TN = 128;

struct ThinConvArgs {
  const float* X;     // [N, H, W, C] fp32
  const float* Wt;    // [9 C, Cout] fp32 (rows (tap, c))
  const float* bias;  // [Cout] optional
  const u16* G;       // TC_GATE: F16 gate [pixels, Cout]
  u16* O1;            // PLAIN: out; ACT: a; GATE: out .* G
  u16* O2;            // ACT: cos; GATE: raw out (optional)
  int N, H, W, C, Cout, sign;
};

// SMAX: upper bound of the k-steps (loop unrolling); VEC4: C % 4 == 0, four consecutive k share a tap (float4 loads).
// One workgroup per (128-pixel tile, 128-column tile), column tiles fastest (the four workgroups that gather the same
// pixels are neighbours: the image patch comes out of L2); three workgroups per CU hide each other's phases.  The gate
// chunks of the epilogue (TC_GATE) are requested FIRST, before the weights are packed and the pixels gathered.
template <int EPI, int SMAX, bool VEC4>
__global__ __launch_bounds__(256, 3) void f16_thin_conv_kernel(ThinConvArgs p) {
  __shared__ __attribute__((aligned(16))) u16 wf[SMAX * 4 * 64 * 8];   // weights [s][nblk][lane][8]
  __shared__ float bsh[TC_TN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int C = p.C, K = 9 * C, S = (K + 15) >> 4;
  const int tiles_n = p.Cout / TC_TN;
  const int tn = blockIdx.x % tiles_n, tm = blockIdx.x / tiles_n;
  const int n0 = tn * TC_TN;
  const int HW = p.H * p.W;
  const long mt = (long)tm * 4 + wave;
  const int nbn = p.Cout >> 5;
  auto chunk = [&](int jn, int s) -> long { return (((mt * nbn + (n0 >> 5) + jn) * 2 + s) * 64 + lane) * 8; };

  u32x4 gq[4][2];
  constexpr bool GQ_LATE = SMAX <= 2;   // (the 12-channel form holds 14 float4 gathers: with 8 more chunks in flight it spills)
  if constexpr (EPI == TC_GATE && !GQ_LATE) {
#pragma unroll
    for (int jn = 0; jn < 4; ++jn)
#pragma unroll
      for (int s = 0; s < 2; ++s) gq[jn][s] = *(const u32x4*)(p.G + chunk(jn, s));
  }
  // ---- weights of this column tile -> fragment order (A role: lane l31 = output column, k = 16 s + 8 lhi + e)
  for (int u = tid; u < S * 4 * 64; u += 256) {
    const int ln = u & 63, nb = (u >> 6) & 3, s = u >> 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 16 * s + 8 * (ln >> 5) + e;
      // (unconditional load from a clamped row + select: a load under a condition costs a branch and, at the merge, an
      // `s_waitcnt vmcnt(0)` -- hipcc serialised all of this kernel's gathers that way, one memory latency each)
      const float t = p.Wt[(long)min(k, K - 1) * p.Cout + n0 + nb * 32 + (ln & 31)];
      v[e] = k < K ? t : 0.f;
    }
    *(u32x4*)(wf + u * 8) = pack8(v);
  }
  if (tid < TC_TN) bsh[tid] = p.bias ? p.bias[n0 + tid] : 0.f;

  // ---- this lane's pixel and its im2col fragments (B role: lane l31 = pixel, k = 16 s + 8 lhi + e), gathered k-step by
  // k-step straight into the MFMA operand registers
  const long pix = (long)tm * TC_TM + wave * 32 + l31;
  const int img = (int)(pix / HW), r = (int)(pix - (long)img * HW);
  const int h = r / p.W, w = r - h * p.W;
  const float* base = p.X + pix * C;
  bf16x8 hx[SMAX];
#pragma unroll
  for (int s = 0; s < SMAX; ++s) {
    float gv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) gv[e] = 0.f;
    {   // (no `if (s < S)`: k >= K already selects zeros, and without the branch all SMAX gathers are in flight together)
      if constexpr (VEC4) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int k = 16 * s + 8 * lhi + 4 * q;
          const int tap = k / C, c = k - tap * C;
          const int kh = tap / 3, kw = tap - 3 * kh;
          const int hh = h - p.sign * (kh - 1), ww = w - p.sign * (kw - 1);
          const bool ok = k < K && hh >= 0 && hh < p.H && ww >= 0 && ww < p.W;
          const float4 v = *(const float4*)(base + (ok ? ((hh - h) * p.W + (ww - w)) * C + c : 0));   // offset 0: this pixel itself
          gv[4 * q] = ok ? v.x : 0.f; gv[4 * q + 1] = ok ? v.y : 0.f; gv[4 * q + 2] = ok ? v.z : 0.f; gv[4 * q + 3] = ok ? v.w : 0.f;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = 16 * s + 8 * lhi + e;
          const int tap = k / C, c = k - tap * C;
          const int kh = tap / 3, kw = tap - 3 * kh;
          const int hh = h - p.sign * (kh - 1), ww = w - p.sign * (kw - 1);
          const bool ok = k < K && hh >= 0 && hh < p.H && ww >= 0 && ww < p.W;
          const float t = base[ok ? ((hh - h) * p.W + (ww - w)) * C + c : 0];
          gv[e] = ok ? t : 0.f;
        }
      }
    }
    hx[s] = __builtin_bit_cast(bf16x8, pack8(gv));
  }
  // the gate chunks (HBM) are requested AFTER the gathers (L2): vmcnt retires in order, so a wait for the gathers must not
  // have the gate stream in front of it; the chunks then fly under the MFMAs and are waited for in the epilogue.  (Round 4's
  // first form issued them first "so that both streams are in flight" -- and every gather wait included them.)
  if constexpr (EPI == TC_GATE && GQ_LATE) {
#pragma unroll
    for (int jn = 0; jn < 4; ++jn)
#pragma unroll
      for (int s = 0; s < 2; ++s) gq[jn][s] = *(const u32x4*)(p.G + chunk(jn, s));
  }
  __syncthreads();   // wf, bsh complete
  f32x16 acc[4];
#pragma unroll
  for (int jn = 0; jn < 4; ++jn)
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) acc[jn][rr] = 0.f;
#pragma unroll
  for (int s = 0; s < SMAX; ++s) {
    if (s < S) {
#pragma unroll
      for (int jn = 0; jn < 4; ++jn) {
        const bf16x8 hw = *(const bf16x8*)(wf + ((s * 4 + jn) * 64 + lane) * 8);
        acc[jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hw, hx[s], acc[jn], 0, 0, 0);   // D[co][pixel]
      }
    }
  }
  // ---- epilogue: acc[jn][r] = out[pixel = 32 mt + l31][co = n0 + 32 jn + (r & 3) + 8 (r >> 2) + 4 lhi]
#pragma unroll
  for (int jn = 0; jn < 4; ++jn)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const long ci = chunk(jn, s);
      float v[8], w8[8];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const float4 b = *(const float4*)(bsh + jn * 32 + 16 * s + 8 * a + 4 * lhi);
        v[4 * a] = acc[jn][8 * s + 4 * a] + b.x; v[4 * a + 1] = acc[jn][8 * s + 4 * a + 1] + b.y;
        v[4 * a + 2] = acc[jn][8 * s + 4 * a + 2] + b.z; v[4 * a + 3] = acc[jn][8 * s + 4 * a + 3] + b.w;
      }
      if constexpr (EPI == TC_ACT) {
#pragma unroll
        for (int e = 0; e < 8; ++e) sin_act(v[e], v[e], w8[e]);
        *(u32x4*)(p.O2 + ci) = pack8(w8);
      } else if constexpr (EPI == TC_GATE) {
        if (p.O2) *(u32x4*)(p.O2 + ci) = pack8(v);
        unpack8(gq[jn][s], w8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= w8[e];
      }
      *(u32x4*)(p.O1 + ci) = pack8(v);
    }
}

template <int EPI, int SMAX, bool VEC4>
int launch_thin_conv(const ThinConvArgs& a, dim3 grid, hipStream_t st) {
  hipLaunchKernelGGL((f16_thin_conv_kernel<EPI, SMAX, VEC4>), grid, dim3(256), 0, st, a);
  return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
}
template <int EPI>
int launch_thin_conv_e(const ThinConvArgs& a, dim3 grid, hipStream_t st) {
  if (a.C <= 3) return launch_thin_conv<EPI, 2, false>(a, grid, st);
  return launch_thin_conv<EPI, 7, true>(a, grid, st);
}

// ---------------------------------------------------------------------------------------------------------------
// f16_chain_conv_kernel (round 6): one series iteration of the log-det estimators as ONE launch --
//   P[M, J] = ((( C3^T v ) .* d2) W2T .* d1) panel        (iresblock.py:257, 269-271: the vector-Jacobian product of g)
// i.e. f16_thin_conv_kernel's gate epilogue + f16_rows_kernel's chain mode without the [M, 512] tensor u2 = (C3^T v) .* d2 between
// them (134 MB written and read back per iteration at batch 128; 54 + 57 thin-convolution launches per training step).  The
// workgroup builds its 64-row A tile itself: (0) the tile's im2col image [64 px][16 S k] is gathered ONCE into LDS as bf16 in
// fragment order (512 threads, 8 k-values each -- the thin kernel gathers per wave); (1) every wave multiplies it by its two
// 32-column blocks of the packed transposed-convolution weight (2 S MFMAs per 32 x 32 block, the thin kernel's k order and
// summation order: the tile is bit-identical to the u2 that kernel stores); (2) the accumulators are gated by d2 (chunks requested
// at kernel start), rounded to bf16 and written into the A tile at their chunk positions -- the accumulator layout IS the
// operand-fragment layout; then rows_tile<R_CHAIN> runs unchanged.  Same HBM bytes per tile as the chain kernel alone (the d2
// chunks instead of the u2 chunks).
struct ChainConvArgs {
  RowsArgs r;
  const float* X;    // [Nimg, H, W, C] fp32
  const u16* Wc;     // packed [S][16][64][8]: Wc[s][nb][lane][e] = bf16(w[16 s + 8 (lane >> 5) + e][32 nb + (lane & 31)]), rows >= 9 C zero
  const u16* G;      // F16 [M, 512]: d2
  u16* T3;           // optional: the ungated convolution output (F16), as the thin kernel's gate epilogue stores it
  u16* U2;           // optional: the gated convolution output (F16) -- the tile itself (EPI_ACT: the activation a1)
  int H, W, C, sign, S;
  const float* cbias;   // EPI_ACT: the convolution's bias [512]; the cosine goes to T3
};
enum { CE_GATE = 0, CE_ACT = 1 };

// T2OUT: the training step's LAST link also stores t3 = C3^T w (a.T3) and t2 = (t3 .* d2) W2T (a.r.O1) for the backward.
// TAIL = R_GEMM (indm_f16_conv_gemm, the backward's  q1 = conv(q0, W1), q1 .* d1, du2 = (q1 .* d1) W2): the tile is followed by the
// plain 64-row GEMM instead of the chain stages; the raw and the gated convolution outputs are stored too (a.T3, a.U2).
// EPI = CE_ACT with TAIL = R_GEMM_ACT_THINN (indm_f16_gnet, the g-net of the block in one launch): the tile is the Sin activation
// a1 = sin(2 pi p) / (2 pi), p = conv(a0, W1) + b1 (a1 -> a.U2, cos -> a.T3: the backward needs both), followed by
// (a2, d2) = Sin(a1 W2 + b2 + cb[image]) and Pn = a2 W3 -- indm_f16_thin_conv(epilogue 1) + indm_f16_rows mode 5.
template <int JB, int TMV, int SMAX, bool VEC4, bool T2OUT, int TAIL = R_CHAIN, int EPI = CE_GATE>
__global__ __launch_bounds__(R_NW * 64, 4) void f16_chain_conv_kernel(ChainConvArgs a) {
  constexpr int MB = TMV / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* im = smem + TMV * R_N * 2;                    // [MB][SMAX] chunks of 1 KB: the im2col fragments
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const long tile = blockIdx.x;
  const int C = a.C, K = 9 * C, HW = a.H * a.W;
  auto chunk = [&](int i, int j, int s) -> long {
    return ((((tile * MB + j) * 16 + 2 * wave + i) * 2 + s) * 64 + lane) * 8;   // bf16 index
  };
  // ---- (0) the im2col image of the tile's TMV pixels: one 16-byte group (pixel, 8 consecutive k) per thread and round
  for (int u = tid; u < MB * SMAX * 64; u += R_NW * 64) {
    const int ln = u & 63, sj = u >> 6;
    const int sidx = sj % SMAX, j = sj / SMAX;
    const long pix = tile * TMV + j * 32 + (ln & 31);
    const int r = (int)(pix % HW);
    const int h = r / a.W, w = r - h * a.W;
    const float* base = a.X + pix * C;
    float gv[8];
    if constexpr (VEC4) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int k = 16 * sidx + 8 * (ln >> 5) + 4 * q;
        const int tap = k / C, c = k - tap * C;
        const int kh = tap / 3, kw = tap - 3 * kh;
        const int hh = h - a.sign * (kh - 1), ww = w - a.sign * (kw - 1);
        const bool ok = k < K && hh >= 0 && hh < a.H && ww >= 0 && ww < a.W;
        const float4 v = *(const float4*)(base + (ok ? ((hh - h) * a.W + (ww - w)) * C + c : 0));
        gv[4 * q] = ok ? v.x : 0.f; gv[4 * q + 1] = ok ? v.y : 0.f; gv[4 * q + 2] = ok ? v.z : 0.f; gv[4 * q + 3] = ok ? v.w : 0.f;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = 16 * sidx + 8 * (ln >> 5) + e;
        const int tap = k / C, c = k - tap * C;
        const int kh = tap / 3, kw = tap - 3 * kh;
        const int hh = h - a.sign * (kh - 1), ww = w - a.sign * (kw - 1);
        const bool ok = k < K && hh >= 0 && hh < a.H && ww >= 0 && ww < a.W;
        const float t = base[ok ? ((hh - h) * a.W + (ww - w)) * C + c : 0];
        gv[e] = ok ? t : 0.f;
      }
    }
    *(u32x4*)(im + (long)u * 16) = pack8(gv);
  }
  // the d2 chunks of this wave's accumulators (HBM): requested behind the gathers, they fly under the convolution's MFMAs
  u32x4 gq[2][MB][2];
  if constexpr (EPI == CE_GATE) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < MB; ++j)
#pragma unroll
        for (int s = 0; s < 2; ++s) gq[i][j][s] = *(const u32x4*)(a.G + chunk(i, j, s));
  }
  __syncthreads();
  // ---- (1) the tile of C3^T v: acc[i][j] = D[co = 32 (2 wave + i) + ..][pixel = 32 j + l31]
  f32x16 acc[2][MB];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < MB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  {
    const u32x4* wc = (const u32x4*)a.Wc + (long)(2 * wave) * 64 + lane;
#pragma unroll
    for (int s = 0; s < SMAX; ++s) {
      if (s < a.S) {
        bf16x8 hx[MB];
#pragma unroll
        for (int j = 0; j < MB; ++j) hx[j] = *(const bf16x8*)(im + ((j * SMAX + s) * 64 + lane) * 16);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const bf16x8 hw = __builtin_bit_cast(bf16x8, wc[(long)s * 1024 + i * 64]);
#pragma unroll
          for (int j = 0; j < MB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hw, hx[j], acc[i][j], 0, 0, 0);
        }
      }
    }
  }
  // ---- (2) gate (or activate), round, store as operand fragments: chunk (m-block j, k-step 2 (2 wave + i) + s) of the A tile
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float bv[16];
    if constexpr (EPI == CE_ACT) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 b = *(const float4*)(a.cbias + 64 * wave + 32 * i + 8 * q + 4 * lhi);
        bv[4 * q] = b.x; bv[4 * q + 1] = b.y; bv[4 * q + 2] = b.z; bv[4 * q + 3] = b.w;
      }
    }
#pragma unroll
    for (int j = 0; j < MB; ++j)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float v[8], g[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = acc[i][j][8 * s + e];
        u32x4 pk;
        if constexpr (EPI == CE_ACT) {
#pragma unroll
          for (int e = 0; e < 8; ++e) sin_act(v[e] + bv[8 * s + e], v[e], g[e]);
          *(u32x4*)(a.T3 + chunk(i, j, s)) = pack8(g);
          pk = pack8(v);
          *(u32x4*)(a.U2 + chunk(i, j, s)) = pk;
        } else {
          if constexpr (T2OUT || TAIL == R_GEMM) {
            if (a.T3) *(u32x4*)(a.T3 + chunk(i, j, s)) = pack8(v);
          }
          unpack8(gq[i][j][s], g);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= g[e];
          pk = pack8(v);
          if constexpr (TAIL == R_GEMM) {
            if (a.U2) *(u32x4*)(a.U2 + chunk(i, j, s)) = pk;
          }
        }
        *(u32x4*)(smem + (j * 32 + 2 * (2 * wave + i) + s) * 1024 + lane * 16) = pk;
      }
  }
  __syncthreads();
  rows_tile<TAIL, JB, T2OUT, TMV, 4>(a.r, smem, tile, tid, lane, wave, l31, lhi);
}

template <int JB, int TMV, int SMAX, bool VEC4>
int launch_gnet(const ChainConvArgs& a, hipStream_t st) {
  constexpr int bytes = TMV * R_N * 2 + (TMV / 32) * SMAX * 1024;
  if (indm_lds_optin<f16_chain_conv_kernel<JB, TMV, SMAX, VEC4, false, R_GEMM_ACT_THINN, CE_ACT>>(bytes) != INDM_OK) return INDM_E_LAUNCH;
  hipLaunchKernelGGL((f16_chain_conv_kernel<JB, TMV, SMAX, VEC4, false, R_GEMM_ACT_THINN, CE_ACT>), dim3(a.r.M / TMV), dim3(R_NW * 64),
                     bytes, st, a);
  return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
}
template <int TMV, int SMAX, bool VEC4>
int launch_conv_gemm(const ChainConvArgs& a, hipStream_t st) {
  constexpr int bytes = TMV * R_N * 2 + (TMV / 32) * SMAX * 1024;
  if (indm_lds_optin<f16_chain_conv_kernel<1, TMV, SMAX, VEC4, false, R_GEMM>>(bytes) != INDM_OK) return INDM_E_LAUNCH;
  hipLaunchKernelGGL((f16_chain_conv_kernel<1, TMV, SMAX, VEC4, false, R_GEMM>), dim3(a.r.M / TMV), dim3(R_NW * 64), bytes, st, a);
  return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
}

template <int JB, int TMV, int SMAX, bool VEC4, bool T2OUT>
int launch_chain_conv_t(const ChainConvArgs& a, hipStream_t st) {
  constexpr int bytes = TMV * R_N * 2 + (TMV / 32) * SMAX * 1024;
  if (indm_lds_optin<f16_chain_conv_kernel<JB, TMV, SMAX, VEC4, T2OUT>>(bytes) != INDM_OK) return INDM_E_LAUNCH;
  hipLaunchKernelGGL((f16_chain_conv_kernel<JB, TMV, SMAX, VEC4, T2OUT>), dim3(a.r.M / TMV), dim3(R_NW * 64), bytes, st, a);
  return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
}
template <int JB, int TMV, int SMAX, bool VEC4>
int launch_chain_conv(const ChainConvArgs& a, hipStream_t st) {
  return a.r.O1 ? launch_chain_conv_t<JB, TMV, SMAX, VEC4, true>(a, st) : launch_chain_conv_t<JB, TMV, SMAX, VEC4, false>(a, st);
}

// ---------------------------------------------------------------------------------------------------------------
// f16_bwd_act_kernel: one Sin level of the iResBlock's hand-written backward on F16 tensors (the F16 form of
// indm_ires_bwd_act_f32):  dp = da .* d - 4 pi^2 a .* (du .* t), seg[img][n] = sum over the rows of image img of dp.
// All five inputs and dp share the F16 layout, so the kernel walks 16-byte groups; a workgroup owns (image, column block,
// half s) and its four waves a quarter of the image's 32-row blocks each; the column sums come out of registers
// (butterfly over the 32 row lanes, then the four waves in a fixed order: deterministic).
struct BwdActArgs {
  const u16 *da, *du, *d, *a, *t;
  u16* dp;
  float* seg;     // [nseg][512] or null
  int mts;        // 32-row blocks per segment (rows_per_seg / 32), a multiple of 4
};
__global__ __launch_bounds__(256) void f16_bwd_act_kernel(BwdActArgs p) {
  __shared__ float red[4][2][8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int img = blockIdx.x >> 5, nb = (blockIdx.x >> 1) & 15, s = blockIdx.x & 1;
  const int per = p.mts >> 2;
  const float K = TWO_PI_F * TWO_PI_F;
  float sum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) sum[e] = 0.f;
  long mt = (long)img * p.mts + wave * per;
#pragma unroll 2
  for (int i = 0; i < per; ++i, ++mt) {
    const long g = (((mt * 16 + nb) * 2 + s) * 64 + lane) * 8;
    float da[8], du[8], dd[8], aa[8], tt[8], o[8];
    unpack8(*(const u32x4*)(p.da + g), da);
    unpack8(*(const u32x4*)(p.du + g), du);
    unpack8(*(const u32x4*)(p.d + g), dd);
    unpack8(*(const u32x4*)(p.a + g), aa);
    unpack8(*(const u32x4*)(p.t + g), tt);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      o[e] = da[e] * dd[e] - K * aa[e] * (du[e] * tt[e]);
      sum[e] += o[e];
    }
    *(u32x4*)(p.dp + g) = pack8(o);
  }
  if (!p.seg) return;
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) sum[e] += __shfl_xor(sum[e], o, 64);   // over the 32 rows of a block: lanes of one lhi
  if ((lane & 31) == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) red[wave][lane >> 5][e] = sum[e];
  }
  __syncthreads();
  if (tid < 16) {
    const int lhi = tid >> 3, e = tid & 7;
    const float v = ((red[0][lhi][e] + red[1][lhi][e]) + red[2][lhi][e]) + red[3][lhi][e];
    p.seg[(long)img * R_N + 32 * nb + 16 * s + 8 * (e >> 2) + 4 * lhi + (e & 3)] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradients: contractions over the ROWS m of F16 tensors.  A fragment of such a contraction needs 8 rows of one
// column per lane, the transpose of what a chunk holds (8 columns of one row): the loaders read chunks into registers
// (1 KB contiguous per wave instruction), apply the optional gate there, and write them TRANSPOSED into an LDS image
// [column][32 rows (+8 pad)] with 2-byte stores; fragments come back with one conflict-free ds_read_b128 (row stride 80
// bytes: 16 lanes x 16 bytes land on 16 distinct 4-bank groups).
constexpr int WG_RS = 40;   // halfwords per image row: 32 rows of the slab + 8 pad

// chunk registers (16 values of lane (m = l31, lhi): s, e -> column 16 s + 8 (e >> 2) + 4 lhi + (e & 3)) -> image[column][m]
__device__ __forceinline__ void transpose_store(u16* img, const float (&v)[2][8], int l31, int lhi) {
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int n = 16 * s + 8 * (e >> 2) + 4 * lhi + (e & 3);
      img[n * WG_RS + l31] = __builtin_bit_cast(u16, (__bf16)v[s][e]);
    }
}

// f16_thin_wgrad_kernel: part[split][t][n] = sum over the split's rows m of thin[m][t] * (wide .* gate)[m][n]
// (thin: fp32 [M, T] row-major, T <= 128; wide, gate: F16 [M, 512]).  grid = 4 column tiles x S splits; wave w of a
// workgroup owns the 32-column block 4 tile + w (its transposed image is wave-private), the thin image is shared.
struct ThinWgradArgs {
  const float* thin;
  const u16 *wide, *gate;
  float* part;
  int T, mts, per;   // 32-row blocks in total / per split
};
template <int TB, bool GATED>
__global__ __launch_bounds__(256, 2) void f16_thin_wgrad_kernel(ThinWgradArgs p) {
  __shared__ __attribute__((aligned(16))) u16 At[2][TB * 32 * WG_RS];    // thin^T  [t][m]
  __shared__ __attribute__((aligned(16))) u16 Bt[2][4][32 * WG_RS];      // wide^T  [wave][n][m]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int tile = blockIdx.x & 3, split = blockIdx.x >> 2;
  const int nb = tile * 4 + wave, T = p.T;
  const int mt0 = split * p.per, mt1 = min(p.mts, mt0 + p.per);
  for (int i = tid; i < 2 * TB * 32 * WG_RS; i += 256) (&At[0][0])[i] = 0;   // rows t >= T stay zero
  f32x16 acc[TB];
#pragma unroll
  for (int tb = 0; tb < TB; ++tb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tb][r] = 0.f;
  constexpr int NTH = (TB * 32 * 32 + 255) / 256;   // thin elements per thread and slab (upper bound)
  u32x4 wq[2], gq[2];
  float th[NTH];
  auto gload = [&](int mt) {
    const long g = (((long)mt * 16 + nb) * 2) * 64 + lane;   // 16-byte group of half 0; half 1 is 64 groups on
    wq[0] = ((const u32x4*)p.wide)[g];
    wq[1] = ((const u32x4*)p.wide)[g + 64];
    if constexpr (GATED) {
      gq[0] = ((const u32x4*)p.gate)[g];
      gq[1] = ((const u32x4*)p.gate)[g + 64];
    }
    const float* ts = p.thin + (long)mt * 32 * T;
#pragma unroll
    for (int i = 0; i < NTH; ++i) {
      const int idx = tid + 256 * i;
      th[i] = idx < 32 * T ? ts[idx] : 0.f;
    }
  };
  auto sstore = [&](int buf) {
    if constexpr (GATED) {
      float v[2][8], gg[2][8];
      unpack8(wq[0], v[0]);
      unpack8(wq[1], v[1]);
      unpack8(gq[0], gg[0]);
      unpack8(gq[1], gg[1]);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[s][e] *= gg[s][e];
      transpose_store(Bt[buf][wave], v, l31, lhi);
    } else {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e)
          Bt[buf][wave][(16 * s + 8 * (e >> 2) + 4 * lhi + (e & 3)) * WG_RS + l31] = (u16)(wq[s][e >> 1] >> (16 * (e & 1)));
    }
#pragma unroll
    for (int i = 0; i < NTH; ++i) {
      const int idx = tid + 256 * i;
      if (idx < 32 * T) {
        const int m = idx / T, t = idx - m * T;
        At[buf][t * WG_RS + m] = __builtin_bit_cast(u16, (__bf16)th[i]);
      }
    }
  };
  __syncthreads();   // the zero fill
  if (mt0 < mt1) {
    gload(mt0);
    sstore(0);
  }
  __syncthreads();
  int buf = 0;
  for (int mt = mt0; mt < mt1; ++mt, buf ^= 1) {
    if (mt + 1 < mt1) gload(mt + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8 hb = *(const bf16x8*)(Bt[buf][wave] + l31 * WG_RS + 16 * ks + 8 * lhi);
#pragma unroll
      for (int tb = 0; tb < TB; ++tb) {
        const bf16x8 ha = *(const bf16x8*)(At[buf] + (tb * 32 + l31) * WG_RS + 16 * ks + 8 * lhi);
        acc[tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, acc[tb], 0, 0, 0);   // D[t][n]
      }
    }
    if (mt + 1 < mt1) sstore(buf ^ 1);
    __syncthreads();
  }
  float* out = p.part + (long)split * T * R_N;
#pragma unroll
  for (int tb = 0; tb < TB; ++tb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = tb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      if (t < T) out[(long)t * R_N + nb * 32 + l31] = acc[tb][r];
    }
}

// out = alpha * sum_s part[s][a][b] (fixed order), written as out[a][b] (ldo) or transposed out[b][a] (ldo)
__global__ __launch_bounds__(256) void f16_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int S,
                                                               int A, int B, float alpha, int transpose, long ldo) {
  const int total = A * B;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int s = 0;
  for (; s + 3 < S; s += 4) {
    s0 += part[(long)s * total + i];
    s1 += part[(long)(s + 1) * total + i];
    s2 += part[(long)(s + 2) * total + i];
    s3 += part[(long)(s + 3) * total + i];
  }
  for (; s < S; ++s) s0 += part[(long)s * total + i];
  const float v = alpha * ((s0 + s1) + (s2 + s3));
  const int a = i / B, b = i - a * B;
  if (transpose) out[(long)b * ldo + a] = v;
  else out[(long)a * ldo + b] = v;
}

// f16_wgrad_gemm_kernel: part[split][n1][n2] = sum over the split's rows m of (X .* G)[m][n1] * Y[m][n2]; X, G, Y: F16
// [M, 512].  128 x 128 output tiles (16 of them), four waves 2 x 2 of 64 x 64; the 16 tiles of one split run on ONE XCD
// (block -> (split, tile) below), so that a slab's chunks come out of that XCD's L2 after the first tile has fetched them.
struct WgradGemmArgs {
  const u16 *X, *G, *Y;
  float* part;
  int mts, per, S;
};
// halfword e of a chunk's four words, as stored (no float round trip for operands that are not gated)
__device__ __forceinline__ u16 half_of(const u32x4& q, int e) { return (u16)(q[e >> 1] >> (16 * (e & 1))); }

// SL = 32-row slabs per barrier: 2 for the plain form (half the barriers: 145 -> 133 us at M = 131072), 1 for the gated form
// (with two slabs in flight its chunk registers spill and it ran 172 -> 215 us)
template <bool GATED, int SL>
__global__ __launch_bounds__(256, 2) void f16_wgrad_gemm_kernel(WgradGemmArgs p) {
  constexpr int RS = 32 * SL + 8;   // halfwords per image row (80- / 144-byte rows: ds_read_b128 conflict-free)
  extern __shared__ __attribute__((aligned(16))) char wg_smem[];     // up to 72 KB: above the 64 KB a kernel gets without opt-in
  u16 (*Xt)[128 * RS] = (u16 (*)[128 * RS])wg_smem;
  u16 (*Yt)[128 * RS] = (u16 (*)[128 * RS])(wg_smem + 2 * 128 * RS * 2);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  // block b runs on XCD b % 8: splits xcd, xcd + 8, ... live there, 16 consecutive local blocks = the tiles of one split
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int tile = loc & 15, split = (loc >> 4) * 8 + xcd;
  if (split >= p.S) return;
  const int t1 = tile >> 2, t2 = tile & 3;           // 128-column tiles of X (rows of D) and of Y (columns of D)
  const int w1 = wave >> 1, w2 = wave & 1;
  const int mt0 = split * p.per, mt1 = min(p.mts, mt0 + p.per);
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // loader: wave w stages the 32-column block w of the X tile and of the Y tile, SL 32-row slabs per step
  u32x4 xq[SL][2], gq[SL][2], yq[SL][2];
  const u32x4 zero = {0u, 0u, 0u, 0u};
  auto gload = [&](int mt) {
#pragma unroll
    for (int u = 0; u < SL; ++u) {
      if (mt + u < mt1) {
        const long gx = (((long)(mt + u) * 16 + t1 * 4 + wave) * 2) * 64 + lane, gy = (((long)(mt + u) * 16 + t2 * 4 + wave) * 2) * 64 + lane;
        xq[u][0] = ((const u32x4*)p.X)[gx];
        xq[u][1] = ((const u32x4*)p.X)[gx + 64];
        yq[u][0] = ((const u32x4*)p.Y)[gy];
        yq[u][1] = ((const u32x4*)p.Y)[gy + 64];
        if constexpr (GATED) {
          gq[u][0] = ((const u32x4*)p.G)[gx];
          gq[u][1] = ((const u32x4*)p.G)[gx + 64];
        }
      } else {   // odd tail of the split: a slab of zeros
        xq[u][0] = xq[u][1] = yq[u][0] = yq[u][1] = zero;
        if constexpr (GATED) gq[u][0] = gq[u][1] = zero;
      }
    }
  };
  auto sstore = [&](int buf) {
    u16* xi = Xt[buf] + wave * 32 * RS;
    u16* yi = Yt[buf] + wave * 32 * RS;
#pragma unroll
    for (int u = 0; u < SL; ++u) {
      if constexpr (GATED) {
        float v[2][8], gg[2][8];
        unpack8(xq[u][0], v[0]);
        unpack8(xq[u][1], v[1]);
        unpack8(gq[u][0], gg[0]);
        unpack8(gq[u][1], gg[1]);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int e = 0; e < 8; ++e)
            xi[(16 * s + 8 * (e >> 2) + 4 * lhi + (e & 3)) * RS + 32 * u + l31] = __builtin_bit_cast(u16, (__bf16)(v[s][e] * gg[s][e]));
      } else {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int e = 0; e < 8; ++e) xi[(16 * s + 8 * (e >> 2) + 4 * lhi + (e & 3)) * RS + 32 * u + l31] = half_of(xq[u][s], e);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) yi[(16 * s + 8 * (e >> 2) + 4 * lhi + (e & 3)) * RS + 32 * u + l31] = half_of(yq[u][s], e);
    }
  };
  if (mt0 < mt1) {
    gload(mt0);
    sstore(0);
  }
  __syncthreads();
  int buf = 0;
  for (int mt = mt0; mt < mt1; mt += SL, buf ^= 1) {
    if (mt + SL < mt1) gload(mt + SL);
#pragma unroll
    for (int ks = 0; ks < 2 * SL; ++ks) {
      bf16x8 ha[2], hb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) ha[i] = *(const bf16x8*)(Xt[buf] + ((w1 * 2 + i) * 32 + l31) * RS + 16 * ks + 8 * lhi);
#pragma unroll
      for (int j = 0; j < 2; ++j) hb[j] = *(const bf16x8*)(Yt[buf] + ((w2 * 2 + j) * 32 + l31) * RS + 16 * ks + 8 * lhi);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha[i], hb[j], acc[i][j], 0, 0, 0);   // D[n1][n2]
    }
    if (mt + SL < mt1) sstore(buf ^ 1);
    __syncthreads();
  }
  float* out = p.part + (long)split * R_N * R_N;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n1 = t1 * 128 + (w1 * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        const int n2 = t2 * 128 + (w2 * 2 + j) * 32 + l31;
        out[(long)n1 * R_N + n2] = acc[i][j][r];
      }
}

// Round 6: the same contraction with the operands copied chunk by chunk straight into LDS (direct-to-LDS, no register
// staging) and the fragments built by TRANSPOSING LDS reads instead of 64 two-byte LDS stores per thread and slab (those
// stores, not the MFMAs or the bytes, bounded f16_wgrad_gemm_kernel: 404 TF-equivalent at M = 131072).
//   * an F16 chunk (mt, nb, s) is 1 KB: lane-slot (lhi, r) holds row r's 8 columns 16 s + 8 (e >> 2) + 4 lhi + (e & 3).  The
//     copy is lane-linear, but WHICH source slot a lane copies is free: LDS slot q <- source slot (lhi = q & 1, r = q >> 1),
//     so a row's two halves are adjacent (32 bytes) and four consecutive rows are 128 contiguous bytes; chunks sit 1152 bytes
//     apart (1 KB + 128: the two chunks a half-wave reads fall into complementary bank halves -> conflict-free).
//   * ds_read_b64_tr_b16: within a 16-lane group lane i supplies the address of piece (i & 3) of row (i >> 2) and receives, for
//     the four rows, column col(i) = {0..3, 8..11, 4..7, 12..15}[i] of the chunk -- the MFMA operand "8 consecutive rows of one
//     column" in two reads.  The columns of a 32-column block therefore reach the MFMA in a fixed permutation, undone when the
//     partial tile is written.
//   * gated form: the X and G fragments are multiplied in fp32 and rounded once (the rounding of the old kernel's store).
// Same arguments, same [split][512][512] partial slabs and reduce kernel as f16_wgrad_gemm_kernel.
__device__ __forceinline__ int tr_col(int i) { return (i & 3) + 8 * ((i >> 2) & 1) + 4 * (i >> 3); }

template <bool GATED>
__global__ __launch_bounds__(256, 2) void f16_wgrad_tr_kernel(WgradGemmArgs p) {
  constexpr int CH = 1152;                         // bytes per chunk slot
  constexpr int OPB = 8 * CH;                      // one operand's 128 columns x 32 rows
  constexpr int NOP = GATED ? 3 : 2;
  constexpr int BUF = NOP * OPB;
  constexpr int NBUF = GATED ? 2 : 3, D = NBUF - 1;   // (gated: three operands x three stages would leave one workgroup per CU)
  constexpr int NDMA = 2 * NOP;                    // copies per wave and slab
  extern __shared__ __attribute__((aligned(16))) char wt_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int tile = loc & 15, split = (loc >> 4) * 8 + xcd;
  if (split >= p.S) return;
  const int t1 = tile >> 2, t2 = tile & 3;
  const int w1 = wave >> 1, w2 = wave & 1;
  const int mt0 = split * p.per, mt1 = min(p.mts, mt0 + p.per);
  const int nk = mt1 - mt0;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // copies: wave w stages chunks 2 w, 2 w + 1 of each operand's tile (chunk c = 2 * local column block + s)
  const unsigned src_lane = (unsigned)((32 * (lane & 1) + (lane >> 1)) * 16);   // source slot of this lane's LDS slot
  auto stage = [&](int buf, int mt) {
    char* base = wt_smem + buf * BUF;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int c = 2 * wave + q;
      const long cx = ((long)mt * 16 + t1 * 4 + (c >> 1)) * 2 + (c & 1), cy = ((long)mt * 16 + t2 * 4 + (c >> 1)) * 2 + (c & 1);
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                   :: "s"(lds_addr(base + c * CH)), "v"(src_lane), "s"((const char*)p.X + cx * 1024) : "memory", "m0");
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                   :: "s"(lds_addr(base + OPB + c * CH)), "v"(src_lane), "s"((const char*)p.Y + cy * 1024) : "memory", "m0");
      if constexpr (GATED)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                     :: "s"(lds_addr(base + 2 * OPB + c * CH)), "v"(src_lane), "s"((const char*)p.G + cx * 1024) : "memory", "m0");
    }
  };
  // transposing reads: lane (i = lane & 15, g1, lhi) supplies piece (i & 3) of row 8 lhi + (i >> 2) (+ 16 ks, + 4 h) of chunk g1
  const int i16 = lane & 15, g1 = (lane >> 4) & 1;
  const unsigned rd = (unsigned)(g1 * CH + (2 * (8 * lhi + (i16 >> 2)) + ((i16 >> 1) & 1)) * 16 + (i16 & 1) * 8);
  typedef short s16x4_t __attribute__((ext_vector_type(4)));
  typedef short s16x8_t __attribute__((ext_vector_type(8)));
  auto trd = [](const char* q) -> s16x4_t {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)q);
  };
  auto frag = [&](const char* op, int blk, int ks) -> bf16x8 {   // rows 16 ks + 8 lhi + 0 .. 7 of this lane's column in block blk
    const char* q = op + blk * 2 * CH + rd + ks * 512;            // (16 rows = 32 slots = 512 bytes)
    const s16x4_t lo = trd(q), hi = trd(q + 128);                 // (4 rows = 8 slots = 128 bytes)
    const s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
  };
  auto wait = [](int n) {   // n = 1: the newest stage may stay in flight (three buffers only)
    if (n == 0 || D == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  };
  auto barrier = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  };
  static_assert(GATED || NDMA == 4, "the counted wait above");
  {
    const int pre = nk < D ? nk : D;
    for (int d = 0; d < pre; ++d) stage(d, mt0 + d);
    wait(nk >= D ? 1 : 0);
    barrier();
  }
  int buf = 0, sbuf = D;
  for (int s = 0; s < nk; ++s) {
    if (s + D < nk) stage(sbuf, mt0 + s + D);
    const char* xs = wt_smem + buf * BUF;
    const char* ys = xs + OPB;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 ha[2], hb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ha[i] = frag(xs, w1 * 2 + i, ks);
        if constexpr (GATED) {
          const bf16x8 hg = frag(xs + 2 * OPB, w1 * 2 + i, ks);
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (float)ha[i][e] * (float)hg[e];
          ha[i] = indm_pack_bf16x8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) hb[j] = frag(ys, w2 * 2 + j, ks);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha[i], hb[j], acc[i][j], 0, 0, 0);   // D[n1][n2]
    }
    __builtin_amdgcn_sched_barrier(0);
    wait(s + 2 < nk ? 1 : 0);
    barrier();
    buf = buf == D ? 0 : buf + 1;
    sbuf = sbuf == D ? 0 : sbuf + 1;
  }
  float* out = p.part + (long)split * R_N * R_N;
  const int n2l = 16 * (l31 >> 4) + tr_col(l31 & 15);            // the column this lane's B fragments carried
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ai = (r & 3) + 8 * (r >> 2) + 4 * lhi;         // the A-operand lane whose column this accumulator row is
        const int n1 = t1 * 128 + (w1 * 2 + i) * 32 + 16 * (ai >> 4) + tr_col(ai & 15);
        const int n2 = t2 * 128 + (w2 * 2 + j) * 32 + n2l;
        out[(long)n1 * R_N + n2] = acc[i][j][r];
      }
}

// f16_tap_spread_kernel: the im2col image of a skinny NHWC fp32 tensor (C % 4 == 0, 9 C <= 512) as an F16 [pixels, 512] operand:
//   out[pixel][(tap, c)] = x[pixel shifted by -sign (tap - 1)][c], columns >= 9 C zero
// -- the 48-channel scale of the 64 x 64 flows (9 C = 432): its 3x3 convolutions with the 512-wide side become the 64-row
// GEMM (modes 0 / 1 / 4 of f16_rows_kernel) on this operand, its weight gradients the row contraction of two F16 tensors.
// One thread per 16-byte group of the output (1 KB contiguous per wave); a group's two float4 halves never straddle a tap.
__global__ __launch_bounds__(256) void f16_tap_spread_kernel(const float* __restrict__ x, u16* __restrict__ y, long M, int H,
                                                             int W, int C, int sign) {
  const long groups = M * 64;
  const int K = 9 * C, HW = H * W;
  for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < groups; g += (long)gridDim.x * 256) {
    const int lane = (int)(g & 63);
    const long c = g >> 6;
    const int s = (int)(c & 1), nb = (int)((c >> 1) & 15);
    const long mt = c >> 5;
    const int l31 = lane & 31, lhi = lane >> 5;
    const long pix = mt * 32 + l31;
    const int r = (int)(pix % HW);
    const int h = r / W, w = r - h * W;
    const float* base = x + pix * C;
    float v[8];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int n0 = 32 * nb + 16 * s + 8 * q + 4 * lhi;
      const int tap = n0 / C, cc = n0 - tap * C;
      const int kh = tap / 3, kw = tap - 3 * kh;
      const int hh = h - sign * (kh - 1), ww = w - sign * (kw - 1);
      const bool ok = n0 < K && hh >= 0 && hh < H && ww >= 0 && ww < W;
      const float4 t = *(const float4*)(base + (ok ? ((hh - h) * W + (ww - w)) * C + cc : 0));
      v[4 * q] = ok ? t.x : 0.f; v[4 * q + 1] = ok ? t.y : 0.f; v[4 * q + 2] = ok ? t.z : 0.f; v[4 * q + 3] = ok ? t.w : 0.f;
    }
    *(u32x4*)(y + g * 8) = pack8(v);
  }
}

inline bool al16(const void* q) { return ((uintptr_t)q & 15) == 0; }

}  // namespace

// ---- C ABI ----------------------------------------------------------------------------------------------------
INDM_API int indm_f16_pack_f32(const float* x, void* y, long M, int N, void* stream) {
  if (M < 0 || N <= 0 || (M & 31) || (N & 31)) return INDM_E_SHAPE;
  if (M == 0) return INDM_OK;
  if (!x || !y) return INDM_E_ARG;
  hipLaunchKernelGGL(f16_pack_kernel, dim3(ew_grid(M * N / 8, 256)), dim3(256), 0, (hipStream_t)stream, x, (u16*)y, M, N);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}
INDM_API int indm_f16_unpack_f32(const void* x, float* y, long M, int N, void* stream) {
  if (M < 0 || N <= 0 || (M & 31) || (N & 31)) return INDM_E_SHAPE;
  if (M == 0) return INDM_OK;
  if (!x || !y) return INDM_E_ARG;
  hipLaunchKernelGGL(f16_unpack_kernel, dim3(ew_grid(M * N / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const u16*)x, y, M, N);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}

INDM_API int indm_f16_thin_conv(const float* x, const float* w, const float* bias, const void* gate, void* out1, void* out2,
                                int Nimg, int H, int W, int C, int Cout, int sign, int epilogue, void* stream) {
  if (Nimg < 0 || H <= 0 || W <= 0 || C <= 0 || Cout <= 0) return INDM_E_SHAPE;
  if (Nimg == 0) return INDM_OK;
  const long M = (long)Nimg * H * W;
  if (!(C <= 3 || (C <= 12 && !(C & 3))) || (Cout % TC_TN) || (M % TC_TM) || M * C >= (1L << 31)) return 1;
  if (!al16(x)) return 1;
  if (!x || !w || !out1 || (epilogue == TC_ACT && !out2) || (epilogue == TC_GATE && !gate)) return INDM_E_ARG;
  if (!al16(bias) || !al16(out1) || !al16(out2) || !al16(gate)) return INDM_E_ARG;
  ThinConvArgs a = {x, w, bias, (const u16*)gate, (u16*)out1, (u16*)out2, Nimg, H, W, C, Cout, sign};
  const int tiles_n = Cout / TC_TN, tiles_m = (int)(M / TC_TM);
  // (several pixel tiles per workgroup and weight pack were measured SLOWER at every production shape -- 64 -> 74 us gated,
  // 90 -> 106 us gated + raw at M = 131072: fewer, longer workgroups overlap each other's phases less)
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(tiles_m * tiles_n);
  if (epilogue == TC_PLAIN) return launch_thin_conv_e<TC_PLAIN>(a, grid, st);
  if (epilogue == TC_ACT) return launch_thin_conv_e<TC_ACT>(a, grid, st);
  if (epilogue == TC_GATE) return launch_thin_conv_e<TC_GATE>(a, grid, st);
  return INDM_E_ARG;
}

INDM_API int indm_f16_tap_spread(const float* x, void* out, int Nimg, int H, int W, int C, int sign, void* stream) {
  if (Nimg < 0 || H <= 0 || W <= 0 || C <= 0) return INDM_E_SHAPE;
  if (Nimg == 0) return INDM_OK;
  const long M = (long)Nimg * H * W;
  if ((C & 3) || 9 * C > R_N || (M % 32) || M * C >= (1L << 31)) return 1;
  if (!x || !out || !al16(x) || !al16(out)) return INDM_E_ARG;
  hipLaunchKernelGGL(f16_tap_spread_kernel, dim3(ew_grid(M * 64, 256)), dim3(256), 0, (hipStream_t)stream, x, (u16*)out, M, H, W,
                     C, sign);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}

INDM_API int indm_f16_rows(int mode, const void* A, const void* Bp, const void* D1, const void* Pp, float* P, void* out1,
                           void* out2, const float* bias, const float* bias2, int M, int J, int rows_per_group, void* stream) {
  if (M < 0) return INDM_E_SHAPE;
  if (M == 0) return INDM_OK;
  if (M % R_TM) return 1;
  if (!A || !al16(A) || !al16(Bp) || !al16(D1) || !al16(Pp) || !al16(out1) || !al16(out2) || !al16(bias) || !al16(bias2))
    return INDM_E_ARG;
  RowsArgs a = {(const u16*)A, (const u16*)Bp, (const u16*)D1, (const u16*)Pp, P, (u16*)out1, (u16*)out2, bias, bias2,
                M, J, rows_per_group > 0 ? rows_per_group : 1};
  hipStream_t st = (hipStream_t)stream;
  switch (mode) {
    case R_GEMM:
      if (!Bp || !out1) return INDM_E_ARG;
      if (bias2 && (a.rpg % R_TM)) return 1;
      return launch_rows<R_GEMM, 1, false>(a, st);
    case R_GEMM_ACT:
      if (!Bp || !out1 || !out2) return INDM_E_ARG;
      if (bias2 && (a.rpg % R_TM)) return 1;
      return launch_rows<R_GEMM_ACT, 1, false>(a, st);
    case R_GEMM_GATE:
      if (!Bp || !D1 || !out1) return INDM_E_ARG;
      return launch_rows<R_GEMM_GATE, 1, false>(a, st);
    case R_GEMM_ACT_THINN:
      if (!Bp || !out1 || !out2 || !Pp || !P || J < 1 || J > 128) return INDM_E_ARG;
      if (bias2 && (a.rpg % R_TM)) return 1;
      return J <= 32 ? launch_rows<R_GEMM_ACT_THINN, 1, false>(a, st) : launch_rows<R_GEMM_ACT_THINN, 4, false>(a, st);
    case R_CHAIN:
      if (!Bp || !D1 || !Pp || !P || J < 1 || J > 448) return INDM_E_ARG;
      if (J <= 32) return out1 ? launch_rows<R_CHAIN, 1, true>(a, st) : launch_rows<R_CHAIN, 1, false>(a, st);
      if (J <= 128) return out1 ? launch_rows<R_CHAIN, 4, true>(a, st) : launch_rows<R_CHAIN, 4, false>(a, st);
      return out1 ? launch_rows<R_CHAIN, 14, true>(a, st) : launch_rows<R_CHAIN, 14, false>(a, st);   // 48 channels: 9 C = 432
    case R_THINN:
      if (!Pp || !P || J < 1 || J > 448) return INDM_E_ARG;
      if (J > 128) return launch_rows<R_THINN, 14, false>(a, st);
      return J <= 32 ? launch_rows<R_THINN, 1, false>(a, st) : launch_rows<R_THINN, 4, false>(a, st);
    default:
      return INDM_E_ARG;
  }
}

INDM_API int indm_f16_chain_conv(const float* x, const void* Wc, const void* gate, const void* W2Tp, const void* D1,
                                 const void* Pp, float* P, void* t3_out, void* t2_out, int Nimg, int H, int W, int C, int sign, int J,
                                 void* stream) {
  if (Nimg < 0 || H <= 0 || W <= 0 || C <= 0) return INDM_E_SHAPE;
  if (Nimg == 0) return INDM_OK;
  const long M = (long)Nimg * H * W;
  if (!(C <= 3 || (C <= 12 && !(C & 3))) || (M % R_TM) || M * C >= (1L << 31) || M >= (1L << 31) || J != 9 * C) return 1;
  if (!x || !Wc || !gate || !W2Tp || !D1 || !Pp || !P) return INDM_E_ARG;
  if (!al16(x) || !al16(Wc) || !al16(gate) || !al16(W2Tp) || !al16(D1) || !al16(Pp) || !al16(t3_out) || !al16(t2_out)) return INDM_E_ARG;
  if (t3_out && !t2_out) return INDM_E_ARG;   // (t3 is stored by the form that stores t2: the training step's last link)
  ChainConvArgs a = {{nullptr, (const u16*)W2Tp, (const u16*)D1, (const u16*)Pp, P, (u16*)t2_out, nullptr, nullptr, nullptr, (int)M, J, 1},
                     x, (const u16*)Wc, (const u16*)gate, (u16*)t3_out, nullptr, H, W, C, sign, (9 * C + 15) >> 4, nullptr};
  hipStream_t st = (hipStream_t)stream;
  const bool small = M / R_TM < 512;   // the tile rule of launch_rows
  if (C <= 3) return small ? launch_chain_conv<1, 32, 2, false>(a, st) : launch_chain_conv<1, 64, 2, false>(a, st);
  return small ? launch_chain_conv<4, 32, 7, true>(a, st) : launch_chain_conv<4, 64, 7, true>(a, st);
}

INDM_API int indm_f16_conv_gemm(const float* x, const void* Wc, const void* gate, const void* Wp, void* out, void* raw_out,
                                void* gated_out, int Nimg, int H, int W, int C, int sign, void* stream) {
  if (Nimg < 0 || H <= 0 || W <= 0 || C <= 0) return INDM_E_SHAPE;
  if (Nimg == 0) return INDM_OK;
  const long M = (long)Nimg * H * W;
  if (!(C <= 3 || (C <= 12 && !(C & 3))) || (M % R_TM) || M * C >= (1L << 31) || M >= (1L << 31)) return 1;
  if (!x || !Wc || !gate || !Wp || !out) return INDM_E_ARG;
  if (!al16(x) || !al16(Wc) || !al16(gate) || !al16(Wp) || !al16(out) || !al16(raw_out) || !al16(gated_out)) return INDM_E_ARG;
  ChainConvArgs a = {{nullptr, (const u16*)Wp, nullptr, nullptr, nullptr, (u16*)out, nullptr, nullptr, nullptr, (int)M, 0, 1},
                     x, (const u16*)Wc, (const u16*)gate, (u16*)raw_out, (u16*)gated_out, H, W, C, sign, (9 * C + 15) >> 4, nullptr};
  hipStream_t st = (hipStream_t)stream;
  const bool small = M / R_TM < 512;
  if (C <= 3) return small ? launch_conv_gemm<32, 2, false>(a, st) : launch_conv_gemm<64, 2, false>(a, st);
  return small ? launch_conv_gemm<32, 7, true>(a, st) : launch_conv_gemm<64, 7, true>(a, st);
}

INDM_API int indm_f16_gnet(const float* x, const void* W1c, const float* b1, void* a1, void* d1, const void* W2p, const float* b2,
                           const float* cb, int rows_per_group, void* a2, void* d2, const void* W3p, float* P, int Nimg, int H, int W,
                           int C, void* stream) {
  if (Nimg < 0 || H <= 0 || W <= 0 || C <= 0) return INDM_E_SHAPE;
  if (Nimg == 0) return INDM_OK;
  const long M = (long)Nimg * H * W;
  const int J = 9 * C;
  if (!(C <= 3 || (C <= 12 && !(C & 3))) || (M % R_TM) || M * C >= (1L << 31) || M >= (1L << 31) || M / R_TM < 512) return 1;
  if (cb && (rows_per_group <= 0 || rows_per_group % R_TM)) return 1;
  if (!x || !W1c || !b1 || !a1 || !d1 || !W2p || !a2 || !d2 || !W3p || !P) return INDM_E_ARG;
  if (!al16(x) || !al16(W1c) || !al16(b1) || !al16(a1) || !al16(d1) || !al16(W2p) || !al16(b2) || !al16(cb) || !al16(a2) || !al16(d2) ||
      !al16(W3p))
    return INDM_E_ARG;
  ChainConvArgs a = {{nullptr, (const u16*)W2p, nullptr, (const u16*)W3p, P, (u16*)a2, (u16*)d2, b2, cb, (int)M, J,
                      rows_per_group > 0 ? rows_per_group : 1},
                     x, (const u16*)W1c, nullptr, (u16*)d1, (u16*)a1, H, W, C, -1, (J + 15) >> 4, b1};
  hipStream_t st = (hipStream_t)stream;
  if (C <= 3) return launch_gnet<1, 64, 2, false>(a, st);
  return launch_gnet<4, 64, 7, true>(a, st);
}

INDM_API long indm_f16_bwd_act_ok(long R, long rows_per_seg) {
  return (R > 0 && rows_per_seg > 0 && !(rows_per_seg % 128) && !(R % rows_per_seg)) ? 1 : 0;
}
INDM_API int indm_f16_bwd_act(const void* da, const void* du, const void* d, const void* a, const void* t, void* dp,
                              float* seg, long R, long rows_per_seg, void* stream) {
  if (R < 0) return INDM_E_SHAPE;
  if (R == 0) return INDM_OK;
  if (!indm_f16_bwd_act_ok(R, rows_per_seg)) return 1;
  if (!da || !du || !d || !a || !t || !dp) return INDM_E_ARG;
  BwdActArgs p = {(const u16*)da, (const u16*)du, (const u16*)d, (const u16*)a, (const u16*)t, (u16*)dp, seg,
                  (int)(rows_per_seg / 32)};
  hipLaunchKernelGGL(f16_bwd_act_kernel, dim3((unsigned)(R / rows_per_seg) * 32), dim3(256), 0, (hipStream_t)stream, p);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}

static int wgrad_splits(int mts, int target) {
  int S = mts < target ? mts : target;
  const int per = (mts + S - 1) / S;
  return (mts + per - 1) / per;
}
INDM_API long indm_f16_wgrad_ws_floats(long M, int T) {
  const int mts = (int)(M / 32);
  if (T > 0) return (long)wgrad_splits(mts, 128) * T * R_N;
  return (long)wgrad_splits(mts, 32) * R_N * R_N;
}
INDM_API int indm_f16_thin_wgrad(const float* thin, const void* wide, const void* gate, float* out, long M, int T, float alpha,
                                 int transpose, long ldo, float* ws, long ws_floats, void* stream) {
  if (M < 0 || T <= 0) return INDM_E_SHAPE;
  if (M == 0) return INDM_OK;
  if ((M % 32) || T > 128 || M >= (1L << 36)) return 1;
  if (!thin || !wide || !out || !ws || !al16(wide) || !al16(gate)) return INDM_E_ARG;
  const int mts = (int)(M / 32), S = wgrad_splits(mts, 128), per = (mts + S - 1) / S;
  if ((long)S * T * R_N > ws_floats) return INDM_E_ARG;
  ThinWgradArgs p = {thin, (const u16*)wide, (const u16*)gate, ws, T, mts, per};
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(4 * S);
  if (T <= 32) {
    if (gate) hipLaunchKernelGGL((f16_thin_wgrad_kernel<1, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((f16_thin_wgrad_kernel<1, false>), grid, dim3(256), 0, st, p);
  } else {
    if (gate) hipLaunchKernelGGL((f16_thin_wgrad_kernel<4, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((f16_thin_wgrad_kernel<4, false>), grid, dim3(256), 0, st, p);
  }
  INDM_CHECK_LAUNCH();
  hipLaunchKernelGGL(f16_wgrad_reduce_kernel, dim3(cdiv((long)T * R_N, 256)), dim3(256), 0, st, ws, out, S, T, R_N, alpha,
                     transpose, ldo);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}
INDM_API int indm_f16_wgrad_gemm(const void* X, const void* gateX, const void* Y, float* out, long M, float alpha, int transpose,
                                 float* ws, long ws_floats, void* stream) {
  if (M < 0) return INDM_E_SHAPE;
  if (M == 0) return INDM_OK;
  if ((M % 32) || M >= (1L << 36)) return 1;
  if (!X || !Y || !out || !ws || !al16(X) || !al16(Y) || !al16(gateX)) return INDM_E_ARG;
  const int mts = (int)(M / 32), S = wgrad_splits(mts, 32), per = (mts + S - 1) / S;
  if ((long)S * R_N * R_N > ws_floats) return INDM_E_ARG;
  WgradGemmArgs p = {(const u16*)X, (const u16*)gateX, (const u16*)Y, ws, mts, per, S};
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(((S + 7) / 8) * 8 * 16);
  static const int use_tr = getenv("INDM_F16_WGRAD_TR") ? atoi(getenv("INDM_F16_WGRAD_TR")) : 1;
  if (use_tr) {   // round 6: direct-to-LDS chunks + transposing reads (f16_wgrad_tr_kernel)
    constexpr int ldsg = 2 * 3 * 8 * 1152, ldsp = 3 * 2 * 8 * 1152;
    if (gateX) {
      if (indm_lds_optin<f16_wgrad_tr_kernel<true>>(ldsg) != INDM_OK) return INDM_E_LAUNCH;
      hipLaunchKernelGGL((f16_wgrad_tr_kernel<true>), grid, dim3(256), ldsg, st, p);
    } else {
      if (indm_lds_optin<f16_wgrad_tr_kernel<false>>(ldsp) != INDM_OK) return INDM_E_LAUNCH;
      hipLaunchKernelGGL((f16_wgrad_tr_kernel<false>), grid, dim3(256), ldsp, st, p);
    }
    INDM_CHECK_LAUNCH();
    hipLaunchKernelGGL(f16_wgrad_reduce_kernel, dim3(cdiv((long)R_N * R_N, 256)), dim3(256), 0, st, ws, out, S, R_N, R_N, alpha,
                       transpose, (long)R_N);
    INDM_CHECK_LAUNCH();
    return INDM_OK;
  }
  constexpr int lds2 = 4 * 128 * (32 * 2 + 8) * 2, lds1 = 4 * 128 * (32 + 8) * 2;
  if (!gateX && indm_lds_optin<f16_wgrad_gemm_kernel<false, 2>>(lds2) != INDM_OK) return INDM_E_LAUNCH;   // 72 KB (the gated form: 40 KB)
  if (gateX) hipLaunchKernelGGL((f16_wgrad_gemm_kernel<true, 1>), grid, dim3(256), lds1, st, p);
  else hipLaunchKernelGGL((f16_wgrad_gemm_kernel<false, 2>), grid, dim3(256), lds2, st, p);
  INDM_CHECK_LAUNCH();
  hipLaunchKernelGGL(f16_wgrad_reduce_kernel, dim3(cdiv((long)R_N * R_N, 256)), dim3(256), 0, st, ws, out, S, R_N, R_N, alpha,
                     transpose, (long)R_N);
  INDM_CHECK_LAUNCH();
  return INDM_OK;
}

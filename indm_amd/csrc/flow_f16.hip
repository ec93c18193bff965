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
  bf16x8 h;
#pragma unroll
  for (int e = 0; e < 8; ++e) h[e] = (__bf16)v[e];
  return __builtin_bit_cast(u32x4, h);
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
enum { R_GEMM = 0, R_GEMM_ACT = 1, R_CHAIN = 2, R_THINN = 3 };
constexpr int R_TM = 64, R_N = 512, R_NW = 8;
constexpr int R_LDS_BYTES = R_TM * R_N * 2;   // the A tile; the stage-2 reduction buffer [8][64][32] fp32 overlays it

struct RowsArgs {
  const u16* A;      // F16 [M, 512]
  const u16* Bp;     // packed [32][16][64][8] (stage 1)
  const u16* D1;     // F16 [M, 512]: gate of the chain epilogue
  const u16* Pp;     // packed panel [32][JB][64][8] (stage 2 / THIN_N)
  float* P;          // [M, J] fp32 row-major
  u16* O1;           // GEMM: C; GEMM_ACT: a; CHAIN: t2 (optional)
  u16* O2;           // GEMM_ACT: cos
  const float* bias;   // [512] (GEMM_ACT / GEMM, optional)
  const float* bias2;  // [M / rpg][512] (optional)
  int M, J, rpg;
};

template <int MODE, int JB, bool T2OUT>
__global__ __launch_bounds__(R_NW * 64, 4) void f16_rows_kernel(RowsArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const long tile = blockIdx.x;
  const long m0 = tile * R_TM;

  // ---- the A tile: 64 rows x 512 columns of F16 memory = 64 consecutive 1 KB chunks; chunk c = 32 j + ks is fragment
  // (m-block j, k-step ks).  Eight chunks per wave, two copy groups.
  {
    const char* src = (const char*)p.A + tile * (long)R_LDS_BYTES + (long)wave * 8192;
    const unsigned voff = 16u * lane;
    dma_x4(lds_addr(smem + wave * 8192), voff, src);
    dma_x4(lds_addr(smem + wave * 8192 + 4096), voff, src + 4096);
  }
  f32x16 acc[2][2];
  bf16x8 hm2[2][2][2];   // stage-2 operand fragments [i][j][s] (CHAIN: gated accumulators; THIN_N: A fragments)
  if constexpr (MODE != R_THINN) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // weights of this wave's two 32-column blocks, k-step ks: Bp[(ks * 16 + 2 wave + i) * 64 + lane] (16 bytes)
    const u32x4* bp = (const u32x4*)p.Bp + (long)(2 * wave) * 64 + lane;
    constexpr int D = 4;   // k-steps of weight fragments in flight (L2 latency / MFMA time of a k-step round)
    u32x4 q[D][2];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
      for (int i = 0; i < 2; ++i) q[d][i] = bp[(long)d * 1024 + i * 64];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's copies have landed ...
    __syncthreads();                                    // ... and everybody else's
#pragma unroll 1
    for (int k0 = 0; k0 < 32; k0 += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int ks = k0 + d;
        bf16x8 hm[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) hm[j] = *(const bf16x8*)(smem + (j * 32 + ks) * 1024 + lane * 16);
        bf16x8 hn[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) hn[i] = __builtin_bit_cast(bf16x8, q[d][i]);
        if (ks + D < 32) {
#pragma unroll
          for (int i = 0; i < 2; ++i) q[d][i] = bp[(long)(ks + D) * 1024 + i * 64];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hn[i], hm[j], acc[i][j], 0, 0, 0);  // D[n][m]
      }
    }
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s = 0; s < 2; ++s)
          hm2[i][j][s] = *(const bf16x8*)(smem + (j * 32 + 2 * (2 * wave + i) + s) * 1024 + lane * 16);
  }

  // chunk (m-block j, column block 2 wave + i, half s) of an F16 [M, 512] tensor, this lane's 16 bytes
  auto chunk = [&](int i, int j, int s) -> long {
    return ((((tile * 2 + j) * 16 + 2 * wave + i) * 2 + s) * 64 + lane) * 8;   // bf16 index
  };

  if constexpr (MODE == R_GEMM || MODE == R_GEMM_ACT) {
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
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          float v[8], w[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = acc[i][j][8 * s + e] + bv[8 * s + e];
          if constexpr (MODE == R_GEMM_ACT) {
#pragma unroll
            for (int e = 0; e < 8; ++e) sin_act(v[e], v[e], w[e]);
            *(u32x4*)(p.O2 + chunk(i, j, s)) = pack8(w);
          }
          *(u32x4*)(p.O1 + chunk(i, j, s)) = pack8(v);
        }
    }
    return;
  }

  if constexpr (MODE == R_CHAIN) {
    // [store t2]; gate by d1; the gated accumulators, rounded to bf16, ARE the next contraction's operand fragments
    // (lane (m, lhi) holds 8 k = n values per half s: the F16 fragment again)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
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
        }
    __syncthreads();   // every wave is done with the A tile: the reduction buffer overlays it
  }
  if constexpr (MODE == R_THINN) __syncthreads();

  // ---- stage 2: out[m][jp] = sum over this wave's 64 k of u[m][k] Wp[k][jp]; 8-way reduction over the waves through LDS
  float* red = (float*)smem;   // [8 waves][64 m][32 jp]
  const u32x4* pp = (const u32x4*)p.Pp + lane;
#pragma unroll 1
  for (int jb = 0; jb < JB; ++jb) {
    f32x16 out[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) out[j][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 hp = __builtin_bit_cast(bf16x8, pp[((long)(2 * (2 * wave + i) + s) * JB + jb) * 64]);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          out[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hm2[i][j][s], hp, out[j], 0, 0, 0);   // D[m][jp]
      }
    if (jb > 0) __syncthreads();   // the previous round's reduction has read `red`
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = j * 32 + 8 * (r >> 2) + (r & 3) + 4 * lhi;
        red[(wave * R_TM + m) * 32 + l31] = out[j][r];
      }
    __syncthreads();
    {  // 64 x 32 outputs, 4 per thread: thread -> (m = tid / 8, jp = (tid % 8) * 4 .. +3)
      const int m = tid >> 3, j0 = (tid & 7) * 4;
      float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int wv = 0; wv < R_NW; ++wv) {
        const float4 v = *(const float4*)(red + (wv * R_TM + m) * 32 + j0);
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

template <int MODE, int JB, bool T2OUT>
int launch_rows(const RowsArgs& a, hipStream_t st) {
  static bool ready = false;
  if (!ready) {
    if (hipFuncSetAttribute((const void*)f16_rows_kernel<MODE, JB, T2OUT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            R_LDS_BYTES) != hipSuccess)
      return INDM_E_LAUNCH;
    ready = true;
  }
  hipLaunchKernelGGL((f16_rows_kernel<MODE, JB, T2OUT>), dim3(a.M / R_TM), dim3(R_NW * 64), R_LDS_BYTES, st, a);
  return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
}

// ---------------------------------------------------------------------------------------------------------------
// f16_thin_conv_kernel: out[pixel, co] = sum_{tap, c} x[pixel shifted by tap, c] * w[(tap, c), co], co < 512, C <= 12.
// 128 pixels x 128 columns per tile, 4 waves (wave w: pixels 32 w .. 32 w + 31, four 32-column blocks), persistent over
// the pixel tiles of one column tile (its weights are packed into LDS once).  K = 9 C is padded to S = ceil(K / 16)
// k-steps; the operand fragments use consecutive k per lane (k = 16 s + 8 lhi + e) on both sides.  A lane GATHERS the
// fragment it multiplies (lane l31 = its pixel, its 8 k per step): the im2col values never pass through LDS, and the
// next tile's gather is in flight under this tile's MFMAs and stores.
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

// SMAX: k-steps the instantiation holds registers for; VEC4: C % 4 == 0, four consecutive k share a tap (float4 loads)
template <int EPI, int SMAX, bool VEC4>
__global__ __launch_bounds__(256, 2) void f16_thin_conv_kernel(ThinConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) char tc_smem[];
  u16* wf = (u16*)tc_smem;   // weights [s][nblk][lane][8]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int C = p.C, K = 9 * C, S = (K + 15) >> 4;
  const int tiles_n = p.Cout / TC_TN;
  const long M = (long)p.N * p.H * p.W;
  const int tiles_m = (int)(M / TC_TM);
  const int tn = blockIdx.x % tiles_n, mstep = gridDim.x / tiles_n;
  const int n0 = tn * TC_TN;
  const int HW = p.H * p.W;

  // ---- weights of this column tile -> fragment order (A role: lane l31 = output column, k = 16 s + 8 lhi + e)
  for (int u = tid; u < S * 4 * 64; u += 256) {
    const int ln = u & 63, nb = (u >> 6) & 3, s = u >> 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 16 * s + 8 * (ln >> 5) + e;
      v[e] = k < K ? p.Wt[(long)k * p.Cout + n0 + nb * 32 + (ln & 31)] : 0.f;
    }
    *(u32x4*)(wf + u * 8) = pack8(v);
  }
  float bv[4][16];
#pragma unroll
  for (int jn = 0; jn < 4; ++jn)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias) b = *(const float4*)(p.bias + n0 + jn * 32 + 8 * a + 4 * lhi);
      bv[jn][4 * a] = b.x; bv[jn][4 * a + 1] = b.y; bv[jn][4 * a + 2] = b.z; bv[jn][4 * a + 3] = b.w;
    }

  // ---- per-thread tap table (tile-invariant): for load unit q (one k, or four k of one tap) the element offset
  // relative to the pixel's own element offset, and (dh + 1) | (dw + 1) << 2 | exists << 4
  constexpr int NU = VEC4 ? SMAX * 2 : SMAX * 8;
  int toff[NU], tmeta[NU];
#pragma unroll
  for (int q = 0; q < NU; ++q) {
    const int k = VEC4 ? 16 * (q >> 1) + 8 * lhi + 4 * (q & 1) : 16 * (q >> 3) + 8 * lhi + (q & 7);
    const int tap = k / C, c = k - tap * C;
    const int kh = tap / 3, kw = tap - 3 * kh;
    const int dh = -p.sign * (kh - 1), dw = -p.sign * (kw - 1);
    toff[q] = (dh * p.W + dw) * C + c;
    tmeta[q] = k < K ? ((dh + 1) | ((dw + 1) << 2) | 16) : 0;
  }
  auto gather = [&](int tm, float (&gv)[SMAX][8]) {
    const long pix = (long)tm * TC_TM + wave * 32 + l31;
    const int img = (int)(pix / HW), r = (int)(pix - (long)img * HW);
    const int h = r / p.W, w = r - h * p.W;
    const float* base = p.X + pix * C;
#pragma unroll
    for (int q = 0; q < NU; ++q) {
      const int hh = h + (tmeta[q] & 3) - 1, ww = w + ((tmeta[q] >> 2) & 3) - 1;
      const bool ok = (tmeta[q] & 16) && hh >= 0 && hh < p.H && ww >= 0 && ww < p.W;
      if constexpr (VEC4) {
        const float4 v = ok ? *(const float4*)(base + toff[q]) : make_float4(0.f, 0.f, 0.f, 0.f);
        gv[q >> 1][4 * (q & 1)] = v.x; gv[q >> 1][4 * (q & 1) + 1] = v.y;
        gv[q >> 1][4 * (q & 1) + 2] = v.z; gv[q >> 1][4 * (q & 1) + 3] = v.w;
      } else {
        gv[q >> 3][q & 7] = ok ? base[toff[q]] : 0.f;
      }
    }
  };

  float gv[SMAX][8];
  int tm = blockIdx.x / tiles_n;
  if (tm < tiles_m) gather(tm, gv);
  __syncthreads();   // wf is complete
  for (; tm < tiles_m; tm += mstep) {
    bf16x8 hx[SMAX];
#pragma unroll
    for (int s = 0; s < SMAX; ++s) hx[s] = __builtin_bit_cast(bf16x8, pack8(gv[s]));
    const int tnext = tm + mstep;
    if (tnext < tiles_m) gather(tnext, gv);   // in flight under the MFMAs and the stores of this tile
    f32x16 acc[4];
#pragma unroll
    for (int jn = 0; jn < 4; ++jn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[jn][r] = 0.f;
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
    // ---- epilogue: acc[jn][r] = out[pixel = 32 (4 tm + wave) + l31][co = n0 + 32 jn + (r & 3) + 8 (r >> 2) + 4 lhi]
    const long mt = (long)tm * 4 + wave;
    const int nbn = p.Cout >> 5;
#pragma unroll
    for (int jn = 0; jn < 4; ++jn)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const long ci = (((mt * nbn + (n0 >> 5) + jn) * 2 + s) * 64 + lane) * 8;
        float v[8], w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = acc[jn][8 * s + e] + bv[jn][8 * s + e];
        if constexpr (EPI == TC_ACT) {
#pragma unroll
          for (int e = 0; e < 8; ++e) sin_act(v[e], v[e], w[e]);
          *(u32x4*)(p.O2 + ci) = pack8(w);
        } else if constexpr (EPI == TC_GATE) {
          if (p.O2) *(u32x4*)(p.O2 + ci) = pack8(v);
          unpack8(*(const u32x4*)(p.G + ci), w);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= w[e];
        }
        *(u32x4*)(p.O1 + ci) = pack8(v);
      }
  }
}

template <int EPI, int SMAX, bool VEC4>
int launch_thin_conv(const ThinConvArgs& a, dim3 grid, hipStream_t st) {
  constexpr int bytes = SMAX * 4 * 64 * 8 * 2;
  hipLaunchKernelGGL((f16_thin_conv_kernel<EPI, SMAX, VEC4>), grid, dim3(256), bytes, st, a);
  return hipGetLastError() == hipSuccess ? INDM_OK : INDM_E_LAUNCH;
}
template <int EPI>
int launch_thin_conv_e(const ThinConvArgs& a, dim3 grid, hipStream_t st) {
  if (a.C <= 3) return launch_thin_conv<EPI, 2, false>(a, grid, st);
  return launch_thin_conv<EPI, 7, true>(a, grid, st);
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
  int groups = 512 / tiles_n;   // two workgroups per CU
  if (groups > tiles_m) groups = tiles_m;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(groups * tiles_n);
  if (epilogue == TC_PLAIN) return launch_thin_conv_e<TC_PLAIN>(a, grid, st);
  if (epilogue == TC_ACT) return launch_thin_conv_e<TC_ACT>(a, grid, st);
  if (epilogue == TC_GATE) return launch_thin_conv_e<TC_GATE>(a, grid, st);
  return INDM_E_ARG;
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
    case R_CHAIN:
      if (!Bp || !D1 || !Pp || !P || J < 1 || J > 128) return INDM_E_ARG;
      if (J <= 32) return out1 ? launch_rows<R_CHAIN, 1, true>(a, st) : launch_rows<R_CHAIN, 1, false>(a, st);
      return out1 ? launch_rows<R_CHAIN, 4, true>(a, st) : launch_rows<R_CHAIN, 4, false>(a, st);
    case R_THINN:
      if (!Pp || !P || J < 1 || J > 128) return INDM_E_ARG;
      return J <= 32 ? launch_rows<R_THINN, 1, false>(a, st) : launch_rows<R_THINN, 4, false>(a, st);
    default:
      return INDM_E_ARG;
  }
}

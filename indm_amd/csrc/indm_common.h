// Shared device/host helpers for the INDM gfx950 kernel library.
// Everything here is fp32 unless stated; wavefront = 64 lanes (CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define INDM_OK 0
#define INDM_E_SHAPE (-1)
#define INDM_E_DTYPE (-2)
#define INDM_E_LAUNCH (-3)
#define INDM_E_ARG (-4)

#define INDM_API extern "C" __attribute__((visibility("default")))

#define INDM_CHECK_LAUNCH()                         \
  do {                                              \
    hipError_t e_ = hipGetLastError();              \
    if (e_ != hipSuccess) return INDM_E_LAUNCH;     \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 indm_bf16x8 __attribute__((ext_vector_type(8)));
// 8 floats -> 8 bf16 (RNE) as ONE vector conversion: four v_cvt_pk_bf16_f32.  Element-by-element casts into a bf16x8
// make hipcc pair (x), (y, z), (w) and repair the lanes with v_perm_b32 / v_alignbit_b32: nine instructions per eight
// values, in loops whose VALU work is what the MFMAs wait for (round 5).  Same rounding, same bits.
__device__ __forceinline__ indm_bf16x8 indm_pack_bf16x8(float a, float b, float c, float d, float e, float f, float g, float h) {
  const f32x8 v = {a, b, c, d, e, f, g, h};
  return __builtin_convertvector(v, indm_bf16x8);
}

// sin / cos of t (radians) for the flow's Sin activation a = sin(2 pi p) / (2 pi), d = cos(2 pi p), with t = fl(2 pi p) -- the
// argument the reference hands to torch.sin / torch.cos (flows/resflow/layers/base/activations.py:7-12).  Branch-free:
// quadrant k = rint(t * 2 / pi), three-term Cody-Waite reduction r = t - k pi/2 (the products are exact for |k| < 2^13,
// i.e. |p| < 2000), the Cephes sinf / cosf kernels on [-pi/4, pi/4]: <= 1.5 ulp of sin(t) / cos(t) for |p| <= 100 (measured
// against fp64 on 2 M normal samples per scale; libm's sincosf is in the same error class but costs ~3x the instructions, has
// a slow path behind branches and spilled 14 registers inside the GEMM epilogues).  Every Sin site of the library calls this
// one function (the fused epilogues stay bit-identical to the two-step form); -ffp-contract=off keeps the evaluation as written.
__device__ __forceinline__ void indm_sincos(float t, float& s, float& c) {
  const float k = __builtin_rintf(t * 0.63661977236758134308f);
  const float r = ((t - k * 1.5703125f) - k * 4.837512969970703125e-4f) - k * 7.549789948768648e-8f;
  const float z = r * r;
  const float sr = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z + -1.6666654611e-1f) * z * r + r;
  const float cr = ((2.443315711809948e-5f * z + -1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
  const int q = (int)k;
  const float s0 = (q & 1) ? cr : sr, c0 = (q & 1) ? sr : cr;
  s = (q & 2) ? -s0 : s0;
  c = ((q + 1) & 2) ? -c0 : c0;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Opt-in for more than 64 KB of dynamic LDS (hipFuncAttributeMaxDynamicSharedMemorySize), once per kernel AND per device:
// the attribute belongs to the function on the CURRENT device, so a process-wide flag would skip the opt-in on a second
// GPU of the same process.  The size only grows; racing threads repeat an idempotent call.
template <auto Kernel>
static inline int indm_lds_optin(size_t bytes) {
  static size_t ready[32] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return INDM_E_LAUNCH;
  const bool tracked = dev >= 0 && dev < 32;
  if (tracked && __atomic_load_n(&ready[dev], __ATOMIC_RELAXED) >= bytes) return INDM_OK;
  if (hipFuncSetAttribute((const void*)Kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
    return INDM_E_LAUNCH;
  if (tracked) __atomic_store_n(&ready[dev], bytes, __ATOMIC_RELAXED);
  return INDM_OK;
}

// Grid size for grid-stride HBM-bound kernels: enough workgroups to fill
// 256 CUs x 8 blocks, no more (cdna guide, Guideline 11).
static inline int ew_grid(long n_items, int block) {
  long g = (n_items + block - 1) / block;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum for blockDim.x a multiple of 64 (<= 1024); `sh` holds >= 16 floats.
__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  float r = (lane < nw) ? sh[lane] : 0.f;
  r = wave_sum(r);
  return r;
}
__device__ __forceinline__ float block_max(float v, float* sh) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  float r = (lane < nw) ? sh[lane] : -INFINITY;
  r = wave_max(r);
  return r;
}

// Counter-based RNG (Philox4x32-10) so dropout masks / noise can be regenerated
// in the backward pass from (seed, offset) instead of being stored in HBM.
__device__ __forceinline__ uint4 philox4x32(uint2 key, uint4 ctr) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = __umulhi(M0, ctr.x), lo0 = M0 * ctr.x;
    uint32_t hi1 = __umulhi(M1, ctr.z), lo1 = M1 * ctr.z;
    ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
    key.x += W0;
    key.y += W1;
  }
  return ctr;
}
__device__ __forceinline__ float u32_to_unit(uint32_t x) {  // (0,1]
  return ((float)(x >> 8) + 1.0f) * (1.0f / 16777216.0f);
}

// Force kernel arguments into SGPRs at the point of the macro (an empty asm that "modifies" them: the compiler must have
// loaded them before it and cannot re-load them from the kernarg segment after it).  hipcc loads an argument with a scalar
// load where it is first needed, which in a kernel with a long first phase is many microseconds after the launch.
#define INDM_PIN1(a) asm volatile("" : "+s"(a))
#define INDM_PIN_ARGS9(a, b, c, d, e, f, g, h, i) \
  do { INDM_PIN1(a); INDM_PIN1(b); INDM_PIN1(c); INDM_PIN1(d); INDM_PIN1(e); INDM_PIN1(f); INDM_PIN1(g); INDM_PIN1(h); INDM_PIN1(i); } while (0)

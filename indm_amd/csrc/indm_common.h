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

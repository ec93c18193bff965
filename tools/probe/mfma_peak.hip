// Pure-MFMA ceiling probe (standalone; run on the GPU box): v_mfma_f32_32x32x2_f32 (and the bf16 32x32x16 form) with
// NO memory traffic in the loop, >= 1 ms per launch -- what the fp32 matrix pipe sustains on this part under a
// full-chip load, i.e. the ceiling the MFMA-bound kernels of this library are priced against.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip     run: ./mfma_peak   (rocprofv3 --kernel-trace --stats -- ./mfma_peak)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));

// ACC independent accumulator chains per wave (back-to-back dependent MFMAs stall 16 passes: 4+ chains hide that)
template <int ACC>
__global__ __launch_bounds__(256) void mfma_f32_loop(float* out, int iters, float a0, float b0) {
  f16v acc[ACC];
#pragma unroll
  for (int i = 0; i < ACC; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < ACC; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 123.456f) out[0] = s;   // never true: keeps the chain alive
}

// the same loop on RANDOM operands that change with every MFMA (eight register pairs of hashed values in [-1, 1)):
// operand toggling is what the power limit responds to -- this is the clock the MFMA-bound kernels actually see
template <int ACC>
__global__ __launch_bounds__(256) void mfma_f32_loop_rnd(float* out, int iters, unsigned seed) {
  f16v acc[ACC];
#pragma unroll
  for (int i = 0; i < ACC; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float a[8], b[8];
  unsigned h = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    h = h * 1664525u + 1013904223u; a[u] = (float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
    h = h * 1664525u + 1013904223u; b[u] = (float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + i) & 7], b[(u + 3 * i) & 7], acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < ACC; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 123.456f) out[0] = s;
}

template <int ACC>
__global__ __launch_bounds__(256) void mfma_bf16_loop(float* out, int iters, float a0) {
  f16v acc[ACC];
#pragma unroll
  for (int i = 0; i < ACC; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  bf8v a, b;
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(a0 + j); b[j] = (__bf16)(0.5f + j); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < ACC; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 123.456f) out[0] = s;
}

// round 6: the SHAPE of f16_rows_kernel's launch without any of its memory traffic -- 2048 workgroups of 512 threads with 64 KB of
// dynamic LDS each (two per CU), 128 MFMAs on four accumulators per wave: what workgroup granularity alone costs
__global__ __launch_bounds__(512, 4) void mfma_bf16_tile(float* out, int mfmas, float a0) {
  extern __shared__ char smem[];
  f16v acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  bf8v a, b;
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(a0 + j + threadIdx.x * 1e-3f); b[j] = (__bf16)(0.5f + j); }
  for (int it = 0; it < mfmas / 4; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 123.456f) { out[0] = s; smem[threadIdx.x] = 1; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <typename F>
static void run(const char* name, F launch, double flop_per_launch, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch();                              // warm-up (clock ramp)
  CK(hipDeviceSynchronize());
  double best = 1e30, sum = 0;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best; sum += ms;
  }
  printf("%-44s %8.3f ms avg %8.3f ms best  %8.1f TFLOP/s avg %8.1f best\n", name, sum / reps, best,
         flop_per_launch / (sum / reps) * 1e-9, flop_per_launch / best * 1e-9);
}

int main(int argc, char** argv) {
  int reps = argc > 1 ? atoi(argv[1]) : 20;
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount;
  printf("device %s, %d CUs, clock %d MHz\n", p.gcnArchName, cus, p.clockRate / 1000);
  float* out; CK(hipMalloc(&out, 64));
  // sustained: ~10 ms launches back to back (power / clock settle); short: ~1.5 ms
  for (int wpc = 1; wpc <= 2; ++wpc) {          // workgroups of 4 waves per CU: 1 -> one wave per SIMD, 2 -> two
    int blocks = cus * wpc;
    for (int iters : {4000, 40000}) {
      double mf = (double)blocks * 4 /*waves*/ * iters * 8 * 4 /*ACC*/;
      char nm[96];
      snprintf(nm, sizeof nm, "fp32 32x32x2  %d wave/SIMD %6d iters", wpc, iters);
      run(nm, [&] { mfma_f32_loop<4><<<blocks, 256>>>(out, iters, 1.0f, 0.5f); }, mf * 2.0 * 32 * 32 * 2, reps);
      snprintf(nm, sizeof nm, "fp32 32x32x2  %d wave/SIMD %6d iters RANDOM", wpc, iters);
      run(nm, [&] { mfma_f32_loop_rnd<4><<<blocks, 256>>>(out, iters, 12345u); }, mf * 2.0 * 32 * 32 * 2, reps);
      snprintf(nm, sizeof nm, "bf16 32x32x16 %d wave/SIMD %6d iters", wpc, iters);
      run(nm, [&] { mfma_bf16_loop<4><<<blocks, 256>>>(out, iters, 1.0f); }, mf * 2.0 * 32 * 32 * 16, reps);
    }
  }
  CK(hipFuncSetAttribute((const void*)mfma_bf16_tile, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  for (int wgs : {2048, 512}) {
    for (int mf : {128, 512, 2048}) {
      char nm[96];
      snprintf(nm, sizeof nm, "bf16 tile-shaped: %4d WGs x 512 thr, %4d MFMA/wave", wgs, mf);
      run(nm, [&] { mfma_bf16_tile<<<wgs, 512, 65536>>>(out, mf, 1.0f); }, (double)wgs * 8 * mf * 2.0 * 32 * 32 * 16, reps);
    }
  }
  CK(hipFree(out));
  return 0;
}

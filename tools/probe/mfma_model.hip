// What costs an MFMA-bound tile loop its last 15-20 %?  (standalone probe; run on the GPU box)
// The loop of the library's direct-to-LDS kernels -- 8 waves per workgroup, 2 workgroups per CU, 32 x v_mfma_f32_32x32x2
// per wave and 16-deep slab on 4 accumulators -- rebuilt from nothing, one ingredient at a time:
//   bit 0: +40 independent VALU instructions per slab (gate multiplies, address updates)
//   bit 1: + one workgroup barrier per slab
//   bit 2: + the slab's LDS fragment reads (8 x ds_read_b128 + 8 x ds_read2_b32 per wave) feeding the MFMAs
//   bit 3: + the slab's direct-to-LDS copies (5 x global_load_lds_dwordx4 per wave, 40 KB per workgroup) with
//          s_waitcnt vmcnt(0) in front of the barrier
//   bit 4: the copies share ONE address register and M0 value (instruction offsets 0 / 1024 / 2048 / 3072 / 4096 bytes)
//   bit 5: wait with vmcnt(5): only for the PREVIOUS slab's copies (a three-buffer rotation)
//   bit 6: gate multiplies as 2-wide packed math
//   bit 8: TILES of 32 slabs: after every 32nd slab the 64 accumulators of a lane are stored to HBM (a 256 x 128 fp32 tile per
//          workgroup, 16 dword stores of 128-byte rows per lane and block) and cleared, and the next tile's first copy is NOT
//          in flight yet (its latency is exposed, as in a kernel launched per tile)
//   bit 9: with bit 8: the next tile's first slab is requested during the last slab of the current one (persistent kernel)
//   bit 7: five copies with five DIFFERENT source rows (as the 256 x 128 tile kernels need), addressed as
//          `voffset, s[base:base+1]` through inline asm: per-lane offsets fixed, the uniform bases advance on the scalar unit
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_model mfma_model.hip      run: ./mfma_model
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

constexpr int BUF = 10240;  // floats per LDS buffer (40 KB), two buffers per workgroup

template <int F>
__global__ __launch_bounds__(512, 4) void model(const float* __restrict__ src, float* out, int nslab) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  f16v acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float va[8], vb[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) { va[u] = 1.0f + lane * 1e-3f + u; vb[u] = 0.5f + u; }
  float fa[8] = {1.f, 1.01f, 1.02f, 1.03f, 1.04f, 1.05f, 1.06f, 1.07f};
  float fb[8] = {.5f, .51f, .52f, .53f, .54f, .55f, .56f, .57f};
  for (int i = tid; i < 2 * BUF; i += 512) smem[i] = 1.0f + (i & 15) * 0.01f;
  __syncthreads();
  const float* ps = src + ((long)blockIdx.x * 512 + tid) * 4;
  // the workgroup's [256][128] fp32 tile image: wave (wm, wn) owns rows 64 wm.., columns 64 wn..; a store instruction writes
  // two 128-byte row segments (lanes 0-31 / 32-63), as the kernels' epilogue does
  float* tile_out = out + 64 + (long)blockIdx.x * 256 * 128 + ((wave >> 1) * 64 + 4 * (lane >> 5)) * 128 + (wave & 1) * 64 + (lane & 31);
  for (int s = 0; s < nslab; ++s) {
    const int buf = s & 1;
    if constexpr ((F & 8) && (F & 256)) {
      if (!(F & 512) && (s & 31) == 0 && s > 0) {   // tile start without prefetch: this slab's copy was not issued ahead
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
    }
    if constexpr (F & 8) {
      float* dst = smem + (buf ^ 1) * BUF + wave * 1280;   // 5 x 1 KB per wave
      if constexpr (F & 128) {
        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)dst;
        const unsigned voff = (unsigned)tid * 16u;
        const unsigned long long sb = (unsigned long long)(src + (long)blockIdx.x * 2048) + (unsigned long long)s * 64ull;
#pragma unroll
        for (int q = 0; q < 5; ++q)
          asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                       :: "s"(lds0 + q * 1024u), "v"(voff), "s"(sb + (unsigned long long)q * 8192ull) : "memory");
      } else if constexpr (F & 16) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ps, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ps, (__attribute__((address_space(3))) void*)dst, 16, 1024, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ps, (__attribute__((address_space(3))) void*)dst, 16, 2048, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ps, (__attribute__((address_space(3))) void*)dst, 16, 3072, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ps + 1024), (__attribute__((address_space(3))) void*)(dst + 1024), 16, 0, 0);
      } else {
#pragma unroll
      for (int q = 0; q < 5; ++q)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ps + q * 2048),
                                         (__attribute__((address_space(3))) void*)(dst + q * 256), 16, 0, 0);
      }
      ps += 16;
    }
    if constexpr (F & 4) {
      const float* as = smem + buf * BUF;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f4v a0 = *(const f4v*)(as + (lane & 31) * 16 + h * 8 + (lane >> 5) * 4);
        f4v g0 = *(const f4v*)(as + 1024 + (lane & 31) * 16 + h * 8 + (lane >> 5) * 4);
        f4v a1 = *(const f4v*)(as + 512 + (lane & 31) * 16 + h * 8 + (lane >> 5) * 4);
        f4v g1 = *(const f4v*)(as + 1536 + (lane & 31) * 16 + h * 8 + (lane >> 5) * 4);
        f4v u0, u1;
        if constexpr (F & 64) {
          typedef float f2v __attribute__((ext_vector_type(2)));
          f2v p0 = __builtin_shufflevector(a0, a0, 0, 1) * __builtin_shufflevector(g0, g0, 0, 1);
          f2v p1 = __builtin_shufflevector(a0, a0, 2, 3) * __builtin_shufflevector(g0, g0, 2, 3);
          f2v p2 = __builtin_shufflevector(a1, a1, 0, 1) * __builtin_shufflevector(g1, g1, 0, 1);
          f2v p3 = __builtin_shufflevector(a1, a1, 2, 3) * __builtin_shufflevector(g1, g1, 2, 3);
          u0 = __builtin_shufflevector(p0, p1, 0, 1, 2, 3);
          u1 = __builtin_shufflevector(p2, p3, 0, 1, 2, 3);
        } else {
          u0 = a0 * g0; u1 = a1 * g1;
        }
        fa[4 * h + 0] = u0.x; fa[4 * h + 1] = u0.y; fa[4 * h + 2] = u0.z; fa[4 * h + 3] = u0.w;
        fb[4 * h + 0] = u1.x; fb[4 * h + 1] = u1.y; fb[4 * h + 2] = u1.z; fb[4 * h + 3] = u1.w;
        const float* q = as + 2048 + (h * 8 + (lane >> 5) * 4) * 512 + wave * 64 + (lane & 31);
        va[4 * h + 0] = q[0]; va[4 * h + 1] = q[512]; va[4 * h + 2] = q[1024]; va[4 * h + 3] = q[1536];
        vb[4 * h + 0] = q[32]; vb[4 * h + 1] = q[512 + 32]; vb[4 * h + 2] = q[1024 + 32]; vb[4 * h + 3] = q[1536 + 32];
      }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[t], fa[t], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[t], fb[t], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(vb[t], fa[t], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(vb[t], fb[t], acc[3], 0, 0, 0);
      if constexpr ((F & 1) && !(F & 4)) {   // 40 VALU per slab = 5 per 4 MFMAs (with bit 2 the reads bring their own)
        va[t] = va[t] * 1.0001f; vb[t] = vb[t] * 0.9999f; fa[t] = fa[t] + 1e-7f; fb[t] = fb[t] - 1e-7f;
        va[(t + 1) & 7] = va[(t + 1) & 7] + 1e-9f;
      }
    }
    if constexpr ((F & 8) && (F & 32)) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (F & 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (F & 2) __syncthreads();
    if constexpr (F & 256) {
      if ((s & 31) == 31) {   // tile epilogue: 4 blocks x 16 row stores per lane, then clear
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            tile_out[((i >> 1) * 32 + 8 * (j >> 2) + (j & 3)) * 128 + (i & 1) * 32] = acc[i][j];
            acc[i][j] = 0.f;
          }
        if constexpr (!(F & 512)) {   // per-tile launch: the next tile's first copy starts only now
          if constexpr (F & 8) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
        }
      }
    }
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) r += acc[i][j];
  if (r == 123.456f) out[0] = r;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int F>
static void run(const char* name, const float* src, float* out, int blocks, int nslab) {
  CK(hipFuncSetAttribute((const void*)model<F>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) model<F><<<blocks, 512, 2 * BUF * 4>>>(src, out, nslab);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int reps = 20;
  for (int r = 0; r < reps; ++r) model<F><<<blocks, 512, 2 * BUF * 4>>>(src, out, nslab);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double flop = (double)blocks * 8 * nslab * 32 * 2.0 * 32 * 32 * 2;
  printf("%-64s %8.3f ms  %7.1f TFLOP/s\n", name, ms, flop / ms * 1e-9);
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int blocks = 2 * p.multiProcessorCount, nslab = 128;   // 4 tiles' worth of 32 slabs per workgroup
  float *src, *out;
  const size_t n = ((size_t)blocks * 512 * 4 + (size_t)nslab * 16 + 5 * 2048 + 64);
  CK(hipMalloc(&src, n * 4)); CK(hipMemset(src, 0, n * 4)); CK(hipMalloc(&out, (size_t)(64 + (size_t)blocks * 256 * 128 + 4096) * 4));
  run<0>("MFMA only (8 waves/WG, 2 WG/CU, 32 MFMA/slab)", src, out, blocks, nslab);
  run<1>("+ 40 VALU per slab", src, out, blocks, nslab);
  run<2>("+ barrier per slab", src, out, blocks, nslab);
  run<3>("+ VALU + barrier", src, out, blocks, nslab);
  run<4>("+ LDS fragment reads (+ gate multiplies)", src, out, blocks, nslab);
  run<6>("+ LDS reads + barrier", src, out, blocks, nslab);
  run<8>("+ direct-to-LDS copies + vmcnt(0)", src, out, blocks, nslab);
  run<10>("+ copies + vmcnt(0) + barrier", src, out, blocks, nslab);
  run<14>("+ LDS reads + copies + vmcnt(0) + barrier  (the real loop)", src, out, blocks, nslab);
  run<14 + 16>("real loop, copies through instruction offsets", src, out, blocks, nslab);
  run<14 + 32>("real loop, vmcnt(5): wait for the previous slab only", src, out, blocks, nslab);
  run<14 + 16 + 32>("real loop, instruction offsets + vmcnt(5)", src, out, blocks, nslab);
  run<14 + 64>("real loop, packed gate multiplies", src, out, blocks, nslab);
  run<14 + 16 + 32 + 64>("real loop, all three", src, out, blocks, nslab);
  run<14 + 128>("real loop, five distinct rows through `voffset, s[base]` asm copies", src, out, blocks, nslab);
  run<8 + 2 + 128>("copies (asm saddr form) + vmcnt(0) + barrier", src, out, blocks, nslab);
  run<14 + 128 + 256>("real loop (asm copies) in TILES of 32 slabs: stores + exposed first copy", src, out, blocks, nslab);
  run<14 + 128 + 256 + 512>("same, next tile's first slab prefetched (persistent)", src, out, blocks, nslab);
  run<8 + 16>("copies only (instruction offsets) + vmcnt(0)", src, out, blocks, nslab);
  run<8 + 2 + 16 + 32>("copies (instruction offsets) + vmcnt(5) + barrier", src, out, blocks, nslab);
  return 0;
}

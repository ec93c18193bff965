// Does `buffer_load_dwordx4 ... offen lds` (gfx950) write ZEROS into LDS for lanes whose offset is out of range?
// (standalone probe; run on the GPU box)  That is what makes a padded convolution tap free: no select on a 64-bit
// address, no zero page -- bit 31 of the lane's offset says "padding".
// build: hipcc --offload-arch=gfx950 -O3 -o buffer_lds buffer_lds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(const float* src, float* out, int nbytes, unsigned soff) {
  extern __shared__ float sm[];
  for (int i = threadIdx.x; i < 1024; i += 64) sm[i] = -7.f;   // poison
  __syncthreads();
  v4i rsrc;
  const unsigned long long base = (unsigned long long)src;
  rsrc.x = (int)(base & 0xffffffffu);
  rsrc.y = (int)((base >> 32) & 0xffff);   // stride 0: raw buffer
  rsrc.z = nbytes;                          // num_records
  rsrc.w = 0x00020000;                      // DATA_FORMAT_32 (gfx90a / gfx94x / gfx950 raw-buffer word 3)
  unsigned voff = threadIdx.x * 16u;
  if (threadIdx.x % 5 == 3) voff |= 0x80000000u;   // "padding" lanes
  const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)sm;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\t"
               "buffer_load_dwordx4 %1, %2, %3 offen offset:1024 lds"
               :: "s"(lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = sm[i];
}
int main() {
  float *src, *out;
  const int n = 4096;
  hipMalloc(&src, n * 4); hipMalloc(&out, 512 * 4);
  float h[4096];
  for (int i = 0; i < n; ++i) h[i] = (float)i;
  hipMemcpy(src, h, n * 4, hipMemcpyHostToDevice);
  k<<<1, 64, 4096 * 4>>>(src, out, n * 4, 64u);   // soffset 64 bytes = 16 floats
  float r[512];
  hipMemcpy(r, out, 512 * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane)
    for (int rep = 0; rep < 2; ++rep)               // second copy: +1024 bytes in global AND LDS
      for (int e = 0; e < 4; ++e) {
        const float want = (lane % 5 == 3) ? 0.f : (float)(lane * 4 + e + 16 + rep * 256);
        const float got = r[rep * 256 + lane * 4 + e];
        if (got != want) { if (bad < 8) printf("lane %d rep %d e %d: got %g want %g\n", lane, rep, e, got, want); ++bad; }
      }
  printf(bad ? "MISMATCH: %d\n" : "buffer_load_dwordx4 lds: in-range lanes copy, out-of-range lanes write zeros, soffset and the immediate apply: OK (%d)\n", bad);
  return bad != 0;
}

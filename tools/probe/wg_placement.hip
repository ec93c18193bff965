// Which workgroups of a 1-D grid share a CU?  (standalone probe; run on the GPU box)  Every workgroup of a 512 x 512-thread
// launch with 80 KB of LDS (the footprint of the library's direct-to-LDS kernels: two per CU) records XCC_ID and the
// HW_ID fields of its first wave, and its start / end clock.  Prints the co-residency pattern.
// build: hipcc --offload-arch=gfx950 -O3 -o wg_placement wg_placement.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ __launch_bounds__(512) void probe(unsigned* out, int spin) {
  extern __shared__ float sm[];
  unsigned hw = __builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (31 << 11));
  unsigned xcc = __builtin_amdgcn_s_getreg((20 /*XCC_ID*/) | (0 << 6) | (31 << 11));
  unsigned long long t0 = __builtin_readcyclecounter();
  sm[threadIdx.x] = (float)threadIdx.x;
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(100);
  __syncthreads();
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) {
    out[blockIdx.x * 4 + 0] = hw;
    out[blockIdx.x * 4 + 1] = xcc;
    out[blockIdx.x * 4 + 2] = (unsigned)(t0 >> 8);
    out[blockIdx.x * 4 + 3] = (unsigned)(t1 >> 8);
  }
}
int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 512;
  unsigned* d;
  hipMalloc(&d, blocks * 16);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  probe<<<blocks, 512, 80 * 1024>>>(d, 2000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(blocks * 4);
  hipMemcpy(h.data(), d, blocks * 16, hipMemcpyDeviceToHost);
  // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]...
  std::map<unsigned, std::vector<int>> by_cu;
  for (int b = 0; b < blocks; ++b) {
    const unsigned hw = h[b * 4], xcc = h[b * 4 + 1] & 0xf;
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    by_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(b);
  }
  printf("%d workgroups on %zu distinct (xcc, se, sh, cu) slots\n", blocks, by_cu.size());
  int shown = 0;
  std::map<int, int> delta;
  for (auto& kv : by_cu) {
    if (shown++ < 12) {
      printf("  xcc %u se %u sh %u cu %2u :", kv.first >> 12, (kv.first >> 8) & 7, (kv.first >> 4) & 1, kv.first & 0xf);
      for (int b : kv.second) printf(" %4d", b);
      printf("\n");
    }
    if (kv.second.size() == 2) delta[kv.second[1] - kv.second[0]]++;
  }
  printf("blockIdx difference of the two co-resident workgroups (histogram):");
  for (auto& kv : delta) printf("  %+d: %d", kv.first, kv.second);
  printf("\nfirst 16 workgroups -> xcc:");
  for (int b = 0; b < 16; ++b) printf(" %u", h[b * 4 + 1] & 0xf);
  printf("\n");
  return 0;
}

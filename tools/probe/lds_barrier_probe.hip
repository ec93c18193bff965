// LDS / barrier integrity probe for a SHARED, OVERSUBSCRIBED GPU (several processes with more user queues than the
// hardware has slots: the scheduler then time-slices queues by saving and restoring wave contexts mid-kernel).
// A workgroup of 256 threads repeats: every thread writes a value that encodes (iteration, thread, workgroup) into its LDS
// slot, __syncthreads(), a few threads (`readers`) re-read ALL slots `spin` times and count every word that is not the
// value its owner wrote for THIS iteration, __syncthreads().  The structure is the statistics phase of the fused GroupNorm
// kernel (csrc/norm_act.hip: 256 partials, barrier, two threads fold them serially, barrier) with integers instead of
// sums, so that a stale slot shows as a mismatch instead of a statistic that is off in the third digit.
// usage: lds_barrier_probe [seconds] [readers (1..256)] [spin]      prints mismatch counts and the first few records
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

struct Rec { int it, t, got, want, block, reader; };

__global__ __launch_bounds__(256) void probe(unsigned long long* bad, Rec* recs, int iters, int readers, int spin, int salt) {
  __shared__ int sh[256];
  const int tid = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    const int base = (it + salt) * 1000003 + (int)blockIdx.x * 257;
    sh[tid] = base + tid * 7;
    __syncthreads();
    if (tid < readers) {
      for (int rep = 0; rep < spin; ++rep)
        for (int t = tid; t < 256; t += readers) {
          const int v = ((volatile int*)sh)[t];
          if (v != base + t * 7) {
            const unsigned long long k = atomicAdd(bad, 1ULL);
            if (k < 64) recs[k] = {it, t, v, base + t * 7, (int)blockIdx.x, tid};
          }
        }
    }
    __syncthreads();
  }
}

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 30.0;
  const int readers = argc > 2 ? atoi(argv[2]) : 2;
  const int spin = argc > 3 ? atoi(argv[3]) : 8;
  unsigned long long* bad;
  Rec* recs;
  hipMalloc(&bad, sizeof(*bad));
  hipMalloc(&recs, 64 * sizeof(Rec));
  hipMemset(bad, 0, sizeof(*bad));
  const auto t0 = std::chrono::steady_clock::now();
  long launches = 0;
  int salt = 0;
  for (;;) {
    for (int i = 0; i < 20; ++i, ++launches) hipLaunchKernelGGL(probe, dim3(2048), dim3(256), 0, 0, bad, recs, 64, readers, spin, salt += 64);
    hipDeviceSynchronize();
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > secs) break;
  }
  unsigned long long h = 0;
  Rec hr[64];
  hipMemcpy(&h, bad, sizeof(h), hipMemcpyDeviceToHost);
  hipMemcpy(hr, recs, sizeof(hr), hipMemcpyDeviceToHost);
  printf("lds_barrier_probe: readers %d spin %d: %ld launches x 2048 workgroups x 64 iterations in %.0f s, %llu stale words\n",
         readers, spin, launches, secs, h);
  for (unsigned long long i = 0; i < h && i < 8; ++i)
    printf("  it %d slot %d: got %d want %d (delta %d) block %d reader %d\n", hr[i].it, hr[i].t, hr[i].got, hr[i].want,
           hr[i].got - hr[i].want, hr[i].block, hr[i].reader);
  return h ? 1 : 0;
}

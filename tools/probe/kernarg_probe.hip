// Late kernel-argument read probe for a SHARED, OVERSUBSCRIBED GPU.
// hipcc loads kernel arguments with scalar loads from the kernarg segment WHEN THEY ARE FIRST NEEDED, which can be many
// microseconds after the kernel has started (round 3's fused GroupNorm loaded its gamma / beta pointers after its serial
// statistics phase).  This probe checks that such a late load still returns THIS launch's arguments: every launch gets a
// unique `magic` (process tag in the high bits, launch counter below); the kernel reads offset 0 of its kernarg segment with
// an explicit s_load_dword at once (`early`), spins, reads it again (`late`), and workgroup 0 stores both; the host compares
// them with the magic it passed.  A wrong value with another process's tag = a stale scalar-cache line / another process's
// kernarg memory at the same virtual address; the own tag with another counter = kernarg memory recycled early.
// usage: kernarg_probe [seconds] [process tag 1..255] [spin microseconds]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Rec { unsigned early, late; };

__global__ __launch_bounds__(64) void probe(unsigned magic, int idx, Rec* out, unsigned long long* sink, long long spin_ticks) {
  auto ka = __builtin_amdgcn_kernarg_segment_ptr();
  unsigned early, late;
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(early) : "s"(ka) : "memory");
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin_ticks) {}
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(late) : "s"(ka) : "memory");
  if (threadIdx.x == 0) {
    if (blockIdx.x == 0) out[idx] = {early, late};
    if (early != magic || late != magic) atomicAdd(sink, 1ULL);   // every workgroup checks against the compiler's own load
  }
}

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 30.0;
  const unsigned tag = argc > 2 ? (unsigned)atoi(argv[2]) : 1u;
  const double spin_us = argc > 3 ? atof(argv[3]) : 20.0;
  const int B = 200;
  Rec* out;
  unsigned long long* sink;
  (void)hipMalloc(&out, B * sizeof(Rec));
  (void)hipMalloc(&sink, sizeof(*sink));
  (void)hipMemset(sink, 0, sizeof(*sink));
  const long long ticks = (long long)(spin_us * 100.0);   // wall_clock64: 100 MHz
  const auto t0 = std::chrono::steady_clock::now();
  unsigned counter = 0;
  long launches = 0, bad = 0;
  std::vector<Rec> h(B);
  std::vector<unsigned> want(B);
  for (;;) {
    (void)hipMemset(out, 0, B * sizeof(Rec));
    for (int i = 0; i < B; ++i, ++launches) {
      want[i] = (tag << 24) | (++counter & 0xffffffu);
      hipLaunchKernelGGL(probe, dim3(512), dim3(64), 0, 0, want[i], i, out, sink, ticks);
    }
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), out, B * sizeof(Rec), hipMemcpyDeviceToHost);
    for (int i = 0; i < B; ++i)
      if (h[i].early != want[i] || h[i].late != want[i]) {
        if (bad < 8)
          printf("  launch magic %08x: early %08x late %08x (late tag %u, counter delta %d)\n", want[i], h[i].early, h[i].late,
                 h[i].late >> 24, (int)(h[i].late & 0xffffff) - (int)(want[i] & 0xffffff));
        ++bad;
      }
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > secs) break;
  }
  unsigned long long s = 0;
  (void)hipMemcpy(&s, sink, sizeof(s), hipMemcpyDeviceToHost);
  printf("kernarg_probe tag %u spin %.0f us: %ld launches in %.0f s: %ld launches whose workgroup 0 read another argument, "
         "%llu workgroups in total\n", tag, spin_us, launches, secs, bad, s);
  return (bad || s) ? 1 : 0;
}

// HBM streaming probe (standalone; run on the GPU box): which access shape reaches the best read / read+write rate?
// build: hipcc --offload-arch=gfx950 -O3 -o bw_probe bw_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void rd1(const float* __restrict__ x, float* out, long n) {
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += x[i];
  if (s == 123.456f) out[0] = s;
}
__global__ __launch_bounds__(256) void rd4(const f4* __restrict__ x, float* out, long n4) {
  f4 s = {0, 0, 0, 0};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) s += x[i];
  if (s.x + s.y + s.z + s.w == 123.456f) out[0] = s.x;
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void rd4u(const f4* __restrict__ x, float* out, long n4) {
  f4 s = {0, 0, 0, 0};
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(x + i + u * stride) : x[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) s += v[u];
  }
  for (; i < n4; i += stride) s += x[i];
  if (s.x + s.y + s.z + s.w == 123.456f) out[0] = s.x;
}
__global__ __launch_bounds__(256) void cp1(const float* __restrict__ x, float* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = 2.f * x[i];
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void cp4u(const f4* __restrict__ x, f4* __restrict__ y, long n4) {
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(x + i + u * stride) : x[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT) __builtin_nontemporal_store(v[u] * 2.f, y + i + u * stride);
      else y[i + u * stride] = v[u] * 2.f;
    }
  }
  for (; i < n4; i += stride) y[i] = x[i] * 2.f;
}
// 3 reads 2 writes (mul2-like)
template <bool V4>
__global__ __launch_bounds__(256) void m2(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                          float* __restrict__ o1, float* __restrict__ o2, long n) {
  if (V4) {
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
      const f4 x = ((const f4*)a)[i], y = ((const f4*)b)[i], z = ((const f4*)c)[i];
      ((f4*)o1)[i] = x * y;
      ((f4*)o2)[i] = x * z;
    }
  } else {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
      const float x = a[i];
      o1[i] = x * b[i];
      o2[i] = x * c[i];
    }
  }
}

template <typename F>
double timeit(F f, int iters = 20) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) f();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / iters * 1e-3;
}

int main() {
  const long n = 131072L * 512;  // 268 MB
  float *a, *b, *c, *o1, *o2, *out;
  hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&c, n * 4); hipMalloc(&o1, n * 4); hipMalloc(&o2, n * 4); hipMalloc(&out, 16);
  hipMemset(a, 0, n * 4); hipMemset(b, 0, n * 4); hipMemset(c, 0, n * 4);
  const double GB = n * 4 / 1e12;
  for (int grid : {1024, 2048, 4096, 8192, 65536}) {
    printf("grid %d\n", grid);
    double t;
    t = timeit([&] { hipLaunchKernelGGL(rd1, dim3(grid), dim3(256), 0, 0, a, out, n); }); printf("  read  scalar        %7.1f us %5.2f TB/s\n", t * 1e6, GB / t);
    t = timeit([&] { hipLaunchKernelGGL(rd4, dim3(grid), dim3(256), 0, 0, (const f4*)a, out, n / 4); }); printf("  read  f4            %7.1f us %5.2f TB/s\n", t * 1e6, GB / t);
    t = timeit([&] { hipLaunchKernelGGL((rd4u<4, false>), dim3(grid), dim3(256), 0, 0, (const f4*)a, out, n / 4); }); printf("  read  f4 x4         %7.1f us %5.2f TB/s\n", t * 1e6, GB / t);
    t = timeit([&] { hipLaunchKernelGGL((rd4u<4, true>), dim3(grid), dim3(256), 0, 0, (const f4*)a, out, n / 4); }); printf("  read  f4 x4 nt      %7.1f us %5.2f TB/s\n", t * 1e6, GB / t);
    t = timeit([&] { hipLaunchKernelGGL((rd4u<8, false>), dim3(grid), dim3(256), 0, 0, (const f4*)a, out, n / 4); }); printf("  read  f4 x8         %7.1f us %5.2f TB/s\n", t * 1e6, GB / t);
    t = timeit([&] { hipLaunchKernelGGL(cp1, dim3(grid), dim3(256), 0, 0, a, o1, n); }); printf("  copy  scalar        %7.1f us %5.2f TB/s\n", t * 1e6, 2 * GB / t);
    t = timeit([&] { hipLaunchKernelGGL((cp4u<1, false>), dim3(grid), dim3(256), 0, 0, (const f4*)a, (f4*)o1, n / 4); }); printf("  copy  f4            %7.1f us %5.2f TB/s\n", t * 1e6, 2 * GB / t);
    t = timeit([&] { hipLaunchKernelGGL((cp4u<4, false>), dim3(grid), dim3(256), 0, 0, (const f4*)a, (f4*)o1, n / 4); }); printf("  copy  f4 x4         %7.1f us %5.2f TB/s\n", t * 1e6, 2 * GB / t);
    t = timeit([&] { hipLaunchKernelGGL((cp4u<4, true>), dim3(grid), dim3(256), 0, 0, (const f4*)a, (f4*)o1, n / 4); }); printf("  copy  f4 x4 nt      %7.1f us %5.2f TB/s\n", t * 1e6, 2 * GB / t);
    t = timeit([&] { hipLaunchKernelGGL((m2<false>), dim3(grid), dim3(256), 0, 0, a, b, c, o1, o2, n); }); printf("  3r2w  scalar        %7.1f us %5.2f TB/s\n", t * 1e6, 5 * GB / t);
    t = timeit([&] { hipLaunchKernelGGL((m2<true>), dim3(grid), dim3(256), 0, 0, a, b, c, o1, o2, n); }); printf("  3r2w  f4            %7.1f us %5.2f TB/s\n", t * 1e6, 5 * GB / t);
  }
  return 0;
}

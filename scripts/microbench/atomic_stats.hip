// Microbenchmark: what do per-workgroup int64 atomic adds into R accumulator rows cost a producer kernel, and what does
// a consumer pay for reading R rows per workgroup in its prologue?   hipcc --offload-arch=gfx950 -O3 atomic_stats.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// producer stand-in: streams `bytes_per_wg` of input (so the launch lasts as long as a small convolution), then thread t
// adds one value to acc[row][t] for t < nvals (row = blockIdx % R)
__global__ __launch_bounds__(256) void producer(const float4* __restrict__ x, int n4_per_wg, long long* __restrict__ acc, int nvals,
                                                int R, int mode, float* __restrict__ rows_out) {
  float4 s = {0.f, 0.f, 0.f, 0.f};
  const float4* p = x + (size_t)blockIdx.x * n4_per_wg;
  for (int i = threadIdx.x; i < n4_per_wg; i += 256) { float4 v = p[i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
  const float val = s.x + s.y + s.z + s.w;
  if (mode == 0) {                                   // partial row per workgroup (what the library does today)
    for (int t = threadIdx.x; t < nvals; t += 256) rows_out[(size_t)blockIdx.x * nvals + t] = val;
  } else {                                           // int64 atomics into R rows
    const int row = blockIdx.x % R;
    for (int t = threadIdx.x; t < nvals; t += 256) {
      const long long q = (long long)(val * 1048576.0f) + t;
      __hip_atomic_fetch_add(acc + (size_t)row * nvals + t, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// consumer stand-in: every workgroup first sums R rows of nvals int64 (prologue), then streams its share of a tensor
__global__ __launch_bounds__(256) void consumer(const float4* __restrict__ x, float4* __restrict__ y, int n4_per_wg,
                                                const long long* __restrict__ acc, int nvals, int R, const float* __restrict__ par) {
  __shared__ float prm[2048];
  if (R > 0) {
    for (int t = threadIdx.x; t < nvals; t += 256) {
      long long s = 0;
      for (int r = 0; r < R; ++r) s += acc[(size_t)r * nvals + t];
      prm[t] = (float)s * 1e-9f;
    }
  } else {
    for (int t = threadIdx.x; t < nvals; t += 256) prm[t] = par[t];
  }
  __syncthreads();
  const float a = prm[threadIdx.x % nvals];
  const float4* p = x + (size_t)blockIdx.x * n4_per_wg;
  float4* q = y + (size_t)blockIdx.x * n4_per_wg;
  for (int i = threadIdx.x; i < n4_per_wg; i += 256) { float4 v = p[i]; v.x += a; v.y += a; v.z += a; v.w += a; q[i] = v; }
}

int main() {
  const size_t bytes = 33554432;                     // one stage-2 tensor (16 x 64 x 64 x 128 floats)
  float4 *x, *y; long long* acc; float *rows, *par;
  CK(hipMalloc(&x, bytes)); CK(hipMalloc(&y, bytes)); CK(hipMalloc(&acc, 64 * 2048 * 8)); CK(hipMalloc(&rows, 4096 * 2048 * 4));
  CK(hipMalloc(&par, 2048 * 4));
  CK(hipMemset(x, 0, bytes)); CK(hipMemset(acc, 0, 64 * 2048 * 8)); CK(hipMemset(par, 0, 2048 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipStream_t st; CK(hipStreamCreate(&st));
  auto time_it = [&](auto launch) {
    for (int i = 0; i < 5; ++i) launch();
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 50; ++i) launch();
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f / 50.f;
  };
  for (int wgs : {512, 2048}) {
    const int n4 = (int)(bytes / 16 / wgs);
    for (int nvals : {128, 512, 768}) {               // 2 sums x 32 ch x hi/lo = 128 ... 3 sums x 128 ch x hi/lo = 768
      const float base = time_it([&] { hipLaunchKernelGGL(producer, dim3(wgs), dim3(256), 0, st, x, n4, acc, nvals, 1, 0, rows); });
      printf("producer wgs %4d nvals %3d: partial rows %.2f us |", wgs, nvals, base);
      for (int R : {1, 2, 4, 8, 16}) {
        const float t = time_it([&] { hipLaunchKernelGGL(producer, dim3(wgs), dim3(256), 0, st, x, n4, acc, nvals, R, 1, rows); });
        printf(" R=%d %.2f", R, t);
      }
      printf("\n");
    }
  }
  for (int wgs : {1024, 2048, 4096}) {
    const int n4 = (int)(bytes / 16 / wgs);
    for (int nvals : {128, 512, 768}) {
      const float base = time_it([&] { hipLaunchKernelGGL(consumer, dim3(wgs), dim3(256), 0, st, x, y, n4, acc, nvals, 0, par); });
      printf("consumer wgs %4d nvals %3d: params only %.2f us |", wgs, nvals, base);
      for (int R : {1, 2, 4, 8, 16}) {
        const float t = time_it([&] { hipLaunchKernelGGL(consumer, dim3(wgs), dim3(256), 0, st, x, y, n4, acc, nvals, R, par); });
        printf(" R=%d %.2f", R, t);
      }
      printf("\n");
    }
  }
  return 0;
}

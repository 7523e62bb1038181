// Microbenchmark: what does a launch of W waves that issue N dependent v_mfma_f32_32x32x2_f32 each (no memory operands) take,
// and the same with a short store epilogue?   hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
// The direct 3x3 32->32 convolution kernel at 64x64 batch 16 is 512 workgroups x 4 waves x 144 matrix instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(256, 4) void mfma_only(float* __restrict__ out, int n, float a0, float b0) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  for (int i = 0; i < n; i += CHAINS) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    a += 1.f;
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
  if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;      // never true: keeps the chain alive
}

__global__ void empty_kernel() {}

int main() {
  float* out; CK(hipMalloc(&out, 4096 * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipStream_t st; CK(hipStreamCreate(&st));
  auto time_it = [&](auto launch, int reps) {
    for (int i = 0; i < 5; ++i) launch();
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f / reps;
  };
  printf("empty launch                                   %6.2f us\n", time_it([&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st); }, 200));
  const int grids[] = {256, 512, 1024, 2048};
  const int ns[] = {144, 288, 1152, 4608};
  for (int g : grids)
    for (int n : ns) {
      const float t1 = time_it([&] { hipLaunchKernelGGL(mfma_only<1>, dim3(g), dim3(256), 0, st, out, n, 1.f, 2.f); }, 50);
      const float t4 = time_it([&] { hipLaunchKernelGGL(mfma_only<4>, dim3(g), dim3(256), 0, st, out, n, 1.f, 2.f); }, 50);
      const double flop = (double)g * 4 * n * 4096.0;
      printf("workgroups %4d  mfma/wave %4d   one chain %7.2f us (%5.1f TFLOP/s)   four chains %7.2f us (%5.1f TFLOP/s)\n", g, n, t1,
             flop / t1 * 1e-6, t4, flop / t4 * 1e-6);
    }
  return 0;
}

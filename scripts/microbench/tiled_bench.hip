// Times the LDS-tiled bfloat16 3x3 kernel alone on the wide layers of SegNet / U-Net at batch 4, 512x512:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off scripts/microbench/tiled_bench.hip -o scripts/microbench/tiled_bench
// Prints, per layer and epilogue (0 plain, 1 statistics, 2 data-gradient form with addend-free BatchNormalization sums),
// the average duration over 20 launches, the achieved TFLOP/s and the algorithmic GB/s.
#include "../../semantic-segmentation_amd/csrc/conv_tiled.hip"
#include <stdarg.h>
#include <vector>
#include <cmath>
#include <cstring>
namespace ssmi {
void set_last_kernel(const char*, ...) {}
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
  struct L { int cin, cout, hw; } layers[] = {{64, 64, 512}, {192, 64, 512}, {64, 192, 512}, {128, 128, 256}, {384, 128, 256}, {256, 256, 128},
                                             {768, 256, 128}, {512, 512, 64}, {1024, 512, 64}, {1024, 1024, 32}};
  const int n = 4;
  size_t max_act = (size_t)n * 512 * 512 * 192 * 2;
  void *x, *y, *bx, *w; float *part, *par;
  CK(hipMalloc(&x, max_act)); CK(hipMalloc(&y, max_act)); CK(hipMalloc(&bx, max_act));
  CK(hipMalloc(&w, (size_t)9 * 1024 * 1024 * 2)); CK(hipMalloc(&part, (size_t)8192 * 3 * 1024 * 4)); CK(hipMalloc(&par, 8 * 1024 * 4));
  // small finite values: 0x3c00 = bf16 0.0078125-ish patterns; fill with a repeating pattern of bf16(0.5)/bf16(-0.25)
  std::vector<unsigned short> h(max_act / 2);
  const bool random_data = argc > 1 && argv[1][0] == 'r';      // "r": normally distributed values (what a network holds) instead of a two-valued pattern
  unsigned long long sd = 88172645463325252ull;
  auto rnd = [&]() { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; return (double)(sd >> 11) / 9007199254740992.0; };
  for (size_t i = 0; i < h.size(); ++i) {
    if (!random_data) { h[i] = (i * 2654435761u >> 28) & 1 ? 0x3f00 : 0xbe80; continue; }
    const float v = (float)(std::sqrt(-2.0 * std::log(rnd() + 1e-300)) * std::cos(6.283185307179586 * rnd())) * 0.5f;
    unsigned u; memcpy(&u, &v, 4);
    h[i] = (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
  }
  CK(hipMemcpy(x, h.data(), max_act, hipMemcpyHostToDevice)); CK(hipMemcpy(bx, h.data(), max_act, hipMemcpyHostToDevice));
  CK(hipMemcpy(w, h.data(), (size_t)9 * 1024 * 1024 * 2, hipMemcpyHostToDevice));
  std::vector<float> hp(8 * 1024, 0.5f);
  CK(hipMemcpy(par, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (auto& l : layers) {
    for (int epi = 0; epi < 3; ++epi) {
      ssmi::BnBwdArgs bn{};
      float* p = nullptr;
      if (epi == 1) p = part;
      if (epi == 2) { bn.x = bx; bn.ld = l.cout; bn.mean = par; bn.invstd = par + 1024; bn.gamma = par + 2048; bn.beta = par + 3072;
                      bn.slope = 0.1f; bn.act = SSMI_ACT_LEAKY; bn.hw = (unsigned)(l.hw * l.hw); bn.part = part; }
      int rows = 0;
      const int reps = 20;
      for (int i = 0; i < 3; ++i)
        if (ssmi::launch_conv_tiled_bf16(n, l.hw, l.hw, l.cin, l.cout, l.cin, l.cout, x, w, par, y, p, &rows, bn, st) != SSMI_OK) return 2;
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < reps; ++i) ssmi::launch_conv_tiled_bf16(n, l.hw, l.hw, l.cin, l.cout, l.cin, l.cout, x, w, par, y, p, &rows, bn, st);
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / reps, M = (double)n * l.hw * l.hw;
      const double fl = 2.0 * M * 9 * l.cin * l.cout, by = M * (l.cin + l.cout + (epi == 2 ? l.cout : 0)) * 2.0 + 18.0 * l.cin * l.cout;
      printf("%4d -> %4d @%3d  epi %d  %8.1f us  %7.1f TFLOP/s  %7.1f GB/s\n", l.cin, l.cout, l.hw, epi, us, fl / us * 1e-6, by / us * 1e-3);
    }
  }
  return 0;
}

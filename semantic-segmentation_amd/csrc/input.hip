// Device pre-pass of the input contract (src/generators.py:711-770 normalisation, :1787-1812 mask index encoding,
// :1814-1825 per-pixel weights): one streaming kernel from the uint8 crops the CPU workers produce to the three
// tensors the training step consumes.  Pure HBM streaming: 4 B read, 20 B written per pixel.
#include "common.h"

namespace ssmi {

// One thread per pixel.  The arithmetic order of the normalisation is the reference's (float32 /255, -0.5, *2,
// -mean, /std; contraction is off for the whole library), so X is bit-identical to the numpy result.
__global__ __launch_bounds__(256) void preprocess_kernel(const uint8_t* __restrict__ images, const uint8_t* __restrict__ masks,
                                                         const int32_t* __restrict__ r_to_class,
                                                         const float* __restrict__ class_weights, int num_classes,
                                                         int num_labeled, const float* __restrict__ mean3,
                                                         const float* __restrict__ std3, unsigned hw, unsigned total,
                                                         float* __restrict__ X, int32_t* __restrict__ Y,
                                                         float* __restrict__ W, int32_t* __restrict__ bad) {
  __shared__ int32_t lut[256];
  lut[threadIdx.x] = r_to_class[threadIdx.x];
  __syncthreads();
  float m[3] = {0.f, 0.f, 0.f}, sd[3] = {1.f, 1.f, 1.f};
  if (mean3) { m[0] = mean3[0]; m[1] = mean3[1]; m[2] = mean3[2]; }
  if (std3) { sd[0] = std3[0]; sd[1] = std3[1]; sd[2] = std3[2]; }
  int nbad = 0;
  for (unsigned p = blockIdx.x * 256u + threadIdx.x; p < total; p += gridDim.x * 256u) {
    const uint8_t* px = images + (size_t)p * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float v = (float)px[c];
      v = v / 255.0f;
      v = v - 0.5f;
      v = v * 2.0f;
      if (mean3) v = v - m[c];
      if (std3) v = v / sd[c];
      X[(size_t)p * 3 + c] = v;
    }
    if (masks) {
      const int r = masks[p];
      const bool labeled = (int)(p / hw) < num_labeled;
      int cls = labeled ? lut[r] : r;
      float wv = 1.f;
      if (labeled) {
        if (cls < 0 || cls >= num_classes) { ++nbad; cls = 0; wv = 0.f; }
        else wv = class_weights[cls];
      }
      Y[p] = cls;
      W[p] = wv;
    }
  }
  if (nbad && bad) atomicAdd(bad, nbad);
}

}  // namespace ssmi

using namespace ssmi;

extern "C" {

ssmi_status ssmi_preprocess_batch(const uint8_t* images, const uint8_t* masks, const int32_t* r_to_class,
                                  const float* class_weights, int32_t num_classes, int32_t num_labeled,
                                  const float* mean3, const float* std3, int32_t n, int32_t h, int32_t w, float* X,
                                  int32_t* Y, float* W, int32_t* bad, void* stream) {
  SSMI_REQUIRE(images && X && n > 0 && h > 0 && w > 0, SSMI_E_INVALID_ARG, "preprocess_batch: bad argument");
  SSMI_REQUIRE(!masks || (r_to_class && class_weights && Y && W && num_classes > 0 && num_labeled >= 0 && num_labeled <= n),
               SSMI_E_INVALID_ARG, "preprocess_batch: masks need the colour table, class weights, Y and W");
  const int64_t total = (int64_t)n * h * w;
  SSMI_REQUIRE(total < 0x7FFFFFFFLL, SSMI_E_UNSUPPORTED, "preprocess_batch: more than 2^31 pixels");
  int64_t nb = ceil_div64(total, 256 * 4);
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(preprocess_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), images, masks,
                     masks ? r_to_class : reinterpret_cast<const int32_t*>(images), class_weights, num_classes, num_labeled,
                     mean3, std3, (unsigned)(h * w), (unsigned)total, X, Y, W, bad);
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

}  // extern "C"

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


// Full-image inference input (src/predict.py:97-125): normalise an uint8 image and centre it in a zero-copy padded
// frame whose border pixels take `cval3` (the reference pads AFTER normalising, with the data set's per-channel mean as
// the fill colour: src/predict.py:124, src/utils/image_utils.py:1400-1434).  One thread per output pixel.
__global__ __launch_bounds__(256) void predict_prepare_kernel(const uint8_t* __restrict__ image, const float* __restrict__ mean3,
                                                              const float* __restrict__ std3, int h, int w, int v0, int h0,
                                                              int hp, int wp, const float* __restrict__ cval3,
                                                              float* __restrict__ X) {
  const unsigned total = (unsigned)hp * (unsigned)wp;
  float m[3] = {0.f, 0.f, 0.f}, sd[3] = {1.f, 1.f, 1.f}, cv[3] = {0.f, 0.f, 0.f};
  if (mean3) { m[0] = mean3[0]; m[1] = mean3[1]; m[2] = mean3[2]; }
  if (std3) { sd[0] = std3[0]; sd[1] = std3[1]; sd[2] = std3[2]; }
  if (cval3) { cv[0] = cval3[0]; cv[1] = cval3[1]; cv[2] = cval3[2]; }
  for (unsigned p = blockIdx.x * 256u + threadIdx.x; p < total; p += gridDim.x * 256u) {
    const int y = (int)(p / (unsigned)wp) - v0, x = (int)(p % (unsigned)wp) - h0;
    float* o = X + (size_t)p * 3;
    if ((unsigned)y < (unsigned)h && (unsigned)x < (unsigned)w) {
      const uint8_t* px = image + ((size_t)y * w + x) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float v = (float)px[c];
        v = v / 255.0f;
        v = v - 0.5f;
        v = v * 2.0f;
        if (mean3) v = v - m[c];
        if (std3) v = v / sd[c];
        o[c] = v;
      }
    } else {
      o[0] = cv[0]; o[1] = cv[1]; o[2] = cv[2];
    }
  }
}

// Crop the padding off a prediction, weight the background class and add it to the running sum of the ensemble
// (src/predict.py:482-527).  `accumulate` = 0 starts the sum (acc = p), 1 continues it (acc += p), which is the order
// numpy's mean over the stacked predictions adds them in.
__global__ __launch_bounds__(256) void predict_accumulate_kernel(const float* __restrict__ pred, int wp, int C, int v0, int h0,
                                                                 int h, int w, float bg_weight, int accumulate,
                                                                 float* __restrict__ acc) {
  const unsigned total = (unsigned)h * (unsigned)w * (unsigned)C;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const unsigned c = i % (unsigned)C, p = i / (unsigned)C;
    const unsigned y = p / (unsigned)w, x = p % (unsigned)w;
    float v = pred[((size_t)(y + v0) * wp + (x + h0)) * C + c];
    if (c == 0) v = v * bg_weight;
    acc[i] = accumulate ? acc[i] + v : v;
  }
}

// Mean of the ensemble (sum / count, one float32 division like numpy's mean) and the k best classes per pixel in
// descending order of activation (src/utils/prediction_utils.py:95-99: argsort(-x)[..., :k]); equal activations rank
// the lower class index first.  One thread per pixel; selection by k passes over the C values held in cache.
__global__ __launch_bounds__(256) void predict_topk_kernel(const float* __restrict__ acc, unsigned npix, int C, float count,
                                                           int k, float* __restrict__ mean_out, int32_t* __restrict__ topk) {
  for (unsigned p = blockIdx.x * 256u + threadIdx.x; p < npix; p += gridDim.x * 256u) {
    const float* a = acc + (size_t)p * C;
    if (mean_out) {
      float* mo = mean_out + (size_t)p * C;
      for (int c = 0; c < C; ++c) mo[c] = a[c] / count;
    }
    if (topk) {
      float prev_v = 0.f;
      int prev_c = -1;
      for (int r = 0; r < k; ++r) {
        float best = 0.f;
        int bc = -1;
        for (int c = 0; c < C; ++c) {
          const float v = a[c] / count;
          // candidates are the values ranked strictly after (prev_v, prev_c) in (descending value, ascending class) order
          const bool after = prev_c < 0 || v < prev_v || (v == prev_v && c > prev_c);
          if (after && (bc < 0 || v > best)) { best = v; bc = c; }
        }
        topk[(size_t)r * npix + p] = bc;
        prev_v = best;
        prev_c = bc;
      }
    }
  }
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

ssmi_status ssmi_predict_prepare(const uint8_t* image, const float* mean3, const float* std3, int32_t h, int32_t w,
                                 int32_t v_pad_before, int32_t h_pad_before, int32_t hp, int32_t wp, const float* cval3,
                                 float* X, void* stream) {
  SSMI_REQUIRE(image && X && h > 0 && w > 0, SSMI_E_INVALID_ARG, "predict_prepare: bad argument");
  SSMI_REQUIRE(v_pad_before >= 0 && h_pad_before >= 0 && hp >= h + v_pad_before && wp >= w + h_pad_before, SSMI_E_INVALID_ARG,
               "predict_prepare: the %dx%d image at (%d,%d) does not fit the %dx%d frame", h, w, v_pad_before, h_pad_before, hp, wp);
  const int64_t total = (int64_t)hp * wp;
  SSMI_REQUIRE(total < 0x7FFFFFFFLL / 3, SSMI_E_UNSUPPORTED, "predict_prepare: frame too large");
  int64_t nb = ceil_div64(total, 256);
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(predict_prepare_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), image, mean3, std3, h, w,
                     v_pad_before, h_pad_before, hp, wp, cval3, X);
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

ssmi_status ssmi_predict_accumulate(const float* pred, int32_t hp, int32_t wp, int32_t num_classes, int32_t v_pad_before,
                                    int32_t h_pad_before, int32_t h, int32_t w, float bg_weight, int32_t accumulate, float* acc,
                                    void* stream) {
  SSMI_REQUIRE(pred && acc && h > 0 && w > 0 && num_classes > 0, SSMI_E_INVALID_ARG, "predict_accumulate: bad argument");
  SSMI_REQUIRE(v_pad_before >= 0 && h_pad_before >= 0 && hp >= h + v_pad_before && wp >= w + h_pad_before, SSMI_E_INVALID_ARG,
               "predict_accumulate: the %dx%d crop at (%d,%d) leaves the %dx%d prediction", h, w, v_pad_before, h_pad_before, hp, wp);
  const int64_t total = (int64_t)h * w * num_classes;
  SSMI_REQUIRE((int64_t)hp * wp * num_classes < 0x7FFFFFFFLL, SSMI_E_UNSUPPORTED, "predict_accumulate: prediction too large");
  int64_t nb = ceil_div64(total, 256 * 4);
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(predict_accumulate_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), pred, wp, num_classes,
                     v_pad_before, h_pad_before, h, w, bg_weight, accumulate, acc);
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

ssmi_status ssmi_predict_topk(const float* acc, int64_t npix, int32_t num_classes, int32_t count, int32_t k, float* mean_out,
                              int32_t* topk, void* stream) {
  SSMI_REQUIRE(acc && npix > 0 && num_classes > 0 && count > 0, SSMI_E_INVALID_ARG, "predict_topk: bad argument");
  SSMI_REQUIRE(!topk || (k >= 1 && k <= num_classes), SSMI_E_INVALID_ARG, "predict_topk: k %d outside [1, %d]", k, num_classes);
  SSMI_REQUIRE(npix * num_classes < 0x7FFFFFFFLL, SSMI_E_UNSUPPORTED, "predict_topk: prediction too large");
  int64_t nb = ceil_div64(npix, 256);
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(predict_topk_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), acc, (unsigned)npix, num_classes,
                     (float)count, k, mean_out, topk);
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

}  // extern "C"

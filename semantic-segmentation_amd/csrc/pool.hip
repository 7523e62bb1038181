// MaxPoolingWithArgmax2D / MaxUnpooling2D / nearest upsampling.  Pure HBM streaming kernels: one thread per
// channel (or group of 4 channels when c % 4 == 0) of an output pixel, consecutive lanes on consecutive channels.
#include "common.h"

namespace ssmi {

__device__ __forceinline__ f32x4 ld4p(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// generic scalar version (any c): one thread per output element
template <typename T>
__global__ __launch_bounds__(256) void maxpool_argmax_fwd_scalar(const T* __restrict__ x, int n, int h, int w, int c,
                                                                 int ho, int wo, T* __restrict__ y, int y_ld,
                                                                 int32_t* __restrict__ idx_i, float* __restrict__ idx_f,
                                                                 int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % c);
    int64_t q = i / c;
    const int ox = (int)(q % wo);
    q /= wo;
    const int oy = (int)(q % ho);
    const int n_ = (int)(q / ho);
    const T* base = x + (int64_t)n_ * h * w * c;
    float best = 0.f;
    int bi = 0;
    bool first = true;
#pragma unroll
    for (int dh = 0; dh < 2; ++dh) {
      const int iy = oy * 2 + dh;
      if (iy >= h) continue;
#pragma unroll
      for (int dw = 0; dw < 2; ++dw) {
        const int ix = ox * 2 + dw;
        if (ix >= w) continue;
        const int flat = (iy * w + ix) * c + ch;
        const float v = Act<T>::gld(base + flat);
        if (first || v > best) { best = v; bi = flat; first = false; }
      }
    }
    Act<T>::gst(y + ((int64_t)(n_ * ho + oy) * wo + ox) * y_ld + ch, best);
    if (idx_i) idx_i[i] = bi;
    if (idx_f) idx_f[i] = (float)bi;
  }
}

// vector version: c % 4 == 0, y_ld % 4 == 0, even h and w
template <typename T>
__global__ __launch_bounds__(256) void maxpool_argmax_fwd_vec(const T* __restrict__ x, int n, int h, int w, int c4n,
                                                              T* __restrict__ y, int y_ld, int32_t* __restrict__ idx_i,
                                                              float* __restrict__ idx_f, int64_t total4) {
  const int c = c4n * 4;
  const int ho = h >> 1, wo = w >> 1;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % c4n) * 4;
    int64_t q = i / c4n;
    const int ox = (int)(q % wo);
    const int64_t q2 = q / wo;
    const int oy = (int)(q2 % ho);
    const int n_ = (int)(q2 / ho);
    const int f00 = ((2 * oy) * w + 2 * ox) * c + ch;
    const T* base = x + (int64_t)n_ * h * w * c;
    const f32x4 v00 = Act<T>::gld4(base + f00), v01 = Act<T>::gld4(base + f00 + c), v10 = Act<T>::gld4(base + f00 + w * c),
                v11 = Act<T>::gld4(base + f00 + w * c + c);
    f32x4 best = v00;
    int bi[4] = {f00, f00 + 1, f00 + 2, f00 + 3};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (v01[j] > best[j]) { best[j] = v01[j]; bi[j] = f00 + c + j; }
      if (v10[j] > best[j]) { best[j] = v10[j]; bi[j] = f00 + w * c + j; }
      if (v11[j] > best[j]) { best[j] = v11[j]; bi[j] = f00 + w * c + c + j; }
    }
    Act<T>::gst4(y + q * y_ld + ch, best);
    if (idx_i) *reinterpret_cast<int4*>(idx_i + i * 4) = make_int4(bi[0], bi[1], bi[2], bi[3]);
    if (idx_f) *reinterpret_cast<f32x4*>(idx_f + i * 4) = f32x4{(float)bi[0], (float)bi[1], (float)bi[2], (float)bi[3]};
  }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_argmax_bwd_kernel(const T* __restrict__ dy, int dy_ld,
                                                                 const int32_t* __restrict__ idx, int n, int h, int w, int c,
                                                                 int ho, int wo, T* __restrict__ dx, int accumulate,
                                                                 int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % c);
    int64_t q = i / c;
    const int ix = (int)(q % w);
    q /= w;
    const int iy = (int)(q % h);
    const int n_ = (int)(q / h);
    const int64_t o = ((int64_t)n_ * ho + (iy >> 1)) * wo + (ix >> 1);
    const int flat = (iy * w + ix) * c + ch;
    const float g = (idx[o * c + ch] == flat) ? Act<T>::gld(dy + o * dy_ld + ch) : 0.f;
    Act<T>::gst(dx + i, accumulate ? Act<T>::gld(dx + i) + g : g);
  }
}

// The same with four channels per thread (c % 4 == 0, dy_ld % 4 == 0, 16-byte aligned tensors): 16-byte accesses and 32-bit
// index arithmetic (every tensor is below 2^31 elements).
template <typename T>
__global__ __launch_bounds__(256) void maxpool_argmax_bwd_vec(const T* __restrict__ dy, int dy_ld, const int32_t* __restrict__ idx,
                                                              int h, int w, int c, int ho, int wo, T* __restrict__ dx,
                                                              int accumulate, unsigned total4) {
  const unsigned c4n = (unsigned)c >> 2;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total4; i += gridDim.x * 256u) {
    const unsigned q0 = i / c4n, ch = (i - q0 * c4n) * 4u;
    const unsigned q1 = q0 / (unsigned)w, ix = q0 - q1 * (unsigned)w;
    const unsigned n_ = q1 / (unsigned)h, iy = q1 - n_ * (unsigned)h;
    const unsigned o = (n_ * (unsigned)ho + (iy >> 1)) * (unsigned)wo + (ix >> 1);
    const int flat = (int)((iy * (unsigned)w + ix) * (unsigned)c + ch);
    const int4 id = *reinterpret_cast<const int4*>(idx + (size_t)o * c + ch);
    const f32x4 g = Act<T>::gld4(dy + (size_t)o * dy_ld + ch);
    f32x4 v;
    v[0] = id.x == flat ? g[0] : 0.f;
    v[1] = id.y == flat + 1 ? g[1] : 0.f;
    v[2] = id.z == flat + 2 ? g[2] : 0.f;
    v[3] = id.w == flat + 3 ? g[3] : 0.f;
    T* d = dx + (size_t)i * 4;
    if (accumulate) v = v + Act<T>::gld4(d);
    Act<T>::gst4(d, v);
  }
}

__global__ __launch_bounds__(256) void maxunpool_scatter_kernel(const float* __restrict__ x, const int32_t* __restrict__ idx,
                                                                int64_t per_image_in, int64_t per_image_out,
                                                                float* __restrict__ y, int64_t total, int* __restrict__ bad) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t n_ = i / per_image_in;
    const int64_t f = idx[i];
    if (f >= 0 && f < per_image_out) atomicAdd(y + n_ * per_image_out + f, x[i]);
  }
}

__global__ __launch_bounds__(256) void maxunpool_local_kernel(const float* __restrict__ x, const int32_t* __restrict__ idx,
                                                              int n, int h, int w, int c, float* __restrict__ y,
                                                              int64_t total) {
  // over the OUTPUT [n, 2h, 2w, c]
  const int H = 2 * h, W = 2 * w;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % c);
    int64_t q = i / c;
    const int ox = (int)(q % W);
    q /= W;
    const int oy = (int)(q % H);
    const int n_ = (int)(q / H);
    const int64_t s = (((int64_t)n_ * h + (oy >> 1)) * w + (ox >> 1)) * c + ch;
    const int flat = (oy * W + ox) * c + ch;
    y[i] = (idx[s] == flat) ? x[s] : 0.f;
  }
}

__global__ __launch_bounds__(256) void maxunpool_bwd_kernel(const float* __restrict__ dy, const int32_t* __restrict__ idx,
                                                            int64_t per_image_in, int64_t per_image_out,
                                                            float* __restrict__ dx, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t n_ = i / per_image_in;
    const int64_t f = idx[i];
    dx[i] = (f >= 0 && f < per_image_out) ? dy[n_ * per_image_out + f] : 0.f;
  }
}

__global__ __launch_bounds__(256) void idx_f32_to_i32_kernel(const float* __restrict__ s, int32_t* __restrict__ d, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) d[i] = (int32_t)s[i];
}

// nearest 2x upsampling; y (forward) / dy (backward) may be a channel slice of a wider tensor (pixel stride ld)
template <typename T>
__global__ __launch_bounds__(256) void upsample_fwd_kernel(const T* __restrict__ x, int n, int h, int w, int c,
                                                           T* __restrict__ y, int y_ld, int64_t total) {
  const int H = 2 * h, W = 2 * w;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % c);
    int64_t q = i / c;
    const int64_t pix = q;
    const int ox = (int)(q % W);
    q /= W;
    const int oy = (int)(q % H);
    const int n_ = (int)(q / H);
    y[pix * y_ld + ch] = x[(((int64_t)n_ * h + (oy >> 1)) * w + (ox >> 1)) * c + ch];      // a copy: no conversion
  }
}

template <typename T>
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const T* __restrict__ dy, int dy_ld, int n, int h, int w,
                                                           int c, T* __restrict__ dx, int64_t total) {
  const int W = 2 * w;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % c);
    int64_t q = i / c;
    const int x_ = (int)(q % w);
    q /= w;
    const int y_ = (int)(q % h);
    const int n_ = (int)(q / h);
    const T* b = dy + ((((int64_t)n_ * 2 * h + 2 * y_) * W) + 2 * x_) * dy_ld + ch;
    Act<T>::gst(dx + i, Act<T>::gld(b) + Act<T>::gld(b + dy_ld) + Act<T>::gld(b + (int64_t)W * dy_ld) +
                            Act<T>::gld(b + (int64_t)W * dy_ld + dy_ld));
  }
}

// Four channels per thread for the three kernels above / below (channel counts and pixel strides that are multiples of four,
// tensors aligned to four elements, fewer than 2^31 elements): 16-byte accesses, 32-bit index arithmetic.
template <typename T>
__global__ __launch_bounds__(256) void upsample_fwd_vec(const T* __restrict__ x, int h, int w, int c, T* __restrict__ y, int y_ld,
                                                        unsigned total4) {
  const unsigned c4n = (unsigned)c >> 2, W = 2u * (unsigned)w, H = 2u * (unsigned)h;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total4; i += gridDim.x * 256u) {
    const unsigned pix = i / c4n, ch = (i - pix * c4n) * 4u;
    const unsigned q1 = pix / W, ox = pix - q1 * W;
    const unsigned n_ = q1 / H, oy = q1 - n_ * H;
    const size_t src = ((size_t)(n_ * (unsigned)h + (oy >> 1)) * (unsigned)w + (ox >> 1)) * (unsigned)c + ch;
    Act<T>::gst4(y + (size_t)pix * y_ld + ch, Act<T>::gld4(x + src));      // a copy: widening and re-rounding is exact
  }
}

template <typename T>
__global__ __launch_bounds__(256) void upsample_bwd_vec(const T* __restrict__ dy, int dy_ld, int h, int w, int c, T* __restrict__ dx,
                                                        unsigned total4) {
  const unsigned c4n = (unsigned)c >> 2, W = 2u * (unsigned)w;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total4; i += gridDim.x * 256u) {
    const unsigned q0 = i / c4n, ch = (i - q0 * c4n) * 4u;
    const unsigned q1 = q0 / (unsigned)w, x_ = q0 - q1 * (unsigned)w;
    const unsigned n_ = q1 / (unsigned)h, y_ = q1 - n_ * (unsigned)h;
    const T* b = dy + ((size_t)(n_ * 2u * (unsigned)h + 2u * y_) * W + 2u * x_) * dy_ld + ch;
    const f32x4 v0 = Act<T>::gld4(b), v1 = Act<T>::gld4(b + dy_ld), v2 = Act<T>::gld4(b + (size_t)W * dy_ld),
                v3 = Act<T>::gld4(b + (size_t)W * dy_ld + dy_ld);
    Act<T>::gst4(dx + (size_t)i * 4, ((v0 + v1) + v2) + v3);      // the order of the scalar kernel's sum
  }
}

template <typename T>
__global__ __launch_bounds__(256) void copy_ld_vec(const T* __restrict__ src, int src_ld, T* __restrict__ dst, int dst_ld, int c,
                                                   int accumulate, unsigned total4) {
  const unsigned c4n = (unsigned)c >> 2;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total4; i += gridDim.x * 256u) {
    const unsigned p = i / c4n, ch = (i - p * c4n) * 4u;
    f32x4 v = Act<T>::gld4(src + (size_t)p * src_ld + ch);
    T* d = dst + (size_t)p * dst_ld + ch;
    if (accumulate) v = Act<T>::gld4(d) + v;
    Act<T>::gst4(d, v);
  }
}

template <typename T>
static bool vec4_ok(const T* a, const T* b, int c, int ld_a, int ld_b, int64_t total) {
  const size_t m = sizeof(T) * 4 - 1;
  return c % 4 == 0 && ld_a % 4 == 0 && ld_b % 4 == 0 && total < 0x7FFFFFFFLL && (reinterpret_cast<uintptr_t>(a) & m) == 0 &&
         (reinterpret_cast<uintptr_t>(b) & m) == 0;
}

// dst[p, 0:c] (=|+=) src[p, 0:c] with independent pixel strides: concat slices and their gradients
template <typename T>
__global__ __launch_bounds__(256) void copy_ld_kernel(const T* __restrict__ src, int src_ld, T* __restrict__ dst,
                                                      int dst_ld, int c, int accumulate, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % c);
    const int64_t p = i / c;
    const float v = Act<T>::gld(src + p * src_ld + ch);
    T* d = dst + p * dst_ld + ch;
    Act<T>::gst(d, accumulate ? Act<T>::gld(d) + v : v);
  }
}

// y[n, p, c] = x[n, p, c] * table[n, c]   (SpatialDropout2D keep/scale table; its own gradient too)
template <typename T>
__global__ __launch_bounds__(256) void scale_nc_kernel(const T* __restrict__ x, const float* __restrict__ table,
                                                       T* __restrict__ y, int64_t hw, int c, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % c);
    const int64_t n_ = i / ((int64_t)c * hw);
    Act<T>::gst(y + i, Act<T>::gld(x + i) * table[n_ * c + ch]);
  }
}

// float32 <-> bfloat16 (round to nearest even), 4 elements per thread where the count allows
__global__ __launch_bounds__(256) void cast_f32_to_bf16_kernel(const float* __restrict__ s, bf16* __restrict__ d, int64_t n) {
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
    Act<bf16>::gst4(d + i * 4, Act<float>::gld4(s + i * 4));
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) d[i] = (bf16)s[i];
}
__global__ __launch_bounds__(256) void cast_bf16_to_f32_kernel(const bf16* __restrict__ s, float* __restrict__ d, int64_t n) {
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
    Act<float>::gst4(d + i * 4, Act<bf16>::gld4(s + i * 4));
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) d[i] = (float)s[i];
}

__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ d, float v, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) d[i] = v;
}

static inline int grid_for(int64_t total) {
  int64_t b = ceil_div64(total, 256 * 4);
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}


// AveragePooling2D over the whole map / GlobalAveragePooling2D (src/models.py:1044-1046, 1667): y[n, c] = mean_p x[n, p, c].
// One workgroup per (image, 64-channel group): 4 pixel slices x 64 channels, each slice adds its pixels in order, the four
// slices are combined through LDS in a fixed order (deterministic), then one multiplication by 1 / pixels.
__global__ __launch_bounds__(256) void global_avgpool_fwd_kernel(const float* __restrict__ x, int64_t hw, int c,
                                                                 float* __restrict__ y) {
  __shared__ float sm[256];
  const int ch = blockIdx.y * 64 + (threadIdx.x & 63), slice = threadIdx.x >> 6;
  const float* xb = x + (int64_t)blockIdx.x * hw * c;
  float s0 = 0.f, s1 = 0.f;
  if (ch < c) {
    int64_t p = slice;
    for (; p + 4 < hw; p += 8) {
      s0 += xb[p * c + ch];
      s1 += xb[(p + 4) * c + ch];
    }
    if (p < hw) s0 += xb[p * c + ch];
  }
  sm[threadIdx.x] = s0 + s1;
  __syncthreads();
  if (slice == 0 && ch < c) {
    const float t = (sm[threadIdx.x] + sm[threadIdx.x + 64]) + (sm[threadIdx.x + 128] + sm[threadIdx.x + 192]);
    y[(int64_t)blockIdx.x * c + ch] = t / (float)hw;
  }
}

__global__ __launch_bounds__(256) void global_avgpool_bwd_kernel(const float* __restrict__ dy, int64_t hw, int c,
                                                                 float* __restrict__ dx, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % c);
    const int64_t n = i / ((int64_t)hw * c);
    dx[i] = dy[n * c + ch] / (float)hw;
  }
}

}  // namespace ssmi

using namespace ssmi;

extern "C" {

}  // extern "C"

template <typename T>
static ssmi_status maxpool_argmax_fwd_impl(const T* x, int32_t n, int32_t h, int32_t w, int32_t c, T* y, int32_t y_ld,
                                           int32_t* idx_i32, float* idx_f32, void* stream) {
  SSMI_REQUIRE(x && y && n > 0 && h > 0 && w > 0 && c > 0 && y_ld >= c, SSMI_E_INVALID_ARG, "maxpool_argmax_fwd: bad argument");
  SSMI_REQUIRE((int64_t)h * w * c < (int64_t)1 << 31, SSMI_E_UNSUPPORTED, "maxpool_argmax_fwd: per-image index exceeds int32");
  const int ho = (h + 1) / 2, wo = (w + 1) / 2;
  const bool vec = (c % 4 == 0) && (y_ld % 4 == 0) && (h % 2 == 0) && (w % 2 == 0) &&
                   ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
  if (vec) {
    const int64_t total4 = (int64_t)n * ho * wo * (c / 4);
    hipLaunchKernelGGL(maxpool_argmax_fwd_vec<T>, dim3(grid_for(total4)), dim3(256), 0, as_stream(stream), x, n, h, w, c / 4, y,
                       y_ld, idx_i32, idx_f32, total4);
  } else {
    const int64_t total = (int64_t)n * ho * wo * c;
    hipLaunchKernelGGL(maxpool_argmax_fwd_scalar<T>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), x, n, h, w, c, ho,
                       wo, y, y_ld, idx_i32, idx_f32, total);
  }
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

template <typename T>
static ssmi_status maxpool_argmax_bwd_impl(const T* dy, int32_t dy_ld, const int32_t* idx, int32_t n, int32_t h, int32_t w,
                                           int32_t c, T* dx, int32_t accumulate, void* stream) {
  SSMI_REQUIRE(dy && idx && dx && n > 0 && h > 0 && w > 0 && c > 0 && dy_ld >= c, SSMI_E_INVALID_ARG, "maxpool_argmax_bwd: bad argument");
  const int ho = (h + 1) / 2, wo = (w + 1) / 2;
  const int64_t total = (int64_t)n * h * w * c;
  const size_t esz = sizeof(T) * 4;
  const bool vec = c % 4 == 0 && dy_ld % 4 == 0 && total < 0x7FFFFFFFLL && (reinterpret_cast<uintptr_t>(dy) & (esz - 1)) == 0 &&
                   (reinterpret_cast<uintptr_t>(dx) & (esz - 1)) == 0 && (reinterpret_cast<uintptr_t>(idx) & 15) == 0;
  if (vec)
    hipLaunchKernelGGL(maxpool_argmax_bwd_vec<T>, dim3(grid_for(total / 4)), dim3(256), 0, as_stream(stream), dy, dy_ld, idx, h, w, c,
                       ho, wo, dx, accumulate, (unsigned)(total / 4));
  else
    hipLaunchKernelGGL(maxpool_argmax_bwd_kernel<T>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), dy, dy_ld, idx, n, h,
                       w, c, ho, wo, dx, accumulate, total);
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

template <typename T>
static ssmi_status upsample_fwd_impl(const T* x, int32_t n, int32_t h, int32_t w, int32_t c, T* y, int32_t y_ld, void* stream) {
  SSMI_REQUIRE(x && y && n > 0 && h > 0 && w > 0 && c > 0 && y_ld >= c, SSMI_E_INVALID_ARG, "upsample_nn2x_fwd: bad argument");
  const int64_t total = (int64_t)n * h * w * c * 4;
  if (vec4_ok(x, y, c, c, y_ld, (int64_t)n * 4 * h * w * y_ld))
    hipLaunchKernelGGL(upsample_fwd_vec<T>, dim3(grid_for(total / 4)), dim3(256), 0, as_stream(stream), x, h, w, c, y, y_ld,
                       (unsigned)(total / 4));
  else
    hipLaunchKernelGGL(upsample_fwd_kernel<T>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), x, n, h, w, c, y, y_ld, total);
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

template <typename T>
static ssmi_status upsample_bwd_impl(const T* dy, int32_t dy_ld, int32_t n, int32_t h, int32_t w, int32_t c, T* dx, void* stream) {
  SSMI_REQUIRE(dy && dx && n > 0 && h > 0 && w > 0 && c > 0 && dy_ld >= c, SSMI_E_INVALID_ARG, "upsample_nn2x_bwd: bad argument");
  const int64_t total = (int64_t)n * h * w * c;
  if (vec4_ok(dy, dx, c, dy_ld, c, (int64_t)n * 4 * h * w * dy_ld))
    hipLaunchKernelGGL(upsample_bwd_vec<T>, dim3(grid_for(total / 4)), dim3(256), 0, as_stream(stream), dy, dy_ld, h, w, c, dx,
                       (unsigned)(total / 4));
  else
    hipLaunchKernelGGL(upsample_bwd_kernel<T>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), dy, dy_ld, n, h, w, c, dx, total);
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

template <typename T>
static ssmi_status copy_ld_impl(const T* src, int32_t src_ld, T* dst, int32_t dst_ld, int64_t npix, int32_t c, int32_t accumulate,
                                void* stream) {
  SSMI_REQUIRE(src && dst && npix > 0 && c > 0 && src_ld >= c && dst_ld >= c, SSMI_E_INVALID_ARG, "copy_ld: bad argument");
  const int64_t total = npix * c;
  if (vec4_ok(src, static_cast<const T*>(dst), c, src_ld, dst_ld, npix * (src_ld > dst_ld ? src_ld : dst_ld)))
    hipLaunchKernelGGL(copy_ld_vec<T>, dim3(grid_for(total / 4)), dim3(256), 0, as_stream(stream), src, src_ld, dst, dst_ld, c,
                       accumulate, (unsigned)(total / 4));
  else
    hipLaunchKernelGGL(copy_ld_kernel<T>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), src, src_ld, dst, dst_ld, c,
                       accumulate, total);
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

template <typename T>
static ssmi_status scale_nc_impl(const T* x, const float* table, T* y, int32_t n, int64_t hw, int32_t c, void* stream) {
  SSMI_REQUIRE(x && table && y && n > 0 && hw > 0 && c > 0, SSMI_E_INVALID_ARG, "scale_nc: bad argument");
  const int64_t total = (int64_t)n * hw * c;
  hipLaunchKernelGGL(scale_nc_kernel<T>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), x, table, y, hw, c, total);
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

#define BF(p) reinterpret_cast<bf16*>(p)
#define CBF(p) reinterpret_cast<const bf16*>(p)

extern "C" {

ssmi_status ssmi_maxpool_argmax_fwd(const float* x, int32_t n, int32_t h, int32_t w, int32_t c, float* y, int32_t y_ld,
                                    int32_t* idx_i32, float* idx_f32, void* stream) {
  return maxpool_argmax_fwd_impl<float>(x, n, h, w, c, y, y_ld, idx_i32, idx_f32, stream);
}
ssmi_status ssmi_maxpool_argmax_fwd_bf16(const uint16_t* x, int32_t n, int32_t h, int32_t w, int32_t c, uint16_t* y,
                                         int32_t y_ld, int32_t* idx_i32, float* idx_f32, void* stream) {
  return maxpool_argmax_fwd_impl<bf16>(CBF(x), n, h, w, c, BF(y), y_ld, idx_i32, idx_f32, stream);
}
ssmi_status ssmi_maxpool_argmax_bwd(const float* dy, int32_t dy_ld, const int32_t* idx, int32_t n, int32_t h, int32_t w,
                                    int32_t c, float* dx, int32_t accumulate, void* stream) {
  return maxpool_argmax_bwd_impl<float>(dy, dy_ld, idx, n, h, w, c, dx, accumulate, stream);
}
ssmi_status ssmi_maxpool_argmax_bwd_bf16(const uint16_t* dy, int32_t dy_ld, const int32_t* idx, int32_t n, int32_t h, int32_t w,
                                         int32_t c, uint16_t* dx, int32_t accumulate, void* stream) {
  return maxpool_argmax_bwd_impl<bf16>(CBF(dy), dy_ld, idx, n, h, w, c, BF(dx), accumulate, stream);
}
ssmi_status ssmi_upsample_nn2x_fwd_bf16(const uint16_t* x, int32_t n, int32_t h, int32_t w, int32_t c, uint16_t* y, int32_t y_ld,
                                        void* stream) {
  return upsample_fwd_impl<bf16>(CBF(x), n, h, w, c, BF(y), y_ld, stream);
}
ssmi_status ssmi_upsample_nn2x_bwd_bf16(const uint16_t* dy, int32_t dy_ld, int32_t n, int32_t h, int32_t w, int32_t c,
                                        uint16_t* dx, void* stream) {
  return upsample_bwd_impl<bf16>(CBF(dy), dy_ld, n, h, w, c, BF(dx), stream);
}
ssmi_status ssmi_copy_ld_bf16(const uint16_t* src, int32_t src_ld, uint16_t* dst, int32_t dst_ld, int64_t npix, int32_t c,
                              int32_t accumulate, void* stream) {
  return copy_ld_impl<bf16>(CBF(src), src_ld, BF(dst), dst_ld, npix, c, accumulate, stream);
}
ssmi_status ssmi_scale_nc_bf16(const uint16_t* x, const float* table, uint16_t* y, int32_t n, int64_t hw, int32_t c, void* stream) {
  return scale_nc_impl<bf16>(CBF(x), table, BF(y), n, hw, c, stream);
}
ssmi_status ssmi_cast_f32_to_bf16(const float* src, uint16_t* dst, int64_t count, void* stream) {
  SSMI_REQUIRE(src && dst && count > 0, SSMI_E_INVALID_ARG, "cast_f32_to_bf16: bad argument");
  SSMI_REQUIRE(((reinterpret_cast<uintptr_t>(src) & 15) | (reinterpret_cast<uintptr_t>(dst) & 7)) == 0, SSMI_E_INVALID_ARG,
               "cast_f32_to_bf16: pointers must be 16 / 8 byte aligned");
  hipLaunchKernelGGL(cast_f32_to_bf16_kernel, dim3(grid_for(count / 4 + 1)), dim3(256), 0, as_stream(stream), src, BF(dst), count);
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}
ssmi_status ssmi_cast_bf16_to_f32(const uint16_t* src, float* dst, int64_t count, void* stream) {
  SSMI_REQUIRE(src && dst && count > 0, SSMI_E_INVALID_ARG, "cast_bf16_to_f32: bad argument");
  SSMI_REQUIRE(((reinterpret_cast<uintptr_t>(dst) & 15) | (reinterpret_cast<uintptr_t>(src) & 7)) == 0, SSMI_E_INVALID_ARG,
               "cast_bf16_to_f32: pointers must be 16 / 8 byte aligned");
  hipLaunchKernelGGL(cast_bf16_to_f32_kernel, dim3(grid_for(count / 4 + 1)), dim3(256), 0, as_stream(stream), CBF(src), dst, count);
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

ssmi_status ssmi_maxunpool_fwd(const float* x, const int32_t* idx, int32_t n, int32_t h, int32_t w, int32_t c, float* y,
                               void* stream) {
  SSMI_REQUIRE(x && idx && y && n > 0 && h > 0 && w > 0 && c > 0, SSMI_E_INVALID_ARG, "maxunpool_fwd: bad argument");
  const int64_t pin = (int64_t)h * w * c, pout = pin * 4;
  SSMI_HIP(hipMemsetAsync(y, 0, (size_t)n * pout * sizeof(float), as_stream(stream)));
  const int64_t total = (int64_t)n * pin;
  hipLaunchKernelGGL(maxunpool_scatter_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), x, idx, pin, pout, y,
                     total, (int*)nullptr);
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

ssmi_status ssmi_maxunpool_local_fwd(const float* x, const int32_t* idx, int32_t n, int32_t h, int32_t w, int32_t c,
                                     float* y, void* stream) {
  SSMI_REQUIRE(x && idx && y && n > 0 && h > 0 && w > 0 && c > 0, SSMI_E_INVALID_ARG, "maxunpool_local_fwd: bad argument");
  const int64_t total = (int64_t)n * h * w * c * 4;
  hipLaunchKernelGGL(maxunpool_local_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), x, idx, n, h, w, c, y,
                     total);
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

ssmi_status ssmi_maxunpool_bwd(const float* dy, const int32_t* idx, int32_t n, int32_t h, int32_t w, int32_t c,
                               float* dx, void* stream) {
  SSMI_REQUIRE(dy && idx && dx && n > 0 && h > 0 && w > 0 && c > 0, SSMI_E_INVALID_ARG, "maxunpool_bwd: bad argument");
  const int64_t pin = (int64_t)h * w * c, pout = pin * 4;
  const int64_t total = (int64_t)n * pin;
  hipLaunchKernelGGL(maxunpool_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), dy, idx, pin, pout, dx,
                     total);
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

ssmi_status ssmi_idx_f32_to_i32(const float* src, int32_t* dst, int64_t count, void* stream) {
  SSMI_REQUIRE(src && dst && count > 0, SSMI_E_INVALID_ARG, "idx_f32_to_i32: bad argument");
  hipLaunchKernelGGL(idx_f32_to_i32_kernel, dim3(grid_for(count)), dim3(256), 0, as_stream(stream), src, dst, count);
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

ssmi_status ssmi_upsample_nn2x_fwd(const float* x, int32_t n, int32_t h, int32_t w, int32_t c, float* y, int32_t y_ld,
                                   void* stream) {
  return upsample_fwd_impl<float>(x, n, h, w, c, y, y_ld, stream);
}

ssmi_status ssmi_upsample_nn2x_bwd(const float* dy, int32_t dy_ld, int32_t n, int32_t h, int32_t w, int32_t c, float* dx,
                                   void* stream) {
  return upsample_bwd_impl<float>(dy, dy_ld, n, h, w, c, dx, stream);
}

ssmi_status ssmi_copy_ld(const float* src, int32_t src_ld, float* dst, int32_t dst_ld, int64_t npix, int32_t c,
                         int32_t accumulate, void* stream) {
  return copy_ld_impl<float>(src, src_ld, dst, dst_ld, npix, c, accumulate, stream);
}

ssmi_status ssmi_scale_nc(const float* x, const float* table, float* y, int32_t n, int64_t hw, int32_t c, void* stream) {
  return scale_nc_impl<float>(x, table, y, n, hw, c, stream);
}

ssmi_status ssmi_fill_f32(float* dst, float value, int64_t count, void* stream) {
  SSMI_REQUIRE(dst && count > 0, SSMI_E_INVALID_ARG, "fill_f32: bad argument");
  hipLaunchKernelGGL(fill_kernel, dim3(grid_for(count)), dim3(256), 0, as_stream(stream), dst, value, count);
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

ssmi_status ssmi_global_avgpool_fwd(const float* x, int32_t n, int64_t hw, int32_t c, float* y, void* stream) {
  SSMI_REQUIRE(x && y && n > 0 && hw > 0 && c > 0, SSMI_E_INVALID_ARG, "global_avgpool_fwd: bad argument");
  hipLaunchKernelGGL(global_avgpool_fwd_kernel, dim3((unsigned)n, (unsigned)((c + 63) / 64)), dim3(256), 0, as_stream(stream), x,
                     hw, c, y);
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

ssmi_status ssmi_global_avgpool_bwd(const float* dy, int32_t n, int64_t hw, int32_t c, float* dx, void* stream) {
  SSMI_REQUIRE(dy && dx && n > 0 && hw > 0 && c > 0, SSMI_E_INVALID_ARG, "global_avgpool_bwd: bad argument");
  const int64_t total = (int64_t)n * hw * c;
  hipLaunchKernelGGL(global_avgpool_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), dy, hw, c, dx, total);
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

}  // extern "C"

// BatchNormalization (+PReLU / LeakyReLU) forward/backward and the fused bottleneck merge.
// All kernels stream NHWC tensors with 16-byte accesses: thread i handles channels [4*(i % C4), +4) of pixel i / C4,
// so consecutive lanes touch consecutive 16-byte chunks.  Per-channel reductions use a fixed (thread, block) ->
// partial-sum mapping followed by a double-precision finalize kernel (deterministic).
#include "common.h"

namespace ssmi {

static constexpr int kMaxRedBlocks = 512;

static inline int red_blocks(int64_t npix, int c4) {
  const int ppi = 256 / c4;
  int64_t b = ceil_div64(npix, (int64_t)ppi * 8);
  if (b > kMaxRedBlocks) b = kMaxRedBlocks;
  if (b < 1) b = 1;
  return (int)b;
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// block-level reduction of NV float4 accumulators over the threads that share a channel group.
// sm must hold 256 * NV float4.  Result for channel group c4 is returned by thread tid == c4 (tid < C4).
template <int NV>
__device__ __forceinline__ void block_reduce_groups(f32x4 (&v)[NV], f32x4* sm, int c4n, int ppi, bool active) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < NV; ++k) sm[k * 256 + tid] = active ? v[k] : f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  if (tid < c4n) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
      for (int q = 0; q < ppi; ++q) s += sm[k * 256 + q * c4n + tid];
      v[k] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// statistics
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_partial(const T* __restrict__ x, int64_t npix, int c4n, int ld,
                                                        float* __restrict__ part) {
  __shared__ f32x4 sm[2 * 256];
  const int ppi = 256 / c4n;
  const int c4 = threadIdx.x % c4n, pl = threadIdx.x / c4n;
  const bool active = pl < ppi;
  f32x4 v[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if (active) {
    // eight pixels per round with all loads in flight at once (a one-load-per-iteration loop costs one memory round
    // trip per pixel); pixels past the end read zeros through the buffer bounds check
    const __amdgpu_buffer_rsrc_t rx = Act<T>::rsrc(x, (unsigned)(npix * ld * 4));
    const int64_t stride = (int64_t)gridDim.x * ppi;
    for (int64_t p0 = (int64_t)blockIdx.x * ppi + pl; p0 < npix; p0 += stride * 8) {
      f32x4 a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t p = p0 + u * stride;
        a[u] = Act<T>::ld4(rx, p < npix ? (unsigned)((p * ld + c4 * 4) * 4) : kOOB);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        v[0] += a[u];
        v[1] += a[u] * a[u];
      }
    }
  }
  block_reduce_groups<2>(v, sm, c4n, ppi, active);
  if (threadIdx.x < c4n) {
    const int c = c4n * 4;
    st4(part + ((int64_t)blockIdx.x * 2 + 0) * c + threadIdx.x * 4, v[0]);
    st4(part + ((int64_t)blockIdx.x * 2 + 1) * c + threadIdx.x * 4, v[1]);
  }
}

// Sums the rows of per-block partials for CH channels per workgroup: thread (chl = tid % CH, slice = tid / CH) adds
// the rows slice, slice + 256/CH, ... in double (eight rows per round, their loads issued together; rows past the end
// and channels past c read zeros through the buffer bounds check), then the slices are combined through LDS in a fixed
// order.  Result valid for tid < CH.  Fewer channels per workgroup = more slices = fewer serial rounds for many rows.
template <int NR, int CH>
__device__ __forceinline__ void final_reduce(const float* __restrict__ part, int nblocks, int c, int ch, double (&out)[NR]) {
  __shared__ double sm[NR][256];
  constexpr int NS = 256 / CH;
  const int slice = threadIdx.x / CH;
  double acc[NR];
#pragma unroll
  for (int k = 0; k < NR; ++k) acc[k] = 0.0;
  {
    const __amdgpu_buffer_rsrc_t rp = make_rsrc(part, (unsigned)((int64_t)nblocks * NR * c * 4));
    const unsigned cho = ch < c ? (unsigned)ch * 4u : kOOB;
    const unsigned rowb = (unsigned)(NR * c) * 4u;
    for (int b0 = slice; b0 < nblocks; b0 += NS * 8) {
      float v[8][NR];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int b = b0 + NS * u;
        const unsigned ro = b < nblocks ? cho + (unsigned)b * rowb : kOOB;
#pragma unroll
        for (int k = 0; k < NR; ++k) v[u][k] = bload(rp, ro + (unsigned)(k * c) * 4u);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < NR; ++k) acc[k] += (double)v[u][k];
    }
  }
#pragma unroll
  for (int k = 0; k < NR; ++k) sm[k][threadIdx.x] = acc[k];
  __syncthreads();
  if (threadIdx.x < CH) {
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      double s = 0.0;
      for (int q = 0; q < NS; ++q) s += sm[k][q * CH + threadIdx.x];
      out[k] = s;
    }
  }
}

template <int CH>
__global__ __launch_bounds__(256) void bn_stats_final(const float* __restrict__ part, int nblocks, int c, int64_t npix,
                                                      float eps, float momentum, int unbiased,
                                                      float* __restrict__ moving_mean, float* __restrict__ moving_var,
                                                      float* __restrict__ save_mean, float* __restrict__ save_invstd) {
  const int ch = blockIdx.x * CH + (threadIdx.x % CH);
  double r[2];
  final_reduce<2, CH>(part, nblocks, c, ch, r);
  if (threadIdx.x >= CH || ch >= c) return;
  const double n = (double)npix;
  const double mean = r[0] / n;
  double var = r[1] / n - mean * mean;
  if (var < 0.0) var = 0.0;
  save_mean[ch] = (float)mean;
  save_invstd[ch] = (float)(1.0 / sqrt(var + (double)eps));
  if (moving_mean) moving_mean[ch] = momentum * moving_mean[ch] + (1.f - momentum) * (float)mean;
  if (moving_var) {
    const double mv = (unbiased && npix > 1) ? var * n / (n - 1.0) : var;
    moving_var[ch] = momentum * moving_var[ch] + (1.f - momentum) * (float)mv;
  }
}

ssmi_status launch_bn_stats_final(const float* part, int rows, int c, int64_t npix, float eps, float momentum,
                                  int unbiased_moving, float* moving_mean, float* moving_var, float* save_mean,
                                  float* save_invstd, hipStream_t stream) {
  SSMI_REQUIRE((int64_t)rows * 2 * c * 4 < 0x7FFFFFFFLL, SSMI_E_UNSUPPORTED, "bn statistics: partial-sum table too large");
  static const bool ablate = getenv("SSMI_ABLATE_BN_FINAL") != nullptr;      // timing experiment only: results are meaningless
  if (ablate) return SSMI_OK;
#define LF(CH) hipLaunchKernelGGL((bn_stats_final<CH>), dim3((c + CH - 1) / CH), dim3(256), 0, stream, part, rows, c, npix, \
                                  eps, momentum, unbiased_moving, moving_mean, moving_var, save_mean, save_invstd)
  if (rows <= 512) LF(8); else if (rows <= 1024) LF(4); else if (rows <= 2048) LF(2); else LF(1);
#undef LF
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

__global__ void bn_infer_stats_kernel(const float* __restrict__ mm, const float* __restrict__ mv, int c, float eps,
                                      float* __restrict__ save_mean, float* __restrict__ save_invstd) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  save_mean[ch] = mm[ch];
  save_invstd[ch] = 1.f / sqrtf(mv[ch] + eps);
}

// ------------------------------------------------------------------------------------------------
// activation helpers
// ------------------------------------------------------------------------------------------------
template <int ACT>
__device__ __forceinline__ float act_fwd(float z, float a) {
  if (ACT == SSMI_ACT_NONE) return z;
  return z > 0.f ? z : a * z;  // PReLU: a = alpha[c]; Leaky: a = slope
}
template <int ACT>
__device__ __forceinline__ float act_grad(float z, float a) {
  if (ACT == SSMI_ACT_NONE) return 1.f;
  return z > 0.f ? 1.f : (z < 0.f ? a : 0.f);
}

// Elementwise kernels: thread (c4 = tid % c4n, pl = tid / c4n) owns channel group c4 of the pixels pl, pl + stride, ...
// so its per-channel parameters are loaded ONCE (re-loading five or more 16-byte parameter vectors for every 16-byte
// payload load made the address unit, not HBM, the limit), and it handles four pixels per round with all payload loads
// issued before the first store.
template <typename T, int ACT>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const T* __restrict__ x, int ld, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ alpha,
                                                         float slope, const float* __restrict__ drop, int64_t hw,
                                                         T* __restrict__ y, int y_ld, int64_t npix, int c4n) {
  const unsigned ppi = 256u / (unsigned)c4n;
  const unsigned c4 = threadIdx.x % (unsigned)c4n, pl = threadIdx.x / (unsigned)c4n;
  if (pl >= ppi) return;
  const int c = (int)c4 * 4;
  const f32x4 m = ld4(mean + c), is = ld4(invstd + c), g = ld4(gamma + c), b = ld4(beta + c);
  f32x4 a = {slope, slope, slope, slope};
  if (ACT == SSMI_ACT_PRELU) a = ld4(alpha + c);
  auto one = [&](const f32x4& xv, unsigned p) {
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = act_fwd<ACT>(g[j] * ((xv[j] - m[j]) * is[j]) + b[j], a[j]);
    if (drop) o = o * ld4(drop + (p / (unsigned)hw) * (unsigned)(c4n * 4) + c);
    return o;
  };
  const unsigned stride = gridDim.x * ppi, n = (unsigned)npix;      // npix * ld < 2^29
  unsigned p = blockIdx.x * ppi + pl;
  for (; (uint64_t)p + 3ull * stride < (uint64_t)n; p += 4u * stride) {
    f32x4 xv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) xv[u] = Act<T>::gld4(x + (p + (unsigned)u * stride) * (unsigned)ld + c);
#pragma unroll
    for (int u = 0; u < 4; ++u) xv[u] = one(xv[u], p + (unsigned)u * stride);
#pragma unroll
    for (int u = 0; u < 4; ++u) Act<T>::gst4(y + (p + (unsigned)u * stride) * (unsigned)y_ld + c, xv[u]);
  }
  for (; p < n; p += stride) Act<T>::gst4(y + p * (unsigned)y_ld + c, one(Act<T>::gld4(x + p * (unsigned)ld + c), p));
}

// backward pass 1: per-channel sums of dz, dz*xhat and (PReLU) dy*min(z,0)
template <typename T, int ACT>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce(const T* __restrict__ dy, const T* __restrict__ x, int ld,
                                                         const float* __restrict__ mean, const float* __restrict__ invstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ alpha, float slope,
                                                         const float* __restrict__ drop, int64_t hw, int dy_ld,
                                                         int64_t npix, int c4n, float* __restrict__ part) {
  __shared__ f32x4 sm[3 * 256];
  const int ppi = 256 / c4n;
  const int c4 = threadIdx.x % c4n, pl = threadIdx.x / c4n;
  const bool active = pl < ppi;
  const int c = c4 * 4;
  f32x4 v[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if (active) {
    const f32x4 m = ld4(mean + c), is = ld4(invstd + c), g = ld4(gamma + c), b = ld4(beta + c);
    f32x4 a = {slope, slope, slope, slope};
    if (ACT == SSMI_ACT_PRELU) a = ld4(alpha + c);
    // four pixels per round, their eight loads in flight together
    const __amdgpu_buffer_rsrc_t rx = Act<T>::rsrc(x, (unsigned)(npix * ld * 4));
    const __amdgpu_buffer_rsrc_t rd = Act<T>::rsrc(dy, (unsigned)(npix * dy_ld * 4));
    const int64_t stride = (int64_t)gridDim.x * ppi;
    for (int64_t p0 = (int64_t)blockIdx.x * ppi + pl; p0 < npix; p0 += stride * 4) {
      f32x4 xv[4], d[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t p = p0 + u * stride;
        const bool ok = p < npix;
        xv[u] = Act<T>::ld4(rx, ok ? (unsigned)((p * ld + c) * 4) : kOOB);
        d[u] = Act<T>::ld4(rd, ok ? (unsigned)((p * dy_ld + c) * 4) : kOOB);   // zero gradient for pixels past the end
      }
      if (drop) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int64_t p = p0 + u * stride;
          d[u] = d[u] * ld4(drop + ((p < npix ? p : npix - 1) / hw) * (c4n * 4) + c);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xh = (xv[u][j] - m[j]) * is[j];
          const float z = g[j] * xh + b[j];
          const float dz = d[u][j] * act_grad<ACT>(z, a[j]);
          v[0][j] += dz;
          v[1][j] += dz * xh;
          if (ACT == SSMI_ACT_PRELU) v[2][j] += d[u][j] * (z < 0.f ? z : 0.f);
        }
      }
    }
  }
  block_reduce_groups<3>(v, sm, c4n, ppi, active);
  if (threadIdx.x < c4n) {
    const int cc = c4n * 4;
#pragma unroll
    for (int k = 0; k < 3; ++k) st4(part + ((int64_t)blockIdx.x * 3 + k) * cc + threadIdx.x * 4, v[k]);
  }
}

// finalize: parameter gradients + coefficients for the apply pass: coef[0]=gamma*invstd, coef[1]=mean(dz), coef[2]=mean(dz*xhat)
template <int CH>
__global__ __launch_bounds__(256) void bn_bwd_final(const float* __restrict__ part, int nblocks, int c, int64_t npix,
                                                    const float* __restrict__ gamma, const float* __restrict__ invstd,
                                                    float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                    float* __restrict__ dalpha, float* __restrict__ coef) {
  const int ch = blockIdx.x * CH + (threadIdx.x % CH);
  double r[3];
  final_reduce<3, CH>(part, nblocks, c, ch, r);
  if (threadIdx.x >= CH || ch >= c) return;
  if (dbeta) dbeta[ch] = (float)r[0];
  if (dgamma) dgamma[ch] = (float)r[1];
  if (dalpha) dalpha[ch] = (float)r[2];
  coef[ch] = gamma[ch] * invstd[ch];
  coef[c + ch] = (float)(r[0] / (double)npix);
  coef[2 * c + ch] = (float)(r[1] / (double)npix);
}

template <typename T, int ACT>
__global__ __launch_bounds__(256) void bn_act_bwd_apply(const T* __restrict__ dy, const T* __restrict__ x, int ld,
                                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ alpha, float slope,
                                                        const float* __restrict__ drop, int64_t hw, int dy_ld,
                                                        const float* __restrict__ coef, T* dx,
                                                        int64_t npix, int c4n) {
  const int cc = c4n * 4;
  const unsigned ppi = 256u / (unsigned)c4n;
  const unsigned c4 = threadIdx.x % (unsigned)c4n, pl = threadIdx.x / (unsigned)c4n;
  if (pl >= ppi) return;
  const int c = (int)c4 * 4;
  const f32x4 m = ld4(mean + c), is = ld4(invstd + c), g = ld4(gamma + c), b = ld4(beta + c);
  const f32x4 k1 = ld4(coef + c), m1 = ld4(coef + cc + c), m2 = ld4(coef + 2 * cc + c);
  f32x4 a = {slope, slope, slope, slope};
  if (ACT == SSMI_ACT_PRELU) a = ld4(alpha + c);
  auto one = [&](const f32x4& xv, f32x4 d, unsigned p) {
    if (drop) d = d * ld4(drop + (p / (unsigned)hw) * (unsigned)cc + c);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float xh = (xv[j] - m[j]) * is[j];
      const float z = g[j] * xh + b[j];
      const float dz = d[j] * act_grad<ACT>(z, a[j]);
      o[j] = k1[j] * (dz - m1[j] - xh * m2[j]);
    }
    return o;
  };
  const unsigned stride = gridDim.x * ppi, n = (unsigned)npix;
  unsigned p = blockIdx.x * ppi + pl;
  for (; (uint64_t)p + 3ull * stride < (uint64_t)n; p += 4u * stride) {
    f32x4 xv[4], d[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned q = p + (unsigned)u * stride;
      xv[u] = Act<T>::gld4(x + q * (unsigned)ld + c);
      d[u] = Act<T>::gld4(dy + q * (unsigned)dy_ld + c);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) d[u] = one(xv[u], d[u], p + (unsigned)u * stride);
#pragma unroll
    for (int u = 0; u < 4; ++u) Act<T>::gst4(dx + (p + (unsigned)u * stride) * (unsigned)ld + c, d[u]);   // dx may alias dy: all loads are done
  }
  for (; p < n; p += stride)
    Act<T>::gst4(dx + p * (unsigned)ld + c, one(Act<T>::gld4(x + p * (unsigned)ld + c), Act<T>::gld4(dy + p * (unsigned)dy_ld + c), p));
}

// ------------------------------------------------------------------------------------------------
// bottleneck merge
// ------------------------------------------------------------------------------------------------
struct MergeArgs {
  int n, h, w, c, mode, c_other;
  const void* main;        // activation type
  const float *mean3, *invstd3, *gamma3, *beta3;
  const float* drop;
  const void* other;       // activation type
  const float *mean_o, *invstd_o, *gamma_o, *beta_o;
  const int32_t* idx;
  const float* alpha;
};

// per-channel parameters of a thread's channel group, loaded once
struct MergeParams {
  f32x4 m, is, g, b;         // BN3
  f32x4 mo, iso, go, bo;     // BN of the skip branch (modes 2, 3)
};
template <int MODE>
__device__ __forceinline__ MergeParams merge_params(const MergeArgs& a, int c) {
  MergeParams q;
  q.m = ld4(a.mean3 + c); q.is = ld4(a.invstd3 + c); q.g = ld4(a.gamma3 + c); q.b = ld4(a.beta3 + c);
  q.mo = q.iso = q.go = q.bo = f32x4{0.f, 0.f, 0.f, 0.f};
  if (MODE >= 2) { q.mo = ld4(a.mean_o + c); q.iso = ld4(a.invstd_o + c); q.go = ld4(a.gamma_o + c); q.bo = ld4(a.beta_o + c); }
  return q;
}

// Raw operands of element group (p, c..c+3): the main-branch value, the skip-branch value and (mode 3) the argmax indices.
// Loads only, so that a caller can issue several pixels' loads back to back.
template <typename T, int MODE>
struct MergeRaw {
  f32x4 mv, ov, dropv;
  int4 iv;
  int flat;
  bool has_other;
};
template <typename T, int MODE>
__device__ __forceinline__ MergeRaw<T, MODE> merge_load(const MergeArgs& a, unsigned p, int c) {
  MergeRaw<T, MODE> r;
  const unsigned hw = (unsigned)(a.h * a.w);     // every tensor is below 2 GiB: 32-bit element arithmetic
  const unsigned n_ = p / hw;
  const T* main_ = static_cast<const T*>(a.main);
  const T* other_ = static_cast<const T*>(a.other);
  r.mv = Act<T>::gld4(main_ + p * a.c + c);
  r.dropv = f32x4{1.f, 1.f, 1.f, 1.f};
  if (a.drop) r.dropv = ld4(a.drop + n_ * (unsigned)a.c + c);
  r.ov = f32x4{0.f, 0.f, 0.f, 0.f};
  r.iv = int4{0, 0, 0, 0};
  r.flat = 0;
  r.has_other = true;
  if (MODE == 0) {
    r.ov = Act<T>::gld4(other_ + p * a.c + c);
  } else if (MODE == 1) {
    r.has_other = c < a.c_other;
    if (r.has_other) r.ov = Act<T>::gld4(other_ + p * a.c_other + c);
  } else {
    const unsigned rem = p - n_ * hw;
    const int y = (int)(rem / (unsigned)a.w), x = (int)(rem - (unsigned)y * (unsigned)a.w);
    const int h2 = a.h >> 1, w2 = a.w >> 1;
    const unsigned q = (n_ * (unsigned)h2 + (unsigned)(y >> 1)) * (unsigned)w2 + (unsigned)(x >> 1);
    r.ov = Act<T>::gld4(other_ + q * a.c + c);
    if (MODE == 3) {
      r.iv = *reinterpret_cast<const int4*>(a.idx + q * a.c + c);
      r.flat = (y * a.w + x) * a.c + c;
    }
  }
  return r;
}
// pre-activation sum s and the BN3 intermediate xhat3 from the raw operands
template <typename T, int MODE>
__device__ __forceinline__ f32x4 merge_sum(const MergeParams& q, const MergeRaw<T, MODE>& r, f32x4& xh3) {
  f32x4 s;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    xh3[j] = (r.mv[j] - q.m[j]) * q.is[j];
    s[j] = (q.g[j] * xh3[j] + q.b[j]) * r.dropv[j];
  }
  if (MODE <= 1) {
    s += r.ov;            // zeros where the pooled skip tensor has no such channel
  } else {
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = q.go[j] * ((r.ov[j] - q.mo[j]) * q.iso[j]) + q.bo[j];
    if (MODE == 3) {
      o[0] = (r.iv.x == r.flat) ? o[0] : 0.f;
      o[1] = (r.iv.y == r.flat + 1) ? o[1] : 0.f;
      o[2] = (r.iv.z == r.flat + 2) ? o[2] : 0.f;
      o[3] = (r.iv.w == r.flat + 3) ? o[3] : 0.f;
    }
    s += o;
  }
  return s;
}

template <typename T, int MODE>
__global__ __launch_bounds__(256) void merge_fwd_kernel(MergeArgs a, T* __restrict__ out, int64_t npix, int c4n) {
  const unsigned ppi = 256u / (unsigned)c4n;
  const unsigned c4 = threadIdx.x % (unsigned)c4n, pl = threadIdx.x / (unsigned)c4n;
  if (pl >= ppi) return;
  const int c = (int)c4 * 4;
  const MergeParams q = merge_params<MODE>(a, c);
  const f32x4 al = ld4(a.alpha + c);
  auto finish = [&](const MergeRaw<T, MODE>& r) {
    f32x4 xh3;
    const f32x4 s = merge_sum<T, MODE>(q, r, xh3);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = s[j] > 0.f ? s[j] : al[j] * s[j];
    return o;
  };
  const unsigned stride = gridDim.x * ppi, n = (unsigned)npix;
  unsigned p = blockIdx.x * ppi + pl;
  for (; (uint64_t)p + 3ull * stride < (uint64_t)n; p += 4u * stride) {
    MergeRaw<T, MODE> r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) r[u] = merge_load<T, MODE>(a, p + (unsigned)u * stride, c);
    f32x4 o[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) o[u] = finish(r[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) Act<T>::gst4(out + ((p + (unsigned)u * stride) * (unsigned)c4n + c4) * 4u, o[u]);
  }
  for (; p < n; p += stride) Act<T>::gst4(out + (p * (unsigned)c4n + c4) * 4u, finish(merge_load<T, MODE>(a, p, c)));
}

// backward pass 1: dsum = dout * prelu'(s); partial sums of dz = dsum*drop, dz*xhat3 and dout*min(s,0)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void merge_bwd_reduce(MergeArgs a, const T* __restrict__ dout,
                                                        T* __restrict__ dsum, int64_t npix, int c4n,
                                                        float* __restrict__ part) {
  __shared__ f32x4 sm[3 * 256];
  const int ppi = 256 / c4n;
  const int c4 = threadIdx.x % c4n, pl = threadIdx.x / c4n;
  const bool active = pl < ppi;
  const int c = c4 * 4;
  f32x4 v[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if (active) {
    const f32x4 al = ld4(a.alpha + c);
    const MergeParams q = merge_params<MODE>(a, c);
    auto accumulate = [&](const MergeRaw<T, MODE>& r, const f32x4& d, f32x4& ds) {
      f32x4 xh3;
      const f32x4 s = merge_sum<T, MODE>(q, r, xh3);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ds[j] = d[j] * (s[j] > 0.f ? 1.f : (s[j] < 0.f ? al[j] : 0.f));
        const float dz = ds[j] * r.dropv[j];
        v[0][j] += dz;
        v[1][j] += dz * xh3[j];
        v[2][j] += d[j] * (s[j] < 0.f ? s[j] : 0.f);
      }
    };
    const unsigned stride = gridDim.x * (unsigned)ppi;
    unsigned p = blockIdx.x * (unsigned)ppi + (unsigned)pl;
    // four pixels per round: all their loads are issued before the first result is stored (a one-pixel loop costs one
    // memory round trip per pixel and leaves the launch far below the bandwidth its 512 workgroups could draw)
    for (; (uint64_t)p + 3ull * stride < (uint64_t)npix; p += 4u * stride) {
      MergeRaw<T, MODE> r[4];
      f32x4 d[4], ds[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned pu = p + (unsigned)u * stride;
        r[u] = merge_load<T, MODE>(a, pu, c);
        d[u] = Act<T>::gld4(dout + (pu * (unsigned)c4n + (unsigned)c4) * 4u);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) accumulate(r[u], d[u], ds[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u) Act<T>::gst4(dsum + ((p + (unsigned)u * stride) * (unsigned)c4n + (unsigned)c4) * 4u, ds[u]);
    }
    for (; p < (unsigned)npix; p += stride) {
      f32x4 ds;
      const MergeRaw<T, MODE> r = merge_load<T, MODE>(a, p, c);
      const f32x4 d = Act<T>::gld4(dout + (p * (unsigned)c4n + (unsigned)c4) * 4u);
      accumulate(r, d, ds);
      Act<T>::gst4(dsum + (p * (unsigned)c4n + (unsigned)c4) * 4u, ds);
    }
  }
  block_reduce_groups<3>(v, sm, c4n, ppi, active);
  if (threadIdx.x < c4n) {
    const int cc = c4n * 4;
#pragma unroll
    for (int k = 0; k < 3; ++k) st4(part + ((int64_t)blockIdx.x * 3 + k) * cc + threadIdx.x * 4, v[k]);
  }
}

// backward pass 2: dmain = gamma3*invstd3 * (dsum*drop - mean(dz) - xhat3*mean(dz*xhat3))
template <typename T>
__global__ __launch_bounds__(256) void merge_bwd_apply(const T* __restrict__ dsum, const T* __restrict__ main,
                                                       const float* __restrict__ mean3, const float* __restrict__ invstd3,
                                                       const float* __restrict__ drop, const float* __restrict__ coef,
                                                       T* __restrict__ dmain, int64_t npix, int c4n, int64_t hw) {
  const int cc = c4n * 4;
  const unsigned ppi = 256u / (unsigned)c4n;
  const unsigned c4 = threadIdx.x % (unsigned)c4n, pl = threadIdx.x / (unsigned)c4n;
  if (pl >= ppi) return;
  const int c = (int)c4 * 4;
  const f32x4 m = ld4(mean3 + c), is = ld4(invstd3 + c);
  const f32x4 k1 = ld4(coef + c), m1 = ld4(coef + cc + c), m2 = ld4(coef + 2 * cc + c);
  auto one = [&](const f32x4& ds, const f32x4& mv, unsigned p) {
    f32x4 dv = {1.f, 1.f, 1.f, 1.f};
    if (drop) dv = ld4(drop + (p / (unsigned)hw) * (unsigned)cc + c);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float xh = (mv[j] - m[j]) * is[j];
      o[j] = k1[j] * (ds[j] * dv[j] - m1[j] - xh * m2[j]);
    }
    return o;
  };
  const unsigned stride = gridDim.x * ppi, n = (unsigned)npix;
  unsigned p = blockIdx.x * ppi + pl;
  for (; (uint64_t)p + 3ull * stride < (uint64_t)n; p += 4u * stride) {
    f32x4 ds[4], mv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned e = ((p + (unsigned)u * stride) * (unsigned)c4n + c4) * 4u;
      ds[u] = Act<T>::gld4(dsum + e);
      mv[u] = Act<T>::gld4(main + e);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) ds[u] = one(ds[u], mv[u], p + (unsigned)u * stride);
#pragma unroll
    for (int u = 0; u < 4; ++u) Act<T>::gst4(dmain + ((p + (unsigned)u * stride) * (unsigned)c4n + c4) * 4u, ds[u]);
  }
  for (; p < n; p += stride) {
    const unsigned e = (p * (unsigned)c4n + c4) * 4u;
    Act<T>::gst4(dmain + e, one(Act<T>::gld4(dsum + e), Act<T>::gld4(main + e), p));
  }
}

// gradient w.r.t. BN_o's output for the upsample (2x2 sum) / unpool (gather at idx) skip branches
template <typename T, int MODE>
__global__ __launch_bounds__(256) void merge_bwd_other(const T* __restrict__ dsum, const int32_t* __restrict__ idx,
                                                       T* __restrict__ dother, int n, int h, int w, int c4n,
                                                       int64_t total4) {
  const int c = c4n * 4;
  const int h2 = h >> 1, w2 = w >> 1;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)total4; i += gridDim.x * 256u) {   // total4 < 2^31
    const unsigned q = i / (unsigned)c4n;
    const int ch = (int)(i - q * (unsigned)c4n) * 4;
    const int x2 = (int)(q % w2);
    const int64_t t = q / w2;
    const int y2 = (int)(t % h2);
    const int n_ = (int)(t / h2);
    f32x4 o;
    if (MODE == 2) {
      const T* base = dsum + (((int64_t)n_ * h + 2 * y2) * w + 2 * x2) * c + ch;
      o = Act<T>::gld4(base) + Act<T>::gld4(base + c) + Act<T>::gld4(base + (int64_t)w * c) + Act<T>::gld4(base + (int64_t)w * c + c);
    } else {
      const int4 iv = *reinterpret_cast<const int4*>(idx + (size_t)i * 4);
      const T* base = dsum + (int64_t)n_ * h * w * c;
      o[0] = Act<T>::gld(base + iv.x);
      o[1] = Act<T>::gld(base + iv.y);
      o[2] = Act<T>::gld(base + iv.z);
      o[3] = Act<T>::gld(base + iv.w);
    }
    Act<T>::gst4(dother + (size_t)i * 4, o);
  }
}

static ssmi_status launch_bn_bwd_final(const float* part, int rows, int c, int64_t npix, const float* gamma,
                                       const float* invstd, float* dgamma, float* dbeta, float* dalpha, float* coef,
                                       hipStream_t st) {
  SSMI_REQUIRE((int64_t)rows * 3 * c * 4 < 0x7FFFFFFFLL, SSMI_E_UNSUPPORTED, "bn backward: partial-sum table too large");
  static const bool ablate = getenv("SSMI_ABLATE_BN_FINAL") != nullptr;      // timing experiment only: results are meaningless
  if (ablate) return SSMI_OK;
#define LF(CH) hipLaunchKernelGGL((bn_bwd_final<CH>), dim3((c + CH - 1) / CH), dim3(256), 0, st, part, rows, c, npix, gamma, \
                                  invstd, dgamma, dbeta, dalpha, coef)
  if (rows <= 512) LF(8); else if (rows <= 1024) LF(4); else if (rows <= 2048) LF(2); else LF(1);
#undef LF
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

// grid of the pixel-strided elementwise kernels: one round of four pixels per thread, at most 4096 workgroups
static inline int ew_blocks_px(int64_t npix, int c4n) {
  const int ppi = 256 / c4n;
  int64_t b = ceil_div64(npix, (int64_t)ppi * 4);
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

static inline int ew_blocks(int64_t total4) {
  int64_t b = ceil_div64(total4, 256 * 4);
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

static ssmi_status check_c(int c, const char* who) {
  SSMI_REQUIRE(c > 0 && c % 4 == 0 && c <= 1024, SSMI_E_UNSUPPORTED, "%s: channel count %d must be a multiple of 4 and <= 1024", who, c);
  return SSMI_OK;
}

template <typename T>
ssmi_status launch_bn_bwd_tail(const float* part, int rows, const T* dy, int ld_dy, const T* x, int ld_x,
                               const float* mean, const float* invstd, const float* gamma, const float* beta, int act,
                               const float* alpha, float slope, const float* drop, int64_t hw, T* dx, float* dgamma,
                               float* dbeta, float* dalpha, int64_t npix, int c, float* coef, hipStream_t st) {
  ssmi_status e = launch_bn_bwd_final(part, rows, c, npix, gamma, invstd, dgamma, dbeta,
                                      act == SSMI_ACT_PRELU ? dalpha : nullptr, coef, st);
  if (e) return e;
  const int c4n = c / 4;
  const int nb2 = ew_blocks_px(npix, c4n);
#define L2(A) hipLaunchKernelGGL((bn_act_bwd_apply<T, A>), dim3(nb2), dim3(256), 0, st, dy, x, ld_x, mean, invstd, gamma, beta, alpha, slope, drop, hw, ld_dy, coef, dx, npix, c4n)
  if (act == SSMI_ACT_NONE) L2(SSMI_ACT_NONE); else if (act == SSMI_ACT_PRELU) L2(SSMI_ACT_PRELU); else L2(SSMI_ACT_LEAKY);
#undef L2
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}
template ssmi_status launch_bn_bwd_tail<float>(const float*, int, const float*, int, const float*, int, const float*, const float*,
                                               const float*, const float*, int, const float*, float, const float*, int64_t, float*,
                                               float*, float*, float*, int64_t, int, float*, hipStream_t);
template ssmi_status launch_bn_bwd_tail<bf16>(const float*, int, const bf16*, int, const bf16*, int, const float*, const float*,
                                              const float*, const float*, int, const float*, float, const float*, int64_t, bf16*,
                                              float*, float*, float*, int64_t, int, float*, hipStream_t);


// Inference-phase statistics of EVERY BatchNormalization layer in one launch: the moving statistics live in one arena
// ([moving_mean | moving_variance] per layer); out mirrors it with mean copied and variance turned into 1/sqrt(var + eps).
__global__ __launch_bounds__(256) void bn_infer_stats_arena_kernel(const float* __restrict__ state, const uint8_t* __restrict__ is_var,
                                                                   int64_t count, float eps, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256)
    out[i] = is_var[i] ? 1.f / sqrtf(state[i] + eps) : state[i];
}

}  // namespace ssmi

using namespace ssmi;

extern "C" {

size_t ssmi_bn_workspace_bytes(int64_t npix, int32_t c) {
  (void)npix;
  // partial sums (3 rows per block) + 3 coefficient rows
  return ((size_t)kMaxRedBlocks * 3 + 3) * (size_t)c * sizeof(float);
}

}  // extern "C" (templates need C++ linkage)

template <typename T>
static ssmi_status bn_train_stats_impl(const T* x, int64_t npix, int32_t c, int32_t ld, float eps, float momentum,
                                       int32_t unbiased_moving, float* moving_mean, float* moving_var, float* save_mean,
                                       float* save_invstd, void* ws, size_t ws_bytes, void* stream) {
  ssmi_status s = check_c(c, "bn_train_stats");
  if (s) return s;
  SSMI_REQUIRE(x && save_mean && save_invstd && npix > 0 && ld >= c && ld % 4 == 0, SSMI_E_INVALID_ARG, "bn_train_stats: bad argument");
  SSMI_REQUIRE(npix * ld * 4 < 0x7FFFFFFFLL, SSMI_E_UNSUPPORTED, "bn_train_stats: tensors of 2^29 elements or more are not supported");
  SSMI_REQUIRE(ws && ws_bytes >= ssmi_bn_workspace_bytes(npix, c), SSMI_E_WORKSPACE, "bn_train_stats: workspace too small");
  const int c4n = c / 4;
  const int nb = red_blocks(npix, c4n);
  float* part = reinterpret_cast<float*>(ws);
  hipLaunchKernelGGL(bn_stats_partial<T>, dim3(nb), dim3(256), 0, as_stream(stream), x, npix, c4n, ld, part);
  SSMI_LAUNCH_CHECK();
  return launch_bn_stats_final(part, nb, c, npix, eps, momentum, unbiased_moving, moving_mean, moving_var, save_mean,
                               save_invstd, as_stream(stream));
}

extern "C" {

ssmi_status ssmi_bn_train_stats(const float* x, int64_t npix, int32_t c, int32_t ld, float eps, float momentum,
                                int32_t unbiased_moving, float* moving_mean, float* moving_var, float* save_mean,
                                float* save_invstd, void* ws, size_t ws_bytes, void* stream) {
  return bn_train_stats_impl<float>(x, npix, c, ld, eps, momentum, unbiased_moving, moving_mean, moving_var, save_mean,
                                    save_invstd, ws, ws_bytes, stream);
}
ssmi_status ssmi_bn_train_stats_bf16(const uint16_t* x, int64_t npix, int32_t c, int32_t ld, float eps, float momentum,
                                     int32_t unbiased_moving, float* moving_mean, float* moving_var, float* save_mean,
                                     float* save_invstd, void* ws, size_t ws_bytes, void* stream) {
  return bn_train_stats_impl<bf16>(reinterpret_cast<const bf16*>(x), npix, c, ld, eps, momentum, unbiased_moving, moving_mean,
                                   moving_var, save_mean, save_invstd, ws, ws_bytes, stream);
}

ssmi_status ssmi_bn_infer_stats_arena(const float* state, const uint8_t* is_var, int64_t count, float eps, float* out,
                                      void* stream) {
  SSMI_REQUIRE(state && is_var && out && count > 0, SSMI_E_INVALID_ARG, "bn_infer_stats_arena: bad argument");
  int64_t nb = ceil_div64(count, 256);
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(bn_infer_stats_arena_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), state, is_var, count, eps, out);
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

ssmi_status ssmi_bn_infer_stats(const float* moving_mean, const float* moving_var, int32_t c, float eps,
                                float* save_mean, float* save_invstd, void* stream) {
  SSMI_REQUIRE(moving_mean && moving_var && save_mean && save_invstd && c > 0, SSMI_E_INVALID_ARG, "bn_infer_stats: bad argument");
  hipLaunchKernelGGL(bn_infer_stats_kernel, dim3((c + 63) / 64), dim3(64), 0, as_stream(stream), moving_mean, moving_var,
                     c, eps, save_mean, save_invstd);
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

}  // extern "C"

template <typename T>
static ssmi_status bn_act_drop_fwd_impl(const T* x, int32_t ld_x, const float* mean, const float* invstd, const float* gamma,
                                        const float* beta, int32_t act, const float* alpha, float slope, const float* drop,
                                        int64_t hw, T* y, int32_t ld_y, int64_t npix, int32_t c, void* stream) {
  ssmi_status s = check_c(c, "bn_act_fwd");
  if (s) return s;
  SSMI_REQUIRE(x && mean && invstd && gamma && beta && y && npix > 0 && ld_x >= c && ld_x % 4 == 0 && ld_y >= c && ld_y % 4 == 0,
               SSMI_E_INVALID_ARG, "bn_act_fwd: bad argument");
  SSMI_REQUIRE(act != SSMI_ACT_PRELU || alpha, SSMI_E_INVALID_ARG, "bn_act_fwd: PReLU needs alpha");
  SSMI_REQUIRE(!drop || (hw > 0 && npix % hw == 0), SSMI_E_SHAPE, "bn_act_fwd: npix must be a multiple of hw");
  SSMI_REQUIRE(npix * ld_x * 4 < 0x7FFFFFFFLL && npix * ld_y * 4 < 0x7FFFFFFFLL, SSMI_E_UNSUPPORTED,
               "bn_act_fwd: tensors of 2^29 elements or more are not supported");
  const int c4n = c / 4;
  const int nb = ew_blocks_px(npix, c4n);
#define L(A) hipLaunchKernelGGL((bn_act_fwd_kernel<T, A>), dim3(nb), dim3(256), 0, as_stream(stream), x, ld_x, mean, invstd, gamma, beta, alpha, slope, drop, hw, y, ld_y, npix, c4n)
  if (act == SSMI_ACT_NONE) L(SSMI_ACT_NONE); else if (act == SSMI_ACT_PRELU) L(SSMI_ACT_PRELU); else if (act == SSMI_ACT_LEAKY) L(SSMI_ACT_LEAKY);
  else { set_error("bn_act_fwd: unknown activation %d", act); return SSMI_E_INVALID_ARG; }
#undef L
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

namespace ssmi {
template <typename T>
ssmi_status bn_act_drop_bwd_impl(const T* dy, int32_t ld_dy, const T* x, int32_t ld_x, const float* mean, const float* invstd,
                                 const float* gamma, const float* beta, int32_t act, const float* alpha, float slope,
                                 const float* drop, int64_t hw, T* dx, float* dgamma, float* dbeta, float* dalpha,
                                 int64_t npix, int32_t c, void* ws, size_t ws_bytes, void* stream) {
  ssmi_status s = check_c(c, "bn_act_bwd");
  if (s) return s;
  SSMI_REQUIRE(dy && x && mean && invstd && gamma && beta && dx && npix > 0 && ld_x >= c && ld_x % 4 == 0 && ld_dy >= c && ld_dy % 4 == 0,
               SSMI_E_INVALID_ARG, "bn_act_bwd: bad argument");
  SSMI_REQUIRE(!drop || (hw > 0 && npix % hw == 0), SSMI_E_SHAPE, "bn_act_bwd: npix must be a multiple of hw");
  SSMI_REQUIRE(npix * ld_x * 4 < 0x7FFFFFFFLL && npix * ld_dy * 4 < 0x7FFFFFFFLL, SSMI_E_UNSUPPORTED,
               "bn_act_bwd: tensors of 2^29 elements or more are not supported");
  SSMI_REQUIRE(act != SSMI_ACT_PRELU || alpha, SSMI_E_INVALID_ARG, "bn_act_bwd: PReLU needs alpha");
  SSMI_REQUIRE(ws && ws_bytes >= ssmi_bn_workspace_bytes(npix, c), SSMI_E_WORKSPACE, "bn_act_bwd: workspace too small");
  const int c4n = c / 4;
  const int nb = red_blocks(npix, c4n);
  float* part = reinterpret_cast<float*>(ws);
  float* coef = part + (size_t)kMaxRedBlocks * 3 * c;
  hipStream_t st = as_stream(stream);
#define L1(A) hipLaunchKernelGGL((bn_act_bwd_reduce<T, A>), dim3(nb), dim3(256), 0, st, dy, x, ld_x, mean, invstd, gamma, beta, alpha, slope, drop, hw, ld_dy, npix, c4n, part)
  if (act == SSMI_ACT_NONE) L1(SSMI_ACT_NONE); else if (act == SSMI_ACT_PRELU) L1(SSMI_ACT_PRELU); else if (act == SSMI_ACT_LEAKY) L1(SSMI_ACT_LEAKY);
  else { set_error("bn_act_bwd: unknown activation %d", act); return SSMI_E_INVALID_ARG; }
#undef L1
  SSMI_LAUNCH_CHECK();
  return launch_bn_bwd_tail<T>(part, nb, dy, ld_dy, x, ld_x, mean, invstd, gamma, beta, act, alpha, slope, drop, hw, dx, dgamma,
                               dbeta, dalpha, npix, c, coef, st);
}
template ssmi_status bn_act_drop_bwd_impl<float>(const float*, int32_t, const float*, int32_t, const float*, const float*, const float*,
                                                 const float*, int32_t, const float*, float, const float*, int64_t, float*, float*,
                                                 float*, float*, int64_t, int32_t, void*, size_t, void*);
template ssmi_status bn_act_drop_bwd_impl<bf16>(const bf16*, int32_t, const bf16*, int32_t, const float*, const float*, const float*,
                                                const float*, int32_t, const float*, float, const float*, int64_t, bf16*, float*,
                                                float*, float*, int64_t, int32_t, void*, size_t, void*);
}  // namespace ssmi

static ssmi_status fill_merge(const ssmi_merge_desc* d, const void* main, const ssmi_bn_params* bn3, const float* drop,
                              const void* other, const ssmi_bn_params* bn_other, const int32_t* idx, const float* alpha,
                              MergeArgs* a) {
  SSMI_REQUIRE(d && main && bn3 && other && alpha, SSMI_E_INVALID_ARG, "merge: NULL argument");
  SSMI_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0, SSMI_E_SHAPE, "merge: bad shape");
  SSMI_REQUIRE(d->dtype == SSMI_F32 || d->dtype == SSMI_BF16, SSMI_E_UNSUPPORTED, "merge: unknown dtype %d", d->dtype);
  SSMI_REQUIRE((int64_t)d->n * d->h * d->w * d->c * 4 < 0x7FFFFFFFLL, SSMI_E_UNSUPPORTED,
               "merge: tensors of 2^29 elements or more are not supported");
  ssmi_status s = check_c(d->c, "merge");
  if (s) return s;
  SSMI_REQUIRE(d->other_mode >= 0 && d->other_mode <= 3, SSMI_E_INVALID_ARG, "merge: other_mode %d", d->other_mode);
  if (d->other_mode == 1) SSMI_REQUIRE(d->c_other > 0 && d->c_other % 4 == 0 && d->c_other <= d->c, SSMI_E_SHAPE, "merge: c_other %d", d->c_other);
  if (d->other_mode >= 2) {
    SSMI_REQUIRE(d->h % 2 == 0 && d->w % 2 == 0, SSMI_E_SHAPE, "merge: upsampled output must have even height/width");
    SSMI_REQUIRE(bn_other != nullptr, SSMI_E_INVALID_ARG, "merge: bn_other required for mode %d", d->other_mode);
  }
  if (d->other_mode == 3) SSMI_REQUIRE(idx != nullptr, SSMI_E_INVALID_ARG, "merge: idx required for mode 3");
  SSMI_REQUIRE(bn3->mean && bn3->invstd && bn3->gamma && bn3->beta, SSMI_E_INVALID_ARG, "merge: bn3 incomplete");
  a->n = d->n; a->h = d->h; a->w = d->w; a->c = d->c; a->mode = d->other_mode; a->c_other = d->c_other;
  a->main = main; a->mean3 = bn3->mean; a->invstd3 = bn3->invstd; a->gamma3 = bn3->gamma; a->beta3 = bn3->beta;
  a->drop = drop; a->other = other; a->idx = idx; a->alpha = alpha;
  a->mean_o = a->invstd_o = a->gamma_o = a->beta_o = nullptr;
  if (bn_other) { a->mean_o = bn_other->mean; a->invstd_o = bn_other->invstd; a->gamma_o = bn_other->gamma; a->beta_o = bn_other->beta; }
  return SSMI_OK;
}

template <typename T>
static ssmi_status merge_fwd_impl(const ssmi_merge_desc* d, const MergeArgs& a, T* out, void* stream) {
  const int c4n = d->c / 4;
  const int64_t npix = (int64_t)d->n * d->h * d->w;
  const int nb = ew_blocks_px(npix, c4n);
#define L(M) hipLaunchKernelGGL((merge_fwd_kernel<T, M>), dim3(nb), dim3(256), 0, as_stream(stream), a, out, npix, c4n)
  switch (d->other_mode) { case 0: L(0); break; case 1: L(1); break; case 2: L(2); break; default: L(3); break; }
#undef L
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

template <typename T>
static ssmi_status merge_bwd_impl(const ssmi_merge_desc* d, const MergeArgs& a, const ssmi_bn_params* bn3, const float* drop,
                                  const int32_t* idx, const T* dout, T* dmain, T* dsum, T* dother_bn, float* dgamma3,
                                  float* dbeta3, float* dalpha, void* ws, void* stream, float* coef_out = nullptr) {
  // coef_out != nullptr: the coefficients of the apply pass go to the caller's buffer and the pass itself is left to the
  // caller (ssmi_merge_bwd_apply, or a consumer that applies them to its operand)
  hipStream_t st = as_stream(stream);
  const int c = d->c, c4n = c / 4;
  const int64_t npix = (int64_t)d->n * d->h * d->w;
  const int nb = red_blocks(npix, c4n);
  float* part = reinterpret_cast<float*>(ws);
  float* coef = coef_out ? coef_out : part + (size_t)kMaxRedBlocks * 3 * c;
#define L(M) hipLaunchKernelGGL((merge_bwd_reduce<T, M>), dim3(nb), dim3(256), 0, st, a, dout, dsum, npix, c4n, part)
  switch (d->other_mode) { case 0: L(0); break; case 1: L(1); break; case 2: L(2); break; default: L(3); break; }
#undef L
  SSMI_LAUNCH_CHECK();
  ssmi_status s = launch_bn_bwd_final(part, nb, c, npix, bn3->gamma, bn3->invstd, dgamma3, dbeta3, dalpha, coef, st);
  if (s) return s;
  if (!coef_out) {
    hipLaunchKernelGGL(merge_bwd_apply<T>, dim3(ew_blocks_px(npix, c4n)), dim3(256), 0, st, dsum, static_cast<const T*>(a.main), bn3->mean,
                       bn3->invstd, drop, coef, dmain, npix, c4n, (int64_t)d->h * d->w);
    SSMI_LAUNCH_CHECK();
  }
  if (d->other_mode >= 2) {
    const int64_t t4 = (int64_t)d->n * (d->h / 2) * (d->w / 2) * c4n;
    if (d->other_mode == 2)
      hipLaunchKernelGGL((merge_bwd_other<T, 2>), dim3(ew_blocks(t4)), dim3(256), 0, st, dsum, idx, dother_bn, d->n, d->h, d->w, c4n, t4);
    else
      hipLaunchKernelGGL((merge_bwd_other<T, 3>), dim3(ew_blocks(t4)), dim3(256), 0, st, dsum, idx, dother_bn, d->n, d->h, d->w, c4n, t4);
    SSMI_LAUNCH_CHECK();
  }
  return SSMI_OK;
}

extern "C" {

ssmi_status ssmi_bn_act_fwd(const float* x, int32_t ld_x, const float* mean, const float* invstd, const float* gamma,
                            const float* beta, int32_t act, const float* alpha, float slope, float* y, int64_t npix,
                            int32_t c, void* stream) {
  return bn_act_drop_fwd_impl<float>(x, ld_x, mean, invstd, gamma, beta, act, alpha, slope, nullptr, 1, y, c, npix, c, stream);
}
ssmi_status ssmi_bn_act_fwd_bf16(const uint16_t* x, int32_t ld_x, const float* mean, const float* invstd, const float* gamma,
                                 const float* beta, int32_t act, const float* alpha, float slope, uint16_t* y, int64_t npix,
                                 int32_t c, void* stream) {
  return bn_act_drop_fwd_impl<bf16>(reinterpret_cast<const bf16*>(x), ld_x, mean, invstd, gamma, beta, act, alpha, slope, nullptr, 1,
                                    reinterpret_cast<bf16*>(y), c, npix, c, stream);
}

ssmi_status ssmi_bn_act_drop_fwd(const float* x, int32_t ld_x, const float* mean, const float* invstd,
                                 const float* gamma, const float* beta, int32_t act, const float* alpha, float slope,
                                 const float* drop, int64_t hw, float* y, int32_t ld_y, int64_t npix, int32_t c,
                                 void* stream) {
  return bn_act_drop_fwd_impl<float>(x, ld_x, mean, invstd, gamma, beta, act, alpha, slope, drop, hw, y, ld_y, npix, c, stream);
}
ssmi_status ssmi_bn_act_drop_fwd_bf16(const uint16_t* x, int32_t ld_x, const float* mean, const float* invstd,
                                      const float* gamma, const float* beta, int32_t act, const float* alpha, float slope,
                                      const float* drop, int64_t hw, uint16_t* y, int32_t ld_y, int64_t npix, int32_t c,
                                      void* stream) {
  return bn_act_drop_fwd_impl<bf16>(reinterpret_cast<const bf16*>(x), ld_x, mean, invstd, gamma, beta, act, alpha, slope, drop, hw,
                                    reinterpret_cast<bf16*>(y), ld_y, npix, c, stream);
}

ssmi_status ssmi_bn_act_bwd(const float* dy, const float* x, int32_t ld_x, const float* mean, const float* invstd,
                            const float* gamma, const float* beta, int32_t act, const float* alpha, float slope,
                            float* dx, float* dgamma, float* dbeta, float* dalpha, int64_t npix, int32_t c, void* ws,
                            size_t ws_bytes, void* stream) {
  return bn_act_drop_bwd_impl<float>(dy, c, x, ld_x, mean, invstd, gamma, beta, act, alpha, slope, nullptr, 1, dx, dgamma, dbeta,
                                     dalpha, npix, c, ws, ws_bytes, stream);
}
ssmi_status ssmi_bn_act_bwd_bf16(const uint16_t* dy, const uint16_t* x, int32_t ld_x, const float* mean, const float* invstd,
                                 const float* gamma, const float* beta, int32_t act, const float* alpha, float slope,
                                 uint16_t* dx, float* dgamma, float* dbeta, float* dalpha, int64_t npix, int32_t c, void* ws,
                                 size_t ws_bytes, void* stream) {
  return bn_act_drop_bwd_impl<bf16>(reinterpret_cast<const bf16*>(dy), c, reinterpret_cast<const bf16*>(x), ld_x, mean, invstd, gamma,
                                    beta, act, alpha, slope, nullptr, 1, reinterpret_cast<bf16*>(dx), dgamma, dbeta, dalpha, npix, c,
                                    ws, ws_bytes, stream);
}

ssmi_status ssmi_bn_act_drop_bwd(const float* dy, int32_t ld_dy, const float* x, int32_t ld_x, const float* mean,
                                 const float* invstd, const float* gamma, const float* beta, int32_t act,
                                 const float* alpha, float slope, const float* drop, int64_t hw, float* dx,
                                 float* dgamma, float* dbeta, float* dalpha, int64_t npix, int32_t c, void* ws,
                                 size_t ws_bytes, void* stream) {
  return bn_act_drop_bwd_impl<float>(dy, ld_dy, x, ld_x, mean, invstd, gamma, beta, act, alpha, slope, drop, hw, dx, dgamma, dbeta,
                                     dalpha, npix, c, ws, ws_bytes, stream);
}
ssmi_status ssmi_bn_act_drop_bwd_bf16(const uint16_t* dy, int32_t ld_dy, const uint16_t* x, int32_t ld_x, const float* mean,
                                      const float* invstd, const float* gamma, const float* beta, int32_t act,
                                      const float* alpha, float slope, const float* drop, int64_t hw, uint16_t* dx,
                                      float* dgamma, float* dbeta, float* dalpha, int64_t npix, int32_t c, void* ws,
                                      size_t ws_bytes, void* stream) {
  return bn_act_drop_bwd_impl<bf16>(reinterpret_cast<const bf16*>(dy), ld_dy, reinterpret_cast<const bf16*>(x), ld_x, mean, invstd,
                                    gamma, beta, act, alpha, slope, drop, hw, reinterpret_cast<bf16*>(dx), dgamma, dbeta, dalpha,
                                    npix, c, ws, ws_bytes, stream);
}

ssmi_status ssmi_merge_fwd(const ssmi_merge_desc* d, const void* main, const ssmi_bn_params* bn3, const float* drop,
                           const void* other, const ssmi_bn_params* bn_other, const int32_t* idx, const float* alpha,
                           void* out, void* stream) {
  MergeArgs a;
  ssmi_status s = fill_merge(d, main, bn3, drop, other, bn_other, idx, alpha, &a);
  if (s) return s;
  SSMI_REQUIRE(out, SSMI_E_INVALID_ARG, "merge_fwd: out is NULL");
  return dispatch_dtype(d->dtype, [&](auto tag) {
    using T = decltype(tag);
    return merge_fwd_impl<T>(d, a, static_cast<T*>(out), stream);
  });
}

size_t ssmi_merge_workspace_bytes(const ssmi_merge_desc* d) {
  if (!d) return 0;
  return ssmi_bn_workspace_bytes(0, d->c);
}

ssmi_status ssmi_merge_bwd(const ssmi_merge_desc* d, const void* dout, const void* main, const ssmi_bn_params* bn3,
                           const float* drop, const void* other, const ssmi_bn_params* bn_other, const int32_t* idx,
                           const float* alpha, void* dmain, void* dsum, void* dother_bn, float* dgamma3,
                           float* dbeta3, float* dalpha, void* ws, size_t ws_bytes, void* stream) {
  MergeArgs a;
  ssmi_status s = fill_merge(d, main, bn3, drop, other, bn_other, idx, alpha, &a);
  if (s) return s;
  SSMI_REQUIRE(dout && dmain && dsum, SSMI_E_INVALID_ARG, "merge_bwd: NULL gradient tensor");
  SSMI_REQUIRE(d->other_mode < 2 || dother_bn, SSMI_E_INVALID_ARG, "merge_bwd: dother_bn required for mode %d", d->other_mode);
  SSMI_REQUIRE(ws && ws_bytes >= ssmi_merge_workspace_bytes(d), SSMI_E_WORKSPACE, "merge_bwd: workspace too small");
  return dispatch_dtype(d->dtype, [&](auto tag) {
    using T = decltype(tag);
    return merge_bwd_impl<T>(d, a, bn3, drop, idx, static_cast<const T*>(dout), static_cast<T*>(dmain), static_cast<T*>(dsum),
                             static_cast<T*>(dother_bn), dgamma3, dbeta3, dalpha, ws, stream);
  });
}

ssmi_status ssmi_merge_bwd_deferred(const ssmi_merge_desc* d, const void* dout, const void* main, const ssmi_bn_params* bn3,
                                    const float* drop, const void* other, const ssmi_bn_params* bn_other, const int32_t* idx,
                                    const float* alpha, void* dsum, void* dother_bn, float* dgamma3, float* dbeta3,
                                    float* dalpha, float* coef, void* ws, size_t ws_bytes, void* stream) {
  MergeArgs a;
  ssmi_status s = fill_merge(d, main, bn3, drop, other, bn_other, idx, alpha, &a);
  if (s) return s;
  SSMI_REQUIRE(dout && dsum && coef, SSMI_E_INVALID_ARG, "merge_bwd_deferred: NULL argument");
  SSMI_REQUIRE(d->other_mode < 2 || dother_bn, SSMI_E_INVALID_ARG, "merge_bwd_deferred: dother_bn required for mode %d", d->other_mode);
  SSMI_REQUIRE(ws && ws_bytes >= ssmi_merge_workspace_bytes(d), SSMI_E_WORKSPACE, "merge_bwd_deferred: workspace too small");
  return dispatch_dtype(d->dtype, [&](auto tag) {
    using T = decltype(tag);
    return merge_bwd_impl<T>(d, a, bn3, drop, idx, static_cast<const T*>(dout), static_cast<T*>(nullptr), static_cast<T*>(dsum),
                             static_cast<T*>(dother_bn), dgamma3, dbeta3, dalpha, ws, stream, coef);
  });
}

ssmi_status ssmi_merge_bwd_apply(const ssmi_merge_desc* d, const void* dsum, const void* main, const ssmi_bn_params* bn3,
                                 const float* drop, const float* coef, void* dmain, void* stream) {
  SSMI_REQUIRE(d && dsum && main && bn3 && bn3->mean && bn3->invstd && coef && dmain, SSMI_E_INVALID_ARG, "merge_bwd_apply: NULL argument");
  SSMI_REQUIRE(d->c > 0 && d->c % 4 == 0, SSMI_E_SHAPE, "merge_bwd_apply: channel count must be a multiple of 4");
  const int c4n = d->c / 4;
  const int64_t npix = (int64_t)d->n * d->h * d->w;
  return dispatch_dtype(d->dtype, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL(merge_bwd_apply<T>, dim3(ew_blocks_px(npix, c4n)), dim3(256), 0, as_stream(stream), static_cast<const T*>(dsum),
                       static_cast<const T*>(main), bn3->mean, bn3->invstd, drop, coef, static_cast<T*>(dmain), npix, c4n,
                       (int64_t)d->h * d->w);
    SSMI_LAUNCH_CHECK();
    return SSMI_OK;
  });
}

}  // extern "C"

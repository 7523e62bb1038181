// BatchNormalization (+PReLU / LeakyReLU) forward/backward and the fused bottleneck merge.
// All kernels stream NHWC tensors with 16-byte accesses: thread i handles channels [4*(i % C4), +4) of pixel i / C4,
// so consecutive lanes touch consecutive 16-byte chunks.  Per-channel reductions use a fixed (thread, block) ->
// partial-sum mapping followed by a double-precision finalize kernel (deterministic).
#include "common.h"
#include <initializer_list>

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

// Elementwise kernels: thread (cg = tid % cgn, pl = tid / cgn) owns channel group cg - VW consecutive channels - of the
// pixels pl, pl + stride, ..., so its per-channel parameters are loaded ONCE (re-loading five or more 16-byte parameter
// vectors for every 16-byte payload load made the address unit, not HBM, the limit), and it handles four pixels per round
// with all payload loads issued before the first store.  VW = 4 elements per access for float32 storage and 8 for
// bfloat16 storage (16 bytes either way: 8-byte accesses run at 0.55-0.7 of the 16-byte rate and leave half-written
// lines behind); bfloat16 tensors whose channel count or strides are no multiple of eight keep VW = 4.
template <int VW>
struct fvec {
  float e[VW];
  __device__ __forceinline__ float& operator[](int i) { return e[i]; }
  __device__ __forceinline__ const float& operator[](int i) const { return e[i]; }
};
template <int VW>
__device__ __forceinline__ fvec<VW> fsplat(float v) {
  fvec<VW> r;
#pragma unroll
  for (int j = 0; j < VW; ++j) r[j] = v;
  return r;
}
// VW consecutive float32 parameters (16-byte aligned)
template <int VW>
__device__ __forceinline__ fvec<VW> ldp(const float* p) {
  fvec<VW> r;
#pragma unroll
  for (int h = 0; h < VW / 4; ++h) {
    const f32x4 v = ld4(p + 4 * h);
#pragma unroll
    for (int j = 0; j < 4; ++j) r[4 * h + j] = v[j];
  }
  return r;
}
// VW consecutive activation elements: plain pointers (gld / gst) and buffer loads with a float32-unit byte offset (bld)
template <typename T, int VW> struct VAcc;
template <typename T> struct VAcc<T, 4> {
  static __device__ __forceinline__ fvec<4> gld(const T* p) {
    const f32x4 v = Act<T>::gld4(p);
    return fvec<4>{{v[0], v[1], v[2], v[3]}};
  }
  static __device__ __forceinline__ void gst(T* p, const fvec<4>& v) { Act<T>::gst4(p, f32x4{v[0], v[1], v[2], v[3]}); }
  static __device__ __forceinline__ fvec<4> bld(__amdgpu_buffer_rsrc_t r, unsigned off4) {
    const f32x4 v = Act<T>::ld4(r, off4);
    return fvec<4>{{v[0], v[1], v[2], v[3]}};
  }
};
template <> struct VAcc<bf16, 8> {
  static __device__ __forceinline__ fvec<8> widen(u32x4 w) {
    fvec<8> r;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const unsigned q = w[h];
      r[2 * h] = __builtin_bit_cast(float, q << 16);
      r[2 * h + 1] = __builtin_bit_cast(float, q & 0xFFFF0000u);
    }
    return r;
  }
  static __device__ __forceinline__ fvec<8> gld(const bf16* p) { return widen(*reinterpret_cast<const u32x4*>(p)); }
  static __device__ __forceinline__ void gst(bf16* p, const fvec<8>& v) {
    u32x4 w;
#pragma unroll
    for (int h = 0; h < 4; ++h) w[h] = pack_bf16x2(v[2 * h], v[2 * h + 1]);
    *reinterpret_cast<u32x4*>(p) = w;
  }
  static __device__ __forceinline__ fvec<8> bld(__amdgpu_buffer_rsrc_t r, unsigned off4) {
    return widen(__builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off4 >> 1, 0, 0)));
  }
};

// block-level reduction of NV accumulator vectors over the threads that share a channel group: sm holds
// NV * (VW / 4) * 256 float4; the result for channel group cg is returned by thread tid == cg (tid < cgn).
template <int NV, int VW>
__device__ __forceinline__ void block_reduce_v(fvec<VW> (&v)[NV], f32x4* sm, int cgn, int ppi, bool active) {
  const int tid = threadIdx.x;
  constexpr int H = VW / 4;
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int h = 0; h < H; ++h)
      sm[(k * H + h) * 256 + tid] = active ? f32x4{v[k][4 * h], v[k][4 * h + 1], v[k][4 * h + 2], v[k][4 * h + 3]} : f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  if (tid < cgn) {
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
      for (int h = 0; h < H; ++h) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < ppi; ++q) s += sm[(k * H + h) * 256 + q * cgn + tid];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[k][4 * h + j] = s[j];
      }
  }
}
// thread tid < cgn writes its NV reduced vectors as rows blockIdx.x * NV + k of part[rows][c]
template <int NV, int VW>
__device__ __forceinline__ void store_partial_rows(const fvec<VW> (&v)[NV], float* __restrict__ part, int cgn) {
  const int cc = cgn * VW;
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int h = 0; h < VW / 4; ++h)
      st4(part + ((int64_t)blockIdx.x * NV + k) * cc + threadIdx.x * VW + 4 * h,
          f32x4{v[k][4 * h], v[k][4 * h + 1], v[k][4 * h + 2], v[k][4 * h + 3]});
}

template <typename T, int ACT, int VW>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const T* __restrict__ x, int ld, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ alpha,
                                                         float slope, const float* __restrict__ drop, int64_t hw,
                                                         T* __restrict__ y, int y_ld, int64_t npix, int cgn) {
  typedef fvec<VW> V;
  constexpr int NP = VW == 8 ? 2 : 4;      // pixels per round: 16-byte accesses carry twice the elements, half the pixels keep the registers
  const unsigned ppi = 256u / (unsigned)cgn;
  const unsigned cg = threadIdx.x % (unsigned)cgn, pl = threadIdx.x / (unsigned)cgn;
  if (pl >= ppi) return;
  const int c = (int)cg * VW;
  const V m = ldp<VW>(mean + c), is = ldp<VW>(invstd + c), g = ldp<VW>(gamma + c), b = ldp<VW>(beta + c);
  V a = fsplat<VW>(slope);
  if (ACT == SSMI_ACT_PRELU) a = ldp<VW>(alpha + c);
  auto one = [&](const V& xv, unsigned p) {
    V o;
#pragma unroll
    for (int j = 0; j < VW; ++j) o[j] = act_fwd<ACT>(g[j] * ((xv[j] - m[j]) * is[j]) + b[j], a[j]);
    if (drop) {
      const V dv = ldp<VW>(drop + (p / (unsigned)hw) * (unsigned)(cgn * VW) + c);
#pragma unroll
      for (int j = 0; j < VW; ++j) o[j] = o[j] * dv[j];
    }
    return o;
  };
  const unsigned stride = gridDim.x * ppi, n = (unsigned)npix;      // npix * ld < 2^29
  unsigned p = blockIdx.x * ppi + pl;
  for (; (uint64_t)p + (uint64_t)(NP - 1) * stride < (uint64_t)n; p += (unsigned)NP * stride) {
    V xv[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) xv[u] = VAcc<T, VW>::gld(x + (p + (unsigned)u * stride) * (unsigned)ld + c);
#pragma unroll
    for (int u = 0; u < NP; ++u) xv[u] = one(xv[u], p + (unsigned)u * stride);
#pragma unroll
    for (int u = 0; u < NP; ++u) VAcc<T, VW>::gst(y + (p + (unsigned)u * stride) * (unsigned)y_ld + c, xv[u]);
  }
  for (; p < n; p += stride) VAcc<T, VW>::gst(y + p * (unsigned)y_ld + c, one(VAcc<T, VW>::gld(x + p * (unsigned)ld + c), p));
}

// backward pass 1: per-channel sums of dz, dz*xhat and (PReLU) dy*min(z,0)
template <typename T, int ACT, int VW>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce(const T* __restrict__ dy, const T* __restrict__ x, int ld,
                                                         const float* __restrict__ mean, const float* __restrict__ invstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ alpha, float slope,
                                                         const float* __restrict__ drop, int64_t hw, int dy_ld,
                                                         int64_t npix, int cgn, float* __restrict__ part) {
  typedef fvec<VW> V;
  constexpr int NP = VW == 8 ? 2 : 4;      // pixels per round: 16-byte accesses carry twice the elements, half the pixels keep the registers
  __shared__ f32x4 sm[3 * (VW / 4) * 256];
  const int ppi = 256 / cgn;
  const int cg = threadIdx.x % cgn, pl = threadIdx.x / cgn;
  const bool active = pl < ppi;
  const int c = cg * VW;
  V v[3] = {fsplat<VW>(0.f), fsplat<VW>(0.f), fsplat<VW>(0.f)};
  if (active) {
    const V m = ldp<VW>(mean + c), is = ldp<VW>(invstd + c), g = ldp<VW>(gamma + c), b = ldp<VW>(beta + c);
    V a = fsplat<VW>(slope);
    if (ACT == SSMI_ACT_PRELU) a = ldp<VW>(alpha + c);
    // four pixels per round, their eight loads in flight together
    const __amdgpu_buffer_rsrc_t rx = Act<T>::rsrc(x, (unsigned)(npix * ld * 4));
    const __amdgpu_buffer_rsrc_t rd = Act<T>::rsrc(dy, (unsigned)(npix * dy_ld * 4));
    const int64_t stride = (int64_t)gridDim.x * ppi;
    for (int64_t p0 = (int64_t)blockIdx.x * ppi + pl; p0 < npix; p0 += stride * NP) {
      V xv[NP], d[NP];
#pragma unroll
      for (int u = 0; u < NP; ++u) {
        const int64_t p = p0 + u * stride;
        const bool ok = p < npix;
        xv[u] = VAcc<T, VW>::bld(rx, ok ? (unsigned)((p * ld + c) * 4) : kOOB);
        d[u] = VAcc<T, VW>::bld(rd, ok ? (unsigned)((p * dy_ld + c) * 4) : kOOB);   // zero gradient for pixels past the end
      }
      if (drop) {
#pragma unroll
        for (int u = 0; u < NP; ++u) {
          const int64_t p = p0 + u * stride;
          const V dv = ldp<VW>(drop + ((p < npix ? p : npix - 1) / hw) * (cgn * VW) + c);
#pragma unroll
          for (int j = 0; j < VW; ++j) d[u][j] = d[u][j] * dv[j];
        }
      }
#pragma unroll
      for (int u = 0; u < NP; ++u) {
#pragma unroll
        for (int j = 0; j < VW; ++j) {
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
  block_reduce_v<3, VW>(v, sm, cgn, ppi, active);
  if (threadIdx.x < cgn) store_partial_rows<3, VW>(v, part, cgn);
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

template <typename T, int ACT, int VW>
__global__ __launch_bounds__(256) void bn_act_bwd_apply(const T* __restrict__ dy, const T* __restrict__ x, int ld,
                                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ alpha, float slope,
                                                        const float* __restrict__ drop, int64_t hw, int dy_ld,
                                                        const float* __restrict__ coef, T* dx,
                                                        int64_t npix, int cgn) {
  typedef fvec<VW> V;
  constexpr int NP = VW == 8 ? 2 : 4;      // pixels per round: 16-byte accesses carry twice the elements, half the pixels keep the registers
  const int cc = cgn * VW;
  const unsigned ppi = 256u / (unsigned)cgn;
  const unsigned cg = threadIdx.x % (unsigned)cgn, pl = threadIdx.x / (unsigned)cgn;
  if (pl >= ppi) return;
  const int c = (int)cg * VW;
  const V m = ldp<VW>(mean + c), is = ldp<VW>(invstd + c), g = ldp<VW>(gamma + c), b = ldp<VW>(beta + c);
  const V k1 = ldp<VW>(coef + c), m1 = ldp<VW>(coef + cc + c), m2 = ldp<VW>(coef + 2 * cc + c);
  V a = fsplat<VW>(slope);
  if (ACT == SSMI_ACT_PRELU) a = ldp<VW>(alpha + c);
  auto one = [&](const V& xv, V d, unsigned p) {
    if (drop) {
      const V dv = ldp<VW>(drop + (p / (unsigned)hw) * (unsigned)cc + c);
#pragma unroll
      for (int j = 0; j < VW; ++j) d[j] = d[j] * dv[j];
    }
    V o;
#pragma unroll
    for (int j = 0; j < VW; ++j) {
      const float xh = (xv[j] - m[j]) * is[j];
      const float z = g[j] * xh + b[j];
      const float dz = d[j] * act_grad<ACT>(z, a[j]);
      o[j] = k1[j] * (dz - m1[j] - xh * m2[j]);
    }
    return o;
  };
  const unsigned stride = gridDim.x * ppi, n = (unsigned)npix;
  unsigned p = blockIdx.x * ppi + pl;
  for (; (uint64_t)p + (uint64_t)(NP - 1) * stride < (uint64_t)n; p += (unsigned)NP * stride) {
    V xv[NP], d[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const unsigned q = p + (unsigned)u * stride;
      xv[u] = VAcc<T, VW>::gld(x + q * (unsigned)ld + c);
      d[u] = VAcc<T, VW>::gld(dy + q * (unsigned)dy_ld + c);
    }
#pragma unroll
    for (int u = 0; u < NP; ++u) d[u] = one(xv[u], d[u], p + (unsigned)u * stride);
#pragma unroll
    for (int u = 0; u < NP; ++u) VAcc<T, VW>::gst(dx + (p + (unsigned)u * stride) * (unsigned)ld + c, d[u]);   // dx may alias dy: all loads are done
  }
  for (; p < n; p += stride)
    VAcc<T, VW>::gst(dx + p * (unsigned)ld + c,
                     one(VAcc<T, VW>::gld(x + p * (unsigned)ld + c), VAcc<T, VW>::gld(dy + p * (unsigned)dy_ld + c), p));
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
template <int VW>
struct MergeParams {
  fvec<VW> m, is, g, b;         // BN3
  fvec<VW> mo, iso, go, bo;     // BN of the skip branch (modes 2, 3)
};
template <int MODE, int VW>
__device__ __forceinline__ MergeParams<VW> merge_params(const MergeArgs& a, int c) {
  MergeParams<VW> q;
  q.m = ldp<VW>(a.mean3 + c); q.is = ldp<VW>(a.invstd3 + c); q.g = ldp<VW>(a.gamma3 + c); q.b = ldp<VW>(a.beta3 + c);
  q.mo = q.iso = q.go = q.bo = fsplat<VW>(0.f);
  if (MODE >= 2) { q.mo = ldp<VW>(a.mean_o + c); q.iso = ldp<VW>(a.invstd_o + c); q.go = ldp<VW>(a.gamma_o + c); q.bo = ldp<VW>(a.beta_o + c); }
  return q;
}

// Raw operands of element group (p, c..c+VW-1): the main-branch value, the skip-branch value and (mode 3) the argmax indices.
// Loads only, so that a caller can issue several pixels' loads back to back.
template <int VW>
struct MergeRaw {
  fvec<VW> mv, ov, dropv;
  int iv[VW];
  int flat;
  bool has_other;
};
template <typename T, int MODE, int VW>
__device__ __forceinline__ MergeRaw<VW> merge_load(const MergeArgs& a, unsigned p, int c) {
  MergeRaw<VW> r;
  const unsigned hw = (unsigned)(a.h * a.w);     // every tensor is below 2 GiB: 32-bit element arithmetic
  const unsigned n_ = p / hw;
  const T* main_ = static_cast<const T*>(a.main);
  const T* other_ = static_cast<const T*>(a.other);
  r.mv = VAcc<T, VW>::gld(main_ + p * a.c + c);
  r.dropv = fsplat<VW>(1.f);
  if (a.drop) r.dropv = ldp<VW>(a.drop + n_ * (unsigned)a.c + c);
  r.ov = fsplat<VW>(0.f);
#pragma unroll
  for (int j = 0; j < VW; ++j) r.iv[j] = 0;
  r.flat = 0;
  r.has_other = true;
  if (MODE == 0) {
    r.ov = VAcc<T, VW>::gld(other_ + p * a.c + c);
  } else if (MODE == 1) {
    r.has_other = c < a.c_other;
    if (r.has_other) r.ov = VAcc<T, VW>::gld(other_ + p * a.c_other + c);
  } else {
    const unsigned rem = p - n_ * hw;
    const int y = (int)(rem / (unsigned)a.w), x = (int)(rem - (unsigned)y * (unsigned)a.w);
    const int h2 = a.h >> 1, w2 = a.w >> 1;
    const unsigned q = (n_ * (unsigned)h2 + (unsigned)(y >> 1)) * (unsigned)w2 + (unsigned)(x >> 1);
    r.ov = VAcc<T, VW>::gld(other_ + q * a.c + c);
    if (MODE == 3) {
#pragma unroll
      for (int h = 0; h < VW / 4; ++h) {
        const int4 t = *reinterpret_cast<const int4*>(a.idx + q * a.c + c + 4 * h);
        r.iv[4 * h] = t.x; r.iv[4 * h + 1] = t.y; r.iv[4 * h + 2] = t.z; r.iv[4 * h + 3] = t.w;
      }
      r.flat = (y * a.w + x) * a.c + c;
    }
  }
  return r;
}
// pre-activation sum s and the BN3 intermediate xhat3 from the raw operands
template <int MODE, int VW>
__device__ __forceinline__ fvec<VW> merge_sum(const MergeParams<VW>& q, const MergeRaw<VW>& r, fvec<VW>& xh3) {
  fvec<VW> s;
#pragma unroll
  for (int j = 0; j < VW; ++j) {
    xh3[j] = (r.mv[j] - q.m[j]) * q.is[j];
    s[j] = (q.g[j] * xh3[j] + q.b[j]) * r.dropv[j];
  }
  if (MODE <= 1) {
#pragma unroll
    for (int j = 0; j < VW; ++j) s[j] += r.ov[j];            // zeros where the pooled skip tensor has no such channel
  } else {
#pragma unroll
    for (int j = 0; j < VW; ++j) {
      float o = q.go[j] * ((r.ov[j] - q.mo[j]) * q.iso[j]) + q.bo[j];
      if (MODE == 3) o = (r.iv[j] == r.flat + j) ? o : 0.f;
      s[j] += o;
    }
  }
  return s;
}

template <typename T, int MODE, int VW>
__global__ __launch_bounds__(256) void merge_fwd_kernel(MergeArgs a, T* __restrict__ out, int64_t npix, int cgn) {
  typedef fvec<VW> V;
  constexpr int NP = VW == 8 ? 2 : 4;      // pixels per round: 16-byte accesses carry twice the elements, half the pixels keep the registers
  const unsigned ppi = 256u / (unsigned)cgn;
  const unsigned cg = threadIdx.x % (unsigned)cgn, pl = threadIdx.x / (unsigned)cgn;
  if (pl >= ppi) return;
  const int c = (int)cg * VW;
  const MergeParams<VW> q = merge_params<MODE, VW>(a, c);
  const V al = ldp<VW>(a.alpha + c);
  auto finish = [&](const MergeRaw<VW>& r) {
    V xh3;
    const V s = merge_sum<MODE, VW>(q, r, xh3);
    V o;
#pragma unroll
    for (int j = 0; j < VW; ++j) o[j] = s[j] > 0.f ? s[j] : al[j] * s[j];
    return o;
  };
  const unsigned stride = gridDim.x * ppi, n = (unsigned)npix;
  unsigned p = blockIdx.x * ppi + pl;
  for (; (uint64_t)p + (uint64_t)(NP - 1) * stride < (uint64_t)n; p += (unsigned)NP * stride) {
    MergeRaw<VW> r[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) r[u] = merge_load<T, MODE, VW>(a, p + (unsigned)u * stride, c);
    V o[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) o[u] = finish(r[u]);
#pragma unroll
    for (int u = 0; u < NP; ++u) VAcc<T, VW>::gst(out + ((p + (unsigned)u * stride) * (unsigned)cgn + cg) * (unsigned)VW, o[u]);
  }
  for (; p < n; p += stride) VAcc<T, VW>::gst(out + (p * (unsigned)cgn + cg) * (unsigned)VW, finish(merge_load<T, MODE, VW>(a, p, c)));
}

// backward pass 1: dsum = dout * prelu'(s); partial sums of dz = dsum*drop, dz*xhat3 and dout*min(s,0)
template <typename T, int MODE, int VW>
__global__ __launch_bounds__(256) void merge_bwd_reduce(MergeArgs a, const T* __restrict__ dout,
                                                        T* __restrict__ dsum, int64_t npix, int cgn,
                                                        float* __restrict__ part) {
  typedef fvec<VW> V;
  constexpr int NP = VW == 8 ? 2 : 4;      // pixels per round: 16-byte accesses carry twice the elements, half the pixels keep the registers
  __shared__ f32x4 sm[3 * (VW / 4) * 256];
  const int ppi = 256 / cgn;
  const int cg = threadIdx.x % cgn, pl = threadIdx.x / cgn;
  const bool active = pl < ppi;
  const int c = cg * VW;
  V v[3] = {fsplat<VW>(0.f), fsplat<VW>(0.f), fsplat<VW>(0.f)};
  if (active) {
    const V al = ldp<VW>(a.alpha + c);
    const MergeParams<VW> q = merge_params<MODE, VW>(a, c);
    auto accumulate = [&](const MergeRaw<VW>& r, const V& d, V& ds) {
      V xh3;
      const V s = merge_sum<MODE, VW>(q, r, xh3);
#pragma unroll
      for (int j = 0; j < VW; ++j) {
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
    for (; (uint64_t)p + (uint64_t)(NP - 1) * stride < (uint64_t)npix; p += (unsigned)NP * stride) {
      MergeRaw<VW> r[NP];
      V d[NP], ds[NP];
#pragma unroll
      for (int u = 0; u < NP; ++u) {
        const unsigned pu = p + (unsigned)u * stride;
        r[u] = merge_load<T, MODE, VW>(a, pu, c);
        d[u] = VAcc<T, VW>::gld(dout + (pu * (unsigned)cgn + (unsigned)cg) * (unsigned)VW);
      }
#pragma unroll
      for (int u = 0; u < NP; ++u) accumulate(r[u], d[u], ds[u]);
#pragma unroll
      for (int u = 0; u < NP; ++u) VAcc<T, VW>::gst(dsum + ((p + (unsigned)u * stride) * (unsigned)cgn + (unsigned)cg) * (unsigned)VW, ds[u]);
    }
    for (; p < (unsigned)npix; p += stride) {
      V ds;
      const MergeRaw<VW> r = merge_load<T, MODE, VW>(a, p, c);
      const V d = VAcc<T, VW>::gld(dout + (p * (unsigned)cgn + (unsigned)cg) * (unsigned)VW);
      accumulate(r, d, ds);
      VAcc<T, VW>::gst(dsum + (p * (unsigned)cgn + (unsigned)cg) * (unsigned)VW, ds);
    }
  }
  block_reduce_v<3, VW>(v, sm, cgn, ppi, active);
  if (threadIdx.x < cgn) store_partial_rows<3, VW>(v, part, cgn);
}

// backward pass 2: dmain = gamma3*invstd3 * (dsum*drop - mean(dz) - xhat3*mean(dz*xhat3))
template <typename T, int VW>
__global__ __launch_bounds__(256) void merge_bwd_apply(const T* __restrict__ dsum, const T* __restrict__ main,
                                                       const float* __restrict__ mean3, const float* __restrict__ invstd3,
                                                       const float* __restrict__ drop, const float* __restrict__ coef,
                                                       T* __restrict__ dmain, int64_t npix, int cgn, int64_t hw) {
  typedef fvec<VW> V;
  constexpr int NP = VW == 8 ? 2 : 4;      // pixels per round: 16-byte accesses carry twice the elements, half the pixels keep the registers
  const int cc = cgn * VW;
  const unsigned ppi = 256u / (unsigned)cgn;
  const unsigned cg = threadIdx.x % (unsigned)cgn, pl = threadIdx.x / (unsigned)cgn;
  if (pl >= ppi) return;
  const int c = (int)cg * VW;
  const V m = ldp<VW>(mean3 + c), is = ldp<VW>(invstd3 + c);
  const V k1 = ldp<VW>(coef + c), m1 = ldp<VW>(coef + cc + c), m2 = ldp<VW>(coef + 2 * cc + c);
  auto one = [&](const V& ds, const V& mv, unsigned p) {
    V dv = fsplat<VW>(1.f);
    if (drop) dv = ldp<VW>(drop + (p / (unsigned)hw) * (unsigned)cc + c);
    V o;
#pragma unroll
    for (int j = 0; j < VW; ++j) {
      const float xh = (mv[j] - m[j]) * is[j];
      o[j] = k1[j] * (ds[j] * dv[j] - m1[j] - xh * m2[j]);
    }
    return o;
  };
  const unsigned stride = gridDim.x * ppi, n = (unsigned)npix;
  unsigned p = blockIdx.x * ppi + pl;
  for (; (uint64_t)p + (uint64_t)(NP - 1) * stride < (uint64_t)n; p += (unsigned)NP * stride) {
    V ds[NP], mv[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const unsigned e = ((p + (unsigned)u * stride) * (unsigned)cgn + cg) * (unsigned)VW;
      ds[u] = VAcc<T, VW>::gld(dsum + e);
      mv[u] = VAcc<T, VW>::gld(main + e);
    }
#pragma unroll
    for (int u = 0; u < NP; ++u) ds[u] = one(ds[u], mv[u], p + (unsigned)u * stride);
#pragma unroll
    for (int u = 0; u < NP; ++u) VAcc<T, VW>::gst(dmain + ((p + (unsigned)u * stride) * (unsigned)cgn + cg) * (unsigned)VW, ds[u]);
  }
  for (; p < n; p += stride) {
    const unsigned e = (p * (unsigned)cgn + cg) * (unsigned)VW;
    VAcc<T, VW>::gst(dmain + e, one(VAcc<T, VW>::gld(dsum + e), VAcc<T, VW>::gld(main + e), p));
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

ssmi_status launch_bn_bwd_final(const float* part, int rows, int c, int64_t npix, const float* gamma,
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
static inline int ew_blocks_px(int64_t npix, int c4n, int px_per_thread = 4) {
  const int ppi = 256 / c4n;
  int64_t b = ceil_div64(npix, (int64_t)ppi * px_per_thread);
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

// bfloat16 tensors with at least 64 channels take eight elements (16 bytes) per access in the passes that only map
// elements (BN apply forward / backward, merge forward, merge backward apply) where every pixel stride and base address
// allows it; everything else takes four.  Measured alone, ENet 512x512 batch 16, 8 against 4 elements per access: merge
// backward apply 16.3 / 18.1 us, merge forward 15.8 / 16.6, BN passes over 16 / 32 channels 6.4 / 5.9 (worse: fewer
// threads per pixel row), the reducing passes 20.2 / 18.5 (worse); SegNet's BN passes over 64-512 channels gain 1 % of its
// step.  These launches run at the memory system's rate plus a fixed 2-3 us either way; the access width is not their
// limit.  The reducing passes stay at four also because their summation order must equal the float32 kernels'.
template <typename T>
static inline bool wide8(int c, int ld_a, int ld_b, std::initializer_list<const void*> ptrs) {
  static const bool off = getenv("SSMI_NO_BF16_WIDE8") != nullptr;      // measurement switch
  if (sizeof(T) != 2 || off || c < 64 || c % 8 != 0 || ld_a % 8 != 0 || ld_b % 8 != 0 || 256 % (c / 8) != 0) return false;
  for (const void* q : ptrs)
    if (reinterpret_cast<uintptr_t>(q) & 15) return false;
  return true;
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
  const bool w8 = wide8<T>(c, ld_x, ld_dy, {dy, x, dx, mean, invstd, gamma, beta, coef, alpha, drop});
  const int cgn = c / (w8 ? 8 : 4);
  const int nb2 = ew_blocks_px(npix, cgn, w8 ? 2 : 4);
#define L2V(A, VW) hipLaunchKernelGGL((bn_act_bwd_apply<T, A, VW>), dim3(nb2), dim3(256), 0, st, dy, x, ld_x, mean, invstd, gamma, beta, alpha, slope, drop, hw, ld_dy, coef, dx, npix, cgn)
#define L2(A) do { if constexpr (sizeof(T) == 2) { if (w8) { L2V(A, 8); break; } } L2V(A, 4); } while (0)
  if (act == SSMI_ACT_NONE) L2(SSMI_ACT_NONE); else if (act == SSMI_ACT_PRELU) L2(SSMI_ACT_PRELU); else L2(SSMI_ACT_LEAKY);
#undef L2
#undef L2V
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
  const bool w8 = wide8<T>(c, ld_x, ld_y, {x, y, mean, invstd, gamma, beta, alpha, drop});
  const int cgn = c / (w8 ? 8 : 4);
  const int nb = ew_blocks_px(npix, cgn, w8 ? 2 : 4);
#define LV(A, VW) hipLaunchKernelGGL((bn_act_fwd_kernel<T, A, VW>), dim3(nb), dim3(256), 0, as_stream(stream), x, ld_x, mean, invstd, gamma, beta, alpha, slope, drop, hw, y, ld_y, npix, cgn)
#define L(A) do { if constexpr (sizeof(T) == 2) { if (w8) { LV(A, 8); break; } } LV(A, 4); } while (0)
  if (act == SSMI_ACT_NONE) L(SSMI_ACT_NONE); else if (act == SSMI_ACT_PRELU) L(SSMI_ACT_PRELU); else if (act == SSMI_ACT_LEAKY) L(SSMI_ACT_LEAKY);
  else { set_error("bn_act_fwd: unknown activation %d", act); return SSMI_E_INVALID_ARG; }
#undef L
#undef LV
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
  float* part = reinterpret_cast<float*>(ws);
  float* coef = part + (size_t)kMaxRedBlocks * 3 * c;
  const bool w8 = false;      // reducing pass: four elements per access (summation order of the float32 kernel)
  const int cgn = c / (w8 ? 8 : 4);
  const int nb = red_blocks(npix, cgn);
  hipStream_t st = as_stream(stream);
#define L1(A) hipLaunchKernelGGL((bn_act_bwd_reduce<T, A, 4>), dim3(nb), dim3(256), 0, st, dy, x, ld_x, mean, invstd, gamma, beta, alpha, slope, drop, hw, ld_dy, npix, cgn, part)
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

// eight-element accesses for a bfloat16 merge: every tensor and parameter vector it touches must allow them
template <typename T>
static inline bool merge_wide8(const ssmi_merge_desc* d, const MergeArgs& a, std::initializer_list<const void*> more) {
  if (!wide8<T>(d->c, d->other_mode == 1 ? d->c_other : d->c, d->c,
                {a.main, a.other, a.mean3, a.invstd3, a.gamma3, a.beta3, a.drop, a.mean_o, a.invstd_o, a.gamma_o, a.beta_o, a.idx, a.alpha}))
    return false;
  for (const void* q : more)
    if (reinterpret_cast<uintptr_t>(q) & 15) return false;
  return true;
}

template <typename T>
static ssmi_status merge_fwd_impl(const ssmi_merge_desc* d, const MergeArgs& a, T* out, void* stream) {
  const bool w8 = merge_wide8<T>(d, a, {out});
  const int cgn = d->c / (w8 ? 8 : 4);
  const int64_t npix = (int64_t)d->n * d->h * d->w;
  const int nb = ew_blocks_px(npix, cgn, w8 ? 2 : 4);
#define LV(M, VW) hipLaunchKernelGGL((merge_fwd_kernel<T, M, VW>), dim3(nb), dim3(256), 0, as_stream(stream), a, out, npix, cgn)
#define L(M) do { if constexpr (sizeof(T) == 2) { if (w8) { LV(M, 8); break; } } LV(M, 4); } while (0)
  switch (d->other_mode) { case 0: L(0); break; case 1: L(1); break; case 2: L(2); break; default: L(3); break; }
#undef L
#undef LV
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
  float* part = reinterpret_cast<float*>(ws);
  float* coef = coef_out ? coef_out : part + (size_t)kMaxRedBlocks * 3 * c;
  const bool w8 = merge_wide8<T>(d, a, {dsum, dmain, coef});      // for the apply pass; the reducing pass takes four
  const int cgn = c / (w8 ? 8 : 4);
  const int nb = red_blocks(npix, c4n);
#define L(M) hipLaunchKernelGGL((merge_bwd_reduce<T, M, 4>), dim3(nb), dim3(256), 0, st, a, dout, dsum, npix, c4n, part)
  switch (d->other_mode) { case 0: L(0); break; case 1: L(1); break; case 2: L(2); break; default: L(3); break; }
#undef L
  SSMI_LAUNCH_CHECK();
  ssmi_status s = launch_bn_bwd_final(part, nb, c, npix, bn3->gamma, bn3->invstd, dgamma3, dbeta3, dalpha, coef, st);
  if (s) return s;
  if (!coef_out) {
    bool done = false;
    if constexpr (sizeof(T) == 2) {
      if (w8) {
        hipLaunchKernelGGL((merge_bwd_apply<T, 8>), dim3(ew_blocks_px(npix, cgn, 2)), dim3(256), 0, st, dsum, static_cast<const T*>(a.main),
                           bn3->mean, bn3->invstd, drop, coef, dmain, npix, cgn, (int64_t)d->h * d->w);
        done = true;
      }
    }
    if (!done)
      hipLaunchKernelGGL((merge_bwd_apply<T, 4>), dim3(ew_blocks_px(npix, c4n)), dim3(256), 0, st, dsum, static_cast<const T*>(a.main),
                         bn3->mean, bn3->invstd, drop, coef, dmain, npix, c4n, (int64_t)d->h * d->w);
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
    hipLaunchKernelGGL((merge_bwd_apply<T, 4>), dim3(ew_blocks_px(npix, c4n)), dim3(256), 0, as_stream(stream), static_cast<const T*>(dsum),
                       static_cast<const T*>(main), bn3->mean, bn3->invstd, drop, coef, static_cast<T*>(dmain), npix, c4n,
                       (int64_t)d->h * d->w);
    SSMI_LAUNCH_CHECK();
    return SSMI_OK;
  });
}

}  // extern "C"

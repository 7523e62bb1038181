// Shared helpers for libssmi (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <initializer_list>
#include <string>

#include "../../include/ssmi.h"

namespace ssmi {

void set_error(const char* fmt, ...);
// Looks up, for the CURRENT device, the address of every file-scope device array the library uses as scratch (loss.hip).
// Called by ssmi_graph_begin_capture in front of the capture, so that no such lookup happens while a stream is capturing.
bool resolve_static_scratch();
// Records which kernel variant the calling thread's last convolution entry point dispatched (profiling aid).
void set_last_kernel(const char* fmt, ...);

#define SSMI_REQUIRE(cond, code, ...)      \
  do {                                     \
    if (!(cond)) {                         \
      ::ssmi::set_error(__VA_ARGS__);      \
      return (code);                       \
    }                                      \
  } while (0)

#define SSMI_LAUNCH_CHECK()                                                        \
  do {                                                                             \
    hipError_t e__ = hipGetLastError();                                            \
    if (e__ != hipSuccess) {                                                       \
      ::ssmi::set_error("%s:%d kernel launch: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
      return SSMI_E_HIP;                                                           \
    }                                                                              \
  } while (0)

#define SSMI_HIP(call)                                                             \
  do {                                                                             \
    hipError_t e__ = (call);                                                       \
    if (e__ != hipSuccess) {                                                       \
      ::ssmi::set_error("%s:%d %s: %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
      return SSMI_E_HIP;                                                           \
    }                                                                              \
  } while (0)

// Finalises per-workgroup column sums part[rows][2][c] (sum, sum of squares) into BatchNormalization statistics and the
// moving-average update (bn.hip); shared by ssmi_bn_train_stats and the convolution with fused statistics.
ssmi_status launch_bn_stats_final(const float* part, int rows, int c, int64_t npix, float eps, float momentum,
                                  int unbiased_moving, float* moving_mean, float* moving_var, float* save_mean,
                                  float* save_invstd, hipStream_t stream);

// sums partial weight gradients part[nstrips][K_aug * cout] in a fixed order into dw[K * cout] (+ dbias[cout]: row K); conv.hip
void launch_wg_reduce(const float* part, int nstrips, int K, int K_aug, int cout, float* dw, float* dbias, hipStream_t st);
// Finalises per-workgroup sums part[rows][3][c] (dz, dz*xhat, d*min(z,0)) into dgamma / dbeta / dalpha and the three
// coefficient vectors of the apply pass, coef = k1 | m1 | m2 (bn.hip)
ssmi_status launch_bn_bwd_final(const float* part, int rows, int c, int64_t npix, const float* gamma, const float* invstd,
                                float* dgamma, float* dbeta, float* dalpha, float* coef, hipStream_t st);

// Backward of a BatchNormalization(+activation, +spatial dropout) fused into the epilogue of the data-gradient kernel that
// produces the gradient of the BN's output: the kernel adds per-workgroup column sums of dz, dz*xhat and d*min(z,0) to
// part[row][3][c] (x == nullptr: disabled).
struct BnBwdArgs {
  const void* x;           // raw input of the BN (the producing convolution's output, activation type), pixel stride ld
  int ld;
  const float *mean, *invstd, *gamma, *beta, *alpha;
  float slope;
  int act;
  const float* drop;       // [n, c] dropout table or nullptr
  unsigned hw;             // pixels per image
  float* part;
  // forward, inference-phase epilogue only (mean != nullptr): y = act(BN(acc) + skip[pixel * skip_ld + channel]) -
  // the identity-skip merge of a bottleneck folded into its last convolution (skip == nullptr: no addend)
  const void* skip;        // activation type
  int skip_ld;
  // forward kernel run as a data gradient (rotated, transposed kernel matrix): x != nullptr selects the BN-backward sums
  // above instead of the inference epilogue; add: gradient arriving over a skip connection, layout of the output
  const void* add;
  // bfloat16 storage only: the kernel as a bfloat16 pack [tap][cout][cin] (nullptr: float32 kernel words, fp32 MFMA)
  const void* wpack;
  int lds_b;               // with wpack, wide layers: the workgroup stages the kernel rows through LDS (set by the launcher)
  // forward, PRE instantiations only: the convolution's input is act(BN(x)) of the raw tensor x it is given, applied to
  // the operand on its way into the matrix core (a 1x1 convolution behind a training-phase BatchNormalization + PReLU:
  // the normalised tensor is never stored).  pre[0..4] = mean, invstd, gamma, beta, alpha (alpha nullptr: no activation)
  const float* pre[5];
  // forward kernel run as a data gradient, PRE == 2 instantiation only: the operand dy is the gradient a bottleneck's merge
  // backward would have stored, formed on the fly from what its reduction pass left behind (merge_bwd_apply's arithmetic):
  //   dy = k1 * (dsum * drop - m1 - ((main - mean3) * invstd3) * m2)
  // `x` of the launch is dsum; pm_main the raw main-branch tensor (same layout); pm[0..4] = mean3, invstd3, k1, m1, m2 per
  // channel; pm_drop the [n, c] dropout table (nullptr: ones), pm_hw the pixels per image
  const void* pm_main;
  const float* pm[5];
  const float* pm_drop;
  unsigned pm_hw;
};

// LDS-tiled bfloat16 implicit GEMM for wide 3x3 / stride-1 / SAME layers (conv_tiled.hip).  _ok: whether the layer shape
// qualifies; the launcher returns SSMI_E_UNSUPPORTED (nothing launched, no error text) for an epilogue or alignment it does
// not take, and the caller then uses the direct kernel.  part != nullptr: BatchNormalization statistics rows
// part[tile][2][cout]; bn.x / bn.add: data-gradient epilogue with rows bn.part[tile][3][cout]; *part_rows = pixel tiles.
bool conv_tiled_bf16_ok(int h, int w, int cin, int cout, int kh, int kw, int stride, int dil, int pad_t, int pad_l, int ho, int wo,
                        int x_ld, int y_ld);
ssmi_status launch_conv_tiled_bf16(int n, int h, int w, int cin, int cout, int x_ld, int y_ld, const void* x, const void* wpack,
                                   const float* bias, void* y, float* part, int* part_rows, const BnBwdArgs& bn, hipStream_t st);

// Weight gradient of the same layers from staged halo patches (conv_tiled.hip): partial sums part[share][K_aug][cout]
// (k = tap * cin + ci; row K, when with_bias, = bias gradient) for the caller's reduction kernel.
struct WgPatchPlan {
  bool use;
  int K, K_aug, ntiles, tiles_per_share, shares;
  size_t bytes;
};
WgPatchPlan plan_wg_patch_bf16(int n, int h, int w, int cin, int cout, int kh, int kw, int stride, int dil, int pad_t, int pad_l,
                               int ho, int wo, int x_ld, int y_ld, bool with_bias);
ssmi_status launch_wg_patch_bf16(const WgPatchPlan& p, int n, int h, int w, int cin, int cout, int x_ld, int y_ld, const void* x,
                                 const void* dy, float* part, hipStream_t st);

// bfloat16 kernel packs of `count` layers (table on the device): fpack[tap][co][ci], rpack[ntaps-1-tap][ci][co] (conv_tiled.hip)
ssmi_status launch_conv_pack_bf16(const ssmi_rot_entry* table, int count, const float* theta, void* fpack, void* rpack, hipStream_t st);

// Per-channel parameters of a BatchNormalization + PReLU whose output a kernel forms on the fly from the raw tensor
// (PRE instantiations of conv_fwd_kernel / conv_bwd_weight_lds); same arithmetic as bn_act_fwd_kernel.
struct PreBn {
  const float *mean, *invstd, *gamma, *beta, *alpha;      // alpha nullptr: no activation
};
__device__ __forceinline__ float pre_bn_act(float x, float m, float is, float g, float b, float a, bool act) {
  const float z = g * ((x - m) * is) + b;
  return (!act || z > 0.f) ? z : a * z;
}

// Finalises part[rows][3][c] into dgamma / dbeta / dalpha and runs the apply pass (bn.hip).  coef: 3*c floats of scratch.
template <typename T>
ssmi_status launch_bn_bwd_tail(const float* part, int rows, const T* dy, int ld_dy, const T* x, int ld_x,
                               const float* mean, const float* invstd, const float* gamma, const float* beta, int act,
                               const float* alpha, float slope, const float* drop, int64_t hw, T* dx, float* dgamma,
                               float* dbeta, float* dalpha, int64_t npix, int c, float* coef, hipStream_t stream);
// BatchNormalization + activation + spatial dropout backward as two plain passes (bn.hip), any activation type
template <typename T>
ssmi_status bn_act_drop_bwd_impl(const T* dy, int32_t ld_dy, const T* x, int32_t ld_x, const float* mean, const float* invstd,
                                 const float* gamma, const float* beta, int32_t act, const float* alpha, float slope,
                                 const float* drop, int64_t hw, T* dx, float* dgamma, float* dbeta, float* dalpha,
                                 int64_t npix, int32_t c, void* ws, size_t ws_bytes, void* stream);

// Optional per-launch timing (ssmi_profile_begin / _end): while enabled, a ProfileScope brackets ONE kernel launch with
// HIP events on the launch stream and records the kernel name (set_last_kernel) with its algorithmic bytes and flops.
struct ProfileScope {
  ProfileScope(hipStream_t st, double bytes, double flops);
  ~ProfileScope();
  hipStream_t st_;
  hipEvent_t e0_ = nullptr, e1_ = nullptr;
  double bytes_, flops_;
};

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// All operand loads are raw buffer loads: the descriptor carries the tensor's byte size, and a lane that must read a
// zero (padding, tile tail, masked channel) simply passes an out-of-range offset - the hardware bounds check returns 0.
// This keeps every load unconditional, so they issue back to back and overlap (a `valid ? load : 0` select makes hipcc
// branch around each load and drain the memory counter after it).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
static constexpr unsigned kOOB = 0x80000000u;   // >= any descriptor size (tensors are limited to 2 GiB - 1)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}
__device__ __forceinline__ f32x4 bload4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}

__device__ __forceinline__ void bstore(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, byte_off, 0, 0);   // dropped when out of range
}

// ------------------------------------------------------------------------------------------------
// Activation storage types.  Kernels are templated on the element type T of their activation tensors (float or bf16);
// parameters, statistics, partial sums and all arithmetic stay float32.  Offsets keep the unit the float32 code uses
// everywhere - BYTES OF A FLOAT32 TENSOR ("off4" = element index * 4) - and the bf16 accessors halve them, so one
// address computation serves both types.  The out-of-range marker survives the halving: kOOB >> 1 = 2^30 is beyond any
// bf16 tensor the size checks admit (element counts below 2^29).
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf16_bits_to_f32(unsigned hi16_in_low_half) {
  return __builtin_bit_cast(float, hi16_in_low_half << 16);
}
__device__ __forceinline__ f32x4 unpack_bf16x4(u32x2 w) {
  f32x4 v;
  v[0] = __builtin_bit_cast(float, w[0] << 16);
  v[1] = __builtin_bit_cast(float, w[0] & 0xFFFF0000u);
  v[2] = __builtin_bit_cast(float, w[1] << 16);
  v[3] = __builtin_bit_cast(float, w[1] & 0xFFFF0000u);
  return v;
}
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {   // round to nearest even (v_cvt_pk_bf16_f32)
  const bf16 x = (bf16)a, y = (bf16)b;
  return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}
__device__ __forceinline__ u32x2 pack_bf16x4(f32x4 v) {
  u32x2 w;
  w[0] = pack_bf16x2(v[0], v[1]);
  w[1] = pack_bf16x2(v[2], v[3]);
  return w;
}

template <typename T> struct Act;
template <> struct Act<float> {
  static constexpr int kDtype = 0;
  static __device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, unsigned bytes4) { return make_rsrc(p, bytes4); }
  static __device__ __forceinline__ float ld(__amdgpu_buffer_rsrc_t r, unsigned off4) { return bload(r, off4); }
  static __device__ __forceinline__ f32x4 ld4(__amdgpu_buffer_rsrc_t r, unsigned off4) { return bload4(r, off4); }
  // raw form of a 4-element load: what sits in registers while the load is in flight; up4() makes it float32 at the use.
  // (Converting right behind the load makes the wave wait for the data there, which defeats every prefetch.)
  typedef f32x4 raw4;
  static __device__ __forceinline__ raw4 ld4r(__amdgpu_buffer_rsrc_t r, unsigned off4) { return bload4(r, off4); }
  static __device__ __forceinline__ f32x4 up4(raw4 v) { return v; }
  static __device__ __forceinline__ void st(__amdgpu_buffer_rsrc_t r, unsigned off4, float v) { bstore(r, off4, v); }
  // plain pointers (element index arithmetic done by the caller on T*)
  static __device__ __forceinline__ float gld(const float* p) { return *p; }
  static __device__ __forceinline__ f32x4 gld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
  static __device__ __forceinline__ void gst(float* p, float v) { *p = v; }
  static __device__ __forceinline__ void gst4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <> struct Act<bf16> {
  static constexpr int kDtype = 1;
  static __device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, unsigned bytes4) { return make_rsrc(p, bytes4 >> 1); }
  static __device__ __forceinline__ float ld(__amdgpu_buffer_rsrc_t r, unsigned off4) {
    return bf16_bits_to_f32((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, off4 >> 1, 0, 0));
  }
  static __device__ __forceinline__ f32x4 ld4(__amdgpu_buffer_rsrc_t r, unsigned off4) {
    return unpack_bf16x4(__builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, off4 >> 1, 0, 0)));
  }
  typedef u32x2 raw4;
  static __device__ __forceinline__ raw4 ld4r(__amdgpu_buffer_rsrc_t r, unsigned off4) {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, off4 >> 1, 0, 0));
  }
  static __device__ __forceinline__ f32x4 up4(raw4 v) { return unpack_bf16x4(v); }
  static __device__ __forceinline__ void st(__amdgpu_buffer_rsrc_t r, unsigned off4, float v) {
    const bf16 h = (bf16)v;
    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, h), r, off4 >> 1, 0, 0);
  }
  static __device__ __forceinline__ float gld(const bf16* p) { return (float)*p; }
  static __device__ __forceinline__ f32x4 gld4(const bf16* p) { return unpack_bf16x4(*reinterpret_cast<const u32x2*>(p)); }
  static __device__ __forceinline__ void gst(bf16* p, float v) { *p = (bf16)v; }
  static __device__ __forceinline__ void gst4(bf16* p, f32x4 v) { *reinterpret_cast<u32x2*>(p) = pack_bf16x4(v); }
};

// Calls `body` with a value of the activation type selected by an ssmi_dtype code: dispatch_dtype(dt, [&](auto tag) {
// using T = decltype(tag); ... }).  Unknown codes return SSMI_E_UNSUPPORTED.
template <typename F>
static inline ssmi_status dispatch_dtype(int dtype, F&& body) {
  if (dtype == SSMI_F32) return body(float{});
  if (dtype == SSMI_BF16) return body(bf16{});
  set_error("unknown activation dtype %d", dtype);
  return SSMI_E_UNSUPPORTED;
}

// Wave-wide (64 lanes) sum.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace ssmi

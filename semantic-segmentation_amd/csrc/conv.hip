// Convolutions as implicit GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), operands read straight
// from NHWC global memory.  One wave owns a 32-row x (NT*32)-column accumulator tile:
//   forward / data-gradient : rows = 32 consecutive pixels, columns = output channels, K = taps x channels
//   weight-gradient         : rows = 32 consecutive (tap,cin) kernel rows, columns = cout, K = pixels
// Kernel families: conv_fwd_kernel / conv_bwd_data_kernel (operands straight from global memory, register double
// buffered), conv_gemm_lds (kernel matrix in LDS, 16x16x4 tiles for narrow outputs), conv_bwd_weight_lds / _kernel
// (strip kernels for narrow layers), conv_bwd_weight_tiled (tiled GEMM for wide layers).  Epilogues optionally emit
// the column sums BatchNormalization needs (forward statistics / backward reductions).
// MFMA operand map (32x32x2, wave64): lane l supplies A[row = l&31][k = l>>5] and B[k = l>>5][col = l&31];
// accumulator register i of lane l is C[row = (i&3) + 8*(i>>2) + 4*(l>>5)][col = l&31].
#include "common.h"
#include <algorithm>
#include <vector>

#include <algorithm>
#include <type_traits>

namespace ssmi {

struct Geo {
  int n, h, w, cin, ho, wo, cout, kh, kw, stride, dil, pad_t, pad_l, x_ld, y_ld;
  const void *wfp, *wrp;      // bfloat16 kernel packs (forward / data-gradient form) or nullptr
};

static Geo to_geo(const ssmi_conv_desc* d) {
  Geo g;
  g.n = d->n; g.h = d->h; g.w = d->w; g.cin = d->cin; g.ho = d->ho; g.wo = d->wo; g.cout = d->cout;
  g.kh = d->kh; g.kw = d->kw; g.stride = d->stride; g.dil = d->dil; g.pad_t = d->pad_t; g.pad_l = d->pad_l;
  g.x_ld = d->x_ld; g.y_ld = d->y_ld;
  g.wfp = d->dtype == SSMI_BF16 ? d->w_fwd_pack : nullptr;
  g.wrp = d->dtype == SSMI_BF16 ? d->w_dgrad_pack : nullptr;
  return g;
}

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

struct Sizes {
  unsigned x_bytes, y_bytes, w_bytes;   // byte sizes of the x-like, y-like and kernel tensors
  int xcd;                              // 1: workgroups take tiles through xcd_tile() (below)
};

// Workgroups are dealt to the eight XCDs round-robin - b and b + 8 share an XCD, and with it an L2 -, so with tile = b the
// eight neighbours of a pixel tile sit behind eight different L2s and every halo row of a multi-tap layer is fetched from
// HBM once per XCD that touches it.  Here the workgroups of one XCD take a CONTIGUOUS range of the n tiles (a bijection
// for any n), so halo rows are shared in that XCD's L2.  Speed only: nothing depends on where a workgroup runs.
__device__ __forceinline__ unsigned xcd_tile(unsigned b, unsigned n, int on) {
  if (!on) return b;
  const unsigned q = n >> 3, r = n & 7u, x = b & 7u;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

// ---------------------------------------------------------------------------------------------------
// forward: y[p, co] = bias[co] + sum_{tap, ci} x[pix(p, tap), ci] * w[tap, ci, co]
// VEC: cin % 8 == 0 and 16-byte aligned rows -> each lane loads 4 consecutive channels at once; the k order inside
// a block of 8 channels is permuted identically for A and B (lane half hh takes channels 4*hh .. 4*hh+3).
// ---------------------------------------------------------------------------------------------------
// The second launch bound (workgroups per CU = waves per SIMD) is the occupancy each variant reaches without spilling.
// Stating it makes the compiler keep the MFMA accumulators in the unified VGPR budget instead of a separate AGPR block
// (ENet step 10.19 -> 10.12 ms, 1024x1024 evaluation 5.06 -> 4.93 ms, SegNet step 59.0 -> 58.3 ms when the bounds went
// in).  A bound that is too tight spills silently: NT = 1 stood at 5 until the bf16 and data-gradient epilogues had
// grown the kernel past 96 registers and 9-11 of them lived in scratch (with 4: ENet bf16 step 7.90 -> 7.68 ms) -
// check `.vgpr_spill_count` / `.private_segment_fixed_size` of the listing after every change to this kernel.  The strip
// and tiled weight-gradient kernels measured slower with such bounds (spills, or no gain) and keep the default.
// PRE (float32 tensors, VEC, 1x1 stride 1, cin <= 64, statistics epilogue): the operand is act(BN(x)), formed from the raw
// x when a step is multiplied (bn.pre; parameters of the <= 64 input channels through LDS).  Its own instantiation, so
// that the plain kernels keep their register counts.
// PRE == 2 (float32, VEC, NT == 1, 1x1 stride 1, <= 128 input channels, BN-backward sums epilogue): the operand is the
// gradient a bottleneck's merge backward would have stored, formed from dsum and the raw main-branch tensor (bn.pm*).
template <typename T, int NT, bool VEC, int PRE = 0>
__global__ __launch_bounds__(256, ((PRE == 1 && NT == 2) || PRE == 2 ? 3 : (NT == 1 ? 4 : (NT == 2 ? 4 : 2)))) void conv_fwd_kernel(Geo g, Sizes sz, const T* __restrict__ x,
                                                       const float* __restrict__ w, const float* __restrict__ bias,
                                                       T* __restrict__ y, float* __restrict__ part, BnBwdArgs bn) {
  __shared__ float stat_sm[4 * 3 * NT * 32];
  constexpr int kPreC = PRE == 2 ? 128 : 64;      // channels the parameter block in LDS covers
  __shared__ __attribute__((aligned(16))) float pre_sm[PRE ? 5 * kPreC : 4];
  if constexpr (PRE != 0) {      // every wave reaches this barrier (these launches always carry a statistics epilogue)
    for (int e = threadIdx.x; e < 5 * g.cin; e += 256) {
      const int k = e / g.cin, c = e - k * g.cin;
      const float* src = PRE == 2 ? bn.pm[k] : bn.pre[k];
      pre_sm[k * kPreC + c] = src ? src[c] : 0.f;
    }
    __syncthreads();
  }
  // bf16 wide layers: the kernel rows of the workgroup's 128 output channels go through LDS (below)
  constexpr bool kLdsB = sizeof(T) == 2 && VEC && NT >= 2;
  constexpr int kBRow = 80;                    // bytes per staged kernel row: 32 channels + 16 bytes of padding
  __shared__ __attribute__((aligned(16))) unsigned char bsm[kLdsB ? NT * 32 * kBRow : 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int64_t M = (int64_t)g.n * g.ho * g.wo;
  const int64_t tile0 = ((int64_t)xcd_tile(blockIdx.x, gridDim.x, sz.xcd & 1) * 4 + wave) * 32;
  bool lds_b = false;
  if constexpr (kLdsB) lds_b = bn.wpack != nullptr && bn.lds_b != 0;      // uniform
  // with statistics, or kernel rows staged by the whole workgroup, every wave reaches the barriers (its rows are masked)
  if (tile0 >= M && part == nullptr && bn.x == nullptr && !lds_b) return;
  const __amdgpu_buffer_rsrc_t rx = Act<T>::rsrc(x, sz.x_bytes), rw = make_rsrc(w, sz.w_bytes);
  const int cbase = blockIdx.y * 32 * NT;
  const unsigned p = (unsigned)tile0 + (unsigned)r;       // M < 2^31: 32-bit pixel arithmetic
  const bool pv = p < (unsigned)M;
  const unsigned pp = pv ? p : 0u;
  const unsigned t1 = pp / (unsigned)g.wo;
  const int ox = (int)(pp - t1 * (unsigned)g.wo);
  const unsigned n_ = t1 / (unsigned)g.ho;
  const int oy = (int)(t1 - n_ * (unsigned)g.ho);
  const unsigned img_off = n_ * (unsigned)(g.h * g.w) * (unsigned)g.x_ld * 4u;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  bool colv[NT];
  unsigned wcol[NT];   // byte offset of the lane's column inside a kernel row, or kOOB
  unsigned wcolr[NT], wcolm[NT];   // the same split into raw offset and out-of-range mask (bit 31)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int c = cbase + t * 32 + r;
    colv[t] = c < g.cout;
    wcol[t] = colv[t] ? (unsigned)c * 4u : kOOB;
    wcolr[t] = colv[t] ? (unsigned)c * 4u : 0u;
    wcolm[t] = colv[t] ? 0u : kOOB;
  }
  const unsigned wrow = (unsigned)g.cout * 4u;   // bytes per (tap, ci) kernel row

  bool packed = false;
  if constexpr (sizeof(T) == 2 && VEC) packed = bn.wpack != nullptr;      // uniform
  if (lds_b) {
    if constexpr (kLdsB) {
      // Wide layers on the bf16 matrix instructions.  Fed straight from global memory every wave fetches its own copy of
      // the workgroup's NT*32 kernel rows of a step - NT 16-byte loads per lane for NT matrix instructions - and the
      // launch runs at the rate of the vector-memory path, not of the matrix cores.  Here the workgroup stages the rows of a
      // 32-channel chunk once ([NT*32 couts][32 channels] bf16, one or two 16-byte loads per thread, one chunk ahead in
      // registers) and the four waves read their B operands from LDS; A (the wave's own 32 pixels) still comes from
      // global memory.
      typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      const int ntaps = g.kh * g.kw;
      const __amdgpu_buffer_rsrc_t rp = make_rsrc(bn.wpack, (unsigned)(ntaps * g.cin * g.cout) * 2u);
      const int srow = threadIdx.x >> 2, sseg = threadIdx.x & 3;       // staging role: rows srow and srow + 64, segment sseg
      constexpr int NB = NT / 2;                                        // staged 16-byte pieces per thread
      unsigned gb[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int c = cbase + srow + 64 * i;
        gb[i] = c < g.cout ? (unsigned)(c * g.cin + sseg * 8) * 2u : kOOB;
      }
      const unsigned tapbytes = (unsigned)(g.cin * g.cout) * 2u;
      int tap = 0, ky = 0, kx = 0, c0 = 0;            // state of the NEXT fetch
      unsigned xo = 0, wtap = 0;
      auto settap = [&]() {
        const int iy = oy * g.stride + ky * g.dil - g.pad_t, ix = ox * g.stride + kx * g.dil - g.pad_l;
        const bool inb = pv && tap < ntaps && (unsigned)iy < (unsigned)g.h && (unsigned)ix < (unsigned)g.w;
        xo = inb ? (n_ * (unsigned)(g.h * g.w) + (unsigned)(iy * g.w + ix)) * (unsigned)g.x_ld * 2u : (kOOB >> 1);
        wtap = tap < ntaps ? (unsigned)tap * tapbytes : kOOB;
      };
      settap();
      u32x4 pa[2], pb[NB];
      auto fetch = [&]() {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const unsigned cb = (unsigned)(c0 + 16 * k + 8 * hh) * 2u;
          pa[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (xo + cb) | (xo & (kOOB >> 1)), 0, 0));
        }
#pragma unroll
        for (int i = 0; i < NB; ++i)
          pb[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rp, (wtap + (unsigned)c0 * 2u + gb[i]) | (wtap & kOOB) | (gb[i] & kOOB), 0, 0));
        c0 += 32;
        if (c0 >= g.cin) {
          c0 = 0;
          ++tap;
          if (++kx == g.kw) { kx = 0; ++ky; }
          settap();
        }
      };
      const int nchunks = ntaps * (g.cin >> 5);
      fetch();
      for (int i = 0; i < nchunks; ++i) {
        __syncthreads();            // everyone is done reading the previous chunk's rows
#pragma unroll
        for (int j = 0; j < NB; ++j) *reinterpret_cast<u32x4*>(bsm + (srow + 64 * j) * kBRow + sseg * 16) = pb[j];
        const u32x4 ca0 = pa[0], ca1 = pa[1];
        __syncthreads();
        fetch();                    // the next chunk, in flight while this one is multiplied; past the end: zeros
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const u32x4 b0 = *reinterpret_cast<const u32x4*>(bsm + (t * 32 + r) * kBRow + hh * 16);
          const u32x4 b1 = *reinterpret_cast<const u32x4*>(bsm + (t * 32 + r) * kBRow + 32 + hh * 16);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ca0), __builtin_bit_cast(bf16x8, b0), acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ca1), __builtin_bit_cast(bf16x8, b1), acc[t], 0, 0, 0);
        }
      }
    }
  } else if (packed) {
    if constexpr (sizeof(T) == 2 && VEC) {
      // bfloat16 matrix instructions: one v_mfma_f32_32x32x16_bf16 per 16 input channels and column tile.  Lane (r, hh)
      // feeds A = 8 consecutive channels of its pixel (one 16-byte load, no unpacking) and B = the same 8 channels of
      // column r's kernel row from the pack [tap][cout][cin] (one 16-byte load).  Two steps of operands in flight.
      typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      const int ntaps = g.kh * g.kw;
      const __amdgpu_buffer_rsrc_t rp = make_rsrc(bn.wpack, (unsigned)(ntaps * g.cin * g.cout) * 2u);
      unsigned bcol[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) bcol[t] = colv[t] ? (unsigned)((cbase + t * 32 + r) * g.cin + 8 * hh) * 2u : kOOB;
      const unsigned tapbytes = (unsigned)(g.cin * g.cout) * 2u;
      int tap = 0, ky = 0, kx = 0, c0 = 0;            // state of the NEXT fetch
      unsigned xo = 0, wtap = 0;
      auto settap = [&]() {
        const int iy = oy * g.stride + ky * g.dil - g.pad_t, ix = ox * g.stride + kx * g.dil - g.pad_l;
        const bool inb = pv && tap < ntaps && (unsigned)iy < (unsigned)g.h && (unsigned)ix < (unsigned)g.w;
        // byte offset of the pixel in the bfloat16 tensor; the out-of-range marker is 2^30 (the descriptor is half as large)
        xo = inb ? (n_ * (unsigned)(g.h * g.w) + (unsigned)(iy * g.w + ix)) * (unsigned)g.x_ld * 2u : (kOOB >> 1);
        wtap = tap < ntaps ? (unsigned)tap * tapbytes : kOOB;
      };
      settap();
      const bool pre_act = PRE == 1 && bn.pre[4] != nullptr;
      auto fetch = [&](u32x4& a, u32x4 (&b)[NT], int* cq = nullptr) {
        const unsigned cb = (unsigned)(c0 + 8 * hh) * 2u;
        if constexpr (PRE == 1) *cq = c0 < g.cin && tap == 0 ? c0 + 8 * hh : 0;      // (steps past the end multiply zeros)
        a = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (xo + cb) | (xo & (kOOB >> 1)), 0, 0));
#pragma unroll
        for (int t = 0; t < NT; ++t)
          b[t] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rp, (wtap + (unsigned)c0 * 2u + bcol[t]) | (wtap & kOOB) | (bcol[t] & kOOB), 0, 0));
        c0 += 16;
        if (c0 >= g.cin) {
          c0 = 0;
          ++tap;
          if (++kx == g.kw) { kx = 0; ++ky; }
          settap();
        }
      };
      auto mma = [&](const u32x4& a, const u32x4 (&b)[NT], int cq = 0) {
        u32x4 av = a;
        if constexpr (PRE == 1) {
          // the lane's eight channels: widened, normalised + activated as bn_act_fwd_kernel does, rounded to bfloat16 as its
          // store would have rounded them
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const float* ps = pre_sm + cq + 4 * h2;
            const f32x4 m = *reinterpret_cast<const f32x4*>(ps), is = *reinterpret_cast<const f32x4*>(ps + 64);
            const f32x4 ga = *reinterpret_cast<const f32x4*>(ps + 128), be = *reinterpret_cast<const f32x4*>(ps + 192);
            const f32x4 al = *reinterpret_cast<const f32x4*>(ps + 256);
            u32x2 raw;
            raw[0] = a[2 * h2]; raw[1] = a[2 * h2 + 1];
            f32x4 v = unpack_bf16x4(raw);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = pre_bn_act(v[j], m[j], is[j], ga[j], be[j], al[j], pre_act);
            const u32x2 pk = pack_bf16x4(v);
            av[2 * h2] = pk[0]; av[2 * h2 + 1] = pk[1];
          }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, b[t]), acc[t], 0, 0, 0);
      };
      // A step is ONE matrix instruction per column tile (32 cycles), far shorter than the memory round trip that feeds
      // it: the loop is as fast as its loads in flight allow, so the operands of PD steps are fetched ahead.
      const int nsteps = ntaps * (g.cin >> 4);
      constexpr int PD = NT == 1 ? 6 : (NT == 2 ? 4 : 3);
      u32x4 pa[PD], pb[PD][NT];
      int pq[PD];
#pragma unroll
      for (int d = 0; d < PD; ++d) fetch(pa[d], pb[d], &pq[d]);
      for (int i = 0; i < nsteps; i += PD) {
#pragma unroll
        for (int d = 0; d < PD; ++d) {
          if (i + d < nsteps) mma(pa[d], pb[d], pq[d]);
          fetch(pa[d], pb[d], &pq[d]);       // step i + d + PD; past the end: zeros (tap >= ntaps)
        }
      }
    }
  } else if (VEC) {
    // Flattened (tap, 8-channel block) reduction, software pipelined: the operands of step i+1 are in flight while
    // step i is multiplied (one A load of 16 bytes + 4*NT kernel words per step, double buffered in registers).
    const int ntaps = g.kh * g.kw;
    int tap = 0, ky = 0, kx = 0, c0 = 0;            // state of the NEXT fetch
    unsigned xo = 0, wtap = 0;
    auto settap = [&]() {
      const int iy = oy * g.stride + ky * g.dil - g.pad_t, ix = ox * g.stride + kx * g.dil - g.pad_l;
      const bool inb = pv && tap < ntaps && (unsigned)iy < (unsigned)g.h && (unsigned)ix < (unsigned)g.w;
      xo = inb ? img_off + (unsigned)(iy * g.w + ix) * (unsigned)g.x_ld * 4u : kOOB;
      wtap = tap < ntaps ? (unsigned)(tap * g.cin) * wrow : kOOB;
    };
    settap();
    const bool pre_act = PRE == 1 && bn.pre[4] != nullptr;
    const __amdgpu_buffer_rsrc_t rmain = Act<T>::rsrc(PRE == 2 ? bn.pm_main : (const void*)x, PRE == 2 ? sz.x_bytes : 0u);
    const unsigned drop_off = PRE == 2 && bn.pm_drop ? n_ * (unsigned)g.cin : 0u;      // the lane's image in the dropout table
    auto fetch = [&](f32x4& a, float (&b)[4][NT], int* cq = nullptr, f32x4* a2 = nullptr, f32x4* dv = nullptr) {
      // valid offsets are below 2^31, so OR-ing bit 31 in (kOOB) is the out-of-range mask: no compare/select per load
      const unsigned cib = (unsigned)(c0 + 4 * hh);
      if constexpr (PRE != 0) *cq = c0 < g.cin && tap == 0 ? (int)cib : 0;      // (steps past the end multiply zeros)
      a = Act<T>::ld4(rx, (xo + cib * 4u) | (xo & kOOB));
      if constexpr (PRE == 2) {
        *a2 = Act<T>::ld4(rmain, (xo + cib * 4u) | (xo & kOOB));
        *dv = bn.pm_drop ? *reinterpret_cast<const f32x4*>(bn.pm_drop + drop_off + (unsigned)*cq) : f32x4{1.f, 1.f, 1.f, 1.f};
      }
      const unsigned wb = wtap + cib * wrow, wm = wtap & kOOB;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < NT; ++t) b[j][t] = bload(rw, (wb + (unsigned)j * wrow + wcolr[t]) | wm | wcolm[t]);
      c0 += 8;
      if (c0 >= g.cin) {
        c0 = 0;
        ++tap;
        if (++kx == g.kw) { kx = 0; ++ky; }
        settap();
      }
    };
    auto mma = [&](const f32x4& a, const float (&b)[4][NT], int cq = 0, const f32x4* a2 = nullptr, const f32x4* dv = nullptr) {
      f32x4 av = a;
      if constexpr (PRE == 2) {      // merge_bwd_apply's arithmetic
        const f32x4 m = *reinterpret_cast<const f32x4*>(pre_sm + cq), is = *reinterpret_cast<const f32x4*>(pre_sm + kPreC + cq);
        const f32x4 k1 = *reinterpret_cast<const f32x4*>(pre_sm + 2 * kPreC + cq), m1 = *reinterpret_cast<const f32x4*>(pre_sm + 3 * kPreC + cq);
        const f32x4 m2 = *reinterpret_cast<const f32x4*>(pre_sm + 4 * kPreC + cq);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xh = ((*a2)[j] - m[j]) * is[j];
          av[j] = k1[j] * (a[j] * (*dv)[j] - m1[j] - xh * m2[j]);
        }
      }
      if constexpr (PRE == 1) {
        const f32x4 m = *reinterpret_cast<const f32x4*>(pre_sm + cq), is = *reinterpret_cast<const f32x4*>(pre_sm + 64 + cq);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(pre_sm + 128 + cq), be = *reinterpret_cast<const f32x4*>(pre_sm + 192 + cq);
        const f32x4 al = *reinterpret_cast<const f32x4*>(pre_sm + 256 + cq);
#pragma unroll
        for (int j = 0; j < 4; ++j) av[j] = pre_bn_act(a[j], m[j], is[j], ga[j], be[j], al[j], pre_act);
        if constexpr (sizeof(T) == 2) av = unpack_bf16x4(pack_bf16x4(av));      // as the stored tensor would have been rounded
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = MFMA32(av[j], b[j][t], acc[t]);
    };
    const int nsteps = ntaps * (g.cin >> 3);
    if constexpr (PRE == 2) {
      f32x4 a0, a1, c0_, c1_, d0, d1;
      float b0[4][NT], b1[4][NT];
      int q0, q1;
      fetch(a0, b0, &q0, &c0_, &d0);
      for (int i = 0; i < nsteps; i += 2) {
        fetch(a1, b1, &q1, &c1_, &d1);               // past the end: zeros (tap >= ntaps)
        mma(a0, b0, q0, &c0_, &d0);
        fetch(a0, b0, &q0, &c0_, &d0);
        if (i + 1 < nsteps) mma(a1, b1, q1, &c1_, &d1);
      }
    } else if constexpr (PRE == 1) {
      // (1x1 over at most 64 channels: at most eight steps, two in flight)
      f32x4 a0, a1, a2;
      float b0[4][NT], b1[4][NT], b2[4][NT];
      int q0, q1, q2;
      fetch(a0, b0, &q0);
      fetch(a1, b1, &q1);
      for (int i = 0; i < nsteps; i += 3) {
        fetch(a2, b2, &q2);
        mma(a0, b0, q0);
        fetch(a0, b0, &q0);
        if (i + 1 < nsteps) mma(a1, b1, q1);
        fetch(a1, b1, &q1);
        if (i + 2 < nsteps) mma(a2, b2, q2);
      }
    } else if constexpr (NT >= 2) {
      // wide tiles run at two waves per SIMD (accumulators): two steps of operands in flight instead of one, so that a
      // memory round trip is covered by two steps of MFMA work per wave (three register sets, rotated by full unroll)
      f32x4 a0, a1, a2;
      float b0[4][NT], b1[4][NT], b2[4][NT];
      fetch(a0, b0);
      fetch(a1, b1);
      for (int i = 0; i < nsteps; i += 3) {
        fetch(a2, b2);             // past the end: zeros (tap >= ntaps)
        mma(a0, b0);
        fetch(a0, b0);
        if (i + 1 < nsteps) mma(a1, b1);
        fetch(a1, b1);
        if (i + 2 < nsteps) mma(a2, b2);
      }
    } else {
      f32x4 a0, a1;
      float b0[4][NT], b1[4][NT];
      fetch(a0, b0);
      for (int i = 0; i < nsteps; i += 2) {
        fetch(a1, b1);               // past the end: zeros (tap >= ntaps)
        mma(a0, b0);
        fetch(a0, b0);
        if (i + 1 < nsteps) mma(a1, b1);
      }
    }
  } else {
    for (int ky = 0; ky < g.kh; ++ky) {
      const int iy = oy * g.stride + ky * g.dil - g.pad_t;
      for (int kx = 0; kx < g.kw; ++kx) {
        const int ix = ox * g.stride + kx * g.dil - g.pad_l;
        const bool inb = pv && (unsigned)iy < (unsigned)g.h && (unsigned)ix < (unsigned)g.w;
        const unsigned xo = inb ? img_off + (unsigned)(iy * g.w + ix) * (unsigned)g.x_ld * 4u : kOOB;
        const unsigned wtap = (unsigned)((ky * g.kw + kx) * g.cin) * wrow;
#pragma unroll 4
        for (int c0 = 0; c0 < g.cin; c0 += 2) {
          const int ci = c0 + hh;
          const bool cv = ci < g.cin;
          const float a = Act<T>::ld(rx, cv ? xo + (unsigned)ci * 4u : kOOB);
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const float b = bload(rw, cv ? wtap + (unsigned)ci * wrow + wcol[t] : kOOB);
            acc[t] = MFMA32(a, b, acc[t]);
          }
        }
      }
    }
  }

  // branch-free epilogue: rows past the end and masked columns get an out-of-range offset, which the store drops
  const __amdgpu_buffer_rsrc_t ry = Act<T>::rsrc(y, sz.y_bytes);
  unsigned rowoff[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const unsigned po = (unsigned)tile0 + (unsigned)((i & 3) + 8 * (i >> 2) + 4 * hh);
    rowoff[i] = po < (unsigned)M ? po * (unsigned)g.y_ld * 4u : kOOB;
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int c = cbase + t * 32 + r;
    const float bv = bias ? bload(make_rsrc(bias, (unsigned)g.cout * 4u), colv[t] ? (unsigned)c * 4u : kOOB) : 0.f;
    if (bn.mean && bn.x == nullptr) {   // uniform; inference phase: BatchNormalization (fixed statistics) + activation applied before the store
      const int cc = colv[t] ? c : 0;
      const float mu = bn.mean[cc], is = bn.invstd[cc], ga = bn.gamma[cc], be = bn.beta[cc];
      const float al = bn.act == SSMI_ACT_PRELU ? bn.alpha[cc] : bn.slope;
      float sk[16];
      if (bn.skip) {   // uniform; stride 1: the skip tensor has the output's pixel grid
        const __amdgpu_buffer_rsrc_t rsk = Act<T>::rsrc(bn.skip, (unsigned)(M * bn.skip_ld * 4));
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const unsigned po = (unsigned)tile0 + (unsigned)((i & 3) + 8 * (i >> 2) + 4 * hh);
          sk[i] = Act<T>::ld(rsk, (colv[t] && rowoff[i] != kOOB) ? (po * (unsigned)bn.skip_ld + (unsigned)c) * 4u : kOOB);
        }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float z = ga * (((acc[t][i] + bv) - mu) * is) + be;
        if (bn.skip) z += sk[i];
        Act<T>::st(ry, colv[t] ? rowoff[i] + (unsigned)c * 4u : kOOB, (bn.act == SSMI_ACT_NONE || z > 0.f) ? z : al * z);
      }
    } else if (bn.add || bn.x) {   // uniform; data-gradient mode: skip gradient added, optional BatchNormalization backward sums
      const __amdgpu_buffer_rsrc_t radd = Act<T>::rsrc(bn.add ? bn.add : (const void*)y, bn.add ? sz.y_bytes : 0u);
      const __amdgpu_buffer_rsrc_t rbx = Act<T>::rsrc(bn.x ? bn.x : (const void*)y, bn.x ? (unsigned)(M * bn.ld * 4) : 0u);
      float av[16], xr[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const unsigned po = (unsigned)tile0 + (unsigned)((i & 3) + 8 * (i >> 2) + 4 * hh);
        av[i] = Act<T>::ld(radd, colv[t] ? rowoff[i] + (unsigned)c * 4u : kOOB);
        xr[i] = Act<T>::ld(rbx, (colv[t] && rowoff[i] != kOOB) ? (po * (unsigned)bn.ld + (unsigned)c) * 4u : kOOB);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) Act<T>::st(ry, colv[t] ? rowoff[i] + (unsigned)c * 4u : kOOB, acc[t][i] + bv + av[i]);
      if (bn.x) {
        const int cc = colv[t] ? c : 0;
        const float mu = bn.mean[cc], is = bn.invstd[cc], ga = bn.gamma[cc], be = bn.beta[cc];
        const float al = bn.act == SSMI_ACT_PRELU ? bn.alpha[cc] : bn.slope;
        float s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const bool ok = colv[t] && rowoff[i] != kOOB;
          float d = acc[t][i] + bv + av[i];
          if (bn.drop) {
            const unsigned po = (unsigned)tile0 + (unsigned)((i & 3) + 8 * (i >> 2) + 4 * hh);
            d *= bn.drop[(ok ? po / bn.hw : 0u) * (unsigned)g.cout + (unsigned)cc];
          }
          const float xh = (xr[i] - mu) * is;
          const float z = ga * xh + be;
          const float gr = bn.act == SSMI_ACT_NONE ? 1.f : (z > 0.f ? 1.f : (z < 0.f ? al : 0.f));
          const float dz = ok ? d * gr : 0.f;
          s1 += dz;
          s2 += dz * xh;
          s3 += ok ? d * (z < 0.f ? z : 0.f) : 0.f;
        }
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        s3 += __shfl_xor(s3, 32, 64);
        if (hh == 0) {
          stat_sm[(wave * 3 + 0) * (NT * 32) + t * 32 + r] = s1;
          stat_sm[(wave * 3 + 1) * (NT * 32) + t * 32 + r] = s2;
          stat_sm[(wave * 3 + 2) * (NT * 32) + t * 32 + r] = s3;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) Act<T>::st(ry, colv[t] ? rowoff[i] + (unsigned)c * 4u : kOOB, acc[t][i] + bv);
    }
    if (part) {   // per-wave column sums of y and y^2 over the valid rows (BatchNormalization statistics)
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float v = rowoff[i] != kOOB ? acc[t][i] + bv : 0.f;
        s1 += v;
        s2 += v * v;
      }
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (hh == 0) {
        stat_sm[(wave * 3 + 0) * (NT * 32) + t * 32 + r] = s1;
        stat_sm[(wave * 3 + 1) * (NT * 32) + t * 32 + r] = s2;
      }
    }
  }
  if (bn.x) {      // uniform; never together with `part`
    __syncthreads();
    if (threadIdx.x < NT * 32) {
      const int c = cbase + threadIdx.x;
      if (c < g.cout) {
        for (int k = 0; k < 3; ++k) {
          float sk = 0.f;
#pragma unroll
          for (int wv = 0; wv < 4; ++wv) sk += stat_sm[(wv * 3 + k) * (NT * 32) + threadIdx.x];
          bn.part[((int64_t)blockIdx.x * 3 + k) * g.cout + c] = sk;
        }
      }
    }
  }
  if (part) {
    __syncthreads();
    if (threadIdx.x < NT * 32) {
      const int c = cbase + threadIdx.x;
      if (c < g.cout) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) {
          s1 += stat_sm[(wv * 3 + 0) * (NT * 32) + threadIdx.x];
          s2 += stat_sm[(wv * 3 + 1) * (NT * 32) + threadIdx.x];
        }
        part[((int64_t)blockIdx.x * 2 + 0) * g.cout + c] = s1;
        part[((int64_t)blockIdx.x * 2 + 1) * g.cout + c] = s2;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// data gradient / transposed-convolution forward (gather form, one launch per output phase):
//   dx[n, iy, ix, ci] = bias[ci] + add[..] + sum_{ky,kx : (iy + pad_t - ky*dil) % stride == 0, ...} sum_co
//                        dy[n, (iy+pad_t-ky*dil)/stride, (ix+pad_l-kx*dil)/stride, co] * w[ky,kx,ci,co]
// blockIdx.z enumerates the stride x stride phases (iy % stride, ix % stride) so the set of contributing taps is
// uniform inside a block.  VEC (cout % 8 == 0): 16-byte loads along co for both dy and the kernel.
// ---------------------------------------------------------------------------------------------------
template <typename T, int NT, bool VEC>
__global__ __launch_bounds__(256, (NT == 1 ? 5 : (NT == 2 ? 4 : 2))) void conv_bwd_data_kernel(Geo g, Sizes sz, const T* __restrict__ dy,
                                                            const float* __restrict__ w,
                                                            const T* __restrict__ add,
                                                            const float* __restrict__ bias, T* __restrict__ dx,
                                                            BnBwdArgs bb) {
  __shared__ float bb_sm[4 * 3 * NT * 32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int s = g.stride;
  const int pa = blockIdx.z / s, pb = blockIdx.z % s;
  const int qh = (g.h - pa + s - 1) / s, qw = (g.w - pb + s - 1) / s;
  if (qh <= 0 || qw <= 0) return;
  const int64_t M = (int64_t)g.n * qh * qw;
  const int64_t tile0 = ((int64_t)xcd_tile(blockIdx.x, gridDim.x, sz.xcd & 2) * 4 + wave) * 32;
  if (tile0 >= M && bb.x == nullptr) return;   // with the fused BN backward every wave reaches the barrier
  const __amdgpu_buffer_rsrc_t ry = Act<T>::rsrc(dy, sz.y_bytes), rw = make_rsrc(w, sz.w_bytes);
  const int cbase = blockIdx.y * 32 * NT;

  const unsigned p = (unsigned)tile0 + (unsigned)r;       // M < 2^31: 32-bit pixel arithmetic
  const bool pv = p < (unsigned)M;
  const unsigned pp = pv ? p : 0u;
  const unsigned t1 = pp / (unsigned)qw;
  const int qx = (int)(pp - t1 * (unsigned)qw);
  const unsigned n_ = t1 / (unsigned)qh;
  const int qy = (int)(t1 - n_ * (unsigned)qh);
  const int iy = qy * s + pa, ix = qx * s + pb;
  const unsigned img_off = n_ * (unsigned)(g.ho * g.wo) * (unsigned)g.y_ld * 4u;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  bool colv[NT];
  unsigned wci[NT];   // byte offset of kernel row (ci = lane's column) inside a tap, or kOOB
  const unsigned wrow = (unsigned)g.cout * 4u;
  unsigned wcir[NT], wcim[NT];   // wci split into raw offset and out-of-range mask (bit 31)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int c = cbase + t * 32 + r;   // input channel ci of the forward conv
    colv[t] = c < g.cin;
    wci[t] = colv[t] ? (unsigned)c * wrow : kOOB;
    wcir[t] = colv[t] ? (unsigned)c * wrow : 0u;
    wcim[t] = colv[t] ? 0u : kOOB;
  }

  if (VEC) {
    // Flattened (valid tap of this phase, 8-channel block) reduction, software pipelined like the forward kernel.
    const int ntaps = g.kh * g.kw;
    int tap = -1, c0 = 0;                           // state of the NEXT fetch
    int tky = 0, tkx = -1;                          // (ky, kx) of `tap`, advanced without divisions
    unsigned yo = kOOB, wtap = kOOB;
    auto nexttap = [&]() {                          // advance to the next tap that contributes to this phase (uniform)
      for (++tap; tap < ntaps; ++tap) {
        if (++tkx == g.kw) { tkx = 0; ++tky; }
        const int ky = tky, kx = tkx;
        const int ty0 = pa + g.pad_t - ky * g.dil, tx0 = pb + g.pad_l - kx * g.dil;
        if (s != 1 && (((ty0 % s) + s) % s != 0 || ((tx0 % s) + s) % s != 0)) continue;
        const int ty = iy + g.pad_t - ky * g.dil, tx = ix + g.pad_l - kx * g.dil;
        int oy = ty, ox = tx;
        if (s != 1) {                               // uniform: stride 1 needs no per-lane divisions
          oy = ty / s;                              // exact when ty, tx >= 0
          ox = tx / s;
        }
        const bool inb = pv && ty >= 0 && tx >= 0 && oy < g.ho && ox < g.wo;
        yo = inb ? img_off + (unsigned)(oy * g.wo + ox) * (unsigned)g.y_ld * 4u : kOOB;
        wtap = (unsigned)(tap * g.cin) * wrow;
        return;
      }
      yo = kOOB;
      wtap = kOOB;
    };
    nexttap();
    int nvalid = s == 1 ? ntaps : 0;
    if (s != 1) {
      for (int t = 0; t < ntaps; ++t) {
        const int ky = t / g.kw, kx = t - ky * g.kw;
        const int ty0 = pa + g.pad_t - ky * g.dil, tx0 = pb + g.pad_l - kx * g.dil;
        nvalid += (((ty0 % s) + s) % s == 0 && ((tx0 % s) + s) % s == 0) ? 1 : 0;
      }
    }
    auto fetch = [&](f32x4& a, f32x4 (&b)[NT]) {
      const unsigned co4 = (unsigned)(c0 + 4 * hh) * 4u;
      a = Act<T>::ld4(ry, (yo + co4) | (yo & kOOB));     // bit 31 of an invalid base is carried over: out of range
      const unsigned wb = wtap + co4, wm = wtap & kOOB;
#pragma unroll
      for (int t = 0; t < NT; ++t) b[t] = bload4(rw, (wb + wcir[t]) | wm | wcim[t]);
      c0 += 8;
      if (c0 >= g.cout) {
        c0 = 0;
        nexttap();
      }
    };
    auto mma = [&](const f32x4& a, const f32x4 (&b)[NT]) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = MFMA32(a[j], b[t][j], acc[t]);
    };
    const int nsteps = nvalid * (g.cout >> 3);
    if constexpr (NT >= 2) {       // two steps of operands in flight, as in the wide forward kernel
      f32x4 a0, a1, a2, b0[NT], b1[NT], b2[NT];
      fetch(a0, b0);
      fetch(a1, b1);
      for (int i = 0; i < nsteps; i += 3) {
        fetch(a2, b2);             // past the end: zeros
        mma(a0, b0);
        fetch(a0, b0);
        if (i + 1 < nsteps) mma(a1, b1);
        fetch(a1, b1);
        if (i + 2 < nsteps) mma(a2, b2);
      }
    } else {
      f32x4 a0, a1, b0[NT], b1[NT];
      fetch(a0, b0);
      for (int i = 0; i < nsteps; i += 2) {
        fetch(a1, b1);               // past the end: zeros
        mma(a0, b0);
        fetch(a0, b0);
        if (i + 1 < nsteps) mma(a1, b1);
      }
    }
  } else {
    for (int ky = 0; ky < g.kh; ++ky) {
      const int ty0 = pa + g.pad_t - ky * g.dil;
      if (((ty0 % s) + s) % s != 0) continue;  // uniform
      const int ty = iy + g.pad_t - ky * g.dil;
      const int oy = s == 1 ? ty : ty / s;  // exact when ty >= 0
      for (int kx = 0; kx < g.kw; ++kx) {
        const int tx0 = pb + g.pad_l - kx * g.dil;
        if (((tx0 % s) + s) % s != 0) continue;  // uniform
        const int tx = ix + g.pad_l - kx * g.dil;
        const int ox = s == 1 ? tx : tx / s;
        const bool inb = pv && ty >= 0 && tx >= 0 && oy < g.ho && ox < g.wo;
        const unsigned yo = inb ? img_off + (unsigned)(oy * g.wo + ox) * (unsigned)g.y_ld * 4u : kOOB;
        const unsigned wtap = (unsigned)((ky * g.kw + kx) * g.cin) * wrow;
#pragma unroll 4
        for (int c0 = 0; c0 < g.cout; c0 += 2) {
          const int co = c0 + hh;
          const bool cv = co < g.cout;
          const float a = Act<T>::ld(ry, cv ? yo + (unsigned)co * 4u : kOOB);
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const float b = bload(rw, cv ? wtap + wci[t] + (unsigned)co * 4u : kOOB);
            acc[t] = MFMA32(a, b, acc[t]);
          }
        }
      }
    }
  }

  // branch-free epilogue: every skip-gradient load is issued before the first store; rows past the end and masked
  // columns use an out-of-range offset (loads return 0, stores are dropped)
  const __amdgpu_buffer_rsrc_t rdx = Act<T>::rsrc(dx, sz.x_bytes);
  const __amdgpu_buffer_rsrc_t radd = Act<T>::rsrc(add ? add : dx, add ? sz.x_bytes : 0u);
  unsigned rowoff[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const unsigned po = (unsigned)tile0 + (unsigned)((i & 3) + 8 * (i >> 2) + 4 * hh);
    const unsigned pq = po < (unsigned)M ? po : 0u;
    unsigned pix = pq;
    if (s != 1) {
      const unsigned t2 = pq / (unsigned)qw;
      const unsigned ox_ = pq - t2 * (unsigned)qw;
      const unsigned nn = t2 / (unsigned)qh;
      const unsigned oy_ = t2 - nn * (unsigned)qh;
      pix = (nn * (unsigned)g.h + (oy_ * (unsigned)s + (unsigned)pa)) * (unsigned)g.w + (ox_ * (unsigned)s + (unsigned)pb);
    }
    rowoff[i] = po < (unsigned)M ? pix * (unsigned)g.x_ld * 4u : kOOB;
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int c = cbase + t * 32 + r;
    const float bv = bias ? bload(make_rsrc(bias, (unsigned)g.cin * 4u), colv[t] ? (unsigned)c * 4u : kOOB) : 0.f;
    float av[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) av[i] = Act<T>::ld(radd, colv[t] ? rowoff[i] + (unsigned)c * 4u : kOOB);
#pragma unroll
    for (int i = 0; i < 16; ++i) Act<T>::st(rdx, colv[t] ? rowoff[i] + (unsigned)c * 4u : kOOB, acc[t][i] + bv + av[i]);
    if (bb.x) {   // uniform; stride 1: the output pixel index is the row index
      const int cc = colv[t] ? c : 0;
      const float mu = bb.mean[cc], is = bb.invstd[cc], ga = bb.gamma[cc], be = bb.beta[cc];
      const float al = bb.act == SSMI_ACT_PRELU ? bb.alpha[cc] : bb.slope;
      const __amdgpu_buffer_rsrc_t rbx = Act<T>::rsrc(bb.x, (unsigned)((int64_t)g.n * g.h * g.w * bb.ld * 4));
      float xr[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const unsigned po = (unsigned)tile0 + (unsigned)((i & 3) + 8 * (i >> 2) + 4 * hh);
        xr[i] = Act<T>::ld(rbx, (colv[t] && rowoff[i] != kOOB) ? (po * (unsigned)bb.ld + (unsigned)c) * 4u : kOOB);
      }
      float s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const bool ok = colv[t] && rowoff[i] != kOOB;
        float d = acc[t][i] + bv + av[i];
        if (bb.drop) {
          const unsigned po = (unsigned)tile0 + (unsigned)((i & 3) + 8 * (i >> 2) + 4 * hh);
          d *= bb.drop[(ok ? po / bb.hw : 0u) * (unsigned)g.cin + (unsigned)cc];
        }
        const float xh = (xr[i] - mu) * is;
        const float z = ga * xh + be;
        const float gr = bb.act == SSMI_ACT_NONE ? 1.f : (z > 0.f ? 1.f : (z < 0.f ? al : 0.f));
        const float dz = ok ? d * gr : 0.f;
        s1 += dz;
        s2 += dz * xh;
        s3 += ok ? d * (z < 0.f ? z : 0.f) : 0.f;
      }
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      s3 += __shfl_xor(s3, 32, 64);
      if (hh == 0) {
        bb_sm[(wave * 3 + 0) * (NT * 32) + t * 32 + r] = s1;
        bb_sm[(wave * 3 + 1) * (NT * 32) + t * 32 + r] = s2;
        bb_sm[(wave * 3 + 2) * (NT * 32) + t * 32 + r] = s3;
      }
    }
  }
  if (bb.x) {
    __syncthreads();
    if (threadIdx.x < NT * 32) {
      const int c = cbase + threadIdx.x;
      if (c < g.cin) {
        for (int k = 0; k < 3; ++k) {
          float sk = 0.f;
#pragma unroll
          for (int wv = 0; wv < 4; ++wv) sk += bb_sm[(wv * 3 + k) * (NT * 32) + threadIdx.x];
          bb.part[((int64_t)blockIdx.x * 3 + k) * g.cin + c] = sk;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// LDS-weight implicit GEMM, shared by the forward convolution and the data gradient / transposed forward:
//     out[p, o] = sum over valid taps t, reduction channels r :  in[pix(p, t), r] * Wl[t][r][o]
//   forward   : in = x,  r = cin,  o = cout, Wl[t][r][o] = w[t][r][o]
//   data grad : in = dy, r = cout, o = cin,  Wl[t][r][o] = w[t][o][r]   (one launch per stride phase: blockIdx.z)
// The whole kernel matrix is staged once per workgroup in LDS (row stride WS words), so the B operand of every MFMA
// is a conflict-free ds_read_b32 instead of a vector-memory load.  A wave owns MT pixel tiles x NT column tiles; the
// A operand is one 16-byte buffer load per lane per k-block (4 consecutive reduction channels of the lane's pixel;
// lane group kq takes channels 4*kq.. of the block, the same permutation is applied to the B rows), double buffered so
// the loads of block b+1 are in flight while block b is multiplied.  k runs over (valid tap, channel) flattened, so
// cin = 4 layers put four taps into one MFMA.  M16 selects v_mfma_f32_16x16x4_f32 (16-pixel x 16-channel tiles) for
// outputs of <= 16 channels.  VEC=false (reduction channels not a multiple of 4) feeds one channel per lane.
// ---------------------------------------------------------------------------------------------------
struct GemmArgs {
  int mode;                      // 0 forward, 1 data gradient
  int C_r, C_o;
  int in_h, in_w, in_ld;         // tensor the A operand is gathered from
  int out_ld;
  int WS;                        // LDS row stride in words
  int q_blk, r_blk, q0_div;      // k-block advance: taps / channels per block (KB / C_r, KB % C_r)
  int ngroups_max;
  // TILE variant: a workgroup owns 8 x 32 output pixels and stages their input halo patch [PH][PW][CP] in LDS
  int tiles_x, tiles_y;          // tiles per image
  int pmin_y, pmin_x;            // smallest tap offset (patch origin relative to the tile origin)
  int PH, PW, CP;                // patch rows, columns, words per pixel (C_r + 4 pad)
  int patch_off;                 // word offset of the patch inside the dynamic LDS block
  const void* wpack;             // conv_regw<bf16, ..., true>: the layer's bfloat16 kernel pack [tap][cout][cin]
};
static constexpr int kMaxTaps = 32;

template <typename T, bool M16, int MT, int NT, bool VEC, int D, bool TILE>
__global__ __launch_bounds__(256) void conv_gemm_lds(Geo g, Sizes sz, GemmArgs a, const T* __restrict__ in,
                                                     const float* __restrict__ w, const float* __restrict__ bias,
                                                     const T* __restrict__ add, T* __restrict__ out,
                                                     float* __restrict__ part, BnBwdArgs bb) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int TS = M16 ? 16 : 32;
  constexpr int KQ = 64 / TS;
  constexpr int NR = M16 ? 4 : 16;
  constexpr int EW = VEC ? 4 : 1;
  constexpr int KB = KQ * EW;             // reduction elements per k-block
  typedef float accv __attribute__((ext_vector_type(NR)));
  // [0]=nvt, then {dy, dx, wl_word_offset} per valid tap, the source tap of every staged slot, and the byte offset
  // (dy * in_w + dx) * in_ld * 4 of the tap's input pixel relative to the output pixel's own
  int* tab = reinterpret_cast<int*>(smem);
  float* Wl = smem + 4 + 5 * kMaxTaps;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & (TS - 1), kq = lane / TS;
  const int s = g.stride;
  const bool bwd = a.mode == 1;
  const int pa = bwd ? blockIdx.z / s : 0, pb = bwd ? blockIdx.z % s : 0;
  const int qh = bwd ? (g.h - pa + s - 1) / s : g.ho, qw = bwd ? (g.w - pb + s - 1) / s : g.wo;
  if (qh <= 0 || qw <= 0) return;
  const int ntaps = g.kh * g.kw;

  // ---- tap table + kernel matrix into LDS ----
  // Lane t of the first wave classifies tap t; a ballot compacts the valid taps in order (a serial loop over the taps
  // in one thread costs ~150 dependent instructions per tap in front of the first barrier).
  if (tid < 64) {
    const int t = tid;
    bool valid = false;
    int dyo = 0, dxo = 0;
    if (t < ntaps) {
      const int ky = t / g.kw, kx = t - ky * g.kw;
      if (!bwd) {
        dyo = ky * g.dil - g.pad_t;
        dxo = kx * g.dil - g.pad_l;
        valid = true;
      } else {
        const int ty = pa + g.pad_t - ky * g.dil, tx = pb + g.pad_l - kx * g.dil;
        valid = (((ty % s) + s) % s) == 0 && (((tx % s) + s) % s) == 0;
        dyo = (ty >= 0 ? ty / s : -((-ty) / s));   // exact: ty is a multiple of s
        dxo = (tx >= 0 ? tx / s : -((-tx) / s));
      }
    }
    const unsigned long long mask = __ballot(valid);
    const int n = __popcll(mask & ((1ull << t) - 1ull));
    if (valid) {
      tab[1 + 3 * n] = dyo;
      tab[2 + 3 * n] = dxo;
      tab[3 + 3 * n] = n * a.C_r * a.WS;     // only the valid taps are staged, packed
      tab[4 + 3 * kMaxTaps + n] = t;         // source tap of staged slot n
      tab[4 + 4 * kMaxTaps + n] = (dyo * a.in_w + dxo) * a.in_ld * 4;
    }
    if (t == 0) tab[0] = __popcll(mask);
  }
  __syncthreads();
  {
    const int nv = tab[0];
    const int copad = NT * TS * ((a.C_o + NT * TS - 1) / (NT * TS));
    const int rows = nv * a.C_r;
    if (copad > a.C_o) {   // zero the pad columns
      const int padw = copad - a.C_o;
      for (int e = tid; e < rows * padw; e += 256) Wl[(e / padw) * a.WS + a.C_o + e % padw] = 0.f;
    }
    const int per_tap = g.cin * g.cout;
    const __amdgpu_buffer_rsrc_t rwt = make_rsrc(w, sz.w_bytes);
    // Eight elements per thread and round, their loads issued together (a load-then-store loop costs one L2 round trip
    // per element); elements past the end read zeros and are not stored.
    if (!bwd && a.WS == g.cout && (g.cout & 3) == 0 && ((reinterpret_cast<uintptr_t>(w) & 15) == 0)) {
      // forward, unpadded rows: the LDS image is the kernel itself (all taps are valid and in order)
      f32x4* dst = reinterpret_cast<f32x4*>(Wl);
      const int n4 = (nv * per_tap) >> 2;
      for (int e0 = tid; e0 < n4; e0 += 256 * 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + u * 256;
          v[u] = bload4(rwt, e < n4 ? (unsigned)e * 16u : kOOB);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + u * 256;
          if (e < n4) dst[e] = v[u];
        }
      }
    } else {
      // element e = (vi * cin + ci) * cout + co walks e = tid, tid + 256, ...; (vi, ci, co) advance incrementally (no
      // division per element); eight loads per round are in flight, across tap boundaries
      const int total = nv * per_tap;
      const int dvi = 256 / per_tap, drem = 256 - dvi * per_tap;
      const int dci = drem / g.cout, dco = drem - dci * g.cout;
      int vi = tid / per_tap;
      const int q0 = tid - vi * per_tap;
      int ci = q0 / g.cout, co = q0 - ci * g.cout;
      for (int e0 = tid; e0 < total; e0 += 256 * 8) {
        float v[8];
        int di[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const bool ok = e0 + u * 256 < total;
          const int srct = tab[4 + 3 * kMaxTaps + (ok ? vi : 0)];
          v[u] = bload(rwt, ok ? (unsigned)(srct * per_tap + ci * g.cout + co) * 4u : kOOB);
          di[u] = ok ? vi * a.C_r * a.WS + (bwd ? co * a.WS + ci : ci * a.WS + co) : -1;
          co += dco;
          if (co >= g.cout) { co -= g.cout; ++ci; }
          ci += dci;
          if (ci >= g.cin) { ci -= g.cin; ++vi; }
          vi += dvi;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (di[u] >= 0) Wl[di[u]] = v[u];
      }
    }
  }
  __syncthreads();
  const int nvt = tab[0];
  const int K = nvt * a.C_r;
  const int nblk = (K + KB - 1) / KB;
  const __amdgpu_buffer_rsrc_t rin = Act<T>::rsrc(in, bwd ? sz.y_bytes : sz.x_bytes);
  const unsigned out_bytes = bwd ? sz.x_bytes : sz.y_bytes;
  const __amdgpu_buffer_rsrc_t rout = Act<T>::rsrc(out, out_bytes);
  const __amdgpu_buffer_rsrc_t radd = Act<T>::rsrc(add ? add : out, add ? out_bytes : 0u);
  const __amdgpu_buffer_rsrc_t rbias = make_rsrc(bias ? (const void*)bias : (const void*)out, bias ? (unsigned)a.C_o * 4u : 0u);
  const int sm_ = bwd ? 1 : s;       // input coordinate = q * sm_ + tap offset
  const int64_t M = (int64_t)g.n * qh * qw;
  const int64_t ngroups = (M + MT * TS - 1) / (MT * TS);
  const int cbase0 = blockIdx.y * NT * TS;
  const int kq_tap0 = (kq * EW) / a.C_r, kq_r0 = (kq * EW) % a.C_r;

  // running column sums: forward with statistics: y, y^2; data gradient with the fused BN backward: dz, dz*xhat, d*min(z,0)
  float st1[NT], st2[NT], st3[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) st1[t] = st2[t] = st3[t] = 0.f;
  const __amdgpu_buffer_rsrc_t rbx = Act<T>::rsrc(bb.x ? bb.x : (const void*)out, bb.x ? (unsigned)((int64_t)g.n * g.h * g.w * bb.ld * 4) : 0u);

  static_assert(!TILE || (MT * TS == 64 && NT == 1 && VEC), "tile variant: 64 pixels per wave, one column tile");
  float* patch = smem + a.patch_off;
  const int64_t nloop = TILE ? (int64_t)g.n * a.tiles_y * a.tiles_x : ngroups;
  const int64_t bxt = (int64_t)xcd_tile(blockIdx.x, gridDim.x, sz.xcd & 4);
  for (int64_t grp = TILE ? bxt : bxt * 4 + wave; grp < nloop;
       grp += TILE ? (int64_t)gridDim.x : (int64_t)gridDim.x * 4) {
    const int64_t tile0 = grp * (MT * TS);
    bool pv[MT];
    int by[MT], bx[MT], img[MT];
    int lbase[MT];                 // TILE: word offset of the lane's pixel inside the patch
    unsigned pbase[MT];            // byte offset of the lane's pixel at tap offset (0, 0)
    int n_t = 0, ty0 = 0, tx0 = 0; // TILE: image and origin of this workgroup's tile (uniform)
    if constexpr (TILE) {
      const unsigned tl = (unsigned)grp;
      const unsigned per_img = (unsigned)(a.tiles_y * a.tiles_x);
      n_t = (int)(tl / per_img);
      const unsigned rem = tl - (unsigned)n_t * per_img;
      const unsigned tyi = rem / (unsigned)a.tiles_x;
      ty0 = (int)tyi * 8;
      tx0 = (int)(rem - tyi * (unsigned)a.tiles_x) * 32;
      __syncthreads();             // every wave is done with the previous tile's patch
      {
        // stage the halo patch: element e = (py * PW + px) * C4 + c4 walks e = tid, tid + 256, ... with incremental
        // coordinates; pixels outside the image read zeros through the bounds check; eight loads per round in flight
        const int C4 = a.C_r >> 2;
        const int elements = a.PH * a.PW * C4;
        const int dpp = 256 / C4, dc4 = 256 - dpp * C4;
        const int dpy = dpp / a.PW, dpx = dpp - dpy * a.PW;
        const int pp0 = tid / C4;
        int c4 = tid - pp0 * C4;
        int py = pp0 / a.PW, px = pp0 - py * a.PW;
        const int gy0 = ty0 + a.pmin_y, gx0 = tx0 + a.pmin_x;
        for (int e0 = tid; e0 < elements; e0 += 256 * 8) {
          f32x4 v[8];
          int dw_[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const bool ev = e0 + u * 256 < elements;
            const int gy = gy0 + py, gx = gx0 + px;
            const bool ok = ev && (unsigned)gy < (unsigned)a.in_h && (unsigned)gx < (unsigned)a.in_w;
            v[u] = Act<T>::ld4(rin, ok ? ((unsigned)((n_t * a.in_h + gy) * a.in_w + gx) * (unsigned)a.in_ld + (unsigned)(c4 * 4)) * 4u : kOOB);
            dw_[u] = ev ? (py * a.PW + px) * a.CP + c4 * 4 : -1;
            c4 += dc4;
            if (c4 >= C4) { c4 -= C4; ++px; }
            px += dpx;
            if (px >= a.PW) { px -= a.PW; ++py; }
            py += dpy;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (dw_[u] >= 0) *reinterpret_cast<f32x4*>(patch + dw_[u]) = v[u];
        }
        if (tid < 4) patch[a.PH * a.PW * a.CP + tid] = 0.f;   // the zero operand of k-blocks past the last tap
      }
      __syncthreads();
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int ly = wave * 2 + (M16 ? (m >> 1) : m), lx = (M16 ? (m & 1) * 16 : 0) + c;
        lbase[m] = (ly * a.PW + lx) * a.CP;
        pv[m] = true; by[m] = bx[m] = img[m] = 0;
        pbase[m] = 0u;
      }
    } else {
      // pixel coordinates in 32-bit arithmetic (M < 2^31: tensors are below 2 GiB); two divisions per pixel
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const unsigned p = (unsigned)tile0 + (unsigned)(m * TS + c);
        pv[m] = p < (unsigned)M;
        const unsigned pp = pv[m] ? p : 0u;
        const unsigned t1 = pp / (unsigned)qw;
        const unsigned qx = pp - t1 * (unsigned)qw;
        const unsigned n_ = t1 / (unsigned)qh;
        by[m] = (int)(t1 - n_ * (unsigned)qh) * sm_;
        bx[m] = (int)qx * sm_;
        img[m] = (int)n_ * a.in_h * a.in_w;
        lbase[m] = 0;
        // byte offset of the pixel the zero tap reads; a tap adds its table offset (no multiplication per tap)
        pbase[m] = (unsigned)(img[m] + by[m] * a.in_w + bx[m]) * (unsigned)a.in_ld * 4u;
      }
    }
    // flat output pixel index of accumulator row `row` of pixel tile m, or 0xFFFFFFFF outside the tensor
    auto row_pixel = [&](int m, int row) -> unsigned {
      if constexpr (TILE) {
        const int yy = ty0 + wave * 2 + (M16 ? (m >> 1) : m), xx = tx0 + (M16 ? (m & 1) * 16 : 0) + row;
        return (yy < qh && xx < qw) ? (unsigned)((n_t * qh + yy) * qw + xx) : 0xFFFFFFFFu;
      } else {
        const unsigned po = (unsigned)tile0 + (unsigned)(m * TS + row);
        return po < (unsigned)M ? po : 0xFFFFFFFFu;
      }
    };
    accv acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < NR; ++i) acc[m][t][i] = 0.f;

    int tapi = kq_tap0, r0 = kq_r0;
    f32x4 A[D][MT];
    int wb[D];
    // the tap-table entry of the NEXT fetch is read at the end of the current one, off the critical path
    int nty, ntx, nwb, ntoff;
    auto lookup = [&]() {
      const int ts_ = tapi < nvt ? tapi : 0;
      nty = tab[1 + 3 * ts_];
      ntx = tab[2 + 3 * ts_];
      nwb = tab[3 + 3 * ts_];
      ntoff = tab[4 + 4 * kMaxTaps + ts_];
    };
    lookup();
    auto fetch = [&](f32x4 (&Ab)[MT], int& wbase) {
      const bool tv = tapi < nvt;
      const int ty = nty, tx = ntx;
      const unsigned toff = (unsigned)ntoff + (unsigned)r0 * 4u;
      wbase = nwb + r0 * a.WS + cbase0 + c;
      if constexpr (TILE) {
        const int toff = tv ? ((ty - a.pmin_y) * a.PW + (tx - a.pmin_x)) * a.CP + r0 : -1;
#pragma unroll
        for (int m = 0; m < MT; ++m)
          Ab[m] = *reinterpret_cast<const f32x4*>(patch + (toff >= 0 ? lbase[m] + toff : a.PH * a.PW * a.CP));
      } else {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int iy = by[m] + ty, ix = bx[m] + tx;
          const bool ok = tv && pv[m] && (unsigned)iy < (unsigned)a.in_h && (unsigned)ix < (unsigned)a.in_w;
          const unsigned off = ok ? pbase[m] + toff : kOOB;
          if (VEC) Ab[m] = Act<T>::ld4(rin, off);
          else Ab[m][0] = Act<T>::ld(rin, off);
        }
      }
      tapi += a.q_blk;
      r0 += a.r_blk;
      if (r0 >= a.C_r) { r0 -= a.C_r; ++tapi; }
      lookup();
    };
    auto mma = [&](f32x4 (&Ab)[MT], int wbase) {
#pragma unroll
      for (int j = 0; j < EW; ++j) {
        float b[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) b[t] = Wl[wbase + j * a.WS + t * TS];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            if constexpr (M16) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ab[m][j], b[t], acc[m][t], 0, 0, 0);
            else acc[m][t] = MFMA32(Ab[m][j], b[t], acc[m][t]);
          }
      }
    };
    // D k-blocks of operand loads are in flight ahead of the multiply: a wave that owns a single pixel group (the usual
    // case: the launch is one round of waves) pays one memory round trip for up to D blocks instead of one per block
#pragma unroll
    for (int d = 0; d < D; ++d) fetch(A[d], wb[d]);
    for (int blk = 0; blk < nblk; blk += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        if (blk + d < nblk) mma(A[d], wb[d]);
        fetch(A[d], wb[d]);          // block blk + d + D; past the end every lane reads zeros (tapi >= nvt)
      }
    }

    // ---- epilogue (branch-free: out-of-range offsets make masked loads return 0 and masked stores vanish) ----
    unsigned rowoff[MT][NR];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const int row = M16 ? kq * 4 + i : (i & 3) + 8 * (i >> 2) + 4 * kq;
        const unsigned po = row_pixel(m, row);
        const unsigned pq = po != 0xFFFFFFFFu ? po : 0u;
        unsigned pix = pq;
        if (bwd && s != 1) {
          const unsigned t2 = pq / (unsigned)qw;
          const unsigned ox_ = pq - t2 * (unsigned)qw;
          const unsigned nn = t2 / (unsigned)qh;
          const unsigned oy_ = t2 - nn * (unsigned)qh;
          pix = (nn * (unsigned)g.h + (oy_ * (unsigned)s + (unsigned)pa)) * (unsigned)g.w + (ox_ * (unsigned)s + (unsigned)pb);
        }
        rowoff[m][i] = po != 0xFFFFFFFFu ? pix * (unsigned)a.out_ld * 4u : kOOB;
      }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = cbase0 + t * TS + c;
      const bool cv = col < a.C_o;
      const float bvv = bias ? bload(rbias, cv ? (unsigned)col * 4u : kOOB) : 0.f;
      float av[MT][NR];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < NR; ++i) av[m][i] = Act<T>::ld(radd, cv ? rowoff[m][i] + (unsigned)col * 4u : kOOB);
      if (!bwd && bb.mean) {   // uniform; inference-phase forward: BatchNormalization (fixed statistics) + activation
        const int cc = cv ? col : 0;
        const float mu = bb.mean[cc], is = bb.invstd[cc], ga = bb.gamma[cc], be = bb.beta[cc];
        const float al = bb.act == SSMI_ACT_PRELU ? bb.alpha[cc] : bb.slope;
        float sk[MT][NR];
        if (bb.skip) {   // uniform; the skip tensor has the output's pixel grid
          const __amdgpu_buffer_rsrc_t rsk = Act<T>::rsrc(bb.skip, (unsigned)(M * bb.skip_ld * 4));
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int i = 0; i < NR; ++i) {
              const int row = M16 ? kq * 4 + i : (i & 3) + 8 * (i >> 2) + 4 * kq;
              const unsigned po = row_pixel(m, row);
              sk[m][i] = Act<T>::ld(rsk, (cv && rowoff[m][i] != kOOB) ? (po * (unsigned)bb.skip_ld + (unsigned)col) * 4u : kOOB);
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int i = 0; i < NR; ++i) {
            float z = ga * (((acc[m][t][i] + bvv) - mu) * is) + be;
            if (bb.skip) z += sk[m][i];
            Act<T>::st(rout, cv ? rowoff[m][i] + (unsigned)col * 4u : kOOB, (bb.act == SSMI_ACT_NONE || z > 0.f) ? z : al * z);
          }
      } else {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int i = 0; i < NR; ++i)
            Act<T>::st(rout, cv ? rowoff[m][i] + (unsigned)col * 4u : kOOB, acc[m][t][i] + bvv + av[m][i]);
      }
      if (part) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int i = 0; i < NR; ++i) {
            const float v = rowoff[m][i] != kOOB ? acc[m][t][i] + bvv : 0.f;
            st1[t] += v;
            st2[t] += v * v;
          }
      }
      if (bb.x) {   // uniform; stride-1 data gradient: the output pixel index is the row index
        const int cc = cv ? col : 0;
        const float mu = bb.mean[cc], is = bb.invstd[cc], ga = bb.gamma[cc], be = bb.beta[cc];
        const float al = bb.act == SSMI_ACT_PRELU ? bb.alpha[cc] : bb.slope;
        float xr[MT][NR];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int i = 0; i < NR; ++i) {
            const int row = M16 ? kq * 4 + i : (i & 3) + 8 * (i >> 2) + 4 * kq;
            const unsigned po = row_pixel(m, row);
            xr[m][i] = Act<T>::ld(rbx, (cv && rowoff[m][i] != kOOB) ? (po * (unsigned)bb.ld + (unsigned)col) * 4u : kOOB);
          }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int i = 0; i < NR; ++i) {
            const bool ok = cv && rowoff[m][i] != kOOB;
            float d = acc[m][t][i] + bvv + av[m][i];
            if (bb.drop) {
              const int row = M16 ? kq * 4 + i : (i & 3) + 8 * (i >> 2) + 4 * kq;
              const unsigned po = row_pixel(m, row);
              d *= bb.drop[(ok ? po / bb.hw : 0u) * (unsigned)a.C_o + (unsigned)cc];
            }
            const float xh = (xr[m][i] - mu) * is;
            const float z = ga * xh + be;
            const float gr = bb.act == SSMI_ACT_NONE ? 1.f : (z > 0.f ? 1.f : (z < 0.f ? al : 0.f));
            const float dz = ok ? d * gr : 0.f;
            st1[t] += dz;
            st2[t] += dz * xh;
            st3[t] += ok ? d * (z < 0.f ? z : 0.f) : 0.f;
          }
      }
    }
  }
  if (part || bb.x) {   // uniform: combine the lane groups that share a column, then the four waves through LDS
    const int ns = bb.x ? 3 : 2;
    float* dst = bb.x ? bb.part : part;
    __syncthreads();             // every wave is done with the kernel matrix in LDS
    float* red = smem;           // [4 waves][3][NT*TS]
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float s1 = st1[t], s2 = st2[t], s3 = st3[t];
#pragma unroll
      for (int o = TS; o < 64; o <<= 1) {
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
        s3 += __shfl_xor(s3, o, 64);
      }
      if (kq == 0) {
        red[(wave * 3 + 0) * (NT * TS) + t * TS + c] = s1;
        red[(wave * 3 + 1) * (NT * TS) + t * TS + c] = s2;
        red[(wave * 3 + 2) * (NT * TS) + t * TS + c] = s3;
      }
    }
    __syncthreads();
    if (tid < NT * TS) {
      const int col = cbase0 + tid;
      if (col < a.C_o) {
        for (int k = 0; k < ns; ++k) {
          float sk = 0.f;
#pragma unroll
          for (int wv = 0; wv < 4; ++wv) sk += red[(wv * 3 + k) * (NT * TS) + tid];
          dst[((int64_t)blockIdx.x * ns + k) * a.C_o + col] = sk;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Four reduction channels per tap (the last decoder stage of the ENets: 3x3 4->4, its stride-2 transposed form, 1x1 4->16 and
// the data gradient of 1x1 16->4).  Padded to 16-wide matrix tiles these layers ran at a tenth of what their bytes allow
// (3x3 4->4 at 256x256, batch 16: 54 us for 34 MB); here they are plain vector arithmetic: a thread owns four output channels
// of one output pixel (OG threads per pixel), gathers one 4-channel operand per tap with a single load and multiplies it by
// the tap's 4 x 4 block of kernel words, which every lane reads from the same LDS address.
//   forward (mode 0): any stride / dilation;  data gradient (mode 1): any stride - the taps a pixel's parity excludes are
//   masked per lane, so the stride-2 transposed convolution is one launch over the whole output.
// Epilogues as conv_gemm_lds: bias, addend, BatchNormalization statistics rows (part) or BN-backward sums (bb.x), one row
// per workgroup; inference phase: BatchNormalization with fixed statistics + activation (+ skip tensor) in front of the store.
// ---------------------------------------------------------------------------------------------------
// RB = 4 (single tap only): sixteen reduction channels into four outputs - 1x1 16->4 and the data gradient of 1x1 4->16,
// four operand loads per pixel; on 16-wide matrix tiles a wave walked sixteen 16-pixel groups one memory round trip at a time.
constexpr int kC4MaxTaps = 9;
template <typename T, int OG, int RB = 1>
__global__ __launch_bounds__(256) void conv_c4_kernel(Geo g, Sizes sz, GemmArgs a, const T* __restrict__ in,
                                                      const float* __restrict__ w, const float* __restrict__ bias,
                                                      const T* __restrict__ add, T* __restrict__ out,
                                                      float* __restrict__ part, BnBwdArgs bb) {
  constexpr int CO = 4 * OG, CR = 4 * RB;
  static_assert(RB == 1 || OG == 1, "shape");
  __shared__ __attribute__((aligned(16))) float wl[(RB == 1 ? kC4MaxTaps : 1) * CR * CO];      // [tap][reduction channel][output channel]
  __shared__ float red[4 * 3 * CO];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool bwd = a.mode == 1;
  const int ntaps = g.kh * g.kw;
  for (int e = tid; e < ntaps * CR * CO; e += 256) {
    const int t = e / (CR * CO), rem = e - t * (CR * CO), r = rem / CO, o = rem - r * CO;
    wl[e] = bwd ? w[(t * g.cin + o) * g.cout + r] : w[(t * g.cin + r) * g.cout + o];
  }
  __syncthreads();
  const int oh = bwd ? g.h : g.ho, ow = bwd ? g.w : g.wo;      // output grid
  const unsigned total = (unsigned)((int64_t)g.n * oh * ow) * (unsigned)OG;
  const __amdgpu_buffer_rsrc_t rin = Act<T>::rsrc(in, bwd ? sz.y_bytes : sz.x_bytes);
  const int og = tid & (OG - 1);            // constant per thread: the loop stride is a multiple of OG
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (bias) bv = f32x4{bias[4 * og], bias[4 * og + 1], bias[4 * og + 2], bias[4 * og + 3]};
  f32x4 mu = bv, is = bv, ga = bv, be = bv, al = bv;
  const bool infer = !bwd && bb.mean != nullptr && bb.x == nullptr;      // inference phase: BatchNormalization + activation (+ skip) in front of the store
  if (bb.x || infer) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int cc = 4 * og + j;
      mu[j] = bb.mean[cc]; is[j] = bb.invstd[cc]; ga[j] = bb.gamma[cc]; be[j] = bb.beta[cc];
      al[j] = bb.act == SSMI_ACT_PRELU ? bb.alpha[cc] : bb.slope;
    }
  }
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1, s3 = s1;
  for (unsigned wi = blockIdx.x * 256u + (unsigned)tid; wi < total; wi += gridDim.x * 256u) {
    const unsigned p = wi / (unsigned)OG;
    const unsigned t1 = p / (unsigned)ow;
    const int ox = (int)(p - t1 * (unsigned)ow);
    const unsigned n_ = t1 / (unsigned)oh;
    const int oy = (int)(t1 - n_ * (unsigned)oh);
    const unsigned img = n_ * (unsigned)(a.in_h * a.in_w);
    f32x4 acc = bv;
    for (int t = 0; t < ntaps; ++t) {
      const int ky = t / g.kw, kx = t - ky * g.kw;
      int iy, ix;
      bool ok;
      if (!bwd) {
        iy = oy * g.stride + ky * g.dil - g.pad_t;
        ix = ox * g.stride + kx * g.dil - g.pad_l;
        ok = true;
      } else {
        const int ty = oy + g.pad_t - ky * g.dil, tx = ox + g.pad_l - kx * g.dil;
        iy = ty / g.stride; ix = tx / g.stride;
        ok = ty >= 0 && tx >= 0 && iy * g.stride == ty && ix * g.stride == tx;
      }
      ok = ok && (unsigned)iy < (unsigned)a.in_h && (unsigned)ix < (unsigned)a.in_w;
      const unsigned off = ok ? (img + (unsigned)(iy * a.in_w + ix)) * (unsigned)a.in_ld * 4u : kOOB;
      f32x4 v[RB];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) v[rb] = Act<T>::ld4(rin, off == kOOB ? kOOB : off + 16u * rb);
      const float* wt = wl + t * (CR * CO) + 4 * og;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(wt + (4 * rb + r) * CO);
          const f32x4 sv = {v[rb][r], v[rb][r], v[rb][r], v[rb][r]};
          acc = __builtin_elementwise_fma(sv, wv, acc);
        }
    }
    const int64_t oo = (int64_t)p * a.out_ld + 4 * og;
    f32x4 res = acc;
    if (infer) {      // uniform
      f32x4 sk = {0.f, 0.f, 0.f, 0.f};
      if (bb.skip) sk = Act<T>::gld4(reinterpret_cast<const T*>(bb.skip) + (int64_t)p * bb.skip_ld + 4 * og);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float z = ga[j] * ((acc[j] - mu[j]) * is[j]) + be[j];
        if (bb.skip) z += sk[j];
        res[j] = (bb.act == SSMI_ACT_NONE || z > 0.f) ? z : al[j] * z;
      }
    } else if (add) {
      const f32x4 av = Act<T>::gld4(add + oo);
#pragma unroll
      for (int j = 0; j < 4; ++j) res[j] = acc[j] + av[j];
    }
    Act<T>::gst4(out + oo, res);
    if (part) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { s1[j] += acc[j]; s2[j] += acc[j] * acc[j]; }
    }
    if (bb.x) {
      const f32x4 xr = Act<T>::gld4(reinterpret_cast<const T*>(bb.x) + (int64_t)p * bb.ld + 4 * og);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float d = res[j];
        if (bb.drop) d *= bb.drop[(p / bb.hw) * (unsigned)CO + (unsigned)(4 * og + j)];
        const float xh = (xr[j] - mu[j]) * is[j];
        const float z = ga[j] * xh + be[j];
        const float gr = bb.act == SSMI_ACT_NONE ? 1.f : (z > 0.f ? 1.f : (z < 0.f ? al[j] : 0.f));
        const float dz = d * gr;
        s1[j] += dz;
        s2[j] += dz * xh;
        s3[j] += d * (z < 0.f ? z : 0.f);
      }
    }
  }
  if (part || bb.x) {      // uniform: the lanes that share an output-channel group, then the four waves through LDS
    const int ns = bb.x ? 3 : 2;
    float* dst = bb.x ? bb.part : part;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v1 = s1[j], v2 = s2[j], v3 = s3[j];
#pragma unroll
      for (int o = OG; o < 64; o <<= 1) {
        v1 += __shfl_xor(v1, o, 64);
        v2 += __shfl_xor(v2, o, 64);
        v3 += __shfl_xor(v3, o, 64);
      }
      if (lane < OG) {
        red[(wave * 3 + 0) * CO + 4 * og + j] = v1;
        red[(wave * 3 + 1) * CO + 4 * og + j] = v2;
        red[(wave * 3 + 2) * CO + 4 * og + j] = v3;
      }
    }
    __syncthreads();
    if (tid < CO) {
      for (int k = 0; k < ns; ++k) {
        float sk = 0.f;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) sk += red[(wv * 3 + k) * CO + tid];
        dst[((int64_t)blockIdx.x * ns + k) * CO + tid] = sk;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Three input channels into at most sixteen outputs (the ENets' first layer: 3x3 / stride 2, 3 -> 13, written into the first
// 13 channels of the 16-channel concatenation buffer): a thread owns one output pixel, reads the 27 input values with scalar
// loads (RGB pixels are 12 / 6 bytes: nothing wider is aligned) and multiplies each by its 16 zero-padded kernel words, which
// all lanes read from one LDS address.  Forward with bias only (the layer has no fused epilogue in either phase).
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void conv_c3_kernel(Geo g, Sizes sz, const T* __restrict__ in, const float* __restrict__ w,
                                                      const float* __restrict__ bias, T* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float wl[kC4MaxTaps * 3 * 16 + 16];      // [tap][ci][16], then the bias row
  const int tid = threadIdx.x;
  const int ntaps = g.kh * g.kw;
  for (int e = tid; e < ntaps * 3 * 16 + 16; e += 256) {
    const int row = e >> 4, o = e & 15;
    float v = 0.f;
    if (o < g.cout) v = row < ntaps * 3 ? w[row * g.cout + o] : (bias ? bias[o] : 0.f);
    wl[e] = v;
  }
  __syncthreads();
  const unsigned M = (unsigned)((int64_t)g.n * g.ho * g.wo);
  const unsigned p = blockIdx.x * 256u + (unsigned)tid;
  if (p >= M) return;
  const unsigned t1 = p / (unsigned)g.wo;
  const int ox = (int)(p - t1 * (unsigned)g.wo);
  const unsigned n_ = t1 / (unsigned)g.ho;
  const int oy = (int)(t1 - n_ * (unsigned)g.ho);
  const __amdgpu_buffer_rsrc_t rin = Act<T>::rsrc(in, sz.x_bytes);
  f32x4 acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) acc[q] = *reinterpret_cast<const f32x4*>(wl + ntaps * 48 + 4 * q);
  for (int t = 0; t < ntaps; ++t) {
    const int ky = t / g.kw, kx = t - ky * g.kw;
    const int iy = oy * g.stride + ky * g.dil - g.pad_t, ix = ox * g.stride + kx * g.dil - g.pad_l;
    const bool ok = (unsigned)iy < (unsigned)g.h && (unsigned)ix < (unsigned)g.w;
    const unsigned off = (n_ * (unsigned)(g.h * g.w) + (unsigned)(iy * g.w + ix)) * (unsigned)g.x_ld * 4u;
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = Act<T>::ld(rin, ok ? off + 4u * c : kOOB);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const f32x4 sv = {v[c], v[c], v[c], v[c]};
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = __builtin_elementwise_fma(sv, *reinterpret_cast<const f32x4*>(wl + (t * 3 + c) * 16 + 4 * q), acc[q]);
    }
  }
  T* op = out + (int64_t)p * g.y_ld;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (4 * q + 4 <= g.cout) {
      Act<T>::gst4(op + 4 * q, acc[q]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (4 * q + j < g.cout) Act<T>::gst(op + 4 * q + j, acc[q][j]);
    }
  }
}

// conv_regw's pixel operand of one 16-channel block: float32 values, or (BF) the four bfloat16 as loaded
template <typename T, bool BF> struct RegwOperand {
  typedef f32x4 type;
  f32x4 v;
  static __device__ __forceinline__ f32x4 ld(__amdgpu_buffer_rsrc_t r, unsigned off4) { return Act<T>::ld4(r, off4); }
};
template <> struct RegwOperand<bf16, true> {
  typedef u32x2 type;
  u32x2 v;
  static __device__ __forceinline__ u32x2 ld(__amdgpu_buffer_rsrc_t r, unsigned off4) { return Act<bf16>::ld4r(r, off4); }
};

// ---------------------------------------------------------------------------------------------------
// Register-weight implicit GEMM for narrow outputs (<= 16 output channels, stride 1, short reductions: K = NTAP * CR <=
// 144): the same GEMM as conv_gemm_lds with 16 x 16 x 4 MFMA tiles, but the wave's share of the kernel matrix -
// K / 4 words per lane - is loaded ONCE into registers straight from global memory (the matrix is a few KB and sits in
// L2), so there is no LDS staging, no barrier in front of the first multiply and no LDS read per MFMA.  The reduction is
// unrolled completely (taps and channel blocks are compile-time constants, tap offsets wave-uniform scalars).  A wave
// walks over several pixel groups; the operand loads of the next group are issued before the current group is
// multiplied, so one memory round trip is hidden behind a whole group's MFMAs and stores.
//   forward (mode 0): in = x, CR = cin;  data gradient (mode 1, stride 1): in = dy, CR = cout, flipped taps.
// Epilogue as conv_gemm_lds: bias, addend, fused BatchNormalization statistics (part) or BN backward sums (bb).
// ---------------------------------------------------------------------------------------------------
// BF (bfloat16 tensors, forward form, the layer's kernel pack [tap][cout][cin] in a.wpack): one v_mfma_f32_16x16x16_bf16 per
// tap and 16-channel block - lane (c, kq) feeds the four channels 4 * kq .. 4 * kq + 3 of its pixel exactly as its 8-byte load
// delivered them, and the same four channels of output channel c's pack row (one 8-byte load per block, once per wave).
// Half the operand registers of the float32 route, a quarter of its matrix cycles, no widening.
template <typename T, int NTAP, int CR, int MT, bool BF = false>
__global__ __launch_bounds__(256) void conv_regw(Geo g, Sizes sz, GemmArgs a, const T* __restrict__ in,
                                                 const float* __restrict__ w, const float* __restrict__ bias,
                                                 const T* __restrict__ add, T* __restrict__ out,
                                                 float* __restrict__ part, BnBwdArgs bb) {
  __shared__ float red[4 * 3 * 16];
  constexpr int CB = CR / 16;              // 16-channel k-blocks per tap
  constexpr int NB = NTAP * CB;            // k-blocks in all
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, kq = lane >> 4;
  const bool bwd = a.mode == 1;
  const int qh = bwd ? g.h : g.ho, qw = bwd ? g.w : g.wo;      // stride 1: the output grid
  const int64_t M = (int64_t)g.n * qh * qw;
  const bool cv = c < a.C_o;

  // tap offsets (uniform) and the lane's kernel words: B[(t * CB + cb) * 4 + j] = Wl[t][cb * 16 + 4 * kq + j][c]
  int tdy[NTAP], tdx[NTAP];
  unsigned toff[NTAP];
  float B[BF ? 1 : NB * 4];
  u32x2 Bp[BF ? NB : 1];
  {
    const __amdgpu_buffer_rsrc_t rwt = BF ? make_rsrc(a.wpack, (unsigned)(NTAP * CR * a.C_o) * 2u) : make_rsrc(w, sz.w_bytes);
    const unsigned cmask = cv ? 0u : kOOB;         // kernel matrices are far below 2 GiB: OR-ing the marker in is exact
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
      const int ky = NTAP == 1 ? 0 : t / g.kw, kx = NTAP == 1 ? 0 : t - (t / g.kw) * g.kw;
      tdy[t] = bwd ? g.pad_t - ky * g.dil : ky * g.dil - g.pad_t;
      tdx[t] = bwd ? g.pad_l - kx * g.dil : kx * g.dil - g.pad_l;
      toff[t] = (unsigned)((tdy[t] * a.in_w + tdx[t]) * a.in_ld * 4);
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        const int r0 = cb * 16 + 4 * kq;
        if constexpr (BF) {
          // pack[t][co = c][ci = r0 .. r0 + 3]: four bfloat16, one 8-byte load
          Bp[t * CB + cb] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rwt, ((unsigned)((t * a.C_o + c) * CR + r0) * 2u) | cmask, 0, 0));
        } else if (bwd) {
          // data gradient: w[t][ci = c][co = r0 .. r0 + 3] is one 16-byte load (cout = CR is a multiple of 16)
          const f32x4 v = bload4(rwt, ((unsigned)((t * g.cin + c) * g.cout + r0) * 4u) | cmask);
#pragma unroll
          for (int j = 0; j < 4; ++j) B[(t * CB + cb) * 4 + j] = v[j];
        } else {
          // forward: w[t][ci = r0 + j][co = c], the 16 lanes of a row read 64 consecutive bytes
#pragma unroll
          for (int j = 0; j < 4; ++j)
            B[(t * CB + cb) * 4 + j] = bload(rwt, ((unsigned)((t * g.cin + r0 + j) * g.cout + c) * 4u) | cmask);
        }
      }
    }
  }
  const __amdgpu_buffer_rsrc_t rin = Act<T>::rsrc(in, bwd ? sz.y_bytes : sz.x_bytes);
  const unsigned out_bytes = bwd ? sz.x_bytes : sz.y_bytes;
  const __amdgpu_buffer_rsrc_t rout = Act<T>::rsrc(out, out_bytes);
  const __amdgpu_buffer_rsrc_t radd = Act<T>::rsrc(add ? add : out, add ? out_bytes : 0u);
  const __amdgpu_buffer_rsrc_t rbx = Act<T>::rsrc(bb.x ? bb.x : (const void*)out, bb.x ? (unsigned)((int64_t)g.n * g.h * g.w * bb.ld * 4) : 0u);
  const float bvv = (bias && cv) ? bias[c] : 0.f;
  const int64_t ngroups = (M + MT * 16 - 1) / (MT * 16);
  float st1 = 0.f, st2 = 0.f, st3 = 0.f;

  typename RegwOperand<T, BF>::type An[MT][NB];      // operands of the group fetched ahead
  auto fetch = [&](int64_t grp) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const unsigned p = (unsigned)(grp * (MT * 16)) + (unsigned)(m * 16 + c);
      const bool pv = grp < ngroups && p < (unsigned)M;
      const unsigned pp = pv ? p : 0u;
      const unsigned t1 = pp / (unsigned)qw;
      const int qx = (int)(pp - t1 * (unsigned)qw);
      const unsigned n_ = t1 / (unsigned)qh;
      const int qy = (int)(t1 - n_ * (unsigned)qh);
      const unsigned pbase = ((n_ * (unsigned)a.in_h + (unsigned)qy) * (unsigned)a.in_w + (unsigned)qx) * (unsigned)a.in_ld * 4u +
                             (unsigned)(kq * 16);
#pragma unroll
      for (int t = 0; t < NTAP; ++t) {
        bool ok = pv;
        if (NTAP > 1) ok = ok & ((unsigned)(qy + tdy[t]) < (unsigned)a.in_h) & ((unsigned)(qx + tdx[t]) < (unsigned)a.in_w);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) An[m][t * CB + cb] = RegwOperand<T, BF>::ld(rin, ok ? pbase + toff[t] + (unsigned)(cb * 64) : kOOB);
      }
    }
  };

  int64_t grp = (int64_t)xcd_tile(blockIdx.x, gridDim.x, sz.xcd & 4) * 4 + wave;
  const int64_t gstep = (int64_t)gridDim.x * 4;
  fetch(grp);
  for (; grp < ngroups; grp += gstep) {
    RegwOperand<T, BF> A[MT][NB];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int b = 0; b < NB; ++b) A[m][b].v = An[m][b];
    fetch(grp + gstep);             // past the end every lane reads zeros
    typedef float accv __attribute__((ext_vector_type(4)));
    accv acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = accv{0.f, 0.f, 0.f, 0.f};
    if constexpr (BF) {
      typedef short s16x4 __attribute__((ext_vector_type(4)));
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int m = 0; m < MT; ++m)
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, A[m][b].v), __builtin_bit_cast(s16x4, Bp[b]), acc[m], 0, 0, 0);
    } else {
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[m][b].v[j], B[b * 4 + j], acc[m], 0, 0, 0);
    }

    // ---- epilogue: accumulator element i of tile m is output pixel tile0 + m * 16 + 4 * kq + i, column c ----
    const unsigned tile0 = (unsigned)(grp * (MT * 16));
    unsigned po[MT][4], rowoff[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        po[m][i] = tile0 + (unsigned)(m * 16 + kq * 4 + i);
        rowoff[m][i] = (po[m][i] < (unsigned)M && cv) ? (po[m][i] * (unsigned)a.out_ld + (unsigned)c) * 4u : kOOB;
      }
    // branch-free: without an addend / a fused BN backward the descriptors have size zero and the loads return 0
    float av[MT][4], xr[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        av[m][i] = Act<T>::ld(radd, rowoff[m][i]);
        xr[m][i] = Act<T>::ld(rbx, rowoff[m][i] != kOOB ? (po[m][i] * (unsigned)bb.ld + (unsigned)c) * 4u : kOOB);
      }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) Act<T>::st(rout, rowoff[m][i], acc[m][i] + bvv + av[m][i]);
    if (part) {          // uniform: BatchNormalization statistics of the forward output
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float v = rowoff[m][i] != kOOB ? acc[m][i] + bvv : 0.f;
          st1 += v;
          st2 += v * v;
        }
    }
    if (bb.x) {          // uniform: BatchNormalization backward sums of the data gradient
      const int cc = cv ? c : 0;
      const float mu = bb.mean[cc], is = bb.invstd[cc], ga = bb.gamma[cc], be = bb.beta[cc];
      const float al = bb.act == SSMI_ACT_PRELU ? bb.alpha[cc] : bb.slope;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bool ok = rowoff[m][i] != kOOB;
          float d = acc[m][i] + bvv + av[m][i];
          if (bb.drop) d *= bb.drop[(ok ? po[m][i] / bb.hw : 0u) * (unsigned)a.C_o + (unsigned)cc];
          const float xh = (xr[m][i] - mu) * is;
          const float z = ga * xh + be;
          const float gr = bb.act == SSMI_ACT_NONE ? 1.f : (z > 0.f ? 1.f : (z < 0.f ? al : 0.f));
          const float dz = ok ? d * gr : 0.f;
          st1 += dz;
          st2 += dz * xh;
          st3 += ok ? d * (z < 0.f ? z : 0.f) : 0.f;
        }
    }
  }
  if (part || bb.x) {   // uniform: combine the four lane groups that share a column, then the four waves through LDS
    const int ns = bb.x ? 3 : 2;
    float* dst = bb.x ? bb.part : part;
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) {
      st1 += __shfl_xor(st1, o, 64);
      st2 += __shfl_xor(st2, o, 64);
      st3 += __shfl_xor(st3, o, 64);
    }
    if (kq == 0) {
      red[(wave * 3 + 0) * 16 + c] = st1;
      red[(wave * 3 + 1) * 16 + c] = st2;
      red[(wave * 3 + 2) * 16 + c] = st3;
    }
    __syncthreads();
    if (tid < 16 && tid < a.C_o) {
      for (int k = 0; k < ns; ++k) {
        float sk = 0.f;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) sk += red[(wv * 3 + k) * 16 + tid];
        dst[((int64_t)blockIdx.x * ns + k) * a.C_o + tid] = sk;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// weight gradient: dw[(tap,ci), co] = sum_p x[pix(p,tap), ci] * dy[p, co]; optional extra row K = sum_p dy[p, co].
// The [K_aug x cout] result is cut into 32x32 output tiles (kt, ct).  A workgroup owns one strip of output rows and
// up to 4*TPW tiles: its waves split the tiles (wave w takes tiles w, w+wps, ...), so the 4 waves stream the same
// pixels (dy and the shifted x rows come from L1/L2 after the first touch) while each keeps TPW accumulators.
// When there are fewer than 4 tiles, the waves of a workgroup take different strips instead.
// Partial sums go to part[strip][K_aug][cout]; conv_bwd_weight_reduce adds the strips in a fixed order.
// ---------------------------------------------------------------------------------------------------
struct WgArgs {
  int K, K_aug, ktiles, cotiles, T;   // T = ktiles * cotiles
  int wps;                            // waves per strip (min(4, T))
  int nstrips, rows_per_strip;
};

template <typename T, int TPW>
__global__ __launch_bounds__(256) void conv_bwd_weight_kernel(Geo g, Sizes sz, WgArgs a, const T* __restrict__ x,
                                                              const T* __restrict__ dy, float* __restrict__ part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int spb = 4 / a.wps;
  const int strip = blockIdx.x * spb + wave / a.wps;
  const int widx = wave % a.wps;
  if (strip >= a.nstrips || wave >= spb * a.wps) return;
  const int tile0 = blockIdx.y * (a.wps * TPW) + widx;
  const __amdgpu_buffer_rsrc_t rx = Act<T>::rsrc(x, sz.x_bytes), ry = Act<T>::rsrc(dy, sz.y_bytes);

  bool tv[TPW], kv[TPW], kb[TPW], colv[TPW];
  int kyd[TPW], kxd[TPW], kt_[TPW], col[TPW];
  unsigned cib[TPW], colb[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int t = tile0 + a.wps * j;
    tv[j] = t < a.T;
    const int ts = tv[j] ? t : 0;
    const int kt = ts / a.cotiles, ct = ts % a.cotiles;
    kt_[j] = kt;
    const int k = kt * 32 + r;
    kv[j] = tv[j] && k < a.K;
    kb[j] = tv[j] && (k == a.K) && (a.K_aug > a.K);
    const int ks = kv[j] ? k : 0;
    const int tap = ks / g.cin;
    cib[j] = (unsigned)(ks % g.cin) * 4u;
    kyd[j] = (tap / g.kw) * g.dil - g.pad_t;
    kxd[j] = (tap % g.kw) * g.dil - g.pad_l;
    col[j] = ct * 32 + r;
    colv[j] = tv[j] && col[j] < g.cout;
    colb[j] = colv[j] ? (unsigned)col[j] * 4u : kOOB;
  }
  f32x16 acc[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

  const int R = g.n * g.ho;
  const int row0 = strip * a.rows_per_strip;
  const int row1 = min(R, row0 + a.rows_per_strip);
  // One dword per lane per operand per MFMA makes the pixel loop latency bound, so operands are fetched in batches of
  // U MFMA steps, two batches in flight: while batch b feeds the matrix core, batch b+1 is already on its way.
  constexpr int U = (TPW == 1) ? 16 : 8;
  const int spr = (g.wo + 1) / 2;          // MFMA steps per output row (2 pixels per step)
  const int bpr = (spr + U - 1) / U;       // batches per row
  const int nbatch = (row1 - row0) * bpr;
  float a0[U][TPW], b0[U][TPW], a1[U][TPW], b1[U][TPW];
  const unsigned xpix = (unsigned)g.x_ld * 4u, ypix = (unsigned)g.y_ld * 4u;

  auto load_batch = [&](int bi, float (&av)[U][TPW], float (&bv)[U][TPW]) {
    const bool bvld = bi < nbatch;
    const int bis = bvld ? bi : 0;
    const int row = row0 + bis / bpr;
    const int step0 = (bis % bpr) * U;
    const int n_ = row / g.ho, oy = row % g.ho;
    const unsigned dyrow = (unsigned)(n_ * g.ho + oy) * (unsigned)g.wo * ypix;
    unsigned xrow[TPW];
    bool yv[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      const int iy = oy * g.stride + kyd[j];
      yv[j] = bvld && kv[j] && (unsigned)iy < (unsigned)g.h;
      xrow[j] = (unsigned)(n_ * g.h + (yv[j] ? iy : 0)) * (unsigned)g.w * xpix + cib[j];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ox = (step0 + u) * 2 + hh;
      const bool ov = bvld && ox < g.wo;
#pragma unroll
      for (int j = 0; j < TPW; ++j) {
        const int ix = ox * g.stride + kxd[j];
        const bool xv = yv[j] && ov && (unsigned)ix < (unsigned)g.w;
        const float t = Act<T>::ld(rx, xv ? xrow[j] + (unsigned)ix * xpix : kOOB);
        av[u][j] = kb[j] ? (ov ? 1.f : 0.f) : t;
        bv[u][j] = Act<T>::ld(ry, ov ? dyrow + (unsigned)ox * ypix + colb[j] : kOOB);
      }
    }
  };
  auto mma_batch = [&](float (&av)[U][TPW], float (&bv)[U][TPW]) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < TPW; ++j) acc[j] = MFMA32(av[u][j], bv[u][j], acc[j]);
  };

  load_batch(0, a0, b0);
  for (int bi = 0; bi < nbatch; bi += 2) {
    load_batch(bi + 1, a1, b1);
    mma_batch(a0, b0);
    load_batch(bi + 2, a0, b0);
    mma_batch(a1, b1);
  }

  float* pbase = part + (int64_t)strip * a.K_aug * g.cout;
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    if (!colv[j]) continue;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int kk = kt_[j] * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
      if (kk < a.K_aug) pbase[(int64_t)kk * g.cout + col[j]] = acc[j][i];
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// weight gradient, LDS-staged variant (used whenever the staging tiles fit): a workgroup walks its strip of output
// rows in chunks of P output pixels.  Per chunk all 256 threads fetch the dy chunk [P][cout] and the x patch
// [kh][Wp][cin] that the chunk's taps touch with 16-byte buffer loads (zero padding comes from the bounds check), park
// them in registers while the previous chunk is being multiplied, then drop them into LDS.  The MFMA operands are
// plain ds_read_b32: lane (r, hh) of tile (kt, ct) reads x-patch word [tap(k), pixel 2*step+hh, ci(k)] and dy word
// [pixel 2*step+hh][ct*32 + r].  Rows k >= K read a zero word, the bias row reads a ones/zeros vector.
// ---------------------------------------------------------------------------------------------------
struct WgLds {
  int P, Wp;            // output pixels per chunk, x-patch width in input pixels
  int xv_n, yv_n;       // 16-byte (VEC) or 4-byte elements per thread for the x patch / dy chunk
  int x_elems, y_elems; // number of such elements
  int coutp;            // dy chunk row length in LDS (cotiles * 32)
  int ones_off, zero_off, yl_off;   // word offsets inside the LDS block
  int lds_words;
};
static constexpr int kXvMax = 12, kYvMax = 8;

// M16: 16x16 output tiles on v_mfma_f32_16x16x4_f32 (4 pixels per step) for cout <= 16, where 32-wide tiles would
// spend half or more of the matrix core on padding columns.
// XV / YV: staging slots per thread that this instantiation carries (>= xv_n / yv_n of the launch).  The slots cost
// registers whether a layer fills them or not, and these launches share the SIMDs with the data-gradient chain of the
// main stream, so the launcher picks the smallest class that fits the layer.
// PRE (float32, VEC, a 1x1 layer whose channel count divides 1024): the x operand is act(BN(x)) of the raw tensor,
// applied when a staged element is committed to LDS.  Every slot of a thread then holds the same four channels
// (256 % (cin / 4) == 0), so the thread keeps their parameters in registers.
template <typename T, int TPW, bool VEC, bool M16, int XV, int YV, bool PRE = false>
__global__ __launch_bounds__(256) void conv_bwd_weight_lds(Geo g, Sizes sz, WgArgs a, WgLds l, const T* __restrict__ x,
                                                           const T* __restrict__ dy, float* __restrict__ part, PreBn pb = PreBn{}) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  f32x4 pm = {0.f, 0.f, 0.f, 0.f}, pis = pm, pga = pm, pbe = pm, pal = pm;
  bool pact = false;
  if constexpr (PRE) {
    const int c = (tid % (g.cin / 4)) * 4;
    pm = *reinterpret_cast<const f32x4*>(pb.mean + c); pis = *reinterpret_cast<const f32x4*>(pb.invstd + c);
    pga = *reinterpret_cast<const f32x4*>(pb.gamma + c); pbe = *reinterpret_cast<const f32x4*>(pb.beta + c);
    pact = pb.alpha != nullptr;
    if (pact) pal = *reinterpret_cast<const f32x4*>(pb.alpha + c);
  }
  const int lane = tid & 63, wave = tid >> 6;
  constexpr int TS = M16 ? 16 : 32;      // tile edge
  constexpr int KQ = 64 / TS;            // pixels consumed per MFMA step
  constexpr int NR = M16 ? 4 : 16;       // accumulator registers per tile
  const int r = lane & (TS - 1), hh = lane / TS;
  const int strip = (int)xcd_tile(blockIdx.x, gridDim.x, sz.xcd & 8);      // neighbouring strips share the halo rows of their patches
  const int tile0 = blockIdx.y * (a.wps * TPW) + wave;
  const __amdgpu_buffer_rsrc_t rx = Act<T>::rsrc(x, sz.x_bytes), ry = Act<T>::rsrc(dy, sz.y_bytes);
  constexpr int EW = VEC ? 4 : 1;   // floats per staged element

  // ---- per-lane MFMA operand addressing (fixed for the whole kernel) ----
  bool colv[TPW];
  int kt_[TPW], col[TPW], abase[TPW], astr[TPW];
  bool has_tile = false;
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int t = tile0 + a.wps * j;
    const bool tv = (wave < a.wps) && t < a.T;
    has_tile |= tv;
    const int ts = tv ? t : 0;
    const int kt = ts / a.cotiles, ct = ts % a.cotiles;
    kt_[j] = kt;
    const int k = kt * TS + r;
    const bool kv = tv && k < a.K;
    const bool kb = tv && (k == a.K) && (a.K_aug > a.K);
    const int ks = kv ? k : 0;
    const int tap = ks / g.cin, ci = ks % g.cin;
    const int ky = tap / g.kw, kx = tap % g.kw;
    abase[j] = kv ? ((ky * l.Wp + kx * g.dil + hh * g.stride) * g.cin + ci) : (kb ? l.ones_off + hh : l.zero_off);
    astr[j] = kv ? KQ * g.stride * g.cin : (kb ? KQ : 0);
    col[j] = ct * TS + r;
    colv[j] = tv && col[j] < g.cout;
  }
  typedef float accv __attribute__((ext_vector_type(NR)));
  accv acc[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j)
#pragma unroll
    for (int i = 0; i < NR; ++i) acc[j][i] = 0.f;

  // ---- per-thread staging maps (independent of the chunk), one packed word per slot ----
  // x patch element e -> (ky, colp, c): LDS word = e*EW; global = row(ky) + colp*x_ld + c.
  // xm = [31] unused | [30:23] ky*dil | [22:12] colp | [11:0] c     ym = [31] unused | [19:12] pixel | [11:0] c
  unsigned xm[XV], ym[YV];
  const int cin_e = g.cin / EW;   // elements per pixel (VEC requires cin % 4 == 0)
#pragma unroll
  for (int i = 0; i < XV; ++i) {
    xm[i] = 0x80000000u;
    if (i < l.xv_n) {        // uniform: the unused slots cost no divisions
      const unsigned e = (unsigned)(tid + i * 256);
      if (e < (unsigned)l.x_elems) {
        const unsigned t = e / (unsigned)cin_e;
        const unsigned ky = t / (unsigned)l.Wp;
        xm[i] = ((ky * (unsigned)g.dil) << 23) | ((t - ky * (unsigned)l.Wp) << 12) | ((e - t * (unsigned)cin_e) * (unsigned)EW);
      }
    }
  }
  const int cout_e = (g.cout + EW - 1) / EW;
#pragma unroll
  for (int i = 0; i < YV; ++i) {
    ym[i] = 0x80000000u;
    if (i < l.yv_n) {
      const unsigned e = (unsigned)(tid + i * 256);
      if (e < (unsigned)l.y_elems) {
        const unsigned q = e / (unsigned)cout_e;
        ym[i] = (q << 12) | ((e - q * (unsigned)cout_e) * (unsigned)EW);
      }
    }
  }
  auto x_kyd = [&](int i) { return (int)((xm[i] >> 23) & 0xFFu); };
  auto x_colp = [&](int i) { return (int)((xm[i] >> 12) & 0x7FFu); };
  auto x_rel = [&](int i) { return (xm[i] & 0xFFFu) * 4u; };       // byte offset of the channel inside the pixel
  auto x_ok = [&](int i) { return (int)xm[i] >= 0; };
  auto y_p = [&](int i) { return (int)((ym[i] >> 12) & 0xFFu); };
  auto y_rel = [&](int i) { return (ym[i] & 0xFFFu) * 4u; };
  auto y_ok = [&](int i) { return (int)ym[i] >= 0; };
  // zero the whole block once (pad columns of the dy chunk, the zero word)
  for (int i = tid; i < l.lds_words; i += 256) smem[i] = 0.f;

  const int R = g.n * g.ho;
  const int row0 = strip * a.rows_per_strip;
  const int row1 = min(R, row0 + a.rows_per_strip);
  const int cpr = (g.wo + l.P - 1) / l.P;
  const int nchunks = (row1 - row0) * cpr;
  const unsigned xpix = (unsigned)g.x_ld * 4u, ypix = (unsigned)g.y_ld * 4u;

  // staged operands stay in their storage form while in flight (bf16: half the registers, and no conversion that would
  // make the wave wait for the loads right here); commit() converts on the way into LDS
  typename Act<T>::raw4 xr[VEC ? XV : 1], yr[VEC ? YV : 1];
  float xs[VEC ? 1 : XV], ys[VEC ? 1 : YV];            // the scalar path (not VEC)
  auto prefetch = [&](int ch) {
    const bool cv = ch < nchunks;
    const int chs = cv ? ch : 0;
    const int row = row0 + chs / cpr;
    const int ox0 = (chs % cpr) * l.P;
    const int n_ = row / g.ho, oy = row % g.ho;
    const int iy0 = oy * g.stride - g.pad_t, ix0 = ox0 * g.stride - g.pad_l;
    const int pv = min(l.P, g.wo - ox0);
    if constexpr (sizeof(T) == 2) {
      // bf16: every offset first, then the loads back to back, and no branch around a load (the unused slots read
      // nothing).  Interleaved and behind uniform branches, this instantiation reused a pending load's destination
      // register for the next address and drained the memory counter in front of every load, so the loads of a chunk
      // went out one round trip at a time.
      unsigned xo[XV], yo[YV];
#pragma unroll
      for (int i = 0; i < XV; ++i) {
        const int iy = iy0 + x_kyd(i), ix = ix0 + x_colp(i);
        const bool ok = cv & x_ok(i) & ((unsigned)iy < (unsigned)g.h) & ((unsigned)ix < (unsigned)g.w);
        const unsigned off = (unsigned)((n_ * g.h + iy) * g.w + ix) * xpix + x_rel(i);
        xo[i] = ok ? off : kOOB;
      }
#pragma unroll
      for (int i = 0; i < YV; ++i) {
        const bool ok = cv & y_ok(i) & (y_p(i) < pv) & (VEC || (y_rel(i) >> 2) < (unsigned)g.cout);
        const unsigned off = (unsigned)((n_ * g.ho + oy) * g.wo + ox0 + y_p(i)) * ypix + y_rel(i);
        yo[i] = ok ? off : kOOB;
      }
#pragma unroll
      for (int i = 0; i < XV; ++i) {
        if constexpr (VEC) xr[i] = Act<T>::ld4r(rx, xo[i]);
        else xs[i] = Act<T>::ld(rx, xo[i]);
      }
#pragma unroll
      for (int i = 0; i < YV; ++i) {
        if constexpr (VEC) yr[i] = Act<T>::ld4r(ry, yo[i]);
        else ys[i] = Act<T>::ld(ry, yo[i]);
      }
    } else {
      // fp32: the loads already leave back to back; skipping the unused slots measures faster than issuing them
#pragma unroll
      for (int i = 0; i < XV; ++i) {
        if (i < l.xv_n) {
          const int iy = iy0 + x_kyd(i), ix = ix0 + x_colp(i);
          const bool ok = cv && x_ok(i) && (unsigned)iy < (unsigned)g.h && (unsigned)ix < (unsigned)g.w;
          const unsigned off = ok ? (unsigned)((n_ * g.h + iy) * g.w + ix) * xpix + x_rel(i) : kOOB;
          if constexpr (VEC) xr[i] = Act<T>::ld4r(rx, off);
          else xs[i] = Act<T>::ld(rx, off);
        }
      }
#pragma unroll
      for (int i = 0; i < YV; ++i) {
        if (i < l.yv_n) {
          const bool ok = cv && y_ok(i) && y_p(i) < pv && (VEC || (y_rel(i) >> 2) < (unsigned)g.cout);
          const unsigned off = ok ? (unsigned)((n_ * g.ho + oy) * g.wo + ox0 + y_p(i)) * ypix + y_rel(i) : kOOB;
          if constexpr (VEC) yr[i] = Act<T>::ld4r(ry, off);
          else ys[i] = Act<T>::ld(ry, off);
        }
      }
    }
  };
  auto commit = [&](int ch) {
    const int ox0 = (ch % cpr) * l.P;
    const int pv = min(l.P, g.wo - ox0);
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      if (i < l.xv_n) {
        const int e = tid + i * 256;
        if (e < l.x_elems) {
          if constexpr (VEC) {
            f32x4 v = Act<T>::up4(xr[i]);
            if constexpr (PRE) {
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = pre_bn_act(v[j], pm[j], pis[j], pga[j], pbe[j], pal[j], pact);
            }
            *reinterpret_cast<f32x4*>(smem + e * 4) = v;
          } else smem[e] = xs[i];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < YV; ++i) {
      if (i < l.yv_n) {
        const int e = tid + i * 256;
        if (e < l.y_elems) {
          float* dst = smem + l.yl_off + y_p(i) * l.coutp + (y_rel(i) >> 2);
          if constexpr (VEC) *reinterpret_cast<f32x4*>(dst) = Act<T>::up4(yr[i]);
          else *dst = ys[i];
        }
      }
    }
    if (tid < l.P) smem[l.ones_off + tid] = tid < pv ? 1.f : 0.f;
    return pv;
  };

  prefetch(0);
  for (int ch = 0; ch < nchunks; ++ch) {
    __syncthreads();            // everyone is done reading the previous chunk
    const int pv = commit(ch);
    __syncthreads();
    prefetch(ch + 1);           // in flight while this chunk is multiplied
    if (has_tile) {
      const int steps = (pv + KQ - 1) / KQ;
      const float* yl = smem + l.yl_off + hh * l.coutp;
      // four steps per round, all eight LDS reads of a round issued before its first MFMA (left to `#pragma unroll`
      // the compiler kept one step per iteration: two reads, a full LDS wait, one MFMA)
      auto run = [&](int st, auto u) {
        constexpr int U = decltype(u)::value;
        float av[U][TPW], bv[U][TPW];
#pragma unroll
        for (int q = 0; q < U; ++q)
#pragma unroll
          for (int j = 0; j < TPW; ++j) {
            av[q][j] = smem[abase[j] + (st + q) * astr[j]];
            bv[q][j] = yl[(st + q) * KQ * l.coutp + col[j]];
          }
#pragma unroll
        for (int q = 0; q < U; ++q)
#pragma unroll
          for (int j = 0; j < TPW; ++j) {
            if constexpr (M16) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q][j], bv[q][j], acc[j], 0, 0, 0);
            else acc[j] = MFMA32(av[q][j], bv[q][j], acc[j]);
          }
      };
      if constexpr (sizeof(T) == 2) {
        constexpr int U = TPW == 1 ? 4 : 2;
        int st = 0;
        for (; st + U <= steps; st += U) run(st, std::integral_constant<int, U>{});
        for (; st < steps; ++st) run(st, std::integral_constant<int, 1>{});
      } else {
        // fp32 measures faster one step at a time: the unrolled body costs it 20-30 registers
#pragma unroll 4
        for (int st = 0; st < steps; ++st) run(st, std::integral_constant<int, 1>{});
      }
    }
  }

  float* pbase = part + (int64_t)strip * a.K_aug * g.cout;
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    if (!colv[j]) continue;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int kk = M16 ? kt_[j] * 16 + hh * 4 + i : kt_[j] * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
      if (kk < a.K_aug) pbase[(int64_t)kk * g.cout + col[j]] = acc[j][i];
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// weight gradient of bf16 tensors on the bf16 matrix instructions (v_mfma_f32_32x32x16_bf16, pixels are the reduction
// dimension, 16 per instruction).  Same strips, chunks and register staging as conv_bwd_weight_lds, but the LDS images
// keep the tensors' own form - rows of pixels, bf16 channels - and the operands are fetched with the transposing read
// (ds_read_b64_tr_b16): a group of 16 lanes reads a block of 4 pixel rows x 16 channels, lane i receives channel i of
// the four rows, i.e. four consecutive pixels of ONE channel, which is what an operand of a product summed over pixels
// needs.  Within a group lane 4q+p supplies the address of row q, channels 4p..4p+3 (8 bytes); rows and 4-channel
// chunks are free, so taps, strides and dilation only move addresses.  Rows of the kernel matrix past K read a
// constant word: {1,0,0,0} for the bias row (dy rows past the chunk are zero, so ones for every pixel are right),
// zeros otherwise; output channels past cout read zeros.  The instruction needs every lane active: no lane-dependent
// branch surrounds the reads.
// LDS (bytes): x image [kh][Wq][cin] bf16 with Wq = (Pq-1)*stride + (kw-1)*dil + 1, Pq = P rounded up to 16 (the pad
// pixels stay zero) | ones word | zero word | dy image [Pq][cout] bf16.
// ---------------------------------------------------------------------------------------------------
struct WgBf {
  int Pq, Wq;
  int ones_b, zero_b, yl_b;    // byte offsets
  int lds_bytes;
};

// PRE: as in conv_bwd_weight_lds - the x operand is act(BN(x)) of the raw tensor, formed (and rounded to bfloat16, as the
// stored tensor would have been) when a staged element is committed to LDS.
template <int TPW, int XV, int YV, bool PRE = false>
__global__ __launch_bounds__(256) void conv_bwd_weight_lds_bf16(Geo g, Sizes sz, WgArgs a, WgLds l, WgBf q, const bf16* __restrict__ x,
                                                                const bf16* __restrict__ dy, float* __restrict__ part, PreBn pb = PreBn{}) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f32x4 pm = {0.f, 0.f, 0.f, 0.f}, pis = pm, pga = pm, pbe = pm, pal = pm;
  bool pact = false;
  if constexpr (PRE) {
    const int c = (threadIdx.x % (g.cin / 4)) * 4;
    pm = *reinterpret_cast<const f32x4*>(pb.mean + c); pis = *reinterpret_cast<const f32x4*>(pb.invstd + c);
    pga = *reinterpret_cast<const f32x4*>(pb.gamma + c); pbe = *reinterpret_cast<const f32x4*>(pb.beta + c);
    pact = pb.alpha != nullptr;
    if (pact) pal = *reinterpret_cast<const f32x4*>(pb.alpha + c);
  }
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  typedef __attribute__((address_space(3))) char* lds_ptr;
  const lds_ptr lds = (lds_ptr)smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int strip = (int)xcd_tile(blockIdx.x, gridDim.x, sz.xcd & 8);
  const int tile0 = blockIdx.y * (a.wps * TPW) + wave;
  const __amdgpu_buffer_rsrc_t rx = Act<bf16>::rsrc(x, sz.x_bytes), ry = Act<bf16>::rsrc(dy, sz.y_bytes);

  // ---- per-lane addresses of the transposing reads (fixed for the whole kernel) ----
  // supplier role of this lane inside its group of 16: row sq of the 4-pixel block, 4-channel chunk sp; half gh of the tile
  const int sq = (lane & 15) >> 2, sp = lane & 3, gh = (lane >> 4) & 1;
  bool colv[TPW];
  int kt_[TPW], col[TPW];
  int abase[TPW], astep[TPW], bbase[TPW], bstep[TPW];   // bytes; step = per 16 pixels, the second read of a step sits a quarter step on
  bool has_tile = false;
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int t = tile0 + a.wps * j;
    const bool tv = (wave < a.wps) && t < a.T;
    has_tile |= tv;
    const int ts = tv ? t : 0;
    const int kt = ts / a.cotiles, ct = ts % a.cotiles;
    kt_[j] = kt;
    col[j] = ct * 32 + r;
    colv[j] = tv && col[j] < g.cout;
    const int kk = kt * 32 + 16 * gh + 4 * sp;           // first kernel-matrix row of the chunk this lane addresses
    const bool kv = tv && kk < a.K;
    const bool kb = tv && kk == a.K && a.K_aug > a.K;
    const int ks = kv ? kk : 0;
    const int tap = ks / g.cin, ci = ks % g.cin;
    const int ky = tap / g.kw, kx = tap % g.kw;
    abase[j] = kv ? ((ky * q.Wq + (8 * hh + sq) * g.stride + kx * g.dil) * g.cin + ci) * 2 : (kb ? q.ones_b : q.zero_b);
    astep[j] = kv ? 16 * g.stride * g.cin * 2 : 0;
    const int co = ct * 32 + 16 * gh + 4 * sp;
    const bool cv = tv && co < g.cout;
    bbase[j] = cv ? q.yl_b + ((8 * hh + sq) * g.cout + co) * 2 : q.zero_b;
    bstep[j] = cv ? 16 * g.cout * 2 : 0;
  }
  f32x16 acc[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

  // ---- per-thread staging maps, one packed word per slot ----
  // xm = [31] unused | [30:23] ky | [22:12] colp | [11:0] c      ym = [31] unused | [19:12] pixel | [11:0] c
  unsigned xm[XV], ym[YV];
  const int cin_e = g.cin / 4;
#pragma unroll
  for (int i = 0; i < XV; ++i) {
    xm[i] = 0x80000000u;
    if (i < l.xv_n) {
      const unsigned e = (unsigned)(tid + i * 256);
      if (e < (unsigned)l.x_elems) {
        const unsigned t = e / (unsigned)cin_e;
        const unsigned ky = t / (unsigned)l.Wp;
        xm[i] = (ky << 23) | ((t - ky * (unsigned)l.Wp) << 12) | ((e - t * (unsigned)cin_e) * 4u);
      }
    }
  }
  const int cout_e = g.cout / 4;
#pragma unroll
  for (int i = 0; i < YV; ++i) {
    ym[i] = 0x80000000u;
    if (i < l.yv_n) {
      const unsigned e = (unsigned)(tid + i * 256);
      if (e < (unsigned)l.y_elems) {
        const unsigned pq = e / (unsigned)cout_e;
        ym[i] = (pq << 12) | ((e - pq * (unsigned)cout_e) * 4u);
      }
    }
  }
  for (int i = tid; i < q.lds_bytes / 4; i += 256) smem[i] = 0.f;
  __syncthreads();
  if (tid == 0) *reinterpret_cast<unsigned*>(reinterpret_cast<char*>(smem) + q.ones_b) = 0x00003F80u;   // bf16 {1, 0}, then {0, 0}

  const int R = g.n * g.ho;
  const int row0 = strip * a.rows_per_strip;
  const int row1 = min(R, row0 + a.rows_per_strip);
  const int cpr = (g.wo + l.P - 1) / l.P;
  const int nchunks = (row1 - row0) * cpr;
  const unsigned xpix = (unsigned)g.x_ld * 4u, ypix = (unsigned)g.y_ld * 4u;   // Act<bf16> offsets count 4-byte units of a float tensor

  u32x2 xr[XV], yr[YV];
  auto prefetch = [&](int ch) {
    const bool cv = ch < nchunks;
    const int chs = cv ? ch : 0;
    const int row = row0 + chs / cpr;
    const int ox0 = (chs % cpr) * l.P;
    const int n_ = row / g.ho, oy = row % g.ho;
    const int iy0 = oy * g.stride - g.pad_t, ix0 = ox0 * g.stride - g.pad_l;
    const int pv = min(l.P, g.wo - ox0);
    unsigned xo[XV], yo[YV];
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int iy = iy0 + (int)((xm[i] >> 23) & 0xFFu) * g.dil, ix = ix0 + (int)((xm[i] >> 12) & 0x7FFu);
      const bool ok = cv & ((int)xm[i] >= 0) & ((unsigned)iy < (unsigned)g.h) & ((unsigned)ix < (unsigned)g.w);
      const unsigned off = (unsigned)((n_ * g.h + iy) * g.w + ix) * xpix + (xm[i] & 0xFFFu) * 4u;
      xo[i] = ok ? off : kOOB;
    }
#pragma unroll
    for (int i = 0; i < YV; ++i) {
      const int yp = (int)((ym[i] >> 12) & 0xFFu);
      const bool ok = cv & ((int)ym[i] >= 0) & (yp < pv);
      const unsigned off = (unsigned)((n_ * g.ho + oy) * g.wo + ox0 + yp) * ypix + (ym[i] & 0xFFFu) * 4u;
      yo[i] = ok ? off : kOOB;
    }
#pragma unroll
    for (int i = 0; i < XV; ++i) xr[i] = Act<bf16>::ld4r(rx, xo[i]);
#pragma unroll
    for (int i = 0; i < YV; ++i) yr[i] = Act<bf16>::ld4r(ry, yo[i]);
  };
  auto commit = [&]() {
    char* base = reinterpret_cast<char*>(smem);
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      if (i < l.xv_n) {
        const int e = tid + i * 256;
        if (e < l.x_elems) {
          const int ky = (int)((xm[i] >> 23) & 0xFFu), colp = (int)((xm[i] >> 12) & 0x7FFu), c = (int)(xm[i] & 0xFFFu);
          u32x2 xv = xr[i];
          if constexpr (PRE) {
            f32x4 v = unpack_bf16x4(xv);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = pre_bn_act(v[j], pm[j], pis[j], pga[j], pbe[j], pal[j], pact);
            xv = pack_bf16x4(v);
          }
          *reinterpret_cast<u32x2*>(base + ((ky * q.Wq + colp) * g.cin + c) * 2) = xv;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < YV; ++i) {
      if (i < l.yv_n) {
        const int e = tid + i * 256;
        if (e < l.y_elems) *reinterpret_cast<u32x2*>(base + q.yl_b + e * 8) = yr[i];
      }
    }
  };

  struct Pair { s16x4 lo, hi; };
  auto operand = [&](int addr, int quarter) {
    Pair pr;
    pr.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + addr));
    pr.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + addr + quarter));
    return __builtin_bit_cast(bf16x8, pr);
  };
  const int steps = q.Pq >> 4;
  prefetch(0);
  for (int ch = 0; ch < nchunks; ++ch) {
    __syncthreads();            // everyone is done reading the previous chunk
    commit();
    __syncthreads();
    prefetch(ch + 1);           // in flight while this chunk is multiplied
    if (has_tile) {             // uniform per wave: every lane of a participating wave takes part in the reads
#pragma unroll 2
      for (int st = 0; st < steps; ++st) {
        bf16x8 av[TPW], bv[TPW];
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
          av[j] = operand(abase[j] + st * astep[j], astep[j] >> 2);
          bv[j] = operand(bbase[j] + st * bstep[j], bstep[j] >> 2);
        }
#pragma unroll
        for (int j = 0; j < TPW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[j], bv[j], acc[j], 0, 0, 0);
      }
    }
  }

  float* pbase = part + (int64_t)strip * a.K_aug * g.cout;
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    if (!colv[j]) continue;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int kk = kt_[j] * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
      if (kk < a.K_aug) pbase[(int64_t)kk * g.cout + col[j]] = acc[j][i];
    }
  }
}

// Strip reduction: 16 outputs x 16 strip-slices per workgroup; every thread adds its strips (slice, slice+16, ...) with
// four independent accumulators, the 16 slices are then combined through LDS in a fixed order (deterministic).
__device__ __forceinline__ void wg_reduce_body(const float* __restrict__ part, int nstrips, int K, int K_aug, int cout,
                                               float* __restrict__ dw, float* __restrict__ dbias, int blk, float* sm) {
  const int E = K_aug * cout;
  const int e = blk * 16 + (threadIdx.x & 15);
  const int slice = threadIdx.x >> 4;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  {
    // four strips per round, loads issued together; strips past the end read zeros through the bounds check
    const __amdgpu_buffer_rsrc_t rp = make_rsrc(part, (unsigned)((int64_t)nstrips * E * 4));
    const unsigned eo = e < E ? (unsigned)e * 4u : kOOB;
    const unsigned rowb = (unsigned)E * 4u;
    for (int c = slice; c < nstrips; c += 64) {
      const float v0 = bload(rp, eo + (unsigned)c * rowb);
      const float v1 = bload(rp, c + 16 < nstrips ? eo + (unsigned)(c + 16) * rowb : kOOB);
      const float v2 = bload(rp, c + 32 < nstrips ? eo + (unsigned)(c + 32) * rowb : kOOB);
      const float v3 = bload(rp, c + 48 < nstrips ? eo + (unsigned)(c + 48) * rowb : kOOB);
      s0 += v0; s1 += v1; s2 += v2; s3 += v3;
    }
  }
  sm[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (threadIdx.x < 16 && e < E) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += sm[q * 16 + threadIdx.x];
    if (e < K * cout) dw[e] = t;
    else if (dbias) dbias[e - K * cout] = t;
  }
}

__global__ __launch_bounds__(256) void conv_bwd_weight_reduce(const float* __restrict__ part, int nstrips, int K,
                                                              int K_aug, int cout, float* __restrict__ dw,
                                                              float* __restrict__ dbias) {
  __shared__ float sm[256];
  wg_reduce_body(part, nstrips, K, K_aug, cout, dw, dbias, (int)blockIdx.x, sm);
}

// The strip reductions of many layers in one launch (their partial sums have been left in place: ssmi_wg_reduce_defer_*): the
// table travels in the kernel arguments, a workgroup finds its layer by its index and runs that layer's reduction unchanged
// (the same additions in the same order as conv_bwd_weight_reduce: results are bit-identical).
static constexpr int kWgBatchMax = 48;
struct WgReduceEntry {
  const float* part;
  float* dw;
  float* dbias;
  int nstrips, K, K_aug, cout;
};
struct WgReduceBatch {
  WgReduceEntry e[kWgBatchMax];
  int first_wg[kWgBatchMax + 1];
  int n;
};
__global__ __launch_bounds__(256) void conv_bwd_weight_reduce_batch(WgReduceBatch b) {
  __shared__ float sm[256];
  int i = 0;
  while (i + 1 < b.n && (int)blockIdx.x >= b.first_wg[i + 1]) ++i;      // uniform
  const WgReduceEntry& q = b.e[i];
  wg_reduce_body(q.part, q.nstrips, q.K, q.K_aug, q.cout, q.dw, q.dbias, (int)blockIdx.x - b.first_wg[i], sm);
}

// The same reduction for long partial rows (the wide layers: hundreds of thousands of elements per row, tens to a few
// hundred rows).  A workgroup owns 256 consecutive elements: 64 lanes x 16 bytes make every load of a wave one
// contiguous kilobyte of a row (the kernel above fetches 64-byte pieces), the four waves take the rows slice, slice + 4,
// ... with four loads in flight per lane, and the four slices are combined through LDS in a fixed order.
// Requires E % 4 == 0 and (K * cout) % 4 == 0.
__global__ __launch_bounds__(256) void conv_bwd_weight_reduce_wide(const float* __restrict__ part, int nstrips, int K,
                                                                   int K_aug, int cout, float* __restrict__ dw,
                                                                   float* __restrict__ dbias) {
  __shared__ f32x4 sm[256];
  const int E = K_aug * cout;
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int e = (blockIdx.x * 64 + lane) * 4;
  const __amdgpu_buffer_rsrc_t rp = make_rsrc(part, (unsigned)((int64_t)nstrips * E * 4));
  const unsigned eo = e < E ? (unsigned)e * 4u : kOOB;
  const unsigned rowb = (unsigned)E * 4u;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  for (int c = slice; c < nstrips; c += 16) {
    const f32x4 v0 = bload4(rp, eo + (unsigned)c * rowb);
    const f32x4 v1 = bload4(rp, c + 4 < nstrips ? eo + (unsigned)(c + 4) * rowb : kOOB);
    const f32x4 v2 = bload4(rp, c + 8 < nstrips ? eo + (unsigned)(c + 8) * rowb : kOOB);
    const f32x4 v3 = bload4(rp, c + 12 < nstrips ? eo + (unsigned)(c + 12) * rowb : kOOB);
    s0 += v0; s1 += v1; s2 += v2; s3 += v3;
  }
  sm[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (slice == 0 && e < E) {
    const f32x4 t = ((sm[lane] + sm[64 + lane]) + sm[128 + lane]) + sm[192 + lane];
    if (e < K * cout) *reinterpret_cast<f32x4*>(dw + e) = t;
    else if (dbias) *reinterpret_cast<f32x4*>(dbias + (e - K * cout)) = t;
  }
}

// Picks the reduction kernel by the shape of the partial-sum matrix.
// reductions recorded instead of launched (ssmi_wg_reduce_defer_begin .. _flush), per host thread
static thread_local std::vector<WgReduceEntry>* g_wg_defer = nullptr;

void launch_wg_reduce(const float* part, int nstrips, int K, int K_aug, int cout, float* dw, float* dbias, hipStream_t st) {
  const int E = K_aug * cout;
  static const bool ablate = getenv("SSMI_ABLATE_WG_REDUCE") != nullptr;     // timing experiment only: results are meaningless
  if (ablate) return;
  if (g_wg_defer && E < 16384) {      // (long rows take the wide kernel at once: their launch is not what they cost)
    g_wg_defer->push_back(WgReduceEntry{part, dw, dbias, nstrips, K, K_aug, cout});
    return;
  }
  static const bool no_wide = getenv("SSMI_NO_WIDE_REDUCE") != nullptr;      // measurement switch
  const bool aligned = ((reinterpret_cast<uintptr_t>(part) | reinterpret_cast<uintptr_t>(dw) | reinterpret_cast<uintptr_t>(dbias)) & 15) == 0;
  if (!no_wide && E >= 16384 && cout % 4 == 0 && aligned)
    hipLaunchKernelGGL(conv_bwd_weight_reduce_wide, dim3((E + 255) / 256), dim3(256), 0, st, part, nstrips, K, K_aug, cout, dw, dbias);
  else
    hipLaunchKernelGGL(conv_bwd_weight_reduce, dim3((E + 15) / 16), dim3(256), 0, st, part, nstrips, K, K_aug, cout, dw, dbias);
}

// column sums of a [npix, c] matrix with row stride ld (bias gradient of a transposed convolution).
// VEC (c % 4 == 0, ld % 4 == 0): thread = (pixel lane, group of 4 channels), 16-byte loads, grid-stride over pixels.
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void colsum_partial(const T* __restrict__ x, int64_t npix, int c, int ld,
                                                      float* __restrict__ part) {
  constexpr int EW = VEC ? 4 : 1;
  const int ce = c / EW;                 // elements per pixel
  const int ppi = 256 / ce;              // pixels per iteration (c <= 256 checked on the host)
  const int cc = threadIdx.x % ce, pl = threadIdx.x / ce;
  __shared__ f32x4 sm[256];
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
  if (pl < ppi) {
    const int64_t stride = (int64_t)gridDim.x * ppi;
    int64_t p = (int64_t)blockIdx.x * ppi + pl;
    for (; p + stride < npix; p += 2 * stride) {
      if (VEC) {
        s0 += Act<T>::gld4(x + p * ld + cc * 4);
        s1 += Act<T>::gld4(x + (p + stride) * ld + cc * 4);
      } else {
        s0[0] += Act<T>::gld(x + p * ld + cc);
        s1[0] += Act<T>::gld(x + (p + stride) * ld + cc);
      }
    }
    if (p < npix) {
      if (VEC) s0 += Act<T>::gld4(x + p * ld + cc * 4);
      else s0[0] += Act<T>::gld(x + p * ld + cc);
    }
  }
  sm[threadIdx.x] = (pl < ppi) ? s0 + s1 : f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  if (threadIdx.x < ce) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < ppi; ++q) t += sm[q * ce + threadIdx.x];
    float* dst = part + (int64_t)blockIdx.x * c + threadIdx.x * EW;
    if (VEC) *reinterpret_cast<f32x4*>(dst) = t;
    else *dst = t[0];
  }
}

__global__ __launch_bounds__(256) void colsum_final(const float* __restrict__ part, int nblocks, int c,
                                                    float* __restrict__ out) {
  // 8 channels x 32 block-slices per workgroup
  __shared__ double sm[256];
  const int ch = blockIdx.x * 8 + (threadIdx.x & 7);
  const int slice = threadIdx.x >> 3;
  double s = 0.0;
  if (ch < c)
    for (int b = slice; b < nblocks; b += 32) s += (double)part[(int64_t)b * c + ch];
  sm[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < 8 && ch < c) {
    double t = 0.0;
    for (int q = 0; q < 32; ++q) t += sm[q * 8 + threadIdx.x];
    out[ch] = (float)t;
  }
}

// Kernel matrix of the data gradient seen as a forward convolution over dy (stride 1):
//   wt[kh-1-ky][kw-1-kx][co][ci] = w[ky][kx][ci][co]
// so that the lanes of an MFMA column read consecutive words, as in the forward pass (in the gather-form data-gradient
// kernel a lane's words are cout * 4 bytes apart: 16-32 cache lines per load on the 64..512-channel layers).
__global__ __launch_bounds__(256) void conv_wt_rot_kernel(const float* __restrict__ w, float* __restrict__ wt, int kh, int kw,
                                                          int cin, int cout) {
  const int total = kh * kw * cin * cout;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int ci = e % cin;
    const int t1 = e / cin;
    const int co = t1 % cout;
    const int tap = t1 / cout;
    const int ky = tap / kw, kx = tap - ky * kw;
    wt[e] = w[(((kh - 1 - ky) * kw + (kw - 1 - kx)) * cin + ci) * cout + co];
  }
}

// Data gradient of wide stride-1 layers through the forward kernel: worth a rotation launch when both channel counts are
// at least 64 (U-Net / SegNet: 3x3 64->64 at 512x512, batch 4: 1579 us gather form vs 793 us forward form).
// A matrix the caller rotated beforehand (ssmi_conv_rotate_weights, one launch for a whole parameter arena) saves the
// per-layer rotation launch; the layers are the same.  Narrower layers were measured too (every stride-1 ENet layer with
// more than 16 input channels, pre-rotated): the step got slower, 9.88 -> 10.22 ms - their gather-form kernels are bound
// by instruction issue, not by the kernel-matrix loads, and the forward kernel's fused-BN-backward epilogue costs more.
static bool dgrad_as_fwd(const Geo& g, bool prerotated = false) {
  (void)prerotated;
  static const bool off = getenv("SSMI_NO_DGRAD_AS_FWD") != nullptr;
  static const int min_ch = [] { const char* e = getenv("SSMI_DGRAD_AS_FWD_MIN"); return e ? atoi(e) : 32; }();   // measurement switch (ENet fp32 step: 64 -> 9.68 ms, 32 -> 9.62, 16 -> 9.84)
  return !off && g.stride == 1 && g.cin >= min_ch && g.cout >= min_ch && g.cout % 8 == 0 && g.y_ld % 4 == 0 && g.ho == g.h && g.wo == g.w;
}
static size_t dgrad_wt_bytes(const Geo& g) { return dgrad_as_fwd(g) ? (size_t)g.kh * g.kw * g.cin * g.cout * sizeof(float) : 0; }

struct RotEntry {          // mirrors ssmi_rot_entry
  int64_t offset;
  int32_t kh, kw, cin, cout;
};
__global__ __launch_bounds__(256) void conv_wt_rot_batch_kernel(const RotEntry* __restrict__ table, const float* __restrict__ theta,
                                                                float* __restrict__ theta_rot) {
  const RotEntry en = table[blockIdx.y];
  const float* w = theta + en.offset;
  float* wt = theta_rot + en.offset;
  const int total = en.kh * en.kw * en.cin * en.cout;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int ci = e % en.cin;
    const int t1 = e / en.cin;
    const int co = t1 % en.cout;
    const int tap = t1 / en.cout;
    const int ky = tap / en.kw, kx = tap - ky * en.kw;
    wt[e] = w[(((en.kh - 1 - ky) * en.kw + (en.kw - 1 - kx)) * en.cin + ci) * en.cout + co];
  }
}

// bfloat16 kernel packs of every table entry: fpack[tap][co][ci] and rpack[ntaps-1-tap][ci][co] = bf16(w[tap][ci][co])
__global__ __launch_bounds__(256) void conv_pack_bf16_batch_kernel(const RotEntry* __restrict__ table, const float* __restrict__ theta,
                                                                   bf16* __restrict__ fpack, bf16* __restrict__ rpack) {
  const RotEntry en = table[blockIdx.y];
  const float* w = theta + en.offset;
  const int per_tap = en.cin * en.cout, ntaps = en.kh * en.kw;
  const int total = ntaps * per_tap;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int tap = e / per_tap;
    const int rem = e - tap * per_tap;
    const int ci = rem / en.cout, co = rem - ci * en.cout;
    const bf16 v = (bf16)w[e];
    // a layer's pack starts at element 2 * offset of the pack array: byte offset 4 * offset, 16-byte aligned like theta's segments
    if (fpack) fpack[2 * en.offset + (int64_t)(tap * en.cout + co) * en.cin + ci] = v;
    if (rpack) rpack[2 * en.offset + (int64_t)(ntaps - 1 - tap) * per_tap + rem] = v;
  }
}

// Which passes of a bfloat16 layer run on the bfloat16 matrix instructions when the packs are there.  Forward: input
// channels in whole 16-blocks (16-byte operand loads) and more than 16 output channels (narrower outputs stay on the
// 16-wide register / LDS kernels).  Data gradient: stride 1, same-size output, the mirror conditions.
// Narrow outputs (<= 16 channels) whose reduction conv_regw keeps in registers: 1x1 over 16 / 64 / 128 channels, 3x3 over 16
// (forward geometry; SSMI_NO_BF16_REGW=1 keeps them on the float32 matrix instructions).
static bool regw_pack_ok(int kh, int kw, int stride, int c_r, int c_o, int in_ld) {
  static const bool off = getenv("SSMI_NO_BF16_REGW") != nullptr;
  const int ntaps = kh * kw;
  static const bool no_c4 = getenv("SSMI_NO_C4") != nullptr;
  if (!no_c4 && ntaps == 1 && c_r == 16 && c_o == 4) return false;      // conv_c4_kernel's shape: no pack
  return !off && c_o <= 16 && stride == 1 && in_ld % 8 == 0 &&
         ((ntaps == 1 && (c_r == 16 || c_r == 64 || c_r == 128)) || (kh == 3 && kw == 3 && c_r == 16));
}
static bool fwd_pack_ok(const Geo& g) {
  static const bool off = getenv("SSMI_NO_BF16_MFMA") != nullptr;
  // cin % 16 == 0 implies the VEC instantiation
  return !off && g.cin % 16 == 0 && g.x_ld % 8 == 0 && (g.cout > 16 || regw_pack_ok(g.kh, g.kw, g.stride, g.cin, g.cout, g.x_ld));
}
static bool dgrad_pack_ok(const Geo& g) {
  static const bool off = getenv("SSMI_NO_BF16_MFMA") != nullptr;
  return !off && g.stride == 1 && g.ho == g.h && g.wo == g.w && g.cout % 16 == 0 && g.y_ld % 8 == 0 &&
         (g.cin > 16 || regw_pack_ok(g.kh, g.kw, g.stride, g.cout, g.cin, g.y_ld));
}

static ssmi_status check_desc(const ssmi_conv_desc* d) {
  SSMI_REQUIRE(d != nullptr, SSMI_E_INVALID_ARG, "conv desc is NULL");
  SSMI_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0 && d->cin > 0 && d->ho > 0 && d->wo > 0 && d->cout > 0, SSMI_E_SHAPE,
               "conv desc: non-positive dimension");
  SSMI_REQUIRE(d->kh > 0 && d->kw > 0 && d->stride > 0 && d->dil > 0 && d->pad_t >= 0 && d->pad_l >= 0, SSMI_E_SHAPE,
               "conv desc: bad kernel/stride/dilation/pad");
  SSMI_REQUIRE(d->x_ld >= d->cin && d->y_ld >= d->cout, SSMI_E_SHAPE, "conv desc: leading dimension < channels");
  SSMI_REQUIRE(d->dtype == SSMI_F32 || d->dtype == SSMI_BF16, SSMI_E_UNSUPPORTED, "conv desc: unknown dtype %d", d->dtype);
  // every output position must read at least ... nothing to check; but the output extent must not exceed what the
  // padded input can produce
  const int64_t need_h = (int64_t)(d->ho - 1) * d->stride + (int64_t)(d->kh - 1) * d->dil + 1 - d->pad_t;
  const int64_t need_w = (int64_t)(d->wo - 1) * d->stride + (int64_t)(d->kw - 1) * d->dil + 1 - d->pad_l;
  // TF SAME allows pad_after up to pad_before + 1
  SSMI_REQUIRE(need_h <= (int64_t)d->h + d->pad_t + 1 + (d->kh - 1) * d->dil && need_w <= (int64_t)d->w + d->pad_l + 1 + (d->kw - 1) * d->dil,
               SSMI_E_SHAPE, "conv desc: output extent inconsistent with input extent");
  return SSMI_OK;
}

// Algorithmic work of one launch (activations of act_size bytes, float32 kernel): forward reads x and the kernel and writes y; the data gradient reads dy, the
// kernel and the fused addend and writes dx.  flops = 2 * output pixels * taps * cin * cout.
static double conv_bytes(const Geo& g, bool with_add, size_t act_size) {
  const double xin = (double)g.n * g.h * g.w * g.cin, yout = (double)g.n * g.ho * g.wo * g.cout;
  return (double)act_size * (xin + yout + (with_add ? xin : 0.0)) + 4.0 * (double)g.kh * g.kw * g.cin * g.cout;
}
static double conv_flops(const Geo& g) {
  return 2.0 * g.n * g.ho * g.wo * (double)g.kh * g.kw * g.cin * g.cout;
}

static int pick_nt(int c) { return c <= 32 ? 1 : (c <= 64 ? 2 : 4); }

// byte sizes for the buffer descriptors; tensors above 2 GiB - 1 are rejected (32-bit buffer offsets)
static ssmi_status make_sizes(const Geo& g, Sizes* sz) {
  const int64_t xb = (int64_t)g.n * g.h * g.w * g.x_ld * 4, yb = (int64_t)g.n * g.ho * g.wo * g.y_ld * 4;
  const int64_t wb = (int64_t)g.kh * g.kw * g.cin * g.cout * 4;
  SSMI_REQUIRE(xb < 0x7FFFFFFFLL && yb < 0x7FFFFFFFLL && wb < 0x7FFFFFFFLL, SSMI_E_UNSUPPORTED,
               "conv: tensors of 2 GiB or more are not supported (x %lld B, y %lld B)", (long long)xb, (long long)yb);
  sz->x_bytes = (unsigned)xb; sz->y_bytes = (unsigned)yb; sz->w_bytes = (unsigned)wb;
  // measurement switch, a bit per kernel family: 1 forward, 2 gather-form data gradient, 4 <=16-channel GEMMs, 8 weight-gradient strips
  static const int xcd_mask = [] { const char* e = getenv("SSMI_XCD_TILES"); return e ? atoi(e) : 0; }();
  sz->xcd = xcd_mask;
  return SSMI_OK;
}

// Picks the LDS-weight kernel when the kernel matrix fits in 64 KiB of LDS; returns false otherwise.
// SSMI_CONV_KERNEL = "lds" / "direct" / "tile": forces a kernel family for A/B measurements and tests (0 = automatic)
static int conv_kernel_override() {
  static const int force = [] {
    const char* e = getenv("SSMI_CONV_KERNEL");
    if (e == nullptr) return 0;
    if (e[0] == 'l') return 1;
    if (e[0] == 'd') return 2;
    if (e[0] == 't') return 3;
    return 0;
  }();
  return force;
}

template <typename T>
static bool launch_gemm_lds(const Geo& g, const Sizes& sz, int mode, const T* in, const float* w, const float* bias,
                            const T* add, T* out, hipStream_t st, ssmi_status* status, float* part = nullptr,
                            int* part_rows = nullptr, BnBwdArgs bb = BnBwdArgs{}, const void* wpack = nullptr) {
  GemmArgs a;
  a.mode = mode;
  a.wpack = wpack;
  a.C_r = mode == 0 ? g.cin : g.cout;
  a.C_o = mode == 0 ? g.cout : g.cin;
  a.in_h = mode == 0 ? g.h : g.ho;
  a.in_w = mode == 0 ? g.w : g.wo;
  a.in_ld = mode == 0 ? g.x_ld : g.y_ld;
  a.out_ld = mode == 0 ? g.y_ld : g.x_ld;
  a.tiles_x = a.tiles_y = a.pmin_y = a.pmin_x = a.PH = a.PW = a.CP = a.patch_off = 0;
  const int ntaps = g.kh * g.kw;
  if (ntaps > kMaxTaps) return false;
  const int force = conv_kernel_override();
  if (force == 2) return false;
  const int s = g.stride;
  const int qh = mode == 0 ? g.ho : (g.h + s - 1) / s, qw = mode == 0 ? g.wo : (g.w + s - 1) / s;
  const bool vec = (a.C_r % 4 == 0) && (a.in_ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(in) & 15) == 0);
  const bool m16 = a.C_o <= 16;
  const int ts = m16 ? 16 : 32;
  const int nt = m16 ? 1 : (a.C_o <= 32 ? 1 : (a.C_o <= 64 ? 2 : 4));
  const int ctiles = (a.C_o + nt * ts - 1) / (nt * ts);
  const int copad = ctiles * nt * ts;
  a.WS = m16 ? copad + 4 : copad;
  // a stride phase of the data gradient sees at most ceil(k/s)^2 taps (all of them when dilated)
  const int staged_taps = (mode == 0 || g.dil > 1) ? ntaps : ((g.kh + s - 1) / s) * ((g.kw + s - 1) / s);
  size_t lds = (size_t)(4 + 5 * kMaxTaps + (size_t)staged_taps * a.C_r * a.WS) * 4;
  if (lds > 64 * 1024) return false;
  if ((part || bb.x) && lds < (size_t)12 * nt * ts * 4) lds = (size_t)12 * nt * ts * 4;   // room for the column-sum combine

  // Three input channels (the first layer): vector arithmetic as well
  static const bool no_c4 = getenv("SSMI_NO_C4") != nullptr;
  if (!no_c4 && force == 0 && !wpack && mode == 0 && g.cin == 3 && g.cout <= 16 && ntaps <= kC4MaxTaps && !add && !part && !bb.x &&
      !bb.mean && g.y_ld % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && (int64_t)g.n * g.ho * g.wo < 0x7FFFFFFFLL) {
    const int64_t M = (int64_t)g.n * g.ho * g.wo;
    set_last_kernel("conv_c3_kernel");
    hipLaunchKernelGGL((conv_c3_kernel<T>), dim3((unsigned)ceil_div64(M, 256)), dim3(256), 0, st, g, sz, in, w, bias, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
      set_error("conv_c3_kernel launch: %s", hipGetErrorString(e));
      *status = SSMI_E_HIP;
    } else {
      *status = SSMI_OK;
    }
    return true;
  }
  // Four reduction channels: the vector-arithmetic kernel (SSMI_NO_C4=1 keeps the 16-wide matrix tiles)
  const bool c4_wide = a.C_r == 16 && a.C_o == 4 && ntaps == 1;      // sixteen reduction channels, single tap
  if (!no_c4 && force == 0 && !wpack && ((a.C_r == 4 && (a.C_o == 4 || a.C_o == 16)) || c4_wide) && vec && ntaps <= kC4MaxTaps &&
      a.out_ld % 4 == 0 && (!bb.x || bb.ld % 4 == 0) && (!bb.skip || bb.skip_ld % 4 == 0) &&
      ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(add) | reinterpret_cast<uintptr_t>(bb.x) |
        reinterpret_cast<uintptr_t>(bb.skip)) & 15) == 0) {
    const int64_t M = (int64_t)g.n * (mode == 0 ? (int64_t)g.ho * g.wo : (int64_t)g.h * g.w);
    if (M * (a.C_o / 4) < 0x7FFFFFFFLL) {
      // (the statistics workspaces hold ceil(M / 128) + 1024 rows)
      static const int c4_wgs = getenv("SSMI_C4_WGS") ? atoi(getenv("SSMI_C4_WGS")) : 1024;
      int64_t bx = ceil_div64(M * (a.C_o / 4), 256);
      const int64_t cap = std::min<int64_t>(c4_wgs > 0 ? c4_wgs : 1024, ceil_div64(M, 128) + 1024);
      if (bx > cap) bx = cap;
      if (part_rows) *part_rows = (int)bx;
      set_last_kernel("conv_c4_kernel<%d, %d>", a.C_o / 4, a.C_r / 4);
      if (c4_wide) hipLaunchKernelGGL((conv_c4_kernel<T, 1, 4>), dim3((unsigned)bx), dim3(256), 0, st, g, sz, a, in, w, bias, add, out, part, bb);
      else if (a.C_o == 4) hipLaunchKernelGGL((conv_c4_kernel<T, 1>), dim3((unsigned)bx), dim3(256), 0, st, g, sz, a, in, w, bias, add, out, part, bb);
      else hipLaunchKernelGGL((conv_c4_kernel<T, 4>), dim3((unsigned)bx), dim3(256), 0, st, g, sz, a, in, w, bias, add, out, part, bb);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) {
        set_error("conv_c4_kernel launch: %s", hipGetErrorString(e));
        *status = SSMI_E_HIP;
      } else {
        *status = SSMI_OK;
      }
      return true;
    }
  }

  // Register-weight kernel: narrow outputs, stride 1, the two reductions that dominate the 16-channel stages of ENet
  // (1x1 over 64 channels, 3x3 over 16 channels).  SSMI_NO_REGW=1 keeps the LDS kernel (A/B measurements).
  static const bool no_regw = getenv("SSMI_NO_REGW") != nullptr;
  if (!no_regw && force == 0 && m16 && vec && s == 1 &&
      ((ntaps == 1 && (a.C_r == 16 || a.C_r == 64 || a.C_r == 128)) || (ntaps == 9 && g.kh == 3 && a.C_r == 16)) &&
      (mode == 0 || (g.ho == g.h && g.wo == g.w)) && !(mode == 0 && bb.mean != nullptr && bb.x == nullptr)) {
    const int64_t M = (int64_t)g.n * qh * qw;
    // Measured on MI355X (batch 16, 128x128): 3x3 16->16 forward 28.3 -> 23.3 us and data gradient 28.4 -> 22.9 us with two
    // tiles per wave and 512 workgroups (four groups per wave, no remainder); 1x1 64->16 19.0 -> 16.7 us and the data
    // gradient of 1x1 16->64 19.9 -> 16.5 us with one tile per wave and 1024 workgroups.  More workgroups re-read the
    // kernel matrix more often, four tiles per wave (1x1) lose occupancy: both measured slower.
    static const int regw_wgs_env = getenv("SSMI_REGW_WGS") ? atoi(getenv("SSMI_REGW_WGS")) : 0;
    const int regw_wgs = regw_wgs_env > 0 ? regw_wgs_env : (ntaps == 1 ? 1024 : 512);
    static const int regw_mt = getenv("SSMI_REGW_MT") ? atoi(getenv("SSMI_REGW_MT")) : 0;
    const int mt_ = (regw_mt == 1 || regw_mt == 2 || regw_mt == 4) ? regw_mt : (ntaps == 1 ? 1 : 2);
    const int64_t ngroups = ceil_div64(M, mt_ * 16);
    int64_t bx = ceil_div64(ngroups, 4);
    if (bx > regw_wgs) bx = regw_wgs;
    if (part_rows) *part_rows = (int)bx;
    bool bf = false;
    if constexpr (sizeof(T) == 2) bf = wpack != nullptr && mode == 0 && (a.in_ld % 4) == 0;
    set_last_kernel(bf ? "conv_regw<%d, %d, %d> bf16 mfma" : "conv_regw<%d, %d, %d>", ntaps, a.C_r, mt_);
#define LR(NTAP_, CR_, MT_)                                                                                                          \
    do {                                                                                                                             \
      if constexpr (sizeof(T) == 2) {                                                                                                \
        if (bf) {                                                                                                                    \
          hipLaunchKernelGGL((conv_regw<T, NTAP_, CR_, MT_, true>), dim3((unsigned)bx), dim3(256), 0, st, g, sz, a, in, w, bias, add, out, part, bb); \
          break;                                                                                                                     \
        }                                                                                                                            \
      }                                                                                                                              \
      hipLaunchKernelGGL((conv_regw<T, NTAP_, CR_, MT_>), dim3((unsigned)bx), dim3(256), 0, st, g, sz, a, in, w, bias, add, out, part, bb); \
    } while (0)
    if (ntaps == 1 && a.C_r == 64) { if (mt_ == 1) LR(1, 64, 1); else if (mt_ == 2) LR(1, 64, 2); else LR(1, 64, 4); }
    else if (ntaps == 1 && a.C_r == 16) { if (mt_ == 1) LR(1, 16, 1); else LR(1, 16, 2); }
    else if (ntaps == 1) { if (mt_ == 1) LR(1, 128, 1); else LR(1, 128, 2); }
    else { if (mt_ == 1) LR(9, 16, 1); else LR(9, 16, 2); }
#undef LR
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
      set_error("conv_regw launch: %s", hipGetErrorString(e));
      *status = SSMI_E_HIP;
    } else {
      *status = SSMI_OK;
    }
    return true;
  }

  if (wpack) return false;      // the caller's `w` need not be this geometry's kernel: only the pack is

  // Tile variant (opt-in, SSMI_CONV_KERNEL=tile): multi-tap stride-1 convolutions into <= 32 channels read every input
  // pixel once per tap; staging the 8 x 32 tile's halo patch in LDS makes that one pass over global memory and removes
  // the per-tap address and bounds arithmetic from the inner loop.  Measured on MI355X (ENet shapes, batch 16) it does
  // not pay: 3x3 16->16 at 128x128 31.6 vs 31.2 us, 3x3 32->32 at 64x64 24.7 vs 18.0 us, whole step 11.69 vs 11.34 ms -
  // these launches are bound by instruction issue and fixed per-workgroup work, not by the operand path.
  bool tile = false;
  if (force == 3 && s == 1 && ntaps > 1 && vec && a.C_o <= 32 && a.C_r <= 32) {
    const int ymin = mode == 0 ? -g.pad_t : g.pad_t - (g.kh - 1) * g.dil, ymax = mode == 0 ? (g.kh - 1) * g.dil - g.pad_t : g.pad_t;
    const int xmin = mode == 0 ? -g.pad_l : g.pad_l - (g.kw - 1) * g.dil, xmax = mode == 0 ? (g.kw - 1) * g.dil - g.pad_l : g.pad_l;
    a.pmin_y = ymin; a.pmin_x = xmin;
    a.PH = 8 + ymax - ymin; a.PW = 32 + xmax - xmin; a.CP = a.C_r + 4;
    a.patch_off = (int)((lds + 15) / 16) * 4;
    const size_t total = (size_t)a.patch_off * 4 + ((size_t)a.PH * a.PW * a.CP + 4) * 4;
    if (total <= 120 * 1024) {
      tile = true;
      lds = total;
      a.tiles_y = (qh + 7) / 8;
      a.tiles_x = (qw + 31) / 32;
    }
  }
  // Measured on MI355X (ENet layer shapes, batch 16): without the tile variant the LDS-weight kernel wins for outputs of
  // <= 16 channels (16x16 MFMA tiles waste no columns) unless the reduction is a single 4-channel tap; elsewhere the
  // direct kernel's larger wave count hides latency better.
  if (!tile && force != 1 && (a.C_o > 16 || (mode == 0 && ntaps * a.C_r <= 4))) return false;

  const int kb = (64 / ts) * (vec ? 4 : 1);
  a.q_blk = kb / a.C_r;
  a.r_blk = kb % a.C_r;
  const int64_t M = (int64_t)g.n * qh * qw;
  // 16-wide tiles: two pixel tiles (32 pixels) per wave keep the kernel at 120 VGPRs = four waves per SIMD, whose
  // prologue, multiply and store phases overlap; four tiles (172 VGPRs, two waves per SIMD) measured 14-24 % slower per
  // layer and 2.6 % per ENet training step, one tile 0.9 % slower than two (MI355X).  SSMI_M16_MT=4 selects the old shape.
  static const int m16_mt = getenv("SSMI_M16_MT") ? atoi(getenv("SSMI_M16_MT")) : 2;
  const int mt = m16 ? ((m16_mt == 4 || tile) ? 4 : 2) : (nt >= 4 ? 1 : 2);   // the tile variant is built for 64 pixels
  const int64_t ngroups = ceil_div64(M, (int64_t)mt * ts);
  int64_t bx = tile ? (int64_t)g.n * a.tiles_y * a.tiles_x : ceil_div64(ngroups, 4);
  // enough workgroups to fill the chip, few enough that each stages the kernel matrix for several pixel groups
  // (measured with 32 pixels per wave: a cap of 512 or 2048 workgroups is 5-15 % slower per layer than 1024)
  if (bx > 1024) bx = 1024;
  dim3 grid((unsigned)bx, (unsigned)ctiles, mode == 0 ? 1u : (unsigned)(s * s));
  if (part_rows) *part_rows = (int)bx;
  hipError_t attr_err = hipSuccess;
#define LG(M16_, MT_, NT_, V_, D_, T_)                                                                                      \
  do {                                                                                                                      \
    auto kfn = conv_gemm_lds<T, M16_, MT_, NT_, V_, D_, T_>;                                                                   \
    if (lds > 64 * 1024)                                                                                                    \
      attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(kfn, grid, dim3(256), lds, st, g, sz, a, in, w, bias, add, out, part, bb);                           \
  } while (0)
  // prefetch depth: cover the whole reduction when it is short, else as deep as the register budget allows
  const int nblk = (staged_taps * a.C_r + kb - 1) / kb;
  if (tile) {
    set_last_kernel("conv_gemm_lds<%s, %d, 1, true, 2, true>", m16 ? "true" : "false", mt);
    if (m16) LG(true, 4, 1, true, 2, true); else LG(false, 2, 1, true, 2, true);
  } else {
    set_last_kernel("conv_gemm_lds<%s, %d, %d, %s, %d, false>", m16 ? "true" : "false", mt, nt, vec ? "true" : "false",
                    m16 ? (nblk <= 2 ? 2 : 4) : (nt == 1 ? (nblk <= 4 ? 4 : 8) : (nt == 2 ? 4 : (nblk <= 4 ? 4 : 8))));
    if (m16) {
#define LM(MT_, D_) do { if (vec) LG(true, MT_, 1, true, D_, false); else LG(true, MT_, 1, false, D_, false); } while (0)
      if (nblk <= 2) { if (mt == 2) LM(2, 2); else LM(4, 2); }
      else { if (mt == 2) LM(2, 4); else LM(4, 4); }
#undef LM
    } else if (nt == 1) {
      if (nblk <= 4) { if (vec) LG(false, 2, 1, true, 4, false); else LG(false, 2, 1, false, 4, false); }
      else { if (vec) LG(false, 2, 1, true, 8, false); else LG(false, 2, 1, false, 8, false); }
    } else if (nt == 2) {
      if (vec) LG(false, 2, 2, true, 4, false); else LG(false, 2, 2, false, 4, false);
    } else {
      if (nblk <= 4) { if (vec) LG(false, 1, 4, true, 4, false); else LG(false, 1, 4, false, 4, false); }
      else { if (vec) LG(false, 1, 4, true, 8, false); else LG(false, 1, 4, false, 8, false); }
    }
  }
#undef LG
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = attr_err;
  if (e != hipSuccess) {
    set_error("conv_gemm_lds launch: %s", hipGetErrorString(e));
    *status = SSMI_E_HIP;
  } else {
    *status = SSMI_OK;
  }
  return true;
}

// part != nullptr: the kernel also writes per-workgroup column sums of y and y^2, part[row][2][cout]; *part_rows = rows.
// bn.mean != nullptr: inference-phase BatchNormalization + activation applied in the epilogue (bn.x stays nullptr).
// forward-form launches (the data gradient of a 1x1 layer) that can form a merge backward's output on load (PRE == 2)
static bool premerge_ok(const Geo& f) {
  return f.kh == 1 && f.kw == 1 && f.stride == 1 && f.cin % 8 == 0 && f.cin <= 128 && f.cout <= 32 && f.x_ld == f.cin;
}

// layers whose forward pass can take act(BN(x)) of a raw input (PRE instantiations of conv_fwd_kernel)
static bool fwd_prebn_ok(const Geo& g) {
  return g.kh == 1 && g.kw == 1 && g.stride == 1 && g.cin % 8 == 0 && g.cin <= 64 && g.cout > 32 && g.x_ld % 4 == 0;
}

template <typename T>
static ssmi_status launch_fwd(const Geo& g, const T* x, const float* w, const float* bias, T* y,
                              hipStream_t st, float* part = nullptr, int* part_rows = nullptr, BnBwdArgs bn = BnBwdArgs{},
                              bool direct_only = false) {
  Sizes sz;
  ssmi_status e = make_sizes(g, &sz);
  if (e) return e;
  ProfileScope prof(st, conv_bytes(g, bn.add != nullptr, sizeof(T)), conv_flops(g));
  bn.wpack = nullptr;
  if (sizeof(T) == 2 && g.wfp && fwd_pack_ok(g) && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(g.wfp) & 15) == 0) {
    bn.wpack = g.wfp;
    direct_only = true;
  }
  const bool pre = bn.pre[0] != nullptr;
  if (pre) direct_only = true;
  if constexpr (sizeof(T) == 2) {
    if (bn.wpack && g.cout <= 16) {      // narrow output: the register-weight kernel on the bfloat16 matrix instructions
      if (!pre && bn.pm_main == nullptr && !(bn.mean != nullptr && bn.x == nullptr) &&
          launch_gemm_lds<T>(g, sz, 0, x, w, bias, static_cast<const T*>(bn.add), y, st, &e, part, part_rows, bn, bn.wpack))
        return e;
      // else (an override or an epilogue that kernel does not have): the direct kernel below, from the same pack
    }
  }
  if constexpr (sizeof(T) == 2) {
    if (bn.wpack && !pre && bn.pm_main == nullptr &&
        conv_tiled_bf16_ok(g.h, g.w, g.cin, g.cout, g.kh, g.kw, g.stride, g.dil, g.pad_t, g.pad_l, g.ho, g.wo, g.x_ld, g.y_ld)) {
      const ssmi_status te = launch_conv_tiled_bf16(g.n, g.h, g.w, g.cin, g.cout, g.x_ld, g.y_ld, x, bn.wpack, bias, y, part, part_rows, bn, st);
      if (te != SSMI_E_UNSUPPORTED) return te;
    }
  }
  if (!direct_only && launch_gemm_lds<T>(g, sz, 0, x, w, bias, (const T*)nullptr, y, st, &e, part, part_rows, bn)) return e;
  if (part_rows) *part_rows = (int)ceil_div64((int64_t)g.n * g.ho * g.wo, 128);
  const int64_t M = (int64_t)g.n * g.ho * g.wo;
  const int nt = pick_nt(g.cout);
  dim3 grid((unsigned)ceil_div64(M, 128), (unsigned)((g.cout + 32 * nt - 1) / (32 * nt)), 1);
  const bool vec = (g.cin % 8 == 0) && (g.x_ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  if (bn.pm_main != nullptr) {
    if constexpr (sizeof(T) == 4) {
      SSMI_REQUIRE(premerge_ok(g) && vec && nt == 1 && bn.x != nullptr, SSMI_E_UNSUPPORTED,
                   "conv2d_bwd_data (merge backward on load): layer shape not supported");
      set_last_kernel("conv_fwd_kernel<1, true> premerge");
      hipLaunchKernelGGL((conv_fwd_kernel<T, 1, true, 2>), grid, dim3(256), 0, st, g, sz, x, w, bias, y, part, bn);
      SSMI_LAUNCH_CHECK();
      return SSMI_OK;
    } else {
      set_error("conv2d_bwd_data (merge backward on load): float32 tensors only");
      return SSMI_E_UNSUPPORTED;
    }
  }
  if (pre) {
    {
      SSMI_REQUIRE(fwd_prebn_ok(g) && vec && part != nullptr && (sizeof(T) == 4 || g.cin % 16 == 0), SSMI_E_UNSUPPORTED,
                   "conv2d_fwd (normalise on load): needs a 1x1 stride-1 layer over 8..64 channels (multiple of 8) with more than 32 "
                   "output channels, 16-byte aligned rows and the statistics epilogue");
      set_last_kernel(bn.wpack ? "conv_fwd_kernel<%d, true> prebn bf16 mfma" : "conv_fwd_kernel<%d, true> prebn", nt);
      if (nt == 2) hipLaunchKernelGGL((conv_fwd_kernel<T, 2, true, 1>), grid, dim3(256), 0, st, g, sz, x, w, bias, y, part, bn);
      else hipLaunchKernelGGL((conv_fwd_kernel<T, 4, true, 1>), grid, dim3(256), 0, st, g, sz, x, w, bias, y, part, bn);
      SSMI_LAUNCH_CHECK();
      return SSMI_OK;
    }
  }
  // kernel rows through LDS: wide layers only (a 1x1 layer over 32 channels has a single chunk and nothing to share)
  static const bool no_lds_b = getenv("SSMI_NO_LDS_B") != nullptr;
  bn.lds_b = (!no_lds_b && bn.wpack && nt >= 2 && vec && g.cin % 32 == 0 && g.kh * g.kw * g.cin >= 256) ? 1 : 0;
  set_last_kernel(bn.wpack ? "conv_fwd_kernel<%d, %s> bf16 mfma" : "conv_fwd_kernel<%d, %s>", nt, vec ? "true" : "false");
#define L(NT, V) hipLaunchKernelGGL((conv_fwd_kernel<T, NT, V>), grid, dim3(256), 0, st, g, sz, x, w, bias, y, part, bn)
  if (vec) {
    if (nt == 1) L(1, true); else if (nt == 2) L(2, true); else L(4, true);
  } else {
    if (nt == 1) L(1, false); else if (nt == 2) L(2, false); else L(4, false);
  }
#undef L
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

// bb.x != nullptr (stride 1 only): the kernel also writes the column sums of the fused BN backward to bb.part[row][3][cin];
// *part_rows = rows.
// ---------------------------------------------------------------------------------------------------
// 2x2 / stride-2 data gradient = forward of a 2x2 / stride-2 transposed convolution ("pixel shuffle"): every dy pixel
// owns a private 2x2 block of dx, so the four stride phases are four column groups of ONE GEMM
//     dx[n, 2y+a, 2x+b, ci] = bias[ci] + add[...] + sum_co dy[n, y, x, co] * w[a, b, ci, co]
//     C[p, j] = sum_k A[p, k] * w[j * cout + k],   j = (a*2 + b) * cin + ci
// A wave owns 32 dy pixels x NT*32 columns: dy is read once (not once per phase), and for x_ld == cin the columns of
// one output row (a fixed) are 2*cin contiguous floats, so a store instruction writes 128 contiguous bytes.
// ---------------------------------------------------------------------------------------------------
template <typename T, int NT>
__global__ __launch_bounds__(256, (NT == 1 ? 6 : (NT == 2 ? 4 : 3))) void conv_bwd_data_2x2s2_kernel(Geo g, Sizes sz, const T* __restrict__ dy,
                                                                  const float* __restrict__ w,
                                                                  const T* __restrict__ add,
                                                                  const float* __restrict__ bias, T* __restrict__ dx) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const unsigned M = (unsigned)(g.n * g.ho * g.wo);
  const unsigned tile0 = (blockIdx.x * 4u + (unsigned)wave) * 32u;
  if (tile0 >= M) return;
  const int N = 4 * g.cin;
  const int jbase = blockIdx.y * 32 * NT;
  const __amdgpu_buffer_rsrc_t ry = Act<T>::rsrc(dy, sz.y_bytes), rw = make_rsrc(w, sz.w_bytes);
  const unsigned p = tile0 + (unsigned)r;
  const unsigned yo = p < M ? p * (unsigned)g.y_ld * 4u : kOOB;
  unsigned wo_[NT];
  bool jv[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int j = jbase + t * 32 + r;
    jv[t] = j < N;
    wo_[t] = jv[t] ? (unsigned)j * (unsigned)g.cout * 4u : kOOB;
  }
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  // reduction over cout in blocks of 8 (lane half hh takes channels 4*hh.. of the block for A and B alike)
  auto fetch = [&](int k0, f32x4& a, f32x4 (&b)[NT]) {
    const unsigned ko = k0 < g.cout ? (unsigned)(k0 + 4 * hh) * 4u : kOOB;
    a = Act<T>::ld4(ry, (yo + ko) | ((yo | ko) & kOOB));
#pragma unroll
    for (int t = 0; t < NT; ++t) b[t] = bload4(rw, (wo_[t] + ko) | ((wo_[t] | ko) & kOOB));
  };
  auto mma = [&](const f32x4& a, const f32x4 (&b)[NT]) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = MFMA32(a[q], b[t][q], acc[t]);
  };
  f32x4 a0, a1, b0[NT], b1[NT];
  fetch(0, a0, b0);
  for (int k0 = 0; k0 < g.cout; k0 += 16) {
    fetch(k0 + 8, a1, b1);             // past the end: zeros
    mma(a0, b0);
    fetch(k0 + 16, a0, b0);
    if (k0 + 8 < g.cout) mma(a1, b1);
  }
  // epilogue: dx pixel of (p, a, b) = 4p - 2x + a*W + b  (W = 2*wo; x = p mod wo)
  const __amdgpu_buffer_rsrc_t rdx = Act<T>::rsrc(dx, sz.x_bytes);
  const __amdgpu_buffer_rsrc_t radd = Act<T>::rsrc(add ? add : dx, add ? sz.x_bytes : 0u);
  const unsigned x0 = tile0 % (unsigned)g.wo;
  unsigned rowpix[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const unsigned off = (unsigned)((i & 3) + 8 * (i >> 2) + 4 * hh);
    const unsigned po = tile0 + off;
    const unsigned xi = (x0 + off) % (unsigned)g.wo;
    rowpix[i] = po < M ? 4u * po - 2u * xi : 0xFFFFFFFFu;
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int j = jbase + t * 32 + r;
    const int ph = jv[t] ? j / g.cin : 0;
    const int ci = jv[t] ? j - ph * g.cin : 0;
    const unsigned coff = (unsigned)((ph >> 1) * g.w + (ph & 1));      // pixel offset of the phase inside the 2x2 block
    const float bv = bias ? bload(make_rsrc(bias, (unsigned)g.cin * 4u), jv[t] ? (unsigned)ci * 4u : kOOB) : 0.f;
    unsigned o[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
      o[i] = (jv[t] && rowpix[i] != 0xFFFFFFFFu) ? ((rowpix[i] + coff) * (unsigned)g.x_ld + (unsigned)ci) * 4u : kOOB;
    float av[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) av[i] = Act<T>::ld(radd, o[i]);
#pragma unroll
    for (int i = 0; i < 16; ++i) Act<T>::st(rdx, o[i], acc[t][i] + bv + av[i]);
  }
}

template <typename T>
static ssmi_status launch_bwd_data(const Geo& g, const T* dy, const float* w, const T* add, const float* bias,
                                   T* dx, hipStream_t st, BnBwdArgs bb = BnBwdArgs{}, int* part_rows = nullptr,
                                   float* wt = nullptr, const float* w_rot = nullptr) {
  Sizes sz;
  ssmi_status e = make_sizes(g, &sz);
  if (e) return e;
  if (sizeof(T) == 2 && g.wrp && dgrad_pack_ok(g) && (reinterpret_cast<uintptr_t>(dy) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(g.wrp) & 15) == 0) {
    // bfloat16 matrix instructions: forward convolution over dy with the data-gradient pack [tap'][ci][co]
    Geo f = g;
    f.h = g.ho; f.w = g.wo; f.cin = g.cout; f.x_ld = g.y_ld;
    f.ho = g.h; f.wo = g.w; f.cout = g.cin; f.y_ld = g.x_ld;
    f.pad_t = (g.kh - 1) * g.dil - g.pad_t;
    f.pad_l = (g.kw - 1) * g.dil - g.pad_l;
    f.wfp = g.wrp;
    f.wrp = nullptr;
    bb.add = add;
    return launch_fwd<T>(f, dy, w, bias, dx, st, nullptr, part_rows, bb, true);
  }
  if ((w_rot && dgrad_as_fwd(g, true)) || (wt && dgrad_as_fwd(g))) {
    // forward convolution over dy with the rotated, transposed kernel matrix: n, (ho, wo, cout) -> (h, w, cin)
    if (w_rot && dgrad_as_fwd(g, true)) {
      wt = const_cast<float*>(w_rot);       // rotated by the caller, read only
    } else {
      const int total = g.kh * g.kw * g.cin * g.cout;
      hipLaunchKernelGGL(conv_wt_rot_kernel, dim3((unsigned)std::min(1024, (total + 255) / 256)), dim3(256), 0, st, w, wt, g.kh,
                         g.kw, g.cin, g.cout);
      SSMI_LAUNCH_CHECK();
    }
    Geo f = g;
    f.h = g.ho; f.w = g.wo; f.cin = g.cout; f.x_ld = g.y_ld;
    f.ho = g.h; f.wo = g.w; f.cout = g.cin; f.y_ld = g.x_ld;
    f.pad_t = (g.kh - 1) * g.dil - g.pad_t;
    f.pad_l = (g.kw - 1) * g.dil - g.pad_l;
    bb.add = add;
    return launch_fwd<T>(f, dy, wt, bias, dx, st, nullptr, part_rows, bb, true);
  }
  ProfileScope prof(st, conv_bytes(g, add != nullptr, sizeof(T)), conv_flops(g));
  if (g.kh == 2 && g.kw == 2 && g.stride == 2 && g.dil == 1 && g.pad_t == 0 && g.pad_l == 0 && g.h == 2 * g.ho &&
      g.w == 2 * g.wo && g.cout % 8 == 0 && g.y_ld % 4 == 0 && bb.x == nullptr && part_rows == nullptr &&
      ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(w)) & 15) == 0 && conv_kernel_override() == 0) {
    const int N = 4 * g.cin;
    const int nt = N <= 32 ? 1 : (N <= 64 ? 2 : (N % 96 == 0 ? 3 : 4));
    const int64_t M = (int64_t)g.n * g.ho * g.wo;
    dim3 grid((unsigned)ceil_div64(M, 128), (unsigned)((N + 32 * nt - 1) / (32 * nt)), 1);
    set_last_kernel("conv_bwd_data_2x2s2_kernel<%d>", nt);
#define L(NT) hipLaunchKernelGGL((conv_bwd_data_2x2s2_kernel<T, NT>), grid, dim3(256), 0, st, g, sz, dy, w, add, bias, dx)
    if (nt == 1) L(1); else if (nt == 2) L(2); else if (nt == 3) L(3); else L(4);
#undef L
    SSMI_LAUNCH_CHECK();
    return SSMI_OK;
  }
  if (launch_gemm_lds<T>(g, sz, 1, dy, w, bias, add, dx, st, &e, nullptr, part_rows, bb)) return e;
  if (part_rows) *part_rows = (int)ceil_div64((int64_t)g.n * g.h * g.w, 128);
  const int s = g.stride;
  const int qh = (g.h + s - 1) / s, qw = (g.w + s - 1) / s;
  const int64_t M = (int64_t)g.n * qh * qw;
  const int nt = pick_nt(g.cin);
  dim3 grid((unsigned)ceil_div64(M, 128), (unsigned)((g.cin + 32 * nt - 1) / (32 * nt)), (unsigned)(s * s));
  const bool vec = (g.cout % 8 == 0) && (g.y_ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(dy) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(w) & 15) == 0);
  set_last_kernel("conv_bwd_data_kernel<%d, %s>", nt, vec ? "true" : "false");
#define L(NT, V) hipLaunchKernelGGL((conv_bwd_data_kernel<T, NT, V>), grid, dim3(256), 0, st, g, sz, dy, w, add, bias, dx, bb)
  if (vec) {
    if (nt == 1) L(1, true); else if (nt == 2) L(2, true); else L(4, true);
  } else {
    if (nt == 1) L(1, false); else if (nt == 2) L(2, false); else L(4, false);
  }
#undef L
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

// ---------------------------------------------------------------------------------------------------
// Weight gradient of WIDE layers (cout a multiple of 64, K = taps*cin >= 256: the U-Net / SegNet convolutions) as a
// tiled GEMM: a workgroup owns a (WK*64) x (WC*64) tile of dW and one share of the pixels; per chunk of 32 pixels it
// stages the im2col slab A[32][WK*64] (for each k-row (tap, ci): x at the tap-shifted pixel) and the dy slab
// B[32][WC*64] in LDS (16-byte buffer loads into registers one chunk ahead, then one LDS store), and each wave
// multiplies its 64x64 sub-tile (2x2 MFMA tiles, K = pixels).  Each operand byte is staged once per tile instead of
// once per 12 tiles.  The bias gradient (column sums of dy) is accumulated by the waves of the first k-tile.
// Partial sums per pixel share: part[share][K_aug][cout], reduced by conv_bwd_weight_reduce.
// ---------------------------------------------------------------------------------------------------
template <typename T, int WK, int WC>
__global__ __launch_bounds__(256) void conv_bwd_weight_tiled(Geo g, Sizes sz, int K, int K_aug, int chunks_per_share,
                                                             const T* __restrict__ x, const T* __restrict__ dy,
                                                             float* __restrict__ part) {
  static_assert(WK * WC == 4, "four waves per workgroup");
  constexpr int TK = WK * 64, TC = WC * 64, P = 32;
  constexpr int KA4 = TK / 4, PA = 256 / KA4, NA = P / PA;     // A staging: thread -> (k group, pixel slots)
  constexpr int KB4 = TC / 4, PB = 256 / KB4, NB = P / PB;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + P * TK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int wk = wave % WK, wc = wave / WK;
  const int k0 = blockIdx.x * TK, c0 = blockIdx.y * TC;
  const __amdgpu_buffer_rsrc_t rx = Act<T>::rsrc(x, sz.x_bytes), ry = Act<T>::rsrc(dy, sz.y_bytes);
  const unsigned M = (unsigned)(g.n * g.ho * g.wo);
  const int nchunks = (int)((M + P - 1) / P);
  const int ch0 = blockIdx.z * chunks_per_share;
  const int ch1 = min(nchunks, ch0 + chunks_per_share);

  // this thread's k group (4 consecutive k-rows inside one tap: cin % 4 == 0) and dy column group
  const int ka = tid % KA4, pa0 = tid / KA4;
  const int kk = k0 + ka * 4;
  const bool kvalid = kk < K;
  const int tap = kvalid ? kk / g.cin : 0;
  const int ci = kvalid ? kk - tap * g.cin : 0;
  const int ky = tap / g.kw, kx = tap - ky * g.kw;
  const int offy = ky * g.dil - g.pad_t, offx = kx * g.dil - g.pad_l;
  const int cb = tid % KB4, pb0 = tid / KB4;
  const int co = c0 + cb * 4;
  const bool covalid = co < g.cout;

  f32x4 ra[NA], rb[NB];
  auto prefetch = [&](int ch) {
    const bool cv = ch < ch1;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const unsigned pix = (unsigned)ch * P + (unsigned)(pa0 + j * PA);
      const bool pv = cv && kvalid && pix < M;
      const unsigned pp = pv ? pix : 0u;
      const unsigned t1 = pp / (unsigned)g.wo;
      const int ox = (int)(pp - t1 * (unsigned)g.wo);
      const unsigned n_ = t1 / (unsigned)g.ho;
      const int oy = (int)(t1 - n_ * (unsigned)g.ho);
      const int iy = oy * g.stride + offy, ix = ox * g.stride + offx;
      const bool ok = pv && (unsigned)iy < (unsigned)g.h && (unsigned)ix < (unsigned)g.w;
      ra[j] = Act<T>::ld4(rx, ok ? (((n_ * (unsigned)g.h + (unsigned)iy) * (unsigned)g.w + (unsigned)ix) * (unsigned)g.x_ld + (unsigned)ci) * 4u : kOOB);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const unsigned pix = (unsigned)ch * P + (unsigned)(pb0 + j * PB);
      rb[j] = Act<T>::ld4(ry, (cv && covalid && pix < M) ? (pix * (unsigned)g.y_ld + (unsigned)co) * 4u : kOOB);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
    for (int b_ = 0; b_ < 2; ++b_)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a_][b_][i] = 0.f;
  float bsum0 = 0.f, bsum1 = 0.f;
  const bool do_bias = (K_aug > K) && blockIdx.x == 0 && wk == 0;

  prefetch(ch0);
  for (int ch = ch0; ch < ch1; ++ch) {
    __syncthreads();            // everyone is done reading the previous chunk
#pragma unroll
    for (int j = 0; j < NA; ++j) *reinterpret_cast<f32x4*>(As + (pa0 + j * PA) * TK + ka * 4) = ra[j];
#pragma unroll
    for (int j = 0; j < NB; ++j) *reinterpret_cast<f32x4*>(Bs + (pb0 + j * PB) * TC + cb * 4) = rb[j];
    __syncthreads();
    prefetch(ch + 1);           // in flight while this chunk is multiplied
    const float* ap = As + hh * TK + wk * 64 + r;
    const float* bp = Bs + hh * TC + wc * 64 + r;
#pragma unroll 4
    for (int st = 0; st < P / 2; ++st) {
      const float a0 = ap[st * 2 * TK], a1 = ap[st * 2 * TK + 32];
      const float b0 = bp[st * 2 * TC], b1 = bp[st * 2 * TC + 32];
      acc[0][0] = MFMA32(a0, b0, acc[0][0]);
      acc[0][1] = MFMA32(a0, b1, acc[0][1]);
      acc[1][0] = MFMA32(a1, b0, acc[1][0]);
      acc[1][1] = MFMA32(a1, b1, acc[1][1]);
      if (do_bias) { bsum0 += b0; bsum1 += b1; }
    }
  }

  float* pbase = part + (int64_t)blockIdx.z * K_aug * g.cout;
#pragma unroll
  for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
    for (int b_ = 0; b_ < 2; ++b_) {
      const int col = c0 + wc * 64 + b_ * 32 + r;
      if (col >= g.cout) continue;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int k = k0 + wk * 64 + a_ * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
        if (k < K) pbase[(int64_t)k * g.cout + col] = acc[a_][b_][i];
      }
    }
  if (do_bias) {
    bsum0 += __shfl_xor(bsum0, 32, 64);
    bsum1 += __shfl_xor(bsum1, 32, 64);
    if (hh == 0) {
      const int col = c0 + wc * 64 + r;
      if (col < g.cout) pbase[(int64_t)K * g.cout + col] = bsum0;
      if (col + 32 < g.cout) pbase[(int64_t)K * g.cout + col + 32] = bsum1;
    }
  }
}

// The same tiled GEMM for bf16 tensors on v_mfma_f32_32x32x16_bf16: the slabs stay bf16 rows of pixels in LDS (64 pixels
// per chunk: a chunk of matrix work is eight times shorter than with the float32 instructions, so chunks are longer to
// keep the two barriers per chunk cheap) and the operands come from the transposing LDS read, as in
// conv_bwd_weight_lds_bf16.  Rows are padded by 64 bytes: the four pixel rows and two 16-channel halves that one
// 32-lane half reads in one instruction then fall into distinct banks.
template <int WK, int WC>
__global__ __launch_bounds__(256) void conv_bwd_weight_tiled_bf16(Geo g, Sizes sz, int K, int K_aug, int pix_per_share,
                                                                  const bf16* __restrict__ x, const bf16* __restrict__ dy,
                                                                  float* __restrict__ part) {
  static_assert(WK * WC == 4, "four waves per workgroup");
  constexpr int TK = WK * 64, TC = WC * 64, P = 64;
  constexpr int SA = TK * 2 + 64, SB = TC * 2 + 64;             // row strides in bytes
  constexpr int KA4 = TK / 4, PA = 256 / KA4, NA = P / PA;     // A staging: thread -> (k group, pixel slots)
  constexpr int KB4 = TC / 4, PB = 256 / KB4, NB = P / PB;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  typedef __attribute__((address_space(3))) char* lds_ptr;
  char* const As = reinterpret_cast<char*>(smem);
  char* const Bs = As + P * SA;
  const lds_ptr lA = (lds_ptr)smem;
  const lds_ptr lB = lA + P * SA;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int sq = (lane & 15) >> 2, sp = lane & 3, gh = (lane >> 4) & 1;   // this lane's part in a transposing read
  const int wk = wave % WK, wc = wave / WK;
  const int k0 = blockIdx.x * TK, c0 = blockIdx.y * TC;
  const __amdgpu_buffer_rsrc_t rx = Act<bf16>::rsrc(x, sz.x_bytes), ry = Act<bf16>::rsrc(dy, sz.y_bytes);
  const unsigned M = (unsigned)(g.n * g.ho * g.wo);
  const unsigned p_begin = (unsigned)blockIdx.z * (unsigned)pix_per_share;
  const unsigned p_end = min(M, p_begin + (unsigned)pix_per_share);
  const int nchunks = p_end > p_begin ? (int)((p_end - p_begin + P - 1) / P) : 0;

  const int ka = tid % KA4, pa0 = tid / KA4;
  const int kk = k0 + ka * 4;
  const bool kvalid = kk < K;
  const int tap = kvalid ? kk / g.cin : 0;
  const int ci = kvalid ? kk - tap * g.cin : 0;
  const int ky = tap / g.kw, kx = tap - ky * g.kw;
  const int offy = ky * g.dil - g.pad_t, offx = kx * g.dil - g.pad_l;
  const int cb = tid % KB4, pb0 = tid / KB4;
  const int co = c0 + cb * 4;
  const bool covalid = co < g.cout;

  u32x2 ra[NA], rb[NB];
  auto prefetch = [&](int ch) {
    const bool cv = ch < nchunks;
    unsigned oa[NA], ob[NB];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const unsigned pix = p_begin + (unsigned)ch * P + (unsigned)(pa0 + j * PA);
      const bool pv = cv & kvalid & (pix < p_end);
      const unsigned pp = pv ? pix : 0u;
      const unsigned t1 = pp / (unsigned)g.wo;
      const int ox = (int)(pp - t1 * (unsigned)g.wo);
      const unsigned n_ = t1 / (unsigned)g.ho;
      const int oy = (int)(t1 - n_ * (unsigned)g.ho);
      const int iy = oy * g.stride + offy, ix = ox * g.stride + offx;
      const bool ok = pv & ((unsigned)iy < (unsigned)g.h) & ((unsigned)ix < (unsigned)g.w);
      oa[j] = ok ? (((n_ * (unsigned)g.h + (unsigned)iy) * (unsigned)g.w + (unsigned)ix) * (unsigned)g.x_ld + (unsigned)ci) * 4u : kOOB;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const unsigned pix = p_begin + (unsigned)ch * P + (unsigned)(pb0 + j * PB);
      ob[j] = (cv & covalid & (pix < p_end)) ? (pix * (unsigned)g.y_ld + (unsigned)co) * 4u : kOOB;
    }
#pragma unroll
    for (int j = 0; j < NA; ++j) ra[j] = Act<bf16>::ld4r(rx, oa[j]);
#pragma unroll
    for (int j = 0; j < NB; ++j) rb[j] = Act<bf16>::ld4r(ry, ob[j]);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
    for (int b_ = 0; b_ < 2; ++b_)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a_][b_][i] = 0.f;
  float bsum0 = 0.f, bsum1 = 0.f;
  const bool do_bias = (K_aug > K) && blockIdx.x == 0 && wk == 0;

  struct Pair { s16x4 lo, hi; };
  auto operand = [&](lds_ptr base, int addr, int quarter) {
    Pair pr;
    pr.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + addr));
    pr.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + addr + quarter));
    return __builtin_bit_cast(bf16x8, pr);
  };
  const int a_addr = (8 * hh + sq) * SA + (wk * 64 + 16 * gh + 4 * sp) * 2;
  const int b_addr = (8 * hh + sq) * SB + (wc * 64 + 16 * gh + 4 * sp) * 2;

  prefetch(0);
  for (int ch = 0; ch < nchunks; ++ch) {
    __syncthreads();            // everyone is done reading the previous chunk
#pragma unroll
    for (int j = 0; j < NA; ++j) *reinterpret_cast<u32x2*>(As + (pa0 + j * PA) * SA + ka * 8) = ra[j];
#pragma unroll
    for (int j = 0; j < NB; ++j) *reinterpret_cast<u32x2*>(Bs + (pb0 + j * PB) * SB + cb * 8) = rb[j];
    __syncthreads();
    prefetch(ch + 1);           // in flight while this chunk is multiplied
#pragma unroll
    for (int st = 0; st < P / 16; ++st) {
      const bf16x8 a0 = operand(lA, a_addr + st * 16 * SA, 4 * SA), a1 = operand(lA, a_addr + st * 16 * SA + 64, 4 * SA);
      const bf16x8 b0 = operand(lB, b_addr + st * 16 * SB, 4 * SB), b1 = operand(lB, b_addr + st * 16 * SB + 64, 4 * SB);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
      if (do_bias) {            // uniform per wave
#pragma unroll
        for (int e = 0; e < 8; ++e) { bsum0 += (float)b0[e]; bsum1 += (float)b1[e]; }
      }
    }
  }

  float* pbase = part + (int64_t)blockIdx.z * K_aug * g.cout;
#pragma unroll
  for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
    for (int b_ = 0; b_ < 2; ++b_) {
      const int col = c0 + wc * 64 + b_ * 32 + r;
      if (col >= g.cout) continue;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int k = k0 + wk * 64 + a_ * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
        if (k < K) pbase[(int64_t)k * g.cout + col] = acc[a_][b_][i];
      }
    }
  if (do_bias) {
    bsum0 += __shfl_xor(bsum0, 32, 64);
    bsum1 += __shfl_xor(bsum1, 32, 64);
    if (hh == 0) {
      const int col = c0 + wc * 64 + r;
      if (col < g.cout) pbase[(int64_t)K * g.cout + col] = bsum0;
      if (col + 32 < g.cout) pbase[(int64_t)K * g.cout + col + 32] = bsum1;
    }
  }
}

struct TiledPlan {
  bool use;
  int wk, K, K_aug, shares, chunks_per_share, ktiles, ctiles;
  size_t bytes;
};

static TiledPlan plan_tiled(const Geo& g, bool with_bias) {
  TiledPlan t;
  t.K = g.kh * g.kw * g.cin;
  t.K_aug = t.K + (with_bias ? 1 : 0);
  t.use = (g.cin % 4 == 0) && (g.cout % 64 == 0) && t.K >= 256 && (g.x_ld % 4 == 0) && (g.y_ld % 4 == 0);
  t.wk = (g.cout % 128 == 0) ? 2 : 4;
  if (!t.use) { t.bytes = 0; return t; }
  const int TK = t.wk * 64, TC = (4 / t.wk) * 64;
  t.ktiles = (t.K + TK - 1) / TK;
  t.ctiles = g.cout / TC;
  const int64_t M = (int64_t)g.n * g.ho * g.wo;
  const int nchunks = (int)ceil_div64(M, 32);
  int shares = 1024 / (t.ktiles * t.ctiles);          // about four workgroups per CU in total
  if (shares > nchunks / 8) shares = nchunks / 8;       // at least 8 chunks per workgroup (amortises the pipeline fill)
  if (shares < 1) shares = 1;
  t.chunks_per_share = (nchunks + shares - 1) / shares;
  t.shares = (nchunks + t.chunks_per_share - 1) / t.chunks_per_share;
  t.bytes = (size_t)t.shares * t.K_aug * g.cout * sizeof(float);
  return t;
}

struct WgPlan {
  WgArgs a;
  WgLds l;
  WgBf q;           // bf16 matrix-instruction variant only
  bool use_lds, vec, m16;
  int tpw, groups, blocks_x;
  size_t bytes;
};

// bf16_mfma: plan for conv_bwd_weight_lds_bf16 (32x32 tiles for every layer, bf16 LDS images); use_lds then says whether
// that kernel can run the layer
static WgPlan plan_wg(const Geo& g, bool with_bias, bool bf16_mfma = false) {
  WgPlan p;
  WgArgs& a = p.a;
  a.K = g.kh * g.kw * g.cin;
  a.K_aug = a.K + (with_bias ? 1 : 0);
  const int R = g.n * g.ho;
  auto tiles = [&](int ts) {
    a.ktiles = (a.K_aug + ts - 1) / ts;
    a.cotiles = (g.cout + ts - 1) / ts;
    a.T = a.ktiles * a.cotiles;
    a.wps = a.T < 4 ? a.T : 4;
    int tpw = (a.T + 3) / 4;
    if (tpw > 3) tpw = 3;
    if (tpw < 1) tpw = 1;
    p.tpw = tpw;
    p.groups = (a.T + a.wps * tpw - 1) / (a.wps * tpw);
  };
  // 16x16 tiles when either side of dW is at most 16 long (1x1 16->64: two 32-row tiles would be half padding and keep
  // two of the four waves idle; four 16x16 column tiles take a quarter of the matrix-core time)
  p.m16 = (g.cout <= 16 || a.K_aug <= 16) && !bf16_mfma;
  tiles(p.m16 ? 16 : 32);

  // LDS-staged variant
  WgLds& l = p.l;
  p.vec = (g.cin % 4 == 0) && (g.cout % 4 == 0) && (g.x_ld % 4 == 0) && (g.y_ld % 4 == 0);
  const int ew = p.vec ? 4 : 1;
  l.coutp = a.cotiles * (p.m16 ? 16 : 32);
  // chunk = the longest run of output pixels of one row whose staging fits the per-thread register budget and 64 KiB
  // of LDS (long chunks keep more bytes in flight per workgroup and amortise the two barriers per chunk)
  const int wo_even = (g.wo + 3) & ~3;   // multiple of the pixels per MFMA step
  p.use_lds = false;
  static const int p_max = [] { const char* e = getenv("SSMI_WG_PMAX"); return e ? atoi(e) : 256; }();   // measurement switch
  for (int P = wo_even < p_max ? wo_even : p_max; P >= 4; P = (P > 64 ? P - 32 : P - 4)) {
    l.P = P;
    l.Wp = (P - 1) * g.stride + (g.kw - 1) * g.dil + 1;
    l.x_elems = g.kh * l.Wp * (g.cin / ew);
    l.y_elems = P * ((g.cout + ew - 1) / ew);
    l.xv_n = (l.x_elems + 255) / 256;
    l.yv_n = (l.y_elems + 255) / 256;
    const int xw = (g.kh * l.Wp * g.cin + 3) & ~3;
    l.ones_off = xw;
    l.zero_off = xw + ((P + 3) & ~3);
    l.yl_off = l.zero_off + 4;
    l.lds_words = l.yl_off + P * l.coutp;
    size_t lds_bytes = (size_t)l.lds_words * 4;
    if (bf16_mfma) {
      WgBf& q = p.q;
      q.Pq = (P + 15) & ~15;
      q.Wq = (q.Pq - 1) * g.stride + (g.kw - 1) * g.dil + 1;
      q.ones_b = (g.kh * q.Wq * g.cin * 2 + 15) & ~15;
      q.zero_b = q.ones_b + 8;
      q.yl_b = q.ones_b + 16;
      q.lds_bytes = q.yl_b + q.Pq * g.cout * 2;
      lds_bytes = (size_t)q.lds_bytes;
    }
    // (the last three bounds are the field widths of the packed staging maps)
    if (l.xv_n <= kXvMax && l.yv_n <= kYvMax && lds_bytes <= 64 * 1024 && g.cin <= 4096 && g.cout <= 4096 &&
        l.Wp < 2048 && (g.kh - 1) * g.dil < 256) {
      p.use_lds = true;
      break;
    }
  }
  static const bool force_direct = getenv("SSMI_WG_DIRECT") != nullptr;      // A/B switch: operands straight from global memory
  if (force_direct) p.use_lds = false;
  if (bf16_mfma && !(p.vec && g.kh < 256)) p.use_lds = false;      // 8-byte staging only; ky travels in 8 bits of the map
  if (!p.use_lds && p.m16) {   // the direct kernel only has 32x32 tiles
    p.m16 = false;
    tiles(32);
  }

  const int spb = p.use_lds ? 1 : 4 / a.wps;
  // About one workgroup per CU per tile group.  Every strip writes a full [K_aug x cout] partial sum that the reduce
  // kernel reads back, which at 512 strips is more traffic than the operands themselves; the weight gradients run
  // beside the data-gradient chain on their own stream, so they do not need to fill the chip on their own
  // (ENet step: 512 strips 11.12 ms, 256: 10.88, 128: 11.56).  SSMI_WG_STRIPS overrides for measurements.
  // The scalar-staging variant (channel counts that are no multiple of four: a network's first convolution) moves four
  // bytes per load and is the last weight gradient of a backward pass, when the chip is otherwise empty: four
  // workgroups per CU (init 3x3/2 3->13 at batch 16: 104 us with 256 strips, 65 with 512, 52 with 1024).
  static const int strips_env = [] { const char* e = getenv("SSMI_WG_STRIPS"); return e ? atoi(e) : 0; }();
  const int strips_lds = strips_env ? strips_env : (p.vec ? 256 : 1024);
  int target = ((p.use_lds ? strips_lds : 256) * spb) / p.groups;
  if (target < spb) target = spb;
  if (target > R) target = R;
  a.rows_per_strip = (R + target - 1) / target;
  a.nstrips = (R + a.rows_per_strip - 1) / a.rows_per_strip;
  p.blocks_x = (a.nstrips + spb - 1) / spb;
  p.bytes = (size_t)a.nstrips * a.K_aug * g.cout * sizeof(float);
  return p;
}


// ------------------------------------------------------------------------------------------------
// Weight gradient of a 1x1, stride-1 convolution without bias, float32 storage, channel counts 16 / 32 / 64 / 128:
//   dW[ci][co] = sum over pixels p of x[p][ci] * dy[p][co]
// A 1x1 layer has no halo and no tap reuse, so nothing is gained by staging its operands in LDS: both are loaded straight
// into the operand layout of v_mfma_f32_16x16x4_f32, which sums over FOUR pixels per instruction.  Lane (i = l & 15,
// kk = l >> 4) of a wave loads, for pixel 4 * group + kk, the CI = cin / 16 consecutive input channels i * CI .. and the
// CO = cout / 16 consecutive output-gradient channels i * CO ..: the 16 lanes of one pixel read its whole channel vector
// with 4- to 16-byte loads.  Matrix instruction (mt, nt) takes element mt of the x registers as A and element nt of the
// dy registers as B, i.e. tile (mt, nt) holds the rows ci = 16-row index * CI + mt and the columns co = column * CO + nt
// of dW (a permutation of the rows and columns that only the final store has to know).  A wave owns every tile
// (CI * CO accumulators of four registers) and walks its share of the workgroup's pixels U groups at a time through two
// register sets (the next round's loads in flight under the current round's products); the four waves' sums are combined through LDS in a fixed order and the
// workgroup writes ONE partial matrix for the strip reduction.  PRE: x is the raw input of a BatchNormalization + PReLU
// whose output is the convolution's input; the lane applies it to its CI channels (arithmetic of bn_act_fwd_kernel).
// Pixels past the workgroup's range contribute nothing: both loads return zero through the buffer bounds check.
// ------------------------------------------------------------------------------------------------
template <int N> struct VecLoad;
template <> struct VecLoad<1> {
  static __device__ __forceinline__ void ld(__amdgpu_buffer_rsrc_t r, unsigned off, float* v) { v[0] = bload(r, off); }
};
template <> struct VecLoad<2> {
  static __device__ __forceinline__ void ld(__amdgpu_buffer_rsrc_t r, unsigned off, float* v) {
    const u32x2 w = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
    // (the elements go through scalars: a bit cast applied directly to a vector-element expression, bit_cast(float, w[1]),
    // is compiled by hipcc 7.2 as if it named element 0, and the load shrinks to one dword)
    const unsigned lo = w[0], hi = w[1];
    v[0] = __builtin_bit_cast(float, lo); v[1] = __builtin_bit_cast(float, hi);
  }
};
template <> struct VecLoad<4> {
  static __device__ __forceinline__ void ld(__amdgpu_buffer_rsrc_t r, unsigned off, float* v) {
    const f32x4 w = bload4(r, off);
    v[0] = w[0]; v[1] = w[1]; v[2] = w[2]; v[3] = w[3];
  }
};
template <> struct VecLoad<8> {
  static __device__ __forceinline__ void ld(__amdgpu_buffer_rsrc_t r, unsigned off, float* v) {
    const f32x4 w0 = bload4(r, off), w1 = bload4(r, off + 16u);
    v[0] = w0[0]; v[1] = w0[1]; v[2] = w0[2]; v[3] = w0[3]; v[4] = w1[0]; v[5] = w1[1]; v[6] = w1[2]; v[7] = w1[3];
  }
};

template <int CI, int CO, bool PRE, int U>
__global__ __launch_bounds__(256) void conv_bwd_weight_1x1(const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ part, int npix, int x_ld, int y_ld,
                                                           int groups_per_wg, PreBn pb) {
  extern __shared__ f32x4 wg1_sm[];                 // [CI * CO][64] accumulators of one wave at a time
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kk = lane >> 4;
  // groups of four pixels: this workgroup owns [g0, g1), wave w takes g0 + w, g0 + w + 4, ...  Both buffer descriptors END at
  // the workgroup's last pixel, so a round that reaches past it reads zeros without any per-lane test (a `valid ? offset :
  // out-of-range` select in front of a 4-byte load is compiled into a branch with a full memory-counter drain per load).
  const int ngroups = (npix + 3) >> 2;
  const int g0 = blockIdx.x * groups_per_wg;
  int g1 = g0 + groups_per_wg;
  if (g1 > ngroups) g1 = ngroups;
  const int pend = 4 * g1 < npix ? 4 * g1 : npix;
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, (unsigned)((int64_t)pend * x_ld * 4));
  const __amdgpu_buffer_rsrc_t rd = make_rsrc(dy, (unsigned)((int64_t)pend * y_ld * 4));
  float pm[CI], pis[CI], pg[CI], pbt[CI], pa[CI];
  bool act = false;
  if constexpr (PRE) {
    act = pb.alpha != nullptr;
#pragma unroll
    for (int t = 0; t < CI; ++t) {
      const int c = i * CI + t;
      pm[t] = pb.mean[c]; pis[t] = pb.invstd[c]; pg[t] = pb.gamma[c]; pbt[t] = pb.beta[c];
      pa[t] = act ? pb.alpha[c] : 0.f;
    }
  }
  f32x4 acc[CI][CO];
#pragma unroll
  for (int m = 0; m < CI; ++m)
#pragma unroll
    for (int n = 0; n < CO; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  // two register sets: the loads of round r + 1 are in flight while round r is multiplied (one wave per SIMD is the usual
  // occupancy of these launches, so nothing else would hide the memory latency)
  float xa[U][CI], da[U][CO], xb[U][CI], db[U][CO];
  auto issue = [&](float (&xv)[U][CI], float (&dv)[U][CO], int gb) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned p = (unsigned)(4 * (gb + 4 * u) + kk);
      VecLoad<CI>::ld(rx, (p * (unsigned)x_ld + (unsigned)(i * CI)) * 4u, xv[u]);
      VecLoad<CO>::ld(rd, (p * (unsigned)y_ld + (unsigned)(i * CO)) * 4u, dv[u]);
    }
  };
  auto consume = [&](float (&xv)[U][CI], float (&dv)[U][CO]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if constexpr (PRE) {
#pragma unroll
        for (int t = 0; t < CI; ++t) xv[u][t] = pre_bn_act(xv[u][t], pm[t], pis[t], pg[t], pbt[t], pa[t], act);
      }
#pragma unroll
      for (int m = 0; m < CI; ++m)
#pragma unroll
        for (int n = 0; n < CO; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[u][m], dv[u][n], acc[m][n], 0, 0, 0);
    }
  };
  int gb = g0 + wave;
  issue(xa, da, gb);
  for (; gb < g1; gb += 8 * U) {
    issue(xb, db, gb + 4 * U);          // groups past g1 read nothing (out-of-range offsets)
    consume(xa, da);
    issue(xa, da, gb + 8 * U);
    consume(xb, db);
  }
  // waves 1..3 hand their sums to wave 0, one after the other (fixed order of additions)
  for (int w = 1; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int m = 0; m < CI; ++m)
#pragma unroll
        for (int n = 0; n < CO; ++n) wg1_sm[(m * CO + n) * 64 + lane] = acc[m][n];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int m = 0; m < CI; ++m)
#pragma unroll
        for (int n = 0; n < CO; ++n) acc[m][n] += wg1_sm[(m * CO + n) * 64 + lane];
    }
    __syncthreads();
  }
  if (wave != 0) return;
  constexpr int cout = 16 * CO;
  float* pbase = part + (size_t)blockIdx.x * (16 * CI) * cout;
#pragma unroll
  for (int m = 0; m < CI; ++m)
#pragma unroll
    for (int n = 0; n < CO; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ci = (4 * kk + r) * CI + m, co = i * CO + n;
        pbase[ci * cout + co] = acc[m][n][r];
      }
}

struct Wg1x1Plan {
  bool use;
  int ci, co, groups_per_wg, nstrips;
  size_t bytes;
};
static Wg1x1Plan plan_wg_1x1(const Geo& g, bool with_bias) {
  Wg1x1Plan p{};
  static const bool off = getenv("SSMI_NO_WG_1X1") != nullptr;      // measurement switch
  auto okc = [](int c) { return c == 16 || c == 32 || c == 64 || c == 128; };
  p.use = !off && !with_bias && g.kh == 1 && g.kw == 1 && g.stride == 1 && okc(g.cin) && okc(g.cout) && g.x_ld % 4 == 0 &&
          g.y_ld % 4 == 0 && g.ho == g.h && g.wo == g.w && !(g.cin == 128 && g.cout == 128);
  if (!p.use) return p;
  p.ci = g.cin / 16;
  p.co = g.cout / 16;
  static const int strips_env = [] { const char* e = getenv("SSMI_WG1X1_STRIPS"); return e ? atoi(e) : 0; }();
  // one workgroup per CU: with two, the partial sums double and the main chain beside these launches loses more than
  // they gain (ENet step 8.83 ms with 256, 8.95 with 384, 8.91-8.96 with 512)
  const int target = strips_env ? strips_env : 256;
  const int64_t npix = (int64_t)g.n * g.h * g.w;
  const int ngroups = (int)((npix + 3) / 4);
  int gpw = (ngroups + target - 1) / target;
  gpw = (gpw + 3) & ~3;                 // whole rounds of the four waves
  if (gpw < 4) gpw = 4;
  p.groups_per_wg = gpw;
  p.nstrips = (ngroups + gpw - 1) / gpw;
  p.bytes = (size_t)p.nstrips * g.cin * g.cout * sizeof(float);
  return p;
}

static bool launch_wg_1x1(const Geo& g, const Wg1x1Plan& p, const float* x, const float* dy, float* part, hipStream_t st,
                          const PreBn* pre) {
  const int npix = g.n * g.h * g.w;
  const size_t lds = (size_t)p.ci * p.co * 64 * sizeof(f32x4);
  const PreBn pb = pre ? *pre : PreBn{};
#define L1(CI, CO, U)                                                                                                           \
  do {                                                                                                                          \
    if (pre) hipLaunchKernelGGL((conv_bwd_weight_1x1<CI, CO, true, U>), dim3(p.nstrips), dim3(256), lds, st, x, dy, part, npix, \
                                g.x_ld, g.y_ld, p.groups_per_wg, pb);                                                           \
    else hipLaunchKernelGGL((conv_bwd_weight_1x1<CI, CO, false, U>), dim3(p.nstrips), dim3(256), lds, st, x, dy, part, npix,    \
                            g.x_ld, g.y_ld, p.groups_per_wg, pb);                                                               \
    return true;                                                                                                                \
  } while (0)
  const int key = p.ci * 16 + p.co;
  switch (key) {
    case 1 * 16 + 1: L1(1, 1, 8);
    case 1 * 16 + 2: L1(1, 2, 8);
    case 1 * 16 + 4: L1(1, 4, 8);
    case 1 * 16 + 8: L1(1, 8, 4);
    case 2 * 16 + 1: L1(2, 1, 8);
    case 2 * 16 + 2: L1(2, 2, 8);
    case 2 * 16 + 4: L1(2, 4, 4);
    case 2 * 16 + 8: L1(2, 8, 4);
    case 4 * 16 + 1: L1(4, 1, 8);
    case 4 * 16 + 2: L1(4, 2, 4);
    case 4 * 16 + 4: L1(4, 4, 4);
    case 4 * 16 + 8: L1(4, 8, 2);
    case 8 * 16 + 1: L1(8, 1, 4);
    case 8 * 16 + 2: L1(8, 2, 4);
    case 8 * 16 + 4: L1(8, 4, 2);
    default: return false;
  }
#undef L1
}

static WgPatchPlan wg_patch_plan(const Geo& g, bool with_bias) {
  static const bool off = getenv("SSMI_NO_BF16_WGRAD") != nullptr || getenv("SSMI_NO_BF16_MFMA") != nullptr;
  if (off) return WgPatchPlan{};
  return plan_wg_patch_bf16(g.n, g.h, g.w, g.cin, g.cout, g.kh, g.kw, g.stride, g.dil, g.pad_t, g.pad_l, g.ho, g.wo, g.x_ld, g.y_ld, with_bias);
}

static size_t wg_ws_bytes(const Geo& g) {
  const TiledPlan t = plan_tiled(g, true);
  const size_t b0 = plan_wg(g, false).bytes, b1 = plan_wg(g, true).bytes;
  const size_t b2 = plan_wg(g, true, true).bytes;      // the bf16 matrix-instruction variant tiles (and so strips) differently
  const size_t b3 = plan_wg_1x1(g, false).bytes;      // direct 1x1 kernel (float32 storage, no bias)
  const size_t b = std::max(std::max(b0, b3), std::max(b1, b2));
  const WgPatchPlan wp = wg_patch_plan(g, true);      // bfloat16 storage only (the activation type is not known here)
  const size_t b4 = wp.use ? wp.bytes : 0;
  return std::max(t.use && t.bytes > b ? t.bytes : b, b4);      // the strip kernels remain the fallback for unaligned pointers
}

// layers whose weight gradient can take act(BN(x)) of a raw x (PRE instantiations of conv_bwd_weight_lds: one tile per wave,
// short patch / long dy chunk staging class)
static bool wg_prebn_ok(const Geo& g, bool with_bias, bool bf16_mfma = false) {
  if (!(g.kh == 1 && g.kw == 1 && g.stride == 1 && g.cin % 4 == 0 && 256 % (g.cin / 4) == 0)) return false;
  if (plan_tiled(g, with_bias).use) return false;
  const WgPlan p = plan_wg(g, with_bias, bf16_mfma);
  return p.use_lds && p.vec && p.tpw == 1 && p.l.xv_n <= 2 && p.l.yv_n <= 8;
}

template <typename T>
static ssmi_status launch_bwd_weight(const Geo& g, const T* x, const T* dy, float* dw, float* dbias, void* ws,
                                     size_t ws_bytes, hipStream_t st, const PreBn* pre = nullptr) {
  Sizes sz;
  ssmi_status e = make_sizes(g, &sz);
  if (e) return e;
  if constexpr (sizeof(T) == 4) {
    // 1x1 layers: operands straight from global memory in the matrix instruction's layout
    const Wg1x1Plan q = plan_wg_1x1(g, dbias != nullptr);
    if (q.use && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0 && !plan_tiled(g, false).use) {
      SSMI_REQUIRE(ws != nullptr && ws_bytes >= q.bytes, SSMI_E_WORKSPACE, "conv2d_bwd_weight: workspace too small (%zu < %zu)",
                   ws_bytes, q.bytes);
      float* part = reinterpret_cast<float*>(ws);
      if (launch_wg_1x1(g, q, x, dy, part, st, pre)) {
        SSMI_LAUNCH_CHECK();
        launch_wg_reduce(part, q.nstrips, g.cin, g.cin, g.cout, dw, nullptr, st);
        SSMI_LAUNCH_CHECK();
        return SSMI_OK;
      }
    }
  }
  if (pre) {
    if constexpr (sizeof(T) == 2) {
      SSMI_REQUIRE(wg_prebn_ok(g, dbias != nullptr, true) && (reinterpret_cast<uintptr_t>(x) & 15) == 0, SSMI_E_UNSUPPORTED,
                   "conv2d_bwd_weight (normalise on load): layer shape not supported");
      const WgPlan p = plan_wg(g, dbias != nullptr, true);
      SSMI_REQUIRE(ws != nullptr && ws_bytes >= p.bytes, SSMI_E_WORKSPACE, "conv2d_bwd_weight: workspace too small (%zu < %zu)",
                   ws_bytes, p.bytes);
      float* part = reinterpret_cast<float*>(ws);
      dim3 grid((unsigned)p.blocks_x, (unsigned)p.groups, 1);
      hipLaunchKernelGGL((conv_bwd_weight_lds_bf16<1, 2, 8, true>), grid, dim3(256), (size_t)p.q.lds_bytes, st, g, sz, p.a, p.l, p.q, x, dy,
                         part, *pre);
      SSMI_LAUNCH_CHECK();
      launch_wg_reduce(part, p.a.nstrips, p.a.K, p.a.K_aug, g.cout, dw, dbias, st);
      SSMI_LAUNCH_CHECK();
      return SSMI_OK;
    }
    if constexpr (sizeof(T) == 4) {
      SSMI_REQUIRE(wg_prebn_ok(g, dbias != nullptr) && (reinterpret_cast<uintptr_t>(x) & 15) == 0, SSMI_E_UNSUPPORTED,
                   "conv2d_bwd_weight (normalise on load): layer shape not supported");
      const WgPlan p = plan_wg(g, dbias != nullptr);
      SSMI_REQUIRE(ws != nullptr && ws_bytes >= p.bytes, SSMI_E_WORKSPACE, "conv2d_bwd_weight: workspace too small (%zu < %zu)",
                   ws_bytes, p.bytes);
      float* part = reinterpret_cast<float*>(ws);
      dim3 grid((unsigned)p.blocks_x, (unsigned)p.groups, 1);
      const size_t lds = (size_t)p.l.lds_words * 4;
      if (p.m16)
        hipLaunchKernelGGL((conv_bwd_weight_lds<T, 1, true, true, 2, 8, true>), grid, dim3(256), lds, st, g, sz, p.a, p.l, x, dy, part, *pre);
      else
        hipLaunchKernelGGL((conv_bwd_weight_lds<T, 1, true, false, 2, 8, true>), grid, dim3(256), lds, st, g, sz, p.a, p.l, x, dy, part, *pre);
      SSMI_LAUNCH_CHECK();
      launch_wg_reduce(part, p.a.nstrips, p.a.K, p.a.K_aug, g.cout, dw, dbias, st);
      SSMI_LAUNCH_CHECK();
      return SSMI_OK;
    }
  }
  if constexpr (sizeof(T) == 2) {
    // wide 3x3 layers: halo patches staged once per tile, all nine taps from LDS (conv_tiled.hip)
    const WgPatchPlan q = wg_patch_plan(g, dbias != nullptr);
    if (q.use && !pre && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0) {
      SSMI_REQUIRE(ws != nullptr && ws_bytes >= q.bytes, SSMI_E_WORKSPACE, "conv2d_bwd_weight: workspace too small (%zu < %zu)",
                   ws_bytes, q.bytes);
      SSMI_REQUIRE(q.bytes < 0x7FFFFFFFULL, SSMI_E_UNSUPPORTED, "conv2d_bwd_weight: partial sums of 2 GiB or more");
      float* part = reinterpret_cast<float*>(ws);
      {
        const double M = (double)g.n * g.ho * g.wo;
        ProfileScope prof(st, M * (g.cin + g.cout) * 2.0 + (double)q.K_aug * g.cout * 4.0, conv_flops(g));
        set_last_kernel("conv3x3_wgrad_patch_bf16");
        e = launch_wg_patch_bf16(q, g.n, g.h, g.w, g.cin, g.cout, g.x_ld, g.y_ld, x, dy, part, st);
        if (e) return e;
      }
      launch_wg_reduce(part, q.shares, q.K, q.K_aug, g.cout, dw, dbias, st);
      SSMI_LAUNCH_CHECK();
      return SSMI_OK;
    }
  }
  const TiledPlan t = plan_tiled(g, dbias != nullptr);
  if (t.use && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0) {
    SSMI_REQUIRE(ws != nullptr && ws_bytes >= t.bytes, SSMI_E_WORKSPACE, "conv2d_bwd_weight: workspace too small (%zu < %zu)",
                 ws_bytes, t.bytes);
    SSMI_REQUIRE(t.bytes < 0x7FFFFFFFULL, SSMI_E_UNSUPPORTED, "conv2d_bwd_weight: partial sums of 2 GiB or more");
    float* part = reinterpret_cast<float*>(ws);
    dim3 grid((unsigned)t.ktiles, (unsigned)t.ctiles, (unsigned)t.shares);
    bool on_bf16_mfma = false;
    if constexpr (sizeof(T) == 2) {
      static const bool off = getenv("SSMI_NO_BF16_WGRAD") != nullptr || getenv("SSMI_NO_BF16_MFMA") != nullptr;
      if (!off) {
        on_bf16_mfma = true;
        const int pps = t.chunks_per_share * 32;
        if (t.wk == 2)
          hipLaunchKernelGGL((conv_bwd_weight_tiled_bf16<2, 2>), grid, dim3(256), (size_t)64 * (2 * (128 * 2 + 64)), st, g, sz, t.K, t.K_aug,
                             pps, x, dy, part);
        else
          hipLaunchKernelGGL((conv_bwd_weight_tiled_bf16<4, 1>), grid, dim3(256), (size_t)64 * ((256 * 2 + 64) + (64 * 2 + 64)), st, g, sz,
                             t.K, t.K_aug, pps, x, dy, part);
      }
    }
    if (on_bf16_mfma) {
    } else if (t.wk == 2)
      hipLaunchKernelGGL((conv_bwd_weight_tiled<T, 2, 2>), grid, dim3(256), (size_t)32 * 256 * 4, st, g, sz, t.K, t.K_aug,
                         t.chunks_per_share, x, dy, part);
    else
      hipLaunchKernelGGL((conv_bwd_weight_tiled<T, 4, 1>), grid, dim3(256), (size_t)32 * 320 * 4, st, g, sz, t.K, t.K_aug,
                         t.chunks_per_share, x, dy, part);
    SSMI_LAUNCH_CHECK();
    launch_wg_reduce(part, t.shares, t.K, t.K_aug, g.cout, dw, dbias, st);
    SSMI_LAUNCH_CHECK();
    return SSMI_OK;
  }
  WgPlan p = plan_wg(g, dbias != nullptr);
  bool bf16_mfma = false;
  if constexpr (sizeof(T) == 2) {
    static const bool off = getenv("SSMI_NO_BF16_WGRAD") != nullptr || getenv("SSMI_NO_BF16_MFMA") != nullptr;
    if (!off) {
      const WgPlan pb = plan_wg(g, dbias != nullptr, true);
      if (pb.use_lds) { p = pb; bf16_mfma = true; }
    }
  }
  SSMI_REQUIRE(ws != nullptr && ws_bytes >= p.bytes, SSMI_E_WORKSPACE, "conv2d_bwd_weight: workspace too small (%zu < %zu)",
               ws_bytes, p.bytes);
  float* part = reinterpret_cast<float*>(ws);
  dim3 grid((unsigned)p.blocks_x, (unsigned)p.groups, 1);
  if (bf16_mfma) {
    if constexpr (sizeof(T) == 2) {
      const int cls = (p.l.xv_n <= 8 && p.l.yv_n <= 2) ? 0 : (p.l.xv_n <= 2 && p.l.yv_n <= 8) ? 1 : 2;
#define LB(TPW, XV, YV) hipLaunchKernelGGL((conv_bwd_weight_lds_bf16<TPW, XV, YV>), grid, dim3(256), (size_t)p.q.lds_bytes, st, g, sz, p.a, p.l, p.q, x, dy, part)
#define LT(TPW) do { if (cls == 0) LB(TPW, 8, 2); else if (cls == 1) LB(TPW, 2, 8); else LB(TPW, kXvMax, kYvMax); } while (0)
      if (p.tpw == 1) LT(1); else if (p.tpw == 2) LT(2); else LT(3);
#undef LT
#undef LB
    }
  } else if (p.use_lds) {
    const size_t lds = (size_t)p.l.lds_words * 4;
    // slot classes (x patch, dy chunk): most layers stage a long patch and a short dy chunk (the widest patches -
    // dilation 16, 5x1, the 2x2 stride-2 reductions - need all twelve patch slots), the 1x1 expansions the reverse;
    // everything else takes the full-size class
    const int cls = (p.l.xv_n <= 8 && p.l.yv_n <= 2) ? 0 : (p.l.xv_n <= 2 && p.l.yv_n <= 8) ? 1 : (p.l.yv_n <= 2) ? 3 : 2;
#define LC(TPW, V, M, XV, YV) hipLaunchKernelGGL((conv_bwd_weight_lds<T, TPW, V, M, XV, YV>), grid, dim3(256), lds, st, g, sz, p.a, p.l, x, dy, part)
#define LL(TPW, V, M) do { if (cls == 0) LC(TPW, V, M, 8, 2); else if (cls == 1) LC(TPW, V, M, 2, 8); else if (cls == 3) LC(TPW, V, M, kXvMax, 2); else LC(TPW, V, M, kXvMax, kYvMax); } while (0)
#define LT(V, M) do { if (p.tpw == 1) LL(1, V, M); else if (p.tpw == 2) LL(2, V, M); else LL(3, V, M); } while (0)
    if (p.vec) { if (p.m16) LT(true, true); else LT(true, false); }
    else { if (p.m16) LT(false, true); else LT(false, false); }
#undef LT
#undef LL
#undef LC
  } else {
#define L(TPW) hipLaunchKernelGGL((conv_bwd_weight_kernel<T, TPW>), grid, dim3(256), 0, st, g, sz, p.a, x, dy, part)
    if (p.tpw == 1) L(1); else if (p.tpw == 2) L(2); else L(3);
#undef L
  }
  SSMI_LAUNCH_CHECK();
  launch_wg_reduce(part, p.a.nstrips, p.a.K, p.a.K_aug, g.cout, dw, dbias, st);
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

}  // namespace ssmi

using namespace ssmi;

extern "C" {

ssmi_status ssmi_same_pad(int32_t in, int32_t k, int32_t stride, int32_t dil, int32_t* out, int32_t* pad_before) {
  SSMI_REQUIRE(in > 0 && k > 0 && stride > 0 && dil > 0 && out && pad_before, SSMI_E_INVALID_ARG, "same_pad: bad argument");
  const int k_eff = (k - 1) * dil + 1;
  const int o = (in + stride - 1) / stride;
  int total = (o - 1) * stride + k_eff - in;
  if (total < 0) total = 0;
  *out = o;
  *pad_before = total / 2;
  return SSMI_OK;
}

ssmi_status ssmi_conv2d_fwd(const ssmi_conv_desc* d, const void* x, const float* w, const float* bias, void* y,
                            void* stream) {
  ssmi_status s = check_desc(d);
  if (s) return s;
  SSMI_REQUIRE(x && w && y, SSMI_E_INVALID_ARG, "conv2d_fwd: NULL tensor");
  return dispatch_dtype(d->dtype, [&](auto tag) {
    using T = decltype(tag);
    return launch_fwd<T>(to_geo(d), static_cast<const T*>(x), w, bias, static_cast<T*>(y), as_stream(stream));
  });
}

ssmi_status ssmi_conv2d_fwd_bn_act_infer(const ssmi_conv_desc* d, const void* x, const float* w, const float* bias,
                                         const float* mean, const float* invstd, const float* gamma, const float* beta,
                                         int32_t act, const float* alpha, float slope, void* y, void* stream) {
  ssmi_status s = check_desc(d);
  if (s) return s;
  SSMI_REQUIRE(x && w && y && mean && invstd && gamma && beta, SSMI_E_INVALID_ARG, "conv2d_fwd_bn_act_infer: NULL tensor");
  SSMI_REQUIRE(act == SSMI_ACT_NONE || act == SSMI_ACT_LEAKY || (act == SSMI_ACT_PRELU && alpha), SSMI_E_INVALID_ARG,
               "conv2d_fwd_bn_act_infer: activation %d", act);
  return ssmi_conv2d_fwd_bn_add_act_infer(d, x, w, bias, mean, invstd, gamma, beta, nullptr, 0, act, alpha, slope, y, stream);
}

ssmi_status ssmi_conv2d_fwd_bn_add_act_infer(const ssmi_conv_desc* d, const void* x, const float* w, const float* bias,
                                             const float* mean, const float* invstd, const float* gamma, const float* beta,
                                             const void* skip, int32_t skip_ld, int32_t act, const float* alpha,
                                             float slope, void* y, void* stream) {
  ssmi_status s = check_desc(d);
  if (s) return s;
  SSMI_REQUIRE(x && w && y && mean && invstd && gamma && beta, SSMI_E_INVALID_ARG, "conv2d_fwd_bn_add_act_infer: NULL tensor");
  SSMI_REQUIRE(act == SSMI_ACT_NONE || act == SSMI_ACT_LEAKY || (act == SSMI_ACT_PRELU && alpha), SSMI_E_INVALID_ARG,
               "conv2d_fwd_bn_add_act_infer: activation %d", act);
  SSMI_REQUIRE(!skip || (skip_ld >= d->cout && (int64_t)d->n * d->ho * d->wo * skip_ld * 4 < 0x7FFFFFFFLL), SSMI_E_INVALID_ARG,
               "conv2d_fwd_bn_add_act_infer: skip tensor stride %d for %d channels", skip_ld, d->cout);
  BnBwdArgs bn{};
  bn.mean = mean; bn.invstd = invstd; bn.gamma = gamma; bn.beta = beta; bn.alpha = alpha;
  bn.slope = slope; bn.act = act; bn.hw = 1; bn.skip = skip; bn.skip_ld = skip ? skip_ld : 0;
  return dispatch_dtype(d->dtype, [&](auto tag) {
    using T = decltype(tag);
    return launch_fwd<T>(to_geo(d), static_cast<const T*>(x), w, bias, static_cast<T*>(y), as_stream(stream), nullptr, nullptr, bn);
  });
}

size_t ssmi_conv2d_fwd_bn_stats_workspace_bytes(const ssmi_conv_desc* d) {
  if (check_desc(d) != SSMI_OK) return 0;
  const int64_t rows = ceil_div64((int64_t)d->n * d->ho * d->wo, 128) + 1024;   // direct kernel: one row per 128 pixels
  return (size_t)rows * 2 * d->cout * sizeof(float);
}

ssmi_status ssmi_conv2d_fwd_bn_stats(const ssmi_conv_desc* d, const void* x, const float* w, const float* bias, void* y,
                                     float eps, float momentum, int32_t unbiased_moving, float* moving_mean,
                                     float* moving_var, float* save_mean, float* save_invstd, void* ws, size_t ws_bytes,
                                     void* stream) {
  ssmi_status s = check_desc(d);
  if (s) return s;
  SSMI_REQUIRE(x && w && y && save_mean && save_invstd, SSMI_E_INVALID_ARG, "conv2d_fwd_bn_stats: NULL tensor");
  SSMI_REQUIRE(ws && ws_bytes >= ssmi_conv2d_fwd_bn_stats_workspace_bytes(d), SSMI_E_WORKSPACE,
               "conv2d_fwd_bn_stats: workspace too small");
  int rows = 0;
  float* part = reinterpret_cast<float*>(ws);
  s = dispatch_dtype(d->dtype, [&](auto tag) {
    using T = decltype(tag);
    return launch_fwd<T>(to_geo(d), static_cast<const T*>(x), w, bias, static_cast<T*>(y), as_stream(stream), part, &rows);
  });
  if (s) return s;
  return launch_bn_stats_final(part, rows, d->cout, (int64_t)d->n * d->ho * d->wo, eps, momentum, unbiased_moving,
                               moving_mean, moving_var, save_mean, save_invstd, as_stream(stream));
}

int32_t ssmi_conv2d_prebn_supported(const ssmi_conv_desc* d, int32_t with_bias) {
  if (check_desc(d) != SSMI_OK) return 0;
  const Geo g = to_geo(d);
  if (d->dtype == SSMI_F32) return fwd_prebn_ok(g) && wg_prebn_ok(g, with_bias != 0);
  // bfloat16: whole 16-channel blocks (the bfloat16 matrix instructions), the weight gradient on the same instructions
  static const bool no_mfma = getenv("SSMI_NO_BF16_WGRAD") != nullptr || getenv("SSMI_NO_BF16_MFMA") != nullptr;
  return !no_mfma && fwd_prebn_ok(g) && g.cin % 16 == 0 && wg_prebn_ok(g, with_bias != 0, true);
}

ssmi_status ssmi_conv2d_fwd_bn_stats_prebn(const ssmi_conv_desc* d, const void* x_raw, const ssmi_bn_params* pre,
                                           const float* pre_alpha, const float* w, const float* bias, void* y, float eps,
                                           float momentum, int32_t unbiased_moving, float* moving_mean, float* moving_var,
                                           float* save_mean, float* save_invstd, void* ws, size_t ws_bytes, void* stream) {
  ssmi_status s = check_desc(d);
  if (s) return s;
  SSMI_REQUIRE(x_raw && pre && pre->mean && pre->invstd && pre->gamma && pre->beta && w && y && save_mean && save_invstd,
               SSMI_E_INVALID_ARG, "conv2d_fwd_bn_stats_prebn: NULL argument");
  SSMI_REQUIRE(ws && ws_bytes >= ssmi_conv2d_fwd_bn_stats_workspace_bytes(d), SSMI_E_WORKSPACE,
               "conv2d_fwd_bn_stats_prebn: workspace too small");
  int rows = 0;
  float* part = reinterpret_cast<float*>(ws);
  BnBwdArgs bn{};
  bn.pre[0] = pre->mean; bn.pre[1] = pre->invstd; bn.pre[2] = pre->gamma; bn.pre[3] = pre->beta; bn.pre[4] = pre_alpha;
  s = dispatch_dtype(d->dtype, [&](auto tag) {
    using T = decltype(tag);
    return launch_fwd<T>(to_geo(d), static_cast<const T*>(x_raw), w, bias, static_cast<T*>(y), as_stream(stream), part, &rows, bn);
  });
  if (s) return s;
  return launch_bn_stats_final(part, rows, d->cout, (int64_t)d->n * d->ho * d->wo, eps, momentum, unbiased_moving,
                               moving_mean, moving_var, save_mean, save_invstd, as_stream(stream));
}

ssmi_status ssmi_conv2d_bwd_weight_prebn(const ssmi_conv_desc* d, const void* x_raw, const ssmi_bn_params* pre,
                                         const float* pre_alpha, const void* dy, float* dw, float* dbias, void* ws,
                                         size_t ws_bytes, void* stream) {
  ssmi_status s = check_desc(d);
  if (s) return s;
  SSMI_REQUIRE(x_raw && pre && pre->mean && pre->invstd && pre->gamma && pre->beta && dy && dw, SSMI_E_INVALID_ARG,
               "conv2d_bwd_weight_prebn: NULL argument");
  PreBn pb{pre->mean, pre->invstd, pre->gamma, pre->beta, pre_alpha};
  return dispatch_dtype(d->dtype, [&](auto tag) {
    using T = decltype(tag);
    return launch_bwd_weight<T>(to_geo(d), static_cast<const T*>(x_raw), static_cast<const T*>(dy), dw, dbias, ws, ws_bytes,
                                as_stream(stream), &pb);
  });
}

ssmi_status ssmi_conv2d_bwd_data(const ssmi_conv_desc* d, const void* dy, const float* w, const void* add, void* dx,
                                 void* stream) {
  ssmi_status s = check_desc(d);
  if (s) return s;
  SSMI_REQUIRE(dy && w && dx, SSMI_E_INVALID_ARG, "conv2d_bwd_data: NULL tensor");
  return dispatch_dtype(d->dtype, [&](auto tag) {
    using T = decltype(tag);
    return launch_bwd_data<T>(to_geo(d), static_cast<const T*>(dy), w, static_cast<const T*>(add), nullptr, static_cast<T*>(dx),
                              as_stream(stream));
  });
}

size_t ssmi_conv2d_bwd_data_workspace_bytes(const ssmi_conv_desc* d) {
  if (check_desc(d) != SSMI_OK) return 0;
  return dgrad_wt_bytes(to_geo(d));
}

ssmi_status ssmi_conv2d_bwd_data_ws(const ssmi_conv_desc* d, const void* dy, const float* w, const void* add, void* dx,
                                    void* ws, size_t ws_bytes, void* stream) {
  ssmi_status s = check_desc(d);
  if (s) return s;
  SSMI_REQUIRE(dy && w && dx, SSMI_E_INVALID_ARG, "conv2d_bwd_data_ws: NULL tensor");
  const size_t need = ssmi_conv2d_bwd_data_workspace_bytes(d);
  SSMI_REQUIRE(need == 0 || (ws && ws_bytes >= need), SSMI_E_WORKSPACE, "conv2d_bwd_data_ws: workspace too small");
  return dispatch_dtype(d->dtype, [&](auto tag) {
    using T = decltype(tag);
    return launch_bwd_data<T>(to_geo(d), static_cast<const T*>(dy), w, static_cast<const T*>(add), nullptr, static_cast<T*>(dx),
                              as_stream(stream), BnBwdArgs{}, nullptr, need ? static_cast<float*>(ws) : nullptr);
  });
}

int32_t ssmi_conv2d_pack_usage(const ssmi_conv_desc* d) {
  if (check_desc(d) != SSMI_OK || d->dtype != SSMI_BF16) return 0;
  const Geo g = to_geo(d);
  return (fwd_pack_ok(g) ? 1 : 0) | (dgrad_pack_ok(g) ? 2 : 0);
}

ssmi_status ssmi_conv_pack_weights_bf16(const ssmi_rot_entry* table, int32_t count, const float* theta, uint16_t* fwd_pack,
                                        uint16_t* dgrad_pack, void* stream) {
  SSMI_REQUIRE(table && theta && (fwd_pack || dgrad_pack) && count > 0 && count <= 65535, SSMI_E_INVALID_ARG,
               "conv_pack_weights_bf16: bad argument");
  static const bool simple = getenv("SSMI_SIMPLE_PACK") != nullptr;      // measurement switch: the element-by-element kernel
  if (!simple) return launch_conv_pack_bf16(table, count, theta, fwd_pack, dgrad_pack, as_stream(stream));
  hipLaunchKernelGGL(conv_pack_bf16_batch_kernel, dim3(64, (unsigned)count), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const RotEntry*>(table), theta, reinterpret_cast<bf16*>(fwd_pack),
                     reinterpret_cast<bf16*>(dgrad_pack));
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

int32_t ssmi_conv2d_bwd_data_uses_rotated(const ssmi_conv_desc* d) {
  if (check_desc(d) != SSMI_OK) return 0;
  return dgrad_as_fwd(to_geo(d), true) ? 1 : 0;
}

ssmi_status ssmi_conv_rotate_weights(const ssmi_rot_entry* table, int32_t count, const float* theta, float* theta_rot,
                                     void* stream) {
  static_assert(sizeof(RotEntry) == sizeof(ssmi_rot_entry), "rotation table entry layout");
  SSMI_REQUIRE(table && theta && theta_rot && count > 0 && count <= 65535, SSMI_E_INVALID_ARG, "conv_rotate_weights: bad argument");
  hipLaunchKernelGGL(conv_wt_rot_batch_kernel, dim3(64, (unsigned)count), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const RotEntry*>(table), theta, theta_rot);
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

ssmi_status ssmi_conv2d_bwd_data_rot(const ssmi_conv_desc* d, const void* dy, const float* w, const float* w_rot, const void* add,
                                     void* dx, void* stream) {
  ssmi_status s = check_desc(d);
  if (s) return s;
  SSMI_REQUIRE(dy && w && dx, SSMI_E_INVALID_ARG, "conv2d_bwd_data_rot: NULL tensor");
  return dispatch_dtype(d->dtype, [&](auto tag) {
    using T = decltype(tag);
    return launch_bwd_data<T>(to_geo(d), static_cast<const T*>(dy), w, static_cast<const T*>(add), nullptr, static_cast<T*>(dx),
                              as_stream(stream), BnBwdArgs{}, nullptr, nullptr, w_rot);
  });
}

size_t ssmi_conv2d_bwd_data_bn_bwd_workspace_bytes(const ssmi_conv_desc* d) {
  if (check_desc(d) != SSMI_OK) return 0;
  const int64_t npix = (int64_t)d->n * d->h * d->w;
  const size_t fused = (size_t)((ceil_div64(npix, 128) + 1024) * 3 + 3) * d->cin * sizeof(float);
  const size_t plain = ssmi_bn_workspace_bytes(npix, d->cin);
  return (fused > plain ? fused : plain) + dgrad_wt_bytes(to_geo(d));      // the rotated kernel matrix sits behind the sums
}

extern "C++" {
template <typename T>
static ssmi_status conv2d_bwd_data_bn_bwd_impl(const ssmi_conv_desc* d, const T* dy, const float* w, const float* w_rot, const T* add, T* da,
                                               const T* bn_x, int32_t ld_x, const float* mean, const float* invstd,
                                               const float* gamma, const float* beta, int32_t act, const float* alpha,
                                               float slope, const float* drop, int64_t hw, T* dx_bn, float* dgamma,
                                               float* dbeta, float* dalpha, void* ws, size_t ws_bytes, void* stream,
                                               const ssmi_premerge* pm = nullptr, float* coef_out = nullptr) {
  const Geo g = to_geo(d);
  const int64_t npix = (int64_t)d->n * d->h * d->w;
  const int c = d->cin;
  hipStream_t st = as_stream(stream);
  float* wt = nullptr;
  if (dgrad_wt_bytes(g)) wt = reinterpret_cast<float*>(static_cast<char*>(ws) + (ws_bytes - dgrad_wt_bytes(g)) / 16 * 16);
  static const bool no_fuse = getenv("SSMI_NO_FUSED_BN_BWD") != nullptr;   // A/B switch for measurements
  if (no_fuse || d->stride != 1 || c % 4 != 0 || c > 1024 || ld_x % 4 != 0 || d->x_ld % 4 != 0 || ld_x < c ||
      npix * ld_x * 4 >= 0x7FFFFFFFLL || (drop && (hw <= 0 || npix % hw != 0))) {
    SSMI_REQUIRE(!coef_out, SSMI_E_UNSUPPORTED, "conv2d_bwd_data_bn_bwd (deferred apply): layer not supported by the fused reduction");
    ssmi_status s = launch_bwd_data<T>(g, dy, w, add, nullptr, da, st, BnBwdArgs{}, nullptr, wt, w_rot);   // unfused: two plain passes
    if (s) return s;
    return bn_act_drop_bwd_impl<T>(da, d->x_ld, bn_x, ld_x, mean, invstd, gamma, beta, act, alpha, slope, drop, hw, dx_bn, dgamma,
                                   dbeta, dalpha, npix, c, ws, ws_bytes - dgrad_wt_bytes(g), stream);
  }
  float* part = reinterpret_cast<float*>(ws);
  BnBwdArgs bb{};
  bb.x = bn_x; bb.ld = ld_x; bb.mean = mean; bb.invstd = invstd; bb.gamma = gamma; bb.beta = beta; bb.alpha = alpha;
  bb.slope = slope; bb.act = act; bb.drop = drop; bb.hw = drop ? (unsigned)hw : 1u; bb.part = part;
  if (pm) {
    bb.pm_main = pm->main;
    bb.pm[0] = pm->mean; bb.pm[1] = pm->invstd; bb.pm[2] = pm->coef; bb.pm[3] = pm->coef + d->cout; bb.pm[4] = pm->coef + 2 * d->cout;
    bb.pm_drop = pm->drop; bb.pm_hw = (unsigned)pm->hw;
  }
  int rows = 0;
  ssmi_status s = launch_bwd_data<T>(g, dy, w, add, nullptr, da, st, bb, &rows, wt, w_rot);
  if (s) return s;
  if (coef_out)      // the caller's consumer runs the apply pass: parameter gradients and the three coefficient vectors only
    return launch_bn_bwd_final(part, rows, c, npix, gamma, invstd, dgamma, dbeta, act == SSMI_ACT_PRELU ? dalpha : nullptr, coef_out, st);
  float* coef = part + (size_t)rows * 3 * c;
  return launch_bn_bwd_tail<T>(part, rows, da, d->x_ld, bn_x, ld_x, mean, invstd, gamma, beta, act, alpha, slope, drop, hw, dx_bn,
                               dgamma, dbeta, dalpha, npix, c, coef, st);
}
}  // extern "C++"

ssmi_status ssmi_conv2d_bwd_data_bn_bwd(const ssmi_conv_desc* d, const void* dy, const float* w, const void* add,
                                        void* da, const void* bn_x, int32_t ld_x, const float* mean,
                                        const float* invstd, const float* gamma, const float* beta, int32_t act,
                                        const float* alpha, float slope, const float* drop, int64_t hw, void* dx_bn,
                                        float* dgamma, float* dbeta, float* dalpha, void* ws, size_t ws_bytes,
                                        void* stream) {
  ssmi_status s = check_desc(d);
  if (s) return s;
  SSMI_REQUIRE(dy && w && da && bn_x && mean && invstd && gamma && beta && dx_bn, SSMI_E_INVALID_ARG,
               "conv2d_bwd_data_bn_bwd: NULL tensor");
  SSMI_REQUIRE(ws && ws_bytes >= ssmi_conv2d_bwd_data_bn_bwd_workspace_bytes(d), SSMI_E_WORKSPACE,
               "conv2d_bwd_data_bn_bwd: workspace too small");
  SSMI_REQUIRE(act >= SSMI_ACT_NONE && act <= SSMI_ACT_LEAKY && (act != SSMI_ACT_PRELU || alpha), SSMI_E_INVALID_ARG,
               "conv2d_bwd_data_bn_bwd: bad activation");
  return dispatch_dtype(d->dtype, [&](auto tag) {
    using T = decltype(tag);
    return conv2d_bwd_data_bn_bwd_impl<T>(d, static_cast<const T*>(dy), w, nullptr, static_cast<const T*>(add), static_cast<T*>(da),
                                          static_cast<const T*>(bn_x), ld_x, mean, invstd, gamma, beta, act, alpha, slope, drop,
                                          hw, static_cast<T*>(dx_bn), dgamma, dbeta, dalpha, ws, ws_bytes, stream);
  });
}

ssmi_status ssmi_conv2d_bwd_data_bn_bwd_rot(const ssmi_conv_desc* d, const void* dy, const float* w, const float* w_rot,
                                            const void* add,
                                        void* da, const void* bn_x, int32_t ld_x, const float* mean,
                                        const float* invstd, const float* gamma, const float* beta, int32_t act,
                                        const float* alpha, float slope, const float* drop, int64_t hw, void* dx_bn,
                                        float* dgamma, float* dbeta, float* dalpha, void* ws, size_t ws_bytes,
                                        void* stream) {
  ssmi_status s = check_desc(d);
  if (s) return s;
  SSMI_REQUIRE(dy && w && da && bn_x && mean && invstd && gamma && beta && dx_bn, SSMI_E_INVALID_ARG,
               "conv2d_bwd_data_bn_bwd_rot: NULL tensor");
  SSMI_REQUIRE(ws && ws_bytes >= ssmi_conv2d_bwd_data_bn_bwd_workspace_bytes(d), SSMI_E_WORKSPACE,
               "conv2d_bwd_data_bn_bwd_rot: workspace too small");
  SSMI_REQUIRE(act >= SSMI_ACT_NONE && act <= SSMI_ACT_LEAKY && (act != SSMI_ACT_PRELU || alpha), SSMI_E_INVALID_ARG,
               "conv2d_bwd_data_bn_bwd_rot: bad activation");
  return dispatch_dtype(d->dtype, [&](auto tag) {
    using T = decltype(tag);
    return conv2d_bwd_data_bn_bwd_impl<T>(d, static_cast<const T*>(dy), w, w_rot, static_cast<const T*>(add), static_cast<T*>(da),
                                          static_cast<const T*>(bn_x), ld_x, mean, invstd, gamma, beta, act, alpha, slope, drop,
                                          hw, static_cast<T*>(dx_bn), dgamma, dbeta, dalpha, ws, ws_bytes, stream);
  });
}

ssmi_status ssmi_wg_reduce_defer_begin(void) {
  if (!g_wg_defer) g_wg_defer = new std::vector<WgReduceEntry>();
  return SSMI_OK;
}

ssmi_status ssmi_wg_reduce_defer_flush(int32_t end, void* stream) {
  if (!g_wg_defer) return SSMI_OK;
  hipStream_t st = as_stream(stream);
  const std::vector<WgReduceEntry>& v = *g_wg_defer;
  for (size_t i0 = 0; i0 < v.size(); i0 += kWgBatchMax) {
    WgReduceBatch b{};
    b.n = (int)std::min<size_t>(kWgBatchMax, v.size() - i0);
    int wgs = 0;
    for (int i = 0; i < b.n; ++i) {
      b.e[i] = v[i0 + i];
      b.first_wg[i] = wgs;
      wgs += (b.e[i].K_aug * b.e[i].cout + 15) / 16;
    }
    b.first_wg[b.n] = wgs;
    hipLaunchKernelGGL(conv_bwd_weight_reduce_batch, dim3((unsigned)wgs), dim3(256), 0, st, b);
    SSMI_LAUNCH_CHECK();
  }
  g_wg_defer->clear();
  if (end) {
    delete g_wg_defer;
    g_wg_defer = nullptr;
  }
  return SSMI_OK;
}

int32_t ssmi_conv2d_bwd_data_bn_bwd_deferred_supported(const ssmi_conv_desc* d, int32_t ld_x) {
  if (check_desc(d) != SSMI_OK) return 0;
  static const bool no_fuse = getenv("SSMI_NO_FUSED_BN_BWD") != nullptr;
  const int64_t npix = (int64_t)d->n * d->h * d->w;
  return !(no_fuse || d->stride != 1 || d->cin % 4 != 0 || d->cin > 1024 || ld_x % 4 != 0 || d->x_ld % 4 != 0 || ld_x < d->cin ||
           npix * ld_x * 4 >= 0x7FFFFFFFLL);
}

ssmi_status ssmi_conv2d_bwd_data_bn_bwd_rot_deferred(const ssmi_conv_desc* d, const void* dy, const float* w, const float* w_rot,
                                                     void* da, const void* bn_x, int32_t ld_x, const float* mean, const float* invstd,
                                                     const float* gamma, const float* beta, int32_t act, const float* alpha, float slope,
                                                     float* dgamma, float* dbeta, float* dalpha, float* coef, void* ws, size_t ws_bytes,
                                                     void* stream) {
  ssmi_status s = check_desc(d);
  if (s) return s;
  SSMI_REQUIRE(dy && w && da && bn_x && mean && invstd && gamma && beta && dgamma && dbeta && coef, SSMI_E_INVALID_ARG,
               "conv2d_bwd_data_bn_bwd_rot_deferred: NULL argument");
  SSMI_REQUIRE(ws && ws_bytes >= ssmi_conv2d_bwd_data_bn_bwd_workspace_bytes(d), SSMI_E_WORKSPACE,
               "conv2d_bwd_data_bn_bwd_rot_deferred: workspace too small");
  SSMI_REQUIRE(act >= SSMI_ACT_NONE && act <= SSMI_ACT_LEAKY && (act != SSMI_ACT_PRELU || (alpha && dalpha)), SSMI_E_INVALID_ARG,
               "conv2d_bwd_data_bn_bwd_rot_deferred: bad activation");
  return dispatch_dtype(d->dtype, [&](auto tag) {
    using T = decltype(tag);
    return conv2d_bwd_data_bn_bwd_impl<T>(d, static_cast<const T*>(dy), w, w_rot, nullptr, static_cast<T*>(da), static_cast<const T*>(bn_x),
                                          ld_x, mean, invstd, gamma, beta, act, alpha, slope, nullptr, 1, static_cast<T*>(da), dgamma, dbeta,
                                          dalpha, ws, ws_bytes, stream, nullptr, coef);
  });
}

int32_t ssmi_conv2d_bwd_data_premerge_supported(const ssmi_conv_desc* d) {
  if (check_desc(d) != SSMI_OK || d->dtype != SSMI_F32) return 0;
  static const bool no_fuse = getenv("SSMI_NO_FUSED_BN_BWD") != nullptr;
  const Geo g = to_geo(d);
  if (no_fuse || !dgrad_as_fwd(g, true) || g.cin % 4 != 0 || g.x_ld % 4 != 0) return 0;
  Geo f = g;      // the forward-form launch of the data gradient
  f.h = g.ho; f.w = g.wo; f.cin = g.cout; f.x_ld = g.y_ld;
  f.ho = g.h; f.wo = g.w; f.cout = g.cin; f.y_ld = g.x_ld;
  return premerge_ok(f) && pick_nt(f.cout) == 1;
}

ssmi_status ssmi_conv2d_bwd_data_bn_bwd_premerge(const ssmi_conv_desc* d, const ssmi_premerge* pm, const float* w,
                                                 const float* w_rot, void* da, const void* bn_x, int32_t ld_x,
                                                 const float* mean, const float* invstd, const float* gamma,
                                                 const float* beta, int32_t act, const float* alpha, float slope,
                                                 void* dx_bn, float* dgamma, float* dbeta, float* dalpha, void* ws,
                                                 size_t ws_bytes, void* stream) {
  ssmi_status s = check_desc(d);
  if (s) return s;
  SSMI_REQUIRE(pm && pm->dsum && pm->main && pm->mean && pm->invstd && pm->coef && w && w_rot && da && bn_x && mean && invstd &&
               gamma && beta && dx_bn, SSMI_E_INVALID_ARG, "conv2d_bwd_data_bn_bwd_premerge: NULL argument");
  SSMI_REQUIRE(ssmi_conv2d_bwd_data_premerge_supported(d), SSMI_E_UNSUPPORTED,
               "conv2d_bwd_data_bn_bwd_premerge: layer not supported (float32, 1x1 stride 1, forward-form data gradient, <= 32 input and "
               "<= 128 output channels)");
  SSMI_REQUIRE(!pm->drop || (pm->hw > 0 && ((int64_t)d->ho * d->wo) % pm->hw == 0), SSMI_E_SHAPE, "conv2d_bwd_data_bn_bwd_premerge: bad hw");
  SSMI_REQUIRE(ws && ws_bytes >= ssmi_conv2d_bwd_data_bn_bwd_workspace_bytes(d), SSMI_E_WORKSPACE,
               "conv2d_bwd_data_bn_bwd_premerge: workspace too small");
  SSMI_REQUIRE(act >= SSMI_ACT_NONE && act <= SSMI_ACT_LEAKY && (act != SSMI_ACT_PRELU || alpha), SSMI_E_INVALID_ARG,
               "conv2d_bwd_data_bn_bwd_premerge: bad activation");
  return conv2d_bwd_data_bn_bwd_impl<float>(d, static_cast<const float*>(pm->dsum), w, w_rot, nullptr, static_cast<float*>(da),
                                            static_cast<const float*>(bn_x), ld_x, mean, invstd, gamma, beta, act, alpha, slope,
                                            nullptr, 1, static_cast<float*>(dx_bn), dgamma, dbeta, dalpha, ws, ws_bytes, stream, pm);
}

size_t ssmi_conv2d_bwd_weight_workspace_bytes(const ssmi_conv_desc* d) {
  if (check_desc(d)) return 0;
  return wg_ws_bytes(to_geo(d));
}

ssmi_status ssmi_conv2d_bwd_weight(const ssmi_conv_desc* d, const void* x, const void* dy, float* dw, float* dbias,
                                   void* ws, size_t ws_bytes, void* stream) {
  ssmi_status s = check_desc(d);
  if (s) return s;
  SSMI_REQUIRE(x && dy && dw, SSMI_E_INVALID_ARG, "conv2d_bwd_weight: NULL tensor");
  return dispatch_dtype(d->dtype, [&](auto tag) {
    using T = decltype(tag);
    return launch_bwd_weight<T>(to_geo(d), static_cast<const T*>(x), static_cast<const T*>(dy), dw, dbias, ws, ws_bytes,
                                as_stream(stream));
  });
}

ssmi_status ssmi_conv2d_transpose_fwd(const ssmi_conv_desc* d, const void* xt, const float* w, const float* bias,
                                      void* yt, void* stream) {
  ssmi_status s = check_desc(d);
  if (s) return s;
  SSMI_REQUIRE(xt && w && yt, SSMI_E_INVALID_ARG, "conv2d_transpose_fwd: NULL tensor");
  return dispatch_dtype(d->dtype, [&](auto tag) {
    using T = decltype(tag);
    return launch_bwd_data<T>(to_geo(d), static_cast<const T*>(xt), w, (const T*)nullptr, bias, static_cast<T*>(yt),
                              as_stream(stream));
  });
}

ssmi_status ssmi_conv2d_transpose_bwd_data(const ssmi_conv_desc* d, const void* dyt, const float* w, void* dxt,
                                           void* stream) {
  ssmi_status s = check_desc(d);
  if (s) return s;
  SSMI_REQUIRE(dyt && w && dxt, SSMI_E_INVALID_ARG, "conv2d_transpose_bwd_data: NULL tensor");
  return dispatch_dtype(d->dtype, [&](auto tag) {
    using T = decltype(tag);
    return launch_fwd<T>(to_geo(d), static_cast<const T*>(dyt), w, nullptr, static_cast<T*>(dxt), as_stream(stream));
  });
}

static int colsum_blocks(int64_t npix, int c) {
  const int ppi = 256 / c;
  int64_t b = ceil_div64(npix, (int64_t)ppi * 8);
  if (b > 512) b = 512;
  if (b < 1) b = 1;
  return (int)b;
}

size_t ssmi_conv2d_transpose_bwd_weight_workspace_bytes(const ssmi_conv_desc* d) {
  if (check_desc(d)) return 0;
  const size_t a = wg_ws_bytes(to_geo(d));
  return a + (size_t)512 * d->cin * sizeof(float);
}

extern "C++" {
template <typename T>
static ssmi_status conv2d_transpose_bwd_weight_impl(const ssmi_conv_desc* d, const T* xt, const T* dyt, float* dw,
                                                    float* dbias, void* ws, void* stream) {
  Geo g = to_geo(d);
  const size_t a = wg_ws_bytes(g);
  ssmi_status s = launch_bwd_weight<T>(g, dyt, xt, dw, nullptr, ws, a, as_stream(stream));
  if (s) return s;
  if (dbias) {
    SSMI_REQUIRE(g.cin <= 256, SSMI_E_UNSUPPORTED, "conv2d_transpose_bwd_weight: bias gradient supports <= 256 channels");
    float* part = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + a);
    const int64_t npix = (int64_t)g.n * g.h * g.w;
    const int nb = colsum_blocks(npix, g.cin);
    if (g.cin % 4 == 0 && g.x_ld % 4 == 0 && (reinterpret_cast<uintptr_t>(dyt) & 15) == 0)
      hipLaunchKernelGGL((colsum_partial<T, true>), dim3(nb), dim3(256), 0, as_stream(stream), dyt, npix, g.cin, g.x_ld, part);
    else
      hipLaunchKernelGGL((colsum_partial<T, false>), dim3(nb), dim3(256), 0, as_stream(stream), dyt, npix, g.cin, g.x_ld, part);
    SSMI_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_final, dim3((g.cin + 7) / 8), dim3(256), 0, as_stream(stream), part, nb, g.cin, dbias);
    SSMI_LAUNCH_CHECK();
  }
  return SSMI_OK;
}
}  // extern "C++"

ssmi_status ssmi_conv2d_transpose_bwd_weight(const ssmi_conv_desc* d, const void* xt, const void* dyt, float* dw,
                                             float* dbias, void* ws, size_t ws_bytes, void* stream) {
  ssmi_status s = check_desc(d);
  if (s) return s;
  SSMI_REQUIRE(xt && dyt && dw, SSMI_E_INVALID_ARG, "conv2d_transpose_bwd_weight: NULL tensor");
  SSMI_REQUIRE(ws_bytes >= ssmi_conv2d_transpose_bwd_weight_workspace_bytes(d), SSMI_E_WORKSPACE,
               "conv2d_transpose_bwd_weight: workspace too small");
  return dispatch_dtype(d->dtype, [&](auto tag) {
    using T = decltype(tag);
    return conv2d_transpose_bwd_weight_impl<T>(d, static_cast<const T*>(xt), static_cast<const T*>(dyt), dw, dbias, ws, stream);
  });
}

}  // extern "C"

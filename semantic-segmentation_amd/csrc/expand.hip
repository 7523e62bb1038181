// The 1x1 expansion of an ENet bottleneck (src/models.py:1142-1172: Conv2D 1x1 -> BatchNormalization -> SpatialDropout2D ->
// Add with the skip branch -> PReLU) WITHOUT its output tensor.  The expansion maps 16 or 32 channels to 64 or 128:
// recomputing it from its 4x smaller input is nearly free, while its output y3 is the widest tensor of the block and would be
// written once and read three times (merge forward, both merge-backward passes).  Every mode of the kernel recomputes
//     y3 = PReLU2(BN2(y2_raw)) * W3                 (normalise on load, as ssmi_conv2d_fwd_bn_stats_prebn)
// per tile and consumes it in place, with the arithmetic of the merge kernels per element:
//   S   (0) column sums of y3 and y3^2 (training-phase statistics of BN3); nothing is stored
//   F   (1) out = PReLU3(drop * BN3(y3) + skip)                                      (ssmi_merge_fwd; skip: the block input,
//           or - downsampling block - its max-pooled input as a channel-padded operand)
//   Ft  (6) F carried THROUGH the next block's 1x1 reduction (src/models.py:1090-1100): the output tile goes back into LDS and
//           is multiplied with that layer's kernel (the waves split the 64 / 128 channels, partial products through LDS);
//           y1' = out * W1' is stored with the column sums of y1' and y1'^2 (BN1' statistics) - the next block starts with its
//           BatchNormalization apply instead of a convolution.  NA (inference phase: fixed statistics): BN1' + PReLU1' are
//           applied right there and the ACTIVATED tensor is stored; no sums
//   B1  (2) dsum = dout * PReLU3'(s), column sums of dz, dz * xhat3, dout * min(s, 0) (ssmi_merge_bwd, reducing pass)
//   B1c (4) B1 whose dout is not read but formed per tile: dout = dy1' * W1'^T + dsum', the data gradient of the NEXT block's
//           1x1 reduction plus that block's skip gradient - the same (C/4 -> C) product shape as the expansion, so the tile
//           loop runs two products and dout is neither written nor read.  NA: dy1' is still the gradient BEHIND that block's
//           BN1 + PReLU1; the apply pass of their backward runs on the operand as it is staged (and writes dy1' back)
//   B2  (3) dmain = k1 * (dsum * drop - m1 - xhat3 * m2)                              (ssmi_merge_bwd, apply pass)
//   B2t (5) B2 carried THROUGH the expansion's own data gradient: the dmain tile goes back into LDS and is multiplied with
//           W3^T (same split as Ft); da2 = dmain * W3^T is stored with the column sums BN2's backward needs (sum dz,
//           sum dz * xhat2, sum da2 * min(z2, 0): ssmi_conv2d_bwd_data_bn_bwd's); dmain is still stored for the expansion's
//           weight gradient but not read again on the main chain
//   WG  (B1c, B2t) the weight gradient of a 1x1 layer formed where its operands already sit in LDS, accumulated in registers
//           over the workgroup's tiles and written as one partial matrix per workgroup (per 32-pixel block of a 64-pixel tile),
//           summed by ssmi_expand_weight_reduce:  B2t: dW3 = a2^T * dmain (dmain is then not stored at all);  B1c: dW1' =
//           out^T * dy1', the NEXT block's reduction (out = PReLU3(s) is what this pass computes anyway)
// A 256-thread workgroup walks 64-pixel (64 channels) or 32-pixel (128 channels) tiles: the raw input tile is normalised on its
// way into LDS, four waves multiply (pixel blocks x channel groups), the accumulators go through LDS, and the elementwise
// phase runs in the merge kernels' layout - a thread owns four consecutive channels of a pixel, 16-byte accesses.  The wide
// operands of a tile are requested before its multiply phase, the narrow ones one tile ahead.
// float32 storage: v_mfma_f32_32x32x2_f32, kernel matrices resident in LDS.  bfloat16 storage: the operand, the kernel and y3
// take the roundings the stored route's tensors and packs have (values equal that route's up to summation order); products on
// v_mfma_f32_32x32x16_bf16, operand tiles in LDS as bfloat16 rows, kernel matrices in registers as matrix-instruction operands
// for the whole launch.  (16 -> 64) and (32 -> 128) channels, the two shapes the ENets have.
#include "common.h"

namespace ssmi {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// eight float32 values as the bfloat16 operand of v_mfma_f32_32x32x16_bf16 (round to nearest even, element 0 lowest)
__device__ __forceinline__ bf16x8 pack_bf16x8(const f32x4& lo, const f32x4& hi) {
  u32x4 w;
  w[0] = pack_bf16x2(lo[0], lo[1]);
  w[1] = pack_bf16x2(lo[2], lo[3]);
  w[2] = pack_bf16x2(hi[0], hi[1]);
  w[3] = pack_bf16x2(hi[2], hi[3]);
  return __builtin_bit_cast(bf16x8, w);
}

struct ExpandArgs {
  int npix, hw;
  const void* y2;           // [npix][C4] raw output of the convolution in front (activation type)
  const float* pre[5];      // mean2, invstd2, gamma2, beta2, alpha2 (alpha2 nullptr: no activation)
  const float* w;           // [C4][C]
  const float *mean3, *invstd3, *gamma3, *beta3, *alpha3;
  const float* drop;        // [n][C] or nullptr
  const void* skip;         // [npix][c_skip]: the first c_skip channels of the skip branch, zeros beyond   (F, B1)
  int c_skip;
  const void* dout;         // [npix][C]                       (B1; B1c: the next block's skip gradient dsum')
  const void* dy_next;      // [npix][C4] gradient at the output of the next block's 1x1 reduction   (B1c)
  const float* w_next;      // [C][C4] that reduction's kernel                                       (B1c)
  // B1c, optional (y1_next != nullptr): dy_next is still the gradient BEHIND that block's BN1 + PReLU1 - their backward's
  // apply pass (ssmi_bn_act_bwd's arithmetic, coefficients k1 | m1 | m2 in coef_next) runs on the operand as it is staged,
  // and the result goes back to dy_next in place (that block's weight gradient reads it)
  const void* y1_next;      // [npix][C4] raw output of that reduction
  const float* bn1_next[5]; // mean, invstd, gamma, beta, alpha (alpha nullptr: no activation)
  const float* coef_next;
  const void* dsum_in;      // [npix][C]                       (B2)
  const float* coef;        // k1 | m1 | m2, C floats each     (B2)
  void* out;                // F: block output; B1: dsum; B2: dmain
  void* y3_out;             // F only, nullable: also store the recomputed expansion output (tests, debugging)
  float* part;              // S: [grid][2][C]; B1: [grid][3][C]
  void* da2;                // B2t: [npix][C4] gradient at the expansion's (normalised) input; Ft: [npix][C4] output of the next block's reduction
  float* part2;             // B2t: [grid][3][C4]; Ft: [grid][2][C4]
  float* wpart;             // WG: [rows][C4][C] (B2t) / [rows][C][C4] (B1c), rows = grid * (pixel blocks per tile)
};

// LDS layout (floats), shared by the kernel and its launcher
template <typename T, int C4, int C, int MODE, bool WG>
struct ExpandLds {
  static constexpr bool kBf = sizeof(T) == 2;
  static constexpr int TP = C == 128 ? 32 : 64;
  static constexpr int W = 0;                                   // (kernel matrices: register operands of the matrix instructions)
  static constexpr int A = (TP * (C4 + 1) + 3) & ~3;            // a narrow operand tile
  static constexpr int S = TP * (C + 4);                        // a staged accumulator tile
  static constexpr int D = (TP * (C * 2 + 16) + 3) / 4;         // a bfloat16 wide tile
  static constexpr int Wl = 0, Al = W, St = Al + A, X = St + S; // X: start of the mode's own regions
  static constexpr int Wn = X, An = Wn + W, Sn = An + A, Ot = Sn + S;                 // B1c
  // B2t / Ft: the partial products Pn [waves along channels][TP][33] reuse the staged tile St, which is dead by then (TP * 33 *
  // (4 / (TP / 32)) <= TP * (C + 4) for both shapes)
  static constexpr int Wt = X, WtN = kBf ? D : 0, Pn = St;      // Wt: the bfloat16 dmain / output tile
  static constexpr int total = MODE == 4 ? Ot + (WG && kBf ? D : 0) : (MODE == 5 || MODE == 6 ? Wt + WtN : X);
  static_assert(4 * 32 * 33 <= S, "partial products fit the staged tile");
};

template <typename T, int C4, int C, int MODE, bool WG = false, bool NA = false>
__global__ __launch_bounds__(256, 2) void expand_kernel(ExpandArgs a) {
  static_assert(!NA || MODE == 4 || MODE == 6, "the next block's BN1: backward apply pass in B1c, inference-phase forward in Ft");
  using L = ExpandLds<T, C4, C, MODE, WG>;
  static_assert(!WG || MODE == 4 || MODE == 5, "the fused weight gradients belong to the B1c and B2t modes");
  constexpr bool kBf = sizeof(T) == 2;      // bfloat16 storage: operands and y3 take the roundings the stored route's tensors have
  constexpr int TP = C == 128 ? 32 : 64;      // pixels per tile (eight elementwise passes per tile cost the 128-wide variant > 256 registers)
  constexpr int PWN = TP / 32, CWN = 4 / PWN; // waves along pixels / along channels
  constexpr int AS = C4 + 1;                  // row stride of the normalised input tile (floats; odd: conflict-free column reads)
  constexpr int SS = C + 4;                   // row stride of the staged accumulators
  constexpr int G = C / 4, PL = 256 / G, NPASS = TP / PL;
  constexpr int NT = C / (32 * CWN);          // 32-column tiles per wave
  constexpr int P4 = C4 / 4, NY = TP * P4 / 256;
  constexpr int NS = MODE == 0 ? 2 : 3;
  constexpr bool kB1 = MODE == 2 || MODE == 4, kChain = MODE == 4;
  constexpr bool kB2 = MODE == 3 || MODE == 5, kThru = MODE == 5;
  constexpr bool kF = MODE == 1 || MODE == 6, kFt = MODE == 6;
  constexpr bool kHasD = kB1 || kB2;          // a wide gradient operand (dout / dsum)
  constexpr int KS = CWN, KW = C / KS;        // B2t: the waves of a pixel block split the reduction over the C channels
  constexpr int WTS = 33;                     // B2t: row stride of W3^T (columns beyond C4 are zero) and of the partial products
  static_assert(!(kThru || kFt) || (NY == 1 && KW % 2 == 0), "shape");
  constexpr bool kSums = MODE == 0 || kB1;
  static_assert(NY >= 1 && 256 % P4 == 0 && 256 % G == 0, "shape");
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* const Al = sm + L::Al;               // [TP][AS]
  float* const St = sm + L::St;               // [TP][SS], later the reduction scratch
  float* const An = sm + L::An;               // B1c: [TP][AS]
  float* const Sn = sm + L::Sn;               // B1c: [TP][SS]
  float* const Wt = sm + L::Wt;               // B2t / Ft, bfloat16: the dmain / output tile [TP][2 C + 16 bytes]
  float* const Pn = sm + L::Pn;               // B2t: [KS][TP][WTS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int pw = wave % PWN, cw = wave / PWN;

  // bfloat16: the kernel matrices as matrix-instruction operands in registers (lane (r, hh) of a 32-wide column tile holds rows
  // 16 s + 8 hh .. + 7 of column r), rounded as the stored route's packs are
  constexpr int ABS = C4 * 2 + 16, DBS = C * 2 + 16;      // row strides (bytes) of the bfloat16 operand tiles in LDS
  constexpr int K1 = C4 / 16, K3 = KW / 16;
  unsigned char* const Alb = reinterpret_cast<unsigned char*>(Al);
  unsigned char* const Anb = reinterpret_cast<unsigned char*>(An);
  unsigned char* const Dtb = reinterpret_cast<unsigned char*>(Wt);      // B2t: the dmain tile [TP][DBS]
  unsigned char* const Otb = reinterpret_cast<unsigned char*>(sm + L::Ot);      // B1c + WG: the block-output tile [TP][DBS]
  f32x16 accw;              // WG: this wave's 32 x 32 block of the weight gradient, over all tiles of the workgroup
#pragma unroll
  for (int i = 0; i < 16; ++i) accw[i] = 0.f;
  // a matrix-instruction operand (16 pixels x 32 channels, pixels as the reduction dimension) out of a pixel-major bfloat16
  // tile in LDS through the transposing read; p points at (first pixel, first channel) of the block
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  const int sq = (lane & 15) >> 2, sp = lane & 3, gh = (lane >> 4) & 1;
  auto operand_t = [&](const unsigned char* p, int rowb) {
    struct Pair { s16x4 lo, hi; } pr;
    const unsigned char* q = p + (8 * hh + sq) * rowb + (16 * gh + 4 * sp) * 2;
    pr.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(q));
    pr.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(q + 4 * rowb));
    return __builtin_bit_cast(bf16x8, pr);
  };
  bf16x8 wf[kBf ? NT : 1][kBf ? K1 : 1], wnf[kBf && kChain ? NT : 1][kBf && kChain ? K1 : 1], wtf[kBf && (kThru || kFt) ? K3 : 1];
  if constexpr (kBf) {
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
      for (int s2 = 0; s2 < K1; ++s2) {
        const int col = cw * (C / CWN) + u * 32 + r;
        f32x4 lo, hi;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          lo[j] = a.w[(16 * s2 + 8 * hh + j) * C + col];
          hi[j] = a.w[(16 * s2 + 8 * hh + 4 + j) * C + col];
        }
        wf[u][s2] = pack_bf16x8(lo, hi);
        if constexpr (kChain) {
          const float* q = a.w_next + col * C4 + 16 * s2 + 8 * hh;
          wnf[u][s2] = pack_bf16x8(*reinterpret_cast<const f32x4*>(q), *reinterpret_cast<const f32x4*>(q + 4));
        }
      }
    if constexpr (kThru) {
#pragma unroll
      for (int s2 = 0; s2 < K3; ++s2) {
        f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
        if (r < C4) {
          const float* q = a.w + r * C + cw * KW + 16 * s2 + 8 * hh;
          lo = *reinterpret_cast<const f32x4*>(q);
          hi = *reinterpret_cast<const f32x4*>(q + 4);
        }
        wtf[s2] = pack_bf16x8(lo, hi);
      }
    }
    if constexpr (kFt) {      // column r of the next block's reduction kernel [C][C4], this wave's share of its rows
#pragma unroll
      for (int s2 = 0; s2 < K3; ++s2) {
        f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
        if (r < C4) {
          const float* q = a.w_next + (cw * KW + 16 * s2 + 8 * hh) * C4 + r;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            lo[j] = q[j * C4];
            hi[j] = q[(4 + j) * C4];
          }
        }
        wtf[s2] = pack_bf16x8(lo, hi);
      }
    }
  }
  // float32: the same, one word per lane and matrix instruction (lane (r, hh) holds row 2 s + hh of column r)
  constexpr int Q1 = C4 / 2, Q3 = KW / 2;
  float wq[!kBf ? NT : 1][!kBf ? Q1 : 1], wnq[!kBf && kChain ? NT : 1][!kBf && kChain ? Q1 : 1], wtq[!kBf && (kThru || kFt) ? Q3 : 1];
  if constexpr (!kBf) {
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int col = cw * (C / CWN) + u * 32 + r;
#pragma unroll
      for (int s2 = 0; s2 < Q1; ++s2) {
        wq[u][s2] = a.w[(2 * s2 + hh) * C + col];
        if constexpr (kChain) wnq[u][s2] = a.w_next[col * C4 + 2 * s2 + hh];      // the next block's reduction kernel, transposed
      }
    }
    if constexpr (kThru || kFt) {      // B2t: W3^T, Ft: the next block's reduction kernel; this wave's share of the C rows, column r
#pragma unroll
      // (the reduction index runs in the order the operand tile is read below: 16 bytes = four consecutive channels per lane)
      for (int s2 = 0; s2 < Q3; ++s2) {
        const int cc = cw * KW + 8 * (s2 >> 2) + 4 * hh + (s2 & 3);
        wtq[s2] = r < C4 ? (kThru ? a.w[r * C + cc] : a.w_next[cc * C4 + r]) : 0.f;
      }
    }
  }

  // the thread's four input channels (normalise-on-load role) and their BN2 / PReLU2 parameters
  const int yc = (tid % P4) * 4;
  const f32x4 m2 = *reinterpret_cast<const f32x4*>(a.pre[0] + yc), is2 = *reinterpret_cast<const f32x4*>(a.pre[1] + yc);
  const f32x4 g2 = *reinterpret_cast<const f32x4*>(a.pre[2] + yc), b2 = *reinterpret_cast<const f32x4*>(a.pre[3] + yc);
  const bool act2 = a.pre[4] != nullptr;
  f32x4 al2 = {0.f, 0.f, 0.f, 0.f};
  if (act2) al2 = *reinterpret_cast<const f32x4*>(a.pre[4] + yc);

  // the thread's four output channels (elementwise role) and their parameters
  const int cg = tid % G, pl = tid / G, c = cg * 4;
  f32x4 m3, is3, g3, b3, al3, k1, q1, q2;
  m3 = is3 = g3 = b3 = al3 = k1 = q1 = q2 = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (MODE != 0) {
    m3 = *reinterpret_cast<const f32x4*>(a.mean3 + c);
    is3 = *reinterpret_cast<const f32x4*>(a.invstd3 + c);
  }
  if constexpr (kF || kB1) {
    g3 = *reinterpret_cast<const f32x4*>(a.gamma3 + c);
    b3 = *reinterpret_cast<const f32x4*>(a.beta3 + c);
    al3 = *reinterpret_cast<const f32x4*>(a.alpha3 + c);
  }
  if constexpr (kB2) {
    k1 = *reinterpret_cast<const f32x4*>(a.coef + c);
    q1 = *reinterpret_cast<const f32x4*>(a.coef + C + c);
    q2 = *reinterpret_cast<const f32x4*>(a.coef + 2 * C + c);
  }
  f32x4 v[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 v2[3];              // mode B2t: this thread's sums over its four input channels
#pragma unroll
  for (int k = 0; k < 3; ++k) v2[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  float lane_s[NT][2];      // mode S: this lane's column sums (channel cw * C/2 + u * 32 + r)
#pragma unroll
  for (int u = 0; u < NT; ++u) lane_s[u][0] = lane_s[u][1] = 0.f;

  // offsets below are byte offsets of a float32 tensor; the bf16 accessors halve them (common.h)
  const unsigned wide_bytes = (unsigned)a.npix * (unsigned)C * 4u;
  const __amdgpu_buffer_rsrc_t ry2 = Act<T>::rsrc(a.y2, (unsigned)a.npix * (unsigned)C4 * 4u);
  const __amdgpu_buffer_rsrc_t rskip = Act<T>::rsrc(kF || kB1 ? a.skip : a.y2,
                                                    kF || kB1 ? (unsigned)a.npix * (unsigned)a.c_skip * 4u : 0u);
  const __amdgpu_buffer_rsrc_t rdn = Act<T>::rsrc(kChain ? a.dy_next : a.y2, kChain ? (unsigned)a.npix * (unsigned)C4 * 4u : 0u);
  constexpr bool apply_next = NA;
  const __amdgpu_buffer_rsrc_t ry1n = Act<T>::rsrc(apply_next ? a.y1_next : a.y2, apply_next ? (unsigned)a.npix * (unsigned)C4 * 4u : 0u);
  const __amdgpu_buffer_rsrc_t rdno = Act<T>::rsrc(apply_next ? const_cast<void*>(a.dy_next) : (void*)a.part,
                                                   apply_next ? (unsigned)a.npix * (unsigned)C4 * 4u : 0u);
  f32x4 n_m, n_is, n_g, n_b, n_al, n_k1, n_m1, n_m2;      // B1c apply pass: the next block's BN1 / PReLU1 parameters of this thread's four channels
  n_m = n_is = n_g = n_b = n_al = n_k1 = n_m1 = n_m2 = f32x4{0.f, 0.f, 0.f, 0.f};
  bool n_act = false;
  if constexpr (NA) {
    {      // (Ft, inference phase: coef_next is unused and may be null)
      n_m = *reinterpret_cast<const f32x4*>(a.bn1_next[0] + yc);
      n_is = *reinterpret_cast<const f32x4*>(a.bn1_next[1] + yc);
      n_g = *reinterpret_cast<const f32x4*>(a.bn1_next[2] + yc);
      n_b = *reinterpret_cast<const f32x4*>(a.bn1_next[3] + yc);
      n_act = a.bn1_next[4] != nullptr;
      if (n_act) n_al = *reinterpret_cast<const f32x4*>(a.bn1_next[4] + yc);
      if constexpr (kChain) {
        n_k1 = *reinterpret_cast<const f32x4*>(a.coef_next + yc);
        n_m1 = *reinterpret_cast<const f32x4*>(a.coef_next + C4 + yc);
        n_m2 = *reinterpret_cast<const f32x4*>(a.coef_next + 2 * C4 + yc);
      }
    }
  }
  const bool has_skip = c < a.c_skip;      // channel-padded skip branch (downsampling blocks): zeros beyond its channels
  const __amdgpu_buffer_rsrc_t rd = Act<T>::rsrc(kB1 ? a.dout : (kB2 ? a.dsum_in : a.y2), kHasD ? wide_bytes : 0u);
  const __amdgpu_buffer_rsrc_t rda = Act<T>::rsrc(kThru || kFt ? a.da2 : (void*)a.part, kThru || kFt ? (unsigned)a.npix * (unsigned)C4 * 4u : 0u);
  const __amdgpu_buffer_rsrc_t ro = Act<T>::rsrc(MODE != 0 && a.out ? a.out : (void*)a.part, MODE != 0 && a.out ? wide_bytes : 0u);      // (B2t + WG: dmain may be NULL, nothing is stored)
  const __amdgpu_buffer_rsrc_t ry3 = Act<T>::rsrc(kF && a.y3_out ? a.y3_out : (void*)a.part, kF && a.y3_out ? wide_bytes : 0u);
  auto store4 = [&](__amdgpu_buffer_rsrc_t rs, unsigned off4, const f32x4& val) {      // out-of-range offsets are dropped
    if constexpr (kBf) __builtin_amdgcn_raw_buffer_store_b64(pack_bf16x4(val), rs, off4 >> 1, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val), rs, off4, 0, 0);
  };

  const int ntiles = (a.npix + TP - 1) / TP;
  // the narrow operands (the expansion's input, B1c: the next block's dy1) are needed first in a tile, in front of a barrier:
  // they are requested one tile ahead (tiles beyond the last: out-of-range offsets, nothing is fetched)
  typename Act<T>::raw4 yv_next[NY], nv_next[kChain ? NY : 1], y1_nx[NA ? NY : 1];      // raw while in flight, widened at the use
  auto request_narrow = [&](int t) {
#pragma unroll
    for (int j = 0; j < NY; ++j) {
      const unsigned p = (unsigned)t * TP + (unsigned)((tid + 256 * j) / P4);
      const unsigned off = (t < ntiles && p < (unsigned)a.npix) ? (p * (unsigned)C4 + (unsigned)yc) * 4u : kOOB;
      yv_next[j] = Act<T>::ld4r(ry2, off);
      if constexpr (kChain) {
        nv_next[j] = Act<T>::ld4r(rdn, off);
        if constexpr (NA) y1_nx[j] = Act<T>::ld4r(ry1n, off);
      }
    }
  };
  request_narrow(blockIdx.x);
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const unsigned p0 = (unsigned)t * TP;
    typename Act<T>::raw4 yv[NY], nv[kChain ? NY : 1], y1v[NA ? NY : 1];
#pragma unroll
    for (int j = 0; j < NY; ++j) {
      yv[j] = yv_next[j];
      if constexpr (kChain) {
        nv[j] = nv_next[j];
        if constexpr (NA) y1v[j] = y1_nx[j];
      }
    }
    typename Act<T>::raw4 xs[kF || kB1 ? NPASS : 1], dv[kHasD ? NPASS : 1];
#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
      const unsigned p = p0 + (unsigned)(q * PL + pl);
      const unsigned off = p < (unsigned)a.npix ? (p * (unsigned)C + (unsigned)c) * 4u : kOOB;
      if constexpr (kF || kB1)
        xs[q] = Act<T>::ld4r(rskip, (p < (unsigned)a.npix && has_skip) ? (p * (unsigned)a.c_skip + (unsigned)c) * 4u : kOOB);
      if constexpr (kHasD) dv[q] = Act<T>::ld4r(rd, off);
    }
    request_narrow(t + (int)gridDim.x);
    __syncthreads();                           // the previous tile's staged accumulators have been read (first tile: W is in LDS)
#pragma unroll
    for (int j = 0; j < NY; ++j) {
      const int px = (tid + 256 * j) / P4;
      const f32x4 yw = Act<T>::up4(yv[j]);
      f32x4 avv;
#pragma unroll
      for (int e = 0; e < 4; ++e) avv[e] = pre_bn_act(yw[e], m2[e], is2[e], g2[e], b2[e], al2[e], act2);
      f32x4 nw = {0.f, 0.f, 0.f, 0.f};
      if constexpr (kChain) {
        nw = Act<T>::up4(nv[j]);
        if constexpr (NA) {      // BN1' + PReLU1' backward, apply pass: dy1' = k1 * (d * act'(z) - m1 - xhat * m2)
          const f32x4 y1w = Act<T>::up4(y1v[j]);
          const unsigned pg = p0 + (unsigned)px;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xh = (y1w[e] - n_m[e]) * n_is[e];
            const float z = n_g[e] * xh + n_b[e];
            const float gr = !n_act ? 1.f : (z > 0.f ? 1.f : (z < 0.f ? n_al[e] : 0.f));
            nw[e] = n_k1[e] * (nw[e] * gr - n_m1[e] - xh * n_m2[e]);
          }
          if constexpr (kBf) nw = unpack_bf16x4(pack_bf16x4(nw));
          store4(rdno, pg < (unsigned)a.npix ? (pg * (unsigned)C4 + (unsigned)yc) * 4u : kOOB, nw);
        }
      }
      if constexpr (kBf) {      // rounded as the stored normalised tensor would have been
        *reinterpret_cast<u32x2*>(Alb + px * ABS + yc * 2) = pack_bf16x4(avv);
        if constexpr (kChain) *reinterpret_cast<u32x2*>(Anb + px * ABS + yc * 2) = pack_bf16x4(nw);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) Al[px * AS + yc + e] = avv[e];
        if constexpr (kChain) {
#pragma unroll
          for (int e = 0; e < 4; ++e) An[px * AS + yc + e] = nw[e];
        }
      }
    }
    __syncthreads();
    f32x16 acc[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[u][i] = 0.f;
    if constexpr (kBf) {
#pragma unroll
      for (int s2 = 0; s2 < K1; ++s2) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(Alb + (pw * 32 + r) * ABS + 32 * s2 + 16 * hh);
#pragma unroll
        for (int u = 0; u < NT; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, wf[u][s2], acc[u], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int s = 0; s < C4 / 2; ++s) {
        const int k = 2 * s + hh;
        const float av = Al[(pw * 32 + r) * AS + k];
#pragma unroll
        for (int u = 0; u < NT; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wq[u][s], acc[u], 0, 0, 0);
      }
    }
    if constexpr (kChain) {      // the data gradient of the next block's reduction, staged beside y3
      f32x16 accn[NT];
#pragma unroll
      for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) accn[u][i] = 0.f;
      if constexpr (kBf) {
#pragma unroll
        for (int s2 = 0; s2 < K1; ++s2) {
          const bf16x8 af = *reinterpret_cast<const bf16x8*>(Anb + (pw * 32 + r) * ABS + 32 * s2 + 16 * hh);
#pragma unroll
          for (int u = 0; u < NT; ++u) accn[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, wnf[u][s2], accn[u], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int s = 0; s < C4 / 2; ++s) {
          const int k = 2 * s + hh;
          const float av = An[(pw * 32 + r) * AS + k];
#pragma unroll
          for (int u = 0; u < NT; ++u) accn[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wnq[u][s], accn[u], 0, 0, 0);
        }
      }
#pragma unroll
      for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) Sn[(pw * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh) * SS + cw * (C / CWN) + u * 32 + r] = accn[u][i];
    }
    if constexpr (MODE == 0) {      // statistics straight from the accumulators: a lane owns 16 pixels of one channel per tile
      const bool full = p0 + TP <= (unsigned)a.npix;      // uniform
#pragma unroll
      for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const unsigned p = p0 + (unsigned)(pw * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh);
          const float yy = (full || p < (unsigned)a.npix) ? acc[u][i] : 0.f;
          lane_s[u][0] += yy;
          lane_s[u][1] += yy * yy;
        }
      continue;
    }
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float yy = acc[u][i];
        if constexpr (kBf) yy = (float)(bf16)yy;      // the value a stored y3 would hold (the statistics above take the unrounded one)
        St[(pw * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh) * SS + cw * (C / CWN) + u * 32 + r] = yy;
      }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
      const int lp = q * PL + pl;
      const unsigned p = p0 + (unsigned)lp;
      const bool live = p < (unsigned)a.npix;
      const f32x4 y3 = *reinterpret_cast<const f32x4*>(St + lp * SS + c);
      f32x4 xq = {0.f, 0.f, 0.f, 0.f}, dq = {0.f, 0.f, 0.f, 0.f};
      if constexpr (kF || kB1) xq = Act<T>::up4(xs[q]);
      if constexpr (kHasD) dq = Act<T>::up4(dv[q]);
      if constexpr (kChain) {      // dout = data gradient + the next block's skip gradient (bfloat16: rounded as the stored dout was)
        const f32x4 dg = *reinterpret_cast<const f32x4*>(Sn + lp * SS + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dq[e] += dg[e];
          if constexpr (kBf) dq[e] = (float)(bf16)dq[e];
        }
      }
      f32x4 dr = {1.f, 1.f, 1.f, 1.f};
      if constexpr (MODE != 0) {
        if (a.drop) dr = *reinterpret_cast<const f32x4*>(a.drop + (size_t)((live ? p : 0u) / (unsigned)a.hw) * C + c);
      }
      if constexpr (kB2) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xh = (y3[e] - m3[e]) * is3[e];
          o[e] = k1[e] * (dq[e] * dr[e] - q1[e] - xh * q2[e]);
        }
        store4(ro, live ? (p * (unsigned)C + (unsigned)c) * 4u : kOOB, o);
        if constexpr (kThru) {      // the tile goes back where this thread read y3 from (bfloat16: as the stored dmain is rounded)
          if (!live) o = f32x4{0.f, 0.f, 0.f, 0.f};
          if constexpr (kBf) *reinterpret_cast<u32x2*>(Dtb + lp * DBS + c * 2) = pack_bf16x4(o);      // (its own region: a bfloat16 row is narrower)
          else *reinterpret_cast<f32x4*>(St + lp * SS + c) = o;
        }
      } else {
        f32x4 o, ov = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xh = (y3[e] - m3[e]) * is3[e];
          float s = (g3[e] * xh + b3[e]) * dr[e];
          s += xq[e];
          if constexpr (kF) {
            o[e] = s > 0.f ? s : al3[e] * s;
          } else {
            if constexpr (WG) ov[e] = live ? (s > 0.f ? s : al3[e] * s) : 0.f;      // the block's output = the next block's input
            const float d = live ? dq[e] : 0.f;
            o[e] = d * (s > 0.f ? 1.f : (s < 0.f ? al3[e] : 0.f));
            const float dz = o[e] * dr[e];
            v[0][e] += dz;
            v[1][e] += dz * xh;
            v[2][e] += d * (s < 0.f ? s : 0.f);
          }
        }
        store4(ro, live ? (p * (unsigned)C + (unsigned)c) * 4u : kOOB, o);
        if constexpr (WG && kChain) {      // where this thread read y3 from (bfloat16: its own region, rounded as the stored output is)
          if constexpr (kBf) *reinterpret_cast<u32x2*>(Otb + lp * DBS + c * 2) = pack_bf16x4(ov);
          else *reinterpret_cast<f32x4*>(St + lp * SS + c) = ov;
        }
        if constexpr (kF) store4(ry3, live ? (p * (unsigned)C + (unsigned)c) * 4u : kOOB, y3);      // (empty descriptor: dropped)
        if constexpr (kFt) {      // the output tile, where this thread read y3 from (bfloat16: its own region, rounded as stored)
          if (!live) o = f32x4{0.f, 0.f, 0.f, 0.f};
          if constexpr (kBf) *reinterpret_cast<u32x2*>(Dtb + lp * DBS + c * 2) = pack_bf16x4(o);
          else *reinterpret_cast<f32x4*>(St + lp * SS + c) = o;
        }
      }
    }
    if constexpr (WG && kChain) {      // dW1' += out^T * dy1': rows = this wave's 32 output channels, columns = the C4 reduced ones
      __syncthreads();
      if constexpr (kBf) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
          accw = __builtin_amdgcn_mfma_f32_32x32x16_bf16(operand_t(Otb + (pw * 32 + 16 * s2) * DBS + cw * 32 * 2, DBS),
                                                         operand_t(Anb + (pw * 32 + 16 * s2) * ABS, ABS), accw, 0, 0, 0);
      } else {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          const int k = pw * 32 + 2 * s + hh;
          accw = __builtin_amdgcn_mfma_f32_32x32x2f32(St[k * SS + cw * 32 + r], r < C4 ? An[k * AS + r] : 0.f, accw, 0, 0, 0);
        }
      }
    }
    if constexpr (kThru || kFt) {
      __syncthreads();
      if constexpr (WG) {      // dW3 += a2^T * dmain: rows = the C4 input channels, columns = this wave's 32 output channels
        if constexpr (kBf) {
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2)
            accw = __builtin_amdgcn_mfma_f32_32x32x16_bf16(operand_t(Alb + (pw * 32 + 16 * s2) * ABS, ABS),
                                                           operand_t(Dtb + (pw * 32 + 16 * s2) * DBS + cw * 32 * 2, DBS), accw, 0, 0, 0);
        } else {
#pragma unroll
          for (int s = 0; s < 16; ++s) {
            const int k = pw * 32 + 2 * s + hh;
            accw = __builtin_amdgcn_mfma_f32_32x32x2f32(r < C4 ? Al[k * AS + r] : 0.f, St[k * SS + cw * 32 + r], accw, 0, 0, 0);
          }
        }
      }
      f32x16 acd;
#pragma unroll
      for (int i = 0; i < 16; ++i) acd[i] = 0.f;
      if constexpr (kBf) {
#pragma unroll
        for (int s2 = 0; s2 < K3; ++s2)
          acd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(Dtb + (pw * 32 + r) * DBS + (cw * KW + 16 * s2 + 8 * hh) * 2),
                                                        wtf[s2], acd, 0, 0, 0);
      } else {
        // one 16-byte read feeds four matrix instructions (a column read of 4 bytes per lane hits every bank twice: the
        // rows are a multiple of 16 bytes apart); lane half hh takes channels 4 hh .. 4 hh + 3 of each group of eight
        static_assert(KW % 8 == 0, "shape");
#pragma unroll
        for (int g8 = 0; g8 < KW / 8; ++g8) {
          const f32x4 a4 = *reinterpret_cast<const f32x4*>(St + (pw * 32 + r) * SS + cw * KW + 8 * g8 + 4 * hh);
#pragma unroll
          for (int j = 0; j < 4; ++j) acd = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j], wtq[4 * g8 + j], acd, 0, 0, 0);
        }
      }
      if constexpr (!kBf) __syncthreads();      // Pn lies over St, which the float32 product above was still reading
#pragma unroll
      for (int i = 0; i < 16; ++i) Pn[(cw * TP + pw * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh) * WTS + r] = acd[i];
      __syncthreads();
      // the thread that normalised (pixel, four input channels) on the way in finishes them: raw y2 is still in its registers
      const int px = tid / P4;
      const bool live2 = p0 + (unsigned)px < (unsigned)a.npix;
      const f32x4 yw = Act<T>::up4(yv[0]);
      f32x4 da;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float d = Pn[px * WTS + yc + e];
#pragma unroll
        for (int k2 = 1; k2 < KS; ++k2) d += Pn[(k2 * TP + px) * WTS + yc + e];
        da[e] = d;
        if constexpr (kFt && NA) {      // inference phase: BN1' (fixed statistics) + PReLU1' applied here, the activated tensor is stored
          da[e] = pre_bn_act(d, n_m[e], n_is[e], n_g[e], n_b[e], n_al[e], n_act);
          continue;
        }
        if constexpr (kFt) {      // BN1' statistics of the next block, from the unrounded values as the convolution kernels take them
          const float yy = live2 ? d : 0.f;
          v2[0][e] += yy;
          v2[1][e] += yy * yy;
          continue;
        }
        const float xh = (yw[e] - m2[e]) * is2[e];
        const float z = g2[e] * xh + b2[e];
        const float gr = !act2 ? 1.f : (z > 0.f ? 1.f : (z < 0.f ? al2[e] : 0.f));
        const float dz = live2 ? d * gr : 0.f;
        v2[0][e] += dz;
        v2[1][e] += dz * xh;
        v2[2][e] += (live2 && act2) ? d * (z < 0.f ? z : 0.f) : 0.f;
      }
      store4(rda, live2 ? ((p0 + (unsigned)px) * (unsigned)C4 + (unsigned)yc) * 4u : kOOB, da);
    }
  }
  if constexpr (WG) {         // this wave's block of the workgroup's partial weight gradient (C4 = 16: half of the block is padding)
    float* wp = a.wpart + ((size_t)blockIdx.x * PWN + pw) * (size_t)(C4 * C);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
      if constexpr (kThru) {
        if (row < C4) wp[row * C + cw * 32 + r] = accw[i];
      } else {
        if (r < C4) wp[(cw * 32 + row) * C4 + r] = accw[i];
      }
    }
  }
  if constexpr (kThru || (kFt && !NA)) {      // one row of BN2-backward sums (B2t) / BN1' statistics (Ft) per workgroup, pixels added in a fixed order
    constexpr int N2 = kThru ? 3 : 2;
    __syncthreads();
    float* red = St;          // [N2][TP][C4]
    const int px = tid / P4;
#pragma unroll
    for (int k = 0; k < N2; ++k) *reinterpret_cast<f32x4*>(red + (k * TP + px) * C4 + yc) = v2[k];
    __syncthreads();
    if (tid < C4) {
#pragma unroll
      for (int k = 0; k < N2; ++k) {
        float t2 = 0.f;
        for (int p2 = 0; p2 < TP; ++p2) t2 += red[(k * TP + p2) * C4 + tid];
        a.part2[((size_t)blockIdx.x * N2 + k) * C4 + tid] = t2;
      }
    }
  }
  if constexpr (MODE == 0) {  // lane halves, then the two pixel-half waves of a channel half, in a fixed order
    __syncthreads();
    float* red = St;          // [2 sums][PWN pixel blocks][C]
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float t2 = lane_s[u][k] + __shfl_xor(lane_s[u][k], 32, 64);
        if (hh == 0) red[(k * PWN + pw) * C + cw * (C / CWN) + u * 32 + r] = t2;
      }
    __syncthreads();
    if (tid < C) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        float t2 = red[(k * PWN) * C + tid];
#pragma unroll
        for (int b = 1; b < PWN; ++b) t2 += red[(k * PWN + b) * C + tid];
        a.part[((size_t)blockIdx.x * 2 + k) * C + tid] = t2;
      }
    }
  } else if constexpr (kSums) {      // one row of column sums per workgroup, pixel lanes added in a fixed order
    __syncthreads();
    float* red = St;          // [NS][PL][C]
#pragma unroll
    for (int k = 0; k < NS; ++k) *reinterpret_cast<f32x4*>(red + (k * PL + pl) * C + c) = v[k];
    __syncthreads();
    if (tid < C) {
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        float s = 0.f;
        for (int p2 = 0; p2 < PL; ++p2) s += red[(k * PL + p2) * C + tid];
        a.part[((size_t)blockIdx.x * NS + k) * C + tid] = s;
      }
    }
  }
}

static constexpr int kExpandMaxGrid = 1024;     // also the largest number of partial-sum rows

static bool expand_shape_ok(const ssmi_expand_desc* d) {
  return d && d->n > 0 && d->h > 0 && d->w > 0 && ((d->cin == 16 && d->cout == 64) || (d->cin == 32 && d->cout == 128)) &&
         (int64_t)d->n * d->h * d->w * d->cout * 4 < 0x7FFFFFFFLL && (d->dtype == SSMI_F32 || d->dtype == SSMI_BF16);
}

static int expand_grid(const ssmi_expand_desc* d) {
  const int tp = d->cout == 128 ? 32 : 64;
  const int ntiles = (d->n * d->h * d->w + tp - 1) / tp;
  static const int gmax = getenv("SSMI_EXPAND_GRID") ? atoi(getenv("SSMI_EXPAND_GRID")) : 512;      // measurement switch; ENet step with 256 / 384 / 512 / 768 / 1024: 8.81 / 8.70 / 8.49 / 8.51 / 8.52 ms
  const int cap = gmax < kExpandMaxGrid ? gmax : kExpandMaxGrid;
  return ntiles < cap ? ntiles : cap;
}

// rows of a fused weight gradient's partial sums: one per workgroup and 32-pixel block of its tiles
static int expand_weight_rows(const ssmi_expand_desc* d) { return expand_grid(d) * (d->cout == 128 ? 1 : 2); }

template <int MODE, bool WG = false, bool NA = false>
static ssmi_status launch_expand(const ssmi_expand_desc* d, const ExpandArgs& a, int* rows, hipStream_t st) {
  // (the B1c mode alone at 256 / 384 workgroups instead of 512: 7.71 / 7.72 against 7.70 ms per float32 ENet step - at 210
  // registers its two waves per SIMD leave the weight-gradient chain no room while it runs, and what half the waves lose the
  // chain gains; three waves per SIMD would need 168 registers, which this mode does not reach without spilling)
  const int grid = expand_grid(d);
  if (rows) *rows = grid;
  return dispatch_dtype(d->dtype, [&](auto tag) -> ssmi_status {
    using T = decltype(tag);
    if constexpr (sizeof(T) == 4 && WG && NA) {      // (float32: the three products' operands + both parameter sets exceed 256 registers)
      set_error("expand_merge_bwd_through: wpart1_next together with next_apply is available for bfloat16 storage only");
      return SSMI_E_UNSUPPORTED;
    } else
    if (d->cin == 16) {
      constexpr size_t lds = (size_t)ExpandLds<T, 16, 64, MODE, WG>::total * 4;
      hipLaunchKernelGGL((expand_kernel<T, 16, 64, MODE, WG, NA>), dim3(grid), dim3(256), lds, st, a);
    } else {
      // (the reduction scratch of the summing modes, 3 sums x 8 pixel lanes x 128 channels, fits the staged tile)
      constexpr size_t lds = (size_t)ExpandLds<T, 32, 128, MODE, WG>::total * 4;
      hipLaunchKernelGGL((expand_kernel<T, 32, 128, MODE, WG, NA>), dim3(grid), dim3(256), lds, st, a);
    }
    SSMI_LAUNCH_CHECK();
    return SSMI_OK;
  });
}

static bool aligned16(std::initializer_list<const void*> ps) {
  for (const void* p : ps)
    if (reinterpret_cast<uintptr_t>(p) & 15) return false;
  return true;
}

static ExpandArgs expand_args(const ssmi_expand_desc* d, const void* y2, const ssmi_bn_params* pre, const float* pre_alpha, const float* w) {
  ExpandArgs a{};
  a.npix = d->n * d->h * d->w;
  a.hw = d->h * d->w;
  a.y2 = y2;
  a.pre[0] = pre->mean; a.pre[1] = pre->invstd; a.pre[2] = pre->gamma; a.pre[3] = pre->beta; a.pre[4] = pre_alpha;
  a.w = w;
  return a;
}

}  // namespace ssmi

using namespace ssmi;

extern "C" {

int32_t ssmi_expand_supported(const ssmi_expand_desc* d) { return expand_shape_ok(d) ? 1 : 0; }

size_t ssmi_expand_workspace_bytes(const ssmi_expand_desc* d) {
  if (!expand_shape_ok(d)) return 0;
  // partial rows + the apply pass's coefficients, for BN3 and (ssmi_expand_merge_bwd_through) for BN2
  return ((size_t)kExpandMaxGrid * 3 + 3) * (size_t)(d->cout + d->cin) * sizeof(float);
}

ssmi_status ssmi_expand_bn_stats(const ssmi_expand_desc* d, const void* x_raw, const ssmi_bn_params* pre, const float* pre_alpha,
                                 const float* w, float eps, float momentum, int32_t unbiased_moving, float* moving_mean,
                                 float* moving_var, float* save_mean, float* save_invstd, void* ws, size_t ws_bytes, void* stream) {
  SSMI_REQUIRE(expand_shape_ok(d), SSMI_E_UNSUPPORTED, "expand_bn_stats: 16 -> 64 or 32 -> 128 channel expansions only");
  SSMI_REQUIRE(x_raw && pre && pre->mean && pre->invstd && pre->gamma && pre->beta && w && save_mean && save_invstd, SSMI_E_INVALID_ARG,
               "expand_bn_stats: NULL argument");
  SSMI_REQUIRE(ws && ws_bytes >= ssmi_expand_workspace_bytes(d), SSMI_E_WORKSPACE, "expand_bn_stats: workspace too small");
  SSMI_REQUIRE(aligned16({x_raw, w, pre->mean, pre->invstd, pre->gamma, pre->beta, pre_alpha, ws}), SSMI_E_INVALID_ARG,
               "expand_bn_stats: pointers must be 16-byte aligned");
  ExpandArgs a = expand_args(d, x_raw, pre, pre_alpha, w);
  a.part = static_cast<float*>(ws);
  int rows = 0;
  ssmi_status s = launch_expand<0>(d, a, &rows, as_stream(stream));
  if (s) return s;
  return launch_bn_stats_final(a.part, rows, d->cout, a.npix, eps, momentum, unbiased_moving, moving_mean, moving_var, save_mean,
                               save_invstd, as_stream(stream));
}

ssmi_status ssmi_expand_merge_fwd(const ssmi_expand_desc* d, const void* x_raw, const ssmi_bn_params* pre, const float* pre_alpha,
                                  const float* w, const ssmi_bn_params* bn3, const float* drop, const void* skip, int32_t c_skip,
                                  const float* alpha, void* out, void* y3_out, void* stream) {
  SSMI_REQUIRE(expand_shape_ok(d), SSMI_E_UNSUPPORTED, "expand_merge_fwd: 16 -> 64 or 32 -> 128 channel expansions only");
  SSMI_REQUIRE(x_raw && pre && pre->mean && pre->invstd && pre->gamma && pre->beta && w && bn3 && bn3->mean && bn3->invstd && bn3->gamma &&
               bn3->beta && skip && alpha && out, SSMI_E_INVALID_ARG, "expand_merge_fwd: NULL argument");
  SSMI_REQUIRE(c_skip > 0 && c_skip <= d->cout && c_skip % 4 == 0, SSMI_E_SHAPE, "expand_merge_fwd: skip channels must be a multiple of 4 within the output's");
  SSMI_REQUIRE(aligned16({x_raw, w, pre->mean, pre->invstd, pre->gamma, pre->beta, pre_alpha, bn3->mean, bn3->invstd, bn3->gamma, bn3->beta,
                          drop, skip, alpha, out}), SSMI_E_INVALID_ARG, "expand_merge_fwd: pointers must be 16-byte aligned");
  ExpandArgs a = expand_args(d, x_raw, pre, pre_alpha, w);
  a.mean3 = bn3->mean; a.invstd3 = bn3->invstd; a.gamma3 = bn3->gamma; a.beta3 = bn3->beta; a.alpha3 = alpha;
  a.drop = drop; a.skip = skip; a.c_skip = c_skip; a.out = out;
  a.y3_out = y3_out;
  SSMI_REQUIRE(aligned16({y3_out}), SSMI_E_INVALID_ARG, "expand_merge_fwd: pointers must be 16-byte aligned");
  return launch_expand<1>(d, a, nullptr, as_stream(stream));
}

// dout given (dy_next, w_next null): the reducing pass reads it; dy_next and w_next given: dout is formed per tile as
// dy_next * w_next^T + dout_or_dsum_next and never stored.
static ssmi_status expand_merge_bwd_impl(const char* who, const ssmi_expand_desc* d, const void* dout, const void* dy_next, const float* w_next,
                                         const void* x_raw, const ssmi_bn_params* pre, const float* pre_alpha, const float* w,
                                         const ssmi_bn_params* bn3, const float* drop, const void* skip, int32_t c_skip, const float* alpha,
                                         void* dmain, void* dsum, float* dgamma3, float* dbeta3, float* dalpha, void* ws, size_t ws_bytes,
                                         void* stream, void* dy2 = nullptr, float* dgamma2 = nullptr, float* dbeta2 = nullptr,
                                         float* dalpha2 = nullptr, float* wpart3 = nullptr, float* wpart1n = nullptr,
                                         const ssmi_expand_next_apply* na = nullptr) {
  SSMI_REQUIRE(expand_shape_ok(d), SSMI_E_UNSUPPORTED, "%s: 16 -> 64 or 32 -> 128 channel expansions only", who);
  SSMI_REQUIRE(dout && x_raw && pre && pre->mean && pre->invstd && pre->gamma && pre->beta && w && bn3 && bn3->mean && bn3->invstd &&
               bn3->gamma && bn3->beta && skip && alpha && (dmain || wpart3) && dsum && dgamma3 && dbeta3 && dalpha, SSMI_E_INVALID_ARG,
               "%s: NULL argument", who);
  SSMI_REQUIRE((!wpart3 || dy2) && (!wpart1n || dy_next), SSMI_E_INVALID_ARG,
               "%s: a fused weight gradient needs the data gradient it rides on (dy2 for wpart3, dy_next for wpart1_next)", who);
  SSMI_REQUIRE(c_skip > 0 && c_skip <= d->cout && c_skip % 4 == 0, SSMI_E_SHAPE, "%s: skip channels must be a multiple of 4 within the output's", who);
  SSMI_REQUIRE(ws && ws_bytes >= ssmi_expand_workspace_bytes(d), SSMI_E_WORKSPACE, "%s: workspace too small", who);
  SSMI_REQUIRE(aligned16({dout, dy_next, w_next, x_raw, w, pre->mean, pre->invstd, pre->gamma, pre->beta, pre_alpha, bn3->mean, bn3->invstd,
                          bn3->gamma, bn3->beta, drop, skip, alpha, dmain, dsum, ws, wpart3, wpart1n}), SSMI_E_INVALID_ARG,
               "%s: pointers must be 16-byte aligned", who);
  hipStream_t st = as_stream(stream);
  ExpandArgs a = expand_args(d, x_raw, pre, pre_alpha, w);
  a.mean3 = bn3->mean; a.invstd3 = bn3->invstd; a.gamma3 = bn3->gamma; a.beta3 = bn3->beta; a.alpha3 = alpha;
  a.drop = drop; a.skip = skip; a.c_skip = c_skip; a.dout = dout;
  a.dy_next = dy_next; a.w_next = w_next;
  if (na) {
    SSMI_REQUIRE(dy_next && na->y1 && na->bn1 && na->bn1->mean && na->bn1->invstd && na->bn1->gamma && na->bn1->beta && na->coef,
                 SSMI_E_INVALID_ARG, "%s: the deferred apply pass needs dy_next, y1, bn1 and coef", who);
    SSMI_REQUIRE(aligned16({na->y1, na->bn1->mean, na->bn1->invstd, na->bn1->gamma, na->bn1->beta, na->alpha, na->coef}), SSMI_E_INVALID_ARG,
                 "%s: pointers must be 16-byte aligned", who);
    a.y1_next = na->y1;
    a.bn1_next[0] = na->bn1->mean; a.bn1_next[1] = na->bn1->invstd; a.bn1_next[2] = na->bn1->gamma; a.bn1_next[3] = na->bn1->beta;
    a.bn1_next[4] = na->alpha;
    a.coef_next = na->coef;
  }
  a.out = dsum;
  a.part = static_cast<float*>(ws);
  float* coef = a.part + (size_t)kExpandMaxGrid * 3 * d->cout;
  int rows = 0;
  a.wpart = wpart1n;
  ssmi_status s;
  if (!dy_next) s = launch_expand<2>(d, a, &rows, st);
  else if (wpart1n) s = na ? launch_expand<4, true, true>(d, a, &rows, st) : launch_expand<4, true>(d, a, &rows, st);
  else s = na ? launch_expand<4, false, true>(d, a, &rows, st) : launch_expand<4>(d, a, &rows, st);
  if (s) return s;
  s = launch_bn_bwd_final(a.part, rows, d->cout, a.npix, bn3->gamma, bn3->invstd, dgamma3, dbeta3, dalpha, coef, st);
  if (s) return s;
  ExpandArgs b = expand_args(d, x_raw, pre, pre_alpha, w);
  b.mean3 = bn3->mean; b.invstd3 = bn3->invstd;
  b.drop = drop; b.dsum_in = dsum; b.coef = coef; b.out = dmain;
  if (!dy2) return launch_expand<3>(d, b, nullptr, st);
  b.da2 = dy2;
  b.part2 = coef + 3 * d->cout;
  float* coef2 = b.part2 + (size_t)kExpandMaxGrid * 3 * d->cin;
  b.wpart = wpart3;
  s = wpart3 ? launch_expand<5, true>(d, b, &rows, st) : launch_expand<5>(d, b, &rows, st);
  if (s) return s;
  return dispatch_dtype(d->dtype, [&](auto tag) -> ssmi_status {      // BN2 / PReLU2 backward in place: da2 -> dy2
    using T = decltype(tag);
    return launch_bn_bwd_tail<T>(b.part2, rows, static_cast<const T*>(dy2), d->cin, static_cast<const T*>(x_raw), d->cin, pre->mean,
                                 pre->invstd, pre->gamma, pre->beta, pre_alpha ? SSMI_ACT_PRELU : SSMI_ACT_NONE, pre_alpha, 0.f, nullptr,
                                 (int64_t)a.hw, static_cast<T*>(dy2), dgamma2, dbeta2, dalpha2, (int64_t)a.npix, d->cin, coef2, st);
  });
}

ssmi_status ssmi_expand_merge_fwd_through(const ssmi_expand_desc* d, const void* x_raw, const ssmi_bn_params* pre, const float* pre_alpha,
                                          const float* w, const ssmi_bn_params* bn3, const float* drop, const void* skip, int32_t c_skip,
                                          const float* alpha, void* out, void* y3_out, const float* w_next, void* y1_next, float eps,
                                          float momentum, int32_t unbiased_moving, float* moving_mean1, float* moving_var1,
                                          float* save_mean1, float* save_invstd1, void* ws, size_t ws_bytes, void* stream) {
  SSMI_REQUIRE(expand_shape_ok(d), SSMI_E_UNSUPPORTED, "expand_merge_fwd_through: 16 -> 64 or 32 -> 128 channel expansions only");
  SSMI_REQUIRE(x_raw && pre && pre->mean && pre->invstd && pre->gamma && pre->beta && w && bn3 && bn3->mean && bn3->invstd && bn3->gamma &&
               bn3->beta && skip && alpha && out && w_next && y1_next && save_mean1 && save_invstd1, SSMI_E_INVALID_ARG,
               "expand_merge_fwd_through: NULL argument");
  SSMI_REQUIRE(c_skip > 0 && c_skip <= d->cout && c_skip % 4 == 0, SSMI_E_SHAPE, "expand_merge_fwd_through: skip channels must be a multiple of 4 within the output's");
  SSMI_REQUIRE(ws && ws_bytes >= ssmi_expand_workspace_bytes(d), SSMI_E_WORKSPACE, "expand_merge_fwd_through: workspace too small");
  SSMI_REQUIRE(aligned16({x_raw, w, pre->mean, pre->invstd, pre->gamma, pre->beta, pre_alpha, bn3->mean, bn3->invstd, bn3->gamma, bn3->beta,
                          drop, skip, alpha, out, y3_out, w_next, y1_next, ws}), SSMI_E_INVALID_ARG,
               "expand_merge_fwd_through: pointers must be 16-byte aligned");
  ExpandArgs a = expand_args(d, x_raw, pre, pre_alpha, w);
  a.mean3 = bn3->mean; a.invstd3 = bn3->invstd; a.gamma3 = bn3->gamma; a.beta3 = bn3->beta; a.alpha3 = alpha;
  a.drop = drop; a.skip = skip; a.c_skip = c_skip; a.out = out;
  a.y3_out = y3_out;
  a.w_next = w_next; a.da2 = y1_next;
  a.part2 = static_cast<float*>(ws);
  int rows = 0;
  ssmi_status s = launch_expand<6>(d, a, &rows, as_stream(stream));
  if (s) return s;
  return launch_bn_stats_final(a.part2, rows, d->cin, a.npix, eps, momentum, unbiased_moving, moving_mean1, moving_var1, save_mean1,
                               save_invstd1, as_stream(stream));
}

ssmi_status ssmi_expand_merge_fwd_through_infer(const ssmi_expand_desc* d, const void* x_raw, const ssmi_bn_params* pre, const float* pre_alpha,
                                                const float* w, const ssmi_bn_params* bn3, const float* drop, const void* skip,
                                                int32_t c_skip, const float* alpha, void* out, const float* w_next,
                                                const ssmi_bn_params* bn1_next, const float* alpha1_next, void* a1_next, void* stream) {
  SSMI_REQUIRE(expand_shape_ok(d), SSMI_E_UNSUPPORTED, "expand_merge_fwd_through_infer: 16 -> 64 or 32 -> 128 channel expansions only");
  SSMI_REQUIRE(x_raw && pre && pre->mean && pre->invstd && pre->gamma && pre->beta && w && bn3 && bn3->mean && bn3->invstd && bn3->gamma &&
               bn3->beta && skip && alpha && out && w_next && a1_next && bn1_next && bn1_next->mean && bn1_next->invstd && bn1_next->gamma &&
               bn1_next->beta, SSMI_E_INVALID_ARG, "expand_merge_fwd_through_infer: NULL argument");
  SSMI_REQUIRE(c_skip > 0 && c_skip <= d->cout && c_skip % 4 == 0, SSMI_E_SHAPE,
               "expand_merge_fwd_through_infer: skip channels must be a multiple of 4 within the output's");
  SSMI_REQUIRE(aligned16({x_raw, w, pre->mean, pre->invstd, pre->gamma, pre->beta, pre_alpha, bn3->mean, bn3->invstd, bn3->gamma, bn3->beta,
                          drop, skip, alpha, out, w_next, a1_next, bn1_next->mean, bn1_next->invstd, bn1_next->gamma, bn1_next->beta,
                          alpha1_next}), SSMI_E_INVALID_ARG, "expand_merge_fwd_through_infer: pointers must be 16-byte aligned");
  ExpandArgs a = expand_args(d, x_raw, pre, pre_alpha, w);
  a.mean3 = bn3->mean; a.invstd3 = bn3->invstd; a.gamma3 = bn3->gamma; a.beta3 = bn3->beta; a.alpha3 = alpha;
  a.drop = drop; a.skip = skip; a.c_skip = c_skip; a.out = out;
  a.w_next = w_next; a.da2 = a1_next;
  a.bn1_next[0] = bn1_next->mean; a.bn1_next[1] = bn1_next->invstd; a.bn1_next[2] = bn1_next->gamma; a.bn1_next[3] = bn1_next->beta;
  a.bn1_next[4] = alpha1_next;
  return launch_expand<6, false, true>(d, a, nullptr, as_stream(stream));
}

ssmi_status ssmi_expand_merge_bwd(const ssmi_expand_desc* d, const void* dout, const void* x_raw, const ssmi_bn_params* pre,
                                  const float* pre_alpha, const float* w, const ssmi_bn_params* bn3, const float* drop, const void* skip,
                                  int32_t c_skip, const float* alpha, void* dmain, void* dsum, float* dgamma3, float* dbeta3, float* dalpha, void* ws,
                                  size_t ws_bytes, void* stream) {
  return expand_merge_bwd_impl("expand_merge_bwd", d, dout, nullptr, nullptr, x_raw, pre, pre_alpha, w, bn3, drop, skip, c_skip, alpha, dmain,
                               dsum, dgamma3, dbeta3, dalpha, ws, ws_bytes, stream);
}

ssmi_status ssmi_expand_merge_bwd_chained(const ssmi_expand_desc* d, const void* dy_next, const float* w_next, const void* dsum_next,
                                          const void* x_raw, const ssmi_bn_params* pre, const float* pre_alpha, const float* w,
                                          const ssmi_bn_params* bn3, const float* drop, const void* skip, int32_t c_skip, const float* alpha,
                                          void* dmain, void* dsum, float* dgamma3, float* dbeta3, float* dalpha, void* ws, size_t ws_bytes,
                                          void* stream) {
  SSMI_REQUIRE(dy_next && w_next && dsum_next, SSMI_E_INVALID_ARG, "expand_merge_bwd_chained: NULL argument");
  return expand_merge_bwd_impl("expand_merge_bwd_chained", d, dsum_next, dy_next, w_next, x_raw, pre, pre_alpha, w, bn3, drop, skip, c_skip,
                               alpha, dmain, dsum, dgamma3, dbeta3, dalpha, ws, ws_bytes, stream);
}

ssmi_status ssmi_expand_merge_bwd_through(const ssmi_expand_desc* d, const void* dout, const void* dy_next, const float* w_next,
                                          const void* dsum_next, const void* x_raw, const ssmi_bn_params* pre, const float* pre_alpha,
                                          const float* w, const ssmi_bn_params* bn3, const float* drop, const void* skip, int32_t c_skip,
                                          const float* alpha, void* dmain, void* dsum, float* dgamma3, float* dbeta3, float* dalpha,
                                          void* dy2, float* dgamma2, float* dbeta2, float* dalpha2, float* wpart3, float* wpart1_next,
                                          const ssmi_expand_next_apply* next_apply, void* ws, size_t ws_bytes, void* stream) {
  SSMI_REQUIRE((dout != nullptr) != (dy_next != nullptr) && (!dy_next || (w_next && dsum_next)), SSMI_E_INVALID_ARG,
               "expand_merge_bwd_through: give dout, or dy_next + w_next + dsum_next");
  SSMI_REQUIRE((dy2 || wpart1_next || next_apply) && (!dy2 || (dgamma2 && dbeta2 && (dalpha2 || !pre_alpha))), SSMI_E_INVALID_ARG,
               "expand_merge_bwd_through: NULL argument");
  SSMI_REQUIRE(aligned16({dy2}), SSMI_E_INVALID_ARG, "expand_merge_bwd_through: pointers must be 16-byte aligned");
  return expand_merge_bwd_impl("expand_merge_bwd_through", d, dout ? dout : dsum_next, dy_next, w_next, x_raw, pre, pre_alpha, w, bn3, drop,
                               skip, c_skip, alpha, dmain, dsum, dgamma3, dbeta3, dalpha, ws, ws_bytes, stream, dy2, dgamma2, dbeta2, dalpha2, wpart3,
                               wpart1_next, next_apply);
}

size_t ssmi_expand_weight_partial_bytes(const ssmi_expand_desc* d) {
  if (!expand_shape_ok(d)) return 0;
  return (size_t)kExpandMaxGrid * 2 * d->cin * d->cout * sizeof(float);
}

ssmi_status ssmi_expand_weight_reduce(const ssmi_expand_desc* d, const float* wpart, float* dw, void* stream) {
  SSMI_REQUIRE(expand_shape_ok(d), SSMI_E_UNSUPPORTED, "expand_weight_reduce: 16 -> 64 or 32 -> 128 channel expansions only");
  SSMI_REQUIRE(wpart && dw && aligned16({wpart, dw}), SSMI_E_INVALID_ARG, "expand_weight_reduce: NULL or misaligned argument");
  launch_wg_reduce(wpart, expand_weight_rows(d), d->cin * d->cout, d->cin * d->cout, 1, dw, nullptr, as_stream(stream));
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

}  // extern "C"

// LDS-tiled implicit GEMM for the wide 3x3 layers of U-Net / SegNet with bfloat16 storage (stride 1, dilation 1, SAME
// padding, input channels in whole 32-blocks, output channels in whole 64-blocks; any map size, border tiles are masked):
//   y[p, co] = bias[co] + sum_{tap, ci} x[pix(p, tap), ci] * w[tap, ci, co]      on v_mfma_f32_32x32x16_bf16.
// Both operands go through LDS.  A workgroup (256 threads, four waves) owns a TH x 16 tile of output pixels of one image
// (TH = 16, or 8 for layers that would otherwise launch fewer than two workgroups per CU) and BN = 128 or 64 output channels.
// For every 32-channel chunk of the input it stages the tile's (TH + 2) x 18 halo patch ONCE (80-byte pixel rows: 64 bytes of
// channels + 16 of padding against bank conflicts) and then walks the nine taps: a tap only shifts the window the A fragments
// are read from, so the nine taps re-read LDS, not L2.  The kernel rows of a step (tap, chunk) - [BN couts][32 channels] from
// the pack [tap][cout][cin] - are staged in a second, double-buffered area; they are requested three steps ahead into a ring
// of register sets, the next chunk's patch nine steps ahead.
// Waves split the tile 2 x 2 (BN = 128, TH = 16: 128 pixels x 64 couts each, 8 accumulator tiles) or 4 x 1 (BN = 64),
// which keeps LDS reads per matrix instruction at 0.75 - 1 ds_read_b128.
// Epilogue: the accumulators go through LDS once more (float32, 64 pixels per pass) so that bias, rounding, the stores and
// the fused reductions run in the elementwise layout - a thread owns four consecutive channels of a pixel, 8-byte accesses:
//   EPI 1  BatchNormalization statistics: per-workgroup column sums of y and y^2 (unrounded) -> part[tile][2][cout]
//   EPI 2  data-gradient form (forward convolution over dy with the data-gradient pack) with the BatchNormalization
//          backward sums of the layer in front (dz, dz * xhat, d * min(z, 0)) -> bn.part[tile][3][cout]
//   EPI 3  data-gradient form with a skip-gradient addend (and the sums of EPI 2 when bn.x is given)
//   EPI 4  inference phase: BatchNormalization with fixed statistics + activation applied before the store
// The same arithmetic as conv_fwd_kernel's epilogues, summed in a different (fixed) order.
#include "common.h"

namespace ssmi {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifndef TILED_ABLATE
#define TILED_ABLATE 0      // microbenchmark builds only: 1 = no reduction loop, 2 = no output stores, 3 = neither (results meaningless)
#endif
static constexpr int kTW = 16, kPW = kTW + 2;      // tiles are 16 pixels wide and TH = 16 or 8 rows high
static constexpr int kRowB = 80;                    // bytes per staged pixel / kernel row

struct TiledArgs {
  int n, h, w, cin, cout, x_ld, y_ld;
  const bf16* x;
  const bf16* wpack;      // [9][cout][cin]
  const float* bias;
  bf16* y;
  float* part;            // EPI 1
  BnBwdArgs bn;           // EPI 2: add, x, ld, mean, invstd, gamma, beta, alpha, slope, act, drop, hw, part
};

__device__ __forceinline__ unsigned spread_over_xcds(unsigned b, unsigned n) {      // a bijection of [0, n) for any n
  const unsigned q = n >> 3, r = n & 7u, x = b & 7u;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}
// CW consecutive bfloat16 channels of one pixel as they travel between HBM and registers in the epilogue
template <int CW> struct Row;
template <> struct Row<8> {
  typedef u32x4 raw;
  static __device__ __forceinline__ raw ld(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0)); }
  static __device__ __forceinline__ void st(__amdgpu_buffer_rsrc_t r, unsigned off, raw v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 0); }
};
template <> struct Row<4> {
  typedef u32x2 raw;
  static __device__ __forceinline__ raw ld(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0)); }
  static __device__ __forceinline__ void st(__amdgpu_buffer_rsrc_t r, unsigned off, raw v) { __builtin_amdgcn_raw_buffer_store_b64(v, r, off, 0, 0); }
};
__device__ __forceinline__ bf16x8 lds_frag(const unsigned char* p) { return *reinterpret_cast<const bf16x8*>(p); }

template <int BN, int WM, int WN, int EPI, int TH>
__global__ __launch_bounds__(256, BN == 64 ? 3 : 2) void conv3x3_tiled_bf16(TiledArgs a) {
  constexpr int kPatchPix = (TH + 2) * kPW, kABytes = kPatchPix * kRowB, BM = TH * kTW;      // 16 rows: 324 pixels, 25920 bytes
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // The 128-wide variants keep TWO patch images: the next chunk's patch is then fetched and committed in two halves in the
  // middle of the current chunk (half the staging registers, no extra barrier at the chunk boundary).  The 64-wide variants
  // have the registers and want three workgroups per CU, i.e. the smaller LDS footprint.
  constexpr bool ADB = BN == 128;
  constexpr int kABufs = ADB ? 2 : 1;
  unsigned char* const Bs = smem + kABufs * kABytes;   // two buffers of BN * kRowB bytes
  constexpr int kBBytes = BN * kRowB;
  constexpr int WPX = BM / WM;                       // pixels per wave
  constexpr int MT = WPX / 32, NTL = BN / WN / 32;
  static_assert(WM * WN == 4 && MT >= 1 && NTL >= 1, "wave layout");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;

  // tile coordinates
  const int tiles_x = (a.w + kTW - 1) / kTW, tiles_y = (a.h + TH - 1) / TH;      // border tiles may be partial
  const int ncb = a.cout / BN;
  // The workgroups of one XCD (b and b + 8 share one, and its L2) take a contiguous range of (tile, column block) pairs,
  // column block fastest: the column blocks of a tile re-read its patch from that L2, neighbouring tiles share halo rows.
  const unsigned v = spread_over_xcds(blockIdx.x, gridDim.x);
  const int t = (int)(v / (unsigned)ncb);
  const int img = t / (tiles_x * tiles_y);
  const int trem = t - img * (tiles_x * tiles_y);
  const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * kTW;
  const int n0 = (int)(v % (unsigned)ncb) * BN;

  const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, (unsigned)((int64_t)a.n * a.h * a.w * a.x_ld * 2));
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.wpack, (unsigned)((int64_t)9 * a.cout * a.cin * 2));

  // staging roles.  Patch: (TH + 2) x 18 pixels x four 16-byte pieces (TH = 16: 1296 pieces, six rounds of 256 threads, the last
  // one partial).  Round j of a thread handles piece tid + 256 j: its LDS slot is a constant away from round 0's, its global
  // offset is recomputed per chunk (a dozen integer instructions per nine steps; kept in registers the six offsets and
  // six slots cost the 128-wide variant the registers it needs to read fragments ahead of the matrix instructions)
  constexpr int ARND = (kPatchPix * 4 + 255) / 256;
  const unsigned al0 = (unsigned)((tid >> 2) * kRowB + (tid & 3) * 16);
  auto a_offset = [&](int j) -> unsigned {
    const int pix = (tid >> 2) + 64 * j;
    const int py = pix / kPW, px = pix - py * kPW;
    const int iy = y0 - 1 + py, ix = x0 - 1 + px;
    const bool ok = pix < kPatchPix && (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w;
    return ok ? (unsigned)(((img * a.h + iy) * a.w + ix) * a.x_ld) * 2u + (unsigned)(tid & 3) * 16u : kOOB;
  };
  constexpr int BRND = BN / 64;                      // kernel rows: BN x four pieces / 256 threads
  unsigned bg[BRND], bl[BRND];
#pragma unroll
  for (int j = 0; j < BRND; ++j) {
    const int idx = tid + 256 * j;
    const int row = idx >> 2, piece = idx & 3;
    bg[j] = (unsigned)((n0 + row) * a.cin) * 2u + (unsigned)piece * 16u;
    bl[j] = (unsigned)(row * kRowB + piece * 16);
  }
  const unsigned tap_stride = (unsigned)(a.cout * a.cin) * 2u;      // bytes between taps in the pack

  // fragment addresses
  unsigned abase[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int pm = wm * WPX + m * 32 + r;
    abase[m] = (unsigned)(((pm >> 4) * kPW + (pm & 15)) * kRowB + hh * 16);
  }
  unsigned bbase[NTL];
#pragma unroll
  for (int q = 0; q < NTL; ++q) bbase[q] = (unsigned)((wn * (BN / WN) + q * 32 + r) * kRowB + hh * 16);

  f32x16 acc[MT][NTL];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int q = 0; q < NTL; ++q)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][q][i] = 0.f;

  const int nchunks = (TILED_ABLATE & 1) ? 0 : a.cin >> 5;
  constexpr int AH = ADB ? (ARND + 1) / 2 : ARND;      // staging rounds held in registers at a time
  u32x4 areg[AH], breg[3][BRND];
  auto load_a = [&](int chunk, int j0) {              // rounds j0 .. j0 + AH - 1 of the patch of `chunk`
#pragma unroll
    for (int j = 0; j < AH; ++j) {
      if (j0 + j >= ARND) break;
      const unsigned o = a_offset(j0 + j);
      areg[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (o + (unsigned)chunk * 64u) | (o & kOOB), 0, 0));
    }
  };
  auto commit_a = [&](unsigned char* dst, int j0) {
#pragma unroll
    for (int j = 0; j < AH; ++j) {
      if (j0 + j >= ARND) break;
      if ((tid >> 2) + 64 * (j0 + j) < kPatchPix) *reinterpret_cast<u32x4*>(dst + al0 + (j0 + j) * (64 * kRowB)) = areg[j];
    }
  };
  // The kernel rows of a step come out of L2 with more latency than a step of matrix work takes: they are requested three
  // steps ahead into a ring of three register sets (slot = tap % 3; nine taps per chunk keep the slots aligned over chunks).
  auto load_b = [&](int slot, int tap, int chunk) {
    const unsigned oob = chunk < nchunks ? 0u : kOOB;      // past the last step: zeros nobody multiplies
#pragma unroll
    for (int j = 0; j < BRND; ++j)
      breg[slot][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, (bg[j] + (unsigned)tap * tap_stride + (unsigned)chunk * 64u) | oob, 0, 0));
  };
  auto commit_b = [&](int slot, int buf) {
#pragma unroll
    for (int j = 0; j < BRND; ++j) *reinterpret_cast<u32x4*>(Bs + buf * kBBytes + bl[j]) = breg[slot][j];
  };

  {      // first patch: every round at once (these registers are free again before the loop)
    u32x4 first[ARND];
#pragma unroll
    for (int j = 0; j < ARND; ++j) first[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, a_offset(j), 0, 0));
    load_b(0, 0, 0);
    load_b(1, 1, 0);
    load_b(2, 2, 0);
#pragma unroll
    for (int j = 0; j < ARND; ++j)
      if ((tid >> 2) + 64 * j < kPatchPix) *reinterpret_cast<u32x4*>(smem + al0 + j * (64 * kRowB)) = first[j];
  }
  commit_b(0, 0);
  __syncthreads();

  int buf = 0, abuf = 0;
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const bool more = chunk + 1 < nchunks;
    const unsigned char* const As = smem + abuf * kABytes;
    unsigned char* const Anext = smem + (kABufs - 1 - abuf) * kABytes;      // (the same image with a single buffer)
    if (!ADB && more) load_a(chunk + 1, 0);           // in flight under the nine taps of this chunk
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const bool last = tap == 8;
      load_b(tap % 3, (tap + 3) % 9, chunk + (tap + 3) / 9);      // this slot's rows went to LDS at the end of the previous step
      if constexpr (ADB) {                            // next patch: first half requested at tap 0, second at tap 3
        if (more && tap == 0) load_a(chunk + 1, 0);
        if (more && tap == 3) { commit_a(Anext, 0); load_a(chunk + 1, AH); }
        if (more && tap == 6) commit_a(Anext, AH);
      }
      const unsigned toff = (unsigned)(((tap / 3) * kPW + (tap % 3)) * kRowB);
      const unsigned char* bsrc = Bs + buf * kBBytes;
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        bf16x8 af[MT], bfr[NTL];
#pragma unroll
        for (int m = 0; m < MT; ++m) af[m] = lds_frag(As + abase[m] + toff + k2 * 32);
#pragma unroll
        for (int q = 0; q < NTL; ++q) bfr[q] = lds_frag(bsrc + bbase[q] + k2 * 32);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int q = 0; q < NTL; ++q) acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m], bfr[q], acc[m][q], 0, 0, 0);
      }
      commit_b((tap + 1) % 3, buf ^ 1);               // the next step's rows into the buffer nobody reads during this step
      if (!ADB && last && more) {
        __syncthreads();                              // every wave is done with this chunk's patch
        commit_a(smem, 0);
      }
      __syncthreads();
      buf ^= 1;
    }
    if constexpr (ADB) abuf ^= 1;
  }

  // ---- epilogue through LDS: 64 pixels per pass, float32 rows of BN values + 16 bytes of padding ----
  constexpr int SROW = BN * 4 + 16;
  constexpr int CW = 4;                             // channels per thread (8-byte accesses; eight would cost ~40 more registers)
  constexpr int G = BN / CW;                          // channel groups
  constexpr int PL = 256 / G;                         // pixel lanes
  constexpr int ITEMS = 64 / PL;
  constexpr bool kDgrad = EPI == 2 || EPI == 3, kAdd = EPI == 3, kInfer = EPI == 4;
  constexpr int NS = kDgrad ? 3 : 2;
  const int cg = tid % G, pl = tid / G;
  const int co = n0 + cg * CW;
  float bv[CW];
#pragma unroll
  for (int e = 0; e < CW; ++e) bv[e] = a.bias ? a.bias[co + e] : 0.f;
  float sums[NS][CW];
#pragma unroll
  for (int k = 0; k < NS; ++k)
#pragma unroll
    for (int e = 0; e < CW; ++e) sums[k][e] = 0.f;
  const bool has_bn = kDgrad && a.bn.x != nullptr;
  float mu[CW], is[CW], ga[CW], be[CW], alp[CW];
  if constexpr (kDgrad || kInfer) {
    if (has_bn || kInfer) {
#pragma unroll
      for (int e = 0; e < CW; ++e) {
        mu[e] = a.bn.mean[co + e]; is[e] = a.bn.invstd[co + e]; ga[e] = a.bn.gamma[co + e]; be[e] = a.bn.beta[co + e];
        alp[e] = a.bn.act == SSMI_ACT_PRELU ? a.bn.alpha[co + e] : a.bn.slope;
      }
    }
  }
  const __amdgpu_buffer_rsrc_t ry = make_rsrc(a.y, (unsigned)((int64_t)a.n * a.h * a.w * a.y_ld * 2));
  const bool has_add = kAdd && a.bn.add != nullptr;
  const __amdgpu_buffer_rsrc_t radd = make_rsrc(has_add ? a.bn.add : (const void*)a.y, has_add ? (unsigned)((int64_t)a.n * a.h * a.w * a.y_ld * 2) : 0u);
  const __amdgpu_buffer_rsrc_t rbx = make_rsrc(has_bn ? a.bn.x : (const void*)a.y, has_bn ? (unsigned)((int64_t)a.n * a.h * a.w * a.bn.ld * 2) : 0u);
  auto pixel_of = [&](int q, int j) {                 // global pixel index of item j of pass q, kOOB for a pixel past the border
    const int tp = 64 * q + pl + j * PL;
    const int yy = y0 + (tp >> 4), xx = x0 + (tp & 15);
    return (yy < a.h && xx < a.w) ? (unsigned)((img * a.h + yy) * a.w + xx) : kOOB;
  };
  auto at = [&](unsigned gp, unsigned ld) {           // byte offset of this thread's channels of pixel gp (out of range for kOOB)
    return gp == kOOB ? kOOB : (gp * ld + (unsigned)co) * 2u;
  };
  // what the data-gradient epilogue reads per element is requested one pass ahead (two register sets): the wait for it
  // would otherwise be paid four times per tile
  typename Row<CW>::raw xin[2][kDgrad ? ITEMS : 1], ain[2][kAdd ? ITEMS : 1];
  auto request = [&](int q) {
    if constexpr (kDgrad) {
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) {
        const unsigned gp = pixel_of(q, j);
        xin[q & 1][j] = Row<CW>::ld(rbx, at(gp, (unsigned)a.bn.ld));
        if constexpr (kAdd) ain[q & 1][j] = Row<CW>::ld(radd, at(gp, (unsigned)a.y_ld));
      }
    }
  };
  request(0);

#pragma unroll
  for (int q = 0; q < BM / 64; ++q) {
    if (q > 0) __syncthreads();                       // the previous pass has been read (pass 0: the loop ended on a barrier)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int pb = wm * WPX + m * 32;               // first tile pixel of this wave's m-th 32-pixel block
      if (pb / 64 == q) {                             // uniform per wave
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) {
          const int col = wn * (BN / WN) + nt * 32 + r;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int row = (pb & 63) + (i & 3) + 8 * (i >> 2) + 4 * hh;
            *reinterpret_cast<float*>(smem + row * SROW + col * 4) = acc[m][nt][i];
          }
        }
      }
    }
    __syncthreads();
    if (q + 1 < BM / 64) request(q + 1);
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
      const int lp = pl + j * PL;                     // pixel of this pass
      const unsigned gp = pixel_of(q, j);
      const float live = gp == kOOB ? 0.f : 1.f;      // pixels past the border add nothing to the fused sums and are not stored
      float v[CW];
#pragma unroll
      for (int e4 = 0; e4 < CW / 4; ++e4) {
        const f32x4 sv = *reinterpret_cast<const f32x4*>(smem + lp * SROW + cg * (CW * 4) + e4 * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e4 * 4 + e] = sv[e] + bv[e4 * 4 + e];
      }
      if constexpr (EPI == 1) {
#pragma unroll
        for (int e = 0; e < CW; ++e) {
          const float vv = v[e] * live;
          sums[0][e] += vv;
          sums[1][e] += vv * vv;
        }
      }
      if constexpr (kDgrad) {
        if constexpr (kAdd) {
#pragma unroll
          for (int e2 = 0; e2 < CW / 2; ++e2) {
            const unsigned wa = ain[q & 1][j][e2];
            v[2 * e2] += __builtin_bit_cast(float, wa << 16);
            v[2 * e2 + 1] += __builtin_bit_cast(float, wa & 0xFFFF0000u);
          }
        }
        if (has_bn) {
          float dr[CW];
#pragma unroll
          for (int e = 0; e < CW; ++e) dr[e] = 1.f;
          if (a.bn.drop) {
            const float* dp = a.bn.drop + (size_t)((gp == kOOB ? 0u : gp) / a.bn.hw) * (unsigned)a.cout + (unsigned)co;
#pragma unroll
            for (int e = 0; e < CW; ++e) dr[e] = dp[e];
          }
#pragma unroll
          for (int e = 0; e < CW; ++e) {
            const unsigned wx = xin[q & 1][j][e >> 1];
            const float xr = __builtin_bit_cast(float, (e & 1) ? (wx & 0xFFFF0000u) : (wx << 16));
            const float d = v[e] * dr[e] * live;
            const float xh = (xr - mu[e]) * is[e];
            const float z = ga[e] * xh + be[e];
            const float gr = a.bn.act == SSMI_ACT_NONE ? 1.f : (z > 0.f ? 1.f : (z < 0.f ? alp[e] : 0.f));
            const float dz = d * gr;
            sums[0][e] += dz;
            sums[1][e] += dz * xh;
            sums[2][e] += d * (z < 0.f ? z : 0.f);
          }
        }
      }
      if constexpr (kInfer) {      // BatchNormalization with fixed statistics + activation (conv_fwd_kernel's inference epilogue)
#pragma unroll
        for (int e = 0; e < CW; ++e) {
          const float z = ga[e] * ((v[e] - mu[e]) * is[e]) + be[e];
          v[e] = (a.bn.act == SSMI_ACT_NONE || z > 0.f) ? z : alp[e] * z;
        }
      }
      typename Row<CW>::raw o;
#pragma unroll
      for (int e2 = 0; e2 < CW / 2; ++e2) o[e2] = pack_bf16x2(v[2 * e2], v[2 * e2 + 1]);
      if (!(TILED_ABLATE & 2) || o[0] == 0x12345678u) Row<CW>::st(ry, at(gp, (unsigned)a.y_ld), o);
    }
  }
  float* dst = EPI == 1 ? a.part : (has_bn ? a.bn.part : nullptr);
  if (EPI != 0 && EPI != 4 && dst != nullptr) {       // uniform
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);      // [NS][PL][BN]
#pragma unroll
    for (int k = 0; k < NS; ++k)
#pragma unroll
      for (int e = 0; e < CW; ++e) red[(k * PL + pl) * BN + cg * CW + e] = sums[k][e];
    __syncthreads();
    if (tid < BN) {
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        float s = 0.f;
        for (int p2 = 0; p2 < PL; ++p2) s += red[(k * PL + p2) * BN + tid];
        dst[((int64_t)t * NS + k) * a.cout + n0 + tid] = s;
      }
    }
  }
}

bool conv_tiled_bf16_ok(int h, int w, int cin, int cout, int kh, int kw, int stride, int dil, int pad_t, int pad_l, int ho, int wo,
                        int x_ld, int y_ld) {
  static const bool off = getenv("SSMI_NO_TILED_BF16") != nullptr;      // measurement switch
  return !off && kh == 3 && kw == 3 && stride == 1 && dil == 1 && pad_t == 1 && pad_l == 1 && ho == h && wo == w && h >= 8 && w >= 8 &&
         cin % 32 == 0 && cout % 64 == 0 && x_ld % 8 == 0 && y_ld % 8 == 0;      // any map size: border tiles are masked
}

// Launches the tiled kernel.  part != nullptr: statistics epilogue; bn.add / bn.x: data-gradient epilogue.  *part_rows
// receives the number of partial-sum rows (pixel tiles).  Returns SSMI_E_UNSUPPORTED for argument combinations it does not
// take (the caller falls back to the direct kernel).
ssmi_status launch_conv_tiled_bf16(int n, int h, int w, int cin, int cout, int x_ld, int y_ld, const void* x, const void* wpack,
                                   const float* bias, void* y, float* part, int* part_rows, const BnBwdArgs& bn, hipStream_t st) {
  const bool infer = bn.mean != nullptr && bn.x == nullptr;                    // BatchNormalization + activation in the epilogue
  if (infer && (bn.skip != nullptr || part != nullptr || bn.add != nullptr)) return SSMI_E_UNSUPPORTED;
  if (bn.x != nullptr && (bn.ld % 8 != 0 || (reinterpret_cast<uintptr_t>(bn.x) & 15))) return SSMI_E_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(wpack) |
       reinterpret_cast<uintptr_t>(bn.add)) & 15)
    return SSMI_E_UNSUPPORTED;
  if (part != nullptr && (bn.x != nullptr || bn.add != nullptr)) return SSMI_E_UNSUPPORTED;
  TiledArgs a{};
  a.n = n; a.h = h; a.w = w; a.cin = cin; a.cout = cout; a.x_ld = x_ld; a.y_ld = y_ld;
  a.x = static_cast<const bf16*>(x); a.wpack = static_cast<const bf16*>(wpack); a.bias = bias; a.y = static_cast<bf16*>(y);
  a.part = part; a.bn = bn;
  const int epi = infer ? 4 : (part ? 1 : (bn.add ? 3 : (bn.x ? 2 : 0)));
  const bool wide = cout % 128 == 0;
  const int bn_cols = wide ? 128 : 64;
  // 16-row tiles share more of the halo and of the kernel rows; layers too small to give every CU two workgroups that way
  // take 8-row tiles
  static const int below = getenv("SSMI_TILED_TH8_BELOW") ? atoi(getenv("SSMI_TILED_TH8_BELOW")) : 512;      // measurement switch
  // (the addend epilogue needs a second prefetch set and exceeds the 16-row variant's register budget: 8-row tiles always)
  const int tx_n = (w + kTW - 1) / kTW;
  const int th = (epi == 3 || h % 16 == 8 || n * ((h + 15) / 16) * tx_n * (cout / bn_cols) < below) ? 8 : 16;
  const int tiles = n * ((h + th - 1) / th) * tx_n;
  // one row of partial sums per tile: the callers' workspaces hold ceil(pixels / 128) + 1024 rows (narrow maps with many
  // partial tiles could exceed that: those go to the direct kernel)
  if ((part != nullptr || bn.x != nullptr) && tiles > ((int64_t)n * h * w + 127) / 128 + 1024) return SSMI_E_UNSUPPORTED;
  if (part_rows) *part_rows = tiles;
  dim3 grid((unsigned)(tiles * (cout / bn_cols)), 1, 1);
  const size_t lds = (size_t)(wide ? 2 : 1) * (th + 2) * kPW * kRowB + 2 * (size_t)bn_cols * kRowB;      // two patch images for the 128-wide variants
#define LT(BN_, WM_, WN_, E_)                                                                                                  \
  do {                                                                                                                          \
    if (th == 16) hipLaunchKernelGGL((conv3x3_tiled_bf16<BN_, WM_, WN_, E_, (E_ == 3 ? 8 : 16)>), grid, dim3(256), lds, st, a); \
    else hipLaunchKernelGGL((conv3x3_tiled_bf16<BN_, WM_, WN_, E_, 8>), grid, dim3(256), lds, st, a);                           \
  } while (0)
  set_last_kernel(wide ? "conv3x3_tiled_bf16<128, 2, 2, %d, %d>" : "conv3x3_tiled_bf16<64, 4, 1, %d, %d>", epi, th);      // as the profiler prints it
  if (wide) { if (epi == 0) LT(128, 2, 2, 0); else if (epi == 1) LT(128, 2, 2, 1); else if (epi == 2) LT(128, 2, 2, 2); else if (epi == 3) LT(128, 2, 2, 3); else LT(128, 2, 2, 4); }
  else { if (epi == 0) LT(64, 4, 1, 0); else if (epi == 1) LT(64, 4, 1, 1); else if (epi == 2) LT(64, 4, 1, 2); else if (epi == 3) LT(64, 4, 1, 3); else LT(64, 4, 1, 4); }
#undef LT
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of the same layers (3x3, stride 1, SAME, bfloat16 storage; input and output channels in whole 64-blocks;
// any map size: dy rows past the border read zeros):
//   dw[tap][ci][co] = sum_pixels x[pixel + tap][ci] * dy[pixel][co]        db[co] = sum_pixels dy[pixel][co]
// A workgroup owns a 64 x 64 block of (ci, co) for all nine taps and a range of 8 x 16 pixel tiles.  Per tile it stages the
// 10 x 18 halo patch of x (64 channels) and the dy tile (64 channels) in LDS ONCE - the strip kernels gather x once per tap -
// and walks the ten patch rows: the x operand of patch row p serves tap row ky for the dy row p - ky, so one patch row
// (three operand reads, one per kx) and one new dy row feed nine matrix instructions; the two previous dy rows stay in
// registers.  Operands come from the transposing LDS read (ds_read_tr16_b64: pixels become the reduction dimension), rows
// are 192 bytes (128 of channels + 64 of padding: the four pixel rows and two channel halves of one read fall into distinct
// banks).  Four waves split the block 2 x 2 (32 x 32 x 9 taps = 144 accumulator registers each).  The next tile's rows
// travel in registers while the current one is multiplied (two workgroups fit a CU; the launch is sized to about one per CU,
// which measured best beside the main chain of a training step).
// Output: partial sums part[share][K_aug][cout] (k = tap * cin + ci, row K = bias gradient), summed by the caller's
// reduction kernel.
// ---------------------------------------------------------------------------------------------------------------------
static constexpr int kWgTH = 8, kWgPW = 18, kWgPH = kWgTH + 2;
static constexpr int kWgRowB = 192;
static constexpr int kWgABytes = kWgPH * kWgPW * kWgRowB;      // 34560
static constexpr int kWgBBytes = kWgTH * 16 * kWgRowB;         // 24576

struct WgPatchArgs {
  int n, h, w, cin, cout, x_ld, y_ld, K, K_aug, ntiles, tiles_per_share, shares;
  const bf16* x;
  const bf16* dy;
  float* part;
};

__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_patch_bf16(WgPatchArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) char* lds_ptr;
  const lds_ptr lA = (lds_ptr)smem;
  const lds_ptr lB = lA + kWgABytes;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int sq = (lane & 15) >> 2, sp = lane & 3, gh = (lane >> 4) & 1;      // this lane's part in a transposing read
  const int wci = wave & 1, wco = wave >> 1;

  // (ci block, co block) fastest, share slowest, XCD-contiguous: the workgroups of an XCD that run together work on the
  // same pixel tiles, so x and dy are fetched into that L2 once and read from it by every block
  const int nbi = a.cin >> 6, nbo = a.cout >> 6;
  const unsigned v = spread_over_xcds(blockIdx.x, gridDim.x);
  const int blk = (int)(v % (unsigned)(nbi * nbo)), share = (int)(v / (unsigned)(nbi * nbo));
  const int bi = blk % nbi, bo = blk / nbi;
  const int t_begin = share * a.tiles_per_share, t_end = min(a.ntiles, t_begin + a.tiles_per_share);
  const int tiles_x = (a.w + 15) >> 4, tiles_y = (a.h + kWgTH - 1) / kWgTH;      // border tiles: dy rows past the border read zeros

  const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, (unsigned)((int64_t)a.n * a.h * a.w * a.x_ld * 2));
  const __amdgpu_buffer_rsrc_t rdy = make_rsrc(a.dy, (unsigned)((int64_t)a.n * a.h * a.w * a.y_ld * 2));

  // staging roles: patch 180 pixels x eight 16-byte pieces = 1440 pieces (six rounds, the last partial); dy 128 x 8 = 1024
  constexpr int ARND = (kWgPH * kWgPW * 8 + 255) / 256, BRND = 4;
  int apy[ARND], apx[ARND];
  unsigned al[ARND], apiece[ARND];
#pragma unroll
  for (int j = 0; j < ARND; ++j) {
    const int idx = tid + 256 * j, pix = idx >> 3;
    apy[j] = pix / kWgPW;
    apx[j] = pix - apy[j] * kWgPW;
    apiece[j] = (unsigned)((bi * 64 + (idx & 7) * 8) * 2);
    al[j] = pix < kWgPH * kWgPW ? (unsigned)(pix * kWgRowB + (idx & 7) * 16) : 0xFFFFFFFFu;
  }
  unsigned brel[BRND], bl[BRND];
  int bpy[BRND], bpx[BRND];
#pragma unroll
  for (int j = 0; j < BRND; ++j) {
    const int idx = tid + 256 * j, pix = idx >> 3;      // tile pixel: row pix >> 4, column pix & 15
    bpy[j] = pix >> 4;
    bpx[j] = pix & 15;
    brel[j] = (unsigned)((bpy[j] * a.w + bpx[j]) * a.y_ld + bo * 64 + (idx & 7) * 8) * 2u;
    bl[j] = (unsigned)(pix * kWgRowB + (idx & 7) * 16);
  }
  u32x4 areg[ARND], breg[BRND];
  auto request = [&](int t) {
    const bool tv = t < t_end;
    const int img = t / (tiles_x * tiles_y);
    const int trem = t - img * (tiles_x * tiles_y);
    const int y0 = (trem / tiles_x) * kWgTH, x0 = (trem % tiles_x) * 16;
#pragma unroll
    for (int j = 0; j < ARND; ++j) {
      const int iy = y0 - 1 + apy[j], ix = x0 - 1 + apx[j];
      const bool ok = tv && al[j] != 0xFFFFFFFFu && (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w;
      const unsigned off = (unsigned)(((img * a.h + iy) * a.w + ix) * a.x_ld) * 2u + apiece[j];
      areg[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? off : kOOB, 0, 0));
    }
    const unsigned base = (unsigned)(((img * a.h + y0) * a.w + x0) * a.y_ld) * 2u;
#pragma unroll
    for (int j = 0; j < BRND; ++j)
      breg[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
          rdy, (tv && y0 + bpy[j] < a.h && x0 + bpx[j] < a.w) ? base + brel[j] : kOOB, 0, 0));
  };

  f32x16 acc[3][3];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[ky][kx][i] = 0.f;
  float bsum = 0.f;
  const bool do_bias = a.K_aug > a.K && bi == 0 && wci == 0;      // uniform per wave

  struct Pair { s16x4 lo, hi; };
  auto operand = [&](lds_ptr p) {
    Pair pr;
    pr.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    pr.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * kWgRowB));
    return __builtin_bit_cast(bf16x8, pr);
  };
  const int a_lane = (8 * hh + sq) * kWgRowB + (wci * 32 + 16 * gh + 4 * sp) * 2;
  const int b_lane = (8 * hh + sq) * kWgRowB + (wco * 32 + 16 * gh + 4 * sp) * 2;

  request(t_begin);
  for (int t = t_begin; t < t_end; ++t) {
    __syncthreads();                                  // everyone is done with the previous tile
#pragma unroll
    for (int j = 0; j < ARND; ++j)
      if (al[j] != 0xFFFFFFFFu) *reinterpret_cast<u32x4*>(smem + al[j]) = areg[j];
#pragma unroll
    for (int j = 0; j < BRND; ++j) *reinterpret_cast<u32x4*>(smem + kWgABytes + bl[j]) = breg[j];
    __syncthreads();
    request(t + 1);                                   // in flight while this tile is multiplied
    bf16x8 b0, b1, b2;                                // dy rows p, p - 1, p - 2
#pragma unroll
    for (int p = 0; p < kWgPH; ++p) {
      bf16x8 ax[3];
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) ax[kx] = operand(lA + a_lane + (p * kWgPW + kx) * kWgRowB);
      b2 = b1;
      b1 = b0;
      if (p < kWgTH) {
        b0 = operand(lB + b_lane + p * 16 * kWgRowB);
        if (do_bias) {
#pragma unroll
          for (int e = 0; e < 8; ++e) bsum += (float)b0[e];
        }
      }
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        if (p < kWgTH) acc[0][kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ax[kx], b0, acc[0][kx], 0, 0, 0);
        if (p >= 1 && p - 1 < kWgTH) acc[1][kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ax[kx], b1, acc[1][kx], 0, 0, 0);
        if (p >= 2 && p - 2 < kWgTH) acc[2][kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ax[kx], b2, acc[2][kx], 0, 0, 0);
      }
    }
  }

  float* pbase = a.part + (int64_t)share * a.K_aug * a.cout;
  const int co = bo * 64 + wco * 32 + r;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ci = bi * 64 + wci * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
        pbase[(int64_t)((ky * 3 + kx) * a.cin + ci) * a.cout + co] = acc[ky][kx][i];
      }
  if (do_bias) {
    bsum += __shfl_xor(bsum, 32, 64);
    if (hh == 0) pbase[(int64_t)a.K * a.cout + co] = bsum;
  }
}

WgPatchPlan plan_wg_patch_bf16(int n, int h, int w, int cin, int cout, int kh, int kw, int stride, int dil, int pad_t, int pad_l,
                               int ho, int wo, int x_ld, int y_ld, bool with_bias) {
  static const bool off = getenv("SSMI_NO_WG_PATCH") != nullptr;      // measurement switch
  WgPatchPlan p{};
  p.use = !off && kh == 3 && kw == 3 && stride == 1 && dil == 1 && pad_t == 1 && pad_l == 1 && ho == h && wo == w && h >= 8 && w >= 8 &&
          cin % 64 == 0 && cout % 64 == 0 && x_ld % 8 == 0 && y_ld % 8 == 0;      // any map size: border tiles are masked
  if (!p.use) return p;
  p.K = 9 * cin;
  p.K_aug = p.K + (with_bias ? 1 : 0);
  p.ntiles = n * ((h + kWgTH - 1) / kWgTH) * ((w + 15) / 16);
  const int nblocks = (cin / 64) * (cout / 64);
  static const int target = getenv("SSMI_WG_PATCH_WGS") ? atoi(getenv("SSMI_WG_PATCH_WGS")) : 256;      // measurement switch
  // one workgroup per CU: inside the step (beside the main chain) 256 measured best - 384 / 512 / 768 / 1024: +0.02 / +0.15 /
  // +0.33 / +0.55 ms on SegNet, 128: +0.3 (every share writes a full partial dW, and the main chain needs room on the CUs)
  int shares = target / nblocks;
  if (shares > p.ntiles / 2) shares = p.ntiles / 2;                // at least two tiles per workgroup
  if (shares < 1) shares = 1;
  p.tiles_per_share = (p.ntiles + shares - 1) / shares;
  p.shares = (p.ntiles + p.tiles_per_share - 1) / p.tiles_per_share;
  p.bytes = (size_t)p.shares * p.K_aug * cout * sizeof(float);
  return p;
}

ssmi_status launch_wg_patch_bf16(const WgPatchPlan& p, int n, int h, int w, int cin, int cout, int x_ld, int y_ld, const void* x,
                                 const void* dy, float* part, hipStream_t st) {
  WgPatchArgs a{};
  a.n = n; a.h = h; a.w = w; a.cin = cin; a.cout = cout; a.x_ld = x_ld; a.y_ld = y_ld; a.K = p.K; a.K_aug = p.K_aug;
  a.ntiles = p.ntiles; a.tiles_per_share = p.tiles_per_share; a.shares = p.shares;
  a.x = static_cast<const bf16*>(x); a.dy = static_cast<const bf16*>(dy); a.part = part;
  const unsigned grid = (unsigned)(p.shares * (cin / 64) * (cout / 64));
  hipLaunchKernelGGL(conv3x3_wgrad_patch_bf16, dim3(grid), dim3(256), (size_t)(kWgABytes + kWgBBytes), st, a);
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// bfloat16 kernel packs of every table entry, once per step after the optimizer: fpack[tap][co][ci] (transposed) and
// rpack[ntaps - 1 - tap][ci][co] = bf16(w[tap][ci][co]).  A workgroup takes 64 x 64 (ci, co) tiles of a layer: float32 rows
// in with 16-byte loads, the straight pack out with 8-byte stores, the transposed pack through LDS so that its stores are
// 8 bytes of consecutive ci as well.  Layers whose channel counts are not multiples of four go element by element.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_pack_bf16_tiled_kernel(const ssmi_rot_entry* __restrict__ table, const float* __restrict__ theta,
                                                                   bf16* __restrict__ fpack, bf16* __restrict__ rpack) {
  __shared__ unsigned short sm[64][68];
  const ssmi_rot_entry en = table[blockIdx.y];
  const float* w = theta + en.offset;
  const int cin = en.cin, cout = en.cout, per_tap = cin * cout, ntaps = en.kh * en.kw;
  bf16* const fp = fpack ? fpack + 2 * en.offset : nullptr;      // byte offset 4 * offset: 16-byte aligned like theta's segments
  bf16* const rp = rpack ? rpack + 2 * en.offset : nullptr;
  if ((cin | cout) & 3) {
    const int total = ntaps * per_tap;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
      const int tap = e / per_tap, rem = e - tap * per_tap;
      const int ci = rem / cout, co = rem - ci * cout;
      const bf16 v = (bf16)w[e];
      if (fp) fp[(int64_t)(tap * cout + co) * cin + ci] = v;
      if (rp) rp[(int64_t)(ntaps - 1 - tap) * per_tap + rem] = v;
    }
    return;
  }
  const int tci_n = (cin + 63) >> 6, tco_n = (cout + 63) >> 6;
  const int ntiles = ntaps * tci_n * tco_n;
  const int row0 = threadIdx.x >> 4, c4 = (threadIdx.x & 15) * 4;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int tap = tile / (tci_n * tco_n), rem = tile - tap * (tci_n * tco_n);
    const int ci0 = (rem / tco_n) * 64, co0 = (rem % tco_n) * 64;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = row0 + 16 * j, ci = ci0 + row, co = co0 + c4;
      u32x2 pk = {0u, 0u};
      if (ci < cin && co < cout) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(w + (int64_t)(tap * cin + ci) * cout + co);
        pk = pack_bf16x4(v);
        if (rp) *reinterpret_cast<u32x2*>(rp + (int64_t)(ntaps - 1 - tap) * per_tap + (int64_t)ci * cout + co) = pk;
      }
      *reinterpret_cast<u32x2*>(&sm[row][c4]) = pk;
    }
    __syncthreads();
    if (fp) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int cor = row0 + 16 * j, co = co0 + cor, ci = ci0 + c4;      // this thread's four consecutive ci of output channel co
        if (co < cout && ci < cin) {
          u32x2 pk;
          pk[0] = (unsigned)sm[c4][cor] | ((unsigned)sm[c4 + 1][cor] << 16);
          pk[1] = (unsigned)sm[c4 + 2][cor] | ((unsigned)sm[c4 + 3][cor] << 16);
          *reinterpret_cast<u32x2*>(fp + (int64_t)(tap * cout + co) * cin + ci) = pk;
        }
      }
    }
    __syncthreads();
  }
}

ssmi_status launch_conv_pack_bf16(const ssmi_rot_entry* table, int count, const float* theta, void* fpack, void* rpack, hipStream_t st) {
  hipLaunchKernelGGL(conv_pack_bf16_tiled_kernel, dim3(192, (unsigned)count), dim3(256), 0, st, table, theta, static_cast<bf16*>(fpack),
                     static_cast<bf16*>(rpack));
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

}  // namespace ssmi

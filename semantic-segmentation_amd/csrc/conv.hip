// Convolutions as implicit GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), operands read straight
// from NHWC global memory.  One wave owns a 32-row x (NT*32)-column accumulator tile:
//   forward / data-gradient : rows = 32 consecutive pixels, columns = output channels, K = taps x channels
//   weight-gradient         : rows = 32 consecutive (tap,cin) kernel rows, columns = cout, K = pixels
// MFMA operand map (32x32x2, wave64): lane l supplies A[row = l&31][k = l>>5] and B[k = l>>5][col = l&31];
// accumulator register i of lane l is C[row = (i&3) + 8*(i>>2) + 4*(l>>5)][col = l&31].
#include "common.h"

namespace ssmi {

struct Geo {
  int n, h, w, cin, ho, wo, cout, kh, kw, stride, dil, pad_t, pad_l, x_ld, y_ld;
};

static Geo to_geo(const ssmi_conv_desc* d) {
  Geo g;
  g.n = d->n; g.h = d->h; g.w = d->w; g.cin = d->cin; g.ho = d->ho; g.wo = d->wo; g.cout = d->cout;
  g.kh = d->kh; g.kw = d->kw; g.stride = d->stride; g.dil = d->dil; g.pad_t = d->pad_t; g.pad_l = d->pad_l;
  g.x_ld = d->x_ld; g.y_ld = d->y_ld;
  return g;
}

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// ---------------------------------------------------------------------------------------------------
// forward: y[p, co] = bias[co] + sum_{tap, ci} x[pix(p, tap), ci] * w[tap, ci, co]
// VEC: cin % 8 == 0 and 16-byte aligned rows -> each lane loads 4 consecutive channels at once; the k order inside
// a block of 8 channels is permuted identically for A and B (lane half hh takes channels 4*hh .. 4*hh+3).
// ---------------------------------------------------------------------------------------------------
template <int NT, bool VEC>
__global__ __launch_bounds__(256) void conv_fwd_kernel(Geo g, const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ y) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int64_t M = (int64_t)g.n * g.ho * g.wo;
  const int64_t tile0 = ((int64_t)blockIdx.x * 4 + wave) * 32;
  if (tile0 >= M) return;
  const int cbase = blockIdx.y * 32 * NT;
  const int64_t p = tile0 + r;
  const bool pv = p < M;
  const int64_t pp = pv ? p : 0;
  const int ox = (int)(pp % g.wo);
  const int64_t t1 = pp / g.wo;
  const int oy = (int)(t1 % g.ho);
  const int n_ = (int)(t1 / g.ho);

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  bool colv[NT];
  int col[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    col[t] = cbase + t * 32 + r;
    colv[t] = col[t] < g.cout;
    if (!colv[t]) col[t] = 0;
  }

  for (int ky = 0; ky < g.kh; ++ky) {
    const int iy = oy * g.stride + ky * g.dil - g.pad_t;
    for (int kx = 0; kx < g.kw; ++kx) {
      const int ix = ox * g.stride + kx * g.dil - g.pad_l;
      const bool inb = pv && (unsigned)iy < (unsigned)g.h && (unsigned)ix < (unsigned)g.w;
      const float* xp = x + (((int64_t)n_ * g.h + (inb ? iy : 0)) * g.w + (inb ? ix : 0)) * g.x_ld;
      const float* wp = w + (int64_t)(ky * g.kw + kx) * g.cin * g.cout;
      if (VEC) {
#pragma unroll 2
        for (int c0 = 0; c0 < g.cin; c0 += 8) {
          const int ci = c0 + 4 * hh;
          f32x4 a = {0.f, 0.f, 0.f, 0.f};
          if (inb) a = *reinterpret_cast<const f32x4*>(xp + ci);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              float b = wp[(int64_t)(ci + j) * g.cout + col[t]];
              b = colv[t] ? b : 0.f;
              acc[t] = MFMA32(a[j], b, acc[t]);
            }
          }
        }
      } else {
#pragma unroll 4
        for (int c0 = 0; c0 < g.cin; c0 += 2) {
          const int ci = c0 + hh;
          const bool cv = ci < g.cin;
          const int cis = cv ? ci : 0;
          float a = xp[cis];
          a = (inb && cv) ? a : 0.f;
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            float b = wp[(int64_t)cis * g.cout + col[t]];
            b = (cv && colv[t]) ? b : 0.f;
            acc[t] = MFMA32(a, b, acc[t]);
          }
        }
      }
    }
  }

#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (!colv[t]) continue;
    const float bv = bias ? bias[col[t]] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
      const int64_t po = tile0 + row;
      if (po < M) y[po * g.y_ld + col[t]] = acc[t][i] + bv;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// data gradient / transposed-convolution forward (gather form, one launch per output phase):
//   dx[n, iy, ix, ci] = bias[ci] + add[..] + sum_{ky,kx : (iy + pad_t - ky*dil) % stride == 0, ...} sum_co
//                        dy[n, (iy+pad_t-ky*dil)/stride, (ix+pad_l-kx*dil)/stride, co] * w[ky,kx,ci,co]
// blockIdx.z enumerates the stride x stride phases (iy % stride, ix % stride) so the set of contributing taps is
// uniform inside a block.
// ---------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void conv_bwd_data_kernel(Geo g, const float* __restrict__ dy,
                                                            const float* __restrict__ w,
                                                            const float* __restrict__ add,
                                                            const float* __restrict__ bias, float* __restrict__ dx) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int s = g.stride;
  const int pa = blockIdx.z / s, pb = blockIdx.z % s;
  const int qh = (g.h - pa + s - 1) / s, qw = (g.w - pb + s - 1) / s;
  if (qh <= 0 || qw <= 0) return;
  const int64_t M = (int64_t)g.n * qh * qw;
  const int64_t tile0 = ((int64_t)blockIdx.x * 4 + wave) * 32;
  if (tile0 >= M) return;
  const int cbase = blockIdx.y * 32 * NT;

  const int64_t p = tile0 + r;
  const bool pv = p < M;
  const int64_t pp = pv ? p : 0;
  const int qx = (int)(pp % qw);
  const int64_t t1 = pp / qw;
  const int qy = (int)(t1 % qh);
  const int n_ = (int)(t1 / qh);
  const int iy = qy * s + pa, ix = qx * s + pb;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  bool colv[NT];
  int col[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    col[t] = cbase + t * 32 + r;  // input channel ci of the forward conv
    colv[t] = col[t] < g.cin;
    if (!colv[t]) col[t] = 0;
  }

  for (int ky = 0; ky < g.kh; ++ky) {
    const int ty0 = pa + g.pad_t - ky * g.dil;
    if (((ty0 % s) + s) % s != 0) continue;  // uniform
    const int ty = iy + g.pad_t - ky * g.dil;
    const int oy = ty / s;  // exact when ty >= 0
    for (int kx = 0; kx < g.kw; ++kx) {
      const int tx0 = pb + g.pad_l - kx * g.dil;
      if (((tx0 % s) + s) % s != 0) continue;  // uniform
      const int tx = ix + g.pad_l - kx * g.dil;
      const int ox = tx / s;
      const bool inb = pv && ty >= 0 && tx >= 0 && oy < g.ho && ox < g.wo;
      const float* yp = dy + (((int64_t)n_ * g.ho + (inb ? oy : 0)) * g.wo + (inb ? ox : 0)) * g.y_ld;
      const float* wp = w + (int64_t)(ky * g.kw + kx) * g.cin * g.cout;
#pragma unroll 4
      for (int c0 = 0; c0 < g.cout; c0 += 2) {
        const int co = c0 + hh;
        const bool cv = co < g.cout;
        const int cos = cv ? co : 0;
        float a = yp[cos];
        a = (inb && cv) ? a : 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          float b = wp[(int64_t)col[t] * g.cout + cos];
          b = (cv && colv[t]) ? b : 0.f;
          acc[t] = MFMA32(a, b, acc[t]);
        }
      }
    }
  }

#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (!colv[t]) continue;
    const float bv = bias ? bias[col[t]] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
      const int64_t po = tile0 + row;
      if (po < M) {
        const int ox_ = (int)(po % qw);
        const int64_t t2 = po / qw;
        const int oy_ = (int)(t2 % qh);
        const int nn = (int)(t2 / qh);
        const int64_t off = (((int64_t)nn * g.h + (oy_ * s + pa)) * g.w + (ox_ * s + pb)) * g.x_ld + col[t];
        float v = acc[t][i] + bv;
        if (add) v += add[off];
        dx[off] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// weight gradient: dw[(tap,ci), co] = sum_p x[pix(p,tap), ci] * dy[p, co]; optional extra row K = sum_p dy[p, co].
// Each wave owns kernel rows [32*ktile, 32*ktile+32) x NT column tiles and a contiguous chunk of output rows,
// and writes its partial sums to part[chunk][K_aug][cout]; conv_bwd_weight_reduce sums the chunks in a fixed
// order (deterministic).
// ---------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void conv_bwd_weight_kernel(Geo g, const float* __restrict__ x,
                                                              const float* __restrict__ dy, float* __restrict__ part,
                                                              int K, int K_aug, int ktiles, int rows_per_chunk,
                                                              int nchunks) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int64_t unit = (int64_t)blockIdx.x * 4 + wave;  // (chunk, ktile)
  if (unit >= (int64_t)nchunks * ktiles) return;
  const int ktile = (int)(unit % ktiles);
  const int chunk = (int)(unit / ktiles);
  const int cbase = blockIdx.y * 32 * NT;

  const int k = ktile * 32 + r;
  const bool kv = k < K;
  const bool kbias = (k == K) && (K_aug > K);
  const int ks = kv ? k : 0;
  const int tap = ks / g.cin, ci = ks % g.cin;
  const int ky = tap / g.kw, kx = tap % g.kw;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  bool colv[NT];
  int col[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    col[t] = cbase + t * 32 + r;
    colv[t] = col[t] < g.cout;
    if (!colv[t]) col[t] = 0;
  }

  const int R = g.n * g.ho;
  const int row0 = chunk * rows_per_chunk;
  const int row1 = min(R, row0 + rows_per_chunk);
  for (int row = row0; row < row1; ++row) {
    const int n_ = row / g.ho, oy = row % g.ho;
    const int iy = oy * g.stride + ky * g.dil - g.pad_t;
    const bool yv = kv && (unsigned)iy < (unsigned)g.h;
    const float* xrow = x + ((int64_t)n_ * g.h + (yv ? iy : 0)) * g.w * g.x_ld + ci;
    const float* dyrow = dy + ((int64_t)n_ * g.ho + oy) * g.wo * g.y_ld;
#pragma unroll 4
    for (int ox0 = 0; ox0 < g.wo; ox0 += 2) {
      const int ox = ox0 + hh;
      const bool ov = ox < g.wo;
      const int oxs = ov ? ox : 0;
      const int ix = oxs * g.stride + kx * g.dil - g.pad_l;
      const bool xv = yv && ov && (unsigned)ix < (unsigned)g.w;
      float a = xrow[(int64_t)(xv ? ix : 0) * g.x_ld];
      a = xv ? a : 0.f;
      if (kbias) a = ov ? 1.f : 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float b = dyrow[(int64_t)oxs * g.y_ld + col[t]];
        b = (ov && colv[t]) ? b : 0.f;
        acc[t] = MFMA32(a, b, acc[t]);
      }
    }
  }

  float* pbase = part + (int64_t)chunk * K_aug * g.cout;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (!colv[t]) continue;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int rr = (i & 3) + 8 * (i >> 2) + 4 * hh;
      const int kk = ktile * 32 + rr;
      if (kk < K_aug) pbase[(int64_t)kk * g.cout + col[t]] = acc[t][i];
    }
  }
}

__global__ __launch_bounds__(256) void conv_bwd_weight_reduce(const float* __restrict__ part, int nchunks, int K,
                                                              int K_aug, int cout, float* __restrict__ dw,
                                                              float* __restrict__ dbias) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int E = K_aug * cout;
  if (e >= E) return;
  float s = 0.f;
  for (int c = 0; c < nchunks; ++c) s += part[(int64_t)c * E + e];
  if (e < K * cout) dw[e] = s;
  else if (dbias) dbias[e - K * cout] = s;
}

// column sums of a [npix, c] matrix with row stride ld (bias gradient of a transposed convolution)
__global__ __launch_bounds__(256) void colsum_partial(const float* __restrict__ x, int64_t npix, int c, int ld,
                                                      float* __restrict__ part) {
  // thread t handles column (t % c) for rows t / c + k * rows_per_iter
  const int ppi = 256 / c;  // c <= 256 checked on the host
  const int cc = threadIdx.x % c, pl = threadIdx.x / c;
  __shared__ float sm[256];
  float s = 0.f;
  if (pl < ppi) {
    for (int64_t p = (int64_t)blockIdx.x * ppi + pl; p < npix; p += (int64_t)gridDim.x * ppi) s += x[p * ld + cc];
  }
  sm[threadIdx.x] = (pl < ppi) ? s : 0.f;
  __syncthreads();
  if (threadIdx.x < c) {
    float t = 0.f;
    for (int q = 0; q < ppi; ++q) t += sm[q * c + threadIdx.x];
    part[(int64_t)blockIdx.x * c + threadIdx.x] = t;
  }
}

__global__ void colsum_final(const float* __restrict__ part, int nblocks, int c, float* __restrict__ out) {
  const int cc = blockIdx.x * blockDim.x + threadIdx.x;
  if (cc >= c) return;
  double s = 0.0;
  for (int b = 0; b < nblocks; ++b) s += (double)part[(int64_t)b * c + cc];
  out[cc] = (float)s;
}

static ssmi_status check_desc(const ssmi_conv_desc* d) {
  SSMI_REQUIRE(d != nullptr, SSMI_E_INVALID_ARG, "conv desc is NULL");
  SSMI_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0 && d->cin > 0 && d->ho > 0 && d->wo > 0 && d->cout > 0, SSMI_E_SHAPE,
               "conv desc: non-positive dimension");
  SSMI_REQUIRE(d->kh > 0 && d->kw > 0 && d->stride > 0 && d->dil > 0 && d->pad_t >= 0 && d->pad_l >= 0, SSMI_E_SHAPE,
               "conv desc: bad kernel/stride/dilation/pad");
  SSMI_REQUIRE(d->x_ld >= d->cin && d->y_ld >= d->cout, SSMI_E_SHAPE, "conv desc: leading dimension < channels");
  // every output position must read at least ... nothing to check; but the output extent must not exceed what the
  // padded input can produce
  const int64_t need_h = (int64_t)(d->ho - 1) * d->stride + (int64_t)(d->kh - 1) * d->dil + 1 - d->pad_t;
  const int64_t need_w = (int64_t)(d->wo - 1) * d->stride + (int64_t)(d->kw - 1) * d->dil + 1 - d->pad_l;
  // TF SAME allows pad_after up to pad_before + 1
  SSMI_REQUIRE(need_h <= (int64_t)d->h + d->pad_t + 1 + (d->kh - 1) * d->dil && need_w <= (int64_t)d->w + d->pad_l + 1 + (d->kw - 1) * d->dil,
               SSMI_E_SHAPE, "conv desc: output extent inconsistent with input extent");
  return SSMI_OK;
}

static int pick_nt(int c) { return c <= 32 ? 1 : (c <= 64 ? 2 : 4); }

static ssmi_status launch_fwd(const Geo& g, const float* x, const float* w, const float* bias, float* y,
                              hipStream_t st) {
  const int64_t M = (int64_t)g.n * g.ho * g.wo;
  const int nt = pick_nt(g.cout);
  dim3 grid((unsigned)ceil_div64(M, 128), (unsigned)((g.cout + 32 * nt - 1) / (32 * nt)), 1);
  const bool vec = (g.cin % 8 == 0) && (g.x_ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
#define L(NT, V) hipLaunchKernelGGL((conv_fwd_kernel<NT, V>), grid, dim3(256), 0, st, g, x, w, bias, y)
  if (vec) {
    if (nt == 1) L(1, true); else if (nt == 2) L(2, true); else L(4, true);
  } else {
    if (nt == 1) L(1, false); else if (nt == 2) L(2, false); else L(4, false);
  }
#undef L
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

static ssmi_status launch_bwd_data(const Geo& g, const float* dy, const float* w, const float* add, const float* bias,
                                   float* dx, hipStream_t st) {
  const int s = g.stride;
  const int qh = (g.h + s - 1) / s, qw = (g.w + s - 1) / s;
  const int64_t M = (int64_t)g.n * qh * qw;
  const int nt = pick_nt(g.cin);
  dim3 grid((unsigned)ceil_div64(M, 128), (unsigned)((g.cin + 32 * nt - 1) / (32 * nt)), (unsigned)(s * s));
#define L(NT) hipLaunchKernelGGL((conv_bwd_data_kernel<NT>), grid, dim3(256), 0, st, g, dy, w, add, bias, dx)
  if (nt == 1) L(1); else if (nt == 2) L(2); else L(4);
#undef L
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

struct WgPlan {
  int K, K_aug, ktiles, nchunks, rows_per_chunk, nt, cgroups;
  size_t bytes;
};

static WgPlan plan_wg(const Geo& g, bool with_bias) {
  WgPlan p;
  p.K = g.kh * g.kw * g.cin;
  p.K_aug = p.K + (with_bias ? 1 : 0);
  p.ktiles = (p.K_aug + 31) / 32;
  p.nt = pick_nt(g.cout);
  p.cgroups = (g.cout + 32 * p.nt - 1) / (32 * p.nt);
  const int R = g.n * g.ho;
  int target = 2048 / (p.ktiles * p.cgroups);
  if (target < 1) target = 1;
  if (target > R) target = R;
  if (target > 256) target = 256;
  p.rows_per_chunk = (R + target - 1) / target;
  p.nchunks = (R + p.rows_per_chunk - 1) / p.rows_per_chunk;
  p.bytes = (size_t)p.nchunks * p.K_aug * g.cout * sizeof(float);
  return p;
}

static ssmi_status launch_bwd_weight(const Geo& g, const float* x, const float* dy, float* dw, float* dbias, void* ws,
                                     size_t ws_bytes, hipStream_t st) {
  // always plan with the bias row so that the workspace query does not depend on dbias
  WgPlan p = plan_wg(g, dbias != nullptr);
  SSMI_REQUIRE(ws != nullptr && ws_bytes >= p.bytes, SSMI_E_WORKSPACE, "conv2d_bwd_weight: workspace too small (%zu < %zu)",
               ws_bytes, p.bytes);
  float* part = reinterpret_cast<float*>(ws);
  dim3 grid((unsigned)ceil_div64((int64_t)p.nchunks * p.ktiles, 4), (unsigned)p.cgroups, 1);
#define L(NT)                                                                                              \
  hipLaunchKernelGGL((conv_bwd_weight_kernel<NT>), grid, dim3(256), 0, st, g, x, dy, part, p.K, p.K_aug, p.ktiles, \
                     p.rows_per_chunk, p.nchunks)
  if (p.nt == 1) L(1); else if (p.nt == 2) L(2); else L(4);
#undef L
  SSMI_LAUNCH_CHECK();
  const int E = p.K_aug * g.cout;
  hipLaunchKernelGGL(conv_bwd_weight_reduce, dim3((E + 255) / 256), dim3(256), 0, st, part, p.nchunks, p.K, p.K_aug,
                     g.cout, dw, dbias);
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

ssmi_status ssmi_conv2d_fwd(const ssmi_conv_desc* d, const float* x, const float* w, const float* bias, float* y,
                            void* stream) {
  ssmi_status s = check_desc(d);
  if (s) return s;
  SSMI_REQUIRE(x && w && y, SSMI_E_INVALID_ARG, "conv2d_fwd: NULL tensor");
  return launch_fwd(to_geo(d), x, w, bias, y, as_stream(stream));
}

ssmi_status ssmi_conv2d_bwd_data(const ssmi_conv_desc* d, const float* dy, const float* w, const float* add, float* dx,
                                 void* stream) {
  ssmi_status s = check_desc(d);
  if (s) return s;
  SSMI_REQUIRE(dy && w && dx, SSMI_E_INVALID_ARG, "conv2d_bwd_data: NULL tensor");
  return launch_bwd_data(to_geo(d), dy, w, add, nullptr, dx, as_stream(stream));
}

size_t ssmi_conv2d_bwd_weight_workspace_bytes(const ssmi_conv_desc* d) {
  if (check_desc(d)) return 0;
  return plan_wg(to_geo(d), true).bytes;
}

ssmi_status ssmi_conv2d_bwd_weight(const ssmi_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias,
                                   void* ws, size_t ws_bytes, void* stream) {
  ssmi_status s = check_desc(d);
  if (s) return s;
  SSMI_REQUIRE(x && dy && dw, SSMI_E_INVALID_ARG, "conv2d_bwd_weight: NULL tensor");
  return launch_bwd_weight(to_geo(d), x, dy, dw, dbias, ws, ws_bytes, as_stream(stream));
}

ssmi_status ssmi_conv2d_transpose_fwd(const ssmi_conv_desc* d, const float* xt, const float* w, const float* bias,
                                      float* yt, void* stream) {
  ssmi_status s = check_desc(d);
  if (s) return s;
  SSMI_REQUIRE(xt && w && yt, SSMI_E_INVALID_ARG, "conv2d_transpose_fwd: NULL tensor");
  return launch_bwd_data(to_geo(d), xt, w, nullptr, bias, yt, as_stream(stream));
}

ssmi_status ssmi_conv2d_transpose_bwd_data(const ssmi_conv_desc* d, const float* dyt, const float* w, float* dxt,
                                           void* stream) {
  ssmi_status s = check_desc(d);
  if (s) return s;
  SSMI_REQUIRE(dyt && w && dxt, SSMI_E_INVALID_ARG, "conv2d_transpose_bwd_data: NULL tensor");
  return launch_fwd(to_geo(d), dyt, w, nullptr, dxt, as_stream(stream));
}

static int colsum_blocks(int64_t npix, int c) {
  const int ppi = 256 / c;
  int64_t b = ceil_div64(npix, (int64_t)ppi * 16);
  if (b > 512) b = 512;
  if (b < 1) b = 1;
  return (int)b;
}

size_t ssmi_conv2d_transpose_bwd_weight_workspace_bytes(const ssmi_conv_desc* d) {
  if (check_desc(d)) return 0;
  const size_t a = plan_wg(to_geo(d), false).bytes;
  return a + (size_t)512 * d->cin * sizeof(float);
}

ssmi_status ssmi_conv2d_transpose_bwd_weight(const ssmi_conv_desc* d, const float* xt, const float* dyt, float* dw,
                                             float* dbias, void* ws, size_t ws_bytes, void* stream) {
  ssmi_status s = check_desc(d);
  if (s) return s;
  SSMI_REQUIRE(xt && dyt && dw, SSMI_E_INVALID_ARG, "conv2d_transpose_bwd_weight: NULL tensor");
  SSMI_REQUIRE(ws_bytes >= ssmi_conv2d_transpose_bwd_weight_workspace_bytes(d), SSMI_E_WORKSPACE,
               "conv2d_transpose_bwd_weight: workspace too small");
  Geo g = to_geo(d);
  const size_t a = plan_wg(g, false).bytes;
  s = launch_bwd_weight(g, dyt, xt, dw, nullptr, ws, a, as_stream(stream));
  if (s) return s;
  if (dbias) {
    SSMI_REQUIRE(g.cin <= 256, SSMI_E_UNSUPPORTED, "conv2d_transpose_bwd_weight: bias gradient supports <= 256 channels");
    float* part = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + a);
    const int64_t npix = (int64_t)g.n * g.h * g.w;
    const int nb = colsum_blocks(npix, g.cin);
    hipLaunchKernelGGL(colsum_partial, dim3(nb), dim3(256), 0, as_stream(stream), dyt, npix, g.cin, g.x_ld, part);
    SSMI_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_final, dim3((g.cin + 63) / 64), dim3(64), 0, as_stream(stream), part, nb, g.cin, dbias);
    SSMI_LAUNCH_CHECK();
  }
  return SSMI_OK;
}

}  // extern "C"

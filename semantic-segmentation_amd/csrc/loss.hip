// Class-weighted sparse softmax cross-entropy fused with its gradient, the prediction argmax and the streaming
// confusion matrix.  One thread per pixel; the pixel's logits (num_classes floats) are held in registers when
// num_classes is one of the compiled sizes, otherwise re-read from cache.  HBM-bound: reads logits + label + weight,
// writes dlogits.
#include "common.h"
#include <mutex>

namespace ssmi {

static constexpr int kLossBlocks = 2048;

template <int C>
struct LogitRegs {
  float v[C];
};

// Shared epilogue: block-reduce loss, flush LDS confusion matrix.
__device__ __forceinline__ void loss_block_finish(float local_loss, int bad, int* cm_lds, int ncls, double* part,
                                                  int64_t* cm, int32_t* bad_labels) {
  __shared__ float red[4];
  const float ws = wave_sum(local_loss);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = ws;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (double)red[0] + (double)red[1] + (double)red[2] + (double)red[3];
  if (cm) {
    for (int e = threadIdx.x; e < ncls * ncls; e += 256) {
      const int v = cm_lds[e];
      if (v) atomicAdd(reinterpret_cast<unsigned long long*>(cm + e), (unsigned long long)v);
    }
  }
  if (bad_labels && bad) atomicAdd(bad_labels, bad);
}

template <typename T, int C>
__global__ __launch_bounds__(256) void xent_kernel(const T* __restrict__ logits, const int32_t* __restrict__ labels,
                                                   const float* __restrict__ weights, int64_t npix, float inv_npix,
                                                   float grad_scale, T* __restrict__ dlogits, double* __restrict__ part,
                                                   int64_t* __restrict__ cm, int64_t npix_metrics, uint64_t ignore_mask,
                                                   int32_t* __restrict__ pred_out, int32_t* __restrict__ bad_labels) {
  extern __shared__ int cm_lds[];
  if (cm) {
    for (int e = threadIdx.x; e < C * C; e += 256) cm_lds[e] = 0;
    __syncthreads();
  }
  float local = 0.f;
  int bad = 0;
  const int64_t nloop = npix_metrics > npix ? npix_metrics : npix;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < nloop; p += (int64_t)gridDim.x * 256) {
    const bool in_loss = p < npix;   // rows past npix feed the confusion matrix only
    const T* zp = logits + p * C;
    float z[C];
    if (C % 4 == 0) {
#pragma unroll
      for (int j = 0; j < C / 4; ++j) {
        const f32x4 t = Act<T>::gld4(zp + 4 * j);
        z[4 * j] = t[0]; z[4 * j + 1] = t[1]; z[4 * j + 2] = t[2]; z[4 * j + 3] = t[3];
      }
    } else {
#pragma unroll
      for (int j = 0; j < C; ++j) z[j] = Act<T>::gld(zp + j);
    }
    float mx = z[0];
    int am = 0;
#pragma unroll
    for (int j = 1; j < C; ++j)
      if (z[j] > mx) { mx = z[j]; am = j; }
    float se = 0.f;
    float e[C];
#pragma unroll
    for (int j = 0; j < C; ++j) { e[j] = __expf(z[j] - mx); se += e[j]; }
    const int lab = labels[p];
    const bool lv = (unsigned)lab < (unsigned)C;
    const float wgt = (lv && in_loss) ? weights[p] : 0.f;
    if (!lv) ++bad;
    float zl = 0.f;
#pragma unroll
    for (int j = 0; j < C; ++j) zl = (j == lab) ? z[j] : zl;
    const float lse = mx + __logf(se);
    local += wgt * (lse - zl);
    if (dlogits && in_loss) {
      const float k = grad_scale * wgt * inv_npix;
      const float inv = 1.f / se;
      T* dp = dlogits + p * C;
      if (C % 4 == 0) {
#pragma unroll
        for (int j = 0; j < C / 4; ++j) {
          f32x4 t;
#pragma unroll
          for (int q = 0; q < 4; ++q) t[q] = k * (e[4 * j + q] * inv - ((4 * j + q) == lab ? 1.f : 0.f));
          Act<T>::gst4(dp + 4 * j, t);
        }
      } else {
#pragma unroll
        for (int j = 0; j < C; ++j) Act<T>::gst(dp + j, k * (e[j] * inv - (j == lab ? 1.f : 0.f)));
      }
    }
    if (pred_out) pred_out[p] = am;
    if (cm && lv && p < npix_metrics && !((ignore_mask >> lab) & 1ull)) atomicAdd(&cm_lds[lab * C + am], 1);
  }
  if (cm) __syncthreads();
  loss_block_finish(local, bad, cm_lds, C, part, cm, bad_labels);
}

// any class count up to 64: logits re-read three times (served by L1/L2)
template <typename T>
__global__ __launch_bounds__(256) void xent_kernel_generic(const T* __restrict__ logits, const int32_t* __restrict__ labels,
                                                           const float* __restrict__ weights, int64_t npix, int C,
                                                           float inv_npix, float grad_scale, T* __restrict__ dlogits,
                                                           double* __restrict__ part, int64_t* __restrict__ cm,
                                                           int64_t npix_metrics, uint64_t ignore_mask,
                                                           int32_t* __restrict__ pred_out, int32_t* __restrict__ bad_labels) {
  extern __shared__ int cm_lds[];
  if (cm) {
    for (int e = threadIdx.x; e < C * C; e += 256) cm_lds[e] = 0;
    __syncthreads();
  }
  float local = 0.f;
  int bad = 0;
  const int64_t nloop = npix_metrics > npix ? npix_metrics : npix;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < nloop; p += (int64_t)gridDim.x * 256) {
    const bool in_loss = p < npix;
    const T* zp = logits + p * C;
    float mx = Act<T>::gld(zp);
    int am = 0;
    for (int j = 1; j < C; ++j) {
      const float v = Act<T>::gld(zp + j);
      if (v > mx) { mx = v; am = j; }
    }
    float se = 0.f;
    for (int j = 0; j < C; ++j) se += __expf(Act<T>::gld(zp + j) - mx);
    const int lab = labels[p];
    const bool lv = (unsigned)lab < (unsigned)C;
    const float wgt = (lv && in_loss) ? weights[p] : 0.f;
    if (!lv) ++bad;
    const float zl = lv ? Act<T>::gld(zp + lab) : 0.f;
    local += wgt * (mx + __logf(se) - zl);
    if (dlogits && in_loss) {
      const float k = grad_scale * wgt * inv_npix;
      const float inv = 1.f / se;
      T* dp = dlogits + p * C;
      for (int j = 0; j < C; ++j) Act<T>::gst(dp + j, k * (__expf(Act<T>::gld(zp + j) - mx) * inv - (j == lab ? 1.f : 0.f)));
    }
    if (pred_out) pred_out[p] = am;
    if (cm && lv && p < npix_metrics && !((ignore_mask >> lab) & 1ull)) atomicAdd(&cm_lds[lab * C + am], 1);
  }
  if (cm) __syncthreads();
  loss_block_finish(local, bad, cm_lds, C, part, cm, bad_labels);
}

__global__ void xent_final(const double* __restrict__ part, int nblocks, double inv_npix, float* __restrict__ loss_out) {
  __shared__ double sm[256];
  double s = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 256) s += part[b];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *loss_out = (float)(sm[0] * inv_npix);
}

// ------------------------------------------------------------------------------------------------
// The network's last layer and the loss in one pass: logits = Conv2DTranspose(2x2, stride 2, 'same')(x) + bias
// (src/models.py:1318, every output pixel reads ONE input pixel through the tap its parity selects) feed the loss
// arithmetic of xent_kernel straight from registers.  The logits tensor (num_classes floats per full-resolution pixel,
// the largest tensor of the network) is neither written nor read back unless the caller asks for it; the evaluation
// phase then moves the 16-channel half-resolution input, the labels and the weights only.
// One thread per OUTPUT pixel; the kernel [tap][cout][cin] is staged once per workgroup as [tap][cin][cout] so that a
// 16-byte LDS read yields four output channels of one input channel (lanes of one parity read the same address).
// bf16 storage: the logits are rounded to bf16 before the loss sees them - the stored logits, when requested, are the
// values the loss, the argmax and the gradient were computed from, as in the two-kernel form.
// ------------------------------------------------------------------------------------------------
// GRAD = false: the evaluation form (no gradient, no staging block; dlogits is ignored).
template <typename T, int C, int CIN, bool GRAD>
__global__ __launch_bounds__(256) void head2x2_xent_kernel(const T* __restrict__ x, int x_ld, const float* __restrict__ w,
                                                           const float* __restrict__ bias, int nimg, int hin, int win,
                                                           const int32_t* __restrict__ labels, const float* __restrict__ weights,
                                                           int npix, float inv_npix, float grad_scale, T* __restrict__ logits_out,
                                                           T* __restrict__ dlogits, double* __restrict__ part,
                                                           int64_t* __restrict__ cm, int npix_metrics, uint64_t ignore_mask,
                                                           int32_t* __restrict__ pred_out, int32_t* __restrict__ bad_labels) {
  extern __shared__ __attribute__((aligned(16))) int head_lds[];
  float* wl = reinterpret_cast<float*>(head_lds);            // [4][CIN][C]
  float* bl = wl + 4 * CIN * C;                               // [C]
  int* cm_lds = head_lds + 4 * CIN * C + C;                   // [C][C]
  // gradient rows on their way out, per wave: [64 pixels][C] floats + the pixels' indices (see `pixel` below)
  float* stg = reinterpret_cast<float*>(cm_lds + C * C) + (threadIdx.x >> 6) * (64 * C + 64);
  int* stp = reinterpret_cast<int*>(stg + 64 * C);
  const int lane = threadIdx.x & 63;
  static_assert(C % 4 == 0 && CIN % 4 == 0, "16-byte accesses");
  for (int e = threadIdx.x; e < 4 * C * CIN; e += 256) {
    const int tap = e / (C * CIN), rem = e % (C * CIN), co = rem / CIN, ci = rem % CIN;
    wl[(tap * CIN + ci) * C + co] = w[e];
  }
  if (threadIdx.x < C) bl[threadIdx.x] = bias ? bias[threadIdx.x] : 0.f;
  if (cm)
    for (int e = threadIdx.x; e < C * C; e += 256) cm_lds[e] = 0;
  __syncthreads();
  const int wout = 2 * win, hout = 2 * hin;
  float local = 0.f;
  int bad = 0;
  const int nloop = npix_metrics > npix ? npix_metrics : npix;
  // A thread owns two output pixels of one row that share a tap: columns ox and ox + 2 (input columns 2 q and 2 q + 1), so
  // every kernel word read from LDS feeds two pixels.  Work item g -> (row, q, tx): consecutive lanes alternate tx.
  const int npair = (win + 1) >> 1;
  const int nwork = nimg * hout * npair * 2;
  // (whole waves walk the loop together: the lanes of a wave hand their gradient rows to each other, below)
  for (int g0 = blockIdx.x * 256 + (threadIdx.x & ~63); g0 < nwork; g0 += gridDim.x * 256) {
    const bool act0 = g0 + lane < nwork;
    const int g = act0 ? g0 + lane : 0;
    const int tx = g & 1, pr = g >> 1;
    const int row = pr / npair, q = pr - row * npair;
    const int img = row / hout, oy = row - img * hout;
    const int pa = row * wout + 4 * q + tx;               // pixel indices in the order of the label / logits tensors
    const bool hasb = 2 * q + 1 < win;
    const bool act = act0 && pa < nloop;
    const T* xp = x + (int64_t)((img * hin + (oy >> 1)) * win + 2 * q) * x_ld;
    float xa[CIN], xb[CIN];
#pragma unroll
    for (int j = 0; j < CIN / 4; ++j) {
      const f32x4 t = Act<T>::gld4(xp + 4 * j);
      xa[4 * j] = t[0]; xa[4 * j + 1] = t[1]; xa[4 * j + 2] = t[2]; xa[4 * j + 3] = t[3];
    }
    const T* xq = hasb ? xp + x_ld : xp;
#pragma unroll
    for (int j = 0; j < CIN / 4; ++j) {
      const f32x4 t = Act<T>::gld4(xq + 4 * j);
      xb[4 * j] = t[0]; xb[4 * j + 1] = t[1]; xb[4 * j + 2] = t[2]; xb[4 * j + 3] = t[3];
    }
    const float* wt = wl + (((oy & 1) * 2 + tx) * CIN) * C;
    f32x4 za[C / 4], zb[C / 4];
#pragma unroll
    for (int m = 0; m < C / 4; ++m) za[m] = zb[m] = *reinterpret_cast<const f32x4*>(bl + 4 * m);
    // kernel words of input channel ci + 1 are fetched while channel ci is multiplied; the scheduling barrier keeps the
    // compiler from hoisting all 96 LDS reads in front of the first multiply (which took 500 registers)
    constexpr int HB = GRAD ? 2 : 1;      // half-channel blocks of kernel words in flight (the gradient form is short of registers)
    f32x4 wv[2][C / 4 / HB];
#pragma unroll
    for (int m = 0; m < C / 4 / HB; ++m) wv[0][m] = *reinterpret_cast<const f32x4*>(wt + 4 * m);
#pragma unroll
    for (int s = 0; s < CIN * HB; ++s) {
      const int ci = s / HB, h0 = (s % HB) * (C / 4 / HB);
      if (s + 1 < CIN * HB) {
        const int cn = (s + 1) / HB, hn = ((s + 1) % HB) * (C / 4 / HB);
#pragma unroll
        for (int m = 0; m < C / 4 / HB; ++m) wv[(s + 1) & 1][m] = *reinterpret_cast<const f32x4*>(wt + cn * C + 4 * (hn + m));
      }
      const f32x4 sa = {xa[ci], xa[ci], xa[ci], xa[ci]}, sb = {xb[ci], xb[ci], xb[ci], xb[ci]};
#pragma unroll
      for (int m = 0; m < C / 4 / HB; ++m) {
        za[h0 + m] = __builtin_elementwise_fma(sa, wv[s & 1][m], za[h0 + m]);
        zb[h0 + m] = __builtin_elementwise_fma(sb, wv[s & 1][m], zb[h0 + m]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // the logits are used under conditions only; without this the compiler moves their multiplies into those branches,
    // behind all the kernel words
#pragma unroll
    for (int m = 0; m < C / 4; ++m) asm volatile("" : "+v"(za[m]), "+v"(zb[m]));
    // the loss arithmetic of xent_kernel on one pixel's logits
    auto pixel = [&](f32x4 (&zq)[C / 4], int p) {
      const bool in_loss = p < npix;
      float z[C];
#pragma unroll
      for (int m = 0; m < C / 4; ++m) {
        if constexpr (sizeof(T) == 2) zq[m] = unpack_bf16x4(pack_bf16x4(zq[m]));      // the logits as bf16 storage holds them
        z[4 * m] = zq[m][0]; z[4 * m + 1] = zq[m][1]; z[4 * m + 2] = zq[m][2]; z[4 * m + 3] = zq[m][3];
      }
      if (logits_out) {
        T* lp = logits_out + (int64_t)p * C;
#pragma unroll
        for (int m = 0; m < C / 4; ++m) Act<T>::gst4(lp + 4 * m, zq[m]);
      }
      float mx = z[0];
      int am = 0;
#pragma unroll
      for (int j = 1; j < C; ++j)
        if (z[j] > mx) { mx = z[j]; am = j; }
      float se = 0.f;
      float e[C];
#pragma unroll
      for (int j = 0; j < C; ++j) { e[j] = __expf(z[j] - mx); se += e[j]; }
      const int lab = labels[p];
      const bool lv = (unsigned)lab < (unsigned)C;
      const float wgt = (lv && in_loss) ? weights[p] : 0.f;
      if (!lv) ++bad;
      float zl = 0.f;
#pragma unroll
      for (int j = 0; j < C; ++j) zl = (j == lab) ? z[j] : zl;
      const float lse = mx + __logf(se);
      local += wgt * (lse - zl);
      if (GRAD && in_loss) {
        const float k = grad_scale * wgt * inv_npix;
        const float inv = 1.f / se;
#pragma unroll
        for (int j = 0; j < C / 4; ++j) {
          f32x4 t;
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) t[q4] = k * (e[4 * j + q4] * inv - ((4 * j + q4) == lab ? 1.f : 0.f));
          *reinterpret_cast<f32x4*>(stg + lane * C + 4 * j) = t;
        }
        stp[lane] = p;
      }
      if (pred_out) pred_out[p] = am;
      if (cm && lv && p < npix_metrics && !((ignore_mask >> lab) & 1ull)) atomicAdd(&cm_lds[lab * C + am], 1);
    };
    // Gradient rows leave through the wave's LDS block: a lane's row is C consecutive floats, so stored straight from
    // the registers every store instruction would touch 64 different cache lines with 16 bytes each; read back chunk by
    // chunk (lane l takes 16-byte chunks l, l + 64, ... of the 64 rows) consecutive lanes write consecutive addresses
    // within a row and within the two neighbouring rows of a lane pair.
    auto flush = [&]() {
      if (!GRAD) return;
#pragma unroll
      for (int k = 0; k < C / 4; ++k) {
        const int c = lane + 64 * k, pi = c / (C / 4), part = c - pi * (C / 4);
        const int pp = stp[pi];
        const f32x4 v = *reinterpret_cast<const f32x4*>(stg + pi * C + 4 * part);
        if (pp >= 0) Act<T>::gst4(dlogits + (int64_t)pp * C + 4 * part, v);
      }
    };
    if (GRAD) stp[lane] = -1;
    if (act) pixel(za, pa);
    flush();
    if (GRAD) stp[lane] = -1;
    if (act && hasb && pa + 2 < nloop) pixel(zb, pa + 2);
    flush();
  }
  if (cm) __syncthreads();
  loss_block_finish(local, bad, cm_lds, C, part, cm, bad_labels);
}


// Mean-teacher consistency cost (src/losses.py:195-215): coef * mean over pixels of mean_c (softmax(teacher) -
// softmax(student))^2, with its gradient w.r.t. the student logits.  CT > 0 keeps both softmax rows in registers;
// CT == 0 handles any class count by re-reading the rows (served by L1/L2).  Pixels below `npix_acc` already hold the
// classification gradient and are accumulated into, the remaining pixels are overwritten.
template <int CT>
__global__ __launch_bounds__(256) void mt_consistency_kernel(const float* __restrict__ logits, const float* __restrict__ teacher,
                                                             int64_t npix, int Cdyn, const float* __restrict__ coef_dev,
                                                             float kscale, int64_t npix_acc, float* __restrict__ dlogits,
                                                             double* __restrict__ part) {
  const int C = CT > 0 ? CT : Cdyn;
  const float coef = *coef_dev;
  const float k = 2.f * coef * kscale;   // kscale = grad_scale / (C * npix)
  float local = 0.f;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (int64_t)gridDim.x * 256) {
    const float* zp = logits + p * C;
    const float* tp = teacher + p * C;
    if (CT > 0) {
      float s[CT > 0 ? CT : 1], t[CT > 0 ? CT : 1];
      if (CT % 4 == 0) {
#pragma unroll
        for (int j = 0; j < CT / 4; ++j) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(zp + 4 * j);
          const f32x4 b = *reinterpret_cast<const f32x4*>(tp + 4 * j);
#pragma unroll
          for (int q = 0; q < 4; ++q) { s[4 * j + q] = a[q]; t[4 * j + q] = b[q]; }
        }
      } else {
#pragma unroll
        for (int j = 0; j < CT; ++j) { s[j] = zp[j]; t[j] = tp[j]; }
      }
      float ms = s[0], mt = t[0];
#pragma unroll
      for (int j = 1; j < CT; ++j) { ms = fmaxf(ms, s[j]); mt = fmaxf(mt, t[j]); }
      float ss = 0.f, st = 0.f;
#pragma unroll
      for (int j = 0; j < CT; ++j) { s[j] = __expf(s[j] - ms); ss += s[j]; t[j] = __expf(t[j] - mt); st += t[j]; }
      const float is = 1.f / ss, it = 1.f / st;
      float sq = 0.f, dot = 0.f;
#pragma unroll
      for (int j = 0; j < CT; ++j) {
        s[j] *= is;
        t[j] = s[j] - t[j] * it;          // d_j = s_j - t_j
        sq += t[j] * t[j];
        dot += t[j] * s[j];
      }
      local += sq;
      if (dlogits) {
        float* dp = dlogits + p * C;
        const bool acc = p < npix_acc;
        if (CT % 4 == 0) {
#pragma unroll
          for (int j = 0; j < CT / 4; ++j) {
            f32x4 g;
#pragma unroll
            for (int q = 0; q < 4; ++q) g[q] = k * s[4 * j + q] * (t[4 * j + q] - dot);
            if (acc) {
              const f32x4 o = *reinterpret_cast<const f32x4*>(dp + 4 * j);
              g += o;
            }
            *reinterpret_cast<f32x4*>(dp + 4 * j) = g;
          }
        } else {
#pragma unroll
          for (int j = 0; j < CT; ++j) {
            const float g = k * s[j] * (t[j] - dot);
            dp[j] = acc ? dp[j] + g : g;
          }
        }
      }
    } else {
      float ms = zp[0], mt = tp[0];
      for (int j = 1; j < C; ++j) { ms = fmaxf(ms, zp[j]); mt = fmaxf(mt, tp[j]); }
      float ss = 0.f, st = 0.f;
      for (int j = 0; j < C; ++j) { ss += __expf(zp[j] - ms); st += __expf(tp[j] - mt); }
      const float is = 1.f / ss, it = 1.f / st;
      float sq = 0.f, dot = 0.f;
      for (int j = 0; j < C; ++j) {
        const float sj = __expf(zp[j] - ms) * is;
        const float d = sj - __expf(tp[j] - mt) * it;
        sq += d * d;
        dot += d * sj;
      }
      local += sq;
      if (dlogits) {
        float* dp = dlogits + p * C;
        const bool acc = p < npix_acc;
        for (int j = 0; j < C; ++j) {
          const float sj = __expf(zp[j] - ms) * is;
          const float d = sj - __expf(tp[j] - mt) * it;
          const float g = k * sj * (d - dot);
          dp[j] = acc ? dp[j] + g : g;
        }
      }
    }
  }
  __shared__ float red[4];
  const float ws = wave_sum(local);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = ws;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (double)red[0] + (double)red[1] + (double)red[2] + (double)red[3];
}

__global__ void mt_consistency_final(const double* __restrict__ part, int nblocks, double inv_count,
                                     const float* __restrict__ coef_dev, int accumulate, float* __restrict__ loss_io,
                                     float* __restrict__ cost_out) {
  __shared__ double sm[256];
  double s = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 256) s += part[b];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float cost = (float)((double)*coef_dev * (sm[0] * inv_count));
    if (cost_out) *cost_out = cost;
    if (loss_io) *loss_io = accumulate ? *loss_io + cost : cost;
  }
}


// ------------------------------------------------------------------------------------------------
// Superpixel cost of the unlabeled rows (src/losses.py:108-192): Sobel gradients of the per-class softmax maps
// (3x3 depthwise, SAME zero padding, per image), clipped from below at epsilon, combined to one magnitude per pixel
// (sqrt over classes of Gx^2 + Gy^2), masked by the superpixel-interior mask and averaged over all pixels.
// Three passes over an [npix, C] workspace triple: softmax S -> (dGx, dGy) + cost -> dlogits.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sp_softmax_kernel(const float* __restrict__ logits, unsigned npix, int C,
                                                         float* __restrict__ S) {
  for (unsigned p = blockIdx.x * 256u + threadIdx.x; p < npix; p += gridDim.x * 256u) {
    const float* z = logits + (size_t)p * C;
    float* o = S + (size_t)p * C;
    float mx = z[0];
    for (int j = 1; j < C; ++j) mx = fmaxf(mx, z[j]);
    float se = 0.f;
    for (int j = 0; j < C; ++j) se += __expf(z[j] - mx);
    const float inv = 1.f / se;
    for (int j = 0; j < C; ++j) o[j] = __expf(z[j] - mx) * inv;
  }
}

// Cross-correlation with S_x = [[1,2,1],[0,0,0],[-1,-2,-1]] (gx) and its transpose (gy) at pixel (y, x), class c.
// `up`/`dn`/`lf`/`rt` say whether the neighbouring row / column exists inside the image.
__device__ __forceinline__ void sobel_pair(const float* __restrict__ T, int wc, int C, int c, bool up, bool dn, bool lf, bool rt,
                                           float& gx, float& gy) {
  // T points at (y, x, 0); wc = w * C
  const float a00 = (up && lf) ? T[-wc - C + c] : 0.f, a01 = up ? T[-wc + c] : 0.f, a02 = (up && rt) ? T[-wc + C + c] : 0.f;
  const float a10 = lf ? T[-C + c] : 0.f, a12 = rt ? T[C + c] : 0.f;
  const float a20 = (dn && lf) ? T[wc - C + c] : 0.f, a21 = dn ? T[wc + c] : 0.f, a22 = (dn && rt) ? T[wc + C + c] : 0.f;
  gx = (a00 + 2.f * a01 + a02) - (a20 + 2.f * a21 + a22);
  gy = (a00 + 2.f * a10 + a20) - (a02 + 2.f * a12 + a22);
}

__global__ __launch_bounds__(256) void sp_grad_kernel(const float* __restrict__ S, const int32_t* __restrict__ mask, int h, int w,
                                                      unsigned npix, int C, const float* __restrict__ coef_dev, float kscale,
                                                      float eps, float* __restrict__ dGx, float* __restrict__ dGy,
                                                      double* __restrict__ part) {
  const float k = *coef_dev * kscale;      // coef * grad_scale / npix
  const int wc = w * C;
  float local = 0.f;
  for (unsigned p = blockIdx.x * 256u + threadIdx.x; p < npix; p += gridDim.x * 256u) {
    const unsigned rem = p % ((unsigned)h * (unsigned)w);
    const int y = (int)(rem / (unsigned)w), x = (int)(rem % (unsigned)w);
    const bool up = y > 0, dn = y < h - 1, lf = x > 0, rt = x < w - 1;
    const float* T = S + (size_t)p * C;
    float sum = 0.f;
    for (int c = 0; c < C; ++c) {
      float gx, gy;
      sobel_pair(T, wc, C, c, up, dn, lf, rt, gx, gy);
      gx = fmaxf(gx, eps);
      gy = fmaxf(gy, eps);
      sum += gx * gx + gy * gy;
    }
    const float gdot = sqrtf(sum);
    const int mv = mask[p];
    const float m = mv > 0 ? 1.f : 0.f;      // clip_by_value(y_true, 0, 1) of an integer mask
    local += gdot * m;
    if (dGx) {
      const float f = k * m / gdot;
      for (int c = 0; c < C; ++c) {
        float gx, gy;
        sobel_pair(T, wc, C, c, up, dn, lf, rt, gx, gy);
        // the clip passes the gradient where the raw value is inside [epsilon, max]
        dGx[(size_t)p * C + c] = gx >= eps ? f * gx : 0.f;
        dGy[(size_t)p * C + c] = gy >= eps ? f * gy : 0.f;
      }
    }
  }
  __shared__ float red[4];
  const float ws = wave_sum(local);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = ws;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (double)red[0] + (double)red[1] + (double)red[2] + (double)red[3];
}

// ds = transposed Sobel stencils applied to (dGx, dGy), then the softmax backward into the logits gradient.
__device__ __forceinline__ float sp_ds(const float* __restrict__ X, const float* __restrict__ Y, int wc, int C, int c, bool up,
                                       bool dn, bool lf, bool rt) {
  // the transpose of a cross-correlation with K is a cross-correlation with K rotated by 180 degrees: -S_x and -S_y
  float gx, gy, t;
  sobel_pair(X, wc, C, c, up, dn, lf, rt, gx, t);
  sobel_pair(Y, wc, C, c, up, dn, lf, rt, t, gy);
  return -(gx + gy);
}

__global__ __launch_bounds__(256) void sp_apply_kernel(const float* __restrict__ S, const float* __restrict__ dGx,
                                                       const float* __restrict__ dGy, int h, int w, unsigned npix, int C,
                                                       int accumulate, float* __restrict__ dlogits) {
  const int wc = w * C;
  for (unsigned p = blockIdx.x * 256u + threadIdx.x; p < npix; p += gridDim.x * 256u) {
    const unsigned rem = p % ((unsigned)h * (unsigned)w);
    const int y = (int)(rem / (unsigned)w), x = (int)(rem % (unsigned)w);
    const bool up = y > 0, dn = y < h - 1, lf = x > 0, rt = x < w - 1;
    const float* T = S + (size_t)p * C;
    const float* X = dGx + (size_t)p * C;
    const float* Y = dGy + (size_t)p * C;
    float dot = 0.f;
    for (int c = 0; c < C; ++c) dot += sp_ds(X, Y, wc, C, c, up, dn, lf, rt) * T[c];
    float* o = dlogits + (size_t)p * C;
    for (int c = 0; c < C; ++c) {
      const float g = T[c] * (sp_ds(X, Y, wc, C, c, up, dn, lf, rt) - dot);
      o[c] = accumulate ? o[c] + g : g;
    }
  }
}


// Classification cost of the encoder-only models (src/losses.py:579-588): soft / multi-hot labels y [n, C], per-sample
// per-class weights w [n, C]:  loss = -mean_n sum_c y * log(clip(softmax(z), eps, 1)) * w.  The clip passes no gradient
// where the softmax is below eps.  One workgroup; thread = sample; the per-sample terms are added in order (deterministic).
__global__ __launch_bounds__(256) void classification_xent_kernel(const float* __restrict__ logits, const float* __restrict__ y,
                                                                  const float* __restrict__ w, int n, int C, float eps,
                                                                  float grad_scale, float* __restrict__ dlogits,
                                                                  float* __restrict__ loss_out) {
  __shared__ float sm[256];
  float local = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float* z = logits + (int64_t)i * C;
    const float* yi = y + (int64_t)i * C;
    const float* wi = w + (int64_t)i * C;
    float mx = z[0];
    for (int j = 1; j < C; ++j) mx = fmaxf(mx, z[j]);
    float se = 0.f;
    for (int j = 0; j < C; ++j) se += __expf(z[j] - mx);
    const float inv = 1.f / se;
    float li = 0.f, gsum = 0.f;      // gsum = sum_c (dL/ds_c) * s_c over the unclipped classes
    for (int j = 0; j < C; ++j) {
      const float s = __expf(z[j] - mx) * inv;
      const float sc = fminf(fmaxf(s, eps), 1.f);
      li += yi[j] * __logf(sc) * wi[j];
      if (s >= eps) gsum += yi[j] * wi[j];     // d(log s)/ds * s = 1
    }
    local -= li;
    if (dlogits) {
      const float k = grad_scale / (float)n;
      float* d = dlogits + (int64_t)i * C;
      for (int j = 0; j < C; ++j) {
        const float s = __expf(z[j] - mx) * inv;
        const float a = s >= eps ? yi[j] * wi[j] : 0.f;      // -dL_i/dlog(s_j)
        d[j] = -k * (a - s * gsum);
      }
    }
  }
  sm[threadIdx.x] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    const int m = n < 256 ? n : 256;
    for (int i = 0; i < m; ++i) t += sm[i];
    *loss_out = t / (float)n;
  }
}

// teacher = a * teacher + (1 - a) * student, evaluated in this order (src/trainers.py:1704)
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ teacher, const float* __restrict__ student, float a,
                                                  float one_minus_a, int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256)
    teacher[i] = __fadd_rn(__fmul_rn(a, teacher[i]), __fmul_rn(one_minus_a, student[i]));
}

}  // namespace ssmi

using namespace ssmi;

extern "C" {

size_t ssmi_softmax_xent_workspace_bytes(int64_t npix) {
  (void)npix;
  return (size_t)kLossBlocks * sizeof(double);
}

}  // extern "C"

template <typename T>
static ssmi_status softmax_xent_impl(const T* logits, const int32_t* labels, const float* weights, int64_t npix,
                                     int32_t num_classes, float grad_scale, T* dlogits, float* loss_out, int64_t* cm,
                                     int64_t npix_metrics, uint64_t ignore_mask, int32_t* pred_out, int32_t* bad_labels, void* ws,
                                     size_t ws_bytes, void* stream) {
  SSMI_REQUIRE(logits && labels && weights && loss_out && npix > 0, SSMI_E_INVALID_ARG, "softmax_xent: bad argument");
  SSMI_REQUIRE(num_classes >= 2 && num_classes <= 64, SSMI_E_UNSUPPORTED, "softmax_xent: num_classes %d outside [2,64]", num_classes);
  SSMI_REQUIRE(ws && ws_bytes >= ssmi_softmax_xent_workspace_bytes(npix), SSMI_E_WORKSPACE, "softmax_xent: workspace too small");
  SSMI_REQUIRE(npix_metrics >= 0, SSMI_E_INVALID_ARG, "softmax_xent: npix_metrics out of range");
  hipStream_t st = as_stream(stream);
  int64_t nb64 = ceil_div64(npix_metrics > npix ? npix_metrics : npix, 256);
  if (nb64 > kLossBlocks) nb64 = kLossBlocks;
  const int nb = (int)nb64;
  double* part = reinterpret_cast<double*>(ws);
  const float inv = (float)(1.0 / (double)npix);
  const size_t lds = cm ? (size_t)num_classes * num_classes * sizeof(int) : 0;
  const bool aligned = ((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(dlogits)) & 15) == 0;
  if (num_classes == 24 && aligned) {
    hipLaunchKernelGGL((xent_kernel<T, 24>), dim3(nb), dim3(256), lds, st, logits, labels, weights, npix, inv, grad_scale,
                       dlogits, part, cm, npix_metrics, ignore_mask, pred_out, bad_labels);
  } else {
    hipLaunchKernelGGL(xent_kernel_generic<T>, dim3(nb), dim3(256), lds, st, logits, labels, weights, npix, num_classes, inv,
                       grad_scale, dlogits, part, cm, npix_metrics, ignore_mask, pred_out, bad_labels);
  }
  SSMI_LAUNCH_CHECK();
  hipLaunchKernelGGL(xent_final, dim3(1), dim3(256), 0, st, part, nb, 1.0 / (double)npix, loss_out);
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

extern "C" {

ssmi_status ssmi_softmax_xent_fwd_bwd(const float* logits, const int32_t* labels, const float* weights, int64_t npix,
                                      int32_t num_classes, float grad_scale, float* dlogits, float* loss_out,
                                      int64_t* cm, int64_t npix_metrics, uint64_t ignore_mask, int32_t* pred_out,
                                      int32_t* bad_labels, void* ws, size_t ws_bytes, void* stream) {
  return softmax_xent_impl<float>(logits, labels, weights, npix, num_classes, grad_scale, dlogits, loss_out, cm, npix_metrics,
                                  ignore_mask, pred_out, bad_labels, ws, ws_bytes, stream);
}

ssmi_status ssmi_softmax_xent_fwd_bwd_bf16(const uint16_t* logits, const int32_t* labels, const float* weights, int64_t npix,
                                           int32_t num_classes, float grad_scale, uint16_t* dlogits, float* loss_out,
                                           int64_t* cm, int64_t npix_metrics, uint64_t ignore_mask, int32_t* pred_out,
                                           int32_t* bad_labels, void* ws, size_t ws_bytes, void* stream) {
  return softmax_xent_impl<bf16>(reinterpret_cast<const bf16*>(logits), labels, weights, npix, num_classes, grad_scale,
                                 reinterpret_cast<bf16*>(dlogits), loss_out, cm, npix_metrics, ignore_mask, pred_out, bad_labels,
                                 ws, ws_bytes, stream);
}

int32_t ssmi_logits_tconv2x2_xent_supported(int32_t cin, int32_t num_classes) { return cin == 16 && num_classes == 24; }

ssmi_status ssmi_logits_tconv2x2_xent_fwd_bwd(int32_t dtype, const void* x, int32_t n, int32_t h, int32_t w, int32_t cin,
                                              int32_t x_ld, const float* kernel, const float* bias, int32_t num_classes,
                                              const int32_t* labels, const float* weights, int64_t npix, float grad_scale,
                                              void* logits_out, void* dlogits, float* loss_out, int64_t* cm,
                                              int64_t npix_metrics, uint64_t ignore_mask, int32_t* pred_out,
                                              int32_t* bad_labels, void* ws, size_t ws_bytes, void* stream) {
  SSMI_REQUIRE(x && kernel && labels && weights && loss_out && n > 0 && h > 0 && w > 0 && npix > 0, SSMI_E_INVALID_ARG,
               "logits_tconv2x2_xent: bad argument");
  SSMI_REQUIRE(ssmi_logits_tconv2x2_xent_supported(cin, num_classes), SSMI_E_UNSUPPORTED,
               "logits_tconv2x2_xent: built for 16 input channels and 24 classes (got %d, %d)", cin, num_classes);
  const int64_t total = (int64_t)n * (2 * h) * (2 * w);
  SSMI_REQUIRE(npix <= total && npix_metrics >= 0 && npix_metrics <= total, SSMI_E_INVALID_ARG,
               "logits_tconv2x2_xent: npix / npix_metrics exceed the %lld output pixels", (long long)total);
  SSMI_REQUIRE(total * num_classes < 0x7FFFFFFFLL && x_ld >= cin && x_ld % 4 == 0, SSMI_E_UNSUPPORTED,
               "logits_tconv2x2_xent: tensor too large for 32-bit pixel indices, or row stride %d not a multiple of 4", x_ld);
  SSMI_REQUIRE(ws && ws_bytes >= ssmi_softmax_xent_workspace_bytes(npix), SSMI_E_WORKSPACE, "logits_tconv2x2_xent: workspace too small");
  const size_t esz = dtype == SSMI_BF16 ? 2 : 4;
  SSMI_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(logits_out) | reinterpret_cast<uintptr_t>(dlogits)) &
                (4 * esz - 1)) == 0, SSMI_E_INVALID_ARG, "logits_tconv2x2_xent: tensors must be aligned to four elements");
  hipStream_t st = as_stream(stream);
  int64_t nb64 = ceil_div64((int64_t)n * 2 * h * ((w + 1) / 2) * 2, 256);      // work items: two pixels of a row each
  // whole rounds of resident workgroups: two or three workgroups per CU fit (512 or 768 in all), 1536 is 3 or 2 rounds
  const int64_t cap = 1536;
  if (nb64 > cap) nb64 = cap;
  const int nb = (int)nb64;
  double* part = reinterpret_cast<double*>(ws);
  const float inv = (float)(1.0 / (double)npix);
  const size_t lds_eval = (size_t)(4 * 16 * 24 + 24 + 24 * 24) * sizeof(int), lds_grad = lds_eval + (size_t)4 * (64 * 24 + 64) * sizeof(int);
  const ssmi_status s = dispatch_dtype(dtype, [&](auto tag) {
    using T = decltype(tag);
    if (dlogits)
      hipLaunchKernelGGL((head2x2_xent_kernel<T, 24, 16, true>), dim3(nb), dim3(256), lds_grad, st, reinterpret_cast<const T*>(x), x_ld,
                         kernel, bias, n, h, w, labels, weights, (int)npix, inv, grad_scale, reinterpret_cast<T*>(logits_out),
                         reinterpret_cast<T*>(dlogits), part, cm, (int)npix_metrics, ignore_mask, pred_out, bad_labels);
    else
      hipLaunchKernelGGL((head2x2_xent_kernel<T, 24, 16, false>), dim3(nb), dim3(256), lds_eval, st, reinterpret_cast<const T*>(x), x_ld,
                         kernel, bias, n, h, w, labels, weights, (int)npix, inv, grad_scale, reinterpret_cast<T*>(logits_out),
                         reinterpret_cast<T*>(dlogits), part, cm, (int)npix_metrics, ignore_mask, pred_out, bad_labels);
    SSMI_LAUNCH_CHECK();
    return SSMI_OK;
  });
  if (s) return s;
  hipLaunchKernelGGL(xent_final, dim3(1), dim3(256), 0, st, part, nb, 1.0 / (double)npix, loss_out);
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

// ------------------------------------------------------------------------------------------------
// The same loss pass for callers that cannot allocate (code under HIP-graph capture, tight serving loops): the per-block
// partial sums live in a file-scope device array instead of a caller-supplied workspace.  Every GPU the library runs on
// holds its own copy of the array for the life of the process; its address on a device is looked up once, with that
// device current, and kept.  In host code the array's NAME is only a handle for that lookup, never a device address.
// ------------------------------------------------------------------------------------------------
static __device__ __align__(256) double g_xent_scratch[kLossBlocks];      // 16 KiB per GPU
static constexpr int kMaxDevices = 64;
static double* g_xent_scratch_ptr[kMaxDevices] = {};
static std::mutex g_xent_scratch_mu;

// Address of this device's copy of g_xent_scratch (nullptr + error text on failure).  No allocation: the array is part
// of the code object.
static double* xent_scratch_for_current_device() {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) {
    set_error("softmax_xent_static: cannot identify the current device");
    return nullptr;
  }
  std::lock_guard<std::mutex> lk(g_xent_scratch_mu);
  if (g_xent_scratch_ptr[dev] == nullptr) {
    void* p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_xent_scratch));
    if (e != hipSuccess || p == nullptr) {
      set_error("softmax_xent_static: hipGetSymbolAddress: %s", hipGetErrorString(e));
      return nullptr;
    }
    g_xent_scratch_ptr[dev] = reinterpret_cast<double*>(p);
  }
  return g_xent_scratch_ptr[dev];
}

}  // extern "C"
namespace ssmi {
bool resolve_static_scratch() { return xent_scratch_for_current_device() != nullptr; }
}  // namespace ssmi
extern "C" {

int64_t ssmi_softmax_xent_static_max_pixels(void) { return (int64_t)1 << 28; }

ssmi_status ssmi_softmax_xent_fwd_bwd_static(const float* logits, const int32_t* labels, const float* weights, int64_t npix,
                                             int32_t num_classes, float grad_scale, float* dlogits, float* loss_out,
                                             int64_t* cm, int64_t npix_metrics, uint64_t ignore_mask, int32_t* pred_out,
                                             int32_t* bad_labels, void* stream) {
  const int64_t span = npix_metrics > npix ? npix_metrics : npix;
  SSMI_REQUIRE(span <= ssmi_softmax_xent_static_max_pixels(), SSMI_E_UNSUPPORTED,
               "softmax_xent_static: %lld pixels exceed the maximum of %lld for the allocation-free entry point",
               (long long)span, (long long)ssmi_softmax_xent_static_max_pixels());
  // The address lookup is not stream-ordered work and must not run while `stream` is recording: a capture begun through
  // ssmi_graph_begin_capture has resolved it already; under any other capture the first call on a device is refused.
  int dev = -1;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (stream != nullptr && hipStreamIsCapturing(as_stream(stream), &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
    bool known = false;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < kMaxDevices) {
      std::lock_guard<std::mutex> lk(g_xent_scratch_mu);
      known = g_xent_scratch_ptr[dev] != nullptr;
    }
    SSMI_REQUIRE(known, SSMI_E_UNSUPPORTED,
                 "softmax_xent_static: the first call on a device must happen outside stream capture (or begin the capture "
                 "with ssmi_graph_begin_capture, which prepares the device)");
  }
  double* scratch = xent_scratch_for_current_device();
  if (scratch == nullptr) return SSMI_E_HIP;
  // every partial-sum slot the final kernel reads is overwritten by this call's blocks first: nothing to clear
  return ssmi_softmax_xent_fwd_bwd(logits, labels, weights, npix, num_classes, grad_scale, dlogits, loss_out, cm, npix_metrics,
                                   ignore_mask, pred_out, bad_labels, scratch, (size_t)kLossBlocks * sizeof(double), stream);
}

size_t ssmi_mt_consistency_workspace_bytes(int64_t npix) {
  (void)npix;
  return (size_t)kLossBlocks * sizeof(double);
}

ssmi_status ssmi_mt_consistency_fwd_bwd(const float* logits, const float* teacher_logits, int64_t npix, int32_t num_classes,
                                        const float* coef_dev, float grad_scale, int64_t npix_accumulate, float* dlogits,
                                        float* loss_io, int32_t accumulate_loss, float* cost_out, void* ws, size_t ws_bytes,
                                        void* stream) {
  SSMI_REQUIRE(logits && teacher_logits && coef_dev && npix > 0, SSMI_E_INVALID_ARG, "mt_consistency: bad argument");
  SSMI_REQUIRE(num_classes >= 2 && num_classes <= 64, SSMI_E_UNSUPPORTED, "mt_consistency: num_classes %d outside [2,64]", num_classes);
  SSMI_REQUIRE(npix_accumulate >= 0 && npix_accumulate <= npix, SSMI_E_INVALID_ARG, "mt_consistency: npix_accumulate out of range");
  SSMI_REQUIRE(ws && ws_bytes >= ssmi_mt_consistency_workspace_bytes(npix), SSMI_E_WORKSPACE, "mt_consistency: workspace too small");
  hipStream_t st = as_stream(stream);
  int64_t nb64 = ceil_div64(npix, 256);
  if (nb64 > kLossBlocks) nb64 = kLossBlocks;
  const int nb = (int)nb64;
  double* part = reinterpret_cast<double*>(ws);
  const double inv_count = 1.0 / ((double)npix * (double)num_classes);
  const float kscale = (float)((double)grad_scale * inv_count);
  const bool aligned = ((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(teacher_logits) |
                         reinterpret_cast<uintptr_t>(dlogits)) & 15) == 0;
  if (num_classes == 24 && aligned) {
    hipLaunchKernelGGL((mt_consistency_kernel<24>), dim3(nb), dim3(256), 0, st, logits, teacher_logits, npix, num_classes,
                       coef_dev, kscale, npix_accumulate, dlogits, part);
  } else {
    hipLaunchKernelGGL((mt_consistency_kernel<0>), dim3(nb), dim3(256), 0, st, logits, teacher_logits, npix, num_classes,
                       coef_dev, kscale, npix_accumulate, dlogits, part);
  }
  SSMI_LAUNCH_CHECK();
  hipLaunchKernelGGL(mt_consistency_final, dim3(1), dim3(256), 0, st, part, nb, inv_count, coef_dev, accumulate_loss, loss_io,
                     cost_out);
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

size_t ssmi_superpixel_workspace_bytes(int64_t npix, int32_t num_classes) {
  return (size_t)3 * (size_t)npix * (size_t)num_classes * sizeof(float) + (size_t)kLossBlocks * sizeof(double) + 64;
}

ssmi_status ssmi_superpixel_cost_fwd_bwd(const float* logits, const int32_t* mask, int32_t n, int32_t h, int32_t w,
                                         int32_t num_classes, const float* coef_dev, float grad_scale, float epsilon,
                                         float* dlogits, int32_t accumulate_grad, float* loss_io, int32_t accumulate_loss,
                                         float* cost_out, void* ws, size_t ws_bytes, void* stream) {
  SSMI_REQUIRE(logits && mask && coef_dev && n > 0 && h > 0 && w > 0, SSMI_E_INVALID_ARG, "superpixel_cost: bad argument");
  SSMI_REQUIRE(num_classes >= 2 && num_classes <= 64, SSMI_E_UNSUPPORTED, "superpixel_cost: num_classes %d outside [2,64]", num_classes);
  const int64_t npix = (int64_t)n * h * w;
  SSMI_REQUIRE(npix * num_classes < 0x7FFFFFFFLL, SSMI_E_UNSUPPORTED, "superpixel_cost: more than 2^31 elements");
  SSMI_REQUIRE(ws && ws_bytes >= ssmi_superpixel_workspace_bytes(npix, num_classes), SSMI_E_WORKSPACE, "superpixel_cost: workspace too small");
  hipStream_t st = as_stream(stream);
  const size_t plane = (size_t)npix * num_classes;
  float* S = reinterpret_cast<float*>(ws);
  float* dGx = S + plane;
  float* dGy = dGx + plane;
  double* part = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(dGy + plane) + 63) & ~(uintptr_t)63);
  int64_t nb64 = ceil_div64(npix, 256);
  if (nb64 > kLossBlocks) nb64 = kLossBlocks;
  const int nb = (int)nb64;
  hipLaunchKernelGGL(sp_softmax_kernel, dim3(nb), dim3(256), 0, st, logits, (unsigned)npix, num_classes, S);
  SSMI_LAUNCH_CHECK();
  const float kscale = (float)((double)grad_scale / (double)npix);
  hipLaunchKernelGGL(sp_grad_kernel, dim3(nb), dim3(256), 0, st, S, mask, h, w, (unsigned)npix, num_classes, coef_dev, kscale,
                     epsilon, dlogits ? dGx : nullptr, dGy, part);
  SSMI_LAUNCH_CHECK();
  if (dlogits) {
    hipLaunchKernelGGL(sp_apply_kernel, dim3(nb), dim3(256), 0, st, S, dGx, dGy, h, w, (unsigned)npix, num_classes,
                       accumulate_grad, dlogits);
    SSMI_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(mt_consistency_final, dim3(1), dim3(256), 0, st, part, nb, 1.0 / (double)npix, coef_dev, accumulate_loss,
                     loss_io, cost_out);
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

ssmi_status ssmi_classification_xent_fwd_bwd(const float* logits, const float* y_true, const float* weights, int32_t n,
                                             int32_t num_classes, float epsilon, float grad_scale, float* dlogits,
                                             float* loss_out, void* stream) {
  SSMI_REQUIRE(logits && y_true && weights && loss_out && n > 0 && num_classes > 0, SSMI_E_INVALID_ARG,
               "classification_xent: bad argument");
  hipLaunchKernelGGL(classification_xent_kernel, dim3(1), dim3(256), 0, as_stream(stream), logits, y_true, weights, n,
                     num_classes, epsilon, grad_scale, dlogits, loss_out);
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

ssmi_status ssmi_ema_update(float* teacher, const float* student, double a, int64_t count, void* stream) {
  SSMI_REQUIRE(teacher && student && count >= 0, SSMI_E_INVALID_ARG, "ema_update: bad argument");
  if (count == 0) return SSMI_OK;
  int64_t nb = ceil_div64(count, 256);
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(ema_kernel, dim3((int)nb), dim3(256), 0, as_stream(stream), teacher, student, (float)a,
                     (float)(1.0 - a), count);
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

ssmi_status ssmi_metrics_from_cm(const int64_t* cm, int32_t n, uint64_t ignore_mask, double* out6) {
  SSMI_REQUIRE(cm && out6 && n >= 1 && n <= 64, SSMI_E_INVALID_ARG, "metrics_from_cm: bad argument");
  double total = 0, diag = 0;
  double mpca_tf = 0, iou_sum = 0, iou_all = 0, mpca_s = 0, miou_s = 0;
  int present = 0, kept = 0;
  for (int i = 0; i < n; ++i) {
    double row = 0, colsum = 0;
    for (int j = 0; j < n; ++j) { row += (double)cm[i * n + j]; colsum += (double)cm[j * n + i]; }
    const double tp = (double)cm[i * n + i];
    total += row;
    diag += tp;
    const double acc_i = row > 0 ? tp / row : 0.0;
    mpca_tf += acc_i;
    const double den = row + colsum - tp;
    const double iou = tp / (den > 0 ? den : 1.0);
    if (den > 0) { iou_sum += iou; ++present; }
    iou_all += iou;
    if (!((ignore_mask >> i) & 1ull)) {
      // src/scripts/calculate_metrics_from_cfms.py:24,45 round the numerator to float32, then divide in float64
      mpca_s += (double)(float)tp / (row > 0 ? row : 1.0);
      miou_s += (double)(float)tp / (den > 0 ? den : 1.0);
      ++kept;
    }
  }
  out6[0] = total > 0 ? diag / total : 0.0;
  out6[1] = mpca_tf / n;
  out6[2] = present ? iou_sum / present : 0.0;
  out6[3] = iou_all / n;
  out6[4] = kept ? mpca_s / kept : 0.0;
  out6[5] = kept ? miou_s / kept : 0.0;
  return SSMI_OK;
}

}  // extern "C"

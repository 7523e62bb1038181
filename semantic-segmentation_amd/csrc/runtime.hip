// Error reporting, optimizers over the flat parameter arena, dropout tables and HIP-graph capture.
#include <stdarg.h>

#include <vector>

#include "common.h"
#include <algorithm>

namespace ssmi {

static thread_local char g_err[512] = "";
static thread_local char g_kernel[128] = "";

void set_last_kernel(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
  va_end(ap);
}

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ------------------------------------------------------------------------------------------------
// Adam (extended_optimizers.py:117-162).  adam_prepare (one thread) turns the step counter into the bias-corrected
// learning rate and advances the counter; adam_apply streams the arena.
// ------------------------------------------------------------------------------------------------
__global__ void adam_prepare(int64_t* step, float lr, float beta_1, float beta_2, float decay, float* lr_t_out) {
  const int64_t it = *step;
  double l = (double)lr;
  if (decay > 0.f) l *= 1.0 / (1.0 + (double)decay * (double)it);
  const double t = (double)(it + 1);
  const double lr_t = l * (sqrt(1.0 - pow((double)beta_2, t)) / (1.0 - pow((double)beta_1, t)));
  *lr_t_out = (float)lr_t;
  *step = it + 1;
}

__global__ __launch_bounds__(256) void adam_apply(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                  float* __restrict__ v, float* __restrict__ vhat, int64_t n,
                                                  const float* __restrict__ lr_t_ptr, float b1, float b2, float eps,
                                                  float gscale, const float* __restrict__ lr_scale) {
  const float lr_t0 = *lr_t_ptr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = g[i] * gscale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    float den = vi;
    if (vhat) {
      den = fmaxf(vhat[i], vi);
      vhat[i] = den;
    }
    const float lr_t = lr_scale ? lr_t0 * lr_scale[i] : lr_t0;
    p[i] = p[i] - lr_t * mi / (sqrtf(den) + eps);
    m[i] = mi;
    v[i] = vi;
  }
}

__global__ void sgd_prepare(int64_t* step, float lr, float decay, float* lr_out) {
  const int64_t it = *step;
  double l = (double)lr;
  if (decay > 0.f) l *= 1.0 / (1.0 + (double)decay * (double)it);
  *lr_out = (float)l;
  *step = it + 1;
}

__global__ __launch_bounds__(256) void sgd_apply(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ mom,
                                                 int64_t n, const float* __restrict__ lr_ptr, float momentum, int nesterov,
                                                 float gscale, const float* __restrict__ lr_scale) {
  const float lr0 = *lr_ptr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float lr = lr_scale ? lr0 * lr_scale[i] : lr0;
    const float gi = g[i] * gscale;
    const float vel = momentum * mom[i] - lr * gi;
    mom[i] = vel;
    p[i] = nesterov ? p[i] + momentum * vel - lr * gi : p[i] + vel;
  }
}

// ------------------------------------------------------------------------------------------------
// The same updates with every hyper-parameter read from a small DEVICE block, so that a step captured once into a HIP
// graph follows later changes of the learning rate / betas / momentum (per-step schedules, reduce-on-plateau): the host
// rewrites the block between graph launches.  Adam block: {lr, beta_1, beta_2, epsilon, decay, grad_scale, -, -};
// SGD block: {lr, momentum, decay, nesterov (0 or 1), grad_scale, -, -, -}.
// ------------------------------------------------------------------------------------------------
__global__ void adam_prepare_hp(int64_t* step, const float* __restrict__ hp, float* lr_t_out) {
  const int64_t it = *step;
  double l = (double)hp[0];
  if (hp[4] > 0.f) l *= 1.0 / (1.0 + (double)hp[4] * (double)it);
  const double t = (double)(it + 1);
  *lr_t_out = (float)(l * (sqrt(1.0 - pow((double)hp[2], t)) / (1.0 - pow((double)hp[1], t))));
  *step = it + 1;
}

__global__ __launch_bounds__(256) void adam_apply_hp(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                     float* __restrict__ v, float* __restrict__ vhat, int64_t n,
                                                     const float* __restrict__ lr_t_ptr, const float* __restrict__ hp,
                                                     const float* __restrict__ lr_scale) {
  const float lr_t0 = *lr_t_ptr;
  const float b1 = hp[1], b2 = hp[2], eps = hp[3], gscale = hp[5];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = g[i] * gscale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    float den = vi;
    if (vhat) {
      den = fmaxf(vhat[i], vi);
      vhat[i] = den;
    }
    const float lr_t = lr_scale ? lr_t0 * lr_scale[i] : lr_t0;
    p[i] = p[i] - lr_t * mi / (sqrtf(den) + eps);
    m[i] = mi;
    v[i] = vi;
  }
}

__global__ void sgd_prepare_hp(int64_t* step, const float* __restrict__ hp, float* lr_out) {
  const int64_t it = *step;
  double l = (double)hp[0];
  if (hp[2] > 0.f) l *= 1.0 / (1.0 + (double)hp[2] * (double)it);
  *lr_out = (float)l;
  *step = it + 1;
}

__global__ __launch_bounds__(256) void sgd_apply_hp(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ mom,
                                                    int64_t n, const float* __restrict__ lr_ptr, const float* __restrict__ hp,
                                                    const float* __restrict__ lr_scale) {
  const float lr0 = *lr_ptr;
  const float momentum = hp[1], gscale = hp[4];
  const bool nesterov = hp[3] != 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float lr = lr_scale ? lr0 * lr_scale[i] : lr0;
    const float gi = g[i] * gscale;
    const float vel = momentum * mom[i] - lr * gi;
    mom[i] = vel;
    p[i] = nesterov ? p[i] + momentum * vel - lr * gi : p[i] + vel;
  }
}

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 -> SpatialDropout2D keep/scale table
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t (&k)[2]) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
  const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
  const uint32_t n0 = hi1 ^ c[1] ^ k[0], n1 = lo1, n2 = hi0 ^ c[3] ^ k[1], n3 = lo0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
  k[0] += 0x9E3779B9u;
  k[1] += 0xBB67AE85u;
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(float* __restrict__ mask, int64_t n, float rate, uint64_t seed,
                                                           const int64_t* __restrict__ step, uint32_t stream_id) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint64_t st = step ? (uint64_t)*step : 0ull;
  uint32_t c[4] = {(uint32_t)i, (uint32_t)((uint64_t)i >> 32), (uint32_t)st ^ (stream_id * 0x9E3779B1u), (uint32_t)(st >> 32) + stream_id};
  uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
  for (int r = 0; r < 10; ++r) philox_round(c, k);
  const float u = (float)(c[0] >> 8) * (1.0f / 16777216.0f);  // [0,1)
  mask[i] = (u >= rate) ? 1.f / (1.f - rate) : 0.f;
}

struct ProfileRecord {
  char name[128];
  hipEvent_t e0, e1;
  double bytes, flops;
  float ms;
};
static bool g_profiling = false;
static std::vector<ProfileRecord> g_records;
static hipStream_t g_profile_stream = nullptr;
static float g_bracket_overhead_ms = 0.f;

// Calibration launch for the event brackets: its own execution time is negligible, so the time between the two events
// around it is what the bracket itself adds to any kernel (event packets + the launch latency on either side).
__global__ void profile_null_kernel() {}

ProfileScope::ProfileScope(hipStream_t st, double bytes, double flops) : st_(st), bytes_(bytes), flops_(flops) {
  if (!g_profiling) return;
  if (hipEventCreate(&e0_) != hipSuccess || hipEventCreate(&e1_) != hipSuccess) { e0_ = e1_ = nullptr; return; }
  g_profile_stream = st_;
  if (hipEventRecord(e0_, st_) != hipSuccess) {   // no record for this launch rather than a bogus one
    (void)hipEventDestroy(e0_);
    (void)hipEventDestroy(e1_);
    e0_ = e1_ = nullptr;
  }
}

ProfileScope::~ProfileScope() {
  if (!e0_ || !e1_) return;
  if (hipEventRecord(e1_, st_) != hipSuccess) {
    (void)hipEventDestroy(e0_);
    (void)hipEventDestroy(e1_);
    return;
  }
  ProfileRecord r;
  snprintf(r.name, sizeof(r.name), "%s", g_kernel);
  r.e0 = e0_; r.e1 = e1_; r.bytes = bytes_; r.flops = flops_; r.ms = -1.f;
  g_records.push_back(r);
}

}  // namespace ssmi

using namespace ssmi;

struct ssmi_graph {
  hipGraph_t graph;
  hipGraphExec_t exec;
};

extern "C" {

const char* ssmi_last_error(void) { return g_err; }
const char* ssmi_last_kernel(void) { return g_kernel; }

ssmi_status ssmi_profile_begin(void) {
  for (auto& r : g_records) {
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  g_records.clear();
  g_profiling = true;
  return SSMI_OK;
}

ssmi_status ssmi_profile_end(int32_t* count) {
  SSMI_REQUIRE(count != nullptr, SSMI_E_INVALID_ARG, "profile_end: count is NULL");
  g_profiling = false;
  g_bracket_overhead_ms = 0.f;
  if (!g_records.empty()) {
    // median of nine brackets around an empty kernel on the same stream (one untimed warm-up first)
    float t[9];
    hipEvent_t a, b;
    SSMI_HIP(hipEventCreate(&a));
    SSMI_HIP(hipEventCreate(&b));
    for (int i = -1; i < 9; ++i) {
      SSMI_HIP(hipEventRecord(a, g_profile_stream));
      hipLaunchKernelGGL(profile_null_kernel, dim3(1), dim3(64), 0, g_profile_stream);
      SSMI_HIP(hipEventRecord(b, g_profile_stream));
      SSMI_HIP(hipEventSynchronize(b));
      float ms = 0.f;
      SSMI_HIP(hipEventElapsedTime(&ms, a, b));
      if (i >= 0) t[i] = ms;
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    std::sort(t, t + 9);
    g_bracket_overhead_ms = t[4];
  }
  for (auto& r : g_records) {
    SSMI_HIP(hipEventSynchronize(r.e1));
    SSMI_HIP(hipEventElapsedTime(&r.ms, r.e0, r.e1));
  }
  *count = (int32_t)g_records.size();
  return SSMI_OK;
}

ssmi_status ssmi_profile_get(int32_t index, char* name, int32_t name_len, float* ms, double* bytes, double* flops) {
  SSMI_REQUIRE(index >= 0 && index < (int32_t)g_records.size() && name && name_len > 0 && ms && bytes && flops,
               SSMI_E_INVALID_ARG, "profile_get: bad argument");
  const ProfileRecord& r = g_records[index];
  snprintf(name, (size_t)name_len, "%s", r.name);
  *ms = r.ms; *bytes = r.bytes; *flops = r.flops;
  return SSMI_OK;
}
float ssmi_profile_bracket_overhead_ms(void) { return g_bracket_overhead_ms; }
int32_t ssmi_version(void) { return 100; }

}  // extern "C"

namespace ssmi {
__global__ void timestamp_kernel(int64_t* __restrict__ out) { *out = (int64_t)wall_clock64(); }
}  // namespace ssmi

extern "C" {

ssmi_status ssmi_device_timestamp(int64_t* out, void* stream) {
  SSMI_REQUIRE(out, SSMI_E_INVALID_ARG, "device_timestamp: NULL output");
  hipLaunchKernelGGL(ssmi::timestamp_kernel, dim3(1), dim3(1), 0, as_stream(stream), out);
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

int32_t ssmi_device_timestamp_khz(void) {
  int dev = 0, khz = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) return 0;
  return khz;
}

ssmi_status ssmi_adam_step(float* params, const float* grads, float* m, float* v, float* vhat, int64_t count,
                           int64_t* step, float lr, float beta_1, float beta_2, float epsilon, float decay,
                           float grad_scale, const float* lr_scale, void* stream) {
  SSMI_REQUIRE(params && grads && m && v && step && count > 0, SSMI_E_INVALID_ARG, "adam_step: bad argument");
  hipStream_t st = as_stream(stream);
  // the 8 scratch bytes behind the counter hold the bias-corrected rate of this step
  float* lr_t = reinterpret_cast<float*>(step + 1);
  hipLaunchKernelGGL(adam_prepare, dim3(1), dim3(1), 0, st, step, lr, beta_1, beta_2, decay, lr_t);
  SSMI_LAUNCH_CHECK();
  int64_t nb = ceil_div64(count, 256 * 4);
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(adam_apply, dim3((unsigned)nb), dim3(256), 0, st, params, grads, m, v, vhat, count, lr_t, beta_1, beta_2,
                     epsilon, grad_scale, lr_scale);
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

ssmi_status ssmi_sgd_step(float* params, const float* grads, float* moments, int64_t count, int64_t* step, float lr,
                          float momentum, float decay, int32_t nesterov, float grad_scale, const float* lr_scale,
                          void* stream) {
  SSMI_REQUIRE(params && grads && moments && step && count > 0, SSMI_E_INVALID_ARG, "sgd_step: bad argument");
  hipStream_t st = as_stream(stream);
  float* lr_eff = reinterpret_cast<float*>(step + 1);
  hipLaunchKernelGGL(sgd_prepare, dim3(1), dim3(1), 0, st, step, lr, decay, lr_eff);
  SSMI_LAUNCH_CHECK();
  int64_t nb = ceil_div64(count, 256 * 4);
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(sgd_apply, dim3((unsigned)nb), dim3(256), 0, st, params, grads, moments, count, lr_eff, momentum, nesterov,
                     grad_scale, lr_scale);
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

ssmi_status ssmi_adam_step_hp(float* params, const float* grads, float* m, float* v, float* vhat, int64_t count,
                              int64_t* step, const float* hyper, const float* lr_scale, void* stream) {
  SSMI_REQUIRE(params && grads && m && v && step && hyper && count > 0, SSMI_E_INVALID_ARG, "adam_step_hp: bad argument");
  hipStream_t st = as_stream(stream);
  float* lr_t = reinterpret_cast<float*>(step + 1);
  hipLaunchKernelGGL(adam_prepare_hp, dim3(1), dim3(1), 0, st, step, hyper, lr_t);
  SSMI_LAUNCH_CHECK();
  int64_t nb = ceil_div64(count, 256 * 4);
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(adam_apply_hp, dim3((unsigned)nb), dim3(256), 0, st, params, grads, m, v, vhat, count, lr_t, hyper, lr_scale);
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

ssmi_status ssmi_sgd_step_hp(float* params, const float* grads, float* moments, int64_t count, int64_t* step,
                             const float* hyper, const float* lr_scale, void* stream) {
  SSMI_REQUIRE(params && grads && moments && step && hyper && count > 0, SSMI_E_INVALID_ARG, "sgd_step_hp: bad argument");
  hipStream_t st = as_stream(stream);
  float* lr_eff = reinterpret_cast<float*>(step + 1);
  hipLaunchKernelGGL(sgd_prepare_hp, dim3(1), dim3(1), 0, st, step, hyper, lr_eff);
  SSMI_LAUNCH_CHECK();
  int64_t nb = ceil_div64(count, 256 * 4);
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(sgd_apply_hp, dim3((unsigned)nb), dim3(256), 0, st, params, grads, moments, count, lr_eff, hyper, lr_scale);
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

ssmi_status ssmi_dropout_mask(float* mask, int64_t count, float rate, uint64_t seed, const int64_t* step_counter,
                              uint32_t stream_id, void* stream) {
  SSMI_REQUIRE(mask && count > 0 && rate >= 0.f && rate < 1.f, SSMI_E_INVALID_ARG, "dropout_mask: bad argument");
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)ceil_div64(count, 256)), dim3(256), 0, as_stream(stream), mask, count,
                     rate, seed, step_counter, stream_id);
  SSMI_LAUNCH_CHECK();
  return SSMI_OK;
}

ssmi_status ssmi_graph_begin_capture(void* stream) {
  SSMI_REQUIRE(stream != nullptr, SSMI_E_INVALID_ARG, "graph capture needs a non-default stream");
  // symbol lookups are runtime calls that are not stream-ordered: do them before the stream starts recording
  if (!resolve_static_scratch()) return SSMI_E_HIP;
  SSMI_HIP(hipStreamBeginCapture(as_stream(stream), hipStreamCaptureModeThreadLocal));
  return SSMI_OK;
}

ssmi_status ssmi_graph_end_capture(void* stream, ssmi_graph** out) {
  SSMI_REQUIRE(stream != nullptr && out != nullptr, SSMI_E_INVALID_ARG, "graph_end_capture: bad argument");
  hipGraph_t g = nullptr;
  SSMI_HIP(hipStreamEndCapture(as_stream(stream), &g));
  hipGraphExec_t e = nullptr;
  hipError_t err = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
  if (err != hipSuccess) {
    (void)hipGraphDestroy(g);
    set_error("hipGraphInstantiate: %s", hipGetErrorString(err));
    return SSMI_E_HIP;
  }
  ssmi_graph* r = new ssmi_graph{g, e};
  *out = r;
  return SSMI_OK;
}

ssmi_status ssmi_graph_launch(ssmi_graph* g, void* stream) {
  SSMI_REQUIRE(g != nullptr, SSMI_E_INVALID_ARG, "graph_launch: NULL graph");
  SSMI_HIP(hipGraphLaunch(g->exec, as_stream(stream)));
  return SSMI_OK;
}

ssmi_status ssmi_graph_destroy(ssmi_graph* g) {
  if (!g) return SSMI_OK;
  const hipError_t e1 = hipGraphExecDestroy(g->exec);
  const hipError_t e2 = hipGraphDestroy(g->graph);
  delete g;
  if (e1 != hipSuccess || e2 != hipSuccess) {
    set_error("graph_destroy: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    return SSMI_E_HIP;
  }
  return SSMI_OK;
}

}  // extern "C"

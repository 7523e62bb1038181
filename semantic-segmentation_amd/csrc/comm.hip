// Collectives of the data-parallel step behind the C ABI (ssmi_comm_*): a thin layer over RCCL, so that a C / C++ host can
// run the exchange step of the hot path (gradient buckets summed over the ranks, int64 confusion matrices summed for sharded
// evaluation, parameters broadcast from rank 0) without Python.  RCCL is looked up at the first call (dlopen): the copy the
// process already holds is reused (PyTorch-ROCm brings its own), the library itself has no link-time dependency on it.
#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include "common.h"

namespace {

struct UniqueId {
  char internal[128];      // ncclUniqueId
};
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(void**, int, UniqueId, int);
typedef int (*CommDestroyFn)(void*);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*BroadcastFn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*ErrStrFn)(int);

struct Rccl {
  void* handle = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  AllReduceFn all_reduce = nullptr;
  BroadcastFn broadcast = nullptr;
  ErrStrFn err_str = nullptr;
};

constexpr int kNcclSum = 0, kNcclInt64 = 4, kNcclFloat32 = 7;      // rccl.h: ncclRedOp_t / ncclDataType_t

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names)            // a copy that is already in the process first
      if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char* n : names)
      if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!r.handle) return;
    r.get_unique_id = reinterpret_cast<GetUniqueIdFn>(dlsym(r.handle, "ncclGetUniqueId"));
    r.comm_init_rank = reinterpret_cast<CommInitRankFn>(dlsym(r.handle, "ncclCommInitRank"));
    r.comm_destroy = reinterpret_cast<CommDestroyFn>(dlsym(r.handle, "ncclCommDestroy"));
    r.all_reduce = reinterpret_cast<AllReduceFn>(dlsym(r.handle, "ncclAllReduce"));
    r.broadcast = reinterpret_cast<BroadcastFn>(dlsym(r.handle, "ncclBroadcast"));
    r.err_str = reinterpret_cast<ErrStrFn>(dlsym(r.handle, "ncclGetErrorString"));
  });
  if (!r.handle || !r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_reduce || !r.broadcast) return nullptr;
  return &r;
}

}  // namespace

struct ssmi_comm {
  void* nccl;
  int rank, nranks;
};

using namespace ssmi;

#define SSMI_RCCL(call, what)                                                                     \
  do {                                                                                            \
    const int rc__ = (call);                                                                      \
    if (rc__ != 0) {                                                                              \
      set_error("%s: RCCL error %d (%s)", what, rc__, R->err_str ? R->err_str(rc__) : "?");       \
      return SSMI_E_HIP;                                                                          \
    }                                                                                             \
  } while (0)

extern "C" {

ssmi_status ssmi_comm_unique_id(char id[128]) {
  Rccl* R = rccl();
  SSMI_REQUIRE(R, SSMI_E_UNSUPPORTED, "ssmi_comm: librccl.so could not be loaded");
  SSMI_REQUIRE(id, SSMI_E_INVALID_ARG, "ssmi_comm_unique_id: NULL");
  UniqueId u;
  SSMI_RCCL(R->get_unique_id(&u), "ssmi_comm_unique_id");
  memcpy(id, u.internal, 128);
  return SSMI_OK;
}

ssmi_status ssmi_comm_init(const char id[128], int32_t rank, int32_t nranks, ssmi_comm** out) {
  Rccl* R = rccl();
  SSMI_REQUIRE(R, SSMI_E_UNSUPPORTED, "ssmi_comm: librccl.so could not be loaded");
  SSMI_REQUIRE(id && out && nranks > 0 && rank >= 0 && rank < nranks, SSMI_E_INVALID_ARG, "ssmi_comm_init: bad argument");
  UniqueId u;
  memcpy(u.internal, id, 128);
  void* c = nullptr;
  SSMI_RCCL(R->comm_init_rank(&c, nranks, u, rank), "ssmi_comm_init");
  *out = new ssmi_comm{c, rank, nranks};
  return SSMI_OK;
}

ssmi_status ssmi_comm_destroy(ssmi_comm* c) {
  if (!c) return SSMI_OK;
  Rccl* R = rccl();
  if (R && c->nccl) SSMI_RCCL(R->comm_destroy(c->nccl), "ssmi_comm_destroy");
  delete c;
  return SSMI_OK;
}

int32_t ssmi_comm_size(const ssmi_comm* c) { return c ? c->nranks : 0; }
int32_t ssmi_comm_rank(const ssmi_comm* c) { return c ? c->rank : -1; }

ssmi_status ssmi_comm_allreduce_sum_f32(ssmi_comm* c, float* buf, int64_t count, void* stream) {
  Rccl* R = rccl();
  SSMI_REQUIRE(R && c && buf && count > 0, SSMI_E_INVALID_ARG, "ssmi_comm_allreduce_sum_f32: bad argument");
  SSMI_RCCL(R->all_reduce(buf, buf, (size_t)count, kNcclFloat32, kNcclSum, c->nccl, as_stream(stream)), "ssmi_comm_allreduce_sum_f32");
  return SSMI_OK;
}

ssmi_status ssmi_comm_allreduce_sum_i64(ssmi_comm* c, int64_t* buf, int64_t count, void* stream) {
  Rccl* R = rccl();
  SSMI_REQUIRE(R && c && buf && count > 0, SSMI_E_INVALID_ARG, "ssmi_comm_allreduce_sum_i64: bad argument");
  SSMI_RCCL(R->all_reduce(buf, buf, (size_t)count, kNcclInt64, kNcclSum, c->nccl, as_stream(stream)), "ssmi_comm_allreduce_sum_i64");
  return SSMI_OK;
}

ssmi_status ssmi_comm_broadcast_f32(ssmi_comm* c, float* buf, int64_t count, int32_t root, void* stream) {
  Rccl* R = rccl();
  SSMI_REQUIRE(R && c && buf && count > 0 && root >= 0 && root < c->nranks, SSMI_E_INVALID_ARG,
               "ssmi_comm_broadcast_f32: bad argument");
  SSMI_RCCL(R->broadcast(buf, buf, (size_t)count, kNcclFloat32, root, c->nccl, as_stream(stream)), "ssmi_comm_broadcast_f32");
  return SSMI_OK;
}

}  // extern "C"

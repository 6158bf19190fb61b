// The exchange step for hosts that do not bring a collective library of their own: a process-wide RCCL communicator
// (one process per GPU over xGMI) behind three calls.  It replaces the reference's queue / condition-variable
// hand-off between replica threads (modeling/sync_batchnorm/comm.py:18-129) and DataParallel's reduce-add
// (train.py:106-109).  librccl is opened on first use, so the kernels of this library load without it.
#include <dlfcn.h>
#include <string.h>

#include "common.h"

namespace segk {

typedef int (*get_unique_id_fn)(void*);
typedef int (*comm_init_rank_fn)(void**, int, segk_comm_id, int);
typedef int (*all_reduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*comm_destroy_fn)(void*);
typedef const char* (*get_error_string_fn)(int);

static struct {
  void* lib;
  get_unique_id_fn get_unique_id;
  comm_init_rank_fn comm_init_rank;
  all_reduce_fn all_reduce;
  comm_destroy_fn comm_destroy;
  get_error_string_fn error_string;
  void* comm;
  int world;
} g_rccl = {};

// RCCL's data-type and reduction tags (rccl.h: ncclInt64 = 4, ncclFloat32 = 7, ncclFloat64 = 8, ncclSum = 0)
constexpr int RCCL_INT64 = 4, RCCL_FLOAT32 = 7, RCCL_FLOAT64 = 8, RCCL_SUM = 0;

static int load_rccl() {
  if (g_rccl.lib) return SEGK_OK;
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) {
    set_error("comm: cannot open librccl: %s", dlerror());
    return SEGK_RCCL_ERROR;
  }
  g_rccl.get_unique_id = (get_unique_id_fn)dlsym(h, "ncclGetUniqueId");
  g_rccl.comm_init_rank = (comm_init_rank_fn)dlsym(h, "ncclCommInitRank");
  g_rccl.all_reduce = (all_reduce_fn)dlsym(h, "ncclAllReduce");
  g_rccl.comm_destroy = (comm_destroy_fn)dlsym(h, "ncclCommDestroy");
  g_rccl.error_string = (get_error_string_fn)dlsym(h, "ncclGetErrorString");
  if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.all_reduce || !g_rccl.comm_destroy) {
    set_error("comm: librccl lacks an expected symbol");
    dlclose(h);
    return SEGK_RCCL_ERROR;
  }
  g_rccl.lib = h;
  return SEGK_OK;
}

static int rccl_fail(const char* what, int code) {
  set_error("comm: %s failed: %s", what, g_rccl.error_string ? g_rccl.error_string(code) : "RCCL error");
  return SEGK_RCCL_ERROR;
}

}  // namespace segk

using namespace segk;

extern "C" int segk_comm_unique_id(segk_comm_id* id) {
  SEGK_REQUIRE(id, "comm_unique_id: null pointer");
  int rc = load_rccl();
  if (rc != SEGK_OK) return rc;
  const int e = g_rccl.get_unique_id(id);
  return e == 0 ? SEGK_OK : rccl_fail("ncclGetUniqueId", e);
}

extern "C" int segk_comm_init(const segk_comm_id* id, int rank, int world) {
  SEGK_REQUIRE(id && world >= 1 && rank >= 0 && rank < world, "comm_init: bad arguments rank=%d world=%d", rank, world);
  SEGK_REQUIRE(!g_rccl.comm, "comm_init: a communicator already exists (one per process)");
  int rc = load_rccl();
  if (rc != SEGK_OK) return rc;
  const int e = g_rccl.comm_init_rank(&g_rccl.comm, world, *id, rank);
  if (e != 0) {
    g_rccl.comm = nullptr;
    return rccl_fail("ncclCommInitRank", e);
  }
  g_rccl.world = world;
  return SEGK_OK;
}

extern "C" int segk_comm_destroy(void) {
  if (!g_rccl.comm) return SEGK_OK;
  const int e = g_rccl.comm_destroy(g_rccl.comm);
  g_rccl.comm = nullptr;
  g_rccl.world = 0;
  return e == 0 ? SEGK_OK : rccl_fail("ncclCommDestroy", e);
}

static int all_reduce_sum(void* buf, long long count, int tag, segk_stream_t stream) {
  SEGK_REQUIRE(buf && count > 0, "allreduce: null pointer or count=%lld", count);
  SEGK_REQUIRE(g_rccl.comm, "allreduce: segk_comm_init has not been called");
  const int e = g_rccl.all_reduce(buf, buf, (size_t)count, tag, RCCL_SUM, g_rccl.comm, (hipStream_t)stream);
  return e == 0 ? SEGK_OK : rccl_fail("ncclAllReduce", e);
}

extern "C" int segk_allreduce_f32(float* buf, long long count, segk_stream_t stream) {
  return all_reduce_sum(buf, count, RCCL_FLOAT32, stream);
}
extern "C" int segk_allreduce_f64(double* buf, long long count, segk_stream_t stream) {
  return all_reduce_sum(buf, count, RCCL_FLOAT64, stream);
}
extern "C" int segk_allreduce_i64(int64_t* buf, long long count, segk_stream_t stream) {
  return all_reduce_sum(buf, count, RCCL_INT64, stream);
}

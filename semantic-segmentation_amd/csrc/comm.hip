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

// ------------------------------------------------------------------------------------------------
// One-shot all-reduce through peer-mapped mailboxes, for the small vectors a synchronised batch norm exchanges
// ([2C+1] fp64, at most 32 KiB) where a library collective is pure latency: 226 of them per training step.
// Every rank owns one allocation (fine-grained device memory, shared with the other processes by IPC handle):
//   flags[2][MAX_WORLD]   the sequence number of the last vector rank q has delivered into slot [phase][q]
//   counter, status       this rank's own sequence number; non-zero status = a wait ran out of time
//   slots[2][world][MBX_MAX_DOUBLES]
// One exchange = one single-workgroup kernel per rank: store the vector into slot [phase][rank] of EVERY rank's
// mailbox (its own included), release, raise flag [phase][rank] there to the exchange's sequence number; wait until
// the own mailbox holds that number from every rank; add the slots in rank order (every rank computes the same
// bits).  phase = sequence number & 1: a rank can be one exchange ahead of a peer, never two - it needs the peer's
// flag of exchange k + 1 to finish k + 1, and the peer raises that only after it has read exchange k - so two slots
// per rank suffice.  The sequence number lives in device memory and advances inside the kernel, so a launch
// recorded in a HIP graph stays correct when it is replayed.  Longer vectors go through in chunks, one exchange
// each, inside the same launch.  Every wait is bounded (default 60 s); when it runs out the kernel records the
// sequence number in `status` and ends, and segk_mailbox_status() reports it.
// ------------------------------------------------------------------------------------------------
namespace segk {

constexpr int MBX_MAX_WORLD = 16;
constexpr int MBX_HEADER_BYTES = 512;
constexpr int MBX_THREADS = 512;

struct MailboxPeers {
  char* base[MBX_MAX_WORLD];
  int rank, world;
  unsigned long long timeout_ticks;   // of the 100 MHz constant clock
};

static struct {
  char* own;            // this rank's allocation
  void* opened[MBX_MAX_WORLD];
  MailboxPeers peers;
  bool connected;
} g_mbx = {};

static size_t mailbox_bytes(int world) {
  return (size_t)MBX_HEADER_BYTES + (size_t)2 * world * SEGK_MAILBOX_MAX_DOUBLES * sizeof(double);
}

__device__ __forceinline__ unsigned long long* mbx_flag(char* base, int phase, int q) {
  return reinterpret_cast<unsigned long long*>(base) + phase * MBX_MAX_WORLD + q;
}
__device__ __forceinline__ double* mbx_slot(char* base, int world, int phase, int q) {
  return reinterpret_cast<double*>(base + MBX_HEADER_BYTES) + ((size_t)phase * world + q) * SEGK_MAILBOX_MAX_DOUBLES;
}

template <typename T>
__global__ __launch_bounds__(MBX_THREADS) void mailbox_allreduce_kernel(const MailboxPeers a, T* buf, long long n) {
  __shared__ unsigned long long s_seq0;
  __shared__ int s_fail;
  const int tid = threadIdx.x;
  char* const own = a.base[a.rank];
  unsigned long long* const counter = reinterpret_cast<unsigned long long*>(own) + 2 * MBX_MAX_WORLD;
  unsigned long long* const status = counter + 1;
  if (tid == 0) {
    s_seq0 = *counter;
    s_fail = 0;
  }
  __syncthreads();
  const unsigned long long seq0 = s_seq0;
  const long long chunks = (n + SEGK_MAILBOX_MAX_DOUBLES - 1) / SEGK_MAILBOX_MAX_DOUBLES;
  long long done = 0;
  for (long long c = 0; c < chunks; ++c) {
    const unsigned long long seq = seq0 + 1 + (unsigned long long)c;
    const int phase = (int)(seq & 1ull);
    T* const part = buf + c * SEGK_MAILBOX_MAX_DOUBLES;
    const int cn = (int)((n - c * SEGK_MAILBOX_MAX_DOUBLES) < SEGK_MAILBOX_MAX_DOUBLES ? (n - c * SEGK_MAILBOX_MAX_DOUBLES)
                                                                                   : SEGK_MAILBOX_MAX_DOUBLES);
    // deliver: this rank's values into its slot of every mailbox, write-through at system scope
    for (int p = 0; p < a.world; ++p) {
      double* dst = mbx_slot(a.base[p], a.world, phase, a.rank);
      for (int i = tid; i < cn; i += MBX_THREADS)
        __hip_atomic_store(dst + i, (double)part[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // (the stores above are write-through; every storing wavefront waits for its own, the workgroup meets, and
    // only the lanes that raise a flag pay for a release - a system-scope fence in all 512 threads costs
    // microseconds)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < a.world)
      __hip_atomic_store(mbx_flag(a.base[tid], phase, a.rank), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // wait for every rank's delivery into the own mailbox: relaxed polls, one acquire once the flag is up (bounded)
    if (tid < a.world) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      unsigned long long* f = mbx_flag(own, phase, tid);
      unsigned polls = 0;
      while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
        __builtin_amdgcn_s_sleep(1);
        if ((++polls & 63u) == 0u && __builtin_amdgcn_s_memrealtime() - t0 > a.timeout_ticks) {
          atomicExch(&s_fail, 1);
          break;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
    __syncthreads();
    if (s_fail) {
      if (tid == 0) __hip_atomic_store(status, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      break;
    }
    // the sum, in rank order on every rank
    for (int i = tid; i < cn; i += MBX_THREADS) {
      double s = 0.0;
      for (int q = 0; q < a.world; ++q)
        s += __hip_atomic_load(mbx_slot(own, a.world, phase, q) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      part[i] = (T)s;
    }
    done = c + 1;
    __syncthreads();
  }
  // (also after a failure: the exchanges that did complete stay counted, so that the ranks' numbers keep agreeing
  // for whoever inspects them)
  if (tid == 0) *counter = seq0 + (unsigned long long)done;
}

static int mailbox_launch_check(const void* buf, long long n) {
  SEGK_REQUIRE(g_mbx.connected, "mailbox: segk_mailbox_connect has not been called");
  SEGK_REQUIRE(buf && n > 0, "mailbox: null pointer or count=%lld", n);
  return SEGK_OK;
}

}  // namespace segk

extern "C" int segk_mailbox_create(int world, void* handle_out) {
  SEGK_REQUIRE(handle_out && world >= 1 && world <= MBX_MAX_WORLD, "mailbox_create: world=%d (1..%d)", world, MBX_MAX_WORLD);
  SEGK_REQUIRE(!g_mbx.own, "mailbox_create: this process already has a mailbox");
  static_assert(sizeof(hipIpcMemHandle_t) <= SEGK_MAILBOX_HANDLE_BYTES, "IPC handle larger than the ABI's buffer");
  static_assert((2 * MBX_MAX_WORLD + 2) * sizeof(unsigned long long) <= MBX_HEADER_BYTES, "mailbox header too small");
  void* p = nullptr;
  const size_t bytes = mailbox_bytes(world);
  hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) {
    set_error("mailbox_create: hipExtMallocWithFlags(%zu, fine-grained) failed: %s", bytes, hipGetErrorString(e));
    return SEGK_HIP_ERROR;
  }
  e = hipMemset(p, 0, bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  hipIpcMemHandle_t h;
  if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p);
  if (e != hipSuccess) {
    set_error("mailbox_create: %s", hipGetErrorString(e));
    (void)hipFree(p);
    return SEGK_HIP_ERROR;
  }
  memset(handle_out, 0, SEGK_MAILBOX_HANDLE_BYTES);
  memcpy(handle_out, &h, sizeof(h));
  g_mbx.own = (char*)p;
  g_mbx.peers.world = world;
  return SEGK_OK;
}

extern "C" int segk_mailbox_connect(int rank, int world, const void* handles, int timeout_ms) {
  SEGK_REQUIRE(g_mbx.own && !g_mbx.connected, "mailbox_connect: call segk_mailbox_create first (once)");
  SEGK_REQUIRE(handles && world == g_mbx.peers.world && rank >= 0 && rank < world, "mailbox_connect: rank=%d world=%d", rank, world);
  for (int q = 0; q < world; ++q) {
    if (q == rank) {
      g_mbx.peers.base[q] = g_mbx.own;
      continue;
    }
    hipIpcMemHandle_t h;
    memcpy(&h, (const char*)handles + (size_t)q * SEGK_MAILBOX_HANDLE_BYTES, sizeof(h));
    void* p = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      set_error("mailbox_connect: hipIpcOpenMemHandle of rank %d's mailbox failed: %s", q, hipGetErrorString(e));
      for (int r = 0; r < q; ++r)
        if (g_mbx.opened[r]) {
          (void)hipIpcCloseMemHandle(g_mbx.opened[r]);
          g_mbx.opened[r] = nullptr;
        }
      return SEGK_HIP_ERROR;
    }
    g_mbx.opened[q] = p;
    g_mbx.peers.base[q] = (char*)p;
  }
  g_mbx.peers.rank = rank;
  g_mbx.peers.timeout_ticks = (unsigned long long)(timeout_ms > 0 ? timeout_ms : 60000) * 100000ull;
  g_mbx.connected = true;
  return SEGK_OK;
}

extern "C" int segk_mailbox_set_timeout(int timeout_ms) {
  SEGK_REQUIRE(g_mbx.connected, "mailbox_set_timeout: segk_mailbox_connect has not been called");
  g_mbx.peers.timeout_ticks = (unsigned long long)(timeout_ms > 0 ? timeout_ms : 60000) * 100000ull;
  return SEGK_OK;
}

extern "C" int segk_mailbox_allreduce_f64(double* buf, long long n, segk_stream_t stream) {
  int rc = mailbox_launch_check(buf, n);
  if (rc != SEGK_OK) return rc;
  mailbox_allreduce_kernel<double><<<1, MBX_THREADS, 0, (hipStream_t)stream>>>(g_mbx.peers, buf, n);
  SEGK_CHECK_LAUNCH("mailbox_allreduce_f64");
  return SEGK_OK;
}

extern "C" int segk_mailbox_allreduce_f32(float* buf, long long n, segk_stream_t stream) {
  int rc = mailbox_launch_check(buf, n);
  if (rc != SEGK_OK) return rc;
  mailbox_allreduce_kernel<float><<<1, MBX_THREADS, 0, (hipStream_t)stream>>>(g_mbx.peers, buf, n);
  SEGK_CHECK_LAUNCH("mailbox_allreduce_f32");
  return SEGK_OK;
}

extern "C" long long segk_mailbox_status(void) {
  if (!g_mbx.own) return 0;
  unsigned long long st[2] = {0, 0};   // counter, status
  if (hipMemcpy(st, g_mbx.own + 2 * MBX_MAX_WORLD * sizeof(unsigned long long), sizeof(st), hipMemcpyDeviceToHost) != hipSuccess) {
    set_error("mailbox_status: cannot read the mailbox");
    return SEGK_HIP_ERROR;
  }
  if (st[1] != 0) {
    set_error("mailbox: exchange %llu timed out waiting for a rank (a peer is gone, or ranks issued different exchanges)", st[1]);
    return SEGK_RCCL_ERROR;
  }
  return (long long)st[0];
}

extern "C" int segk_mailbox_destroy(void) {
  if (!g_mbx.own) return SEGK_OK;
  (void)hipDeviceSynchronize();
  for (int q = 0; q < MBX_MAX_WORLD; ++q)
    if (g_mbx.opened[q]) (void)hipIpcCloseMemHandle(g_mbx.opened[q]);
  (void)hipFree(g_mbx.own);
  g_mbx = {};
  return SEGK_OK;
}

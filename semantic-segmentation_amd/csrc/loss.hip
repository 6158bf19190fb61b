// Cross-entropy with ignore index (fused log-softmax + NLL, forward and backward), class argmax and the
// confusion-matrix histogram.  HBM-bound: logits are NCHW fp32, one thread per pixel walks the K class
// planes, so every wavefront access is a contiguous run of one plane.
#include <math.h>

#include <mutex>

#include "common.h"

namespace segk {

constexpr int CE_THREADS = 256;
constexpr int CE_MAX_ROWS = 8192;

static int ce_rows(long long pixels) {
  long long r = (pixels + CE_THREADS - 1) / CE_THREADS;
  if (r < 1) r = 1;
  if (r > CE_MAX_ROWS) r = CE_MAX_ROWS;
  return (int)r;
}

__device__ __forceinline__ int label_of(float t) { return (int)(long long)t; }  // .long(): truncation

__global__ __launch_bounds__(CE_THREADS) void ce_fwd_kernel(const float* __restrict__ logits,
                                                             const float* __restrict__ target,
                                                             const float* __restrict__ weight, int K, int HW,
                                                             long long P, int ignore_index, float* __restrict__ lse_out,
                                                             float* __restrict__ partials) {
  float loss = 0.f, wsum = 0.f;
  for (long long i = (long long)blockIdx.x * CE_THREADS + threadIdx.x; i < P; i += (long long)gridDim.x * CE_THREADS) {
    const long long n = i / HW;
    const int p = (int)(i - n * HW);
    const float* l = logits + (size_t)n * K * HW + p;
    // online log-sum-exp over groups of eight class planes: the eight loads of a group are independent, so a thread
    // keeps eight requests in flight instead of one per branch of the running maximum.  The label's logit is
    // picked from the registers on the way (a per-lane gather would cost a cache line per lane).
    const int t = label_of(target[i]);
    float m = -INFINITY, s = 0.f, vt = 0.f;
    for (int k0 = 0; k0 < K; k0 += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = k0 + u < K ? l[(size_t)(k0 + u) * HW] : -INFINITY;
      float gm = v[0];
#pragma unroll
      for (int u = 1; u < 8; ++u) gm = fmaxf(gm, v[u]);
      if (gm > m) {
        s *= __expf(m - gm);
        m = gm;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        s += __expf(v[u] - m);
        if (k0 + u == t) vt = v[u];
      }
    }
    const float lse = m + __logf(s);
    lse_out[i] = lse;
    if (t != ignore_index && t >= 0 && t < K) {
      const float w = weight ? weight[t] : 1.f;
      loss += w * (lse - vt);
      wsum += w;
    }
  }
  __shared__ float ra[CE_THREADS / 64], rb[CE_THREADS / 64];
  loss = wave_sum(loss);
  wsum = wave_sum(wsum);
  if ((threadIdx.x & 63) == 0) {
    ra[threadIdx.x >> 6] = loss;
    rb[threadIdx.x >> 6] = wsum;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[2 * blockIdx.x + 0] = (ra[0] + ra[1]) + (ra[2] + ra[3]);
    partials[2 * blockIdx.x + 1] = (rb[0] + rb[1]) + (rb[2] + rb[3]);
  }
}

// one block; rows are summed in a fixed order (strided per thread, then a fixed tree) in double
__global__ __launch_bounds__(256) void ce_finish_kernel(const float* __restrict__ partials, int rows,
                                                         float* __restrict__ sums) {
  __shared__ double sa[256], sb[256];
  double a = 0.0, b = 0.0;
  for (int r = threadIdx.x; r < rows; r += 256) {
    a += (double)partials[2 * r];
    b += (double)partials[2 * r + 1];
  }
  sa[threadIdx.x] = a;
  sb[threadIdx.x] = b;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      sa[threadIdx.x] += sa[threadIdx.x + o];
      sb[threadIdx.x] += sb[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    sums[0] = (float)sa[0];
    sums[1] = (float)sb[0];
  }
}

__global__ __launch_bounds__(CE_THREADS) void ce_bwd_kernel(const float* __restrict__ logits,
                                                             const float* __restrict__ target,
                                                             const float* __restrict__ weight,
                                                             const float* __restrict__ lse, int K, int HW, long long P,
                                                             int ignore_index, const float* __restrict__ coef_dev,
                                                             float* __restrict__ dlogits) {
  const float coef = *coef_dev;
  for (long long i = (long long)blockIdx.x * CE_THREADS + threadIdx.x; i < P; i += (long long)gridDim.x * CE_THREADS) {
    const long long n = i / HW;
    const int p = (int)(i - n * HW);
    const size_t base = (size_t)n * K * HW + p;
    const int t = label_of(target[i]);
    const bool valid = (t != ignore_index && t >= 0 && t < K);
    if (!valid) {
      for (int k = 0; k < K; ++k) dlogits[base + (size_t)k * HW] = 0.f;
      continue;
    }
    const float c = coef * (weight ? weight[t] : 1.f);
    const float e = lse[i];
    for (int k = 0; k < K; ++k) {
      const float sm = __expf(logits[base + (size_t)k * HW] - e);
      dlogits[base + (size_t)k * HW] = c * (sm - (k == t ? 1.f : 0.f));
    }
  }
}

__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ logits, int K, int HW, long long P,
                                                      int64_t* __restrict__ pred) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < P; i += (long long)gridDim.x * 256) {
    const long long n = i / HW;
    const int p = (int)(i - n * HW);
    const float* l = logits + (size_t)n * K * HW + p;
    float best = l[0];
    int bi = 0;
    for (int k0 = 1; k0 < K; k0 += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = k0 + u < K ? l[(size_t)(k0 + u) * HW] : 0.f;   // eight loads in flight
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        // first maximal index; a NaN is maximal (and the first NaN wins)
        if (k0 + u < K && (v[u] > best || (v[u] != v[u] && best == best))) {
          best = v[u];
          bi = k0 + u;
        }
      }
    }
    pred[i] = bi;
  }
}

// LDS-privatised histogram: each block counts into K*K (+1) 32-bit bins in LDS and flushes the non-zero
// bins with one 64-bit integer atomic each (integer atomics commute, so the result is exact and
// reproducible).  Each block sees < 2^32 pixels.
template <bool GT_I64, bool USE_LDS>
__global__ __launch_bounds__(256) void confmat_kernel(const void* __restrict__ gt_, const int64_t* __restrict__ pred,
                                                       long long count, int K, int copies,
                                                       unsigned long long* __restrict__ cm) {
  // `copies` (a power of two) private histograms per block, chosen by lane: neighbouring pixels mostly fall into the
  // same bin, and LDS atomics of one wavefront on one address are served one after the other
  extern __shared__ __attribute__((aligned(16))) unsigned int bins[];
  const int nb = K * K + 1;
  if (USE_LDS) {
    for (int b = threadIdx.x; b < nb * copies; b += 256) bins[b] = 0u;
    __syncthreads();
  }
  unsigned int* mine = bins + (threadIdx.x & (copies - 1)) * nb;
  auto label = [&](long long i, bool& ok) -> long long {
    if (GT_I64) {
      const long long g = ((const int64_t*)gt_)[i];
      ok = g >= 0 && g < K;
      return g;
    }
    const float f = ((const float*)gt_)[i];
    ok = f >= 0.f && f < (float)K;
    return ok ? (long long)f : 0;
  };
  auto count_one = [&](long long g, bool ok, long long pr) {
    if (!ok) return;
    const int b = (pr >= 0 && pr < K) ? (int)(g * K + pr) : K * K;
    if (USE_LDS)
      atomicAdd(&mine[b], 1u);
    else
      atomicAdd(&cm[b], 1ull);
  };
  const long long step = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * step < count; i += 4 * step) {   // eight loads in flight before the first atomic
    long long g[4], pr[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      g[u] = label(i + u * step, ok[u]);
      pr[u] = pred[i + u * step];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) count_one(g[u], ok[u], pr[u]);
  }
  for (; i < count; i += step) {
    bool ok;
    const long long g = label(i, ok);
    count_one(g, ok, pred[i]);
  }
  if (USE_LDS) {
    __syncthreads();
    for (int b = threadIdx.x; b < nb; b += 256) {
      unsigned int v = 0u;
      for (int c = 0; c < copies; ++c) v += bins[c * nb + b];
      if (v) atomicAdd(&cm[b], (unsigned long long)v);
    }
  }
}

int ce_partial_rows(long long pixels) { return ce_rows(pixels); }
int launch_ce_finish(const float* partials, int rows, float* sums, hipStream_t st) {
  ce_finish_kernel<<<1, 256, 0, st>>>(partials, rows, sums);
  SEGK_CHECK_LAUNCH("ce_finish");
  return SEGK_OK;
}

}  // namespace segk

using namespace segk;

extern "C" int segk_ce_partial_rows(long long pixels) { return pixels > 0 ? ce_rows(pixels) : SEGK_INVALID_ARG; }

extern "C" int segk_ce_fwd(const float* logits, const float* target, const float* weight, int N, int K, int HW,
                           int ignore_index, float* lse, float* partials, float* sums, segk_stream_t stream) {
  SEGK_REQUIRE(logits && target && lse && partials && sums, "ce_fwd: null pointer");
  SEGK_REQUIRE(N > 0 && K > 0 && HW > 0, "ce_fwd: bad shape N=%d K=%d HW=%d", N, K, HW);
  const long long P = (long long)N * HW;
  const int rows = ce_rows(P);
  hipStream_t st = (hipStream_t)stream;
  ce_fwd_kernel<<<rows, CE_THREADS, 0, st>>>(logits, target, weight, K, HW, P, ignore_index, lse, partials);
  SEGK_CHECK_LAUNCH("ce_fwd");
  ce_finish_kernel<<<1, 256, 0, st>>>(partials, rows, sums);
  SEGK_CHECK_LAUNCH("ce_finish");
  return SEGK_OK;
}

// ---- segk_ce_fwd without any allocation ---------------------------------------------------------------
// The partial rows of segk_ce_fwd live in a file-scope device array instead of a caller workspace.  Every GPU
// the library runs on holds its own copy at its own address, found once per device with hipGetSymbolAddress
// (host code never passes the array's name to a kernel).
constexpr long long CE_STATIC_MAX_PIXELS = 1ll << 25;  // <= 16 pixels per thread of the 8192 x 256 grid
static __device__ __align__(256) float scratch[2 * CE_MAX_ROWS];

namespace {
constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mu;
float* g_scratch_ptr[MAX_DEVICES] = {};

// Address of this device's copy of `scratch` (looked up on the first call per device, under a mutex).  The lookup
// answers for the CURRENT device, the launch goes to the stream's: a stream of another device is refused.
int scratch_for_current_device(float** out, hipStream_t st) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess || dev < 0 || dev >= MAX_DEVICES) {
    set_error("ce_fwd_static: hipGetDevice: %s (device %d)", hipGetErrorString(e), dev);
    return SEGK_HIP_ERROR;
  }
  if (st) {
    hipDevice_t sdev = -1;
    e = hipStreamGetDevice(st, &sdev);
    if (e == hipSuccess && (int)sdev != dev) {
      set_error("ce_fwd_static: the stream belongs to device %d but device %d is current; make the stream's device "
                "current before the call", (int)sdev, dev);
      return SEGK_INVALID_ARG;
    }
  }
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !p) {
      set_error("ce_fwd_static: hipGetSymbolAddress: %s", hipGetErrorString(e));
      return SEGK_HIP_ERROR;
    }
    g_scratch_ptr[dev] = (float*)p;
  }
  *out = g_scratch_ptr[dev];
  return SEGK_OK;
}
}  // namespace

extern "C" long long segk_ce_fwd_static_max_pixels(void) { return CE_STATIC_MAX_PIXELS; }

extern "C" int segk_ce_fwd_static(const float* logits, const float* target, const float* weight, int N, int K, int HW,
                                  int ignore_index, float* lse, float* sums, segk_stream_t stream) {
  SEGK_REQUIRE(logits && target && lse && sums, "ce_fwd_static: null pointer");
  SEGK_REQUIRE(N > 0 && K > 0 && HW > 0, "ce_fwd_static: bad shape N=%d K=%d HW=%d", N, K, HW);
  const long long P = (long long)N * HW;
  SEGK_UNSUPPORTED(P <= CE_STATIC_MAX_PIXELS,
                   "ce_fwd_static: %lld pixels exceed the %lld this entry point's fixed scratch serves; use "
                   "segk_ce_fwd with a caller workspace", P, CE_STATIC_MAX_PIXELS);
  float* part = nullptr;
  hipStream_t st = (hipStream_t)stream;
  const int rc = scratch_for_current_device(&part, st);
  if (rc != SEGK_OK) return rc;
  const int rows = ce_rows(P);
  // every row below `rows` is overwritten by its workgroup and the finish kernel reads only those rows, so the
  // previous call's contents need no clearing
  ce_fwd_kernel<<<rows, CE_THREADS, 0, st>>>(logits, target, weight, K, HW, P, ignore_index, lse, part);
  SEGK_CHECK_LAUNCH("ce_fwd_static");
  ce_finish_kernel<<<1, 256, 0, st>>>(part, rows, sums);
  SEGK_CHECK_LAUNCH("ce_finish");
  return SEGK_OK;
}

extern "C" int segk_ce_bwd(const float* logits, const float* target, const float* weight, const float* lse, int N,
                           int K, int HW, int ignore_index, const float* coef_dev, float* dlogits,
                           segk_stream_t stream) {
  SEGK_REQUIRE(logits && target && lse && coef_dev && dlogits, "ce_bwd: null pointer");
  SEGK_REQUIRE(N > 0 && K > 0 && HW > 0, "ce_bwd: bad shape N=%d K=%d HW=%d", N, K, HW);
  const long long P = (long long)N * HW;
  long long blocks = (P + CE_THREADS - 1) / CE_THREADS;
  if (blocks > 16384) blocks = 16384;
  ce_bwd_kernel<<<(int)blocks, CE_THREADS, 0, (hipStream_t)stream>>>(logits, target, weight, lse, K, HW, P, ignore_index,
                                                                     coef_dev, dlogits);
  SEGK_CHECK_LAUNCH("ce_bwd");
  return SEGK_OK;
}

extern "C" int segk_argmax(const float* logits, int N, int K, int HW, int64_t* pred, segk_stream_t stream) {
  SEGK_REQUIRE(logits && pred, "argmax: null pointer");
  SEGK_REQUIRE(N > 0 && K > 0 && HW > 0, "argmax: bad shape N=%d K=%d HW=%d", N, K, HW);
  const long long P = (long long)N * HW;
  long long blocks = (P + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  argmax_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>(logits, K, HW, P, pred);
  SEGK_CHECK_LAUNCH("argmax");
  return SEGK_OK;
}

extern "C" int segk_confmat(const void* gt, int gt_is_i64, const int64_t* pred, long long count, int K, int64_t* cm,
                            segk_stream_t stream) {
  SEGK_REQUIRE(gt && pred && cm, "confmat: null pointer");
  SEGK_REQUIRE(count >= 0 && K > 0 && K <= 32768, "confmat: bad shape count=%lld K=%d", count, K);
  if (count == 0) return SEGK_OK;
  const size_t one = ((size_t)K * K + 1) * sizeof(unsigned int);
  const bool use_lds = one <= 60 * 1024;
  int copies = 1;
  while (copies < 16 && one * copies * 2 <= 32 * 1024) copies *= 2;
  const size_t lds = one * copies;
  // each block must see fewer than 2^32 pixels; 1024 blocks of >= 16K pixels keep the flush cost small
  long long blocks = (count + 16383) / 16384;
  if (blocks < 1) blocks = 1;
  if (blocks > 1024) blocks = 1024;
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* c = (unsigned long long*)cm;
  if (use_lds) {
    if (gt_is_i64)
      confmat_kernel<true, true><<<(int)blocks, 256, lds, st>>>(gt, pred, count, K, copies, c);
    else
      confmat_kernel<false, true><<<(int)blocks, 256, lds, st>>>(gt, pred, count, K, copies, c);
  } else {
    if (gt_is_i64)
      confmat_kernel<true, false><<<(int)blocks, 256, 0, st>>>(gt, pred, count, K, 1, c);
    else
      confmat_kernel<false, false><<<(int)blocks, 256, 0, st>>>(gt, pred, count, K, 1, c);
  }
  SEGK_CHECK_LAUNCH("confmat");
  return SEGK_OK;
}

// SGD with momentum and weight decay over many tensors in one launch (the optimizer of train.py:56-61:
// torch.optim.SGD(momentum=0.9, weight_decay=5e-4, nesterov=False) on two parameter groups).
//
// The host cuts every tensor into chunks of at most SGD_CHUNK elements and hands the kernel one table entry per
// chunk; workgroup b owns entry b.  HBM-bound: 12 bytes read and 8 written per fp32 parameter; a bf16 parameter
// reads its 2-byte gradient and writes a 2-byte copy next to its fp32 master.
#include "common.h"

namespace segk {

constexpr int SGD_THREADS = 256;

template <typename TP, typename TG>
__device__ __forceinline__ void sgd_chunk(const segk_sgd_chunk& c, float lr, float momentum, float wd) {
  TP* __restrict__ p = (TP*)c.param;
  const TG* __restrict__ g = (const TG*)c.grad;
  float* __restrict__ m = c.momentum;
  float* __restrict__ w = c.master;
  const int n = c.n;
  const bool vec = (((uintptr_t)p % (4 * sizeof(TP))) | ((uintptr_t)g % (4 * sizeof(TG))) |
                    (((uintptr_t)m | (uintptr_t)w) % 16)) == 0;
  const int n4 = vec ? n / 4 : 0;
  for (int i = threadIdx.x; i < n4; i += SGD_THREADS) {
    const f32x4 wv = reinterpret_cast<const f32x4*>(w)[i];
    const f32x4 mv = reinterpret_cast<const f32x4*>(m)[i];
    float gv[4];
    if (sizeof(TG) == 4) {
      const f32x4 t = reinterpret_cast<const f32x4*>(g)[i];
      gv[0] = t[0]; gv[1] = t[1]; gv[2] = t[2]; gv[3] = t[3];
    } else {
      const u32x2 t = reinterpret_cast<const u32x2*>(g)[i];
      gv[0] = __uint_as_float(t[0] << 16); gv[1] = __uint_as_float(t[0] & 0xffff0000u);
      gv[2] = __uint_as_float(t[1] << 16); gv[3] = __uint_as_float(t[1] & 0xffff0000u);
    }
    f32x4 wn, mn;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = gv[e] + wd * wv[e];
      mn[e] = momentum * mv[e] + d;
      wn[e] = wv[e] - lr * mn[e];
    }
    reinterpret_cast<f32x4*>(m)[i] = mn;
    reinterpret_cast<f32x4*>(w)[i] = wn;
    if (sizeof(TP) == 2) {
      u32x2 o = {pack_bf16x2(wn[0], wn[1]), pack_bf16x2(wn[2], wn[3])};
      reinterpret_cast<u32x2*>(p)[i] = o;
    }
  }
  for (int i = n4 * 4 + threadIdx.x; i < n; i += SGD_THREADS) {
    const float wv = w[i];
    const float d = Elem<TG>::load(g + i) + wd * wv;
    const float mn = momentum * m[i] + d;
    const float wn = wv - lr * mn;
    m[i] = mn;
    w[i] = wn;
    if (sizeof(TP) == 2) Elem<TP>::store(p + i, wn);
  }
}

__global__ __launch_bounds__(SGD_THREADS) void sgd_kernel(const segk_sgd_chunk* __restrict__ table, float lr,
                                                          float momentum, float wd) {
  const segk_sgd_chunk c = table[blockIdx.x];
  const int pd = c.dtype & 0xff, gd = (c.dtype >> 8) & 0xff;   // gradient type in the second byte
  if (pd == SEGK_BF16) {
    if (gd == SEGK_BF16) sgd_chunk<bf16_t, bf16_t>(c, lr, momentum, wd);
    else sgd_chunk<bf16_t, float>(c, lr, momentum, wd);
  } else {
    sgd_chunk<float, float>(c, lr, momentum, wd);
  }
}

// the same update with the three hyper-parameters read from device memory: a launch recorded in a HIP graph keeps
// following the learning-rate schedule, which the host writes into `hyper` before each replay
__global__ __launch_bounds__(SGD_THREADS) void sgd_dev_kernel(const segk_sgd_chunk* __restrict__ table,
                                                              const float* __restrict__ hyper) {
  const float lr = hyper[0], momentum = hyper[1], wd = hyper[2];
  const segk_sgd_chunk c = table[blockIdx.x];
  const int pd = c.dtype & 0xff, gd = (c.dtype >> 8) & 0xff;
  if (pd == SEGK_BF16) {
    if (gd == SEGK_BF16) sgd_chunk<bf16_t, bf16_t>(c, lr, momentum, wd);
    else sgd_chunk<bf16_t, float>(c, lr, momentum, wd);
  } else {
    sgd_chunk<float, float>(c, lr, momentum, wd);
  }
}

// dst[i] = (float)src[i] over many tensors: packs gradients of mixed types into a flat fp32 all-reduce bucket
__global__ __launch_bounds__(SGD_THREADS) void gather_f32_kernel(const segk_gather_chunk* __restrict__ table) {
  const segk_gather_chunk c = table[blockIdx.x];
  float* __restrict__ d = c.dst;
  if (c.dtype == SEGK_BF16) {
    const bf16_t* __restrict__ s = (const bf16_t*)c.src;
    for (int i = threadIdx.x; i < c.n; i += SGD_THREADS) d[i] = bf16_to_f32(s[i]);
  } else {
    const float* __restrict__ s = (const float*)c.src;
    for (int i = threadIdx.x; i < c.n; i += SGD_THREADS) d[i] = s[i];
  }
}

}  // namespace segk

using namespace segk;

extern "C" int segk_sgd_step(const segk_sgd_chunk* table, int n_chunks, float lr, float momentum, float weight_decay,
                             segk_stream_t stream) {
  SEGK_REQUIRE(table && n_chunks > 0, "sgd_step: empty table");
  sgd_kernel<<<n_chunks, SGD_THREADS, 0, (hipStream_t)stream>>>(table, lr, momentum, weight_decay);
  SEGK_CHECK_LAUNCH("sgd_step");
  return SEGK_OK;
}

extern "C" int segk_sgd_step_dev(const segk_sgd_chunk* table, int n_chunks, const float* hyper, segk_stream_t stream) {
  SEGK_REQUIRE(table && n_chunks > 0, "sgd_step_dev: empty table");
  SEGK_REQUIRE(hyper && ((uintptr_t)hyper & 3) == 0, "sgd_step_dev: hyper must be a 4-byte aligned device pointer");
  sgd_dev_kernel<<<n_chunks, SGD_THREADS, 0, (hipStream_t)stream>>>(table, hyper);
  SEGK_CHECK_LAUNCH("sgd_step_dev");
  return SEGK_OK;
}

extern "C" int segk_gather_f32(const segk_gather_chunk* table, int n_chunks, segk_stream_t stream) {
  SEGK_REQUIRE(table && n_chunks > 0, "gather_f32: empty table");
  gather_f32_kernel<<<n_chunks, SGD_THREADS, 0, (hipStream_t)stream>>>(table);
  SEGK_CHECK_LAUNCH("gather_f32");
  return SEGK_OK;
}

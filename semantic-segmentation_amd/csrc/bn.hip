// Batch-normalisation kernels (training with global statistics, evaluation, and their gradients).
// All HBM-bound.  Two layouts: pixel-major [M][ld] (head internals) and NCHW (the feeder's tensors).
// Statistics are produced as per-block partial rows that a second kernel sums in index order, so the
// result does not depend on scheduling (no float atomics).
#include <algorithm>
#include <cstdlib>
#include <initializer_list>
#include <utility>

#include "common.h"

namespace segk {

constexpr int BN_THREADS = 256;
constexpr int BN_MAX_ROWS_PM = 1024;
constexpr int BN_MAX_ROWS_NCHW = 32;

template <typename T, int VEC>
__device__ __forceinline__ void load_vec(const T* p, float (&v)[VEC]);
template <>
__device__ __forceinline__ void load_vec<float, 8>(const float* p, float (&v)[8]) {
  f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
  v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
template <>
__device__ __forceinline__ void load_vec<float, 4>(const float* p, float (&v)[4]) {
  f32x4 t = *reinterpret_cast<const f32x4*>(p);
  v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
template <>
__device__ __forceinline__ void load_vec<float, 1>(const float* p, float (&v)[1]) { v[0] = *p; }
template <>
__device__ __forceinline__ void load_vec<bf16_t, 8>(const bf16_t* p, float (&v)[8]) {
  u32x4 t = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[2 * k] = __uint_as_float(t[k] << 16);
    v[2 * k + 1] = __uint_as_float(t[k] & 0xffff0000u);
  }
}
template <>
__device__ __forceinline__ void load_vec<bf16_t, 4>(const bf16_t* p, float (&v)[4]) {
  u32x2 t = *reinterpret_cast<const u32x2*>(p);
  v[0] = __uint_as_float(t[0] << 16); v[1] = __uint_as_float(t[0] & 0xffff0000u);
  v[2] = __uint_as_float(t[1] << 16); v[3] = __uint_as_float(t[1] & 0xffff0000u);
}
template <>
__device__ __forceinline__ void load_vec<bf16_t, 1>(const bf16_t* p, float (&v)[1]) { v[0] = bf16_to_f32(*p); }

template <typename T, int VEC>
__device__ __forceinline__ void store_vec(T* p, const float (&v)[VEC]);
template <>
__device__ __forceinline__ void store_vec<float, 8>(float* p, const float (&v)[8]) {
  f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
  *reinterpret_cast<f32x4*>(p) = a;
  *reinterpret_cast<f32x4*>(p + 4) = b;
}
template <>
__device__ __forceinline__ void store_vec<float, 4>(float* p, const float (&v)[4]) {
  f32x4 t = {v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4*>(p) = t;
}
template <>
__device__ __forceinline__ void store_vec<float, 1>(float* p, const float (&v)[1]) { *p = v[0]; }
template <>
__device__ __forceinline__ void store_vec<bf16_t, 8>(bf16_t* p, const float (&v)[8]) {
  u32x4 t = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
  *reinterpret_cast<u32x4*>(p) = t;
}
template <>
__device__ __forceinline__ void store_vec<bf16_t, 4>(bf16_t* p, const float (&v)[4]) {
  u32x2 t = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
  *reinterpret_cast<u32x2*>(p) = t;
}
template <>
__device__ __forceinline__ void store_vec<bf16_t, 1>(bf16_t* p, const float (&v)[1]) { *p = f32_to_bf16(v[0]); }

template <int VEC>
__device__ __forceinline__ void load_mask(const uint8_t* p, float (&v)[VEC]);
template <>
__device__ __forceinline__ void load_mask<8>(const uint8_t* p, float (&v)[8]) {
  u32x2 t = *reinterpret_cast<const u32x2*>(p);
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = (float)((t[k >> 2] >> (8 * (k & 3))) & 0xff);
}
template <>
__device__ __forceinline__ void load_mask<4>(const uint8_t* p, float (&v)[4]) {
  unsigned int t = *reinterpret_cast<const unsigned int*>(p);
  v[0] = (float)(t & 0xff); v[1] = (float)((t >> 8) & 0xff); v[2] = (float)((t >> 16) & 0xff); v[3] = (float)(t >> 24);
}
template <>
__device__ __forceinline__ void load_mask<1>(const uint8_t* p, float (&v)[1]) { v[0] = (float)*p; }

// ---------------------------------------------------------------------------------------------
// Pixel-major kernels.  A block owns a contiguous pixel range (blockIdx.x) and a slab of channels
// (blockIdx.y): up to BN_THREADS*VEC of them in the elementwise kernels, 64 (one 128-byte line of bf16
// per pixel) in the statistics kernels, whose partial rows then stay few however the chip is filled.
// Thread t owns channel group (t % cg) and walks the pixels (t / cg) + k*rpb, four at a time so that
// several 16-byte loads are in flight per lane.
// ---------------------------------------------------------------------------------------------
struct PmThread {
  int c_base, Cs, cg, rpb, cgi, pr;
  long long m0, m1;
  bool active;
};

template <int VEC>
__device__ __forceinline__ PmThread pm_thread(int C, long long M, int slab = BN_THREADS * VEC) {
  PmThread t;
  t.c_base = blockIdx.y * slab;
  t.Cs = C - t.c_base < slab ? C - t.c_base : slab;
  t.cg = t.Cs / VEC;
  t.rpb = BN_THREADS / t.cg;
  t.cgi = threadIdx.x % t.cg;
  t.pr = threadIdx.x / t.cg;
  t.active = t.pr < t.rpb;
  const long long per = (M + gridDim.x - 1) / gridDim.x;
  t.m0 = (long long)blockIdx.x * per;
  t.m1 = t.m0 + per < M ? t.m0 + per : M;
  return t;
}

// Combine the per-thread accumulators of all pixel rows of the block and write one partial row.
template <int VEC, typename ACC>
__device__ __forceinline__ void block_reduce_pm(ACC (&a)[VEC], ACC (&b)[VEC], const PmThread& t, double* out_row, int C) {
  __shared__ ACC red[2 * BN_THREADS * 8];
  const int Cs = t.Cs;
  if (t.active) {  // layout red[pr][2][Cs]
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      red[(t.pr * 2 + 0) * Cs + t.cgi * VEC + j] = a[j];
      red[(t.pr * 2 + 1) * Cs + t.cgi * VEC + j] = b[j];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * Cs; i += BN_THREADS) {
    double s = 0.0;
    for (int r = 0; r < t.rpb; ++r) s += (double)red[r * 2 * Cs + i];
    const int which = i / Cs, c = i - which * Cs;
    out_row[which * C + t.c_base + c] = s;
  }
}

template <typename TX, int VEC>
__global__ __launch_bounds__(BN_THREADS) void bn_stats_pm_kernel(const TX* __restrict__ x, int ld, long long M, int C,
                                                                  int slab, double* __restrict__ partials) {
  const PmThread t = pm_thread<VEC>(C, M, slab);
  // fp64 accumulation: squares of fp32 values are exact in fp64, so E[x^2] - E[x]^2 loses nothing
  double a[VEC], b[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) a[j] = b[j] = 0.0;
  if (t.active) {
    const TX* xp = x + t.c_base + t.cgi * VEC;
    const long long step = t.rpb;
    long long m = t.m0 + t.pr;
    for (; m + 3 * step < t.m1; m += 4 * step) {
      float v[4][VEC];
#pragma unroll
      for (int k = 0; k < 4; ++k) load_vec<TX, VEC>(xp + (m + k * step) * ld, v[k]);
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const double d = (double)v[k][j];
          a[j] += d;
          b[j] = fma(d, d, b[j]);
        }
    }
    for (; m < t.m1; m += step) {
      float v[VEC];
      load_vec<TX, VEC>(xp + m * ld, v);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const double d = (double)v[j];
        a[j] += d;
        b[j] = fma(d, d, b[j]);
      }
    }
  }
  block_reduce_pm<VEC, double>(a, b, t, partials + (size_t)blockIdx.x * 2 * C, C);
}

// g = dz * mask * mask_scale * [activation > 0]; the activation is x*scale+shift, or, for layers that add a
// residual before the ReLU, the layer's stored output `act`
template <int VEC>
__device__ __forceinline__ void grad_through_act(float (&gz)[VEC], const float (&xv)[VEC], const float* sc,
                                                 const float* sh, int relu, const uint8_t* mask_p, float mask_scale,
                                                 const float* act = nullptr, int bits = -1) {
  if (mask_p) {
    float mk[VEC];
    load_mask<VEC>(mask_p, mk);
#pragma unroll
    for (int j = 0; j < VEC; ++j) gz[j] *= mk[j] * mask_scale;
  }
  if (relu) {
    if (bits >= 0) {   // the stored ReLU pattern, one bit per channel of this thread's group
#pragma unroll
      for (int j = 0; j < VEC; ++j)
        if (!((bits >> j) & 1)) gz[j] = 0.f;
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j)
        if (!((act ? act[j] : fmaf(xv[j], sc[j], sh[j])) > 0.f)) gz[j] = 0.f;
    }
  }
}

// RES: the layer added a residual before its ReLU; the ReLU pattern then comes from the layer's stored output `act`
// (RES = 1) or from the bit pattern the forward apply stored, one byte per pixel and group of eight channels
// (RES = 2, VEC = 8: `act` then points to those bytes, act_ld bytes per pixel).  A template flag, not a run-time
// test: a branch around a load keeps the compiler from issuing the row's loads together.
template <typename TG, typename TX, int VEC, int RES>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_stats_pm_kernel(
    const TG* __restrict__ dz, int dz_ld, const TX* __restrict__ x, int x_ld, long long M, int C,
    const float* __restrict__ mean, const float* __restrict__ inv_std, const float* __restrict__ scale,
    const float* __restrict__ shift, int relu, const uint8_t* __restrict__ mask, float mask_scale,
    const TG* __restrict__ act, int act_ld, int slab, double* __restrict__ partials) {
  const PmThread t = pm_thread<VEC>(C, M, slab);
  extern __shared__ float cf[];  // [4][Cs]: mean, inv_std, scale, shift of the slab, loaded once per workgroup
  for (int i = threadIdx.x; i < t.Cs; i += BN_THREADS) {
    const int c = t.c_base + i;
    cf[i] = mean[c];
    cf[t.Cs + i] = inv_std[c];
    cf[2 * t.Cs + i] = scale[c];
    cf[3 * t.Cs + i] = shift[c];
  }
  __syncthreads();
  float a[VEC], b[VEC], mu[VEC], is[VEC], sc[VEC], sh[VEC];
  const int c0 = t.c_base + (t.active ? t.cgi * VEC : 0);
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const int i = (t.active ? t.cgi * VEC : 0) + j;
    a[j] = b[j] = 0.f;
    mu[j] = cf[i]; is[j] = cf[t.Cs + i]; sc[j] = cf[2 * t.Cs + i]; sh[j] = cf[3 * t.Cs + i];
  }
  if (t.active) {
    const long long step = t.rpb;
    long long m = t.m0 + t.pr;
    for (; m + step < t.m1; m += 2 * step) {
      float gz[2][VEC], xv[2][VEC], av[2][VEC];
      int bits[2] = {-1, -1};
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        load_vec<TG, VEC>(dz + (m + k * step) * dz_ld + c0, gz[k]);
        load_vec<TX, VEC>(x + (m + k * step) * x_ld + c0, xv[k]);
        if (RES == 1) load_vec<TG, VEC>(act + (m + k * step) * act_ld + c0, av[k]);
        if (RES == 2) bits[k] = reinterpret_cast<const uint8_t*>(act)[(m + k * step) * act_ld + (c0 >> 3)];
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        grad_through_act<VEC>(gz[k], xv[k], sc, sh, relu, mask ? mask + (m + k * step) * C + c0 : nullptr, mask_scale,
                              RES == 1 ? av[k] : nullptr, bits[k]);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          a[j] += gz[k][j];
          b[j] = fmaf(gz[k][j], (xv[k][j] - mu[j]) * is[j], b[j]);
        }
      }
    }
    for (; m < t.m1; m += step) {
      float gz[VEC], xv[VEC], av[VEC];
      int bits = -1;
      load_vec<TG, VEC>(dz + m * dz_ld + c0, gz);
      load_vec<TX, VEC>(x + m * x_ld + c0, xv);
      if (RES == 1) load_vec<TG, VEC>(act + m * act_ld + c0, av);
      if (RES == 2) bits = reinterpret_cast<const uint8_t*>(act)[m * act_ld + (c0 >> 3)];
      grad_through_act<VEC>(gz, xv, sc, sh, relu, mask ? mask + m * C + c0 : nullptr, mask_scale,
                            RES == 1 ? av : nullptr, bits);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        a[j] += gz[j];
        b[j] = fmaf(gz[j], (xv[j] - mu[j]) * is[j], b[j]);
      }
    }
  }
  block_reduce_pm<VEC, float>(a, b, t, partials + (size_t)blockIdx.x * 2 * C, C);
}

template <typename TX, typename TY, int VEC, bool RES>
__global__ __launch_bounds__(BN_THREADS) void bn_apply_pm_kernel(const TX* __restrict__ x, int x_ld, TY* __restrict__ y,
                                                                  int y_ld, long long M, int C,
                                                                  const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, int relu,
                                                                  const uint8_t* __restrict__ mask, float mask_scale,
                                                                  const TY* __restrict__ res, int res_ld,
                                                                  uint8_t* __restrict__ relu_bits, int bits_ld) {
  const PmThread t = pm_thread<VEC>(C, M);
  if (!t.active) return;
  const int c0 = t.c_base + t.cgi * VEC;
  float sc[VEC], sh[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { sc[j] = scale[c0 + j]; sh[j] = shift[c0 + j]; }
#pragma unroll 4
  for (long long m = t.m0 + t.pr; m < t.m1; m += t.rpb) {
    float v[VEC], o[VEC], rv[VEC];
    load_vec<TX, VEC>(x + m * x_ld + c0, v);
    if (RES) load_vec<TY, VEC>(res + m * res_ld + c0, rv);
    unsigned pattern = 0u;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float r = fmaf(v[j], sc[j], sh[j]);
      if (RES) r += rv[j];   // residual added before the ReLU (bottleneck shortcut)
      if (VEC == 8) {
        // the pattern is taken from the value as it is stored (rounded to TY): what the backward would otherwise
        // read back from y
        float q = r;
        if (sizeof(TY) == 2) q = bf16_to_f32(f32_to_bf16(r));
        pattern |= (q > 0.f ? 1u : 0u) << j;
      }
      if (relu) r = r > 0.f ? r : 0.f;
      o[j] = r;
    }
    if (VEC == 8 && relu_bits) relu_bits[m * bits_ld + (c0 >> 3)] = (uint8_t)pattern;
    if (mask) {
      float mk[VEC];
      load_mask<VEC>(mask + m * C + c0, mk);
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[j] *= mk[j] * mask_scale;
    }
    store_vec<TY, VEC>(y + m * y_ld + c0, o);
  }
}

template <typename TG, typename TX, int VEC, int RES>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_apply_pm_kernel(
    const TG* __restrict__ dz, int dz_ld, const TX* __restrict__ x, int x_ld, TG* __restrict__ dx, int dx_ld,
    long long M, int C, const float* __restrict__ mean, const float* __restrict__ inv_std,
    const float* __restrict__ scale, const float* __restrict__ shift, int relu, const uint8_t* __restrict__ mask,
    float mask_scale, const double* __restrict__ gsums, int eval, const TG* __restrict__ act, int act_ld,
    TG* __restrict__ dres, int dres_ld, int slab) {
  const PmThread t = pm_thread<VEC>(C, M, slab);
  // the slab's per-channel coefficients go through LDS: one coalesced load per workgroup instead of 6 x VEC
  // scattered loads (two of them fp64) per thread
  extern __shared__ float cf[];  // [6][Cs]: scale, shift, mean, inv_std, sum_g / count, sum_g_xhat / count
  {
    const float inv_count = eval ? 0.f : (float)(1.0 / gsums[2 * C]);
    for (int i = threadIdx.x; i < t.Cs; i += BN_THREADS) {
      const int c = t.c_base + i;
      cf[i] = scale[c];
      cf[t.Cs + i] = shift[c];
      cf[2 * t.Cs + i] = mean[c];
      cf[3 * t.Cs + i] = inv_std[c];
      cf[4 * t.Cs + i] = eval ? 0.f : (float)gsums[c] * inv_count;
      cf[5 * t.Cs + i] = eval ? 0.f : (float)gsums[C + c] * inv_count;
    }
  }
  __syncthreads();
  if (!t.active) return;
  const int c0 = t.c_base + t.cgi * VEC;
  float sc[VEC], sh[VEC], mu[VEC], is[VEC], k1[VEC], k2[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const int i = t.cgi * VEC + j;
    sc[j] = cf[i]; sh[j] = cf[t.Cs + i]; mu[j] = cf[2 * t.Cs + i]; is[j] = cf[3 * t.Cs + i];
    k1[j] = cf[4 * t.Cs + i]; k2[j] = cf[5 * t.Cs + i];
  }
#pragma unroll 2
  for (long long m = t.m0 + t.pr; m < t.m1; m += t.rpb) {
    float gz[VEC], xv[VEC], o[VEC], av[VEC];
    int bits = -1;
    load_vec<TG, VEC>(dz + m * dz_ld + c0, gz);
    load_vec<TX, VEC>(x + m * x_ld + c0, xv);
    if (RES == 1) load_vec<TG, VEC>(act + m * act_ld + c0, av);
    if (RES == 2) bits = reinterpret_cast<const uint8_t*>(act)[m * act_ld + (c0 >> 3)];
    grad_through_act<VEC>(gz, xv, sc, sh, relu, mask ? mask + m * C + c0 : nullptr, mask_scale,
                          RES == 1 ? av : nullptr, bits);
    if (RES && dres) store_vec<TG, VEC>(dres + m * dres_ld + c0, gz);   // gradient of the residual input (unless the consumer masks dz itself)
#pragma unroll
    for (int j = 0; j < VEC; ++j)
      o[j] = eval ? sc[j] * gz[j] : sc[j] * (gz[j] - k1[j] - (xv[j] - mu[j]) * is[j] * k2[j]);
    store_vec<TG, VEC>(dx + m * dx_ld + c0, o);
  }
}

// ---------------------------------------------------------------------------------------------
// NCHW kernels: block (c, s) reduces pixel slice s of channel c over all images.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <typename ACC>
__device__ __forceinline__ void block_reduce2(ACC& a, ACC& b) {
  __shared__ ACC ra[BN_THREADS / 64], rb[BN_THREADS / 64];
  a = wave_sum(a);
  b = wave_sum(b);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { ra[w] = a; rb[w] = b; }
  __syncthreads();
  a = ra[0] + ra[1] + ra[2] + ra[3];
  b = rb[0] + rb[1] + rb[2] + rb[3];
}

template <typename TX>
__global__ __launch_bounds__(BN_THREADS) void bn_stats_nchw_kernel(const TX* __restrict__ x, int N, int C, int HW,
                                                                    double* __restrict__ partials) {
  const int c = blockIdx.x, s = blockIdx.y, S = gridDim.y;
  const int per = (HW + S - 1) / S;
  const int p0 = s * per, p1 = min(HW, p0 + per);
  double a = 0.0, b = 0.0;
  for (int n = 0; n < N; ++n) {
    const TX* row = x + ((size_t)n * C + c) * HW;
    for (int p = p0 + threadIdx.x; p < p1; p += BN_THREADS) {
      const double v = (double)Elem<TX>::load(row + p);
      a += v;
      b = fma(v, v, b);
    }
  }
  block_reduce2(a, b);
  if (threadIdx.x == 0) {
    partials[((size_t)s * 2 + 0) * C + c] = a;
    partials[((size_t)s * 2 + 1) * C + c] = b;
  }
}

template <typename TG, typename TX>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_stats_nchw_kernel(
    const TG* __restrict__ dz, const TX* __restrict__ x, int N, int C, int HW, const float* __restrict__ mean,
    const float* __restrict__ inv_std, const float* __restrict__ scale, const float* __restrict__ shift, int relu,
    const uint8_t* __restrict__ mask, float mask_scale, double* __restrict__ partials) {
  const int c = blockIdx.x, s = blockIdx.y, S = gridDim.y;
  const int per = (HW + S - 1) / S;
  const int p0 = s * per, p1 = min(HW, p0 + per);
  const float mu = mean[c], is = inv_std[c], sc = scale[c], sh = shift[c];
  float a = 0.f, b = 0.f;
  for (int n = 0; n < N; ++n) {
    const size_t base = ((size_t)n * C + c) * HW;
    for (int p = p0 + threadIdx.x; p < p1; p += BN_THREADS) {
      float g = Elem<TG>::load(dz + base + p);
      const float v = Elem<TX>::load(x + base + p);
      if (mask) g *= (float)mask[base + p] * mask_scale;
      if (relu && !(fmaf(v, sc, sh) > 0.f)) g = 0.f;
      a += g;
      b = fmaf(g, (v - mu) * is, b);
    }
  }
  block_reduce2(a, b);
  if (threadIdx.x == 0) {
    partials[((size_t)s * 2 + 0) * C + c] = a;
    partials[((size_t)s * 2 + 1) * C + c] = b;
  }
}

template <typename TX, typename TY>
__global__ __launch_bounds__(BN_THREADS) void bn_apply_nchw_kernel(const TX* __restrict__ x, TY* __restrict__ y, int C,
                                                                    int HW, const float* __restrict__ scale,
                                                                    const float* __restrict__ shift, int relu,
                                                                    const uint8_t* __restrict__ mask, float mask_scale) {
  const int c = blockIdx.x % C;
  const size_t base = (size_t)blockIdx.x * HW;
  const int per = (HW + gridDim.y - 1) / gridDim.y;
  const int p0 = blockIdx.y * per, p1 = min(HW, p0 + per);
  const float sc = scale[c], sh = shift[c];
  for (int p = p0 + threadIdx.x; p < p1; p += BN_THREADS) {
    float t = fmaf(Elem<TX>::load(x + base + p), sc, sh);
    if (relu) t = t > 0.f ? t : 0.f;
    if (mask) t *= (float)mask[base + p] * mask_scale;
    Elem<TY>::store(y + base + p, t);
  }
}

template <typename TG, typename TX>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_apply_nchw_kernel(
    const TG* __restrict__ dz, const TX* __restrict__ x, TG* __restrict__ dx, int C, int HW,
    const float* __restrict__ mean, const float* __restrict__ inv_std, const float* __restrict__ scale,
    const float* __restrict__ shift, int relu, const uint8_t* __restrict__ mask, float mask_scale,
    const double* __restrict__ gsums, int eval) {
  const int c = blockIdx.x % C;
  const size_t base = (size_t)blockIdx.x * HW;
  const int per = (HW + gridDim.y - 1) / gridDim.y;
  const int p0 = blockIdx.y * per, p1 = min(HW, p0 + per);
  const float mu = mean[c], is = inv_std[c], sc = scale[c], sh = shift[c];
  const float inv_count = eval ? 0.f : (float)(1.0 / gsums[2 * C]);
  const float k1 = eval ? 0.f : (float)gsums[c] * inv_count, k2 = eval ? 0.f : (float)gsums[C + c] * inv_count;
  for (int p = p0 + threadIdx.x; p < p1; p += BN_THREADS) {
    float g = Elem<TG>::load(dz + base + p);
    const float v = Elem<TX>::load(x + base + p);
    if (mask) g *= (float)mask[base + p] * mask_scale;
    if (relu && !(fmaf(v, sc, sh) > 0.f)) g = 0.f;
    const float o = eval ? sc * g : sc * (g - k1 - (v - mu) * is * k2);
    Elem<TG>::store(dx + base + p, o);
  }
}

// ---------------------------------------------------------------------------------------------
// Block = 16 columns x 16 row phases; each thread sums rows phase, phase+16, ... of its column, then the 16
// phase sums of a column are added in phase order: a fixed summation order, independent of scheduling.
// Sum of rows ph, ph + 16, ... of one column, added in that order.  Sixteen loads are issued before the first
// addition: these launches sit between two kernels that wait for them and are pure latency (a few hundred rows of
// a few KB), so the number of dependent round trips to memory is what they cost.  (Rows past the end add 0.0.)
__device__ __forceinline__ double phase_sum(const double* __restrict__ partials, int rows, int n2, int col, int ph) {
  double s = 0.0;
  for (int r0 = ph; r0 < rows; r0 += 16 * 16) {
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int r = r0 + u * 16;
      v[u] = r < rows ? partials[(size_t)r * n2 + col] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) s += v[u];
  }
  return s;
}

__global__ __launch_bounds__(256) void bn_reduce_kernel(const double* __restrict__ partials, int rows, int C,
                                                         double count, double* __restrict__ sums,
                                                         float* __restrict__ sums_f32) {
  __shared__ double red[16][17];
  const int col = blockIdx.x * 16 + (threadIdx.x & 15);
  const int ph = threadIdx.x >> 4;
  const int n2 = 2 * C;
  double s = 0.0;
  if (col < n2) s = phase_sum(partials, rows, n2, col, ph);
  red[ph][threadIdx.x & 15] = s;
  __syncthreads();
  if (threadIdx.x < 16) {
    const int c = blockIdx.x * 16 + threadIdx.x;
    if (c < n2) {
      double t = 0.0;
#pragma unroll
      for (int p = 0; p < 16; ++p) t += red[p][threadIdx.x];
      sums[c] = t;
      if (sums_f32) sums_f32[c] = (float)t;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) sums[n2] = count;
}

// global sums of one channel -> mean, inv_std, running statistics, scale, shift
__device__ __forceinline__ void bn_coeff_one(double s, double q, double cnt, int c, const float* gamma, const float* beta,
                                             float eps, float momentum, int legacy_clamp, float* running_mean,
                                             float* running_var, float* mean, float* inv_std, float* scale,
                                             float* shift) {
  // sums are fp64 end to end (exact squares, fp64 accumulation), so this cancellation is harmless
  const double mu = s / cnt;
  double sumvar = q - s * mu;
  if (sumvar < 0.0) sumvar = 0.0;
  const double var_b = sumvar / cnt;
  const double var_u = sumvar / (cnt - 1.0);
  const double is = legacy_clamp ? 1.0 / sqrt(var_b > (double)eps ? var_b : (double)eps) : 1.0 / sqrt(var_b + (double)eps);
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
  if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)var_u;
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  const float muf = (float)mu, isf = (float)is;
  mean[c] = muf;
  inv_std[c] = isf;
  const float sc = g * isf;
  scale[c] = sc;
  shift[c] = b - muf * sc;
}

__global__ void bn_coeffs_kernel(const double* __restrict__ sums, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, float eps, float momentum, int legacy_clamp,
                                 float* __restrict__ running_mean, float* __restrict__ running_var,
                                 float* __restrict__ mean, float* __restrict__ inv_std, float* __restrict__ scale,
                                 float* __restrict__ shift, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  bn_coeff_one(sums[c], sums[C + c], sums[2 * C], c, gamma, beta, eps, momentum, legacy_clamp, running_mean, running_var,
               mean, inv_std, scale, shift);
}

// bn_reduce_kernel + bn_coeffs_kernel in one launch for statistics that need no exchange between ranks: a block
// owns 8 channels (columns c and C + c of the partial rows), same fixed summation order as bn_reduce_kernel.
__global__ __launch_bounds__(256) void bn_reduce_coeffs_kernel(const double* __restrict__ partials, int rows, int C,
                                                                double count, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float eps, float momentum,
                                                                int legacy_clamp, float* __restrict__ running_mean,
                                                                float* __restrict__ running_var, float* __restrict__ mean,
                                                                float* __restrict__ inv_std, float* __restrict__ scale,
                                                                float* __restrict__ shift, double* __restrict__ sums) {
  __shared__ double red[16][17];
  const int j = threadIdx.x & 15, ph = threadIdx.x >> 4;
  const int ch = blockIdx.x * 8 + (j & 7);
  const int n2 = 2 * C;
  double s = 0.0;
  if (ch < C) {
    s = phase_sum(partials, rows, n2, (j >> 3) * C + ch, ph);
  }
  red[ph][j] = s;
  __syncthreads();
  if (threadIdx.x < 8 && ch < C) {
    double t0 = 0.0, t1 = 0.0;
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      t0 += red[p][threadIdx.x];
      t1 += red[p][threadIdx.x + 8];
    }
    if (sums) {
      sums[ch] = t0;
      sums[C + ch] = t1;
    }
    bn_coeff_one(t0, t1, count, ch, gamma, beta, eps, momentum, legacy_clamp, running_mean, running_var, mean, inv_std,
                 scale, shift);
  }
  if (sums && blockIdx.x == 0 && threadIdx.x == 0) sums[n2] = count;
}

__global__ void bn_coeffs_eval_kernel(const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                      const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                      float* __restrict__ mean, float* __restrict__ inv_std, float* __restrict__ scale,
                                      float* __restrict__ shift, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float mu = running_mean[c];
  const float is = 1.f / sqrtf(running_var[c] + eps);
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  mean[c] = mu;
  inv_std[c] = is;
  scale[c] = g * is;
  shift[c] = b - mu * (g * is);
}

static bool aligned(const void* p, size_t a) { return ((uintptr_t)p % a) == 0; }

// channels per workgroup (grid.y slabs) of the pixel-major statistics kernels
static int pm_stat_slab(int C) { return C > 64 ? 64 : C; }

static int pm_stat_blocks() {
  static const int blocks = [] {
#ifdef SEGK_EXPERIMENTS
    const char* e = getenv("SEGK_BN_STAT_BLOCKS");
#else
    const char* e = nullptr;
#endif
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : 512;
  }();
  return blocks;
}

// partial rows (= pixel ranges = grid.x) of the pixel-major statistics kernels: about pm_stat_blocks() workgroups
// over rows x slabs, at least 64 pixels per row
static int pm_rows(long long M, int C) {
  const int slabs = ceil_div(C, pm_stat_slab(C));
  long long r = ceil_div(pm_stat_blocks(), slabs);
  const long long by_pixels = (M + 63) / 64;
  if (r > by_pixels) r = by_pixels;
  if (r > BN_MAX_ROWS_PM) r = BN_MAX_ROWS_PM;
  if (r < 1) r = 1;
  return (int)r;
}

// widest channel vector every operand allows: 8 or 4 elements need C and the leading dimensions to be
// multiples of it and 16-byte aligned base pointers (8-byte for 4 bf16)
static int pm_vec(int C, std::initializer_list<int> lds, std::initializer_list<std::pair<const void*, int>> ptrs,
                  const uint8_t* mask) {
  for (int v : {8, 4}) {
    bool ok = C % v == 0 && (!mask || aligned(mask, v));
    for (int ld : lds) ok = ok && ld % v == 0;
    for (auto& p : ptrs) ok = ok && aligned(p.first, p.second == SEGK_BF16 ? 2 * v : 16);
    if (ok) return v;
  }
  return 1;
}

// Grid of the pixel-major elementwise kernels: about `wgs` workgroups over pixel ranges x channel slabs, at least
// `min_rows` pixels per thread.  Measured with operands that are not in the memory-side cache (tools/bn_apply_sweep.sh,
// 16 x 33 x 33 x 1024 bf16): the forward pass, whose threads load their own two coefficient vectors, is fastest with
// many short workgroups (4096, two pixels per thread: 17.5 -> 15.2 us); the backward pass stages six coefficient
// vectors of its slab through LDS and wants that amortised - slabs of 128 channels (3 KB of coefficients instead of
// 24 KB for a 1024-channel row) and about 1024 workgroups: 30.9 -> 26.4 us with the ReLU bits, 33.5 -> 25.4 without.
static dim3 pm_apply_grid(long long M, int C, int vec, int wgs, int min_rows, int slab = 0) {
  if (slab <= 0) slab = BN_THREADS * vec;
  const int cg = (C < slab ? C : slab) / vec;
  const int rpb = BN_THREADS / cg;
  const int slabs = ceil_div(C, slab);
  const long long want = wgs / slabs > 0 ? wgs / slabs : 1;
  long long rpt = (M + (long long)rpb * want - 1) / ((long long)rpb * want);
  if (rpt < min_rows) rpt = min_rows;
  long long gx = (M + rpb * rpt - 1) / (rpb * rpt);
  if (gx < 1) gx = 1;
  return dim3((unsigned)gx, (unsigned)slabs);
}

static int env_int(const char* name, int fallback) {
#ifdef SEGK_EXPERIMENTS
  const char* e = getenv(name);
  if (e && atoi(e) > 0) return atoi(e);
#endif
  (void)name;
  return fallback;
}

#define SEGK_DISPATCH_VEC(vec, V, ...)             \
  do {                                             \
    if ((vec) == 8) { constexpr int V = 8; __VA_ARGS__; }      \
    else if ((vec) == 4) { constexpr int V = 4; __VA_ARGS__; } \
    else { constexpr int V = 1; __VA_ARGS__; }                 \
  } while (0)

static int nchw_rows(int N, int HW) {
  long long r = ((long long)N * HW + 8191) / 8192;
  if (r < 1) r = 1;
  if (r > BN_MAX_ROWS_NCHW) r = BN_MAX_ROWS_NCHW;
  if (r > HW) r = HW;
  return (int)r;
}

// slices of the pixel range per (n, c) plane for the NCHW apply kernels
static int nchw_apply_slices(long long planes, int HW) {
  if (planes >= 2048 || HW <= 4096) return 1;
  long long s = (2048 + planes - 1) / planes;
  long long cap = (HW + 4095) / 4096;
  if (s > cap) s = cap;
  if (s > 65535) s = 65535;
  return (int)(s < 1 ? 1 : s);
}

}  // namespace segk

using namespace segk;

#define BN_COMMON_CHECK(name)                                                                                       \
  SEGK_REQUIRE(N > 0 && C > 0 && HW > 0, name ": bad shape N=%d C=%d HW=%d", N, C, HW);                             \
  SEGK_REQUIRE(layout == SEGK_PM || layout == SEGK_NCHW, name ": bad layout %d", layout);                           \
  SEGK_REQUIRE(layout == SEGK_PM || (long long)N * C < 2147483647LL, name ": too many planes")

extern "C" int segk_bn_stat_rows(int layout, int N, int C, int HW) {
  if (N <= 0 || HW <= 0 || C <= 0) return SEGK_INVALID_ARG;
  return layout == SEGK_PM ? pm_rows((long long)N * HW, C) : nchw_rows(N, HW);
}

extern "C" int segk_bn_stats(const void* x, int dtype, int layout, int N, int C, int HW, int ld, double* partials,
                             segk_stream_t stream) {
  SEGK_REQUIRE(x && partials && valid_dtype(dtype), "bn_stats: null pointer or bad dtype");
  BN_COMMON_CHECK("bn_stats");
  hipStream_t st = (hipStream_t)stream;
  if (layout == SEGK_PM) {
    SEGK_REQUIRE(ld >= C, "bn_stats: ld %d < C %d", ld, C);
    const long long M = (long long)N * HW;
    const int rows = pm_rows(M, C);
    const int vec = pm_vec(C, {ld}, {{x, dtype}}, nullptr);
    const int slab = pm_stat_slab(C);
    SEGK_DISPATCH_DTYPE(dtype, T, SEGK_DISPATCH_VEC(vec, V, (bn_stats_pm_kernel<T, V><<<dim3(rows, ceil_div(C, slab)), BN_THREADS, 0, st>>>(
        (const T*)x, ld, M, C, slab, partials))));
  } else {
    SEGK_REQUIRE(C <= 2147483647, "bn_stats: C too large");
    dim3 grid(C, nchw_rows(N, HW));
    SEGK_DISPATCH_DTYPE(dtype, T, (bn_stats_nchw_kernel<T><<<grid, BN_THREADS, 0, st>>>((const T*)x, N, C, HW, partials)));
  }
  SEGK_CHECK_LAUNCH("bn_stats");
  return SEGK_OK;
}

extern "C" int segk_bn_reduce(const double* partials, int rows, int C, double count, double* sums, float* sums_f32,
                              segk_stream_t stream) {
  SEGK_REQUIRE(partials && sums && rows > 0 && C > 0, "bn_reduce: bad arguments rows=%d C=%d", rows, C);
  bn_reduce_kernel<<<ceil_div(2 * C, 16), 256, 0, (hipStream_t)stream>>>(partials, rows, C, count, sums, sums_f32);
  SEGK_CHECK_LAUNCH("bn_reduce");
  return SEGK_OK;
}

extern "C" int segk_bn_coeffs(const double* sums, const float* gamma, const float* beta, float eps, float momentum,
                              int legacy_clamp, float* running_mean, float* running_var, float* mean, float* inv_std,
                              float* scale, float* shift, int C, segk_stream_t stream) {
  SEGK_REQUIRE(sums && mean && inv_std && scale && shift && C > 0, "bn_coeffs: null pointer or C=%d", C);
  bn_coeffs_kernel<<<ceil_div(C, 256), 256, 0, (hipStream_t)stream>>>(sums, gamma, beta, eps, momentum, legacy_clamp,
                                                                      running_mean, running_var, mean, inv_std, scale,
                                                                      shift, C);
  SEGK_CHECK_LAUNCH("bn_coeffs");
  return SEGK_OK;
}

extern "C" int segk_bn_reduce_coeffs(const double* partials, int rows, int C, double count, const float* gamma,
                                     const float* beta, float eps, float momentum, int legacy_clamp, float* running_mean,
                                     float* running_var, float* mean, float* inv_std, float* scale, float* shift,
                                     double* sums, segk_stream_t stream) {
  SEGK_REQUIRE(partials && mean && inv_std && scale && shift && rows > 0 && C > 0 && count > 0.0,
               "bn_reduce_coeffs: bad arguments rows=%d C=%d", rows, C);
  bn_reduce_coeffs_kernel<<<ceil_div(C, 8), 256, 0, (hipStream_t)stream>>>(partials, rows, C, count, gamma, beta, eps,
                                                                          momentum, legacy_clamp, running_mean,
                                                                          running_var, mean, inv_std, scale, shift, sums);
  SEGK_CHECK_LAUNCH("bn_reduce_coeffs");
  return SEGK_OK;
}

extern "C" int segk_bn_coeffs_eval(const float* running_mean, const float* running_var, const float* gamma,
                                   const float* beta, float eps, float* mean, float* inv_std, float* scale,
                                   float* shift, int C, segk_stream_t stream) {
  SEGK_REQUIRE(running_mean && running_var && mean && inv_std && scale && shift && C > 0,
               "bn_coeffs_eval: null pointer or C=%d", C);
  bn_coeffs_eval_kernel<<<ceil_div(C, 256), 256, 0, (hipStream_t)stream>>>(running_mean, running_var, gamma, beta, eps,
                                                                           mean, inv_std, scale, shift, C);
  SEGK_CHECK_LAUNCH("bn_coeffs_eval");
  return SEGK_OK;
}

extern "C" int segk_bn_apply(const void* x, int x_dtype, int x_ld, void* y, int y_dtype, int y_ld, int layout, int N,
                             int C, int HW, const float* scale, const float* shift, int relu, const uint8_t* mask,
                             float mask_scale, const void* res, int res_ld, uint8_t* relu_bits, int relu_bits_ld,
                             segk_stream_t stream) {
  SEGK_REQUIRE(x && y && scale && shift && valid_dtype(x_dtype) && valid_dtype(y_dtype),
               "bn_apply: null pointer or bad dtype");
  BN_COMMON_CHECK("bn_apply");
  hipStream_t st = (hipStream_t)stream;
  if (layout == SEGK_PM) {
    SEGK_REQUIRE(x_ld >= C && y_ld >= C && (!res || res_ld >= C), "bn_apply: ld (%d,%d,%d) < C %d", x_ld, y_ld, res_ld, C);
    const long long M = (long long)N * HW;
    const int vec = res ? pm_vec(C, {x_ld, y_ld, res_ld}, {{x, x_dtype}, {y, y_dtype}, {res, y_dtype}}, mask)
                        : pm_vec(C, {x_ld, y_ld}, {{x, x_dtype}, {y, y_dtype}}, mask);
    SEGK_REQUIRE(!relu_bits || relu_bits_ld * 8 >= C, "bn_apply: relu_bits_ld %d < C / 8 (C = %d)", relu_bits_ld, C);
    SEGK_UNSUPPORTED(!relu_bits || vec == 8,
                     "bn_apply: the ReLU bit pattern needs C and the leading dimensions to be multiples of 8 and "
                     "16-byte aligned operands");
    static const int wgs = env_int("SEGK_BN_APPLY_WGS", 4096), min_rows = env_int("SEGK_BN_APPLY_MINROWS", 2);
    const dim3 grid = pm_apply_grid(M, C, vec, wgs, min_rows);
#define LAUNCH_APPLY(TX, TY)                                                                                       \
  SEGK_DISPATCH_VEC(vec, V, {                                                                                      \
    if (res)                                                                                                       \
      bn_apply_pm_kernel<TX, TY, V, true><<<grid, BN_THREADS, 0, st>>>((const TX*)x, x_ld, (TY*)y, y_ld, M, C, scale, \
                                                                       shift, relu, mask, mask_scale, (const TY*)res, res_ld, \
                                                                       relu_bits, relu_bits_ld);                       \
    else                                                                                                           \
      bn_apply_pm_kernel<TX, TY, V, false><<<grid, BN_THREADS, 0, st>>>((const TX*)x, x_ld, (TY*)y, y_ld, M, C, scale, \
                                                                        shift, relu, mask, mask_scale, nullptr, 0,  \
                                                                        relu_bits, relu_bits_ld);                      \
  })
    if (x_dtype == SEGK_BF16 && y_dtype == SEGK_BF16) { LAUNCH_APPLY(bf16_t, bf16_t); }
    else if (x_dtype == SEGK_BF16) { LAUNCH_APPLY(bf16_t, float); }
    else if (y_dtype == SEGK_BF16) { LAUNCH_APPLY(float, bf16_t); }
    else { LAUNCH_APPLY(float, float); }
#undef LAUNCH_APPLY
  } else {
    SEGK_REQUIRE(!res && !relu_bits, "bn_apply: the residual input and the ReLU bit pattern need the pixel-major layout");
    const long long planes = (long long)N * C;
    dim3 grid((unsigned)planes, nchw_apply_slices(planes, HW));
#define LAUNCH_APPLY(TX, TY) \
  bn_apply_nchw_kernel<TX, TY><<<grid, BN_THREADS, 0, st>>>((const TX*)x, (TY*)y, C, HW, scale, shift, relu, mask, mask_scale)
    if (x_dtype == SEGK_BF16 && y_dtype == SEGK_BF16) { LAUNCH_APPLY(bf16_t, bf16_t); }
    else if (x_dtype == SEGK_BF16) { LAUNCH_APPLY(bf16_t, float); }
    else if (y_dtype == SEGK_BF16) { LAUNCH_APPLY(float, bf16_t); }
    else { LAUNCH_APPLY(float, float); }
#undef LAUNCH_APPLY
  }
  SEGK_CHECK_LAUNCH("bn_apply");
  return SEGK_OK;
}

extern "C" int segk_bn_bwd_stats(const void* dz, int dz_dtype, int dz_ld, const void* x, int x_dtype, int x_ld,
                                 int layout, int N, int C, int HW, const float* mean, const float* inv_std,
                                 const float* scale, const float* shift, int relu, const uint8_t* mask,
                                 float mask_scale, const void* act, int act_ld, int act_bits, double* partials,
                                 segk_stream_t stream) {
  SEGK_REQUIRE(dz && x && mean && inv_std && scale && shift && partials && valid_dtype(dz_dtype) && valid_dtype(x_dtype),
               "bn_bwd_stats: null pointer or bad dtype");
  BN_COMMON_CHECK("bn_bwd_stats");
  hipStream_t st = (hipStream_t)stream;
  if (layout == SEGK_PM) {
    SEGK_REQUIRE(x_ld >= C && dz_ld >= C && (!act || (act_bits ? act_ld * 8 >= C : act_ld >= C)),
                 "bn_bwd_stats: ld (%d,%d,%d) < C %d", dz_ld, x_ld, act_ld, C);
    SEGK_REQUIRE(!act_bits || act, "bn_bwd_stats: act_bits without act");
    const long long M = (long long)N * HW;
    const int rows = pm_rows(M, C);
    const int vec = (act && !act_bits) ? pm_vec(C, {x_ld, dz_ld, act_ld}, {{x, x_dtype}, {dz, dz_dtype}, {act, dz_dtype}}, mask)
                                       : pm_vec(C, {x_ld, dz_ld}, {{x, x_dtype}, {dz, dz_dtype}}, mask);
    SEGK_UNSUPPORTED(!act_bits || vec == 8, "bn_bwd_stats: the ReLU bit pattern needs 8-channel vector access");
    const int slab = pm_stat_slab(C);
    const size_t lds = (size_t)4 * slab * sizeof(float);
#define LAUNCH_BS(TG, TX)                                                                                           \
  SEGK_DISPATCH_VEC(vec, V, {                                                                                       \
    const dim3 grid_(rows, ceil_div(C, slab));                                                                      \
    if (act && act_bits)                                                                                            \
      bn_bwd_stats_pm_kernel<TG, TX, V, 2><<<grid_, BN_THREADS, lds, st>>>(                                         \
          (const TG*)dz, dz_ld, (const TX*)x, x_ld, M, C, mean, inv_std, scale, shift, relu, mask, mask_scale,      \
          (const TG*)act, act_ld, slab, partials);                                                                  \
    else if (act)                                                                                                   \
      bn_bwd_stats_pm_kernel<TG, TX, V, 1><<<grid_, BN_THREADS, lds, st>>>(                                         \
          (const TG*)dz, dz_ld, (const TX*)x, x_ld, M, C, mean, inv_std, scale, shift, relu, mask, mask_scale,      \
          (const TG*)act, act_ld, slab, partials);                                                                  \
    else                                                                                                            \
      bn_bwd_stats_pm_kernel<TG, TX, V, 0><<<grid_, BN_THREADS, lds, st>>>(                                         \
          (const TG*)dz, dz_ld, (const TX*)x, x_ld, M, C, mean, inv_std, scale, shift, relu, mask, mask_scale,      \
          nullptr, 0, slab, partials);                                                                              \
  })
    if (dz_dtype == SEGK_BF16 && x_dtype == SEGK_BF16) { LAUNCH_BS(bf16_t, bf16_t); }
    else if (dz_dtype == SEGK_BF16) { LAUNCH_BS(bf16_t, float); }
    else if (x_dtype == SEGK_BF16) { LAUNCH_BS(float, bf16_t); }
    else { LAUNCH_BS(float, float); }
#undef LAUNCH_BS
  } else {
    SEGK_REQUIRE(!act, "bn_bwd_stats: a stored activation needs the pixel-major layout");
    dim3 grid(C, nchw_rows(N, HW));
#define LAUNCH_BS(TG, TX)                                                                                        \
  bn_bwd_stats_nchw_kernel<TG, TX><<<grid, BN_THREADS, 0, st>>>((const TG*)dz, (const TX*)x, N, C, HW, mean, inv_std, \
                                                                scale, shift, relu, mask, mask_scale, partials)
    if (dz_dtype == SEGK_BF16 && x_dtype == SEGK_BF16) { LAUNCH_BS(bf16_t, bf16_t); }
    else if (dz_dtype == SEGK_BF16) { LAUNCH_BS(bf16_t, float); }
    else if (x_dtype == SEGK_BF16) { LAUNCH_BS(float, bf16_t); }
    else { LAUNCH_BS(float, float); }
#undef LAUNCH_BS
  }
  SEGK_CHECK_LAUNCH("bn_bwd_stats");
  return SEGK_OK;
}

extern "C" int segk_bn_bwd_apply(const void* dz, int dz_dtype, int dz_ld, const void* x, int x_dtype, int x_ld, void* dx,
                                 int dx_dtype, int dx_ld, int layout, int N, int C, int HW, const float* mean,
                                 const float* inv_std, const float* scale, const float* shift, int relu,
                                 const uint8_t* mask, float mask_scale, const double* gsums, int eval,
                                 const void* act, int act_ld, int act_bits, void* dres, int dres_ld,
                                 segk_stream_t stream) {
  SEGK_REQUIRE(dz && x && dx && mean && inv_std && scale && shift && valid_dtype(dz_dtype) && valid_dtype(x_dtype),
               "bn_bwd_apply: null pointer or bad dtype");
  SEGK_REQUIRE(dx_dtype == dz_dtype, "bn_bwd_apply: dx dtype must equal dz dtype");
  SEGK_REQUIRE(eval || gsums, "bn_bwd_apply: training mode needs the global sums");
  BN_COMMON_CHECK("bn_bwd_apply");
  hipStream_t st = (hipStream_t)stream;
  if (layout == SEGK_PM) {
    SEGK_REQUIRE(x_ld >= C && dz_ld >= C && dx_ld >= C, "bn_bwd_apply: ld (%d,%d,%d) < C %d", dz_ld, x_ld, dx_ld, C);
    const long long M = (long long)N * HW;
    SEGK_REQUIRE(act != nullptr || dres == nullptr, "bn_bwd_apply: a residual gradient needs the stored activation");
    SEGK_REQUIRE((!act || (act_bits ? act_ld * 8 >= C : act_ld >= C)) && (!dres || dres_ld >= C),
                 "bn_bwd_apply: ld (%d,%d) < C %d", act_ld, dres_ld, C);
    SEGK_REQUIRE(!act_bits || act, "bn_bwd_apply: act_bits without act");
    int vec = pm_vec(C, {x_ld, dz_ld, dx_ld}, {{x, x_dtype}, {dz, dz_dtype}, {dx, dx_dtype}}, mask);
    if (act && !act_bits) vec = std::min(vec, pm_vec(C, {act_ld}, {{act, dz_dtype}}, nullptr));
    if (dres) vec = std::min(vec, pm_vec(C, {dres_ld}, {{dres, dz_dtype}}, nullptr));
    SEGK_UNSUPPORTED(!act_bits || vec == 8, "bn_bwd_apply: the ReLU bit pattern needs 8-channel vector access");
    static const int e_slab = env_int("SEGK_BN_BAPPLY_SLAB", 128), wgs = env_int("SEGK_BN_BAPPLY_WGS", 1024),
                     min_rows = env_int("SEGK_BN_BAPPLY_MINROWS", 4);
    const int slab = (vec == 8 && C % e_slab == 0 && e_slab % 8 == 0 && e_slab <= BN_THREADS * vec) ? e_slab : BN_THREADS * vec;
    const dim3 grid = pm_apply_grid(M, C, vec, wgs, min_rows, slab);
    const size_t lds = (size_t)6 * (C < slab ? C : slab) * sizeof(float);
#define LAUNCH_BA(TG, TX)                                                                                            \
  SEGK_DISPATCH_VEC(vec, V, {                                                                                        \
    if (act && act_bits)                                                                                             \
      bn_bwd_apply_pm_kernel<TG, TX, V, 2><<<grid, BN_THREADS, lds, st>>>(                                           \
          (const TG*)dz, dz_ld, (const TX*)x, x_ld, (TG*)dx, dx_ld, M, C, mean, inv_std, scale, shift, relu, mask,   \
          mask_scale, gsums, eval, (const TG*)act, act_ld, (TG*)dres, dres_ld, slab);                                \
    else if (act)                                                                                                    \
      bn_bwd_apply_pm_kernel<TG, TX, V, 1><<<grid, BN_THREADS, lds, st>>>(                                           \
          (const TG*)dz, dz_ld, (const TX*)x, x_ld, (TG*)dx, dx_ld, M, C, mean, inv_std, scale, shift, relu, mask,   \
          mask_scale, gsums, eval, (const TG*)act, act_ld, (TG*)dres, dres_ld, slab);                                \
    else                                                                                                             \
      bn_bwd_apply_pm_kernel<TG, TX, V, 0><<<grid, BN_THREADS, lds, st>>>(                                           \
          (const TG*)dz, dz_ld, (const TX*)x, x_ld, (TG*)dx, dx_ld, M, C, mean, inv_std, scale, shift, relu, mask,   \
          mask_scale, gsums, eval, nullptr, 0, nullptr, 0, slab);                                                    \
  })
    if (dz_dtype == SEGK_BF16 && x_dtype == SEGK_BF16) { LAUNCH_BA(bf16_t, bf16_t); }
    else if (dz_dtype == SEGK_BF16) { LAUNCH_BA(bf16_t, float); }
    else if (x_dtype == SEGK_BF16) { LAUNCH_BA(float, bf16_t); }
    else { LAUNCH_BA(float, float); }
#undef LAUNCH_BA
  } else {
    SEGK_REQUIRE(!act && !dres, "bn_bwd_apply: residual operands need the pixel-major layout");
    const long long planes = (long long)N * C;
    dim3 grid((unsigned)planes, nchw_apply_slices(planes, HW));
#define LAUNCH_BA(TG, TX)                                                                                          \
  bn_bwd_apply_nchw_kernel<TG, TX><<<grid, BN_THREADS, 0, st>>>((const TG*)dz, (const TX*)x, (TG*)dx, C, HW, mean,  \
                                                                inv_std, scale, shift, relu, mask, mask_scale, gsums, \
                                                                eval)
    if (dz_dtype == SEGK_BF16 && x_dtype == SEGK_BF16) { LAUNCH_BA(bf16_t, bf16_t); }
    else if (dz_dtype == SEGK_BF16) { LAUNCH_BA(bf16_t, float); }
    else if (x_dtype == SEGK_BF16) { LAUNCH_BA(float, bf16_t); }
    else { LAUNCH_BA(float, float); }
#undef LAUNCH_BA
  }
  SEGK_CHECK_LAUNCH("bn_bwd_apply");
  return SEGK_OK;
}

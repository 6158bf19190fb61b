// Global average pooling and align-corners bilinear resampling (forward and gather-form backward).
// HBM-bound kernels: every output element is written once, no atomics anywhere.
#include "common.h"

namespace segk {

// ------------------------------------------------------------------ pooling ----
// Block (chunk, n): 64 channel groups of 4 channels x 4 pixel phases.
template <typename T>
__global__ __launch_bounds__(256) void pool_sum_kernel(const T* __restrict__ x, int ld, int HW, int C, float scale,
                                                        float* __restrict__ out) {
  __shared__ float red[4][256];
  const int n = blockIdx.y;
  const int cl = (threadIdx.x & 63) * 4;       // channel offset inside the 256-channel slab
  const int c0 = blockIdx.x * 256 + cl;
  const int ph = threadIdx.x >> 6;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  const T* base = x + (size_t)n * HW * ld;
  if (c0 < C) {
    for (int p = ph; p < HW; p += 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (c0 + j < C) a[j] += Elem<T>::load(base + (size_t)p * ld + c0 + j);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) red[ph][cl + j] = a[j];
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < C) {
    const float s = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    out[(size_t)n * C + c] = s * scale;
  }
}

// --------------------------------------------------------------- resampling ----
// Source position of output index o exactly as the framework computes it for align_corners=True:
// scale = (in-1)/(out-1) in fp32 (0 when out == 1), position = scale*o, lower index by truncation.
struct Tap {
  int i0, i1;
  float w0, w1;
};

__device__ __forceinline__ Tap source_taps(int o, float scale, int in_size) {
  Tap t;
  const float s = scale * (float)o;
  int i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  float l = s - (float)i0;
  l = l < 0.f ? 0.f : (l > 1.f ? 1.f : l);
  t.i0 = i0;
  t.i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  t.w1 = l;
  t.w0 = 1.f - l;
  return t;
}

// weight with which output index o reads input index i
__device__ __forceinline__ float tap_weight(int o, int i, float scale, int in_size) {
  const Tap t = source_taps(o, scale, in_size);
  return (t.i0 == i ? t.w0 : 0.f) + (t.i1 == i ? t.w1 : 0.f);
}

// inclusive range of output indices that can read input index i (conservative; weights decide)
__device__ __forceinline__ void reader_range(int i, float scale, int out_size, int& lo, int& hi) {
  if (scale <= 0.f) {
    lo = 0;
    hi = out_size - 1;
    return;
  }
  const float inv = 1.f / scale;
  lo = (int)floorf(((float)i - 1.f) * inv) - 1;
  hi = (int)ceilf(((float)i + 1.f) * inv) + 1;
  if (lo < 0) lo = 0;
  if (hi > out_size - 1) hi = out_size - 1;
}

static inline float host_scale(int in_size, int out_size) {
  return out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
}

template <typename T>
__device__ __forceinline__ void load4(const T* p, float (&v)[4]);
template <>
__device__ __forceinline__ void load4<float>(const float* p, float (&v)[4]) {
  f32x4 t = *reinterpret_cast<const f32x4*>(p);
  v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
template <>
__device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float (&v)[4]) {
  u32x2 t = *reinterpret_cast<const u32x2*>(p);
  v[0] = __uint_as_float(t[0] << 16); v[1] = __uint_as_float(t[0] & 0xffff0000u);
  v[2] = __uint_as_float(t[1] << 16); v[3] = __uint_as_float(t[1] & 0xffff0000u);
}
template <typename T>
__device__ __forceinline__ void store4(T* p, const float (&v)[4]);
template <>
__device__ __forceinline__ void store4<float>(float* p, const float (&v)[4]) {
  f32x4 t = {v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4*>(p) = t;
}
template <>
__device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float (&v)[4]) {
  u32x2 t = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
  *reinterpret_cast<u32x2*>(p) = t;
}

// V = 4 or 8 consecutive channels of one pixel (8: one 16-byte access for bf16, two for fp32)
template <typename T, int V>
__device__ __forceinline__ void loadv(const T* p, float (&v)[V]);
template <>
__device__ __forceinline__ void loadv<float, 4>(const float* p, float (&v)[4]) { load4<float>(p, v); }
template <>
__device__ __forceinline__ void loadv<bf16_t, 4>(const bf16_t* p, float (&v)[4]) { load4<bf16_t>(p, v); }
template <>
__device__ __forceinline__ void loadv<float, 8>(const float* p, float (&v)[8]) {
  f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
  v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
template <>
__device__ __forceinline__ void loadv<bf16_t, 8>(const bf16_t* p, float (&v)[8]) {
  u32x4 t = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[2 * k] = __uint_as_float(t[k] << 16);
    v[2 * k + 1] = __uint_as_float(t[k] & 0xffff0000u);
  }
}
template <typename T, int V>
__device__ __forceinline__ void storev(T* p, const float (&v)[V]);
template <>
__device__ __forceinline__ void storev<float, 4>(float* p, const float (&v)[4]) { store4<float>(p, v); }
template <>
__device__ __forceinline__ void storev<bf16_t, 4>(bf16_t* p, const float (&v)[4]) { store4<bf16_t>(p, v); }
template <>
__device__ __forceinline__ void storev<float, 8>(float* p, const float (&v)[8]) {
  f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
  *reinterpret_cast<f32x4*>(p) = a;
  *reinterpret_cast<f32x4*>(p + 4) = b;
}
template <>
__device__ __forceinline__ void storev<bf16_t, 8>(bf16_t* p, const float (&v)[8]) {
  u32x4 t = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
  *reinterpret_cast<u32x4*>(p) = t;
}

// Vector form for channel counts that are multiples of 8 with 16-byte aligned rows: block (chunk, n) owns 128
// channels = 16 groups of 8 channels (one 16-byte load for bf16, two for fp32) x 16 pixel phases, four rows in flight
// per thread; twice as many workgroups as the scalar form and 8x fewer load instructions.
template <typename T>
__global__ __launch_bounds__(256) void pool_sum_vec_kernel(const T* __restrict__ x, int ld, int HW, int C, float scale,
                                                            float* __restrict__ out) {
  __shared__ float red[16][129];
  const int n = blockIdx.y;
  const int g = threadIdx.x & 15, ph = threadIdx.x >> 4;
  const int c0 = blockIdx.x * 128 + g * 8;
  float a[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = 0.f;
  if (c0 < C) {
    const T* base = x + (size_t)n * HW * ld + c0;
    int p = ph;
    for (; p + 48 < HW; p += 64) {
      float v[4][8];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        load4<T>(base + (size_t)(p + 16 * k) * ld, *reinterpret_cast<float(*)[4]>(&v[k][0]));
        load4<T>(base + (size_t)(p + 16 * k) * ld + 4, *reinterpret_cast<float(*)[4]>(&v[k][4]));
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += v[k][j];
    }
    for (; p < HW; p += 16) {
      float v[8];
      load4<T>(base + (size_t)p * ld, *reinterpret_cast<float(*)[4]>(&v[0]));
      load4<T>(base + (size_t)p * ld + 4, *reinterpret_cast<float(*)[4]>(&v[4]));
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[ph][g * 8 + j] = a[j];
  __syncthreads();
  if (threadIdx.x < 128) {
    const int c = blockIdx.x * 128 + threadIdx.x;
    if (c < C) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) s += red[q][threadIdx.x];
      out[(size_t)n * C + c] = s * scale;
    }
  }
}

// pixel-major -> pixel-major; thread = (output pixel, group of V channels); C % V == 0.  V = 8 where the operands
// allow 16-byte accesses: the launch is bound by its stores (the source is a sixteenth of the output and stays in
// the cache), and a wavefront then writes 1 KB per instruction instead of 512 bytes.
template <typename TI, typename TO, int V>
__global__ __launch_bounds__(256) void upsample_fwd_pm_kernel(const TI* __restrict__ in, int in_ld, int C, int Hi, int Wi,
                                                               TO* __restrict__ out, int out_ld, int Ho, int Wo,
                                                               float sy, float sx, long long total) {
  const int cg = C / V;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long pix = i / cg;
    const int c0 = (int)(i - pix * cg) * V;
    const int ox = (int)(pix % Wo);
    const long long t = pix / Wo;
    const int oy = (int)(t % Ho);
    const long long n = t / Ho;
    const Tap ty = source_taps(oy, sy, Hi), tx = source_taps(ox, sx, Wi);
    const TI* b = in + (size_t)n * Hi * Wi * in_ld + c0;
    float v00[V], v01[V], v10[V], v11[V], o[V];
    loadv<TI, V>(b + ((size_t)ty.i0 * Wi + tx.i0) * in_ld, v00);
    loadv<TI, V>(b + ((size_t)ty.i0 * Wi + tx.i1) * in_ld, v01);
    loadv<TI, V>(b + ((size_t)ty.i1 * Wi + tx.i0) * in_ld, v10);
    loadv<TI, V>(b + ((size_t)ty.i1 * Wi + tx.i1) * in_ld, v11);
#pragma unroll
    for (int j = 0; j < V; ++j)
      o[j] = ty.w0 * (tx.w0 * v00[j] + tx.w1 * v01[j]) + ty.w1 * (tx.w0 * v10[j] + tx.w1 * v11[j]);
    storev<TO, V>(out + (size_t)pix * out_ld + c0, o);
  }
}

// pixel-major -> NCHW fp32; thread = output pixel (x fastest => coalesced plane writes)
// The bilinear blend with every product and sum rounded on its own (no fused multiply-add), so that the kernels
// that materialise the upsampled logits and those that consume them on the fly see bit-identical values.
__device__ __forceinline__ float lerp4(const Tap& ty, const Tap& tx, float a, float b, float c, float d) {
  const float top = __fadd_rn(__fmul_rn(tx.w0, a), __fmul_rn(tx.w1, b));
  const float bot = __fadd_rn(__fmul_rn(tx.w0, c), __fmul_rn(tx.w1, d));
  return __fadd_rn(__fmul_rn(ty.w0, top), __fmul_rn(ty.w1, bot));
}

template <typename TI>
__global__ __launch_bounds__(256) void upsample_fwd_nchw_kernel(const TI* __restrict__ in, int in_ld, int C, int Hi,
                                                                 int Wi, float* __restrict__ out, int Ho, int Wo,
                                                                 float sy, float sx) {
  const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
  const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int n = blockIdx.z;
  if (ox >= Wo || oy >= Ho) return;
  const Tap ty = source_taps(oy, sy, Hi), tx = source_taps(ox, sx, Wi);
  const TI* b = in + (size_t)n * Hi * Wi * in_ld;
  const TI* p00 = b + ((size_t)ty.i0 * Wi + tx.i0) * in_ld;
  const TI* p01 = b + ((size_t)ty.i0 * Wi + tx.i1) * in_ld;
  const TI* p10 = b + ((size_t)ty.i1 * Wi + tx.i0) * in_ld;
  const TI* p11 = b + ((size_t)ty.i1 * Wi + tx.i1) * in_ld;
  const size_t plane = (size_t)Ho * Wo;
  float* o = out + (size_t)n * C * plane + (size_t)oy * Wo + ox;
  for (int c0 = 0; c0 < C; c0 += 4) {
    float v00[4], v01[4], v10[4], v11[4];
    load4<TI>(p00 + c0, v00);
    load4<TI>(p01 + c0, v01);
    load4<TI>(p10 + c0, v10);
    load4<TI>(p11 + c0, v11);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (c0 + j < C)
        o[(size_t)(c0 + j) * plane] = lerp4(ty, tx, v00[j], v01[j], v10[j], v11[j]);
  }
}

// gradient, pixel-major dout -> pixel-major din; thread = (input pixel, group of V channels).  The output pixels that
// read an input pixel are walked row by row, four of a row at a time: their loads are issued together, whatever
// their weights, and added in ascending order where the weight is not zero (the order of a one-by-one walk).
template <typename TG, int V>
__global__ __launch_bounds__(256) void upsample_bwd_pm_kernel(const TG* __restrict__ dout, int dout_ld, int C, int Ho,
                                                               int Wo, TG* __restrict__ din, int din_ld, int cpad,
                                                               int Hi, int Wi, float sy, float sx, long long total) {
  const int cg = cpad / V;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long pix = i / cg;
    const int c0 = (int)(i - pix * cg) * V;
    const int ix = (int)(pix % Wi);
    const long long t = pix / Wi;
    const int iy = (int)(t % Hi);
    const long long n = t / Hi;
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    if (c0 < C) {
      int ylo, yhi, xlo, xhi;
      reader_range(iy, sy, Ho, ylo, yhi);
      reader_range(ix, sx, Wo, xlo, xhi);
      // the range is conservative: drop the columns at its ends that do not read this pixel (loads, not just weights)
      while (xlo < xhi && tap_weight(xlo, ix, sx, Wi) == 0.f) ++xlo;
      while (xhi > xlo && tap_weight(xhi, ix, sx, Wi) == 0.f) --xhi;
      const TG* b = dout + (size_t)n * Ho * Wo * dout_ld + c0;
      for (int oy = ylo; oy <= yhi; ++oy) {
        const float wy = tap_weight(oy, iy, sy, Hi);
        if (wy == 0.f) continue;
        const TG* row = b + (size_t)oy * Wo * dout_ld;
        for (int ox = xlo; ox <= xhi; ox += 4) {
          float w[4], g[4][V];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const bool in = ox + u <= xhi;
            const int x = in ? ox + u : xhi;
            w[u] = in ? wy * tap_weight(x, ix, sx, Wi) : 0.f;
            loadv<TG, V>(row + (size_t)x * dout_ld, g[u]);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (w[u] != 0.f) {
#pragma unroll
              for (int j = 0; j < V; ++j) acc[j] = fmaf(w[u], g[u][j], acc[j]);
            }
        }
      }
    }
    storev<TG, V>(din + (size_t)pix * din_ld + c0, acc);
  }
}

// gradient, NCHW fp32 dout -> pixel-major din; thread = (input pixel, channel), x fastest
template <typename TG>
__global__ __launch_bounds__(256) void upsample_bwd_nchw_kernel(const float* __restrict__ dout, int C, int Ho, int Wo,
                                                                 TG* __restrict__ din, int din_ld, int cpad, int Hi,
                                                                 int Wi, float sy, float sx) {
  const int ix = blockIdx.x * 64 + (threadIdx.x & 63);
  const int iy = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int n = blockIdx.z / cpad;
  const int c = blockIdx.z % cpad;
  if (ix >= Wi || iy >= Hi) return;
  float acc = 0.f;
  if (c < C) {
    int ylo, yhi, xlo, xhi;
    reader_range(iy, sy, Ho, ylo, yhi);
    reader_range(ix, sx, Wo, xlo, xhi);
    const float* b = dout + ((size_t)n * C + c) * Ho * Wo;
    for (int oy = ylo; oy <= yhi; ++oy) {
      const float wy = tap_weight(oy, iy, sy, Hi);
      if (wy == 0.f) continue;
      for (int ox = xlo; ox <= xhi; ++ox) {
        const float w = wy * tap_weight(ox, ix, sx, Wi);
        if (w != 0.f) acc = fmaf(w, b[(size_t)oy * Wo + ox], acc);
      }
    }
  }
  Elem<TG>::store(din + (((size_t)n * Hi + iy) * Wi + ix) * din_ld + c, acc);
}

// Separable form of the NCHW gradient (any ratio), two passes through an fp32 workspace [N][C][Ho][Wi]:
//   rows:  ws[n][c][oy][ix] = sum_ox wx(ox, ix) * dout[n][c][oy][ox]     reads the gradient once, coalesced
//   cols:  din[n][iy][ix][c] = sum_oy wy(oy, iy) * ws[n][c][oy][ix]      reads the (Wo/Wi times smaller) workspace
// HBM traffic = gradient + 2 x workspace + result, against ~9 x slower gather kernels that re-read through L2.
// gradient rows per workgroup of the row pass: 16, or 4 for rows too wide for 16 of them in 64 KB of LDS

// A thread owns one source column ix for all rows of the workgroup: the tap weights depend on (ox, ix) only, so they
// are computed once per workgroup and column, and each output accumulates over ox in ascending order.
template <int UR_ROWS>
__global__ __launch_bounds__(256) void upsample_bwd_rows_kernel(const float* __restrict__ dout, long long rows, int Wo,
                                                                int Wi, float sx, float* __restrict__ ws) {
  extern __shared__ float rowbuf[];  // [UR_ROWS][Wo]
  const long long r0 = (long long)blockIdx.x * UR_ROWS;
  const int nr = (int)(rows - r0 < UR_ROWS ? rows - r0 : UR_ROWS);
  const float* src = dout + r0 * Wo;
  const int cnt = nr * Wo;
  // four independent loads in flight per thread and trip
  int i = threadIdx.x;
  for (; i + 768 < cnt; i += 1024) {
    const float a = src[i], b = src[i + 256], c = src[i + 512], d = src[i + 768];
    rowbuf[i] = a;
    rowbuf[i + 256] = b;
    rowbuf[i + 512] = c;
    rowbuf[i + 768] = d;
  }
  for (; i < cnt; i += 256) rowbuf[i] = src[i];
  for (int j = cnt + threadIdx.x; j < UR_ROWS * Wo; j += 256) rowbuf[j] = 0.f;   // rows past the end
  __syncthreads();
  for (int ix = threadIdx.x; ix < Wi; ix += 256) {
    int lo, hi;
    reader_range(ix, sx, Wo, lo, hi);
    float a[UR_ROWS];
#pragma unroll
    for (int r = 0; r < UR_ROWS; ++r) a[r] = 0.f;
    for (int ox = lo; ox <= hi; ++ox) {
      const float w = tap_weight(ox, ix, sx, Wi);
      if (w != 0.f) {
#pragma unroll
        for (int r = 0; r < UR_ROWS; ++r) a[r] = fmaf(w, rowbuf[r * Wo + ox], a[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < UR_ROWS; ++r)
      if (r < nr) ws[(r0 + r) * Wi + ix] = a[r];
  }
}

constexpr int UC_PIX = 64, UC_CH = 32;

template <typename TG>
__global__ __launch_bounds__(256) void upsample_bwd_cols_kernel(const float* __restrict__ ws, int C, int Ho, int Hi, int Wi,
                                                                float sy, TG* __restrict__ din, int din_ld, int cpad) {
  __shared__ float outs[UC_PIX][UC_CH + 1];
  const int x0 = blockIdx.x * UC_PIX, iy = blockIdx.y % Hi, n = blockIdx.y / Hi, c0 = blockIdx.z * UC_CH;
  const int lx = threadIdx.x & (UC_PIX - 1), cg = threadIdx.x / UC_PIX;   // 4 channel groups of 8
  int lo, hi;
  reader_range(iy, sy, Ho, lo, hi);
  const int ix = x0 + lx;
  // the eight channels of a thread advance together along oy: eight independent loads per tap row
  constexpr int CPT = UC_CH / 4;
  float a[CPT];
  const float* col[CPT];
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const int c = c0 + cg * CPT + k;
    a[k] = 0.f;
    col[k] = (c < C && ix < Wi) ? ws + (((size_t)n * C + c) * Ho) * Wi + ix : nullptr;
  }
  for (int oy = lo; oy <= hi; ++oy) {
    const float w = tap_weight(oy, iy, sy, Hi);
    if (w == 0.f) continue;
    float v[CPT];
#pragma unroll
    for (int k = 0; k < CPT; ++k) v[k] = col[k] ? col[k][(size_t)oy * Wi] : 0.f;
#pragma unroll
    for (int k = 0; k < CPT; ++k) a[k] = fmaf(w, v[k], a[k]);
  }
#pragma unroll
  for (int k = 0; k < CPT; ++k) outs[lx][cg * CPT + k] = a[k];
  __syncthreads();
  const int nch = min(UC_CH, cpad - c0);
  for (int idx = threadIdx.x; idx < UC_PIX * nch; idx += 256) {
    const int p = idx / nch, c = idx - p * nch;
    if (x0 + p < Wi) Elem<TG>::store(din + (((size_t)n * Hi + iy) * Wi + x0 + p) * din_ld + c0 + c, outs[p][c]);
  }
}

// ---------------------------------------------------------------------------------------------
// nn.MaxPool2d(kernel_size=3, stride=2, padding=1) on pixel-major tensors (the ResNet stem).  Forward keeps the
// position of the maximum inside its window (0..8, one byte per element) instead of a 64-bit flat index; backward is
// a gather: an input pixel belongs to at most four windows (one per coordinate when the coordinate is even, two when
// it is odd) and adds the gradient of those whose stored position points at it, in ascending window order.
// The first maximum in scan order wins and a NaN is a maximum, as in ATen.  thread = (pixel, group of 8 channels).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool3x3s2_fwd_kernel(const T* __restrict__ x, int x_ld, int C, int Hi, int Wi,
                                                                T* __restrict__ y, int y_ld, uint8_t* __restrict__ idx,
                                                                int Ho, int Wo, long long total) {
  const int cg = C >> 3;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long pix = i / cg;
    const int c0 = (int)(i - pix * cg) * 8;
    const int ox = (int)(pix % Wo);
    const long long t = pix / Wo;
    const int oy = (int)(t % Ho);
    const long long n = t / Ho;
    const T* b = x + (size_t)n * Hi * Wi * x_ld + c0;
    float best[8];
    unsigned pos[8];
    bool first = true;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = 2 * oy - 1 + ky;
      if (iy < 0 || iy >= Hi) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = 2 * ox - 1 + kx;
        if (ix < 0 || ix >= Wi) continue;
        float v[8];
        loadv<T, 8>(b + ((size_t)iy * Wi + ix) * x_ld, v);
        const unsigned k = ky * 3 + kx;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          // the first position of the window holds the index until something is greater than -inf (or a NaN)
          if (first) { best[j] = -INFINITY; pos[j] = k; }
          if (v[j] > best[j] || v[j] != v[j]) { best[j] = v[j]; pos[j] = k; }
        }
        first = false;
      }
    }
    storev<T, 8>(y + (size_t)pix * y_ld + c0, best);
    u32x2 packed = {pos[0] | (pos[1] << 8) | (pos[2] << 16) | (pos[3] << 24), pos[4] | (pos[5] << 8) | (pos[6] << 16) | (pos[7] << 24)};
    *reinterpret_cast<u32x2*>(idx + (size_t)pix * C + c0) = packed;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_kernel(const T* __restrict__ dy, int dy_ld, const uint8_t* __restrict__ idx,
                                                                int C, int Hi, int Wi, T* __restrict__ dx, int dx_ld,
                                                                int Ho, int Wo, long long total) {
  const int cg = C >> 3;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long pix = i / cg;
    const int c0 = (int)(i - pix * cg) * 8;
    const int ix = (int)(pix % Wi);
    const long long t = pix / Wi;
    const int iy = (int)(t % Hi);
    const long long n = t / Hi;
    // windows that contain row iy: oy = iy / 2 with ky = 1 for an even row; (iy - 1) / 2 with ky = 2, then
    // (iy + 1) / 2 with ky = 0 for an odd one (ascending oy); the same along x
    const int ny = (iy & 1) ? 2 : 1, nx = (ix & 1) ? 2 : 1;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int a = 0; a < ny; ++a) {
      const int oy = (iy & 1) ? (a == 0 ? (iy - 1) / 2 : (iy + 1) / 2) : iy / 2;
      const int ky = (iy & 1) ? (a == 0 ? 2 : 0) : 1;
      if (oy >= Ho) continue;
      for (int bq = 0; bq < nx; ++bq) {
        const int ox = (ix & 1) ? (bq == 0 ? (ix - 1) / 2 : (ix + 1) / 2) : ix / 2;
        const int kx = (ix & 1) ? (bq == 0 ? 2 : 0) : 1;
        if (ox >= Wo) continue;
        const size_t op = ((size_t)n * Ho + oy) * Wo + ox;
        const u32x2 pk = *reinterpret_cast<const u32x2*>(idx + op * C + c0);
        float g[8];
        loadv<T, 8>(dy + op * dy_ld + c0, g);
        const unsigned k = ky * 3 + kx;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (((pk[j >> 2] >> (8 * (j & 3))) & 0xffu) == k) acc[j] += g[j];
      }
    }
    storev<T, 8>(dx + (size_t)pix * dx_ld + c0, acc);
  }
}

// ---------------------------------------------------------------------------------------------
// Final upsample fused with the cross entropy: the full-resolution logits are never written.
//   forward : thread per output pixel interpolates its K logits from the four source pixels (pixel-major fp32,
//             cache resident: 2 MB per image), log-sum-exp on the fly, NLL of the label.
//   backward: the separable upsample gradient with its first pass computing
//             g[k][ox] = coef * w_t * (exp(logit_k - lse) - [k == t]) instead of loading it.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int ce_label_of(float t) { return (int)(long long)t; }  // .long(): truncation



__global__ __launch_bounds__(256) void upsample_ce_fwd_kernel(const float* __restrict__ low, int ld, int K, int Hi, int Wi,
                                                              const float* __restrict__ target,
                                                              const float* __restrict__ weight, int Ho, int Wo,
                                                              long long P, float sy, float sx, int ignore_index,
                                                              float* __restrict__ lse_out, float* __restrict__ partials) {
  float loss = 0.f, wsum = 0.f;
  const int HWo = Ho * Wo;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < P; i += (long long)gridDim.x * 256) {
    const int n = (int)(i / HWo);
    const int rem = (int)(i - (long long)n * HWo);
    const int oy = rem / Wo, ox = rem - oy * Wo;
    const Tap ty = source_taps(oy, sy, Hi), tx = source_taps(ox, sx, Wi);
    const float* b = low + (size_t)n * Hi * Wi * ld;
    const float* p00 = b + ((size_t)ty.i0 * Wi + tx.i0) * ld;
    const float* p01 = b + ((size_t)ty.i0 * Wi + tx.i1) * ld;
    const float* p10 = b + ((size_t)ty.i1 * Wi + tx.i0) * ld;
    const float* p11 = b + ((size_t)ty.i1 * Wi + tx.i1) * ld;
    const int t = ce_label_of(target[i]);
    float m = -INFINITY, s = 0.f, vt = 0.f;
    for (int k0 = 0; k0 < K; k0 += 4) {
      float a[4], bb[4], c[4], d[4];
      load4<float>(p00 + k0, a);
      load4<float>(p01 + k0, bb);
      load4<float>(p10 + k0, c);
      load4<float>(p11 + k0, d);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (k0 + j < K) {
          const float v = lerp4(ty, tx, a[j], bb[j], c[j], d[j]);
          if (k0 + j == t) vt = v;
          if (v > m) {
            s = s * __expf(m - v) + 1.f;
            m = v;
          } else {
            s += __expf(v - m);
          }
        }
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
  __shared__ float ra[4], rb[4];
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

// np.argmax over the classes of the upsampled logits (train.py:215-217 after deeplab.py:252) without writing them:
// first maximal index, a NaN is maximal.
__global__ __launch_bounds__(256) void upsample_argmax_kernel(const float* __restrict__ low, int ld, int K, int Hi, int Wi,
                                                              int Ho, int Wo, long long P, float sy, float sx,
                                                              int64_t* __restrict__ pred) {
  const int HWo = Ho * Wo;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < P; i += (long long)gridDim.x * 256) {
    const int n = (int)(i / HWo);
    const int rem = (int)(i - (long long)n * HWo);
    const int oy = rem / Wo, ox = rem - oy * Wo;
    const Tap ty = source_taps(oy, sy, Hi), tx = source_taps(ox, sx, Wi);
    const float* b = low + (size_t)n * Hi * Wi * ld;
    const float* p00 = b + ((size_t)ty.i0 * Wi + tx.i0) * ld;
    const float* p01 = b + ((size_t)ty.i0 * Wi + tx.i1) * ld;
    const float* p10 = b + ((size_t)ty.i1 * Wi + tx.i0) * ld;
    const float* p11 = b + ((size_t)ty.i1 * Wi + tx.i1) * ld;
    float best = 0.f;
    int bi = -1;
    for (int k0 = 0; k0 < K; k0 += 4) {
      float a[4], bb[4], c[4], d[4];
      load4<float>(p00 + k0, a);
      load4<float>(p01 + k0, bb);
      load4<float>(p10 + k0, c);
      load4<float>(p11 + k0, d);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (k0 + j < K) {
          const float v = lerp4(ty, tx, a[j], bb[j], c[j], d[j]);
          if (bi < 0 || v > best || (v != v && best == best)) {
            best = v;
            bi = k0 + j;
          }
        }
      }
    }
    pred[i] = bi;
  }
}

// block = (image row (n, oy), chunk of KC classes): the per-pixel quantities of the row (label, coefficient, lse,
// x tap) go to LDS tables once, phase 1 fills rowbuf[kl][ox] with the loss gradient of the row (one item = one
// pixel x four classes, so all threads stay busy whatever the width), phase 2 reduces it along x onto the source
// columns -> ws[n][k][oy][ix]
constexpr int UCE_THREADS = 512;

__global__ __launch_bounds__(UCE_THREADS) void upsample_ce_bwd_rows_kernel(
    const float* __restrict__ low, int ld, int K, int KC, int Hi, int Wi, const float* __restrict__ target,
    const float* __restrict__ weight, const float* __restrict__ lse, int Ho, int Wo, float sy, float sx,
    int ignore_index, const float* __restrict__ coef_dev, float* __restrict__ ws) {
  extern __shared__ float rowbuf[];  // [KC][Wo], then five [Wo] tables
  int* t_i0 = reinterpret_cast<int*>(rowbuf + (size_t)KC * Wo);
  float* t_w1 = reinterpret_cast<float*>(t_i0 + Wo);
  float* t_c = t_w1 + Wo;
  float* t_e = t_c + Wo;
  int* t_t = reinterpret_cast<int*>(t_e + Wo);
  const int n = blockIdx.x / Ho, oy = blockIdx.x - n * Ho;
  const int kbase = blockIdx.y * KC;
  const int kend = min(K, kbase + KC);
  const float coef = *coef_dev;
  const Tap ty = source_taps(oy, sy, Hi);
  const size_t row_i = ((size_t)n * Ho + oy) * Wo;
  for (int ox = threadIdx.x; ox < Wo; ox += UCE_THREADS) {
    const int t = ce_label_of(target[row_i + ox]);
    const bool valid = t != ignore_index && t >= 0 && t < K;
    const Tap tx = source_taps(ox, sx, Wi);
    t_i0[ox] = tx.i0;
    t_w1[ox] = tx.w1;
    t_c[ox] = valid ? coef * (weight ? weight[t] : 1.f) : 0.f;
    t_e[ox] = lse[row_i + ox];
    t_t[ox] = valid ? t : -1;
  }
  __syncthreads();
  const float* r0 = low + ((size_t)n * Hi + ty.i0) * Wi * ld;
  const float* r1 = low + ((size_t)n * Hi + ty.i1) * Wi * ld;
  const int nq = (kend - kbase + 3) >> 2;
  for (int item = threadIdx.x; item < nq * Wo; item += UCE_THREADS) {
    const int q = item / Wo, ox = item - q * Wo;
    const int k0 = kbase + q * 4;
    const int t = t_t[ox];
    const int i0 = t_i0[ox], i1 = i0 + (i0 < Wi - 1 ? 1 : 0);
    Tap tx;
    tx.w1 = t_w1[ox];
    tx.w0 = 1.f - tx.w1;
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    if (t >= 0) {
      const float c = t_c[ox], e = t_e[ox];
      float a[4], bb[4], cc[4], d[4];
      load4<float>(r0 + (size_t)i0 * ld + k0, a);
      load4<float>(r0 + (size_t)i1 * ld + k0, bb);
      load4<float>(r1 + (size_t)i0 * ld + k0, cc);
      load4<float>(r1 + (size_t)i1 * ld + k0, d);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        g[j] = c * (__expf(lerp4(ty, tx, a[j], bb[j], cc[j], d[j]) - e) - (k0 + j == t ? 1.f : 0.f));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (k0 + j < kend) rowbuf[(size_t)(q * 4 + j) * Wo + ox] = g[j];
  }
  __syncthreads();
  const int nk = kend - kbase;
  for (int idx = threadIdx.x; idx < nk * Wi; idx += UCE_THREADS) {
    const int kl = idx / Wi, ix = idx - kl * Wi;
    int lo, hi;
    reader_range(ix, sx, Wo, lo, hi);
    const float* row = rowbuf + (size_t)kl * Wo;
    float a = 0.f;
    for (int ox = lo; ox <= hi; ++ox) {
      const int i0 = t_i0[ox];
      const float l = t_w1[ox];
      const float w = (i0 == ix ? 1.f - l : 0.f) + (i0 + (i0 < Wi - 1 ? 1 : 0) == ix ? l : 0.f);
      if (w != 0.f) a = fmaf(w, row[ox], a);
    }
    ws[(((size_t)n * K + kbase + kl) * Ho + oy) * Wi + ix] = a;
  }
}

static bool aligned_to(const void* p, size_t a) { return ((uintptr_t)p % a) == 0; }

}  // namespace segk

using namespace segk;

extern "C" int segk_pool_sum(const void* x, int dtype, int ld, int N, int HW, int C, float scale, float* out,
                             segk_stream_t stream) {
  SEGK_REQUIRE(x && out && valid_dtype(dtype), "pool_sum: null pointer or bad dtype");
  SEGK_REQUIRE(N > 0 && N <= 65535 && HW > 0 && C > 0 && ld >= C, "pool_sum: bad shape N=%d HW=%d C=%d ld=%d", N, HW, C,
               ld);
  hipStream_t st = (hipStream_t)stream;
  if (C % 8 == 0 && ld % 8 == 0 && aligned_to(x, 16)) {
    dim3 grid(ceil_div(C, 128), N);
    SEGK_DISPATCH_DTYPE(dtype, T, (pool_sum_vec_kernel<T><<<grid, 256, 0, st>>>((const T*)x, ld, HW, C, scale, out)));
  } else {
    dim3 grid(ceil_div(C, 256), N);
    SEGK_DISPATCH_DTYPE(dtype, T, (pool_sum_kernel<T><<<grid, 256, 0, st>>>((const T*)x, ld, HW, C, scale, out)));
  }
  SEGK_CHECK_LAUNCH("pool_sum");
  return SEGK_OK;
}

extern "C" int segk_upsample_fwd(const void* in, int in_dtype, int in_ld, int N, int C, int Hi, int Wi, void* out,
                                 int out_dtype, int out_layout, int out_ld, int Ho, int Wo, segk_stream_t stream) {
  SEGK_REQUIRE(in && out && valid_dtype(in_dtype) && valid_dtype(out_dtype), "upsample_fwd: null pointer or bad dtype");
  SEGK_REQUIRE(N > 0 && C > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "upsample_fwd: bad shape");
  SEGK_REQUIRE(in_ld % 4 == 0 && in_ld >= round_up(C, 4) && aligned_to(in, in_dtype == SEGK_BF16 ? 8 : 16),
               "upsample_fwd: source needs ld %% 4 == 0, ld >= C rounded up to 4, 4-element aligned base (ld=%d C=%d)",
               in_ld, C);
  const float sy = host_scale(Hi, Ho), sx = host_scale(Wi, Wo);
  hipStream_t st = (hipStream_t)stream;
  if (out_layout == SEGK_PM) {
    SEGK_REQUIRE(C % 4 == 0 && out_ld % 4 == 0 && out_ld >= C && aligned_to(out, out_dtype == SEGK_BF16 ? 8 : 16),
                 "upsample_fwd: pixel-major destination needs C %% 4 == 0 and 4-element alignment (C=%d ld=%d)", C,
                 out_ld);
    const bool v8 = C % 8 == 0 && in_ld % 8 == 0 && out_ld % 8 == 0 && aligned_to(in, 16) && aligned_to(out, 16);
    const long long total = (long long)N * Ho * Wo * (C / (v8 ? 8 : 4));
    long long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
#define LAUNCH_UP(TI, TO)                                                                                             \
  do {                                                                                                                \
    if (v8)                                                                                                           \
      upsample_fwd_pm_kernel<TI, TO, 8><<<(int)blocks, 256, 0, st>>>((const TI*)in, in_ld, C, Hi, Wi, (TO*)out, out_ld, \
                                                                     Ho, Wo, sy, sx, total);                          \
    else                                                                                                              \
      upsample_fwd_pm_kernel<TI, TO, 4><<<(int)blocks, 256, 0, st>>>((const TI*)in, in_ld, C, Hi, Wi, (TO*)out, out_ld, \
                                                                     Ho, Wo, sy, sx, total);                          \
  } while (0)
    if (in_dtype == SEGK_BF16 && out_dtype == SEGK_BF16) { LAUNCH_UP(bf16_t, bf16_t); }
    else if (in_dtype == SEGK_BF16) { LAUNCH_UP(bf16_t, float); }
    else if (out_dtype == SEGK_BF16) { LAUNCH_UP(float, bf16_t); }
    else { LAUNCH_UP(float, float); }
#undef LAUNCH_UP
  } else if (out_layout == SEGK_NCHW) {
    SEGK_REQUIRE(out_dtype == SEGK_F32, "upsample_fwd: the NCHW destination is fp32");
    SEGK_REQUIRE(N <= 65535 && ceil_div(Ho, 4) <= 65535, "upsample_fwd: grid too large");
    dim3 grid(ceil_div(Wo, 64), ceil_div(Ho, 4), N);
    SEGK_DISPATCH_DTYPE(in_dtype, T, (upsample_fwd_nchw_kernel<T><<<grid, 256, 0, st>>>((const T*)in, in_ld, C, Hi, Wi,
                                                                                        (float*)out, Ho, Wo, sy, sx)));
  } else {
    SEGK_REQUIRE(false, "upsample_fwd: bad layout %d", out_layout);
  }
  SEGK_CHECK_LAUNCH("upsample_fwd");
  return SEGK_OK;
}

extern "C" long long segk_upsample_bwd_workspace(int dout_layout, int N, int C, int Ho, int Wi) {
  if (dout_layout != SEGK_NCHW || N <= 0 || C <= 0 || Ho <= 0 || Wi <= 0) return 0;
  return (long long)N * C * Ho * Wi;
}

extern "C" int segk_upsample_bwd(const void* dout, int dout_dtype, int dout_layout, int dout_ld, int N, int C, int Ho,
                                 int Wo, void* din, int din_dtype, int din_ld, int din_cpad, int Hi, int Wi,
                                 float* workspace, segk_stream_t stream) {
  SEGK_REQUIRE(dout && din && valid_dtype(dout_dtype) && valid_dtype(din_dtype), "upsample_bwd: null pointer or bad dtype");
  SEGK_REQUIRE(N > 0 && C > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && din_cpad >= C && din_ld >= din_cpad,
               "upsample_bwd: bad shape C=%d cpad=%d ld=%d", C, din_cpad, din_ld);
  const float sy = host_scale(Hi, Ho), sx = host_scale(Wi, Wo);
  hipStream_t st = (hipStream_t)stream;
  if (dout_layout == SEGK_PM) {
    SEGK_REQUIRE(dout_dtype == din_dtype, "upsample_bwd: pixel-major gradient keeps its dtype");
    SEGK_REQUIRE(C % 4 == 0 && din_cpad % 4 == 0 && dout_ld % 4 == 0 && din_ld % 4 == 0 && dout_ld >= C &&
                     aligned_to(dout, dout_dtype == SEGK_BF16 ? 8 : 16) && aligned_to(din, din_dtype == SEGK_BF16 ? 8 : 16),
                 "upsample_bwd: pixel-major form needs C %% 4 == 0 and 4-element alignment");
    const bool v8 = C % 8 == 0 && din_cpad % 8 == 0 && dout_ld % 8 == 0 && din_ld % 8 == 0 && aligned_to(dout, 16) &&
                    aligned_to(din, 16);
    const long long total = (long long)N * Hi * Wi * (din_cpad / (v8 ? 8 : 4));
    long long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (v8) {
      SEGK_DISPATCH_DTYPE(dout_dtype, T,
                          (upsample_bwd_pm_kernel<T, 8><<<(int)blocks, 256, 0, st>>>((const T*)dout, dout_ld, C, Ho, Wo, (T*)din,
                                                                                    din_ld, din_cpad, Hi, Wi, sy, sx, total)));
    } else {
      SEGK_DISPATCH_DTYPE(dout_dtype, T,
                          (upsample_bwd_pm_kernel<T, 4><<<(int)blocks, 256, 0, st>>>((const T*)dout, dout_ld, C, Ho, Wo, (T*)din,
                                                                                    din_ld, din_cpad, Hi, Wi, sy, sx, total)));
    }
  } else if (dout_layout == SEGK_NCHW) {
    SEGK_REQUIRE(dout_dtype == SEGK_F32, "upsample_bwd: the NCHW gradient is fp32");
    const long long rows = (long long)N * C * Ho;
    if (workspace && (long long)Wo * 4 * 4 <= 65536 && ceil_div(rows, 4) < 2147483647LL &&
        (long long)N * Hi <= 65535 && ceil_div(din_cpad, UC_CH) <= 65535) {
      if ((long long)Wo * 16 * 4 <= 65536)
        upsample_bwd_rows_kernel<16><<<ceil_div(rows, 16), 256, (size_t)16 * Wo * sizeof(float), st>>>(
            (const float*)dout, rows, Wo, Wi, sx, workspace);
      else
        upsample_bwd_rows_kernel<4><<<ceil_div(rows, 4), 256, (size_t)4 * Wo * sizeof(float), st>>>(
            (const float*)dout, rows, Wo, Wi, sx, workspace);
      SEGK_CHECK_LAUNCH("upsample_bwd rows");
      dim3 grid(ceil_div(Wi, UC_PIX), N * Hi, ceil_div(din_cpad, UC_CH));
      SEGK_DISPATCH_DTYPE(din_dtype, T, (upsample_bwd_cols_kernel<T><<<grid, 256, 0, st>>>(workspace, C, Ho, Hi, Wi, sy, (T*)din,
                                                                                         din_ld, din_cpad)));
    } else {
      SEGK_REQUIRE((long long)N * din_cpad <= 65535 && ceil_div(Hi, 4) <= 65535, "upsample_bwd: grid too large");
      dim3 grid(ceil_div(Wi, 64), ceil_div(Hi, 4), N * din_cpad);
      SEGK_DISPATCH_DTYPE(din_dtype, T, (upsample_bwd_nchw_kernel<T><<<grid, 256, 0, st>>>((const float*)dout, C, Ho,
                                                                                           Wo, (T*)din, din_ld, din_cpad,
                                                                                           Hi, Wi, sy, sx)));
    }
  } else {
    SEGK_REQUIRE(false, "upsample_bwd: bad layout %d", dout_layout);
  }
  SEGK_CHECK_LAUNCH("upsample_bwd");
  return SEGK_OK;
}

extern "C" int segk_upsample_ce_fwd(const float* low, int low_ld, int N, int K, int Hi, int Wi, const float* target,
                                    const float* weight, int Ho, int Wo, int ignore_index, float* lse, float* partials,
                                    float* sums, segk_stream_t stream) {
  SEGK_REQUIRE(low && target && lse && partials && sums, "upsample_ce_fwd: null pointer");
  SEGK_REQUIRE(N > 0 && K > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "upsample_ce_fwd: bad shape");
  SEGK_REQUIRE(low_ld % 4 == 0 && low_ld >= round_up(K, 4) && aligned_to(low, 16),
               "upsample_ce_fwd: logits need a 16-byte aligned base and ld %% 4 == 0, ld >= K rounded up to 4 (ld=%d K=%d)",
               low_ld, K);
  SEGK_REQUIRE((long long)Ho * Wo < 2147483647LL, "upsample_ce_fwd: image too large");
  const long long P = (long long)N * Ho * Wo;
  const int rows = ce_partial_rows(P);
  hipStream_t st = (hipStream_t)stream;
  upsample_ce_fwd_kernel<<<rows, 256, 0, st>>>(low, low_ld, K, Hi, Wi, target, weight, Ho, Wo, P, host_scale(Hi, Ho),
                                               host_scale(Wi, Wo), ignore_index, lse, partials);
  SEGK_CHECK_LAUNCH("upsample_ce_fwd");
  return launch_ce_finish(partials, rows, sums, st);
}

extern "C" int segk_upsample_ce_bwd(const float* low, int low_ld, int N, int K, int Hi, int Wi, const float* target,
                                    const float* weight, const float* lse, int Ho, int Wo, int ignore_index,
                                    const float* coef_dev, float* workspace, float* dlow, int dlow_ld, int dlow_cpad,
                                    segk_stream_t stream) {
  SEGK_REQUIRE(low && target && lse && coef_dev && workspace && dlow, "upsample_ce_bwd: null pointer");
  SEGK_REQUIRE(N > 0 && K > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && dlow_cpad >= K && dlow_ld >= dlow_cpad,
               "upsample_ce_bwd: bad shape K=%d cpad=%d ld=%d", K, dlow_cpad, dlow_ld);
  SEGK_REQUIRE(low_ld % 4 == 0 && low_ld >= round_up(K, 4) && aligned_to(low, 16),
               "upsample_ce_bwd: logits need a 16-byte aligned base and ld %% 4 == 0, ld >= K rounded up to 4 (ld=%d K=%d)",
               low_ld, K);
  int KC = K < 32 ? round_up(K, 4) : 32;   // classes per workgroup: KC + 5 rows of Wo floats must fit in LDS
  while (KC > 4 && (long long)(KC + 5) * Wo * 4 > 131072) KC = (KC / 2 + 3) & ~3;
  SEGK_UNSUPPORTED((long long)(KC + 5) * Wo * 4 <= 131072, "upsample_ce_bwd: output width %d too large", Wo);
  SEGK_REQUIRE((long long)N * Ho < 2147483647LL && (long long)N * Hi <= 65535 && ceil_div(dlow_cpad, UC_CH) <= 65535 &&
                   ceil_div(K, KC) <= 65535,
               "upsample_ce_bwd: grid too large");
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)(KC + 5) * Wo * sizeof(float);
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute((const void*)upsample_ce_bwd_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072) !=
        hipSuccess) {
      set_error("upsample_ce_bwd: cannot raise the dynamic LDS limit");
      return SEGK_HIP_ERROR;
    }
    configured = true;
  }
  const float sy = host_scale(Hi, Ho), sx = host_scale(Wi, Wo);
  upsample_ce_bwd_rows_kernel<<<dim3(N * Ho, ceil_div(K, KC)), UCE_THREADS, lds, st>>>(low, low_ld, K, KC, Hi, Wi, target, weight, lse,
                                                                            Ho, Wo, sy, sx, ignore_index, coef_dev,
                                                                            workspace);
  SEGK_CHECK_LAUNCH("upsample_ce_bwd rows");
  dim3 grid(ceil_div(Wi, UC_PIX), N * Hi, ceil_div(dlow_cpad, UC_CH));
  upsample_bwd_cols_kernel<float><<<grid, 256, 0, st>>>(workspace, K, Ho, Hi, Wi, sy, dlow, dlow_ld, dlow_cpad);
  SEGK_CHECK_LAUNCH("upsample_ce_bwd cols");
  return SEGK_OK;
}

extern "C" int segk_upsample_argmax(const float* low, int low_ld, int N, int K, int Hi, int Wi, int Ho, int Wo,
                                    int64_t* pred, segk_stream_t stream) {
  SEGK_REQUIRE(low && pred, "upsample_argmax: null pointer");
  SEGK_REQUIRE(N > 0 && K > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "upsample_argmax: bad shape");
  SEGK_REQUIRE(low_ld % 4 == 0 && low_ld >= round_up(K, 4) && aligned_to(low, 16),
               "upsample_argmax: logits need a 16-byte aligned base and ld %% 4 == 0, ld >= K rounded up to 4 (ld=%d K=%d)",
               low_ld, K);
  SEGK_REQUIRE((long long)Ho * Wo < 2147483647LL, "upsample_argmax: image too large");
  const long long P = (long long)N * Ho * Wo;
  long long blocks = (P + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  upsample_argmax_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>(low, low_ld, K, Hi, Wi, Ho, Wo, P, host_scale(Hi, Ho),
                                                                      host_scale(Wi, Wo), pred);
  SEGK_CHECK_LAUNCH("upsample_argmax");
  return SEGK_OK;
}

static int maxpool_check(const char* name, int N, int C, int Hi, int Wi, std::initializer_list<const void*> ptrs,
                         std::initializer_list<int> lds, int dtype) {
  SEGK_REQUIRE(N > 0 && C > 0 && Hi > 0 && Wi > 0 && valid_dtype(dtype), "%s: bad shape or dtype", name);
  bool ok = C % 8 == 0;
  for (const void* p : ptrs) ok = ok && p != nullptr && aligned_to(p, 16);
  for (int ld : lds) ok = ok && ld % 8 == 0 && ld >= C;
  SEGK_UNSUPPORTED(ok, "%s: needs C %% 8 == 0 (C=%d), 16-byte aligned operands and leading dimensions that are multiples of 8",
                   name, C);
  return SEGK_OK;
}

extern "C" int segk_maxpool3x3s2_fwd(const void* x, int dtype, int x_ld, int N, int C, int Hi, int Wi, void* y, int y_ld,
                                     uint8_t* idx, segk_stream_t stream) {
  int rc = maxpool_check("maxpool_fwd", N, C, Hi, Wi, {x, y, idx}, {x_ld, y_ld}, dtype);
  if (rc != SEGK_OK) return rc;
  const int Ho = (Hi - 1) / 2 + 1, Wo = (Wi - 1) / 2 + 1;
  const long long total = (long long)N * Ho * Wo * (C / 8);
  long long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  SEGK_DISPATCH_DTYPE(dtype, T, (maxpool3x3s2_fwd_kernel<T><<<(int)blocks, 256, 0, (hipStream_t)stream>>>(
      (const T*)x, x_ld, C, Hi, Wi, (T*)y, y_ld, idx, Ho, Wo, total)));
  SEGK_CHECK_LAUNCH("maxpool_fwd");
  return SEGK_OK;
}

extern "C" int segk_maxpool3x3s2_bwd(const void* dy, int dtype, int dy_ld, const uint8_t* idx, int N, int C, int Hi, int Wi,
                                     void* dx, int dx_ld, segk_stream_t stream) {
  int rc = maxpool_check("maxpool_bwd", N, C, Hi, Wi, {dy, dx, idx}, {dy_ld, dx_ld}, dtype);
  if (rc != SEGK_OK) return rc;
  const int Ho = (Hi - 1) / 2 + 1, Wo = (Wi - 1) / 2 + 1;
  const long long total = (long long)N * Hi * Wi * (C / 8);
  long long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  SEGK_DISPATCH_DTYPE(dtype, T, (maxpool3x3s2_bwd_kernel<T><<<(int)blocks, 256, 0, (hipStream_t)stream>>>(
      (const T*)dy, dy_ld, idx, C, Hi, Wi, (T*)dx, dx_ld, Ho, Wo, total)));
  SEGK_CHECK_LAUNCH("maxpool_bwd");
  return SEGK_OK;
}

// Layout changes at the boundary of the head: NCHW <-> pixel-major, weight packing, weight-gradient
// finishing.  All HBM-bound; each element is read once and written once.
#include "common.h"

namespace segk {

// One block transposes a 32(channel) x 64(pixel) tile of one image through LDS so that both the NCHW
// side (pixels contiguous) and the pixel-major side (channels contiguous) are accessed in runs.
template <typename TD>
__global__ __launch_bounds__(256) void nchw_to_pm_kernel(const float* __restrict__ src, TD* __restrict__ dst,
                                                          int C, int HW, int ld) {
  __shared__ float tile[32][65];
  const int n = blockIdx.z;
  const int c0 = blockIdx.y * 32;
  const int p0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63;
  const int ty = threadIdx.x >> 6;  // 0..3
  const float* s = src + (size_t)n * C * HW;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int c = c0 + ty + i * 4;
    int p = p0 + tx;
    tile[ty + i * 4][tx] = (c < C && p < HW) ? s[(size_t)c * HW + p] : 0.f;
  }
  __syncthreads();
  const int cx = threadIdx.x & 31;
  const int py = threadIdx.x >> 5;  // 0..7
  TD* d = dst + (size_t)n * HW * ld;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int p = p0 + py + i * 8;
    int c = c0 + cx;
    if (p < HW && c < C) Elem<TD>::store(d + (size_t)p * ld + c, tile[cx][py + i * 8]);
  }
}

template <typename TS>
__global__ __launch_bounds__(256) void pm_to_nchw_kernel(const TS* __restrict__ src, float* __restrict__ dst,
                                                          int C, int HW, int ld) {
  __shared__ float tile[64][33];
  const int n = blockIdx.z;
  const int c0 = blockIdx.y * 32;
  const int p0 = blockIdx.x * 64;
  const int cx = threadIdx.x & 31;
  const int py = threadIdx.x >> 5;
  const TS* s = src + (size_t)n * HW * ld;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int p = p0 + py + i * 8;
    int c = c0 + cx;
    tile[py + i * 8][cx] = (p < HW && c < C) ? Elem<TS>::load(s + (size_t)p * ld + c) : 0.f;
  }
  __syncthreads();
  const int tx = threadIdx.x & 63;
  const int ty = threadIdx.x >> 6;
  float* d = dst + (size_t)n * C * HW;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int c = c0 + ty + i * 4;
    int p = p0 + tx;
    if (c < C && p < HW) d[(size_t)c * HW + p] = tile[tx][ty + i * 4];
  }
}

// dst[tapd][row][k]: transpose == 0: row = co, k = ci, tapd = tap; transpose == 1: row = ci, k = co,
// tapd = taps-1-tap.  k runs to Ck (multiple of 32); the tail is zero.
// The source is any dense layout of the [Cout][Cin][taps] weight given by three element strides (OIHW: s_tap = 1,
// channels_last: s_ci = 1).  One block moves a PCO(co) x 32(ci) patch of TPB taps through LDS: it reads in the
// source's own order and writes runs along k.  TPB = all taps for a tap-fastest source (runs of 32*taps elements
// per output channel), 1 for a channel-fastest source (blockIdx.z = tap; runs of 32 elements).
template <typename TS, typename TD, int TPB, int PCO>
__device__ __forceinline__ void pack_patch(const TS* __restrict__ w, long long s_co, long long s_ci, long long s_tap,
                                           TD* __restrict__ dst, int Cout, int Cin, int taps, int rows, int Ck,
                                           int transpose, int bx, int by, int bz, float (*tile)[PCO][33]) {
  const int co0 = bx * PCO, ci0 = by * 32;
  const int tap0 = TPB == 1 ? bz : 0;
  constexpr int PATCH = PCO * 32 * TPB;
  for (int i = threadIdx.x; i < PATCH; i += 256) {
    int co_l, ci_l, tap;
    if (TPB > 1) {               // (co, ci, tap): tap then ci contiguous in the source
      co_l = i / (32 * TPB);
      const int rem = i - co_l * (32 * TPB);
      ci_l = rem / TPB;
      tap = rem - ci_l * TPB;
    } else {                     // (co, ci): ci contiguous in the source
      co_l = i >> 5;
      ci_l = i & 31;
      tap = 0;
    }
    const int co = co0 + co_l, ci = ci0 + ci_l;
    float v = 0.f;
    if (co < Cout && ci < Cin) v = Elem<TS>::load(w + co * s_co + ci * s_ci + (tap0 + tap) * s_tap);
    tile[tap][co_l][ci_l] = v;
  }
  __syncthreads();
  if (!transpose) {              // rows = co, k = ci: runs of 32 along ci
    for (int i = threadIdx.x; i < PATCH; i += 256) {
      const int k_l = i & 31, r_l = (i >> 5) % PCO, tap = i / (32 * PCO);
      const int row = co0 + r_l, k = ci0 + k_l;
      if (row < rows && k < Ck) Elem<TD>::store(dst + ((size_t)(tap0 + tap) * rows + row) * Ck + k, tile[tap][r_l][k_l]);
    }
  } else {                       // rows = ci, k = co, taps reversed: runs of PCO along co
    for (int i = threadIdx.x; i < PATCH; i += 256) {
      const int k_l = i % PCO, r_l = (i / PCO) & 31, tap = i / (32 * PCO);
      const int row = ci0 + r_l, k = co0 + k_l;
      if (row < rows && k < Ck)
        Elem<TD>::store(dst + ((size_t)(taps - 1 - tap0 - tap) * rows + row) * Ck + k, tile[tap][k_l][r_l]);
    }
  }
}

template <typename TS, typename TD, int TPB, int PCO>
__global__ __launch_bounds__(256) void pack_weight_kernel(const TS* __restrict__ w, long long s_co, long long s_ci,
                                                           long long s_tap, TD* __restrict__ dst, int Cout, int Cin,
                                                           int taps, int rows, int Ck, int transpose) {
  __shared__ float tile[TPB][PCO][33];
  pack_patch<TS, TD, TPB, PCO>(w, s_co, s_ci, s_tap, dst, Cout, Cin, taps, rows, Ck, transpose, blockIdx.x, blockIdx.y,
                               blockIdx.z, tile);
}

// workgroup grid of one weight's pack: nine taps of a 16 x 32 patch per workgroup for a tap-fastest 3x3 source, else
// one tap of a 32 x 32 patch
__host__ __device__ inline void pack_grid(int Cout, int Cin, int taps, long long s_tap, int transpose, int& gx, int& gy,
                                          int& gz, bool& all_taps) {
  const int kdim = transpose ? Cout : Cin;
  const int Ck = (kdim + 31) / 32 * 32;
  const int co_ext = transpose ? Ck : Cout, ci_ext = transpose ? Cin : Ck;
  all_taps = taps == 9 && s_tap == 1;
  gx = (co_ext + (all_taps ? 15 : 31)) / (all_taps ? 16 : 32);
  gy = (ci_ext + 31) / 32;
  gz = all_taps ? 1 : taps;
}

// Many weights in one launch: workgroup b serves the item whose [first_block, next first_block) range holds b and
// moves the same patch, by the same code, as the single-weight kernel would.
__global__ __launch_bounds__(256) void pack_weights_kernel(const segk_pack_item* __restrict__ items, int n) {
  __shared__ float tile_raw[9 * 16 * 33];
  const int b = blockIdx.x;
  int lo = 0, hi = n - 1;
  while (lo < hi) {              // last item with first_block <= b
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].first_block <= b) lo = mid;
    else hi = mid - 1;
  }
  const segk_pack_item it = items[lo];
  const int taps = it.R * it.R;
  int gx, gy, gz;
  bool all_taps;
  pack_grid(it.Cout, it.Cin, taps, it.s_tap, it.transpose, gx, gy, gz, all_taps);
  const int local = b - it.first_block;
  if (local >= gx * gy * gz) return;   // (a table whose block counts were not made by segk_pack_weight_blocks)
  const int bx = local % gx, by = (local / gx) % gy, bz = local / (gx * gy);
  const int rows = it.transpose ? it.Cin : it.Cout;
  const int Ck = ((it.transpose ? it.Cout : it.Cin) + 31) / 32 * 32;
#define SEGK_PACK(TS, TD)                                                                                             \
  do {                                                                                                                \
    if (all_taps)                                                                                                     \
      pack_patch<TS, TD, 9, 16>((const TS*)it.w, it.s_co, it.s_ci, it.s_tap, (TD*)it.dst, it.Cout, it.Cin, taps, rows, \
                                Ck, it.transpose, bx, by, bz, reinterpret_cast<float (*)[16][33]>(tile_raw));          \
    else                                                                                                              \
      pack_patch<TS, TD, 1, 32>((const TS*)it.w, it.s_co, it.s_ci, it.s_tap, (TD*)it.dst, it.Cout, it.Cin, taps, rows, \
                                Ck, it.transpose, bx, by, bz, reinterpret_cast<float (*)[32][33]>(tile_raw));          \
  } while (0)
  if (it.w_dtype == SEGK_BF16) {
    if (it.dst_dtype == SEGK_BF16) SEGK_PACK(bf16_t, bf16_t);
    else SEGK_PACK(bf16_t, float);
  } else {
    if (it.dst_dtype == SEGK_BF16) SEGK_PACK(float, bf16_t);
    else SEGK_PACK(float, float);
  }
#undef SEGK_PACK
}

// the gradient tensors of the convolutions of one launch (one for a single weight gradient)
struct FinishGroup {
  void* dw[SEGK_WGRAD_GROUP_MAX];
};

// dw[g][co][ci][tap] = sum_s slabs[g][s][tap][ci][co]; blockIdx.z = g * taps + tap
template <typename TD>
__global__ void wgrad_finish_kernel(const float* __restrict__ slabs, int splits, int taps, int Cin, int Cout,
                                    const FinishGroup group, long long s_co, long long s_ci, long long s_tap,
                                    int Cout_true, int Cin_true) {
  // tile transpose over (ci, co) for one tap: reads run along co, writes along (ci, tap)
  __shared__ float tile[32][33];
  const int gi = blockIdx.z / taps, tap = blockIdx.z - gi * taps;
  TD* __restrict__ dw = (TD*)group.dw[gi];
  slabs += (size_t)gi * splits * taps * Cin * Cout;
  const int co0 = blockIdx.x * 32, ci0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const size_t slab = (size_t)taps * Cin * Cout;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int ci = ci0 + ty + i * 8, co = co0 + tx;
    float acc = 0.f;
    if (ci < Cin && co < Cout) {
      const float* p = slabs + ((size_t)tap * Cin + ci) * Cout + co;
      // sixteen loads in flight, added in slab order (parts past the end add 0.0): one round trip to memory for
      // up to sixteen parts - the launch is latency, not bandwidth, for the part counts the plans choose
      for (int s = 0; s < splits; s += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = s + u < splits ? p[(size_t)(s + u) * slab] : 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += v[u];
      }
    }
    tile[ty + i * 8][tx] = acc;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int co = co0 + ty + i * 8, ci = ci0 + tx;
    if (co < Cout_true && ci < Cin_true) Elem<TD>::store(dw + co * s_co + ci * s_ci + tap * s_tap, tile[tx][ty + i * 8]);
  }
}

// The same sum for a 3 x 3 gradient whose taps are the innermost index of dw (nn.Conv2d's own [co][ci][3][3] layout):
// a workgroup owns 32 co x 32 ci of ALL nine taps, so that each co row of its output is one run of 288 consecutive
// elements.  (One tap per workgroup writes every ninth element: nine visits to each memory line, by workgroups far
// apart in time - the head's 3 x 3 gradients took 0.4 ms that way.)  Same order of additions as above.
template <typename TD>
__global__ __launch_bounds__(256) void wgrad_finish_taps_kernel(const float* __restrict__ slabs, int splits, int Cin,
                                                                int Cout, const FinishGroup group, long long s_co,
                                                                int Cout_true, int Cin_true) {
  constexpr int TAPS = 9;
  __shared__ float tile[TAPS][32][33];
  const int gi = blockIdx.z;
  TD* __restrict__ dw = (TD*)group.dw[gi];
  const size_t slab = (size_t)TAPS * Cin * Cout;
  slabs += (size_t)gi * splits * slab;
  const int co0 = blockIdx.x * 32, ci0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  // all 36 loads of a part are issued before its additions: with one or two parts (the head's 3 x 3 gradients over
  // 2048 input channels) a walk tap by tap would leave one load in flight per thread
  float acc[TAPS][4];
#pragma unroll
  for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[tap][i] = 0.f;
  const int co = co0 + tx;
  for (int sp = 0; sp < splits; ++sp) {
    const float* p = slabs + (size_t)sp * slab + co;
    float v[TAPS][4];
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ci = ci0 + ty + i * 8;
        v[tap][i] = (ci < Cin && co < Cout) ? p[((size_t)tap * Cin + ci) * Cout] : 0.f;
      }
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[tap][i] += v[tap][i];
  }
#pragma unroll
  for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
    for (int i = 0; i < 4; ++i) tile[tap][ty + i * 8][tx] = acc[tap][i];
  __syncthreads();
  // row co: elements (ci0 .. ci0 + 31) x 9 taps are consecutive in dw
  for (int idx = threadIdx.x; idx < 32 * 32 * TAPS; idx += 256) {
    const int co_l = idx / (32 * TAPS), rem = idx - co_l * (32 * TAPS);
    const int ci_l = rem / TAPS, tap = rem - ci_l * TAPS;
    const int co = co0 + co_l, ci = ci0 + ci_l;
    if (co < Cout_true && ci < Cin_true) Elem<TD>::store(dw + co * s_co + (long long)ci * TAPS + tap, tile[tap][ci_l][co_l]);
  }
}

// The same sum for small gradients cut into many parts (the 1x1 convolutions onto 48 or 21 channels at 129 x 129:
// a 64 KB slab, hundreds of parts so that the multiplication fills the chip).  A workgroup owns 32 co x 8 ci; its
// eight thread rows each add every eighth part (parts r, r + 8, ... in ascending order), then the eight row sums are
// added in row order: a fixed order, and eight times the loads in flight of the serial walk above.
template <typename TD>
__global__ __launch_bounds__(256) void wgrad_finish_tree_kernel(const float* __restrict__ slabs, int splits, int taps,
                                                                int Cin, int Cout, const FinishGroup group, long long s_co,
                                                                long long s_ci, long long s_tap, int Cout_true,
                                                                int Cin_true) {
  __shared__ float part[8][8][33];
  const int gi = blockIdx.z / taps, tap = blockIdx.z - gi * taps;
  TD* __restrict__ dw = (TD*)group.dw[gi];
  slabs += (size_t)gi * splits * taps * Cin * Cout;
  const int co0 = blockIdx.x * 32, ci0 = blockIdx.y * 8;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const size_t slab = (size_t)taps * Cin * Cout;
  const int co = co0 + tx;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  if (co < Cout) {
    // four parts of this thread row per trip: 32 loads in flight, added in part order
    for (int sp = ty; sp < splits; sp += 32) {
      float v[4][8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool in = sp + 8 * q < splits;
        const float* p = slabs + (size_t)(in ? sp + 8 * q : sp) * slab + ((size_t)tap * Cin + ci0) * Cout + co;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[q][i] = in && ci0 + i < Cin ? p[(size_t)i * Cout] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += v[q][i];
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) part[ty][i][tx] = acc[i];
  __syncthreads();
  const int ci = ci0 + ty;   // thread row ty now finishes input channel ci0 + ty
  if (co < Cout_true && ci < Cin_true) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) t += part[r][ty][tx];
    Elem<TD>::store(dw + co * s_co + ci * s_ci + tap * s_tap, t);
  }
}

// ---------------------------------------------------------------------------------------------
// Dropout keep-masks: one byte per element, 1 with probability 1 - p.  Philox4x32-10 keyed by a 64-bit seed, counter =
// (index of the 16-element group, 64-bit stream offset): every element has 32 random bits of its own, whatever the
// launch shape.  One launch writes the mask; drawing fp32 uniforms, comparing and narrowing took three launches and
// eight times the bytes.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0], p1 = (unsigned long long)0xCD9E8D57u * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (unsigned)p1;
    c[3] = (unsigned)p0;
    c[0] = n0;
    c[2] = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(uint8_t* __restrict__ mask, long long n, unsigned threshold,
                                                           const long long* __restrict__ seed) {
  const unsigned long long key = (unsigned long long)seed[0], stream_off = (unsigned long long)seed[1];
  const long long groups = (n + 15) / 16;
  for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < groups; g += (long long)gridDim.x * 256) {
    unsigned w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned c[4] = {(unsigned)((unsigned long long)g * 4 + q), (unsigned)(((unsigned long long)g * 4 + q) >> 32),
                       (unsigned)stream_off, (unsigned)(stream_off >> 32)};
      philox4x32_10(c, (unsigned)key, (unsigned)(key >> 32));
      w[q] = (c[0] >= threshold ? 1u : 0u) | (c[1] >= threshold ? 0x100u : 0u) | (c[2] >= threshold ? 0x10000u : 0u) |
             (c[3] >= threshold ? 0x1000000u : 0u);
    }
    const long long e0 = g * 16;
    if (e0 + 16 <= n) {
      u32x4 t = {w[0], w[1], w[2], w[3]};
      *reinterpret_cast<u32x4*>(mask + e0) = t;
    } else {
      for (long long e = e0; e < n; ++e) mask[e] = (uint8_t)((w[(e - e0) >> 2] >> (8 * ((e - e0) & 3))) & 0xffu);
    }
  }
}

}  // namespace segk

using namespace segk;

extern "C" int segk_nchw_to_pm(const float* src, void* dst, int dst_dtype, int N, int C, int H, int W, int dst_ld,
                               segk_stream_t stream) {
  SEGK_REQUIRE(src && dst && valid_dtype(dst_dtype), "nchw_to_pm: null pointer or bad dtype");
  SEGK_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && dst_ld >= C, "nchw_to_pm: bad shape N=%d C=%d H=%d W=%d ld=%d", N,
               C, H, W, dst_ld);
  SEGK_REQUIRE(N <= 65535 && ceil_div(C, 32) <= 65535, "nchw_to_pm: grid too large");
  const int HW = H * W;
  dim3 grid(ceil_div(HW, 64), ceil_div(C, 32), N);
  hipStream_t st = (hipStream_t)stream;
  if (dst_dtype == SEGK_BF16)
    nchw_to_pm_kernel<bf16_t><<<grid, 256, 0, st>>>(src, (bf16_t*)dst, C, HW, dst_ld);
  else
    nchw_to_pm_kernel<float><<<grid, 256, 0, st>>>(src, (float*)dst, C, HW, dst_ld);
  SEGK_CHECK_LAUNCH("nchw_to_pm");
  return SEGK_OK;
}

extern "C" int segk_pm_to_nchw(const void* src, int src_dtype, int src_ld, float* dst, int N, int C, int H, int W,
                               segk_stream_t stream) {
  SEGK_REQUIRE(src && dst && valid_dtype(src_dtype), "pm_to_nchw: null pointer or bad dtype");
  SEGK_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && src_ld >= C, "pm_to_nchw: bad shape N=%d C=%d H=%d W=%d ld=%d", N,
               C, H, W, src_ld);
  SEGK_REQUIRE(N <= 65535 && ceil_div(C, 32) <= 65535, "pm_to_nchw: grid too large");
  const int HW = H * W;
  dim3 grid(ceil_div(HW, 64), ceil_div(C, 32), N);
  hipStream_t st = (hipStream_t)stream;
  if (src_dtype == SEGK_BF16)
    pm_to_nchw_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)src, dst, C, HW, src_ld);
  else
    pm_to_nchw_kernel<float><<<grid, 256, 0, st>>>((const float*)src, dst, C, HW, src_ld);
  SEGK_CHECK_LAUNCH("pm_to_nchw");
  return SEGK_OK;
}

template <typename TS, typename TD>
static void launch_pack(const void* w, long long s_co, long long s_ci, long long s_tap, void* dst, int Cout, int Cin,
                        int taps, int rows, int Ck, int transpose, hipStream_t st) {
  // patches cover co up to Cout (or up to Ck when co is the padded k index) and ci likewise
  int gx, gy, gz;
  bool all_taps;
  pack_grid(Cout, Cin, taps, s_tap, transpose, gx, gy, gz, all_taps);
  dim3 grid(gx, gy, gz);
  if (all_taps)   // OIHW: all nine taps of a 16 x 32 patch per block
    pack_weight_kernel<TS, TD, 9, 16><<<grid, 256, 0, st>>>((const TS*)w, s_co, s_ci, s_tap, (TD*)dst, Cout, Cin, taps, rows, Ck, transpose);
  else            // channels_last, or a 1x1 kernel: one tap of a 32 x 32 patch per block
    pack_weight_kernel<TS, TD, 1, 32><<<grid, 256, 0, st>>>((const TS*)w, s_co, s_ci, s_tap, (TD*)dst, Cout, Cin, taps, rows, Ck, transpose);
}

extern "C" int segk_pack_weight_strided(const void* w, int w_dtype, long long s_co, long long s_ci, long long s_tap,
                                        void* dst, int dst_dtype, int Cout, int Cin, int R, int transpose,
                                        segk_stream_t stream) {
  SEGK_REQUIRE(w && dst && valid_dtype(dst_dtype) && valid_dtype(w_dtype), "pack_weight: null pointer or bad dtype");
  SEGK_REQUIRE(Cout > 0 && Cin > 0 && (R == 1 || R == 3), "pack_weight: bad shape Cout=%d Cin=%d R=%d", Cout, Cin, R);
  SEGK_REQUIRE(s_co > 0 && s_ci > 0 && s_tap > 0, "pack_weight: strides must be positive");
  const int taps = R * R;
  const int rows = transpose ? Cin : Cout;
  const int kdim = transpose ? Cout : Cin;
  const int Ck = round_up(kdim, 32);
  SEGK_REQUIRE(ceil_div(transpose ? Cin : Ck, 32) <= 65535, "pack_weight: too many input channels");
  hipStream_t st = (hipStream_t)stream;
  if (w_dtype == SEGK_BF16) {
    if (dst_dtype == SEGK_BF16) launch_pack<bf16_t, bf16_t>(w, s_co, s_ci, s_tap, dst, Cout, Cin, taps, rows, Ck, transpose, st);
    else launch_pack<bf16_t, float>(w, s_co, s_ci, s_tap, dst, Cout, Cin, taps, rows, Ck, transpose, st);
  } else {
    if (dst_dtype == SEGK_BF16) launch_pack<float, bf16_t>(w, s_co, s_ci, s_tap, dst, Cout, Cin, taps, rows, Ck, transpose, st);
    else launch_pack<float, float>(w, s_co, s_ci, s_tap, dst, Cout, Cin, taps, rows, Ck, transpose, st);
  }
  SEGK_CHECK_LAUNCH("pack_weight");
  return SEGK_OK;
}

extern "C" int segk_pack_weight_blocks(int Cout, int Cin, int R, long long s_tap, int transpose) {
  SEGK_REQUIRE(Cout > 0 && Cin > 0 && (R == 1 || R == 3) && s_tap > 0, "pack_weight_blocks: bad shape Cout=%d Cin=%d R=%d",
               Cout, Cin, R);
  int gx, gy, gz;
  bool all_taps;
  pack_grid(Cout, Cin, R * R, s_tap, transpose, gx, gy, gz, all_taps);
  const long long blocks = (long long)gx * gy * gz;
  SEGK_REQUIRE(blocks < 2147483647LL, "pack_weight_blocks: too many workgroups");
  return (int)blocks;
}

extern "C" int segk_pack_weights(const segk_pack_item* items, int n, int total_blocks, segk_stream_t stream) {
  SEGK_REQUIRE(items && n > 0 && total_blocks > 0, "pack_weights: empty table");
  pack_weights_kernel<<<dim3((unsigned)total_blocks), 256, 0, (hipStream_t)stream>>>(items, n);
  SEGK_CHECK_LAUNCH("pack_weights");
  return SEGK_OK;
}

extern "C" int segk_pack_weight(const float* w, void* dst, int dst_dtype, int Cout, int Cin, int R, int transpose,
                                segk_stream_t stream) {
  const long long taps = (long long)R * R;
  return segk_pack_weight_strided(w, SEGK_F32, (long long)Cin * taps, taps, 1, dst, dst_dtype, Cout, Cin, R, transpose,
                                  stream);
}

extern "C" int segk_wgrad_finish_group(const float* slabs, int groups, int splits, int R, int Cin, int Cout,
                                       void* const* dw, int dw_dtype, long long s_co, long long s_ci, long long s_tap,
                                       int Cout_true, int Cin_true, segk_stream_t stream) {
  SEGK_REQUIRE(slabs && dw && valid_dtype(dw_dtype), "wgrad_finish: null pointer or bad dtype");
  SEGK_REQUIRE(groups >= 1 && groups <= SEGK_WGRAD_GROUP_MAX, "wgrad_finish: groups=%d (1..%d)", groups, SEGK_WGRAD_GROUP_MAX);
  SEGK_REQUIRE(splits >= 1 && (R == 1 || R == 3) && Cin > 0 && Cout > 0 && Cout_true > 0 && Cout_true <= Cout &&
                   Cin_true > 0 && Cin_true <= Cin,
               "wgrad_finish: bad shape splits=%d R=%d Cin=%d Cout=%d (%d,%d)", splits, R, Cin, Cout, Cout_true,
               Cin_true);
  SEGK_REQUIRE(s_co > 0 && s_ci > 0 && s_tap > 0, "wgrad_finish: strides must be positive");
  FinishGroup group;
  for (int g = 0; g < SEGK_WGRAD_GROUP_MAX; ++g) {
    group.dw[g] = g < groups ? dw[g] : nullptr;
    SEGK_REQUIRE(g >= groups || dw[g], "wgrad_finish: gradient %d is null", g);
  }
  hipStream_t st = (hipStream_t)stream;
  const unsigned gz = (unsigned)(R * R * groups);
  if (splits >= 32 && (long long)R * R * Cin * Cout <= (1ll << 18)) {
    dim3 tgrid(ceil_div(Cout, 32), ceil_div(Cin, 8), gz);
    if (dw_dtype == SEGK_BF16)
      wgrad_finish_tree_kernel<bf16_t><<<tgrid, 256, 0, st>>>(slabs, splits, R * R, Cin, Cout, group, s_co, s_ci, s_tap,
                                                             Cout_true, Cin_true);
    else
      wgrad_finish_tree_kernel<float><<<tgrid, 256, 0, st>>>(slabs, splits, R * R, Cin, Cout, group, s_co, s_ci, s_tap,
                                                            Cout_true, Cin_true);
    SEGK_CHECK_LAUNCH("wgrad_finish");
    return SEGK_OK;
  }
  if (R == 3 && s_tap == 1 && s_ci == 9) {
    dim3 agrid(ceil_div(Cout, 32), ceil_div(Cin, 32), (unsigned)groups);
    if (dw_dtype == SEGK_BF16)
      wgrad_finish_taps_kernel<bf16_t><<<agrid, 256, 0, st>>>(slabs, splits, Cin, Cout, group, s_co, Cout_true, Cin_true);
    else
      wgrad_finish_taps_kernel<float><<<agrid, 256, 0, st>>>(slabs, splits, Cin, Cout, group, s_co, Cout_true, Cin_true);
    SEGK_CHECK_LAUNCH("wgrad_finish");
    return SEGK_OK;
  }
  dim3 grid(ceil_div(Cout, 32), ceil_div(Cin, 32), gz);
  if (dw_dtype == SEGK_BF16)
    wgrad_finish_kernel<bf16_t><<<grid, 256, 0, st>>>(slabs, splits, R * R, Cin, Cout, group, s_co, s_ci, s_tap, Cout_true,
                                                      Cin_true);
  else
    wgrad_finish_kernel<float><<<grid, 256, 0, st>>>(slabs, splits, R * R, Cin, Cout, group, s_co, s_ci, s_tap, Cout_true,
                                                     Cin_true);
  SEGK_CHECK_LAUNCH("wgrad_finish");
  return SEGK_OK;
}

extern "C" int segk_wgrad_finish_strided(const float* slabs, int splits, int R, int Cin, int Cout, void* dw,
                                         int dw_dtype, long long s_co, long long s_ci, long long s_tap, int Cout_true,
                                         int Cin_true, segk_stream_t stream) {
  SEGK_REQUIRE(dw, "wgrad_finish: null pointer");
  void* one[1] = {dw};
  return segk_wgrad_finish_group(slabs, 1, splits, R, Cin, Cout, one, dw_dtype, s_co, s_ci, s_tap, Cout_true, Cin_true,
                                 stream);
}

extern "C" int segk_wgrad_finish(const float* slabs, int splits, int R, int Cin, int Cout, float* dw, int Cout_true,
                                 int Cin_true, segk_stream_t stream) {
  const long long taps = (long long)R * R;
  return segk_wgrad_finish_strided(slabs, splits, R, Cin, Cout, dw, SEGK_F32, (long long)Cin_true * taps, taps, 1,
                                   Cout_true, Cin_true, stream);
}

extern "C" int segk_dropout_mask(uint8_t* mask, long long n, float p, const long long* seed, segk_stream_t stream) {
  SEGK_REQUIRE(mask && seed && n > 0, "dropout_mask: null pointer or n=%lld", n);
  SEGK_REQUIRE(p >= 0.f && p < 1.f, "dropout_mask: p=%g outside [0, 1)", (double)p);
  SEGK_REQUIRE(((uintptr_t)mask % 16) == 0 && ((uintptr_t)seed % 8) == 0, "dropout_mask: unaligned buffers");
  // an element is dropped when its 32 random bits fall below p * 2^32
  const double scaled = (double)p * 4294967296.0;
  const unsigned threshold = scaled >= 4294967295.0 ? 4294967295u : (unsigned)(scaled + 0.5);
  const long long groups = (n + 15) / 16;
  long long blocks = (groups + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  dropout_mask_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(mask, n, threshold, seed);
  SEGK_CHECK_LAUNCH("dropout_mask");
  return SEGK_OK;
}

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
template <typename TD>
__global__ void pack_weight_kernel(const float* __restrict__ w, TD* __restrict__ dst, int Cout, int Cin, int taps,
                                   int rows, int kdim, int Ck, int transpose, long long total) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int k = (int)(i % Ck);
  long long t = i / Ck;
  int row = (int)(t % rows);
  int tapd = (int)(t / rows);
  float v = 0.f;
  if (k < kdim) {
    int co = transpose ? k : row;
    int ci = transpose ? row : k;
    int tap = transpose ? (taps - 1 - tapd) : tapd;
    v = w[((size_t)co * Cin + ci) * taps + tap];
  }
  Elem<TD>::store(dst + i, v);
}

// dw[co][ci][tap] = sum_s slabs[s][tap][ci][co]
__global__ void wgrad_finish_kernel(const float* __restrict__ slabs, int splits, int taps, int Cin, int Cout,
                                    float* __restrict__ dw, int Cout_true, int Cin_true) {
  // tile transpose over (ci, co) for one tap: reads run along co, writes along (ci, tap)
  __shared__ float tile[32][33];
  const int tap = blockIdx.z;
  const int co0 = blockIdx.x * 32, ci0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const size_t slab = (size_t)taps * Cin * Cout;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int ci = ci0 + ty + i * 8, co = co0 + tx;
    float acc = 0.f;
    if (ci < Cin && co < Cout) {
      const float* p = slabs + ((size_t)tap * Cin + ci) * Cout + co;
      for (int s = 0; s < splits; ++s) acc += p[s * slab];
    }
    tile[ty + i * 8][tx] = acc;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int co = co0 + ty + i * 8, ci = ci0 + tx;
    if (co < Cout_true && ci < Cin_true) dw[((size_t)co * Cin_true + ci) * taps + tap] = tile[tx][ty + i * 8];
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

extern "C" int segk_pack_weight(const float* w, void* dst, int dst_dtype, int Cout, int Cin, int R, int transpose,
                                segk_stream_t stream) {
  SEGK_REQUIRE(w && dst && valid_dtype(dst_dtype), "pack_weight: null pointer or bad dtype");
  SEGK_REQUIRE(Cout > 0 && Cin > 0 && (R == 1 || R == 3), "pack_weight: bad shape Cout=%d Cin=%d R=%d", Cout, Cin, R);
  const int taps = R * R;
  const int rows = transpose ? Cin : Cout;
  const int kdim = transpose ? Cout : Cin;
  const int Ck = round_up(kdim, 32);
  const long long total = (long long)taps * rows * Ck;
  const int grid = ceil_div(total, 256);
  hipStream_t st = (hipStream_t)stream;
  if (dst_dtype == SEGK_BF16)
    pack_weight_kernel<bf16_t><<<grid, 256, 0, st>>>(w, (bf16_t*)dst, Cout, Cin, taps, rows, kdim, Ck, transpose, total);
  else
    pack_weight_kernel<float><<<grid, 256, 0, st>>>(w, (float*)dst, Cout, Cin, taps, rows, kdim, Ck, transpose, total);
  SEGK_CHECK_LAUNCH("pack_weight");
  return SEGK_OK;
}

extern "C" int segk_wgrad_finish(const float* slabs, int splits, int R, int Cin, int Cout, float* dw, int Cout_true,
                                 int Cin_true, segk_stream_t stream) {
  SEGK_REQUIRE(slabs && dw, "wgrad_finish: null pointer");
  SEGK_REQUIRE(splits >= 1 && (R == 1 || R == 3) && Cin > 0 && Cout > 0 && Cout_true > 0 && Cout_true <= Cout &&
                   Cin_true > 0 && Cin_true <= Cin,
               "wgrad_finish: bad shape splits=%d R=%d Cin=%d Cout=%d (%d,%d)", splits, R, Cin, Cout, Cout_true,
               Cin_true);
  dim3 grid(ceil_div(Cout, 32), ceil_div(Cin, 32), R * R);
  wgrad_finish_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(slabs, splits, R * R, Cin, Cout, dw, Cout_true, Cin_true);
  SEGK_CHECK_LAUNCH("wgrad_finish");
  return SEGK_OK;
}

// Memory-bound pieces of the position / channel attention modules: row softmax and its gradient, batched
// transposes, the final summation of split matrix products, linear combinations and a dot product.
// The matrix products themselves run on the MFMA kernels of conv.hip (segk_gemm_nt / segk_gemm_tn).
#include "common.h"

namespace segk {

constexpr int AT_THREADS = 256;

__device__ __forceinline__ float block_max(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  __syncthreads();  // red may still be read by the previous reduction
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = red[0];
  for (int w = 1; w < AT_THREADS / 64; ++w) r = fmaxf(r, red[w]);
  return r;
}

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = red[0];
  for (int w = 1; w < AT_THREADS / 64; ++w) r += red[w];
  return r;
}

// One workgroup per row.  A row (at most a few thousand fp32) stays in L1/L2 across the three passes.
// NaN rows (0/0 upstream) propagate as NaN, like torch.
template <typename TA>
__global__ __launch_bounds__(AT_THREADS) void softmax_rows_kernel(const float* __restrict__ e, int cols, int ld_e,
                                                                  int reversed, TA* __restrict__ att, int ld_att,
                                                                  int out_cols) {
  __shared__ float red[AT_THREADS / 64];
  const long long r = blockIdx.x;
  const float* er = e + r * ld_e;
  TA* ar = att + r * ld_att;
  float mx = -INFINITY, mn = INFINITY;
  for (int j = threadIdx.x; j < cols; j += AT_THREADS) {
    const float v = er[j];
    mx = fmaxf(mx, v);
    mn = fminf(mn, v);
  }
  mx = block_max(mx, red);
  // reversed: z = mx - e, whose own maximum is mx - min(e)
  float zmax = mx;
  if (reversed) zmax = mx - (-block_max(-mn, red));
  float s = 0.f;
  for (int j = threadIdx.x; j < cols; j += AT_THREADS) {
    const float z = reversed ? mx - er[j] : er[j];
    s += expf(z - zmax);
  }
  s = block_sum(s, red);
  for (int j = threadIdx.x; j < out_cols; j += AT_THREADS) {
    float o = 0.f;
    if (j < cols) {
      const float z = reversed ? mx - er[j] : er[j];
      o = expf(z - zmax) / s;
    }
    Elem<TA>::store(ar + j, o);
  }
}

template <typename TA, typename TD>
__global__ __launch_bounds__(AT_THREADS) void softmax_rows_bwd_kernel(const TA* __restrict__ att, int ld_att,
                                                                      const float* __restrict__ datt, int ld_datt,
                                                                      int cols, float sign, TD* __restrict__ de,
                                                                      int ld_de, int out_cols) {
  __shared__ float red[AT_THREADS / 64];
  const long long r = blockIdx.x;
  const TA* ar = att + r * ld_att;
  const float* gr = datt + r * ld_datt;
  TD* dr = de + r * ld_de;
  float s = 0.f;
  for (int j = threadIdx.x; j < cols; j += AT_THREADS) s = fmaf(Elem<TA>::load(ar + j), gr[j], s);
  s = block_sum(s, red);
  for (int j = threadIdx.x; j < out_cols; j += AT_THREADS) {
    float o = 0.f;
    if (j < cols) o = sign * Elem<TA>::load(ar + j) * (gr[j] - s);
    Elem<TD>::store(dr + j, o);
  }
}

// 64 x 64 tiles through LDS (+1 padding): coalesced on both sides.
template <typename T>
__global__ __launch_bounds__(AT_THREADS) void transpose_kernel(const T* __restrict__ src, int R, int Cc, long long src_batch,
                                                               int src_ld, T* __restrict__ dst, long long dst_batch,
                                                               int dst_ld, int dst_cols) {
  __shared__ T tile[64][65];
  const T* s = src + (long long)blockIdx.z * src_batch;
  T* d = dst + (long long)blockIdx.z * dst_batch;
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += AT_THREADS / 64) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < Cc) ? s[(long long)r * src_ld + c] : (T)0;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += AT_THREADS / 64) {
    const int c = c0 + i, r = r0 + tx;  // dst row = source column
    if (c < Cc && r < dst_cols) d[(long long)c * dst_ld + r] = tile[tx][i];
  }
}

template <typename TC>
__global__ __launch_bounds__(AT_THREADS) void gemm_tn_finish_kernel(const float* __restrict__ slabs, int splits, int M, int N,
                                                                    TC* __restrict__ C, long long c_batch, int ldc,
                                                                    int rows, int cols) {
  const int b = blockIdx.z;
  const int m = blockIdx.y;
  const float* sl = slabs + ((size_t)b * splits * M + m) * N;
  TC* crow = C + (long long)b * c_batch + (long long)m * ldc;
  for (int n = blockIdx.x * AT_THREADS + threadIdx.x; n < cols; n += gridDim.x * AT_THREADS) {
    float v = 0.f;
    for (int s = 0; s < splits; ++s) v += sl[(size_t)s * M * N + n];
    Elem<TC>::store(crow + n, v);
  }
  (void)rows;
}

template <typename T, typename TO>
__global__ __launch_bounds__(AT_THREADS) void lincomb_kernel(TO* __restrict__ out, int out_ld, long long M, int C,
                                                             const T* __restrict__ x0, int ld0, float s0,
                                                             const float* __restrict__ d0, const T* __restrict__ x1,
                                                             int ld1, float s1, const float* __restrict__ d1,
                                                             const T* __restrict__ x2, int ld2, float s2,
                                                             const float* __restrict__ d2) {
  const float a0 = s0 * (d0 ? *d0 : 1.f), a1 = s1 * (d1 ? *d1 : 1.f), a2 = s2 * (d2 ? *d2 : 1.f);
  const int c = blockIdx.x * AT_THREADS + threadIdx.x;
  if (c >= C) return;
  for (long long m = blockIdx.y; m < M; m += gridDim.y) {
    float v = a0 * Elem<T>::load(x0 + m * ld0 + c);
    if (x1) v = fmaf(a1, Elem<T>::load(x1 + m * ld1 + c), v);
    if (x2) v = fmaf(a2, Elem<T>::load(x2 + m * ld2 + c), v);
    Elem<TO>::store(out + m * out_ld + c, v);
  }
}

constexpr int DOT_MAX_ROWS = 1024;

template <typename T>
__global__ __launch_bounds__(AT_THREADS) void dot_partial_kernel(const T* __restrict__ a, int lda, const T* __restrict__ b,
                                                                 int ldb, long long M, int C, double* __restrict__ partials) {
  __shared__ double red[AT_THREADS / 64];
  double s = 0.0;
  for (long long m = blockIdx.x; m < M; m += gridDim.x) {
    float rs = 0.f;
    for (int c = threadIdx.x; c < C; c += AT_THREADS) rs = fmaf(Elem<T>::load(a + m * lda + c), Elem<T>::load(b + m * ldb + c), rs);
    s += (double)rs;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < AT_THREADS / 64; ++w) t += red[w];
    partials[blockIdx.x] = t;
  }
}

__global__ void dot_finish_kernel(const double* __restrict__ partials, int rows, float* __restrict__ result) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < rows; ++i) t += partials[i];
    result[0] = (float)t;
  }
}

static int dot_rows(long long M) {
  long long r = M;
  if (r < 1) r = 1;
  if (r > DOT_MAX_ROWS) r = DOT_MAX_ROWS;
  return (int)r;
}

}  // namespace segk

using namespace segk;

extern "C" int segk_softmax_rows(const float* e, long long rows, int cols, int ld_e, int reversed, void* att,
                                 int att_dtype, int ld_att, int out_cols, segk_stream_t stream) {
  SEGK_REQUIRE(e && att && valid_dtype(att_dtype), "softmax_rows: null pointer or bad dtype");
  SEGK_REQUIRE(rows > 0 && rows < 2147483647LL && cols > 0 && cols <= ld_e && out_cols >= cols && out_cols <= ld_att,
               "softmax_rows: bad shape rows=%lld cols=%d ld_e=%d out_cols=%d ld_att=%d", rows, cols, ld_e, out_cols, ld_att);
  hipStream_t st = (hipStream_t)stream;
  SEGK_DISPATCH_DTYPE(att_dtype, TA, (softmax_rows_kernel<TA><<<(unsigned)rows, AT_THREADS, 0, st>>>(
                                         e, cols, ld_e, reversed, (TA*)att, ld_att, out_cols)));
  SEGK_CHECK_LAUNCH("softmax_rows");
  return SEGK_OK;
}

extern "C" int segk_softmax_rows_bwd(const void* att, int att_dtype, int ld_att, const float* datt, int ld_datt,
                                     long long rows, int cols, float sign, void* de, int de_dtype, int ld_de,
                                     int out_cols, segk_stream_t stream) {
  SEGK_REQUIRE(att && datt && de && valid_dtype(att_dtype) && valid_dtype(de_dtype), "softmax_rows_bwd: null pointer or bad dtype");
  SEGK_REQUIRE(rows > 0 && rows < 2147483647LL && cols > 0 && cols <= ld_att && cols <= ld_datt && out_cols >= cols &&
                   out_cols <= ld_de,
               "softmax_rows_bwd: bad shape rows=%lld cols=%d", rows, cols);
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH_SB(TA, TD)                                                                                         \
  softmax_rows_bwd_kernel<TA, TD><<<(unsigned)rows, AT_THREADS, 0, st>>>((const TA*)att, ld_att, datt, ld_datt, cols, \
                                                                         sign, (TD*)de, ld_de, out_cols)
  if (att_dtype == SEGK_BF16 && de_dtype == SEGK_BF16) { LAUNCH_SB(bf16_t, bf16_t); }
  else if (att_dtype == SEGK_BF16) { LAUNCH_SB(bf16_t, float); }
  else if (de_dtype == SEGK_BF16) { LAUNCH_SB(float, bf16_t); }
  else { LAUNCH_SB(float, float); }
#undef LAUNCH_SB
  SEGK_CHECK_LAUNCH("softmax_rows_bwd");
  return SEGK_OK;
}

extern "C" int segk_transpose(const void* src, int dtype, int batch, int R, int Cc, long long src_batch, int src_ld,
                              void* dst, long long dst_batch, int dst_ld, int dst_cols, segk_stream_t stream) {
  SEGK_REQUIRE(src && dst && valid_dtype(dtype), "transpose: null pointer or bad dtype");
  SEGK_REQUIRE(batch > 0 && batch <= 65535 && R > 0 && Cc > 0 && src_ld >= Cc && dst_cols >= R && dst_ld >= dst_cols,
               "transpose: bad shape batch=%d R=%d C=%d src_ld=%d dst_ld=%d dst_cols=%d", batch, R, Cc, src_ld, dst_ld, dst_cols);
  dim3 grid(ceil_div(dst_cols, 64), ceil_div(Cc, 64), batch);
  SEGK_REQUIRE(grid.y <= 65535, "transpose: too many columns");
  hipStream_t st = (hipStream_t)stream;
  SEGK_DISPATCH_DTYPE(dtype, T, (transpose_kernel<T><<<grid, AT_THREADS, 0, st>>>((const T*)src, R, Cc, src_batch, src_ld, (T*)dst,
                                                                                 dst_batch, dst_ld, dst_cols)));
  SEGK_CHECK_LAUNCH("transpose");
  return SEGK_OK;
}

extern "C" int segk_gemm_tn_finish(const float* slabs, int batch, int splits, int M, int N, void* C, int c_dtype,
                                   long long c_batch, int ldc, int rows, int cols, segk_stream_t stream) {
  SEGK_REQUIRE(slabs && C && valid_dtype(c_dtype), "gemm_tn_finish: null pointer or bad dtype");
  SEGK_REQUIRE(batch > 0 && batch <= 65535 && splits >= 1 && rows > 0 && rows <= M && rows <= 65535 && cols > 0 && cols <= N &&
                   cols <= ldc,
               "gemm_tn_finish: bad shape batch=%d splits=%d M=%d N=%d rows=%d cols=%d ldc=%d", batch, splits, M, N, rows,
               cols, ldc);
  dim3 grid(ceil_div(cols, AT_THREADS) > 8 ? 8 : ceil_div(cols, AT_THREADS), rows, batch);
  hipStream_t st = (hipStream_t)stream;
  SEGK_DISPATCH_DTYPE(c_dtype, TC, (gemm_tn_finish_kernel<TC><<<grid, AT_THREADS, 0, st>>>(slabs, splits, M, N, (TC*)C, c_batch,
                                                                                          ldc, rows, cols)));
  SEGK_CHECK_LAUNCH("gemm_tn_finish");
  return SEGK_OK;
}

extern "C" int segk_lincomb(void* out, int out_dtype, int out_ld, long long M, int C, const void* x0, int ld0, float s0,
                            const float* dev0, const void* x1, int ld1, float s1, const float* dev1, const void* x2,
                            int ld2, float s2, const float* dev2, int dtype, segk_stream_t stream) {
  SEGK_REQUIRE(out && x0 && valid_dtype(dtype) && valid_dtype(out_dtype), "lincomb: null pointer or bad dtype");
  SEGK_REQUIRE(M > 0 && C > 0 && out_ld >= C && ld0 >= C && (!x1 || ld1 >= C) && (!x2 || ld2 >= C),
               "lincomb: bad shape M=%lld C=%d", M, C);
  dim3 grid(ceil_div(C, AT_THREADS), (unsigned)(M < 4096 ? M : 4096));
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH_LC(T, TO)                                                                                              \
  lincomb_kernel<T, TO><<<grid, AT_THREADS, 0, st>>>((TO*)out, out_ld, M, C, (const T*)x0, ld0, s0, dev0, (const T*)x1, ld1, \
                                                     s1, dev1, (const T*)x2, ld2, s2, dev2)
  if (dtype == SEGK_BF16 && out_dtype == SEGK_BF16) { LAUNCH_LC(bf16_t, bf16_t); }
  else if (dtype == SEGK_BF16) { LAUNCH_LC(bf16_t, float); }
  else if (out_dtype == SEGK_BF16) { LAUNCH_LC(float, bf16_t); }
  else { LAUNCH_LC(float, float); }
#undef LAUNCH_LC
  SEGK_CHECK_LAUNCH("lincomb");
  return SEGK_OK;
}

extern "C" int segk_dot_partial_rows(long long M) { return M > 0 ? dot_rows(M) : SEGK_INVALID_ARG; }

extern "C" int segk_dot(const void* a, int lda, const void* b, int ldb, int dtype, long long M, int C, double* partials,
                        float* result, segk_stream_t stream) {
  SEGK_REQUIRE(a && b && partials && result && valid_dtype(dtype), "dot: null pointer or bad dtype");
  SEGK_REQUIRE(M > 0 && C > 0 && lda >= C && ldb >= C, "dot: bad shape M=%lld C=%d", M, C);
  const int rows = dot_rows(M);
  hipStream_t st = (hipStream_t)stream;
  SEGK_DISPATCH_DTYPE(dtype, T, (dot_partial_kernel<T><<<rows, AT_THREADS, 0, st>>>((const T*)a, lda, (const T*)b, ldb, M, C, partials)));
  SEGK_CHECK_LAUNCH("dot");
  dot_finish_kernel<<<1, 64, 0, st>>>(partials, rows, result);
  SEGK_CHECK_LAUNCH("dot_finish");
  return SEGK_OK;
}

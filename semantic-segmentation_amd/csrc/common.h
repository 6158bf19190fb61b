// Shared helpers for the libsegk.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "segk.h"

namespace segk {

void set_error(const char* fmt, ...);

#define SEGK_REQUIRE(cond, ...)                    \
  do {                                             \
    if (!(cond)) {                                 \
      ::segk::set_error(__VA_ARGS__);              \
      return SEGK_INVALID_ARG;                     \
    }                                              \
  } while (0)

#define SEGK_UNSUPPORTED(cond, ...)                \
  do {                                             \
    if (!(cond)) {                                 \
      ::segk::set_error(__VA_ARGS__);              \
      return SEGK_UNSUPPORTED_SHAPE;               \
    }                                              \
  } while (0)

// Call after every launch: launch-configuration errors surface here without a sync.
#define SEGK_CHECK_LAUNCH(name)                                                      \
  do {                                                                               \
    hipError_t e_ = hipGetLastError();                                               \
    if (e_ != hipSuccess) {                                                          \
      ::segk::set_error("%s: launch failed: %s", name, hipGetErrorString(e_));       \
      return SEGK_HIP_ERROR;                                                         \
    }                                                                                \
  } while (0)

// loss.hip: rows of {sum w*nll, sum w} partials for `pixels` pixels, and their fixed-order fp64 sum -> sums[2]
int ce_partial_rows(long long pixels);
int launch_ce_finish(const float* partials, int rows, float* sums, hipStream_t st);

static inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }
static inline bool valid_dtype(int d) { return d == SEGK_F32 || d == SEGK_BF16; }
static inline int dtype_size(int d) { return d == SEGK_BF16 ? 2 : 4; }

typedef unsigned short bf16_t;  // raw bits
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned int)v) << 16); }
// Round to nearest even; a plain cast keeps NaN a NaN (lowers to v_cvt_pk_bf16_f32).
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
  return (unsigned int)f32_to_bf16(lo) | ((unsigned int)f32_to_bf16(hi) << 16);
}

template <typename T>
struct Elem;
template <>
struct Elem<float> {
  static __device__ __forceinline__ float load(const float* p) { return *p; }
  static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
};
template <>
struct Elem<bf16_t> {
  static __device__ __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// Sum over the 64 lanes of a wavefront; every lane receives the total.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Dispatch a (possibly two-) dtype template on runtime tags.
#define SEGK_DISPATCH_DTYPE(dt, T, ...)            \
  if ((dt) == SEGK_BF16) {                         \
    typedef ::segk::bf16_t T;                      \
    __VA_ARGS__;                                   \
  } else {                                         \
    typedef float T;                               \
    __VA_ARGS__;                                   \
  }

}  // namespace segk

// Convolution as implicit GEMM on the CDNA4 matrix cores.
//
// Forward / data gradient (conv_fwd_kernel):  out[m][co] = sum_seg sum_tap sum_c in[m + tap][c] * w[tap][co][c]
//   GEMM view: rows = output channels (MFMA A operand = packed weights, K-contiguous),
//              cols = pixels         (MFMA B operand = pixel-major activations, K-contiguous),
//              K    = (segment, tap, channel).  A 128(co) x 128(pixel) tile per workgroup, 4 wavefronts in a
//   2x2 grid, each owning 64x64 = 4x4 MFMA 16x16 tiles.  Both operand tiles are staged through LDS as
//   [128 rows][BKB bytes] with a 16-byte-chunk XOR swizzle so the ds_read_b128 fragment reads are
//   conflict-free.  Taps whose shifted window lies outside the image for every pixel of the tile are skipped
//   (atrous rates comparable to the map size make most off-centre taps empty).
//   bf16: v_mfma_f32_16x16x32_bf16; fp32: v_mfma_f32_16x16x4_f32 (exact fp32 fma chain).
//
// Weight gradient (conv_wgrad_kernel): dw[tap][ci][co] = sum_m dout[m][co] * in[m + tap][ci]
//   GEMM view: rows = co, cols = ci, K = pixels.  Operands are pixel-major, i.e. K is the slow index, so
//   bf16 fragments are read with the transposing ds_read_b64_tr_b16 and fp32 fragments with ds_read_b32.
//   The pixel range is split over workgroups; each split writes its own fp32 slab (summed later in fixed
//   order, so results are reproducible: no float atomics).
#include <math.h>
#include <stdlib.h>

#include <atomic>
#include <type_traits>
#include <cstdio>
#include "common.h"

namespace segk {

// Kernel-selection switches for experiments (tools/conv_bench.py, tools/conv_stamps.py) exist only in builds made
// with -DSEGK_EXPERIMENTS (make EXPERIMENTS=1); the product library reads no environment variable and contains no
// ablation kernel.
#ifdef SEGK_EXPERIMENTS
static const char* experiment(const char* name) { return getenv(name); }
#else
static constexpr const char* experiment(const char*) { return nullptr; }
#endif

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;

constexpr int CONV_THREADS = 256;
constexpr int TILE = 128;  // both GEMM tile edges

struct SegDev {
  const char* in;
  const char* w;
  long long in_ld_b;  // bytes between pixels
  long long w_tap_b;  // bytes between taps of the packed weight
  long long in_batch_b, w_batch_b;  // bytes between the operands of consecutive batch entries (blockIdx.y)
  int w_ld_b;         // bytes between weight rows (= ck_b for packed weights)
  int ck_b;           // contracted bytes per (pixel, tap)
  int R, dil;
};

struct ConvDev {
  int H, W, HW;
  long long M;
  int Cout, nseg, n_tiles;
  SegDev seg[SEGK_MAX_SEG];
  char* out;
  int out_ld;
  long long out_batch_b;
  int dil_step;          // batch entry b convolves with dilation seg.dil + b * dil_step (the ASPP rate ladder)
  long long stat_batch;  // doubles between the statistic partials of consecutive batch entries
  const float* bias;
  const float* img_bias;
  const char* addend;    // nullptr, or [M][addend_ld] of the output dtype, added to the rounded result (ring tile only)
  long long addend_ld_b;
  const uint8_t* addend_bits;   // nullptr, or [M][addend_bits_ld] bytes: bit c % 8 of byte c / 8 keeps addend[m][c], a clear bit adds zero
  int addend_bits_ld;
  double* stat;
  // Backward statistics of the batch norm whose output this launch differentiates (ring tile, bf16 output): with
  // g = out through that layer's ReLU, partial sums of g and g * xhat per channel, one row per 80-pixel half of a tile
  // as for `stat`.  bst_x = the layer's input [M][bst_x_ld_b bytes] (bf16); the ReLU pattern comes from bst_bits
  // ([M][bst_bits_ld] bytes, one bit per channel) or, without them, from x * scale + shift > 0.
  double* bst_rows;
  const char* bst_x;
  long long bst_x_ld_b;
  const float* bst_mean;
  const float* bst_inv_std;
  const float* bst_scale;
  const float* bst_shift;
  const uint8_t* bst_bits;
  int bst_bits_ld;
  int bst_relu;
  long long m_begin;     // first pixel of this launch (a problem may be cut into a main and a tail launch)
  long long stat_row0;   // statistic partial row of this launch's first pixel tile
  int ring_tiles;        // ring tile: pixel tiles x output-channel tiles of the launch (workgroups may walk over several)
  int abl;               // timing experiments of the LDS-DMA tile (experiment builds only; results are garbage), a
                         // bit mask: 1 = every load reads the zero page (ring tile: out of range, no memory access), 2 = no fragment reads / MFMAs, 4 = no
                         // per-step address bookkeeping, 8 = no output stores, 16 = no K loop (fixed cost only),
                         // 32 = raised issue priority for the multiplying wavefronts as well, 64 = none for the loading ones
                         // (both with valid results)
};

template <typename T>
struct Mma;
template <>
struct Mma<bf16_t> {
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <>
struct Mma<float> {
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& c) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[e]), __uint_as_float(b[e]), c, 0, 0, 0);
  }
};

// pixel index within its image; 32-bit arithmetic whenever the global pixel index fits (a 64-bit modulo is ~50
// instructions, and the weight-gradient loader needs one per row per chunk)
__device__ __forceinline__ int pixel_in_image(long long m, int HW) {
  return (m >> 31) == 0 ? (int)((unsigned)m % (unsigned)HW) : (int)(m % HW);
}

// byte offset of 16-byte chunk c of row r in a [128][BKB] tile
template <int BKB>
__device__ __forceinline__ int tile_off(int r, int c) {
  if (BKB == 128) return r * 128 + ((c ^ (r & 7)) << 4);
  return r * 64 + ((c ^ ((-(r >> 2)) & 3)) << 4);
}

template <typename TO>
__device__ __forceinline__ void store_out4(char* p, const float (&v)[4]);
template <>
__device__ __forceinline__ void store_out4<float>(char* p, const float (&v)[4]) {
  f32x4 t = {v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4*>(p) = t;
}
template <>
__device__ __forceinline__ void store_out4<bf16_t>(char* p, const float (&v)[4]) {
  u32x2 t = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
  *reinterpret_cast<u32x2*>(p) = t;
}

// a + b of two packed bf16 pairs, each lane rounded once more to bf16: what adding two bf16 tensors gives
__device__ __forceinline__ unsigned int add_bf16x2(unsigned int a, unsigned int b) {
  const float lo = __uint_as_float(a << 16) + __uint_as_float(b << 16);
  const float hi = __uint_as_float(a & 0xffff0000u) + __uint_as_float(b & 0xffff0000u);
  return pack_bf16x2(lo, hi);
}

// the bf16 pair `v` with the element of a clear bit replaced by +0.0: bits b0 (low half) and b1 (high half)
__device__ __forceinline__ unsigned int keep_bf16x2(unsigned int v, unsigned int b0, unsigned int b1) {
  return v & ((b0 ? 0x0000ffffu : 0u) | (b1 ? 0xffff0000u : 0u));
}

// Tile geometry: a workgroup owns BCO output channels x BPX pixels with WCO x WPX wavefronts; each wavefront
// owns (BCO/WCO) x (BPX/WPX) = MI x MJ MFMA 16x16 tiles.
//   Small: 128 x 128, 2 x 2 waves (64 x 64 per wave), 2 workgroups per CU - problems with few tiles.
//   Big:   256 x 256, 4 x 2 waves (64 co x 128 px per wave), 1 workgroup per CU - twice the MFMA work per byte
//          staged through LDS and four times the work per barrier (hides the global-load latency of the
//          one-step-ahead prefetch).
template <int BCO_, int BPX_, int WCO_, int WPX_>
struct TileCfg {
  static constexpr int BCO = BCO_, BPX = BPX_, WCO = WCO_, WPX = WPX_;
  static constexpr int THREADS = 64 * WCO_ * WPX_;
  static constexpr int MI = BCO_ / WCO_ / 16, MJ = BPX_ / WPX_ / 16;
};
typedef TileCfg<128, 128, 2, 2> TileSmall;
typedef TileCfg<256, 256, 4, 2> TileBig;

// Sum of a double over the 16 lanes of a DPP row, result in every lane: two quad permutes, then the half-row and the
// row mirror (VALU only; a 64-bit __shfl_xor goes through the LDS crossbar twice per step).
template <int CTRL>
__device__ __forceinline__ double dpp_add_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum(double v) {
  v = dpp_add_f64<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_add_f64<0x4E>(v);   // quad_perm [2,3,0,1]
  v = dpp_add_f64<0x141>(v);  // row_half_mirror
  v = dpp_add_f64<0x140>(v);  // row_mirror
  return v;
}

// One DPP move of a double: lane i of the result holds v of the lane that CTRL names for it; lanes outside the row /
// bank masks keep `old`.
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ double dpp_mov_f64(double old, double v) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROW_MASK, BANK_MASK, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROW_MASK, BANK_MASK, false);
  return __hiloint2double(hi, lo);
}

// Sums NV doubles per lane over the 16 lanes of a DPP row and leaves them spread over those lanes: afterwards
// v[0 .. NV/16) of lane l hold the row totals of the original indices row16_scatter_base<NV>(l) + k.  Four halving
// exchanges (partner lane l^1, l^2, l^8, l^4): a lane keeps the half of its values its bit selects and adds the
// partner's copy of that half, so the exchange of step s moves NV / 2^s values instead of NV - 15 additions per 16
// values in all, where summing every value with row16_sum takes 64.
template <int NV>
__device__ __forceinline__ void row16_reduce_scatter(double (&v)[NV], int lane) {
  static_assert(NV % 16 == 0, "one or more values per lane must remain");
  {
    const bool up = lane & 1;
    constexpr int H = NV / 2;
#pragma unroll
    for (int k = 0; k < H; ++k) {
      const double keep = up ? v[k + H] : v[k], send = up ? v[k] : v[k + H];
      v[k] = keep + dpp_mov_f64<0xB1>(0.0, send);          // quad_perm [1,0,3,2]
    }
  }
  {
    const bool up = lane & 2;
    constexpr int H = NV / 4;
#pragma unroll
    for (int k = 0; k < H; ++k) {
      const double keep = up ? v[k + H] : v[k], send = up ? v[k] : v[k + H];
      v[k] = keep + dpp_mov_f64<0x4E>(0.0, send);          // quad_perm [2,3,0,1]
    }
  }
  {
    const bool up = lane & 8;
    constexpr int H = NV / 8;
#pragma unroll
    for (int k = 0; k < H; ++k) {
      const double keep = up ? v[k + H] : v[k], send = up ? v[k] : v[k + H];
      v[k] = keep + dpp_mov_f64<0x128>(0.0, send);         // row_ror:8
    }
  }
  {
    const bool up = lane & 4;
    constexpr int H = NV / 16;
#pragma unroll
    for (int k = 0; k < H; ++k) {
      const double keep = up ? v[k + H] : v[k], send = up ? v[k] : v[k + H];
      // partner l^4: lanes 0-3 and 8-11 of a row (banks 0, 2) read four lanes up, the others four lanes down
      double t = dpp_mov_f64<0x104, 0xf, 0x5>(0.0, send);  // row_shl:4
      t = dpp_mov_f64<0x114, 0xf, 0xA>(t, send);           // row_shr:4
      v[k] = keep + t;
    }
  }
}
template <int NV>
__device__ __forceinline__ int row16_scatter_base(int lane) {
  return ((lane & 1) ? NV / 2 : 0) + ((lane & 2) ? NV / 4 : 0) + ((lane & 8) ? NV / 8 : 0) + ((lane & 4) ? NV / 16 : 0);
}

// Epilogue shared by the forward kernels: optional fp64 batch-norm statistics of the unbiased accumulators, then
// bias / per-image bias and the store.  acc[i][j][r] = out[pixel = wc*MJ*16 + j*16 + (lane&15)]
//                                                         [co = wr*MI*16 + i*16 + (lane>>4)*4 + r]
template <typename TO, typename G, int ABL = 0>
__device__ __forceinline__ void conv_epilogue(const ConvDev& p, f32x4 (&acc)[G::MI][G::MJ], char* smem, int wr, int wc,
                                              int lane, int tid, int co0, long long m0, long long mt, long long bz) {
  constexpr int MI = G::MI, MJ = G::MJ, NT = G::THREADS;
  const int fq = lane >> 4, fr = lane & 15;
  // timing ablations 5 (no output stores) and 6 (no statistics either): p.M < 0 is never true, the code stays live
  if (p.stat && !(ABL == 6 && p.M >= 0)) {
    // fp64 sums: the squares of the fp32 accumulators are exact in fp64
    double* red = reinterpret_cast<double*>(smem);  // [WPX pixel parts][2][BCO]
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      double s[4], q[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[r] = 0.0;
        q[r] = 0.0;
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
          const double v = (double)acc[i][j][r];
          s[r] += v;
          q[r] = fma(v, v, q[r]);
        }
        s[r] = row16_sum(s[r]);
        q[r] = row16_sum(q[r]);
      }
      if (fr == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cl = wr * (MI * 16) + i * 16 + fq * 4 + r;
          red[(wc * 2 + 0) * G::BCO + cl] = s[r];
          red[(wc * 2 + 1) * G::BCO + cl] = q[r];
        }
      }
    }
    __syncthreads();
    for (int t = tid; t < 2 * G::BCO; t += NT) {
      const int which = t / G::BCO, cl = t % G::BCO;
      const int co = co0 + cl;
      if (co < p.Cout) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < G::WPX; ++w) v += red[(w * 2 + which) * G::BCO + cl];
        p.stat[bz * p.stat_batch + ((size_t)(p.stat_row0 + mt) * 2 + which) * p.Cout + co] = v;
      }
    }
  }

  const int cpad = (p.Cout + 3) & ~3;
  if ((ABL == 5 || ABL == 6) && p.M >= 0) return;
  if (p.abl & 8) return;
#pragma unroll
  for (int j = 0; j < MJ; ++j) {
    const long long m = m0 + wc * (MJ * 16) + j * 16 + fr;
    if (m >= p.M) continue;
    const long long n = m / p.HW;
    char* orow = p.out + bz * p.out_batch_b + (size_t)m * p.out_ld * sizeof(TO);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int c = co0 + wr * (MI * 16) + i * 16 + fq * 4;
      if (c >= cpad) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t = acc[i][j][r];
        if (c + r < p.Cout) {
          if (p.bias) t += p.bias[c + r];
          if (p.img_bias) t += p.img_bias[n * p.Cout + c + r];
        } else {
          t = 0.f;
        }
        v[r] = t;
      }
      store_out4<TO>(orow + (size_t)c * sizeof(TO), v);
    }
  }
}

template <typename T, int BKB, typename TO, typename G, int ABL = 0>
__global__ __launch_bounds__(G::THREADS, (G::THREADS == 256 && G::BCO == 256) ? 1 : 2) void conv_fwd_kernel(const ConvDev p) {
  constexpr int NT = G::THREADS;
  constexpr int CH = BKB / 16;               // 16-byte chunks per tile row
  constexpr int RPP = NT / CH;               // rows loaded per pass
  constexpr int NPA = G::BCO / RPP;          // passes for the weight tile
  constexpr int NPB = G::BPX / RPP;          // passes for the activation tile
  constexpr int TA_B = G::BCO * BKB, TB_B = G::BPX * BKB, STAGE_B = TA_B + TB_B;
  constexpr int KS = BKB / 64;               // MFMA k-steps per tile (one 16-byte chunk per lane per step)
  constexpr int MI = G::MI, MJ = G::MJ;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int taps_l[SEGK_MAX_SEG * 9];
  __shared__ int ntaps_l[SEGK_MAX_SEG];
  __shared__ unsigned int mask_l[2];
  // per listed tap: byte offsets of its weight slice and of its shifted pixel window, and the shift (filled once;
  // the K loop then reads a table instead of dividing and multiplying per step)
  __shared__ long long tap_a_l[SEGK_MAX_SEG * 9];
  __shared__ long long tap_b_l[SEGK_MAX_SEG * 9];
  __shared__ int tap_dy_l[SEGK_MAX_SEG * 9];
  __shared__ int tap_dx_l[SEGK_MAX_SEG * 9];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave / G::WPX, wc = wave % G::WPX;   // wave row (channels) / column (pixels)
  const int nt = blockIdx.x % p.n_tiles;
  const long long mt = blockIdx.x / p.n_tiles;
  const int co0 = nt * G::BCO;
  const long long m0 = p.m_begin + mt * G::BPX;
  const long long bz = blockIdx.y;  // batch entry (batched GEMM use; 0 for convolutions)

  const int lc = tid % CH, lr = tid / CH;
  int wrow[NPA], ry[NPB], rx[NPB];
  long long rm[NPB];
  bool rvalid[NPB];
#pragma unroll
  for (int i = 0; i < NPA; ++i) {
    const int co = co0 + lr + i * RPP;
    wrow[i] = co < p.Cout ? co : p.Cout - 1;
  }
#pragma unroll
  for (int i = 0; i < NPB; ++i) {
    const long long m = m0 + lr + i * RPP;
    rvalid[i] = m < p.M;
    const long long mm = rvalid[i] ? m : 0;
    rm[i] = mm;
    const int rem = pixel_in_image(mm, p.HW);
    ry[i] = rem / p.W;
    rx[i] = rem - ry[i] * p.W;
  }

  // ---- which taps touch this tile at all -------------------------------------------------
  if (tid < 2) mask_l[tid] = 0u;
  __syncthreads();
  {
    unsigned long long mine = 0ull;
    for (int sg = 0; sg < p.nseg; ++sg) {
      const int R = p.seg[sg].R, dil = p.seg[sg].dil + (int)blockIdx.y * p.dil_step;
      for (int tap = 0; tap < R * R; ++tap) {
        const int dy = (tap / R - (R >> 1)) * dil, dx = (tap % R - (R >> 1)) * dil;
        bool any = false;
#pragma unroll
        for (int i = 0; i < NPB; ++i)
          any |= rvalid[i] && (unsigned)(ry[i] + dy) < (unsigned)p.H && (unsigned)(rx[i] + dx) < (unsigned)p.W;
        if (any) mine |= 1ull << (sg * 9 + tap);
      }
    }
    const unsigned int lo = (unsigned int)mine, hi = (unsigned int)(mine >> 32);
    if (lo) atomicOr(&mask_l[0], lo);
    if (hi) atomicOr(&mask_l[1], hi);
  }
  __syncthreads();
  if (tid < SEGK_MAX_SEG * 9) {
    // thread (segment, tap): a tap that touches the tile takes the list slot given by the number of listed taps
    // before it
    const unsigned long long all = ((unsigned long long)mask_l[1] << 32) | mask_l[0];
    const int sg = tid / 9, tap = tid - sg * 9;
    if (sg < p.nseg) {
      const unsigned seg_bits = (unsigned)(all >> (sg * 9)) & 0x1ffu;
      if (tap == 0) ntaps_l[sg] = __popc(seg_bits);
      if ((seg_bits >> tap) & 1u) {
        const SegDev& sd = p.seg[sg];
        const int slot = sg * 9 + __popc(seg_bits & ((1u << tap) - 1u));
        const int R = sd.R, dil = sd.dil + (int)blockIdx.y * p.dil_step;
        const int dy = (tap / R - (R >> 1)) * dil, dx = (tap % R - (R >> 1)) * dil;
        taps_l[slot] = tap;
        tap_a_l[slot] = (long long)tap * sd.w_tap_b;
        tap_b_l[slot] = ((long long)dy * p.W + dx) * sd.in_ld_b;
        tap_dy_l[slot] = dy;
        tap_dx_l[slot] = dx;
      }
    }
  }
  __syncthreads();
  int total = 0;
  for (int sg = 0; sg < p.nseg; ++sg) total += ntaps_l[sg] * (p.seg[sg].ck_b / BKB);

  f32x4 acc[MI][MJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < MJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (total > 0) {
    // ---- K-loop state ---------------------------------------------------------------------
    // Order: segment -> 64-channel chunk -> tap (the nine shifted windows of one channel chunk follow each
    // other, so their overlapping pixel rows stay in L2).  Addresses are a wave-uniform 64-bit base (tile
    // origin + tap shift + chunk) plus a per-thread 32-bit offset, which keeps address registers few.
    int sg = 0;
    while (ntaps_l[sg] == 0) ++sg;
    int ti = 0, kc = 0, nkc = p.seg[sg].ck_b / BKB;
    unsigned aoff[NPA], boff[NPB];
    bool bok[NPB];
    const char* abase = nullptr;
    const char* bbase = nullptr;
    const char* aseg = nullptr;   // weight rows / pixel rows of this tile in the current segment
    const char* bseg = nullptr;
    auto setup_seg = [&]() {
      const SegDev& s = p.seg[sg];
#pragma unroll
      for (int i = 0; i < NPA; ++i) aoff[i] = (unsigned)(wrow[i] - co0) * (unsigned)s.w_ld_b + lc * 16;
#pragma unroll
      for (int i = 0; i < NPB; ++i) boff[i] = (unsigned)(rm[i] >= m0 ? rm[i] - m0 : 0) * (unsigned)s.in_ld_b + lc * 16;
      aseg = s.w + bz * s.w_batch_b + (long long)co0 * s.w_ld_b;
      bseg = s.in + bz * s.in_batch_b + m0 * s.in_ld_b;
    };
    auto setup_tap = [&]() {
      const int slot = sg * 9 + ti;
      const int dy = tap_dy_l[slot], dx = tap_dx_l[slot];
      abase = aseg + tap_a_l[slot] + kc * BKB;
      bbase = bseg + tap_b_l[slot] + kc * BKB;
#pragma unroll
      for (int i = 0; i < NPB; ++i)
        bok[i] = rvalid[i] && (unsigned)(ry[i] + dy) < (unsigned)p.H && (unsigned)(rx[i] + dx) < (unsigned)p.W;
    };
    auto advance = [&]() {
      if (++ti >= ntaps_l[sg]) {
        ti = 0;
        if (++kc >= nkc) {
          kc = 0;
          ++sg;
          while (ntaps_l[sg] == 0) ++sg;
          nkc = p.seg[sg].ck_b / BKB;
          setup_seg();
        }
      }
      setup_tap();
    };
    u32x4 ra[NPA], rb[NPB];
    auto load_tile = [&]() {
      if (ABL == 1 || ABL == 3) {  // ablation: no global loads (results are garbage, timing only)
#pragma unroll
        for (int i = 0; i < NPA; ++i) asm volatile("" : "+v"(ra[i]));
#pragma unroll
        for (int i = 0; i < NPB; ++i) asm volatile("" : "+v"(rb[i]));
        return;
      }
#pragma unroll
      for (int i = 0; i < NPA; ++i) ra[i] = *reinterpret_cast<const u32x4*>(abase + aoff[i]);
#pragma unroll
      for (int i = 0; i < NPB; ++i) {
        u32x4 z = {0u, 0u, 0u, 0u};
        rb[i] = bok[i] ? *reinterpret_cast<const u32x4*>(bbase + boff[i]) : z;
      }
    };
    auto store_tile = [&](int stage) {
      if (ABL == 2 || ABL == 3) {  // ablation: no LDS writes
#pragma unroll
        for (int i = 0; i < NPA; ++i) asm volatile("" ::"v"(ra[i]));
#pragma unroll
        for (int i = 0; i < NPB; ++i) asm volatile("" ::"v"(rb[i]));
        return;
      }
      char* sa = smem + stage * STAGE_B;
      char* sb = sa + TA_B;
#pragma unroll
      for (int i = 0; i < NPA; ++i) *reinterpret_cast<u32x4*>(sa + tile_off<BKB>(lr + i * RPP, lc)) = ra[i];
#pragma unroll
      for (int i = 0; i < NPB; ++i) *reinterpret_cast<u32x4*>(sb + tile_off<BKB>(lr + i * RPP, lc)) = rb[i];
    };

    setup_seg();
    setup_tap();
    load_tile();
    store_tile(0);
    __syncthreads();
    const int frow = lane & 15, fg = lane >> 4;
    for (int st = 0; st < total; ++st) {
      const int cur = st & 1;
      const bool more = st + 1 < total;
      if (more) {
        advance();
        load_tile();
      }
      const char* sa = smem + cur * STAGE_B;
      const char* sb = sa + TA_B;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        u32x4 fb[MJ];
#pragma unroll
        for (int j = 0; j < MJ; ++j)
          fb[j] = *reinterpret_cast<const u32x4*>(sb + tile_off<BKB>(wc * (MJ * 16) + j * 16 + frow, ks * 4 + fg));
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const u32x4 fa = *reinterpret_cast<const u32x4*>(sa + tile_off<BKB>(wr * (MI * 16) + i * 16 + frow, ks * 4 + fg));
#pragma unroll
          for (int j = 0; j < MJ; ++j) Mma<T>::run(fa, fb[j], acc[i][j]);
        }
      }
      if (more) store_tile(cur ^ 1);
      __syncthreads();
    }
  }

  conv_epilogue<TO, G, ABL>(p, acc, smem, wr, wc, lane, tid, co0, m0, mt, bz);
}

// ------------------------------------------------------------------------------------------------
// Forward / data gradient, deep-pipelined variant: bf16 operands, Cout a multiple of 256, 128-byte K rows.
//
// 256(co) x 256(pixel) tile, 8 wavefronts as 2 (co) x 4 (pixel), 128 x 64 per wavefront.  Operands reach LDS by
// LDS-DMA (global_load_lds_dwordx4: no staging registers) in half-tiles of 128 rows x 128 bytes = 16 KiB; LDS is a
// ring of 8 such slots.  A K-tile (64 channels of one tap) consists of four half-tiles
//      A0 / A1 = weight rows {wr*128 + h*64 ..+64}  (the h-th 64-row half of BOTH wavefront rows)
//      B0 / B1 = pixels      {wc*64  + h*32 ..+32}  (the h-th 32-pixel half of all four wavefront columns)
// and is consumed in four phases of 16 MFMAs each (one quadrant of the wavefront's 8 x 4 accumulator tiles):
//      phase 4t+0: read A0(t)           Q00 += A0 x B0(t)
//      phase 4t+1: read B1(t)           Q01 += A0 x B1
//      phase 4t+2: read A1(t)           Q11 += A1 x B1
//      phase 4t+3: read B0(t+1)         Q10 += A1 x B0(t)      (B0 alternates between two register sets)
// so the half-tile read in phase n sits in ring slot n & 7.  In phase n every wavefront also issues its share
// (two 1-KiB pieces) of the half-tile that phase n+6 will read, into slot (n-2) & 7, whose last reader was phase
// n-2; `s_waitcnt vmcnt(10)` then leaves the five youngest half-tiles in flight and retires the one phase n+1
// reads.  A phase is {LDS reads, DMA issue, counted wait | barrier | 16 MFMA | barrier}; wavefronts 4-7 run one
// barrier behind wavefronts 0-3, so on every SIMD one wavefront issues MFMAs while the other one loads.
// Ordering rules the schedule relies on: a slot is read one phase (or more) after the wait that retired its DMA,
// with a barrier in between that every wavefront reaches after its own wait; a slot is re-filled two phases after
// its last read, by which time both groups have passed the `lgkmcnt(0)` that follows that read.
// Rows outside the image (zero padding) are fetched from a zero page; phases past the last K-tile fetch it too,
// which keeps the wait count uniform.
// ------------------------------------------------------------------------------------------------
constexpr int PP_SLOT = 16384;
constexpr int PP_TABLE = 8 * PP_SLOT;             // tap tables behind the ring (one LDS array for everything)
constexpr int PP_LDS = PP_TABLE + 512;
typedef TileCfg<256, 256, 2, 4> TilePipe;

__device__ __attribute__((aligned(256))) unsigned char pp_zero_page[256];

__device__ __forceinline__ void glds16(const char* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// ABL (timing experiments only, results are garbage): 1 = every DMA reads the zero page, 2 = no DMA at all,
// 3 = no MFMA, 4 = no fragment reads, 6 = no DMA and no fragment reads (MFMA + barriers only),
// 7 = no main loop, 9 = s_memtime stamps per phase section, summed per wavefront into the statistics buffer,
// 8 = every workgroup fetches the pixels of tile 0 (real addressing, cache-resident data)
template <typename TO, int ABL = 0>
__global__ __launch_bounds__(512, 1) void conv_fwd_pipe_kernel(const ConvDev p) {
  typedef TilePipe G;
  constexpr int BKB = 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* taps_l = reinterpret_cast<int*>(smem + PP_TABLE);           // [SEGK_MAX_SEG * 9]
  int* ntaps_l = taps_l + SEGK_MAX_SEG * 9;                        // [SEGK_MAX_SEG]
  unsigned int* mask_l = reinterpret_cast<unsigned int*>(ntaps_l + SEGK_MAX_SEG);  // [2]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int nt = blockIdx.x % p.n_tiles;
  const long long mt = blockIdx.x / p.n_tiles;
  const int co0 = nt * G::BCO;
  const long long m0 = p.m_begin + mt * G::BPX;
  const long long bz = blockIdx.y;

  // ---- loader geometry: this thread's rows of every half-tile ------------------------------
  // piece q = i*8 + wave (i = 0,1) covers half-tile rows q*8 .. q*8+7; lane l -> row q*8 + (l>>3), LDS slot l&7,
  // which holds source chunk (l&7) ^ (row&7) (the XOR swizzle the fragment reads undo)
  const int lrow = lane >> 3;
  const int lchunk = ((lane & 7) ^ lrow) << 4;
  int a_row[2][2], b_pix[2][2], b_y[2][2], b_x[2][2];
  bool b_in[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rr = (i * 8 + wave) * 8 + lrow;
      a_row[h][i] = (rr >> 6) * 128 + h * 64 + (rr & 63);
      b_pix[h][i] = (rr >> 5) * 64 + h * 32 + (rr & 31);
      const long long m = m0 + b_pix[h][i];
      b_in[h][i] = m < p.M;
      const int rem = pixel_in_image(b_in[h][i] ? m : 0, p.HW);
      b_y[h][i] = rem / p.W;
      b_x[h][i] = rem - b_y[h][i] * p.W;
    }

  // ---- which taps touch this tile at all -------------------------------------------------
  if (tid < 2) mask_l[tid] = 0u;
  __syncthreads();
  {
    unsigned long long mine = 0ull;
    for (int sg = 0; sg < p.nseg; ++sg) {
      const int R = p.seg[sg].R, dil = p.seg[sg].dil + (int)blockIdx.y * p.dil_step;
      for (int tap = 0; tap < R * R; ++tap) {
        const int dy = (tap / R - (R >> 1)) * dil, dx = (tap % R - (R >> 1)) * dil;
        bool any = false;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            any |= b_in[h][i] && (unsigned)(b_y[h][i] + dy) < (unsigned)p.H && (unsigned)(b_x[h][i] + dx) < (unsigned)p.W;
        if (any) mine |= 1ull << (sg * 9 + tap);
      }
    }
    const unsigned int lo = (unsigned int)mine, hi = (unsigned int)(mine >> 32);
    if (lo) atomicOr(&mask_l[0], lo);
    if (hi) atomicOr(&mask_l[1], hi);
  }
  __syncthreads();
  if (tid == 0) {
    const unsigned long long all = ((unsigned long long)mask_l[1] << 32) | mask_l[0];
    for (int sg = 0; sg < p.nseg; ++sg) {
      int cnt = 0;
      const int R = p.seg[sg].R;
      for (int tap = 0; tap < R * R; ++tap)
        if ((all >> (sg * 9 + tap)) & 1ull) taps_l[sg * 9 + cnt++] = tap;
      ntaps_l[sg] = cnt;
    }
  }
  __syncthreads();
  int total = 0;
  for (int sg = 0; sg < p.nseg; ++sg) total += ntaps_l[sg] * (p.seg[sg].ck_b / BKB);
  if (ABL == 7) total = 0;  // prologue + epilogue only

  f32x4 acc[G::MI][G::MJ];
#pragma unroll
  for (int i = 0; i < G::MI; ++i)
#pragma unroll
    for (int j = 0; j < G::MJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (total > 0) {
    // ---- loader state: the K-tile whose half-tiles are being issued --------------------------
    int l_tile = 0, l_sg = 0, l_ti = 0, l_kc = 0, l_nkc = 0, l_dy = 0, l_dx = 0;
    while (ntaps_l[l_sg] == 0) ++l_sg;
    unsigned aoff[2][2], boff[2][2];
    const char* abase = nullptr;
    const char* bbase = nullptr;
    auto l_setup_seg = [&]() {
      const SegDev& s = p.seg[l_sg];
      l_nkc = s.ck_b / BKB;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          aoff[h][i] = (unsigned)a_row[h][i] * (unsigned)s.w_ld_b + lchunk;
          boff[h][i] = (unsigned)b_pix[h][i] * (unsigned)s.in_ld_b + lchunk;
        }
    };
    auto l_setup_tap = [&]() {
      const SegDev& s = p.seg[l_sg];
      const int tap = taps_l[l_sg * 9 + l_ti];
      const int R = s.R;
      const int dil = s.dil + (int)bz * p.dil_step;
      l_dy = (tap / R - (R >> 1)) * dil;
      l_dx = (tap % R - (R >> 1)) * dil;
      abase = s.w + bz * s.w_batch_b + (long long)tap * s.w_tap_b + (long long)co0 * s.w_ld_b + (long long)l_kc * BKB;
      bbase = s.in + bz * s.in_batch_b + ((ABL == 8 ? 1024 : m0) + (long long)l_dy * p.W + l_dx) * s.in_ld_b + (long long)l_kc * BKB;
    };
    auto l_advance = [&]() {
      if (++l_tile >= total) return;
      if (++l_ti >= ntaps_l[l_sg]) {
        l_ti = 0;
        if (++l_kc >= l_nkc) {
          l_kc = 0;
          ++l_sg;
          while (ntaps_l[l_sg] == 0) ++l_sg;
          l_setup_seg();
        }
      }
      l_setup_tap();
    };
    const char* zsrc = reinterpret_cast<const char*>(pp_zero_page) + ((lane & 7) << 4);
    // issue this wavefront's two pieces of one half-tile: sub 0 = B0, 1 = A0, 2 = B1, 3 = A1 of the loader's K-tile
    auto issue = [&](int sub, int slot) {
      char* dst = smem + slot * PP_SLOT + wave * 1024;
      if (ABL == 2 || ABL == 6) return;
      const bool live = ABL == 1 ? false : l_tile < total;
      const int h = sub >> 1;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const char* src;
        if (sub & 1) {
          src = live ? abase + aoff[h][i] : zsrc;
        } else {
          const bool ok = live && b_in[h][i] && (unsigned)(b_y[h][i] + l_dy) < (unsigned)p.H &&
                          (unsigned)(b_x[h][i] + l_dx) < (unsigned)p.W;
          src = ok ? bbase + boff[h][i] : zsrc;
        }
        glds16(src, dst + i * 8192);
      }
    };

    // ---- fragment addressing ------------------------------------------------------------------
    const int frow = lane & 15, fg = lane >> 4;
    const int fsw = (fg ^ (frow & 7)) << 4;
    const int fa_off = (wr * 64 + frow) * 128 + fsw;   // + i*2048, ^ (ks << 6)
    const int fb_off = (wc * 32 + frow) * 128 + fsw;   // + j*2048, ^ (ks << 6)
    u32x4 fa[4][2], fb1[2][2], fb0[2][2][2];

    auto read_a = [&](int slot) {
      if (ABL == 4 || ABL == 6) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+v"(fa[i][ks]));
        return;
      }
      const char* base = smem + slot * PP_SLOT;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          fa[i][ks] = *reinterpret_cast<const u32x4*>(base + ((fa_off + i * 2048) ^ (ks << 6)));
    };
    auto read_b = [&](int slot, u32x4 (&dst)[2][2]) {
      if (ABL == 4 || ABL == 6) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+v"(dst[j][ks]));
        return;
      }
      const char* base = smem + slot * PP_SLOT;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          dst[j][ks] = *reinterpret_cast<const u32x4*>(base + ((fb_off + j * 2048) ^ (ks << 6)));
    };
    auto mma_quad = [&](int i0, int j0, const u32x4 (&b)[2][2]) {
      if (ABL == 3) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(fa[i][ks]));
#pragma unroll
          for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(b[j][ks]));
        }
        return;
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) Mma<bf16_t>::run(fa[i][ks], b[j][ks], acc[i0 + i][j0 + j]);
    };

    // ---- prologue: half-tiles of data index -1 .. 5 (issue numbers 0 .. 6) -----------------------
    l_setup_seg();
    l_setup_tap();
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      issue(k & 3, (k + 7) & 7);
      if ((k & 3) == 3) l_advance();
    }
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");   // B0(0) and A0(0) have landed
    __builtin_amdgcn_s_barrier();
    read_b(7, fb0[0]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (wr == 1) __builtin_amdgcn_s_barrier();          // second group runs one barrier behind

    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
    auto stamp = [&](int k) {
      if (ABL != 9) return;
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      if (k >= 0) tacc[k] += now - tprev;
      tprev = now;
    };
    stamp(-1);
    // one phase; P = phase number modulo 8 (compile-time)
#define SEGK_PHASE(P)                                                                          \
  {                                                                                            \
    stamp(5);                                                                                  \
    if (((P) & 3) == 0) read_a((P) & 7);                                                       \
    if (((P) & 3) == 1) read_b((P) & 7, fb1);                                                  \
    if (((P) & 3) == 2) read_a((P) & 7);                                                       \
    if (((P) & 3) == 3) read_b((P) & 7, fb0[1 - ((P) >> 2)]);                                  \
    issue(((P) + 3) & 3, ((P) + 6) & 7);                                                       \
    if (((P) & 3) == 0) l_advance();                                                           \
    stamp(0);                                                                                  \
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");                                          \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    stamp(1);                                                                                  \
    __builtin_amdgcn_s_barrier();                                                              \
    stamp(2);                                                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                         \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    stamp(3);                                                                                  \
    __builtin_amdgcn_s_setprio(1);                                                             \
    if (((P) & 3) == 0) mma_quad(0, 0, fb0[(P) >> 2]);                                         \
    if (((P) & 3) == 1) mma_quad(0, 2, fb1);                                                   \
    if (((P) & 3) == 2) mma_quad(4, 2, fb1);                                                   \
    if (((P) & 3) == 3) mma_quad(4, 0, fb0[(P) >> 2]);                                         \
    __builtin_amdgcn_s_setprio(0);                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    stamp(4);                                                                                  \
    __builtin_amdgcn_s_barrier();                                                              \
  }
    for (int t2 = 0; t2 < (total >> 1); ++t2) {
      SEGK_PHASE(0) SEGK_PHASE(1) SEGK_PHASE(2) SEGK_PHASE(3)
      SEGK_PHASE(4) SEGK_PHASE(5) SEGK_PHASE(6) SEGK_PHASE(7)
    }
    if (total & 1) {
      SEGK_PHASE(0) SEGK_PHASE(1) SEGK_PHASE(2) SEGK_PHASE(3)
    }
#undef SEGK_PHASE
    if (wr == 0) __builtin_amdgcn_s_barrier();          // pairs with the second group's last barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the dummy fetches of the last phases
    if (ABL == 9 && p.stat && lane == 0) {
      // [tile][wave][6] sums of cycles: load section, vmcnt wait, barrier 1, lgkmcnt wait, MFMA section, barrier 2
      for (int k = 0; k < 6; ++k) p.stat[((size_t)mt * 2) * p.Cout + wave * 8 + k] = (double)tacc[k];
      p.stat[((size_t)mt * 2) * p.Cout + wave * 8 + 6] = (double)total;
    }
  }
  __syncthreads();                                      // the epilogue reuses the ring for its reduction
  if (ABL == 9) return;
  conv_epilogue<TO, G>(p, acc, smem, wr, wc, lane, tid, co0, m0, mt, bz);
}

// ------------------------------------------------------------------------------------------------
// Weight gradient
// ------------------------------------------------------------------------------------------------
struct WgradDev {
  int H, W, HW;
  long long M;
  const char* in;
  const char* dout;
  long long in_ld_b, dout_ld_b;
  int Cin, Cout, R, dil, splits;
  int co_tiles, ci_tiles;
  long long per_split;  // pixels per split (multiple of the pixel chunk)
  float* slabs;
  long long in_batch_b, dout_batch_b;  // batched GEMM use (blockIdx.y); 0 for convolutions
  // groups > 0 (ring tile only): blockIdx.y names one of several convolutions of the same shape, each with its own
  // operands; its slabs follow those of the convolution before it
  int groups;
  const char* in_g[SEGK_WGRAD_GROUP_MAX];
  const char* dout_g[SEGK_WGRAD_GROUP_MAX];
};

// Tile = [BP pixels][128 channels]; row bytes RB = 128*sizeof(T).
// bf16: 32-byte granules XOR-swizzled by f(row) so that the transposing reads of 8 pixel rows x 32 bytes
//       cover one 256-byte bank window exactly; fp32: 16-float halves swapped on odd rows for ds_read_b32.
template <typename T>
__device__ __forceinline__ int wg_chunk_off(int row, int c16);  // byte offset of 16-byte chunk c16 of row
template <>
__device__ __forceinline__ int wg_chunk_off<bf16_t>(int row, int c16) {
  const int f = (row & 3) | ((row >> 1) & 4);
  return row * 256 + ((((c16 >> 1) ^ f) & 7) << 5) + ((c16 & 1) << 4);
}
template <>
__device__ __forceinline__ int wg_chunk_off<float>(int row, int c16) {
  return row * 512 + ((c16 ^ ((row & 1) << 2)) << 4);
}

template <typename T>
struct WgFrag;
// bf16: one MFMA k-step = 32 pixels.  Fragment for 16 channels starting at ch0 (multiple of 16).
template <>
struct WgFrag<bf16_t> {
  static constexpr int BP = 64;       // pixels per tile
  static constexpr int KSTEPS = 2;    // 32 pixels each
  static __device__ __forceinline__ u32x4 read(const char* tile, int ks, int ch0, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int c32 = ch0 >> 4;
    u32x4 out;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = ks * 32 + g * 8 + h * 4 + q;
      const int f = (row & 3) | ((row >> 1) & 4);
      const int off = row * 256 + (((c32 ^ f) & 7) << 5) + pp * 8;
      s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(__attribute__((address_space(3))) char*)(tile + off));
      u32x2 u = __builtin_bit_cast(u32x2, v);
      out[2 * h] = u[0];
      out[2 * h + 1] = u[1];
    }
    return out;
  }
  static __device__ __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
// fp32: one MFMA k-step = 4 pixels; the "fragment" packs 4 consecutive k-steps (16 pixels) in a u32x4.
template <>
struct WgFrag<float> {
  static constexpr int BP = 32;
  static constexpr int KSTEPS = 2;    // 16 pixels each (4 MFMAs)
  static __device__ __forceinline__ u32x4 read(const char* tile, int ks, int ch0, int lane) {
    const int g = lane >> 4, col = ch0 + (lane & 15);
    u32x4 out;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = ks * 16 + e * 4 + g;
      const int off = row * 512 + ((col ^ ((row & 1) << 4)) << 2);
      out[e] = *reinterpret_cast<const unsigned int*>(tile + off);
    }
    return out;
  }
  static __device__ __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x4& c) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[e]), __uint_as_float(b[e]), c, 0, 0, 0);
  }
};

// NSUB = 1: 128(co) x 128(ci) tile, 4 wavefronts of 64 x 64, two workgroups per CU.
// NSUB = 2: 256 x 256 tile (each operand tile = two 128-channel sub-tiles side by side), 8 wavefronts of
//           128(co) x 64(ci), one workgroup per CU: twice the MFMA work per byte that crosses the vector L1.
template <typename T, int NSUB>
__global__ __launch_bounds__(CONV_THREADS * NSUB, NSUB == 1 ? 2 : 1) void conv_wgrad_kernel(const WgradDev p) {
  constexpr int NT = CONV_THREADS * NSUB;
  constexpr int TCH = TILE * NSUB;               // channels per operand tile
  constexpr int BP = WgFrag<T>::BP;
  constexpr int RB = TILE * (int)sizeof(T);      // sub-tile row bytes
  constexpr int CHS = RB / 16;                   // 16-byte chunks per sub-tile row (16 bf16 / 32 fp32)
  constexpr int CH = CHS * NSUB;                 // chunks per operand-tile row
  constexpr int RPP = NT / CH;                   // rows per pass
  constexpr int NP = BP / RPP;                   // passes (4 in every variant)
  constexpr int SUB_B = BP * RB;                 // 16 KB per sub-tile
  constexpr int TILE_B = SUB_B * NSUB;
  constexpr int MI = 4 * NSUB, MJ = 4;           // MFMA tiles per wavefront: co x ci
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 stages][dout tile | in tile]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = NSUB == 1 ? wave >> 1 : wave >> 2;   // co block of MI*16 channels
  const int wc = NSUB == 1 ? wave & 1 : wave & 3;     // ci block of 64 channels
  // each XCD (workgroups go to them round-robin by block index) gets a contiguous run of tiles: the tiles and taps
  // of one pixel range, which read the same rows of both operands, then share an L2
  int b;
  {
    const unsigned nb = gridDim.x, q = nb >> 3, r = nb & 7u, xcd = blockIdx.x & 7u;
    b = (int)(xcd * q + (xcd < r ? xcd : r) + (blockIdx.x >> 3));
  }
  const int cot = b % p.co_tiles; b /= p.co_tiles;
  const int cit = b % p.ci_tiles; b /= p.ci_tiles;
  const int taps = p.R * p.R;
  const int tap = b % taps;
  const int sp = b / taps;
  const int co0 = cot * TCH, ci0 = cit * TCH;
  const int dy = (tap / p.R - (p.R >> 1)) * p.dil, dx = (tap % p.R - (p.R >> 1)) * p.dil;
  const long long pbeg = (long long)sp * p.per_split;
  long long pend = pbeg + p.per_split;
  if (pend > p.M) pend = p.M;
  const int nchunks = pbeg < pend ? (int)((pend - pbeg + BP - 1) / BP) : 0;

  const int lc = tid % CH, lr = tid / CH;
  const int lsub = lc / CHS, lcs = lc % CHS;     // sub-tile and chunk within it
  const int esz = (int)sizeof(T);
  const int ch_per_chunk = 16 / esz;
  const bool d_ok = co0 + lc * ch_per_chunk < p.Cout;  // Cout, Cin are multiples of 16 / 32 elements
  const bool x_ok = ci0 + lc * ch_per_chunk < p.Cin;
  const long long bz = blockIdx.y;
  const char* dbase = p.dout + bz * p.dout_batch_b + (size_t)(co0 + lc * ch_per_chunk) * esz;
  const char* xbase = p.in + bz * p.in_batch_b + (size_t)(ci0 + lc * ch_per_chunk) * esz;

  f32x4 acc[MI][MJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < MJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // wavefronts whose input or output channels lie entirely beyond the tensor (partial last tile) skip the MFMA work
  const bool wave_live = ci0 + wc * 64 < p.Cin && co0 + wr * (MI * 16) < p.Cout;
  u32x4 rd[NP], rx[NP];
  // per loader row: pixel index, (y, x) and source addresses, advanced by one chunk per call with additions and a
  // wrap test (a division pair and two 64-bit multiplications per row and chunk would cost as much as the MFMAs)
  long long mrow[NP];
  int my[NP], mx[NP];
  const char* dptr[NP];
  const char* xptr[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    mrow[i] = pbeg + lr + i * RPP;
    const int rem = pixel_in_image(mrow[i] < p.M ? mrow[i] : 0, p.HW);
    my[i] = rem / p.W;
    mx[i] = rem - my[i] * p.W;
    dptr[i] = dbase + mrow[i] * p.dout_ld_b;
    xptr[i] = xbase + (mrow[i] + (long long)dy * p.W + dx) * p.in_ld_b;
  }
  const int step_y = BP / p.W, step_x = BP - step_y * p.W;
  const long long dstep = (long long)BP * p.dout_ld_b, xstep = (long long)BP * p.in_ld_b;
  // load the next chunk (chunks are taken in order); returns whether any of this thread's shifted rows is inside
  // the image
  auto load_chunk = [&](int) -> int {
    int any = 0;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      u32x4 z = {0u, 0u, 0u, 0u};
      rd[i] = z;
      rx[i] = z;
      if (mrow[i] < pend && (unsigned)(my[i] + dy) < (unsigned)p.H && (unsigned)(mx[i] + dx) < (unsigned)p.W) {
        any = 1;
        if (d_ok) rd[i] = *reinterpret_cast<const u32x4*>(dptr[i]);
        if (x_ok) rx[i] = *reinterpret_cast<const u32x4*>(xptr[i]);
      }
      mrow[i] += BP;
      dptr[i] += dstep;
      xptr[i] += xstep;
      mx[i] += step_x;
      my[i] += step_y;
      if (mx[i] >= p.W) {
        mx[i] -= p.W;
        ++my[i];
      }
      while (my[i] >= p.H) my[i] -= p.H;   // next image (several at once only for images smaller than a chunk)
    }
    return any;
  };
  auto store_chunk = [&](int stage) {
    char* sd = smem + stage * 2 * TILE_B + lsub * SUB_B;
    char* sx = sd + TILE_B;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int off = wg_chunk_off<T>(lr + i * RPP, lcs);
      *reinterpret_cast<u32x4*>(sd + off) = rd[i];
      *reinterpret_cast<u32x4*>(sx + off) = rx[i];
    }
  };

  if (nchunks > 0) {
    int any = load_chunk(0);
    store_chunk(0);
    int cur_any = __syncthreads_or(any);
    for (int ck = 0; ck < nchunks; ++ck) {
      const int cur = ck & 1;
      const bool more = ck + 1 < nchunks;
      any = 0;
      if (more) any = load_chunk(ck + 1);
      if (cur_any && wave_live) {
        // this wavefront's co channels lie in one sub-tile (NSUB = 2: wr selects it), its ci channels too
        const char* sd = smem + cur * 2 * TILE_B + (NSUB == 1 ? 0 : wr * SUB_B);
        const char* sx = smem + cur * 2 * TILE_B + TILE_B + (NSUB == 1 ? 0 : (wc >> 1) * SUB_B);
        const int cob = NSUB == 1 ? wr * 64 : 0;
        const int cib = NSUB == 1 ? wc * 64 : (wc & 1) * 64;
#pragma unroll
        for (int ks = 0; ks < WgFrag<T>::KSTEPS; ++ks) {
          u32x4 fb[MJ];
#pragma unroll
          for (int j = 0; j < MJ; ++j) fb[j] = WgFrag<T>::read(sx, ks, cib + j * 16, lane);
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            const u32x4 fa = WgFrag<T>::read(sd, ks, cob + i * 16, lane);
#pragma unroll
            for (int j = 0; j < MJ; ++j) WgFrag<T>::mma(fa, fb[j], acc[i][j]);
          }
        }
      }
      if (more) store_chunk(cur ^ 1);
      cur_any = __syncthreads_or(any);
    }
  }

  // acc[i][j][r] = dw[co = co0 + wr*MI*16 + i*16 + (lane>>4)*4 + r][ci = ci0 + wc*64 + j*16 + (lane&15)]
  const int fq = lane >> 4, fr = lane & 15;
  float* slab = p.slabs + (((size_t)bz * p.splits + sp) * taps + tap) * p.Cin * p.Cout;
#pragma unroll
  for (int j = 0; j < MJ; ++j) {
    const int ci = ci0 + wc * 64 + j * 16 + fr;
    if (ci >= p.Cin) continue;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int co = co0 + wr * (MI * 16) + i * 16 + fq * 4;
      if (co >= p.Cout) continue;
      *reinterpret_cast<f32x4*>(slab + (size_t)ci * p.Cout + co) = acc[i][j];
    }
  }
}

// Weight gradient, 128 x 128 tile through LDS-DMA (bf16): one 32 KB stage ([64 px][128 co] + [64 px][128 ci])
// filled by global_load_lds, two barriers per 64-pixel chunk, about 100 VGPRs - three workgroups per CU cover each
// other's load phases.  Same LDS images, fragments and slab output as conv_wgrad_kernel<bf16, 1>.
__global__ __launch_bounds__(256, 3) void conv_wgrad_dma_kernel(const WgradDev p) {
  typedef bf16_t T;
  constexpr int BP = WgFrag<T>::BP;              // 64 pixels per chunk
  constexpr int SUB_B = BP * 256;                // 16 KB per operand image
  constexpr int MI = 4, MJ = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [dout image | in image]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  // each XCD (workgroups go to them round-robin by block index) gets a contiguous run of tiles: the tiles and taps
  // of one pixel range, which read the same rows of both operands, then share an L2
  int b;
  {
    const unsigned nb = gridDim.x, q = nb >> 3, r = nb & 7u, xcd = blockIdx.x & 7u;
    b = (int)(xcd * q + (xcd < r ? xcd : r) + (blockIdx.x >> 3));
  }
  const int cot = b % p.co_tiles; b /= p.co_tiles;
  const int cit = b % p.ci_tiles; b /= p.ci_tiles;
  const int taps = p.R * p.R;
  const int tap = b % taps;
  const int sp = b / taps;
  const int co0 = cot * TILE, ci0 = cit * TILE;
  const int dy = (tap / p.R - (p.R >> 1)) * p.dil, dx = (tap % p.R - (p.R >> 1)) * p.dil;
  const long long pbeg = (long long)sp * p.per_split;
  long long pend = pbeg + p.per_split;
  if (pend > p.M) pend = p.M;
  const int nchunks = pbeg < pend ? (int)((pend - pbeg + BP - 1) / BP) : 0;
  const long long bz = blockIdx.y;

  // loader geometry: piece q = i*4 + wave (i = 0..3) covers image rows (pixels) q*4 .. q*4+3; lane l -> row
  // q*4 + (l>>4), LDS slot l&15 of that row, which holds the source chunk the granule swizzle assigns to it
  const int lr4 = lane >> 4, slot = lane & 15;
  int lrow[4], csrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    lrow[i] = (i * 4 + wave) * 4 + lr4;
    const int f = (lrow[i] & 3) | ((lrow[i] >> 1) & 4);
    csrc[i] = ((((slot >> 1) ^ f) & 7) << 1) | (slot & 1);   // inverse of wg_chunk_off<bf16_t>
  }
  const char* dbase = p.dout + bz * p.dout_batch_b + (size_t)co0 * 2;
  const char* xbase = p.in + bz * p.in_batch_b + (size_t)ci0 * 2;
  const char* zsrc = reinterpret_cast<const char*>(pp_zero_page) + (slot << 4);

  f32x4 acc[MI][MJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < MJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool wave_live = ci0 + wc * 64 < p.Cin && co0 + wr * 64 < p.Cout;
  char* const sd_w = smem + wave * 1024;            // + i * 4096
  char* const sx_w = smem + SUB_B + wave * 1024;

  // Per loader row: pixel index, its (y, x) and the two source addresses, advanced by one chunk (BP pixels) per trip
  // with additions and one wrap test - a division pair and two 64-bit multiplications per row and chunk cost more
  // issue slots than the chunk's MFMAs.
  long long mrow[4];
  int my[4], mx[4];
  const char* dptr[4];
  const char* xptr[4];
  bool dcol[4], xcol[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    mrow[i] = pbeg + lrow[i];
    const int rem = pixel_in_image(mrow[i] < p.M ? mrow[i] : 0, p.HW);
    my[i] = rem / p.W;
    mx[i] = rem - my[i] * p.W;
    dcol[i] = co0 + csrc[i] * 8 < p.Cout;
    xcol[i] = ci0 + csrc[i] * 8 < p.Cin;
    dptr[i] = dbase + mrow[i] * p.dout_ld_b + csrc[i] * 16;
    xptr[i] = xbase + (mrow[i] + (long long)dy * p.W + dx) * p.in_ld_b + csrc[i] * 16;
  }
  const int step_y = BP / p.W, step_x = BP - step_y * p.W;
  const long long dstep = (long long)BP * p.dout_ld_b, xstep = (long long)BP * p.in_ld_b;

  for (int ck = 0; ck < nchunks; ++ck) {
    int any = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const char* ds = zsrc;
      const char* xs = zsrc;
      if (mrow[i] < pend && (unsigned)(my[i] + dy) < (unsigned)p.H && (unsigned)(mx[i] + dx) < (unsigned)p.W) {
        any = 1;
        if (dcol[i]) ds = dptr[i];
        if (xcol[i]) xs = xptr[i];
      }
      glds16(ds, sd_w + i * 4096);
      glds16(xs, sx_w + i * 4096);
      mrow[i] += BP;
      dptr[i] += dstep;
      xptr[i] += xstep;
      mx[i] += step_x;
      my[i] += step_y;
      if (mx[i] >= p.W) {
        mx[i] -= p.W;
        ++my[i];
      }
      while (my[i] >= p.H) my[i] -= p.H;   // next image (several at once only for images smaller than a chunk)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int cur_any = __syncthreads_or(any);
    if (cur_any && wave_live) {
      const char* sd = smem;
      const char* sx = smem + SUB_B;
#pragma unroll
      for (int ks = 0; ks < WgFrag<T>::KSTEPS; ++ks) {
        u32x4 fb[MJ];
#pragma unroll
        for (int j = 0; j < MJ; ++j) fb[j] = WgFrag<T>::read(sx, ks, wc * 64 + j * 16, lane);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const u32x4 fa = WgFrag<T>::read(sd, ks, wr * 64 + i * 16, lane);
#pragma unroll
          for (int j = 0; j < MJ; ++j) WgFrag<T>::mma(fa, fb[j], acc[i][j]);
        }
      }
    }
    __syncthreads();
  }

  const int fq = lane >> 4, fr = lane & 15;
  float* slab = p.slabs + (((size_t)bz * p.splits + sp) * taps + tap) * p.Cin * p.Cout;
#pragma unroll
  for (int j = 0; j < MJ; ++j) {
    const int ci = ci0 + wc * 64 + j * 16 + fr;
    if (ci >= p.Cin) continue;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int co = co0 + wr * 64 + i * 16 + fq * 4;
      if (co >= p.Cout) continue;
      *reinterpret_cast<f32x4*>(slab + (size_t)ci * p.Cout + co) = acc[i][j];
    }
  }
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// hipFuncSetAttribute is per device: remember, per kernel instantiation, the devices it has been done on (a bit per
// device ordinal; two threads racing here both set the same value, which is harmless).
struct LdsConfigured {
  std::atomic<unsigned long long> devices{0ull};
};

template <typename K>
static int set_lds(K kernel, int bytes, const char* name, LdsConfigured& done) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) {
    set_error("%s: hipGetDevice: %s", name, hipGetErrorString(e));
    return SEGK_HIP_ERROR;
  }
  const unsigned long long bit = 1ull << (dev & 63);
  if (dev < 64 && (done.devices.load(std::memory_order_acquire) & bit)) return SEGK_OK;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    set_error("%s: hipFuncSetAttribute(%d bytes LDS): %s", name, bytes, hipGetErrorString(e));
    return SEGK_HIP_ERROR;
  }
  if (dev < 64) done.devices.fetch_or(bit, std::memory_order_release);
  return SEGK_OK;
}

static int launch_wgrad_ring(WgradDev& d, int batch, hipStream_t st, const char* name);

// Launch the weight-gradient kernel; the 256 x 256 tile is used when the output channels fill it exactly.
// tile = 384: the 256(co) x 128(ci) ring tile (bf16).
template <typename T>
static int launch_wgrad(WgradDev& d, int batch, hipStream_t st, const char* name, int tile = 0) {
  if (tile == 384 && sizeof(T) != 2) {
    set_error("%s: the ring tile is bf16 only", name);
    return SEGK_UNSUPPORTED_SHAPE;
  }
  // bf16 onto a multiple of 256 output channels: the ring tile unless the caller names another
  if (tile == 384 || (tile == 0 && sizeof(T) == 2 && d.Cout % 256 == 0 && !experiment("SEGK_WGRAD_TILE")))
    return launch_wgrad_ring(d, batch, st, name);
  static const char* force = experiment("SEGK_WGRAD_TILE");  // experiments: "small" (register-staged 128 tile) / "big" / "dma"
  // The 256 x 256 tile when the output channels fill it exactly and the last input-channel tile is more than half
  // full (320 input channels would run a second tile that is a quarter full); otherwise the 128 x 128 tile, for
  // bf16 the LDS-DMA one.
  bool big = d.Cin >= 256 && d.Cout % 256 == 0 && (d.Cin % 256 == 0 || d.Cin % 256 > 128);
  if (force && force[0] == 's') big = false;
  if (force && force[0] == 'b') big = d.Cin >= 256 && d.Cout % 256 == 0;
  if (force && force[0] == 'd') big = false;   // "dma": the 128 x 128 LDS-DMA tile everywhere (bf16)
  if (tile == 128) big = false;
  if (tile == 256) big = d.Cin >= 256 && d.Cout % 256 == 0;
  if (sizeof(T) == 2 && !big && !(force && force[0] == 's')) {
    d.co_tiles = ceil_div(d.Cout, TILE);
    d.ci_tiles = ceil_div(d.Cin, TILE);
    const long long nb = (long long)d.co_tiles * d.ci_tiles * d.R * d.R * d.splits;
    if (nb >= 2147483647LL) {
      set_error("%s: too many tiles", name);
      return SEGK_INVALID_ARG;
    }
    conv_wgrad_dma_kernel<<<dim3((unsigned)nb, (unsigned)batch), 256, 2 * 16384, st>>>(d);
    return SEGK_OK;
  }
  const int tch = big ? 2 * TILE : TILE;
  d.co_tiles = ceil_div(d.Cout, tch);
  d.ci_tiles = ceil_div(d.Cin, tch);
  const long long blocks = (long long)d.co_tiles * d.ci_tiles * d.R * d.R * d.splits;
  if (blocks >= 2147483647LL) {
    set_error("%s: too many tiles", name);
    return SEGK_INVALID_ARG;
  }
  const int lds = 4 * 16384 * (big ? 2 : 1);
  int rc;
  static LdsConfigured configured_big, configured_small;  // per instantiation
  if (big) {
    rc = set_lds(conv_wgrad_kernel<T, 2>, lds, name, configured_big);
    if (rc != SEGK_OK) return rc;
    conv_wgrad_kernel<T, 2><<<dim3((unsigned)blocks, (unsigned)batch), CONV_THREADS * 2, lds, st>>>(d);
  } else {
    rc = set_lds(conv_wgrad_kernel<T, 1>, lds, name, configured_small);
    if (rc != SEGK_OK) return rc;
    conv_wgrad_kernel<T, 1><<<dim3((unsigned)blocks, (unsigned)batch), CONV_THREADS, lds, st>>>(d);
  }
  return SEGK_OK;
}

// px_tiles < 0: every pixel tile from d.m_begin to the end
template <typename T, int BKB, typename TO, typename G, int ABL = 0>
static int launch_fwd(ConvDev& d, hipStream_t st, int batch = 1, long long px_tiles = -1) {
  const int lds = 2 * (G::BCO + G::BPX) * BKB;
  static LdsConfigured configured;  // per instantiation
  int rc = set_lds(conv_fwd_kernel<T, BKB, TO, G, ABL>, lds, "conv_fwd", configured);
  if (rc != SEGK_OK) return rc;
  d.n_tiles = ceil_div(d.Cout, G::BCO);
  if (px_tiles < 0) px_tiles = (d.M - d.m_begin + G::BPX - 1) / G::BPX;
  const long long blocks = px_tiles * d.n_tiles;
  if (blocks >= 2147483647LL) {
    set_error("conv: too many tiles");
    return SEGK_INVALID_ARG;
  }
  conv_fwd_kernel<T, BKB, TO, G, ABL><<<dim3((unsigned)blocks, (unsigned)batch), G::THREADS, lds, st>>>(d);
  return SEGK_OK;
}

// ------------------------------------------------------------------------------------------------
// Forward / data gradient, small tile with LDS-DMA: 128 x 128 tile, bf16 operands, 128-byte K rows, one LDS stage
// of 32 KB filled by global_load_lds (no staging registers), two barriers per tap step.  Nothing is overlapped
// inside a workgroup on purpose: with ~100 VGPRs and 32 KB of LDS three to four workgroups share a CU, and while
// one waits for its loads the others multiply.
// ------------------------------------------------------------------------------------------------
// PX = pixels per tile: 128 (wave tile 64 x 64, three workgroups per CU) or 256 (64 x 128, two per CU)

// STAGES = 2 (experiment builds only): two LDS stages, the loads of step k+1 are in flight under the MFMAs of step k
// and one barrier per step remains; 64 KB of LDS, two workgroups per CU.
// STAGES = 3: the two-stage tile with SPECIALISED wavefronts, 512 threads: wavefronts 0-3 only issue the LDS-DMA of
// step k+1 (an issuing wavefront is held ~120 cycles per 1-KiB global_load_lds, so it cannot put its own loads under
// its own MFMAs), wavefronts 4-7 only read fragments and issue the MFMAs of step k; one barrier per step joins them.
// For launches of at most two workgroups per CU (the 274-tile convolutions on 33 x 33 maps), where no other workgroup
// is there to multiply while one loads.
template <typename TO, int PX, int STAGES = 1>
__global__ __launch_bounds__(STAGES == 3 ? 512 : 256, (PX == 128 && STAGES == 1) ? 3 : 2) void conv_fwd_dma_kernel(const ConvDev p) {
  typedef TileCfg<128, PX, 2, 2> G;
  constexpr int BKB = 128, MI = G::MI, MJ = G::MJ;
  constexpr int NB = PX / 32;   // activation pieces per wavefront
  constexpr int STAGE_B = (128 + PX) * 128;
  constexpr bool SPEC = STAGES == 3;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // stages x ([A 128 rows][B PX rows] x 128 B, swizzled)
  __shared__ int taps_l[SEGK_MAX_SEG * 9];
  __shared__ int ntaps_l[SEGK_MAX_SEG];
  __shared__ unsigned int mask_l[2];
  // per listed tap, filled once: byte offsets of its weight slice and of its shifted pixel window, and the shift
  // itself - the K loop then needs two table reads and two additions per step instead of two divisions, a run of
  // 64-bit multiplications and a dynamically indexed read of the segment descriptor
  __shared__ long long tap_a_l[SEGK_MAX_SEG * 9];
  __shared__ long long tap_b_l[SEGK_MAX_SEG * 9];
  __shared__ int tap_dy_l[SEGK_MAX_SEG * 9];
  __shared__ int tap_dx_l[SEGK_MAX_SEG * 9];

  // specialised form: threads 0-255 load (and own the tile-row geometry below), threads 256-511 multiply
  const bool loader = !SPEC || threadIdx.x < 256;
  const bool multiplier = !SPEC || threadIdx.x >= 256;
  const int tid = SPEC ? (int)(threadIdx.x & 255) : (int)threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave / G::WPX, wc = wave % G::WPX;
  // Workgroups go to the eight XCDs round-robin by block index; give each XCD a contiguous run of tiles, so that
  // the output-channel tiles of one pixel tile and the neighbouring pixel tiles (which share the rows the 3x3 taps
  // reach into) meet in the same L2.
  unsigned tile;
  {
    const unsigned nb = gridDim.x, q = nb >> 3, r = nb & 7u, xcd = blockIdx.x & 7u;
    tile = xcd * q + (xcd < r ? xcd : r) + (blockIdx.x >> 3);
  }
  const int nt = tile % p.n_tiles;
  const long long mt = tile / p.n_tiles;
  const int co0 = nt * G::BCO;
  const long long m0 = p.m_begin + mt * G::BPX;
  const long long bz = blockIdx.y;

  // loader geometry: piece q = i*4 + wave (i = 0..3) covers tile rows q*8 .. q*8+7; lane l -> row q*8 + (l>>3),
  // LDS slot l&7, which holds source chunk (l&7) ^ (row&7)
  const int lrow = lane >> 3;
  const int lchunk = ((lane & 7) ^ lrow) << 4;
  int wrow[4], ry[NB], rx[NB];
  bool rvalid[NB];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int co = co0 + (i * 4 + wave) * 8 + lrow;
    wrow[i] = (co < p.Cout ? co : p.Cout - 1) - co0;
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int r = (i * 4 + wave) * 8 + lrow;
    const long long m = m0 + r;
    rvalid[i] = m < p.M;
    const int rem = pixel_in_image(rvalid[i] ? m : 0, p.HW);
    ry[i] = rem / p.W;
    rx[i] = rem - ry[i] * p.W;
  }

  if (threadIdx.x < 2) mask_l[threadIdx.x] = 0u;
  __syncthreads();
  if (loader) {
    unsigned long long mine = 0ull;
    for (int sg = 0; sg < p.nseg; ++sg) {
      const int R = p.seg[sg].R, dil = p.seg[sg].dil + (int)blockIdx.y * p.dil_step;
      for (int tap = 0; tap < R * R; ++tap) {
        const int dy = (tap / R - (R >> 1)) * dil, dx = (tap % R - (R >> 1)) * dil;
        bool any = false;
#pragma unroll
        for (int i = 0; i < NB; ++i)
          any |= rvalid[i] && (unsigned)(ry[i] + dy) < (unsigned)p.H && (unsigned)(rx[i] + dx) < (unsigned)p.W;
        if (any) mine |= 1ull << (sg * 9 + tap);
      }
    }
    const unsigned int lo = (unsigned int)mine, hi = (unsigned int)(mine >> 32);
    if (lo) atomicOr(&mask_l[0], lo);
    if (hi) atomicOr(&mask_l[1], hi);
  }
  __syncthreads();
  if (threadIdx.x < SEGK_MAX_SEG * 9) {
    // thread (segment, tap): a tap that touches the tile takes the list slot given by the number of listed taps
    // before it
    const unsigned long long all = ((unsigned long long)mask_l[1] << 32) | mask_l[0];
    const int sg = threadIdx.x / 9, tap = threadIdx.x - sg * 9;
    if (sg < p.nseg) {
      const unsigned seg_bits = (unsigned)(all >> (sg * 9)) & 0x1ffu;
      if (tap == 0) ntaps_l[sg] = __popc(seg_bits);
      if ((seg_bits >> tap) & 1u) {
        const SegDev& sd = p.seg[sg];
        const int slot = sg * 9 + __popc(seg_bits & ((1u << tap) - 1u));
        const int R = sd.R, dil = sd.dil + (int)blockIdx.y * p.dil_step;
        const int dy = (tap / R - (R >> 1)) * dil, dx = (tap % R - (R >> 1)) * dil;
        taps_l[slot] = tap;
        tap_a_l[slot] = (long long)tap * sd.w_tap_b;
        tap_b_l[slot] = ((long long)dy * p.W + dx) * sd.in_ld_b;
        tap_dy_l[slot] = dy;
        tap_dx_l[slot] = dx;
      }
    }
  }
  __syncthreads();
  int total = 0;
  for (int sg = 0; sg < p.nseg; ++sg) total += ntaps_l[sg] * (p.seg[sg].ck_b / BKB);
  if (p.abl & 16) total = 0;

  f32x4 acc[MI][MJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < MJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (total > 0) {
    int sg = 0;
    while (ntaps_l[sg] == 0) ++sg;
    int ti = 0, kc = 0, nkc = p.seg[sg].ck_b / BKB;
    unsigned aoff[4], boff[NB];
    bool bok[NB];
    const char* abase = nullptr;
    const char* bbase = nullptr;
    const char* aseg = nullptr;   // weight rows of this tile / pixel rows of this tile in the current segment
    const char* bseg = nullptr;
    auto setup_seg = [&]() {
      const SegDev& s = p.seg[sg];
#pragma unroll
      for (int i = 0; i < 4; ++i) aoff[i] = (unsigned)wrow[i] * (unsigned)s.w_ld_b + lchunk;
#pragma unroll
      for (int i = 0; i < NB; ++i) boff[i] = (unsigned)((i * 4 + wave) * 8 + lrow) * (unsigned)s.in_ld_b + lchunk;
      aseg = s.w + bz * s.w_batch_b + (long long)co0 * s.w_ld_b;
      bseg = s.in + bz * s.in_batch_b + m0 * s.in_ld_b;
    };
    auto setup_tap = [&]() {
      const int slot = sg * 9 + ti;
      const int dy = tap_dy_l[slot], dx = tap_dx_l[slot];
      abase = aseg + tap_a_l[slot] + kc * BKB;
      bbase = bseg + tap_b_l[slot] + kc * BKB;
#pragma unroll
      for (int i = 0; i < NB; ++i)
        bok[i] = rvalid[i] && (unsigned)(ry[i] + dy) < (unsigned)p.H && (unsigned)(rx[i] + dx) < (unsigned)p.W;
    };
    auto advance = [&]() {
      if (p.abl & 4) return;            // timing experiment: every step loads the first step's tiles
      if (++ti >= ntaps_l[sg]) {
        ti = 0;
        if (++kc >= nkc) {
          kc = 0;
          ++sg;
          while (ntaps_l[sg] == 0) ++sg;
          nkc = p.seg[sg].ck_b / BKB;
          setup_seg();
        }
      }
      setup_tap();
    };
    const char* zsrc = reinterpret_cast<const char*>(pp_zero_page) + ((lane & 7) << 4);
    char* const sa_w = smem + wave * 1024;            // + i * 4096: this wavefront's pieces of the weight tile
    char* const sb_w = smem + 128 * 128 + wave * 1024;
    const char* const sa = smem;
    const char* const sb = smem + 128 * 128;
    const int frow = lane & 15, fg = lane >> 4;

    setup_seg();
    setup_tap();
    auto issue = [&](int stage) {
      const bool fake = p.abl & 1;    // timing experiment: every load reads the zero page
#pragma unroll
      for (int i = 0; i < 4; ++i) glds16(fake ? zsrc : abase + aoff[i], sa_w + stage * STAGE_B + i * 4096);
#pragma unroll
      for (int i = 0; i < NB; ++i) glds16(bok[i] && !fake ? bbase + boff[i] : zsrc, sb_w + stage * STAGE_B + i * 4096);
    };
    auto multiply = [&](int stage) {
      if (p.abl & 2) return;          // timing experiment: loads and barriers only
      const char* a = sa + stage * STAGE_B;
      const char* b = sb + stage * STAGE_B;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4 fb[MJ];
#pragma unroll
        for (int j = 0; j < MJ; ++j)
          fb[j] = *reinterpret_cast<const u32x4*>(b + tile_off<BKB>(wc * (MJ * 16) + j * 16 + frow, ks * 4 + fg));
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const u32x4 fa = *reinterpret_cast<const u32x4*>(a + tile_off<BKB>(wr * (MI * 16) + i * 16 + frow, ks * 4 + fg));
#pragma unroll
          for (int j = 0; j < MJ; ++j) Mma<bf16_t>::run(fa, fb[j], acc[i][j]);
        }
      }
    };
    if (STAGES == 1) {
      for (int st = 0; st < total; ++st) {
        issue(0);
        if (st + 1 < total) advance();   // address arithmetic of the next step under the loads
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        multiply(0);
        __syncthreads();
      }
    } else if (STAGES == 2) {
      issue(0);
      for (int st = 0; st < total; ++st) {
        const bool more = st + 1 < total;
        if (more) advance();             // addresses of step st+1 while the loads of step st are in flight
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                 // step st has landed for everyone; everyone is done reading step st-1
        if (more) issue((st + 1) & 1);   // ... whose stage these loads overwrite, under the MFMAs below
        multiply(st & 1);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                   // the epilogue reuses the LDS
    } else if (loader) {
      // barrier k (k = 0 .. total-1): the loads of step k have landed, the multipliers are done with step k-1
      issue(0);
      for (int st = 0; st < total; ++st) {
        const bool more = st + 1 < total;
        if (more) advance();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (more) issue((st + 1) & 1);   // into the stage the multipliers left before this barrier
      }
      __syncthreads();                   // the multipliers are done with the last step (their epilogue reuses the LDS)
    } else {
      for (int st = 0; st < total; ++st) {
        __syncthreads();
        multiply(st & 1);
      }
      __syncthreads();
    }
  }

  if (SPEC && !multiplier) {
    if (p.stat) __syncthreads();         // the barrier inside the statistics part of the epilogue
    return;
  }
  conv_epilogue<TO, G>(p, acc, smem, wr, wc, lane, tid, co0, m0, mt, bz);
}

// ------------------------------------------------------------------------------------------------
// Ring tile: 128 output channels x 160 pixels per tile, K steps of 64 channels through a ring of LDS stages,
// wavefronts with fixed roles, persistent workgroups.  What the measurements behind it say (tools/micro/burst2.hip):
//   * a CU pulls 64 B/clk through LDS-DMA when two bursts are in flight, 40 when it waits for each burst: one
//     burst's latency (~440 cycles) has to be hidden behind the transfer of the next;
//   * __syncthreads() is preceded by s_waitcnt vmcnt(0) (its release fence), which empties that pipeline at every
//     barrier: the loops below use the bare s_barrier and counted vmcnt waits;
//   * a wavefront's LDS reads queue behind its own LDS-DMA in flight (a ds_read_b128 issued after eight loads
//     returns after ~435 cycles instead of ~100): wavefronts that load never read LDS (tap offsets sit in lanes and
//     are fetched with v_readlane), wavefronts that multiply never load;
//   * addresses are a per-step scalar base (buffer resource) plus per-lane offsets that only change with the
//     segment; rows outside the image get an out-of-range offset (the buffer load then writes zeros and touches no
//     memory); LDS destinations are scalar.
// Wavefronts 0-3 load (9 one-KiB pieces each per step: 4 of the weight tile, 5 of the pixel tile) STAGES - 1 steps
// ahead; the others multiply.  One barrier per step: the loads of step k have landed (each loader waits for its own
// with a counted vmcnt: later steps may still be in flight) and the multipliers have left step k-1, whose stage the
// loaders refill.  Nothing else is shared: every wavefront derives the tile's tap list itself (which of the 3 x 3
// taps reach into the image from any of the tile's pixels, by wavefront-wide votes), so there is no table in LDS and
// no barrier outside the K loop, and a workgroup that walks over several tiles loads the first steps of the next tile
// while its multipliers still store the previous one: the fixed cost of a tile (about a third of a 36-step tile)
// disappears behind the K loops of its neighbours.
//   STAGES = 3, LW = 4 loading and MW = 8 multiplying wavefronts (4 x 2, 32 channels x 80 pixels each): 110 KB of
//   LDS, one workgroup per CU - the form the library uses.  Others (experiment builds): STAGES = 2, MW = 4 (2 x 2,
//   64 x 80 each), two workgroups per CU; STAGES = 4; LW = 8.
constexpr int RING_PX = 160;

__device__ __forceinline__ void bare_barrier() {
  asm volatile("" ::: "memory");        // the compiler keeps memory operations on their side of the barrier
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

typedef __attribute__((address_space(3))) void* lds_void_ptr;

// bits of the 3 x 3 taps (index ty * 3 + tx) that pixel (y, x) may read under dilation d; the centre always
__device__ __forceinline__ unsigned taps_of_pixel(int y, int x, int d, int H, int W) {
  const unsigned xb = (x - d >= 0 ? 1u : 0u) | 2u | (x + d < W ? 4u : 0u);
  return (y - d >= 0 ? xb : 0u) | (xb << 3) | (y + d < H ? xb << 6 : 0u);
}

// BST: the variant whose epilogue can also emit a batch norm's backward statistics (ConvDev::bst_rows); a separate
// instantiation, so that the register allocation of the plain kernel is not touched by it.
template <typename TO, int STAGES, int MW, int LW, int BCO = 128, bool BST = false>
// (second launch bound = wavefronts per SIMD: one workgroup per CU with three stages, two with two)
__global__ __launch_bounds__(64 * (LW + MW), (STAGES >= 3 ? 1 : 2) * (LW + MW) / 4) void conv_fwd_ring_kernel(const ConvDev p) {
  typedef TileCfg<BCO, RING_PX, MW / 2, 2> G;
  constexpr int BKB = 128, MI = G::MI, MJ = G::MJ;
  constexpr int STAGE_B = (BCO + RING_PX) * 128;      // one stage: the weight tile, then the pixel tile
  // LW loading wavefronts share the BCO / 8 + 20 one-KiB pieces of a step: piece q = i * LW + wave
  constexpr int NA = BCO / 8 / LW;                    // weight pieces per loading wavefront
  constexpr int NB = (RING_PX / 8 + LW - 1) / LW;     // pixel pieces (the last one only in the first wavefronts
                                                      // when 20 is not a multiple of LW)
  // (array bounds stay literal: with bounds that depend on a template parameter the host pass silently drops the
  // kernel's launch stub, and the library then fails to load with an undefined symbol)
  constexpr int NA_MAX = 8, NB_MAX = 5;
  static_assert(NA <= NA_MAX && NB <= NB_MAX, "loader piece tables too small");
  constexpr int AHEAD = STAGES - 1;   // steps the loaders run ahead of the multipliers
  extern __shared__ __attribute__((aligned(16))) char smem[];   // STAGES x ([A 128 rows][B 160 rows] x 128 B, swizzled)

  const int wave_all = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool loader = wave_all < LW;
  const int wave = loader ? wave_all : wave_all - LW;
  const int lane = threadIdx.x & 63;
  const long long bz = blockIdx.y;
  const int nseg = __builtin_amdgcn_readfirstlane(p.nseg);
  const unsigned ntiles = (unsigned)__builtin_amdgcn_readfirstlane(p.ring_tiles);

  // The listed taps of a tile, per segment, and its number of K steps: the same in every wavefront.  Lane l looks
  // at tile rows l, l + 64 and l + 128.
  auto tile_taps = [&](long long m0, unsigned (&seg_bits)[SEGK_MAX_SEG]) -> int {
    int y[3], x[3];
    bool v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int r = lane + 64 * k;
      const long long m = m0 + r;
      v[k] = r < RING_PX && m < p.M;
      const int rem = pixel_in_image(v[k] ? m : 0, p.HW);
      y[k] = rem / p.W;
      x[k] = rem - y[k] * p.W;
    }
    int total = 0;
#pragma unroll
    for (int sg = 0; sg < SEGK_MAX_SEG; ++sg) {
      seg_bits[sg] = 0u;
      if (sg < nseg) {
        const int R = p.seg[sg].R, d = p.seg[sg].dil + (int)blockIdx.y * p.dil_step;
        unsigned any = 0u;
#pragma unroll
        for (int k = 0; k < 3; ++k)
          if (v[k]) any |= R == 1 ? 1u : taps_of_pixel(y[k], x[k], d, p.H, p.W);
        unsigned bits = 0u;
        for (int t = 0; t < R * R; ++t)
          if (__builtin_amdgcn_ballot_w64((any >> t) & 1u) != 0ull) bits |= 1u << t;
        seg_bits[sg] = __builtin_amdgcn_readfirstlane(bits);
        total += __popc(bits) * (p.seg[sg].ck_b / BKB);
      }
    }
    total = __builtin_amdgcn_readfirstlane(total);
    return (p.abl & 16) ? 0 : total;     // timing experiment: the fixed cost of a tile
  };
  // virtual block -> tile: each XCD (virtual blocks with the same index modulo 8; the grid is a multiple of 8) walks
  // over a contiguous run of tiles, so that the output-channel tiles of one pixel tile and neighbouring pixel tiles
  // (which share the rows the 3 x 3 taps reach into) meet in the same L2
  auto tile_of = [&](unsigned v) -> unsigned {
    const unsigned q = ntiles >> 3, r = ntiles & 7u, xcd = v & 7u;
    return xcd * q + (xcd < r ? xcd : r) + (v >> 3);
  };
  // (v >> 3 exceeds an XCD's run only for virtual blocks past the end, which the loops below never reach: the run of
  // XCD x holds q + (x < r) tiles and v < ntiles rounded up to the grid stride is filtered by in_range)
  auto in_range = [&](unsigned v) -> bool {
    const unsigned q = ntiles >> 3, r = ntiles & 7u, xcd = v & 7u;
    return (v >> 3) < q + (xcd < r ? 1u : 0u);
  };

  if (loader) {
    // the loading wavefronts issue ahead of the multiplying ones on their SIMD: a load delayed behind a run of MFMAs
    // is a late stage for everybody (kernel durations 1-4 % shorter on 19 of 21 shapes; bit 64: default priority)
    if (!(p.abl & 64)) __builtin_amdgcn_s_setprio(3);
    const int lrow = lane >> 3;
    const int lchunk = ((lane & 7) ^ lrow) << 4;   // piece rows are 8-aligned: row & 7 == lrow
    int fill = 0;                                  // stage of the next step to issue
    for (unsigned v = blockIdx.x; in_range(v); v += gridDim.x) {
      const unsigned tile = tile_of(v);
      const int co0 = (int)(tile % (unsigned)p.n_tiles) * G::BCO;
      const long long m0 = p.m_begin + (long long)(tile / (unsigned)p.n_tiles) * G::BPX;
      unsigned seg_bits[SEGK_MAX_SEG];
      const int total = tile_taps(m0, seg_bits);
      // piece q = i*4 + wave covers tile rows q*8 .. q*8+7; lane l -> row q*8 + (l>>3), LDS slot l&7, which holds
      // source chunk (l&7) ^ (row&7)
      int wrow[NA_MAX], ry[NB_MAX], rx[NB_MAX];
      bool rvalid[NB_MAX];
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int co = co0 + (i * LW + wave) * 8 + lrow;
        wrow[i] = (co < p.Cout ? co : p.Cout - 1) - co0;
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const long long m = m0 + (i * LW + wave) * 8 + lrow;
        rvalid[i] = m < p.M && (i * LW + wave) * 8 < RING_PX;
        const int rem = pixel_in_image(rvalid[i] ? m : 0, p.HW);
        ry[i] = rem / p.W;
        rx[i] = rem - ry[i] * p.W;
      }
      int aoff[NA_MAX], boff[NB_MAX];
      const bool last_b = (NB - 1) * LW + wave < RING_PX / 8;   // this wavefront has a pixel piece in the last round
      // s_waitcnt vmcnt(steps * loads per step of this wavefront): the immediate has to be a literal
      auto wait_behind = [&](int steps) {
        const int n = steps <= 0 ? 0 : steps * (NA + NB - (last_b ? 0 : 1));
        if (n == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (n == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if (n == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (n == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else if (n == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (n == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (n == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
        else if (n == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (not reached: LW is 4 or 8, at most two steps behind)
      };
#ifdef SEGK_EXPERIMENTS
      // timing experiment (bit 256): cycles this wavefront spends waiting for its loads, at the barrier and issuing
      // the loads, and in the whole tile (the rest is bookkeeping); the first workgroup's loading wavefronts leave
      // them in the statistics buffer (whose contents are then no statistics)
      long long c_wait = 0, c_bar = 0, c_iss = 0;
      const bool stamp = (p.abl & 256) != 0;
      const long long c_begin = stamp ? __builtin_readcyclecounter() : 0;
#endif
      // The hand-over protocol of a tile: its first AHEAD steps are issued at once (their stages are free: the
      // multipliers left them before the last barrier of the tile before); from then on the oldest step in flight
      // is handed over before the next is issued - this wavefront's loads of it have landed (those of the AHEAD - 1
      // steps behind it may be in flight), and behind the barrier the multipliers have left the step before it,
      // whose stage is filled next.
      int issued = 0, handed = 0;
      auto hand_over = [&](int behind) {
#ifdef SEGK_EXPERIMENTS
        if (stamp) {
          const long long t0 = __builtin_readcyclecounter();
          wait_behind(behind);
          const long long t1 = __builtin_readcyclecounter();
          bare_barrier();
          c_wait += t1 - t0;
          c_bar += __builtin_readcyclecounter() - t1;
          ++handed;
          return;
        }
#endif
        wait_behind(behind);
        bare_barrier();
        ++handed;
      };
      int sg = -1;
      for (int done = 0; done < total;) {
        // the next segment with listed taps (at most SEGK_MAX_SEG per tile)
        unsigned bits = 0u;
        do {
          ++sg;
          // (readfirstlane: the compiler's uniformity analysis loses these values, and a resource descriptor it takes
          // for divergent costs a readfirstlane loop around every load)
          bits = __builtin_amdgcn_readfirstlane(sg == 0 ? seg_bits[0] : sg == 1 ? seg_bits[1] : sg == 2 ? seg_bits[2] : seg_bits[3]);
        } while (bits == 0u);
        sg = __builtin_amdgcn_readfirstlane(sg);
        const int ntap = __popc(bits);
        const SegDev& s = p.seg[sg];
        const int R = s.R, d = s.dil + (int)blockIdx.y * p.dil_step;
        const int nkc = s.ck_b / BKB;
#pragma unroll
        for (int i = 0; i < NA; ++i) aoff[i] = wrow[i] * s.w_ld_b + lchunk;
#pragma unroll
        for (int i = 0; i < NB; ++i) boff[i] = ((i * LW + wave) * 8 + lrow) * (int)s.in_ld_b + lchunk;
#ifdef SEGK_EXPERIMENTS
        if (p.abl & 1) {                   // timing experiment: every load out of range (zeros, no memory access)
#pragma unroll
          for (int i = 0; i < NA; ++i) aoff[i] = (int)0x80000000;
        }
#endif
        const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(s.w + bz * s.w_batch_b + (long long)co0 * s.w_ld_b), 0, 0x7fffffff, 0x00020000);
        const char* const bseg = s.in + bz * s.in_batch_b + m0 * s.in_ld_b;
        // lane l takes the l-th listed tap: its index and its two operand offsets
        int tap = 0, seen = 0;
        for (int t = 0; t < R * R; ++t) {
          const bool listed = (bits >> t) & 1u;
          if (listed && seen == lane) tap = t;
          seen += listed ? 1 : 0;
        }
        const int dy = R == 1 ? 0 : (tap / 3 - 1) * d, dx = R == 1 ? 0 : (tap % 3 - 1) * d;
        const int v_tap = tap;
        const int v_ta = (int)((long long)tap * s.w_tap_b);
        const int v_tb = (int)(((long long)dy * p.W + dx) * s.in_ld_b);
        // Everything a step needs is laid out here, once per segment, under a compile-time tap slot t: the offsets of
        // this lane's pixel pieces (rows the tap may not read get an offset beyond the resource's range: the load
        // writes zeros and touches no memory).  A step is then two v_readlane with a literal lane, two additions and
        // the loads; looking the tap up per step (a dependent chain of v_readlane, 64-bit shifts and selects) cost
        // 450 of the 1040 cycles a loading wavefront needed per step.
        unsigned m9[NB_MAX];
#pragma unroll
        for (int i = 0; i < NB; ++i) m9[i] = !rvalid[i] ? 0u : R == 1 ? 1u : taps_of_pixel(ry[i], rx[i], d, p.H, p.W);
        int voff[9][NB_MAX];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int tp = __builtin_amdgcn_readlane(v_tap, t);
#pragma unroll
          for (int i = 0; i < NB; ++i) voff[t][i] = ((m9[i] >> tp) & 1u) ? boff[i] : (int)0x80000000;
#ifdef SEGK_EXPERIMENTS
          if (p.abl & 1) {
#pragma unroll
            for (int i = 0; i < NB; ++i) voff[t][i] = (int)0x80000000;
          }
#endif
        }
        for (int kc = 0; kc < nkc; ++kc) {
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            if (t < ntap) {
              if (issued >= AHEAD) hand_over(AHEAD - 1);
#ifdef SEGK_EXPERIMENTS
              const long long i0 = stamp ? __builtin_readcyclecounter() : 0;
#endif
              const int a_soff = __builtin_amdgcn_readlane(v_ta, t) + kc * BKB;
              const char* const b_base = bseg + __builtin_amdgcn_readlane(v_tb, t) + kc * BKB;
              char* const sa_w = smem + fill * STAGE_B + wave * 1024;
              char* const sb_w = sa_w + BCO * 128;
#pragma unroll
              for (int i = 0; i < NA; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (lds_void_ptr)(sa_w + i * (LW * 1024)), 16, aoff[i], a_soff, 0, 0);
              const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc((void*)b_base, 0, 0x7fffffff, 0x00020000);
#pragma unroll
              for (int i = 0; i < NB; ++i)
                if (i + 1 < NB || last_b)
                  __builtin_amdgcn_raw_ptr_buffer_load_lds(brsrc, (lds_void_ptr)(sb_w + i * (LW * 1024)), 16, voff[t][i], 0, 0, 0);
              fill = fill == STAGES - 1 ? 0 : fill + 1;
              ++issued;
#ifdef SEGK_EXPERIMENTS
              if (stamp) c_iss += __builtin_readcyclecounter() - i0;
#endif
            }
          }
        }
        done += ntap * nkc;
      }
      while (handed < issued) hand_over(issued - 1 - handed < AHEAD - 1 ? issued - 1 - handed : AHEAD - 1);
#ifdef SEGK_EXPERIMENTS
      if (stamp && p.stat && blockIdx.x == 0 && lane == 0 && v == blockIdx.x) {
        double* o = p.stat + wave * 8;
        const long long all = __builtin_readcyclecounter() - c_begin;
        o[0] = (double)c_wait; o[1] = (double)c_bar; o[2] = (double)(all - c_wait - c_bar - c_iss); o[3] = (double)c_iss;
        o[4] = (double)total;
      }
#endif
    }
    return;
  }

  const int wr = wave >> 1, wc = wave & 1;
  const int frow = lane & 15, fg = lane >> 4;
  int stage = 0;
  if (p.abl & 32) __builtin_amdgcn_s_setprio(3);     // timing experiment: multiplying wavefronts issue first
  for (unsigned v = blockIdx.x; in_range(v); v += gridDim.x) {
    const unsigned tile = tile_of(v);
    const int co0 = (int)(tile % (unsigned)p.n_tiles) * G::BCO;
    const long long mt = tile / (unsigned)p.n_tiles;
    const long long m0 = p.m_begin + mt * G::BPX;
    unsigned seg_bits[SEGK_MAX_SEG];
    const int total = tile_taps(m0, seg_bits);
    f32x4 acc[MI][MJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < MJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (MI == 4) {
      // 256 output channels per tile: a multiplying wavefront owns 64 channels x 80 pixels, 40 MFMAs per step in four
      // units of ten (one half of the step's 64 contracted channels x one pair of 16-channel blocks).  The fragments
      // of a unit arrive while the unit before it multiplies: the pixel fragments of a half step in a buffer of
      // their own (two buffers), the weight fragments of a pair in one of two small ones - 56 registers of fragments
      // beside 80 accumulators, which keeps three wavefronts per SIMD (one loading, two multiplying) resident.
      u32x4 fb[2][MJ];   // [half of the step]
      u32x4 fa[2][2];    // [unit parity][block of the pair]
      auto read_b = [&](int ks) {
        const char* b = smem + stage * STAGE_B + BCO * 128;
#pragma unroll
        for (int j = 0; j < MJ; ++j)
          fb[ks][j] = *reinterpret_cast<const u32x4*>(b + tile_off<BKB>(wc * (MJ * 16) + j * 16 + frow, ks * 4 + fg));
      };
      auto read_a = [&](int buf, int pair, int ks) {
        const char* a = smem + stage * STAGE_B;
#pragma unroll
        for (int i = 0; i < 2; ++i)
          fa[buf][i] = *reinterpret_cast<const u32x4*>(a + tile_off<BKB>(wr * 64 + (pair * 2 + i) * 16 + frow, ks * 4 + fg));
        __builtin_amdgcn_sched_barrier(0);   // the reads are issued here, not after the MFMAs that follow
      };
      auto half_unit = [&](int buf, int pair, int ks, int i) {
#pragma unroll
        for (int j = 0; j < MJ; ++j) Mma<bf16_t>::run(fa[buf][i], fb[ks][j], acc[pair * 2 + i][j]);
        __builtin_amdgcn_sched_barrier(0);
      };
      // The two multiplying wavefronts of a SIMD (wave and wave + 4) leave every barrier together; were their
      // streams identical, both would issue fragment reads at the same moments and both MFMA bursts would collide.
      // The second group therefore issues the reads of the next unit in the middle of its MFMAs instead of in
      // front of them: one group's reads fall into the other's MFMAs.
      auto k_loop = [&](auto late_tag) {
        constexpr bool LATE = decltype(late_tag)::value;
        // one unit: wait for its fragments, issue the reads `next` of the unit after it, ten MFMAs
        auto unit = [&](int buf, int pair, int ks, auto next) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (!LATE) next();
          half_unit(buf, pair, ks, 0);
          if (LATE) next();
          half_unit(buf, pair, ks, 1);
        };
        bare_barrier();
        read_b(0);
        read_a(0, 0, 0);
        for (int st = 0; st + 1 < total; ++st) {
          unit(0, 0, 0, [&] { read_a(1, 1, 0); });
          unit(1, 1, 0, [&] { read_b(1); read_a(0, 0, 1); });
          unit(0, 0, 1, [&] { read_a(1, 1, 1); });
          // (the wait in front of this unit: every read of step st is complete, its stage is free)
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          stage = stage == STAGES - 1 ? 0 : stage + 1;
          bare_barrier();
          unit(1, 1, 1, [&] { read_b(0); read_a(0, 0, 0); });
        }
        unit(0, 0, 0, [&] { read_a(1, 1, 0); });
        unit(1, 1, 0, [&] { read_b(1); read_a(0, 0, 1); });
        unit(0, 0, 1, [&] { read_a(1, 1, 1); });
        unit(1, 1, 1, [&] {});
        stage = stage == STAGES - 1 ? 0 : stage + 1;
      };
      if (total > 0) {
        if (wave < MW / 2) k_loop(std::false_type{});
        else k_loop(std::true_type{});
      }
    } else if (MW == 8 || STAGES == 3) {
      // Fragment reads run one half step (32 channels) ahead of the MFMAs: while the MFMAs of one half run, the next
      // half's fragments arrive in the other register buffer.  Once the second half of step k is on its way the
      // barrier of step k + 1 follows (its loads have landed; the multipliers' reads of step k are complete), so no
      // fragment read waits in front of an MFMA - the two multiplying wavefronts of a SIMD leave a barrier together
      // and would otherwise both sit out the LDS latency before either has an MFMA to issue.
      u32x4 fa[2][MI], fb[2][MJ];          // [buffer][tile]
      auto read_half = [&](int buf, int ks) {
#ifdef SEGK_EXPERIMENTS
        if (p.abl & 2) return;             // timing experiment: loads and barriers only
#endif
        const char* a = smem + stage * STAGE_B;
        const char* b = a + BCO * 128;
#pragma unroll
        for (int j = 0; j < MJ; ++j)
          fb[buf][j] = *reinterpret_cast<const u32x4*>(b + tile_off<BKB>(wc * (MJ * 16) + j * 16 + frow, ks * 4 + fg));
#pragma unroll
        for (int i = 0; i < MI; ++i)
          fa[buf][i] = *reinterpret_cast<const u32x4*>(a + tile_off<BKB>(wr * (MI * 16) + i * 16 + frow, ks * 4 + fg));
        __builtin_amdgcn_sched_barrier(0);   // the reads are issued here, not after the MFMAs that follow
      };
      auto mfma_block = [&](int buf, int i) {
#ifdef SEGK_EXPERIMENTS
        if (p.abl & (2 | 128)) return;     // timing experiments: ... and fragment reads without the MFMAs (128)
#endif
#pragma unroll
        for (int j = 0; j < MJ; ++j) Mma<bf16_t>::run(fa[buf][i], fb[buf][j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
      };
      // (the second group of multiplying wavefronts - the SIMD partners of the first - issues the next half's reads
      // after the first of its MFMA blocks instead of in front of them: see the 256-channel form above)
      auto k_loop = [&](auto late_tag) {
        constexpr bool LATE = decltype(late_tag)::value;
        auto half = [&](int buf, auto next) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (!LATE) next();
          mfma_block(buf, 0);
          if (LATE) next();
#pragma unroll
          for (int i = 1; i < MI; ++i) mfma_block(buf, i);
        };
        bare_barrier();
        read_half(0, 0);
        // (the last step is peeled off: were its missing barrier and reads a branch inside the loop, the compiler's
        // counted waits in front of the MFMAs would have to assume the path without reads and wait for all of them)
        for (int st = 0; st + 1 < total; ++st) {
          half(0, [&] { read_half(1, 1); });
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every read of step st is complete: its stage is free
          stage = stage == STAGES - 1 ? 0 : stage + 1;
          bare_barrier();
          half(1, [&] { read_half(0, 0); });
        }
        half(0, [&] { read_half(1, 1); });
        half(1, [&] {});
        stage = stage == STAGES - 1 ? 0 : stage + 1;
      };
      if (total > 0) {
        if (wave < MW / 2) k_loop(std::false_type{});
        else k_loop(std::true_type{});
      }
    } else {
      for (int st = 0; st < total; ++st) {
        bare_barrier();
        if (!(p.abl & 2)) {                // timing experiment: loads and barriers only
          const char* a = smem + stage * STAGE_B;
          const char* b = a + BCO * 128;
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            u32x4 fb[MJ];
#pragma unroll
            for (int j = 0; j < MJ; ++j)
              fb[j] = *reinterpret_cast<const u32x4*>(b + tile_off<BKB>(wc * (MJ * 16) + j * 16 + frow, ks * 4 + fg));
#pragma unroll
            for (int i = 0; i < MI; ++i) {
              const u32x4 fa = *reinterpret_cast<const u32x4*>(a + tile_off<BKB>(wr * (MI * 16) + i * 16 + frow, ks * 4 + fg));
#pragma unroll
              for (int j = 0; j < MJ; ++j) Mma<bf16_t>::run(fa, fb[j], acc[i][j]);
            }
          }
        }
        stage = stage == STAGES - 1 ? 0 : stage + 1;
      }
    }

    // Epilogue without LDS or barriers (the loaders are already filling the ring for the next tile): a wavefront's
    // statistics go to a partial row of its own (two rows per pixel tile, one per 80-pixel half).
    // acc[i][j][r] = out[pixel = wc*MJ*16 + j*16 + (lane&15)][co = wr*MI*16 + i*16 + (lane>>4)*4 + r]
#ifdef SEGK_EXPERIMENTS
    const bool want_stat = p.stat && !(p.abl & 256);   // (bit 256: the buffer holds the loaders' cycle counts)
#else
    const bool want_stat = p.stat != nullptr;
#endif
    if (want_stat) {
      // per lane: sums over its MJ pixels (fp64: the squares of fp32 accumulators are exact), then over the 16
      // pixels of the DPP row; two 16-channel blocks at a time (index (i & 1) * 4 + r for the sums, 8 more for the
      // squares) so that only 16 doubles are live beside the accumulators
      double* row = p.stat + bz * p.stat_batch + (size_t)((p.stat_row0 + mt) * 2 + wc) * 2 * p.Cout;
      const int base = row16_scatter_base<16>(lane);
      const int co_l = co0 + wr * (MI * 16) + ((base & 7) >> 2) * 16 + fg * 4 + (base & 3);
#pragma unroll
      for (int i0 = 0; i0 < MI; i0 += 2) {
        double v[16];
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            double sum = 0.0, sq = 0.0;
#pragma unroll
            for (int j = 0; j < MJ; ++j) {
              const double t = (double)acc[i0 + ii][j][r];
              sum += t;
              sq = fma(t, t, sq);
            }
            v[ii * 4 + r] = sum;
            v[8 + ii * 4 + r] = sq;
          }
        row16_reduce_scatter<16>(v, lane);
        const int co = co_l + i0 * 16;
        if (co < p.Cout) row[(size_t)(base >> 3) * p.Cout + co] = v[0];
      }
    }
    if (p.abl & 8) continue;               // timing experiment: no output stores
    const int cpad = (p.Cout + 3) & ~3;
    char* const obase = p.out + bz * p.out_batch_b;
    // every channel of the tile exists and nothing is added to the accumulators (the same in every lane): no
    // per-element tests; bf16 outputs leave as 16 bytes per lane - a lane swaps its four channels of an odd / even
    // 16-channel block with the lane 16 further on, which holds the four channels beside them
    const bool plain = __builtin_amdgcn_readfirstlane((co0 + BCO <= p.Cout && !p.bias && !p.img_bias) ? 1 : 0) != 0;
    if (plain) {
      if constexpr (sizeof(TO) == 2) {
        const bool wide16 = __builtin_amdgcn_readfirstlane(((((uintptr_t)obase) & 15) == 0 && (p.out_ld & 7) == 0 &&
                                                             (!p.addend || ((((uintptr_t)p.addend) | (uintptr_t)p.addend_ld_b) & 15) == 0)) ? 1 : 0) != 0;
        if constexpr (BST) {
          static_assert(sizeof(TO) == 2, "the statistics epilogue stores bf16");
          if (wide16 && __builtin_amdgcn_readfirstlane(p.bst_rows != nullptr ? 1 : 0)) {
          const int cbase = co0 + wr * (MI * 16) + ((fg & 1) ? 16 + (fg - 1) * 4 : fg * 4);
          // Per pair of 16-channel blocks a lane finishes eight consecutive channels (cbase + i * 16 .. + 7) of its MJ
          // pixels.  The producing layer's input and ReLU pattern are requested for all MJ pixels before the first
          // result is stored: behind a store the compiler may not move a load up (the buffers might overlap), and MJ
          // loads taken one at a time are MJ memory latencies in a row.  (No addend in this variant.)
          const bool bits_given = __builtin_amdgcn_readfirstlane(p.bst_bits != nullptr ? 1 : 0) != 0;
          const bool recompute = __builtin_amdgcn_readfirstlane((p.bst_relu && !p.bst_bits) ? 1 : 0) != 0;
          const long long mrow = m0 + wc * (MJ * 16) + frow;   // pixel of j = 0
          auto finish_store = [&](int i, int j, long long m) -> u32x4 {
            unsigned a0 = pack_bf16x2(acc[i][j][0], acc[i][j][1]), a1 = pack_bf16x2(acc[i][j][2], acc[i][j][3]);
            unsigned b0 = pack_bf16x2(acc[i + 1][j][0], acc[i + 1][j][1]), b1 = pack_bf16x2(acc[i + 1][j][2], acc[i + 1][j][3]);
            const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
            const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
            if (m < p.M) *reinterpret_cast<u32x4*>(obase + ((size_t)m * p.out_ld + cbase) * 2 + i * 32) = o;
            return o;
          };
#pragma unroll
          for (int i = 0; i < MI; i += 2) {
            const int c8 = cbase + i * 16;
            // Statistics of the producing layer's batch-norm backward from the values as they are stored: per lane the
            // sums over its MJ pixels of g and g * xhat for its eight channels (fp32, as the stand-alone pass
            // accumulates them), then over the 16 pixels of the DPP row in fp64; one partial row per wavefront column,
            // as for the forward statistics.
            u32x4 xr[MJ];
            unsigned kb[MJ];
#pragma unroll
            for (int j = 0; j < MJ; ++j) {
              const long long m = mrow + j * 16;
              xr[j] = u32x4{0u, 0u, 0u, 0u};
              kb[j] = 0xffu;
              if (m < p.M) {
                xr[j] = *reinterpret_cast<const u32x4*>(p.bst_x + (size_t)m * p.bst_x_ld_b + (size_t)c8 * 2);
                if (bits_given) kb[j] = p.bst_bits[(size_t)m * p.bst_bits_ld + (c8 >> 3)];
              }
            }
            float mu[8], is[8], sa[8], sb[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const f32x4 t0 = *reinterpret_cast<const f32x4*>(p.bst_mean + c8 + h * 4);
              const f32x4 t1 = *reinterpret_cast<const f32x4*>(p.bst_inv_std + c8 + h * 4);
#pragma unroll
              for (int k = 0; k < 4; ++k) { mu[h * 4 + k] = t0[k]; is[h * 4 + k] = t1[k]; }
            }
            if (recompute) {   // the ReLU pattern from x * scale + shift, into kb
              float sc[8], sh[8];
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                const f32x4 t2 = *reinterpret_cast<const f32x4*>(p.bst_scale + c8 + h * 4);
                const f32x4 t3 = *reinterpret_cast<const f32x4*>(p.bst_shift + c8 + h * 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) { sc[h * 4 + k] = t2[k]; sh[h * 4 + k] = t3[k]; }
              }
#pragma unroll
              for (int j = 0; j < MJ; ++j) {
                unsigned pat = 0u;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                  const unsigned xw = xr[j][k >> 1];
                  const float xv = __uint_as_float((k & 1) ? (xw & 0xffff0000u) : (xw << 16));
                  pat |= (fmaf(xv, sc[k], sh[k]) > 0.f ? 1u : 0u) << k;
                }
                kb[j] = pat;
              }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) sa[k] = sb[k] = 0.f;
#pragma unroll
            for (int j = 0; j < MJ; ++j) {
              const long long m = mrow + j * 16;
              const u32x4 o = finish_store(i, j, m);
              if (m < p.M) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                  const unsigned ow = o[k >> 1], xw = xr[j][k >> 1];
                  float g = __uint_as_float((k & 1) ? (ow & 0xffff0000u) : (ow << 16));
                  const float xv = __uint_as_float((k & 1) ? (xw & 0xffff0000u) : (xw << 16));
                  g = ((kb[j] >> k) & 1u) ? g : 0.f;
                  sa[k] += g;
                  sb[k] = fmaf(g, (xv - mu[k]) * is[k], sb[k]);
                }
              }
            }
            double v[16];
#pragma unroll
            for (int k = 0; k < 8; ++k) { v[k] = (double)sa[k]; v[8 + k] = (double)sb[k]; }
            row16_reduce_scatter<16>(v, lane);
            const int base = row16_scatter_base<16>(lane);
            double* brow = p.bst_rows + (size_t)((p.stat_row0 + mt) * 2 + wc) * 2 * p.Cout;
            brow[(size_t)(base >> 3) * p.Cout + c8 + (base & 7)] = v[0];
          }
          continue;
          }
        }
        if (wide16) {
          const int cbase = co0 + wr * (MI * 16) + ((fg & 1) ? 16 + (fg - 1) * 4 : fg * 4);
#pragma unroll
          for (int j = 0; j < MJ; ++j) {
            const long long m = m0 + wc * (MJ * 16) + j * 16 + frow;
            char* orow = obase + ((size_t)m * p.out_ld + cbase) * 2;
#pragma unroll
            for (int i = 0; i < MI; i += 2) {
              unsigned a0 = pack_bf16x2(acc[i][j][0], acc[i][j][1]), a1 = pack_bf16x2(acc[i][j][2], acc[i][j][3]);
              unsigned b0 = pack_bf16x2(acc[i + 1][j][0], acc[i + 1][j][1]), b1 = pack_bf16x2(acc[i + 1][j][2], acc[i + 1][j][3]);
              const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
              const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
              u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
              if (m < p.M) {
                if (p.addend) {
                  u32x4 ad = *reinterpret_cast<const u32x4*>(p.addend + (size_t)m * p.addend_ld_b + (size_t)cbase * 2 + i * 32);
                  if (p.addend_bits) {   // this lane's eight channels are one byte of the pattern
                    const unsigned b = p.addend_bits[(size_t)m * p.addend_bits_ld + ((cbase + i * 16) >> 3)];
#pragma unroll
                    for (int e = 0; e < 4; ++e) ad[e] = keep_bf16x2(ad[e], (b >> (2 * e)) & 1u, (b >> (2 * e + 1)) & 1u);
                  }
#pragma unroll
                  for (int e = 0; e < 4; ++e) o[e] = add_bf16x2(o[e], ad[e]);
                }
                *reinterpret_cast<u32x4*>(orow + i * 32) = o;
              }
            }
          }
          continue;
        }
      }
#pragma unroll
      for (int j = 0; j < MJ; ++j) {
        const long long m = m0 + wc * (MJ * 16) + j * 16 + frow;
        if (m >= p.M) continue;
        const size_t coff = (size_t)(co0 + wr * (MI * 16) + fg * 4) * sizeof(TO);
        char* orow = obase + (size_t)m * p.out_ld * sizeof(TO) + coff;
        const char* arow = p.addend ? p.addend + (size_t)m * p.addend_ld_b + coff : nullptr;
        const uint8_t* brow = p.addend_bits ? p.addend_bits + (size_t)m * p.addend_bits_ld : nullptr;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          float o[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
          if (arow) {
            // the four channels of this lane: one nibble of the pattern
            const int c4 = co0 + wr * (MI * 16) + i * 16 + fg * 4;
            const unsigned nb = brow ? (brow[c4 >> 3] >> (c4 & 4)) & 15u : 15u;
            if constexpr (sizeof(TO) == 2) {
              u32x2 ad = *reinterpret_cast<const u32x2*>(arow + i * 32);
              ad[0] = keep_bf16x2(ad[0], nb & 1u, nb & 2u);
              ad[1] = keep_bf16x2(ad[1], nb & 4u, nb & 8u);
              const unsigned r0 = add_bf16x2(pack_bf16x2(o[0], o[1]), ad[0]), r1 = add_bf16x2(pack_bf16x2(o[2], o[3]), ad[1]);
              const u32x2 t = {r0, r1};
              *reinterpret_cast<u32x2*>(orow + i * 32) = t;
              continue;
            } else {
              const f32x4 ad = *reinterpret_cast<const f32x4*>(arow + i * 64);
#pragma unroll
              for (int r = 0; r < 4; ++r) o[r] += ((nb >> r) & 1u) ? ad[r] : 0.f;
            }
          }
          store_out4<TO>(orow + i * 16 * sizeof(TO), o);
        }
      }
      continue;
    }
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
      const long long m = m0 + wc * (MJ * 16) + j * 16 + frow;
      if (m >= p.M) continue;
      const float* ib = p.img_bias ? p.img_bias + (m / p.HW) * p.Cout : nullptr;
      char* orow = obase + (size_t)m * p.out_ld * sizeof(TO);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int c = co0 + wr * (MI * 16) + i * 16 + fg * 4;
        if (c >= cpad) continue;
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float t = acc[i][j][r];
          if (c + r < p.Cout) {
            if (p.bias) t += p.bias[c + r];
            if (ib) t += ib[c + r];
          } else {
            t = 0.f;
          }
          o[r] = t;
        }
        store_out4<TO>(orow + (size_t)c * sizeof(TO), o);
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------
// A 1 x 1 convolution over a handful of rows (fp32): the ASPP image-pooling branch, whose "pixels" are the N pooled
// vectors of the batch (modeling/aspp.py:121-123) - 16 rows x 2048 channels onto 256.  The implicit-GEMM tiles give
// such a problem two workgroups that walk 64 K steps one after the other (160 us for 17 MFLOP).  Here a workgroup owns
// four output channels for all rows, its 256 threads split K (float4 per thread and trip), and the partial sums meet
// in a fixed order: lanes by a butterfly, wavefronts in index order.  Statistics (one partial row) and bias as in
// the tiles' epilogue.  M <= SMALL_ROWS_MAX.
// ------------------------------------------------------------------------------------------------
constexpr int SMALL_ROWS_MAX = 32, SMALL_ROWS_CO = 4;

template <typename TO>
__global__ __launch_bounds__(256) void conv_small_rows_kernel(const ConvDev p) {
  constexpr int RG = 16;                                 // rows per pass
  __shared__ float part[4][RG * SMALL_ROWS_CO];
  __shared__ float vals[RG * SMALL_ROWS_CO];
  const SegDev& sg = p.seg[0];
  const int co0 = blockIdx.x * SMALL_ROWS_CO;
  const int K = sg.ck_b >> 2, M = (int)p.M;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* wrow[SMALL_ROWS_CO];
#pragma unroll
  for (int c = 0; c < SMALL_ROWS_CO; ++c)
    wrow[c] = reinterpret_cast<const float*>(sg.w + (size_t)(co0 + c < p.Cout ? co0 + c : p.Cout - 1) * sg.w_ld_b);
  double s1 = 0.0, s2 = 0.0;                              // threads 0..3: statistics of channel co0 + threadIdx.x
  for (int m0 = 0; m0 < M; m0 += RG) {
    float acc[RG][SMALL_ROWS_CO];
#pragma unroll
    for (int r = 0; r < RG; ++r)
#pragma unroll
      for (int c = 0; c < SMALL_ROWS_CO; ++c) acc[r][c] = 0.f;
    for (int k = threadIdx.x * 4; k < K; k += 1024) {
      f32x4 wv[SMALL_ROWS_CO];
#pragma unroll
      for (int c = 0; c < SMALL_ROWS_CO; ++c) wv[c] = *reinterpret_cast<const f32x4*>(wrow[c] + k);
#pragma unroll
      for (int r = 0; r < RG; ++r) {
        const int m = m0 + r < M ? m0 + r : M - 1;
        const f32x4 xv = *reinterpret_cast<const f32x4*>(sg.in + (size_t)m * sg.in_ld_b + (size_t)k * 4);
#pragma unroll
        for (int c = 0; c < SMALL_ROWS_CO; ++c)
          acc[r][c] = fmaf(xv[3], wv[c][3], fmaf(xv[2], wv[c][2], fmaf(xv[1], wv[c][1], fmaf(xv[0], wv[c][0], acc[r][c]))));
      }
    }
#pragma unroll
    for (int r = 0; r < RG; ++r)
#pragma unroll
      for (int c = 0; c < SMALL_ROWS_CO; ++c) {
        float v = acc[r][c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) part[wave][r * SMALL_ROWS_CO + c] = v;
      }
    __syncthreads();
    if (threadIdx.x < RG * SMALL_ROWS_CO) {
      const int r = threadIdx.x / SMALL_ROWS_CO, c = threadIdx.x - r * SMALL_ROWS_CO;
      const int m = m0 + r, co = co0 + c;
      float v = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
      vals[threadIdx.x] = m < M ? v : 0.f;               // the statistics see the convolution alone, as in the tiles
      if (m < M && co < p.Cout) {
        if (p.bias) v += p.bias[co];
        if (p.img_bias) v += p.img_bias[(size_t)(m / p.HW) * p.Cout + co];
        Elem<TO>::store(reinterpret_cast<TO*>(p.out) + (size_t)m * p.out_ld + co, v);
      }
    }
    __syncthreads();
    if (threadIdx.x < SMALL_ROWS_CO) {
#pragma unroll
      for (int r = 0; r < RG; ++r) {
        const double t = (double)vals[r * SMALL_ROWS_CO + threadIdx.x];
        s1 += t;
        s2 = fma(t, t, s2);
      }
    }
    __syncthreads();
  }
  if (p.stat && threadIdx.x < SMALL_ROWS_CO && co0 + (int)threadIdx.x < p.Cout) {
    p.stat[co0 + threadIdx.x] = s1;
    p.stat[p.Cout + co0 + threadIdx.x] = s2;
  }
  // output columns between Cout and Cout rounded up to 4 are zero, as the tiles leave them
  if (blockIdx.x == 0) {
    const int cpad = (p.Cout + 3) & ~3;
    for (int i = threadIdx.x; i < M * (cpad - p.Cout); i += 256) {
      const int m = i / (cpad - p.Cout), c = p.Cout + i % (cpad - p.Cout);
      Elem<TO>::store(reinterpret_cast<TO*>(p.out) + (size_t)m * p.out_ld + c, 0.f);
    }
  }
}

// the problems conv_small_rows_kernel takes
static bool small_rows_problem(const segk_conv_args* a) {
  return a->dtype == SEGK_F32 && a->nseg == 1 && a->seg[0].R == 1 && (long long)a->N * a->H * a->W <= SMALL_ROWS_MAX &&
         a->batch <= 1 && !a->addend && a->seg[0].Ck % 4 == 0;
}

static int cu_count() {
  static int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
      v = 256;
    return v;
  }();
  return n;
}

// The grid of a ring launch: every tile its own workgroup while they all fit on the chip at once, otherwise as many
// workgroups as fit (a multiple of 8: one run of tiles per XCD), each walking over several tiles.
template <typename TO, int STAGES, int MW, int LW = 4, int BCO = 128, bool BST = false>
static int launch_ring(ConvDev& d, hipStream_t st, int batch, long long px_tiles) {
  constexpr int lds = STAGES * (BCO + RING_PX) * 128;
  d.n_tiles = ceil_div(d.Cout, BCO);
  const long long tiles = px_tiles * d.n_tiles;
  if (tiles >= 2147483647LL) {
    set_error("conv: too many tiles");
    return SEGK_INVALID_ARG;
  }
  d.ring_tiles = (int)tiles;
  const long long slots = (long long)cu_count() * (STAGES >= 3 ? 1 : 2);
  long long grid = (tiles + 7) / 8 * 8;
  if (grid * batch > slots) {
    grid = slots / batch / 8 * 8;
    if (grid < 8) grid = 8;
  }
  static LdsConfigured configured;  // per instantiation
  int rc = set_lds(conv_fwd_ring_kernel<TO, STAGES, MW, LW, BCO, BST>, lds, "conv_fwd_ring", configured);
  if (rc != SEGK_OK) return rc;
  conv_fwd_ring_kernel<TO, STAGES, MW, LW, BCO, BST><<<dim3((unsigned)grid, (unsigned)batch), 64 * (LW + MW), lds, st>>>(d);
  return SEGK_OK;
}

// ------------------------------------------------------------------------------------------------
// Weight gradient on the ring (bf16): one workgroup owns 256(co) x 128(ci) of one tap over one part of the pixel
// range and walks over it in chunks of 64 pixels.  A stage holds three LDS images of [64 px][128 channels]
// (dout channels 0-127, dout channels 128-255, input), the same images and transposing fragment reads as
// conv_wgrad_dma_kernel; three stages, 144 KB, one workgroup per CU.  Wavefronts 0-3 load (12 one-KiB pieces each
// per chunk, two chunks ahead, counted vmcnt waits), wavefronts 4-11 multiply (64 x 64 each, fragment reads half a
// chunk ahead of the MFMAs), one bare barrier per chunk.  Chunks in which the tap reaches no pixel (large
// dilations on small maps) cost nothing: every wavefront finds them by the same wavefront-wide vote (lane l looks
// at pixel l of the chunk), so they take no stage and no barrier.
// ------------------------------------------------------------------------------------------------
constexpr int WRING_STAGES = 3;
constexpr int WRING_SUB_B = 64 * 256;             // one [64 px][128 channels] image
constexpr int WRING_STAGE_B = 3 * WRING_SUB_B;

#ifdef SEGK_EXPERIMENTS
// timing experiment (SEGK_WGRAD_STAMP=1): cycles of the first workgroup's wavefronts per phase, [wavefront][8]:
// loaders: vote + bookkeeping, waiting for loads, barrier, issue, chunks that took a stage; multipliers: barrier
// waits, everything else in the K loop, -, -, chunks
__device__ long long wring_stamps[12 * 8];
__device__ int wring_stamp_on;
#endif

template <int MW>
__global__ __launch_bounds__(64 * (4 + MW), (4 + MW) / 4) void conv_wgrad_ring_kernel(const WgradDev p) {
  typedef bf16_t T;
  constexpr int BP = 64, MI = MW == 8 ? 4 : 8, MJ = 4, STAGES = WRING_STAGES, AHEAD = STAGES - 1;
  constexpr int LPW = 12;                           // loads per loading wavefront and chunk
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int wave_all = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool loader = wave_all < 4;
  const int wave = loader ? wave_all : wave_all - 4;
  const int lane = threadIdx.x & 63;
  int b;
  {
    const unsigned nb = gridDim.x, q = nb >> 3, r = nb & 7u, xcd = blockIdx.x & 7u;
    b = (int)(xcd * q + (xcd < r ? xcd : r) + (blockIdx.x >> 3));
  }
  const int cot = b % p.co_tiles; b /= p.co_tiles;
  const int cit = b % p.ci_tiles; b /= p.ci_tiles;
  const int taps = p.R * p.R;
  const int tap = b % taps;
  const int sp = b / taps;
  const int co0 = cot * 256, ci0 = cit * 128;
  const int dy = (tap / p.R - (p.R >> 1)) * p.dil, dx = (tap % p.R - (p.R >> 1)) * p.dil;
  const long long pbeg = (long long)sp * p.per_split;
  long long pend = pbeg + p.per_split;
  if (pend > p.M) pend = p.M;
  const int nchunks = pbeg < pend ? (int)((pend - pbeg + BP - 1) / BP) : 0;
  const long long bz = blockIdx.y;

  // lane l follows pixel l of the current chunk: which pixels of the chunk the tap reaches, as a 64-bit vote
  long long qm = pbeg + lane;
  int qy, qx;
  {
    const int rem = pixel_in_image(qm < p.M ? qm : 0, p.HW);
    qy = rem / p.W;
    qx = rem - qy * p.W;
  }
  const int step_y = BP / p.W, step_x = BP - step_y * p.W;
  auto chunk_vote = [&]() -> unsigned long long {
    const bool ok = qm < pend && (unsigned)(qy + dy) < (unsigned)p.H && (unsigned)(qx + dx) < (unsigned)p.W;
    const unsigned long long vote = __builtin_amdgcn_ballot_w64(ok);
    qm += BP;
    qx += step_x;
    qy += step_y;
    if (qx >= p.W) {
      qx -= p.W;
      ++qy;
    }
    while (qy >= p.H) qy -= p.H;   // next image (several at once only for images smaller than a chunk)
    return vote;
  };

  if (loader) {
    // piece q = i*4 + wave (i = 0..3) covers image rows (pixels) q*4 .. q*4+3; lane l -> row q*4 + (l>>4), LDS slot
    // l&15 of that row, which holds the source chunk the granule swizzle of wg_chunk_off<bf16_t> assigns to it (the
    // swizzle key of these rows does not depend on i)
    const int lr4 = lane >> 4, slot = lane & 15;
    const int fkey = lr4 | (((wave >> 1) & 1) << 2);
    const int csrc = ((((slot >> 1) ^ fkey) & 7) << 1) | (slot & 1);
    const bool dcol0 = co0 + csrc * 8 < p.Cout, dcol1 = co0 + 128 + csrc * 8 < p.Cout, xcol = ci0 + csrc * 8 < p.Cin;
    int doff[4], xoff[4], lrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      lrow[i] = (i * 4 + wave) * 4 + lr4;
      doff[i] = lrow[i] * (int)p.dout_ld_b + csrc * 16;
      xoff[i] = lrow[i] * (int)p.in_ld_b + csrc * 16;
    }
    const char* dbase = (p.groups > 0 ? p.dout_g[bz] : p.dout + bz * p.dout_batch_b) + (size_t)co0 * 2 + pbeg * p.dout_ld_b;
    const char* xbase = (p.groups > 0 ? p.in_g[bz] : p.in + bz * p.in_batch_b) + (size_t)ci0 * 2 +
                        (pbeg + (long long)dy * p.W + dx) * p.in_ld_b;
    const long long dstep = (long long)BP * p.dout_ld_b, xstep = (long long)BP * p.in_ld_b;
    int fill = 0, issued = 0, handed = 0;
    auto wait_behind = [&](int steps) {   // the immediate has to be a literal
      if (steps <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      static_assert(LPW == 12 && AHEAD == 2, "counted waits are written for 12 loads per chunk, two chunks ahead");
    };
#ifdef SEGK_EXPERIMENTS
    const bool stamp = wring_stamp_on != 0 && blockIdx.x == 0 && blockIdx.y == 0;
    long long c_wait = 0, c_bar = 0, c_iss = 0;
    const long long c_begin = __builtin_readcyclecounter();
#endif
    for (int ck = 0; ck < nchunks; ++ck) {
      const unsigned long long vote = chunk_vote();
      if (vote != 0ull) {
        if (issued >= AHEAD) {
          // hand over the oldest chunk in flight: its loads have landed (the next one's may still fly); behind the
          // barrier the multipliers have also left the chunk before it, whose stage is filled next
#ifdef SEGK_EXPERIMENTS
          const long long t0 = stamp ? __builtin_readcyclecounter() : 0;
#endif
          wait_behind(AHEAD - 1);
#ifdef SEGK_EXPERIMENTS
          const long long t1 = stamp ? __builtin_readcyclecounter() : 0;
#endif
          bare_barrier();
          ++handed;
#ifdef SEGK_EXPERIMENTS
          if (stamp) {
            c_wait += t1 - t0;
            c_bar += __builtin_readcyclecounter() - t1;
          }
#endif
        }
#ifdef SEGK_EXPERIMENTS
        const long long i0 = stamp ? __builtin_readcyclecounter() : 0;
#endif
        const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc((void*)dbase, 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xbase, 0, 0x7fffffff, 0x00020000);
        char* const sw = smem + fill * WRING_STAGE_B + wave * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          // rows the tap may not read get an offset beyond the resource's range: the load writes zeros, touches no
          // memory
          const bool ok = (vote >> lrow[i]) & 1ull;
          const int v0 = ok && dcol0 ? doff[i] : (int)0x80000000;
          const int v1 = ok && dcol1 ? doff[i] + 256 : (int)0x80000000;
          const int vx = ok && xcol ? xoff[i] : (int)0x80000000;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(drsrc, (lds_void_ptr)(sw + i * 4096), 16, v0, 0, 0, 0);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(drsrc, (lds_void_ptr)(sw + WRING_SUB_B + i * 4096), 16, v1, 0, 0, 0);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_void_ptr)(sw + 2 * WRING_SUB_B + i * 4096), 16, vx, 0, 0, 0);
        }
        fill = fill == STAGES - 1 ? 0 : fill + 1;
        ++issued;
#ifdef SEGK_EXPERIMENTS
        if (stamp) c_iss += __builtin_readcyclecounter() - i0;
#endif
      }
      dbase += dstep;
      xbase += xstep;
    }
#ifdef SEGK_EXPERIMENTS
    const long long c_loop = __builtin_readcyclecounter() - c_begin;
#endif
    while (handed < issued) {
      wait_behind(issued - 1 - handed < AHEAD - 1 ? issued - 1 - handed : AHEAD - 1);
      bare_barrier();
      ++handed;
    }
#ifdef SEGK_EXPERIMENTS
    if (stamp && lane == 0) {
      long long* o = wring_stamps + wave_all * 8;
      o[0] = c_loop - c_wait - c_bar - c_iss; o[1] = c_wait; o[2] = c_bar; o[3] = c_iss; o[4] = issued;
    }
#endif
    return;
  }

  // multipliers: the number of chunks that take a stage
  int total = 0;
  for (int ck = 0; ck < nchunks; ++ck) total += chunk_vote() != 0ull ? 1 : 0;
  total = __builtin_amdgcn_readfirstlane(total);

  const int wr = wave >> 1, wc = wave & 1;   // co block of MI * 16 channels, ci block of 64
  const bool wave_live = ci0 + wc * 64 < p.Cin && co0 + wr * (MI * 16) < p.Cout;
  f32x4 acc[MI][MJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < MJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (!wave_live) {
    for (int st = 0; st < total; ++st) bare_barrier();
  } else if (total > 0) {
    int stage = 0;
    u32x4 fa[2][MI], fb[2][MJ];          // [buffer][tile]
    auto read_half = [&](int buf, int ks) {
      const char* sd = smem + stage * WRING_STAGE_B + (MW == 8 ? wr >> 1 : wr) * WRING_SUB_B;
      const char* sx = smem + stage * WRING_STAGE_B + 2 * WRING_SUB_B;
#pragma unroll
      for (int j = 0; j < MJ; ++j) fb[buf][j] = WgFrag<T>::read(sx, ks, wc * 64 + j * 16, lane);
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[buf][i] = WgFrag<T>::read(sd, ks, (MW == 8 ? (wr & 1) * 64 : 0) + i * 16, lane);
      __builtin_amdgcn_sched_barrier(0);   // the reads are issued here, not after the MFMAs that follow
    };
    auto mfmas = [&](int buf) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j) WgFrag<T>::mma(fa[buf][i], fb[buf][j], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
    };
#ifdef SEGK_EXPERIMENTS
    const bool stamp = wring_stamp_on != 0 && blockIdx.x == 0 && blockIdx.y == 0;
    long long c_bar = 0;
    const long long c_begin = __builtin_readcyclecounter();
#endif
    bare_barrier();
    read_half(0, 0);
    // (the last chunk is peeled off: were its missing barrier and reads a branch inside the loop, the compiler's
    // counted waits in front of the MFMAs would have to assume the path without reads and wait for all of them)
    for (int st = 0; st + 1 < total; ++st) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      read_half(1, 1);
      stage = stage == STAGES - 1 ? 0 : stage + 1;
      mfmas(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every read of chunk st is complete: its stage is free
#ifdef SEGK_EXPERIMENTS
      const long long t0 = stamp ? __builtin_readcyclecounter() : 0;
#endif
      bare_barrier();
#ifdef SEGK_EXPERIMENTS
      if (stamp) c_bar += __builtin_readcyclecounter() - t0;
#endif
      read_half(0, 0);
      mfmas(1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    read_half(1, 1);
    mfmas(0);
    mfmas(1);
#ifdef SEGK_EXPERIMENTS
    if (stamp && lane == 0) {
      long long* o = wring_stamps + wave_all * 8;
      o[0] = c_bar; o[1] = __builtin_readcyclecounter() - c_begin - c_bar; o[4] = total;
    }
#endif
  }

  // acc[i][j][r] = dw[co = co0 + wr*MI*16 + i*16 + (lane>>4)*4 + r][ci = ci0 + wc*64 + j*16 + (lane&15)]
  const int fq = lane >> 4, fr = lane & 15;
  float* slab = p.slabs + (((size_t)bz * p.splits + sp) * taps + tap) * p.Cin * p.Cout;
#pragma unroll
  for (int j = 0; j < MJ; ++j) {
    const int ci = ci0 + wc * 64 + j * 16 + fr;
    if (ci >= p.Cin) continue;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int co = co0 + wr * (MI * 16) + i * 16 + fq * 4;
      if (co >= p.Cout) continue;
      *reinterpret_cast<f32x4*>(slab + (size_t)ci * p.Cout + co) = acc[i][j];
    }
  }
}

#ifdef SEGK_EXPERIMENTS
// (experiment builds only, not part of the interface) out[12][8]: the stamps of the last ring weight-gradient launch
extern "C" int segk_debug_wring_stamps(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(wring_stamps), sizeof(long long) * 96) == hipSuccess ? 0 : -1;
}
#endif

static int launch_wgrad_ring(WgradDev& d, int batch, hipStream_t st, const char* name) {
#ifdef SEGK_EXPERIMENTS
  {
    static const char* e = experiment("SEGK_WGRAD_STAMP");
    static bool told = false;
    if (!told) {
      const int on = e && e[0] == '1' ? 1 : 0;
      (void)hipMemcpyToSymbol(HIP_SYMBOL(wring_stamp_on), &on, sizeof(int));
      told = true;
    }
  }
#endif
  d.co_tiles = ceil_div(d.Cout, 256);
  d.ci_tiles = ceil_div(d.Cin, 128);
  const long long nb = (long long)d.co_tiles * d.ci_tiles * d.R * d.R * d.splits;
  if (nb >= 2147483647LL) {
    set_error("%s: too many tiles", name);
    return SEGK_INVALID_ARG;
  }
  // per-lane row offsets are 32-bit: 64 rows of the operands have to stay below 2 GB
  if (d.dout_ld_b * 64 >= 0x7fffffffLL || d.in_ld_b * 64 >= 0x7fffffffLL) {
    set_error("%s: row stride too large for the ring tile", name);
    return SEGK_UNSUPPORTED_SHAPE;
  }
  constexpr int lds = WRING_STAGES * WRING_STAGE_B;
#ifdef SEGK_EXPERIMENTS
  static const char* e = experiment("SEGK_WGRAD_RING");   // 4: four multiplying wavefronts of 128 x 64
  if (e && e[0] == '4') {
    static LdsConfigured configured4;
    int rc4 = set_lds(conv_wgrad_ring_kernel<4>, lds, name, configured4);
    if (rc4 != SEGK_OK) return rc4;
    conv_wgrad_ring_kernel<4><<<dim3((unsigned)nb, (unsigned)batch), 512, lds, st>>>(d);
    return SEGK_OK;
  }
#endif
  static LdsConfigured configured;
  int rc = set_lds(conv_wgrad_ring_kernel<8>, lds, name, configured);
  if (rc != SEGK_OK) return rc;
  conv_wgrad_ring_kernel<8><<<dim3((unsigned)nb, (unsigned)batch), 768, lds, st>>>(d);
  return SEGK_OK;
}

// px_tiles < 0: every pixel tile from d.m_begin to the end
template <typename TO, int PX = 128>
static int launch_dma(ConvDev& d, hipStream_t st, int batch = 1, long long px_tiles = -1) {
  constexpr int lds = (128 + PX) * 128;
  d.n_tiles = ceil_div(d.Cout, 128);
  if (px_tiles < 0) px_tiles = (d.M - d.m_begin + PX - 1) / PX;
  const long long blocks = px_tiles * d.n_tiles;
  if (blocks >= 2147483647LL) {
    set_error("conv: too many tiles");
    return SEGK_INVALID_ARG;
  }
  {
    // at most two workgroups per CU: the form with specialised wavefronts (SEGK_CONV_STAGES=1 in experiment builds
    // keeps the one-stage tile for comparison)
    static const char* st_env = experiment("SEGK_CONV_STAGES");
    const bool want = st_env ? st_env[0] == '3' : blocks * batch <= 2LL * cu_count();
    if (PX == 128 && want) {
      constexpr int lds3 = 2 * 256 * 128;
      static LdsConfigured configured3;  // per instantiation
      int rc = set_lds(conv_fwd_dma_kernel<TO, 128, 3>, lds3, "conv_fwd_dma (specialised wavefronts)", configured3);
      if (rc != SEGK_OK) return rc;
      conv_fwd_dma_kernel<TO, 128, 3><<<dim3((unsigned)blocks, (unsigned)batch), 512, lds3, st>>>(d);
      return SEGK_OK;
    }
  }
#ifdef SEGK_EXPERIMENTS
  // SEGK_CONV_STAGES=2: the two-stage form of the 128 x 128 tile.  Measured (rocprofv3 kernel durations): level with
  // the one-stage form on launches of one workgroup per CU (55.4 us both on the 274-tile 3x3 of ResNet layer 3) and
  // 25-45 % slower on larger ones (two workgroups per CU instead of three), so it is not a default anywhere.
  static const char* stages = experiment("SEGK_CONV_STAGES");
  if (PX == 128 && stages && stages[0] == '2') {
    static LdsConfigured configured2;  // per instantiation
    int rc = set_lds(conv_fwd_dma_kernel<TO, 128, 2>, 2 * 256 * 128, "conv_fwd_dma (two stages)", configured2);
    if (rc != SEGK_OK) return rc;
    conv_fwd_dma_kernel<TO, 128, 2><<<dim3((unsigned)blocks, (unsigned)batch), 256, 2 * 256 * 128, st>>>(d);
    return SEGK_OK;
  }
#endif
  static LdsConfigured configured;  // per instantiation
  int rc = set_lds(conv_fwd_dma_kernel<TO, PX>, lds, "conv_fwd_dma", configured);
  if (rc != SEGK_OK) return rc;
  conv_fwd_dma_kernel<TO, PX><<<dim3((unsigned)blocks, (unsigned)batch), 256, lds, st>>>(d);
  return SEGK_OK;
}

// the 128 x 256 LDS-DMA tile serves the big-tile plan for bf16; SEGK_CONV_BIGDMA=0 brings back the register-staged
// 256 x 256 tile
static bool use_bigdma() {
  static const char* e = experiment("SEGK_CONV_BIGDMA");
  return !(e && e[0] == '0');
}

// default for bf16 operands with 128-byte K rows; SEGK_CONV_DMA=0 falls back to the register-staged small tile
static bool use_dma() {
  static const char* e = experiment("SEGK_CONV_DMA");
  return !(e && e[0] == '0');
}

template <typename TO, int ABL = 0>
static int launch_pipe(ConvDev& d, hipStream_t st, int batch = 1) {
  static LdsConfigured configured;  // per instantiation
  int rc = set_lds(conv_fwd_pipe_kernel<TO, ABL>, PP_LDS, "conv_fwd_pipe", configured);
  if (rc != SEGK_OK) return rc;
  d.n_tiles = d.Cout / TilePipe::BCO;
  const long long blocks = ((d.M + TilePipe::BPX - 1) / TilePipe::BPX) * d.n_tiles;
  if (blocks >= 2147483647LL) {
    set_error("conv: too many tiles");
    return SEGK_INVALID_ARG;
  }
  conv_fwd_pipe_kernel<TO, ABL><<<dim3((unsigned)blocks, (unsigned)batch), 512, PP_LDS, st>>>(d);
  return SEGK_OK;
}

// The deep-pipelined kernel is opt-in (SEGK_CONV_PIPE=1): measured on the decoder's 3x3 convolutions it ties the
// register-staged big tile, because both are bound by the rate at which a CU can pull operand tiles through its
// vector L1 (about 20 bytes per cycle), not by staging registers or by the barrier structure.
static bool use_pipe() {
  static const char* e = experiment("SEGK_CONV_PIPE");
  return e && e[0] == '1';
}

// Rows of statistic partials = pixel tiles of the geometry chosen for this problem.
static bool use_big_tile(const ConvDev& d, bool wide, int batch = 1) {
  // the big tile needs 128-byte K rows, full 256-channel tiles and enough tiles to fill 256 CUs twice
  if (!wide || d.Cout % TileBig::BCO != 0) return false;
  static const char* force = experiment("SEGK_CONV_TILE");  // experiments: "small" / "big"
  if (force && force[0] == 's') return false;
  if (force && (force[0] == 'b' || force[0] == 'h')) return true;
  // Rounds of workgroups each geometry needs (a last round under 15 % full is not counted: the CUs drift apart
  // enough to absorb it, and the big tile hands such a remainder to a small-tile launch anyway).  One workgroup of
  // the big tile per CU, three of the small (LDS-DMA) one; a round of small tiles takes 0.77 of a round of big ones
  // (measured on the decoder convolutions), and covers a quarter of the output per workgroup.
  const long long tiles = ((d.M + TileBig::BPX - 1) / TileBig::BPX) * (d.Cout / TileBig::BCO) * batch;
  const double cus = cu_count();
  auto rounds = [](double r) {
    const double whole = floor(r);
    const double n = whole + (r - whole > 0.15 ? 1.0 : 0.0);
    return n < 1.0 ? 1.0 : n;
  };
  return rounds(tiles / cus) < 0.77 * rounds(4.0 * tiles / (3.0 * cus));
}

// Tile plan of one forward launch.  With one 256 x 256 workgroup per CU a problem of T tiles takes ceil(T / CUs)
// rounds; when the last round would be nearly empty (1041 tiles on 256 CUs: 17 tiles in round five) the big tile
// only takes the whole rounds and the remaining pixels go to a second launch of 128 x 128 tiles, which fills the
// chip four times finer.
struct TilePlan {
  bool big;
  long long main_tiles;  // pixel tiles of the first launch
  bool tail;             // remaining pixels in a second, small-tile launch
  int ring;              // form of the 128 x 160 ring tile (bf16 operands, 128-byte K rows), 0 = not used
};

// Which form of the ring tile serves a problem, 0 = none: every convolution with bf16 operands and 128-byte K rows
// takes form 1 (three stages, four loading and eight multiplying wavefronts, one persistent workgroup per CU).
// Kernel durations (rocprofv3, batch 16) against the one-stage LDS-DMA tiles it replaces: ResNet layer-3 3x3 46 ->
// 28 us, layer-4 3x3 109 -> 86, ASPP 1x1 45 -> 27, one dilated ASPP 3x3 256 -> 147, the three of them in one launch
// 555 -> 492, decoder 3x3 514 -> 432 and 400 -> 362, their data gradients 540 -> 478 and 395 -> 355, the ASPP data
// gradient (four segments, fp32 output) 481 -> 392, decoder 1x1 onto 48 channels 59 -> 55.
// Experiment builds keep three other forms for comparison (SEGK_CONV_RING=2: two stages, four multipliers of 64 x 80,
// two workgroups per CU - level on the large launches, 40 % slower on launches of one workgroup per CU; 3: four
// stages - level; 4: eight loading wavefronts - level: the K loop is not bound by the loaders' instruction issue).
// Form 6 is the same kernel on 256(co) x 160(px) tiles (64 x 80 per multiplying wavefront): half the barriers, 0.72 of
// the LDS-DMA bytes and 0.64 of the fragment reads per MFMA, but half as many tiles to spread over the CUs.  It is
// taken when the output channels fill 256-wide tiles and the rounds of workgroups (one per CU, each walking over
// ceil(tiles / CUs) tiles) cost less that way; RING_WIDE_STEP is the time of a K step of the wide tile relative to
// two steps of the narrow one.
constexpr double RING_WIDE_STEP = 0.82;
static int ring_form(const ConvDev& d, bool wide, bool bf16, int batch) {
  if (!wide || !bf16) return 0;
  static const char* e = experiment("SEGK_CONV_RING");
  if (e && e[0] == '6') return d.Cout % 256 == 0 ? 6 : 1;
  if (e) return e[0] >= '1' && e[0] <= '5' ? e[0] - '0' : 0;
  if (d.Cout % 256 != 0) return 1;
  const long long px_tiles = (d.M + RING_PX - 1) / RING_PX, cus = cu_count();
  const long long narrow = px_tiles * (d.Cout / 128) * batch, wider = px_tiles * (d.Cout / 256) * batch;
  const double t_narrow = (double)((narrow + cus - 1) / cus), t_wide = 2.0 * RING_WIDE_STEP * (double)((wider + cus - 1) / cus);
  return t_wide < t_narrow ? 6 : 1;
}

static TilePlan plan_tiles(const ConvDev& d, bool wide, int batch, bool bf16) {
  TilePlan pl;
  pl.ring = ring_form(d, wide, bf16, batch);
  if (pl.ring) {
    pl.big = false;
    pl.tail = false;
    pl.main_tiles = (d.M + RING_PX - 1) / RING_PX;
    return pl;
  }
  pl.big = use_big_tile(d, wide, batch);
  pl.tail = false;
  pl.main_tiles = (d.M + (pl.big ? TileBig::BPX : TileSmall::BPX) - 1) / (pl.big ? TileBig::BPX : TileSmall::BPX);
  static const char* e = experiment("SEGK_CONV_TAIL");
  if (pl.big && batch == 1 && !use_pipe() && !(e && e[0] == '0')) {
    const long long n_co = d.Cout / TileBig::BCO, blocks = pl.main_tiles * n_co, slots = cu_count();
    const long long rounds = blocks / slots, rem = blocks % slots;
    if (rounds >= 1 && rem > 0 && rem * 10 <= slots * 3) {
      const long long keep = rounds * slots / n_co;
      if (keep >= 1 && keep < pl.main_tiles) {
        pl.main_tiles = keep;
        pl.tail = true;
      }
    }
  }
  return pl;
}

static int plan_stat_rows(const ConvDev& d, const TilePlan& pl) {
  if (pl.ring) return (int)(pl.main_tiles * 2);   // a row per 80-pixel half of a ring tile
  if (!pl.tail) return (int)pl.main_tiles;
  return (int)(pl.main_tiles + (d.M - pl.main_tiles * TileBig::BPX + TileSmall::BPX - 1) / TileSmall::BPX);
}

}  // namespace segk

using namespace segk;

static int conv_check(const segk_conv_args* a) {
  SEGK_REQUIRE(a, "conv: null args");
  SEGK_REQUIRE(valid_dtype(a->dtype) && valid_dtype(a->out_dtype), "conv: bad dtype");
  SEGK_REQUIRE(a->N > 0 && a->H > 0 && a->W > 0 && a->Cout > 0, "conv: bad shape N=%d H=%d W=%d Cout=%d", a->N, a->H,
               a->W, a->Cout);
  SEGK_REQUIRE(a->nseg >= 1 && a->nseg <= SEGK_MAX_SEG, "conv: nseg=%d", a->nseg);
  SEGK_REQUIRE(a->out && a->out_ld >= ((a->Cout + 3) & ~3) && a->out_ld % 4 == 0,
               "conv: out_ld=%d must be a multiple of 4 and >= Cout rounded up to 4 (Cout=%d)", a->out_ld, a->Cout);
  SEGK_REQUIRE(((uintptr_t)a->out % (a->out_dtype == SEGK_BF16 ? 8 : 16)) == 0, "conv: output base misaligned");
  const int es = dtype_size(a->dtype);
  for (int s = 0; s < a->nseg; ++s) {
    const segk_conv_seg& g = a->seg[s];
    SEGK_REQUIRE(g.in && g.w && aligned16(g.in) && aligned16(g.w), "conv: segment %d pointers null or not 16-byte aligned", s);
    SEGK_REQUIRE(g.R == 1 || g.R == 3, "conv: segment %d kernel size %d (1 or 3)", s, g.R);
    SEGK_REQUIRE(g.dil >= 1, "conv: segment %d dilation %d", s, g.dil);
    SEGK_REQUIRE(g.Ck > 0 && g.Ck % 32 == 0 && g.Ck <= g.in_ld, "conv: segment %d Ck=%d must be a multiple of 32 and <= ld=%d",
                 s, g.Ck, g.in_ld);
    SEGK_REQUIRE(((long long)g.in_ld * es) % 16 == 0, "conv: segment %d row stride not a multiple of 16 bytes", s);
    SEGK_REQUIRE(g.w_row_stride >= 0 && g.w_tap_stride >= 0 && (g.w_row_stride * es) % 16 == 0 &&
                     (g.w_tap_stride * es) % 16 == 0 && g.w_row_stride * es < 2147483647LL / 256 &&
                     (g.w_row_stride == 0 || g.w_row_stride >= g.Ck),
                 "conv: segment %d weight strides (%lld, %lld) must be non-negative multiples of 16 bytes, rows >= Ck",
                 s, g.w_row_stride, g.w_tap_stride);
  }
  return SEGK_OK;
}

// Device-side description of the problem; returns whether every segment has 128-byte K rows.
static bool fill_dev(const segk_conv_args* a, ConvDev& d) {
  d.H = a->H; d.W = a->W; d.HW = a->H * a->W;
  d.M = (long long)a->N * d.HW;
  d.Cout = a->Cout; d.nseg = a->nseg;
  d.n_tiles = 0;
  d.ring_tiles = 0;
  {
    static const char* abl = experiment("SEGK_DMA_ABL");
    d.abl = abl ? atoi(abl) : 0;
  }
  const int es = dtype_size(a->dtype);
  bool wide = true;
  for (int s = 0; s < SEGK_MAX_SEG; ++s) {
    const segk_conv_seg& g = a->seg[s < a->nseg ? s : 0];
    d.seg[s].in = (const char*)g.in;
    d.seg[s].w = (const char*)g.w;
    d.seg[s].in_ld_b = (long long)g.in_ld * es;
    d.seg[s].ck_b = g.Ck * es;
    d.seg[s].w_ld_b = (int)((g.w_row_stride > 0 ? g.w_row_stride : (long long)g.Ck) * es);
    d.seg[s].in_batch_b = 0;
    d.seg[s].w_batch_b = (long long)a->w_batch * es;
    d.seg[s].w_tap_b = (g.w_tap_stride > 0 ? g.w_tap_stride : (long long)a->Cout * g.Ck) * es;
    d.seg[s].R = g.R;
    d.seg[s].dil = g.dil;
    if (s < a->nseg && (d.seg[s].ck_b % 128) != 0) wide = false;
  }
  d.out = (char*)a->out; d.out_ld = a->out_ld;
  d.out_batch_b = (long long)a->out_batch * dtype_size(a->out_dtype);
  d.dil_step = a->dil_step;
  d.stat_batch = 0;  // set once the tile geometry (= number of partial rows) is known
  d.m_begin = 0; d.stat_row0 = 0;
  d.bias = a->bias; d.img_bias = a->img_bias; d.stat = a->stat_partials;
  d.addend = (const char*)a->addend;
  d.addend_ld_b = (long long)a->addend_ld * dtype_size(a->out_dtype);
  d.addend_bits = (const uint8_t*)a->addend_bits;
  d.addend_bits_ld = a->addend_bits_ld;
  d.bst_rows = nullptr;  // set by segk_conv_fwd once the launch is known to emit them from every tile
  d.bst_x = (const char*)a->bst_x;
  d.bst_x_ld_b = (long long)a->bst_x_ld * 2;
  d.bst_mean = a->bst_mean; d.bst_inv_std = a->bst_inv_std; d.bst_scale = a->bst_scale; d.bst_shift = a->bst_shift;
  d.bst_bits = (const uint8_t*)a->bst_bits;
  d.bst_bits_ld = a->bst_bits_ld;
  d.bst_relu = a->bst_relu;
  return wide;
}

// Whether a launch with this plan writes the backward-statistics rows from every tile: ring tiles whose stores take the
// 16-byte path (bf16 output, whole tiles of channels, aligned rows, no biases, no addend), one problem per launch.
static bool bstat_problem(const segk_conv_args* a, const TilePlan& plan, int batch) {
  if (!a->bst_x || !a->bst_mean || !a->bst_inv_std) return false;
  // (the 128-channel form only: the 256-channel form has no registers to spare, and the loads its epilogue would need
  // cost it more than the stand-alone pass takes)
  if (plan.ring != 1 || a->out_dtype != SEGK_BF16 || batch != 1 || a->bias || a->img_bias || a->addend) return false;
  if (a->Cout % 128 != 0) return false;
  if (!aligned16(a->out) || a->out_ld % 8 != 0) return false;
  if (!aligned16(a->bst_x) || a->bst_x_ld % 8 != 0 || a->bst_x_ld < a->Cout) return false;
  if (!aligned16(a->bst_mean) || !aligned16(a->bst_inv_std)) return false;
  if (a->bst_bits) return (long long)a->bst_bits_ld * 8 >= a->Cout;
  if (a->bst_relu) return a->bst_scale && a->bst_shift && aligned16(a->bst_scale) && aligned16(a->bst_shift);
  return true;
}

extern "C" int segk_conv_bstat_rows(const segk_conv_args* a) {
  if (!a || a->N <= 0 || a->H <= 0 || a->W <= 0 || a->nseg < 1 || a->nseg > SEGK_MAX_SEG || !valid_dtype(a->dtype))
    return SEGK_INVALID_ARG;
  if (small_rows_problem(a)) return 0;
  ConvDev d;
  const bool wide = fill_dev(a, d);
  const int batch = a->batch > 1 ? a->batch : 1;
  const TilePlan plan = plan_tiles(d, wide, batch, a->dtype == SEGK_BF16);
  return bstat_problem(a, plan, batch) ? plan_stat_rows(d, plan) : 0;
}

extern "C" int segk_conv_stat_rows(const segk_conv_args* a) {
  if (!a || a->N <= 0 || a->H <= 0 || a->W <= 0 || a->nseg < 1 || a->nseg > SEGK_MAX_SEG || !valid_dtype(a->dtype))
    return SEGK_INVALID_ARG;
  if (small_rows_problem(a)) return 1;
  ConvDev d;
  const bool wide = fill_dev(a, d);
  return plan_stat_rows(d, plan_tiles(d, wide, a->batch > 1 ? a->batch : 1, a->dtype == SEGK_BF16));
}

extern "C" int segk_conv_fwd(const segk_conv_args* a, segk_stream_t stream) {
  int rc = conv_check(a);
  if (rc != SEGK_OK) return rc;
  ConvDev d;
  const bool wide = fill_dev(a, d);
  const int batch = a->batch > 1 ? a->batch : 1;
  SEGK_REQUIRE(batch <= 65535 && (batch == 1 || (!a->bias && !a->img_bias)),
               "conv: batch=%d (biases are not batched)", batch);
  if (small_rows_problem(a)) {
    const unsigned blocks = (unsigned)ceil_div(a->Cout, SMALL_ROWS_CO);
    if (a->out_dtype == SEGK_BF16) conv_small_rows_kernel<bf16_t><<<blocks, 256, 0, (hipStream_t)stream>>>(d);
    else conv_small_rows_kernel<float><<<blocks, 256, 0, (hipStream_t)stream>>>(d);
    SEGK_CHECK_LAUNCH("conv_fwd");
    return SEGK_OK;
  }
  const TilePlan plan = plan_tiles(d, wide, batch, a->dtype == SEGK_BF16);
  const bool big = plan.big;
  if (a->addend) {
    SEGK_REQUIRE(a->addend_ld >= a->Cout && ((uintptr_t)a->addend % (a->out_dtype == SEGK_BF16 ? 8 : 16)) == 0 &&
                     ((long long)a->addend_ld * dtype_size(a->out_dtype)) % (a->out_dtype == SEGK_BF16 ? 8 : 16) == 0,
                 "conv: addend misaligned or addend_ld=%d < Cout=%d", a->addend_ld, a->Cout);
    SEGK_UNSUPPORTED(plan.ring && a->Cout % 128 == 0 && !a->bias && !a->img_bias && batch == 1,
                     "conv: an addend needs bf16 operands with 128-byte K rows, Cout a multiple of 128 and no biases");
    SEGK_REQUIRE(!a->addend_bits || (long long)a->addend_bits_ld * 8 >= a->Cout, "conv: addend_bits_ld=%d < Cout / 8 (Cout=%d)",
                 a->addend_bits_ld, a->Cout);
  } else {
    SEGK_REQUIRE(!a->addend_bits, "conv: addend_bits without an addend");
  }
  if (a->bst_rows) {
    SEGK_UNSUPPORTED(bstat_problem(a, plan, batch),
                     "conv: backward statistics need the 128-channel ring tiles with bf16 output, whole channel tiles, 16-byte "
                     "aligned rows and coefficient vectors, no biases, no addend (ask segk_conv_bstat_rows first)");
    d.bst_rows = a->bst_rows;
  }
  d.stat_batch = (long long)plan_stat_rows(d, plan) * 2 * d.Cout;
  hipStream_t st = (hipStream_t)stream;
#ifdef SEGK_EXPERIMENTS
  static const char* abl = experiment("SEGK_CONV_ABL");
  if (abl && abl[0] && big && use_pipe() && a->dtype == SEGK_BF16 && a->out_dtype == SEGK_BF16) {
    rc = abl[0] == '1' ? launch_pipe<bf16_t, 1>(d, st)
       : abl[0] == '2' ? launch_pipe<bf16_t, 2>(d, st)
       : abl[0] == '3' ? launch_pipe<bf16_t, 3>(d, st)
       : abl[0] == '6' ? launch_pipe<bf16_t, 6>(d, st)
       : abl[0] == '7' ? launch_pipe<bf16_t, 7>(d, st)
       : abl[0] == '8' ? launch_pipe<bf16_t, 8>(d, st)
       : abl[0] == '9' ? launch_pipe<bf16_t, 9>(d, st)
                       : launch_pipe<bf16_t, 4>(d, st);
    if (rc != SEGK_OK) return rc;
    SEGK_CHECK_LAUNCH("conv_fwd");
    return SEGK_OK;
  }
  if (abl && abl[0] && big && a->dtype == SEGK_BF16 && a->out_dtype == SEGK_BF16) {
    rc = abl[0] == '1' ? launch_fwd<bf16_t, 128, bf16_t, TileBig, 1>(d, st)
       : abl[0] == '2' ? launch_fwd<bf16_t, 128, bf16_t, TileBig, 2>(d, st)
       : abl[0] == '5' ? launch_fwd<bf16_t, 128, bf16_t, TileBig, 5>(d, st)
       : abl[0] == '6' ? launch_fwd<bf16_t, 128, bf16_t, TileBig, 6>(d, st)
                       : launch_fwd<bf16_t, 128, bf16_t, TileBig, 3>(d, st);
    if (rc != SEGK_OK) return rc;
    SEGK_CHECK_LAUNCH("conv_fwd");
    return SEGK_OK;
  }
#endif
  if (plan.ring) {
    const int f = plan.ring;
    (void)f;
#define SEGK_RING(S, W, L) (a->out_dtype == SEGK_BF16 ? launch_ring<bf16_t, S, W, L>(d, st, batch, plan.main_tiles) \
                                                      : launch_ring<float, S, W, L>(d, st, batch, plan.main_tiles))
#ifdef SEGK_EXPERIMENTS
    if (f == 2) rc = SEGK_RING(2, 4, 4);
    else if (f == 3) rc = SEGK_RING(4, 8, 4);
    else if (f == 4) rc = SEGK_RING(3, 8, 8);
    else if (f == 5) rc = SEGK_RING(3, 4, 4);
    else
#endif
    if (d.bst_rows)   // (bstat_problem: the 128-channel form, bf16 output)
      rc = launch_ring<bf16_t, 3, 8, 4, 128, true>(d, st, batch, plan.main_tiles);
    else if (f == 6)
      rc = a->out_dtype == SEGK_BF16 ? launch_ring<bf16_t, 3, 8, 4, 256>(d, st, batch, plan.main_tiles)
                                     : launch_ring<float, 3, 8, 4, 256>(d, st, batch, plan.main_tiles);
    else
      rc = SEGK_RING(3, 8, 4);
#undef SEGK_RING
    if (rc != SEGK_OK) return rc;
    SEGK_CHECK_LAUNCH("conv_fwd");
    return SEGK_OK;
  }
#define SEGK_LAUNCH(T, TO)                                                          \
  (big ? launch_fwd<T, 128, TO, TileBig>(d, st, batch)                              \
       : (wide ? launch_fwd<T, 128, TO, TileSmall>(d, st, batch) : launch_fwd<T, 64, TO, TileSmall>(d, st, batch)))
  if (a->dtype == SEGK_BF16 && wide && !big && use_dma()) {
    rc = a->out_dtype == SEGK_BF16 ? launch_dma<bf16_t>(d, st, batch) : launch_dma<float>(d, st, batch);
    if (rc != SEGK_OK) return rc;
    SEGK_CHECK_LAUNCH("conv_fwd");
    return SEGK_OK;
  }
  if (plan.tail) {
    // whole rounds on the big tile, the remainder on the small one
#define SEGK_SPLIT(T, TO)                                                               \
  do {                                                                                  \
    rc = (sizeof(T) == 2 && use_bigdma()) ? launch_dma<TO, 256>(d, st, 1, plan.main_tiles) \
                                          : launch_fwd<T, 128, TO, TileBig>(d, st, 1, plan.main_tiles); \
    if (rc == SEGK_OK) {                                                                \
      SEGK_CHECK_LAUNCH("conv_fwd");                                                    \
      d.m_begin = plan.main_tiles * TileBig::BPX;                                       \
      d.stat_row0 = plan.main_tiles;                                                    \
      rc = (sizeof(T) == 2 && use_dma()) ? launch_dma<TO>(d, st, 1)                     \
                                         : launch_fwd<T, 128, TO, TileSmall>(d, st, 1); \
    }                                                                                   \
  } while (0)
    if (a->dtype == SEGK_BF16) {
      if (a->out_dtype == SEGK_BF16) SEGK_SPLIT(bf16_t, bf16_t); else SEGK_SPLIT(bf16_t, float);
    } else {
      if (a->out_dtype == SEGK_BF16) SEGK_SPLIT(float, bf16_t); else SEGK_SPLIT(float, float);
    }
#undef SEGK_SPLIT
  } else if (a->dtype == SEGK_BF16 && big && use_bigdma()) {
    rc = a->out_dtype == SEGK_BF16 ? launch_dma<bf16_t, 256>(d, st, batch) : launch_dma<float, 256>(d, st, batch);
#ifdef SEGK_EXPERIMENTS
  } else if (a->dtype == SEGK_BF16 && big && use_pipe()) {
    rc = a->out_dtype == SEGK_BF16 ? launch_pipe<bf16_t>(d, st, batch) : launch_pipe<float>(d, st, batch);
#endif
  } else if (a->dtype == SEGK_BF16) {
    rc = a->out_dtype == SEGK_BF16 ? SEGK_LAUNCH(bf16_t, bf16_t) : SEGK_LAUNCH(bf16_t, float);
  } else {
    // fp32 rows of 32 channels are always 128 bytes
    rc = a->out_dtype == SEGK_BF16
             ? (big ? launch_fwd<float, 128, bf16_t, TileBig>(d, st, batch) : launch_fwd<float, 128, bf16_t, TileSmall>(d, st, batch))
             : (big ? launch_fwd<float, 128, float, TileBig>(d, st, batch) : launch_fwd<float, 128, float, TileSmall>(d, st, batch));
  }
#undef SEGK_LAUNCH
  if (rc != SEGK_OK) return rc;
  SEGK_CHECK_LAUNCH("conv_fwd");
  return SEGK_OK;
}

// shape and stride checks shared by the single and the grouped weight gradient; fills everything but the operands
static int wgrad_problem(const segk_wgrad_args* a, WgradDev& d, const char* name) {
  SEGK_REQUIRE(valid_dtype(a->dtype), "%s: bad dtype", name);
  SEGK_REQUIRE(a->N > 0 && a->H > 0 && a->W > 0, "%s: bad shape", name);
  SEGK_REQUIRE(a->Cin > 0 && a->Cin % 32 == 0 && a->Cin <= a->in_ld, "%s: Cin=%d must be a multiple of 32 and <= ld=%d", name,
               a->Cin, a->in_ld);
  SEGK_REQUIRE(a->Cout > 0 && a->Cout % 16 == 0 && a->Cout <= a->dout_ld, "%s: Cout=%d must be a multiple of 16 and <= ld=%d",
               name, a->Cout, a->dout_ld);
  SEGK_REQUIRE((a->R == 1 || a->R == 3) && a->dil >= 1 && a->splits >= 1, "%s: R=%d dil=%d splits=%d", name, a->R, a->dil,
               a->splits);
  const int es = dtype_size(a->dtype);
  SEGK_REQUIRE(a->slabs && aligned16(a->slabs) && ((long long)a->in_ld * es) % 16 == 0 && ((long long)a->dout_ld * es) % 16 == 0,
               "%s: slabs must be 16-byte aligned and row strides multiples of 16 bytes", name);
  SEGK_REQUIRE(a->tile == 0 || a->tile == 128 || a->tile == 256 || a->tile == 384, "%s: tile=%d (0, 128, 256 or 384)", name,
               a->tile);
  d.H = a->H; d.W = a->W; d.HW = a->H * a->W;
  d.M = (long long)a->N * d.HW;
  d.in = nullptr; d.dout = nullptr;
  d.in_ld_b = (long long)a->in_ld * es; d.dout_ld_b = (long long)a->dout_ld * es;
  d.Cin = a->Cin; d.Cout = a->Cout; d.R = a->R; d.dil = a->dil; d.splits = a->splits;
  const int bp = a->dtype == SEGK_BF16 ? WgFrag<bf16_t>::BP : WgFrag<float>::BP;
  long long per = (d.M + a->splits - 1) / a->splits;
  per = (per + bp - 1) / bp * bp;
  d.per_split = per;
  d.slabs = a->slabs;
  d.in_batch_b = 0; d.dout_batch_b = 0;
  d.groups = 0;
  for (int g = 0; g < SEGK_WGRAD_GROUP_MAX; ++g) d.in_g[g] = d.dout_g[g] = nullptr;
  return SEGK_OK;
}

extern "C" int segk_conv_wgrad(const segk_wgrad_args* a, segk_stream_t stream) {
  SEGK_REQUIRE(a && a->in && a->dout && a->slabs, "wgrad: null pointer");
  SEGK_REQUIRE(aligned16(a->in) && aligned16(a->dout), "wgrad: operands must be 16-byte aligned");
  WgradDev d;
  int rc = wgrad_problem(a, d, "wgrad");
  if (rc != SEGK_OK) return rc;
  d.in = (const char*)a->in; d.dout = (const char*)a->dout;
  hipStream_t st = (hipStream_t)stream;
  rc = a->dtype == SEGK_BF16 ? launch_wgrad<bf16_t>(d, 1, st, "conv_wgrad", a->tile)
                             : launch_wgrad<float>(d, 1, st, "conv_wgrad", a->tile);
  if (rc != SEGK_OK) return rc;
  SEGK_CHECK_LAUNCH("conv_wgrad");
  return SEGK_OK;
}

extern "C" int segk_conv_wgrad_group(const segk_wgrad_group_args* a, segk_stream_t stream) {
  SEGK_REQUIRE(a && a->conv.slabs, "wgrad_group: null pointer");
  SEGK_REQUIRE(a->groups >= 1 && a->groups <= SEGK_WGRAD_GROUP_MAX, "wgrad_group: groups=%d (1..%d)", a->groups,
               SEGK_WGRAD_GROUP_MAX);
  if (a->conv.dtype != SEGK_BF16 || a->conv.Cout % 256 != 0 || (a->conv.tile != 0 && a->conv.tile != 384)) {
    set_error("wgrad_group: bf16 onto a multiple of 256 output channels only (the ring tile), got dtype=%d Cout=%d tile=%d",
              a->conv.dtype, a->conv.Cout, a->conv.tile);
    return SEGK_UNSUPPORTED_SHAPE;
  }
  WgradDev d;
  int rc = wgrad_problem(&a->conv, d, "wgrad_group");
  if (rc != SEGK_OK) return rc;
  for (int g = 0; g < a->groups; ++g) {
    SEGK_REQUIRE(a->in[g] && a->dout[g] && aligned16(a->in[g]) && aligned16(a->dout[g]),
                 "wgrad_group: operands of convolution %d null or not 16-byte aligned", g);
    d.in_g[g] = (const char*)a->in[g];
    d.dout_g[g] = (const char*)a->dout[g];
  }
  d.groups = a->groups;
  rc = launch_wgrad_ring(d, a->groups, (hipStream_t)stream, "conv_wgrad_group");
  if (rc != SEGK_OK) return rc;
  SEGK_CHECK_LAUNCH("conv_wgrad_group");
  return SEGK_OK;
}

// ------------------------------------------------------------------------------------------------
// Batched matrix products on the same kernels (attention modules)
// ------------------------------------------------------------------------------------------------
extern "C" int segk_gemm_nt(const segk_gemm_args* a, segk_stream_t stream) {
  SEGK_REQUIRE(a, "gemm_nt: null args");
  SEGK_REQUIRE(valid_dtype(a->dtype) && valid_dtype(a->c_dtype), "gemm_nt: bad dtype");
  SEGK_REQUIRE(a->batch > 0 && a->batch <= 65535 && a->M > 0 && a->N > 0, "gemm_nt: bad shape batch=%d M=%d N=%d",
               a->batch, a->M, a->N);
  SEGK_REQUIRE(a->nterms >= 1 && a->nterms <= SEGK_MAX_SEG, "gemm_nt: nterms=%d", a->nterms);
  const int es = dtype_size(a->dtype), cs = dtype_size(a->c_dtype);
  SEGK_REQUIRE(a->C && a->ldc % 4 == 0 && a->ldc >= ((a->N + 3) & ~3) && ((uintptr_t)a->C % (4 * cs)) == 0 &&
                   (a->c_batch * cs) % (4 * cs) == 0,
               "gemm_nt: C must be aligned to 4 elements with ldc a multiple of 4 and >= N rounded up to 4");
  ConvDev d;
  d.H = 1; d.W = a->M; d.HW = a->M; d.M = a->M;
  d.Cout = a->N; d.nseg = a->nterms; d.n_tiles = 0;
  d.abl = 0;
  bool wide = true;
  for (int s = 0; s < SEGK_MAX_SEG; ++s) {
    const segk_gemm_term& t = a->term[s < a->nterms ? s : 0];
    if (s < a->nterms) {
      SEGK_REQUIRE(t.A && t.B && aligned16(t.A) && aligned16(t.B), "gemm_nt: term %d pointers null or not 16-byte aligned", s);
      SEGK_REQUIRE(t.K > 0 && t.K % 32 == 0 && t.K <= t.lda && t.K <= t.ldb,
                   "gemm_nt: term %d K=%d must be a multiple of 32 and <= lda=%d, ldb=%d", s, t.K, t.lda, t.ldb);
      SEGK_REQUIRE(((long long)t.lda * es) % 16 == 0 && ((long long)t.ldb * es) % 16 == 0 && (t.a_batch * es) % 16 == 0 &&
                       (t.b_batch * es) % 16 == 0,
                   "gemm_nt: term %d strides must be multiples of 16 bytes", s);
      SEGK_REQUIRE((long long)t.ldb * es < 2147483647LL / 256, "gemm_nt: term %d ldb too large", s);
    }
    d.seg[s].in = (const char*)t.A;
    d.seg[s].w = (const char*)t.B;
    d.seg[s].in_ld_b = (long long)t.lda * es;
    d.seg[s].w_ld_b = t.ldb * es;
    d.seg[s].ck_b = t.K * es;
    d.seg[s].w_tap_b = 0;
    d.seg[s].in_batch_b = t.a_batch * es;
    d.seg[s].w_batch_b = t.b_batch * es;
    d.seg[s].R = 1;
    d.seg[s].dil = 1;
    if (s < a->nterms && (d.seg[s].ck_b % 128) != 0) wide = false;
  }
  d.out = (char*)a->C; d.out_ld = a->ldc; d.out_batch_b = a->c_batch * cs;
  d.dil_step = 0; d.stat_batch = 0;
  d.m_begin = 0; d.stat_row0 = 0;
  d.bias = nullptr; d.img_bias = nullptr; d.stat = nullptr;
  d.addend = nullptr; d.addend_ld_b = 0;
  d.addend_bits = nullptr; d.addend_bits_ld = 0;
  const long long big_tiles = (long long)ceil_div(a->M, TileBig::BPX) * (a->N / TileBig::BCO) * a->batch;
  const bool big = wide && a->N % TileBig::BCO == 0 && big_tiles >= 512;
  static const char* gemm_env = experiment("SEGK_GEMM_DMA");
  const bool gemm_dma = !(gemm_env && gemm_env[0] == '0');
  hipStream_t st = (hipStream_t)stream;
  int rc;
#define SEGK_LAUNCH(T, TO)                                                                   \
  (big ? launch_fwd<T, 128, TO, TileBig>(d, st, a->batch)                                    \
       : (wide ? launch_fwd<T, 128, TO, TileSmall>(d, st, a->batch) : launch_fwd<T, 64, TO, TileSmall>(d, st, a->batch)))
#ifdef SEGK_EXPERIMENTS
  if (a->dtype == SEGK_BF16 && big && use_pipe()) {
    rc = a->c_dtype == SEGK_BF16 ? launch_pipe<bf16_t>(d, st, a->batch) : launch_pipe<float>(d, st, a->batch);
  } else
#endif
  if (a->dtype == SEGK_BF16 && big && use_bigdma() && gemm_dma) {
    rc = a->c_dtype == SEGK_BF16 ? launch_dma<bf16_t, 256>(d, st, a->batch) : launch_dma<float, 256>(d, st, a->batch);
  } else if (a->dtype == SEGK_BF16 && wide && !big && use_dma() && gemm_dma) {
    rc = a->c_dtype == SEGK_BF16 ? launch_dma<bf16_t>(d, st, a->batch) : launch_dma<float>(d, st, a->batch);
  } else if (a->dtype == SEGK_BF16) {
    rc = a->c_dtype == SEGK_BF16 ? SEGK_LAUNCH(bf16_t, bf16_t) : SEGK_LAUNCH(bf16_t, float);
  } else {
    rc = a->c_dtype == SEGK_BF16
             ? (big ? launch_fwd<float, 128, bf16_t, TileBig>(d, st, a->batch) : launch_fwd<float, 128, bf16_t, TileSmall>(d, st, a->batch))
             : (big ? launch_fwd<float, 128, float, TileBig>(d, st, a->batch) : launch_fwd<float, 128, float, TileSmall>(d, st, a->batch));
  }
#undef SEGK_LAUNCH
  if (rc != SEGK_OK) return rc;
  SEGK_CHECK_LAUNCH("gemm_nt");
  return SEGK_OK;
}

extern "C" int segk_gemm_tn(const segk_gemm_tn_args* a, segk_stream_t stream) {
  SEGK_REQUIRE(a && a->A && a->B && a->slabs, "gemm_tn: null pointer");
  SEGK_REQUIRE(valid_dtype(a->dtype), "gemm_tn: bad dtype");
  SEGK_REQUIRE(a->batch > 0 && a->batch <= 65535 && a->K > 0 && a->splits >= 1, "gemm_tn: bad shape batch=%d K=%d splits=%d",
               a->batch, a->K, a->splits);
  SEGK_REQUIRE(a->M > 0 && a->M % 32 == 0 && a->M <= a->lda, "gemm_tn: M=%d must be a multiple of 32 and <= lda=%d", a->M, a->lda);
  SEGK_REQUIRE(a->N > 0 && a->N % 16 == 0 && a->N <= a->ldb, "gemm_tn: N=%d must be a multiple of 16 and <= ldb=%d", a->N, a->ldb);
  const int es = dtype_size(a->dtype);
  SEGK_REQUIRE(aligned16(a->A) && aligned16(a->B) && aligned16(a->slabs) && ((long long)a->lda * es) % 16 == 0 &&
                   ((long long)a->ldb * es) % 16 == 0 && (a->a_batch * es) % 16 == 0 && (a->b_batch * es) % 16 == 0,
               "gemm_tn: operands must be 16-byte aligned with 16-byte multiple strides");
  WgradDev d;
  d.H = 1; d.W = a->K; d.HW = a->K; d.M = a->K;
  d.in = (const char*)a->A; d.dout = (const char*)a->B;
  d.in_ld_b = (long long)a->lda * es; d.dout_ld_b = (long long)a->ldb * es;
  d.Cin = a->M; d.Cout = a->N; d.R = 1; d.dil = 1; d.splits = a->splits;
  const int bp = a->dtype == SEGK_BF16 ? WgFrag<bf16_t>::BP : WgFrag<float>::BP;
  long long per = ((long long)a->K + a->splits - 1) / a->splits;
  per = (per + bp - 1) / bp * bp;
  d.per_split = per;
  d.slabs = a->slabs;
  d.in_batch_b = a->a_batch * es; d.dout_batch_b = a->b_batch * es;
  d.groups = 0;
  for (int g = 0; g < SEGK_WGRAD_GROUP_MAX; ++g) d.in_g[g] = d.dout_g[g] = nullptr;
  hipStream_t st = (hipStream_t)stream;
  int rc = a->dtype == SEGK_BF16 ? launch_wgrad<bf16_t>(d, a->batch, st, "gemm_tn") : launch_wgrad<float>(d, a->batch, st, "gemm_tn");
  if (rc != SEGK_OK) return rc;
  SEGK_CHECK_LAUNCH("gemm_tn");
  return SEGK_OK;
}

/*
 * segk.h — C ABI of libsegk.so: MI355X (gfx950) kernels for the DeepLabv3+ segmentation-head
 * training step (ASPP, decoder, synchronized batch norm, cross-entropy, confusion matrix).
 *
 * Conventions
 *   - Every entry point returns SEGK_OK (0) or a negative segk_status; segk_last_error() gives the
 *     message of the calling thread's last failure.  No C++ exception crosses this boundary.
 *   - All pointers are DEVICE pointers borrowed from the caller (e.g. torch's allocator) unless a
 *     parameter is documented as host.  The library allocates no device memory and keeps no pointer
 *     after a call returns.  Kernels are enqueued on `stream` (a hipStream_t passed as void*; NULL =
 *     the default stream) and the call returns without synchronising.
 *   - Activations inside the head are "pixel-major": a [M][ld] matrix with M = N*H*W pixels, the
 *     channels of one pixel contiguous, `ld` >= channel count (elements) so that a tensor may be a
 *     channel slice of a wider concat buffer.  dtype is SEGK_F32 or SEGK_BF16.
 *   - Contracted channel counts of the convolution entry points must be multiples of 32; callers pad
 *     buffers with zero channels (weights are padded by segk_pack_weight).
 *
 * Each declaration cites the reference code (paths relative to the reference tree) that it replaces.
 */
#ifndef SEGK_H
#define SEGK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  SEGK_OK = 0,
  SEGK_INVALID_ARG = -1,
  SEGK_UNSUPPORTED_SHAPE = -2,
  SEGK_WORKSPACE_TOO_SMALL = -3,
  SEGK_HIP_ERROR = -4,
  SEGK_RCCL_ERROR = -5
} segk_status;

typedef enum { SEGK_F32 = 0, SEGK_BF16 = 1 } segk_dtype;

typedef void* segk_stream_t; /* hipStream_t */

#define SEGK_MAX_SEG 4

/* ---- library ------------------------------------------------------------------------------- */
int segk_version(void);
const char* segk_last_error(void);
/* Number of visible HIP devices, or a negative status; touches no device memory. */
int segk_device_count(void);

/* ---- layout changes at the boundary --------------------------------------------------------- */
/* NCHW fp32 (the feeder's tensors, modeling/deeplab.py:249-250) -> pixel-major [N*H*W][dst_ld]. */
int segk_nchw_to_pm(const float* src, void* dst, int dst_dtype, int N, int C, int H, int W, int dst_ld,
                    segk_stream_t stream);
/* pixel-major -> NCHW fp32 (gradients handed back to the feeder's autograd). */
int segk_pm_to_nchw(const void* src, int src_dtype, int src_ld, float* dst, int N, int C, int H, int W,
                    segk_stream_t stream);

/* ---- convolution as implicit GEMM ----------------------------------------------------------- */
/* nn.Conv2d weights [Cout][Cin][R][R] fp32 (modeling/aspp.py:10-11, modeling/decoder.py:132,142-153)
 * -> operand layout of segk_conv_fwd.
 *   transpose = 0: dst[tap][rows = Cout][Ck = Cin rounded up to 32]           (forward)
 *   transpose = 1: dst[R*R-1-tap][rows = Cin][Ck = Cout rounded up to 32]     (data gradient)
 * Padding columns are written as zeros. */
int segk_pack_weight(const float* w_oihw, void* dst, int dst_dtype, int Cout, int Cin, int R, int transpose,
                     segk_stream_t stream);
/* The same from a weight of either dtype in any dense layout, given by the element strides of its output-channel,
 * input-channel and tap (r*R + s) indices: contiguous OIHW is (Cin*R*R, R*R, 1), a channels_last nn.Conv2d weight
 * (the feeder's bottleneck convolutions, newmodeling/backbone/resnet101_2.py:19-21, under a channels_last model) is
 * (R*R*Cin, 1, Cin). */
int segk_pack_weight_strided(const void* w, int w_dtype, long long s_co, long long s_ci, long long s_tap, void* dst,
                             int dst_dtype, int Cout, int Cin, int R, int transpose, segk_stream_t stream);

/* Many weights in one launch (every convolution of the feeder bottlenecks, newmodeling/backbone/resnet101_2.py:31-51,
 * and of the head re-packs its weight once per optimizer step, train.py:143: one launch instead of ~100 small ones).
 * `items` is a table of n rows in DEVICE memory; first_block is the running sum of segk_pack_weight_blocks() over
 * the rows before, total_blocks the sum over all rows.  Each row is packed exactly as segk_pack_weight_strided would. */
typedef struct segk_pack_item {
  const void* w;     /* the weight, as segk_pack_weight_strided */
  void* dst;         /* the packed operand */
  long long s_co, s_ci, s_tap;
  int w_dtype, dst_dtype;
  int Cout, Cin, R, transpose;
  int first_block;
  int reserved;
} segk_pack_item;
/* workgroups one row needs (host-side arithmetic, no launch); negative status on a bad shape */
int segk_pack_weight_blocks(int Cout, int Cin, int R, long long s_tap, int transpose);
int segk_pack_weights(const segk_pack_item* items, int n, int total_blocks, segk_stream_t stream);

typedef struct {
  const void* in;  /* [M][in_ld] activations, args.dtype */
  const void* w;   /* packed weights [R*R][rows = Cout][Ck], args.dtype */
  int in_ld;
  int Ck;          /* contracted channels, multiple of 32, <= in_ld */
  int R;           /* 1 or 3 (square kernel, stride 1, padding = dil*(R-1)/2: output size = input size) */
  int dil;
  /* Elements between the rows (output channels) and between the taps of `w`; 0 = the packed layout above
   * (Ck and Cout*Ck).  A channels_last nn.Conv2d weight [Cout][R*R][Cin] of the operand dtype is used in place with
   * (R*R*Cin, Cin) when Cin is a multiple of 32: no packing pass for the forward convolution. */
  long long w_row_stride;
  long long w_tap_stride;
} segk_conv_seg;

typedef struct {
  int N, H, W;
  int dtype;        /* of in / w */
  int Cout;         /* output channels (rows of every segment's packed weight) */
  int nseg;         /* 1..SEGK_MAX_SEG; the segments' products are summed into one output */
  segk_conv_seg seg[SEGK_MAX_SEG];
  void* out;        /* [M][out_ld] */
  int out_ld;
  int out_dtype;
  const float* bias;      /* [Cout] or NULL (modeling/decoder.py:152-153, classifier) */
  const float* img_bias;  /* [N][Cout] or NULL: added to every pixel of image n */
  double* stat_partials;  /* NULL, or [batch][segk_conv_stat_rows(&a)][2][Cout] fp64: per-row-tile sum and
                             sum of squares of the UNBIASED output, for batch-norm statistics */
  /* Several convolutions of the SAME input in one launch (the dilated ASPP branches, modeling/aspp.py:132-135):
   * entry b uses weights w + b*w_batch (elements), dilation dil + b*dil_step and writes out + b*out_batch
   * (elements; e.g. the next column slice of a concat buffer).  batch <= 1: a single convolution; biases must
   * be NULL when batch > 1. */
  int batch;
  long long w_batch;
  long long out_batch;
  int dil_step;
  /* NULL, or [M][addend_ld] of out_dtype: out = round(conv) + addend, the conv result rounded to out_dtype first
   * (bit for bit what a separate addition of the two tensors gives).  It stands for the addition autograd performs
   * where a bottleneck's input receives two gradients - from conv1's data gradient and from the shortcut
   * (newmodeling/backbone/resnet101_2.py:31-51, `out += residual`).  Needs bf16 operands with 128-byte K rows, Cout
   * a multiple of 128, no biases, batch <= 1; SEGK_UNSUPPORTED_SHAPE otherwise. */
  const void* addend;
  int addend_ld;
  /* NULL, or [M][addend_bits_ld] bytes beside the addend: bit c % 8 of byte c / 8 set = addend[m][c] is added, clear =
   * zero is added.  It is the ReLU pattern segk_bn_apply stored for the block's last unit: the shortcut's gradient is
   * the block's incoming gradient where the ReLU was active, so the consumer masks that gradient itself and the
   * batch norm's backward need not write a masked copy (segk_bn_bwd_apply with dres = NULL). */
  const void* addend_bits;
  int addend_bits_ld;
  /* Backward statistics of a batch norm from the launch that produces its output gradient.  When this launch is the
   * data gradient of the convolution that consumed relu(bn(x)) (newmodeling/backbone/resnet101_2.py:33-45: conv2 after
   * bn1 + relu, conv3 after bn2 + relu), its result `out` is dz of that batch norm, and the sums segk_bn_bwd_stats would read dz and x again for leave with the epilogue instead:
   * bst_rows [segk_conv_bstat_rows()][2][Cout] receives per-channel partial sums of g and g * xhat, g = out where the
   * layer's ReLU was active (bst_relu != 0) and zero elsewhere, xhat = (x - mean) * inv_std.  bst_x: the layer's input
   * [M][bst_x_ld] (bf16, like out); bst_mean .. bst_shift: its coefficient vectors [Cout] (16-byte aligned); bst_bits:
   * NULL (the ReLU pattern is x * scale + shift > 0) or the pattern segk_bn_apply stored, [M][bst_bits_ld] bytes.
   * segk_bn_reduce turns the rows into the sums segk_bn_bwd_apply takes (sync_batchnorm/batchnorm.py:113-125 as
   * autograd differentiates it).  bst_rows == NULL: nothing of this is read. */
  double* bst_rows;
  const void* bst_x;
  int bst_x_ld;
  const float* bst_mean;
  const float* bst_inv_std;
  const float* bst_scale;
  const float* bst_shift;
  const void* bst_bits;
  int bst_bits_ld;
  int bst_relu;
} segk_conv_args;

/* nn.Conv2d forward (modeling/aspp.py:18, modeling/decoder.py:172,178) and, with transposed packed
 * weights, its data gradient.  out[m][co] = sum_seg sum_tap sum_c in[m + tap][c] * w[tap][co][c]. */
int segk_conv_fwd(const segk_conv_args* a, segk_stream_t stream);
/* Number of partial rows segk_conv_fwd writes to stat_partials for this problem. */
int segk_conv_stat_rows(const segk_conv_args* a);
/* Number of rows segk_conv_fwd writes to bst_rows for this problem (every bst_* field but bst_rows filled in), or 0
 * when this problem's launch cannot emit them (segk_conv_fwd then refuses a non-NULL bst_rows with
 * SEGK_UNSUPPORTED_SHAPE and the caller runs segk_bn_bwd_stats). */
int segk_conv_bstat_rows(const segk_conv_args* a);

typedef struct {
  int N, H, W;
  int dtype;       /* of in / dout */
  const void* in;  /* [M][in_ld] forward input */
  int in_ld;
  int Cin;         /* multiple of 32 (padded channels included) */
  const void* dout;/* [M][dout_ld] gradient w.r.t. the conv output */
  int dout_ld;
  int Cout;        /* multiple of 16 */
  int R, dil;
  int splits;      /* >= 1: the pixel range is cut into `splits` parts summed later in fixed order */
  float* slabs;    /* [splits][R*R][Cin][Cout] fp32 partial weight gradients */
  int tile;        /* 0: the library chooses the workgroup tile; 128 / 256: that square tile (256 needs Cout % 256 == 0
                      and Cin >= 256); 384: the 256(co) x 128(ci) ring tile (bf16 only; what 0 selects for bf16 when
                      Cout % 256 == 0).  Callers that size `splits` for a given tile pass it here. */
} segk_wgrad_args;

/* Weight gradient of nn.Conv2d (autograd of modeling/aspp.py:18, modeling/decoder.py:172,178). */
int segk_conv_wgrad(const segk_wgrad_args* a, segk_stream_t stream);
/* Sum the split slabs in index order and write nn.Conv2d layout:
 * dw[co][ci][r][s] = sum_split slabs[split][tap][ci][co], co < Cout_true, ci < Cin_true. */
int segk_wgrad_finish(const float* slabs, int splits, int R, int Cin, int Cout, float* dw_oihw, int Cout_true,
                      int Cin_true, segk_stream_t stream);
/* The same into a gradient tensor of either dtype laid out like its weight (strides as segk_pack_weight_strided):
 * autograd wants the gradient of a bf16 channels_last parameter in that dtype and layout. */
int segk_wgrad_finish_strided(const float* slabs, int splits, int R, int Cin, int Cout, void* dw, int dw_dtype,
                              long long s_co, long long s_ci, long long s_tap, int Cout_true, int Cin_true,
                              segk_stream_t stream);

/* Weight gradients of several convolutions of one shape in one launch.  A ResNet stage repeats one bottleneck many
 * times (newmodeling/backbone/resnet101_2.py:100-122, `for i in range(1, blocks)`: 22 identical blocks in layer 3), and
 * on the 33 x 33 maps of that stage one weight gradient alone has too few tiles for the chip: it has to be cut into many
 * pixel parts, each of which writes a whole fp32 copy of the gradient.  Together the same launch is filled with a
 * fraction of the parts per convolution.  conv.in / conv.dout are ignored; conv.slabs holds
 * [groups][splits][R*R][Cin][Cout] floats.  bf16 operands and Cout a multiple of 256 (the ring tile) only:
 * SEGK_UNSUPPORTED_SHAPE otherwise.  The values equal segk_conv_wgrad's with the same `splits`. */
#define SEGK_WGRAD_GROUP_MAX 8
typedef struct {
  segk_wgrad_args conv;
  int groups; /* 1 .. SEGK_WGRAD_GROUP_MAX */
  const void* in[SEGK_WGRAD_GROUP_MAX];
  const void* dout[SEGK_WGRAD_GROUP_MAX];
} segk_wgrad_group_args;
int segk_conv_wgrad_group(const segk_wgrad_group_args* a, segk_stream_t stream);
/* segk_wgrad_finish_strided for the slabs of segk_conv_wgrad_group: dw[g] receives the gradient of convolution g;
 * all share dtype, strides and true channel counts. */
int segk_wgrad_finish_group(const float* slabs, int groups, int splits, int R, int Cin, int Cout, void* const* dw,
                            int dw_dtype, long long s_co, long long s_ci, long long s_tap, int Cout_true,
                            int Cin_true, segk_stream_t stream);

/* ---- batched matrix products (attention: modeling/damm_cam_pam.py) --------------------------- */
typedef struct {
  const void* A;     /* [batch][M][lda] */
  long long a_batch; /* elements between batch entries (0 = shared) */
  int lda;
  const void* B;     /* [batch][N][ldb] */
  long long b_batch;
  int ldb;
  int K;             /* contracted length, multiple of 32; both operands hold zeros beyond their true length */
} segk_gemm_term;

typedef struct {
  int batch, M, N;
  int dtype;         /* of every A and B */
  int nterms;        /* 1..SEGK_MAX_SEG products summed into C */
  segk_gemm_term term[SEGK_MAX_SEG];
  void* C;           /* [batch][M][ldc], ldc a multiple of 4 and >= N rounded up to 4 (the rounding columns get zeros) */
  long long c_batch;
  int ldc;
  int c_dtype;
} segk_gemm_args;

/* C[b][m][n] = sum_t sum_k A_t[b][m][k] * B_t[b][n][k], fp32 accumulation: torch.bmm with the second factor
 * stored row-major over its contracted index (modeling/damm_cam_pam.py:38,43,83,90 and their gradients). */
int segk_gemm_nt(const segk_gemm_args* a, segk_stream_t stream);

typedef struct {
  int batch, M, N, K;
  int dtype;          /* of A and B */
  const void* A;      /* [batch][K][lda], M <= lda, M a multiple of 32 (padding columns zero) */
  long long a_batch;
  int lda;
  const void* B;      /* [batch][K][ldb], N <= ldb, N a multiple of 16 */
  long long b_batch;
  int ldb;
  int splits;         /* >= 1 parts of the K range, summed later in index order */
  float* slabs;       /* [batch][splits][M][N] fp32 */
} segk_gemm_tn_args;

/* slabs[b][s][m][n] = sum over the s-th part of k of A[b][k][m] * B[b][k][n]: products contracted over the
 * slow index of both factors (autograd of the torch.bmm calls in modeling/damm_cam_pam.py:38,43,83,90). */
int segk_gemm_tn(const segk_gemm_tn_args* a, segk_stream_t stream);
/* C[b][m][n] = sum_s slabs[b][s][m][n] for m < rows, n < cols (rows <= M, cols <= N); C is [batch][rows][ldc]. */
int segk_gemm_tn_finish(const float* slabs, int batch, int splits, int M, int N, void* C, int c_dtype,
                        long long c_batch, int ldc, int rows, int cols, segk_stream_t stream);

/* ---- attention helpers (modeling/damm_cam_pam.py) --------------------------------------------- */
/* Row softmax of fp32 energies e[rows][ld_e] over `cols` columns (nn.Softmax(dim=-1), damm_cam_pam.py:39).
 * reversed != 0: softmax of (row maximum - e) (damm_cam_pam.py:84-87).  Output columns cols..out_cols are
 * written as zeros so the result can be contracted over a padded length. */
int segk_softmax_rows(const float* e, long long rows, int cols, int ld_e, int reversed, void* att, int att_dtype,
                      int ld_att, int out_cols, segk_stream_t stream);
/* Gradient of the row softmax w.r.t. its input: de[r][j] = sign * att[r][j] * (datt[r][j] - sum_j' att*datt);
 * sign = -1 for the reversed form.  Output columns cols..out_cols are zeros. */
int segk_softmax_rows_bwd(const void* att, int att_dtype, int ld_att, const float* datt, int ld_datt, long long rows,
                          int cols, float sign, void* de, int de_dtype, int ld_de, int out_cols,
                          segk_stream_t stream);
/* dst[b][c][r] = src[b][r][c] for r < R, c < Cc; dst columns R..dst_cols are zeros
 * (the .view/.permute pairs of damm_cam_pam.py:33-35,41,81-82 on pixel-major data). */
int segk_transpose(const void* src, int dtype, int batch, int R, int Cc, long long src_batch, int src_ld, void* dst,
                   long long dst_batch, int dst_ld, int dst_cols, segk_stream_t stream);
/* out[m][c] = a0*x0 + a1*x1 + a2*x2 with a_i = s_i * (*dev_i) (dev_i NULL = 1): gamma*out + x of
 * damm_cam_pam.py:46,93 and x1 + x2 of :133 in one pass.  x1/x2 may be NULL.  All operands [M][ld_i], dtype. */
int segk_lincomb(void* out, int out_dtype, int out_ld, long long M, int C, const void* x0, int ld0, float s0,
                 const float* dev0, const void* x1, int ld1, float s1, const float* dev1, const void* x2, int ld2,
                 float s2, const float* dev2, int dtype, segk_stream_t stream);
/* result[0] = sum over [M][C] of a*b in fp64 order-fixed (gradient of the gamma scalars, damm_cam_pam.py:46,93);
 * partials: segk_dot_partial_rows(M) doubles. */
int segk_dot(const void* a, int lda, const void* b, int ldb, int dtype, long long M, int C, double* partials,
             float* result, segk_stream_t stream);
int segk_dot_partial_rows(long long M);

/* ---- batch normalisation (synchronized) ----------------------------------------------------- */
/* Layouts for the BN entry points. */
typedef enum { SEGK_PM = 0 /* [M][ld], channels contiguous */, SEGK_NCHW = 1 /* [N][C][HW] */ } segk_layout;

/* Per-channel sum and sum of squares (modeling/sync_batchnorm/batchnorm.py:60-62), accumulated in fp64
 * (the square of an fp32 value is exact in fp64, so var = E[x^2] - E[x]^2 does not cancel catastrophically
 * even for the image-pooling branch, whose few samples per channel are nearly equal).
 * partials: [segk_bn_stat_rows(...)][2][C] fp64. */
int segk_bn_stats(const void* x, int dtype, int layout, int N, int C, int HW, int ld, double* partials,
                  segk_stream_t stream);
int segk_bn_stat_rows(int layout, int N, int C, int HW);
/* Sum `rows` partial rows in index order: sums[0..C) = first sum, sums[C..2C) = second sum, sums[2C] = count.
 * The [2C+1] fp64 vector is what ranks all-reduce (replaces ReduceAddCoalesced, batchnorm.py:101-102).
 * sums_f32 (NULL, or [2C] fp32) receives the same two sums rounded to fp32: in the backward pass they are this
 * rank's d(beta) and d(gamma). */
int segk_bn_reduce(const double* partials, int rows, int C, double count, double* sums, float* sums_f32,
                   segk_stream_t stream);
/* Global sums -> mean, inv_std, running statistics and the affine coefficients
 * scale = gamma*inv_std, shift = beta - mean*scale (batchnorm.py:113-125 and :73).
 * legacy_clamp != 0: inv_std = clamp(var, eps)^-1/2 (batchnorm.py:125); 0: (var + eps)^-1/2 (F.batch_norm,
 * batchnorm.py:50-53).  running_* may be NULL.  gamma/beta NULL = 1/0. */
int segk_bn_coeffs(const double* sums, const float* gamma, const float* beta, float eps, float momentum,
                   int legacy_clamp, float* running_mean, float* running_var, float* mean, float* inv_std,
                   float* scale, float* shift, int C, segk_stream_t stream);
/* segk_bn_reduce followed by segk_bn_coeffs in one launch, for statistics that are not exchanged between ranks
 * (a single rank, or a plain nn.BatchNorm2d inside the head: batchnorm.py:50-53).  sums ([2C+1] fp64) may be NULL. */
int segk_bn_reduce_coeffs(const double* partials, int rows, int C, double count, const float* gamma,
                          const float* beta, float eps, float momentum, int legacy_clamp, float* running_mean,
                          float* running_var, float* mean, float* inv_std, float* scale, float* shift, double* sums,
                          segk_stream_t stream);
/* Evaluation mode (batchnorm.py:50-53 with training False): coefficients from running statistics. */
int segk_bn_coeffs_eval(const float* running_mean, const float* running_var, const float* gamma,
                        const float* beta, float eps, float* mean, float* inv_std, float* scale, float* shift,
                        int C, segk_stream_t stream);
/* y = x*scale + shift (+ res), optional ReLU, optional dropout mask (mask[m][c] in {0,1} bytes, laid out like y
 * with leading dimension C; the kept values are multiplied by mask_scale = 1/(1-p)).
 * (batchnorm.py:73, modeling/aspp.py:21,147, modeling/decoder.py:145-151)
 * res (NULL, or pixel-major [M][res_ld] of y's dtype) is added before the ReLU: the shortcut of a bottleneck block
 * (modeling/backbone/resnet101.py Bottleneck.forward: bn3 -> += residual -> relu) in the same pass.
 * relu_bits (NULL, or [M][relu_bits_ld] bytes, relu_bits_ld >= C / 8; pixel-major layout with C, the leading
 * dimensions and the operand addresses multiples of 8 elements / 16 bytes) receives the ReLU pattern of the stored
 * output: bit j of byte [m][c / 8] is set when y[m][8 * (c / 8) + j] > 0.  The backward passes read it in place of
 * the stored output (1/16 of its bytes). */
int segk_bn_apply(const void* x, int x_dtype, int x_ld, void* y, int y_dtype, int y_ld, int layout, int N, int C,
                  int HW, const float* scale, const float* shift, int relu, const uint8_t* mask,
                  float mask_scale, const void* res, int res_ld, uint8_t* relu_bits, int relu_bits_ld,
                  segk_stream_t stream);
/* Backward, pass 1: with g = dz * mask * [relu active], per-channel partial sums of g and g*xhat.
 * partials: [segk_bn_stat_rows(...)][2][C].  act (NULL, or pixel-major [M][act_ld] of dz's dtype) is the layer's
 * stored output when a residual was added before the ReLU; the ReLU pattern is then act > 0 instead of
 * x*scale + shift > 0.  act_bits != 0: act points to the bit pattern segk_bn_apply wrote ([M][act_ld] bytes). */
int segk_bn_bwd_stats(const void* dz, int dz_dtype, int dz_ld, const void* x, int x_dtype, int x_ld, int layout,
                      int N, int C, int HW, const float* mean, const float* inv_std, const float* scale,
                      const float* shift, int relu, const uint8_t* mask, float mask_scale, const void* act,
                      int act_ld, int act_bits, double* partials, segk_stream_t stream);
/* Backward, pass 2 (act / act_bits as in pass 1; dres, NULL or [M][dres_ld] of dz's dtype, receives g = the gradient
 * of the residual input; NULL with act given: the caller derives it from dz and the bit pattern itself, see
 * segk_conv_args.addend_bits): dx = scale * (g - sum_g/count - xhat * sum_gxhat/count) with the GLOBAL sums
 * gsums = [sum g (C), sum g*xhat (C), count] (autograd of batchnorm.py:66-73 through the reduction);
 * eval != 0: dx = scale * g.  */
int segk_bn_bwd_apply(const void* dz, int dz_dtype, int dz_ld, const void* x, int x_dtype, int x_ld, void* dx,
                      int dx_dtype, int dx_ld, int layout, int N, int C, int HW, const float* mean,
                      const float* inv_std, const float* scale, const float* shift, int relu,
                      const uint8_t* mask, float mask_scale, const double* gsums, int eval, const void* act,
                      int act_ld, int act_bits, void* dres, int dres_ld, segk_stream_t stream);

/* The keep-mask of nn.Dropout in training mode (modeling/aspp.py:128, modeling/decoder.py:146,151): mask[i] = 1 with
 * probability 1 - p, else 0, one byte per element, the form segk_bn_apply's `mask` takes.  seed: two 64-bit words in
 * DEVICE memory (key, stream offset) - the caller draws them from its own generator, inside a recorded graph as well,
 * so that every replay gets a fresh mask; the same words give the same mask.  Philox4x32-10, 32 random bits per
 * element.  0 <= p < 1; mask 16-byte aligned. */
int segk_dropout_mask(uint8_t* mask, long long n, float p, const long long* seed, segk_stream_t stream);

/* ---- pooling / resampling ------------------------------------------------------------------- */
/* nn.AdaptiveAvgPool2d((1,1)) on a pixel-major tensor (modeling/aspp.py:121): out[n][c] fp32 =
 * scale * sum_pixels x[n][p][c]  (scale = 1/HW forward; 1 to sum a gradient). */
int segk_pool_sum(const void* x, int dtype, int ld, int N, int HW, int C, float scale, float* out,
                  segk_stream_t stream);
/* nn.MaxPool2d(kernel_size=3, stride=2, padding=1) of the ResNet stem (newmodeling/backbone/resnet101_2.py:79,149) on a
 * pixel-major tensor [N*Hi*Wi][x_ld] -> [N*Ho*Wo][y_ld], Ho = (Hi - 1) / 2 + 1.  idx ([N*Ho*Wo][C] bytes) receives the
 * position of the maximum inside its window (0..8; the first maximum in scan order, a NaN is a maximum: ATen's rule);
 * the backward pass gathers with it: dx[p] = sum of dy over the (at most four) windows whose maximum is p, added in
 * ascending window order in fp32.  C a multiple of 8, 16-byte aligned operands, leading dimensions multiples of 8:
 * SEGK_UNSUPPORTED_SHAPE otherwise. */
int segk_maxpool3x3s2_fwd(const void* x, int dtype, int x_ld, int N, int C, int Hi, int Wi, void* y, int y_ld,
                          uint8_t* idx, segk_stream_t stream);
int segk_maxpool3x3s2_bwd(const void* dy, int dtype, int dy_ld, const uint8_t* idx, int N, int C, int Hi, int Wi,
                          void* dx, int dx_ld, segk_stream_t stream);
/* F.interpolate(mode='bilinear', align_corners=True) (modeling/aspp.py:137, modeling/decoder.py:176,
 * modeling/deeplab.py:252).  Source is pixel-major [N*Hi*Wi][in_ld]; the destination is pixel-major
 * (out_layout = SEGK_PM, [N*Ho*Wo][out_ld], out_dtype) or NCHW fp32 (out_layout = SEGK_NCHW). */
int segk_upsample_fwd(const void* in, int in_dtype, int in_ld, int N, int C, int Hi, int Wi, void* out,
                      int out_dtype, int out_layout, int out_ld, int Ho, int Wo, segk_stream_t stream);
/* Its gradient, gather form (no atomics): din[n][iy][ix][c] = sum over the output pixels that read it.
 * dout is pixel-major or NCHW fp32 as above; din is pixel-major [N*Hi*Wi][din_ld]; channels
 * C..din_cpad of din are written as zero. */
int segk_upsample_bwd(const void* dout, int dout_dtype, int dout_layout, int dout_ld, int N, int C, int Ho,
                      int Wo, void* din, int din_dtype, int din_ld, int din_cpad, int Hi, int Wi,
                      float* workspace, segk_stream_t stream);
/* fp32 elements of `workspace` that let the NCHW form run as two separable passes (rows, then columns) instead
 * of one gather kernel; 0 for the pixel-major form.  workspace may be NULL (the gather kernel is used). */
long long segk_upsample_bwd_workspace(int dout_layout, int N, int C, int Ho, int Wi);

/* ---- loss ----------------------------------------------------------------------------------- */
/* nn.CrossEntropyLoss(weight, ignore_index) on NCHW fp32 logits and fp32 label maps (utils/loss.py:23-35).
 * Writes lse[N*H*W] (log-sum-exp per pixel) and sums[2] = {sum w*nll, sum w} over non-ignored pixels
 * (fp32, reduced in fixed order through `partials`, which must hold segk_ce_partial_rows(P)*2 floats). */
int segk_ce_fwd(const float* logits, const float* target, const float* weight, int N, int K, int HW,
                int ignore_index, float* lse, float* partials, float* sums, segk_stream_t stream);
int segk_ce_partial_rows(long long pixels);
/* dlogits[n][k][p] = coef * w_t * (softmax_k - [k == t]) for non-ignored pixels, 0 otherwise, where
 * coef = *coef_dev (device scalar: grad_out / (sum w * batch)). */
int segk_ce_bwd(const float* logits, const float* target, const float* weight, const float* lse, int N, int K,
                int HW, int ignore_index, const float* coef_dev, float* dlogits, segk_stream_t stream);

/* segk_ce_fwd for callers that cannot allocate device memory or carry a workspace (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference.
 * The partial rows live in a fixed 64 KiB device array inside the library (one copy per GPU, held until the
 * process ends); the call allocates nothing.  Same results as segk_ce_fwd, bit for bit.  At most
 * segk_ce_fwd_static_max_pixels() = 2^25 pixels (N*HW); more returns SEGK_UNSUPPORTED_SHAPE.  All calls on one
 * GPU share that array: calls that overlap on one GPU (two streams, two threads) race; calls ordered on one
 * stream do not.  The device the stream belongs to must be the current device (the array's address is looked up
 * for the current device); a stream of another device returns SEGK_INVALID_ARG. */
int segk_ce_fwd_static(const float* logits, const float* target, const float* weight, int N, int K, int HW,
                       int ignore_index, float* lse, float* sums, segk_stream_t stream);
long long segk_ce_fwd_static_max_pixels(void);

/* Final upsample (modeling/deeplab.py:252) fused with the cross entropy (utils/loss.py:23-35): the loss of
 * F.interpolate(low, (Ho,Wo), bilinear, align_corners=True) without writing the full-resolution logits.
 * low: pixel-major fp32 [N*Hi*Wi][low_ld], K classes; target fp32 [N][Ho][Wo]; lse, partials, sums as segk_ce_fwd
 * (partials: segk_ce_partial_rows(N*Ho*Wo) rows of 2 floats). */
int segk_upsample_ce_fwd(const float* low, int low_ld, int N, int K, int Hi, int Wi, const float* target,
                         const float* weight, int Ho, int Wo, int ignore_index, float* lse, float* partials,
                         float* sums, segk_stream_t stream);
/* Its gradient with respect to low: dlow pixel-major fp32 [N*Hi*Wi][dlow_ld], channels K..dlow_cpad zero-filled.
 * workspace: segk_upsample_bwd_workspace(SEGK_NCHW, N, K, Ho, Wi) floats; coef_dev as segk_ce_bwd. */
int segk_upsample_ce_bwd(const float* low, int low_ld, int N, int K, int Hi, int Wi, const float* target,
                         const float* weight, const float* lse, int Ho, int Wo, int ignore_index,
                         const float* coef_dev, float* workspace, float* dlow, int dlow_ld, int dlow_cpad,
                         segk_stream_t stream);

/* np.argmax(F.interpolate(low, (Ho,Wo), bilinear, align_corners=True), axis=1) (train.py:215-217) without the
 * full-resolution logits; same blend arithmetic as segk_upsample_fwd's NCHW output, so the predictions equal
 * segk_argmax of that output bit for bit.  pred: int64 [N][Ho][Wo]. */
int segk_upsample_argmax(const float* low, int low_ld, int N, int K, int Hi, int Wi, int Ho, int Wo, int64_t* pred,
                         segk_stream_t stream);

/* ---- evaluator ------------------------------------------------------------------------------ */
/* np.argmax(logits, axis=1) (train.py:217): first maximal index, NaN counts as maximal. */
int segk_argmax(const float* logits, int N, int K, int HW, int64_t* pred, segk_stream_t stream);
/* Evaluator._generate_matrix + add_batch (utils/metrics.py:44-54): for pixels with 0 <= gt < K,
 * cm[K*int(gt) + pred] += 1.  cm is int64 [K*K + 1]; the extra slot counts pixels whose prediction lies
 * outside [0,K) (np.bincount would fail on those).  gt is fp32 (gt_is_i64 = 0) or int64. */
int segk_confmat(const void* gt, int gt_is_i64, const int64_t* pred, long long count, int K, int64_t* cm,
                 segk_stream_t stream);

/* ---- optimizer ------------------------------------------------------------------------------ */
/* torch.optim.SGD(momentum, weight_decay, dampening 0, nesterov off) as the trainers build it (train.py:56-61),
 * over many tensors in one launch.  The caller cuts each tensor into chunks (any size; 64 Ki elements is a good one)
 * and passes a device table with one entry per chunk:
 *   d = grad + weight_decay * master;  momentum_buf = momentum * momentum_buf + d;  master -= lr * momentum_buf
 * (a zero-initialised buffer reproduces torch's first step, which copies d).  dtype: low byte = type of param
 * (for SEGK_F32, master must be the parameter itself; for SEGK_BF16, master is its fp32 copy and param receives
 * the rounded result), second byte = type of grad (an fp32 parameter takes an fp32 gradient; a bf16 parameter
 * takes its own bf16 gradient or an fp32 one, e.g. a slice of an all-reduced bucket). */
typedef struct {
  void* param;
  const void* grad;
  float* momentum;
  float* master;
  int n;
  int dtype;
} segk_sgd_chunk;
/* dst[i] = (float)src[i] for every chunk of a device table: packs the gradients of many parameters (fp32 or
 * bf16) into flat fp32 buckets, the operand of the gradient all-reduce that replaces DataParallel's reduce-add
 * (train.py:106-109). */
typedef struct {
  const void* src;
  float* dst;
  int n;
  int dtype;
} segk_gather_chunk;
int segk_gather_f32(const segk_gather_chunk* table, int n_chunks, segk_stream_t stream);
int segk_sgd_step(const segk_sgd_chunk* table, int n_chunks, float lr, float momentum, float weight_decay,
                  segk_stream_t stream);
/* The same update with {lr, momentum, weight_decay} read from three consecutive floats in device memory, for a
 * launch recorded in a HIP graph: the poly schedule of utils/lr_scheduler.py:50-73 writes a new learning rate into
 * param_groups before every iteration (train.py:143), and a recorded launch keeps its by-value arguments. */
int segk_sgd_step_dev(const segk_sgd_chunk* table, int n_chunks, const float* hyper, segk_stream_t stream);

/* ---- exchange step (one process per GPU) ------------------------------------------------------ */
/* A process-wide RCCL communicator for hosts without a collective library of their own; it stands where the
 * reference has SyncMaster / SlavePipe (modeling/sync_batchnorm/comm.py:18-129) and DataParallel's gradient
 * reduce-add (train.py:106-109).  Rank 0 draws an id with segk_comm_unique_id and hands the 128 bytes to the other
 * processes by any means; every rank then calls segk_comm_init with the GPU it will use made current.  The
 * all-reduces are in-place SUMs, asynchronous on the given stream: [2C+1] fp64 batch-norm statistics, the three
 * fp32 loss sums, fp32 gradient buckets, the int64 confusion matrix.  librccl is opened on first use.  The Python
 * package reaches RCCL through torch.distributed instead (same library underneath). */
typedef struct { char bytes[128]; } segk_comm_id;
int segk_comm_unique_id(segk_comm_id* id);
int segk_comm_init(const segk_comm_id* id, int rank, int world);
int segk_comm_destroy(void);
int segk_allreduce_f32(float* buf, long long count, segk_stream_t stream);
int segk_allreduce_f64(double* buf, long long count, segk_stream_t stream);
int segk_allreduce_i64(int64_t* buf, long long count, segk_stream_t stream);

/* One-shot all-reduce (in-place SUM) through peer-mapped mailboxes: the latency path for the 226 small exchanges
 * per training step of the synchronised batch norm - the [sum, sum of squares, count] vector of every layer forward
 * and [sum g, sum g*xhat] backward - which the reference gathers on replica 0 through Python queues, reduces with
 * ReduceAddCoalesced and broadcasts back (modeling/sync_batchnorm/batchnorm.py:90-111, comm.py:97-125).  One
 * single-workgroup kernel per rank stores the vector into every rank's mailbox, waits for the others' and adds
 * them in rank order, so every rank holds bit-identical sums; no host call, no library collective, and the launch
 * can be recorded in a HIP graph (its sequence number lives in device memory).
 *   segk_mailbox_create   allocates this process's mailbox on the current device (fine-grained device memory owned
 *                         by the library until segk_mailbox_destroy) and writes its IPC handle
 *                         (SEGK_MAILBOX_HANDLE_BYTES) to handle_out;
 *   segk_mailbox_connect  takes the `world` handles in rank order (exchanged by the host by any means), maps the
 *                         peers' mailboxes; timeout_ms bounds every in-kernel wait (<= 0: 60 000);
 *   segk_mailbox_set_timeout  the same bound for the launches issued from now on (a short one for a connection test
 *                         between ranks that have just met at a host barrier, a long one for training, where the
 *                         first step's algorithm search can hold a rank back for seconds);
 *   segk_mailbox_allreduce_f64 / _f32: buf[0..n) on the device, any n (vectors longer than SEGK_MAILBOX_MAX_DOUBLES
 *                         go through in chunks inside the one launch; sums are formed in fp64 and rounded once);
 *                         every rank must issue the same sequence of calls;
 *   segk_mailbox_status   synchronous: the number of exchanges completed so far, or a negative status when an
 *                         in-kernel wait timed out (results from then on are not sums). */
#define SEGK_MAILBOX_HANDLE_BYTES 64
#define SEGK_MAILBOX_MAX_DOUBLES 4352
int segk_mailbox_create(int world, void* handle_out);
int segk_mailbox_connect(int rank, int world, const void* handles, int timeout_ms);
int segk_mailbox_set_timeout(int timeout_ms);
int segk_mailbox_allreduce_f64(double* buf, long long n, segk_stream_t stream);
int segk_mailbox_allreduce_f32(float* buf, long long n, segk_stream_t stream);
long long segk_mailbox_status(void);
int segk_mailbox_destroy(void);

#ifdef __cplusplus
}
#endif
#endif /* SEGK_H */

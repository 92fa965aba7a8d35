/*
 * ldm_hip.h -- C ABI of libldm_hip.so: hand-written HIP kernels (gfx950 / MI355X)
 * for the latent-diffusion SAMPLING path of chao-ji/ldm_tf2.
 *
 * The reference has no FFI/plugin interface (it is pure Python on TensorFlow;
 * SURVEY.md section 8b), so this ABI is build-defined.  Each entry point names the
 * reference call site(s) whose arithmetic it replaces.  Conventions:
 *   - extern "C", plain pointers and sizes, no torch / C++ types;
 *   - every pointer is a DEVICE pointer unless it says "host";
 *   - the library never allocates, frees or synchronises; every kernel is
 *     enqueued on the caller's `stream` (a hipStream_t passed as void*), so the
 *     caller may capture any sequence of calls into a HIP graph;
 *   - return value: 0 = ok, negative = error (ldm_last_error() has the text);
 *   - activations are NHWC ("pixel rows" of C channels, explicit pixel stride
 *     `ld*` in ELEMENTS so a tensor may be a channel slice of a wider buffer:
 *     that is how the U-Net skip concatenation (unet.py:135) costs nothing);
 *   - dtype: 0 = float32, 1 = bfloat16 (storage; accumulation, statistics,
 *     softmax and the DDIM scheduler are always float32);
 *   - biases / gammas / betas / per-row addends are always float32.
 *
 * Why there is no opaque context (no ldm_create / ldm_destroy / *_sync): every entry point is
 * a pure function of its arguments that only ENQUEUES work on the caller's stream.  The library
 * owns no device memory (scratch such as the split-K workspace is passed in by the caller, who
 * therefore decides which stream / model it belongs to), keeps no per-call state (the only
 * mutable datum is the thread-local last-error string) and never synchronises, so there is
 * nothing for a handle to hold and nothing for a *_sync to wait on: the caller synchronises its
 * own stream.  Re-entrancy follows from that: concurrent calls on different streams are safe as
 * long as they do not share caller-owned scratch.  The library reads NO environment variable (the
 * timing ablations and A/B switches of the tools exist only in the separate tools build,
 * -DLDM_TOOLS_BUILD, which the product never loads).
 */
#ifndef LDM_HIP_H
#define LDM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LDM_F32 0
#define LDM_BF16 1

#define LDM_OK 0
#define LDM_ERR_ARG (-1)
#define LDM_ERR_LAUNCH (-2)
#define LDM_ERR_WORKSPACE (-3)

/* epilogue activations of ldm_gemm */
#define LDM_ACT_NONE 0
#define LDM_ACT_GELU 1  /* exact erf gelu: transformer.py:169 (tf.nn.gelu)        */
#define LDM_ACT_GEGLU 2 /* value*gelu(gate): unet.py:323-325; weight rows interleaved
                           in blocks of 64 = 32 value rows then their 32 gate rows  */
#define LDM_ACT_SILU 3  /* unet.py:72 Dense(activation="silu")                     */

int ldm_version(void);
/* copies the calling thread's last error text into buf (host), returns its length */
int ldm_last_error(char* buf, int n);

/*
 * Dense / 1x1-conv / attention-projection GEMM and implicit-GEMM 3x3 convolution.
 *
 *   out[b][m][n] = act( alpha * sum_k A[b][m][k] * Wt[b][n][k] + bias[n]
 *                       + addend[(m / add_rows) * add_ld + n] ) + residual[b][m][n]
 *
 * A rows are K-contiguous; Wt is the weight matrix stored [N][K] (K contiguous),
 * i.e. the TRANSPOSE of the reference's Dense kernel [in,out] / the OHWI form of
 * its HWIO conv kernel -- the host re-lays weights out once at load time.
 *
 * conv = 1: A is gathered on the fly from an NHWC image [B][H][W][Cin] (pixel
 *   stride lda): m = (b, oy, ox) over [B][OH][OW], k = (kh, kw, ci) over 3x3xCin,
 *   source pixel (oy*stride + kh - 1, ox*stride + kw - 1), zero outside; with
 *   upsample = 1 the source image is first nearest-2x upsampled
 *   (src[i][j] = img[i/2][j/2]).  Cin must be a multiple of 128/sizeof(elem).
 *   Replaces: Conv2D 3x3 SAME (unet.py:22,40,71,375,378; autoencoder.py:32,35,
 *   148,275), pad(1,1)+stride-2 VALID (unet.py:26-27), ResizeNearestNeighbor+conv
 *   (unet.py:44-47, autoencoder.py:152-155).
 * conv = 0: plain rows.  Replaces Dense (unet.py:72-73,320,332,350,353,376,379;
 *   autoencoder.py:36,69-72; transformer.py:137-138), Projection einsums
 *   (transformer.py:68,70) and, batched, the materialised attention einsums of
 *   autoencoder.py:86,93.
 *
 * Output element (m, n) of batch b is stored at out + b*stride_c + m*ldc_m + n*ldc_n
 * (ldc_n = 1 for row-major; ldc_m = 1 gives the transposed store used for V^T).
 * With LDM_ACT_GEGLU the stored width is N/2.
 * split_k > 1 needs `workspace` >= split_k*M*N*4 bytes (batch must be 1); the f32 partial
 * slabs are reduced (+ epilogue) by a second small launch.  (An in-kernel "last block
 * reduces" fix-up was measured and rejected: on this multi-XCD part the agent-scope fence /
 * device-coherent accesses it needs cost far more than the launch it saves; DESIGN.md.)
 * One workspace serves one stream at a time.
 */
typedef struct {
  const void* a;
  const void* w;
  const float* bias;     /* [N] or NULL */
  const float* addend;   /* per-row-group addend or NULL (ResBlock temb: unet.py:386-388) */
  const void* residual;  /* same dtype as out, or NULL */
  void* out;
  void* workspace;
  size_t workspace_bytes;
  int64_t lda, ldr, ldc_m, ldc_n;
  int64_t stride_a, stride_w, stride_c, stride_r; /* per batch, elements */
  int64_t add_ld;        /* row stride of addend (0 = one row for all groups) */
  int32_t M, N, K, batch;
  int32_t add_rows;      /* rows of out per addend row (OH*OW) */
  int32_t conv, B, H, W, Cin, OH, OW, stride, upsample;
  int32_t act;
  int32_t dtype;         /* of a, w */
  int32_t out_dtype;     /* of out, residual */
  int32_t split_k;       /* 0 = let the library choose */
  int32_t tile;          /* 0 = auto; 1-19 force an implicit-GEMM tile (9-12: bf16 16x16x32 MFMA path, 13-14: persistent ping-pong kernel; there split_k < 0 sets the workgroups per row panel, 15-16: tiles 9 / 11 with the halo-staged A operand for stride-1 convolutions whose 256-row M-tile is whole lines of one image, W = 16 or 32, 17-19: the 64x64 / 128x64 / 128x128 tiles with a 4- / 3- / 3-stage LDS ring for launches of 1-3 workgroups per CU), */
  float alpha;
  /* conv: 0 = one zero row/column before the image (Keras SAME at stride 1; pad (1,1)+VALID at
   * stride 2, unet.py:26-27); 1 = none before, one after -- the autoencoder's
   * pad [[0,1],[0,1]] + stride-2 VALID downsample (autoencoder.py:133-136); stride 2 only. */
  int32_t no_lead_pad;
  /* Second output: LayerNormalization of the row just written (unet.py:309-313: every
   * residual-stream update of the BasicTransformerBlock is followed by a LayerNorm whose output
   * feeds the next projection).  ln_out[m][:] = LN(out[m][:]) * ln_gamma + ln_beta over all N
   * columns, computed from the values as stored (i.e. after rounding to out_dtype), same dtype
   * as out, row stride ld_ln.  Needs a tile that holds whole rows: plain rows, batch 1,
   * N == 320, row-major 16-byte-aligned output, no GEGLU (ldm_gemm_ln_supported).  NULL = off. */
  void* ln_out;
  const float* ln_gamma;
  const float* ln_beta;
  int64_t ld_ln;
  float ln_eps;
  /* Second, TRANSPOSED output for the columns n >= n_split (the self-attention q|k|v projection
   * in one launch: q|k row-major into `out`, v straight into the V^T layout the attention kernel
   * reads, unet.py:270-276): element (m, n) goes to
   *   out2 + (m / rows2) * stride2 + (n - n_split) * ld2 + (m % rows2)      (dtype of out).
   * n_split must be a multiple of 160 and 128 or the forced tile's width (every tile then lies on
   * one side), rows2 a multiple of 4; plain rows, batch 1, no split-K, no residual on that part.
   * NULL = off. */
  void* out2;
  int64_t ld2, stride2;
  int32_t n_split, rows2;
  /* LayerNormalization of the A ROWS folded into the product (unet.py:309-313: LN -> q|k / v / q /
   * GEGLU projections).  With w = bf16(gamma (.) W), bias = b + W beta and ln_cs[n] = sum_k w[n][k]
   * (all prepared by the host, float32 [N], 16-byte aligned),
   *     LN(a) W^T + b  =  rstd_m * (sum_k a[m][k] w[n][k]  -  mean_m * ln_cs[n]) + bias[n],
   * and the kernel derives mean_m / rstd_m (eps = ln_eps, biased variance over the K columns) from
   * the A tiles it stages anyway: no LayerNorm launch, no normalised copy of the rows.  bf16 plain
   * rows (conv = 0), batch 1, persistent tiles 13 / 14 (chosen automatically), epilogues: bias,
   * bias + GEGLU, or the transposed store (out2 with n_split = 0).  NULL = off. */
  const float* ln_cs;
  /* Split-K only: 1 = leave the float32 partial slabs in `workspace` and do NOT launch the reduce.  The
   * caller completes the product later, on the same stream and with the SAME parameters, either with
   * ldm_gemm_reduce (the plain reduce + epilogue) or with ldm_groupnorm_splitk (reduce + epilogue +
   * the GroupNormalization that follows, one launch).  Ignored when the plan does not split K
   * (ldm_gemm_splits tells: the product is then complete when ldm_gemm returns). */
  int32_t defer_reduce;
  /* conv = 1, stride 1, no upsample: a SECOND A operand for the K columns beyond 9*Cin.  K = 9*Cin + Cin2 and
   *     out = conv3x3(a) . W[:, :9*Cin]^T  +  a2 . W[:, 9*Cin:]^T  (+ bias ...),
   * the extra columns being a 1x1 convolution over the NHWC image a2 [B][H][W][Cin2] (pixel stride lda2, dtype of
   * a) read at the output pixel.  The ResidualBlock's shortcut (unet.py:379-380, :393-397: a Dense over the block
   * input where the channel count changes, added to the second convolution's output) then accumulates inside that
   * convolution's K loop: no launch and no [M][Cout] tensor of its own; pass bias = conv bias + shortcut bias.
   * Cin2 a multiple of the K-tile (64 bf16 / 32 f32 elements); implicit-GEMM tiles 1-12 and 17-19 (not the
   * persistent or halo-staged ones); split-K, deferred reduce and every epilogue as without it.  NULL = off. */
  const void* a2;
  int64_t lda2;
  int32_t Cin2;
} ldm_gemm_params;

int ldm_gemm(const ldm_gemm_params* p, void* stream);
/* number of K slabs ldm_gemm writes for these parameters (1 = no split-K; host only) */
int ldm_gemm_splits(const ldm_gemm_params* p);
/* completes a product whose reduce was deferred (defer_reduce = 1): slabs -> bias / addend /
 * activation / residual -> out.  Same parameters as the ldm_gemm call. */
int ldm_gemm_reduce(const ldm_gemm_params* p, void* stream);
/*
 * Completes a deferred split-K product AND applies the GroupNormalization (+SiLU) that consumes it
 * (unet.py:383-392: conv -> GroupNorm -> SiLU -> conv), in ONE launch: a workgroup owns whole
 * groups of one sample, sums the slabs of its [HW][channels] slab (+ bias / per-sample addend /
 * residual, in the order of the plain reduce: the stored value is bit-identical), keeps the values
 * ROUNDED to the output dtype in registers, writes them to p->out when `store_out` (skip it when
 * nothing else reads the product: conv1 of a ResBlock) and writes GroupNorm(+SiLU) of them to
 * gn_out (pixel stride ld_gn, dtype of p->out).  p: the parameters of the deferred ldm_gemm call
 * (M = B*HW rows, N = C channels, row-major output, no activation, batch 1).  Shapes:
 * ldm_groupnorm_splitk_supported(B, HW, C, groups, out dtype).
 */
/* 1 if ldm_groupnorm_splitk supports this shape (host query; a subset of ldm_groupnorm_fused's) */
int ldm_groupnorm_splitk_supported(int B, int HW, int C, int groups, int dtype);
int ldm_groupnorm_splitk(const ldm_gemm_params* p, const float* gamma, const float* beta, void* gn_out,
                         int64_t ld_gn, int B, int HW, int groups, float eps, int silu, int store_out,
                         void* stream);
/* 1 if ldm_gemm accepts ln_out for a [M][N] output of this dtype (host query) */
int ldm_gemm_ln_supported(int N, int dtype);
/* the tile configuration and split-K factor ldm_gemm would pick for these params (host only;
 * nothing is launched) -- for tools and tests of the cost model */
int ldm_gemm_plan(const ldm_gemm_params* p, int* tile, int* split_k);
/* bytes of workspace ldm_gemm may need for these params with split_k = 0 (auto) */
size_t ldm_gemm_workspace_bytes(const ldm_gemm_params* p);

/*
 * Small direct 3x3 SAME convolution for tiny channel counts (Cin <= 8 or Cout <= 8):
 * conv_in 4->C (unet.py:71, autoencoder.py:275) and conv_out C->4/3 (unet.py:116,
 * autoencoder.py:289).  x [B][H][W][Cin] dtype in_dtype (pixel stride ldx), kernel
 * HWIO float32, bias float32, out dtype out_dtype (pixel stride ldo).
 */
int ldm_conv3x3_small(const void* x, int64_t ldx, int in_dtype, const float* kernel_hwio,
                      const float* bias, void* out, int64_t ldo, int out_dtype,
                      int B, int H, int W, int Cin, int Cout, void* stream);

/*
 * GroupNormalization over NHWC (Keras semantics: biased variance over (H,W,C/G)),
 * optionally followed by SiLU.  Two launches:
 *   ldm_groupnorm_partial : per (b, chunk, g) sums  -> partial[B][nchunks][G][2] float
 *   ldm_groupnorm_apply   : finalises mean/rstd, writes (x-mu)*rstd*gamma+beta [silu]
 * nchunks is chosen by the caller (ldm_groupnorm_nchunks gives the library's choice).
 * Replaces GroupNormalization(+tf.nn.silu): unet.py:115,137,354,374,377,383,390;
 * autoencoder.py:31,33,68,237,288.
 */
int ldm_groupnorm_nchunks(int B, int HW, int C);
int ldm_groupnorm_partial(const void* x, int64_t ldx, float* partial, int B, int HW, int C,
                          int groups, int nchunks, int dtype, void* stream);
int ldm_groupnorm_apply(const void* x, int64_t ldx, const float* partial, const float* gamma,
                        const float* beta, void* out, int64_t ldo, int B, int HW, int C,
                        int groups, int nchunks, float eps, int silu, int dtype, void* stream);
/* Single-launch variant for small images (every U-Net GroupNorm): one workgroup owns whole
 * groups of one sample and keeps its slab in registers between the passes, so x is read once
 * and written once, and the variance is the two-pass (centred) form.  _supported is a pure
 * host query; ldm_groupnorm_fused fails (LDM_ERR_ARG) on an unsupported shape. */
int ldm_groupnorm_fused_supported(int B, int HW, int C, int groups, int dtype);
int ldm_groupnorm_fused(const void* x, int64_t ldx, const float* gamma, const float* beta, void* out,
                        int64_t ldo, int B, int HW, int C, int groups, float eps, int silu, int dtype,
                        void* stream);
/* LayerNormalization over the last axis (unet.py:304-306; transformer.py:165,170,209). */
int ldm_layernorm(const void* x, int64_t ldx, const float* gamma, const float* beta, void* out,
                  int64_t ldo, int rows, int C, float eps, int dtype, void* stream);

/* Row softmax of a [rows][cols] matrix after scaling by `scale` (autoencoder.py:86-90:
 * einsum * C**-0.5 then softmax).  out may alias x when the dtypes are equal. */
int ldm_softmax_rows(const void* x, int64_t ldx, int in_dtype, void* out, int64_t ldo,
                     int out_dtype, int rows, int cols, float scale, void* stream);

/*
 * Fused multi-head attention, logits never materialised:
 *   out[b][q][h][:] = softmax_c( scale * q[b][q][h][:] . k[b][c][h][:] ) @ v[b][c][h][:]
 * q: [batch][Tq] rows of heads*Sp elements (row stride ldq), k likewise [batch][Tk];
 * vt: V transposed per head: [batch][heads][Sp][ldvt] with ldvt >= Tk keys contiguous and ldvt a
 *     multiple of 16 bytes; the padding columns [Tk, ldvt) are read in whole 16-byte chunks but
 *     masked to zero in the kernel: they need not be initialised (NaN / Inf there are harmless);
 * out: [batch][Tq] rows of heads*Sp (row stride ldo).  Sp = head size padded to a
 * multiple of 32 with zeros (the padding lives in the re-laid-out projection weights).
 * Replaces unet.py:280-287 and transformer.py:107-116 (scale applied AFTER q.k^T).
 */
int ldm_attention(const void* q, int64_t ldq, int64_t q_bs, const void* k, int64_t ldk,
                  int64_t k_bs, const void* vt, int64_t ldvt, int64_t vt_bs, void* out,
                  int64_t ldo, int64_t o_bs, int batch, int heads, int Tq, int Tk, int Sp,
                  float scale, int dtype, void* stream);

/*
 * The same attention for bf16 heads of 40 dims padded to Sp = 48, with the softmax bookkeeping moved
 * onto the matrix cores ("matrix-side softmax"): the caller's projections put the padding to work --
 *   q : already in the exp2 domain: the query projection's weights carry scale * log2(e) (so the
 *       scale is applied BEFORE q.k^T here; for bf16 that is a rounding-order difference);
 *       q[..][h][40..47] = 0
 *   k : k[..][h][40] = 1.0, k[..][h][41..47] = 0
 *   vt: row 40 of every head = 1.0 (rows 41..47 = 0); padding COLUMNS [Tk, ldvt) as for ldm_attention
 * The kernel writes -m (each query's reference point, kept a bf16 number) into dim 40 of its Q
 * fragments, so K.Q^T returns logit - m, and row 40 of V^T.P^T accumulates the softmax denominator from
 * the same bf16 P as the numerator.  out[..][h][40..47] = 0.  Same result as ldm_attention up to bf16
 * rounding (tests/test_round3_gpu.py); 2 of the 5 VALU operations per logit remain.
 */
int ldm_attention_ms(const void* q, int64_t ldq, int64_t q_bs, const void* k, int64_t ldk,
                     int64_t k_bs, const void* vt, int64_t ldvt, int64_t vt_bs, void* out,
                     int64_t ldo, int64_t o_bs, int batch, int heads, int Tq, int Tk, int Sp,
                     int dtype, void* stream);

/*
 * Fused feed-forward of the BasicTransformerBlock (unet.py:313, :323-325, :335-338) for bf16 rows of C = 320:
 *     out[m] = x[m] + b2 + W2 . ( a * gelu(g) ),   (a | g) = W1 . LayerNorm(x[m]) + b1
 * as ONE launch per 128-row panel: the panel's input rows stay resident in LDS (A operand of all hidden
 * chunks, source of the LayerNorm statistics, residual), the [M, 4C] hidden activation never leaves the CU,
 * only weights stream.  x: [M][C] bf16 (row stride ldx); w1: [8C][C] bf16, gamma-folded and
 * GEGLU-interleaved (layout.ln_fold of layout.geglu_kernel); aux: float32 [8C / 128][256]: per 128 rows of
 * w1 their column sums (128) then their folded bias (128); w2: [C][4C] bf16; b2: [C]; out: [M][C] bf16.
 * Same arithmetic as ldm_gemm(ln_cs, GEGLU) followed by ldm_gemm(residual) except that the hidden
 * activation is rounded to bf16 in LDS instead of HBM (identical rounding points).
 */
/* The same with the neighbouring products folded in (one launch for the tail of a SpatialTransformer block):
 *     h   = r0 + bo + Wo . att        cross-attention output projection + residual (unet.py:312; att [M][K0], K0 = 384)
 *     y   = feed-forward(h)           as above (h and y live only in LDS)
 *     out = r1 + bp + Wp . y          proj_out + the block's input as residual (unet.py:363-365)
 * wo [C][K0], wp [C][C] bf16 (K contiguous); r0, r1, out [M][C] bf16. */
int ldm_st_tail(const void* att, int64_t lda, int K0, const void* wo, const float* bo, const void* r0,
                int64_t ldr0, const void* w1, const float* aux, const void* w2, const float* b2,
                const void* wp, const float* bp, const void* r1, int64_t ldr1, void* out, int64_t ldo,
                int M, int C, float eps, int dtype, void* stream);
/* ldm_st_tail with the cross-attention itself in front (unet.py:273-291, :311): the input rows are the QUERY
 * projection q [M][K0] in ldm_attention_ms's layout (exp2-domain logits, 8 heads of 40 padded to 48); ctx_k
 * [R][Tk][K0] / ctx_vt [R][K0][ldv] are the context keys / values^T of the M / T samples in the same layout (Tk <= 80:
 * one key tile, plain softmax; T query rows per sample, a multiple of 128).  The attention runs in place in the LDS
 * panel, its output never reaches HBM. */
int ldm_st_xtail(const void* q, int64_t ldq, int K0, const void* ctx_k, const void* ctx_vt, int Tk, int ldv, int T,
                 const void* wo, const float* bo, const void* r0, int64_t ldr0, const void* w1, const float* aux,
                 const void* w2, const float* b2, const void* wp, const float* bp, const void* r1, int64_t ldr1,
                 void* out, int64_t ldo, int M, int C, float eps, int dtype, void* stream);
/* ... and with the block's middle in front of that (unet.py:310-311): the input rows are the SELF-attention's output
 * att [M][K0]; h1 = r0 + bo1 + Wo1 . att and the LayerNorm-folded query projection q = Wq . LayerNorm(h1) + bq (wq
 * [K0][C] gamma-folded with the attention scale * log2(e) and the padded rows of ldm_attention_ms's layout, qcs its
 * column sums, qb the folded bias: layout.ln_fold) run first; h1 is the residual of the second o-projection.  One
 * launch from the self-attention's output to the SpatialTransformer's output; `out` doubles as scratch for h1 and
 * must not alias an input.
 * `in_rows` (0 or M: off; else M / 2): att, r0 and r1 hold only the first in_rows rows and output row m >= in_rows
 * reads input row m - in_rows -- the classifier-free-guidance pair of the DDIM loop (model_runners.py:449-452 runs
 * the U-Net on concat([xt, xt]): in front of the first cross-attention both halves of the batch are the same
 * numbers, so the launches that feed this one ran once, on half the rows; ctx_k / ctx_vt cover all M / T samples). */
int ldm_st_block(const void* att, int64_t lda, int K0, const void* wo1, const float* bo1, const void* r0, int64_t ldr0,
                 const void* wq, const float* qcs, const float* qb, const void* ctx_k, const void* ctx_vt, int Tk,
                 int ldv, int T, const void* wo2, const float* bo2, const void* w1, const float* aux, const void* w2,
                 const float* b2, const void* wp, const float* bp, const void* r1, int64_t ldr1, void* out, int64_t ldo,
                 int M, int in_rows, int C, float eps, int dtype, void* stream);
int ldm_ffn_geglu_supported(int M, int C, int dtype);
int ldm_ffn_geglu(const void* x, int64_t ldx, const void* w1, const float* aux, const void* w2,
                  const float* b2, void* out, int64_t ldo, int M, int C, float eps, int dtype, void* stream);

/* Sinusoidal timestep embedding, cos first (unet.py:401-422): out[r][0:half]=cos(t*f),
 * out[r][half:]=sin(t*f), f_k = exp(-ln(10000)*k/half), float32.  t is read from
 * steps[*index] when `index` != NULL (device-resident DDIM index, graph replay) else
 * from t_rows[r]. */
int ldm_time_embedding(const int32_t* t_rows, const int32_t* steps, const int32_t* index,
                       float* out, int rows, int channels, void* stream);

/* y[r][n] = act_out( sum_k act_in(x[r][k]) * Wt[n][k] + bias[n] ), all float32 except Wt
 * (dtype).  Skinny (rows <= 64) Dense used for the timestep MLP and the 22 ResBlock
 * temb projections (unet.py:72-73,127,386).  act_*: LDM_ACT_NONE / LDM_ACT_SILU. */
int ldm_gemv(const float* x, int64_t ldx, const void* wt, const float* bias, float* y,
             int64_t ldy, int rows, int N, int K, int act_in, int act_out, int dtype,
             void* stream);

/* out[0:cols] = table[i][0:cols] with i = *index, float32; `pre_decrement` != 0: i = --(*index) first.  The DDIM
 * loop (model_runners.py:484-502) evaluates the timestep embedding, its MLP and the 22 ResBlock temb projections
 * (unet.py:125-127, :386) -- functions of the step's t alone -- for ALL its steps once (a [N_steps][sum Cout] table)
 * and each replayed step selects its row here; the same launch moves the device-resident loop counter (one
 * workgroup: no launch may read *index while another moves it), so the step needs no decrement of its own. */
int ldm_select_row(const float* table, int64_t ld, int rows, int cols, int32_t* index, int pre_decrement, float* out,
                   void* stream);

/*
 * Classifier-free guidance + DDIM update, float32 (model_runners.py:451-468):
 *   eps = eps_u + s*(eps_c - eps_u); x0 = c1*xt - c2*eps; mean = sqrt(a_prev)*x0
 *   + sqrt(1 - a_prev - sigma^2)*eps; xt' = mean + noise*sigma.
 * eps_all [2B][n] (uncond rows first), xt/xt_out [B][n]; noise (may be NULL when
 * sigma = 0) is read at noise + (*index) * noise_index_stride, i.e. a [N_steps][B][n]
 * table indexed by the DDIM index (stride 0 = one [B][n] buffer).
 * coef = device table [N_steps][4] of float32 (c1, c2, a_prev, sigma) gathered at
 * *index (the `_extract` cast-then-gather, :41-44).  If `dec_index` != 0 *index is
 * decremented afterwards (device-side loop counter for graph replay).
 * pred_x0_out (optional) receives x0 (:455-460, `return_pred_x0`).
 * x_unet_out (optional, dtype x_dtype) receives concat([xt', xt']) for the next step
 * (:452) so the next U-Net call reads it directly.
 */
int ldm_cfg_ddim_update(const float* eps_all, const float* xt, const float* noise,
                        int64_t noise_index_stride, float* xt_out, float* pred_x0_out,
                        void* x_unet_out, int x_dtype, const float* coef, int32_t* index,
                        int dec_index, float guidance_scale, int clip_denoised, int B,
                        int64_t n_per_sample, void* stream);

/* decode_first_stage prologue (model_runners.py:426 + autoencoder.py:362,434):
 * out = Dense_{C->C}(latents / scale_factor), C <= 8; float32 in, out_dtype out. */
int ldm_post_quant(const float* latents, float scale_factor, const float* kernel_io,
                   const float* bias, void* out, int out_dtype, int64_t pixels, int C,
                   void* stream);

/* DiagonalGaussian.sample / .mode (distribution.py:15-25,50): moments [pixels][2C] float32 =
 * (mean | logvar); out[p][c] = (mean + exp(0.5*logvar) * noise[p][c]) * out_scale, noise NULL
 * = mode.  The reference forms std from the UNCLIPPED logvar (:18); reproduced. */
int ldm_gaussian_sample(const float* moments, const float* noise, float* out, float out_scale,
                        int64_t pixels, int C, void* stream);

/* Nearest-codebook lookup (quantize.py:57-78): for each row z of [rows][C] (C <= 8):
 * idx = argmin_e |z|^2+|e|^2-2 z.e over codebook [V][C]; out = z + (e[idx]-z);
 * indices (int64) optional. */
int ldm_vq_nearest(const float* z, const float* codebook, float* out, int64_t* indices,
                   int64_t rows, int V, int C, void* stream);

/* tok_emb[ids] + pos_emb[0..T-1] (transformer.py:261-268); ids int64 [rows*T]. */
int ldm_embedding(const int64_t* ids, const float* tok_emb, const float* pos_emb, void* out,
                  int rows, int T, int D, int vocab, int out_dtype, void* stream);

/* tensor_to_image (run_ldm_sampler.py:18-25): per image (x-min)/(max-min)*255 -> uint8
 * (truncation).  scratch: 128 floats per image. */
int ldm_minmax_u8(const void* x, int in_dtype, uint8_t* out, float* scratch, int B,
                  int64_t n_per_image, void* stream);

/* dtype conversion / strided copy of [rows][cols] (cols contiguous). */
int ldm_cast(const void* x, int64_t ldx, int in_dtype, void* out, int64_t ldo, int out_dtype,
             int64_t rows, int cols, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LDM_HIP_H */

/*
 * dmhomo_hip.h — C ABI of libdmhomo_hip.so: the gfx950 (MI355X / CDNA4) kernels under
 * the DGM denoising hot path of lhaippp/DMHomo.
 *
 * The reference is pure Python/PyTorch and has NO native / FFI layer (SURVEY.md §0 fact 1,
 * §8b): every entry point below replaces a *sequence of stock ATen ops* in the reference,
 * cited per function as
 *      CFG = DGM/denoising_diffusion_models/classifier_free_guidance.py
 *      DDP = DGM/denoising_diffusion_models/denoising_diffusion_pytorch.py
 * and is bound from Python with ctypes (dmhomo_amd/_lib.py; INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - plain pointers + sizes; every pointer is DEVICE memory owned by the caller
 *     (torch tensors' data_ptr()), unless the name says host;
 *   - activations are fp32 NHWC ([B][H][W][C], C % 4 == 0); API-boundary images are the
 *     reference's fp32 NCHW;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     all work is enqueued asynchronously on it, nothing synchronises, nothing allocates
 *     (graph-capture safe);
 *   - return 0 on success, a negative DMH_E* code otherwise; dmh_last_error() gives the
 *     thread-local message. No exceptions cross the ABI.
 */
#ifndef DMHOMO_HIP_H
#define DMHOMO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DMH_OK 0
#define DMH_EINVAL (-1)  /* bad argument / unsupported shape */
#define DMH_ELAUNCH (-2) /* hip launch error */

const char* dmh_last_error(void);
/* DMH_ABI_VERSION of the library: bumped whenever a struct layout or an entry point changes; a binding must refuse a
 * library whose version differs from the header it was written against (dmhomo_amd/_lib.py does). */
#define DMH_ABI_VERSION 500
int dmh_version(void);

/* ---------------------------------------------------------------------------------------
 * Weight preparation (once per load_state_dict; weights are constant while sampling)
 * ------------------------------------------------------------------------------------- */

/* N1  WeightStandardizedConv2d weight fold, CFG:120-126:
 *   w_out[o] = (w[o] - mean_o) * rsqrt(var_biased_o + eps), stats over the K = Cin*kh*kw
 *   elements of output channel o.  w, w_out: [Cout][K]. */
int dmh_ws_standardize(const float* w, float* w_out, int Cout, int K, float eps, void* stream);

/* number of floats of the packed image of an OIHW weight for dmh_conv2d.
 * Every pure size function of this header (dmh_*_floats, dmh_conv_tiles, dmh_*_splits, dmh_multi_blocks) answers -1 for a
 * dimension outside (0, 2^20] (pixel counts: 2^26) or a NULL array instead of entering the size arithmetic with it. */
int64_t dmh_conv_pack_floats(int Cout, int C0, int C1, int KH, int KW);

/* OIHW [Cout][C0+C1][KH][KW] -> the image dmh_conv2d consumes (an opaque blob of dmh_conv_pack_floats floats; pack
 * again whenever the weight changes).  Default kernels (3x3, 1x1, 7x7, 4x4/stride 2): every output channel scaled by
 * a power of two, split into two fp16 planes and stored in MFMA-fragment order, followed by the per-channel 2^-k
 * (dmhomo_amd/csrc/conv_f16x3.hip); exact-fp32 kernels (2x2/stride 2, channel counts that are not a multiple of 32
 * for the strided conv, or DMH_CONV3_VARIANT != 9): tile-major fp32 [ntile][chunk][tap][64][KC].  Chunks never
 * straddle the two concatenated sources; padding is zero. */
int dmh_pack_conv_weight(const float* w_oihw, float* wpack, int Cout, int C0, int C1, int KH, int KW,
                         void* stream);
/* Many weights at once (the training step re-packs ~140 convolution weights after every optimiser update, DDP:1857: one
 * image for the forward conv and one for the data-gradient conv of each): three kernel phases over ALL jobs — weight
 * standardisation (N1) where a job asks for it, the per-output-channel power-of-two scales, the fp16-piece images — instead of
 * two or three launches per weight.  Only the fp16-piece images of stride-1 1x1 / 3x3 convolutions (the default kernels).
 *   src          OIHW weight the image is made from (after standardisation when ws != NULL)
 *   ws           NULL, or [Cout_src][K] scratch: src is standardised into it first (CFG:120-126) and the image is made from it;
 *                a later job may name this buffer as its src (phase 1 completes before phase 2 starts)
 *   transposed   0: the image of conv(src).  1: the image of the DATA-GRADIENT conv of conv(src) — taps flipped, (Cout, Cin)
 *                exchanged — for which Cout / C0 below are Cin_src / Cout_src and src is [C0][Cout][KH][KH]
 * jobs is a HOST array; nothing is read from it after the call returns. */
typedef struct DmhPackJob {
  const float* src;
  float* ws;
  float* wpack;       /* dmh_conv_pack_floats(Cout, C0, C1, KH, KH) floats */
  int32_t Cout, C0, C1, KH;
  int32_t transposed;
} DmhPackJob;
int dmh_pack_conv_weights_multi(const DmhPackJob* jobs, int njobs, float ws_eps, void* stream);

/* Upsample(nearest x2) + conv3x3 (CFG:106-107) in its sub-pixel form: output pixel (2y+dy, 2x+dx) depends on a 2x2
 * low-resolution neighbourhood through sums of the 3x3 taps, so the conv runs as four 2x2 convs (16 instead of 36
 * multiply-adds per low-resolution pixel; tap sums are formed in fp32 at pack time).  dmh_conv_up2_pack_floats returns
 * -1 when the build's conv variant / the shape (Cout % 64, one source) does not offer it; use the image with
 * DmhConv.upsample2 = 2, KH = KW = 3, stride = 1. */
int64_t dmh_conv_up2_pack_floats(int Cout, int C0);
int dmh_pack_conv_weight_up2(const float* w_oihw, float* wpack, int Cout, int C0, void* stream);

/* ---------------------------------------------------------------------------------------
 * Row subsets (round 5): de-duplication of the two classifier-free-guidance passes inside a captured denoise step.
 * forward_with_cond_scale (CFG:403-410) runs the UNet on the batch twice: once with the class embedding of every row
 * replaced by the null embedding with probability cond_drop_prob (CFG:415-425 — 0.5 in DGM, also while sampling), once with
 * all of them replaced.  A row the first pass dropped has exactly the inputs of its row in the second pass, so its logits are
 * the second pass's logits and need not be computed.  Which rows are kept is DEVICE data (the mask is drawn inside the
 * captured step), so every batched entry point of the UNet takes an optional `rows`:
 *     rows == NULL          all B rows of the launch
 *     rows[0]               n <= B, the number of active rows
 *     rows[1 + j], j < n    the row (sample slot) logical row j works on
 * The launch keeps its B-row grid; workgroups of logical rows >= n retire before touching memory, tensors are addressed by
 * the slot, nothing is gathered or scattered, and the slots that are not listed are neither read nor written.  Rows are
 * independent of each other in every kernel of the path, so the listed rows come out bit for bit as in the full launch.
 * dmh_rows_from_keep builds the list: the slots b < B with keep[b] != 0 in ascending order, followed by B .. B+extra-1
 * (the null pass's rows when both passes share one 2B-row launch); rows: int32 [1 + B + extra], all DEVICE memory. */
int dmh_rows_from_keep(const uint8_t* keep, int B, int extra, int32_t* rows, void* stream);

/* ---------------------------------------------------------------------------------------
 * K1/K2  convolution as an implicit GEMM.  fp32 tensors in and out; by default the products run on the fp16
 * matrix cores with every fp32 operand carried as block-scaled fp16 pieces (three v_mfma_f32_16x16x32_f16 per
 * product block, fp32 accumulation; error at the level of the fp32 accumulation rounding, fp32 exponent range —
 * DESIGN.md 3.1).  DMH_CONV3_VARIANT=0..3 / 6 (environment, read once) selects the exact-fp32 MFMA kernels.
 * ------------------------------------------------------------------------------------- */
typedef struct DmhConv {
  uint64_t struct_size; /* sizeof(DmhConv) of the header the caller was built against: dmh_conv2d refuses any other value
                           (the struct has grown between library versions; a shorter one would be read past its end) */
  const float* src0; /* NHWC [B][Hin][Win][C0] */
  const float* src1; /* NHWC [B][Hin][Win][C1] or NULL: torch.cat((src0, src1), dim=1) fused (CFG:454,457,463) */
  const float* wpack; /* dmh_pack_conv_weight image */
  const float* bias;  /* [Cout] or NULL */
  /* prologue on src0 (C1 must be 0): x <- SiLU(a[b][c]*x + b[b][c]) with coef = [B][2][C0]
   * from dmh_gn_finalize: GroupNorm + (scale+1, shift) + SiLU of the producing Block, CFG:206-212 */
  const float* in_coef;
  /* epilogue residual, NHWC [B][Hout][Wout][Cout] or NULL:
   *   res_coef == NULL : out += res                      (Residual, CFG:102-103)
   *   res_coef != NULL : out += SiLU(a*res + b)          (h + res_conv(x), CFG:241, h = block2 output) */
  const float* res;
  const float* res_coef;
  float* out;   /* NHWC [B][Hout][Wout][Cout] */
  float* stats; /* NULL or [B][tiles][Cout][2] per-tile (sum, sum of squares) of `out` for GroupNorm */
  int32_t B, Hin, Win, C0, C1, Cout;
  int32_t KH, KW;     /* 1x1, 3x3, 7x7 (stride 1, pad k/2); 4x4 (stride 2, pad 1); 2x2 (stride 2, pad 0) */
  int32_t stride;     /* 1 or 2 */
  int32_t upsample2;  /* 1: nearest x2 of the input fused into the 3x3 gather (Upsample, CFG:106-107);
                         2: the same conv with a wpack from dmh_pack_conv_weight_up2 (four 2x2 sub-pixel convs) */
  /* optional, with in_coef: in_bound[b * in_bound_n + i], i < in_bound_n — upper bounds of |a*x + b| over the elements
   * of sample b (dmh_gn_finalize_bound: one per GroupNorm group).  The fp16-piece kernels then take their block scale
   * from max_i in_bound instead of searching the staged tile for its maximum; the result is the same convolution. */
  const float* in_bound;
  int32_t in_bound_n;
  /* optional (1x1, stride 1, Cout <= 64, the fp16-piece kernels): a second, pointwise projection applied to the finished
   * output pixel while it is still in registers — fin_out[b][o][y][x] = fin_b[o] + sum_c fin_w[o][c] * out[b][y][x][c],
   * o < fin_n <= 8, written NCHW: the UNet's final_conv (CFG:341, 471-472) fused into the last ResnetBlock's res_conv
   * launch, bit for bit what dmh_final_conv_nchw computes from `out`.  With fin_w set, out may be NULL (not stored). */
  int32_t fin_n;
  const float* fin_w; /* [fin_n][Cout] */
  const float* fin_b; /* [fin_n] or NULL */
  float* fin_out;     /* NCHW [B][fin_n][Hout][Wout] */
  /* optional (1x1, stride 1, Cout == 64, the fp16-piece kernels): pix_stats[(b*Hout + y)*Wout + x][2] = (mean, rstd) of the
   * channel LayerNorm (CFG:137-141, eps pix_eps) of every finished output pixel — bit for bit what dmh_pixel_stats computes
   * from `out` — for the fused LinearAttention that consumes this launch's output (the up path's res_conv, CFG:241 -> 246). */
  float* pix_stats;
  float pix_eps;
  /* optional: the active row subset of this launch (see "Row subsets" below); NULL = all B rows */
  const int32_t* rows;
} DmhConv;

/* tiles per sample that dmh_conv2d will use for this geometry (size of the stats buffer) */
int dmh_conv_tiles(int Hout, int Wout, int KH, int stride);
int dmh_conv2d(const DmhConv* d, void* stream);

/* ---------------------------------------------------------------------------------------
 * Normalisation / activation glue (HBM-bound)
 * ------------------------------------------------------------------------------------- */

/* N2  GroupNorm statistics -> per-(sample, channel) affine of Block.forward, CFG:206-210:
 *   coef[b][0][c] = a = rstd*gamma*(scale+1), coef[b][1][c] = (beta - mean*rstd*gamma)*(scale+1) + shift
 * stats: [B][tiles][C][2] from dmh_conv2d; ss: NULL or row b at ss + b*ss_stride holds
 * (scale[C], shift[C]) (ResnetBlock mlp output chunk(2), CFG:233-235). Reduction in f64, fixed order. */
int dmh_gn_finalize(const float* stats, int tiles, const float* gamma, const float* beta, const float* ss,
                    int64_t ss_stride, float* coef, int B, int C, int groups, int hw, float eps, const int32_t* rows,
                    void* stream);

/* dmh_gn_finalize + bound[B][groups]: max over the group's channels of |a| * sqrt(sum x^2) + |b + mean*a| >= |a*x + b|
 * for every element x of the (sample, group) the statistics describe (DmhConv.in_bound) */
int dmh_gn_finalize_bound(const float* stats, int tiles, const float* gamma, const float* beta, const float* ss,
                          int64_t ss_stride, float* coef, float* bound, int B, int C, int groups, int hw, float eps,
                          const int32_t* rows, void* stream);

/* out = SiLU(a*y + b) + res   (identity res_conv branch of ResnetBlock, CFG:225,241) */
int dmh_gn_silu_residual(const float* y, const float* coef, const float* res, float* out, int B, int HW, int C,
                         const int32_t* rows, void* stream);

/* same, also writing pstats [B*HW][2] = (mean, rstd) of the channel LayerNorm (CFG:137-141) of every output pixel —
 * bitwise what dmh_pixel_stats(out) gives — for a LinearAttention that follows the block (CFG:176-183). C in {64,128,256}. */
int dmh_gn_silu_residual_stats(const float* y, const float* coef, const float* res, float* out, float* pstats, int B,
                               int HW, int C, float eps, const int32_t* rows, void* stream);

/* N4  channel LayerNorm (biased var, gain only), CFG:137-141, optionally + res (Residual, CFG:103).
 * rows (optional row subset) comes with pix_per_row = the pixels of one row, npix = B * pix_per_row. */
int dmh_chan_layernorm(const float* x, const float* g, const float* res, float* out, int64_t npix, int C,
                       float eps, const int32_t* rows, int64_t pix_per_row, void* stream);

/* ---------------------------------------------------------------------------------------
 * K3  LinearAttention core, CFG:258-269.  qkv: NHWC [B][n][384] = (q|k|v) x 4 heads x 32.
 * ------------------------------------------------------------------------------------- */
int dmh_linattn_splits(int n);
int64_t dmh_linattn_partial_floats(int B, int n);
/* pass 1: per split of the n pixels, running max / sum of exp / unnormalised k^T v per head */
int dmh_linattn_context(const float* qkv, float* partial, int B, int n, const int32_t* rows, void* stream);
/* merge the splits: ctx[b][h][d][e] = softmax_n(k)[d,:] . (v/n)[e,:] */
int dmh_linattn_merge(const float* partial, float* ctx, int B, int n, const int32_t* rows, void* stream);
/* pass 2: out[b][p][h*32+e] = sum_d ctx[d][e] * (softmax_d(q[p]) * scale)[d]; out NHWC [B][n][128] */
int dmh_linattn_apply(const float* qkv, const float* ctx, float* out, int B, int n, float scale, const int32_t* rows,
                      void* stream);

/* Fused LinearAttention (PreNorm LayerNorm + to_qkv + attention core, CFG:96-103,246-269): q, k, v stay on chip.
 *   stats  = dmh_pixel_stats(x)                         per-pixel (mean, rstd) of the PreNorm LayerNorm, [npix][2]
 *   wpack  = dmh_linattn_fused_pack(to_qkv.weight)      once per weight version; dmh_linattn_fused_pack_floats(C) floats
 *   dmh_linattn_fused_context -> partial[B][dmh_linattn_fused_splits(B,n)][4][1088]
 *   dmh_linattn_merge_n       -> ctx[B][4][32][32]
 *   dmh_linattn_fused_apply   -> out[B][n][128]         (then to_out conv + LayerNorm + residual as before)
 * x: NHWC [B][n][C], C a multiple of 32. */
int dmh_pixel_stats(const float* x, float* stats, int64_t npix, int C, float eps, const int32_t* rows, int64_t pix_per_row,
                    void* stream);
int64_t dmh_linattn_fused_pack_floats(int C);
int dmh_linattn_fused_pack(const float* w_qkv, float* wpack, int C, void* stream);
int dmh_linattn_fused_splits(int B, int n);
int dmh_linattn_fused_context(const float* x, const float* stats, const float* ln_g, const float* wpack,
                              float* partial, int B, int n, int C, const int32_t* rows, void* stream);
int dmh_linattn_merge_n(const float* partial, float* ctx, int B, int n, int nsplit, const int32_t* rows, void* stream);
int dmh_linattn_fused_apply(const float* x, const float* stats, const float* ln_g, const float* wpack,
                            const float* ctx, float* out, int B, int n, int C, float scale, const int32_t* rows,
                            void* stream);
/* C == 64: pass 2 carries the rest of the block too — y[B][n][64] = x + LayerNorm(to_out(attention) + bias) * g
 * (CFG:254-256, 103): the to_out weight [64][128] is packed once by dmh_linattn_out_pack
 * (dmh_linattn_out_pack_floats() floats). */
int64_t dmh_linattn_out_pack_floats(void);
int dmh_linattn_out_pack(const float* w_out, float* wpack, void* stream);
int dmh_linattn_fused_apply_out(const float* x, const float* stats, const float* ln_g, const float* wpack,
                                const float* ctx, const float* wopack, const float* out_bias, const float* out_ln_g,
                                float* y, int B, int n, int C, float scale, float eps, const int32_t* rows, void* stream);

/* K4  Attention core, CFG:287-295: softmax_j((q*scale)^T k) v; out NHWC [B][n][128] */
int dmh_attention(const float* qkv, float* out, int B, int n, float scale, const int32_t* rows, void* stream);

/* ---------------------------------------------------------------------------------------
 * K5  embeddings / small linears
 * ------------------------------------------------------------------------------------- */
/* N7  SinusoidalPosEmb, CFG:165-172: out[r] = (sin(t_r*f), cos(t_r*f)); freq[dim/2] fp32 table */
int dmh_sinusoidal_embed(const int64_t* t, const float* freq, float* out, int R, int dim, void* stream);
/* RandomOrLearnedSinusoidalPosEmb, CFG:175-190 [DDP:179-195] (Unet(learned_sinusoidal_cond / random_fourier_features)):
 * out[r] = (t_r, sin(t_r*w_i*2*pi) i < half, cos(t_r*w_i*2*pi) i < half), out [R][2*half + 1]; weights [half] */
int dmh_fourier_embed(const int64_t* t, const float* weights, float* out, int R, int half, void* stream);
/* N8  classes_emb lookup + null swap, CFG:419-425: keep==NULL keeps every row; table [num_classes][dim] — a kept class id
 * outside [0, num_classes) yields a NaN row (device data: it cannot be refused at launch) */
int dmh_class_embed(const int64_t* classes, const uint8_t* keep, const float* table, const float* null_emb,
                    float* out, int R, int dim, int num_classes, void* stream);
/* y[r][o] = act_out( sum_i act_in(x[r][i]) * wt[i][o] + bias[o] ), wt = W^T [in][out].
 * act: 0 none, 1 SiLU, 2 GELU(erf).  (time_mlp / classes_mlp CFG:353,362; ResnetBlock.mlp CFG:220) */
int dmh_linear(const float* x, int64_t x_stride, const float* wt, const float* bias, float* y, int64_t y_stride,
               int R, int in_dim, int out_dim, int act_in, int act_out, void* stream);

/* The (scale, shift) rows of all ResnetBlocks (ResnetBlock.mlp CFG:220,231-235 over [time_mlp(t) | classes_mlp(c)], CFG:427,435)
 * of a REPLAYED denoise step from tables: out[b][0..N) = (T[*cursor] + C[keep[b] ? classes[b] : ncls]) + bias, with T [S][N] the
 * time half's partial sums of dmh_linear per denoise step and C [ncls+1][N] the class half's per class (last row: null
 * embedding), both made by dmh_linear itself on inputs whose other half is zero; bitwise dmh_linear on the full input.
 * cursor: the step cursor of dmh_sampler_seek (device); keep may be NULL (every class kept).  S = rows of T: a cursor outside
 * [0, S) or a kept class outside [0, ncls) cannot be refused at launch (device data) and yields a NaN row. */
int dmh_ss_gather(const float* T, const float* C, const float* bias, const int32_t* cursor, const int64_t* classes,
                  const uint8_t* keep, int ncls, float* out, int B, int N, int S, void* stream);

/* ---------------------------------------------------------------------------------------
 * K6  sampler glue (NCHW at the API boundary)
 * ------------------------------------------------------------------------------------- */
/* network input: out NHWC [reps*B][H][W][Cpad] <- cat(a [B][Ca][H][W], b*m [B][Cb][H][W]) zero padded;
 * b, m may be NULL (Cb = 0).  torch.concat((x, rgb_flow*mask)) CFG:430 / cat((x_self_cond, x)) DDP:411 */
int dmh_assemble_input(const float* a, int Ca, const float* b, int Cb, const float* m, float* out, int B, int reps,
                       int HW, int Cpad, void* stream);
/* final_conv 1x1 (CFG:401,466): NHWC [R][HW][C] -> NCHW [R][Cout][HW], Cout <= 16 */
int dmh_final_conv_nchw(const float* x, const float* w, const float* bias, float* out, int R, int HW, int C,
                        int Cout, void* stream);

typedef struct DmhStep {
  int32_t objective;  /* 0 pred_noise, 1 pred_x0, 2 pred_v            CFG:614-628 */
  int32_t clip;       /* clamp x_start to [-1,1]                       CFG:612 */
  int32_t mode;       /* 0 DDIM update, 1 DDIM last (img = x_start), 2 DDPM posterior step */
  float cond_scale;   /* CFG blend null + (cond-null)*s, CFG:410; ignored when model_null == NULL */
  float sqrt_recip_ac, sqrt_recipm1_ac, sqrt_ac, sqrt_1m_ac; /* extract(..., t) CFG:586-601 */
  float c0, c1, c2;   /* DDIM: sqrt(alpha_next), c, sigma (CFG:697-707); DDPM: coef1, coef2, exp(.5 logvar) */
} DmhStep;
/* one sampler step on NCHW tensors of n elements: writes img_out and (if non-NULL) x_start, pred_noise.
 * keep (optional, with model_null; per_row = elements of one row): uint8 [n / per_row] — rows with keep == 0 were not computed
 * by the conditional pass (row subsets, above) and take model_null as their conditional logits, which is what CFG:404,409
 * compute for a row whose class was dropped. */
int dmh_sampler_step(const DmhStep* s, const float* model_cond, const float* model_null, const float* x,
                     const float* noise, float* img_out, float* x_start, float* pred_noise, int64_t n,
                     const uint8_t* keep, int64_t per_row, void* stream);

/* The same step with its DmhStep read from DEVICE memory, for a sampling loop (CFG:683-707, DDP:647-735) that replays
 * ONE captured denoise step from a HIP graph: the host fills `table` (one DmhStep per denoise step, host-computed as
 * the reference computes them, CFG:697-701) and `times` once; a device cursor selects the current entry.
 * img_out may alias x (the loop's `img = ...` in place); mode and objective are validated when the table is built. */
int dmh_sampler_step_dev(const DmhStep* cur_dev, const float* model_cond, const float* model_null, const float* x,
                         const float* noise, float* img_out, float* x_start, float* pred_noise, int64_t n,
                         const uint8_t* keep, int64_t per_row, void* stream);
/* cursor handling of that loop (one tiny launch): k >= 0: *cursor = k; k < 0: *cursor = min(*cursor + 1, S - 1); then
 * *cur = table[*cursor] and tcond[0 .. B) = times[*cursor] (the `time_cond` tensor of CFG:684 / DDP:700).
 * table: [S] DmhStep, times: [S] int64, cursor: int32, all device memory. */
int dmh_sampler_seek(int32_t* cursor, int k, const DmhStep* table, const int64_t* times, int S, DmhStep* cur,
                     int64_t* tcond, int B, void* stream);

/* Noise of the sampling loop keyed by GLOBAL sample index: replaces torch.randn(shape) CFG:679,
 * torch.randn_like(img) CFG:705 and torch.zeros(B).uniform_(0, 1) CFG:90 where a run is sharded over ranks.
 * out [B][per_sample]: element e of row b = f(seed, sample_ids[b], draw, e) with f = Philox4x32-10 (key = seed, counter =
 * (e / 4, draw, sample id lo, hi)) followed by Box-Muller (kind 0: N(0,1)), the top 24 bits * 2^-24 (kind 1: uniform
 * [0,1)) or nothing (kind 2: the raw 32-bit words as float bit patterns, for known-answer tests) — so any row is the
 * same whichever rank, batch size or row position computes it.
 * state: 4 x uint64 in DEVICE memory: [0] seed, [1] draw index — read by every workgroup, advanced by one by the last
 * workgroup of the launch to finish, so a captured launch replays with the next draw —, [2] arrival tickets (0 between
 * launches), [3] reserved.  sample_ids: [B] int64. */
int dmh_rng_indexed(float* out, int B, int64_t per_sample, const int64_t* sample_ids, uint64_t* state, int kind,
                    void* stream);

/* prob_mask_like CFG:84-90 on the indexed generator in one launch: keep[b] = (the kind-1 uniform of row b at the current
 * draw) < prob, uint8; advances the draw index by one (it IS that draw) */
int dmh_rng_keep_mask(uint8_t* keep, int B, const int64_t* sample_ids, uint64_t* state, float prob, void* stream);

/* y = x*scale + shift elementwise (normalize / unnormalize, CFG:69-74) */
int dmh_affine(const float* x, float* y, float scale, float shift, int64_t n, void* stream);
/* in place on channels >= c0 of an NCHW tensor: x = x*scale + shift (flow channel remap, DDP:679,728) */
int dmh_affine_tail(float* x, int B, int C, int HW, int c0, float scale, float shift, void* stream);
/* D4 / D7 with a timestep per row: out = ca[b]*x (+ cb[b]*y) (/ dv[b]) (clamped to [-1,1] when clamp != 0); y, cb, dv may be
 * NULL.  predict_start_from_noise / predict_noise_from_start / predict_v / predict_start_from_v CFG:586-601, the posterior
 * mean of q_posterior CFG:603-608 and the optional clamp of model_predictions CFG:612, in the reference's op order */
int dmh_rows_lincomb(const float* x, const float* y, const float* ca, const float* cb, const float* dv, float* out, int B,
                     int64_t per_sample, int clamp, void* stream);
/* D9 q_sample CFG:738-742: out = ca[b]*x_start + cb[b]*noise, ca/cb = extract(sqrt_ac / sqrt_1m_ac, t) */
int dmh_q_sample(const float* x_start, const float* noise, const float* ca, const float* cb, float* out, int B,
                 int64_t per_sample, void* stream);
/* D9 loss terms of p_losses (CFG:796-806), forward value only:
 *   out[b] = mean over (C,HW) of m[b][hw] * |a - b|   (squared != 0: (a - b)^2); m may be NULL.
 * Deterministic two-stage reduction; ws: [B][64] f64 scratch. */
int dmh_diff_mean(const float* a, const float* b, const float* m, int squared, double* ws, float* out, int B, int C,
                  int HW, void* stream);
/* loss.mean() + (w * photo).mean() over the B per-sample means (CFG:804-806) -> out[0] */
int dmh_loss_combine(const float* l, const float* photo, const float* w, float* out, int B, void* stream);
/* G6  saveTrainPair DDP:1673: uint8 truncation of img*255 */
int dmh_to_uint8(const float* img, uint8_t* out, int64_t n, void* stream);

/* ---------------------------------------------------------------------------------------
 * K7-K9  condition builder & geometry
 * ------------------------------------------------------------------------------------- */
/* G2+G3  homo_to_flow DDP:927-975 + flow_to_image DDP:1471-1486.  Hm: [B][9] f64 (device);
 * flow NCHW [B][2][H][W] fp32, rgb NCHW [B][3][H][W] fp32 (either may be NULL) */
int dmh_homography_flow(const double* Hm, float* flow, float* rgb, int B, int H, int W, float max_flow,
                        void* stream);
/* G3 alone: flow NCHW [B][2][HW] -> rgb NCHW [B][3][HW] (flow_to_image DDP:1471-1486) */
int dmh_flow_to_image(const float* flow, float* rgb, int B, int HW, float max_flow, void* stream);
/* G4  flow_warp DDP:1262-1299 (grid_sample, align_corners=True); x/out NCHW [B][C][H][W], flow [B][2][H][W].
 * pad: 0 'border' (the reference's default), 1 'zeros', 2 'reflection'; mode: 0 'bilinear' (default), 1 'nearest', 2 'bicubic' — the
 * values flow_warp forwards to grid_sample's padding_mode / mode (DDP:1262,1270-1274).
 * x0/y0 (NULL or int32 [B][H][W]; pad == 0 && mode == 0 only) receive the top-left corner indices (bit-exact contract). */
int dmh_flow_warp(const float* x, const float* flow, float* out, int32_t* x0, int32_t* y0, int B, int C, int H,
                  int W, int pad, int mode, void* stream);
/* get_grid DDP:1558-1574: out (B,2,H,W) fp32 = (x + start, y + start) of every pixel */
int dmh_pixel_grid(float* out, int B, int H, int W, float start, void* stream);
/* norm_grid DDP:1292-1299: v (B,2,H,W) -> out (B,H,W,2) = (2.0*v_x/(W-1) - 1.0, 2.0*v_y/(H-1) - 1.0), fp32, that op order */
int dmh_norm_grid(const float* v, float* out, int B, int H, int W, void* stream);
/* get_flow_np DDP:927-969 in its general form: Hm (B,divide,3,3) f64 — row y uses band min(y / (H/divide), divide-1) —,
 * idx (Bi,3,H,W) f64 homogeneous coordinates (Bi == B or 1: mesh_grid_np), flow (B,2,H,W) f64 = H.p / (w' + 1e-6) - p */
int dmh_homography_flow_points(const double* Hm, const double* idx, double* flow, int B, int divide, int Bi, int H, int W,
                               void* stream);
/* DLT_solve DDP:1612-1643 on explicit correspondences: src, off (N,P,2) f64 -> Hout (N,3,3) f64, one least-squares
 * homography per system (P == 4: the square system solved directly; P > 4: f64 normal equations as dmh_dlt_homography);
 * ws: [N][DMH_DLT_BLOCKS][44] f64 (unused for P == 4) */
int dmh_dlt_points(const double* src, const double* off, double* ws, double* Hout, int N, int P, void* stream);
/* G5  homo_gen / DLT_solve DDP:1577-1661 through f64 normal equations; ws: [B][DMH_DLT_BLOCKS][44] f64 */
#define DMH_DLT_BLOCKS 64
int dmh_dlt_homography(const float* flow, double* ws, double* Hout, int B, int H, int W, void* stream);

/* ---------------------------------------------------------------------------------------
 * Training: the pieces of loss.backward() / clip_grad_norm_ / Adam.step / ema.update (DDP:1843-1865);
 * dmhomo_amd/train.py strings them into the optimiser step.
 * ------------------------------------------------------------------------------------- */

/* weight / bias gradient of a stride-1 KHxKH convolution, i.e. autograd of F.conv2d (CFG:128) wrt weight and bias:
 *   dw[o][c][ky][kx] = sum dy[b][y][x][o] * X[b][y+ky-p][x+kx-p][c],  db[o] = sum dy[b][y][x][o],
 * X = cat(src0, src1) with the optional consumer-side prologue X = SiLU(a*src0 + b) of dmh_conv2d.
 *   KH = 1, 3, 7: 'same' conv (p = KH/2); ups = 1 (KH = 3): the conv saw the nearest x2 upsampling of the stored
 *                 input [B][H/2][W/2][C0] (Upsample, CFG:106-107);
 *   KH = 2      : 'valid' 2x2 conv (p = 0) over a stored input [B][H+1][W+1][C0] — the space-to-depth form
 *                 (dmh_s2d_shift) of the 4x4 / stride-2 Downsample conv (CFG:110-111).
 * KH = 2, 3: fp16 pieces of both operands on the fp16 matrix cores, fp32 accumulate (error at the fp32-accumulation
 * level; DMH_WGRAD_VARIANT=0: exact fp32 everywhere); KH = 1, 7: exact fp32 (v_mfma_f32_16x16x4_f32).  Deterministic
 * (pixel splits reduced in a fixed order).  The DATA gradient is
 * dmh_conv2d itself on dy with the weight flipped in both taps and transposed in (Cout, Cin).
 * dy: NHWC [B][H][W][Cout]; dw: OIHW [Cout][C0+C1][KH][KH]; db: [Cout] or NULL; work: ..._workspace_floats floats. */
int64_t dmh_conv_wgrad_workspace_floats(int B, int H, int W, int C0, int C1, int Cout, int KH);
int dmh_conv_wgrad(const float* dy, const float* src0, const float* src1, const float* in_coef, float* dw, float* db,
                   float* work, int B, int H, int W, int C0, int C1, int Cout, int KH, int ups, void* stream);
/* layout helpers of the strided / upsampled convs' backward:
 *   dmh_s2d_shift  X[b][cy][cx][(py*2+px)*C + c] = x[b][2cy-1+py][2cx-1+px][c] (0 outside), [B][H/2+1][W/2+1][4C]
 *   dmh_d2s        out[b][2m+ry][2l+rx][c] = in[b][m][l][(ry*2+rx)*C + c]        [B][H][W][4C] -> [B][2H][2W][C]
 *   dmh_sumpool2   out[b][m][l][c] = sum of the 2x2 block of in                  [B][2H][2W][C] -> [B][H][W][C] */
int dmh_s2d_shift(const float* x, float* X, int B, int H, int W, int C, void* stream);
int dmh_d2s(const float* in, float* out, int B, int H, int W, int C, void* stream);
int dmh_sumpool2(const float* in, float* out, int B, int H, int W, int C, void* stream);

/* dmh_gn_finalize that also saves (mean, rstd) per (sample, group) for the backward pass: mr [B][groups][2] */
int dmh_gn_finalize_train(const float* stats, int tiles, const float* gamma, const float* beta, const float* ss,
                          int64_t ss_stride, float* coef, float* mr, int B, int C, int groups, int hw, float eps,
                          void* stream);

/* backward of Block's GroupNorm -> (scale+1, shift) -> SiLU (CFG:206-212) given the conv output y it normalised:
 *   dout: gradient wrt SiLU(GN(y)...) [B][HW][C];  y, coef [B][2][C], mr [B][groups][2]: as seen / saved by the forward;
 *   dy [B][HW][C];  pg [B][4][C]: per-sample parts of (dgamma, dbeta) — add them over b with dmh_sum_over_batch — and
 *   (d scale, d shift) of the ResnetBlock mlp output;  part [B][dmh_gn_bwd_chunks(HW)][C][2], bcoef [B][3][C]: work. */
int dmh_gn_bwd_chunks(int HW);
int dmh_gn_silu_backward(const float* dout, const float* y, const float* coef, const float* mr, const float* gamma,
                         const float* beta, const float* ss, int64_t ss_stride, float* dy, float* pg, float* part,
                         float* bcoef, int B, int HW, int C, int groups, void* stream);
/* out[i] = sum_b in[b][i], i < per (fixed order) */
int dmh_sum_over_batch(const float* in, float* out, int B, int64_t per, void* stream);
/* backward of the weight standardisation CFG:120-126: dw from the gradient dwh wrt the standardised weight. [Cout][K] */
int dmh_ws_backward(const float* w, const float* dwh, float* dw, int Cout, int K, float eps, void* stream);

/* backward of the channel LayerNorm N4 (CFG:137-141): dx [npix][C] and dmh_lnb_blocks() partial rows of dg
 * (dg_part [blocks][C]; add them with dmh_sum_over_batch).  d res = dout when the forward added a residual. */
int dmh_lnb_blocks(void);
int dmh_chan_layernorm_backward(const float* x, const float* g, const float* dout, float* dx, float* dg_part,
                                int64_t npix, int C, float eps, void* stream);

/* backward of the LinearAttention core (CFG:258-269) on a stored qkv [B][n][384]:  forward with dmh_linattn_context,
 * dmh_linattn_merge_ms (saves ms [B][4][32][2], the softmax-over-n statistics of k), dmh_linattn_apply;  then
 * dmh_linattn_backward(dout [B][n][128]) -> dqkv [B][n][384].  work: dmh_linattn_bwd_workspace_floats(B, n) floats. */
int dmh_linattn_merge_ms(const float* partial, float* ctx, float* ms, int B, int n, void* stream);
int64_t dmh_linattn_bwd_workspace_floats(int B, int n);
int dmh_linattn_backward(const float* qkv, const float* ctx, const float* ms, const float* dout, float* dqkv, float* work,
                         int B, int n, float scale, void* stream);

/* small strided batched fp32 GEMM (exact fp32 MFMA) for the backward pass: C[bo][bi] = alpha * A[bo][bi] (MxK) * B[bo][bi]
 * (KxN); sa / sb / sc: HOST arrays of 4 strides in floats {outer batch, inner batch, row, column}.  Row softmax and its
 * backward (dP <- P * (dP - sum_j dP*P), in place) for the bottleneck Attention (CFG:287-295). */
int dmh_bgemm(const float* A, const int64_t* sa, const float* B, const int64_t* sb, float* C, const int64_t* sc, int M,
              int N, int K, int nbo, int nbi, float alpha, void* stream);
int dmh_softmax_rows(const float* S, float* P, int64_t rows, int n, void* stream);
int dmh_softmax_rows_backward(const float* P, float* dP, int64_t rows, int n, void* stream);

/* elementwise SiLU (mode 1) / exact GELU (mode 2): dy == NULL -> out = f(x), else out = dy * f'(x) */
int dmh_act(const float* x, const float* dy, float* out, int64_t n, int mode, void* stream);
/* backward of the class embedding lookup + null-class select (CFG:419-425): d [B][D] -> dtable [num_classes][D], dnull [D];
 * keep: uint8 [B] (the mask drawn in the forward) */
int dmh_class_embed_backward(const float* d, const int64_t* classes, const unsigned char* keep, float* dtable, float* dnull,
                             int B, int D, int num_classes, void* stream);
/* gradient of p_losses (CFG:796-806) wrt the UNet output: out, target [B][6][H][W]; warped = flow_warp(out[:,3:6], flow)
 * [B][3][H][W]; mask [B][1][H][W]; abar [B] = alphas_cumprod[t]; squared: 0 L1, 1 L2.  dout [B][6][H][W]; gD [B][3][H][W]
 * work.  The transpose of the bilinear gather uses float atomics (order not fixed where the flow field folds). */
int dmh_loss_backward(const float* out, const float* target, const float* warped, const float* mask, const float* flow,
                      const float* abar, float* dout, float* gD, int B, int H, int W, int squared, void* stream);
/* optimiser (DDP:1852-1862): per-tensor partial sums of squares (f64 [dmh_sumsq_blocks()] each) -> global norm and clip
 * coefficient (norm_out[0], norm_out[1] = min(1, max_norm / (norm + 1e-6))) -> Adam with the gradient scaled by
 * gscale[1] (torch.optim.Adam, no weight decay) -> EMA lerp */
int dmh_sumsq_blocks(void);
int dmh_sumsq(const float* g, int64_t n, double* part, void* stream);
int dmh_gradnorm_finalize(const double* part, int n, float max_norm, float* norm_out, void* stream);
int dmh_adam(float* p, const float* g, float* m, float* v, const float* gscale, int64_t n, float lr, float b1, float b2,
             float eps, int step, void* stream);
int dmh_ema(float* ema, const float* p, int64_t n, float decay, void* stream);
/* multi-tensor forms of the two above: p / g / m / v / n are HOST arrays of ``count`` device pointers / element counts
 * (the ~280 parameter tensors of the UNet go out in count/48 launches); part: f64 [dmh_multi_blocks(n, count)], to be
 * reduced by dmh_gradnorm_finalize */
int64_t dmh_multi_blocks(const int64_t* n, int count);
int dmh_sumsq_multi(const float* const* g, const int64_t* n, int count, double* part, void* stream);
int dmh_adam_multi(float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* n, int count,
                   const float* gscale, float lr, float b1, float b2, float eps, int step, void* stream);


/* ---------------------------------------------------------------------------------------
 * condition dataset path: UnHomoTrainData.__getitem__ DDP:1097-1163, a batch per launch
 * ------------------------------------------------------------------------------------- */
/* cv2.resize(img.astype(float32) / div, (Wd, Hd)) with INTER_LINEAR (DDP:1118-1123): src [B][Hs][Ws][C] uint8 interleaved
 * (as decoded), dst plane c of image b at dst + b*dst_bstride + c*Hd*Wd (the batch tensor's own planes) */
int dmh_resize_bilinear_u8(const unsigned char* src, float* dst, int B, int Hs, int Ws, int C, int Hd, int Wd,
                           int64_t dst_bstride, float div, void* stream);
/* cv2.dilate(cv2.erode(cv2.resize(mask, (Wd, Hd), INTER_NEAREST), ones(3,3)), ones(3,3)) (DDP:1129-1133):
 * src [B][Hs][Ws] fp32, dst plane of image b at dst + b*dst_bstride */
int dmh_mask_open_nearest(const float* src, float* dst, int B, int Hs, int Ws, int Hd, int Wd, int64_t dst_bstride,
                          void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DMHOMO_HIP_H */

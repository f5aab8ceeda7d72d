/* glowtts_hip.h - C ABI of libglowtts_hip.so (gfx950 / MI355X).
 *
 * The drop-in boundary of the MI355X-native Glow-TTS hot path.  Plain pointers and sizes only:
 * every pointer is a DEVICE pointer unless stated otherwise, `stream` is a hipStream_t passed as
 * void* (NULL = default stream), every function is asynchronous on `stream` and returns 0 on
 * success or a negative GLOWTTS_E_* code (no exceptions cross the boundary).
 *
 * The reference (CODEJIN/Glow_TTS) has no FFI layer; each entry point cites the reference
 * interface it replaces (file:line relative to the reference repository).  INTEGRATION.md shows
 * the binding a reference maintainer would add.
 */
#ifndef GLOWTTS_HIP_H
#define GLOWTTS_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GLOWTTS_OK            0
#define GLOWTTS_E_ARG        -1   /* bad argument / unsupported size */
#define GLOWTTS_E_LAUNCH     -2   /* hip launch error */
#define GLOWTTS_ABI_VERSION    7

/* Library / device identification.  Returns the ABI version (currently 7: glowtts_prior_loss, glowtts_dur_proj_*, glowtts_prior_split_*, the `path` argument of
 * glowtts_expand_pair_targets, the `da_unit` argument of glowtts_mse_loss_fwd, glowtts_gate_bwd_io with out == NULL, glowtts_fx_to_float and glowtts_flow_grads.dcond typed int64_t *; 6: glowtts_cond_linear_supported, the direct 3x3 stride-2 conv trio glowtts_conv3x3s2_*; 5: additions only - glowtts_cond_linear_fwd / _bwd, glowtts_prep_launch_dev,
 * glowtts_rpr_attention_bwd_partial_rows (and NULL drelk / drelv), glowtts_sum_slices / _seg, GLOWTTS_F_GATE_IN0, GLOWTTS_F_COND_FX, glowtts_flow_acts.skip may be NULL on
 * the fused forward launch, glowtts_flow_grads.dcond holds 64-bit fixed-point accumulators; 4: glowtts_flow_acts grew next_* / actnorm_done - the next flow's ActNorm + 1x1 conv in the
 * fused coupling launch's epilogue - and glowtts_proj_layernorm / glowtts_layernorm_qkv were added; 3: glowtts_prep_job / glowtts_prep_launch / glowtts_wavenet_prep_jobs,
 * glowtts_actnorm_inv1x1_pass_bf, glowtts_flow_acts.xa_bf, glowtts_flow_grads.dh0_bf16, GLOWTTS_WIO_DMA; 2: glowtts_flow_params grew wn_img / wn_img_t; round 2's additions to
 * glowtts_mle_loss_bwd, glowtts_flow_params.cond_rows and glowtts_flow_grads.pitch_rows belong to version 2 as well). */
int glowtts_abi_version(void);
/* Writes the gfx arch string of device 0 into buf (host pointer).  0 on success. */
int glowtts_device_arch(char *buf, int buflen);

/* Launch log (diagnostics; no reference counterpart).  The library counts, on the host, the launches it issues per kernel class
 * ("conv_dma<GATE,5>", "conv_chain<LINEAR,DGATE>", "conv_cl<LINEAR,1,f32>", "wgrad<5,bf16,dybf16,xbf16,wide>/grouped", ...), so a
 * parity test can assert which kernel served a call.  glowtts_launch_count sums every class whose name starts with `kernel_class`
 * (host string); _dump writes "name count" lines into buf (host pointer). */
int64_t glowtts_launch_count(const char *kernel_class);
void glowtts_launch_log_reset(void);
int glowtts_launch_log_dump(char *buf, int buflen);

/* Sustained matrix clock (measurement aid for bench.py's roofline; no reference counterpart).  Launches `nwg` workgroups of four waves
 * (one per SIMD), each issuing 4 * iters back-to-back v_mfma_f32_32x32x16_bf16 on pseudo-random operands; workgroup i writes the shader
 * cycles (s_memtime) and the constant-rate wall ticks its loop took to out[2i], out[2i+1] (device pointer, 2 * nwg int64).  *wall_khz
 * (host pointer, optional) receives the wall counter's rate.  cycles / ticks * rate = the clock the chip holds under a dense bf16 MFMA
 * load, which on MI355X is well below the 2.4 GHz the 2.5 PFLOP/s datasheet peak assumes. */
int glowtts_mfma_clock_probe(long long *out, int nwg, int iters, int *wall_khz, void *stream);
/* Diagnostics: a one-thread kernel that writes the constant-rate wall counter (the rate glowtts_mfma_clock_probe reports) to *slot when the
 * stream reaches it - a timeline of a replayed hipGraph from inside the graph (tools/step_timeline.py). */
int glowtts_debug_stamp(long long *slot, void *stream);
/* The dropout seed word of a training step (ABI 6; Modules.py:481, 561-569, 862: torch's Dropout draws from the global generator).  state (device, 2 words:
 * counter, base - the base drawn once from torch's generator) -> counter += 1, out[0] = a 31-bit hash of (base, counter): every dropout kernel of the step adds its own
 * layer constant to this word (the `seed_ptr` arguments).  A launch of its own instead of torch.randint: inside a replayed hipGraph the philox draw costs two fills of
 * the generator's offset words in front of EVERY replay and an RNG launch on each of the two streams' chains. */
int glowtts_step_seed(uint32_t *state, uint32_t *out, void *stream);

/* ------------------------------------------------------------------------------------------
 * Monotonic Alignment Search.
 * Replaces monotonic_align/core.pyx:40 `maximum_path_c(paths, values, t_xs, t_ys, max_neg_val)`
 * (and the pure-Python twin Modules.py:951-980).  Bit-exact with core.pyx:9-35 on identical fp32 input.
 *
 *   value    [B][Tx][Ty] f32, already multiplied by the mask (monotonic_align/__init__.py:11).
 *            Read-only unless q_out == value (then clobbered into cumulative scores like core.pyx:30).
 *   t_xs,t_ys[B] i32      valid tokens / frames per utterance, 1 <= t_x <= Tx, 0 <= t_y <= Ty (else: all-zero path, idx = -1).
 *            t_x > t_y (no monotonic alignment exists) is reproduced as core.pyx behaves: nothing is accumulated and the
 *            backtrack walks the raw inputs from row t_x - 1.
 *   idx_out  [B][Ty] i32  token index aligned to each frame, -1 for y >= t_y.           (may be NULL)
 *   q_out    [B][Tx][Ty] f32 cumulative scores exactly as core.pyx leaves `values`      (may be NULL)
 *   Tx <= 512.
 */
int glowtts_mas_dp_f32(const float *value, const int32_t *t_xs, const int32_t *t_ys,
                       int32_t *idx_out, float *q_out, int B, int Tx, int Ty,
                       float max_neg_val, void *stream);
/* Same search on the transposed score matrix value_t [B][Ty][Tx] (token index contiguous) - the layout the
 * fused log-prior GEMM writes, where one frame of scores is one coalesced row.  q_out_t likewise transposed.
 * 64 < Tx <= 128 with Tx even and value_t 16-byte aligned (the training shapes) takes the hand-scheduled kernel of
 * csrc/mas_dp2.hip; every other shape the general one of csrc/mas.hip.  Same results bit for bit. */
int glowtts_mas_dp_f32_t(const float *value_t, const int32_t *t_xs, const int32_t *t_ys,
                         int32_t *idx_out, float *q_out_t, int B, int Tx, int Ty,
                         float max_neg_val, void *stream);
/* Dense 0/1 path from idx (core.pyx:32-35 writes these ones into a pre-zeroed array; here every
 * element is written, so `path` need not be zeroed).  out_dtype: 0 = int32, 1 = float32. */
int glowtts_mas_path_from_idx(const int32_t *idx, void *path, int B, int Tx, int Ty,
                              int out_dtype, void *stream);
/* Both steps: the direct replacement of maximum_path_c.  `scratch_idx` [B][Ty] i32. */
int glowtts_mas_f32(const float *value, int32_t *path, const int32_t *t_xs, const int32_t *t_ys,
                    int32_t *scratch_idx, int B, int Tx, int Ty, float max_neg_val, void *stream);
/* HOST twin (no stream, every pointer is a HOST pointer): exactly `maximum_path_c` of core.pyx:40 - `value` [B][Tx][Ty] is clobbered into
 * the cumulative scores, `path` must arrive zeroed; an utterance with t_x > t_y (no monotonic alignment exists) is treated as core.pyx
 * treats it: nothing is accumulated into `value`, the backtrack walks the RAW scores from row t_x - 1 and writes its ones into `path` (same
 * as the device entry points above).  num_threads <= 0: one thread per
 * hardware thread (core.pyx:44 `prange`).  An explicit entry point for host-resident score matrices, never a fallback of the GPU path. */
int glowtts_mas_f32_host(float *value, int32_t *path, const int32_t *t_xs, const int32_t *t_ys,
                         int B, int Tx, int Ty, float max_neg_val, int num_threads);


/* ------------------------------------------------------------------------------------------
 * Channels-last implicit-GEMM convolution on MFMA (the dense contractions of the path):
 *     Y[r][n] = epilogue( sum_{t<taps} sum_{c<ca} A[r + t - pad][c] * W[n][c][t] )
 * Replaces every torch.nn.Conv1d / conv2d / bmm call site on the hot path
 * (Modules.py:791 Start, :861 In_i k=5, :871 Res_Skip_i, :793 End, and their autograd
 * transposes; encoder convs Modules.py:484,565,568 and RPR_MHA.py:82-84,93).
 *
 * Layout: activations are fp32 "rows x channels" (channels contiguous).  A row is one (utterance,
 * frame); utterances are laid out back to back with >= pad zero rows between them so that the taps
 * of one utterance never read another's frames (rows < 0 or >= rows read as zero).
 * precision: GLOWTTS_F32 uses v_mfma_f32_32x32x2_f32 (exact fp32, the reference's arithmetic),
 *            GLOWTTS_BF16 rounds operands to bf16 (v_mfma_f32_32x32x16_bf16), fp32 accumulate.
 */
#define GLOWTTS_F32   0
#define GLOWTTS_BF16  1

/* weight permutations applied by glowtts_pack_weight to the packed N (or K, when transposed) index */
#define GLOWTTS_PERM_NONE 0
#define GLOWTTS_PERM_PAIR 1   /* packed p*64 + h*32 + j  <->  original h*H + (p*32 + j): halves interleaved per 32 */

/* Packs W_eff fp32 [O][I][taps] (torch Conv1d weight layout) into MFMA tile order
 * [tap][kchunk][npad][KC] in the requested precision.
 *   transpose = 0: N = O, K = I                       (forward)
 *   transpose = 1: N = I, K = O, taps flipped          (data gradient)
 *   perm/perm_h : permutation of the O index (PAIR for gate / coupling halves), applied to whichever
 *                 of N / K is O.  npad_out / kchunks_out (host pointers, may be NULL) receive the
 *                 padded sizes; `packed` must hold taps*kchunks*npad*64 bytes
 *                 (query with packed == NULL). */
int glowtts_pack_weight(const float *w, int O, int I, int taps, int transpose, int perm, int perm_h,
                        int precision, void *packed, int *npad_out, int *kchunks_out, void *stream);

/* Many weights of different shapes packed by ONE launch (the encoder's convs, forward and transposed: ~70 launches of
 * glowtts_pack_weight per training step otherwise).  Jobs live in device memory; fill them on the host with glowtts_pack_job_init
 * (same arguments as glowtts_pack_weight; block0 = running sum of the previous jobs' *blocks_out), copy the table to the device
 * once - it stays valid while the weight and packed pointers do - and launch with total_blocks = the final running sum. */
typedef struct glowtts_pack_job {
    const float *w; void *packed;
    int O, I, taps, transpose, perm, perm_h;
    int N, K, npad, kchunks;               /* derived (glowtts_pack_job_init) */
    int block0, reserved;
} glowtts_pack_job;
int glowtts_pack_job_init(glowtts_pack_job *job, const float *w, int O, int I, int taps, int transpose, int perm, int perm_h,
                          int precision, void *packed, int block0, int *blocks_out, int64_t *bytes_out);
int glowtts_pack_weight_multi(const glowtts_pack_job *dev_jobs, int njobs, int total_blocks, int precision, void *stream);

/* `batch` independent weights of identical shape, w [batch][O][I][taps] -> packed [batch][taps*kchunks*npad*64 bytes] */
int glowtts_pack_weight_batched(const float *w, int batch, int O, int I, int taps, int transpose, int perm, int perm_h,
                                int precision, void *packed, int *npad_out, int *kchunks_out, void *stream);

/* The same with a two-level destination: weight b (0 <= b < batch) is written at packed + (b / inner) * outer_stride + (b % inner) *
 * inner_stride (bytes, may be negative) - how glowtts_wavenet_pack_images places every conv of a flow inside that flow's weight image.
 * w_stride: elements between consecutive source weights (0 = O * I * taps; larger: each weight is the leading [O] slice of a bigger tensor). */
int glowtts_pack_weight_strided(const float *w, int batch, int inner, int O, int I, int taps, int transpose, int perm, int perm_h,
                                int precision, void *packed, int64_t outer_stride, int64_t inner_stride, int64_t w_stride, void *stream);

/* Weight preparation of a training step in ONE launch (round 4; replaces glowtts_weightnorm_fwd + the glowtts_pack_weight_* launches of the decoder:
 * Modules.py:766, 818, 825 weight_norm + the tile images of every conv).  A job = one glowtts_pack_weight_strided call whose source is the
 * weight-norm pair (v [batch][O][I][taps], g [batch][O]) instead of w: w = g * v / ||v|| (norm over (I, taps) per output channel, old-style
 * torch weight_norm) is formed on the fly in fp32, rounded to bf16 exactly as the two-step path rounds it, and never written.  g == NULL: v is a
 * plain weight.  inv_out (optional) [batch][O] receives 1 / ||v|| for glowtts_weightnorm_bwd.  g_stride: elements between consecutive convs'
 * g / inv_out rows (0 = O; larger when a job covers the leading [O] slice of a bigger conv).  bf16 images only.  I * taps <= 1024 (plain weights: 4096).
 * Jobs are a HOST array (fill with glowtts_prep_job_init, block0 = running sum of *blocks_out; at most GLOWTTS_PREP_MAX_JOBS per launch): the launch
 * carries the table in its argument segment, so a captured step needs no copy node for it. */
#define GLOWTTS_PREP_MAX_JOBS 22
typedef struct glowtts_prep_job {
    const float *v; const float *g; float *inv_out; void *packed;
    int64_t outer_stride, inner_stride, w_stride, g_stride;
    int batch, inner, O, I, taps, transpose, perm, perm_h;
    int o_ext, npad, kchunks, tiles;       /* derived (glowtts_prep_job_init) */
    int block0, reserved;
    /* ABI 7: twin output (glowtts_prep_jobs_twin_in): a forward PAIR-packed job whose tiles ALSO write the transposed half images of the same rows - the pairs
     * are read once for all three images.  twin == NULL: none */
    void *twin; int64_t twin_outer, twin_inner, twin_half;
    int twin_batch, twin_npad;
} glowtts_prep_job;
/* Folds the two transposed In_l half jobs of a table built by glowtts_wavenet_prep_jobs (forward images, then the backward images of the FIRST F_bwd flows)
 * into the forward In_l job as its twin output, removes them and renumbers block0.  No-op (returns GLOWTTS_OK) when the pattern is not found. */
int glowtts_prep_jobs_twin_in(glowtts_prep_job *jobs /* host */, int *njobs, int *blocks);
int glowtts_prep_job_init(glowtts_prep_job *job /* host */, const float *v, const float *g, float *inv_out, int batch, int inner, int O, int I, int taps,
                          int transpose, int perm, int perm_h, void *packed, int64_t outer_stride, int64_t inner_stride, int64_t w_stride,
                          int64_t g_stride, int block0, int *blocks_out /* host */);
/* max_cols = the largest I * taps among the jobs (sizes the LDS tile) */
int glowtts_prep_launch(const glowtts_prep_job *host_jobs, int njobs, int total_blocks, int max_cols, void *stream);
/* Round 5: the same kernel over a job table in DEVICE memory, any number of jobs - for tables that do not change between steps (the text encoder's ~60
 * images of plain conv weights: built and uploaded once).  Measured against glowtts_pack_weight_multi's element-wise gather on those images: 47 vs 33 us
 * alone - the host library keeps the gather and no longer calls this entry point (round 6). */
int glowtts_prep_launch_dev(const glowtts_prep_job *dev_jobs, int njobs, int total_blocks, int max_cols, void *stream);

#define GLOWTTS_APRO_NONE    0
#define GLOWTTS_APRO_PAIRMUL 1  /* a[r][c] = A[r][2c] * A[r][2c+1]   (tanh*sigmoid gates, Modules.py:885-887) */
#define GLOWTTS_APRO_SQNEG   2  /* a[r][c] = c < ca1 ? -0.5 * A[r][c]^2 : A[r][c - ca1]   (log-prior operand, Modules.py:112-113) */

#define GLOWTTS_EPI_LINEAR   0  /* v = acc (+bias[n]) (relu) (+in0[r][n]) (*rowmask[r]) (+= out0) -> out0[r][n]      */
#define GLOWTTS_EPI_GATE     1  /* PAIR-packed cols: out0[r][2j],[2j+1] = tanh(a), sigmoid(s)   Modules.py:861-870   */
#define GLOWTTS_EPI_RESSKIP  2  /* n<h: out0 = (in0+acc+b)*mask ; n>=h: out1 (+)= acc+b         Modules.py:871-881   */
#define GLOWTTS_EPI_COUPLE   3  /* PAIR-packed (m,logs): out0 = x_b' ; out1 = (m,logs) kept     Modules.py:795-806   */
#define GLOWTTS_EPI_DGATE    4  /* dacts -> (da, ds) through tanh/sigmoid, PAIR-packed out0      (autograd of :885-887) */

#define GLOWTTS_F_BIAS     1
#define GLOWTTS_F_RELU     2
#define GLOWTTS_F_ADD_IN0  4
#define GLOWTTS_F_MASK     8
#define GLOWTTS_F_ACCUM   16
#define GLOWTTS_F_FIRST   32   /* RESSKIP: skip accumulator is written, not accumulated */
#define GLOWTTS_F_LAST    64   /* RESSKIP: last WaveNet layer (n = h outputs, all skip, *mask) */
#define GLOWTTS_F_REVERSE 128  /* COUPLE: inverse coupling x_b = (x_b - m) * exp(-logs) * mask */
#define GLOWTTS_IO_A_BF16    1
#define GLOWTTS_IO_IN0_BF16  2
#define GLOWTTS_IO_OUT0_BF16 4
#define GLOWTTS_F_DROPOUT 512  /* LINEAR: dropout(p = drop_p, seed) after the optional ReLU, before residual / mask */
#define GLOWTTS_F_COLMASK 256  /* LINEAR: zero columns n >= ncols_valid[batch]  (attention mask, Modules.py:102) */
#define GLOWTTS_F_COND_FX 4096  /* DGATE (round 5): out1 is an array of int64 FIXED-POINT accumulators (units of 2^-40, same shape and ld1 in elements) instead of
                                  * floats; the per-utterance sums are added with 64-bit integer atomics - order-independent, bit-reproducible.  value = (double)acc * 2^-40 */
#define GLOWTTS_F_GATE_IN0 2048 /* LINEAR (round 5): in0 is the KEPT OUTPUT of a relu / dropout layer and gates the result instead of being added:
                                  * out = value * (in0 != 0 ? 1 / (1 - drop_p) : 0) - the backward of relu and dropout (Modules.py:566-567) in the epilogue of
                                  * the data-gradient conv that feeds it (was a pass of its own: glowtts_gate_bwd).  Not with GLOWTTS_F_ADD_IN0 / _DROPOUT. */
#define GLOWTTS_F_COND_ROWS 1024 /* GATE: `cond` holds one row per ACTIVATION row, [rows][ldcond] (per-frame conditioning: the GR-mode
                                  * pitch term, Modules.py:867-869, summed with the per-utterance speaker / prosody terms by the caller) */

typedef struct glowtts_conv_args {
    const float *a;  int64_t lda;      /* A rows (floats per row = lda) */
    const float *a2; int64_t lda2;     /* optional second source for channels >= ca1 */
    int ca1, ca;                       /* channels taken from a / total K per tap */
    int apro;                          /* GLOWTTS_APRO_* */
    int rows;                          /* number of rows R */
    const void *w;                     /* packed weights */
    int n, npad, kchunks;              /* logical columns, padded columns, K chunks (from pack) */
    int taps, pad;
    int precision;                     /* GLOWTTS_F32 / GLOWTTS_BF16 */
    int epi, flags;
    int h;                             /* half size for PAIR epilogues / RESSKIP split */
    int rows_per_utt;                  /* rows per utterance (for cond lookup) */
    const float *bias;                 /* [n] original (un-permuted) order */
    const float *rowmask;              /* [rows] 1 = valid frame */
    const float *cond; int64_t ldcond; /* GATE: per-utterance conditioning [B][ldcond], original order */
    float *out0; int64_t ld0;
    float *out1; int64_t ld1;          /* (DGATE: optional d conditioning [B][ld1], original order, accumulated with atomic adds BEFORE the
                                        *  dropout mask is applied to the gate gradients - zero it first) */
    const float *in0; int64_t ldi0;
    /* batched problems (gridDim.z): `rows` rows per problem; element strides between problems (0 = shared) */
    int batch;
    int64_t a_bstride, w_bstride /* bytes */, bias_bstride, out_bstride, mask_bstride;
    const int32_t *ncols_valid;        /* [batch] for GLOWTTS_F_COLMASK */
    /* GATE / DGATE: dropout on the conv output before the conditioning is added (Modules.py:861-862), p = drop_p.
     * The keep mask is a counter hash of (seed, row, channel): the backward regenerates it from the same seed. */
    uint32_t seed; float drop_p;
    const uint32_t *seed_ptr;          /* optional DEVICE word added to `seed` (lets a captured hipGraph draw new masks per replay) */
    /* bf16 activation storage (GLOWTTS_BF16 precision only): which of the `const float*` tensors actually hold bf16 elements
     * (strides then count bf16 elements).  A_BF16: a / a2 (GATE, RESSKIP+PAIRMUL, LINEAR only; ca, lda multiples of 8);
     * IN0_BF16: in0 (RESSKIP residual, DGATE gates); OUT0_BF16: out0 (GATE gates, RESSKIP state, DGATE gate gradients). */
    int io_flags;
} glowtts_conv_args;

/* Two chained 1x1 convs over the same rows in ONE launch (bf16 precision, bf16-stored `first->a`, 192 intermediate channels): the first
 * conv's output is handed to the second through LDS.  Supported pairs: (RESSKIP with F_LAST -> COUPLE): first->out1 = fp32 skip sum of the
 * earlier layers (read), first->out0 = bf16 final skip rows (written; the A operand `second->a` is ignored); (LINEAR with F_MASK -> DGATE):
 * first->out0 = bf16 d(skip) rows (written).  Everything else as in glowtts_conv_cl for the two epilogues. */
int glowtts_conv_chain(const glowtts_conv_args *first, const glowtts_conv_args *second, void *stream);
int glowtts_conv_cl(const glowtts_conv_args *args /* host pointer */, void *stream);

/* ------------------------------------------------------------------------------------------
 * Flow-decoder elementwise / reduction steps ("rows" layout).
 * rows tensors are fp32 [B][Tp][C], Tp = T + 2*GLOWTTS_ROW_PAD, channels contiguous; the first and last
 * GLOWTTS_ROW_PAD rows of every utterance are zero so the k=5 taps never cross utterances.
 * rowmask [B*Tp] is 1 for valid squeezed frames (Modules.py:903), 0 for padding and pad rows.
 */
#define GLOWTTS_ROW_PAD 2

/* Squeeze (Modules.py:895-907): mel [B][Cm][Tm] -> rows [B][Tp][ns*Cm], rows[b][PAD+t][s*Cm+c] = mel[b][c][ns*t+s]*mask;
 * also writes rowmask (may be NULL).  T = Tm / ns (an odd tail frame is dropped like :897-898). lengths: i64 [B]. */
int glowtts_squeeze_rows(const float *mel, float *rows, float *rowmask, const int64_t *lengths,
                         int B, int Cm, int Tm, int ns, void *stream);
/* Unsqueeze (Modules.py:914-924) back to [B][Cm][Tm] (* mask); with use_fill the masked frames are set to `fill`
 * (GlowTTS.inference masked_fill_, Modules.py:202). */
int glowtts_unsqueeze_rows(const float *rows, float *mel, const int64_t *lengths,
                           int B, int Cm, int Tm, int ns, int use_fill, float fill, void *stream);
/* Per-flow 4x4 inverse and log|det| (torch.inverse / torch.logdet, Modules.py:743,747).
 * W [F][4][4] -> winfo [F][36] = { W[16], W^-1[16], logdet, sign, 0, 0 }. */
int glowtts_inv1x1_prepare(const float *W, float *winfo, int F, void *stream);
/* ActNorm (Modules.py:689-694) + invertible 1x1 conv (Modules.py:738-756) in one pass over the rows.
 * reverse = 0: xout = W (bias + exp(logs) xin) ; reverse = 1: xout = (W^-1 xin - bias) exp(-logs).  In-place allowed. */
int glowtts_actnorm_inv1x1(const float *xin, float *xout, const float *logs, const float *bias,
                           const float *winfo, const float *rowmask, int64_t rows, int C, int reverse, void *stream);
/* forward form that also writes the first C/2 output channels to xpass (the coupling layer's pass-through half, Modules.py:808) */
/* ... and (xa_bf != NULL) a bf16 copy of the first C/2 output channels [rows][C/2] (glowtts_flow_acts.xa_bf) */
int glowtts_actnorm_inv1x1_pass_bf(const float *xin, float *xout, float *xpass, void *xa_bf, const float *logs, const float *bias, const float *winfo,
                                   const float *rowmask, int64_t rows, int C, void *stream);
int glowtts_actnorm_inv1x1_pass(const float *xin, float *xout, float *xpass, const float *logs, const float *bias, const float *winfo,
                                const float *rowmask, int64_t rows, int C, void *stream);
/* ActNorm data-dependent init statistics (Modules.py:698-703): stats [2C+1] = { sum x*m [C], sum x^2*m [C], sum m }.
 * Deterministic two-stage reduction; scratch holds glowtts_actnorm_stats_scratch_floats(rows, C) floats.
 * Under data parallelism the caller all-reduces `stats` before glowtts_actnorm_from_stats. */
int glowtts_actnorm_stats(const float *x, const float *rowmask, float *stats, float *scratch,
                          int64_t rows, int C, void *stream);
int64_t glowtts_actnorm_stats_scratch_floats(int64_t rows, int C);
/* logs = -0.5 log(max(var, 1e-7)), bias = -mean exp(logs)   (Modules.py:704-711) */
int glowtts_actnorm_from_stats(const float *stats, float *logs, float *bias, int C, void *stream);
/* Backward of the affine coupling transform (autograd of Modules.py:805-806).  dz [rows][C] in/out (x_b half becomes
 * d x_b), xmid = coupling input, outs/douts PAIR-packed (m, logs) [rows][ldo], dlogdet [B] = dL/dlogdet. */
int glowtts_coupling_bwd(float *dz, const float *xmid, const float *outs, float *douts, const float *rowmask,
                         const float *dlogdet, int64_t rows, int C, int ldo, int rows_per_utt, void *stream);
/* same, and a second bf16 copy of douts (act_bf16 mode: the A operand of the End conv's data gradient on the LDS-DMA / chained path) */
int glowtts_coupling_bwd_bf16(float *dz, const float *xmid, const float *outs, float *douts, void *douts_bf16, const float *rowmask,
                              const float *dlogdet, int64_t rows, int C, int ldo, int rows_per_utt, void *stream);
int glowtts_fill_zero(float *p, int64_t n, void *stream);
/* Backward of inv-1x1 + ActNorm (autograd of Modules.py:693,749-756).  dz -> dx (may alias), x = flow input.
 * param_grads [2C+16] = { dlogs[C], dbias[C], dW[16] } (data terms only; the log-det terms are added by the caller).
 * scratch: glowtts_actnorm_stats_scratch_floats(rows, C) floats. */
int glowtts_actnorm_inv1x1_bwd(const float *dz, float *dx, const float *x, const float *logs, const float *bias,
                               const float *winfo, const float *rowmask, float *param_grads, float *scratch,
                               int64_t rows, int C, void *stream);
/* number of per-block partial rows glowtts_actnorm_inv1x1_bwd leaves in scratch (each 2C+16 floats) when param_grads == NULL */
int64_t glowtts_actnorm_bwd_blocks(int64_t rows);
/* glowtts_actnorm_inv1x1_bwd (param_grads = NULL form) of flow f FUSED with glowtts_coupling_bwd(_bf16) of flow f-1: dx leaves as the
 * gradient of flow f-1's coupling input, prev_douts (+ optional bf16 copy) are filled.  One launch and one pass over the rows fewer per flow. */
int glowtts_actnorm_inv1x1_bwd_coupling(const float *dz, float *dx, const float *x, const float *logs, const float *bias, const float *winfo,
                                        const float *rowmask, float *scratch, int64_t rows, int C,
                                        const float *prev_xmid, const float *prev_outs, float *prev_douts, void *prev_douts_bf16,
                                        const float *dlogdet, int ldo, int rows_per_utt, void *stream);
/* out[b][i] = sum_r partial[b][r][i], i < n, r < nrows (deterministic): reduces the per-block partials of several
 * glowtts_actnorm_inv1x1_bwd calls made with param_grads = NULL (their `scratch` buffers, part_stride floats apart) in one launch */
int glowtts_colsum_batched(const float *partial, float *out, int nrows, int n, int batch, int64_t part_stride, int64_t out_stride, void *stream);
/* Parameter gradients of ActNorm / inv-1x1 of all F flows from the reduced data terms d_an [F][2C+16] (glowtts_colsum_batched) plus the
 * log-determinant terms of Modules.py:694, 747: dlogs [F][C], dbias [F][C], dw [F][4][4]; rowmask [B][Tp], dlogdet [B], winfo [F][36]. */
int glowtts_decoder_param_grads(const float *d_an, const float *dlogdet, const float *rowmask, const float *winfo,
                                float *dlogs, float *dbias, float *dw, int F, int B, int Tp, int C, void *stream);
/* Decoder log-determinant (Modules.py:309): logdet[b] = sum_f [ (sum logs_f + logdet W_f * C/4) * len_b + sum logs^coupling ].
 * outs_all: the F kept (m, logs) buffers, flow_stride floats apart; part [F*B] scratch. */
int glowtts_decoder_logdet(const float *outs_all, int64_t flow_stride, const float *logs_all, const float *winfo_all,
                           const float *rowmask, float *part, float *logdet, int F, int B, int Tp, int C, int ldo, void *stream);

/* ------------------------------------------------------------------------------------------
 * Weight / bias gradient of the channels-last convolution (autograd of the conv call sites above):
 *     dW[o][c][t] (+)= sum_r DY[r][o] * X[r + t - pad][c]        dbias[o] (+)= sum_r DY[r][o]
 * DY columns may be PAIR-packed (gate / coupling buffers): perm maps packed column -> o.
 * dW is written in torch Conv1d layout [O][ca][taps] fp32.  With splits > 1 (or splits = 0: automatic) or
 * accumulate != 0 the result is ADDED with fp32 atomics: the caller zeroes dW / dbias first.
 */
typedef struct glowtts_wgrad_args {
    const float *dy; int64_t lddy;     /* [rows][lddy], columns [0, m) used */
    const float *x;  int64_t ldx;      /* [rows][ldx] */
    int xpro;                          /* GLOWTTS_APRO_NONE / GLOWTTS_APRO_PAIRMUL */
    const float *xmask;                /* optional [rows] multiplier on X rows */
    int rows, m, ca, taps, pad;
    int perm, perm_h;                  /* DY column permutation */
    int precision;
    int splits, accumulate;
    float *dw;                         /* [O][ca][taps] */
    float *dbias;                      /* [O] or NULL */
    int io_flags;                      /* GLOWTTS_WIO_*: dy / x hold bf16 elements (bf16 precision only; strides count elements).
                                        * Supported: DY|X with no prologue, X alone with or without PAIRMUL. */
} glowtts_wgrad_args;
#define GLOWTTS_WIO_DY_BF16 1
#define GLOWTTS_WIO_X_BF16  2
#define GLOWTTS_WIO_WIDE    4   /* bf16 precision, no prologue, both operands stored alike (both bf16 or both fp32): the caller promises m, ca, lddy,
                                  ldx multiples of 8 and 16-byte aligned dy / x for EVERY job: operands are then staged 8 channels per item */
#define GLOWTTS_WIO_DMA     8   /* glowtts_wgrad_grouped_io only (ABI 3): the LDS-DMA kernel (192 (o) x 64 (c) x taps tiles; 192 x 192 at one tap; edge tiles are
                                 * masked on store); on top of WIDE with both operands bf16 (GLOWTTS_WIO_DY_BF16 | GLOWTTS_WIO_X_BF16) the caller promises for EVERY job: m, ca, lddy, ldx
                                 * multiples of 8, dy / x 16-byte aligned, each operand below 2 GiB, and (mt, nt) / tile0 / total_tiles counted in THAT tiling - mt = ceil(m / 192),
                                 * nt = ceil(ca / 64) (ceil(ca / 192) at one tap); precision bf16, no prologue, pad = (taps - 1) / 2, splits = 1, no accumulation (else GLOWTTS_E_ARG) */
int glowtts_wgrad_cl(const glowtts_wgrad_args *args /* host pointer */, void *stream);

/* Grouped form: many weight-gradient problems that share (rows, taps, pad, precision) in ONE launch, so that the
 * chip is filled by output tiles of different layers instead of by split-K over rows (no atomics, deterministic).
 * `dev_jobs` is a DEVICE array; tile0 = running sum of mt*nt over the preceding jobs (mt = ceil(m/128), nt = ceil(ca/64)),
 * total_tiles = sum over all jobs. */
typedef struct glowtts_wgrad_job {
    const float *dy; const float *x; const float *xmask; float *dw; float *dbias;
    int64_t lddy, ldx;
    int m, ca, xpro, perm, perm_h;
    int tile0, mt, nt;
    int rows;                          /* ABI 7 (was reserved, 0): > 0 = this job's own row count (glowtts_wgrad_grouped_phased), 0 = the launch's */
    int reserved;
} glowtts_wgrad_job;
/* every job of one launch uses the same X prologue `xpro` (job.xpro is ignored); m and ca must be multiples of 4 */
int glowtts_wgrad_grouped(const glowtts_wgrad_job *dev_jobs, int njobs, int total_tiles, int rows, int taps, int pad,
                          int xpro, int precision, int splits, int accumulate, void *stream);
/* same, with the storage types of DY / X (GLOWTTS_WIO_*) shared by every job of the launch */
int glowtts_wgrad_grouped_io(const glowtts_wgrad_job *dev_jobs, int njobs, int total_tiles, int rows, int taps, int pad,
                             int xpro, int precision, int splits, int accumulate, int io_flags, void *stream);

/* ABI 7.  The LDS-DMA kernel (GLOWTTS_WIO_DMA's conditions: bf16 rows operands, no prologue, 192 x 64 tiles - 192 x 192 at one tap) over a job table in TWO
 * PHASES for load balance: a launch of one-workgroup-per-CU tiles that exceeds the CU count by a fraction runs a second, almost empty round (288 tiles on 256
 * CUs: 2 x 169 us).  The caller puts whole problems first (`whole_tiles` tiles, at most the CU count) and the remaining problems behind them cut into row
 * splits (job.rows = rows / S, row-offset operand pointers, partial outputs summed afterwards by glowtts_sum_slices_seg): the short tiles fill the idle CUs and
 * the tail of the round.  Workgroup -> tile mapping stays XCD-contiguous inside each phase.  A cut must fall on an utterance boundary of the rows layout (the
 * pad rows make it exact for any tap count). */
int glowtts_wgrad_grouped_phased(const glowtts_wgrad_job *dev_jobs, int njobs, int total_tiles, int whole_tiles, int rows, int taps, void *stream);

/* ------------------------------------------------------------------------------------------
 * One flow step of the decoder = Activation_Norm -> Invertible_1x1_Conv -> Affine_Coupling_Layer
 * (Modules.py:653-668 AIA), launched as one host call: forward (training, keeps the activations the
 * backward needs), inverse (inference, Modules.py:664 reversed order) and backward.
 */
#define GLOWTTS_MAX_WN_LAYERS 8

typedef struct glowtts_packed {           /* result of glowtts_pack_weight */
    const void *w; int npad, kchunks;
} glowtts_packed;

typedef struct glowtts_flow_dims {
    int B, T;          /* utterances, squeezed frames per utterance (rows per utterance = T + 2*GLOWTTS_ROW_PAD) */
    int C;             /* flow channels = Mel_Dim * Num_Squeeze            (160) */
    int H;             /* Affine_Coupling.Calc_Channels                     (192) */
    int L;             /* WaveNet.Num_Layers                                (4)   */
    int ksize;         /* WaveNet.Kernel_Size                               (5)   */
    int precision;     /* GLOWTTS_F32 / GLOWTTS_BF16 for the MFMA contractions */
    float drop_p;      /* WaveNet.Dropout_Rate in training mode, 0 in eval mode          (Modules.py:854-862) */
    uint32_t seed;     /* dropout seed of this flow step (layer l uses seed + l); same value in forward and backward */
    const uint32_t *seed_ptr;  /* optional device word added to the seed (graph replay) */
    int act_bf16;      /* 1 (bf16 precision only): the GEMM-only activations - WaveNet states hs[], gates[], gate gradients dins[] -
                        * and, in the backward, dskip and dh[l >= 1] are bf16 tensors (same shapes; halves their HBM traffic and lets
                        * the LDS-DMA conv kernel stage them).  skip, outs, dh[0], the flow variable stay fp32. */
} glowtts_flow_dims;

typedef struct glowtts_flow_params {
    const float *an_logs, *an_bias;       /* ActNorm [C]                                   Modules.py:675-680 */
    const float *winfo;                   /* [36] from glowtts_inv1x1_prepare              Modules.py:725     */
    glowtts_packed start, in[GLOWTTS_MAX_WN_LAYERS], rs[GLOWTTS_MAX_WN_LAYERS], end;          /* forward images  */
    glowtts_packed start_t, in_t[GLOWTTS_MAX_WN_LAYERS], rs_t[GLOWTTS_MAX_WN_LAYERS], end_t;  /* transposed (backward only) */
    const float *b_start, *b_in[GLOWTTS_MAX_WN_LAYERS], *b_rs[GLOWTTS_MAX_WN_LAYERS], *b_end; /* biases, original order */
    const float *cond; int64_t ldcond;    /* optional conditioning [B][ldcond]; layer l reads cond + l*2H   Modules.py:863-866 */
    int cond_rows;                        /* 1: cond is per ROW, [R][ldcond] (forward / inverse only; GR-mode pitch, Modules.py:867-869) */
    const void *wn_img, *wn_img_t;        /* optional (ABI 2): this flow's weight images for the fused coupling-network kernels
                                           * (glowtts_wavenet_pack_images); NULL: the per-conv launches above are used */
} glowtts_flow_params;

typedef struct glowtts_flow_acts {        /* rows tensors, R = B*(T+2*PAD) rows */
    float *xin;                           /* [R][C] flow input                                   (kept) */
    float *xmid;                          /* [R][C] after ActNorm + inv-1x1 = coupling input     (kept) */
    float *xout;                          /* [R][C] flow output                                         */
    float *hs[GLOWTTS_MAX_WN_LAYERS];     /* [R][H]  WaveNet state entering layer l              (kept) */
    float *gates[GLOWTTS_MAX_WN_LAYERS];  /* [R][2H] (tanh, sigmoid) interleaved                 (kept) */
    float *skip;                          /* [R][H]  sum of skip outputs * mask                  (kept) */
    float *outs;                          /* [R][ldo] PAIR-packed (m, logs), ldo = end.npad      (kept) */
    const float *rowmask;                 /* [R] */
    float *acts[GLOWTTS_MAX_WN_LAYERS];   /* act_bf16 only (else NULL): [R][H] bf16 tanh * sigmoid of layer l                (kept) */
    float *skip_bf;                       /* act_bf16 only (else NULL): [R][H] bf16 copy of `skip` (End conv / its weight gradient) (kept) */
    void *xa_bf;                          /* optional (ABI 3): [R][C/2] bf16 copy of xmid[:, :C/2] = x_a, written by the flow's ActNorm + 1x1 pass: the X operand
                                           * of the Start conv's weight gradient at half the bytes (kept) */
    /* ABI 4, glowtts_flow_forward on the fused coupling network only (params->wn_img set): when next_xmid is given, the launch's coupling epilogue also
     * applies the NEXT flow's ActNorm + invertible 1x1 conv (Modules.py:693-694, 738-756; parameters next_an_logs / next_an_bias [C], next_winfo) to the
     * rows it has just produced and writes what that flow's own pass would have written: next_xmid [R][C], the x_a half of next_xout [R][C],
     * next_xa_bf (may be NULL).  The next flow is then called with actnorm_done != 0 and skips its pass. */
    const float *next_an_logs, *next_an_bias, *next_winfo;
    float *next_xmid, *next_xout;
    void *next_xa_bf;
    int actnorm_done;
} glowtts_flow_acts;

typedef struct glowtts_flow_grads {       /* backward outputs; weight grads are ADDED (zero them first) */
    float *dx;                            /* [R][C] in: dL/dxout, out: dL/dxin (in place) */
    const float *dlogdet;                 /* [B] dL/dlogdet */
    float *douts;                         /* [R][ldo] scratch (pad columns must be zero on entry).  ABI 6: may be NULL with douts_bf given and defer_wgrad set -
                                           * the coupling backward then writes the bf16 copy alone (no reader of the fp32 rows is left on that path) */
    float *dskip;                         /* [R][H] scratch */
    float *dh[GLOWTTS_MAX_WN_LAYERS];     /* [R][H] d(WaveNet state entering layer l) * mask; may alias as a ping-pong pair
                                             (dh[l] != dh[l+1]) unless defer_wgrad, which needs them all distinct */
    float *dins[GLOWTTS_MAX_WN_LAYERS];   /* [R][ldin] PAIR-packed gate pre-activation grads (pad columns zero on entry);
                                             may all alias unless defer_wgrad */
    int defer_wgrad;                      /* 1: skip every weight-gradient launch (the caller runs glowtts_wgrad_grouped later) */
    float *scratch;                       /* glowtts_actnorm_stats_scratch_floats(R, C) floats */
    float *d_an;                          /* [2C+16] = dlogs, dbias, dW(inv-1x1) data terms (overwritten); NULL: left as per-block partials
                                           * in `scratch` ([ceil(R/64)][2C+16]) for one glowtts_colsum_batched over all flows */
    float *dw_start, *db_start;           /* [H][C/2][1], [H] */
    float *dw_in[GLOWTTS_MAX_WN_LAYERS], *db_in[GLOWTTS_MAX_WN_LAYERS];   /* [2H][H][k], [2H] */
    float *dw_rs[GLOWTTS_MAX_WN_LAYERS], *db_rs[GLOWTTS_MAX_WN_LAYERS];   /* [2H|H][H][1], [2H|H] */
    float *dw_end, *db_end;               /* [C][H][1], [C] */
    int64_t *dcond;                       /* [B][ldcond] or NULL: grad of the conditioning, ACCUMULATED (zero it first): int64 fixed-point accumulators in units of
                                           * 2^-40 (integer atomics: reproducible sums; ABI 5 - typed int64_t since ABI 7, a float buffer no longer compiles).  A non-finite
                                           * or out-of-range (|v| >= 2^21) addend POISONS its accumulator (|acc| >= 2^61); glowtts_fx_to_float turns it into NaN */
    float *douts_bf;                      /* act_bf16 only (else NULL): [R][ldo] bf16 copy of douts, scratch (End data gradient operand) */
    /* fusion across flows (backward runs flow F-1 .. 0): */
    int coupling_done;                    /* 1: the previous call already applied THIS flow's coupling backward (dx, douts, douts_bf are ready) */
    const float *prev_xmid, *prev_outs;   /* not NULL: after this flow's ActNorm / 1x1 backward, apply the coupling backward of the flow that */
    float *prev_douts, *prev_douts_bf;    /*           runs next (f-1) in the same kernel; that call must then set coupling_done */
    /* GR-mode per-frame pitch conditioning (Modules.py:846-852, 867-869): the Pitch_l conv's weight gradient is
     * sum_r dpre[r][n] * pitch[r][j] over the gate-pre-activation gradients BEFORE the dropout mask, which only exist in the gate-derivative
     * epilogue.  pitch_rows [R][pitch_ns] (the squeezed pitch in the rows layout, pitch_ns <= 2) or NULL; when given, `dcond` has
     * B + pitch_ns rows and row B + j accumulates that sum for tap j (columns as in the rows above). */
    const float *pitch_rows; int pitch_ns;
    int dh0_bf16;                         /* ABI 3: dh[0] is stored as bf16 like dh[l >= 1] (act_bf16 only): the Start conv's data gradient reads it as bf16 rows and its
                                           * weight gradient takes it as a raw bf16 operand (with acts->xa_bf as X) */
} glowtts_flow_grads;

/* training forward: xin -> xout, fills every kept buffer of `acts` */
int glowtts_flow_forward(const glowtts_flow_dims *d, const glowtts_flow_params *p, const glowtts_flow_acts *a, void *stream);
/* inference inverse: xout(in) -> xin(out).  Uses hs[0], hs[1], gates[0], skip as scratch; xmid as scratch. */
int glowtts_flow_inverse(const glowtts_flow_dims *d, const glowtts_flow_params *p, const glowtts_flow_acts *a, void *stream);
/* backward of glowtts_flow_forward */
int glowtts_flow_backward(const glowtts_flow_dims *d, const glowtts_flow_params *p, const glowtts_flow_acts *a,
                          const glowtts_flow_grads *g, void *stream);
/* ------------------------------------------------------------------------------------------
 * Fused coupling network (Modules.py:785-806 Affine_Coupling_Layer.forward with its WaveNet, :858-887): Start conv, L x [In_l k = 5 +
 * dropout + conditioning + tanh * sigmoid, Res_Skip_l + residual / skip], End conv + affine coupling in ONE launch per flow - one
 * persistent workgroup per 64-row window (64 - 4 (L - 1) valid rows, recomputed halo), the WaveNet state in LDS, every weight of the flow
 * streamed as 24-KiB slabs through an LDS ring by LDS-DMA.  bf16 precision with act_bf16, H = 192, k = 5, L <= 4, 64 < C/2 <= 96 (the
 * reference's default decoder); anything else returns GLOWTTS_E_ARG and the caller uses the per-conv launches.
 * Weight image of one flow (forward): slabs of GLOWTTS_WN_SLAB_BYTES = [384 n][64 B]:
 *   [Start: 2 slabs][layer l: In_l 30 slabs = (tap, K chunk), Res_Skip_l 6 slabs (PAIR-packed: residual | skip per 32 channels; last
 *   layer: 3 slabs of two K chunks x 192 columns)][End: 3 slabs]  = 36 L + 2 slabs. */
#define GLOWTTS_WN_SLAB_BYTES 24576
#define GLOWTTS_WN_FUSED_MAX_LAYERS 4
int glowtts_wavenet_image_bytes(int L, int transposed, int64_t *bytes_out /* host */);
/* Packs the effective weights of F flows (stacked fp32, torch Conv1d layouts: w_start [F][H][C2][1], w_in [F][L][2H][H][5], w_rs
 * [F][L-1][2H][H][1] (NULL when L = 1), w_rs_last [F][H][H][1], w_end [F][2 C2][H][1]) into F forward images (img_fwd) and, when
 * img_bwd != NULL, F images of the transposed weights for the fused backward; image f starts at f * glowtts_wavenet_image_bytes(). */
int glowtts_wavenet_pack_images(const float *w_start, const float *w_in, const float *w_rs, const float *w_rs_last, const float *w_end,
                                int F, int L, int C2, void *img_fwd, void *img_bwd, void *stream);
/* The same images described as glowtts_prep_job entries (host array `jobs`, capacity max_jobs; appends at *njobs and advances *njobs / *block), from
 * the weight-norm pairs of the four normalised convs (v_* / g_* stacked like w_* above) and the plain End weight.  img_fwd: F forward images (its jobs
 * also write inv_* [F (* L)][O]); img_bwd: the transposed images of the first F_bwd flows (either may be NULL). */
int glowtts_wavenet_prep_jobs(glowtts_prep_job *jobs, int max_jobs, int *njobs, int *block,
                              const float *v_start, const float *g_start, float *inv_start, const float *v_in, const float *g_in, float *inv_in,
                              const float *v_rs, const float *g_rs, float *inv_rs, const float *v_rsl, const float *g_rsl, float *inv_rsl,
                              const float *w_end, int F, int L, int C2, void *img_fwd, int F_bwd, void *img_bwd);
/* xsrc [R][C]: channels [0, C/2) = x_a (WaveNet input), [C/2, C) = x_b; writes x_b' = m + exp(logs) x_b (reverse: (x_b - m) exp(-logs)),
 * masked, to xdst[r][C/2 ...].  keep != 0: fills a->hs / gates / acts / skip / outs (the activations glowtts_flow_backward reads).
 * Needs p->wn_img. */
int glowtts_wavenet_fwd(const glowtts_flow_dims *d, const glowtts_flow_params *p, const glowtts_flow_acts *a,
                        const float *xsrc, float *xdst, int reverse, int keep, void *stream);
/* Its backward, data gradients only (steps 2-4 of glowtts_flow_backward: End^T, per layer Res_Skip^T + gate derivative + In^T, Start^T), in
 * ONE launch: reads g->douts_bf (the coupling backward's bf16 d(m, logs)) and a->gates, writes g->dskip, g->dins[l] (PAIR-packed), g->dh[l]
 * and accumulates d x_a into g->dx - the operands the grouped weight-gradient launches read.  Needs p->wn_img_t (the transposed image:
 * [End^T 3 slabs][layer L-1 .. 0: Res_Skip^T 3 / 6, In^T tanh-side 15, In^T sigmoid-side 15][Start^T 2]) and g->defer_wgrad.  With g->dcond and
 * p->cond set it also ACCUMULATES the per-utterance conditioning gradient (atomic adds, like the per-conv path).  Not served: the GR-mode
 * per-row conditioning (p->cond_rows / g->pitch_rows) - GLOWTTS_E_ARG, the caller packs the per-conv images for those flows instead. */
int glowtts_wavenet_bwd(const glowtts_flow_dims *d, const glowtts_flow_params *p, const glowtts_flow_acts *a, const glowtts_flow_grads *g,
                        void *stream);

/* Diagnostics (no reference counterpart): on != 0 makes the fused kernels wait conservatively (drain their own stores) at every slab instead
 * of with exact operation counts; results must be bit-identical either way (tests/test_gpu_wavenet_fused.py). */
void glowtts_wavenet_debug_safe_waits(int on);

/* per-utterance column sums: out[b][n] = sum over the rows of utterance b of x[r][col(n)]  (conditioning grads) */
int glowtts_utt_colsum(const float *x, int64_t ldx, float *out, int64_t ldout, int B, int rows_per_utt, int n,
                       int perm, int perm_h, void *stream);

/* ------------------------------------------------------------------------------------------
 * Per-utterance conditioning of the WaveNet gates (Modules.py:832-845: Speaker_l / Prosody_l, weight-normalised Conv1d(D -> 2H, k = 1), and :863-866 where
 * their outputs join the gate pre-activation; round 5, csrc/cond_ops.hip).  All N = F * L * 2H output channels of one kind in one launch, straight from
 * the (weight_g, weight_v) pairs:   out[b][n] = (accumulate ? out[b][n] : 0) + bias[n] + g[n] / ||v[n]|| <v[n], vec[b]>
 *   v [N][D] (D <= 512), g [N], bias [N] (may be NULL), vec [B][D] (B <= 64), out [B][N]; inv_out [N] (may be NULL) receives 1 / ||v[n]|| for the backward.
 * _bwd: from dcond [B][ldd] (this kind's channel n at column n, ldd >= N): dv [N][D], dg [N], dbias [N] (may be NULL) - the weight-norm backward applied in
 * place - and, when dvec [B][D] is given, the vectors' gradient as a deterministic two-stage sum through `scratch`
 * (glowtts_cond_linear_bwd_scratch_floats(N, D, B) floats; always required). */
int glowtts_cond_linear_fwd(const float *v, const float *g, const float *bias, const float *vec, float *out, float *inv_out,
                            int N, int D, int B, int accumulate, void *stream);
int64_t glowtts_cond_linear_bwd_scratch_floats(int N, int D, int B);
/* 1 when glowtts_cond_linear_fwd AND _bwd take (N, D, B) - D in {128, 256, 384, 512}, B <= 64 and both kernels' LDS tiles ((32 + B') (D + 4) floats
 * plus the backward's extras) within 160 KiB, e.g. not D = 512 with B > 40 - else 0: the caller then forms the product itself (ABI 6). */
int glowtts_cond_linear_supported(int N, int D, int B);
/* ABI 7.  The fixed-point accumulators of glowtts_flow_grads.dcond -> float: out[i] = (float)((double)acc[i] * 2^-40), NaN where the accumulator is poisoned
 * (a NaN / Inf / out-of-range addend: the conditioning gradients propagate non-finite values like fp32 sums would). */
int glowtts_fx_to_float(const int64_t *acc, float *out, int64_t n, void *stream);
int glowtts_cond_linear_bwd(const float *dcond, int64_t ldd, const float *v, const float *g, const float *inv, const float *vec,
                            float *dv, float *dg, float *dbias, float *dvec, float *scratch, int N, int D, int B, void *stream);

/* ------------------------------------------------------------------------------------------
 * GRU recurrence of the GST prosody encoder (Modules.py:338-343, 371: torch.nn.GRU, one layer, batch_first, h0 = 0), one launch per
 * direction instead of MIOpen's ~30 launches per time step.  The caller does the GEMMs around it: gi = x W_ih^T + b_ih before,
 * dx = dgi W_ih, dW_ih = dgi^T x, dW_hh = dgh^T h_prev, db = column sums after.   3H <= 1024.
 *   gi [B][T][3H] (r | z | n pre-activations of the input side), w_hh [3H][H], b_hh [3H]
 *   hs [B][T][H] every step's state; keep [B][T][4H] = (r, z, n, W_hn h + b_hn) for the backward
 *   dhs [B][T][H] gradient w.r.t. every step's output -> dgi, dgh [B][T][3H] (gradients of the input- / hidden-side pre-activations) */
int glowtts_gru_fwd(const float *gi, const float *w_hh, const float *b_hh, float *hs, float *keep, int B, int T, int H, void *stream);
int glowtts_gru_bwd(const float *dhs, const float *hs, const float *keep, const float *w_hh, float *dgi, float *dgh,
                    int B, int T, int H, void *stream);
/* ------------------------------------------------------------------------------------------
 * Style-token tail of the GST prosody encoder (Modules.py:345-355, 371-385; ABI 6, csrc/gst_ops.hip): the GRU state at each utterance's last valid step
 * (index ceil(length / stride_prod) - 1, :373) attends over tanh(gst_Tokens) with `H` heads (RPR_MHA.py:69-128 without relative positions / masks, one query):
 *   hs [B][Tp][G] GRU states, lengths [B] i64 (mel frames), tokens [I][NT] = gst_Tokens, Wq [C][G], Wk / Wv [C][I], Wp [C][C] (Conv1d k = 1 weights) + biases [C]
 *   (may be NULL) -> out [B][C].  K, V [C][NT] (batch independent) and keep (glowtts_gst_keep_floats floats) are written for the backward; fp32 arithmetic.
 * _bwd: dout [B][C] -> dhs [B][Tp][G] (fully written: zeros but the gathered step) and every parameter gradient (overwritten; deterministic sums over the batch);
 * scratch: B (2 C + H NT) + 2 C NT floats.  C a multiple of 64, <= 1024; NT <= 256; H <= 8 (glowtts_gst_supported). */
int glowtts_gst_supported(int B, int Tp, int G, int C, int H, int NT, int I);
int64_t glowtts_gst_keep_floats(int B, int G, int C, int H, int NT, int I);
int glowtts_gst_fwd(const float *hs, const int64_t *lengths, int stride_prod, const float *tokens, const float *Wq, const float *bq, const float *Wk,
                    const float *bk, const float *Wv, const float *bv, const float *Wp, const float *bp, float *K, float *V, float *out, float *keep,
                    int B, int Tp, int G, int C, int H, int NT, int I, void *stream);
int glowtts_gst_bwd(const float *dout, const float *keep, const int64_t *lengths, int stride_prod, const float *tokens, const float *Wq, const float *Wk,
                    const float *Wv, const float *Wp, const float *K, const float *V, float *dhs, float *scratch,
                    float *dWq, float *dbq, float *dWk, float *dbk, float *dWv, float *dbv, float *dWp, float *dbp, float *dtokens,
                    int B, int Tp, int G, int C, int H, int NT, int I, void *stream);

/* ------------------------------------------------------------------------------------------
 * Direct Conv2d(3x3, stride 2, padding 1, no bias) (+ ReLU) on channels-last activations x [B][H][W][Ci] -> y [B][Ho][Wo][Co], Ho = ceil(H / 2), Wo =
 * ceil(W / 2): the six layers of the GST reference encoder (Modules.py:320-333, 366-368; ABI 6, csrc/conv2d_ops.hip).  w is the torch Conv2d weight
 * [Co][Ci][3][3] fp32.  No patch matrix, no layout change: the forward and the data gradient are implicit GEMMs on MFMA whose A rows are gathered while
 * they are staged, the weight gradient reads row-major tiles of both operands (exact-fp32 MFMA, or bf16 MFMA on fragments rounded in registers; layer 0: fp32 VALU).  Supported (glowtts_conv3x3s2_supported): Ci = 1 with
 * Co % 4 == 0, Co <= 128 (VALU kernels), or Ci, Co in {32, 64, 128}; B H W max(Ci, Co) < 2^31.
 *
 * Weight images (MFMA tile order, like glowtts_pack_weight): one forward image per layer with Ci > 1 and four data-gradient images, one per parity class
 * cls = 2 (h & 1) + (w & 1) of the INPUT pixel (a pixel of class (1, 1) is read by four taps, (0, 0) by one).  glowtts_conv3x3s2_image_bytes reports the
 * sizes; all images of a stack are written by ONE launch over a device job table (fill the jobs on the host with glowtts_conv3x3s2_pack_job_init - cls
 * = -1: forward image -, block0 = running sum of *blocks_out, copy the table to the device once: it stays valid while the pointers do). */
typedef struct glowtts_c2d_pack_job {
    const float *w; void *img;
    int Ci, Co, cls, N, K, npad, kchunks, block0;
} glowtts_c2d_pack_job;
int glowtts_conv3x3s2_supported(int B, int H, int W, int Ci, int Co);
int glowtts_conv3x3s2_image_bytes(int Ci, int Co, int precision, int64_t *fwd_bytes /* host */, int64_t *dgrad_bytes /* host [4] */);
int glowtts_conv3x3s2_pack_job_init(glowtts_c2d_pack_job *job /* host */, const float *w, int Ci, int Co, int cls, int precision, void *img,
                                    int block0, int *blocks_out /* host */);
int glowtts_conv3x3s2_pack(const glowtts_c2d_pack_job *dev_jobs, int njobs, int total_blocks, int precision, void *stream);
/* y = relu?(conv(x)).  Ci = 1: reads w; else reads img_fwd (w may be NULL). */
int glowtts_conv3x3s2_fwd(const float *x, const float *w, const void *img_fwd, float *y, int B, int H, int W, int Ci, int Co, int relu,
                          int precision, void *stream);
/* dx [B][H][W][Ci] from dpre [B][Ho][Wo][Co] (the gradient of the conv's output BEFORE its ReLU); gate (optional, shaped like dx: the layer's input = the
 * previous layer's ReLU output): dx = gate > 0 ? dx : 0, i.e. dx leaves as the previous layer's dpre.  img_dgrad: HOST array of the four class images. */
int glowtts_conv3x3s2_dgrad(const float *dpre, const void *const *img_dgrad, const float *gate, float *dx, int B, int H, int W, int Ci, int Co,
                            int precision, void *stream);
/* Weight gradient in two steps: _wgrad writes *splits_out partial images [splits][Co][(kh, kw, ci)] into `partial`
 * (glowtts_conv3x3s2_wgrad_scratch_floats(...) floats), one glowtts_conv3x3s2_wgrad_reduce over up to GLOWTTS_C2D_MAX_LAYERS layers sums them in a fixed
 * order into dw [Co][Ci][3][3] (overwritten).  Deterministic, no atomics. */
#define GLOWTTS_C2D_MAX_LAYERS 8
typedef struct glowtts_c2d_reduce_job { const float *partial; float *dw; int splits, Ci, Co, block0 /* set by the call */; } glowtts_c2d_reduce_job;
int64_t glowtts_conv3x3s2_wgrad_scratch_floats(int B, int H, int W, int Ci, int Co);
int glowtts_conv3x3s2_wgrad(const float *x, const float *dpre, float *partial, int B, int H, int W, int Ci, int Co, int precision,
                            int *splits_out /* host */, void *stream);
int glowtts_conv3x3s2_wgrad_reduce(const glowtts_c2d_reduce_job *jobs /* host */, int njobs, void *stream);

/* ------------------------------------------------------------------------------------------
 * Text-encoder kernels that are not convolutions (rows layout).
 */
/* y = rowmask * dropout( relu?( LayerNorm_C(a + b) * gamma + beta ) )   (Modules.py:485-487, 561-562, 569-571; eps 1e-4).
 * b may be NULL; when given, s_out receives a + b (the LayerNorm input the backward needs).  stats [rows][2] = (mean, rstd). */
int glowtts_layernorm_fwd(const float *a, const float *b, float *s_out, const float *gamma, const float *beta, const float *rowmask,
                          float *y, float *stats, int64_t rows, int C, float eps, int relu, float drop_p, uint32_t seed,
                          const uint32_t *seed_ptr, void *stream);
/* same, and y_bf16 (may be NULL) additionally receives y rounded to bf16: the A operand of the LDS-DMA convs that consume y */
int glowtts_layernorm_fwd_io(const float *a, const float *b, float *s_out, const float *gamma, const float *beta, const float *rowmask,
                             float *y, float *stats, int64_t rows, int C, float eps, int relu, float drop_p, uint32_t seed,
                             const uint32_t *seed_ptr, uint16_t *y_bf16, void *stream);
/* Round 4: the attention block's output projection and the LayerNorm behind it in one launch (Modules.py:560-562; replaces glowtts_conv_cl on the
 * fp32 attention rows + glowtts_layernorm_fwd_io):  proj = Dropout(a W^T + bias) (kept in proj_kept when drop_p > 0: the backward's dropout gate),
 * s = proj + x, stats = (mean, rstd) of s over C = 192 channels, y = (LayerNorm(s) gamma + beta) rowmask, y_bf16 = y rounded.  w: the packed bf16
 * image of the [192][192][1] weight (glowtts_pack_weight*, npad columns).  bf16 MFMA operands, fp32 everything else. */
int glowtts_proj_layernorm(const float *a, int64_t lda, const void *w, int npad, const float *bias, const float *x, const float *gamma,
                           const float *beta, const float *rowmask, float *proj_kept, float *s, float *stats, float *y, uint16_t *y_bf16,
                           int64_t rows, int C, float eps, float drop_p, uint32_t seed, const uint32_t *seed_ptr, void *stream);
/* Round 4: the LayerNorm that closes a transformer block and the fused Q / K / V 1x1 conv of the next block in one launch (Modules.py:571 ->
 * RPR_MHA.py:82-84; replaces glowtts_layernorm_fwd_io + glowtts_conv_cl on the bf16 rows):  s = a + b, stats, y, y_bf16 as glowtts_layernorm_fwd_io
 * (no relu, no dropout); qkv [rows][576] = y_bf16 Wqkv^T + bias, wqkv the packed bf16 image of the [576][192][1] weight (npad columns). */
int glowtts_layernorm_qkv(const float *a, const float *b, const float *gamma, const float *beta, const float *rowmask, float *s, float *stats,
                          float *y, uint16_t *y_bf16, const void *wqkv, int npad, const float *bias, float *qkv, int64_t rows, int C, float eps,
                          void *stream);
int64_t glowtts_layernorm_scratch_floats(int64_t rows, int C);
/* ds = dL/d(a + b); dgamma_dbeta [2C].  gated != 0: the forward applied relu and/or dropout, y is its output (zero where cut). */
int glowtts_layernorm_bwd(const float *dy, const float *y, const float *s, const float *stats, const float *gamma, const float *rowmask,
                          float *ds, float *dgamma_dbeta, float *scratch, int64_t rows, int C, int gated, float drop_p, void *stream);
/* same, and: ds_bf16 (may be NULL) additionally receives ds rounded to bf16 - with gate_out ([rows][C] fp32, the forward output of the conv
 * that produced the LayerNorm input) as d(pre-activation) of that conv, ds * (gate_out != 0 ? gate_scale : 0), i.e. through its relu /
 * dropout gate; dgamma_dbeta == NULL leaves the per-workgroup partials in `scratch` ([glowtts_layernorm_scratch_floats / (2C)][2C]) for one
 * glowtts_colsum_batched over several calls. */
int glowtts_layernorm_bwd_io(const float *dy, const float *y, const float *s, const float *stats, const float *gamma, const float *rowmask,
                             float *ds, float *dgamma_dbeta, float *scratch, int64_t rows, int C, int gated, float drop_p,
                             uint16_t *ds_bf16, const float *gate_out, float gate_scale, void *stream);
/* dz = dy * (out != 0 ? scale : 0) * rowmask : backward gate of relu / dropout given the forward output */
int glowtts_gate_bwd(const float *dy, const float *out, const float *rowmask, float *dz, int64_t rows, int C, float scale, void *stream);
/* io_flags: 1 = dy, 2 = out, 4 = dz stored as bf16 instead of fp32 (C a multiple of 4).  ABI 7: out == NULL (with a rowmask) = no gate, dz = dy * scale * rowmask. */
int glowtts_gate_bwd_io(const void *dy, const void *out, const float *rowmask, void *dz, int64_t rows, int C, float scale, int io_flags, void *stream);
/* rows[b][PAD+t][:] = table[tokens[b][t]][:] * scale * mask (Modules.py:267), and its gradient (deterministic) */
/* mask [B][T] = (t < lengths[b]) and the rows layout's row mask [B][T + 2 GLOWTTS_ROW_PAD] (zero pad rows) in one launch (Modules.py:206-211; round 5) */
int glowtts_token_masks(const int64_t *lengths, float *mask, float *rowmask, int B, int T, void *stream);
int glowtts_embedding_fwd(const int64_t *tokens, const float *table, const float *rowmask, float *rows, int B, int T, int C, float scale, void *stream);
int glowtts_embedding_bwd(const int64_t *tokens, const float *drows, const float *rowmask, float *dtable, int V, int B, int T, int C, float scale, void *stream);
/* Relative-position multi-head self-attention core (RPR_MHA.py:95-128; window `win`, embeddings shared over heads).
 * qkv rows [B][Tp][3*H*D] (Q | K | V); out rows [B][Tp][H*D]; P [B][H][Tp][Tp] is kept for the backward (opaque to the caller:
 * the MFMA paths - D in {64, 96}, win <= 15; one workgroup per (utterance, head) for Tp <= 128, query / key blocks above - store the
 * probabilities before dropout, the general path after). Tp <= 256.
 * The backward takes the forward's (seed, seed_ptr): it regenerates the dropout keep mask. */
int glowtts_rpr_attention_fwd(const float *qkv, const float *relk, const float *relv, const float *rowmask, float *out, float *P,
                              int B, int Tp, int H, int D, int win, float drop_p, uint32_t seed, const uint32_t *seed_ptr, void *stream);
int64_t glowtts_rpr_attention_scratch_floats(int B, int Tp, int H, int D, int win);
/* scratch: glowtts_rpr_attention_scratch_floats(...) + 2*(2*win+1)*D floats; dS: [B][H][Tp][Tp] floats (general path only);
 * drelk / drelv [2*win+1][D] each; passing them adjacent (one [2][2*win+1][D] tensor) saves two copies */
int glowtts_rpr_attention_bwd(const float *qkv, const float *relk, const float *relv, const float *rowmask, const float *P, const float *dout,
                              float *dS, float *dqkv, float *drelk, float *drelv, float *scratch,
                              int B, int Tp, int H, int D, int win, float drop_p, uint32_t seed, const uint32_t *seed_ptr, void *stream);
/* The same with an arithmetic mode (GLOWTTS_F32 = the two functions above; GLOWTTS_BF16: for Tp <= 128 the five contractions of the core run on
 * bf16 MFMAs - operands rounded to bf16 in registers, fp32 accumulate, fp32 softmax - the other paths are fp32 in either mode). */
int glowtts_rpr_attention_fwd_prec(const float *qkv, const float *relk, const float *relv, const float *rowmask, float *out, float *P,
                                   int B, int Tp, int H, int D, int win, float drop_p, uint32_t seed, const uint32_t *seed_ptr, int precision, void *stream);
/* rows of per-workgroup partial sums of (d relK | d relV) [rows][2 (2 win + 1) D] that glowtts_rpr_attention_bwd* leaves at the head of `scratch`; with
 * drelk == drelv == NULL (round 5) the call stops there and the caller sums them itself (glowtts_colsum_batched over several layers at once). */
int64_t glowtts_rpr_attention_bwd_partial_rows(int B, int Tp, int H, int D, int win);
int glowtts_rpr_attention_bwd_prec(const float *qkv, const float *relk, const float *relv, const float *rowmask, const float *P, const float *dout,
                                   float *dS, float *dqkv, float *drelk, float *drelv, float *scratch,
                                   int B, int Tp, int H, int D, int win, float drop_p, uint32_t seed, const uint32_t *seed_ptr, int precision, void *stream);

/* ------------------------------------------------------------------------------------------
 * Alignment expansion and likelihood loss.
 */
/* out[b][c][y] = idx[b][y] >= 0 ? src[b][c][idx[b][y]] : 0   ==  src @ attentions (Modules.py:120-121), attentions one-hot per frame */
/* Operands of the log-prior GEMM (Modules.py:108-114) in one launch: the fp32 MFMA weight image of (sigma^-2 | mu sigma^-2)
 * [B][kchunks][npad][16] (what glowtts_pack_weight_batched(F32) would produce from [B][Tx][2 Cm]), the per-token constant cb [B][Tx], the
 * frame mask [B][Ty] and the lengths as int32; mel lengths are first rounded down to a multiple of `mel_multiple` (Decoder.Num_Squeeze,
 * Modules.py:897-898).  packed == NULL: size query (npad_out, kchunks_out). */
int glowtts_logprior_prep(const float *mean, const float *log_std, const int64_t *token_lengths, const int64_t *mel_lengths, float *packed,
                          float *cb, float *fmask, int32_t *tx32, int32_t *ty32, int B, int Cm, int Tx, int Ty, int mel_multiple,
                          int *npad_out, int *kchunks_out, void *stream);
int glowtts_expand_fwd(const float *src, const int32_t *idx, float *out, int B, int C, int Tx, int Ty, void *stream);
/* its gradient w.r.t. src: a segment sum over the (contiguous) frames of each token */
int glowtts_expand_bwd(const float *dout, const int32_t *idx, float *dsrc, int B, int C, int Tx, int Ty, void *stream);
/* log_Duration_Targets = log(frames per token + 1e-7) * token_mask   (Modules.py:122); out [B][Tx] */
int glowtts_duration_targets(const int32_t *idx, const int64_t *token_lengths, float *out, int B, int Tx, int Ty, void *stream);
/* MLE_Loss (Modules.py:1020-1029) over n = B*mel_dim*T_mel elements; loss and inv_denom are device scalars; scratch 1024 floats */
int glowtts_mle_loss_fwd(const float *z, const float *mean, const float *log_std, const float *log_dets, const int64_t *lengths,
                         float *loss, float *inv_denom, float *scratch, int64_t n, int B, int n_squeeze, int mel_dim, void *stream);
/* dlogdet (optional, [B]): also writes d loss / d log_dets[b] = -dloss * inv_denom */
int glowtts_mle_loss_bwd(const float *z, const float *mean, const float *log_std, const float *dloss, const float *inv_denom,
                         float *dz, float *dmean, float *dlog_std, int64_t n, float *dlogdet, int B, void *stream);
/* ABI 6.  Both expansions and the duration targets of Modules.py:120-122 in one launch (Ty % 4 == 0, 16-byte aligned idx / outputs): mel_mean, mel_log_std
 * [B][C][Ty] as glowtts_expand_fwd, targets [B][Tx] as glowtts_duration_targets.  ABI 7: path (optional, fp32 [B][Tx][Ty], 16-byte aligned): the dense 0/1
 * attentions of Modules.py:116 as glowtts_mas_path_from_idx writes them, in the same launch. */
int glowtts_expand_pair_targets(const float *mean, const float *log_std, const int32_t *idx, const int64_t *token_lengths, float *mel_mean,
                                float *mel_log_std, float *targets, float *path, int B, int C, int Tx, int Ty, void *stream);
/* ABI 6.  glowtts_mle_loss_bwd THROUGH the expansion, in one launch: z [B][C][Ty], the TOKEN-space mean / log_std [B][C][Tx] the expansion gathered from and idx
 * [B][Ty] -> dz [B][C][Ty] and the token-space gradients dmean / dlog_std [B][C][Tx] (each token's frames are one contiguous run: segment sums, no atomics;
 * the same bits as glowtts_mle_loss_bwd followed by two glowtts_expand_bwd); dlogdet optional as above.  Tx <= 1024. */
int glowtts_prior_loss_bwd(const float *z, const float *mean, const float *log_std, const int32_t *idx, const float *dloss, const float *inv_denom,
                           float *dz, float *dmean, float *dlog_std, float *dlogdet, int B, int C, int Tx, int Ty, void *stream);
/* ABI 6.  loss = s * sum (a - target)^2 and da = 2 s dloss (a - target), one launch per direction; loss / dloss are device scalars.  s = `scale` (Train.py:203-211:
 * MSELoss on the log durations, scale = 1 / n), or - lengths [B] i64 given - 1 / (B max(lengths)): the mean over the batch's own longest text when the token axis is
 * padded to a shape bucket, or - extent (device scalar) given - 1 / (B extent[0]): data parallel, the global batch's longest text. */
int glowtts_mse_loss_fwd(const float *a, const float *target, float *loss, int64_t n, float scale, const int64_t *lengths, int B, const float *extent,
                         float *da_unit, void *stream);
int glowtts_mse_loss_bwd(const float *a, const float *target, const float *dloss, float *da, int64_t n, float scale, const int64_t *lengths, int B,
                         const float *extent, void *stream);
/* ABI 7.  glowtts_mse_loss_fwd's da_unit (optional, [n]): the gradient for d loss = 1, written by the forward launch (glowtts_mse_loss_bwd's value for that seed). */
/* ABI 7.  MLE_Loss (Modules.py:1020-1029) on the expanded prior - value AND gradients for the seed dloss[0] - in ONE launch: glowtts_mle_loss_fwd on
 * (z, mel_mean, mel_log_std) [B][C][Ty] and glowtts_prior_loss_bwd on the token-space (mean, log_std) [B][C][Tx] + idx [B][Ty], the same bits as those calls.
 * scratch: 1024 floats; counter: one uint32 zeroed once by the caller (the kernel leaves it zero).  dlogdet optional.  Tx <= 1024. */
int glowtts_prior_loss(const float *z, const float *mel_mean, const float *mel_log_std, const float *mean, const float *log_std, const int32_t *idx,
                       const float *log_dets, const int64_t *lengths, const float *dloss, float *scratch, uint32_t *counter, float *loss, float *inv_denom,
                       float *dz, float *dmean, float *dlog_std, float *dlogdet, int B, int C, int Tx, int Ty, int n_squeeze, int mel_dim, void *stream);
/* ABI 7.  Duration_Predictor's Projection (Modules.py:596-618: Conv1d(C -> 1, k = 1) on masked features, times the mask) on rows: d [B][T + 2 pad][C] fp32
 * (16-byte aligned, C % 4 == 0, C <= 1024), w [C], bias [1], mask [B][T] -> out [B][T].  Backward: g [B][T] -> dd rows (pad rows zero), dw [C], dbias [1];
 * scratch B * (C + 1) floats, counter one uint32 zeroed once (left zero).  Fixed summation order. */
int glowtts_dur_proj_supported(int C);
int glowtts_dur_proj_fwd(const float *d, const float *w, const float *bias, const float *mask, float *out, int B, int T, int pad, int C, void *stream);
int glowtts_dur_proj_bwd(const float *g, const float *mask, const float *d, const float *w, float *dd, float *dw, float *dbias, float *scratch,
                         uint32_t *counter, int B, int T, int pad, int C, void *stream);
/* ABI 7.  The encoder's projected rows [B][T + 2 pad][2 M] -> mean, log_std [B][M][T] (Modules.py:283-286) and back (dmean / dlog_std may be NULL = zero;
 * pad rows of drows are zeroed). */
int glowtts_prior_split_fwd(const float *rows, float *mean, float *log_std, int B, int T, int pad, int M, void *stream);
int glowtts_prior_split_bwd(const float *dmean, const float *dlog_std, float *drows, int B, int T, int pad, int M, void *stream);

/* ------------------------------------------------------------------------------------------
 * Old-style weight normalisation of the WaveNet convolutions (Modules.py:766,818,825,838,845: torch.nn.utils.weight_norm,
 * norm over (in, k) per output channel) for a whole stack of convs per launch.
 *   v [rows][cols], g [rows]  ->  w = g * v / ||v||  [rows][cols],  inv_norm [rows] = 1 / ||v||   (kept for the backward)
 *   backward: dg = <dw, v> / ||v||,  dv = g / ||v|| * (dw - v <dw, v> / ||v||^2)
 * rows = stacked convs x output channels, cols = in * k.  All pointers fp32 device tensors. */
int glowtts_weightnorm_fwd(const float *v, const float *g, float *w, float *inv_norm, int64_t rows, int cols, void *stream);
int glowtts_weightnorm_bwd(const float *dw, const float *v, const float *g, const float *inv_norm, float *dv, float *dg,
                           int64_t rows, int cols, void *stream);
/* ABI 7.  Up to GLOWTTS_WN_BWD_MAX_JOBS glowtts_weightnorm_bwd problems in one launch (`jobs` is a HOST array; same arithmetic, same bits). */
#define GLOWTTS_WN_BWD_MAX_JOBS 8
typedef struct glowtts_wn_bwd_job {
    const float *dw, *v, *g, *inv_norm; float *dv, *dg;
    int64_t rows; int cols, reserved;
} glowtts_wn_bwd_job;
int glowtts_weightnorm_bwd_multi(const glowtts_wn_bwd_job *jobs, int njobs, void *stream);

/* ------------------------------------------------------------------------------------------
 * Optimizer side of the training step: multi-tensor launches over a DEVICE job table.
 * A job = one contiguous run of fp32 elements (a stacked weight class is one job); block0 = running sum of ceil(n / glowtts_opt_chunk())
 * over the preceding jobs, total_blocks = that sum over all jobs.
 *   glowtts_multi_grad_norm : norm_and_coef[0] = ||all gradients||_2, [1] = min(1, max_norm / (norm + 1e-6))   (torch.nn.utils.clip_grad_norm_,
 *                             Train.py:228-231); partial: total_blocks floats of scratch; deterministic two-stage reduction
 *   glowtts_multi_grad_scale: g *= coef[0] in place (clip_grad_norm_'s second half)
 *   glowtts_radam_step      : Rectified Adam exactly as Radam.py:45-90.  hyper (device, 8 floats): lr, beta1, beta2, eps, weight_decay,
 *                             step_size, rectified (N_sma >= 5), unused - computed on the host per step like Radam.py:63-79 and uploaded, so a
 *                             captured hipGraph of the step stays valid while the learning rate changes; grad_scale (device, optional): the
 *                             clip coefficient applied on the fly instead of a separate scale pass. */
typedef struct glowtts_opt_job {
    float *p; const float *g; float *m; float *v;     /* parameter, gradient, exp_avg, exp_avg_sq */
    int64_t n;
    int64_t block0;
} glowtts_opt_job;
int glowtts_opt_chunk(void);
int glowtts_multi_grad_norm(const glowtts_opt_job *dev_jobs, int njobs, int total_blocks, float max_norm, float *partial,
                            float *norm_and_coef, void *stream);
/* out[i] = sum over s < S of partial[s][i], i < n, in a fixed order (round 5; n % 4 == 0, both pointers 16-byte aligned): the row splits of the text
 * encoder's weight gradients summed into the gradients (no reference counterpart: autograd of Modules.py:438-573 accumulates in one pass). */
int glowtts_sum_slices(const float *partial, float *out, int S, int64_t n, void *stream);
/* ... into up to GLOWTTS_SUM_MAX_SEGS destination tensors: partial [S][stride]; segment k = elements [off, off + n) of a slice -> dst[0 .. n); the segments tile
 * [0, sum n) in ascending order, every n a multiple of 4, every dst 16-byte aligned (the decoder's one-tap weight gradients in row splits). */
#define GLOWTTS_SUM_MAX_SEGS 12
typedef struct glowtts_sum_seg { float *dst; int64_t off, n; } glowtts_sum_seg;
int glowtts_sum_slices_seg(const float *partial, int S, int64_t stride, const glowtts_sum_seg *segs /* host */, int nseg, void *stream);
int glowtts_multi_grad_scale(const glowtts_opt_job *dev_jobs, int njobs, int total_blocks, const float *coef, void *stream);
int glowtts_radam_step(const glowtts_opt_job *dev_jobs, int njobs, int total_blocks, const float *hyper, const float *grad_scale, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GLOWTTS_HIP_H */

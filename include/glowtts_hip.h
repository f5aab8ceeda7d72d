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

/* Library / device identification.  Returns the ABI version (currently 1). */
int glowtts_abi_version(void);
/* Writes the gfx arch string of device 0 into buf (host pointer).  0 on success. */
int glowtts_device_arch(char *buf, int buflen);

/* ------------------------------------------------------------------------------------------
 * Monotonic Alignment Search.
 * Replaces monotonic_align/core.pyx:40 `maximum_path_c(paths, values, t_xs, t_ys, max_neg_val)`
 * (and the pure-Python twin Modules.py:951-980).  Bit-exact with core.pyx:9-35 on identical fp32 input.
 *
 *   value    [B][Tx][Ty] f32, already multiplied by the mask (monotonic_align/__init__.py:11).
 *            Read-only unless q_out == value (then clobbered into cumulative scores like core.pyx:30).
 *   t_xs,t_ys[B] i32      valid tokens / frames per utterance (1 <= t_x <= t_y required, as in the
 *            reference where t_x > t_y is undefined; such rows get an all-zero path and idx = -1).
 *   idx_out  [B][Ty] i32  token index aligned to each frame, -1 for y >= t_y.           (may be NULL)
 *   q_out    [B][Tx][Ty] f32 cumulative scores exactly as core.pyx leaves `values`      (may be NULL)
 *   Tx <= 512.
 */
int glowtts_mas_dp_f32(const float *value, const int32_t *t_xs, const int32_t *t_ys,
                       int32_t *idx_out, float *q_out, int B, int Tx, int Ty,
                       float max_neg_val, void *stream);
/* Dense 0/1 path from idx (core.pyx:32-35 writes these ones into a pre-zeroed array; here every
 * element is written, so `path` need not be zeroed).  out_dtype: 0 = int32, 1 = float32. */
int glowtts_mas_path_from_idx(const int32_t *idx, void *path, int B, int Tx, int Ty,
                              int out_dtype, void *stream);
/* Both steps: the direct replacement of maximum_path_c.  `scratch_idx` [B][Ty] i32. */
int glowtts_mas_f32(const float *value, int32_t *path, const int32_t *t_xs, const int32_t *t_ys,
                    int32_t *scratch_idx, int B, int Tx, int Ty, float max_neg_val, void *stream);


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

#define GLOWTTS_APRO_NONE    0
#define GLOWTTS_APRO_PAIRMUL 1  /* a[r][c] = A[r][2c] * A[r][2c+1]   (tanh*sigmoid gates, Modules.py:885-887) */

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
    float *out1; int64_t ld1;
    const float *in0; int64_t ldi0;
} glowtts_conv_args;

int glowtts_conv_cl(const glowtts_conv_args *args /* host pointer */, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GLOWTTS_HIP_H */

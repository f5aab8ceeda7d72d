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

#ifdef __cplusplus
}
#endif
#endif /* GLOWTTS_HIP_H */

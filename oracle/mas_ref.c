/* ORACLE - test infrastructure, not product code.
 *
 * Plain-C restatement of the reference's Monotonic Alignment Search:
 *   monotonic_align/core.pyx:9-35  maximum_path_each  (forward DP + backtrack)
 *   monotonic_align/core.pyx:40-45 maximum_path_c     (loop over the batch; `prange`, serial in
 *                                                      the default build - setup.py:5-9 passes no -fopenmp)
 * Semantics kept exactly (fp32 add/compare only, int32 path):
 *   - band:        x in [max(0, t_x + y - t_y), min(t_x, y + 1))          core.pyx:18
 *   - v_cur:       x == y ? max_neg_val : value[x][y-1]                    core.pyx:19-22
 *   - v_prev:      x == 0 ? (y == 0 ? 0 : max_neg_val) : value[x-1][y-1]   core.pyx:23-29
 *   - value[x][y] += max(v_cur, v_prev) with max(a,b) = (b > a) ? b : a    core.pyx:30 (Cython's lowering)
 *   - backtrack:   index = t_x-1; for y = t_y-1..0: path[index][y] = 1;
 *                  if index != 0 && (index == y || value[index][y-1] < value[index-1][y-1]) index--   core.pyx:32-35
 * `value` is clobbered into cumulative scores inside the band (cells outside keep their input),
 * `path` must arrive zeroed.  Pinned bit-exact against the reference's compiled core.pyx
 * (oracle/_ref) by tests/golden/make_golden.py and replayed from tests/golden/mas_*.npz.
 */
#include <stdint.h>

static void mas_each(int32_t *path, float *value, int t_x, int t_y, int Ty, float max_neg_val)
{
    for (int y = 0; y < t_y; ++y) {
        int lo = t_x + y - t_y; if (lo < 0) lo = 0;
        int hi = (y + 1 < t_x) ? y + 1 : t_x;
        for (int x = lo; x < hi; ++x) {
            float v_cur = (x == y) ? max_neg_val : value[(long)x * Ty + y - 1];
            float v_prev;
            if (x == 0) v_prev = (y == 0) ? 0.0f : max_neg_val;
            else        v_prev = value[(long)(x - 1) * Ty + y - 1];
            float m = (v_prev > v_cur) ? v_prev : v_cur;
            value[(long)x * Ty + y] = m + value[(long)x * Ty + y];
        }
    }
    int index = t_x - 1;
    for (int y = t_y - 1; y >= 0; --y) {
        path[(long)index * Ty + y] = 1;
        if (index != 0 && (index == y || value[(long)index * Ty + y - 1] < value[(long)(index - 1) * Ty + y - 1]))
            index -= 1;
    }
}

/* values [B][Tx][Ty] in/out, paths [B][Tx][Ty] pre-zeroed, t_xs/t_ys [B]. */
void mas_ref_maximum_path_c(int32_t *paths, float *values, const int32_t *t_xs, const int32_t *t_ys,
                            int B, int Tx, int Ty, float max_neg_val)
{
    for (int b = 0; b < B; ++b)
        mas_each(paths + (long)b * Tx * Ty, values + (long)b * Tx * Ty, t_xs[b], t_ys[b], Ty, max_neg_val);
}

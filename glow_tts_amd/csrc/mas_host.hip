// Host twin of the Monotonic Alignment Search (SURVEY 2a `mas_cpu.cpp`, 8b-B2 "the same without stream for the CPU twin"): the C-ABI
// counterpart of monotonic_align/core.pyx:40 `maximum_path_c` for callers that hold the score matrix in HOST memory (dataset tools,
// the reference's own CPU code path, Hyper_Parameters `Device: '-1'`).  It is an explicit entry point, NOT a fallback: nothing in the
// GPU path ever routes here, and the device entry points fail loudly without a GPU.
// Same arithmetic as core.pyx:9-35 (fp32 add / compare only, strict `<` in the backtrack), utterances spread over host threads
// (core.pyx:44 `prange`; the reference's default build is serial because setup.py passes no OpenMP flag).
#include <stdint.h>
#include <algorithm>
#include <thread>
#include <vector>
#include "../../include/glowtts_hip.h"

namespace {

void mas_each(int32_t* path, float* value, int t_x, int t_y, int Ty, float neg)
{
    if (t_x < 1) return;
    // (t_y < t_x - no monotonic alignment exists: the loops below are then empty for every column and the backtrack walks the raw inputs,
    // exactly as core.pyx does; its last test, at y == 0, would read value[index][-1] but cannot change the path any more: skipped)
    for (int y = 0; y < t_y; ++y) {
        const int lo = std::max(0, t_x + y - t_y), hi = std::min(t_x, y + 1);
        for (int x = lo; x < hi; ++x) {
            const float v_cur = (x == y) ? neg : value[(size_t)x * Ty + y - 1];
            const float v_prev = (x == 0) ? (y == 0 ? 0.f : neg) : value[(size_t)(x - 1) * Ty + y - 1];
            value[(size_t)x * Ty + y] += (v_prev > v_cur) ? v_prev : v_cur;                        // core.pyx:30 max(v_prev, v_cur)
        }
    }
    int index = t_x - 1;
    for (int y = t_y - 1; y >= 0; --y) {
        path[(size_t)index * Ty + y] = 1;
        if (y > 0 && index != 0 && (index == y || value[(size_t)index * Ty + y - 1] < value[(size_t)(index - 1) * Ty + y - 1])) index -= 1;
    }
}

}  // namespace

extern "C" int glowtts_mas_f32_host(float* value, int32_t* path, const int32_t* t_xs, const int32_t* t_ys,
                                    int B, int Tx, int Ty, float max_neg_val, int num_threads)
{
    if (!value || !path || !t_xs || !t_ys || B < 0 || Tx < 1 || Ty < 1) return GLOWTTS_E_ARG;
    for (int b = 0; b < B; ++b) if (t_xs[b] > Tx || t_ys[b] > Ty || t_xs[b] < 0 || t_ys[b] < 0) return GLOWTTS_E_ARG;
    int nt = num_threads > 0 ? num_threads : (int)std::thread::hardware_concurrency();
    nt = std::max(1, std::min(nt, B));
    auto run = [&](int t) {
        for (int b = t; b < B; b += nt)
            mas_each(path + (size_t)b * Tx * Ty, value + (size_t)b * Tx * Ty, t_xs[b], t_ys[b], Ty, max_neg_val);
    };
    if (nt == 1) { run(0); return GLOWTTS_OK; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back(run, t);
    for (auto& x : th) x.join();
    return GLOWTTS_OK;
}

// Shared by mas.hip (general kernel) and mas_dp2.hip (the production shape): degenerate utterances and the backtrack.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace glowtts_detail {
// mas_dp2.hip: transposed scores [B][Ty][Tx], Tx even and <= 128, 8-byte aligned, Tx * Ty * 4 < 2^31; lds = bytes of parked bit words
int launch_mas_dp2(const float* value_t, const int32_t* t_xs, const int32_t* t_ys, int32_t* idx_out, float* q_out_t,
                   int B, int Tx, int Ty, float neg, size_t lds, hipStream_t s);
}

namespace {

// Lengths outside the tensors (empty alignment) and more tokens than frames; true when the utterance has been dealt with.
template <bool TR>
__device__ __forceinline__ bool mas_degenerate(const float* vb, int32_t* idx_b, int tx, int ty, int Tx, int Ty, int lane)
{
    if (tx < 1 || ty < 0 || tx > Tx || ty > Ty) {             // lengths outside the tensors: empty alignment
        if (idx_b) for (int y = lane; y < Ty; y += 64) idx_b[y] = -1;
        return true;
    }
    if (ty < tx) {
        // More tokens than frames: no monotonic alignment exists.  core.pyx:15-17's loops are then empty for every column (lo >= hi), `value`
        // stays as it was passed in, and the backtrack (:31-35) walks those RAW inputs from row t_x - 1; reproduced literally (a serial,
        // wave-uniform walk: not a case a model produces; the test at y == 0 cannot change the path any more and is skipped).
        int index = tx - 1;
        for (int y = ty - 1; y >= 0; --y) {
            if (idx_b && lane == 0) idx_b[y] = index;
            if (y > 0 && index != 0) {
                const float a = TR ? vb[(size_t)(y - 1) * Tx + index] : vb[(size_t)index * Ty + y - 1];
                const float c = TR ? vb[(size_t)(y - 1) * Tx + index - 1] : vb[(size_t)(index - 1) * Ty + y - 1];
                if (index == y || a < c) index -= 1;
            }
        }
        if (idx_b) for (int y = max(ty, 0) + lane; y < Ty; y += 64) idx_b[y] = -1;
        return true;
    }
    return false;
}

// Backtrack (core.pyx:31-35) over the parked bit words dec[blk][j][lane] (bit 31 - c <-> column blk*32 + c, row lane*R + j): a wave-uniform
// walk on the scalar unit, one step per row change.  One 32-column block: entered on row `index`, left on the row of the block below.
template <int R>
__device__ __forceinline__ void mas_backtrack_block(const unsigned int* dec, int32_t* idx_b, int blk, int& index, int ty, int Ty, int lane)
{
    const int i0 = index;
    const int row = i0 - lane;
    unsigned int W = 0u;
    if (lane <= 32 && row >= 0) W = dec[(blk * R + (row % R)) * 64 + (row / R)];
    int myidx = -1;
    int yy = min(31, ty - 1 - blk * 32);                    // highest column of this block inside the utterance
    while (yy >= 0) {
        if (lane <= yy) myidx = index;                       // path[index][y] = 1 for every column down to the move
        if (index == 0) break;
        const unsigned int w = __builtin_amdgcn_readlane(W, i0 - index);
        const unsigned int masked = w & (0xFFFFFFFFu << (31 - yy));            // columns <= yy
        const int c_bit = masked ? 31 - (__builtin_ffs(masked) - 1) : -1;       // highest column <= yy whose bit is set
        const int c_diag = index - blk * 32;                                    // forced move where index == y
        const int c_move = max(c_bit, (c_diag >= 0 && c_diag <= yy) ? c_diag : -1);
        if (c_move < 0) break;                                                   // stays on this row for the rest of the block
        index = __builtin_amdgcn_readfirstlane(index - 1);
        yy = c_move - 1;
    }
    if (idx_b && lane < 32 && blk * 32 + lane < Ty) idx_b[blk * 32 + lane] = (blk * 32 + lane < ty) ? myidx : -1;
}

template <int R>
__device__ __forceinline__ void mas_backtrack(const unsigned int* dec, int32_t* idx_b, int tx, int ty, int Ty, int lane)
{
    const int nblk = (ty + 31) >> 5;
    __syncthreads();
    int index = tx - 1;
    for (int blk = nblk - 1; blk >= 0; --blk) mas_backtrack_block<R>(dec, idx_b, blk, index, ty, Ty, lane);
    if (idx_b) for (int y = nblk * 32 + lane; y < Ty; y += 64) idx_b[y] = -1;
}

}  // namespace

// Monotonic Alignment Search for gfx950 - one wavefront per utterance.
//
// Replaces monotonic_align/core.pyx:9-45 (reference).  Same fp32 add/compare sequence per DP cell,
// so paths and cumulative scores are bit-exact with the C code Cython generates.
//
// Mapping (CDNA4, wave64):
//   * lane l owns R consecutive token rows x = l*R + j (R = ceil(Tx/64), compile-time 1..8); the
//     previous column of cumulative scores lives in R VGPRs.  The only cross-lane dependency,
//     Q[x-1][y-1] for j == 0, is one DPP wave_shr:1 move per column (no LDS, no barrier).
//   * `value` is streamed straight from HBM into registers: each lane reads 16 B (4 columns) per
//     row per load, a whole 32-column block (8 loads per row) is issued one block ahead of its
//     use, which covers the ~900-cycle HBM latency.
//   * back-pointers are 1 bit per cell, accumulated in registers for 32 columns and parked in LDS
//     (Ty*R*8 bytes instead of the 4*Tx*Ty-byte matrix the reference re-reads).
//   * backtrack runs on the scalar unit: for each 32-column block, lanes 0..32 fetch the bit words of
//     the 33 rows the path can visit, and the 32 dependent steps are v_readlane + s_bitcmp.
//   * the dense 0/1 path the reference API returns is written by a second, fully parallel kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/glowtts_hip.h"

namespace {

__device__ __forceinline__ float wave_shr1(float v) {
    // lane l receives lane l-1's value; lane 0 keeps its own (bound_ctrl = false, old = v)
    int r = __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
    return __int_as_float(r);
}

template <int R, bool VEC4>
__global__ __launch_bounds__(64) void mas_dp_kernel(const float* __restrict__ value,
                                                    const int32_t* __restrict__ t_xs,
                                                    const int32_t* __restrict__ t_ys,
                                                    int32_t* __restrict__ idx_out, float* q_out,
                                                    int Tx, int Ty, float neg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned int dec[];   // [nblk][R][64]
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    const int tx = t_xs[b], ty = t_ys[b];
    int32_t* idx_b = idx_out ? idx_out + (size_t)b * Ty : nullptr;
    if (tx < 1 || ty < tx || tx > Tx || ty > Ty) {            // undefined in the reference -> empty alignment
        if (idx_b) for (int y = lane; y < Ty; y += 64) idx_b[y] = -1;
        return;
    }
    const float* vb = value + (size_t)b * Tx * Ty;
    float* qb = q_out ? q_out + (size_t)b * Tx * Ty : nullptr;
    const int nblk = (ty + 31) >> 5;

    // row pointers (rows >= Tx are clamped: they are never inside the band)
    const float* rowp[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        int x = lane * R + j;
        rowp[j] = vb + (size_t)(x < Tx ? x : Tx - 1) * Ty;
    }

    float4 cur[8][R], nxt[8][R];
    auto load_block = [&](float4 (&dst)[8][R], int blk) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int y0 = blk * 32 + c * 4;
#pragma unroll
            for (int j = 0; j < R; ++j) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (VEC4) {
                    if (y0 < Ty) v = *reinterpret_cast<const float4*>(rowp[j] + y0);
                } else {
                    if (y0 + 0 < Ty) v.x = rowp[j][y0 + 0];
                    if (y0 + 1 < Ty) v.y = rowp[j][y0 + 1];
                    if (y0 + 2 < Ty) v.z = rowp[j][y0 + 2];
                    if (y0 + 3 < Ty) v.w = rowp[j][y0 + 3];
                }
                dst[c][j] = v;
            }
        }
    };

    float q[R];
#pragma unroll
    for (int j = 0; j < R; ++j) q[j] = 0.f;

    load_block(cur, 0);
    for (int blk = 0; blk < nblk; ++blk) {
        if (blk + 1 < nblk) load_block(nxt, blk + 1);
        unsigned int bits[R];
#pragma unroll
        for (int j = 0; j < R; ++j) bits[j] = 0u;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int y = blk * 32 + c * 4 + e;
                if (y < ty) {                                   // wave-uniform
                    const int lo = max(0, tx + y - ty);         // core.pyx:18
                    const int hi = min(tx, y + 1);
                    const float up = wave_shr1(q[R - 1]);       // Q[x-1][y-1] of the first row of this lane
                    float qo[R];
#pragma unroll
                    for (int j = 0; j < R; ++j) qo[j] = q[j];
#pragma unroll
                    for (int j = 0; j < R; ++j) {
                        const int x = lane * R + j;
                        const float prevq = (j == 0) ? up : qo[j - 1];
                        const float v_cur = (x == y) ? neg : qo[j];                       // core.pyx:19-22
                        const float v_prev = (x == 0) ? (y == 0 ? 0.f : neg) : prevq;     // core.pyx:23-29
                        bits[j] |= (qo[j] < prevq ? 1u : 0u) << (c * 4 + e);              // core.pyx:34 test, for the backtrack
                        const float m = (v_prev > v_cur) ? v_prev : v_cur;                // Cython max(a,b) = b>a ? b : a
                        const float val = (e == 0) ? cur[c][j].x : (e == 1) ? cur[c][j].y : (e == 2) ? cur[c][j].z : cur[c][j].w;
                        const bool inb = (x >= lo) && (x < hi);
                        q[j] = inb ? (m + val) : val;                                      // core.pyx:30; cells outside the band keep the input
                        if (qb && inb) qb[(size_t)x * Ty + y] = q[j];
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < R; ++j) dec[(blk * R + j) * 64 + lane] = bits[j];
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int j = 0; j < R; ++j) cur[c][j] = nxt[c][j];
    }
    __syncthreads();

    // ---- backtrack (core.pyx:31-35), wave-uniform state kept in SGPRs ----
    int index = tx - 1;
    for (int blk = nblk - 1; blk >= 0; --blk) {
        const int i0 = index;
        const int row = i0 - lane;
        unsigned int W = 0u;
        if (lane <= 32 && row >= 0) W = dec[(blk * R + (row % R)) * 64 + (row / R)];
        int myidx = -1;
        for (int yy = 31; yy >= 0; --yy) {
            const int y = blk * 32 + yy;
            if (y >= ty) continue;
            if (lane == yy) myidx = index;                       // path[index][y] = 1
            const unsigned int w = __builtin_amdgcn_readlane(W, i0 - index);
            const bool mv = (index != 0) && (index == y || ((w >> yy) & 1u));
            index = __builtin_amdgcn_readfirstlane(index - (mv ? 1 : 0));
        }
        if (idx_b && lane < 32 && blk * 32 + lane < Ty) idx_b[blk * 32 + lane] = myidx;
    }
    if (idx_b) for (int y = nblk * 32 + lane; y < Ty; y += 64) idx_b[y] = -1;
}

template <typename T>
__global__ __launch_bounds__(256) void mas_path_kernel(const int32_t* __restrict__ idx, T* __restrict__ path,
                                                       int Tx, int Ty)
{
    // grid: (ceil(Ty/1024), Tx, B); each thread writes 4 consecutive frames of one token row
    const int b = blockIdx.z, x = blockIdx.y;
    const int y0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (y0 >= Ty) return;
    const int32_t* ib = idx + (size_t)b * Ty;
    T* out = path + ((size_t)b * Tx + x) * Ty + y0;
    T v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (y0 + e < Ty && ib[y0 + e] == x) ? (T)1 : (T)0;
    if (y0 + 3 < Ty && (Ty & 3) == 0) {
        *reinterpret_cast<float4*>(out) = *reinterpret_cast<float4*>(v);      // T is 4 bytes wide
    } else {
        for (int e = 0; e < 4 && y0 + e < Ty; ++e) out[e] = v[e];
    }
}

template <int R>
int launch_dp(const float* value, const int32_t* t_xs, const int32_t* t_ys, int32_t* idx_out, float* q_out,
              int B, int Tx, int Ty, float neg, hipStream_t s)
{
    const size_t lds = (size_t)((Ty + 31) / 32) * R * 64 * sizeof(unsigned int);
    if (lds > 160 * 1024) return GLOWTTS_E_ARG;
    const bool vec = (Ty % 4 == 0) && ((reinterpret_cast<uintptr_t>(value) & 15) == 0);
    auto kv = mas_dp_kernel<R, true>;
    auto ks = mas_dp_kernel<R, false>;
    if (lds > 48 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kv), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (vec) hipLaunchKernelGGL(kv, dim3(B), dim3(64), lds, s, value, t_xs, t_ys, idx_out, q_out, Tx, Ty, neg);
    else     hipLaunchKernelGGL(ks, dim3(B), dim3(64), lds, s, value, t_xs, t_ys, idx_out, q_out, Tx, Ty, neg);
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

}  // namespace

extern "C" int glowtts_mas_dp_f32(const float* value, const int32_t* t_xs, const int32_t* t_ys,
                                  int32_t* idx_out, float* q_out, int B, int Tx, int Ty,
                                  float max_neg_val, void* stream)
{
    if (!value || !t_xs || !t_ys || B < 0 || Tx < 1 || Ty < 1 || Tx > 512) return GLOWTTS_E_ARG;
    if (B == 0) return GLOWTTS_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int R = (Tx + 63) / 64;
    switch (R) {
        case 1: return launch_dp<1>(value, t_xs, t_ys, idx_out, q_out, B, Tx, Ty, max_neg_val, s);
        case 2: return launch_dp<2>(value, t_xs, t_ys, idx_out, q_out, B, Tx, Ty, max_neg_val, s);
        case 3: return launch_dp<3>(value, t_xs, t_ys, idx_out, q_out, B, Tx, Ty, max_neg_val, s);
        case 4: return launch_dp<4>(value, t_xs, t_ys, idx_out, q_out, B, Tx, Ty, max_neg_val, s);
        case 5: case 6: return launch_dp<6>(value, t_xs, t_ys, idx_out, q_out, B, Tx, Ty, max_neg_val, s);
        default: return launch_dp<8>(value, t_xs, t_ys, idx_out, q_out, B, Tx, Ty, max_neg_val, s);
    }
}

extern "C" int glowtts_mas_path_from_idx(const int32_t* idx, void* path, int B, int Tx, int Ty,
                                         int out_dtype, void* stream)
{
    if (!idx || !path || B < 0 || Tx < 1 || Ty < 1 || (out_dtype != 0 && out_dtype != 1)) return GLOWTTS_E_ARG;
    if (B == 0) return GLOWTTS_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    dim3 grid((Ty + 1023) / 1024, Tx, B);
    if (out_dtype == 0) hipLaunchKernelGGL(mas_path_kernel<int32_t>, grid, dim3(256), 0, s, idx, static_cast<int32_t*>(path), Tx, Ty);
    else                hipLaunchKernelGGL(mas_path_kernel<float>, grid, dim3(256), 0, s, idx, static_cast<float*>(path), Tx, Ty);
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

extern "C" int glowtts_mas_f32(const float* value, int32_t* path, const int32_t* t_xs, const int32_t* t_ys,
                               int32_t* scratch_idx, int B, int Tx, int Ty, float max_neg_val, void* stream)
{
    if (!path || !scratch_idx) return GLOWTTS_E_ARG;
    int rc = glowtts_mas_dp_f32(value, t_xs, t_ys, scratch_idx, nullptr, B, Tx, Ty, max_neg_val, stream);
    if (rc != GLOWTTS_OK) return rc;
    return glowtts_mas_path_from_idx(scratch_idx, path, B, Tx, Ty, 0, stream);
}

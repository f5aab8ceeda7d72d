// Monotonic Alignment Search for gfx950 - one wavefront per utterance.
//
// Replaces monotonic_align/core.pyx:9-45 (reference).  Same fp32 add/compare sequence per DP cell,
// so paths and cumulative scores are bit-exact with the C code Cython generates.
//
// Mapping (CDNA4, wave64):
//   * lane l owns R consecutive token rows x = l*R + j (R = ceil(Tx/64), compile-time 1..8); the
//     previous column of cumulative scores lives in R VGPRs.  The only cross-lane dependency,
//     Q[x-1][y-1] for j == 0, is one DPP wave_shr:1 move per column (no LDS, no barrier); lane 0
//     receives the "row -1" sentinel (0 at y == 0, max_neg_val after) through the DPP `old` operand.
//   * `value` is streamed straight from HBM into registers: 16 B (4 columns) per row per load through a
//     statically indexed ring of 16 chunk buffers, i.e. every load is issued 64 columns ahead of its
//     use (covers the HBM / Infinity-Cache latency with a single wave per CU).
//   * per cell: select(x == y) + compare/select max + add + 2 instructions for the back-pointer bit
//     (v_cmp into VCC, v_addc_co shifts it into a 32-column bit word).  Cells outside the band are
//     computed too (never read by cells inside it), so no band predicate sits on the critical path.
//   * back-pointers: 1 bit per cell, parked in LDS per 32-column block (Ty*R*8 bytes instead of the
//     4*Tx*Ty-byte matrix the reference re-reads).
//   * backtrack runs on the scalar unit: per 32-column block lanes 0..32 fetch the bit words of the 33
//     rows the path can visit; the walk jumps from move to move with s_ff1 (find-first-bit), i.e.
//     ~(t_x + t_y/32) dependent steps instead of t_y.
//   * the dense 0/1 path the reference API returns is written by a second, fully parallel kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/glowtts_hip.h"
#include "launch_log.h"
#include "tunable.h"
#include <algorithm>
#include "mas_common.h"

namespace {

__device__ __forceinline__ float wave_shr1(float v, float lane0_value) {
    // lane l receives lane l-1's v; lane 0 keeps `old` = lane0_value (bound_ctrl = false)
    int r = __builtin_amdgcn_update_dpp(__float_as_int(lane0_value), __float_as_int(v), 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
    return __int_as_float(r);
}

// bits = (bits << 1) | (a < b)      (v_cmp -> VCC, v_addc_co: bits + bits + carry-in)
__device__ __forceinline__ void push_lt_bit(unsigned int& bits, float a, float b) {
    // (s_nop 1: on gfx950 a VALU that reads VCC written by a VALU needs two wait states; nothing inside an asm string is padded by hipcc)
    asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\ts_nop 1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "v"(a), "v"(b) : "vcc");
}

// Off the diagonal (x != y) the back-pointer test `Q[x][y-1] < Q[x-1][y-1]` (core.pyx:34) and the forward maximum `v_prev > v_cur`
// (core.pyx:30, v_cur = Q[x][y-1]) are the SAME comparison: one v_cmp feeds the bit push and the select.  Returns max as Cython writes it
// ((v_prev > v_cur) ? v_prev : v_cur): 3 VALU instructions instead of 5 per cell, on a loop that is bound by single-wave VALU issue.
__device__ __forceinline__ float push_lt_bit_and_max(unsigned int& bits, float q_cur, float v_prev) {
    float m;
    // (the select first: v_addc_co overwrites VCC with its carry-out)
    asm volatile("v_cmp_lt_f32 vcc, %2, %3\n\ts_nop 1\n\tv_cndmask_b32 %1, %2, %3, vcc\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                 : "+v"(bits), "=&v"(m) : "v"(q_cur), "v"(v_prev) : "vcc");
    return m;
}

template <int R, bool VEC4, bool WRITEQ, bool TR>
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
    const float* vb = value + (size_t)b * Tx * Ty;
    if (mas_degenerate<TR>(vb, idx_b, tx, ty, Tx, Ty, lane)) return;
    float* qb = WRITEQ ? q_out + (size_t)b * Tx * Ty : nullptr;

    // row pointers (rows >= Tx are clamped: they are never inside the band)
    const float* rowp[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        int x = lane * R + j;
        rowp[j] = vb + (size_t)(x < Tx ? x : Tx - 1) * Ty;
    }

    // one chunk = 4 columns x R rows (16 B per row per lane)
    // TR: value is stored transposed, [Ty][Tx] (token index contiguous): one column is one coalesced row
    const int xt = min(lane * R, Tx - R > 0 ? Tx - R : 0);      // first token row of this lane, clamped (TR + VEC4 path)
    auto load_chunk = [&](float4 (&dst)[R], int chunk) {
        // unconditional loads with clamped addresses: a predicated load would need a select after it,
        // i.e. an s_waitcnt right behind the load, which destroys the look-ahead.  Columns >= Ty hold
        // garbage that no cell inside an utterance ever reads.
        if (TR) {
            float v[4][R];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float* col = vb + (uint32_t)min(chunk * 4 + e, Ty - 1) * (uint32_t)Tx;       // (Tx * Ty < 2^31 elements per utterance: host-checked)
                if (VEC4 && R == 2) { const float2 t = *reinterpret_cast<const float2*>(col + xt); v[e][0] = t.x; v[e][R - 1] = t.y; }
                else if (VEC4 && R == 4) { const float4 t = *reinterpret_cast<const float4*>(col + xt); v[e][0] = t.x; v[e][1 % R] = t.y; v[e][2 % R] = t.z; v[e][3 % R] = t.w; }
                else {
#pragma unroll
                    for (int j = 0; j < R; ++j) v[e][j] = col[min(lane * R + j, Tx - 1)];
                }
            }
#pragma unroll
            for (int j = 0; j < R; ++j) dst[j] = make_float4(v[0][j], v[1][j], v[2][j], v[3][j]);
            return;
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {
            if (VEC4) {
                const int y0 = min(chunk * 4, Ty - 4);
                dst[j] = *reinterpret_cast<const float4*>(rowp[j] + y0);
            } else {
                const int y0 = chunk * 4;
                dst[j].x = rowp[j][min(y0 + 0, Ty - 1)];
                dst[j].y = rowp[j][min(y0 + 1, Ty - 1)];
                dst[j].z = rowp[j][min(y0 + 2, Ty - 1)];
                dst[j].w = rowp[j][min(y0 + 3, Ty - 1)];
            }
        }
    };

    float q[R];
    unsigned int bits[R];
#pragma unroll
    for (int j = 0; j < R; ++j) { q[j] = 0.f; bits[j] = 0u; }

    float upkeep = neg;
    // DIAG: the chunk may hold cells on the diagonal x == y (columns < Tx); beyond it the cheaper fused compare is exact
    auto compute_chunk = [&](const float4 (&cur)[R], int chunk, auto DIAG_) {
        constexpr bool DIAG = decltype(DIAG_)::value;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int y = chunk * 4 + e;                                          // wave-uniform
            // Q[x-1][y-1] from the lane below; lane 0 (row -1, core.pyx:23-27) keeps `old`: 0 at y = 0, else neg.  Off the diagonal region `old`
            // is the previous shift's result - its lane 0 still holds neg, the other lanes are overwritten - so the DPP move needs no
            // re-initialised destination (one VALU less per column).
            float up;
            if constexpr (DIAG) up = wave_shr1(q[R - 1], y == 0 ? 0.f : neg);
            else { up = wave_shr1(q[R - 1], upkeep); upkeep = up; }
            float qo[R];
#pragma unroll
            for (int j = 0; j < R; ++j) qo[j] = q[j];
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const int x = lane * R + j;
                const float v_prev = (j == 0) ? up : qo[j - 1];                   // core.pyx:23-29
                float m;
                if constexpr (DIAG) {
                    const float v_cur = (x == y) ? neg : qo[j];                   // core.pyx:19-22
                    push_lt_bit(bits[j], qo[j], v_prev);                          // core.pyx:34 test, kept for the backtrack
                    m = (v_prev > v_cur) ? v_prev : v_cur;                        // Cython max(a,b) = b>a ? b : a
                } else {
                    m = push_lt_bit_and_max(bits[j], qo[j], v_prev);
                }
                const float val = (e == 0) ? cur[j].x : (e == 1) ? cur[j].y : (e == 2) ? cur[j].z : cur[j].w;
                q[j] = m + val;                                                   // core.pyx:30
                if (WRITEQ) {
                    const bool inb = (y < ty) && (x >= max(0, tx + y - ty)) && (x < min(tx, y + 1));   // core.pyx:18
                    if (inb) qb[TR ? (size_t)y * Tx + x : (size_t)x * Ty + y] = q[j];
                }
            }
        }
    };

    // ring of 16 chunk buffers = 64 columns of look-ahead; statically indexed (fully unrolled)
    constexpr int D = 16;
    float4 ring[D][R];
#pragma unroll
    for (int c = 0; c < D; ++c) load_chunk(ring[c], c);
    const int niter = (ty + 63) >> 6;
    // 64 columns per iteration; the first ceil(Tx / 64) iterations may touch the diagonal, the others take the cheaper cell update
    auto iteration = [&](int it, auto DIAG_) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < D; ++c) {
            float4 cur[R];
#pragma unroll
            for (int j = 0; j < R; ++j) cur[j] = ring[c][j];
            load_chunk(ring[c], (it + 1) * D + c);                               // refill this slot, 64 columns ahead
            compute_chunk(cur, it * D + c, DIAG_);
            if ((c & 7) == 7) {                                                  // 32 columns done: park the bit words
                const int blk = it * 2 + (c >> 3);
#pragma unroll
                for (int j = 0; j < R; ++j) { dec[(blk * R + j) * 64 + lane] = bits[j]; bits[j] = 0u; }   // bit (31 - c) <-> column blk*32 + c
            }
        }
    };
    const int it_diag = min(niter, (Tx + 63) >> 6);
    int it = 0;
    for (; it < it_diag; ++it) iteration(it, std::true_type{});
    for (; it < niter; ++it) iteration(it, std::false_type{});
    mas_backtrack<R>(dec, idx_b, tx, ty, Ty, lane);
}

template <typename T>
__global__ __launch_bounds__(256) void mas_path_kernel(const int32_t* __restrict__ idx, T* __restrict__ path,
                                                       int Tx, int Ty, int vec_ok, int rows_per_block)
{
    // grid: (ceil(Tx / rows_per_block), B); a block writes rows_per_block token rows x all frames; each thread 4 consecutive frames
    const int b = blockIdx.y;
    const int x0 = blockIdx.x * rows_per_block;
    const int x1 = min(Tx, x0 + rows_per_block);
    const int32_t* ib = idx + (size_t)b * Ty;
    for (int y0 = threadIdx.x * 4; y0 < Ty; y0 += 1024) {
        int id[4];
        if (vec_ok && y0 + 3 < Ty) {
            const int4 v = *reinterpret_cast<const int4*>(ib + y0);
            id[0] = v.x; id[1] = v.y; id[2] = v.z; id[3] = v.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) id[e] = (y0 + e < Ty) ? ib[y0 + e] : -1;
        }
        T* out = path + ((size_t)b * Tx + x0) * Ty + y0;
        for (int x = x0; x < x1; ++x, out += Ty) {
            T v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (id[e] == x) ? (T)1 : (T)0;
            if (vec_ok && y0 + 3 < Ty) {
                *reinterpret_cast<float4*>(out) = *reinterpret_cast<float4*>(v);      // T is 4 bytes wide
            } else {
                for (int e = 0; e < 4 && y0 + e < Ty; ++e) out[e] = v[e];
            }
        }
    }
}

// The dense path when every row is a whole number of 16-byte groups: a linear grid-stride sweep over the output (the store pattern of a
// fill), the frame's token index re-read from L2 per group.  12.3 MB at B = 32, 120 x 800: 17.5 us with the row-block kernel above
// (a dependent load, then 8 strided stores per thread), measured against 3.7 us for a plain 12 MiB fill on the same chip.
template <typename T>
__global__ __launch_bounds__(256) void mas_path_linear_kernel(const int32_t* __restrict__ idx, T* __restrict__ path, int Tx, int Ty, unsigned int total4)
{
    const unsigned int q = (unsigned int)Ty >> 2;                // 16-byte groups per row
    for (unsigned int i = blockIdx.x * 256u + threadIdx.x; i < total4; i += gridDim.x * 256u) {
        const unsigned int row = i / q, y4 = i - row * q;        // row = b * Tx + x
        const unsigned int b = row / (unsigned int)Tx, x = row - b * (unsigned int)Tx;
        const int4 id = *reinterpret_cast<const int4*>(idx + (size_t)b * Ty + (size_t)y4 * 4);
        T v[4];
        v[0] = id.x == (int)x ? (T)1 : (T)0; v[1] = id.y == (int)x ? (T)1 : (T)0; v[2] = id.z == (int)x ? (T)1 : (T)0; v[3] = id.w == (int)x ? (T)1 : (T)0;
        *reinterpret_cast<float4*>(path + (size_t)i * 4) = *reinterpret_cast<float4*>(v);
    }
}

template <int R>
int launch_dp(const float* value, const int32_t* t_xs, const int32_t* t_ys, int32_t* idx_out, float* q_out,
              int B, int Tx, int Ty, float neg, bool transposed, hipStream_t s)
{
    const size_t lds = (size_t)((Ty + 63) / 64) * 2 * R * 64 * sizeof(unsigned int);
    if (lds > 160 * 1024) return GLOWTTS_E_ARG;
    const bool al = (reinterpret_cast<uintptr_t>(value) & 15) == 0;
    void (*k)(const float*, const int32_t*, const int32_t*, int32_t*, float*, int, int, float);
    bool vec_used;
    if (transposed && R == 2 && al && (Tx % 2 == 0) && (int64_t)Tx * Ty * 4 < ((int64_t)1 << 31) && GLOWTTS_TUNABLE("GLOWTTS_MAS_DP2", 1)) {
        return glowtts_detail::launch_mas_dp2(value, t_xs, t_ys, idx_out, q_out, B, Tx, Ty, neg, lds, s);
    }
    if (transposed) {
        const bool vec = al && (R == 2 || R == 4) && (Tx % R == 0) && (((size_t)Tx * sizeof(float)) % (R * sizeof(float)) == 0) && Tx >= R;
        vec_used = vec;
        if (q_out) k = vec ? mas_dp_kernel<R, true, true, true> : mas_dp_kernel<R, false, true, true>;
        else       k = vec ? mas_dp_kernel<R, true, false, true> : mas_dp_kernel<R, false, false, true>;
    } else {
        const bool vec = al && (Ty % 4 == 0);
        vec_used = vec;
        if (q_out) k = vec ? mas_dp_kernel<R, true, true, false> : mas_dp_kernel<R, false, true, false>;
        else       k = vec ? mas_dp_kernel<R, true, false, false> : mas_dp_kernel<R, false, false, false>;
    }
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    GLOWTTS_NOTE("mas_dp<R%d,%s,%s,%s>", R, vec_used ? "vec" : "scalar", q_out ? "q" : "noq", transposed ? "t" : "n");
    hipLaunchKernelGGL(k, dim3(B), dim3(64), lds, s, value, t_xs, t_ys, idx_out, q_out, Tx, Ty, neg);
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

int dispatch_dp(const float* value, const int32_t* t_xs, const int32_t* t_ys, int32_t* idx_out, float* q_out,
                int B, int Tx, int Ty, float neg, bool tr, void* stream)
{
    if (!value || !t_xs || !t_ys || B < 0 || Tx < 1 || Ty < 1 || Tx > 512 || (int64_t)Tx * Ty >= ((int64_t)1 << 31)) return GLOWTTS_E_ARG;
    if (B == 0) return GLOWTTS_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int R = (Tx + 63) / 64;
    switch (R) {
        case 1: return launch_dp<1>(value, t_xs, t_ys, idx_out, q_out, B, Tx, Ty, neg, tr, s);
        case 2: return launch_dp<2>(value, t_xs, t_ys, idx_out, q_out, B, Tx, Ty, neg, tr, s);
        case 3: return launch_dp<3>(value, t_xs, t_ys, idx_out, q_out, B, Tx, Ty, neg, tr, s);
        case 4: return launch_dp<4>(value, t_xs, t_ys, idx_out, q_out, B, Tx, Ty, neg, tr, s);
        case 5: case 6: return launch_dp<6>(value, t_xs, t_ys, idx_out, q_out, B, Tx, Ty, neg, tr, s);
        default: return launch_dp<8>(value, t_xs, t_ys, idx_out, q_out, B, Tx, Ty, neg, tr, s);
    }
}

}  // namespace

extern "C" int glowtts_mas_dp_f32(const float* value, const int32_t* t_xs, const int32_t* t_ys,
                                  int32_t* idx_out, float* q_out, int B, int Tx, int Ty,
                                  float max_neg_val, void* stream)
{
    return dispatch_dp(value, t_xs, t_ys, idx_out, q_out, B, Tx, Ty, max_neg_val, false, stream);
}

extern "C" int glowtts_mas_dp_f32_t(const float* value_t, const int32_t* t_xs, const int32_t* t_ys,
                                    int32_t* idx_out, float* q_out_t, int B, int Tx, int Ty,
                                    float max_neg_val, void* stream)
{
    return dispatch_dp(value_t, t_xs, t_ys, idx_out, q_out_t, B, Tx, Ty, max_neg_val, true, stream);
}

extern "C" int glowtts_mas_path_from_idx(const int32_t* idx, void* path, int B, int Tx, int Ty,
                                         int out_dtype, void* stream)
{
    if (!idx || !path || B < 0 || Tx < 1 || Ty < 1 || (out_dtype != 0 && out_dtype != 1)) return GLOWTTS_E_ARG;
    if (B == 0) return GLOWTTS_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint64_t total4 = (uint64_t)B * Tx * Ty / 4;
    if ((Ty % 4 == 0) && ((reinterpret_cast<uintptr_t>(path) & 15) == 0) && ((reinterpret_cast<uintptr_t>(idx) & 15) == 0) && total4 < (1ull << 31) &&
        GLOWTTS_TUNABLE("GLOWTTS_PATH_LINEAR", 1)) {
        const unsigned int blocks = (unsigned int)std::min<uint64_t>((total4 + 255) / 256, 8192);
        if (out_dtype == 0) hipLaunchKernelGGL(mas_path_linear_kernel<int32_t>, dim3(blocks), dim3(256), 0, s, idx, static_cast<int32_t*>(path), Tx, Ty, (unsigned int)total4);
        else                hipLaunchKernelGGL(mas_path_linear_kernel<float>, dim3(blocks), dim3(256), 0, s, idx, static_cast<float*>(path), Tx, Ty, (unsigned int)total4);
        return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
    }
    const int rpb = std::max(1, GLOWTTS_TUNABLE("GLOWTTS_PATH_ROWS", 8));
    dim3 grid((Tx + rpb - 1) / rpb, B);
    const int vec_ok = (Ty % 4 == 0) && ((reinterpret_cast<uintptr_t>(path) & 15) == 0) && ((reinterpret_cast<uintptr_t>(idx) & 15) == 0);
    if (out_dtype == 0) hipLaunchKernelGGL(mas_path_kernel<int32_t>, grid, dim3(256), 0, s, idx, static_cast<int32_t*>(path), Tx, Ty, vec_ok, rpb);
    else                hipLaunchKernelGGL(mas_path_kernel<float>, grid, dim3(256), 0, s, idx, static_cast<float*>(path), Tx, Ty, vec_ok, rpb);
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

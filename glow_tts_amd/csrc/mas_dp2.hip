// Monotonic Alignment Search, the production shape: transposed scores [B][Ty][Tx] with up to 128 token rows (two per lane), one
// wavefront per utterance.  Same recurrence, bit words and backtrack as mas.hip (replaces monotonic_align/core.pyx:9-45, bit-exact);
// what differs is the column step: hand-scheduled, 9 VALU instructions, operands through a buffer descriptor.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/glowtts_hip.h"
#include "launch_log.h"
#include "mas_common.h"

// tools/mas_lab.hip includes this file with its own stamp hook (phase timing of one workgroup); the library has none
#ifndef MAS_LAB_STAMP
#define MAS_LAB_STAMP(i)
#endif

namespace {

// One DP column, hand-scheduled: the two rows of a lane are independent chains (compare -> select -> add) that interleave; the second
// row's compare goes to an SGPR pair so both chains keep their masks; the back-pointer pushes (v_addc_co) sit behind the adds, off the
// recurrence.  Same fp32 compare / select / add per cell as mas_dp_kernel (bit-exact).  `up` is Q[x-1][y-1] for the lane's first row:
// wave_shr:1 of the second rows; lane 0 is not written by the DPP move and keeps the "row -1" sentinel (core.pyx:23-27).
// gfx950 wait states, all met by the order of the nine instructions (nothing is padded): a VALU that reads an SGPR pair / VCC written by a
// VALU needs two instructions in between (hipcc pads its own v_cmp -> v_cndmask with s_nop 1); the DPP read of q1 comes five VALU
// instructions after its write (two required; PAD: the caller has just written q1 itself, one more wait state in front).
__device__ __forceinline__ void dp2_column(float& q0, float& q1, unsigned int& b0, unsigned int& b1, float& up, float v0, float v1, bool PAD = false)
{
    float m1;
    unsigned long long s;
    if (PAD) asm volatile("s_nop 0");
    asm volatile("v_cmp_lt_f32_e64 %6, %1, %0\n\t"                                         // s   = Q[x1] < Q[x0]       (row 1: v_prev is row 0)
                 "v_mov_b32_dpp %4, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"          // up  = Q[x0 - 1]
                 "v_cmp_lt_f32_e32 vcc, %0, %4\n\t"                                        // vcc = Q[x0] < up
                 "v_cndmask_b32_e64 %5, %1, %0, %6\n\t"                                    // m1  = s ? Q[x0] : Q[x1]
                 "v_add_f32_e32 %1, %5, %8\n\t"                                            // Q[x1] = m1 + value
                 "v_cndmask_b32_e32 %0, %0, %4, vcc\n\t"                                   // m0  = vcc ? up : Q[x0]
                 "v_addc_co_u32_e64 %3, %6, %3, %3, %6\n\t"                                // bits1 = bits1 * 2 + s
                 "v_add_f32_e32 %0, %0, %7\n\t"                                            // Q[x0] = m0 + value
                 "v_addc_co_u32_e32 %2, vcc, %2, %2, vcc"                                  // bits0 = bits0 * 2 + vcc
                 : "+v"(q0), "+v"(q1), "+v"(b0), "+v"(b1), "+v"(up), "=&v"(m1), "=&s"(s) : "v"(v0), "v"(v1) : "vcc");
}

// Backtrack (core.pyx:31-35) over the parked bit words dec[blk][row & 1][row >> 1] (bit 31 - c <-> column blk*32 + c), one 32-column block
// at a time, walking only the MOVES on the scalar unit: lane j holds the word of row i0 - j (i0 = the row the block is entered on), with
// the forced move of the diagonal (index == y, core.pyx:34) OR-ed in as one more bit; the word of the current row is masked to the
// columns still ahead, s_ff1 finds the move, the moves of a block are collected in one SGPR mask and the token index of its 32 frames is a
// popcount per lane afterwards.  The words of the next block (rows i0 .. i0 - 63 of the current entry row) are fetched from LDS while
// the current block is walked.
__device__ __forceinline__ void mas_backtrack2(const unsigned int* dec, int32_t* idx_b, int tx, int ty, int Ty, int lane)
{
    const int nblk = (ty + 31) >> 5;
    __syncthreads();
    int index = tx - 1;
    int blk = nblk - 1;
    // lane j fetches the word of row base - j in block bk (row 0 never moves and rows below it do not exist: 0).  The LDS read is issued
    // by hand and waited for by hand one block later (hipcc would wait for it right behind the issue).
    const uint32_t dec0 = (uint32_t)reinterpret_cast<uintptr_t>(dec);
    auto issue = [&](int bk, int base, unsigned int& raw) -> bool {
        const int row = base - lane;
        const bool ok = bk >= 0 && row >= 1;
        const uint32_t addr = dec0 + (ok ? (uint32_t)((bk * 2 + (row & 1)) * 64 + (row >> 1)) * 4u : 0u);
        asm volatile("ds_read_b32 %0, %1" : "=v"(raw) : "v"(addr));
        return ok;
    };
    int base = index;
    unsigned int raw;
    bool ok = issue(blk, base, raw);
    while (blk >= 0) {
        const int i0 = index;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw));
        unsigned int W = ok ? raw : 0u;
        {   // row r = base - lane sits on the diagonal at column r of the utterance: bit 31 - (r - blk*32) if that column is in this block
            const int r = base - lane, cd = r - blk * 32;
            if (r >= 1 && cd >= 0 && cd <= 31) W |= 0x80000000u >> cd;
        }
        unsigned int raw_n;
        const bool ok_n = issue(blk - 1, i0, raw_n);          // next block's words, relative to this block's entry row
        int sel = base - i0;                                  // lane of row `index` in W
        const int yy = min(31, ty - 1 - blk * 32);            // highest column of this block inside the utterance
        unsigned int mask = 0xFFFFFFFFu << (31 - yy);         // bit p <-> column 31 - p: columns <= yy
        // per move, branch-free: the row's word masked to the columns still ahead; s_ff1 = the highest such column whose bit is set (the
        // move); the columns below it remain.  No bit left (it stays on this row for the rest of the block): s_ff1 gives -1, the shift by
        // 31 clears the mask and every later move of the block is a no-op (its s_bitset1_b64 lands in bit 63, the unused half).  The word
        // of the NEXT row is read (v_readlane -> SGPR) one move ahead.  Six scalar instructions per move, one branch per four moves.
        unsigned int wa, wb;
        unsigned long long moves64 = 0ull;
#define MAS_MOVE_(cur, nxt) "s_add_i32 %4, %4, 1\n\tv_readlane_b32 " nxt ", %5, %4\n\ts_and_b32 " cur ", " cur ", %2\n\t" \
                            "s_ff1_i32_b32 " cur ", " cur "\n\ts_lshl_b32 %2, -2, " cur "\n\ts_bitset1_b64 %3, " cur "\n\t"
        asm volatile("v_readlane_b32 %0, %5, %4\n\t"
                     "1:\n\t" MAS_MOVE_("%0", "%1") MAS_MOVE_("%1", "%0") MAS_MOVE_("%0", "%1") MAS_MOVE_("%1", "%0")
                     "s_cmp_lg_u32 %2, 0\n\ts_cbranch_scc1 1b"
                     : "=&s"(wa), "=&s"(wb), "+s"(mask), "+s"(moves64), "+s"(sel) : "v"(W) : "scc");
#undef MAS_MOVE_
        const unsigned int moves = (unsigned int)moves64;
        if (idx_b && lane < 32 && blk * 32 + lane < Ty) {
            const int below = __builtin_popcount(moves & ((1u << (31 - lane)) - 1u));     // moves at columns above this lane's column
            idx_b[blk * 32 + lane] = (blk * 32 + lane < ty) ? i0 - below : -1;
        }
        index = i0 - __builtin_popcount(moves);
        raw = raw_n; ok = ok_n; base = i0; blk -= 1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw));
    if (idx_b) for (int y = nblk * 32 + lane; y < Ty; y += 64) idx_b[y] = -1;
}

template <bool WRITEQ>
__global__ __launch_bounds__(64) void mas_dp2_kernel(const float* __restrict__ value,
                                                     const int32_t* __restrict__ t_xs,
                                                     const int32_t* __restrict__ t_ys,
                                                     int32_t* __restrict__ idx_out, float* q_out,
                                                     int Tx, int Ty, float neg)
{
    // value [B][Ty][Tx], Tx even and <= 128, 8-byte aligned, Tx * Ty * 4 < 2^31 (launch_dp2 checks); lane l owns rows 2l, 2l + 1
    extern __shared__ __attribute__((aligned(16))) unsigned int dec[];   // [nblk][2][64]
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    const int tx = t_xs[b], ty = t_ys[b];
    int32_t* idx_b = idx_out ? idx_out + (size_t)b * Ty : nullptr;
    const float* vb = value + (size_t)b * Tx * Ty;
    if (mas_degenerate<true>(vb, idx_b, tx, ty, Tx, Ty, lane)) return;
    float* qb = WRITEQ ? q_out + (size_t)b * Tx * Ty : nullptr;
    MAS_LAB_STAMP(0);

    // one column = one coalesced 8-byte load per lane through a buffer descriptor: lane offset in the VGPR, column offset in an SGPR
    // (soffset is not bounds-checked: columns are clamped to Ty - 1 where the look-ahead can pass the end of the tensor)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(vb), 0, Tx * Ty * 4, 0x00020000);
    const uint32_t voff = (uint32_t)min(lane * 2, Tx - 2) * 4u;       // rows >= Tx are clamped: they are never inside the band
    const int rowbytes = Tx * 4, lastoff = (Ty - 1) * rowbytes;
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    auto load_col = [&](int soff) -> float2 {
        const u32x2_t w = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
        return make_float2(__uint_as_float(w.x), __uint_as_float(w.y));
    };

    // statically indexed ring: every load is issued 64 columns ahead of its use
    float2 ring[64];
#pragma unroll
    for (int c = 0; c < 64; ++c) ring[c] = load_col(min(c * rowbytes, lastoff));
    float q0 = 0.f, q1 = 0.f, up = 0.f;          // up in lane 0: Q[-1][y-1] = 0 for the first column, max_neg_val after it
    unsigned int b0 = 0u, b1 = 0u;
    // DIAG: columns y < Tx hold a cell on the diagonal x == y, whose v_cur is max_neg_val (core.pyx:19-22): row y's stale value of column
    // y - 1 (above the diagonal, never read by a cell inside the band) is overwritten with it before the column is computed.
    auto iteration = [&](int it, auto DIAG_, auto CLAMP_) __attribute__((always_inline)) {
        constexpr bool DIAG = decltype(DIAG_)::value, CLAMP = decltype(CLAMP_)::value;
        int soff = (it + 1) * 64 * rowbytes;
#pragma unroll
        for (int c = 0; c < 64; ++c) {
            const int y = it * 64 + c;                                            // wave-uniform
            if constexpr (DIAG) {
                // row y's stale value <- max_neg_val: one select under a lane mask built on the scalar unit (rows >= Tx are never inside
                // the band, so columns >= Tx of these iterations may be patched as well)
                const unsigned long long diag_lane = 1ull << ((y >> 1) & 63);
                if (c & 1) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(q1) : "v"(neg), "s"(diag_lane));
                else       asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(q0) : "v"(neg), "s"(diag_lane));
            }
            dp2_column(q0, q1, b0, b1, up, ring[c].x, ring[c].y, DIAG && (c & 1));
            // refill behind the column step: the load may land in the registers the step has just read (no copies), 64 columns ahead of its use
            ring[c] = load_col(CLAMP ? min(soff, lastoff) : soff);
            soff += rowbytes;
            if constexpr (DIAG) {
                if (c == 0 && it == 0) up = (lane == 0) ? neg : up;
            }
            if (WRITEQ) {
                const int xlo = max(0, tx + y - ty), xhi = min(tx, y + 1);        // core.pyx:18
                const int x = lane * 2;
                if (y < ty && x >= xlo && x < xhi) qb[(size_t)y * Tx + x] = q0;
                if (y < ty && x + 1 >= xlo && x + 1 < xhi) qb[(size_t)y * Tx + x + 1] = q1;
            }
            if ((c & 31) == 31) {                                                 // 32 columns done: park the bit words
                const int blk = it * 2 + (c >> 5);
                dec[(blk * 2 + 0) * 64 + lane] = b0;                              // bit (31 - c) <-> column blk*32 + c
                dec[(blk * 2 + 1) * 64 + lane] = b1;
            }
        }
    };
    const int niter = (ty + 63) >> 6;
    const int it_diag = min(niter, (Tx + 63) >> 6);
    int it = 0;
    for (; it < it_diag; ++it) { MAS_LAB_STAMP(4 + it); iteration(it, std::true_type{}, std::true_type{}); }
    for (; it < niter && (it + 2) * 64 <= Ty; ++it) { MAS_LAB_STAMP(4 + it); iteration(it, std::false_type{}, std::false_type{}); }
    for (; it < niter; ++it) { MAS_LAB_STAMP(4 + it); iteration(it, std::false_type{}, std::true_type{}); }
    MAS_LAB_STAMP(1);
    mas_backtrack2(dec, idx_b, tx, ty, Ty, lane);
    MAS_LAB_STAMP(2);
}

}  // namespace

int glowtts_detail::launch_mas_dp2(const float* value, const int32_t* t_xs, const int32_t* t_ys, int32_t* idx_out, float* q_out,
                                   int B, int Tx, int Ty, float neg, size_t lds, hipStream_t s)
{
    auto k = q_out ? mas_dp2_kernel<true> : mas_dp2_kernel<false>;
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    GLOWTTS_NOTE("mas_dp2<%s>", q_out ? "q" : "noq");
    hipLaunchKernelGGL(k, dim3(B), dim3(64), lds, s, value, t_xs, t_ys, idx_out, q_out, Tx, Ty, neg);
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

// Backward of the fused coupling network (wavenet_fused.hip) for gfx950: the DATA gradients of one flow's Start conv, WaveNet and End conv
// (autograd of Modules.py:785-806, 858-887) in ONE launch.  The weight gradients stay with the grouped wgrad launches at the end of the
// decoder backward; this kernel leaves them their operands (d skip, the gate gradients dins_l, d x_l) in global memory.
//
//   d skip = (d(m, logs) W_end^T) * mask                                              [End^T:   K = 192, 3 slabs]
//   for l = L-1 .. 0:
//     d acts = [d x_{l+1} | d skip] W_rs_l^T   (last layer: d skip only)              [RS^T:    K = 384 / 192, 6 / 3 slabs]
//     (da, ds) = d acts * (s (1 - t^2), t s (1 - s)) -> dropout mask -> dins_l        (gates (t, s) kept by the forward)
//     d x_l = (conv5^T(dins_l) + d x_{l+1}) * mask                                    [In^T:    K = 5 x 384, 30 slabs]
//   d x_a += d x_0 W_start^T                                                          [Start^T: K = 192, 2 slabs]
//
// Same machine as the forward: 12 waves, a 64-row window per workgroup whose valid region shrinks by 2 rows per layer on each side (here
// from the last layer down; 52 owned rows for L = 4), operands of the k = 5 transposed conv in LDS with their halo, all transposed weights
// of the flow as one image of 24-KiB slabs streamed through an LDS ring by LDS-DMA.  What differs:
//   * every GEMM here has 192 output columns = 6 fragments x 2 row fragments = one 32 x 32 fragment per wave.  For the large one (In^T)
//     that tiling would read 2 KiB of LDS per MFMA; instead waves work in PAIRS over the two K chunks of a slab - wave (rf, cp, kh)
//     multiplies rows rf, columns [64 cp, 64 cp + 64) over chunk kh (1.5 KiB per MFMA, as in the forward) - and the partners swap halves
//     of their partial sums through LDS after the last slab; each wave ends up with exactly its fragment (rf, 2 cp + kh).
//   * LDS budget: d x_{l+1} (24 KiB), d skip (26 KiB) and the gate gradients with their halo (384 channels x 68 rows = 52 KiB) do not fit
//     beside a ring.  In^T therefore runs in two K passes over ONE 26-KiB tile: first the tanh-side gradients da (K chunks 0..5 of the
//     un-paired transposed image), then the tile is rewritten with the sigmoid-side gradients ds (chunks 6..11) from registers (the pair is
//     kept packed as bf16x2: 16 VGPRs).  The ring has three slots (two slabs in flight).
// Arithmetic and roundings are those of the per-conv backward (bf16 d skip, dins, d x_l; fp32 d x_0): results differ by accumulation order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "../../include/glowtts_hip.h"
#include "tunable.h"
#include "launch_log.h"
#include "wavenet_common.h"

namespace {

constexpr int BW_NS = 3;                                        // ring slots
// LDS map (bytes)
constexpr int BOFF_DX = 0;                                      // d x_{l+1}: [6][64][64] (window rows)
constexpr int BSZ_DX = WN_KCH * WN_WIN * 64;
constexpr int BOFF_DS = BOFF_DX + BSZ_DX;                       // d skip: [6][68][64] (tile rows = window rows - 2 .. + 66)
constexpr int BSZ_T = WN_KCH * WN_XR * 64;
constexpr int BOFF_DT = BOFF_DS + BSZ_T;                        // da or ds of the layer: [6][68][64]; before that: d(m, logs) rows
constexpr int BOFF_RING = BOFF_DT + BSZ_T;
constexpr int BOFF_MK = BOFF_RING + BW_NS * WN_SLAB;
constexpr int BOFF_UT = BOFF_MK + WN_XR * 4;                    // utterance of the 68 tile rows (conditioned models)
constexpr int BW_LDS = BOFF_UT + WN_XR * 4;
static_assert(BW_LDS <= 160 * 1024, "LDS budget");
static_assert(BSZ_T >= WN_NW * 8 * 256, "the partial-sum exchange (8 registers per wave and round) reuses the gate-gradient tile");

struct wn_bwd_args {
    int rows, rows_per_utt, L, C2;
    const unsigned char* wimg;                    // transposed weight image of the flow
    const void* douts_bf; int64_t ldo;            // [rows][ldo] bf16 PAIR-packed d(m, logs), pad columns zero
    const float* rowmask;
    const void* gates[WN_MAXL];                   // kept (t, s) pairs, bf16 [rows][2 H]
    float drop_p; uint32_t seed; const uint32_t* seed_ptr;
    void* dskip;                                  // out: bf16 [rows][H]
    void* dins[WN_MAXL]; int64_t ldin;            // out: bf16 [rows][ldin] PAIR-packed (da | ds per 32 channels)
    void* dh[WN_MAXL];                            // out: d x_l, l >= 1 bf16 [rows][H]; l = 0 fp32 [rows][H] (bf16 like the others when dh0_bf16)
    int dh0_bf16;
    float* dx; int64_t lddx;                      // in / out: [rows][lddx] fp32, channels [0, C2) += d x_a
    float* dcond; int64_t ldcond;                 // COND: d conditioning [utterances][ldcond], layer l at + l * 2 H; ACCUMULATED (atomic adds)
};

// COND: the per-utterance conditioning joins the gate pre-activation AFTER the dropout (Modules.py:861-866): its gradient is the sum over an
// utterance's rows of (da, ds) BEFORE the keep mask, which only exists here in registers.  Every workgroup adds the sums of its OWNED rows to
// dcond with atomic adds, one run per utterance (as the per-conv DGATE epilogue does: order-dependent in the last bits).
// ABL (tools builds only, tools/bench_wn.py): timing ablations - 1: no weight DMAs after the prologue, 2: no MFMAs, 4: no global stores (copy-outs, d x_0, d x_a),
// 8: no gate loads, 16: no partial-sum exchange, 32: no fragment reads in the In_l^T loop.  Wrong results by design.
template <bool DROP, bool COND, int ABL = 0>
__global__ __launch_bounds__(WN_NT) void wn_bwd_kernel(const wn_bwd_args p)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char wb_smem[];
    unsigned char* const DX = wb_smem + BOFF_DX;
    unsigned char* const DS = wb_smem + BOFF_DS;
    unsigned char* const DT = wb_smem + BOFF_DT;
    float* const MK = reinterpret_cast<float*>(wb_smem + BOFF_MK);
    int* const UT = reinterpret_cast<int*>(wb_smem + BOFF_UT);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int rf = wave >= 6 ? 1 : 0, cf = wave - rf * 6;      // this wave's 32 x 32 output fragment: row fragment, 32-channel block
    const int cp = cf >> 1, kh = cf & 1;                       // In^T: column pair of the wave pair / K chunk of this wave
    const int L = p.L;
    const int halo = WN_PAD * (L - 1);
    const int nvalid = WN_WIN - 2 * halo;
    const int v0 = blockIdx.x * nvalid;                        // first owned row
    const int t0 = v0 - halo;                                  // global row of window row 0
    const int xr0 = t0 - WN_PAD;                               // global row of tile row 0
    const int nslabs = 36 * L + 2;
    const int lim = (p.rows - v0) < nvalid ? (p.rows - v0) : nvalid;
    const int jch = cf * 32 + l31;                             // channel of this lane in 192-wide tensors

    // ---- weight stream (see wavenet_fused.hip): slab s -> ring slot s % 3, this wave's two 1-KiB units are rows [32 wave, 32 wave + 32) ----
    const int lrow = lane >> 2, qa = (lane & 3) ^ ((lane >> 4) & 3);
    const unsigned char* const wsrc = p.wimg + (uint32_t)((wave * 32 + lrow) * 64 + qa * 16);
    auto issue = [&](int s) __attribute__((always_inline)) {
        const unsigned char* src = wsrc + (size_t)s * WN_SLAB;
        unsigned char* dst = wb_smem + BOFF_RING + (s % BW_NS) * WN_SLAB + wave * 2048;
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src, (void __attribute__((address_space(3)))*)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + 1024), (void __attribute__((address_space(3)))*)(dst + 1024), 16, 0, 0);
    };
    int snext = 0;
    // slab `snext` has landed (one younger slab may fly), everyone is done with slab snext - 1 -> its slot; the last slab drains the ring.
    // (Conservative count: the wave also waits for its own earlier stores; the forward measured no gain from exact counts.)
    auto begin_step = [&]() __attribute__((always_inline)) -> const unsigned char* {
        if (snext + 1 < nslabs) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        return wb_smem + BOFF_RING + (snext % BW_NS) * WN_SLAB;
    };
    auto end_step = [&]() __attribute__((always_inline)) {
        if (!(ABL & 1) && snext + BW_NS - 1 < nslabs) issue(snext + BW_NS - 1);
        ++snext;
    };
    auto plain_barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto mfma_bf16 = [](const Chunk16& a, const Chunk16& b, const f32x16& c) __attribute__((always_inline)) -> f32x16 {
        if constexpr ((ABL & 2) != 0) { f32x16 r = c; asm volatile("" : "+v"(r) : "v"(a), "v"(b)); return r; }
        else return ::mfma_bf16(a, b, c);
    };

    // ---- prologue: d(m, logs) rows of the tile -> DT (A operand of End^T), row masks ----
    issue(0); issue(1);
    {
        const int per_row = (WN_H * 2) / 16;                   // 24 16-byte pieces per row (ldo = 192)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int idx = tid + k * WN_NT;
            if (idx < WN_XR * per_row) {
                const int i = idx / per_row, pc = idx - i * per_row;
                int g = xr0 + i;
                g = g < 0 ? 0 : (g >= p.rows ? p.rows - 1 : g);
                const Chunk16 v = *reinterpret_cast<const Chunk16*>(static_cast<const unsigned char*>(p.douts_bf) + ((int64_t)g * p.ldo) * 2 + pc * 16);
                *reinterpret_cast<Chunk16*>(DT + (pc >> 2) * (WN_XR * 64) + swz(i, pc & 3)) = v;
            }
        }
        if (tid < WN_XR) {
            int g = xr0 + tid;
            g = g < 0 ? 0 : (g >= p.rows ? p.rows - 1 : g);
            MK[tid] = p.rowmask[g];
            if (COND) UT[tid] = g / p.rows_per_utt;
        }
    }
    int bl[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) bl[s2] = swz(l31, 2 * s2 + lhi);
    const int offA = rf * 2048;

    // copy of the owned rows of an LDS tile [6 chunks][trows][64 B] to global rows: piece (chunk kc, slot q) of row r goes to
    // dst + r * row_bytes + off + kc * chunk_bytes + 16 q
    auto copy_out = [&](const unsigned char* tile, int trows, int row_off, void* dst, int row_bytes, int chunk_bytes, int off) __attribute__((always_inline)) {
        if constexpr ((ABL & 4) != 0) return;
        const Rsrc rd = mk_rsrc(dst, (long)p.rows * row_bytes);
        int tid_ = tid;
        asm volatile("" : "+v"(tid_));
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int idx = tid_ + k * WN_NT;
            const int r = idx / 24, pc = idx - r * 24;
            const bool ok = r < lim;
            const Chunk16 v = lds16(tile + (pc >> 2) * (trows * 64) + swz(row_off + (ok ? r : 0), pc & 3));
            __builtin_amdgcn_raw_buffer_store_b128(v, rd, ok ? (uint32_t)((v0 + r) * row_bytes + off + (pc >> 2) * chunk_bytes + (pc & 3) * 16) : OOB, 0, 0);
        }
    };
    auto tile_bases = [&](int rb, int (&tb)[2][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int og = 0; og < 2; ++og)
#pragma unroll
            for (int cy = 0; cy < 2; ++cy) tb[og][cy] = rb * 64 + ((((l31 >> 3) ^ (lhi + 2 * og + cy)) & 3) << 4) + (l31 & 7) * 2;
    };
#define WN_TOFF(tb, reg, extra) ((tb)[((reg) >> 2) & 1][(((reg) & 3) + (extra)) >> 2] + (frag_row(reg) + (extra)) * 64)

    f32x16 acc0, acc1;
    auto zero = [](f32x16& a) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = 0.f;
    };
    // one-fragment GEMM step over a slab of two K chunks x 192 columns: A tiles a0 / a1 (already offset to the wave's rows), third row
    // fragment (rows 64..95 of the same tiles) into acc1 when `third`
    auto mma192 = [&](const unsigned char* slot, const unsigned char* a0, const unsigned char* a1, bool third) __attribute__((always_inline)) {
        Chunk16 fa[2][2], fb[2][2], f3[2][2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                fa[c][s2] = lds16((c ? a1 : a0) + bl[s2]);
                fb[c][s2] = lds16(slot + c * 12288 + cf * 2048 + bl[s2]);
                if (third) f3[c][s2] = lds16((c ? a1 : a0) + (2 - rf) * 2048 + bl[s2]);
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                acc0 = mfma_bf16(fa[c][s2], fb[c][s2], acc0);
                if (third) acc1 = mfma_bf16(f3[c][s2], fb[c][s2], acc1);
            }
        __builtin_amdgcn_sched_barrier(0);
    };
    const bool w3 = wave < 6;                                   // waves 0..5 (rf = 0) also carry the third row fragment (tile rows 64..67) where 68 rows are needed

    // ================= End^T: d skip = (d(m, logs) W_end^T) * mask on the 68 tile rows =================
    zero(acc0); zero(acc1);
#pragma unroll 1
    for (int j = 0; j < 3; ++j) {
        const unsigned char* slot = begin_step();
        mma192(slot, DT + (2 * j) * (WN_XR * 64) + offA, DT + (2 * j + 1) * (WN_XR * 64) + offA, w3);
        end_step();
    }
    {
        int rb = rf * 32 + 4 * lhi;
        asm volatile("" : "+v"(rb));
        int tb[2][2];
        tile_bases(rb, tb);
        unsigned char* const dc = DS + cf * (WN_XR * 64);
        const float* const mk = MK + rb;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
            *reinterpret_cast<unsigned short*>(dc + WN_TOFF(tb, reg, 0)) = bf16_bits(acc0[reg] * mk[frag_row(reg)]);
        if (w3 && lhi == 0) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                *reinterpret_cast<unsigned short*>(dc + 64 * 64 + WN_TOFF(tb, reg, 0)) = bf16_bits(acc1[reg] * mk[64 + reg]);
        }
    }

    const uint32_t thr = drop_threshold(p.drop_p);
    const float ik = drop_inv_keep(thr);
    uint32_t seed0 = p.seed;
    if (DROP && p.seed_ptr) seed0 += *p.seed_ptr;
    const uint32_t jkey = drop_colkey((uint32_t)jch);

    // ================= layers, last to first =================
#pragma unroll 1
    for (int l = L - 1; l >= 0; --l) {
        const bool last = l == L - 1;
        // ---- RS_l^T: d acts = [d x_{l+1} | d skip] W^T.  Last layer: d skip only, on all 68 tile rows; else on the 64 window rows ----
        zero(acc0); zero(acc1);
        if (last) {
#pragma unroll 1
            for (int j = 0; j < 3; ++j) {
                const unsigned char* slot = begin_step();
                if (j == 0) copy_out(DS, WN_XR, halo + WN_PAD, p.dskip, WN_H * 2, 64, 0);            // d skip (kept: DY of the Res_Skip weight gradients)
                mma192(slot, DS + (2 * j) * (WN_XR * 64) + offA, DS + (2 * j + 1) * (WN_XR * 64) + offA, w3);
                end_step();
            }
        } else {
#pragma unroll 1
            for (int j = 0; j < 6; ++j) {
                const unsigned char* slot = begin_step();
                if (j == 0) copy_out(DX, WN_WIN, halo, pick4(p.dh, l + 1), WN_H * 2, 64, 0);           // d x_{l+1} (kept: DY of the Res_Skip weight gradient)
                // K chunks 0..5: d x_{l+1} (window rows), 6..11: d skip (window row r = tile row r + 2)
                const unsigned char* a0 = j < 3 ? DX + (2 * j) * (WN_WIN * 64) + offA : DS + (2 * j - 6) * (WN_XR * 64) + offA;
                const unsigned char* a1 = j < 3 ? DX + (2 * j + 1) * (WN_WIN * 64) + offA : DS + (2 * j - 5) * (WN_XR * 64) + offA;
                if (j < 3) mma192(slot, a0, a1, false);
                else {                                          // rows shifted by two: the swizzle phase changes, compute the offsets for row + 2
                    Chunk16 fa[2][2], fb[2][2];
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2) {
                            fa[c][s2] = lds16((c ? a1 : a0) - offA + swz(rf * 32 + l31 + WN_PAD, 2 * s2 + lhi));
                            fb[c][s2] = lds16(slot + c * 12288 + cf * 2048 + bl[s2]);
                        }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2) acc0 = mfma_bf16(fa[c][s2], fb[c][s2], acc0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                end_step();
            }
        }
        // ---- gate derivative (autograd of Modules.py:885-887 and of the dropout at :862): (da, ds) kept packed as bf16 pairs ----
        const int roff = last ? 0 : WN_PAD;                     // tile row of accumulator row 0
        uint32_t pk[16], pk3[4];
        {
            int rb = rf * 32 + 4 * lhi;
            asm volatile("" : "+v"(rb));
            const Rsrc rg = mk_rsrc(pick4(p.gates, l), (long)p.rows * (2 * WN_H * 2));
            const int g0 = xr0 + roff + rb;                     // global row of register 0
            const uint32_t rk0 = (uint32_t)g0 * 0x9E3779B1u + seed0 + (uint32_t)l;
            // conditioning gradient: running sums of the current utterance's owned rows (this lane's column pair)
            float sa = 0.f, ss = 0.f;
            int cur_u = -1;
            float* const dcl = COND ? p.dcond + (long)l * (2 * WN_H) + jch : nullptr;
            // flush: called by ALL lanes (it shuffles).  Lanes l and l + 32 hold the same column, rows 4 apart: when both close a run of the
            // same utterance the upper half hands its sums to the lower one, which alone issues the atomics.
            auto flush = [&](const bool need) __attribute__((always_inline)) {
                const int other_u = __shfl_xor(need ? cur_u : -2, 32, 64);
                const bool pair = need && other_u == cur_u;
                const float oa = __shfl_xor(sa, 32, 64), os = __shfl_xor(ss, 32, 64);
                if (pair) { sa += oa; ss += os; }
                if (need && cur_u >= 0 && !(pair && lhi)) {
                    float* dst = dcl + (long)cur_u * p.ldcond;
                    unsafeAtomicAdd(dst, sa); unsafeAtomicAdd(dst + WN_H, ss);
                }
                if (need) sa = ss = 0.f;
            };
            // (wave-uniform) all owned rows of this workgroup in one utterance - the rule, an utterance is hundreds of rows: no run logic
            const int own_lo = halo + WN_PAD, own_hi = own_lo + lim;                     // owned TILE rows [own_lo, own_hi)
            const bool one_utt = COND && lim > 0 && UT[halo + WN_PAD] == UT[halo + WN_PAD + lim - 1];
            auto gate = [&](float d, uint32_t w, int c) __attribute__((always_inline)) -> uint32_t {
                const float t = __uint_as_float(w << 16), sg = __uint_as_float(w & 0xFFFF0000u);
                const float dsg = d * sg;
                float da = dsg * (1.f - t * t), ds = dsg * t * (1.f - sg);
                if constexpr (COND) {
                    const int tr = roff + rb + c;                                  // tile row of this accumulator row
                    const bool own = tr >= own_lo && tr < own_hi;
                    if (one_utt) { sa += own ? da : 0.f; ss += own ? ds : 0.f; }
                    else {
                        const int u = own ? UT[tr < WN_XR ? tr : WN_XR - 1] : cur_u;
                        const bool need = own && u != cur_u;
                        if (__any(need)) flush(need);
                        if (need) cur_u = u;
                        sa += own ? da : 0.f; ss += own ? ds : 0.f;
                    }
                }
                if constexpr (DROP) {
                    uint32_t x = rk0 + (uint32_t)c * 0x9E3779B1u; x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13;
                    const uint32_t dd = drop_draw(x, jkey);
                    da *= drop_keep_lo(dd, thr, ik); ds *= drop_keep_hi(dd, thr, ik);
                }
                return pack_bf16x2(da, ds);
            };
            uint32_t gw[16];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {                // (rows outside the tensor: clamped garbage is fine, those rows are never stored or valid)
                int g = g0 + frag_row(reg);
                g = g < 0 ? 0 : (g >= p.rows ? p.rows - 1 : g);
                gw[reg] = (ABL & 8) ? 0x3f003e80u + (uint32_t)reg : __builtin_amdgcn_raw_buffer_load_b32(rg, (uint32_t)(g * (2 * WN_H * 2) + jch * 4), 0, 0);
            }
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) pk[reg] = gate(acc0[reg], gw[reg], frag_row(reg));
            if constexpr (COND) {
                if (one_utt) cur_u = UT[halo + WN_PAD];
                flush(true);
                cur_u = -1;                                     // (the third row fragment below holds tile rows 64..67: never owned)
            }
            if (last && w3) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    int g = g0 + 64 + reg;
                    g = g < 0 ? 0 : (g >= p.rows ? p.rows - 1 : g);
                    const uint32_t w = __builtin_amdgcn_raw_buffer_load_b32(rg, (uint32_t)(g * (2 * WN_H * 2) + jch * 4), 0, 0);
                    pk3[reg] = gate(acc1[reg], w, 64 + reg);
                }
            }
        }
        // the tanh-side half -> DT (rows roff ..), then In^T pass 0; the sigmoid-side half is written at the pass boundary
        auto write_half = [&](int h) __attribute__((always_inline)) {
            int rb = rf * 32 + 4 * lhi;
            asm volatile("" : "+v"(rb));
            int tb[2][2];
            tile_bases(rb, tb);
            unsigned char* const dc = DT + cf * (WN_XR * 64);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const unsigned short v = (unsigned short)(h ? pk[reg] >> 16 : pk[reg] & 0xFFFFu);
                if (last) *reinterpret_cast<unsigned short*>(dc + WN_TOFF(tb, reg, 0)) = v;
                else      *reinterpret_cast<unsigned short*>(dc + WN_TOFF(tb, reg, WN_PAD)) = v;
            }
            if (last && w3 && lhi == 0) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    *reinterpret_cast<unsigned short*>(dc + 64 * 64 + WN_TOFF(tb, reg, 0)) = (unsigned short)(h ? pk3[reg] >> 16 : pk3[reg] & 0xFFFFu);
            }
        };
        plain_barrier();                                        // every wave is done reading DT (End^T operand / the previous layer's exchange)
        write_half(0);

        // ---- In_l^T: two K passes (da, ds) x 5 taps x 3 slabs; wave (rf, cp, kh) multiplies chunk kh of every slab; reads one step ahead ----
        zero(acc0); zero(acc1);
        {
            Chunk16 fa[2][2] = {}, fb[2][2][2] = {};
            auto mma = [&](auto SET_) __attribute__((always_inline)) {
                constexpr int st = decltype(SET_)::value;
                acc0 = mfma_bf16(fa[st][0], fb[st][0][0], acc0);
                acc1 = mfma_bf16(fa[st][0], fb[st][0][1], acc1);
                acc0 = mfma_bf16(fa[st][1], fb[st][1][0], acc0);
                acc1 = mfma_bf16(fa[st][1], fb[st][1][1], acc1);
            };
            auto reads = [&](auto SET_, const unsigned char* slot, int n) __attribute__((always_inline)) {
                constexpr int st = decltype(SET_)::value;
                if constexpr ((ABL & 32) != 0) { asm volatile("" : "+v"(fa[st][0]), "+v"(fa[st][1]), "+v"(fb[st][0][0]), "+v"(fb[st][0][1]), "+v"(fb[st][1][0]), "+v"(fb[st][1][1])); return; }
                const int m = n >= 15 ? n - 15 : n, t = m / 3, jj = m - 3 * t;
                const unsigned char* At = DT + (2 * jj + kh) * (WN_XR * 64);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    fa[st][s2] = lds16(At + swz(rf * 32 + l31 + t, 2 * s2 + lhi));
                    fb[st][s2][0] = lds16(slot + kh * 12288 + cp * 4096 + bl[s2]);
                    fb[st][s2][1] = lds16(slot + kh * 12288 + cp * 4096 + 2048 + bl[s2]);
                }
            };
#pragma unroll 1
            for (int n2 = 0; n2 < 15; ++n2) {
                {   // even step n = 2 n2
                    const unsigned char* slot = begin_step();
                    if (n2 == 0) copy_out(DT, WN_XR, halo + WN_PAD, pick4(p.dins, l), (int)p.ldin * 2, 128, 0);       // da half of dins_l
                    reads(IC<0>{}, slot, 2 * n2);
                    if (n2 > 0) mma(IC<1>{});
                    end_step();
                }
                {   // odd step n = 2 n2 + 1; n = 15 opens the second pass: rewrite the tile with the sigmoid-side half
                    const unsigned char* slot = begin_step();
                    if (n2 == 7) {
                        write_half(1);
                        plain_barrier();
                        copy_out(DT, WN_XR, halo + WN_PAD, pick4(p.dins, l), (int)p.ldin * 2, 128, 64);              // ds half
                    }
                    reads(IC<1>{}, slot, 2 * n2 + 1);
                    mma(IC<0>{});
                    end_step();
                }
            }
            mma(IC<1>{});                                       // slab 29
        }
        // ---- partners swap halves of their partial sums (two rounds of 8 registers through the tile) ----
        {
            float* const xs = reinterpret_cast<float*>(DT);
#pragma unroll
            for (int rd = 0; rd < ((ABL & 16) ? 0 : 2); ++rd) {
                plain_barrier();                                // (round 0: every wave is done reading the tile; round 1: done reading round 0)
#pragma unroll
                for (int i = 0; i < 8; ++i) xs[(wave * 8 + i) * 64 + lane] = kh ? acc0[rd * 8 + i] : acc1[rd * 8 + i];
                plain_barrier();
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float o = xs[((wave ^ 1) * 8 + i) * 64 + lane];
                    if (kh) acc1[rd * 8 + i] += o; else acc0[rd * 8 + i] += o;
                }
            }
            if (kh) acc0 = acc1;                                // this wave's fragment (rf, cf = 2 cp + kh)
        }
        // ---- d x_l = (conv^T + d x_{l+1}) * mask -> DX (bf16, in place); l = 0: fp32 rows for the Start conv's weight gradient ----
        {
            int rb = rf * 32 + 4 * lhi;
            asm volatile("" : "+v"(rb));
            int tb[2][2];
            tile_bases(rb, tb);
            unsigned char* const xc = DX + cf * (WN_WIN * 64);
            const float* const mk = MK + rb + WN_PAD;
            const Rsrc r0 = mk_rsrc(p.dh[0], (l == 0 && !p.dh0_bf16) ? (long)p.rows * (WN_H * 4) : 0);
            const uint32_t v00 = (uint32_t)((t0 + rb) * (WN_H * 4) + jch * 4);
            const uint32_t own0 = (uint32_t)(rb - halo);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int c = frag_row(reg);
                unsigned short* xp = reinterpret_cast<unsigned short*>(xc + WN_TOFF(tb, reg, 0));
                const float xin = last ? 0.f : __uint_as_float((uint32_t)*xp << 16);
                const float v = (acc0[reg] + xin) * mk[c];
                *xp = bf16_bits(v);
                const bool ok = own0 + (uint32_t)c < (uint32_t)lim;
                if constexpr (!(ABL & 4)) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r0, ok ? v00 + (uint32_t)(c * (WN_H * 4)) : OOB, 0, 0);
            }
        }
    }

    // ================= Start^T: d x_a += d x_0 W_start^T (K = 192, 96 columns of a 128-column image: 2 slabs of 3 chunks) =================
    zero(acc0);
#pragma unroll 1
    for (int j = 0; j < 2; ++j) {
        const unsigned char* slot = begin_step();
        // d x_0 as bf16 rows (DY of the Start conv's weight gradient): the finished tile, 16 bytes per store, behind the step's barrier
        if (j == 0 && p.dh0_bf16) copy_out(DX, WN_WIN, halo, p.dh[0], WN_H * 2, 64, 0);
        if (cf < 3) {
            Chunk16 fa[3][2], fb[3][2];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    fa[c][s2] = lds16(DX + (3 * j + c) * (WN_WIN * 64) + offA + bl[s2]);
                    fb[c][s2] = lds16(slot + c * 8192 + cf * 2048 + bl[s2]);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) acc0 = mfma_bf16(fa[c][s2], fb[c][s2], acc0);
        }
        end_step();
    }
    if (cf < 3) {
        int rb = rf * 32 + 4 * lhi;
        asm volatile("" : "+v"(rb));
        const Rsrc rx = mk_rsrc(p.dx, (long)p.rows * p.lddx * 4);
        const bool cok = jch < p.C2;
        const uint32_t vx0 = (uint32_t)((t0 + rb) * (int)p.lddx + jch) * 4u;
        const uint32_t own0 = (uint32_t)(rb - halo);
        float old[16];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int c = frag_row(reg);
            const bool ok = cok && own0 + (uint32_t)c < (uint32_t)lim;
            old[reg] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, ok ? vx0 + (uint32_t)(c * (int)p.lddx * 4) : OOB, 0, 0));
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int c = frag_row(reg);
            const bool ok = cok && own0 + (uint32_t)c < (uint32_t)lim;
            if constexpr (!(ABL & 4)) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(old[reg] + acc0[reg]), rx, ok ? vx0 + (uint32_t)(c * (int)p.lddx * 4) : OOB, 0, 0);
        }
    }
#undef WN_TOFF
}

template <bool DROP, bool COND, int ABL = 0>
int launch_wn_bwd(const wn_bwd_args& k, dim3 grid, hipStream_t s)
{
#ifdef GLOWTTS_TOOLS
    if constexpr (ABL == 0 && DROP && !COND) {
        switch (GLOWTTS_TUNABLE("GLOWTTS_WN_BWD_ABL", 0)) {
            case 1: return launch_wn_bwd<DROP, COND, 1>(k, grid, s);
            case 2: return launch_wn_bwd<DROP, COND, 2>(k, grid, s);
            case 3: return launch_wn_bwd<DROP, COND, 3>(k, grid, s);
            case 4: return launch_wn_bwd<DROP, COND, 4>(k, grid, s);
            case 8: return launch_wn_bwd<DROP, COND, 8>(k, grid, s);
            case 12: return launch_wn_bwd<DROP, COND, 12>(k, grid, s);
            case 13: return launch_wn_bwd<DROP, COND, 13>(k, grid, s);
            case 14: return launch_wn_bwd<DROP, COND, 14>(k, grid, s);
            case 16: return launch_wn_bwd<DROP, COND, 16>(k, grid, s);
            case 32: return launch_wn_bwd<DROP, COND, 32>(k, grid, s);
            case 46: return launch_wn_bwd<DROP, COND, 46>(k, grid, s);
            case 47: return launch_wn_bwd<DROP, COND, 47>(k, grid, s);
            default: break;
        }
    }
#endif
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wn_bwd_kernel<DROP, COND, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return GLOWTTS_E_LAUNCH;
        attr_done = true;
    }
    GLOWTTS_NOTE_STATIC("wn_bwd<%s%s>", DROP ? "drop" : "nodrop", COND ? ",cond" : "");
    hipLaunchKernelGGL((wn_bwd_kernel<DROP, COND, ABL>), grid, dim3(WN_NT), BW_LDS, s, k);
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

}  // namespace

// the transposed images of F flows (see glowtts_wavenet_pack_images in include/glowtts_hip.h), in the order the backward consumes them
int glowtts_wavenet_pack_bwd_images(const float* w_start, const float* w_in, const float* w_rs, const float* w_rs_last, const float* w_end,
                                    int F, int L, int C2, void* img_bwd, void* stream)
{
    const int H = WN_H;
    const int64_t S = WN_SLAB, stride = (int64_t)(36 * L + 2) * S;
    unsigned char* img = static_cast<unsigned char*>(img_bwd);
    // [End^T 3][layer L-1: RS^T 3, In^T da 15, In^T ds 15][layer l < L-1: RS^T 6, da 15, ds 15 ...][Start^T 2]; In^T da of layer l starts at
    // slab 42 + 36 (L - 2 - l) for every l (6 for the last layer), RS^T of layer l < L - 1 six slabs before it
    int rc = glowtts_pack_weight_strided(w_end, F, 1, 2 * C2, H, 1, 1, GLOWTTS_PERM_PAIR, C2, GLOWTTS_BF16, img, stride, 0, 0, stream);
    if (rc == GLOWTTS_OK) rc = glowtts_pack_weight_strided(w_rs_last, F, 1, H, H, 1, 1, GLOWTTS_PERM_NONE, 0, GLOWTTS_BF16, img + 3 * S, stride, 0, 0, stream);
    if (rc == GLOWTTS_OK && L > 1)
        rc = glowtts_pack_weight_strided(w_rs, F * (L - 1), L - 1, 2 * H, H, 1, 1, GLOWTTS_PERM_NONE, 0, GLOWTTS_BF16, img + (36 + 36 * (int64_t)(L - 2)) * S, stride, -36 * S, 0, stream);
    for (int half = 0; half < 2 && rc == GLOWTTS_OK; ++half)
        rc = glowtts_pack_weight_strided(w_in + (int64_t)half * H * H * WN_TAPS, F * L, L, H, H, WN_TAPS, 1, GLOWTTS_PERM_NONE, 0, GLOWTTS_BF16,
                                         img + (42 + 36 * (int64_t)(L - 2) + 15 * half) * S, stride, -36 * S, (int64_t)2 * H * H * WN_TAPS, stream);
    if (rc == GLOWTTS_OK) rc = glowtts_pack_weight_strided(w_start, F, 1, H, C2, 1, 1, GLOWTTS_PERM_NONE, 0, GLOWTTS_BF16, img + 36 * (int64_t)L * S, stride, 0, 0, stream);
    return rc;
}

extern "C" int glowtts_wavenet_bwd(const glowtts_flow_dims* d, const glowtts_flow_params* p, const glowtts_flow_acts* a, const glowtts_flow_grads* g,
                                   void* stream)
{
    if (!d || !p || !a || !g || !p->wn_img_t || !a->rowmask || !g->dx || !g->douts_bf || !g->dskip) return GLOWTTS_E_ARG;
    const int C2 = d->C / 2;
    if (d->precision != GLOWTTS_BF16 || !d->act_bf16 || d->H != WN_H || d->ksize != WN_TAPS || d->L < 1 || d->L > WN_MAXL ||
        (d->C & 7) || C2 <= 64 || C2 > 96 || p->end.npad != 192 || p->in[0].npad != 2 * WN_H) return GLOWTTS_E_ARG;
    // inline weight gradients and the GR-mode per-row (pitch) conditioning gradient: per-conv path only
    if (!g->defer_wgrad || g->pitch_rows || p->cond_rows) return GLOWTTS_E_ARG;
    const bool cnd = g->dcond && p->cond;
    if (cnd && (int64_t)d->B * p->ldcond * 4 >= ((int64_t)1 << 31)) return GLOWTTS_E_ARG;
    const int Tp = d->T + 2 * GLOWTTS_ROW_PAD;
    const int64_t R = (int64_t)d->B * Tp;
    if (R * 2 * WN_H * 2 >= ((int64_t)1 << 31) || R * d->C * 4 >= ((int64_t)1 << 31)) return GLOWTTS_E_ARG;
    wn_bwd_args k;
    memset(&k, 0, sizeof(k));
    k.rows = (int)R; k.rows_per_utt = Tp; k.L = d->L; k.C2 = C2;
    k.wimg = static_cast<const unsigned char*>(p->wn_img_t);
    k.douts_bf = g->douts_bf; k.ldo = p->end.npad; k.rowmask = a->rowmask;
    k.drop_p = d->drop_p; k.seed = d->seed; k.seed_ptr = d->seed_ptr;
    k.dskip = g->dskip; k.ldin = p->in[0].npad; k.dx = g->dx; k.lddx = d->C;
    for (int l = 0; l < d->L; ++l) {
        if (!a->gates[l] || !g->dins[l] || !g->dh[l]) return GLOWTTS_E_ARG;
        k.gates[l] = a->gates[l]; k.dins[l] = g->dins[l]; k.dh[l] = g->dh[l];
        k.dh0_bf16 = g->dh0_bf16;
    }
    const int nvalid = WN_WIN - 2 * WN_PAD * (d->L - 1);
    const dim3 grid((unsigned)((R + nvalid - 1) / nvalid));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (cnd) { k.dcond = g->dcond; k.ldcond = p->ldcond; }
    if (d->drop_p > 0.f) return cnd ? launch_wn_bwd<true, true>(k, grid, s) : launch_wn_bwd<true, false>(k, grid, s);
    return cnd ? launch_wn_bwd<false, true>(k, grid, s) : launch_wn_bwd<false, false>(k, grid, s);
}

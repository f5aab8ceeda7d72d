// Fused coupling network of one decoder flow for gfx950: Start conv -> L x [In_l (k = 5) + dropout + conditioning + tanh * sigmoid ->
// Res_Skip_l + residual / skip] -> End conv + affine coupling (Modules.py:785-806, 858-887) in ONE launch.
//
// Why (round-2 profile): as ten launches per flow the chain sat at 0.12 of the MFMA peak - every kernel is a fetch
// burst, a short K loop and a store burst that overlap with nothing, and B = 32 leaves 12 928 rows, ~50 per CU.  Here one persistent
// workgroup per CU owns a row tile through the whole network:
//   * 12 waves (3 per SIMD).  A 64-row compute window per workgroup; the k = 5 taps make the valid region shrink by 2 rows per layer on each
//     side, so the window yields 64 - 4 (L - 1) = 52 valid output rows for L = 4 (recomputed halo: 1.23 x the MFMA work, 249 workgroups at
//     B = 32: one round on 256 CUs).  The WaveNet state x_l lives in LDS as bf16 ([6 K chunks][68 rows][64 B], swizzled like every MFMA
//     A tile of this library), tanh * sigmoid likewise; the skip sum stays in accumulator registers across the layers.
//   * ALL weights of the flow are one pre-packed image of 24 KiB slabs ([384 n][64 B] = one (tap, K chunk) of In_l, one K chunk of
//     Res_Skip_l, two K chunks of the 192-column convs), streamed in order through a 4-slot LDS ring by LDS-DMA (global_load_lds_dwordx4,
//     source-side swizzle), two 1-KiB units per wave and slab: one counted s_waitcnt vmcnt + one raw s_barrier per slab, two slabs in
//     flight while one is multiplied.  Per slab every wave issues 4 x v_mfma_f32_32x32x16_bf16 (32 rows x 64 columns x 32 k).
//     Every CU reads every weight byte (3.5 MB per flow and CU from L2): at ~50 rows per CU this path is co-bound by the matrix pipes and
//     by L2 -> LDS bandwidth (64 flop per weight byte), which is why the tile is not smaller than 64 rows.
//   * kept activations (training) leave through the epilogues: gate pairs straight from registers, x_l / tanh * sigmoid as 16-byte
//     pieces copied out of their LDS tiles while the next GEMM runs.
// Arithmetic is the unfused path's (same bf16 roundings of x_l, gates, acts, same dropout hash): results differ only by fp32
// accumulation order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "../../include/glowtts_hip.h"
#include "tunable.h"
#include "launch_log.h"
#include "wavenet_common.h"

namespace {

// LDS map (bytes)
constexpr int OFF_XT = 0;                                       // x_l: [6][68][64]; after the last layer: bf16 skip sum [6][64][64]
constexpr int SZ_XT = WN_KCH * WN_XR * 64;
constexpr int OFF_AT = OFF_XT + SZ_XT;                          // tanh * sigmoid: [6][64][64]; before layer 0: x_a [3][96][64]
constexpr int SZ_AT = WN_KCH * WN_WIN * 64;
constexpr int OFF_RING = OFF_AT + SZ_AT;
constexpr int OFF_BT = OFF_RING + WN_NS * WN_SLAB;              // biases (floats)
constexpr int BT_START = 0, BT_IN = 192, BT_RS = BT_IN + WN_MAXL * 384, BT_RSL = BT_RS + (WN_MAXL - 1) * 384, BT_END = BT_RSL + 192;
constexpr int BT_FLOATS = BT_END + 192;
constexpr int OFF_MK = OFF_BT + BT_FLOATS * 4;                  // rowmask of the 68 state rows
constexpr int OFF_UT = OFF_MK + WN_XR * 4;                      // conditioning row index of the 68 state rows
constexpr int WN_LDS = OFF_UT + WN_XR * 4;
static_assert(WN_LDS <= 160 * 1024, "LDS budget");
static_assert(SZ_AT >= 3 * WN_SROWS * 64, "the Start operand tile aliases the acts tile");

struct wn_fwd_args {
    int rows, rows_per_utt, L, C2, reverse, keep, safe_waits;
    const float* xsrc; int64_t ldx;               // [rows][ldx]: channels [0, C2) = x_a, [C2, 2 C2) = x_b
    float* xdst; int64_t ldxd;                    // x_b' -> xdst[r][C2 + j]
    const float* rowmask;
    const unsigned char* wimg;
    const float* b_start; const float* b_in[WN_MAXL]; const float* b_rs[WN_MAXL]; const float* b_end;
    const float* cond; int64_t ldcond; int cond_rows;
    float drop_p; uint32_t seed; const uint32_t* seed_ptr;
    void* hs[WN_MAXL]; void* gates[WN_MAXL]; void* acts[WN_MAXL];       // kept (bf16)
    float* skip; float* outs; int64_t ldo;                              // kept (fp32)
    long long* tl;                                                      // tools builds (ABL & 16): per-workgroup phase stamps [grid][32]
};

// DROP / COND: training-mode dropout / conditioning present (compile-time, so that the unrolled gate epilogue is straight-line code)
// ABL (tools builds only, tools/bench_wn.py): timing ablations - 1: no weight DMAs after the prologue, 2: no MFMAs, 4: no kept-activation stores
template <bool DROP, bool COND, int ABL = 0>
__global__ __launch_bounds__(WN_NT) void wn_fwd_kernel(const wn_fwd_args p)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char wn_smem[];
    unsigned char* const XT = wn_smem + OFF_XT;
    unsigned char* const AT = wn_smem + OFF_AT;
    float* const BT = reinterpret_cast<float*>(wn_smem + OFF_BT);
    float* const MK = reinterpret_cast<float*>(wn_smem + OFF_MK);
    int* const UT = reinterpret_cast<int*>(wn_smem + OFF_UT);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int rf = wave >= 6 ? 1 : 0, pi = wave - rf * 6;      // row fragment / column pair (64 packed columns) of this wave
    const int L = p.L;
    const int halo = WN_PAD * (L - 1);
    const int nvalid = WN_WIN - 2 * halo;                      // valid output rows of the window
    const int v0 = blockIdx.x * nvalid;                        // first valid (owned) row
    const int t0 = v0 - halo;                                  // global row of window row 0
    const int xr0 = t0 - WN_PAD;                               // global row of state-tile row 0
    const bool keep = p.keep != 0 && !(ABL & 4);

    // ---- weight stream: slab s -> ring slot s % 4; this wave's two 1-KiB units are rows [32 wave, 32 wave + 32) of the slab ----
    const int lrow = lane >> 2, qa = (lane & 3) ^ ((lane >> 4) & 3);
    // LD6 (experiment, ABL & 256): only waves 0..5 - the older half, which the arbiter favours and which otherwise idles ~480 clocks at every
    // slab barrier - issue DMAs (four units each); the younger half, last in line for the matrix pipe, is spared the ~300 clocks of DMA issue
    constexpr bool LD6 = (ABL & 256) != 0;
    constexpr int DPS = LD6 ? 4 : 2;                           // DMA instructions per slab of an issuing wave
    const unsigned char* const wsrc = p.wimg + (uint32_t)((wave * (16 * DPS) + lrow) * 64 + qa * 16);
    auto issue = [&](int s) __attribute__((always_inline)) {
        if (LD6 && wave >= 6) return;
        const unsigned char* src = wsrc + (size_t)s * WN_SLAB;
        unsigned char* dst = wn_smem + OFF_RING + (s & (WN_NS - 1)) * WN_SLAB + wave * (1024 * DPS);
#pragma unroll
        for (int u = 0; u < DPS; ++u)
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + u * 1024), (void __attribute__((address_space(3)))*)(dst + u * 1024), 16, 0, 0);
    };
    int snext = 0;                                             // slab being multiplied
    bool stamp_on = false; int stamp_base = (blockIdx.x * WN_NW + wave) * 32;      // (ABL & 32)
    int tli = 0;
    auto TLS = [&]() __attribute__((always_inline)) { if constexpr ((ABL & 16) != 0) { if (tid == 0) p.tl[blockIdx.x * 32 + tli] = (long long)__builtin_readcyclecounter(); ++tli; } };
    TLS();
    // A slab step: begin_step() = this wave's DMAs of slab `snext` have landed (the two slabs behind it may fly: vmcnt(4)), barrier
    // (everyone's have landed, everyone is done with slab snext - 1), -> its ring slot; ... fragment reads, MFMAs ...; end_step() refills
    // the slot the barrier freed with slab snext + 3 - issued BEHIND the step's MFMAs, so that the matrix pipe runs while the DMA
    // instructions issue.  The last three slabs (the End conv) drain the ring: waits 4 / 2 / 0, no refill.
    // X = vector-memory operations this wave has issued BEHIND the DMAs of slab snext besides the two younger slabs' four DMAs: the global
    // stores of the epilogue in front of this GEMM and of the copy-outs.  Memory operations retire in order (vmcnt counts loads and stores
    // alike on gfx9), so vmcnt(4 + X) waits for exactly slab snext; with a smaller count the wave would also wait for its own stores to be
    // acknowledged (measured: ~4 000 clocks at the head of every GEMM).  X must never exceed the real count: every counted operation is an
    // unconditional buffer instruction (invalid rows are dropped through an out-of-range offset, not branched around); `p.safe_waits`
    // (tests) runs the conservative vmcnt(4) everywhere and must give bit-identical results.
    auto begin_step = [&](auto X_) __attribute__((always_inline)) -> const unsigned char* {
        constexpr int X = decltype(X_)::value;
        if constexpr ((ABL & 32) != 0) {                       // per-wave step anatomy (tools/bench_wn.py): before wait / after wait / after barrier
            const bool on = stamp_on;
            if (on && lane == 0) p.tl[stamp_base + 0] = (long long)__builtin_readcyclecounter();
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(2 * DPS) : "memory");
            if (on && lane == 0) p.tl[stamp_base + 1] = (long long)__builtin_readcyclecounter();
            asm volatile("s_barrier" ::: "memory");
            if (on && lane == 0) p.tl[stamp_base + 2] = (long long)__builtin_readcyclecounter();
            if (on) stamp_base += 3;
            return wn_smem + OFF_RING + (snext & (WN_NS - 1)) * WN_SLAB;
        }
        if (X == 0 || p.safe_waits) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(2 * DPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(2 * DPS + X) : "memory");
        return wn_smem + OFF_RING + (snext & (WN_NS - 1)) * WN_SLAB;
    };
    constexpr int XG = 16;                                     // stores of a gate / last Res_Skip epilogue (one per accumulator register)
    constexpr int XC = 2;                                      // stores of a copy_out
    auto end_step = [&]() __attribute__((always_inline)) {
        if (!(ABL & 1)) issue(snext + WN_NS - 1);
        ++snext;
    };

    // ---- prologue: x_a -> bf16 operand tile of the Start conv, biases, row masks; the first three slabs stream in meanwhile ----
    {
        const int C2 = p.C2, per_row = C2 >> 2;
        unsigned char* const ST = AT;
        float4 xv[2]; int xi[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int idx = tid + k * WN_NT;
            xi[k] = idx < WN_XR * per_row ? idx : -1;
            const int i = idx / per_row, c4 = idx - i * per_row;
            int g = xr0 + i;
            g = g < 0 ? 0 : (g >= p.rows ? p.rows - 1 : g);
            xv[k] = xi[k] >= 0 ? *reinterpret_cast<const float4*>(p.xsrc + (int64_t)g * p.ldx + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // biases -> LDS table (every load is issued before the first use; unrolled over the layers: no dynamic indexing of the arguments)
        {
            const int i = tid;
            if (i < WN_H) { BT[BT_START + i] = p.b_start[i]; BT[BT_RSL + i] = pick4(p.b_rs, L - 1)[i]; BT[BT_END + i] = i < 2 * C2 ? p.b_end[i] : 0.f; }
            if (i < 2 * WN_H) {
#pragma unroll
                for (int l = 0; l < WN_MAXL; ++l) {
                    if (l < L) BT[BT_IN + l * 384 + i] = p.b_in[l][i];
                    if (l < L - 1) BT[BT_RS + l * 384 + i] = p.b_rs[l][i];
                }
            }
        }
        if (tid < WN_XR) {
            int g = xr0 + tid;
            g = g < 0 ? 0 : (g >= p.rows ? p.rows - 1 : g);
            MK[tid] = p.rowmask[g];
            UT[tid] = p.cond_rows ? g : g / p.rows_per_utt;
        }
        issue(0); issue(1); issue(2);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (xi[k] < 0) continue;
            const int i = xi[k] / per_row, c4 = xi[k] - i * per_row;
            const int ch = c4 * 4, kc = ch >> 5, cc = ch & 31;
            uint2 o; o.x = pack_bf16x2(xv[k].x, xv[k].y); o.y = pack_bf16x2(xv[k].z, xv[k].w);
            *reinterpret_cast<uint2*>(ST + kc * (WN_SROWS * 64) + swz(i, cc >> 3) + (cc & 7) * 2) = o;
        }
        const int npad4 = (96 - C2) >> 2;                      // zero the K padding [C2, 96) (the packed weights are zero there; LDS garbage may be NaN)
        for (int idx = tid; idx < WN_XR * npad4; idx += WN_NT) {
            const int i = idx / npad4, ch = C2 + (idx - i * npad4) * 4, kc = ch >> 5, cc = ch & 31;
            *reinterpret_cast<uint2*>(ST + kc * (WN_SROWS * 64) + swz(i, cc >> 3) + (cc & 7) * 2) = make_uint2(0u, 0u);
        }
    }

    // per-lane fragment offsets.  Row n of a slab (or tile) is 64 bytes, k-step s2 of lane half lhi reads slot q = 2 s2 + lhi; a block of 32
    // rows further on is +2048 bytes (the swizzle has period 16 rows), so two lane values + wave-uniform offsets address every fragment
    int bl[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) bl[s2] = swz(l31, 2 * s2 + lhi);
    const int offP = 2 * pi * 2048;                            // pair kind: fragments 2 pi (+0) and 2 pi + 1 (+2048) of a 384-row slab
    const int off1 = pi * 2048;                                // one-fragment kind: fragment pi of a 192-row half slab
    const int offA = rf * 2048;                                // this wave's row fragment of a 64-row tile
    // 16-byte copy of the valid rows of an LDS tile [6 chunks][trows][64 B] to a bf16 rows tensor [rows][192]
    auto copy_out = [&](const unsigned char* tile, int trows, int row_off, void* dst) __attribute__((always_inline)) {
        const Rsrc rd = mk_rsrc(dst, (long)p.rows * (WN_H * 2));
        int tid_ = tid;
        asm volatile("" : "+v"(tid_));                         // (opaque: keeps this address arithmetic out of the registers that live across the GEMM loops)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int idx = tid_ + k * WN_NT;
            const int r = idx / 24, pc = idx - r * 24;
            const bool ok = r < nvalid && v0 + r < p.rows;
            const Chunk16 v = lds16(tile + (pc >> 2) * (trows * 64) + swz(row_off + (ok ? r : 0), pc & 3));
            __builtin_amdgcn_raw_buffer_store_b128(v, rd, ok ? (uint32_t)((v0 + r) * (WN_H * 2) + pc * 16) : OOB, 0, 0);
        }
    };

    f32x16 acc0, acc1, skp;
    auto zero = [](f32x16& a) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = 0.f;
    };
    zero(skp);

    TLS();
    // ================= Start conv: x_0 = (W x_a + b) * mask on the 68 state rows (Modules.py:791) =================
    zero(acc0); zero(acc1);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const unsigned char* slot = begin_step(IC<0>{});
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (j == 1 && c == 1) break;                       // K = 96 = 3 chunks: slab 1 holds one chunk (the rest of it is never multiplied)
            const unsigned char* At = AT + (2 * j + c) * (WN_SROWS * 64);
            Chunk16 fa[2], fa2[2], fb[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                fb[s2] = lds16(slot + c * 12288 + off1 + bl[s2]);
                fa[s2] = lds16(At + offA + bl[s2]);
                if (wave < 6) fa2[s2] = lds16(At + bl[s2] + 64 * 64);       // third row fragment: rows 64..95 (rf = 0 here)
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                acc0 = mfma_bf16<!(ABL & 2)>(fa[s2], fb[s2], acc0);
                if (wave < 6) acc1 = mfma_bf16<!(ABL & 2)>(fa2[s2], fb[s2], acc1);
            }
        }
        end_step();
    }
    TLS();
    // Epilogue addressing.  A wave's accumulator element `reg` sits in row rb + frag_row(reg), rb = 32 rf + 4 lhi, column l31.  Its bf16 slot
    // in a swizzled [rows][64 B] tile is row * 64 + ((q ^ (row >> 2 & 3)) << 4) + (l31 & 7) * 2 with q = l31 >> 3; because rb is a multiple of 4
    // with (rb >> 2) & 3 == lhi, the XOR term takes one of four lane constants, selected by compile-time properties of reg: the odd / even
    // 8-row group and whether (reg & 3) + extra carries into the next group of four rows.  Address = base[og][cy] + compile-time offset.
    // `rb` is made opaque per use (empty asm) so that the 16 x per-register invariants are not hoisted out of the layer loop (they spilled).
    auto tile_bases = [&](int rb, int (&tb)[2][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int og = 0; og < 2; ++og)
#pragma unroll
            for (int cy = 0; cy < 2; ++cy) tb[og][cy] = rb * 64 + ((((l31 >> 3) ^ (lhi + 2 * og + cy)) & 3) << 4) + (l31 & 7) * 2;
    };
#define WN_TOFF(tb, reg, extra) ((tb)[((reg) >> 2) & 1][(((reg) & 3) + (extra)) >> 2] + (frag_row(reg) + (extra)) * 64)
    const int jch = pi * 32 + l31;                             // channel of this lane in 192-wide tensors
    const int lim = (p.rows - v0) < nvalid ? (p.rows - v0) : nvalid;      // owned rows that exist

    {   // x_0 -> state tile (rows 0..67)
        int rb = rf * 32 + 4 * lhi;
        asm volatile("" : "+v"(rb));
        int tb[2][2];
        tile_bases(rb, tb);
        const float b = BT[BT_START + jch];
        unsigned char* const xc = XT + pi * (WN_XR * 64);
        const float* const mk = MK + rb;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
            *reinterpret_cast<unsigned short*>(xc + WN_TOFF(tb, reg, 0)) = bf16_bits((acc0[reg] + b) * mk[frag_row(reg)]);
        if (wave < 6 && lhi == 0) {                            // rows 64..67 of the third fragment: registers 0..3 of the lower lane half
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                *reinterpret_cast<unsigned short*>(xc + 64 * 64 + WN_TOFF(tb, reg, 0)) = bf16_bits((acc1[reg] + b) * mk[64 + reg]);
        }
    }

    TLS();
    // dropout / conditioning constants
    const uint32_t thr = drop_threshold(p.drop_p);
    const float ik = drop_inv_keep(thr);
    uint32_t seed0 = p.seed;
    if (DROP && p.seed_ptr) seed0 += *p.seed_ptr;
    const Rsrc rcond = mk_rsrc(p.cond, COND ? (long)(p.cond_rows ? p.rows : p.rows / p.rows_per_utt) * p.ldcond * 4 : 0);
    const uint32_t jkey = drop_colkey((uint32_t)jch);
    // Per-utterance conditioning and utterances of at least a tile's 68 rows (>= 128 mel frames): the tile holds at most TWO utterances,
    // rows [0, cbnd) of the first and the rest of the second - four loads per layer, issued under the GEMM's last slab, and a per-row
    // select instead of 32 per-row loads whose round trips the epilogue waited for twice (+6.7 us per launch against the unconditioned kernel).
    const bool two = COND && !p.cond_rows && p.rows_per_utt >= WN_XR;
    int cu_lo = 0, cu_hi = 0, cbnd = WN_XR;
    if (two) {
        cu_lo = UT[0]; cu_hi = UT[WN_XR - 1];
        cbnd = (cu_lo + 1) * p.rows_per_utt - xr0;             // tile row of the second utterance's first row
    }

    // ================= WaveNet layers =================
#pragma unroll 1
    for (int l = 0; l < L; ++l) {
        const bool last = l == L - 1;
        // ---- In_l: k = 5 conv over the state tile, 30 slabs = (tap, K chunk) ----
        zero(acc0); zero(acc1);
        float cpre[4] = {0.f, 0.f, 0.f, 0.f};                  // conditioning of the tile's first / second utterance (this lane's tanh, sigmoid channel)
        {
            // Software pipeline over the slab steps: the fragments of slab j are read (LDS -> registers) during step j, its MFMAs run during
            // step j + 1 from the other register set.  A barrier-synchronous "read, wait, multiply" step would alternate between an LDS burst
            // (12 waves x 6 KiB right behind the barrier, matrix pipes idle) and an MFMA burst (LDS idle): measured 970 clocks per slab
            // against 384 of matrix work.  The ring protocol is unchanged: slab j's slot is read only inside step j.
            Chunk16 fa[2][2], fb[2][2][2];                     // [set][k step], [set][k step][fragment]
            auto mma = [&](auto SET_) __attribute__((always_inline)) {
                constexpr int st = decltype(SET_)::value;
                acc0 = mfma_bf16<!(ABL & 2)>(fa[st][0], fb[st][0][0], acc0);
                acc1 = mfma_bf16<!(ABL & 2)>(fa[st][0], fb[st][0][1], acc1);
                acc0 = mfma_bf16<!(ABL & 2)>(fa[st][1], fb[st][1][0], acc0);
                acc1 = mfma_bf16<!(ABL & 2)>(fa[st][1], fb[st][1][1], acc1);
            };
#pragma unroll 1
            for (int t = 0; t < WN_TAPS; ++t) {
                if constexpr ((ABL & 32) != 0) stamp_on = (l == 1 && t == 2);
                int ao[2];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) ao[s2] = swz(rf * 32 + l31 + t, 2 * s2 + lhi);
#pragma unroll
                for (int kc = 0; kc < WN_KCH; ++kc) {
                    constexpr int dummy = 0; (void)dummy;
                    // (the epilogue in front of this GEMM issues no global operation; the copy-out below sits behind slab 0's wait)
                    const unsigned char* slot = (t == 0 && keep && (kc == 1 || kc == 2)) ? begin_step(IC<XC>{}) : begin_step(IC<0>{});
                    if (kc == 0 && t == 0 && keep) copy_out(XT, WN_XR, halo + WN_PAD, pick4(p.hs, l));       // x_l (kept: X of the In_l weight gradient)
                    const int st = kc & 1;
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        fa[st][s2] = lds16(XT + kc * (WN_XR * 64) + ao[s2]);
                        fb[st][s2][0] = lds16(slot + offP + bl[s2]);
                        fb[st][s2][1] = lds16(slot + offP + 2048 + bl[s2]);
                    }
                    if constexpr ((ABL & 192) != 0) __builtin_amdgcn_sched_barrier(0);          // order experiments: reads first, pinned
                    if constexpr ((ABL & 128) != 0) { end_step(); __builtin_amdgcn_sched_barrier(0); }   // ... then the DMAs, then the MFMAs
                    if (kc > 0) { if (st) mma(IC<0>{}); else mma(IC<1>{}); }
                    else if (t > 0) mma(IC<1>{});
                    if constexpr ((ABL & 192) != 0) __builtin_amdgcn_sched_barrier(0);
                    if constexpr ((ABL & 128) == 0) end_step();
                }
            }
            if constexpr (COND) {
                if (two) {
                    const uint32_t cl = (uint32_t)(cu_lo * (int)p.ldcond + l * 2 * WN_H + jch) * 4u, ch = (uint32_t)(cu_hi * (int)p.ldcond + l * 2 * WN_H + jch) * 4u;
                    cpre[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rcond, cl, 0, 0));
                    cpre[1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rcond, cl + WN_H * 4u, 0, 0));
                    cpre[2] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rcond, ch, 0, 0));
                    cpre[3] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rcond, ch + WN_H * 4u, 0, 0));
                }
            }
            mma(IC<1>{});                                      // slab 29
        }
        TLS();
        // ---- gate epilogue: (conv + b) -> dropout -> + conditioning -> tanh, sigmoid (Modules.py:861-870, 885-887) ----
        {
            int rb = rf * 32 + 4 * lhi;
            asm volatile("" : "+v"(rb));
            int tb[2][2];
            tile_bases(rb, tb);
            const float b0 = BT[BT_IN + l * 384 + jch], b1 = BT[BT_IN + l * 384 + WN_H + jch];
            const Rsrc rg = mk_rsrc(pick4(p.gates, l), keep ? (long)p.rows * (2 * WN_H * 2) : 0);
            const uint32_t rk0 = (uint32_t)(t0 + rb) * 0x9E3779B1u + seed0 + (uint32_t)l;      // drop_rowkey(seed, row) = mix(row * M + seed)
            const uint32_t vg0 = (uint32_t)((t0 + rb) * (2 * WN_H * 2) + jch * 4);
            const uint32_t own0 = (uint32_t)(rb - halo);                                       // owned <=> (rb + c - halo) < lim (unsigned)
            unsigned char* const ac = AT + pi * (WN_WIN * 64);
            const int* const ut = UT + rb + WN_PAD;
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                float c0[8], c1[8];
                if constexpr (COND) {
                    if (two) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const bool lo = rb + WN_PAD + frag_row(hb * 8 + q) < cbnd;
                            c0[q] = lo ? cpre[0] : cpre[2]; c1[q] = lo ? cpre[1] : cpre[3];
                        }
                    } else {                                   // per-row conditioning / short utterances: the 16 loads of 8 rows in flight together
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const uint32_t co = (uint32_t)(ut[frag_row(hb * 8 + q)] * (int)p.ldcond + l * 2 * WN_H + jch) * 4u;
                            c0[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rcond, co, 0, 0));
                            c1[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rcond, co + WN_H * 4u, 0, 0));
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    constexpr uint32_t M1 = 0x9E3779B1u;
                    const int reg = hb * 8 + q, c = frag_row(reg);
                    float x0 = acc0[reg] + b0, x1 = acc1[reg] + b1;
                    if constexpr (DROP) {
                        uint32_t x = rk0 + (uint32_t)c * M1; x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13;
                        const uint32_t d = drop_draw(x, jkey);
                        x0 *= drop_keep_lo(d, thr, ik); x1 *= drop_keep_hi(d, thr, ik);
                    }
                    if constexpr (COND) { x0 += c0[q]; x1 += c1[q]; }
                    const float tg = tanh_<false>(x0), sg = sigmoid_<false>(x1);
                    const bool ok = own0 + (uint32_t)c < (uint32_t)lim;
                    __builtin_amdgcn_raw_buffer_store_b32(pack_bf16x2(tg, sg), rg, ok ? vg0 + (uint32_t)(c * (2 * WN_H * 2)) : OOB, 0, 0);
                    *reinterpret_cast<unsigned short*>(ac + WN_TOFF(tb, reg, 0)) = bf16_bits(tg * sg);
                }
            }
        }
        TLS();
        if (!last) {
            // ---- Res_Skip_l: 1x1 on tanh * sigmoid, PAIR-packed columns: fragment 0 = residual, fragment 1 = skip of channels [32 pi, 32 pi + 32) ----
            zero(acc0); zero(acc1);
            {
                Chunk16 fa[2][2], fb[2][2][2];                 // (pipelined like In_l)
                auto mma = [&](auto SET_) __attribute__((always_inline)) {
                    constexpr int st = decltype(SET_)::value;
                    acc0 = mfma_bf16<!(ABL & 2)>(fa[st][0], fb[st][0][0], acc0);
                    acc1 = mfma_bf16<!(ABL & 2)>(fa[st][0], fb[st][0][1], acc1);
                    acc0 = mfma_bf16<!(ABL & 2)>(fa[st][1], fb[st][1][0], acc0);
                    acc1 = mfma_bf16<!(ABL & 2)>(fa[st][1], fb[st][1][1], acc1);
                };
#pragma unroll
                for (int kc = 0; kc < WN_KCH; ++kc) {
                    // behind the gate epilogue's XG stores; the copy-out's XC stores sit behind slab 0's wait; from step 3 on all are older
                    const unsigned char* slot = kc == 0 ? begin_step(IC<XG>{}) : (kc <= 2 ? (keep ? begin_step(IC<XG + XC>{}) : begin_step(IC<XG>{})) : begin_step(IC<0>{}));
                    if (keep && kc == 0) copy_out(AT, WN_WIN, halo, pick4(p.acts, l));
                    const int st = kc & 1;
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        fa[st][s2] = lds16(AT + kc * (WN_WIN * 64) + offA + bl[s2]);
                        fb[st][s2][0] = lds16(slot + offP + bl[s2]);
                        fb[st][s2][1] = lds16(slot + offP + 2048 + bl[s2]);
                    }
                    if (kc > 0) { if (st) mma(IC<0>{}); else mma(IC<1>{}); }
                    end_step();
                }
                mma(IC<1>{});
            }
            TLS();
            // x_{l+1} = (x_l + res + b) * mask in place; skip += skip_l + b (Modules.py:871-879)
            int rb = rf * 32 + 4 * lhi;
            asm volatile("" : "+v"(rb));
            int tb[2][2];
            tile_bases(rb, tb);
            const float br = BT[BT_RS + l * 384 + jch], bs = BT[BT_RS + l * 384 + WN_H + jch];
            unsigned char* const xc = XT + pi * (WN_XR * 64);
            const float* const mk = MK + rb + WN_PAD;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                unsigned short* xp = reinterpret_cast<unsigned short*>(xc + WN_TOFF(tb, reg, WN_PAD));
                const float xin = __uint_as_float((uint32_t)*xp << 16);
                *xp = bf16_bits((xin + acc0[reg] + br) * mk[frag_row(reg)]);
                skp[reg] = skp[reg] + acc1[reg] + bs;
            }
        } else {
            // ---- last layer: Res_Skip has only skip outputs (192 columns, one fragment per wave, two K chunks per slab) ----
            zero(acc0);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const unsigned char* slot = j == 0 ? begin_step(IC<XG>{}) : (keep ? begin_step(IC<XG + XC>{}) : begin_step(IC<XG>{}));
                if (keep && j == 0) copy_out(AT, WN_WIN, halo, pick4(p.acts, l));
                Chunk16 fa[2][2], fb[2][2];
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        fa[c][s2] = lds16(AT + (2 * j + c) * (WN_WIN * 64) + offA + bl[s2]);
                        fb[c][s2] = lds16(slot + c * 12288 + off1 + bl[s2]);
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) acc0 = mfma_bf16<!(ABL & 2)>(fa[c][s2], fb[c][s2], acc0);
                __builtin_amdgcn_sched_barrier(0);
                end_step();
            }
            TLS();
            // output = (sum of skips + b) * mask (Modules.py:880-883): fp32 rows kept for the End conv's weight gradient, bf16 tile for the End conv
            int rb = rf * 32 + 4 * lhi;
            asm volatile("" : "+v"(rb));
            int tb[2][2];
            tile_bases(rb, tb);
            const float b = BT[BT_RSL + jch];
            const Rsrc rs = mk_rsrc(p.skip, keep ? (long)p.rows * (WN_H * 4) : 0);
            const uint32_t vs0 = (uint32_t)((t0 + rb) * (WN_H * 4) + jch * 4);
            const uint32_t own0 = (uint32_t)(rb - halo);
            unsigned char* const sc = XT + pi * (WN_WIN * 64);
            const float* const mk = MK + rb + WN_PAD;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int c = frag_row(reg);
                const float v = (skp[reg] + acc0[reg] + b) * mk[c];
                const bool ok = own0 + (uint32_t)c < (uint32_t)lim;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, ok ? vs0 + (uint32_t)(c * (WN_H * 4)) : OOB, 0, 0);
                *reinterpret_cast<unsigned short*>(sc + WN_TOFF(tb, reg, 0)) = bf16_bits(v);
            }
        }
    }

    TLS();
    // ================= End conv + affine coupling (Modules.py:793-806): PAIR-packed (m | logs), 6 waves x 2 fragments =================
    const int rfe = wave >= 3 ? 1 : 0, pe = wave - 3 * rfe;    // (waves 0..5 only)
    int laneE = lane;
    asm volatile("" : "+v"(laneE));                          // (opaque: nothing of this phase is computed early and kept live across the layers)
    const int l31e = laneE & 31, lhie = laneE >> 5;
    const int je = pe * 32 + l31e;
    const bool cok = wave < 6 && je < p.C2;
    const Rsrc rx = mk_rsrc(p.xsrc, (long)p.rows * p.ldx * 4), rz = mk_rsrc(p.xdst, (long)p.rows * p.ldxd * 4);
    int rbe = rfe * 32 + 4 * lhie;
    asm volatile("" : "+v"(rbe));
    const uint32_t owne = (uint32_t)(rbe - halo);
    float xb[16];
    {                                                          // (every wave issues the 16 loads - waves 6..11 out of range - so that the operation count below is uniform)
        const uint32_t vx0 = (uint32_t)((t0 + rbe) * (int)p.ldx + p.C2 + je) * 4u;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int c = frag_row(reg);
            const bool ok = cok && owne + (uint32_t)c < (uint32_t)lim;
            xb[reg] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, ok ? vx0 + (uint32_t)(c * (int)p.ldx * 4) : OOB, 0, 0));
        }
    }
    zero(acc0); zero(acc1);
#pragma unroll
    for (int j = 0; j < 3; ++j) {                              // the last three slabs: the ring drains
        // behind these slabs' DMAs: the XG skip stores of the last epilogue and the 16 x_b loads
        if (p.safe_waits) {
            if (j == 0)      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(2 * DPS) : "memory");
            else if (j == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(DPS) : "memory");
            else             asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
            if (j == 0)      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(2 * DPS + XG + 16) : "memory");
            else if (j == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(DPS + XG + 16) : "memory");
            else             asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(XG + 16) : "memory");
        }
        const unsigned char* slot = wn_smem + OFF_RING + (snext & (WN_NS - 1)) * WN_SLAB;
        ++snext;
        if (wave < 6) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const unsigned char* At = XT + (2 * j + c) * (WN_WIN * 64);
                Chunk16 fa[2], fb[2][2];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int q = 2 * s2 + lhie;
                    fa[s2] = lds16(At + swz(rfe * 32 + l31e, q));
                    fb[s2][0] = lds16(slot + c * 12288 + swz((2 * pe) * 32 + l31e, q));
                    fb[s2][1] = lds16(slot + c * 12288 + swz((2 * pe + 1) * 32 + l31e, q));
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    acc0 = mfma_bf16<!(ABL & 2)>(fa[s2], fb[s2][0], acc0);
                    acc1 = mfma_bf16<!(ABL & 2)>(fa[s2], fb[s2][1], acc1);
                }
            }
        }
    }
    TLS();
    if (wave < 6) {
        const float bm = cok ? BT[BT_END + je] : 0.f, bl = cok ? BT[BT_END + p.C2 + je] : 0.f;
        const Rsrc ro = mk_rsrc(p.outs, keep ? (long)p.rows * p.ldo * 4 : 0);
        const bool rev = p.reverse != 0;
        const uint32_t vz0 = (uint32_t)((t0 + rbe) * (int)p.ldxd + p.C2 + je) * 4u, vo0 = (uint32_t)((t0 + rbe) * (int)p.ldo + pe * 64 + l31e) * 4u;
        const float* const mk = MK + rbe + WN_PAD;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int c = frag_row(reg);
            const bool ok = cok && owne + (uint32_t)c < (uint32_t)lim;
            const float m = acc0[reg] + bm, lg = acc1[reg] + bl;
            const float z = rev ? (xb[reg] - m) * exp_<false>(-lg) * mk[c] : (m + exp_<false>(lg) * xb[reg]) * mk[c];
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(z), rz, ok ? vz0 + (uint32_t)(c * (int)p.ldxd * 4) : OOB, 0, 0);
            const uint32_t vo = ok ? vo0 + (uint32_t)(c * (int)p.ldo * 4) : OOB;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(m), ro, vo, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(lg), ro, vo + 128u, 0, 0);
        }
    }
    TLS();
#undef WN_TOFF
}

template <bool DROP, bool COND, int ABL = 0>
int launch_wn_fwd(const wn_fwd_args& k, dim3 grid, hipStream_t s)
{
#ifdef GLOWTTS_TOOLS
    if constexpr (ABL == 0 && !COND) {
        switch (GLOWTTS_TUNABLE("GLOWTTS_WN_ABL", 0)) {
            case 1: return launch_wn_fwd<DROP, COND, 1>(k, grid, s);
            case 2: return launch_wn_fwd<DROP, COND, 2>(k, grid, s);
            case 3: return launch_wn_fwd<DROP, COND, 3>(k, grid, s);
            case 4: return launch_wn_fwd<DROP, COND, 4>(k, grid, s);
            case 7: return launch_wn_fwd<DROP, COND, 7>(k, grid, s);
            case 16: return launch_wn_fwd<DROP, COND, 16>(k, grid, s);
            case 32: return launch_wn_fwd<DROP, COND, 32>(k, grid, s);
            case 64: return launch_wn_fwd<DROP, COND, 64>(k, grid, s);
            case 128: return launch_wn_fwd<DROP, COND, 128>(k, grid, s);
            case 96: return launch_wn_fwd<DROP, COND, 96>(k, grid, s);
            case 160: return launch_wn_fwd<DROP, COND, 160>(k, grid, s);
            case 256: return launch_wn_fwd<DROP, COND, 256>(k, grid, s);
            case 288: return launch_wn_fwd<DROP, COND, 288>(k, grid, s);
            default: break;
        }
    }
#endif
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wn_fwd_kernel<DROP, COND, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return GLOWTTS_E_LAUNCH;
        attr_done = true;
    }
    GLOWTTS_NOTE_STATIC("wn_fwd<%s%s>", DROP ? "drop" : "nodrop", COND ? ",cond" : "");
    hipLaunchKernelGGL((wn_fwd_kernel<DROP, COND, ABL>), grid, dim3(WN_NT), WN_LDS, s, k);
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

}  // namespace

int glowtts_wavenet_pack_bwd_images(const float* w_start, const float* w_in, const float* w_rs, const float* w_rs_last, const float* w_end,
                                    int F, int L, int C2, void* img_bwd, void* stream);      // wavenet_fused_bwd.hip
static int g_wn_safe_waits = 0;
extern "C" void glowtts_wavenet_debug_safe_waits(int on) { g_wn_safe_waits = on ? 1 : 0; }

extern "C" int glowtts_wavenet_image_bytes(int L, int transposed, int64_t* bytes_out)
{
    if (L < 1 || L > WN_MAXL || !bytes_out) return GLOWTTS_E_ARG;
    (void)transposed;
    *bytes_out = (int64_t)(36 * L + 2) * WN_SLAB;
    return GLOWTTS_OK;
}

extern "C" int glowtts_wavenet_pack_images(const float* w_start, const float* w_in, const float* w_rs, const float* w_rs_last, const float* w_end,
                                           int F, int L, int C2, void* img_fwd, void* img_bwd, void* stream)
{
    if (!w_start || !w_in || !w_rs_last || !w_end || (L > 1 && !w_rs) || F < 1 || L < 1 || L > WN_MAXL || C2 <= 64 || C2 > 96 || (C2 & 3)) return GLOWTTS_E_ARG;
    const int H = WN_H;
    const int64_t stride = (int64_t)(36 * L + 2) * WN_SLAB;
    if (img_fwd) {
        unsigned char* img = static_cast<unsigned char*>(img_fwd);
        // [Start: 2 slabs][layer l: In_l 30 slabs, Res_Skip_l 6 slabs (last layer: 3)][End: 3 slabs]
        int rc = glowtts_pack_weight_strided(w_start, F, 1, H, C2, 1, 0, GLOWTTS_PERM_NONE, 0, GLOWTTS_BF16, img, stride, 0, 0, stream);
        if (rc == GLOWTTS_OK) rc = glowtts_pack_weight_strided(w_in, F * L, L, 2 * H, H, WN_TAPS, 0, GLOWTTS_PERM_PAIR, H, GLOWTTS_BF16, img + 2 * WN_SLAB, stride, 36 * (int64_t)WN_SLAB, 0, stream);
        if (rc == GLOWTTS_OK && L > 1) rc = glowtts_pack_weight_strided(w_rs, F * (L - 1), L - 1, 2 * H, H, 1, 0, GLOWTTS_PERM_PAIR, H, GLOWTTS_BF16, img + 32 * WN_SLAB, stride, 36 * (int64_t)WN_SLAB, 0, stream);
        if (rc == GLOWTTS_OK) rc = glowtts_pack_weight_strided(w_rs_last, F, 1, H, H, 1, 0, GLOWTTS_PERM_NONE, 0, GLOWTTS_BF16, img + (int64_t)(36 * (L - 1) + 32) * WN_SLAB, stride, 0, 0, stream);
        if (rc == GLOWTTS_OK) rc = glowtts_pack_weight_strided(w_end, F, 1, 2 * C2, H, 1, 0, GLOWTTS_PERM_PAIR, C2, GLOWTTS_BF16, img + (int64_t)(36 * (L - 1) + 35) * WN_SLAB, stride, 0, 0, stream);
        if (rc != GLOWTTS_OK) return rc;
    }
    if (img_bwd) return glowtts_wavenet_pack_bwd_images(w_start, w_in, w_rs, w_rs_last, w_end, F, L, C2, img_bwd, stream);
    return GLOWTTS_OK;
}

extern "C" int glowtts_wavenet_fwd(const glowtts_flow_dims* d, const glowtts_flow_params* p, const glowtts_flow_acts* a,
                                   const float* xsrc, float* xdst, int reverse, int keep, void* stream)
{
    if (!d || !p || !a || !xsrc || !xdst || !p->wn_img || !a->rowmask) return GLOWTTS_E_ARG;
    const int C2 = d->C / 2;
    if (d->precision != GLOWTTS_BF16 || !d->act_bf16 || d->H != WN_H || d->ksize != WN_TAPS || d->L < 1 || d->L > WN_MAXL ||
        (d->C & 7) || C2 <= 64 || C2 > 96 || p->end.npad != 192 || p->start.kchunks != 3) return GLOWTTS_E_ARG;
    const int Tp = d->T + 2 * GLOWTTS_ROW_PAD;
    const int64_t R = (int64_t)d->B * Tp;
    if (R * 2 * WN_H * 2 >= ((int64_t)1 << 31) || R * d->C * 4 >= ((int64_t)1 << 31)) return GLOWTTS_E_ARG;
    if (p->cond && ((p->cond_rows ? R : (int64_t)d->B) * p->ldcond * 4 >= ((int64_t)1 << 31))) return GLOWTTS_E_ARG;   // 32-bit buffer offsets (per-row / per-utterance table)
    wn_fwd_args k;
    memset(&k, 0, sizeof(k));
    k.rows = (int)R; k.rows_per_utt = Tp; k.L = d->L; k.C2 = C2; k.reverse = reverse; k.keep = keep;
    k.xsrc = xsrc; k.ldx = d->C; k.xdst = xdst; k.ldxd = d->C; k.rowmask = a->rowmask;
    k.wimg = static_cast<const unsigned char*>(p->wn_img);
    k.b_start = p->b_start; k.b_end = p->b_end;
    for (int l = 0; l < d->L; ++l) { k.b_in[l] = p->b_in[l]; k.b_rs[l] = p->b_rs[l]; }
    k.cond = p->cond; k.ldcond = p->ldcond; k.cond_rows = p->cond_rows;
    k.drop_p = d->drop_p; k.seed = d->seed; k.seed_ptr = d->seed_ptr;
    k.safe_waits = g_wn_safe_waits;
#ifdef GLOWTTS_TOOLS
    if (GLOWTTS_TUNABLE("GLOWTTS_WN_ABL", 0) & 48) k.tl = reinterpret_cast<long long*>(a->skip_bf);      // tools/bench_wn.py passes the stamp buffer here
#endif
    if (keep) {
        if (!a->skip || !a->outs) return GLOWTTS_E_ARG;
        for (int l = 0; l < d->L; ++l) {
            if (!a->hs[l] || !a->gates[l] || !a->acts[l]) return GLOWTTS_E_ARG;
            k.hs[l] = a->hs[l]; k.gates[l] = a->gates[l]; k.acts[l] = a->acts[l];
        }
        k.skip = a->skip; k.outs = a->outs; k.ldo = p->end.npad;
    }
    const int nvalid = WN_WIN - 2 * WN_PAD * (d->L - 1);
    const dim3 grid((unsigned)((R + nvalid - 1) / nvalid));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool drop = d->drop_p > 0.f, cnd = p->cond != nullptr;
    if (drop && cnd) return launch_wn_fwd<true, true>(k, grid, s);
    if (drop) return launch_wn_fwd<true, false>(k, grid, s);
    if (cnd) return launch_wn_fwd<false, true>(k, grid, s);
    return launch_wn_fwd<false, false>(k, grid, s);
}

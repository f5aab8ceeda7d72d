// Channels-last implicit-GEMM convolution on MFMA for gfx950 (see include/glowtts_hip.h).
//
//   Y[r][n] = epilogue( sum_t sum_c A[r + t - pad][c] * W[n][c][t] )
//
// One kernel family serves every dense contraction of the Glow-TTS path: the WaveNet k=5 convs
// (Modules.py:861), the 1x1 Start / Res_Skip / End convs (:791,:871,:793), their data-gradients
// (same kernel, weights packed transposed + taps flipped) and the encoder convs.
//
// CDNA4 mapping
//   * rows (utterance, frame) are the MFMA M dimension, output channels the N dimension, input
//     channels x taps the K dimension.  Activations stay channels-last in HBM so both operands are
//     K-contiguous: no transposes anywhere on the forward / dgrad path.
//   * K is consumed in 64-byte chunks (32 bf16 or 16 f32 channels).  Per chunk the A tile
//     [BM + taps - 1 rows][64 B] is staged ONCE in LDS and reused by all taps: a tap is just a row
//     offset into the same LDS tile (the halo rows come with it), so the k=5 conv reads its input
//     once, not five times.  The weight tile [BN][64 B] for (tap, chunk) is a contiguous slab of the
//     pre-packed weight image.
//   * LDS tiles are [rows][4 x 16 B] with the 16-B slot XOR-swizzled by (row >> 2) & 3: the
//     ds_read_b128 fragment reads of v_mfma_f32_32x32x16_bf16 (lane = row, 16 B = 8 k-values) are
//     bank-conflict free at every tap offset.
//   * bf16 mode: fp32 activations are rounded to bf16 while being staged (v_cvt_pk_bf16_f32), fp32
//     accumulate.  f32 mode: the same tiles hold fp32 and feed v_mfma_f32_32x32x2_f32 (exact fp32).
//   * register-staged double buffering: global loads for step s+1 are issued before the MFMAs of step
//     s, written to the other LDS buffer after them; one barrier per step.
//   * epilogues are fused: bias, conditioning, tanh*sigmoid gate, residual/skip update, affine
//     coupling, gate derivative (see GLOWTTS_EPI_*).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/glowtts_hip.h"
#include "tunable.h"
#include "launch_log.h"
#include "device_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------------
// one packed element: i = ((t * kchunks + kc) * npad + n) * KC + kk  ->  the fp32 weight it holds (0 in the padding)
template <int KC>
__device__ __forceinline__ float pack_element(const float* __restrict__ w, long i, int O, int I, int taps, int transpose, int perm, int perm_h,
                                              int N, int K, int npad, int kchunks)
{
    const int kk = i % KC;
    long r = i / KC;
    const int n = r % npad; r /= npad;
    const int kc = r % kchunks;
    const int t = r / kchunks;
    const int k = kc * KC + kk;
    // map the packed (n, k) to original (o, c)
    int o_idx = transpose ? k : n;      // index that runs over O (possibly permuted)
    int c_idx = transpose ? n : k;
    const int o_lim = transpose ? K : N;   // padded logical extent of the O-side index
    bool ok = o_idx < o_lim;
    int o = o_idx;
    if (perm == GLOWTTS_PERM_PAIR) {
        const int p = o_idx >> 6, hsel = (o_idx >> 5) & 1, j = (p << 5) + (o_idx & 31);
        ok = ok && (j < perm_h);
        o = hsel * perm_h + j;
    }
    ok = ok && (o < O) && (c_idx < I);
    return ok ? w[((long)o * I + c_idx) * taps + (transpose ? (taps - 1 - t) : t)] : 0.f;
}

template <typename CT>
__global__ void pack_weight_kernel(const float* __restrict__ w, CT* __restrict__ out, int O, int I, int taps,
                                   int transpose, int perm, int perm_h, int N, int K, int npad, int kchunks,
                                   int inner, long outer_stride, long inner_stride /* bytes; inner = 0: images back to back */, long w_stride /* elements */)
{
    constexpr int KC = 64 / sizeof(CT);
    const long total = (long)taps * kchunks * npad * KC;
    w += (long)blockIdx.y * w_stride;              // batch of independent weights
    if (inner > 0) out = reinterpret_cast<CT*>(reinterpret_cast<unsigned char*>(out) + (blockIdx.y / inner) * outer_stride + (blockIdx.y % inner) * inner_stride);
    else out += (long)blockIdx.y * total;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        out[i] = (CT)pack_element<KC>(w, i, O, I, taps, transpose, perm, perm_h, N, K, npad, kchunks);
}

// many weights of different shapes in one launch (the encoder's ~30 convs, forward and transposed): a device job table, PACK_CHUNK
// packed elements per workgroup, the job found by bisection over the jobs' first workgroup
constexpr int PACK_CHUNK = 2048;
template <typename CT>
__global__ __launch_bounds__(256) void pack_weight_multi_kernel(const glowtts_pack_job* __restrict__ jobs, int njobs)
{
    constexpr int KC = 64 / sizeof(CT);
    int lo = 0, hi = njobs - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (jobs[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
    const glowtts_pack_job j = jobs[lo];
    const long total = (long)j.taps * j.kchunks * j.npad * KC;
    const long base = (long)(blockIdx.x - j.block0) * PACK_CHUNK;
    CT* out = static_cast<CT*>(j.packed);
#pragma unroll
    for (int e = 0; e < PACK_CHUNK / 256; ++e) {
        const long i = base + e * 256 + threadIdx.x;
        if (i < total) out[i] = (CT)pack_element<KC>(j.w, i, j.O, j.I, j.taps, j.transpose, j.perm, j.perm_h, j.N, j.K, j.npad, j.kchunks);
    }
}

inline int pad_to(int v, int m) { return (v + m - 1) / m * m; }

// ------------------------------------------------------------------------------------------------
// the GEMM kernel
// ------------------------------------------------------------------------------------------------
template <typename CT> struct Prec;
template <> struct Prec<float>  { static constexpr int KC = 16; static constexpr int E = 4; };
template <> struct Prec<__bf16> { static constexpr int KC = 32; static constexpr int E = 8; };

constexpr int MAX_TAPS = 5;

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>); the register-set selection below
// must be resolved by `if constexpr`, not by a run-time select, or the sets are demoted to scratch memory
template <int V> struct IC { static constexpr int value = V; };
template <int N> struct StaticFor {
    template <class F> __device__ __forceinline__ static void run(F&& f) { StaticFor<N - 1>::run(f); f(IC<N - 1>{}); }
};
template <> struct StaticFor<0> { template <class F> __device__ __forceinline__ static void run(F&&) {} };


// ------------------------------------------------------------------------------------------------
// the fused epilogues (shared by conv_cl_kernel and conv_dma_kernel).  acc[mi][ni] = 32x32 fragments of the wave's
// (MI*32) x (NI*32) tile whose first row / column is (m0 + wm*MI*32, n0 + wn*NI*32); BM = rows of the workgroup tile.
// ------------------------------------------------------------------------------------------------
// PITCH: the DGATE epilogue also accumulates the GR-mode Pitch_l weight gradient when asked to (GLOWTTS_F_COND_ROWS).  Only the register-staged
// kernel instantiates that variant: inside the 1024-thread LDS-DMA / chain kernels its extra live values cost the plain path 38 VGPRs and
// spills (measured +0.12 ms/step), so those kernels decline such calls on the host side.
template <typename CT, int MI, int NI, int EPI, bool PITCH = true>
__device__ __forceinline__ void conv_epilogue(const glowtts_conv_args& p, f32x16 (&acc)[MI][NI], const int m0, const int n0, const int BM,
                                              const int wm, const int wn, const int lane, const int tid, long long* tlbuf)
{
    constexpr bool EX = sizeof(CT) == 4;
    const int l31 = lane & 31, lhi = lane >> 5;
    // dropout seed: the device word (graph replays) is read here, not at kernel start, where the scalar load would sit in front
    // of the first tile loads
    uint32_t seed = p.seed;
    if (p.seed_ptr && p.drop_p > 0.f) seed += *p.seed_ptr;
    // accumulator element: row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), col = lane & 31.
    // Every global access goes through a buffer descriptor: rows >= p.rows (last tile) and invalid columns (voffset = OOB)
    // are dropped / read as 0 by the hardware bounds check, so the row loops carry no branches and a row's address is one
    // scalar offset on a per-thread voffset.  The loads of an 8-row block are issued together before its arithmetic (in / out
    // tensors may alias for all the compiler knows, so it cannot hoist them over the stores itself).  The epilogue is VALU-issue
    // bound (tools/conv_timeline.py): instruction count per element is what matters here.
    const bool in0_bf = (p.io_flags & GLOWTTS_IO_IN0_BF16) != 0, out0_bf = (p.io_flags & GLOWTTS_IO_OUT0_BF16) != 0;
    const int fl = p.flags;
    constexpr uint32_t OOB = 0x80000000u;
    const int rb = m0 + wm * MI * 32 + 4 * lhi;               // row of (mi = 0, reg = 0)
    auto roff = [](int mi, int reg) { return mi * 32 + (reg & 3) + 8 * (reg >> 2); };
    auto mk = [](const void* ptr, long bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, (int)bytes, 0x00020000); };
    auto ld32 = [](Rsrc r, uint32_t vo, uint32_t so) { return __builtin_amdgcn_raw_buffer_load_b32(r, vo, so, 0); };
    auto ldf = [&](Rsrc r, uint32_t vo, uint32_t so) { return __uint_as_float(ld32(r, vo, so)); };
    auto ldh = [](Rsrc r, uint32_t vo, uint32_t so) { return __uint_as_float((uint32_t)__builtin_amdgcn_raw_buffer_load_b16(r, vo, so, 0) << 16); };
    auto st32 = [](uint32_t v, Rsrc r, uint32_t vo, uint32_t so) { __builtin_amdgcn_raw_buffer_store_b32(v, r, vo, so, 0); };
    auto stf = [&](float v, Rsrc r, uint32_t vo, uint32_t so) { st32(__float_as_uint(v), r, vo, so); };
    auto sth = [](float v, Rsrc r, uint32_t vo, uint32_t so) {
        const __bf16 b = (__bf16)v;
        __builtin_amdgcn_raw_buffer_store_b16(*reinterpret_cast<const unsigned short*>(&b), r, vo, so, 0);
    };
    const Rsrc rmk = mk(p.rowmask, (long)p.rows * 4);

    if constexpr (EPI == GLOWTTS_EPI_LINEAR) {
        const uint32_t esz = out0_bf ? 2 : 4;
        const uint32_t eiz = in0_bf ? 2 : 4;
        const Rsrc ro = mk(p.out0, (long)p.rows * p.ld0 * esz), ri = mk(p.in0, (long)p.rows * p.ldi0 * eiz);
        const int ncv = (fl & GLOWTTS_F_COLMASK) ? p.ncols_valid[blockIdx.z] : 0x7FFFFFFF;
        uint32_t vo[NI], vi[NI], idn[NI]; float bs[NI]; bool cz[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n0 + (wn * NI + ni) * 32 + l31;
            const bool ok = n < p.n;
            vo[ni] = ok ? (uint32_t)(rb * (int)p.ld0 + n) * esz : OOB;
            vi[ni] = ok ? (uint32_t)(rb * (int)p.ldi0 + n) * eiz : OOB;
            bs[ni] = ((fl & GLOWTTS_F_BIAS) && ok) ? p.bias[n] : 0.f;
            cz[ni] = n >= ncv;
            idn[ni] = (uint32_t)rb * (uint32_t)p.n + (uint32_t)n;
        }
        const float ikl = 1.f / (1.f - p.drop_p);
        const float floor_ = (fl & GLOWTTS_F_RELU) ? 0.f : -__builtin_inff();
        const bool want_mask = (fl & GLOWTTS_F_MASK) && p.rowmask;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                // flag tests are hoisted out of the element loops: neutral elements make the arithmetic unconditional
                float xin[8][NI], xold[8][NI], mk8[8], v[8][NI];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    mk8[q] = 1.f;
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) { xin[q][ni] = 0.f; xold[q][ni] = 0.f; }
                }
                if (want_mask) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) mk8[q] = ldf(rmk, (uint32_t)rb * 4u, roff(mi, hb * 8 + q) * 4);
                }
                if (fl & (GLOWTTS_F_ADD_IN0 | GLOWTTS_F_GATE_IN0)) {
                    if (in0_bf) {
#pragma unroll
                        for (int q = 0; q < 8; ++q)
#pragma unroll
                            for (int ni = 0; ni < NI; ++ni) xin[q][ni] = ldh(ri, vi[ni], roff(mi, hb * 8 + q) * (int)p.ldi0 * 2);
                    } else {
#pragma unroll
                        for (int q = 0; q < 8; ++q)
#pragma unroll
                            for (int ni = 0; ni < NI; ++ni) xin[q][ni] = ldf(ri, vi[ni], roff(mi, hb * 8 + q) * (int)p.ldi0 * 4);
                    }
                }
                if (fl & GLOWTTS_F_ACCUM) {
#pragma unroll
                    for (int q = 0; q < 8; ++q)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) xold[q][ni] = ldf(ro, vo[ni], roff(mi, hb * 8 + q) * (int)p.ld0 * 4);
                }
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) v[q][ni] = fmaxf(acc[mi][ni][hb * 8 + q] + bs[ni], floor_);
                if (fl & GLOWTTS_F_DROPOUT) {
#pragma unroll
                    for (int q = 0; q < 8; ++q)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
                            v[q][ni] *= drop_scale(seed, idn[ni] + (uint32_t)roff(mi, hb * 8 + q) * (uint32_t)p.n, p.drop_p, ikl);
                }
                if (fl & GLOWTTS_F_GATE_IN0) {           // in0 = the kept output of the relu / dropout layer whose backward this is
#pragma unroll
                    for (int q = 0; q < 8; ++q)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) { v[q][ni] *= xin[q][ni] != 0.f ? ikl : 0.f; xin[q][ni] = 0.f; }
                }
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        const float t = (v[q][ni] + xin[q][ni]) * mk8[q];
                        v[q][ni] = (cz[ni] ? 0.f : t) + xold[q][ni];
                    }
                if (out0_bf) {
#pragma unroll
                    for (int q = 0; q < 8; ++q)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) sth(v[q][ni], ro, vo[ni], roff(mi, hb * 8 + q) * (int)p.ld0 * 2);
                } else {
#pragma unroll
                    for (int q = 0; q < 8; ++q)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) stf(v[q][ni], ro, vo[ni], roff(mi, hb * 8 + q) * (int)p.ld0 * 4);
                }
            }
        }
    } else if constexpr (EPI == GLOWTTS_EPI_RESSKIP) {
        // Modules.py:871-883.  Columns [0, h): x = (x + res) * mask -> out0; columns [h, 2h): output += skip -> out1;
        // last layer (n = h): output = (output + res_skip) * mask -> out1.
        const bool last = (fl & GLOWTTS_F_LAST) != 0, first = (fl & GLOWTTS_F_FIRST) != 0;
        const bool copy_bf = last && out0_bf && (const void*)p.out0 != (const void*)p.out1;     // last layer: out0 = optional bf16 copy of the sum
        const uint32_t e0 = out0_bf ? 2 : 4, ei = in0_bf ? 2 : 4;
        const Rsrc ro0 = mk(p.out0, (long)p.rows * p.ld0 * e0), rin = mk(p.in0, (long)p.rows * p.ldi0 * ei), ro1 = mk(p.out1, (long)p.rows * p.ld1 * 4);
        uint32_t v0[NI], vin[NI], v1[NI], vl0[NI]; float bs[NI]; bool anyres[NI], anyskip[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int nb = n0 + (wn * NI + ni) * 32, n = nb + l31;
            const bool ok = n < p.n, res = !last && n < p.h;
            bs[ni] = ok ? p.bias[n] : 0.f;
            v0[ni] = (ok && res) ? (uint32_t)(rb * (int)p.ld0 + n) * e0 : OOB;
            vin[ni] = (ok && res) ? (uint32_t)(rb * (int)p.ldi0 + n) * ei : OOB;
            v1[ni] = (ok && !res) ? (uint32_t)(rb * (int)p.ld1 + (last ? n : n - p.h)) * 4u : OOB;
            vl0[ni] = (ok && last) ? (uint32_t)(rb * (int)p.ld0 + n) * 2u : OOB;
            anyres[ni] = !last && nb < p.h;                   // wave-uniform: does this fragment hold residual / skip columns at all
            anyskip[ni] = last || nb + 31 >= p.h;
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                float mk8[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) mk8[q] = 1.f;
                if (p.rowmask) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) mk8[q] = ldf(rmk, (uint32_t)rb * 4u, roff(mi, hb * 8 + q) * 4);
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    // wave-uniform tests per fragment; the loads of the 8 rows are in flight together
                    float xin[8], xold[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) { xin[q] = 0.f; xold[q] = 0.f; }
                    if (anyres[ni]) {
                        if (in0_bf) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) xin[q] = ldh(rin, vin[ni], roff(mi, hb * 8 + q) * (int)p.ldi0 * 2);
                        } else {
#pragma unroll
                            for (int q = 0; q < 8; ++q) xin[q] = ldf(rin, vin[ni], roff(mi, hb * 8 + q) * (int)p.ldi0 * 4);
                        }
                    }
                    if (anyskip[ni] && !first) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) xold[q] = ldf(ro1, v1[ni], roff(mi, hb * 8 + q) * (int)p.ld1 * 4);
                    }
                    if (anyres[ni]) {
                        if (out0_bf) {
#pragma unroll
                            for (int q = 0; q < 8; ++q)
                                sth((xin[q] + acc[mi][ni][hb * 8 + q] + bs[ni]) * mk8[q], ro0, v0[ni], roff(mi, hb * 8 + q) * (int)p.ld0 * 2);
                        } else {
#pragma unroll
                            for (int q = 0; q < 8; ++q)
                                stf((xin[q] + acc[mi][ni][hb * 8 + q] + bs[ni]) * mk8[q], ro0, v0[ni], roff(mi, hb * 8 + q) * (int)p.ld0 * 4);
                        }
                    }
                    if (anyskip[ni]) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const float so = xold[q] + acc[mi][ni][hb * 8 + q] + bs[ni];
                            stf(last ? so * mk8[q] : so, ro1, v1[ni], roff(mi, hb * 8 + q) * (int)p.ld1 * 4);
                            // last layer + bf16 storage: out0 (otherwise unused) gets a bf16 copy of the final sum - the End conv's A operand
                            if (copy_bf) sth(so * mk8[q], ro0, vl0[ni], roff(mi, hb * 8 + q) * (int)p.ld0 * 2);
                        }
                    }
                }
            }
        }
    } else if constexpr (EPI == GLOWTTS_EPI_GATE) {
        // In_l + dropout + conditioning + tanh / sigmoid (Modules.py:861-870, 885-887); PAIR-packed columns: fragment 2*pi holds
        // the tanh half, 2*pi+1 the sigmoid half of 32 channels.  Output: interleaved (tanh, sigmoid) pairs.
        static_assert(NI % 2 == 0, "pair epilogues need NI even");
        const uint32_t esz = out0_bf ? 2 : 4;
        const Rsrc ro = mk(p.out0, (long)p.rows * p.ld0 * esz);
        // optional second output (bf16 storage only): acts = tanh * sigmoid as plain bf16 rows [rows][ld1], the A operand of the
        // Res_Skip conv and the X operand of its weight gradient (both then need no prologue).  No out1: zero-sized descriptor,
        // the stores are dropped by the bounds check.
        const Rsrc ra = mk(p.out1, p.out1 ? (long)p.rows * p.ld1 * 2 : 0);
        // conditioning rows: one per utterance, or (GLOWTTS_F_COND_ROWS: per-frame conditioning, GR-mode pitch) one per activation row
        const int Tp = (fl & GLOWTTS_F_COND_ROWS) ? 1 : (p.rows_per_utt > 0 ? p.rows_per_utt : 1);
        const int nutt = p.rows / Tp;
        const Rsrc rc = mk(p.cond, (long)nutt * p.ldcond * 4);
        const bool drop = p.drop_p > 0.f, cnd = p.cond != nullptr;
        const uint32_t thr = drop_threshold(p.drop_p);
        const float ik = drop_inv_keep(thr);
        const int u0 = m0 / Tp, rnext = (u0 + 1) * Tp;
        const bool two = m0 + BM <= rnext + Tp;               // the tile touches at most two utterances: their cond rows are kept in registers
        uint32_t vo[NI / 2], jkey[NI / 2], vc[NI / 2], va[NI / 2];
        float b0[NI / 2], b1[NI / 2];                                   // bias of the (tanh, sigmoid) channel
        float ca0[NI / 2], ca1[NI / 2], cb0[NI / 2], cb1[NI / 2];       // conditioning of utterance u0 / u0 + 1
        float sa0[NI / 2], sa1[NI / 2], sb0[NI / 2], sb1[NI / 2];       // bias + conditioning
#pragma unroll
        for (int pi = 0; pi < NI / 2; ++pi) {
            const int j = ((n0 + (wn * NI + 2 * pi) * 32) >> 6) * 32 + l31;
            const bool ok = j < p.h;
            vo[pi] = ok ? (uint32_t)(rb * (int)p.ld0 + 2 * j) * esz : OOB;
            vc[pi] = ok ? (uint32_t)j * 4u : OOB;
            va[pi] = ok ? (uint32_t)(rb * (int)p.ld1 + j) * 2u : OOB;
            jkey[pi] = drop_colkey((uint32_t)j);
            b0[pi] = ok ? p.bias[j] : 0.f;
            b1[pi] = ok ? p.bias[p.h + j] : 0.f;
            ca0[pi] = ca1[pi] = cb0[pi] = cb1[pi] = 0.f;
            if (cnd && two) {
                ca0[pi] = ldf(rc, vc[pi], u0 * (int)p.ldcond * 4);       ca1[pi] = ldf(rc, vc[pi] + (uint32_t)p.h * 4u, u0 * (int)p.ldcond * 4);
                cb0[pi] = ldf(rc, vc[pi], (u0 + 1) * (int)p.ldcond * 4); cb1[pi] = ldf(rc, vc[pi] + (uint32_t)p.h * 4u, (u0 + 1) * (int)p.ldcond * 4);
            }
            sa0[pi] = ca0[pi] + b0[pi]; sa1[pi] = ca1[pi] + b1[pi]; sb0[pi] = cb0[pi] + b0[pi]; sb1[pi] = cb1[pi] + b1[pi];
        }
        // three straight-line variants, chosen once: MODE 0 = no dropout, 1 = dropout (both with the <= 2-utterance conditioning
        // registers), 2 = general (conditioning row looked up per row; short utterances)
        auto rows_loop = [&](auto MODE_, auto OBF_) __attribute__((always_inline)) {
            constexpr int MODE = decltype(MODE_)::value;
            constexpr bool OBF = decltype(OBF_)::value != 0;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int ro_ = roff(mi, reg), r = rb + ro_;
                    const uint32_t rk = (MODE == 1 || (MODE == 2 && drop)) ? drop_rowkey(seed, (uint32_t)r) : 0u;
                    const bool sel = r >= rnext;
#pragma unroll
                    for (int pi = 0; pi < NI / 2; ++pi) {
                        float x0 = acc[mi][2 * pi][reg], x1 = acc[mi][2 * pi + 1][reg];
                        if constexpr (MODE == 2) {
                            x0 += b0[pi]; x1 += b1[pi];
                            if (drop) { const uint32_t d = drop_draw(rk, jkey[pi]); x0 *= drop_keep_lo(d, thr, ik); x1 *= drop_keep_hi(d, thr, ik); }
                            if (cnd) {
                                const uint32_t uo = (uint32_t)((r / Tp) * (int)p.ldcond) * 4u;     // per-lane row: goes into the voffset
                                x0 += ldf(rc, vc[pi] + uo, 0);
                                x1 += ldf(rc, vc[pi] + uo + (uint32_t)p.h * 4u, 0);
                            }
                        } else if constexpr (MODE == 1) {                          // dropout acts on conv + bias, the conditioning is added after it
                            const uint32_t d = drop_draw(rk, jkey[pi]);
                            x0 = (x0 + b0[pi]) * drop_keep_lo(d, thr, ik) + (sel ? cb0[pi] : ca0[pi]);
                            x1 = (x1 + b1[pi]) * drop_keep_hi(d, thr, ik) + (sel ? cb1[pi] : ca1[pi]);
                        } else {
                            x0 += sel ? sb0[pi] : sa0[pi];
                            x1 += sel ? sb1[pi] : sa1[pi];
                        }
                        float2 g = make_float2(tanh_<EX>(x0), sigmoid_<EX>(x1));
                        if constexpr (OBF) { st32(pack_bf16x2(g.x, g.y), ro, vo[pi], ro_ * (int)p.ld0 * 2); sth(g.x * g.y, ra, va[pi], ro_ * (int)p.ld1 * 2); }
                        else {
                            u32x2 w; w[0] = __float_as_uint(g.x); w[1] = __float_as_uint(g.y);
                            __builtin_amdgcn_raw_buffer_store_b64(w, ro, vo[pi], ro_ * (int)p.ld0 * 4, 0);
                        }
                    }
                }
            }
        };
        auto run_mode = [&](auto OBF_) __attribute__((always_inline)) {
            if (cnd && !two) rows_loop(IC<2>{}, OBF_);
            else if (drop)   rows_loop(IC<1>{}, OBF_);
            else             rows_loop(IC<0>{}, OBF_);
        };
        if constexpr (sizeof(CT) == 2) { if (out0_bf) run_mode(IC<1>{}); else run_mode(IC<0>{}); }      // (io_flags are bf16-precision only)
        else run_mode(IC<0>{});
    } else if constexpr (EPI == GLOWTTS_EPI_DGATE) {
        // d gate pre-activations from d acts and the kept gates (autograd of Modules.py:885-887 and of the Dropout at :862)
        const uint32_t esz = out0_bf ? 2 : 4, ei = in0_bf ? 2 : 4;
        const Rsrc ro = mk(p.out0, (long)p.rows * p.ld0 * esz), rg = mk(p.in0, (long)p.rows * p.ldi0 * ei);
        const bool drop = p.drop_p > 0.f;
        const uint32_t thr = drop_threshold(p.drop_p);
        const float ik = drop_inv_keep(thr);
        uint32_t vo[NI], vg[NI], jkey[NI];
        // out1 (optional): d conditioning [utterances][ld1], original channel order, ACCUMULATED (atomic adds; zero it first).  The
        // conditioning joins the pre-activation AFTER the dropout (Modules.py:861-866), so its gradient is the per-utterance sum of the
        // gate gradients BEFORE the keep mask is applied - which only exists here, in registers.
        float* dcnd = p.out1;
        const bool fx = (fl & GLOWTTS_F_COND_FX) != 0;             // out1 holds 64-bit fixed-point accumulators (device_common.h fx_atomic_add)
        const int Tp = p.rows_per_utt > 0 ? p.rows_per_utt : 1;
        int dcol[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int j = n0 + (wn * NI + ni) * 32 + l31;                          // gate channel (natural order)
            const bool ok = j < p.n;
            const int pc = (j >> 5) * 64 + (j & 31);
            vo[ni] = ok ? (uint32_t)(rb * (int)p.ld0 + pc) * esz : OOB;
            vg[ni] = ok ? (uint32_t)(rb * (int)p.ldi0 + 2 * j) * ei : OOB;
            jkey[ni] = drop_colkey((uint32_t)j);
            dcol[ni] = ok ? j : -1;
        }
        float sa[NI], ss[NI]; int cur_u = -1;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) sa[ni] = ss[ni] = 0.f;
        // GLOWTTS_F_COND_ROWS (GR-mode pitch): `cond` = the squeezed pitch rows [rows][ldcond <= 2]; rows nutt + j of out1 accumulate
        // sum_r (da, ds)[r][n] * pitch[r][j] - the Pitch_l conv's weight gradient, also taken before the keep mask
        const bool pit = PITCH && dcnd && p.cond && (fl & GLOWTTS_F_COND_ROWS);
        const int pns = pit ? (int)p.ldcond : 0;
        const Rsrc rp = mk(p.cond, pit ? (long)p.rows * pns * 4 : 0);
        float pa[2][NI], ps[2][NI];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) pa[j][ni] = ps[j][ni] = 0.f;
        // flush(need): called by ALL lanes of the wave (it shuffles); lanes with `need` add their run's sums to out1 and start a new run.
        // Lanes l and l + 32 hold the same column (rows 4 apart): when both flush the same utterance run - the rule, a run is hundreds of
        // rows - the upper half hands its sums to the lower one and only that issues the atomics (half the atomic traffic).
        auto flush = [&](const bool need) __attribute__((always_inline)) {
            const int other_u = __shfl_xor(need ? cur_u : -2, 32, 64);
            const bool pair = need && other_u == cur_u;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const float oa = __shfl_xor(sa[ni], 32, 64), os = __shfl_xor(ss[ni], 32, 64);
                if (pair) { sa[ni] += oa; ss[ni] += os; }
            }
            if (need && cur_u >= 0 && !(pair && lhi)) {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    if (dcol[ni] >= 0) {
                        if (fx) {
                            long long* dst = reinterpret_cast<long long*>(dcnd) + (long)cur_u * p.ld1 + dcol[ni];
                            fx_atomic_add(dst, sa[ni]); fx_atomic_add(dst + p.n, ss[ni]);
                        } else {
                            float* dst = dcnd + (long)cur_u * p.ld1 + dcol[ni];
                            unsafeAtomicAdd(dst, sa[ni]); unsafeAtomicAdd(dst + p.n, ss[ni]);
                        }
                    }
            }
            if (need) {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) sa[ni] = ss[ni] = 0.f;
            }
        };
        auto rows_loop = [&](auto DROP_, auto PIT_) __attribute__((always_inline)) {
            constexpr bool DROP = decltype(DROP_)::value != 0, PIT = decltype(PIT_)::value != 0;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {
                    float gt[8][NI], gs[8][NI], pv[8][2];
                    if constexpr (PIT) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) {         // (rows past the end read 0 through the descriptor's bounds check)
                            const uint32_t so = (uint32_t)(roff(mi, hb * 8 + q) * pns * 4);
                            pv[q][0] = ldf(rp, (uint32_t)(rb * pns) * 4u, so);
                            pv[q][1] = pns > 1 ? ldf(rp, (uint32_t)(rb * pns) * 4u + 4u, so) : 0.f;
                        }
                    }
                    if (in0_bf) {
#pragma unroll
                        for (int q = 0; q < 8; ++q)
#pragma unroll
                            for (int ni = 0; ni < NI; ++ni) {
                                const uint32_t w = ld32(rg, vg[ni], roff(mi, hb * 8 + q) * (int)p.ldi0 * 2);
                                gt[q][ni] = __uint_as_float(w << 16); gs[q][ni] = __uint_as_float(w & 0xFFFF0000u);
                            }
                    } else {
#pragma unroll
                        for (int q = 0; q < 8; ++q)
#pragma unroll
                            for (int ni = 0; ni < NI; ++ni) {
                                const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rg, vg[ni], roff(mi, hb * 8 + q) * (int)p.ldi0 * 4, 0);
                                gt[q][ni] = __uint_as_float(w[0]); gs[q][ni] = __uint_as_float(w[1]);
                            }
                    }
                    float da[8][NI], ds[8][NI];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int reg = hb * 8 + q;
                        const uint32_t rk = DROP ? drop_rowkey(seed, (uint32_t)(rb + roff(mi, reg))) : 0u;
                        if (dcnd) {                                        // rows ascend with (mi, reg): one run of rows per utterance and lane
                            const int r = rb + roff(mi, reg);
                            const int u = r < p.rows ? r / Tp : -1;        // (rows past the end hold clamped garbage: not accumulated)
                            const bool need = u != cur_u;
                            if (__builtin_amdgcn_ballot_w64(need)) { flush(need); if (need) cur_u = u; }      // (wave-uniform branch)
                        }
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) {
                            const float d = acc[mi][ni][reg], t = gt[q][ni], sg = gs[q][ni];
                            const float dsg = d * sg;
                            da[q][ni] = dsg * (1.f - t * t);
                            ds[q][ni] = dsg * t * (1.f - sg);
                            if (dcnd) { sa[ni] += da[q][ni]; ss[ni] += ds[q][ni]; }
                            if constexpr (PIT) {
#pragma unroll
                                for (int j = 0; j < 2; ++j) { pa[j][ni] += da[q][ni] * pv[q][j]; ps[j][ni] += ds[q][ni] * pv[q][j]; }
                            }
                            if constexpr (DROP) { const uint32_t w = drop_draw(rk, jkey[ni]); da[q][ni] *= drop_keep_lo(w, thr, ik); ds[q][ni] *= drop_keep_hi(w, thr, ik); }
                        }
                    }
                    if (out0_bf) {
#pragma unroll
                        for (int q = 0; q < 8; ++q)
#pragma unroll
                            for (int ni = 0; ni < NI; ++ni) {
                                const int so = roff(mi, hb * 8 + q) * (int)p.ld0 * 2;
                                sth(da[q][ni], ro, vo[ni], so); sth(ds[q][ni], ro, vo[ni] + 64u, so);
                            }
                    } else {
#pragma unroll
                        for (int q = 0; q < 8; ++q)
#pragma unroll
                            for (int ni = 0; ni < NI; ++ni) {
                                const int so = roff(mi, hb * 8 + q) * (int)p.ld0 * 4;
                                stf(da[q][ni], ro, vo[ni], so); stf(ds[q][ni], ro, vo[ni] + 128u, so);
                            }
                    }
                }
            }
        };
        if constexpr (PITCH) {
            if (pit) { if (drop) rows_loop(IC<1>{}, IC<1>{}); else rows_loop(IC<0>{}, IC<1>{}); }
            else     { if (drop) rows_loop(IC<1>{}, IC<0>{}); else rows_loop(IC<0>{}, IC<0>{}); }
        } else {
            if (drop) rows_loop(IC<1>{}, IC<0>{}); else rows_loop(IC<0>{}, IC<0>{});
        }
        if (dcnd) flush(true);
        if (pit) {
            const int nutt = p.rows / Tp;
            for (int j = 0; j < pns; ++j)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    if (dcol[ni] >= 0) {
                        if (fx) {
                            long long* dst = reinterpret_cast<long long*>(dcnd) + (long)(nutt + j) * p.ld1 + dcol[ni];
                            fx_atomic_add(dst, pa[j][ni]); fx_atomic_add(dst + p.n, ps[j][ni]);
                        } else {
                            float* dst = dcnd + (long)(nutt + j) * p.ld1 + dcol[ni];
                            unsafeAtomicAdd(dst, pa[j][ni]); unsafeAtomicAdd(dst + p.n, ps[j][ni]);
                        }
                    }
        }
    } else {
        // affine coupling on (m, logs) = End conv output (Modules.py:795-806); PAIR-packed like GATE: fragment 2*pi holds m, 2*pi+1 logs
        static_assert(EPI == GLOWTTS_EPI_COUPLE && NI % 2 == 0, "pair epilogues need NI even");
        const float* xsrc = p.in0 ? p.in0 : p.out0;           // x_b is read from the kept coupling input when given
        const long ldx = p.in0 ? p.ldi0 : p.ld0;
        const Rsrc rx = mk(xsrc, (long)p.rows * ldx * 4), ro = mk(p.out0, (long)p.rows * p.ld0 * 4), rk = mk(p.out1, p.out1 ? (long)p.rows * p.ld1 * 4 : 0);
        const bool rev = (fl & GLOWTTS_F_REVERSE) != 0;
        uint32_t vx[NI / 2], vo[NI / 2], vk[NI / 2]; float bm[NI / 2], bl[NI / 2];
#pragma unroll
        for (int pi = 0; pi < NI / 2; ++pi) {
            const int pcol = n0 + (wn * NI + 2 * pi) * 32, j = (pcol >> 6) * 32 + l31;      // packed column of m / channel inside a half
            const bool ok = j < p.h;
            vx[pi] = ok ? (uint32_t)(rb * (int)ldx + j) * 4u : OOB;
            vo[pi] = ok ? (uint32_t)(rb * (int)p.ld0 + j) * 4u : OOB;
            vk[pi] = ok ? (uint32_t)(rb * (int)p.ld1 + pcol + l31) * 4u : OOB;
            bm[pi] = ok ? p.bias[j] : 0.f;
            bl[pi] = ok ? p.bias[p.h + j] : 0.f;
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                float xb[8][NI / 2], mk8[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    mk8[q] = p.rowmask ? ldf(rmk, (uint32_t)rb * 4u, roff(mi, hb * 8 + q) * 4) : 1.f;
#pragma unroll
                    for (int pi = 0; pi < NI / 2; ++pi) xb[q][pi] = ldf(rx, vx[pi], roff(mi, hb * 8 + q) * (int)ldx * 4);
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int reg = hb * 8 + q, ro_ = roff(mi, reg);
#pragma unroll
                    for (int pi = 0; pi < NI / 2; ++pi) {
                        const float v0 = acc[mi][2 * pi][reg] + bm[pi];             // m
                        const float v1 = acc[mi][2 * pi + 1][reg] + bl[pi];         // logs
                        const float z = rev ? (xb[q][pi] - v0) * exp_<EX>(-v1) * mk8[q] : (v0 + exp_<EX>(v1) * xb[q][pi]) * mk8[q];
                        stf(z, ro, vo[pi], ro_ * (int)p.ld0 * 4);
                        if (p.out1) { stf(v0, rk, vk[pi], ro_ * (int)p.ld1 * 4); stf(v1, rk, vk[pi] + 128u, ro_ * (int)p.ld1 * 4); }
                    }
                }
            }
        }
    }
}

// Pipeline ("super-steps").  TAPS and APRO are compile-time, so one super-step is straight-line code:
//   * multi-tap conv (TAPS > 1): super-step = one 64-byte K chunk; the A tile [BM + TAPS - 1 rows] is staged once and shared
//     by the TAPS sub-steps (a tap is a row offset), the TAPS weight tiles of the chunk are staged together.
//   * 1x1 conv (TAPS == 1): super-step = NSUB consecutive K chunks, each with its own A and weight tile.
//   Per super-step and wave: NSUB * 2 * MI * NI MFMAs (40 for the k=5 WaveNet conv) between ONE pair of barriers; LDS is
//   single-buffered, the look-ahead lives in registers: every global load of super-step ss+1 is issued (unconditionally, with
//   clamped addresses, kept raw) before the MFMAs of super-step ss, and is masked / converted / written to LDS after them.
//   (History, measured on MI355X: per-tap steps with a barrier each ran ~1800 cycles per 256-cycle MFMA step; step-conditional
//   loads additionally degrade every s_waitcnt to vmcnt(0).)
// ABF: the A operand (and A2) is stored as bf16 in HBM (GLOWTTS_IO_A_BF16; bf16 precision only): staging is then a raw 16-byte
// copy per LDS slot - half the bytes, no conversion in the loop.
template <typename CT, int MI, int NI, int WM, int WN, int EPI, int TAPS, int APRO, bool ABF>
__global__ __launch_bounds__(WM * WN * 64) void conv_cl_kernel(const glowtts_conv_args pin)
{
    glowtts_conv_args p = pin;
    if (p.batch > 1) {                      // batched problems: shift every base pointer
        const long bz = blockIdx.z;
        p.a += bz * p.a_bstride;
        p.w = reinterpret_cast<const unsigned char*>(p.w) + bz * p.w_bstride;
        if (p.bias) p.bias += bz * p.bias_bstride;
        if (p.rowmask) p.rowmask += bz * p.mask_bstride;
        p.out0 += bz * p.out_bstride;
    }
    constexpr int BM = WM * MI * 32, BN = WN * NI * 32, NT = WM * WN * 64;
    constexpr bool T1 = (TAPS == 1);
    constexpr int NSUB = T1 ? ((APRO == GLOWTTS_APRO_NONE) ? 3 : 2) : TAPS;   // sub-steps per super-step
    constexpr int NAT = T1 ? NSUB : 1;                                        // A tiles per super-step
    constexpr int AROWS = BM + TAPS - 1;
    constexpr int KC = Prec<CT>::KC, E = Prec<CT>::E;          // channels per 64-B chunk / per 16-B slot
    constexpr int A_IT = (AROWS * 4 + NT - 1) / NT;
    constexpr int W_IT = (BN * 4) / NT;
    static_assert(!ABF || sizeof(CT) == 2, "bf16 activation storage needs bf16 precision");
    constexpr int AES = ABF ? 2 : 4;                                         // bytes per stored A element
    constexpr int AEL = (APRO == GLOWTTS_APRO_PAIRMUL) ? 2 * E : E;          // A elements loaded per 16-B LDS slot
    constexpr int NLD = AEL * AES / 16;                                      // 16-byte loads per LDS slot
    constexpr int A_TILE = AROWS * 64, W_TILE = BN * 64;
    static_assert((BN * 4) % NT == 0, "weight tile must divide evenly");

    __shared__ __attribute__((aligned(16))) unsigned char smem[NAT * A_TILE + NSUB * W_TILE];
    unsigned char* As = smem;                         // [NAT][AROWS][64]
    unsigned char* Ws = smem + NAT * A_TILE;          // [NSUB][BN][64]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // 1-D grid.  Workgroup b runs on XCD b % 8 (observed dispatch order): give every XCD a contiguous range of tiles with the
    // N tile fastest, so the workgroups that share an A row block (and its halo) hit the same L2.  Speed only, never correctness.
    int m_tile, n_tile;
    {
        const int gy = (p.npad + BN - 1) / BN;
        const int total = gridDim.x, lin = blockIdx.x;
        const int q = total >> 3, r = total & 7, xcd = lin & 7, j = lin >> 3;
        const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
        n_tile = t % gy; m_tile = t / gy;
    }
    const int m0 = m_tile * BM;
    const int n0 = n_tile * BN;
    const int KCH = p.kchunks;
    const int NSS = T1 ? (KCH + NSUB - 1) / NSUB : KCH;        // super-steps
    // All workgroups of a launch read the SAME weight tiles; started in lock-step they would all hit the same L2 channel at
    // the same moment (measured: ~2 us per super-step).  Each workgroup therefore walks the K chunks in a rotated order.
    const int rot = (int)((m_tile * 5 + n_tile * 3) % NSS);
    auto ssmap = [&](int ss) __attribute__((always_inline)) { int v = ss + rot; return v >= NSS ? v - NSS : v; };
    const int pad = (TAPS - 1) / 2;
    const int l31 = lane & 31, lhi = lane >> 5;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    typedef Chunk16 ARegs[A_IT][NLD];
    typedef Chunk16 WRegs[W_IT];
    ARegs ra[NAT];
    WRegs rw[NSUB];

    // ---- per-thread constants of the staging pattern ----
    const unsigned char* arow[A_IT];    // clamped source row of each A item (first source), as a byte pointer
    const unsigned char* arow2[A_IT];   // second source (dual-source A), same row
    bool aok[A_IT];                 // row inside [0, rows)
    int woff[W_IT];                 // clamped byte offset inside a weight tile slab
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int row = (tid + it * NT) >> 2;
        const long g = (long)m0 - pad + row;
        aok[it] = (g >= 0) && (g < p.rows) && (row < AROWS);
        const long gc = g < 0 ? 0 : (g >= p.rows ? p.rows - 1 : g);
        arow[it] = reinterpret_cast<const unsigned char*>(p.a) + gc * p.lda * AES;
        arow2[it] = p.a2 ? reinterpret_cast<const unsigned char*>(p.a2) + gc * p.lda2 * AES : arow[it];
    }
    {
        const int lim = (p.npad - n0) * 64 - 16;               // last valid 16-B piece of this tile's slab
#pragma unroll
        for (int it = 0; it < W_IT; ++it) woff[it] = min((tid + it * NT) * 16, lim);
    }

    // ---- global -> registers (raw, unconditional, clamped) ----
    auto gload_a = [&](ARegs& r, int kc) __attribute__((always_inline)) {
        kc = min(kc, KCH - 1);
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int c = kc * KC + ((tid + it * NT) & 3) * E;
            const unsigned char* src;                   // column offsets in A elements, clamped so that the AEL-element load stays inside the row
            if (APRO == GLOWTTS_APRO_PAIRMUL)      src = arow[it] + (long)min(2 * c, (int)p.lda - AEL) * AES;
            else if (APRO == GLOWTTS_APRO_SQNEG)   src = arow[it] + (long)min(c < p.ca1 ? c : c - p.ca1, (int)p.lda - AEL) * AES;
            else {
                const bool second = (p.a2 != nullptr) && (c >= p.ca1);
                src = second ? arow2[it] + (long)min(c - p.ca1, (int)p.lda2 - AEL) * AES : arow[it] + (long)min(c, (int)p.lda - AEL) * AES;
            }
#pragma unroll
            for (int j = 0; j < NLD; ++j) r[it][j] = *reinterpret_cast<const Chunk16*>(src + 16 * j);
        }
    };
    auto gload_w = [&](WRegs& r, int kc, int t) __attribute__((always_inline)) {
        kc = min(kc, KCH - 1);
        const unsigned char* base = reinterpret_cast<const unsigned char*>(p.w) + ((long)(t * KCH + kc) * p.npad + n0) * 64;
#pragma unroll
        for (int it = 0; it < W_IT; ++it) r[it] = *reinterpret_cast<const Chunk16*>(base + woff[it]);
    };
    // ---- registers -> LDS (mask, prologue, convert) ----
    auto sstore_a = [&](const ARegs& r, int tile, int kc) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int idx = tid + it * NT;
            const int row = idx >> 2, q = idx & 3;
            if (row >= AROWS) continue;
            const int c = kc * KC + q * E;
            const bool rowok = aok[it] && (kc < KCH);
            Chunk16 o;
            if constexpr (ABF && APRO == GLOWTTS_APRO_NONE) {
                // raw copy; a slot is valid or not as a whole (ca is a multiple of 8 on this path, checked on the host)
                const bool ok = rowok && (c < p.ca);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = ok ? r[it][0][e] : 0u;
            } else {
                float f[E];
                if constexpr (ABF) {                           // PAIRMUL on bf16 pairs: word = (tanh, sigmoid)
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const uint32_t w = r[it][e / 4][e % 4];
                        f[e] = __uint_as_float(w << 16) * __uint_as_float(w & 0xFFFF0000u);
                    }
                } else if (APRO == GLOWTTS_APRO_PAIRMUL) {
#pragma unroll
                    for (int j = 0; j < NLD; ++j) { f[2 * j] = __uint_as_float(r[it][j][0]) * __uint_as_float(r[it][j][1]); f[2 * j + 1] = __uint_as_float(r[it][j][2]) * __uint_as_float(r[it][j][3]); }
                } else {
#pragma unroll
                    for (int j = 0; j < NLD; ++j) { f[4 * j] = __uint_as_float(r[it][j][0]); f[4 * j + 1] = __uint_as_float(r[it][j][1]); f[4 * j + 2] = __uint_as_float(r[it][j][2]); f[4 * j + 3] = __uint_as_float(r[it][j][3]); }
                    if (APRO == GLOWTTS_APRO_SQNEG) {
                        if (c < p.ca1) {
#pragma unroll
                            for (int e = 0; e < E; ++e) f[e] = -0.5f * f[e] * f[e];
                        }
                    }
                }
#pragma unroll
                for (int e = 0; e < E; ++e) f[e] = (rowok && (c + e < p.ca)) ? f[e] : 0.f;
                if constexpr (sizeof(CT) == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(f[2 * e], f[2 * e + 1]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = __float_as_uint(f[e]);
                }
            }
            *reinterpret_cast<Chunk16*>(As + tile * A_TILE + swz(row, q)) = o;
        }
    };
    auto sstore_w = [&](const WRegs& r, int tile) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int idx = tid + it * NT;
            *reinterpret_cast<Chunk16*>(Ws + tile * W_TILE + swz(idx >> 2, idx & 3)) = r[it];
        }
    };

    // ---- MFMA over one sub-step: A tile `at` with row offset `roff`, weight tile `wt` ----
    auto compute = [&](int at, int roff, int wt) __attribute__((always_inline)) {
        const unsigned char* Ab = As + at * A_TILE;
        const unsigned char* Wb = Ws + wt * W_TILE;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int q = 2 * s2 + lhi;
            Chunk16 af[MI], bfr[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                af[mi] = *reinterpret_cast<const Chunk16*>(Ab + swz((wm * MI + mi) * 32 + l31 + roff, q));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                bfr[ni] = *reinterpret_cast<const Chunk16*>(Wb + swz((wn * NI + ni) * 32 + l31, q));
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    if constexpr (sizeof(CT) == 2) {
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            *reinterpret_cast<const bf16x8*>(&af[mi]), *reinterpret_cast<const bf16x8*>(&bfr[ni]), acc[mi][ni], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                __uint_as_float(af[mi][e]), __uint_as_float(bfr[ni][e]), acc[mi][ni], 0, 0, 0);
                    }
                }
        }
    };

    // loads / stores / MFMAs of one super-step (all static after unrolling)
    auto gload_ss = [&](int ss) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NSUB; ++j) {
            if constexpr (T1) { gload_w(rw[j], ss * NSUB + j, 0); gload_a(ra[j], ss * NSUB + j); }
            else              { gload_w(rw[j], ss, j); }
        }
        if constexpr (!T1) gload_a(ra[0], ss);
    };
    auto sstore_ss = [&](int ss) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NSUB; ++j) {
            sstore_w(rw[j], j);
            if constexpr (T1) sstore_a(ra[j], j, ss * NSUB + j);
        }
        if constexpr (!T1) sstore_a(ra[0], 0, ss);
    };
    gload_ss(ssmap(0));
    sstore_ss(ssmap(0));
    __syncthreads();
    for (int ss = 0; ss < NSS; ++ss) {
        const int cur = ssmap(ss), nxt = ssmap(ss + 1 < NSS ? ss + 1 : ss);
        gload_ss(nxt);                                        // next super-step, in flight during the MFMAs below
        {
#pragma unroll
            for (int j = 0; j < NSUB; ++j) {
                if constexpr (T1) { if (cur * NSUB + j < KCH) compute(j, 0, j); }
                else              compute(0, j, j);
            }
        }
        __syncthreads();                                      // every wave is done reading the tiles
        sstore_ss(nxt);
        __syncthreads();
    }

    // ---- fused epilogue ----
    conv_epilogue<CT, MI, NI, EPI>(p, acc, m0, n0, BM, wm, wn, lane, tid, nullptr);
}

// ------------------------------------------------------------------------------------------------
// conv_dma_kernel: the multi-tap bf16 convolution over bf16-STORED activations (GLOWTTS_IO_A_BF16, no A prologue), staged
// with LDS-DMA.  Why a second kernel (tools/conv_timeline.py on conv_cl_kernel, B = 32 WaveNet In conv): with 2-3 small
// workgroups per CU all in the same phase, the register-staged kernel spends a third of every super-step writing the SAME
// weight tiles into each workgroup's LDS (ds_write_b128 moves ~79 B/clk/CU) between two barriers, MFMA pipes idle.  Here:
//   * ONE fat workgroup per CU: WMR waves (4..16, chosen by the host so that all tiles fit the chip in one round), each wave
//     a 32-row x 64-column strip, so a weight tile is staged once per CU instead of once per 128 rows;
//   * staging is global_load_lds_dwordx4 (no VGPR round trip, no ds_write): 1 KiB per wave-instruction into a linear LDS
//     image; the XOR swizzle the fragment reads expect is applied to the per-lane SOURCE address (same involution);
//   * three LDS stages, one barrier per K chunk: chunks ss+1 and ss+2 stream in while the 2*TAPS*NI MFMAs of chunk ss run
//     (counted vmcnt waits, raw s_barrier: __syncthreads() would drain the DMA queue).
// Rows outside [0, rows) are clamped, not zeroed: they only feed outputs of rows outside the tensor (dropped by the epilogue's
// bounds check) and of the first / last utterance's outermost pad rows, which every consumer masks.
// ------------------------------------------------------------------------------------------------
extern __shared__ __attribute__((aligned(1024))) unsigned char dma_smem[];
constexpr int DMA1_CPS = 2;             // 1x1 convs: K chunks per pipeline stage

// KS (round 2): the K loop is bound by LDS fragment reads - a 32-row x 64-column strip per wave reads 1 A + 2 B fragments (3 KiB) for
// two MFMAs per 16-wide k-step, 1.5 KiB per MFMA against the 1 KiB per MFMA the LDS can feed at full matrix rate.  With KS the waves
// work in PAIRS: wave (g, kh) multiplies a 64-row x 64-column tile over only ONE half (kh) of every 32-wide K chunk: 2 A + 2 B fragments for
// four MFMAs = 1 KiB per MFMA, same MFMA count per wave, same LDS image, same staging.  After the loop the partners swap halves of their
// partial sums through LDS (each keeps the 32-row fragment kh of the pair's 64 rows), which leaves every wave with exactly the
// accumulators of the plain kernel: the epilogue is unchanged.
template <int EPI, int TAPS, int NI = 2, bool KS = false>
__global__ __launch_bounds__(1024) void conv_dma_kernel(const glowtts_conv_args pin, const int nst /* LDS stages: 2 or 3 */,
                                                        const int nload /* loader waves (0: every wave stages its share) */)
{
    typedef __bf16 CT;
    constexpr int BN = NI * 32, KC = 32;                      // NI = 2: 64-column strips; NI = 3: 96 (when that balances the SIMDs better)
    constexpr int WTB = BN * 64, WUT = BN / 16;               // bytes / DMA units of one weight tile
    // A stage holds SUB sub-steps.  Multi-tap: one K chunk = one A tile (with its TAPS - 1 halo rows) shared by the TAPS weight
    // tiles.  1x1 (TAPS == 1): DMA1_CPS consecutive K chunks, each with its own A tile and weight tile.
    constexpr bool T1 = (TAPS == 1);
    constexpr int SUB = T1 ? DMA1_CPS : TAPS;
    constexpr int NAT = T1 ? DMA1_CPS : 1;
    glowtts_conv_args p = pin;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);           // scalar: unit indices, LDS bases and branches below are wave-uniform
    const int WMR = (blockDim.x >> 6) - nload, BM = WMR * 32; // compute waves (the last `nload` waves only issue DMAs)
    const int AU = (BM + TAPS - 1 + 15) >> 4;                 // 16-row (1 KiB) DMA units of one A tile
    constexpr int WU = SUB * WUT;                             // 16-column units of the SUB weight tiles
    const int A_BYTES = NAT * AU * 1024, STAGE = A_BYTES + WU * 1024;
    int m_tile, n_tile;
    {   // XCD-aware tile order (see conv_cl_kernel)
        const int gy = p.npad / BN;
        const int total = gridDim.x, lin = blockIdx.x;
        const int q = total >> 3, r = total & 7, xcd = lin & 7, j = lin >> 3;
        const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
        n_tile = t % gy; m_tile = t / gy;
    }
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    const int KCH = p.kchunks;
    const int NSS = T1 ? KCH / DMA1_CPS : KCH;                // stages (host: KCH % DMA1_CPS == 0 on the 1x1 path)
    const int NSS1 = (T1 && p.a2) ? (p.ca1 / KC) / DMA1_CPS : NSS;      // stages that read the first A source (dual-source 1x1)
    const int rot = (T1 && p.a2) ? 0 : (int)((m_tile * 5 + n_tile * 3) % NSS);
    auto ssmap = [&](int ss) __attribute__((always_inline)) { int v = ss + rot; return v >= NSS ? v - NSS : v; };
    constexpr int pad = (TAPS - 1) / 2;
    const int l31 = lane & 31, lhi = lane >> 5;

    // lane -> (row or column inside a 16-unit, logical 16-byte slot): LDS position `lane` of a unit holds slot q of row lane >> 2
    const int lrow = lane >> 2, qa = (lane & 3) ^ ((lane >> 4) & 3);
    const int nau = NAT * AU, nun = nau + WU;
    const unsigned char* const abase = reinterpret_cast<const unsigned char*>(p.a);
    const unsigned char* const a2base = reinterpret_cast<const unsigned char*>(p.a2);
    const unsigned char* const wbase = reinterpret_cast<const unsigned char*>(p.w);
    const uint32_t wkstep = (uint32_t)p.npad * 64u * (T1 ? DMA1_CPS : 1);
    constexpr uint32_t akstep = KC * 2 * (T1 ? DMA1_CPS : 1);
    // stage-0 source offset of DMA unit u (32-bit byte offset from p.a (p.a2) / p.w, host-checked < 2^31)
    auto unit_off = [&](int u) __attribute__((always_inline)) -> uint32_t {
        if (u < nau) {
            const int ja = T1 ? u / AU : 0, ur = u - ja * AU;
            int g = m0 - pad + ur * 16 + lrow;
            g = g < 0 ? 0 : (g >= p.rows ? p.rows - 1 : g);
            return (uint32_t)g * (uint32_t)(p.lda * 2) + (uint32_t)(qa * 16 + ja * (KC * 2));
        }
        const int w = u - nau, t = w / WUT, cg = w - t * WUT; // t: tap (multi-tap) or chunk inside the stage (1x1)
        return (uint32_t)((T1 ? t : t * KCH) * p.npad + n0 + cg * 16 + lrow) * 64u + (uint32_t)(qa * 16);
    };
    auto dma = [&](int u, uint32_t off, int buf, int st) __attribute__((always_inline)) {
        const unsigned char* src;
        if (u >= nau)       src = wbase + (off + (uint32_t)st * wkstep);
        else if (st < NSS1) src = abase + (off + (uint32_t)st * akstep);
        else                src = a2base + (off + (uint32_t)(st - NSS1) * akstep);
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                         (void __attribute__((address_space(3)))*)(dma_smem + buf * STAGE + u * 1024), 16, 0, 0);
    };
    if (wave >= WMR) {
        // ---- loader wave (wave specialisation): an LDS-DMA instruction costs its issuing wave 100-200 clk in a phase that also reads
        // LDS and feeds the matrix pipe; two waves that do nothing else stage every tile (two LDS stages: stage ss+1 streams in while
        // the compute waves multiply stage ss).  They take part in every barrier and leave before the epilogue.
        const int lw = wave - WMR;
        for (int u = lw; u < nun; u += nload) dma(u, unit_off(u), 0, ssmap(0));
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        for (int ss = 0; ss + 1 < NSS; ++ss) {
            const int st = ssmap(ss + 1), buf = (ss + 1) & 1;
            for (int u = lw; u < nun; u += nload) dma(u, unit_off(u), buf, st);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        return;
    }

    constexpr int MIK = KS ? 2 : 1;
    f32x16 acc[MIK][NI];
#pragma unroll
    for (int mi = 0; mi < MIK; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    const int kgrp = wave >> 1, kh = wave & 1;                // KS: pair index and K half of this wave

    // DMA units of this wave when every wave stages its share: u = wave + i * WMR.  Their stage-0 source offsets are computed once; a
    // stage is a uniform byte step (SUB-or-1 x 64 B along an A row, SUB-or-1 [npad][64 B] slabs of the packed weights).
    constexpr int MAXU = NI == 2 ? 8 : 10;                    // >= ceil(units / WMR) for every WMR >= 4
    uint32_t uoff[MAXU];
#pragma unroll
    for (int i = 0; i < MAXU; ++i) uoff[i] = unit_off(wave + i * WMR);       // wave-uniform unit index (wave comes from readfirstlane)
    const int nmine = nload ? 0 : (nun - wave + WMR - 1) / WMR;              // DMA instructions this wave issues per stage
    auto issue_unit = [&](auto I_, int buf, int st) __attribute__((always_inline)) {
        constexpr int i = decltype(I_)::value;
        if constexpr (i < MAXU) {
            const int u = wave + i * WMR;
            if (u < nun) dma(u, uoff[i], buf, st);
        }
    };
    // MFMAs of stage `buf`; when `nbuf >= 0` the DMAs of a later stage are issued between the sub-steps, so that their issue cost
    // hides under the matrix pipe instead of serialising after the barrier.
    // Fragment reads run one sub-step ahead of the MFMAs in a second register set (a wave's next MFMA otherwise waits a full LDS
    // round trip: measured 3000 instead of 1920 clk per chunk at 3 waves per SIMD).
    Chunk16 fa[2][2], fb[2][2][NI];                           // [set][k half][fragment]
    auto compute = [&](int buf, int nbuf, int nst) __attribute__((always_inline)) {
        const unsigned char* Ab = dma_smem + buf * STAGE;
        const unsigned char* Wb = Ab + A_BYTES;
        auto load_frags = [&](auto T_) __attribute__((always_inline)) {
            constexpr int t = decltype(T_)::value, set = t & 1;
            const unsigned char* At = T1 ? Ab + t * (AU * 1024) : Ab;
            if constexpr (KS) {                               // fa[set][mi]: the pair's two row fragments, fb[set][0][ni]; K half kh only
                const int q = 2 * kh + lhi;
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) fa[set][mi] = *reinterpret_cast<const Chunk16*>(At + swz(kgrp * 64 + mi * 32 + l31 + (T1 ? 0 : t), q));
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) fb[set][0][ni] = *reinterpret_cast<const Chunk16*>(Wb + t * WTB + swz(ni * 32 + l31, q));
            } else {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int q = 2 * s2 + lhi;
                    fa[set][s2] = *reinterpret_cast<const Chunk16*>(At + swz(wave * 32 + l31 + (T1 ? 0 : t), q));
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) fb[set][s2][ni] = *reinterpret_cast<const Chunk16*>(Wb + t * WTB + swz(ni * 32 + l31, q));
                }
            }
        };
        load_frags(IC<0>{});
        StaticFor<SUB>::run([&](auto T_) __attribute__((always_inline)) {
            constexpr int t = decltype(T_)::value, set = t & 1;
            if constexpr (t + 1 < SUB) load_frags(IC<t + 1>{});
            if constexpr (KS) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&fa[set][mi]),
                                                                             *reinterpret_cast<const bf16x8*>(&fb[set][0][ni]), acc[mi][ni], 0, 0, 0);
            } else {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[0][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&fa[set][s2]),
                                                                            *reinterpret_cast<const bf16x8*>(&fb[set][s2][ni]), acc[0][ni], 0, 0, 0);
            }
            if (nbuf >= 0) {
                constexpr int PER = (MAXU + SUB - 1) / SUB;
                StaticFor<PER>::run([&](auto J_) __attribute__((always_inline)) { issue_unit(IC<t * PER + decltype(J_)::value>{}, nbuf, nst); });
            }
        });
    };

    // Three LDS stages, stages ss+1 and ss+2 in flight while stage ss is multiplied.  s_waitcnt takes an immediate, the number of
    // DMAs a wave has outstanding per stage (nmine) is uniform but only known at run time: dispatch once per wait.
    auto wait_keep = [&](int keep) __attribute__((always_inline)) {        // wait until at most `keep` of this wave's DMAs are outstanding, then barrier
        switch (keep) {
            case 0:  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
            case 1:  asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
            case 2:  asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
            case 3:  asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
            case 4:  asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
            case 5:  asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
            case 6:  asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
            case 7:  asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
        }
    };
    // nst LDS stages: stage ss+1 .. ss+nst-1 are in flight while stage ss is multiplied (nst = 3: two ahead; nst = 2: one ahead, two
    // thirds of the LDS, so that a second workgroup - of another kernel on another stream - can be co-resident on the CU)
    const int ahead = nst - 1;                                // (loader waves: the host passes nst = 2)
    if (!nload) StaticFor<MAXU>::run([&](auto I_) __attribute__((always_inline)) { issue_unit(I_, 0, ssmap(0)); });
    if (!nload && ahead > 1 && NSS > 1) StaticFor<MAXU>::run([&](auto I_) __attribute__((always_inline)) { issue_unit(I_, 1, ssmap(1)); });
    int cur = 0;                                              // LDS stage of pipeline stage ss (ss % nst)
    for (int ss = 0; ss < NSS; ++ss) {
        // this wave's DMAs of stage ss have landed (those of later stages may still fly); after the barrier so have everyone's,
        // and every wave is done reading the LDS stage of ss-1, which is refilled with stage ss+ahead during the MFMAs below
        wait_keep((ahead > 1 && ss + 1 < NSS) ? nmine : 0);
        const bool more = !nload && ss + ahead < NSS;
        const int nxt = cur == 0 ? nst - 1 : cur - 1;         // (ss + ahead) % nst
        compute(cur, more ? nxt : -1, more ? ssmap(ss + ahead) : 0);
        cur = cur == nst - 1 ? 0 : cur + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (KS) {
        // partners swap halves of their partial sums: wave (g, kh) hands over row fragment 1 - kh and keeps fragment kh, i.e. rows
        // [32 * wave, 32 * wave + 32) of the tile - the plain kernel's assignment.  The stage buffers are free now (every DMA has landed,
        // the barrier below orders the last fragment reads before the overwrite).
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        float* xmine = reinterpret_cast<float*>(dma_smem) + wave * (NI * 16 * 64) + lane;
        const float* xpart = reinterpret_cast<const float*>(dma_smem) + (wave ^ 1) * (NI * 16 * 64) + lane;
        if (kh) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) xmine[(ni * 16 + r) * 64] = acc[0][ni][r];
        } else {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) xmine[(ni * 16 + r) * 64] = acc[1][ni][r];
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kh) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][ni][r] = acc[1][ni][r] + xpart[(ni * 16 + r) * 64];
        } else {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][ni][r] += xpart[(ni * 16 + r) * 64];
        }
    }
    f32x16 (&accf)[1][NI] = reinterpret_cast<f32x16 (&)[1][NI]>(acc);         // row fragment of this wave: acc[0][*]
    conv_epilogue<CT, 1, NI, EPI, false>(p, accf, m0, n0, BM, wave, 0, lane, tid, nullptr);
}

// ------------------------------------------------------------------------------------------------
// conv_chain_kernel: TWO chained 1x1 convs over the same rows in one launch; the intermediate (N1 = 192 channels) never leaves the CU.
// The decoder's dependent kernel chain has row-local pairs whose second GEMM consumes all channels of the first one's output:
//   forward : last Res_Skip (acts -> final skip sum)  ->  End conv + affine coupling     (EPI1 = RESSKIP (LAST), EPI2 = COUPLE)
//   backward: End data gradient (douts -> d skip)     ->  last layer's gate derivative   (EPI1 = LINEAR + mask,  EPI2 = DGATE)
// Workgroup = 64 rows x all 192 intermediate channels = 2 x 3 waves (32 rows x 64 columns each), 202 workgroups at B = 32.  GEMM 1 is
// the LDS-DMA pipeline of conv_dma_kernel's 1x1 mode (bf16 A rows + weight slabs, two stages of two K chunks); its epilogue applies the
// first conv's tail, writes what later kernels need to global memory (bf16) and the same values as six swizzled K-chunk tiles into LDS;
// GEMM 2 multiplies those tiles with the second weight (streamed through the same stage buffers) and ends in the second epilogue.
// ------------------------------------------------------------------------------------------------
#ifndef GLOWTTS_CH_BM
#define GLOWTTS_CH_BM 64
#endif
constexpr int CH_WN = 3, CH_BN = CH_WN * 64, CH_BM = GLOWTTS_CH_BM, CH_WM = CH_BM / 32, CH_KC2 = CH_BN / 32;      // 192 intermediate channels = 6 K chunks of GEMM 2

template <int EPI1, int EPI2>
__global__ __launch_bounds__(CH_WN * CH_WM * 64) void conv_chain_kernel(const glowtts_conv_args pin1, const glowtts_conv_args pin2)
{
    typedef __bf16 CT;
    constexpr int NI = 2, KC = 32, CPS = DMA1_CPS;
    constexpr int AU = CH_BM / 16, WUT = CH_BN / 16;          // DMA units of one A chunk tile / one weight chunk tile
    constexpr int A_BYTES = CPS * AU * 1024, STAGE = A_BYTES + CPS * WUT * 1024;
    glowtts_conv_args p = pin1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / CH_WN, wn = wave - wm * CH_WN;
    const int m0 = blockIdx.x * CH_BM;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int lrow = lane >> 2, qa = (lane & 3) ^ ((lane >> 4) & 3);
    unsigned char* A2 = dma_smem + 2 * STAGE;                 // [6 chunks][64 rows][64 B], swizzled like a staged A tile
    long long* tlbuf = nullptr; (void)tlbuf;

    f32x16 acc[1][NI];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][ni][r] = 0.f;
    };
    // DMA of stage `st` of GEMM `which` (1: A chunks + weight slabs, 2: weight slabs only) into stage buffer `buf`
    auto issue = [&](int which, const glowtts_conv_args& q, int buf, int st) __attribute__((always_inline)) {
        const int nau = which == 1 ? CPS * AU : 0, nun = nau + CPS * WUT;
        for (int u = wave; u < nun; u += CH_WN * CH_WM) {
            const unsigned char* src;
            int dstu;
            if (u < nau) {
                const int ja = u / AU, ur = u - ja * AU;
                int g = m0 + ur * 16 + lrow;
                g = g >= q.rows ? q.rows - 1 : g;
                src = reinterpret_cast<const unsigned char*>(q.a) + (uint32_t)g * (uint32_t)(q.lda * 2) + (uint32_t)(qa * 16 + (st * CPS + ja) * (KC * 2));
                dstu = u;
            } else {
                const int w = u - nau, t = w / WUT, cg = w - t * WUT;
                src = reinterpret_cast<const unsigned char*>(q.w) + (uint32_t)((st * CPS + t) * q.npad + cg * 16 + lrow) * 64u + (uint32_t)(qa * 16);
                dstu = CPS * AU + w;
            }
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                             (void __attribute__((address_space(3)))*)(dma_smem + buf * STAGE + dstu * 1024), 16, 0, 0);
        }
    };
    // MFMAs of one stage: A tiles from `Abase` (stage buffer or the resident A2 tiles), weight tiles from the stage buffer
    auto compute = [&](const unsigned char* Abase, int atile_bytes, const unsigned char* Wb) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < CPS; ++t) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int q = 2 * s2 + lhi;
                const Chunk16 af = *reinterpret_cast<const Chunk16*>(Abase + t * atile_bytes + swz(wm * 32 + l31, q));
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const Chunk16 bf = *reinterpret_cast<const Chunk16*>(Wb + t * (CH_BN * 64) + swz((wn * NI + ni) * 32 + l31, q));
                    acc[0][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&af), *reinterpret_cast<const bf16x8*>(&bf),
                                                                        acc[0][ni], 0, 0, 0);
                }
            }
        }
    };
    auto wait_all = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // ---------------- GEMM 1 ----------------
    const int NSS1 = p.kchunks / CPS;
    zero_acc();
    issue(1, p, 0, 0);
    for (int ss = 0; ss < NSS1; ++ss) {
        wait_all();
        if (ss + 1 < NSS1) issue(1, p, (ss + 1) & 1, ss + 1);
        else               issue(2, pin2, (ss + 1) & 1, 0);   // first weight stage of GEMM 2 streams in under GEMM 1's last MFMAs + epilogue
        const unsigned char* sb = dma_smem + (ss & 1) * STAGE;
        compute(sb, AU * 1024, sb + A_BYTES);
    }
    // ---------------- epilogue 1: values -> global (what later kernels need) and -> LDS tiles (A operand of GEMM 2) ----------------
    {
        const int rb = m0 + wm * 32 + 4 * lhi;
        typedef __amdgpu_buffer_rsrc_t Rs;
        auto mkr = [](const void* ptr, long bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, (int)bytes, 0x00020000); };
        const Rs rmk = mkr(p.rowmask, (long)p.rows * 4);
        const Rs rold = mkr(p.out1, p.out1 ? (long)p.rows * p.ld1 * 4 : 0);          // RESSKIP: the fp32 skip sum of the earlier layers
        const Rs rout = mkr(p.out0, (long)p.rows * p.ld0 * 2);                        // bf16 result rows [rows][ld0]
        float mk[16];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
            mk[reg] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rmk, (uint32_t)rb * 4u, ((reg & 3) + 8 * (reg >> 2)) * 4, 0));
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = (wn * NI + ni) * 32 + l31;                                  // intermediate channel, < 192
            const float bias = (EPI1 == GLOWTTS_EPI_RESSKIP || (p.flags & GLOWTTS_F_BIAS)) ? p.bias[n] : 0.f;
            const uint32_t vo = (uint32_t)(rb * (int)p.ld0 + n) * 2u, vs = (uint32_t)(rb * (int)p.ld1 + n) * 4u;
            unsigned char* tile = A2 + (n >> 5) * (CH_BM * 64);
            const int q = (n & 31) >> 3, e2 = (n & 7) * 2;
            float old[16];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) old[reg] = 0.f;
            if (EPI1 == GLOWTTS_EPI_RESSKIP && !(p.flags & GLOWTTS_F_FIRST)) {       // the loads of the 16 rows are in flight together
#pragma unroll
                for (int reg = 0; reg < 16; ++reg)
                    old[reg] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rold, vs, ((reg & 3) + 8 * (reg >> 2)) * (int)p.ld1 * 4, 0));
            }
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int rr = (reg & 3) + 8 * (reg >> 2);                            // row offset inside the wave's 32 rows (+ 4 lhi in rb)
                const float v = (old[reg] + acc[0][ni][reg] + bias) * mk[reg];        // RESSKIP (last layer, Modules.py:880-883) / LINEAR + mask
                if constexpr (EPI1 == GLOWTTS_EPI_RESSKIP)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rold, vs, rr * (int)p.ld1 * 4, 0);    // the fp32 sum stays complete too
                const __bf16 hb = (__bf16)v;
                const unsigned short bits = *reinterpret_cast<const unsigned short*>(&hb);
                __builtin_amdgcn_raw_buffer_store_b16(bits, rout, vo, rr * (int)p.ld0 * 2, 0);
                *reinterpret_cast<unsigned short*>(tile + swz(wm * 32 + 4 * lhi + rr, q) + e2) = bits;
            }
        }
    }
    // ---------------- GEMM 2 ----------------
    const glowtts_conv_args& p2 = pin2;
    constexpr int NSS2 = CH_KC2 / CPS;
    zero_acc();
    for (int ss = 0; ss < NSS2; ++ss) {
        wait_all();                                            // weights of stage ss landed; (ss = 0) every wave's A2 tiles are written
        const int buf = (NSS1 + ss) & 1;
        if (ss + 1 < NSS2) issue(2, p2, buf ^ 1, ss + 1);
        compute(A2 + ss * CPS * (CH_BM * 64), CH_BM * 64, dma_smem + buf * STAGE + A_BYTES);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {
        glowtts_conv_args pe = pin2;
        conv_epilogue<CT, 1, NI, EPI2, false>(pe, acc, m0, 0, CH_BM, wm, wn, lane, tid, nullptr);
    }
}

// ------------------------------------------------------------------------------------------------
// conv_skinny_kernel: 1x1 convs with a SHORT K on fp32 rows (the coupling's Start conv 80 -> 192 and its data gradient 192 -> 80,
// Modules.py:791; the encoder's 192-channel projections).  At B = 32 these problems are 404 row fragments of 32 rows: too few for any tile
// to amortise staging, and the tiled kernel spends its 9-10 us in load -> LDS -> barrier -> MFMA -> epilogue phases that cannot overlap.
// Here ONE WAVE owns one 32-row fragment and all (<= 192) output columns: every operand load - the A rows straight from global memory
// as MFMA fragments (fp32 -> bf16 in registers), the whole packed weight image from L2 - is issued up front, no LDS, no barrier; waves
// of different fragments drift apart, so loads, MFMAs and stores of neighbours overlap.  Registers: KS16 * 8 (A) + KS16 * NI * 4 (B) +
// NI * 16 (accumulators) <= ~340 for K = 192, NI = 3: one wave per SIMD, which is all this problem size offers anyway.
// ------------------------------------------------------------------------------------------------
template <int KS16 /* 16-wide k steps = ca / 16 */, int NI /* 32-column fragments */, bool ABF = false /* A rows stored as bf16: the fragments are raw copies */>
__global__ __launch_bounds__(64) void conv_skinny_kernel(const glowtts_conv_args pin)
{
    typedef __bf16 CT;
    const glowtts_conv_args& p = pin;
    const int lane = threadIdx.x, l31 = lane & 31, lhi = lane >> 5;
    const int m0 = blockIdx.x * 32;
    int row = m0 + l31;
    row = row < p.rows ? row : p.rows - 1;                                   // clamped: rows past the end are dropped by the epilogue
    const float* arow = p.a + (long)row * p.lda + lhi * 8;
    f32x4 araw[KS16][2];
    Chunk16 abf[KS16];
#pragma unroll
    for (int s = 0; s < KS16; ++s) {
        if constexpr (ABF) {
            abf[s] = *reinterpret_cast<const Chunk16*>(reinterpret_cast<const unsigned char*>(p.a) + ((long)row * p.lda + s * 16 + lhi * 8) * 2);
        } else {
            araw[s][0] = *reinterpret_cast<const f32x4*>(arow + s * 16);
            araw[s][1] = *reinterpret_cast<const f32x4*>(arow + s * 16 + 4);
        }
    }
    Chunk16 bfr[KS16][NI];
    const unsigned char* wb = reinterpret_cast<const unsigned char*>(p.w);
#pragma unroll
    for (int s = 0; s < KS16; ++s)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)                                      // packed image [k chunk][n][64 B]: chunk s / 2, 16-byte slot 2 (s & 1) + lhi
            bfr[s][ni] = *reinterpret_cast<const Chunk16*>(wb + ((size_t)((s >> 1) * p.npad + ni * 32 + l31) * 64 + (size_t)((2 * (s & 1) + lhi) * 16)));
    f32x16 acc[1][NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][ni][r] = 0.f;
#pragma unroll
    for (int s = 0; s < KS16; ++s) {
        Chunk16 af;
        if constexpr (ABF) af = abf[s];
        else {
            af[0] = pack_bf16x2(araw[s][0][0], araw[s][0][1]); af[1] = pack_bf16x2(araw[s][0][2], araw[s][0][3]);
            af[2] = pack_bf16x2(araw[s][1][0], araw[s][1][1]); af[3] = pack_bf16x2(araw[s][1][2], araw[s][1][3]);
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
            acc[0][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&af), *reinterpret_cast<const bf16x8*>(&bfr[s][ni]), acc[0][ni], 0, 0, 0);
    }
    conv_epilogue<CT, 1, NI, GLOWTTS_EPI_LINEAR, false>(p, acc, m0, 0, 32, 0, 0, lane, lane, nullptr);
}

template <int KS16, int NI, bool ABF = false>
int launch_skinny(const glowtts_conv_args& a, hipStream_t s)
{
    GLOWTTS_NOTE_STATIC("conv_skinny<%d,%d%s>", KS16 * 16, NI * 32, ABF ? ",abf16" : "");
    hipLaunchKernelGGL((conv_skinny_kernel<KS16, NI, ABF>), dim3((a.rows + 31) / 32), dim3(64), 0, s, a);
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

// ------------------------------------------------------------------------------------------------
// proj_ln_kernel (round 4): the attention block's output projection AND the LayerNorm behind it in ONE launch (Modules.py:560-562):
//   proj = Dropout(att W^T + bias);  s = proj + x;  y = LayerNorm_192(s) * gamma + beta, * rowmask   (+ the bf16 copy of y, the (mean, rstd) pairs)
// conv_skinny_kernel's loads (A rows straight from global memory as MFMA fragments, the packed weight image from L2, everything issued up front).  Replaces a
// register-staged conv launch (the skinny kernel takes no dropout) + ln_fwd_kernel on the encoder's forward chain, six times a step.
// Arithmetic: the conv epilogue's (bias, dropout keyed by row * C + column) and ln_fwd_kernel's two-pass statistics; sums in another order.
// ------------------------------------------------------------------------------------------------
struct proj_ln_args {
    const float* a; long lda;                    // attention output rows, fp32 [rows][lda]
    const void* w; int npad;                     // packed bf16 image of the projection weight [192 -> 192]
    const float* bias; const float* x;           // bias [192]; residual rows [rows][192]
    const float* gamma; const float* beta; const float* rowmask;
    float* proj;                                 // out (may be NULL): the dropped projection, kept for the LayerNorm backward's dropout gate
    float* s; float* stats; float* y; unsigned short* yb;
    long rows; float eps, drop_p; uint32_t seed; const uint32_t* seed_ptr;
};

__global__ __launch_bounds__(384) void proj_ln_kernel(const proj_ln_args p)
{
    // One workgroup = one 32-row fragment, SIX waves = its six 32-column fragments: a single wave per fragment (conv_skinny_kernel's shape) ran 20.8 us
    // alone - 72 weight loads, 72 MFMAs and a 16-row epilogue in one dependent stream - against 13.0 us for the two launches it replaces.  Each wave loads
    // the fragment's A rows (the other five find them in L1), its own 12 weight chunks, does 12 MFMAs; row statistics meet in LDS (two exchanges: mean, then
    // the centred squares - ln_fwd_kernel's two passes).
    constexpr int KS16 = 12, NW = 6, C = 192;
    __shared__ float part[2][NW][32];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * 32;
    // the fragment's A rows: once, coalesced, into LDS (as per-lane fragment loads - a lane per row - each of the six waves walked 32 cache lines per
    // instruction: 17 us for the launch); rows of 196 floats: a fragment read is conflict-free
    constexpr int LDA = C + 4;
    __shared__ __attribute__((aligned(16))) float atile[32 * LDA];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 384 * i, r = idx / 48, c4 = idx - r * 48;
        long gr = m0 + r;
        gr = gr < p.rows ? gr : p.rows - 1;
        *reinterpret_cast<f32x4*>(&atile[r * LDA + c4 * 4]) = *reinterpret_cast<const f32x4*>(p.a + gr * p.lda + c4 * 4);
    }
    Chunk16 bfr[KS16];
    const unsigned char* wb = reinterpret_cast<const unsigned char*>(p.w);
#pragma unroll
    for (int k = 0; k < KS16; ++k)
        bfr[k] = *reinterpret_cast<const Chunk16*>(wb + ((size_t)((k >> 1) * p.npad + wave * 32 + l31) * 64 + (size_t)((2 * (k & 1) + lhi) * 16)));
    // epilogue operands, asked for before the MFMAs: accumulator element reg: row = (reg & 3) + 8 (reg >> 2) + 4 lhi, column = 32 wave + l31; rows past the
    // end are dropped / read as zero by the descriptors' bounds check
    auto mk = [](const void* ptr, long bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, (int)bytes, 0x00020000); };
    const Rsrc rx = mk(p.x, p.rows * C * 4), rs = mk(p.s, p.rows * C * 4), ry = mk(p.y, p.rows * C * 4), ryb = mk(p.yb, p.rows * C * 2);
    const Rsrc rpj = mk(p.proj, p.proj ? p.rows * C * 4 : 0), rst = mk(p.stats, p.rows * 8), rmk = mk(p.rowmask, p.rows * 4);
    const int rb = m0 + 4 * lhi, col = wave * 32 + l31;
    auto rof = [](int reg) { return (reg & 3) + 8 * (reg >> 2); };
    float xin[16], mk16[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int r = rb + rof(reg);
        xin[reg] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, (uint32_t)(r * C + col) * 4u, 0, 0));
        mk16[reg] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rmk, (uint32_t)r * 4u, 0, 0));
    }
    const float bs = p.bias[col], gm = p.gamma[col], bt = p.beta[col];
    uint32_t seed = p.seed;
    if (p.seed_ptr && p.drop_p > 0.f) seed += *p.seed_ptr;
    const bool drop = p.drop_p > 0.f;
    const float ik = drop ? 1.f / (1.f - p.drop_p) : 1.f;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    __syncthreads();
    f32x4 araw[KS16][2];
#pragma unroll
    for (int k = 0; k < KS16; ++k) {
        araw[k][0] = *reinterpret_cast<const f32x4*>(&atile[l31 * LDA + lhi * 8 + k * 16]);
        araw[k][1] = *reinterpret_cast<const f32x4*>(&atile[l31 * LDA + lhi * 8 + k * 16 + 4]);
    }
#pragma unroll
    for (int k = 0; k < KS16; ++k) {
        Chunk16 af;
        af[0] = pack_bf16x2(araw[k][0][0], araw[k][0][1]); af[1] = pack_bf16x2(araw[k][0][2], araw[k][0][3]);
        af[2] = pack_bf16x2(araw[k][1][0], araw[k][1][1]); af[3] = pack_bf16x2(araw[k][1][2], araw[k][1][3]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&af), *reinterpret_cast<const bf16x8*>(&bfr[k]), acc, 0, 0, 0);
    }
    // sum over the 32 lanes of a half wave, valid in its LAST lane (31 / 63): four row_shr steps inside each 16-lane row, then row_bcast:15 adds row 0's
    // total into row 1 (row 2's into row 3) - five VALU instructions with DPP operands (as ds_bpermute shuffles the 32 sums of a lane were 960 LDS
    // instructions per workgroup)
    auto half_sum32 = [](float v) __attribute__((always_inline)) -> float {
        auto dpp = [](float x, auto CTRL, auto RMASK) __attribute__((always_inline)) -> float {
            return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(CTRL)::value, decltype(RMASK)::value, 0xF, true));
        };
        v += dpp(v, IC<0x111>{}, IC<0xF>{});
        v += dpp(v, IC<0x112>{}, IC<0xF>{});
        v += dpp(v, IC<0x114>{}, IC<0xF>{});
        v += dpp(v, IC<0x118>{}, IC<0xF>{});
        v += dpp(v, IC<0x142>{}, IC<0xA>{});
        return v;
    };
    float v[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const uint32_t id = (uint32_t)(rb + rof(reg)) * (uint32_t)C + (uint32_t)col;
        float t = acc[reg] + bs;
        if (drop) {
            t *= drop_scale(seed, id, p.drop_p, ik);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(t), rpj, id * 4u, 0, 0);
        }
        t += xin[reg];
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(t), rs, id * 4u, 0, 0);
        v[reg] = t;
        const float ps = half_sum32(t);
        if (l31 == 31) part[0][wave][4 * lhi + rof(reg)] = ps;
    }
    __syncthreads();
    float mean[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int rr = 4 * lhi + rof(reg);
        float sm = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) sm += part[0][w][rr];
        mean[reg] = sm * (1.f / C);
        const float d = v[reg] - mean[reg];
        const float pq = half_sum32(d * d);
        if (l31 == 31) part[1][wave][rr] = pq;
    }
    __syncthreads();
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int rr = 4 * lhi + rof(reg), r = rb + rof(reg);
        float sq = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) sq += part[1][w][rr];
        const float rstd = rsqrtf(sq * (1.f / C) + p.eps);
        if (wave == 0 && l31 < 2) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(l31 ? rstd : mean[reg]), rst, (uint32_t)(2 * r + l31) * 4u, 0, 0);
        const uint32_t id = (uint32_t)r * (uint32_t)C + (uint32_t)col;
        const float o = ((v[reg] - mean[reg]) * rstd * gm + bt) * mk16[reg];
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o), ry, id * 4u, 0, 0);
        const __bf16 ob = (__bf16)o;
        __builtin_amdgcn_raw_buffer_store_b16(*reinterpret_cast<const unsigned short*>(&ob), ryb, id * 2u, 0, 0);
    }
}

// ------------------------------------------------------------------------------------------------
// ln_qkv_kernel (round 4): the LayerNorm that closes a transformer block AND the fused Q / K / V 1x1 conv that opens the next one, in ONE launch
// (Modules.py:571 -> RPR_MHA.py:82-84):  s = a + b;  y = (LayerNorm_192(s) gamma + beta) rowmask (+ its bf16 copy, the (mean, rstd) pairs);
// qkv = y_bf16 Wqkv^T + bias.  proj_ln_kernel's shape: one workgroup per 32-row fragment, six waves = the six 32-column fragments of the LayerNorm
// (row statistics meet in LDS), then the same six waves = three 32-column fragments each of the 576 output columns, A fragments from the bf16 tile
// the LayerNorm left in LDS, weight chunks asked for at the top of the kernel.
// ------------------------------------------------------------------------------------------------
struct ln_qkv_args {
    const float* a; const float* b;              // LayerNorm input and residual rows, fp32 [rows][192]
    const float* gamma; const float* beta; const float* rowmask;
    float* s; float* stats; float* y; unsigned short* yb;
    const void* w; int npad; const float* bias;  // packed bf16 image of Wqkv [192 -> 576], bias [576]
    float* qkv;                                  // out: fp32 [rows][576]
    long rows; float eps;
};

__global__ __launch_bounds__(384) void ln_qkv_kernel(const ln_qkv_args p)
{
    constexpr int KS16 = 12, NW = 6, C = 192, NF = 3, N = 576, LDT = C + 8;
    __shared__ float part[2][NW][32];
    __shared__ __attribute__((aligned(16))) unsigned short ytile[32 * LDT];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * 32;
    auto mk = [](const void* ptr, long bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, (int)bytes, 0x00020000); };
    const Rsrc ra = mk(p.a, p.rows * C * 4), rbb = mk(p.b, p.rows * C * 4), rs = mk(p.s, p.rows * C * 4), ry = mk(p.y, p.rows * C * 4);
    const Rsrc ryb = mk(p.yb, p.rows * C * 2), rst = mk(p.stats, p.rows * 8), rmk = mk(p.rowmask, p.rows * 4), rq = mk(p.qkv, p.rows * N * 4);
    const int rb = m0 + 4 * lhi, col = wave * 32 + l31;
    auto rof = [](int reg) { return (reg & 3) + 8 * (reg >> 2); };
    // the loads up front: the LayerNorm's operands of this lane (element reg: row = rof(reg) + 4 lhi, column = 32 wave + l31) and half of the
    // 36 weight chunks of this wave's three output fragments
    float v[16], mk16[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int r = rb + rof(reg);
        const uint32_t off = (uint32_t)(r * C + col) * 4u;
        v[reg] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ra, off, 0, 0)) + __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rbb, off, 0, 0));
        mk16[reg] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rmk, (uint32_t)r * 4u, 0, 0));
    }
    const float gm = p.gamma[col], bt = p.beta[col];
    Chunk16 bfr[KS16][NF];
    const unsigned char* wb = reinterpret_cast<const unsigned char*>(p.w);
    auto load_w = [&](int k0, int k1) __attribute__((always_inline)) {
#pragma unroll
        for (int k = k0; k < k1; ++k)
#pragma unroll
            for (int f = 0; f < NF; ++f)
                bfr[k][f] = *reinterpret_cast<const Chunk16*>(wb + ((size_t)((k >> 1) * p.npad + (wave * NF + f) * 32 + l31) * 64 + (size_t)((2 * (k & 1) + lhi) * 16)));
    };
    load_w(0, KS16 / 2);                                        // (the second half behind the LayerNorm: all 36 chunks live across it spilled)
    float bq[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) bq[f] = p.bias[(wave * NF + f) * 32 + l31];
    auto half_sum32 = [](float x) __attribute__((always_inline)) -> float {      // valid in lanes 31 / 63 (see proj_ln_kernel)
        auto dpp = [](float t, auto CTRL, auto RMASK) __attribute__((always_inline)) -> float {
            return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), decltype(CTRL)::value, decltype(RMASK)::value, 0xF, true));
        };
        x += dpp(x, IC<0x111>{}, IC<0xF>{});
        x += dpp(x, IC<0x112>{}, IC<0xF>{});
        x += dpp(x, IC<0x114>{}, IC<0xF>{});
        x += dpp(x, IC<0x118>{}, IC<0xF>{});
        x += dpp(x, IC<0x142>{}, IC<0xA>{});
        return x;
    };
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[reg]), rs, (uint32_t)((rb + rof(reg)) * C + col) * 4u, 0, 0);
        const float ps = half_sum32(v[reg]);
        if (l31 == 31) part[0][wave][4 * lhi + rof(reg)] = ps;
    }
    __syncthreads();
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int rr = 4 * lhi + rof(reg);
        float sm = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) sm += part[0][w][rr];
        v[reg] -= sm * (1.f / C);                               // centred from here on (the mean itself is re-summed for the statistics store below)
        const float pq = half_sum32(v[reg] * v[reg]);
        if (l31 == 31) part[1][wave][rr] = pq;
    }
    __syncthreads();
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int rr = 4 * lhi + rof(reg), r = rb + rof(reg);
        float sq = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) sq += part[1][w][rr];
        const float rstd = rsqrtf(sq * (1.f / C) + p.eps);
        if (wave == 0 && l31 < 2) {
            float sm = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) sm += part[0][w][rr];
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(l31 ? rstd : sm * (1.f / C)), rst, (uint32_t)(2 * r + l31) * 4u, 0, 0);
        }
        const uint32_t id = (uint32_t)r * (uint32_t)C + (uint32_t)col;
        const float o = (v[reg] * rstd * gm + bt) * mk16[reg];
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o), ry, id * 4u, 0, 0);
        const __bf16 ob = (__bf16)o;
        const unsigned short ou = *reinterpret_cast<const unsigned short*>(&ob);
        __builtin_amdgcn_raw_buffer_store_b16(ou, ryb, id * 2u, 0, 0);
        ytile[rr * LDT + col] = ou;
    }
    load_w(KS16 / 2, KS16);
    __syncthreads();
    // ---- qkv = y_bf16 Wqkv^T + bias: this wave's output fragments 3 wave .. 3 wave + 2 ----
    f32x16 acc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
#pragma unroll
    for (int k = 0; k < KS16; ++k) {
        const Chunk16 af = *reinterpret_cast<const Chunk16*>(&ytile[l31 * LDT + k * 16 + lhi * 8]);
#pragma unroll
        for (int f = 0; f < NF; ++f)
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&af), *reinterpret_cast<const bf16x8*>(&bfr[k][f]), acc[f], 0, 0, 0);
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int r = rb + rof(reg);
#pragma unroll
        for (int f = 0; f < NF; ++f)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[f][reg] + bq[f]), rq, (uint32_t)(r * N + (wave * NF + f) * 32 + l31) * 4u, 0, 0);
    }
}

// the short-K 1x1 problems of the path (LINEAR epilogue, fp32 A rows, bf16 MFMA, <= 192 columns): -1 = not one of them
inline int try_skinny(const glowtts_conv_args& a, hipStream_t s)
{
    if (!(GLOWTTS_TUNABLE("GLOWTTS_SKINNY", 1) && a.precision == GLOWTTS_BF16 && a.taps == 1 && a.epi == GLOWTTS_EPI_LINEAR && a.apro == GLOWTTS_APRO_NONE &&
          !(a.io_flags & GLOWTTS_IO_IN0_BF16) && !a.a2 && a.batch <= 1 && !(a.flags & (GLOWTTS_F_COLMASK | GLOWTTS_F_DROPOUT | GLOWTTS_F_GATE_IN0)) &&
          (a.ca % 16) == 0 && a.lda >= a.ca && (a.lda & 3) == 0 && a.kchunks * 32 >= a.ca && a.rows >= 32)) return -1;
    const int ni = (a.n + 31) / 32;
    if (ni * 32 > a.npad) return -1;
    if (a.io_flags & GLOWTTS_IO_A_BF16) {                                        // bf16-stored A rows: the Start data gradient on bf16 d h0
        if (a.ca == 192 && ni == 3 && (a.lda & 7) == 0 && !(a.io_flags & GLOWTTS_IO_OUT0_BF16)) return launch_skinny<12, 3, true>(a, s);
        return -1;
    }
    switch (a.ca / 16) {
        case 5:  if (ni == 6) return launch_skinny<5, 6>(a, s); break;           // Start conv: 80 -> 192
        case 12: if (ni == 3) return launch_skinny<12, 3>(a, s);                 // Start data gradient: 192 -> 80
                 if (ni == 6) return launch_skinny<12, 6>(a, s);                 // encoder 192 -> 192 projections
                 if (ni == 5) return launch_skinny<12, 5>(a, s); break;          // encoder Project: 192 -> 160
        default: break;
    }
    return -1;
}

inline const char* epi_name(int e)
{
    switch (e) { case GLOWTTS_EPI_LINEAR: return "LINEAR"; case GLOWTTS_EPI_GATE: return "GATE"; case GLOWTTS_EPI_RESSKIP: return "RESSKIP";
                 case GLOWTTS_EPI_COUPLE: return "COUPLE"; case GLOWTTS_EPI_DGATE: return "DGATE"; default: return "?"; }
}

template <int EPI1, int EPI2>
int launch_chain(const glowtts_conv_args& a1, const glowtts_conv_args& a2, hipStream_t s)
{
    constexpr int lds = 2 * ((DMA1_CPS * (CH_BM / 16) + DMA1_CPS * (CH_BN / 16)) * 1024) + CH_KC2 * CH_BM * 64;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_chain_kernel<EPI1, EPI2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return GLOWTTS_E_LAUNCH;
        attr_done = true;
    }
    GLOWTTS_NOTE_STATIC("conv_chain<%s,%s>", epi_name(EPI1), epi_name(EPI2));
    hipLaunchKernelGGL((conv_chain_kernel<EPI1, EPI2>), dim3((a1.rows + CH_BM - 1) / CH_BM), dim3(CH_WN * CH_WM * 64), lds, s, a1, a2);
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

int num_cus()
{
    static const int n = [] { int dev = 0; hipDeviceProp_t pr; if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 256;
                              return pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256; }();
    return n;
}

// is the LDS-DMA kernel applicable to this problem?
bool dma_ok(const glowtts_conv_args& a)
{
    const bool enabled = GLOWTTS_TUNABLE("GLOWTTS_DMA", 1) != 0;
    if (!(enabled && a.precision == GLOWTTS_BF16 && (a.io_flags & GLOWTTS_IO_A_BF16) && a.apro == GLOWTTS_APRO_NONE &&
          a.batch <= 1 && a.kchunks * 32 == a.ca && (a.npad % 64) == 0 && a.rows >= 128)) return false;
    if (a.epi == GLOWTTS_EPI_DGATE && (a.flags & GLOWTTS_F_COND_ROWS)) return false;       // the pitch-gradient variant lives in conv_cl_kernel only
    if (a.taps > 1) return !a.a2;
    // 1x1: whole stages of DMA1_CPS chunks; a second source must start on a stage boundary and share the row stride
    if (a.kchunks % DMA1_CPS) return false;
    return !a.a2 || ((a.ca1 % (32 * DMA1_CPS)) == 0 && a.lda2 == a.lda);
}

// 96-column strips (NI = 3) balance the SIMDs better on paper for the WaveNet In data gradient (12928 rows x 192 columns: 4 waves x 202
// workgroups, one wave per SIMD, instead of 5 x 243 with two waves on one SIMD), but one wave per SIMD hides no latency: measured
// 7.13 vs 7.00 ms/step.  Kept behind GLOWTTS_DMA_NI=3 for other shapes.
inline bool dma_prefers_96(const glowtts_conv_args& a)
{
    const int force = GLOWTTS_TUNABLE("GLOWTTS_DMA_NI", 0);
    return force == 3 && (a.npad % 96) == 0;
}

// 32-column strips (NI = 1) for k-tap convs whose 64-column tiling would leave half the chip idle (the text encoder's 768 -> 192 FFN
// convs at B = 32: 31 row tiles x 3 strips = 93 workgroups): twice the workgroups, half the MFMAs on each wave's dependent chain.
inline bool dma_prefers_32(const glowtts_conv_args& a)
{
    const long tiles64 = (long)(((a.rows + 31) / 32 + 3) / 4) * (a.npad / 64);
    return GLOWTTS_TUNABLE("GLOWTTS_DMA_NI1", 1) != 0 && (a.npad % 32) == 0 && 2 * tiles64 <= num_cus();
}

template <int EPI, int TAPS, int NI = 2>
int launch_dma(const glowtts_conv_args& a, hipStream_t s)
{
    // wave pairs over the two K halves (see conv_dma_kernel): measured on MI355X at B = 32 (tools/ab_conv.sh), a third less LDS fragment
    // traffic bought nothing - In_l forward 18.3 vs 17.0 us, data gradient 19.9 vs 19.6, 1x1 convs 13.0 vs 12.6: the K loop is not bound
    // by LDS read bandwidth but by its barrier / DMA-issue / latency structure.  Instantiated in tools builds only.
#ifdef GLOWTTS_TOOLS
    const int ksplit = GLOWTTS_TUNABLE("GLOWTTS_DMA_KSPLIT", 0);
#else
    constexpr int ksplit = 0;
#endif
    // waves per workgroup: all tiles resident at once (one workgroup per CU) if possible, else the fewest rounds
    const int force = GLOWTTS_TUNABLE("GLOWTTS_DMA_WAVES", 0);
    // GLOWTTS_DMA_LOADERS = 1 | 2: that many extra waves per workgroup do all the LDS-DMA staging (wave specialisation)
    const int nload_ = GLOWTTS_TUNABLE("GLOWTTS_DMA_LOADERS", 0), nload = (nload_ >= 0 && nload_ <= 4) ? nload_ : 0;
    // GLOWTTS_DMA_CUS: CUs the chain kernels plan for (default: all).  Leaving a few CUs to the concurrently running encoder stream
    // can pay: these kernels are latency-bound, a fatter workgroup on fewer CUs costs them little.
    const int cu_budget = GLOWTTS_TUNABLE("GLOWTTS_DMA_CUS", 0);
    const int gy = a.npad / (NI * 32), ncu = (cu_budget >= 32 && cu_budget <= num_cus()) ? cu_budget : num_cus(), frags = (a.rows + 31) / 32;
    int best = 4; long best_cost = -1;
    const int WMAX = (TAPS == 1 ? 10 : 16) - nload;       // three LDS stages must fit 160 KiB; at most 16 waves
    for (int w = 4; w <= WMAX; ++w) {
        const long tiles = (long)((frags + w - 1) / w) * gy;
        const long cost = ((tiles + ncu - 1) / ncu) * (w + 4);      // rounds x (strip work + fixed prologue / epilogue share)
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = w; }
    }
    // 1x1 convs (K <= 384: a short K loop between a load phase and a long epilogue): measured on the training step at B = 32
    // (tools/ab.sh, two boxes), 4-wave workgroups that co-reside three to a CU (48 KiB LDS each) beat the single fat one - their
    // pipelines drift apart, so one's epilogue (VALU / memory) runs under another's loads and MFMAs: 6.31 -> 6.25 and 6.42 -> 6.36
    // ms/step.  Only while every workgroup is resident at once; otherwise the round-count model above decides.  The same idea for the
    // k-tap convs (7 waves, two per CU) was box-dependent: -0.07 ms on one, +0.05 on the other; not adopted.
    const int coop = GLOWTTS_TUNABLE("GLOWTTS_DMA_COOP", 1);
    if (coop && !nload && TAPS == 1) {
        const long tiles = (long)((frags + 3) / 4) * gy;
        if (tiles > ncu && tiles <= 3L * ncu) best = 4;
    }
    // k-tap convs with few columns (the WaveNet In data gradient: 192 columns = 3 strips): the round model picks 5 waves x 243 workgroups,
    // i.e. 1.25 waves per SIMD; 7 waves x 174 workgroups hide more of the load latency and leave 80 CUs to the encoder stream's backward:
    // 6.21 -> 6.15 and 6.27 -> 6.21 ms/step on two boxes (8 waves: -0.04, 10: +0.05).  Only while that still fills half the chip.
    if (coop && !nload && TAPS > 1 && gy <= 3 && best < 7) {
        const long tiles = (long)((frags + 6) / 7) * gy;
        if (tiles <= ncu && 2 * tiles >= ncu) best = 7;
    }
    const int force_t1 = GLOWTTS_TUNABLE("GLOWTTS_DMA_WAVES_T1", 0);
    if (force >= 4 && force <= WMAX) best = force;
    const int force_t5 = GLOWTTS_TUNABLE("GLOWTTS_DMA_WAVES_T5", 0);
    if (TAPS == 1 && force_t1 >= 4 && force_t1 <= WMAX) best = force_t1;
    if (TAPS > 1 && force_t5 >= 4 && force_t5 <= WMAX) best = force_t5;
    const int force_t1n = GLOWTTS_TUNABLE("GLOWTTS_DMA_WAVES_T1_NARROW", 0);      // 1x1 convs with <= 192 columns
    if (TAPS == 1 && gy <= 3 && force_t1n >= 4 && force_t1n <= WMAX) best = force_t1n;
    const int force_t5n = GLOWTTS_TUNABLE("GLOWTTS_DMA_WAVES_T5_NARROW", 0);      // k-tap convs with <= 192 columns
    if (TAPS > 1 && gy <= 3 && force_t5n >= 4 && force_t5n <= WMAX) best = force_t5n;
    const bool ks = ksplit && !nload && NI == 2;
    if (ks && (best & 1)) best += (best < WMAX) ? 1 : -1;                 // wave pairs
    const int BM = best * 32;
    const int nat = TAPS == 1 ? DMA1_CPS : 1, sub = TAPS == 1 ? DMA1_CPS : TAPS;
    // LDS stages: 2 by default.  Alone, the kernel is as fast with 2 as with 3 (17.7 us either way); in the training step the smaller
    // footprint (83 instead of 124 KiB at 10 waves) lets encoder-stream workgroups share the CU: 7.1 vs 7.35 ms/step.
    const int nst0 = GLOWTTS_TUNABLE("GLOWTTS_DMA_STAGES", 2) == 3 ? 3 : 2;
    const int nst_narrow = GLOWTTS_TUNABLE("GLOWTTS_DMA_STAGES_NARROW", 0);
    const int nst = (TAPS > 1 && gy <= 3 && (nst_narrow == 2 || nst_narrow == 3)) ? nst_narrow : nst0;
    int lds = (nload ? 2 : nst) * ((nat * ((BM + TAPS - 1 + 15) >> 4) + sub * (NI * 2)) * 1024);     // three stages: <= 159 KiB (16 waves x 5 taps, 10 waves 1x1)
    if (ks) lds = std::max(lds, best * NI * 16 * 64 * 4);                 // the partial-sum exchange reuses the stage buffers
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_dma_kernel<EPI, TAPS, NI, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return GLOWTTS_E_LAUNCH;
#ifdef GLOWTTS_TOOLS
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_dma_kernel<EPI, TAPS, NI, NI == 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return GLOWTTS_E_LAUNCH;
#endif
        attr_done = true;
    }
    dim3 grid(((a.rows + BM - 1) / BM) * gy);
    GLOWTTS_NOTE_STATIC("conv_dma<%s,%d>", epi_name(EPI), TAPS);
#ifdef GLOWTTS_TOOLS
    if (ks) { hipLaunchKernelGGL((conv_dma_kernel<EPI, TAPS, NI, NI == 2>), grid, dim3(best * 64), lds, s, a, nst, 0); return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH; }
#endif
    hipLaunchKernelGGL((conv_dma_kernel<EPI, TAPS, NI, false>), grid, dim3((best + nload) * 64), lds, s, a, nload ? 2 : nst, nload);
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

template <typename CT, int MI, int NI, int WM, int WN, int EPI, int TAPS, int APRO, bool ABF>
int launch_k(const glowtts_conv_args& a, hipStream_t s)
{
    constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
    dim3 grid(((a.rows + BM - 1) / BM) * ((a.npad + BN - 1) / BN), 1, a.batch > 1 ? a.batch : 1);
    GLOWTTS_NOTE_STATIC("conv_cl<%s,%d,%s%s>", epi_name(EPI), TAPS, sizeof(CT) == 2 ? "bf16" : "f32", ABF ? ",abf" : "");
    hipLaunchKernelGGL((conv_cl_kernel<CT, MI, NI, WM, WN, EPI, TAPS, APRO, ABF>), grid, dim3(WM * WN * 64), 0, s, a);
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

template <typename CT, int EPI, int TAPS, int APRO, bool ABF = false>
int launch_tile(const glowtts_conv_args& a, hipStream_t s)
{
    // Tile choice.  These GEMMs are short (K <= 1920) and their operands come from L2, so a workgroup spends most of its
    // life waiting for a load phase or in its epilogue: what matters is how many workgroups are co-resident per CU to
    // overlap that, not the tile's arithmetic intensity.  GLOWTTS_TILE = 0: 128x128, 1: 64x128, 2: 128x64, 3: 64x64.
    const int force = GLOWTTS_TUNABLE("GLOWTTS_TILE", -1);
    const int force_t1 = GLOWTTS_TUNABLE("GLOWTTS_TILE_T1", -1);
    int cfg = (TAPS == 1 && force_t1 >= 0) ? force_t1 : force;
    if (cfg < 0) cfg = (TAPS == 1) ? 1 : 2;   // measured at B = 32: 128 x 64 for the k-tap convs (tools/bench_conv.py), 64 x 128 for the 1x1
                                              // convs whose wide fp32 A rows would otherwise be re-read by six N tiles (bench.py: 21.9 -> 20.3 ms)
    switch (cfg) {
        case 0: return launch_k<CT, 2, 2, 2, 2, EPI, TAPS, APRO, ABF>(a, s);
        case 1: return launch_k<CT, 1, 2, 2, 2, EPI, TAPS, APRO, ABF>(a, s);
        case 2: return launch_k<CT, 1, 2, 4, 1, EPI, TAPS, APRO, ABF>(a, s);
        default: return launch_k<CT, 1, 2, 2, 1, EPI, TAPS, APRO, ABF>(a, s);
    }
}

template <typename CT, int EPI, int APRO, bool ABF = false>
int launch_taps(const glowtts_conv_args& a, hipStream_t s)
{
    switch (a.taps) {
        case 1: return launch_tile<CT, EPI, 1, APRO, ABF>(a, s);
        case 3: return launch_tile<CT, EPI, 3, APRO, ABF>(a, s);
        case 5: return launch_tile<CT, EPI, 5, APRO, ABF>(a, s);
        default: return GLOWTTS_E_ARG;
    }
}

// the (epilogue, prologue, taps) combinations the Glow-TTS path uses; anything else is rejected
template <typename CT>
int launch_prec(const glowtts_conv_args& a, hipStream_t s)
{
    const int N = GLOWTTS_APRO_NONE, PM = GLOWTTS_APRO_PAIRMUL;
    if constexpr (sizeof(CT) == 2) {
        { const int rc = try_skinny(a, s); if (rc != -1) return rc; }
        if (dma_ok(a)) {
            if (a.epi == GLOWTTS_EPI_GATE && a.taps == 5) return launch_dma<GLOWTTS_EPI_GATE, 5>(a, s);
            if (a.epi == GLOWTTS_EPI_LINEAR && a.taps == 5) return dma_prefers_96(a) ? launch_dma<GLOWTTS_EPI_LINEAR, 5, 3>(a, s) : launch_dma<GLOWTTS_EPI_LINEAR, 5>(a, s);
            if (a.epi == GLOWTTS_EPI_GATE && a.taps == 3) return launch_dma<GLOWTTS_EPI_GATE, 3>(a, s);
            if (a.epi == GLOWTTS_EPI_LINEAR && a.taps == 3) return dma_prefers_32(a) ? launch_dma<GLOWTTS_EPI_LINEAR, 3, 1>(a, s) : launch_dma<GLOWTTS_EPI_LINEAR, 3>(a, s);
            if (a.epi == GLOWTTS_EPI_RESSKIP && a.taps == 1) return launch_dma<GLOWTTS_EPI_RESSKIP, 1>(a, s);
            if (a.epi == GLOWTTS_EPI_DGATE && a.taps == 1) return launch_dma<GLOWTTS_EPI_DGATE, 1>(a, s);
            if (a.epi == GLOWTTS_EPI_LINEAR && a.taps == 1) return launch_dma<GLOWTTS_EPI_LINEAR, 1>(a, s);
            if (a.epi == GLOWTTS_EPI_COUPLE && a.taps == 1) return launch_dma<GLOWTTS_EPI_COUPLE, 1>(a, s);
        }
        if (a.io_flags & GLOWTTS_IO_A_BF16) {      // bf16-stored A operand: the WaveNet state / gates / gate gradients
            if (a.epi == GLOWTTS_EPI_GATE && a.apro == N) return launch_taps<CT, GLOWTTS_EPI_GATE, GLOWTTS_APRO_NONE, true>(a, s);
            if (a.epi == GLOWTTS_EPI_LINEAR && a.apro == N) return launch_taps<CT, GLOWTTS_EPI_LINEAR, GLOWTTS_APRO_NONE, true>(a, s);
            if (a.epi == GLOWTTS_EPI_RESSKIP && a.apro == PM && a.taps == 1) return launch_tile<CT, GLOWTTS_EPI_RESSKIP, 1, GLOWTTS_APRO_PAIRMUL, true>(a, s);
            if (a.epi == GLOWTTS_EPI_RESSKIP && a.apro == N && a.taps == 1) return launch_tile<CT, GLOWTTS_EPI_RESSKIP, 1, GLOWTTS_APRO_NONE, true>(a, s);
            if (a.epi == GLOWTTS_EPI_DGATE && a.apro == N && a.taps == 1) return launch_tile<CT, GLOWTTS_EPI_DGATE, 1, GLOWTTS_APRO_NONE, true>(a, s);
            if (a.epi == GLOWTTS_EPI_COUPLE && a.apro == N && a.taps == 1) return launch_tile<CT, GLOWTTS_EPI_COUPLE, 1, GLOWTTS_APRO_NONE, true>(a, s);
            return GLOWTTS_E_ARG;
        }
    } else if (a.io_flags) return GLOWTTS_E_ARG;   // fp32 precision keeps every activation in fp32
    switch (a.epi) {
        case GLOWTTS_EPI_LINEAR:
            if (a.apro == N) return launch_taps<CT, GLOWTTS_EPI_LINEAR, GLOWTTS_APRO_NONE>(a, s);
            if (a.apro == PM && a.taps == 1) return launch_tile<CT, GLOWTTS_EPI_LINEAR, 1, GLOWTTS_APRO_PAIRMUL>(a, s);
            if constexpr (sizeof(CT) == 4) {
                if (a.apro == GLOWTTS_APRO_SQNEG && a.taps == 1) return launch_tile<CT, GLOWTTS_EPI_LINEAR, 1, GLOWTTS_APRO_SQNEG>(a, s);
            }
            return GLOWTTS_E_ARG;
        case GLOWTTS_EPI_GATE:    return a.apro == N ? launch_taps<CT, GLOWTTS_EPI_GATE, GLOWTTS_APRO_NONE>(a, s) : GLOWTTS_E_ARG;
        case GLOWTTS_EPI_RESSKIP: return (a.apro == PM && a.taps == 1) ? launch_tile<CT, GLOWTTS_EPI_RESSKIP, 1, GLOWTTS_APRO_PAIRMUL>(a, s) : GLOWTTS_E_ARG;
        case GLOWTTS_EPI_COUPLE:  return (a.apro == N && a.taps == 1) ? launch_tile<CT, GLOWTTS_EPI_COUPLE, 1, GLOWTTS_APRO_NONE>(a, s) : GLOWTTS_E_ARG;
        case GLOWTTS_EPI_DGATE:   return (a.apro == N && a.taps == 1) ? launch_tile<CT, GLOWTTS_EPI_DGATE, 1, GLOWTTS_APRO_NONE>(a, s) : GLOWTTS_E_ARG;
        default: return GLOWTTS_E_ARG;
    }
}

}  // namespace

namespace {
int pack_batched(const float* w, int batch, int O, int I, int taps, int transpose, int perm, int perm_h, int precision, void* packed,
                 int* npad_out, int* kchunks_out, int inner, long outer_stride, long inner_stride, long w_stride, void* stream);
}

extern "C" int glowtts_pack_weight_batched(const float* w, int batch, int O, int I, int taps, int transpose, int perm, int perm_h,
                                           int precision, void* packed, int* npad_out, int* kchunks_out, void* stream)
{
    return pack_batched(w, batch, O, I, taps, transpose, perm, perm_h, precision, packed, npad_out, kchunks_out, 0, 0, 0, 0, stream);
}

extern "C" int glowtts_pack_weight_strided(const float* w, int batch, int inner, int O, int I, int taps, int transpose, int perm, int perm_h,
                                           int precision, void* packed, int64_t outer_stride, int64_t inner_stride, int64_t w_stride, void* stream)
{
    if (inner < 1 || !packed || (outer_stride & 15) || (inner_stride & 15) || w_stride < 0) return GLOWTTS_E_ARG;
    return pack_batched(w, batch, O, I, taps, transpose, perm, perm_h, precision, packed, nullptr, nullptr, inner, (long)outer_stride, (long)inner_stride, (long)w_stride, stream);
}

namespace {
int pack_batched(const float* w, int batch, int O, int I, int taps, int transpose, int perm, int perm_h, int precision, void* packed,
                 int* npad_out, int* kchunks_out, int inner, long outer_stride, long inner_stride, long w_stride, void* stream)
{
    if (w_stride == 0) w_stride = (long)O * I * taps;
    if (batch < 1 || O < 1 || I < 1 || taps < 1 || taps > MAX_TAPS || (precision != GLOWTTS_F32 && precision != GLOWTTS_BF16)) return GLOWTTS_E_ARG;
    if (perm == GLOWTTS_PERM_PAIR && (perm_h < 1 || 2 * perm_h != O)) return GLOWTTS_E_ARG;
    const int KC = precision == GLOWTTS_BF16 ? 32 : 16;
    // logical extents of the O-side index after permutation (PAIR pads each half to a multiple of 32)
    const int o_ext = (perm == GLOWTTS_PERM_PAIR) ? pad_to(perm_h, 32) * 2 : O;
    const int N = transpose ? I : o_ext;
    const int K = transpose ? o_ext : I;
    const int npad = pad_to(N, 64);
    const int kchunks = (K + KC - 1) / KC;
    if (npad_out) *npad_out = npad;
    if (kchunks_out) *kchunks_out = kchunks;
    if (!packed) return GLOWTTS_OK;
    if (!w) return GLOWTTS_E_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long total = (long)taps * kchunks * npad * KC;
    const int blocks = (int)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256);
    GLOWTTS_NOTE_STATIC("pack_weight");
    if (precision == GLOWTTS_BF16)
        hipLaunchKernelGGL(pack_weight_kernel<__bf16>, dim3(blocks, batch), dim3(256), 0, s, w, static_cast<__bf16*>(packed), O, I, taps, transpose, perm, perm_h, N, kchunks * KC, npad, kchunks,
                           inner, outer_stride, inner_stride, w_stride);
    else
        hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(blocks, batch), dim3(256), 0, s, w, static_cast<float*>(packed), O, I, taps, transpose, perm, perm_h, N, kchunks * KC, npad, kchunks,
                           inner, outer_stride, inner_stride, w_stride);
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}
}  // namespace

extern "C" int glowtts_pack_job_init(glowtts_pack_job* job, const float* w, int O, int I, int taps, int transpose, int perm, int perm_h,
                                     int precision, void* packed, int block0, int* blocks_out, int64_t* bytes_out)
{
    if (!job) return GLOWTTS_E_ARG;
    int npad = 0, kchunks = 0;
    const int rc = glowtts_pack_weight_batched(nullptr, 1, O, I, taps, transpose, perm, perm_h, precision, nullptr, &npad, &kchunks, nullptr);
    if (rc != GLOWTTS_OK) return rc;
    const int KC = precision == GLOWTTS_BF16 ? 32 : 16;
    const int o_ext = (perm == GLOWTTS_PERM_PAIR) ? pad_to(perm_h, 32) * 2 : O;
    *job = glowtts_pack_job{};
    job->w = w; job->packed = packed; job->O = O; job->I = I; job->taps = taps; job->transpose = transpose; job->perm = perm; job->perm_h = perm_h;
    job->N = transpose ? I : o_ext; job->K = kchunks * KC; job->npad = npad; job->kchunks = kchunks; job->block0 = block0;
    const long total = (long)taps * kchunks * npad * KC;
    if (blocks_out) *blocks_out = (int)((total + PACK_CHUNK - 1) / PACK_CHUNK);
    if (bytes_out) *bytes_out = (int64_t)taps * kchunks * npad * 64;
    return GLOWTTS_OK;
}

extern "C" int glowtts_pack_weight_multi(const glowtts_pack_job* dev_jobs, int njobs, int total_blocks, int precision, void* stream)
{
    if (!dev_jobs || njobs < 1 || total_blocks < 1 || (precision != GLOWTTS_F32 && precision != GLOWTTS_BF16)) return GLOWTTS_E_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (precision == GLOWTTS_BF16) hipLaunchKernelGGL(pack_weight_multi_kernel<__bf16>, dim3(total_blocks), dim3(256), 0, s, dev_jobs, njobs);
    else hipLaunchKernelGGL(pack_weight_multi_kernel<float>, dim3(total_blocks), dim3(256), 0, s, dev_jobs, njobs);
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

extern "C" int glowtts_pack_weight(const float* w, int O, int I, int taps, int transpose, int perm, int perm_h,
                                   int precision, void* packed, int* npad_out, int* kchunks_out, void* stream)
{
    return glowtts_pack_weight_batched(w, 1, O, I, taps, transpose, perm, perm_h, precision, packed, npad_out, kchunks_out, stream);
}

extern "C" int glowtts_conv_cl(const glowtts_conv_args* args, void* stream)
{
    if (!args || !args->a || !args->w || !args->out0 || args->rows < 1 || args->taps < 1 || args->taps > MAX_TAPS) return GLOWTTS_E_ARG;
    if ((args->lda & 3) || (reinterpret_cast<uintptr_t>(args->a) & 15)) return GLOWTTS_E_ARG;      // rows start 16-byte aligned (lda % 8 for bf16 A, below)
    if (args->a2 && ((args->lda2 & 3) || (reinterpret_cast<uintptr_t>(args->a2) & 15))) return GLOWTTS_E_ARG;
    glowtts_conv_args a = *args;
    if (!a.a2 && a.apro != GLOWTTS_APRO_SQNEG) a.ca1 = a.ca;
    if (a.apro == GLOWTTS_APRO_SQNEG && ((a.ca1 % (a.precision == GLOWTTS_BF16 ? 8 : 4)) || a.ca != 2 * a.ca1)) return GLOWTTS_E_ARG;
    if ((a.flags & GLOWTTS_F_COLMASK) && !a.ncols_valid) return GLOWTTS_E_ARG;
    if ((a.flags & GLOWTTS_F_GATE_IN0) && (a.epi != GLOWTTS_EPI_LINEAR || !a.in0 || (a.flags & (GLOWTTS_F_ADD_IN0 | GLOWTTS_F_DROPOUT)) || a.drop_p < 0.f || a.drop_p >= 1.f)) return GLOWTTS_E_ARG;
    {   // loads are unconditional 16-byte vectors covering one LDS slot (E = 8 bf16 / 4 f32 channels): every row must be wide enough
        // to contain the last, possibly partial, slot:  lda >= round_up(channels read from it, E)   (lda in A elements)
        const int E = a.precision == GLOWTTS_BF16 ? 8 : 4;
        auto up = [E](int v) { return (v + E - 1) / E * E; };
        const bool abf = (a.io_flags & GLOWTTS_IO_A_BF16) != 0;
        if ((a.ca & 3) || a.kchunks * (a.precision == GLOWTTS_BF16 ? 32 : 16) < a.ca) return GLOWTTS_E_ARG;
        if ((a.io_flags & GLOWTTS_IO_OUT0_BF16) && (a.flags & GLOWTTS_F_ACCUM)) return GLOWTTS_E_ARG;
        // the epilogue addresses out0 / out1 / in0 / cond with 32-bit byte offsets (buffer descriptors)
        const int64_t ldmax = std::max(std::max(a.ld0, a.epi == GLOWTTS_EPI_DGATE ? 0 : a.ld1), a.ldi0);
        if ((int64_t)a.rows * ldmax * 4 >= (int64_t)1 << 31) return GLOWTTS_E_ARG;
        if (a.cond) {       // one conditioning row per utterance, or per activation row (GLOWTTS_F_COND_ROWS)
            const int64_t crows = (a.flags & GLOWTTS_F_COND_ROWS) ? a.rows : a.rows / std::max(a.rows_per_utt, 1);
            if (crows * a.ldcond * 4 >= (int64_t)1 << 31) return GLOWTTS_E_ARG;
        }
        if (abf && ((a.ca & 7) || (a.lda & 7) || (a.a2 && ((a.lda2 & 7) || (a.ca1 & 7))))) return GLOWTTS_E_ARG;
        if (a.apro == GLOWTTS_APRO_PAIRMUL) { if (a.lda < 2 * up(a.ca)) return GLOWTTS_E_ARG; }
        else if (a.apro == GLOWTTS_APRO_SQNEG) { if (a.lda < up(a.ca1)) return GLOWTTS_E_ARG; }
        else if (a.a2) { if ((a.ca1 % E) || a.lda < a.ca1 || a.lda2 < up(a.ca - a.ca1)) return GLOWTTS_E_ARG; }
        else if (a.lda < up(a.ca)) return GLOWTTS_E_ARG;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (a.precision == GLOWTTS_BF16) return launch_prec<__bf16>(a, s);
    if (a.precision == GLOWTTS_F32) return launch_prec<float>(a, s);
    return GLOWTTS_E_ARG;
}

extern "C" int glowtts_conv_chain(const glowtts_conv_args* first, const glowtts_conv_args* second, void* stream)
{
    if (!first || !second || !first->a || !first->w || !first->out0 || !second->w || !second->out0) return GLOWTTS_E_ARG;
    const glowtts_conv_args &a = *first, &b = *second;
    if (b.epi == GLOWTTS_EPI_DGATE && (b.flags & GLOWTTS_F_COND_ROWS)) return GLOWTTS_E_ARG;   // (see conv_epilogue's PITCH)
    // both 1x1, bf16 MFMA on bf16-stored rows; the intermediate has exactly 192 channels (Calc_Channels of the reference's model)
    if (a.precision != GLOWTTS_BF16 || b.precision != GLOWTTS_BF16 || a.taps != 1 || b.taps != 1 || !(a.io_flags & GLOWTTS_IO_A_BF16) ||
        a.apro != GLOWTTS_APRO_NONE || a.a2 || a.batch > 1 || b.batch > 1 || a.rows != b.rows || a.rows < 1) return GLOWTTS_E_ARG;
    if (a.n != CH_BN || a.npad != CH_BN || a.kchunks * 32 != a.ca || (a.kchunks % DMA1_CPS) || (a.lda & 7) || b.kchunks != CH_KC2 || b.npad != CH_BN ||
        b.ca != CH_BN) return GLOWTTS_E_ARG;
    const int64_t ldmax = std::max(std::max(std::max(a.ld0, a.ld1), std::max(b.ld0, b.epi == GLOWTTS_EPI_DGATE ? 0 : b.ld1)), b.ldi0);
    if ((int64_t)a.rows * ldmax * 4 >= (int64_t)1 << 31) return GLOWTTS_E_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (a.epi == GLOWTTS_EPI_RESSKIP && (a.flags & GLOWTTS_F_LAST) && b.epi == GLOWTTS_EPI_COUPLE) return launch_chain<GLOWTTS_EPI_RESSKIP, GLOWTTS_EPI_COUPLE>(a, b, s);
    if (a.epi == GLOWTTS_EPI_LINEAR && (a.flags & GLOWTTS_F_MASK) && b.epi == GLOWTTS_EPI_DGATE) return launch_chain<GLOWTTS_EPI_LINEAR, GLOWTTS_EPI_DGATE>(a, b, s);
    return GLOWTTS_E_ARG;
}

extern "C" int glowtts_proj_layernorm(const float* a, int64_t lda, const void* w, int npad, const float* bias, const float* x, const float* gamma,
                                      const float* beta, const float* rowmask, float* proj_kept, float* s, float* stats, float* y, uint16_t* y_bf16,
                                      int64_t rows, int C, float eps, float drop_p, uint32_t seed, const uint32_t* seed_ptr, void* stream)
{
    if (!a || !w || !bias || !x || !gamma || !beta || !rowmask || !s || !stats || !y || !y_bf16 || rows < 1) return GLOWTTS_E_ARG;
    if (C != 192 || npad < 192 || lda < C || (lda & 3) || rows * C * 4 >= ((int64_t)1 << 31) || drop_p < 0.f || drop_p >= 1.f) return GLOWTTS_E_ARG;
    if (drop_p > 0.f && !proj_kept) return GLOWTTS_E_ARG;
    proj_ln_args p;
    p.a = a; p.lda = lda; p.w = w; p.npad = npad; p.bias = bias; p.x = x; p.gamma = gamma; p.beta = beta; p.rowmask = rowmask;
    p.proj = drop_p > 0.f ? proj_kept : nullptr; p.s = s; p.stats = stats; p.y = y; p.yb = y_bf16;
    p.rows = rows; p.eps = eps; p.drop_p = drop_p; p.seed = seed; p.seed_ptr = seed_ptr;
    GLOWTTS_NOTE_STATIC("proj_ln<%d>", 192);
    hipLaunchKernelGGL(proj_ln_kernel, dim3((unsigned)((rows + 31) / 32)), dim3(384), 0, static_cast<hipStream_t>(stream), p);
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

extern "C" int glowtts_layernorm_qkv(const float* a, const float* b, const float* gamma, const float* beta, const float* rowmask, float* s, float* stats,
                                     float* y, uint16_t* y_bf16, const void* wqkv, int npad, const float* bias, float* qkv, int64_t rows, int C, float eps,
                                     void* stream)
{
    if (!a || !b || !gamma || !beta || !rowmask || !s || !stats || !y || !y_bf16 || !wqkv || !bias || !qkv || rows < 1) return GLOWTTS_E_ARG;
    if (C != 192 || npad < 576 || rows * 576 * 4 >= ((int64_t)1 << 31)) return GLOWTTS_E_ARG;
    ln_qkv_args p;
    p.a = a; p.b = b; p.gamma = gamma; p.beta = beta; p.rowmask = rowmask; p.s = s; p.stats = stats; p.y = y; p.yb = y_bf16;
    p.w = wqkv; p.npad = npad; p.bias = bias; p.qkv = qkv; p.rows = rows; p.eps = eps;
    GLOWTTS_NOTE_STATIC("ln_qkv<%d>", 192);
    hipLaunchKernelGGL(ln_qkv_kernel, dim3((unsigned)((rows + 31) / 32)), dim3(384), 0, static_cast<hipStream_t>(stream), p);
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}


// Weight gradient of the channels-last convolution for gfx950:
//     dW[o][c][t] (+)= sum_r DY[r][o] * X[r + t - pad][c]          dbias[o] (+)= sum_r DY[r][o]
// (autograd of Modules.py:791,861,871,793 and of the encoder convs).
//
// CDNA4 mapping
//   * MFMA M = output channel o, N = input channel c, K = rows (utterance, frame).  The reduction runs
//     over the SLOW (row) axis of both channels-last operands, so the fragments must be read
//     transposed out of LDS:
//       bf16: tiles are kept in their natural [row][channel] layout and read with ds_read_b64_tr_b16;
//             probed on gfx950 (tools/probe_tr16.hip): in a 16-lane group, lane s supplies the address of
//             4 contiguous bf16 and lane i receives element (i & 3) of the rows supplied by lanes
//             (i >> 2) + 4e, e = 0..3.  With lane s pointing at [k0 + (s >> 2)][c0 + 4 (s & 3)] lane i gets
//             channel c0 + i for the 4 rows k0..k0+3 - two reads give the 8 k-values of a 32x32x16 fragment.
//       f32 : v_mfma_f32_32x32x2_f32 takes one k per lane: plain ds_read_b32 from the same natural layout.
//   * all taps share one staged X tile ([32 + taps - 1 rows][64 channels]); a tap is a row offset of the
//     fragment address, so X is read once for the 5 taps.  Accumulators: 128(o) x 64(c) x taps per block.
//   * rows are split over gridDim.z (split-K); partial tiles are combined with fp32 atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include "../../include/glowtts_hip.h"
#include "tunable.h"
#include "launch_log.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BMO = 128;      // o per block
constexpr int BNC = 64;       // c per block
constexpr int BK = 64;        // rows per step
#ifndef WGRAD_SETS_MAXTAPS
#define WGRAD_SETS_MAXTAPS 1
#endif
#ifndef WGRAD_SETS
#define WGRAD_SETS 2
#endif
#ifndef WGRAD_WAVES
#define WGRAD_WAVES 8
#endif
constexpr int NT = WGRAD_WAVES * 64;          // threads per workgroup: 8 waves = 4 (o) x 2 (c) wave tiles of 32 x 32 (x taps)
constexpr int MI = WGRAD_WAVES == 8 ? 1 : 2; // 32-row fragments of o per wave

__device__ __forceinline__ uint32_t pk_bf16(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v; v[0] = (__bf16)lo; v[1] = (__bf16)hi;
    return *reinterpret_cast<uint32_t*>(&v);
}

template <int V> struct WIC { static constexpr int value = V; };
template <int N> struct WStaticFor {
    template <class F> __device__ __forceinline__ static void run(F&& f) { WStaticFor<N - 1>::run(f); f(WIC<N - 1>{}); }
};
template <> struct WStaticFor<0> { template <class F> __device__ __forceinline__ static void run(F&&) {} };

struct WCommon { int rows, pad, accumulate, njobs, xcd; int phase = 0; };      // phase: tiles of the first phase (glowtts_wgrad_grouped_phased), 0 = one phase

// pointers that come out of the device job table are "generic" to the compiler, which would emit flat_load (counted on
// BOTH vmcnt and lgkmcnt, i.e. every LDS wait would also drain the prefetch).  They are global: say so.
typedef float f32x4n __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldg4(const float* p) {
    const f32x4n v = *(const f32x4n __attribute__((address_space(1)))*)(p);
    return make_float4(v[0], v[1], v[2], v[3]);
}

// NW 32-bit words from a global byte address (8-byte aligned for NW == 2, 16-byte otherwise)
typedef uint32_t u32x2n __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4n __attribute__((ext_vector_type(4)));
template <int NW>
__device__ __forceinline__ void ldgw(uint32_t* dst, const unsigned char* src) {
    if constexpr (NW == 2) {
        const u32x2n v = *(const u32x2n __attribute__((address_space(1)))*)(src);
        dst[0] = v[0]; dst[1] = v[1];
    } else {
#pragma unroll
        for (int j = 0; j < NW / 4; ++j) {
            const u32x4n v = *(const u32x4n __attribute__((address_space(1)))*)(src + 16 * j);
            dst[4 * j] = v[0]; dst[4 * j + 1] = v[1]; dst[4 * j + 2] = v[2]; dst[4 * j + 3] = v[3];
        }
    }
}

// job of this workgroup: p (by value when there is a single problem, else looked up in the device table by tile index)
// DYBF / XBF: DY / X are stored as bf16 in HBM (bf16 precision only; glowtts_wgrad_args.io_flags) - then staging is a raw copy.
// WIDE (both operands stored as bf16, no X prologue): 8 channels = 16 bytes per staged item instead of 4 - half the loads, LDS stores and
// mask selects per MFMA (the 1x1 problems do 4 MFMAs per wave and 64-row step: staging instructions are what they spend their time on)
template <typename CT, int TAPS, int XPRO, bool DYBF, bool XBF, bool WIDE = false>
__global__ __launch_bounds__(NT) void wgrad_kernel(const glowtts_wgrad_job single, const glowtts_wgrad_job* __restrict__ table, const WCommon cm)
{
    glowtts_wgrad_job p = single;
    int tile = blockIdx.x;
    if (cm.xcd) {
        // Workgroup b runs on XCD b % 8, each with its own L2.  Hand every XCD a CONTIGUOUS range of tiles: the mt x nt tiles of one job
        // (which all stream the same DY / X rows) then share one L2 instead of pulling 8 copies of both operands over the fabric.
        const int G = gridDim.x, q = G >> 3, r = G & 7, x = tile & 7, k = tile >> 3;
        tile = x * q + min(x, r) + k;
    }
    if (table) {                                  // binary search for the last job with tile0 <= tile (wave-uniform)
        int lo = 0, hi = cm.njobs - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (table[mid].tile0 <= tile) lo = mid; else hi = mid - 1; }
        p = table[lo];
        tile -= p.tile0;
    }
    const int tile_o = tile % p.mt, tile_c = tile / p.mt;
    constexpr int ES = sizeof(CT);
    constexpr int XROWS = BK + TAPS - 1;
    // LDS row strides (bytes): natural width + 64 B so that 4 consecutive rows hit 4 different 64-B bank segments
    constexpr int LDY = BMO * ES + 64;
    constexpr int LDX = BNC * ES + 64;
    constexpr int DY_BYTES = BK * LDY, X_BYTES = XROWS * LDX;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (DY_BYTES + X_BYTES)];
    constexpr int GRP = WIDE ? 8 : 4;                         // channels per staged item
    static_assert(!WIDE || (DYBF == XBF && XPRO == GLOWTTS_APRO_NONE && sizeof(CT) == 2), "8-channel items: bf16 MFMA, no prologue, both operands stored alike");
    float (*bias_red)[BMO] = reinterpret_cast<float (*)[BMO]>(smem);      // [NT / (BMO / GRP)][BMO], used after the last step (<= 16 KiB)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;                 // wave tile: MI * 32 (o) x 32 (c)
    const int o0 = tile_o * BMO, c0 = tile_c * BNC;
    const int l31 = lane & 31, lhi = lane >> 5;

    // rows handled by this split
    const long chunk = ((((long)cm.rows + gridDim.z - 1) / gridDim.z) + BK - 1) / BK * BK;
    const long rbeg = (long)blockIdx.z * chunk;
    const long rend = min((long)cm.rows, rbeg + chunk);
    if (rbeg >= rend) return;
    const int nsteps = (int)((rend - rbeg + BK - 1) / BK);

    f32x16 acc[MI][TAPS];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][t][r] = 0.f;

    // ---- staging: raw unconditional loads (clamped addresses) two steps ahead; masks / prologue / bf16 conversion at the
    //      LDS store (same reasoning as conv_cl_kernel: a fixed number of loads per step keeps the vmcnt waits counted) ----
    constexpr int DY_IT = (BK * BMO / GRP) / NT;              // channel groups per thread for the DY tile
    constexpr int X_IT = (XROWS * BNC / GRP + NT - 1) / NT;   // channel groups per thread for the X tile
    constexpr int XL = (XPRO == GLOWTTS_APRO_PAIRMUL) ? 2 : 1;
    static_assert(!(DYBF || XBF) || ES == 2, "bf16 activation storage needs bf16 precision");
    // raw register image of one item = 4 (x XL) stored elements: 16 B (f32) / 8 B (bf16) per 4 elements
    constexpr int DYW = DYBF ? GRP / 2 : GRP;                 // 32-bit words per DY item
    constexpr int XW = (XBF ? GRP / 2 : GRP) * XL;            // 32-bit words per X item
    typedef uint32_t DYRegs[DY_IT][DYW];
    typedef uint32_t XRegs[X_IT][XW];
    // NS register sets = loads of NS - 1 steps in flight.  A step of the 1x1 problems is 4 MFMAs per wave, far shorter than the HBM
    // latency: with two sets every step waited a full round trip (1.45 us per 64 rows, measured)
    constexpr int NS = (TAPS <= WGRAD_SETS_MAXTAPS && (DY_IT * DYW + X_IT * XW) <= 24) ? WGRAD_SETS : 2;
    DYRegs rdy[NS];
    XRegs rx[NS];
    float bsum[GRP];
#pragma unroll
    for (int e = 0; e < GRP; ++e) bsum[e] = 0.f;
    const bool want_bias = (p.dbias != nullptr) && (tile_c == 0);
    const int lim_dy = (int)p.lddy - GRP;
    const int lim_x = (int)p.ldx - GRP * XL;

    auto gload = [&](DYRegs& rdy, XRegs& rx, long r0) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < DY_IT; ++it) {
            const int idx = tid + it * NT;
            const int row = idx / (BMO / GRP), c4 = idx % (BMO / GRP);
            const long r = min(r0 + row, (long)cm.rows - 1);
            ldgw<DYW>(rdy[it], reinterpret_cast<const unsigned char*>(p.dy) + (r * p.lddy + min(o0 + c4 * GRP, lim_dy)) * (DYBF ? 2 : 4));
        }
#pragma unroll
        for (int it = 0; it < X_IT; ++it) {
            const int idx = tid + it * NT;
            const int row = idx / (BNC / GRP), c4 = idx % (BNC / GRP);
            long r = r0 + row - cm.pad;
            r = r < 0 ? 0 : (r >= cm.rows ? cm.rows - 1 : r);
            ldgw<XW>(rx[it], reinterpret_cast<const unsigned char*>(p.x) + (r * p.ldx + min((c0 + c4 * GRP) * XL, lim_x)) * (XBF ? 2 : 4));
        }
    };
    auto bf_lo = [](uint32_t w) { return __uint_as_float(w << 16); };
    auto bf_hi = [](uint32_t w) { return __uint_as_float(w & 0xFFFF0000u); };
    auto sstore = [&](const DYRegs& rdy, const XRegs& rx, int buf, long r0) __attribute__((always_inline)) {
        unsigned char* dyb = smem + buf * (DY_BYTES + X_BYTES);
        unsigned char* xb = dyb + DY_BYTES;
#pragma unroll
        for (int it = 0; it < DY_IT; ++it) {
            const int idx = tid + it * NT;
            const int row = idx / (BMO / GRP), c4 = idx % (BMO / GRP);
            const bool ok = (r0 + row < rend) && (o0 + c4 * GRP < p.m);          // m is a multiple of GRP (checked on the host / promised by WIO_WIDE)
            if constexpr (WIDE && DYBF) {
                uint32_t w[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { w[e] = ok ? rdy[it][e] : 0u; bsum[2 * e] += bf_lo(w[e]); bsum[2 * e + 1] += bf_hi(w[e]); }
                *reinterpret_cast<uint4*>(dyb + row * LDY + c4 * 16) = make_uint4(w[0], w[1], w[2], w[3]);
            } else if constexpr (WIDE) {          // 8 fp32 values -> one 16-byte LDS store of 8 bf16
                uint32_t w[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = ok ? __uint_as_float(rdy[it][2 * e]) : 0.f, b = ok ? __uint_as_float(rdy[it][2 * e + 1]) : 0.f;
                    bsum[2 * e] += a; bsum[2 * e + 1] += b;
                    w[e] = pk_bf16(a, b);
                }
                *reinterpret_cast<uint4*>(dyb + row * LDY + c4 * 16) = make_uint4(w[0], w[1], w[2], w[3]);
            } else if constexpr (DYBF) {
                const uint32_t w0 = ok ? rdy[it][0] : 0u, w1 = ok ? rdy[it][1] : 0u;
                bsum[0] += bf_lo(w0); bsum[1] += bf_hi(w0); bsum[2] += bf_lo(w1); bsum[3] += bf_hi(w1);
                *reinterpret_cast<uint2*>(dyb + row * LDY + c4 * 8) = make_uint2(w0, w1);
            } else {
                float4 v = make_float4(__uint_as_float(rdy[it][0]), __uint_as_float(rdy[it][1]), __uint_as_float(rdy[it][2]), __uint_as_float(rdy[it][3]));
                v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
                bsum[0] += v.x; bsum[1] += v.y; bsum[2] += v.z; bsum[3] += v.w;
                if constexpr (ES == 2) {
                    uint2 w = make_uint2(pk_bf16(v.x, v.y), pk_bf16(v.z, v.w));
                    *reinterpret_cast<uint2*>(dyb + row * LDY + c4 * 8) = w;
                } else {
                    *reinterpret_cast<float4*>(dyb + row * LDY + c4 * 16) = v;
                }
            }
        }
#pragma unroll
        for (int it = 0; it < X_IT; ++it) {
            const int idx = tid + it * NT;
            const int row = idx / (BNC / GRP), c4 = idx % (BNC / GRP);
            if (row >= XROWS) continue;
            const long r = r0 + row - cm.pad;
            // rows outside [0, rows) are zero; rows outside this split's range ARE used (halo of the split)
            const bool ok = (r >= 0) && (r < cm.rows) && (c0 + c4 * GRP < p.ca);  // ca is a multiple of GRP (checked on the host / WIO_WIDE)
            if constexpr (WIDE && XBF) {
                *reinterpret_cast<uint4*>(xb + row * LDX + c4 * 16) = make_uint4(ok ? rx[it][0] : 0u, ok ? rx[it][1] : 0u, ok ? rx[it][2] : 0u, ok ? rx[it][3] : 0u);
            } else if constexpr (WIDE) {
                uint32_t w[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    w[e] = ok ? pk_bf16(__uint_as_float(rx[it][2 * e]), __uint_as_float(rx[it][2 * e + 1])) : 0u;
                *reinterpret_cast<uint4*>(xb + row * LDX + c4 * 16) = make_uint4(w[0], w[1], w[2], w[3]);
            } else if constexpr (XBF && XPRO == GLOWTTS_APRO_NONE) {
                *reinterpret_cast<uint2*>(xb + row * LDX + c4 * 8) = make_uint2(ok ? rx[it][0] : 0u, ok ? rx[it][1] : 0u);
            } else {
                float4 v;
                if constexpr (XBF) {          // PAIRMUL on bf16 (tanh, sigmoid) words
                    v = make_float4(bf_lo(rx[it][0]) * bf_hi(rx[it][0]), bf_lo(rx[it][1]) * bf_hi(rx[it][1]), bf_lo(rx[it][2]) * bf_hi(rx[it][2]), bf_lo(rx[it][3]) * bf_hi(rx[it][3]));
                } else if constexpr (XPRO == GLOWTTS_APRO_PAIRMUL) {
                    v = make_float4(__uint_as_float(rx[it][0]) * __uint_as_float(rx[it][1]), __uint_as_float(rx[it][2]) * __uint_as_float(rx[it][3]),
                                    __uint_as_float(rx[it][XW - 4]) * __uint_as_float(rx[it][XW - 3]), __uint_as_float(rx[it][XW - 2]) * __uint_as_float(rx[it][XW - 1]));
                } else {
                    v = make_float4(__uint_as_float(rx[it][0]), __uint_as_float(rx[it][1]), __uint_as_float(rx[it][2]), __uint_as_float(rx[it][3]));
                }
                v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
                if constexpr (ES == 2) {
                    uint2 w = make_uint2(pk_bf16(v.x, v.y), pk_bf16(v.z, v.w));
                    *reinterpret_cast<uint2*>(xb + row * LDX + c4 * 8) = w;
                } else {
                    *reinterpret_cast<float4*>(xb + row * LDX + c4 * 16) = v;
                }
            }
        }
    };

    auto compute = [&](int buf) __attribute__((always_inline)) {
        const unsigned char* dyb = smem + buf * (DY_BYTES + X_BYTES);
        const unsigned char* xb = dyb + DY_BYTES;
        if constexpr (ES == 2) {
            // transposed fragment reads: 16-lane group gq = (lane >> 4) & 1 covers channels 16*gq..+15 of the 32-wide fragment
            const int s = lane & 15, gq = (lane >> 4) & 1;
#pragma unroll
            for (int k16 = 0; k16 < BK / 16; ++k16) {
                const int krow = k16 * 16 + 8 * lhi + (s >> 2);          // + 4 for the second read
                bf16x8 af[MI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int col = (wm * MI + mi) * 32 + gq * 16 + 4 * (s & 3);
                    const unsigned char* a0 = dyb + krow * LDY + col * 2;
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0 + 4 * LDY));
                    s16x4 tmp[2] = {lo, hi};
                    af[mi] = *reinterpret_cast<bf16x8*>(tmp);
                }
#pragma unroll
                for (int t = 0; t < TAPS; ++t) {
                    const int col = wn * 32 + gq * 16 + 4 * (s & 3);
                    const unsigned char* b0 = xb + (krow + t) * LDX + col * 2;
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(b0));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(b0 + 4 * LDX));
                    s16x4 tmp[2] = {lo, hi};
                    const bf16x8 bfr = *reinterpret_cast<bf16x8*>(tmp);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        acc[mi][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bfr, acc[mi][t], 0, 0, 0);
                }
            }
        } else {
#pragma unroll 4
            for (int k2 = 0; k2 < BK / 2; ++k2) {
                const int krow = k2 * 2 + lhi;
                float af[MI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    af[mi] = *reinterpret_cast<const float*>(dyb + krow * LDY + ((wm * MI + mi) * 32 + l31) * 4);
#pragma unroll
                for (int t = 0; t < TAPS; ++t) {
                    const float bv = *reinterpret_cast<const float*>(xb + (krow + t) * LDX + (wn * 32 + l31) * 4);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        acc[mi][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi], bv, acc[mi][t], 0, 0, 0);
                }
            }
        }
    };

    // NS steps per iteration so that the register sets are selected at compile time; every load is unconditional
    // (rows past the end are clamped, their contribution is masked to zero when stored)
    WStaticFor<NS>::run([&](auto j) __attribute__((always_inline)) { gload(rdy[j.value], rx[j.value], rbeg + (long)j.value * BK); });
    sstore(rdy[0], rx[0], 0, rbeg);
    __syncthreads();
    for (int s = 0; s < nsteps; s += NS) {
        const long rs = rbeg + (long)s * BK;
        WStaticFor<NS>::run([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = jc.value, jn = (j + 1) % NS;
            gload(rdy[j], rx[j], rs + (long)(j + NS) * BK);              // set j was stored for step s + j: refill it for step s + j + NS
            if (s + j < nsteps) compute(j & 1);
            sstore(rdy[jn], rx[jn], (j + 1) & 1, rs + (long)(j + 1) * BK);
            __syncthreads();
        });
    }

    // ---- epilogue: dW[o][c][t] ----
    const bool atomic = gridDim.z > 1 || cm.accumulate;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int pcol = o0 + (wm * MI + mi) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lhi;    // DY column (possibly PAIR-packed)
            if (pcol >= p.m) continue;
            int o = pcol;
            if (p.perm == GLOWTTS_PERM_PAIR) {
                const int j = (pcol >> 6) * 32 + (pcol & 31);
                if (j >= p.perm_h) continue;
                o = ((pcol >> 5) & 1) * p.perm_h + j;
            }
            const int c = c0 + wn * 32 + l31;
            if (c >= p.ca) continue;
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                float* dst = p.dw + ((long)o * p.ca + c) * TAPS + t;
                if (atomic) atomicAdd(dst, acc[mi][t][reg]);
                else *dst = acc[mi][t][reg];
            }
        }
    }
    // ---- dbias[o] = column sums of DY (first c-tile only) ----
    if (want_bias) {
        const int c4 = tid % (BMO / GRP), rgrp = tid / (BMO / GRP);   // NT / (BMO / GRP) row groups share each column group
#pragma unroll
        for (int e = 0; e < GRP; ++e) bias_red[rgrp][c4 * GRP + e] = bsum[e];
        __syncthreads();
        if (tid < BMO) {
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < NT / (BMO / GRP); ++g) s += bias_red[g][tid];
            const int pcol = o0 + tid;
            if (pcol < p.m) {
                int o = pcol; bool ok = true;
                if (p.perm == GLOWTTS_PERM_PAIR) {
                    const int j = (pcol >> 6) * 32 + (pcol & 31);
                    ok = j < p.perm_h;
                    o = ((pcol >> 5) & 1) * p.perm_h + j;
                }
                if (ok) { if (atomic) atomicAdd(p.dbias + o, s); else p.dbias[o] = s; }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// wgrad_dma_kernel (round 4): the same weight gradient for problems whose operands are BOTH bf16 rows tensors with no prologue (the decoder's In_l
// group: 36 problems of 384 x 192 x 5 taps over 12 928 rows), rebuilt on what tools/wn_lab.hip measured:
//   * ds_read_b64_tr_b16 delivers ~118 B/clk per CU (profiles/r04_wn_lab_tr.txt), not the 256 of plain 8-byte reads: the staged kernel's 32 x 32
//     (x taps) wave tiles need 1.2 transposing reads per 16x16x32-equivalent MFMA - ~1 950 clocks of reads per workgroup step against 1 280 of MFMAs.
//     Wave tiles of 96 (o) x 32 (c) x taps here: 0.53 reads per MFMA, 240 accumulators a wave at 5 taps, one workgroup of four waves per CU.
//   * operands go global -> LDS by LDS-DMA (buffer_load ... lds, 16 bytes per lane): no VGPR staging, no ds_write (316 of the staged kernel's 1 280
//     MFMA clocks per step).  Rows outside [0, rows) - the tap halo, a partial last step - read as zeros through the descriptor's bounds check.
//   * v_mfma_f32_16x16x32_bf16: between back-to-back 32x32x16 MFMAs a SIMD issues no vector-memory instruction.
//   * tiles stay row-major (that is how the DMA delivers them), padded so that every transposing read is conflict-free AND addressed by one
//     lane-constant register plus an immediate; ds_read_b64_tr_b16 transposes.
//   * one software pipeline across steps: fragments two iterations ahead, DMAs two to three steps ahead, one barrier per step (below).
// Alone on the chip, the In_l group: 338 us (1.01 PFLOP/s, 216 workgroups) against 491-540 us for the staged kernel; MFMAs alone 216 us, DMAs
// alone 155 us, reads alone 114 us (tools/bench_wgrad.py and its ablations).
// ------------------------------------------------------------------------------------------------
typedef float f32x4w __attribute__((ext_vector_type(4)));
typedef int i32x4w __attribute__((ext_vector_type(4)));
constexpr int DBK = 64, DBMO = 192, DNW = 4, DNT = DNW * 64;
constexpr int DMO = 6;                                       // 16-channel o fragments of a wave tile; waves: 2 (o halves of 96) x 2 (c halves)
// LDS layouts, chosen so that EVERY fragment read is one lane-constant VGPR plus an immediate (the 56 reads of a step with XOR-swizzled natural rows took ~30
// address registers and spilled the accumulator tile).  The k index of the MFMA is free as long as both operands agree: element e = r + 4 h of lane group
// kg is tile row 32 k + kg + 4 r + 16 h, so the 16 lane rows of one transposing read are 16 CONSECUTIVE tile rows.
// Both tiles are row-major with rows of (tile channels x 2 + 32 pad) bytes - an odd multiple of 32: the 16 consecutive rows x 32 bytes of a read land on the 8
// bank groups twice each (the minimum for 512 bytes) at any tap shift.  (A fragment-ordered DY tile - 512 contiguous bytes per read, no pad - reads as well
// but its DMAs gather 64-byte pieces of 16 rows each and ran at ~24 B/clk per CU; whole 384-byte rows per instruction restore the streaming rate.)
// The DMA places any 16 global bytes at any 16 LDS bytes (per-lane source offsets), so both layouts cost nothing to produce; pad lanes fetch zeros from beyond
// the descriptor.
constexpr int DYRS = DBMO * 2 + 32, DDY_BYTES = DBK * DYRS, DNDY = DDY_BYTES / 1024;       // 416-byte rows, 26 KiB = 26 DMA instructions per stage
static_assert(DDY_BYTES % 1024 == 0 && DYRS % 64 == 32, "DY tile geometry");
// Tile shape per tap count.  k taps re-use a DY fragment k times, so 5 taps are MFMA-bound on a 192 x 64 tile (240 accumulators a wave); at 1 tap the same
// tile would move 35 KiB per 384 MFMA clocks - the one-tap kernel takes ALL of a 192-channel input per tile instead (192 x 192: DY is read once, 144
// accumulators, 51 KiB per 1 152 MFMA clocks) and runs on as few CUs as it has tiles, leaving the rest of the chip to the launches beside it.
template <int TAPS> struct DCfg {
    static constexpr int NC = TAPS == 1 ? 6 : 2;             // 16-channel c fragments of a wave tile
    static constexpr int BNC = 2 * 16 * NC;                  // tile input channels: 192 / 64
    static constexpr int XRS = BNC * 2 + 32;                 // X row stride in LDS: 416 / 160 bytes
    static constexpr int X_BYTES = (DBK + TAPS - 1) * XRS;   // rows of a step + the tap halo
    static constexpr int STAGE = DDY_BYTES + X_BYTES;        // 53 248 (x 3 stages) / 37 504 (x 4 stages) bytes: one workgroup per CU
    static constexpr int STAGES = TAPS == 1 ? 3 : 4;
    static constexpr int NDMA = DNDY + (X_BYTES + 1023) / 1024;
    static constexpr int SLOTS = (NDMA + DNW - 1) / DNW;     // DMA instructions per wave and stage: 13 / 9
    static_assert(XRS % 64 == 32 && STAGE % 16 == 0 && STAGES * STAGE <= 160 * 1024, "LDS geometry");
};

// The 240 accumulator registers of a wave live in AGPRs BY CONSTRUCTION: left to the register allocator the 5-tap kernel came out with 428 v_accvgpr moves and
// scratch reloads (each behind an s_waitcnt vmcnt(0) that also waits for the DMAs in flight) inside the step loop.
static __device__ __forceinline__ void mfma_acc(f32x4w& c, const bf16x8& a, const bf16x8& b)
{
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

template <int TAPS>
__global__ __launch_bounds__(DNT) void wgrad_dma_kernel(const glowtts_wgrad_job* __restrict__ table, const WCommon cm)
{
    typedef DCfg<TAPS> K;
    constexpr int DNC = K::NC, DBNC = K::BNC, DXRS = K::XRS, DX_BYTES = K::X_BYTES, DSTAGE = K::STAGE, DSTAGES = K::STAGES, DNDMA = K::NDMA, DSLOTS = K::SLOTS;
    extern __shared__ __attribute__((aligned(1024))) unsigned char dsm[];
    int tile = blockIdx.x;
    if (cm.xcd) {
        // workgroup b runs on XCD b & 7: the workgroups of an XCD take a contiguous run of tiles (a problem's tiles share operand columns: one L2 fetches them).
        // Phased launches (cm.phase): the same inside each phase [lo, hi) of workgroups / tiles.  cnt(N, x) = workgroups below N on XCD x.
        const int b = tile, x = b & 7;
        const int plo = (cm.phase > 0 && b >= cm.phase) ? cm.phase : 0, phi = (cm.phase > 0 && b < cm.phase) ? cm.phase : (int)gridDim.x;
        auto cnt = [](int N, int y) __attribute__((always_inline)) -> int { return (N - y + 7) >> 3; };
        int off = 0;
        for (int y = 0; y < x; ++y) off += cnt(phi, y) - cnt(plo, y);
        tile = plo + off + cnt(b, x) - cnt(plo, x);
    }
    int lo = 0, hi = cm.njobs - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (table[mid].tile0 <= tile) lo = mid; else hi = mid - 1; }
    const glowtts_wgrad_job p = table[lo];
    tile -= p.tile0;
    const int tile_o = tile % p.mt, tile_c = tile / p.mt;
    const int o0 = tile_o * DBMO, c0 = tile_c * DBNC;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                 // wave tile: o [96 wm, + 96) x c [16 NC wn, + 16 NC)
    const int s16 = lane & 15, kg = lane >> 4;
    const int jrows = p.rows > 0 ? p.rows : cm.rows;         // (a row split of a phased launch has its own row count)
    const int nsteps = (jrows + DBK - 1) / DBK;
    // buffer descriptors by hand (raw buffer, 32-bit offsets checked against the byte size) for the DMA below, which is issued from inline assembly: through the
    // builtin the compiler orders every later LDS read behind it with s_waitcnt vmcnt(0) - a full memory round trip at the head of every step
    auto make_rsrc = [](const void* ptr, int64_t bytes) __attribute__((always_inline)) -> i32x4w {
        const uint64_t a = reinterpret_cast<uint64_t>(ptr);
        return i32x4w{(int)__builtin_amdgcn_readfirstlane((int)(uint32_t)a), (int)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32) & 0xFFFF),
                      (int)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
    };
    const i32x4w rdy = make_rsrc(p.dy, (int64_t)jrows * p.lddy * 2), rx = make_rsrc(p.x, (int64_t)jrows * p.ldx * 2);
    // ---- this wave's DMA slots per stage: slot j = instruction SLOTS wave + j (those past the last repeat it: same bytes to the same place, and every wave counts SLOTS).
    // Per-lane byte offsets inside the operand for step 0; + 64 rows per step ----
    // (branch-free: q is a wave-uniform run-time value.)  The last X instruction is pulled back to END at the tile's end - it overlaps its predecessor
    // with the same data - so that nothing is written into the next stage.
    uint32_t voff[DSLOTS];
    auto x_base = [](int q) __attribute__((always_inline)) -> int { const int b = (q - DNDY) * 1024; return b > DX_BYTES - 1024 ? DX_BYTES - 1024 : b; };
#pragma unroll
    for (int j = 0; j < DSLOTS; ++j) {
        int q = wave * DSLOTS + j;
        q = q >= DNDMA ? DNDMA - 1 : q;
        // padded rows; pad bytes come from beyond any operand, X rows in front of the tensor wrap past the descriptor's size (zeros both)
        const int Pd = q * 1024 + lane * 16, rowd = Pd / DYRS, wd = Pd % DYRS;
        const uint32_t vd = wd < DBMO * 2 ? (uint32_t)((rowd * p.lddy + o0) * 2 + wd) : 0x80000000u;
        const int P = x_base(q) + lane * 16, rowx = P / DXRS, w = P % DXRS;
        const uint32_t vx = w < DBNC * 2 ? (uint32_t)(((rowx - cm.pad) * p.ldx + c0) * 2 + w) : 0x80000000u;
        voff[j] = q < DNDY ? vd : vx;
    }
    const uint32_t stepdy = (uint32_t)(DBK * p.lddy * 2), stepx = (uint32_t)(DBK * p.ldx * 2);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(void __attribute__((address_space(3)))*)dsm;
    auto issue = [&](int step) __attribute__((always_inline)) {        // (called for steps 0, 1, 2, ... in order: the offsets advance in place)
        const uint32_t st = lds0 + (uint32_t)((step % DSTAGES) * DSTAGE);
#pragma unroll
        for (int j = 0; j < DSLOTS; ++j) {
            int q = wave * DSLOTS + j;
            q = q >= DNDMA ? DNDMA - 1 : q;
            const bool isx = q >= DNDY;
            const uint32_t dst = __builtin_amdgcn_readfirstlane(st + (uint32_t)(isx ? DDY_BYTES + x_base(q) : q * 1024));
            const i32x4w rs = isx ? rx : rdy;
            // (m0 = the LDS destination of a `buffer_load ... lds`.  hipcc reserves m0 and warns about it in a clobber list ("clobber list contains reserved
            //  registers", 119 x per build until round 6); the clobber stays - it is what tells the compiler that the s_mov changes m0 - and the diagnostic is
            //  silenced for this one statement)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
            asm volatile("s_mov_b32 m0, %2\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(voff[j]), "s"(rs), "s"(dst) : "memory", "m0");
#pragma clang diagnostic pop
            voff[j] += isx ? stepx : stepdy;
        }
    };

    f32x4w acc[DMO][DNC][TAPS];
    float bsum[DMO];                                         // dbias partial of this lane: DY column 64 wm + 16 mo + s16, the rows of its k group
#pragma unroll
    for (int mo = 0; mo < DMO; ++mo) {
        bsum[mo] = 0.f;
#pragma unroll
        for (int nc = 0; nc < DNC; ++nc)
#pragma unroll
            for (int t = 0; t < TAPS; ++t) acc[mo][nc][t] = f32x4w{0.f, 0.f, 0.f, 0.f};
    }
    const bool want_bias = (p.dbias != nullptr) && tile_c == 0 && wn == 0;

    auto tr = [](const unsigned char* ptr) __attribute__((always_inline)) -> s16x4 {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(ptr));
    };
    const int a_lane = (kg + 4 * (s16 >> 2)) * DYRS + wm * (DMO * 32) + (s16 & 3) * 8;            // + (32 k + 16 h) * 416 + mo * 32
    const int b_lane = DDY_BYTES + (kg + 4 * (s16 >> 2)) * DXRS + wn * (DNC * 32) + (s16 & 3) * 8;   // + (32 k + 16 h + t) * XRS + nc * 32

    // ---- one software pipeline over ALL (step, k block, tap) iterations.  With one wave per SIMD nobody else fills a wave's stalls, so nothing may wait for
    // what was just asked for: the B fragments are read PF iterations ahead (a ring of RING sets; NIT is a multiple of RING, so the slot of an iteration is
    // static across steps), the A fragments of the next k block replace the current ones as soon as their last MFMA has been issued, and both run across
    // the step boundary: the step's only barrier sits PF iterations before its end - there every wave's DMAs of step s + 1 have landed (with four stages those of
    // s + 2 may fly) and every wave has left step s - 1, whose stage the DMAs of step s + STAGES - 1 are then issued into.  Every step issues its DSLOTS DMAs, past the end
    // too (rows beyond the operands read as zeros into a stage nobody reads any more), so the wait count is a constant; the reads that run ahead of the
    // last step fetch stale bytes that no MFMA consumes. ----
    constexpr int NIT = (DBK / 32) * TAPS, RING = TAPS == 1 ? 2 : TAPS, PF = TAPS == 1 ? 1 : 2;
    static_assert(NIT % RING == 0 && PF < RING && DSTAGES >= 3, "pipeline geometry");
    bf16x8 af[DMO], bfr[RING][DNC];
    auto readA1 = [&](const unsigned char* base, int k32, int mo) __attribute__((always_inline)) {
        s16x4 tmp[2] = {tr(base + (32 * k32) * DYRS + mo * 32), tr(base + (32 * k32 + 16) * DYRS + mo * 32)};
        af[mo] = *reinterpret_cast<bf16x8*>(tmp);
    };
    auto readB = [&](const unsigned char* base, int it) __attribute__((always_inline)) {
        const int k32 = it / TAPS, t = it % TAPS;
#pragma unroll
        for (int nc = 0; nc < DNC; ++nc) {
            s16x4 tmp[2] = {tr(base + (32 * k32 + t) * DXRS + nc * 32), tr(base + (32 * k32 + 16 + t) * DXRS + nc * 32)};
            bfr[it % RING][nc] = *reinterpret_cast<bf16x8*>(tmp);
        }
    };
#pragma unroll
    for (int i = 0; i < DSTAGES - 1; ++i) issue(i);
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"((DSTAGES - 2) * DSLOTS) : "memory");
#pragma unroll
    for (int mo = 0; mo < DMO; ++mo) readA1(dsm + a_lane, 0, mo);
#pragma unroll
    for (int i = 0; i < PF; ++i) readB(dsm + b_lane, i);
    for (int s = 0; s < nsteps; ++s) {
        const unsigned char* sa = dsm + (s % DSTAGES) * DSTAGE + a_lane;
        const unsigned char* sb = dsm + (s % DSTAGES) * DSTAGE + b_lane;
        const unsigned char* na = dsm + ((s + 1) % DSTAGES) * DSTAGE + a_lane;
        const unsigned char* nb = dsm + ((s + 1) % DSTAGES) * DSTAGE + b_lane;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int k32 = it / TAPS, t = it % TAPS;
            if (it == NIT - PF) {
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"((DSTAGES - 3) * DSLOTS) : "memory");
                issue(s + DSTAGES - 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (it + PF < NIT) readB(sb, it + PF);
            else               readB(nb, it + PF - NIT);
            __builtin_amdgcn_sched_barrier(0);
            if (want_bias && t == 0) {                       // (VALU, in the shadow of the MFMAs; one wave of four in a third of the tiles)
#pragma unroll
                for (int mo = 0; mo < DMO; ++mo) {
                    const uint32_t* w = reinterpret_cast<const uint32_t*>(&af[mo]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) bsum[mo] += __uint_as_float(w[e] << 16) + __uint_as_float(w[e] & 0xFFFF0000u);
                }
            }
#pragma unroll
            for (int nc = 0; nc < DNC; ++nc)
#pragma unroll
                for (int mo = 0; mo < DMO; ++mo) {
                    mfma_acc(acc[mo][nc][t], af[mo], bfr[it % RING][nc]);
                    if (nc == DNC - 1 && t == TAPS - 1) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (it + 1 < NIT) readA1(sa, k32 + 1, mo);
                        else              readA1(na, 0, mo);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // (the last MFMAs' results are read below by instructions the compiler does not know to depend on an MFMA)
    // ---- epilogue: dW[o][c][t]; accumulator element i of fragment (mo, nc): o column 64 wm + 16 mo + 4 kg + i, c = 16 NC wn + 16 nc + s16 ----
    if (want_bias) {                                         // the four k groups of a column: lanes s16, s16 + 16, + 32, + 48
#pragma unroll
        for (int mo = 0; mo < DMO; ++mo) {
            float v = bsum[mo];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int pcol = o0 + wm * (16 * DMO) + mo * 16 + s16;
            int o = pcol;
            bool ok = pcol < p.m;
            if (p.perm == GLOWTTS_PERM_PAIR) {
                const int j = (pcol >> 6) * 32 + (pcol & 31);
                ok = ok && j < p.perm_h;
                o = ((pcol >> 5) & 1) * p.perm_h + j;
            }
            if (ok && kg == 0) p.dbias[o] = v;
        }
    }
#pragma unroll
    for (int mo = 0; mo < DMO; ++mo)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pcol = o0 + wm * (16 * DMO) + mo * 16 + 4 * kg + i;            // DY column (possibly PAIR-packed)
            int o = pcol;
            bool ok = pcol < p.m;
            if (p.perm == GLOWTTS_PERM_PAIR) {
                const int j = (pcol >> 6) * 32 + (pcol & 31);
                ok = ok && j < p.perm_h;
                o = ((pcol >> 5) & 1) * p.perm_h + j;
            }
            if (!ok) continue;
#pragma unroll
            for (int nc = 0; nc < DNC; ++nc) {
                const int c = c0 + wn * (DNC * 16) + nc * 16 + s16;
                if (c < p.ca) {
#pragma unroll
                    for (int t = 0; t < TAPS; ++t) p.dw[((long)o * p.ca + c) * TAPS + t] = acc[mo][nc][t][i];
                }
            }
        }
}

// the big decoder group qualifies when every job does: checked on the host by the caller (glowtts_wgrad_grouped_io, WIO_DMA)
template <int TAPS>
int launch_wgrad_dma(const glowtts_wgrad_job* table, const WCommon& cm, dim3 grid, hipStream_t s)
{
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_dma_kernel<TAPS>), hipFuncAttributeMaxDynamicSharedMemorySize, DCfg<TAPS>::STAGES * DCfg<TAPS>::STAGE) != hipSuccess) return GLOWTTS_E_LAUNCH;
        attr_done = true;
    }
    GLOWTTS_NOTE_STATIC("wgrad_dma<%d>/grouped", TAPS);
    hipLaunchKernelGGL((wgrad_dma_kernel<TAPS>), grid, dim3(DNT), DCfg<TAPS>::STAGES * DCfg<TAPS>::STAGE, s, table, cm);
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

template <typename CT, int XPRO, bool DYBF, bool XBF, bool WIDE = false>
int launch_x(const glowtts_wgrad_job& one, const glowtts_wgrad_job* table, const WCommon& cm, int taps, dim3 grid, hipStream_t s)
{
    GLOWTTS_NOTE("wgrad<%d,%s,dy%s,x%s%s%s>%s", taps, sizeof(CT) == 2 ? "bf16" : "f32", DYBF ? "bf16" : "f32", XBF ? "bf16" : "f32",
                 XPRO == GLOWTTS_APRO_PAIRMUL ? ",pairmul" : "", WIDE ? ",wide" : "", table ? "/grouped" : "");
    switch (taps) {
        case 1: hipLaunchKernelGGL((wgrad_kernel<CT, 1, XPRO, DYBF, XBF, WIDE>), grid, dim3(NT), 0, s, one, table, cm); break;
        case 3: hipLaunchKernelGGL((wgrad_kernel<CT, 3, XPRO, DYBF, XBF, WIDE>), grid, dim3(NT), 0, s, one, table, cm); break;
        case 5: hipLaunchKernelGGL((wgrad_kernel<CT, 5, XPRO, DYBF, XBF, WIDE>), grid, dim3(NT), 0, s, one, table, cm); break;
        default: return GLOWTTS_E_ARG;
    }
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}
// all jobs of one launch share the X prologue (`xpro`) and the storage types (`io`): grouped launches are formed per class.
// bf16 storage combinations the Glow-TTS path uses: (DY, X) both bf16 with no prologue (In conv: gate gradients x WaveNet state)
// and X alone, with PAIRMUL (Res_Skip conv: fp32 gradients x gates) or without (x stored tanh * sigmoid); DY alone without a prologue.
template <typename CT>
int launch_w(const glowtts_wgrad_job& one, const glowtts_wgrad_job* table, const WCommon& cm, int taps, int xpro, int io, dim3 grid, hipStream_t s)
{
    if constexpr (sizeof(CT) == 2) {
        if (io == (GLOWTTS_WIO_DY_BF16 | GLOWTTS_WIO_X_BF16 | GLOWTTS_WIO_WIDE) && xpro == GLOWTTS_APRO_NONE)
            return launch_x<CT, GLOWTTS_APRO_NONE, true, true, true>(one, table, cm, taps, grid, s);
        if (io == GLOWTTS_WIO_WIDE && xpro == GLOWTTS_APRO_NONE) return launch_x<CT, GLOWTTS_APRO_NONE, false, false, true>(one, table, cm, taps, grid, s);
        if (io & GLOWTTS_WIO_WIDE) return GLOWTTS_E_ARG;
        if (io == (GLOWTTS_WIO_DY_BF16 | GLOWTTS_WIO_X_BF16) && xpro == GLOWTTS_APRO_NONE) return launch_x<CT, GLOWTTS_APRO_NONE, true, true>(one, table, cm, taps, grid, s);
        if (io == GLOWTTS_WIO_X_BF16 && xpro == GLOWTTS_APRO_PAIRMUL) return launch_x<CT, GLOWTTS_APRO_PAIRMUL, false, true>(one, table, cm, taps, grid, s);
        if (io == GLOWTTS_WIO_X_BF16 && xpro == GLOWTTS_APRO_NONE) return launch_x<CT, GLOWTTS_APRO_NONE, false, true>(one, table, cm, taps, grid, s);
        // DY alone (round 5): the encoder's projection conv - bf16 gate gradients x the attention core's fp32 output rows
        if (io == GLOWTTS_WIO_DY_BF16 && xpro == GLOWTTS_APRO_NONE) return launch_x<CT, GLOWTTS_APRO_NONE, true, false>(one, table, cm, taps, grid, s);
    }
    if (io) return GLOWTTS_E_ARG;
    if (xpro == GLOWTTS_APRO_PAIRMUL) return launch_x<CT, GLOWTTS_APRO_PAIRMUL, false, false>(one, table, cm, taps, grid, s);
    if (xpro == GLOWTTS_APRO_NONE) return launch_x<CT, GLOWTTS_APRO_NONE, false, false>(one, table, cm, taps, grid, s);
    return GLOWTTS_E_ARG;
}

}  // namespace

static int xcd_mode() { return GLOWTTS_TUNABLE("GLOWTTS_WGRAD_XCD", 1); }

extern "C" int glowtts_wgrad_cl(const glowtts_wgrad_args* args, void* stream)
{
    if (!args || !args->dy || !args->x || !args->dw || args->rows < 1 || args->m < 1 || args->ca < 1) return GLOWTTS_E_ARG;
    if ((args->lddy & 3) || (args->ldx & 3) || (reinterpret_cast<uintptr_t>(args->dy) & 15) || (reinterpret_cast<uintptr_t>(args->x) & 15)) return GLOWTTS_E_ARG;
    const glowtts_wgrad_args& a = *args;
    glowtts_wgrad_job j;
    j.dy = a.dy; j.x = a.x; j.xmask = a.xmask; j.dw = a.dw; j.dbias = a.dbias; j.lddy = a.lddy; j.ldx = a.ldx;
    j.m = a.m; j.ca = a.ca; j.xpro = a.xpro; j.perm = a.perm; j.perm_h = a.perm_h; j.tile0 = 0;
    j.mt = (a.m + BMO - 1) / BMO; j.nt = (a.ca + BNC - 1) / BNC; j.reserved = 0;
    const int tiles = j.mt * j.nt;
    int splits = a.splits;
    if (splits < 1) {                                   // fill ~2 workgroups per CU, at least 256 rows per split
        splits = (512 + tiles - 1) / tiles;
        const int maxs = (a.rows + 255) / 256;
        if (splits > maxs) splits = maxs;
        if (splits < 1) splits = 1;
    }
    WCommon cm{a.rows, a.pad, a.accumulate, 1, xcd_mode()};
    dim3 grid(tiles, 1, splits);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (a.xmask) return GLOWTTS_E_ARG;          // reserved: X row masks must be applied by the producer
    if ((a.m & 3) || (a.ca & 3) || a.lddy < 4 || a.ldx < (a.xpro == GLOWTTS_APRO_PAIRMUL ? 8 : 4)) return GLOWTTS_E_ARG;
    if (a.precision == GLOWTTS_BF16) return launch_w<__bf16>(j, nullptr, cm, a.taps, a.xpro, a.io_flags, grid, s);
    if (a.precision == GLOWTTS_F32) return launch_w<float>(j, nullptr, cm, a.taps, a.xpro, a.io_flags, grid, s);
    return GLOWTTS_E_ARG;
}

extern "C" int glowtts_wgrad_grouped_phased(const glowtts_wgrad_job* dev_jobs, int njobs, int total_tiles, int whole_tiles, int rows, int taps, void* stream)
{
    if (!dev_jobs || njobs < 1 || total_tiles < 1 || whole_tiles < 0 || whole_tiles > total_tiles || rows < 1) return GLOWTTS_E_ARG;
    WCommon cmd{rows, (taps - 1) / 2, 0, njobs, xcd_mode()};
    cmd.phase = (whole_tiles > 0 && whole_tiles < total_tiles) ? whole_tiles : 0;
    hipStream_t sd = static_cast<hipStream_t>(stream);
    switch (taps) {
        case 1: return launch_wgrad_dma<1>(dev_jobs, cmd, dim3(total_tiles), sd);
        case 3: return launch_wgrad_dma<3>(dev_jobs, cmd, dim3(total_tiles), sd);
        case 5: return launch_wgrad_dma<5>(dev_jobs, cmd, dim3(total_tiles), sd);
        default: return GLOWTTS_E_ARG;
    }
}

extern "C" int glowtts_wgrad_grouped(const glowtts_wgrad_job* dev_jobs, int njobs, int total_tiles, int rows, int taps, int pad,
                                     int xpro, int precision, int splits, int accumulate, void* stream)
{
    return glowtts_wgrad_grouped_io(dev_jobs, njobs, total_tiles, rows, taps, pad, xpro, precision, splits, accumulate, 0, stream);
}

extern "C" int glowtts_wgrad_grouped_io(const glowtts_wgrad_job* dev_jobs, int njobs, int total_tiles, int rows, int taps, int pad,
                                        int xpro, int precision, int splits, int accumulate, int io_flags, void* stream)
{
    if (!dev_jobs || njobs < 1 || total_tiles < 1 || rows < 1) return GLOWTTS_E_ARG;
    if (splits < 1) splits = 1;
    if (io_flags & GLOWTTS_WIO_DMA) {
        // the caller promises: bf16 precision, both operands bf16 rows, no prologue, (mt, nt) of every job counted in 192 x 64 tiles (192 x 192 at one tap), 16-byte
        // aligned operands and row strides, operands below 2 GiB; no split-K, no accumulation
        if (precision != GLOWTTS_BF16 || xpro != GLOWTTS_APRO_NONE || splits != 1 || accumulate || pad != (taps - 1) / 2 ||
            (io_flags & (GLOWTTS_WIO_DY_BF16 | GLOWTTS_WIO_X_BF16)) != (GLOWTTS_WIO_DY_BF16 | GLOWTTS_WIO_X_BF16)) return GLOWTTS_E_ARG;
        WCommon cmd{rows, pad, 0, njobs, xcd_mode()};
        hipStream_t sd = static_cast<hipStream_t>(stream);
        switch (taps) {
            case 1: return launch_wgrad_dma<1>(dev_jobs, cmd, dim3(total_tiles), sd);
            case 3: return launch_wgrad_dma<3>(dev_jobs, cmd, dim3(total_tiles), sd);
            case 5: return launch_wgrad_dma<5>(dev_jobs, cmd, dim3(total_tiles), sd);
            default: return GLOWTTS_E_ARG;
        }
    }
    glowtts_wgrad_job dummy;
    memset(&dummy, 0, sizeof(dummy));
    dummy.mt = 1; dummy.nt = 1;
    WCommon cm{rows, pad, accumulate, njobs, xcd_mode()};
    dim3 grid(total_tiles, 1, splits);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (precision == GLOWTTS_BF16) return launch_w<__bf16>(dummy, dev_jobs, cm, taps, xpro, io_flags, grid, s);
    if (precision == GLOWTTS_F32) return launch_w<float>(dummy, dev_jobs, cm, taps, xpro, io_flags, grid, s);
    return GLOWTTS_E_ARG;
}

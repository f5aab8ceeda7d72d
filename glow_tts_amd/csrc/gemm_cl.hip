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
#include <stdint.h>
#include "../../include/glowtts_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef uint32_t Chunk16 __attribute__((ext_vector_type(4)));      // one 16-byte LDS slot (native vector: stays in VGPRs)

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return *reinterpret_cast<uint32_t*>(&v);
}

__device__ __forceinline__ int swz(int row, int q) { return row * 64 + ((q ^ ((row >> 2) & 3)) << 4); }

// EXACT = f32 mode: libm-grade transcendental functions; bf16 mode: hardware exp (v_exp_f32)
template <bool EXACT> __device__ __forceinline__ float exp_(float x) { return EXACT ? expf(x) : __expf(x); }
template <bool EXACT> __device__ __forceinline__ float sigmoid_(float x) { return 1.f / (1.f + exp_<EXACT>(-x)); }
template <bool EXACT> __device__ __forceinline__ float tanh_(float x) {
    if (EXACT) return tanhf(x);
    const float e = __expf(2.f * x);          // tanh(x) = 1 - 2 / (exp(2x) + 1)
    return 1.f - 2.f / (e + 1.f);
}

// ------------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------------
template <typename CT>
__global__ void pack_weight_kernel(const float* __restrict__ w, CT* __restrict__ out, int O, int I, int taps,
                                   int transpose, int perm, int perm_h, int N, int K, int npad, int kchunks)
{
    constexpr int KC = 64 / sizeof(CT);
    const long total = (long)taps * kchunks * npad * KC;
    w += (long)blockIdx.y * O * I * taps;          // batch of independent weights
    out += (long)blockIdx.y * total;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int kk = i % KC;
        long r = i / KC;
        const int n = r % npad; r /= npad;
        const int kc = r % kchunks;
        const int t = r / kchunks;
        const int k = kc * KC + kk;
        float v = 0.f;
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
        if (ok) v = w[((long)o * I + c_idx) * taps + (transpose ? (taps - 1 - t) : t)];
        out[i] = (CT)v;
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

// TAPS and APRO are compile-time so that the main loop is straight-line code with a FIXED number of global loads per
// step: the compiler's s_waitcnt bookkeeping then emits counted waits (vmcnt(N), N > 0) and the loads issued for step
// s+2 stay in flight across the MFMAs and the barrier of step s.  (With step-dependent `if (s + 2 < S) load` the counts
// become path dependent and every wait degenerates to vmcnt(0): measured 1700 cycles per 256-cycle MFMA step.)
// For the same reason every load is unconditional with a clamped address and is kept RAW in registers; validity
// masks, the fp32->bf16 conversion and the A prologue are applied when the registers are written to LDS two steps later.
template <typename CT, int MI, int NI, int WM, int WN, int EPI, int TAPS, int APRO>
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
    constexpr int AROWS = BM + TAPS - 1;
    constexpr int KC = Prec<CT>::KC, E = Prec<CT>::E;          // channels per 64-B chunk / per 16-B slot
    constexpr bool EX = sizeof(CT) == 4;
    constexpr int A_IT = (AROWS * 4 + NT - 1) / NT;
    constexpr int W_IT = (BN * 4) / NT;
    constexpr int NLD = (APRO == GLOWTTS_APRO_PAIRMUL) ? E / 2 : E / 4;      // float4 loads per 16-B LDS slot
    constexpr bool T1 = (TAPS == 1);
    static_assert((BN * 4) % NT == 0, "weight tile must divide evenly");

    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * AROWS * 64 + 2 * BN * 64];
    unsigned char* As = smem;                         // [2][AROWS][64]
    unsigned char* Ws = smem + 2 * AROWS * 64;        // [2][BN][64]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int KCH = p.kchunks;
    const int S = KCH * TAPS;
    const int pad = (TAPS - 1) / 2;
    const int l31 = lane & 31, lhi = lane >> 5;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    typedef float4 ARegs[A_IT][NLD];
    typedef Chunk16 WRegs[W_IT];
    ARegs raA, raB;
    WRegs rwA, rwB;

    // ---- global -> registers (raw, unconditional, clamped) ----
    auto gload_a = [&](ARegs& ra, int kc) __attribute__((always_inline)) {
        kc = min(kc, KCH - 1);
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int idx = tid + it * NT;
            const int row = idx >> 2, q = idx & 3;
            long g = (long)m0 - pad + row;
            g = g < 0 ? 0 : (g >= p.rows ? p.rows - 1 : g);
            const int c = kc * KC + q * E;
            const float* src;
            if (APRO == GLOWTTS_APRO_PAIRMUL)      src = p.a + g * p.lda + min(2 * c, (int)p.lda - 2 * E);
            else if (APRO == GLOWTTS_APRO_SQNEG)   src = p.a + g * p.lda + min(c < p.ca1 ? c : c - p.ca1, (int)p.lda - E);
            else {
                const bool second = (p.a2 != nullptr) && (c >= p.ca1);
                const float* base = second ? p.a2 : p.a;
                const long ld = second ? p.lda2 : p.lda;
                src = base + g * ld + min(second ? c - p.ca1 : c, (int)ld - E);
            }
#pragma unroll
            for (int j = 0; j < NLD; ++j) ra[it][j] = *reinterpret_cast<const float4*>(src + 4 * j);
        }
    };
    auto gload_w = [&](WRegs& rw, int s) __attribute__((always_inline)) {
        s = min(s, S - 1);
        const int kc = s / TAPS, t = s - kc * TAPS;
        const unsigned char* base = reinterpret_cast<const unsigned char*>(p.w) + ((long)(t * KCH + kc) * p.npad + n0) * 64;
        const int lim = (p.npad - n0) * 64 - 16;               // last valid 16-B piece of this tile's slab
#pragma unroll
        for (int it = 0; it < W_IT; ++it) rw[it] = *reinterpret_cast<const Chunk16*>(base + min((tid + it * NT) * 16, lim));
    };
    // ---- registers -> LDS (mask, prologue, convert) ----
    auto sstore_a = [&](const ARegs& ra, int buf, int kc) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int idx = tid + it * NT;
            const int row = idx >> 2, q = idx & 3;
            if (row >= AROWS) continue;
            const long g = (long)m0 - pad + row;
            const int c = kc * KC + q * E;
            const bool rowok = (g >= 0) && (g < p.rows) && (kc < KCH);
            float f[E];
            if (APRO == GLOWTTS_APRO_PAIRMUL) {
#pragma unroll
                for (int j = 0; j < NLD; ++j) { f[2 * j] = ra[it][j].x * ra[it][j].y; f[2 * j + 1] = ra[it][j].z * ra[it][j].w; }
            } else {
#pragma unroll
                for (int j = 0; j < NLD; ++j) { f[4 * j] = ra[it][j].x; f[4 * j + 1] = ra[it][j].y; f[4 * j + 2] = ra[it][j].z; f[4 * j + 3] = ra[it][j].w; }
                if (APRO == GLOWTTS_APRO_SQNEG) {
                    if (c < p.ca1) {
#pragma unroll
                        for (int e = 0; e < E; ++e) f[e] = -0.5f * f[e] * f[e];
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < E; ++e) f[e] = (rowok && (c + e < p.ca)) ? f[e] : 0.f;
            Chunk16 o;
            if constexpr (sizeof(CT) == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(f[2 * e], f[2 * e + 1]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = __float_as_uint(f[e]);
            }
            *reinterpret_cast<Chunk16*>(As + buf * (AROWS * 64) + swz(row, q)) = o;
        }
    };
    auto sstore_w = [&](const WRegs& rw, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int idx = tid + it * NT;
            *reinterpret_cast<Chunk16*>(Ws + buf * (BN * 64) + swz(idx >> 2, idx & 3)) = rw[it];
        }
    };

    // ---- MFMA over one (chunk, tap) step ----
    auto compute = [&](int abuf, int wbuf, int tap) __attribute__((always_inline)) {
        const unsigned char* Ab = As + abuf * (AROWS * 64);
        const unsigned char* Wb = Ws + wbuf * (BN * 64);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int q = 2 * s2 + lhi;
            Chunk16 af[MI], bfr[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                af[mi] = *reinterpret_cast<const Chunk16*>(Ab + swz((wm * MI + mi) * 32 + l31 + tap, q));
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

    // ---- main loop: two K chunks (2*TAPS steps) per iteration so that every buffer / register-set index is static ----
    gload_a(raA, 0);
    gload_w(rwA, 0);
    gload_w(rwB, 1);
    if constexpr (T1) gload_a(raB, 1);
    sstore_a(raA, 0, 0);
    sstore_w(rwA, 0);
    __syncthreads();
    auto step = [&](WRegs& wl, const WRegs& ws, ARegs& al, const ARegs& as, int half, int par, int tap, int kc) __attribute__((always_inline)) {
        const int s = kc * TAPS + tap;
        gload_w(wl, s + 2);                                   // W(s+2) goes into the set that held W(s)
        if constexpr (T1) {
            gload_a(al, kc + 2);
            if (kc < KCH) compute(half, par, 0);
            sstore_w(ws, par ^ 1);                            // W(s+1), loaded during step s-1
            sstore_a(as, half ^ 1, kc + 1);
        } else {
            if (tap == 0) gload_a(al, kc + 1);
            if (kc < KCH) compute(half, par, tap);
            sstore_w(ws, par ^ 1);
            if (tap == TAPS - 1) sstore_a(al, half ^ 1, kc + 1);
        }
        __syncthreads();
    };
    // explicit compile-time unrolling (u = step inside the two-chunk group); see STEP_U below
#define STEP_U(U)                                                                                              \
    if constexpr ((U) < 2 * TAPS) {                                                                            \
        constexpr int half_ = (U) / TAPS, tap_ = (U) % TAPS;                                                   \
        if constexpr (T1) {                                                                                    \
            if constexpr ((U) & 1) step(rwB, rwA, raB, raA, half_, 1, tap_, kc2 + half_);                      \
            else                   step(rwA, rwB, raA, raB, half_, 0, tap_, kc2 + half_);                      \
        } else {                                                                                               \
            if constexpr ((U) & 1) step(rwB, rwA, raA, raA, half_, 1, tap_, kc2 + half_);                      \
            else                   step(rwA, rwB, raA, raA, half_, 0, tap_, kc2 + half_);                      \
        }                                                                                                      \
    }
    for (int kc2 = 0; kc2 < KCH; kc2 += 2) {
        STEP_U(0) STEP_U(1) STEP_U(2) STEP_U(3) STEP_U(4) STEP_U(5) STEP_U(6) STEP_U(7) STEP_U(8) STEP_U(9)
    }
#undef STEP_U
    static_assert(2 * TAPS <= 10, "extend the STEP_U list");

    // ---- fused epilogue ----
    // accumulator element: row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), col = lane & 31
    const int fl = p.flags;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int r = m0 + (wm * MI + mi) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lhi;
            if (r >= p.rows) continue;
            const float mask = p.rowmask ? p.rowmask[r] : 1.f;
            if constexpr (EPI == GLOWTTS_EPI_LINEAR) {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int n = n0 + (wn * NI + ni) * 32 + l31;
                    if (n >= p.n) continue;
                    float v = acc[mi][ni][reg];
                    if (fl & GLOWTTS_F_BIAS) v += p.bias[n];
                    if (fl & GLOWTTS_F_RELU) v = fmaxf(v, 0.f);
                    if (fl & GLOWTTS_F_ADD_IN0) v += p.in0[(long)r * p.ldi0 + n];
                    if (fl & GLOWTTS_F_MASK) v *= mask;
                    if ((fl & GLOWTTS_F_COLMASK) && n >= p.ncols_valid[blockIdx.z]) v = 0.f;
                    float* o = p.out0 + (long)r * p.ld0 + n;
                    if (fl & GLOWTTS_F_ACCUM) v += *o;
                    *o = v;
                }
            } else if constexpr (EPI == GLOWTTS_EPI_RESSKIP) {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int n = n0 + (wn * NI + ni) * 32 + l31;
                    if (n >= p.n) continue;
                    const float v = acc[mi][ni][reg] + p.bias[n];
                    if (fl & GLOWTTS_F_LAST) {                      // Modules.py:880-883: output += res_skips; return output * mask
                        float* o = p.out1 + (long)r * p.ld1 + n;
                        *o = (((fl & GLOWTTS_F_FIRST) ? 0.f : *o) + v) * mask;
                    } else if (n < p.h) {                           // Modules.py:878: x = (x + res) * mask
                        p.out0[(long)r * p.ld0 + n] = (p.in0[(long)r * p.ldi0 + n] + v) * mask;
                    } else {                                        // Modules.py:879: output += outs
                        float* o = p.out1 + (long)r * p.ld1 + (n - p.h);
                        *o = ((fl & GLOWTTS_F_FIRST) ? 0.f : *o) + v;
                    }
                }
            } else {
                // PAIR-packed columns: fragment 2*pi holds the first half, 2*pi+1 the second half of 32 channels
                static_assert(EPI == GLOWTTS_EPI_LINEAR || EPI == GLOWTTS_EPI_RESSKIP || NI % 2 == 0 || EPI == GLOWTTS_EPI_DGATE, "pair epilogues need NI even");
                if constexpr (EPI == GLOWTTS_EPI_GATE || EPI == GLOWTTS_EPI_COUPLE) {
#pragma unroll
                    for (int pi = 0; pi < NI / 2; ++pi) {
                        const int pcol = n0 + (wn * NI + 2 * pi) * 32;           // packed column of the first half
                        const int j = (pcol >> 6) * 32 + l31;                     // channel inside a half
                        if (j >= p.h) continue;
                        float v0 = acc[mi][2 * pi][reg] + p.bias[j];
                        float v1 = acc[mi][2 * pi + 1][reg] + p.bias[p.h + j];
                        if constexpr (EPI == GLOWTTS_EPI_GATE) {
                            if (p.cond) {                                          // Modules.py:863-866 (added after the conv)
                                const float* cb = p.cond + (long)(r / p.rows_per_utt) * p.ldcond;
                                v0 += cb[j];
                                v1 += cb[p.h + j];
                            }
                            float2 g = make_float2(tanh_<EX>(v0), sigmoid_<EX>(v1));  // Modules.py:885-887
                            *reinterpret_cast<float2*>(p.out0 + (long)r * p.ld0 + 2 * j) = g;
                        } else {
                            // v0 = m, v1 = logs                                     Modules.py:795-806
                            float* xb = p.out0 + (long)r * p.ld0 + j;
                            const float x = p.in0 ? p.in0[(long)r * p.ldi0 + j] : *xb;     // x_b read from the kept coupling input when given
                            if (fl & GLOWTTS_F_REVERSE) *xb = (x - v0) * exp_<EX>(-v1) * mask;
                            else                        *xb = (v0 + exp_<EX>(v1) * x) * mask;
                            if (p.out1) {
                                p.out1[(long)r * p.ld1 + pcol + l31] = v0;
                                p.out1[(long)r * p.ld1 + pcol + 32 + l31] = v1;
                            }
                        }
                    }
                } else if constexpr (EPI == GLOWTTS_EPI_DGATE) {
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        const int j = n0 + (wn * NI + ni) * 32 + l31;             // gate channel (natural order)
                        if (j >= p.n) continue;
                        const float d = acc[mi][ni][reg];
                        const float2 g = *reinterpret_cast<const float2*>(p.in0 + (long)r * p.ldi0 + 2 * j);   // (tanh, sigmoid)
                        const float da = d * g.y * (1.f - g.x * g.x);
                        const float ds = d * g.x * g.y * (1.f - g.y);
                        const int pc = (j >> 5) * 64 + (j & 31);
                        p.out0[(long)r * p.ld0 + pc] = da;
                        p.out0[(long)r * p.ld0 + pc + 32] = ds;
                    }
                }
            }
        }
    }
}

template <typename CT, int MI, int NI, int WM, int WN, int EPI, int TAPS, int APRO>
int launch_k(const glowtts_conv_args& a, hipStream_t s)
{
    constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
    dim3 grid((a.rows + BM - 1) / BM, (a.npad + BN - 1) / BN, a.batch > 1 ? a.batch : 1);
    hipLaunchKernelGGL((conv_cl_kernel<CT, MI, NI, WM, WN, EPI, TAPS, APRO>), grid, dim3(WM * WN * 64), 0, s, a);
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

template <typename CT, int EPI, int TAPS, int APRO>
int launch_tile(const glowtts_conv_args& a, hipStream_t s)
{
    // tile choice: 128x128 by default; 64-row tiles when that is needed to put >= ~256 workgroups on the chip
    const long tiles128 = (long)((a.rows + 127) / 128) * ((a.npad + 127) / 128) * (a.batch > 1 ? a.batch : 1);
    if (tiles128 >= 256) return launch_k<CT, 2, 2, 2, 2, EPI, TAPS, APRO>(a, s);
    return launch_k<CT, 1, 2, 2, 2, EPI, TAPS, APRO>(a, s);
}

template <typename CT, int EPI, int APRO>
int launch_taps(const glowtts_conv_args& a, hipStream_t s)
{
    switch (a.taps) {
        case 1: return launch_tile<CT, EPI, 1, APRO>(a, s);
        case 3: return launch_tile<CT, EPI, 3, APRO>(a, s);
        case 5: return launch_tile<CT, EPI, 5, APRO>(a, s);
        default: return GLOWTTS_E_ARG;
    }
}

// the (epilogue, prologue, taps) combinations the Glow-TTS path uses; anything else is rejected
template <typename CT>
int launch_prec(const glowtts_conv_args& a, hipStream_t s)
{
    const int N = GLOWTTS_APRO_NONE, PM = GLOWTTS_APRO_PAIRMUL;
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

extern "C" int glowtts_pack_weight_batched(const float* w, int batch, int O, int I, int taps, int transpose, int perm, int perm_h,
                                           int precision, void* packed, int* npad_out, int* kchunks_out, void* stream)
{
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
    if (precision == GLOWTTS_BF16)
        hipLaunchKernelGGL(pack_weight_kernel<__bf16>, dim3(blocks, batch), dim3(256), 0, s, w, static_cast<__bf16*>(packed), O, I, taps, transpose, perm, perm_h, N, kchunks * KC, npad, kchunks);
    else
        hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(blocks, batch), dim3(256), 0, s, w, static_cast<float*>(packed), O, I, taps, transpose, perm, perm_h, N, kchunks * KC, npad, kchunks);
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
    if ((args->lda & 3) || (reinterpret_cast<uintptr_t>(args->a) & 15)) return GLOWTTS_E_ARG;
    if (args->a2 && ((args->lda2 & 3) || (reinterpret_cast<uintptr_t>(args->a2) & 15))) return GLOWTTS_E_ARG;
    glowtts_conv_args a = *args;
    if (!a.a2 && a.apro != GLOWTTS_APRO_SQNEG) a.ca1 = a.ca;
    if (a.apro == GLOWTTS_APRO_SQNEG && ((a.ca1 % (a.precision == GLOWTTS_BF16 ? 8 : 4)) || a.ca != 2 * a.ca1)) return GLOWTTS_E_ARG;
    if ((a.flags & GLOWTTS_F_COLMASK) && !a.ncols_valid) return GLOWTTS_E_ARG;
    {   // loads are unconditional 16-byte vectors: rows must be wide enough for the last (possibly partial) K slot
        const int E = a.precision == GLOWTTS_BF16 ? 8 : 4;
        if ((a.ca & 3) || a.kchunks * (a.precision == GLOWTTS_BF16 ? 32 : 16) < a.ca) return GLOWTTS_E_ARG;
        if (a.apro == GLOWTTS_APRO_PAIRMUL) { if (a.lda < 2 * E) return GLOWTTS_E_ARG; }
        else if (a.lda < E || (a.a2 && a.lda2 < E)) return GLOWTTS_E_ARG;
        if (a.a2 && (a.ca1 % E)) return GLOWTTS_E_ARG;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (a.precision == GLOWTTS_BF16) return launch_prec<__bf16>(a, s);
    if (a.precision == GLOWTTS_F32) return launch_prec<float>(a, s);
    return GLOWTTS_E_ARG;
}

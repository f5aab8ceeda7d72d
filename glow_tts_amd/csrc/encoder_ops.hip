// Text-encoder kernels that are not convolutions (gfx950): LayerNorm over channels with its fused neighbours
// (Modules.py:472-489, 523-526, 541-544, 561-571), embedding lookup (Modules.py:242-250, 267), dropout/ReLU backward gating,
// and the relative-position multi-head self-attention core (RPR_MHA.py:95-128) forward and backward.
// Layout: rows tensors [B][Tp][C], channels contiguous, zero pad rows (include/glowtts_hip.h).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/glowtts_hip.h"
#include "tunable.h"
#include "launch_log.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float drop_scale(uint32_t seed, uint32_t id, float p, float inv_keep) {
    uint32_t h = id * 0x9E3779B1u + seed;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return ((h >> 8) * (1.0f / 16777216.0f) >= p) ? inv_keep : 0.f;
}

constexpr int LN_MAXK = 16;          // channels per lane: C <= 1024

// ------------------------------------------------------------------------------------------------
// y = rowmask * dropout( relu?( LayerNorm(a + b) * gamma + beta ) ),  one wavefront per row.
// Keeps s = a + b (when b is given) and (mean, rstd) per row for the backward.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint16_t f2bf(float v) { const __bf16 b = (__bf16)v; return *reinterpret_cast<const uint16_t*>(&b); }
__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }

// KT > 0: C == 64 * KT exactly (192- and 256-channel rows: every row of the encoder): no per-element predicates, and EVERY load of the row -
// a, b, gamma, beta, the row mask - is issued before the first use.  (The generic form below loads under `if (c < C)`: hipcc then branches
// around each load and waits for it before it issues the next - ten dependent round trips to L2 per row, 6.7 us per launch for 3 MB.)
template <int KT>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ s_out,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ rowmask, float* __restrict__ y, float* __restrict__ stats,
                                                     long rows, int C, float eps, int relu, float drop_p, uint32_t seed,
                                                     const uint32_t* __restrict__ seed_ptr, uint16_t* __restrict__ yb)
{
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    if constexpr (KT > 0) {
        constexpr int CC = 64 * KT;
        const float* ar = a + r * CC + lane;
        float v[KT], bv[KT], gm[KT], bt[KT];
#pragma unroll
        for (int k = 0; k < KT; ++k) v[k] = ar[64 * k];
        if (b) {
            const float* br = b + r * CC + lane;
#pragma unroll
            for (int k = 0; k < KT; ++k) bv[k] = br[64 * k];
        }
#pragma unroll
        for (int k = 0; k < KT; ++k) { gm[k] = gamma[lane + 64 * k]; bt[k] = beta[lane + 64 * k]; }
        const float m = rowmask ? rowmask[r] : 1.f;
        if (seed_ptr) seed += *seed_ptr;
        if (b) {
#pragma unroll
            for (int k = 0; k < KT; ++k) v[k] += bv[k];
        }
        if (s_out) {
#pragma unroll
            for (int k = 0; k < KT; ++k) s_out[r * CC + lane + 64 * k] = v[k];
        }
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < KT; ++k) sum += v[k];
        const float mean = wave_sum(sum) / CC;
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < KT; ++k) { const float d = v[k] - mean; sq += d * d; }
        const float rstd = rsqrtf(wave_sum(sq) / CC + eps);
        if (lane == 0) { stats[2 * r] = mean; stats[2 * r + 1] = rstd; }
        const float ik = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            const int c = lane + 64 * k;
            float o = (v[k] - mean) * rstd * gm[k] + bt[k];
            if (relu) o = fmaxf(o, 0.f);
            if (drop_p > 0.f) o *= drop_scale(seed, (uint32_t)(r * CC + c), drop_p, ik);
            y[r * CC + c] = o * m;
            if (yb) yb[r * CC + c] = f2bf(o * m);          // the same rows as an MFMA operand (LDS-DMA convs read raw bf16)
        }
        return;
    }
    if (seed_ptr) seed += *seed_ptr;
    const int K = (C + 63) / 64;
    float v[LN_MAXK];
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXK; ++k) {
        if (k >= K) break;
        const int c = lane + 64 * k;
        float x = 0.f;
        if (c < C) { x = a[r * C + c]; if (b) x += b[r * C + c]; if (s_out) s_out[r * C + c] = x; }
        v[k] = x; sum += x;
    }
    const float mean = wave_sum(sum) / C;
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXK; ++k) { if (k >= K) break; const int c = lane + 64 * k; const float d = (c < C) ? v[k] - mean : 0.f; sq += d * d; }
    const float rstd = rsqrtf(wave_sum(sq) / C + eps);
    if (lane == 0) { stats[2 * r] = mean; stats[2 * r + 1] = rstd; }
    const float m = rowmask ? rowmask[r] : 1.f;
    const float ik = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
#pragma unroll
    for (int k = 0; k < LN_MAXK; ++k) {
        if (k >= K) break;
        const int c = lane + 64 * k;
        if (c >= C) continue;
        float o = (v[k] - mean) * rstd * gamma[c] + beta[c];
        if (relu) o = fmaxf(o, 0.f);
        if (drop_p > 0.f) o *= drop_scale(seed, (uint32_t)(r * C + c), drop_p, ik);
        y[r * C + c] = o * m;
        if (yb) yb[r * C + c] = f2bf(o * m);          // the same rows as an MFMA operand (LDS-DMA convs read raw bf16)
    }
}

// backward: dz = dy * mask * gate(y) ; ds = rstd (g dz - mean(g dz) - xhat mean(g dz xhat)) ; partial dgamma / dbeta per block
// gate(y): relu and/or dropout -> (y != 0) * 1/(1-p)   (y is the forward output: zero exactly where relu / dropout / mask cut)
// KT: C / 64 for the exact widths 192 (encoder) and 256 (duration predictor), 0 = generic: the per-lane arrays then have exactly KT entries,
// there are no per-element predicates and every load of a row is issued before the first use.  ONE row per wavefront, 16 wavefronts = 16 rows per workgroup: every load of the pass is in flight at once (four waves walking
// four rows each in dependent round trips to HBM took 22.6 us for 12 MB, 6 % of HBM bandwidth; now 9) and the second stage still sums only
// rows / 16 partials per column.
constexpr int LN_BWD_WAVES = 16;
template <int KT>
__global__ __launch_bounds__(LN_BWD_WAVES * 64) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ s,
                                                     const float* __restrict__ stats, const float* __restrict__ gamma,
                                                     const float* __restrict__ rowmask, float* __restrict__ ds, float* __restrict__ partial,
                                                     long rows, int C, int gated, float drop_p, int rows_per_block, uint16_t* __restrict__ dsb,
                                                     const float* __restrict__ gate_out, float gate_scale)
{
    extern __shared__ float red[];                 // [LN_BWD_WAVES][2][C]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int KM = KT > 0 ? KT : LN_MAXK;
    const int K = KT > 0 ? KT : (C + 63) / 64;
    const float ik = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    float accg[KM], accb[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) { accg[k] = 0.f; accb[k] = 0.f; }
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(rows, r0 + rows_per_block);
    if constexpr (KT > 0) {
        // C == 64 * KT exactly (the launcher picks KT only then): no per-element predicates and every load of the row - statistics, mask,
        // dy, y, s, the conv's kept output - issued before the first use (under `if (c < C)` each was a dependent round trip to L2:
        // 12.8-14.9 us per launch inside the step)
        constexpr int CC = 64 * KT;
        float gm[KT];
#pragma unroll
        for (int k = 0; k < KT; ++k) gm[k] = gamma[lane + 64 * k];
        for (long r = r0 + wave; r < r1; r += LN_BWD_WAVES) {
            const long o = r * CC + lane;
            float dyv[KT], yv[KT], sv[KT], gv[KT];
            const float mean = stats[2 * r], rstd = stats[2 * r + 1];
            const float m = rowmask ? rowmask[r] : 1.f;
#pragma unroll
            for (int k = 0; k < KT; ++k) { dyv[k] = dy[o + 64 * k]; sv[k] = s[o + 64 * k]; }
            if (gated) {
#pragma unroll
                for (int k = 0; k < KT; ++k) yv[k] = y[o + 64 * k];
            }
            const bool gout = dsb && gate_out;
            if (gout) {
#pragma unroll
                for (int k = 0; k < KT; ++k) gv[k] = gate_out[o + 64 * k];
            }
            float dz[KT], xh[KT];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                float d = dyv[k] * m;
                if (gated) d = (yv[k] != 0.f) ? d * ik : 0.f;
                const float x = (sv[k] - mean) * rstd;
                accg[k] += d * x; accb[k] += d;
                d *= gm[k];
                dz[k] = d; xh[k] = x; s1 += d; s2 += d * x;
            }
            s1 = wave_sum(s1) / CC; s2 = wave_sum(s2) / CC;
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                const float g = rstd * (dz[k] - s1 - xh[k] * s2);
                ds[o + 64 * k] = g;
                if (dsb) dsb[o + 64 * k] = f2bf(gout ? (gv[k] != 0.f ? g * gate_scale : 0.f) : g);
            }
        }
    } else
    for (long r = r0 + wave; r < r1; r += LN_BWD_WAVES) {
        const float mean = stats[2 * r], rstd = stats[2 * r + 1];
        const float m = rowmask ? rowmask[r] : 1.f;
        float dz[KM], xh[KM];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            if (k >= K) break;
            const int c = lane + 64 * k;
            float d = 0.f, x = 0.f;
            if (c < C) {
                d = dy[r * C + c] * m;
                if (gated) d = (y[r * C + c] != 0.f) ? d * ik : 0.f;
                x = (s[r * C + c] - mean) * rstd;
                accg[k] += d * x; accb[k] += d;
                d *= gamma[c];
            }
            dz[k] = d; xh[k] = x; s1 += d; s2 += d * x;
        }
        s1 = wave_sum(s1) / C; s2 = wave_sum(s2) / C;
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            if (k >= K) break;
            const int c = lane + 64 * k;
            if (c < C) {
                const float g = rstd * (dz[k] - s1 - xh[k] * s2);
                ds[r * C + c] = g;
                // bf16 copy for the conv that produced the LayerNorm input: as is, or already through that conv's dropout / relu gate
                // (d(pre-activation) = ds * (out != 0 ? 1 / keep : 0): saves the separate gate pass over the rows)
                if (dsb) dsb[r * C + c] = f2bf(gate_out ? (gate_out[r * C + c] != 0.f ? g * gate_scale : 0.f) : g);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < KM; ++k) {
        if (k >= K) break;
        const int c = lane + 64 * k;
        if (c < C) { red[(wave * 2 + 0) * C + c] = accg[k]; red[(wave * 2 + 1) * C + c] = accb[k]; }
    }
    __syncthreads();
    float* out = partial + (long)blockIdx.x * 2 * C;
    for (int i = threadIdx.x; i < 2 * C; i += LN_BWD_WAVES * 64) {
        const int which = i / C, c = i - which * C;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < LN_BWD_WAVES; ++w) t += red[(w * 2 + which) * C + c];
        out[i] = t;
    }
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int nblk, int n)
{
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= n) return;
    float s = 0.f;
    for (int k = lane; k < nblk; k += 64) s += partial[(long)k * n + i];
    s = wave_sum(s);
    if (lane == 0) out[i] = s;
}

// dz = dy * (out != 0 ? scale : 0) * rowmask      (backward gate of relu and/or dropout, see ln_bwd_kernel)
template <bool DYB, bool OUTB, bool DZB>
__global__ __launch_bounds__(256) void gate_bwd_kernel(const void* __restrict__ dy_, const void* __restrict__ out_, const float* __restrict__ rowmask,
                                                       void* __restrict__ dz_, long rows, int C, float scale)
{
    // DYB / OUTB / DZB: that tensor is stored as bf16 (operands of the LDS-DMA convs and of the wide weight-gradient staging)
    const long total = rows * C / 4;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = (i * 4) / C;
        const float m = (rowmask ? rowmask[r] : 1.f) * scale;
        float4 d, o;
        if (DYB) { const ushort4 t = reinterpret_cast<const ushort4*>(dy_)[i]; d = make_float4(bf2f(t.x), bf2f(t.y), bf2f(t.z), bf2f(t.w)); }
        else d = reinterpret_cast<const float4*>(dy_)[i];
        if (!out_) o = make_float4(1.f, 1.f, 1.f, 1.f);       // (no gate: the row mask alone - the backward of a conv whose output is only masked)
        else if (OUTB) { const ushort4 t = reinterpret_cast<const ushort4*>(out_)[i]; o = make_float4(bf2f(t.x), bf2f(t.y), bf2f(t.z), bf2f(t.w)); }
        else o = reinterpret_cast<const float4*>(out_)[i];
        const float4 z = make_float4(o.x != 0.f ? d.x * m : 0.f, o.y != 0.f ? d.y * m : 0.f, o.z != 0.f ? d.z * m : 0.f, o.w != 0.f ? d.w * m : 0.f);
        if (DZB) { ushort4 t; t.x = f2bf(z.x); t.y = f2bf(z.y); t.z = f2bf(z.z); t.w = f2bf(z.w); reinterpret_cast<ushort4*>(dz_)[i] = t; }
        else reinterpret_cast<float4*>(dz_)[i] = z;
    }
}

// ------------------------------------------------------------------------------------------------
// embedding: rows[b][PAD + t][c] = E[tok[b][t]][c] * scale * mask ; pad rows zero.   (Modules.py:267)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* __restrict__ tok, const float* __restrict__ E, const float* __restrict__ rowmask,
                                                        float* __restrict__ rows, int B, int T, int C, float scale)
{
    const int Tp = T + 2 * GLOWTTS_ROW_PAD;
    const long total = (long)B * Tp * C;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / C; const int c = (int)(i - r * C);
        const int b = (int)(r / Tp), tp = (int)(r - (long)b * Tp) - GLOWTTS_ROW_PAD;
        float v = 0.f;
        if (tp >= 0 && tp < T) v = E[tok[(long)b * T + tp] * C + c] * scale * rowmask[r];
        rows[i] = v;
    }
}
// dE[v][c] = scale * sum over rows whose token is v of d[row][c] * mask     (deterministic: one block per token id)
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int64_t* __restrict__ tok, const float* __restrict__ d, const float* __restrict__ rowmask,
                                                        float* __restrict__ dE, int B, int T, int C, float scale)
{
    // one workgroup per token id: (1) every thread scans a contiguous slice of the token array and the matching rows are
    // compacted into LDS in position order (prefix sum over per-thread counts -> deterministic order), (2) threads walk channels.
    extern __shared__ int lst[];                               // [256 counts/offsets] + [B*T rows]
    int* cnt = lst;
    int* rowsl = lst + 256;
    const int v = blockIdx.x;
    const int Tp = T + 2 * GLOWTTS_ROW_PAD;
    const int N = B * T;
    const int per = (N + 255) / 256;
    const int lo = threadIdx.x * per, hi = min(N, lo + per);
    int c0 = 0;
    for (int i = lo; i < hi; ++i) c0 += (tok[i] == v);
    cnt[threadIdx.x] = c0;
    __syncthreads();
    if (threadIdx.x == 0) { int run = 0; for (int i = 0; i < 256; ++i) { const int t = cnt[i]; cnt[i] = run; run += t; } }
    __syncthreads();
    int pos = cnt[threadIdx.x];
    for (int i = lo; i < hi; ++i) if (tok[i] == v) { const int b = i / T, t = i - b * T; rowsl[pos++] = b * Tp + GLOWTTS_ROW_PAD + t; }
    __syncthreads();
    const int total = cnt[255] + ((255 * per < N) ? 0 : 0);
    // number of matches = offset of the last thread + its own count
    int n = 0;
    { const int lo2 = 255 * per, hi2 = min(N, lo2 + per); int c1 = 0; for (int i = lo2; i < hi2; ++i) c1 += (tok[i] == v); n = cnt[255] + c1; }
    (void)total;
    for (int c = threadIdx.x; c < C; c += 256) {
        float acc = 0.f;
        for (int k = 0; k < n; ++k) { const long r = rowsl[k]; acc += d[r * C + c] * rowmask[r]; }
        dE[(long)v * C + c] = acc * scale;
    }
}

// ------------------------------------------------------------------------------------------------
// Relative-position self-attention core (RPR_MHA.py:95-128), fp32, one workgroup per (utterance, head, 16-query tile).
//   scores_ij = (q_i . k_j + [|j-i| <= w] q_i . relK[j-i+w]) / sqrt(D) ; masked_fill(-1e4) ; softmax ; dropout
//   out_i     = sum_j P_ij v_j + sum_{|d| <= w} P_{i,i+d} relV[d+w]
// qkv rows: [B][Tp][3][H][D] (Q | K | V thirds of one fused 1x1 conv).  K and V of the (b, h) pair are staged in LDS.
// P (after dropout, the matrix both PV terms use) is kept for the backward: [B][H][Tp][Tp].
// ------------------------------------------------------------------------------------------------
constexpr int ATT_QT = 16;     // queries per workgroup (4 per wavefront)

// K and V are staged one after the other in the SAME LDS buffer (phase 1: scores / softmax with K, phase 2: P V with V), so a
// 200-token utterance (Tp = 204, D = 96: 79 KB per operand) fits the 160 KB LDS with room for a second workgroup.
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ qkv, const float* __restrict__ relk, const float* __restrict__ relv,
                                                       const float* __restrict__ rowmask, float* __restrict__ out, float* __restrict__ P,
                                                       int B, int Tp, int H, int D, int win, float drop_p, uint32_t seed, const uint32_t* __restrict__ seed_ptr)
{
    extern __shared__ float sm[];
    const int C = H * D, ld = 3 * C;
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * ATT_QT;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ldk = D + 1;                                   // padded: lanes walk keys, bank = (key * (D+1)) % 32
    float* KV = sm;                                          // [Tp][D+1]  K, then V
    float* Rr = KV + Tp * ldk;                               // [2w+1][D]  relK, then relV
    float* Qs = Rr + (2 * win + 1) * D;                      // [4 waves][D]
    float* Ps = Qs + 4 * D;                                  // [ATT_QT][Tp]
    if (seed_ptr) seed += *seed_ptr;
    const float* base = qkv + (long)b * Tp * ld + h * D;
    for (int i = threadIdx.x; i < Tp * D; i += 256) { const int j = i / D, d = i - j * D; KV[j * ldk + d] = base[(long)j * ld + C + d]; }
    for (int i = threadIdx.x; i < (2 * win + 1) * D; i += 256) Rr[i] = relk[i];
    __syncthreads();
    const float* rm = rowmask + (long)b * Tp;
    const float isd = rsqrtf((float)D);
    const float ik = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    for (int qi = wave; qi < ATT_QT; qi += 4) {              // phase 1: P rows
        const int i = q0 + qi;
        if (i >= Tp) break;                                  // wave-uniform
        float* q = Qs + wave * D;
        for (int d = lane; d < D; d += 64) q[d] = base[(long)i * ld + d];
        float sc[4];
        float mx = -3.0e38f;
        const float mi = rm[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = lane + 64 * k;
            float s = -3.0e38f;
            if (j < Tp) {
                float acc = 0.f;
                const float* kr = KV + j * ldk;
                const int dd = j - i;
                if (dd >= -win && dd <= win) { const float* rr = Rr + (dd + win) * D;
#pragma unroll 8
                                               for (int d = 0; d < D; ++d) acc += q[d] * (kr[d] + rr[d]); }
                else                         {
#pragma unroll 8
                                               for (int d = 0; d < D; ++d) acc += q[d] * kr[d]; }
                s = acc * isd;
                if (mi * rm[j] == 0.f) s = -1e4f;            // RPR_MHA.py:117
            }
            sc[k] = s; mx = fmaxf(mx, s);
        }
        mx = wave_max(mx);
        float den = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int j = lane + 64 * k; sc[k] = (j < Tp) ? __expf(sc[k] - mx) : 0.f; den += sc[k]; }
        den = 1.f / wave_sum(den);
        float* pr = Ps + qi * Tp;
        float* Pg = P + (((long)b * H + h) * Tp + i) * Tp;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = lane + 64 * k;
            if (j < Tp) {
                float p = sc[k] * den;
                if (drop_p > 0.f) p *= drop_scale(seed, (uint32_t)((((long)b * H + h) * Tp + i) * Tp + j), drop_p, ik);   // RPR_MHA.py:120
                pr[j] = p; Pg[j] = p;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < Tp * D; i += 256) { const int j = i / D, d = i - j * D; KV[j * ldk + d] = base[(long)j * ld + 2 * C + d]; }
    for (int i = threadIdx.x; i < (2 * win + 1) * D; i += 256) Rr[i] = relv[i];
    __syncthreads();
    for (int qi = wave; qi < ATT_QT; qi += 4) {              // phase 2: out_i[d] = sum_j P_ij V[j][d] + sum_dd P_{i,i+dd} relV[dd+w][d]
        const int i = q0 + qi;
        if (i >= Tp) break;
        const float* pr = Ps + qi * Tp;
        for (int d = lane; d < D; d += 64) {
            float acc = 0.f;
            for (int j = 0; j < Tp; ++j) acc += pr[j] * KV[j * ldk + d];
            for (int dd = -win; dd <= win; ++dd) { const int j = i + dd; if (j >= 0 && j < Tp) acc += pr[j] * Rr[(dd + win) * D + d]; }
            out[((long)b * Tp + i) * C + h * D + d] = acc;
        }
    }
}

// backward, pass A (per query row).  P holds the DROPPED matrix Pd = P0 * keep / (1 - p) that both P V terms used.
//   dPd_ij = dO_i . (v_j + [band] relV[j-i+w]) ; D_i = sum_j Pd_ij dPd_ij ; dS_ij = (Pd_ij dPd_ij - P0_ij D_i) / sqrt(D)
// (P0 dP0 = Pd dPd because dP0 = dPd keep/(1-p).)  P0 is recomputed from the scores when dropout is on.  Phase 1 uses V, phase 2 K.
__global__ __launch_bounds__(256) void attn_bwd_a_kernel(const float* __restrict__ qkv, const float* __restrict__ relk, const float* __restrict__ relv,
                                                         const float* __restrict__ rowmask, const float* __restrict__ P, const float* __restrict__ dout,
                                                         float* __restrict__ dS, float* __restrict__ dqkv,
                                                         int B, int Tp, int H, int D, int win, float drop_p)
{
    extern __shared__ float sm[];
    const int C = H * D, ld = 3 * C;
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * ATT_QT;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ldk = D + 1;
    float* KV = sm;                                          // [Tp][D+1]  V, then K
    float* Rr = KV + Tp * ldk;                               // [2w+1][D]  relV, then relK
    float* Qs = Rr + (2 * win + 1) * D;                      // [4][D]  dO_i (phase 1) / q_i (phase 2)
    float* Ds = Qs + 4 * D;                                  // [ATT_QT][Tp]  dPd rows, then dS rows
    const float* base = qkv + (long)b * Tp * ld + h * D;
    for (int i = threadIdx.x; i < Tp * D; i += 256) { const int j = i / D, d = i - j * D; KV[j * ldk + d] = base[(long)j * ld + 2 * C + d]; }
    for (int i = threadIdx.x; i < (2 * win + 1) * D; i += 256) Rr[i] = relv[i];
    __syncthreads();
    const float* rm = rowmask + (long)b * Tp;
    const float isd = rsqrtf((float)D);
    for (int qi = wave; qi < ATT_QT; qi += 4) {              // phase 1: dPd rows
        const int i = q0 + qi;
        if (i >= Tp) break;
        float* go = Qs + wave * D;
        for (int d = lane; d < D; d += 64) go[d] = dout[((long)b * Tp + i) * C + h * D + d];
        float* dr = Ds + qi * Tp;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = lane + 64 * k;
            if (j < Tp) {
                const float* vr = KV + j * ldk;
                const int dd = j - i;
                float a = 0.f;
                if (dd >= -win && dd <= win) { const float* rv = Rr + (dd + win) * D; for (int d = 0; d < D; ++d) a += go[d] * (vr[d] + rv[d]); }
                else                         { for (int d = 0; d < D; ++d) a += go[d] * vr[d]; }
                dr[j] = a;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < Tp * D; i += 256) { const int j = i / D, d = i - j * D; KV[j * ldk + d] = base[(long)j * ld + C + d]; }
    for (int i = threadIdx.x; i < (2 * win + 1) * D; i += 256) Rr[i] = relk[i];
    __syncthreads();
    for (int qi = wave; qi < ATT_QT; qi += 4) {              // phase 2: dS rows and dQ
        const int i = q0 + qi;
        if (i >= Tp) break;
        float* q = Qs + wave * D;
        for (int d = lane; d < D; d += 64) q[d] = base[(long)i * ld + d];
        const float mi = rm[i];
        const float* Pg = P + (((long)b * H + h) * Tp + i) * Tp;
        float* dr = Ds + qi * Tp;
        float pd[4], dpd[4], p0[4];
        float mx = -3.0e38f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = lane + 64 * k;
            pd[k] = 0.f; dpd[k] = 0.f; p0[k] = -3.0e38f;
            if (j < Tp) {
                pd[k] = Pg[j]; dpd[k] = dr[j];
                if (drop_p > 0.f) {
                    const float* kr = KV + j * ldk;
                    const int dd = j - i;
                    float s = 0.f;
                    if (dd >= -win && dd <= win) { const float* rk = Rr + (dd + win) * D; for (int d = 0; d < D; ++d) s += q[d] * (kr[d] + rk[d]); }
                    else                         { for (int d = 0; d < D; ++d) s += q[d] * kr[d]; }
                    s *= isd;
                    if (mi * rm[j] == 0.f) s = -1e4f;
                    p0[k] = s; mx = fmaxf(mx, s);
                }
            }
        }
        float Di = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) Di += pd[k] * dpd[k];
        Di = wave_sum(Di);
        if (drop_p > 0.f) {                                  // undropped probabilities from the scores (same arithmetic as the forward)
            mx = wave_max(mx);
            float den = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int j = lane + 64 * k; p0[k] = (j < Tp) ? __expf(p0[k] - mx) : 0.f; den += p0[k]; }
            den = 1.f / wave_sum(den);
#pragma unroll
            for (int k = 0; k < 4; ++k) p0[k] *= den;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) p0[k] = pd[k];
        }
        float* dSg = dS + (((long)b * H + h) * Tp + i) * Tp;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = lane + 64 * k;
            if (j < Tp) {
                float g = pd[k] * dpd[k] - p0[k] * Di;
                if (mi * rm[j] == 0.f) g = 0.f;              // masked_fill: no gradient to the masked scores
                g *= isd;                                    // scores = (q.k + q.relK) / sqrt(D)
                dr[j] = g; dSg[j] = g;
            }
        }
        // dQ_i[d] = sum_j dS_ij (K[j][d] + [band] relK[j-i+w][d])       (dr written by this wave only; same-wave LDS order suffices)
        for (int d = lane; d < D; d += 64) {
            float acc = 0.f;
            for (int j = 0; j < Tp; ++j) acc += dr[j] * KV[j * ldk + d];
            for (int dd = -win; dd <= win; ++dd) { const int j = i + dd; if (j >= 0 && j < Tp) acc += dr[j] * Rr[(dd + win) * D + d]; }
            dqkv[((long)b * Tp + i) * ld + h * D + d] = acc;
        }
    }
}

// backward, pass B (per key row j): dK_j = sum_i dS_ij q_i ; dV_j = sum_i Pd_ij dO_i.  One wavefront per key, lanes walk d.
__global__ __launch_bounds__(256) void attn_bwd_b_kernel(const float* __restrict__ qkv, const float* __restrict__ P, const float* __restrict__ dS,
                                                         const float* __restrict__ dout, float* __restrict__ dqkv, int B, int Tp, int H, int D)
{
    const int C = H * D, ld = 3 * C;
    const int b = blockIdx.z, h = blockIdx.y;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (j >= Tp) return;
    const float* Pg = P + ((long)b * H + h) * Tp * Tp + j;
    const float* Sg = dS + ((long)b * H + h) * Tp * Tp + j;
    const float* qb = qkv + (long)b * Tp * ld + h * D;
    const float* ob = dout + (long)b * Tp * C + h * D;
    for (int d = lane; d < D; d += 64) {
        float ak = 0.f, av = 0.f;
        for (int i = 0; i < Tp; ++i) { ak += Sg[(long)i * Tp] * qb[(long)i * ld + d]; av += Pg[(long)i * Tp] * ob[(long)i * C + d]; }
        dqkv[((long)b * Tp + j) * ld + C + h * D + d] = ak;
        dqkv[((long)b * Tp + j) * ld + 2 * C + h * D + d] = av;
    }
}

// relative embeddings: drelK[dd+w][d] = sum_{b,h,i} dS_{i,i+dd} q_i[d] ; drelV[dd+w][d] = sum_{b,h,i} Pd_{i,i+dd} dO_i[d]
// grid (2w+1, B*H) -> partial [B*H][2][2w+1][D]; reduced in a fixed order by colsum_final_kernel.
__global__ __launch_bounds__(128) void attn_bwd_rel_kernel(const float* __restrict__ qkv, const float* __restrict__ P, const float* __restrict__ dS,
                                                           const float* __restrict__ dout, float* __restrict__ partial, int B, int Tp, int H, int D, int win)
{
    const int C = H * D, ld = 3 * C;
    const int w = blockIdx.x, bh = blockIdx.y, b = bh / H, h = bh - b * H;
    const int dd = w - win;
    const float* Pg = P + (long)bh * Tp * Tp;
    const float* Sg = dS + (long)bh * Tp * Tp;
    const float* qb = qkv + (long)b * Tp * ld + h * D;
    const float* ob = dout + (long)b * Tp * C + h * D;
    for (int d = threadIdx.x; d < D; d += 128) {
        float ak = 0.f, av = 0.f;
        for (int i = 0; i < Tp; ++i) {
            const int j = i + dd;
            if (j < 0 || j >= Tp) continue;
            ak += Sg[(long)i * Tp + j] * qb[(long)i * ld + d];
            av += Pg[(long)i * Tp + j] * ob[(long)i * C + d];
        }
        const int nw = 2 * win + 1;
        partial[((long)bh * 2 + 0) * nw * D + w * D + d] = ak;
        partial[((long)bh * 2 + 1) * nw * D + w * D + d] = av;
    }
}

inline int grid_for(long total, int per = 256, int cap = 2048) { long g = (total + per - 1) / per; return (int)(g > cap ? cap : (g < 1 ? 1 : g)); }
#define RET_LAUNCH() return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH

}  // namespace

extern "C" int glowtts_layernorm_fwd_io(const float* a, const float* b, float* s_out, const float* gamma, const float* beta, const float* rowmask,
                                        float* y, float* stats, int64_t rows, int C, float eps, int relu, float drop_p, uint32_t seed,
                                        const uint32_t* seed_ptr, uint16_t* y_bf16, void* stream)
{
    if (!a || !gamma || !beta || !y || !stats || rows < 1 || C < 1 || C > 64 * LN_MAXK || (b && !s_out)) return GLOWTTS_E_ARG;
#define LN_FWD_ARGS a, b, s_out, gamma, beta, rowmask, y, stats, (long)rows, C, eps, relu, drop_p, seed, seed_ptr, y_bf16
    const dim3 grid((unsigned)((rows + 3) / 4));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (C == 192)      hipLaunchKernelGGL(ln_fwd_kernel<3>, grid, dim3(256), 0, st, LN_FWD_ARGS);
    else if (C == 256) hipLaunchKernelGGL(ln_fwd_kernel<4>, grid, dim3(256), 0, st, LN_FWD_ARGS);
    else               hipLaunchKernelGGL(ln_fwd_kernel<0>, grid, dim3(256), 0, st, LN_FWD_ARGS);
#undef LN_FWD_ARGS
    RET_LAUNCH();
}
extern "C" int glowtts_layernorm_fwd(const float* a, const float* b, float* s_out, const float* gamma, const float* beta, const float* rowmask,
                                     float* y, float* stats, int64_t rows, int C, float eps, int relu, float drop_p, uint32_t seed,
                                     const uint32_t* seed_ptr, void* stream)
{
    return glowtts_layernorm_fwd_io(a, b, s_out, gamma, beta, rowmask, y, stats, rows, C, eps, relu, drop_p, seed, seed_ptr, nullptr, stream);
}

constexpr int LN_BWD_RPB = LN_BWD_WAVES;  // rows per workgroup of ln_bwd_kernel (one per wavefront)
extern "C" int64_t glowtts_layernorm_scratch_floats(int64_t rows, int C) { return ((rows + LN_BWD_RPB - 1) / LN_BWD_RPB) * 2 * (int64_t)C; }

extern "C" int glowtts_layernorm_bwd_io(const float* dy, const float* y, const float* s, const float* stats, const float* gamma, const float* rowmask,
                                        float* ds, float* dgamma_dbeta /* [2C] or NULL */, float* scratch, int64_t rows, int C, int gated, float drop_p,
                                        uint16_t* ds_bf16, const float* gate_out, float gate_scale, void* stream)
{
    if (!dy || !s || !stats || !gamma || !ds || !scratch || rows < 1 || C < 1 || C > 64 * LN_MAXK || (gated && !y) || (gate_out && !ds_bf16)) return GLOWTTS_E_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int rpb = LN_BWD_RPB;
    const int nblk = (int)((rows + rpb - 1) / rpb);
#define LN_BWD_ARGS dy, y, s, stats, gamma, rowmask, ds, scratch, (long)rows, C, gated, drop_p, rpb, ds_bf16, gate_out, gate_scale
    if (C == 192)      hipLaunchKernelGGL(ln_bwd_kernel<3>, dim3(nblk), dim3(LN_BWD_WAVES * 64), 2 * LN_BWD_WAVES * C * sizeof(float), st, LN_BWD_ARGS);
    else if (C == 256) hipLaunchKernelGGL(ln_bwd_kernel<4>, dim3(nblk), dim3(LN_BWD_WAVES * 64), 2 * LN_BWD_WAVES * C * sizeof(float), st, LN_BWD_ARGS);
    else {
        if (2 * LN_BWD_WAVES * C * sizeof(float) > 64 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(&ln_bwd_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return GLOWTTS_E_LAUNCH;
        hipLaunchKernelGGL(ln_bwd_kernel<0>, dim3(nblk), dim3(LN_BWD_WAVES * 64), 2 * LN_BWD_WAVES * C * sizeof(float), st, LN_BWD_ARGS);
    }
#undef LN_BWD_ARGS
    // dgamma_dbeta == NULL: the per-workgroup partials stay in `scratch` ([ceil(rows / 16)][2C]) for one glowtts_colsum_batched over several calls
    if (dgamma_dbeta) hipLaunchKernelGGL(colsum_final_kernel, dim3((2 * C + 3) / 4), dim3(256), 0, st, scratch, dgamma_dbeta, nblk, 2 * C);
    RET_LAUNCH();
}
extern "C" int glowtts_layernorm_bwd(const float* dy, const float* y, const float* s, const float* stats, const float* gamma, const float* rowmask,
                                     float* ds, float* dgamma_dbeta /* [2C] */, float* scratch, int64_t rows, int C, int gated, float drop_p, void* stream)
{
    if (!dgamma_dbeta) return GLOWTTS_E_ARG;
    return glowtts_layernorm_bwd_io(dy, y, s, stats, gamma, rowmask, ds, dgamma_dbeta, scratch, rows, C, gated, drop_p, nullptr, nullptr, 1.f, stream);
}

extern "C" int glowtts_gate_bwd_io(const void* dy, const void* out, const float* rowmask, void* dz, int64_t rows, int C, float scale, int io_flags, void* stream)
{
    if (!dy || (!out && !rowmask) || !dz || rows < 1 || C < 4 || (C & 3) || (io_flags & ~7)) return GLOWTTS_E_ARG;
    const dim3 grid(grid_for(rows * C / 4));
    hipStream_t st = static_cast<hipStream_t>(stream);
#define GATE_CASE(F, A, B, Z) case F: hipLaunchKernelGGL((gate_bwd_kernel<A, B, Z>), grid, dim3(256), 0, st, dy, out, rowmask, dz, (long)rows, C, scale); break;
    switch (io_flags) {
        GATE_CASE(0, false, false, false) GATE_CASE(1, true, false, false) GATE_CASE(2, false, true, false) GATE_CASE(3, true, true, false)
        GATE_CASE(4, false, false, true)  GATE_CASE(5, true, false, true)  GATE_CASE(6, false, true, true)  GATE_CASE(7, true, true, true)
    }
#undef GATE_CASE
    RET_LAUNCH();
}
extern "C" int glowtts_gate_bwd(const float* dy, const float* out, const float* rowmask, float* dz, int64_t rows, int C, float scale, void* stream)
{
    return glowtts_gate_bwd_io(dy, out, rowmask, dz, rows, C, scale, 0, stream);
}

// token masks of a batch in one launch (Modules.py:206-211 Mask_Generate + the rows layout's padded row mask): mask [B][T] = t < len_b,
// rowmask [B][T + 2 PAD] = the same with GLOWTTS_ROW_PAD zero rows on either side (was arange, compare, cast, pad = 5 launches at the head of the encoder's chain)
__global__ __launch_bounds__(256) void token_masks_kernel(const int64_t* __restrict__ len, float* __restrict__ mask, float* __restrict__ rowmask, int B, int T)
{
    const int Tp = T + 2 * GLOWTTS_ROW_PAD;
    const long total = (long)B * Tp;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int b = (int)(i / Tp), t = (int)(i - (long)b * Tp) - GLOWTTS_ROW_PAD;
        const float v = (t >= 0 && t < T && t < len[b]) ? 1.f : 0.f;
        rowmask[i] = v;
        if (t >= 0 && t < T) mask[(long)b * T + t] = v;
    }
}
extern "C" int glowtts_token_masks(const int64_t* lengths, float* mask, float* rowmask, int B, int T, void* stream)
{
    if (!lengths || !mask || !rowmask || B < 1 || T < 1) return GLOWTTS_E_ARG;
    const long total = (long)B * (T + 2 * GLOWTTS_ROW_PAD);
    hipLaunchKernelGGL(token_masks_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 1024)), dim3(256), 0, static_cast<hipStream_t>(stream), lengths, mask, rowmask, B, T);
    RET_LAUNCH();
}

extern "C" int glowtts_embedding_fwd(const int64_t* tokens, const float* table, const float* rowmask, float* rows, int B, int T, int C, float scale, void* stream)
{
    if (!tokens || !table || !rowmask || !rows || B < 1 || T < 1 || C < 1) return GLOWTTS_E_ARG;
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(grid_for((long)B * (T + 2 * GLOWTTS_ROW_PAD) * C)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       tokens, table, rowmask, rows, B, T, C, scale);
    RET_LAUNCH();
}

extern "C" int glowtts_embedding_bwd(const int64_t* tokens, const float* drows, const float* rowmask, float* dtable, int V, int B, int T, int C, float scale, void* stream)
{
    if (!tokens || !drows || !rowmask || !dtable || V < 1 || B < 1 || T < 1 || C < 1) return GLOWTTS_E_ARG;
    const size_t lds = (256 + (size_t)B * T) * sizeof(int);
    if (lds > 160 * 1024) return GLOWTTS_E_ARG;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(embed_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(V), dim3(256), lds, static_cast<hipStream_t>(stream), tokens, drows, rowmask, dtable, B, T, C, scale);
    RET_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
// MFMA attention core (Tp <= 128, D in {64, 96}, 2*win+1 <= 32): one workgroup per (utterance, head), wave w owns the
// queries [32w, 32w+32).  All contractions run on v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate: the f32 mode stays
// exact; the problem is far too small for the precision of the operands to matter for speed):
//   S = Q K^T (+ Q relK^T on the band) -> masked softmax -> P0 (kept for the backward) -> dropout -> O = Pd V + Pd_band relV.
// MFMA operand element of lane (l31, lhi) at step ks: A[row l31][k = 2 ks + lhi], B[k = 2 ks + lhi][col l31]; the
// accumulator holds C[row (reg & 3) + 8 (reg >> 2) + 4 lhi][col l31].  LDS rows have odd strides (D + 1, 129): the 32 lanes
// of a half-wave then hit 32 different banks whether they walk rows (A / K-as-B operands) or columns (V-as-B operand).
// P stores the probabilities BEFORE dropout (the backward regenerates the keep mask from the same hash).
// ------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 at_bf16x8 __attribute__((ext_vector_type(8)));
constexpr int AT_TP = 128, AT_LDP = AT_TP + 1;

// BF (bf16 precision only): the same contractions on v_mfma_f32_32x32x16_bf16.  The LDS tiles stay fp32; a lane gathers the 8 consecutive
// k-values of its fragment (k = 16 step + 8 lhi + e) with the same addressing as the one fp32 value of the 32x32x2 form and rounds them to
// bf16 in registers, fp32 accumulate, softmax and all element-wise arithmetic unchanged.  One bf16 MFMA does the work of 8 fp32 ones in
// half the matrix-pipe time: the fp32 form spends 64 clk per 4096 FLOP and made these kernels matrix-pipe bound.
template <bool BF> struct AtOp { typedef float T; static constexpr int KSTEP = 2, UNR = 4; };
template <> struct AtOp<true> { typedef at_bf16x8 T; static constexpr int KSTEP = 16, UNR = 2; };
template <bool BF, class F>
__device__ __forceinline__ typename AtOp<BF>::T at_frag(F&& f)
{
    if constexpr (BF) {
        at_bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (__bf16)f(e);
        return v;
    } else {
        return f(0);
    }
}
template <bool BF>
__device__ __forceinline__ f32x16 at_mma(typename AtOp<BF>::T a, typename AtOp<BF>::T b, f32x16 c)
{
    if constexpr (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// Reductions over the 32 lanes that share lane >> 5.  Inside a row of 16 lanes they are DPP moves on the VALU (quad swaps, then the two
// mirrors: after the quad steps every lane holds its quad's total, so mirroring 8 and then 16 lanes completes the row); only the step across
// the two rows goes through the LDS crossbar.  (As five __shfl_xor = ds_bpermute_b32 each, the softmax of one 32 x 128 score tile issued 160
// dependent LDS round trips per wave.)
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
constexpr int DPP_QUAD_1032 = 0xB1, DPP_QUAD_2301 = 0x4E, DPP_ROW_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140;
__device__ __forceinline__ float half_max(float v) {
    v = fmaxf(v, dpp_mov<DPP_QUAD_1032>(v));
    v = fmaxf(v, dpp_mov<DPP_QUAD_2301>(v));
    v = fmaxf(v, dpp_mov<DPP_ROW_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_mov<DPP_ROW_MIRROR>(v));
    return fmaxf(v, __shfl_xor(v, 16, 64));
}
__device__ __forceinline__ float half_sum(float v) {
    v += dpp_mov<DPP_QUAD_1032>(v);
    v += dpp_mov<DPP_QUAD_2301>(v);
    v += dpp_mov<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_mov<DPP_ROW_MIRROR>(v);
    return v + __shfl_xor(v, 16, 64);
}
__device__ __forceinline__ int acc_row(int reg, int lhi) { return (reg & 3) + 8 * (reg >> 2) + 4 * lhi; }

// rows [0, nrows) x D floats of a [*, ld]-strided global tile -> LDS [ROWS][LD] (rows >= nrows zero), NT threads (a workgroup, or one
// wave with NT = 64).  The loads of a pass are issued TOGETHER, unconditionally (rows past the end re-read row 0 and are zeroed on the way to
// LDS), and only then written: as a loop of "if (row < nrows) load; store" hipcc waits for every load before the next one is issued - 12
// dependent L2 / HBM round trips per 128 x 96 operand, which is what the attention kernels spent most of their time on (round 3: forward
// 36 -> see DESIGN.md section 5).
constexpr int stage_chunk(int it) { int c = it < 12 ? it : 12; while (it % c) --c; return c; }      // loads in flight per pass: the largest divisor <= 12
template <int D, int LD, int ROWS, int NT = 256>
__device__ __forceinline__ void stage_rows(float* dst, const float* src, long ld, int nrows, int tid)
{
    constexpr int Q4 = D / 4, N = ROWS * Q4, IT = (N + NT - 1) / NT, CH = stage_chunk(IT);
    static_assert(IT % CH == 0, "passes of equal size");
    if (nrows <= 0) {                              // (uniform) nothing to read: e.g. a wave whose 32 rows lie past the utterance
        for (int i = tid; i < N; i += NT) { const int r = i / Q4, c = (i - r * Q4) * 4; float* o = dst + r * LD + c; o[0] = 0.f; o[1] = 0.f; o[2] = 0.f; o[3] = 0.f; }
        return;
    }
#pragma unroll 1
    for (int p0 = 0; p0 < IT; p0 += CH) {
        float4 v[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int i = tid + (p0 + u) * NT, r = i / Q4, c = (i - r * Q4) * 4;
            const bool ok = (N % NT == 0 || i < N) && r < nrows;
            v[u] = *reinterpret_cast<const float4*>(src + (ok ? (long)r * ld + c : 0L));
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int i = tid + (p0 + u) * NT, r = i / Q4, c = (i - r * Q4) * 4;
            if (N % NT != 0 && i >= N) continue;
            const bool ok = r < nrows;
            float* o = dst + r * LD + c;
            o[0] = ok ? v[u].x : 0.f; o[1] = ok ? v[u].y : 0.f; o[2] = ok ? v[u].z : 0.f; o[3] = ok ? v[u].w : 0.f;
        }
    }
}

template <int ND, bool BF = false>
__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ relk, const float* __restrict__ relv,
                                                            const float* __restrict__ rowmask, float* __restrict__ out, float* __restrict__ P,
                                                            int B, int Tp, int H, int win, float drop_p, uint32_t seed, const uint32_t* __restrict__ seed_ptr)
{
    constexpr int D = ND * 32, LD = D + 1;
    constexpr int KSTEP = AtOp<BF>::KSTEP, UNR = AtOp<BF>::UNR;
    extern __shared__ float sm[];
    float* KV = sm;                               // [128][LD]   K, then V
    float* RL = KV + AT_TP * LD;                  // [32][LD]    relK, then relV (rows >= 2 win + 1 are zero)
    float* PT = RL + 32 * LD;                     // [4][32][129] per wave: Q staging, then the (dropped) probabilities
    float* RQ = PT + 4 * 32 * AT_LDP;             // [4][32][33]  per wave: q_i . relK[dd]
    const int h = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int C = H * D, ld = 3 * C, nw = 2 * win + 1;
    const float* base = qkv + (long)b * Tp * ld + h * D;
    if (seed_ptr && drop_p > 0.f) seed += *seed_ptr;
    float* myP = PT + wave * 32 * AT_LDP;
    float* myR = RQ + wave * 32 * 33;

    stage_rows<D, LD, AT_TP>(KV, base + C, ld, Tp, tid);
    stage_rows<D, LD, 32>(RL, relk, D, nw, tid);
    // this wave's 32 query rows -> A fragments (through its own P region)
    stage_rows<D, LD, 32, 64>(myP, base + (long)wave * 32 * ld, ld, Tp - wave * 32, lane);
    __syncthreads();
    // ---- phase 1: scores ----
    // (the row / column masks the softmax needs: all loads issued here, ahead of the MFMAs - inside the softmax loop each was a
    // dependent round trip to L2 per accumulator row)
    const float* rm = rowmask + (long)b * Tp;
    float mi_[16], mj[4];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) { const int i = 32 * wave + acc_row(reg, lhi); mi_[reg] = rm[min(i, Tp - 1)]; }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) mj[nt] = rm[min(32 * nt + l31, Tp - 1)];
    f32x16 S[4], R;
#pragma unroll
    for (int r = 0; r < 16; ++r) { S[0][r] = 0.f; S[1][r] = 0.f; S[2][r] = 0.f; S[3][r] = 0.f; R[r] = 0.f; }
    const int koff = BF ? 8 * lhi : lhi;          // first k of this lane's fragment inside a step
#pragma unroll UNR
    for (int kb = 0; kb < D; kb += KSTEP) {
        const int k = kb + koff;
        const auto a = at_frag<BF>([&](int e) { return myP[l31 * LD + k + e]; });          // Q fragment (staged with row stride LD)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) S[nt] = at_mma<BF>(a, at_frag<BF>([&](int e) { return KV[(32 * nt + l31) * LD + k + e]; }), S[nt]);
        R = at_mma<BF>(a, at_frag<BF>([&](int e) { return RL[l31 * LD + k + e]; }), R);
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) myR[acc_row(reg, lhi) * 33 + l31] = R[reg];
    __syncthreads();                              // RQ visible; every wave is done with K / relK and with its Q staging

    const float isd = rsqrtf((float)D);
    const float ik = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) mj[nt] = (32 * nt + l31 < Tp) ? mj[nt] : 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = acc_row(reg, lhi), i = 32 * wave + row;
        const float mi = i < Tp ? mi_[reg] : 0.f;
        float sc[4], mx = -3.0e38f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int j = 32 * nt + l31, dd = j - i + win;
            float v = S[nt][reg];
            if (dd >= 0 && dd < nw) v += myR[row * 33 + dd];
            v *= isd;
            if (mi * mj[nt] == 0.f) v = -1e4f;                        // RPR_MHA.py:117
            if (j >= Tp) v = -3.0e38f;
            sc[nt] = v; mx = fmaxf(mx, v);
        }
        mx = half_max(mx);
        float den = 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) { sc[nt] = (32 * nt + l31 < Tp) ? __expf(sc[nt] - mx) : 0.f; den += sc[nt]; }
        den = 1.f / half_sum(den);
        float* Pg = P + (((long)b * H + h) * Tp + i) * Tp;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int j = 32 * nt + l31;
            float p = sc[nt] * den;
            if (i < Tp && j < Tp) Pg[j] = p;
            if (drop_p > 0.f) p *= drop_scale(seed, (uint32_t)((((long)b * H + h) * Tp + i) * Tp + j), drop_p, ik);   // RPR_MHA.py:120
            myP[row * AT_LDP + j] = p;
        }
    }
    // ---- phase 2: O = Pd V + Pd_band relV ----
    stage_rows<D, LD, AT_TP>(KV, base + 2 * C, ld, Tp, tid);
    stage_rows<D, LD, 32>(RL, relv, D, nw, tid);
    __syncthreads();
    f32x16 O[ND];
#pragma unroll
    for (int nd = 0; nd < ND; ++nd)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[nd][r] = 0.f;
#pragma unroll UNR
    for (int kb = 0; kb < AT_TP; kb += KSTEP) {
        const int k = kb + koff;
        const auto a = at_frag<BF>([&](int e) { return myP[l31 * AT_LDP + k + e]; });
#pragma unroll
        for (int nd = 0; nd < ND; ++nd) O[nd] = at_mma<BF>(a, at_frag<BF>([&](int e) { return KV[(k + e) * LD + 32 * nd + l31]; }), O[nd]);
    }
    {
        const int i = 32 * wave + l31;
        for (int db = 0; db < nw; db += KSTEP) {          // (RL has 32 rows, those >= nw are zero)
            const int dd0 = db + koff;
            const auto a = at_frag<BF>([&](int e) { const int dd = dd0 + e, j = i + dd - win; return (dd < nw && j >= 0 && j < Tp) ? myP[l31 * AT_LDP + j] : 0.f; });
#pragma unroll
            for (int nd = 0; nd < ND; ++nd) O[nd] = at_mma<BF>(a, at_frag<BF>([&](int e) { return RL[(dd0 + e) * LD + 32 * nd + l31]; }), O[nd]);
        }
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int i = 32 * wave + acc_row(reg, lhi);
        if (i >= Tp) continue;
#pragma unroll
        for (int nd = 0; nd < ND; ++nd) out[((long)b * Tp + i) * C + h * D + 32 * nd + l31] = O[nd][reg];
    }
}

// Backward of the MFMA attention core, one workgroup per (utterance, head), wave w owns query rows AND key rows [32w, 32w+32).
//   dPd = dO V^T (+ dO relV^T on the band);  Pd = P0 keep / (1-p);  D_i = sum_j Pd dPd;  dS = P0 (keep/(1-p) dPd - D_i) / sqrt(D)
//   dV = Pd^T dO   drelV[dd] = sum_i Pd[i][i+dd-w] dO_i      dQ = dS K + dS_band relK      dK = dS^T Q   drelK[dd] = sum_i dS[i][i+dd-w] q_i
// The 128 x 128 matrices Pd and dS pass through LDS (PT) so that they can be read both row-wise (A operand of dQ) and
// column-wise (A operand of dV / dK); the 128 x D operands are staged one after the other into the same LDS buffer.
// drelK / drelV: per-wave partial sums -> part[(bh * 4 + wave)][2][nw][D] (summed by colsum_final_kernel; deterministic).
template <int ND, bool BF = false>
__global__ __launch_bounds__(256) void attn_bwd_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ relk, const float* __restrict__ relv,
                                                            const float* __restrict__ rowmask, const float* __restrict__ P, const float* __restrict__ dout,
                                                            float* __restrict__ dqkv, float* __restrict__ part,
                                                            int B, int Tp, int H, int win, float drop_p, uint32_t seed, const uint32_t* __restrict__ seed_ptr)
{
    constexpr int D = ND * 32, LD = D + 1;
    constexpr int KSTEP = AtOp<BF>::KSTEP, UNR = AtOp<BF>::UNR;
    extern __shared__ float sm[];
    float* KV = sm;                               // [128][LD]   V, dO, K, Q in turn
    float* RL = KV + AT_TP * LD;                  // [32][LD]    relV, then relK
    float* PT = RL + 32 * LD;                     // [128][129]  dO staging (per wave), then Pd, then dS
    float* RQ = PT + 4 * 32 * AT_LDP;             // [4][32][33] per wave: dO_i . relV[dd]
    const int h = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int C = H * D, ld = 3 * C, nw = 2 * win + 1;
    const float* base = qkv + (long)b * Tp * ld + h * D;
    const float* dob = dout + (long)b * Tp * C + h * D;
    float* dbase = dqkv + (long)b * Tp * ld + h * D;
    if (seed_ptr && drop_p > 0.f) seed += *seed_ptr;
    float* myP = PT + wave * 32 * AT_LDP;
    float* myR = RQ + wave * 32 * 33;
    float* mypart = part + ((long)(b * H + h) * 4 + wave) * 2 * nw * D;

    // ---- phase 1: dPd, D_i, dS (registers), Pd (LDS) ----
    stage_rows<D, LD, AT_TP>(KV, base + 2 * C, ld, Tp, tid);
    stage_rows<D, LD, 32>(RL, relv, D, nw, tid);
    stage_rows<D, LD, 32, 64>(myP, dob + (long)wave * 32 * C, C, Tp - wave * 32, lane);
    __syncthreads();
    // the forward's probabilities of this wave's 16 x 4 accumulator cells: all loads issued here, ahead of the MFMAs (inside the loop below
    // every accumulator row waited for its own round trip to L2)
    float p0r[16][4];
    {
        const float* Pb = P + ((long)b * H + h) * Tp * Tp;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int i = 32 * wave + acc_row(reg, lhi);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) { const int j = 32 * nt + l31; p0r[reg][nt] = Pb[(i < Tp && j < Tp) ? (long)i * Tp + j : 0L]; }
        }
    }
    f32x16 S[4], R;
#pragma unroll
    for (int r = 0; r < 16; ++r) { S[0][r] = 0.f; S[1][r] = 0.f; S[2][r] = 0.f; S[3][r] = 0.f; R[r] = 0.f; }
    const int koff = BF ? 8 * lhi : lhi;
#pragma unroll UNR
    for (int kb = 0; kb < D; kb += KSTEP) {
        const int k = kb + koff;
        const auto a = at_frag<BF>([&](int e) { return myP[l31 * LD + k + e]; });
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) S[nt] = at_mma<BF>(a, at_frag<BF>([&](int e) { return KV[(32 * nt + l31) * LD + k + e]; }), S[nt]);
        R = at_mma<BF>(a, at_frag<BF>([&](int e) { return RL[l31 * LD + k + e]; }), R);
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) myR[acc_row(reg, lhi) * 33 + l31] = R[reg];
    __syncthreads();
    const float isd = rsqrtf((float)D);
    const float ik = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = acc_row(reg, lhi), i = 32 * wave + row;
        float p0[4], kd[4], dsum = 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int j = 32 * nt + l31, dd = j - i + win;
            float dpd = S[nt][reg];
            if (dd >= 0 && dd < nw) dpd += myR[row * 33 + dd];
            p0[nt] = (i < Tp && j < Tp) ? p0r[reg][nt] : 0.f;
            float keep = 1.f;
            if (drop_p > 0.f) keep = drop_scale(seed, (uint32_t)((((long)b * H + h) * Tp + i) * Tp + j), drop_p, ik);
            kd[nt] = keep * dpd;
            dsum += p0[nt] * kd[nt];                                  // Pd dPd
            myP[row * AT_LDP + j] = p0[nt] * keep;                    // Pd
        }
        dsum = half_sum(dsum);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) S[nt][reg] = p0[nt] * (kd[nt] - dsum) * isd;     // dS
    }
    __syncthreads();                              // Pd complete; V / relV no longer needed

    // column-block products with a full 128-row operand in KV:  out[j in own block][d] = sum_i PT[i][j] KV[i][d], plus the
    // relative-embedding gradient of the own query rows:        rel[dd][d]          = sum_{i own} PT[i][i + dd - w] KV[i][d]
    auto col_products = [&](float* dst /* rows j */, float* relpart) __attribute__((always_inline)) {
        f32x16 O[ND], RV[ND];
#pragma unroll
        for (int nd = 0; nd < ND; ++nd)
#pragma unroll
            for (int r = 0; r < 16; ++r) { O[nd][r] = 0.f; RV[nd][r] = 0.f; }
#pragma unroll UNR
        for (int kb = 0; kb < AT_TP; kb += KSTEP) {
            const int i = kb + koff;
            const auto a = at_frag<BF>([&](int e) { return PT[(i + e) * AT_LDP + 32 * wave + l31]; });
#pragma unroll
            for (int nd = 0; nd < ND; ++nd) O[nd] = at_mma<BF>(a, at_frag<BF>([&](int e) { return KV[(i + e) * LD + 32 * nd + l31]; }), O[nd]);
        }
#pragma unroll UNR
        for (int kb = 0; kb < 32; kb += KSTEP) {
            const int i0 = 32 * wave + kb + koff;
            const auto a = at_frag<BF>([&](int e) { const int i = i0 + e, j = i + l31 - win; return (l31 < nw && j >= 0 && j < Tp) ? PT[i * AT_LDP + j] : 0.f; });
#pragma unroll
            for (int nd = 0; nd < ND; ++nd) RV[nd] = at_mma<BF>(a, at_frag<BF>([&](int e) { return KV[(i0 + e) * LD + 32 * nd + l31]; }), RV[nd]);
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = acc_row(reg, lhi), j = 32 * wave + row;
#pragma unroll
            for (int nd = 0; nd < ND; ++nd) {
                if (j < Tp) dst[(long)j * ld + 32 * nd + l31] = O[nd][reg];
                if (row < nw) relpart[row * D + 32 * nd + l31] = RV[nd][reg];
            }
        }
    };

    // ---- phase 2: dV, drelV ----
    stage_rows<D, LD, AT_TP>(KV, dob, C, Tp, tid);
    __syncthreads();
    col_products(dbase + 2 * C, mypart + nw * D);
    __syncthreads();                              // everyone is done with Pd and dO
    // ---- phase 3: dS -> LDS; dQ = dS K + dS_band relK ----
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = acc_row(reg, lhi);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) myP[row * AT_LDP + 32 * nt + l31] = S[nt][reg];
    }
    stage_rows<D, LD, AT_TP>(KV, base + C, ld, Tp, tid);
    stage_rows<D, LD, 32>(RL, relk, D, nw, tid);
    __syncthreads();
    {
        f32x16 O[ND];
#pragma unroll
        for (int nd = 0; nd < ND; ++nd)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[nd][r] = 0.f;
#pragma unroll UNR
        for (int kb = 0; kb < AT_TP; kb += KSTEP) {
            const int k = kb + koff;
            const auto a = at_frag<BF>([&](int e) { return myP[l31 * AT_LDP + k + e]; });
#pragma unroll
            for (int nd = 0; nd < ND; ++nd) O[nd] = at_mma<BF>(a, at_frag<BF>([&](int e) { return KV[(k + e) * LD + 32 * nd + l31]; }), O[nd]);
        }
        const int i = 32 * wave + l31;
        for (int db = 0; db < nw; db += KSTEP) {
            const int dd0 = db + koff;
            const auto a = at_frag<BF>([&](int e) { const int dd = dd0 + e, j = i + dd - win; return (dd < nw && j >= 0 && j < Tp) ? myP[l31 * AT_LDP + j] : 0.f; });
#pragma unroll
            for (int nd = 0; nd < ND; ++nd) O[nd] = at_mma<BF>(a, at_frag<BF>([&](int e) { return RL[(dd0 + e) * LD + 32 * nd + l31]; }), O[nd]);
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int qi = 32 * wave + acc_row(reg, lhi);
            if (qi >= Tp) continue;
#pragma unroll
            for (int nd = 0; nd < ND; ++nd) dbase[(long)qi * ld + 32 * nd + l31] = O[nd][reg];
        }
    }
    __syncthreads();                              // everyone is done with K
    // ---- phase 4: dK, drelK ----
    stage_rows<D, LD, AT_TP>(KV, base, ld, Tp, tid);
    __syncthreads();
    col_products(dbase + C, mypart);
}

// ------------------------------------------------------------------------------------------------
// The same attention core for 128 < Tp <= 256 (the reference trains on texts of up to 200 tokens).  Scores of one query against 256
// keys are 8 accumulator tiles per wave, and a [queries][257] probability tile per wave no longer fits LDS next to four waves and a
// full K / V, so: a workgroup = 2 waves = 64 queries, K and V pass through the 128-row LDS buffer in two halves, and the backward
// is two kernels - attn_bwd_long_q_kernel per query block (dS -> global, dQ, d relK, d relV: everything that reduces over keys) and
// attn_bwd_long_kv_kernel per 64-key block (dK, dV: reductions over all queries, with Pd / dS column blocks staged from global).
// ------------------------------------------------------------------------------------------------
constexpr int ATL_LDP = 257;

template <int ND>
__global__ __launch_bounds__(128) void attn_fwd_long_kernel(const float* __restrict__ qkv, const float* __restrict__ relk, const float* __restrict__ relv,
                                                            const float* __restrict__ rowmask, float* __restrict__ out, float* __restrict__ P,
                                                            int B, int Tp, int H, int win, float drop_p, uint32_t seed, const uint32_t* __restrict__ seed_ptr)
{
    constexpr int D = ND * 32, LD = D + 1, KS = D / 2;
    extern __shared__ float sm[];
    float* KV = sm;                               // [128][LD]   one half of K, then of V
    float* RL = KV + 128 * LD;                    // [32][LD]
    float* PT = RL + 32 * LD;                     // [2][32][257]
    float* RQ = PT + 2 * 32 * ATL_LDP;            // [2][32][33]
    const int qb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int C = H * D, ld = 3 * C, nw = 2 * win + 1, q0 = qb * 64 + wave * 32;
    const float* base = qkv + (long)b * Tp * ld + h * D;
    if (seed_ptr && drop_p > 0.f) seed += *seed_ptr;
    // rows / keys beyond Tp are zero in every staged tile and probability: their 32-key tiles and key pairs are skipped (the same bits, 256 keys' worth of
    // matrix work only when there are 256 keys)
    const int ntile = (Tp + 31) >> 5;
    auto khalf = [&](int hf) __attribute__((always_inline)) -> int { const int n = (Tp - hf * 128 + 1) >> 1; return n < 0 ? 0 : (n > 64 ? 64 : n); };
    float* myP = PT + wave * 32 * ATL_LDP;
    float* myR = RQ + wave * 32 * 33;
    stage_rows<D, LD, 32, 128>(RL, relk, D, nw, tid);
    stage_rows<D, LD, 32, 64>(myP, base + (long)q0 * ld, ld, Tp - q0, lane);          // this wave's 32 query rows
    f32x16 S[8], R;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) S[t][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) R[r] = 0.f;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        __syncthreads();
        stage_rows<D, LD, 128, 128>(KV, base + C + (long)hf * 128 * ld, ld, Tp - hf * 128, tid);
        __syncthreads();
#pragma unroll 2
        for (int ks = 0; ks < KS; ++ks) {
            const int k = 2 * ks + lhi;
            const float a = myP[l31 * LD + k];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) { if (hf * 4 + nt < ntile) S[hf * 4 + nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, KV[(32 * nt + l31) * LD + k], S[hf * 4 + nt], 0, 0, 0); }
            if (hf == 0) R = __builtin_amdgcn_mfma_f32_32x32x2f32(a, RL[l31 * LD + k], R, 0, 0, 0);
        }
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) myR[acc_row(reg, lhi) * 33 + l31] = R[reg];
    __syncthreads();
    const float isd = rsqrtf((float)D);
    const float ik = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    const float* rm = rowmask + (long)b * Tp;
    float mj[8], mi_[16];                          // (all mask loads issued together: see attn_fwd_mfma_kernel)
#pragma unroll
    for (int t = 0; t < 8; ++t) mj[t] = rm[min(32 * t + l31, Tp - 1)];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) mi_[reg] = rm[min(q0 + acc_row(reg, lhi), Tp - 1)];
#pragma unroll
    for (int t = 0; t < 8; ++t) mj[t] = (32 * t + l31 < Tp) ? mj[t] : 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = acc_row(reg, lhi), i = q0 + row;
        const float mi = i < Tp ? mi_[reg] : 0.f;
        float sc[8], mx = -3.0e38f;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int j = 32 * t + l31, dd = j - i + win;
            float v = S[t][reg];
            if (dd >= 0 && dd < nw) v += myR[row * 33 + dd];
            v *= isd;
            if (mi * mj[t] == 0.f) v = -1e4f;
            if (j >= Tp) v = -3.0e38f;
            sc[t] = v; mx = fmaxf(mx, v);
        }
        mx = half_max(mx);
        float den = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) { sc[t] = (32 * t + l31 < Tp) ? __expf(sc[t] - mx) : 0.f; den += sc[t]; }
        den = 1.f / half_sum(den);
        float* Pg = P + (((long)b * H + h) * Tp + i) * Tp;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int j = 32 * t + l31;
            float p = sc[t] * den;
            if (i < Tp && j < Tp) Pg[j] = p;
            if (drop_p > 0.f) p *= drop_scale(seed, (uint32_t)((((long)b * H + h) * Tp + i) * Tp + j), drop_p, ik);
            myP[row * ATL_LDP + j] = p;
        }
    }
    f32x16 O[ND];
#pragma unroll
    for (int nd = 0; nd < ND; ++nd)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[nd][r] = 0.f;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        __syncthreads();
        stage_rows<D, LD, 128, 128>(KV, base + 2 * C + (long)hf * 128 * ld, ld, Tp - hf * 128, tid);
        if (hf == 0) stage_rows<D, LD, 32, 128>(RL, relv, D, nw, tid);
        __syncthreads();
#pragma unroll 4
        for (int ks = 0; ks < khalf(hf); ++ks) {
            const int k = 2 * ks + lhi;
            const float a = myP[l31 * ATL_LDP + hf * 128 + k];
#pragma unroll
            for (int nd = 0; nd < ND; ++nd) O[nd] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, KV[k * LD + 32 * nd + l31], O[nd], 0, 0, 0);
        }
    }
    {
        const int i = q0 + l31;
        for (int ks = 0; ks < (nw + 1) / 2; ++ks) {
            const int dd = 2 * ks + lhi, j = i + dd - win;
            const float a = (dd < nw && j >= 0 && j < Tp) ? myP[l31 * ATL_LDP + j] : 0.f;
#pragma unroll
            for (int nd = 0; nd < ND; ++nd) O[nd] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, RL[dd * LD + 32 * nd + l31], O[nd], 0, 0, 0);
        }
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int i = q0 + acc_row(reg, lhi);
        if (i >= Tp) continue;
#pragma unroll
        for (int nd = 0; nd < ND; ++nd) out[((long)b * Tp + i) * C + h * D + 32 * nd + l31] = O[nd][reg];
    }
}

template <int ND>
__global__ __launch_bounds__(128) void attn_bwd_long_q_kernel(const float* __restrict__ qkv, const float* __restrict__ relk, const float* __restrict__ relv,
                                                              const float* __restrict__ rowmask, const float* __restrict__ P, const float* __restrict__ dout,
                                                              float* __restrict__ dSg, float* __restrict__ dqkv, float* __restrict__ part,
                                                              int B, int Tp, int H, int win, float drop_p, uint32_t seed, const uint32_t* __restrict__ seed_ptr)
{
    constexpr int D = ND * 32, LD = D + 1, KS = D / 2;
    extern __shared__ float sm[];
    float* KV = sm;
    float* RL = KV + 128 * LD;
    float* PT = RL + 32 * LD;
    float* RQ = PT + 2 * 32 * ATL_LDP;
    const int qb = blockIdx.x, nqb = gridDim.x, h = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int C = H * D, ld = 3 * C, nw = 2 * win + 1, q0 = qb * 64 + wave * 32;
    const float* base = qkv + (long)b * Tp * ld + h * D;
    const float* dob = dout + (long)b * Tp * C + h * D;
    float* dbase = dqkv + (long)b * Tp * ld + h * D;
    if (seed_ptr && drop_p > 0.f) seed += *seed_ptr;
    // rows / keys beyond Tp are zero in every staged tile and probability: their 32-key tiles and key pairs are skipped (the same bits, 256 keys' worth of
    // matrix work only when there are 256 keys)
    const int ntile = (Tp + 31) >> 5;
    auto khalf = [&](int hf) __attribute__((always_inline)) -> int { const int n = (Tp - hf * 128 + 1) >> 1; return n < 0 ? 0 : (n > 64 ? 64 : n); };
    float* myP = PT + wave * 32 * ATL_LDP;
    float* myR = RQ + wave * 32 * 33;
    float* mypart = part + ((((long)b * H + h) * nqb + qb) * 2 + wave) * 2 * nw * D;
    // ---- phase 1: dPd, D_i, dS (registers + global), Pd (LDS) ----
    stage_rows<D, LD, 32, 128>(RL, relv, D, nw, tid);
    stage_rows<D, LD, 32, 64>(myP, dob + (long)q0 * C, C, Tp - q0, lane);             // this wave's 32 rows of dO
    f32x16 S[8], R;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) S[t][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) R[r] = 0.f;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        __syncthreads();
        stage_rows<D, LD, 128, 128>(KV, base + 2 * C + (long)hf * 128 * ld, ld, Tp - hf * 128, tid);
        __syncthreads();
#pragma unroll 2
        for (int ks = 0; ks < KS; ++ks) {
            const int k = 2 * ks + lhi;
            const float a = myP[l31 * LD + k];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) { if (hf * 4 + nt < ntile) S[hf * 4 + nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, KV[(32 * nt + l31) * LD + k], S[hf * 4 + nt], 0, 0, 0); }
            if (hf == 0) R = __builtin_amdgcn_mfma_f32_32x32x2f32(a, RL[l31 * LD + k], R, 0, 0, 0);
        }
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) myR[acc_row(reg, lhi) * 33 + l31] = R[reg];
    __syncthreads();
    const float isd = rsqrtf((float)D);
    const float ik = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    // the forward's probabilities of an accumulator row: its eight loads issued together, unconditionally, one row AHEAD of their use
    // (`(i < Tp && j < Tp) ? Pg[j] : 0` in the loop made hipcc wait for each of the 128 loads of a wave before issuing the next)
    const float* Pb = P + ((long)b * H + h) * Tp * Tp;
    auto load_p = [&](int reg, float (&dst)[8]) {
        const int i = q0 + acc_row(reg, lhi);
#pragma unroll
        for (int t = 0; t < 8; ++t) { const int j = 32 * t + l31; dst[t] = Pb[(i < Tp && j < Tp) ? (long)i * Tp + j : 0L]; }
    };
    float pnext[8];
    load_p(0, pnext);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = acc_row(reg, lhi), i = q0 + row;
        float* dSr = dSg + (((long)b * H + h) * Tp + i) * Tp;
        float p0[8], kd[8], dsum = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) p0[t] = (i < Tp && 32 * t + l31 < Tp) ? pnext[t] : 0.f;
        if (reg + 1 < 16) load_p(reg + 1, pnext);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int j = 32 * t + l31, dd = j - i + win;
            float dpd = S[t][reg];
            if (dd >= 0 && dd < nw) dpd += myR[row * 33 + dd];
            float keep = 1.f;
            if (drop_p > 0.f) keep = drop_scale(seed, (uint32_t)((((long)b * H + h) * Tp + i) * Tp + j), drop_p, ik);
            kd[t] = keep * dpd;
            dsum += p0[t] * kd[t];
            myP[row * ATL_LDP + j] = p0[t] * keep;
        }
        dsum = half_sum(dsum);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int j = 32 * t + l31;
            const float ds = p0[t] * (kd[t] - dsum) * isd;
            S[t][reg] = ds;
            if (i < Tp && j < Tp) dSr[j] = ds;
        }
    }
    // rel[dd][d] = sum over the wave's 32 queries i of PT[i][i + dd - w] * X[i][d]   (X = dO or Q rows of this workgroup, staged in KV)
    auto rel_partial = [&](const float* xrows, long xld, float* dst) __attribute__((always_inline)) {
        __syncthreads();
        stage_rows<D, LD, 64, 128>(KV, xrows + (long)qb * 64 * xld, xld, Tp - qb * 64, tid);
        __syncthreads();
        f32x16 RV[ND];
#pragma unroll
        for (int nd = 0; nd < ND; ++nd)
#pragma unroll
            for (int r = 0; r < 16; ++r) RV[nd][r] = 0.f;
#pragma unroll 4
        for (int ks = 0; ks < 16; ++ks) {
            const int il = 2 * ks + lhi, j = q0 + il + l31 - win;
            const float a = (l31 < nw && j >= 0 && j < Tp) ? myP[il * ATL_LDP + j] : 0.f;
#pragma unroll
            for (int nd = 0; nd < ND; ++nd) RV[nd] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, KV[(wave * 32 + il) * LD + 32 * nd + l31], RV[nd], 0, 0, 0);
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = acc_row(reg, lhi);
            if (row >= nw) continue;
#pragma unroll
            for (int nd = 0; nd < ND; ++nd) dst[row * D + 32 * nd + l31] = RV[nd][reg];
        }
    };
    rel_partial(dob, C, mypart + nw * D);                 // d relV (PT holds Pd)
    __syncthreads();
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = acc_row(reg, lhi);
#pragma unroll
        for (int t = 0; t < 8; ++t) myP[row * ATL_LDP + 32 * t + l31] = S[t][reg];
    }
    // ---- dQ = dS K + dS_band relK ----
    f32x16 O[ND];
#pragma unroll
    for (int nd = 0; nd < ND; ++nd)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[nd][r] = 0.f;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        __syncthreads();
        stage_rows<D, LD, 128, 128>(KV, base + C + (long)hf * 128 * ld, ld, Tp - hf * 128, tid);
        if (hf == 0) stage_rows<D, LD, 32, 128>(RL, relk, D, nw, tid);
        __syncthreads();
#pragma unroll 4
        for (int ks = 0; ks < khalf(hf); ++ks) {
            const int k = 2 * ks + lhi;
            const float a = myP[l31 * ATL_LDP + hf * 128 + k];
#pragma unroll
            for (int nd = 0; nd < ND; ++nd) O[nd] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, KV[k * LD + 32 * nd + l31], O[nd], 0, 0, 0);
        }
    }
    {
        const int i = q0 + l31;
        for (int ks = 0; ks < (nw + 1) / 2; ++ks) {
            const int dd = 2 * ks + lhi, j = i + dd - win;
            const float a = (dd < nw && j >= 0 && j < Tp) ? myP[l31 * ATL_LDP + j] : 0.f;
#pragma unroll
            for (int nd = 0; nd < ND; ++nd) O[nd] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, RL[dd * LD + 32 * nd + l31], O[nd], 0, 0, 0);
        }
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int qi = q0 + acc_row(reg, lhi);
        if (qi >= Tp) continue;
#pragma unroll
        for (int nd = 0; nd < ND; ++nd) dbase[(long)qi * ld + 32 * nd + l31] = O[nd][reg];
    }
    rel_partial(base, ld, mypart);                        // d relK (PT holds dS)
}

// dK / dV of one 64-key block: out[j][d] = sum_i X[i][j] Y[i][d] with (X, Y) = (Pd, dO) and (dS, Q)
template <int ND>
__global__ __launch_bounds__(128) void attn_bwd_long_kv_kernel(const float* __restrict__ qkv, const float* __restrict__ P, const float* __restrict__ dSg,
                                                               const float* __restrict__ dout, float* __restrict__ dqkv,
                                                               int B, int Tp, int H, float drop_p, uint32_t seed, const uint32_t* __restrict__ seed_ptr)
{
    constexpr int D = ND * 32, LD = D + 1;
    extern __shared__ float sm[];
    float* XT = sm;                               // [256][65]  column block of Pd, then of dS
    float* KV = XT + 256 * 65;                    // [128][LD]  half of dO, then of Q
    const int jb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int C = H * D, ld = 3 * C, j0 = jb * 64;
    const float* base = qkv + (long)b * Tp * ld + h * D;
    const float* dob = dout + (long)b * Tp * C + h * D;
    float* dbase = dqkv + (long)b * Tp * ld + h * D;
    if (seed_ptr && drop_p > 0.f) seed += *seed_ptr;
    // rows / keys beyond Tp are zero in every staged tile and probability: their 32-key tiles and key pairs are skipped (the same bits, 256 keys' worth of
    // matrix work only when there are 256 keys)
    const int ntile = (Tp + 31) >> 5;
    (void)ntile;
    auto khalf = [&](int hf) __attribute__((always_inline)) -> int { const int n = (Tp - hf * 128 + 1) >> 1; return n < 0 ? 0 : (n > 64 ? 64 : n); };
    const float ik = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    const float* Pb = P + ((long)b * H + h) * Tp * Tp;
    const float* Sb = dSg + ((long)b * H + h) * Tp * Tp;
    auto product = [&](const float* yrows, long yld, float* dst) __attribute__((always_inline)) {
        f32x16 O[ND];
#pragma unroll
        for (int nd = 0; nd < ND; ++nd)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[nd][r] = 0.f;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            __syncthreads();
            stage_rows<D, LD, 128, 128>(KV, yrows + (long)hf * 128 * yld, yld, Tp - hf * 128, tid);
            __syncthreads();
#pragma unroll 4
            for (int ks = 0; ks < khalf(hf); ++ks) {
                const int il = 2 * ks + lhi;
                const float a = XT[(hf * 128 + il) * 65 + 32 * wave + l31];
#pragma unroll
                for (int nd = 0; nd < ND; ++nd) O[nd] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, KV[il * LD + 32 * nd + l31], O[nd], 0, 0, 0);
            }
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int j = j0 + 32 * wave + acc_row(reg, lhi);
            if (j >= Tp) continue;
#pragma unroll
            for (int nd = 0; nd < ND; ++nd) dst[(long)j * ld + 32 * nd + l31] = O[nd][reg];
        }
    };
    for (int idx = tid; idx < ((Tp + 1) & ~1) * 64; idx += 128) {       // Pd[:, block]
        const int i = idx >> 6, jj = idx & 63, j = j0 + jj;
        float v = 0.f;
        if (i < Tp && j < Tp) {
            v = Pb[(long)i * Tp + j];
            if (drop_p > 0.f) v *= drop_scale(seed, (uint32_t)((((long)b * H + h) * Tp + i) * Tp + j), drop_p, ik);
        }
        XT[i * 65 + jj] = v;
    }
    product(dob, C, dbase + 2 * C);                         // dV
    __syncthreads();
    for (int idx = tid; idx < ((Tp + 1) & ~1) * 64; idx += 128) {       // dS[:, block]
        const int i = idx >> 6, jj = idx & 63, j = j0 + jj;
        XT[i * 65 + jj] = (i < Tp && j < Tp) ? Sb[(long)i * Tp + j] : 0.f;
    }
    product(base, ld, dbase + C);                           // dK
}

static bool attn_mfma_ok(int Tp, int D, int win)          // 1: single-workgroup kernels (Tp <= 128); 2: long kernels (Tp <= 256); 0: general fp32 FMA kernels
{
    const bool enabled = GLOWTTS_TUNABLE("GLOWTTS_ATTN_MFMA", 1) != 0;
    return enabled && Tp <= AT_TP && Tp < GLOWTTS_TUNABLE("GLOWTTS_ATTN_LONG_MIN", AT_TP + 1) && (D == 64 || D == 96) && 2 * win + 1 <= 32;
}
static bool attn_long_ok(int Tp, int D, int win)
{
    const bool enabled = GLOWTTS_TUNABLE("GLOWTTS_ATTN_MFMA", 1) != 0;
    return enabled && Tp >= GLOWTTS_TUNABLE("GLOWTTS_ATTN_LONG_MIN", AT_TP + 1) && Tp <= 256 && (D == 64 || D == 96) && 2 * win + 1 <= 32;
}
static size_t attn_long_lds(int D) { return ((size_t)(128 + 32) * (D + 1) + 2 * 32 * ATL_LDP + 2 * 32 * 33) * sizeof(float); }
static size_t attn_long_kv_lds(int D) { return ((size_t)256 * 65 + (size_t)128 * (D + 1)) * sizeof(float); }
static size_t attn_mfma_lds(int D) { return ((size_t)(AT_TP + 32) * (D + 1) + 4 * 32 * AT_LDP + 4 * 32 * 33) * sizeof(float); }

static size_t attn_lds_bytes(int Tp, int D, int win, bool)
{
    return ((size_t)Tp * (D + 1) + (size_t)(2 * win + 1) * D + 4 * (size_t)D + (size_t)ATT_QT * Tp) * sizeof(float);
}

template <int ND, bool BF>
static void launch_attn_fwd_mfma(size_t lds, hipStream_t st, const float* qkv, const float* relk, const float* relv, const float* rowmask, float* out, float* P,
                                 int B, int Tp, int H, int win, float drop_p, uint32_t seed, const uint32_t* seed_ptr)
{
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_mfma_kernel<ND, BF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    GLOWTTS_NOTE_STATIC("attn_fwd_mfma<%d,%s>", ND * 32, BF ? "bf16" : "f32");
    hipLaunchKernelGGL((attn_fwd_mfma_kernel<ND, BF>), dim3(H, B), dim3(256), lds, st, qkv, relk, relv, rowmask, out, P, B, Tp, H, win, drop_p, seed, seed_ptr);
}
template <int ND, bool BF>
static void launch_attn_bwd_mfma(size_t lds, hipStream_t st, const float* qkv, const float* relk, const float* relv, const float* rowmask, const float* P,
                                 const float* dout, float* dqkv, float* scratch, int B, int Tp, int H, int win, float drop_p, uint32_t seed, const uint32_t* seed_ptr)
{
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_mfma_kernel<ND, BF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    GLOWTTS_NOTE_STATIC("attn_bwd_mfma<%d,%s>", ND * 32, BF ? "bf16" : "f32");
    hipLaunchKernelGGL((attn_bwd_mfma_kernel<ND, BF>), dim3(H, B), dim3(256), lds, st, qkv, relk, relv, rowmask, P, dout, dqkv, scratch, B, Tp, H, win, drop_p, seed, seed_ptr);
}

extern "C" int glowtts_rpr_attention_fwd_prec(const float* qkv, const float* relk, const float* relv, const float* rowmask, float* out, float* P,
                                              int B, int Tp, int H, int D, int win, float drop_p, uint32_t seed, const uint32_t* seed_ptr, int precision, void* stream)
{
    if (!qkv || !relk || !relv || !rowmask || !out || !P || B < 1 || Tp < 1 || Tp > 256 || H < 1 || D < 1 || win < 0) return GLOWTTS_E_ARG;
    if (precision != GLOWTTS_F32 && precision != GLOWTTS_BF16) return GLOWTTS_E_ARG;
    const bool bf = precision == GLOWTTS_BF16;
    if (attn_mfma_ok(Tp, D, win)) {
        const size_t l2 = attn_mfma_lds(D);
        hipStream_t st = static_cast<hipStream_t>(stream);
        if (D == 96) {
            if (bf) launch_attn_fwd_mfma<3, true>(l2, st, qkv, relk, relv, rowmask, out, P, B, Tp, H, win, drop_p, seed, seed_ptr);
            else    launch_attn_fwd_mfma<3, false>(l2, st, qkv, relk, relv, rowmask, out, P, B, Tp, H, win, drop_p, seed, seed_ptr);
        } else {
            if (bf) launch_attn_fwd_mfma<2, true>(l2, st, qkv, relk, relv, rowmask, out, P, B, Tp, H, win, drop_p, seed, seed_ptr);
            else    launch_attn_fwd_mfma<2, false>(l2, st, qkv, relk, relv, rowmask, out, P, B, Tp, H, win, drop_p, seed, seed_ptr);
        }
        RET_LAUNCH();
    }
    if (attn_long_ok(Tp, D, win)) {
        const size_t l2 = attn_long_lds(D);
        hipStream_t st = static_cast<hipStream_t>(stream);
        const dim3 grid((Tp + 63) / 64, H, B);
        if (D == 96) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_long_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
            hipLaunchKernelGGL(attn_fwd_long_kernel<3>, grid, dim3(128), l2, st, qkv, relk, relv, rowmask, out, P, B, Tp, H, win, drop_p, seed, seed_ptr);
        } else {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_long_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
            hipLaunchKernelGGL(attn_fwd_long_kernel<2>, grid, dim3(128), l2, st, qkv, relk, relv, rowmask, out, P, B, Tp, H, win, drop_p, seed, seed_ptr);
        }
        RET_LAUNCH();
    }
    const size_t lds = attn_lds_bytes(Tp, D, win, false);
    if (lds > 160 * 1024) return GLOWTTS_E_ARG;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(attn_fwd_kernel, dim3((Tp + ATT_QT - 1) / ATT_QT, H, B), dim3(256), lds, static_cast<hipStream_t>(stream),
                       qkv, relk, relv, rowmask, out, P, B, Tp, H, D, win, drop_p, seed, seed_ptr);
    RET_LAUNCH();
}

extern "C" int glowtts_rpr_attention_fwd(const float* qkv, const float* relk, const float* relv, const float* rowmask, float* out, float* P,
                                         int B, int Tp, int H, int D, int win, float drop_p, uint32_t seed, const uint32_t* seed_ptr, void* stream)
{
    return glowtts_rpr_attention_fwd_prec(qkv, relk, relv, rowmask, out, P, B, Tp, H, D, win, drop_p, seed, seed_ptr, GLOWTTS_F32, stream);
}

extern "C" int64_t glowtts_rpr_attention_scratch_floats(int B, int Tp, int H, int D, int win) { return (int64_t)B * H * 8 * 2 * (2 * win + 1) * D; }

extern "C" int64_t glowtts_rpr_attention_bwd_partial_rows(int B, int Tp, int H, int D, int win)
{
    if (attn_mfma_ok(Tp, D, win)) return (int64_t)B * H * 4;
    if (attn_long_ok(Tp, D, win)) return (int64_t)B * H * ((Tp + 63) / 64) * 2;
    return (int64_t)B * H;
}

extern "C" int glowtts_rpr_attention_bwd_prec(const float* qkv, const float* relk, const float* relv, const float* rowmask, const float* P, const float* dout,
                                              float* dS /* [B][H][Tp][Tp] scratch */, float* dqkv, float* drelk, float* drelv, float* scratch,
                                              int B, int Tp, int H, int D, int win, float drop_p, uint32_t seed, const uint32_t* seed_ptr, int precision, void* stream)
{
    // drelk == drelv == NULL (round 5): the partial rows stay in `scratch` ([glowtts_rpr_attention_bwd_partial_rows][2 nw D]) for the caller's own reduction
    // (one glowtts_colsum_batched over all layers, off the encoder's backward chain)
    if (!qkv || !relk || !relv || !rowmask || !P || !dout || !dS || !dqkv || (!drelk != !drelv) || !scratch || Tp > 256) return GLOWTTS_E_ARG;
    if (precision != GLOWTTS_F32 && precision != GLOWTTS_BF16) return GLOWTTS_E_ARG;
    const bool bf = precision == GLOWTTS_BF16;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nw = 2 * win + 1;
    // per-(utterance, head[, wave]) partials of drelK | drelV, summed into `both`; when the caller's drelk / drelv are adjacent
    // (one [2][nw][D] tensor) the sum is written in place
    int prow;
    if (attn_mfma_ok(Tp, D, win)) {
        const size_t l2 = attn_mfma_lds(D);
        if (D == 96) {
            if (bf) launch_attn_bwd_mfma<3, true>(l2, st, qkv, relk, relv, rowmask, P, dout, dqkv, scratch, B, Tp, H, win, drop_p, seed, seed_ptr);
            else    launch_attn_bwd_mfma<3, false>(l2, st, qkv, relk, relv, rowmask, P, dout, dqkv, scratch, B, Tp, H, win, drop_p, seed, seed_ptr);
        } else {
            if (bf) launch_attn_bwd_mfma<2, true>(l2, st, qkv, relk, relv, rowmask, P, dout, dqkv, scratch, B, Tp, H, win, drop_p, seed, seed_ptr);
            else    launch_attn_bwd_mfma<2, false>(l2, st, qkv, relk, relv, rowmask, P, dout, dqkv, scratch, B, Tp, H, win, drop_p, seed, seed_ptr);
        }
        prow = B * H * 4;
    } else if (attn_long_ok(Tp, D, win)) {
        const size_t l2 = attn_long_lds(D), l3 = attn_long_kv_lds(D);
        const int nqb = (Tp + 63) / 64;
        const dim3 grid(nqb, H, B);
        if (D == 96) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_long_q_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_long_kv_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l3);
            hipLaunchKernelGGL(attn_bwd_long_q_kernel<3>, grid, dim3(128), l2, st, qkv, relk, relv, rowmask, P, dout, dS, dqkv, scratch, B, Tp, H, win, drop_p, seed, seed_ptr);
            hipLaunchKernelGGL(attn_bwd_long_kv_kernel<3>, grid, dim3(128), l3, st, qkv, P, dS, dout, dqkv, B, Tp, H, drop_p, seed, seed_ptr);
        } else {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_long_q_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_long_kv_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l3);
            hipLaunchKernelGGL(attn_bwd_long_q_kernel<2>, grid, dim3(128), l2, st, qkv, relk, relv, rowmask, P, dout, dS, dqkv, scratch, B, Tp, H, win, drop_p, seed, seed_ptr);
            hipLaunchKernelGGL(attn_bwd_long_kv_kernel<2>, grid, dim3(128), l3, st, qkv, P, dS, dout, dqkv, B, Tp, H, drop_p, seed, seed_ptr);
        }
        prow = B * H * nqb * 2;
    } else {
        const size_t lds = attn_lds_bytes(Tp, D, win, true);
        if (lds > 160 * 1024) return GLOWTTS_E_ARG;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_a_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(attn_bwd_a_kernel, dim3((Tp + ATT_QT - 1) / ATT_QT, H, B), dim3(256), lds, st, qkv, relk, relv, rowmask, P, dout, dS, dqkv, B, Tp, H, D, win, drop_p);
        hipLaunchKernelGGL(attn_bwd_b_kernel, dim3((Tp + 3) / 4, H, B), dim3(256), 0, st, qkv, P, dS, dout, dqkv, B, Tp, H, D);
        hipLaunchKernelGGL(attn_bwd_rel_kernel, dim3(nw, B * H), dim3(128), 0, st, qkv, P, dS, dout, scratch, B, Tp, H, D, win);
        prow = B * H;
    }
    if (!drelk) { RET_LAUNCH(); }
    const bool adjacent = (drelv == drelk + (size_t)nw * D);
    float* both = adjacent ? drelk : scratch + glowtts_rpr_attention_scratch_floats(B, Tp, H, D, win);   // else the caller's scratch has 2*nw*D more floats
    hipLaunchKernelGGL(colsum_final_kernel, dim3((2 * nw * D + 3) / 4), dim3(256), 0, st, scratch, both, prow, 2 * nw * D);
    if (!adjacent) {
        if (hipMemcpyAsync(drelk, both, (size_t)nw * D * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return GLOWTTS_E_LAUNCH;
        if (hipMemcpyAsync(drelv, both + nw * D, (size_t)nw * D * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return GLOWTTS_E_LAUNCH;
    }
    RET_LAUNCH();
}

extern "C" int glowtts_rpr_attention_bwd(const float* qkv, const float* relk, const float* relv, const float* rowmask, const float* P, const float* dout,
                                         float* dS, float* dqkv, float* drelk, float* drelv, float* scratch,
                                         int B, int Tp, int H, int D, int win, float drop_p, uint32_t seed, const uint32_t* seed_ptr, void* stream)
{
    return glowtts_rpr_attention_bwd_prec(qkv, relk, relv, rowmask, P, dout, dS, dqkv, drelk, drelv, scratch, B, Tp, H, D, win, drop_p, seed, seed_ptr,
                                          GLOWTTS_F32, stream);
}

// Bandwidth-bound pieces of the flow decoder for gfx950 (everything that is not a dense contraction):
// squeeze / unsqueeze (Modules.py:895-924), ActNorm (:682-711), invertible 1x1 conv (:727-758),
// the coupling backward, log-determinant bookkeeping.  All activations are fp32 "rows" tensors
// [B][Tp][C] (channels contiguous, Tp = T + 2*GLOWTTS_ROW_PAD, zero pad rows around every utterance).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/glowtts_hip.h"
#include "launch_log.h"
#include "tunable.h"
#include "actnorm_math.h"

namespace {

#define PADR GLOWTTS_ROW_PAD

// ------------------------------------------------------------------------------------------------
// squeeze: mel [B][Cm][Tm] -> rows [B][Tp][ns*Cm];  rows[b][PADR+t][s*Cm+c] = mel[b][c][ns*t+s] * mask'[t]
// mask'[t] = (ns*t + ns-1 < len[b])   (Modules.py:903: mask[:, :, ns-1::ns]).  Also writes rowmask.
// ------------------------------------------------------------------------------------------------
#ifndef GLOWTTS_SQ_TS
#define GLOWTTS_SQ_TS 16
#endif
constexpr int SQ_TS = GLOWTTS_SQ_TS;            // squeezed frames per workgroup of squeeze_kernel
template <bool TO_ROWS>
__global__ __launch_bounds__(256) void squeeze_kernel(float* __restrict__ mel, float* __restrict__ rows,
                                                      float* __restrict__ rowmask, const int64_t* __restrict__ lengths,
                                                      int Cm, int Tm, int T, int ns, float fill, int use_fill)
{
    // block: one utterance, SQ_TS squeezed frames (= SQ_TS*ns mel frames), all channels.  LDS tile [Cm][SQ_TS*ns + 1]
    // (16 frames: 800+ workgroups of 10 KiB at the bench size; with 32 the 416 workgroups left the copy latency-bound at 0.9 TB/s)
    extern __shared__ float tile[];
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * SQ_TS;
    const int Tp = T + 2 * PADR;
    const int W = SQ_TS * ns;                 // mel frames per tile
    const int ldt = W + 1;
    const int C = Cm * ns;
    const long len = lengths[b];
    float* melb = mel + (long)b * Cm * Tm;
    float* rowb = rows + ((long)b * Tp + PADR) * C;
    if (TO_ROWS) {
        for (int i = threadIdx.x; i < Cm * W; i += 256) {
            const int c = i / W, y = i - c * W;
            const int yy = t0 * ns + y;
            tile[c * ldt + y] = (yy < T * ns) ? melb[(long)c * Tm + yy] : 0.f;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < SQ_TS * C; i += 256) {
            const int t = i / C, ch = i - t * C;
            if (t0 + t >= T) continue;
            const int s = ch / Cm, c = ch - s * Cm;
            const float m = ((long)(t0 + t) * ns + ns - 1 < len) ? 1.f : 0.f;
            rowb[(long)(t0 + t) * C + ch] = tile[c * ldt + t * ns + s] * m;
            if (ch == 0 && rowmask) rowmask[(long)b * Tp + PADR + t0 + t] = m;
        }
        if (blockIdx.x == 0) {             // zero the pad rows of this utterance (and their mask)
            float* base = rows + (long)b * Tp * C;
            for (int i = threadIdx.x; i < PADR * C; i += 256) { base[i] = 0.f; base[(long)(PADR + T) * C + i] = 0.f; }
            if (rowmask && threadIdx.x < PADR) { rowmask[(long)b * Tp + threadIdx.x] = 0.f; rowmask[(long)b * Tp + PADR + T + threadIdx.x] = 0.f; }
        }
    } else {
        // rows -> mel (unsqueeze, Modules.py:914-924): mel[b][c][ns*t+s] = rows[b][t][s*Cm+c] * mask'[t]; optional pad fill
        for (int i = threadIdx.x; i < SQ_TS * C; i += 256) {
            const int t = i / C, ch = i - t * C;
            const int s = ch / Cm, c = ch - s * Cm;
            float v = 0.f;
            if (t0 + t < T) {
                const bool valid = ((long)(t0 + t) * ns + ns - 1 < len);
                v = valid ? rowb[(long)(t0 + t) * C + ch] : (use_fill ? fill : 0.f);
            }
            tile[c * ldt + t * ns + s] = v;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < Cm * W; i += 256) {
            const int c = i / W, y = i - c * W;
            const int yy = t0 * ns + y;
            if (yy < T * ns) melb[(long)c * Tm + yy] = tile[c * ldt + y];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 4x4 helper: inverse + log-determinant of the inv-1x1 weights of all flows (torch.inverse / torch.logdet,
// Modules.py:743,747).  One thread per flow, Gauss-Jordan with partial pivoting in fp32... in fp64 for safety.
// out[f] = { W[16], Winv[16], logdet, sign }  (34 floats, stride 36)
// ------------------------------------------------------------------------------------------------
__global__ void inv4x4_kernel(const float* __restrict__ W, float* __restrict__ out, int F)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    double a[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { a[i][j] = W[f * 16 + i * 4 + j]; a[i][4 + j] = (i == j) ? 1.0 : 0.0; }
    double det = 1.0;
    for (int col = 0; col < 4; ++col) {
        int piv = col;
        for (int r = col + 1; r < 4; ++r) if (fabs(a[r][col]) > fabs(a[piv][col])) piv = r;
        if (piv != col) { for (int j = 0; j < 8; ++j) { double t = a[col][j]; a[col][j] = a[piv][j]; a[piv][j] = t; } det = -det; }
        const double d = a[col][col];
        det *= d;
        const double inv = 1.0 / d;
        for (int j = 0; j < 8; ++j) a[col][j] *= inv;
        for (int r = 0; r < 4; ++r) if (r != col) { const double m = a[r][col]; for (int j = 0; j < 8; ++j) a[r][j] -= m * a[col][j]; }
    }
    float* o = out + f * 36;
    for (int i = 0; i < 16; ++i) { o[i] = W[f * 16 + i]; o[16 + i] = (float)a[i / 4][4 + (i % 4)]; }
    o[32] = (float)log(fabs(det));          // torch.logdet is nan for det < 0; the reference keeps det > 0 (Modules.py:722-723)
    o[33] = det > 0 ? 1.f : -1.f;
}

// ------------------------------------------------------------------------------------------------
// ActNorm + invertible 1x1, forward and inverse, one pass over the rows.
// thread = (row, group g): channels {2g, 2g+1, C/2+2g, C/2+2g+1} <-> split index 0..3  (Modules.py:738-740)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t an_pack_bf16x2(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v; v[0] = (__bf16)lo; v[1] = (__bf16)hi;
    return *reinterpret_cast<uint32_t*>(&v);
}

template <bool REVERSE>
__global__ __launch_bounds__(256) void actnorm_inv_kernel(const float* __restrict__ xin, float* __restrict__ xout,
                                                          const float* __restrict__ logs, const float* __restrict__ bias,
                                                          const float* __restrict__ winfo, const float* __restrict__ rowmask,
                                                          long rows, int C, float* __restrict__ xpass, uint32_t* __restrict__ xa_bf = nullptr)
{
    // xpass (forward only, may be null): the first C/2 output channels are also written there - the coupling layer passes x_a
    // through unchanged (Modules.py:808), so the flow's output buffer gets its first half without a separate copy kernel
    const int G = C / 4, C2 = C / 2;
    const long total = rows * G;
    const float* Wm = winfo + (REVERSE ? 16 : 0);
    float w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = Wm[i];
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / G;
        const int g = (int)(i - r * G);
        const float m = rowmask[r];
        const float2 lo = *reinterpret_cast<const float2*>(xin + r * C + 2 * g);
        const float2 hi = *reinterpret_cast<const float2*>(xin + r * C + C2 + 2 * g);
        float v[4] = {lo.x, lo.y, hi.x, hi.y};
        const int ch[4] = {2 * g, 2 * g + 1, C2 + 2 * g, C2 + 2 * g + 1};
        float o[4];
        if (!REVERSE) {
            float e4[4], b4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { e4[k] = expf(logs[ch[k]]); b4[k] = bias[ch[k]]; }
            actnorm_mix4(v, e4, b4, w, m, o);                  // (shared with the fused coupling launch's epilogue: actnorm_math.h)
        } else {
            float u[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) u[k] = (w[k * 4 + 0] * v[0] + w[k * 4 + 1] * v[1] + w[k * 4 + 2] * v[2] + w[k * 4 + 3] * v[3]) * m;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (u[k] - bias[ch[k]]) * expf(-logs[ch[k]]) * m;      // Modules.py:690
        }
        *reinterpret_cast<float2*>(xout + r * C + 2 * g) = make_float2(o[0], o[1]);
        *reinterpret_cast<float2*>(xout + r * C + C2 + 2 * g) = make_float2(o[2], o[3]);
        if (!REVERSE && xpass) *reinterpret_cast<float2*>(xpass + r * C + 2 * g) = make_float2(o[0], o[1]);
        if (!REVERSE && xa_bf) xa_bf[(r * C2 + 2 * g) >> 1] = an_pack_bf16x2(o[0], o[1]);      // x_a as bf16 rows [rows][C/2] (Start conv weight gradient operand)
    }
}

// ------------------------------------------------------------------------------------------------
// column statistics over rows (deterministic two-stage): sums of x*m, x*x*m per channel and of m.
// Used by the ActNorm data-dependent init (Modules.py:698-711).  partial: [nblk][2*C + 1]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colstats_partial_kernel(const float* __restrict__ x, const float* __restrict__ rowmask,
                                                               float* __restrict__ partial, long rows, int C, int rows_per_block)
{
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(rows, r0 + rows_per_block);
    float* out = partial + (long)blockIdx.x * (2 * C + 1);
    for (int c = threadIdx.x; c < C; c += 256) {
        float s1 = 0.f, s2 = 0.f;
        for (long r = r0; r < r1; ++r) { const float m = rowmask[r]; const float v = x[r * C + c]; s1 += v * m; s2 += v * v * m; }
        out[c] = s1; out[C + c] = s2;
    }
    if (threadIdx.x == 0) { float sm = 0.f; for (long r = r0; r < r1; ++r) sm += rowmask[r]; out[2 * C] = sm; }
}
__global__ __launch_bounds__(256) void colstats_final_kernel(const float* __restrict__ partial, float* __restrict__ stats, int nblk, int n,
                                                             long part_stride = 0, long out_stride = 0)
{
    // one wavefront per output: lanes stride over the partial rows, fixed-order shuffle tree (deterministic); blockIdx.y = problem
    partial += (long)blockIdx.y * part_stride; stats += (long)blockIdx.y * out_stride;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= n) return;
    float s = 0.f;
    // (eight loads in flight, same order of additions: as a rolled loop every strided load was waited for on its own)
    for (int k0 = lane; k0 < nblk; k0 += 8 * 64) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int k = k0 + 64 * u; v[u] = partial[(long)(k < nblk ? k : nblk - 1) * n + i]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) if (k0 + 64 * u < nblk) s += v[u];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (lane == 0) stats[i] = s;
}
// logs = -0.5*log(max(var,1e-7)), bias = -mean*exp(logs)   from stats = [sum x, sum x^2, sum m]
__global__ void actnorm_from_stats_kernel(const float* __restrict__ stats, float* __restrict__ logs, float* __restrict__ bias, int C)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float den = stats[2 * C];
    const float mean = stats[c] / den;
    const float sq = stats[C + c] / den;
    const float l = 0.5f * logf(fmaxf(sq - mean * mean, 1e-7f));
    logs[c] = -l;
    bias[c] = -mean * expf(-l);
}

// ------------------------------------------------------------------------------------------------
// coupling backward (autograd of Modules.py:805-806), elementwise:
//   z_b = (m + exp(logs) x_b) mask ; logdet = sum logs*mask
//   d m = dz_b mask ; d logs = (dz_b exp(logs) x_b + dld[b]) mask ; d x_b = dz_b exp(logs) mask
// outs / douts are PAIR-packed [R][npair*64] (m in the first 32 of each 64, logs in the second 32)
// ------------------------------------------------------------------------------------------------
// douts16 (optional): a second, bf16 copy of douts - the A operand of the End conv's data gradient on the LDS-DMA / chained path
__global__ __launch_bounds__(256) void coupling_bwd_kernel(float* __restrict__ dz, const float* __restrict__ xmid,
                                                           const float* __restrict__ outs, float* __restrict__ douts, __bf16* __restrict__ douts16,
                                                           const float* __restrict__ rowmask, const float* __restrict__ dld,
                                                           long rows, int C, int ldo, int rows_per_utt)
{
    // one thread per (row, pair slot): slots past C/2 are the pad columns of the PAIR packing; they are written as zeros here so that the
    // caller needs no memset of douts (the End conv's data / weight gradients read them)
    const int C2 = C / 2, P2 = ldo / 2;
    const long total = rows * P2;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / P2;
        const int j = (int)(i - r * P2);
        const int pc = (j >> 5) * 64 + (j & 31);
        if (j >= C2) {
            if (douts) { douts[r * ldo + pc] = 0.f; douts[r * ldo + pc + 32] = 0.f; }
            if (douts16) { douts16[r * ldo + pc] = (__bf16)0.f; douts16[r * ldo + pc + 32] = (__bf16)0.f; }
            continue;
        }
        const float m = rowmask[r];
        const float logs = outs[r * ldo + pc + 32];
        const float e = expf(logs);
        const float d = dz[r * C + C2 + j];
        const float xb = xmid[r * C + C2 + j];
        const float dm = d * m, dl = (d * e * xb + dld[r / rows_per_utt]) * m;
        if (douts) {
            douts[r * ldo + pc] = dm;
            douts[r * ldo + pc + 32] = dl;
        }
        if (douts16) { douts16[r * ldo + pc] = (__bf16)dm; douts16[r * ldo + pc + 32] = (__bf16)dl; }
        dz[r * C + C2 + j] = d * e * m;
    }
}
// zero the pad columns of a PAIR-packed rows buffer once (so dgrad / wgrad read zeros there)
__global__ void zero_kernel(float* __restrict__ p, long n)
{
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] = 0.f;
}

// ------------------------------------------------------------------------------------------------
struct NextCoupling {            // the previous flow's coupling backward, fused into actnorm_inv_bwd_kernel (all null: not fused)
    const float* xmid; const float* outs; float* douts; __bf16* douts16; const float* dld; int ldo; int rows_per_utt;
};

// inv-1x1 + ActNorm backward.  dz: grad wrt the inv-1x1 output (rows, C).  x: the flow input (ActNorm input).
//   y = (bias + exp(logs) x) mask ; z = (W y) mask
//   dy = W^T (dz mask) ; dx = dy exp(logs) mask ; dW += (dz mask) y^T ; dlogs += dy exp(logs) x mask ; dbias += dy mask
// parameter-grad partials per block: [nblk][2*C + 16], reduced by colstats_final_kernel.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void actnorm_inv_bwd_kernel(const float* __restrict__ dz, float* __restrict__ dx,
                                                              const float* __restrict__ x, const float* __restrict__ logs,
                                                              const float* __restrict__ bias, const float* __restrict__ winfo,
                                                              const float* __restrict__ rowmask, float* __restrict__ partial,
                                                              long rows, int C, int rows_per_block, const NextCoupling nc)
{
    // nc.outs != null: dx is the gradient of the PREVIOUS flow's output, and that flow's affine-coupling backward (coupling_bwd_kernel) is
    // applied to it on the fly - the thread that produces d x_b[2g], d x_b[2g+1] has everything that elementwise step needs
    // thread owns group g = threadIdx.x % G for rows r0 + threadIdx.x / G, stepping by 256 / G ... simpler: loop
    const int G = C / 4, C2 = C / 2;
    extern __shared__ float red[];                 // [256][...] not used: accumulate per-thread then LDS reduce per channel
    float w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = winfo[i];
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(rows, r0 + rows_per_block);
    // each thread handles a fixed group (g = tid % G) and rows r0 + tid / G + k * (256 / G)
    const int tpg = 256 / G;                       // row lanes per pass (>= 1 since G <= 256 is checked on the host)
    const int g = threadIdx.x % G;
    const int rl = threadIdx.x / G;
    float accW[16], accL[4], accB[4];
#pragma unroll
    for (int i = 0; i < 16; ++i) accW[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { accL[i] = 0.f; accB[i] = 0.f; }
    const int ch[4] = {2 * g, 2 * g + 1, C2 + 2 * g, C2 + 2 * g + 1};
    float el[4], bs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { el[k] = expf(logs[ch[k]]); bs[k] = bias[ch[k]]; }
    if (rl < tpg) {
        for (long r = r0 + rl; r < r1; r += tpg) {
            const float m = rowmask[r];
            const float2 dlo = *reinterpret_cast<const float2*>(dz + r * C + 2 * g);
            const float2 dhi = *reinterpret_cast<const float2*>(dz + r * C + C2 + 2 * g);
            const float2 xlo = *reinterpret_cast<const float2*>(x + r * C + 2 * g);
            const float2 xhi = *reinterpret_cast<const float2*>(x + r * C + C2 + 2 * g);
            const float d[4] = {dlo.x * m, dlo.y * m, dhi.x * m, dhi.y * m};
            const float xv[4] = {xlo.x, xlo.y, xhi.x, xhi.y};
            float y[4], dy[4], o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) y[k] = (bs[k] + el[k] * xv[k]) * m;
#pragma unroll
            for (int k = 0; k < 4; ++k) dy[k] = (w[0 * 4 + k] * d[0] + w[1 * 4 + k] * d[1] + w[2 * 4 + k] * d[2] + w[3 * 4 + k] * d[3]) * m;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int k = 0; k < 4; ++k) accW[a * 4 + k] += d[a] * y[k];
#pragma unroll
            for (int k = 0; k < 4; ++k) { o[k] = dy[k] * el[k]; accL[k] += o[k] * xv[k]; accB[k] += dy[k]; }
            *reinterpret_cast<float2*>(dx + r * C + 2 * g) = make_float2(o[0], o[1]);
            if (nc.outs) {
                const float dl = nc.dld[r / nc.rows_per_utt];
                const float2 xb = *reinterpret_cast<const float2*>(nc.xmid + r * C + C2 + 2 * g);
                float ob[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int j = 2 * g + k, pc = (j >> 5) * 64 + (j & 31);
                    const float d = o[2 + k], e = expf(nc.outs[r * nc.ldo + pc + 32]);
                    const float dm = d * m, dlg = (d * e * (k ? xb.y : xb.x) + dl) * m;
                    if (nc.douts) { nc.douts[r * nc.ldo + pc] = dm; nc.douts[r * nc.ldo + pc + 32] = dlg; }
                    if (nc.douts16) { nc.douts16[r * nc.ldo + pc] = (__bf16)dm; nc.douts16[r * nc.ldo + pc + 32] = (__bf16)dlg; }
                    ob[k] = d * e * m;
                    for (int jp = C2 + j; jp < nc.ldo / 2; jp += C2) {       // pad slots of the PAIR packing: zero (see coupling_bwd_kernel)
                        const int pp = (jp >> 5) * 64 + (jp & 31);
                        if (nc.douts) { nc.douts[r * nc.ldo + pp] = 0.f; nc.douts[r * nc.ldo + pp + 32] = 0.f; }
                        if (nc.douts16) { nc.douts16[r * nc.ldo + pp] = (__bf16)0.f; nc.douts16[r * nc.ldo + pp + 32] = (__bf16)0.f; }
                    }
                }
                *reinterpret_cast<float2*>(dx + r * C + C2 + 2 * g) = make_float2(ob[0], ob[1]);
            } else {
                *reinterpret_cast<float2*>(dx + r * C + C2 + 2 * g) = make_float2(o[2], o[3]);
            }
        }
    }
    // block reduction through LDS: red[24][256]
    float* out = partial + (long)blockIdx.x * (2 * C + 16);
    for (int q = 0; q < 24; ++q) {
        float v = (q < 16) ? accW[q] : (q < 20 ? accL[q - 16] : accB[q - 20]);
        red[q * 256 + threadIdx.x] = (rl < tpg) ? v : 0.f;
    }
    __syncthreads();
    // dW: sum over all threads; dlogs/dbias: sum over threads with the same g
    {   // 16 lanes per entry, 16 values each, then a 16-wide butterfly
        const int q = threadIdx.x >> 4, part = threadIdx.x & 15;
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) s += red[q * 256 + t * 16 + part];
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) s += __shfl_xor(s, o, 16);
        if (part == 0) out[2 * C + q] = s;
    }
    for (int i = threadIdx.x; i < 2 * 4 * G; i += 256) {          // (which in {L,B}) x k x g
        const int which = i / (4 * G), k = (i / G) % 4, gg = i % G;
        float s = 0.f;
        for (int t = gg; t < tpg * G; t += G) s += red[(16 + which * 4 + k) * 256 + t];
        const int c = (k < 2) ? (2 * gg + k) : (C2 + 2 * gg + (k - 2));
        out[which * C + c] = s;
    }
}

// The same pass for C % 8 == 0 with 16-byte accesses: a thread owns the two mixing groups {4q..4q+3} u {C/2+4q..C/2+4q+3} of a row, so
// every load / store is a float4 (the 8-byte version above spent 19 us on 48 MB at the bench size).  Same per-block partial layout.
__global__ __launch_bounds__(256) void actnorm_inv_bwd4_kernel(const float* __restrict__ dz, float* __restrict__ dx,
                                                               const float* __restrict__ x, const float* __restrict__ logs,
                                                               const float* __restrict__ bias, const float* __restrict__ winfo,
                                                               const float* __restrict__ rowmask, float* __restrict__ partial,
                                                               long rows, int C, int rows_per_block, const NextCoupling nc)
{
    const int Q = C / 8, C2 = C / 2;
    extern __shared__ float red[];                 // [32][256]
    float w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = winfo[i];
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(rows, r0 + rows_per_block);
    const int tpq = 256 / Q;                       // row lanes per pass
    const int q = threadIdx.x % Q, rl = threadIdx.x / Q;
    float accW[16], accL[8], accB[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) accW[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { accL[i] = 0.f; accB[i] = 0.f; }
    // slot s of the thread: 0..3 = channels 4q..4q+3, 4..7 = C/2+4q..; group e (0, 1) = slots {2e, 2e+1, 4+2e, 5+2e}
    float el[8], bs[8];
    {
        const float4 la = *reinterpret_cast<const float4*>(logs + 4 * q), lb = *reinterpret_cast<const float4*>(logs + C2 + 4 * q);
        const float4 ba = *reinterpret_cast<const float4*>(bias + 4 * q), bb = *reinterpret_cast<const float4*>(bias + C2 + 4 * q);
        const float lv[8] = {la.x, la.y, la.z, la.w, lb.x, lb.y, lb.z, lb.w}, bv[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) { el[k] = expf(lv[k]); bs[k] = bv[k]; }
    }
    if (rl < tpq) {
        for (long r = r0 + rl; r < r1; r += tpq) {
            const float m = rowmask[r];
            const float4 dlo = *reinterpret_cast<const float4*>(dz + r * C + 4 * q), dhi = *reinterpret_cast<const float4*>(dz + r * C + C2 + 4 * q);
            const float4 xlo = *reinterpret_cast<const float4*>(x + r * C + 4 * q), xhi = *reinterpret_cast<const float4*>(x + r * C + C2 + 4 * q);
            float4 lg = make_float4(0.f, 0.f, 0.f, 0.f), xb = lg;
            float dl = 0.f;
            const int j0 = 4 * q, pc = (j0 >> 5) * 64 + (j0 & 31);
            if (nc.outs) {
                lg = *reinterpret_cast<const float4*>(nc.outs + r * nc.ldo + pc + 32);
                xb = *reinterpret_cast<const float4*>(nc.xmid + r * C + C2 + 4 * q);
                dl = nc.dld[r / nc.rows_per_utt];
            }
            const float dv[8] = {dlo.x * m, dlo.y * m, dlo.z * m, dlo.w * m, dhi.x * m, dhi.y * m, dhi.z * m, dhi.w * m};
            const float xv[8] = {xlo.x, xlo.y, xlo.z, xlo.w, xhi.x, xhi.y, xhi.z, xhi.w};
            float o[8];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int sl[4] = {2 * e, 2 * e + 1, 4 + 2 * e, 5 + 2 * e};
                float y[4], dy[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) y[k] = (bs[sl[k]] + el[sl[k]] * xv[sl[k]]) * m;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    dy[k] = (w[0 * 4 + k] * dv[sl[0]] + w[1 * 4 + k] * dv[sl[1]] + w[2 * 4 + k] * dv[sl[2]] + w[3 * 4 + k] * dv[sl[3]]) * m;
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int k = 0; k < 4; ++k) accW[a * 4 + k] += dv[sl[a]] * y[k];
#pragma unroll
                for (int k = 0; k < 4; ++k) { o[sl[k]] = dy[k] * el[sl[k]]; accL[sl[k]] += o[sl[k]] * xv[sl[k]]; accB[sl[k]] += dy[k]; }
            }
            *reinterpret_cast<float4*>(dx + r * C + 4 * q) = make_float4(o[0], o[1], o[2], o[3]);
            if (nc.outs) {
                const float lgv[4] = {lg.x, lg.y, lg.z, lg.w}, xbv[4] = {xb.x, xb.y, xb.z, xb.w};
                float dm[4], dg[4], ob[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float d = o[4 + k], e = expf(lgv[k]);
                    dm[k] = d * m; dg[k] = (d * e * xbv[k] + dl) * m; ob[k] = d * e * m;
                }
                float* dob = nc.douts ? nc.douts + r * nc.ldo : nullptr;      // (NULL: only the bf16 copy is kept - round 6, the fp32 rows had no reader on the bf16 path)
                if (dob) {
                    *reinterpret_cast<float4*>(dob + pc) = make_float4(dm[0], dm[1], dm[2], dm[3]);
                    *reinterpret_cast<float4*>(dob + pc + 32) = make_float4(dg[0], dg[1], dg[2], dg[3]);
                }
                typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
                if (nc.douts16) {
                    bf4 a, b;
#pragma unroll
                    for (int k = 0; k < 4; ++k) { a[k] = (__bf16)dm[k]; b[k] = (__bf16)dg[k]; }
                    *reinterpret_cast<bf4*>(nc.douts16 + r * nc.ldo + pc) = a;
                    *reinterpret_cast<bf4*>(nc.douts16 + r * nc.ldo + pc + 32) = b;
                }
                for (int jp = C2 + j0; jp < nc.ldo / 2; jp += C2) {      // pad slots of the PAIR packing: zero (see coupling_bwd_kernel)
                    const int pp = (jp >> 5) * 64 + (jp & 31);
                    if (dob) {
                        *reinterpret_cast<float4*>(dob + pp) = make_float4(0.f, 0.f, 0.f, 0.f);
                        *reinterpret_cast<float4*>(dob + pp + 32) = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                    if (nc.douts16) {
                        *reinterpret_cast<uint2*>(nc.douts16 + r * nc.ldo + pp) = make_uint2(0u, 0u);
                        *reinterpret_cast<uint2*>(nc.douts16 + r * nc.ldo + pp + 32) = make_uint2(0u, 0u);
                    }
                }
                *reinterpret_cast<float4*>(dx + r * C + C2 + 4 * q) = make_float4(ob[0], ob[1], ob[2], ob[3]);
            } else {
                *reinterpret_cast<float4*>(dx + r * C + C2 + 4 * q) = make_float4(o[4], o[5], o[6], o[7]);
            }
        }
    }
    float* out = partial + (long)blockIdx.x * (2 * C + 16);
    for (int i = 0; i < 32; ++i) {
        const float v = (i < 16) ? accW[i] : (i < 24 ? accL[i - 16] : accB[i - 24]);
        red[i * 256 + threadIdx.x] = (rl < tpq) ? v : 0.f;
    }
    __syncthreads();
    {   // dW: 16 lanes per entry, 16 values each, then a 16-wide butterfly
        const int e = threadIdx.x >> 4, part = threadIdx.x & 15;
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) s += red[e * 256 + t * 16 + part];
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) s += __shfl_xor(s, o, 16);
        if (part == 0) out[2 * C + e] = s;
    }
    for (int i = threadIdx.x; i < 2 * 8 * Q; i += 256) {            // (which in {L, B}) x slot x q: sum over the row lanes
        const int which = i / (8 * Q), sl = (i / Q) % 8, qq = i % Q;
        float s = 0.f;
        for (int t = qq; t < tpq * Q; t += Q) s += red[(16 + which * 8 + sl) * 256 + t];
        const int c = (sl < 4) ? (4 * qq + sl) : (C2 + 4 * qq + (sl - 4));
        out[which * C + c] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// log-determinant of the whole decoder (Modules.py:309 sum of the 3*F per-layer vectors):
//   logdet[b] = sum_f [ (sum_c logs_f[c] + logdet(W_f) * C/4) * len'_b + sum_{valid rows of b} sum_j logs^{coupling}_f ]
// stage 1: grid (F, B) -> part[f][b];  stage 2: fixed-order sum over f.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void logdet_partial_kernel(const float* __restrict__ outs_all, long flow_stride,
                                                              const float* __restrict__ logs_all, const float* __restrict__ winfo_all,
                                                              const float* __restrict__ rowmask, float* __restrict__ part,
                                                              int B, int Tp, int C, int ldo)
{
    // 16 wavefronts, one row per wavefront and pass: lane j reads the log-scale of pair slot j (PAIR packing: 32-wide runs at +32 of
    // every 64 columns), so no index arithmetic beyond shifts and every load is a 128-byte run
    __shared__ float red[3][16];
    const int f = blockIdx.x, b = blockIdx.y;
    const int C2 = C / 2, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float* outs = outs_all + (long)f * flow_stride + (long)b * Tp * ldo;
    const float* rm = rowmask + (long)b * Tp;
    float s = 0.f, len = 0.f, ls = 0.f;
    // four rows per pass, all loads issued before the first use (the row mask multiplies: no branch between them)
    for (int t0 = wave; t0 < Tp; t0 += 64) {
        float m[4], v[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = t0 + 16 * u;
            const bool in = t < Tp;
            m[u] = in ? rm[t] : 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int j = lane + 64 * h;
                v[u][h] = (in && j < C2) ? outs[(long)t * ldo + (j >> 5) * 64 + 32 + (j & 31)] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { len += m[u]; s += (v[u][0] + v[u][1]) * m[u]; }
    }
    for (int j = lane + 128; j < C2; j += 64)                       // C/2 > 128 (not the reference's shapes): remaining pair slots
        for (int t = wave; t < Tp; t += 16) s += outs[(long)t * ldo + (j >> 5) * 64 + 32 + (j & 31)] * rm[t];
    for (int c = threadIdx.x; c < C; c += 1024) ls += logs_all[f * C + c];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { s += __shfl_xor(s, o); ls += __shfl_xor(ls, o); }
    if (lane == 0) { red[0][wave] = s; red[1][wave] = len; red[2][wave] = ls; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float S = 0.f, L = 0.f, A = 0.f;
        for (int w = 0; w < 16; ++w) { S += red[0][w]; L += red[1][w]; A += red[2][w]; }
        part[f * B + b] = S + (A + winfo_all[f * 36 + 32] * (float)(C / 4)) * L;
    }
}
__global__ void logdet_final_kernel(const float* __restrict__ part, float* __restrict__ logdet, int F, int B)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float s = 0.f;
    for (int f = 0; f < F; ++f) s += part[f * B + b];
    logdet[b] = s;
}

inline int grid_for(long total, int per = 256, int cap = 2048) { long g = (total + per - 1) / per; return (int)(g > cap ? cap : (g < 1 ? 1 : g)); }
#define RET_LAUNCH() return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH

}  // namespace

extern "C" int glowtts_squeeze_rows(const float* mel, float* rows, float* rowmask, const int64_t* lengths,
                                    int B, int Cm, int Tm, int ns, void* stream)
{
    if (!mel || !rows || !lengths || B < 1 || Cm < 1 || ns < 1 || Tm < ns) return GLOWTTS_E_ARG;
    const int T = Tm / ns;
    const size_t lds = (size_t)Cm * (SQ_TS * ns + 1) * sizeof(float);
    hipLaunchKernelGGL(squeeze_kernel<true>, dim3((T + SQ_TS - 1) / SQ_TS, B), dim3(256), lds, static_cast<hipStream_t>(stream),
                       const_cast<float*>(mel), rows, rowmask, lengths, Cm, Tm, T, ns, 0.f, 0);
    RET_LAUNCH();
}

extern "C" int glowtts_unsqueeze_rows(const float* rows, float* mel, const int64_t* lengths,
                                      int B, int Cm, int Tm, int ns, int use_fill, float fill, void* stream)
{
    if (!mel || !rows || !lengths || B < 1 || Cm < 1 || ns < 1 || Tm < ns) return GLOWTTS_E_ARG;
    const int T = Tm / ns;
    const size_t lds = (size_t)Cm * (SQ_TS * ns + 1) * sizeof(float);
    hipLaunchKernelGGL(squeeze_kernel<false>, dim3((T + SQ_TS - 1) / SQ_TS, B), dim3(256), lds, static_cast<hipStream_t>(stream),
                       mel, const_cast<float*>(rows), (float*)nullptr, lengths, Cm, Tm, T, ns, fill, use_fill);
    RET_LAUNCH();
}

extern "C" int glowtts_inv1x1_prepare(const float* W, float* winfo, int F, void* stream)
{
    if (!W || !winfo || F < 1) return GLOWTTS_E_ARG;
    hipLaunchKernelGGL(inv4x4_kernel, dim3((F + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), W, winfo, F);
    RET_LAUNCH();
}

extern "C" int glowtts_actnorm_inv1x1(const float* xin, float* xout, const float* logs, const float* bias,
                                      const float* winfo, const float* rowmask, int64_t rows, int C, int reverse, void* stream)
{
    if (!xin || !xout || !logs || !bias || !winfo || !rowmask || rows < 1 || C < 4 || (C & 3)) return GLOWTTS_E_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int grid = grid_for(rows * (C / 4));
    if (reverse) hipLaunchKernelGGL(actnorm_inv_kernel<true>, dim3(grid), dim3(256), 0, s, xin, xout, logs, bias, winfo, rowmask, (long)rows, C, (float*)nullptr);
    else         hipLaunchKernelGGL(actnorm_inv_kernel<false>, dim3(grid), dim3(256), 0, s, xin, xout, logs, bias, winfo, rowmask, (long)rows, C, (float*)nullptr);
    RET_LAUNCH();
}

extern "C" int glowtts_actnorm_inv1x1_pass(const float* xin, float* xout, float* xpass, const float* logs, const float* bias, const float* winfo,
                                           const float* rowmask, int64_t rows, int C, void* stream)
{
    if (!xin || !xout || !xpass || !logs || !bias || !winfo || !rowmask || rows < 1 || C < 4 || (C & 3)) return GLOWTTS_E_ARG;
    hipLaunchKernelGGL(actnorm_inv_kernel<false>, dim3(grid_for(rows * (C / 4))), dim3(256), 0, static_cast<hipStream_t>(stream),
                       xin, xout, logs, bias, winfo, rowmask, (long)rows, C, xpass == xout ? (float*)nullptr : xpass);
    RET_LAUNCH();
}

extern "C" int glowtts_actnorm_inv1x1_pass_bf(const float* xin, float* xout, float* xpass, void* xa_bf, const float* logs, const float* bias, const float* winfo,
                                              const float* rowmask, int64_t rows, int C, void* stream)
{
    if (!xin || !xout || !logs || !bias || !winfo || !rowmask || rows < 1 || C < 4 || (C & 3)) return GLOWTTS_E_ARG;
    GLOWTTS_NOTE_STATIC("actnorm_inv1x1_pass%s", "");
    hipLaunchKernelGGL(actnorm_inv_kernel<false>, dim3(grid_for(rows * (C / 4))), dim3(256), 0, static_cast<hipStream_t>(stream),
                       xin, xout, logs, bias, winfo, rowmask, (long)rows, C, (xpass == xout || !xpass) ? (float*)nullptr : xpass, static_cast<uint32_t*>(xa_bf));
    RET_LAUNCH();
}

extern "C" int glowtts_actnorm_stats(const float* x, const float* rowmask, float* stats, float* scratch,
                                     int64_t rows, int C, void* stream)
{
    if (!x || !rowmask || !stats || !scratch || rows < 1 || C < 1) return GLOWTTS_E_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int rpb = 64;
    const int nblk = (int)((rows + rpb - 1) / rpb);
    hipLaunchKernelGGL(colstats_partial_kernel, dim3(nblk), dim3(256), 0, s, x, rowmask, scratch, (long)rows, C, rpb);
    const int n = 2 * C + 1;
    hipLaunchKernelGGL(colstats_final_kernel, dim3((n + 3) / 4), dim3(256), 0, s, scratch, stats, nblk, n, 0L, 0L);
    RET_LAUNCH();
}

// the float4 form needs C % 8 == 0, C / 8 <= 256 and 16-byte aligned rows (GLOWTTS_AN_WIDE=0 forces the 8-byte form)
static bool an_bwd_wide(int C, const float* dz, const float* dx, const float* x, int ldo)
{
    const int on = GLOWTTS_TUNABLE("GLOWTTS_AN_WIDE", 1);
    return on && (C % 8) == 0 && !((reinterpret_cast<uintptr_t>(dz) | reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(x)) & 15) && (ldo % 4) == 0;
}
// rows per block of actnorm_inv_bwd_kernel: small blocks (36 rows = three passes of the float4 kernel, 360 blocks at the bench size; 16..48
// measure the same within noise, 64 was 4 % of a step slower) keep every CU busy; the per-block partials are
// reduced later by colstats_final_kernel / glowtts_colsum_batched
static int an_bwd_rpb() { const int x = GLOWTTS_TUNABLE("GLOWTTS_AN_RPB", 36); return x < 16 ? 16 : x; }
extern "C" int64_t glowtts_actnorm_bwd_blocks(int64_t rows) { return (rows + an_bwd_rpb() - 1) / an_bwd_rpb(); }
extern "C" int64_t glowtts_actnorm_stats_scratch_floats(int64_t rows, int C) { return ((rows + 15) / 16) * (2 * (int64_t)C + 16); }

extern "C" int glowtts_actnorm_from_stats(const float* stats, float* logs, float* bias, int C, void* stream)
{
    if (!stats || !logs || !bias || C < 1) return GLOWTTS_E_ARG;
    hipLaunchKernelGGL(actnorm_from_stats_kernel, dim3((C + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), stats, logs, bias, C);
    RET_LAUNCH();
}

extern "C" int glowtts_coupling_bwd(float* dz, const float* xmid, const float* outs, float* douts, const float* rowmask,
                                    const float* dlogdet, int64_t rows, int C, int ldo, int rows_per_utt, void* stream)
{
    if (!dz || !xmid || !outs || !douts || !rowmask || !dlogdet || rows < 1 || C < 2) return GLOWTTS_E_ARG;
    if ((ldo & 63) || ldo < C) return GLOWTTS_E_ARG;
    hipLaunchKernelGGL(coupling_bwd_kernel, dim3(grid_for(rows * (ldo / 2))), dim3(256), 0, static_cast<hipStream_t>(stream),
                       dz, xmid, outs, douts, (__bf16*)nullptr, rowmask, dlogdet, (long)rows, C, ldo, rows_per_utt);
    RET_LAUNCH();
}

extern "C" int glowtts_coupling_bwd_bf16(float* dz, const float* xmid, const float* outs, float* douts, void* douts_bf16, const float* rowmask,
                                         const float* dlogdet, int64_t rows, int C, int ldo, int rows_per_utt, void* stream)
{
    if (!dz || !xmid || !outs || !douts_bf16 || !rowmask || !dlogdet || rows < 1 || C < 2) return GLOWTTS_E_ARG;       // (douts may be NULL: bf16 copy only)
    if ((ldo & 63) || ldo < C) return GLOWTTS_E_ARG;
    hipLaunchKernelGGL(coupling_bwd_kernel, dim3(grid_for(rows * (ldo / 2))), dim3(256), 0, static_cast<hipStream_t>(stream),
                       dz, xmid, outs, douts, static_cast<__bf16*>(douts_bf16), rowmask, dlogdet, (long)rows, C, ldo, rows_per_utt);
    RET_LAUNCH();
}

extern "C" int glowtts_fill_zero(float* p, int64_t n, void* stream)
{
    if (!p || n < 0) return GLOWTTS_E_ARG;
    if (n == 0) return GLOWTTS_OK;
    hipLaunchKernelGGL(zero_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), p, (long)n);
    RET_LAUNCH();
}

extern "C" int glowtts_actnorm_inv1x1_bwd(const float* dz, float* dx, const float* x, const float* logs, const float* bias,
                                          const float* winfo, const float* rowmask, float* param_grads /* [2C+16]: dlogs, dbias, dW */,
                                          float* scratch, int64_t rows, int C, void* stream)
{
    if (!dz || !dx || !x || !logs || !bias || !winfo || !rowmask || !scratch || rows < 1 || C < 4 || (C & 3) || C / 4 > 256) return GLOWTTS_E_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int rpb = an_bwd_rpb();
    const int nblk = (int)((rows + rpb - 1) / rpb);
    const NextCoupling none{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 1};
    if (an_bwd_wide(C, dz, dx, x, 0))
        hipLaunchKernelGGL(actnorm_inv_bwd4_kernel, dim3(nblk), dim3(256), 32 * 256 * sizeof(float), s, dz, dx, x, logs, bias, winfo, rowmask,
                           scratch, (long)rows, C, rpb, none);
    else
        hipLaunchKernelGGL(actnorm_inv_bwd_kernel, dim3(nblk), dim3(256), 24 * 256 * sizeof(float), s, dz, dx, x, logs, bias, winfo, rowmask,
                           scratch, (long)rows, C, rpb, none);
    const int n = 2 * C + 16;
    // param_grads == NULL: the caller reduces the per-block partials of all its flows later with glowtts_colsum_batched
    if (param_grads) hipLaunchKernelGGL(colstats_final_kernel, dim3((n + 3) / 4), dim3(256), 0, s, scratch, param_grads, nblk, n, 0L, 0L);
    RET_LAUNCH();
}

extern "C" int glowtts_actnorm_inv1x1_bwd_coupling(const float* dz, float* dx, const float* x, const float* logs, const float* bias, const float* winfo,
                                                   const float* rowmask, float* scratch, int64_t rows, int C,
                                                   const float* prev_xmid, const float* prev_outs, float* prev_douts, void* prev_douts_bf16,
                                                   const float* dlogdet, int ldo, int rows_per_utt, void* stream)
{
    if (!dz || !dx || !x || !logs || !bias || !winfo || !rowmask || !scratch || rows < 1 || C < 4 || (C & 3) || C / 4 > 256) return GLOWTTS_E_ARG;
    if (!prev_xmid || !prev_outs || (!prev_douts && !prev_douts_bf16) || !dlogdet || (ldo & 63) || ldo < C || rows_per_utt < 1) return GLOWTTS_E_ARG;
    const int rpb = an_bwd_rpb();
    const int nblk = (int)((rows + rpb - 1) / rpb);
    const NextCoupling nc{prev_xmid, prev_outs, prev_douts, static_cast<__bf16*>(prev_douts_bf16), dlogdet, ldo, rows_per_utt};
    const bool al = !((reinterpret_cast<uintptr_t>(prev_xmid) | reinterpret_cast<uintptr_t>(prev_outs) | reinterpret_cast<uintptr_t>(prev_douts)) & 15) &&
                    !(reinterpret_cast<uintptr_t>(prev_douts_bf16) & 7);
    if (al && an_bwd_wide(C, dz, dx, x, ldo))
        hipLaunchKernelGGL(actnorm_inv_bwd4_kernel, dim3(nblk), dim3(256), 32 * 256 * sizeof(float), static_cast<hipStream_t>(stream), dz, dx, x, logs, bias,
                           winfo, rowmask, scratch, (long)rows, C, rpb, nc);
    else
        hipLaunchKernelGGL(actnorm_inv_bwd_kernel, dim3(nblk), dim3(256), 24 * 256 * sizeof(float), static_cast<hipStream_t>(stream), dz, dx, x, logs, bias,
                           winfo, rowmask, scratch, (long)rows, C, rpb, nc);
    RET_LAUNCH();
}

extern "C" int glowtts_colsum_batched(const float* partial, float* out, int nrows, int n, int batch, int64_t part_stride, int64_t out_stride, void* stream)
{
    if (!partial || !out || nrows < 1 || n < 1 || batch < 1) return GLOWTTS_E_ARG;
    hipLaunchKernelGGL(colstats_final_kernel, dim3((n + 3) / 4, batch), dim3(256), 0, static_cast<hipStream_t>(stream), partial, out, nrows, n,
                       (long)part_stride, (long)out_stride);
    RET_LAUNCH();
}

// Parameter gradients of ActNorm / invertible 1x1 from the reduced data terms d_an [F][2C+16] plus the log-determinant terms
// (Modules.py:694, 747: logdet_b += (sum logs + logdet(W) C/4) * len_b):  s = sum_b dlogdet[b] * len_b,
//   dlogs[f][c] = d_an[f][c] + s,   dbias[f][c] = d_an[f][C+c],   dW[f] = d_an[f][2C..2C+16) + s (C/4) (W_f^-1)^T
// One small launch instead of ~10 PyTorch kernels at the very end of the decoder's backward chain.
__global__ __launch_bounds__(256) void decoder_param_grads_kernel(const float* __restrict__ d_an, const float* __restrict__ dlogdet,
                                                                  const float* __restrict__ rowmask, const float* __restrict__ winfo,
                                                                  float* __restrict__ dlogs, float* __restrict__ dbias, float* __restrict__ dw,
                                                                  int F, int B, int Tp, int C)
{
    __shared__ float red[256];
    float s = 0.f;
    // every block recomputes s (B * Tp floats: trivial) - no second launch.  Eight utterances x up to four loads per thread are issued together, unconditionally
    // (clamped address + select): as rolled loops every load was followed by s_waitcnt vmcnt(0) - 64 exposed round trips, 28 us for this launch, the last of the
    // decoder's chain in front of the gradient norm.  The additions keep their order (a missing element adds 0).
    if (Tp <= 1024) {
        for (int b0 = 0; b0 < B; b0 += 8) {
            float m[8][4], dl[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = b0 + u < B ? b0 + u : B - 1;
                dl[u] = dlogdet[b];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int t = threadIdx.x + 256 * k;
                    const float v = rowmask[(long)b * Tp + (t < Tp ? t : Tp - 1)];
                    m[u][k] = t < Tp ? v : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float len = (((0.f + m[u][0]) + m[u][1]) + m[u][2]) + m[u][3];
                if (b0 + u < B) s += len * dl[u];
            }
        }
    } else {
        for (int b = 0; b < B; ++b) {
            float len = 0.f;
            for (int t = threadIdx.x; t < Tp; t += 256) len += rowmask[(long)b * Tp + t];
            s += len * dlogdet[b];
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    s = red[0];
    const int f = blockIdx.x, n = 2 * C + 16;
    const float* d = d_an + (long)f * n;
    for (int c = threadIdx.x; c < C; c += 256) { dlogs[(long)f * C + c] = d[c] + s; dbias[(long)f * C + c] = d[C + c]; }
    if (threadIdx.x < 16) {
        const int i = threadIdx.x >> 2, j = threadIdx.x & 3;
        dw[f * 16 + threadIdx.x] = d[2 * C + threadIdx.x] + s * (0.25f * C) * winfo[f * 36 + 16 + j * 4 + i];       // (W^-1)^T[i][j] = W^-1[j][i]
    }
}

extern "C" int glowtts_decoder_param_grads(const float* d_an, const float* dlogdet, const float* rowmask, const float* winfo,
                                           float* dlogs, float* dbias, float* dw, int F, int B, int Tp, int C, void* stream)
{
    if (!d_an || !dlogdet || !rowmask || !winfo || !dlogs || !dbias || !dw || F < 1 || B < 1 || Tp < 1 || C < 1) return GLOWTTS_E_ARG;
    hipLaunchKernelGGL(decoder_param_grads_kernel, dim3(F), dim3(256), 0, static_cast<hipStream_t>(stream), d_an, dlogdet, rowmask, winfo, dlogs, dbias, dw, F, B, Tp, C);
    RET_LAUNCH();
}

extern "C" int glowtts_decoder_logdet(const float* outs_all, int64_t flow_stride, const float* logs_all, const float* winfo_all,
                                      const float* rowmask, float* part, float* logdet, int F, int B, int Tp, int C, int ldo, void* stream)
{
    if (!outs_all || !logs_all || !winfo_all || !rowmask || !part || !logdet || F < 1 || B < 1) return GLOWTTS_E_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(logdet_partial_kernel, dim3(F, B), dim3(1024), 0, s, outs_all, (long)flow_stride, logs_all, winfo_all, rowmask, part, B, Tp, C, ldo);
    hipLaunchKernelGGL(logdet_final_kernel, dim3((B + 255) / 256), dim3(256), 0, s, part, logdet, F, B);
    RET_LAUNCH();
}

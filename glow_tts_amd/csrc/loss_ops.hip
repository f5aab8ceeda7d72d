// Alignment expansion and likelihood loss of the Glow-TTS training graph for gfx950:
//   expand   mel_Mean = mean @ attentions, mel_Log_Std = log_Std @ attentions (Modules.py:120-121): attentions has exactly one 1
//            per valid frame, so the batched matmul is a gather by the MAS token index; its gradient is a segment sum.
//   targets  log_Duration_Targets = log(sum_t attentions + 1e-7) * token_mask (Modules.py:122): run lengths of the index.
//   mle      MLE_Loss (Modules.py:1020-1029) and its gradient.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/glowtts_hip.h"
#include "launch_log.h"

namespace {

// src [B][C][Tx] -> out [B][C][Ty], out[b][c][y] = idx[b][y] >= 0 ? src[b][c][idx[b][y]] : 0
__global__ __launch_bounds__(256) void expand_fwd_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx, float* __restrict__ out,
                                                         int C, int Tx, int Ty)
{
    const int b = blockIdx.z, c = blockIdx.y;
    const float* s = src + ((long)b * C + c) * Tx;
    float* o = out + ((long)b * C + c) * Ty;
    const int32_t* ib = idx + (long)b * Ty;
    for (int y = blockIdx.x * 256 + threadIdx.x; y < Ty; y += gridDim.x * 256) { const int x = ib[y]; o[y] = x >= 0 ? s[x] : 0.f; }
}
// dsrc[b][c][x] = sum_{y : idx[b][y] == x} dout[b][c][y].  idx is non-decreasing over the valid frames, so the frames of a token are a
// contiguous run.  One wavefront per (b, c) row: 64 frames at a time, a segmented (keyed by token) inclusive scan over the lanes, and
// the lane at the end of each run adds the run's sum to its token's LDS slot.  Cost is independent of how skewed the durations are
// (MAS on untrained weights gives runs of hundreds of frames), the order of the additions is fixed: deterministic.
__global__ __launch_bounds__(256) void expand_bwd_kernel(const float* __restrict__ dout, const int32_t* __restrict__ idx, float* __restrict__ dsrc,
                                                         int C, int Tx, int Ty)
{
    extern __shared__ float sm_acc[];                    // [4 waves][Tx]
    const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + wave;
    if (c >= C) return;                                  // (no workgroup barrier below)
    float* acc = sm_acc + wave * Tx;
    for (int x = lane; x < Tx; x += 64) acc[x] = 0.f;
    const int32_t* ib = idx + (long)b * Ty;
    const float* d = dout + ((long)b * C + c) * Ty;
    for (int y0 = 0; y0 < Ty; y0 += 64) {
        const int y = y0 + lane;
        float v = y < Ty ? d[y] : 0.f;
        const int x = y < Ty ? ib[y] : -1;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float vv = __shfl_up(v, o);
            const int xx = __shfl_up(x, o);
            if (lane >= o && xx == x) v += vv;
        }
        const int xn = __shfl_down(x, 1);
        if (x >= 0 && (lane == 63 || xn != x)) acc[x] += v;          // one lane per token and pass; LDS accesses of a wave complete in order
    }
    __builtin_amdgcn_wave_barrier();
    for (int x = lane; x < Tx; x += 64) dsrc[((long)b * C + c) * Tx + x] = acc[x];
}
// target[b][x] = log(count_x + 1e-7) * (x < t_x[b])
__global__ __launch_bounds__(256) void dur_target_kernel(const int32_t* __restrict__ idx, const int64_t* __restrict__ t_x, float* __restrict__ out, int Tx, int Ty)
{
    extern __shared__ int cnt[];
    const int b = blockIdx.x;
    for (int x = threadIdx.x; x < Tx; x += 256) cnt[x] = 0;
    __syncthreads();
    const int32_t* ib = idx + (long)b * Ty;
    for (int y = threadIdx.x; y < Ty; y += 256) { const int x = ib[y]; if (x >= 0) atomicAdd(&cnt[x], 1); }    // integer LDS atomics: exact
    __syncthreads();
    const int tx = (int)t_x[b];
    for (int x = threadIdx.x; x < Tx; x += 256) out[(long)b * Tx + x] = (x < tx) ? logf((float)cnt[x] + 1e-7f) : 0.f;
}

// MLE_Loss (Modules.py:1025): partial sums of  log_std + 0.5 * exp(-2 log_std) * (z - mean)^2  (two-stage, deterministic)
template <bool PUBLISH>
__device__ __forceinline__ void mle_partial_block(const float* __restrict__ z, const float* __restrict__ mean, const float* __restrict__ ls,
                                                  float* __restrict__ partial, long n, int blk, int nblk, float* red)
{
    float acc = 0.f;
    for (long i = blk * 256L + threadIdx.x; i < n; i += (long)nblk * 256) {
        const float d = z[i] - mean[i];
        acc += ls[i] + 0.5f * expf(-2.f * ls[i]) * d * d;
    }
    red[threadIdx.x] = acc; __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; __syncthreads(); }
    if (threadIdx.x == 0) {
        if (PUBLISH) {                                         // a RETURNING device-scope exchange: when its value is back it has been performed (see prior_loss_kernel)
            const float old = atomicExch(partial + blk, red[0]);
            asm volatile("s_waitcnt vmcnt(0)" :: "v"(old) : "memory");
        } else partial[blk] = red[0];
    }
}
__global__ __launch_bounds__(256) void mle_partial_kernel(const float* __restrict__ z, const float* __restrict__ mean, const float* __restrict__ ls,
                                                          float* __restrict__ partial, long n)
{
    __shared__ float red[256];
    mle_partial_block<false>(z, mean, ls, partial, n, blockIdx.x, gridDim.x, red);
}
// 1 / (sum(len // ns) * ns * mel_dim): the denominator of the loss (Modules.py:1026) - a function of the lengths alone
__device__ __forceinline__ double mle_denominator(const int64_t* __restrict__ lengths, int B, int ns, int mel_dim)
{
    long frames = 0;
    for (int i = 0; i < B; ++i) frames += (lengths[i] / ns) * ns;
    return (double)frames * mel_dim;
}
// loss = (sum partial - sum logdet) / (sum(len // ns) * ns * mel_dim) + 0.5 log(2 pi);  also writes 1/denominator for the backward
// COHERENT: `partial` was published by other workgroups of the SAME launch (prior_loss_kernel's last workgroup): read back by device-scope atomics
template <bool COHERENT>
__device__ __forceinline__ void mle_final_block(const float* __restrict__ partial, int nblk, const float* __restrict__ logdet, const int64_t* __restrict__ lengths,
                                                int B, int ns, int mel_dim, float* __restrict__ loss, float* __restrict__ inv_denom, double* red)
{
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256)
        acc += COHERENT ? __uint_as_float(atomicOr(reinterpret_cast<unsigned int*>(const_cast<float*>(partial)) + i, 0u)) : partial[i];
    for (int i = threadIdx.x; i < B; i += 256) acc -= logdet[i];
    red[threadIdx.x] = acc; __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; __syncthreads(); }
    if (threadIdx.x == 0) {
        const double den = mle_denominator(lengths, B, ns, mel_dim);
        *loss = (float)(red[0] / den + 0.5 * 1.8378770664093453);
        *inv_denom = (float)(1.0 / den);
    }
}
__global__ __launch_bounds__(256) void mle_final_kernel(const float* __restrict__ partial, int nblk, const float* __restrict__ logdet, const int64_t* __restrict__ lengths,
                                                        int B, int ns, int mel_dim, float* __restrict__ loss, float* __restrict__ inv_denom)
{
    __shared__ double red[256];
    mle_final_block<false>(partial, nblk, logdet, lengths, B, ns, mel_dim, loss, inv_denom, red);
}
// gradients: dz = g e (z - m), dmean = -dz, dls = g (1 - e (z - m)^2), with e = exp(-2 ls), g = dloss / denom
__global__ __launch_bounds__(256) void mle_bwd_kernel(const float* __restrict__ z, const float* __restrict__ mean, const float* __restrict__ ls,
                                                      const float* __restrict__ dloss, const float* __restrict__ inv_denom,
                                                      float* __restrict__ dz, float* __restrict__ dmean, float* __restrict__ dls, long n,
                                                      float* __restrict__ dlogdet, int B)
{
    const float g = dloss[0] * inv_denom[0];
    if (dlogdet && blockIdx.x == 0) for (int b = threadIdx.x; b < B; b += 256) dlogdet[b] = -g;          // d loss / d log_dets[b]  (Modules.py:1025-1027)
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float d = z[i] - mean[i], e = expf(-2.f * ls[i]);
        const float t = g * e * d;
        dz[i] = t; dmean[i] = -t; dls[i] = g * (1.f - e * d * d);
    }
}

// Round 6.  Both expansions (Modules.py:120-121), the duration targets (:122) and - path != NULL - the dense attentions (:116) in ONE launch: blocks [0, nexp)
// sweep 16-byte groups of the two outputs (expand_fwd4_kernel's pattern, one int4 of token indices serves both rows), blocks [nexp, nexp + npath) write the
// 0/1 matrix, the last B blocks count the run lengths (dur_target_kernel).
__global__ __launch_bounds__(256) void expand_pair_kernel(const float* __restrict__ mean, const float* __restrict__ ls, const int32_t* __restrict__ idx,
                                                          const int64_t* __restrict__ t_x, float* __restrict__ omean, float* __restrict__ ols,
                                                          float* __restrict__ targets, int C, int Tx, int Ty, unsigned int total4, int nexp,
                                                          float* __restrict__ path, unsigned int ptotal4, int npath)
{
    extern __shared__ int cnt[];
    if ((int)blockIdx.x >= nexp + npath) {
        const int b = blockIdx.x - nexp - npath;
        for (int x = threadIdx.x; x < Tx; x += 256) cnt[x] = 0;
        __syncthreads();
        const int32_t* ib = idx + (long)b * Ty;
        for (int y = threadIdx.x; y < Ty; y += 256) { const int x = ib[y]; if (x >= 0) atomicAdd(&cnt[x], 1); }    // integer LDS atomics: exact
        __syncthreads();
        const int tx = (int)t_x[b];
        for (int x = threadIdx.x; x < Tx; x += 256) targets[(long)b * Tx + x] = (x < tx) ? logf((float)cnt[x] + 1e-7f) : 0.f;
        return;
    }
    const unsigned int q = (unsigned int)Ty >> 2;
    if ((int)blockIdx.x >= nexp) {
        // the dense 0/1 attentions [B][Tx][Ty] (Modules.py:116; only RETURNED by GlowTTS.forward): mas.hip's mas_path_linear_kernel, same bytes
        for (unsigned int i = (blockIdx.x - (unsigned int)nexp) * 256u + threadIdx.x; i < ptotal4; i += (unsigned int)npath * 256u) {
            const unsigned int row = i / q, y4 = i - row * q;      // row = b * Tx + x
            const unsigned int b = row / (unsigned int)Tx, x = row - b * (unsigned int)Tx;
            const int4 id = *reinterpret_cast<const int4*>(idx + (size_t)b * Ty + (size_t)y4 * 4);
            float4 v;
            v.x = id.x == (int)x ? 1.f : 0.f; v.y = id.y == (int)x ? 1.f : 0.f; v.z = id.z == (int)x ? 1.f : 0.f; v.w = id.w == (int)x ? 1.f : 0.f;
            *reinterpret_cast<float4*>(path + (size_t)i * 4) = v;
        }
        return;
    }
    for (unsigned int i = blockIdx.x * 256u + threadIdx.x; i < total4; i += (unsigned int)nexp * 256u) {
        const unsigned int row = i / q, y4 = i - row * q;          // row = b * C + c
        const unsigned int b = row / (unsigned int)C;
        const int4 id = *reinterpret_cast<const int4*>(idx + (size_t)b * Ty + (size_t)y4 * 4);
        const float* sm = mean + (size_t)row * Tx;
        const float* sl = ls + (size_t)row * Tx;
        float4 v, w;
        v.x = id.x >= 0 ? sm[id.x] : 0.f; v.y = id.y >= 0 ? sm[id.y] : 0.f; v.z = id.z >= 0 ? sm[id.z] : 0.f; v.w = id.w >= 0 ? sm[id.w] : 0.f;
        w.x = id.x >= 0 ? sl[id.x] : 0.f; w.y = id.y >= 0 ? sl[id.y] : 0.f; w.z = id.z >= 0 ? sl[id.z] : 0.f; w.w = id.w >= 0 ? sl[id.w] : 0.f;
        *reinterpret_cast<float4*>(omean + (size_t)i * 4) = v;
        *reinterpret_cast<float4*>(ols + (size_t)i * 4) = w;
    }
}

// Round 6.  The backward of MLE_Loss (Modules.py:1020-1029) THROUGH the expansion (:120-121) in one launch: d z per frame, and the gradients of the
// TOKEN-space mean / log_std as sums over each token's frames - a monotonic alignment gives every token a contiguous run, so the expanded gradients
// (2 x 8 MB at B = 32) are never written and the two expand_bwd passes (2 x 56 us in the step, at the head of the text encoder's backward) disappear.
// One wavefront per (utterance, channel) row, 64 frames per pass, the token-keyed segmented scan of expand_bwd_kernel on both sums at once; the values
// and the order of every addition are those of mle_bwd_kernel + expand_bwd_kernel: the results are the same bits.
__device__ __forceinline__ void prior_bwd_block(const float* __restrict__ z, const float* __restrict__ mean, const float* __restrict__ ls,
                                                const int32_t* __restrict__ idx, const float g, float* __restrict__ dz, float* __restrict__ dmean,
                                                float* __restrict__ dls, float* __restrict__ dlogdet, int B, int C, int Tx, int Ty, int bx, int b, float* pl_sm)
{
    // pl_sm per wave: m [Tx], e [Tx], acc1 [Tx], acc2 [Tx]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (dlogdet && bx == 0 && b == 0) for (int i = threadIdx.x; i < B; i += 256) dlogdet[i] = -g;      // Modules.py:1025-1027
    const int c = bx * 4 + wave;
    if (c >= C) return;                                  // (no workgroup barrier below)
    float* m = pl_sm + (size_t)wave * 4 * Tx;
    float* e = m + Tx;
    float* a1 = e + Tx;
    float* a2 = a1 + Tx;
    const long trow = ((long)b * C + c) * Tx, frow = ((long)b * C + c) * Ty;
    for (int x = lane; x < Tx; x += 64) { m[x] = mean[trow + x]; e[x] = expf(-2.f * ls[trow + x]); a1[x] = 0.f; a2[x] = 0.f; }
    __builtin_amdgcn_wave_barrier();
    const int32_t* ib = idx + (long)b * Ty;
    // (the next pass's two loads are issued before this pass's scan: the 13 passes of a row were 13 exposed round trips)
    int xn_ = lane < Ty ? ib[lane] : -1;
    float zn_ = lane < Ty ? z[frow + lane] : 0.f;
    for (int y0 = 0; y0 < Ty; y0 += 64) {
        const int y = y0 + lane;
        const int x = xn_;
        const float zz = zn_;
        {
            const int yn = y + 64;
            xn_ = yn < Ty ? ib[yn] : -1;
            zn_ = yn < Ty ? z[frow + yn] : 0.f;
        }
        // frames outside the alignment read mean = log_std = 0 from the expansion (Modules.py:120-121 multiply by an all-zero column)
        const float mm = x >= 0 ? m[x] : 0.f, ee = x >= 0 ? e[x] : 1.f;
        const float d = zz - mm;
        const float t = g * ee * d;
        if (y < Ty) dz[frow + y] = t;
        float v1 = x >= 0 ? -t : 0.f;
        float v2 = x >= 0 ? g * (1.f - ee * d * d) : 0.f;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float u1 = __shfl_up(v1, o), u2 = __shfl_up(v2, o);
            const int xx = __shfl_up(x, o);
            if (lane >= o && xx == x) { v1 += u1; v2 += u2; }
        }
        const int xn = __shfl_down(x, 1);
        if (x >= 0 && (lane == 63 || xn != x)) { a1[x] += v1; a2[x] += v2; }     // one lane per token and pass; LDS accesses of a wave complete in order
    }
    __builtin_amdgcn_wave_barrier();
    for (int x = lane; x < Tx; x += 64) { dmean[trow + x] = a1[x]; dls[trow + x] = a2[x]; }
}
__global__ __launch_bounds__(256) void prior_loss_bwd_kernel(const float* __restrict__ z, const float* __restrict__ mean, const float* __restrict__ ls,
                                                             const int32_t* __restrict__ idx, const float* __restrict__ dloss,
                                                             const float* __restrict__ inv_denom, float* __restrict__ dz, float* __restrict__ dmean,
                                                             float* __restrict__ dls, float* __restrict__ dlogdet, int B, int C, int Tx, int Ty)
{
    extern __shared__ float pl_sm[];
    prior_bwd_block(z, mean, ls, idx, dloss[0] * inv_denom[0], dz, dmean, dls, dlogdet, B, C, Tx, Ty, blockIdx.x, blockIdx.y, pl_sm);
}

// Round 6.  MLE_Loss on the expanded prior, forward AND the gradients for d loss = dloss[0] (the caller passes the constant 1 its backward will be seeded with),
// in ONE launch: workgroups [0, nblk) are mle_partial_kernel's, the next cg * B are prior_loss_bwd_kernel's (their scale 1 / denominator depends on the
// lengths alone, not on the sum), and the workgroup that finishes LAST (a device counter, reset for the next launch) runs mle_final_kernel's reduction.
// Same values in the same order as the three launches: the same bits.  What it buys: the chain between the alignment search and the flow decoder's
// backward is two dependent launches shorter, and the 15-us gradient pass runs beside the 14-us reduction.
__global__ __launch_bounds__(256) void prior_loss_kernel(const float* __restrict__ z, const float* __restrict__ mel_mean, const float* __restrict__ mel_ls,
                                                         const float* __restrict__ mean, const float* __restrict__ ls, const int32_t* __restrict__ idx,
                                                         const float* __restrict__ logdet, const int64_t* __restrict__ lengths, const float* __restrict__ dloss,
                                                         float* __restrict__ partial, unsigned int* __restrict__ counter, float* __restrict__ loss,
                                                         float* __restrict__ inv_denom, float* __restrict__ dz, float* __restrict__ dmean,
                                                         float* __restrict__ dls, float* __restrict__ dlogdet, long n, int nblk, int cg, int B, int C, int Tx,
                                                         int Ty, int ns, int mel_dim)
{
    extern __shared__ float pl_sm[];
    __shared__ double redd[256];
    __shared__ int last;
    if ((int)blockIdx.x >= nblk) {
        // (the gradient workgroups publish nothing: their outputs are read by later launches only)
        const int w = blockIdx.x - nblk;
        if (threadIdx.x == 0) redd[0] = 1.0 / mle_denominator(lengths, B, ns, mel_dim);
        __syncthreads();
        const float inv = (float)redd[0];
        prior_bwd_block(z, mean, ls, idx, dloss[0] * inv, dz, dmean, dls, dlogdet, B, C, Tx, Ty, w % cg, w / cg, pl_sm);
        return;
    }
    // Cross-workgroup hand-over WITHOUT a release fence: on this chip a device-scope release writes back the whole L2 of the XCD (eight L2s, one agent) - with
    // 8 MB of d z dirty in it, once per workgroup, that cost the step more than the two launches it saves (measured: + 70 us).  Device-scope atomic
    // read-modify-writes are performed at the memory side: the partial sum goes out as an atomic exchange, the wave waits for its return, then counts itself;
    // the workgroup that counts last reads the partial sums back with atomic ORs of zero.
    mle_partial_block<true>(z, mel_mean, mel_ls, partial, n, blockIdx.x, nblk, reinterpret_cast<float*>(redd));
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        last = atomicAdd(counter, 1u) == (unsigned int)nblk - 1;
    }
    __syncthreads();
    if (!last) return;
    mle_final_block<true>(partial, nblk, logdet, lengths, B, ns, mel_dim, loss, inv_denom, redd);
    if (threadIdx.x == 0) atomicExch(counter, 0u);
}

// Duration loss (Train.py:203-211: MSELoss(log_Durations, log_Duration_Targets), mean over the padded [B, 1, T_tok]) and its gradient: one workgroup, one launch
// per direction (torch: sub / pow / mean forward, three more backward, each a launch on the encoder stream's chain).  scale = 1 / elements (or the caller's own
// normaliser); n up to a few thousand.
// scale: 1 / denominator given by the host, or - lengths != NULL - 1 / (B * max(lengths)) (the trainer's mean over the batch's own longest text), or - extent != NULL -
// 1 / (B * extent[0]) (data parallel: the longest text of the GLOBAL batch, a device scalar)
__device__ __forceinline__ float mse_scale(float scale, const int64_t* lengths, int B, const float* extent)
{
    if (extent) return 1.f / ((float)B * extent[0]);
    if (lengths) { long mx = 1; for (int i = 0; i < B; ++i) mx = lengths[i] > mx ? lengths[i] : mx; return 1.f / ((float)B * (float)mx); }
    return scale;
}
// da_unit != NULL: also the gradient for d loss = 1 (mse_bwd_kernel's value for that seed, same bits) - the caller's backward returns it without a launch
// when it is seeded with the constant 1 (alignment.DurationMSE)
__global__ __launch_bounds__(256) void mse_fwd_kernel(const float* __restrict__ a, const float* __restrict__ t, float* __restrict__ loss, long n, float scale,
                                                      const int64_t* __restrict__ lengths, int B, const float* __restrict__ extent, float* __restrict__ da_unit)
{
    __shared__ double red[256];
    double acc = 0.0;
    const float sc = mse_scale(scale, lengths, B, extent);
    const float g = 2.f * sc * 1.f;
#pragma unroll 4
    for (long i = threadIdx.x; i < n; i += 256) { const float d = a[i] - t[i]; acc += (double)d * d; if (da_unit) da_unit[i] = g * d; }
    red[threadIdx.x] = acc; __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; __syncthreads(); }
    if (threadIdx.x == 0) *loss = (float)(red[0] * (double)sc);
}
__global__ __launch_bounds__(256) void mse_bwd_kernel(const float* __restrict__ a, const float* __restrict__ t, const float* __restrict__ dloss,
                                                      float* __restrict__ da, long n, float scale, const int64_t* __restrict__ lengths, int B,
                                                      const float* __restrict__ extent)
{
    const float g = 2.f * mse_scale(scale, lengths, B, extent) * dloss[0];
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) da[i] = g * (a[i] - t[i]);
}

#define RET_LAUNCH() return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH
}  // namespace

// Operands of the log-prior GEMM (Modules.py:108-114) from the encoder's mean / log_std [B][Cm][Tx] in ONE launch (was 15 small
// PyTorch kernels between the decoder and MAS, on the step's critical path):
//   packed  [b][k chunk][n = token, padded to npad][16]  fp32 MFMA weight image of (sigma^-2 | mu sigma^-2)      (K = 2 Cm, taps = 1)
//   cb      [b][token]   = sum_c (-0.5 log 2pi - log_std - 0.5 mu^2 sigma^-2)
//   fmask   [b][frame]   = frame < mel_length ;  tx32 / ty32 = the lengths as int32 (MAS takes int32)
__global__ __launch_bounds__(256) void logprior_prep_kernel(const float* __restrict__ mean, const float* __restrict__ ls, const int64_t* __restrict__ tlen,
                                                            const int64_t* __restrict__ mlen, float* __restrict__ packed, float* __restrict__ cb,
                                                            float* __restrict__ fmask, int32_t* __restrict__ tx32, int32_t* __restrict__ ty32,
                                                            int B, int Cm, int Tx, int Ty, int npad, int kch, int ns)
{
    const long tid = blockIdx.x * 256L + threadIdx.x, nth = (long)gridDim.x * 256;
    const long per = (long)kch * npad * 16;
    for (long i = tid; i < B * per; i += nth) {
        const int b = (int)(i / per);
        const long r = i - b * per;
        const int kk = (int)(r & 15), n = (int)((r >> 4) % npad), kc = (int)((r >> 4) / npad);
        const int k = kc * 16 + kk;
        float v = 0.f;
        if (n < Tx && k < 2 * Cm) {
            const int c = k < Cm ? k : k - Cm;
            const long src = ((long)b * Cm + c) * Tx + n;
            const float rr = expf(-2.f * ls[src]);
            v = k < Cm ? rr : mean[src] * rr;
        }
        packed[i] = v;
    }
    for (long i = tid; i < (long)B * Tx; i += nth) {                    // consecutive threads = consecutive tokens: coalesced over x
        const int b = (int)(i / Tx), x = (int)(i - (long)b * Tx);
        float acc = 0.f;
        for (int c = 0; c < Cm; ++c) {
            const long src = ((long)b * Cm + c) * Tx + x;
            const float l = ls[src], m = mean[src];
            acc += -0.9189385332046727f - l - 0.5f * m * m * expf(-2.f * l);
        }
        cb[i] = acc;
    }
    for (long i = tid; i < (long)B * Ty; i += nth) { const int b = (int)(i / Ty); fmask[i] = (i - (long)b * Ty) < (mlen[b] / ns) * ns ? 1.f : 0.f; }
    for (long i = tid; i < B; i += nth) { tx32[i] = (int32_t)tlen[i]; ty32[i] = (int32_t)((mlen[i] / ns) * ns); }
}

// Tiled form of logprior_prep_kernel for 2 Cm * 65 * 4 bytes <= 64 KiB of LDS (the reference's 80 mel channels: 41 KiB): a workgroup owns
// 64 tokens of one utterance, reads mean / log_std once (coalesced over tokens), keeps (sigma^-2 | mu sigma^-2) in LDS, reduces the
// per-token constant over four channel groups and writes its 64 x 16 runs of every K chunk contiguously.  ~5 us instead of 33.
__global__ __launch_bounds__(256) void logprior_prep_tile_kernel(const float* __restrict__ mean, const float* __restrict__ ls, const int64_t* __restrict__ tlen,
                                                                 const int64_t* __restrict__ mlen, float* __restrict__ packed, float* __restrict__ cb,
                                                                 float* __restrict__ fmask, int32_t* __restrict__ tx32, int32_t* __restrict__ ty32,
                                                                 int B, int Cm, int Tx, int Ty, int npad, int kch, int ns)
{
    extern __shared__ float lp_tile[];                 // [2 Cm][65] operand values, then [4][64] partial constants
    const int b = blockIdx.y, n0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int x = n0 + lane;
    float* part = lp_tile + 2 * Cm * 65;
    float acc = 0.f;
    // wave `grp` takes channels grp, grp + 4, ...: loads coalesced over the 64 tokens.  Round 5: every load of a block of LP_CH channels is issued before the
    // first use, unconditionally (clamped addresses) - the predicated one-channel-at-a-time loop was a chain of Cm / 4 dependent round trips to memory (15 us
    // alone, 45 us with another stream's HBM traffic beside it, on the chain between the decoder's forward and the alignment search)
    constexpr int LP_CH = 8;
    const int xs = x < Tx ? x : Tx - 1;
    for (int c0 = grp; c0 < Cm; c0 += 4 * LP_CH) {
        float lv[LP_CH], mv[LP_CH];
#pragma unroll
        for (int i = 0; i < LP_CH; ++i) {
            const int c = c0 + 4 * i;
            const long src = ((long)b * Cm + (c < Cm ? c : Cm - 1)) * Tx + xs;
            lv[i] = ls[src]; mv[i] = mean[src];
        }
#pragma unroll
        for (int i = 0; i < LP_CH; ++i) {
            const int c = c0 + 4 * i;
            if (c < Cm) {
                float r = 0.f, mr = 0.f;
                if (x < Tx) {
                    r = expf(-2.f * lv[i]); mr = mv[i] * r;
                    acc += -0.9189385332046727f - lv[i] - 0.5f * mv[i] * mr;
                }
                lp_tile[c * 65 + lane] = r;
                lp_tile[(Cm + c) * 65 + lane] = mr;
            }
        }
    }
    part[grp * 64 + lane] = acc;
    __syncthreads();
    if (grp == 0 && x < Tx) cb[(long)b * Tx + x] = (part[lane] + part[64 + lane]) + (part[128 + lane] + part[192 + lane]);
    float* out = packed + (long)b * kch * npad * 16;
    for (int kc = 0; kc < kch; ++kc)
        for (int e = threadIdx.x; e < 64 * 16; e += 256) {
            const int n = e >> 4, kk = e & 15, k = kc * 16 + kk;
            out[((long)kc * npad + n0 + n) * 16 + kk] = (k < 2 * Cm) ? lp_tile[k * 65 + n] : 0.f;      // tokens >= Tx hold zeros already
        }
    // masks and lengths: spread over the whole grid
    const long tid = ((long)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x, nth = (long)gridDim.x * gridDim.y * 256;
    for (long i = tid; i < (long)B * Ty; i += nth) { const int bb = (int)(i / Ty); fmask[i] = (i - (long)bb * Ty) < (mlen[bb] / ns) * ns ? 1.f : 0.f; }
    for (long i = tid; i < B; i += nth) { tx32[i] = (int32_t)tlen[i]; ty32[i] = (int32_t)((mlen[i] / ns) * ns); }
}

extern "C" int glowtts_logprior_prep(const float* mean, const float* log_std, const int64_t* token_lengths, const int64_t* mel_lengths, float* packed,
                                     float* cb, float* fmask, int32_t* tx32, int32_t* ty32, int B, int Cm, int Tx, int Ty, int mel_multiple,
                                     int* npad_out, int* kchunks_out, void* stream)
{
    if (B < 1 || Cm < 1 || Tx < 1 || Ty < 1 || mel_multiple < 1) return GLOWTTS_E_ARG;
    const int npad = (Tx + 63) / 64 * 64, kch = (2 * Cm + 15) / 16;
    if (npad_out) *npad_out = npad;
    if (kchunks_out) *kchunks_out = kch;
    if (!packed) return GLOWTTS_OK;                                     // size query
    if (!mean || !log_std || !token_lengths || !mel_lengths || !cb || !fmask || !tx32 || !ty32) return GLOWTTS_E_ARG;
    const size_t lds = ((size_t)2 * Cm * 65 + 256) * sizeof(float);
    if (lds <= 64 * 1024) {
        hipLaunchKernelGGL(logprior_prep_tile_kernel, dim3(npad / 64, B), dim3(256), lds, static_cast<hipStream_t>(stream), mean, log_std, token_lengths,
                           mel_lengths, packed, cb, fmask, tx32, ty32, B, Cm, Tx, Ty, npad, kch, mel_multiple);
        RET_LAUNCH();
    }
    const long work = (long)B * kch * npad * 16;
    hipLaunchKernelGGL(logprior_prep_kernel, dim3((int)((work + 1023) / 1024 > 2048 ? 2048 : (work + 1023) / 1024)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       mean, log_std, token_lengths, mel_lengths, packed, cb, fmask, tx32, ty32, B, Cm, Tx, Ty, npad, kch, mel_multiple);
    RET_LAUNCH();
}
// the same gather as a linear sweep over 16-byte groups of the output (four frames of one (utterance, channel) row per thread: one int4
// of token indices, four gathers from the 480-byte source row, one 16-byte store)
__global__ __launch_bounds__(256) void expand_fwd4_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx, float* __restrict__ out,
                                                          int C, int Tx, int Ty, unsigned int total4)
{
    const unsigned int q = (unsigned int)Ty >> 2;
    for (unsigned int i = blockIdx.x * 256u + threadIdx.x; i < total4; i += gridDim.x * 256u) {
        const unsigned int row = i / q, y4 = i - row * q;          // row = b * C + c
        const unsigned int b = row / (unsigned int)C;
        const int4 id = *reinterpret_cast<const int4*>(idx + (size_t)b * Ty + (size_t)y4 * 4);
        const float* s = src + (size_t)row * Tx;
        float4 v;
        v.x = id.x >= 0 ? s[id.x] : 0.f; v.y = id.y >= 0 ? s[id.y] : 0.f; v.z = id.z >= 0 ? s[id.z] : 0.f; v.w = id.w >= 0 ? s[id.w] : 0.f;
        *reinterpret_cast<float4*>(out + (size_t)i * 4) = v;
    }
}

extern "C" int glowtts_expand_fwd(const float* src, const int32_t* idx, float* out, int B, int C, int Tx, int Ty, void* stream)
{
    if (!src || !idx || !out || B < 1 || C < 1 || Tx < 1 || Ty < 1) return GLOWTTS_E_ARG;
    const uint64_t total4 = (uint64_t)B * C * Ty / 4;
    if ((Ty & 3) == 0 && ((reinterpret_cast<uintptr_t>(idx) | reinterpret_cast<uintptr_t>(out)) & 15) == 0 && total4 < (1ull << 31)) {
        const unsigned int blocks = (unsigned int)((total4 + 255) / 256 < 8192 ? (total4 + 255) / 256 : 8192);
        hipLaunchKernelGGL(expand_fwd4_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), src, idx, out, C, Tx, Ty, (unsigned int)total4);
        RET_LAUNCH();
    }
    hipLaunchKernelGGL(expand_fwd_kernel, dim3((Ty + 255) / 256, C, B), dim3(256), 0, static_cast<hipStream_t>(stream), src, idx, out, C, Tx, Ty);
    RET_LAUNCH();
}
extern "C" int glowtts_expand_bwd(const float* dout, const int32_t* idx, float* dsrc, int B, int C, int Tx, int Ty, void* stream)
{
    if (!dout || !idx || !dsrc || B < 1 || C < 1 || Tx < 1 || Ty < 1) return GLOWTTS_E_ARG;
    const int gx = (C + 3) / 4;                                 // one (b, c) row per wavefront
    const size_t lds = (size_t)4 * Tx * sizeof(float);
    if (lds > 64 * 1024) return GLOWTTS_E_ARG;
    hipLaunchKernelGGL(expand_bwd_kernel, dim3(gx, B), dim3(256), lds, static_cast<hipStream_t>(stream), dout, idx, dsrc, C, Tx, Ty);
    RET_LAUNCH();
}
extern "C" int glowtts_duration_targets(const int32_t* idx, const int64_t* token_lengths, float* out, int B, int Tx, int Ty, void* stream)
{
    if (!idx || !token_lengths || !out || B < 1 || Tx < 1 || Ty < 1) return GLOWTTS_E_ARG;
    hipLaunchKernelGGL(dur_target_kernel, dim3(B), dim3(256), Tx * sizeof(int), static_cast<hipStream_t>(stream), idx, token_lengths, out, Tx, Ty);
    RET_LAUNCH();
}
extern "C" int glowtts_mle_loss_fwd(const float* z, const float* mean, const float* log_std, const float* log_dets, const int64_t* lengths,
                                    float* loss, float* inv_denom, float* scratch /* 1024 floats */, int64_t n, int B, int n_squeeze, int mel_dim, void* stream)
{
    if (!z || !mean || !log_std || !log_dets || !lengths || !loss || !inv_denom || !scratch || n < 1 || B < 1) return GLOWTTS_E_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nblk = (int)((n + 4095) / 4096 > 1024 ? 1024 : (n + 4095) / 4096);
    hipLaunchKernelGGL(mle_partial_kernel, dim3(nblk), dim3(256), 0, s, z, mean, log_std, scratch, (long)n);
    hipLaunchKernelGGL(mle_final_kernel, dim3(1), dim3(256), 0, s, scratch, nblk, log_dets, lengths, B, n_squeeze, mel_dim, loss, inv_denom);
    RET_LAUNCH();
}
extern "C" int glowtts_mle_loss_bwd(const float* z, const float* mean, const float* log_std, const float* dloss, const float* inv_denom,
                                    float* dz, float* dmean, float* dlog_std, int64_t n, float* dlogdet, int B, void* stream)
{
    if (!z || !mean || !log_std || !dloss || !inv_denom || !dz || !dmean || !dlog_std || n < 1 || (dlogdet && B < 1)) return GLOWTTS_E_ARG;
    const long g = (n + 255) / 256;
    GLOWTTS_NOTE_STATIC("mle_loss_bwd");
    hipLaunchKernelGGL(mle_bwd_kernel, dim3((int)(g > 2048 ? 2048 : g)), dim3(256), 0, static_cast<hipStream_t>(stream), z, mean, log_std, dloss, inv_denom, dz, dmean, dlog_std, (long)n,
                       dlogdet, B);
    RET_LAUNCH();
}

extern "C" int glowtts_expand_pair_targets(const float* mean, const float* log_std, const int32_t* idx, const int64_t* token_lengths, float* mel_mean,
                                           float* mel_log_std, float* targets, float* path, int B, int C, int Tx, int Ty, void* stream)
{
    if (!mean || !log_std || !idx || !token_lengths || !mel_mean || !mel_log_std || !targets || B < 1 || C < 1 || Tx < 1 || Ty < 4 || (Ty & 3)) return GLOWTTS_E_ARG;
    const uint64_t total4 = (uint64_t)B * C * Ty / 4, ptotal4 = path ? (uint64_t)B * Tx * Ty / 4 : 0;
    if (((reinterpret_cast<uintptr_t>(idx) | reinterpret_cast<uintptr_t>(mel_mean) | reinterpret_cast<uintptr_t>(mel_log_std) | reinterpret_cast<uintptr_t>(path)) & 15) != 0 ||
        total4 >= (1ull << 31) || ptotal4 >= (1ull << 31) || (size_t)Tx * sizeof(int) > 64 * 1024) return GLOWTTS_E_ARG;
    const int nexp = (int)((total4 + 255) / 256 < 8192 ? (total4 + 255) / 256 : 8192);
    const int npath = (int)((ptotal4 + 255) / 256 < 8192 ? (ptotal4 + 255) / 256 : 8192);
    GLOWTTS_NOTE_STATIC("expand_pair");
    hipLaunchKernelGGL(expand_pair_kernel, dim3(nexp + npath + B), dim3(256), Tx * sizeof(int), static_cast<hipStream_t>(stream), mean, log_std, idx, token_lengths,
                       mel_mean, mel_log_std, targets, C, Tx, Ty, (unsigned int)total4, nexp, path, (unsigned int)ptotal4, npath);
    RET_LAUNCH();
}
extern "C" int glowtts_prior_loss_bwd(const float* z, const float* mean, const float* log_std, const int32_t* idx, const float* dloss, const float* inv_denom,
                                      float* dz, float* dmean, float* dlog_std, float* dlogdet, int B, int C, int Tx, int Ty, void* stream)
{
    if (!z || !mean || !log_std || !idx || !dloss || !inv_denom || !dz || !dmean || !dlog_std || B < 1 || C < 1 || Tx < 1 || Ty < 1) return GLOWTTS_E_ARG;
    const size_t lds = (size_t)16 * Tx * sizeof(float);
    if (lds > 64 * 1024) return GLOWTTS_E_ARG;
    GLOWTTS_NOTE_STATIC("prior_loss_bwd");
    hipLaunchKernelGGL(prior_loss_bwd_kernel, dim3((C + 3) / 4, B), dim3(256), lds, static_cast<hipStream_t>(stream), z, mean, log_std, idx, dloss, inv_denom,
                       dz, dmean, dlog_std, dlogdet, B, C, Tx, Ty);
    RET_LAUNCH();
}
extern "C" int glowtts_prior_loss(const float* z, const float* mel_mean, const float* mel_log_std, const float* mean, const float* log_std, const int32_t* idx,
                                  const float* log_dets, const int64_t* lengths, const float* dloss, float* scratch /* 1024 floats */, uint32_t* counter /* zeroed once */,
                                  float* loss, float* inv_denom, float* dz, float* dmean, float* dlog_std, float* dlogdet, int B, int C, int Tx, int Ty,
                                  int n_squeeze, int mel_dim, void* stream)
{
    if (!z || !mel_mean || !mel_log_std || !mean || !log_std || !idx || !log_dets || !lengths || !dloss || !scratch || !counter || !loss || !inv_denom || !dz ||
        !dmean || !dlog_std || B < 1 || C < 1 || Tx < 1 || Ty < 1 || n_squeeze < 1 || mel_dim < 1) return GLOWTTS_E_ARG;
    const size_t lds = (size_t)16 * Tx * sizeof(float);
    if (lds > 64 * 1024) return GLOWTTS_E_ARG;
    const int64_t n = (int64_t)B * C * Ty;
    const int nblk = (int)((n + 4095) / 4096 > 1024 ? 1024 : (n + 4095) / 4096);      // (glowtts_mle_loss_fwd's)
    const int cg = (C + 3) / 4;
    GLOWTTS_NOTE_STATIC("prior_loss");
    hipLaunchKernelGGL(prior_loss_kernel, dim3(nblk + cg * B), dim3(256), lds, static_cast<hipStream_t>(stream), z, mel_mean, mel_log_std, mean, log_std, idx, log_dets,
                       lengths, dloss, scratch, counter, loss, inv_denom, dz, dmean, dlog_std, dlogdet, (long)n, nblk, cg, B, C, Tx, Ty, n_squeeze, mel_dim);
    RET_LAUNCH();
}
extern "C" int glowtts_mse_loss_fwd(const float* a, const float* target, float* loss, int64_t n, float scale, const int64_t* lengths, int B, const float* extent,
                                    float* da_unit, void* stream)
{
    if (!a || !target || !loss || n < 1 || ((lengths || extent) && B < 1)) return GLOWTTS_E_ARG;
    GLOWTTS_NOTE_STATIC("mse_loss_fwd");
    hipLaunchKernelGGL(mse_fwd_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), a, target, loss, (long)n, scale, lengths, B, extent, da_unit);
    RET_LAUNCH();
}
extern "C" int glowtts_mse_loss_bwd(const float* a, const float* target, const float* dloss, float* da, int64_t n, float scale, const int64_t* lengths, int B,
                                    const float* extent, void* stream)
{
    if (!a || !target || !dloss || !da || n < 1 || ((lengths || extent) && B < 1)) return GLOWTTS_E_ARG;
    const long g = (n + 255) / 256;
    GLOWTTS_NOTE_STATIC("mse_loss_bwd");
    hipLaunchKernelGGL(mse_bwd_kernel, dim3((int)(g > 256 ? 256 : g)), dim3(256), 0, static_cast<hipStream_t>(stream), a, target, dloss, da, (long)n, scale, lengths, B,
                       extent);
    RET_LAUNCH();
}

// The two ends of the text encoder that are not GEMMs, for gfx950 (round 6; each was a handful of PyTorch elementwise / reduce / fill / copy launches
// at the tail of the encoder's forward chain and at the HEAD of its backward chain, where every launch waits for a CU beside the flow decoder's
// chip-filling kernels):
//   dur_proj    Duration_Predictor's Projection (Modules.py:596-618): Conv1d(C -> 1, k = 1) on the masked features, times the mask - N = 1 is not a
//               GEMM: a dot product per token row; its backward writes the feature gradient rows and the weight / bias gradients (two-stage, fixed order).
//   prior_split the encoder's Project output rows [B][T + 2 pad][2 M] -> mean, log_std [B][M][T] (Modules.py:283-286: torch.split of the projected
//               channels; this library keeps activations as rows, the log-prior / MAS / MLE side wants channel-first) and the gather back.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/glowtts_hip.h"
#include "launch_log.h"

namespace {

// out[b][t] = (sum_c d[(b Tp + pad + t)][c] w[c] + bias) * mask[b][t]; one wavefront per token
__global__ __launch_bounds__(256) void dur_proj_fwd_kernel(const float* __restrict__ d, const float* __restrict__ w, const float* __restrict__ bias,
                                                           const float* __restrict__ mask, float* __restrict__ out, int B, int T, int Tp, int pad, int C)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + wave;
    if (item >= B * T) return;
    const int b = item / T, t = item - b * T;
    const float* row = d + ((long)b * Tp + pad + t) * C;
    float acc = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(row + c), ww = *reinterpret_cast<const float4*>(w + c);
        acc += v.x * ww.x + v.y * ww.y + v.z * ww.z + v.w * ww.w;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) out[item] = (acc + bias[0]) * mask[item];
}

// g [B][T] -> dd rows [B Tp][C] (zero in the pad rows), dw [C], dbias [1].  Workgroup b: its four waves take the tokens t = wave (mod 4), lane l the channels
// [4 l, 4 l + 4) of every 256-channel chunk; the four waves' sums meet in LDS in wave order, the workgroups' in `partial` [B][C + 1] in utterance order
// (the last workgroup to finish, a device counter reset for the next launch): fixed order, deterministic.
constexpr int DP_MAXCH = 4;                                    // C <= 1024
__global__ __launch_bounds__(256) void dur_proj_bwd_kernel(const float* __restrict__ g, const float* __restrict__ mask, const float* __restrict__ d,
                                                           const float* __restrict__ w, float* __restrict__ dd, float* __restrict__ dw,
                                                           float* __restrict__ dbias, float* __restrict__ partial, unsigned int* __restrict__ counter,
                                                           int B, int T, int Tp, int pad, int C)
{
    __shared__ float4 red[4][DP_MAXCH][64];
    __shared__ float redb[4];
    __shared__ int last;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, b = blockIdx.x;
    const int nch = (C + 255) >> 8;
    float4 acc[DP_MAXCH], ww[DP_MAXCH];
#pragma unroll
    for (int k = 0; k < DP_MAXCH; ++k) {
        acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int c = k * 256 + lane * 4;
        ww[k] = (k < nch && c < C) ? *reinterpret_cast<const float4*>(w + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float accb = 0.f;
    // pad rows of this utterance: zero
    for (int i = threadIdx.x; i < 2 * pad * (C >> 2); i += 256) {
        const int r = i / (C >> 2), c4 = i - r * (C >> 2);
        const int row = r < pad ? r : Tp - 2 * pad + r;
        *reinterpret_cast<float4*>(dd + ((long)b * Tp + row) * C + c4 * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int t = wave; t < T; t += 4) {
        const float gm = g[b * T + t] * mask[b * T + t];
        accb += gm;
        const long ro = ((long)b * Tp + pad + t) * C;
#pragma unroll
        for (int k = 0; k < DP_MAXCH; ++k) {
            const int c = k * 256 + lane * 4;
            if (k < nch && c < C) {
                const float4 v = *reinterpret_cast<const float4*>(d + ro + c);
                acc[k].x += gm * v.x; acc[k].y += gm * v.y; acc[k].z += gm * v.z; acc[k].w += gm * v.w;
                *reinterpret_cast<float4*>(dd + ro + c) = make_float4(gm * ww[k].x, gm * ww[k].y, gm * ww[k].z, gm * ww[k].w);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < DP_MAXCH; ++k) red[wave][k][lane] = acc[k];
    if (lane == 0) redb[wave] = accb;
    __syncthreads();
    // (hand-over without a release fence - a device-scope release writes back the XCD's whole L2 on this chip: the partial sums go out as device-scope atomic
    //  exchanges, performed at the memory side, each wave waits for its own before the workgroup counts itself; loss_ops.hip prior_loss_kernel)
    float* pb = partial + (long)b * (C + 1);
    for (int i = threadIdx.x; i < C; i += 256) {
        const int k = i >> 8, l = (i & 255) >> 2, j = i & 3;
        float s = 0.f;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) { const float4 v = red[wv][k][l]; s += j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); }
        const float old = atomicExch(pb + i, s);               // (returning: performed when the value is back)
        asm volatile("s_waitcnt vmcnt(0)" :: "v"(old) : "memory");
    }
    if (threadIdx.x == 0) {
        const float old = atomicExch(pb + C, redb[0] + redb[1] + redb[2] + redb[3]);
        asm volatile("s_waitcnt vmcnt(0)" :: "v"(old) : "memory");
    }
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(counter, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    for (int i = threadIdx.x; i <= C; i += 256) {
        float s = 0.f;
        for (int bb = 0; bb < B; ++bb) s += __uint_as_float(atomicOr(reinterpret_cast<unsigned int*>(partial + (long)bb * (C + 1) + i), 0u));
        if (i < C) dw[i] = s; else dbias[0] = s;
    }
    if (threadIdx.x == 0) atomicExch(counter, 0u);
}

// rows [B][Tp][2 M] -> mean, log_std [B][M][T]: a workgroup owns 32 tokens of one utterance, through an LDS tile (coalesced on both sides)
__global__ __launch_bounds__(256) void prior_split_fwd_kernel(const float* __restrict__ rows, float* __restrict__ mean, float* __restrict__ ls, int T, int Tp, int pad,
                                                              int M)
{
    extern __shared__ float tile[];                            // [32][2 M + 1]
    const int b = blockIdx.y, t0 = blockIdx.x * 32, C2 = 2 * M, ld = C2 + 1;
    const int nt = T - t0 < 32 ? T - t0 : 32;
    for (int i = threadIdx.x; i < nt * C2; i += 256) {
        const int r = i / C2, c = i - r * C2;
        tile[r * ld + c] = rows[((long)b * Tp + pad + t0 + r) * C2 + c];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C2 * 32; i += 256) {
        const int c = i >> 5, r = i & 31;
        if (r < nt) (c < M ? mean + ((long)b * M + c) * T : ls + ((long)b * M + (c - M)) * T)[t0 + r] = tile[r * ld + c];
    }
}
// dmean, dlog_std [B][M][T] (either may be NULL: zeros) -> drows [B][Tp][2 M], zero in the pad rows
__global__ __launch_bounds__(256) void prior_split_bwd_kernel(const float* __restrict__ dmean, const float* __restrict__ dls, float* __restrict__ drows, int T, int Tp,
                                                              int pad, int M)
{
    extern __shared__ float tile[];
    const int b = blockIdx.y, t0 = blockIdx.x * 32, C2 = 2 * M, ld = C2 + 1;
    const int nt = T - t0 < 32 ? T - t0 : 32;
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < 2 * pad * C2; i += 256) {
            const int r = i / C2, c = i - r * C2;
            drows[((long)b * Tp + (r < pad ? r : Tp - 2 * pad + r)) * C2 + c] = 0.f;
        }
    }
    for (int i = threadIdx.x; i < C2 * 32; i += 256) {
        const int c = i >> 5, r = i & 31;
        const float* src = c < M ? dmean : dls;
        if (r < nt) tile[r * ld + c] = src ? src[((long)b * M + (c < M ? c : c - M)) * T + t0 + r] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nt * C2; i += 256) {
        const int r = i / C2, c = i - r * C2;
        drows[((long)b * Tp + pad + t0 + r) * C2 + c] = tile[r * ld + c];
    }
}

#define RET_LAUNCH() return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH
}  // namespace

extern "C" int glowtts_dur_proj_supported(int C) { return C >= 4 && C <= 256 * DP_MAXCH && (C & 3) == 0; }

extern "C" int glowtts_dur_proj_fwd(const float* d, const float* w, const float* bias, const float* mask, float* out, int B, int T, int pad, int C, void* stream)
{
    if (!d || !w || !bias || !mask || !out || B < 1 || T < 1 || pad < 0 || !glowtts_dur_proj_supported(C) ||
        ((reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(w)) & 15)) return GLOWTTS_E_ARG;
    GLOWTTS_NOTE_STATIC("dur_proj_fwd");
    hipLaunchKernelGGL(dur_proj_fwd_kernel, dim3((B * T + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), d, w, bias, mask, out, B, T, T + 2 * pad, pad, C);
    RET_LAUNCH();
}
extern "C" int glowtts_dur_proj_bwd(const float* g, const float* mask, const float* d, const float* w, float* dd, float* dw, float* dbias,
                                    float* scratch /* B * (C + 1) floats */, uint32_t* counter /* zeroed once */, int B, int T, int pad, int C, void* stream)
{
    if (!g || !mask || !d || !w || !dd || !dw || !dbias || !scratch || !counter || B < 1 || T < 1 || pad < 0 || !glowtts_dur_proj_supported(C) ||
        ((reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(dd)) & 15)) return GLOWTTS_E_ARG;
    GLOWTTS_NOTE_STATIC("dur_proj_bwd");
    hipLaunchKernelGGL(dur_proj_bwd_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), g, mask, d, w, dd, dw, dbias, scratch, counter, B, T, T + 2 * pad,
                       pad, C);
    RET_LAUNCH();
}
extern "C" int glowtts_prior_split_fwd(const float* rows, float* mean, float* log_std, int B, int T, int pad, int M, void* stream)
{
    if (!rows || !mean || !log_std || B < 1 || T < 1 || pad < 0 || M < 1 || (size_t)32 * (2 * M + 1) * 4 > 64 * 1024) return GLOWTTS_E_ARG;
    GLOWTTS_NOTE_STATIC("prior_split_fwd");
    hipLaunchKernelGGL(prior_split_fwd_kernel, dim3((T + 31) / 32, B), dim3(256), (size_t)32 * (2 * M + 1) * 4, static_cast<hipStream_t>(stream), rows, mean, log_std, T,
                       T + 2 * pad, pad, M);
    RET_LAUNCH();
}
extern "C" int glowtts_prior_split_bwd(const float* dmean, const float* dlog_std, float* drows, int B, int T, int pad, int M, void* stream)
{
    if (!drows || B < 1 || T < 1 || pad < 0 || M < 1 || (size_t)32 * (2 * M + 1) * 4 > 64 * 1024) return GLOWTTS_E_ARG;
    GLOWTTS_NOTE_STATIC("prior_split_bwd");
    hipLaunchKernelGGL(prior_split_bwd_kernel, dim3((T + 31) / 32, B), dim3(256), (size_t)32 * (2 * M + 1) * 4, static_cast<hipStream_t>(stream), dmean, dlog_std, drows, T,
                       T + 2 * pad, pad, M);
    RET_LAUNCH();
}

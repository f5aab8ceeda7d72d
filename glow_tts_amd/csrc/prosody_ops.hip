// GRU recurrence of the GST prosody encoder (Modules.py:338-343, 371: torch.nn.GRU, one layer, batch_first, h0 = 0) for gfx950.
// Off the hot path by FLOPs but not by time: MIOpen runs the 13-step recurrence of BASELINE config 5 as ~400
// launches of 3-5 us each (forward + backward), 2 ms of a 8.7 ms training step.  Here the recurrence is ONE launch per direction - one
// workgroup per utterance, one thread per gate row - and the four GEMMs around it (input projection, its transposes) stay with the caller.
//   r = sigmoid(gi_r + gh_r)   z = sigmoid(gi_z + gh_z)   n = tanh(gi_n + r * gh_n)   h' = (1 - z) * n + z * h,   gh = W_hh h + b_hh
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include "../../include/glowtts_hip.h"
#include "launch_log.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// The same with the thread's row of W_hh held in REGISTERS for the whole recurrence (HH = H floats: 128 VGPRs at the GST encoder's size; the generic kernel
// re-reads its row - 3 H^2 floats per workgroup - from L2 at every time step: 81 us for 13 steps, round 5 profile of config 5).
template <int HH>
__global__ __launch_bounds__(3 * HH) void gru_fwd_reg_kernel(const float* __restrict__ gi, const float* __restrict__ w_hh, const float* __restrict__ b_hh,
                                                             float* __restrict__ hs, float* __restrict__ keep, int T)
{
    __shared__ __attribute__((aligned(16))) float h[HH];
    __shared__ float gh[3 * HH];
    constexpr int H = HH, G = 3 * HH;
    const int b = blockIdx.x, j = threadIdx.x;
    float w[HH];
#pragma unroll
    for (int k = 0; k < HH; k += 4) {
        const float4 x = *reinterpret_cast<const float4*>(w_hh + (size_t)j * H + k);
        w[k] = x.x; w[k + 1] = x.y; w[k + 2] = x.z; w[k + 3] = x.w;
    }
    if (j < H) h[j] = 0.f;
    const float bj = b_hh[j];
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        // (the same fused-multiply-add chain as the generic kernel: bit-identical results)
        float s = bj;
        float g0 = 0.f, g1 = 0.f, g2 = 0.f;
        if (j < H) {                                        // this step's input-side pre-activations: in flight under the dot product
            const float* g = gi + ((size_t)b * T + t) * G;
            g0 = g[j]; g1 = g[H + j]; g2 = g[2 * H + j];
        }
#pragma unroll
        for (int k = 0; k < HH; k += 4) {
            const float4 x = *reinterpret_cast<const float4*>(h + k);
            s = fmaf(w[k], x.x, s); s = fmaf(w[k + 1], x.y, s); s = fmaf(w[k + 2], x.z, s); s = fmaf(w[k + 3], x.w, s);
        }
        gh[j] = s;
        __syncthreads();
        if (j < H) {
            const float r = sigmoidf_(g0 + gh[j]);
            const float z = sigmoidf_(g1 + gh[H + j]);
            const float hn = gh[2 * H + j];
            const float n = tanhf(g2 + r * hn);
            const float hnew = (1.f - z) * n + z * h[j];
            float* kp = keep + ((size_t)b * T + t) * 4 * H;
            kp[j] = r; kp[H + j] = z; kp[2 * H + j] = n; kp[3 * H + j] = hn;
            hs[((size_t)b * T + t) * H + j] = hnew;
            h[j] = hnew;
        }
        __syncthreads();
    }
}

// backward, likewise: thread (g, k) keeps column k of gate group g of W_hh (HH floats) in registers
template <int HH>
__global__ __launch_bounds__(3 * HH) void gru_bwd_reg_kernel(const float* __restrict__ dhs, const float* __restrict__ hs, const float* __restrict__ keep,
                                                             const float* __restrict__ w_hh, float* __restrict__ dgi, float* __restrict__ dgh, int T)
{
    __shared__ float dh[HH];
    __shared__ __attribute__((aligned(16))) float dg[3 * HH];
    __shared__ float part[3 * HH];
    constexpr int H = HH, G = 3 * HH;
    const int b = blockIdx.x, j = threadIdx.x;
    const int gq = j / H, kq = j - gq * H;
    float w[HH];
#pragma unroll
    for (int i = 0; i < HH; ++i) w[i] = w_hh[(size_t)gq * H * H + (size_t)i * H + kq];
    if (j < H) dh[j] = 0.f;
    __syncthreads();
    for (int t = T - 1; t >= 0; --t) {
        float carry = 0.f;
        if (j < H) {
            const size_t bt = (size_t)b * T + t;
            const float* kp = keep + bt * 4 * H;
            const float r = kp[j], z = kp[H + j], n = kp[2 * H + j], hn = kp[3 * H + j];
            const float hprev = t > 0 ? hs[(bt - 1) * H + j] : 0.f;
            const float d = dh[j] + dhs[bt * H + j];
            const float dpn = d * (1.f - z) * (1.f - n * n);
            const float dpz = d * (hprev - n) * z * (1.f - z);
            const float dpr = dpn * hn * r * (1.f - r);
            float* gi_ = dgi + bt * G; float* gh_ = dgh + bt * G;
            gi_[j] = dpr; gi_[H + j] = dpz; gi_[2 * H + j] = dpn;
            gh_[j] = dpr; gh_[H + j] = dpz; gh_[2 * H + j] = dpn * r;
            dg[j] = dpr; dg[H + j] = dpz; dg[2 * H + j] = dpn * r;
            carry = d * z;
        }
        __syncthreads();
        {
            const float* dgg = dg + gq * H;
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < HH; i += 4) {
                const float4 x = *reinterpret_cast<const float4*>(dgg + i);
                s = fmaf(w[i], x.x, s); s = fmaf(w[i + 1], x.y, s); s = fmaf(w[i + 2], x.z, s); s = fmaf(w[i + 3], x.w, s);
            }
            part[j] = s;
        }
        __syncthreads();
        if (j < H) dh[j] = carry + part[j] + part[H + j] + part[2 * H + j];
        __syncthreads();
    }
}

// block = 3H threads (thread j owns gate row j of W_hh); dynamic LDS: h [H] + gh [3H]
__global__ void gru_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ w_hh, const float* __restrict__ b_hh,
                               float* __restrict__ hs, float* __restrict__ keep, int T, int H)
{
    extern __shared__ float sm[];
    float* h = sm; float* gh = sm + H;
    const int b = blockIdx.x, j = threadIdx.x, G = 3 * H;
    if (j < H) h[j] = 0.f;
    const float bj = b_hh[j];
    const float* wrow = w_hh + (size_t)j * H;
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        float s = bj;
        if ((H & 3) == 0) {
            for (int k = 0; k < H; k += 4) {
                const float4 w = *reinterpret_cast<const float4*>(wrow + k);
                s = fmaf(w.x, h[k], s); s = fmaf(w.y, h[k + 1], s); s = fmaf(w.z, h[k + 2], s); s = fmaf(w.w, h[k + 3], s);
            }
        } else {
            for (int k = 0; k < H; ++k) s = fmaf(wrow[k], h[k], s);
        }
        gh[j] = s;
        __syncthreads();
        if (j < H) {
            const float* g = gi + ((size_t)b * T + t) * G;
            const float r = sigmoidf_(g[j] + gh[j]);
            const float z = sigmoidf_(g[H + j] + gh[H + j]);
            const float hn = gh[2 * H + j];
            const float n = tanhf(g[2 * H + j] + r * hn);
            const float hnew = (1.f - z) * n + z * h[j];
            float* kp = keep + ((size_t)b * T + t) * 4 * H;
            kp[j] = r; kp[H + j] = z; kp[2 * H + j] = n; kp[3 * H + j] = hn;
            hs[((size_t)b * T + t) * H + j] = hnew;
            h[j] = hnew;
        }
        __syncthreads();
    }
}

// block = 3H threads; dynamic LDS: dh [H] + dgh [3H] + partial [3H]
__global__ void gru_bwd_kernel(const float* __restrict__ dhs, const float* __restrict__ hs, const float* __restrict__ keep,
                               const float* __restrict__ w_hh, float* __restrict__ dgi, float* __restrict__ dgh, int T, int H)
{
    extern __shared__ float sm[];
    float* dh = sm; float* dg = sm + H; float* part = sm + 4 * H;
    const int b = blockIdx.x, j = threadIdx.x, G = 3 * H;
    if (j < H) dh[j] = 0.f;
    __syncthreads();
    for (int t = T - 1; t >= 0; --t) {
        float carry = 0.f;
        if (j < H) {
            const size_t bt = (size_t)b * T + t;
            const float* kp = keep + bt * 4 * H;
            const float r = kp[j], z = kp[H + j], n = kp[2 * H + j], hn = kp[3 * H + j];
            const float hprev = t > 0 ? hs[(bt - 1) * H + j] : 0.f;
            const float d = dh[j] + dhs[bt * H + j];
            const float dpn = d * (1.f - z) * (1.f - n * n);
            const float dpz = d * (hprev - n) * z * (1.f - z);
            const float dpr = dpn * hn * r * (1.f - r);
            float* gi_ = dgi + bt * G; float* gh_ = dgh + bt * G;
            gi_[j] = dpr; gi_[H + j] = dpz; gi_[2 * H + j] = dpn;
            gh_[j] = dpr; gh_[H + j] = dpz; gh_[2 * H + j] = dpn * r;
            dg[j] = dpr; dg[H + j] = dpz; dg[2 * H + j] = dpn * r;
            carry = d * z;
        }
        __syncthreads();
        {   // dh_prev[k] = carry[k] + sum_j W_hh[j][k] dgh[j]: thread (g, k) sums the H gate rows of group g (coalesced over k)
            const int g = j / H, k = j - g * H;
            const float* w = w_hh + (size_t)g * H * H + k;
            const float* dgg = dg + g * H;
            float s = 0.f;
            for (int i = 0; i < H; ++i) s = fmaf(w[(size_t)i * H], dgg[i], s);
            part[j] = s;
        }
        __syncthreads();
        if (j < H) dh[j] = carry + part[j] + part[H + j] + part[2 * H + j];
        __syncthreads();
    }
}

}  // namespace

extern "C" int glowtts_gru_fwd(const float* gi, const float* w_hh, const float* b_hh, float* hs, float* keep, int B, int T, int H, void* stream)
{
    if (!gi || !w_hh || !b_hh || !hs || !keep || B < 1 || T < 1 || H < 1 || 3 * H > 1024) return GLOWTTS_E_ARG;
    if (H == 128 && (reinterpret_cast<uintptr_t>(w_hh) & 15) == 0) {
        hipLaunchKernelGGL(gru_fwd_reg_kernel<128>, dim3(B), dim3(384), 0, static_cast<hipStream_t>(stream), gi, w_hh, b_hh, hs, keep, T);
        return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
    }
    hipLaunchKernelGGL(gru_fwd_kernel, dim3(B), dim3(3 * H), 4 * H * sizeof(float), static_cast<hipStream_t>(stream), gi, w_hh, b_hh, hs, keep, T, H);
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

extern "C" int glowtts_gru_bwd(const float* dhs, const float* hs, const float* keep, const float* w_hh, float* dgi, float* dgh,
                               int B, int T, int H, void* stream)
{
    if (!dhs || !hs || !keep || !w_hh || !dgi || !dgh || B < 1 || T < 1 || H < 1 || 3 * H > 1024) return GLOWTTS_E_ARG;
    if (H == 128) {
        hipLaunchKernelGGL(gru_bwd_reg_kernel<128>, dim3(B), dim3(384), 0, static_cast<hipStream_t>(stream), dhs, hs, keep, w_hh, dgi, dgh, T);
        return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
    }
    hipLaunchKernelGGL(gru_bwd_kernel, dim3(B), dim3(3 * H), 7 * H * sizeof(float), static_cast<hipStream_t>(stream), dhs, hs, keep, w_hh, dgi, dgh, T, H);
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

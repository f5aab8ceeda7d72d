// Parameter-side kernels (gfx950): old-style weight normalisation w = g * v / ||v|| of the WaveNet convs
// (Modules.py:766, 818, 825, 838, 845: torch.nn.utils.weight_norm, norm over (in, k) per output channel), forward and
// backward, for a whole stack of convolutions in one launch: rows = (stacked convs x output channels), cols = in * k.
// One wavefront per row; a row (<= a few KB) is read once and kept in registers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/glowtts_hip.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

constexpr int WN_MAXK = 32;            // elements per lane kept in registers: cols <= 2048 (longer rows take the strided path)

__global__ __launch_bounds__(256) void weightnorm_fwd_kernel(const float* __restrict__ v, const float* __restrict__ g, float* __restrict__ w,
                                                             float* __restrict__ inv_norm, long rows, int cols)
{
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float* vr = v + r * cols;
    float* wr = w + r * cols;
    float x[WN_MAXK];
    float s = 0.f;
    const bool fits = cols <= 64 * WN_MAXK;
    if (fits) {
#pragma unroll
        for (int k = 0; k < WN_MAXK; ++k) { const int c = lane + 64 * k; x[k] = c < cols ? vr[c] : 0.f; s += x[k] * x[k]; }
    } else {
        for (int c = lane; c < cols; c += 64) { const float t = vr[c]; s += t * t; }
    }
    s = wave_sum(s);
    const float inv = 1.f / sqrtf(s);
    const float sc = g[r] * inv;
    if (fits) {
#pragma unroll
        for (int k = 0; k < WN_MAXK; ++k) { const int c = lane + 64 * k; if (c < cols) wr[c] = x[k] * sc; }
    } else {
        for (int c = lane; c < cols; c += 64) wr[c] = vr[c] * sc;
    }
    if (lane == 0) inv_norm[r] = inv;
}

// dg = <dw, v> / ||v||,   dv = g / ||v|| * (dw - v * <dw, v> / ||v||^2)
__global__ __launch_bounds__(256) void weightnorm_bwd_kernel(const float* __restrict__ dw, const float* __restrict__ v, const float* __restrict__ g,
                                                             const float* __restrict__ inv_norm, float* __restrict__ dv, float* __restrict__ dg,
                                                             long rows, int cols)
{
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float* vr = v + r * cols;
    const float* dr = dw + r * cols;
    float* or_ = dv + r * cols;
    float xv[WN_MAXK], xd[WN_MAXK];
    float dot = 0.f;
    const bool fits = cols <= 64 * WN_MAXK;
    if (fits) {
#pragma unroll
        for (int k = 0; k < WN_MAXK; ++k) {
            const int c = lane + 64 * k;
            xv[k] = c < cols ? vr[c] : 0.f; xd[k] = c < cols ? dr[c] : 0.f;
            dot += xv[k] * xd[k];
        }
    } else {
        for (int c = lane; c < cols; c += 64) dot += vr[c] * dr[c];
    }
    dot = wave_sum(dot);
    const float inv = inv_norm[r], sc = g[r] * inv, k2 = dot * inv * inv;
    if (fits) {
#pragma unroll
        for (int k = 0; k < WN_MAXK; ++k) { const int c = lane + 64 * k; if (c < cols) or_[c] = sc * (xd[k] - xv[k] * k2); }
    } else {
        for (int c = lane; c < cols; c += 64) or_[c] = sc * (dr[c] - vr[c] * k2);
    }
    if (lane == 0) dg[r] = dot * inv;
}

}  // namespace

extern "C" int glowtts_weightnorm_fwd(const float* v, const float* g, float* w, float* inv_norm, int64_t rows, int cols, void* stream)
{
    if (!v || !g || !w || !inv_norm || rows < 1 || cols < 1) return GLOWTTS_E_ARG;
    hipLaunchKernelGGL(weightnorm_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), v, g, w, inv_norm, (long)rows, cols);
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

extern "C" int glowtts_weightnorm_bwd(const float* dw, const float* v, const float* g, const float* inv_norm, float* dv, float* dg,
                                      int64_t rows, int cols, void* stream)
{
    if (!dw || !v || !g || !inv_norm || !dv || !dg || rows < 1 || cols < 1) return GLOWTTS_E_ARG;
    hipLaunchKernelGGL(weightnorm_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), dw, v, g, inv_norm, dv, dg,
                       (long)rows, cols);
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

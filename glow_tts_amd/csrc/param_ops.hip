// Parameter-side kernels (gfx950): old-style weight normalisation w = g * v / ||v|| of the WaveNet convs
// (Modules.py:766, 818, 825, 838, 845: torch.nn.utils.weight_norm, norm over (in, k) per output channel), forward and
// backward, for a whole stack of convolutions in one launch: rows = (stacked convs x output channels), cols = in * k.
// One wavefront per row; a row (<= a few KB) is read once and kept in registers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "../../include/glowtts_hip.h"
#include "launch_log.h"

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

// several weight-norm backward problems in ONE launch (the decoder's four weight-normalised classes: four dependent launches of 5 - 40 us at the very end of the
// step's critical chain; as one launch they run side by side).  The table travels in the argument segment.
struct wn_bwd_table { glowtts_wn_bwd_job job[GLOWTTS_WN_BWD_MAX_JOBS]; int block0[GLOWTTS_WN_BWD_MAX_JOBS]; };
__global__ __launch_bounds__(256) void weightnorm_bwd_multi_kernel(const wn_bwd_table tab, int njobs)
{
    int lo = 0;
#pragma unroll
    for (int i = 1; i < GLOWTTS_WN_BWD_MAX_JOBS; ++i) lo = (i < njobs && tab.block0[i] <= (int)blockIdx.x) ? i : lo;
    const glowtts_wn_bwd_job& j = tab.job[lo];
    const long r = (long)(blockIdx.x - tab.block0[lo]) * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, cols = j.cols;
    if (r >= j.rows) return;
    const float* vr = j.v + r * cols;
    const float* dr = j.dw + r * cols;
    float* or_ = j.dv + r * cols;
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
    const float inv = j.inv_norm[r], sc = j.g[r] * inv, k2 = dot * inv * inv;
    if (fits) {
#pragma unroll
        for (int k = 0; k < WN_MAXK; ++k) { const int c = lane + 64 * k; if (c < cols) or_[c] = sc * (xd[k] - xv[k] * k2); }
    } else {
        for (int c = lane; c < cols; c += 64) or_[c] = sc * (dr[c] - vr[c] * k2);
    }
    if (lane == 0) j.dg[r] = dot * inv;
}

}  // namespace

extern "C" int glowtts_weightnorm_bwd_multi(const glowtts_wn_bwd_job* jobs, int njobs, void* stream)
{
    if (!jobs || njobs < 1 || njobs > GLOWTTS_WN_BWD_MAX_JOBS) return GLOWTTS_E_ARG;
    wn_bwd_table tab;
    memset(&tab, 0, sizeof(tab));
    long blocks = 0;
    for (int i = 0; i < njobs; ++i) {
        const glowtts_wn_bwd_job& j = jobs[i];
        if (!j.dw || !j.v || !j.g || !j.inv_norm || !j.dv || !j.dg || j.rows < 1 || j.cols < 1) return GLOWTTS_E_ARG;
        tab.job[i] = j;
        tab.block0[i] = (int)blocks;
        blocks += (j.rows + 3) / 4;
    }
    if (blocks >= (1L << 31)) return GLOWTTS_E_ARG;
    GLOWTTS_NOTE_STATIC("weightnorm_bwd_multi");
    hipLaunchKernelGGL(weightnorm_bwd_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), tab, njobs);
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

extern "C" int glowtts_weightnorm_fwd(const float* v, const float* g, float* w, float* inv_norm, int64_t rows, int cols, void* stream)
{
    if (!v || !g || !w || !inv_norm || rows < 1 || cols < 1) return GLOWTTS_E_ARG;
    GLOWTTS_NOTE_STATIC("weightnorm_fwd");
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

// ------------------------------------------------------------------------------------------------
// Optimizer side of the step: Rectified Adam exactly as the reference's Radam.py:25-90 and the global gradient
// norm / clipping of torch.nn.utils.clip_grad_norm_ (Train.py:228-231), as multi-tensor launches over a device job table:
// the reference walks ~500 parameter tensors in a Python loop with ~10 small torch kernels each.
// One job = one contiguous run of elements (a stacked weight class is ONE job); blocks of OPT_CHUNK elements.
// ------------------------------------------------------------------------------------------------
namespace {

constexpr int OPT_CHUNK = 4096;        // elements per workgroup

__device__ __forceinline__ const glowtts_opt_job& find_job(const glowtts_opt_job* __restrict__ jobs, int njobs, long block, long& first)
{
    int lo = 0, hi = njobs - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (jobs[mid].block0 <= block) lo = mid; else hi = mid - 1; }
    first = (block - jobs[lo].block0) * OPT_CHUNK;
    return jobs[lo];
}

// partial[b] = sum of g^2 over block b
__global__ __launch_bounds__(256) void multi_sqnorm_kernel(const glowtts_opt_job* __restrict__ jobs, int njobs, float* __restrict__ partial)
{
    long first;
    const glowtts_opt_job& j = find_job(jobs, njobs, blockIdx.x, first);
    const long end = min(first + OPT_CHUNK, (long)j.n);
    float s = 0.f;
    if (end - first == OPT_CHUNK && (reinterpret_cast<uintptr_t>(j.g + first) & 15) == 0) {
        // a full chunk on a 16-byte boundary: the four 16-byte loads of a thread issued together (the scalar loop ran at 2.7 TB/s)
        const float4* g4 = reinterpret_cast<const float4*>(j.g + first);
        float4 x[OPT_CHUNK / 1024];
#pragma unroll
        for (int k = 0; k < OPT_CHUNK / 1024; ++k) x[k] = g4[threadIdx.x + 256 * k];
#pragma unroll
        for (int k = 0; k < OPT_CHUNK / 1024; ++k) s += (x[k].x * x[k].x + x[k].y * x[k].y) + (x[k].z * x[k].z + x[k].w * x[k].w);
    } else {
        for (long i = first + threadIdx.x; i < end; i += 256) { const float g = j.g[i]; s += g * g; }
    }
    s = wave_sum(s);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// norm[0] = sqrt(sum partial), norm[1] = clip coefficient min(1, max_norm / (norm + 1e-6))   (one workgroup, fixed order)
__global__ __launch_bounds__(256) void norm_final_kernel(const float* __restrict__ partial, int n, float max_norm, float* __restrict__ norm)
{
    float s = 0.f;
    // (eight loads in flight, same order of additions)
    for (int i0 = threadIdx.x; i0 < n; i0 += 8 * 256) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + 256 * u; v[u] = partial[i < n ? i : n - 1]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) if (i0 + 256 * u < n) s += v[u];
    }
    s = wave_sum(s);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
        norm[0] = t;
        const float c = max_norm / (t + 1e-6f);
        norm[1] = c < 1.f ? c : 1.f;
    }
}
__global__ __launch_bounds__(256) void multi_scale_kernel(const glowtts_opt_job* __restrict__ jobs, int njobs, const float* __restrict__ coef)
{
    long first;
    const glowtts_opt_job& j = find_job(jobs, njobs, blockIdx.x, first);
    const long end = min(first + OPT_CHUNK, (long)j.n);
    const float c = coef[0];
    if (c == 1.f) return;
    float* g = const_cast<float*>(j.g);
    for (long i = first + threadIdx.x; i < end; i += 256) g[i] *= c;
}
// hyper (device): [0] lr  [1] beta1  [2] beta2  [3] eps  [4] weight_decay  [5] step_size  [6] rectified (N_sma >= 5 ? 1 : 0)
__global__ __launch_bounds__(256) void radam_kernel(const glowtts_opt_job* __restrict__ jobs, int njobs, const float* __restrict__ hyper,
                                                    const float* __restrict__ gscale)
{
    long first;
    const glowtts_opt_job& j = find_job(jobs, njobs, blockIdx.x, first);
    const long end = min(first + OPT_CHUNK, (long)j.n);
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], step_size = hyper[5];
    const bool rect = hyper[6] != 0.f;
    const float gs = gscale ? gscale[0] : 1.f;
    if (end - first == OPT_CHUNK && ((reinterpret_cast<uintptr_t>(j.g + first) | reinterpret_cast<uintptr_t>(j.p + first) |
                                      reinterpret_cast<uintptr_t>(j.v + first) | reinterpret_cast<uintptr_t>(j.m + first)) & 15) == 0) {
        // full, 16-byte aligned chunk: float4 accesses, every load of a pass in flight before the arithmetic (same per-element formulas as below)
        const float4* g4 = reinterpret_cast<const float4*>(j.g + first);
        float4* p4 = reinterpret_cast<float4*>(j.p + first);
        float4* v4 = reinterpret_cast<float4*>(j.v + first);
        float4* m4 = reinterpret_cast<float4*>(j.m + first);
#pragma unroll
        for (int h = 0; h < OPT_CHUNK / 2048; ++h) {
            float4 G[2], Pp[2], V[2], M[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int q = threadIdx.x + 256 * (2 * h + k);
                G[k] = g4[q]; Pp[k] = p4[q]; V[k] = v4[q]; M[k] = m4[q];
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int q = threadIdx.x + 256 * (2 * h + k);
                float* gp = reinterpret_cast<float*>(&G[k]); float* pp = reinterpret_cast<float*>(&Pp[k]);
                float* vp = reinterpret_cast<float*>(&V[k]); float* mp = reinterpret_cast<float*>(&M[k]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g = gp[e] * gs;
                    float p = pp[e];
                    const float v = vp[e] * b2 + (1.f - b2) * g * g;
                    const float m = mp[e] * b1 + (1.f - b1) * g;
                    vp[e] = v; mp[e] = m;
                    if (wd != 0.f) p += -wd * lr * p;
                    if (rect) p += -step_size * lr * m / (sqrtf(v) + eps);
                    else      p += -step_size * lr * m;
                    pp[e] = p;
                }
                v4[q] = V[k]; m4[q] = M[k]; p4[q] = Pp[k];
            }
        }
        return;
    }
    for (long i = first + threadIdx.x; i < end; i += 256) {
        const float g = j.g[i] * gs;
        float p = j.p[i];
        const float v = j.v[i] * b2 + (1.f - b2) * g * g;              // Radam.py:59
        const float m = j.m[i] * b1 + (1.f - b1) * g;                  // Radam.py:60
        j.v[i] = v; j.m[i] = m;
        if (wd != 0.f) p += -wd * lr * p;                              // Radam.py:81-82
        if (rect) p += -step_size * lr * m / (sqrtf(v) + eps);         // Radam.py:85-87
        else      p += -step_size * lr * m;                            // Radam.py:88-89
        j.p[i] = p;
    }
}

}  // namespace

extern "C" int glowtts_multi_grad_norm(const glowtts_opt_job* dev_jobs, int njobs, int total_blocks, float max_norm, float* partial,
                                       float* norm_and_coef, void* stream)
{
    if (!dev_jobs || njobs < 1 || total_blocks < 1 || !partial || !norm_and_coef) return GLOWTTS_E_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(multi_sqnorm_kernel, dim3(total_blocks), dim3(256), 0, s, dev_jobs, njobs, partial);
    hipLaunchKernelGGL(norm_final_kernel, dim3(1), dim3(256), 0, s, partial, total_blocks, max_norm, norm_and_coef);
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

extern "C" int glowtts_multi_grad_scale(const glowtts_opt_job* dev_jobs, int njobs, int total_blocks, const float* coef, void* stream)
{
    if (!dev_jobs || njobs < 1 || total_blocks < 1 || !coef) return GLOWTTS_E_ARG;
    hipLaunchKernelGGL(multi_scale_kernel, dim3(total_blocks), dim3(256), 0, static_cast<hipStream_t>(stream), dev_jobs, njobs, coef);
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

extern "C" int glowtts_radam_step(const glowtts_opt_job* dev_jobs, int njobs, int total_blocks, const float* hyper, const float* grad_scale,
                                  void* stream)
{
    if (!dev_jobs || njobs < 1 || total_blocks < 1 || !hyper) return GLOWTTS_E_ARG;
    hipLaunchKernelGGL(radam_kernel, dim3(total_blocks), dim3(256), 0, static_cast<hipStream_t>(stream), dev_jobs, njobs, hyper, grad_scale);
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

extern "C" int glowtts_opt_chunk(void) { return OPT_CHUNK; }

// out[i] = partial[0][i] + partial[1][i] + ... + partial[S - 1][i]  (fixed order; n % 4 == 0, 16-byte aligned): the row splits of the text encoder's
// weight gradients (conv_fn.WgradTape) summed into the gradients' arena
namespace {
__global__ __launch_bounds__(256) void sum_slices_kernel(const float4* __restrict__ partial, float4* __restrict__ out, int S, long n4)
{
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= n4) return;
    float4 a = partial[i];
    for (int s = 1; s < S; ++s) { const float4 b = partial[(long)s * n4 + i]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    out[i] = a;
}
}
extern "C" int glowtts_sum_slices(const float* partial, float* out, int S, int64_t n, void* stream)
{
    if (!partial || !out || S < 1 || n < 4 || (n & 3) || ((reinterpret_cast<uintptr_t>(partial) | reinterpret_cast<uintptr_t>(out)) & 15)) return GLOWTTS_E_ARG;
    const long n4 = n / 4;
    hipLaunchKernelGGL(sum_slices_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float4*>(partial), reinterpret_cast<float4*>(out), S, n4);
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

// The same for a partial image that is cut into up to GLOWTTS_SUM_MAX_SEGS destination tensors (the decoder's one-tap weight gradients: several classes, some of them
// views of the returned gradient arena): element i of [off_k, off_k + n_k) goes to dst_k[i - off_k].  The table travels in the argument segment.
namespace {
struct sum_seg_table { glowtts_sum_seg seg[GLOWTTS_SUM_MAX_SEGS]; };
__global__ __launch_bounds__(256) void sum_slices_seg_kernel(const float4* __restrict__ partial, int S, long stride4, long n4, const sum_seg_table tab, int nseg)
{
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= n4) return;
    float4 a = partial[i];
    for (int s = 1; s < S; ++s) { const float4 b = partial[(long)s * stride4 + i]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    int k = 0;
#pragma unroll
    for (int j = 1; j < GLOWTTS_SUM_MAX_SEGS; ++j) k = (j < nseg && tab.seg[j].off <= 4 * i) ? j : k;
    reinterpret_cast<float4*>(tab.seg[k].dst)[(4 * i - tab.seg[k].off) >> 2] = a;
}
}
extern "C" int glowtts_sum_slices_seg(const float* partial, int S, int64_t stride, const glowtts_sum_seg* segs, int nseg, void* stream)
{
    if (!partial || S < 1 || !segs || nseg < 1 || nseg > GLOWTTS_SUM_MAX_SEGS || (stride & 3) || (reinterpret_cast<uintptr_t>(partial) & 15)) return GLOWTTS_E_ARG;
    sum_seg_table tab;
    int64_t end = 0;
    for (int k = 0; k < nseg; ++k) {                           // segments tile [0, n) in ascending order
        if (!segs[k].dst || segs[k].off != end || segs[k].n < 4 || (segs[k].n & 3) || (reinterpret_cast<uintptr_t>(segs[k].dst) & 15)) return GLOWTTS_E_ARG;
        tab.seg[k] = segs[k];
        end += segs[k].n;
    }
    for (int k = nseg; k < GLOWTTS_SUM_MAX_SEGS; ++k) tab.seg[k] = segs[0];
    if (end > stride) return GLOWTTS_E_ARG;
    const long n4 = end / 4;
    hipLaunchKernelGGL(sum_slices_seg_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float4*>(partial), S, (long)(stride / 4), n4, tab, nseg);
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}


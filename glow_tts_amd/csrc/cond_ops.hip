// Per-utterance conditioning of the WaveNet gates (Modules.py:832-845, 863-866): cond[b][n] = Speaker_l(spk_b)[n] + Prosody_l(pro_b)[n] for all
// n = (flow, layer, gate channel) - F * L * 2H = 18 432 weight-normalised 1x1 convs of D (= 256) inputs, one [B, D] x [D, N] product per kind.
// As torch ops (weight norm + einsum + bias add, rocBLAS serves the M = 32 product at 80 us) this sat at the head of the decoder's chain in every
// conditioned mode; its backward was three more skinny GEMMs and the weight-norm backward at the tail.  Both directions are row-local in n:
// A workgroup owns CT_ROWS = 32 rows n; v[n][:] is read once, coalesced, into a padded LDS tile next to the utterances' vectors, and every product is
// register-tiled fp32 FMAs on 16-byte LDS reads (the first version - one wavefront per row, 6 shuffles per (row, utterance) - ran 330 us):
//   forward : ||v|| per row while staging; thread (row r, utterance set s) keeps <v[r], vec[s + 8 j]> for j < NJ; out[b][n] = bias + g / ||v|| dot,
//             rows of one utterance written as 128-byte runs.
//   backward: d w[r][:] = sum_b d cond[b][r] vec[b][:] (thread: row x 32 columns), the weight-norm backward applied in place
//             (d g = <d w, v> / ||v||, d v = g / ||v|| (d w - <d w, v> v / ||v||^2), d bias = sum_b d cond[b][n]); the gradient of the
//             vectors, sum_n d cond[b][n] w[n][:], per workgroup into partial[wg][b][:] and summed by a second launch in a fixed order
//             (deterministic, no atomics).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/glowtts_hip.h"
#include "launch_log.h"
#include "device_common.h"

namespace {

constexpr int CO_MAXB = 64;          // utterances per launch
constexpr int CO_MAXD = 512;         // inputs per conv
constexpr int CT_ROWS = 32;          // rows n per workgroup
constexpr int CT_NT = 256;

__device__ __forceinline__ float co_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// LDS layout shared by both kernels: sv [CT_ROWS][D + 4] (rows of v), svec [BP][D + 4] (zero rows past B), ssc [CT_ROWS] / sinv [CT_ROWS]
// (row pitch D + 4 floats: consecutive rows start 4 banks apart, 16-byte reads of 8 consecutive rows are conflict-free)
__device__ __forceinline__ void co_stage(const float* __restrict__ v, const float* __restrict__ vec, float* sv, float* svec, float* ssq, int n0, int N, int D, int B, int BP)
{
    const int P = D + 4, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < BP * (D / 4); i += CT_NT) {
        const int b = i / (D / 4), q = i - b * (D / 4);
        const float4 x = b < B ? *reinterpret_cast<const float4*>(vec + (int64_t)b * D + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(svec + b * P + 4 * q) = x;
    }
    // a wave stages CT_ROWS / 4 rows; a row = D / 4 float4 pieces over the lanes
    for (int rr = 0; rr < CT_ROWS / 4; ++rr) {
        const int r = wave * (CT_ROWS / 4) + rr, n = n0 + r;
        float s = 0.f;
        for (int q = lane; q < D / 4; q += 64) {
            const float4 x = n < N ? *reinterpret_cast<const float4*>(v + (int64_t)n * D + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(sv + r * P + 4 * q) = x;
            s += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
        }
        s = co_wave_sum(s);
        if (lane == 0) ssq[r] = s;
    }
}

// grid: ceil(N / CT_ROWS).  NJ = ceil(B / 8): thread (s = tid / 32, r = tid % 32) owns utterances s + 8 j.
template <int NJ>
__global__ __launch_bounds__(CT_NT) void cond_fwd_kernel(const float* __restrict__ v, const float* __restrict__ g, const float* __restrict__ bias,
                                                         const float* __restrict__ vec, float* __restrict__ out, float* __restrict__ inv_out,
                                                         int N, int D, int B, int accumulate)
{
    extern __shared__ __attribute__((aligned(16))) float co_smem[];
    const int P = D + 4, BP = 8 * NJ;
    float* const sv = co_smem;
    float* const svec = sv + CT_ROWS * P;
    float* const ssq = svec + BP * P;
    const int n0 = blockIdx.x * CT_ROWS;
    co_stage(v, vec, sv, svec, ssq, n0, N, D, B, BP);
    __syncthreads();
    const int r = threadIdx.x & 31, s = threadIdx.x >> 5, n = n0 + r;
    float acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = 0.f;
    const float* a = sv + r * P;
    const float* w = svec + s * P;
    for (int q = 0; q < D; q += 4) {
        const float4 x = *reinterpret_cast<const float4*>(a + q);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const float4 y = *reinterpret_cast<const float4*>(w + 8 * j * P + q);
            acc[j] += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
        }
    }
    if (n >= N) return;
    const float inv = 1.f / sqrtf(ssq[r]), sc = g[n] * inv, bn = bias ? bias[n] : 0.f;
    if (inv_out && s == 0) inv_out[n] = inv;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int b = s + 8 * j;
        if (b < B) {
            float* o = out + (int64_t)b * N + n;
            const float val = bn + sc * acc[j];
            *o = accumulate ? *o + val : val;
        }
    }
}

// grid: ceil(N / CT_ROWS).  dcond [B][ldd], partial [grid][B][D].  DJ = D / 64 (thread: 4 DJ columns ... see below), B <= 64.
// Phase 1: thread (r = tid % 32, cs = tid / 32) owns d w[r][c] for the DC = D / 8 columns c = 4 cs + 32 m + e (m < DC / 4, e < 4).
// Phase 2: thread (cs2 = tid % 32 ... ) see code.
template <int DC>
__global__ __launch_bounds__(CT_NT) void cond_bwd_kernel(const float* __restrict__ dcond, int64_t ldd, const float* __restrict__ v, const float* __restrict__ g,
                                                         const float* __restrict__ inv_in, const float* __restrict__ vec, float* __restrict__ dv,
                                                         float* __restrict__ dg, float* __restrict__ db, float* __restrict__ partial, int N, int B, int BP)
{
    constexpr int D = 8 * DC, P = D + 4, M = DC / 4;
    extern __shared__ __attribute__((aligned(16))) float cb_smem[];
    float* const sv = cb_smem;                               // [CT_ROWS][P]: v rows, then (in place) w rows
    float* const svec = sv + CT_ROWS * P;                    // [BP][P]
    float* const ssq = svec + BP * P;                        // [CT_ROWS]
    float* const sd = ssq + CT_ROWS;                         // [BP][CT_ROWS + 1]  d cond of the tile, utterance-major (zero rows past B)
    float* const sdot = sd + BP * (CT_ROWS + 1);             // [8][CT_ROWS] partial <d w, v>
    const int n0 = blockIdx.x * CT_ROWS, tid = threadIdx.x;
    co_stage(v, vec, sv, svec, ssq, n0, N, D, B, BP);
    for (int i = tid; i < BP * CT_ROWS; i += CT_NT) {
        const int b = i / CT_ROWS, r = i - b * CT_ROWS;
        sd[b * (CT_ROWS + 1) + r] = (b < B && n0 + r < N) ? dcond[(int64_t)b * ldd + n0 + r] : 0.f;
    }
    __syncthreads();
    const int r = tid & 31, cs = tid >> 5, n = n0 + r;
    const bool ok = n < N;
    // ---- phase 1: d w, the weight-norm backward ----
    float dw[M][4];
#pragma unroll
    for (int m = 0; m < M; ++m) { dw[m][0] = dw[m][1] = dw[m][2] = dw[m][3] = 0.f; }
    float bsum = 0.f;
    for (int b = 0; b < B; ++b) {
        const float d = sd[b * (CT_ROWS + 1) + r];
        bsum += d;
        const float* w = svec + b * P + 4 * cs;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const float4 y = *reinterpret_cast<const float4*>(w + 32 * m);
            dw[m][0] += d * y.x; dw[m][1] += d * y.y; dw[m][2] += d * y.z; dw[m][3] += d * y.w;
        }
    }
    float dot = 0.f;
    float4 x[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        x[m] = *reinterpret_cast<const float4*>(sv + r * P + 4 * cs + 32 * m);
        dot += dw[m][0] * x[m].x + dw[m][1] * x[m].y + dw[m][2] * x[m].z + dw[m][3] * x[m].w;
    }
    sdot[cs * CT_ROWS + r] = dot;
    __syncthreads();
    dot = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) dot += sdot[k * CT_ROWS + r];
    const float inv = ok ? inv_in[n] : 0.f, gn = ok ? g[n] : 0.f;
    const float sc = gn * inv, c2 = dot * inv * inv;         // d v = sc (d w - <d w, v> / ||v||^2 v)
#pragma unroll
    for (int m = 0; m < M; ++m) {
        if (ok) *reinterpret_cast<float4*>(dv + (int64_t)n * D + 4 * cs + 32 * m) =
            make_float4(sc * (dw[m][0] - c2 * x[m].x), sc * (dw[m][1] - c2 * x[m].y), sc * (dw[m][2] - c2 * x[m].z), sc * (dw[m][3] - c2 * x[m].w));
        // the tile becomes w = sc v for phase 2 (every thread rewrites exactly the pieces it read)
        *reinterpret_cast<float4*>(sv + r * P + 4 * cs + 32 * m) = make_float4(sc * x[m].x, sc * x[m].y, sc * x[m].z, sc * x[m].w);
    }
    if (ok && cs == 0) { dg[n] = dot * inv; if (db) db[n] = bsum; }
    __syncthreads();
    // ---- phase 2: partial[wg][b][c] = sum_r d cond[b][r] w[r][c]: thread (bs = tid % 8 ... ) owns utterances b = bs + 8 j and the 4 columns 4 q ..
    // with q = tid / 8 + 32 m2: BP / 8 utterances x (D / 128) column groups ----
    if (!partial) return;
    const int bs = tid & 7, q0 = tid >> 3;
    for (int j = 0; j < BP / 8; ++j) {
        const int b = bs + 8 * j;
        const float* dd = sd + b * (CT_ROWS + 1);
#pragma unroll
        for (int m2 = 0; m2 < D / 128; ++m2) {
            const int c = 4 * (q0 + 32 * m2);
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            for (int rr = 0; rr < CT_ROWS; ++rr) {
                const float d = dd[rr];
                const float4 w = *reinterpret_cast<const float4*>(sv + rr * P + c);
                a0 += d * w.x; a1 += d * w.y; a2 += d * w.z; a3 += d * w.w;
            }
            if (b < B) *reinterpret_cast<float4*>(partial + ((int64_t)blockIdx.x * B + b) * D + c) = make_float4(a0, a1, a2, a3);
        }
    }
}

// dvec[i] = sum_wg partial[wg][i], i over B * D: a workgroup owns 64 elements, its 16 waves split the workgroups, fixed order within and across waves
__global__ __launch_bounds__(1024) void cond_dvec_kernel(const float* __restrict__ partial, float* __restrict__ dvec, int nwg, int64_t n)
{
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = blockIdx.x * 64L + lane;
    float s = 0.f;
    if (i < n) for (int w = wave; w < nwg; w += 16) s += partial[(int64_t)w * n + i];
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][lane];
        dvec[i] = t;
    }
}

}  // namespace

// LDS bytes of the two kernels for (D, B): the shapes the launches below accept are exactly those whose tiles fit 160 KiB
static size_t cond_fwd_lds(int D, int B) { const int NJ = (B + 7) / 8; return ((size_t)(CT_ROWS + 8 * NJ) * (D + 4) + CT_ROWS) * sizeof(float); }
static size_t cond_bwd_lds(int D, int B)
{
    const int BP = (B + 7) / 8 * 8;
    return ((size_t)(CT_ROWS + BP) * (D + 4) + CT_ROWS + (size_t)BP * (CT_ROWS + 1) + 8 * CT_ROWS) * sizeof(float);
}

extern "C" int glowtts_cond_linear_supported(int N, int D, int B)
{
    if (N < 1 || B < 1 || B > CO_MAXB) return 0;
    if (D != 128 && D != 256 && D != 384 && D != 512) return 0;
    return cond_fwd_lds(D, B) <= 160 * 1024 && cond_bwd_lds(D, B) <= 160 * 1024;
}

extern "C" int glowtts_cond_linear_fwd(const float* v, const float* g, const float* bias, const float* vec, float* out, float* inv_out,
                                       int N, int D, int B, int accumulate, void* stream)
{
    if (!v || !g || !vec || !out || N < 1 || D < 4 || (D & 3) || D > CO_MAXD || B < 1 || B > CO_MAXB) return GLOWTTS_E_ARG;
    GLOWTTS_NOTE_STATIC("cond_linear_fwd");
    const dim3 grid((N + CT_ROWS - 1) / CT_ROWS);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int NJ = (B + 7) / 8;
    const size_t lds = cond_fwd_lds(D, B);
    if (lds > 160 * 1024) return GLOWTTS_E_ARG;
#define CF_LAUNCH(J) do { static bool done = false; if (!done) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(&cond_fwd_kernel<J>), \
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return GLOWTTS_E_LAUNCH; done = true; } \
        hipLaunchKernelGGL(cond_fwd_kernel<J>, grid, dim3(CT_NT), lds, s, v, g, bias, vec, out, inv_out, N, D, B, accumulate); } while (0)
    switch (NJ) {
        case 1: CF_LAUNCH(1); break; case 2: CF_LAUNCH(2); break; case 3: CF_LAUNCH(3); break; case 4: CF_LAUNCH(4); break;
        case 5: CF_LAUNCH(5); break; case 6: CF_LAUNCH(6); break; case 7: CF_LAUNCH(7); break; default: CF_LAUNCH(8); break;
    }
#undef CF_LAUNCH
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

extern "C" int64_t glowtts_cond_linear_bwd_scratch_floats(int N, int D, int B) { return (int64_t)((N + CT_ROWS - 1) / CT_ROWS) * B * D; }

extern "C" int glowtts_cond_linear_bwd(const float* dcond, int64_t ldd, const float* v, const float* g, const float* inv, const float* vec,
                                       float* dv, float* dg, float* dbias, float* dvec, float* scratch, int N, int D, int B, void* stream)
{
    if (!dcond || !v || !g || !inv || !vec || !dv || !dg || !scratch || N < 1 || B < 1 || B > CO_MAXB || ldd < N) return GLOWTTS_E_ARG;
    if (D != 128 && D != 256 && D != 384 && D != 512) return GLOWTTS_E_ARG;         // (phase 2 tiles the columns in groups of 128)
    const int BP = (B + 7) / 8 * 8;
    const size_t lds = cond_bwd_lds(D, B);
    if (lds > 160 * 1024) return GLOWTTS_E_ARG;
    const int nwg = (N + CT_ROWS - 1) / CT_ROWS;
    hipStream_t s = static_cast<hipStream_t>(stream);
    GLOWTTS_NOTE_STATIC("cond_linear_bwd");
    float* part = dvec ? scratch : nullptr;
#define CB_LAUNCH(DC) do { static bool done = false; if (!done) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(&cond_bwd_kernel<DC>), \
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return GLOWTTS_E_LAUNCH; done = true; } \
        hipLaunchKernelGGL(cond_bwd_kernel<DC>, dim3(nwg), dim3(CT_NT), lds, s, dcond, ldd, v, g, inv, vec, dv, dg, dbias, part, N, B, BP); } while (0)
    switch (D) { case 128: CB_LAUNCH(16); break; case 256: CB_LAUNCH(32); break; case 384: CB_LAUNCH(48); break; default: CB_LAUNCH(64); break; }
#undef CB_LAUNCH
    if (dvec) {
        const int64_t n = (int64_t)B * D;
        hipLaunchKernelGGL(cond_dvec_kernel, dim3((int)((n + 63) / 64)), dim3(1024), 0, s, scratch, dvec, nwg, n);
    }
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

// The conditioning gradient's fixed-point accumulators (device_common.h fx_atomic_add) -> float; a poisoned accumulator (non-finite / out-of-range addend) reads NaN.
namespace {
__global__ __launch_bounds__(256) void fx_to_float_kernel(const long long* __restrict__ acc, float* __restrict__ out, long n)
{
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long long a = acc[i];
        const bool bad = a >= GLOWTTS_FX_POISON || a <= -GLOWTTS_FX_POISON;
        out[i] = bad ? __int_as_float(0x7FC00000) : (float)((double)a * (1.0 / 1099511627776.0));
    }
}
}  // namespace
extern "C" int glowtts_fx_to_float(const int64_t* acc, float* out, int64_t n, void* stream)
{
    if (!acc || !out || n < 1) return GLOWTTS_E_ARG;
    const long g = (n + 255) / 256;
    GLOWTTS_NOTE_STATIC("fx_to_float");
    hipLaunchKernelGGL(fx_to_float_kernel, dim3((int)(g > 4096 ? 4096 : g)), dim3(256), 0, static_cast<hipStream_t>(stream), reinterpret_cast<const long long*>(acc), out, (long)n);
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

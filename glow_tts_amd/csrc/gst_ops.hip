// Style-token tail of the GST prosody encoder (Modules.py:345-355, 371-385) for gfx950: behind the GRU, the reference takes the state at each utterance's
// last valid step, and attends with it (RPR_MHA.py:69-128 without relative positions or masks, one query) over tanh(gst_Tokens) - per utterance a 128 -> 256
// projection, four heads x 128 tokens of scores, a softmax, 256 x 128 products and a 256 -> 256 projection: ~0.2 MFLOP that PyTorch spreads over ~35 launches
// forward and ~45 backward (index arithmetic, four 1x1 convs as GEMMs, bias adds, softmax, bmm), all of them on the chain in front of the flow decoder
// (forward) or in front of the conv stack's backward.  Here:
//   gst_tanh_kernel / gst_kv_kernel   tanh(T), then K = Wk tanh(T) + bk, V = Wv tanh(T) + bv   [C][NT]   (batch independent, once per step)
//   gst_attn_fwd_kernel   one workgroup per utterance, one thread per channel: gather, query, scores, softmax, context, projection
//   gst_attn_bwd_kernel   its backward per utterance: d(GRU states) (zeros but the gathered step) and the per-utterance vectors of the parameter gradients
//   gst_grads1_kernel     the sums over the batch: d Wp, d bp, d Wq, d bq, d K, d V (outer products of those vectors)
//   gst_grads2_kernel     d Wk, d bk, d Wv, d bv and d gst_Tokens through tanh
// fp32 FMA throughout (the arithmetic is negligible; both arithmetic modes take these kernels).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/glowtts_hip.h"
#include "launch_log.h"

namespace {

constexpr int GST_MAX_NT = 256;     // style tokens
constexpr int GST_MAX_H = 8;        // heads

__device__ __forceinline__ int gst_last_step(const int64_t* lengths, int b, int stride_prod, int Tp)
{
    long n = (lengths[b] + stride_prod - 1) / stride_prod - 1;          // Modules.py:373  ceil(length / prod(strides)) - 1
    n = n < 0 ? 0 : n;
    return (int)(n >= Tp ? Tp - 1 : n);
}

// TT = tanh(tokens) once (the keys of the attention; kept for the backward): every later kernel reads it instead of re-evaluating 2 x 8.4 M tanhf
__global__ __launch_bounds__(256) void gst_tanh_kernel(const float* __restrict__ tokens, float* __restrict__ TT, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) TT[i] = tanhf(tokens[i]);
}

// block = one output channel c, thread = token t: K[c][t] = bk[c] + sum_i Wk[c][i] TT[i][t] (the weight row is wave-uniform, TT rows are coalesced)
__global__ __launch_bounds__(GST_MAX_NT) void gst_kv_kernel(const float* __restrict__ TT /* [I][NT] */, const float* __restrict__ Wk, const float* __restrict__ bk,
                                                           const float* __restrict__ Wv, const float* __restrict__ bv, float* __restrict__ K, float* __restrict__ V,
                                                           int I, int NT)
{
    const int c = blockIdx.x, t = threadIdx.x;
    if (t >= NT) return;
    float ak[4] = {bk ? bk[c] : 0.f, 0.f, 0.f, 0.f}, av[4] = {bv ? bv[c] : 0.f, 0.f, 0.f, 0.f};
    const float* wk = Wk + (long)c * I;
    const float* wv = Wv + (long)c * I;
    int i = 0;
    for (; i + 4 <= I; i += 4) {
        float tt[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) tt[u] = TT[(long)(i + u) * NT + t];
#pragma unroll
        for (int u = 0; u < 4; ++u) { ak[u] += wk[i + u] * tt[u]; av[u] += wv[i + u] * tt[u]; }
    }
    for (; i < I; ++i) { const float tt = TT[(long)i * NT + t]; ak[0] += wk[i] * tt; av[0] += wv[i] * tt; }
    K[(long)c * NT + t] = (ak[0] + ak[1]) + (ak[2] + ak[3]);
    V[(long)c * NT + t] = (av[0] + av[1]) + (av[2] + av[3]);
}

// dot products with several loads in flight: `n` terms, element i of the two operands at a[i * sa] and b[i * sb]
__device__ __forceinline__ float gst_dot(const float* __restrict__ a, long sa, const float* __restrict__ b, long sb, int n)
{
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int i = 0;
    for (; i + 8 <= n; i += 8) {
        float x[8], y[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { x[u] = a[(long)(i + u) * sa]; y[u] = b[(long)(i + u) * sb]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] += x[u] * y[u];
    }
    for (; i < n; ++i) acc[0] += a[(long)i * sa] * b[(long)i * sb];
    return ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
}

struct gst_dims { int B, Tp, G, C, H, NT, I, stride_prod; };

// keep [B][2 C + H NT + G]: q | a | p | h   (what the backward needs)
__global__ __launch_bounds__(1024) void gst_attn_fwd_kernel(const float* __restrict__ hs /* [B][Tp][G] */, const int64_t* __restrict__ lengths,
                                                            const float* __restrict__ Wq, const float* __restrict__ bq, const float* __restrict__ K,
                                                            const float* __restrict__ V, const float* __restrict__ Wp, const float* __restrict__ bp,
                                                            float* __restrict__ out /* [B][C] */, float* __restrict__ keep, const gst_dims d)
{
    extern __shared__ float gs[];                       // h [G] | q [C] | sc [H][NT] | a [C]
    float* h = gs;
    float* q = h + d.G;
    float* sc = q + d.C;
    float* a = sc + d.H * d.NT;
    const int b = blockIdx.x, c = threadIdx.x, D = d.C / d.H;
    const int step = gst_last_step(lengths, b, d.stride_prod, d.Tp);
    for (int g = c; g < d.G; g += d.C) h[g] = hs[((long)b * d.Tp + step) * d.G + g];
    __syncthreads();
    {
        q[c] = (bq ? bq[c] : 0.f) + gst_dot(Wq + (long)c * d.G, 1, h, 1, d.G);
    }
    __syncthreads();
    const float scale = rsqrtf((float)D);
    for (int j = c; j < d.H * d.NT; j += d.C) {
        const int hh = j / d.NT, t = j - hh * d.NT;
        sc[j] = gst_dot(q + hh * D, 1, K + (long)(hh * D) * d.NT + t, d.NT, D) * scale;
    }
    __syncthreads();
    {                                                   // softmax over the tokens, one wavefront per head (round robin)
        const int wave = c >> 6, lane = c & 63, nw = d.C >> 6;
        for (int hh = wave; hh < d.H; hh += nw) {
            float m = -3.0e38f;
            for (int t = lane; t < d.NT; t += 64) m = fmaxf(m, sc[hh * d.NT + t]);
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
            float s = 0.f;
            for (int t = lane; t < d.NT; t += 64) { const float e = expf(sc[hh * d.NT + t] - m); sc[hh * d.NT + t] = e; s += e; }
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            const float inv = 1.f / s;
            for (int t = lane; t < d.NT; t += 64) sc[hh * d.NT + t] *= inv;
        }
    }
    __syncthreads();
    {
        const int hh = c / D;
        a[c] = gst_dot(sc + hh * d.NT, 1, V + (long)c * d.NT, 1, d.NT);
    }
    __syncthreads();
    {
        out[(long)b * d.C + c] = (bp ? bp[c] : 0.f) + gst_dot(Wp + (long)c * d.C, 1, a, 1, d.C);
    }
    if (keep) {
        float* kb = keep + (long)b * (2 * d.C + d.H * d.NT + d.G);
        kb[c] = q[c];
        kb[d.C + c] = a[c];
        for (int j = c; j < d.H * d.NT; j += d.C) kb[2 * d.C + j] = sc[j];
        for (int g = c; g < d.G; g += d.C) kb[2 * d.C + d.H * d.NT + g] = h[g];
    }
}

// vec [B][2 C + H NT]: dq | da | ds      (with keep: the operands of the parameter-gradient sums);  dhs [B][Tp][G] fully written
__global__ __launch_bounds__(1024) void gst_attn_bwd_kernel(const float* __restrict__ dout /* [B][C] */, const float* __restrict__ keep, const int64_t* __restrict__ lengths,
                                                            const float* __restrict__ Wq, const float* __restrict__ K, const float* __restrict__ V,
                                                            const float* __restrict__ Wp, float* __restrict__ dhs, float* __restrict__ vec, const gst_dims d)
{
    extern __shared__ float gs[];                       // dy [C] | da [C] | p [H][NT] | ds [H][NT] | dq [C] | red [H]
    float* dy = gs;
    float* da = dy + d.C;
    float* p = da + d.C;
    float* ds = p + d.H * d.NT;
    float* dq = ds + d.H * d.NT;
    float* red = dq + d.C;
    const int b = blockIdx.x, c = threadIdx.x, D = d.C / d.H;
    const float* kb = keep + (long)b * (2 * d.C + d.H * d.NT + d.G);
    const float qc = kb[c];
    dy[c] = dout[(long)b * d.C + c];
    for (int j = c; j < d.H * d.NT; j += d.C) p[j] = kb[2 * d.C + j];
    __syncthreads();
    {                                                   // d a = Wp^T d y   (column c of Wp: consecutive threads read consecutive addresses)
        da[c] = gst_dot(Wp + c, d.C, dy, 1, d.C);
    }
    __syncthreads();
    for (int j = c; j < d.H * d.NT; j += d.C) {         // d p[h][t] = sum_d d a[hD + d] V[hD + d][t]
        const int hh = j / d.NT, t = j - hh * d.NT;
        ds[j] = gst_dot(da + hh * D, 1, V + (long)(hh * D) * d.NT + t, d.NT, D);
    }
    __syncthreads();
    {                                                   // softmax backward, then the 1 / sqrt(D) of the scores
        const int wave = c >> 6, lane = c & 63, nw = d.C >> 6;
        for (int hh = wave; hh < d.H; hh += nw) {
            float s = 0.f;
            for (int t = lane; t < d.NT; t += 64) s += p[hh * d.NT + t] * ds[hh * d.NT + t];
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if (lane == 0) red[hh] = s;
        }
    }
    __syncthreads();
    const float scale = rsqrtf((float)D);
    for (int j = c; j < d.H * d.NT; j += d.C) { const int hh = j / d.NT; ds[j] = p[j] * (ds[j] - red[hh]) * scale; }
    __syncthreads();
    {                                                   // d q[c] = sum_t d s[h(c)][t] K[c][t]
        const int hh = c / D;
        dq[c] = gst_dot(ds + hh * d.NT, 1, K + (long)c * d.NT, 1, d.NT);
    }
    __syncthreads();
    const int step = gst_last_step(lengths, b, d.stride_prod, d.Tp);
    for (int g = c; g < d.G; g += d.C) {                // d h = Wq^T d q at the gathered step, zeros elsewhere
        const float acc = gst_dot(Wq + g, d.G, dq, 1, d.C);
        for (int s = 0; s < d.Tp; ++s) dhs[((long)b * d.Tp + s) * d.G + g] = s == step ? acc : 0.f;
    }
    float* vb = vec + (long)b * (2 * d.C + d.H * d.NT);
    vb[c] = dq[c];
    vb[d.C + c] = da[c];
    for (int j = c; j < d.H * d.NT; j += d.C) vb[2 * d.C + j] = ds[j];
    (void)qc;
}

// sums over the batch (fixed order).  blockIdx.y selects the tensor: 0 d Wp [C][C], 1 d Wq [C][G], 2 d K [C][NT], 3 d V [C][NT], 4 biases d bp | d bq [2 C]
__global__ __launch_bounds__(256) void gst_grads1_kernel(const float* __restrict__ dout, const float* __restrict__ keep, const float* __restrict__ vec,
                                                         float* __restrict__ dWp, float* __restrict__ dbp, float* __restrict__ dWq, float* __restrict__ dbq,
                                                         float* __restrict__ dK, float* __restrict__ dV, const gst_dims d)
{
    const int D = d.C / d.H;
    const long ks = 2 * d.C + d.H * d.NT + d.G, vs = 2 * d.C + d.H * d.NT;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    switch (blockIdx.y) {
        case 0: {                                       // d Wp[c'][c] = sum_b dy_b[c'] a_b[c]
            if (i >= (long)d.C * d.C) return;
            const int r = (int)(i / d.C), c = (int)(i - (long)r * d.C);
            dWp[i] = gst_dot(dout + r, d.C, keep + d.C + c, ks, d.B);
            return;
        }
        case 1: {                                       // d Wq[c][g] = sum_b dq_b[c] h_b[g]
            if (i >= (long)d.C * d.G) return;
            const int r = (int)(i / d.G), g = (int)(i - (long)r * d.G);
            dWq[i] = gst_dot(vec + r, vs, keep + 2 * d.C + d.H * d.NT + g, ks, d.B);
            return;
        }
        case 2: {                                       // d K[c][t] = sum_b q_b[c] ds_b[h(c)][t]
            if (i >= (long)d.C * d.NT) return;
            const int c = (int)(i / d.NT), t = (int)(i - (long)c * d.NT), hh = c / D;
            dK[i] = gst_dot(keep + c, ks, vec + 2 * d.C + hh * d.NT + t, vs, d.B);
            return;
        }
        case 3: {                                       // d V[c][t] = sum_b da_b[c] p_b[h(c)][t]
            if (i >= (long)d.C * d.NT) return;
            const int c = (int)(i / d.NT), t = (int)(i - (long)c * d.NT), hh = c / D;
            dV[i] = gst_dot(vec + d.C + c, vs, keep + 2 * d.C + hh * d.NT + t, ks, d.B);
            return;
        }
        default: {
            if (i >= 2L * d.C) return;
            float acc = 0.f;
            if (i < d.C) { for (int b = 0; b < d.B; ++b) acc += dout[(long)b * d.C + i]; if (dbp) dbp[i] = acc; }
            else { const long c = i - d.C; for (int b = 0; b < d.B; ++b) acc += vec[b * vs + c]; if (dbq) dbq[c] = acc; }
            return;
        }
    }
}

// blockIdx.y: 0 d Wk [C][I] (+ d bk), 1 d Wv [C][I] (+ d bv), 2 d tokens [I][NT]
__global__ __launch_bounds__(256) void gst_grads2_kernel(const float* __restrict__ TT, const float* __restrict__ Wk, const float* __restrict__ Wv,
                                                         const float* __restrict__ dK, const float* __restrict__ dV, float* __restrict__ dWk, float* __restrict__ dbk,
                                                         float* __restrict__ dWv, float* __restrict__ dbv, float* __restrict__ dtok, const gst_dims d)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.y < 2) {
        const float* dX = blockIdx.y == 0 ? dK : dV;
        float* dW = blockIdx.y == 0 ? dWk : dWv;
        float* db = blockIdx.y == 0 ? dbk : dbv;
        if (i >= (long)d.C * d.I) return;
        const int c = (int)(i / d.I), ii = (int)(i - (long)c * d.I);
        dW[i] = gst_dot(dX + (long)c * d.NT, 1, TT + (long)ii * d.NT, 1, d.NT);
        if (db && ii == 0) { float bs = 0.f; for (int t = 0; t < d.NT; ++t) bs += dX[(long)c * d.NT + t]; db[c] = bs; }
        return;
    }
    if (i >= (long)d.I * d.NT) return;
    const int ii = (int)(i / d.NT), t = (int)(i - (long)ii * d.NT);
    const float acc = gst_dot(Wk + ii, d.I, dK + t, d.NT, d.C) + gst_dot(Wv + ii, d.I, dV + t, d.NT, d.C);
    const float tt = TT[i];
    dtok[i] = acc * (1.f - tt * tt);
}

bool gst_ok(int B, int Tp, int G, int C, int H, int NT, int I)
{
    return B >= 1 && Tp >= 1 && G >= 1 && G <= 4096 && C >= 64 && C <= 1024 && (C % 64) == 0 && H >= 1 && H <= GST_MAX_H && (C % H) == 0 && NT >= 1 && NT <= GST_MAX_NT &&
           I >= 1 && (size_t)(G + 3 * C + 2 * H * NT + H) * sizeof(float) <= 64 * 1024;
}

}  // namespace

extern "C" int glowtts_gst_supported(int B, int Tp, int G, int C, int H, int NT, int I) { return gst_ok(B, Tp, G, C, H, NT, I) ? 1 : 0; }

extern "C" int64_t glowtts_gst_keep_floats(int B, int G, int C, int H, int NT, int I) { return (int64_t)B * (2 * C + H * NT + G) + (int64_t)I * NT; }

extern "C" int glowtts_gst_fwd(const float* hs, const int64_t* lengths, int stride_prod, const float* tokens, const float* Wq, const float* bq, const float* Wk,
                               const float* bk, const float* Wv, const float* bv, const float* Wp, const float* bp, float* K, float* V, float* out, float* keep,
                               int B, int Tp, int G, int C, int H, int NT, int I, void* stream)
{
    if (!hs || !lengths || !tokens || !Wq || !Wk || !Wv || !Wp || !K || !V || !out || !keep || stride_prod < 1 || !gst_ok(B, Tp, G, C, H, NT, I)) return GLOWTTS_E_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const gst_dims d = {B, Tp, G, C, H, NT, I, stride_prod};
    GLOWTTS_NOTE_STATIC("gst_fwd");
    float* TT = keep + (long)B * (2 * C + H * NT + G);               // tanh(tokens), kept behind the per-utterance vectors
    hipLaunchKernelGGL(gst_tanh_kernel, dim3((I * NT + 255) / 256), dim3(256), 0, s, tokens, TT, I * NT);
    hipLaunchKernelGGL(gst_kv_kernel, dim3(C), dim3((NT + 63) / 64 * 64), 0, s, TT, Wk, bk, Wv, bv, K, V, I, NT);
    hipLaunchKernelGGL(gst_attn_fwd_kernel, dim3(B), dim3(C), (size_t)(G + 2 * C + H * NT) * sizeof(float), s, hs, lengths, Wq, bq, K, V, Wp, bp, out, keep, d);
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

extern "C" int glowtts_gst_bwd(const float* dout, const float* keep, const int64_t* lengths, int stride_prod, const float* tokens, const float* Wq, const float* Wk,
                               const float* Wv, const float* Wp, const float* K, const float* V, float* dhs, float* scratch /* B (2C + H NT) + 2 C NT floats */,
                               float* dWq, float* dbq, float* dWk, float* dbk, float* dWv, float* dbv, float* dWp, float* dbp, float* dtokens,
                               int B, int Tp, int G, int C, int H, int NT, int I, void* stream)
{
    if (!dout || !keep || !lengths || !tokens || !Wq || !Wk || !Wv || !Wp || !K || !V || !dhs || !scratch || !dWq || !dWk || !dWv || !dWp || !dtokens || stride_prod < 1 ||
        !gst_ok(B, Tp, G, C, H, NT, I)) return GLOWTTS_E_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const gst_dims d = {B, Tp, G, C, H, NT, I, stride_prod};
    float* vec = scratch;
    float* dK = scratch + (long)B * (2 * C + H * NT);
    float* dV = dK + (long)C * NT;
    GLOWTTS_NOTE_STATIC("gst_bwd");
    hipLaunchKernelGGL(gst_attn_bwd_kernel, dim3(B), dim3(C), (size_t)(3 * C + 2 * H * NT + H) * sizeof(float), s, dout, keep, lengths, Wq, K, V, Wp, dhs, vec, d);
    long m1 = (long)C * C;
    if ((long)C * G > m1) m1 = (long)C * G;
    if ((long)C * NT > m1) m1 = (long)C * NT;
    hipLaunchKernelGGL(gst_grads1_kernel, dim3((unsigned)((m1 + 255) / 256), 5), dim3(256), 0, s, dout, keep, vec, dWp, dbp, dWq, dbq, dK, dV, d);
    long m2 = (long)C * I;
    if ((long)I * NT > m2) m2 = (long)I * NT;
    const float* TT = keep + (long)B * (2 * C + H * NT + G);
    hipLaunchKernelGGL(gst_grads2_kernel, dim3((unsigned)((m2 + 255) / 256), 3), dim3(256), 0, s, TT, Wk, Wv, dK, dV, dWk, dbk, dWv, dbv, dtokens, d);
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

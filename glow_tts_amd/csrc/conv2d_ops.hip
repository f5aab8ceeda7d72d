// Direct Conv2d(3x3, stride 2, padding 1, no bias) + ReLU on channels-last activations [B][H][W][C] for gfx950: the conv stack of the GST
// reference encoder (Modules.py:320-333, 366-368: six layers, 1 -> 32 -> 32 -> 64 -> 64 -> 128 -> 128 channels over an 80 x T mel image) -
// forward, data gradient and weight gradient without patch matrices and without layout changes (round 6; replaces the im2col + GEMM path of
// round 2 and MIOpen's direct kernels, which cost BASELINE config 5 ~100 launches in front of everything the prosody vector conditions).
//
//   forward   y[b][ho][wo][co] = relu( sum_{kh,kw,ci} x[b][2ho+kh-1][2wo+kw-1][ci] w[co][ci][kh][kw] )
//             An implicit GEMM whose A row (one output pixel) is GATHERED while it is staged: in channels-last memory the three kw taps of a
//             pixel are 3 Ci contiguous floats, so the K axis (kh, kw, ci) is three contiguous runs per row - no patch matrix exists anywhere.
//   dgrad     dx[b][h][w][ci] = sum dpre[b][ho][wo][co] w[co][ci][kh][kw] over the (kh, ho), (kw, wo) with 2ho + kh - 1 = h, 2wo + kw - 1 = w.
//             An input pixel of parity class (h & 1, w & 1) is read by 1, 2, 2 or 4 taps; each class is the same gather GEMM over dpre with its
//             own weight image (K = taps * Co) and a strided destination - the four classes are one launch.  The ReLU of the layer that PRODUCED
//             x is applied in the epilogue (dx *= x > 0), so what leaves the kernel is that layer's pre-activation gradient.
//   wgrad     dw[co][ci][kh][kw] = sum over pixels of dpre[px][co] * patch[px][(kh, kw, ci)]: the reduction runs over the slow axis of both
//             operands, which is exactly the operand layout of v_mfma_f32_32x32x2_f32 (a lane supplies ONE element, row-major tiles are read
//             conflict-free with ds_read_b32) - the f32 mode; bf16 mode: the same tiles, eight rows gathered and rounded per fragment for
//             v_mfma_f32_32x32x16_bf16.  Rows split over workgroups, partial images summed in a fixed order by one launch for all layers
//             (deterministic, no atomics).
//   layer 0   (Ci = 1, K = 9) is bandwidth work - 8 MB of mel in, 65 MB of activations out at B = 32 - and runs on the VALU.
// bf16 mode: forward / dgrad operands are rounded to bf16 while staged (v_mfma_f32_32x32x16_bf16, fp32 accumulate), f32 mode: exact fp32 MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/glowtts_hip.h"
#include "launch_log.h"
#include "device_common.h"

namespace {

template <typename CT> struct P2;
template <> struct P2<float>  { static constexpr int KC = 16; static constexpr int E = 4; };
template <> struct P2<__bf16> { static constexpr int KC = 32; static constexpr int E = 8; };

inline int pad32(int v) { return (v + 31) / 32 * 32; }

// ------------------------------------------------------------------------------------------------------------------------------------
// weight images: [kchunk][npad][64 B] like glowtts_pack_weight (taps = 1).  cls < 0: forward, N = Co, k = (kh, kw, ci);
// cls = 2 ph + pw: data gradient of parity class (ph, pw), N = Ci, k = (seg, j, co) with kh = ph ? (seg ? 0 : 2) : 1, kw = pw ? (j ? 0 : 2) : 1
// ------------------------------------------------------------------------------------------------------------------------------------
template <typename CT>
__global__ __launch_bounds__(256) void c2d_pack_kernel(const glowtts_c2d_pack_job* __restrict__ jobs, int njobs)
{
    constexpr int KC = P2<CT>::KC;
    int j = 0;
    for (int i = 1; i < njobs; ++i) if ((int)blockIdx.x >= jobs[i].block0) j = i;
    const glowtts_c2d_pack_job job = jobs[j];
    const long total = (long)job.kchunks * job.npad * KC;
    const long e = (long)(blockIdx.x - job.block0) * 256 + threadIdx.x;
    if (e >= total) return;
    const int kk = (int)(e % KC);
    const int n = (int)((e / KC) % job.npad);
    const int kc = (int)(e / ((long)KC * job.npad));
    const int k = kc * KC + kk;
    float v = 0.f;
    if (n < job.N && k < job.K) {
        int co, ci, kh, kw;
        if (job.cls < 0) {
            co = n; kh = k / (3 * job.Ci); kw = (k / job.Ci) % 3; ci = k % job.Ci;
        } else {
            const int ph = job.cls >> 1, pw = job.cls & 1, npix = 1 + pw;
            const int seg = k / (npix * job.Co), jj = (k / job.Co) % npix;
            co = k % job.Co; ci = n;
            kh = ph ? (seg ? 0 : 2) : 1;
            kw = pw ? (jj ? 0 : 2) : 1;
        }
        v = job.w[(((long)co * job.Ci + ci) * 3 + kh) * 3 + kw];
    }
    reinterpret_cast<CT*>(job.img)[e] = (CT)v;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// the gather GEMM (forward: one class; data gradient: up to four)
// ------------------------------------------------------------------------------------------------------------------------------------
struct c2d_class {
    const void* w; int npad, kchunks, K;      // packed weight image of this class
    int Hr, Wr;                                // row space: row = (b, hr, wr)
    int sh, oh0, nseg, sw, ow0, npix;          // source lines hr sh + oh0 + s (s < nseg), source pixels wr sw + ow0 + j (j < npix)
    int osh, ooh, osw, oow;                    // destination pixel (hr osh + ooh, wr osw + oow)
    int tile0;                                 // first workgroup of the class
};
struct c2d_args {
    const float* src; int Hs, Ws, Cs;          // [B][Hs][Ws][Cs]
    float* dst; int Hd, Wd, N;                 // [B][Hd][Wd][N]
    const float* gate;                         // optional, shaped like dst: dst = gate > 0 ? value : 0
    int B, relu, ncls;
    c2d_class cls[4];
};

#define C2D_PICK(ci, f) ((ci) == 0 ? p.cls[0].f : (ci) == 1 ? p.cls[1].f : (ci) == 2 ? p.cls[2].f : p.cls[3].f)

// NI x 32 columns per workgroup (blockIdx.y walks the column tiles: small problems are cut along N as well, so that more than a dozen CUs work on them),
// WM x 32 rows, NSUB K chunks per super-step (one barrier pair; deep K with few rows: 4, so that the serial chain of steps is short)
template <typename CT, int NI, int WM, int NSUB>
__global__ __launch_bounds__(WM * 64) void c2d_gemm_kernel(const c2d_args p)
{
    constexpr int KC = P2<CT>::KC, E = P2<CT>::E;
    constexpr int BM = WM * 32, BN = NI * 32, NT = WM * 64;
    constexpr int A_IT = 2;                                    // BM * 4 slots / NT threads
    constexpr int W_IT = (BN * 4 + NT - 1) / NT;
    constexpr int NLD = E / 4;                                 // 16-byte global loads per LDS slot
    __shared__ __attribute__((aligned(16))) unsigned char As[NSUB][BM * 64];
    __shared__ __attribute__((aligned(16))) unsigned char Wsm[NSUB][BN * 64];
    __shared__ int rowoff[BM];

    int ci = 0;
    if (p.ncls > 1 && (int)blockIdx.x >= p.cls[1].tile0) ci = 1;
    if (p.ncls > 2 && (int)blockIdx.x >= p.cls[2].tile0) ci = 2;
    if (p.ncls > 3 && (int)blockIdx.x >= p.cls[3].tile0) ci = 3;
    const int n0 = blockIdx.y * BN, npad = C2D_PICK(ci, npad);
    const unsigned char* wimg = reinterpret_cast<const unsigned char*>(C2D_PICK(ci, w)) + (long)n0 * 64;
    const int KCH = C2D_PICK(ci, kchunks), K = C2D_PICK(ci, K);
    const int Hr = C2D_PICK(ci, Hr), Wr = C2D_PICK(ci, Wr);
    const int sh = C2D_PICK(ci, sh), oh0 = C2D_PICK(ci, oh0), nseg = C2D_PICK(ci, nseg);
    const int sw = C2D_PICK(ci, sw), ow0 = C2D_PICK(ci, ow0), npix = C2D_PICK(ci, npix);
    const int tile0 = C2D_PICK(ci, tile0);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const long rows = (long)p.B * Hr * Wr;
    const long m0 = (long)((int)blockIdx.x - tile0) * BM;
    const int HW = Hr * Wr;

    if (tid < BM) {
        const long r = m0 + tid;
        int off = -1;
        if (r < rows) {
            const int b = (int)(r / HW), rem = (int)(r - (long)b * HW);
            const int hr = rem / Wr, wr = rem - hr * Wr;
            off = ((b * p.Hd + hr * C2D_PICK(ci, osh) + C2D_PICK(ci, ooh)) * p.Wd + wr * C2D_PICK(ci, osw) + C2D_PICK(ci, oow)) * p.N;
        }
        rowoff[tid] = off;
    }

    // ---- per-thread constants of the A gather: two rows, one 16-byte slot column each ----
    const float* aptr[A_IT];
    unsigned amask[A_IT];                                      // bits 0..2: source line s in range, bits 4..6: source pixel j in range
    const long linestride = (long)p.Ws * p.Cs;
    const int SL = npix * p.Cs;                                // floats per segment
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int row = (tid + it * NT) >> 2;
        const long r = m0 + row;
        aptr[it] = p.src;
        amask[it] = 0;
        if (r < rows) {
            const int b = (int)(r / HW), rem = (int)(r - (long)b * HW);
            const int hr = rem / Wr, wr = rem - hr * Wr;
            const int hs0 = hr * sh + oh0, ws0 = wr * sw + ow0;
            aptr[it] = p.src + (((long)b * p.Hs + hs0) * p.Ws + ws0) * p.Cs;     // (dereferenced only where the masks allow)
            unsigned m = 0;
            for (int s = 0; s < nseg; ++s) if (hs0 + s >= 0 && hs0 + s < p.Hs) m |= 1u << s;
            for (int j = 0; j < npix; ++j) if (ws0 + j >= 0 && ws0 + j < p.Ws) m |= 16u << j;
            amask[it] = m;
        }
    }
    const int q = tid & 3;

    f32x16 acc[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;

    Chunk16 ra[NSUB][A_IT][NLD];
    Chunk16 rw[NSUB][W_IT];
    unsigned aok[NSUB];

    auto gload = [&](int ss) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NSUB; ++j) {
            const int kc = min(ss * NSUB + j, KCH - 1);
            const unsigned char* wb = wimg + (long)kc * npad * 64;
#pragma unroll
            for (int it = 0; it < W_IT; ++it) {
                const int idx = min(tid + it * NT, BN * 4 - 1);
                rw[j][it] = *reinterpret_cast<const Chunk16*>(wb + idx * 16);
            }
            const int c = kc * KC + q * E;
            const int seg = c / SL, off = c - seg * SL, jj = off / p.Cs;
            unsigned okbits = 0;
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                const bool ok = (c < K) && ((amask[it] >> seg) & 1u) && ((amask[it] >> (4 + jj)) & 1u);
                const float* s = ok ? aptr[it] + seg * linestride + off : p.src;
#pragma unroll
                for (int l = 0; l < NLD; ++l) ra[j][it][l] = *reinterpret_cast<const Chunk16*>(s + 4 * l);
                okbits |= ok ? (1u << it) : 0u;
            }
            aok[j] = okbits;
        }
    };
    auto sstore = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NSUB; ++j) {
#pragma unroll
            for (int it = 0; it < W_IT; ++it) {
                const int idx = tid + it * NT;
                if (idx < BN * 4) *reinterpret_cast<Chunk16*>(Wsm[j] + swz(idx >> 2, idx & 3)) = rw[j][it];
            }
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                const int row = (tid + it * NT) >> 2;
                const bool ok = (aok[j] >> it) & 1u;
                Chunk16 o;
                if constexpr (sizeof(CT) == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float f0 = __uint_as_float(ra[j][it][e >> 1][(2 * e) & 3]), f1 = __uint_as_float(ra[j][it][e >> 1][(2 * e + 1) & 3]);
                        o[e] = ok ? pack_bf16x2(f0, f1) : 0u;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = ok ? ra[j][it][0][e] : 0u;
                }
                *reinterpret_cast<Chunk16*>(As[j] + swz(row, q)) = o;
            }
        }
    };
    auto compute = [&](int j) __attribute__((always_inline)) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int qq = 2 * s2 + lhi;
            const Chunk16 af = *reinterpret_cast<const Chunk16*>(As[j] + swz(wave * 32 + l31, qq));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const Chunk16 bf = *reinterpret_cast<const Chunk16*>(Wsm[j] + swz(ni * 32 + l31, qq));
                if constexpr (sizeof(CT) == 2) {
                    acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&af), *reinterpret_cast<const bf16x8*>(&bf), acc[ni], 0, 0, 0);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(af[e]), __uint_as_float(bf[e]), acc[ni], 0, 0, 0);
                }
            }
        }
    };

    const int NSS = (KCH + NSUB - 1) / NSUB;
    gload(0);
    sstore();
    __syncthreads();
    for (int ss = 0; ss < NSS; ++ss) {
        const int nxt = ss + 1 < NSS ? ss + 1 : ss;
        gload(nxt);                                           // in flight during the MFMAs below
#pragma unroll
        for (int j = 0; j < NSUB; ++j)
            if (ss * NSUB + j < KCH) compute(j);
        __syncthreads();
        sstore();
        __syncthreads();
    }

    // ---- epilogue: relu / gate, scattered rows, 128-byte runs along the channels ----
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = wave * 32 + (r >> 2) * 8 + lhi * 4 + (r & 3);
        const int off = rowoff[row];
        if (off < 0) continue;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n0 + ni * 32 + l31;
            if (n >= p.N) continue;
            float v = acc[ni][r];
            if (p.relu) v = fmaxf(v, 0.f);
            if (p.gate) v = p.gate[(long)off + n] > 0.f ? v : 0.f;
            p.dst[(long)off + n] = v;
        }
    }
}

template <typename CT, int NI, int WM, int NSUB>
int c2d_launch(const c2d_args& a, int tiles, int ny, hipStream_t s)
{
    hipLaunchKernelGGL((c2d_gemm_kernel<CT, NI, WM, NSUB>), dim3(tiles, ny), dim3(WM * 64), 0, s, a);
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

template <typename CT>
int c2d_dispatch_t(c2d_args& a, hipStream_t s)
{
    // rows per workgroup: 128 when that still gives every CU two workgroups, else 64; few row tiles: one 32-column tile per workgroup
    long t128 = 0;
    int kmax = 0;
    for (int c = 0; c < a.ncls; ++c) {
        const long rows = (long)a.B * a.cls[c].Hr * a.cls[c].Wr;
        t128 += (rows + 127) / 128;
        if (a.cls[c].kchunks > kmax) kmax = a.cls[c].kchunks;
    }
    const bool wide = t128 >= 512;
    int t0 = 0;
    for (int c = 0; c < a.ncls; ++c) {
        const long rows = (long)a.B * a.cls[c].Hr * a.cls[c].Wr;
        a.cls[c].tile0 = t0;
        t0 += (int)(wide ? (rows + 127) / 128 : (rows + 63) / 64);
    }
    const int NIall = pad32(a.N) / 32;
    if (NIall != 1 && NIall != 2 && NIall != 4) return GLOWTTS_E_ARG;
    if (wide) {
        switch (NIall) {
            case 1: return c2d_launch<CT, 1, 4, 2>(a, t0, 1, s);
            case 2: return c2d_launch<CT, 2, 4, 2>(a, t0, 1, s);
            default: return c2d_launch<CT, 4, 4, 2>(a, t0, 1, s);
        }
    }
    const bool cutn = NIall > 1 && (long)t0 * NIall <= 1024;           // column tiles of 32: the chip still is not over-subscribed
    const bool deep = kmax >= 12;
    if (cutn) return deep ? c2d_launch<CT, 1, 2, 4>(a, t0, NIall, s) : c2d_launch<CT, 1, 2, 2>(a, t0, NIall, s);
    switch (NIall) {
        case 1: return deep ? c2d_launch<CT, 1, 2, 4>(a, t0, 1, s) : c2d_launch<CT, 1, 2, 2>(a, t0, 1, s);
        case 2: return deep ? c2d_launch<CT, 2, 2, 4>(a, t0, 1, s) : c2d_launch<CT, 2, 2, 2>(a, t0, 1, s);
        default: return deep ? c2d_launch<CT, 4, 2, 4>(a, t0, 1, s) : c2d_launch<CT, 4, 2, 2>(a, t0, 1, s);
    }
}

int c2d_dispatch(c2d_args& a, int precision, hipStream_t s)
{
    return precision == GLOWTTS_BF16 ? c2d_dispatch_t<__bf16>(a, s) : c2d_dispatch_t<float>(a, s);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// layer 0 (Ci = 1): forward and weight gradient on the VALU.  A thread owns (pixel, 4 output channels): a wave's store is 1 KiB contiguous.
// ------------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void first_patch(const float* __restrict__ x, int b, int ho, int wo, int H, int W, float (&xv)[9])
{
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int h = 2 * ho + kh - 1;
        const bool hok = h >= 0 && h < H;
        const float* line = x + ((long)b * H + (hok ? h : 0)) * W;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int w = 2 * wo + kw - 1;
            const bool ok = hok && w >= 0 && w < W;
            const float v = line[ok ? w : 0];
            xv[kh * 3 + kw] = ok ? v : 0.f;
        }
    }
}

__global__ __launch_bounds__(256) void c2d_first_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                            int B, int H, int W, int Ho, int Wo, int Co, int relu)
{
    __shared__ __attribute__((aligned(16))) float wl[9 * 128];             // [k][co]
    for (int i = threadIdx.x; i < 9 * Co; i += 256) wl[(i % 9) * Co + i / 9] = w[i];
    __syncthreads();
    const int G = Co >> 2;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long px = t / G;
    const int g = (int)(t - px * G);
    if (px >= (long)B * Ho * Wo) return;
    const int b = (int)(px / (Ho * Wo)), rem = (int)(px - (long)b * Ho * Wo);
    const int ho = rem / Wo, wo = rem - ho * Wo;
    float xv[9];
    first_patch(x, b, ho, wo, H, W, xv);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(&wl[k * Co + 4 * g]);
        acc += xv[k] * wv;
    }
    if (relu) { acc[0] = fmaxf(acc[0], 0.f); acc[1] = fmaxf(acc[1], 0.f); acc[2] = fmaxf(acc[2], 0.f); acc[3] = fmaxf(acc[3], 0.f); }
    *reinterpret_cast<f32x4*>(y + px * Co + 4 * g) = acc;
}

// partial[block][co][k] = sum over the block's pixels of dpre[px][co] * patch[px][k]; 512 threads = (512 / G) pixel slots x G channel groups
__global__ __launch_bounds__(512) void c2d_first_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dpre, float* __restrict__ partial,
                                                              int B, int H, int W, int Ho, int Wo, int Co)
{
    extern __shared__ float red[];                                         // [512][36 + 1]
    const int G = Co >> 2, slots = 512 / G;
    const int g = threadIdx.x % G, slot = threadIdx.x / G;
    const long npx = (long)B * Ho * Wo;
    float acc[9][4];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[k][i] = 0.f;
    if (slot < slots) {
        // four pixels per round: their 4 + 36 loads are issued together (a round is one memory round trip, whatever it carries)
        const long stride = (long)gridDim.x * slots;
        for (long px0 = (long)blockIdx.x * slots + slot; px0 < npx; px0 += 4 * stride) {
            float xv[4][9];
            f32x4 d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long px = px0 + u * stride;
                const bool ok = px < npx;
                const long pc = ok ? px : px0;
                const int b = (int)(pc / (Ho * Wo)), rem = (int)(pc - (long)b * Ho * Wo);
                const int ho = rem / Wo, wo = rem - ho * Wo;
                first_patch(x, b, ho, wo, H, W, xv[u]);
                const f32x4 dv = *reinterpret_cast<const f32x4*>(dpre + pc * Co + 4 * g);
                d[u] = ok ? dv : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 9; ++k)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[k][i] += xv[u][k] * d[u][i];
        }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[threadIdx.x * 37 + k * 4 + i] = acc[k][i];
    __syncthreads();
    for (int o = threadIdx.x; o < Co * 9; o += 512) {                      // o = co * 9 + k, summed over the slots in a fixed order
        const int co = o / 9, k = o - co * 9;
        float s = 0.f;
        for (int sl = 0; sl < slots; ++sl) s += red[(sl * G + (co >> 2)) * 37 + k * 4 + (co & 3)];
        partial[(long)blockIdx.x * Co * 9 + o] = s;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// weight gradient, Ci % 4 == 0, Co = 32 MF: partial[split][co][col], col = (kh, kw, ci)
// ------------------------------------------------------------------------------------------------------------------------------------
struct c2d_wgrad_args {
    const float* x; const float* dpre; float* partial;
    int B, H, W, Ci, Ho, Wo, Co, K9;
    long rows; int rows_per_split;
};

// BF: bf16 arithmetic mode - the fragments are gathered from the same row-major fp32 tiles (eight ds_read_b32 at the row pitch), rounded to bf16 in registers and
// multiplied by v_mfma_f32_32x32x16_bf16 (1/16 of the exact-fp32 MFMA's time for the same 16 rows; the staging then sets the pace).  !BF: exact fp32.
template <int MF, bool BF>
__global__ __launch_bounds__(256) void c2d_wgrad_kernel(const c2d_wgrad_args p)
{
    constexpr int RS = MF <= 2 ? 64 : 32, CO = MF * 32;                 // rows per step (a step is one memory round trip: the fatter the better; LDS: 2 x RS x (AST + BST) x 4 B)
    constexpr int AST = CO + ((MF & 1) ? 0 : 32), BST = 128 + 32;         // row strides = 32 mod 64 floats: the two half-waves of a read hit disjoint banks
    constexpr int A_IT = RS * CO / 4 / 256 > 0 ? RS * CO / 4 / 256 : 1;    // float4 items per thread (Co = 32: one)
    constexpr int B_IT = RS * 128 / 4 / 256;                               // 4 or 8
    extern __shared__ __attribute__((aligned(16))) float wg_smem[];         // At [2][RS * AST] | Bt [2][RS * BST] (80 KiB at Co = 128: dynamic)
    float (*At)[RS * AST] = reinterpret_cast<float (*)[RS * AST]>(wg_smem);
    float (*Bt)[RS * BST] = reinterpret_cast<float (*)[RS * BST]>(wg_smem + 2 * RS * AST);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int n0 = blockIdx.x * 128;
    const long r_begin = (long)blockIdx.y * p.rows_per_split;
    const long r_end = min(r_begin + p.rows_per_split, p.rows);
    const int HWo = p.Ho * p.Wo;

    // B staging: this thread's column group (fixed) and its rows rr = tid / 32 + 8 it
    const int c4 = tid & 31;
    const int col = n0 + 4 * c4;
    const bool colok = col < p.K9;
    const int kh = colok ? col / (3 * p.Ci) : 0, kw = colok ? (col / p.Ci) % 3 : 0, cc = colok ? col % p.Ci : 0;

    // Row table of a step: for each of its RS rows the element offset of input pixel (2 ho - 1, 2 wo - 1) of that row's utterance and 3 + 3 validity bits of the
    // lines 2 ho - 1 + kh / the pixels 2 wo - 1 + kw.  RS threads fill it two steps ahead (one row -> (b, ho, wo) decomposition each); the gather of a
    // thread's 4 - 8 rows is then an LDS read, an add of its column's constant and a shift instead of two integer divisions per row (they were the loop).
    __shared__ int rinfo[2][RS][2];
    const long coloff = colok ? ((long)kh * p.W + kw) * p.Ci + cc : 0;
    const unsigned colbit = (1u << kh) | (8u << kw);
    auto fill_rinfo = [&](long r0, int slot) __attribute__((always_inline)) {
        if (tid < RS) {
            const long r = r0 + tid;
            int base = 0;
            unsigned m = 0;
            if (r < r_end) {
                const int b = (int)(r / HWo), rem = (int)(r - (long)b * HWo);
                const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
                const int h0 = 2 * ho - 1, w0 = 2 * wo - 1;
                base = (int)((((long)b * p.H + h0) * p.W + w0) * p.Ci);          // (may be negative; only added to in-range columns.  |base| < 2^31: host guard)
#pragma unroll
                for (int k = 0; k < 3; ++k) { if (h0 + k >= 0 && h0 + k < p.H) m |= 1u << k; if (w0 + k >= 0 && w0 + k < p.W) m |= 8u << k; }
            }
            rinfo[slot][tid][0] = base;
            rinfo[slot][tid][1] = (int)m;
        }
    };
    f32x4 ra[A_IT], rb[B_IT];
    auto gload = [&](long r0, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int idx = tid + it * 256;
            const int rr = idx / (CO / 4), c = idx % (CO / 4);
            const long r = r0 + rr;
            const bool ok = (rr < RS) && (r < r_end);
            const f32x4 v = *reinterpret_cast<const f32x4*>(p.dpre + (ok ? r : 0) * CO + 4 * c);
            ra[it] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int rr = (tid >> 5) + 8 * it;
            const int base = rinfo[slot][rr][0];
            const unsigned m = (unsigned)rinfo[slot][rr][1];
            const bool ok = colok && ((m & colbit) == colbit);
            const f32x4 v = *reinterpret_cast<const f32x4*>(p.x + (ok ? (long)base + coloff : 0));
            rb[it] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto sstore = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int idx = tid + it * 256;
            const int rr = idx / (CO / 4), c = idx % (CO / 4);
            if (rr < RS) *reinterpret_cast<f32x4*>(&At[buf][rr * AST + 4 * c]) = ra[it];
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int rr = (tid >> 5) + 8 * it;
            *reinterpret_cast<f32x4*>(&Bt[buf][rr * BST + 4 * c4]) = rb[it];
        }
    };

    f32x16 acc[MF];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mf][r] = 0.f;

    const int nsteps = (int)((r_end - r_begin + RS - 1) / RS);
    fill_rinfo(r_begin, 0);
    __syncthreads();
    if (nsteps > 0) gload(r_begin, 0);
    fill_rinfo(r_begin + RS, 1);
    if (nsteps > 0) sstore(0);
    __syncthreads();
    for (int st = 0; st < nsteps; ++st) {
        const int buf = st & 1;
        if (st + 1 < nsteps) gload(r_begin + (long)(st + 1) * RS, (st + 1) & 1);
        if constexpr (BF) {
#pragma unroll
            for (int k16 = 0; k16 < RS / 16; ++k16) {
                const int r0 = k16 * 16 + lhi * 8;               // this lane's eight consecutive rows (the MFMA's k group)
                Chunk16 bb;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    bb[e] = pack_bf16x2(Bt[buf][(r0 + 2 * e) * BST + wave * 32 + l31], Bt[buf][(r0 + 2 * e + 1) * BST + wave * 32 + l31]);
#pragma unroll
                for (int mf = 0; mf < MF; ++mf) {
                    Chunk16 aa;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        aa[e] = pack_bf16x2(At[buf][(r0 + 2 * e) * AST + mf * 32 + l31], At[buf][(r0 + 2 * e + 1) * AST + mf * 32 + l31]);
                    acc[mf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&aa), *reinterpret_cast<const bf16x8*>(&bb), acc[mf], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int k2 = 0; k2 < RS / 2; ++k2) {
                const int rk = 2 * k2 + lhi;
                const float bv = Bt[buf][rk * BST + wave * 32 + l31];
#pragma unroll
                for (int mf = 0; mf < MF; ++mf) {
                    const float av = At[buf][rk * AST + mf * 32 + l31];
                    acc[mf] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[mf], 0, 0, 0);
                }
            }
        }
        // (the table slot of step st was last read by gload(st), one barrier ago: it takes step st + 2's rows now)
        if (st + 2 < nsteps) fill_rinfo(r_begin + (long)(st + 2) * RS, buf);
        if (st + 1 < nsteps) sstore(buf ^ 1);
        __syncthreads();
    }
    const int n = n0 + wave * 32 + l31;
    if (n < p.K9) {
        float* out = p.partial + (long)blockIdx.y * CO * p.K9;
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = mf * 32 + (r >> 2) * 8 + lhi * 4 + (r & 3);
                out[(long)co * p.K9 + n] = acc[mf][r];
            }
    }
}

// dw[co][ci][kh][kw] = sum over the splits (fixed order) of partial[s][co][(kh, kw, ci)], every layer of the stack in one launch
struct c2d_reduce_table { glowtts_c2d_reduce_job j[GLOWTTS_C2D_MAX_LAYERS]; int n; };
#define C2D_RPICK(i, f) ((i) == 0 ? t.j[0].f : (i) == 1 ? t.j[1].f : (i) == 2 ? t.j[2].f : (i) == 3 ? t.j[3].f : (i) == 4 ? t.j[4].f : (i) == 5 ? t.j[5].f : \
                         (i) == 6 ? t.j[6].f : t.j[7].f)

__global__ __launch_bounds__(256) void c2d_reduce_kernel(const c2d_reduce_table t)
{
    // 64 elements per workgroup x 4 split lanes: lane g of an element sums the splits g, g + 4, ... four loads at a time (a serial walk over 500 partial
    // images was 180 us of load latency), the four lane sums are added in a fixed order: deterministic
    __shared__ float lanes[4][64];
    int i = 0;
#pragma unroll
    for (int k = 1; k < GLOWTTS_C2D_MAX_LAYERS; ++k) if (k < t.n && (int)blockIdx.x >= t.j[k].block0) i = k;
    const float* partial = C2D_RPICK(i, partial);
    float* dw = C2D_RPICK(i, dw);
    const int splits = C2D_RPICK(i, splits), Ci = C2D_RPICK(i, Ci), Co = C2D_RPICK(i, Co), block0 = C2D_RPICK(i, block0);
    const int K9 = 9 * Ci;
    const long total = (long)Co * K9;
    const int el = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long e = (long)((int)blockIdx.x - block0) * 64 + el;                    // e = co * K9 + col
    float s = 0.f;
    if (e < total) {
        int sp = g;
        for (; sp + 12 < splits; sp += 16) {
            const float a0 = partial[(long)sp * total + e], a1 = partial[(long)(sp + 4) * total + e];
            const float a2 = partial[(long)(sp + 8) * total + e], a3 = partial[(long)(sp + 12) * total + e];
            s += a0; s += a1; s += a2; s += a3;
        }
        for (; sp < splits; sp += 4) s += partial[(long)sp * total + e];
    }
    lanes[g][el] = s;
    __syncthreads();
    if (g == 0 && e < total) {
        s = ((lanes[0][el] + lanes[1][el]) + lanes[2][el]) + lanes[3][el];
        const int co = (int)(e / K9), col = (int)(e - (long)co * K9);
        const int kh = col / (3 * Ci), kw = (col / Ci) % 3, ci = col % Ci;
        dw[(((long)co * Ci + ci) * 3 + kh) * 3 + kw] = s;
    }
}

bool c2d_shape_ok(int B, int H, int W, int Ci, int Co)
{
    if (B < 1 || H < 1 || W < 1 || Ci < 1 || Co < 4) return false;
    if ((long)B * H * W * (Ci > Co ? Ci : Co) >= (1L << 31)) return false;          // element offsets are 32-bit inside the kernels
    if (Ci == 1) return (Co % 4) == 0 && Co <= 128 && 512 % (Co / 4) == 0;
    return (Ci % 8) == 0 && Ci <= 128 && (Co == 32 || Co == 64 || Co == 128) && (Ci == 32 || Ci == 64 || Ci == 128);
}

void wgrad_plan(int B, int H, int W, int Ci, int Co, int* nsplit, int* rows_per_split)
{
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long rows = (long)B * Ho * Wo;
    if (Ci == 1) { *nsplit = (int)((rows + 255) / 256 < 512 ? (rows + 255) / 256 : 512); if (*nsplit < 1) *nsplit = 1; *rows_per_split = 0; return; }
    const int K9 = 9 * Ci, ntn = (K9 + 127) / 128;
    const int RS = Co <= 64 ? 64 : 32;
    long cap_bytes = (long)(12 * 1024 * 1024) / ((long)Co * K9 * 4);       // partial images: at most ~12 MB per layer
    long ns = 1024 / ntn;                                                  // ~4 workgroups per CU
    if (cap_bytes < ns) ns = cap_bytes;
    if ((rows + RS - 1) / RS < ns) ns = (rows + RS - 1) / RS;
    if (ns < 1) ns = 1;
    long rps = ((rows + ns - 1) / ns + RS - 1) / RS * RS;
    *rows_per_split = (int)rps;
    *nsplit = (int)((rows + rps - 1) / rps);
}

}  // namespace

extern "C" int glowtts_conv3x3s2_supported(int B, int H, int W, int Ci, int Co) { return c2d_shape_ok(B, H, W, Ci, Co) ? 1 : 0; }

extern "C" int glowtts_conv3x3s2_image_bytes(int Ci, int Co, int precision, int64_t* fwd_bytes, int64_t* dgrad_bytes /* [4] */)
{
    if (Ci < 1 || Co < 1 || (precision != GLOWTTS_F32 && precision != GLOWTTS_BF16)) return GLOWTTS_E_ARG;
    const int KC = precision == GLOWTTS_BF16 ? 32 : 16;
    if (fwd_bytes) *fwd_bytes = (int64_t)((9 * Ci + KC - 1) / KC) * pad32(Co) * 64;
    if (dgrad_bytes)
        for (int cls = 0; cls < 4; ++cls) {
            const int K = (1 + (cls >> 1)) * (1 + (cls & 1)) * Co;
            dgrad_bytes[cls] = (int64_t)((K + KC - 1) / KC) * pad32(Ci) * 64;
        }
    return GLOWTTS_OK;
}

extern "C" int glowtts_conv3x3s2_pack_job_init(glowtts_c2d_pack_job* job, const float* w, int Ci, int Co, int cls, int precision, void* img,
                                               int block0, int* blocks_out)
{
    if (!job || !w || !img || Ci < 1 || Co < 1 || cls < -1 || cls > 3) return GLOWTTS_E_ARG;
    const int KC = precision == GLOWTTS_BF16 ? 32 : 16;
    job->w = w; job->img = img; job->Ci = Ci; job->Co = Co; job->cls = cls;
    if (cls < 0) { job->N = Co; job->K = 9 * Ci; }
    else         { job->N = Ci; job->K = (1 + (cls >> 1)) * (1 + (cls & 1)) * Co; }
    job->npad = pad32(job->N);
    job->kchunks = (job->K + KC - 1) / KC;
    job->block0 = block0;
    const long total = (long)job->kchunks * job->npad * KC;
    if (blocks_out) *blocks_out = (int)((total + 255) / 256);
    return GLOWTTS_OK;
}

extern "C" int glowtts_conv3x3s2_pack(const glowtts_c2d_pack_job* dev_jobs, int njobs, int total_blocks, int precision, void* stream)
{
    if (!dev_jobs || njobs < 1 || total_blocks < 1) return GLOWTTS_E_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    GLOWTTS_NOTE_STATIC("conv3x3s2_pack");
    if (precision == GLOWTTS_BF16) hipLaunchKernelGGL(c2d_pack_kernel<__bf16>, dim3(total_blocks), dim3(256), 0, s, dev_jobs, njobs);
    else                           hipLaunchKernelGGL(c2d_pack_kernel<float>, dim3(total_blocks), dim3(256), 0, s, dev_jobs, njobs);
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

extern "C" int glowtts_conv3x3s2_fwd(const float* x, const float* w, const void* img_fwd, float* y, int B, int H, int W, int Ci, int Co, int relu,
                                     int precision, void* stream)
{
    if (!x || !y || !c2d_shape_ok(B, H, W, Ci, Co)) return GLOWTTS_E_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    if (Ci == 1) {
        if (!w) return GLOWTTS_E_ARG;
        GLOWTTS_NOTE_STATIC("conv3x3s2_first_fwd");
        const long threads = (long)B * Ho * Wo * (Co / 4);
        hipLaunchKernelGGL(c2d_first_fwd_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, x, w, y, B, H, W, Ho, Wo, Co, relu);
        return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
    }
    if (!img_fwd) return GLOWTTS_E_ARG;
    const int KC = precision == GLOWTTS_BF16 ? 32 : 16;
    c2d_args a = {};
    a.src = x; a.Hs = H; a.Ws = W; a.Cs = Ci;
    a.dst = y; a.Hd = Ho; a.Wd = Wo; a.N = Co;
    a.gate = nullptr; a.B = B; a.relu = relu; a.ncls = 1;
    c2d_class& c = a.cls[0];
    c.w = img_fwd; c.npad = pad32(Co); c.K = 9 * Ci; c.kchunks = (c.K + KC - 1) / KC;
    c.Hr = Ho; c.Wr = Wo; c.sh = 2; c.oh0 = -1; c.nseg = 3; c.sw = 2; c.ow0 = -1; c.npix = 3;
    c.osh = 1; c.ooh = 0; c.osw = 1; c.oow = 0;
    GLOWTTS_NOTE("conv3x3s2_fwd<%s>", precision == GLOWTTS_BF16 ? "bf16" : "f32");
    return c2d_dispatch(a, precision, s);
}

extern "C" int glowtts_conv3x3s2_dgrad(const float* dpre, const void* const* img_dgrad /* [4] host array of device pointers */, const float* gate,
                                       float* dx, int B, int H, int W, int Ci, int Co, int precision, void* stream)
{
    if (!dpre || !img_dgrad || !dx || Ci == 1 || !c2d_shape_ok(B, H, W, Ci, Co)) return GLOWTTS_E_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const int KC = precision == GLOWTTS_BF16 ? 32 : 16;
    c2d_args a = {};
    a.src = dpre; a.Hs = Ho; a.Ws = Wo; a.Cs = Co;
    a.dst = dx; a.Hd = H; a.Wd = W; a.N = Ci;
    a.gate = gate; a.B = B; a.relu = 0; a.ncls = 0;
    for (int cls = 0; cls < 4; ++cls) {
        const int ph = cls >> 1, pw = cls & 1;
        const int Hr = (H - ph + 1) / 2, Wr = (W - pw + 1) / 2;
        if (Hr < 1 || Wr < 1) continue;                        // (H == 1 or W == 1: no odd lines / columns)
        if (!img_dgrad[cls]) return GLOWTTS_E_ARG;
        c2d_class& c = a.cls[a.ncls++];
        c.w = img_dgrad[cls]; c.npad = pad32(Ci); c.K = (1 + ph) * (1 + pw) * Co; c.kchunks = (c.K + KC - 1) / KC;
        c.Hr = Hr; c.Wr = Wr; c.sh = 1; c.oh0 = 0; c.nseg = 1 + ph; c.sw = 1; c.ow0 = 0; c.npix = 1 + pw;
        c.osh = 2; c.ooh = ph; c.osw = 2; c.oow = pw;
    }
    GLOWTTS_NOTE("conv3x3s2_dgrad<%s>", precision == GLOWTTS_BF16 ? "bf16" : "f32");
    return c2d_dispatch(a, precision, s);
}

extern "C" int64_t glowtts_conv3x3s2_wgrad_scratch_floats(int B, int H, int W, int Ci, int Co)
{
    if (!c2d_shape_ok(B, H, W, Ci, Co)) return -1;
    int ns, rps;
    wgrad_plan(B, H, W, Ci, Co, &ns, &rps);
    return (int64_t)ns * Co * 9 * Ci;
}

extern "C" int glowtts_conv3x3s2_wgrad(const float* x, const float* dpre, float* partial, int B, int H, int W, int Ci, int Co, int precision, int* splits_out,
                                       void* stream)
{
    if (!x || !dpre || !partial || !c2d_shape_ok(B, H, W, Ci, Co)) return GLOWTTS_E_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    int ns, rps;
    wgrad_plan(B, H, W, Ci, Co, &ns, &rps);
    if (splits_out) *splits_out = ns;
    if (Ci == 1) {
        GLOWTTS_NOTE_STATIC("conv3x3s2_first_wgrad");
        static bool done = false;
        if (!done) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&c2d_first_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(512 * 37 * sizeof(float))) != hipSuccess) return GLOWTTS_E_LAUNCH;
            done = true;
        }
        hipLaunchKernelGGL(c2d_first_wgrad_kernel, dim3(ns), dim3(512), 512 * 37 * sizeof(float), s, x, dpre, partial, B, H, W, Ho, Wo, Co);
        return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
    }
    c2d_wgrad_args a;
    a.x = x; a.dpre = dpre; a.partial = partial;
    a.B = B; a.H = H; a.W = W; a.Ci = Ci; a.Ho = Ho; a.Wo = Wo; a.Co = Co; a.K9 = 9 * Ci;
    a.rows = (long)B * Ho * Wo; a.rows_per_split = rps;
    const dim3 grid((a.K9 + 127) / 128, ns);
    GLOWTTS_NOTE("conv3x3s2_wgrad<%s>", precision == GLOWTTS_BF16 ? "bf16" : "f32");
#define C2D_WG(MF, BF) do { constexpr int AST_ = MF * 32 + ((MF & 1) ? 0 : 32); constexpr size_t lds = (size_t)2 * (MF <= 2 ? 64 : 32) * (AST_ + 160) * sizeof(float); \
        static bool done = false; if (!done) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(&c2d_wgrad_kernel<MF, BF>), \
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return GLOWTTS_E_LAUNCH; done = true; } \
        hipLaunchKernelGGL((c2d_wgrad_kernel<MF, BF>), grid, dim3(256), lds, s, a); } while (0)
    const bool bf = precision == GLOWTTS_BF16;
    switch (Co / 32) {
        case 1: if (bf) C2D_WG(1, true); else C2D_WG(1, false); break;
        case 2: if (bf) C2D_WG(2, true); else C2D_WG(2, false); break;
        case 4: if (bf) C2D_WG(4, true); else C2D_WG(4, false); break;
        default: return GLOWTTS_E_ARG;
    }
#undef C2D_WG
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

extern "C" int glowtts_conv3x3s2_wgrad_reduce(const glowtts_c2d_reduce_job* jobs /* host */, int njobs, void* stream)
{
    if (!jobs || njobs < 1 || njobs > GLOWTTS_C2D_MAX_LAYERS) return GLOWTTS_E_ARG;
    c2d_reduce_table t = {};
    int b0 = 0;
    for (int i = 0; i < njobs; ++i) {
        if (!jobs[i].partial || !jobs[i].dw || jobs[i].splits < 1) return GLOWTTS_E_ARG;
        t.j[i] = jobs[i];
        t.j[i].block0 = b0;
        b0 += (int)(((long)jobs[i].Co * 9 * jobs[i].Ci + 63) / 64);
    }
    t.n = njobs;
    GLOWTTS_NOTE_STATIC("conv3x3s2_wgrad_reduce");
    hipLaunchKernelGGL(c2d_reduce_kernel, dim3(b0), dim3(256), 0, static_cast<hipStream_t>(stream), t);
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

// Host-side orchestration of one decoder flow step (Modules.py:653-668 AIA) for gfx950: a fixed sequence of
// launches of the MFMA conv kernel (gemm_cl.hip), the weight-gradient kernel (wgrad_cl.hip) and the
// bandwidth-bound flow kernels (flow_ops.hip), all on the caller's stream.  No device synchronisation, no
// allocation: every buffer comes from the caller (see glowtts_flow_acts / glowtts_flow_grads).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include "../../include/glowtts_hip.h"
#include "tunable.h"

namespace {

#define CHECK(x) do { int rc_ = (x); if (rc_ != GLOWTTS_OK) return rc_; } while (0)

struct Ctx {
    const glowtts_flow_dims* d; const glowtts_flow_params* p; const glowtts_flow_acts* a; void* s;
    int Tp, R, C2, H, pad;
};

Ctx make_ctx(const glowtts_flow_dims* d, const glowtts_flow_params* p, const glowtts_flow_acts* a, void* s) {
    Ctx c{d, p, a, s, 0, 0, 0, 0, 0};
    c.Tp = d->T + 2 * GLOWTTS_ROW_PAD; c.R = d->B * c.Tp; c.C2 = d->C / 2; c.H = d->H; c.pad = (d->ksize - 1) / 2;
    return c;
}

int check_dims(const glowtts_flow_dims* d) {
    if (!d || d->B < 1 || d->T < 1 || d->C < 4 || (d->C & 3) || d->H < 1 || (d->H & 3) || d->L < 1 || d->L > GLOWTTS_MAX_WN_LAYERS) return GLOWTTS_E_ARG;
    if (d->ksize != 1 && d->ksize != 3 && d->ksize != 5) return GLOWTTS_E_ARG;
    if ((d->ksize - 1) / 2 > GLOWTTS_ROW_PAD) return GLOWTTS_E_ARG;
    if (d->act_bf16 && (d->precision != GLOWTTS_BF16 || (d->H & 7))) return GLOWTTS_E_ARG;
    return GLOWTTS_OK;
}

glowtts_conv_args base_args(const Ctx& c, const glowtts_packed& w, int taps) {
    glowtts_conv_args a;
    memset(&a, 0, sizeof(a));
    a.rows = c.R; a.w = w.w; a.npad = w.npad; a.kchunks = w.kchunks; a.taps = taps; a.pad = (taps - 1) / 2;
    a.precision = c.d->precision; a.rowmask = c.a->rowmask; a.rows_per_utt = c.Tp;
    return a;
}

// Start conv + WaveNet + End conv with the coupling epilogue (Modules.py:785-806 / 858-883).
// xsrc: rows whose first C2 channels are x_a and last C2 channels x_b; xdst: where x_b' goes.
int coupling_net(const Ctx& c, const float* xsrc, float* xdst, bool reverse, bool keep) {
    const glowtts_flow_params* p = c.p; const glowtts_flow_acts* A = c.a;
    const int H = c.H, L = c.d->L;
    const bool bf = c.d->act_bf16 != 0;          // hs / gates stored as bf16
    // the whole network in ONE launch when the caller packed this flow's weight image (wavenet_fused.hip; the image's Res_Skip slabs are
    // PAIR-packed, which the per-conv launches below do not read: no silent fallback from here)
    if (p->wn_img) return glowtts_wavenet_fwd(c.d, p, A, xsrc, xdst, reverse ? 1 : 0, keep ? 1 : 0, c.s);
    // Start: h0 = (W x_a + b) * mask                                         Modules.py:791
    {
        glowtts_conv_args a = base_args(c, p->start, 1);
        a.a = xsrc; a.lda = c.d->C; a.ca = c.C2; a.n = H;
        a.epi = GLOWTTS_EPI_LINEAR; a.flags = GLOWTTS_F_BIAS | GLOWTTS_F_MASK; a.bias = p->b_start;
        a.out0 = A->hs[0]; a.ld0 = H; a.io_flags = bf ? GLOWTTS_IO_OUT0_BF16 : 0;
        CHECK(glowtts_conv_cl(&a, c.s));
    }
    for (int l = 0; l < L; ++l) {
        float* hin = keep ? A->hs[l] : A->hs[l & 1];
        float* hout = keep ? (l + 1 < L ? A->hs[l + 1] : nullptr) : A->hs[(l + 1) & 1];
        float* g = keep ? A->gates[l] : A->gates[0];
        float* acts = bf ? (keep ? A->acts[l] : A->acts[0]) : nullptr;
        {   // In_l (k taps) + conditioning + tanh*sigmoid                       Modules.py:861-870
            glowtts_conv_args a = base_args(c, p->in[l], c.d->ksize);
            a.a = hin; a.lda = H; a.ca = H; a.n = 2 * H; a.h = H;
            a.epi = GLOWTTS_EPI_GATE; a.bias = p->b_in[l]; a.drop_p = c.d->drop_p; a.seed = c.d->seed + (uint32_t)l; a.seed_ptr = c.d->seed_ptr;
            if (p->cond) { a.cond = p->cond + (int64_t)l * 2 * H; a.ldcond = p->ldcond; if (p->cond_rows) a.flags |= GLOWTTS_F_COND_ROWS; }
            a.out0 = g; a.ld0 = 2 * H; a.io_flags = bf ? (GLOWTTS_IO_A_BF16 | GLOWTTS_IO_OUT0_BF16) : 0;
            if (acts) { a.out1 = acts; a.ld1 = H; }              // bf16 tanh * sigmoid for the Res_Skip conv below
            CHECK(glowtts_conv_cl(&a, c.s));
        }
        const bool last = (l == L - 1);
        glowtts_conv_args rs = base_args(c, p->rs[l], 1);      // Res_Skip_l on acts = tanh*sigmoid                   Modules.py:871-881
        {
            glowtts_conv_args& a = rs;
            if (acts) { a.a = acts; a.lda = H; a.ca = H; }                                          // plain bf16 operand: LDS-DMA kernel
            else      { a.a = g; a.lda = 2 * H; a.ca = H; a.apro = GLOWTTS_APRO_PAIRMUL; }
            a.n = last ? H : 2 * H; a.h = H;
            a.epi = GLOWTTS_EPI_RESSKIP; a.flags = (l == 0 ? GLOWTTS_F_FIRST : 0) | (last ? GLOWTTS_F_LAST : 0);
            a.bias = p->b_rs[l];
            a.in0 = hin; a.ldi0 = H; a.out0 = last ? A->skip : hout; a.ld0 = H; a.out1 = A->skip; a.ld1 = H;
            a.io_flags = bf ? (GLOWTTS_IO_A_BF16 | GLOWTTS_IO_IN0_BF16 | GLOWTTS_IO_OUT0_BF16) : 0;
            if (bf && last && A->skip_bf) { a.out0 = A->skip_bf; a.ld0 = H; }     // last layer: out0 = bf16 copy of the final skip sum
        }
        if (!last) { CHECK(glowtts_conv_cl(&rs, c.s)); continue; }
        // last layer: Res_Skip, then End + affine coupling                       Modules.py:793-806
        glowtts_conv_args a = base_args(c, p->end, 1);
        if (bf && A->skip_bf) { a.a = A->skip_bf; a.io_flags = GLOWTTS_IO_A_BF16; } else a.a = A->skip;
        a.lda = H; a.ca = H; a.n = c.d->C; a.h = c.C2;
        a.epi = GLOWTTS_EPI_COUPLE; a.flags = reverse ? GLOWTTS_F_REVERSE : 0; a.bias = p->b_end;
        a.in0 = xsrc + c.C2; a.ldi0 = c.d->C;
        a.out0 = xdst + c.C2; a.ld0 = c.d->C;
        a.out1 = keep ? A->outs : nullptr; a.ld1 = p->end.npad;
        // both in ONE launch when the shapes allow (bf16 storage, 192 channels): the final skip sum reaches the End conv through LDS
        const bool chain = GLOWTTS_TUNABLE("GLOWTTS_CHAIN", 1) != 0;
        if (chain && bf && acts && A->skip_bf && glowtts_conv_chain(&rs, &a, c.s) == GLOWTTS_OK) continue;
        CHECK(glowtts_conv_cl(&rs, c.s));
        CHECK(glowtts_conv_cl(&a, c.s));
    }
    return GLOWTTS_OK;
}

__global__ void copy_half_kernel(const float* __restrict__ src, float* __restrict__ dst, long rows, int C, int n)
{
    // dst[r][0:n] = src[r][0:n], float4 granules
    const int q = n / 4;
    const long total = rows * q;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / q; const int c = (int)(i - r * q) * 4;
        *reinterpret_cast<float4*>(dst + r * C + c) = *reinterpret_cast<const float4*>(src + r * C + c);
    }
}
int copy_half(const float* src, float* dst, long rows, int C, int n, void* s) {
    if (src == dst) return GLOWTTS_OK;
    if (n & 3) return GLOWTTS_E_ARG;
    long g = (rows * (n / 4) + 255) / 256; if (g > 2048) g = 2048;
    hipLaunchKernelGGL(copy_half_kernel, dim3((int)g), dim3(256), 0, static_cast<hipStream_t>(s), src, dst, rows, C, n);
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

template <typename XT>
__global__ __launch_bounds__(256) void utt_colsum_kernel(const XT* __restrict__ x, long ldx, float* __restrict__ out, long ldout,
                                                         int rows_per_utt, int n, int perm, int perm_h)
{
    const int b = blockIdx.y;
    const int col = blockIdx.x * 256 + threadIdx.x;          // output (original-order) column
    if (col >= n) return;
    int pc = col;
    if (perm == GLOWTTS_PERM_PAIR) { const int hs = col / perm_h, j = col - hs * perm_h; pc = (j >> 5) * 64 + hs * 32 + (j & 31); }
    const XT* xb = x + (long)b * rows_per_utt * ldx + pc;
    float s = 0.f;
    for (int r = 0; r < rows_per_utt; ++r) s += (float)xb[(long)r * ldx];
    out[(long)b * ldout + col] = s;
}

int utt_colsum(const float* x, int64_t ldx, float* out, int64_t ldout, int B, int rows_per_utt, int n, int perm, int perm_h, bool x_bf16, void* stream)
{
    if (!x || !out || B < 1 || rows_per_utt < 1 || n < 1) return GLOWTTS_E_ARG;
    const dim3 grid((n + 255) / 256, B);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (x_bf16) hipLaunchKernelGGL(utt_colsum_kernel<__bf16>, grid, dim3(256), 0, s, reinterpret_cast<const __bf16*>(x), (long)ldx, out, (long)ldout, rows_per_utt, n, perm, perm_h);
    else hipLaunchKernelGGL(utt_colsum_kernel<float>, grid, dim3(256), 0, s, x, (long)ldx, out, (long)ldout, rows_per_utt, n, perm, perm_h);
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

}  // namespace

extern "C" int glowtts_utt_colsum(const float* x, int64_t ldx, float* out, int64_t ldout, int B, int rows_per_utt, int n,
                                  int perm, int perm_h, void* stream)
{
    return utt_colsum(x, ldx, out, ldout, B, rows_per_utt, n, perm, perm_h, false, stream);
}

extern "C" int glowtts_flow_forward(const glowtts_flow_dims* d, const glowtts_flow_params* p, const glowtts_flow_acts* a, void* stream)
{
    CHECK(check_dims(d));
    if (!p || !a || !a->xin || !a->xmid || !a->xout || !a->rowmask || !a->outs) return GLOWTTS_E_ARG;
    if (!a->skip && !(p->wn_img && a->skip_bf)) return GLOWTTS_E_ARG;      // (the fused launch may keep the bf16 copy of the skip sum only)
    if (d->act_bf16) for (int l = 0; l < d->L; ++l) if (!a->acts[l]) return GLOWTTS_E_ARG;
    const Ctx c = make_ctx(d, p, a, stream);
    // ActNorm + invertible 1x1                                                 Modules.py:693-694, 738-756
    // (x_a passes through, Modules.py:808: written to xout by the same kernel)
    // (actnorm_done: the previous flow's fused coupling launch has already applied it, glowtts_flow_acts.next_*)
    if (a->next_xmid && (!p->wn_img || !a->next_an_logs || !a->next_an_bias || !a->next_winfo || !a->next_xout)) return GLOWTTS_E_ARG;
    if (!a->actnorm_done)
        CHECK(glowtts_actnorm_inv1x1_pass_bf(a->xin, a->xmid, a->xout, a->xa_bf, p->an_logs, p->an_bias, p->winfo, a->rowmask, c.R, d->C, stream));
    return coupling_net(c, a->xmid, a->xout, false, true);
}

extern "C" int glowtts_flow_inverse(const glowtts_flow_dims* d, const glowtts_flow_params* p, const glowtts_flow_acts* a, void* stream)
{
    CHECK(check_dims(d));
    if (!p || !a || !a->xin || !a->xmid || !a->xout || !a->rowmask) return GLOWTTS_E_ARG;
    if (!p->wn_img) {                            // scratch of the per-conv launches (the fused kernel keeps all of it on chip)
        if (!a->skip || !a->hs[0] || !a->hs[1] || !a->gates[0]) return GLOWTTS_E_ARG;
        if (d->act_bf16 && !a->acts[0]) return GLOWTTS_E_ARG;
    }
    const Ctx c = make_ctx(d, p, a, stream);
    // reversed layer order (Modules.py:664): coupling^-1, then inv-1x1^-1, then ActNorm^-1
    CHECK(copy_half(a->xout, a->xmid, c.R, d->C, c.C2, stream));
    CHECK(coupling_net(c, a->xout, a->xmid, true, false));
    return glowtts_actnorm_inv1x1(a->xmid, a->xin, p->an_logs, p->an_bias, p->winfo, a->rowmask, c.R, d->C, 1, stream);
}

extern "C" int glowtts_flow_backward(const glowtts_flow_dims* d, const glowtts_flow_params* p, const glowtts_flow_acts* a,
                                     const glowtts_flow_grads* g, void* stream)
{
    CHECK(check_dims(d));
    if (!p || !a || !g || !g->dx || !g->dlogdet || !g->dskip || !g->dh[0] || !g->dins[0] || !g->scratch) return GLOWTTS_E_ARG;
    const Ctx c = make_ctx(d, p, a, stream);
    const int H = c.H, L = d->L, C = d->C, C2 = c.C2, R = c.R;
    const int ldo = p->end.npad;          // PAIR-packed (m, logs) width
    const int ldin = p->in[0].npad;       // PAIR-packed gate pre-activation width
    auto wargs = [&](const float* dy, int lddy, int m, const float* x, int ldx, int ca, int taps, float* dw, float* db) {
        glowtts_wgrad_args w; memset(&w, 0, sizeof(w));
        w.dy = dy; w.lddy = lddy; w.m = m; w.x = x; w.ldx = ldx; w.ca = ca; w.rows = R; w.taps = taps; w.pad = (taps - 1) / 2;
        w.precision = d->precision; w.splits = 0; w.accumulate = 1; w.dw = dw; w.dbias = db;
        return w;
    };

    const bool bf = d->act_bf16 != 0;         // gates / hs / dins stored as bf16
    const bool bfg = bf;                      // ... and so are dskip and dh[l >= 1] (dh[0] stays fp32: it feeds the fp32 Start conv gradients)
    // 1. affine coupling backward                                               autograd of Modules.py:805-806
    const bool dbf = bfg && g->douts_bf != nullptr;        // a bf16 copy of douts feeds the End data gradient (DMA / chained kernel)
    if (!g->douts && (!dbf || !g->defer_wgrad)) return GLOWTTS_E_ARG;      // (douts == NULL: the bf16 copy alone is kept - every reader here must take it)
    if (g->coupling_done) { /* fused into the previous call's last kernel */ }
    else if (dbf) CHECK(glowtts_coupling_bwd_bf16(g->dx, a->xmid, a->outs, g->douts, g->douts_bf, a->rowmask, g->dlogdet, R, C, ldo, c.Tp, stream));
    else     CHECK(glowtts_coupling_bwd(g->dx, a->xmid, a->outs, g->douts, a->rowmask, g->dlogdet, R, C, ldo, c.Tp, stream));
    // 2-4 in ONE launch when the caller packed the transposed weight image (wavenet_fused_bwd.hip; its per-conv transposed images are then
    // not packed at all: no fallback from here)
    if (p->wn_img_t) {
        CHECK(glowtts_wavenet_bwd(d, p, a, g, stream));
        if (g->prev_outs && g->d_an) return GLOWTTS_E_ARG;
        if (g->prev_outs)
            return glowtts_actnorm_inv1x1_bwd_coupling(g->dx, g->dx, a->xin, p->an_logs, p->an_bias, p->winfo, a->rowmask, g->scratch, R, C,
                                                       g->prev_xmid, g->prev_outs, g->prev_douts, g->prev_douts_bf, g->dlogdet, ldo, c.Tp, stream);
        return glowtts_actnorm_inv1x1_bwd(g->dx, g->dx, a->xin, p->an_logs, p->an_bias, p->winfo, a->rowmask, g->d_an, g->scratch, R, C, stream);
    }
    // 2. End conv: data gradient -> d(skip) (masked), weight gradient
    glowtts_conv_args endq = base_args(c, p->end_t, 1);
    endq.a = dbf ? g->douts_bf : g->douts; endq.lda = ldo; endq.ca = ldo; endq.n = H; endq.epi = GLOWTTS_EPI_LINEAR; endq.flags = GLOWTTS_F_MASK;
    endq.out0 = g->dskip; endq.ld0 = H; endq.io_flags = (dbf ? GLOWTTS_IO_A_BF16 : 0) | (bfg ? GLOWTTS_IO_OUT0_BF16 : 0);
    bool end_done = false;           // (bf16 storage, 192 channels: it runs chained with the last layer's gate derivative below)
    if (!g->defer_wgrad) {
        glowtts_wgrad_args w = wargs(g->douts, ldo, ldo, a->skip, H, H, 1, g->dw_end, g->db_end);          // fp32 operands
        w.perm = GLOWTTS_PERM_PAIR; w.perm_h = C2;
        CHECK(glowtts_wgrad_cl(&w, stream));
    }
    // 3. WaveNet layers, last to first.  dh[l] holds d x_l * mask.
    const bool wg = !g->defer_wgrad;
    for (int l = L - 1; l >= 0; --l) {
        const bool last = (l == L - 1);
        float* dnext = last ? nullptr : g->dh[l + 1];   // d x_{l+1}
        float* dthis = g->dh[l];                        // d x_l (written below)
        float* dins = g->dins[l];
        {   // Res_Skip data gradient + gate derivative -> dins (PAIR-packed (da, ds))
            glowtts_conv_args q = base_args(c, p->rs_t[l], 1);
            if (last) { q.a = g->dskip; q.lda = H; q.ca = H; }
            else      { q.a = dnext; q.lda = H; q.ca1 = H; q.a2 = g->dskip; q.lda2 = H; q.ca = 2 * H; }
            q.n = H; q.epi = GLOWTTS_EPI_DGATE; q.in0 = a->gates[l]; q.ldi0 = 2 * H; q.out0 = dins; q.ld0 = ldin;
            q.drop_p = d->drop_p; q.seed = d->seed + (uint32_t)l; q.seed_ptr = d->seed_ptr;
            q.io_flags = bf ? (GLOWTTS_IO_IN0_BF16 | GLOWTTS_IO_OUT0_BF16 | (bfg ? GLOWTTS_IO_A_BF16 : 0)) : 0;
            // conditioning gradient (autograd of Modules.py:863-866): per-utterance sums of the gate gradients before the dropout mask
            if (g->dcond && p->cond) {
                // (64-bit fixed-point accumulators, glowtts_flow_grads.dcond: layer l starts 2 H elements in)
                q.out1 = reinterpret_cast<float*>(g->dcond + (int64_t)l * 2 * H); q.ld1 = p->ldcond; q.flags |= GLOWTTS_F_COND_FX;
                if (g->pitch_rows) {       // + the Pitch_l weight gradient, into the rows behind the utterances' (see glowtts_flow_grads)
                    if (g->pitch_ns < 1 || g->pitch_ns > 2) return GLOWTTS_E_ARG;
                    q.cond = g->pitch_rows; q.ldcond = g->pitch_ns; q.flags |= GLOWTTS_F_COND_ROWS;
                }
            }
            if (last) {
                const bool chain = GLOWTTS_TUNABLE("GLOWTTS_CHAIN", 1) != 0 && GLOWTTS_TUNABLE("GLOWTTS_CHAIN_BWD", 1) != 0;
                if (!(chain && dbf && glowtts_conv_chain(&endq, &q, stream) == GLOWTTS_OK)) {
                    CHECK(glowtts_conv_cl(&endq, stream));
                    CHECK(glowtts_conv_cl(&q, stream));
                }
                end_done = true;
            } else {
                CHECK(glowtts_conv_cl(&q, stream));
            }
        }
        if (wg) {   // Res_Skip weight gradient: rows [0,H) <- d res, rows [H,2H) <- d skip (last layer: only H rows <- d skip)
            if (last) {
                glowtts_wgrad_args w = bf ? wargs(g->dskip, H, H, a->acts[l], H, H, 1, g->dw_rs[l], g->db_rs[l])
                                          : wargs(g->dskip, H, H, a->gates[l], 2 * H, H, 1, g->dw_rs[l], g->db_rs[l]);
                w.xpro = bf ? GLOWTTS_APRO_NONE : GLOWTTS_APRO_PAIRMUL; w.io_flags = bf ? (GLOWTTS_WIO_X_BF16 | (bfg ? GLOWTTS_WIO_DY_BF16 : 0)) : 0;
                CHECK(glowtts_wgrad_cl(&w, stream));
            } else {
                glowtts_wgrad_args w = bf ? wargs(dnext, H, H, a->acts[l], H, H, 1, g->dw_rs[l], g->db_rs[l])
                                          : wargs(dnext, H, H, a->gates[l], 2 * H, H, 1, g->dw_rs[l], g->db_rs[l]);
                w.xpro = bf ? GLOWTTS_APRO_NONE : GLOWTTS_APRO_PAIRMUL; w.io_flags = bf ? (GLOWTTS_WIO_X_BF16 | (bfg ? GLOWTTS_WIO_DY_BF16 : 0)) : 0;
                CHECK(glowtts_wgrad_cl(&w, stream));
                glowtts_wgrad_args w2 = bf ? wargs(g->dskip, H, H, a->acts[l], H, H, 1, g->dw_rs[l] + (int64_t)H * H, g->db_rs[l] + H)
                                           : wargs(g->dskip, H, H, a->gates[l], 2 * H, H, 1, g->dw_rs[l] + (int64_t)H * H, g->db_rs[l] + H);
                w2.xpro = bf ? GLOWTTS_APRO_NONE : GLOWTTS_APRO_PAIRMUL; w2.io_flags = bf ? (GLOWTTS_WIO_X_BF16 | (bfg ? GLOWTTS_WIO_DY_BF16 : 0)) : 0;
                CHECK(glowtts_wgrad_cl(&w2, stream));
            }
        }
        {   // In_l data gradient: d x_l = (conv^T(dins) + d x_{l+1}) * mask
            glowtts_conv_args q = base_args(c, p->in_t[l], d->ksize);
            q.a = dins; q.lda = ldin; q.ca = ldin; q.n = H; q.epi = GLOWTTS_EPI_LINEAR;
            q.flags = GLOWTTS_F_MASK | (last ? 0 : GLOWTTS_F_ADD_IN0);
            q.in0 = last ? nullptr : dnext; q.ldi0 = H; q.out0 = dthis; q.ld0 = H;
            q.io_flags = bf ? (GLOWTTS_IO_A_BF16 | (bfg && !last ? GLOWTTS_IO_IN0_BF16 : 0) | (bfg && (l > 0 || g->dh0_bf16) ? GLOWTTS_IO_OUT0_BF16 : 0)) : 0;
            CHECK(glowtts_conv_cl(&q, stream));
        }
        if (wg) {   // In_l weight gradient
            glowtts_wgrad_args w = wargs(dins, ldin, ldin, a->hs[l], H, H, d->ksize, g->dw_in[l], g->db_in[l]);
            w.perm = GLOWTTS_PERM_PAIR; w.perm_h = H; w.io_flags = bf ? (GLOWTTS_WIO_DY_BF16 | GLOWTTS_WIO_X_BF16) : 0;
            CHECK(glowtts_wgrad_cl(&w, stream));
        }
    }
    if (!end_done) return GLOWTTS_E_ARG;
    float* dh0 = g->dh[0];                    // d h0 * mask
    // 4. Start conv: data gradient accumulates into d x_a, weight gradient
    {
        glowtts_conv_args q = base_args(c, p->start_t, 1);
        const bool h0bf = bfg && g->dh0_bf16;      // d h0 stored as bf16 rows (the deferred weight gradient then takes (d h0, acts->xa_bf) as raw bf16 operands)
        q.a = dh0; q.lda = H; q.ca = H; q.n = C2; q.epi = GLOWTTS_EPI_LINEAR; q.flags = GLOWTTS_F_ACCUM;
        q.out0 = g->dx; q.ld0 = C; q.io_flags = h0bf ? GLOWTTS_IO_A_BF16 : 0;
        CHECK(glowtts_conv_cl(&q, stream));
        if (wg) {
            if (h0bf && !a->xa_bf) return GLOWTTS_E_ARG;
            glowtts_wgrad_args w = h0bf ? wargs(dh0, H, H, reinterpret_cast<const float*>(a->xa_bf), C2, C2, 1, g->dw_start, g->db_start)
                                        : wargs(dh0, H, H, a->xmid, C, C2, 1, g->dw_start, g->db_start);
            if (h0bf) w.io_flags = GLOWTTS_WIO_DY_BF16 | GLOWTTS_WIO_X_BF16;
            CHECK(glowtts_wgrad_cl(&w, stream));
        }
    }
    // 5. inv-1x1 + ActNorm backward (dx in place), parameter-gradient data terms -> d_an; optionally the next flow's coupling backward
    if (g->prev_outs && g->d_an) return GLOWTTS_E_ARG;      // the fused form leaves the parameter-gradient partials to the caller
    if (g->prev_outs)
        return glowtts_actnorm_inv1x1_bwd_coupling(g->dx, g->dx, a->xin, p->an_logs, p->an_bias, p->winfo, a->rowmask, g->scratch, R, C,
                                                   g->prev_xmid, g->prev_outs, g->prev_douts, g->prev_douts_bf, g->dlogdet, ldo, c.Tp, stream);
    return glowtts_actnorm_inv1x1_bwd(g->dx, g->dx, a->xin, p->an_logs, p->an_bias, p->winfo, a->rowmask, g->d_an, g->scratch, R, C, stream);
}

// Per-step weight preparation of the decoder in ONE launch (round 4): old-style weight normalisation w = g * v / ||v|| (Modules.py:766, 818, 825:
// torch.nn.utils.weight_norm over (in, k) per output channel) fused with the packing of w into the bf16 MFMA tile images the conv kernels read
// (what glowtts_weightnorm_fwd + ~22 glowtts_pack_weight_* launches did: 0.42 ms at the head of every training step, round-3 timeline).
//
// A job is one glowtts_pack_weight_strided call with (v, g) in place of w.  A workgroup owns 32 consecutive PACKED output-channel indices of one
// conv of the job - under the PAIR permutation these are still 32 consecutive source rows - and
//   1. reads those rows of v once, coalesced (a row = I * taps floats, kept in registers), reduces ||v||^2 per row,
//   2. writes the scaled rows as bf16 into an LDS tile [32][I * taps],
//   3. writes the tile's part of the image in 16-byte pieces: forward image: rows n of every (tap, K chunk), 2 KiB runs; transposed image: one K chunk
//      (its 32 k are the tile's rows) of every (tap, n), 12-KiB runs.  The old element-wise gather read 4-byte words 20 B or 3 840 B apart.
// Zero padding (K up to a multiple of 32, N up to a multiple of 64, invalid PAIR rows) is written like the element-wise kernel wrote it.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "../../include/glowtts_hip.h"
#include "launch_log.h"
#include "tunable.h"
#include "device_common.h"

namespace {

constexpr int PREP_NT = 256;
constexpr int PREP_TR = 16;                      // source rows (packed O indices) per workgroup: 31-KiB LDS tiles, five workgroups per CU in different phases
constexpr int PREP_RP = 4;                       // rows of a wave in flight per pass
constexpr int PREP_MAXPL = 16;                   // floats of a row per lane: I * taps <= 1024

__device__ __forceinline__ float prep_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ unsigned short bf16_bits_prep(float v) { const __bf16 b = (__bf16)v; return *reinterpret_cast<const unsigned short*>(&b); }

// The job table travels in the kernel's ARGUMENT segment (<= PREP_MAXJOBS x sizeof(glowtts_prep_job) = 24 x 120 bytes): inside a captured training step a device-side table would
// need a host-to-device copy node in front of the launch, re-executed at every replay, on the critical path of the decoder's stream.
constexpr int PREP_MAXJOBS = GLOWTTS_PREP_MAX_JOBS;
struct prep_table { glowtts_prep_job jobs[PREP_MAXJOBS]; };

// (part of) K chunk kc = tl * PREP_TR / 32 of every (tap, n) of a TRANSPOSED image: piece = (t', n, q): 8 consecutive k = tile rows 8 q ..; n = source channel, taps
// reversed (dgrad)
__device__ __forceinline__ void prep_write_transposed(const unsigned short* tile, int pitch, unsigned char* img, int tl, int taps, int kch, int npad, int I, int tid,
                                                      int abl)
{
    constexpr int QP = PREP_TR / 8;                               // pieces of a 64-byte row this tile owns
    const int kc = (tl * PREP_TR) >> 5, q0 = ((tl * PREP_TR) & 31) >> 3;
    const int npieces = taps * npad * QP;
    for (int pc = tid; pc < npieces; pc += PREP_NT) {
        const int q = pc % QP, tn = pc / QP;
        const int t = tn / npad, n = tn - t * npad;
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        if (n < I) {
            const int src = n * taps + (taps - 1 - t);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t a = tile[(q * 8 + 2 * e) * pitch + src], bb = tile[(q * 8 + 2 * e + 1) * pitch + src];
                w[e] = a | (bb << 16);
            }
        }
        if (!(abl & 1)) *reinterpret_cast<uint4*>(img + ((int64_t)(t * kch + kc) * npad + n) * 64 + (q0 + q) * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// abl (tools builds, GLOWTTS_PREP_ABL): 1 = no image stores (phase 3), 2 = no LDS tile writes (phase 2), 4 = no phase 3 at all
__device__ __forceinline__ void prep_body(const glowtts_prep_job& j, const int rel, const int abl = 0)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char prep_smem[];
    unsigned short* const tile = reinterpret_cast<unsigned short*>(prep_smem);
    const int b = rel / j.tiles, tl = rel - b * j.tiles;             // conv of the batch, tile of PREP_TR packed O indices
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cols = j.I * j.taps;                                   // floats per source row
    const int pitch = cols + 2;                                      // (bf16 units; odd dword pitch: the transposed gather walks rows)
    const float* const vb = j.v + (int64_t)b * j.w_stride;
    // ---- 1 + 2: rows -> norm -> scaled bf16 tile (PREP_RP rows per pass: PREP_RP x nk loads in flight, unconditional - a predicated load makes hipcc
    // branch around it and wait for it before the next one is issued) ----
    const int nk = (cols + 63) >> 6;
    if (!j.g) {
        // a plain weight (no norm to reduce): rows straight into the tile, any row length the tile holds (the text encoder's 768-channel k = 3 conv: 2 304
        // floats per row), four loads in flight per lane
#pragma unroll 1
        for (int rr = 0; rr < PREP_TR / 4; ++rr) {
            const int r = wave * (PREP_TR / 4) + rr, oi = tl * PREP_TR + r;
            bool ok = oi < j.o_ext;
            int o = oi;
            if (j.perm == GLOWTTS_PERM_PAIR) {
                const int p = oi >> 6, hsel = (oi >> 5) & 1, jj = (p << 5) + (oi & 31);
                ok = ok && jj < j.perm_h;
                o = hsel * j.perm_h + jj;
            }
            ok = ok && o < j.O;
            const float* vr = vb + (int64_t)(ok ? o : 0) * cols;
            for (int c0 = 0; c0 < cols; c0 += 256) {
                float x[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { const int c = c0 + lane + 64 * k; x[k] = vr[c < cols ? c : cols - 1]; }
#pragma unroll
                for (int k = 0; k < 4; ++k) { const int c = c0 + lane + 64 * k; if (c < cols) tile[r * pitch + c] = ok ? bf16_bits_prep(x[k]) : (unsigned short)0; }
            }
        }
    } else
#pragma unroll 1
    for (int rr = 0; rr < PREP_TR / 4; rr += PREP_RP) {
        float x[PREP_RP][PREP_MAXPL];
        bool ok[PREP_RP];
        int o[PREP_RP];
#pragma unroll
        for (int h = 0; h < PREP_RP; ++h) {
            const int oi = tl * PREP_TR + wave * (PREP_TR / 4) + rr + h;   // packed index on the O side
            o[h] = oi;
            ok[h] = oi < j.o_ext;
            if (j.perm == GLOWTTS_PERM_PAIR) {
                const int p = oi >> 6, hsel = (oi >> 5) & 1, jj = (p << 5) + (oi & 31);
                ok[h] = ok[h] && jj < j.perm_h;
                o[h] = hsel * j.perm_h + jj;
            }
            ok[h] = ok[h] && o[h] < j.O;
            const float* vr = vb + (int64_t)(ok[h] ? o[h] : 0) * cols;
#pragma unroll
            for (int k = 0; k < PREP_MAXPL; ++k) {
                x[h][k] = 0.f;
                if (k < nk) {
                    const int c = lane + 64 * k;
                    x[h][k] = vr[c < cols ? c : cols - 1];
                }
            }
        }
        // (round 6: the four rows' reductions side by side and their g loads issued with the row loads above - same arithmetic, same bits; measured neutral:
        //  58-61 us per launch either way.  tools/bench_prep.py ablations: rows -> norm -> tile alone 41 us of the 58, the image writes 16, i.e. the kernel is
        //  bound by its load phase at ~3 TB/s with 4 workgroups per CU - 119 VGPRs -, not by the LDS traffic of the transposition)
        float gv[PREP_RP], ss[PREP_RP];
#pragma unroll
        for (int h = 0; h < PREP_RP; ++h) gv[h] = (j.g && ok[h]) ? j.g[(int64_t)b * j.g_stride + o[h]] : 0.f;
#pragma unroll
        for (int h = 0; h < PREP_RP; ++h) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < PREP_MAXPL; ++k) {
                const int c = lane + 64 * k;
                x[h][k] = (ok[h] && c < cols) ? x[h][k] : 0.f;
                s += x[h][k] * x[h][k];
            }
            ss[h] = s;
        }
        if (j.g) {
#pragma unroll
            for (int o_ = 32; o_ > 0; o_ >>= 1) {
#pragma unroll
                for (int h = 0; h < PREP_RP; ++h) ss[h] += __shfl_xor(ss[h], o_, 64);
            }
        }
#pragma unroll
        for (int h = 0; h < PREP_RP; ++h) {
            const int r = wave * (PREP_TR / 4) + rr + h;
            float sc = 1.f;
            if (j.g) {
                const float inv = ok[h] ? 1.f / sqrtf(ss[h]) : 0.f;
                sc = ok[h] ? gv[h] * inv : 0.f;
                if (j.inv_out && ok[h] && lane == 0) j.inv_out[(int64_t)b * j.g_stride + o[h]] = inv;
            }
#pragma unroll
            for (int k = 0; k < PREP_MAXPL; ++k) {
                const int c = lane + 64 * k;
                if (k < nk && c < cols && !(abl & 2)) tile[r * pitch + c] = bf16_bits_prep(x[h][k] * sc);
            }
        }
    }
    __syncthreads();
    if (abl & 4) return;
    // ---- 3: the tile's part of the image ----
    unsigned char* const img = static_cast<unsigned char*>(j.packed) + (int64_t)(b / j.inner) * j.outer_stride + (int64_t)(b % j.inner) * j.inner_stride;
    const int taps = j.taps, kch = j.kchunks, npad = j.npad, I = j.I;
    if (!j.transpose) {
        // rows n = tl * PREP_TR .. of every (tap, K chunk): piece = (t, kc, r, q): 8 consecutive k = source channels kc * 32 + 8 q ..
        const int npieces = taps * kch * (PREP_TR * 4);
        for (int pc = tid; pc < npieces; pc += PREP_NT) {
            const int q = pc & 3, r = (pc >> 2) % PREP_TR, tk = pc / (PREP_TR * 4);
            const int t = tk / kch, kc = tk - t * kch;
            const int c0 = kc * 32 + q * 8;
            uint32_t w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ca = c0 + 2 * e, cb = ca + 1;
                const uint32_t a = ca < I ? tile[r * pitch + ca * taps + t] : 0u, bb = cb < I ? tile[r * pitch + cb * taps + t] : 0u;
                w[e] = a | (bb << 16);
            }
            if (!(abl & 1)) *reinterpret_cast<uint4*>(img + ((int64_t)(t * kch + kc) * npad + tl * PREP_TR + r) * 64 + q * 16) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    } else {
        prep_write_transposed(tile, pitch, img, tl, taps, kch, npad, I, tid, abl);
    }
    if (!j.transpose && j.twin && b < j.twin_batch) {
        // twin output: this forward tile's 16 PAIR-packed rows are 16 consecutive source rows of half hsel (rows hsel * perm_h + 32 p + 16 s ..): tile 2 p + s
        // of that half's transposed image
        const int oi0 = tl * PREP_TR, hsel = (oi0 >> 5) & 1, tl2 = ((oi0 >> 6) << 1) + ((oi0 >> 4) & 1);
        unsigned char* const img2 = static_cast<unsigned char*>(j.twin) + (int64_t)(b / j.inner) * j.twin_outer + (int64_t)(b % j.inner) * j.twin_inner +
                                    (int64_t)hsel * j.twin_half;
        prep_write_transposed(tile, pitch, img2, tl2, taps, (j.perm_h + 31) / 32, j.twin_npad, I, tid, abl);
    }
}

__global__ __launch_bounds__(PREP_NT) void prep_kernel(const prep_table tab, int njobs, int abl)
{
    int lo = 0;
#pragma unroll 1
    for (int i = 1; i < njobs; ++i) lo = tab.jobs[i].block0 <= (int)blockIdx.x ? i : lo;
    const glowtts_prep_job& j = tab.jobs[lo];
#ifdef GLOWTTS_TOOLS
    prep_body(j, blockIdx.x - j.block0, abl);
#else
    prep_body(j, blockIdx.x - j.block0);
#endif
}

// the same with the job table in device memory (a table that does not change between steps - the text encoder's ~60 images: built and uploaded once)
__global__ __launch_bounds__(PREP_NT) void prep_dev_kernel(const glowtts_prep_job* __restrict__ jobs, int njobs)
{
    int lo = 0, hi = njobs - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (jobs[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
    const glowtts_prep_job j = jobs[lo];
    prep_body(j, blockIdx.x - j.block0);
}

constexpr int PREP_MAX_COLS = 4096;              // plain-weight rows: a 16-row bf16 tile of 128 KiB

inline int prep_pad_to(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace

extern "C" int glowtts_prep_job_init(glowtts_prep_job* job, const float* v, const float* g, float* inv_out, int batch, int inner, int O, int I, int taps,
                                     int transpose, int perm, int perm_h, void* packed, int64_t outer_stride, int64_t inner_stride, int64_t w_stride,
                                     int64_t g_stride, int block0, int* blocks_out)
{
    if (!job || batch < 1 || inner < 1 || O < 1 || I < 1 || taps < 1 || taps > 5 || I * taps > (g ? 64 * PREP_MAXPL : PREP_MAX_COLS)) return GLOWTTS_E_ARG;
    if (perm == GLOWTTS_PERM_PAIR && (perm_h < 1 || 2 * perm_h != O)) return GLOWTTS_E_ARG;
    if ((outer_stride & 15) || (inner_stride & 15) || w_stride < 0 || g_stride < 0) return GLOWTTS_E_ARG;
    const int o_ext = (perm == GLOWTTS_PERM_PAIR) ? prep_pad_to(perm_h, 32) * 2 : O;
    const int N = transpose ? I : o_ext, K = transpose ? o_ext : I;
    memset(job, 0, sizeof(*job));
    job->v = v; job->g = g; job->inv_out = inv_out; job->packed = packed;
    job->outer_stride = outer_stride; job->inner_stride = inner_stride;
    job->w_stride = w_stride ? w_stride : (int64_t)O * I * taps;
    job->g_stride = g_stride ? g_stride : O;
    job->batch = batch; job->inner = inner; job->O = O; job->I = I; job->taps = taps; job->transpose = transpose; job->perm = perm; job->perm_h = perm_h;
    job->o_ext = o_ext; job->npad = prep_pad_to(N, 64); job->kchunks = (K + 31) / 32;
    // forward image: every PREP_TR rows of the padded N get a tile (rows past o_ext are zeros); transposed: 32 / PREP_TR tiles per K chunk
    job->tiles = (transpose ? job->kchunks * 32 : job->npad) / PREP_TR;
    job->block0 = block0;
    if (blocks_out) *blocks_out = job->tiles * batch;
    return GLOWTTS_OK;
}

extern "C" int glowtts_prep_jobs_twin_in(glowtts_prep_job* jobs, int* njobs, int* blocks)
{
    if (!jobs || !njobs || !blocks || *njobs < 1) return GLOWTTS_E_ARG;
    int fi = -1, ti[2] = {-1, -1};
    for (int i = 0; i < *njobs; ++i) {
        const glowtts_prep_job& a = jobs[i];
        if (!a.transpose && a.perm == GLOWTTS_PERM_PAIR && a.taps > 1 && a.g && !a.twin && fi < 0) fi = i;
    }
    if (fi < 0) return GLOWTTS_OK;
    const glowtts_prep_job f = jobs[fi];
    const int64_t half_rows = (int64_t)f.perm_h * f.I * f.taps;              // floats between the two halves of one conv
    for (int i = 0; i < *njobs; ++i) {
        const glowtts_prep_job& a = jobs[i];
        if (!a.transpose || a.taps != f.taps || a.I != f.I || a.O != f.perm_h || a.perm != GLOWTTS_PERM_NONE || a.inner != f.inner || a.batch > f.batch) continue;
        for (int h = 0; h < 2; ++h)
            if (a.v == f.v + h * half_rows && a.g == f.g + h * f.perm_h && a.w_stride == f.w_stride && a.g_stride == f.g_stride) ti[h] = i;
    }
    if (ti[0] < 0 || ti[1] < 0) return GLOWTTS_OK;
    const glowtts_prep_job &t0 = jobs[ti[0]], &t1 = jobs[ti[1]];
    if (t0.batch != t1.batch || t0.outer_stride != t1.outer_stride || t0.inner_stride != t1.inner_stride || t0.npad != t1.npad || (f.perm_h & 31) ||
        t0.kchunks != (f.perm_h + 31) / 32) return GLOWTTS_OK;
    jobs[fi].twin = t0.packed;
    jobs[fi].twin_outer = t0.outer_stride; jobs[fi].twin_inner = t0.inner_stride;
    jobs[fi].twin_half = static_cast<const unsigned char*>(t1.packed) - static_cast<const unsigned char*>(t0.packed);
    jobs[fi].twin_batch = t0.batch; jobs[fi].twin_npad = t0.npad;
    // drop the two transposed jobs, renumber the workgroups
    int n = 0, blk = 0;
    for (int i = 0; i < *njobs; ++i) {
        if (i == ti[0] || i == ti[1]) continue;
        jobs[n] = jobs[i];
        jobs[n].block0 = blk;
        blk += jobs[n].tiles * jobs[n].batch;
        ++n;
    }
    *njobs = n; *blocks = blk;
    return GLOWTTS_OK;
}

extern "C" int glowtts_prep_launch_dev(const glowtts_prep_job* dev_jobs, int njobs, int total_blocks, int max_cols, void* stream)
{
    if (!dev_jobs || njobs < 1 || total_blocks < 1 || max_cols < 1 || max_cols > PREP_MAX_COLS) return GLOWTTS_E_ARG;
    const int lds = PREP_TR * (max_cols + 2) * 2;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&prep_dev_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, PREP_TR * (PREP_MAX_COLS + 2) * 2) != hipSuccess) return GLOWTTS_E_LAUNCH;
        attr_done = true;
    }
    GLOWTTS_NOTE_STATIC("prep_weights_dev");
    hipLaunchKernelGGL(prep_dev_kernel, dim3(total_blocks), dim3(PREP_NT), lds, static_cast<hipStream_t>(stream), dev_jobs, njobs);
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

extern "C" int glowtts_prep_launch(const glowtts_prep_job* host_jobs, int njobs, int total_blocks, int max_cols, void* stream)
{
    if (!host_jobs || njobs < 1 || njobs > PREP_MAXJOBS || total_blocks < 1 || max_cols < 1 || max_cols > PREP_MAX_COLS) return GLOWTTS_E_ARG;
    prep_table tab;
    memset(&tab, 0, sizeof(tab));
    memcpy(tab.jobs, host_jobs, (size_t)njobs * sizeof(glowtts_prep_job));
    const int lds = PREP_TR * (max_cols + 2) * 2;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&prep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, PREP_TR * (PREP_MAX_COLS + 2) * 2) != hipSuccess) return GLOWTTS_E_LAUNCH;
        attr_done = true;
    }
    GLOWTTS_NOTE_STATIC("prep_weights");
    hipLaunchKernelGGL(prep_kernel, dim3(total_blocks), dim3(PREP_NT), lds, static_cast<hipStream_t>(stream), tab, njobs, GLOWTTS_TUNABLE("GLOWTTS_PREP_ABL", 0));
    return hipGetLastError() == hipSuccess ? GLOWTTS_OK : GLOWTTS_E_LAUNCH;
}

// The jobs of the per-flow weight images of the fused coupling-network kernels: the same placement as glowtts_wavenet_pack_images /
// glowtts_wavenet_pack_bwd_images, from (g, v) pairs.  Appends to jobs[*njobs ..], advances *njobs and *block.  inv_*: [F (* L)][O] 1 / ||v||
// for the weight-norm backward, written by the forward image's jobs (img_fwd must be given when they are).
extern "C" int glowtts_wavenet_prep_jobs(glowtts_prep_job* jobs, int max_jobs, int* njobs, int* block,
                                         const float* v_start, const float* g_start, float* inv_start, const float* v_in, const float* g_in, float* inv_in,
                                         const float* v_rs, const float* g_rs, float* inv_rs, const float* v_rsl, const float* g_rsl, float* inv_rsl,
                                         const float* w_end, int F, int L, int C2, void* img_fwd, int F_bwd, void* img_bwd)
{
    if (!jobs || !njobs || !block || !v_start || !g_start || !v_in || !g_in || !v_rsl || !g_rsl || !w_end || (L > 1 && (!v_rs || !g_rs)) ||
        F < 1 || L < 1 || L > GLOWTTS_WN_FUSED_MAX_LAYERS || C2 <= 64 || C2 > 96 || (C2 & 3) || F_bwd < 0 || F_bwd > F) return GLOWTTS_E_ARG;
    const int H = 192, T = 5;
    const int64_t S = GLOWTTS_WN_SLAB_BYTES, stride = (int64_t)(36 * L + 2) * S;
    int rc = GLOWTTS_OK;
    auto add = [&](const float* v, const float* g, float* inv, int batch, int inner, int O, int I, int taps, int tr, int perm, int perm_h, unsigned char* at,
                   int64_t inner_stride, int64_t w_stride, int64_t g_stride) {
        if (rc != GLOWTTS_OK) return;
        if (*njobs >= max_jobs) { rc = GLOWTTS_E_ARG; return; }
        int nb = 0;
        rc = glowtts_prep_job_init(&jobs[*njobs], v, g, inv, batch, inner, O, I, taps, tr, perm, perm_h, at, stride, inner_stride, w_stride, g_stride, *block, &nb);
        if (rc == GLOWTTS_OK) { ++*njobs; *block += nb; }
    };
    if (img_fwd) {
        unsigned char* img = static_cast<unsigned char*>(img_fwd);
        add(v_start, g_start, inv_start, F, 1, H, C2, 1, 0, GLOWTTS_PERM_NONE, 0, img, 0, 0, 0);
        add(v_in, g_in, inv_in, F * L, L, 2 * H, H, T, 0, GLOWTTS_PERM_PAIR, H, img + 2 * S, 36 * S, 0, 0);
        if (L > 1) add(v_rs, g_rs, inv_rs, F * (L - 1), L - 1, 2 * H, H, 1, 0, GLOWTTS_PERM_PAIR, H, img + 32 * S, 36 * S, 0, 0);
        add(v_rsl, g_rsl, inv_rsl, F, 1, H, H, 1, 0, GLOWTTS_PERM_NONE, 0, img + (int64_t)(36 * (L - 1) + 32) * S, 0, 0, 0);
        add(w_end, nullptr, nullptr, F, 1, 2 * C2, H, 1, 0, GLOWTTS_PERM_PAIR, C2, img + (int64_t)(36 * (L - 1) + 35) * S, 0, 0, 0);
    }
    if (img_bwd && F_bwd > 0) {
        unsigned char* img = static_cast<unsigned char*>(img_bwd);
        const int Fb = F_bwd;
        add(w_end, nullptr, nullptr, Fb, 1, 2 * C2, H, 1, 1, GLOWTTS_PERM_PAIR, C2, img, 0, 0, 0);
        add(v_rsl, g_rsl, nullptr, Fb, 1, H, H, 1, 1, GLOWTTS_PERM_NONE, 0, img + 3 * S, 0, 0, 0);
        if (L > 1) add(v_rs, g_rs, nullptr, Fb * (L - 1), L - 1, 2 * H, H, 1, 1, GLOWTTS_PERM_NONE, 0, img + (36 + 36 * (int64_t)(L - 2)) * S, -36 * S, 0, 0);
        for (int half = 0; half < 2; ++half)
            add(v_in + (int64_t)half * H * H * T, g_in + half * H, nullptr, Fb * L, L, H, H, T, 1, GLOWTTS_PERM_NONE, 0,
                img + (42 + 36 * (int64_t)(L - 2) + 15 * half) * S, -36 * S, (int64_t)2 * H * H * T, 2 * H);
        add(v_start, g_start, nullptr, Fb, 1, H, C2, 1, 1, GLOWTTS_PERM_NONE, 0, img + 36 * (int64_t)L * S, 0, 0, 0);
    }
    return rc;
}

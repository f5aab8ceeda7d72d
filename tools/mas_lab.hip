// MAS lab (GPU box): phase timing of mas_dp2_kernel (column loop vs backtrack, s_memtime of workgroup 0), a bit-exactness cross-check
// against the general kernel of the built library, and single-wave issue-rate probes (what bounds a one-wave-per-utterance recurrence).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iglow_tts_amd/csrc tools/mas_lab.hip -o tools/_build/mas_lab -ldl
//   tools/_build/mas_lab [B Tx Ty]            (reads glow_tts_amd/libglowtts_hip.so for the cross-check)
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

__device__ unsigned long long g_mas_stamps[64];
#define MAS_LAB_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_mas_stamps[i] = wall_clock64(); } while (0)
#include "../glow_tts_amd/csrc/mas_dp2.hip"
void glowtts_note_launch(const char*) {}

// ---- issue-rate probes: one wave, N instructions between two s_memtime reads ----
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
__global__ void probe_kernel(unsigned long long* out, float* sink)
{
    float a = threadIdx.x, b = 1.f, c = 2.f, d = 3.f, e = 4.f, f = 5.f, g = 6.f, h = 7.f;
    unsigned long long t0, t1;
    // 0: 512 dependent v_add_f32
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 8; ++i) asm volatile(REP64("v_add_f32 %0, %0, %1\n\t") : "+v"(a) : "v"(b));
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    // 1: 512 independent v_add_f32 (8 chains)
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 8; ++i)
        asm volatile(REP8("v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %8\n\tv_add_f32 %2, %2, %8\n\tv_add_f32 %3, %3, %8\n\t"
                          "v_add_f32 %4, %4, %8\n\tv_add_f32 %5, %5, %8\n\tv_add_f32 %6, %6, %8\n\tv_add_f32 %7, %7, %8\n\t")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(1.0f));
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[1] = t1 - t0;
    // 2: 512 dependent (v_cmp -> v_cndmask) pairs = 1024 instructions
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 8; ++i) asm volatile(REP64("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\t") : "+v"(a) : "v"(b) : "vcc");
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[2] = t1 - t0;
    // 3: 512 dependent (v_mov_dpp wave_shr -> v_add) pairs = 1024 instructions
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 8; ++i) asm volatile(REP64("s_nop 1\n\tv_mov_b32_dpp %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32 %0, %0, %1\n\t") : "+v"(a), "+v"(c) : : "vcc");
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[3] = t1 - t0;
    // 4: 512 independent s_add_i32
    int s0 = 1, s1 = 2;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 8; ++i) asm volatile(REP64("s_add_i32 %0, %0, 1\n\t") : "+s"(s0) : : "scc");
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[4] = t1 - t0;
    // 5: 256 x (v_add dependent, s_add) interleaved = 1024 instructions (does SALU issue beside VALU?)
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 8; ++i) asm volatile(REP64("v_add_f32 %0, %0, %2\n\ts_add_i32 %1, %1, 1\n\t") : "+v"(a), "+s"(s1) : "v"(b) : "scc");
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[5] = t1 - t0;
    // 6: the column step of dp2 itself, 512 x (9 VALU), operands in registers
    {
        float q0 = a, q1 = b, up = c; unsigned int b0 = 0, b1 = 0;
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 64; ++i) { REP8(dp2_column(q0, q1, b0, b1, up, d, e);) }
        t1 = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 0) out[6] = t1 - t0;
        a += q0 + q1 + up + b0 + b1;
    }
    sink[threadIdx.x] = a + b + c + d + e + f + g + h + s0 + s1;
}

static float gauss(unsigned long long& st)
{
    auto u = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return ((st >> 11) + 1) * (1.0 / 9007199254740993.0); };
    return (float)(sqrt(-2.0 * log(u())) * cos(6.283185307179586 * u()));
}

int main(int argc, char** argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 32, Tx = argc > 2 ? atoi(argv[2]) : 120, Ty = argc > 3 ? atoi(argv[3]) : 800;
    // probes
    unsigned long long* dout; float* dsink;
    hipMalloc(&dout, 64); hipMalloc(&dsink, 256);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, 0, dout, dsink);
    unsigned long long po[8]; hipMemcpy(po, dout, 56, hipMemcpyDeviceToHost);
    printf("probe (s_memtime ticks, one wave): dependent v_add %.2f / instr; independent v_add %.2f; dependent cmp+cndmask %.2f per instr; "
           "nop1+dpp+add %.2f per triple; s_add %.2f; (v_add dep + s_add) %.2f per pair; dp2 column step %.1f per column\n",
           po[0] / 512.0, po[1] / 512.0, po[2] / 1024.0, po[3] / 512.0, po[4] / 512.0, po[5] / 512.0, po[6] / 512.0);

    // data: value_t [B][Ty][Tx] ~ N(-100, 30), ragged lengths
    std::vector<float> vt((size_t)B * Ty * Tx), vn((size_t)B * Tx * Ty);
    std::vector<int> tx(B), ty(B);
    unsigned long long st = 1234;
    for (int b = 0; b < B; ++b) {
        ty[b] = b == 0 ? Ty : Ty - (int)((st >> 33) % (Ty / 4)); st = st * 6364136223846793005ull + 1;
        tx[b] = b == 0 ? Tx : (int)fmax(1.0, round((double)Tx * ty[b] / Ty));
        for (int y = 0; y < Ty; ++y) for (int x = 0; x < Tx; ++x) {
            float v = (x < tx[b] && y < ty[b]) ? -100.f + 30.f * gauss(st) : 0.f;
            vt[((size_t)b * Ty + y) * Tx + x] = v; vn[((size_t)b * Tx + x) * Ty + y] = v;
        }
    }
    float *dvt, *dvn, *dq, *dqn; int *dtx, *dty, *didx, *didx_ref;
    hipMalloc(&dvt, vt.size() * 4); hipMalloc(&dvn, vn.size() * 4); hipMalloc(&dq, vt.size() * 4); hipMalloc(&dqn, vt.size() * 4);
    hipMalloc(&dtx, B * 4); hipMalloc(&dty, B * 4); hipMalloc(&didx, (size_t)B * Ty * 4); hipMalloc(&didx_ref, (size_t)B * Ty * 4);
    hipMemcpy(dvt, vt.data(), vt.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dvn, vn.data(), vn.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dtx, tx.data(), B * 4, hipMemcpyHostToDevice); hipMemcpy(dty, ty.data(), B * 4, hipMemcpyHostToDevice);
    hipMemset(dq, 0, vt.size() * 4); hipMemset(dqn, 0, vt.size() * 4);
    const size_t lds = (size_t)((Ty + 63) / 64) * 2 * 2 * 64 * 4;

    // cross-check against the library's general kernel (non-transposed entry point)
    void* h = dlopen("glow_tts_amd/libglowtts_hip.so", RTLD_NOW | RTLD_LOCAL);
    if (h) {
        typedef int (*dp_fn)(const float*, const int32_t*, const int32_t*, int32_t*, float*, int, int, int, float, void*);
        dp_fn ref = (dp_fn)dlsym(h, "glowtts_mas_dp_f32");
        int rc = ref(dvn, dtx, dty, didx_ref, dqn, B, Tx, Ty, -1e9f, nullptr);
        int rc2 = glowtts_detail::launch_mas_dp2(dvt, dtx, dty, didx, dq, B, Tx, Ty, -1e9f, lds, 0);
        hipDeviceSynchronize();
        std::vector<int> i0((size_t)B * Ty), i1((size_t)B * Ty); std::vector<float> q0(vt.size()), q1(vt.size());
        hipMemcpy(i0.data(), didx_ref, i0.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(i1.data(), didx, i1.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(q0.data(), dqn, q0.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(q1.data(), dq, q1.size() * 4, hipMemcpyDeviceToHost);
        size_t bad_i = 0, bad_q = 0;
        for (size_t i = 0; i < i0.size(); ++i) bad_i += i0[i] != i1[i];
        for (int b = 0; b < B; ++b) for (int y = 0; y < Ty; ++y) for (int x = 0; x < Tx; ++x) {
            uint32_t u0, u1; memcpy(&u0, &q0[((size_t)b * Tx + x) * Ty + y], 4); memcpy(&u1, &q1[((size_t)b * Ty + y) * Tx + x], 4);
            bad_q += u0 != u1;
        }
        printf("cross-check vs the general kernel (rc %d / %d): %zu of %zu path entries differ, %zu of %zu cumulative scores differ\n", rc, rc2, bad_i, i0.size(), bad_q, q0.size());
    } else printf("library not found: no cross-check (%s)\n", dlerror());

    // timing: events around 50 launches, and the phase stamps of workgroup 0
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 5; ++w) glowtts_detail::launch_mas_dp2(dvt, dtx, dty, didx, nullptr, B, Tx, Ty, -1e9f, lds, 0);
    hipEventRecord(e0, 0);
    for (int w = 0; w < 50; ++w) glowtts_detail::launch_mas_dp2(dvt, dtx, dty, didx, nullptr, B, Tx, Ty, -1e9f, lds, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long stp[64]; hipMemcpyFromSymbol(stp, HIP_SYMBOL(g_mas_stamps), 512);
    printf("mas_dp2 B=%d %dx%d: %.2f us per launch; workgroup 0 (100 MHz wall counter): column loop %.2f us (%.1f ns per column), backtrack %.2f us\n",
           B, Tx, Ty, ms * 1e3 / 50, (stp[1] - stp[0]) / 100.0, (stp[1] - stp[0]) * 10.0 / Ty, (stp[2] - stp[1]) / 100.0);
    printf("  64-column iterations (us):"); for (int it = 0; it + 1 < (Ty + 63) / 64; ++it) printf(" %.2f", (stp[5 + it] - stp[4 + it]) / 100.0); printf("\n");
    return 0;
}

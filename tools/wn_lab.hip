// Coupling-network lab (GPU box, round 4): what bounds a fused WaveNet GEMM loop on one CU?
//   (1) VALU issue rate with 1 / 2 / 3 waves per SIMD (does a second wave fill the other half of a 4-clock VALU slot?)
//   (2) the GEMM loop of a candidate kernel: B (weight) fragments straight from global memory (L2) into registers in MFMA fragment order -
//       no LDS ring, no LDS-DMA, no per-slab barrier - A (state) fragments from an LDS tile; wave tile R x C fragments of 32 x 32.
//       Variants drop one ingredient at a time (no B loads / no A reads / no MFMAs) to see which resource the loop waits for.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/wn_lab.hip -o tools/_build/wn_lab && tools/_build/wn_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <utility>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t Chunk16 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t Rsrc;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ int swz(int row, int q) { return row * 64 + ((q ^ ((row >> 2) & 3)) << 4); }
// slot swizzle under which the fragment reads of v_mfma_f32_16x16x32_bf16 (lane = (row & 15, slot lane >> 4)) are bank-conflict-free at any row shift
__device__ __forceinline__ int swz16(int row, int q) { return row * 64 + ((q ^ (((row >> 2) & 1) << 1)) << 4); }
__device__ __forceinline__ f32x16 mfma(const Chunk16& a, const Chunk16& b, const f32x16& c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
}

// ---------------------------------------------------------------- (1) VALU issue rate
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int KIND>
__global__ void valu_probe(unsigned long long* out, float* sink)
{
    float a = threadIdx.x, b = 1.f, c = 2.f, d = 3.f, e = 4.f, f = 5.f, g = 6.f, h = 7.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 8; ++i) {
        if (KIND == 0)
            asm volatile(REP8("v_fma_f32 %0, %0, %8, %8\n\tv_fma_f32 %1, %1, %8, %8\n\tv_fma_f32 %2, %2, %8, %8\n\tv_fma_f32 %3, %3, %8, %8\n\t"
                              "v_fma_f32 %4, %4, %8, %8\n\tv_fma_f32 %5, %5, %8, %8\n\tv_fma_f32 %6, %6, %8, %8\n\tv_fma_f32 %7, %7, %8, %8\n\t")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(1.0f));
        else if (KIND == 1)
            asm volatile(REP8("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\t"
                              "v_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\t")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
        else if (KIND == 2)
            asm volatile(REP8("v_mul_lo_u32 %0, %0, %8\n\tv_mul_lo_u32 %1, %1, %8\n\tv_mul_lo_u32 %2, %2, %8\n\tv_mul_lo_u32 %3, %3, %8\n\t"
                              "v_mul_lo_u32 %4, %4, %8\n\tv_mul_lo_u32 %5, %5, %8\n\tv_mul_lo_u32 %6, %6, %8\n\tv_mul_lo_u32 %7, %7, %8\n\t")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(3));
        else
            asm volatile(REP8("v_pk_fma_f32 %0, %0, %4, %4\n\tv_pk_fma_f32 %1, %1, %4, %4\n\tv_pk_fma_f32 %2, %2, %4, %4\n\tv_pk_fma_f32 %3, %3, %4, %4\n\t"
                              "v_pk_fma_f32 %0, %0, %4, %4\n\tv_pk_fma_f32 %1, %1, %4, %4\n\tv_pk_fma_f32 %2, %2, %4, %4\n\tv_pk_fma_f32 %3, %3, %4, %4\n\t")
                         : "+v"(*(double*)&a), "+v"(*(double*)&c), "+v"(*(double*)&e), "+v"(*(double*)&g) : "v"(1.0));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2] = t1 - t0; out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2 + 1] = t2 - t0; }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h;
}

// ---------------------------------------------------------------- (2) GEMM loop
constexpr int NW = 12, XR = 68, KCH = 6;
// MODE bits: 1 = no B loads in the loop, 2 = no A reads in the loop, 4 = no MFMAs, 8 = B through the cache with "nt" hint
template <int R, int C, int D, int MODE>
__global__ __launch_bounds__(NW * 64) void gemm_direct(const unsigned char* __restrict__ wimg, float* out, unsigned long long* clk, int nstep)
{
    __shared__ __attribute__((aligned(1024))) unsigned char XT[KCH * XR * 64 + 2048];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    for (int i = tid; i < (KCH * XR * 64) / 4; i += NW * 64) reinterpret_cast<uint32_t*>(XT)[i] = 0x3c003c00u + (i & 7);
    __syncthreads();
    // this wave's weight stream: per step 2 C fragments of 1 KiB ([k step][fragment][lane][16 B])
    const unsigned char* wp = wimg + (size_t)wave * (2 * C * 1024) + lane * 16;
    const size_t step_stride = (size_t)NW * 2 * C * 1024;
    f32x16 acc[R][C];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[r][c][i] = 0.f;
    Chunk16 bq[D][2][C];
    Chunk16 af[2][2][R];
    auto loadB = [&](int slot, int s) __attribute__((always_inline)) {
        const unsigned char* p = wp + (size_t)s * step_stride;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if (MODE & 8) bq[slot][k2][c] = __builtin_nontemporal_load(reinterpret_cast<const Chunk16*>(p + (k2 * C + c) * 1024));
                else bq[slot][k2][c] = *reinterpret_cast<const Chunk16*>(p + (k2 * C + c) * 1024);
            }
    };
    auto loadA = [&](int set, int s) __attribute__((always_inline)) {
        const int kc = s % KCH, t = (s / KCH) % 5;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int r = 0; r < R; ++r)
                af[set][k2][r] = *reinterpret_cast<const Chunk16*>(XT + kc * (XR * 64) + swz(((R == 1 ? (wave & 1) : r) * 32) + l31 + t, 2 * k2 + lhi));
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int d = 0; d < D; ++d) loadB(d, d);
    loadA(0, 0);
    auto step = [&](auto I_, int s) __attribute__((always_inline)) {
        constexpr int i = decltype(I_)::value;
        constexpr int slot = i % D, set = i & 1;
        if (!(MODE & 2)) loadA(set ^ 1, s + 1);
        if (!(MODE & 4)) {
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int c = 0; c < C; ++c) acc[r][c] = mfma(af[set][k2][r], bq[slot][k2][c], acc[r][c]);
        } else {
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int c = 0; c < C; ++c) acc[0][0][0] += __uint_as_float(bq[slot][k2][c][0] ^ af[set][k2][0][1]);
        }
        if (!(MODE & 1)) loadB(slot, s + D);
    };
    constexpr int U = (D % 2 == 0) ? D : 2 * D;
    for (int s = 0; s < nstep; s += U) {
        [&]<int... Is>(std::integer_sequence<int, Is...>) __attribute__((always_inline)) {
            (step(std::integral_constant<int, Is>{}, s + Is), ...);
        }(std::make_integer_sequence<int, U>{});
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) sum += acc[r][c][i];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int c = 0; c < C; ++c) sum += __uint_as_float(bq[d][0][c][0]);
    out[(size_t)blockIdx.x * NW * 64 + tid] = sum;
    if (lane == 0) clk[blockIdx.x * NW + wave] = t1 - t0;
}



// Variant: the B loads of slab s + D are issued at the TOP of step s (into the slot whose MFMAs were issued during step s - 1; D + 1 slots),
// before the wave waits for slab s and before its MFMAs queue up behind the other waves' - the memory pipe never waits for the matrix pipe.
// PIN: sched_barriers keep that order.
template <int D, int PIN, int MODE>
__global__ __launch_bounds__(NW * 64) void gemm_direct2(const unsigned char* __restrict__ wimg, float* out, unsigned long long* clk, int nstep)
{
    constexpr int R = 2, C = 1;
    __shared__ __attribute__((aligned(1024))) unsigned char XT[KCH * XR * 64 + 2048];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    for (int i = tid; i < (KCH * XR * 64) / 4; i += NW * 64) reinterpret_cast<uint32_t*>(XT)[i] = 0x3c003c00u + (i & 7);
    __syncthreads();
    const unsigned char* wp = wimg + (size_t)wave * (2 * C * 1024) + lane * 16;
    const size_t step_stride = (size_t)NW * 2 * C * 1024;
    f32x16 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
    Chunk16 bq[D + 1][2];
    Chunk16 af[2][2][R];
    auto loadB = [&](int slot, int s) __attribute__((always_inline)) {
        const unsigned char* p = wp + (size_t)s * step_stride;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) bq[slot][k2] = *reinterpret_cast<const Chunk16*>(p + k2 * 1024);
    };
    auto loadA = [&](int set, int t, int kc) __attribute__((always_inline)) {
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int r = 0; r < R; ++r)
                af[set][k2][r] = *reinterpret_cast<const Chunk16*>(XT + kc * (XR * 64) + swz(r * 32 + l31 + t, 2 * k2 + lhi));
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int d = 0; d < D; ++d) loadB(d, d);
    loadA(0, 0, 0);
    // nstep = 30 * (D + 1) * m: one "layer" = 30 steps (5 taps x 6 K chunks); the slot / set pattern repeats every lcm(30, D + 1, 2) steps
    constexpr int U = 30 * (D + 1);
    for (int s0 = 0; s0 < nstep; s0 += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int slot = u % (D + 1), set = u & 1, nslot = (u + D) % (D + 1);
            const int un = (u + 1) % 30, t = un / 6, kc = un % 6;
            if (!(MODE & 1)) loadB(nslot, s0 + u + D);
            if (PIN) __builtin_amdgcn_sched_barrier(0);
            loadA(set ^ 1, t, kc);
            if (PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] = mfma(af[set][k2][r], bq[slot][k2], acc[r]);
            if (PIN) __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) sum += acc[r][i];
#pragma unroll
    for (int d = 0; d <= D; ++d) sum += __uint_as_float(bq[d][0][0]);
    out[(size_t)blockIdx.x * NW * 64 + tid] = sum;
    if (lane == 0) clk[blockIdx.x * NW + wave] = t1 - t0;
}

template <int D, int PIN, int MODE>
static void run_gemm2(const char* name, const unsigned char* wimg, float* out, unsigned long long* clk, int grid)
{
    const int nstep = 30 * (D + 1) * (D == 4 ? 1 : (D == 2 ? 2 : (D == 3 ? 1 : 1)));
    const int nslab = nstep;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm_direct2<D, PIN, MODE>), dim3(grid), dim3(NW * 64), 0, 0, wimg, out, clk, nstep);
    CK(hipDeviceSynchronize());
    const int n = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL((gemm_direct2<D, PIN, MODE>), dim3(grid), dim3(NW * 64), 0, 0, wimg, out, clk, nstep);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(grid * NW);
    CK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    printf("%-44s grid %3d, %3d slabs: %7.1f us/launch = %6.1f ns per slab | per slab: median wave %6.0f clk, slowest wave %6.0f clk | %5.0f TFLOP/s\n", name, grid, nslab, ms * 1e3 / n,
           ms * 1e6 / n / nslab, (double)h[h.size() / 2] / nslab, (double)h.back() / nslab, 2.0 * 64 * 384 * 32 * nslab * grid / (ms * 1e-3 / n) * 1e-12);
}


// Role-separated variant: waves [0, 4) only stream the weight image (KIND 0: into VGPRs, 1: LDS-DMA into a ring nobody reads), waves [4, 12) only run MFMAs
// (AREAD: with A fragments read from LDS).  Do the matrix pipe and the memory pipe overlap when no wave waits for both?
template <int KIND, int AREAD, int WHO>
__global__ __launch_bounds__(NW * 64) void mix_roles(const unsigned char* __restrict__ img, float* out, unsigned long long* clk, int nslab)
{
    __shared__ __attribute__((aligned(1024))) unsigned char XT[KCH * XR * 64 + 2048];
    __shared__ __attribute__((aligned(1024))) unsigned char ring[32 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    for (int i = tid; i < (KCH * XR * 64) / 4; i += NW * 64) reinterpret_cast<uint32_t*>(XT)[i] = 0x3c003c00u + (i & 7);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
    if (wave < 4) {
        if (WHO & 1) {
            Chunk16 q[8];
            Chunk16 acc = {0u, 0u, 0u, 0u};
            const int n = nslab * 6;                       // 24 KiB per slab over 4 waves = 6 KiB per wave and slab
            auto ld = [&](int d, int i) __attribute__((always_inline)) {
                const unsigned char* p = img + (size_t)(i * 4 + wave) * 1024 + lane * 16;
                if (KIND == 0) q[d] = *reinterpret_cast<const Chunk16*>(p);
                else __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)p, (void __attribute__((address_space(3)))*)(ring + ((i & 7) * 4 + wave) * 1024), 16, 0, 0);
            };
#pragma unroll
            for (int d = 0; d < 8; ++d) ld(d, d);
            for (int i = 0; i < n; i += 8) {
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    if (KIND == 1) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
                    else acc ^= q[d];
                    ld(d, i + 8 + d);
                }
            }
            if (KIND == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else {
#pragma unroll
                for (int d = 0; d < 8; ++d) acc ^= q[d];
            }
            sum = __uint_as_float(acc[0] ^ acc[1] ^ acc[2] ^ acc[3]);
        }
    } else if (WHO & 2) {
        f32x16 a0, a1, a2;
#pragma unroll
        for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; a2[i] = 0.f; }
        Chunk16 fa = *reinterpret_cast<const Chunk16*>(XT + swz(l31, lhi)), fb = *reinterpret_cast<const Chunk16*>(XT + 4096 + swz(l31, lhi));
        for (int s = 0; s < nslab; ++s) {                  // 48 MFMAs per slab over 8 waves = 6 per wave and slab
            if (AREAD) {
                const int kc = s % KCH;
                Chunk16 f0 = *reinterpret_cast<const Chunk16*>(XT + kc * (XR * 64) + swz(l31 + (s & 3), lhi));
                Chunk16 f1 = *reinterpret_cast<const Chunk16*>(XT + kc * (XR * 64) + swz(l31 + (s & 3), 2 + lhi));
                Chunk16 f2 = *reinterpret_cast<const Chunk16*>(XT + kc * (XR * 64) + swz(32 + l31 + (s & 3), lhi));
                Chunk16 f3 = *reinterpret_cast<const Chunk16*>(XT + kc * (XR * 64) + swz(32 + l31 + (s & 3), 2 + lhi));
                a0 = mfma(f0, fb, a0); a1 = mfma(f1, fb, a1); a2 = mfma(f2, fb, a2);
                a0 = mfma(f3, fb, a0); a1 = mfma(f0, f1, a1); a2 = mfma(f2, f3, a2);
            } else {
                a0 = mfma(fa, fb, a0); a1 = mfma(fa, fb, a1); a2 = mfma(fa, fb, a2);
                a0 = mfma(fa, fb, a0); a1 = mfma(fa, fb, a1); a2 = mfma(fa, fb, a2);
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) sum += a0[i] + a1[i] + a2[i];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[(size_t)blockIdx.x * NW * 64 + tid] = sum;
    if (lane == 0) clk[blockIdx.x * NW + wave] = t1 - t0;
}

template <int KIND, int AREAD, int WHO>
static void run_mix(const char* name, const unsigned char* img, float* out, unsigned long long* clk, int nslab, int grid)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((mix_roles<KIND, AREAD, WHO>), dim3(grid), dim3(NW * 64), 0, 0, img, out, clk, nslab);
    CK(hipDeviceSynchronize());
    const int n = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL((mix_roles<KIND, AREAD, WHO>), dim3(grid), dim3(NW * 64), 0, 0, img, out, clk, nslab);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(grid * NW);
    CK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
    double ml = 0, mm = 0;
    for (int b = 0; b < grid; ++b) for (int w = 0; w < NW; ++w) { if (w < 4) ml = std::max(ml, (double)h[b * NW + w]); else mm = std::max(mm, (double)h[b * NW + w]); }
    printf("roles %-40s grid %3d: %6.1f us/launch = %6.1f ns per slab | slowest loader %5.0f clk per slab, slowest MFMA wave %5.0f clk per slab\n", name, grid, ms * 1e3 / n,
           ms * 1e6 / n / nslab, ml / nslab, mm / nslab);
}


// Same-SIMD test: loader waves = the waves with (wave & 3) == 0 (waves of a workgroup go to the SIMDs round-robin, so these share ONE SIMD), the other
// nine waves (three per remaining SIMD) run MFMAs: 48 per slab over 9 waves -> 16 per three slabs.  NL = number of loader waves that actually load.
template <int KIND, int NL, int WHO>
__global__ __launch_bounds__(NW * 64) void mix_simd(const unsigned char* __restrict__ img, float* out, unsigned long long* clk, int nslab)
{
    __shared__ __attribute__((aligned(1024))) unsigned char XT[KCH * XR * 64 + 2048];
    __shared__ __attribute__((aligned(1024))) unsigned char ring[32 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    for (int i = tid; i < (KCH * XR * 64) / 4; i += NW * 64) reinterpret_cast<uint32_t*>(XT)[i] = 0x3c003c00u + (i & 7);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
    if ((wave & 3) == 0) {
        const int li = wave >> 2;
        if ((WHO & 1) && li < NL) {
            Chunk16 q[8];
            Chunk16 acc = {0u, 0u, 0u, 0u};
            const int n = nslab * 24 / NL;
            const Rsrc rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(img), 0, 0x7fffffff, 0x00020000);
            auto ld = [&](int d, int i) __attribute__((always_inline)) {
                const unsigned char* p = img + (size_t)(i * NL + li) * 1024 + lane * 16;
                if (KIND == 0) q[d] = *reinterpret_cast<const Chunk16*>(p);
                else if (KIND == 2) q[d] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (i * NL + li) * 1024, 0);
                else __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)p, (void __attribute__((address_space(3)))*)(ring + ((i & 7) * NL + li) * 1024), 16, 0, 0);
            };
#pragma unroll
            for (int d = 0; d < 8; ++d) ld(d, d);
            for (int i = 0; i < n; i += 8) {
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    if (KIND == 1) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
                    else acc ^= q[d];
                    ld(d, i + 8 + d);
                }
            }
            if (KIND == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else {
#pragma unroll
                for (int d = 0; d < 8; ++d) acc ^= q[d];
            }
            sum = __uint_as_float(acc[0] ^ acc[1] ^ acc[2] ^ acc[3]);
        }
    } else if (WHO & 2) {
        f32x16 a0, a1, a2, a3;
#pragma unroll
        for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; a2[i] = 0.f; a3[i] = 0.f; }
        Chunk16 fa = *reinterpret_cast<const Chunk16*>(XT + swz(l31, lhi)), fb = *reinterpret_cast<const Chunk16*>(XT + 4096 + swz(l31, lhi));
        for (int s = 0; s < nslab; s += 3) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { a0 = mfma(fa, fb, a0); a1 = mfma(fa, fb, a1); a2 = mfma(fa, fb, a2); a3 = mfma(fa, fb, a3); }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) sum += a0[i] + a1[i] + a2[i] + a3[i];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[(size_t)blockIdx.x * NW * 64 + tid] = sum;
    if (lane == 0) clk[blockIdx.x * NW + wave] = t1 - t0;
}

template <int KIND, int NL, int WHO>
static void run_mix_simd(const char* name, const unsigned char* img, float* out, unsigned long long* clk, int nslab, int grid)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((mix_simd<KIND, NL, WHO>), dim3(grid), dim3(NW * 64), 0, 0, img, out, clk, nslab);
    CK(hipDeviceSynchronize());
    const int n = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL((mix_simd<KIND, NL, WHO>), dim3(grid), dim3(NW * 64), 0, 0, img, out, clk, nslab);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(grid * NW);
    CK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
    double ml = 0, mm = 0;
    for (int b = 0; b < grid; ++b) for (int w = 0; w < NW; ++w) { if ((w & 3) == 0) ml = std::max(ml, (double)h[b * NW + w]); else mm = std::max(mm, (double)h[b * NW + w]); }
    printf("simd  %-40s grid %3d: %6.1f us/launch = %6.1f ns per slab | slowest loader %5.0f clk per slab, slowest MFMA wave %5.0f clk per slab\n", name, grid, ms * 1e3 / n,
           ms * 1e6 / n / nslab, ml / nslab, mm / nslab);
}


// Loaders CO-LOCATED with MFMA waves (waves 0..3 = one loader per SIMD, waves 4..11 = two MFMA waves per SIMD), looking for a form of the weight DMA that a
// SIMD issues while its matrix pipe is busy.  KIND 1: global_load_lds with a VGPR address; 3: the same at s_setprio 3; 4: buffer_load ... lds with NO VGPR
// operand (descriptor with ADD_TID_ENABLE, stride 16: lane i reads base + soffset + 16 i); 5: = 4 at s_setprio 3; 6: MFMA waves run 16x16x32 (4-pass) MFMAs, loader as 1
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int KIND, int DEPTH>
__global__ __launch_bounds__(NW * 64) void mix_co(const unsigned char* __restrict__ img, float* out, unsigned long long* clk, int nslab)
{
    __shared__ __attribute__((aligned(1024))) unsigned char XT[KCH * XR * 64 + 2048];
    __shared__ __attribute__((aligned(1024))) unsigned char ring[64 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    for (int i = tid; i < (KCH * XR * 64) / 4; i += NW * 64) reinterpret_cast<uint32_t*>(XT)[i] = 0x3c003c00u + (i & 7);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
    if (wave < 4) {
        if (KIND == 3 || KIND == 5) __builtin_amdgcn_s_setprio(3);
        const int n = nslab * 6;
        const Rsrc rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(img), 16, 0x7fffffff, 0x00820000);
        auto ld = [&](int i) __attribute__((always_inline)) {
            unsigned char* dst = ring + ((i % DEPTH) * 4 + wave) * 1024;
            if (KIND == 4 || KIND == 5)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, (void __attribute__((address_space(3)))*)dst, 16, 0, (i * 4 + wave) * 1024, 0, 0);
            else
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(img + (size_t)(i * 4 + wave) * 1024 + lane * 16),
                                                 (void __attribute__((address_space(3)))*)dst, 16, 0, 0);
        };
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) ld(d);
        for (int i = 0; i < n; i += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DEPTH - 1) : "memory");
                ld(i + DEPTH + d);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sum = (float)ring[(tid * 16) & 0xffff];
    } else {
        Chunk16 fa = *reinterpret_cast<const Chunk16*>(XT + swz(l31, lhi)), fb = *reinterpret_cast<const Chunk16*>(XT + 4096 + swz(l31, lhi));
        if (KIND == 6) {
            f32x4v a[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) a[k] = f32x4v{0.f, 0.f, 0.f, 0.f};
            for (int s = 0; s < nslab; ++s) {              // 6 x 32x32x16 = 12 x 16x16x32 per wave and slab
#pragma unroll
                for (int k = 0; k < 12; ++k)
                    a[k % 6] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&fa), *reinterpret_cast<const bf16x8*>(&fb), a[k % 6], 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) sum += a[k][0] + a[k][1] + a[k][2] + a[k][3];
        } else {
            f32x16 a0, a1, a2;
#pragma unroll
            for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; a2[i] = 0.f; }
            for (int s = 0; s < nslab; ++s) {
                a0 = mfma(fa, fb, a0); a1 = mfma(fa, fb, a1); a2 = mfma(fa, fb, a2);
                a0 = mfma(fa, fb, a0); a1 = mfma(fa, fb, a1); a2 = mfma(fa, fb, a2);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) sum += a0[i] + a1[i] + a2[i];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[(size_t)blockIdx.x * NW * 64 + tid] = sum;
    if (lane == 0) clk[blockIdx.x * NW + wave] = t1 - t0;
}

template <int KIND, int DEPTH>
static void run_mix_co(const char* name, const unsigned char* img, float* out, unsigned long long* clk, int nslab, int grid)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((mix_co<KIND, DEPTH>), dim3(grid), dim3(NW * 64), 0, 0, img, out, clk, nslab);
    CK(hipDeviceSynchronize());
    const int n = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL((mix_co<KIND, DEPTH>), dim3(grid), dim3(NW * 64), 0, 0, img, out, clk, nslab);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(grid * NW);
    CK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
    double ml = 0, mm = 0;
    for (int b = 0; b < grid; ++b) for (int w = 0; w < NW; ++w) { if (w < 4) ml = std::max(ml, (double)h[b * NW + w]); else mm = std::max(mm, (double)h[b * NW + w]); }
    printf("co    %-40s depth %2d grid %3d: %6.1f us/launch = %6.1f ns per slab | slowest loader %5.0f clk per slab, slowest MFMA wave %5.0f clk per slab\n", name, DEPTH, grid, ms * 1e3 / n,
           ms * 1e6 / n / nslab, ml / nslab, mm / nslab);
}


// Model of today's fused kernel loop (wavenet_fused.hip): 4-slot LDS ring of 24-KiB slabs filled by LDS-DMA (every wave two 1-KiB units per slab, issued
// behind the step's MFMAs), counted vmcnt + one s_barrier per slab, wave tile 32 rows x 64 columns, fragments of slab j + 1 read under the MFMAs of slab j.
// MF16 = 0: 4 x v_mfma_f32_32x32x16_bf16 per wave and slab; 1: the same products as 8 x v_mfma_f32_16x16x32_bf16 (A: 2 x 16 rows, B: 4 x 16 columns).
// RT = 1: wave tile 64 rows x 32 columns instead (6 waves per row... all 12 waves: two row fragments x one column fragment; A 4 KiB + B 2 KiB per slab)
template <int MF16, int DMA_FIRST>
__global__ __launch_bounds__(NW * 64) void ring_loop(const unsigned char* __restrict__ img, float* out, unsigned long long* clk, int nslab)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char sm[];
    unsigned char* const XT = sm;                          // [6][68][64]
    unsigned char* const RING = sm + 32 * 1024;            // 4 x 24 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5, l15 = lane & 15, lq = lane >> 4;
    const int rf = wave >= 6 ? 1 : 0, pi = wave - 6 * rf;
    for (int i = tid; i < (KCH * XR * 64) / 4; i += NW * 64) reinterpret_cast<uint32_t*>(XT)[i] = 0x3c003c00u + (i & 7);
    __syncthreads();
    const int lrow = lane >> 2, qa = (lane & 3) ^ ((lane >> 4) & 3);
    const unsigned char* const wsrc = img + (uint32_t)((wave * 32 + lrow) * 64 + qa * 16);
    auto issue = [&](int s) __attribute__((always_inline)) {
        const unsigned char* src = wsrc + (size_t)s * 24576;
        unsigned char* dst = RING + (s & 3) * 24576 + wave * 2048;
#pragma unroll
        for (int u = 0; u < 2; ++u)
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + u * 1024), (void __attribute__((address_space(3)))*)(dst + u * 1024), 16, 0, 0);
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    issue(0); issue(1); issue(2);
    float sum = 0.f;
    if (!MF16) {
        f32x16 acc0, acc1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
        Chunk16 fa[2][2], fb[2][2][2];
        int bl[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) bl[s2] = swz(l31, 2 * s2 + lhi);
        const int offP = 2 * pi * 2048;
        auto mma = [&](int st) __attribute__((always_inline)) {
            acc0 = mfma(fa[st][0], fb[st][0][0], acc0);
            acc1 = mfma(fa[st][0], fb[st][0][1], acc1);
            acc0 = mfma(fa[st][1], fb[st][1][0], acc0);
            acc1 = mfma(fa[st][1], fb[st][1][1], acc1);
        };
        for (int s = 0; s < nslab; s += 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                const unsigned char* slot = RING + ((s + h) & 3) * 24576;
                const int kc = (s + h) % KCH, t = ((s + h) / KCH) % 5;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    fa[h][s2] = *reinterpret_cast<const Chunk16*>(XT + kc * (XR * 64) + swz(rf * 32 + l31 + t, 2 * s2 + lhi));
                    fb[h][s2][0] = *reinterpret_cast<const Chunk16*>(slot + offP + bl[s2]);
                    fb[h][s2][1] = *reinterpret_cast<const Chunk16*>(slot + offP + 2048 + bl[s2]);
                }
                if (DMA_FIRST) issue(s + h + 3);
                mma(h ^ 1);
                if (!DMA_FIRST) issue(s + h + 3);
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) sum += acc0[i] + acc1[i];
    } else {
        f32x4v acc[2][4];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[r][c] = f32x4v{0.f, 0.f, 0.f, 0.f};
        Chunk16 fa[2][2], fb[2][4];                        // [set][16-row fragment], [set][16-column fragment]: one k step of 32 per slab
        auto mma = [&](int st) __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&fa[st][r]), *reinterpret_cast<const bf16x8*>(&fb[st][c]), acc[r][c], 0, 0, 0);
        };
        for (int s = 0; s < nslab; s += 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                const unsigned char* slot = RING + ((s + h) & 3) * 24576;
                const int kc = (s + h) % KCH, t = ((s + h) / KCH) % 5;
#pragma unroll
                for (int r = 0; r < 2; ++r) fa[h][r] = *reinterpret_cast<const Chunk16*>(XT + kc * (XR * 64) + (MF16 == 2 ? swz16(rf * 32 + r * 16 + l15 + t, lq) : swz(rf * 32 + r * 16 + l15 + t, lq)));
#pragma unroll
                for (int c = 0; c < 4; ++c) fb[h][c] = *reinterpret_cast<const Chunk16*>(slot + (MF16 == 2 ? swz16(pi * 64 + c * 16 + l15, lq) : swz(pi * 64 + c * 16 + l15, lq)));
                if (DMA_FIRST) issue(s + h + 3);
                mma(h ^ 1);
                if (!DMA_FIRST) issue(s + h + 3);
            }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) sum += acc[r][c][0] + acc[r][c][1] + acc[r][c][2] + acc[r][c][3];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[(size_t)blockIdx.x * NW * 64 + tid] = sum;
    if (lane == 0) clk[blockIdx.x * NW + wave] = t1 - t0;
}

template <int MF16, int DMA_FIRST>
static void run_ring(const char* name, const unsigned char* img, float* out, unsigned long long* clk, int nslab, int grid)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int lds = 32 * 1024 + 4 * 24576;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ring_loop<MF16, DMA_FIRST>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((ring_loop<MF16, DMA_FIRST>), dim3(grid), dim3(NW * 64), lds, 0, img, out, clk, nslab);
    CK(hipDeviceSynchronize());
    const int n = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL((ring_loop<MF16, DMA_FIRST>), dim3(grid), dim3(NW * 64), lds, 0, img, out, clk, nslab);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(grid * NW);
    CK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    printf("ring  %-44s grid %3d: %6.1f us/launch = %6.1f ns per slab | median wave %5.0f clk per slab | %5.0f TFLOP/s\n", name, grid, ms * 1e3 / n,
           ms * 1e6 / n / nslab, (double)h[h.size() / 2] / nslab, 2.0 * 64 * 384 * 32 * nslab * grid / (ms * 1e-3 / n) * 1e-12);
}


// ---------------------------------------------------------------- (4) throughput of the transposing LDS read (what bounds the weight-gradient kernels?)
// PAT 0: ds_read_b64, lane-linear (reference);  1: ds_read_b64_tr_b16, the register-staged wgrad kernel's pattern (32x32x16 fragments: 16-lane group g reads 4
// rows x 32 B, groups 0 / 1 = two column halves of the same rows, row pitch 320 B);  2: wgrad_dma_kernel's DY pattern (16x16x32 fragments: group g reads rows
// r + 8 g .., swizzled 256-byte rows);  3: its X pattern (128-byte rows) at tap shift 1
typedef short s16x4l __attribute__((ext_vector_type(4)));
template <int PAT>
__global__ __launch_bounds__(1024) void tr_probe(unsigned long long* out, float* sink)
{
    __shared__ __attribute__((aligned(1024))) unsigned char sm[64 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 16 * 1024; i += blockDim.x) reinterpret_cast<uint32_t*>(sm)[i] = i;
    __syncthreads();
    const int s = lane & 15, g = lane >> 4, lhi = lane >> 5, gq = (lane >> 4) & 1;
    int addr;
    if (PAT == 0) addr = lane * 8;
    else if (PAT == 1) addr = (8 * lhi + (s >> 2)) * 320 + (gq * 16 + 4 * (s & 3)) * 2;
    else if (PAT == 2) { const int row = 8 * g + (s >> 2); addr = row * 256 + ((((0 ^ (row & 3)) & 3) << 2 | ((0 ^ ((row >> 3) & 1)) << 1) | ((s & 3) >> 1)) << 4) + 8 * (s & 1); }
    else { const int row = 8 * g + (s >> 2) + 1; addr = row * 128 + (((((0 ^ ((row >> 1) & 1)) & 1) << 2) | ((0 ^ ((row >> 3) & 1)) << 1) | ((s & 3) >> 1)) << 4) + 8 * (s & 1); }
    addr += (wave & 3) * 8192;
    s16x4l acc = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s16x4l v;
            if (PAT == 0) v = *reinterpret_cast<const s16x4l*>(sm + addr + u * 512 * (PAT == 0));
            else v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4l __attribute__((address_space(3)))*)(sm + addr + u * 16 * (PAT == 1 ? 320 : (PAT == 2 ? 256 : 128)) % 8192));
            acc ^= v;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[(blockIdx.x * 16 + wave) * 2] = t1 - t0; out[(blockIdx.x * 16 + wave) * 2 + 1] = t2 - t0; }
    sink[blockIdx.x * blockDim.x + tid] = (float)(acc[0] ^ acc[1] ^ acc[2] ^ acc[3]);
}

// ---------------------------------------------------------------- (3) how fast can ONE CU take in a weight stream from L2?
// KIND 0: global_load_dwordx4 -> VGPR; 1: buffer_load_dwordx4; 2: buffer_load sc1; 3: buffer_load sc0 sc1; 4: global_load_lds_dwordx4 (LDS-DMA into a ring,
// nothing reads it); 5: global_load_dwordx2 -> VGPR; 6: buffer_load nt
template <int KIND, int DEPTH>
__global__ __launch_bounds__(1024) void ingest(const unsigned char* __restrict__ img, float* out, unsigned long long* clk, int n)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char ring[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = blockDim.x >> 6;
    const Rsrc rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(img), 0, 0x7fffffff, 0x00020000);
    Chunk16 q[DEPTH];
    Chunk16 acc = {0u, 0u, 0u, 0u};
    auto ld = [&](int d, int i) __attribute__((always_inline)) {
        const uint32_t off = (uint32_t)((i * nwv + wave) * 1024 + lane * 16);
        if (KIND == 0) q[d] = *reinterpret_cast<const Chunk16*>(img + off);
        else if (KIND == 1) q[d] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        else if (KIND == 2) q[d] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
        else if (KIND == 3) q[d] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 17);
        else if (KIND == 6) q[d] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 2);
        else if (KIND == 4)
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(img + off),
                                             (void __attribute__((address_space(3)))*)(ring + ((i % 4) * nwv + wave) * 1024), 16, 0, 0);
        else {
            const u_int64_t* p = reinterpret_cast<const u_int64_t*>(img + (size_t)(i * nwv + wave) * 1024 + lane * 8);
            const u_int64_t a = p[0], b = p[64];
            q[d][0] = (uint32_t)a; q[d][1] = (uint32_t)(a >> 32); q[d][2] = (uint32_t)b; q[d][3] = (uint32_t)(b >> 32);
        }
    };
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) ld(d, d);
    for (int i = 0; i < n; i += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (KIND == 4) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DEPTH - 1) : "memory");
            else acc ^= q[d];
            ld(d, i + DEPTH + d);
        }
    }
    if (KIND == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) if (KIND != 4) acc ^= q[d];
    out[(size_t)blockIdx.x * 1024 + tid] = __uint_as_float(acc[0] ^ acc[1] ^ acc[2] ^ acc[3]);
    if (lane == 0) clk[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int KIND, int DEPTH>
static void run_ingest(const char* name, const unsigned char* img, float* out, unsigned long long* clk, int kib, int nwv, int grid)
{
    const int n = kib / nwv;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ingest<KIND, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((ingest<KIND, DEPTH>), dim3(grid), dim3(nwv * 64), 64 * 1024, 0, img, out, clk, n);
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((ingest<KIND, DEPTH>), dim3(grid), dim3(nwv * 64), 64 * 1024, 0, img, out, clk, n);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(grid * 16);
    CK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
    double mx = 0;
    for (int b = 0; b < grid; ++b) for (int w = 0; w < nwv; ++w) mx = std::max(mx, (double)h[b * 16 + w]);
    printf("ingest %-28s depth %2d, %2d waves, grid %3d: %6.1f us/launch, slowest wave %7.0f clk -> %5.1f B/clk/CU, %5.1f GB/s/CU by the events, chip %5.2f TB/s\n", name, DEPTH, nwv, grid,
           ms * 1e3 / reps, mx, kib * 1024.0 / mx, kib * 1024.0 / (ms * 1e-3 / reps) * 1e-9, kib * 1024.0 * grid / (ms * 1e-3 / reps) * 1e-12);
}

template <int R, int C, int D, int MODE>
static void run_gemm(const char* name, const unsigned char* wimg, float* out, unsigned long long* clk, int nslab, int grid)
{
    // a "slab" = 64 rows x 384 columns x 32 k = 48 MFMAs per workgroup; a wave step = 2 R C MFMAs, so steps per wave = 48 nslab / (12 * 2 R C)
    const int nstep = nslab * 48 / (NW * 2 * R * C);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm_direct<R, C, D, MODE>), dim3(grid), dim3(NW * 64), 0, 0, wimg, out, clk, nstep);
    CK(hipDeviceSynchronize());
    const int n = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL((gemm_direct<R, C, D, MODE>), dim3(grid), dim3(NW * 64), 0, 0, wimg, out, clk, nstep);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(grid * NW);
    CK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2], mx = (double)h.back();
    printf("%-44s grid %3d: %7.1f us/launch | per slab: median wave %6.0f clk, slowest wave %6.0f clk | %5.0f TFLOP/s\n", name, grid, ms * 1e3 / n,
           med / nslab, mx / nslab, 2.0 * 64 * 384 * 32 * nslab * grid / (ms * 1e-3 / n) * 1e-12);
}

int main(int argc, char** argv)
{
    const int nslab = argc > 1 ? atoi(argv[1]) : 144;
    // ---- (1)
    {
        unsigned long long* out; float* sink;
        CK(hipMalloc(&out, 256 * 16 * 2 * 8)); CK(hipMalloc(&sink, 256 * 1024 * 4));
        const char* names[4] = {"v_fma_f32", "v_exp_f32", "v_mul_lo_u32", "v_pk_fma_f32"};
        for (int kind = 0; kind < 4; ++kind)
            for (int nw : {1, 4, 8, 12, 16}) {
                for (int rep = 0; rep < 2; ++rep) {
                    if (kind == 0) hipLaunchKernelGGL(valu_probe<0>, dim3(256), dim3(nw * 64), 0, 0, out, sink);
                    if (kind == 1) hipLaunchKernelGGL(valu_probe<1>, dim3(256), dim3(nw * 64), 0, 0, out, sink);
                    if (kind == 2) hipLaunchKernelGGL(valu_probe<2>, dim3(256), dim3(nw * 64), 0, 0, out, sink);
                    if (kind == 3) hipLaunchKernelGGL(valu_probe<3>, dim3(256), dim3(nw * 64), 0, 0, out, sink);
                }
                CK(hipDeviceSynchronize());
                std::vector<unsigned long long> h(256 * 16 * 2);
                CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
                double own = 0, all = 0;
                for (int w = 0; w < nw; ++w) { own += (double)h[(7 * 16 + w) * 2]; all = std::max(all, (double)h[(7 * 16 + w) * 2 + 1]); }
                own /= nw;
                const double per_simd = (nw + 3) / 4;
                printf("%-13s %2d waves/CU (%.0f per SIMD): one wave's 512 instructions take %6.0f clk (%.2f clk each); all done after %6.0f clk = %.2f clk per instruction and SIMD\n",
                       names[kind], nw, per_simd, own, own / 512, all, all / (512 * per_simd));
            }
    }
    // ---- (2)
    unsigned char* wimg; float* out; unsigned long long* clk;
    const size_t wbytes = (size_t)(nslab + 16) * 24576 * 2;
    CK(hipMalloc(&wimg, wbytes)); CK(hipMalloc(&out, 256 * 1024 * 4)); CK(hipMalloc(&clk, 256 * 16 * 8));
    {
        std::vector<unsigned short> h(wbytes / 2);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3c00 + ((i * 2654435761u) >> 27));     // bf16 values near 0.01: random-ish mantissas
        CK(hipMemcpy(wimg, h.data(), wbytes, hipMemcpyHostToDevice));
    }


    if (argc > 2 && !strcmp(argv[2], "g2")) {
        for (int grid : {249, 8}) {
            run_gemm<2, 1, 3, 0>("R2 C1 D3 (baseline of the first run)", wimg, out, clk, 144, grid);
            run_gemm2<2, 0, 0>("loads first, D2", wimg, out, clk, grid);
            run_gemm2<3, 0, 0>("loads first, D3", wimg, out, clk, grid);
            run_gemm2<4, 0, 0>("loads first, D4", wimg, out, clk, grid);
            run_gemm2<2, 1, 0>("loads first, D2, pinned", wimg, out, clk, grid);
            run_gemm2<3, 1, 0>("loads first, D3, pinned", wimg, out, clk, grid);
            run_gemm2<4, 1, 0>("loads first, D4, pinned", wimg, out, clk, grid);
            run_gemm2<3, 1, 1>("pinned D3, no B loads in the loop", wimg, out, clk, grid);
        }
        return 0;
    }

    if (argc > 2 && !strcmp(argv[2], "roles")) {
        for (int grid : {249, 8}) {
            run_mix<0, 0, 1>("loaders only (VGPR)", wimg, out, clk, 144, grid);
            run_mix<1, 0, 1>("loaders only (LDS-DMA)", wimg, out, clk, 144, grid);
            run_mix<0, 0, 2>("MFMA waves only", wimg, out, clk, 144, grid);
            run_mix<0, 1, 2>("MFMA waves only, A from LDS", wimg, out, clk, 144, grid);
            run_mix<0, 0, 3>("both (VGPR loads)", wimg, out, clk, 144, grid);
            run_mix<1, 0, 3>("both (LDS-DMA)", wimg, out, clk, 144, grid);
            run_mix<0, 1, 3>("both (VGPR loads), A from LDS", wimg, out, clk, 144, grid);
            run_mix<1, 1, 3>("both (LDS-DMA), A from LDS", wimg, out, clk, 144, grid);
        }
        return 0;
    }

    if (argc > 2 && !strcmp(argv[2], "simd")) {
        for (int grid : {249, 8}) {
            run_mix_simd<0, 3, 1>("3 loaders alone (VGPR)", wimg, out, clk, 144, grid);
            run_mix_simd<1, 3, 1>("3 loaders alone (LDS-DMA)", wimg, out, clk, 144, grid);
            run_mix_simd<1, 1, 1>("1 loader alone (LDS-DMA)", wimg, out, clk, 144, grid);
            run_mix_simd<0, 3, 2>("9 MFMA waves alone (3 SIMDs)", wimg, out, clk, 144, grid);
            run_mix_simd<0, 3, 3>("3 loaders (VGPR) + 9 MFMA waves", wimg, out, clk, 144, grid);
            run_mix_simd<2, 3, 3>("3 loaders (buffer, SGPR offset) + 9 MFMA", wimg, out, clk, 144, grid);
            run_mix_simd<1, 3, 3>("3 loaders (LDS-DMA) + 9 MFMA waves", wimg, out, clk, 144, grid);
            run_mix_simd<1, 2, 3>("2 loaders (LDS-DMA) + 9 MFMA waves", wimg, out, clk, 144, grid);
            run_mix_simd<1, 1, 3>("1 loader (LDS-DMA) + 9 MFMA waves", wimg, out, clk, 144, grid);
        }
        return 0;
    }

    if (argc > 2 && !strcmp(argv[2], "co")) {
        for (int grid : {249, 8}) {
            run_mix_co<1, 8>("global_load_lds, VGPR address", wimg, out, clk, 144, grid);
            run_mix_co<3, 8>("the same, loader at s_setprio 3", wimg, out, clk, 144, grid);
            run_mix_co<4, 8>("buffer_load lds, no VGPR (add_tid)", wimg, out, clk, 144, grid);
            run_mix_co<4, 16>("buffer_load lds, no VGPR (add_tid)", wimg, out, clk, 144, grid);
            run_mix_co<5, 8>("no VGPR + s_setprio 3", wimg, out, clk, 144, grid);
            run_mix_co<6, 8>("VGPR address, MFMA waves on 16x16x32", wimg, out, clk, 144, grid);
        }
        return 0;
    }


    if (argc > 2 && !strcmp(argv[2], "tr")) {
        unsigned long long* o2; float* sk2;
        CK(hipMalloc(&o2, 256 * 16 * 2 * 8)); CK(hipMalloc(&sk2, 256 * 1024 * 4));
        const char* nm[4] = {"ds_read_b64 linear", "tr_b16, staged-kernel pattern", "tr_b16, dma-kernel DY pattern", "tr_b16, dma-kernel X pattern (tap 1)"};
        for (int pat = 0; pat < 4; ++pat)
            for (int nw : {4, 8, 16}) {
                for (int rep = 0; rep < 2; ++rep) {
                    if (pat == 0) hipLaunchKernelGGL(tr_probe<0>, dim3(256), dim3(nw * 64), 0, 0, o2, sk2);
                    if (pat == 1) hipLaunchKernelGGL(tr_probe<1>, dim3(256), dim3(nw * 64), 0, 0, o2, sk2);
                    if (pat == 2) hipLaunchKernelGGL(tr_probe<2>, dim3(256), dim3(nw * 64), 0, 0, o2, sk2);
                    if (pat == 3) hipLaunchKernelGGL(tr_probe<3>, dim3(256), dim3(nw * 64), 0, 0, o2, sk2);
                }
                CK(hipDeviceSynchronize());
                std::vector<unsigned long long> h(256 * 16 * 2);
                CK(hipMemcpy(h.data(), o2, h.size() * 8, hipMemcpyDeviceToHost));
                double all = 0;
                for (int w = 0; w < nw; ++w) all = std::max(all, (double)h[(7 * 16 + w) * 2 + 1]);
                printf("tr    %-40s %2d waves: 512 reads per wave, all done after %7.0f clk = %5.1f B/clk/CU\n", nm[pat], nw, all, 512.0 * 512 * nw / all);
            }
        return 0;
    }
    if (argc > 2 && !strcmp(argv[2], "ring")) {
        for (int grid : {249, 8}) {
            run_ring<0, 0>("today: 32x32x16, DMA behind the MFMAs", wimg, out, clk, 144, grid);
            run_ring<0, 1>("32x32x16, DMA before the MFMAs", wimg, out, clk, 144, grid);
            run_ring<1, 0>("16x16x32, DMA behind the MFMAs", wimg, out, clk, 144, grid);
            run_ring<1, 1>("16x16x32, DMA before the MFMAs", wimg, out, clk, 144, grid);
            run_ring<2, 0>("16x16x32, conflict-free swizzle, DMA behind", wimg, out, clk, 144, grid);
            run_ring<2, 1>("16x16x32, conflict-free swizzle, DMA before", wimg, out, clk, 144, grid);
        }
        return 0;
    }
    // ---- (3)
    if (argc > 2) {
        const int kib = 3456;
        for (int grid : {249, 8}) {
            for (int nwv : {4, 8, 12, 16}) {
                run_ingest<0, 8>("global_load_dwordx4", wimg, out, clk, kib, nwv, grid);
                run_ingest<0, 4>("global_load_dwordx4", wimg, out, clk, kib, nwv, grid);
                run_ingest<1, 8>("buffer_load_dwordx4", wimg, out, clk, kib, nwv, grid);
                run_ingest<2, 8>("buffer_load_dwordx4 sc1", wimg, out, clk, kib, nwv, grid);
                run_ingest<3, 8>("buffer_load_dwordx4 sc0 sc1", wimg, out, clk, kib, nwv, grid);
                run_ingest<6, 8>("buffer_load_dwordx4 nt", wimg, out, clk, kib, nwv, grid);
                run_ingest<4, 4>("global_load_lds_dwordx4", wimg, out, clk, kib, nwv, grid);
                run_ingest<4, 2>("global_load_lds_dwordx4", wimg, out, clk, kib, nwv, grid);
                run_ingest<5, 8>("2 x global_load_dwordx2", wimg, out, clk, kib, nwv, grid);
            }
        }
        return 0;
    }
    for (int grid : {249, 64, 8}) {
        run_gemm<2, 1, 3, 0>("R2 C1 D3 (12 waves x 64 rows x 32 cols)", wimg, out, clk, nslab, grid);
        run_gemm<2, 1, 4, 0>("R2 C1 D4", wimg, out, clk, nslab, grid);
        run_gemm<2, 1, 6, 0>("R2 C1 D6", wimg, out, clk, nslab, grid);
        run_gemm<2, 1, 4, 8>("R2 C1 D4 nt loads", wimg, out, clk, nslab, grid);
        run_gemm<2, 1, 4, 1>("R2 C1 D4 no B loads", wimg, out, clk, nslab, grid);
        run_gemm<2, 1, 4, 2>("R2 C1 D4 no A reads", wimg, out, clk, nslab, grid);
        run_gemm<2, 1, 4, 3>("R2 C1 D4 MFMAs only", wimg, out, clk, nslab, grid);
        run_gemm<2, 1, 4, 4>("R2 C1 D4 no MFMAs", wimg, out, clk, nslab, grid);
        run_gemm<2, 2, 2, 0>("R2 C2 D2 (K split: 6 tiles 64 x 64, 2 K halves)", wimg, out, clk, nslab, grid);
        run_gemm<2, 2, 3, 0>("R2 C2 D3", wimg, out, clk, nslab, grid);
        run_gemm<2, 2, 3, 1>("R2 C2 D3 no B loads", wimg, out, clk, nslab, grid);
        run_gemm<1, 2, 4, 0>("R1 C2 D4 (today's tiling, B loaded by 2 waves)", wimg, out, clk, nslab, grid);
    }
    return 0;
}

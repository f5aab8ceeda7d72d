// Effective shader clock under different loads: every workgroup stamps clock64() (s_memtime, shader cycles) and wall_clock64() (100 MHz)
// around a loop of (a) s_sleep, (b) dependent v_fma, (c) bf16 MFMAs on random data.  build: hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o tools/_build/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ __launch_bounds__(256) void probe(long long* out, const float* seed, int iters)
{
    const long long c0 = clock64(), w0 = wall_clock64();
    if (MODE == 0) { for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(8); }
    else if (MODE == 1) { float x = seed[threadIdx.x]; for (int i = 0; i < iters * 16; ++i) x = __builtin_fmaf(x, 1.0001f, 0.5f); if (x == 12345.f) out[0] = 1; }
    else {
        f32x16 acc[4] = {};
        bf16x8 a, b;
        for (int k = 0; k < 8; ++k) { a[k] = (__bf16)seed[(threadIdx.x * 8 + k) & 4095]; b[k] = (__bf16)seed[(threadIdx.x * 8 + k + 1777) & 4095]; }
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
        }
        float s = 0; for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
        if (s == 12345.f) out[0] = 1;
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[2 * blockIdx.x + 2] = c1 - c0; out[2 * blockIdx.x + 3] = w1 - w0; }
}
int main(int argc, char** argv)
{
    const int nwg = argc > 1 ? atoi(argv[1]) : 512;
    const int only = argc > 2 ? atoi(argv[2]) : -1;
    long long* d; float* seed;
    hipMalloc(&d, (2 * nwg + 4) * 8); hipMalloc(&seed, 4096 * 4);
    std::vector<float> h(4096); srand(1); for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
    hipMemcpy(seed, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    std::vector<long long> r(2 * nwg + 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"s_sleep", "v_fma", "mfma bf16"};
    for (int mode = 0; mode < 3; ++mode) {
        if (only >= 0 && mode != only) continue;
        for (int iters : {200, 2000, 20000, 200000, 2000000}) {
            if (mode == 0 && iters > 200000) continue;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) probe<0><<<nwg, 256>>>(d, seed, iters);
                if (mode == 1) probe<1><<<nwg, 256>>>(d, seed, iters);
                if (mode == 2) probe<2><<<nwg, 256>>>(d, seed, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                hipMemcpy(r.data(), d, r.size() * 8, hipMemcpyDeviceToHost);
                double c = 0, w = 0; for (int i = 0; i < nwg; ++i) { c += r[2 * i + 2]; w += r[2 * i + 3]; }
                printf("%-10s iters %8d rep %d: kernel %9.1f us  mean cycles/wg %12.0f  wall ticks %10.0f  -> shader clock %.3f GHz%s\n", names[mode], iters, rep, ms * 1e3,
                       c / nwg, w / nwg, (c / w) * 0.1, mode == 2 ? "" : "");
                long long cmin = r[2], cmax = r[2]; for (int i = 0; i < nwg; ++i) { cmin = std::min(cmin, r[2 * i + 2]); cmax = std::max(cmax, r[2 * i + 2]); }
                if (mode == 2 && rep == 1) printf("           cycles per MFMA issued by one wave: mean %.1f  min %.1f  max %.1f   (256-thread WGs: one wave per SIMD per WG)\n", (c / nwg) / (iters * 4.0), cmin / (iters * 4.0), cmax / (iters * 4.0));
            }
        }
    }
    return 0;
}

// Is the SGPR offset of a raw buffer store part of the hardware bounds check on gfx950?  (LLVM documents soffset as "excluded from bounds
// checking"; the epilogues of gemm_cl.hip address rows through soffset.)  hipcc --offload-arch=gfx950 tools/probe_soffset.hip -o tools/probe_soffset
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(float* buf, int nrec_bytes, int voff, int soff)
{
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf, 0, nrec_bytes, 0x00020000);
    if (threadIdx.x == 0) __builtin_amdgcn_raw_buffer_store_b32(0x3f800000u, r, (uint32_t)voff, soff, 0);
}
int main()
{
    float* d; hipMalloc(&d, 4096); 
    struct { int voff, soff; const char* what; } cases[] = {
        {0, 0, "voffset 0, soffset 0 (in range)"}, {252, 0, "last element via voffset"}, {256, 0, "one past the end via voffset"},
        {0, 256, "one past the end via SOFFSET"}, {128, 256, "voffset in range + soffset past the end"}, {(int)0x80000000u, 0, "OOB marker"}, {(int)0x80000000u, 64, "OOB marker + soffset"}};
    for (auto& c : cases) {
        hipMemset(d, 0, 4096);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 256, c.voff, c.soff);
        hipDeviceSynchronize();
        float h[1024]; hipMemcpy(h, d, 4096, hipMemcpyDeviceToHost);
        int where = -1; for (int i = 0; i < 1024; ++i) if (h[i] != 0.f) where = i;
        printf("%-45s -> %s (element %d)\n", c.what, where < 0 ? "dropped" : "WRITTEN", where);
    }
    return 0;
}

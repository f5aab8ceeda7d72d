// Probe: pins the lane->element mapping of ds_read_b64_tr_b16 on gfx950 (used by the wgrad kernel).
// LDS holds u16 value = its own element index; every lane passes its own byte address.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short v4s __attribute__((ext_vector_type(4)));

__global__ void probe(const int* addr_in, int* out)
{
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    int a = addr_in[threadIdx.x];   // element index (u16 units), must be multiple of 4
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(lds + a));
    for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = (unsigned short)r[e];
}

int main()
{
    int h_addr[64], h_out[256];
    int *d_addr, *d_out;
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    // experiment 1: lane l -> row l (row stride 64 elements), column block 0: addr = l*64
    // experiment 2: 16-lane groups: lane l -> addr = (l&15)*64 + (l>>4)*4
    for (int exp = 0; exp < 3; ++exp) {
        for (int l = 0; l < 64; ++l) {
            if (exp == 0) h_addr[l] = l * 64;
            if (exp == 1) h_addr[l] = (l & 15) * 64 + (l >> 4) * 4;
            if (exp == 2) h_addr[l] = (l & 3) * 64 + ((l >> 2) & 3) * 4 + (l >> 4) * 1024;
        }
        hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        printf("exp %d\n", exp);
        for (int l = 0; l < 64; ++l)
            printf("  lane %2d addr %4d -> %4d %4d %4d %4d   (row,col) = (%d,%d) (%d,%d) (%d,%d) (%d,%d)\n", l, h_addr[l],
                   h_out[l*4], h_out[l*4+1], h_out[l*4+2], h_out[l*4+3],
                   h_out[l*4]/64, h_out[l*4]%64, h_out[l*4+1]/64, h_out[l*4+1]%64, h_out[l*4+2]/64, h_out[l*4+2]%64, h_out[l*4+3]/64, h_out[l*4+3]%64);
    }
    return 0;
}

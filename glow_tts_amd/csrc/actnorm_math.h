// ActNorm + invertible 1x1 conv of one channel group (Modules.py:693-694, 738-756), shared by flow_ops.hip (the pass of its own) and
// wavenet_fused.hip (the same arithmetic in the previous flow's coupling epilogue): every multiply-add is an EXPLICIT fma in a fixed order, so
// that the two kernels produce the same bits whatever the compiler would contract on its own (a last-bit difference in x_mid becomes a
// 1e-3 difference three flows later through the bf16 operands).
#pragma once
__device__ __forceinline__ void actnorm_mix4(const float (&x)[4], const float (&e)[4], const float (&b)[4], const float (&w)[16], float m, float (&o)[4])
{
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = __builtin_fmaf(e[k], x[k], b[k]) * m;                         // Modules.py:693
#pragma unroll
    for (int k = 0; k < 4; ++k)
        o[k] = __builtin_fmaf(w[k * 4 + 3], v[3], __builtin_fmaf(w[k * 4 + 2], v[2], __builtin_fmaf(w[k * 4 + 1], v[1], w[k * 4 + 0] * v[0]))) * m;   // :749-756
}

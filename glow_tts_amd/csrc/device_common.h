// Device helpers shared by the MFMA kernels (gemm_cl.hip, wavenet_fused.hip): vector types, the LDS slot swizzle, bf16 packing,
// transcendental wrappers and the dropout hashes that the forward and backward epilogues must agree on.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef uint32_t Chunk16 __attribute__((ext_vector_type(4)));      // one 16-byte LDS slot (native vector: stays in VGPRs)

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return *reinterpret_cast<uint32_t*>(&v);
}

__device__ __forceinline__ int swz(int row, int q) { return row * 64 + ((q ^ ((row >> 2) & 3)) << 4); }

// EXACT = f32 mode: libm-grade transcendental functions; bf16 mode: hardware exp (v_exp_f32)
template <bool EXACT> __device__ __forceinline__ float exp_(float x) { return EXACT ? expf(x) : __expf(x); }
template <bool EXACT> __device__ __forceinline__ float rcp_(float x) { return EXACT ? 1.f / x : __builtin_amdgcn_rcpf(x); }
template <bool EXACT> __device__ __forceinline__ float sigmoid_(float x) { return rcp_<EXACT>(1.f + exp_<EXACT>(-x)); }
template <bool EXACT> __device__ __forceinline__ float tanh_(float x) {
    if (EXACT) return tanhf(x);
    const float e = __expf(2.f * x);          // tanh(x) = 1 - 2 / (exp(2x) + 1)
    return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
}

typedef __amdgpu_buffer_rsrc_t Rsrc;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// Dropout of the WaveNet gate pre-activations (Modules.py:862).  The GATE (forward) and DGATE (backward) epilogues must draw the
// same mask, and integer multiplies are quarter rate, so: one key per row (2 multiplies), one 32-bit draw per (row, channel
// pair) (1 multiply) whose low / high 16 bits decide the tanh / the sigmoid channel.  keep <=> half >= thr, thr = round(p * 2^16);
// kept values are scaled by 2^16 / (2^16 - thr), the exact inverse keep rate of that threshold.
__device__ __forceinline__ uint32_t drop_threshold(float p) { return p > 0.f ? (uint32_t)(p * 65536.f + 0.5f) : 0u; }
__device__ __forceinline__ float drop_inv_keep(uint32_t thr) { return 65536.f / (65536.f - (float)thr); }
__device__ __forceinline__ uint32_t drop_rowkey(uint32_t seed, uint32_t r) { uint32_t x = r * 0x9E3779B1u + seed; x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; return x; }
__device__ __forceinline__ uint32_t drop_colkey(uint32_t j) { return (j + 1u) * 0x27D4EB2Fu; }
__device__ __forceinline__ uint32_t drop_draw(uint32_t rowkey, uint32_t colkey) { const uint32_t x = (rowkey ^ colkey) * 0xC2B2AE35u; return x ^ (x >> 16); }
__device__ __forceinline__ float drop_keep_lo(uint32_t d, uint32_t thr, float ik) { return (d & 0xFFFFu) >= thr ? ik : 0.f; }
__device__ __forceinline__ float drop_keep_hi(uint32_t d, uint32_t thr, float ik) { return (d >> 16) >= thr ? ik : 0.f; }

// dropout keep decision of the LINEAR epilogue: counter hash (murmur3 finalizer) of (seed, element id) -> 24-bit uniform
__device__ __forceinline__ float drop_scale(uint32_t seed, uint32_t id, float p, float inv_keep) {
    uint32_t h = id * 0x9E3779B1u + seed;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return ((h >> 8) * (1.0f / 16777216.0f) >= p) ? inv_keep : 0.f;
}


}  // namespace

// Conditioning-gradient accumulators (round 5): fixed point, units of 2^-40, 64-bit integer atomics.  Integer addition commutes, so the per-utterance sums that the
// workgroups of a launch (and the launches of the twelve flows' layers) add into one buffer no longer depend on the order they arrive in: the gradients of the
// Speaker_l / Prosody_l / Pitch_l convs are bit-reproducible from run to run (fp32 atomic adds were not).  Range +- 2.1e6 (beyond: poisoned, below), resolution 9.1e-13 per addend.
constexpr float GLOWTTS_FX_SCALE = 1099511627776.f;          // 2^40
// A non-finite addend, or one beyond +- 2^21 (2^61 units - a quarter of the range), POISONS the accumulator instead of vanishing in the float -> integer conversion
// (NaN -> 0, Inf -> saturation): it is set to the largest value, later addends only move it within |acc| >= 2^61, and glowtts_fx_to_float (cond_ops.hip)
// reads such an accumulator as NaN - the Speaker_l / Prosody_l / Pitch_l gradients propagate a diverging step like fp32 sums did.
constexpr long long GLOWTTS_FX_POISON = 0x2000000000000000LL;        // 2^61
__device__ __forceinline__ void fx_atomic_add(long long* dst, float v) {
    const float s = v * GLOWTTS_FX_SCALE;
    if (!(fabsf(s) < 2.3e18f)) { atomicExch(reinterpret_cast<unsigned long long*>(dst), 0x7FFFFFFFFFFFFFFFull); return; }
    atomicAdd(reinterpret_cast<unsigned long long*>(dst), (unsigned long long)__float2ll_rn(s));
}


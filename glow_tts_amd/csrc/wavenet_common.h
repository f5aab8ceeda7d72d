// Constants and device helpers shared by the fused coupling-network kernels (wavenet_fused.hip: forward / inverse, wavenet_fused_bwd.hip: backward).
#pragma once
#include "device_common.h"
#include "../../include/glowtts_hip.h"

namespace {

constexpr int WN_H = 192;                         // Calc_Channels the kernel is written for
constexpr int WN_KCH = WN_H / 32;                 // K chunks (64 B of bf16) per 192 channels
constexpr int WN_TAPS = 5, WN_PAD = 2;
constexpr int WN_WIN = 64;                        // rows of the compute window (two 32-row MFMA fragments)
constexpr int WN_XR = WN_WIN + 2 * WN_PAD;        // rows of the state tile
constexpr int WN_SLAB = GLOWTTS_WN_SLAB_BYTES;    // one weight slab
constexpr int WN_NS = 4;                          // ring slots
constexpr int WN_NW = 12;                         // waves per workgroup
constexpr int WN_NT = WN_NW * 64;
constexpr int WN_MAXL = GLOWTTS_WN_FUSED_MAX_LAYERS;
constexpr int WN_SROWS = 96;                      // rows of the Start conv's operand tile (three fragments cover the 68 state rows)

typedef __amdgpu_buffer_rsrc_t Rsrc;
template <int V> struct IC { static constexpr int value = V; };
constexpr uint32_t OOB = 0x80000000u;
// f(IC<0>{}), f(IC<1>{}), ... f(IC<N - 1>{}): straight-line code over compile-time step numbers
template <int N> struct StaticForN {
    template <class F> __device__ __forceinline__ static void run(F&& f) { StaticForN<N - 1>::run(f); f(IC<N - 1>{}); }
};
template <> struct StaticForN<0> { template <class F> __device__ __forceinline__ static void run(F&&) {} };

__device__ __forceinline__ Rsrc mk_rsrc(const void* ptr, long bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, (int)bytes, 0x00020000); }
__device__ __forceinline__ int frag_row(int reg) { return (reg & 3) + 8 * (reg >> 2); }       // + 4 * (lane >> 5): row of accumulator element `reg`
template <bool ON = true>
__device__ __forceinline__ f32x16 mfma_bf16(const Chunk16& a, const Chunk16& b, const f32x16& c) {
    if constexpr (!ON) { f32x16 r = c; r[0] += __uint_as_float(a[0] ^ b[0]); return r; }
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
}
// the same product on the 4-pass instruction: A = 16 rows x 32 k (lane = (row & 15, k slot lane >> 4)), B = 16 columns x 32 k, C element i = row 4 (lane >> 4) + i,
// column lane & 15.  A SIMD keeps issuing vector-memory instructions between these; between back-to-back 32x32x16 MFMAs it does not (tools/wn_lab.hip)
template <bool ON = true>
__device__ __forceinline__ f32x4 mfma16_bf16(const Chunk16& a, const Chunk16& b, const f32x4& c) {
    if constexpr (!ON) { f32x4 r = c; r[0] += __uint_as_float(a[0] ^ b[0]); return r; }
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
}
// LDS slot swizzle of [rows][64 B] tiles read as 16x16x32 fragments: bank-conflict free for ds_read_b128 at any row shift (period 8 rows)
__device__ __forceinline__ int swz16(int row, int q) { return row * 64 + ((q ^ (((row >> 2) & 1) << 1)) << 4); }
// Fragment reads the COMPILER DOES NOT SEE (inline asm), paired with a hand-placed counted wait: hipcc's own s_waitcnt insertion puts lgkmcnt(0) in front
// of the MFMAs of slab j where lgkmcnt(6) is meant (the six reads of slab j + 1, issued right before, may stay in flight) - the software pipeline then
// waits out an LDS round trip every other slab.  `addr`: LDS byte address; OFF: immediate offset (< 65536).  lgkm_wait<N>(a, b) waits until at most N
// LGKM operations are outstanding and is tied to the fragment registers, so no use of them can be scheduled in front of it.
template <int OFF> __device__ __forceinline__ Chunk16 lds16_asm(uint32_t addr) {
    Chunk16 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int N> __device__ __forceinline__ void lgkm_wait(Chunk16 (&a)[2], Chunk16 (&b)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N));
}
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)p; }
__device__ __forceinline__ Chunk16 lds16(const unsigned char* p) { return *reinterpret_cast<const Chunk16*>(p); }
template <class T> __device__ __forceinline__ T pick4(T const (&a)[4], int l) { return l == 0 ? a[0] : (l == 1 ? a[1] : (l == 2 ? a[2] : a[3])); }   // (no dynamic indexing of kernel arguments)
__device__ __forceinline__ unsigned short bf16_bits(float v) { const __bf16 b = (__bf16)v; return *reinterpret_cast<const unsigned short*>(&b); }


}  // namespace

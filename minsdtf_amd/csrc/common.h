// Shared device / host helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/minsdtf_hip.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;  // MFMA 16x16x32 A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;   // MFMA 16x16 accumulator fragment
typedef uint16_t bf16_t;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN-preserving
    return __builtin_bit_cast(bf16_t, b);
}
typedef __bf16 msd_bf16x2 __attribute__((ext_vector_type(2)));
typedef float msd_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    const msd_f32x2 v = {lo, hi};  // ONE v_cvt_pk_bf16_f32 (the scalar form costs cvt + cvt + or)
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, msd_bf16x2));
}
__device__ __forceinline__ float bf_lo(uint32_t v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf_hi(uint32_t v) { return __uint_as_float(v & 0xFFFF0000u); }

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float silu_f(float v) { return v * fast_rcp(1.0f + __expf(-v)); }
// tanh-approximated GELU gate of the reference (diffusion_model.py:151-153)
__device__ __forceinline__ float geglu_f(float x, float gate) {
    float u = gate * 0.7978845608f * (1.0f + 0.044715f * gate * gate);
    float th = 1.0f - 2.0f * fast_rcp(1.0f + __expf(2.0f * u));
    return x * 0.5f * gate * (1.0f + th);
}

// unpack 8 bf16 (one 16-byte vector) to floats
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
    f[0] = bf_lo(v.x); f[1] = bf_hi(v.x); f[2] = bf_lo(v.y); f[3] = bf_hi(v.y);
    f[4] = bf_lo(v.z); f[5] = bf_hi(v.z); f[6] = bf_lo(v.w); f[7] = bf_hi(v.w);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
    uint4 v;
    v.x = pack_bf2(f[0], f[1]); v.y = pack_bf2(f[2], f[3]);
    v.z = pack_bf2(f[4], f[5]); v.w = pack_bf2(f[6], f[7]);
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// Bijective XCD-aware remap of a linear workgroup id (cdna_hip_programming.md T1): ids b and b+8
// share an XCD, so XCD x gets the contiguous work items [start(x), start(x+1)).
__device__ __forceinline__ int xcd_remap(int bid, int n) {
    const int xcd = bid & 7, loc = bid >> 3;
    const int q = n >> 3, r = n & 7;
    return xcd * q + min(xcd, r) + loc;   // = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc, branch-free
}

// x / d for 0 <= x < 2^31 with magic = floor(2^32 / d) precomputed on the host: umulhi gives floor(x/d) or one less,
// one compare fixes it — 4 instructions instead of the ~25 dependent ones of an integer division by a runtime value
// (the kernels' prologues and epilogues are latency chains of exactly such scalar code).
__device__ __forceinline__ int udiv_magic(int x, int d, uint32_t magic) {
    int q = (int)__umulhi((uint32_t)x, magic);
    if (x - q * d >= d) ++q;
    return q;
}
static inline uint32_t udiv_magic_of(int d) { return d <= 1 ? 0xFFFFFFFFu : (uint32_t)((1ull << 32) / (uint64_t)d); }

// ---- host side ---------------------------------------------------------------------------
void msd_set_error(const char* fmt, ...);
#define MSD_FAIL(code, ...)          \
    do {                             \
        msd_set_error(__VA_ARGS__);  \
        return (code);               \
    } while (0)
#define MSD_CHECK_LAUNCH()                                            \
    do {                                                              \
        hipError_t e__ = hipGetLastError();                           \
        if (e__ != hipSuccess) {                                      \
            msd_set_error("launch failed: %s", hipGetErrorString(e__)); \
            return (int)e__;                                          \
        }                                                             \
    } while (0)
static inline bool msd_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

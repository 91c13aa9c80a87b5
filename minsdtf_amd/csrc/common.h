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
// 0.5 (1 + tanh u) = sigmoid(2 u) = 1 / (1 + 2^(-2 u log2 e)), u = sqrt(2/pi) (g + 0.044715 g^3): 6 VALU + 2 transcendental
// instructions per element (the GEGLU projections evaluate 10.5 M of them per UNet forward at 64x64: the epilogue's VALU
// time is of the order of the layer's MFMA time)
__device__ __forceinline__ float geglu_f(float x, float gate) {
    constexpr float C1 = 2.0f * 0.7978845608f * 1.4426950408889634f, C2 = C1 * 0.044715f;
    const float w = gate * (C1 + C2 * (gate * gate));   // 2 u log2(e)
    return (x * gate) * fast_rcp(1.0f + __builtin_amdgcn_exp2f(-w));
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

// Wave-wide sum / max, result on every lane, without the LDS crossbar: four DPP steps inside each 16-lane row (quad
// permutes, row rotates: fused into the v_add / v_max) and two row swaps (v_permlane16_swap / v_permlane32_swap).
// __shfl_xor compiles to ds_bpermute_b32, ~100+ cycles of latency per step in the middle of a dependency chain (six steps
// per reduction: these small normalisation kernels are latency chains).  Fixed order -> bit-reproducible.
#define MSD_DPP(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xF, 0xF, true))
__device__ __forceinline__ float wave_sum(float v) {
    v += MSD_DPP(v, 0xB1);    // quad_perm [1,0,3,2]
    v += MSD_DPP(v, 0x4E);    // quad_perm [2,3,0,1]
    v += MSD_DPP(v, 0x124);   // row_ror:4
    v += MSD_DPP(v, 0x128);   // row_ror:8
    const uint32_t u = __float_as_uint(v);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const uint32_t w = __float_as_uint(v);
    auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, MSD_DPP(v, 0xB1));
    v = fmaxf(v, MSD_DPP(v, 0x4E));
    v = fmaxf(v, MSD_DPP(v, 0x124));
    v = fmaxf(v, MSD_DPP(v, 0x128));
    const uint32_t u = __float_as_uint(v);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const uint32_t w = __float_as_uint(v);
    auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
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

// ---- LDS-DMA helpers (compiler-invisible on purpose: see cdna_hip_programming.md §5.7) ---------
// One wave instruction copies 64 x 16 B from per-lane global addresses to LDS bytes
// [lds_dst, lds_dst + 1024) in lane order.  M0 carries the LDS base and is restored.
__device__ __forceinline__ void dma16(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}
// same, source = wave-uniform 64-bit base (SGPR pair) + per-lane 32-bit byte offset
__device__ __forceinline__ void dma16s(const void* sbase, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_dst)
        : "memory");
}
// same under an explicit lane mask (the wave must be fully active at the call): lanes off write nothing
__device__ __forceinline__ void dma16sm(const void* sbase, uint32_t voff, uint32_t lds_dst, uint64_t lanes) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_mov_b64 exec, %4\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_dst), "s"(lanes)
        : "memory");
}
// per-lane 64-bit sources under an explicit lane mask (the wave must be fully active at the call)
__device__ __forceinline__ void dma16m(const void* gsrc, uint32_t lds_dst, uint64_t lanes) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_mov_b64 exec, %3\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst), "s"(lanes)
        : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most min(later, MAXT) tiles of L DMA instructions each are still in flight
template <int L, int MAXT>
__device__ __forceinline__ void wait_vmcnt_tiles(int later) {
    if constexpr (MAXT <= 0) {
        wait_vmcnt<0>();
    } else {
        if (later >= MAXT) wait_vmcnt<(MAXT * L < 63 ? MAXT * L : 63)>();
        else wait_vmcnt_tiles<L, MAXT - 1>(later);
    }
}

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

// Cycles per instruction of the bf16 MFMA shapes (and of v_exp_f32 next to them) on gfx950: one wave per SIMD, s_memtime around
// an unrolled loop.  Used to decide the attention kernel's QK^T shape; not part of the product library.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE>
__global__ __launch_bounds__(256) void k_rate(unsigned long long* cyc, float* sink, int iters) {
    bf16x8 a = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x}, b = {2, 3, 4, 5, 6, 7, 8, (short)(threadIdx.x * 3)};
    bf16x4 a4 = {1, 2, 3, (short)threadIdx.x}, b4 = {2, 3, 4, (short)(threadIdx.x * 3)};
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    f32x16 d0 = {0}, d1 = {0};
    float e0 = threadIdx.x * 0.001f, e1 = e0 + 1, e2 = e0 + 2, e3 = e0 + 3;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) {   // 16x16x32, 4 independent accumulators
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
            } else if (MODE == 1) {   // 16x16x16 (the CDNA3 shape)
                c0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, c3, 0, 0, 0);
            } else if (MODE == 2) {   // 32x32x16
                d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d1, 0, 0, 0);
                d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d1, 0, 0, 0);
            } else if (MODE == 3) {   // 32x32x8 (the CDNA3 shape)
                d0 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, b4, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, b4, d1, 0, 0, 0);
                d0 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, b4, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, b4, d1, 0, 0, 0);
            } else if (MODE == 4) {   // 4 x v_exp_f32 alone
                e0 = __builtin_amdgcn_exp2f(e0); e1 = __builtin_amdgcn_exp2f(e1); e2 = __builtin_amdgcn_exp2f(e2); e3 = __builtin_amdgcn_exp2f(e3);
            } else if (MODE == 5) {   // 4 x (16x16x32 + v_exp)
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0); e0 = __builtin_amdgcn_exp2f(e0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0); e1 = __builtin_amdgcn_exp2f(e1);
                c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0); e2 = __builtin_amdgcn_exp2f(e2);
                c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0); e3 = __builtin_amdgcn_exp2f(e3);
            } else if (MODE == 6) {   // 4 x (16x16x32 + 2 v_exp)
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0); e0 = __builtin_amdgcn_exp2f(e0); e1 = __builtin_amdgcn_exp2f(e1);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0); e2 = __builtin_amdgcn_exp2f(e2); e3 = __builtin_amdgcn_exp2f(e3);
                c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0); e0 = __builtin_amdgcn_exp2f(e0); e1 = __builtin_amdgcn_exp2f(e1);
                c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0); e2 = __builtin_amdgcn_exp2f(e2); e3 = __builtin_amdgcn_exp2f(e3);
            } else if (MODE == 7) {   // 2 x (32x32x16 + 4 v_exp)
                d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d0, 0, 0, 0);
                e0 = __builtin_amdgcn_exp2f(e0); e1 = __builtin_amdgcn_exp2f(e1); e2 = __builtin_amdgcn_exp2f(e2); e3 = __builtin_amdgcn_exp2f(e3);
                d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d1, 0, 0, 0);
                e0 = __builtin_amdgcn_exp2f(e0); e1 = __builtin_amdgcn_exp2f(e1); e2 = __builtin_amdgcn_exp2f(e2); e3 = __builtin_amdgcn_exp2f(e3);
            } else if (MODE == 8) {   // 4 x v_fma_f32
                e0 = fmaf(e0, 1.0001f, 0.5f); e1 = fmaf(e1, 1.0001f, 0.5f); e2 = fmaf(e2, 1.0001f, 0.5f); e3 = fmaf(e3, 1.0001f, 0.5f);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    float s = c0[0] + c1[1] + c2[2] + c3[3] + d0[0] + d1[5] + e0 + e1 + e2 + e3;
    if (s == 12345.678f) sink[0] = s;
}

extern "C" int mfma_rate(int mode, unsigned long long* cyc, float* sink, int iters, int blocks, void* stream) {
    hipStream_t st = (hipStream_t)stream;
#define L(M) case M: hipLaunchKernelGGL(k_rate<M>, dim3(blocks), dim3(256), 0, st, cyc, sink, iters); break;
    switch (mode) { L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) default: return 1; }
    return (int)hipGetLastError();
}

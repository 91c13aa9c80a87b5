// Toolchain / ISA probe (not part of the product): confirms on a real gfx950 that
//  (1) a hipcc-7.2-built C-ABI .so launches kernels on torch's stream and memory,
//  (2) the MFMA 16x16x32 bf16 operand / accumulator lane maps used by the kernels,
//  (3) DPP row_ror all-reduce inside 16-lane rows, and cross-row shuffles.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

extern "C" __global__ void k_axpy(const float* x, float* y, float a, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = a * x[i] + y[i];
}

// A: [16][32] bf16 row-major (k contiguous), Bt: [16][32] bf16 (n rows, k contiguous), C: [16][16] f32
extern "C" __global__ void k_mfma_probe(const uint16_t* A, const uint16_t* Bt, float* C) {
    int l = threadIdx.x;
    int r = l & 15, g = l >> 4;
    bf16x8 a = *reinterpret_cast<const bf16x8*>(A + r * 32 + 8 * g);
    bf16x8 b = *reinterpret_cast<const bf16x8*>(Bt + r * 32 + 8 * g);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    for (int j = 0; j < 4; ++j) C[(4 * g + j) * 16 + r] = acc[j];
}

__device__ __forceinline__ float row_ror_max(float v) {
    // all-reduce max over each 16-lane row with DPP row_ror 8,4,2,1
    int x;
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, false); v = fmaxf(v, __int_as_float(x));
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xF, 0xF, false); v = fmaxf(v, __int_as_float(x));
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xF, 0xF, false); v = fmaxf(v, __int_as_float(x));
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xF, 0xF, false); v = fmaxf(v, __int_as_float(x));
    return v;
}

extern "C" __global__ void k_xlane_probe(const float* in, float* out_rowmax, float* out_x16, float* out_x32) {
    int l = threadIdx.x;
    float v = in[l];
    out_rowmax[l] = row_ror_max(v);
    out_x16[l] = __shfl_xor(v, 16);
    out_x32[l] = __shfl_xor(v, 32);
}

extern "C" int probe_axpy(const float* x, float* y, float a, int n, void* stream) {
    hipLaunchKernelGGL(k_axpy, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, y, a, n);
    return (int)hipGetLastError();
}
extern "C" int probe_mfma(const uint16_t* A, const uint16_t* Bt, float* C, void* stream) {
    hipLaunchKernelGGL(k_mfma_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, A, Bt, C);
    return (int)hipGetLastError();
}
extern "C" int probe_xlane(const float* in, float* a, float* b, float* c, void* stream) {
    hipLaunchKernelGGL(k_xlane_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, in, a, b, c);
    return (int)hipGetLastError();
}

// Micro-probe (not part of the product): per-CU L2 -> LDS staging rate of the forms a GEMM tile loader can
// take on gfx950, one workgroup per CU, every wave streaming "tiles" of PIECES x 1 KiB (8 rows x 128 B,
// row stride `row_stride` bytes) with DEPTH tiles in flight:
//   mode 0: global_load_lds_dwordx4 through the m0 save / restore statement the kernels use (dma16)
//   mode 1: global_load_lds_dwordx4, m0 written once per piece, no save / restore
//   mode 2: global_load_dwordx4 into VGPRs (asm, hand-counted vmcnt), ds_write_b128 when the tile has landed (register staging)
//   mode 3: buffer_load_dwordx4 ... offen lds (SRD + per-lane offset; m0 written per piece)
// out[wg*8 + 0..3] = {100 MHz ticks of the streaming loop (thread 0), ticks to issue the first tile, 0, 0}
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ void dma16_keep(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma16_raw(const void* gsrc, uint32_t lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory", "m0");
}
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16_buf(v4i srd, uint32_t voff, uint32_t lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %0, 0 offen lds" : : "s"(srd), "v"(voff), "s"(lds_dst) : "memory", "m0");
}

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int MODE, int PIECES, int DEPTH>
__global__ __launch_bounds__(512) void stream_kernel(const char* src, size_t src_bytes, int row_stride, int tiles, unsigned long long* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = blockDim.x >> 6;
    // per lane: row = lane>>3 (8 rows per piece), chunk = lane&7
    const size_t wg_base = ((size_t)blockIdx.x * 1048576) % (src_bytes / 2);
    const char* lane_src = src + wg_base + (size_t)(lane >> 3) * row_stride + (lane & 7) * 16;
    const uint32_t lds_wave = lds0 + (uint32_t)wave * (PIECES * 1024u);
    const uint32_t stage_bytes = (uint32_t)nw * PIECES * 1024u;
    v4i srd;
    {
        const uint64_t a = (uint64_t)(src + wg_base);
        srd[0] = (int)(uint32_t)a; srd[1] = (int)((uint32_t)(a >> 32) & 0xFFFFu); srd[2] = (int)0x7FFFFFFF; srd[3] = 0x00020000;
    }
    const uint32_t lane_off = (uint32_t)((lane >> 3) * row_stride + (lane & 7) * 16);
    v4i regs[DEPTH][PIECES];
    auto issue = [&](int t, int slot) {
        // tile t of this wave: PIECES pieces, each 8 rows further down; tiles advance along the row (k)
        const size_t toff = (size_t)(t & 15) * 128 + (size_t)wave * PIECES * 8 * row_stride;
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            const char* g = lane_src + toff + (size_t)j * 8 * row_stride;
            const uint32_t dst = lds_wave + (uint32_t)slot * stage_bytes + (uint32_t)j * 1024u;
            if (MODE == 0) dma16_keep(g, dst);
            else if (MODE == 1) dma16_raw(g, dst);
            else if (MODE == 3) dma16_buf(srd, lane_off + (uint32_t)toff + (uint32_t)(j * 8 * row_stride), dst);
            else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(regs[slot][j]) : "v"(g) : "memory");
        }
    };
    auto retire = [&](int slot) {   // mode 2: registers -> LDS
        if (MODE == 2) {
#pragma unroll
            for (int j = 0; j < PIECES; ++j)
                asm volatile("ds_write_b128 %0, %1" : : "v"(lds_wave + (uint32_t)slot * stage_bytes + (uint32_t)(j * 1024 + lane * 16)), "v"(regs[slot][j]) : "memory");
        }
    };
    __syncthreads();
    const long long c0 = wall_clock64();
    issue(0, 0);
    const long long c1 = wall_clock64();
#pragma unroll
    for (int s = 1; s < DEPTH; ++s) issue(s, s);
    for (int t = 0; t < tiles; t += DEPTH) {   // slot is a compile-time constant inside the unrolled body
#pragma unroll
        for (int slot = 0; slot < DEPTH; ++slot) {
            wait_vm<(DEPTH - 1) * PIECES>();
            retire(slot);
            __builtin_amdgcn_s_barrier();
            issue(t + slot + DEPTH, slot);
        }
    }
    wait_vm<0>();
    const long long c2 = wall_clock64();
    if (tid == 0) {
        out[blockIdx.x * 4 + 0] = (unsigned long long)(c2 - c0);
        out[blockIdx.x * 4 + 1] = (unsigned long long)(c1 - c0);
        out[blockIdx.x * 4 + 2] = ((const volatile unsigned*)smem)[17];
    }
}

#define LAUNCH(MODE, P, D)                                                                                          \
    if (mode == MODE && pieces == P && depth == D) {                                                               \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_kernel<MODE, P, D>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
        hipLaunchKernelGGL((stream_kernel<MODE, P, D>), dim3(grid), dim3(threads), lds, (hipStream_t)stream, (const char*)src, \
                           src_bytes, row_stride, tiles, out);                                                     \
        return (int)hipGetLastError();                                                                              \
    }

extern "C" int ldsdma_probe(const void* src, size_t src_bytes, int row_stride, int tiles, int mode, int pieces, int depth,
                            int threads, int grid, unsigned long long* out, void* stream) {
    const int lds = depth * (threads / 64) * pieces * 1024;
    LAUNCH(0, 4, 3) LAUNCH(1, 4, 3) LAUNCH(2, 4, 3) LAUNCH(3, 4, 3)
    LAUNCH(0, 4, 6) LAUNCH(1, 4, 6) LAUNCH(2, 4, 6) LAUNCH(3, 4, 6)
    LAUNCH(0, 2, 4) LAUNCH(1, 2, 4) LAUNCH(2, 2, 4) LAUNCH(3, 2, 4)
    return -1;
}

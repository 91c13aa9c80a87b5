// Shared pieces of the implicit-GEMM conv kernels (conv_gemm.hip, conv_halo.hip): argument block,
// epilogues, LDS-DMA helpers.
#pragma once
#include "common.h"

struct CGArgs {
    const bf16_t* a0; const bf16_t* a1; const bf16_t* w;
    const float* bias; const float* rowvec; const int32_t* step_ptr;
    const bf16_t* residual; void* out; bf16_t* out1; bf16_t* out2; float* ws;
    int batch, h_in, w_in, c0, c1, h_out, w_out, ksize, stride, pad, upsample;
    int M, N, K, hw_out, nkc, nk, nk_per, tiles_n;
    int tiles_m, m_fast;   // m_fast: consecutive tiles (= same XCD) share the WEIGHT rows instead of the pixel rows
    int act, out_f32, out_ld, res_ld, rv_step_stride, rv_batch_stride;
    int split_mode, ns0, ns1, out1_ld, out2_ld;
};

static __device__ __attribute__((aligned(128))) uint32_t g_zero_page[32];  // source of padding rows (one copy per TU)

// ---- epilogue for one group of 4 consecutive output columns of one row -----------------------
__device__ __forceinline__ void cg_store4(const CGArgs& p, int m, int b, int n, int step, float v[4]) {
    if (p.bias) {
        const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
        v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
    }
    if (p.rowvec) {
        const float4 rv = *reinterpret_cast<const float4*>(
            p.rowvec + (size_t)step * p.rv_step_stride + (size_t)b * p.rv_batch_stride + n);
        v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
    }
    if (p.act == MSD_ACT_SILU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
    } else if (p.act == MSD_ACT_QUICK_GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] * fast_rcp(1.0f + __expf(-1.702f * v[e]));
    }
    if (p.split_mode == 0) {
        if (p.residual) {
            const uint2 rr = *reinterpret_cast<const uint2*>(p.residual + (size_t)m * p.res_ld + n);
            v[0] += bf_lo(rr.x); v[1] += bf_hi(rr.x); v[2] += bf_lo(rr.y); v[3] += bf_hi(rr.y);
        }
        if (p.out_f32) {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.out_ld + n) =
                make_float4(v[0], v[1], v[2], v[3]);
        } else {
            uint2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]);
            *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.out_ld + n) = o;
        }
    } else {
        uint2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]);
        if (n < p.ns0) {
            *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.out_ld + n) = o;
        } else if (n < p.ns0 + p.ns1) {
            *reinterpret_cast<uint2*>(p.out1 + (size_t)m * p.out1_ld + (n - p.ns0)) = o;
        } else {
            const int nv = p.N - p.ns0 - p.ns1;
            const int nn = n - p.ns0 - p.ns1;
            const int s = m - b * p.hw_out;
            bf16_t* dst = p.out2 + ((size_t)b * nv + nn) * p.out2_ld + s;
            dst[0] = (bf16_t)(o.x & 0xFFFF);
            dst[(size_t)p.out2_ld] = (bf16_t)(o.x >> 16);
            dst[(size_t)2 * p.out2_ld] = (bf16_t)(o.y & 0xFFFF);
            dst[(size_t)3 * p.out2_ld] = (bf16_t)(o.y >> 16);
        }
    }
}

// Whole-wave epilogue: lane (r = lane&15, g = lane>>4) holds, per (j, i), output channels
// n..n+3 (n = nbase + 16j + 4g) of pixel m (= mbase + 16i + r).
// Rows: fragment i of the wave covers pixels mrow[i] + r (r = 0..15), so a spatially blocked tile
// (conv_halo) and a linear one (conv_gemm: mrow[i] = mbase + 16 i) share the epilogue.
template <int MI, int NJ>
__device__ __forceinline__ void cg_epilogue(const CGArgs& p, f32x4 (&acc)[NJ][MI], const int (&mrow)[MI], int nbase, int r, int g) {
    if (gridDim.y > 1) {
        float* ws = p.ws + (size_t)blockIdx.y * p.M * p.N;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = mrow[i] + r;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = nbase + j * 16 + 4 * g;
                if (n >= p.N) continue;
                *reinterpret_cast<float4*>(ws + (size_t)m * p.N + n) =
                    make_float4(acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]);
            }
        }
        return;
    }
    const int step = p.step_ptr ? *p.step_ptr : 0;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = mrow[i] + r;
        if (m >= p.M) continue;
        const int b = m / p.hw_out;
        if (p.act == MSD_ACT_GEGLU) {
            if constexpr (NJ % 2 == 0) {   // (x|gate pairs: the host never sends GEGLU to an odd-NJ tile)
#pragma unroll
            for (int j = 0; j < NJ; j += 2) {
                const int nb = nbase + j * 16;  // multiple of 32: x columns nb+[0,16), gate nb+16+[0,16)
                const int n = nb + 4 * g;
                if (n >= p.N) continue;
                float v[4];
                float4 bx = make_float4(0, 0, 0, 0), bg = bx;
                if (p.bias) {
                    bx = *reinterpret_cast<const float4*>(p.bias + n);
                    bg = *reinterpret_cast<const float4*>(p.bias + n + 16);
                }
                v[0] = geglu_f(acc[j][i][0] + bx.x, acc[j + 1][i][0] + bg.x);
                v[1] = geglu_f(acc[j][i][1] + bx.y, acc[j + 1][i][1] + bg.y);
                v[2] = geglu_f(acc[j][i][2] + bx.z, acc[j + 1][i][2] + bg.z);
                v[3] = geglu_f(acc[j][i][3] + bx.w, acc[j + 1][i][3] + bg.w);
                const int no = (nb >> 1) + 4 * g;
                if (p.residual) {
                    const uint2 rr = *reinterpret_cast<const uint2*>(p.residual + (size_t)m * p.res_ld + no);
                    v[0] += bf_lo(rr.x); v[1] += bf_hi(rr.x); v[2] += bf_lo(rr.y); v[3] += bf_hi(rr.y);
                }
                uint2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]);
                *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.out_ld + no) = o;
            }
            }
        } else {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = nbase + j * 16 + 4 * g;
                if (n >= p.N) continue;
                float v[4] = {acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]};
                cg_store4(p, m, b, n, step, v);
            }
        }
    }
}

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


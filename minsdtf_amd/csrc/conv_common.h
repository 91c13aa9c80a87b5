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
    int nslices;           // split-K slices (= gridDim.y; kept in the argument block so that no kernel reads the implicit
                           // grid-size arguments: that is a second, dependent scalar-memory round trip at kernel start)
    int tiles_m, m_fast;   // m_fast: consecutive tiles (= same XCD) share the WEIGHT rows instead of the pixel rows
    int act, out_f32, out_ld, res_ld, rv_step_stride, rv_batch_stride;
    int split_mode, ns0, ns1, out1_ld, out2_ld;
    uint32_t mg_tdiv, mg_hw, mg_w, mg_tps, mg_tx, mg_nkc;   // floor(2^32 / d) for d = tile-mapping divisor, hw_out, w_out, tiles per sample, tiles per row (udiv_magic)
    const bf16_t* a2; const bf16_t* a3; int c2, nk_main;         // shortcut operand: K tiles >= nk_main read a2|a3 at the output pixel
    const float* ln_in; const float* ln_colsum; float* ln_out;   // LayerNorm fold (minsdtf_hip.h)
    int ln_in_slots, ln_out_slots;
    float ln_eps, ln_inv_k;
};

static __device__ __attribute__((aligned(128))) uint32_t g_zero_page[32];  // source of padding rows (one copy per TU)


// In-kernel timeline stamps (tools/gemm_stamps.py; `make stamps` builds a separate instrumented
// library, the product library never carries them): thread 0 of every workgroup records the 100 MHz
// wall clock at a few points of conv_gemm_dma_kernel and its epilogue.
#ifdef MSD_STAMPS
static __device__ unsigned long long g_stamps[16 * 8192];
#define MSD_STAMP(i)                                                                                             \
    do {                                                                                                         \
        if (threadIdx.x == 0) g_stamps[(((size_t)blockIdx.y * gridDim.x + blockIdx.x) & 8191) * 16 + (i)] = wall_clock64(); \
    } while (0)
#define MSD_STAMP_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define MSD_STAMP(i)
#define MSD_STAMP_DRAIN()
#endif

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
constexpr int LN_MAX_SLOTS = 20;   // row-moment partials per row (column tiles of the producing launch)
// `lnred` (with ln_out): LDS scratch of WGN x BM float2, free for reuse (the caller has passed a barrier after
// its last fragment read); `wn` / `wgn`: this wave's column slab and the number of slabs; `row0`: first row
// of the wave's tile inside the workgroup tile; `tile_n`: column tile index = the partial's slot.
// UB: every row of the workgroup's tile belongs to ONE sample (the spatial tiles of conv_halo), so the time-embedding
// row is loaded once per column group instead of per (row fragment, column group): 48 fewer live registers on the
// 64x64-per-wave tile, which otherwise spills.
template <int MI, int NJ, bool UB = false>
__device__ __forceinline__ void cg_epilogue(const CGArgs& p, f32x4 (&acc)[NJ][MI], const int (&mrow)[MI], int nbase, int r, int g,
                                            float* lnred = nullptr, int wn = 0, int wgn = 1, int row0 = 0, int bm = 0, int tile_n = 0) {
    if (p.nslices > 1) {
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
    // Two passes: every load of the epilogue first, then the arithmetic and the stores.  `out` may alias
    // `residual` as far as the compiler knows, so in a single pass each group's loads wait behind the
    // previous group's store: MI x NJ serial memory round trips (in-kernel stamps: 4.4 us of a 9 us launch
    // for the 128x128 tile).  Batched, the epilogue is one load round trip plus the stores.
    // The loads are UNCONDITIONAL per element (addresses clamped into the tensor; one wave-uniform branch
    // around each whole batch) and end in one explicit vmcnt(0): a per-element "load or zero" select makes
    // hipcc branch around every load and, at the control-flow joins, wait vmcnt(0) before every group —
    // which on gfx9 also waits for the previous group's STORE (cdna_hip_programming.md §5, trap (c)).
    float4 bv[NJ];
    if (p.bias) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) bv[j] = *reinterpret_cast<const float4*>(p.bias + min(nbase + j * 16 + 4 * g, p.N - 4));
    } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j) bv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // LayerNorm fold, consumer side: partial row moments (lane group g takes slots g, g+4, ...) and column sums
    constexpr int LNS = LN_MAX_SLOTS / 4;
    float2 lnp[MI][LNS];
    float4 lcs[NJ];
    if (p.ln_in) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const float2* src = reinterpret_cast<const float2*>(p.ln_in) + (size_t)min(mrow[i] + r, p.M - 1) * p.ln_in_slots;
#pragma unroll
            for (int k = 0; k < LNS; ++k) lnp[i][k] = src[min(g + 4 * k, p.ln_in_slots - 1)];   // (clamped: masked below)
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) lcs[j] = *reinterpret_cast<const float4*>(p.ln_colsum + min(nbase + j * 16 + 4 * g, p.N - 4));
    }
    auto ln_apply = [&]() {   // after the vmcnt(0): acc <- rstd * (acc - mean * colsum)
        if (!p.ln_in) return;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int k = 0; k < LNS; ++k)
                if (g + 4 * k < p.ln_in_slots) { s1 += lnp[i][k].x; s2 += lnp[i][k].y; }
            s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);   // fixed order: (g0+g1) + (g2+g3) on every lane
            s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
            const float mean = s1 * p.ln_inv_k;
            const float rstd = rsqrtf(fmaxf(s2 * p.ln_inv_k - mean * mean, 0.f) + p.ln_eps);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                acc[j][i][0] = rstd * (acc[j][i][0] - mean * lcs[j].x);
                acc[j][i][1] = rstd * (acc[j][i][1] - mean * lcs[j].y);
                acc[j][i][2] = rstd * (acc[j][i][2] - mean * lcs[j].z);
                acc[j][i][3] = rstd * (acc[j][i][3] - mean * lcs[j].w);
            }
        }
    };
    if (p.act == MSD_ACT_GEGLU) {
        if constexpr (NJ % 2 == 0) {   // (x|gate pairs: the host never sends GEGLU to an odd-NJ tile)
            uint2 rr[MI][NJ / 2];
            if (p.residual) {
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const size_t mo = (size_t)min(mrow[i] + r, p.M - 1) * p.res_ld;
#pragma unroll
                    for (int j = 0; j < NJ; j += 2)
                        rr[i][j / 2] = *reinterpret_cast<const uint2*>(p.residual + mo + min(((nbase + j * 16) >> 1) + 4 * g, (p.N >> 1) - 4));
                }
            } else {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; j += 2) rr[i][j / 2] = make_uint2(0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): every epilogue load has landed
            ln_apply();
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m = mrow[i] + r;
                if (m >= p.M) continue;
#pragma unroll
                for (int j = 0; j < NJ; j += 2) {
                    const int nb = nbase + j * 16;  // multiple of 32: x columns nb+[0,16), gate nb+16+[0,16)
                    if (nb + 4 * g >= p.N) continue;
                    float v[4];
                    v[0] = geglu_f(acc[j][i][0] + bv[j].x, acc[j + 1][i][0] + bv[j + 1].x);
                    v[1] = geglu_f(acc[j][i][1] + bv[j].y, acc[j + 1][i][1] + bv[j + 1].y);
                    v[2] = geglu_f(acc[j][i][2] + bv[j].z, acc[j + 1][i][2] + bv[j + 1].z);
                    v[3] = geglu_f(acc[j][i][3] + bv[j].w, acc[j + 1][i][3] + bv[j + 1].w);
                    const int no = (nb >> 1) + 4 * g;
                    const uint2 q = rr[i][j / 2];
                    v[0] += bf_lo(q.x); v[1] += bf_hi(q.x); v[2] += bf_lo(q.y); v[3] += bf_hi(q.y);
                    uint2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]);
                    *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.out_ld + no) = o;
                }
            }
        }
        return;
    }
    const bool plain_res = p.split_mode == 0 && p.residual != nullptr;
    uint2 rr[MI][NJ];
    constexpr int RI = UB ? 1 : MI;
    float4 rv[RI][NJ];
    int bidx[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) bidx[i] = (p.rowvec || p.split_mode) ? udiv_magic(min(mrow[i] + r, p.M - 1), p.hw_out, p.mg_hw) : 0;
    if (plain_res) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const size_t mo = (size_t)min(mrow[i] + r, p.M - 1) * p.res_ld;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                rr[i][j] = *reinterpret_cast<const uint2*>(p.residual + mo + min(nbase + j * 16 + 4 * g, p.N - 4));
        }
    } else {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) rr[i][j] = make_uint2(0, 0);
    }
    if (p.rowvec) {
#pragma unroll
        for (int i = 0; i < RI; ++i) {
            const float* rvp = p.rowvec + (size_t)step * p.rv_step_stride + (size_t)bidx[i] * p.rv_batch_stride;
#pragma unroll
            for (int j = 0; j < NJ; ++j) rv[i][j] = *reinterpret_cast<const float4*>(rvp + min(nbase + j * 16 + 4 * g, p.N - 4));
        }
    } else {
#pragma unroll
        for (int i = 0; i < RI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) rv[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): every epilogue load has landed
    ln_apply();
    float ls1[MI], ls2[MI];   // ln_out: this wave's partial row moments of the values it stores
#pragma unroll
    for (int i = 0; i < MI; ++i) { ls1[i] = 0.f; ls2[i] = 0.f; }
    MSD_STAMP_DRAIN();
    MSD_STAMP(6);
    // The common case (plain mode, bf16 output, no activation: every residual / LayerNorm-producer / shortcut GEMM and
    // the second conv of a ResBlock) takes a compact straight-line path of its own instead of threading through the
    // branches of every other mode (in-kernel stamps: arithmetic + store issue 0.72 -> 0.48 us on the 64x64 tile, 1.92 ->
    // 1.68 us on 128x128; a kernel starts with a cold instruction cache and the generic loop touches several times
    // more instruction lines than this case executes).
    if (p.split_mode == 0 && !p.out_f32 && p.act == MSD_ACT_NONE) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = mrow[i] + r;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = nbase + j * 16 + 4 * g;
                if (m >= p.M || n >= p.N) continue;
                const float v0 = acc[j][i][0] + bv[j].x + rv[UB ? 0 : i][j].x + bf_lo(rr[i][j].x);
                const float v1 = acc[j][i][1] + bv[j].y + rv[UB ? 0 : i][j].y + bf_hi(rr[i][j].x);
                const float v2 = acc[j][i][2] + bv[j].z + rv[UB ? 0 : i][j].z + bf_lo(rr[i][j].y);
                const float v3 = acc[j][i][3] + bv[j].w + rv[UB ? 0 : i][j].w + bf_hi(rr[i][j].y);
                uint2 o; o.x = pack_bf2(v0, v1); o.y = pack_bf2(v2, v3);
                *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.out_ld + n) = o;
                if (p.ln_out) {   // moments of the ROUNDED values: what a LayerNorm reading `out` would see
                    const float q0 = bf_lo(o.x), q1 = bf_hi(o.x), q2 = bf_lo(o.y), q3 = bf_hi(o.y);
                    ls1[i] += (q0 + q1) + (q2 + q3);
                    ls2[i] += (q0 * q0 + q1 * q1) + (q2 * q2 + q3 * q3);
                }
            }
        }
    } else
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = mrow[i] + r;
        if (m >= p.M) continue;
        const int b = bidx[i];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = nbase + j * 16 + 4 * g;
            if (n >= p.N) continue;
            float v[4] = {acc[j][i][0] + bv[j].x + rv[UB ? 0 : i][j].x, acc[j][i][1] + bv[j].y + rv[UB ? 0 : i][j].y,
                          acc[j][i][2] + bv[j].z + rv[UB ? 0 : i][j].z, acc[j][i][3] + bv[j].w + rv[UB ? 0 : i][j].w};
            if (p.act == MSD_ACT_SILU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
            } else if (p.act == MSD_ACT_QUICK_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] * fast_rcp(1.0f + __expf(-1.702f * v[e]));
            }
            if (p.split_mode == 0) {
                v[0] += bf_lo(rr[i][j].x); v[1] += bf_hi(rr[i][j].x); v[2] += bf_lo(rr[i][j].y); v[3] += bf_hi(rr[i][j].y);
                if (p.out_f32) {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.out_ld + n) =
                        make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    uint2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]);
                    *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.out_ld + n) = o;
                    if (p.ln_out) {   // moments of the ROUNDED values: what a LayerNorm reading `out` would see
                        const float q0 = bf_lo(o.x), q1 = bf_hi(o.x), q2 = bf_lo(o.y), q3 = bf_hi(o.y);
                        ls1[i] += (q0 + q1) + (q2 + q3);
                        ls2[i] += (q0 * q0 + q1 * q1) + (q2 * q2 + q3 * q3);
                    }
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
                    const int sidx = m - b * p.hw_out;
                    bf16_t* dst = p.out2 + ((size_t)b * nv + nn) * p.out2_ld + sidx;
                    dst[0] = (bf16_t)(o.x & 0xFFFF);
                    dst[(size_t)p.out2_ld] = (bf16_t)(o.x >> 16);
                    dst[(size_t)2 * p.out2_ld] = (bf16_t)(o.y & 0xFFFF);
                    dst[(size_t)3 * p.out2_ld] = (bf16_t)(o.y >> 16);
                }
            }
        }
    }
    MSD_STAMP(7);
    if (p.ln_out) {
        __builtin_amdgcn_s_barrier();   // every wave is past its last fragment read: the ring can be reused
        // lanes of one row (g = 0..3) -> wave partial; the WGN column slabs of the workgroup are summed in
        // slab order by the wn == 0 wave through LDS -> one (sum, sumsq) per row and column tile
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            ls1[i] += __shfl_xor(ls1[i], 16); ls2[i] += __shfl_xor(ls2[i], 16);
            ls1[i] += __shfl_xor(ls1[i], 32); ls2[i] += __shfl_xor(ls2[i], 32);
            if (g == 0) {
                lnred[((size_t)wn * bm + row0 + i * 16 + r) * 2 + 0] = ls1[i];
                lnred[((size_t)wn * bm + row0 + i * 16 + r) * 2 + 1] = ls2[i];
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the partials are in LDS (raw barrier: the stores stay in flight)
        __builtin_amdgcn_s_barrier();
        if (wn == 0 && g == 0) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m = mrow[i] + r;
                if (m >= p.M) continue;
                float a = 0.f, q = 0.f;
                for (int w = 0; w < wgn; ++w) {
                    a += lnred[((size_t)w * bm + row0 + i * 16 + r) * 2 + 0];
                    q += lnred[((size_t)w * bm + row0 + i * 16 + r) * 2 + 1];
                }
                reinterpret_cast<float2*>(p.ln_out)[(size_t)m * p.ln_out_slots + tile_n] = make_float2(a, q);
            }
        }
    }
}

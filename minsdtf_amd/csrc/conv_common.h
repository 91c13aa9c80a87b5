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
    int vec16;             // rows of out / out1 / residual are 16-byte aligned (and split parts multiples of 16 columns): 16-byte epilogue form
    int split_mode, ns0, ns1, out1_ld, out2_ld;
    uint32_t mg_tdiv, mg_hw, mg_w, mg_tps, mg_tx, mg_nkc;   // floor(2^32 / d) for d = tile-mapping divisor, hw_out, w_out, tiles per sample, tiles per row (udiv_magic)
    const bf16_t* a2; const bf16_t* a3; int c2, nk_main;         // shortcut operand: K tiles >= nk_main read a2|a3 at the output pixel
    const float* ln_in; const float* ln_colsum; float* ln_out;   // LayerNorm fold (minsdtf_hip.h)
    int ln_in_slots, ln_out_slots;
    float ln_eps, ln_inv_k;
    int kmajor;            // conv_big.hip: 1 = walk K chunk-major (for every 64-channel chunk its 9 taps: the halo-tile kernel's order, so its bits), 0 = tap-major
    uint32_t w_rs, w_ks;   // weight addressing in bytes: row (output column) stride, K-chunk stride ([N][K]: 2K, 128; chunk-major: 128, 128 N)
};

// Kernarg preload: the kernels take the fields their prologue needs FIRST as leading scalar arguments (15 dwords) in front of the
// argument block; built with -mllvm -amdgpu-kernarg-preload-count=16 these arrive in SGPRs with the wave, so tile mapping and loader
// coordinates start without waiting for the first scalar-memory round trip (a struct argument is never preloaded).  CG_HOT_ARGS(a) is
// the launch side, CG_HOT_PARAMS the kernel side (the prologue reads hot_* where it would read p.*; the same values).
#define CG_HOT_PARAMS const bf16_t* hot_a0, const bf16_t* hot_w, const bf16_t* hot_a1, int hot_M, int hot_N, uint32_t hot_pk_c, uint32_t hot_pk_tiles, \
                      uint32_t hot_pk_nk, uint32_t hot_mg_tdiv, uint32_t hot_w_rs, uint32_t hot_w_ks
// 14 dwords = all the preload the hardware offers (16 user SGPRs less the kernarg pointer), so three pairs travel packed; cg_hot_ok()
// is the host-side range check.  CG_HOT_UNPACK declares the fields the prologues read.
#define CG_HOT_ARGS(a) (a).a0, (a).w, (a).a1, (a).M, (a).N, ((uint32_t)(a).c0 | ((uint32_t)(a).c1 << 16)),                           \
                       ((uint32_t)(a).tiles_m | ((uint32_t)(a).tiles_n << 23) | ((uint32_t)((a).m_fast ? 1u : 0u) << 31)),          \
                       ((uint32_t)(a).nk_per | ((uint32_t)(a).nk << 16)), (a).mg_tdiv, (a).w_rs, (a).w_ks
#define CG_HOT_UNPACK                                                                                                          \
    const int hot_c0 = (int)(hot_pk_c & 0xFFFFu), hot_c1 = (int)(hot_pk_c >> 16);                                             \
    const int hot_tiles_m = (int)(hot_pk_tiles & 0x7FFFFFu), hot_tiles_n = (int)((hot_pk_tiles >> 23) & 0xFFu);                \
    const int hot_m_fast = (int)(hot_pk_tiles >> 31);                                                                          \
    const int hot_nk_per = (int)(hot_pk_nk & 0xFFFFu), hot_nk = (int)(hot_pk_nk >> 16);                                        \
    (void)hot_c1; (void)hot_a1; (void)hot_w_rs; (void)hot_w_ks; (void)hot_nk
static inline bool cg_hot_ok(const CGArgs& a) {
    return a.c0 >= 0 && a.c0 < 65536 && a.c1 >= 0 && a.c1 < 65536 && a.tiles_m > 0 && a.tiles_m < (1 << 23) && a.tiles_n > 0 && a.tiles_n < 256 &&
           a.nk_per > 0 && a.nk_per < 65536 && a.nk > 0 && a.nk < 65536;
}

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

// The same with the operands already in registers (splitk_finalize_kernel issues their loads together with the slab loads); an absent
// operand arrives as zeros with its flag off.  Same expressions in the same order as cg_store4: the same bits.
__device__ __forceinline__ void cg_store4_pre(const CGArgs& p, int m, int b, int n, float v[4], float4 bv, bool has_b, float4 rv, bool has_rv,
                                              uint2 rr, bool has_r) {
    if (has_b) { v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w; }
    if (has_rv) { v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w; }
    if (p.act == MSD_ACT_SILU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
    } else if (p.act == MSD_ACT_QUICK_GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] * fast_rcp(1.0f + __expf(-1.702f * v[e]));
    }
    if (p.split_mode == 0) {
        if (has_r) { v[0] += bf_lo(rr.x); v[1] += bf_hi(rr.x); v[2] += bf_lo(rr.y); v[3] += bf_hi(rr.y); }
        if (p.out_f32) {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.out_ld + n) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            uint2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]);
            *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.out_ld + n) = o;
        }
        return;
    }
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

// Whole-wave epilogue.  Fragment map: the weight rows of a 16-column block are read into the MFMA in the order
// 0-3, 8-11, 4-7, 12-15 (cg_wrow below), so lane (r = lane & 15, g = lane >> 4) holds, per (j, i), the 4 output channels
// n .. n+3, n = nbase + 16 j + cg_col(g), cg_col = {0, 8, 4, 12}, of pixel m = mrow[i] + r.  Lanes l and l + 32 then hold
// the two halves of one 8-channel run, and ONE v_permlane32_swap per register turns the accumulators of two row
// fragments (i, i + 1) into 8 consecutive channels of one pixel per lane — the lower half-wave for fragment i, the upper
// one for i + 1 — which are stored (and whose residual is loaded) with 16-byte instead of 8-byte vectors: half the
// store / load instructions of the epilogue, whose tail is bound by store ISSUE, not bandwidth (in-kernel stamps;
// cdna_hip_programming.md T21).  Which lane holds which channel does not change any value.
// Rows: fragment i of the wave covers pixels mrow[i] + r (r = 0..15), so a spatially blocked tile
// (conv_halo) and a linear one (conv_gemm: mrow[i] = mbase + 16 i) share the epilogue.
constexpr int LN_MAX_SLOTS = 20;   // row-moment partials per row (column tiles of the producing launch)
__device__ __forceinline__ int cg_wrow(int r) { return (r & 3) | ((r & 4) << 1) | ((r & 8) >> 1); }   // MFMA row -> weight row of the block
__device__ __forceinline__ int cg_col(int g) { return ((g & 1) << 3) | ((g & 2) << 1); }
// `lnred` (with ln_out): LDS scratch of WGN x BM float2, free for reuse (the caller has passed a barrier after
// its last fragment read); `wn` / `wgn`: this wave's column slab and the number of slabs; `row0`: first row
// of the wave's tile inside the workgroup tile; `tile_n`: column tile index = the partial's slot.
// UB: every row of the workgroup's tile belongs to ONE sample (the spatial tiles of conv_halo), so the time-embedding
// row is loaded once per column group instead of per (row fragment, column group): 48 fewer live registers on the
// 64x64-per-wave tile, which otherwise spills.
// DF (dense form): the 1x1 / Dense kernel's epilogue carries the LayerNorm-fold consumer (ln_in), the GEGLU gate and the
// q | k | v^T split — none of which a 3x3 / strided / upsampling conv ever has — and no time-embedding row (rowvec: only
// ever the 3x3 conv1 of a ResBlock); the general and halo kernels' epilogue (DF = false) is the reverse.  Neither form
// pays registers or instruction-cache lines for the other's cases (the host routes each launch to the form it needs).
template <int MI, int NJ, bool UB = false, bool DF = true>
__device__ __forceinline__ void cg_epilogue(const CGArgs& p, f32x4 (&acc)[NJ][MI], const int (&mrow)[MI], int nbase, int r, int g,
                                            float* lnred = nullptr, int wn = 0, int wgn = 1, int row0 = 0, int bm = 0, int tile_n = 0) {
    const int cgo = cg_col(g);
    if (p.nslices > 1) {
        float* ws = p.ws + (size_t)blockIdx.y * p.M * p.N;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = mrow[i] + r;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = nbase + j * 16 + cgo;
                if (n >= p.N) continue;
                *reinterpret_cast<float4*>(ws + (size_t)m * p.N + n) =
                    make_float4(acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]);
            }
        }
        return;
    }
    const int step = p.step_ptr ? *p.step_ptr : 0;
    // 16-byte form: needs fragment pairs and 16-byte aligned rows (host: p.vec16); the lane's row / 8-channel run after the swap
    constexpr int MP = MI / 2;
    const bool v16 = (MI % 2 == 0) && p.vec16;
    const int cg8 = (g & 1) << 3;
    int msw[MP > 0 ? MP : 1];
#pragma unroll
    for (int ip = 0; ip < MP; ++ip) msw[ip] = (g < 2 ? mrow[2 * ip] : mrow[2 * ip + 1]) + r;
    auto swap8 = [&](const float (&a)[4], const float (&b)[4], float (&v)[8]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[e]), __float_as_uint(b[e]), false, false);
            v[e] = __uint_as_float(sw[0]);       // lower half-wave: own a[e]; upper: b[e] of lane - 32 (channels cgo - 4 + e)
            v[4 + e] = __uint_as_float(sw[1]);   // lower: a[e] of lane + 32 (channels cgo + 4 + e); upper: own b[e]
        }
    };
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
        for (int j = 0; j < NJ; ++j) bv[j] = *reinterpret_cast<const float4*>(p.bias + min(nbase + j * 16 + cgo, p.N - 4));
    } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j) bv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // LayerNorm fold, consumer side: partial row moments (lane group g takes slots g, g+4, ...) and column sums
    constexpr int LNS = LN_MAX_SLOTS / 4;
    constexpr bool LN = DF, RV = !DF;
    float2 lnp[LN ? MI : 1][LNS];
    float4 lcs[NJ];
    if (LN && p.ln_in) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const float2* src = reinterpret_cast<const float2*>(p.ln_in) + (size_t)min(mrow[i] + r, p.M - 1) * p.ln_in_slots;
#pragma unroll
            for (int k = 0; k < LNS; ++k) lnp[i][k] = src[min(g + 4 * k, p.ln_in_slots - 1)];   // (clamped: masked below)
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) lcs[j] = *reinterpret_cast<const float4*>(p.ln_colsum + min(nbase + j * 16 + cgo, p.N - 4));
    }
    auto ln_apply = [&]() {   // after the vmcnt(0): acc <- rstd * (acc - mean * colsum)
        if (!LN || !p.ln_in) return;
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
    if (DF && p.act == MSD_ACT_GEGLU) {
        if constexpr (DF && NJ % 2 == 0) {   // (x|gate pairs: the host never sends GEGLU to an odd-NJ tile)
            // (GEGLU launches carry no residual in this pipeline; one is still honoured on the 8-byte path)
            const bool g16 = v16 && !p.residual;
            uint2 rr[MI][NJ / 2];
            if (p.residual) {
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const size_t mo = (size_t)min(mrow[i] + r, p.M - 1) * p.res_ld;
#pragma unroll
                    for (int j = 0; j < NJ; j += 2)
                        rr[i][j / 2] = *reinterpret_cast<const uint2*>(p.residual + mo + min(((nbase + j * 16) >> 1) + cgo, (p.N >> 1) - 4));
                }
            } else {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; j += 2) rr[i][j / 2] = make_uint2(0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): every epilogue load has landed
            ln_apply();
            auto gl4 = [&](int i, int j, float (&v)[4]) {   // x columns nb + [0,16), gate nb + 16 + [0,16): same channel map in both
                v[0] = geglu_f(acc[j][i][0] + bv[j].x, acc[j + 1][i][0] + bv[j + 1].x);
                v[1] = geglu_f(acc[j][i][1] + bv[j].y, acc[j + 1][i][1] + bv[j + 1].y);
                v[2] = geglu_f(acc[j][i][2] + bv[j].z, acc[j + 1][i][2] + bv[j + 1].z);
                v[3] = geglu_f(acc[j][i][3] + bv[j].w, acc[j + 1][i][3] + bv[j + 1].w);
            };
            if (g16) {
#pragma unroll
                for (int ip = 0; ip < MP; ++ip)
#pragma unroll
                    for (int j = 0; j < NJ; j += 2) {
                        float a[4], b[4], v[8];
                        gl4(2 * ip, j, a);
                        gl4(2 * ip + 1, j, b);
                        swap8(a, b, v);
                        const int no = ((nbase + j * 16) >> 1) + cg8;
                        if (msw[ip] >= p.M || no >= (p.N >> 1)) continue;
                        bf16_t* dst = reinterpret_cast<bf16_t*>(p.out) + (size_t)msw[ip] * p.out_ld + no;
                        if (no + 8 <= (p.N >> 1)) *reinterpret_cast<uint4*>(dst) = pack8(v);
                        else *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                    }
            } else {
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int m = mrow[i] + r;
                    if (m >= p.M) continue;
#pragma unroll
                    for (int j = 0; j < NJ; j += 2) {
                        const int nb = nbase + j * 16;  // multiple of 32
                        if (nb + cgo >= p.N) continue;
                        float v[4];
                        gl4(i, j, v);
                        const int no = (nb >> 1) + cgo;
                        const uint2 q = rr[i][j / 2];
                        v[0] += bf_lo(q.x); v[1] += bf_hi(q.x); v[2] += bf_lo(q.y); v[3] += bf_hi(q.y);
                        uint2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]);
                        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.out_ld + no) = o;
                    }
                }
            }
        }
        return;
    }
    const int split_mode = DF ? p.split_mode : 0;
    const bool plain_res = split_mode == 0 && p.residual != nullptr;
    const bool compact = split_mode == 0 && !p.out_f32 && p.act == MSD_ACT_NONE;
    const bool c16 = v16 && compact;                                  // plain bf16 output, 16-byte stores
    const bool s16 = v16 && split_mode != 0 && p.act == MSD_ACT_NONE;   // q | k parts of the split epilogue (ns0, ns1 multiples of 16: host)
    // Time-embedding row (one per SAMPLE): when all rows of the wave's fragments lie in one sample — always for the spatial
    // tiles (UB) and for every linear tile except a 128-row one at the 8x8 level — it is ONE float4 per column group,
    // loaded with the batch below; otherwise (rare) the fragments add theirs one at a time after the batch (rv_slow).
    // (A per-fragment array costs 64 registers on the 64x64-per-wave tiles, which sit at the register limit.)
    int bidx[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) bidx[i] = ((RV && p.rowvec) || split_mode) ? udiv_magic(min(mrow[i] + r, p.M - 1), p.hw_out, p.mg_hw) : 0;
    const int b_first = UB ? bidx[0] : udiv_magic(min(mrow[0], p.M - 1), p.hw_out, p.mg_hw);
    const int b_last = UB ? bidx[0] : udiv_magic(min(mrow[MI - 1] + 15, p.M - 1), p.hw_out, p.mg_hw);
    const bool rv_one = RV && p.rowvec && b_first == b_last, rv_slow = RV && p.rowvec && !rv_one;   // (wave-uniform)
    float4 rv[NJ];
    if (rv_one) {
        const float* rvp = p.rowvec + (size_t)step * p.rv_step_stride + (size_t)b_first * p.rv_batch_stride;
#pragma unroll
        for (int j = 0; j < NJ; ++j) rv[j] = *reinterpret_cast<const float4*>(rvp + min(nbase + j * 16 + cgo, p.N - 4));
    } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j) rv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    MSD_STAMP_DRAIN();
    MSD_STAMP(6);
    // ln_out (LayerNorm fold, producer side): moments of the ROUNDED values this launch stores, per row and column tile.
    // Canonical order, independent of the wave layout and of the 8- / 16-byte form (a sample's bits must not depend on
    // launch parameters that are tuned per batch): per 16-column block (s[0..3] + s[4..7]) + (s[8..11] + s[12..15]) with
    // s[c..c+3] = (q0 + q1) + (q2 + q3), written to LDS per (row, block); then the blocks of the column tile in ascending
    // order (below).  The cross-lane adds are row swaps (VALU), not LDS shuffles.
    const int nb16 = wgn * NJ;   // 16-column blocks of the workgroup's column tile
    if (p.ln_out) __builtin_amdgcn_s_barrier();   // every wave is past its last fragment read: the ring can be reused
    auto xadd16 = [](float x) {
        auto t = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
        return __uint_as_float(t[0]) + __uint_as_float(t[1]);
    };
    auto xadd32 = [](float x) {
        auto t = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
        return __uint_as_float(t[0]) + __uint_as_float(t[1]);
    };
    auto ln_put = [&](int i, int j, float s1, float s2) {   // s1, s2: the block's sums, complete on the calling lane
        float* dst = lnred + ((size_t)(row0 + i * 16 + r) * nb16 + wn * NJ + j) * 2;
        dst[0] = s1; dst[1] = s2;
    };
    // bias + time-embedding row go INTO the accumulators as soon as the loads have landed (add_bias_rowvec, called after
    // the vmcnt(0) of either form): their registers (up to 80 on the 64x64-per-wave tiles) are dead before the swaps,
    // packs and stores begin
    auto add_bias_rowvec = [&]() {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                acc[j][i][0] = (acc[j][i][0] + bv[j].x) + rv[j].x; acc[j][i][1] = (acc[j][i][1] + bv[j].y) + rv[j].y;
                acc[j][i][2] = (acc[j][i][2] + bv[j].z) + rv[j].z; acc[j][i][3] = (acc[j][i][3] + bv[j].w) + rv[j].w;
            }
        if (rv_slow) {   // rows of the wave in more than one sample: per-fragment rows, one round trip each
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const float* rvp = p.rowvec + (size_t)step * p.rv_step_stride + (size_t)bidx[i] * p.rv_batch_stride;
                float4 t[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) t[j] = *reinterpret_cast<const float4*>(rvp + min(nbase + j * 16 + cgo, p.N - 4));
                __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
                for (int j = 0; j < NJ; ++j) { acc[j][i][0] += t[j].x; acc[j][i][1] += t[j].y; acc[j][i][2] += t[j].z; acc[j][i][3] += t[j].w; }
            }
        }
    };
    auto pre4 = [&](int i, int j, float (&v)[4]) { v[0] = acc[j][i][0]; v[1] = acc[j][i][1]; v[2] = acc[j][i][2]; v[3] = acc[j][i][3]; };
    // Two complete forms, chosen by one wave-uniform branch, each with its own residual registers (so the register peak is
    // the larger of the two, not their sum: the 64x64-per-wave tiles sit at the 256-register limit).
    // The common case (plain mode, bf16 output, no activation: every residual / LayerNorm-producer / shortcut GEMM and
    // the second conv of a ResBlock) takes a compact straight-line path of its own instead of threading through the
    // branches of every other mode (a kernel starts with a cold instruction cache and the generic loop touches several
    // times more instruction lines than this case executes).
    if (c16) {
        uint4 rr16[MP > 0 ? MP : 1][NJ];
        if (plain_res) {
#pragma unroll
            for (int ip = 0; ip < MP; ++ip) {
                const size_t mo = (size_t)min(msw[ip], p.M - 1) * p.res_ld;
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    rr16[ip][j] = *reinterpret_cast<const uint4*>(p.residual + mo + min(nbase + j * 16 + cg8, p.N - 8));
            }
        } else {
#pragma unroll
            for (int ip = 0; ip < MP; ++ip)
#pragma unroll
                for (int j = 0; j < NJ; ++j) rr16[ip][j] = make_uint4(0, 0, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): every epilogue load has landed
        ln_apply();
        add_bias_rowvec();
        // the half-wave exchange IN PLACE (acc[j][2 ip] <- channels run .. run+3, acc[j][2 ip + 1] <- run+4 .. run+7 of the
        // lane's row): no second copy of the tile's values is ever live
#pragma unroll
        for (int ip = 0; ip < MP; ++ip)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[j][2 * ip][e]), __float_as_uint(acc[j][2 * ip + 1][e]), false, false);
                    acc[j][2 * ip][e] = __uint_as_float(sw[0]);
                    acc[j][2 * ip + 1][e] = __uint_as_float(sw[1]);
                }
#pragma unroll
        for (int ip = 0; ip < MP; ++ip)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float v[8], q[8];
                unpack8(rr16[ip][j], q);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = acc[j][2 * ip][e] + q[e]; v[4 + e] = acc[j][2 * ip + 1][e] + q[4 + e]; }
                const uint4 o = pack8(v);
                const int n8 = nbase + j * 16 + cg8;
                const bool lo = msw[ip] < p.M && n8 < p.N, hi = lo && n8 + 8 <= p.N;   // (N % 4 == 0: a run cut by the edge keeps its first 4 channels)
                bf16_t* dst = reinterpret_cast<bf16_t*>(p.out) + (size_t)msw[ip] * p.out_ld + n8;
                if (hi) *reinterpret_cast<uint4*>(dst) = o;
                else if (lo) *reinterpret_cast<uint2*>(dst) = make_uint2(o.x, o.y);
                if (p.ln_out) {   // (every lane takes part in the row swap of xadd16: predicated values, no early exit)
                    unpack8(o, q);
                    float s1 = (lo ? (q[0] + q[1]) + (q[2] + q[3]) : 0.f) + (hi ? (q[4] + q[5]) + (q[6] + q[7]) : 0.f);
                    float s2 = (lo ? (q[0] * q[0] + q[1] * q[1]) + (q[2] * q[2] + q[3] * q[3]) : 0.f) +
                               (hi ? (q[4] * q[4] + q[5] * q[5]) + (q[6] * q[6] + q[7] * q[7]) : 0.f);
                    s1 = xadd16(s1); s2 = xadd16(s2);                        // the two 8-channel halves of the block
                    if ((g & 1) == 0) ln_put(2 * ip + (g >> 1), j, s1, s2);  // lower half-wave: fragment 2 ip, upper: 2 ip + 1
                }
            }
    } else {
        uint2 rr[MI][NJ];
        if (plain_res) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const size_t mo = (size_t)min(mrow[i] + r, p.M - 1) * p.res_ld;
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    rr[i][j] = *reinterpret_cast<const uint2*>(p.residual + mo + min(nbase + j * 16 + cgo, p.N - 4));
            }
        } else {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) rr[i][j] = make_uint2(0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): every epilogue load has landed
        ln_apply();
        add_bias_rowvec();
        if (compact) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m = mrow[i] + r;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int n = nbase + j * 16 + cgo;
                    const bool ok = m < p.M && n < p.N;
                    float v[4];
                    pre4(i, j, v);
                    v[0] += bf_lo(rr[i][j].x); v[1] += bf_hi(rr[i][j].x); v[2] += bf_lo(rr[i][j].y); v[3] += bf_hi(rr[i][j].y);
                    uint2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]);
                    if (ok) *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.out_ld + n) = o;
                    if (p.ln_out) {   // (every lane takes part in the row swaps: predicated values, no early exit)
                        const float q0 = bf_lo(o.x), q1 = bf_hi(o.x), q2 = bf_lo(o.y), q3 = bf_hi(o.y);
                        float s1 = ok ? (q0 + q1) + (q2 + q3) : 0.f;
                        float s2 = ok ? (q0 * q0 + q1 * q1) + (q2 * q2 + q3 * q3) : 0.f;
                        s1 = xadd32(s1); s2 = xadd32(s2);   // channels [0,4) + [4,8) (lane groups 0, 2) and [8,12) + [12,16) (1, 3)
                        s1 = xadd16(s1); s2 = xadd16(s2);
                        if (g == 0) ln_put(i, j, s1, s2);
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int nb = nbase + j * 16;
                // q | k parts of the split epilogue on the 16-byte form (a 16-column block lies in ONE part: ns0, ns1 % 16 == 0)
                if (s16 && nb < p.ns0 + p.ns1) {
#pragma unroll
                    for (int ip = 0; ip < MP; ++ip) {
                        float a[4], b[4], v[8];
                        pre4(2 * ip, j, a);
                        pre4(2 * ip + 1, j, b);
                        swap8(a, b, v);
                        const int n8 = nb + cg8;
                        if (msw[ip] >= p.M) continue;
                        bf16_t* dst = n8 < p.ns0 ? reinterpret_cast<bf16_t*>(p.out) + (size_t)msw[ip] * p.out_ld + n8
                                                 : p.out1 + (size_t)msw[ip] * p.out1_ld + (n8 - p.ns0);
                        *reinterpret_cast<uint4*>(dst) = pack8(v);
                    }
                    continue;
                }
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int m = mrow[i] + r;
                    const int n = nb + cgo;
                    if (m >= p.M || n >= p.N) continue;
                    const int b = bidx[i];
                    float v[4];
                    pre4(i, j, v);
                    if (p.act == MSD_ACT_SILU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
                    } else if (p.act == MSD_ACT_QUICK_GELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] * fast_rcp(1.0f + __expf(-1.702f * v[e]));
                    }
                    if (split_mode == 0) {
                        v[0] += bf_lo(rr[i][j].x); v[1] += bf_hi(rr[i][j].x); v[2] += bf_lo(rr[i][j].y); v[3] += bf_hi(rr[i][j].y);
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
        }
    }
    MSD_STAMP(7);
    if (p.ln_out) {   // (plain mode, bf16 output, no activation: the two `compact` forms above; host-checked)
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the block sums are in LDS (raw barrier: the stores stay in flight)
        __builtin_amdgcn_s_barrier();
        if (wn == 0 && g == 0) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m = mrow[i] + r;
                if (m >= p.M) continue;
                const float* src = lnred + (size_t)(row0 + i * 16 + r) * nb16 * 2;
                float a = 0.f, q = 0.f;
                for (int k = 0; k < nb16; ++k) { a += src[2 * k]; q += src[2 * k + 1]; }   // ascending blocks: the canonical order
                reinterpret_cast<float2*>(p.ln_out)[(size_t)m * p.ln_out_slots + tile_n] = make_float2(a, q);
            }
        }
    }
}

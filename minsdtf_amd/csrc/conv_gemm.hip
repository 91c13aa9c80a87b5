// Implicit-GEMM convolution / dense layer for gfx950: bf16 MFMA 16x16x32, fp32 accumulate.
//
// Replaces PaddedConv2D / Dense / UpSampling2D+conv / Concatenate+conv / GEGLU of the reference
// (layers.py:17-25, diffusion_model.py:22-153) — see include/minsdtf_hip.h for the contract.
//
// Structure (one workgroup = one 128 x BN output tile, every wave owns a 64 x 32 sub-tile):
//   * K is walked in 64-channel tiles of one filter tap at a time, so the A tile is a gather of 128
//     pixel rows x 128 contiguous bytes (NHWC): 8 lanes fetch one pixel's 128 B -> coalesced;
//     zero padding, stride 2, nearest x2 upsampling and the channel concat of two tensors are all
//     address generation in the loader, nothing is materialised in HBM;
//   * tiles go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip) into a ring
//     of S stages; S-1 tiles are in flight while one is consumed, retired with a COUNTED
//     s_waitcnt vmcnt(N) + one raw s_barrier per K tile.  (The first version staged through
//     registers one tile ahead and was bound by the L2 round trip: ~1.3 us per K tile against
//     ~0.1 us of MFMA work.)  Out-of-image / out-of-range rows read a 128-byte page of zeros;
//   * the LDS image is [row][8 x 16 B]; LDS-DMA writes are lane-linear, so the bank swizzle
//     (16-byte chunk index XOR (row>>1)&7) is applied to the per-lane SOURCE address and undone in
//     the fragment reads: ds_read_b128 of 16 rows x one chunk is conflict-free;
//   * the MFMA is issued "swapped" (A operand = weights, B operand = activations) so each lane's
//     accumulator holds 4 CONSECUTIVE output channels of one pixel: the epilogue reads bias /
//     time-embedding / residual and writes the result with 8- or 16-byte vectors;
//   * small-M layers (8x8 and 16x16 levels at batch 1) are split over K into fp32 partial slabs
//     reduced by a second tiny kernel, deterministically (no atomics).
#include "conv_common.h"

#ifdef MSD_STAMPS
extern "C" MSD_API int msd_debug_stamps(unsigned long long* host_out, int count) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * (size_t)count);
}
#endif

// Tile configurations (BM x BN output tile, WGM x WGN waves, each wave (BM/WGM) x (BN/WGN)):
//   128x128 / 2x4 waves (64x32 per wave)   128x64 / 2x2 (64x32)   64x64 / 2x2 (32x32)
//   64x128 / 2x2 (32x64)                   256x128 / 4x2 (64x64)
// Smaller tiles put more workgroups on the chip for the small-M layers of the 16x16 / 8x8 levels;
// larger tiles move fewer L2->LDS bytes per FLOP (the per-CU LDS-DMA rate is what bounds this
// kernel).  The host picks per layer shape (measured table, minsdtf_amd/tuning.py).
//
// DENSE = true is the 1x1 / Dense form (ksize 1, stride 1, no upsampling): a row of A is one pixel's
// channel vector, so the loader keeps ONE 32-bit byte offset per staged row and a K tile costs one
// v_add + one LDS-DMA (scalar base + vector offset) per piece.  The general form generates tap /
// padding / stride / upsampling addresses (two scalar divisions and a bounds test per piece) for every
// K tile; in-kernel stamps showed that work sitting on the critical path of each K step (~1.2k cycles
// per step against 128-256 of MFMA).  Both forms now issue the fragment ds_reads of the current tile
// BEFORE the address generation + DMA of the tile S-1 ahead, so that work overlaps the LDS latency.
template <int BM, int BN, int WGM, int WGN, int S, bool DENSE>
__global__ __launch_bounds__(WGM * WGN * 64) void conv_gemm_dma_kernel(CG_HOT_PARAMS, const CGArgs p) {
    CG_HOT_UNPACK;
    constexpr int NW = WGM * WGN;               // waves per workgroup
    constexpr int NT = NW * 64;                 // threads
    constexpr int WMT = BM / WGM, WNT = BN / WGN;   // wave tile
    constexpr int MI = WMT / 16, NJ = WNT / 16;
    constexpr int RPP = NT / 8;                 // rows covered by one pass of the workgroup
    constexpr int BNP = (BN + RPP - 1) / RPP * RPP;     // weight rows as staged (BN = 80: padded to the DMA round)
    constexpr int AR = BM * 8 / NT, BR = BNP * 8 / NT;  // 16-byte pieces per thread per tile
    static_assert(AR >= 1 && BR >= 1 && (BM * 8) % NT == 0 && WMT % 16 == 0 && WNT % 16 == 0, "tile config");
    constexpr int L = AR + BR;                  // DMA instructions per thread per tile
    constexpr int A_BYTES = BM * 128, B_BYTES = BNP * 128, ST_BYTES = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int r = lane & 15, g = lane >> 4;
    MSD_STAMP(0);
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one),
    // so give each XCD a CONTIGUOUS run of tiles: neighbouring tiles re-read the same pixel rows
    // (9 taps, all n-tiles) and then hit that XCD's private 4 MiB L2 instead of the Infinity Cache.
    // Pure speed: any placement computes the same result.
    const int tile = xcd_remap(blockIdx.x, hot_tiles_m * hot_tiles_n);   // (= gridDim.x)
    // which operand the XCD-contiguous run shares: the pixel rows (n fastest) when the activation
    // tensor is the bigger one, the weight rows (m fastest) for the weight-heavy small-M layers —
    // otherwise every XCD's L2 pulls its own copy of up to 59 MB of weights per layer
    const int tdiv = hot_m_fast ? hot_tiles_m : hot_tiles_n;   // one division, by a host-prepared magic number
    const int tq = udiv_magic(tile, tdiv, hot_mg_tdiv), tr = tile - tq * tdiv;
    const int tile_n = hot_m_fast ? tq : tr;
    const int tile_m = hot_m_fast ? tr : tq;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kt_begin = blockIdx.y * hot_nk_per;
    const int kt_end = min(hot_nk, kt_begin + hot_nk_per);
    const int nkt = kt_end - kt_begin;

    // ---- loader coordinates: thread -> (row = lrow + RPP*i, LDS chunk position = tid&7) ----------
    const int cpos = tid & 7, lrow = tid >> 3;
    const int Hl = p.upsample ? 2 * p.h_in : p.h_in;
    const int Wl = p.upsample ? 2 * p.w_in : p.w_in;
    const char* zero = reinterpret_cast<const char*>(g_zero_page) + cpos * 16;
    int ab[AR], ay[AR], ax[AR], asrc[AR];
    uint32_t aoff0[AR], aoff1[AR], woff[BR];   // DENSE: byte offsets of the staged rows from a0 / a1 / w
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int row = lrow + RPP * i;
        const int m = m0 + row;
        asrc[i] = (cpos ^ ((row >> 1) & 7)) * 8;  // swizzle on the SOURCE chunk (LDS-DMA writes linearly)
        if constexpr (DENSE) {
            // rows past M re-read the last row: their accumulators are never stored
            const uint32_t mc = (uint32_t)min(m, hot_M - 1);
            aoff0[i] = (mc * (uint32_t)hot_c0 + (uint32_t)asrc[i]) * 2u;
            aoff1[i] = (mc * (uint32_t)hot_c1 + (uint32_t)asrc[i]) * 2u;
            ab[i] = ay[i] = ax[i] = 0;
            continue;
        }
        if (m < hot_M) {
            const int b = udiv_magic(m, p.hw_out, p.mg_hw);
            const int rem = m - b * p.hw_out;
            const int y = udiv_magic(rem, p.w_out, p.mg_w);
            const int x = rem - y * p.w_out;
            ab[i] = b * p.h_in * p.w_in;
            ay[i] = y * p.stride - p.pad;
            ax[i] = x * p.stride - p.pad;
        } else {
            ab[i] = 0; ay[i] = -(1 << 20); ax[i] = -(1 << 20);
        }
    }
#pragma unroll
    for (int i = 0; i < BR; ++i) {
        const int row = lrow + RPP * i;
        const int n = n0 + row;
        // columns past N re-read the last weight row (never stored)
        woff[i] = (uint32_t)min(n, hot_N - 1) * hot_w_rs + (uint32_t)((cpos ^ ((row >> 1) & 7)) * 16);
    }
    // LDS destination of this wave's pass i: 8 rows x 128 B, lane-linear
    const uint32_t lds_wave = lds0 + (uint32_t)(wave * 8) * 128u;

    auto issue_tile = [&](int kt, int stage) {
        if constexpr (DENSE) {
            const uint32_t sbase = lds_wave + (uint32_t)stage * ST_BYTES;
            const int c = kt * 64;
            const bool first = c < hot_c0;                       // wave-uniform: which tensor of the concat
            const bf16_t* abase = first ? hot_a0 : hot_a1;
            const uint32_t cb = (uint32_t)(first ? c : c - hot_c0) * 2u;
#pragma unroll
            for (int i = 0; i < AR; ++i) dma16s(abase, (first ? aoff0[i] : aoff1[i]) + cb, sbase + (uint32_t)(RPP * i) * 128u);
#pragma unroll
            for (int i = 0; i < BR; ++i) dma16s(hot_w, woff[i] + (uint32_t)kt * hot_w_ks, sbase + A_BYTES + (uint32_t)(RPP * i) * 128u);
            return;
        }
        // general form, branch-free: coordinates clamped into the image, an out-of-image tap selects the zero page by
        // mask arithmetic on the 64-bit address; (tap, channel chunk) by a magic-number division; ky = tap / 3 as
        // (tap * 11) >> 5 (exact for tap < 9; ksize 1 has tap 0)
        // K tiles past the taps of a0|a1 (ResBlock shortcut folded into conv2): channels of a2|a3 at the output pixel,
        // i.e. the centre tap (pad 1, stride 1) or the only tap (1x1)
        const bool extra = kt >= p.nk_main;
        const int tap = extra ? 0 : udiv_magic(kt, p.nkc, p.mg_nkc);
        const int c = extra ? (kt - p.nk_main) * 64 : (kt - tap * p.nkc) * 64;
        const int ky = extra ? p.pad : (tap * 11) >> 5;
        const int kx = extra ? p.pad : tap - ky * 3;
        const int cA = extra ? p.c2 : hot_c0;
        const bool first = c < cA;
        const uint64_t sb = (uint64_t)(extra ? (first ? p.a2 : p.a3) : (first ? hot_a0 : hot_a1));
        const int csrc = first ? cA : (extra ? p.K - p.nk_main * 64 - p.c2 : hot_c1), coff = first ? c : c - cA;
        const uint32_t sbase = lds_wave + (uint32_t)stage * ST_BYTES;
        const uint64_t zaddr = (uint64_t)zero;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            int iy = ay[i] + ky, ix = ax[i] + kx;
            const uint32_t m32 = (((unsigned)iy < (unsigned)Hl) && ((unsigned)ix < (unsigned)Wl)) ? 0xFFFFFFFFu : 0u;
            iy = min(max(iy, 0), Hl - 1); ix = min(max(ix, 0), Wl - 1);
            if (p.upsample) { iy >>= 1; ix >>= 1; }
            const uint32_t off = (uint32_t)((ab[i] + iy * p.w_in + ix) * csrc + coff + asrc[i]) * 2u;
            const uint64_t m64 = ((uint64_t)m32 << 32) | m32;
            dma16(reinterpret_cast<const void*>(((sb + off) & m64) | (zaddr & ~m64)), sbase + (uint32_t)(RPP * i) * 128u);
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) dma16s(hot_w, woff[i] + (uint32_t)kt * hot_w_ks, sbase + A_BYTES + (uint32_t)(RPP * i) * 128u);
    };

    f32x4 acc[NJ][MI];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int s = 0; s < S - 1; ++s)
        if (s < nkt) issue_tile(kt_begin + s, s);
    MSD_STAMP(1);

    const int swz = r >> 1;  // (row>>1)&7 for row = 16*q + r
    const int rw = cg_wrow(r), swzw = rw >> 1;   // weight rows enter the MFMA in the order 0-3, 8-11, 4-7, 12-15 (cg_epilogue)
    struct Frags { bf16x8 a[2][MI], w[2][NJ]; };
    auto read_frags = [&](Frags& f, int stg) {
        const char* bA = smem + stg * ST_BYTES + (wm * WMT + r) * 128;
        const char* bB = smem + stg * ST_BYTES + A_BYTES + (wn * WNT + rw) * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int coff = ((ks * 4 + g) ^ swz) << 4, coffw = ((ks * 4 + g) ^ swzw) << 4;
#pragma unroll
            for (int i = 0; i < MI; ++i) f.a[ks][i] = *reinterpret_cast<const bf16x8*>(bA + i * 16 * 128 + coff);
#pragma unroll
            for (int j = 0; j < NJ; ++j) f.w[ks][j] = *reinterpret_cast<const bf16x8*>(bB + j * 16 * 128 + coffw);
        }
    };
    auto mfma_tile = [&](const Frags& f) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.w[ks][j], f.a[ks][i], acc[j][i], 0, 0, 0);
    };
    // (Tried: reading the fragments of tile it+1 into a second register set before the MFMAs of tile it, for
    //  rings of 4+ stages — 0.37 vs 0.31 us per K tile on the 64x64 tile: one tile fewer in flight costs more
    //  than the overlapped LDS reads gain.)
    int stage = 0;
    for (int it = 0; it < nkt; ++it) {
        // retire tile `it`: all but the tiles issued after it may stay in flight
        const int later = min(nkt, it + S - 1) - (it + 1);
        wait_vmcnt_tiles<L, S - 2>(later);
        __builtin_amdgcn_s_barrier();  // tile `it` visible to all waves; stage (it-1)%S free for reuse
#ifdef MSD_STAMPS
        if (it == 0) MSD_STAMP(2);
        if (it == (nkt >> 1)) MSD_STAMP(5);
#endif
        // fragments of tile `it` first (their LDS latency runs under the address generation + DMA issue of
        // the tile S-1 ahead), then the MFMAs
        Frags f;
        read_frags(f, stage);
        if (it + S - 1 < nkt) {   // stage (it-1)%S: every wave finished reading it before the barrier above
            int st = stage + S - 1;
            if (st >= S) st -= S;
            issue_tile(kt_begin + it + S - 1, st);
        }
        mfma_tile(f);
        if (++stage == S) stage = 0;
    }
    MSD_STAMP(3);
    int mrow[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) mrow[i] = m0 + wm * WMT + i * 16;
    cg_epilogue<MI, NJ, false, DENSE>(p, acc, mrow, n0 + wn * WNT, r, g, reinterpret_cast<float*>(smem), wn, WGN, wm * WMT, BM, tile_n);
#ifdef MSD_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores have left the wave
    MSD_STAMP(4);
#endif
}

// split-K: sum the fp32 slabs in slice order, then the same epilogue (plain mode only).
// SL = the slice count as a compile-time constant for the counts the tuner uses (every slab load in flight at once, none
// redundant: the generic form keeps 8 clamped loads in flight, i.e. issues 8 loads for 3 slices); SL = 0: any count.
template <int SL>
__global__ __launch_bounds__(256) void splitk_finalize_kernel(const float* hot_ws, int hot_M, int hot_N, uint32_t mg_nq, int slices, const CGArgs p) {
    // (leading scalars: kernarg preload, conv_common.h — the slab loads below depend on nothing else)
    const int nq = hot_N >> 2;
    const int idx = blockIdx.x * 256 + threadIdx.x;   // (M * N / 4 < 2^31: checked on the host)
    if (idx >= hot_M * nq) return;
    const int m = udiv_magic(idx, nq, mg_nq);         // (was a 64-bit division by a runtime value: ~100 instructions)
    const int n = (idx - m * nq) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    const float* src = hot_ws + (size_t)m * hot_N + n;
    const size_t zs = (size_t)hot_M * hot_N;
    if constexpr (SL > 0) {
        // ONE memory round trip: the epilogue's operands (bias, time-embedding row, residual) do not depend on the slabs, so their loads go
        // out together with the slab loads — unconditionally, from addresses that are valid whatever is present (an absent operand
        // reads the slab and is masked after the wait: a load under a branch would make hipcc wait vmcnt(0) at the join).  Before:
        // slabs -> wait -> bias -> wait -> row -> wait -> residual -> wait, four dependent round trips in a 5 us launch.
        const int step = p.step_ptr ? *p.step_ptr : 0;   // (scalar load, wave-uniform)
        const int b = udiv_magic(m, p.hw_out, p.mg_hw);
        const bool has_b = p.bias != nullptr, has_rv = p.rowvec != nullptr;
        const bool has_r = p.residual != nullptr && p.split_mode == 0;
        const float* dummy = src;
        const float4 bv = *reinterpret_cast<const float4*>(has_b ? p.bias + n : dummy);
        const float4 rv = *reinterpret_cast<const float4*>(has_rv ? p.rowvec + (size_t)step * p.rv_step_stride + (size_t)b * p.rv_batch_stride + n : dummy);
        const uint2 rr = *reinterpret_cast<const uint2*>(has_r ? reinterpret_cast<const char*>(p.residual + (size_t)m * p.res_ld + n)
                                                               : reinterpret_cast<const char*>(dummy));
        float4 t[SL];
#pragma unroll
        for (int u = 0; u < SL; ++u) t[u] = *reinterpret_cast<const float4*>(src + (size_t)u * zs);
#pragma unroll
        for (int u = 0; u < SL; ++u) { v[0] += t[u].x; v[1] += t[u].y; v[2] += t[u].z; v[3] += t[u].w; }   // slice order
        cg_store4_pre(p, m, b, n, v, has_b ? bv : make_float4(0.f, 0.f, 0.f, 0.f), has_b, has_rv ? rv : make_float4(0.f, 0.f, 0.f, 0.f), has_rv,
                      has_r ? rr : make_uint2(0u, 0u), has_r);
        return;
    } else {
        // 8 slab loads in flight per thread (a one-load-per-iteration loop pays a full memory round trip per slice:
        // 6 us for 12 slices of a 128 x 1280 layer); the adds keep the slice order, so the result is unchanged
        for (int z0 = 0; z0 < slices; z0 += 8) {
            float4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const float4*>(src + (size_t)min(z0 + u, slices - 1) * zs);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (z0 + u < slices) { v[0] += t[u].x; v[1] += t[u].y; v[2] += t[u].z; v[3] += t[u].w; }
        }
    }
    const int step = p.step_ptr ? *p.step_ptr : 0;
    cg_store4(p, m, udiv_magic(m, p.hw_out, p.mg_hw), n, step, v);
}
static void launch_finalize(const CGArgs& a, int slices, hipStream_t stream) {
    const long long quads = (long long)a.M * (a.N / 4);
    const dim3 grid((unsigned)((quads + 255) / 256));
    const uint32_t mg = udiv_magic_of(a.N / 4);
    switch (slices) {
#define F(SL) case SL: hipLaunchKernelGGL(splitk_finalize_kernel<SL>, grid, dim3(256), 0, stream, a.ws, a.M, a.N, mg, slices, a); break;
        F(2) F(3) F(4) F(6) F(8) F(12)
#undef F
        default: hipLaunchKernelGGL(splitk_finalize_kernel<0>, grid, dim3(256), 0, stream, a.ws, a.M, a.N, mg, slices, a); break;
    }
}

// ---- tile configurations of the LDS-DMA kernel ------------------------------------------------
struct TileCfg { int bm, bn, threads, lds, stages; };
#define MSD_TILE_CFGS(X) \
    X(0, 128, 128, 2, 4, 3, 3) \
    X(1, 128, 64, 2, 2, 3, 3)  \
    X(2, 64, 64, 2, 2, 4, 4)   \
    X(3, 64, 128, 2, 2, 3, 3)  \
    X(4, 256, 128, 4, 2, 3, 3) \
    X(5, 128, 128, 2, 4, 4, 4) \
    X(6, 64, 64, 2, 2, 8, 8)   \
    X(7, 64, 128, 2, 2, 5, 5)  \
    X(8, 128, 64, 2, 2, 5, 5)  \
    X(9, 128, 80, 4, 1, 3, 3)  \
    X(10, 128, 80, 4, 1, 4, 4) \
    X(11, 64, 64, 2, 4, 4, 14) \
    X(12, 128, 64, 4, 2, 3, 13) \
    X(13, 64, 128, 2, 4, 3, 13) \
    X(14, 128, 128, 2, 2, 3, 23) \
    X(15, 128, 128, 2, 2, 4, 24) \
    X(16, 128, 64, 2, 1, 4, 24)  \
    X(17, 64, 128, 1, 2, 4, 24)  \
    X(18, 128, 160, 4, 1, 3, 3)
// Last column = the `stages` request that selects the entry (the first entry of a tile size is its
// default).  Codes 10 + depth are the same tile on 8 waves (32x16 / 32x32 per wave): two waves per SIMD
// even when a launch puts one workgroup on a CU, so one wave's LDS-DMA issue and LDS latency overlap the
// other's MFMAs.  Codes 20 + depth: 64x64 per wave (4 / 2 waves): the fewest fragment ds_reads per MFMA (8 reads
// feed 16 MFMAs) — with 32x32 or 64x32 wave tiles the LDS read port (128 B/clk per CU), not the matrix
// cores, bounds the K loop.
// (128x80: for N = 320 / 640 at small batch — 64 x 4 = 256 workgroups at M = 8192, one per CU, where
//  64-wide tiles make 320 and 128-wide ones 192; the 80 weight rows are staged as 96)
// (128x160, four waves of 32 rows x 160 columns: for the 16x16-level GEGLU projection at batch 1 — M = 512, N = 10,240, K = 1,280
//  — 4 x 64 = 256 workgroups, one per CU, each moving (128 + 160) rows per K step instead of 640 workgroups of 64x128 moving
//  (64 + 128): the layer is bound by the L2 -> LDS bytes per CU, and this is the fewest for a grid that fills the chip once)
constexpr int cfg_lds(int bm, int bn, int threads, int st) { return st * (bm + (bn + threads / 8 - 1) / (threads / 8) * (threads / 8)) * 128; }
static const TileCfg g_cfgs[] = {
#define X(id, bm, bn, wgm, wgn, st, code) {bm, bn, wgm * wgn * 64, cfg_lds(bm, bn, wgm * wgn * 64, st), code},
    MSD_TILE_CFGS(X)
#undef X
};
constexpr int NUM_TILE_CFGS = 19;

int msd_conv_halo_launch(const CGArgs& a, int th, int bn, int stages, int variant, int slices, hipStream_t stream);
int msd_conv_wreg_nj(int bm, int bn, int stages);   // 16-column blocks per wave of a built configuration, 0: not built
int msd_conv_wreg_launch(const CGArgs& a, int bm, int bn, int stages, int slices, bool dense, hipStream_t stream);
int msd_conv_big_nj(int bm, int bn, int code);      // conv_big.hip: 16-column blocks per wave of a built configuration, 0: not built
int msd_conv_big_launch(const CGArgs& a, int bm, int bn, int code, int slices, bool dense, hipStream_t stream);
int msd_conv_bighalo_nj(int bn, int code);              // conv_big.hip, halo-image variant (tile_m 5256, stages 20 + code)
int msd_conv_bighalo_launch(const CGArgs& a, int bn, int code, int slices, hipStream_t stream);
// tile_m ranges of the forms (one predicate each, mirrored by minsdtf_amd/tuning.py form_of): [1000, 3000) halo tiles, [3000, 4000)
// row panels, [4000, 5000) wreg, [5000, 6000) big; anything from 6000 up is refused
static inline bool cg_is_wreg(int tile_m) { return tile_m >= 4000 && tile_m < 5000; }
static inline bool cg_is_big(int tile_m) { return tile_m >= 5000 && tile_m < 6000; }
bool msd_conv_rowpanel_eligible(const CGArgs& a, int rows, int wg_cols);
int msd_conv_rowpanel_launch(CGArgs a, int rows, int wg_cols, hipStream_t stream);

static bool g_cg_attr_done = false;
static int g_conv_dense = 1; // 1 = 1x1 / Dense layers take the DENSE loader (default), 0 = the general loader (A/B runs)
void msd_set_conv_dense(int v) { g_conv_dense = v; }

int msd_conv_gemm_init() {
    if (g_cg_attr_done) return MSD_OK;
    hipError_t e;
    e = hipSuccess;
#define X(id, bm, bn, wgm, wgn, st, code)                                                                           \
    if (e == hipSuccess)                                                                                            \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_dma_kernel<bm, bn, wgm, wgn, st, false>),  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, cfg_lds(bm, bn, wgm * wgn * 64, st));   \
    if (e == hipSuccess)                                                                                            \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_dma_kernel<bm, bn, wgm, wgn, st, true>),   \
                                hipFuncAttributeMaxDynamicSharedMemorySize, cfg_lds(bm, bn, wgm * wgn * 64, st));
    MSD_TILE_CFGS(X)
#undef X
    if (e != hipSuccess) MSD_FAIL((int)e, "hipFuncSetAttribute(conv_gemm): %s", hipGetErrorString(e));
    g_cg_attr_done = true;
    return MSD_OK;
}

// tile width the launch will use (same rules as msd_conv_gemm below)
static int cg_effective_bn(const MsdConvGemm* q) {
    if (cg_is_wreg(q->tile_m) || cg_is_big(q->tile_m)) return q->tile_n;   // wreg / big form: the request IS the tile (the launch fails if it is not built)
    int bn = q->tile_m >= 3000 ? 64 : q->tile_n;   // (a row-panel request that is not eligible runs on the 128x64 tile)
    if (bn == 0) bn = (q->N % 128 == 0 || q->N > 1024) ? 128 : 64;
    if (bn == 80 && q->act == MSD_ACT_GEGLU) bn = 64;
    return bn;
}

extern "C" int msd_conv_gemm_ln_slots(const MsdConvGemm* q) {
    if (!q || q->N <= 0) MSD_FAIL(MSD_E_ARG, "conv_gemm_ln_slots: bad arguments");
    if (q->tile_m >= 1000 && q->ksize == 3) MSD_FAIL(MSD_E_UNSUPPORTED, "conv_gemm_ln_slots: 1x1 / dense launches only");
    const int bn = cg_effective_bn(q);
    return (q->N + bn - 1) / bn;
}

extern "C" int msd_conv_gemm(const MsdConvGemm* q, msd_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!q) MSD_FAIL(MSD_E_ARG, "conv_gemm: null params");
    if (!q->a0 || !q->w) MSD_FAIL(MSD_E_ARG, "conv_gemm: null a0/w");
    if (!q->out && !(q->split_mode == 1 && q->ns0 == 0)) MSD_FAIL(MSD_E_ARG, "conv_gemm: null out");
    if (q->batch <= 0 || q->h_in <= 0 || q->w_in <= 0 || q->h_out <= 0 || q->w_out <= 0)
        MSD_FAIL(MSD_E_ARG, "conv_gemm: non-positive dims");
    if (q->c0 <= 0 || (q->c0 % 64) || q->c1 < 0 || (q->c1 % 64) || (q->c1 > 0 && !q->a1))
        MSD_FAIL(MSD_E_ARG, "conv_gemm: channel counts must be positive multiples of 64 (c0=%d c1=%d)", q->c0, q->c1);
    // `pad` is the LEADING (top / left) zero padding; the trailing padding is implied by the output
    // size and must be 0 or 1: symmetric pad 1 (the UNet / decoder convs) and the VAE encoder's
    // ((0,1),(0,1)) stride-2 padding (image_encoder.py:28) are both expressible.
    if (!((q->ksize == 1 && q->pad == 0) || (q->ksize == 3 && (q->pad == 0 || q->pad == 1))))
        MSD_FAIL(MSD_E_UNSUPPORTED, "conv_gemm: ksize/pad %d/%d", q->ksize, q->pad);
    if (q->stride != 1 && q->stride != 2) MSD_FAIL(MSD_E_UNSUPPORTED, "conv_gemm: stride %d", q->stride);
    {
        const int hl = q->upsample ? 2 * q->h_in : q->h_in, wl = q->upsample ? 2 * q->w_in : q->w_in;
        auto dim_ok = [&](int in, int out) {
            for (int e = 0; e <= (q->ksize == 3 ? 1 : 0); ++e) {
                const int span = in + q->pad + e - q->ksize;
                if (span >= 0 && span / q->stride + 1 == out) return true;
            }
            return false;
        };
        if (!dim_ok(hl, q->h_out) || !dim_ok(wl, q->w_out))
            MSD_FAIL(MSD_E_ARG, "conv_gemm: output dims %dx%d do not match the geometry (in %dx%d, k%d s%d pad %d)", q->h_out,
                     q->w_out, hl, wl, q->ksize, q->stride, q->pad);
    }
    if (q->N <= 0 || (q->N % 4)) MSD_FAIL(MSD_E_ARG, "conv_gemm: N=%d must be a positive multiple of 4", q->N);
    if (!msd_aligned16(q->a0) || !msd_aligned16(q->a1) || !msd_aligned16(q->w) || !msd_aligned16(q->out) ||
        !msd_aligned16(q->bias) || !msd_aligned16(q->rowvec) || !msd_aligned16(q->residual) ||
        !msd_aligned16(q->workspace) || !msd_aligned16(q->out1) || !msd_aligned16(q->out2))
        MSD_FAIL(MSD_E_ALIGN, "conv_gemm: pointers must be 16-byte aligned");
    if ((q->out_ld % 4) || (q->residual && (q->res_ld % 4)) || (q->rv_step_stride % 4) || (q->rv_batch_stride % 4))
        MSD_FAIL(MSD_E_ALIGN, "conv_gemm: leading dimensions must be multiples of 4");
    if (q->act < 0 || q->act > 3) MSD_FAIL(MSD_E_ARG, "conv_gemm: act");
    if (q->act == MSD_ACT_GEGLU && ((q->N % 32) || q->out_dtype != MSD_OUT_BF16 || q->split_mode || q->rowvec))
        MSD_FAIL(MSD_E_UNSUPPORTED, "conv_gemm: GEGLU needs N%%32==0, bf16 out, plain mode");
    if (q->split_mode) {
        if (q->split_mode != 1 || (q->ns0 % 4) || (q->ns1 % 4) || q->ns0 < 0 || q->ns1 < 0 || q->ns0 + q->ns1 > q->N ||
            q->out_dtype != MSD_OUT_BF16 || q->residual || q->act == MSD_ACT_GEGLU)
            MSD_FAIL(MSD_E_ARG, "conv_gemm: bad split mode arguments");
        if ((q->ns0 > 0 && (q->out_ld % 4)) || (q->ns1 > 0 && (!q->out1 || (q->out1_ld % 4))) ||
            (q->ns0 + q->ns1 < q->N && (!q->out2 || q->out2_ld < q->h_out * q->w_out)))
            MSD_FAIL(MSD_E_ARG, "conv_gemm: split outputs");
    }
    int rc = msd_conv_gemm_init();
    if (rc) return rc;

    CGArgs a;
    a.a0 = (const bf16_t*)q->a0; a.a1 = (const bf16_t*)q->a1; a.w = (const bf16_t*)q->w;
    a.bias = q->bias; a.rowvec = q->rowvec; a.step_ptr = q->step_ptr;
    a.residual = (const bf16_t*)q->residual; a.out = q->out; a.out1 = (bf16_t*)q->out1; a.out2 = (bf16_t*)q->out2;
    a.ws = q->workspace;
    a.batch = q->batch; a.h_in = q->h_in; a.w_in = q->w_in; a.c0 = q->c0; a.c1 = q->c1;
    a.h_out = q->h_out; a.w_out = q->w_out; a.ksize = q->ksize; a.stride = q->stride; a.pad = q->pad;
    a.upsample = q->upsample ? 1 : 0;
    a.hw_out = q->h_out * q->w_out;
    a.mg_hw = udiv_magic_of(a.hw_out);
    a.mg_w = udiv_magic_of(a.w_out);
    const long long M = (long long)q->batch * a.hw_out;
    if (M > (1ll << 30)) MSD_FAIL(MSD_E_ARG, "conv_gemm: M too large");
    a.M = (int)M; a.N = q->N;
    const int cin = q->c0 + q->c1;
    a.K = q->ksize * q->ksize * cin;
    a.nkc = cin / 64;
    a.mg_nkc = udiv_magic_of(a.nkc);
    // the loaders address activations and weights with 32-bit byte offsets from the tensor bases
    if ((long long)q->batch * q->h_in * q->w_in * (q->c0 > q->c1 ? q->c0 : q->c1) * 2 >= (1ll << 32) - 4096 ||
        (long long)q->N * q->ksize * q->ksize * cin * 2 >= (1ll << 32) - 4096)
        MSD_FAIL(MSD_E_UNSUPPORTED, "conv_gemm: an operand of 4 GB or more");
    a.nk = q->ksize * q->ksize * a.nkc;
    a.act = q->act; a.out_f32 = (q->out_dtype == MSD_OUT_F32);
    a.out_ld = q->out_ld; a.res_ld = q->res_ld;
    a.rv_step_stride = q->rv_step_stride; a.rv_batch_stride = q->rv_batch_stride;
    a.split_mode = q->split_mode; a.ns0 = q->ns0; a.ns1 = q->ns1; a.out1_ld = q->out1_ld; a.out2_ld = q->out2_ld;
    // 16-byte epilogue form (conv_common.h): every row it stores to / loads from starts on a 16-byte boundary, and in split
    // mode a 16-column block lies in one part
    // (N % 8: the form's residual load covers 8 channels; with N % 8 == 4 its last run would read 4 channels too far left)
    a.vec16 = (q->N % 8 == 0) && (q->out_ld % 8 == 0) && (!q->residual || q->res_ld % 8 == 0) &&
              (!q->split_mode || ((q->ns0 % 16) == 0 && (q->ns1 % 16) == 0 && (q->ns1 == 0 || q->out1_ld % 8 == 0)));

    a.a2 = (const bf16_t*)q->a2; a.a3 = (const bf16_t*)q->a3; a.c2 = q->c2; a.nk_main = a.nk;
    if (q->a2) {
        if (q->c2 <= 0 || (q->c2 % 64) || q->c3 < 0 || (q->c3 % 64) || (q->c3 > 0 && !q->a3) || q->stride != 1 || q->upsample ||
            q->h_out != q->h_in || q->w_out != q->w_in || !msd_aligned16(q->a2) || !msd_aligned16(q->a3))
            MSD_FAIL(MSD_E_ARG, "conv_gemm: shortcut operand needs stride 1, a same-size output and channel counts that are multiples of 64");
        if ((long long)a.M * (q->c2 > q->c3 ? q->c2 : q->c3) * 2 >= (1ll << 32) - 4096) MSD_FAIL(MSD_E_UNSUPPORTED, "conv_gemm: an operand of 4 GB or more");
        a.K += q->c2 + q->c3;
        a.nk += (q->c2 + q->c3) / 64;
    } else if (q->c2 || q->c3 || q->a3) {
        MSD_FAIL(MSD_E_ARG, "conv_gemm: c2 / c3 / a3 without a2");
    }
    // (again, with the shortcut channels in K: the weight loader's 32-bit byte offsets are n * K * 2 + ...)
    if ((long long)a.N * a.K * 2 >= (1ll << 32) - 4096) MSD_FAIL(MSD_E_UNSUPPORTED, "conv_gemm: an operand of 4 GB or more");
    if (q->w_layout < 0 || q->w_layout > 2) MSD_FAIL(MSD_E_ARG, "conv_gemm: w_layout %d", q->w_layout);
    if (q->tile_m >= 6000 || q->tile_m < 0) MSD_FAIL(MSD_E_ARG, "conv_gemm: tile_m %d names no kernel form", q->tile_m);
    const bool wreg = cg_is_wreg(q->tile_m), big = cg_is_big(q->tile_m);
    // big form, stages code + 10: chunk-major K walk (the halo-tile kernel's order and numerics class)
    // stages code + 20: the same walk on a staged 18 x 18-pixel halo per chunk (3x3 / stride 1 / pad 1 on whole 16 x 16-pixel tiles)
    const bool big_hi = big && q->stages >= 20;
    const bool big_km = big && q->stages >= 10 && !big_hi;
    const int big_code = big_hi ? q->stages - 20 : (big_km ? q->stages - 10 : q->stages);
    if (big) {
        if ((big_km || big_hi) && (q->ksize != 3 || (q->a2 && !big_hi)))
            MSD_FAIL(MSD_E_UNSUPPORTED, "conv_gemm: the chunk-major walk of the big-tile form is for 3x3 convs (with a shortcut operand: on the staged-halo variant only)");
        if (big_hi) {
            if (q->tile_m != 5256 || !msd_conv_bighalo_nj(q->tile_n, big_code))
                MSD_FAIL(MSD_E_UNSUPPORTED, "conv_gemm: no halo-image big-tile configuration %d x %d code %d", q->tile_m - 5000, q->tile_n, big_code);
            const int up = q->upsample ? 2 : 1;
            if (q->stride != 1 || q->pad != 1 || q->h_out != up * q->h_in || q->w_out != up * q->w_in || (q->h_out % 16) || (q->w_out % 16))
                MSD_FAIL(MSD_E_UNSUPPORTED, "conv_gemm: the halo-image big-tile form runs 3x3 / stride 1 / pad 1 convs on output images of whole 16 x 16-pixel tiles (%d x %d)", q->h_out, q->w_out);
        } else if (!msd_conv_big_nj(q->tile_m - 5000, q->tile_n, big_code))
            MSD_FAIL(MSD_E_UNSUPPORTED, "conv_gemm: no big-tile configuration %d x %d code %d", q->tile_m - 5000, q->tile_n, big_code);
        if (q->ln_out) MSD_FAIL(MSD_E_UNSUPPORTED, "conv_gemm: the big-tile form has no LayerNorm-producer epilogue (ln_out)");
        // its general loader forms pixel * row bytes with a 24-bit multiply
        if ((long long)q->batch * q->h_in * q->w_in >= (1ll << 24) || (long long)(q->c0 > q->c1 ? q->c0 : q->c1) * 2 >= (1ll << 24))
            MSD_FAIL(MSD_E_UNSUPPORTED, "conv_gemm: big-tile form needs fewer than 2^24 input pixels");
        // ... and keeps the product in 32 bits
        if ((long long)q->batch * q->h_in * q->w_in * (q->c0 > q->c1 ? q->c0 : q->c1) * 2 >= (1ll << 32) - 4096)
            MSD_FAIL(MSD_E_UNSUPPORTED, "conv_gemm: big-tile form needs input tensors under 4 GB");
        // the shortcut operand (staged-halo variant) is addressed the same way, by OUTPUT pixel x c2 / c3 row bytes (the checks above
        // hold a same-size output and no upsampling for it; the 4 GB bound on M x max(c2, c3) too)
        if (q->a2 && ((long long)q->batch * q->h_out * q->w_out >= (1ll << 24) || (long long)(q->c2 > q->c3 ? q->c2 : q->c3) * 2 >= (1ll << 24)))
            MSD_FAIL(MSD_E_UNSUPPORTED, "conv_gemm: big-tile form needs fewer than 2^24 output pixels / shortcut row bytes");
    }
    if (wreg != (q->w_layout == 2))
        MSD_FAIL(MSD_E_ARG, "conv_gemm: tile_m %d with w_layout %d (the fragment-major weight image, w_layout 2, is read by the wreg form, tile_m 4000 + rows, and by nothing else)",
                 q->tile_m, q->w_layout);
    if (wreg && ((q->N % 16) || !msd_conv_wreg_nj(q->tile_m - 4000, q->tile_n, q->stages)))
        MSD_FAIL(MSD_E_UNSUPPORTED, "conv_gemm: no wreg configuration %d x %d stages %d (N %% 16 must be 0: N = %d)", q->tile_m - 4000, q->tile_n, q->stages, q->N);
    a.w_rs = q->w_layout ? 128u : (uint32_t)a.K * 2u;
    a.w_ks = q->w_layout ? (uint32_t)a.N * 128u : 128u;
    a.ln_in = q->ln_in; a.ln_colsum = q->ln_colsum; a.ln_out = q->ln_out;
    a.ln_in_slots = q->ln_in_slots; a.ln_out_slots = q->ln_out_slots; a.ln_eps = q->ln_eps;
    a.ln_inv_k = 1.0f / (float)a.K;
    if (q->ln_in) {
        if (!q->ln_colsum || q->ln_in_slots < 1 || q->ln_in_slots > LN_MAX_SLOTS || q->ksize != 1 || !(q->ln_eps > 0.f))
            MSD_FAIL(MSD_E_ARG, "conv_gemm: LayerNorm fold needs ln_colsum, 1 <= ln_in_slots <= %d, ksize 1, ln_eps > 0", LN_MAX_SLOTS);
        if (!msd_aligned16(q->ln_colsum) || (((uintptr_t)q->ln_in) & 7u)) MSD_FAIL(MSD_E_ALIGN, "conv_gemm: ln_in / ln_colsum alignment");
    }
    if (q->ln_out) {
        if (q->split_mode || q->out_dtype != MSD_OUT_BF16 || q->act != MSD_ACT_NONE || q->ksize != 1 || (q->tile_m >= 1000 && q->tile_m < 3000) || big ||
            (((uintptr_t)q->ln_out) & 7u))
            MSD_FAIL(MSD_E_ARG, "conv_gemm: ln_out needs a plain 1x1 launch with a bf16 output and no activation");
        if (q->ln_out_slots != msd_conv_gemm_ln_slots(q))
            MSD_FAIL(MSD_E_ARG, "conv_gemm: ln_out_slots=%d, this launch writes %d partials per row", q->ln_out_slots,
                     msd_conv_gemm_ln_slots(q));
    }

    int splitk = q->splitk < 1 ? 1 : q->splitk;
    if ((q->ln_in || q->ln_out) && splitk > 1) MSD_FAIL(MSD_E_UNSUPPORTED, "conv_gemm: the LayerNorm fold is plain-K only (splitk=%d)", splitk);
    // halo variant (tile_m = 1000 + pixels per tile: 1128 = 8x16, 1256 = 16x16): spatially blocked 3x3
    int halo_th = 0;
    if (q->tile_m >= 1000 && q->tile_m < 3000) {
        const int th = (q->tile_m % 1000) / 16;   // 1128 / 1256: 8x16 / 16x16 pixels; 2128: 8x16 on 8 waves
        // (a shortcut operand - stride 1, same-size output, < 4 GB: checked above - is walked behind the slice's main chunks, conv_halo.hip)
        const bool ok = q->ksize == 3 && q->stride == 1 && q->pad == 1 && q->h_out == q->h_in &&
                        q->w_out == q->w_in && !q->upsample && (q->w_in % 16) == 0 &&
                        (long long)q->N * a.K * 2 < (1ll << 32) - 4096 &&   // 32-bit weight / activation byte offsets
                        (long long)a.M * (q->c0 > q->c1 ? q->c0 : q->c1) * 2 < (1ll << 32) - 4096 &&
                        (th == 8 || th == 16) && (q->h_in % th) == 0;
        if (ok) halo_th = th;
    }
    a.kmajor = big_km ? 1 : 0;
    if (halo_th || big_km || big_hi) {  // split-K is over 64-channel chunks (each = 9 K steps)
        if (splitk > a.nkc) splitk = a.nkc;
        a.nk_per = (a.nkc + splitk - 1) / splitk;
    } else {
        if (splitk > a.nk) splitk = a.nk;
        a.nk_per = (a.nk + splitk - 1) / splitk;
    }
    const int slices = (halo_th || big_km || big_hi) ? (a.nkc + a.nk_per - 1) / a.nk_per : (a.nk + a.nk_per - 1) / a.nk_per;
    a.nslices = slices;
    if (big_km) a.nk_per *= 9;   // (the big form counts K tiles, nine per chunk)
    if (slices > 1) {
        if (q->split_mode || q->act == MSD_ACT_GEGLU) MSD_FAIL(MSD_E_UNSUPPORTED, "conv_gemm: split-K needs plain mode");
        if ((long long)a.M * a.N / 4 >= (1ll << 31)) MSD_FAIL(MSD_E_UNSUPPORTED, "conv_gemm: split-K output too large");
        if (!q->workspace || q->workspace_floats < (long long)slices * a.M * a.N)
            MSD_FAIL(MSD_E_WORKSPACE, "conv_gemm: split-K workspace too small (%lld < %lld floats)",
                     (long long)q->workspace_floats, (long long)slices * a.M * a.N);
    }
    // tile configuration: explicit (tile_m, tile_n) or the size heuristic
    int bm = q->tile_m, bn = cg_effective_bn(q);   // (80 = 5 fragments: no x|gate pairing -> 64 for GEGLU)
    if (bm == 0 || bm >= 1000) bm = 128;   // a halo request that is not eligible falls back to 128-row tiles
    if (halo_th) {
        if (halo_th == 16 && bn != 80) bn = 128;
        a.tiles_n = (a.N + bn - 1) / bn;
        a.tiles_m = a.batch * (a.h_in / halo_th) * (a.w_in / 16);
        a.m_fast = (a.N > a.M) ? 1 : 0;
        a.mg_tdiv = udiv_magic_of(a.m_fast ? a.tiles_m : a.tiles_n);
        a.mg_tps = udiv_magic_of((a.h_in / halo_th) * (a.w_in / 16));
        a.mg_tx = udiv_magic_of(a.w_in / 16);
        rc = msd_conv_halo_launch(a, halo_th, bn, q->stages, q->tile_m >= 2000 ? 1 : 0, slices, stream);
        if (rc) return rc;
        MSD_CHECK_LAUNCH();
        if (slices > 1) {
            launch_finalize(a, slices, stream);
            MSD_CHECK_LAUNCH();
        }
        return MSD_OK;
    }
    // row-panel Dense kernel (tile_m = 3000 + rows per workgroup, tile_n = columns per workgroup): conv_rowpanel.hip
    if (q->tile_m >= 3000 && q->tile_m < 4000 && msd_conv_rowpanel_eligible(a, q->tile_m - 3000, q->tile_n)) {
        rc = msd_conv_rowpanel_launch(a, q->tile_m - 3000, q->tile_n, stream);
        if (rc) return rc;
        MSD_CHECK_LAUNCH();
        return MSD_OK;
    }
    if (wreg) {   // weights global -> VGPR in fragment order, activations through the LDS ring: conv_wreg.hip
        const int wbm = q->tile_m - 4000, wbn = q->tile_n;
        if (q->act == MSD_ACT_GEGLU && (msd_conv_wreg_nj(wbm, wbn, q->stages) % 2)) MSD_FAIL(MSD_E_UNSUPPORTED, "conv_gemm: GEGLU needs x | gate fragment pairs per wave");
        a.tiles_m = (a.M + wbm - 1) / wbm;
        a.tiles_n = (a.N + wbn - 1) / wbn;
        a.m_fast = (a.N > a.M) ? 1 : 0;
        a.mg_tdiv = udiv_magic_of(a.m_fast ? a.tiles_m : a.tiles_n);
        a.mg_tps = a.mg_tx = 0;
        const bool needs_dense = q->ln_in || q->act == MSD_ACT_GEGLU || q->split_mode;
        const bool dense = !q->rowvec && !q->a2 && q->ksize == 1 && q->stride == 1 && !q->upsample && q->h_out == q->h_in && q->w_out == q->w_in;
        if (needs_dense && !dense)
            MSD_FAIL(MSD_E_UNSUPPORTED, "conv_gemm: the LayerNorm fold, GEGLU and the q|k|v^T split run on the 1x1 / Dense form only");
        if (!cg_hot_ok(a)) MSD_FAIL(MSD_E_UNSUPPORTED, "conv_gemm: tile / K-tile / channel counts beyond the packed launch arguments (K tiles, channels < 65536; row tiles < 2^23; column tiles < 256)");
        rc = msd_conv_wreg_launch(a, wbm, wbn, q->stages, slices, dense, stream);
        if (rc) return rc;
        MSD_CHECK_LAUNCH();
        if (slices > 1) {
            launch_finalize(a, slices, stream);
            MSD_CHECK_LAUNCH();
        }
        return MSD_OK;
    }
    if (big_hi) {   // the big form on a staged halo: conv_big.hip (conv_bighalo_kernel); nk_per stays in chunks
        a.tiles_m = a.batch * (a.h_out / 16) * (a.w_out / 16);
        a.tiles_n = (a.N + q->tile_n - 1) / q->tile_n;
        a.m_fast = (a.N > a.M) ? 1 : 0;
        a.mg_tdiv = udiv_magic_of(a.m_fast ? a.tiles_m : a.tiles_n);
        a.mg_tps = udiv_magic_of((a.h_out / 16) * (a.w_out / 16));
        a.mg_tx = udiv_magic_of(a.w_out / 16);
        if (!cg_hot_ok(a)) MSD_FAIL(MSD_E_UNSUPPORTED, "conv_gemm: tile / K-tile / channel counts beyond the packed launch arguments (K tiles, channels < 65536; row tiles < 2^23; column tiles < 256)");
        rc = msd_conv_bighalo_launch(a, q->tile_n, big_code, slices, stream);
        if (rc) return rc;
        MSD_CHECK_LAUNCH();
        if (slices > 1) {
            launch_finalize(a, slices, stream);
            MSD_CHECK_LAUNCH();
        }
        return MSD_OK;
    }
    if (big) {   // 256-row macro tiles, staggered half-workgroups: conv_big.hip
        const int bbm = q->tile_m - 5000, bbn = q->tile_n;
        if (q->act == MSD_ACT_GEGLU && (msd_conv_big_nj(bbm, bbn, big_code) % 2)) MSD_FAIL(MSD_E_UNSUPPORTED, "conv_gemm: GEGLU needs x | gate fragment pairs per wave");
        a.tiles_m = (a.M + bbm - 1) / bbm;
        a.tiles_n = (a.N + bbn - 1) / bbn;
        a.m_fast = (a.N > a.M) ? 1 : 0;
        a.mg_tdiv = udiv_magic_of(a.m_fast ? a.tiles_m : a.tiles_n);
        a.mg_tps = a.mg_tx = 0;
        const bool needs_dense = q->ln_in || q->act == MSD_ACT_GEGLU || q->split_mode;
        const bool dense = !q->rowvec && !q->a2 && q->ksize == 1 && q->stride == 1 && !q->upsample && q->h_out == q->h_in && q->w_out == q->w_in;
        if (needs_dense && !dense)
            MSD_FAIL(MSD_E_UNSUPPORTED, "conv_gemm: the LayerNorm fold, GEGLU and the q|k|v^T split run on the 1x1 / Dense form only");
        if (!cg_hot_ok(a)) MSD_FAIL(MSD_E_UNSUPPORTED, "conv_gemm: tile / K-tile / channel counts beyond the packed launch arguments (K tiles, channels < 65536; row tiles < 2^23; column tiles < 256)");
        rc = msd_conv_big_launch(a, bbm, bbn, big_code, slices, dense, stream);
        if (rc) return rc;
        MSD_CHECK_LAUNCH();
        if (slices > 1) {
            launch_finalize(a, slices, stream);
            MSD_CHECK_LAUNCH();
        }
        return MSD_OK;
    }
    int cfg = -1;
    // (GEGLU pairs the fragments of a wave: the 8-wave 64x64 tile has a single one per wave)
    const int stages_req = (q->act == MSD_ACT_GEGLU && q->stages >= 10 && q->stages < 20) ? 0 : q->stages;
    for (int i = 0; i < NUM_TILE_CFGS; ++i)
        if (g_cfgs[i].bm == bm && g_cfgs[i].bn == bn && (cfg < 0 || g_cfgs[i].stages == stages_req)) cfg = i;
    // (the first entry of a tile size is its default ring depth; `stages` selects a deeper ring: more
    //  bytes in flight per CU for the weight-streaming small-M layers that run one workgroup per CU)
    if (cfg < 0)
        MSD_FAIL(MSD_E_ARG, "conv_gemm: unsupported tile %dx%d (have 128x128 128x64 64x64 64x128 256x128)", bm, bn);
    const int tiles_m = (a.M + bm - 1) / bm;
    a.tiles_n = (a.N + bn - 1) / bn;
    a.tiles_m = tiles_m;
    a.m_fast = (a.N > a.M) ? 1 : 0;   // weights are the bigger operand: keep each weight panel on one XCD
    a.mg_tdiv = udiv_magic_of(a.m_fast ? a.tiles_m : a.tiles_n);
    a.mg_tps = a.mg_tx = 0;
    dim3 grid(tiles_m * a.tiles_n, slices);
    if (!cg_hot_ok(a)) MSD_FAIL(MSD_E_UNSUPPORTED, "conv_gemm: tile / K-tile / channel counts beyond the packed launch arguments (K tiles, channels < 65536; row tiles < 2^23; column tiles < 256)");
    // 1x1 / Dense form: 32-bit byte offsets from the tensor bases
    // (the DENSE kernel carries the LayerNorm-fold consumer and no time-embedding row, the general kernel the reverse)
    const bool needs_dense = q->ln_in || q->act == MSD_ACT_GEGLU || q->split_mode;
    const bool dense = (g_conv_dense || needs_dense) && !q->rowvec && !q->a2 && q->ksize == 1 && q->stride == 1 && !q->upsample && q->h_out == q->h_in && q->w_out == q->w_in &&
                       (long long)a.M * (q->c0 > q->c1 ? q->c0 : q->c1) * 2 < (1ll << 32) - 4096 &&
                       (long long)a.N * a.K * 2 < (1ll << 32) - 4096;
    if (needs_dense && !dense)
        MSD_FAIL(MSD_E_UNSUPPORTED, "conv_gemm: the LayerNorm fold, GEGLU and the q|k|v^T split run on the 1x1 / Dense form only (ksize 1, "
                                    "stride 1, no upsampling, no rowvec, no shortcut operand)");
    switch (cfg) {
#define X(id, bm_, bn_, wgm, wgn, st, code)                                                                             \
    case id:                                                                                                            \
        if (dense)                                                                                                      \
            hipLaunchKernelGGL((conv_gemm_dma_kernel<bm_, bn_, wgm, wgn, st, true>), grid, dim3(wgm * wgn * 64),        \
                               cfg_lds(bm_, bn_, wgm * wgn * 64, st), stream, CG_HOT_ARGS(a), a);                       \
        else                                                                                                            \
            hipLaunchKernelGGL((conv_gemm_dma_kernel<bm_, bn_, wgm, wgn, st, false>), grid, dim3(wgm * wgn * 64),       \
                               cfg_lds(bm_, bn_, wgm * wgn * 64, st), stream, CG_HOT_ARGS(a), a);                       \
        break;
        MSD_TILE_CFGS(X)
#undef X
    }
    MSD_CHECK_LAUNCH();
    if (slices > 1) {
        launch_finalize(a, slices, stream);
        MSD_CHECK_LAUNCH();
    }
    return MSD_OK;
}

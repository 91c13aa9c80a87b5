// 3x3 stride-1 convolution with an LDS-staged HALO tile reused by the 9 filter taps.
//
// Same contraction, operands, epilogue and numerics as conv_gemm.hip; what changes is the A-operand
// traffic.  conv_gemm treats each (tap, 64-channel chunk) as an independent K tile and re-gathers the
// pixel rows for every tap: 9 x BM x 128 B per chunk.  That kernel is bound by the per-CU L2->LDS
// rate (~50-70 GB/s), so here a workgroup owns a SPATIAL output tile of TH x 16 pixels, loads the
// (TH+2) x 18 input halo of a chunk ONCE (23 KB for 8x16, 41 KB for 16x16) and walks the 9 taps over
// it by shifting the LDS fragment address; only the BN x 64 weight tile of each tap still streams
// per K step.  L2->LDS bytes per FLOP drop 1.7x (8x16 tile, BN=128) to 2.3x (16x16, BN=128).
//
//   LDS: halo[2][(TH+2)*18 rows x 128 B, padded to the DMA round]  +  weights ring[S][BN x 128 B]
//   pipeline per K step (chunk c, tap): counted vmcnt -> s_barrier -> issue {halo(c+1) at tap 0,
//   weights(step+2)} -> 2 x (ds_read fragments, MFMA).  Halo rows outside the image read the zero page.
//   The swizzle (16-byte chunk XOR (row>>1)&7) is keyed on the HALO row, so a fragment = 16 consecutive
//   halo rows starting anywhere is still bank-conflict free.
// Requirements (host-checked, else the generic kernel runs): ksize 3, stride 1, no upsample,
// w % 16 == 0, h % TH == 0.
#include "conv_common.h"

template <int TH, int BN, int WGM, int WGN, int S>
__global__ __launch_bounds__(WGM * WGN * 64) void conv3x3_halo_kernel(const CGArgs p) {
    constexpr int TW = 16, BM = TH * TW;
    constexpr int NW = WGM * WGN, NT = NW * 64;
    constexpr int WMT = BM / WGM, WNT = BN / WGN;
    constexpr int MI = WMT / 16, NJ = WNT / 16;           // MI = tile rows per wave
    constexpr int HW_ = TW + 2;                            // halo width (18)
    constexpr int HROWS = (TH + 2) * HW_;
    constexpr int RPP = NT / 8;                            // LDS rows written per DMA round
    constexpr int HR = (HROWS + RPP - 1) / RPP;            // DMA rounds (= instructions per thread) per halo
    constexpr int H_BYTES = HR * RPP * 128;
    constexpr int BR = BN * 8 / NT;                        // weight DMA instructions per thread per K step
    constexpr int W_BYTES = BN * 128;
    static_assert((BN * 8) % NT == 0 && (NJ % 2) == 0 && S == 3, "config");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int r = lane & 15, g = lane >> 4;

    // ---- which tile: (n tile, sample, tile row, tile column), XCD-aware order ------------------
    const int tiles_x = p.w_in / TW, tiles_y = p.h_in / TH;
    const int tps = tiles_x * tiles_y;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = p.m_fast ? tile / p.tiles_m : tile % p.tiles_n;
    const int tmi = p.m_fast ? tile % p.tiles_m : tile / p.tiles_n;
    const int b = tmi / tps;
    const int trem = tmi - b * tps;
    const int ty0 = (trem / tiles_x) * TH, tx0 = (trem - (trem / tiles_x) * tiles_x) * TW;
    const int n0 = tile_n * BN;
    const int nchunks = p.nkc;                             // 64-channel chunks of the (concatenated) input
    const int c_begin = blockIdx.y * p.nk_per;             // split-K is over chunks here
    const int c_end = min(nchunks, c_begin + p.nk_per);
    const int nkt = (c_end - c_begin) * 9;

    // ---- loader coordinates -----------------------------------------------------------------------
    const int cpos = tid & 7, lrow = tid >> 3;
    const char* zero = reinterpret_cast<const char*>(g_zero_page) + cpos * 16;
    int hpix[HR], hsrc[HR];
#pragma unroll
    for (int i = 0; i < HR; ++i) {
        const int hrow = lrow + RPP * i;
        hsrc[i] = (cpos ^ ((hrow >> 1) & 7)) * 8;
        hpix[i] = -1;
        if (hrow < HROWS) {
            const int hy = hrow / HW_, hx = hrow - hy * HW_;
            const int iy = ty0 + hy - 1, ix = tx0 + hx - 1;
            if ((unsigned)iy < (unsigned)p.h_in && (unsigned)ix < (unsigned)p.w_in) hpix[i] = (b * p.h_in + iy) * p.w_in + ix;
        }
    }
    const bf16_t* wsrc[BR];
#pragma unroll
    for (int i = 0; i < BR; ++i) {
        const int row = lrow + RPP * i;
        const int n = n0 + row;
        wsrc[i] = (n < p.N) ? (p.w + (size_t)n * p.K + (cpos ^ ((row >> 1) & 7)) * 8) : nullptr;
    }
    const uint32_t lds_wave = lds0 + (uint32_t)(wave * 8) * 128u;   // this wave's 8 rows inside a DMA round
    const int cin = p.c0 + p.c1;

    auto issue_halo = [&](int c, int buf) {
        const int ch = c * 64;
        const bf16_t* src; int csrc, coff;
        if (ch < p.c0) { src = p.a0; csrc = p.c0; coff = ch; } else { src = p.a1; csrc = p.c1; coff = ch - p.c0; }
        const uint32_t base = lds_wave + (uint32_t)buf * H_BYTES;
#pragma unroll
        for (int i = 0; i < HR; ++i) {
            const void* gp = (hpix[i] >= 0) ? static_cast<const void*>(src + (size_t)hpix[i] * csrc + coff + hsrc[i])
                                            : static_cast<const void*>(zero);
            dma16(gp, base + (uint32_t)(RPP * i) * 128u);
        }
    };
    auto issue_w = [&](int c, int tap, int stage) {
        const size_t koff = (size_t)tap * cin + (size_t)c * 64;
        const uint32_t base = lds_wave + 2u * H_BYTES + (uint32_t)stage * W_BYTES;
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const void* gp = wsrc[i] ? static_cast<const void*>(wsrc[i] + koff) : static_cast<const void*>(zero);
            dma16(gp, base + (uint32_t)(RPP * i) * 128u);
        }
    };

    f32x4 acc[NJ][MI];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // program order of the DMA queue: H(c_begin), W(0), W(1) | it: [H(next chunk) at tap 0], W(it+2)
    if (nkt > 0) {
        issue_halo(c_begin, 0);
        issue_w(c_begin, 0, 0);
        issue_w(c_begin, 1, 1);
    }
    int c = c_begin, tap = 0, stage = 0, hbuf = 0;
    bool halo_prev = false;  // a halo was issued during the previous iteration (after W(it)'s issue point)
    for (int it = 0; it < nkt; ++it) {
        // Retire W(it) (and, being older in the queue, the halo of this chunk).  Younger than W(it):
        // W(it+1) if it exists, preceded by the halo issued at iteration it-1 if there was one.
        const bool later_w = (it + 1 < nkt);
        if (!later_w) wait_vmcnt<0>();
        else if (halo_prev) wait_vmcnt<BR + HR>();
        else wait_vmcnt<BR>();
        __builtin_amdgcn_s_barrier();
        halo_prev = false;
        if (tap == 0 && c + 1 < c_end) { issue_halo(c + 1, hbuf ^ 1); halo_prev = true; }
        if (it + 2 < nkt) {
            int c2 = c, t2 = tap + 2;
            if (t2 >= 9) { t2 -= 9; c2 += 1; }
            int st = stage + 2; if (st >= S) st -= S;
            issue_w(c2, t2, st);
        }
        const int ky = tap / 3, kx = tap - ky * 3;
        const char* bW = smem + 2 * H_BYTES + stage * W_BYTES + (wn * WNT + r) * 128;
        const char* bH = smem + hbuf * H_BYTES;
        int hrow[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) hrow[i] = (wm * MI + i + ky) * HW_ + kx + r;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[MI], wf[NJ];
            const int wc = ((ks * 4 + g) ^ (r >> 1)) << 4;
#pragma unroll
            for (int i = 0; i < MI; ++i)
                af[i] = *reinterpret_cast<const bf16x8*>(bH + hrow[i] * 128 + (((ks * 4 + g) ^ ((hrow[i] >> 1) & 7)) << 4));
#pragma unroll
            for (int j = 0; j < NJ; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(bW + j * 16 * 128 + wc);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[j][i], 0, 0, 0);
        }
        if (++stage == S) stage = 0;
        if (++tap == 9) { tap = 0; ++c; hbuf ^= 1; }
    }
    int mrow[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) mrow[i] = (b * p.h_in + ty0 + wm * MI + i) * p.w_in + tx0;
    cg_epilogue<MI, NJ>(p, acc, mrow, n0 + wn * WNT, r, g);
}

// (tile height, BN, waves m x n)
#define MSD_HALO_CFGS(X) \
    X(8, 64, 2, 2)       \
    X(8, 128, 2, 4)      \
    X(16, 128, 4, 2)

template <int TH, int BN, int WGM, int WGN>
static constexpr int halo_lds() {
    constexpr int NT = WGM * WGN * 64, RPP = NT / 8, HROWS = (TH + 2) * 18, HR = (HROWS + RPP - 1) / RPP;
    return 2 * HR * RPP * 128 + 3 * BN * 128;
}

static bool g_halo_attr_done = false;
int msd_conv_halo_init() {
    if (g_halo_attr_done) return MSD_OK;
    hipError_t e = hipSuccess;
#define X(th, bn, wgm, wgn)                                                                                   \
    if (e == hipSuccess)                                                                                      \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel<th, bn, wgm, wgn, 3>),    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, halo_lds<th, bn, wgm, wgn>());
    MSD_HALO_CFGS(X)
#undef X
    if (e != hipSuccess) MSD_FAIL((int)e, "hipFuncSetAttribute(conv_halo): %s", hipGetErrorString(e));
    g_halo_attr_done = true;
    return MSD_OK;
}

// Launch for an already validated argument block; returns MSD_E_UNSUPPORTED if (th, bn) is not built.
int msd_conv_halo_launch(const CGArgs& a, int th, int bn, int slices, hipStream_t stream) {
    int rc = msd_conv_halo_init();
    if (rc) return rc;
    const int tiles = a.batch * (a.h_in / th) * (a.w_in / 16) * a.tiles_n;
    dim3 grid(tiles, slices);
#define X(th_, bn_, wgm, wgn)                                                                                             \
    if (th == th_ && bn == bn_) {                                                                                         \
        hipLaunchKernelGGL((conv3x3_halo_kernel<th_, bn_, wgm, wgn, 3>), grid, dim3(wgm * wgn * 64),                       \
                           (halo_lds<th_, bn_, wgm, wgn>()), stream, a);                                                  \
        return MSD_OK;                                                                                                    \
    }
    MSD_HALO_CFGS(X)
#undef X
    MSD_FAIL(MSD_E_UNSUPPORTED, "conv_halo: no %dx16 x %d configuration", th, bn);
}

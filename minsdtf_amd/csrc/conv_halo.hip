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
// w % 16 == 0, h % TH == 0.  A shortcut operand (a2 | a3: extra K chunks read at the output pixel) is walked behind the
// slice's main chunks (round 6).
#include "conv_common.h"

#ifdef MSD_STAMPS
extern "C" MSD_API int msd_debug_stamps_halo(unsigned long long* host_out, int count) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * (size_t)count);
}
#endif

// LDS bytes of an instance that walks a shortcut operand: what the main walk needs and no more (asking for room for five shortcut slots
// everywhere took the small-ring configurations from two workgroups per CU to one: batch-1 loop + 3.9 %)
constexpr int halo_xsc_lds(int base, int xslot) { return base >= 2 * xslot ? base : 2 * xslot; }

// wait until at most min(later, MAXL) weight tiles (BR instructions each) + optionally one halo (HR) are in flight
template <int BR, int HR, int MAXL>
__device__ __forceinline__ void halo_wait(int later, bool halo) {
    if constexpr (MAXL <= 0) {
        wait_vmcnt<0>();
    } else {
        if (later >= MAXL) {
            if (halo) wait_vmcnt<(MAXL * BR + HR < 63 ? MAXL * BR + HR : 63)>();
            else wait_vmcnt<(MAXL * BR < 63 ? MAXL * BR : 63)>();
        } else {
            halo_wait<BR, HR, MAXL - 1>(later, halo);
        }
    }
}

// TAPS = filter taps per K step (1, or 3 = one filter row): with 3 the weight ring moves a whole filter row per stage and
// there is one wait + barrier per 3 taps; inside a step the fragment reads of the next tap have no barrier between them
// and the previous tap's MFMAs (a 1-tap step measured 0.59 us for 0.16 us of MFMA work on the 8x16x80 tile: 0.32 us
// issuing fragment reads + DMAs, 0.16 us wait + barrier).  The taps of a chunk are still walked in the order 0..8, so the
// bits do not change.
// NL = loader waves behind the WGM x WGN compute waves (0: every wave stages its share of each tile, the form above).  With
// NL > 0 the compute waves' K step is barrier -> fragment reads -> MFMAs and nothing else; the loaders run the same DMA program
// (counted waits, the barrier, the issues) over a row distribution of their own and leave before the epilogue.
// Leading scalars = kernarg preload (conv_common.h CG_HOT_PARAMS): what tile mapping, halo coordinates and the first DMAs need
// (14 dwords: tiles_m | tiles_n << 23 | m_fast << 31 travel packed, range-checked by cg_hot_ok on the host)
#define HALO_HOT_PARAMS const bf16_t* hot_a0, const bf16_t* hot_w, int hot_w_in, int hot_h_in, uint32_t hot_pk_tiles, uint32_t hot_mg_tdiv, uint32_t hot_mg_tps, \
                        uint32_t hot_mg_tx, int hot_nkc, int hot_nk_per, int hot_c0, int hot_N
#define HALO_HOT_ARGS(a) (a).a0, (a).w, (a).w_in, (a).h_in, ((uint32_t)(a).tiles_m | ((uint32_t)(a).tiles_n << 23) | ((uint32_t)((a).m_fast ? 1u : 0u) << 31)), \
                         (a).mg_tdiv, (a).mg_tps, (a).mg_tx, (a).nkc, (a).nk_per, (a).c0, (a).N
// PFX = 1 (`stages` 90 + depth: 3 taps per step, ring of 3, two loader waves): the K loop is ROTATED so that
// no fragment read stands alone in front of the MFMAs it feeds.  In the form above a step is barrier -> reads of tap 0 -> {reads 1, MFMAs 0,
// reads 2, MFMAs 1, MFMAs 2}: the first tap's reads (0.3-0.4 us on the LDS port, four waves at once) and the last tap's MFMAs overlap
// nothing.  Rotated: reads 1 | MFMAs 0 | reads 2 | MFMAs 1 | wait + barrier (the NEXT filter row's weights are visible) | reads of the next
// step's tap 0 | MFMAs 2 — every read is issued in front of MFMAs that do not depend on it.  The barrier in the middle of a step orders the
// stage reuse: in front of it every wave has retired the reads of the step's own stage (lgkmcnt(0)), so that stage takes filter row it + 3
// right behind the barrier and a row keeps two steps to land.  Same taps in the same order: the same bits.
// XSC = the instance that walks a shortcut operand behind the main chunks (its own instantiation: the plain 3x3 convs keep the register
// allocation and schedule they were tuned with - with the shortcut steps compiled into the one kernel the batch-1 loop measured 0.3 % slower)
template <int TH, int BN, int WGM, int WGN, int S, int TAPS = 1, int NL = 0, int PFX = 0, bool XSC = false>
__global__ __launch_bounds__((WGM * WGN + NL) * 64) void conv3x3_halo_kernel(HALO_HOT_PARAMS, const CGArgs p) {
    const int hot_tiles_m = (int)(hot_pk_tiles & 0x7FFFFFu), hot_tiles_n = (int)((hot_pk_tiles >> 23) & 0xFFu), hot_m_fast = (int)(hot_pk_tiles >> 31);
    constexpr int TW = 16, BM = TH * TW;
    constexpr int NW = WGM * WGN, NT = (NL ? NL : NW) * 64;   // NT = threads that stage
    constexpr int WMT = BM / WGM, WNT = BN / WGN;
    constexpr int MI = WMT / 16, NJ = WNT / 16;           // MI = tile rows per wave
    constexpr int HW_ = TW + 2;                            // halo width (18)
    constexpr int HROWS = (TH + 2) * HW_;
    constexpr int RPP = NT / 8;                            // LDS rows written per DMA round
    constexpr int HR = (HROWS + RPP - 1) / RPP;            // DMA rounds (= instructions per thread) per halo
    constexpr int H_BYTES = HR * RPP * 128;
    constexpr int BNP = (BN + RPP - 1) / RPP * RPP;        // weight rows as staged (BN = 80: padded to the DMA round)
    constexpr int BR1 = BNP * 8 / NT;                      // weight DMA instructions per thread per tap
    constexpr int BR = BR1 * TAPS;                         // ... per K step
    constexpr int W1_BYTES = BNP * 128, W_BYTES = W1_BYTES * TAPS;
    constexpr int SPC = 9 / TAPS;                          // K steps per 64-channel chunk
    static_assert(TAPS == 1 || TAPS == 3, "taps per step");
    static_assert(WNT % 16 == 0 && WMT % 16 == 0, "config");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;

    const int tid0 = threadIdx.x, lane = tid0 & 63;
    const int wave0 = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const bool loader = NL > 0 && wave0 >= NW;            // (wave-uniform)
    const bool computes = NL == 0 || !loader;
    const int tid = NL ? tid0 - NW * 64 : tid0;           // staging thread index (loaders only when NL > 0)
    const int wave = NL ? wave0 - NW : wave0;             // ... and staging wave index
    const int wm = wave0 / WGN, wn = wave0 % WGN;
    const int r = lane & 15, g = lane >> 4;
    MSD_STAMP(0);

    // ---- which tile: (n tile, sample, tile row, tile column), XCD-aware order ------------------
    const int tiles_x = hot_w_in / TW, tiles_y = hot_h_in / TH;
    const int tps = tiles_x * tiles_y;
    const int tile = xcd_remap(blockIdx.x, hot_tiles_m * hot_tiles_n);   // (= gridDim.x)
    const int tdiv = hot_m_fast ? hot_tiles_m : hot_tiles_n;   // divisions by host-prepared magic numbers
    const int tq = udiv_magic(tile, tdiv, hot_mg_tdiv), tr = tile - tq * tdiv;
    const int tile_n = hot_m_fast ? tq : tr;
    const int tmi = hot_m_fast ? tr : tq;
    const int b = udiv_magic(tmi, tps, hot_mg_tps);
    const int trem = tmi - b * tps;
    const int tyi = udiv_magic(trem, tiles_x, hot_mg_tx);
    const int ty0 = tyi * TH, tx0 = (trem - tyi * tiles_x) * TW;
    const int n0 = tile_n * BN;
    const int nchunks = hot_nkc;                             // 64-channel chunks of the (concatenated) input
    const int c_begin = blockIdx.y * hot_nk_per;             // split-K is over chunks here
    const int c_end = min(nchunks, c_begin + hot_nk_per);
    const int nkt = (c_end - c_begin) * SPC;

    // ---- loader coordinates -----------------------------------------------------------------------
    const int cpos = tid & 7, lrow = tid >> 3;
    // An out-of-image halo row reads the zero page: the select is mask arithmetic on the 64-bit address (32-bit offset
    // from the tensor base), and the weight rows use scalar base + 32-bit offset with out-of-range rows clamped (their
    // columns are never stored): no divergent branches in the loader (the ternary form made hipcc wrap every piece
    // in an exec-mask branch).
    int hpix[HR], hsrc[HR];   // pixel index of the staged halo row (-1: outside the image), swizzled source chunk (elements)
#pragma unroll
    for (int i = 0; i < HR; ++i) {
        const int hrow = lrow + RPP * i;
        const int hy = hrow / HW_, hx = hrow - hy * HW_;
        const int iy = ty0 + hy - 1, ix = tx0 + hx - 1;
        const bool ok = hrow < HROWS && (unsigned)iy < (unsigned)hot_h_in && (unsigned)ix < (unsigned)hot_w_in;
        // (coordinates clamped into the image instead of a conditional: no branch; `ok` only selects the sentinel)
        const int pix = (b * hot_h_in + min(max(iy, 0), hot_h_in - 1)) * hot_w_in + min(max(ix, 0), hot_w_in - 1);
        hpix[i] = ok ? pix : -1;
        hsrc[i] = (cpos ^ ((hrow >> 1) & 7)) * 8;
    }
    uint32_t woff[BR1];
#pragma unroll
    for (int i = 0; i < BR1; ++i) {
        const int row = lrow + RPP * i;
        woff[i] = (uint32_t)min(n0 + row, hot_N - 1) * p.w_rs + (uint32_t)((cpos ^ ((row >> 1) & 7)) * 16);
    }
    const uint32_t lds_wave = lds0 + (uint32_t)(wave * 8) * 128u;   // this wave's 8 rows inside a DMA round
    const uint64_t zaddr = (uint64_t)(reinterpret_cast<const char*>(g_zero_page) + cpos * 16);

    auto issue_halo = [&](int c, int buf) {
        const int ch = c * 64;
        const bool first = ch < hot_c0;                                  // wave-uniform: which tensor of the concat
        const uint64_t sb = (uint64_t)(first ? hot_a0 : p.a1);
        const int csrc = first ? hot_c0 : p.c1, coff = first ? ch : ch - hot_c0;
        const uint32_t base = __builtin_amdgcn_readfirstlane(lds_wave + (uint32_t)buf * H_BYTES);
#pragma unroll
        for (int i = 0; i < HR; ++i) {
            const uint32_t m32 = (uint32_t)(~hpix[i] >> 31);             // all ones for a pixel inside the image (hpix >= 0)
            const uint64_t m64 = ((uint64_t)m32 << 32) | m32;
            const uint32_t off = (uint32_t)(hpix[i] * csrc + coff + hsrc[i]) * 2u;
            const uint64_t a = ((sb + off) & m64) | (zaddr & ~m64);
            dma16(reinterpret_cast<const void*>(a), base + (uint32_t)(RPP * i) * 128u);
        }
    };
    auto issue_w = [&](int c, int tg, int stage) {   // tg = tap (TAPS = 1) or filter row (TAPS = 3)
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            const uint32_t koff = (uint32_t)((tg * TAPS + t) * nchunks + c) * p.w_ks;
            // (readfirstlane: with the tap loop unrolled hipcc no longer proves this sum wave-uniform and M0 needs an SGPR)
            const uint32_t base = __builtin_amdgcn_readfirstlane(lds_wave + 2u * H_BYTES + (uint32_t)stage * W_BYTES + (uint32_t)t * W1_BYTES);
#pragma unroll
            for (int i = 0; i < BR1; ++i) dma16s(hot_w, woff[i] + koff, base + (uint32_t)(RPP * i) * 128u);
        }
    };

    f32x4 acc[NJ][MI];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Program order of the DMA queue: H(c_begin), W(0) .. W(S-2) | iteration it: [H(next chunk) at tap 0],
    // W(it+S-1).  With one workgroup per CU (batch 1-2) a K step costs (L2 latency) / (tiles in flight):
    // S = 3 measured ~0.7 us per step against ~0.13 us of MFMA work, hence the deeper rings.
    static_assert(S >= 3 && S - 1 <= SPC, "ring depth");
    const bool stages_tiles = NL == 0 || loader;   // this wave issues DMAs (and waits for them)
    if (nkt > 0 && stages_tiles) {
        issue_halo(c_begin, 0);
#pragma unroll
        for (int s = 0; s < S - 1; ++s)
            if (s < nkt) issue_w(c_begin, s, s);
    }
    MSD_STAMP(1);
    int c = c_begin, tap = 0, stage = 0, hbuf = 0;
    int cw = c_begin, tw = S - 1, sw = S - 1;   // (chunk, tap, stage) of the next weight tile to issue
    if (tw >= SPC) { tw -= SPC; ++cw; }
    int since_halo = 1 << 20;                   // iterations since a halo was issued
    if constexpr (PFX && TAPS == 1) {
        // One tap per step (`stages` 150 + depth): the fragments of tap it + 1 are read under the MFMAs of tap it.  Per step: lgkmcnt(0) (the
        // fragments of tap it — the last reads of stage it % S — are in registers) -> counted wait for tap it + 1's weights -> barrier ->
        // DMA of tap it + S into the stage just handed back -> reads of tap it + 1 -> MFMAs of tap it.  The ring holds S taps in
        // flight or landed instead of S - 1 (all S stages are filled in the prologue), so the prefetch costs no ring depth.
        const int rw = cg_wrow(r);
        auto read_tap_at = [&](int hb, int stg, int tp, bf16x8 (&af)[2][MI], bf16x8 (&wf)[2][NJ]) {
            const int ky = (tp * 11) >> 5, kx = tp - ky * 3;   // (tp / 3 for tp < 9)
            const char* bH = smem + hb * H_BYTES;
            const char* bW = smem + 2 * H_BYTES + stg * W_BYTES + (wn * WNT + rw) * 128;
            int hrow[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) hrow[i] = (wm * MI + i + ky) * HW_ + kx + r;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int wc = ((ks * 4 + g) ^ (rw >> 1)) << 4;
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    af[ks][i] = *reinterpret_cast<const bf16x8*>(bH + hrow[i] * 128 + (((ks * 4 + g) ^ ((hrow[i] >> 1) & 7)) << 4));
#pragma unroll
                for (int j = 0; j < NJ; ++j) wf[ks][j] = *reinterpret_cast<const bf16x8*>(bW + j * 16 * 128 + wc);
            }
        };
        auto mfma_frags = [&](const bf16x8 (&af)[2][MI], const bf16x8 (&wf)[2][NJ]) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int i = 0; i < MI; ++i)
                        acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][j], af[ks][i], acc[j][i], 0, 0, 0);
        };
        bf16x8 fa[2][MI], wa[2][NJ], fb[2][MI], wb[2][NJ];
        if (stages_tiles && S - 1 < nkt) {   // the ring's last stage too
            issue_w(cw, tw, sw);
            if (++tw == SPC) { tw = 0; ++cw; }
            if (++sw == S) sw = 0;
        }
        if (stages_tiles) halo_wait<BR, HR, S - 1>(min(S - 1, nkt - 1), false);   // tap 0 and the first halo: everything younger may fly on
        __builtin_amdgcn_s_barrier();
        MSD_STAMP(2);
        if (computes && nkt > 0) read_tap_at(0, 0, 0, fa, wa);
        auto step = [&](int it, bf16x8 (&xa)[2][MI], bf16x8 (&xw)[2][NJ], bf16x8 (&ya)[2][MI], bf16x8 (&yw)[2][NJ]) {
#ifdef MSD_STAMPS
            if (it == (nkt >> 1)) MSD_STAMP(5);
#endif
            if (computes) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // tap it + 1 (and, at a chunk's last tap, the next chunk's halo) has landed; younger: taps it + 2 .. it + S - 1 and a halo
            // issued within the last S - 2 steps
            if (stages_tiles) halo_wait<BR, HR, S - 2>(max(0, min(S - 2, nkt - 2 - it)), since_halo <= S - 2);
            __builtin_amdgcn_s_barrier();
            if (stages_tiles) {
                if (tap == 0 && c + 1 < c_end) { issue_halo(c + 1, hbuf ^ 1); since_halo = 0; }
                if (it + S < nkt) {
                    issue_w(cw, tw, sw);
                    if (++tw == SPC) { tw = 0; ++cw; }
                    if (++sw == S) sw = 0;
                }
            }
            ++since_halo;
            if (++stage == S) stage = 0;
            if (++tap == SPC) { tap = 0; ++c; hbuf ^= 1; }
            if (computes) {
                if (it + 1 < nkt) read_tap_at(hbuf, stage, tap, ya, yw);
                mfma_frags(xa, xw);
            }
        };
        for (int it = 0; it < nkt; it += 2) {
            step(it, fa, wa, fb, wb);
            if (it + 1 < nkt) step(it + 1, fb, wb, fa, wa);
        }
    }
    if constexpr (PFX && TAPS == 3) {
        static_assert(TAPS == 3 && S == 3, "rotated loop: 3 taps per step, ring of 3 filter rows");
        const int rw = cg_wrow(r);   // weight rows enter the MFMA in the order 0-3, 8-11, 4-7, 12-15 (cg_epilogue)
        auto read_tap_at = [&](int hb, int stg, int ky, int kx, bf16x8 (&af)[2][MI], bf16x8 (&wf)[2][NJ]) {
            const char* bH = smem + hb * H_BYTES;
            const char* bW = smem + 2 * H_BYTES + stg * W_BYTES + kx * W1_BYTES + (wn * WNT + rw) * 128;
            int hrow[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) hrow[i] = (wm * MI + i + ky) * HW_ + kx + r;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int wc = ((ks * 4 + g) ^ (rw >> 1)) << 4;
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    af[ks][i] = *reinterpret_cast<const bf16x8*>(bH + hrow[i] * 128 + (((ks * 4 + g) ^ ((hrow[i] >> 1) & 7)) << 4));
#pragma unroll
                for (int j = 0; j < NJ; ++j) wf[ks][j] = *reinterpret_cast<const bf16x8*>(bW + j * 16 * 128 + wc);
            }
        };
        auto mfma_frags = [&](const bf16x8 (&af)[2][MI], const bf16x8 (&wf)[2][NJ]) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int i = 0; i < MI; ++i)
                        acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][j], af[ks][i], acc[j][i], 0, 0, 0);
        };
        bf16x8 fa[2][MI], wa[2][NJ], fb[2][MI], wb[2][NJ];
        // The ring's third stage is filled in the prologue too: a stage is handed back at the barrier of the step that reads it (its
        // last reads are retired by the lgkmcnt(0) in front of that barrier), so filter row it + 3 goes out in step it and a row has
        // two steps to land, as in the lock-step form.
        if (stages_tiles && S - 1 < nkt) {
            issue_w(cw, tw, sw);
            if (++tw == SPC) { tw = 0; ++cw; }
            if (++sw == S) sw = 0;
        }
        // filter row 0 (+ the first halo) visible: only rows 1 and 2 are younger
        if (stages_tiles) halo_wait<BR, HR, 2>(min(2, nkt - 1), false);
        __builtin_amdgcn_s_barrier();
        MSD_STAMP(2);
        if (computes && nkt > 0) read_tap_at(0, 0, 0, 0, fa, wa);
        bool halo_prev = false;   // a halo was issued in the previous step (it is younger than the row this step waits for)
        // one step; X holds tap 0 on entry and the step leaves the next step's tap 0 in Y
        auto step = [&](int it, bf16x8 (&xa)[2][MI], bf16x8 (&xw)[2][NJ], bf16x8 (&ya)[2][MI], bf16x8 (&yw)[2][NJ]) {
#ifdef MSD_STAMPS
            if (it == (nkt >> 1)) MSD_STAMP(5);
#endif
            if (computes) {
                read_tap_at(hbuf, stage, tap, 1, ya, yw);
                mfma_frags(xa, xw);
                read_tap_at(hbuf, stage, tap, 2, xa, xw);
                mfma_frags(ya, yw);
            }
            // every read of THIS step's stage has landed (tap 2's fragments wait in registers for their MFMAs behind the barrier) ...
            if (computes) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // ... and filter row it + 1 (with the halo its chunk starts with) has landed; younger in the queue and allowed to fly on:
            // row it + 2 and, in a chunk's second step, the next chunk's halo issued one step ago
            if (stages_tiles) halo_wait<BR, HR, 1>(it + 2 < nkt ? 1 : 0, halo_prev);
            __builtin_amdgcn_s_barrier();
            if (stages_tiles) {   // this step's stage and, at a chunk's first step, the other halo buffer are free now
                halo_prev = tap == 0 && c + 1 < c_end;
                if (halo_prev) issue_halo(c + 1, hbuf ^ 1);
                if (it + S < nkt) {
                    issue_w(cw, tw, sw);
                    if (++tw == SPC) { tw = 0; ++cw; }
                    if (++sw == S) sw = 0;
                }
            }
            if (++stage == S) stage = 0;
            if (++tap == SPC) { tap = 0; ++c; hbuf ^= 1; }
            if (computes) {
                if (it + 1 < nkt) read_tap_at(hbuf, stage, tap, 0, ya, yw);
                mfma_frags(xa, xw);
            }
        };
        for (int it = 0; it < nkt; it += 2) {
            step(it, fa, wa, fb, wb);
            if (it + 1 < nkt) step(it + 1, fb, wb, fa, wa);
        }
    }
    for (int it = 0; !PFX && it < nkt; ++it) {
        // Retire W(it) (and, being older in the queue, the halo of this chunk).  Younger than W(it):
        // W(it+1) .. W(it+S-2) as far as they exist, plus the halo if one was issued in iterations
        // it-S+2 .. it-1 (loads complete in order, so "at most N outstanding" retires everything older).
#ifdef MSD_STAMPS
        const bool probe = it == 20 || (nkt <= 20 && it == nkt - 2);   // one K step under the microscope (slots 10..15)
        if (probe) MSD_STAMP(10);
#endif
        if (stages_tiles) halo_wait<BR, HR, S - 2>(min(S - 2, nkt - 1 - it), since_halo <= S - 2);
#ifdef MSD_STAMPS
        if (probe) MSD_STAMP(11);
#endif
        __builtin_amdgcn_s_barrier();
#ifdef MSD_STAMPS
        if (it == 0) MSD_STAMP(2);
        if (it == (nkt >> 1)) MSD_STAMP(5);
        if (probe) MSD_STAMP(12);
#endif
        // fragments of this K step's first tap, then the DMA issue of the tiles ahead (it runs under the LDS latency), then
        // per tap: MFMAs (the next tap's fragment reads are independent of them: no barrier inside a step)
        const char* bH = smem + hbuf * H_BYTES;
        auto read_tap = [&](int t, bf16x8 (&af)[2][MI], bf16x8 (&wf)[2][NJ]) {
            const int tp = tap * TAPS + t;
            const int ky = TAPS == 3 ? tap : tp / 3, kx = TAPS == 3 ? t : tp - ky * 3;
            const int rw = cg_wrow(r);   // weight rows enter the MFMA in the order 0-3, 8-11, 4-7, 12-15 (cg_epilogue)
            const char* bW = smem + 2 * H_BYTES + stage * W_BYTES + t * W1_BYTES + (wn * WNT + rw) * 128;
            int hrow[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) hrow[i] = (wm * MI + i + ky) * HW_ + kx + r;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int wc = ((ks * 4 + g) ^ (rw >> 1)) << 4;
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    af[ks][i] = *reinterpret_cast<const bf16x8*>(bH + hrow[i] * 128 + (((ks * 4 + g) ^ ((hrow[i] >> 1) & 7)) << 4));
#pragma unroll
                for (int j = 0; j < NJ; ++j) wf[ks][j] = *reinterpret_cast<const bf16x8*>(bW + j * 16 * 128 + wc);
            }
        };
        auto mfma_tap = [&](const bf16x8 (&af)[2][MI], const bf16x8 (&wf)[2][NJ]) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int i = 0; i < MI; ++i)
                        acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][j], af[ks][i], acc[j][i], 0, 0, 0);
        };
        bf16x8 af[2][MI], wf[2][NJ];
        if (computes) read_tap(0, af, wf);
        if (stages_tiles) {
            if (tap == 0 && c + 1 < c_end) { issue_halo(c + 1, hbuf ^ 1); since_halo = 0; }
            if (it + S - 1 < nkt) {
                issue_w(cw, tw, sw);
                if (++tw == SPC) { tw = 0; ++cw; }
                if (++sw == S) sw = 0;
            }
        }
        ++since_halo;
#ifdef MSD_STAMPS
        if (probe) {
            MSD_STAMP(13);                                           // fragment reads + DMA issued
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            MSD_STAMP(14);                                           // fragments in registers
        }
#endif
        if (computes) {
            if constexpr (TAPS == 1) {
                mfma_tap(af, wf);
            } else {
                bf16x8 af2[2][MI], wf2[2][NJ];
                read_tap(1, af2, wf2);
                mfma_tap(af, wf);
                read_tap(2, af, wf);
                mfma_tap(af2, wf2);
                mfma_tap(af, wf);
            }
        }
#ifdef MSD_STAMPS
        if (probe) {
            asm volatile("v_mov_b32 %0, %0" : "+v"(acc[NJ - 1][MI - 1][3]));   // last MFMA of the step has retired
            MSD_STAMP(15);
        }
#endif
        if (++stage == S) stage = 0;
        if (++tap == SPC) { tap = 0; ++c; hbuf ^= 1; }
    }
    // ---- shortcut operand (ResBlock: conv2(h) + conv_shortcut(x) as ONE contraction, diffusion_model.py:34-38,50): after the slice's main
    //      chunks, its share of the 64-channel chunks of a2 | a3, read at the OUTPUT pixel - one K step each on the plain TH x 16-pixel tile.
    //      The same walk, the same dealing of the shortcut chunks over the slices and therefore the same bits as conv_bighalo_kernel's
    //      (conv_big.hip), which takes these layers from about 128 of its 16 x 16-pixel workgroups on; here they run at ONE image per GPU
    //      without leaving that numerics class.  The whole LDS of the workgroup (the halo buffers and the weight ring are free by now) is a
    //      ring of XS slots {A tile: BM rows x 128 B | the chunk's BN x 64 weight tile}, XS - 1 chunks in flight behind a counted wait: with
    //      two slots and one step of lead every step stood for a whole HBM latency (the weights are cold in the loop) and the form lost 0.3 %
    //      of the batch-1 loop against the tile kernels it beats by 10 % in isolation.
    const int nxc = XSC ? p.nk - p.nk_main : 0;
    if constexpr (XSC) {
        const int eps = (nxc + p.nslices - 1) / p.nslices;
        const int e0 = min(nxc, (int)blockIdx.y * eps), e1 = min(nxc, e0 + eps);
        const int cx3 = p.K - p.nk_main * 64 - p.c2;   // channels of a3
        constexpr int AR = BM / RPP;                  // DMA rounds of the A tile
        constexpr int XSLOT = BM * 128 + W1_BYTES;    // bytes per ring slot
        constexpr int LDS_ALL = halo_xsc_lds(2 * H_BYTES + S * W_BYTES, XSLOT);
        constexpr int XS = LDS_ALL / XSLOT < 6 ? LDS_ALL / XSLOT : 6;
        constexpr int XI = AR + BR1;                  // DMA instructions per thread and chunk
        static_assert(BM % RPP == 0 && XS >= 2 && (XS - 2) * XI <= 63, "shortcut ring: rows per DMA round, slots, countable waits");
        int xpix[AR], xsrc[AR];
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int q = lrow + RPP * i;             // row of the tile = pixel (q >> 4, q & 15)
            xpix[i] = (b * hot_h_in + ty0 + (q >> 4)) * hot_w_in + tx0 + (q & 15);
            xsrc[i] = (cpos ^ ((q >> 1) & 7)) * 16;   // (swizzle keyed on the LDS row, as everywhere)
        }
        auto issue_extra = [&](int e, int slot) {
            const int ce = e * 64;
            const bool first = ce < p.c2;
            const uint64_t sb = (uint64_t)(first ? p.a2 : p.a3) + (uint64_t)(uint32_t)((first ? ce : ce - p.c2) * 2);
            const uint32_t cs2 = (uint32_t)(first ? p.c2 : cx3) * 2u;
            const uint32_t baseA = __builtin_amdgcn_readfirstlane(lds_wave + (uint32_t)slot * (uint32_t)XSLOT);
#pragma unroll
            for (int i = 0; i < AR; ++i)
                dma16(reinterpret_cast<const void*>(sb + ((uint64_t)(uint32_t)xpix[i] * cs2 + (uint32_t)xsrc[i])), baseA + (uint32_t)(RPP * i) * 128u);
            const uint32_t koff = (uint32_t)(p.nk_main + e) * p.w_ks;
            const uint32_t baseW = baseA + (uint32_t)(BM * 128);
#pragma unroll
            for (int i = 0; i < BR1; ++i) dma16s(hot_w, woff[i] + koff, baseW + (uint32_t)(RPP * i) * 128u);
        };
        if (computes) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the main walk's last fragments are in registers ...
        if (stages_tiles) wait_vmcnt<0>();                                   // ... and nothing of it is still on its way into LDS
        __builtin_amdgcn_s_barrier();
        if (stages_tiles) {
#pragma unroll
            for (int k = 0; k < XS - 1; ++k)
                if (e0 + k < e1) issue_extra(e0 + k, k);
        }
        const int rw = cg_wrow(r);
        int slot = 0, islot = XS - 1;   // slot of chunk e; slot the next issue goes to (= the one chunk e - 1 was read from)
        for (int e = e0; e < e1; ++e) {
            // chunk e has landed; younger in the queue and free to fly on: the chunks e + 1 .. e + XS - 2 as far as they exist
            if (stages_tiles) halo_wait<XI, 0, XS - 2>(min(XS - 2, e1 - 1 - e), false);
            if (computes) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // ... for every wave; the readers of chunk e - 1's slot have their fragments
            if (stages_tiles && e + XS - 1 < e1) issue_extra(e + XS - 1, islot);
            if (computes) {
                const char* bA = smem + slot * XSLOT;
                const char* bW = bA + BM * 128 + (wn * WNT + rw) * 128;
                bf16x8 af[2][MI], wf[2][NJ];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int wc = ((ks * 4 + g) ^ (rw >> 1)) << 4;
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        const int q = (wm * MI + i) * 16 + r;
                        af[ks][i] = *reinterpret_cast<const bf16x8*>(bA + q * 128 + (((ks * 4 + g) ^ ((q >> 1) & 7)) << 4));
                    }
#pragma unroll
                    for (int j = 0; j < NJ; ++j) wf[ks][j] = *reinterpret_cast<const bf16x8*>(bW + j * 16 * 128 + wc);
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int i = 0; i < MI; ++i)
                            acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][j], af[ks][i], acc[j][i], 0, 0, 0);
            }
            if (++slot == XS) slot = 0;
            if (++islot == XS) islot = 0;
        }
    }
    MSD_STAMP(3);
    if (NL > 0 && loader) return;
    int mrow[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) mrow[i] = (b * p.h_in + ty0 + wm * MI + i) * p.w_in + tx0;
    cg_epilogue<MI, NJ, true, false>(p, acc, mrow, n0 + wn * WNT, r, g);   // (a spatial tile lies inside one sample)
#ifdef MSD_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MSD_STAMP(4);
#endif
}

// (tile height, BN, waves m x n).  BN = 80 exists for the N = 320 / 640 layers at batch 1-2: 8x16-pixel
// tiles x 80 channels give exactly 256 workgroups at 64x64 (one per CU) where 64-wide tiles give 320
// (a second, quarter-full round) and 128-wide ones 192.
// Last column: weight-ring depth.  3 is the default of every tile; the deeper rings (selected with
// MsdConvGemm.stages) are for launches that put a single workgroup on a CU.
// (Tried: 64 pixels x 64 channels per wave on 4 waves — 16x16 x 64 and 8x16 x 128 tiles, fewer fragment ds_reads
//  per MFMA — and a deeper ring for 16x16 x 128: none beat the tiles below on any UNet / VAE shape.)
// Variant 1 (tile_m 2128): the same tile on 8 waves — two waves per SIMD even when the launch puts a
// single workgroup on a CU, so one wave's LDS-DMA issue stalls and LDS latency hide under the other's MFMAs.
// `stages` 30 + depth: 3 taps (one filter row) per K step with a ring of `depth` rows (8x16 tiles x 64 / 80 channels: the
// ring holds depth x 3 weight tiles).
#define MSD_HALO_CFGS(X)      \
    X(8, 64, 2, 2, 3, 0, 1, 0)   \
    X(8, 128, 2, 4, 3, 0, 1, 0)  \
    X(16, 128, 4, 2, 3, 0, 1, 0) \
    X(8, 80, 4, 1, 3, 0, 1, 0)   \
    X(16, 80, 4, 1, 3, 0, 1, 0)  \
    X(8, 64, 2, 2, 8, 0, 1, 0)   \
    X(8, 128, 2, 4, 6, 0, 1, 0)  \
    X(8, 80, 4, 1, 8, 0, 1, 0)   \
    X(16, 80, 4, 1, 5, 0, 1, 0)  \
    X(8, 64, 4, 2, 3, 1, 1, 0)   \
    X(8, 80, 8, 1, 3, 1, 1, 0)   \
    X(8, 64, 2, 2, 33, 0, 3, 0)  \
    X(8, 64, 2, 2, 34, 0, 3, 0)  \
    X(8, 80, 4, 1, 33, 0, 3, 0)  \
    X(8, 64, 4, 2, 33, 1, 3, 0)  \
    X(8, 80, 4, 1, 63, 0, 3, 2)  \
    X(8, 64, 2, 2, 63, 0, 3, 2)  \
    X(8, 80, 4, 1, 93, 0, 3, 2)  \
    X(8, 64, 2, 2, 93, 0, 3, 2)  \
    X(8, 64, 2, 2, 153, 0, 1, 0) \
    X(8, 64, 2, 2, 158, 0, 1, 0) \
    X(8, 128, 2, 4, 153, 0, 1, 0) \
    X(8, 128, 2, 4, 156, 0, 1, 0) \
    X(16, 128, 4, 2, 153, 0, 1, 0) \
    X(8, 80, 4, 1, 158, 0, 1, 0) \
    X(16, 80, 4, 1, 153, 0, 1, 0) \
    X(16, 80, 4, 1, 155, 0, 1, 0)

template <int TH, int BN, int WGM, int WGN, int SC, int TAPS, int NL, bool XSC = false>
static constexpr int halo_lds() {
    constexpr int S = SC % 30;   // (SC = stages code: 30 + depth for the 3-taps-per-step form, 60 + depth: the same with 2 loader waves)
    constexpr int NT = (NL ? NL : WGM * WGN) * 64, RPP = NT / 8, HROWS = (TH + 2) * 18, HR = (HROWS + RPP - 1) / RPP;
    constexpr int BNP = (BN + RPP - 1) / RPP * RPP;
    constexpr int base = 2 * HR * RPP * 128 + S * TAPS * BNP * 128;
    constexpr int bytes = XSC ? halo_xsc_lds(base, TH * 16 * 128 + BNP * 128) : base;
    static_assert(bytes <= 160 * 1024, "LDS budget");
    return bytes;
}

static bool g_halo_attr_done = false;
int msd_conv_halo_init() {
    if (g_halo_attr_done) return MSD_OK;
    hipError_t e = hipSuccess;
#define X(th, bn, wgm, wgn, st, var, taps, nl)                                                                         \
    if (e == hipSuccess)                                                                                              \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel<th, bn, wgm, wgn, st % 30, taps, nl, (st >= 90), false>), \
                                hipFuncAttributeMaxDynamicSharedMemorySize, halo_lds<th, bn, wgm, wgn, st, taps, nl>());                           \
    if (e == hipSuccess)                                                                                              \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel<th, bn, wgm, wgn, st % 30, taps, nl, (st >= 90), true>), \
                                hipFuncAttributeMaxDynamicSharedMemorySize, halo_lds<th, bn, wgm, wgn, st, taps, nl, true>());
    MSD_HALO_CFGS(X)
#undef X
    if (e != hipSuccess) MSD_FAIL((int)e, "hipFuncSetAttribute(conv_halo): %s", hipGetErrorString(e));
    g_halo_attr_done = true;
    return MSD_OK;
}

// Launch for an already validated argument block; returns MSD_E_UNSUPPORTED if (th, bn) is not built.
int msd_conv_halo_launch(const CGArgs& a, int th, int bn, int stages, int variant, int slices, hipStream_t stream) {
    int rc = msd_conv_halo_init();
    if (rc) return rc;
    if (a.tiles_m <= 0 || a.tiles_m >= (1 << 23) || a.tiles_n <= 0 || a.tiles_n >= 256)
        MSD_FAIL(MSD_E_UNSUPPORTED, "conv_halo: tile counts beyond the packed launch arguments (row tiles < 2^23, column tiles < 256)");
    const int tiles = a.batch * (a.h_in / th) * (a.w_in / 16) * a.tiles_n;
    dim3 grid(tiles, slices);
    // `stages` picks the ring depth if that variant is built, otherwise the tile's default (3)
    bool have = false;
#define X(th_, bn_, wgm, wgn, st, var, taps, nl) have = have || (th == th_ && bn == bn_ && stages == st && variant == var);
    MSD_HALO_CFGS(X)
#undef X
    if (!have) {   // unknown ring depth -> the tile's default; unknown 8-wave variant -> the 4-wave tile
        stages = 3;
        have = false;
#define X(th_, bn_, wgm, wgn, st, var, taps, nl) have = have || (th == th_ && bn == bn_ && stages == st && variant == var);
        MSD_HALO_CFGS(X)
#undef X
        if (!have) variant = 0;
    }
#define X(th_, bn_, wgm, wgn, st, var, taps, nl)                                                                          \
    if (th == th_ && bn == bn_ && stages == st && variant == var) {                                                       \
        if (a.nk > a.nk_main)   /* a shortcut operand: the instance with the extra steps */                              \
            hipLaunchKernelGGL((conv3x3_halo_kernel<th_, bn_, wgm, wgn, st % 30, taps, nl, (st >= 90), true>), grid, dim3((wgm * wgn + nl) * 64), \
                               (halo_lds<th_, bn_, wgm, wgn, st, taps, nl, true>()), stream, HALO_HOT_ARGS(a), a);                          \
        else                                                                                                              \
            hipLaunchKernelGGL((conv3x3_halo_kernel<th_, bn_, wgm, wgn, st % 30, taps, nl, (st >= 90), false>), grid, dim3((wgm * wgn + nl) * 64), \
                               (halo_lds<th_, bn_, wgm, wgn, st, taps, nl>()), stream, HALO_HOT_ARGS(a), a);                                \
        return MSD_OK;                                                                                                    \
    }
    MSD_HALO_CFGS(X)
#undef X
    MSD_FAIL(MSD_E_UNSUPPORTED, "conv_halo: no %dx16 x %d configuration (variant %d)", th, bn, variant);
}

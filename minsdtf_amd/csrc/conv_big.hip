// Implicit-GEMM convolution / dense layer on 256-row macro tiles for M >= 8192 ("big" form; round 5).
//
// Same contraction, K walk (64-channel tiles of one tap, ascending), MFMA operand placement and epilogue as conv_gemm.hip, so the
// same bits (tests compare them): which form runs a layer is a per-(shape, batch) speed choice inside the layer's numerics class.
// What changes is the shape of the K loop.  conv_gemm's tiles top out at 64 x 64 per wave and 256 x 128 per workgroup: at batch >= 4
// its K step is bound by the L2 -> LDS bytes a CU can pull (48 KB per K tile in 0.79-0.88 us against 0.43 us of MFMA
// time), and every wave of a workgroup is in the same phase (all read fragments, then all multiply).  Here:
//   * macro tile 256 x 256 (128 FLOP per staged byte instead of 85), 256 x 160 for N = 320 / 640 / 960, 256 x 128, 128 x 256; 8 waves,
//     each 128 x 64 (64 x 80, 64 x 64): 128 (80, 64) accumulator registers;
//   * the wave's K tile is cut into PH phases (2 x 2 quadrants of the wave tile in snake order, or row parts): a phase reads only the
//     fragments its quadrant needs (48-64 operand registers live, not 96), then issues its 8-20 MFMAs under s_setprio;
//   * the two half-workgroups (waves 0-3 / 4-7 = the two waves of every SIMD) run ONE BARRIER APART: while one half issues its MFMA
//     cluster the other issues fragment reads, address generation and LDS-DMAs.  Two raw s_barriers per phase keep the halves in
//     that lock step (MI355X_MICROARCH.md, Two waves per SIMD: matrix beside memory is the pairing that pays);
//   * operands go L2 -> LDS by LDS-DMA in NEED ORDER (the rows phase 0 reads first), ~2 DMAs per thread and phase, one whole K tile
//     (NB = 2 buffers) or two (NB = 3) ahead; a counted s_waitcnt vmcnt(N) in front of the barrier that precedes the phase that
//     reads a round retires exactly the rounds that phase needs - vmcnt never reaches 0 inside the loop;
//   * the LDS image holds rows in need order ([A part 0 | A part 1 ..] and [W part 0 | W part 1]), 128-byte rows, 16-byte chunk
//     index XOR (row >> 1) & 7 on the SOURCE address (LDS-DMA writes lane-linear), undone in the ds_read_b128 fragment reads.
// Time is counted in SLOTS (barrier to barrier).  Half 0 runs LOAD(P) in slot 2P and MFMA(P) in slot 2P + 1, half 1 one slot later.
// Hazards (BigGeo::hazards_ok holds them for every configuration built):
//   RAW  every wave waits for the rounds phase P + 1 reads at the end of its LOAD(P) slot; the later half's LOAD(P) slot ends with the
//        barrier in front of the earlier half's LOAD(P + 1), so both halves' DMAs have landed and are visible when either half reads;
//   WAR  a round of a buffer is re-issued no earlier than 3 slots after the later half's read of its previous occupant was ISSUED
//        (that read has returned at the lgkmcnt(0) which heads the slot in between).
// No ln_out (LayerNorm-fold producer) epilogue: those layers have K = N <= 1280 and stay on the small tiles; the host refuses.
#include <type_traits>
#include "conv_common.h"

#ifdef MSD_STAMPS
// (tools/big_stamps.py, `make stamps` library only.)  Beside the workgroup timeline of conv_common.h: waves 0 and 4 (the first wave of
// either half) sum the shader clock between fixed points of every phase - 0 fragment reads issued, 1 DMAs issued, 2 counted wait
// passed, 3 barrier passed, 4 MFMAs issued (fragments landed), 5 second barrier passed - rows [workgroup & 1023][half][8].
static __device__ unsigned long long g_bstamps[1024 * 2 * 8];
extern "C" MSD_API int msd_debug_stamps_big(unsigned long long* host_out, int count) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * (size_t)count);
}
extern "C" MSD_API int msd_debug_phase_stamps_big(unsigned long long* host_out, int count) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_bstamps), sizeof(unsigned long long) * (size_t)count);
}
// (stamps 0-3 sit among ds_reads that are still in flight: their values are only read after the lgkmcnt(0) of BSTAMP_END and must stay
//  in the SGPRs the instruction named - no spill in between; stamps 4-6 wait for themselves: the MFMA slot waits lgkmcnt(0) anyway)
#define BSTAMP(i) do { if ((i) < 4) asm volatile("s_memtime %0" : "=s"(bst[i]) : : "memory"); else asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(bst[i]) : : "memory"); } while (0)
#define BSTAMP_END()                                                              \
    do {                                                                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                        \
        _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) bsum[i_] += bst[i_ + 1] - bst[i_]; \
        ++bsum[6];                                                                \
    } while (0)
#else
#define BSTAMP(i)
#define BSTAMP_END()
#endif

// ---- geometry of a configuration (all compile-time) --------------------------------------------------------------------------
template <int BM, int BN, int WGM, int WGN, int IH, int JH, int NB>
struct BigGeo {
    static_assert(WGM * WGN == 8, "8 waves");
    static_assert(BM % (WGM * 16) == 0 && BN % (WGN * 16) == 0 && BM % 64 == 0 && BN % 32 == 0, "tile");
    static_assert((JH == 1 && (IH == 2 || IH == 4)) || (JH == 2 && IH == 2), "phase structure");
    static constexpr int WMT = BM / WGM, WNT = BN / WGN, MI = WMT / 16, NJ = WNT / 16;
    static_assert(MI % IH == 0 && NJ >= JH && (MI <= 4 || MI % 4 == 0), "parts");
    static constexpr int IHS = MI / IH;                              // row fragments per A part
    static constexpr int JS0 = (NJ + JH - 1) / JH, JS1 = NJ - JS0;   // column fragments of W part 0 / 1
    static constexpr int PH = IH * JH;                               // phases per K tile
    static constexpr int APR = WGM * IHS * 16;                       // rows of one A part
    static constexpr int WP0 = WGN * JS0 * 16;                       // rows of W part 0
    static_assert(APR % 64 == 0, "an A part is a whole number of 64-row rounds");
    static constexpr int ARP = APR / 64, AR = BM / 64;               // rounds per A part / per tile
    static constexpr int WRF = BN / 64, WRH = (BN % 64) / 32, WR = WRF + WRH;   // full W rounds, a half round (32 rows) when BN % 64 == 32
    static constexpr int G = AR + WR;                                // DMA rounds (= vector-memory operations per thread) per K tile
    static constexpr int LEADT = NB - 1;                             // K tiles in flight ahead of the one being read
    static constexpr int A_BYTES = BM * 128, ST_BYTES = (BM + BN) * 128;
    static constexpr int LDS = NB * ST_BYTES;
    static_assert(LDS <= 160 * 1024, "LDS");
    // need order: A part 0, all of W (part 0 then part 1), A parts 1 ..
    static constexpr bool seq_is_a(int s) { return s < ARP || s >= ARP + WR; }
    static constexpr int seq_round(int s) { return s < ARP ? s : (s < ARP + WR ? s - ARP : s - WR); }
    // rounds issued in phase p, and up to and including it
    static constexpr int g_of(int p) { return G / PH + (p < G % PH ? 1 : 0); }
    static constexpr int cum(int p) { int c = 0; for (int q = 0; q <= p; ++q) c += g_of(q); return c; }
    static constexpr int issue_phase(int s) { for (int p = 0; p < PH; ++p) if (s < cum(p)) return p; return PH - 1; }
    // position of A round `s` among the A rounds its phase issues, and the most A rounds any phase issues
    static constexpr int a_index_in_phase(int s) { int n = 0; for (int q = (issue_phase(s) == 0 ? 0 : cum(issue_phase(s) - 1)); q < s; ++q) n += seq_is_a(q) ? 1 : 0; return n; }
    static constexpr int max_a_per_phase() {
        int best = 0;
        for (int p = 0; p < PH; ++p) { int n = 0; for (int q = (p == 0 ? 0 : cum(p - 1)); q < cum(p); ++q) n += seq_is_a(q) ? 1 : 0; best = n > best ? n : best; }
        return best;
    }
    // highest sequence index phase p (or an earlier one) reads
    static constexpr int need(int p) {
        if (JH == 2) return p == 0 ? ARP + (WP0 - 1) / 64 : (p == 1 ? ARP + WR - 1 : G - 1);
        return ARP * (p + 1) + WR - 1;
    }
    // phase in which the rows of sequence element s are read (the later one, for a W round that straddles both parts)
    static constexpr int read_phase(int s) {
        if (s < ARP) return 0;
        if (s < ARP + WR) return (JH == 2 && (s - ARP + 1) * 64 > WP0) ? 1 : 0;
        const int part = 1 + (s - ARP - WR) / ARP;
        return JH == 2 ? 2 : part;
    }
    static constexpr bool hazards_ok() {
        for (int s = 0; s < G; ++s) {
            int np = 0;
            while (need(np) < s) ++np;
            if (LEADT * PH + np - issue_phase(s) < 3) return false;      // RAW lead: >= 3 phases between issue and first read
            // WAR: issue slot 2 (PH t + is) >= 2 (PH (t + LEADT - NB) + lr) + 1 + 3  <=>  is >= lr - PH + 2
            if (issue_phase(s) < read_phase(s) - PH + 2) return false;
        }
        return true;
    }
    static_assert(hazards_ok(), "issue schedule violates the lead / reuse rules");
};

__device__ __forceinline__ void big_wait(int n) {   // s_waitcnt vmcnt(n), n wave-uniform in [0, 31]: the loop's tail and the prologue only
    switch (n) {
#define W1(x) case x: wait_vmcnt<x>(); break;
#define W8(b) W1(b) W1(b + 1) W1(b + 2) W1(b + 3) W1(b + 4) W1(b + 5) W1(b + 6) W1(b + 7)
        W8(0) W8(8) W8(16) W8(24)
#undef W8
#undef W1
        default: wait_vmcnt<32>(); break;   // (stricter than asked: n > 32 never occurs with G <= 8, LEADT <= 2)
    }
}

// LD = loader form: 0 general (tap / padding / stride / concat / shortcut-operand addresses), 1 dense (1x1 / Dense: a row is one pixel's
// channel vector), 2 general with nearest x2 upsampling (the tap's pixel is not "row + constant": its own address arithmetic)
template <int BM, int BN, int WGM, int WGN, int IH, int JH, int NB, int LD>
__global__ __launch_bounds__(512) void conv_big_kernel(CG_HOT_PARAMS, const CGArgs p) {
    CG_HOT_UNPACK;
    constexpr bool DENSE = LD == 1, UPS = LD == 2;
    using Geo = BigGeo<BM, BN, WGM, WGN, IH, JH, NB>;
    constexpr int MI = Geo::MI, NJ = Geo::NJ, IHS = Geo::IHS, JS0 = Geo::JS0, JS1 = Geo::JS1, PH = Geo::PH;
    constexpr int AR = Geo::AR, WRF = Geo::WRF, WR = Geo::WR, G = Geo::G, LEADT = Geo::LEADT;
    constexpr int WMT = Geo::WMT, WNT = Geo::WNT, APR = Geo::APR, WP0 = Geo::WP0;
    constexpr int A_BYTES = Geo::A_BYTES, ST_BYTES = Geo::ST_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave - wm * WGN;
    const int grp = wave >> 2;   // waves w and w + 4 share a SIMD: the two halves are the two waves of every SIMD
    const int r = lane & 15, g = lane >> 4;
    MSD_STAMP(0);
    const int tile = xcd_remap(blockIdx.x, hot_tiles_m * hot_tiles_n);
    const int tdiv = hot_m_fast ? hot_tiles_m : hot_tiles_n;
    const int tq = udiv_magic(tile, tdiv, hot_mg_tdiv), tr = tile - tq * tdiv;
    const int tile_n = hot_m_fast ? tq : tr;
    const int tile_m = hot_m_fast ? tr : tq;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kt_begin = blockIdx.y * hot_nk_per;
    const int kt_end = min(hot_nk, kt_begin + hot_nk_per);
    const int nkt = kt_end - kt_begin;

    // ---- loader coordinates.  A round k: LDS row (need order) 64 k + 8 wave + (lane >> 3), chunk position lane & 7 ----------------
    const int cpos = lane & 7, lr8 = lane >> 3;
    // swizzled SOURCE chunk in bytes; the same for every full round: (row >> 1) & 7 with row = 64 k + 8 wave + lr8
    const uint32_t asrc2 = (uint32_t)((cpos ^ ((4 * wave + (lane >> 4)) & 7)) * 16);
    const char* zero = reinterpret_cast<const char*>(g_zero_page) + cpos * 16;
    const int Hl = UPS ? 2 * p.h_in : p.h_in;
    const int Wl = UPS ? 2 * p.w_in : p.w_in;
    // general form: st0 = pixel index of the row's tap (0, 0) (may be negative at the image border), st1 = 9-bit mask of the taps that
    // lie inside the image; upsampling convs: st0 = the sample's first pixel, st1 = (ay + 2) | (ax + 2) << 16 (0: row past M).
    // DENSE: st0 / st1 = byte offsets of the row from a0 / a1.
    int st0[AR], st1[AR];
#pragma unroll
    for (int k = 0; k < AR; ++k) {
        const int q = 64 * k + 8 * wave + lr8;                     // need-order row
        const int part = q / APR, rem = q - part * APR;
        const int wmr = rem / (IHS * 16), rr = rem - wmr * (IHS * 16);
        const int m = m0 + wmr * WMT + part * (IHS * 16) + rr;
        if constexpr (DENSE) {
            const uint32_t mc = (uint32_t)min(m, hot_M - 1);       // rows past M re-read the last row: never stored
            st0[k] = (int)(mc * (uint32_t)hot_c0 * 2u + asrc2);
            st1[k] = (int)(mc * (uint32_t)hot_c1 * 2u + asrc2);
        } else if (m < hot_M) {
            const int b = udiv_magic(m, p.hw_out, p.mg_hw);
            const int remp = m - b * p.hw_out;
            const int y = udiv_magic(remp, p.w_out, p.mg_w);
            const int x = remp - y * p.w_out;
            const int ay = y * p.stride - p.pad, ax = x * p.stride - p.pad;
            if constexpr (UPS) {
                st0[k] = b * p.h_in * p.w_in;
                st1[k] = (ay + 2) | ((ax + 2) << 16);
            } else {
                st0[k] = b * p.h_in * p.w_in + ay * p.w_in + ax;
                // bit 3 ky + kx = tap (ky, kx) lies inside the image: the outer product of a row mask and a column mask
                const int xm = ((unsigned)ax < (unsigned)Wl ? 1 : 0) | ((unsigned)(ax + 1) < (unsigned)Wl ? 2 : 0) | ((unsigned)(ax + 2) < (unsigned)Wl ? 4 : 0);
                int mask = (unsigned)ay < (unsigned)Hl ? xm : 0;
                if (p.ksize == 3) mask |= ((unsigned)(ay + 1) < (unsigned)Hl ? xm << 3 : 0) | ((unsigned)(ay + 2) < (unsigned)Hl ? xm << 6 : 0);
                else mask &= 1;
                st1[k] = mask;
            }
        } else {
            st0[k] = 0;
            st1[k] = 0;   // no tap valid (upsampling form: ay = ax = -2, every tap outside the image)
        }
    }
    // W round k: row 64 k + 8 wave + lr8 of the W region (the half round: 64 k + 4 wave + lr8, lanes < 32)
    uint32_t woff[WR];
#pragma unroll
    for (int k = 0; k < WR; ++k) {
        const bool half = k >= WRF;
        const int q = 64 * k + (half ? 4 : 8) * wave + lr8;
        const int part = (JS1 > 0 && q >= WP0) ? 1 : 0;
        const int qq = q - part * WP0;
        const int js = part ? JS1 : JS0;
        const int wnr = qq / (js * 16), rr = qq - wnr * (js * 16);
        const int n = n0 + wnr * WNT + part * JS0 * 16 + rr;
        woff[k] = (uint32_t)min(n, hot_N - 1) * hot_w_rs + (uint32_t)((cpos ^ ((q >> 1) & 7)) * 16);   // columns past N re-read the last weight row (never stored)
    }
    const uint32_t ldsA_wave = lds0 + (uint32_t)(wave * 8) * 128u;
    const uint32_t ldsW_wave = lds0 + A_BYTES + (uint32_t)(wave * 8) * 128u;
    const uint32_t ldsWh_wave = lds0 + A_BYTES + (uint32_t)(wave * 4) * 128u;

    // per-K-tile scalars of the A loader (wave-uniform), prepared when the tile's first round is issued
    struct TileA { uint64_t sb; uint32_t csrc2; int dpix; int tapbit; int ky, kx; bool first; uint32_t cb; int kt; };   // kt: the WEIGHT tile's index
    auto tile_a = [&](int kt) {
        TileA t;
        t.kt = kt;
        if constexpr (DENSE) {
            const int c = kt * 64;
            t.first = c < hot_c0;
            t.cb = (uint32_t)(t.first ? c : c - hot_c0) * 2u;
            t.sb = 0; t.csrc2 = 0; t.dpix = 0; t.tapbit = 0; t.ky = t.kx = 0;
            return t;
        } else {
            const bool extra = kt >= p.nk_main;   // shortcut operand (a2 | a3) read at the output pixel: the centre / only tap
            int tap, cidx;
            if (p.kmajor) {   // chunk-major walk (3x3, no shortcut operand: host): K step kt = tap kt % 9 of chunk kt / 9
                cidx = udiv_magic(kt, 9, 0x1C71C71Cu);
                tap = kt - cidx * 9;
                t.kt = tap * p.nkc + cidx;   // the weight matrix keeps its (tap, channel) column order
            } else {
                tap = extra ? 0 : udiv_magic(kt, p.nkc, p.mg_nkc);
                cidx = kt - tap * p.nkc;
            }
            const int c = extra ? (kt - p.nk_main) * 64 : cidx * 64;
            const int ky = p.ksize == 3 ? (extra ? p.pad : (tap * 11) >> 5) : 0;
            const int kx = p.ksize == 3 ? (extra ? p.pad : tap - ky * 3) : 0;
            const int cA = extra ? p.c2 : hot_c0;
            const bool first = c < cA;
            const uint64_t base = (uint64_t)(extra ? (first ? p.a2 : p.a3) : (first ? hot_a0 : hot_a1));
            const int csrc = first ? cA : (extra ? p.K - p.nk_main * 64 - p.c2 : hot_c1), coff = first ? c : c - cA;
            t.sb = base + (uint64_t)(uint32_t)(coff * 2);
            t.csrc2 = (uint32_t)csrc * 2u;
            t.dpix = ky * p.w_in + kx;
            t.tapbit = 1 << (ky * 3 + kx);
            t.ky = ky; t.kx = kx; t.first = first; t.cb = 0;
            return t;
        }
    };
    // source of A round k of the K tile described by `t`: a 64-bit address (general forms; the zero page for a tap outside the image)
    // or a 32-bit byte offset from a0 / a1 (dense).  Computed one phase ahead of its DMA, inside the MFMA slot (see `phase`).
    auto addr_a_round = [&](const TileA& t, int k) -> uint64_t {
        if constexpr (DENSE) {
            return (uint64_t)((uint32_t)(t.first ? st0[k] : st1[k]) + t.cb);
        } else {
            uint32_t off;
            bool ok;
            if constexpr (UPS) {   // nearest x2: the tap's pixel is ((y - 1 + ky) >> 1, (x - 1 + kx) >> 1)
                int iy = (st1[k] & 0xFFFF) - 2 + t.ky, ix = (int)((uint32_t)st1[k] >> 16) - 2 + t.kx;
                ok = ((unsigned)iy < (unsigned)Hl) && ((unsigned)ix < (unsigned)Wl);
                iy = min(max(iy, 0), Hl - 1) >> 1; ix = min(max(ix, 0), Wl - 1) >> 1;
                off = (uint32_t)(st0[k] + iy * p.w_in + ix) * t.csrc2 + asrc2;
            } else {            // pixel = row's tap-(0,0) pixel + a per-tile constant; validity = one bit of the row's mask
                ok = (st1[k] & t.tapbit) != 0;
                off = __umul24((uint32_t)(st0[k] + t.dpix), t.csrc2) + asrc2;   // (pixels < 2^24, row bytes < 2^24: host-checked; a masked lane may hold garbage)
            }
            // branch-free select (a conditional expression becomes an exec-masked block, which cuts the MFMA slot's basic block in two
            // and keeps the scheduler from spreading this arithmetic over the MFMA gaps)
            const uint32_t m32 = ok ? 0xFFFFFFFFu : 0u;
            const uint64_t m64 = ((uint64_t)m32 << 32) | m32;
            return ((t.sb + off) & m64) | ((uint64_t)zero & ~m64);
        }
    };
    auto fire_a_round = [&](const TileA& t, uint64_t a, uint32_t lds_dst) {
        if constexpr (DENSE) dma16s(t.first ? hot_a0 : hot_a1, (uint32_t)a, lds_dst);
        else dma16(reinterpret_cast<const void*>(a), lds_dst);
    };
    auto issue_a_round = [&](const TileA& t, int k, uint32_t lds_dst) { fire_a_round(t, addr_a_round(t, k), lds_dst); };
    // one round of the need-order sequence (S compile-time) of K tile described by `t` into buffer `ib`
    constexpr int PA = Geo::max_a_per_phase();   // A rounds a phase issues at most: their sources are prepared one phase ahead
    uint64_t pa[PA > 0 ? PA : 1];
    auto issue_seq = [&](auto S_, const TileA& t, int ib, auto PRE_) {   // PRE: the A sources were prepared (the loop; the prologue computes them in place)
        constexpr int S = decltype(S_)::value;
        constexpr int k = Geo::seq_round(S);
        const uint32_t boff = (uint32_t)ib * (uint32_t)ST_BYTES + (uint32_t)k * 8192u;
        if constexpr (Geo::seq_is_a(S)) {
            if constexpr (decltype(PRE_)::value) fire_a_round(t, pa[Geo::a_index_in_phase(S)], ldsA_wave + boff);
            else issue_a_round(t, k, ldsA_wave + boff);
        } else if constexpr (k < WRF) {
            dma16s(hot_w, woff[k] + (uint32_t)t.kt * hot_w_ks, ldsW_wave + boff);
        } else {
            dma16sm(hot_w, woff[k] + (uint32_t)t.kt * hot_w_ks, ldsWh_wave + boff, 0xFFFFFFFFull);
        }
    };
    auto issue_range = [&](auto LO_, auto HI_, const TileA& t, int ib, auto PRE_) {   // rounds [LO, HI) of the sequence
        constexpr int LO = decltype(LO_)::value, HI = decltype(HI_)::value;
        if constexpr (LO < HI) {
            issue_seq(std::integral_constant<int, LO>{}, t, ib, PRE_);
            if constexpr (LO + 1 < HI) issue_seq(std::integral_constant<int, LO + 1>{}, t, ib, PRE_);
            if constexpr (LO + 2 < HI) issue_seq(std::integral_constant<int, LO + 2>{}, t, ib, PRE_);
            if constexpr (LO + 3 < HI) issue_seq(std::integral_constant<int, LO + 3>{}, t, ib, PRE_);
            if constexpr (LO + 4 < HI) issue_seq(std::integral_constant<int, LO + 4>{}, t, ib, PRE_);
            if constexpr (LO + 5 < HI) issue_seq(std::integral_constant<int, LO + 5>{}, t, ib, PRE_);
            if constexpr (LO + 6 < HI) issue_seq(std::integral_constant<int, LO + 6>{}, t, ib, PRE_);
            if constexpr (LO + 7 < HI) issue_seq(std::integral_constant<int, LO + 7>{}, t, ib, PRE_);
            static_assert(HI - LO <= 8, "rounds per call");
        }
    };
    // sources of the A rounds phase Q issues, for the K tile described by `t`
    auto prep_phase = [&](auto Q_, const TileA& t) {
        constexpr int Q = decltype(Q_)::value;
        constexpr int LO = Q == 0 ? 0 : Geo::cum(Q - 1), HI = Geo::cum(Q);
#pragma unroll
        for (int sq = LO; sq < HI; ++sq)
            if (Geo::seq_is_a(sq)) pa[Geo::a_index_in_phase(sq)] = addr_a_round(t, Geo::seq_round(sq));
    };

    // ---- accumulators in chunks of EC row fragments (the epilogue runs once per chunk: conv_wreg.hip) ---------------------------
    // (EC = 2 here, 4 in conv_wreg: with 256 registers per wave the epilogue's operand loads for four row fragments spill 240-350 registers)
    constexpr int EC = 2, EH = MI / EC;
    f32x4 acc[EH][NJ][EC];
#pragma unroll
    for (int h = 0; h < EH; ++h)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < EC; ++i) acc[h][j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- prologue: K tiles 0 .. LEADT - 1 whole ---------------------------------------------------------------------------------
#pragma unroll
    for (int tt = 0; tt < LEADT; ++tt)
        if (tt < nkt) {
            const TileA t = tile_a(kt_begin + tt);
            issue_range(std::integral_constant<int, 0>{}, std::integral_constant<int, G>{}, t, tt, std::false_type{});
        }
    MSD_STAMP(1);
    big_wait((G - 1 - Geo::need(0)) + (min(nkt, LEADT) - 1) * G);
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();   // the second half runs one slot behind the first

    // ---- fragment addressing ----------------------------------------------------------------------------------------------------
    const int rw = cg_wrow(r);   // weight rows enter the MFMA in the order 0-3, 8-11, 4-7, 12-15 (cg_epilogue)
    const int aoffs = (wm * IHS * 16 + r) * 128;                       // + part * APR * 128 + ii * 2048
    const int woffs0 = A_BYTES + (wn * JS0 * 16 + rw) * 128;           // + jj * 2048
    const int woffs1 = A_BYTES + (WP0 + wn * JS1 * 16 + rw) * 128;
    const int ca0 = ((g ^ (r >> 1)) << 4), ca1 = (((4 + g) ^ (r >> 1)) << 4);
    const int cw0 = ((g ^ (rw >> 1)) << 4), cw1 = (((4 + g) ^ (rw >> 1)) << 4);
    bf16x8 af[2][IHS], wf0[2][JS0], wf1[2][JS1 > 0 ? JS1 : 1];
    auto load_a = [&](int part, int rb) {
        const char* b = smem + rb * ST_BYTES + aoffs + part * (APR * 128);
#pragma unroll
        for (int ii = 0; ii < IHS; ++ii) {
            af[0][ii] = *reinterpret_cast<const bf16x8*>(b + ii * 2048 + ca0);
            af[1][ii] = *reinterpret_cast<const bf16x8*>(b + ii * 2048 + ca1);
        }
    };
    auto load_w0 = [&](int rb) {
        const char* b = smem + rb * ST_BYTES + woffs0;
#pragma unroll
        for (int jj = 0; jj < JS0; ++jj) {
            wf0[0][jj] = *reinterpret_cast<const bf16x8*>(b + jj * 2048 + cw0);
            wf0[1][jj] = *reinterpret_cast<const bf16x8*>(b + jj * 2048 + cw1);
        }
    };
    auto load_w1 = [&](int rb) {
        const char* b = smem + rb * ST_BYTES + woffs1;
#pragma unroll
        for (int jj = 0; jj < JS1; ++jj) {
            wf1[0][jj] = *reinterpret_cast<const bf16x8*>(b + jj * 2048 + cw0);
            wf1[1][jj] = *reinterpret_cast<const bf16x8*>(b + jj * 2048 + cw1);
        }
    };
    // MFMAs of (A part ip, W part jp): ks outermost, so every accumulator sees the K tile's two halves in ascending order
    auto mfma_part = [&](auto IP_, auto JP_) {
        constexpr int ip = decltype(IP_)::value, jp = decltype(JP_)::value;
        constexpr int js = jp ? JS1 : JS0;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int jj = 0; jj < js; ++jj)
#pragma unroll
                for (int ii = 0; ii < IHS; ++ii) {
                    const int i = ip * IHS + ii, j = jp * JS0 + jj;
                    if constexpr (jp == 0)
                        acc[i / EC][j][i % EC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[ks][jj], af[ks][ii], acc[i / EC][j][i % EC], 0, 0, 0);
                    else
                        acc[i / EC][j][i % EC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[ks][jj], af[ks][ii], acc[i / EC][j][i % EC], 0, 0, 0);
                }
    };

    // ---- K loop -------------------------------------------------------------------------------------------------------------------
#ifdef MSD_STAMPS
    unsigned long long bst[7], bsum[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
    int rb = 0, ib = LEADT % NB;   // buffer being read / being filled
    TileA nt = tile_a(kt_begin + min(LEADT, max(nkt - 1, 0))), ntn = nt;   // K tile being issued / the next one
    prep_phase(std::integral_constant<int, 0>{}, nt);
    auto phase = [&](auto P_, int rem) {
        constexpr int P = decltype(P_)::value;
        BSTAMP(0);
        // -- LOAD slot: this phase's fragments, then this phase's share of the K tile LEADT ahead, then the wait for the next phase's rounds
        if constexpr (JH == 2) {
            if constexpr (P == 0) { load_w0(rb); load_a(0, rb); }
            if constexpr (P == 1) load_w1(rb);
            if constexpr (P == 2) load_a(1, rb);
        } else {
            if constexpr (P == 0) load_w0(rb);
            load_a(P, rb);
        }
        __builtin_amdgcn_sched_barrier(0);
        BSTAMP(1);
        if (rem >= LEADT) {
            constexpr int LO = P == 0 ? 0 : Geo::cum(P - 1), HI = Geo::cum(P);
            issue_range(std::integral_constant<int, LO>{}, std::integral_constant<int, HI>{}, nt, ib, std::true_type{});
        }
        BSTAMP(2);
        if constexpr (P + 1 < PH) {
            if constexpr (Geo::need(P + 1) > Geo::need(P)) {
                constexpr int base = G - 1 - Geo::need(P + 1);
                if (rem >= LEADT) wait_vmcnt<(base + (LEADT - 1) * G + Geo::cum(P) < 63 ? base + (LEADT - 1) * G + Geo::cum(P) : 63)>();
                else big_wait(base + rem * G);
            }
        } else {
            constexpr int base = G - 1 - Geo::need(0);
            if (rem >= LEADT) wait_vmcnt<(base + (LEADT - 1) * G < 63 ? base + (LEADT - 1) * G : 63)>();
            else if (rem >= 1) big_wait(base + (rem - 1) * G);
        }
        BSTAMP(3);
        __builtin_amdgcn_s_barrier();
        BSTAMP(4);
        // -- MFMA slot.  The address arithmetic of the NEXT phase's A rounds rides in it: a MFMA holds the issue port 8 cycles of its 16, the
        //    VALU work of two sources fits the gaps, and the LOAD slot (which the other half's MFMAs wait behind) is left with bare DMAs
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        if constexpr (P + 1 < PH) prep_phase(std::integral_constant<int, P + 1>{}, nt);
        else prep_phase(std::integral_constant<int, 0>{}, ntn);
        if constexpr (JH == 2) {
            if constexpr (P == 0) mfma_part(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            if constexpr (P == 1) mfma_part(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
            if constexpr (P == 2) mfma_part(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
            if constexpr (P == 3) mfma_part(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
        } else {
            mfma_part(std::integral_constant<int, P>{}, std::integral_constant<int, 0>{});
        }
        // interleave: one MFMA, then up to two of the address-arithmetic VALU instructions (left alone hipcc puts all of them in front
        // of the first MFMA, where they cost their full issue time with the matrix pipe idle)
#pragma unroll
        for (int q = 0; q < 24; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x006, 2, 0);
        }
        // (pin: the sources are "used" here, or hipcc sinks their arithmetic into the LOAD slot's issue block, next to the DMAs)
#pragma unroll
        for (int q = 0; q < (PA > 0 ? PA : 1); ++q) asm volatile("" : "+v"(pa[q]));
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        BSTAMP(5);
        __builtin_amdgcn_s_barrier();
        BSTAMP(6);
        BSTAMP_END();
    };
    for (int t = 0; t < nkt; ++t) {
        const int rem = nkt - 1 - t;
        if (rem > LEADT) ntn = tile_a(kt_begin + t + 1 + LEADT);
#ifdef MSD_STAMPS
        if (t == 0) MSD_STAMP(2);
        if (t == (nkt >> 1)) MSD_STAMP(5);
#endif
        phase(std::integral_constant<int, 0>{}, rem);
        phase(std::integral_constant<int, 1>{}, rem);
        if constexpr (PH == 4) {
            phase(std::integral_constant<int, 2>{}, rem);
            phase(std::integral_constant<int, 3>{}, rem);
        }
        if (++rb == NB) rb = 0;
        if (++ib == NB) ib = 0;
        nt = ntn;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();   // (the first half's extra barrier: both halves have passed the same number)
    MSD_STAMP(3);
#ifdef MSD_STAMPS
    if ((wave & 3) == 0 && lane == 0) {
        unsigned long long* dst = g_bstamps + ((size_t)(blockIdx.x & 1023) * 2 + grp) * 8;
#pragma unroll
        for (int i = 0; i < 7; ++i) dst[i] = bsum[i];
    }
#endif

    auto epilogue_chunk = [&](auto H_) {   // (explicit instances: left as a loop, hipcc does not unroll it and acc[h] goes to scratch)
        constexpr int h = decltype(H_)::value;
        int mrow[EC];
#pragma unroll
        for (int i = 0; i < EC; ++i) mrow[i] = m0 + wm * WMT + (h * EC + i) * 16;
        cg_epilogue<EC, NJ, false, DENSE>(p, acc[h], mrow, n0 + wn * WNT, r, g, reinterpret_cast<float*>(smem), wn, WGN, wm * WMT + h * EC * 16, BM, tile_n);
    };
    epilogue_chunk(std::integral_constant<int, 0>{});
    if constexpr (EH > 1) epilogue_chunk(std::integral_constant<int, 1>{});
    if constexpr (EH > 2) epilogue_chunk(std::integral_constant<int, 2>{});
    if constexpr (EH > 3) epilogue_chunk(std::integral_constant<int, 3>{});
    static_assert(EH <= 4 && MI % EC == 0, "epilogue chunks");
#ifdef MSD_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores have left the wave
    MSD_STAMP(4);
#endif
}

// ---- halo-image variant: 3x3 / stride 1 / pad 1 convs (optionally behind a nearest x2 upsampling) on 16 x 16-pixel tiles -------------
// The tap-shifted tiles of the kernel above re-stage every input pixel nine times per 64-channel chunk (9 x 32 KB of A per chunk beside
// 9 x 20 KB of weights at BN = 160); here a chunk's 18 x 18-pixel halo is staged ONCE (41.5 KB) and the nine taps read it at shifted rows,
// as conv_halo.hip does - with this file's K loop around it: 256 x BN macro tile, two half-workgroups one barrier apart, two phases per
// K step (= one tap of one chunk; the wave's image rows in two parts), counted waits.  Vector-memory operations per K step and thread:
// the WR weight rounds of the step LEAD ahead plus ONE halo-stream operation (round `tap` of the NEXT chunk's halo for taps 0-5 - five
// 64-row rounds and a 16-row one - and a 1 KB zero-page round into never-read rows otherwise: every step issues the same count, so the
// waits are compile-time constants).  K order = chunk-major (conv_halo's: for every chunk its nine taps), fragments and epilogue are
// conv_halo's: the same bits (tests).  Hazards, in the slot count of the header:
//   W ring   W(kt + LEAD) goes into the buffer W(kt - 1) was read from in phase 0 of step kt - 1 (later half: slot 4 kt - 3); first issued in
//            slot 4 kt: 3 slots.  RAW: the wait that ends step kt retires W(kt + 1) - issued LEAD steps earlier - and everything older;
//   halo     chunk c + 1's rounds are issued in phase 1 of taps 0-5 of chunk c into the buffer chunk c - 1 was last read from in phase 1 of
//            its tap 8 (later half: 3 slots before the earlier half's phase 1 of tap 0); they are older than W(first step of c + 1), which is
//            issued at tap 9 - LEAD >= 6, so the wait in front of that step retires them.
template <int BN, int WGM, int WGN, int NBW>
__global__ __launch_bounds__(512) void conv_bighalo_kernel(CG_HOT_PARAMS, const CGArgs p) {
    CG_HOT_UNPACK;
    constexpr int BM = 256, HW_ = 18, HROWS = 18 * 18, HPAD = 336;     // staged halo rows: 5 rounds of 64 + one of 16 (324 real)
    static_assert(WGM * WGN == 8 && BN % 32 == 0 && BN % (WGN * 16) == 0, "tile");
    constexpr int WMT = BM / WGM, WNT = BN / WGN, MI = WMT / 16, NJ = WNT / 16, IHS = MI / 2;
    static_assert(MI % 2 == 0 && MI * WGM == 16, "a wave owns MI image rows of the 16 x 16 tile, read in two parts");
    constexpr int H_BYTES = HPAD * 128, W_BYTES = BN * 128;
    constexpr int WRF = BN / 64, WRH = (BN % 64) / 32, WR = WRF + WRH, W0 = (WR + 1) / 2;   // weight rounds per step / those phase 0 issues
    constexpr int LEAD = NBW - 1;
    static_assert(LEAD >= 1 && LEAD <= 3, "weight steps in flight (the halo RAW rule needs <= 3)");
    static_assert(2 * H_BYTES + NBW * W_BYTES <= 160 * 1024, "LDS");
    constexpr int NWAIT = LEAD + WR * (LEAD - 1);                      // operations younger than the next step's last weight round
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave - wm * WGN;
    const int grp = wave >> 2;
    const int r = lane & 15, g = lane >> 4;
    MSD_STAMP(0);
    const int ups = p.upsample ? 1 : 0;
    const int Hl = p.h_in << ups, Wl = p.w_in << ups;                  // the image the 3x3 window slides over (= the output image)
    const int tiles_x = Wl >> 4, tps = tiles_x * (Hl >> 4);
    const int tile = xcd_remap(blockIdx.x, hot_tiles_m * hot_tiles_n);
    const int tdiv = hot_m_fast ? hot_tiles_m : hot_tiles_n;
    const int tq = udiv_magic(tile, tdiv, hot_mg_tdiv), tr = tile - tq * tdiv;
    const int tile_n = hot_m_fast ? tq : tr;
    const int tmi = hot_m_fast ? tr : tq;
    const int b = udiv_magic(tmi, tps, p.mg_tps);
    const int trem = tmi - b * tps;
    const int tyi = udiv_magic(trem, tiles_x, p.mg_tx);
    const int ty0 = tyi * 16, tx0 = (trem - tyi * tiles_x) * 16;
    const int n0 = tile_n * BN;
    const int c_begin = blockIdx.y * hot_nk_per;                       // split-K is over 64-channel chunks (host: nk_per in chunks)
    const int nch = min(p.nkc, c_begin + hot_nk_per) - c_begin;
    const int nkt = nch * 9;

    // ---- loader coordinates -------------------------------------------------------------------------------------------------------
    const int cpos = lane & 7, lr8 = lane >> 3;
    const char* zero = reinterpret_cast<const char*>(g_zero_page) + cpos * 16;
    // halo round k < 5: halo row 64 k + 8 wave + lr8; round 5: 320 + 2 wave + lr8 on lanes < 16.  hpix = pixel index, -1 outside the image
    int hpix[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int hrow = k < 5 ? 64 * k + 8 * wave + lr8 : 320 + 2 * wave + (lr8 & 1);
        const int hy = (hrow * 3641) >> 16, hx = hrow - hy * HW_;      // / 18 for hrow < 336
        const int iy = ty0 + hy - 1, ix = tx0 + hx - 1;                // in the conv's input image: the x2 image for an upsampling conv
        const bool ok = hrow < HROWS && (unsigned)iy < (unsigned)Hl && (unsigned)ix < (unsigned)Wl;
        const int pix = (b * p.h_in + (min(max(iy, 0), Hl - 1) >> ups)) * p.w_in + (min(max(ix, 0), Wl - 1) >> ups);   // nearest x2: source pixel (y >> 1, x >> 1)
        hpix[k] = ok ? pix : -1;
    }
    // swizzled source chunk (bytes): key (hrow >> 1) & 7 = (4 wave + (lane >> 4)) & 7 for rounds 0-4, wave & 7 for round 5
    const uint32_t hsrc = (uint32_t)((cpos ^ ((4 * wave + (lane >> 4)) & 7)) * 16);
    const uint32_t hsrc5 = (uint32_t)((cpos ^ (wave & 7)) * 16);
    uint32_t woff[WR];
#pragma unroll
    for (int k = 0; k < WR; ++k) {
        const int q = 64 * k + (k >= WRF ? 4 : 8) * wave + lr8;
        woff[k] = (uint32_t)min(n0 + q, hot_N - 1) * hot_w_rs + (uint32_t)((cpos ^ ((q >> 1) & 7)) * 16);   // columns past N re-read the last weight row (never stored)
    }
    const uint32_t ldsH_wave = lds0 + (uint32_t)(wave * 8) * 128u;
    const uint32_t ldsH5_wave = lds0 + (uint32_t)(320 + wave * 2) * 128u;
    const uint32_t ldsW_wave = lds0 + 2u * H_BYTES + (uint32_t)(wave * 8) * 128u;
    const uint32_t ldsWh_wave = lds0 + 2u * H_BYTES + (uint32_t)(wave * 4) * 128u;

    // source of a chunk's halo: base + pixel * row bytes (wave-uniform per chunk; which tensor of the concat)
    struct ChunkA { uint64_t sb; uint32_t csrc2; };
    auto chunk_a = [&](int c) {
        const int ch = c * 64;
        const bool first = ch < hot_c0;
        ChunkA t;
        t.sb = (uint64_t)(first ? hot_a0 : hot_a1) + (uint64_t)(uint32_t)((first ? ch : ch - hot_c0) * 2);
        t.csrc2 = (uint32_t)(first ? hot_c0 : hot_c1) * 2u;
        return t;
    };
    auto halo_addr = [&](auto K_, const ChunkA& t, bool live) -> uint64_t {   // branch-free select of the zero page (see addr_a_round above)
        constexpr int k = decltype(K_)::value;
        const uint32_t off = __umul24((uint32_t)hpix[k], t.csrc2) + (k < 5 ? hsrc : hsrc5);   // (pixels, row bytes < 2^24: host)
        const uint32_t m32 = (uint32_t)(~hpix[k] >> 31) & (live ? 0xFFFFFFFFu : 0u);
        const uint64_t m64 = ((uint64_t)m32 << 32) | m32;
        return ((t.sb + off) & m64) | ((uint64_t)zero & ~m64);
    };
    auto fire_halo = [&](auto K_, uint64_t a, int hb) {
        constexpr int k = decltype(K_)::value;
        const uint32_t boff = (uint32_t)hb * (uint32_t)H_BYTES;
        if constexpr (k < 5) dma16(reinterpret_cast<const void*>(a), ldsH_wave + boff + (uint32_t)k * 8192u);
        else if constexpr (k == 5) dma16m(reinterpret_cast<const void*>(a), ldsH5_wave + boff, 0xFFFFull);
        else dma16(zero, lds0 + boff + 328u * 128u);                   // the filler: rows 328-335 are never read
    };
    auto issue_w = [&](auto LO_, auto HI_, uint32_t koff, int ib) {    // weight rounds [LO, HI) of one K step
        constexpr int LO = decltype(LO_)::value, HI = decltype(HI_)::value;
        const uint32_t boff = (uint32_t)ib * (uint32_t)W_BYTES;
#pragma unroll
        for (int k = LO; k < HI; ++k) {
            // (the step's K offset goes into the scalar base: added to the lane offsets, hipcc hoists the 9 x WR sums out of the chunk loop and spills them)
            const char* wb = reinterpret_cast<const char*>(hot_w) + koff;
            if (k < WRF) dma16s(wb, woff[k], ldsW_wave + boff + (uint32_t)k * 8192u);
            else dma16sm(wb, woff[k], ldsWh_wave + boff + (uint32_t)k * 8192u, 0xFFFFFFFFull);
        }
    };
    auto w_koff = [&](int c, int tap) { return (uint32_t)(tap * p.nkc + c) * hot_w_ks; };   // the weight matrix keeps its (tap, channel) column order

    constexpr int EC = 2, EH = MI / EC;
    f32x4 acc[EH][NJ][EC];
#pragma unroll
    for (int h = 0; h < EH; ++h)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < EC; ++i) acc[h][j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- prologue: the first chunk's halo, then the first LEAD steps' weights, each followed by a filler so that every step - these too -
    //      has put WR + 1 operations into the queue ------------------------------------------------------------------------------------
    {
        const ChunkA t = chunk_a(c_begin);
        fire_halo(std::integral_constant<int, 0>{}, halo_addr(std::integral_constant<int, 0>{}, t, true), 0);
        fire_halo(std::integral_constant<int, 1>{}, halo_addr(std::integral_constant<int, 1>{}, t, true), 0);
        fire_halo(std::integral_constant<int, 2>{}, halo_addr(std::integral_constant<int, 2>{}, t, true), 0);
        fire_halo(std::integral_constant<int, 3>{}, halo_addr(std::integral_constant<int, 3>{}, t, true), 0);
        fire_halo(std::integral_constant<int, 4>{}, halo_addr(std::integral_constant<int, 4>{}, t, true), 0);
        fire_halo(std::integral_constant<int, 5>{}, halo_addr(std::integral_constant<int, 5>{}, t, true), 0);
#pragma unroll
        for (int tt = 0; tt < LEAD; ++tt) {                            // (nkt >= 9 > LEAD: every one of these steps exists)
            issue_w(std::integral_constant<int, 0>{}, std::integral_constant<int, WR>{}, w_koff(c_begin, tt), tt);
            fire_halo(std::integral_constant<int, 6>{}, 0, 1);
        }
    }
    MSD_STAMP(1);
    wait_vmcnt<1 + (LEAD - 1) * (WR + 1)>();
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();   // the second half runs one slot behind the first

    // ---- fragment addressing --------------------------------------------------------------------------------------------------------
    const int rw = cg_wrow(r);
    const int woffs = 2 * H_BYTES + (wn * WNT + rw) * 128;
    const int cw0 = ((g ^ (rw >> 1)) << 4), cw1 = (((4 + g) ^ (rw >> 1)) << 4);
    bf16x8 af[2][IHS], wf[2][NJ];
    int fa[IHS];                                   // LDS byte offsets of the next phase's A fragments (ks = 0; ks = 1: ^ 64), prepared in the MFMA slot
    auto prep_frag = [&](int part, int ky, int kx, int hb) {
#pragma unroll
        for (int ii = 0; ii < IHS; ++ii) {
            const int hr = (wm * MI + part * IHS + ii + ky) * HW_ + kx + r;
            fa[ii] = hb * H_BYTES + hr * 128 + ((g ^ ((hr >> 1) & 7)) << 4);
        }
    };
    auto load_a = [&]() {
#pragma unroll
        for (int ii = 0; ii < IHS; ++ii) {
            af[0][ii] = *reinterpret_cast<const bf16x8*>(smem + fa[ii]);
            af[1][ii] = *reinterpret_cast<const bf16x8*>(smem + (fa[ii] ^ 64));
        }
    };
    auto load_w = [&](int rb) {
        const char* bw = smem + rb * W_BYTES + woffs;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            wf[0][jj] = *reinterpret_cast<const bf16x8*>(bw + jj * 2048 + cw0);
            wf[1][jj] = *reinterpret_cast<const bf16x8*>(bw + jj * 2048 + cw1);
        }
    };
    auto mfma_part = [&](auto IP_) {               // ks outermost: every accumulator sees the step's two K halves in ascending order
        constexpr int ip = decltype(IP_)::value;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
                for (int ii = 0; ii < IHS; ++ii) {
                    const int i = ip * IHS + ii;
                    acc[i / EC][jj][i % EC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][jj], af[ks][ii], acc[i / EC][jj][i % EC], 0, 0, 0);
                }
    };
    auto mfma_slot_open = [&]() {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
    };
    auto mfma_slot_close = [&]() {
        // one MFMA, then up to two of the address-arithmetic VALU instructions that ride in the slot
#pragma unroll
        for (int q = 0; q < 24; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x006, 2, 0);
        }
#pragma unroll
        for (int ii = 0; ii < IHS; ++ii) asm volatile("" : "+v"(fa[ii]));   // (pin: see conv_big_kernel)
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };

    // ---- K loop: chunks x 9 taps x 2 phases -------------------------------------------------------------------------------------------
    int rb = 0, ib = LEAD % NBW;
    prep_frag(0, 0, 0, 0);
    uint64_t ph = 0;                               // source of this step's halo-stream operation, prepared in phase 0's MFMA slot
    ChunkA nx = chunk_a(c_begin);
    for (int c = 0; c < nch; ++c) {
        const int hb = c & 1;
        const bool has_next = c + 1 < nch;
        if (has_next) nx = chunk_a(c_begin + c + 1);
#ifdef MSD_STAMPS
        if (c == 0) MSD_STAMP(2);
        if (c == (nch >> 1)) MSD_STAMP(5);
#endif
        auto step = [&](auto TAP_) {
            constexpr int TAP = decltype(TAP_)::value;
            constexpr int ky = TAP / 3, kx = TAP - ky * 3;
            constexpr int TAPW = (TAP + LEAD) % 9, CW = (TAP + LEAD) / 9;          // the step whose weights this one issues
            constexpr int TAPN = (TAP + 1) % 9;
            const int rem = nkt - 1 - (c * 9 + TAP);
            const uint32_t koff = w_koff(c_begin + c + CW, TAPW);
            // -- phase 0: LOAD (all weight fragments, image rows part 0; first weight rounds of step + LEAD), MFMA
            load_w(rb);
            load_a();
            __builtin_amdgcn_sched_barrier(0);
            if (rem >= LEAD) issue_w(std::integral_constant<int, 0>{}, std::integral_constant<int, W0>{}, koff, ib);
            mfma_slot_open();
            prep_frag(1, ky, kx, hb);
            if constexpr (TAP < 6) ph = halo_addr(std::integral_constant<int, TAP>{}, nx, has_next);
            mfma_part(std::integral_constant<int, 0>{});
            if constexpr (TAP < 6) asm volatile("" : "+v"(ph));
            mfma_slot_close();
            // -- phase 1: LOAD (image rows part 1; the other weight rounds, the halo-stream operation, the wait for the next step), MFMA
            load_a();
            __builtin_amdgcn_sched_barrier(0);
            if (rem >= LEAD) issue_w(std::integral_constant<int, W0>{}, std::integral_constant<int, WR>{}, koff, ib);
            if constexpr (TAP < 6) fire_halo(std::integral_constant<int, TAP>{}, ph, hb ^ 1);
            else fire_halo(std::integral_constant<int, 6>{}, 0, hb ^ 1);
            if (rem >= LEAD) wait_vmcnt<NWAIT>();
            else if (rem >= 1) big_wait(LEAD + WR * max(0, min(LEAD - 1, rem - 1)));
            mfma_slot_open();
            prep_frag(0, TAPN / 3, TAPN % 3, TAP == 8 ? hb ^ 1 : hb);
            mfma_part(std::integral_constant<int, 1>{});
            mfma_slot_close();
            if (++rb == NBW) rb = 0;
            if (++ib == NBW) ib = 0;
        };
        step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
        step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();   // (the first half's extra barrier: both halves have passed the same number)
    wait_vmcnt<0>();                              // the last fillers still write this workgroup's LDS

    // ---- shortcut operand (ResBlock: conv2(h) + conv_shortcut(x) as one contraction): after the slice's main chunks, its share of the
    //      64-channel chunks of a2 | a3, read at the OUTPUT pixel - one K step each on a plain 256-row tile (16 x 16 pixels x 128 B, staged in
    //      the halo buffers; weights in two slots of the ring).  Both halves in lock step, one step of lead: these steps are <= a quarter of
    //      the walk.  The slice's extras come AFTER its main chunks and the extras are dealt over the slices in order: part of the numerics
    //      class (the only kernel that walks a shortcut-folded conv chunk-major).
    const int nxc = p.nk - p.nk_main;
    if (nxc > 0) {
        const int eps = (nxc + p.nslices - 1) / p.nslices;
        const int e0 = min(nxc, (int)blockIdx.y * eps), e1 = min(nxc, e0 + eps);
        const int cx3 = p.K - p.nk_main * 64 - p.c2;   // channels of a3
        auto issue_extra = [&](int e, int slot) {
            const int ce = e * 64;
            const bool first = ce < p.c2;
            const uint64_t sb = (uint64_t)(first ? p.a2 : p.a3) + (uint64_t)(uint32_t)((first ? ce : ce - p.c2) * 2);
            const uint32_t cs2 = (uint32_t)(first ? p.c2 : cx3) * 2u;
            const uint32_t base = __builtin_amdgcn_readfirstlane(ldsH_wave + (uint32_t)slot * (uint32_t)H_BYTES);
#pragma unroll
            for (int k = 0; k < 4; ++k) {   // row q = 64 k + 8 wave + lr8 of the tile = pixel (q >> 4, q & 15); same swizzle key as the halo rounds
                const int q = 64 * k + 8 * wave + lr8;
                const int pix = (b * Hl + ty0 + (q >> 4)) * Wl + tx0 + (q & 15);
                dma16(reinterpret_cast<const void*>(sb + (__umul24((uint32_t)pix, cs2) + hsrc)), base + (uint32_t)k * 8192u);
            }
            issue_w(std::integral_constant<int, 0>{}, std::integral_constant<int, WR>{}, (uint32_t)(p.nk_main + e) * hot_w_ks, slot);
        };
        auto prep_xfrag = [&](int part, int slot) {
#pragma unroll
            for (int ii = 0; ii < IHS; ++ii) {
                const int q = (wm * MI + part * IHS + ii) * 16 + r;
                fa[ii] = slot * H_BYTES + q * 128 + ((g ^ ((q >> 1) & 7)) << 4);
            }
        };
        int slot = 0;
        if (e0 < e1) issue_extra(e0, 0);
        for (int e = e0; e < e1; ++e) {
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();   // this step's operands have landed for every wave; the other slot's readers are past their MFMAs
            if (e + 1 < e1) issue_extra(e + 1, slot ^ 1);
            load_w(slot);
            prep_xfrag(0, slot);
            load_a();
            mfma_part(std::integral_constant<int, 0>{});
            prep_xfrag(1, slot);
            load_a();
            mfma_part(std::integral_constant<int, 1>{});
            slot ^= 1;
        }
    }
    MSD_STAMP(3);

    auto epilogue_chunk = [&](auto H_) {
        constexpr int h = decltype(H_)::value;
        int mrow[EC];
#pragma unroll
        for (int i = 0; i < EC; ++i) mrow[i] = (b * Hl + ty0 + wm * MI + h * EC + i) * Wl + tx0;
        cg_epilogue<EC, NJ, true, false>(p, acc[h], mrow, n0 + wn * WNT, r, g);   // (a spatial tile lies inside one sample)
    };
    epilogue_chunk(std::integral_constant<int, 0>{});
    if constexpr (EH > 1) epilogue_chunk(std::integral_constant<int, 1>{});
    if constexpr (EH > 2) epilogue_chunk(std::integral_constant<int, 2>{});
    if constexpr (EH > 3) epilogue_chunk(std::integral_constant<int, 3>{});
    static_assert(EH <= 4, "epilogue chunks");
#ifdef MSD_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MSD_STAMP(4);
#endif
}

// ---- configurations: (BM, BN, WGM, WGN, IH, JH, NB, code); selected by tile_m = 5000 + BM, tile_n = BN, stages = code --------------
#ifndef MSD_BIG_CFGS
#define MSD_BIG_CFGS(X)          \
    X(256, 256, 2, 4, 2, 2, 2, 0) \
    X(256, 160, 4, 2, 2, 2, 3, 0) \
    X(256, 160, 4, 2, 2, 1, 3, 1) \
    X(256, 160, 4, 2, 2, 2, 2, 2) \
    X(256, 128, 4, 2, 2, 1, 3, 0) \
    X(256, 128, 4, 2, 2, 2, 3, 1) \
    X(128, 256, 2, 4, 2, 1, 3, 0) \
    X(128, 256, 2, 4, 2, 2, 3, 1)
#endif

constexpr int big_lds(int bm, int bn, int nb) { return nb * (bm + bn) * 128; }   // = BigGeo<...>::LDS

// halo-image configurations: (BN, WGM, WGN, weight ring depth, code); selected by tile_m = 5256, tile_n = BN, stages = 20 + code
#ifndef MSD_BIGHALO_CFGS
#define MSD_BIGHALO_CFGS(X) \
    X(160, 4, 2, 3, 0)      \
    X(128, 4, 2, 3, 0)      \
    X(128, 4, 2, 4, 1)
#endif
constexpr int bighalo_lds(int bn, int nbw) { return 2 * 336 * 128 + nbw * bn * 128; }

static bool g_big_attr_done = false;
static int msd_conv_big_init() {
    if (g_big_attr_done) return MSD_OK;
    hipError_t e = hipSuccess;
#define X(bm, bn, wgm, wgn, ih, jh, nb, code)                                                                                 \
    if (e == hipSuccess)                                                                                                      \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_big_kernel<bm, bn, wgm, wgn, ih, jh, nb, 0>),             \
                                hipFuncAttributeMaxDynamicSharedMemorySize, big_lds(bm, bn, nb));                             \
    if (e == hipSuccess)                                                                                                      \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_big_kernel<bm, bn, wgm, wgn, ih, jh, nb, 1>),             \
                                hipFuncAttributeMaxDynamicSharedMemorySize, big_lds(bm, bn, nb));                             \
    if (e == hipSuccess)                                                                                                      \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_big_kernel<bm, bn, wgm, wgn, ih, jh, nb, 2>),             \
                                hipFuncAttributeMaxDynamicSharedMemorySize, big_lds(bm, bn, nb));
    MSD_BIG_CFGS(X)
#undef X
#define X(bn, wgm, wgn, nbw, code)                                                                                            \
    if (e == hipSuccess)                                                                                                      \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bighalo_kernel<bn, wgm, wgn, nbw>),                       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, bighalo_lds(bn, nbw));
    MSD_BIGHALO_CFGS(X)
#undef X
    if (e != hipSuccess) MSD_FAIL((int)e, "hipFuncSetAttribute(conv_big): %s", hipGetErrorString(e));
    g_big_attr_done = true;
    return MSD_OK;
}

// 16-column blocks per wave of the configuration a (bm, bn, code) request selects, 0 if it is not built
int msd_conv_big_nj(int bm, int bn, int code) {
#define X(bm_, bn_, wgm, wgn, ih, jh, nb, code_) if (bm == bm_ && bn == bn_ && code == code_) return bn_ / wgn / 16;
    MSD_BIG_CFGS(X)
#undef X
    return 0;
}

// Launch for an already validated argument block (tiles_m / tiles_n / m_fast / nk_per / nslices set by msd_conv_gemm).
int msd_conv_big_launch(const CGArgs& a, int bm, int bn, int code, int slices, bool dense, hipStream_t stream) {
    int rc = msd_conv_big_init();
    if (rc) return rc;
    const dim3 grid(a.tiles_m * a.tiles_n, slices);
#define X(bm_, bn_, wgm, wgn, ih, jh, nb, code_)                                                                              \
    if (bm == bm_ && bn == bn_ && code == code_) {                                                                            \
        if (dense)                                                                                                            \
            hipLaunchKernelGGL((conv_big_kernel<bm_, bn_, wgm, wgn, ih, jh, nb, 1>), grid, dim3(512),                         \
                               big_lds(bm_, bn_, nb), stream, CG_HOT_ARGS(a), a);                                             \
        else if (a.upsample)                                                                                                  \
            hipLaunchKernelGGL((conv_big_kernel<bm_, bn_, wgm, wgn, ih, jh, nb, 2>), grid, dim3(512),                         \
                               big_lds(bm_, bn_, nb), stream, CG_HOT_ARGS(a), a);                                             \
        else                                                                                                                  \
            hipLaunchKernelGGL((conv_big_kernel<bm_, bn_, wgm, wgn, ih, jh, nb, 0>), grid, dim3(512),                         \
                               big_lds(bm_, bn_, nb), stream, CG_HOT_ARGS(a), a);                                             \
        return MSD_OK;                                                                                                        \
    }
    MSD_BIG_CFGS(X)
#undef X
    MSD_FAIL(MSD_E_UNSUPPORTED, "conv_big: no %d x %d configuration with code %d", bm, bn, code);
}

// ---- halo-image variant: host side ------------------------------------------------------------------------------------------------
int msd_conv_bighalo_nj(int bn, int code) {
#define X(bn_, wgm, wgn, nbw, code_) if (bn == bn_ && code == code_) return bn_ / wgn / 16;
    MSD_BIGHALO_CFGS(X)
#undef X
    return 0;
}

// Launch for an already validated argument block (3x3 / stride 1 / pad 1, h_in and w_in multiples of 16, no shortcut operand; tiles_m =
// batch x 16x16-pixel tiles, mg_tps / mg_tx set, nk_per in 64-channel CHUNKS: msd_conv_gemm).
int msd_conv_bighalo_launch(const CGArgs& a, int bn, int code, int slices, hipStream_t stream) {
    int rc = msd_conv_big_init();
    if (rc) return rc;
    const dim3 grid(a.tiles_m * a.tiles_n, slices);
#define X(bn_, wgm, wgn, nbw, code_)                                                                                          \
    if (bn == bn_ && code == code_) {                                                                                         \
        hipLaunchKernelGGL((conv_bighalo_kernel<bn_, wgm, wgn, nbw>), grid, dim3(512), bighalo_lds(bn_, nbw), stream,         \
                           CG_HOT_ARGS(a), a);                                                                                \
        return MSD_OK;                                                                                                        \
    }
    MSD_BIGHALO_CFGS(X)
#undef X
    MSD_FAIL(MSD_E_UNSUPPORTED, "conv_big: no halo-image configuration 256 x %d with code %d", bn, code);
}

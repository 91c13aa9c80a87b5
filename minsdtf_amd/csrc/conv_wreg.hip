// Implicit-GEMM convolution / dense layer with the WEIGHTS loaded global -> VGPR in MFMA-fragment order ("wreg" form).
//
// Same contraction, K walk, MFMA operand placement and epilogue as conv_gemm.hip (so the same bits: tests compare them);
// what changes is how the B operand (weights) reaches the matrix core:
//   * the weights are static, so they are laid out at pack time exactly as the lanes of a `v_mfma_f32_16x16x32_bf16` A
//     operand hold them (packing.fragment_major, MsdConvGemm.w_layout = 2): per 64-deep K tile, per 16-column block and per
//     32-deep half, the 64 lanes x 16 bytes of one fragment are ONE contiguous, aligned 1 KiB — a wave fetches a fragment
//     with one fully coalesced `global_load_dwordx4`, two K tiles ahead, into registers: no LDS write, no LDS read, no
//     barrier dependence for B.  Only the activation tile keeps the LDS-DMA ring;
//   * every wave of a workgroup owns ALL BM rows and its own NJ 16-column blocks (the waves split N only), so no weight byte
//     is loaded twice per workgroup and a wave reads BM/16 A fragments per 32-deep step for BM/16 x NJ MFMAs: 1/NJ fragment
//     reads per MFMA (conv_gemm's 64x32 wave tile: 0.75, its 32x32 one: 1.0);
//   * the LDS ring holds S x BM x 128 B (48 KB for 128 rows x 3 stages) where conv_gemm holds S x (BM + BN) x 128 B: two or
//     three workgroups fit a CU, so one's wait -> barrier -> fragment reads overlap another's MFMAs, and a grid a little
//     larger than the chip (320 workgroups) is resident at once instead of running a quarter-full second round;
//   * BM = M for the weight-streaming layers of the 8x8 level (M = 128): every weight byte is fetched once chip-wide.
// The register loads are inline asm (hipcc would otherwise wait vmcnt(0) for them while LDS-DMAs are in flight:
// cdna_hip_programming.md §5 "Three .s-level traps" (b)); both queues are counted by hand: one group = AR LDS-DMAs + 2 NJ
// register loads per K tile, `s_waitcnt vmcnt((S - 2) x group)` retires the oldest group, and an empty asm statement naming
// the destination registers keeps every consumer below that wait (§5.7 form (ii)).
#include <type_traits>
#include "conv_common.h"

typedef uint32_t wr_u32x4 __attribute__((ext_vector_type(4)));

#ifdef MSD_STAMPS
// (tools/gemm_stamps.py, `make stamps` library only: the in-kernel timeline of this translation unit's kernels)
extern "C" MSD_API int msd_debug_stamps_wreg(unsigned long long* host_out, int count) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * (size_t)count);
}
#endif

// one 16-column x 32-deep weight fragment: 64 lanes x 16 B from `base` (wave-uniform, SGPR pair) + per-lane byte offset (+ 1 KiB
// for the second half of the K tile).  The first load of a group opens with s_nop 4 (a scalar base fresh from SALU / readfirstlane
// arithmetic read by a VMEM instruction inside an asm string: nothing pads it for us).
__device__ __forceinline__ void wr_load_first(wr_u32x4& lo, wr_u32x4& hi, const void* base, uint32_t voff) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024"
                 : "=&v"(lo), "=&v"(hi) : "v"(voff), "s"(base) : "memory");
}
__device__ __forceinline__ void wr_load(wr_u32x4& lo, wr_u32x4& hi, const void* base, uint32_t voff) {
    asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024"
                 : "=&v"(lo), "=&v"(hi) : "v"(voff), "s"(base) : "memory");
}

// BM x (NW x NJ x 16) output tile on NW waves (all split over N), S-stage LDS ring for A, S - 1 register sets for B.
// KT = K tiles per ring stage / per wait + barrier (1 or 2).  The small-M layers of the 16x16 and 8x8 levels are latency chains:
// a 64-deep K step of a 64-row tile is wait -> barrier -> fragment reads -> 8-16 MFMAs, ~0.3 us for ~0.05 us of MFMA work, and with
// KT = 2 that chain is paid once per 128 channels.  In conv_gemm.hip the same idea cost the second workgroup per CU (both
// operands in a 98-131 KB ring); here the ring holds A only (64 rows x 2 tiles x 3 stages = 48 KB).  The K tiles are still walked
// in ascending order, so the bits do not change.
// WGM = wave rows (1: every wave owns all BM rows; 2: the waves form a 2 x NW/2 grid, each owning BM/2 rows x NJ blocks — for
// 256-row tiles on 8 waves: 48 KB of operands per K tile where two 128x128 workgroups move 64 KB; the two waves of a column
// pair load the same fragments, the second from the CU's L1).
template <int BM, int NJ, int NW, int S, bool DENSE, int KT = 1, int WGM = 1>
__global__ __launch_bounds__(NW * 64) void conv_wreg_kernel(CG_HOT_PARAMS, const CGArgs p) {
    CG_HOT_UNPACK;
    constexpr int NT = NW * 64;
    constexpr int WGN = NW / WGM;
    constexpr int BN = WGN * NJ * 16;
    constexpr int WMT = BM / WGM;                  // rows per wave
    constexpr int MI = WMT / 16;
    static_assert(NW % WGM == 0 && BM % (16 * WGM) == 0, "wave grid");
    constexpr int RPP = NT / 8;                   // rows covered by one pass of the workgroup's DMAs
    constexpr int AR = BM * 8 / NT;               // LDS-DMA instructions per thread per K tile
    constexpr int L = KT * (AR + 2 * NJ);         // vector-memory operations per thread per ring stage (one group)
    constexpr int PB = S - 1;                     // register sets for the weight fragments (= stages in flight)
    constexpr int A_BYTES = BM * 128;             // one K tile of A; a ring stage holds KT of them
    static_assert(AR >= 1 && (BM * 8) % NT == 0 && BM % 16 == 0 && S >= 3 && (S - 2) * L <= 63 && (KT == 1 || KT == 2), "tile config");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave - wm * WGN;
    const int r = lane & 15, g = lane >> 4;
    MSD_STAMP(0);
    // tile order as in conv_gemm.hip: XCD-contiguous runs that share the pixel rows (n fastest) or the weight panel (m fastest)
    const int tile = xcd_remap(blockIdx.x, hot_tiles_m * hot_tiles_n);
    const int tdiv = hot_m_fast ? hot_tiles_m : hot_tiles_n;
    const int tq = udiv_magic(tile, tdiv, hot_mg_tdiv), tr = tile - tq * tdiv;
    const int tile_n = hot_m_fast ? tq : tr;
    const int tile_m = hot_m_fast ? tr : tq;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kt_begin = blockIdx.y * hot_nk_per;
    const int kt_end = min(hot_nk, kt_begin + hot_nk_per);
    const int nkt = (kt_end - kt_begin + KT - 1) / KT;   // ring stages (= steps) of this slice; the last one may hold fewer than KT tiles

    // ---- A loader coordinates (conv_gemm.hip's): thread -> (row = lrow + RPP i, 16-byte chunk position tid & 7) ----------
    const int cpos = tid & 7, lrow = tid >> 3;
    const int Hl = p.upsample ? 2 * p.h_in : p.h_in;
    const int Wl = p.upsample ? 2 * p.w_in : p.w_in;
    const char* zero = reinterpret_cast<const char*>(g_zero_page) + cpos * 16;
    int ab[AR], ay[AR], ax[AR], asrc[AR];
    uint32_t aoff0[AR], aoff1[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int row = lrow + RPP * i;
        const int m = m0 + row;
        asrc[i] = (cpos ^ ((row >> 1) & 7)) * 8;   // swizzle on the SOURCE chunk (LDS-DMA writes linearly)
        if constexpr (DENSE) {
            const uint32_t mc = (uint32_t)min(m, hot_M - 1);   // rows past M re-read the last row: never stored
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
    const uint32_t lds_wave = lds0 + (uint32_t)(wave * 8) * 128u;
    auto issue_a = [&](int kt, int slot) {   // slot = stage * KT + tile inside the stage
        const uint32_t sbase = lds_wave + (uint32_t)slot * A_BYTES;
        if constexpr (DENSE) {
            const int c = kt * 64;
            const bool first = c < hot_c0;                       // wave-uniform: which tensor of the concat
            const bf16_t* abase = first ? hot_a0 : hot_a1;
            const uint32_t cb = (uint32_t)(first ? c : c - hot_c0) * 2u;
#pragma unroll
            for (int i = 0; i < AR; ++i) dma16s(abase, (first ? aoff0[i] : aoff1[i]) + cb, sbase + (uint32_t)(RPP * i) * 128u);
            return;
        }
        // general form (tap / padding / stride / upsampling / shortcut-operand addresses), branch-free: see conv_gemm.hip
        const bool extra = kt >= p.nk_main;
        const int tap = extra ? 0 : udiv_magic(kt, p.nkc, p.mg_nkc);
        const int c = extra ? (kt - p.nk_main) * 64 : (kt - tap * p.nkc) * 64;
        const int ky = extra ? p.pad : (tap * 11) >> 5;
        const int kx = extra ? p.pad : tap - ky * 3;
        const int cA = extra ? p.c2 : hot_c0;
        const bool first = c < cA;
        const uint64_t sb = (uint64_t)(extra ? (first ? p.a2 : p.a3) : (first ? hot_a0 : hot_a1));
        const int csrc = first ? cA : (extra ? p.K - p.nk_main * 64 - p.c2 : hot_c1), coff = first ? c : c - cA;
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
    };

    // ---- B: this wave's NJ column blocks of K tile kt = 2 NJ contiguous KiB of the fragment-major image ------------------
    // image: [K / 64][N / 16][2][64 lanes][16 B]; blocks past N re-read the last block (their columns are never stored)
    const int NB = hot_N >> 4;
    const int nb0 = (n0 >> 4) + wn * NJ;
    uint32_t boff[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) boff[j] = (uint32_t)min(nb0 + j, NB - 1) * 2048u + (uint32_t)lane * 16u;
    const size_t kt_bytes = (size_t)NB * 2048u;
    wr_u32x4 bw[PB][KT][NJ][2];
    auto issue_b = [&](int kt, wr_u32x4 (&dst)[NJ][2]) {
        const char* base = reinterpret_cast<const char*>(hot_w) + (size_t)kt * kt_bytes;
        wr_load_first(dst[0][0], dst[0][1], base, boff[0]);
#pragma unroll
        for (int j = 1; j < NJ; ++j) wr_load(dst[j][0], dst[j][1], base, boff[j]);
    };

    // accumulators in chunks of EC row fragments (64 rows): the epilogue runs once per chunk, so its loads (residual, LayerNorm
    // partials, ...) hold registers for 4 row fragments at a time, whatever BM — the register profile of conv_gemm's 64-row wave tiles
    constexpr int EC = MI < 4 ? MI : 4, EH = MI / EC;
    static_assert(MI % EC == 0, "row fragments per epilogue chunk");
    f32x4 acc[EH][NJ][EC];
#pragma unroll
    for (int h = 0; h < EH; ++h)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < EC; ++i) acc[h][j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // a group = the KT tiles of one ring stage.  Every group has exactly L operations (the counted waits rely on it): a tile past
    // the slice's end (odd tile count with KT = 2) is issued as a re-read of the slice's last tile and never multiplied
    auto issue_group_a = [&](int stp, int stage) {
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) issue_a(min(kt_begin + stp * KT + kk, kt_end - 1), stage * KT + kk);
    };
    auto issue_group_b = [&](int stp, wr_u32x4 (&dst)[KT][NJ][2]) {
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) issue_b(min(kt_begin + stp * KT + kk, kt_end - 1), dst[kk]);
    };
    // prologue: groups 0 .. S-2 (A tiles into ring stage s, B fragments into register set s)
#pragma unroll
    for (int s = 0; s < PB; ++s)
        if (s < nkt) { issue_group_a(s, s); issue_group_b(s, bw[s]); }
    MSD_STAMP(1);

    const int swz = r >> 1;
    int stage = 0;
    // one ring stage on register set U (compile-time: the sets rotate by unrolling the loop PB times)
    auto step = [&](int it, wr_u32x4 (&bs)[KT][NJ][2]) {
        // retire group `it`: all but the groups issued after it may stay in flight
        const int later = min(nkt, it + S - 1) - (it + 1);
        wait_vmcnt_tiles<L, S - 2>(later);
#pragma unroll
        for (int kk = 0; kk < KT; ++kk)
#pragma unroll
            for (int j = 0; j < NJ; ++j) { asm volatile("" : "+v"(bs[kk][j][0])); asm volatile("" : "+v"(bs[kk][j][1])); }   // consumers stay below the wait
        __builtin_amdgcn_s_barrier();   // stage `it` visible to all waves; ring stage (it - 1) % S free for reuse
#ifdef MSD_STAMPS
        if (it == 0) MSD_STAMP(2);
        if (it == (nkt >> 1)) MSD_STAMP(5);
#endif
        const bool more = it + S - 1 < nkt;
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) {
            if (kk > 0 && kt_begin + it * KT + kk >= kt_end) break;   // (wave-uniform: the slice's last stage holds one tile)
            const char* bA = smem + (stage * KT + kk) * A_BYTES + (wm * WMT + r) * 128;
            bf16x8 a0[MI], a1[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) a0[i] = *reinterpret_cast<const bf16x8*>(bA + i * 16 * 128 + ((g ^ swz) << 4));
            if (kk == 0 && more) {   // ring stage (it - 1) % S: every wave finished reading it before the barrier above
                int st = stage + S - 1;
                if (st >= S) st -= S;
                issue_group_a(it + S - 1, st);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) a1[i] = *reinterpret_cast<const bf16x8*>(bA + i * 16 * 128 + (((4 + g) ^ swz) << 4));
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    acc[i / EC][j][i % EC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bs[kk][j][0]), a0[i], acc[i / EC][j][i % EC], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    acc[i / EC][j][i % EC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bs[kk][j][1]), a1[i], acc[i / EC][j][i % EC], 0, 0, 0);
        }
        // this set's registers are free once the MFMAs above have been ISSUED (they read their operands at issue): the fragments of
        // stage it + S - 1 go into the same set, the last operations of the group
        if (more) issue_group_b(it + S - 1, bs);
        if (++stage == S) stage = 0;
    };
    for (int it = 0; it < nkt; it += PB) {
#pragma unroll
        for (int u = 0; u < PB; ++u)
            if (it + u < nkt) step(it + u, bw[u]);
    }

    MSD_STAMP(3);
    auto epilogue_chunk = [&](auto H_) {   // (explicit instances: a loop that hipcc does not unroll sends acc[h] to scratch)
        constexpr int h = decltype(H_)::value;
        int mrow[EC];
#pragma unroll
        for (int i = 0; i < EC; ++i) mrow[i] = m0 + wm * WMT + (h * EC + i) * 16;
        cg_epilogue<EC, NJ, false, DENSE>(p, acc[h], mrow, n0 + wn * NJ * 16, r, g, reinterpret_cast<float*>(smem), wn, WGN, wm * WMT + h * EC * 16, BM, tile_n);
    };
    epilogue_chunk(std::integral_constant<int, 0>{});
    if constexpr (EH > 1) epilogue_chunk(std::integral_constant<int, 1>{});
    if constexpr (EH > 2) epilogue_chunk(std::integral_constant<int, 2>{});
    if constexpr (EH > 3) epilogue_chunk(std::integral_constant<int, 3>{});
    static_assert(EH <= 4, "epilogue chunks");
#ifdef MSD_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores have left the wave
    MSD_STAMP(4);
#endif
}

// ---- configurations: (BM, NJ, NW, S, KT, WGM); selected by tile_m = 4000 + BM, tile_n = (NW / WGM) x NJ x 16,
//      stages = S (+ 10 for 8 waves) (+ 20 for two K tiles per stage) ------------------------------------------------------------
#define MSD_WREG_CFGS(X) \
    X(128, 2, 4, 3, 1, 1)  \
    X(128, 2, 4, 4, 1, 1)  \
    X(128, 1, 4, 3, 1, 1)  \
    X(128, 1, 4, 4, 1, 1)  \
    X(64, 2, 4, 3, 1, 1)   \
    X(64, 2, 4, 4, 1, 1)   \
    X(64, 4, 4, 3, 1, 1)   \
    X(64, 4, 4, 4, 1, 1)   \
    X(64, 1, 4, 4, 1, 1)   \
    X(256, 1, 4, 3, 1, 1)  \
    X(128, 1, 8, 3, 1, 1)  \
    X(64, 1, 4, 3, 2, 1)   \
    X(64, 1, 4, 4, 2, 1)   \
    X(64, 2, 4, 3, 2, 1)   \
    X(64, 2, 4, 4, 2, 1)   \
    X(64, 4, 4, 3, 2, 1)   \
    X(128, 1, 4, 3, 2, 1)  \
    X(128, 2, 4, 3, 2, 1)  \
    X(256, 2, 8, 3, 1, 2)  \
    X(256, 2, 8, 4, 1, 2)

constexpr int wreg_lds(int bm, int nj, int nw, int s, int kt, int wgm) {
    // the A ring; the LayerNorm-producer epilogue reuses it for BM x (BN / 16) float2 block sums
    const int ring = s * kt * bm * 128, red = bm * (nw / wgm) * nj * 8;
    return ring > red ? ring : red;
}
#define X(bm, nj, nw, st, kt, wgm) static_assert(wreg_lds(bm, nj, nw, st, kt, wgm) <= 160 * 1024, "wreg configuration exceeds the CU's LDS");
MSD_WREG_CFGS(X)
#undef X
constexpr int wreg_code(int nw, int s, int kt) { return s + (nw == 8 ? 10 : 0) + (kt == 2 ? 20 : 0); }

static bool g_wreg_attr_done = false;
int msd_conv_wreg_init() {
    if (g_wreg_attr_done) return MSD_OK;
    hipError_t e = hipSuccess;
#define X(bm, nj, nw, st, kt, wgm)                                                                                        \
    if (e == hipSuccess)                                                                                               \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wreg_kernel<bm, nj, nw, st, false, kt, wgm>),      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, wreg_lds(bm, nj, nw, st, kt, wgm));        \
    if (e == hipSuccess)                                                                                               \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wreg_kernel<bm, nj, nw, st, true, kt, wgm>),       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, wreg_lds(bm, nj, nw, st, kt, wgm));
    MSD_WREG_CFGS(X)
#undef X
    if (e != hipSuccess) MSD_FAIL((int)e, "hipFuncSetAttribute(conv_wreg): %s", hipGetErrorString(e));
    g_wreg_attr_done = true;
    return MSD_OK;
}

// 16-column blocks per wave of the configuration a (bm, bn, stages) request selects, 0 if it is not built
int msd_conv_wreg_nj(int bm, int bn, int stages) {
#define X(bm_, nj, nw, st, kt, wgm) if (bm == bm_ && bn == (nw / wgm) * nj * 16 && stages == wreg_code(nw, st, kt)) return nj;
    MSD_WREG_CFGS(X)
#undef X
    return 0;
}

// Launch for an already validated argument block (tiles_m / tiles_n / m_fast / nk_per / nslices set by msd_conv_gemm).
int msd_conv_wreg_launch(const CGArgs& a, int bm, int bn, int stages, int slices, bool dense, hipStream_t stream) {
    int rc = msd_conv_wreg_init();
    if (rc) return rc;
    const dim3 grid(a.tiles_m * a.tiles_n, slices);
#define X(bm_, nj, nw, st, kt, wgm)                                                                                    \
    if (bm == bm_ && bn == (nw / wgm) * nj * 16 && stages == wreg_code(nw, st, kt)) {                                  \
        if (dense)                                                                                                     \
            hipLaunchKernelGGL((conv_wreg_kernel<bm_, nj, nw, st, true, kt, wgm>), grid, dim3(nw * 64), wreg_lds(bm_, nj, nw, st, kt, wgm), stream, CG_HOT_ARGS(a), a); \
        else                                                                                                           \
            hipLaunchKernelGGL((conv_wreg_kernel<bm_, nj, nw, st, false, kt, wgm>), grid, dim3(nw * 64), wreg_lds(bm_, nj, nw, st, kt, wgm), stream, CG_HOT_ARGS(a), a); \
        return MSD_OK;                                                                                                 \
    }
    MSD_WREG_CFGS(X)
#undef X
    MSD_FAIL(MSD_E_UNSUPPORTED, "conv_wreg: no %d x %d configuration with stages %d", bm, bn, stages);
}

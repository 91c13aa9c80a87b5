// Fused scaled-dot-product attention for gfx950 (flash-style, online softmax in registers).
//
// Replaces CrossAttention.call's einsum / softmax / einsum (diffusion_model.py:110-127): the
// reference materialises (B, 8, S, T) fp32 scores (2.9 GB per UNet forward at 512x512); here the
// scores live only in MFMA accumulators.
//
// Work split: one workgroup = 4 waves = 128 queries of one (batch, head); each wave owns 32
// queries (two 16-query MFMA column blocks).  K/V are walked in 64-key tiles staged in LDS.
//
// Both products are issued TRANSPOSED so that the query index sits on the MFMA column (= lane&15):
//   S^T[key, q] = sum_d K[key, d] * Q[q, d]      A = K rows (LDS, 16-byte reads), B = Q (registers)
//   O^T[d, q]   = sum_key V^T[d, key] * P^T[key, q]   A = V^T rows (LDS), B = P (from S^T registers)
// Consequences: (1) every lane works on ONE query per column block, so the running max / sum and
// the rescale factor are lane-local scalars and the row max needs only two cross-lane steps
// (lane ^ 16, lane ^ 32); (2) the S^T accumulator registers of two 16-key blocks ARE the B-operand
// fragment of the PV product after a bf16 pack — no LDS round trip, no transpose: the k-slot
// order inside an MFMA is free as long as both operands agree, so V^T is read from LDS in the
// accumulator's key order (keys 4g..4g+3 and 16+4g..16+4g+3 of each 32-key step);
// (3) V is consumed as V^T[d][key], which the QKV projection epilogue writes directly
// (msd_conv_gemm split mode), so no transposing loads anywhere.
// head_dim 40 / 80 are zero-padded to 64 / 96 in the QK^T k-dimension only (LDS pad columns stay 0).
#include "common.h"
#include <type_traits>

struct AArgs {
    const bf16_t* q; const bf16_t* k; const bf16_t* vt; bf16_t* out;
    int batch, heads, s, t, q_ld, k_ld, vt_ld, o_ld;
    float sl2;  // scale * log2(e)
    int causal; // key index > query index is masked (CLIP text encoder, text_encoder.py:75-78)
    int presc;  // q already carries scale * log2(e) (folded into the projection that produced it)
    uint32_t mg_qtiles, mg_heads;   // floor(2^32 / d) for the workgroup-id decomposition (udiv_magic)
    int q_begin, q_count;           // attention32_kernel: the launch covers queries [q_begin, q_begin + q_count) of every sample (mg_qtiles: of that range)
    float* ws; int nsplit_shift;    // attention512_kernel: key walk split over 1 << nsplit_shift workgroups per query tile, partial (O, m, l) to ws
};

// Kernarg preload (conv_common.h CG_HOT_PARAMS): the 16 dwords the kernels' prologues need, as leading scalar arguments
#define ATTN_HOT_PARAMS const bf16_t* hot_q, const bf16_t* hot_k, const bf16_t* hot_vt, int hot_heads, int hot_s, int hot_t, int hot_q_ld, int hot_k_ld, \
                        int hot_batch, uint32_t hot_mg_qtiles, uint32_t hot_mg_heads, int hot_q_begin, int hot_q_count
#define ATTN_HOT_ARGS(a) (a).q, (a).k, (a).vt, (a).heads, (a).s, (a).t, (a).q_ld, (a).k_ld, (a).batch, (a).mg_qtiles, (a).mg_heads, (a).q_begin, (a).q_count

// Maximum over the 4 lanes {l, l^16, l^32, l^48} (the four 16-lane rows of the wave), result on every lane: two
// VALU row swaps (v_permlane16_swap / v_permlane32_swap) instead of two ds_bpermute round trips through the LDS crossbar
// (each ~100+ cycles of latency in the middle of the softmax's dependency chain).  v_max via asm: fmaxf() on these
// bit-cast values makes hipcc canonicalise both inputs first (two more VALU ops per step).
__device__ __forceinline__ float rows_max4(float v) {
    const uint32_t u = __float_as_uint(v);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);   // a[0]: rows {0,0,2,2}, a[1]: rows {1,1,3,3}
    float y;
    asm("v_max_f32 %0, %1, %2\n\ts_nop 1" : "=v"(y) : "v"(a[0]), "v"(a[1]));   // (s_nop: VALU write -> v_permlane read)
    const uint32_t w = __float_as_uint(y);
    auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);   // b[0]: lower half everywhere, b[1]: upper half
    asm("v_max_f32 %0, %1, %2" : "=v"(y) : "v"(b[0]), "v"(b[1]));
    return y;
}

// Lazy rescaling (defer-max): the exponentials of a tile are taken against a REFERENCE maximum m_ref that is only moved
// when a tile's maximum exceeds it by more than ATTN_THR (in exp2 units), so p <= 2^ATTN_THR instead of p <= 1.  bf16 / fp32
// are floating point, so the relative precision of P, of the row sum and of O does not depend on that factor; what it
// buys is that after the first tile the O-wide rescale (and its exp2) almost never runs, and that the per-tile code has
// ONE rarely-taken wave-uniform branch instead of one taken-half-the-time branch per query block.  Tile 0 always takes
// the branch (m_ref := the tile's maximum), which also guarantees a row sum >= 1.
#define ATTN_THR 8.0f

// QF = 16-query MFMA column blocks per wave: 2 (128 queries per workgroup) or 1 (64 queries per workgroup: twice the
// workgroups — for launches whose 128-query grid leaves CUs idle or a SIMD with fewer than 3 waves).
// PRESC: q already carries scale * log2(e) (folded into the weights of the projection that produced it, see
// MsdAttention.q_prescaled): the S^T accumulator chain then STARTS from -m_ref (MFMA C operand) and ends as the exp2
// argument itself — no per-score fma.
// NBUF = K/V tile buffers in LDS and with them the loop form:
//   1  load tile -> store -> compute, two barriers per tile (d = 160: no registers left for a prefetch);
//   2  tile t+1 is loaded to registers while tile t is computed and stored behind it, one barrier per tile;
// (A software-pipelined form — the S^T MFMAs of tile t+1 issued between the exponentials of tile t, third LDS buffer, two
//  accumulator sets — measured SLOWER: 102.8 vs 98.6 us at S = 4096 d = 40 batch 2, 434 vs 338 us at batch 8.  The loop is
//  bound by the SIMD's instruction ISSUE, which both co-resident waves share and which an MFMA also occupies for 8 of its
//  16 cycles: ~750 issue cycles per tile and wave, measured ~1600 per tile for the two waves of a SIMD; overlapping inside
//  one wave adds registers (190 vs 152: one wave per SIMD fewer) and removes no instruction.)
template <int D, int NBUF, int QF = 2, bool PRESC = false>
__global__ __launch_bounds__(256) void attention_kernel(ATTN_HOT_PARAMS, const AArgs p) {
    constexpr bool PIPE = NBUF == 2;
    constexpr int QT = 64 * QF;          // queries per workgroup
    constexpr int DPAD = ((D + 31) / 32) * 32;
    constexpr int KS = DPAD / 32;        // k-steps of QK^T
    constexpr int DF = (D + 15) / 16;    // 16-row blocks of O^T
    constexpr int KROW = DPAD * 2 + 16;  // bytes; odd number of 16-B slots -> conflict-free fragment reads
    constexpr int VROW = 64 * 2 + 16;
    constexpr int DCH = D / 8;           // 16-byte chunks per K row
    constexpr int BUF_BYTES = 64 * KROW + DF * 16 * VROW;   // one K tile [64][KROW] + one V^T tile [DF*16][VROW]
    constexpr bool ONES_ROW = (D % 16) != 0;  // a spare padding row of the V^T tile carries the softmax denominator
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    // 1-D grid, XCD-aware: all query tiles of one (batch, head) run on one XCD so its K / V^T
    // (655 KB at S=4096, d=40) stay in that XCD's L2 instead of every XCD streaming all heads
    const int qtiles = (hot_s + QT - 1) / QT;
    const int wi = xcd_remap(blockIdx.x, qtiles * hot_heads * hot_batch);   // (= gridDim.x, without reading the implicit arguments)
    const int bh = udiv_magic(wi, qtiles, hot_mg_qtiles);
    const int b = udiv_magic(bh, hot_heads, hot_mg_heads), h = bh - b * hot_heads;
    const int q0 = (wi - bh * qtiles) * QT + wave * (16 * QF);

    // zero the whole LDS image once: pad columns / pad rows are never written afterwards.  (Head sizes that are whole 32-channel k-steps
    // and whole 16-row blocks - d = 160, 64 - have neither: every byte a fragment read touches is written by lstore, so the fill and the
    // barrier that orders it in front of the first loads are skipped; not measurable in the loop: 94.4 vs 94.3 ms.)
    constexpr bool NEEDS_ZERO = DPAD != D || DF * 16 != D || ONES_ROW;
    if constexpr (NEEDS_ZERO)
        for (int off = tid * 16; off < NBUF * BUF_BYTES; off += 256 * 16)
            *reinterpret_cast<uint4*>(smem + off) = make_uint4(0, 0, 0, 0);

    if (ONES_ROW) {
        __syncthreads();
        if (tid < 16) {  // 64 keys x bf16(1.0) in row D of each V^T buffer (never overwritten: tiles write rows < D)
#pragma unroll
            for (int bufi = 0; bufi < NBUF; ++bufi)
                *reinterpret_cast<uint2*>(smem + bufi * BUF_BYTES + 64 * KROW + D * VROW + tid * 8) = make_uint2(0x3F803F80u, 0x3F803F80u);
        }
    }

    f32x4 oacc[DF][QF];
#pragma unroll
    for (int df = 0; df < DF; ++df)
#pragma unroll
        for (int f = 0; f < QF; ++f) oacc[df][f] = (f32x4){0, 0, 0, 0};
    float mref[QF], lrun[QF];   // reference maximum of the exponentials (raw-score units; PRESC: exp2 units), row sums
    f32x4 negm[QF];             // PRESC: {-m_ref x 4}, the C operand that starts every S^T accumulator chain
#pragma unroll
    for (int f = 0; f < QF; ++f) { mref[f] = 0.f; lrun[f] = 0.f; negm[f] = (f32x4){0, 0, 0, 0}; }

    const bf16_t* kbase = hot_k + (size_t)b * hot_t * hot_k_ld + h * D;
    const bf16_t* vbase = hot_vt + ((size_t)b * hot_heads + h) * D * p.vt_ld;
    const int ntiles = (hot_t + 63) / 64;

    // K / V^T tile staging: global -> registers (issued ahead when PIPE) -> LDS.
    // Per-thread source pointers / LDS offsets are fixed for the whole kernel; a full tile
    // (t0 + 64 <= t, i.e. every self-attention tile) takes a branch-free path, only a ragged last
    // tile (text context) pays for bounds checks and for zeroing the V^T padding columns.
    constexpr int KCH = (64 * DCH + 255) / 256, VCH = (D * 8 + 255) / 256;
    uint4 rk[KCH], rv[VCH];
    const bf16_t* kptr[KCH];
    const bf16_t* vptr[VCH];
    int klds[KCH], vlds[VCH], krow[KCH], vkey[VCH];
    bool kin[KCH], vin[VCH];
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
        const int idx = tid + 256 * i;
        kin[i] = idx < 64 * DCH;
        const int row = kin[i] ? idx / DCH : 0, ch = kin[i] ? idx - row * DCH : 0;
        krow[i] = row;
        kptr[i] = kbase + (size_t)row * hot_k_ld + ch * 8;
        klds[i] = row * KROW + ch * 16;
    }
#pragma unroll
    for (int i = 0; i < VCH; ++i) {
        const int idx = tid + 256 * i;
        vin[i] = idx < D * 8;
        const int d = vin[i] ? idx >> 3 : 0, ch = idx & 7;
        vkey[i] = ch * 8;
        vptr[i] = vbase + (size_t)d * p.vt_ld + ch * 8;
        vlds[i] = d * VROW + ch * 16;
    }
    auto gload = [&](int t0) {
        const size_t koff = (size_t)t0 * hot_k_ld;
        if (t0 + 64 <= hot_t) {
#pragma unroll
            for (int i = 0; i < KCH; ++i)
                if (kin[i]) rk[i] = *reinterpret_cast<const uint4*>(kptr[i] + koff);
#pragma unroll
            for (int i = 0; i < VCH; ++i)
                if (vin[i]) rv[i] = *reinterpret_cast<const uint4*>(vptr[i] + t0);
        } else {
#pragma unroll
            for (int i = 0; i < KCH; ++i) {
                rk[i] = make_uint4(0, 0, 0, 0);
                if (kin[i] && t0 + krow[i] < hot_t) rk[i] = *reinterpret_cast<const uint4*>(kptr[i] + koff);
            }
#pragma unroll
            for (int i = 0; i < VCH; ++i) {
                rv[i] = make_uint4(0, 0, 0, 0);
                const int key0 = t0 + vkey[i];
                if (vin[i] && key0 + 8 <= p.vt_ld && key0 < hot_t) rv[i] = *reinterpret_cast<const uint4*>(vptr[i] + t0);
            }
        }
    };
    auto lstore = [&](char* dK, int t0) {
        char* dV = dK + 64 * KROW;
#pragma unroll
        for (int i = 0; i < KCH; ++i)
            if (kin[i]) *reinterpret_cast<uint4*>(dK + klds[i]) = rk[i];
        if (t0 + 64 <= hot_t) {
#pragma unroll
            for (int i = 0; i < VCH; ++i)
                if (vin[i]) *reinterpret_cast<uint4*>(dV + vlds[i]) = rv[i];
        } else {
#pragma unroll
            for (int i = 0; i < VCH; ++i) {
                if (!vin[i]) continue;
                uint4 v = rv[i];
                const int valid = hot_t - (t0 + vkey[i]);  // keys >= t are padding of unspecified content: force to 0
                if (valid < 8) {
                    uint32_t* u = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (2 * j >= valid) u[j] = 0;
                        else if (2 * j + 1 >= valid) u[j] &= 0xFFFFu;
                    }
                }
                *reinterpret_cast<uint4*>(dV + vlds[i]) = v;
            }
        }
    };

    if (PIPE && NEEDS_ZERO) __syncthreads();  // zero fill done (a __syncthreads also waits for every outstanding load: keep it AHEAD of them)
    // Q fragments and K/V tile 0: issued together, one memory round trip
    bf16x8 qf[QF][KS];
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        int qrow = q0 + f * 16 + r;
        if (qrow > hot_s - 1) qrow = hot_s - 1;
        const bf16_t* qp = hot_q + ((size_t)b * hot_s + qrow) * hot_q_ld + h * D;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d0 = ks * 32 + 8 * g;
            if (d0 < D) qf[f][ks] = *reinterpret_cast<const bf16x8*>(qp + d0);
            else qf[f][ks] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
    }

    if (PIPE) {
        gload(0);
        lstore(smem, 0);
    }
    // Retire the Q loads HERE, before the tile loop (otherwise hipcc's wait for them lands on their first use inside
    // the loop as vmcnt(0), which also drains the K/V prefetch issued just before it every iteration) — but after the
    // loads of K/V tile 0 were issued, so the two round trips overlap.
#pragma unroll
    for (int f = 0; f < QF; ++f)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[f][ks]));

    // ---- pieces of one 64-key tile --------------------------------------------------------------
    // S^T block kf (16 keys x all queries of the wave) = K Q^T [- m_ref]
    auto qk_block = [&](f32x4 (&s)[4][QF], const char* sK, int kf) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 kfrag = *reinterpret_cast<const bf16x8*>(sK + (kf * 16 + r) * KROW + ks * 64 + g * 16);
#pragma unroll
            for (int f = 0; f < QF; ++f)
                s[kf][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    kfrag, qf[f][ks], ks == 0 ? (PRESC ? negm[f] : (f32x4){0, 0, 0, 0}) : s[kf][f], 0, 0, 0);
        }
    };
    // masks, tile maximum, and the (rare) move of the reference maximum.  Lane holds keys t0 + kf*16 + 4g + e of query
    // f*16 + r.  generic: scores stay RAW in the accumulators, p = exp2(s*c - m_ref*c), c = scale*log2(e), one fma per
    // score; PRESC: the accumulators already hold s*c - m_ref, p = exp2(acc).
    auto tile_prepare = [&](f32x4 (&s)[4][QF], int tile) {
        const int t0 = tile * 64;
        // (the key index is made opaque INSIDE each branch: otherwise hipcc hoists its 16 adds out of both branches into
        //  every tile's straight-line path — 64 issue cycles of ~750 — although self-attention tiles take neither branch)
        if (t0 + 64 > hot_t) {  // ragged key tail (text context): only this tile pays for the masking
            int key0 = t0 + 4 * g;
            asm volatile("" : "+v"(key0));
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (key0 + kf * 16 + e >= hot_t) {
#pragma unroll
                        for (int f = 0; f < QF; ++f) s[kf][f][e] = -1e30f;
                    }
        }
        if (p.causal) {       // every query keeps key 0, so no row is ever fully masked
            int key0 = t0 + 4 * g;
            asm volatile("" : "+v"(key0));
#pragma unroll
            for (int f = 0; f < QF; ++f) {
                const int qi = q0 + f * 16 + r;
#pragma unroll
                for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (key0 + kf * 16 + e > qi) s[kf][f][e] = -1e30f;
            }
        }
        // tile maximum per query: a v_max3 chain over the lane's 16 scores, then across the four 16-lane rows
        float mx[QF];
        bool need[QF], any = false;
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            float m = fmaxf(s[0][f][0], s[0][f][1]);
            m = fmaxf(fmaxf(m, s[0][f][2]), s[0][f][3]);
#pragma unroll
            for (int kf = 1; kf < 4; ++kf) {
                m = fmaxf(fmaxf(m, s[kf][f][0]), s[kf][f][1]);
                m = fmaxf(fmaxf(m, s[kf][f][2]), s[kf][f][3]);
            }
            mx[f] = rows_max4(m);
            need[f] = tile == 0 || (PRESC ? (mx[f] > ATTN_THR) : ((mx[f] - mref[f]) * p.sl2 > ATTN_THR));
            any = any || need[f];
        }
        if (__builtin_amdgcn_ballot_w64(any) != 0) {   // wave-uniform; after the first tile: rare (see ATTN_THR)
            // Only the queries that need it move their reference: for the others delta = 0 / alpha = 1 exactly, so what a
            // query's result is does not depend on which other queries share its wave (64- vs 128-query workgroups are
            // picked by grid size, i.e. by batch: results must not depend on how a batch is sharded).
#pragma unroll
            for (int f = 0; f < QF; ++f) {
                float alpha;
                if (PRESC) {
                    const float delta = need[f] ? mx[f] : 0.f;   // (after tile 0: > ATTN_THR, so m_ref only grows)
                    alpha = __builtin_amdgcn_exp2f(-delta);
                    mref[f] += delta;
                    negm[f] = (f32x4){-mref[f], -mref[f], -mref[f], -mref[f]};
#pragma unroll
                    for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                        for (int e = 0; e < 4; ++e) s[kf][f][e] -= delta;
                } else {
                    const float mnew = need[f] ? mx[f] : mref[f];   // (need: mx > m_ref + THR / scale, or tile 0)
                    alpha = __builtin_amdgcn_exp2f((mref[f] - mnew) * p.sl2);
                    mref[f] = mnew;
                }
                if (tile != 0) {   // (tile 0: O and the row sum are still zero; alpha would be exp2(-first maximum))
                    lrun[f] *= alpha;
#pragma unroll
                    for (int df = 0; df < DF; ++df) {
                        oacc[df][f][0] *= alpha; oacc[df][f][1] *= alpha; oacc[df][f][2] *= alpha; oacc[df][f][3] *= alpha;
                    }
                }
            }
        }
    };
    // exponentials of S^T block kf, in place
    auto exp_block = [&](f32x4 (&s)[4][QF], int kf) {
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            const float nm = -mref[f] * p.sl2;
            float ls = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pv = __builtin_amdgcn_exp2f(PRESC ? s[kf][f][e] : fmaf(s[kf][f][e], p.sl2, nm));
                s[kf][f][e] = pv;
                if (!ONES_ROW) ls += pv;
            }
            if (!ONES_ROW) lrun[f] += ls;
        }
    };
    // O^T += V^T P^T over the 32 keys of S^T blocks 2kk, 2kk+1 (P packed to bf16 = the MFMA's B fragment)
    auto pv_half = [&](f32x4 (&s)[4][QF], const char* sV, int kk) {
        bf16x8 pb[QF];
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            union { bf16x8 v; uint32_t u[4]; } pk;
            pk.u[0] = pack_bf2(s[2 * kk][f][0], s[2 * kk][f][1]);
            pk.u[1] = pack_bf2(s[2 * kk][f][2], s[2 * kk][f][3]);
            pk.u[2] = pack_bf2(s[2 * kk + 1][f][0], s[2 * kk + 1][f][1]);
            pk.u[3] = pack_bf2(s[2 * kk + 1][f][2], s[2 * kk + 1][f][3]);
            pb[f] = pk.v;
        }
#pragma unroll
        for (int df = 0; df < DF; ++df) {
            union { bf16x8 v; uint2 h2[2]; } vf;
            const char* vp = sV + (df * 16 + r) * VROW + kk * 64 + g * 8;
            vf.h2[0] = *reinterpret_cast<const uint2*>(vp);
            vf.h2[1] = *reinterpret_cast<const uint2*>(vp + 32);
#pragma unroll
            for (int f = 0; f < QF; ++f)
                oacc[df][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pb[f], oacc[df][f], 0, 0, 0);
        }
    };

    for (int tile = 0; tile < ntiles; ++tile) {
        const int t0 = tile * 64;
        char* sK = smem + (PIPE ? (tile & 1) : 0) * BUF_BYTES;
        if (PIPE) {
            __syncthreads();  // tile `tile` staged by every wave; the other buffer no longer read by anyone
            if (tile + 1 < ntiles) gload(t0 + 64);  // next tile's loads fly while this tile is computed
        } else {
            __syncthreads();  // previous tile fully consumed (also orders the zero fill on the first pass)
            gload(t0);
            lstore(sK, t0);
            __syncthreads();
        }
        f32x4 sacc[4][QF];
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) qk_block(sacc, sK, kf);
        tile_prepare(sacc, tile);
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) exp_block(sacc, kf);
        pv_half(sacc, sK + 64 * KROW, 0);
        pv_half(sacc, sK + 64 * KROW, 1);
        if (PIPE && tile + 1 < ntiles) lstore(smem + ((tile & 1) ^ 1) * BUF_BYTES, t0 + 64);
    }

#pragma unroll
    for (int f = 0; f < QF; ++f) {
        float lt;
        if (ONES_ROW) {
            // row D of V^T is all ones, so O^T[D][q] accumulated sum_k P[k][q] on the matrix core
            // (same alpha rescaling as O, same bf16-rounded P as the numerator); it sits in lane
            // group g = (D%16)/4, element (D%4) of the last d-block: broadcast it to the other groups
            lt = __shfl(oacc[DF - 1][f][D % 4], (lane & 15) + 16 * ((D % 16) / 4));
        } else {
            lt = lrun[f];
            lt += __shfl_xor(lt, 16);
            lt += __shfl_xor(lt, 32);
        }
        const float inv = 1.0f / lt;
        const int qrow = q0 + f * 16 + r;
        if (qrow < hot_s) {
            bf16_t* op = p.out + ((size_t)b * hot_s + qrow) * p.o_ld + h * D;
#pragma unroll
            for (int df = 0; df < DF; ++df) {
                const int d = df * 16 + 4 * g;
                if (d < D) {
                    uint2 o;
                    o.x = pack_bf2(oacc[df][f][0] * inv, oacc[df][f][1] * inv);
                    o.y = pack_bf2(oacc[df][f][2] * inv, oacc[df][f][3] * inv);
                    *reinterpret_cast<uint2*>(op + d) = o;
                }
            }
        }
    }
}

// ---- the same attention on 32x32x16 MFMAs --------------------------------------------------------------------------------
// The 16x16 form above is bound by instruction ISSUE, not by the matrix pipe (a v_mfma_f32_16x16x32_bf16 holds the SIMD's
// issue port for 8 of its 16 cycles, and per 64-key tile a wave issues 28 of them beside 32 v_exp_f32, the packs and the
// maximum chain: ~840 issue cycles per wave and tile).  A 32x32x16 MFMA does twice the work per issued instruction (8 of 32
// cycles), and its k-dimension of 16 fits the head sizes without the 16x16x32 form's padding: d = 40 -> 48 instead of 64 in
// QK^T.  (tools/mfma_rate.py: the older 16x16x16 shape, which would pad d = 40 to 48 as well, still takes 16 cycles on gfx950 — no
// gain there.)  Per 64-key tile and wave: 6 + 8 MFMAs instead of 16 + 12 at d = 40, 10 + 12 instead of 24 + 20 at d = 80.
//
// A wave owns 32 queries = the 32 MFMA columns; lane (c = lane & 31, h = lane >> 5) works on query c and holds, of a 32-key
// S^T block, the 16 keys (i & 3) + 8 (i >> 2) + 4 h, i = 0..15.  Everything said above about the transposed products holds:
// the query's maximum / sum / rescale are lane-local, the row maximum needs ONE half-wave swap, and the S^T registers
// 8 s' .. 8 s' + 7 ARE the B operand of O^T += V^T P^T for the 16 keys of step s' once packed to bf16 — in the k-slot order
// {4h + b, 8 + 4h + b}, which the V^T tile is laid out for in LDS when it is staged (two 8-byte pieces per 16-byte chunk), so
// that an A fragment is one ds_read_b128.  The maximum is checked per LANE first: only when some lane of the wave holds a score
// above the threshold (or on tile 0) is the true row maximum formed across the two half-waves — the same decisions, the
// cross-lane step only on the rare path.
// NW = waves per workgroup (4: 128 queries; 2: 64 queries, for launches whose 128-query grid leaves CUs idle).
typedef __attribute__((ext_vector_type(16))) float f32x16;

// In-kernel phase timeline of attention32_kernel (tools/attn_stamps.py; only in the `make stamps` library): wave 0 of every
// workgroup sums, over its tiles, the shader-clock time between fixed points of the tile loop.  Each stamp is tied to a value
// the phase before it produced (so it cannot move above that phase's last instruction) and is a compiler memory barrier.
#ifdef MSD_STAMPS
#ifndef MSD_ASTAMP_MASK
#define MSD_ASTAMP_MASK 0xFF
#endif
#define MSD_ASTAMP_WAVES 1
static __device__ unsigned long long g_astamps[16 * 4096];
extern "C" MSD_API int msd_debug_stamps_attn(unsigned long long* host_out, int count) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_astamps), sizeof(unsigned long long) * (size_t)count);
}
// (s_memtime returns through the scalar data cache: the stamps of a tile are only waited for — lgkmcnt — at the tile's end,
// so that a stamp does not drain the wave's LDS queue in the middle of the code it measures)
#define ASTAMP(i, x)                                                                                   \
    do {                                                                                               \
        if ((MSD_ASTAMP_MASK >> (i)) & 1) asm volatile("s_memtime %0" : "=s"(st_t[i]), "+v"(x) : : "memory"); \
    } while (0)
// ORDER: the stamps of one tile in program order, as a hex string of digits, e.g. 0x012345 (most significant first)
#define ASTAMP_END(ORDER, N)                                                                           \
    do {                                                                                               \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(st_t[0]), "+s"(st_t[1]), "+s"(st_t[2]), "+s"(st_t[3]), "+s"(st_t[4]), "+s"(st_t[5]), "+s"(st_t[6]), "+s"(st_t[7]) : : "memory"); \
        _Pragma("unroll") for (int k_ = (N) - 1; k_ >= 0; --k_) {                                      \
            const int i_ = ((ORDER) >> (4 * k_)) & 15;                                                 \
            if ((MSD_ASTAMP_MASK >> i_) & 1) { st_acc[i_] += st_t[i_] - st_prev; st_prev = st_t[i_]; }  \
        }                                                                                              \
    } while (0)
#else
#define ASTAMP(i, x)
#define ASTAMP_END(ORDER, N)
#endif

// loader waves of the software-pipelined form (NBUF 4) beside its NW compute waves.  One wave gets an LDS-DMA instruction out
// every 75-85 cycles whatever else it does (its priority, scalar lane masks instead of branches: no change), waves issue in
// parallel; a tile is 13 DMAs at d = 40 and 23 at d = 80, and the compute waves are through one in 1000-1600 cycles.
__host__ __device__ constexpr int attn32_loaders(int nbuf, int nw, int d) { return nbuf == 4 ? ((nw >= 8 || d > 64) ? 2 : 1) : 0; }

template <int D, int NBUF, int NW, bool PRESC>
__global__ __launch_bounds__(64 * (NW + attn32_loaders(NBUF, NW, D))) void attention32_kernel(ATTN_HOT_PARAMS, const AArgs p) {
    constexpr bool SWP = NBUF == 4;          // software-pipelined tile loop (below)
    // SWP, prescaled q, head size with a padded k-step (d = 40 -> 48): the reference maximum rides in the padding — channel
    // D of every K row is 1, channel D of the query holds -m_ref (a bf16 value; any reference works as long as every use
    // agrees) — instead of in 16 accumulator-start registers
    constexpr bool PADREF = SWP && PRESC && (D % 16 == 8);
    constexpr bool PIPE = NBUF == 2;
    constexpr int NT = 64 * NW, QT = 32 * NW;
    constexpr int KS = (D + 15) / 16;        // 16-channel k-steps of QK^T
    constexpr int DPAD = KS * 16;
    constexpr int DB = (D + 31) / 32;        // 32-row blocks of O^T
    constexpr int KROW = DPAD * 2 + 16;      // bytes; an odd number of 16-byte slots -> conflict-free ds_read_b128 fragments
    constexpr int VROW = 64 * 2 + 16;
    constexpr int DCH = D / 8;               // 16-byte chunks per K row
    constexpr int BUF_BYTES = 64 * KROW + DB * 32 * VROW;   // K tile [64][KROW] + V^T tile [DB*32][VROW] (keys permuted per 16)
    constexpr bool ONES_ROW = (D % 32) != 0;  // a spare padding row of the V^T tile carries the softmax denominator
    static_assert(((KROW / 16) & 1) == 1 && ((VROW / 16) & 1) == 1, "row strides must be odd multiples of 16 bytes");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int qtiles = (hot_q_count + QT - 1) / QT;
    const int q_end = hot_q_begin + hot_q_count;   // (<= hot_s)
    const int wi = xcd_remap(blockIdx.x, qtiles * hot_heads * hot_batch);
    const int bh = udiv_magic(wi, qtiles, hot_mg_qtiles);
    const int b = udiv_magic(bh, hot_heads, hot_mg_heads), hd = bh - b * hot_heads;
    const int q0 = hot_q_begin + (wi - bh * qtiles) * QT + wave * 32;

    if constexpr (SWP) {
        // The DMAs write every data slot of a tile image before it is read; only what they leave alone needs a value: the pad
        // slots of the K rows (zeros; PADREF: 1.0 in channel D) and the V^T rows >= D (row D: ones, the denominator's row).
        constexpr int KP = KROW / 16 - DCH;           // pad slots per K row (the last one is never read)
        constexpr int VR = DB * 32 - D, VS = VROW / 16;
        constexpr int PER = 64 * KP + VR * VS;        // 16-byte pieces per ring slot
        const int nthreads = 64 * (NW + attn32_loaders(NBUF, NW, D));
        for (int idx = tid; idx < NBUF * PER; idx += nthreads) {
            const int bufi = idx / PER, r = idx - bufi * PER;
            char* base = smem + bufi * BUF_BYTES;
            if (r < 64 * KP) {
                const int row = r / KP, sl = r - row * KP;
                *reinterpret_cast<uint4*>(base + row * KROW + (DCH + sl) * 16) = make_uint4((PADREF && sl == 0) ? 0x3F80u : 0u, 0, 0, 0);
            } else {
                const int rr = r - 64 * KP, row = rr / VS, sl = rr - row * VS;
                const uint32_t v = (ONES_ROW && row == 0) ? 0x3F803F80u : 0u;
                *reinterpret_cast<uint4*>(base + 64 * KROW + (D + row) * VROW + sl * 16) = make_uint4(v, v, v, v);
            }
        }
    } else {
    for (int off = tid * 16; off < NBUF * BUF_BYTES; off += NT * 16)
        *reinterpret_cast<uint4*>(smem + off) = make_uint4(0, 0, 0, 0);
    if (ONES_ROW) {
        __syncthreads();
        if (tid < 16) {  // 64 keys x bf16(1.0) in row D of each V^T buffer (never overwritten: tiles write rows < D)
#pragma unroll
            for (int bufi = 0; bufi < NBUF; ++bufi)
                *reinterpret_cast<uint2*>(smem + bufi * BUF_BYTES + 64 * KROW + D * VROW + tid * 8) = make_uint2(0x3F803F80u, 0x3F803F80u);
        }
    }
    }

    f32x16 oacc[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[db][i] = 0.f;
    float mref = 0.f, lrun = 0.f;
    f32x16 negm;   // PRESC: {-m_ref x 16}, the C operand that starts every S^T accumulator chain
#pragma unroll
    for (int i = 0; i < 16; ++i) negm[i] = 0.f;

    const bf16_t* kbase = hot_k + (size_t)b * hot_t * hot_k_ld + hd * D;
    const bf16_t* vbase = hot_vt + ((size_t)b * hot_heads + hd) * D * p.vt_ld;
    const int ntiles = (hot_t + 63) / 64;
    constexpr int KCH = (64 * DCH + NT - 1) / NT, VCH = (D * 8 + NT - 1) / NT;
    uint4 rk[KCH], rv[VCH];
    const bf16_t* kptr[KCH];
    const bf16_t* vptr[VCH];
    int klds[KCH], vlds[VCH], krow[KCH], vkey[VCH];
    bool kin[KCH], vin[VCH];
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
        const int idx = tid + NT * i;
        kin[i] = idx < 64 * DCH;
        const int row = kin[i] ? idx / DCH : 0, ch = kin[i] ? idx - row * DCH : 0;
        krow[i] = row;
        kptr[i] = kbase + (size_t)row * hot_k_ld + ch * 8;
        klds[i] = row * KROW + ch * 16;
    }
#pragma unroll
    for (int i = 0; i < VCH; ++i) {
        const int idx = tid + NT * i;
        vin[i] = idx < D * 8;
        const int d = vin[i] ? idx >> 3 : 0, ch = idx & 7;
        vkey[i] = ch * 8;
        vptr[i] = vbase + (size_t)d * p.vt_ld + ch * 8;
        // keys 8 ch + {0..3} -> positions 16 (ch >> 1) + 4 (ch & 1) + {0..3}, keys 8 ch + 4 + {0..3} -> 8 further on
        vlds[i] = d * VROW + ((ch >> 1) * 16 + (ch & 1) * 4) * 2;
    }
    auto gload = [&](int t0) {
        const size_t koff = (size_t)t0 * hot_k_ld;
        if (t0 + 64 <= hot_t) {
#pragma unroll
            for (int i = 0; i < KCH; ++i)
                if (kin[i]) rk[i] = *reinterpret_cast<const uint4*>(kptr[i] + koff);
#pragma unroll
            for (int i = 0; i < VCH; ++i)
                if (vin[i]) rv[i] = *reinterpret_cast<const uint4*>(vptr[i] + t0);
        } else {
#pragma unroll
            for (int i = 0; i < KCH; ++i) {
                rk[i] = make_uint4(0, 0, 0, 0);
                if (kin[i] && t0 + krow[i] < hot_t) rk[i] = *reinterpret_cast<const uint4*>(kptr[i] + koff);
            }
#pragma unroll
            for (int i = 0; i < VCH; ++i) {
                rv[i] = make_uint4(0, 0, 0, 0);
                const int key0 = t0 + vkey[i];
                if (vin[i] && key0 + 8 <= p.vt_ld && key0 < hot_t) rv[i] = *reinterpret_cast<const uint4*>(vptr[i] + t0);
            }
        }
    };
    auto lstore = [&](char* dK, int t0) {
        char* dV = dK + 64 * KROW;
#pragma unroll
        for (int i = 0; i < KCH; ++i)
            if (kin[i]) *reinterpret_cast<uint4*>(dK + klds[i]) = rk[i];
#pragma unroll
        for (int i = 0; i < VCH; ++i) {
            if (!vin[i]) continue;
            uint4 v = rv[i];
            if (t0 + 64 > hot_t) {   // ragged last tile: keys >= t are padding of unspecified content, force them to 0
                const int valid = hot_t - (t0 + vkey[i]);
                if (valid < 8) {
                    uint32_t* u = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (2 * j >= valid) u[j] = 0;
                        else if (2 * j + 1 >= valid) u[j] &= 0xFFFFu;
                    }
                }
            }
            *reinterpret_cast<uint2*>(dV + vlds[i]) = make_uint2(v.x, v.y);
            *reinterpret_cast<uint2*>(dV + vlds[i] + 16) = make_uint2(v.z, v.w);
        }
    };

    if (PIPE) __syncthreads();
    bf16x8 qf[KS];
    {
        int qrow = q0 + c;
        if (qrow > hot_s - 1) qrow = hot_s - 1;
        const bf16_t* qp = hot_q + ((size_t)b * hot_s + qrow) * hot_q_ld + hd * D;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d0 = ks * 16 + 8 * h;
            if (d0 < D) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + d0);
            else qf[ks] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    if (PIPE) {
        gload(0);
        lstore(smem, 0);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));   // retire the Q loads before the tile loop (see above)
#ifdef MSD_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev, st_first;
    const unsigned long long st_wall0 = wall_clock64();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev) : : "memory");
    st_first = st_prev;
#endif

    if constexpr (SWP) {
    // ---- software-pipelined loop on LDS-DMA staging.  The plain loop below runs QK^T(t) -> maximum -> exponentials -> PV(t)
    // as one dependent chain per tile and stages the next tile through registers.  tools/attn_stamps.py on it (S = 4096,
    // d = 40, two waves per SIMD): 2270 cycles per wave and tile against 448 cycles of matrix pipe; ~450 of them go to ISSUING
    // the four ds_write of a tile (all waves of the CU store right after their barrier; the VGPR -> LDS path backs up) and
    // ~700 to compiler-placed vmcnt(0) waits between the predicated blocks of the staging code.  Here
    //   * K and V^T tiles go global -> LDS by DMA (global_load_lds_dwordx4: no staging registers, no ds_write, nothing
    //     for the compiler's counter model to see); the 1 KiB pieces of a tile image are dealt round-robin to the waves,
    //     lanes that fall on a row's pad slot are switched off (the pad keeps the zeros of the initial fill);
    //   * a LOADER wave (the workgroup's last) issues them into a ring of four slots, two tiles ahead of the products;
    //   * the products of tile t + 1 are ISSUED before the softmax of tile t (two score register sets, swapped by
    //     unrolling twice): the matrix pipe works on S^T(t + 1) while the VALU takes the maximum of S^T(t);
    //   * the K rows are fed to the MFMA in the order that makes a lane's 8 score registers of a PV step 8 CONSECUTIVE keys
    //     (row index bits 2 and 3 swapped: register i of lane (c, h) is key (i & 7) + 8 h + 16 (i >> 3) of the 32-key block),
    //     so the V^T image is the tensor's own layout and DMA can write it;
    //   * the tiles that need masks (ragged tail, causal) run a second instance of the step: the common one is three
    //     basic blocks, [QK^T(t + 1) + maximum(t)] [rare rescale] [exponentials(t) + PV(t)].
    // A ragged last tile stages its V^T piece through registers (keys >= t are padding of unspecified content and must
    // be stored as 0); its K rows >= t read the last valid row (finite; their scores are masked).
    constexpr int KSL = KROW / 16, VSL = VROW / 16;       // 16-byte slots per image row (data + pad)
    constexpr int KI = KSL, VI = (D * VSL + 63) / 64;     // DMA wave-instructions (1 KiB pieces) per K / V^T tile image
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // the initial fill of the ring is in place before any DMA may land on it
    if (wave_u >= NW) {
        // ---- the loader waves: nothing but the staging of every tile.  (Issued by the compute waves themselves, right after
        // their barrier, the 13 DMAs of a d = 40 tile kept each of them 600 cycles at the issue port, and that is time
        // their MFMAs do not get.)  Loader l takes the pieces l, l + NL, ... of each image.
        constexpr int NL = attn32_loaders(NBUF, NW, D);
        constexpr int NKL = (KI + NL - 1) / NL, NVL = (VI + NL - 1) / NL;
        const int ldr = wave_u - NW;
        const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem + (uint32_t)(ldr * 1024);
        uint32_t kvo[NKL], vvo[NVL];   // per-lane source byte offsets of the pieces
        uint64_t kon[NKL], von[NVL];   // ... and their lane masks (off: pad slots, lanes past the image)
#pragma unroll
        for (int n = 0; n < NKL; ++n) {
            const int q = ldr + NL * n;
            const int sidx = 64 * q + lane, row = sidx / KSL, ch = sidx - row * KSL;
            const bool on = q < KI && ch < DCH;
            kvo[n] = on ? (uint32_t)(row * hot_k_ld * 2 + ch * 16) : 0u;
            kon[n] = __builtin_amdgcn_ballot_w64(on);
        }
#pragma unroll
        for (int n = 0; n < NVL; ++n) {
            const int q = ldr + NL * n;
            const int sidx = 64 * q + lane, row = sidx / VSL, ch = sidx - row * VSL;
            const bool on = q < VI && ch < 8 && row < D;
            vvo[n] = on ? (uint32_t)(row * p.vt_ld * 2 + ch * 16) : 0u;
            von[n] = __builtin_amdgcn_ballot_w64(on);
        }
        const size_t kstep = (size_t)hot_k_ld * 128;          // bytes per 64-key tile
        const char* kb_next = reinterpret_cast<const char*>(kbase);   // source of the next tile to stage
        const char* vb_next = reinterpret_cast<const char*>(vbase);
        int t0_next = 0;
        auto stage = [&](int slot) {   // the next tile -> ring slot; returns with the DMAs in flight
            const uint32_t dst0 = lds0 + (uint32_t)(slot * BUF_BYTES);
            if (t0_next + 64 <= hot_t) {
#pragma unroll
                for (int n = 0; n < NKL; ++n)
                    if (kon[n]) dma16sm(kb_next, kvo[n], dst0 + (uint32_t)(n * NL * 1024), kon[n]);
#pragma unroll
                for (int n = 0; n < NVL; ++n)
                    if (von[n]) dma16sm(vb_next, vvo[n], dst0 + (uint32_t)(64 * KROW + n * NL * 1024), von[n]);
            } else {
                const int t0 = t0_next;
                const uint32_t klim = (uint32_t)(((hot_t - 1 - t0) * hot_k_ld + (D - 8)) * 2);
#pragma unroll
                for (int n = 0; n < NKL; ++n)
                    if (kon[n]) dma16sm(kb_next, min(kvo[n], klim), dst0 + (uint32_t)(n * NL * 1024), kon[n]);
                typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
                char* dV = smem + slot * BUF_BYTES + 64 * KROW;
                const uint32_t clim = (uint32_t)(((hot_t - 1 - t0) >> 3) * 16);
                for (int idx = ldr * 64 + lane; idx < D * 8; idx += 64 * NL) {
                    const int d = idx >> 3, ch = idx & 7;
                    u32x4_t v = *reinterpret_cast<const u32x4_t*>(vb_next + ((size_t)d * p.vt_ld * 2 + min((uint32_t)(ch * 16), clim)));
                    const int valid = hot_t - (t0 + ch * 8);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (2 * j >= valid) v[j] = 0;
                        else if (2 * j + 1 >= valid) v[j] &= 0xFFFFu;
                    }
                    *reinterpret_cast<u32x4_t*>(dV + d * VROW + ch * 16) = v;
                }
                wait_vmcnt<0>();
            }
            kb_next += kstep;
            vb_next += 128;
            t0_next += 64;
        };
        // Ring of 4 slots, two tiles ahead: tile + 3 is issued right after barrier `tile` (its slot held tile - 1, whose last
        // reader passed that barrier), tile + 1 has to be in LDS by barrier `tile`: at that point only the pieces of tile + 2
        // may still be in flight — a counted wait, exact because every piece of a full tile is one DMA instruction of this
        // wave (a ragged tile is the last one and is waited for in full when it is staged).
        auto wait_newest = [&]() {   // everything but the pieces of the tile staged last (a full one) has landed
            if (NL == 1 || ldr == 0) wait_vmcnt<(KI + NL - 1) / NL + (VI + NL - 1) / NL>();
            else wait_vmcnt<KI / NL + VI / NL>();   // (NL == 2: loader 1 has the odd pieces)
        };
        static_assert(NL <= 2, "wait_newest counts the pieces of loader 0 and loader 1");
#ifdef MSD_STAMPS
        unsigned long long pt0, pt1, pt2;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pt0) : : "memory");
#endif
        stage(0);
        if (ntiles > 1) stage(1);
        if (ntiles > 2) stage(2);
#ifdef MSD_STAMPS
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pt1) : : "memory");
#endif
        if (ntiles > 2) wait_newest(); else wait_vmcnt<0>();
#ifdef MSD_STAMPS
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pt2) : : "memory");
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#ifdef MSD_STAMPS
        unsigned long long lt[4] = {0, 0, 0, 0}, la[3] = {0, 0, 0};
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(lt[0]) : : "memory");
#define LSTAMP(i) do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(lt[(i) + 1]) : : "memory"); la[i] += lt[(i) + 1] - lt[i]; } while (0)
#else
#define LSTAMP(i)
#endif
        for (int tile = 0; tile < ntiles; ++tile) {
            if (tile + 2 < ntiles) wait_newest(); else wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            LSTAMP(0);
            __builtin_amdgcn_s_barrier();
            LSTAMP(1);
            if (tile + 3 < ntiles) stage((tile + 3) & 3);
            LSTAMP(2);
#ifdef MSD_STAMPS
            lt[0] = lt[3];
#endif
        }
#ifdef MSD_STAMPS
        if (lane == 0 && ldr == 0) {
            unsigned long long* dst = g_astamps + (size_t)(blockIdx.x & 2047) * 16 + 12;
            dst[0] = la[0]; dst[1] = la[1]; dst[2] = la[2]; dst[3] = ((pt1 - pt0) << 32) | (pt2 - pt1);
        }
#endif
        return;
    }
    const int cperm = (c & ~12) | ((c & 4) << 1) | ((c & 8) >> 1);   // K row this lane feeds to MFMA row c
    auto qk = [&](f32x16 (&sc)[2], const char* sK) {
        bf16x8 kfrag[2][KS];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                kfrag[kb][ks] = *reinterpret_cast<const bf16x8*>(sK + (kb * 32 + cperm) * KROW + ks * 32 + h * 16);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {   // (the two blocks' chains alternate: no MFMA waits for the one before it)
                if (ks == 0) {
                    if (PRESC && !PADREF) sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfrag[kb][0], qf[0], negm, 0, 0, 0);
                    else {
                        f32x16 z;
#pragma unroll
                        for (int i = 0; i < 16; ++i) z[i] = 0.f;
                        sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfrag[kb][0], qf[0], z, 0, 0, 0);
                    }
                } else {
                    sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfrag[kb][ks], qf[ks], sc[kb], 0, 0, 0);
                }
            }
    };
    // one tile: `cur` holds S^T(tile) (issued one step earlier), `nxt` receives S^T(tile + 1); slot = ring slot of `tile`
    auto step = [&](auto masked_, f32x16 (&cur)[2], f32x16 (&nxt)[2], int tile, int slot) {
        constexpr bool MASKED = decltype(masked_)::value;
        const int t0 = tile * 64;
        const int slot1 = (slot + 1) & 3;
        __builtin_amdgcn_s_barrier();          // tile + 1 is in LDS (the loader waited for it); everybody is done with tile - 1
        ASTAMP(0, mref);
        qk(nxt, smem + slot1 * BUF_BYTES);   // (past the last tile: a slot of finite leftovers, the result is never used)
        if (MASKED) {
            if (t0 + 64 > hot_t) {
                int key0 = t0 + 8 * h;
                asm volatile("" : "+v"(key0));
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if (key0 + kb * 32 + (i & 7) + 16 * (i >> 3) >= hot_t) cur[kb][i] = -1e30f;
            }
            if (p.causal) {
                int key0 = t0 + 8 * h;
                asm volatile("" : "+v"(key0));
                const int qi = q0 + c;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if (key0 + kb * 32 + (i & 7) + 16 * (i >> 3) > qi) cur[kb][i] = -1e30f;
            }
        }
        // ---- lane-local maximum; the row maximum and the (rare) move of the reference only when some lane asks for it
        float m = fmaxf(cur[0][0], cur[0][1]);
#pragma unroll
        for (int i = 2; i < 16; i += 2) m = fmaxf(fmaxf(m, cur[0][i]), cur[0][i + 1]);
#pragma unroll
        for (int i = 0; i < 16; i += 2) m = fmaxf(fmaxf(m, cur[1][i]), cur[1][i + 1]);
        ASTAMP(2, m);
        const bool ask = tile == 0 || (PRESC ? (m > ATTN_THR) : ((m - mref) * p.sl2 > ATTN_THR));
        if (__builtin_amdgcn_ballot_w64(ask) != 0) {
            const uint32_t u = __float_as_uint(m);
            auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            float mx;
            asm("v_max_f32 %0, %1, %2" : "=v"(mx) : "v"(sw[0]), "v"(sw[1]));   // both half-waves: the query's maximum over the tile
            const bool need = tile == 0 || (PRESC ? (mx > ATTN_THR) : ((mx - mref) * p.sl2 > ATTN_THR));
            float alpha;
            if (PADREF) {
                // the new reference is a bf16 value; the move actually applied is the difference of the two references
                const uint32_t rb = pack_bf2(mref + (need ? mx : 0.f), 0.f) & 0xFFFFu;
                const float rnew = __uint_as_float(rb << 16), delta = rnew - mref;
                alpha = __builtin_amdgcn_exp2f(-delta);
                mref = rnew;
                if (h == 1) {   // channel D = k-slot 8 of the last k-step: element 0 of the upper half-wave's fragment
                    union { bf16x8 v; uint32_t u[4]; } qq;
                    qq.v = qf[KS - 1];
                    qq.u[0] = (qq.u[0] & 0xFFFF0000u) | (rb ^ 0x8000u);
                    qf[KS - 1] = qq.v;
                }
                // (S^T(tile + 1) is already under way against the old reference: it moves too)
#pragma unroll
                for (int i = 0; i < 16; ++i) { cur[0][i] -= delta; cur[1][i] -= delta; nxt[0][i] -= delta; nxt[1][i] -= delta; }
            } else if (PRESC) {
                const float delta = need ? mx : 0.f;
                alpha = __builtin_amdgcn_exp2f(-delta);
                mref += delta;
                // (S^T(tile + 1) is already under way against the old reference: it moves too)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    negm[i] = -mref; cur[0][i] -= delta; cur[1][i] -= delta; nxt[0][i] -= delta; nxt[1][i] -= delta;
                }
            } else {
                const float mnew = need ? mx : mref;
                alpha = __builtin_amdgcn_exp2f((mref - mnew) * p.sl2);
                mref = mnew;
            }
            if (tile != 0) {
                lrun *= alpha;
#pragma unroll
                for (int db = 0; db < DB; ++db)
#pragma unroll
                    for (int i = 0; i < 16; ++i) oacc[db][i] *= alpha;
            }
        }
        // ---- exponentials in place
        {
            const float nm = -mref * p.sl2;
            float ls = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float pv = __builtin_amdgcn_exp2f(PRESC ? cur[kb][i] : fmaf(cur[kb][i], p.sl2, nm));
                    cur[kb][i] = pv;
                    if (!ONES_ROW) ls += pv;
                }
            if (!ONES_ROW) lrun += ls;
        }
        ASTAMP(3, cur[1][15]);
        // ---- O^T += V^T P^T, 16 keys per step: registers 8 (st & 1) .. + 7 of block st >> 1 are keys 16 st + 8 h + 0..7
        const char* sV = smem + slot * BUF_BYTES + 64 * KROW;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int kb = st >> 1, i0 = (st & 1) * 8;
            union { bf16x8 v; uint32_t u[4]; } pk;
            pk.u[0] = pack_bf2(cur[kb][i0 + 0], cur[kb][i0 + 1]);
            pk.u[1] = pack_bf2(cur[kb][i0 + 2], cur[kb][i0 + 3]);
            pk.u[2] = pack_bf2(cur[kb][i0 + 4], cur[kb][i0 + 5]);
            pk.u[3] = pack_bf2(cur[kb][i0 + 6], cur[kb][i0 + 7]);
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sV + (db * 32 + c) * VROW + st * 32 + h * 16);
                oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pk.v, oacc[db], 0, 0, 0);
            }
        }
        ASTAMP(4, oacc[DB - 1][0]);
        ASTAMP_END(0x0234, 4);
    };
    f32x16 sA[2], sB[2];
    __builtin_amdgcn_s_barrier();            // tiles 0 and 1 are in LDS
    qk(sA, smem);
    int tile = 0, slot = 0;
    if (!p.causal) {
        const int nfull = hot_t >> 6;   // tiles without a ragged tail
        for (; tile + 2 <= nfull; tile += 2) {
            step(std::false_type{}, sA, sB, tile, slot);
            slot = (slot + 1) & 3;
            step(std::false_type{}, sB, sA, tile + 1, slot);
            slot = (slot + 1) & 3;
        }
    }
    for (; tile < ntiles; ++tile) {
        step(std::true_type{}, sA, sB, tile, slot);
        slot = (slot + 1) & 3;
        sA[0] = sB[0];
        sA[1] = sB[1];
    }

    } else {
    for (int tile = 0; tile < ntiles; ++tile) {
        const int t0 = tile * 64;
        char* sK = smem + (PIPE ? (tile & 1) : 0) * BUF_BYTES;
        if (PIPE) {
            __syncthreads();
            ASTAMP(0, mref);
            if (tile + 1 < ntiles) gload(t0 + 64);
        } else {
            __syncthreads();
            gload(t0);
            lstore(sK, t0);
            __syncthreads();
        }
        // ---- S^T = K Q^T [- m_ref]: two 32-key blocks
        f32x16 sacc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 kfrag = *reinterpret_cast<const bf16x8*>(sK + (kb * 32 + c) * KROW + ks * 32 + h * 16);
                if (ks == 0) {
                    if (PRESC) sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfrag, qf[0], negm, 0, 0, 0);
                    else {
                        f32x16 z;
#pragma unroll
                        for (int i = 0; i < 16; ++i) z[i] = 0.f;
                        sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfrag, qf[0], z, 0, 0, 0);
                    }
                } else {
                    sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfrag, qf[ks], sacc[kb], 0, 0, 0);
                }
            }
        ASTAMP(1, sacc[1][0]);
        // ---- masks (ragged key tail, causal): only the tiles that need them pay
        if (t0 + 64 > hot_t) {
            int key0 = t0 + 4 * h;
            asm volatile("" : "+v"(key0));
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (key0 + kb * 32 + (i & 3) + 8 * (i >> 2) >= hot_t) sacc[kb][i] = -1e30f;
        }
        if (p.causal) {
            int key0 = t0 + 4 * h;
            asm volatile("" : "+v"(key0));
            const int qi = q0 + c;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (key0 + kb * 32 + (i & 3) + 8 * (i >> 2) > qi) sacc[kb][i] = -1e30f;
        }
        // ---- lane-local maximum; the row maximum and the (rare) move of the reference only when some lane asks for it
        float m = fmaxf(sacc[0][0], sacc[0][1]);
#pragma unroll
        for (int i = 2; i < 16; i += 2) m = fmaxf(fmaxf(m, sacc[0][i]), sacc[0][i + 1]);
#pragma unroll
        for (int i = 0; i < 16; i += 2) m = fmaxf(fmaxf(m, sacc[1][i]), sacc[1][i + 1]);
        ASTAMP(2, m);
        const bool ask = tile == 0 || (PRESC ? (m > ATTN_THR) : ((m - mref) * p.sl2 > ATTN_THR));
        if (__builtin_amdgcn_ballot_w64(ask) != 0) {
            const uint32_t u = __float_as_uint(m);
            auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            float mx;
            asm("v_max_f32 %0, %1, %2" : "=v"(mx) : "v"(sw[0]), "v"(sw[1]));   // both half-waves: the query's maximum over the tile
            const bool need = tile == 0 || (PRESC ? (mx > ATTN_THR) : ((mx - mref) * p.sl2 > ATTN_THR));
            float alpha;
            if (PRESC) {
                const float delta = need ? mx : 0.f;
                alpha = __builtin_amdgcn_exp2f(-delta);
                mref += delta;
#pragma unroll
                for (int i = 0; i < 16; ++i) { negm[i] = -mref; sacc[0][i] -= delta; sacc[1][i] -= delta; }
            } else {
                const float mnew = need ? mx : mref;
                alpha = __builtin_amdgcn_exp2f((mref - mnew) * p.sl2);
                mref = mnew;
            }
            if (tile != 0) {
                lrun *= alpha;
#pragma unroll
                for (int db = 0; db < DB; ++db)
#pragma unroll
                    for (int i = 0; i < 16; ++i) oacc[db][i] *= alpha;
            }
        }
        // ---- exponentials in place
        {
            const float nm = -mref * p.sl2;
            float ls = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float pv = __builtin_amdgcn_exp2f(PRESC ? sacc[kb][i] : fmaf(sacc[kb][i], p.sl2, nm));
                    sacc[kb][i] = pv;
                    if (!ONES_ROW) ls += pv;
                }
            if (!ONES_ROW) lrun += ls;
        }
        ASTAMP(3, sacc[1][15]);
        // ---- O^T += V^T P^T, 16 keys per step
        const char* sV = sK + 64 * KROW;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int kb = st >> 1, i0 = (st & 1) * 8;
            union { bf16x8 v; uint32_t u[4]; } pk;
            pk.u[0] = pack_bf2(sacc[kb][i0 + 0], sacc[kb][i0 + 1]);
            pk.u[1] = pack_bf2(sacc[kb][i0 + 2], sacc[kb][i0 + 3]);
            pk.u[2] = pack_bf2(sacc[kb][i0 + 4], sacc[kb][i0 + 5]);
            pk.u[3] = pack_bf2(sacc[kb][i0 + 6], sacc[kb][i0 + 7]);
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sV + (db * 32 + c) * VROW + st * 32 + h * 16);
                oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pk.v, oacc[db], 0, 0, 0);
            }
        }
        ASTAMP(4, oacc[DB - 1][0]);
        if (PIPE && tile + 1 < ntiles) lstore(smem + ((tile & 1) ^ 1) * BUF_BYTES, t0 + 64);
        ASTAMP(5, mref);
        ASTAMP_END(0x012345, 6);
    }
    }
#ifdef MSD_STAMPS
    if (lane == 0 && (wave == 0 || (MSD_ASTAMP_WAVES && blockIdx.x < 256))) {
        // (MSD_ASTAMP_WAVES: rows 2048 + 8 block + wave hold every compute wave of the first 256 workgroups)
        unsigned long long* dst = g_astamps + (wave == 0 ? (size_t)(blockIdx.x & 2047) : (size_t)(2048 + blockIdx.x * 8 + wave)) * 16;
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[i] = st_acc[i];
        dst[8] = st_prev - st_first;
        dst[9] = wall_clock64() - st_wall0;
        dst[10] = st_wall0;
        dst[11] = (unsigned long long)ntiles;
    }
#endif

    // ---- normalise and store: lane (c, h) holds channels db*32 + 8 (i >> 2) + 4 h + (i & 3) of query q0 + c
    float lt;
    if (ONES_ROW) {   // row D of V^T is all ones: O^T[D][q] = sum_k P[k][q] (same bf16-rounded P as the numerator)
        constexpr int rr = D % 32, h0 = (rr >> 2) & 1, i0 = (rr & 3) + 4 * (rr >> 3);
        lt = __shfl(oacc[DB - 1][i0], c + 32 * h0);
    } else {
        lt = lrun + __shfl_xor(lrun, 32);
    }
    const float inv = 1.0f / lt;
    const int qrow = q0 + c;
    bf16_t* op = p.out + ((size_t)b * hot_s + min(qrow, hot_s - 1)) * p.o_ld + hd * D;
    if ((p.o_ld & 7) == 0) {
        // 16-byte stores: one half-wave swap per register turns (group 2j of this lane, group 2j of the other half) into 8
        // consecutive channels — lower half: channels 16 j + 0..7, upper half: 16 j + 8..15 of the 32-channel block
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(oacc[db][8 * j + e] * inv), __float_as_uint(oacc[db][8 * j + 4 + e] * inv), false, false);
                    v[e] = __uint_as_float(sw[0]);
                    v[4 + e] = __uint_as_float(sw[1]);
                }
                const int d = db * 32 + 16 * j + 8 * h;
                if (d < D && qrow < q_end) *reinterpret_cast<uint4*>(op + d) = pack8(v);
            }
    } else {
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = db * 32 + 8 * g4 + 4 * h;
                if (d < D && qrow < q_end) {
                    uint2 o;
                    o.x = pack_bf2(oacc[db][4 * g4 + 0] * inv, oacc[db][4 * g4 + 1] * inv);
                    o.y = pack_bf2(oacc[db][4 * g4 + 2] * inv, oacc[db][4 * g4 + 3] * inv);
                    *reinterpret_cast<uint2*>(op + d) = o;
                }
            }
    }
}

// ---- cross-attention with its query projection inside (text context: T <= 96 keys) --------------------------------
// CrossAttention over the text context (diffusion_model.py:102-127 with a context of 77 tokens): in the launch list the
// query projection (LayerNorm-fold Dense, C -> C) and the attention over 77 keys are two latency-bound launches of ~12 us
// each for 1.7 + 0.8 GFLOP.  Here a workgroup = (64 queries, one head): each wave loads its 16 rows of the RAW block
// input straight into MFMA operand registers, the head's D rows of the (gamma-folded, prescaled) to_q weights arrive by
// LDS-DMA, and q^T = W_h x^T comes out of the matrix core with the query on the MFMA column — i.e. already in the layout of
// the S^T = K q^T product's B operand: the accumulators of two 16-row blocks, LayerNorm-corrected and packed to bf16 (the
// same rounding the stored q had), ARE the operand for 32 head channels (k-slot order {4g+e, 16+4g+e}; the K fragment is
// read from LDS in the same order, two 8-byte pieces per lane).  All T keys fit one tile, so the softmax is a single pass
// over 5-6 key blocks in registers, and P packs into the B operand of O^T = V^T P^T the same way.  x is re-read by the 8
// heads of a query tile, which run next to each other on one XCD.
struct XArgs {
    const bf16_t* x; const float* ln_in; const bf16_t* wq; const float* colsum; const float* bias;
    const bf16_t* k; const bf16_t* vt; bf16_t* out;
    int batch, heads, s, t, c, k_ld, vt_ld, o_ld, ln_slots;
    float ln_inv_k, ln_eps;
    uint32_t w_rs, w_ks;            // to_q weight addressing in bytes: row stride, 64-channel chunk stride
    uint32_t mg_heads, mg_qtiles;
};

// NW waves = 16 NW queries per workgroup (4 or 8: the weight / K / V^T images are shared by twice the queries; the host picks 8
// when the 64-query grid would need more than one round of workgroups).
template <int D, int NW>
// (leading scalars: kernarg preload, 14 dwords — x, wq, k + batch, heads, s, mg_heads, mg_qtiles, w_rs, w_ks, t; k_ld / ln_slots follow unpreloaded)
#define XATTN_HOT_PARAMS const bf16_t* hot_x, const bf16_t* hot_wq, const bf16_t* hot_k, int hot_batch, int hot_heads, int hot_s, uint32_t hot_mg_heads, \
                         uint32_t hot_mg_qtiles, uint32_t hot_w_rs, uint32_t hot_w_ks, int hot_t, int hot_k_ld, int hot_ln_slots
#define XATTN_HOT_ARGS(a) (a).x, (a).wq, (a).k, (a).batch, (a).heads, (a).s, (a).mg_heads, (a).mg_qtiles, (a).w_rs, (a).w_ks, (a).t, (a).k_ld, (a).ln_slots
__global__ __launch_bounds__(64 * NW) void xattn_q_kernel(XATTN_HOT_PARAMS, const XArgs p) {
    constexpr int NT = 64 * NW, QT = 16 * NW;
    constexpr int C = 8 * D;                  // (8 heads: SD1.5)
    constexpr int KC = C / 64;                // 64-channel chunks of the projection's K
    constexpr int DB = (D + 15) / 16;         // 16-row blocks of q^T / O^T
    constexpr int DCK = (DB + 1) / 2;         // 32-channel chunks of the QK^T contraction
    constexpr int WR = (DB * 16 + 31) / 32 * 32;   // weight rows staged per chunk (DMA rounds of 32 rows)
    constexpr int W_BYTES = KC * WR * 128;
    constexpr int TB = 6, TPAD = TB * 16;     // key blocks (T <= 96), 32-key chunks of the PV contraction = 3
    constexpr int KROW = DCK * 64 + 16, VROW = TPAD * 2 + 16;
    constexpr int K_BYTES = TPAD * KROW, V_BYTES = DB * 16 * VROW;
    constexpr int DCH = D / 8;                // 16-byte chunks per K row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sW = smem;
    char* sK = smem + W_BYTES;
    char* sV = sK + K_BYTES;
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int qtiles = (hot_s + QT - 1) / QT;
    const int wi = xcd_remap(blockIdx.x, qtiles * hot_heads * hot_batch);
    const int qt_b = udiv_magic(wi, hot_heads, hot_mg_heads), h = wi - qt_b * hot_heads;     // head fastest: the 8 heads of a tile share x
    const int b = udiv_magic(qt_b, qtiles, hot_mg_qtiles);
    const int q0 = (qt_b - b * qtiles) * QT + wave * 16;

    // ---- 1. everything this workgroup reads, issued at once: weight DMAs first (oldest in the queue), then plain loads
    if (wave < 4) {   // (waves 0-3 issue the weight DMAs: 32 rows x 128 B per instruction round)
        const int cpos = tid & 7, lrow = tid >> 3;
        const uint32_t lds_wave = lds0 + (uint32_t)(wave * 8) * 128u;
#pragma unroll
        for (int rr = 0; rr < WR / 32; ++rr) {
            const int row = rr * 32 + lrow;
            const uint32_t off = (uint32_t)(h * D + min(row, D - 1)) * hot_w_rs + (uint32_t)((cpos ^ ((row >> 1) & 7)) * 16);
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)
                dma16s(hot_wq, off + (uint32_t)kc * hot_w_ks, __builtin_amdgcn_readfirstlane(lds_wave + (uint32_t)(kc * WR + rr * 32) * 128u));
        }
    }
    bf16x8 xf[KC * 2];
    const int qrow = min(q0 + r, hot_s - 1);
    {
        const bf16_t* src = hot_x + ((size_t)b * hot_s + qrow) * C + g * 8;
#pragma unroll
        for (int ks = 0; ks < KC * 2; ++ks) xf[ks] = *reinterpret_cast<const bf16x8*>(src + ks * 32);
    }
    constexpr int LNS = 5;   // (LN_MAX_SLOTS / 4 of conv_common.h)
    float2 lnp[LNS];
    {
        const float2* src = reinterpret_cast<const float2*>(p.ln_in) + ((size_t)b * hot_s + qrow) * hot_ln_slots;
#pragma unroll
        for (int k = 0; k < LNS; ++k) lnp[k] = src[min(g + 4 * k, hot_ln_slots - 1)];
    }
    float4 csv[DB], bsv[DB];   // column sums / folded bias of this lane's 4 head channels per block (clamped: masked below)
#pragma unroll
    for (int bb = 0; bb < DB; ++bb) {
        const int d = min(bb * 16 + 4 * g, D - 4);
        csv[bb] = *reinterpret_cast<const float4*>(p.colsum + h * D + d);
        bsv[bb] = p.bias ? *reinterpret_cast<const float4*>(p.bias + h * D + d) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    constexpr int KCH = (TPAD * DCH + NT - 1) / NT, VCH = (D * (TPAD / 8) + NT - 1) / NT;
    uint4 rk[KCH], rv[VCH];
    const bf16_t* kbase = hot_k + (size_t)b * hot_t * hot_k_ld + h * D;
    const bf16_t* vbase = p.vt + ((size_t)b * hot_heads + h) * D * p.vt_ld;
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
        const int idx = tid + NT * i, key = idx / DCH, ch = idx - key * DCH;
        rk[i] = make_uint4(0, 0, 0, 0);
        if (key < hot_t) rk[i] = *reinterpret_cast<const uint4*>(kbase + (size_t)key * hot_k_ld + ch * 8);
    }
#pragma unroll
    for (int i = 0; i < VCH; ++i) {
        const int idx = tid + NT * i, d = idx / (TPAD / 8), ch = idx - d * (TPAD / 8);
        rv[i] = make_uint4(0, 0, 0, 0);
        if (d < D && ch * 8 + 8 <= p.vt_ld && ch * 8 < hot_t) rv[i] = *reinterpret_cast<const uint4*>(vbase + (size_t)d * p.vt_ld + ch * 8);
    }
    // K / V^T images: zero everything (pad rows / columns are never written), then the loaded pieces
    for (int off = tid * 16; off < K_BYTES + V_BYTES; off += NT * 16) *reinterpret_cast<uint4*>(sK + off) = make_uint4(0, 0, 0, 0);
    __syncthreads();   // (also waits for every load above)
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
        const int idx = tid + NT * i, key = idx / DCH, ch = idx - key * DCH;
        if (key < hot_t) *reinterpret_cast<uint4*>(sK + key * KROW + ch * 16) = rk[i];
    }
#pragma unroll
    for (int i = 0; i < VCH; ++i) {
        const int idx = tid + NT * i, d = idx / (TPAD / 8), ch = idx - d * (TPAD / 8);
        if (d >= D || ch * 8 >= hot_t) continue;
        uint4 v = rv[i];
        const int valid = hot_t - ch * 8;   // keys >= t are padding of unspecified content: force to 0
        if (valid < 8) {
            uint32_t* u = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (2 * j >= valid) u[j] = 0;
                else if (2 * j + 1 >= valid) u[j] &= 0xFFFFu;
            }
        }
        *reinterpret_cast<uint4*>(sV + d * VROW + ch * 16) = v;
    }
    // LayerNorm row moments, summed as cg_epilogue does (lane group g: slots g, g + 4, ...; then (g0 + g1) + (g2 + g3))
    float mean, rstd;
    {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < LNS; ++k)
            if (g + 4 * k < hot_ln_slots) { s1 += lnp[k].x; s2 += lnp[k].y; }
        s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
        mean = s1 * p.ln_inv_k;
        rstd = rsqrtf(fmaxf(s2 * p.ln_inv_k - mean * mean, 0.f) + p.ln_eps);
    }
    wait_vmcnt<0>();
    __syncthreads();   // weights (DMA) and the K / V^T images are in LDS

    // ---- 2. q^T = W_h x^T: block bb holds head channels 16 bb + 4 g + e of query r
    f32x4 qacc[DB];
#pragma unroll
    for (int bb = 0; bb < DB; ++bb) qacc[bb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int bb = 0; bb < DB; ++bb) {
                const int row = bb * 16 + r;
                const bf16x8 wf = *reinterpret_cast<const bf16x8*>(sW + (kc * WR + row) * 128 + (((ks * 4 + g) ^ ((row >> 1) & 7)) << 4));
                qacc[bb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[kc * 2 + ks], qacc[bb], 0, 0, 0);
            }
    // LayerNorm fold + bias (the Dense epilogue's expressions), head channels >= D forced to zero; two blocks -> one operand
    uint32_t qpk[DB + (DB & 1)][2];
#pragma unroll
    for (int bb = 0; bb < DB + (DB & 1); ++bb) {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (bb < DB) {
            const bool ok = bb * 16 + 4 * g < D;   // (D % 4 == 0: a lane's 4 channels are all inside or all outside)
            v[0] = ok ? (rstd * (qacc[bb][0] - mean * csv[bb].x) + bsv[bb].x) : 0.f;
            v[1] = ok ? (rstd * (qacc[bb][1] - mean * csv[bb].y) + bsv[bb].y) : 0.f;
            v[2] = ok ? (rstd * (qacc[bb][2] - mean * csv[bb].z) + bsv[bb].z) : 0.f;
            v[3] = ok ? (rstd * (qacc[bb][3] - mean * csv[bb].w) + bsv[bb].w) : 0.f;
        }
        qpk[bb][0] = pack_bf2(v[0], v[1]);
        qpk[bb][1] = pack_bf2(v[2], v[3]);
    }
    // ---- 3. S^T = K q^T (q carries scale * log2 e): lane holds keys 16 kb + 4 g + e of query r
    f32x4 sacc[TB];
#pragma unroll
    for (int kb = 0; kb < TB; ++kb) {
        sacc[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < DCK; ++c) {
            union { bf16x8 v; uint32_t u[4]; } qo;
            qo.u[0] = qpk[2 * c][0]; qo.u[1] = qpk[2 * c][1]; qo.u[2] = qpk[2 * c + 1][0]; qo.u[3] = qpk[2 * c + 1][1];
            union { bf16x8 v; uint2 h2[2]; } kf;
            const char* kp = sK + (kb * 16 + r) * KROW + c * 64 + g * 8;
            kf.h2[0] = *reinterpret_cast<const uint2*>(kp);
            kf.h2[1] = *reinterpret_cast<const uint2*>(kp + 32);
            sacc[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf.v, qo.v, sacc[kb], 0, 0, 0);
        }
    }
    // single-pass softmax over the T keys of the lane's query
    float m = -1e30f;
#pragma unroll
    for (int kb = 0; kb < TB; ++kb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (kb * 16 + 4 * g + e >= hot_t) sacc[kb][e] = -1e30f;
            m = fmaxf(m, sacc[kb][e]);
        }
    m = rows_max4(m);
    float lsum = 0.f;
#pragma unroll
    for (int kb = 0; kb < TB; ++kb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float pv = __builtin_amdgcn_exp2f(sacc[kb][e] - m);
            sacc[kb][e] = pv;
            lsum += pv;
        }
    lsum += __shfl_xor(lsum, 16);
    lsum += __shfl_xor(lsum, 32);
    // ---- 4. O^T = V^T P^T over 32-key chunks (P packed to bf16 = the B operand, key order {4g+e, 16+4g+e} of the chunk)
    f32x4 oacc[DB];
#pragma unroll
    for (int bb = 0; bb < DB; ++bb) oacc[bb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < TB / 2; ++c) {
        union { bf16x8 v; uint32_t u[4]; } pk;
        pk.u[0] = pack_bf2(sacc[2 * c][0], sacc[2 * c][1]);
        pk.u[1] = pack_bf2(sacc[2 * c][2], sacc[2 * c][3]);
        pk.u[2] = pack_bf2(sacc[2 * c + 1][0], sacc[2 * c + 1][1]);
        pk.u[3] = pack_bf2(sacc[2 * c + 1][2], sacc[2 * c + 1][3]);
#pragma unroll
        for (int bb = 0; bb < DB; ++bb) {
            union { bf16x8 v; uint2 h2[2]; } vf;
            const char* vp = sV + (bb * 16 + r) * VROW + c * 64 + g * 8;
            vf.h2[0] = *reinterpret_cast<const uint2*>(vp);
            vf.h2[1] = *reinterpret_cast<const uint2*>(vp + 32);
            oacc[bb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pk.v, oacc[bb], 0, 0, 0);
        }
    }
    if (q0 + r < hot_s) {
        const float inv = 1.0f / lsum;
        bf16_t* op = p.out + ((size_t)b * hot_s + q0 + r) * p.o_ld + h * D;
#pragma unroll
        for (int bb = 0; bb < DB; ++bb) {
            const int d = bb * 16 + 4 * g;
            if (d < D) {
                uint2 o;
                o.x = pack_bf2(oacc[bb][0] * inv, oacc[bb][1] * inv);
                o.y = pack_bf2(oacc[bb][2] * inv, oacc[bb][3] * inv);
                *reinterpret_cast<uint2*>(op + d) = o;
            }
        }
    }
}

template <int D>
static constexpr int xattn_lds() {
    constexpr int DB = (D + 15) / 16, DCK = (DB + 1) / 2, WR = (DB * 16 + 31) / 32 * 32;
    return (8 * D / 64) * WR * 128 + 96 * (DCK * 64 + 16) + DB * 16 * (96 * 2 + 16);
}
static_assert(xattn_lds<40>() <= 160 * 1024 && xattn_lds<80>() <= 160 * 1024, "LDS budget");

// ---- the same at C = 1280 (d = 160: the 16x16 and 8x8 levels) -------------------------------------------------------
// A head's slice of to_q is 160 rows x 1280 channels = 400 KB: it cannot wait in LDS like the 25 / 100 KB slices above, and
// a wave's 16 raw rows are 160 registers.  So the projection is a K loop over the twenty 64-channel chunks: the chunk's
// 160 x 64 weight block (20 KB) arrives by LDS-DMA in a four-stage ring, three chunks in flight; the wave's two x fragments
// of a chunk travel global -> VGPR beside it (inline asm: both queues are counted by hand, see conv_wreg.hip) — ten
// accumulators per wave, 20 MFMAs per chunk.  What follows the loop is the kernel above with D = 160: q^T packed to bf16 IS
// the B operand of S^T = K q^T (five 32-channel steps per key block), single-pass softmax over T <= 96 keys, O^T = V^T P^T.
// The K / V^T images need no zero fill here (160 = 10 x 16 = 5 x 32: no pad rows or columns are ever read; key rows / key
// chunks >= t are stored as zeros by their owners).  A workgroup is bound by the 400 KB + 16 NW x 2.5 KB it streams through
// one CU (~6-8 us at the ~70 GB/s a CU takes in); head h runs on XCD h (workgroup id & 7), so an XCD's L2 holds one head's
// weights and the matrix leaves HBM once per launch.
typedef uint32_t xq_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void xq_load(xq_u32x4& lo, xq_u32x4& hi, const void* ptr) {
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:64" : "=&v"(lo), "=&v"(hi) : "v"(ptr) : "memory");
}
__device__ __forceinline__ void xr_load(xq_u32x4& dst, const void* ptr) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst) : "v"(ptr) : "memory");
}

template <int NW, int MODE = 0>   // MODE bits (experiments): 1 = no MFMAs in the projection loop, 2 = every chunk's loads aimed at chunk 0, 4 = no ring loads after chunk 0, 8 = no attention after the projection
__global__ __launch_bounds__(64 * NW) void xattn_q160_kernel(XATTN_HOT_PARAMS, const XArgs p) {
    constexpr int D = 160, C = 8 * D, KC = C / 64, DB = D / 16, DCK = DB / 2;
    constexpr int NT = 64 * NW, QT = 16 * NW;
    constexpr int NS = 4, PF = 3;                 // ring stages; chunks in flight
    constexpr int DMAW = (D / 8) / NW;            // weight DMAs per wave and chunk (8 rows x 128 B each)
    constexpr int OPS = DMAW + 2;                 // vector-memory operations per wave and chunk (+ the two x fragments)
    constexpr int W_STAGE = D * 128;
    constexpr int TB = 6, TPAD = TB * 16;
    constexpr int KROW = DCK * 64 + 16, VROW = TPAD * 2 + 16;
    constexpr int K_BYTES = TPAD * KROW;
    constexpr int DCH = D / 8;
    static_assert((D / 8) % NW == 0 && PF == NS - 1 && (PF - 1) * OPS <= 63 && KC % NS == 0, "ring configuration");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sW = smem;
    char* sK = smem + NS * W_STAGE;
    char* sV = sK + K_BYTES;
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int qtiles = (hot_s + QT - 1) / QT;
    const int h = blockIdx.x & 7, qt_b = blockIdx.x >> 3;       // head = XCD under round-robin placement (speed only)
    const int b = udiv_magic(qt_b, qtiles, hot_mg_qtiles);
    const int q0 = (qt_b - b * qtiles) * QT + wave * 16;
    const int qrow = min(q0 + r, hot_s - 1);

    // ---- 1. the small operands (oldest in the queue), then the ring.  K / V^T pass through registers into their LDS images
    // before the loop; their loads are inline asm like the x fragments (hipcc would wait for them one by one with counts
    // that know nothing of the DMAs behind them, i.e. for most of the ring's first three chunks).
    constexpr int KCH = (TPAD * DCH + NT - 1) / NT, VCH = (D * (TPAD / 8) + NT - 1) / NT;
    xq_u32x4 rk[KCH], rv[VCH];
    constexpr int LNS = 5;   // (LN_MAX_SLOTS / 4 of conv_common.h)
    float2 lnp[LNS];
    {
        const float2* src = reinterpret_cast<const float2*>(p.ln_in) + ((size_t)b * hot_s + qrow) * hot_ln_slots;
#pragma unroll
        for (int k = 0; k < LNS; ++k) lnp[k] = src[min(g + 4 * k, hot_ln_slots - 1)];
    }
    float4 csv[DB], bsv[DB];   // column sums / folded bias of this lane's 4 head channels per block
#pragma unroll
    for (int bb = 0; bb < DB; ++bb) {
        const int d = bb * 16 + 4 * g;
        csv[bb] = *reinterpret_cast<const float4*>(p.colsum + h * D + d);
        bsv[bb] = p.bias ? *reinterpret_cast<const float4*>(p.bias + h * D + d) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const bf16_t* kbase = hot_k + (size_t)b * hot_t * hot_k_ld + h * D;
    const bf16_t* vbase = p.vt + ((size_t)b * hot_heads + h) * D * p.vt_ld;
    // (clamped addresses, no branches: what lies beyond key t - 1 is replaced by zeros when the images are stored)
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
        const int idx = tid + NT * i, key = idx / DCH, ch = idx - key * DCH;
        xr_load(rk[i], kbase + (size_t)min(key, hot_t - 1) * hot_k_ld + ch * 8);
    }
    const int vch_last = (hot_t - 1) >> 3;   // (vt_ld >= t rounded up to 8: this chunk lies inside every row)
#pragma unroll
    for (int i = 0; i < VCH; ++i) {
        const int idx = tid + NT * i, d = idx / (TPAD / 8), ch = idx - d * (TPAD / 8);
        xr_load(rv[i], vbase + (size_t)min(d, D - 1) * p.vt_ld + min(ch, vch_last) * 8);
    }
    // ring: wave w moves weight rows [8 DMAW w, 8 DMAW (w + 1)) of every chunk and loads its own 16 x 64 block of x
    uint32_t woff[DMAW];
    {
        const int cpos = lane & 7, lrow = lane >> 3;
#pragma unroll
        for (int i = 0; i < DMAW; ++i) {
            const int row = (wave * DMAW + i) * 8 + lrow;
            woff[i] = (uint32_t)(h * D + row) * hot_w_rs + (uint32_t)((cpos ^ ((row >> 1) & 7)) * 16);
        }
    }
    const uint32_t lds_wave = lds0 + (uint32_t)(wave * DMAW) * 1024u;
    const char* xptr = reinterpret_cast<const char*>(hot_x + ((size_t)b * hot_s + qrow) * C + g * 8);
    xq_u32x4 xq[NS][2];
    auto issue = [&](int kc, int slot) {
        const uint32_t ko = (uint32_t)kc * hot_w_ks;
#pragma unroll
        for (int i = 0; i < DMAW; ++i) {
            if ((MODE & 4) && kc > 0) continue;   // (the waits then count operations that were never issued: they pass at once)
            dma16s(hot_wq, (MODE & 2) ? woff[i] : woff[i] + ko, __builtin_amdgcn_readfirstlane(lds_wave + (uint32_t)(slot * W_STAGE + i * 1024)));
        }
        if (!((MODE & 4) && kc > 0)) xq_load(xq[slot][0], xq[slot][1], xptr + ((MODE & 2) ? 0 : kc * 128));
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) issue(s, s);
    // the K / V^T images (every 16-byte piece of them is written: no zero fill)
    wait_vmcnt<PF * OPS>();
#pragma unroll
    for (int i = 0; i < KCH; ++i) asm volatile("" : "+v"(rk[i]));   // consumers stay below the wait
#pragma unroll
    for (int i = 0; i < VCH; ++i) asm volatile("" : "+v"(rv[i]));
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
        const int idx = tid + NT * i, key = idx / DCH, ch = idx - key * DCH;
        if (key < TPAD) *reinterpret_cast<xq_u32x4*>(sK + key * KROW + ch * 16) = key < hot_t ? rk[i] : (xq_u32x4){0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < VCH; ++i) {
        const int idx = tid + NT * i, d = idx / (TPAD / 8), ch = idx - d * (TPAD / 8);
        if (d >= D) continue;
        xq_u32x4 v = rv[i];
        const int valid = hot_t - ch * 8;   // keys >= t are padding of unspecified content: force to 0
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (2 * j >= valid) v[j] = 0;
            else if (2 * j + 1 >= valid) v[j] &= 0xFFFFu;
        }
        *reinterpret_cast<xq_u32x4*>(sV + d * VROW + ch * 16) = v;
    }

    // ---- 2. q^T = W_h x^T over the chunks: block bb holds head channels 16 bb + 4 g + e of query r
    f32x4 qacc[DB];
#pragma unroll
    for (int bb = 0; bb < DB; ++bb) qacc[bb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int sw = (r >> 1) & 7;
#pragma unroll 1
    for (int kc0 = 0; kc0 < KC; kc0 += NS) {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int kc = kc0 + u;
            wait_vmcnt_tiles<OPS, PF - 1>(KC - 1 - kc);   // chunk kc has landed (this wave's share of it)
            __syncthreads();                               // ... every wave's; and stage (kc - 1) & 3 has been read by all
            if (kc + PF < KC) issue(kc + PF, (u + PF) % NS);
            asm volatile("" : "+v"(xq[u][0]));             // consumers stay below the wait
            asm volatile("" : "+v"(xq[u][1]));
            const char* ws = sW + u * W_STAGE + r * 128;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 xf = __builtin_bit_cast(bf16x8, xq[u][ks]);
#pragma unroll
                for (int bb = 0; bb < DB; ++bb) {
                    const bf16x8 wf = *reinterpret_cast<const bf16x8*>(ws + bb * 2048 + (((ks * 4 + g) ^ sw) << 4));
                    if ((MODE & 1) && kc > 0) continue;
                    qacc[bb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, qacc[bb], 0, 0, 0);
                }
            }
        }
    }
    // LayerNorm row moments, summed as cg_epilogue does (lane group g: slots g, g + 4, ...; then (g0 + g1) + (g2 + g3))
    float mean, rstd;
    {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < LNS; ++k)
            if (g + 4 * k < hot_ln_slots) { s1 += lnp[k].x; s2 += lnp[k].y; }
        s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
        mean = s1 * p.ln_inv_k;
        rstd = rsqrtf(fmaxf(s2 * p.ln_inv_k - mean * mean, 0.f) + p.ln_eps);
    }
    // LayerNorm fold + bias (the Dense epilogue's expressions); two blocks -> one operand
    uint32_t qpk[DB][2];
#pragma unroll
    for (int bb = 0; bb < DB; ++bb) {
        qpk[bb][0] = pack_bf2(rstd * (qacc[bb][0] - mean * csv[bb].x) + bsv[bb].x, rstd * (qacc[bb][1] - mean * csv[bb].y) + bsv[bb].y);
        qpk[bb][1] = pack_bf2(rstd * (qacc[bb][2] - mean * csv[bb].z) + bsv[bb].z, rstd * (qacc[bb][3] - mean * csv[bb].w) + bsv[bb].w);
    }
    __syncthreads();   // the K / V^T images of the other waves (the loop's barriers have long ordered them; kept for the reader)
    if (MODE & 8) {   // (no attention: the packed q of the head as the result, so that nothing above is dead code)
        if (q0 + r < hot_s) {
            bf16_t* op = p.out + ((size_t)b * hot_s + q0 + r) * p.o_ld + h * D;
#pragma unroll
            for (int bb = 0; bb < DB; ++bb) *reinterpret_cast<uint2*>(op + bb * 16 + 4 * g) = make_uint2(qpk[bb][0], qpk[bb][1]);
        }
        return;
    }
    // ---- 3. S^T = K q^T (q carries scale * log2 e): lane holds keys 16 kb + 4 g + e of query r
    f32x4 sacc[TB];
#pragma unroll
    for (int kb = 0; kb < TB; ++kb) {
        sacc[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < DCK; ++c) {
            union { bf16x8 v; uint32_t u[4]; } qo;
            qo.u[0] = qpk[2 * c][0]; qo.u[1] = qpk[2 * c][1]; qo.u[2] = qpk[2 * c + 1][0]; qo.u[3] = qpk[2 * c + 1][1];
            union { bf16x8 v; uint2 h2[2]; } kf;
            const char* kp = sK + (kb * 16 + r) * KROW + c * 64 + g * 8;
            kf.h2[0] = *reinterpret_cast<const uint2*>(kp);
            kf.h2[1] = *reinterpret_cast<const uint2*>(kp + 32);
            sacc[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf.v, qo.v, sacc[kb], 0, 0, 0);
        }
    }
    // single-pass softmax over the T keys of the lane's query
    float m = -1e30f;
#pragma unroll
    for (int kb = 0; kb < TB; ++kb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (kb * 16 + 4 * g + e >= hot_t) sacc[kb][e] = -1e30f;
            m = fmaxf(m, sacc[kb][e]);
        }
    m = rows_max4(m);
    float lsum = 0.f;
#pragma unroll
    for (int kb = 0; kb < TB; ++kb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float pv = __builtin_amdgcn_exp2f(sacc[kb][e] - m);
            sacc[kb][e] = pv;
            lsum += pv;
        }
    lsum += __shfl_xor(lsum, 16);
    lsum += __shfl_xor(lsum, 32);
    // ---- 4. O^T = V^T P^T over 32-key chunks (P packed to bf16 = the B operand, key order {4g+e, 16+4g+e} of the chunk)
    f32x4 oacc[DB];
#pragma unroll
    for (int bb = 0; bb < DB; ++bb) oacc[bb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < TB / 2; ++c) {
        union { bf16x8 v; uint32_t u[4]; } pk;
        pk.u[0] = pack_bf2(sacc[2 * c][0], sacc[2 * c][1]);
        pk.u[1] = pack_bf2(sacc[2 * c][2], sacc[2 * c][3]);
        pk.u[2] = pack_bf2(sacc[2 * c + 1][0], sacc[2 * c + 1][1]);
        pk.u[3] = pack_bf2(sacc[2 * c + 1][2], sacc[2 * c + 1][3]);
#pragma unroll
        for (int bb = 0; bb < DB; ++bb) {
            union { bf16x8 v; uint2 h2[2]; } vf;
            const char* vp = sV + (bb * 16 + r) * VROW + c * 64 + g * 8;
            vf.h2[0] = *reinterpret_cast<const uint2*>(vp);
            vf.h2[1] = *reinterpret_cast<const uint2*>(vp + 32);
            oacc[bb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pk.v, oacc[bb], 0, 0, 0);
        }
    }
    if (q0 + r < hot_s) {
        const float inv = 1.0f / lsum;
        bf16_t* op = p.out + ((size_t)b * hot_s + q0 + r) * p.o_ld + h * D;
#pragma unroll
        for (int bb = 0; bb < DB; ++bb) {
            uint2 o;
            o.x = pack_bf2(oacc[bb][0] * inv, oacc[bb][1] * inv);
            o.y = pack_bf2(oacc[bb][2] * inv, oacc[bb][3] * inv);
            *reinterpret_cast<uint2*>(op + bb * 16 + 4 * g) = o;
        }
    }
}
template <int NW>
static constexpr int xattn160_lds() { return 4 * 160 * 128 + 96 * (5 * 64 + 16) + 160 * (96 * 2 + 16); }
static_assert(xattn160_lds<4>() <= 160 * 1024, "LDS budget");

// ---- d = 512: the VAE's single-head AttentionBlock (layers.py:28-59) ----------------------------------------------
// The head does not fit the kernel above (O^T of 32 queries x 512 channels alone is 256 registers per lane), and the
// reference's route — materialise softmax(q k^T / sqrt(C)) — is 64 MB of fp32 scores + 32 MB of probabilities per image
// at 512x512 (340 MB at 768x768) through HBM.  Here: one workgroup = 4 waves x 16 queries; a wave keeps its 16 x 512
// query fragments (64 registers) and its O^T accumulator (32 MFMA row blocks = 128 registers) for the whole kernel and
// walks the keys in 32-key tiles.  A tile is 32 KB of K rows + 32 KB of V^T rows, moved L2 -> LDS by LDS-DMA (no
// registers to stage through) into a two-stage ring, the next tile in flight while this one is multiplied:
//   * K image [32 keys][1 KB]: one DMA instruction = one key row; the 16-byte chunk index is XOR-swizzled with the key's
//     low 4 bits on the SOURCE address (LDS-DMA writes lane-linear), undone in the ds_read_b128 fragment reads, which are
//     then conflict-free (rows are a multiple of the 256-byte bank row apart);
//   * V^T image [512 channels][64 B]: one DMA instruction = 16 channel rows; chunk index XOR ((channel >> 2) & 3), so the
//     8-byte fragment reads of 16 consecutive channels at one key offset spread over all 64 banks.
// Same transposed products, in-register softmax and lazy reference maximum as above; the row sum is a VALU sum (no
// spare V^T row when d is a multiple of 16).  Per tile and wave: 64 MFMAs (1024 cycles) against 64 KB of LDS-DMA per
// workgroup, so at one workgroup per CU the tile loop runs at the L2 -> LDS rate (~1 us per tile), about the time of the
// three launches it replaces at batch 1, and it scales with the batch where those ran sample by sample.
__global__ __launch_bounds__(256) void attention512_kernel(const AArgs p) {
    constexpr int D = 512, KT = 32, KS = D / 32, DF = D / 16;
    constexpr int K_BYTES = KT * D * 2, V_BYTES = D * KT * 2, ST_BYTES = K_BYTES + V_BYTES;   // 32 KB + 32 KB per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int qtiles = (p.s + 63) / 64;
    // key split (round 5): the VAE's one head at 512x512 is 64 query tiles - a quarter of the chip, each walking all 4096 keys.  With
    // nsplit = 4 a workgroup walks a quarter of the keys and leaves its unnormalised O, reference maximum and row sum in `ws`;
    // attn512_merge_kernel adds the parts in part order.  The split follows the KEY COUNT only (host), never the batch.
    const int wi_all = xcd_remap(blockIdx.x, (qtiles * p.heads * p.batch) << p.nsplit_shift);
    const int wi = wi_all >> p.nsplit_shift, part = wi_all - (wi << p.nsplit_shift);
    const int bh = udiv_magic(wi, qtiles, p.mg_qtiles);
    const int b = udiv_magic(bh, p.heads, p.mg_heads), h = bh - b * p.heads;
    const int q0 = (wi - bh * qtiles) * 64 + wave * 16;
    const int ntiles = (p.t / KT) >> p.nsplit_shift;   // (host: t % (32 << nsplit_shift) == 0)
    const int key0 = part * ntiles * KT;
    const bf16_t* kbase = p.k + ((size_t)b * p.t + key0) * p.k_ld + h * D;
    const bf16_t* vbase = p.vt + ((size_t)b * p.heads + h) * D * p.vt_ld + key0;

    // this wave's share of a tile: key rows 8 wave .. 8 wave + 7 and channel rows 128 wave .. 128 wave + 127
    auto issue_tile = [&](int tile, int stage) {
        const uint32_t sb = lds0 + (uint32_t)stage * ST_BYTES;
        const int t0 = tile * KT;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = wave * 8 + i;
            const int chunk = (lane & 48) | ((lane & 15) ^ (row & 15));
            dma16(kbase + (size_t)(t0 + row) * p.k_ld + chunk * 8, sb + (uint32_t)row * 1024u);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = (wave * 8 + j) * 16 + (lane >> 2);
            const int chunk = (lane & 3) ^ ((d >> 2) & 3);
            dma16(vbase + (size_t)d * p.vt_ld + t0 + chunk * 8, sb + K_BYTES + (uint32_t)(wave * 8 + j) * 1024u);
        }
    };
    issue_tile(0, 0);

    bf16x8 qf[KS];
    {
        int qrow = q0 + r;
        if (qrow > p.s - 1) qrow = p.s - 1;
        const bf16_t* qp = p.q + ((size_t)b * p.s + qrow) * p.q_ld + h * D;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 32 + 8 * g);
    }
    f32x4 oacc[DF];
#pragma unroll
    for (int df = 0; df < DF; ++df) oacc[df] = (f32x4){0, 0, 0, 0};
    float mref = 0.f, lrun = 0.f;
    wait_vmcnt<0>();   // tile 0 and the query fragments have landed
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));

    for (int tile = 0; tile < ntiles; ++tile) {
        const int stage = tile & 1;
        __builtin_amdgcn_s_barrier();   // every wave's share of tile `tile` is in LDS; the other stage is no longer read
        if (tile + 1 < ntiles) issue_tile(tile + 1, stage ^ 1);
        const char* sK = smem + stage * ST_BYTES;
        const char* sV = sK + K_BYTES;
        // ---- S^T = K Q^T: 2 key blocks x 16 k-steps
        f32x4 sacc[2];
#pragma unroll
        for (int kf = 0; kf < 2; ++kf) sacc[kf] = (f32x4){0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int chunk = ks * 4 + g;
            const int pos = (chunk & 48) | ((chunk & 15) ^ r);   // (key row & 15 = r for both key blocks)
#pragma unroll
            for (int kf = 0; kf < 2; ++kf) {
                const bf16x8 kfrag = *reinterpret_cast<const bf16x8*>(sK + (kf * 16 + r) * 1024 + pos * 16);
                sacc[kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfrag, qf[ks], sacc[kf], 0, 0, 0);
            }
        }
        // ---- softmax with the lazy reference maximum (lane: keys tile*32 + kf*16 + 4g + e of query r)
        float m = fmaxf(fmaxf(sacc[0][0], sacc[0][1]), fmaxf(sacc[0][2], sacc[0][3]));
        m = fmaxf(m, fmaxf(fmaxf(sacc[1][0], sacc[1][1]), fmaxf(sacc[1][2], sacc[1][3])));
        const float mx = rows_max4(m);
        const bool need = tile == 0 || (mx - mref) * p.sl2 > ATTN_THR;
        if (__builtin_amdgcn_ballot_w64(need) != 0) {
            const float mnew = need ? mx : mref;
            const float alpha = __builtin_amdgcn_exp2f((mref - mnew) * p.sl2);
            mref = mnew;
            if (tile != 0) {
                lrun *= alpha;
#pragma unroll
                for (int df = 0; df < DF; ++df) { oacc[df][0] *= alpha; oacc[df][1] *= alpha; oacc[df][2] *= alpha; oacc[df][3] *= alpha; }
            }
        }
        const float nm = -mref * p.sl2;
        float ls = 0.f;
#pragma unroll
        for (int kf = 0; kf < 2; ++kf)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[kf][e], p.sl2, nm));
                sacc[kf][e] = pv;
                ls += pv;
            }
        lrun += ls;
        union { bf16x8 v; uint32_t u[4]; } pk;
        pk.u[0] = pack_bf2(sacc[0][0], sacc[0][1]);
        pk.u[1] = pack_bf2(sacc[0][2], sacc[0][3]);
        pk.u[2] = pack_bf2(sacc[1][0], sacc[1][1]);
        pk.u[3] = pack_bf2(sacc[1][2], sacc[1][3]);
        // ---- O^T += V^T P^T: 32 channel blocks, one 32-key k-step each
#pragma unroll
        for (int df = 0; df < DF; ++df) {
            const int d = df * 16 + r;
            const int sw = (d >> 2) & 3;
            const char* vrow = sV + d * 64 + (g & 1) * 8;
            union { bf16x8 v; uint2 h2[2]; } vf;
            vf.h2[0] = *reinterpret_cast<const uint2*>(vrow + (((g >> 1)) ^ sw) * 16);        // keys 4g .. 4g+3
            vf.h2[1] = *reinterpret_cast<const uint2*>(vrow + ((2 + (g >> 1)) ^ sw) * 16);    // keys 16+4g .. 16+4g+3
            oacc[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pk.v, oacc[df], 0, 0, 0);
        }
        wait_vmcnt<0>();   // this wave's share of tile + 1 has landed (before the barrier that publishes it)
    }

    float lt = lrun;
    lt += __shfl_xor(lt, 16);
    lt += __shfl_xor(lt, 32);
    const int qrow = q0 + r;
    if (p.nsplit_shift) {   // this part's unnormalised O (fp32), its reference maximum and its row sum
        if (qrow < p.s) {
            const size_t rows = (size_t)p.batch * p.heads * p.s, row = ((size_t)b * p.heads + h) * p.s + qrow;
            float* op = p.ws + ((size_t)part * rows + row) * D;
#pragma unroll
            for (int df = 0; df < DF; ++df) *reinterpret_cast<f32x4*>(op + df * 16 + 4 * g) = oacc[df];
            if (g == 0) {
                float* st = p.ws + (rows << p.nsplit_shift) * D + ((size_t)part * rows + row) * 2;
                st[0] = mref; st[1] = lt;
            }
        }
        return;
    }
    const float inv = 1.0f / lt;
    if (qrow < p.s) {
        bf16_t* op = p.out + ((size_t)b * p.s + qrow) * p.o_ld + h * D;
#pragma unroll
        for (int df = 0; df < DF; ++df) {
            uint2 o;
            o.x = pack_bf2(oacc[df][0] * inv, oacc[df][1] * inv);
            o.y = pack_bf2(oacc[df][2] * inv, oacc[df][3] * inv);
            *reinterpret_cast<uint2*>(op + df * 16 + 4 * g) = o;
        }
    }
}
constexpr int ATTN512_LDS = 2 * (32 * 512 * 2 + 512 * 32 * 2);

// out = (sum over the parts, in part order, of 2^((m_part - m) sl2) O_part) / (the same sum of l_part), m = max over the parts:
// one thread per (row, 4 channels), 128 threads per row
__global__ __launch_bounds__(256) void attn512_merge_kernel(const float* ws, bf16_t* out, int rows_total, int s, int heads, int o_ld, int nsplit, float sl2) {
    constexpr int D = 512;
    const int row = blockIdx.x * 2 + (threadIdx.x >> 7), c = (threadIdx.x & 127) * 4;
    if (row >= rows_total) return;
    const float* st = ws + (size_t)nsplit * rows_total * D;
    float m = st[(size_t)row * 2];
    for (int pt = 1; pt < nsplit; ++pt) m = fmaxf(m, st[((size_t)pt * rows_total + row) * 2]);
    float l = 0.f;
    f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int pt = 0; pt < nsplit; ++pt) {
        const float* sp = st + ((size_t)pt * rows_total + row) * 2;
        const float w = __builtin_amdgcn_exp2f((sp[0] - m) * sl2);
        l += w * sp[1];
        const f32x4 v = *reinterpret_cast<const f32x4*>(ws + ((size_t)pt * rows_total + row) * D + c);
        o[0] += w * v[0]; o[1] += w * v[1]; o[2] += w * v[2]; o[3] += w * v[3];
    }
    const float inv = 1.0f / l;
    const int bh = row / s, q = row - bh * s, b = bh / heads, h = bh - b * heads;
    uint2 r;
    r.x = pack_bf2(o[0] * inv, o[1] * inv);
    r.y = pack_bf2(o[2] * inv, o[3] * inv);
    *reinterpret_cast<uint2*>(out + ((size_t)b * s + q) * o_ld + h * D + c) = r;
}

// ---- launch configuration ---------------------------------------------------------------------
template <int D>
static constexpr int attn_nbuf() { return D <= 80 ? 2 : 1; }   // D = 160 has no registers left for the prefetch
template <int D, int NBUF>
static constexpr int attn_lds_bytes() {
    return NBUF * (64 * (((D + 31) / 32) * 32 * 2 + 16) + ((D + 15) / 16) * 16 * (64 * 2 + 16));
}

// Which head sizes run on the 32x32x16 form.  In place in the UNet step (bench.py's per-call pass, batch 1): d = 40, S = 4096
// 102.7 -> 73 us; d = 80, S = 1024 35.8 -> 27 us; d = 160, S = 256 15.6 -> 17.5 us (and its 77-key cross-attention 13.3 -> 15.2):
// at d = 160 the form needs 260 registers (one wave per SIMD) for grids that are latency-bound anyway, so it stays on 16x16.
// The choice follows the head size only — the two forms round differently, and a sample's bits must not depend on its batch.
template <int D>
static constexpr bool attn_form32() { return D == 40 || D == 80; }
template <int D, int NBUF>
static constexpr int attn32_lds_bytes() {
    return NBUF * (64 * (((D + 15) / 16) * 16 * 2 + 16) + ((D + 31) / 32) * 32 * (64 * 2 + 16));
}

static bool g_attn_attr_done = false;
static int g_attn_cus = 256;   // compute units of the device (msd_attention_init)
static int g_attn_qf = 0;     // 0 = automatic, 1 / 2 = 64 / 128 queries per workgroup forced (A/B runs)
static int g_attn_qf4_min = 512;   // d = 40, software-pipelined form: 256-query workgroups from this many 128-query workgroups on (round 6: 512, was 256:
                                   // the ONE-copy self-attention of the shared classifier-free-guidance prefix - 8 batch-heads at S = 4096 - filled half
                                   // the chip with 256-query workgroups; 128-query ones fill it: batch-1 loop -0.3 %; scheduling only, same bits)
void msd_set_attn_qf4_min(int v) { g_attn_qf4_min = v; }
void msd_set_attn_qf(int v) { g_attn_qf = v; }
static int g_attn_form = 2;   // head sizes 40 and 80: 2 = 32x32x16 MFMA form, software-pipelined for long key walks [default], 1 = 32x32x16 plain loop, 0 = 16x16x32 form like the other head sizes (A/B runs)
void msd_set_attn_form(int v) { g_attn_form = v; }
static int g_attn_d160_pipe = 1;   // d = 160, 64-query workgroups: 1 = K/V tile t+1 prefetched to registers under the products of tile t (NBUF 2; 204 registers)
                                   // [default], 0 = load -> store -> compute (NBUF 1: the only form the 128-query workgroups have registers for).  Staging only:
                                   // the same bits.  The S <= 256 launches of the 16x16 / 8x8 levels put one wave on a SIMD of a quarter of the CUs: a serial chain
void msd_set_attn_d160_pipe(int v) { g_attn_d160_pipe = v; }
#ifdef MSD_STAMPS
static int g_xattn160_mode = 0;   // (experiments: see xattn_q160_kernel's MODE; the instrumented build only)
int msd_set_xattn160_mode(int v) { g_xattn160_mode = v; return 1; }
#else
int msd_set_xattn160_mode(int v) { return v == 0; }   // the release library has the product instantiation only: any other mode is refused
#endif
static int g_xattn_nw = 0;    // 0 = automatic, 4 / 8 = waves (x 16 queries) per workgroup of the fused cross-attention (A/B runs)
void msd_set_xattn_nw(int v) { g_xattn_nw = v; }

template <int D, int NBUF, int QF, bool PRESC>
static hipError_t attn_attr1() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_kernel<D, NBUF, QF, PRESC>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, attn_lds_bytes<D, NBUF>());
}
template <int D, int NBUF, int NW, bool PRESC>
static hipError_t attn32_attr1() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&attention32_kernel<D, NBUF, NW, PRESC>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, attn32_lds_bytes<D, NBUF>());
}
template <int D>
static hipError_t attn_attr() {
    constexpr int NB = attn_nbuf<D>();
    hipError_t e = attn_attr1<D, NB, 1, false>();
    if (e == hipSuccess) e = attn_attr1<D, NB, 2, false>();
    if (e == hipSuccess) e = attn_attr1<D, NB, 1, true>();
    if (e == hipSuccess) e = attn_attr1<D, NB, 2, true>();
    if constexpr (D == 160) {   // the prefetching form of the 64-query workgroups (g_attn_d160_pipe)
        if (e == hipSuccess) e = attn_attr1<D, 2, 1, false>();
        if (e == hipSuccess) e = attn_attr1<D, 2, 1, true>();
    }
    if constexpr (attn_form32<D>()) {
        if (e == hipSuccess) e = attn32_attr1<D, NB, 2, false>();
        if (e == hipSuccess) e = attn32_attr1<D, NB, 4, false>();
        if (e == hipSuccess) e = attn32_attr1<D, NB, 2, true>();
        if (e == hipSuccess) e = attn32_attr1<D, NB, 4, true>();
        if (e == hipSuccess) e = attn32_attr1<D, 4, 2, false>();
        if (e == hipSuccess) e = attn32_attr1<D, 4, 4, false>();
        if (e == hipSuccess) e = attn32_attr1<D, 4, 2, true>();
        if (e == hipSuccess) e = attn32_attr1<D, 4, 4, true>();
        if (e == hipSuccess) e = attn32_attr1<D, 4, 8, false>();
        if (e == hipSuccess) e = attn32_attr1<D, 4, 8, true>();
    }
    return e;
}
int msd_attention_init() {
    if (g_attn_attr_done) return MSD_OK;
    hipError_t e = attn_attr<40>();
    if (e == hipSuccess) e = attn_attr<80>();
    if (e == hipSuccess) e = attn_attr<160>();
    if (e == hipSuccess) e = attn_attr<64>();
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention512_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, ATTN512_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_q_kernel<40, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, xattn_lds<40>());
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_q_kernel<40, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, xattn_lds<40>());
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_q_kernel<80, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, xattn_lds<80>());
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_q_kernel<80, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, xattn_lds<80>());
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_q160_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, xattn160_lds<4>());
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_q160_kernel<4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, xattn160_lds<4>());
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_q160_kernel<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, xattn160_lds<4>());
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_q160_kernel<4, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, xattn160_lds<4>());
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_q160_kernel<4, 7>), hipFuncAttributeMaxDynamicSharedMemorySize, xattn160_lds<4>());
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_q160_kernel<4, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, xattn160_lds<4>());
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_q160_kernel<4, 15>), hipFuncAttributeMaxDynamicSharedMemorySize, xattn160_lds<4>());
    if (e != hipSuccess) MSD_FAIL((int)e, "hipFuncSetAttribute(attention): %s", hipGetErrorString(e));
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            g_attn_cus = prop.multiProcessorCount;
    }
    g_attn_attr_done = true;
    return MSD_OK;
}

template <int D, int NBUF, int QF>
static void attn_launch2(const AArgs& a, dim3 grid, hipStream_t stream) {
    constexpr int lds = attn_lds_bytes<D, NBUF>();
    if (a.presc) hipLaunchKernelGGL((attention_kernel<D, NBUF, QF, true>), grid, dim3(256), lds, stream, ATTN_HOT_ARGS(a), a);
    else hipLaunchKernelGGL((attention_kernel<D, NBUF, QF, false>), grid, dim3(256), lds, stream, ATTN_HOT_ARGS(a), a);
}
template <int D, int NBUF, int NW>
static void attn32_launch2(const AArgs& a, dim3 grid, hipStream_t stream) {
    constexpr int lds = attn32_lds_bytes<D, NBUF>();
    constexpr int threads = 64 * (NW + attn32_loaders(NBUF, NW, D));   // (NBUF 4: the software-pipelined form, NW compute waves + the loader wave)
    if (a.presc) hipLaunchKernelGGL((attention32_kernel<D, NBUF, NW, true>), grid, dim3(threads), lds, stream, ATTN_HOT_ARGS(a), a);
    else hipLaunchKernelGGL((attention32_kernel<D, NBUF, NW, false>), grid, dim3(threads), lds, stream, ATTN_HOT_ARGS(a), a);
}
// The software-pipelined form needs a walk long enough to pay for its longer prologue (three tiles staged before the first
// product): key walks of 4 tiles and more.  A function of the head size and the key count only — the forms round differently.
static bool attn_swp(const AArgs& a, int d) { return g_attn_form == 2 && (d == 40 || d == 80) && a.t >= 256 && !a.causal; }

template <int D>
static void attn_launch(const AArgs& a, int qf, hipStream_t stream) {
    const int qt = 64 * qf;
    const dim3 grid(((a.q_count + qt - 1) / qt) * a.heads * a.batch);
    if constexpr (attn_form32<D>()) {
        if (attn_swp(a, D)) {
            if (qf == 1) attn32_launch2<D, 4, 2>(a, grid, stream);
            else if (qf == 2 || D != 40) attn32_launch2<D, 4, 4>(a, grid, stream);
            else attn32_launch2<D, 4, 8>(a, grid, stream);
            return;
        }
        if (g_attn_form >= 1) {
            if (qf == 1) attn32_launch2<D, attn_nbuf<D>(), 2>(a, grid, stream);
            else attn32_launch2<D, attn_nbuf<D>(), 4>(a, grid, stream);
            return;
        }
    }
    if constexpr (D == 160) {
        if (qf == 1 && g_attn_d160_pipe) { attn_launch2<D, 2, 1>(a, grid, stream); return; }
    }
    if (qf == 1) attn_launch2<D, attn_nbuf<D>(), 1>(a, grid, stream);
    else attn_launch2<D, attn_nbuf<D>(), 2>(a, grid, stream);
}

#ifdef MSD_STAMPS
// (tools, `make stamps` library only: workgroups of the software-pipelined form the runtime expects to fit on one CU)
extern "C" MSD_API int msd_debug_attn32_occupancy(int d, int nw, int presc) {
    int n = -1;
#define OCC(D_, NW_, P_)                                                                                            \
    if (d == D_ && nw == NW_ && presc == P_)                                                                        \
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(&attention32_kernel<D_, 4, NW_, P_>), \
                                                           64 * (NW_ + attn32_loaders(4, NW_, D_)), attn32_lds_bytes<D_, 4>());
    OCC(40, 2, true) OCC(40, 4, true) OCC(40, 8, true) OCC(80, 2, true) OCC(80, 4, true) OCC(40, 4, false) OCC(80, 2, false)
#undef OCC
    return n;
}
#endif

extern "C" int msd_attention(const MsdAttention* q, msd_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!q || !q->q || !q->k || !q->vt || !q->out) MSD_FAIL(MSD_E_ARG, "attention: null pointer");
    if (q->batch <= 0 || q->heads <= 0 || q->s <= 0 || q->t <= 0) MSD_FAIL(MSD_E_ARG, "attention: bad dims");
    if ((q->q_ld % 8) || (q->k_ld % 8) || (q->vt_ld % 8) || (q->o_ld % 4))
        MSD_FAIL(MSD_E_ALIGN, "attention: leading dimensions must be multiples of 8 (o_ld: 4)");
    if (q->q_ld < q->heads * q->head_dim || q->k_ld < q->heads * q->head_dim || q->o_ld < q->heads * q->head_dim ||
        q->vt_ld < q->t)
        MSD_FAIL(MSD_E_ARG, "attention: leading dimension smaller than the row");
    if (!msd_aligned16(q->q) || !msd_aligned16(q->k) || !msd_aligned16(q->vt) || !msd_aligned16(q->out))
        MSD_FAIL(MSD_E_ALIGN, "attention: pointers must be 16-byte aligned");
    int rc = msd_attention_init();
    if (rc) return rc;
    AArgs a;
    a.q = (const bf16_t*)q->q; a.k = (const bf16_t*)q->k; a.vt = (const bf16_t*)q->vt; a.out = (bf16_t*)q->out;
    a.batch = q->batch; a.heads = q->heads; a.s = q->s; a.t = q->t;
    a.q_ld = q->q_ld; a.k_ld = q->k_ld; a.vt_ld = q->vt_ld; a.o_ld = q->o_ld;
    a.sl2 = q->scale * 1.4426950408889634f;
    a.causal = q->causal ? 1 : 0;
    a.presc = q->q_prescaled ? 1 : 0;
    a.ws = nullptr; a.nsplit_shift = 0;
    if (a.causal && q->s != q->t) MSD_FAIL(MSD_E_ARG, "attention: causal masking needs s == t");
    // Workgroup size in queries: 128, or 64 when the 128-query grid has fewer workgroups than ~1.5 x the CUs (S = 1024
    // and below at batch 2: 128 / 32 / 8 workgroups).  Measured on one box (us, 128 vs 64 queries per workgroup): S=4096
    // d=40 116 / 125; S=1024 d=80 35 / 28; S=256 d=160 17 / 13; S=4096 T=77 8.8 / 9.8; S=9216 525 / 584.
    // The software-pipelined form at d = 40 takes 256 queries (8 compute waves + 2 loaders: one workgroup per CU is all that
    // fits beside the loaders' wave slots, so it had better be a big one) once that grid has >= 256 workgroups (g_attn_qf4_min); at d = 80
    // (201 registers: 8 wave slots per CU) 64 queries while the grid is small, 128 beyond.
    const long long wgs128 = (long long)((q->s + 127) / 128) * q->heads * q->batch;
    int qf = g_attn_qf ? g_attn_qf : (wgs128 < 384 ? 1 : 2);
    if (attn_swp(a, q->head_dim) && !g_attn_qf) {
        if (q->head_dim == 40) qf = wgs128 >= g_attn_qf4_min ? 4 : (wgs128 >= 128 ? 2 : 1);
        else qf = wgs128 <= 256 ? 1 : 2;
    }
    if (qf == 4 && !(attn_swp(a, q->head_dim) && q->head_dim == 40)) qf = 2;
    if (qf != 1 && qf != 4) qf = 2;
    a.q_begin = 0;
    a.q_count = q->s;
    a.mg_qtiles = udiv_magic_of((q->s + 64 * qf - 1) / (64 * qf));
    a.mg_heads = udiv_magic_of(q->heads);
    // 256-query workgroups whose last round of the chip is at most half full (768x768: 576 of them on 256 CUs = 2.25 rounds):
    // the queries of that round go to a second launch of 64-query workgroups, which fills every CU — a quarter of the work takes a
    // shorter round instead of a full one.  Scheduling only: the workgroup size changes no value.
    if (qf == 4 && !g_attn_qf) {
        const int cus = g_attn_cus;
        const long long bh = (long long)q->heads * q->batch, qt4 = (q->s + 255) / 256, tiles = qt4 * bh, rem = tiles % cus;
        const long long tail_qt = (rem + bh - 1) / bh;   // 256-query tiles per (batch, head) that make up the partial round
        if (tiles > cus && rem > 0 && 2 * rem <= cus && tail_qt < qt4 && q->head_dim == 40) {
            a.q_count = (int)((qt4 - tail_qt) * 256);
            a.mg_qtiles = udiv_magic_of((int)(qt4 - tail_qt));
            attn_launch<40>(a, 4, stream);
            MSD_CHECK_LAUNCH();
            a.q_begin = a.q_count;
            a.q_count = q->s - a.q_begin;
            const long long w64 = (long long)((a.q_count + 63) / 64) * bh;
            const int tqf = w64 <= 2ll * cus ? 1 : 2;
            a.mg_qtiles = udiv_magic_of((a.q_count + 64 * tqf - 1) / (64 * tqf));
            attn_launch<40>(a, tqf, stream);
            MSD_CHECK_LAUNCH();
            return MSD_OK;
        }
    }
    switch (q->head_dim) {
        case 40: attn_launch<40>(a, qf, stream); break;
        case 80: attn_launch<80>(a, qf, stream); break;
        case 160: attn_launch<160>(a, qf, stream); break;
        case 64: attn_launch<64>(a, qf, stream); break;
        case 512: {
            if ((q->t % 32) || a.causal || a.presc) MSD_FAIL(MSD_E_UNSUPPORTED, "attention: head_dim 512 needs t %% 32 == 0, no causal mask, no prescaled q");
            a.mg_qtiles = udiv_magic_of((q->s + 63) / 64);
            // key split: four parts when the caller lent a workspace and every part still walks >= 16 tiles; a function of the key
            // count alone (the parts are summed in part order: a sample's bits must not follow its batch)
            const int nsplit = (q->workspace && q->t >= 2048 && (q->t % 128) == 0) ? 4 : 1;
            const long long rows = (long long)q->batch * q->heads * q->s;
            if (nsplit > 1 && q->workspace_floats < (long long)nsplit * rows * (512 + 2))
                MSD_FAIL(MSD_E_WORKSPACE, "attention: head_dim 512 workspace too small (%lld < %lld floats)", (long long)q->workspace_floats,
                         (long long)nsplit * rows * (512 + 2));
            if (nsplit > 1 && (((uintptr_t)q->workspace) & 15u)) MSD_FAIL(MSD_E_ALIGN, "attention: workspace must be 16-byte aligned");
            a.ws = q->workspace; a.nsplit_shift = nsplit == 4 ? 2 : 0;
            hipLaunchKernelGGL(attention512_kernel, dim3(((q->s + 63) / 64) * q->heads * q->batch * nsplit), dim3(256), ATTN512_LDS, stream, a);
            if (nsplit > 1) {
                MSD_CHECK_LAUNCH();
                hipLaunchKernelGGL(attn512_merge_kernel, dim3((unsigned)((rows + 1) / 2)), dim3(256), 0, stream, (const float*)q->workspace, (bf16_t*)q->out,
                                   (int)rows, q->s, q->heads, q->o_ld, nsplit, a.sl2);
            }
            break;
        }
        default: MSD_FAIL(MSD_E_UNSUPPORTED, "attention: head_dim %d (supported: 40, 64, 80, 160, 512)", q->head_dim);
    }
    MSD_CHECK_LAUNCH();
    return MSD_OK;
}

extern "C" int msd_cross_attention_q(const MsdCrossAttnQ* q, msd_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!q || !q->x || !q->ln_in || !q->wq || !q->ln_colsum || !q->k || !q->vt || !q->out) MSD_FAIL(MSD_E_ARG, "cross_attention_q: null pointer");
    if (q->batch <= 0 || q->s <= 0 || q->t <= 0) MSD_FAIL(MSD_E_ARG, "cross_attention_q: bad dims");
    if (q->heads != 8 || (q->head_dim != 40 && q->head_dim != 80 && q->head_dim != 160) || q->t > 96)
        MSD_FAIL(MSD_E_UNSUPPORTED, "cross_attention_q: 8 heads of 40, 80 or 160 channels and at most 96 keys (heads=%d head_dim=%d t=%d)", q->heads,
                 q->head_dim, q->t);
    const int C = q->heads * q->head_dim;
    if (q->ln_in_slots < 1 || q->ln_in_slots > 20 || !(q->ln_eps > 0.f)) MSD_FAIL(MSD_E_ARG, "cross_attention_q: 1 <= ln_in_slots <= 20, ln_eps > 0");
    if ((q->k_ld % 8) || (q->vt_ld % 8) || (q->o_ld % 4) || q->k_ld < C || q->o_ld < C || q->vt_ld < q->t)
        MSD_FAIL(MSD_E_ALIGN, "cross_attention_q: leading dimensions");
    if (!msd_aligned16(q->x) || !msd_aligned16(q->wq) || !msd_aligned16(q->ln_colsum) || !msd_aligned16(q->bias) || !msd_aligned16(q->k) ||
        !msd_aligned16(q->vt) || !msd_aligned16(q->out) || (((uintptr_t)q->ln_in) & 7u))
        MSD_FAIL(MSD_E_ALIGN, "cross_attention_q: pointer alignment");
    if (q->w_layout != 0 && q->w_layout != 1) MSD_FAIL(MSD_E_ARG, "cross_attention_q: w_layout");
    int rc = msd_attention_init();
    if (rc) return rc;
    XArgs a;
    a.x = (const bf16_t*)q->x; a.ln_in = q->ln_in; a.wq = (const bf16_t*)q->wq; a.colsum = q->ln_colsum; a.bias = q->bias;
    a.k = (const bf16_t*)q->k; a.vt = (const bf16_t*)q->vt; a.out = (bf16_t*)q->out;
    a.batch = q->batch; a.heads = q->heads; a.s = q->s; a.t = q->t; a.c = C; a.k_ld = q->k_ld; a.vt_ld = q->vt_ld; a.o_ld = q->o_ld;
    a.ln_slots = q->ln_in_slots; a.ln_inv_k = 1.0f / (float)C; a.ln_eps = q->ln_eps;
    a.w_rs = q->w_layout ? 128u : (uint32_t)C * 2u;
    a.w_ks = q->w_layout ? (uint32_t)C * 128u : 128u;
    if (q->head_dim == 160) {   // streamed projection (xattn_q160_kernel): 64 queries x one head per workgroup, head = workgroup id & 7
        if ((long long)q->batch * q->s * C * 2 >= (1ll << 32)) MSD_FAIL(MSD_E_UNSUPPORTED, "cross_attention_q: batch * s too large");
        const int qtiles = (q->s + 63) / 64;
        a.mg_heads = udiv_magic_of(q->heads); a.mg_qtiles = udiv_magic_of(qtiles);
        const dim3 grid160((unsigned)qtiles * q->batch * 8u);
#ifdef MSD_STAMPS   // the kernel's experiment instantiations (deliberately wrong results) exist in the instrumented build only
        switch (g_xattn160_mode) {
            case 1: hipLaunchKernelGGL((xattn_q160_kernel<4, 1>), grid160, dim3(256), xattn160_lds<4>(), stream, XATTN_HOT_ARGS(a), a); break;
            case 2: hipLaunchKernelGGL((xattn_q160_kernel<4, 2>), grid160, dim3(256), xattn160_lds<4>(), stream, XATTN_HOT_ARGS(a), a); break;
            case 3: hipLaunchKernelGGL((xattn_q160_kernel<4, 3>), grid160, dim3(256), xattn160_lds<4>(), stream, XATTN_HOT_ARGS(a), a); break;
            case 7: hipLaunchKernelGGL((xattn_q160_kernel<4, 7>), grid160, dim3(256), xattn160_lds<4>(), stream, XATTN_HOT_ARGS(a), a); break;
            case 8: hipLaunchKernelGGL((xattn_q160_kernel<4, 8>), grid160, dim3(256), xattn160_lds<4>(), stream, XATTN_HOT_ARGS(a), a); break;
            case 15: hipLaunchKernelGGL((xattn_q160_kernel<4, 15>), grid160, dim3(256), xattn160_lds<4>(), stream, XATTN_HOT_ARGS(a), a); break;
            default: hipLaunchKernelGGL((xattn_q160_kernel<4>), grid160, dim3(256), xattn160_lds<4>(), stream, XATTN_HOT_ARGS(a), a);
        }
#else
        hipLaunchKernelGGL((xattn_q160_kernel<4>), grid160, dim3(256), xattn160_lds<4>(), stream, XATTN_HOT_ARGS(a), a);
#endif
        MSD_CHECK_LAUNCH();
        return MSD_OK;
    }
    // 128 queries per workgroup where the 64-query grid would need more than ~1.5 workgroups per CU (a query's result does
    // not depend on which queries share its workgroup)
    const long long wgs64 = (long long)((q->s + 63) / 64) * q->heads * q->batch;
    const int nw = g_xattn_nw ? g_xattn_nw : (wgs64 >= 768 ? 8 : 4);
    const int qtiles = (q->s + 16 * nw - 1) / (16 * nw);
    a.mg_heads = udiv_magic_of(q->heads); a.mg_qtiles = udiv_magic_of(qtiles);
    const dim3 grid((unsigned)qtiles * q->heads * q->batch);
    if (q->head_dim == 40) {
        if (nw == 8) hipLaunchKernelGGL((xattn_q_kernel<40, 8>), grid, dim3(512), xattn_lds<40>(), stream, XATTN_HOT_ARGS(a), a);
        else hipLaunchKernelGGL((xattn_q_kernel<40, 4>), grid, dim3(256), xattn_lds<40>(), stream, XATTN_HOT_ARGS(a), a);
    } else {
        if (nw == 8) hipLaunchKernelGGL((xattn_q_kernel<80, 8>), grid, dim3(512), xattn_lds<80>(), stream, XATTN_HOT_ARGS(a), a);
        else hipLaunchKernelGGL((xattn_q_kernel<80, 4>), grid, dim3(256), xattn_lds<80>(), stream, XATTN_HOT_ARGS(a), a);
    }
    MSD_CHECK_LAUNCH();
    return MSD_OK;
}

// ---- row softmax (VAE single-head attention scores, layers.py:48-50) ----------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* x, bf16_t* out, int cols, int ld_in, int ld_out,
                                                           float sl2) {
    __shared__ float red[4];
    const float* row = x + (size_t)blockIdx.x * ld_in;
    bf16_t* orow = out + (size_t)blockIdx.x * ld_out;
    const int t = threadIdx.x;
    const int ncv = cols >> 2;
    float mx = -1e30f;
    for (int cv = t; cv < ncv; cv += 256) {
        const float4 v = *reinterpret_cast<const float4*>(row + cv * 4);
        mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y) * 1.0f), fmaxf(v.z, v.w));
    }
    mx = wave_max(mx);
    if ((t & 63) == 0) red[t >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * sl2;  // scale > 0: max commutes with it
    __syncthreads();
    float sum = 0.f;
    for (int cv = t; cv < ncv; cv += 256) {
        const float4 v = *reinterpret_cast<const float4*>(row + cv * 4);
        sum += __builtin_amdgcn_exp2f(v.x * sl2 - mx) + __builtin_amdgcn_exp2f(v.y * sl2 - mx) +
               __builtin_amdgcn_exp2f(v.z * sl2 - mx) + __builtin_amdgcn_exp2f(v.w * sl2 - mx);
    }
    sum = wave_sum(sum);
    if ((t & 63) == 0) red[t >> 6] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    for (int cv = t; cv < ncv; cv += 256) {
        const float4 v = *reinterpret_cast<const float4*>(row + cv * 4);
        uint2 o;
        o.x = pack_bf2(__builtin_amdgcn_exp2f(v.x * sl2 - mx) * inv, __builtin_amdgcn_exp2f(v.y * sl2 - mx) * inv);
        o.y = pack_bf2(__builtin_amdgcn_exp2f(v.z * sl2 - mx) * inv, __builtin_amdgcn_exp2f(v.w * sl2 - mx) * inv);
        *reinterpret_cast<uint2*>(orow + cv * 4) = o;
    }
}

extern "C" int msd_softmax_rows(const float* x, void* out, int64_t rows, int32_t cols, int32_t ld_in, int32_t ld_out,
                                float scale, msd_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !out || rows <= 0 || cols <= 0 || (cols % 8) || (ld_in % 4) || (ld_out % 4) || ld_in < cols || ld_out < cols ||
        !(scale > 0.f))
        MSD_FAIL(MSD_E_ARG, "softmax_rows: bad arguments");
    if (!msd_aligned16(x) || !msd_aligned16(out)) MSD_FAIL(MSD_E_ALIGN, "softmax_rows: pointers must be 16-byte aligned");
    if (rows > 0x7fffffffLL) MSD_FAIL(MSD_E_ARG, "softmax_rows: too many rows");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, stream, x, (bf16_t*)out, cols, ld_in, ld_out,
                       scale * 1.4426950408889634f);
    MSD_CHECK_LAUNCH();
    return MSD_OK;
}

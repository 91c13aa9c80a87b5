// Row-panel Dense kernel: the A rows of a wave stay in REGISTERS for the whole launch, only the weights stream through LDS.
//
// The tile kernel (conv_gemm.hip) is bound by the LDS read port on the short-K Dense layers of the transformer blocks:
// with 32x32 wave tiles every K step reads 2 A + 2 W fragments for 4 MFMAs, 8 waves x 8 ds_read_b128 x 8 clocks = 512
// clocks of LDS port per K step against 128-256 clocks of MFMA (in-kernel stamps: 0.30 us per K step = 720 clocks), and
// every one of the N / BN column tiles re-stages the same A rows from L2.  Here a wave owns MI x 16 rows and loads their
// K <= 640 channels ONCE, straight from global memory into MFMA operand registers (80 / 160 VGPRs); a workgroup (4 waves
// = 128 rows) then walks its share of the output columns in steps of 32: per step the 32 x K weight tile arrives by LDS-DMA
// (3-stage ring), each wave reads 2 W fragments per 32-channel slice and issues 2 x MI = 4 MFMAs on them — 0.5 fragment
// reads per MFMA instead of 1.0 — and finishes the 32 columns (LayerNorm-fold correction, bias,
// GEGLU gate or q | k | v^T split, store) while the next weight tile is already in LDS.
//
// Serves the three LayerNorm-consumer GEMMs of a transformer block (q|k|v, attn2.to_q, the GEGLU projection:
// diffusion_model.py:102-108,132-153): ksize 1, one input tensor, K = 320 or 640, N % 32 == 0, no residual, bf16 output,
// 16-byte-aligned rows.  Everything else falls back to the tile kernel (host: msd_conv_gemm).
// Numerics: the K walk (64-channel chunks ascending, two 32-channel MFMA slices each) and every epilogue expression are
// those of conv_gemm_dma_kernel / cg_epilogue, so the bits are the tile kernel's (tests/test_ops_gpu.py compares them).
#include "conv_common.h"

// RL = the residual / LayerNorm-producer form (proj_in, attn1.to_out, attn2.to_out of a transformer block: plain mode, no
// activation, no LayerNorm-fold input): `residual` rows added in the 16-byte form, `ln_out` row-moment partials per 64-column
// tile — the numerics class of these layers (tuning.numerics_class) — accumulated in registers over the two 32-column
// steps of a tile in the tile kernel's canonical order.  At most RL_STEPS steps (128 columns) per workgroup: the residual
// of all of them is loaded with the A rows, before the column loop, so that no load sits between the weight DMAs.
constexpr int RL_STEPS = 4;
template <int KC, int MI, bool RL = false>
__global__ __launch_bounds__(256) void dense_rowpanel_kernel(const bf16_t* hot_a0, const bf16_t* hot_w, int hot_M, int hot_N, int hot_c0, int hot_tiles_m, int hot_tiles_n,
                                                             uint32_t hot_w_rs, uint32_t hot_w_ks, int wg_cols, int nsplits, uint32_t mg_nsplits, const CGArgs p) {
    // (leading scalars: kernarg preload, conv_common.h CG_HOT_PARAMS)
    constexpr int S = 3, NJ = 2, MP = MI / 2;
    constexpr int BM = 4 * MI * 16;
    constexpr int W_BYTES = KC * 4096;   // one step's weight tile: KC chunks x 32 weight rows x 128 B
    static_assert(MI % 2 == 0, "row fragments pair up for the 16-byte stores");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
    float* l_bias = reinterpret_cast<float*>(smem + S * W_BYTES);
    float* l_cs = l_bias + wg_cols;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int tile = xcd_remap(blockIdx.x, hot_tiles_m * hot_tiles_n);   // (= gridDim.x; consecutive tiles = one XCD = one row panel)
    const int panel = udiv_magic(tile, nsplits, mg_nsplits), ns = tile - panel * nsplits;
    const int m0 = panel * BM + wave * (MI * 16);
    const int n0 = ns * wg_cols;
    const int nsteps = (min(wg_cols, hot_N - n0)) >> 5;

    // ---- weight DMA: thread -> (row = tid >> 3 of the 32, 16-byte piece tid & 7), one instruction per 64-channel chunk
    const int cpos = tid & 7, lrow = tid >> 3;
    const uint32_t wsw = (uint32_t)((cpos ^ ((lrow >> 1) & 7)) * 16);
    const uint32_t lds_wave = lds0 + (uint32_t)(wave * 8) * 128u;
    auto issue_w = [&](int step, int stage) {
        const uint32_t off = (uint32_t)min(n0 + step * 32 + lrow, hot_N - 1) * hot_w_rs + wsw;
        const uint32_t base = __builtin_amdgcn_readfirstlane(lds_wave + (uint32_t)stage * W_BYTES);
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) dma16s(hot_w, off + (uint32_t)kc * hot_w_ks, base + (uint32_t)kc * 4096u);
    };
    if (nsteps > 0) issue_w(0, 0);
    if (nsteps > 1) issue_w(1, 1);

    // ---- this wave's rows into MFMA operand registers: lane (r, g) holds channels 32 ks + 8 g .. + 8 of row m0 + 16 i + r
    bf16x8 af[MI][KC * 2];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const bf16_t* src = hot_a0 + (size_t)min(m0 + i * 16 + r, hot_M - 1) * hot_c0 + g * 8;
#pragma unroll
        for (int ks = 0; ks < KC * 2; ++ks) af[i][ks] = *reinterpret_cast<const bf16x8*>(src + ks * 32);
    }
    // bias and LayerNorm column sums of this workgroup's columns -> LDS (read per step without a memory round trip)
    for (int t = tid; t < wg_cols; t += 256) {
        const int n = min(n0 + t, p.N - 1);
        l_bias[t] = p.bias ? p.bias[n] : 0.f;
        l_cs[t] = (!RL && p.ln_in) ? p.ln_colsum[n] : 0.f;
    }
    const int cgo = cg_col(g), cg8 = (g & 1) << 3;
    const int rw = cg_wrow(r);
    int msw[MP], bidx[MI];
#pragma unroll
    for (int ip = 0; ip < MP; ++ip) msw[ip] = m0 + (g < 2 ? 2 * ip : 2 * ip + 1) * 16 + r;
    // RL: the residual runs of every step (lane = row msw[ip], 8 channels from n0 + 32 step + 16 j + cg8)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 rres[RL ? RL_STEPS : 1][MP][NJ];
    if (RL) {
#pragma unroll
        for (int st = 0; st < RL_STEPS; ++st)
#pragma unroll
            for (int ip = 0; ip < MP; ++ip)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    rres[st][ip][j] = (u32x4){0, 0, 0, 0};
                    if (p.residual && st < nsteps)   // (wave-uniform)
                        rres[st][ip][j] = *reinterpret_cast<const u32x4*>(p.residual + (size_t)min(msw[ip], p.M - 1) * p.res_ld +
                                                                          min(n0 + st * 32 + j * 16 + cg8, p.N - 8));
                }
    }
    // LayerNorm fold, consumer side: row moments from the producer's partials, summed exactly as cg_epilogue does
    // (lane group g takes slots g, g + 4, ...; then (g0 + g1) + (g2 + g3)) — once per row instead of once per tile
    constexpr int LNS = LN_MAX_SLOTS / 4;
    float mean[MI], rstd[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) { mean[i] = 0.f; rstd[i] = 1.f; }
    if (!RL && p.ln_in) {
        float2 lnp[MI][LNS];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const float2* src = reinterpret_cast<const float2*>(p.ln_in) + (size_t)min(m0 + i * 16 + r, p.M - 1) * p.ln_in_slots;
#pragma unroll
            for (int k = 0; k < LNS; ++k) lnp[i][k] = src[min(g + 4 * k, p.ln_in_slots - 1)];
        }
        wait_vmcnt<0>();
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int k = 0; k < LNS; ++k)
                if (g + 4 * k < p.ln_in_slots) { s1 += lnp[i][k].x; s2 += lnp[i][k].y; }
            s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
            s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
            mean[i] = s1 * p.ln_inv_k;
            rstd[i] = rsqrtf(fmaxf(s2 * p.ln_inv_k - mean[i] * mean[i], 0.f) + p.ln_eps);
        }
    }
    wait_vmcnt<0>();   // A rows, weight tiles 0 and 1
    // The compiler does not see that wait (nor the DMAs): left alone it puts its OWN waits for the A-row loads on their
    // first uses inside the column loop — s_waitcnt vmcnt(19) ... vmcnt(1) in front of the MFMAs of every step — and from the
    // second step on those count the weight DMAs of tile step + 2 and the previous step's stores instead: every step then
    // stalls until the tile it has just requested has landed, a ring of depth one.  Consuming the registers here retires the
    // loads in the compiler's book-keeping before the loop.
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int ks = 0; ks < KC * 2; ++ks) asm volatile("" : "+v"(af[i][ks]));
    if (RL) {
#pragma unroll
        for (int st = 0; st < RL_STEPS; ++st)
#pragma unroll
            for (int ip = 0; ip < MP; ++ip)
#pragma unroll
                for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(rres[st][ip][j]));
    }
    __syncthreads();   // ... and the bias / column-sum rows in LDS

    float lna[MP], lnq[MP];   // RL + ln_out: running moments of the lane's row over the 16-column blocks of the current 64-column tile
#pragma unroll
    for (int ip = 0; ip < MP; ++ip) { lna[ip] = 0.f; lnq[ip] = 0.f; }
#pragma unroll
    for (int i = 0; i < MI; ++i) bidx[i] = p.split_mode ? udiv_magic(min(m0 + i * 16 + r, p.M - 1), p.hw_out, p.mg_hw) : 0;
    auto swap8 = [&](const float (&a)[4], const float (&b)[4], float (&v)[8]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[e]), __float_as_uint(b[e]), false, false);
            v[e] = __uint_as_float(sw[0]);
            v[4 + e] = __uint_as_float(sw[1]);
        }
    };

    int stage = 0;
    for (int step = 0; step < nsteps; ++step) {
        // Retire weight tile `step`.  Younger in the queue, in issue order: the KC DMAs of tile step + 1, then the stores
        // of the previous step's epilogue; vmcnt(KC) therefore lets stores (and what is left of the KC) stay in flight.
        // The LAST step has no tile step + 1 behind it: the KC youngest operations are then the two previous epilogues' stores AND
        // the youngest DMAs of tile `step` itself, so vmcnt(KC) would let the step read a tile that has not landed (round 5: found
        // as a 1-in-400 run-to-run difference of the GEGLU projections when their weights came from HBM; never seen on hot
        // weights).  It waits for everything.
        if (step) {
            if (step + 1 < nsteps) wait_vmcnt<KC>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();   // tile `step` visible to all waves; stage (step - 1) % S free
        }
        if (step + 2 < nsteps) {
            int st = stage + 2;
            if (st >= S) st -= S;
            issue_w(step + 2, st);
        }
        f32x4 acc[NJ][MI];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < MI; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const char* bW = smem + stage * W_BYTES + rw * 128;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            bf16x8 wf[2][NJ];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    wf[ks][j] = *reinterpret_cast<const bf16x8*>(bW + kc * 4096 + j * 16 * 128 + (((ks * 4 + g) ^ (rw >> 1)) << 4));
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int i = 0; i < MI; ++i)
                        acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][j], af[i][kc * 2 + ks], acc[j][i], 0, 0, 0);
        }
        // ---- the 32 columns nb .. nb + 31 of this wave's rows
        const int nb = n0 + step * 32;
        float4 bv[NJ], cs[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            bv[j] = *reinterpret_cast<const float4*>(l_bias + step * 32 + j * 16 + cgo);
            cs[j] = *reinterpret_cast<const float4*>(l_cs + step * 32 + j * 16 + cgo);
        }
        if (!RL && p.ln_in) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    acc[j][i][0] = rstd[i] * (acc[j][i][0] - mean[i] * cs[j].x);
                    acc[j][i][1] = rstd[i] * (acc[j][i][1] - mean[i] * cs[j].y);
                    acc[j][i][2] = rstd[i] * (acc[j][i][2] - mean[i] * cs[j].z);
                    acc[j][i][3] = rstd[i] * (acc[j][i][3] - mean[i] * cs[j].w);
                }
        }
        if constexpr (RL) {
            // cg_epilogue's compact 16-byte form, expression for expression: (acc + bias) + time-embedding row (absent: 0),
            // half-wave exchange, + residual run (absent: zeros), round to bf16, store; moments of the ROUNDED values per 16-column
            // block as (s[0..3] + s[4..7]) + (s[8..11] + s[12..15]), blocks of a 64-column tile added in ascending order from 0
            const float zero = 0.f;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    acc[j][i][0] = (acc[j][i][0] + bv[j].x) + zero; acc[j][i][1] = (acc[j][i][1] + bv[j].y) + zero;
                    acc[j][i][2] = (acc[j][i][2] + bv[j].z) + zero; acc[j][i][3] = (acc[j][i][3] + bv[j].w) + zero;
                }
            if ((step & 1) == 0) {
#pragma unroll
                for (int ip = 0; ip < MP; ++ip) { lna[ip] = 0.f; lnq[ip] = 0.f; }
            }
#pragma unroll
            for (int ip = 0; ip < MP; ++ip)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    float a[4] = {acc[j][2 * ip][0], acc[j][2 * ip][1], acc[j][2 * ip][2], acc[j][2 * ip][3]};
                    float b[4] = {acc[j][2 * ip + 1][0], acc[j][2 * ip + 1][1], acc[j][2 * ip + 1][2], acc[j][2 * ip + 1][3]};
                    float v[8], q[8];
                    swap8(a, b, v);
                    // (the step index of the residual registers must be a compile-time constant: select, do not index)
                    u32x4 rr = rres[0][ip][j];
#pragma unroll
                    for (int st = 1; st < RL_STEPS; ++st) if (step == st) rr = rres[st][ip][j];
                    unpack8(make_uint4(rr.x, rr.y, rr.z, rr.w), q);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] + q[e];
                    const uint4 o = pack8(v);
                    const bool ok = msw[ip] < p.M;
                    if (ok) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)msw[ip] * p.out_ld + nb + j * 16 + cg8) = o;
                    if (p.ln_out) {   // (every lane takes part in the row swap: predicated values, no early exit)
                        unpack8(o, q);
                        float s1 = (ok ? (q[0] + q[1]) + (q[2] + q[3]) : 0.f) + (ok ? (q[4] + q[5]) + (q[6] + q[7]) : 0.f);
                        float s2 = (ok ? (q[0] * q[0] + q[1] * q[1]) + (q[2] * q[2] + q[3] * q[3]) : 0.f) +
                                   (ok ? (q[4] * q[4] + q[5] * q[5]) + (q[6] * q[6] + q[7] * q[7]) : 0.f);
                        auto t1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(s1), __float_as_uint(s1), false, false);
                        auto t2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(s2), __float_as_uint(s2), false, false);
                        lna[ip] += __uint_as_float(t1[0]) + __uint_as_float(t1[1]);   // the two 8-channel halves of the block
                        lnq[ip] += __uint_as_float(t2[0]) + __uint_as_float(t2[1]);
                    }
                }
            if (p.ln_out && (step & 1) && (g & 1) == 0) {   // tile (nb - 32) / 64 complete: lanes g = 0 / 2 hold rows of fragments 2 ip / 2 ip + 1
#pragma unroll
                for (int ip = 0; ip < MP; ++ip)
                    if (msw[ip] < p.M)
                        reinterpret_cast<float2*>(p.ln_out)[(size_t)msw[ip] * p.ln_out_slots + (nb >> 6)] = make_float2(lna[ip], lnq[ip]);
            }
        } else if (p.act == MSD_ACT_GEGLU) {   // x columns nb + [0,16), gate nb + 16 + [0,16) -> output columns nb / 2 + [0,16)
            auto gl4 = [&](int i, float (&v)[4]) {
                v[0] = geglu_f(acc[0][i][0] + bv[0].x, acc[1][i][0] + bv[1].x);
                v[1] = geglu_f(acc[0][i][1] + bv[0].y, acc[1][i][1] + bv[1].y);
                v[2] = geglu_f(acc[0][i][2] + bv[0].z, acc[1][i][2] + bv[1].z);
                v[3] = geglu_f(acc[0][i][3] + bv[0].w, acc[1][i][3] + bv[1].w);
            };
#pragma unroll
            for (int ip = 0; ip < MP; ++ip) {
                float a[4], b[4], v[8];
                gl4(2 * ip, a);
                gl4(2 * ip + 1, b);
                swap8(a, b, v);
                if (msw[ip] < p.M)
                    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)msw[ip] * p.out_ld + (nb >> 1) + cg8) = pack8(v);
            }
        } else {
            const float zero = 0.f;   // (the tile kernel's epilogue adds its absent time-embedding row as 0.f: mirrored)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    acc[j][i][0] = (acc[j][i][0] + bv[j].x) + zero; acc[j][i][1] = (acc[j][i][1] + bv[j].y) + zero;
                    acc[j][i][2] = (acc[j][i][2] + bv[j].z) + zero; acc[j][i][3] = (acc[j][i][3] + bv[j].w) + zero;
                }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int nbj = nb + j * 16;
                if (p.split_mode == 0 || nbj < p.ns0 + p.ns1) {   // plain output, or the q | k parts (a 16-column block lies in one part)
#pragma unroll
                    for (int ip = 0; ip < MP; ++ip) {
                        float a[4] = {acc[j][2 * ip][0], acc[j][2 * ip][1], acc[j][2 * ip][2], acc[j][2 * ip][3]};
                        float b[4] = {acc[j][2 * ip + 1][0], acc[j][2 * ip + 1][1], acc[j][2 * ip + 1][2], acc[j][2 * ip + 1][3]};
                        float v[8];
                        swap8(a, b, v);
                        if (p.split_mode == 0) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = v[e] + zero;   // (the absent residual of the tile kernel's 16-byte form)
                        }
                        const int n8 = nbj + cg8;
                        if (msw[ip] >= p.M) continue;
                        bf16_t* dst = (p.split_mode == 0 || n8 < p.ns0) ? reinterpret_cast<bf16_t*>(p.out) + (size_t)msw[ip] * p.out_ld + n8
                                                                         : p.out1 + (size_t)msw[ip] * p.out1_ld + (n8 - p.ns0);
                        *reinterpret_cast<uint4*>(dst) = pack8(v);
                    }
                } else {   // v part: transposed, token contiguous
                    const int nv = p.N - p.ns0 - p.ns1;
                    const int nn = nbj + cgo - p.ns0 - p.ns1;
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        const int m = m0 + i * 16 + r;
                        if (m >= p.M) continue;
                        const int sidx = m - bidx[i] * p.hw_out;
                        const uint32_t ox = pack_bf2(acc[j][i][0], acc[j][i][1]), oy = pack_bf2(acc[j][i][2], acc[j][i][3]);
                        bf16_t* dst = p.out2 + ((size_t)bidx[i] * nv + nn) * p.out2_ld + sidx;
                        dst[0] = (bf16_t)(ox & 0xFFFF);
                        dst[(size_t)p.out2_ld] = (bf16_t)(ox >> 16);
                        dst[(size_t)2 * p.out2_ld] = (bf16_t)(oy & 0xFFFF);
                        dst[(size_t)3 * p.out2_ld] = (bf16_t)(oy >> 16);
                    }
                }
            }
        }
        if (++stage == S) stage = 0;
    }
}

static bool g_rp_attr_done = false;
constexpr int rp_lds(int kc, int wg_cols) { return 3 * kc * 4096 + 2 * wg_cols * 4; }
constexpr int RP_MAX_COLS = 1024;   // columns per workgroup (bias + column sums in LDS)

int msd_conv_rowpanel_init() {
    if (g_rp_attr_done) return MSD_OK;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_rowpanel_kernel<5, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, rp_lds(5, RP_MAX_COLS));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_rowpanel_kernel<10, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, rp_lds(10, RP_MAX_COLS));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_rowpanel_kernel<5, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, rp_lds(5, 32 * RL_STEPS));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_rowpanel_kernel<10, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, rp_lds(10, 32 * RL_STEPS));
    if (e != hipSuccess) MSD_FAIL((int)e, "hipFuncSetAttribute(conv_rowpanel): %s", hipGetErrorString(e));
    g_rp_attr_done = true;
    return MSD_OK;
}

// Whether a validated launch can run here: rows (256 or 128 per workgroup) and columns per workgroup as requested.
bool msd_conv_rowpanel_eligible(const CGArgs& a, int rows, int wg_cols) {
    const bool rl = a.residual || a.ln_out;   // the residual / LayerNorm-producer form: whole 64-column tiles, <= 128 columns per workgroup
    const bool shape = a.ksize == 1 && a.stride == 1 && !a.upsample && a.c1 == 0 && !a.a2 && (a.K == 320 || a.K == 640) && (a.N % 32) == 0 &&
                       !a.rowvec && !a.out_f32 && a.nslices == 1 && a.vec16 &&
                       (a.act == MSD_ACT_NONE || (a.act == MSD_ACT_GEGLU && a.split_mode == 0)) &&
                       (!rl || (a.act == MSD_ACT_NONE && a.split_mode == 0 && !a.ln_in && (a.N % 64) == 0 && (wg_cols % 64) == 0 &&
                                wg_cols <= 32 * RL_STEPS)) &&
                       (long long)a.M * a.c0 * 2 < (1ll << 32) - 4096;
    // (MI = 4, 256 rows per workgroup, was built for K = 320: 256 VGPRs, one wave per SIMD — slower than MI = 2 on every
    //  shape, 27.5 vs 24.5 us on the 64x64 GEGLU projection: with two workgroups per CU one's epilogue overlaps the other's MFMAs)
    const bool cfg = rows == 128;
    return shape && cfg && wg_cols >= 32 && wg_cols <= RP_MAX_COLS && (wg_cols % 32) == 0;
}

int msd_conv_rowpanel_launch(CGArgs a, int rows, int wg_cols, hipStream_t stream) {
    int rc = msd_conv_rowpanel_init();
    if (rc) return rc;
    if (wg_cols > a.N) wg_cols = a.N;
    const int panels = (a.M + rows - 1) / rows, nsplits = (a.N + wg_cols - 1) / wg_cols;
    a.tiles_m = panels; a.tiles_n = nsplits;
    const dim3 grid(panels * nsplits);
    const uint32_t mg = udiv_magic_of(nsplits);
    const bool rl = a.residual || a.ln_out;
    if (a.K == 320) {
        if (rl) hipLaunchKernelGGL((dense_rowpanel_kernel<5, 2, true>), grid, dim3(256), rp_lds(5, wg_cols), stream, a.a0, a.w, a.M, a.N, a.c0, a.tiles_m, a.tiles_n, a.w_rs, a.w_ks, wg_cols, nsplits, mg, a);
        else hipLaunchKernelGGL((dense_rowpanel_kernel<5, 2>), grid, dim3(256), rp_lds(5, wg_cols), stream, a.a0, a.w, a.M, a.N, a.c0, a.tiles_m, a.tiles_n, a.w_rs, a.w_ks, wg_cols, nsplits, mg, a);
    } else {
        if (rl) hipLaunchKernelGGL((dense_rowpanel_kernel<10, 2, true>), grid, dim3(256), rp_lds(10, wg_cols), stream, a.a0, a.w, a.M, a.N, a.c0, a.tiles_m, a.tiles_n, a.w_rs, a.w_ks, wg_cols, nsplits, mg, a);
        else hipLaunchKernelGGL((dense_rowpanel_kernel<10, 2>), grid, dim3(256), rp_lds(10, wg_cols), stream, a.a0, a.w, a.M, a.N, a.c0, a.tiles_m, a.tiles_n, a.w_rs, a.w_ks, wg_cols, nsplits, mg, a);
    }
    return MSD_OK;
}

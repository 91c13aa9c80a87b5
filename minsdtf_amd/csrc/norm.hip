// GroupNormalization(32)(+swish) and LayerNormalization for NHWC bf16 tensors, fp32 statistics.
//
// Replaces keras GroupNormalization / Activation("swish") / LayerNormalization on the hot path
// (diffusion_model.py:27-28,32-33,57,84-88,277-278; layers.py:32,66-68,78-79;
// image_decoder.py:51-52).  These kernels are HBM-bound: every access is a 16-byte vector
// (8 bf16 channels) and consecutive lanes cover consecutive channels of one pixel, so a wave
// instruction touches whole 128-byte lines.  The channel concat of the UNet up path is read
// through two base pointers (x0|x1) and never materialised.
#include "common.h"

struct GNArgs {
    const bf16_t* x0; const bf16_t* x1;
    const float* gamma; const float* beta;
    float* stats; float* partials; bf16_t* out;
    int batch, hw, c0, c1, C, cv, tpp, pl, ppb, silu;
    float eps;
    uint32_t mg_tpp, mg_cpg;   // floor(2^32 / d) for d = tpp, C / 32 (udiv_magic: these launches are latency chains of scalar code)
    int nchunks_stats;         // grid x of the statistics pass (kept here: reading gridDim costs a second kernarg round trip)
};

// Kernarg preload (see conv_common.h CG_HOT_PARAMS): what the single-launch kernels need before their first load, as leading scalars
#define GN_HOT_PARAMS const bf16_t* hot_x0, const bf16_t* hot_x1, int hot_C, int hot_hw, int hot_c0, int hot_c1, int upp, int ppp, uint32_t mg_upp
#define GN_HOT_ARGS(a) (a).x0, (a).x1, (a).C, (a).hw, (a).c0, (a).c1, upp, ppp, mg

// NT threads per workgroup: 256, or 1024 for the mid-sized tensors (the UNet's 64x64 level) where a
// workgroup's whole 64-pixel chunk is then in flight at once — those launches are bounded by memory
// latency x bytes in flight, not by bandwidth.
template <int VPT, int NT>
__global__ __launch_bounds__(NT) void gn_stats_kernel(const GNArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* part = reinterpret_cast<float*>(smem_raw);  // [pl][C][2]
    float* chan = part + (size_t)p.pl * p.C * 2;        // [C][2]
    const int t = threadIdx.x, b = blockIdx.y;
    const int pli = udiv_magic(t, p.tpp, p.mg_tpp), cvi = t - pli * p.tpp;
    const bool active = pli < p.pl;
    const int p_begin = blockIdx.x * p.ppb;
    const int p_end = min(p.hw, p_begin + p.ppb);

    float s[VPT][8], ss[VPT][8];
#pragma unroll
    for (int v = 0; v < VPT; ++v)
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[v][e] = 0.f; ss[v][e] = 0.f; }

    if (active) {
        // these tensors are small (launch / latency bound): keep U independent 16-byte loads in
        // flight per thread instead of one load -> accumulate -> next load
        constexpr int U = 4;
        const bf16_t* base[VPT];
        int cstride[VPT];
        bool on[VPT];
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const int cv = cvi + v * p.tpp;
            on[v] = cv < p.cv;
            const int c = cv * 8;
            const bool first = c < p.c0;
            cstride[v] = first ? p.c0 : p.c1;
            base[v] = (first ? p.x0 + c : p.x1 + (c - p.c0)) + (size_t)b * p.hw * cstride[v];
        }
        for (int px = p_begin + pli; px < p_end; px += U * p.pl) {
            uint4 raw[U][VPT];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int v = 0; v < VPT; ++v) {
                    const int q = px + u * p.pl;
                    raw[u][v] = (on[v] && q < p_end) ? *reinterpret_cast<const uint4*>(base[v] + (size_t)q * cstride[v])
                                                     : make_uint4(0, 0, 0, 0);
                }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int v = 0; v < VPT; ++v) {
                    float f[8];
                    unpack8(raw[u][v], f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { s[v][e] += f[e]; ss[v][e] += f[e] * f[e]; }
                }
        }
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const int cv = cvi + v * p.tpp;
            if (cv < p.cv) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    part[((size_t)pli * p.C + cv * 8 + e) * 2 + 0] = s[v][e];
                    part[((size_t)pli * p.C + cv * 8 + e) * 2 + 1] = ss[v][e];
                }
            }
        }
    }
    __syncthreads();
    for (int c = t; c < p.C; c += NT) {
        float a = 0.f, q = 0.f;
        for (int l = 0; l < p.pl; ++l) {
            a += part[((size_t)l * p.C + c) * 2 + 0];
            q += part[((size_t)l * p.C + c) * 2 + 1];
        }
        chan[c * 2 + 0] = a;
        chan[c * 2 + 1] = q;
    }
    __syncthreads();
    if (t < 32) {
        const int cpg = p.C / 32;
        float a = 0.f, q = 0.f;
        for (int c = t * cpg; c < (t + 1) * cpg; ++c) { a += chan[c * 2]; q += chan[c * 2 + 1]; }
        // per-workgroup partial moments; summed in a fixed order by gn_finalize_kernel (deterministic)
        float* dst = p.partials + (((size_t)b * p.nchunks_stats + blockIdx.x) * 32 + t) * 2;
        dst[0] = a;
        dst[1] = q;
    }
}

// one 64-thread workgroup per sample: ordered sum of the partials -> {mean, rstd} per group
__global__ __launch_bounds__(1024) void gn_finalize_kernel(const GNArgs p, int nchunks) {
    // 16 row groups x 64 columns: each thread sums every 16th chunk, then the 16 partial sums are
    // added in a fixed order -> same bits every run
    __shared__ float red[16][64];
    const int b = blockIdx.x, t = threadIdx.x & 63, rg = threadIdx.x >> 6;  // t = group*2 + {0: sum, 1: sumsq}
    const float* src = p.partials + (size_t)b * nchunks * 64 + t;
    float part = 0.f;
    // 8 loads in flight per thread (a load-add-load chain pays a memory round trip per chunk: 17 us for the VAE's 1024
    // chunks); the adds keep the chunk order
    for (int c0 = rg; c0 < nchunks; c0 += 16 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(size_t)min(c0 + 16 * u, nchunks - 1) * 64];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (c0 + 16 * u < nchunks) part += v[u];
    }
    red[rg][t] = part;
    __syncthreads();
    if (rg != 0) return;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += red[i][t];
    const float other = __shfl_xor(acc, 1);
    const float sum = (t & 1) ? other : acc, sq = (t & 1) ? acc : other;
    const float cnt = (float)p.hw * (float)(p.C / 32);
    const float mean = sum / cnt;
    const float var = fmaxf(sq / cnt - mean * mean, 0.f);
    p.stats[(size_t)b * 64 + t] = (t & 1) ? rsqrtf(var + p.eps) : mean;
}

// FUSED: few chunks (<= 64) -> every workgroup reduces the partial moments itself, in the same fixed
// order, and the separate finalize launch disappears (small tensors are launch-bound, not byte-bound)
template <int VPT, bool FUSED, int NT>
__global__ __launch_bounds__(NT) void gn_apply_kernel(const GNArgs p, int nchunks) {
    __shared__ float s_mean[32], s_rstd[32];
    __shared__ float red[NT / 64][64];
    constexpr int U = 4;  // pixels in flight per thread
    const int t = threadIdx.x, b = blockIdx.y;
    const int pli = udiv_magic(t, p.tpp, p.mg_tpp), cvi = t - pli * p.tpp;
    const bool active = pli < p.pl;
    const int p_begin = blockIdx.x * p.ppb;
    const int p_end = min(p.hw, p_begin + p.ppb);

    // (1) everything that does not depend on the statistics is issued FIRST, so its latency runs
    //     under the prologue: the first U pixel vectors, gamma and beta
    const bf16_t* base[VPT];
    int cstride[VPT];
    bool on[VPT];
    float4 g0[VPT], g1[VPT], b0[VPT], b1[VPT];
    uint4 raw[U][VPT];
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        const int cv = cvi + v * p.tpp;
        on[v] = active && cv < p.cv;
        const int c = on[v] ? cv * 8 : 0;
        const bool first = c < p.c0;
        cstride[v] = first ? p.c0 : p.c1;
        base[v] = (first ? p.x0 + c : p.x1 + (c - p.c0)) + (size_t)b * p.hw * cstride[v];
        g0[v] = *reinterpret_cast<const float4*>(p.gamma + c);
        g1[v] = *reinterpret_cast<const float4*>(p.gamma + c + 4);
        b0[v] = *reinterpret_cast<const float4*>(p.beta + c);
        b1[v] = *reinterpret_cast<const float4*>(p.beta + c + 4);
    }
    auto load_batch = [&](int px) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                const int q = px + u * p.pl;
                if (on[v] && q < p_end) raw[u][v] = *reinterpret_cast<const uint4*>(base[v] + (size_t)q * cstride[v]);
            }
    };
    load_batch(p_begin + pli);

    // (2) statistics: reduce the per-workgroup partial moments in a fixed order (FUSED) or read them
    if (FUSED) {
        const int col = t & 63, rg = t >> 6;
        const float* src = p.partials + (size_t)b * nchunks * 64 + col;
        float part = 0.f;
        for (int c = rg; c < nchunks; c += NT / 64) part += src[(size_t)c * 64];
        red[rg][col] = part;
        __syncthreads();
        if (t < 64) {
            float acc = red[0][t];
#pragma unroll
            for (int w = 1; w < NT / 64; ++w) acc += red[w][t];   // fixed order: bit-reproducible
            const float other = __shfl_xor(acc, 1);
            const float sum = (t & 1) ? other : acc, sq = (t & 1) ? acc : other;
            const float cnt = (float)p.hw * (float)(p.C / 32);
            const float mean = sum / cnt;
            const float var = fmaxf(sq / cnt - mean * mean, 0.f);
            const float rstd = rsqrtf(var + p.eps);
            if (t & 1) s_rstd[t >> 1] = rstd;
            else s_mean[t >> 1] = mean;
            if (blockIdx.x == 0) p.stats[(size_t)b * 64 + t] = (t & 1) ? rstd : mean;   // part of the contract on every path
        }
    } else if (t < 32) {
        s_mean[t] = p.stats[((size_t)b * 32 + t) * 2 + 0];
        s_rstd[t] = p.stats[((size_t)b * 32 + t) * 2 + 1];
    }
    __syncthreads();
    if (!active) return;

    // (3) per-channel scale / shift
    const int cpg = p.C / 32;
    float ca[VPT][8], cb[VPT][8];
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        const int c0 = (cvi + v * p.tpp) * 8;
        const float gm[8] = {g0[v].x, g0[v].y, g0[v].z, g0[v].w, g1[v].x, g1[v].y, g1[v].z, g1[v].w};
        const float bt[8] = {b0[v].x, b0[v].y, b0[v].z, b0[v].w, b1[v].x, b1[v].y, b1[v].z, b1[v].w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int g = on[v] ? udiv_magic(c0 + e, cpg, p.mg_cpg) : 0;
            const float a = s_rstd[g] * gm[e];
            ca[v][e] = a;
            cb[v][e] = bt[e] - s_mean[g] * a;
        }
    }
    // (4) normalise (+ SiLU) and store, U pixels per trip
    for (int px = p_begin + pli; px < p_end; px += U * p.pl) {
        if (px != p_begin + pli) load_batch(px);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                const int q = px + u * p.pl;
                if (on[v] && q < p_end) {
                    float f[8];
                    unpack8(raw[u][v], f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float y = f[e] * ca[v][e] + cb[v][e];
                        f[e] = p.silu ? silu_f(y) : y;
                    }
                    *reinterpret_cast<uint4*>(p.out + ((size_t)b * p.hw + q) * p.C + (cvi + v * p.tpp) * 8) = pack8(f);
                }
            }
    }
}

// ---- single-launch GroupNorm for the tensors of the UNet --------------------------------------
// One 1024-thread workgroup per (sample, group); the group's H*W x (C/32) slab (<= 256 KB) is read
// ONCE into registers, reduced (fixed order: bit-reproducible), normalised and written.  Replaces
// stats + apply (two launches, two reads) for every tensor whose group slab fits: at batch 1-2 these
// layers are bounded by launch latency, not bytes.  A thread owns V consecutive 4-byte words (2V
// channels) at a FIXED channel offset and walks the pixels with a constant stride, so gamma / beta
// and the concat source (x0 or x1) are per-thread constants.  A group's channels are a 20..160-byte
// run inside each pixel's row, so a wave touches several lines per instruction — fine at this size.
// Workgroup ids are dealt so that one XCD owns 4 adjacent groups (each 128-byte line is then pulled
// into one or two of the 8 private L2s, not into six).
template <int V> struct gn_vec;
template <> struct gn_vec<1> { typedef uint32_t type; };
template <> struct gn_vec<2> { typedef uint2 type; };
template <> struct gn_vec<4> { typedef uint4 type; };

template <int NPT, int V>
__global__ __launch_bounds__(1024) void gn_group_kernel(GN_HOT_PARAMS, const GNArgs p) {
    typedef typename gn_vec<V>::type vec_t;
    __shared__ float red[16][2];
    __shared__ float s_stat[2];
    const int t = threadIdx.x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int b = slot >> 2, g = xcd * 4 + (slot & 3);
    const int cpg = hot_C / 32;
    // (ppp = 1024 / upp: pixels per pass of the workgroup)
    const int pl = udiv_magic(t, upp, mg_upp), u = t - pl * upp;   // my pixel lane, my unit inside the group's run
    const bool active = pl < ppp;
    const int c = g * cpg + u * 2 * V;          // first of my 2V channels
    const size_t row0 = (size_t)b * hot_hw;
    const bool first = c < hot_c0;
    const int sstride = first ? hot_c0 : hot_c1;
    const bf16_t* src = first ? hot_x0 + row0 * hot_c0 + c : hot_x1 + row0 * hot_c1 + (c - hot_c0);

    union { vec_t v; uint32_t w[V]; } x[NPT];
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const int px = pl + k * ppp;
        if (active && px < hot_hw) x[k].v = *reinterpret_cast<const vec_t*>(src + (size_t)px * sstride);
        else {
#pragma unroll
            for (int e = 0; e < V; ++e) x[k].w[e] = 0u;
        }
    }
    float gm[2 * V], bt[2 * V];
#pragma unroll
    for (int e = 0; e < 2 * V; ++e) { gm[e] = p.gamma[c + e]; bt[e] = p.beta[c + e]; }

    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int k = 0; k < NPT; ++k)
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float lo = bf_lo(x[k].w[e]), hi = bf_hi(x[k].w[e]);
            s += lo + hi;
            ss += lo * lo + hi * hi;
        }
    s = wave_sum(s);
    ss = wave_sum(ss);
    if ((t & 63) == 0) { red[t >> 6][0] = s; red[t >> 6][1] = ss; }
    __syncthreads();
    if (t == 0) {
        float a = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) { a += red[w][0]; q += red[w][1]; }
        const float cnt = (float)p.hw * (float)cpg;
        const float mean = a / cnt;
        const float var = fmaxf(q / cnt - mean * mean, 0.f);
        const float rstd = rsqrtf(var + p.eps);
        s_stat[0] = mean; s_stat[1] = rstd;
        p.stats[((size_t)b * 32 + g) * 2 + 0] = mean;
        p.stats[((size_t)b * 32 + g) * 2 + 1] = rstd;
    }
    __syncthreads();
    if (!active) return;
    const float mean = s_stat[0], rstd = s_stat[1];
#pragma unroll
    for (int e = 0; e < 2 * V; ++e) { gm[e] *= rstd; bt[e] -= mean * gm[e]; }
    bf16_t* dst = p.out + row0 * p.C + c;
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const int px = pl + k * ppp;
        if (px < p.hw) {
            union { vec_t v; uint32_t w[V]; } o;
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float lo = bf_lo(x[k].w[e]) * gm[2 * e] + bt[2 * e];
                float hi = bf_hi(x[k].w[e]) * gm[2 * e + 1] + bt[2 * e + 1];
                if (p.silu) { lo = silu_f(lo); hi = silu_f(hi); }
                o.w[e] = pack_bf2(lo, hi);
            }
            *reinterpret_cast<vec_t*>(dst + (size_t)px * p.C) = o.v;
        }
    }
}

// ---- single-launch GroupNorm over a CLUSTER of workgroups per (sample, group) -------------------------------------------
// The 64x64 level (and the VAE's 64x64 / 128x128 stages) used to take two launches — partial moments, then a second read to
// normalise — because ONE workgroup per (sample, group) leaves 3/4 of the chip idle there (64 slabs of 80-250 KB at batch 1).
// Here P = 2 / 4 / 8 workgroups share a slab by pixel range: each reads its part ONCE into registers, reduces it, publishes
// its partial moments, collects the other parts', and normalises from registers: one read, one write, one launch, and
// 32 x batch x P workgroups.  P depends on the sample's size only (the parts are summed in part order, so P is part of the
// arithmetic: a sample's bits must not depend on the batch it runs in).
//
// The exchange (MI355X_MICROARCH.md, hand-off price list: data-tagged granules): every partial moment travels as ONE naturally
// aligned 8-byte {value, epoch} word written by one agent-scope store and polled with agent-scope loads — the tag arrives
// with the data, so no store -> flag ordering is needed.  The epoch comes from a per-(P, sample, group) ticket counter in the
// caller's `sync` block (zero-initialised, used by nothing else, one block per stream): every launch adds exactly P to it, so
// ticket / P + 1 is the same number in the P workgroups of a launch and differs from every earlier launch's.
// Progress: the parts of a group are consecutive workgroup ids, so under in-order dispatch at most ONE group is ever partly
// resident and every other resident group completes; should the dispatch ever be out of order the poll is BOUNDED (it gives
// up, raises the block's error word and the launch finishes with wrong numbers instead of hanging the GPU).
constexpr int GN_SYNC_WORDS_PER_SLOT = 64;   // [0] ticket, [8] error, [16..31] sum granules, [32..47] sum-of-squares granules
// A sample's share of the sync block (MSD_GN_SYNC_WORDS_PER_SAMPLE): 3 x 32 slots of the per-(sample, group) cluster form, then the
// region of the row-major form (gn_rows_kernel): one header slot (ticket at [0]) and up to 64 parts x 64 granules of 8 bytes
constexpr int GN_SYNC_GROUP_WORDS = 3 * 32 * GN_SYNC_WORDS_PER_SLOT;
constexpr int GN_ROWS_MAX_PARTS = 64;
constexpr int GN_SYNC_SAMPLE_WORDS = 16384;
static_assert(GN_SYNC_GROUP_WORDS + GN_SYNC_WORDS_PER_SLOT + GN_ROWS_MAX_PARTS * 64 * 2 <= GN_SYNC_SAMPLE_WORDS, "sync layout");
constexpr int GN_POLL_LIMIT = 1 << 18;   // default bound of the exchange poll (set_option "gn_poll_limit": tests shorten it)

template <int NPT, int V>
__global__ __launch_bounds__(1024) void gn_cluster_kernel(GN_HOT_PARAMS, int pshift, int ppart, int xsh, const GNArgs p,
                                                          uint32_t* sync_region, int poll_limit) {
    typedef typename gn_vec<V>::type vec_t;
    __shared__ float red[16][2];
    __shared__ float s_stat[2];
    const int t = threadIdx.x;
    const int P = 1 << pshift;
    // xsh = 0: the parts of a (sample, group) are consecutive workgroup ids.  xsh = 1 + log2(batch P / 8) (power of two): XCD x (=
    // blockIdx.x & 7 under round-robin placement) takes the pixel ranges [x batch P / 8, (x + 1) batch P / 8) of the batch's row space for
    // every group — the rows the conv / dense launches in front of and behind this one give that XCD (xcd_remap, n fastest), so what they
    // stored and what they will read stays in its L2 (tools/launch_floor.py: another XCD's fresh stores arrive at half the rate).  The parts
    // of a group, for all samples, then lie within batch P consecutive ids (host: <= 128): the progress argument below holds with that
    // window.  Placement only: the same parts summed in the same order, the same bits.
    int part, slot;   // slot = b * 32 + g
    if (xsh) {
        const int sh = xsh - 1, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int gp = (xcd << sh) + (j & ((1 << sh) - 1));   // global part index b P + part
        part = gp & (P - 1);
        slot = (gp >> pshift) * 32 + (j >> sh);
    } else {
        part = blockIdx.x & (P - 1);
        slot = blockIdx.x >> pshift;
    }
    const int b = slot >> 5, g = slot & 31;
    const int cpg = hot_C / 32;
    const int pl = udiv_magic(t, upp, mg_upp), u = t - pl * upp;   // my pixel lane, my unit inside the group's run
    const bool active = pl < ppp;
    const int px0 = part * ppart, px1 = min(hot_hw, px0 + ppart);
    const int c = g * cpg + u * 2 * V;          // first of my 2V channels
    const size_t row0 = (size_t)b * hot_hw;
    const bool first = c < hot_c0;
    const int sstride = first ? hot_c0 : hot_c1;
    const bf16_t* src = first ? hot_x0 + row0 * hot_c0 + c : hot_x1 + row0 * hot_c1 + (c - hot_c0);

    // (slot-major, one 64-word block per P: where a block lies does not depend on the batch of the launch, so launches
    //  of different batch sizes that share a sync block still keep one counter per (sample, group, P))
    uint32_t* blk = sync_region + (size_t)b * GN_SYNC_SAMPLE_WORDS + ((size_t)g * 3 + (pshift - 1)) * GN_SYNC_WORDS_PER_SLOT;
    // The ticket's round trip to the memory side (~1.5-2 us) sat between the reduction and the publication.  With 8- and 16-byte units (the
    // slabs whose loads take at least as long) it is drawn FIRST and runs beside the loads: 64x64 x 640 10.1 -> 8.2 us, the VAE's 64x64 x 512
    // 9.2 -> 7.4.  Not with 4-byte units: vector-memory results return in order, so wave 0 then stands at its first use of the loaded data
    // until the ticket is back (64x64 x 320: 7.2 -> 7.6 us); drawn behind the loads it gained nothing either.
    uint32_t ticket = 0;
    constexpr bool early = V >= 2;   // (compile-time: behind a test of a kernel ARGUMENT the atomic waits for the argument load and the head start is gone)
    if (early && t == 0) ticket = __hip_atomic_fetch_add(blk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::: "memory");   // (the loads below stay below: hipcc is free to move them across a relaxed atomic, and did)

    union { vec_t v; uint32_t w[V]; } x[NPT];
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const int px = px0 + pl + k * ppp;
        if (active && px < px1) x[k].v = *reinterpret_cast<const vec_t*>(src + (size_t)px * sstride);
        else {
#pragma unroll
            for (int e = 0; e < V; ++e) x[k].w[e] = 0u;
        }
    }
    float gm[2 * V], bt[2 * V];
#pragma unroll
    for (int e = 0; e < 2 * V; ++e) { gm[e] = p.gamma[c + e]; bt[e] = p.beta[c + e]; }

    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int k = 0; k < NPT; ++k)
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float lo = bf_lo(x[k].w[e]), hi = bf_hi(x[k].w[e]);
            s += lo + hi;
            ss += lo * lo + hi * hi;
        }
    s = wave_sum(s);
    ss = wave_sum(ss);
    if ((t & 63) == 0) { red[t >> 6][0] = s; red[t >> 6][1] = ss; }
    __syncthreads();
    if (t < 64) {   // wave 0: publish this part's moments, collect all P parts (lane j polls part j), sum them in part order
        unsigned long long* ga = reinterpret_cast<unsigned long long*>(blk + 16);
        unsigned long long* gq = reinterpret_cast<unsigned long long*>(blk + 32);
        uint32_t epoch = 0;
        if (t == 0) {
            float a = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < 16; ++w) { a += red[w][0]; q += red[w][1]; }
            if constexpr (!early) ticket = __hip_atomic_fetch_add(blk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            epoch = (ticket >> pshift) + 1u;
            __hip_atomic_store(ga + part, ((unsigned long long)epoch << 32) | __float_as_uint(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(gq + part, ((unsigned long long)epoch << 32) | __float_as_uint(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        epoch = __shfl(epoch, 0);
        unsigned long long va = 0, vq = 0;
        bool done = t >= P;
        for (int it = 0; it < poll_limit; ++it) {
            if (!done) {
                va = __hip_atomic_load(ga + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                vq = __hip_atomic_load(gq + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                done = (uint32_t)(va >> 32) == epoch && (uint32_t)(vq >> 32) == epoch;
            }
            if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
            __builtin_amdgcn_s_sleep(2);
        }
        // gave up (never under in-order dispatch): flag the slot AND word [8] of the whole block (= slot 0's error word), which is
        // the ONE word a caller reads per job (minsdtf_amd/engine.py check_gn_sync raises HipExtensionError on it)
        if (__builtin_amdgcn_ballot_w64(!done) != 0 && t == 0) { blk[8] = 1u; sync_region[8] = 1u; }
        float a = 0.f, q = 0.f;
        for (int j = 0; j < P; ++j) {   // fixed order: ((p0 + p1) + p2) + ...
            a += __uint_as_float((uint32_t)__shfl(va, j));
            q += __uint_as_float((uint32_t)__shfl(vq, j));
        }
        if (t == 0) {
            const float cnt = (float)p.hw * (float)cpg;
            const float mean = a / cnt;
            const float var = fmaxf(q / cnt - mean * mean, 0.f);
            const float rstd = rsqrtf(var + p.eps);
            s_stat[0] = mean; s_stat[1] = rstd;
            if (part == 0) {
                p.stats[((size_t)b * 32 + g) * 2 + 0] = mean;
                p.stats[((size_t)b * 32 + g) * 2 + 1] = rstd;
            }
        }
    }
    __syncthreads();
    if (!active) return;
    const float mean = s_stat[0], rstd = s_stat[1];
#pragma unroll
    for (int e = 0; e < 2 * V; ++e) { gm[e] *= rstd; bt[e] -= mean * gm[e]; }
    bf16_t* dst = p.out + row0 * p.C + c;
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const int px = px0 + pl + k * ppp;
        if (px < px1) {
            union { vec_t v; uint32_t w[V]; } o;
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float lo = bf_lo(x[k].w[e]) * gm[2 * e] + bt[2 * e];
                float hi = bf_hi(x[k].w[e]) * gm[2 * e + 1] + bt[2 * e + 1];
                if (p.silu) { lo = silu_f(lo); hi = silu_f(hi); }
                o.w[e] = pack_bf2(lo, hi);
            }
            *reinterpret_cast<vec_t*>(dst + (size_t)px * p.C) = o.v;
        }
    }
}

// ---- single-launch GroupNorm, ROW-MAJOR parts: a workgroup owns a pixel range of a sample with ALL its channels ---------------------
// The per-(sample, group) forms above read a group's 2-20 channels of every pixel: 4-40 bytes out of every 128-byte line, and 32 workgroups
// fetch each line once each (gn_cluster<4,1> at the 64x64 level, batch 4: 23.7 us for 42 MB).  Here a workgroup's part is ONE contiguous
// block of memory, read with 16-byte accesses (thread = one 8-channel chunk of a pixel row, the same chunk for all its pixels); it
// reduces per-channel moments through LDS, folds them into the 32 groups' moments, publishes 64 tagged granules and collects the other
// parts' (the exchange of gn_cluster_kernel, 64 values per part instead of 2), then normalises from registers: one read, one write.
// The parts are summed in part order and P depends on the sample's size only, so a sample's bits do not depend on its batch.
// Q = 1 << qshift workgroups share a part by CHANNELS (Q quarters / halves of the 8-channel chunks = 32 / Q whole groups each): the same
// thread -> (pixel row, channel chunk) map, the same per-thread, per-channel and per-group sums in the same order, the same granules at the
// same places - only who computes which of a part's 64 granules changes, so Q is PLACEMENT (chosen from the launch's size: B x P x Q
// workgroups should cover the chip), not arithmetic.  A workgroup polls only its own groups' granules of the other parts.
template <int NPT>
__global__ __launch_bounds__(1024) void gn_rows_kernel(const bf16_t* hot_x0, const bf16_t* hot_x1, int hot_C, int hot_hw, int hot_c0, int hot_c1,
                                                       int cpr, int rows, uint32_t mg_cpr, int pshift, int ppart, int qshift, const GNArgs p,
                                                       uint32_t* sync_region, int poll_limit) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int t = threadIdx.x, T = cpr * rows;               // cpr: 8-channel chunks per pixel row THIS workgroup owns (C / 8 / Q)
    const int Cq = cpr * 8, Gq = 32 >> qshift;               // its channels, its groups
    float* red = reinterpret_cast<float*>(smem_raw);         // [T][16]: per thread 8 sums, 8 sums of squares
    float* chan = red + (size_t)T * 16;                      // [2][Cq]
    float* parts = chan + 2 * Cq;                            // [P][2 Gq]
    float* tot = parts + ((size_t)(2 * Gq) << pshift);       // [2 Gq]: sums, sums of squares of the whole sample (this workgroup's groups)
    float* stat = tot + 64;                                  // [Gq][2] mean, rstd
    uint32_t* s_epoch = reinterpret_cast<uint32_t*>(stat + 64);
    const int P = 1 << pshift;
    const int qi = blockIdx.x & ((1 << qshift) - 1);
    const int part = (blockIdx.x >> qshift) & (P - 1), b = blockIdx.x >> (pshift + qshift);
    const int row = udiv_magic(t, cpr, mg_cpr), ch = t - row * cpr;
    const bool active = t < T;
    const int px0 = part * ppart, px1 = min(hot_hw, px0 + ppart);
    const int c = qi * Cq + ch * 8;
    const size_t row0 = (size_t)b * hot_hw;
    const bool first = c < hot_c0;
    const int sstride = first ? hot_c0 : hot_c1;
    const bf16_t* src = first ? hot_x0 + row0 * hot_c0 + c : hot_x1 + row0 * hot_c1 + (c - hot_c0);

    union { uint4 v; uint32_t w[4]; } x[NPT];
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const int px = px0 + row + k * rows;
        if (active && px < px1) x[k].v = *reinterpret_cast<const uint4*>(src + (size_t)px * sstride);
        else x[k].v = make_uint4(0u, 0u, 0u, 0u);
    }
    float gm[8], bt[8];
    if (active) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { gm[e] = p.gamma[c + e]; bt[e] = p.beta[c + e]; }
    }
    uint32_t ticket = 0;
    // (every launch adds exactly 256 to the sample's ticket - 256 / (P Q) per workgroup - whatever its P and Q: ticket / 256 + 1 is the same number in
    //  all workgroups of a launch and differs from every earlier launch's, of any P, that wrote these granules; 2^32 is a multiple of 256, so that
    //  holds across the counter's wrap too)
    if (t == 0) ticket = __hip_atomic_fetch_add(sync_region + (size_t)b * GN_SYNC_SAMPLE_WORDS + GN_SYNC_GROUP_WORDS, 256u >> (pshift + qshift), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (active) {
        float sm[8], sq[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { sm[e] = 0.f; sq[e] = 0.f; }
#pragma unroll
        for (int k = 0; k < NPT; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lo = bf_lo(x[k].w[e]), hi = bf_hi(x[k].w[e]);
                sm[2 * e] += lo; sm[2 * e + 1] += hi;
                sq[2 * e] += lo * lo; sq[2 * e + 1] += hi * hi;
            }
        float4* rp = reinterpret_cast<float4*>(red + (size_t)t * 16);
        rp[0] = make_float4(sm[0], sm[1], sm[2], sm[3]); rp[1] = make_float4(sm[4], sm[5], sm[6], sm[7]);
        rp[2] = make_float4(sq[0], sq[1], sq[2], sq[3]); rp[3] = make_float4(sq[4], sq[5], sq[6], sq[7]);
    }
    if (t == 0) *s_epoch = (ticket >> 8) + 1u;
    __syncthreads();
    // per-channel moments of the part: channel cc, moment m <- sum over the pixel rows (fixed order)
    for (int j = t; j < 2 * Cq; j += blockDim.x) {
        const int m = j >= Cq ? 1 : 0, cc = j - m * Cq;
        const float* q = red + (size_t)(cc >> 3) * 16 + (cc & 7) + 8 * m;
        float a = 0.f;
        for (int r = 0; r < rows; ++r) a += q[(size_t)r * cpr * 16];
        chan[j] = a;
    }
    __syncthreads();
    const uint32_t epoch = *s_epoch;
    unsigned long long* gran = reinterpret_cast<unsigned long long*>(sync_region + (size_t)b * GN_SYNC_SAMPLE_WORDS + GN_SYNC_GROUP_WORDS + GN_SYNC_WORDS_PER_SLOT);
    const int cpg = hot_C >> 5, g0 = qi * Gq;                // this workgroup's groups: g0 .. g0 + Gq - 1
    if (t < 2 * Gq) {   // group moments of the part: published as {value, epoch} granules (granule g + 32 m of the part's 64, as with Q = 1)
        const int gl = t & (Gq - 1), m = t >> (5 - qshift);
        const float* q = chan + m * Cq + gl * cpg;
        float a = 0.f;
        for (int e = 0; e < cpg; ++e) a += q[e];
        __hip_atomic_store(gran + (size_t)part * 64 + g0 + gl + 32 * m, ((unsigned long long)epoch << 32) | __float_as_uint(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // collect the P x 2 Gq granules of this workgroup's groups (thread i: granules i, i + blockDim, .. - at most 4, polled TOGETHER: a poll is a
    // round trip to the memory side); bounded, as in gn_cluster_kernel
    {
        const int ng = (2 * Gq) << pshift, nt = (int)blockDim.x;
        unsigned long long v[4] = {0, 0, 0, 0};
        bool done[4];
        const unsigned long long* at[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = t + nt * k;                           // local granule: part' = i / (2 Gq), then (m, gl)
            done[k] = i >= ng;
            const int pp = i >> (6 - qshift), rr = i & (2 * Gq - 1);
            at[k] = gran + (size_t)(done[k] ? 0 : pp) * 64 + g0 + (rr & (Gq - 1)) + 32 * (rr >> (5 - qshift));
        }
        for (int it = 0; it < poll_limit; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (!done[k]) v[k] = __hip_atomic_load(at[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (!done[k]) done[k] = (uint32_t)(v[k] >> 32) == epoch;
            if (done[0] && done[1] && done[2] && done[3]) break;
            __builtin_amdgcn_s_sleep(2);
        }
        if (!(done[0] && done[1] && done[2] && done[3])) { sync_region[(size_t)b * GN_SYNC_SAMPLE_WORDS + GN_SYNC_GROUP_WORDS + 8] = 1u; sync_region[8] = 1u; }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (t + nt * k < ng) parts[t + nt * k] = __uint_as_float((uint32_t)v[k]);
    }
    __syncthreads();
    if (t < 2 * Gq) {   // the sample's moments: parts in part order, ((p0 + p1) + p2) + ...   (t = gl + Gq m, as the parts' rows are laid out)
        float a = 0.f;
        for (int j = 0; j < P; ++j) a += parts[j * 2 * Gq + t];
        tot[t] = a;
    }
    __syncthreads();
    if (t < Gq) {
        const float cnt = (float)p.hw * (float)cpg;
        const float mean = tot[t] / cnt;
        const float var = fmaxf(tot[Gq + t] / cnt - mean * mean, 0.f);
        const float rstd = rsqrtf(var + p.eps);
        stat[2 * t] = mean; stat[2 * t + 1] = rstd;
        if (part == 0) {
            p.stats[((size_t)b * 32 + g0 + t) * 2 + 0] = mean;
            p.stats[((size_t)b * 32 + g0 + t) * 2 + 1] = rstd;
        }
    }
    __syncthreads();
    if (!active) return;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int g = udiv_magic(c + e, cpg, p.mg_cpg) - g0;
        const float mean = stat[2 * g], rstd = stat[2 * g + 1];
        gm[e] *= rstd; bt[e] -= mean * gm[e];
    }
    bf16_t* dst = p.out + row0 * p.C + c;
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const int px = px0 + row + k * rows;
        if (px < px1) {
            union { uint4 v; uint32_t w[4]; } o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float lo = bf_lo(x[k].w[e]) * gm[2 * e] + bt[2 * e];
                float hi = bf_hi(x[k].w[e]) * gm[2 * e + 1] + bt[2 * e + 1];
                if (p.silu) { lo = silu_f(lo); hi = silu_f(hi); }
                o.w[e] = pack_bf2(lo, hi);
            }
            *reinterpret_cast<uint4*>(dst + (size_t)px * p.C) = o.v;
        }
    }
}

static int g_gn_poll_limit = GN_POLL_LIMIT;
void msd_set_gn_poll_limit(int v) { g_gn_poll_limit = v; }

static int g_gn_xmap = 1;   // cluster GroupNorm: 1 = XCD x takes the pixel ranges the convs around it give that XCD [default], 0 = a group's parts on consecutive ids (A/B runs; same bits)
void msd_set_gn_xmap(int v) { g_gn_xmap = v ? 1 : 0; }

template <int V>
static void gn_cluster_launch(const GNArgs& a, int upp, int npt, dim3 grid, int pshift, int ppart, uint32_t* region, hipStream_t stream) {
    const dim3 block(1024);
    const int ppp = 1024 / upp;
    const uint32_t mg = udiv_magic_of(upp);
    int xsh = 0;
    {
        const int bp = (int)(grid.x >> 5);   // batch x P (grid = batch x 32 groups x P parts)
        if (g_gn_xmap && bp >= 8 && bp <= 128 && (bp & (bp - 1)) == 0 && (int)grid.x == bp * 32) {
            int l = 0;
            while ((8 << l) < bp) ++l;
            xsh = l + 1;
        }
    }
    if (npt <= 2) hipLaunchKernelGGL((gn_cluster_kernel<2, V>), grid, block, 0, stream, GN_HOT_ARGS(a), pshift, ppart, xsh, a, region, g_gn_poll_limit);
    else if (npt <= 4) hipLaunchKernelGGL((gn_cluster_kernel<4, V>), grid, block, 0, stream, GN_HOT_ARGS(a), pshift, ppart, xsh, a, region, g_gn_poll_limit);
    else if (npt <= 8) hipLaunchKernelGGL((gn_cluster_kernel<8, V>), grid, block, 0, stream, GN_HOT_ARGS(a), pshift, ppart, xsh, a, region, g_gn_poll_limit);
    else if constexpr (V < 4) hipLaunchKernelGGL((gn_cluster_kernel<16, V>), grid, block, 0, stream, GN_HOT_ARGS(a), pshift, ppart, xsh, a, region, g_gn_poll_limit);
}

template <int V>
static void gn_group_launch(const GNArgs& a, int upp, int npt, dim3 grid, hipStream_t stream) {
    const dim3 block(1024);
    const int ppp = 1024 / upp;
    const uint32_t mg = udiv_magic_of(upp);
    if (npt <= 2) hipLaunchKernelGGL((gn_group_kernel<2, V>), grid, block, 0, stream, GN_HOT_ARGS(a), a);
    else if (npt <= 4) hipLaunchKernelGGL((gn_group_kernel<4, V>), grid, block, 0, stream, GN_HOT_ARGS(a), a);
    else if (npt <= 8) hipLaunchKernelGGL((gn_group_kernel<8, V>), grid, block, 0, stream, GN_HOT_ARGS(a), a);
    else if constexpr (V < 4) hipLaunchKernelGGL((gn_group_kernel<16, V>), grid, block, 0, stream, GN_HOT_ARGS(a), a);
}

static int g_gn_impl = 1;  // 1 = single-launch per-group kernel where the group slab fits, 0 = always stats/finalize/apply
static int g_gn_wide = 1;  // 1 = 1024-thread stats / apply workgroups for mid-sized tensors, 0 = always 256 (A/B runs)
static int g_gn_cluster = 256;    // pixels per part the cluster kernel aims at: P = largest power of two <= pixels / this, at most 8
                                  // (P = 1: the one-workgroup kernel); 0 = never the cluster kernel (A/B runs)
void msd_set_gn_cluster(int v) { g_gn_cluster = v; }
static int g_gn_rows = 9216;      // row-major parts (gn_rows_kernel) for samples of at least this many PIXELS; 0 = never (A/B runs).
                                  // 4096 adds the 64x64 level: same-box loop -1.7 % at batch 4, -0.7 % at batch 2, +0.6 % at batch 1 (2 samples
                                  // x 32 parts = 64 workgroups pull 80 KB each: a quarter of the chip's requests in flight) - a serving choice
void msd_set_gn_rows(int v) { g_gn_rows = v; }
static int g_gn_rows_q = -1;      // row-major form: log2 of the workgroups that share a part by channels; -1 = from the launch's size (A/B runs; same bits)
void msd_set_gn_rows_q(int v) { g_gn_rows_q = v; }
void msd_set_gn_impl(int v) { g_gn_impl = v; }
void msd_set_gn_wide(int v) { g_gn_wide = v; }

extern "C" int msd_group_norm(const MsdGroupNorm* q, msd_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!q || !q->x0 || !q->gamma || !q->beta || !q->stats || !q->partials || !q->out)
        MSD_FAIL(MSD_E_ARG, "group_norm: null pointer");
    if (q->batch <= 0 || q->hw <= 0 || q->c0 <= 0 || q->c1 < 0 || (q->c1 > 0 && !q->x1))
        MSD_FAIL(MSD_E_ARG, "group_norm: bad dims");
    const int C = q->c0 + q->c1;
    if ((q->c0 % 8) || (q->c1 % 8) || (C % 32) || C > 4096)
        MSD_FAIL(MSD_E_UNSUPPORTED, "group_norm: c0,c1 must be multiples of 8, C of 32, C<=4096 (c0=%d c1=%d)", q->c0, q->c1);
    if (!msd_aligned16(q->x0) || !msd_aligned16(q->x1) || !msd_aligned16(q->out))
        MSD_FAIL(MSD_E_ALIGN, "group_norm: pointers must be 16-byte aligned");
    GNArgs a;
    a.x0 = (const bf16_t*)q->x0; a.x1 = (const bf16_t*)q->x1; a.gamma = q->gamma; a.beta = q->beta;
    a.stats = q->stats; a.partials = q->partials; a.out = (bf16_t*)q->out;
    a.batch = q->batch; a.hw = q->hw; a.c0 = q->c0; a.c1 = q->c1; a.C = C; a.cv = C / 8;
    // mid-sized tensors (the UNet's 64x64 level at batch 1-8): 1024-thread workgroups, see gn_stats_kernel
    const bool wide = g_gn_wide && (long long)q->hw * C >= (1ll << 20) && (long long)q->hw * C <= (4ll << 20) && C <= 2048;
    const int NT = wide ? 1024 : 256;
    a.tpp = a.cv < 256 ? a.cv : 256;
    a.pl = NT / a.tpp;
    a.mg_tpp = udiv_magic_of(a.tpp);
    a.mg_cpg = udiv_magic_of(C / 32);
    a.silu = q->silu ? 1 : 0; a.eps = q->eps;
    // row-major parts: samples of >= g_gn_rows pixels (default: 96x96 at 768^2 and up) whose part fits in registers;
    // smaller samples are latency chains, where this form's longer chain (LDS reductions, 64 granules per part) loses: tools/gn_bench.py
    if (g_gn_rows > 0 && g_gn_cluster && q->sync && q->hw >= g_gn_rows && a.cv <= 512) {
        const int cpr = a.cv, rows = 1024 / cpr;
        int pshift = 2;
        while (pshift <= 6 && (((q->hw + (1 << pshift) - 1) >> pshift) + rows - 1) / rows > 6) ++pshift;
        if (pshift <= 6) {
            const int P = 1 << pshift;
            const int ppart = (q->hw + P - 1) / P;
            const int npt = (ppart + rows - 1) / rows;
            const long long need = (long long)q->batch * GN_SYNC_SAMPLE_WORDS;
            if (q->sync_words < need)
                MSD_FAIL(MSD_E_WORKSPACE, "group_norm: sync block too small (%lld < %lld words)", (long long)q->sync_words, need);
            if (((uintptr_t)q->sync) & 7u) MSD_FAIL(MSD_E_ALIGN, "group_norm: sync must be 8-byte aligned");
            // Q workgroups per part, by channels (placement only, see the kernel): enough of them that batch x P x Q covers the chip
            int qshift = 0;
            if (g_gn_rows_q < 0) {
                while (qshift < 2 && ((long long)q->batch << (pshift + qshift)) < 256 && (cpr % (2 << qshift)) == 0 && pshift + qshift < 8) ++qshift;
            } else {
                while (qshift < g_gn_rows_q && qshift < 2 && (cpr % (2 << qshift)) == 0 && pshift + qshift < 8) ++qshift;
            }
            const int cprq = cpr >> qshift;
            const int ngran = (64 >> qshift) << pshift;   // granules a workgroup collects: at most four per thread
            int nthreads = (cprq * rows + 63) / 64 * 64;
            if (nthreads * 4 < ngran) nthreads = ((ngran + 3) / 4 + 63) / 64 * 64;
            const size_t lds = ((size_t)cprq * rows * 16 + 2 * (size_t)cprq * 8 + ((size_t)(64 >> qshift) << pshift) + 64 + 64 + 4) * sizeof(float);
            static bool attr_done = false;
            if (!attr_done) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_rows_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
                if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_rows_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
                if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_rows_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
                if (e != hipSuccess) MSD_FAIL((int)e, "hipFuncSetAttribute(gn_rows): %s", hipGetErrorString(e));
                attr_done = true;
            }
            const dim3 grid((unsigned)q->batch << (pshift + qshift));
            const uint32_t mgc = udiv_magic_of(cprq);
#define GN_ROWS_LAUNCH(N_) hipLaunchKernelGGL((gn_rows_kernel<N_>), grid, dim3(nthreads), lds, stream, a.x0, a.x1, a.C, a.hw, a.c0, a.c1, cprq, rows, mgc, pshift, ppart, qshift, a, q->sync, g_gn_poll_limit)
            if (npt <= 2) GN_ROWS_LAUNCH(2);
            else if (npt <= 4) GN_ROWS_LAUNCH(4);
            else GN_ROWS_LAUNCH(6);
#undef GN_ROWS_LAUNCH
            MSD_CHECK_LAUNCH();
            return MSD_OK;
        }
    }
    {
        // single-launch path: V = words per thread-unit (widest that divides the group's run), at most
        // 24 units per thread (register budget of a 1024-thread workgroup: 128 VGPRs)
        const int cpg = C / 32;
        if (g_gn_impl == 1 && (cpg % 2) == 0 && cpg <= 160) {
            const int dpp = cpg / 2;
            const int V = (dpp % 4 == 0) ? 4 : (dpp % 2 == 0) ? 2 : 1;
            const int upp = dpp / V, ppp = 1024 / upp;
            // cluster form: P parts per (sample, group), from the SAMPLE's size only
            int pshift = 0;
            if (g_gn_cluster && q->sync) {
                int P = q->hw / g_gn_cluster;
                if (P > 8) P = 8;
                while ((2 << pshift) <= P) ++pshift;
            }
            if (pshift > 0) {
                const int P = 1 << pshift;
                const int ppart = (q->hw + P - 1) / P;
                const int nptc = (ppart + ppp - 1) / ppp;
                const long long need = (long long)q->batch * GN_SYNC_SAMPLE_WORDS;
                if (nptc <= (V == 4 ? 8 : 16) && (long long)q->batch * 32 * P < (1ll << 30)) {
                    if (q->sync_words < need)
                        MSD_FAIL(MSD_E_WORKSPACE, "group_norm: sync block too small (%lld < %lld words)", (long long)q->sync_words, need);
                    if (((uintptr_t)q->sync) & 7u) MSD_FAIL(MSD_E_ALIGN, "group_norm: sync must be 8-byte aligned");
                    uint32_t* region = q->sync;
                    const dim3 grid(32 * q->batch * P);
                    if (V == 4) gn_cluster_launch<4>(a, upp, nptc, grid, pshift, ppart, region, stream);
                    else if (V == 2) gn_cluster_launch<2>(a, upp, nptc, grid, pshift, ppart, region, stream);
                    else gn_cluster_launch<1>(a, upp, nptc, grid, pshift, ppart, region, stream);
                    MSD_CHECK_LAUNCH();
                    return MSD_OK;
                }
            }
            const int npt = (q->hw + ppp - 1) / ppp;
            // <= 32 data registers per thread.  Bigger slabs (the 64x64 level: 21+ units per thread) measured
            // SLOWER this way than stats + apply (22 vs 20 us at C=320, 33 vs 27 us at C=640): with 64
            // workgroups the scattered 4/8-byte accesses, not the launch count, set the time.
            if (npt <= (V == 4 ? 8 : 16)) {
                const dim3 grid(32 * q->batch);
                if (V == 4) gn_group_launch<4>(a, upp, npt, grid, stream);
                else if (V == 2) gn_group_launch<2>(a, upp, npt, grid, stream);
                else gn_group_launch<1>(a, upp, npt, grid, stream);
                MSD_CHECK_LAUNCH();
                return MSD_OK;
            }
        }
    }
    const int vpt = (a.cv + 255) / 256;  // 1 or 2
    // pixels per block.  Small tensors (the whole UNet) are launch-bound: at most 64 chunks per
    // sample, which the apply kernel reduces itself (2 launches).  Large tensors (VAE at 256^2 /
    // 512^2) are byte-bound: ~1024 workgroups chip-wide and a separate ordered finalize (3 launches).
    const bool small = (long long)q->hw * C <= (4ll << 20);
    // (wide: 128 chunks per sample — 256 workgroups at batch 2, one per CU; 64 leave half the chip idle)
    // (chunks per SAMPLE, whatever the batch: the partial moments are summed chunk by chunk, so the chunking is part of the
    //  arithmetic, and a sample's result must not depend on the batch it is computed in)
    const long long target_blocks = small ? (wide ? 128 : 64) : 1024;
    int ppb = (int)((q->hw + target_blocks - 1) / target_blocks);
    const int min_ppb = wide ? a.pl : a.pl * 4;
    if (ppb < min_ppb) ppb = min_ppb;
    if (ppb > q->hw) ppb = q->hw;
    a.ppb = ppb;
    const int nchunks = (q->hw + ppb - 1) / ppb;
    const bool fused = nchunks <= (wide ? 128 : 64);
    if (q->partials_floats < (long long)q->batch * nchunks * 64)
        MSD_FAIL(MSD_E_WORKSPACE, "group_norm: partials scratch too small (%lld < %lld floats)",
                 (long long)q->partials_floats, (long long)q->batch * nchunks * 64);
    dim3 grid(nchunks, q->batch);
    a.nchunks_stats = nchunks;
    const size_t lds = ((size_t)a.pl * C * 2 + (size_t)C * 2) * sizeof(float);
    if (wide) {
        static bool attr_done = false;
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_stats_kernel<1, 1024>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) MSD_FAIL((int)e, "hipFuncSetAttribute(gn_stats): %s", hipGetErrorString(e));
            attr_done = true;
        }
        if (vpt != 1 || lds > 160 * 1024) MSD_FAIL(MSD_E_UNSUPPORTED, "group_norm: wide path needs C <= 2048");
        if (fused) {
            // The statistics pass runs on 4x coarser chunks than the apply pass: every apply workgroup re-reduces ALL
            // the partial moments of its sample (fixed order), and with one partial set per apply chunk that is as
            // many bytes as the tensor itself (128 chunks x 256 B per workgroup x 256 workgroups = 8 MB).
            GNArgs as = a;
            as.ppb = ppb * 4 < q->hw ? ppb * 4 : q->hw;
            const int nchunks_s = (q->hw + as.ppb - 1) / as.ppb;
            as.nchunks_stats = nchunks_s;
            hipLaunchKernelGGL((gn_stats_kernel<1, 1024>), dim3(nchunks_s, q->batch), dim3(1024), lds, stream, as);
            MSD_CHECK_LAUNCH();
            hipLaunchKernelGGL((gn_apply_kernel<1, true, 1024>), grid, dim3(1024), 0, stream, a, nchunks_s);
            MSD_CHECK_LAUNCH();
            return MSD_OK;
        }
        hipLaunchKernelGGL((gn_stats_kernel<1, 1024>), grid, dim3(1024), lds, stream, a);
        MSD_CHECK_LAUNCH();
        {
            hipLaunchKernelGGL(gn_finalize_kernel, dim3(q->batch), dim3(1024), 0, stream, a, nchunks);
            MSD_CHECK_LAUNCH();
            hipLaunchKernelGGL((gn_apply_kernel<1, false, 1024>), grid, dim3(1024), 0, stream, a, nchunks);
        }
        MSD_CHECK_LAUNCH();
        return MSD_OK;
    }
    if (vpt == 1) hipLaunchKernelGGL((gn_stats_kernel<1, 256>), grid, dim3(256), lds, stream, a);
    else hipLaunchKernelGGL((gn_stats_kernel<2, 256>), grid, dim3(256), lds, stream, a);
    MSD_CHECK_LAUNCH();
    if (!fused) {
        hipLaunchKernelGGL(gn_finalize_kernel, dim3(q->batch), dim3(1024), 0, stream, a, nchunks);
        MSD_CHECK_LAUNCH();
    }
    if (vpt == 1) {
        if (fused) hipLaunchKernelGGL((gn_apply_kernel<1, true, 256>), grid, dim3(256), 0, stream, a, nchunks);
        else hipLaunchKernelGGL((gn_apply_kernel<1, false, 256>), grid, dim3(256), 0, stream, a, nchunks);
    } else {
        if (fused) hipLaunchKernelGGL((gn_apply_kernel<2, true, 256>), grid, dim3(256), 0, stream, a, nchunks);
        else hipLaunchKernelGGL((gn_apply_kernel<2, false, 256>), grid, dim3(256), 0, stream, a, nchunks);
    }
    MSD_CHECK_LAUNCH();
    return MSD_OK;
}

// ---- LayerNorm: one wave per row, the row lives in registers between the two reductions ------
template <int NV>
__global__ __launch_bounds__(256) void layer_norm_kernel(const bf16_t* x, const float* gamma, const float* beta,
                                                         bf16_t* out, int rows, int c, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int ncv = c >> 3;
    float f[NV][8];
    float sum = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int cv = lane + 64 * v;
        if (cv < ncv) {
            const uint4 raw = *reinterpret_cast<const uint4*>(x + (size_t)row * c + cv * 8);
            unpack8(raw, f[v]);
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += f[v][e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[v][e] = 0.f;
        }
    }
    const float mean = wave_sum(sum) / (float)c;
    float sq = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int cv = lane + 64 * v;
        if (cv < ncv) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = f[v][e] - mean; sq += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)c + eps);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int cv = lane + 64 * v;
        if (cv < ncv) {
            const float4 g0 = *reinterpret_cast<const float4*>(gamma + cv * 8);
            const float4 g1 = *reinterpret_cast<const float4*>(gamma + cv * 8 + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(beta + cv * 8);
            const float4 b1 = *reinterpret_cast<const float4*>(beta + cv * 8 + 4);
            float o[8];
            o[0] = (f[v][0] - mean) * rstd * g0.x + b0.x; o[1] = (f[v][1] - mean) * rstd * g0.y + b0.y;
            o[2] = (f[v][2] - mean) * rstd * g0.z + b0.z; o[3] = (f[v][3] - mean) * rstd * g0.w + b0.w;
            o[4] = (f[v][4] - mean) * rstd * g1.x + b1.x; o[5] = (f[v][5] - mean) * rstd * g1.y + b1.y;
            o[6] = (f[v][6] - mean) * rstd * g1.z + b1.z; o[7] = (f[v][7] - mean) * rstd * g1.w + b1.w;
            *reinterpret_cast<uint4*>(out + (size_t)row * c + cv * 8) = pack8(o);
        }
    }
}

extern "C" int msd_layer_norm(const void* x, const float* gamma, const float* beta, void* out, int32_t rows, int32_t c,
                              float eps, msd_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !gamma || !beta || !out) MSD_FAIL(MSD_E_ARG, "layer_norm: null pointer");
    if (rows <= 0 || c <= 0 || (c % 8) || c > 2048) MSD_FAIL(MSD_E_UNSUPPORTED, "layer_norm: c=%d must be a multiple of 8, <=2048", c);
    if (!msd_aligned16(x) || !msd_aligned16(out) || !msd_aligned16(gamma) || !msd_aligned16(beta))
        MSD_FAIL(MSD_E_ALIGN, "layer_norm: pointers must be 16-byte aligned");
    const int nv = (c / 8 + 63) / 64;
    dim3 grid((rows + 3) / 4);
    const bf16_t* xi = (const bf16_t*)x;
    bf16_t* o = (bf16_t*)out;
    switch (nv) {
        case 1: hipLaunchKernelGGL(layer_norm_kernel<1>, grid, dim3(256), 0, stream, xi, gamma, beta, o, rows, c, eps); break;
        case 2: hipLaunchKernelGGL(layer_norm_kernel<2>, grid, dim3(256), 0, stream, xi, gamma, beta, o, rows, c, eps); break;
        case 3: hipLaunchKernelGGL(layer_norm_kernel<3>, grid, dim3(256), 0, stream, xi, gamma, beta, o, rows, c, eps); break;
        default: hipLaunchKernelGGL(layer_norm_kernel<4>, grid, dim3(256), 0, stream, xi, gamma, beta, o, rows, c, eps); break;
    }
    MSD_CHECK_LAUNCH();
    return MSD_OK;
}

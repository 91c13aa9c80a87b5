// Host-loop arithmetic of the reference moved onto the device, plus small elementwise helpers.
//
// msd_cfg_step fuses stable_diffusion.py:458-461 (classifier-free guidance and
// rescale_noise_cfg, :304-315) with Scheduler.step's deterministic branch
// (scheduler.py:272-285,308-312) so the latent never leaves HBM between UNet calls.  The
// reference does this in numpy: float32 for CFG / std, float64 for the sampler coefficients;
// here everything is fp32; the four per-sample moments are taken about a shift inside the data (fp32 per wave, fp64 across waves).
#include "common.h"

struct CfgArgs {
    const float* eps; float* latent; const float* coef; int32_t* step_ptr;
    int advance_in_kernel;   // the workgroup that finishes last advances *step_ptr itself (ticket at step_ptr[1])
    int batch, n, num_steps;
    float guidance, rescale;
    const float* ip_init; const float* ip_noise; const float* ip_mask;
    const float* step_noise; const float* noise_coef;
};

// One 1024-thread workgroup per sample.  eps (uncond / cond) is read ONCE into registers (up to NPT elements per
// thread; larger latents take the re-reading loop) and the four moments go through one block reduction with the
// same summation order as four separate ones (wave butterfly, then the 16 wave sums in wave order).
template <int NPT>
__global__ __launch_bounds__(1024) void cfg_step_kernel(const float* hot_eps, float* hot_latent, const float* hot_coef, int32_t* hot_step_ptr, int hot_batch, int hot_n,
                                                       int hot_num_steps, float hot_guidance, const CfgArgs p) {
    // (leading scalars: kernarg preload — the step index -> coefficient loads and the data loads start without the kernarg round trip)
    __shared__ double red[16][4];
    const int b = blockIdx.x, t = threadIdx.x;
    const int step_raw = hot_step_ptr ? *hot_step_ptr : 0;
    int step = step_raw;
    if (step > hot_num_steps - 1) step = hot_num_steps - 1;
    if (step < 0) step = 0;
    const float sr = hot_coef[step * 4 + 0], nr = hot_coef[step * 4 + 1];
    const float ca = hot_coef[step * 4 + 2], cb = hot_coef[step * 4 + 3];   // x' = ca * x0 + cb * eps (+ cz * z)
    const float cz = p.step_noise ? p.noise_coef[step] : 0.0f;
    const float* z = p.step_noise ? p.step_noise + ((size_t)step * hot_batch + b) * hot_n : nullptr;
    float* lat = hot_latent + (size_t)b * hot_n;
    const bool cfg = hot_guidance > 0.0f;
    const float* u = hot_eps + (size_t)b * hot_n;
    const float* c = cfg ? hot_eps + (size_t)(hot_batch + b) * hot_n : u;
    // 16-byte accesses (n = h * w * 4 is a multiple of 4): thread t owns elements 4 (t + 1024 k) .. + 3.  One CU issues every load of
    // its sample: as 4-byte accesses that was 48 wave-instructions per wave x 16 waves through one address unit, most of the
    // launch's 19.8 us at 64x64 (round 5)
    // (KEEP_L: the latent is prefetched with eps only where the registers allow it: 1024 threads = 128 registers per lane; at NPT = 36
    //  three arrays spill, so the latent is read in the update pass instead)
    constexpr bool KEEP_L = NPT <= 16;
    float ru[NPT > 0 ? NPT : 1], rc[NPT > 0 ? NPT : 1], rl[(NPT > 0 && KEEP_L) ? NPT : 1];
    if (NPT > 0) {   // everything this thread touches, issued up front
#pragma unroll
        for (int k = 0; k < NPT / 4; ++k) {
            const int i = min(4 * (t + k * 1024), hot_n - 4);
            const float4 a = *reinterpret_cast<const float4*>(u + i), bq = *reinterpret_cast<const float4*>(c + i);
            ru[4 * k] = a.x; ru[4 * k + 1] = a.y; ru[4 * k + 2] = a.z; ru[4 * k + 3] = a.w;
            rc[4 * k] = bq.x; rc[4 * k + 1] = bq.y; rc[4 * k + 2] = bq.z; rc[4 * k + 3] = bq.w;
            if constexpr (KEEP_L) {
                const float4 l = *reinterpret_cast<const float4*>(lat + i);
                rl[4 * k] = l.x; rl[4 * k + 1] = l.y; rl[4 * k + 2] = l.z; rl[4 * k + 3] = l.w;
            }
        }
    }
    float factor = 1.0f;
    if (cfg && p.rescale > 0.0f) {
        // Moments about a per-sample shift (the first element of each tensor): s = sum(x - x0), q = sum((x - x0)^2) in fp32
        // per thread (<= 36 terms) and per wave (fixed-order DPP / row-swap tree, no LDS round trips), then the 16 wave sums
        // in fp64.  The variance is shift-invariant, and about a shift inside the data's range the fp32 sums lose nothing
        // that matters: (q - s^2/n)/n has no large-number cancellation.  (Was: fp64 per element and six 64-bit
        // ds_bpermute steps per moment: 48 LDS round trips on the step's critical path.)
        const float c0 = c[0], g0 = u[0] + hot_guidance * (c[0] - u[0]);
        float f4[4] = {0.f, 0.f, 0.f, 0.f};   // s_c, q_c, s_g, q_g (shifted)
        auto acc = [&](float cu, float cc) {
            const float gq = cu + hot_guidance * (cc - cu);
            const float dc = cc - c0, dg = gq - g0;
            f4[0] += dc; f4[1] += dc * dc; f4[2] += dg; f4[3] += dg * dg;
        };
        if (NPT > 0) {
#pragma unroll
            for (int k = 0; k < NPT; ++k)
                if (4 * (t + (k >> 2) * 1024) < hot_n) acc(ru[k], rc[k]);   // (a quad is inside the sample or outside it)
        } else {
            for (int i = t; i < hot_n; i += 1024) acc(u[i], c[i]);
        }
        double m4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) f4[q] = wave_sum(f4[q]);
        if ((t & 63) == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) red[t >> 6][q] = (double)f4[q];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double s = 0.0;
            for (int w = 0; w < 16; ++w) s += red[w][q];
            m4[q] = s;
        }
        const double n = (double)hot_n;
        const double mc = m4[0] / n, mg = m4[2] / n;   // (means of the shifted values: the shift drops out of the variance)
        const double vc = fmax(m4[1] / n - mc * mc, 0.0), vg = fmax(m4[3] / n - mg * mg, 0.0);
        const float std_text = (float)sqrt(vc);
        const float std_cfg = (float)sqrt(vg) + 1e-5f;
        factor = p.rescale * (std_text / std_cfg) + (1.0f - p.rescale);
    }
    auto value = [&](int i, float cu, float cc, float l) {
        const float e = cfg ? (cu + hot_guidance * (cc - cu)) * factor : cu;
        const float x0 = (l - nr * e) / sr;
        float x = ca * x0 + cb * e;
        if (z) x += cz * z[i];
        if (p.ip_mask) {   // inpainting: keep the (re-noised) original outside the mask
            const float m = p.ip_mask[i];
            const float org = sr * p.ip_init[i] + nr * p.ip_noise[(size_t)b * hot_n + i];
            x = org * (1.0f - m) + x * m;
        }
        return x;
    };
    if (NPT > 0) {
#pragma unroll
        for (int k = 0; k < NPT / 4; ++k) {
            const int i = 4 * (t + k * 1024);
            if (i < hot_n) {
                float4 l;
                if constexpr (KEEP_L) l = make_float4(rl[4 * k], rl[4 * k + 1], rl[4 * k + 2], rl[4 * k + 3]);
                else l = *reinterpret_cast<const float4*>(lat + i);
                *reinterpret_cast<float4*>(lat + i) = make_float4(value(i, ru[4 * k], rc[4 * k], l.x), value(i + 1, ru[4 * k + 1], rc[4 * k + 1], l.y),
                                                                  value(i + 2, ru[4 * k + 2], rc[4 * k + 2], l.z), value(i + 3, ru[4 * k + 3], rc[4 * k + 3], l.w));
            }
        }
    } else {
        for (int i = t; i < hot_n; i += 1024) lat[i] = value(i, u[i], c[i], lat[i]);
    }
    if (p.advance_in_kernel && t == 0) {
        // every workgroup read *step_ptr at its start and takes its ticket here, at its end: the one that draws the last
        // ticket knows that all of them have read the step, so it may move it (and clears the tickets for the next launch)
        const int ticket = __hip_atomic_fetch_add(hot_step_ptr + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ticket == hot_batch - 1) {
            __hip_atomic_store(hot_step_ptr + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(hot_step_ptr, step_raw + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ void step_advance_kernel(int32_t* step_ptr) { *step_ptr = *step_ptr + 1; }

extern "C" int msd_cfg_step(const MsdCfgStep* q, msd_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!q || !q->eps || !q->latent || !q->coef) MSD_FAIL(MSD_E_ARG, "cfg_step: null pointer");
    if (q->batch <= 0 || q->n <= 0 || q->num_steps <= 0) MSD_FAIL(MSD_E_ARG, "cfg_step: bad dims");
    const bool vec4 = (q->n % 4) == 0 && q->n >= 4 && msd_aligned16(q->eps) && msd_aligned16(q->latent);   // (the register forms use 16-byte accesses)
    if (q->advance && !q->step_ptr) MSD_FAIL(MSD_E_ARG, "cfg_step: advance needs step_ptr");
    if (q->advance < 0 || q->advance > 2) MSD_FAIL(MSD_E_ARG, "cfg_step: advance takes 0, 1 or 2");
    CfgArgs a;
    a.eps = q->eps; a.latent = q->latent; a.coef = q->coef; a.step_ptr = q->step_ptr;
    a.advance_in_kernel = q->advance == 2 ? 1 : 0;
    a.batch = q->batch; a.n = q->n; a.num_steps = q->num_steps; a.guidance = q->guidance; a.rescale = q->guidance_rescale;
    a.ip_init = q->inpaint_init; a.ip_noise = q->inpaint_noise; a.ip_mask = q->inpaint_mask;
    if (a.ip_mask && (!a.ip_init || !a.ip_noise)) MSD_FAIL(MSD_E_ARG, "cfg_step: inpaint_mask needs inpaint_init and inpaint_noise");
    a.step_noise = q->step_noise; a.noise_coef = q->noise_coef;
    if (a.step_noise && !a.noise_coef) MSD_FAIL(MSD_E_ARG, "cfg_step: step_noise needs noise_coef");
    if (vec4 && q->n <= 16 * 1024) hipLaunchKernelGGL(cfg_step_kernel<16>, dim3(q->batch), dim3(1024), 0, stream, a.eps, a.latent, a.coef, a.step_ptr, a.batch, a.n, a.num_steps, a.guidance, a);        // <= 64x64 latents
    else if (vec4 && q->n <= 36 * 1024) hipLaunchKernelGGL(cfg_step_kernel<36>, dim3(q->batch), dim3(1024), 0, stream, a.eps, a.latent, a.coef, a.step_ptr, a.batch, a.n, a.num_steps, a.guidance, a);   // 96x96
    else hipLaunchKernelGGL(cfg_step_kernel<0>, dim3(q->batch), dim3(1024), 0, stream, a.eps, a.latent, a.coef, a.step_ptr, a.batch, a.n, a.num_steps, a.guidance, a);
    MSD_CHECK_LAUNCH();
    if (q->advance == 1) {
        hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, stream, q->step_ptr);
        MSD_CHECK_LAUNCH();
    }
    return MSD_OK;
}

__global__ __launch_bounds__(256) void add_bf16_kernel(const uint4* a, const uint4* b, uint4* o, long long nvec) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        float fa[8], fb[8];
        unpack8(a[i], fa); unpack8(b[i], fb);
#pragma unroll
        for (int e = 0; e < 8; ++e) fa[e] += fb[e];
        o[i] = pack8(fa);
    }
}
// out = bf16(a + b), a bf16, b fp32, summed in fp32 (8 elements per thread: one 16-byte and two 16-byte loads)
__global__ __launch_bounds__(256) void add_f32_bf16_kernel(const uint4* a, const float4* b, uint4* o, long long nvec) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        float fa[8];
        unpack8(a[i], fa);
        const float4 b0 = b[2 * i], b1 = b[2 * i + 1];
        fa[0] += b0.x; fa[1] += b0.y; fa[2] += b0.z; fa[3] += b0.w;
        fa[4] += b1.x; fa[5] += b1.y; fa[6] += b1.z; fa[7] += b1.w;
        o[i] = pack8(fa);
    }
}
__global__ __launch_bounds__(256) void cast_f2b_kernel(const float* in, bf16_t* out, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = f2bf(in[i]);
}
__global__ __launch_bounds__(256) void cast_b2f_kernel(const bf16_t* in, float* out, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = bf2f(in[i]);
}
static unsigned grid_for(long long n) {
    long long b = (n + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (unsigned)b;
}
extern "C" int msd_add_bf16(const void* a, const void* b, void* out, int64_t n, msd_stream_t stream_) {
    if (!a || !b || !out || n <= 0 || (n % 8)) MSD_FAIL(MSD_E_ARG, "add_bf16: bad arguments");
    if (!msd_aligned16(a) || !msd_aligned16(b) || !msd_aligned16(out)) MSD_FAIL(MSD_E_ALIGN, "add_bf16: alignment");
    hipLaunchKernelGGL(add_bf16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream_, (const uint4*)a,
                       (const uint4*)b, (uint4*)out, (long long)(n / 8));
    MSD_CHECK_LAUNCH();
    return MSD_OK;
}
extern "C" int msd_add_f32_bf16(const void* a, const float* b, void* out, int64_t n, msd_stream_t stream_) {
    if (!a || !b || !out || n <= 0 || (n % 8)) MSD_FAIL(MSD_E_ARG, "add_f32_bf16: bad arguments");
    if (!msd_aligned16(a) || !msd_aligned16(b) || !msd_aligned16(out)) MSD_FAIL(MSD_E_ALIGN, "add_f32_bf16: alignment");
    hipLaunchKernelGGL(add_f32_bf16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream_, (const uint4*)a,
                       (const float4*)b, (uint4*)out, (long long)(n / 8));
    MSD_CHECK_LAUNCH();
    return MSD_OK;
}
extern "C" int msd_cast_f32_to_bf16(const float* in, void* out, int64_t n, msd_stream_t stream_) {
    if (!in || !out || n <= 0) MSD_FAIL(MSD_E_ARG, "cast: bad arguments");
    hipLaunchKernelGGL(cast_f2b_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream_, in, (bf16_t*)out, (long long)n);
    MSD_CHECK_LAUNCH();
    return MSD_OK;
}
extern "C" int msd_cast_bf16_to_f32(const void* in, float* out, int64_t n, msd_stream_t stream_) {
    if (!in || !out || n <= 0) MSD_FAIL(MSD_E_ARG, "cast: bad arguments");
    hipLaunchKernelGGL(cast_b2f_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream_, (const bf16_t*)in, out, (long long)n);
    MSD_CHECK_LAUNCH();
    return MSD_OK;
}
// dst holds `copies` replicas of the `nvec` 16-byte vectors at src, back to back; src == dst: replica 0 is in place
__global__ __launch_bounds__(256) void replicate_kernel(const uint4* src, uint4* dst, long long nvec, int copies, int first) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        const uint4 v = src[i];
        for (int j = first; j < copies; ++j) dst[(long long)j * nvec + i] = v;
    }
}
extern "C" int msd_replicate(const void* src, void* dst, int64_t bytes, int32_t copies, msd_stream_t stream_) {
    if (!src || !dst || bytes <= 0 || (bytes % 16) || copies < 1 || copies > 64) MSD_FAIL(MSD_E_ARG, "replicate: bad arguments (bytes %% 16 == 0, 1 <= copies <= 64)");
    if (!msd_aligned16(src) || !msd_aligned16(dst)) MSD_FAIL(MSD_E_ALIGN, "replicate: alignment");
    const char* s0 = (const char*)src;
    const char* d0 = (const char*)dst;
    const bool in_place = s0 == d0;
    // any other overlap of the source with the replicas would read bytes this launch writes
    if (!in_place && s0 < d0 + (long long)copies * bytes && d0 < s0 + bytes) MSD_FAIL(MSD_E_ARG, "replicate: src overlaps dst (only src == dst is allowed)");
    if (in_place && copies == 1) return MSD_OK;
    hipLaunchKernelGGL(replicate_kernel, dim3(grid_for(bytes / 16)), dim3(256), 0, (hipStream_t)stream_, (const uint4*)src, (uint4*)dst,
                       (long long)(bytes / 16), (int)copies, in_place ? 1 : 0);
    MSD_CHECK_LAUNCH();
    return MSD_OK;
}

extern "C" int msd_memset_zero(void* ptr, int64_t bytes, msd_stream_t stream_) {
    if (!ptr || bytes <= 0) MSD_FAIL(MSD_E_ARG, "memset_zero: bad arguments");
    hipError_t e = hipMemsetAsync(ptr, 0, (size_t)bytes, (hipStream_t)stream_);
    if (e != hipSuccess) MSD_FAIL((int)e, "hipMemsetAsync: %s", hipGetErrorString(e));
    return MSD_OK;
}


// ---- CLIP token + position embedding (text_encoder.py:22-33) ------------------------------------
__global__ __launch_bounds__(256) void embedding_sum_kernel(const int32_t* tokens, const int32_t* positions, const float* tok,
                                                            const float* pos, bf16_t* out, int rows, int dim, int vocab,
                                                            int max_len, int32_t* status) {
    const int nq = dim >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)rows * nq) return;
    const int r = (int)(idx / nq), c = (int)(idx - (long long)r * nq) * 4;
    int ti = tokens[r], pi = positions[r];
    if ((unsigned)ti >= (unsigned)vocab || (unsigned)pi >= (unsigned)max_len) {
        if (status) *status = 1;
        ti = 0; pi = 0;
    }
    const float4 a = *reinterpret_cast<const float4*>(tok + (size_t)ti * dim + c);
    const float4 b = *reinterpret_cast<const float4*>(pos + (size_t)pi * dim + c);
    uint2 o;
    o.x = pack_bf2(a.x + b.x, a.y + b.y);
    o.y = pack_bf2(a.z + b.z, a.w + b.w);
    *reinterpret_cast<uint2*>(out + (size_t)r * dim + c) = o;
}

extern "C" int msd_embedding_sum(const int32_t* tokens, const int32_t* positions, const float* tok_table, const float* pos_table,
                                 void* out, int32_t rows, int32_t dim, int32_t vocab, int32_t max_len, int32_t* status,
                                 msd_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!tokens || !positions || !tok_table || !pos_table || !out) MSD_FAIL(MSD_E_ARG, "embedding_sum: null pointer");
    if (rows <= 0 || dim <= 0 || (dim % 4) || vocab <= 0 || max_len <= 0) MSD_FAIL(MSD_E_ARG, "embedding_sum: bad dims");
    if (!msd_aligned16(tok_table) || !msd_aligned16(pos_table) || (((uintptr_t)out) & 7u))
        MSD_FAIL(MSD_E_ALIGN, "embedding_sum: tables must be 16-byte aligned, out 8-byte aligned");
    const long long n = (long long)rows * (dim / 4);
    hipLaunchKernelGGL(embedding_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, tokens, positions,
                       tok_table, pos_table, (bf16_t*)out, rows, dim, vocab, max_len, status);
    MSD_CHECK_LAUNCH();
    return MSD_OK;
}

// Small-channel direct convolution / dense layer on vector FMAs (fp32 weights, fp32 accumulate).
//
// For the handful of layers whose channel counts cannot fill an MFMA tile (see
// include/minsdtf_hip.h): UNet conv_in / conv_out (diffusion_model.py:191,279), VAE post_quant /
// conv_in / conv_out (image_decoder.py:28-29,53), HintNet (control_net.py:14-30) and the
// time-embedding MLP + per-ResBlock projections (diffusion_model.py:30,184-188).
// Work item = (output pixel, group of CG output channels).  Consecutive work items walk the
// channel groups of one pixel first, so a wave reads ONE input pixel (broadcast) and a contiguous
// run of the Keras-layout weights [tap][c_in][c_out] (coalesced); with a single channel group
// (c_out <= 4) consecutive lanes are consecutive pixels and the weights broadcast instead.
#include "common.h"

struct CDArgs {
    const void* in; const float* w; const float* bias; const bf16_t* residual; void* out;
    int batch, in_batch_mod, h_in, w_in, c_in, h_out, w_out, c_out, ksize, stride, pad;
    int in_f32, out_dtype, act, act_in, ncg;
    long long total;
    uint32_t mg_ncg, mg_hw, mg_w, mg_bmod;   // floor(2^32 / d) for ncg, h_out*w_out, w_out, in_batch_mod (udiv_magic; total < 2^31 checked)
    float in_scale;
};

template <int CG, bool IN_F32>
__global__ __launch_bounds__(256) void conv_direct_kernel(const CDArgs p) {
    // (32-bit index math with host-prepared magic numbers: the 64-bit divisions by runtime values this replaced were
    //  several hundred instructions per thread)
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int)p.total) return;
    const int pix = udiv_magic(gid, p.ncg, p.mg_ncg);
    const int cg = gid - pix * p.ncg;
    const int hw = p.h_out * p.w_out;
    const int b = udiv_magic(pix, hw, p.mg_hw);
    const int rem = pix - b * hw;
    const int y = udiv_magic(rem, p.w_out, p.mg_w), x = rem - y * p.w_out;
    const int co = cg * CG;
    const bool full = (co + CG <= p.c_out);

    float acc[CG];
#pragma unroll
    for (int k = 0; k < CG; ++k) acc[k] = (p.bias && co + k < p.c_out) ? p.bias[co + k] : 0.f;

    const int bi = b - udiv_magic(b, p.in_batch_mod, p.mg_bmod) * p.in_batch_mod;
    for (int ky = 0; ky < p.ksize; ++ky) {
        const int iy = y * p.stride + ky - p.pad;
        if ((unsigned)iy >= (unsigned)p.h_in) continue;
        for (int kx = 0; kx < p.ksize; ++kx) {
            const int ix = x * p.stride + kx - p.pad;
            if ((unsigned)ix >= (unsigned)p.w_in) continue;
            const size_t ioff = ((size_t)(bi * p.h_in + iy) * p.w_in + ix) * p.c_in;
            const float* wp = p.w + (size_t)((ky * p.ksize + kx) * p.c_in) * p.c_out + co;
            for (int ci = 0; ci < p.c_in; ++ci) {
                float xv;
                if (IN_F32) xv = reinterpret_cast<const float*>(p.in)[ioff + ci];
                else xv = bf2f(reinterpret_cast<const bf16_t*>(p.in)[ioff + ci]);
                xv *= p.in_scale;
                if (p.act_in) xv = silu_f(xv);
                const float* wr = wp + (size_t)ci * p.c_out;
                if (full) {
#pragma unroll
                    for (int k = 0; k < CG; ++k) acc[k] += xv * wr[k];
                } else {
#pragma unroll
                    for (int k = 0; k < CG; ++k)
                        if (co + k < p.c_out) acc[k] += xv * wr[k];
                }
            }
        }
    }
    const size_t ooff = (size_t)pix * p.c_out + co;
#pragma unroll
    for (int k = 0; k < CG; ++k) {
        if (co + k >= p.c_out) break;
        float v = acc[k];
        if (p.act == MSD_ACT_SILU) v = silu_f(v);
        if (p.residual) v += bf2f(p.residual[ooff + k]);
        if (p.out_dtype == MSD_OUT_F32) reinterpret_cast<float*>(p.out)[ooff + k] = v;
        else if (p.out_dtype == MSD_OUT_BF16) reinterpret_cast<bf16_t*>(p.out)[ooff + k] = f2bf(v);
        else {
            // stable_diffusion.py:483-486: clip(((v + 1) * 0.5) * 255, 0, 255).astype(uint8) — truncation
            float u = ((v + 1.0f) * 0.5f) * 255.0f;
            u = fminf(fmaxf(u, 0.0f), 255.0f);
            reinterpret_cast<uint8_t*>(p.out)[ooff + k] = (uint8_t)u;
        }
    }
}

// ---- Dense on a handful of rows (the time-embedding MLP: `steps` rows x 320 -> 1280 -> 1280, fp32) ---------------
// conv_direct_kernel gives such a launch one thread per (row, 8 outputs) = 16 workgroups walking K serially (170-680 us
// per layer).  Here a workgroup owns 64 output columns x DR_ROWS rows: wave q takes a quarter of K, lane n one column —
// a weight row segment is one coalesced 256-byte load per wave and k, the input values are wave-uniform (scalar
// operands of the FMAs) — and the four partial sums of a column meet in LDS, added in wave order.
constexpr int DR_ROWS = 16;
__global__ __launch_bounds__(256) void dense_rows_kernel(const CDArgs p) {
    __shared__ float part[4][DR_ROWS][64];
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = blockIdx.x * 64 + lane;
    const int row0 = blockIdx.y * DR_ROWS;
    const int nrows = min(DR_ROWS, p.batch - row0);   // (wave-uniform)
    const int K = p.c_in, kq = K >> 2;
    const float* __restrict__ x = reinterpret_cast<const float*>(p.in) + (size_t)row0 * K + q * kq;
    const float* __restrict__ w = p.w + (size_t)(q * kq) * p.c_out + min(n, p.c_out - 1);
    float acc[DR_ROWS];
#pragma unroll
    for (int m = 0; m < DR_ROWS; ++m) acc[m] = 0.f;
    for (int k0 = 0; k0 < kq; k0 += 8) {
        float wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) wv[u] = w[(size_t)(k0 + u) * p.c_out];
#pragma unroll
        for (int m = 0; m < DR_ROWS; ++m) {
            if (m < nrows) {
                const float* xr = x + (size_t)m * K + k0;
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[m] += xr[u] * wv[u];
            }
        }
    }
#pragma unroll
    for (int m = 0; m < DR_ROWS; ++m) part[q][m][lane] = acc[m];
    __syncthreads();
    // 64 columns x nrows rows, 4 per thread
    for (int e = threadIdx.x; e < nrows * 64; e += 256) {
        const int m = e >> 6, c = e & 63, nn = blockIdx.x * 64 + c;
        if (nn >= p.c_out) continue;
        float v = ((part[0][m][c] + part[1][m][c]) + part[2][m][c]) + part[3][m][c];
        if (p.bias) v += p.bias[nn];
        if (p.act == MSD_ACT_SILU) v = silu_f(v);
        const size_t o = (size_t)(row0 + m) * p.c_out + nn;
        if (p.out_dtype == MSD_OUT_F32) reinterpret_cast<float*>(p.out)[o] = v;
        else reinterpret_cast<bf16_t*>(p.out)[o] = f2bf(v);
    }
}

// ---- few-output-channel specialisation (UNet conv_out 320->4, VAE conv_out 128->3) -------------
// 4 lanes share one output pixel: lane s reads the 16-byte channel vectors cv = s, s+4, ... of each
// tap (so a wave's loads stay contiguous per pixel), the whole fp32 filter sits in LDS as
// [tap*c_in + c] -> float4 (c_out padded to 4; every lane of a wave with the same s reads the same
// address, which LDS broadcasts), and the 4 partial sums are combined with two lane shuffles.
__global__ __launch_bounds__(256) void conv_pix4_kernel(const CDArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float4* wl = reinterpret_cast<float4*>(smem_raw);
    const int nk = p.ksize * p.ksize * p.c_in;
    for (int i = threadIdx.x; i < nk; i += 256) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* wr = p.w + (size_t)i * p.c_out;
        v.x = wr[0];
        if (p.c_out > 1) v.y = wr[1];
        if (p.c_out > 2) v.z = wr[2];
        if (p.c_out > 3) v.w = wr[3];
        wl[i] = v;
    }
    __syncthreads();
    const int sub = threadIdx.x & 3;
    const int pix = blockIdx.x * 64 + (threadIdx.x >> 2);
    const int npix = p.batch * p.h_out * p.w_out;
    const bool live = pix < npix;
    const int pc = live ? pix : npix - 1;  // keep all lanes in the shuffles
    const int hw = p.h_out * p.w_out;
    const int b = udiv_magic(pc, hw, p.mg_hw);
    const int rem = pc - b * hw;
    const int y = udiv_magic(rem, p.w_out, p.mg_w), x = rem - y * p.w_out;
    const int bi = b - udiv_magic(b, p.in_batch_mod, p.mg_bmod) * p.in_batch_mod;
    const int ncv = p.c_in >> 3;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int ky = 0; ky < p.ksize; ++ky) {
        const int iy = y * p.stride + ky - p.pad;
        if ((unsigned)iy >= (unsigned)p.h_in) continue;
        for (int kx = 0; kx < p.ksize; ++kx) {
            const int ix = x * p.stride + kx - p.pad;
            if ((unsigned)ix >= (unsigned)p.w_in) continue;
            const bf16_t* ip = reinterpret_cast<const bf16_t*>(p.in) + ((size_t)(bi * p.h_in + iy) * p.w_in + ix) * p.c_in;
            const float4* wt = wl + (size_t)(ky * p.ksize + kx) * p.c_in;
            for (int cv = sub; cv < ncv; cv += 4) {
                float f[8];
                unpack8(*reinterpret_cast<const uint4*>(ip + cv * 8), f);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float xv = f[e] * p.in_scale;
                    if (p.act_in) xv = silu_f(xv);
                    const float4 w4 = wt[cv * 8 + e];
                    a0 += xv * w4.x; a1 += xv * w4.y; a2 += xv * w4.z; a3 += xv * w4.w;
                }
            }
        }
    }
    a0 += __shfl_xor(a0, 1); a1 += __shfl_xor(a1, 1); a2 += __shfl_xor(a2, 1); a3 += __shfl_xor(a3, 1);
    a0 += __shfl_xor(a0, 2); a1 += __shfl_xor(a1, 2); a2 += __shfl_xor(a2, 2); a3 += __shfl_xor(a3, 2);
    if (!live || sub >= p.c_out) return;
    float v = sub == 0 ? a0 : (sub == 1 ? a1 : (sub == 2 ? a2 : a3));  // lane s writes channel s
    if (p.bias) v += p.bias[sub];
    if (p.act == MSD_ACT_SILU) v = silu_f(v);
    const size_t o = (size_t)pix * p.c_out + sub;
    if (p.residual) v += bf2f(p.residual[o]);
    if (p.out_dtype == MSD_OUT_F32) reinterpret_cast<float*>(p.out)[o] = v;
    else if (p.out_dtype == MSD_OUT_BF16) reinterpret_cast<bf16_t*>(p.out)[o] = f2bf(v);
    else {
        float u = ((v + 1.0f) * 0.5f) * 255.0f;  // stable_diffusion.py:483-486, truncating cast
        u = fminf(fmaxf(u, 0.0f), 255.0f);
        reinterpret_cast<uint8_t*>(p.out)[o] = (uint8_t)u;
    }
}

// ---- 4-channel fp32 input, 3x3, stride 1 (UNet / ControlNet conv_in 4 -> 320, VAE decoder.conv_in 4 -> 512) ---------
// K = 36: nothing for an MFMA tile to chew on, and the generic kernel above walks it with runtime loop bounds and scalar
// weight loads (34 us per denoise step for 0.19 GFLOP).  Here a thread OWNS 4 output channels: their 36 x 4 filter taps
// sit in 144 registers for the whole kernel, a workgroup takes a TPX-pixel row segment, stages its 3 x (TPX + 2) x 4
// input patch in LDS (zero padded) and every thread walks the pixels of its pixel group: 36 broadcast LDS reads + 144
// FMAs + one 8-byte store per pixel; a pixel's row of c_out bf16 values is written by c_out / 4 consecutive threads.
template <int TPX>
__global__ __launch_bounds__(256) void conv_in4_kernel(const CDArgs p, int cq_n, int pxg_n, int tiles_x, uint32_t mg_tx, uint32_t mg_cq) {
    __shared__ float4 patch[3][TPX + 2];
    const int t = threadIdx.x;
    const int row = udiv_magic(blockIdx.x, tiles_x, mg_tx);          // (b, y)
    const int x0 = (blockIdx.x - row * tiles_x) * TPX;
    const int b = udiv_magic(row, p.h_out, p.mg_hw), y = row - b * p.h_out;   // (mg_hw = magic of h_out for this kernel)
    const int bi = b - udiv_magic(b, p.in_batch_mod, p.mg_bmod) * p.in_batch_mod;
    if (t < 3 * (TPX + 2)) {
        const int ky = t / (TPX + 2), xx = t - ky * (TPX + 2);
        const int iy = y + ky - 1, ix = x0 + xx - 1;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)iy < (unsigned)p.h_in && (unsigned)ix < (unsigned)p.w_in) {
            v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.in) + ((size_t)(bi * p.h_in + iy) * p.w_in + ix) * 4);
            v.x *= p.in_scale; v.y *= p.in_scale; v.z *= p.in_scale; v.w *= p.in_scale;
        }
        patch[ky][xx] = v;
    }
    const int pg = udiv_magic(t, cq_n, mg_cq), cq = t - pg * cq_n;
    const bool active = pg < pxg_n;
    const int co = (active ? cq : 0) * 4;
    float4 w[36];
#pragma unroll
    for (int k = 0; k < 36; ++k) w[k] = *reinterpret_cast<const float4*>(p.w + (size_t)k * p.c_out + co);
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) bv = *reinterpret_cast<const float4*>(p.bias + co);
    __syncthreads();
    if (!active) return;
    for (int px = pg; px < TPX; px += pxg_n) {
        float4 a = bv;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float4 in = patch[ky][px + kx];
                const int k = (ky * 3 + kx) * 4;
                a.x += in.x * w[k].x + in.y * w[k + 1].x + in.z * w[k + 2].x + in.w * w[k + 3].x;
                a.y += in.x * w[k].y + in.y * w[k + 1].y + in.z * w[k + 2].y + in.w * w[k + 3].y;
                a.z += in.x * w[k].z + in.y * w[k + 1].z + in.z * w[k + 2].z + in.w * w[k + 3].z;
                a.w += in.x * w[k].w + in.y * w[k + 1].w + in.z * w[k + 2].w + in.w * w[k + 3].w;
            }
        const size_t o = ((size_t)(b * p.h_out + y) * p.w_out + x0 + px) * p.c_out + co;
        if (p.residual) {
            const uint2 rr = *reinterpret_cast<const uint2*>(p.residual + o);
            a.x += bf_lo(rr.x); a.y += bf_hi(rr.x); a.z += bf_lo(rr.y); a.w += bf_hi(rr.y);
        }
        uint2 ov; ov.x = pack_bf2(a.x, a.y); ov.y = pack_bf2(a.z, a.w);
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + o) = ov;
    }
}

extern "C" int msd_conv_direct(const MsdConvDirect* q, msd_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!q || !q->in || !q->w || !q->out) MSD_FAIL(MSD_E_ARG, "conv_direct: null pointer");
    if (q->batch <= 0 || q->in_batch_mod <= 0 || q->h_in <= 0 || q->w_in <= 0 || q->c_in <= 0 || q->c_out <= 0 ||
        q->h_out <= 0 || q->w_out <= 0)
        MSD_FAIL(MSD_E_ARG, "conv_direct: bad dims");
    if (q->ksize < 1 || q->ksize > 3 || q->stride < 1 || q->stride > 2 || q->pad < 0 || q->pad > 1)
        MSD_FAIL(MSD_E_UNSUPPORTED, "conv_direct: ksize/stride/pad");
    {
        const int ho = (q->h_in + 2 * q->pad - q->ksize) / q->stride + 1, wo = (q->w_in + 2 * q->pad - q->ksize) / q->stride + 1;
        if (ho != q->h_out || wo != q->w_out) MSD_FAIL(MSD_E_ARG, "conv_direct: output dims do not match the geometry");
    }
    if (q->in_dtype != MSD_OUT_BF16 && q->in_dtype != MSD_OUT_F32) MSD_FAIL(MSD_E_ARG, "conv_direct: in_dtype");
    if (q->out_dtype < 0 || q->out_dtype > 2) MSD_FAIL(MSD_E_ARG, "conv_direct: out_dtype");
    if (q->act != MSD_ACT_NONE && q->act != MSD_ACT_SILU) MSD_FAIL(MSD_E_ARG, "conv_direct: act");
    CDArgs a;
    a.in = q->in; a.w = q->w; a.bias = q->bias; a.residual = (const bf16_t*)q->residual; a.out = q->out;
    a.batch = q->batch; a.in_batch_mod = q->in_batch_mod; a.h_in = q->h_in; a.w_in = q->w_in; a.c_in = q->c_in;
    a.h_out = q->h_out; a.w_out = q->w_out; a.c_out = q->c_out; a.ksize = q->ksize; a.stride = q->stride; a.pad = q->pad;
    a.in_f32 = q->in_dtype == MSD_OUT_F32; a.out_dtype = q->out_dtype; a.act = q->act; a.act_in = q->act_in ? 1 : 0;
    a.in_scale = q->in_scale;
    a.mg_hw = udiv_magic_of(q->h_out * q->w_out); a.mg_w = udiv_magic_of(q->w_out); a.mg_bmod = udiv_magic_of(q->in_batch_mod);
    a.mg_ncg = 0;
    if ((long long)q->batch * q->h_out * q->w_out * ((q->c_out + 3) / 4) >= (1ll << 31))
        MSD_FAIL(MSD_E_UNSUPPORTED, "conv_direct: more than 2^31 work items");
    // 4-channel fp32 input, 3x3 s1, bf16 output: the register-resident-filter kernel
    if (a.in_f32 && q->c_in == 4 && q->ksize == 3 && q->stride == 1 && q->pad == 1 && q->out_dtype == MSD_OUT_BF16 &&
        q->act == MSD_ACT_NONE && !q->act_in && (q->c_out % 4) == 0 && q->c_out >= 64 && q->c_out <= 1024 && (q->w_out % 8) == 0 &&
        msd_aligned16(q->in) && msd_aligned16(q->w) && msd_aligned16(q->bias) && (((uintptr_t)q->out) & 7u) == 0 &&
        (((uintptr_t)q->residual) & 7u) == 0) {
        const int tpx = (q->w_out % 32) == 0 ? 32 : ((q->w_out % 16) == 0 ? 16 : 8);
        const int cq_n = q->c_out / 4, pxg_n = 256 / cq_n < tpx ? 256 / cq_n : tpx;
        const int tiles_x = q->w_out / tpx;
        const unsigned blocks = (unsigned)((long long)q->batch * q->h_out * tiles_x);
        a.mg_hw = udiv_magic_of(q->h_out);
        if (tpx == 32) hipLaunchKernelGGL(conv_in4_kernel<32>, dim3(blocks), dim3(256), 0, stream, a, cq_n, pxg_n, tiles_x, udiv_magic_of(tiles_x), udiv_magic_of(cq_n));
        else if (tpx == 16) hipLaunchKernelGGL(conv_in4_kernel<16>, dim3(blocks), dim3(256), 0, stream, a, cq_n, pxg_n, tiles_x, udiv_magic_of(tiles_x), udiv_magic_of(cq_n));
        else hipLaunchKernelGGL(conv_in4_kernel<8>, dim3(blocks), dim3(256), 0, stream, a, cq_n, pxg_n, tiles_x, udiv_magic_of(tiles_x), udiv_magic_of(cq_n));
        MSD_CHECK_LAUNCH();
        return MSD_OK;
    }
    // Dense on a few rows (time-embedding MLP): fp32 in, fp32 / bf16 out, K a multiple of 32
    if (a.in_f32 && q->ksize == 1 && q->h_in == 1 && q->w_in == 1 && q->stride == 1 && q->pad == 0 && q->in_batch_mod == q->batch &&
        q->batch <= 4096 && (q->c_in % 32) == 0 && q->c_out >= 64 && !q->residual && !q->act_in && q->in_scale == 1.0f &&
        q->out_dtype != MSD_OUT_U8) {
        hipLaunchKernelGGL(dense_rows_kernel, dim3((unsigned)((q->c_out + 63) / 64), (unsigned)((q->batch + DR_ROWS - 1) / DR_ROWS)), dim3(256), 0,
                           stream, a);
        MSD_CHECK_LAUNCH();
        return MSD_OK;
    }
    const int cg = q->c_out <= 4 ? 4 : 8;
    {
        const long long nk = (long long)q->ksize * q->ksize * q->c_in;
        if (q->c_out <= 4 && !a.in_f32 && (q->c_in % 32) == 0 && nk * 16 <= 64 * 1024 && msd_aligned16(q->in)) {
            const long long npix = (long long)q->batch * q->h_out * q->w_out;
            a.ncg = 1; a.total = npix;
            hipLaunchKernelGGL(conv_pix4_kernel, dim3((unsigned)((npix + 63) / 64)), dim3(256), (size_t)nk * 16, stream, a);
            MSD_CHECK_LAUNCH();
            return MSD_OK;
        }
    }
    a.ncg = (q->c_out + cg - 1) / cg;
    a.mg_ncg = udiv_magic_of(a.ncg);
    a.total = (long long)q->batch * q->h_out * q->w_out * a.ncg;
    const unsigned blocks = (unsigned)((a.total + 255) / 256);
    if (cg == 4) {
        if (a.in_f32) hipLaunchKernelGGL((conv_direct_kernel<4, true>), dim3(blocks), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((conv_direct_kernel<4, false>), dim3(blocks), dim3(256), 0, stream, a);
    } else {
        if (a.in_f32) hipLaunchKernelGGL((conv_direct_kernel<8, true>), dim3(blocks), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((conv_direct_kernel<8, false>), dim3(blocks), dim3(256), 0, stream, a);
    }
    MSD_CHECK_LAUNCH();
    return MSD_OK;
}

/* minsdtf_hip.h — C ABI of libminsdtf_hip.so: the MI355X (gfx950) kernels behind the
 * SD1.5 denoise hot path of cpuimage/minSDTF.
 *
 * The reference has no native / FFI boundary (it is pure Python over Keras), so
 * there is no reference header to mirror.  Each entry point below replaces the stock Keras op(s)
 * the reference's hot path invokes; the reference call site is cited per function
 * (paths relative to /root/reference/stable_diffusion/).
 *
 * Conventions
 *  - every pointer is a raw DEVICE pointer (hipMalloc / torch tensor.data_ptr()); the caller owns
 *    all memory, including workspaces; nothing is allocated or freed inside the library;
 *  - activations are NHWC ("channels_last", like the reference's Keras tensors), bf16 unless noted;
 *  - `stream` is a hipStream_t; all work is stream-ordered, never synchronises, and is capturable
 *    into a hipGraph;
 *  - return value: 0 on success, a negative MSD_E_* code for an argument error (nothing was
 *    launched), or a positive hipError_t from the launch; msd_last_error() gives the text;
 *  - `step_ptr` (nullable) points at a device int32 holding the current denoise-step index, so a
 *    captured step graph can be replayed for every step: per-step tables (time-embedding
 *    projections, scheduler coefficients) are indexed with it on the device.
 */
#ifndef MINSDTF_HIP_H
#define MINSDTF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSD_ABI_VERSION 11

#define MSD_OK 0
#define MSD_E_ARG (-1)      /* bad / inconsistent argument */
#define MSD_E_ALIGN (-2)    /* pointer or leading dimension not aligned as required */
#define MSD_E_UNSUPPORTED (-3)
#define MSD_E_WORKSPACE (-4) /* workspace too small */

typedef void* msd_stream_t; /* hipStream_t */

/* The library is built with -fvisibility=hidden: the functions marked MSD_API below are its WHOLE dynamic symbol
 * table (tests/test_host_cpu.py compares `nm -D --defined-only` with this header). */
#if defined(__GNUC__) || defined(__clang__)
#define MSD_API __attribute__((visibility("default")))
#else
#define MSD_API
#endif

MSD_API int msd_abi_version(void);
MSD_API const char* msd_last_error(void);
/* One-time per-process set-up (raises the dynamic-LDS limits of the kernels). Idempotent. */
MSD_API int msd_init(void);
/* Tuning / A-B switches, e.g. ("conv_dense", 0|1), ("attn_qf", 0|1|2|4), ("attn_form", 0|1|2). Returns MSD_E_ARG for an unknown key. */
MSD_API int msd_set_option(const char* key, int value);

/* ------------------------------------------------------------------------------------------
 * msd_conv_gemm — implicit-GEMM convolution / dense layer on MFMA (bf16 in, fp32 accumulate).
 * Replaces: PaddedConv2D 3x3 s1/s2 and 1x1 (layers.py:17-25; diffusion_model.py:29,34,38,62,67,
 * 136,200,209,218), keras Dense (diffusion_model.py:60-65,90,102-108,146; layers.py:33-36),
 * UpSampling2D(2)+conv (diffusion_model.py:132-139; image_decoder.py:36-47), Concatenate feeding
 * a conv (diffusion_model.py:237-273) and the GEGLU gate (diffusion_model.py:142-153).
 *
 *   out[m, n] = epilogue( sum_k A[m, k] * W[n, k] )
 *   m = (b, y, x) output pixel / token,  M = batch*h_out*w_out
 *   k = (tap, c): tap = ky*ksize+kx, c over the concatenated channels of a0|a1, K = taps*(c0+c1)
 *   A[m,(tap,c)] = in[b, y*stride+ky-pad, x*stride+kx-pad, c]  (zero outside; with `upsample`
 *                  the logical input is the stored tensor repeated 2x2, nearest)
 *   W is pre-packed [N][K] bf16 (k contiguous) — see minsdtf_amd/packing.py.
 * Requirements: c0 % 64 == 0, c1 % 64 == 0, N % 4 == 0, all pointers 16-byte aligned,
 *               leading dimensions multiples of 4 elements.
 * Epilogue order: [LayerNorm fold] -> + bias[n] -> + rowvec[step, b, n] -> act -> + residual[m, n] -> store.
 *
 * LayerNorm fold (diffusion_model.py:84-88 followed by a Dense, :102-108 / :146): LN(x) W^T =
 * rstd[m] * (x (gamma*W)^T - mean[m] * colsum[n]) + (beta W^T)[n], so the LayerNormalization in front of
 * a Dense is this GEMM on the RAW rows with gamma folded into W (packing.py), corrected per row in the
 * epilogue; the row moments come from the GEMM that PRODUCED x: with `ln_out` set, a launch writes, per
 * output row and per column tile, the partial (sum, sum of squares) of the bf16 values it stored
 * (float2 [M][ln_out_slots], ln_out_slots = ceil(N / tile_n)); a launch with `ln_in` set sums the
 * `ln_in_slots` partials of each of its rows in a fixed order (bit-reproducible) and applies
 * acc <- rstd * (acc - mean * ln_colsum[n]) before the rest of the epilogue (bias then carries beta W^T + b).
 * Both are plain-K only (no split-K); `ln_out` needs plain mode with a bf16 output.
 */
#define MSD_ACT_NONE 0
#define MSD_ACT_SILU 1
#define MSD_ACT_GEGLU 2 /* W rows interleaved in 16-column x|gate groups; writes N/2 columns */
#define MSD_ACT_QUICK_GELU 3 /* x * sigmoid(1.702 x)  (CLIP MLP, text_encoder.py:100-101) */

#define MSD_OUT_BF16 0
#define MSD_OUT_F32 1

typedef struct MsdConvGemm {
    const void* a0;      /* bf16 [batch][h_in][w_in][c0] */
    const void* a1;      /* bf16 [batch][h_in][w_in][c1] or NULL (channel concat a0|a1) */
    const void* w;       /* bf16 [N][K] */
    const float* bias;   /* [N] or NULL */
    const float* rowvec; /* fp32, element (step*rv_step_stride + b*rv_batch_stride + n) or NULL */
    const int32_t* step_ptr; /* device int32 or NULL (step = 0) */
    const void* residual;    /* bf16 [M][res_ld] or NULL */
    void* out;           /* bf16 or fp32 [M][out_ld]  (split mode: part 0) */
    void* out1;          /* split mode part 1: bf16 [M][out1_ld] */
    void* out2;          /* split mode part 2: bf16 TRANSPOSED [batch][N-ns0-ns1][out2_ld] (token contiguous) */
    float* workspace;    /* fp32 split-K partial slabs, >= splitk*M*N floats (or NULL if splitk<=1) */
    int64_t workspace_floats;
    int32_t batch, h_in, w_in, c0, c1;
    int32_t h_out, w_out;
    int32_t ksize;       /* 1 or 3 */
    int32_t stride;      /* 1 or 2 */
    int32_t pad;         /* 0 or 1 */
    int32_t upsample;    /* 0 or 1: nearest x2 of the stored input before the conv */
    int32_t N;
    int32_t act;
    int32_t out_dtype;
    int32_t out_ld, res_ld;
    int32_t rv_step_stride, rv_batch_stride;
    int32_t split_mode;  /* 0: plain; 1: columns [0,ns0)->out, [ns0,ns0+ns1)->out1, rest->out2 transposed */
    int32_t ns0, ns1, out1_ld, out2_ld;
    int32_t splitk;      /* >=1; K-tiles are divided over this many slices */
    int32_t tile_n;      /* 0 = auto, else 64, 80 or 128 */
    int32_t tile_m;      /* 0 = 128, else 64 / 128 / 256; valid (tile_m x tile_n): 128x128 128x64 64x64 64x128 256x128 128x80;
                            1128 / 1256 = halo-tile 3x3 kernel with 8x16 / 16x16 pixel tiles (x 64 / 80 / 128 channels;
                            falls back if not eligible).  The 80-wide tiles serve N = 320 / 640 at small batch, where
                            they make the workgroup count a multiple of the 256 CUs;
                            4000 + rows (4064 / 4128 / 4256) = the wreg form (csrc/conv_wreg.hip): the weights are read global ->
                            VGPR from the fragment-major image (w_layout 2, required: MSD_E_ARG with any other layout), only the
                            activation tile goes through LDS; tile_n x stages must name a built configuration (MSD_E_UNSUPPORTED
                            otherwise: there is no fallback, the other forms cannot read that image); N % 16 == 0.  Same K walk
                            and epilogue as the tile kernel: the same results bit for bit;
                            5000 + rows (5256 / 5128) = the big form (csrc/conv_big.hip) for M >= 8192: 256 x 256 / 256 x 160 / 256 x 128 /
                            128 x 256 macro tiles on 8 waves whose two halves run one barrier apart (one multiplies while the other
                            reads fragments and issues the LDS-DMAs of the K tile one or two ahead); w_layout 0 or 1; no ln_out;
                            fewer than 2^24 input pixels; tile_n x stages must name a built configuration (MSD_E_UNSUPPORTED
                            otherwise).  Same K walk and epilogue as the tile kernel: the same results bit for bit.
                            tile_m >= 6000: MSD_E_ARG */
    int32_t stages;      /* 0 = default LDS ring depth of the tile; deeper rings built: 128x128:4 64x64:8 64x128:5 128x64:5
                            128x80:4; halo tiles 1128x64:8 1128x128:6 1128x80:8 1256x80:5 (an unknown depth = the default);
                            10 + depth = the tile on 8 waves (two per SIMD): 64x64:14 128x64:13 64x128:13;
                            20 + depth = 64x64 per wave: 128x128:23/24 128x64:24 64x128:24;
                            30 + depth = halo tiles with 3 filter taps (one filter row) per K step: 1128x64:33/34 1128x80:33
                            2128x64:33; 60 + depth = the same with two loader waves that do all the staging: 1128x64:63 1128x80:63;
                            90 + depth = the 60s' form with the K loop rotated (a step's barrier sits between its second and third tap, the
                            next step's first fragments are read under the third tap's MFMAs; same taps in the same order, same bits): 1128x64:93 1128x80:93;
                            150 + depth = the one-tap form rotated (the fragments of tap t + 1 are read under the MFMAs of tap t, every ring
                            stage in flight; same bits): 1128x64:153/158 1128x128:153/156 1128x80:158 1256x80:153/155 1256x128:153;
                            wreg form (tile_m 4000 + rows): depth 3 / 4, + 10 = 8 waves (4256x128: a 2 x 4 wave grid),
                            + 20 = two K tiles (128 channels) per ring stage: 4064x64:4/23/24 4064x128:3/4/23/24 4064x256:3/4/23
                            4128x64:3/4/23 4128x128:3/4/13/23 4256x64:3 4256x128:13/14;
                            big form (tile_m 5000 + rows): a configuration code, not a depth: 5256x256:0 (2 x 2 quadrant phases, 2 K-tile
                            buffers) 5256x160:0/1/2 (quadrants x 3 buffers / row halves x 3 / quadrants x 2) 5256x128:0/1 5128x256:0/1
                            (row halves / quadrants, 3 buffers); + 10 = the same configuration walking K chunk-major (3x3 convs without a
                            shortcut operand only): the halo-tile kernel's order of sums, hence ITS bits instead of the tile kernel's;
                            20 + code with tile_m 5256 = that chunk-major walk over a STAGED 18 x 18-pixel halo per 64-channel chunk
                            (3x3 / stride 1 / pad 1, with or without `upsample`, h_out and w_out multiples of 16; otherwise
                            MSD_E_UNSUPPORTED): 5256x160:20 5256x128:20/21 (weight ring of 3 / 4).  With a shortcut operand (a2) a
                            slice's shortcut chunks follow its main chunks; since round 6 the halo tiles (tile_m 1128 / 2128 / 1256, any
                            of their `stages`) walk a shortcut operand in exactly that order: one numerics class, the halo tiles for
                            small launches, this form from about 128 workgroups on */
    const float* ln_in;      /* float2 [M][ln_in_slots] row-moment partials of the input rows, or NULL */
    const float* ln_colsum;  /* [N]: sum_k W[n][k] of the gamma-folded bf16 weights (with ln_in) */
    float* ln_out;           /* float2 [M][ln_out_slots] row-moment partials of the stored output, or NULL */
    int32_t ln_in_slots;     /* 1..20 */
    int32_t ln_out_slots;    /* must equal ceil(N / tile_n) of this launch (msd_conv_gemm_ln_slots) */
    float ln_eps;
    /* Shortcut operand (ResBlock, diffusion_model.py:36-38,50: conv2(h) + conv_shortcut(x) as ONE contraction): after the
     * ksize*ksize*(c0+c1) taps of a0|a1, K continues with the channels of a2|a3 read at the OUTPUT pixel (a 1x1 tap);
     * W rows are [taps of a0|a1 ... | channels of a2|a3], K = ksize*ksize*(c0+c1) + c2 + c3.  Needs stride 1, same-size
     * output, c2 % 64 == 0, c3 % 64 == 0.  On the tile / wreg / big forms the shortcut chunks are K tiles behind the last tap
     * (tap-major class); on the halo tiles and the staged-halo big form they follow each split-K slice's main chunks, dealt
     * over the slices in order (chunk-major class; round 6). */
    const void* a2;          /* bf16 [batch][h_out][w_out][c2] or NULL */
    const void* a3;          /* bf16 [batch][h_out][w_out][c3] or NULL (channel concat a2|a3) */
    int32_t c2, c3;
    /* Storage order of w.  0: [N][K] rows.  1: chunk-major [K/64][N][64] — the 64-element K chunk kc of ALL output
     * columns is one contiguous N x 128-byte run, so the weight tile of a K step (any tile_n, any column offset) is a
     * single contiguous block of HBM instead of tile_n pieces of 128 bytes K*2 bytes apart (the packed form the
     * models keep; same values, same K order, same results).  2: fragment-major [K/64][N/16][2][64][8] — the operand
     * registers of v_mfma_f32_16x16x32_bf16 laid out in memory: lane l = 16 g + r of fragment (K tile kt, 16-column block
     * nb, half ks) holds W[16 nb + rho(r)][64 kt + 32 ks + 8 g .. + 8], rho = (0,1,2,3,8,9,10,11,4,5,6,7,12,13,14,15), at
     * byte 16 l of the fragment's contiguous KiB (minsdtf_amd/packing.py fragment_major); read by the wreg form only
     * (tile_m 4000 + rows), which loads a fragment with one coalesced global_load_dwordx4 per wave. */
    int32_t w_layout;
} MsdConvGemm;

/* Number of row-moment partials per row a launch with these parameters writes to `ln_out`
 * (= its number of column tiles), or a negative MSD_E_* code. */
MSD_API int msd_conv_gemm_ln_slots(const MsdConvGemm* p);

MSD_API int msd_conv_gemm(const MsdConvGemm* p, msd_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * msd_conv_direct — small-channel direct convolution / dense on vector FMAs (fp32 weights).
 * For the layers whose channel counts cannot fill an MFMA tile: UNet conv_in 4->320 and conv_out
 * 320->4 (diffusion_model.py:191,279), VAE post_quant/conv_in/conv_out (image_decoder.py:28-29,53),
 * HintNet (control_net.py:14-30), the time-embedding MLP and the per-ResBlock time projections
 * (diffusion_model.py:30,184-188) evaluated for all steps at once.
 *   w: fp32 Keras layout [ksize][ksize][c_in][c_out];  in: fp32 or bf16 NHWC;  batch index of the
 *   input is (b % in_batch_mod) so one latent can feed the cond and uncond halves.
 * Epilogue: (+bias) -> act(SiLU optional) -> (+residual bf16) -> out as bf16 / fp32 / uint8 where
 * uint8 = clip(trunc(((v+1)*0.5)*255), 0, 255) (stable_diffusion.py:483-486).
 */
#define MSD_OUT_U8 2
typedef struct MsdConvDirect {
    const void* in;
    const float* w;
    const float* bias;     /* or NULL */
    const void* residual;  /* bf16 [M][c_out] or NULL */
    void* out;
    int32_t batch, in_batch_mod, h_in, w_in, c_in;
    int32_t h_out, w_out, c_out;
    int32_t ksize, stride, pad;
    int32_t in_dtype;      /* MSD_OUT_BF16 or MSD_OUT_F32 */
    int32_t out_dtype;     /* MSD_OUT_BF16 / MSD_OUT_F32 / MSD_OUT_U8 */
    int32_t act;           /* MSD_ACT_NONE or MSD_ACT_SILU */
    int32_t act_in;        /* apply SiLU to the input values as they are read (0/1) */
    float in_scale;        /* input multiplier (VAE Rescaling 1/0.18215, image_decoder.py:27) */
} MsdConvDirect;

MSD_API int msd_conv_direct(const MsdConvDirect* p, msd_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * msd_group_norm — GroupNormalization(groups=32, eps) [+ swish], NHWC, fp32 statistics.
 * Replaces keras GroupNormalization + Activation("swish") (diffusion_model.py:27-28,32-33,57,
 * 277-278; layers.py:32,66-68,78-79; image_decoder.py:51-52) and the Concatenate in front of it
 * (reads x0|x1 through two base pointers, never materialised).
 *   x0: bf16 [batch][hw][c0], x1: bf16 [batch][hw][c1] or NULL;  out: bf16 [batch][hw][c0+c1]
 *   stats:    fp32 [batch][32][2] scratch; receives {mean, rstd} per group
 *   partials: fp32 scratch for per-workgroup partial moments, >= batch * MSD_GN_MAX_CHUNKS * 64
 *             floats is always enough; summed in a fixed order (no atomics: results are
 *             bit-reproducible run to run)
 *   sync:     NULL, or >= batch * MSD_GN_SYNC_WORDS_PER_SAMPLE 32-bit words, 8-byte aligned, ZERO when first used and
 *             written by nothing but msd_group_norm launches of ONE stream at a time (launches that may run concurrently
 *             need a block each).  With it, tensors of >= 2048 pixels per sample whose group slab fits in registers run as
 *             ONE launch in which 2 / 4 / 8 workgroups share each (sample, group) and exchange their partial moments
 *             through this block (ticket counters + {value, epoch} words; the parts are summed in part order:
 *             bit-reproducible).  GIVE-UP FLAG: the exchange poll is bounded (so that an out-of-order dispatch can never hang
 *             the GPU); a workgroup that gives up sets word [8] of its 64-word slot AND word [8] of the block (sync[8]) to 1
 *             and the launch ends with WRONG numbers in that group.  The return code cannot report it (the launch is
 *             asynchronous), so the CALLER must read sync[8] once per job, after the work has completed, and treat a
 *             non-zero word as a failed job (minsdtf_amd/engine.py: check_gn_sync raises HipExtensionError).  Never observed
 *             under in-order dispatch.  Without `sync`: statistics + apply launches, no exchange, no flag.
 *             Samples of >= 9216 pixels (msd_set_option "gn_rows"; 4096 pays from batch 2 per GPU) whose parts fit in registers take the
 *             ROW-MAJOR form of the same exchange: 4-64 parts per SAMPLE (each shared by 1 / 2 / 4 workgroups by channels when the launch
 *             would otherwise leave the chip idle: placement only, msd_set_option "gn_rows_q"), a part = a contiguous pixel range with all its
 *             channels (16-byte accesses), 64 granules per part; it shares the block (second region of every sample's share)
 *             and the give-up word.  (ABI 10: the share grew from 6,144 to 16,384 words for it.)
 *             The per-(sample, group) form deals its workgroups by XCD (an XCD takes the pixel ranges whose rows the conv launches
 *             around it give it; msd_set_option "gn_xmap" 0 = a group's parts on consecutive ids): placement only, the same bits.
 */
#define MSD_GN_SYNC_WORDS_PER_SAMPLE 16384
#define MSD_GN_MAX_CHUNKS 1024
typedef struct MsdGroupNorm {
    const void* x0;
    const void* x1;
    const float* gamma; /* [c0+c1] */
    const float* beta;  /* [c0+c1] */
    float* stats;
    float* partials;
    int64_t partials_floats;
    void* out;
    int32_t batch, hw, c0, c1;
    int32_t silu; /* 0/1 */
    float eps;
    uint32_t* sync;
    int64_t sync_words;
} MsdGroupNorm;

MSD_API int msd_group_norm(const MsdGroupNorm* p, msd_stream_t stream);

/* msd_layer_norm — LayerNormalization(eps) over the last axis (diffusion_model.py:84-88).
 * x, out: bf16 [rows][c], c % 8 == 0, c <= 2048. */
MSD_API int msd_layer_norm(const void* x, const float* gamma, const float* beta, void* out, int32_t rows, int32_t c,
                   float eps, msd_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * msd_attention — fused scaled-dot-product attention, online softmax, MFMA QK^T and PV.
 * Replaces einsum/softmax/einsum of CrossAttention.call (diffusion_model.py:110-127); the
 * (B,heads,S,T) score tensor never exists in HBM.
 *   q:  bf16 [batch][s][q_ld], head h at columns [h*d, (h+1)*d)
 *   k:  bf16 [batch][t][k_ld]
 *   vt: bf16 [batch][heads*d][vt_ld]  (V transposed: key index contiguous; vt_ld % 8 == 0,
 *       vt_ld >= t; columns >= t are padding and are ignored)
 *   out: bf16 [batch][s][o_ld]
 *   softmax(scale * q k^T) v, scale applied to the scores (diffusion_model.py:105,123).
 * head_dim in {40, 80, 160} (UNet) and 64 (CLIP text encoder).  causal != 0 (needs s == t) masks
 * key index > query index, the additive -inf upper triangle of text_encoder.py:75-78.
 */
typedef struct MsdAttention {
    const void* q;
    const void* k;
    const void* vt;
    void* out;
    int32_t batch, heads, head_dim;
    int32_t s, t;
    int32_t q_ld, k_ld, vt_ld, o_ld;
    float scale;
    int32_t causal;
    int32_t q_prescaled; /* != 0: q already carries the factor scale * log2(e) (folded into the weights of the projection that
                            produced it, minsdtf_amd/models.py), so the kernel takes exp2 of q k^T directly; `scale` is then
                            not applied again.  0: softmax(scale * q k^T) as written above. */
    /* ABI 9 - head_dim 512 only (the VAE's single-head AttentionBlock, layers.py:28-59): scratch for a 4-way split of the KEY walk,
     * >= 4 * batch * heads * s * (512 + 2) floats, 16-byte aligned, caller-owned, or NULL.  With it (and t >= 2048, t % 128 == 0) every
     * query tile is served by four workgroups that each walk a quarter of the keys and a merge launch adds their partial results in
     * part order (the one head at 512x512 is otherwise 64 workgroups on a 256-CU chip).  Whether the split runs depends on t and on
     * this pointer only, never on the batch; results with and without it differ in rounding. */
    float* workspace;
    int64_t workspace_floats;
} MsdAttention;

MSD_API int msd_attention(const MsdAttention* p, msd_stream_t stream);

/* msd_cross_attention_q — cross-attention over a short context with its query projection inside: the `attn2.to_q` Dense
 * (with the LayerNormalization in front of it folded in, as MsdConvGemm.ln_in does) and the attention over the text
 * context (diffusion_model.py:102-127, context = 77 tokens) as ONE launch instead of two latency-bound ones.
 *   q = LN(x) Wq^T computed per (64 queries, head) as rstd * (x wq^T - mean * ln_colsum) + bias, rounded to bf16;
 *   out = softmax(q k^T) v with the softmax as exp2 of the raw product: wq MUST carry scale * log2(e) (q_prescaled form).
 * x: bf16 [batch*s][c] (the block's raw rows), ln_in: float2 [batch*s][ln_in_slots] (the producer's ln_out partials),
 * wq: bf16 gamma-folded weights [c][c] in `w_layout`, ln_colsum / bias: [c]; k: bf16 [batch][t][k_ld], vt: bf16
 * [batch][heads][head_dim][vt_ld] (key contiguous), out: bf16 [batch*s][o_ld].  heads = 8, head_dim 40, 80 or 160, t <= 96
 * (160: the head's 400 KB of weights are streamed through a four-stage LDS ring instead of waiting in LDS). */
typedef struct MsdCrossAttnQ {
    const void* x;
    const float* ln_in;
    const void* wq;
    const float* ln_colsum;
    const float* bias;       /* [c] or NULL */
    const void* k;
    const void* vt;
    void* out;
    int32_t batch, heads, head_dim, s, t;
    int32_t k_ld, vt_ld, o_ld;
    int32_t ln_in_slots;
    float ln_eps;
    int32_t w_layout;        /* 0: [c][c] rows, 1: chunk-major [c/64][c][64] (MsdConvGemm.w_layout) */
} MsdCrossAttnQ;

MSD_API int msd_cross_attention_q(const MsdCrossAttnQ* p, msd_stream_t stream);

/* msd_softmax_rows — out[r, :cols] = softmax(scale * x[r, :cols]); x fp32 [rows][ld_in], out bf16
 * [rows][ld_out] (VAE single-head attention, layers.py:48-50). cols % 8 == 0. */
MSD_API int msd_softmax_rows(const float* x, void* out, int64_t rows, int32_t cols, int32_t ld_in, int32_t ld_out, float scale,
                     msd_stream_t stream);

/* msd_embedding_sum — out[r, :] = bf16(tok_table[tokens[r], :] + pos_table[positions[r], :]).
 * Replaces CLIPEmbedding.call (text_encoder.py:22-33): two Embedding lookups and their sum.
 * tables fp32 [vocab][dim] / [max_len][dim], tokens / positions int32 [rows], dim % 4 == 0; ids outside
 * their table are an argument error reported through `status` (device int32, set to 1; may be NULL). */
MSD_API int msd_embedding_sum(const int32_t* tokens, const int32_t* positions, const float* tok_table, const float* pos_table,
                      void* out, int32_t rows, int32_t dim, int32_t vocab, int32_t max_len, int32_t* status,
                      msd_stream_t stream);

/* msd_replicate (ABI 11) — dst = `copies` replicas of the `bytes` bytes at src, back to back (bytes % 16 == 0; src == dst: replica 0
 * stays where it is; any other overlap: MSD_E_ARG).  The classifier-free-guidance pair of one step (stable_diffusion.py:454-457: the
 * unconditional and the conditioned call get the SAME latent and time embedding) is identical up to the first cross-attention
 * (diffusion_model.py:191-196: conv_in, down_blocks.0.resnets.0, and norm / proj_in / attn1 of down_blocks.0.attentions.0); the fused
 * batch computes that prefix once per image and replicates three tensors (engine.SHARE_CFG_PREFIX). */
MSD_API int msd_replicate(const void* src, void* dst, int64_t bytes, int32_t copies, msd_stream_t stream);

/* msd_memset_zero — stream-ordered hipMemsetAsync(ptr, 0, bytes) (GroupNorm statistic slots). */
MSD_API int msd_memset_zero(void* ptr, int64_t bytes, msd_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * msd_cfg_step — classifier-free guidance + guidance rescale + sampler step, fp32, one launch.
 * Replaces stable_diffusion.py:458-461 (CFG, rescale_noise_cfg :304-315) and Scheduler.step
 * non-TCD branch (scheduler.py:272-285,308-312).
 *   eps:    fp32 [2*batch][n]  rows [0,batch) = unconditional, [batch,2*batch) = text-conditioned
 *           (guidance <= 0: fp32 [batch][n], used as is — stable_diffusion.py:462-467)
 *   latent: fp32 [batch][n], updated in place
 *   coef:   fp32 [steps][4] = {signal_rate[t], noise_rate[t], A, B}: x0 = (x - noise_rate eps) / signal_rate,
 *           x' = A x0 + B eps.  Deterministic sampler: A, B = signal / noise rate of t_prev and {1, 0} on
 *           the last step (x' = x0).  TCD sampler (scheduler.py:286-307): the gamma-sampling coefficients,
 *           plus x' += noise_coef[step] * step_noise[step][b][:] when step_noise != NULL
 *           (step_noise fp32 [steps][batch][n] = the per-step N(0,1) draws, noise_coef fp32 [steps]).
 *   The row used is coef[*step_ptr]; after the update *step_ptr is incremented when advance != 0:
 *   advance = 1 by a second one-thread launch; advance = 2 inside the launch, by the workgroup that finishes last — then
 *   step_ptr points at TWO ints {step, ticket}, ticket zero when first used and touched by nothing else.
 *   Inpainting (stable_diffusion.py:469-475), when inpaint_mask != NULL: after the sampler step
 *   latent = origin * (1 - mask) + latent * mask with origin = signal_rate[t] * inpaint_init +
 *   noise_rate[t] * inpaint_noise at the CURRENT timestep t (the reference re-noises the encoded
 *   image at t, not t_prev).  inpaint_init fp32 [n] (one image, shared by the batch), inpaint_noise
 *   fp32 [batch][n], inpaint_mask fp32 [n] (the latent-resolution mask repeated over the channels).
 */
typedef struct MsdCfgStep {
    const float* eps;
    float* latent;
    const float* coef;
    int32_t* step_ptr;
    int32_t batch, n, num_steps;
    float guidance, guidance_rescale;
    int32_t advance;
    const float* inpaint_init;
    const float* inpaint_noise;
    const float* inpaint_mask;
    const float* step_noise;
    const float* noise_coef;
} MsdCfgStep;

MSD_API int msd_cfg_step(const MsdCfgStep* p, msd_stream_t stream);

/* msd_add_bf16 — out = a + b elementwise on bf16. n % 8 == 0. */
MSD_API int msd_add_bf16(const void* a, const void* b, void* out, int64_t n, msd_stream_t stream);
/* msd_add_f32_bf16 — out = bf16(a + b), a / out bf16, b fp32, summed in fp32 (may run in place, out == a).  The ControlNet
 * residual adds of diffusion_model.py:230-234 when the 13 residuals arrive over the model boundary as fp32 arrays
 * (DiffusionModel.predict_on_batch); the fused device loop has no such launch: there the adds are the epilogue residual of
 * the ControlNet's own 1x1 "zero" convs (msd_conv_gemm with residual == out). n % 8 == 0. */
MSD_API int msd_add_f32_bf16(const void* a, const float* b, void* out, int64_t n, msd_stream_t stream);

/* msd_cast_f32_to_bf16 / msd_cast_bf16_to_f32 — dtype conversion of a contiguous buffer. */
MSD_API int msd_cast_f32_to_bf16(const float* in, void* out, int64_t n, msd_stream_t stream);
MSD_API int msd_cast_bf16_to_f32(const void* in, float* out, int64_t n, msd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MINSDTF_HIP_H */

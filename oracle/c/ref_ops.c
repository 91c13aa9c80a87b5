/* ORACLE — TEST INFRASTRUCTURE ONLY (never linked or loaded by the product path).
 *
 * Plain-C restatement of the primitive ops the reference's hot path invokes through Keras, written
 * as direct NHWC loops with double accumulation.  It is the second, independent implementation the
 * torch-based oracle (oracle/sd_oracle.py) is cross-checked against: the two
 * share no code, so agreement pins the op semantics restated from the Keras-3 documentation.
 * "parity unpinned" w.r.t. Keras itself — see oracle/sd_oracle.py.
 *
 * Reference call sites (paths relative to /root/reference/stable_diffusion/):
 *   ref_conv2d_nhwc     layers.py:17-25 (ZeroPadding2D + VALID Conv2D, HWIO kernel)
 *   ref_upsample2_nhwc  diffusion_model.py:135, image_decoder.py:36 (UpSampling2D(2), nearest)
 *   ref_group_norm      diffusion_model.py:27 (GroupNormalization(groups=32, eps=1e-5))
 *   ref_layer_norm      diffusion_model.py:84 (LayerNormalization(eps=1e-5))
 *   ref_dense           diffusion_model.py:102-108 (x @ W(in,out) + b)
 *   ref_swish           layers.Activation("swish")
 *   ref_geglu           diffusion_model.py:148-153 (tanh approximation)
 *   ref_attention       diffusion_model.py:110-127 (scale after QK^T, softmax over keys)
 *   ref_timestep_embedding  stable_diffusion.py:543-553
 *   ref_cfg_rescale     stable_diffusion.py:458, 304-315
 *   ref_sched_step      scheduler.py:285, 308-312
 */
#include <math.h>
#include <stdlib.h>

void ref_conv2d_nhwc(const float* x, const float* w, const float* b, float* y, int B, int H, int W, int Cin, int Cout, int ks,
                     int stride, int pad) {
    int Ho = (H + 2 * pad - ks) / stride + 1, Wo = (W + 2 * pad - ks) / stride + 1;
    for (int n = 0; n < B; ++n)
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox)
                for (int co = 0; co < Cout; ++co) {
                    double acc = b ? b[co] : 0.0;
                    for (int ky = 0; ky < ks; ++ky) {
                        int iy = oy * stride + ky - pad;
                        if (iy < 0 || iy >= H) continue;
                        for (int kx = 0; kx < ks; ++kx) {
                            int ix = ox * stride + kx - pad;
                            if (ix < 0 || ix >= W) continue;
                            const float* xp = x + (((size_t)n * H + iy) * W + ix) * Cin;
                            const float* wp = w + ((size_t)(ky * ks + kx) * Cin) * Cout + co;
                            for (int ci = 0; ci < Cin; ++ci) acc += (double)xp[ci] * wp[(size_t)ci * Cout];
                        }
                    }
                    y[(((size_t)n * Ho + oy) * Wo + ox) * Cout + co] = (float)acc;
                }
}

void ref_upsample2_nhwc(const float* x, float* y, int B, int H, int W, int C) {
    for (int n = 0; n < B; ++n)
        for (int oy = 0; oy < 2 * H; ++oy)
            for (int ox = 0; ox < 2 * W; ++ox)
                for (int c = 0; c < C; ++c)
                    y[(((size_t)n * 2 * H + oy) * 2 * W + ox) * C + c] = x[(((size_t)n * H + oy / 2) * W + ox / 2) * C + c];
}

void ref_group_norm(const float* x, const float* gamma, const float* beta, float* y, int B, int HW, int C, int groups, float eps) {
    int cpg = C / groups;
    for (int n = 0; n < B; ++n)
        for (int g = 0; g < groups; ++g) {
            double s = 0, ss = 0;
            for (int p = 0; p < HW; ++p)
                for (int c = g * cpg; c < (g + 1) * cpg; ++c) s += x[((size_t)n * HW + p) * C + c];
            double mean = s / ((double)HW * cpg);
            for (int p = 0; p < HW; ++p)
                for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
                    double d = x[((size_t)n * HW + p) * C + c] - mean;
                    ss += d * d;
                }
            double rstd = 1.0 / sqrt(ss / ((double)HW * cpg) + eps);  /* biased variance */
            for (int p = 0; p < HW; ++p)
                for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
                    size_t i = ((size_t)n * HW + p) * C + c;
                    y[i] = (float)((x[i] - mean) * rstd * gamma[c] + beta[c]);
                }
        }
}

void ref_layer_norm(const float* x, const float* gamma, const float* beta, float* y, int rows, int C, float eps) {
    for (int r = 0; r < rows; ++r) {
        double s = 0, ss = 0;
        for (int c = 0; c < C; ++c) s += x[(size_t)r * C + c];
        double mean = s / C;
        for (int c = 0; c < C; ++c) {
            double d = x[(size_t)r * C + c] - mean;
            ss += d * d;
        }
        double rstd = 1.0 / sqrt(ss / C + eps);
        for (int c = 0; c < C; ++c) y[(size_t)r * C + c] = (float)((x[(size_t)r * C + c] - mean) * rstd * gamma[c] + beta[c]);
    }
}

void ref_dense(const float* x, const float* w, const float* b, float* y, int rows, int Cin, int Cout) {
    for (int r = 0; r < rows; ++r)
        for (int o = 0; o < Cout; ++o) {
            double acc = b ? b[o] : 0.0;
            for (int i = 0; i < Cin; ++i) acc += (double)x[(size_t)r * Cin + i] * w[(size_t)i * Cout + o];
            y[(size_t)r * Cout + o] = (float)acc;
        }
}

void ref_swish(const float* x, float* y, long n) {
    for (long i = 0; i < n; ++i) y[i] = (float)(x[i] / (1.0 + exp(-(double)x[i])));
}

/* h: [rows][2*n] = value | gate  ->  y: [rows][n] */
void ref_geglu(const float* h, float* y, int rows, int n) {
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < n; ++c) {
            double a = h[(size_t)r * 2 * n + c], g = h[(size_t)r * 2 * n + n + c];
            double t = tanh(g * 0.7978845608 * (1.0 + 0.044715 * g * g));
            y[(size_t)r * n + c] = (float)(a * 0.5 * g * (1.0 + t));
        }
}

/* q: [B][S][heads*d], k, v: [B][T][heads*d] -> o: [B][S][heads*d]; softmax(scale * q k^T) v */
void ref_attention(const float* q, const float* k, const float* v, float* o, int B, int S, int T, int heads, int d, float scale) {
    int C = heads * d;
    double* p = (double*)malloc(sizeof(double) * T);
    for (int n = 0; n < B; ++n)
        for (int h = 0; h < heads; ++h)
            for (int s = 0; s < S; ++s) {
                const float* qp = q + ((size_t)n * S + s) * C + h * d;
                double mx = -1e300;
                for (int t = 0; t < T; ++t) {
                    const float* kp = k + ((size_t)n * T + t) * C + h * d;
                    double acc = 0;
                    for (int i = 0; i < d; ++i) acc += (double)qp[i] * kp[i];
                    p[t] = acc * scale;
                    if (p[t] > mx) mx = p[t];
                }
                double sum = 0;
                for (int t = 0; t < T; ++t) { p[t] = exp(p[t] - mx); sum += p[t]; }
                for (int i = 0; i < d; ++i) {
                    double acc = 0;
                    for (int t = 0; t < T; ++t) acc += p[t] * v[((size_t)n * T + t) * C + h * d + i];
                    o[((size_t)n * S + s) * C + h * d + i] = (float)(acc / sum);
                }
            }
    free(p);
}

/* [cos | sin] of t * exp(-ln(max_period) * i / half); the reference builds freqs and args in float32 */
void ref_timestep_embedding(int timestep, float* out, int dim, float max_period) {
    int half = dim / 2;
    for (int i = 0; i < half; ++i) {
        float freq = (float)exp((double)(-logf(max_period) * (float)i / (float)half));
        float arg = (float)timestep * freq;
        out[i] = (float)cos((double)arg);
        out[half + i] = (float)sin((double)arg);
    }
}

/* u, c: [B][n]; out = phi * cfg * std(c)/(std(cfg)+1e-5) + (1-phi) * cfg with cfg = u + g (c - u); population std per sample */
void ref_cfg_rescale(const float* u, const float* c, float* out, int B, int n, float g, float phi) {
    for (int b = 0; b < B; ++b) {
        const float* ub = u + (size_t)b * n;
        const float* cb = c + (size_t)b * n;
        float* ob = out + (size_t)b * n;
        double sc = 0, sg = 0;
        for (int i = 0; i < n; ++i) { ob[i] = ub[i] + g * (cb[i] - ub[i]); sc += cb[i]; sg += ob[i]; }
        if (phi > 0.f) {
            double mc = sc / n, mg = sg / n, vc = 0, vg = 0;
            for (int i = 0; i < n; ++i) { vc += (cb[i] - mc) * (cb[i] - mc); vg += (ob[i] - mg) * (ob[i] - mg); }
            double f = phi * (sqrt(vc / n) / (sqrt(vg / n) + 1e-5)) + (1.0 - phi);
            for (int i = 0; i < n; ++i) ob[i] = (float)(ob[i] * f);
        }
    }
}

/* x0 = (x - nr*eps)/sr ; out = last ? x0 : sr_prev*x0 + nr_prev*eps   (double, like the reference's f64 coefficients) */
void ref_sched_step(const double* x, const float* eps, double* out, long n, double sr, double nr, double sr_prev, double nr_prev, int last) {
    for (long i = 0; i < n; ++i) {
        double x0 = (x[i] - nr * eps[i]) / sr;
        out[i] = last ? x0 : sr_prev * x0 + nr_prev * eps[i];
    }
}

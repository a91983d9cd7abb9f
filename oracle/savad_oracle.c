/*
 * savad_oracle.c -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object; the product path (voice_activity_detection_amd/, libsavad.so) never does.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle_golden.py) against
 * golden vectors produced by running the reference itself in the build container
 * (tests/golden/make_golden.py imports /root/reference unmodified; the reference's own
 * tests hold no numeric fixtures for this path -- SURVEY.md section 4).
 *
 * Every function cites the reference lines it restates (paths relative to /root/reference).
 * Arithmetic: fp32 storage everywhere, like the reference.  acc64 = 0 accumulates dot
 * products / statistics in fp32 (k-sequential), acc64 = 1 in fp64 ("truth" mode used to
 * measure the fp32 noise floor).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define SAVAD_ORACLE_API __attribute__((visibility("default")))

/* ---- a3: SinusoidalPositionalEncoding.build_positional_encoding, vad/modeling/transformer.py:403-414
 * pe[t,2i] = sin(t * w_i), pe[t,2i+1] = cos(t * w_i), w_i = exp(float32(2i) * float32(-(ln 1e4)/D)),
 * all in fp32 (position * div_term is an fp32 product).  The caller divides by sqrt(D) (:401). */
SAVAD_ORACLE_API void savad_oracle_pe(int T, int D, float* pe /* [T][D] */) {
    const float c = (float)(-(log(10000.0) / (double)D));
    for (int i = 0; i < D / 2; ++i) {
        const float arg = (float)(2 * i) * c;
        const float w = (float)exp((double)arg); /* correctly-rounded fp32 exp */
        for (int t = 0; t < T; ++t) {
            const float a = (float)t * w;
            pe[(size_t)t * D + 2 * i] = (float)sin((double)a);
            pe[(size_t)t * D + 2 * i + 1] = (float)cos((double)a);
        }
    }
}

/* y[T][N] = x[T][K] . W[N][K]^T + b   (nn.Linear; W is [out,in] row-major) */
static void linear(int T, int K, int N, const float* x, const float* W, const float* b, float* y, int acc64) {
    if (acc64) {
        for (int t = 0; t < T; ++t)
            for (int n = 0; n < N; ++n) {
                double s = 0.0;
                for (int k = 0; k < K; ++k) s += (double)x[(size_t)t * K + k] * (double)W[(size_t)n * K + k];
                y[(size_t)t * N + n] = (float)(s + (double)b[n]);
            }
        return;
    }
    /* fp32: W is handed over PRE-TRANSPOSED ([K][N], see transpose_params) so that the i-k-j loop
     * vectorises over n without reassociating the k-sequential sum */
    const float* Wt = W;
    for (int t = 0; t < T; ++t) {
        float* yr = y + (size_t)t * N;
        for (int n = 0; n < N; ++n) yr[n] = 0.0f;
        for (int k = 0; k < K; ++k) {
            const float xv = x[(size_t)t * K + k];
            const float* wr = Wt + (size_t)k * N;
            for (int n = 0; n < N; ++n) yr[n] += xv * wr[n];
        }
        for (int n = 0; n < N; ++n) yr[n] += b[n];
    }
}

static float* transposed(const float* W, int N, int K) {
    float* Wt = (float*)malloc(sizeof(float) * (size_t)K * N);
    if (!Wt) return NULL;
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) Wt[(size_t)k * N + n] = W[(size_t)n * K + k];
    return Wt;
}

/* nn.LayerNorm(D): eps = 1e-5, biased variance, affine (vad/modeling/transformer.py:22,231) */
static void layer_norm(int T, int D, const float* x, const float* g, const float* b, float* y, int acc64) {
    for (int t = 0; t < T; ++t) {
        const float* xr = x + (size_t)t * D;
        float* yr = y + (size_t)t * D;
        if (acc64) {
            double m = 0, v = 0;
            for (int d = 0; d < D; ++d) m += xr[d];
            m /= D;
            for (int d = 0; d < D; ++d) v += (xr[d] - m) * (xr[d] - m);
            v /= D;
            const double r = 1.0 / sqrt(v + 1e-5);
            for (int d = 0; d < D; ++d) yr[d] = (float)((xr[d] - m) * r * g[d] + b[d]);
        } else {
            float m = 0, v = 0;
            for (int d = 0; d < D; ++d) m += xr[d];
            m /= (float)D;
            for (int d = 0; d < D; ++d) v += (xr[d] - m) * (xr[d] - m);
            v /= (float)D;
            const float r = 1.0f / sqrtf(v + 1e-5f);
            for (int d = 0; d < D; ++d) yr[d] = (xr[d] - m) * r * g[d] + b[d];
        }
    }
}

/* Parameter order = voice_activity_detection_amd/seeded.py:state_dict_spec
 *   [0] input_layer.0.weight [D,F]  [1] input_layer.0.bias
 *   per layer l (base 2+16l): q.w q.b k.w k.b v.w v.b final.w final.b ln1.w ln1.b ff0.w ff0.b ff3.w ff3.b ln2.w ln2.b
 *   then encoder.layer_norm.w/.b, classifier.w [2,D], classifier.b                            */
typedef struct {
    float *h, *n, *q, *k, *v, *ctx, *o, *ff, *s, *kt;
} scratch_t;

static void forward_one(const float* const* P, const float* x, int T, int F, int L, int D, const float* pe,
                        float* out, int acc64, scratch_t* w, float* tap_input, float* tap_ctx0, float* tap_enc) {
    const int DFF = 4 * D; /* vad/models/self_attention.py:10 */
    /* a2 + a3: input Linear (self_attention.py:13), x + pe[:T]/sqrt(D) (transformer.py:401); dropout = identity */
    linear(T, F, D, x, P[0], P[1], w->h, acc64);
    const float scale = (float)sqrt((double)D); /* transformer.py:389 */
    for (size_t i = 0; i < (size_t)T * D; ++i) w->h[i] = w->h[i] + pe[i] / scale;
    if (tap_input) memcpy(tap_input, w->h, sizeof(float) * (size_t)T * D);

    const float dh = (float)sqrt((double)D); /* np.sqrt(d_head), n_heads = 1: transformer.py:362, self_attention.py:18 */
    for (int l = 0; l < L; ++l) {
        const float* const* Q = P + 2 + 16 * l;
        /* a5: pre-LN sublayer (transformer.py:234-238) around a6..a10 */
        layer_norm(T, D, w->h, Q[8], Q[9], w->n, acc64);
        linear(T, D, D, w->n, Q[0], Q[1], w->q, acc64); /* transformer.py:281 */
        linear(T, D, D, w->n, Q[2], Q[3], w->k, acc64); /* :283 */
        linear(T, D, D, w->n, Q[4], Q[5], w->v, acc64); /* :284 */
        /* a7: scores = q k^T / sqrt(d_head) (transformer.py:351-363); a8: softmax over keys (:333) */
        for (int j = 0; j < T; ++j)
            for (int d = 0; d < D; ++d) w->kt[(size_t)d * T + j] = w->k[(size_t)j * D + d];
        for (int i = 0; i < T; ++i) {
            float* s = w->s;
            if (acc64) {
                for (int j = 0; j < T; ++j) {
                    double a = 0;
                    for (int d = 0; d < D; ++d) a += (double)w->q[(size_t)i * D + d] * (double)w->k[(size_t)j * D + d];
                    s[j] = (float)(a / sqrt((double)D));
                }
            } else {
                for (int j = 0; j < T; ++j) s[j] = 0.0f;
                for (int d = 0; d < D; ++d) {
                    const float qv = w->q[(size_t)i * D + d];
                    const float* kr = w->kt + (size_t)d * T;
                    for (int j = 0; j < T; ++j) s[j] += qv * kr[j];
                }
                for (int j = 0; j < T; ++j) s[j] = s[j] / dh;
            }
            float mx = s[0];
            for (int j = 1; j < T; ++j) mx = s[j] > mx ? s[j] : mx;
            float* c = w->ctx + (size_t)i * D;
            if (acc64) {
                double den = 0;
                double accd[1024];
                for (int d = 0; d < D; ++d) accd[d] = 0;
                for (int j = 0; j < T; ++j) {
                    const double e = exp((double)s[j] - (double)mx);
                    den += e;
                    for (int d = 0; d < D; ++d) accd[d] += e * (double)w->v[(size_t)j * D + d];
                }
                for (int d = 0; d < D; ++d) c[d] = (float)(accd[d] / den);
            } else {
                float den = 0;
                for (int j = 0; j < T; ++j) {
                    s[j] = expf(s[j] - mx);
                    den += s[j];
                }
                for (int j = 0; j < T; ++j) s[j] = s[j] / den; /* attention probabilities */
                /* a9: ctx = A . V (transformer.py:338-346) */
                for (int d = 0; d < D; ++d) c[d] = 0.0f;
                for (int j = 0; j < T; ++j) {
                    const float a = s[j];
                    const float* vr = w->v + (size_t)j * D;
                    for (int d = 0; d < D; ++d) c[d] += a * vr[d];
                }
            }
        }
        if (l == 0 && tap_ctx0) memcpy(tap_ctx0, w->ctx, sizeof(float) * (size_t)T * D);
        /* a10: final_projection (:347) + residual onto the un-normalised x (:237) */
        linear(T, D, D, w->ctx, Q[6], Q[7], w->o, acc64);
        for (size_t i = 0; i < (size_t)T * D; ++i) w->h[i] = w->o[i] + w->h[i];
        /* a11: FFN sublayer (transformer.py:366-382) */
        layer_norm(T, D, w->h, Q[14], Q[15], w->n, acc64);
        linear(T, D, DFF, w->n, Q[10], Q[11], w->ff, acc64);
        for (size_t i = 0; i < (size_t)T * DFF; ++i) w->ff[i] = w->ff[i] > 0.0f ? w->ff[i] : 0.0f;
        linear(T, DFF, D, w->ff, Q[12], Q[13], w->o, acc64);
        for (size_t i = 0; i < (size_t)T * D; ++i) w->h[i] = w->o[i] + w->h[i];
    }
    /* a4: final encoder LayerNorm (transformer.py:22,33) */
    const float* const* Z = P + 2 + 16 * L;
    layer_norm(T, D, w->h, Z[0], Z[1], w->n, acc64);
    if (tap_enc) memcpy(tap_enc, w->n, sizeof(float) * (size_t)T * D);
    /* a12: classifier Linear(D,2) + LogSoftmax(dim=2) (self_attention.py:20-21,26-27) */
    for (int t = 0; t < T; ++t) {
        double z[2];
        for (int c = 0; c < 2; ++c) {
            if (acc64) {
                double a = 0;
                for (int d = 0; d < D; ++d) a += (double)w->n[(size_t)t * D + d] * (double)Z[2][(size_t)c * D + d];
                z[c] = (double)(float)(a + (double)Z[3][c]);
            } else {
                float a = 0;
                for (int d = 0; d < D; ++d) a += w->n[(size_t)t * D + d] * Z[2][(size_t)c * D + d];
                z[c] = (double)(a + Z[3][c]);
            }
        }
        if (acc64) {
            const double m = z[0] > z[1] ? z[0] : z[1];
            const double lse = m + log(exp(z[0] - m) + exp(z[1] - m));
            out[(size_t)t * 2 + 0] = (float)(z[0] - lse);
            out[(size_t)t * 2 + 1] = (float)(z[1] - lse);
        } else {
            const float z0 = (float)z[0], z1 = (float)z[1];
            const float m = z0 > z1 ? z0 : z1;
            const float lse = m + logf(expf(z0 - m) + expf(z1 - m));
            out[(size_t)t * 2 + 0] = z0 - lse;
            out[(size_t)t * 2 + 1] = z1 - lse;
        }
    }
}

static int scratch_alloc(scratch_t* w, int T, int D) {
    const size_t td = (size_t)T * D;
    float* base = (float*)malloc(sizeof(float) * (8 * td + 4 * td + (size_t)T + td));
    if (!base) return -1;
    w->h = base;
    w->n = base + td;
    w->q = base + 2 * td;
    w->k = base + 3 * td;
    w->v = base + 4 * td;
    w->ctx = base + 5 * td;
    w->o = base + 6 * td;
    w->kt = base + 7 * td;
    w->ff = base + 8 * td; /* 4*td */
    w->s = base + 12 * td; /* T */
    return 0;
}

/* a1: SelfAttentiveVAD.forward (vad/models/self_attention.py:23-28) on x[B][T][F] -> log-probs out[B][T][2].
 * taps (may be NULL): post input_layer [B][T][D], layer-0 attention context [B][T][D], encoder output [B][T][D].
 * threads <= 0: all OpenMP threads.  Returns 0, or -1 on allocation failure / unsupported D. */
SAVAD_ORACLE_API int savad_oracle_forward(const float* const* params, const float* x, int B, int T, int F, int L, int D,
                                          float* out, int acc64, int threads, float* tap_input, float* tap_ctx0,
                                          float* tap_enc) {
    if (D > 1024 || D % 2) return -1;
    if (B == 0 || T == 0) return 0;
    float* pe = (float*)malloc(sizeof(float) * (size_t)T * D);
    if (!pe) return -1;
    savad_oracle_pe(T, D, pe);
    int fail = 0;
    const int np = 2 + 16 * L + 4;
    const float** P = (const float**)malloc(sizeof(float*) * np);
    float** owned = (float**)calloc(np, sizeof(float*));
    for (int i = 0; i < np; ++i) P[i] = params[i];
    if (!acc64) { /* fp32 mode: linear() wants [K][N] weights */
        owned[0] = transposed(params[0], D, F);
        for (int l = 0; l < L; ++l) {
            const int b0 = 2 + 16 * l;
            for (int j = 0; j < 8; j += 2) owned[b0 + j] = transposed(params[b0 + j], D, D);
            owned[b0 + 10] = transposed(params[b0 + 10], 4 * D, D);
            owned[b0 + 12] = transposed(params[b0 + 12], D, 4 * D);
        }
        for (int i = 0; i < np; ++i)
            if (owned[i]) P[i] = owned[i];
    }
    params = P;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        scratch_t w;
        int ok = scratch_alloc(&w, T, D) == 0;
        if (!ok) {
#pragma omp atomic write
            fail = 1;
        }
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < B; ++b) {
            if (!ok) continue;
            const size_t td = (size_t)T * D;
            forward_one(params, x + (size_t)b * T * F, T, F, L, D, pe, out + (size_t)b * T * 2, acc64, &w,
                        tap_input ? tap_input + b * td : NULL, tap_ctx0 ? tap_ctx0 + b * td : NULL,
                        tap_enc ? tap_enc + b * td : NULL);
        }
        if (ok) free(w.h);
    }
    free(pe);
    for (int i = 0; i < np; ++i) free(owned[i]);
    free(owned);
    free((void*)P);
    return fail ? -1 : 0;
}

/* Window length W = 2*(half-1)/jump + 3 (vad/predictor.py:57-59) and the relative offsets
 * arange(-half, 0, jump) ++ [0] ++ arange(1, half+1, jump) (vad/predictor.py:186-212). Returns W. */
SAVAD_ORACLE_API int savad_oracle_window_offsets(int half, int jump, int* offsets /* >= W ints, may be NULL */) {
    int w = 0;
    for (int o = -half; o < 0; o += jump) {
        if (offsets) offsets[w] = o;
        ++w;
    }
    if (offsets) offsets[w] = 0;
    ++w;
    for (int o = 1; o < half + 1; o += jump) {
        if (offsets) offsets[w] = o;
        ++w;
    }
    return w;
}

/* a13: window gather (vad/predictor.py:180-220).  feature[N][F]; item i in [first, first+count) has centre
 * half+i; windows[i-first][w][:] = feature[half + i + off[w]], positions[i-first][w] = half + i + off[w]. */
SAVAD_ORACLE_API void savad_oracle_gather_windows(const float* feature, int N, int F, int half, int jump, int first,
                                                  int count, float* windows, int64_t* positions) {
    int off[64];
    const int W = savad_oracle_window_offsets(half, jump, off);
    (void)N;
    for (int i = 0; i < count; ++i)
        for (int w = 0; w < W; ++w) {
            const int pos = half + first + i + off[w];
            memcpy(windows + ((size_t)i * W + w) * F, feature + (size_t)pos * F, sizeof(float) * F);
            positions[(size_t)i * W + w] = pos;
        }
}

/* a14: boosted prediction (vad/predictor.py:238-258 and :95).
 * boosted[N][W][2] = 0; boosted[pos[b][w]][w] = logp[b][w]; probs = softmax(boosted, axis=2)[:,:,1]
 * (unfilled slots stay [0,0] -> exactly 0.5 and ARE averaged in); mean = probs.mean(axis=1). */
SAVAD_ORACLE_API void savad_oracle_boost(const float* logp, const int64_t* positions, int count, int N, int W,
                                         float* probs /* [N][W] */, float* mean /* [N] */) {
    float* boosted = (float*)calloc((size_t)N * W * 2, sizeof(float));
    for (int b = 0; b < count; ++b)
        for (int w = 0; w < W; ++w) {
            const int64_t p = positions[(size_t)b * W + w];
            boosted[((size_t)p * W + w) * 2 + 0] = logp[((size_t)b * W + w) * 2 + 0];
            boosted[((size_t)p * W + w) * 2 + 1] = logp[((size_t)b * W + w) * 2 + 1];
        }
    for (int n = 0; n < N; ++n) {
        float acc = 0.0f; /* numpy float32 mean over 7 values: pairwise == sequential here */
        for (int w = 0; w < W; ++w) {
            /* scipy.special.softmax on float32: exp(x - max) / sum */
            const float a = boosted[((size_t)n * W + w) * 2 + 0], c = boosted[((size_t)n * W + w) * 2 + 1];
            const float m = a > c ? a : c;
            const float ea = expf(a - m), ec = expf(c - m);
            const float p = ec / (ea + ec);
            probs[(size_t)n * W + w] = p;
            acc += p;
        }
        if (mean) mean[n] = acc / (float)W;
    }
    free(boosted);
}

/* Streaming long-form mode (BASELINE.json configs[4]) -- the build's own definition (include/savad.h),
 * not a reference mode: windows [hop*w, hop*w+T) zero-padded past N; probs[n] = mean over covering
 * windows of softmax(logp[w][n-hop*w])[1]. */
SAVAD_ORACLE_API int savad_oracle_stream_window_count(int N, int T, int hop) { return N <= T ? 1 : (N - T + hop - 1) / hop + 1; }

SAVAD_ORACLE_API void savad_oracle_gather_strided(const float* feature, int N, int F, int T, int hop, int first, int count,
                                                  float* windows) {
    for (int w = 0; w < count; ++w)
        for (int t = 0; t < T; ++t) {
            const long frame = (long)hop * (first + w) + t;
            float* dst = windows + ((size_t)w * T + t) * F;
            if (frame < N)
                memcpy(dst, feature + (size_t)frame * F, sizeof(float) * F);
            else
                memset(dst, 0, sizeof(float) * F);
        }
}

SAVAD_ORACLE_API void savad_oracle_overlap_merge(const float* logp, int W, int N, int T, int hop, float* probs) {
    for (int n = 0; n < N; ++n) {
        int w_hi = n / hop;
        if (w_hi > W - 1) w_hi = W - 1;
        float acc = 0.0f;
        int cnt = 0;
        for (int w = w_hi; w >= 0 && n - hop * w < T; --w) {
            const float a = logp[((size_t)w * T + (n - hop * w)) * 2], c = logp[((size_t)w * T + (n - hop * w)) * 2 + 1];
            const float m = a > c ? a : c;
            const float ea = expf(a - m), ec = expf(c - m);
            acc += ec / (ea + ec);
            ++cnt;
        }
        probs[n] = cnt ? acc / (float)cnt : 0.5f;
    }
}

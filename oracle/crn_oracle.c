/*
 * crn_oracle.c - CPU restatement of the reference's TemporalCRN streaming hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The shipped path is the HIP engine
 * (speech_enhancement_mi_amd/csrc); it never calls into this file and has no CPU fallback.
 *
 * Parity status: PINNED for everything inside TemporalCRN.forward / segmentation / over_add /
 * realtime_process / cal_si_snr by golden vectors produced by running the genuine reference
 * (tests/golden/make_golden.py -> tests/golden/crn_golden.npz; tests/test_oracle_golden.py).
 * The STFT/ISTFT arithmetic lives in speechbrain (un-vendored, unpinned, absent): "parity unpinned" at
 * that boundary; the restatement follows torch.stft/torch.istft (what speechbrain wraps) and is
 * checked against torch.stft/istft outputs of this container's torch.
 *
 * Plain C, fp32 arithmetic like the reference's CPU path (DFT twiddles/accumulation in double, result
 * rounded to float).  Layouts follow the reference (NCHW = [B,C,F,T]) so taps compare 1:1 with goldens.
 * Citations are file:line into /root/reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAXL 8
#define EPS 1e-8f /* CRN.py:11 */

typedef struct {
    int num_levels;
    int channels[MAXL];
    int num_freqs, hidden, num_layers, num_inputs, kernel_size;
    int n_fft, win, hop, segment_length;
    int variant; /* 0 = CRN.py (ReLU), 1 = CRN_ELU.py (ELU, gated 1x1, 3 preconv blocks, atan2), 2 = distillation_crn.py student
                    (as 1 but arctan phase and gLN denominator sqrt(var)+eps) */
} crn_cfg;

typedef struct {
    float *w, *b;   /* conv weight / bias */
    float *nw, *nb; /* norm affine */
    float *mw, *mb, *mnw, *mnb, *rw, *rb; /* decoder: residualmask, residualnorm, residual */
    float *tw, *tb, *gw, *gb;             /* CRN_ELU encoder/preconv: conv_trans, conv_gated (1x1) */
} layer_w;

typedef struct crn_oracle {
    crn_cfg c;
    int F[MAXL + 1];   /* freq size per encoder level: F[0]=num_freqs, F[i+1]=conv out */
    int Cin[MAXL + 1]; /* channels per level: Cin[0]=2M-1 */
    int T, D;
    layer_w enc[MAXL], dec[MAXL], pre[3];
    float *pbuf[3]; /* preconv time buffers [B,Cin0,F0,4] (CRN_ELU.py:335-340) */
    float *wih[4], *whh[4], *bih[4], *bhh[4], *fcw, *fcb, *gnw, *gnb;
    /* streaming state (CRN.py:319-337, 254/281) */
    int B;
    float *buf[MAXL]; /* [B,Cin,F,2d] */
    float *h;         /* [layers,B,H] */
    /* taps of the last forward() */
    float *tap_feat, *tap_enc[MAXL], *tap_gru, *tap_dec[MAXL], *tap_pre;
    double *tw_c, *tw_s; /* n_fft twiddles */
    float *window;       /* n_fft, hamming(win) centred */
    char err[256];
} crn_oracle;

static void *xcalloc(size_t n, size_t s) {
    void *p = calloc(n ? n : 1, s);
    if (!p) { fprintf(stderr, "crn_oracle: out of memory\n"); abort(); }
    return p;
}

const char *crn_oracle_error(crn_oracle *o) { return o->err; }

crn_oracle *crn_oracle_create(const crn_cfg *cfg) {
    crn_oracle *o = (crn_oracle *)xcalloc(1, sizeof(*o));
    o->c = *cfg;
    int L = cfg->num_levels;
    if (L < 1 || L > MAXL || cfg->num_layers > 4 || cfg->kernel_size != 3) { free(o); return NULL; }
    o->T = 1 + cfg->segment_length / cfg->hop; /* torch.stft center=True */
    o->F[0] = cfg->num_freqs;
    o->Cin[0] = 2 * cfg->num_inputs - 1; /* CRN.py:433 */
    for (int i = 0; i < L; i++) {
        o->F[i + 1] = (o->F[i] + 4 - 5) / 2 + 1; /* Conv2d k=5,s=2,p=2: CRN.py:435 */
        o->Cin[i + 1] = cfg->channels[i];
    }
    o->D = (cfg->num_freqs / 16 + 1) * cfg->channels[L - 1]; /* CRN.py:450 */
    if (o->D != o->F[L] * cfg->channels[L - 1]) {
        /* the reference's reshape (CRN.py:478) would fail too */
        free(o);
        return NULL;
    }
    int N = cfg->n_fft;
    o->tw_c = (double *)xcalloc(N, sizeof(double));
    o->tw_s = (double *)xcalloc(N, sizeof(double));
    for (int i = 0; i < N; i++) {
        o->tw_c[i] = cos(2.0 * M_PI * i / N);
        o->tw_s[i] = sin(2.0 * M_PI * i / N);
    }
    o->window = (float *)xcalloc(N, sizeof(float));
    int left = (N - cfg->win) / 2; /* torch.stft pads the window to n_fft, centred */
    for (int i = 0; i < cfg->win; i++) /* torch.hamming_window(periodic=True) */
        o->window[left + i] = (float)(0.54 - 0.46 * cos(2.0 * M_PI * i / cfg->win));
    return o;
}

static void free_state(crn_oracle *o) {
    for (int i = 0; i < MAXL; i++) {
        free(o->buf[i]); o->buf[i] = NULL;
        free(o->tap_enc[i]); o->tap_enc[i] = NULL;
        free(o->tap_dec[i]); o->tap_dec[i] = NULL;
    }
    for (int i = 0; i < 3; i++) { free(o->pbuf[i]); o->pbuf[i] = NULL; }
    free(o->tap_pre); o->tap_pre = NULL;
    free(o->h); o->h = NULL;
    free(o->tap_feat); o->tap_feat = NULL;
    free(o->tap_gru); o->tap_gru = NULL;
}

void crn_oracle_destroy(crn_oracle *o) {
    if (!o) return;
    free_state(o);
    for (int i = 0; i < MAXL; i++) {
        layer_w *ls[3] = {&o->enc[i], &o->dec[i], &o->pre[i < 3 ? i : 0]};
        for (int k = 0; k < (i < 3 ? 3 : 2); k++) {
            free(ls[k]->tw); free(ls[k]->tb); free(ls[k]->gw); free(ls[k]->gb);
            ls[k]->tw = ls[k]->tb = ls[k]->gw = ls[k]->gb = NULL;
            free(ls[k]->w); free(ls[k]->b); free(ls[k]->nw); free(ls[k]->nb); free(ls[k]->mw);
            free(ls[k]->mb); free(ls[k]->mnw); free(ls[k]->mnb); free(ls[k]->rw); free(ls[k]->rb);
        }
    }
    for (int i = 0; i < 4; i++) { free(o->wih[i]); free(o->whh[i]); free(o->bih[i]); free(o->bhh[i]); }
    free(o->fcw); free(o->fcb); free(o->gnw); free(o->gnb);
    free(o->tw_c); free(o->tw_s); free(o->window);
    free(o);
}

/* ---- parameter loading by reference checkpoint key (SURVEY.md 8b key list) ---- */
static int set_param(float **slot, const float *data, long n, long expect, crn_oracle *o, const char *key) {
    if (n != expect) {
        snprintf(o->err, sizeof(o->err), "shape mismatch for %s: got %ld elements, expected %ld", key, n, expect);
        return -2;
    }
    free(*slot);
    *slot = (float *)xcalloc(n, sizeof(float));
    memcpy(*slot, data, n * sizeof(float));
    return 0;
}

int crn_oracle_load(crn_oracle *o, const char *key, const float *data, const int64_t *shape, int ndim) {
    long n = 1;
    for (int i = 0; i < ndim; i++) n *= shape[i];
    int L = o->c.num_levels, H = o->c.hidden, idx;
    char rest[128];
    if (sscanf(key, "preconvlist.%d.%127s", &idx, rest) == 2 && idx >= 0 && idx < 3) {
        layer_w *l = &o->pre[idx];
        long c = o->Cin[0];
        if (!strcmp(rest, "conv.weight") || !strcmp(rest, "net.0.weight")) return set_param(&l->w, data, n, c * c * 25, o, key);
        if (!strcmp(rest, "conv.bias") || !strcmp(rest, "net.0.bias")) return set_param(&l->b, data, n, c, o, key);
        if (!strcmp(rest, "conv_trans.weight")) return set_param(&l->tw, data, n, c * c, o, key);
        if (!strcmp(rest, "conv_trans.bias")) return set_param(&l->tb, data, n, c, o, key);
        if (!strcmp(rest, "conv_gated.weight")) return set_param(&l->gw, data, n, c * c, o, key);
        if (!strcmp(rest, "conv_gated.bias")) return set_param(&l->gb, data, n, c, o, key);
        if (!strcmp(rest, "norm.weight")) return set_param(&l->nw, data, n, c, o, key);
        if (!strcmp(rest, "norm.bias")) return set_param(&l->nb, data, n, c, o, key);
    } else if (sscanf(key, "convlist.%d.%127s", &idx, rest) == 2 && idx >= 0 && idx < L) {
        layer_w *l = &o->enc[idx];
        long ci = o->Cin[idx], co = o->Cin[idx + 1];
        if (!strcmp(rest, "conv_trans.weight")) return set_param(&l->tw, data, n, co * co, o, key);
        if (!strcmp(rest, "conv_trans.bias")) return set_param(&l->tb, data, n, co, o, key);
        if (!strcmp(rest, "conv_gated.weight")) return set_param(&l->gw, data, n, co * co, o, key);
        if (!strcmp(rest, "conv_gated.bias")) return set_param(&l->gb, data, n, co, o, key);
        if (!strcmp(rest, "conv.weight") || !strcmp(rest, "net.0.weight")) return set_param(&l->w, data, n, co * ci * 15, o, key);
        if (!strcmp(rest, "conv.bias") || !strcmp(rest, "net.0.bias")) return set_param(&l->b, data, n, co, o, key);
        if (!strcmp(rest, "norm.weight")) return set_param(&l->nw, data, n, co, o, key);
        if (!strcmp(rest, "norm.bias")) return set_param(&l->nb, data, n, co, o, key);
    } else if (sscanf(key, "deconvlist.%d.%127s", &idx, rest) == 2 && idx >= 0 && idx < L) {
        layer_w *l = &o->dec[idx];
        int lvl = L - 1 - idx;
        long ci = o->Cin[lvl + 1], co = lvl == 0 ? 2 : o->Cin[lvl];
        if (!strcmp(rest, "conv.weight") || !strcmp(rest, "net.0.weight")) return set_param(&l->w, data, n, co * ci * 15, o, key);
        if (!strcmp(rest, "conv.bias") || !strcmp(rest, "net.0.bias")) return set_param(&l->b, data, n, co, o, key);
        if (!strcmp(rest, "norm.weight")) return set_param(&l->nw, data, n, co, o, key);
        if (!strcmp(rest, "norm.bias")) return set_param(&l->nb, data, n, co, o, key);
        if (!strcmp(rest, "residualmask.weight")) return set_param(&l->mw, data, n, co * co, o, key);
        if (!strcmp(rest, "residualmask.bias")) return set_param(&l->mb, data, n, co, o, key);
        if (!strcmp(rest, "residualnorm.weight")) return set_param(&l->mnw, data, n, co, o, key);
        if (!strcmp(rest, "residualnorm.bias")) return set_param(&l->mnb, data, n, co, o, key);
        if (!strcmp(rest, "residual.weight")) return set_param(&l->rw, data, n, co * co, o, key);
        if (!strcmp(rest, "residual.bias")) return set_param(&l->rb, data, n, co, o, key);
    } else if (sscanf(key, "gru.sequence_model.weight_ih_l%d", &idx) == 1 && idx < o->c.num_layers) {
        return set_param(&o->wih[idx], data, n, 3L * H * (idx == 0 ? o->D : H), o, key);
    } else if (sscanf(key, "gru.sequence_model.weight_hh_l%d", &idx) == 1 && idx < o->c.num_layers) {
        return set_param(&o->whh[idx], data, n, 3L * H * H, o, key);
    } else if (sscanf(key, "gru.sequence_model.bias_ih_l%d", &idx) == 1 && idx < o->c.num_layers) {
        return set_param(&o->bih[idx], data, n, 3L * H, o, key);
    } else if (sscanf(key, "gru.sequence_model.bias_hh_l%d", &idx) == 1 && idx < o->c.num_layers) {
        return set_param(&o->bhh[idx], data, n, 3L * H, o, key);
    } else if (!strcmp(key, "gru.fc_output_layer.weight")) return set_param(&o->fcw, data, n, (long)o->D * H, o, key);
    else if (!strcmp(key, "gru.fc_output_layer.bias")) return set_param(&o->fcb, data, n, o->D, o, key);
    else if (!strcmp(key, "gru.norm.weight")) return set_param(&o->gnw, data, n, o->D, o, key);
    else if (!strcmp(key, "gru.norm.bias")) return set_param(&o->gnb, data, n, o->D, o, key);
    snprintf(o->err, sizeof(o->err), "unknown parameter key %s", key);
    return -1;
}

/* ---- state: TemporalCRN.reset (CRN.py:498-503); buffers are lazily zero-filled (CRN.py:325-326) ---- */
void crn_oracle_reset(crn_oracle *o, int B) {
    free_state(o);
    o->B = B;
    int L = o->c.num_levels, T = o->T;
    for (int i = 0; i < L; i++) {
        int d = 1 << i;
        o->buf[i] = (float *)xcalloc((size_t)B * o->Cin[i] * o->F[i] * 2 * d, sizeof(float));
        o->tap_enc[i] = (float *)xcalloc((size_t)B * o->Cin[i + 1] * o->F[i + 1] * T, sizeof(float));
        int lvl = L - 1 - i;
        int co = lvl == 0 ? 2 : o->Cin[lvl];
        o->tap_dec[i] = (float *)xcalloc((size_t)B * co * o->F[lvl] * T, sizeof(float));
    }
    if (o->c.variant) {
        for (int i = 0; i < 3; i++) o->pbuf[i] = (float *)xcalloc((size_t)B * o->Cin[0] * o->F[0] * 4, sizeof(float));
        o->tap_pre = (float *)xcalloc((size_t)B * o->Cin[0] * o->F[0] * T, sizeof(float));
    }
    o->h = (float *)xcalloc((size_t)o->c.num_layers * B * o->c.hidden, sizeof(float));
    o->tap_feat = (float *)xcalloc((size_t)B * o->Cin[0] * o->F[0] * T, sizeof(float));
    o->tap_gru = (float *)xcalloc((size_t)B * o->D * T, sizeof(float));
}

/* ---- A2: torch.stft(n_fft, hop, win, hamming, center=True, pad_mode="constant", onesided) per segment.
 * x [n, K] -> out [n, F, T, 2]  (the layout CRN.py:511 hands to forward, per mic).  ---- */
void crn_oracle_stft(crn_oracle *o, const float *x, int n, float *out) {
    int N = o->c.n_fft, hop = o->c.hop, K = o->c.segment_length, T = o->T, F = o->c.num_freqs;
    int pad = N / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int s = 0; s < n; s++)
        for (int t = 0; t < T; t++) {
            double fr[2048];
            const float *xs = x + (size_t)s * K;
            for (int i = 0; i < N; i++) {
                int src = t * hop + i - pad;
                float v = (src >= 0 && src < K) ? xs[src] : 0.0f;
                fr[i] = (double)(v * o->window[i]); /* fp32 product, like torch */
            }
            for (int f = 0; f < F; f++) {
                double re = 0, im = 0;
                for (int i = 0; i < N; i++) {
                    int k = (int)(((long)f * i) % N);
                    re += fr[i] * o->tw_c[k];
                    im -= fr[i] * o->tw_s[k];
                }
                size_t off = (((size_t)s * F + f) * T + t) * 2;
                /* a real-input FFT returns exactly +0 imaginary parts at DC and Nyquist (torch/pocketfft do); CRN_ELU's
                 * atan2 (CRN_ELU.py:370) turns a -1e-17 there into a 2*pi phase flip, so the restatement must too */
                if (f == 0 || 2 * f == N) im = 0.0;
                out[off] = (float)re;
                out[off + 1] = (float)im;
            }
        }
}

/* ---- A3: torch.istft(center=True, length=None): irfft, window, overlap-add, / sum w^2, trim n_fft/2.
 * X [n, F, T, 2] -> y [n, hop*(T-1)] ---- */
void crn_oracle_istft(crn_oracle *o, const float *X, int n, float *y) {
    int N = o->c.n_fft, hop = o->c.hop, T = o->T, F = o->c.num_freqs;
    int full = N + hop * (T - 1), pad = N / 2, outlen = hop * (T - 1);
    float *env = (float *)xcalloc(full, sizeof(float));
    for (int t = 0; t < T; t++)
        for (int i = 0; i < N; i++) env[t * hop + i] += o->window[i] * o->window[i];
#pragma omp parallel for schedule(static)
    for (int s = 0; s < n; s++) {
        float *acc = (float *)xcalloc(full, sizeof(float));
        for (int t = 0; t < T; t++) {
            for (int i = 0; i < N; i++) {
                /* C2R: imaginary parts of DC and Nyquist are ignored */
                double v = X[(((size_t)s * F + 0) * T + t) * 2];
                if (N % 2 == 0) v += ((i & 1) ? -1.0 : 1.0) * X[(((size_t)s * F + N / 2) * T + t) * 2];
                for (int f = 1; f < (N + 1) / 2; f++) {
                    size_t off = (((size_t)s * F + f) * T + t) * 2;
                    int k = (int)(((long)f * i) % N);
                    v += 2.0 * (X[off] * o->tw_c[k] - X[off + 1] * o->tw_s[k]);
                }
                float frame = (float)(v / N);
                acc[t * hop + i] += frame * o->window[i];
            }
        }
        for (int i = 0; i < outlen; i++) y[(size_t)s * outlen + i] = acc[pad + i] / env[pad + i];
        free(acc);
    }
    free(env);
}

/* ---- A6: GlobalLayerNorm(time=False) forward, CRN.py:135-149.  x [B, n] per sample; affine index
 * = (i / inner) % dim, covering both [1,C,1,1] (inner=F*T, dim=C) and last=True [1,1,1,D] (inner=1). ---- */
static int g_gln_eps_outside_only = 0; /* student: sqrt(var)+eps (distillation_crn.py:51); else sqrt(var+eps)+eps (CRN.py:149) */
static void gln(float *x, int B, long n, const float *w, const float *b, long inner, long dim) {
#pragma omp parallel for schedule(static)
    for (int s = 0; s < B; s++) {
        float *p = x + (size_t)s * n;
        double sum = 0;
        for (long i = 0; i < n; i++) sum += p[i];
        float mean = (float)(sum / n);
        double sq = 0;
        for (long i = 0; i < n; i++) { float d = p[i] - mean; sq += (double)(d * d); }
        float var = (float)(sq / n);
        float den = (g_gln_eps_outside_only ? sqrtf(var) : sqrtf(var + EPS)) + EPS;
        for (long i = 0; i < n; i++) {
            long c = (i / inner) % dim;
            p[i] = (p[i] - mean) / den * w[c] + b[c];
        }
    }
}

static inline float sigmoidf(float v) { return 1.0f / (1.0f + expf(-v)); }

static inline float eluf(float v) { return v > 0 ? v : (expf(v) - 1.0f); } /* nn.ELU(alpha=1) */

/* ---- A5: TemporalConv2d.forward, CRN.py:321-338; CRN_ELU.py:233-247 (ELU, conv_trans * sigmoid(conv_gated)).
 * Generic geometry: kernel (5,KT), freq stride sf, freq dilation dilf, freq pad padf, time dilation dilt, causal
 * time buffer of P=(KT-1)*dilt columns.  Encoder: KT=3, sf=2, dilf=1, padf=2, dilt=2^i.  Preconv (CRN_ELU.py:335-340):
 * KT=5, sf=1, dilf=fd, padf=2*fd, dilt=1. ---- */
static void conv_block(crn_oracle *o, const layer_w *l, float *buf, const float *x, float *y, int Ci, int Co, int Fi, int Fo,
                       int KT, int sf, int dilf, int padf, int dilt) {
    int B = o->B, T = o->T, P = (KT - 1) * dilt, elu = o->c.variant != 0;
    int W = P + T;
    float *inp = (float *)xcalloc((size_t)B * Ci * Fi * W, sizeof(float)); /* cat([buffer, x], -1) */
    for (size_t r = 0; r < (size_t)B * Ci * Fi; r++) {
        memcpy(inp + r * W, buf + r * P, P * sizeof(float));
        memcpy(inp + r * W + P, x + r * T, T * sizeof(float));
    }
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; b++)
        for (int co = 0; co < Co; co++) {
            float *yo = y + ((size_t)b * Co + co) * Fo * T;
            for (int fo = 0; fo < Fo; fo++) {
                float acc[64];
                for (int t = 0; t < T; t++) acc[t] = l->b[co];
                for (int ci = 0; ci < Ci; ci++)
                    for (int kf = 0; kf < 5; kf++) {
                        int fi = sf * fo - padf + kf * dilf;
                        if (fi < 0 || fi >= Fi) continue;
                        const float *ir = inp + (((size_t)b * Ci + ci) * Fi + fi) * W;
                        const float *wr = l->w + (((size_t)co * Ci + ci) * 5 + kf) * KT;
                        for (int kt = 0; kt < KT; kt++) {
                            float w = wr[kt];
                            const float *iv = ir + kt * dilt;
                            for (int t = 0; t < T; t++) acc[t] += w * iv[t];
                        }
                    }
                for (int t = 0; t < T; t++) yo[fo * T + t] = elu ? eluf(acc[t]) : (acc[t] > 0 ? acc[t] : 0);
            }
        }
    free(inp);
    if (elu) { /* out = conv_trans(out) * sigmoid(conv_gated(out)), 1x1 convs (CRN_ELU.py:240) */
        size_t per = (size_t)Co * Fo * T;
        float *tmp = (float *)xcalloc((size_t)B * per, sizeof(float));
        memcpy(tmp, y, (size_t)B * per * sizeof(float));
#pragma omp parallel for collapse(2) schedule(static)
        for (int b = 0; b < B; b++)
            for (int co = 0; co < Co; co++)
                for (int p = 0; p < Fo * T; p++) {
                    float at = l->tb[co], ag = l->gb[co];
                    for (int ci = 0; ci < Co; ci++) {
                        float v = tmp[(size_t)b * per + (size_t)ci * Fo * T + p];
                        at += l->tw[co * Co + ci] * v;
                        ag += l->gw[co * Co + ci] * v;
                    }
                    y[(size_t)b * per + (size_t)co * Fo * T + p] = at * sigmoidf(ag);
                }
        free(tmp);
    }
    gln(y, B, (long)Co * Fo * T, l->nw, l->nb, (long)Fo * T, Co);
    /* buffer <- last P columns of x (T > P branch, CRN.py:333-334; else-branch 335-337) */
    for (size_t r = 0; r < (size_t)B * Ci * Fi; r++) {
        float *br = buf + r * P;
        const float *xr = x + r * T;
        if (T > P) {
            for (int p = 0; p < P; p++) br[p] = xr[T - P + p];
        } else {
            for (int p = 0; p < P - T; p++) br[p] = br[p + T];
            for (int p = 0; p < T; p++) br[P - T + p] = xr[p];
        }
    }
}

static void enc_block(crn_oracle *o, int i, const float *x, float *y) {
    conv_block(o, &o->enc[i], o->buf[i], x, y, o->Cin[i], o->Cin[i + 1], o->F[i], o->F[i + 1], 3, 2, 1, 2, 1 << i);
}

/* ---- A8: TemporalConvTranspose2d.forward, CRN.py:379-397.  j = decoder index; dilation 2^j
 * (deconvlist is built by prepending, CRN.py:438-444, so deconvlist[0] has dilation 1). ---- */
static void dec_block(crn_oracle *o, int j, const float *x, const float *res, float *y) {
    int L = o->c.num_levels, lvl = L - 1 - j;
    int B = o->B, T = o->T, Ci = o->Cin[lvl + 1], Co = lvl == 0 ? 2 : o->Cin[lvl];
    int Fi = o->F[lvl + 1], Fo = 2 * Fi - 1, Fr = o->F[lvl], d = 1 << j;
    const layer_w *l = &o->dec[j];
    /* y has Fr rows per channel (res freq size, or Fo when no skip); rows >= Fo are the zero pad */
    int Fy = res ? Fr : Fo;
    float *tmp = (float *)xcalloc((size_t)B * Co * Fo * T, sizeof(float));
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; b++)
        for (int co = 0; co < Co; co++) {
            float *yo = tmp + ((size_t)b * Co + co) * Fo * T;
            for (int fo = 0; fo < Fo; fo++) {
                float acc[64];
                for (int t = 0; t < T; t++) acc[t] = l->b[co];
                for (int kf = 0; kf < 5; kf++) {
                    int num = fo + 2 - kf; /* fo = 2*fi - 2 + kf */
                    if (num < 0 || (num & 1)) continue;
                    int fi = num / 2;
                    if (fi >= Fi) continue;
                    for (int ci = 0; ci < Ci; ci++) {
                        const float *xr = x + (((size_t)b * Ci + ci) * Fi + fi) * T;
                        const float *wr = l->w + (((size_t)ci * Co + co) * 5 + kf) * 3; /* [Cin,Cout,5,3] */
                        for (int kt = 0; kt < 3; kt++) {
                            float w = wr[kt];
                            int sh = (2 - kt) * d; /* keep last T of T+2d: out col t <- in col t+sh (CRN.py:383) */
                            for (int t = 0; t + sh < T; t++) acc[t] += w * xr[t + sh];
                        }
                    }
                }
                for (int t = 0; t < T; t++) yo[fo * T + t] = o->c.variant ? eluf(acc[t]) : (acc[t] > 0 ? acc[t] : 0);
            }
        }
    gln(tmp, B, (long)Co * Fo * T, l->nw, l->nb, (long)Fo * T, Co);
    if (!res) {
        memcpy(y, tmp, (size_t)B * Co * Fo * T * sizeof(float));
        free(tmp);
        return;
    }
    /* skip path CRN.py:387-396: pad/crop F to the skip's, mask = sigmoid(gLN(conv1x1(res))) */
    size_t per = (size_t)Co * Fr * T;
    float *mk = (float *)xcalloc((size_t)B * per, sizeof(float));
    float *rv = (float *)xcalloc((size_t)B * per, sizeof(float));
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; b++)
        for (int co = 0; co < Co; co++)
            for (int p = 0; p < Fr * T; p++) {
                float am = l->mb[co], ar = l->rb[co];
                for (int ci = 0; ci < Co; ci++) {
                    float v = res[((size_t)b * Co + ci) * Fr * T + p];
                    am += l->mw[co * Co + ci] * v;
                    ar += l->rw[co * Co + ci] * v;
                }
                mk[(size_t)b * per + (size_t)co * Fr * T + p] = am;
                rv[(size_t)b * per + (size_t)co * Fr * T + p] = o->c.variant ? eluf(ar) : (ar > 0 ? ar : 0); /* CRN_ELU.py:306 */
            }
    gln(mk, B, (long)per, l->mnw, l->mnb, (long)Fr * T, Co);
    for (int b = 0; b < B; b++)
        for (int co = 0; co < Co; co++)
            for (int f = 0; f < Fy; f++)
                for (int t = 0; t < T; t++) {
                    size_t oi = (size_t)b * per + ((size_t)co * Fr + f) * T + t;
                    float ov = f < Fo ? tmp[(((size_t)b * Co + co) * Fo + f) * T + t] : 0.0f;
                    float m = sigmoidf(mk[oi]);
                    y[oi] = m * rv[oi] + (1.0f - m) * ov;
                }
    free(tmp); free(mk); free(rv);
}

/* ---- A7: SequenceModel.forward (GRU + Linear + ReLU + gLN(last)), CRN.py:256-282.  x,y [B, D, T] ---- */
static void gru_block(crn_oracle *o, const float *x, float *y) {
    int B = o->B, T = o->T, D = o->D, H = o->c.hidden, NL = o->c.num_layers;
    float *seq = (float *)xcalloc((size_t)B * T * (D > H ? D : H), sizeof(float));
    float *nxt = (float *)xcalloc((size_t)B * T * H, sizeof(float));
    for (int b = 0; b < B; b++) /* permute(0,2,1): [B,D,T] -> [B,T,D] */
        for (int k = 0; k < D; k++)
            for (int t = 0; t < T; t++) seq[((size_t)b * T + t) * D + k] = x[((size_t)b * D + k) * T + t];
    int in = D;
    for (int l = 0; l < NL; l++) {
        const float *wih = o->wih[l], *whh = o->whh[l], *bih = o->bih[l], *bhh = o->bhh[l];
#pragma omp parallel for schedule(static)
        for (int b = 0; b < B; b++) {
            float *h = o->h + ((size_t)l * B + b) * H;
            float *gi = (float *)xcalloc(3 * H, sizeof(float));
            float *gh = (float *)xcalloc(3 * H, sizeof(float));
            for (int t = 0; t < T; t++) {
                const float *xt = seq + ((size_t)b * T + t) * in;
                for (int g = 0; g < 3 * H; g++) {
                    float a = 0;
                    const float *wr = wih + (size_t)g * in;
#pragma omp simd reduction(+ : a)
                    for (int k = 0; k < in; k++) a += wr[k] * xt[k];
                    gi[g] = a + bih[g];
                    float c = 0;
                    const float *hr = whh + (size_t)g * H;
#pragma omp simd reduction(+ : c)
                    for (int k = 0; k < H; k++) c += hr[k] * h[k];
                    gh[g] = c + bhh[g];
                }
                float *ho = nxt + ((size_t)b * T + t) * H;
                for (int k = 0; k < H; k++) { /* torch.nn.GRU gate order r,z,n */
                    float r = sigmoidf(gi[k] + gh[k]);
                    float z = sigmoidf(gi[H + k] + gh[H + k]);
                    float nn = tanhf(gi[2 * H + k] + r * gh[2 * H + k]);
                    ho[k] = (1.0f - z) * nn + z * h[k];
                }
                memcpy(h, ho, H * sizeof(float));
            }
            free(gi); free(gh);
        }
        memcpy(seq, nxt, (size_t)B * T * H * sizeof(float));
        in = H;
    }
    /* Linear H->D, ReLU, gLN(last=True) over [B,1,T,D], permute back */
    float *fc = (float *)xcalloc((size_t)B * T * D, sizeof(float));
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; b++)
        for (int t = 0; t < T; t++) {
            const float *hv = seq + ((size_t)b * T + t) * H;
            for (int k = 0; k < D; k++) {
                float a = 0;
                const float *wr = o->fcw + (size_t)k * H;
#pragma omp simd reduction(+ : a)
                for (int q = 0; q < H; q++) a += wr[q] * hv[q];
                a += o->fcb[k];
                fc[((size_t)b * T + t) * D + k] = o->c.variant ? eluf(a) : (a > 0 ? a : 0); /* CRN_ELU.py:365 output_activate_function */
            }
        }
    gln(fc, B, (long)T * D, o->gnw, o->gnb, 1, D);
    for (int b = 0; b < B; b++)
        for (int k = 0; k < D; k++)
            for (int t = 0; t < T; t++) y[((size_t)b * D + k) * T + t] = fc[((size_t)b * T + t) * D + k];
    free(seq); free(nxt); free(fc);
}

/* ---- TemporalCRN.forward, CRN.py:454-496.  x [B,M,F,T,2] -> y [B,F,T,2]; mutates state ---- */
int crn_oracle_forward(crn_oracle *o, const float *x, float *y) {
    int B = o->B, M = o->c.num_inputs, F = o->F[0], T = o->T, L = o->c.num_levels;
    if (B <= 0) { snprintf(o->err, sizeof(o->err), "forward before reset"); return -3; }
    g_gln_eps_outside_only = o->c.variant == 2;
    size_t FT = (size_t)F * T;
    float *feat = o->tap_feat;
    /* A4 featurise, CRN.py:463-467 */
    for (int b = 0; b < B; b++) {
        for (int m = 0; m < M; m++)
            for (size_t p = 0; p < FT; p++) {
                float re = x[(((size_t)b * M + m) * FT + p) * 2], im = x[(((size_t)b * M + m) * FT + p) * 2 + 1];
                feat[((size_t)b * (2 * M - 1) + m) * FT + p] = sqrtf(re * re + im * im + 1e-10f);
            }
        for (int m = 1; m < M; m++)
            for (size_t p = 0; p < FT; p++) {
                const float *x0 = x + (((size_t)b * M + 0) * FT + p) * 2, *xm = x + (((size_t)b * M + m) * FT + p) * 2;
                float a0, am;
                if (o->c.variant == 1) { a0 = atan2f(x0[1], x0[0]); am = atan2f(xm[1], xm[0]); } /* CRN_ELU.py:370 */
                else { a0 = atanf(x0[1] / (x0[0] + EPS) + EPS); am = atanf(xm[1] / (xm[0] + EPS) + EPS); }
                feat[((size_t)b * (2 * M - 1) + M + m - 1) * FT + p] = a0 - am;
            }
    }
    if (o->c.variant) { /* x = m(x) + x for the three frequency-dilated 5x5 blocks (CRN_ELU.py:375-376) */
        size_t nfe = (size_t)B * o->Cin[0] * FT;
        for (int i = 0; i < 3; i++) {
            int fd = 1 << i;
            conv_block(o, &o->pre[i], o->pbuf[i], feat, o->tap_pre, o->Cin[0], o->Cin[0], F, F, 5, 1, fd, 2 * fd, 1);
            for (size_t q = 0; q < nfe; q++) feat[q] = o->tap_pre[q] + feat[q];
        }
    }
    const float *cur = feat;
    for (int i = 0; i < L; i++) { enc_block(o, i, cur, o->tap_enc[i]); cur = o->tap_enc[i]; }
    gru_block(o, cur, o->tap_gru); /* reshape [B,C,F,T]->[B,C*F,T] is a no-op in this layout */
    cur = o->tap_gru;
    for (int j = 0; j < L; j++) {
        int lvl = L - 1 - j;
        const float *res = lvl > 0 ? o->tap_enc[lvl - 1] : NULL; /* residuals[-2-j], CRN.py:484-489 */
        dec_block(o, j, cur, res, o->tap_dec[j]);
        cur = o->tap_dec[j];
    }
    /* A9: decompress_cIRM (utility.py:439-442) + complex multiply with mic 0 (CRN.py:491-495) */
    for (int b = 0; b < B; b++)
        for (size_t p = 0; p < FT; p++) {
            float m2[2];
            for (int c = 0; c < 2; c++) {
                float m = cur[((size_t)b * 2 + c) * FT + p];
                m = m >= 9.9f ? 9.9f : (m <= -9.9f ? -9.9f : m);
                m2[c] = -10.0f * logf((10.0f - m) / (10.0f + m));
            }
            const float *n0 = x + (((size_t)b * M) * FT + p) * 2;
            y[((size_t)b * FT + p) * 2] = m2[0] * n0[0] - m2[1] * n0[1];
            y[((size_t)b * FT + p) * 2 + 1] = m2[1] * n0[0] + m2[0] * n0[1];
        }
    return 0;
}

/* ---- A1: utility.padding / segmentation (utility.py:312-370): x [B,M,L] -> seg [B*N,M,K]; returns N ---- */
long crn_oracle_gap(long L, long K) { long P = K / 2; return K - (P + L % K) % K; }

long crn_oracle_segment(const float *x, int B, int M, long L, long K, float *seg, long *gap_out) {
    long P = K / 2, gap = crn_oracle_gap(L, K);
    long N = 2 * (L + gap + P) / K;
    if (gap_out) *gap_out = gap;
    if (!seg) return N;
    for (int b = 0; b < B; b++)
        for (long n = 0; n < N; n++)
            for (int m = 0; m < M; m++)
                for (long k = 0; k < K; k++) {
                    long src = n * P + k - P; /* padded signal = [0]*P | x | [0]*gap | [0]*P, segment n at n*P */
                    seg[(((size_t)b * N + n) * M + m) * K + k] = (src >= 0 && src < L) ? x[((size_t)b * M + m) * L + src] : 0.0f;
                }
    return N;
}

/* ---- A1: utility.over_add (utility.py:373-403): y [B,N,K] -> out [B, N*K/2 - K/2 - gap] ---- */
long crn_oracle_overadd(const float *y, int B, long N, long K, long gap, float *out) {
    long P = K / 2, Lout = (N / 2) * K - P - gap;
    if (!out) return Lout;
    for (int b = 0; b < B; b++)
        for (long i = 0; i < Lout; i++) {
            long i1 = i + P, i2 = i; /* stream1[P:], stream2[:-P] */
            float a = y[((size_t)b * N + 2 * (i1 / K)) * K + i1 % K];
            float c = y[((size_t)b * N + 2 * (i2 / K) + 1) * K + i2 % K];
            out[(size_t)b * Lout + i] = (a + c) / 2;
        }
    return Lout;
}

/* ---- TemporalCRN.realtime_process, CRN.py:560-589.  mix [B,M,L] -> out [B,L]; flag as in the reference.
 * With flag != 0 the previous state (set by an earlier call with the same B) is carried. ---- */
int crn_oracle_realtime(crn_oracle *o, const float *mix, int B, long L, int flag, float *out) {
    int M = o->c.num_inputs, F = o->F[0], T = o->T;
    long K = o->c.segment_length, P = K / 2;
    long Lp = flag ? L : L + P;
    float *xp = (float *)xcalloc((size_t)B * M * Lp, sizeof(float));
    for (int b = 0; b < B; b++)
        for (int m = 0; m < M; m++)
            memcpy(xp + ((size_t)b * M + m) * Lp + (flag ? 0 : P), mix + ((size_t)b * M + m) * L, L * sizeof(float));
    long gap, N = crn_oracle_segment(xp, B, M, Lp, K, NULL, &gap);
    float *seg = (float *)xcalloc((size_t)B * N * M * K, sizeof(float));
    crn_oracle_segment(xp, B, M, Lp, K, seg, &gap);
    if (!flag) crn_oracle_reset(o, B);
    else if (o->B != B) { snprintf(o->err, sizeof(o->err), "flag=True with batch %d but state holds %d", B, o->B); free(xp); free(seg); return -4; }
    size_t spec_n = (size_t)F * T * 2;
    float *spec = (float *)xcalloc((size_t)B * N * M * spec_n, sizeof(float));
    crn_oracle_stft(o, seg, (int)(B * N * M), spec);
    float *xin = (float *)xcalloc((size_t)B * M * spec_n, sizeof(float));
    float *yo = (float *)xcalloc((size_t)B * spec_n, sizeof(float));
    float *ysp = (float *)xcalloc((size_t)B * N * spec_n, sizeof(float));
    for (long n = 0; n < N; n++) {
        for (int b = 0; b < B; b++)
            memcpy(xin + (size_t)b * M * spec_n, spec + ((size_t)b * N + n) * M * spec_n, M * spec_n * sizeof(float));
        int rc = crn_oracle_forward(o, xin, yo);
        if (rc) return rc;
        for (int b = 0; b < B; b++)
            memcpy(ysp + ((size_t)b * N + n) * spec_n, yo + (size_t)b * spec_n, spec_n * sizeof(float));
    }
    float *yseg = (float *)xcalloc((size_t)B * N * K, sizeof(float));
    crn_oracle_istft(o, ysp, (int)(B * N), yseg);
    long Lo = crn_oracle_overadd(yseg, B, N, K, gap, NULL);
    float *full = (float *)xcalloc((size_t)B * Lo, sizeof(float));
    crn_oracle_overadd(yseg, B, N, K, gap, full);
    long skip = flag ? 0 : P; /* CRN.py:587-588 */
    for (int b = 0; b < B; b++) memcpy(out + (size_t)b * L, full + (size_t)b * Lo + skip, L * sizeof(float));
    free(xp); free(seg); free(spec); free(xin); free(yo); free(ysp); free(yseg); free(full);
    return (Lo - skip == L) ? 0 : -5;
}

/* taps of the last forward(): name in {feat, enc0.., gru, dec0..}; returns pointer + element count */
const float *crn_oracle_tap(crn_oracle *o, const char *name, long *n) {
    int idx, L = o->c.num_levels, T = o->T, B = o->B;
    if (!strcmp(name, "feat")) { *n = (long)B * o->Cin[0] * o->F[0] * T; return o->tap_feat; }
    if (!strcmp(name, "gru")) { *n = (long)B * o->D * T; return o->tap_gru; }
    if (sscanf(name, "enc%d", &idx) == 1 && idx < L) { *n = (long)B * o->Cin[idx + 1] * o->F[idx + 1] * T; return o->tap_enc[idx]; }
    if (sscanf(name, "dec%d", &idx) == 1 && idx < L) {
        int lvl = L - 1 - idx;
        *n = (long)B * (lvl == 0 ? 2 : o->Cin[lvl]) * o->F[lvl] * T;
        return o->tap_dec[idx];
    }
    *n = 0;
    return NULL;
}

/* state export for tests: encoder buffer i [B,Cin,F,2d], GRU h [layers,B,H] */
const float *crn_oracle_state(crn_oracle *o, const char *name, long *n) {
    int idx;
    if (!strcmp(name, "h")) { *n = (long)o->c.num_layers * o->B * o->c.hidden; return o->h; }
    if (sscanf(name, "pbuf%d", &idx) == 1 && idx < 3 && o->c.variant) { *n = (long)o->B * o->Cin[0] * o->F[0] * 4; return o->pbuf[idx]; }
    if (sscanf(name, "buf%d", &idx) == 1 && idx < o->c.num_levels) { *n = (long)o->B * o->Cin[idx] * o->F[idx] * 2 * (1 << idx); return o->buf[idx]; }
    *n = 0;
    return NULL;
}

/* ---- A12 (SI-SNR term only): utility.cal_si_snr, utility.py:207-223 ---- */
float crn_oracle_si_snr(const float *sep, const float *src, int B, long L, const int64_t *length) {
    double total = 0;
    for (int b = 0; b < B; b++) {
        long n = length ? length[b] : L;
        const float *s = sep + (size_t)b * L, *r = src + (size_t)b * L;
        double ms = 0, mr = 0;
        for (long i = 0; i < n; i++) { ms += s[i]; mr += r[i]; }
        ms /= n; mr /= n;
        double dot = 0, rr = 0;
        for (long i = 0; i < n; i++) { dot += (s[i] - ms) * (r[i] - mr); rr += (r[i] - mr) * (r[i] - mr); }
        double sc = dot / (rr + 1e-8), tt = 0, ee = 0;
        for (long i = 0; i < n; i++) {
            double tr = sc * (r[i] - mr), e = (s[i] - ms) - tr;
            tt += tr * tr; ee += e * e;
        }
        total += 20.0 * log10(1e-8 + sqrt(tt) / (sqrt(ee) + 1e-8));
    }
    return (float)(total / B);
}

/* ---- small per-op entry points so the golden per-block vectors can be checked directly ---- */
void crn_oracle_gln(float *x, int B, long n, const float *w, const float *b, long inner, long dim) {
    gln(x, B, n, w, b, inner, dim);
}

void crn_oracle_decompress_cirm(const float *m_in, long n, float *out) { /* utility.py:439-442 */
    for (long i = 0; i < n; i++) {
        float m = m_in[i];
        m = m >= 9.9f ? 9.9f : (m <= -9.9f ? -9.9f : m);
        out[i] = -10.0f * logf((10.0f - m) / (10.0f + m));
    }
}

/* thread control for the cpu_baseline leg of bench.py (returns the thread count in effect) */
#ifdef _OPENMP
#include <omp.h>
int crn_oracle_set_threads(int n) { if (n > 0) omp_set_num_threads(n); return omp_get_max_threads(); }
#else
int crn_oracle_set_threads(int n) { (void)n; return 1; }
#endif

/*
 * fsn_oracle.c - CPU restatement of the reference's FullSubNet streaming path.
 *
 * TEST INFRASTRUCTURE ONLY (see crn_oracle.c header).  Pinned by golden vectors produced by running the genuine
 * reference class fullsubnet.FullSubNet (tests/golden/make_golden.py -> fsn_golden.npz).  STFT/iSTFT/segmentation/
 * over_add/decompress_cIRM are shared with crn_oracle.c (same speechbrain boundary: parity unpinned there).
 * Citations are file:line into /root/reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define EPS 1e-8f /* fullsubnet.py:12 */

typedef struct {
    int num_freqs, num_mics, fb_hidden, sb_hidden, num_layers, sb_neighbors, fb_neighbors, look_ahead;
    int n_fft, win, hop, segment_length;
} fsn_cfg;

/* from crn_oracle.c */
typedef struct crn_oracle crn_oracle;
typedef struct {
    int num_levels;
    int channels[8];
    int num_freqs, hidden, num_layers, num_inputs, kernel_size;
    int n_fft, win, hop, segment_length;
    int variant;
} crn_cfg;
crn_oracle *crn_oracle_create(const crn_cfg *cfg);
void crn_oracle_destroy(crn_oracle *o);
void crn_oracle_stft(crn_oracle *o, const float *x, int n, float *out);
void crn_oracle_istft(crn_oracle *o, const float *X, int n, float *y);
long crn_oracle_segment(const float *x, int B, int M, long L, long K, float *seg, long *gap_out);
long crn_oracle_overadd(const float *y, int B, long N, long K, long gap, float *out);
void crn_oracle_decompress_cirm(const float *m_in, long n, float *out);

typedef struct {
    int in, H;
    float *wih[4], *whh[4], *bih[4], *bhh[4]; /* per layer */
    float *fcw, *fcb;
    int out;
} seq_w;

typedef struct fsn_oracle {
    fsn_cfg c;
    int T, F, M;
    seq_w fb, sb;
    crn_oracle *sig; /* STFT / iSTFT helper */
    /* state: LSTM (h, c) per model [layers][rows][H]; CumLayerNorm running mean + step per stream (fullsubnet.py:177-205) */
    int B;
    float *fh, *fc, *sh, *sc;
    float *mean_fb, *mean_sb;
    int step_fb, step_sb, have_fb, have_sb;
    float *tap_fb; /* fb_output [B,F,T] of the last forward */
    char err[256];
} fsn_oracle;

static void *xcalloc(size_t n, size_t s) {
    void *p = calloc(n ? n : 1, s);
    if (!p) { fprintf(stderr, "fsn_oracle: out of memory\n"); abort(); }
    return p;
}
static inline float sigmoidf(float v) { return 1.0f / (1.0f + expf(-v)); }

const char *fsn_oracle_error(fsn_oracle *o) { return o->err; }

fsn_oracle *fsn_oracle_create(const fsn_cfg *cfg) {
    if (cfg->num_layers < 1 || cfg->num_layers > 4 || cfg->fb_neighbors != 0 || cfg->look_ahead != 0) return NULL;
    fsn_oracle *o = (fsn_oracle *)xcalloc(1, sizeof(*o));
    o->c = *cfg;
    o->T = 1 + cfg->segment_length / cfg->hop;
    o->F = cfg->num_freqs;
    o->M = cfg->num_mics;
    o->fb.in = cfg->num_freqs * cfg->num_mics; o->fb.H = cfg->fb_hidden; o->fb.out = cfg->num_freqs;       /* fullsubnet.py:728-736 */
    o->sb.in = (cfg->sb_neighbors * 2 + 1) + (cfg->fb_neighbors * 2 + 1); o->sb.H = cfg->sb_hidden; o->sb.out = 2; /* 738-746 */
    crn_cfg sc;
    memset(&sc, 0, sizeof(sc));
    sc.num_levels = 1; sc.channels[0] = 2; sc.num_freqs = cfg->num_freqs; sc.hidden = 4; sc.num_layers = 1; sc.num_inputs = cfg->num_mics;
    sc.kernel_size = 3; sc.n_fft = cfg->n_fft; sc.win = cfg->win; sc.hop = cfg->hop; sc.segment_length = cfg->segment_length;
    /* the helper only needs the STFT tables; pick a level count whose reshape check passes */
    for (int L = 1; L <= 8 && !o->sig; L++) {
        sc.num_levels = L;
        for (int i = 0; i < L; i++) sc.channels[i] = 2;
        o->sig = crn_oracle_create(&sc);
    }
    if (!o->sig) { free(o); return NULL; }
    return o;
}

static void free_state(fsn_oracle *o) {
    free(o->fh); free(o->fc); free(o->sh); free(o->sc); free(o->mean_fb); free(o->mean_sb); free(o->tap_fb);
    o->fh = o->fc = o->sh = o->sc = o->mean_fb = o->mean_sb = o->tap_fb = NULL;
}

void fsn_oracle_destroy(fsn_oracle *o) {
    if (!o) return;
    free_state(o);
    seq_w *ws[2] = {&o->fb, &o->sb};
    for (int k = 0; k < 2; k++) {
        for (int l = 0; l < 4; l++) { free(ws[k]->wih[l]); free(ws[k]->whh[l]); free(ws[k]->bih[l]); free(ws[k]->bhh[l]); }
        free(ws[k]->fcw); free(ws[k]->fcb);
    }
    crn_oracle_destroy(o->sig);
    free(o);
}

static int set_param(float **slot, const float *data, long n, long expect, fsn_oracle *o, const char *key) {
    if (n != expect) { snprintf(o->err, sizeof(o->err), "shape mismatch for %s: got %ld elements, expected %ld", key, n, expect); return -2; }
    free(*slot);
    *slot = (float *)xcalloc(n, sizeof(float));
    memcpy(*slot, data, n * sizeof(float));
    return 0;
}

int fsn_oracle_load(fsn_oracle *o, const char *key, const float *data, const int64_t *shape, int ndim) {
    long n = 1;
    for (int i = 0; i < ndim; i++) n *= shape[i];
    seq_w *w = NULL;
    const char *rest = NULL;
    if (!strncmp(key, "fb_model.", 9)) { w = &o->fb; rest = key + 9; }
    else if (!strncmp(key, "sb_model.", 9)) { w = &o->sb; rest = key + 9; }
    if (w) {
        int l;
        if (sscanf(rest, "sequence_model.weight_ih_l%d", &l) == 1 && l < o->c.num_layers) return set_param(&w->wih[l], data, n, 4L * w->H * (l == 0 ? w->in : w->H), o, key);
        if (sscanf(rest, "sequence_model.weight_hh_l%d", &l) == 1 && l < o->c.num_layers) return set_param(&w->whh[l], data, n, 4L * w->H * w->H, o, key);
        if (sscanf(rest, "sequence_model.bias_ih_l%d", &l) == 1 && l < o->c.num_layers) return set_param(&w->bih[l], data, n, 4L * w->H, o, key);
        if (sscanf(rest, "sequence_model.bias_hh_l%d", &l) == 1 && l < o->c.num_layers) return set_param(&w->bhh[l], data, n, 4L * w->H, o, key);
        if (!strcmp(rest, "fc_output_layer.weight")) return set_param(&w->fcw, data, n, (long)w->out * w->H, o, key);
        if (!strcmp(rest, "fc_output_layer.bias")) return set_param(&w->fcb, data, n, w->out, o, key);
    }
    snprintf(o->err, sizeof(o->err), "unknown parameter key %s", key);
    return -1;
}

/* reset_state, fullsubnet.py:826-832 */
void fsn_oracle_reset(fsn_oracle *o, int B) {
    free_state(o);
    o->B = B;
    int NL = o->c.num_layers;
    o->fh = (float *)xcalloc((size_t)NL * B * o->fb.H, sizeof(float));
    o->fc = (float *)xcalloc((size_t)NL * B * o->fb.H, sizeof(float));
    o->sh = (float *)xcalloc((size_t)NL * B * o->F * o->sb.H, sizeof(float));
    o->sc = (float *)xcalloc((size_t)NL * B * o->F * o->sb.H, sizeof(float));
    o->mean_fb = (float *)xcalloc(B, sizeof(float));
    o->mean_sb = (float *)xcalloc(B, sizeof(float));
    o->tap_fb = (float *)xcalloc((size_t)B * o->F * o->T, sizeof(float));
    o->step_fb = o->step_sb = 0;
    o->have_fb = o->have_sb = 0;
}

/* CumLayerNorm.forward, fullsubnet.py:184-201: per-stream mean over all non-batch dims; running mean with
 * alpha = step/(step+1), step capped at 80; in place x /= mean + EPS. */
static void cum_norm(float *x, int B, long n, float *run_mean, int *step, int *have) {
    for (int b = 0; b < B; b++) {
        float *p = x + (size_t)b * n;
        double s = 0;
        for (long i = 0; i < n; i++) s += p[i];
        float mean = (float)(s / n);
        if (!*have) run_mean[b] = mean;
        else {
            float alpha = (float)*step / (float)(*step + 1);
            run_mean[b] = alpha * run_mean[b] + (1.0f - alpha) * mean;
        }
    }
    *have = 1;
    *step += 1;
    if (*step > 80) *step = 80;
    for (int b = 0; b < B; b++) {
        float *p = x + (size_t)b * n;
        float den = run_mean[b] + EPS;
        for (long i = 0; i < n; i++) p[i] /= den;
    }
}

/* SequenceModel.forward (LSTM + Linear + activation), fullsubnet.py:274-292.  x [R, in, T] -> y [R, out, T];
 * h, c [layers][R][H] carried.  torch.nn.LSTM gate order i, f, g, o. */
static void seq_forward(const seq_w *w, int NL, int R, int T, const float *x, float *h, float *c, float *y, int relu) {
    int H = w->H;
    size_t maxw = (size_t)(w->in > H ? w->in : H);
#pragma omp parallel for schedule(dynamic, 8)
    for (int r = 0; r < R; r++) {
        float *seq = (float *)xcalloc((size_t)T * maxw, sizeof(float));
        float *nxt = (float *)xcalloc((size_t)T * H, sizeof(float));
        float *g = (float *)xcalloc(4 * (size_t)H, sizeof(float));
        for (int k = 0; k < w->in; k++)
            for (int t = 0; t < T; t++) seq[(size_t)t * w->in + k] = x[((size_t)r * w->in + k) * T + t]; /* permute(0,2,1) */
        int in = w->in;
        for (int l = 0; l < NL; l++) {
            float *hl = h + ((size_t)l * R + r) * H, *cl = c + ((size_t)l * R + r) * H;
            for (int t = 0; t < T; t++) {
                const float *xt = seq + (size_t)t * in;
                for (int q = 0; q < 4 * H; q++) {
                    float a = 0, d = 0;
                    const float *wi = w->wih[l] + (size_t)q * in, *wh = w->whh[l] + (size_t)q * H;
#pragma omp simd reduction(+ : a)
                    for (int k = 0; k < in; k++) a += wi[k] * xt[k];
#pragma omp simd reduction(+ : d)
                    for (int k = 0; k < H; k++) d += wh[k] * hl[k];
                    g[q] = (a + w->bih[l][q]) + (d + w->bhh[l][q]);
                }
                float *ho = nxt + (size_t)t * H;
                for (int k = 0; k < H; k++) {
                    float ig = sigmoidf(g[k]), fg = sigmoidf(g[H + k]), gg = tanhf(g[2 * H + k]), og = sigmoidf(g[3 * H + k]);
                    cl[k] = fg * cl[k] + ig * gg;
                    ho[k] = og * tanhf(cl[k]);
                }
                memcpy(hl, ho, H * sizeof(float));
            }
            memcpy(seq, nxt, (size_t)T * H * sizeof(float));
            in = H;
        }
        for (int t = 0; t < T; t++)
            for (int k = 0; k < w->out; k++) {
                float a = 0;
                const float *wr = w->fcw + (size_t)k * H, *hv = seq + (size_t)t * H;
#pragma omp simd reduction(+ : a)
                for (int q = 0; q < H; q++) a += wr[q] * hv[q];
                a += w->fcb[k];
                y[((size_t)r * w->out + k) * T + t] = (relu && a < 0) ? 0 : a;
            }
        free(seq); free(nxt); free(g);
    }
}

/* FullSubNet.forward, fullsubnet.py:769-824.  x [B, 2M, F, T] (re x M, then im x M) -> crm [B, 2, F, T] */
int fsn_oracle_forward(fsn_oracle *o, const float *x, float *crm) {
    int B = o->B, M = o->M, F = o->F, T = o->T, NB = o->c.sb_neighbors, W = 2 * NB + 1, SI = o->sb.in;
    if (B <= 0) { snprintf(o->err, sizeof(o->err), "forward before reset"); return -3; }
    size_t FT = (size_t)F * T;
    float *noisy = (float *)xcalloc((size_t)B * M * FT, sizeof(float));
    for (int b = 0; b < B; b++)
        for (int m = 0; m < M; m++)
            for (size_t p = 0; p < FT; p++) {
                float re = x[((size_t)b * 2 * M + m) * FT + p], im = x[((size_t)b * 2 * M + M + m) * FT + p];
                noisy[((size_t)b * M + m) * FT + p] = sqrtf(re * re + im * im + EPS); /* :782 */
            }
    /* norm_fb normalises `noisy` IN PLACE (x /= mean+EPS, :200), so the sub-band branch below sees the normalised mic 0 */
    cum_norm(noisy, B, (long)M * FT, o->mean_fb, &o->step_fb, &o->have_fb);
    float *fb_out = o->tap_fb;
    seq_forward(&o->fb, o->c.num_layers, B, T, noisy, o->fh, o->fc, fb_out, 1); /* fb_input = [B, M*F, T] is `noisy` itself */
    /* sub-band input [B, F, 31+1, T]: reflect-padded unfold of mic 0 (:300-331, 796-801), then the fb output */
    float *sbin = (float *)xcalloc((size_t)B * F * SI * T, sizeof(float));
    for (int b = 0; b < B; b++)
        for (int f = 0; f < F; f++) {
            for (int j = 0; j < W; j++) {
                int fi = f - NB + j; /* index into the reflect-padded axis */
                if (fi < 0) fi = -fi;
                if (fi >= F) fi = 2 * (F - 1) - fi;
                memcpy(sbin + (((size_t)b * F + f) * SI + j) * T, noisy + ((size_t)b * M * F + fi) * T, T * sizeof(float));
            }
            memcpy(sbin + (((size_t)b * F + f) * SI + W) * T, fb_out + ((size_t)b * F + f) * T, T * sizeof(float));
        }
    cum_norm(sbin, B, (long)F * SI * T, o->mean_sb, &o->step_sb, &o->have_sb);
    float *mask = (float *)xcalloc((size_t)B * F * 2 * T, sizeof(float));
    seq_forward(&o->sb, o->c.num_layers, B * F, T, sbin, o->sh, o->sc, mask, 0);
    for (int b = 0; b < B; b++) /* [B*F, 2, T] -> [B, 2, F, T], :814 */
        for (int f = 0; f < F; f++)
            for (int k = 0; k < 2; k++)
                memcpy(crm + (((size_t)b * 2 + k) * F + f) * T, mask + (((size_t)b * F + f) * 2 + k) * T, T * sizeof(float));
    free(noisy); free(sbin); free(mask);
    return 0;
}

/* FullSubNet.realtime_process(mixture, source, flag, train=False), fullsubnet.py:903-961 (the `source` argument only feeds
 * return values that predict_fullsubnet.py:75 discards).  mix [B, M, L] -> out [B, L] */
int fsn_oracle_realtime(fsn_oracle *o, const float *mix, int B, long L, int flag, float *out) {
    int M = o->M, F = o->F, T = o->T;
    long K = o->c.segment_length, P = K / 2;
    long Lp = flag ? L : L + P;
    float *xp = (float *)xcalloc((size_t)B * M * Lp, sizeof(float));
    for (int b = 0; b < B; b++)
        for (int m = 0; m < M; m++)
            memcpy(xp + ((size_t)b * M + m) * Lp + (flag ? 0 : P), mix + ((size_t)b * M + m) * L, L * sizeof(float));
    long gap, N = crn_oracle_segment(xp, B, M, Lp, K, NULL, &gap);
    float *seg = (float *)xcalloc((size_t)B * N * M * K, sizeof(float));
    crn_oracle_segment(xp, B, M, Lp, K, seg, &gap);
    if (!flag) fsn_oracle_reset(o, B);
    else if (o->B != B) { snprintf(o->err, sizeof(o->err), "flag=True with batch %d but state holds %d", B, o->B); free(xp); free(seg); return -4; }
    size_t FT = (size_t)F * T;
    float *spec = (float *)xcalloc((size_t)B * N * M * FT * 2, sizeof(float)); /* [B*N*M, F, T, 2] */
    crn_oracle_stft(o->sig, seg, (int)(B * N * M), spec);
    float *xin = (float *)xcalloc((size_t)B * 2 * M * FT, sizeof(float));
    float *crm = (float *)xcalloc((size_t)B * 2 * FT, sizeof(float));
    float *dm = (float *)xcalloc((size_t)B * 2 * FT, sizeof(float));
    float *ysp = (float *)xcalloc((size_t)B * N * FT * 2, sizeof(float));
    for (long n = 0; n < N; n++) {
        for (int b = 0; b < B; b++)
            for (int m = 0; m < M; m++)
                for (size_t p = 0; p < FT; p++) { /* stft_trans: re x M then im x M, :835-844 */
                    const float *s = spec + ((((size_t)b * N + n) * M + m) * FT + p) * 2;
                    xin[((size_t)b * 2 * M + m) * FT + p] = s[0];
                    xin[((size_t)b * 2 * M + M + m) * FT + p] = s[1];
                }
        int rc = fsn_oracle_forward(o, xin, crm);
        if (rc) return rc;
        crn_oracle_decompress_cirm(crm, (long)B * 2 * FT, dm); /* :949 */
        for (int b = 0; b < B; b++)
            for (size_t p = 0; p < FT; p++) {
                float mr = dm[((size_t)b * 2) * FT + p], mi = dm[((size_t)b * 2 + 1) * FT + p];
                float re = xin[((size_t)b * 2 * M) * FT + p], im = xin[((size_t)b * 2 * M + M) * FT + p];
                float *y = ysp + (((size_t)b * N + n) * FT + p) * 2;
                y[0] = mr * re - mi * im; /* :951-952 */
                y[1] = mi * re + mr * im;
            }
    }
    float *yseg = (float *)xcalloc((size_t)B * N * K, sizeof(float));
    crn_oracle_istft(o->sig, ysp, (int)(B * N), yseg);
    long Lo = crn_oracle_overadd(yseg, B, N, K, gap, NULL);
    float *full = (float *)xcalloc((size_t)B * Lo, sizeof(float));
    crn_oracle_overadd(yseg, B, N, K, gap, full);
    long skip = flag ? 0 : P;
    for (int b = 0; b < B; b++) memcpy(out + (size_t)b * L, full + (size_t)b * Lo + skip, L * sizeof(float));
    free(xp); free(seg); free(spec); free(xin); free(crm); free(dm); free(ysp); free(yseg); free(full);
    return (Lo - skip == L) ? 0 : -5;
}

const float *fsn_oracle_tap(fsn_oracle *o, const char *name, long *n) {
    if (!strcmp(name, "fb_out")) { *n = (long)o->B * o->F * o->T; return o->tap_fb; }
    if (!strcmp(name, "fh")) { *n = (long)o->c.num_layers * o->B * o->fb.H; return o->fh; }
    if (!strcmp(name, "fc")) { *n = (long)o->c.num_layers * o->B * o->fb.H; return o->fc; }
    if (!strcmp(name, "sh")) { *n = (long)o->c.num_layers * o->B * o->F * o->sb.H; return o->sh; }
    if (!strcmp(name, "sc")) { *n = (long)o->c.num_layers * o->B * o->F * o->sb.H; return o->sc; }
    if (!strcmp(name, "mean_fb")) { *n = o->B; return o->mean_fb; }
    if (!strcmp(name, "mean_sb")) { *n = o->B; return o->mean_sb; }
    *n = 0;
    return NULL;
}

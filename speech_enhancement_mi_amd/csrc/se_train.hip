// se_train.hip - round-3 kernels of the TRAINING step (reference train.py:195-204 = torch autograd over
// CRN.py:111-159 (GlobalLayerNorm), 256-282 (GRU), 387-396 (skip gate), 463-467 (features), 505-520 (STFT / iSTFT),
// utility.py:373-403 (over_add), 439-442 (decompress_cIRM)).  Own translation unit; exported through the se_train_* C ABI of
// include/se_engine.h.  Built with -fno-slp-vectorize like every unit of the library.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <string>

#include <cmath>
#include <vector>

#include "../../include/se_engine.h"
#include "gru_pseq.hip.h"
#include "stft.hip.h"
#include "train_fused.hip.h"

namespace se {

static thread_local std::string g_train_error;

int train_fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_train_error = buf;
    return code;
}

static int pseq_kj(int H) {  // smallest register block that keeps the K split within eight waves
    for (int kj : {4, 8, 16, 32})
        if (H % (4 * kj) == 0 && H / (4 * kj) <= 8) return kj;
    return 0;
}

void launch_arrange_w(const float *w, float *out, long sCo, long sCi, int Co, int Ci, int ntap, int CC, int nchunk, int CoPad, int one_by_one,
                      const int *kf, const int *kt, hipStream_t st) {
    TArrangeArgs a{};
    a.w = w; a.out = out; a.sCo = sCo; a.sCi = sCi; a.Co = Co; a.Ci = Ci; a.ntap = ntap; a.CC = CC; a.nchunk = nchunk; a.CoPad = CoPad;
    a.one_by_one = one_by_one;
    for (int t = 0; t < ntap && t < 15; t++) { a.tap_kf[t] = kf[t]; a.tap_kt[t] = kt[t]; }
    const long total = (long)nchunk * ntap * CC * CoPad;
    hipLaunchKernelGGL(k_arrange_w, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
}

}  // namespace se

using se::train_fail;

// STFT geometry + tables of the training step's signal stages (the engine's k_stft / k_istft kernels without an engine handle)
struct se_sig {
    int device = 0, N = 0, win = 0, hop = 0, K = 0, T = 0, F = 0;
    float *window = nullptr, *env = nullptr, *tw = nullptr;
    se::FftPlan plan{};
};

extern "C" {

const char *se_train_last_error(void) { return se::g_train_error.c_str(); }

int se_train_gru_pseq_supported(int B, int H) { return B >= 1 && H > 0 && H % 16 == 0 && se::pseq_kj(H) != 0; }

// One launch for ANY number of independent streams: groups of Bg <= 32 streams (blockIdx.y), each group = H/16 resident workgroups with
// their own arrival counter and exchange slab.  Blocks are dispatched group-major, so a group's workgroups become resident together and a
// group that does not fit yet simply waits in the dispatcher until an earlier group drains - groups never wait on each other.
#define SE_PSEQ_LAUNCH(KERNEL, ARGS)                                                                         \
    do {                                                                                                     \
        const dim3 grid(H / 16, groups), block(64 * (H / (4 * kj)));                                         \
        if (MT == 1) {                                                                                       \
            if (kj == 4) hipLaunchKernelGGL((se::KERNEL<4, 1>), grid, block, 0, st, ARGS);                   \
            else if (kj == 8) hipLaunchKernelGGL((se::KERNEL<8, 1>), grid, block, 0, st, ARGS);              \
            else if (kj == 16) hipLaunchKernelGGL((se::KERNEL<16, 1>), grid, block, 0, st, ARGS);            \
            else hipLaunchKernelGGL((se::KERNEL<32, 1>), grid, block, 0, st, ARGS);                          \
        } else {                                                                                             \
            if (kj == 4) hipLaunchKernelGGL((se::KERNEL<4, 2>), grid, block, 0, st, ARGS);                   \
            else if (kj == 8) hipLaunchKernelGGL((se::KERNEL<8, 2>), grid, block, 0, st, ARGS);              \
            else if (kj == 16) hipLaunchKernelGGL((se::KERNEL<16, 2>), grid, block, 0, st, ARGS);            \
            else hipLaunchKernelGGL((se::KERNEL<32, 2>), grid, block, 0, st, ARGS);                          \
        }                                                                                                    \
    } while (0)

static void pseq_groups(int B, int &Bg, int &groups, int &MT) {
    // 16-stream groups (one MFMA row tile) once there are more streams than two groups' worth: more groups in flight, shorter steps
    Bg = B <= 32 ? B : 16;
    if (B <= 32 && B > 16) Bg = B;
    groups = (B + Bg - 1) / Bg;
    MT = Bg > 16 ? 2 : 1;
}

int se_train_gru_pseq_scratch_floats(int B, int H) {
    int Bg, groups, MT;
    pseq_groups(B, Bg, groups, MT);
    return 16 * groups + groups * 2 * Bg * 3 * H;
}

int se_train_gru_pseq_fwd(const float *gi, const float *h0, const float *whh, const float *bhh, float *out, float *gates, float *hT,
                          float *scratch, int B, int T, int H, int Tseg, int64_t ldN, int64_t ldB, void *stream) {
    if (!gi || !h0 || !whh || !bhh || !out || !hT || !scratch || T <= 0 || Tseg <= 0) return train_fail(SE_ERR_ARG, "null / bad argument");
    if (!se_train_gru_pseq_supported(B, H)) return train_fail(SE_ERR_ARG, "persistent GRU: B = %d, H = %d unsupported", B, H);
    hipStream_t st = static_cast<hipStream_t>(stream);
    int Bg, groups, MT;
    pseq_groups(B, Bg, groups, MT);
    if (groups > 1 && ldN != 0) return train_fail(SE_ERR_ARG, "more than 32 streams need rows [B][T] (ldN = 0)");
    const int kj = se::pseq_kj(H);
    if (hipMemsetAsync(scratch, 0, (size_t)64 * groups, st) != hipSuccess) return train_fail(SE_ERR_HIP, "memset failed");  // arrival counters + timeout word
    se::GruPseqFwdArgs a{gi, h0, whh, bhh, out, gates, hT, scratch + 16 * groups, reinterpret_cast<unsigned *>(scratch), B, T, H, Tseg, (long)ldN, (long)ldB, Bg};
    SE_PSEQ_LAUNCH(k_gru_pseq_fwd, a);
    return hipGetLastError() == hipSuccess ? SE_OK : train_fail(SE_ERR_HIP, "persistent GRU forward launch failed");
}

int se_train_gru_pseq_bwd(const float *dout, const float *dhT, const float *gates, const float *out, const float *h0, const float *whh_t,
                          float *dgi, float *dgh, float *scratch, int B, int T, int H, int Tseg, int64_t ldN, int64_t ldB, int seg_len,
                          void *stream) {
    if (!dout || !gates || !out || !h0 || !whh_t || !dgi || !dgh || !scratch || T <= 0 || Tseg <= 0) return train_fail(SE_ERR_ARG, "null / bad argument");
    if (!se_train_gru_pseq_supported(B, H)) return train_fail(SE_ERR_ARG, "persistent GRU: B = %d, H = %d unsupported", B, H);
    hipStream_t st = static_cast<hipStream_t>(stream);
    int Bg, groups, MT;
    pseq_groups(B, Bg, groups, MT);
    if (groups > 1 && ldN != 0) return train_fail(SE_ERR_ARG, "more than 32 streams need rows [B][T] (ldN = 0)");
    const int kj = se::pseq_kj(H);
    if (hipMemsetAsync(scratch, 0, (size_t)64 * groups, st) != hipSuccess) return train_fail(SE_ERR_HIP, "memset failed");
    se::GruPseqBwdArgs a{dout, dhT, gates, out, h0, whh_t, dgi, dgh, scratch + 16 * groups, reinterpret_cast<unsigned *>(scratch), B, T, H, Tseg, seg_len, (long)ldN, (long)ldB, Bg};
    SE_PSEQ_LAUNCH(k_gru_pseq_bwd, a);
    return hipGetLastError() == hipSuccess ? SE_OK : train_fail(SE_ERR_HIP, "persistent GRU backward launch failed");
}

#define TCHECK(call)                                                                                             \
    do {                                                                                                         \
        if ((call) != hipSuccess) return train_fail(SE_ERR_HIP, "%s failed (%s:%d)", #call, __FILE__, __LINE__); \
    } while (0)

int se_sig_create(int n_fft, int win, int hop, int K, int device, se_sig **out) {
    if (!out || n_fft <= 0 || n_fft % 2 || win <= 0 || win > n_fft || hop <= 0 || K <= 0 || K % hop) return train_fail(SE_ERR_ARG, "bad STFT geometry");
    se_sig *g = new se_sig();
    g->device = device; g->N = n_fft; g->win = win; g->hop = hop; g->K = K; g->T = 1 + K / hop; g->F = n_fft / 2 + 1;
    g->plan.N = n_fft;
    g->plan.npass = fft_plan(n_fft / 2, g->plan.radices);
    if (!g->plan.npass || g->plan.npass > se::kMaxRadices) { delete g; return train_fail(SE_ERR_ARG, "n_fft must factor into 2s and 5s"); }
    if (hipSetDevice(device) != hipSuccess) { delete g; return train_fail(SE_ERR_HIP, "hipSetDevice failed"); }
    const int N = n_fft, T = g->T;
    std::vector<float> w(N, 0.0f), tw(2 * (size_t)N), env(K, 0.0f);
    const int left = (N - win) / 2;
    for (int i = 0; i < win; i++) w[left + i] = (float)(0.54 - 0.46 * cos(2.0 * M_PI * i / win));
    for (int i = 0; i < N; i++) { tw[2 * i] = (float)cos(2.0 * M_PI * i / N); tw[2 * i + 1] = (float)-sin(2.0 * M_PI * i / N); }
    for (int i = 0; i < K; i++) {
        const int pos = N / 2 + i;
        float sum = 0;
        for (int t = 0; t < T; t++) { const int n = pos - t * hop; if (n >= 0 && n < N) sum += w[n] * w[n]; }
        env[i] = sum;
    }
    auto up = [&](float *&d, const std::vector<float> &h) {
        return hipMalloc(reinterpret_cast<void **>(&d), h.size() * sizeof(float)) == hipSuccess &&
               hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
    };
    if (!up(g->window, w) || !up(g->tw, tw) || !up(g->env, env)) { se_sig_destroy(g); return train_fail(SE_ERR_HIP, "table upload failed"); }
    se::aux_set_fft_lds((int)se::stft_lds_bytes(K, N), (int)se::istft_lds_bytes(T, N));
    *out = g;
    return SE_OK;
}

void se_sig_destroy(se_sig *g) {
    if (!g) return;
    (void)hipSetDevice(g->device);
    if (g->window) (void)hipFree(g->window);
    if (g->tw) (void)hipFree(g->tw);
    if (g->env) (void)hipFree(g->env);
    delete g;
}

int se_sig_stft(se_sig *g, const float *wav, int B, int M, int64_t L, int64_t off0, int64_t seg_off, int nseg, float *spec, void *stream) {
    if (!g || !wav || !spec || B <= 0 || M <= 0 || nseg <= 0) return train_fail(SE_ERR_ARG, "bad argument");
    se::StftArgs a{};
    a.src = wav; a.strideB = (long)M * L; a.strideM = L; a.M = M; a.off = off0; a.L = L;
    a.K = g->K; a.T = g->T; a.F = g->F; a.hop = g->hop;
    a.spec = reinterpret_cast<cf2 *>(spec); a.sR = (long)g->T * g->F; a.sT = g->F; a.sF = 1;
    a.window = g->window; a.tw = reinterpret_cast<const cf2 *>(g->tw); a.plan = g->plan;
    a.seg_off = seg_off; a.seg_spec = (long)B * M * g->T * g->F;
    se::launch_k_stft(dim3(B * M, nseg), se::stft_lds_bytes(g->K, g->N), static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? SE_OK : train_fail(SE_ERR_HIP, "stft launch failed");
}

int se_sig_istft(se_sig *g, const float *spec, int rows, float *wav, void *stream) {
    if (!g || !spec || !wav || rows <= 0) return train_fail(SE_ERR_ARG, "bad argument");
    se::IstftArgs a{};
    a.spec = reinterpret_cast<const cf2 *>(spec); a.sR = (long)g->T * g->F; a.sT = g->F; a.sF = 1;
    a.K = g->K; a.T = g->T; a.F = g->F; a.hop = g->hop;
    a.wav = wav; a.wav_ld = g->K; a.window = g->window; a.env = g->env; a.tw = reinterpret_cast<const cf2 *>(g->tw); a.plan = g->plan;
    se::launch_k_istft(dim3(rows, 1), se::istft_lds_bytes(g->T, g->N), static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? SE_OK : train_fail(SE_ERR_HIP, "istft launch failed");
}

int se_train_ola_fwd(se_sig *g, const float *yseg, float *out, int B, int64_t L, int64_t skip, void *stream) {
    if (!g || !yseg || !out || B <= 0 || L <= 0) return train_fail(SE_ERR_ARG, "bad argument");
    hipLaunchKernelGGL(se::k_tola_fwd, dim3((unsigned)((L + 255) / 256), B), dim3(256), 0, static_cast<hipStream_t>(stream), yseg, out, B, g->K, (long)L, (long)skip);
    return hipGetLastError() == hipSuccess ? SE_OK : train_fail(SE_ERR_HIP, "launch failed");
}

int se_train_ola_bwd(se_sig *g, const float *dout, float *gseg, int B, int nseg, int64_t L, int64_t skip, void *stream) {
    if (!g || !dout || !gseg || B <= 0 || nseg <= 0) return train_fail(SE_ERR_ARG, "bad argument");
    hipLaunchKernelGGL(se::k_tola_bwd, dim3((g->K + 255) / 256, B, nseg), dim3(256), 0, static_cast<hipStream_t>(stream), dout, g->env, gseg, B, g->K, (long)L, (long)skip);
    return hipGetLastError() == hipSuccess ? SE_OK : train_fail(SE_ERR_HIP, "launch failed");
}

int se_train_feat(const float *spec, float *feat, int S, int M, int T, int F, int atan2_phase, void *stream) {
    if (!spec || !feat || S <= 0) return train_fail(SE_ERR_ARG, "bad argument");
    hipLaunchKernelGGL(se::k_tfeat, dim3((T * F + 255) / 256, S), dim3(256), 0, static_cast<hipStream_t>(stream), reinterpret_cast<const cf2 *>(spec), feat, M, T * F, atan2_phase);
    return hipGetLastError() == hipSuccess ? SE_OK : train_fail(SE_ERR_HIP, "launch failed");
}

int se_train_mask_fwd(const float *x, const float *spec, float *Y, int S, int M, int T, int F, void *stream) {
    if (!x || !spec || !Y || S <= 0) return train_fail(SE_ERR_ARG, "bad argument");
    hipLaunchKernelGGL(se::k_tmask_fwd, dim3((T * F + 255) / 256, S), dim3(256), 0, static_cast<hipStream_t>(stream), x, reinterpret_cast<const cf2 *>(spec),
                       reinterpret_cast<cf2 *>(Y), M, T * F);
    return hipGetLastError() == hipSuccess ? SE_OK : train_fail(SE_ERR_HIP, "launch failed");
}

int se_train_mask_bwd(const float *dY, const float *x, const float *spec, float *dx, int S, int M, int T, int F, int n_fft, void *stream) {
    if (!dY || !x || !spec || !dx || S <= 0) return train_fail(SE_ERR_ARG, "bad argument");
    hipLaunchKernelGGL(se::k_tmask_bwd, dim3((T * F + 255) / 256, S), dim3(256), 0, static_cast<hipStream_t>(stream), reinterpret_cast<const cf2 *>(dY), x,
                       reinterpret_cast<const cf2 *>(spec), dx, M, T, F, 1.0f / (float)n_fft);
    return hipGetLastError() == hipSuccess ? SE_OK : train_fail(SE_ERR_HIP, "launch failed");
}

int se_train_gln_fwd(const float *x, int64_t xS, int64_t xC, int64_t xT, float *y, int64_t yS, int64_t yC, int64_t yT, const float *w, const float *b,
                     float *stats, int S, int C, int T, int Fi, int Fo, int mode, int act, int eps_mode, void *stream) {
    if (!x || !y || !w || !b || !stats || S <= 0 || Fo < Fi) return train_fail(SE_ERR_ARG, "bad argument");
    se::TGlnArgs a{};
    a.x = x; a.xS = xS; a.xC = xC; a.xT = xT; a.y = y; a.yS = yS; a.yC = yC; a.yT = yT; a.w = w; a.b = b; a.stats = stats;
    a.C = C; a.T = T; a.Fi = Fi; a.Fo = Fo; a.mode = mode; a.act = act; a.eps_mode = eps_mode;
    hipLaunchKernelGGL(se::k_tgln_fwd, dim3(S), dim3(se::kTT), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? SE_OK : train_fail(SE_ERR_HIP, "launch failed");
}

int se_train_gln_bwd(const float *dy, int64_t dS, int64_t dC, int64_t dT, const float *x, int64_t xS, int64_t xC, int64_t xT, float *dx, const float *w,
                     const float *stats, float *dw_part, float *db_part, float *dpre_part, int S, int C, int T, int Fi, int mode, int act,
                     int eps_mode, void *stream) {
    if (!dy || !x || !dx || !w || !stats || S <= 0) return train_fail(SE_ERR_ARG, "bad argument");
    se::TGlnArgs a{};
    a.dy = dy; a.dS = dS; a.dC = dC; a.dT = dT; a.x = x; a.xS = xS; a.xC = xC; a.xT = xT; a.y = dx; a.w = w;
    a.stats = const_cast<float *>(stats); a.dw_part = dw_part; a.db_part = db_part; a.dpre_part = dpre_part;
    a.C = C; a.T = T; a.Fi = Fi; a.Fo = Fi; a.mode = mode; a.act = act; a.eps_mode = eps_mode;
    if (mode) hipLaunchKernelGGL(se::k_tgln_bwd_d, dim3(S), dim3(se::kTT), 0, static_cast<hipStream_t>(stream), a);
    else hipLaunchKernelGGL(se::k_tgln_bwd_c, dim3(S), dim3(se::kTT), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? SE_OK : train_fail(SE_ERR_HIP, "launch failed");
}

int se_train_colsum(const float *p0, float *o0, int n0, const float *p1, float *o1, int n1, const float *p2, float *o2, int n2, int R, int accumulate,
                    void *stream) {
    if (!p0 || !o0 || n0 <= 0 || R <= 0) return train_fail(SE_ERR_ARG, "bad argument");
    se::TColsumArgs a{};
    a.part[0] = p0; a.out[0] = o0; a.n[0] = n0; a.part[1] = p1; a.out[1] = o1; a.n[1] = n1; a.part[2] = p2; a.out[2] = o2; a.n[2] = n2;
    a.R = R; a.accumulate = accumulate;
    const int nmax = std::max(n0, std::max(p1 ? n1 : 0, p2 ? n2 : 0));
    hipLaunchKernelGGL(se::k_colsum, dim3((nmax + 255) / 256, p2 ? 3 : (p1 ? 2 : 1)), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? SE_OK : train_fail(SE_ERR_HIP, "launch failed");
}

/* out[j] (+)= sum_r x[r][j] for a tall matrix: chunks of 64 rows into ws[ceil(R/64)][n], then a fixed-order fold */
int se_train_colsum_tall(const float *x, int64_t R, int n, float *ws, float *out, int accumulate, void *stream) {
    if (!x || !ws || !out || R <= 0 || n <= 0) return train_fail(SE_ERR_ARG, "bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nch = (int)((R + 63) / 64);
    hipLaunchKernelGGL(se::k_colsum_chunks, dim3((n + 255) / 256, nch), dim3(256), 0, st, x, ws, (int)R, n);
    se::TColsumArgs a{};
    a.part[0] = ws; a.out[0] = out; a.n[0] = n; a.R = nch; a.accumulate = accumulate;
    hipLaunchKernelGGL(se::k_colsum, dim3((n + 255) / 256, 1), dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? SE_OK : train_fail(SE_ERR_HIP, "launch failed");
}

int se_train_skip_fwd(const float *uv, const float *z, const float *nw, const float *nb, float *out, float *stats, int S, int Co, int T, int F, int act,
                      int eps_mode, void *stream) {
    if (!uv || !z || !nw || !nb || !out || !stats || S <= 0) return train_fail(SE_ERR_ARG, "bad argument");
    se::TSkipArgs a{};
    a.uv = uv; a.z = z; a.nw = nw; a.nb = nb; a.out = out; a.stats = stats; a.Co = Co; a.T = T; a.F = F; a.act = act; a.eps_mode = eps_mode;
    hipLaunchKernelGGL(se::k_tskip_fwd, dim3(S), dim3(se::kTT), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? SE_OK : train_fail(SE_ERR_HIP, "launch failed");
}

int se_train_skip_bwd(const float *dout, const float *uv, const float *z, const float *nw, const float *nb, const float *stats, float *duv, float *dz,
                      float *dnw_part, float *dnb_part, float *dbias_part, int S, int Co, int T, int F, int act, int eps_mode, void *stream) {
    if (!dout || !uv || !z || !nw || !nb || !stats || !duv || !dz || !dnw_part || !dnb_part || !dbias_part || S <= 0) return train_fail(SE_ERR_ARG, "bad argument");
    se::TSkipArgs a{};
    a.dout = dout; a.uv = uv; a.z = z; a.nw = nw; a.nb = nb; a.stats = const_cast<float *>(stats); a.duv = duv; a.dz = dz;
    a.dnw_part = dnw_part; a.dnb_part = dnb_part; a.dbias_part = dbias_part; a.Co = Co; a.T = T; a.F = F; a.act = act; a.eps_mode = eps_mode;
    hipLaunchKernelGGL(se::k_tskip_bwd, dim3(S), dim3(se::kTT), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? SE_OK : train_fail(SE_ERR_HIP, "launch failed");
}

int se_train_gate_fwd(const float *tg, const float *w, const float *b, float *y, int64_t yS, int64_t yC, int64_t yT, float *stats, int S, int C, int T, int F,
                      int eps_mode, void *stream) {
    if (!tg || !w || !b || !y || !stats || S <= 0) return train_fail(SE_ERR_ARG, "bad argument");
    se::TGateArgs a{};
    a.tg = tg; a.w = w; a.b = b; a.y = y; a.stats = stats; a.C = C; a.T = T; a.F = F; a.eps_mode = eps_mode; a.yS = yS; a.yC = yC; a.yT = yT;
    hipLaunchKernelGGL(se::k_tgate_fwd, dim3(S), dim3(se::kTT), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? SE_OK : train_fail(SE_ERR_HIP, "launch failed");
}

int se_train_gate_bwd(const float *dy, int64_t dS, int64_t dC, int64_t dT, const float *tg, const float *w, const float *stats, float *dtg, float *dw_part,
                      float *db_part, float *dbias_part, int S, int C, int T, int F, int eps_mode, void *stream) {
    if (!dy || !tg || !w || !stats || !dtg || !dw_part || !db_part || !dbias_part || S <= 0) return train_fail(SE_ERR_ARG, "bad argument");
    se::TGateArgs a{};
    a.dy = dy; a.tg = tg; a.w = w; a.stats = const_cast<float *>(stats); a.dtg = dtg; a.dw_part = dw_part; a.db_part = db_part; a.dbias_part = dbias_part;
    a.C = C; a.T = T; a.F = F; a.eps_mode = eps_mode; a.yS = dS; a.yC = dC; a.yT = dT;
    hipLaunchKernelGGL(se::k_tgate_bwd, dim3(S), dim3(se::kTT), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? SE_OK : train_fail(SE_ERR_HIP, "launch failed");
}

int se_train_elu_bwd(float *da, const float *act, float *dpre_part, int S, int C, int T, int F, void *stream) {
    if (!da || !act || !dpre_part || S <= 0) return train_fail(SE_ERR_ARG, "bad argument");
    hipLaunchKernelGGL(se::k_telu_bwd, dim3(S), dim3(se::kTT), 0, static_cast<hipStream_t>(stream), da, act, dpre_part, C, T * F);
    return hipGetLastError() == hipSuccess ? SE_OK : train_fail(SE_ERR_HIP, "launch failed");
}

/* mode 0: y = act(conv5x5(x | xprev)); mode 1: dx from dy; mode 2: dw_part [S][C*C*25] from dy and x | xprev */
int se_train_pre5(int mode, const float *x, const float *xprev, const float *w, const float *bias, const float *dy, float *out, int S, int C, int T, int F,
                  int fd, int act, void *stream) {
    if (!w || !out || S <= 0 || C <= 0 || C > 8 || fd <= 0) return train_fail(SE_ERR_ARG, "bad argument (C <= 8)");
    se::TPre5Args a{};
    a.x = x; a.xprev = xprev; a.w = w; a.bias = bias; a.dy = dy; a.C = C; a.T = T; a.F = F; a.fd = fd; a.act = act;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (mode == 0) {
        if (!x || !bias) return train_fail(SE_ERR_ARG, "null argument");
        a.y = out;
        hipLaunchKernelGGL(se::k_pre5_fwd, dim3((T * F + 255) / 256, S), dim3(256), 0, st, a);
    } else if (mode == 1) {
        if (!dy) return train_fail(SE_ERR_ARG, "null argument");
        a.dx = out;
        hipLaunchKernelGGL(se::k_pre5_dx, dim3((T * F + 255) / 256, S), dim3(256), 0, st, a);
    } else {
        if (!dy || !x) return train_fail(SE_ERR_ARG, "null argument");
        a.dw_part = out;
        hipLaunchKernelGGL(se::k_pre5_dw, dim3(C * C, S), dim3(256), 0, st, a);
    }
    return hipGetLastError() == hipSuccess ? SE_OK : train_fail(SE_ERR_HIP, "launch failed");
}

int se_train_add3(float *dst, const float *a, const float *b, int64_t n, void *stream) {
    if (!dst || !a || !b || n <= 0) return train_fail(SE_ERR_ARG, "bad argument");
    hipLaunchKernelGGL(se::k_tadd3, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), dst, a, b, (long)n);
    return hipGetLastError() == hipSuccess ? SE_OK : train_fail(SE_ERR_HIP, "launch failed");
}

int se_train_add(float *dst, const float *src, int64_t n, void *stream) {
    if (!dst || !src || n <= 0) return train_fail(SE_ERR_ARG, "bad argument");
    hipLaunchKernelGGL(se::k_tadd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), dst, src, (long)n);
    return hipGetLastError() == hipSuccess ? SE_OK : train_fail(SE_ERR_HIP, "launch failed");
}

int se_train_gru_hprev(const float *out, const float *h0, float *hp, int B, int T, int H, int Tseg, int64_t ldN, int64_t ldB, void *stream) {
    if (!out || !h0 || !hp || B <= 0 || T <= 0) return train_fail(SE_ERR_ARG, "bad argument");
    hipLaunchKernelGGL(se::k_gru_hprev, dim3((H + 255) / 256, T, B), dim3(256), 0, static_cast<hipStream_t>(stream), out, h0, hp, B, T, H, Tseg, (long)ldN, (long)ldB);
    return hipGetLastError() == hipSuccess ? SE_OK : train_fail(SE_ERR_HIP, "launch failed");
}

}  // extern "C"

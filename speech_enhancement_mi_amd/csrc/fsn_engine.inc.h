// fsn_engine.inc.h - FullSubNet engine behind the fsn_* C ABI (include/se_engine.h); included at the end of se_engine.hip
// (one translation unit, so the kernels in the shared headers are defined once).
// Reference: FullSubNet.forward / realtime_process(train=False), fullsubnet.py:769-824, 903-961.

struct fsn_engine {
    fsn_config c{};
    int device = 0, T = 0, F = 0, M = 0, K = 0, N = 0, Kp = 0, SI = 0, NL = 0;
    std::string err;
    std::map<std::string, std::vector<float>> params;
    bool weights_ready = false;
    se_engine *sig = nullptr;  // STFT/iSTFT tables and launches are shared with the CRN engine object
    struct Model {
        int in = 0, inp = 0, H = 0, out = 0;
        DevBuf Wp[4], bias[4];      // per layer: bf16x3 planes of [W_ih | W_hh] (K padded to 32s), b_ih + b_hh
        DevBuf fcw, fcw_x, fcb;
        DevBuf h[4][2], c[4];       // state
        int hcur[4]{};
    } fb, sb;
    int B = 0;
    DevBuf spec, maskspec, mag, fb_seq, fb_out, sbin, mask, part_fb, part_sb, mean_fb, mean_sb, denom_fb, denom_sb, yseg;
    int step_fb = 0, step_sb = 0, have_fb = 0, have_sb = 0, nslot_fb = 0, nslot_sb = 0;
    // realtime_process: the full-band model of window n + 1 runs on `side` while the sub-band model of window n runs on the caller's stream
    // (the full-band recurrence is 42 launches of 32 workgroups per window, 11 % of the serial time, latency-bound: profiles/r03_fsn_*)
    hipStream_t side = nullptr;
    hipEvent_t ev_ready[2]{}, ev_consumed[2]{}, ev_done[2]{}, ev_fork = nullptr;
    // ... and inside stage B the sub-band model's layer 1 runs one time step behind layer 0 on `side2`: both launches are 4.7 rounds of
    // one-workgroup-per-CU tiles, the second fills the first one's last round
    hipStream_t side2 = nullptr;
    std::vector<hipEvent_t> ev_l0, ev_l1;  // per time step: layer 0 / layer 1 (+ its output layer) done
    int pipeline = 1;     // SE_FSN_PIPELINE=0: one stream, stage after stage (read at fsn_create)
    int lstm_big = 1;     // SE_FSN_BIG=0: keep the 128 x 128 step tiles where the 256-row x 64-unit tile would be picked (read at fsn_create)
};

namespace {

thread_local std::string g_fsn_create_error;

int ffail(fsn_engine *e, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (e) e->err = buf;
    else g_fsn_create_error = buf;
    return code;
}

#define FHIP(e, call)                                                                                                  \
    do {                                                                                                               \
        hipError_t _st = (call);                                                                                       \
        if (_st != hipSuccess) return ffail(e, SE_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_st), __FILE__, __LINE__); \
    } while (0)

int falloc(fsn_engine *e, DevBuf &b, size_t n) {
    if (b.p && b.n >= n) return 0;
    if (b.p) FHIP(e, hipFree(b.p));
    b.p = nullptr; b.n = 0;
    FHIP(e, hipMalloc(reinterpret_cast<void **>(&b.p), (n ? n : 1) * sizeof(float)));
    b.n = n;
    return 0;
}

int fupload(fsn_engine *e, DevBuf &b, const std::vector<float> &h) {
    int rc = falloc(e, b, h.size());
    if (rc) return rc;
    FHIP(e, hipMemcpy(b.p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

int fupload_planes(fsn_engine *e, DevBuf &b, const std::vector<float> &w) {  // three bf16 planes of w
    const size_t n = w.size();
    std::vector<uint16_t> planes(3 * n);
    for (size_t i = 0; i < n; i++) {
        const float x = w[i];
        const uint16_t h = bf16_rne(x);
        const float r1 = x - bf16_to_f32(h);
        const uint16_t m = bf16_rne(r1);
        planes[i] = h; planes[n + i] = m; planes[2 * n + i] = bf16_rne(r1 - bf16_to_f32(m));
    }
    int rc = falloc(e, b, (3 * n + 1) / 2);
    if (rc) return rc;
    FHIP(e, hipMemcpy(b.p, planes.data(), planes.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    return 0;
}

const std::vector<float> *fparam(fsn_engine *e, const std::string &key, size_t expect) {
    auto it = e->params.find(key);
    if (it == e->params.end()) { ffail(e, SE_ERR_PARAM_MISSING, "parameter %s was never loaded", key.c_str()); return nullptr; }
    if (it->second.size() != expect) { ffail(e, SE_ERR_SHAPE, "parameter %s has %zu elements, expected %zu", key.c_str(), it->second.size(), expect); return nullptr; }
    return &it->second;
}

int fsn_prepare(fsn_engine *e) {
    if (e->weights_ready) return 0;
    struct { fsn_engine::Model *m; const char *name; } models[2] = {{&e->fb, "fb_model"}, {&e->sb, "sb_model"}};
    for (auto &mm : models) {
        fsn_engine::Model &m = *mm.m;
        const int H = m.H, Hp = (H + 31) & ~31;
        for (int l = 0; l < e->NL; l++) {
            const std::string p = std::string(mm.name) + ".sequence_model.", s = std::to_string(l);
            const int in = l == 0 ? m.in : H, inp = l == 0 ? m.inp : Hp;
            auto *wih = fparam(e, p + "weight_ih_l" + s, 4 * (size_t)H * in);
            auto *whh = fparam(e, p + "weight_hh_l" + s, 4 * (size_t)H * H);
            auto *bih = fparam(e, p + "bias_ih_l" + s, 4 * (size_t)H);
            auto *bhh = fparam(e, p + "bias_hh_l" + s, 4 * (size_t)H);
            if (!wih || !whh || !bih || !bhh) return SE_ERR_PARAM_MISSING;
            const int Kt = inp + Hp;
            std::vector<float> cat((size_t)4 * H * Kt, 0.0f), bias(4 * (size_t)H);
            for (int r = 0; r < 4 * H; r++) {
                for (int k = 0; k < in; k++) cat[(size_t)r * Kt + k] = (*wih)[(size_t)r * in + k];
                for (int k = 0; k < H; k++) cat[(size_t)r * Kt + inp + k] = (*whh)[(size_t)r * H + k];
                bias[r] = (*bih)[r] + (*bhh)[r];
            }
            int rc;
            if ((rc = fupload_planes(e, m.Wp[l], cat)) || (rc = fupload(e, m.bias[l], bias))) return rc;
        }
        auto *fw = fparam(e, std::string(mm.name) + ".fc_output_layer.weight", (size_t)m.out * H);
        auto *fbv = fparam(e, std::string(mm.name) + ".fc_output_layer.bias", m.out);
        if (!fw || !fbv) return SE_ERR_PARAM_MISSING;
        int rc;
        if ((rc = fupload(e, m.fcw, *fw)) || (rc = fupload_planes(e, m.fcw_x, *fw)) || (rc = fupload(e, m.fcb, *fbv))) return rc;
    }
    e->weights_ready = true;
    return 0;
}

int fsn_lstm_step(fsn_engine *e, fsn_engine::Model &m, int l, const float *x, long ldx, int K1, int K1p, int R, float *hseq, long ldseq, hipStream_t st) {
    const int hc = m.hcur[l];
    LstmStepArgs a{x, ldx, K1, K1p, m.h[l][hc].p, reinterpret_cast<const __bf16 *>(m.Wp[l].p), m.bias[l].p, m.c[l].p, m.h[l][hc ^ 1].p, hseq, ldseq, R, m.H};
#ifdef SE_LSTM_STAMPS
    static unsigned long long *stamps = nullptr;
    static int nlaunch = 0;
    if (!stamps) { (void)hipMalloc(&stamps, 64); (void)hipMemset(stamps, 0, 64); }
    a.stamps = stamps;
    if (R >= 8192 && ++nlaunch % 997 == 0) {
        (void)hipStreamSynchronize(st);
        unsigned long long h[7];
        (void)hipMemcpy(h, stamps, sizeof h, hipMemcpyDeviceToHost);
        if (e->lstm_big == 1)
            fprintf(stderr, "[stamps big] before layer %d: nck %llu per-chunk cycles: barrier1 %.0f store+wait %.0f barrier2 %.0f gate0 %.0f dma+issue %.0f rest %.0f\n", l, h[6],
                    (double)h[0] / h[6], (double)h[1] / h[6], (double)h[2] / h[6], (double)h[3] / h[6], (double)h[4] / h[6], (double)h[5] / h[6]);
        else
            fprintf(stderr, "[stamps] before layer %d: nck %llu per-chunk cycles: barrier1 %.0f vmwait %.0f stage %.0f barrier2 %.0f issue+mfma %.0f\n", l, h[5],
                    (double)h[0] / h[5], (double)h[1] / h[5], (double)h[2] / h[5], (double)h[3] / h[5], (double)h[4] / h[5]);
    }
#endif
    // big tile (256 rows x 64 units) where it fills the chip: the sub-band model at B >= 32 streams
    if (e->lstm_big == 1 && m.H % kLbU == 0 && R >= 32 * kLbM) {
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_lstm_step_big<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_lstm_step_big<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr = true;
        }
        const dim3 gb(m.H / kLbU, (R + kLbM - 1) / kLbM);
        const int PL = e->c.precision == 2 ? 2 : 3;
        const size_t lds = (size_t)(PL == 2 ? 4 : 3) * PL * kLbPlane * sizeof(__bf16);  // bf16x3: both operands double-buffered
        if (PL == 2) hipLaunchKernelGGL((k_lstm_step_big<2>), gb, dim3(512), lds, st, a);
        else hipLaunchKernelGGL((k_lstm_step_big<3>), gb, dim3(512), lds, st, a);
        m.hcur[l] = hc ^ 1;
        return 0;
    }
    const dim3 grid((m.H + 31) / 32, (R + kGemmBM - 1) / kGemmBM);
    if (e->c.precision == 2) hipLaunchKernelGGL((k_lstm_step_x6<2>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_lstm_step_x6<3>), grid, dim3(256), 0, st, a);
    m.hcur[l] = hc ^ 1;
    return 0;
}

// forward on device.  Spectrum given by (re, im) pointers + strides in float units for (b, m, t, f).
// Stage A of a window: |X|, its CumLayerNorm, the full-band LSTM and its output layer -> e->mag, e->fb_out (read by stage B's unfold only)
int fsn_stage_fb(fsn_engine *e, const float *re, const float *im, long sB, long sM, long sT, long sF, hipStream_t st) {
    const int B = e->B, T = e->T, F = e->F, M = e->M, Kp = e->Kp;
    {  // |X| + CumLayerNorm of the full-band input (fullsubnet.py:782-788)
        FsnMagArgs a{re, im, sB, sM, sT, sF, e->mag.p, e->part_fb.p, M, T, F, Kp};
        hipLaunchKernelGGL(k_fsn_mag, dim3(e->nslot_fb, B), dim3(256), 0, st, a);
        const float alpha = (float)e->step_fb / (float)(e->step_fb + 1);
        hipLaunchKernelGGL(k_fsn_runmean, dim3((B + 255) / 256), dim3(256), 0, st, e->part_fb.p, e->nslot_fb, (double)M * T * F, e->mean_fb.p,
                           e->denom_fb.p, B, e->have_fb ? 0 : 1, alpha);
        e->have_fb = 1;
        e->step_fb = std::min(e->step_fb + 1, 80);
        hipLaunchKernelGGL(k_fsn_scale, dim3(16, B), dim3(256), 0, st, e->mag.p, (long)T * Kp, e->denom_fb.p);
        FHIP(e, hipGetLastError());
    }
    for (int t = 0; t < T; t++) {  // full-band LSTM (2 layers interleaved per step), fullsubnet.py:789
        for (int l = 0; l < e->NL; l++) {
            const bool last = l + 1 == e->NL;
            if (l == 0) fsn_lstm_step(e, e->fb, 0, e->mag.p + (long)t * Kp, (long)T * Kp, Kp, Kp, B, last ? e->fb_seq.p + (long)t * e->fb.H : nullptr, (long)T * e->fb.H, st);
            else fsn_lstm_step(e, e->fb, l, e->fb.h[l - 1][e->fb.hcur[l - 1]].p, e->fb.H, e->fb.H, (e->fb.H + 31) & ~31, B,
                               last ? e->fb_seq.p + (long)t * e->fb.H : nullptr, (long)T * e->fb.H, st);
        }
    }
    FHIP(e, hipGetLastError());
    {  // fc_output_layer + ReLU (fullsubnet.py:288-290): [B*T, H] -> [B*T, F]
        GemmX6Args g{e->fb_seq.p, reinterpret_cast<const __bf16 *>(e->fb.fcw_x.p), e->fb.fcb.p, e->fb_out.p, B * T, F, e->fb.H, (long)e->fb.H, (long)F, 1};
        hipLaunchKernelGGL(k_gemm_x<3>, dim3((F + kGemmBN - 1) / kGemmBN, (B * T + kGemmBM - 1) / kGemmBM), dim3(256), 0, st, g);
    }
    FHIP(e, hipGetLastError());
    return 0;
}

// Stage B: sub-band input, its CumLayerNorm, the sub-band LSTM over B*F rows, mask.  `consumed` (optional) is recorded once e->mag / e->fb_out
// have been read, i.e. when the next window's stage A may overwrite them.
int fsn_stage_sb(fsn_engine *e, const float *re, const float *im, long sB, long sT, long sF, float *crm_out, cf2 *spec_out,
                 long oB, long oT, long oF, hipStream_t st, hipEvent_t consumed) {
    const int B = e->B, T = e->T, F = e->F, Kp = e->Kp, SI = e->SI;
    const int R = B * F;
    {  // sub-band input + its CumLayerNorm (fullsubnet.py:796-802)
        FsnUnfoldArgs a{e->mag.p, e->fb_out.p, e->sbin.p, e->part_sb.p, B, T, F, Kp, e->c.sb_neighbors, SI};
        hipLaunchKernelGGL(k_fsn_unfold, dim3(e->nslot_sb, B), dim3(256), 0, st, a);
        if (consumed) FHIP(e, hipEventRecord(consumed, st));
        const float alpha = (float)e->step_sb / (float)(e->step_sb + 1);
        hipLaunchKernelGGL(k_fsn_runmean, dim3((B + 255) / 256), dim3(256), 0, st, e->part_sb.p, e->nslot_sb, (double)F * SI * T, e->mean_sb.p,
                           e->denom_sb.p, B, e->have_sb ? 0 : 1, alpha);
        e->have_sb = 1;
        e->step_sb = std::min(e->step_sb + 1, 80);
        hipLaunchKernelGGL(k_fsn_scale_sb, dim3(2048), dim3(256), 0, st, e->sbin.p, B, T, F, SI, e->denom_sb.p);
        FHIP(e, hipGetLastError());
    }
    // two layers (the reference configuration) on two streams: layer 0 of step t + 1 next to layer 1 of step t.  h of layer 0 ping-pongs
    // between two buffers: step t + 2 of layer 0 overwrites what layer 1 of step t reads, hence ev_l1[t].
    const bool wave = consumed != nullptr && e->side2 && e->NL == 2 && (int)e->ev_l0.size() >= T;
    for (int t = 0; t < T; t++) {  // sub-band LSTM over B*F rows + Linear(H -> 2)  (fullsubnet.py:812-814)
        for (int l = 0; l < e->NL; l++) {
            hipStream_t sl = wave && l == 1 ? e->side2 : st;
            if (wave && l == 0 && t >= 2) FHIP(e, hipStreamWaitEvent(st, e->ev_l1[t - 2], 0));
            if (wave && l == 1) FHIP(e, hipStreamWaitEvent(sl, e->ev_l0[t], 0));
            if (l == 0) fsn_lstm_step(e, e->sb, 0, e->sbin.p + (long)t * R * SI, SI, SI, (SI + 31) & ~31, R, nullptr, 0, sl);
            else fsn_lstm_step(e, e->sb, l, e->sb.h[l - 1][e->sb.hcur[l - 1]].p, e->sb.H, e->sb.H, (e->sb.H + 31) & ~31, R, nullptr, 0, sl);
            if (wave && l == 0) FHIP(e, hipEventRecord(e->ev_l0[t], st));
        }
        const int ll = e->NL - 1;
        hipStream_t so = wave ? e->side2 : st;
        hipLaunchKernelGGL(k_fsn_sbfc, dim3(2048), dim3(256), 0, so, e->sb.h[ll][e->sb.hcur[ll]].p, e->sb.fcw.p, e->sb.fcb.p, e->mask.p, R, e->sb.H, T, t);
        if (wave) FHIP(e, hipEventRecord(e->ev_l1[t], so));
    }
    if (wave) FHIP(e, hipStreamWaitEvent(st, e->ev_l1[T - 1], 0));  // join: the mask needs every step's output
    FHIP(e, hipGetLastError());
    {
        FsnMaskArgs a{e->mask.p, spec_out ? re : nullptr, spec_out ? im : nullptr, sB, sT, sF, spec_out, oB, oT, oF, crm_out, T, F};
        launch_k_fsn_mask(dim3((T * F + 255) / 256, B), st, a);
        FHIP(e, hipGetLastError());
    }
    return 0;
}

int fsn_forward_dev(fsn_engine *e, const float *re, const float *im, long sB, long sM, long sT, long sF, float *crm_out, cf2 *spec_out,
                    long oB, long oT, long oF, hipStream_t st) {
    int rc;
    if ((rc = fsn_stage_fb(e, re, im, sB, sM, sT, sF, st))) return rc;
    return fsn_stage_sb(e, re, im, sB, sT, sF, crm_out, spec_out, oB, oT, oF, st, nullptr);
}

int fsn_reset_on(fsn_engine *e, int batch, hipStream_t st) {
    if (!e || batch <= 0) return ffail(e, SE_ERR_ARG, "batch must be positive");
    FHIP(e, hipSetDevice(e->device));
    int rc = fsn_prepare(e);
    if (rc) return rc;
    const int B = batch, T = e->T, F = e->F, M = e->M, R = B * F;
    e->B = B;
    e->nslot_fb = 8;
    e->nslot_sb = 32;
    // spec holds two windows: stage A of window n + 1 transforms while stage B of window n still masks its spectrum
    if ((rc = falloc(e, e->spec, (size_t)2 * B * M * T * F * 2)) || (rc = falloc(e, e->maskspec, (size_t)B * T * F * 2)) ||
        (rc = falloc(e, e->mag, (size_t)B * T * e->Kp)) || (rc = falloc(e, e->fb_seq, (size_t)B * T * e->fb.H)) ||
        (rc = falloc(e, e->fb_out, (size_t)B * T * F)) || (rc = falloc(e, e->sbin, (size_t)T * R * e->SI)) ||
        (rc = falloc(e, e->mask, (size_t)R * 2 * T)) || (rc = falloc(e, e->part_fb, (size_t)B * e->nslot_fb)) ||
        (rc = falloc(e, e->part_sb, (size_t)B * e->nslot_sb)) || (rc = falloc(e, e->mean_fb, B)) || (rc = falloc(e, e->mean_sb, B)) ||
        (rc = falloc(e, e->denom_fb, B)) || (rc = falloc(e, e->denom_sb, B)))
        return rc;
    FHIP(e, hipMemsetAsync(e->mag.p, 0, (size_t)B * T * e->Kp * sizeof(float), st));  // padding columns must be zero
    for (int l = 0; l < e->NL; l++) {
        for (int p = 0; p < 2; p++) {
            if ((rc = falloc(e, e->fb.h[l][p], (size_t)B * e->fb.H)) || (rc = falloc(e, e->sb.h[l][p], (size_t)R * e->sb.H))) return rc;
            FHIP(e, hipMemsetAsync(e->fb.h[l][p].p, 0, (size_t)B * e->fb.H * sizeof(float), st));
            FHIP(e, hipMemsetAsync(e->sb.h[l][p].p, 0, (size_t)R * e->sb.H * sizeof(float), st));
        }
        if ((rc = falloc(e, e->fb.c[l], (size_t)B * e->fb.H)) || (rc = falloc(e, e->sb.c[l], (size_t)R * e->sb.H))) return rc;
        FHIP(e, hipMemsetAsync(e->fb.c[l].p, 0, (size_t)B * e->fb.H * sizeof(float), st));
        FHIP(e, hipMemsetAsync(e->sb.c[l].p, 0, (size_t)R * e->sb.H * sizeof(float), st));
        e->fb.hcur[l] = e->sb.hcur[l] = 0;
    }
    e->step_fb = e->step_sb = e->have_fb = e->have_sb = 0;
    e->sig->B = B;  // the shared STFT/iSTFT launchers size their grids from B
    return 0;
}

}  // namespace

extern "C" {

const char *fsn_last_error(const fsn_engine *e) { return e ? e->err.c_str() : g_fsn_create_error.c_str(); }

int fsn_create(const fsn_config *cfg, int device, fsn_engine **out) {
    if (!cfg || !out) return ffail(nullptr, SE_ERR_ARG, "null argument");
    *out = nullptr;
    if (cfg->fb_neighbors != 0 || cfg->look_ahead != 0) return ffail(nullptr, SE_ERR_ARG, "only fb_num_neighbors = 0, look_ahead = 0 (config.yaml:154-157) are supported");
    if (cfg->num_layers < 1 || cfg->num_layers > 4) return ffail(nullptr, SE_ERR_ARG, "num_layers %d out of range", cfg->num_layers);
    if (cfg->fb_hidden % 8 || cfg->sb_hidden % 8 || cfg->fb_hidden <= 0 || cfg->sb_hidden <= 0) return ffail(nullptr, SE_ERR_ARG, "hidden sizes must be positive multiples of 8");
    if ((2 * cfg->sb_neighbors + 2) % 4) return ffail(nullptr, SE_ERR_ARG, "sub-band input width 2*sb_num_neighbors+2 must be a multiple of 4");
    if (cfg->precision != 0 && cfg->precision != 2) return ffail(nullptr, SE_ERR_ARG, "fsn precision %d unknown (0 fp32-accurate, 2 bf16x3)", cfg->precision);
    // STFT tables / launchers: borrow a CRN engine object configured with the same STFT geometry
    se_config sc{};
    sc.num_levels = 4; for (int i = 0; i < 4; i++) sc.channels[i] = 8;
    sc.num_freqs = cfg->num_freqs; sc.hidden = 16; sc.num_layers = 1; sc.num_inputs = cfg->num_mics; sc.kernel_size = 3;
    sc.n_fft = cfg->n_fft; sc.win = cfg->win; sc.hop = cfg->hop; sc.segment_length = cfg->segment_length; sc.variant = 0;
    se_engine *sig = nullptr;
    int rc = se_create(&sc, device, &sig);
    if (rc) { g_fsn_create_error = std::string("STFT setup: ") + se_last_error(nullptr); return rc; }
    fsn_engine *e = new fsn_engine();
    e->c = *cfg; e->device = device; e->sig = sig;
    e->T = sig->T; e->F = cfg->num_freqs; e->M = cfg->num_mics; e->K = cfg->segment_length; e->N = cfg->n_fft; e->NL = cfg->num_layers;
    e->Kp = (cfg->num_freqs * cfg->num_mics + 31) & ~31;
    e->SI = 2 * cfg->sb_neighbors + 2;
    e->fb.in = cfg->num_freqs * cfg->num_mics; e->fb.inp = e->Kp; e->fb.H = cfg->fb_hidden; e->fb.out = cfg->num_freqs;
    e->sb.in = e->SI; e->sb.inp = (e->SI + 31) & ~31; e->sb.H = cfg->sb_hidden; e->sb.out = 2;
    if (const char *s = getenv("SE_FSN_BIG")) e->lstm_big = atoi(s);
    if (const char *s = getenv("SE_FSN_PIPELINE")) e->pipeline = atoi(s);
    if (e->pipeline) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        bool ok = hipStreamCreateWithPriority(&e->side, hipStreamNonBlocking, hi) == hipSuccess;  // the short full-band launches go first when a CU frees up
        ok = ok && hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; i < 2 && ok; i++)
            ok = hipEventCreateWithFlags(&e->ev_ready[i], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&e->ev_consumed[i], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&e->ev_done[i], hipEventDisableTiming) == hipSuccess;
        ok = ok && hipStreamCreateWithFlags(&e->side2, hipStreamNonBlocking) == hipSuccess;
        e->ev_l0.assign(e->T, nullptr); e->ev_l1.assign(e->T, nullptr);
        for (int t = 0; t < e->T && ok; t++)
            ok = hipEventCreateWithFlags(&e->ev_l0[t], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&e->ev_l1[t], hipEventDisableTiming) == hipSuccess;
        if (!ok) { fsn_destroy(e); return ffail(nullptr, SE_ERR_HIP, "fsn_create: side stream / events: %s", hipGetErrorString(hipGetLastError())); }
    }
    *out = e;
    return SE_OK;
}

void fsn_destroy(fsn_engine *e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipDeviceSynchronize();
    for (fsn_engine::Model *m : {&e->fb, &e->sb}) {
        for (int l = 0; l < 4; l++) { dev_free(m->Wp[l]); dev_free(m->bias[l]); dev_free(m->h[l][0]); dev_free(m->h[l][1]); dev_free(m->c[l]); }
        dev_free(m->fcw); dev_free(m->fcw_x); dev_free(m->fcb);
    }
    for (DevBuf *b : {&e->spec, &e->maskspec, &e->mag, &e->fb_seq, &e->fb_out, &e->sbin, &e->mask, &e->part_fb, &e->part_sb, &e->mean_fb,
                      &e->mean_sb, &e->denom_fb, &e->denom_sb, &e->yseg})
        dev_free(*b);
    if (e->side) (void)hipStreamDestroy(e->side);
    if (e->side2) (void)hipStreamDestroy(e->side2);
    for (hipEvent_t ev : e->ev_l0) if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : e->ev_l1) if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : {e->ev_fork, e->ev_ready[0], e->ev_ready[1], e->ev_consumed[0], e->ev_consumed[1], e->ev_done[0], e->ev_done[1]})
        if (ev) (void)hipEventDestroy(ev);
    se_destroy(e->sig);
    delete e;
}

int fsn_load_param(fsn_engine *e, const char *key, const float *host_data, const int64_t *shape, int ndim) {
    if (!e || !key || !host_data) return ffail(e, SE_ERR_ARG, "null argument");
    const std::string k(key);
    bool ok = false;
    for (const char *mname : {"fb_model.", "sb_model."}) {
        if (k.rfind(mname, 0) != 0) continue;
        const std::string rest = k.substr(9);
        int l = -1, n = 0;
        char what[32];
        if (sscanf(rest.c_str(), "sequence_model.%31[a-z_]%d%n", what, &l, &n) == 2)
            ok = l >= 0 && l < e->NL && (size_t)n == rest.size() &&
                 (!strcmp(what, "weight_ih_l") || !strcmp(what, "weight_hh_l") || !strcmp(what, "bias_ih_l") || !strcmp(what, "bias_hh_l"));
        else ok = rest == "fc_output_layer.weight" || rest == "fc_output_layer.bias";
    }
    if (!ok) return ffail(e, SE_ERR_KEY, "unknown parameter key %s", key);
    size_t cnt = 1;
    for (int i = 0; i < ndim; i++) cnt *= (size_t)shape[i];
    e->params[k].assign(host_data, host_data + cnt);
    e->weights_ready = false;
    return SE_OK;
}

int fsn_reset(fsn_engine *e, int batch) {
    int rc = fsn_reset_on(e, batch, nullptr);
    if (rc) return rc;
    FHIP(e, hipDeviceSynchronize());
    return SE_OK;
}

// FullSubNet.forward: x [B, 2M, F, T] (re x M then im x M) -> crm [B, 2, F, T]
int fsn_forward(fsn_engine *e, const float *x, float *crm, void *stream) {
    if (!e || !x || !crm) return ffail(e, SE_ERR_ARG, "null argument");
    if (e->B <= 0) return ffail(e, SE_ERR_STATE, "fsn_forward before fsn_reset");
    FHIP(e, hipSetDevice(e->device));
    int rc = fsn_prepare(e);
    if (rc) return rc;
    const long F = e->F, T = e->T, M = e->M;
    return fsn_forward_dev(e, x, x + M * F * T, 2 * M * F * T, F * T, 1, T, crm, nullptr, 0, 0, 0, static_cast<hipStream_t>(stream));
}

// FullSubNet.realtime_process(mixture, source, flag, train=False)[0]: mixture [B, M, L] -> [B, L]
int fsn_realtime_process(fsn_engine *e, const float *mixture, int batch, int64_t length, int flag, float *out, void *stream) {
    if (!e || !mixture || !out || batch <= 0 || length <= 0) return ffail(e, SE_ERR_ARG, "bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc;
    if (!flag) { if ((rc = fsn_reset_on(e, batch, st))) return rc; }
    else {
        if (e->B != batch) return ffail(e, SE_ERR_STATE, "flag=True with batch %d but the carried state holds %d streams", batch, e->B);
        FHIP(e, hipSetDevice(e->device));
        if ((rc = fsn_prepare(e))) return rc;
    }
    const long K = e->K, P = K / 2, lead = flag ? 0 : P, Lp = length + lead;
    const long gap = K - (P + Lp % K) % K, Nseg = 2 * (Lp + gap + P) / K;
    if ((rc = falloc(e, e->yseg, (size_t)batch * Nseg * K))) return rc;
    const long F = e->F, T = e->T, M = e->M;
    cf2 *spec = reinterpret_cast<cf2 *>(e->spec.p);
    cf2 *ms = reinterpret_cast<cf2 *>(e->maskspec.p);
    const size_t spec_floats = (size_t)batch * M * T * F * 2;
    const bool piped = e->pipeline && e->side && Nseg > 1;
    hipStream_t sa = piped ? e->side : st;
    if (piped) {  // the side stream starts after whatever the caller's stream holds (reset, the previous call's tail)
        FHIP(e, hipEventRecord(e->ev_fork, st));
        FHIP(e, hipStreamWaitEvent(sa, e->ev_fork, 0));
    }
    for (long n = 0; n < Nseg; n++) {
        const long off = n * P - P - lead;
        const int slot = piped ? (int)(n & 1) : 0;
        const float *sp = e->spec.p + slot * spec_floats;
        if (piped) {
            if (n >= 2) FHIP(e, hipStreamWaitEvent(sa, e->ev_done[slot], 0));          // window n - 2 has masked spectrum[slot]
            if (n >= 1) FHIP(e, hipStreamWaitEvent(sa, e->ev_consumed[slot ^ 1], 0));  // window n - 1 has unfolded mag / fb_out
        }
        if (launch_stft(e->sig, mixture, (long)M * length, length, (int)M, off, length, batch * (int)M, spec + slot * (spec_floats / 2), T * F, F, 1, sa))
            return ffail(e, SE_ERR_HIP, "stft: %s", se_last_error(e->sig));
        if ((rc = fsn_stage_fb(e, sp, sp + 1, 2 * M * T * F, 2 * T * F, 2 * F, 2, sa))) return rc;
        if (piped) {
            FHIP(e, hipEventRecord(e->ev_ready[slot], sa));
            FHIP(e, hipStreamWaitEvent(st, e->ev_ready[slot], 0));
        }
        if ((rc = fsn_stage_sb(e, sp, sp + 1, 2 * M * T * F, 2 * F, 2, nullptr, ms, T * F, F, 1, st, piped ? e->ev_consumed[slot] : nullptr))) return rc;
        if (launch_istft(e->sig, ms, T * F, F, 1, batch, e->yseg.p + n * K, Nseg * K, st)) return ffail(e, SE_ERR_HIP, "istft: %s", se_last_error(e->sig));
        if (piped) FHIP(e, hipEventRecord(e->ev_done[slot], st));
    }
    launch_k_overlap_avg(dim3((unsigned)((length + 255) / 256), batch), st, e->yseg.p, out, (int)Nseg, (int)K, (long)length, lead);
    FHIP(e, hipGetLastError());
    return SE_OK;
}

int fsn_read_tap(fsn_engine *e, const char *name, float *host_out, int64_t capacity, int64_t *count, void *stream) {
    if (!e || !name || !host_out) return ffail(e, SE_ERR_ARG, "null argument");
    if (e->B <= 0) return ffail(e, SE_ERR_STATE, "no forward has run");
    const float *src = nullptr;
    size_t n = 0;
    if (!strcmp(name, "fb_out")) { src = e->fb_out.p; n = (size_t)e->B * e->T * e->F; }          // [B*T][F]
    else if (!strcmp(name, "mean_fb")) { src = e->mean_fb.p; n = e->B; }
    else if (!strcmp(name, "mean_sb")) { src = e->mean_sb.p; n = e->B; }
    else return ffail(e, SE_ERR_KEY, "unknown tap %s", name);
    if (count) *count = (int64_t)n;
    if ((int64_t)n > capacity) return ffail(e, SE_ERR_ARG, "buffer too small: need %zu floats", n);
    FHIP(e, hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    FHIP(e, hipMemcpy(host_out, src, n * sizeof(float), hipMemcpyDeviceToHost));
    return SE_OK;
}

double fsn_flops_per_frame(const fsn_engine *e) {
    if (!e) return 0;
    double mac = 0;
    const fsn_engine::Model *ms[2] = {&e->fb, &e->sb};
    const double rows[2] = {1.0, (double)e->F};
    for (int i = 0; i < 2; i++) {
        const fsn_engine::Model &m = *ms[i];
        for (int l = 0; l < e->NL; l++) mac += rows[i] * e->T * 4.0 * m.H * ((l == 0 ? m.in : m.H) + m.H);
        mac += rows[i] * e->T * (double)m.out * m.H;
    }
    return 2.0 * mac;
}

}  // extern "C"

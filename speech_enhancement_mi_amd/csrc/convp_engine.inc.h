// convp_engine.inc.h - engine side of the plane-layout convolution path (conv_p.hip.h); included by se_engine.hip.
//
// Encoder and decoder of TemporalCRN.forward (CRN.py:463-495) with every MFMA operand tensor kept in HBM as split-bf16
// planes ("P layout") and every pre-norm convolution output as channel-octet-innermost fp32 ("R layout"):
//
//   k_featurize_p -> P            4 x ( k_conv_p -> R + statistics ; k_gln_p -> P )      [last level: -> fp32 GRU input]
//   bottleneck (unchanged GEMM / GRU kernels) -> k_gln2_p -> P
//   3 x ( k_conv_p even / odd -> R parity-planar ; k_conv_p 1x1 statistics pass ; k_conv_p 1x1 + fused skip gate -> P )
//   k_conv_p (both parities merged, 4 GEMM rows) -> R ; k_final_mask_p
//
// The ring slot r of encoder level i is xinP[i] + r * slot_elems (ONE allocation per level, so that the history slot is
// addressable through the same buffer descriptor as the current one).

namespace {

struct ConvPPlan {
    ConvPArgs a{};
    int NT = 0, CO = 1, grid_x = 0;
    size_t lds = 0;
    bool active = false;
    DevBuf wx, bias;
    double flops = 0;
    struct Geo { int NT = 0, tpw = 0, n_wg = 0, grouped = 0; size_t lds = 0; long items = 0; } geo[16];
    int ngeo = 0;
    int npair = 0, terms = 6;
    std::string name;  // launch-site label ("enc3", "skip0", ...): SE_CONVP_NT_<name>=n forces a tiling for experiments
};

struct SkipPPlan {  // k_skip_p: the whole skip gate of one decoder level, one workgroup per stream
    SkipPArgs a{};
    DevBuf wx, cst;
    int KP = 0;
    size_t lds = 0;
    bool active = false;
    double flops = 0;
};

struct PLevel {
    ConvPPlan enc, dec_even, dec_odd, skip, skipm;
    ConvPPlan gate[2];  // CRN_ELU / student: conv_trans * sigmoid(conv_gated) after the encoder convolution, <= 64 channels per launch
    SkipPPlan sk;
    bool dec_merged = false;
};

// GEMM row -> logical row of the caller's weight selector, for the epilogue's register ownership (conv_p.hip.h)
inline int convp_row_logical(int mode, int row) {
    const int mt = row >> 5, rp = row & 31, h = (rp >> 2) & 1, r = (rp & 3) + 4 * (rp >> 3);
    if (mode == kPOutPre) return (mt == 0 && h == 0 && r < 8) ? r : (1 << 20);  // channels 0-7 = registers 0-7 of lane half 0
    if (mode == kPOutP) return mt * 32 + 16 * h + r;                       // registers 0..15 of half h = channels 16 h + r
    if (mode == kPOutBlend || mode == kPOutGate) return 2 * (mt * 16 + 8 * h + (r >> 1)) + (r & 1);  // pairs (residualmask, residual) / (trans, gated) of channel 8 h + r/2
    return row;
}

}  // namespace

struct se_convp_state {
    PLevel pv[SE_MAX_LEVELS];
    DevBuf xinP[SE_MAX_LEVELS];   // [kRing][B][C8][PL][T][F] uint4
    long slot_elems[SE_MAX_LEVELS]{};  // uint4 per ring slot
    DevBuf encR[SE_MAX_LEVELS];   // [B][Co8][T*Fo][8]
    DevBuf encA[SE_MAX_LEVELS];   // CRN_ELU / student: act(conv) in the P layout, the input of the gated 1x1 pair
    // CRN_ELU / student: the three 5x5 frequency-dilated pre-conv blocks on the plane path (CRN_ELU.py:335-340, 375-376)
    bool pre_p = false;
    ConvPPlan pre[3];
    DevBuf pinP[3];               // block inputs, [kRing][B][1][PL][T][F0] (ring: the previous slot's tail = the 4 history frames)
    long pslot_elems = 0;         // uint4 per ring slot
    DevBuf preR, pre_gw[3];       // gated output of a block, R layout [B][1][T*F0][8]; gate weights of each block
    DevBuf decinP[kRing];         // decoder input, P layout
    DevBuf decR[SE_MAX_LEVELS];   // [B][Co8][T*2Fi][8] parity-planar
    DevBuf decP[SE_MAX_LEVELS];   // blended decoder level outputs, P layout
    bool ready = false;
};

namespace {

template <class WSel>
int plan_conv_p(se_engine *e, ConvPPlan &pl, int Ci, int Co, int FP, int Fi, int s, int colpad, int tlo_off, int ngroup, int dil, int St,
                const std::vector<std::array<int, 4>> &taps, WSel wsel, const std::vector<float> &bias_logical, int relu_lo, int relu_hi,
                int act, int out_mode) {
    pl.active = FP > 0;
    if (!pl.active) return 0;
    const int T = e->T, ntap = (int)taps.size(), PL = operand_planes(e->precision);
    const int P = T * FP, tiles = (P + 31) / 32;
    const int CoPad = (Co + 31) / 32 * 32;
    if (CoPad > 128 || CoPad == 96) return fail(e, SE_ERR_ARG, "conv with %d GEMM rows is not supported", Co);
    const int MT = CoPad / 32, NCG = 4 / MT;
    const int C8 = (Ci + 7) / 8;
    int CO = 1;
    if (ntap == 1) CO = C8 >= 4 ? 4 : (C8 >= 2 ? 2 : 1);
    if (C8 % CO) return fail(e, SE_ERR_ARG, "channel octets %d not a multiple of the chunk %d", C8, CO);
    const int nchunk = C8 / CO, npair = (ntap * CO + 1) / 2;
    pl.ngeo = 0;
    static const int nts[] = {1, 2, 3, 4, 6, 8, 10, 12};
    for (int nt : nts) {
        if (!conv_p_has_instance(ntap, nt, CO)) continue;
        const int tpw_max = NCG * nt;
        const int n_wg = (tiles + tpw_max - 1) / tpw_max;
        const int tpw = (tiles + n_wg - 1) / n_wg;
        if ((tpw + NCG - 1) / NCG != nt) continue;  // a smaller instance covers this tiling
        int rows_pos = (tpw * 32 + FP - 1) / FP + 1;
        if (rows_pos > T) rows_pos = T;
        const int grouped = ngroup * rows_pos < rows_pos + (ngroup - 1) * dil;
        const int Rmax = grouped ? ngroup * rows_pos : rows_pos + (ngroup - 1) * dil;
        const long items = (long)PL * CO * Rmax * St;
        if (items > 256L * kPNiMax) continue;
        // whole LDS-DMA instructions: the last one may write zero pieces past `items`, so the allocation covers NI * 256 pieces
        const size_t lds = (size_t)((items + 255) / 256) * 4096;
        if (lds > 160 * 1024) continue;
        ConvPPlan::Geo &g = pl.geo[pl.ngeo++];
        g.NT = nt; g.tpw = tpw; g.n_wg = n_wg; g.grouped = grouped; g.lds = lds; g.items = items;
    }
    if (!pl.ngeo) return fail(e, SE_ERR_ARG, "no conv_p tiling fits (taps %d, Ci %d, Co %d, FP %d, St %d)", ntap, Ci, Co, FP, St);
    ConvPArgs &a = pl.a;
    a.C8 = C8; a.Ci = Ci; a.Co = Co; a.CoPad = CoPad; a.T = T; a.Fi = Fi; a.FP = FP;
    a.s = s; a.colpad = colpad; a.tlo_off = tlo_off; a.ngroup = ngroup; a.dil = dil; a.ntap = ntap;
    a.deint = (s == 2 && e->convp_deint) ? 1 : 0;
    a.Sh = (St + 1) / 2;
    for (int t = 0; t < ntap; t++) {
        a.rowgrp[t] = taps[t][2];
        const int c = taps[t][3];
        a.coloff[t] = a.deint ? (c & 1) * a.Sh + (c >> 1) : c;
    }
    a.nchunk = nchunk; a.St = St; a.act = act; a.relu_lo = relu_lo; a.relu_hi = relu_hi;
    a.out_mode = out_mode; a.row_perm = out_mode == kPOutP || out_mode == kPOutBlend || out_mode == kPOutGate;
    a.valid_m = FP; a.par_rows = 0; a.stats = nullptr;
    pl.CO = CO; pl.npair = npair; pl.terms = PL == 3 ? 6 : (PL == 2 ? 3 : 1);
    // weights [chunk][pair][plane][mtile][row 32][k 16]: k = half*8 + c <-> entry 2*pair + half = (tap, octet) tap-major
    std::vector<uint16_t> wx((size_t)nchunk * npair * PL * MT * 32 * 16, 0);
    std::vector<float> bias(CoPad, 0.0f);
    for (int row = 0; row < CoPad; row++) {
        const int lg = convp_row_logical(out_mode, row);
        if (lg < Co) bias[row] = bias_logical[lg];
    }
    for (int ch = 0; ch < nchunk; ch++)
        for (int st = 0; st < npair; st++)
            for (int m = 0; m < MT; m++)
                for (int r = 0; r < 32; r++) {
                    const int lg = convp_row_logical(out_mode, m * 32 + r);
                    if (lg >= Co) continue;
                    for (int k = 0; k < 16; k++) {
                        const int en = 2 * st + k / 8, tp = en / CO, oc = en % CO;
                        const int ci = (ch * CO + oc) * 8 + k % 8;
                        if (tp >= ntap || ci >= Ci) continue;
                        const float x = wsel(ci, lg, taps[tp][0], taps[tp][1]);
                        const uint16_t h = bf16_rne(x);
                        const float r1 = x - bf16_to_f32(h);
                        const uint16_t md = bf16_rne(r1);
                        const float r2 = r1 - bf16_to_f32(md);
                        const uint16_t parts[3] = {PL == 1 ? f16_rne(x) : h, md, bf16_rne(r2)};
                        for (int pln = 0; pln < PL; pln++)
                            wx[(((((size_t)ch * npair + st) * PL + pln) * MT + m) * 32 + r) * 16 + k] = parts[pln];
                    }
                }
    int rc = dev_alloc(e, pl.wx, (wx.size() + 1) / 2);
    if (rc) return rc;
    HIPCHECK(e, hipMemcpy(pl.wx.p, wx.data(), wx.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    a.wx = reinterpret_cast<const uint4 *>(pl.wx.p);
    if ((rc = dev_upload(e, pl.bias, bias))) return rc;
    a.bias = pl.bias.p;
    return 0;
}

// Picks the tiling for the current batch: time ~ rounds x (workgroups per CU x MFMA cycles + staging + epilogue)
void select_convp_geometry(se_engine *e, ConvPPlan &pl) {
    if (!pl.active) return;
    double best = 0;
    int pick = -1;
    int force = getenv("SE_CONVP_NT") ? atoi(getenv("SE_CONVP_NT")) : 0;
    if (const char *s = getenv(("SE_CONVP_NT_" + pl.name).c_str())) force = atoi(s);
    for (int k = 0; k < pl.ngeo; k++) {
        const ConvPPlan::Geo &g = pl.geo[k];
        const int wgpc = (g.NT <= 6 && g.lds <= 80 * 1024) ? 2 : 1;  // k_conv_p<.., NT <= 6> is built for two workgroups per CU
        const double rounds = std::ceil((double)g.n_wg * e->B / ((double)e->num_cu * wgpc));
        const double mfma = (double)pl.a.nchunk * pl.npair * g.NT * pl.terms * 32.0;
        const double stage = (double)pl.a.nchunk * g.items * 1.0 + pl.a.nchunk * 2500.0;  // ~16 B/cycle/CU + DMA latency and barriers per chunk
        const double epi = g.NT * 600.0 + 3000.0;
        const double cost = rounds * (wgpc * mfma + stage + epi);
        if (pick < 0 || cost < best) { best = cost; pick = k; }
        if (force && g.NT == force) { pick = k; break; }
    }
    if (getenv("SE_CONVP_VERBOSE")) {
        const ConvPPlan::Geo &g = pl.geo[pick];
        fprintf(stderr, "[conv_p] %-10s taps %2d Ci %3d rows %3d FP %3d: NT %2d tiles/wg %2d wgs/stream %2d lds %6zu items %5ld chunks %d pairs %d | candidates:", pl.name.c_str(),
                pl.a.ntap, pl.a.Ci, pl.a.Co, pl.a.FP, g.NT, g.tpw, g.n_wg, g.lds, g.items, pl.a.nchunk, pl.npair);
        for (int k = 0; k < pl.ngeo; k++) fprintf(stderr, " %d", pl.geo[k].NT);
        fprintf(stderr, "\n");
    }
    const ConvPPlan::Geo &g = pl.geo[pick];
    pl.NT = g.NT; pl.grid_x = g.n_wg; pl.lds = g.lds;
    pl.a.tiles_per_wg = g.tpw; pl.a.grouped = g.grouped;
}

#ifdef SE_CP_TRACE
struct CpTraceSites {
    std::vector<std::pair<std::string, unsigned long long *>> sites;
    unsigned long long *get(const char *label, int NT, int grid_x) {
        const std::string key = std::string(label ? label : "?") + " NT=" + std::to_string(NT) + " wg/stream=" + std::to_string(grid_x);
        for (auto &s : sites) if (s.first == key) return s.second;
        unsigned long long *p = nullptr;
        (void)hipMalloc(&p, 64); (void)hipMemset(p, 0, 64);
        sites.emplace_back(key, p);
        return p;
    }
    void dump() {
        for (auto &s : sites) {
            unsigned long long h[8];
            if (hipMemcpy(h, s.second, sizeof h, hipMemcpyDeviceToHost) != hipSuccess || !h[7]) continue;
            const double n = (double)h[7];
            fprintf(stderr, "[cp trace %-34s] %8llu WGs, cycles/WG: prologue %7.0f  barrier1 %7.0f  dma issue %7.0f  barrier2+dma wait %7.0f  mfma loop %7.0f  epilogue %7.0f\n",
                    s.first.c_str(), h[7], h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n, h[5] / n);
            (void)hipMemset(s.second, 0, 64);
        }
    }
};
static CpTraceSites g_cp_trace_sites;
#endif

int launch_conv_p(se_engine *e, const ConvPPlan &pl, const ConvPArgs &a_in, hipStream_t st, const char *label) {
    if (!pl.active) return 0;
    ProfScope ps(e, "k_conv_p", label, pl.flops * e->B, st);
    ConvPArgs a = a_in;
    a.trace = nullptr;
#ifdef SE_CP_TRACE
    a.trace = g_cp_trace_sites.get(label, pl.NT, pl.grid_x);
#endif
    if (conv_p_launch(a.ntap, pl.NT, pl.CO, operand_planes(e->precision), dim3(pl.grid_x, e->Bact), pl.lds, st, a))
        return fail(e, SE_ERR_ARG, "no conv_p kernel instance for %d taps x %d tiles x %d octets", a.ntap, pl.NT, pl.CO);
    HIPCHECK(e, hipGetLastError());
    return 0;
}

uint4 *decin_p(se_engine *e, int slot) { return reinterpret_cast<uint4 *>(e->cp->decinP[slot].p); }

void select_all_p(se_engine *e) {
    for (int i = 0; i < e->L; i++)
        for (ConvPPlan *p : {&e->cp->pv[i].enc, &e->cp->pv[i].dec_even, &e->cp->pv[i].dec_odd, &e->cp->pv[i].skip, &e->cp->pv[i].skipm, &e->cp->pv[i].gate[0], &e->cp->pv[i].gate[1]})
            select_convp_geometry(e, *p);
    if (e->cp->pre_p)
        for (int i = 0; i < e->npre; i++) select_convp_geometry(e, e->cp->pre[i]);
}

void free_state_p(se_engine *e) {
    if (!e->cp) return;
    se_convp_state &S = *e->cp;
    for (int i = 0; i < SE_MAX_LEVELS; i++) {
        for (ConvPPlan *p : {&S.pv[i].enc, &S.pv[i].dec_even, &S.pv[i].dec_odd, &S.pv[i].skip, &S.pv[i].skipm, &S.pv[i].gate[0], &S.pv[i].gate[1]}) { dev_free(p->wx); dev_free(p->bias); }
        dev_free(S.pv[i].sk.wx); dev_free(S.pv[i].sk.cst);
        dev_free(S.xinP[i]); dev_free(S.encR[i]); dev_free(S.encA[i]); dev_free(S.decR[i]); dev_free(S.decP[i]);
    }
    for (int r = 0; r < kRing; r++) dev_free(S.decinP[r]);
    for (int i = 0; i < 3; i++) { dev_free(S.pre[i].wx); dev_free(S.pre[i].bias); dev_free(S.pinP[i]); dev_free(S.pre_gw[i]); }
    dev_free(S.preR);
    delete e->cp;
    e->cp = nullptr;
}

bool convp_supported(const se_engine *e) {
    // (CRN_ELU / student: the three 5-channel pre-conv blocks stay on their fp32 vector-ALU kernel and hand over to the plane
    //  path with one conversion; the gated 1x1 pair of every encoder block is a k_conv_p epilogue, kPOutGate)
    if (e->variant && e->Ch[0] > 8) return false;
    for (int i = 0; i <= e->L; i++)
        if (e->Ch[i] > 128) return false;
    return true;
}

int prepare_weights_p(se_engine *e) {
    se_convp_state &S = *e->cp;
    const int L = e->L, T = e->T;
    // The pre-conv blocks run on k_conv_p only with fp16 operands: a workgroup stages a 6-7 row x (F + 4 fd) column patch for
    // 1-2 output rows, and with three bf16 planes that staging costs more than the scalar-weight vector-ALU kernel takes
    // (measured, student B = 1024: 760 vs 450 us per block in f32, equal in bf16x3, 0.6x in fp16).  SE_PRE_P=0/1 overrides.
    S.pre_p = e->npre > 0 && (getenv("SE_PRE_P") ? atoi(getenv("SE_PRE_P")) != 0 : operand_planes(e->precision) == 1);
    for (int i = 0; i < e->npre && S.pre_p; i++) {  // Conv2d(5x5, dilation (fd,1), padding (2fd,4)) + ELU + gated pair (CRN_ELU.py:335-340)
        const int C0 = e->Ch[0], F0 = e->F[0], fd = 1 << i;
        const std::string p = "preconvlist." + std::to_string(i) + ".";
        auto *w = param(e, p + "conv.weight", (size_t)C0 * C0 * 25);
        auto *b = param(e, p + "conv.bias", C0);
        auto *tw = param(e, p + "conv_trans.weight", (size_t)C0 * C0);
        auto *tb = param(e, p + "conv_trans.bias", C0);
        auto *gw = param(e, p + "conv_gated.weight", (size_t)C0 * C0);
        auto *gb = param(e, p + "conv_gated.bias", C0);
        if (!w || !b || !tw || !tb || !gw || !gb) return SE_ERR_PARAM_MISSING;
        std::vector<std::array<int, 4>> taps;
        for (int kf = 0; kf < 5; kf++)
            for (int kt = 0; kt < 5; kt++) taps.push_back({kf, kt, kt, kf * fd});
        const float *wp = w->data();
        int rc = plan_conv_p(e, S.pre[i], C0, C0, F0, F0, 1, 2 * fd, -4, 5, 1, F0 + 4 * fd, taps,
                             [=](int ci, int co, int kf, int kt) { return wp[(((size_t)co * C0 + ci) * 5 + kf) * 5 + kt]; }, *b, 0, C0, 2, kPOutPre);
        if (rc) { S.pre_p = false; e->err.clear(); break; }  // (no tiling fits: the blocks stay on the fp32 vector-ALU kernel)
        S.pre[i].name = "pre" + std::to_string(i);
        ConvPArgs &a = S.pre[i].a;
        a.Cy = C0; a.oT = F0; a.oo = 0; a.y_npos = T * F0; a.y_stream = (long)T * F0 * 8;
        std::vector<float> g;
        g.insert(g.end(), tw->begin(), tw->end());
        g.insert(g.end(), gw->begin(), gw->end());
        g.insert(g.end(), tb->begin(), tb->end());
        g.insert(g.end(), gb->begin(), gb->end());
        if ((rc = dev_upload(e, S.pre_gw[i], g))) return rc;
        a.gatew = S.pre_gw[i].p;
        S.pre[i].flops = 2.0 * C0 * C0 * 25 * F0 * T + 2.0 * 2 * C0 * C0 * F0 * T;
    }
    for (int i = 0; i < L; i++) {
        const int Ci = e->Ch[i], Co = e->Ch[i + 1], Fi = e->F[i], Fo = e->F[i + 1], d = 1 << i;
        const std::string p = "convlist." + std::to_string(i) + ".";
        auto *w = param(e, p + "conv.weight", (size_t)Co * Ci * 15);
        auto *b = param(e, p + "conv.bias", Co);
        if (!w || !b) return SE_ERR_PARAM_MISSING;
        std::vector<std::array<int, 4>> taps;
        for (int kf = 0; kf < 5; kf++)
            for (int kt = 0; kt < 3; kt++) taps.push_back({kf, kt, kt, kf});
        const float *wp = w->data();
        int rc = plan_conv_p(e, S.pv[i].enc, Ci, Co, Fo, Fi, 2, 2, -2 * d, 3, d, Fi + 4, taps,
                             [=](int ci, int co, int kf, int kt) { return wp[(((size_t)co * Ci + ci) * 5 + kf) * 3 + kt]; }, *b, 0, Co, e->act,
                             e->variant ? kPOutP : kPOutR);
        if (rc) return rc;
        S.pv[i].enc.name = "enc" + std::to_string(i);
        ConvPArgs &a = S.pv[i].enc.a;
        a.oT = Fo; a.oo = 0; a.y_npos = T * Fo; a.y_stream = (long)((Co + 7) / 8) * T * Fo * 8;
        a.stats_lo = 0; a.stats_hi = Co;
        S.pv[i].gate[0].active = S.pv[i].gate[1].active = false;
        if (e->variant) {  // act(conv) -> P layout -> gated 1x1 pair (CRN_ELU.py:223-224, 240) -> R layout + statistics
            a.Cy = Co; a.Fy = Fo; a.yp_stream = (long)((Co + 7) / 8) * operand_planes(e->precision) * T * Fo;
            auto *tw = param(e, p + "conv_trans.weight", (size_t)Co * Co);
            auto *tb = param(e, p + "conv_trans.bias", Co);
            auto *gw = param(e, p + "conv_gated.weight", (size_t)Co * Co);
            auto *gb = param(e, p + "conv_gated.bias", Co);
            if (!tw || !tb || !gw || !gb) return SE_ERR_PARAM_MISSING;
            const float *twp = tw->data(), *gwp = gw->data();
            const int nparts = Co > 64 ? 2 : 1, cpart = ((Co + nparts - 1) / nparts + 7) / 8 * 8;
            std::vector<std::array<int, 4>> t1 = {{0, 0, 0, 0}};
            for (int part = 0; part < nparts; part++) {
                const int c0 = part * cpart, cn = std::min(cpart, Co - c0);
                std::vector<float> bias2(2 * cn);
                for (int c = 0; c < cn; c++) { bias2[2 * c] = (*tb)[c0 + c]; bias2[2 * c + 1] = (*gb)[c0 + c]; }
                ConvPPlan &g = S.pv[i].gate[part];
                if ((rc = plan_conv_p(e, g, Co, 2 * cn, Fo, Fo, 1, 0, 0, 1, 0, Fo, t1,
                                      [=](int ci, int row, int, int) { const int c = c0 + row / 2; return (row & 1) ? gwp[(size_t)c * Co + ci] : twp[(size_t)c * Co + ci]; },
                                      bias2, 0, 0, 0, kPOutGate))) return rc;
                g.name = "gate" + std::to_string(i);
                g.a.Cy = cn; g.a.gate_c0 = c0; g.a.oT = Fo; g.a.oo = 0; g.a.y_npos = T * Fo; g.a.y_stream = a.y_stream;
                g.flops = 2.0 * 2 * cn * Co * Fo * T;
            }
        }
        S.pv[i].enc.flops = 2.0 * Co * Ci * 15 * Fo * T;
    }
    for (int j = 0; j < L; j++) {
        const int lvl = L - 1 - j;
        const int Ci = e->Ch[lvl + 1], Co = lvl == 0 ? 2 : e->Ch[lvl], Fi = e->F[lvl + 1], d = 1 << j;
        const std::string p = "deconvlist." + std::to_string(j) + ".";
        auto *w = param(e, p + "conv.weight", (size_t)Co * Ci * 15);
        auto *b = param(e, p + "conv.bias", Co);
        if (!w || !b) return SE_ERR_PARAM_MISSING;
        const float *wp = w->data();
        auto wsel = [=](int ci, int co, int kf, int kt) { return wp[(((size_t)ci * Co + co) * 5 + kf) * 3 + kt]; };
        PLevel &pv = S.pv[j];
        int rc;
        pv.dec_even.name = "dec" + std::to_string(j) + "_even"; pv.dec_odd.name = "dec" + std::to_string(j) + "_odd";
        pv.skip.name = "skip" + std::to_string(j); pv.skipm.name = "skipstat" + std::to_string(j);
        pv.dec_merged = lvl == 0;  // the last block (2 mask channels): both parities as 4 GEMM rows of ONE launch
        if (pv.dec_merged) {
            std::vector<std::array<int, 4>> tu;
            for (int kf = 0; kf < 5; kf++)
                for (int kt = 0; kt < 3; kt++) tu.push_back({kf, kt, 2 - kt, (kf & 1) ? 1 + (3 - kf) / 2 : 2 - kf / 2});
            std::vector<float> bias2(2 * Co);
            for (int c = 0; c < Co; c++) bias2[2 * c] = bias2[2 * c + 1] = (*b)[c];
            rc = plan_conv_p(e, pv.dec_even, Ci, 2 * Co, Fi, Fi, 1, 1, 0, 3, d, Fi + 2, tu,
                             [=](int ci, int row, int kf, int kt) { return ((row & 1) == (kf & 1)) ? wsel(ci, row >> 1, kf, kt) : 0.0f; },
                             bias2, 0, 2 * Co, e->act, kPOutR);
            if (rc) return rc;
            ConvPArgs &a = pv.dec_even.a;
            a.par_rows = 1; a.oT = Fi; a.oo = 0; a.y_npos = T * Fi; a.y_stream = (long)T * Fi * 8;
            a.stats_lo = 0; a.stats_hi = 2 * Co;
            pv.dec_even.flops = 2.0 * Ci * Co * 15 * Fi * T;
            pv.dec_odd.active = false;
        } else {
            std::vector<std::array<int, 4>> te, to;
            for (int kf = 0; kf < 5; kf += 2)
                for (int kt = 0; kt < 3; kt++) te.push_back({kf, kt, 2 - kt, 2 - kf / 2});
            for (int kf = 1; kf < 5; kf += 2)
                for (int kt = 0; kt < 3; kt++) to.push_back({kf, kt, 2 - kt, 1 + (3 - kf) / 2});
            if ((rc = plan_conv_p(e, pv.dec_even, Ci, Co, Fi, Fi, 1, 1, 0, 3, d, Fi + 2, te, wsel, *b, 0, Co, e->act, kPOutR))) return rc;
            if ((rc = plan_conv_p(e, pv.dec_odd, Ci, Co, Fi - 1, Fi, 1, 1, 0, 3, d, Fi + 2, to, wsel, *b, 0, Co, e->act, kPOutR))) return rc;
            for (ConvPPlan *q : {&pv.dec_even, &pv.dec_odd}) {
                ConvPArgs &a = q->a;
                a.oT = 2 * Fi; a.oo = q == &pv.dec_odd ? Fi : 0; a.y_npos = T * 2 * Fi; a.y_stream = (long)((Co + 7) / 8) * T * 2 * Fi * 8;
                a.stats_lo = 0; a.stats_hi = Co;
            }
            pv.dec_even.flops = 2.0 * Ci * Co * 9 * Fi * T;
            pv.dec_odd.flops = 2.0 * Ci * Co * 6 * Fi * T;
        }
        if (lvl > 0) {  // skip path (CRN.py:387-396): statistics pass of residualmask + the gated 1x1 pair
            auto *mw = param(e, p + "residualmask.weight", (size_t)Co * Co);
            auto *mb = param(e, p + "residualmask.bias", Co);
            auto *rw = param(e, p + "residual.weight", (size_t)Co * Co);
            auto *rb = param(e, p + "residual.bias", Co);
            if (!mw || !mb || !rw || !rb) return SE_ERR_PARAM_MISSING;
            const float *mwp = mw->data(), *rwp = rw->data();
            std::vector<std::array<int, 4>> t1 = {{0, 0, 0, 0}};
            const int Fr = e->F[lvl];
            std::vector<float> biasp(2 * Co);
            for (int c = 0; c < Co; c++) { biasp[2 * c] = (*mb)[c]; biasp[2 * c + 1] = (*rb)[c]; }
            if ((rc = plan_conv_p(e, pv.skip, Co, 2 * Co, Fr, Fr, 1, 0, 0, 1, 0, Fr, t1,
                                  [=](int ci, int row, int, int) { const int c = row >> 1; return (row & 1) ? rwp[(size_t)c * Co + ci] : mwp[(size_t)c * Co + ci]; },
                                  biasp, 0, 0, e->act, kPOutBlend))) return rc;
            if ((rc = plan_conv_p(e, pv.skipm, Co, Co, Fr, Fr, 1, 0, 0, 1, 0, Fr, t1,
                                  [=](int ci, int co, int, int) { return mwp[(size_t)co * Co + ci]; }, *mb, 0, 0, e->act, kPOutR))) return rc;
            pv.skipm.a.stats_lo = 0; pv.skipm.a.stats_hi = Co; pv.skipm.a.oT = Fr; pv.skipm.a.oo = 0; pv.skipm.a.y_npos = T * Fr;
            pv.skip.a.Cy = Co; pv.skip.a.Fy = Fr; pv.skip.a.oo = 0;
            pv.skip.flops = 2.0 * 2 * Co * Co * Fr * T;
            pv.skipm.flops = 0;  // recomputation, not algorithmic work
            {   // streaming form (k_skip_p): M tiles [0, MTh) = residualmask, [MTh, 2 MTh) = residual; rows permuted like kPOutP
                auto *mnw = param(e, p + "residualnorm.weight", Co);
                auto *mnb = param(e, p + "residualnorm.bias", Co);
                auto *nw = param(e, p + "norm.weight", Co);
                auto *nb = param(e, p + "norm.bias", Co);
                if (!mnw || !mnb || !nw || !nb) return SE_ERR_PARAM_MISSING;
                SkipPPlan &sk = pv.sk;
                const int PL = operand_planes(e->precision), C8 = (Co + 7) / 8, MTh = (Co + 31) / 32, KP = (C8 + 1) / 2, Cp = MTh * 32;
                sk.active = e->skip_stream != 0 && MTh <= 2 && (KP == 1 || KP == 2 || KP == 4);
                if (sk.active) {
                    std::vector<uint16_t> wx((size_t)2 * MTh * KP * PL * 32 * 16, 0);
                    for (int m2 = 0; m2 < 2 * MTh; m2++)
                        for (int kp = 0; kp < KP; kp++)
                            for (int r = 0; r < 32; r++) {
                                const int mt = m2 % MTh, kind = m2 / MTh;
                                const int ch = convp_row_logical(kPOutP, mt * 32 + r);
                                if (ch >= Co) continue;
                                for (int k = 0; k < 16; k++) {
                                    const int ci = (2 * kp + k / 8) * 8 + k % 8;
                                    if (ci >= Co) continue;
                                    const float x = kind ? rwp[(size_t)ch * Co + ci] : mwp[(size_t)ch * Co + ci];
                                    const uint16_t h = bf16_rne(x);
                                    const float r1 = x - bf16_to_f32(h);
                                    const uint16_t md = bf16_rne(r1);
                                    const uint16_t parts[3] = {PL == 1 ? f16_rne(x) : h, md, bf16_rne(r1 - bf16_to_f32(md))};
                                    for (int pln = 0; pln < PL; pln++)
                                        wx[((((size_t)m2 * KP + kp) * PL + pln) * 32 + r) * 16 + k] = parts[pln];
                                }
                            }
                    if ((rc = dev_alloc(e, sk.wx, (wx.size() + 1) / 2))) return rc;
                    HIPCHECK(e, hipMemcpy(sk.wx.p, wx.data(), wx.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
                    std::vector<float> cst((size_t)6 * Cp, 0.0f);
                    for (int c = 0; c < Co; c++) {
                        cst[c] = (*mb)[c]; cst[Cp + c] = (*rb)[c]; cst[2 * Cp + c] = (*nw)[c]; cst[3 * Cp + c] = (*nb)[c];
                        cst[4 * Cp + c] = (*mnw)[c]; cst[5 * Cp + c] = (*mnb)[c];
                    }
                    if ((rc = dev_upload(e, sk.cst, cst))) return rc;
                    SkipPArgs &a = sk.a;
                    a.C = Co; a.C8 = C8; a.T = T; a.Fr = Fr; a.TF = T * Fr; a.MTh = MTh; a.KP = KP; a.act = e->act;
                    a.wx = reinterpret_cast<const uint4 *>(sk.wx.p); a.cst = sk.cst.p;
                    a.x_stream = (long)C8 * PL * T * Fr; a.out_stream = a.x_stream;
                    a.Fh = Fi; a.Fo = 2 * Fi - 1; a.y_stream = pv.dec_even.a.y_stream; a.eps_mode = e->eps_mode;
                    sk.KP = KP;
                    sk.lds = (size_t)2 * MTh * KP * PL * 64 * 16 + (size_t)6 * Cp * 4;
                    sk.flops = pv.skip.flops;
                }
            }
        } else {
            pv.skip.active = pv.skipm.active = false;
        }
    }
    S.ready = true;
    return 0;
}

int alloc_state_p(se_engine *e, hipStream_t st) {
    se_convp_state &S = *e->cp;
    const int L = e->L, T = e->T, B = e->B, PL = operand_planes(e->precision);
    int rc;
    for (int i = 0; i < L; i++) {
        const long per = (long)((e->Ch[i] + 7) / 8) * PL * T * e->F[i];  // uint4 per stream
        S.slot_elems[i] = per * B;
        const size_t bytes = (size_t)S.slot_elems[i] * kRing * 16;
        if (bytes >= (size_t)0xFFFFFF00u) return fail(e, SE_ERR_ARG, "activation ring of level %d needs %zu bytes: beyond one buffer descriptor (batch too large)", i, bytes);
        if ((rc = dev_alloc(e, S.xinP[i], bytes / 4))) return rc;
        HIPCHECK(e, hipMemsetAsync(S.xinP[i].p, 0, (size_t)S.slot_elems[i] * 16, st));  // slot 0 = the all-zero history of the first segment
        const int Co = e->Ch[i + 1], Fo = e->F[i + 1];
        if ((rc = dev_alloc(e, S.encR[i], (size_t)B * ((Co + 7) / 8) * T * Fo * 8))) return rc;
        // channel slots beyond Co of the last octet are never written; consumers scale them by zero, so they must be finite
        HIPCHECK(e, hipMemsetAsync(S.encR[i].p, 0, (size_t)B * ((Co + 7) / 8) * T * Fo * 8 * sizeof(float), st));
        if (e->variant && (rc = dev_alloc(e, S.encA[i], (size_t)B * ((Co + 7) / 8) * PL * T * Fo * 4))) return rc;
    }
    for (int r = 0; r < kRing; r++)
        if ((rc = dev_alloc(e, S.decinP[r], (size_t)B * ((e->Ch[L] + 7) / 8) * PL * T * e->F[L] * 4))) return rc;
    if (S.pre_p) {
        S.pslot_elems = (long)PL * T * e->F[0] * B;
        for (int i = 0; i < e->npre; i++) {
            if ((rc = dev_alloc(e, S.pinP[i], (size_t)S.pslot_elems * kRing * 4))) return rc;
            HIPCHECK(e, hipMemsetAsync(S.pinP[i].p, 0, (size_t)S.pslot_elems * kRing * 16, st));  // every slot: zero history wherever the ring starts
            if ((rc = dev_alloc(e, e->pre_stats[i], (size_t)B * 2 * (S.pre[i].grid_x + 1)))) return rc;
        }
        if ((rc = dev_alloc(e, S.preR, (size_t)B * T * e->F[0] * 8))) return rc;
    }
    for (int j = 0; j < L; j++) {
        const int lvl = L - 1 - j, Co = lvl == 0 ? 2 : e->Ch[lvl], Fi = e->F[lvl + 1];
        const size_t nR = lvl == 0 ? (size_t)B * T * Fi * 8 : (size_t)B * ((Co + 7) / 8) * T * 2 * Fi * 8;
        if ((rc = dev_alloc(e, S.decR[j], nR))) return rc;
        HIPCHECK(e, hipMemsetAsync(S.decR[j].p, 0, nR * sizeof(float), st));
        if (lvl > 0 && (rc = dev_alloc(e, S.decP[j], (size_t)B * ((Co + 7) / 8) * PL * T * e->F[lvl] * 4))) return rc;
    }
    // statistics slabs are shared with the first-generation path; make sure they hold the new grids
    for (int i = 0; i < L; i++) {
        PLevel &pv = S.pv[i];
        if ((rc = dev_alloc(e, e->enc_stats[i], (size_t)B * 2 * (pv.enc.grid_x + (pv.gate[0].active ? pv.gate[0].grid_x : 0) + (pv.gate[1].active ? pv.gate[1].grid_x : 0) + 1)))) return rc;
        if ((rc = dev_alloc(e, e->dec_stats[i], (size_t)B * 2 * (pv.dec_even.grid_x + (pv.dec_odd.active ? pv.dec_odd.grid_x : 0) + 1)))) return rc;
        if ((rc = dev_alloc(e, e->skip_stats[i], (size_t)B * 2 * ((pv.skipm.active ? pv.skipm.grid_x : 0) + 1)))) return rc;
    }
    return 0;
}

// Stage 1 (plane path): features + encoder -> xinP[*][cur], gru_in[cur] (fp32, the bottleneck GEMM's A operand)
int stage_encoder_p(se_engine *e, int cur, int prev, const cf2 *spec, long sB, long sM, long sT, long sF, hipStream_t st) {
    se_convp_state &S = *e->cp;
    const int L = e->L, T = e->T, B = e->B, PL = operand_planes(e->precision);
    const int Ba = e->Bact;  // launch batch: the prefix of streams that take part in this segment (se_realtime_process_ragged), B otherwise
    int rc;
    if (e->variant && S.pre_p) {  // features and the three pre-conv blocks x = m(x) + x on the plane path
        const int TF = T * e->F[0], C0 = e->Ch[0];
        {
            ProfScope ps(e, "k_featurize_p", "featurize", 0, st);
            FeatPArgs f{spec, sB, sM, sT, sF, reinterpret_cast<uint4 *>(S.pinP[0].p) + (long)cur * S.pslot_elems, (long)PL * TF, e->M, T, e->F[0], e->atan2_phase};
            launch_k_featurize_p(PL, dim3((TF + 255) / 256, Ba), st, f);
            HIPCHECK(e, hipGetLastError());
        }
        for (int i = 0; i < e->npre; i++) {
            ConvPPlan &pl = S.pre[i];
            ConvPArgs a = pl.a;
            a.xbase = reinterpret_cast<const uint4 *>(S.pinP[i].p);
            a.xbytes = (unsigned)((size_t)S.pslot_elems * kRing * 16);
            a.cur_off = (long)cur * S.pslot_elems;
            a.prev_off = (long)prev * S.pslot_elems;
            a.y = S.preR.p;
            a.stats = e->pre_stats[i].p; a.stats_nslot = pl.grid_x; a.stats_slot0 = 0;
            if ((rc = launch_conv_p(e, pl, a, st, ("pre" + std::to_string(i)).c_str()))) return rc;
            ProfScope ps(e, "k_gln_p", "gln", 0, st);
            GlnPArgs g{};
            g.x = S.preR.p; g.x_stream = a.y_stream; g.C = C0; g.C8 = 1; g.T = T; g.F = e->F[0]; g.in_oT = e->F[0];
            g.w = e->lv[i].pre_nw.p; g.b = e->lv[i].pre_nb.p;
            g.st = SlabStats{e->pre_stats[i].p, pl.grid_x, (long)C0 * TF, e->eps_mode};
            g.res = reinterpret_cast<const uint4 *>(S.pinP[i].p) + (long)cur * S.pslot_elems; g.res_stream = (long)PL * TF;
            g.mode = 0;
            g.y = i + 1 < e->npre ? reinterpret_cast<uint4 *>(S.pinP[i + 1].p) + (long)cur * S.pslot_elems
                                  : reinterpret_cast<uint4 *>(S.xinP[0].p) + (long)cur * S.slot_elems[0];
            g.y_stream = (long)PL * TF;
            launch_k_gln_p(PL, dim3((TF + 1023) / 1024, 1, Ba), st, g);
            HIPCHECK(e, hipGetLastError());
        }
    } else if (e->variant) {  // the same blocks on their fp32 vector-ALU kernel (first generation), then one conversion into the P layout
        if ((rc = stage_features_pre(e, cur, spec, sB, sM, sT, sF, st))) return rc;
        ProfScope ps(e, "k_f32_to_p", "feat_to_p", 0, st);
        const int TF = T * e->F[0];
        F32ToPArgs f{e->xin[0][cur].p, reinterpret_cast<uint4 *>(S.xinP[0].p) + (long)cur * S.slot_elems[0], (long)PL * TF, e->Ch[0], TF};
        launch_k_f32_to_p(PL, dim3((TF + 255) / 256, Ba), st, f);
        HIPCHECK(e, hipGetLastError());
    } else {
        ProfScope ps(e, "k_featurize_p", "featurize", 0, st);
        const int TF = T * e->F[0];
        FeatPArgs f{spec, sB, sM, sT, sF, reinterpret_cast<uint4 *>(S.xinP[0].p) + (long)cur * S.slot_elems[0], (long)PL * TF, e->M, T, e->F[0], e->atan2_phase};
        launch_k_featurize_p(PL, dim3((TF + 255) / 256, Ba), st, f);
        HIPCHECK(e, hipGetLastError());
    }
    for (int i = 0; i < L; i++) {
        const int Co = e->Ch[i + 1], Fo = e->F[i + 1];
        ConvPPlan &pl = S.pv[i].enc;
        ConvPArgs a = pl.a;
        a.xbase = reinterpret_cast<const uint4 *>(S.xinP[i].p);
        a.xbytes = (unsigned)((size_t)S.slot_elems[i] * kRing * 16);
        a.cur_off = (long)cur * S.slot_elems[i];
        a.prev_off = (long)prev * S.slot_elems[i];
        int nslot = pl.grid_x;
        if (!e->variant) {
            a.y = S.encR[i].p;
            a.stats = e->enc_stats[i].p; a.stats_nslot = pl.grid_x; a.stats_slot0 = 0;
            if ((rc = launch_conv_p(e, pl, a, st, ("enc" + std::to_string(i)).c_str()))) return rc;
        } else {
            a.yp = reinterpret_cast<uint4 *>(S.encA[i].p);
            if ((rc = launch_conv_p(e, pl, a, st, ("enc" + std::to_string(i)).c_str()))) return rc;
            const int n0 = S.pv[i].gate[0].grid_x, n1 = S.pv[i].gate[1].active ? S.pv[i].gate[1].grid_x : 0;
            nslot = n0 + n1;
            for (int part = 0; part < 2; part++) {
                ConvPPlan &gp = S.pv[i].gate[part];
                if (!gp.active) continue;
                ConvPArgs g = gp.a;
                g.xbase = reinterpret_cast<const uint4 *>(S.encA[i].p);
                g.xbytes = (unsigned)((size_t)B * ((Co + 7) / 8) * PL * T * Fo * 16);
                g.cur_off = 0; g.prev_off = -1;
                g.y = S.encR[i].p;
                g.stats = e->enc_stats[i].p; g.stats_nslot = nslot; g.stats_slot0 = part ? n0 : 0;
                if ((rc = launch_conv_p(e, gp, g, st, ("gate" + std::to_string(i)).c_str()))) return rc;
            }
        }
        ProfScope ps(e, "k_gln_p", "gln", 0, st);
        GlnPArgs g{};
        g.x = S.encR[i].p; g.x_stream = a.y_stream; g.C = Co; g.C8 = (Co + 7) / 8; g.T = T; g.F = Fo; g.in_oT = Fo;
        g.w = e->lv[i].enc_nw.p; g.b = e->lv[i].enc_nb.p;
        g.st = SlabStats{e->enc_stats[i].p, nslot, (long)Co * T * Fo, e->eps_mode};
        if (i + 1 < L) {
            g.mode = 0;
            g.y = reinterpret_cast<uint4 *>(S.xinP[i + 1].p) + (long)cur * S.slot_elems[i + 1];
            g.y_stream = (long)g.C8 * PL * T * Fo;
            launch_k_gln_p(PL, dim3((T * Fo + 1023) / 1024, g.C8, Ba), st, g);
        } else if (e->gemm_p) {  // last level = A operand of the GRU input projection, as split-bf16 planes (k_gemm_p)
            g.mode = 1;
            g.y = reinterpret_cast<uint4 *>(e->gruinP[cur].p);
            g.y_plane = (long)B * T * g.C8 * Fo;
            launch_k_gln_p(PL, dim3((T * Fo + 1023) / 1024, g.C8, Ba), st, g);
        } else {  // last level feeds the fp32 GEMM of the bottleneck: [T][C*F] rows (gln_ew mode 1 of the first generation)
            GlnEwArgs ge{nullptr, e->gru_in[cur].p, g.w, g.b, g.st, 3, Co, T, Fo, nullptr};
            ge.x = S.encR[i].p;
            const long n = (long)Co * T * Fo;
            hipLaunchKernelGGL(k_gln_r2t, dim3((unsigned)((n / 8 + 255) / 256), B), dim3(256), 0, st, ge, a.y_stream);
        }
        HIPCHECK(e, hipGetLastError());
    }
    return 0;
}

// Stage 3 (plane path): decoder + mask.  The bottleneck's k_gln2_p has already written decinP[cur].
int stage_decoder_p(se_engine *e, int cur, const cf2 *spec, long sB, long sT, long sF, cf2 *out, long oB, long oT, long oF, hipStream_t st) {
    se_convp_state &S = *e->cp;
    const int L = e->L, T = e->T, B = e->B, PL = operand_planes(e->precision);
    const int Ba = e->Bact;  // launch batch: the prefix of streams that take part in this segment (se_realtime_process_ragged), B otherwise
    int rc;
    const uint4 *xin = reinterpret_cast<const uint4 *>(S.decinP[cur].p);
    size_t xin_bytes = (size_t)B * ((e->Ch[L] + 7) / 8) * PL * T * e->F[L] * 16;
    for (int j = 0; j < L; j++) {
        const int lvl = L - 1 - j;
        const int Co = lvl == 0 ? 2 : e->Ch[lvl], Fi = e->F[lvl + 1], Fo = 2 * Fi - 1;
        PLevel &pv = S.pv[j];
        const int ne = pv.dec_even.grid_x, no = pv.dec_odd.active ? pv.dec_odd.grid_x : 0;
        for (ConvPPlan *q : {&pv.dec_even, &pv.dec_odd}) {
            if (!q->active) continue;
            ConvPArgs a = q->a;
            a.xbase = xin; a.xbytes = (unsigned)xin_bytes; a.cur_off = 0; a.prev_off = -1;
            a.y = S.decR[j].p;
            a.stats = e->dec_stats[j].p; a.stats_nslot = ne + no; a.stats_slot0 = q == &pv.dec_odd ? ne : 0;
            if ((rc = launch_conv_p(e, *q, a, st, ("dec" + std::to_string(j) + (q == &pv.dec_odd ? "_odd" : "_even")).c_str()))) return rc;
        }
        const SlabStats sy{e->dec_stats[j].p, ne + no, (long)Co * T * Fo, e->eps_mode};
        if (lvl > 0) {
            const int Fr = e->F[lvl];
            const long nu = (long)Co * T * Fr;
            const uint4 *res = reinterpret_cast<const uint4 *>(S.xinP[lvl].p);
            const unsigned res_bytes = (unsigned)((size_t)S.slot_elems[lvl] * kRing * 16);
            // one workgroup per stream: below ~100 streams most CUs would idle, the two k_conv_p launches (statistics, gate) spread
            // over positions instead (B = 1: 24 vs 45 us per level).  SE_SKIP_STREAM=2 keeps the streaming kernel at any batch.
            if (pv.sk.active && (B >= e->skip_min_batch || e->skip_stream == 2)) {
                SkipPArgs a = pv.sk.a;
                a.x = res + (long)cur * S.slot_elems[lvl];
                a.ydec = S.decR[j].p;
                a.sy = sy;
                a.out = reinterpret_cast<uint4 *>(S.decP[j].p);
                ProfScope ps(e, "k_skip_p", ("skip" + std::to_string(j)).c_str(), pv.sk.flops * B, st);
                if (launch_k_skip_p(PL, pv.sk.KP, Ba, pv.sk.lds, st, a)) return fail(e, SE_ERR_ARG, "no k_skip_p instance for %d planes x %d K steps", PL, pv.sk.KP);
                HIPCHECK(e, hipGetLastError());
            } else {
            {
                ConvPArgs a = pv.skipm.a;
                a.xbase = res; a.xbytes = res_bytes; a.cur_off = (long)cur * S.slot_elems[lvl]; a.prev_off = -1;
                a.y = nullptr; a.out_mode = kPOutStats;
                a.stats = e->skip_stats[j].p; a.stats_nslot = pv.skipm.grid_x; a.stats_slot0 = 0;
                if ((rc = launch_conv_p(e, pv.skipm, a, st, ("skipstat" + std::to_string(j)).c_str()))) return rc;
            }
            {
                ConvPArgs a = pv.skip.a;
                a.xbase = res; a.xbytes = res_bytes; a.cur_off = (long)cur * S.slot_elems[lvl]; a.prev_off = -1;
                a.yp = reinterpret_cast<uint4 *>(S.decP[j].p);
                a.yp_stream = (long)((Co + 7) / 8) * PL * T * Fr;
                a.bl_ydec = S.decR[j].p; a.bl_stream = pv.dec_even.a.y_stream; a.bl_oT = 2 * Fi; a.bl_Fh = Fi; a.bl_Fo = Fo;
                a.bl_nw = e->lv[j].dec_nw.p; a.bl_nb = e->lv[j].dec_nb.p; a.bl_mnw = e->lv[j].dec_mnw.p; a.bl_mnb = e->lv[j].dec_mnb.p;
                a.bl_sy = sy; a.bl_su = SlabStats{e->skip_stats[j].p, pv.skipm.grid_x, nu, e->eps_mode};
                if ((rc = launch_conv_p(e, pv.skip, a, st, ("skip" + std::to_string(j)).c_str()))) return rc;
            }
            }
            xin = reinterpret_cast<const uint4 *>(S.decP[j].p);
            xin_bytes = (size_t)B * ((Co + 7) / 8) * PL * T * Fr * 16;
        } else {
            if (Fo != e->F[0]) return fail(e, SE_ERR_ARG, "decoder output has %d bins, spectrum has %d", Fo, e->F[0]);
            MaskPArgs m{S.decR[j].p, (long)T * Fi * 8, Fi, e->lv[j].dec_nw.p, e->lv[j].dec_nb.p, sy, spec, sB, sT, sF, out, oB, oT, oF, T, e->F[0]};
            ProfScope ps(e, "k_final_mask_p", "final_mask", 0, st);
            launch_k_final_mask_p(dim3((T * e->F[0] + 1023) / 1024, Ba), st, m);
            HIPCHECK(e, hipGetLastError());
        }
    }
    return 0;
}

// P layout (device) -> reference layout [B][C][F][T] on the host (debug taps / state export); fp32 = sum of the planes
int p_to_host(se_engine *e, const float *dev, int C, int F, float *host_out, hipStream_t st, bool to_bcft) {
    const int T = e->T, B = e->B, PL = operand_planes(e->precision), C8 = (C + 7) / 8;
    const size_t n16 = (size_t)B * C8 * PL * T * F;  // uint4 count
    std::vector<uint16_t> h(n16 * 8);
    HIPCHECK(e, hipStreamSynchronize(st));
    HIPCHECK(e, hipMemcpy(h.data(), dev, n16 * 16, hipMemcpyDeviceToHost));
    for (int b = 0; b < B; b++)
        for (int c = 0; c < C; c++)
            for (int t = 0; t < T; t++)
                for (int f = 0; f < F; f++) {
                    float v = 0;
                    for (int pl = 0; pl < PL; pl++) {
                        const uint16_t u = h[(((((size_t)b * C8 + c / 8) * PL + pl) * T + t) * F + f) * 8 + c % 8];
                        if (PL == 1) { _Float16 hv; memcpy(&hv, &u, 2); v += (float)hv; }
                        else v += bf16_to_f32(u);
                    }
                    if (to_bcft) host_out[(((size_t)b * C + c) * F + f) * T + t] = v;
                    else host_out[(((size_t)b * C + c) * T + t) * F + f] = v;
                }
    return SE_OK;
}

// host [B][C][T][F] fp32 -> P layout on the device (state import)
int host_to_p(se_engine *e, const std::vector<float> &src, int C, int F, float *dev) {
    const int T = e->T, B = e->B, PL = operand_planes(e->precision), C8 = (C + 7) / 8;
    std::vector<uint16_t> h((size_t)B * C8 * PL * T * F * 8, 0);
    for (int b = 0; b < B; b++)
        for (int c = 0; c < C; c++)
            for (int t = 0; t < T; t++)
                for (int f = 0; f < F; f++) {
                    const float x = src[(((size_t)b * C + c) * T + t) * F + f];
                    uint16_t parts[3];
                    if (PL == 1) parts[0] = f16_rne(x);
                    else {
                        parts[0] = bf16_rne(x);
                        const float r1 = x - bf16_to_f32(parts[0]);
                        parts[1] = bf16_rne(r1);
                        parts[2] = bf16_rne(r1 - bf16_to_f32(parts[1]));
                    }
                    for (int pl = 0; pl < PL; pl++) h[(((((size_t)b * C8 + c / 8) * PL + pl) * T + t) * F + f) * 8 + c % 8] = parts[pl];
                }
    HIPCHECK(e, hipMemcpy(dev, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    return SE_OK;
}

}  // namespace

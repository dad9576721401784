// se_engine.hip - persistent stream-batch engine behind the C ABI of include/se_engine.h.
//
// One engine owns, on one MI355X: all weights (re-laid-out once for the kernels), all per-stream
// state for B streams (ping-pong activation tensors that double as the reference's conv time buffers,
// GRU hidden state) and a fixed kernel sequence that advances every stream by one K-sample window:
//
//   k_stft -> k_featurize -> 4x (k_conv_igemm + k_gln) -> 2x (k_gemm_tn + T x k_gru_step) -> k_gemm_tn
//   -> k_gln -> 4x (k_conv_igemm even/odd [+ 1x1 skip GEMM + k_dec_blend]) -> k_final_mask -> k_istft
//
// Reference: TemporalCRN.forward / realtime_process (CRN.py:454-496, 560-589).  No CPU fallback exists:
// every entry point fails with SE_ERR_HIP if the device path is unavailable.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <array>
#include <map>
#include <string>
#include <vector>

#include "../../include/se_engine.h"
#include "conv_dispatch.h"
#include "convp_dispatch.h"
#include "fsn.hip.h"
#include "fft_lds.h"
#include "gemm.hip.h"
#include "norm.hip.h"
#include "stft.hip.h"

using namespace se;

namespace {

thread_local std::string g_create_error;

struct DevBuf {
    float *p = nullptr;
    size_t n = 0;
};

constexpr int kFftSub = 8;      // segments per STFT / iSTFT launch inside the pipelined path
constexpr int kPipeChunk = 64;  // segments per pipelined chunk (bounds spec_all / mask_all)
constexpr int kRing = 4;  // stage-crossing activation slots (see se_engine::slot)

struct ConvPlan {
    ConvArgs a{};  // pointers filled per launch
    int NT = 1;
    int grid_x = 0;
    size_t lds = 0;
    DevBuf w, bias;
    bool active = false;
    bool x6 = false;  // bf16x6 kernel (k_conv_x6) instead of the fp32-MFMA k_conv_igemm
    int CO = 1;       // k_conv_x6: channel octets staged per chunk
    DevBuf wx;
    DevBuf gatew;     // k_conv_small: fused gated 1x1 pair weights
    // k_conv_x6 tilings that fit (index = tiles per wave - 1); the weights do not depend on the tiling, so the one that
    // fills the chip with the fewest partial rounds is picked per batch size in select_conv_geometry()
    struct Geo { int NT = 0, tpw = 0, n_wg = 0, grouped = 0; size_t lds = 0; } geo[4];
    double flops = 0;  // algorithmic FLOPs per stream per launch (SURVEY.md 8d accounting)
};

struct Level {  // one encoder/decoder level
    ConvPlan enc;
    ConvPlan dec_even, dec_odd, skip;
    ConvPlan skipm;        // fused skip gate: statistics-only pass of the residualmask convolution
    bool skip_fused = false;
    ConvPlan gate[2];      // CRN_ELU encoder: conv_trans/conv_gated 1x1 pair, <= 64 output channels per launch
    ConvPlan pre;          // CRN_ELU preconv block i (levels 0..2): 5x5 frequency-dilated conv with the gated pair fused in
    DevBuf enc_nw, enc_nb, dec_nw, dec_nb, dec_mnw, dec_mnb, pre_nw, pre_nb;
};

}  // namespace

struct se_engine {
    se_config c{};
    int device = 0;
    int L = 0, T = 0, D = 0, M = 0, K = 0, N = 0, H = 0, NL = 0;
    int F[SE_MAX_LEVELS + 1]{};
    int Ch[SE_MAX_LEVELS + 1]{};  // Ch[0] = 2M-1, Ch[i+1] = channels[i]
    std::string err;
    std::map<std::string, std::vector<float>> params;  // host copies by canonical key
    std::map<std::string, std::vector<int64_t>> shapes;
    bool weights_ready = false;
    size_t conv_lds_budget = 48 * 1024;
    int gru_direct = -1;      // SE_GRU_DIRECT: 1 = always k_gru_step (W_hh streamed from L2), 0 = always k_gru_step2 (LDS slice),
                              // default -1 = k_gru_step in the overlapped (pipelined) bottleneck stage, k_gru_step2 otherwise
    int gru_seq = 0;          // SE_GRU_SEQ=1: one launch per layer (k_gru_seq, in-launch hand-off between steps) instead of one per
                              // time step (k_gru_step2).  Measured SLOWER on MI355X (17 us vs 13.5 us per step at B=256): off by default
    DevBuf gru_sync;          // [0,64) group counters, [64] timeout word of k_gru_seq

    // constant tables
    DevBuf window, env, tw;
    FftPlan plan{};

    // weights
    Level lv[SE_MAX_LEVELS];  // enc i at lv[i]; decoder j at lv[j] (dec_* members)
    DevBuf wih[4], whh[4], bih[4], bhh[4], fcw, fcb, gnw, gnb;
    DevBuf wih_x[4], fcw_x;  // bf16x3 planes [3][N][K] of the GEMM weights (k_gemm_bf16x6)
    int gemm_mode = 6;        // SE_GEMM_MODE: 0 = fp32 MFMA (k_gemm_tn), 6 = bf16x6 (default)
    int variant = 0, act = 1, eps_mode = 0, atan2_phase = 0, npre = 0;  // derived from se_config.variant
    int precision = 0;        // se_config.precision: 0 = bf16x6 (3 operand planes), 1 = fp16 operands (1 plane), 2 = bf16x3 (2 planes)
    int num_cu = 256;         // compute units of the device (MI355X: 256)
    int dec_merge = 1;        // SE_DEC_MERGE=0: narrow decoder blocks as two parity launches like the wide ones
    int skip_fuse = 1;        // SE_SKIP_FUSE=0: skip convolution writes both tensors, k_dec_blend_ew applies the gate
    int conv_small16 = 1;     // SE_CONV_SMALL16=0: first encoder block on k_conv_igemm instead of the vector-ALU kernel
    int conv_geo_fixed = 0;   // SE_CONV_GEO_FIXED=1: always the largest k_conv_x6 tiling (no per-batch selection)
    int conv_mode = 6;        // SE_CONV_MODE: 0 = fp32 MFMA (k_conv_igemm), 6 = bf16x6 where Cin % 8 == 0 (default)

    // state + activations for B streams
    int B = 0;
    int parity = 0;  // index of the "current" half of the encoder-private ping-pong buffers (pin)
    // Buffers that cross the three stages of a segment (encoder -> recurrent bottleneck -> decoder) live in a ring of
    // kRing slots, so that in se_realtime_process the encoder of segment n+1.. can run on its own HIP stream while the
    // latency-bound GRU recurrence of segment n and the decoder of segment n-1 are still in flight.  slot = segment % kRing;
    // xin[i][slot-1] doubles as the causal time history of the level-i convolution (CRN.py:333-334).
    int slot = 0;
    DevBuf spec[kRing], maskspec;
    DevBuf spec_all, mask_all;         // pipelined se_realtime_process: spectra / masked spectra of one chunk of segments
    DevBuf xin[SE_MAX_LEVELS][kRing];  // encoder level inputs (xin[0] = features)
    DevBuf enc_raw[SE_MAX_LEVELS];
    DevBuf gru_in[kRing], gi0[kRing], gil[4], seqr[4][kRing], hbuf[4][2], fc_out, dec_in[kRing];  // gi0 / seqr cross stage streams: rings
    int pipeline = 1;                  // SE_PIPELINE=0: one stream, stages back to back
    hipStream_t stage_stream[3]{};      // encoder (+ GRU input projection) / decoder (fc + norm first) / recurrence (all layers)
    hipEvent_t ev_enc[kRing]{}, ev_gru[kRing]{}, ev_dec[kRing]{}, ev_fork{}, ev_join{};
    int gru_lag = 1;                    // SE_GRU_LAG=0: layers back to back on the recurrence stream (NL * T launches per segment)
    bool stage_ready = false;
    int hcur[4]{};
    DevBuf dec_raw[SE_MAX_LEVELS], dec_uv[SE_MAX_LEVELS], dec_out[SE_MAX_LEVELS];
    DevBuf enc_stats[SE_MAX_LEVELS], dec_stats[SE_MAX_LEVELS], skip_stats[SE_MAX_LEVELS];  // [B][slots][2] norm partials
    DevBuf enc_g[SE_MAX_LEVELS];           // CRN_ELU: gated encoder output before the norm
    DevBuf pin[3][2], pre_g, pre_stats[3];  // CRN_ELU preconv chain (inputs ping-ponged: they carry 4 history columns)
    DevBuf yseg;
    DevBuf ragged_len;       // se_realtime_process_ragged: per-stream lengths (int64) on the device
    bool ragged_on = false;
    // Prefix compaction of a ragged batch: every layout is stream-major, so when the lengths are non-increasing the streams that still take
    // part in segment n are a PREFIX of the batch and every launch of that segment simply covers Bact < B streams (grids, GEMM rows, GRU
    // rows); strides and plane sizes stay those of the allocation batch B.  Bact is set per stage call by se_realtime_process.
    int Bact = 0;
    int bact_slot[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // launch batch of the segment living in each ring slot (the lagged GRU rounds mix segments)
    std::vector<int> ragged_nseg;  // per stream: segments it takes part in (empty: all)
    // second-generation convolution path (conv_p.hip.h): activations as split-bf16 planes; SE_PATH=0 selects the first generation
    struct se_convp_state *cp = nullptr;
    int path = 1;
    bool use_p = false;
    // second-generation bottleneck GEMMs (k_gemm_p): A operands arrive as split-bf16 planes from their producers
    bool gemm_p = false;      // decided in ensure_ready: plane path active, K dimensions multiples of 32 (SE_GEMM_P=0 disables);
                              // off for batches of <= 64 GEMM rows (se_reset), which run on the skinny fp32 kernel instead
    bool gemm_p_cap = false;
    // measured crossovers (profiles: B = 16 / 32 / 64 / 128, 512-pt): the skinny kernel wins up to 640 GEMM rows (B = 32:
    // 133 vs 228 us for the three GEMMs) and ties at 1280; the two-launch skip gate wins up to B = 64 (119 vs 147 us)
    int skinny_rows = 800;    // SE_GEMM_SKINNY_ROWS: bottleneck GEMMs with up to this many rows run on the skinny fp32 kernel
    int skip_min_batch = 96;  // SE_SKIP_MIN_BATCH: the streaming skip kernel needs at least this many streams
    int gemm_p_env = 1;
    int convp_deint = 1;      // SE_CONVP_DEINT=0: stride-2 convolutions keep interleaved LDS patch rows (2-way ds_read_b128 bank conflicts)
    int gemm_band = 0;        // SE_GEMM_BAND: 0 (default) = banded tile->XCD map where no equal 8-block split exists, 1 = always banded, -1 = never
    DevBuf gruinP[kRing], seqP[4][kRing];  // [PL][B*T][D'] / [PL][B*T][H] bf16 planes
    DevBuf wih_xp;                          // W_ih0 planes with K in the engine's feature order (k_gemm_p)
    int dbg_skip = 0;         // SE_DBG_SKIP bit mask, TIMING EXPERIMENTS ONLY (results are wrong): 1 = no GRU step launches, 2 = no bottleneck
                              // GEMMs, 4 = no encoder convolutions, 8 = no decoder
    int skip_stream = 1;      // SE_SKIP_STREAM=0: decoder skip gate as two k_conv_p launches instead of the streaming k_skip_p

    // optional per-kernel timing with HIP events on the launch stream (bench.py roofline leg)
    bool prof_on = false;
    struct ProfRec { int label; hipEvent_t a, b; };
    std::vector<ProfRec> prof_recs;
    std::vector<hipEvent_t> prof_pool;
    struct ProfLabel { std::string kernel, label; double flops; double ms = 0; long launches = 0; };
    std::vector<ProfLabel> prof_labels;
};

namespace {

int fail(se_engine *e, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (e) e->err = buf;
    else g_create_error = buf;
    return code;
}

#define HIPCHECK(e, call)                                                                      \
    do {                                                                                       \
        hipError_t _st = (call);                                                               \
        if (_st != hipSuccess)                                                                 \
            return fail(e, SE_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_st), \
                        __FILE__, __LINE__);                                                   \
    } while (0)

int dev_alloc(se_engine *e, DevBuf &b, size_t n) {
    if (b.p && b.n >= n) return 0;
    if (b.p) HIPCHECK(e, hipFree(b.p));
    b.p = nullptr;
    b.n = 0;
    HIPCHECK(e, hipMalloc(reinterpret_cast<void **>(&b.p), (n ? n : 1) * sizeof(float)));
    b.n = n;
    return 0;
}

int dev_upload(se_engine *e, DevBuf &b, const std::vector<float> &h) {
    int rc = dev_alloc(e, b, h.size());
    if (rc) return rc;
    HIPCHECK(e, hipMemcpy(b.p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

void dev_free(DevBuf &b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.n = 0;
}

int prof_label(se_engine *e, const char *kernel, const std::string &label, double flops) {
    for (size_t i = 0; i < e->prof_labels.size(); i++)
        if (e->prof_labels[i].label == label) { e->prof_labels[i].flops = flops; return (int)i; }
    e->prof_labels.push_back({kernel, label, flops});
    return (int)e->prof_labels.size() - 1;
}

struct ProfScope {  // brackets one launch with two events when profiling is on
    se_engine *e;
    hipStream_t st;
    int idx = -1;
    ProfScope(se_engine *e_, const char *kernel, const std::string &label, double flops, hipStream_t st_) : e(e_), st(st_) {
        if (!e->prof_on) return;
        hipEvent_t ev[2];
        for (auto &x : ev) {
            if (!e->prof_pool.empty()) { x = e->prof_pool.back(); e->prof_pool.pop_back(); }
            else if (hipEventCreate(&x) != hipSuccess) return;
        }
        e->prof_recs.push_back({prof_label(e, kernel, label, flops), ev[0], ev[1]});
        idx = (int)e->prof_recs.size() - 1;
        (void)hipEventRecord(ev[0], st);
    }
    ~ProfScope() {
        if (idx >= 0) (void)hipEventRecord(e->prof_recs[idx].b, st);
    }
};

uint16_t f16_rne(float x) {  // IEEE half, round to nearest even (the host compiler's _Float16 conversion)
    const _Float16 h = (_Float16)x;
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}
uint16_t bf16_rne(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
float bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
// operand planes per precision mode: 0 -> 3 bf16 planes (hi, mid, lo; six products), 1 -> 1 fp16 plane, 2 -> 2 bf16 planes
// (hi, mid; three products hi*hi + hi*mid + mid*hi)
inline int operand_planes(int precision) { return precision == 1 ? 1 : (precision == 2 ? 2 : 3); }

// bf16 planes (hi, mid, lo) of a [rows][cols] fp32 matrix -> device buffer of PL*rows*cols uint16
int upload_split3(se_engine *e, DevBuf &b, const std::vector<float> &w) {
    const size_t n = w.size();
    if (e->precision == 1) {  // one fp16 plane
        std::vector<uint16_t> plane(n);
        for (size_t i = 0; i < n; i++) plane[i] = f16_rne(w[i]);
        int rc = dev_alloc(e, b, (n + 1) / 2);
        if (rc) return rc;
        HIPCHECK(e, hipMemcpy(b.p, plane.data(), n * sizeof(uint16_t), hipMemcpyHostToDevice));
        return 0;
    }
    const int PL = operand_planes(e->precision);
    std::vector<uint16_t> planes(PL * n);
    for (size_t i = 0; i < n; i++) {
        const float x = w[i];
        const uint16_t h = bf16_rne(x);
        const float r1 = x - bf16_to_f32(h);
        const uint16_t m = bf16_rne(r1);
        const float r2 = r1 - bf16_to_f32(m);
        planes[i] = h; planes[n + i] = m;
        if (PL > 2) planes[2 * n + i] = bf16_rne(r2);
    }
    int rc = dev_alloc(e, b, (PL * n + 1) / 2);
    if (rc) return rc;
    HIPCHECK(e, hipMemcpy(b.p, planes.data(), planes.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    return 0;
}

std::string canon(const char *key) {
    std::string k(key);
    size_t pos = k.find(".net.0.");
    if (pos != std::string::npos) k.replace(pos, 7, ".conv.");  // CRN.py:314-316 alias
    return k;
}

const std::vector<float> *param(se_engine *e, const std::string &key, size_t expect) {
    auto it = e->params.find(key);
    if (it == e->params.end()) {
        fail(e, SE_ERR_PARAM_MISSING, "parameter %s was never loaded", key.c_str());
        return nullptr;
    }
    if (it->second.size() != expect) {
        fail(e, SE_ERR_SHAPE, "parameter %s has %zu elements, expected %zu", key.c_str(), it->second.size(), expect);
        return nullptr;
    }
    return &it->second;
}

// ---- conv planning --------------------------------------------------------------------------------
// taps: list of (kf, kt) with their patch offsets; wsel(ci, co, kf, kt) fetches the reference weight.
template <class WSel>
int plan_conv(se_engine *e, ConvPlan &pl, int Ci, int Co, int FP, int Fi, int Fy, int s, int os, int oo, int colpad,
              int tlo_off, int ngroup, int dil, int St, const std::vector<std::array<int, 4>> &taps /*kf,kt,rowgrp,coloff*/,
              WSel wsel, const std::vector<float> &bias, int relu_lo, int relu_hi, int act = 1, int gate_pairs = 0, int Cy = -1,
              int cy0 = 0) {
    if (Cy < 0) Cy = Co;
    pl.a.par_rows = 0;
    pl.x6 = false;
    pl.active = FP > 0;
    if (!pl.active) return 0;
    const int T = e->T;
    const int ntap = (int)taps.size();
    const int P = T * FP, tiles = (P + 31) / 32;
    // vector-ALU variant: one thread per position, direct global reads, weights [tap][ci][CW] in LDS.  Also taken by the first
    // encoder block (5 -> 16 channels at full frequency resolution): K = 75 is too shallow for the matrix pipe to matter and
    // the patch halo of the MFMA kernels costs more than the arithmetic
    if ((Co <= 8 || (Co <= 16 && Ci <= 8 && e->conv_small16)) && !gate_pairs) {
        const int CW = Co <= 4 ? 4 : (Co <= 8 ? 8 : 16);
        ConvArgs &a = pl.a;
        a.Ci = Ci; a.Co = Co; a.CoPad = CW; a.T = T; a.Fi = Fi; a.FP = FP; a.Fy = Fy;
        a.s = s; a.os = os; a.oo = oo; a.colpad = colpad; a.tlo_off = tlo_off; a.ngroup = ngroup; a.dil = dil; a.grouped = 0;
        a.ntap = ntap; a.CC = Ci; a.nchunk = 1; a.tiles_per_wg = 8; a.St = St;
        a.relu_lo = relu_lo; a.relu_hi = relu_hi; a.act = act; a.gate_pairs = 0; a.Cy = Cy; a.cy0 = cy0;
        for (int t = 0; t < ntap; t++) { a.rowgrp[t] = taps[t][2]; a.coloff[t] = taps[t][3]; }
        pl.NT = 0;  // marks the small kernel
        pl.grid_x = (P + 255) / 256;
        pl.lds = sizeof(float) * (size_t)ntap * Ci * CW;
        if (pl.lds > 64 * 1024) return fail(e, SE_ERR_ARG, "small-conv weights do not fit LDS");
        std::vector<float> w((size_t)ntap * Ci * CW, 0.0f);
        for (int t = 0; t < ntap; t++)
            for (int ci = 0; ci < Ci; ci++)
                for (int co = 0; co < Co; co++) w[((size_t)t * Ci + ci) * CW + co] = wsel(ci, co, taps[t][0], taps[t][1]);
        int rc = dev_upload(e, pl.w, w);
        if (rc) return rc;
        return dev_upload(e, pl.bias, bias);
    }
    const int CoPad = (Co + 31) / 32 * 32;
    if (CoPad > 128) return fail(e, SE_ERR_ARG, "conv with %d output channels is not supported (max 128 per GEMM)", Co);
    const int MT = CoPad / 32;
    if (MT == 3) return fail(e, SE_ERR_ARG, "conv output channels %d need 3 row tiles (unsupported)", Co);
    const int NCG = 4 / MT, NTmax = 4;
    if (e->conv_mode == 6 && (Ci % 8 == 0 || Ci >= 5)) {  // ---- bf16x6 path: K step = 2 taps x 8 channels (Cin zero-padded to 8s) ----
        // channel octets per chunk: 1x1 convolutions take up to 4 (32 channels) so that both halves of every K step carry
        // real channels and a chunk holds several K steps; multi-tap convolutions already have NTAP entries per octet
        int CO = 1;
        if (ntap == 1) { const int oct = (Ci + 7) / 8; CO = oct >= 4 ? 4 : (oct >= 2 ? 2 : 1); }
        if (const char *s = getenv("SE_X6_CO")) CO = std::max(1, std::min(CO, atoi(s)));
        int tpw = 0, n_wg = 0, NT = 0, Rmax = 0, grouped = 0;
        for (; CO >= 1; CO >>= 1) {
            for (int k = 0; k < 4; k++) pl.geo[k] = ConvPlan::Geo{};
            NT = 0;
            for (int ntmax = 1; ntmax <= NTmax; ntmax++) {  // ends on the largest tiling that fits = the default geometry
                const int c_wg = (tiles + NCG * ntmax - 1) / (NCG * ntmax);
                const int c_tpw = (tiles + c_wg - 1) / c_wg;
                const int c_NT = (c_tpw + NCG - 1) / NCG;
                int rows_pos = (c_tpw * 32 + FP - 1) / FP + 1;
                if (rows_pos > T) rows_pos = T;
                const int c_grouped = ngroup * rows_pos < rows_pos + (ngroup - 1) * dil;
                const int c_Rmax = c_grouped ? ngroup * rows_pos : rows_pos + (ngroup - 1) * dil;
                if ((size_t)CO * c_Rmax * St > 256 * kX6PosPerThread) continue;
                ConvPlan::Geo &g = pl.geo[c_NT - 1];
                g.NT = c_NT; g.tpw = c_tpw; g.n_wg = c_wg; g.grouped = c_grouped;
                g.lds = std::max<size_t>((size_t)operand_planes(e->precision) * CO * c_Rmax * St * 16, 64);
                tpw = c_tpw; n_wg = c_wg; NT = c_NT; Rmax = c_Rmax; grouped = c_grouped;
            }
            if (NT > 0) break;
        }
        if (NT > 0) {
            const int nstep = (ntap * CO + 1) / 2, nchunk = (Ci + 8 * CO - 1) / (8 * CO);
            ConvArgs &a = pl.a;
            a.Ci = Ci; a.Co = Co; a.CoPad = CoPad; a.T = T; a.Fi = Fi; a.FP = FP; a.Fy = Fy;
            a.s = s; a.os = os; a.oo = oo; a.colpad = colpad; a.tlo_off = tlo_off; a.ngroup = ngroup; a.dil = dil; a.grouped = grouped;
            a.ntap = ntap; a.CC = 8 * CO; a.nchunk = nchunk; a.tiles_per_wg = tpw; a.St = St;
            a.relu_lo = relu_lo; a.relu_hi = relu_hi; a.act = act; a.gate_pairs = gate_pairs; a.Cy = Cy; a.cy0 = cy0;
            for (int t = 0; t < ntap; t++) { a.rowgrp[t] = taps[t][2]; a.coloff[t] = taps[t][3]; }
            pl.NT = NT; pl.grid_x = n_wg; pl.x6 = true; pl.CO = CO;
            pl.lds = std::max<size_t>((size_t)operand_planes(e->precision) * CO * Rmax * St * 16, 64);
            // weights: [chunk][step][plane][mtile][co 32][k 16], k = half*8 + c <-> entry 2*step+half = (tap, octet) tap-major,
            // channel chunk*8*CO + octet*8 + c
            const int PL = operand_planes(e->precision);
            std::vector<uint16_t> wx((size_t)nchunk * nstep * PL * MT * 32 * 16, 0);
            for (int ch = 0; ch < nchunk; ch++)
                for (int st = 0; st < nstep; st++)
                    for (int m = 0; m < MT; m++)
                        for (int r = 0; r < 32; r++)
                            for (int k = 0; k < 16; k++) {
                                const int en = 2 * st + k / 8, tp = en / CO, oc = en % CO;
                                const int ci = (ch * CO + oc) * 8 + k % 8, co = m * 32 + r;
                                if (tp >= ntap || co >= Co || ci >= Ci) continue;
                                const float x = wsel(ci, co, taps[tp][0], taps[tp][1]);
                                const uint16_t h = bf16_rne(x);
                                const float r1 = x - bf16_to_f32(h);
                                const uint16_t md = bf16_rne(r1);
                                const float r2 = r1 - bf16_to_f32(md);
                                const uint16_t parts[3] = {PL == 1 ? f16_rne(x) : h, md, bf16_rne(r2)};
                                for (int pln = 0; pln < PL; pln++)
                                    wx[(((((size_t)ch * nstep + st) * PL + pln) * MT + m) * 32 + r) * 16 + k] = parts[pln];
                            }
            int rc = dev_alloc(e, pl.wx, (wx.size() + 1) / 2);
            if (rc) return rc;
            HIPCHECK(e, hipMemcpy(pl.wx.p, wx.data(), wx.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
            return dev_upload(e, pl.bias, bias);
        }
    }
    const int n_wg = (tiles + NCG * NTmax - 1) / (NCG * NTmax);
    const int tpw = (tiles + n_wg - 1) / n_wg;
    const int NT = (tpw + NCG - 1) / NCG;
    int rows_pos = (tpw * 32 + FP - 1) / FP + 1;
    if (rows_pos > T) rows_pos = T;
    const int grouped = ngroup * rows_pos < rows_pos + (ngroup - 1) * dil;
    const int Rmax = grouped ? ngroup * rows_pos : rows_pos + (ngroup - 1) * dil;
    const int CiPad = (Ci + 1) / 2 * 2;
    auto bytes = [&](int cc) { return sizeof(float) * ((size_t)ntap * cc * CoPad + (size_t)cc * Rmax * St); };
    int CC = CiPad;
    auto fits = [&](int cc) {
        return bytes(cc) <= e->conv_lds_budget && (size_t)cc * Rmax * St <= 256 * kPatchPerThread &&
               (size_t)ntap * cc * CoPad <= 256 * 4 * kWeightPerThread;
    };
    while (CC > 2 && !fits(CC)) CC -= 2;
    if (!fits(CC)) return fail(e, SE_ERR_ARG, "conv chunk does not fit the staging registers (Rmax %d, St %d, taps %d, Cout %d)", Rmax, St, ntap, Co);
    int nchunk = (CiPad + CC - 1) / CC;
    CC = ((CiPad + nchunk - 1) / nchunk + 1) / 2 * 2;
    nchunk = (CiPad + CC - 1) / CC;
    if (bytes(CC) > 150 * 1024) return fail(e, SE_ERR_ARG, "conv tile does not fit LDS (%zu bytes)", bytes(CC));
    ConvArgs &a = pl.a;
    a.Ci = Ci; a.Co = Co; a.CoPad = CoPad; a.T = T; a.Fi = Fi; a.FP = FP; a.Fy = Fy;
    a.s = s; a.os = os; a.oo = oo; a.colpad = colpad; a.tlo_off = tlo_off; a.ngroup = ngroup; a.dil = dil; a.grouped = grouped;
    a.ntap = ntap; a.CC = CC; a.nchunk = nchunk; a.tiles_per_wg = tpw; a.St = St;
    a.relu_lo = relu_lo; a.relu_hi = relu_hi; a.act = act; a.gate_pairs = gate_pairs; a.Cy = Cy; a.cy0 = cy0;
    for (int t = 0; t < ntap; t++) { a.rowgrp[t] = taps[t][2]; a.coloff[t] = taps[t][3]; }
    pl.NT = NT;
    pl.grid_x = n_wg;
    pl.lds = bytes(CC);
    std::vector<float> w((size_t)nchunk * ntap * CC * CoPad, 0.0f);
    for (int ch = 0; ch < nchunk; ch++)
        for (int t = 0; t < ntap; t++)
            for (int c = 0; c < CC; c++) {
                const int ci = ch * CC + c;
                if (ci >= Ci) continue;
                float *dst = &w[(((size_t)ch * ntap + t) * CC + c) * CoPad];
                for (int co = 0; co < Co; co++) dst[co] = wsel(ci, co, taps[t][0], taps[t][1]);
            }
    int rc = dev_upload(e, pl.w, w);
    if (rc) return rc;
    return dev_upload(e, pl.bias, bias);
}

int prepare_weights(se_engine *e) {
    if (e->weights_ready) return 0;
    const int L = e->L, H = e->H, D = e->D;
    for (int i = 0; i < L; i++) {
        const int Ci = e->Ch[i], Co = e->Ch[i + 1], Fi = e->F[i], Fo = e->F[i + 1], d = 1 << i;
        const std::string p = "convlist." + std::to_string(i) + ".";
        auto *w = param(e, p + "conv.weight", (size_t)Co * Ci * 15);
        auto *b = param(e, p + "conv.bias", Co);
        auto *nw = param(e, p + "norm.weight", Co);
        auto *nb = param(e, p + "norm.bias", Co);
        if (!w || !b || !nw || !nb) return SE_ERR_PARAM_MISSING;
        std::vector<std::array<int, 4>> taps;
        for (int kf = 0; kf < 5; kf++)
            for (int kt = 0; kt < 3; kt++) taps.push_back({kf, kt, kt, kf});
        const float *wp = w->data();
        int rc = plan_conv(e, e->lv[i].enc, Ci, Co, Fo, Fi, Fo, 2, 1, 0, 2, -2 * d, 3, d, Fi + 4, taps,
                           [=](int ci, int co, int kf, int kt) { return wp[(((size_t)co * Ci + ci) * 5 + kf) * 3 + kt]; },
                           *b, 0, Co, e->act);
        if (rc) return rc;
        e->lv[i].enc.flops = 2.0 * Co * Ci * 15 * Fo * e->T;
        e->lv[i].gate[0].active = e->lv[i].gate[1].active = false;
        if (e->variant) {  // conv_trans / conv_gated 1x1 pair (CRN_ELU.py:223-224,240), <= 64 output channels per launch
            auto *tw = param(e, p + "conv_trans.weight", (size_t)Co * Co);
            auto *tb = param(e, p + "conv_trans.bias", Co);
            auto *gw = param(e, p + "conv_gated.weight", (size_t)Co * Co);
            auto *gb = param(e, p + "conv_gated.bias", Co);
            if (!tw || !tb || !gw || !gb) return SE_ERR_PARAM_MISSING;
            const float *twp = tw->data(), *gwp = gw->data();
            const int nparts = Co > 64 ? 2 : 1, cpart = (Co + nparts - 1) / nparts;
            std::vector<std::array<int, 4>> t1 = {{0, 0, 0, 0}};
            for (int part = 0; part < nparts; part++) {
                const int c0 = part * cpart, cn = std::min(cpart, Co - c0);
                std::vector<float> bias2(2 * cn);
                for (int c = 0; c < cn; c++) { bias2[2 * c] = (*tb)[c0 + c]; bias2[2 * c + 1] = (*gb)[c0 + c]; }
                rc = plan_conv(e, e->lv[i].gate[part], Co, 2 * cn, Fo, Fo, Fo, 1, 1, 0, 0, 0, 1, 0, Fo, t1,
                               [=](int ci, int row, int, int) { const int c = c0 + row / 2; return (row & 1) ? gwp[(size_t)c * Co + ci] : twp[(size_t)c * Co + ci]; },
                               bias2, 0, 0, 0, /*gate_pairs=*/1, /*Cy=*/Co, /*cy0=*/c0);
                if (rc) return rc;
                e->lv[i].gate[part].flops = 2.0 * 2 * cn * Co * Fo * e->T;
            }
        }
        if ((rc = dev_upload(e, e->lv[i].enc_nw, *nw))) return rc;
        if ((rc = dev_upload(e, e->lv[i].enc_nb, *nb))) return rc;
    }
    for (int j = 0; j < L; j++) {
        const int lvl = L - 1 - j;
        const int Ci = e->Ch[lvl + 1], Co = lvl == 0 ? 2 : e->Ch[lvl], Fi = e->F[lvl + 1], Fo = 2 * Fi - 1, d = 1 << j;
        const std::string p = "deconvlist." + std::to_string(j) + ".";
        auto *w = param(e, p + "conv.weight", (size_t)Co * Ci * 15);
        auto *b = param(e, p + "conv.bias", Co);
        auto *nw = param(e, p + "norm.weight", Co);
        auto *nb = param(e, p + "norm.bias", Co);
        if (!w || !b || !nw || !nb) return SE_ERR_PARAM_MISSING;
        const float *wp = w->data();
        auto wsel = [=](int ci, int co, int kf, int kt) { return wp[(((size_t)ci * Co + co) * 5 + kf) * 3 + kt]; };
        std::vector<std::array<int, 4>> te, to;
        for (int kf = 0; kf < 5; kf += 2)
            for (int kt = 0; kt < 3; kt++) te.push_back({kf, kt, 2 - kt, 2 - kf / 2});
        for (int kf = 1; kf < 5; kf += 2)
            for (int kt = 0; kt < 3; kt++) to.push_back({kf, kt, 2 - kt, 1 + (3 - kf) / 2});
        int rc = 0;
        bool merged = false;
        if (e->dec_merge && Co > 4 && Co <= 16 && e->conv_mode == 6) {
            // narrow block: both parities in ONE 15-tap launch, rows (2c, 2c+1) = (even, odd) parity of channel c (ConvArgs::par_rows)
            std::vector<std::array<int, 4>> tu;
            for (int kf = 0; kf < 5; kf++)
                for (int kt = 0; kt < 3; kt++) tu.push_back({kf, kt, 2 - kt, (kf & 1) ? 1 + (3 - kf) / 2 : 2 - kf / 2});
            std::vector<float> bias2(2 * Co);
            for (int c = 0; c < Co; c++) bias2[2 * c] = bias2[2 * c + 1] = (*b)[c];
            rc = plan_conv(e, e->lv[j].dec_even, Ci, 2 * Co, Fi, Fi, Fo, 1, 2, 0, 1, 0, 3, d, Fi + 2, tu,
                           [=](int ci, int row, int kf, int kt) { return ((row & 1) == (kf & 1)) ? wsel(ci, row >> 1, kf, kt) : 0.0f; },
                           bias2, 0, 2 * Co, e->act, 0, /*Cy=*/Co, 0);
            if (rc) return rc;
            merged = e->lv[j].dec_even.x6;
            if (merged) { e->lv[j].dec_even.a.par_rows = 1; e->lv[j].dec_odd.active = false; e->lv[j].dec_odd.grid_x = 0; }
        }
        if (!merged) {
            rc = plan_conv(e, e->lv[j].dec_even, Ci, Co, Fi, Fi, Fo, 1, 2, 0, 1, 0, 3, d, Fi + 2, te, wsel, *b, 0, Co, e->act);
            if (rc) return rc;
            rc = plan_conv(e, e->lv[j].dec_odd, Ci, Co, Fi - 1, Fi, Fo, 1, 2, 1, 1, 0, 3, d, Fi + 2, to, wsel, *b, 0, Co, e->act);
            if (rc) return rc;
        }
        // SURVEY 8d counts a transposed conv as Cin*Cout*15*Fi*T MACs; split 9:6 over the two parity launches
        e->lv[j].dec_even.flops = 2.0 * Ci * Co * (merged ? 15 : 9) * Fi * e->T;
        e->lv[j].dec_odd.flops = merged ? 0.0 : 2.0 * Ci * Co * 6 * Fi * e->T;
        if ((rc = dev_upload(e, e->lv[j].dec_nw, *nw))) return rc;
        if ((rc = dev_upload(e, e->lv[j].dec_nb, *nb))) return rc;
        if (lvl > 0) {  // skip path exists (CRN.py:485-487)
            auto *mw = param(e, p + "residualmask.weight", (size_t)Co * Co);
            auto *mb = param(e, p + "residualmask.bias", Co);
            auto *rw = param(e, p + "residual.weight", (size_t)Co * Co);
            auto *rb = param(e, p + "residual.bias", Co);
            auto *mnw = param(e, p + "residualnorm.weight", Co);
            auto *mnb = param(e, p + "residualnorm.bias", Co);
            if (!mw || !mb || !rw || !rb || !mnw || !mnb) return SE_ERR_PARAM_MISSING;
            const float *mwp = mw->data(), *rwp = rw->data();
            std::vector<float> bias2(2 * Co);
            for (int c = 0; c < Co; c++) { bias2[c] = (*mb)[c]; bias2[Co + c] = (*rb)[c]; }
            std::vector<std::array<int, 4>> t1 = {{0, 0, 0, 0}};
            const int Fr = e->F[lvl];
            // Fused form (k_conv_x6 only): rows (2c, 2c+1) = (residualmask_c, residual_c), gate applied in the epilogue, preceded
            // by a statistics-only pass of the residualmask rows.  Falls back to the two-tensor form + k_dec_blend_ew otherwise.
            e->lv[j].skip_fused = false;
            e->lv[j].skipm.active = false;
            if (e->skip_fuse) {
                std::vector<float> biasp(2 * Co);
                for (int c = 0; c < Co; c++) { biasp[2 * c] = (*mb)[c]; biasp[2 * c + 1] = (*rb)[c]; }
                rc = plan_conv(e, e->lv[j].skip, Co, 2 * Co, Fr, Fr, Fr, 1, 1, 0, 0, 0, 1, 0, Fr, t1,
                               [=](int ci, int row, int, int) { const int c = row >> 1; return (row & 1) ? rwp[(size_t)c * Co + ci] : mwp[(size_t)c * Co + ci]; },
                               biasp, 0, 0, e->act, /*gate_pairs=*/0, /*Cy=*/Co, /*cy0=*/0);
                if (rc) return rc;
                if (e->lv[j].skip.x6) {
                    rc = plan_conv(e, e->lv[j].skipm, Co, Co, Fr, Fr, Fr, 1, 1, 0, 0, 0, 1, 0, Fr, t1,
                                   [=](int ci, int co, int, int) { return mwp[(size_t)co * Co + ci]; }, *mb, 0, 0, e->act);
                    if (rc) return rc;
                    e->lv[j].skip_fused = e->lv[j].skipm.x6;
                    e->lv[j].skipm.flops = 0;  // recomputation, not algorithmic work (SURVEY 8d accounting counts the skip GEMM once)
                }
            }
            if (!e->lv[j].skip_fused) {
                e->lv[j].skipm.active = false;
                rc = plan_conv(e, e->lv[j].skip, Co, 2 * Co, Fr, Fr, Fr, 1, 1, 0, 0, 0, 1, 0, Fr, t1,
                               [=](int ci, int co, int, int) { return co < Co ? mwp[(size_t)co * Co + ci] : rwp[(size_t)(co - Co) * Co + ci]; },
                               bias2, Co, 2 * Co, e->act);
                if (rc) return rc;
            }
            e->lv[j].skip.flops = 2.0 * 2 * Co * Co * Fr * e->T;
            if ((rc = dev_upload(e, e->lv[j].dec_mnw, *mnw))) return rc;
            if ((rc = dev_upload(e, e->lv[j].dec_mnb, *mnb))) return rc;
        }
    }
    for (int i = 0; i < e->npre; i++) {  // CRN_ELU.py:335-340: Conv2d(5x5, dilation (fd,1), padding (2fd,4)) + gated pair + gLN
        const int C0 = e->Ch[0], F0 = e->F[0], fd = 1 << i;
        const std::string p = "preconvlist." + std::to_string(i) + ".";
        auto *w = param(e, p + "conv.weight", (size_t)C0 * C0 * 25);
        auto *b = param(e, p + "conv.bias", C0);
        auto *tw = param(e, p + "conv_trans.weight", (size_t)C0 * C0);
        auto *tb = param(e, p + "conv_trans.bias", C0);
        auto *gw = param(e, p + "conv_gated.weight", (size_t)C0 * C0);
        auto *gb = param(e, p + "conv_gated.bias", C0);
        auto *nw = param(e, p + "norm.weight", C0);
        auto *nb = param(e, p + "norm.bias", C0);
        if (!w || !b || !tw || !tb || !gw || !gb || !nw || !nb) return SE_ERR_PARAM_MISSING;
        if (C0 > 8) return fail(e, SE_ERR_ARG, "preconv blocks support up to 8 feature channels (num_inputs <= 4)");
        std::vector<std::array<int, 4>> taps;
        for (int kf = 0; kf < 5; kf++)
            for (int kt = 0; kt < 5; kt++) taps.push_back({kf, kt, kt, kf * fd});
        const float *wp = w->data(), *twp = tw->data(), *gwp = gw->data();
        int rc = plan_conv(e, e->lv[i].pre, C0, C0, F0, F0, F0, 1, 1, 0, 2 * fd, -4, 5, 1, F0 + 4 * fd, taps,
                           [=](int ci, int co, int kf, int kt) { return wp[(((size_t)co * C0 + ci) * 5 + kf) * 5 + kt]; }, *b, 0, C0, 2);
        if (rc) return rc;
        e->lv[i].pre.flops = 2.0 * C0 * C0 * 25 * F0 * e->T;
        {  // the gated 1x1 pair is fused into the conv's epilogue (all 5 channels of a position sit in one thread)
            std::vector<float> g;
            g.insert(g.end(), twp, twp + (size_t)C0 * C0);
            g.insert(g.end(), gwp, gwp + (size_t)C0 * C0);
            g.insert(g.end(), tb->begin(), tb->end());
            g.insert(g.end(), gb->begin(), gb->end());
            if ((rc = dev_upload(e, e->lv[i].pre.gatew, g))) return rc;
            e->lv[i].pre.lds += g.size() * sizeof(float);
            e->lv[i].pre.flops += 2.0 * 2 * C0 * C0 * F0 * e->T;
        }
        if ((rc = dev_upload(e, e->lv[i].pre_nw, *nw)) || (rc = dev_upload(e, e->lv[i].pre_nb, *nb))) return rc;
    }
    for (int l = 0; l < e->NL; l++) {
        const std::string s = std::to_string(l);
        const size_t in = l == 0 ? D : H;
        auto *a = param(e, "gru.sequence_model.weight_ih_l" + s, 3 * (size_t)H * in);
        auto *b = param(e, "gru.sequence_model.weight_hh_l" + s, 3 * (size_t)H * H);
        auto *c = param(e, "gru.sequence_model.bias_ih_l" + s, 3 * (size_t)H);
        auto *d = param(e, "gru.sequence_model.bias_hh_l" + s, 3 * (size_t)H);
        if (!a || !b || !c || !d) return SE_ERR_PARAM_MISSING;
        int rc;
        if ((rc = upload_split3(e, e->wih_x[l], *a))) return rc;
        if ((rc = dev_upload(e, e->wih[l], *a)) || (rc = dev_upload(e, e->whh[l], *b)) ||
            (rc = dev_upload(e, e->bih[l], *c)) || (rc = dev_upload(e, e->bhh[l], *d)))
            return rc;
    }
    {
        auto *a = param(e, "gru.fc_output_layer.weight", (size_t)D * H);
        auto *b = param(e, "gru.fc_output_layer.bias", D);
        auto *c = param(e, "gru.norm.weight", D);
        auto *d = param(e, "gru.norm.bias", D);
        if (!a || !b || !c || !d) return SE_ERR_PARAM_MISSING;
        int rc;
        if ((rc = upload_split3(e, e->fcw_x, *a))) return rc;
        if ((rc = dev_upload(e, e->fcw, *a)) || (rc = dev_upload(e, e->fcb, *b)) || (rc = dev_upload(e, e->gnw, *c)) ||
            (rc = dev_upload(e, e->gnb, *d)))
            return rc;
    }
    e->weights_ready = true;
    return 0;
}

// Picks, for the current batch, the k_conv_x6 tiling with the least modelled time.  Measured on MI355X (NT sweep,
// profiles/r01_v4_conv_nt_sweep.txt): a workgroup costs ~(NT + 0.28) units (the constant is the first weight fragments and
// the barriers), and a partial last round of resident workgroups costs about its share (a workgroup alone on a CU gets the
// matrix pipe to itself), so time ~ workgroups x (NT + 0.28), quantised only when the whole launch is below one round.
void select_conv_geometry(se_engine *e, ConvPlan &pl) {
    if (!pl.active || !pl.x6) return;
    const double slots = 2.0 * e->num_cu;
    double best = 0;
    int pick = -1;
    for (int k = 0; k < 4; k++) {
        const ConvPlan::Geo &g = pl.geo[k];
        if (!g.NT) continue;
        const double rounds = std::max(1.0, (double)g.n_wg * e->B / slots);
        const double cost = rounds * (g.NT + 0.28);
        if (pick < 0 || cost <= best) { best = cost; pick = k; }
    }
    if (e->conv_geo_fixed) for (int k = 3; k >= 0; k--) if (pl.geo[k].NT) { pick = k; break; }
    if (const char *s = getenv("SE_CONV_NT")) { const int k = atoi(s) - 1; if (k >= 0 && k < 4 && pl.geo[k].NT) pick = k; }
    const ConvPlan::Geo &g = pl.geo[pick];
    pl.NT = g.NT; pl.grid_x = g.n_wg; pl.lds = g.lds;
    pl.a.tiles_per_wg = g.tpw; pl.a.grouped = g.grouped;
}

struct ConvBlend {  // operands of the fused decoder skip gate (ConvArgs::blend)
    const float *ydec, *nw, *nb, *mnw, *mnb;
    SlabStats sy, su;
    int Fo;
};

int launch_conv(se_engine *e, const ConvPlan &pl, const float *x, const float *xprev, float *y, hipStream_t st, const char *label,
                float *stats = nullptr, int nslot = 0, int slot0 = 0, int stats_lo = 0, int stats_hi = 0, const ConvBlend *blend = nullptr) {
    if (!pl.active) return 0;
    // algorithmic MACs of this launch as SURVEY.md 8d counts them are attributed by the caller via pl.flops
    ProfScope ps(e, pl.x6 ? "k_conv_x6" : (pl.NT == 0 ? "k_conv_small" : "k_conv_igemm"), label, pl.flops * e->B, st);
    ConvArgs a = pl.a;
    a.x = x; a.xprev = xprev; a.y = y; a.w = pl.w.p; a.bias = pl.bias.p; a.gatew = pl.gatew.p;
    a.stats = stats; a.stats_nslot = nslot; a.stats_slot0 = slot0; a.stats_lo = stats_lo; a.stats_hi = stats_hi;
    a.blend = 0;
    if (blend) {
        if (!pl.x6 || a.ntap != 1) return fail(e, SE_ERR_ARG, "fused skip gate needs the 1x1 k_conv_x6 path");
        a.blend = 1; a.bl_ydec = blend->ydec; a.bl_nw = blend->nw; a.bl_nb = blend->nb; a.bl_mnw = blend->mnw; a.bl_mnb = blend->mnb;
        a.bl_sy = blend->sy; a.bl_su = blend->su; a.bl_Fo = blend->Fo;
    }
    dim3 grid(pl.grid_x, e->B);
    if (pl.x6) {
        ConvX6Args xa{a, reinterpret_cast<const uint4 *>(pl.wx.p)};
#ifdef SE_X6_TRACE
        { const char *want = getenv("SE_X6_TRACE_LABEL"); xa.trace_slot = (want && label && strcmp(want, label) == 0) ? 0 : -1; }
#endif
        if (conv_x6_launch(a.ntap, pl.NT, pl.CO, operand_planes(e->precision), grid, pl.lds, st, xa))
            return fail(e, SE_ERR_ARG, "no x6 conv kernel instance for %d taps x %d tiles x %d octets", a.ntap, pl.NT, pl.CO);
        HIPCHECK(e, hipGetLastError());
        return 0;
    }
    if (conv_igemm_launch(a.ntap, pl.NT, a.CoPad, grid, pl.lds, st, a))
        return fail(e, SE_ERR_ARG, "no conv kernel instance for %d taps x %d tiles", a.ntap, pl.NT);
    HIPCHECK(e, hipGetLastError());
    return 0;
}

int launch_gemm(se_engine *e, const float *A, long lda, const float *W, long ldw, const float *bias, float *C, long ldc,
                int Mr, int Nc, int Kd, int relu, hipStream_t st, const char *label, const float *Wx = nullptr) {
    if (Mr <= e->skinny_rows && lda == Kd && ldw == Kd && Kd % 8 == 0 && e->gemm_mode == 6) {  // few rows: 32 x 32 tiles, K split over the waves (fp32-exact MFMA)
        ProfScope ps(e, "k_gemm_skinny", label, 2.0 * Mr * Nc * Kd, st);
        if (se_train_gemm(A, W, bias, C, Mr, Nc, Kd, relu, st)) return fail(e, SE_ERR_HIP, "skinny GEMM launch failed: %s", se_train_last_error());
        return 0;
    }
    dim3 ggrid((Nc + kGemmBN - 1) / kGemmBN, (Mr + kGemmBM - 1) / kGemmBM);
    if (Wx && e->gemm_mode == 6 && Kd % 8 == 0 && Kd >= 8 && lda % 4 == 0 && ldw == Kd) {
        ProfScope ps(e, e->precision == 1 ? "k_gemm_f16" : (e->precision == 2 ? "k_gemm_bf16x3" : "k_gemm_bf16x6"), label, 2.0 * Mr * Nc * Kd, st);
        GemmX6Args g{A, reinterpret_cast<const __bf16 *>(Wx), bias, C, Mr, Nc, Kd, lda, ldc, relu};
        if (e->precision == 1) hipLaunchKernelGGL(k_gemm_x<1>, ggrid, dim3(256), 0, st, g);
        else if (e->precision == 2) hipLaunchKernelGGL(k_gemm_x<2>, ggrid, dim3(256), 0, st, g);
        else hipLaunchKernelGGL(k_gemm_x<3>, ggrid, dim3(256), 0, st, g);
        HIPCHECK(e, hipGetLastError());
        return 0;
    }
    ProfScope ps(e, "k_gemm_tn", label, 2.0 * Mr * Nc * Kd, st);
    if (Kd % 4 || Kd < 4 || lda % 4 || ldw % 4) return fail(e, SE_ERR_ARG, "GEMM inner dimension %d must be a multiple of 4", Kd);
    GemmArgs g{A, W, bias, C, Mr, Nc, Kd, lda, ldw, ldc, relu};
    dim3 grid((Nc + kGemmBN - 1) / kGemmBN, (Mr + kGemmBM - 1) / kGemmBM);
    hipLaunchKernelGGL(k_gemm_tn, grid, dim3(256), 0, st, g);
    HIPCHECK(e, hipGetLastError());
    return 0;
}

// C = act(A W^T + bias) with both operands as split-bf16 planes ([PL][rows][K] bf16, K % 32 == 0): k_gemm_p
int launch_gemm_p(se_engine *e, const float *Ap, const float *Wp, const float *bias, float *C, long ldc, int Mr, int Nc, int Kd, int relu,
                  hipStream_t st, const char *label) {
    const int PL = operand_planes(e->precision);
    ProfScope ps(e, "k_gemm_p", label, 2.0 * Mr * Nc * Kd, st);
    const int nrt = (Mr + kGemmPBM - 1) / kGemmPBM, nct = (Nc + kGemmPBN - 1) / kGemmPBN;
    // XCD blocks: the split of the tile grid into 8 equal blocks (gx x gy) that fetches the fewest bytes into the 8 private
    // L2s - every row tile of A is fetched by the gy XCDs of its block row, every column tile of W by gx
    int gx = 0, gy = 0;
    double best = 0;
    for (int cx : {1, 2, 4, 8}) {
        if (nrt % cx || nct % (8 / cx)) continue;
        const double bytes = (double)Mr * (8 / cx) + (double)Nc * cx;  // x Kd x bytes per element, common to all candidates
        if (!gx || bytes < best) { gx = cx; gy = 8 / cx; best = bytes; }
    }
    int nblocks = nrt * nct;
    if (e->gemm_band >= 0 && (!gx || e->gemm_band == 1) && nrt * nct >= 16) {
        // no equal 8-block split divides the tile grid (B = 256: 21 x 12 tiles): block id -> XCD is id mod 8, so plain row-major order
        // hands every row tile of A to all eight L2s.  Banded: XCD x takes the `per` consecutive tiles [x per, (x + 1) per)
        const int per = (nrt * nct + 7) / 8;
        gx = -1; gy = per;
        nblocks = 8 * per;
    }
    // the A planes are laid out for the ALLOCATION batch ([PL][B * T][K]); a ragged call multiplies a prefix of the rows only (Mr = Bact * T)
    const long Ma = std::max<long>(Mr, (long)e->B * e->T);
    GemmPArgs g{reinterpret_cast<const uint4 *>(Ap), reinterpret_cast<const uint4 *>(Wp), Ma * Kd / 8, (long)Nc * Kd / 8, bias, C, Mr, Nc, Kd, ldc, relu,
                (unsigned)((size_t)PL * Ma * Kd * 2), (unsigned)((size_t)PL * Nc * Kd * 2), nrt, nct, gx, gy};
    const dim3 grid(nblocks);
    const size_t lds = (size_t)2 * 1536 * PL * 16;
    if (PL == 1) hipLaunchKernelGGL(k_gemm_p<1>, grid, dim3(512), lds, st, g);
    else if (PL == 2) hipLaunchKernelGGL(k_gemm_p<2>, grid, dim3(512), lds, st, g);
    else hipLaunchKernelGGL(k_gemm_p<3>, grid, dim3(512), lds, st, g);
    HIPCHECK(e, hipGetLastError());
    return 0;
}

int launch_gln(se_engine *e, const float *x, float *y, const float *w, const float *b, long n, int mode, int C, int T,
               int F, hipStream_t st) {
    ProfScope ps(e, "k_gln", "gln", 0, st);
    GlnArgs g{x, y, w, b, n, mode, C, T, F, e->eps_mode};
    hipLaunchKernelGGL(k_gln, dim3(e->B), dim3(1024), 0, st, g);
    HIPCHECK(e, hipGetLastError());
    return 0;
}

int launch_gln_ew(se_engine *e, const float *x, float *y, const float *w, const float *b, const float *slab, int nslot, long n,
                  int mode, int C, int T, int F, hipStream_t st, const float *res = nullptr) {
    ProfScope ps(e, "k_gln_ew", "gln", 0, st);
    GlnEwArgs g{x, y, w, b, SlabStats{slab, nslot, n, e->eps_mode}, mode, C, T, F, res};
    // grid-stride loop: about 8 workgroups per CU over the whole batch (every workgroup first reduces the producer's partial
    // statistics, so thousands of 4-element-per-thread workgroups per launch spent their time there)
    const int want = std::max(1, (8 * e->num_cu + e->B - 1) / e->B);
    if (n % 4 == 0) {
        const int gx = std::min((int)((n / 4 + 255) / 256), want);
        hipLaunchKernelGGL(k_gln_ew<4>, dim3(gx, e->B), dim3(256), 0, st, g);
    } else {
        const int gx = std::min((int)((n + 255) / 256), want);
        hipLaunchKernelGGL(k_gln_ew<1>, dim3(gx, e->B), dim3(256), 0, st, g);
    }
    HIPCHECK(e, hipGetLastError());
    return 0;
}

// TemporalCRN.forward on device in three stages; each reads/writes the ring slot `cur` (history from slot `prev`).
// spec: (b, m, t, f) strides; out: (b, t, f) strides (cf2 units).
// Stage 1: features + (CRN_ELU pre-convs) + encoder  ->  xin[*][cur], gru_in[cur]
// features (CRN.py:463-467) and, for CRN_ELU / the student, the three frequency-dilated pre-conv blocks -> xin[0][cur] (fp32)
int stage_features_pre(se_engine *e, int cur, const cf2 *spec, long sB, long sM, long sT, long sF, hipStream_t st) {
    const int T = e->T, B = e->B;
    int rc;
    e->parity ^= 1;
    const int pcur = e->parity, pprev = pcur ^ 1;
    {
        ProfScope ps(e, "k_featurize", "featurize", 0, st);
        float *dst = e->npre ? e->pin[0][pcur].p : e->xin[0][cur].p;
        FeatArgs f{spec, sB, sM, sT, sF, dst, e->M, T, e->F[0], e->atan2_phase};
        const int TF = T * e->F[0];
        hipLaunchKernelGGL(k_featurize, dim3((TF + 255) / 256, B), dim3(256), 0, st, f);
        HIPCHECK(e, hipGetLastError());
    }
    for (int i = 0; i < e->npre; i++) {  // x = m(x) + x, three frequency-dilated blocks (CRN_ELU.py:375-376)
        const int C0 = e->Ch[0], F0 = e->F[0];
        const long n = (long)C0 * T * F0;
        const int ns = e->lv[i].pre.grid_x;
        if ((rc = launch_conv(e, e->lv[i].pre, e->pin[i][pcur].p, e->pin[i][pprev].p, e->pre_g.p, st, ("pre" + std::to_string(i)).c_str(),
                              e->pre_stats[i].p, ns, 0, 0, C0))) return rc;
        float *dst = i + 1 < e->npre ? e->pin[i + 1][pcur].p : e->xin[0][cur].p;
        if ((rc = launch_gln_ew(e, e->pre_g.p, dst, e->lv[i].pre_nw.p, e->lv[i].pre_nb.p, e->pre_stats[i].p, ns, n, 0, C0, T, F0, st,
                                e->pin[i][pcur].p))) return rc;
    }
    return 0;
}

int stage_encoder(se_engine *e, int cur, int prev, const cf2 *spec, long sB, long sM, long sT, long sF, hipStream_t st) {
    const int L = e->L, T = e->T;
    int rc;
    if ((rc = stage_features_pre(e, cur, spec, sB, sM, sT, sF, st))) return rc;
    for (int i = 0; i < L; i++) {  // encoder (CRN.py:471-474; CRN_ELU.py:233-247)
        const int Co = e->Ch[i + 1], Fo = e->F[i + 1];
        const long n = (long)Co * T * Fo;
        const float *normed_src = e->enc_raw[i].p;
        int ns = e->lv[i].enc.grid_x;
        if (!e->variant) {
            if ((rc = launch_conv(e, e->lv[i].enc, e->xin[i][cur].p, e->xin[i][prev].p, e->enc_raw[i].p, st, ("enc" + std::to_string(i)).c_str(),
                                  e->enc_stats[i].p, ns, 0, 0, Co))) return rc;
        } else {
            if ((rc = launch_conv(e, e->lv[i].enc, e->xin[i][cur].p, e->xin[i][prev].p, e->enc_raw[i].p, st, ("enc" + std::to_string(i)).c_str()))) return rc;
            const int n0 = e->lv[i].gate[0].grid_x, n1 = e->lv[i].gate[1].active ? e->lv[i].gate[1].grid_x : 0;
            ns = n0 + n1;
            if ((rc = launch_conv(e, e->lv[i].gate[0], e->enc_raw[i].p, nullptr, e->enc_g[i].p, st, ("gate" + std::to_string(i)).c_str(),
                                  e->enc_stats[i].p, ns, 0, 0, Co))) return rc;
            if (n1 && (rc = launch_conv(e, e->lv[i].gate[1], e->enc_raw[i].p, nullptr, e->enc_g[i].p, st, ("gate" + std::to_string(i)).c_str(),
                                        e->enc_stats[i].p, ns, n0, 0, Co))) return rc;
            normed_src = e->enc_g[i].p;
        }
        if (i + 1 < L) rc = launch_gln_ew(e, normed_src, e->xin[i + 1][cur].p, e->lv[i].enc_nw.p, e->lv[i].enc_nb.p, e->enc_stats[i].p, ns, n, 0, Co, T, Fo, st);
        else rc = launch_gln_ew(e, normed_src, e->gru_in[cur].p, e->lv[i].enc_nw.p, e->lv[i].enc_nb.p, e->enc_stats[i].p, ns, n, 1, Co, T, Fo, st);
        if (rc) return rc;
    }
    return 0;
}

uint4 *decin_p(se_engine *e, int slot);  // convp_engine.inc.h: decoder-input ring slot of the plane path

// Stage 2: the recurrent bottleneck (CRN.py:476-481, 256-282)  gru_in[cur] -> dec_in[cur]
// The bottleneck is cut where its data dependencies are (CRN.py:256-282):
//   stage_gru_proj0  x W_ih0^T for all T frames: depends only on the encoder output, so it runs on the ENCODER stream
//   stage_gru_layer  layer l: (l > 0: input projection of layer l-1's sequence) + T dependent step launches; each layer has
//                    its own stream in the pipelined path - layer 1 of segment n runs beside layer 0 of segment n+1
//   stage_gru_out    fc_output_layer + activation + gLN(last): depends only on the last layer's sequence: DECODER stream
// so the chain of dependent launches per segment is T steps (+ one GEMM) per stream instead of 2 T + 3 GEMMs on one stream.
// `overlapped`: the stage shares the chip with the other streams.  The GRU step then uses the small-footprint kernel
// (k_gru_step: 54 VGPRs, 6 KB LDS, W_hh straight from L2) whose waves fit next to resident convolution workgroups, instead
// of k_gru_step2 (188 VGPRs, 96 KB LDS slice of W_hh), which is 15 % faster alone but has to wait for a convolution
// workgroup to retire on every CU at every step.
int stage_gru_proj0(se_engine *e, int cur, hipStream_t st) {
    const int T = e->T, B = e->B, H = e->H, D = e->D;
    if (e->dbg_skip & 2) return 0;
    const int Ba = e->Bact;
    (void)B;
    if (e->gemm_p) return launch_gemm_p(e, e->gruinP[cur].p, e->wih_xp.p, e->bih[0].p, e->gi0[cur].p, 3L * H, Ba * T, 3 * H, D, 0, st, "gru_ih0");
    return launch_gemm(e, e->gru_in[cur].p, D, e->wih[0].p, D, e->bih[0].p, e->gi0[cur].p, 3L * H, Ba * T, 3 * H, D, 0, st, "gru_ih0", e->wih_x[0].p);
}

int stage_gru_layer(se_engine *e, int l, int cur, hipStream_t st, bool overlapped) {
    const int T = e->T, B = e->B, H = e->H;
    const int Ba = e->Bact;  // rows that take part (prefix of the batch)
    int rc;
    const float *gi = e->gi0[cur].p;
    if (l > 0) {
        gi = e->gil[l].p;
        if (e->dbg_skip & 2) {}
        else if (e->gemm_p) { if ((rc = launch_gemm_p(e, e->seqP[l - 1][cur].p, e->wih_x[l].p, e->bih[l].p, e->gil[l].p, 3L * H, Ba * T, 3 * H, H, 0, st, ("gru_ih" + std::to_string(l)).c_str()))) return rc; }
        else if ((rc = launch_gemm(e, e->seqr[l - 1][cur].p, H, e->wih[l].p, H, e->bih[l].p, e->gil[l].p, 3L * H, B * T, 3 * H, H, 0, st,
                                   ("gru_ih" + std::to_string(l)).c_str(), e->wih_x[l].p))) return rc;
    }
    float *seq = e->seqr[l][cur].p;
    const int ngroup = (B + 31) / 32, nhid = (H + 15) / 16;
    const bool use_seq = e->gru_seq && e->gru_direct <= 0 && (H == 512 || H == 128) && nhid <= 256;
    if (use_seq) {
        // groups per launch: all workgroups of a launch must be able to become resident (<= 256 CUs, one per CU)
        const int gmax = std::max(1, 256 / nhid);
        for (int g0 = 0; g0 < ngroup; g0 += gmax) {
            const int ng = std::min(gmax, ngroup - g0), rows0 = g0 * 32, rows = std::min(B - rows0, ng * 32);
            const int hc = e->hcur[l];
            HIPCHECK(e, hipMemsetAsync(e->gru_sync.p, 0, 64 * sizeof(unsigned), st));
            GruSeqArgs g{gi + (long)rows0 * T * 3 * H, e->hbuf[l][hc].p + (long)rows0 * H, e->hbuf[l][hc].p + (long)rows0 * H,
                         e->hbuf[l][hc ^ 1].p + (long)rows0 * H, e->whh[l].p, e->bhh[l].p, seq + (long)rows0 * T * H,
                         reinterpret_cast<unsigned *>(e->gru_sync.p), reinterpret_cast<unsigned *>(e->gru_sync.p) + 64, rows, H, T};
            ProfScope ps(e, "k_gru_seq", "gru_seq", 2.0 * rows * 3 * H * H * T, st);
            if (H == 512) hipLaunchKernelGGL(k_gru_seq<16>, dim3(nhid, ng), dim3(256), (size_t)192 * H, st, g);
            else hipLaunchKernelGGL(k_gru_seq<4>, dim3(nhid, ng), dim3(256), (size_t)192 * H, st, g);
        }
        if (T & 1) e->hcur[l] ^= 1;  // the last step (t = T-1) wrote P1 when T is odd, P0 when even
    } else
    for (int t = 0; t < T && !(e->dbg_skip & 1); t++) {
        const int hc = e->hcur[l];
        GruStepArgs g{gi + (long)t * 3 * H, (long)T * 3 * H, e->hbuf[l][hc].p, e->whh[l].p, e->bhh[l].p,
                      e->hbuf[l][hc ^ 1].p, seq + (long)t * H, (long)T * H, Ba, H};
        if (e->gemm_p) {
            g.seqp = reinterpret_cast<__bf16 *>(e->seqP[l][cur].p) + (long)t * H;
            g.seqp_ld = (long)T * H; g.seqp_plane = (long)B * T * H; g.seqp_pl = operand_planes(e->precision);
        }
        ProfScope ps(e, "k_gru_step", "gru_step", 2.0 * B * 3 * H * H, st);
        const dim3 grid((H + 15) / 16, (Ba + 31) / 32);
        // (<= 16 streams: the register-streaming kernel splits K over all four waves and beats the LDS-slice kernel: 9.3 vs 10.4 us)
        const bool direct = e->gru_direct >= 0 ? e->gru_direct != 0 : (overlapped || B <= 16);
        if (H == 512 && !direct) hipLaunchKernelGGL(k_gru_step2<16>, grid, dim3(256), (size_t)192 * H, st, g);
        else if (H == 128 && !direct) hipLaunchKernelGGL(k_gru_step2<4>, grid, dim3(256), (size_t)192 * H, st, g);
        else if (B <= 16) hipLaunchKernelGGL(k_gru_step8, grid, dim3(512), 0, st, g);
        else hipLaunchKernelGGL(k_gru_step, grid, dim3(256), 0, st, g);
        e->hcur[l] = hc ^ 1;
    }
    HIPCHECK(e, hipGetLastError());
    return 0;
}

// Pipelined form of the recurrence: ONE stream, layer l works on the segment in ring slot slots[l] (-1: none).  Layer l lags
// l segments behind layer 0, so the steps of all layers of a round are independent of each other and every time step is ONE
// launch (k_gru_step_multi, blockIdx.z = layer) instead of one per layer: T dependent launches per segment instead of NL * T.
int stage_gru_round(se_engine *e, const int *slots, hipStream_t st) {
    const int T = e->T, B = e->B, H = e->H;
    int rc;
    const float *gi[4] = {nullptr, nullptr, nullptr, nullptr};
    int nact = 0;
    for (int l = 0; l < e->NL; l++) {
        if (slots[l] < 0) continue;
        nact++;
        gi[l] = e->gi0[slots[l]].p;
        if (l > 0) {
            gi[l] = e->gil[l].p;
            if (e->dbg_skip & 2) {}
            else if (e->gemm_p) { if ((rc = launch_gemm_p(e, e->seqP[l - 1][slots[l]].p, e->wih_x[l].p, e->bih[l].p, e->gil[l].p, 3L * H, e->bact_slot[slots[l]] * T, 3 * H, H, 0, st, ("gru_ih" + std::to_string(l)).c_str()))) return rc; }
            else if ((rc = launch_gemm(e, e->seqr[l - 1][slots[l]].p, H, e->wih[l].p, H, e->bih[l].p, e->gil[l].p, 3L * H, B * T, 3 * H, H, 0, st,
                                       ("gru_ih" + std::to_string(l)).c_str(), e->wih_x[l].p))) return rc;
        }
    }
    if (!nact) return 0;
    int bmax = 0;  // every layer's own row count is in its arguments; the grid covers the largest
    for (int l = 0; l < e->NL; l++)
        if (slots[l] >= 0) bmax = std::max(bmax, e->bact_slot[slots[l]]);
    for (int t = 0; t < T && !(e->dbg_skip & 1); t++) {
        GruStepMulti m{};
        int z = 0;
        for (int l = 0; l < e->NL; l++) {
            if (slots[l] < 0) continue;
            const int hc = e->hcur[l];
            GruStepArgs &ga = m.a[z++];
            ga = GruStepArgs{gi[l] + (long)t * 3 * H, (long)T * 3 * H, e->hbuf[l][hc].p, e->whh[l].p, e->bhh[l].p,
                             e->hbuf[l][hc ^ 1].p, e->seqr[l][slots[l]].p + (long)t * H, (long)T * H, e->bact_slot[slots[l]], H};
            if (e->gemm_p) {
                ga.seqp = reinterpret_cast<__bf16 *>(e->seqP[l][slots[l]].p) + (long)t * H;
                ga.seqp_ld = (long)T * H; ga.seqp_plane = (long)B * T * H; ga.seqp_pl = operand_planes(e->precision);
            }
            e->hcur[l] = hc ^ 1;
        }
        ProfScope ps(e, "k_gru_step", "gru_step", 2.0 * B * 3 * H * H * z, st);
        // (the same arithmetic as the single-stream path: batches of <= 16 streams split K over eight waves)
        if (B <= 16) hipLaunchKernelGGL(k_gru_step_multi8, dim3((H + 15) / 16, 1, z), dim3(512), 0, st, m);
        else hipLaunchKernelGGL(k_gru_step_multi, dim3((H + 15) / 16, (bmax + 31) / 32, z), dim3(256), 0, st, m);
    }
    HIPCHECK(e, hipGetLastError());
    return 0;
}

int stage_gru_out(se_engine *e, int cur, hipStream_t st) {
    const int L = e->L, T = e->T, B = e->B, H = e->H, D = e->D;
    const int Ba = e->Bact;
    (void)B;
    int rc;
    if (e->dbg_skip & 2) {}
    else if (e->gemm_p) { if ((rc = launch_gemm_p(e, e->seqP[e->NL - 1][cur].p, e->fcw_x.p, e->fcb.p, e->fc_out.p, D, Ba * T, D, H, e->act, st, "gru_fc"))) return rc; }
    else if ((rc = launch_gemm(e, e->seqr[e->NL - 1][cur].p, H, e->fcw.p, H, e->fcb.p, e->fc_out.p, D, Ba * T, D, H, e->act, st, "gru_fc", e->fcw_x.p))) return rc;
    if (e->use_p) {  // decoder input in the plane layout
        const int PL = operand_planes(e->precision), C = e->Ch[L], C8 = (C + 7) / 8;
        ProfScope ps(e, "k_gln2_p", "gln", 0, st);
        Gln2PArgs g{e->fc_out.p, e->gnw.p, e->gnb.p, decin_p(e, cur), (long)C8 * PL * T * e->F[L], T, e->F[L], C, C8, e->eps_mode};
        launch_k_gln2_p(PL, dim3(Ba), st, g);
        HIPCHECK(e, hipGetLastError());
        return 0;
    }
    return launch_gln(e, e->fc_out.p, e->dec_in[cur].p, e->gnw.p, e->gnb.p, (long)T * D, 2, e->Ch[L], T, e->F[L], st);
}

// single-stream form: all bottleneck stages back to back
int stage_bottleneck(se_engine *e, int cur, hipStream_t st, bool overlapped = false) {
    int rc;
    if ((rc = stage_gru_proj0(e, cur, st))) return rc;
    for (int l = 0; l < e->NL; l++)
        if ((rc = stage_gru_layer(e, l, cur, st, overlapped))) return rc;
    return stage_gru_out(e, cur, st);
}

// Stage 3: decoder (CRN.py:483-489) + mask application  dec_in[cur], xin[*][cur], spec -> out
int stage_decoder(se_engine *e, int cur, const cf2 *spec, long sB, long sT, long sF, cf2 *out, long oB, long oT, long oF, hipStream_t st) {
    const int L = e->L, T = e->T, B = e->B;
    int rc;
    const float *x = e->dec_in[cur].p;
    for (int j = 0; j < L; j++) {
        const int lvl = L - 1 - j;
        const int Co = lvl == 0 ? 2 : e->Ch[lvl], Fi = e->F[lvl + 1], Fo = 2 * Fi - 1;
        const int ne = e->lv[j].dec_even.active ? e->lv[j].dec_even.grid_x : 0, no = e->lv[j].dec_odd.active ? e->lv[j].dec_odd.grid_x : 0;
        if ((rc = launch_conv(e, e->lv[j].dec_even, x, nullptr, e->dec_raw[j].p, st, ("dec" + std::to_string(j) + "_even").c_str(),
                              e->dec_stats[j].p, ne + no, 0, 0, Co))) return rc;
        if ((rc = launch_conv(e, e->lv[j].dec_odd, x, nullptr, e->dec_raw[j].p, st, ("dec" + std::to_string(j) + "_odd").c_str(),
                              e->dec_stats[j].p, ne + no, ne, 0, Co))) return rc;
        const SlabStats sy{e->dec_stats[j].p, ne + no, (long)Co * T * Fo, e->eps_mode};
        if (lvl > 0) {
            const int Fr = e->F[lvl];
            const long nu = (long)Co * T * Fr;
            if (e->lv[j].skip_fused) {
                const int nm = e->lv[j].skipm.grid_x;
                if ((rc = launch_conv(e, e->lv[j].skipm, e->xin[lvl][cur].p, nullptr, nullptr, st, ("skipstat" + std::to_string(j)).c_str(),
                                      e->skip_stats[j].p, nm, 0, 0, Co))) return rc;
                ConvBlend bl{e->dec_raw[j].p, e->lv[j].dec_nw.p, e->lv[j].dec_nb.p, e->lv[j].dec_mnw.p, e->lv[j].dec_mnb.p, sy,
                             SlabStats{e->skip_stats[j].p, nm, nu, e->eps_mode}, Fo};
                if ((rc = launch_conv(e, e->lv[j].skip, e->xin[lvl][cur].p, nullptr, e->dec_out[j].p, st, ("skip" + std::to_string(j)).c_str(),
                                      nullptr, 0, 0, 0, 0, &bl))) return rc;
            } else {
                const int nk = e->lv[j].skip.grid_x;
                if ((rc = launch_conv(e, e->lv[j].skip, e->xin[lvl][cur].p, nullptr, e->dec_uv[j].p, st, ("skip" + std::to_string(j)).c_str(),
                                      e->skip_stats[j].p, nk, 0, 0, Co))) return rc;
                if (nu % 4) return fail(e, SE_ERR_ARG, "decoder tensor size %ld not a multiple of 4", nu);
                BlendEwArgs bl{e->dec_raw[j].p, e->dec_uv[j].p, e->dec_out[j].p, e->lv[j].dec_nw.p, e->lv[j].dec_nb.p,
                               e->lv[j].dec_mnw.p, e->lv[j].dec_mnb.p, sy, SlabStats{e->skip_stats[j].p, nk, nu, e->eps_mode}, Co, T, Fo, Fr};
                ProfScope ps(e, "k_dec_blend_ew", "dec_blend", 0, st);
                hipLaunchKernelGGL(k_dec_blend_ew, dim3((unsigned)((nu / 4 + 1023) / 1024), B), dim3(256), 0, st, bl);
                HIPCHECK(e, hipGetLastError());
            }
            x = e->dec_out[j].p;
        } else {
            if (Fo != e->F[0]) return fail(e, SE_ERR_ARG, "decoder output has %d bins, spectrum has %d", Fo, e->F[0]);
            MaskEwArgs m{e->dec_raw[j].p, e->lv[j].dec_nw.p, e->lv[j].dec_nb.p, sy, spec, sB, sT, sF, out, oB, oT, oF, T, e->F[0]};
            ProfScope ps(e, "k_final_mask_ew", "final_mask", 0, st);
            launch_k_final_mask_ew(dim3((T * e->F[0] + 1023) / 1024, B), st, m);
            HIPCHECK(e, hipGetLastError());
        }
    }
    return 0;
}

}  // namespace

#include "convp_engine.inc.h"

namespace {

int run_encoder(se_engine *e, int cur, int prev, const cf2 *spec, long sB, long sM, long sT, long sF, hipStream_t st) {
    if (e->dbg_skip & 4) return 0;
    return e->use_p ? stage_encoder_p(e, cur, prev, spec, sB, sM, sT, sF, st) : stage_encoder(e, cur, prev, spec, sB, sM, sT, sF, st);
}
int run_decoder(se_engine *e, int cur, const cf2 *spec, long sB, long sT, long sF, cf2 *out, long oB, long oT, long oF, hipStream_t st) {
    if (e->dbg_skip & 8) return 0;
    return e->use_p ? stage_decoder_p(e, cur, spec, sB, sT, sF, out, oB, oT, oF, st) : stage_decoder(e, cur, spec, sB, sT, sF, out, oB, oT, oF, st);
}

int forward_dev(se_engine *e, const cf2 *spec, long sB, long sM, long sT, long sF, cf2 *out, long oB, long oT, long oF,
                hipStream_t st) {
    const int prev = e->slot, cur = (e->slot + 1) % kRing;
    e->slot = cur;
    int rc;
    if ((rc = run_encoder(e, cur, prev, spec, sB, sM, sT, sF, st))) return rc;
    if ((rc = stage_bottleneck(e, cur, st))) return rc;
    return run_decoder(e, cur, spec, sB, sT, sF, out, oB, oT, oF, st);
}

// nseg > 1: one launch for nseg consecutive segments (segment y reads off + y*seg_off, writes spec + y*seg_spec)
int launch_stft(se_engine *e, const float *src, long strideB, long strideM, int M, long off, long Lsrc, int rows,
                cf2 *spec, long sR, long sT, long sF, hipStream_t st, int nseg = 1, long seg_off = 0, long seg_spec = 0) {
    StftArgs a{};
    a.seg_off = seg_off; a.seg_spec = seg_spec;
    a.src = src; a.strideB = strideB; a.strideM = strideM; a.M = M; a.off = off; a.L = Lsrc;
    a.K = e->K; a.T = e->T; a.F = e->F[0]; a.hop = e->c.hop;
    a.spec = spec; a.sR = sR; a.sT = sT; a.sF = sF;
    a.window = e->window.p; a.tw = reinterpret_cast<const cf2 *>(e->tw.p); a.plan = e->plan;
    a.Lrow = e->ragged_on ? reinterpret_cast<const long *>(e->ragged_len.p) : nullptr;
    ProfScope ps(e, "k_stft", "stft", 0, st);
    launch_k_stft(dim3(rows, nseg), stft_lds_bytes(e->K, e->N), st, a);
    HIPCHECK(e, hipGetLastError());
    return 0;
}

int launch_istft(se_engine *e, const cf2 *spec, long sR, long sT, long sF, int rows, float *wav, long wav_ld, hipStream_t st,
                 int nseg = 1, long seg_spec = 0, long seg_wav = 0) {
    IstftArgs a{};
    a.seg_spec = seg_spec; a.seg_wav = seg_wav;
    a.spec = spec; a.sR = sR; a.sT = sT; a.sF = sF; a.K = e->K; a.T = e->T; a.F = e->F[0]; a.hop = e->c.hop;
    a.wav = wav; a.wav_ld = wav_ld; a.window = e->window.p; a.env = e->env.p; a.tw = reinterpret_cast<const cf2 *>(e->tw.p); a.plan = e->plan;
    ProfScope ps(e, "k_istft", "istft", 0, st);
    launch_k_istft(dim3(rows, nseg), istft_lds_bytes(e->T, e->N), st, a);
    HIPCHECK(e, hipGetLastError());
    return 0;
}

// The one place that decides the bottleneck GEMM route of the current batch: plane GEMMs (k_gemm_p) when the engine can run
// them AND the batch has more rows than the skinny fp32 kernel wins at.  Called after every re-plan and from se_reset.
void select_gemm_route(se_engine *e) { e->gemm_p = e->gemm_p_cap && e->B > 0 && (long)e->B * e->T > e->skinny_rows; }

int ensure_ready(se_engine *e) {
    if (!e) return SE_ERR_ARG;
    HIPCHECK(e, hipSetDevice(e->device));
    const bool replanned = !e->weights_ready;
    int rc = prepare_weights(e);
    if (rc) return rc;
    if (replanned) {
        e->use_p = e->path != 0 && convp_supported(e);
        if (e->use_p && (rc = prepare_weights_p(e))) return rc;
        // capability only: whether THIS batch takes the plane GEMMs is select_gemm_route()'s decision (a batch on the skinny route
        // never allocated gruinP / seqP, so a weight reload must not switch the plane route back on behind its back)
        e->gemm_p_cap = e->use_p && e->gemm_p_env && !e->gru_seq && e->D % 32 == 0 && e->H % 32 == 0 && e->Ch[e->L] % 8 == 0;
        select_gemm_route(e);
        if (e->gemm_p_cap) {  // W_ih0 with its K axis in the engine's feature order k' = (o * F + f) * 8 + c  (reference d = (8 o + c) * F + f)
            const int Fl = e->F[e->L], D = e->D, H = e->H;
            const std::vector<float> &w = e->params["gru.sequence_model.weight_ih_l0"];
            std::vector<float> wp((size_t)3 * H * D);
            for (int r = 0; r < 3 * H; r++)
                for (int d = 0; d < D; d++) {
                    const int cabs = d / Fl, f = d - cabs * Fl;
                    wp[(size_t)r * D + ((size_t)(cabs >> 3) * Fl + f) * 8 + (cabs & 7)] = w[(size_t)r * D + d];
                }
            if ((rc = upload_split3(e, e->wih_xp, wp))) return rc;
        }
        if (e->use_p && e->B > 0) select_all_p(e);
    }
    if (replanned && e->B > 0)  // new weights re-made the plans with their default tiling: restore the per-batch choice
        for (int i = 0; i < SE_MAX_LEVELS; i++)
            for (ConvPlan *p : {&e->lv[i].enc, &e->lv[i].dec_even, &e->lv[i].dec_odd, &e->lv[i].skip, &e->lv[i].skipm, &e->lv[i].gate[0], &e->lv[i].gate[1], &e->lv[i].pre})
                select_conv_geometry(e, *p);
    return 0;
}

}  // namespace

namespace {
// P layout [b][octet][plane][t][f] (16-byte pieces of 8 channels) -> fp32 [b][c][t][f]: the operands of a feature-tap re-run
template <int PL>
__global__ void k_tap_p_to_f32(const uint4 *P, float *dst, int B, int C, int T, int F) {
    const int C8 = (C + 7) >> 3;
    const long n = (long)B * C8 * T * F;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long tf = i % ((long)T * F), bo = i / ((long)T * F);
        const int o = (int)(bo % C8), b = (int)(bo / C8);
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int pl = 0; pl < PL; pl++) {
            const uint4 q = P[(((long)b * C8 + o) * PL + pl) * T * F + tf];
            if (PL == 1) {
                typedef _Float16 h8 __attribute__((ext_vector_type(8)));
                const h8 h = __builtin_bit_cast(h8, q);
#pragma unroll
                for (int c = 0; c < 8; c++) v[c] += (float)h[c];
            } else {
                const bf16x8 h = __builtin_bit_cast(bf16x8, q);
#pragma unroll
                for (int c = 0; c < 8; c++) v[c] += (float)h[c];
            }
        }
#pragma unroll
        for (int c = 0; c < 8; c++)
            if (o * 8 + c < C) dst[((long)b * C + o * 8 + c) * T * F + tf] = v[c];
    }
}
// [b][c][t][f] -> [b][c][f][t] (the reference's feature-map layout, distillation_crn.py:467-477)
__global__ void k_tap_tf_to_ft(const float *src, float *dst, long BC, int T, int F) {
    const long n = BC * T * F;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int t = (int)(i % T);
        const long r = i / T;
        const int f = (int)(r % F);
        const long bc = r / F;
        dst[i] = src[(bc * T + t) * F + f];
    }
}
int tap_p_to_f32_dev(se_engine *e, const float *psrc, int C, int F, float *dst, hipStream_t st) {
    const int PL = operand_planes(e->precision);
    const dim3 g(1024), b(256);
    const uint4 *P = reinterpret_cast<const uint4 *>(psrc);
    if (PL == 1) hipLaunchKernelGGL(k_tap_p_to_f32<1>, g, b, 0, st, P, dst, e->B, C, e->T, F);
    else if (PL == 2) hipLaunchKernelGGL(k_tap_p_to_f32<2>, g, b, 0, st, P, dst, e->B, C, e->T, F);
    else hipLaunchKernelGGL(k_tap_p_to_f32<3>, g, b, 0, st, P, dst, e->B, C, e->T, F);
    HIPCHECK(e, hipGetLastError());
    return 0;
}
}  // namespace

extern "C" {

int se_abi_version(void) { return 4; }

int se_config_size(void) { return (int)sizeof(se_config); }
int fsn_config_size(void) { return (int)sizeof(fsn_config); }

const char *se_last_error(const se_engine *e) { return e ? e->err.c_str() : g_create_error.c_str(); }

int se_create(const se_config *cfg, int device, se_engine **out) {
    if (!cfg || !out) return fail(nullptr, SE_ERR_ARG, "null argument");
    *out = nullptr;
    const int L = cfg->num_levels;
    if (L < 1 || L > SE_MAX_LEVELS) return fail(nullptr, SE_ERR_ARG, "num_levels %d out of range", L);
    if (cfg->kernel_size != 3) return fail(nullptr, SE_ERR_ARG, "kernel_size %d unsupported (reference config uses 3)", cfg->kernel_size);
    if (cfg->num_layers < 1 || cfg->num_layers > 4) return fail(nullptr, SE_ERR_ARG, "num_layers %d out of range", cfg->num_layers);
    if (cfg->hidden <= 0 || cfg->hidden % 16) return fail(nullptr, SE_ERR_ARG, "hidden must be a positive multiple of 16 (k_gru_step walks 16-deep k blocks)");
    if (cfg->n_fft % 2 || cfg->num_freqs != cfg->n_fft / 2 + 1) return fail(nullptr, SE_ERR_ARG, "num_freqs must be n_fft/2+1 (CRN.py:511)");
    if (cfg->win > cfg->n_fft || cfg->hop <= 0 || cfg->segment_length % cfg->hop) return fail(nullptr, SE_ERR_ARG, "bad STFT geometry");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, SE_ERR_HIP, "no HIP device available: the engine has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(nullptr, SE_ERR_ARG, "device %d out of range (%d visible)", device, ndev);
    se_engine *e = new se_engine();
    e->c = *cfg;
    e->device = device;
    e->L = L; e->M = cfg->num_inputs; e->K = cfg->segment_length; e->N = cfg->n_fft; e->H = cfg->hidden; e->NL = cfg->num_layers;
    if (cfg->variant < 0 || cfg->variant > 2) { delete e; return fail(nullptr, SE_ERR_ARG, "variant %d unknown (0 CRN, 1 CRN_ELU, 2 student)", cfg->variant); }
    if (cfg->precision < 0 || cfg->precision > 2) { delete e; return fail(nullptr, SE_ERR_ARG, "precision %d unknown (0 fp32-accurate, 1 fp16 operands, 2 bf16x3)", cfg->precision); }
    e->precision = cfg->precision;
    e->variant = cfg->variant;
    e->act = cfg->variant ? 2 : 1;
    e->npre = cfg->variant ? 3 : 0;
    e->atan2_phase = cfg->variant == 1;
    e->eps_mode = cfg->variant == 2;
    e->T = 1 + cfg->segment_length / cfg->hop;
    e->F[0] = cfg->num_freqs;
    e->Ch[0] = 2 * cfg->num_inputs - 1;
    for (int i = 0; i < L; i++) { e->F[i + 1] = (e->F[i] - 1) / 2 + 1; e->Ch[i + 1] = cfg->channels[i]; }
    e->D = (cfg->num_freqs / 16 + 1) * cfg->channels[L - 1];
    auto bail = [&](int code, const char *msg) { g_create_error = msg; delete e; return code; };
    if (e->D != e->F[L] * cfg->channels[L - 1]) return bail(SE_ERR_ARG, "num_freqs/num_levels combination breaks the reference reshape (CRN.py:450,478)");
    if (e->T <= 2 * (1 << (L - 1))) return bail(SE_ERR_ARG, "segment shorter than the largest dilation history (CRN.py:333)");
    e->plan.N = cfg->n_fft;
    e->plan.npass = fft_plan(cfg->n_fft / 2, e->plan.radices);
    if (!e->plan.npass || e->plan.npass > kMaxRadices) return bail(SE_ERR_ARG, "n_fft must factor into 2s and 5s");
    if (const char *s = getenv("SE_CONV_LDS_KB")) e->conv_lds_budget = (size_t)atoi(s) * 1024;
    if (const char *s = getenv("SE_GRU_DIRECT")) e->gru_direct = atoi(s) != 0 ? 1 : 0;
    if (const char *s = getenv("SE_GRU_SEQ")) e->gru_seq = atoi(s);
    if (const char *s = getenv("SE_GEMM_MODE")) e->gemm_mode = atoi(s);
    if (const char *s = getenv("SE_CONV_MODE")) e->conv_mode = atoi(s);
    if (const char *s = getenv("SE_CONV_GEO_FIXED")) e->conv_geo_fixed = atoi(s);
    if (const char *s = getenv("SE_PIPELINE")) e->pipeline = atoi(s);
    if (const char *s = getenv("SE_CONV_SMALL16")) e->conv_small16 = atoi(s);
    if (const char *s = getenv("SE_SKIP_FUSE")) e->skip_fuse = atoi(s);
    if (const char *s = getenv("SE_DEC_MERGE")) e->dec_merge = atoi(s);
    if (const char *s = getenv("SE_PATH")) e->path = atoi(s);
    if (const char *s = getenv("SE_SKIP_STREAM")) e->skip_stream = atoi(s);
#ifdef SE_DEBUG_KNOBS  // timing-experiment build only (profiles/pipe_split.sh): drops whole stages, the audio is WRONG
    if (const char *s = getenv("SE_DBG_SKIP")) { e->dbg_skip = atoi(s); fprintf(stderr, "se_engine: SE_DBG_SKIP=%d - stages are skipped, results are WRONG (timing experiment)\n", e->dbg_skip); }
#else
    if (getenv("SE_DBG_SKIP")) return bail(SE_ERR_ARG, "SE_DBG_SKIP is set but this library was built without -DSE_DEBUG_KNOBS: refusing to run (the knob drops whole stages and produces wrong audio)");
#endif
    if (const char *s = getenv("SE_GRU_LAG")) e->gru_lag = atoi(s);
    if (const char *s = getenv("SE_GEMM_P")) e->gemm_p_env = atoi(s);
    if (const char *s = getenv("SE_GEMM_BAND")) e->gemm_band = atoi(s);
    if (const char *s = getenv("SE_CONVP_DEINT")) e->convp_deint = atoi(s);
    if (const char *s = getenv("SE_GEMM_SKINNY_ROWS")) e->skinny_rows = atoi(s);
    if (const char *s = getenv("SE_SKIP_MIN_BATCH")) e->skip_min_batch = atoi(s);
    e->cp = new se_convp_state();
    {
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0) e->num_cu = ncu;
    }
    if (hipSetDevice(device) != hipSuccess) return bail(SE_ERR_HIP, "hipSetDevice failed");
    // tables: hamming(win) centred in n_fft (torch.stft), twiddles, overlap-add envelope
    const int N = e->N, T = e->T, hop = cfg->hop, K = e->K;
    std::vector<float> win(N, 0.0f), tw(2 * (size_t)N), env(K, 0.0f);
    const int left = (N - cfg->win) / 2;
    for (int i = 0; i < cfg->win; i++) win[left + i] = (float)(0.54 - 0.46 * cos(2.0 * M_PI * i / cfg->win));
    for (int i = 0; i < N; i++) { tw[2 * i] = (float)cos(2.0 * M_PI * i / N); tw[2 * i + 1] = (float)-sin(2.0 * M_PI * i / N); }
    for (int i = 0; i < K; i++) {
        const int pos = N / 2 + i;
        float s = 0;
        for (int t = 0; t < T; t++) { const int n = pos - t * hop; if (n >= 0 && n < N) s += win[n] * win[n]; }
        env[i] = s;
    }
    int rc;
    if ((rc = dev_upload(e, e->window, win)) || (rc = dev_upload(e, e->tw, tw)) || (rc = dev_upload(e, e->env, env))) {
        g_create_error = e->err; delete e; return rc;
    }
    // opt in to large dynamic LDS for the FFT kernels
    aux_set_fft_lds((int)stft_lds_bytes(K, N), (int)istft_lds_bytes(T, N));
    conv_set_attributes();
    conv_p_set_attributes();
    skip_p_set_attributes();
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_p<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 1536 * 1 * 16);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_p<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 1536 * 2 * 16);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_p<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 1536 * 3 * 16);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gru_step2<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 192 * 512);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gru_seq<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 192 * 512);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gru_seq<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 192 * 128);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gru_step2<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 192 * 128);
    *out = e;
    return SE_OK;
}

void se_destroy(se_engine *e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipDeviceSynchronize();
    conv_x6_trace_dump();
#ifdef SE_CP_TRACE
    g_cp_trace_sites.dump();
#endif
    DevBuf *singles[] = {&e->window, &e->env, &e->tw, &e->fcw, &e->fcb, &e->gnw, &e->gnb, &e->maskspec,
                         &e->gru_sync, &e->fcw_x, &e->wih_xp, &e->pre_g, &e->spec_all, &e->mask_all, &e->fc_out, &e->yseg};
    for (DevBuf *b : singles) dev_free(*b);
    for (int r = 0; r < kRing; r++) {
        dev_free(e->spec[r]); dev_free(e->gru_in[r]); dev_free(e->dec_in[r]); dev_free(e->gi0[r]); dev_free(e->gruinP[r]);
        for (int l = 0; l < 4; l++) dev_free(e->seqP[l][r]);
        for (int l = 0; l < 4; l++) dev_free(e->seqr[l][r]);
        for (int i = 0; i < SE_MAX_LEVELS; i++) dev_free(e->xin[i][r]);
        if (e->stage_ready) { (void)hipEventDestroy(e->ev_enc[r]); (void)hipEventDestroy(e->ev_dec[r]); (void)hipEventDestroy(e->ev_gru[r]); }
    }
    if (e->stage_ready) {
        (void)hipEventDestroy(e->ev_fork); (void)hipEventDestroy(e->ev_join);
        for (hipStream_t q : e->stage_stream) (void)hipStreamDestroy(q);
    }
    for (int i = 0; i < 4; i++) {
        dev_free(e->wih[i]); dev_free(e->whh[i]); dev_free(e->bih[i]); dev_free(e->bhh[i]); dev_free(e->wih_x[i]);
        dev_free(e->hbuf[i][0]); dev_free(e->hbuf[i][1]); dev_free(e->gil[i]);
    }
    for (int i = 0; i < SE_MAX_LEVELS; i++) {
        Level &l = e->lv[i];
        for (ConvPlan *p : {&l.enc, &l.dec_even, &l.dec_odd, &l.skip, &l.skipm}) { dev_free(p->w); dev_free(p->bias); dev_free(p->wx); }
        for (DevBuf *b : {&l.enc_nw, &l.enc_nb, &l.dec_nw, &l.dec_nb, &l.dec_mnw, &l.dec_mnb}) dev_free(*b);
        dev_free(e->enc_raw[i]);
        dev_free(e->dec_raw[i]); dev_free(e->dec_uv[i]); dev_free(e->dec_out[i]);
        dev_free(e->enc_stats[i]); dev_free(e->dec_stats[i]); dev_free(e->skip_stats[i]); dev_free(e->enc_g[i]);
        for (ConvPlan *p : {&l.gate[0], &l.gate[1], &l.pre}) { dev_free(p->w); dev_free(p->bias); dev_free(p->wx); dev_free(p->gatew); }
        dev_free(l.pre_nw); dev_free(l.pre_nb);
        if (i < 3) { dev_free(e->pin[i][0]); dev_free(e->pin[i][1]); dev_free(e->pre_stats[i]); }
    }
    free_state_p(e);
    delete e;
}

int se_load_param(se_engine *e, const char *key, const float *host_data, const int64_t *shape, int ndim) {
    if (!e || !key || !host_data) return fail(e, SE_ERR_ARG, "null argument");
    const std::string k = canon(key);
    // accept exactly the reference's key set
    int idx = -1, n = 0;
    char rest[64];
    bool ok = false;
    auto gated_key = [&](const char *r) {
        return e->variant && (!strcmp(r, "conv_trans.weight") || !strcmp(r, "conv_trans.bias") || !strcmp(r, "conv_gated.weight") || !strcmp(r, "conv_gated.bias"));
    };
    if (sscanf(k.c_str(), "preconvlist.%d.%63s", &idx, rest) == 2)
        ok = e->variant && idx >= 0 && idx < 3 && (!strcmp(rest, "conv.weight") || !strcmp(rest, "conv.bias") || !strcmp(rest, "norm.weight") || !strcmp(rest, "norm.bias") || gated_key(rest));
    else if (sscanf(k.c_str(), "convlist.%d.%63s", &idx, rest) == 2)
        ok = idx >= 0 && idx < e->L && (!strcmp(rest, "conv.weight") || !strcmp(rest, "conv.bias") || !strcmp(rest, "norm.weight") || !strcmp(rest, "norm.bias") || gated_key(rest));
    else if (sscanf(k.c_str(), "deconvlist.%d.%63s", &idx, rest) == 2) {
        static const char *names[] = {"conv.weight", "conv.bias", "norm.weight", "norm.bias", "residualmask.weight", "residualmask.bias",
                                      "residualnorm.weight", "residualnorm.bias", "residual.weight", "residual.bias"};
        for (const char *nm : names) ok = ok || !strcmp(rest, nm);
        ok = ok && idx >= 0 && idx < e->L;
    } else if (sscanf(k.c_str(), "gru.sequence_model.%63[a-z_]%d%n", rest, &idx, &n) == 2)
        ok = idx >= 0 && idx < e->NL && (size_t)n == k.size() &&
             (!strcmp(rest, "weight_ih_l") || !strcmp(rest, "weight_hh_l") || !strcmp(rest, "bias_ih_l") || !strcmp(rest, "bias_hh_l"));
    else
        ok = k == "gru.fc_output_layer.weight" || k == "gru.fc_output_layer.bias" || k == "gru.norm.weight" || k == "gru.norm.bias";
    if (!ok) return fail(e, SE_ERR_KEY, "unknown parameter key %s", key);
    size_t cnt = 1;
    std::vector<int64_t> shp;
    for (int i = 0; i < ndim; i++) { if (shape[i] < 0) return fail(e, SE_ERR_SHAPE, "negative dimension in %s", key); cnt *= (size_t)shape[i]; shp.push_back(shape[i]); }
    e->params[k].assign(host_data, host_data + cnt);
    e->shapes[k] = shp;
    e->weights_ready = false;
    return SE_OK;
}

// (Re)allocates state for `batch` streams if needed and zeroes it asynchronously on `st`.
static int reset_on_stream(se_engine *e, int batch, hipStream_t st) {
    if (!e || batch <= 0) return fail(e, SE_ERR_ARG, "batch must be positive");
    int rc = ensure_ready(e);
    if (rc) return rc;
    const int L = e->L, T = e->T, B = batch, H = e->H, D = e->D, F0 = e->F[0];
    e->B = B;
    e->Bact = B;
    for (int &v : e->bact_slot) v = B;
    // a few streams give the bottleneck GEMMs a few hundred rows: a 256 x 128 tile per workgroup leaves 12-36 workgroups
    // walking K = 2048 alone (124 us at B = 1); the skinny 32 x 32-tile fp32 kernel (K split over the waves) takes 24
    select_gemm_route(e);
    for (int i = 0; i < SE_MAX_LEVELS; i++)
        for (ConvPlan *p : {&e->lv[i].enc, &e->lv[i].dec_even, &e->lv[i].dec_odd, &e->lv[i].skip, &e->lv[i].skipm, &e->lv[i].gate[0], &e->lv[i].gate[1], &e->lv[i].pre})
            select_conv_geometry(e, *p);
    size_t spec_n = (size_t)B * e->M * T * F0 * 2;
    if ((rc = dev_alloc(e, e->maskspec, (size_t)B * T * F0 * 2))) return rc;
    for (int r = 0; r < kRing; r++)
        if ((rc = dev_alloc(e, e->spec[r], spec_n)) || (rc = dev_alloc(e, e->gru_in[r], (size_t)B * T * D)) || (rc = dev_alloc(e, e->dec_in[r], (size_t)B * T * D))) return rc;
    for (int i = 0; i < L; i++) {
        const size_t nin = (size_t)B * e->Ch[i] * T * e->F[i];
        for (int r = 0; r < kRing; r++)
            if ((rc = dev_alloc(e, e->xin[i][r], nin))) return rc;
        HIPCHECK(e, hipMemsetAsync(e->xin[i][0].p, 0, nin * sizeof(float), st));  // slot 0 = the all-zero history of the first segment
        if ((rc = dev_alloc(e, e->enc_raw[i], (size_t)B * e->Ch[i + 1] * T * e->F[i + 1]))) return rc;
        if ((rc = dev_alloc(e, e->enc_stats[i], (size_t)B * 2 * (e->lv[i].enc.grid_x + e->lv[i].gate[0].grid_x + e->lv[i].gate[1].grid_x + 1)))) return rc;
        if (e->variant && (rc = dev_alloc(e, e->enc_g[i], (size_t)B * e->Ch[i + 1] * T * e->F[i + 1]))) return rc;
        if ((rc = dev_alloc(e, e->dec_stats[i], (size_t)B * 2 * (e->lv[i].dec_even.grid_x + e->lv[i].dec_odd.grid_x + 1)))) return rc;
        if ((rc = dev_alloc(e, e->skip_stats[i], (size_t)B * 2 * (std::max(e->lv[i].skip.grid_x, e->lv[i].skipm.grid_x) + 1)))) return rc;
        const int lvl = L - 1 - i;  // decoder index i
        const int Co = lvl == 0 ? 2 : e->Ch[lvl], Fo = 2 * e->F[lvl + 1] - 1, Fr = e->F[lvl];
        if ((rc = dev_alloc(e, e->dec_raw[i], (size_t)B * Co * T * Fo))) return rc;
        if (lvl > 0) {
            if ((rc = dev_alloc(e, e->dec_uv[i], (size_t)B * 2 * Co * T * Fr))) return rc;
            if ((rc = dev_alloc(e, e->dec_out[i], (size_t)B * Co * T * Fr))) return rc;
        }
    }
    for (int i = 0; i < e->npre; i++) {
        const size_t nf = (size_t)B * e->Ch[0] * T * F0;
        for (int p = 0; p < 2; p++) {
            if ((rc = dev_alloc(e, e->pin[i][p], nf))) return rc;
            HIPCHECK(e, hipMemsetAsync(e->pin[i][p].p, 0, nf * sizeof(float), st));
        }
        if ((rc = dev_alloc(e, e->pre_stats[i], (size_t)B * 2 * (e->lv[i].pre.grid_x + 1)))) return rc;
        if ((rc = dev_alloc(e, e->pre_g, nf))) return rc;
    }
    if ((rc = dev_alloc(e, e->fc_out, (size_t)B * T * D))) return rc;
    for (int r = 0; r < kRing; r++) {
        if ((rc = dev_alloc(e, e->gi0[r], (size_t)B * T * 3 * H))) return rc;
        for (int l = 0; l < e->NL; l++)
            if ((rc = dev_alloc(e, e->seqr[l][r], (size_t)B * T * H))) return rc;
    }
    for (int l = 1; l < e->NL; l++)
        if ((rc = dev_alloc(e, e->gil[l], (size_t)B * T * 3 * H))) return rc;
    if (e->gemm_p) {
        const size_t PLn = operand_planes(e->precision);
        for (int r = 0; r < kRing; r++) {
            if ((rc = dev_alloc(e, e->gruinP[r], (PLn * B * T * D + 1) / 2))) return rc;
            for (int l = 0; l < e->NL; l++)
                if ((rc = dev_alloc(e, e->seqP[l][r], (PLn * B * T * H + 1) / 2))) return rc;
        }
    }
    for (int l = 0; l < e->NL; l++)
        for (int p = 0; p < 2; p++) {
            if ((rc = dev_alloc(e, e->hbuf[l][p], (size_t)B * H))) return rc;
            HIPCHECK(e, hipMemsetAsync(e->hbuf[l][p].p, 0, (size_t)B * H * sizeof(float), st));
        }
    if (e->use_p) {
        select_all_p(e);
        if ((rc = alloc_state_p(e, st))) return rc;
    }
    if ((rc = dev_alloc(e, e->gru_sync, 128))) return rc;
    HIPCHECK(e, hipMemsetAsync(e->gru_sync.p, 0, 128 * sizeof(float), st));
    for (int l = 0; l < 4; l++) e->hcur[l] = 0;
    e->parity = 0;
    e->slot = 0;
    return SE_OK;
}

int se_reset(se_engine *e, int batch) {
    int rc = reset_on_stream(e, batch, nullptr);
    if (rc) return rc;
    HIPCHECK(e, hipDeviceSynchronize());
    return SE_OK;
}

int se_reset_stream(se_engine *e, int stream_index, void *stream) {
    if (!e) return SE_ERR_ARG;
    if (e->B <= 0) return fail(e, SE_ERR_STATE, "se_reset_stream before se_reset");
    if (stream_index < 0 || stream_index >= e->B) return fail(e, SE_ERR_ARG, "stream index %d outside the batch of %d", stream_index, e->B);
    HIPCHECK(e, hipSetDevice(e->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int T = e->T, H = e->H, b = stream_index;
    // conv time buffers = the tail of the current ring slot (the next window reads it as history)
    for (int i = 0; i < e->L; i++) {
        if (e->use_p) {  // plane path: [slot][b][C8][PL][T][F] pieces of 16 bytes
            const size_t per16 = (size_t)(e->cp->slot_elems[i] / e->B);
            HIPCHECK(e, hipMemsetAsync(e->cp->xinP[i].p + ((size_t)e->slot * e->cp->slot_elems[i] + per16 * b) * 4, 0, per16 * 16, st));
            continue;
        }
        const size_t per = (size_t)e->Ch[i] * T * e->F[i];
        HIPCHECK(e, hipMemsetAsync(e->xin[i][e->slot].p + per * b, 0, per * sizeof(float), st));
    }
    for (int i = 0; i < e->npre; i++) {
        if (e->use_p && e->cp->pre_p) {
            const size_t per16 = (size_t)(e->cp->pslot_elems / e->B);
            HIPCHECK(e, hipMemsetAsync(e->cp->pinP[i].p + ((size_t)e->slot * e->cp->pslot_elems + per16 * b) * 4, 0, per16 * 16, st));
            continue;
        }
        const size_t per = (size_t)e->Ch[0] * T * e->F[0];
        HIPCHECK(e, hipMemsetAsync(e->pin[i][e->parity].p + per * b, 0, per * sizeof(float), st));
    }
    for (int l = 0; l < e->NL; l++)
        HIPCHECK(e, hipMemsetAsync(e->hbuf[l][e->hcur[l]].p + (size_t)H * b, 0, (size_t)H * sizeof(float), st));
    return SE_OK;
}

int se_forward(se_engine *e, const float *x, float *y, void *stream) {
    if (!e || !x || !y) return fail(e, SE_ERR_ARG, "null argument");
    if (e->B <= 0) return fail(e, SE_ERR_STATE, "se_forward before se_reset");
    int rc = ensure_ready(e);
    if (rc) return rc;
    const long F = e->F[0], T = e->T, M = e->M;
    return forward_dev(e, reinterpret_cast<const cf2 *>(x), M * F * T, F * T, 1, T, reinterpret_cast<cf2 *>(y), F * T, 1, T,
                       static_cast<hipStream_t>(stream));
}

#ifdef SE_DBG_STFT
// k_stft with self-checks (co-execution study): counters[0] = sig words changed during the kernel, [1] = window/twiddle
// words changed, [2] = output values that differ when the FFT of the same round is recomputed, [3] = rounds checked
__device__ unsigned long long g_stft_dbg[8];
// redundant-execution checks: the same FP32 chain twice in registers (VALU), and the same LDS butterfly pass twice
__global__ __launch_bounds__(256) void k_dbg_redundant(int iters) {
    extern __shared__ __align__(16) float lv[];
    const int tid = threadIdx.x;
    unsigned bad_valu = 0, bad_lds = 0;
    for (int it = 0; it < iters; it++) {
        float a0 = 1.0f + tid * 1e-3f + it, b0 = 0.5f + tid * 1e-4f;
        float a1 = a0, b1 = b0;
        asm volatile("" : "+v"(a1), "+v"(b1));
        float x0 = a0, x1 = a1, y0 = b0, y1 = b1;
#pragma unroll 16
        for (int k = 0; k < 128; k++) {
            x0 = fmaf(x0, 0.999f, y0); y0 = fmaf(y0, 1.001f, -x0 * 1e-3f);
            x1 = fmaf(x1, 0.999f, y1); y1 = fmaf(y1, 1.001f, -x1 * 1e-3f);
            asm volatile("" : "+v"(x1), "+v"(y1));
        }
        bad_valu += (x0 != x1) || (y0 != y1);
        // LDS: write per-thread values, barrier, read a permuted neighbour's, combine, twice into two buffers, compare
        float *A = lv, *B = lv + 4096, *C = lv + 8192;
        for (int i = tid; i < 4096; i += 256) A[i] = x0 + i;
        __syncthreads();
        for (int i = tid; i < 4096; i += 256) {
            const int j = (i * 5 + 1) & 4095, k2 = (i * 13 + 7) & 4095;
            B[i] = A[j] * 1.25f + A[k2];
        }
        __syncthreads();
        for (int i = tid; i < 4096; i += 256) {
            const int j = (i * 5 + 1) & 4095, k2 = (i * 13 + 7) & 4095;
            C[i] = A[j] * 1.25f + A[k2];
        }
        __syncthreads();
        for (int i = tid; i < 4096; i += 256) bad_lds += B[i] != C[i];
        __syncthreads();
    }
    if (bad_valu) atomicAdd(&g_stft_dbg[4], (unsigned long long)bad_valu);
    if (bad_lds) atomicAdd(&g_stft_dbg[5], (unsigned long long)bad_lds);
    if (tid == 0) atomicAdd(&g_stft_dbg[6], 1ull);
}
__global__ __launch_bounds__(256) void k_stft_dbg(StftArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int N = a.plan.N, N2 = N / 2, K = a.K, T = a.T, F = a.F, pad = N / 2;
    float *sig = reinterpret_cast<float *>(smem);
    float *win = sig + K + N;
    cf2 *tw = reinterpret_cast<cf2 *>(win + N);
    cf2 *bufA = tw + N;
    cf2 *bufB = bufA + kFftBatch * N2;
    const int row = blockIdx.x, tid = threadIdx.x, nth = blockDim.x;
    const float *src = a.src + (long)(row / a.M) * a.strideB + (long)(row % a.M) * a.strideM;
    for (int i = tid; i < K + N; i += nth) {
        const long k = (long)i - pad + a.off;
        sig[i] = (i >= pad && i < pad + K && k >= 0 && k < a.L) ? src[k] : 0.0f;
    }
    for (int i = tid; i < N; i += nth) { win[i] = a.window[i]; tw[i] = a.tw[i]; }
    __syncthreads();
    unsigned bad_out = 0, bad_fill = 0, bad_p1 = 0;
    cf2 *bufC = bufB + kFftBatch * N2;  // third buffer for the redundant fill / first pass (launch adds its bytes)
    for (int t0 = 0; t0 < T; t0 += kFftBatch) {
        const int nf = min(kFftBatch, T - t0);
        // redundant fill: A and C from the same sig / win
        for (int i = tid; i < nf * N2; i += nth) {
            const int f = i / N2, n = i - f * N2;
            const float *sp = sig + (t0 + f) * a.hop + 2 * n;
            bufA[i] = cf2{win[2 * n] * sp[0], win[2 * n + 1] * sp[1]};
        }
        for (int i = tid; i < nf * N2; i += nth) {
            const int f = i / N2, n = i - f * N2;
            const float *sp = sig + (t0 + f) * a.hop + 2 * n;
            bufC[i] = cf2{win[2 * n] * sp[0], win[2 * n + 1] * sp[1]};
        }
        __syncthreads();
        for (int i = tid; i < nf * N2; i += nth) bad_fill += (bufA[i].x != bufC[i].x) || (bufA[i].y != bufC[i].y);
        __syncthreads();
        // redundant first pass: A -> B and A -> C
        const int R0 = a.plan.radices[0];
        if (R0 == 4) {
            fft_pass<4>(bufA, bufB, N2, N2, nf, 1, tw, tid, nth, 2);
            __syncthreads();
            fft_pass<4>(bufA, bufC, N2, N2, nf, 1, tw, tid, nth, 2);
            __syncthreads();
            for (int i = tid; i < nf * N2; i += nth) bad_p1 += (bufB[i].x != bufC[i].x) || (bufB[i].y != bufC[i].y);
            __syncthreads();
        }
        cf2 first[12];  // this thread's outputs of the first computation (nf*F / 256 <= 12)
        for (int rep = 0; rep < 2; rep++) {
            for (int i = tid; i < nf * N2; i += nth) {
                const int f = i / N2, n = i - f * N2;
                const float *sp = sig + (t0 + f) * a.hop + 2 * n;
                bufA[i] = cf2{win[2 * n] * sp[0], win[2 * n + 1] * sp[1]};
            }
            __syncthreads();
            const cf2 *Z = fft_run(bufA, bufB, a.plan, nf, tw);
            int q = 0;
            for (int i = tid; i < nf * F; i += nth, q++) {
                const int f = i / F, k = i - f * F;
                const cf2 v = rfft_post(Z + f * N2, k, N2, tw);
                if (rep == 0) {
                    if (q < 12) first[q] = v;
                    a.spec[(long)row * a.sR + (long)(t0 + f) * a.sT + (long)k * a.sF] = v;
                } else if (q < 12) {
                    bad_out += (v.x != first[q].x) || (v.y != first[q].y);
                }
            }
            __syncthreads();
        }
    }
    if (bad_fill) atomicAdd(&g_stft_dbg[4], (unsigned long long)bad_fill);
    if (bad_p1) atomicAdd(&g_stft_dbg[5], (unsigned long long)bad_p1);
    unsigned bad_sig = 0, bad_tab = 0;
    for (int i = tid; i < K + N; i += nth) {
        const long k = (long)i - pad + a.off;
        const float want = (i >= pad && i < pad + K && k >= 0 && k < a.L) ? src[k] : 0.0f;
        bad_sig += sig[i] != want;
    }
    for (int i = tid; i < N; i += nth) bad_tab += (win[i] != a.window[i]) || (tw[i].x != a.tw[i].x) || (tw[i].y != a.tw[i].y);
    if (bad_sig) atomicAdd(&g_stft_dbg[0], (unsigned long long)bad_sig);
    if (bad_tab) atomicAdd(&g_stft_dbg[1], (unsigned long long)bad_tab);
    if (bad_out) atomicAdd(&g_stft_dbg[2], (unsigned long long)bad_out);
    if (tid == 0) atomicAdd(&g_stft_dbg[3], 1ull);
}
#endif

int se_stft(se_engine *e, const float *seg, int n, float *spec, void *stream) {
    if (!e || !seg || !spec || n <= 0) return fail(e, SE_ERR_ARG, "bad argument");
    HIPCHECK(e, hipSetDevice(e->device));
    const long F = e->F[0], T = e->T;
#ifdef SE_DBG_STFT
    if (getenv("SE_DBG_REDUNDANT")) {
        hipLaunchKernelGGL(k_dbg_redundant, dim3(n), dim3(256), 3 * 4096 * 4, static_cast<hipStream_t>(stream), 20);
        static int calls2 = 0;
        if (++calls2 % 1000 == 0) {
            (void)hipDeviceSynchronize();
            unsigned long long t[8];
            (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_stft_dbg), sizeof(t));
            fprintf(stderr, "[redundant dbg] after %d calls: VALU chain mismatches %llu, LDS pass mismatches %llu, workgroups %llu\n", calls2, t[4], t[5], t[6]);
        }
        return SE_OK;
    }
    if (getenv("SE_DBG_STFT_RUN")) {
        StftArgs a{};
        a.src = seg; a.strideB = e->K; a.strideM = 0; a.M = 1; a.off = 0; a.L = e->K;
        a.K = e->K; a.T = e->T; a.F = e->F[0]; a.hop = e->c.hop;
        a.spec = reinterpret_cast<cf2 *>(spec); a.sR = F * T; a.sT = 1; a.sF = T;
        a.window = e->window.p; a.tw = reinterpret_cast<const cf2 *>(e->tw.p); a.plan = e->plan;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_stft_dbg), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(stft_lds_bytes(e->K, e->N) + sizeof(cf2) * kFftBatch * (e->N / 2)));
        hipLaunchKernelGGL(k_stft_dbg, dim3(n), dim3(256), stft_lds_bytes(e->K, e->N) + sizeof(cf2) * kFftBatch * (e->N / 2), static_cast<hipStream_t>(stream), a);
        static int calls = 0;
        if (++calls % 1000 == 0 || getenv("SE_DBG_STFT_PRINT")) {
            (void)hipDeviceSynchronize();
            unsigned long long t[8];
            (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_stft_dbg), sizeof(t));
            fprintf(stderr, "[stft dbg] after %d calls: sig words changed %llu, table words changed %llu, recomputed outputs differing %llu, workgroups %llu, redundant fill mismatches %llu, redundant first-pass mismatches %llu\n", calls, t[0], t[1], t[2], t[3], t[4], t[5]);
        }
        return SE_OK;
    }
#endif
    return launch_stft(e, seg, e->K, 0, 1, 0, e->K, n, reinterpret_cast<cf2 *>(spec), F * T, 1, T, static_cast<hipStream_t>(stream));
}

int se_istft(se_engine *e, const float *spec, int n, float *wav, void *stream) {
    if (!e || !spec || !wav || n <= 0) return fail(e, SE_ERR_ARG, "bad argument");
    HIPCHECK(e, hipSetDevice(e->device));
    const long F = e->F[0], T = e->T;
    return launch_istft(e, reinterpret_cast<const cf2 *>(spec), F * T, 1, T, n, wav, e->K, static_cast<hipStream_t>(stream));
}

static int ensure_stage_streams(se_engine *e) {
    if (e->stage_ready) return 0;
    int least = 0, greatest = 0;
    HIPCHECK(e, hipDeviceGetStreamPriorityRange(&least, &greatest));
    for (int k = 0; k < 3; k++)  // the recurrence is a chain of short dependent launches: its waves go first when a CU frees up
        HIPCHECK(e, hipStreamCreateWithPriority(&e->stage_stream[k], hipStreamNonBlocking, k == 2 ? greatest : least));
    for (int r = 0; r < kRing; r++) {
        HIPCHECK(e, hipEventCreateWithFlags(&e->ev_enc[r], hipEventDisableTiming));
        HIPCHECK(e, hipEventCreateWithFlags(&e->ev_gru[r], hipEventDisableTiming));
        HIPCHECK(e, hipEventCreateWithFlags(&e->ev_dec[r], hipEventDisableTiming));
    }
    HIPCHECK(e, hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
    HIPCHECK(e, hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
    e->stage_ready = true;
    return 0;
}

static int step_dev(se_engine *e, const float *src, long strideB, long strideM, long off, long Lsrc, float *wav_out, long wav_ld, hipStream_t st) {
    const long F = e->F[0], T = e->T, M = e->M;
    int rc;
    cf2 *spec = reinterpret_cast<cf2 *>(e->spec[(e->slot + 1) % kRing].p);
    cf2 *ms = reinterpret_cast<cf2 *>(e->maskspec.p);
    if ((rc = launch_stft(e, src, strideB, strideM, (int)M, off, Lsrc, e->Bact * (int)M, spec, T * F, F, 1, st))) return rc;
    if ((rc = forward_dev(e, spec, M * T * F, T * F, F, 1, ms, T * F, F, 1, st))) return rc;
    return launch_istft(e, ms, T * F, F, 1, e->Bact, wav_out, wav_ld, st);
}

int se_step(se_engine *e, const float *wav_in, float *wav_out, void *stream) {
    if (!e || !wav_in || !wav_out) return fail(e, SE_ERR_ARG, "null argument");
    if (e->B <= 0) return fail(e, SE_ERR_STATE, "se_step before se_reset");
    int rc = ensure_ready(e);
    if (rc) return rc;
    return step_dev(e, wav_in, (long)e->M * e->K, e->K, 0, e->K, wav_out, e->K, static_cast<hipStream_t>(stream));
}

int se_realtime_process(se_engine *e, const float *mixture, int batch, int64_t length, int flag, float *out, void *stream) {
    if (!e || !mixture || !out || batch <= 0 || length <= 0) return fail(e, SE_ERR_ARG, "bad argument");
    int rc;
    if (!flag) {
        if ((rc = reset_on_stream(e, batch, static_cast<hipStream_t>(stream)))) return rc;  // CRN.py:574-575
    } else {
        if (e->B != batch) return fail(e, SE_ERR_STATE, "flag=True with batch %d but the carried state holds %d streams", batch, e->B);
        if ((rc = ensure_ready(e))) return rc;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long K = e->K, P = K / 2;
    const long lead = flag ? 0 : P;                      // CRN.py:568-570
    const long Lp = length + lead;
    const long gap = K - (P + Lp % K) % K;               // utility.py:327-329
    const long Nseg = 2 * (Lp + gap + P) / K;            // utility.py:360-368
    if ((rc = dev_alloc(e, e->yseg, (size_t)batch * Nseg * K))) return rc;
    bool piped = e->pipeline && !e->prof_on && Nseg > 1;
    if (piped && ensure_stage_streams(e)) {  // no extra streams / events available: fall back to the caller's stream for good
        e->pipeline = 0;
        piped = false;
    }
    // launch batch of segment n: with non-increasing lengths the streams still running are a prefix of the batch (see se_engine::Bact)
    static const bool compact_env = [] { const char *v = getenv("SE_RAGGED_COMPACT"); return !(v && v[0] == '0'); }();
    const bool compact = compact_env && !e->ragged_nseg.empty() && e->use_p && e->gemm_p && e->variant == 0;
    auto bact_of = [&](long n) {
        if (!compact) return e->B;
        int c = 0;
        while (c < e->B && e->ragged_nseg[c] > n) c++;
        return std::max(c, 1);
    };
    struct RestoreBact { se_engine *e; ~RestoreBact() { e->Bact = e->B; for (int &v : e->bact_slot) v = e->B; } } restore_bact{e};
    if (!piped) {
        for (long n = 0; n < Nseg; n++) {
            // segment n covers padded[n*P, n*P+K) with padded = [0]*P | [0]*lead | x | zeros
            const long off = n * P - P - lead;
            e->Bact = bact_of(n);
            // yseg is [B][Nseg][K]: the iSTFT writes segment n of every stream with row stride Nseg*K
            if ((rc = step_dev(e, mixture, (long)e->M * length, length, off, length, e->yseg.p + n * K, Nseg * K, st))) return rc;
        }
    } else {
        // Segments are sequentially dependent only WITHIN a stage (conv history, GRU state), so the three stages run as a
        // software pipeline over segments on three streams: while the bottleneck of segment n walks its 2 x T dependent
        // GRU launches, the encoder of n+1.. and the decoder of n-1 keep the matrix cores and HBM busy.
        hipStream_t sE = e->stage_stream[0], sD = e->stage_stream[1], sG = e->stage_stream[2];
        const long F = e->F[0], T = e->T, M = e->M;
        const size_t spec_n = (size_t)e->B * M * T * F * 2, mask_n = (size_t)e->B * T * F * 2;
        // The STFT of kFftSub segments is one launch on the encoder stream ahead of their encoders, the iSTFT one launch on
        // the decoder stream behind their decoders: both overlap with the other stages of neighbouring segments.
        const long CH = std::min<long>(Nseg, kPipeChunk);
        if ((rc = dev_alloc(e, e->spec_all, spec_n * CH)) || (rc = dev_alloc(e, e->mask_all, mask_n * CH))) return rc;
        // layer l of a recurrence round works on segment (round - l); SE_GRU_DIRECT=0 pins the LDS-slice step kernel, which has
        // no multi-layer form: layers then run back to back
        const bool lagged = e->gru_lag && e->gru_direct != 0 && !e->gru_seq;
        const int NL = e->NL, lag = lagged ? NL - 1 : 0;
        for (long c0 = 0; c0 < Nseg; c0 += CH) {
            const long cn = std::min(CH, Nseg - c0);
            HIPCHECK(e, hipEventRecord(e->ev_fork, st));
            for (hipStream_t q : e->stage_stream) HIPCHECK(e, hipStreamWaitEvent(q, e->ev_fork, 0));
            const int slot0 = e->slot;  // segment i of this chunk lives in ring slot (slot0 + 1 + i) % kRing
            auto slot_of = [&](long i) { return (int)((slot0 + 1 + i) % kRing); };
            for (long i = 0; i < cn + lag; i++) {
                if (i < cn) {  // ---- encoder stream: STFT batch, features + encoder, GRU input projection of segment i ----
                    const int cur = slot_of(i), prev = slot_of(i - 1);
                    e->slot = cur;
                    e->Bact = e->bact_slot[cur] = bact_of(c0 + i);
                    const cf2 *spec = reinterpret_cast<const cf2 *>(e->spec_all.p + spec_n * i);
                    if (i % kFftSub == 0) {
                        const long ns = std::min<long>(kFftSub, cn - i);
                        if ((rc = launch_stft(e, mixture, (long)e->M * length, length, (int)M, (c0 + i) * P - P - lead, length, e->Bact * (int)M,
                                              reinterpret_cast<cf2 *>(e->spec_all.p + spec_n * i), T * F, F, 1, sE, (int)ns, P, (long)(spec_n / 2)))) return rc;
                    }
                    if (i >= kRing) HIPCHECK(e, hipStreamWaitEvent(sE, e->ev_dec[cur], 0));  // slot cur was last read by the decoder of segment i - kRing
                    if ((rc = run_encoder(e, cur, prev, spec, M * T * F, T * F, F, 1, sE))) return rc;
                    if ((rc = stage_gru_proj0(e, cur, sE))) return rc;
                    HIPCHECK(e, hipEventRecord(e->ev_enc[cur], sE));
                    HIPCHECK(e, hipStreamWaitEvent(sG, e->ev_enc[cur], 0));
                }
                // ---- recurrence stream ----
                long done = -1;  // segment whose last layer completes in this round
                if (lagged) {
                    int slots[4] = {-1, -1, -1, -1};
                    for (int l = 0; l < NL; l++)
                        if (i - l >= 0 && i - l < cn) slots[l] = slot_of(i - l);
                    if ((rc = stage_gru_round(e, slots, sG))) return rc;
                    done = i - (NL - 1);
                } else {
                    e->Bact = e->bact_slot[slot_of(i)];
                    for (int l = 0; l < NL; l++)
                        if ((rc = stage_gru_layer(e, l, slot_of(i), sG, /*overlapped=*/true))) return rc;
                    done = i;
                }
                if (done < 0 || done >= cn) continue;
                // ---- decoder stream: fc + norm, decoder, mask, iSTFT batch of segment `done` ----
                const int dcur = slot_of(done);
                e->Bact = e->bact_slot[dcur];
                HIPCHECK(e, hipEventRecord(e->ev_gru[dcur], sG));
                HIPCHECK(e, hipStreamWaitEvent(sD, e->ev_gru[dcur], 0));
                const cf2 *dspec = reinterpret_cast<const cf2 *>(e->spec_all.p + spec_n * done);
                cf2 *ms = reinterpret_cast<cf2 *>(e->mask_all.p + mask_n * done);
                if ((rc = stage_gru_out(e, dcur, sD))) return rc;
                if ((rc = run_decoder(e, dcur, dspec, M * T * F, F, 1, ms, T * F, F, 1, sD))) return rc;
                if (done % kFftSub == kFftSub - 1 || done == cn - 1) {
                    const long i0 = done - done % kFftSub;
                    if ((rc = launch_istft(e, reinterpret_cast<const cf2 *>(e->mask_all.p + mask_n * i0), T * F, F, 1, bact_of(c0 + i0) /* (the ring slot of segment i0 has been reused by now) */,
                                           e->yseg.p + (c0 + i0) * K, Nseg * K, sD, (int)(done - i0 + 1), (long)(mask_n / 2), K))) return rc;
                }
                HIPCHECK(e, hipEventRecord(e->ev_dec[dcur], sD));
            }
            for (hipStream_t q : e->stage_stream) {  // the last decoder implies every earlier stage, but the join costs nothing
                HIPCHECK(e, hipEventRecord(e->ev_join, q));
                HIPCHECK(e, hipStreamWaitEvent(st, e->ev_join, 0));
            }
        }
    }
    const long skip = lead;  // CRN.py:587-588
    launch_k_overlap_avg(dim3((unsigned)((length + 255) / 256), batch), st, e->yseg.p, out, (int)Nseg, (int)K, (long)length, skip,
                         e->ragged_on ? reinterpret_cast<const long *>(e->ragged_len.p) : nullptr);
    HIPCHECK(e, hipGetLastError());
    return SE_OK;
}

int se_realtime_process_ragged(se_engine *e, const float *mixture, int batch, int64_t max_length, const int64_t *lengths_host, int flag, float *out,
                               void *stream) {
    if (!e || !lengths_host || batch <= 0) return fail(e, SE_ERR_ARG, "bad argument");
    for (int b = 0; b < batch; b++)
        if (lengths_host[b] <= 0 || lengths_host[b] > max_length) return fail(e, SE_ERR_ARG, "length of stream %d (%lld) outside (0, %lld]", b, (long long)lengths_host[b], (long long)max_length);
    int rc = dev_alloc(e, e->ragged_len, (size_t)batch * 2);
    if (rc) return rc;
    static_assert(sizeof(long) == sizeof(int64_t), "per-stream lengths are passed to the kernels as long");
    HIPCHECK(e, hipMemcpyAsync(e->ragged_len.p, lengths_host, (size_t)batch * sizeof(int64_t), hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)));
    HIPCHECK(e, hipStreamSynchronize(static_cast<hipStream_t>(stream)));  // the host array is borrowed for the call only
    // segments stream b takes part in (the count it would have alone: utility.py:327-329, 360-368), for the prefix compaction; only when the
    // lengths are non-increasing (the Python shim sorts the batch), otherwise every stream runs every segment as before
    e->ragged_nseg.clear();
    bool sorted = true;
    for (int b = 1; b < batch; b++) sorted = sorted && lengths_host[b] <= lengths_host[b - 1];
    if (sorted) {
        const long K = e->K, P = K / 2, lead = flag ? 0 : P;
        for (int b = 0; b < batch; b++) {
            const long Lp = lengths_host[b] + lead, gap = K - (P + Lp % K) % K;
            e->ragged_nseg.push_back((int)(2 * (Lp + gap + P) / K));
        }
    }
    e->ragged_on = true;
    rc = se_realtime_process(e, mixture, batch, max_length, flag, out, stream);
    e->ragged_on = false;
    e->ragged_nseg.clear();
    return rc;
}

static int copy_out(se_engine *e, const float *dev, size_t n, float *host, int64_t cap, int64_t *count, hipStream_t st) {
    if (count) *count = (int64_t)n;
    if ((int64_t)n > cap) return fail(e, SE_ERR_ARG, "buffer too small: need %zu floats", n);
    HIPCHECK(e, hipStreamSynchronize(st));
    HIPCHECK(e, hipMemcpy(host, dev, n * sizeof(float), hipMemcpyDeviceToHost));
    return SE_OK;
}

static int read_tap_impl(se_engine *e, const char *name, float *host_out, float *dev_out, int64_t capacity, int64_t *count, void *stream);

int se_read_tap(se_engine *e, const char *name, float *host_out, int64_t capacity, int64_t *count, void *stream) {
    if (!e || !name || !host_out) return fail(e, SE_ERR_ARG, "null argument");
    return read_tap_impl(e, name, host_out, nullptr, capacity, count, stream);
}

// The distillation feature maps "ft0".."ft<L>" straight into DEVICE memory ([B, C, F, T] fp32), enqueued on `stream`, no host copy and no
// synchronisation: what a distillation training loop consumes (distillation_crn.py:467-477).  Other taps are host-only (se_read_tap).
int se_read_tap_dev(se_engine *e, const char *name, float *dev_out, int64_t capacity, int64_t *count, void *stream) {
    if (!e || !name || !dev_out) return fail(e, SE_ERR_ARG, "null argument");
    if (strncmp(name, "ft", 2)) return fail(e, SE_ERR_KEY, "se_read_tap_dev serves the feature taps ft0..ft%d only (got %s)", e->L, name);
    return read_tap_impl(e, name, nullptr, dev_out, capacity, count, stream);
}

static int read_tap_impl(se_engine *e, const char *name, float *host_out, float *dev_out, int64_t capacity, int64_t *count, void *stream) {
    if (e->B <= 0) return fail(e, SE_ERR_STATE, "no forward has run");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int L = e->L, T = e->T, B = e->B, cur = e->slot;
    const float *src = nullptr;
    int C = 0, F = 0, idx = -1;
    bool gru_layout = false;
    if (e->use_p && strncmp(name, "ft", 2)) {  // plane path: operand tensors are split-bf16 planes, summed back on the host
        se_convp_state &S = *e->cp;
        const float *psrc = nullptr;
        if (!strcmp(name, "feat")) { psrc = S.xinP[0].p + (size_t)cur * S.slot_elems[0] * 4; C = e->Ch[0]; F = e->F[0]; }
        else if (!strcmp(name, "gru")) { psrc = S.decinP[cur].p; C = e->Ch[L]; F = e->F[L]; }
        else if (sscanf(name, "enc%d", &idx) == 1 && idx >= 0 && idx < L - 1) { psrc = S.xinP[idx + 1].p + (size_t)cur * S.slot_elems[idx + 1] * 4; C = e->Ch[idx + 1]; F = e->F[idx + 1]; }
        else if (sscanf(name, "dec%d", &idx) == 1 && idx >= 0 && idx < L - 1) { psrc = S.decP[idx].p; C = e->Ch[L - 1 - idx]; F = e->F[L - 1 - idx]; }
        if (psrc) {
            const size_t n = (size_t)B * C * T * F;
            if (count) *count = (int64_t)n;
            if ((int64_t)n > capacity) return fail(e, SE_ERR_ARG, "buffer too small: need %zu floats", n);
            return p_to_host(e, psrc, C, F, host_out, st, true);
        }
        if (e->gemm_p && sscanf(name, "enc%d", &idx) == 1 && idx == L - 1) {  // the GRU input lives as GEMM A planes [PL][B*T][(o*F+f)*8+c]
            const int Cl = e->Ch[L], Fl = e->F[L], PLn = operand_planes(e->precision), D = e->D;
            const size_t n = (size_t)B * Cl * T * Fl;
            if (count) *count = (int64_t)n;
            if ((int64_t)n > capacity) return fail(e, SE_ERR_ARG, "buffer too small: need %zu floats", n);
            std::vector<uint16_t> h((size_t)PLn * B * T * D);
            HIPCHECK(e, hipStreamSynchronize(st));
            HIPCHECK(e, hipMemcpy(h.data(), e->gruinP[cur].p, h.size() * 2, hipMemcpyDeviceToHost));
            for (int b = 0; b < B; b++)
                for (int c = 0; c < Cl; c++)
                    for (int t = 0; t < T; t++)
                        for (int f = 0; f < Fl; f++) {
                            float v = 0;
                            for (int pl = 0; pl < PLn; pl++) {
                                const uint16_t u = h[((size_t)pl * B * T + (size_t)b * T + t) * D + ((size_t)(c >> 3) * Fl + f) * 8 + (c & 7)];
                                if (PLn == 1) { _Float16 hv; memcpy(&hv, &u, 2); v += (float)hv; } else v += bf16_to_f32(u);
                            }
                            host_out[(((size_t)b * Cl + c) * Fl + f) * T + t] = v;
                        }
            return SE_OK;
        }
        idx = -1;  // enc{L-1} (the fp32 GRU input) and unknown names fall through
    }
    if (!strcmp(name, "feat")) { src = e->xin[0][cur].p; C = e->Ch[0]; F = e->F[0]; }
    else if (!strcmp(name, "gru")) { src = e->dec_in[cur].p; C = e->Ch[L]; F = e->F[L]; }
    else if (sscanf(name, "enc%d", &idx) == 1 && idx >= 0 && idx < L) {
        C = e->Ch[idx + 1]; F = e->F[idx + 1];
        if (idx + 1 < L) src = e->xin[idx + 1][cur].p;
        else { src = e->gru_in[cur].p; gru_layout = true; }
    } else if (sscanf(name, "dec%d", &idx) == 1 && idx >= 0 && idx < L - 1) {
        const int lvl = L - 1 - idx;
        src = e->dec_out[idx].p; C = e->Ch[lvl]; F = e->F[lvl];
    } else if (sscanf(name, "ft%d", &idx) == 1 && idx >= 0 && idx <= L) {
        // Pre-activation feature maps of the distillation student (distillation_crn.py:222-228, 126-128, 262-264, 467-477):
        // ft0 = last encoder block's convolution output, ft1 = fc_output_layer output ([B, T, D] memory viewed as [B, C, F, T],
        // distillation_crn.py:364), ft2.. = transposed-convolution outputs of decoder blocks 0..L-2 (before activation, norm and
        // frequency padding).  The hot path fuses the activation into the producing kernel, so the tap RE-RUNS that one kernel
        // without activation on the inputs still held in the ring slot; nothing on the inference path pays for it.
        const int prev = (cur + kRing - 1) % kRing;
        int rc = ensure_ready(e);
        if (rc) return rc;
        DevBuf tmp;
        // On the plane path the operands of the re-run live in the P layout: k_tap_p_to_f32 sums the planes back into the
        // first-generation fp32 buffers the re-run reads (a tap is a training / debugging aid, not part of the hot path).
        auto p_to_f32 = [&](const float *psrc, int C_, int F_, float *dst) -> int { return tap_p_to_f32_dev(e, psrc, C_, F_, dst, st); };
        auto finish = [&](int C_, int F_, bool raw_flat) -> int {
            const size_t n_ = (size_t)B * C_ * T * F_;
            if (count) *count = (int64_t)n_;
            if ((int64_t)n_ > capacity) { dev_free(tmp); return fail(e, SE_ERR_ARG, "buffer too small: need %zu floats", n_); }
            if (dev_out) {  // stays on the device: a copy (ft1: already [B, T, D] = [B, C, F, T] memory) or one transposition
                hipError_t er;
                if (raw_flat) er = hipMemcpyAsync(dev_out, tmp.p, n_ * sizeof(float), hipMemcpyDeviceToDevice, st);
                else { hipLaunchKernelGGL(k_tap_tf_to_ft, dim3(1024), dim3(256), 0, st, tmp.p, dev_out, (long)B * C_, T, F_); er = hipGetLastError(); }
                if (er == hipSuccess) er = hipStreamSynchronize(st);  // tmp is freed below (the re-run itself is the expensive part)
                dev_free(tmp);
                return er == hipSuccess ? SE_OK : fail(e, SE_ERR_HIP, "tap copy failed: %s", hipGetErrorString(er));
            }
            std::vector<float> h(n_);
            hipError_t st1 = hipStreamSynchronize(st), st2 = hipMemcpy(h.data(), tmp.p, n_ * sizeof(float), hipMemcpyDeviceToHost);
            dev_free(tmp);
            if (st1 != hipSuccess || st2 != hipSuccess) return fail(e, SE_ERR_HIP, "tap copy failed");
            if (raw_flat) { memcpy(host_out, h.data(), n_ * sizeof(float)); return SE_OK; }
            for (int b = 0; b < B; b++)
                for (int c = 0; c < C_; c++)
                    for (int t = 0; t < T; t++)
                        for (int f = 0; f < F_; f++) host_out[(((size_t)b * C_ + c) * F_ + f) * T + t] = h[(((size_t)b * C_ + c) * T + t) * F_ + f];
            return SE_OK;
        };
        if (idx == 0) {
            const int i = L - 1, Co = e->Ch[L], Fo = e->F[L];
            if (e->use_p) {
                se_convp_state &S = *e->cp;
                if ((rc = p_to_f32(S.xinP[i].p + (size_t)cur * S.slot_elems[i] * 4, e->Ch[i], e->F[i], e->xin[i][cur].p)) ||
                    (rc = p_to_f32(S.xinP[i].p + (size_t)prev * S.slot_elems[i] * 4, e->Ch[i], e->F[i], e->xin[i][prev].p))) return rc;
            }
            if ((rc = dev_alloc(e, tmp, (size_t)B * Co * T * Fo))) return rc;
            ConvPlan pl = e->lv[i].enc;
            pl.a.relu_lo = pl.a.relu_hi = 0;
            if ((rc = launch_conv(e, pl, e->xin[i][cur].p, e->xin[i][prev].p, tmp.p, st, "tap_ft"))) { dev_free(tmp); return rc; }
            return finish(Co, Fo, false);
        }
        if (idx == 1) {
            const int D = e->D, H = e->H;
            if ((rc = dev_alloc(e, tmp, (size_t)B * T * D))) return rc;
            if ((rc = launch_gemm(e, e->seqr[e->NL - 1][cur].p, H, e->fcw.p, H, e->fcb.p, tmp.p, D, B * T, D, H, 0, st, "tap_ft", e->fcw_x.p))) { dev_free(tmp); return rc; }
            return finish(e->Ch[L], e->F[L], true);
        }
        const int j = idx - 2, lvl = L - 1 - j;
        if (lvl <= 0) return fail(e, SE_ERR_KEY, "unknown tap %s", name);
        const int Co = e->Ch[lvl], Fo = 2 * e->F[lvl + 1] - 1;
        if ((rc = dev_alloc(e, tmp, (size_t)B * Co * T * Fo))) return rc;
        const float *xin = j == 0 ? e->dec_in[cur].p : e->dec_out[j - 1].p;
        if (e->use_p) {
            se_convp_state &S = *e->cp;
            const int Cin = e->Ch[lvl + 1], Fin = e->F[lvl + 1];
            if ((rc = p_to_f32(j == 0 ? S.decinP[cur].p : S.decP[j - 1].p, Cin, Fin, const_cast<float *>(xin)))) { dev_free(tmp); return rc; }
        }
        ConvPlan pe = e->lv[j].dec_even, po = e->lv[j].dec_odd;
        pe.a.relu_lo = pe.a.relu_hi = 0;
        po.a.relu_lo = po.a.relu_hi = 0;
        if ((rc = launch_conv(e, pe, xin, nullptr, tmp.p, st, "tap_ft")) || (rc = launch_conv(e, po, xin, nullptr, tmp.p, st, "tap_ft"))) { dev_free(tmp); return rc; }
        return finish(Co, Fo, false);
    } else return fail(e, SE_ERR_KEY, "unknown tap %s", name);
    const size_t n = (size_t)B * C * T * F;
    if (count) *count = (int64_t)n;
    if ((int64_t)n > capacity) return fail(e, SE_ERR_ARG, "buffer too small: need %zu floats", n);
    if (!host_out) return fail(e, SE_ERR_KEY, "tap %s has no device form", name);
    std::vector<float> h(n);
    HIPCHECK(e, hipStreamSynchronize(st));
    HIPCHECK(e, hipMemcpy(h.data(), src, n * sizeof(float), hipMemcpyDeviceToHost));
    for (int b = 0; b < B; b++)
        for (int c = 0; c < C; c++)
            for (int t = 0; t < T; t++)
                for (int f = 0; f < F; f++) {
                    const size_t si = gru_layout ? (((size_t)b * T + t) * C + c) * F + f : (((size_t)b * C + c) * T + t) * F + f;
                    host_out[(((size_t)b * C + c) * F + f) * T + t] = h[si];
                }
    return SE_OK;
}

int se_export_state(se_engine *e, const char *name, float *host_out, int64_t capacity, int64_t *count, void *stream) {
    if (!e || !name || !host_out) return fail(e, SE_ERR_ARG, "null argument");
    if (e->B <= 0) return fail(e, SE_ERR_STATE, "no state: call se_reset first");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int B = e->B, T = e->T, H = e->H;
    int idx = -1;
    if (!strcmp(name, "h")) {
        const size_t n = (size_t)e->NL * B * H;
        if (count) *count = (int64_t)n;
        if ((int64_t)n > capacity) return fail(e, SE_ERR_ARG, "buffer too small: need %zu floats", n);
        HIPCHECK(e, hipStreamSynchronize(st));
        for (int l = 0; l < e->NL; l++)
            HIPCHECK(e, hipMemcpy(host_out + (size_t)l * B * H, e->hbuf[l][e->hcur[l]].p, (size_t)B * H * sizeof(float), hipMemcpyDeviceToHost));
        return SE_OK;
    }
    const bool is_pbuf = sscanf(name, "pbuf%d", &idx) == 1 && idx >= 0 && idx < e->npre;
    if (is_pbuf || (sscanf(name, "buf%d", &idx) == 1 && idx >= 0 && idx < e->L)) {
        const int C = is_pbuf ? e->Ch[0] : e->Ch[idx], F = is_pbuf ? e->F[0] : e->F[idx], P = is_pbuf ? 4 : (2 << idx);
        const size_t n = (size_t)B * C * F * P, nsrc = (size_t)B * C * T * F;
        if (count) *count = (int64_t)n;
        if ((int64_t)n > capacity) return fail(e, SE_ERR_ARG, "buffer too small: need %zu floats", n);
        std::vector<float> h(nsrc);
        HIPCHECK(e, hipStreamSynchronize(st));
        if (e->use_p && !is_pbuf) {
            int rc = p_to_host(e, e->cp->xinP[idx].p + (size_t)e->slot * e->cp->slot_elems[idx] * 4, C, F, h.data(), st, false);
            if (rc) return rc;
        } else if (e->use_p && e->cp->pre_p) {
            int rc = p_to_host(e, e->cp->pinP[idx].p + (size_t)e->slot * e->cp->pslot_elems * 4, C, F, h.data(), st, false);
            if (rc) return rc;
        } else
        HIPCHECK(e, hipMemcpy(h.data(), (is_pbuf ? e->pin[idx][e->parity] : e->xin[idx][e->slot]).p, nsrc * sizeof(float), hipMemcpyDeviceToHost));
        for (size_t bc = 0; bc < (size_t)B * C; bc++)
            for (int f = 0; f < F; f++)
                for (int p = 0; p < P; p++) host_out[(bc * F + f) * P + p] = h[(bc * T + (T - P + p)) * F + f];
        return SE_OK;
    }
    return fail(e, SE_ERR_KEY, "unknown state %s", name);
}

int se_import_state(se_engine *e, const char *name, const float *host_in, int64_t count, void *stream) {
    if (!e || !name || !host_in) return fail(e, SE_ERR_ARG, "null argument");
    if (e->B <= 0) return fail(e, SE_ERR_STATE, "no state: call se_reset first");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int B = e->B, T = e->T, H = e->H;
    int idx = -1;
    HIPCHECK(e, hipStreamSynchronize(st));
    if (!strcmp(name, "h")) {
        if (count != (int64_t)e->NL * B * H) return fail(e, SE_ERR_SHAPE, "state h needs %d floats", e->NL * B * H);
        for (int l = 0; l < e->NL; l++)
            HIPCHECK(e, hipMemcpy(e->hbuf[l][e->hcur[l]].p, host_in + (size_t)l * B * H, (size_t)B * H * sizeof(float), hipMemcpyHostToDevice));
        return SE_OK;
    }
    const bool is_pbuf = sscanf(name, "pbuf%d", &idx) == 1 && idx >= 0 && idx < e->npre;
    if (is_pbuf || (sscanf(name, "buf%d", &idx) == 1 && idx >= 0 && idx < e->L)) {
        const int C = is_pbuf ? e->Ch[0] : e->Ch[idx], F = is_pbuf ? e->F[0] : e->F[idx], P = is_pbuf ? 4 : (2 << idx);
        if (count != (int64_t)B * C * F * P) return fail(e, SE_ERR_SHAPE, "state %s needs %ld floats", name, (long)B * C * F * P);
        const size_t nsrc = (size_t)B * C * T * F;
        std::vector<float> h(nsrc, 0.0f);
        for (size_t bc = 0; bc < (size_t)B * C; bc++)
            for (int f = 0; f < F; f++)
                for (int p = 0; p < P; p++) h[(bc * T + (T - P + p)) * F + f] = host_in[(bc * F + f) * P + p];
        if (e->use_p && !is_pbuf) return host_to_p(e, h, C, F, e->cp->xinP[idx].p + (size_t)e->slot * e->cp->slot_elems[idx] * 4);
        if (e->use_p && e->cp->pre_p) return host_to_p(e, h, C, F, e->cp->pinP[idx].p + (size_t)e->slot * e->cp->pslot_elems * 4);
        HIPCHECK(e, hipMemcpy((is_pbuf ? e->pin[idx][e->parity] : e->xin[idx][e->slot]).p, h.data(), nsrc * sizeof(float), hipMemcpyHostToDevice));
        return SE_OK;
    }
    return fail(e, SE_ERR_KEY, "unknown state %s", name);
}

double se_flops_per_frame(const se_engine *e) {
    if (!e) return 0;
    const int L = e->L, T = e->T, H = e->H, D = e->D;
    double mac = 0;
    for (int i = 0; i < L; i++) mac += (double)e->Ch[i + 1] * e->Ch[i] * 15 * e->F[i + 1] * T;
    if (e->variant) {
        for (int i = 0; i < L; i++) mac += 2.0 * e->Ch[i + 1] * e->Ch[i + 1] * e->F[i + 1] * T;  // conv_trans + conv_gated
        mac += 3.0 * ((double)e->Ch[0] * e->Ch[0] * 25 + 2.0 * e->Ch[0] * e->Ch[0]) * e->F[0] * T;  // preconv blocks
    }
    for (int j = 0; j < L; j++) {
        const int lvl = L - 1 - j, Ci = e->Ch[lvl + 1], Co = lvl == 0 ? 2 : e->Ch[lvl];
        mac += (double)Ci * Co * 15 * e->F[lvl + 1] * T;
        if (lvl > 0) mac += 2.0 * Co * Co * e->F[lvl] * T;
    }
    for (int l = 0; l < e->NL; l++) mac += (double)T * (3.0 * H * (l == 0 ? D : H) + 3.0 * H * H);
    mac += (double)T * D * H;
    return 2.0 * mac;
}

int se_frames_per_segment(const se_engine *e) { return e ? e->T : 0; }

int se_profile(se_engine *e, int enable) {
    if (!e) return SE_ERR_ARG;
    HIPCHECK(e, hipSetDevice(e->device));
    HIPCHECK(e, hipDeviceSynchronize());
    for (auto &r : e->prof_recs) { e->prof_pool.push_back(r.a); e->prof_pool.push_back(r.b); }
    e->prof_recs.clear();
    for (auto &l : e->prof_labels) { l.ms = 0; l.launches = 0; }
    e->prof_on = enable != 0;
    return SE_OK;
}

int se_profile_read(se_engine *e, int index, char *kernel, char *label, int cap, double *ms_total, int64_t *launches, double *flops_per_launch) {
    if (!e) return SE_ERR_ARG;
    if (!e->prof_recs.empty()) {  // fold pending records
        HIPCHECK(e, hipDeviceSynchronize());
        for (auto &r : e->prof_recs) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { e->prof_labels[r.label].ms += ms; e->prof_labels[r.label].launches++; }
            e->prof_pool.push_back(r.a); e->prof_pool.push_back(r.b);
        }
        e->prof_recs.clear();
    }
    if (index < 0 || index >= (int)e->prof_labels.size()) return 1;  // end of list
    const auto &l = e->prof_labels[index];
    if (kernel) snprintf(kernel, cap, "%s", l.kernel.c_str());
    if (label) snprintf(label, cap, "%s", l.label.c_str());
    if (ms_total) *ms_total = l.ms;
    if (launches) *launches = l.launches;
    if (flops_per_launch) *flops_per_launch = l.flops;
    return SE_OK;
}

}  // extern "C"

#include "fsn_engine.inc.h"
#include "train_ops.inc.h"

/*
 * se_engine.h - C ABI of the MI355X-native streaming speech-enhancement engine (libse_engine.so).
 *
 * The reference (KI-D/Speech-Enhancement-Mi) has no FFI: its plug-in point is the Python model class
 * (README.md:22).  This ABI is what a drop-in TemporalCRN class binds instead of running torch ops; each
 * entry point names the reference interface it replaces (file:line into the reference tree).  The
 * reference-side binding (ctypes) is shown in INTEGRATION.md and shipped in
 * speech_enhancement_mi_amd/crn.py.
 *
 * Conventions
 *  - plain C, no torch types.  All tensor pointers are DEVICE pointers (HIP, fp32, contiguous) unless
 *    a parameter says "host".  Pointers are borrowed for the duration of the call only.
 *  - every call enqueues on the caller's hipStream_t (passed as void*; NULL = default stream) and does
 *    not synchronise; the caller serialises calls per handle (the reference class is not re-entrant
 *    either, CRN.py:325-337).
 *  - return 0 on success, negative se_status on error; se_last_error() gives the message (the Python
 *    shim re-raises it as RuntimeError, mirroring the reference's exceptions-only convention).
 */
#ifndef SE_ENGINE_H
#define SE_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SE_MAX_LEVELS 8

typedef enum {
    SE_OK = 0,
    SE_ERR_ARG = -1,        /* bad argument / unsupported configuration */
    SE_ERR_KEY = -2,        /* unknown checkpoint key */
    SE_ERR_SHAPE = -3,      /* parameter shape mismatch */
    SE_ERR_STATE = -4,      /* call order (step before reset, flag=True with another batch, ...) */
    SE_ERR_HIP = -5,        /* HIP runtime error */
    SE_ERR_PARAM_MISSING = -6 /* a weight was never loaded */
} se_status;

/* TemporalCRN.__init__ kwargs (CRN.py:415-417, config.yaml:205-217).  win/hop are in samples
 * (= round(sample_rate/1000 * win_length_ms), the speechbrain STFT convention, CRN.py:421-425). */
typedef struct {
    int32_t num_levels;                 /* len(num_channels) */
    int32_t channels[SE_MAX_LEVELS];    /* num_channels */
    int32_t num_freqs;                  /* n_fft/2+1 */
    int32_t hidden;
    int32_t num_layers;                 /* GRU layers */
    int32_t num_inputs;                 /* microphones M */
    int32_t kernel_size;                /* 3 */
    int32_t n_fft, win, hop;
    int32_t segment_length;             /* K = 3200 */
    int32_t variant;                    /* 0 = CRN.py TemporalCRN (ReLU); 1 = CRN_ELU.py TemporalCRN (ELU, gated 1x1 convs,
                                           three 5x5 frequency-dilated preconv blocks, atan2 phase; CRN_ELU.py:321-365);
                                           2 = distillation_crn.py TemporalCRN, the student architecture (as 1, but arctan
                                           phase and gLN denominator sqrt(var)+eps; distillation_crn.py:51,340) */
    int32_t precision;                  /* 0 = fp32-accurate contractions (6-term split-bf16 MFMA); 1 = fp16 MFMA operands with
                                           fp32 accumulation for the convolutions and dense layers (`model.half()`; OUTSIDE the
                                           1e-4 parity bar: 2e-3 relative); 2 = 3-term split-bf16 (hi*hi + hi*mid + mid*hi, 16
                                           mantissa bits per operand, fp32 accumulation: inside the 1e-4 / 0.02 dB bar, half the
                                           matrix work of mode 0; the fast mode of BASELINE config 5).  The recurrence stays fp32. */
} se_config;

typedef struct se_engine se_engine;

/* TemporalCRN(**config['TemporalCRN'])  (CRN.py:415-451; train.py:58, predict.py:45) */
int se_create(const se_config *cfg, int device, se_engine **out);
void se_destroy(se_engine *e);
/* message of the last failing call on this handle (or of se_create when e == NULL) */
const char *se_last_error(const se_engine *e);

/* load_state_dict(): one tensor per call, by reference checkpoint key (e.g.
 * "convlist.0.conv.weight", "gru.sequence_model.weight_hh_l1").  `data` is a HOST pointer.
 * The `net.0.*` aliases the reference's state_dict carries (CRN.py:314-316) are accepted. */
int se_load_param(se_engine *e, const char *key, const float *host_data, const int64_t *shape, int ndim);

/* TemporalCRN.reset() + lazy state allocation for B streams (CRN.py:498-503, 325-326). */
int se_reset(se_engine *e, int batch);

/* Reset ONE stream of the batch (a call ends, a new caller takes its slot) while the others keep their state: zeroes that
 * stream's conv time buffers and GRU state, i.e. what TemporalCRN.reset() does (CRN.py:498-503) restricted to row
 * `stream_index` of every state tensor.  The reference can only reset the whole batch; this is the continuation API a
 * server needs around the path.  Enqueued on `stream`. */
int se_reset_stream(se_engine *e, int stream_index, void *stream);

/* One hot-path step: all B streams advance by one K-sample window.
 * wav_in [B, M, K] -> wav_out [B, K]  == istft_trans(forward(stft_trans(x)))  (CRN.py:505-520, 454-496) */
int se_step(se_engine *e, const float *wav_in, float *wav_out, void *stream);

/* TemporalCRN.realtime_process(mixture, flag) (CRN.py:560-589): mixture [B, M, L] -> out [B, L].
 * flag == 0: reset + K/2 left pad (stripped again); flag != 0: carry state, B must match. */
int se_realtime_process(se_engine *e, const float *mixture, int batch, int64_t length, int flag,
                        float *out, void *stream);

/* Ragged batch (the reference's utterances are 1 .. 3.75 s long, data_c.py:155-173): stream b holds lengths[b] <= max_length valid
 * samples of mixture [B, M, max_length]; every stream gets exactly the output it would get alone - its own zero padding (samples past
 * its length read as zeros) and out[b, lengths[b]:] = 0.  lengths is a HOST array.
 * Prefix compaction: when the lengths are NON-INCREASING (sort the batch; the Python shim does) every segment is launched for the prefix
 * of streams that still take part in it only - every layout is stream-major, so nothing moves; grids, GEMM rows and recurrence rows
 * shrink (plane path, CRN.py variant, batches on the plane-GEMM route; SE_RAGGED_COMPACT=0 turns it off).  With unsorted lengths every
 * stream runs every segment.  Either way the carried state of a stream that ended early is not a continuation state: follow with
 * flag = 0, or se_reset_stream, for those streams. */
int se_realtime_process_ragged(se_engine *e, const float *mixture, int batch, int64_t max_length, const int64_t *lengths_host, int flag, float *out,
                               void *stream);

/* Per-stage entry points (parity tests; same arithmetic as inside se_step).
 * se_stft:    stft_trans  (CRN.py:505-512): seg [n, K] -> spec [n, F, T, 2]   (n = B*M rows)
 * se_istft:   istft_trans (CRN.py:514-520): spec [n, F, T, 2] -> wav [n, K]
 * se_forward: TemporalCRN.forward (CRN.py:454-496): x [B, M, F, T, 2] -> y [B, F, T, 2]; stateful. */
int se_stft(se_engine *e, const float *seg, int n, float *spec, void *stream);
int se_istft(se_engine *e, const float *spec, int n, float *wav, void *stream);
int se_forward(se_engine *e, const float *x, float *y, void *stream);

/* Debug taps of the last forward, converted to the reference's [B, C, F, T] layout, copied to HOST.
 * name: "feat" (encoder input, after the preconv blocks for variants 1/2), "enc0".."encN", "gru", "dec0".."decN";
 * "ft0".."ftL": the pre-activation feature maps the distillation student returns (distillation_crn.py:467-477): last encoder
 * convolution [B, C, F, T], fc_output_layer output ([B, T, D] memory), transposed-convolution outputs of decoder blocks 0..L-2.
 * Synchronises the stream.  *count = elements. */
int se_read_tap(se_engine *e, const char *name, float *host_out, int64_t capacity, int64_t *count, void *stream);
/* The student's distillation feature maps "ft0".."ft<L>" (distillation_crn.py:467-477: last encoder convolution, fc layer, transposed
 * convolutions, all before their activation) written to DEVICE memory as [B, C, F, T] fp32 on `stream`; no host copy.  Other tap names:
 * SE_ERR_KEY (host-only debugging taps). */
int se_read_tap_dev(se_engine *e, const char *name, float *dev_out, int64_t capacity, int64_t *count, void *stream);

/* Streaming state hand-over: encoder time buffers "buf<i>" [B, Cin, F, 2d] and GRU
 * hidden "h" [layers, B, H] in the reference's layouts, HOST pointers; variants 1/2 add the preconv buffers
 * "pbuf<i>" [B, 2M-1, F, 4].  Synchronises. */
int se_export_state(se_engine *e, const char *name, float *host_out, int64_t capacity, int64_t *count, void *stream);
int se_import_state(se_engine *e, const char *name, const float *host_in, int64_t count, void *stream);

/* Introspection: algorithmic FLOPs per frame (dense contractions only) and geometry. */
double se_flops_per_frame(const se_engine *e);
int se_frames_per_segment(const se_engine *e); /* T */

/* Per-kernel timing for bench.py's roofline leg: with profiling enabled every kernel launch is bracketed by
 * two HIP events on the launch stream.  se_profile(e, on) clears the counters; se_profile_read() folds the
 * pending events (synchronises) and returns entry `index` (kernel function name, launch-site label, total
 * milliseconds, launches, algorithmic FLOPs per launch); returns 1 past the last entry. */
int se_profile(se_engine *e, int enable);
int se_profile_read(se_engine *e, int index, char *kernel, char *label, int cap, double *ms_total,
                    int64_t *launches, double *flops_per_launch);

int se_abi_version(void);  /* 4 since round 3 (fsn_config.precision, se_sig_*, fused training stages, se_realtime_process_ragged, se_read_tap_dev, se_loss_stoi_*) */
/* sizeof(se_config) / sizeof(fsn_config) as this library was built: a binding checks its own struct mirror against these
 * before the first se_create (a short struct would leave `precision` reading whatever follows it). */
int se_config_size(void);
int fsn_config_size(void);

/* ---- FullSubNet (reference fullsubnet.py:685-961; BASELINE config 3) -------------------
 * Same conventions as the se_* calls above.  Checkpoint keys: fb_model.* / sb_model.* of FullSubNet.state_dict(). */
typedef struct {
    int32_t num_freqs, num_mics;
    int32_t fb_hidden, sb_hidden;       /* fb_model_hidden_size, sb_model_hidden_size */
    int32_t num_layers;
    int32_t sb_neighbors, fb_neighbors; /* sb_num_neighbors (15), fb_num_neighbors (0: the only supported value) */
    int32_t look_ahead;                 /* 0 */
    int32_t n_fft, win, hop, segment_length;
    int32_t precision;                  /* ABI 4: 0 = fp32-accurate LSTM contractions (6-term split-bf16 MFMA); 2 = 3-term split-bf16
                                           ("bf16x3": inside the 1e-4 / 0.02 dB parity bar, half the matrix work); 1 is not offered */
} fsn_config;

typedef struct fsn_engine fsn_engine;

/* FullSubNet(**config['FullSubNet'])  (fullsubnet.py:686-767; config.yaml:153-172; LSTM, ReLU / no output activation) */
int fsn_create(const fsn_config *cfg, int device, fsn_engine **out);
void fsn_destroy(fsn_engine *e);
const char *fsn_last_error(const fsn_engine *e);
int fsn_load_param(fsn_engine *e, const char *key, const float *host_data, const int64_t *shape, int ndim);
/* reset_state (fullsubnet.py:826-832): zero LSTM states, reset both CumLayerNorm running means */
int fsn_reset(fsn_engine *e, int batch);
/* FullSubNet.forward (fullsubnet.py:769-824): x [B, 2M, F, T] (re x M, then im x M) -> compressed mask [B, 2, F, T]; stateful */
int fsn_forward(fsn_engine *e, const float *x, float *crm, void *stream);
/* FullSubNet.realtime_process(mixture, source, flag, train=False)[0] (fullsubnet.py:903-961): [B, M, L] -> [B, L] */
int fsn_realtime_process(fsn_engine *e, const float *mixture, int batch, int64_t length, int flag, float *out, void *stream);
/* host copies of "fb_out" [B*T, F], "mean_fb" [B], "mean_sb" [B] after the last forward */
int fsn_read_tap(fsn_engine *e, const char *name, float *host_out, int64_t capacity, int64_t *count, void *stream);
double fsn_flops_per_frame(const fsn_engine *e);

/* ---- training loss (reference CRN.py:593-617 compute_loss) ------------------------------------------------
 * SI-SNR term, utility.cal_si_snr (utility.py:207-223), device-resident: separated / source [B, L] fp32 device tensors,
 * lens [B] int64 device tensor (samples that count per utterance).  fwd writes the per-utterance SI-SNR in dB to per_utt [B]
 * and 8 doubles of saved scalars per utterance to stats; bwd writes grad [B, L] = gscale[0] * d SI-SNR_b / d separated
 * (gscale is a 1-element device tensor: upstream gradient / B for the batch mean).  Enqueued on `stream`, no sync. */
int se_loss_sisnr_fwd(const float *separated, const float *source, const int64_t *lens, int batch, int64_t length, float *per_utt,
                      double *stats, void *stream);
/* STOI term of compute_loss (utility.stoi_loss, utility.py:821-916) as kernels, forward and backward w.r.t. the prediction.
 * Tables (device memory, built by the caller once: losses._plan): rs_w [5][W] + rs_first [5] = the 16 kHz -> 10 kHz polyphase filter
 * (Kaldi LinearResample, augment.py:478-545), hann_sym [256] = np.hanning(256) (utility.py:522), hann_per [256] = periodic Hann of the
 * spectrogram, band_lo / band_hi [15] = third-octave band bin ranges (utility.thirdoct, utility.py:480-518).  ws = se_loss_stoi_ws_floats
 * floats of scratch that carry the forward's intermediates to the backward.  D [batch] = STOI per utterance (0.99 for utterances too
 * short to score, utility.py:877-880); gD = d loss / d D; dpred [batch][length]. */
int64_t se_loss_stoi_ws_floats(int batch, int64_t length);
int se_loss_stoi_fwd(const float *clean, const float *pred, const int64_t *lens, int batch, int64_t length, const float *rs_w, const int *rs_first, int W,
                     const float *hann_sym, const float *hann_per, const int *band_lo, const int *band_hi, float *ws, float *D, void *stream);
int se_loss_stoi_bwd(const float *gD, const int64_t *lens, int batch, int64_t length, const float *rs_w, const int *rs_first, int W, const float *hann_sym,
                     const float *hann_per, const int *band_lo, const int *band_hi, float *ws, float *dpred, void *stream);
const char *se_loss_stoi_last_error(void);
int se_loss_sisnr_bwd(const float *separated, const float *source, const int64_t *lens, int batch, int64_t length, const double *stats,
                      const float *gscale, float *grad, void *stream);

/* ---- training-step building blocks (reference train.py:195-204 = torch autograd over CRN.py:290-401, 196-287) --
 * fp32-exact MFMA kernels on device tensors; activations are [B][C][T][F] (F innermost).  The Python side
 * (speech_enhancement_mi_amd/train_ops.py) wires them into torch.autograd.Function objects; torch autograd is the checker.
 * All calls enqueue on `stream` and return 0 or a negative se_status; se_train_last_error() gives the message. */
typedef struct {
    int32_t ntap, CC, nchunk, CoPad, FP;   /* weights are passed pre-arranged as [nchunk][ntap][CC][CoPad] fp32 (zero padded) */
    int32_t tap_kf[15], tap_kt[15];        /* tap t of that arrangement = reference kernel element (kf, kt) */
} se_train_conv_layout;
const char *se_train_last_error(void);
/* kind 0: TemporalConv2d (CRN.py:314: 5x3, stride (2,1), padding (2,0), dilation (1,d), causal with the history rows `xprev`);
 * kind 1 / 2: even / odd output-frequency parity of TemporalConvTranspose2d (CRN.py:369, keeping the last T columns).
 * The input gradient of kind 0 is kinds 1 + 2 applied to dy with the SAME weight tensor (and vice versa): same index algebra. */
int se_train_conv_layout_query(int kind, int Ci, int Co, int T, int Fi, int Fy, int dil, se_train_conv_layout *out);
int se_train_conv(int kind, const float *x, const float *xprev, const float *w_arranged, const float *bias, float *y, int B, int Ci, int Co,
                  int T, int Fi, int Fy, int dil, int act, void *stream);
/* C[a][b][5][3] = sum_{batch,t,m} G[a][t][m] * S[b][t-(2-kt)d][2m+kf-2]: weight gradient of kind 0 (G = dy, S = x, Sprev = history)
 * and of the transposed convolution (G = x, S = dy, Sprev = NULL) */
int se_train_conv_wgrad(const float *G, const float *S, const float *Sprev, float *C, int B, int Ca, int Cb, int T, int Fm, int Fs, int dil,
                        void *stream);
/* C[M][N] = act(A[M][K] W[N][K]^T + bias[N])  (act: 0 none, 1 ReLU, 2 ELU); bias may be NULL; K % 8 == 0 */
int se_train_gemm(const float *A, const float *W, const float *bias, float *C, int M, int N, int K, int act, void *stream);
/* C[Na][Nb] = sum_r A[r][i] B[r][j], both operands row-major [R][.]: dense / 1x1-convolution weight gradients over R rows */
int se_train_gemm_tn(const float *A, const float *B, float *C, int64_t R, int Na, int Nb, void *stream);
/* one GRU time step (CRN.py:269 nn.GRU cell), saving r, z, n, gh_n per row in `gates` for the backward pass */
int se_train_gru_step(const float *gi, int64_t gi_ld, const float *hprev, const float *whh, const float *bhh, float *hout, float *seq,
                      int64_t seq_ld, float *gates, int64_t gates_ld, int B, int H, void *stream);
/* gate derivatives of one step: dh = d1 + d2 + d3 (NULL = absent) -> dgi, dgh (rows of 3H), dhz = z * dh */
int se_train_gru_bwd_gates(const float *d1, int64_t d1_ld, const float *d2, const float *d3, const float *gates, int64_t gates_ld, const float *hprev,
                           int64_t hprev_ld, float *dgi, float *dgh, int64_t dg_ld, float *dhz, int B, int H, void *stream);
/* all T steps of one GRU layer in one call: forward (gate values saved) and the BPTT sweep with the gradient cut at segment
 * boundaries ((t + 1) % seg_len == 0: the carried state is detached per segment, CRN.py:281).  gi [B][T][3H], out [B][T][H],
 * gates [B][T][4H], whh_t = W_hh^T [H][3H]; scratch: 2 B H floats (forward), 4 B H floats (backward); backward: B <= 16 */
int se_train_gru_seq_fwd(const float *gi, const float *h0, const float *whh, const float *bhh, float *out, float *gates, float *hT, float *scratch,
                         int B, int T, int H, void *stream);
int se_train_gru_seq_bwd(const float *dout, const float *dhT, const float *gates, const float *out, const float *h0, const float *whh_t, float *dgi,
                         float *dgh, float *scratch, int B, int T, int H, int seg_len, void *stream);

/* ONE persistent launch per GRU layer and direction (csrc/gru_pseq.hip.h): H/16 resident workgroups per group of <= 32 streams keep
 * their W_hh slice in registers and exchange the state vector per step through write-through stores + an agent-scope arrival counter.
 * Replaces T dependent step launches (round 2).  Any number of streams: more than 32 run as independent 16-stream groups of the same
 * launch (rows must then be [B][T], ldN = 0) - which is how the training backward runs, because the state is detached at every segment
 * seam (CRN.py:281): S = segments x utterances independent streams of Tseg steps.  Rows of gi / out / gates / dout / dgi / dgh are
 * addressed as row(b, s) = (s / Tseg) * ldN + b * ldB + s % Tseg, which covers [B][T] (Tseg = T, ldN = 0, ldB = T) and the training
 * forward's segment-major [N][B][Tseg] (ldN = B * Tseg, ldB = Tseg).  scratch: se_train_gru_pseq_scratch_floats(B, H) floats; word [1]
 * of scratch is non-zero after a bounded-spin timeout (results invalid).  gates may be NULL in the forward (inference / no_grad).
 * seg_len as in se_train_gru_seq_bwd. */
int se_train_gru_pseq_scratch_floats(int B, int H);
int se_train_gru_pseq_supported(int B, int H);
int se_train_gru_pseq_fwd(const float *gi, const float *h0, const float *whh, const float *bhh, float *out, float *gates, float *hT,
                          float *scratch, int B, int T, int H, int Tseg, int64_t ldN, int64_t ldB, void *stream);
int se_train_gru_pseq_bwd(const float *dout, const float *dhT, const float *gates, const float *out, const float *h0, const float *whh_t,
                          float *dgi, float *dgh, float *scratch, int B, int T, int H, int Tseg, int64_t ldN, int64_t ldB, int seg_len,
                          void *stream);

/* ---- round 3: the norm / pointwise / signal stages of the training step, forward and backward (csrc/train_fused.hip.h) ----
 * Layout: activations [S][C][T][F] fp32, S = N segments x B utterances, SEGMENT-major (stream n * B + b), so the time history of
 * segment n is the same tensor one slab (B streams) earlier and the per-segment loop of realtime_process (CRN.py:577-586) becomes one
 * launch per layer.  All gradients of per-channel parameters are produced as [S][C] slabs and folded by se_train_colsum: the
 * training step contains no float atomics (bit-reproducible gradients). */
typedef struct se_sig se_sig;   /* STFT tables: hamming(win) centred in n_fft, twiddles, overlap-add envelope */
int se_sig_create(int n_fft, int win, int hop, int segment_length, int device, se_sig **out);
void se_sig_destroy(se_sig *g);
/* stft_trans of ALL segments (CRN.py:505-512 + utility.padding / segmentation, utility.py:312-370): wav [B][M][L]; segment y of row
 * (b, m) covers samples off0 + y * seg_off + [0, K) (zero outside [0, L)) -> spec [nseg][B*M][T][F][2] */
int se_sig_stft(se_sig *g, const float *wav, int B, int M, int64_t L, int64_t off0, int64_t seg_off, int nseg, float *spec, void *stream);
/* istft_trans (CRN.py:514-520): spec [rows][T][F][2] -> wav [rows][K].  Its adjoint is se_sig_stft of (dy / envelope) followed by the
 * irfft weights, which se_train_ola_bwd and se_train_mask_bwd apply. */
int se_sig_istft(se_sig *g, const float *spec, int rows, float *wav, void *stream);
/* utility.over_add + the K/2 strip (utility.py:373-403, CRN.py:587-588) on segment-major yseg [nseg][B][K] -> out [B][L]; skip = K/2
 * when realtime_process padded the input (flag=False), else 0.  bwd: gseg [nseg][B][K] = adjoint(dout) / envelope. */
int se_train_ola_fwd(se_sig *g, const float *yseg, float *out, int B, int64_t L, int64_t skip, void *stream);
int se_train_ola_bwd(se_sig *g, const float *dout, float *gseg, int B, int nseg, int64_t L, int64_t skip, void *stream);
/* features (CRN.py:463-467): spec [S][M][T][F][2] -> feat [S][2M-1][T][F] (no backward: the input carries no gradient) */
int se_train_feat(const float *spec, float *feat, int S, int M, int T, int F, int atan2_phase, void *stream);
/* decompress_cIRM + complex multiply with microphone 0 (utility.py:439-442, CRN.py:491-495): x [S][2][T][F] -> Y [S][T][F][2];
 * bwd takes dY = se_sig_stft(gseg) (raw) and returns dx */
int se_train_mask_fwd(const float *x, const float *spec, float *Y, int S, int M, int T, int F, void *stream);
int se_train_mask_bwd(const float *dY, const float *x, const float *spec, float *dx, int S, int M, int T, int F, int n_fft, void *stream);
/* y = GlobalLayerNorm(act(x)) (CRN.py:135-149) per stream; element (s, c, t, f) of x at s*xS + c*xC + t*xT + f (same form for y, dy);
 * mode 0: affine per channel, mode 1: per feature c*Fi + f (last=True); act 0 none / 1 ReLU / 2 ELU; Fo >= Fi zero-fills the
 * decoder's frequency pad (CRN.py:389-392); stats [S][2] = mean, 1/(sqrt(var+eps)+eps).
 * bwd: dx (x strides) = gradient w.r.t. the PRE-activation x; dw_part / db_part / dpre_part [S][NA] slabs (NA = C or C*Fi; any may be
 * NULL), dpre = per-channel sum of dx = the producing convolution's bias gradient. */
int se_train_gln_fwd(const float *x, int64_t xS, int64_t xC, int64_t xT, float *y, int64_t yS, int64_t yC, int64_t yT, const float *w, const float *b,
                     float *stats, int S, int C, int T, int Fi, int Fo, int mode, int act, int eps_mode, void *stream);
int se_train_gln_bwd(const float *dy, int64_t dS, int64_t dC, int64_t dT, const float *x, int64_t xS, int64_t xC, int64_t xT, float *dx, const float *w,
                     const float *stats, float *dw_part, float *db_part, float *dpre_part, int S, int C, int T, int Fi, int mode, int act,
                     int eps_mode, void *stream);
/* out_k[j] (+)= sum over the R rows of part_k[r][j], k < 3 (fixed order); _tall: R up to millions via 64-row chunks in ws[ceil(R/64)][n] */
int se_train_colsum(const float *p0, float *o0, int n0, const float *p1, float *o1, int n1, const float *p2, float *o2, int n2, int R, int accumulate,
                    void *stream);
int se_train_colsum_tall(const float *x, int64_t R, int n, float *ws, float *out, int accumulate, void *stream);
/* decoder skip gate (CRN.py:387-396): uv [S][2Co][T][F] = the stacked 1x1 convolutions (residual | residualmask) of the skip tensor,
 * z [S][Co][T][F] = padded gLN(act(deconv)); out = m * act(u) + (1 - m) * z, m = sigmoid(gLN(v)) */
int se_train_skip_fwd(const float *uv, const float *z, const float *nw, const float *nb, float *out, float *stats, int S, int Co, int T, int F, int act,
                      int eps_mode, void *stream);
int se_train_skip_bwd(const float *dout, const float *uv, const float *z, const float *nw, const float *nb, const float *stats, float *duv, float *dz,
                      float *dnw_part, float *dnb_part, float *dbias_part, int S, int Co, int T, int F, int act, int eps_mode, void *stream);
int se_train_add(float *dst, const float *src, int64_t n, void *stream);
int se_train_add3(float *dst, const float *a, const float *b, int64_t n, void *stream);   /* dst = a + b */
/* CRN_ELU deltas (CRN_ELU.py:194-252, 335-340): gated 1x1 pair + norm: tg [S][2C][T][F] = (conv_trans | conv_gated)(a);
 * y = gLN(t * sigmoid(g)); bwd -> dtg and slabs dw_part / db_part [S][C], dbias_part [S][2C].  se_train_elu_bwd: in place
 * da -> dy through a = ELU(y) from the saved activation, + the [S][C] slab of per-channel sums.  se_train_pre5: the 5-channel 5x5
 * frequency-dilated pre-conv blocks (C <= 8, vector ALU): mode 0 forward (act 2 stores ELU), 1 input gradient, 2 weight-gradient
 * slab [S][C*C*25]. */
int se_train_gate_fwd(const float *tg, const float *w, const float *b, float *y, int64_t yS, int64_t yC, int64_t yT, float *stats, int S, int C, int T, int F,
                      int eps_mode, void *stream);
int se_train_gate_bwd(const float *dy, int64_t dS, int64_t dC, int64_t dT, const float *tg, const float *w, const float *stats, float *dtg, float *dw_part,
                      float *db_part, float *dbias_part, int S, int C, int T, int F, int eps_mode, void *stream);
int se_train_elu_bwd(float *da, const float *act, float *dpre_part, int S, int C, int T, int F, void *stream);
int se_train_pre5(int mode, const float *x, const float *xprev, const float *w, const float *bias, const float *dy, float *out, int S, int C, int T, int F,
                  int fd, int act, void *stream);
/* h_{s-1} rows for the recurrent weight gradient (row addressing as se_train_gru_pseq_*) */
int se_train_gru_hprev(const float *out, const float *h0, float *hp, int B, int T, int H, int Tseg, int64_t ldN, int64_t ldB, void *stream);
/* se_train_conv with the weights in their checkpoint layout (element (row, col, kf, kt) at w[row*sCo + col*sCi + kf*3 + kt]); kind 3 =
 * 1x1 convolution.  ws: se_train_conv_ws_floats() floats of scratch for the staged arrangement. */
int se_train_conv_ws_floats(int kind, int Ci, int Co, int T, int Fi, int Fy, int dil);
int se_train_conv_w(int kind, const float *x, const float *xprev, const float *w, int64_t sCo, int64_t sCi, const float *bias, float *y, float *ws,
                    int B, int Ci, int Co, int T, int Fi, int Fy, int dil, int act, int Cy, int cy0, void *stream);
                    /* Cy > 0: y has Cy channels per stream and this launch writes channels [cy0, cy0 + Co) */
/* deterministic weight gradients: partial tiles per row split into ws[<= 64][...] (fold with se_train_colsum); ntap 15 or 1 */
int se_train_conv_wgrad_det(const float *G, const float *S, const float *Sprev, float *ws, int *nsplit_out, int B, int Ca, int Cb, int T, int Fm,
                            int Fs, int dil, int ntap, void *stream);
int se_train_gemm_tn_det(const float *A, const float *B, float *ws, int *nsplit_out, int64_t R, int Na, int Nb, void *stream);

/* ---- 8f-4: synthetic multi-microphone training data on the GPU (csrc/se_synth.hip) -------------------------------------
 * Replaces the reference's CPU/gpuRIR input pipeline for DP training: multichannel.py:37-103 (Single2Multi.simulate: shoebox
 * image-source RIRs, dry source * RIR), augment.py:29-77 (AddNoise.forward), data_c.py:236-250 (dynamic_mix, MAX_AMP guard).
 * All pointers are DEVICE pointers, all calls enqueue on `stream`; 0 or a negative se_status, se_synth_last_error() = message.
 *   se_synth_rir   rir[R][S][M][Lr]: image-source model of R rooms (room[R][3] sizes in m, beta[R][6] wall reflection
 *                  coefficients x0,x1,y0,y1,z0,z1, src[R][S][3], mic[R][M][3]); images -n/2 .. n/2-1 per axis (nx, ny, nz);
 *                  fractional delays by a Hann-windowed sinc of 8 ms; Lr <= 36864
 *   se_synth_fir   y[R][S][M][L] = x[R][S][L] * rir, truncated to L samples (gpuRIR.simulateTrajectory of a static source)
 *   se_synth_mix   the last source is the noise: mix[R][M][L], noise[R][M][L], absmax[R][M] (scratch) */
const char *se_synth_last_error(void);
int se_synth_rir(const float *room, const float *beta, const float *src, const float *mic, int R, int S, int M, int nx, int ny, int nz,
                 float fs, float c, int Lr, float *rir, void *stream);
/* optional second stage of gpuRIR (PARITY UNPINNED): for samples n >= tdiff[r] * fs the image-source response is replaced by a
 * stochastic tail, rms(last 10 ms before Tdiff) * exp(-6.9078 (n - Td) / (T60 fs)) * unit-variance logistic noise from a
 * counter-based hash of (seed, RIR index, n); tdiff / t60 [R] device arrays in seconds */
int se_synth_rir_tail(float *rir, const float *tdiff, const float *t60, int R, int S, int M, int Lr, float fs, uint32_t seed, void *stream);
int se_synth_fir(const float *x, const float *rir, int R, int S, int M, int64_t L, int Lr, float *y, void *stream);
int se_synth_mix(const float *y, const float *snr_db, int R, int S, int M, int64_t L, float max_amp, float *mix, float *noise, float *absmax,
                 void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SE_ENGINE_H */

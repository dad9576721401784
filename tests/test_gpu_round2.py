"""GPU parity tests added in round 2 (VERDICT r01 'what's weak' 1-3, 'missing' 4-6): state hand-over through
se_import_state, the student's distillation feature taps, FullSubNet at B = 256, the 3-term split-bf16 mode against the
REFERENCE goldens at north_star's tolerance, and B = 1024 (BASELINE config 5)."""
import numpy as np
import pytest
import torch

from conftest import FSN_FULL, FULL400, FULL512, STUDENT400, TINY, fsn_spec, rel_rms, spec_of, spec_of_variant
from speech_enhancement_mi_amd import synth

pytestmark = pytest.mark.gpu

TOL = 1e-4       # north_star: waveforms within 1e-4 RMS of the reference CPU path (relative form, see test_gpu_parity.py)
TOL_DB = 0.02    # north_star: SI-SDR within +-0.02 dB


def _cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def _engine(cfg, variant=0, precision=0, seed=0):
    from speech_enhancement_mi_amd import engine
    c = engine.make_config(cfg["num_channels"], cfg["num_freqs"], cfg["hidden"], cfg["segment_length"], cfg["num_layers"],
                           cfg["num_inputs"], cfg["kernel_size"], cfg["sample_rate"], cfg["win_length"], cfg["hop_length"], cfg["n_fft"],
                           variant=variant, precision=precision)
    e = engine.Engine(c, 0)
    e.load_state_dict(synth.make_state_dict(spec_of_variant(cfg, variant), seed=seed))
    return e


# ---- 8f-3: state hand-over -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg,variant", [(FULL400, 0), (TINY, 1)])
def test_import_state_into_fresh_engine(cfg, variant):
    """Export every state tensor of engine A after 3 windows, import into a freshly reset engine B: the next window is
    bit-equal (CRN.py:568-575 carried state: conv time buffers + GRU h; the variants add the preconv buffers).  Then one
    stream of A moves into batch slot 0 of a 1-stream engine C (a caller changes servers): same output within rounding."""
    B, L = 2, len(cfg["num_channels"])
    a, b, c = (_engine(cfg, variant, seed=3) for _ in range(3))
    mix, _ = synth.synth_utterances(B, 3200 * 4, 3, seed=61)
    a.reset(B)
    for k in range(3):
        a.step(_cuda(mix[:, :, 3200 * k:3200 * (k + 1)]))
    names = ["h"] + [f"buf{i}" for i in range(L)] + ([f"pbuf{i}" for i in range(3)] if variant else [])
    state = {n: a.export_state(n) for n in names}
    b.reset(B)
    for n in names:
        b.import_state(n, state[n])
    w = _cuda(mix[:, :, 3200 * 3:3200 * 4])
    ya, yb = a.step(w).cpu().numpy(), b.step(w).cpu().numpy()
    assert np.array_equal(ya, yb)
    for n in names:  # and the states after that window agree too
        assert np.array_equal(a.export_state(n), b.export_state(n)), n
    # stream 1 of the 2-stream batch -> slot 0 of a 1-stream engine
    c.reset(1)
    H = cfg["hidden"]
    for n in names:
        v = state[n]
        one = v.reshape(cfg["num_layers"], B, H)[:, 1:2] if n == "h" else v.reshape(B, -1)[1:2]
        c.import_state(n, np.ascontiguousarray(one))
    yc = c.step(_cuda(mix[1:2, :, 3200 * 3:3200 * 4])).cpu().numpy()
    assert rel_rms(yc[0], ya[1]) < 2e-6
    with pytest.raises(RuntimeError, match="needs"):
        c.import_state("h", np.zeros(3, np.float32))


# ---- a13: the student's distillation feature taps -----------------------------------------------------------------------
def test_student_feature_taps_golden(vgolden):
    """distillation_crn.TemporalCRN.realtime_process returns (pred, [5 pre-activation feature maps of [N*B, C, F, T]])
    (distillation_crn.py:467-477); fixtures student_tiny_feat0..4 = rows 2..5 (segments 1 and 2, both streams) from the genuine
    reference module."""
    from speech_enhancement_mi_amd.distillation_crn import TemporalCRN as Student
    m = Student(**TINY)
    sd = synth.make_state_dict(spec_of_variant(TINY, 2), seed=0)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m = m.cuda()
    mix, _ = synth.synth_utterances(2, 8000 + 4800, 3, seed=7)
    y, feats = m.realtime_process(_cuda(mix[..., :8000]))
    assert rel_rms(y.cpu().numpy(), vgolden["student_tiny_out"]) < TOL
    assert len(feats) == 5 == len(m.get_channel_num())
    assert all(f.is_cuda for f in feats)  # se_read_tap_dev: the feature maps never cross the host
    for i, (f, ch) in enumerate(zip(feats, m.get_channel_num())):
        ref = vgolden[f"student_tiny_feat{i}"]
        assert f.shape[1] == ch and tuple(f.shape[1:]) == ref.shape[1:], (i, f.shape, ref.shape)
        assert rel_rms(f[2:6].cpu().numpy(), ref) < 2e-5, i
    # the per-segment path carries state like the fused one: continuation matches the reference too
    y2, _ = m.realtime_process(_cuda(mix[..., 8000:]), True)
    assert rel_rms(y2.cpu().numpy(), vgolden["student_tiny_cont_out"]) < TOL
    # forward() returns the same five maps for one segment
    m.return_features = False
    y3, none = m.realtime_process(_cuda(mix[..., :8000]))
    assert none is None and rel_rms(y3.cpu().numpy(), vgolden["student_tiny_out"]) < TOL


# ---- config 3: FullSubNet at B = 256 ------------------------------------------------------------------------------------
def test_fsn_batch256_properties():
    """BASELINE configs[2] size (B*F = 51 456 sub-band rows): batch independence vs B = 4 (all norms / states are per stream),
    bit-repeatability after reset, finite output - the size-independent properties test_full_size_batch256_properties uses."""
    from speech_enhancement_mi_amd import engine
    cfg = FSN_FULL
    e = engine.FsnEngine(cfg["num_freqs"], cfg["num_mics"], cfg["fb_model_hidden_size"], cfg["sb_model_hidden_size"], cfg["num_layers"],
                         cfg["sb_num_neighbors"], cfg["fb_num_neighbors"], cfg["look_ahead"], cfg["sample_rate"], cfg["segment_length"],
                         cfg["win_length"], cfg["hop_length"], cfg["n_fft"])
    e.load_state_dict(synth.make_state_dict(fsn_spec(cfg), seed=0))
    L = 6400
    base, _ = synth.synth_utterances(4, L, 3, seed=23)
    big = np.ascontiguousarray(np.tile(base, (64, 1, 1)))
    y_big = e.realtime_process(_cuda(big)).cpu().numpy()
    assert y_big.shape == (256, L) and np.isfinite(y_big).all()
    y_small = e.realtime_process(_cuda(base)).cpu().numpy()
    for i in (0, 1, 2, 3, 101, 255):
        assert rel_rms(y_big[i], y_small[i % 4]) < 5e-6, i
    assert np.array_equal(e.realtime_process(_cuda(big)).cpu().numpy(), y_big)
    # prefix consistency: segments fully inside a prefix do not depend on what follows
    y_prefix = e.realtime_process(_cuda(base[..., :4800])).cpu().numpy()
    assert rel_rms(y_prefix[:, :3200], y_small[:, :3200]) < 5e-6


# ---- config 5: the 3-term split-bf16 mode against the REFERENCE goldens ---------------------------------------------------
@pytest.mark.parametrize("tag,cfg,variant,L", [("student_full400", STUDENT400, 2, 6400), ("elu_full400", FULL400, 1, 6400),
                                               ("full400", FULL400, 0, 8000), ("full512", FULL512, 0, 8000)])
def test_bf16x3_mode_vs_reference_goldens(golden, vgolden, tag, cfg, variant, L):
    """precision = 2 (hi*hi + hi*mid + mid*hi on the bf16 matrix cores, fp32 accumulation; recurrence, norms and storage
    fp32) against the outputs of the genuine reference modules, at north_star's bar: relative RMS < 1e-4 and SI-SDR of
    build and reference against the clean signal within 0.02 dB."""
    e = _engine(cfg, variant, precision=2)
    mix, clean = synth.synth_utterances(2, L + (3200 if tag == "full400" else 0), 3, seed=7)
    y = e.realtime_process(_cuda(mix[..., :L])).cpu().numpy()
    ref = (vgolden if variant else golden)[f"{tag}_out"]
    err = rel_rms(y, ref)
    assert err < TOL, err
    d = synth.si_sdr(clean[:, :L], y) - synth.si_sdr(clean[:, :L], ref)
    assert np.abs(d).max() < TOL_DB, d
    if tag == "full400":
        y2 = e.realtime_process(_cuda(mix[..., L:]), flag=True).cpu().numpy()
        assert rel_rms(y2, golden["full400_cont_out"]) < TOL


def test_student_batch1024_bf16x3():
    """BASELINE configs[4] size: the distilled student at B = 1024 in the fast mode.  Stream i of the big batch equals the
    same utterance in a batch of 4 run by the fp32-accurate engine to 1e-4 (parity of the mode is pinned against the
    reference goldens above; this checks the B = 1024 launch geometry), deterministic, finite."""
    e3, e6 = _engine(STUDENT400, 2, precision=2, seed=1), _engine(STUDENT400, 2, precision=0, seed=1)
    L = 9600
    base, clean = synth.synth_utterances(4, L, 3, seed=71)
    big = np.ascontiguousarray(np.tile(base, (256, 1, 1)))
    y_big = e3.realtime_process(_cuda(big)).cpu().numpy()
    assert y_big.shape == (1024, L) and np.isfinite(y_big).all()
    y_ref = e6.realtime_process(_cuda(base)).cpu().numpy()
    for i in (0, 1, 2, 3, 513, 1023):
        assert rel_rms(y_big[i], y_ref[i % 4]) < TOL, i
        assert abs(synth.si_sdr(clean[i % 4], y_big[i]) - synth.si_sdr(clean[i % 4], y_ref[i % 4])) < TOL_DB
    assert np.array_equal(e3.realtime_process(_cuda(big)).cpu().numpy(), y_big)


def test_set_precision_on_dropin_class(golden):
    from speech_enhancement_mi_amd import TemporalCRN
    m = TemporalCRN(**FULL400)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(spec_of(FULL400), seed=0).items()})
    m = m.cuda().set_precision("bf16x3")
    mix, _ = synth.synth_utterances(2, 8000 + 3200, 3, seed=7)  # the golden's input: first 8000 samples of this utterance
    y = m.realtime_process(_cuda(mix[..., :8000]))
    assert m._eng_precision == 2
    assert rel_rms(y.cpu().numpy(), golden["full400_out"]) < TOL
    with pytest.raises(ValueError):
        m.set_precision("int4")


def test_config_struct_size_matches_library():
    import ctypes
    from speech_enhancement_mi_amd import engine
    lib = engine.load_library()
    assert lib.se_config_size() == ctypes.sizeof(engine.SeConfig) == 4 * (1 + 8 + 11)
    assert lib.fsn_config_size() == ctypes.sizeof(engine.FsnConfig)


# ---- 8f-2: device-resident training loss ---------------------------------------------------------------------------------
def _loss_fixture():
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from loss_inputs import make_loss_inputs
    return np.load(os.path.join(ROOT, "tests", "golden", "loss_golden.npz")), make_loss_inputs()


def test_compute_loss_on_gpu_vs_reference_fixture():
    """TemporalCRN.compute_loss on GPU tensors: the 3-tuple and d loss / d pred against the genuine reference's values
    (tests/golden/make_golden_loss.py).  Everything stays on the device: the fused HIP SI-SNR kernel pair (se_loss_sisnr_*)
    and the batched STOI restatement; the result tensors live on the GPU."""
    from speech_enhancement_mi_amd import TemporalCRN
    lg, (clean, pred, lens) = _loss_fixture()
    m = TemporalCRN(**TINY).cuda()
    src = _cuda(clean)
    p = _cuda(pred).requires_grad_(True)
    loss, stoi, sisnr = m.compute_loss(src, p, torch.from_numpy(lens).cuda())
    assert loss.is_cuda and stoi.is_cuda and sisnr.is_cuda
    loss.backward()
    got = np.array([float(loss.detach()), float(stoi.detach()), float(sisnr.detach())])
    assert np.abs(got - lg["loss"]).max() < 1e-4, got
    g = p.grad.cpu().numpy()
    ref = lg["grad_pred_s5"]
    assert np.linalg.norm(g[:, ::5] - ref) / np.linalg.norm(ref) < 2e-3
    assert np.all(g[1, lens[1]:] == 0)


def test_sisnr_hip_kernels_vs_torch_autograd():
    """se_loss_sisnr_fwd / _bwd against the torch restatement of utility.cal_si_snr and its autograd gradient: ragged lengths,
    a high-SNR pair (|s - s_t| << |s_t|) and a negatively correlated pair."""
    from speech_enhancement_mi_amd import losses
    rng = np.random.default_rng(5)
    B, L = 6, 20011
    r = rng.standard_normal((B, L)).astype(np.float32)
    s = (r + 0.3 * rng.standard_normal((B, L))).astype(np.float32)
    s[1] = r[1] + 1e-3 * rng.standard_normal(L).astype(np.float32)   # ~60 dB
    s[2] = -0.5 * r[2] + 0.1 * rng.standard_normal(L).astype(np.float32)
    s[3] += 0.7                                                         # a DC offset is removed
    lens = np.array([L, L, 12345, 1, 777, 20000])
    s_cpu = torch.from_numpy(s).requires_grad_(True)
    v_ref = losses._cal_si_snr_torch(s_cpu, torch.from_numpy(r), torch.from_numpy(lens))
    v_ref.backward()
    s_gpu = _cuda(s).requires_grad_(True)
    v = losses.cal_si_snr(s_gpu, _cuda(r), torch.from_numpy(lens))
    v.backward()
    assert abs(float(v.detach()) - float(v_ref.detach())) < 2e-4 * max(1.0, abs(float(v_ref.detach())))
    g, g_ref = s_gpu.grad.cpu().numpy(), s_cpu.grad.numpy()
    for i in range(B):
        if lens[i] < 2:
            continue
        assert np.linalg.norm(g[i] - g_ref[i]) <= 2e-3 * np.linalg.norm(g_ref[i]) + 1e-12, i
        assert np.all(g[i, lens[i]:] == 0)


# ---- 8f-1: hand-written training kernels (forward + backward) vs torch autograd ---------------------------------------------
def _rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("Ci,Co,Fi,d", [(5, 16, 201, 1), (16, 32, 101, 2), (64, 128, 26, 8), (3, 4, 9, 1)])
def test_hip_conv_block_vs_autograd(Ci, Co, Fi, d):
    """TemporalConv2d convolution (CRN.py:314,327): forward, input gradient (a transposed-convolution launch), weight gradient
    (k_corr_wgrad) and bias gradient against torch conv2d + autograd, with a non-zero history buffer."""
    import torch.nn.functional as Fn
    from speech_enhancement_mi_amd import train_ops as K
    torch.manual_seed(Ci * 100 + d)
    B, T = 3, 21
    x = torch.randn(B, Ci, T, Fi, device="cuda", requires_grad=True)
    prev = torch.randn(B, Ci, T, Fi, device="cuda")
    w = (torch.randn(Co, Ci, 5, 3, device="cuda") * 0.1).requires_grad_(True)
    b = torch.randn(Co, device="cuda", requires_grad=True)
    y = K.conv_block(x, prev, w, b, d)
    # reference in the reference's own layout [B, C, F, T]: cat(buffer = last 2d time columns of prev, x) -> conv2d
    xr = x.detach().clone().requires_grad_(True)
    wr, br = w.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    inp = torch.cat([prev[:, :, -2 * d:].permute(0, 1, 3, 2), xr.permute(0, 1, 3, 2)], dim=-1)
    yr = Fn.conv2d(inp, wr, br, stride=(2, 1), padding=(2, 0), dilation=(1, d)).permute(0, 1, 3, 2)
    assert y.shape == yr.shape and _rel(y, yr) < 1e-5
    g = torch.randn_like(yr)
    y.backward(g)
    yr.backward(g)
    assert _rel(x.grad, xr.grad) < 1e-5 and _rel(w.grad, wr.grad) < 1e-4 and _rel(b.grad, br.grad) < 1e-5


@pytest.mark.parametrize("Ci,Co,Fi,d", [(128, 64, 13, 1), (32, 16, 51, 4), (16, 2, 101, 8), (8, 4, 7, 2)])
def test_hip_deconv_block_vs_autograd(Ci, Co, Fi, d):
    """TemporalConvTranspose2d convolution keeping the last T columns (CRN.py:369,383) against conv_transpose2d + autograd."""
    import torch.nn.functional as Fn
    from speech_enhancement_mi_amd import train_ops as K
    torch.manual_seed(Ci + d)
    B, T = 2, 21
    x = torch.randn(B, Ci, T, Fi, device="cuda", requires_grad=True)
    w = (torch.randn(Ci, Co, 5, 3, device="cuda") * 0.1).requires_grad_(True)
    b = torch.randn(Co, device="cuda", requires_grad=True)
    y = K.deconv_block(x, w, b, d)
    xr, wr, br = (t.detach().clone().requires_grad_(True) for t in (x, w, b))
    yr = Fn.conv_transpose2d(xr.permute(0, 1, 3, 2), wr, br, stride=(2, 1), padding=(2, 0), dilation=(1, d))[..., -T:].permute(0, 1, 3, 2)
    assert y.shape == yr.shape and _rel(y, yr) < 1e-5
    g = torch.randn_like(yr)
    y.backward(g)
    yr.backward(g)
    assert _rel(x.grad, xr.grad) < 1e-5 and _rel(w.grad, wr.grad) < 1e-4 and _rel(b.grad, br.grad) < 1e-5


@pytest.mark.parametrize("In,H,B", [(1664, 512, 4), (104, 16, 3)])
def test_hip_gru_layer_and_linear_vs_autograd(In, H, B):
    """One GRU layer over T = 21 steps with a carried state (forward + BPTT) and the fc layer, against nn.GRU / linear autograd."""
    from speech_enhancement_mi_amd import train_ops as K
    torch.manual_seed(H)
    T = 21
    ref = torch.nn.GRU(In, H, 1, batch_first=True).cuda()
    x = torch.randn(B, T, In, device="cuda") * 0.5
    h0 = torch.randn(B, H, device="cuda") * 0.5
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ps = [p.detach().clone().requires_grad_(True) for p in (ref.weight_ih_l0, ref.weight_hh_l0, ref.bias_ih_l0, ref.bias_hh_l0)]
    out, hT = K.gru_layer(xa, h0, *ps)
    out_r, hT_r = ref(xb, h0[None])
    assert _rel(out, out_r) < 1e-5 and _rel(hT, hT_r[0]) < 1e-5
    g = torch.randn_like(out_r)
    out.backward(g)
    out_r.backward(g)
    assert _rel(xa.grad, xb.grad) < 2e-5
    for p, q in zip(ps, (ref.weight_ih_l0, ref.weight_hh_l0, ref.bias_ih_l0, ref.bias_hh_l0)):
        assert _rel(p.grad, q.grad) < 1e-4
    w, b = torch.randn(37, In, device="cuda", requires_grad=True), torch.randn(37, device="cuda", requires_grad=True)
    y = K.linear(xa.detach().requires_grad_(True), w, b)
    yr = torch.nn.functional.linear(x, w.detach(), b.detach())
    assert _rel(y, yr) < 1e-5


@pytest.mark.parametrize("cfgname", ["tiny", "full400"])
def test_train_step_on_gpu_matches_cpu_autograd(cfgname):
    """BASELINE config 4 on hardware: one training forward/backward (full loss: 0.7 * STOI + 0.3 * (-SI-SNR)) with the
    hand-written kernels on the GPU against the torch-autograd run of the same model on the CPU: loss and every parameter
    gradient (<= 1e-4 relative on the whole flat gradient, per tensor <= 2e-3 of the largest tensor norm)."""
    from speech_enhancement_mi_amd.training import FlatBucket, TrainableCRN
    cfg = TINY if cfgname == "tiny" else FULL400
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(spec_of(cfg), seed=2).items()}
    mix, clean = synth.synth_utterances(2, 8000 if cfgname == "tiny" else 4800, 3, seed=81)
    L = mix.shape[-1]
    lens = torch.tensor([L, L - 700])

    def run(device, hip):
        m = TrainableCRN(**cfg)
        m.load_state_dict(sd)
        m = m.to(device).use_hip_kernels(hip)
        bucket = FlatBucket(list(m.parameters()))
        pred = m.realtime_process_train(torch.from_numpy(mix).to(device))
        loss = m.compute_loss(torch.from_numpy(clean).to(device), pred, lens.to(device))[0]
        loss.backward()
        return float(loss.detach()), bucket.flat.detach().cpu(), [(n, p.grad.detach().cpu()) for n, p in m.named_parameters()]

    l_cpu, g_cpu, named_cpu = run("cpu", False)
    l_hip, g_hip, named_hip = run("cuda", True)
    l_gpu, g_gpu, _ = run("cuda", False)
    assert abs(l_hip - l_cpu) < 1e-4 * max(1.0, abs(l_cpu))
    assert _rel(g_gpu, g_cpu) < 2e-4                     # torch ops on the GPU vs the CPU: the rounding floor of this comparison
    assert _rel(g_hip, g_cpu) < 2e-4, _rel(g_hip, g_cpu)  # hand-written kernels vs CPU autograd
    gmax = max(float(g.norm()) for _, g in named_cpu)
    for (n, a), (_, b) in zip(named_hip, named_cpu):
        assert float((a - b).norm()) < 2e-3 * gmax, n


def test_plane_gemm_matches_first_generation_gemm(monkeypatch):
    """k_gemm_p (split-bf16 A planes written by the producers, LDS-DMA ring) against k_gemm_x (fp32 A split in the
    kernel) on the whole bottleneck: same products, different summation grouping -> fp32 round-off only.  Also the
    'enc{L-1}' tap, which the plane path reconstructs from the GEMM's A planes."""
    from test_gpu_parity import FULL512, _engine, _cuda, rel_rms
    mix, _ = synth.synth_utterances(4, 9600, 3, seed=77)
    x = _cuda(mix)
    last = "enc%d" % (len(FULL512["num_channels"]) - 1)
    outs, taps = {}, {}
    for flag in ("1", "0"):
        monkeypatch.setenv("SE_GEMM_P", flag)
        e = _engine(FULL512, seed=9)
        outs[flag] = e.realtime_process(x).cpu().numpy()
        e.reset(4)
        e.step(x[:, :, :3200].contiguous())
        taps[flag] = np.asarray(e.read_tap(last))
    assert rel_rms(outs["1"], outs["0"]) < 2e-6
    assert taps["1"].shape == taps["0"].shape and rel_rms(taps["1"], taps["0"]) < 1e-6


@pytest.mark.parametrize("variant", [1, 2])
def test_preconv_blocks_on_plane_path_match_vector_kernel(monkeypatch, variant):
    """CRN_ELU / student: the three 5x5 pre-conv blocks on k_conv_p<25,...> (default only with fp16 operands, SE_PRE_P=1 forces
    it) against the fp32 vector-ALU kernel, fp32-accurate mode: outputs, the pre-chain output tap and the exported
    4-frame pre-conv buffers agree to fp32 round-off (CRN_ELU.py:335-340, 375-376)."""
    from test_gpu_parity import FULL400, STUDENT400, _engine_v, _cuda, rel_rms
    cfg = FULL400 if variant == 1 else STUDENT400
    mix, _ = synth.synth_utterances(3, 9600, 3, seed=91)
    x = _cuda(mix)
    res = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("SE_PRE_P", flag)
        e = _engine_v(cfg, variant, seed=4)
        y = e.realtime_process(x).cpu().numpy()
        e.reset(3)
        for k in range(2):
            e.step(x[:, :, 3200 * k:3200 * (k + 1)].contiguous())
        res[flag] = (y, np.asarray(e.read_tap("feat")), [np.asarray(e.export_state(f"pbuf{i}")) for i in range(3)])
    assert rel_rms(res["1"][0], res["0"][0]) < 2e-6
    assert rel_rms(res["1"][1], res["0"][1]) < 1e-6
    for a, b in zip(res["1"][2], res["0"][2]):
        assert a.shape == b.shape and rel_rms(a, b) < 1e-6


def test_fused_training_forward_equals_per_segment_autograd():
    """The fused training path runs every layer once over N x B segment streams (segment-major; history = the same tensor one
    slab earlier; GRU state carried segment by segment inside one persistent launch).  Same outputs, continuation state and
    gradients as the torch-autograd path that walks one segment at a time (CRN.py:577-586), including a flag=True continuation."""
    from speech_enhancement_mi_amd.training import TrainableCRN
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(spec_of(FULL400), seed=6).items()}
    mix, clean = synth.synth_utterances(2, 11200, 3, seed=83)
    res = {}
    for hip in (True, False):
        m = TrainableCRN(**FULL400)
        m.load_state_dict(sd)
        m = m.cuda().use_hip_kernels(hip)
        x = torch.from_numpy(mix).cuda()
        y1 = m.realtime_process_train(x[..., :6400])
        y2 = m.realtime_process_train(x[..., 6400:].contiguous(), flag=True)
        loss = (y1 ** 2).mean() + (y2 * torch.from_numpy(clean[:, :y2.shape[-1]]).cuda()).mean()
        loss.backward()
        res[hip] = (y1.detach().cpu(), y2.detach().cpu(), torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).flatten() for p in m.parameters()]).cpu())
    assert _rel(res[True][0], res[False][0]) < 1e-5 and _rel(res[True][1], res[False][1]) < 1e-5
    # both are fp32 implementations with different kinks (ReLU / +-9.9 clamp decisions within rounding of the threshold differ for
    # single elements): measured 4e-4 on the flat gradient here, 5e-6 at the bench shape against float64 autograd (test_gpu_round3.py)
    assert _rel(res[True][2], res[False][2]) < 2e-3


@pytest.mark.parametrize("pre_p", ["0", "1"])
def test_reset_single_stream_student_plane_path(monkeypatch, pre_p):
    """se_reset_stream on the student (pre-conv rings of the plane path with SE_PRE_P=1, fp32 pre-conv buffers otherwise):
    after resetting stream 1 only, stream 1 behaves like a freshly reset engine and stream 0 like an untouched one."""
    from test_gpu_parity import STUDENT400, _engine_v, _cuda, rel_rms
    monkeypatch.setenv("SE_PRE_P", pre_p)
    e, e_fresh, e_cont = (_engine_v(STUDENT400, 2, seed=3) for _ in range(3))
    a, _ = synth.synth_utterances(2, 3200 * 3, 3, seed=41)
    bnew, _ = synth.synth_utterances(1, 3200 * 2, 3, seed=42)
    e.reset(2); e_cont.reset(2); e_fresh.reset(1)
    for k in range(3):
        w = _cuda(a[:, :, 3200 * k:3200 * (k + 1)])
        e.step(w); e_cont.step(w)
    e.reset_stream(1)
    for k in range(2):
        w0 = a[:1, :, 3200 * k:3200 * (k + 1)]
        w1 = bnew[:, :, 3200 * k:3200 * (k + 1)]
        y = e.step(_cuda(np.concatenate([w0, w1], 0))).cpu().numpy()
        y_cont = e_cont.step(_cuda(np.concatenate([w0, w0], 0))).cpu().numpy()
        y_fresh = e_fresh.step(_cuda(w1)).cpu().numpy()
        assert rel_rms(y[1], y_fresh[0]) < 2e-6, k
        assert np.array_equal(y[0], y_cont[0]), k


def test_small_batch_routes_match_large_batch_kernels(monkeypatch):
    """Small batches route the bottleneck GEMMs to the skinny fp32 kernel and the skip gate to the two-launch k_conv_p form
    (se_reset decides by batch); the large-batch kernels (k_gemm_p, k_skip_p) forced onto the same small batch must agree
    to fp32 round-off, end to end and on the decoder taps."""
    from test_gpu_parity import FULL512, _engine, _cuda, rel_rms
    mix, _ = synth.synth_utterances(3, 9600, 3, seed=55)
    x = _cuda(mix)
    res = {}
    for tag, rows, minb in (("small", "100000", "100000"), ("large", "0", "1")):
        monkeypatch.setenv("SE_GEMM_SKINNY_ROWS", rows)
        monkeypatch.setenv("SE_SKIP_MIN_BATCH", minb)
        e = _engine(FULL512, seed=12)
        y = e.realtime_process(x).cpu().numpy()
        e.reset(3)
        e.step(x[:, :, :3200].contiguous())
        res[tag] = (y, np.asarray(e.read_tap("gru")), np.asarray(e.read_tap("dec0")), np.asarray(e.read_tap("dec2")))
    for a, b in zip(res["small"], res["large"]):
        assert a.shape == b.shape and rel_rms(a, b) < 2e-6

"""GPU tests added in round 3: the persistent training GRU (one launch per layer and direction), the weight-reload route
fix, wide training batches, config 5 in its named dtype at size, and the
hand-written norm / pointwise / signal kernels of the training step."""
import numpy as np
import pytest
import torch

from conftest import FULL400, FULL512, STUDENT400, TINY, rel_rms, spec_of, spec_of_variant
from speech_enhancement_mi_amd import synth

pytestmark = pytest.mark.gpu


def _cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def _rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


def _engine(cfg, variant=0, precision=0, seed=0):
    from speech_enhancement_mi_amd import engine
    c = engine.make_config(cfg["num_channels"], cfg["num_freqs"], cfg["hidden"], cfg["segment_length"], cfg["num_layers"],
                           cfg["num_inputs"], cfg["kernel_size"], cfg["sample_rate"], cfg["win_length"], cfg["hop_length"], cfg["n_fft"],
                           variant=variant, precision=precision)
    e = engine.Engine(c, 0)
    e.load_state_dict(synth.make_state_dict(spec_of_variant(cfg, variant), seed=seed))
    return e


# ---- a weight reload must not switch the plane GEMM route back on for a small batch ---------------------
def test_weight_reload_keeps_small_batch_gemm_route():
    """reset(B=1) puts the bottleneck GEMMs on the skinny route (no plane buffers allocated); se_load_param clears
    weights_ready and the next step re-plans.  The re-plan used to set gemm_p from the capability alone -> k_gemm_p on null
    plane buffers.  Now: step, reload the same weights, step again == a fresh engine fed the same two windows."""
    sd = synth.make_state_dict(spec_of(FULL512), seed=5)
    mix, _ = synth.synth_utterances(1, 6400, 3, seed=13)
    w0, w1 = _cuda(mix[:, :, :3200]), _cuda(mix[:, :, 3200:])
    a, b = _engine(FULL512, seed=5), _engine(FULL512, seed=5)
    a.reset(1); b.reset(1)
    ya0 = a.step(w0).cpu().numpy()
    a.load_state_dict(sd)          # optimizer step / load_state_dict on the drop-in class does exactly this
    ya1 = a.step(w1).cpu().numpy()
    yb0 = b.step(w0).cpu().numpy()
    yb1 = b.step(w1).cpu().numpy()
    assert np.array_equal(ya0, yb0) and np.array_equal(ya1, yb1)
    # and through the drop-in class: forward at B <= 38, load_state_dict, forward again
    from speech_enhancement_mi_amd import TemporalCRN
    m = TemporalCRN(**FULL512)
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    m.load_state_dict(tsd)
    m = m.cuda()
    y1 = m.realtime_process(_cuda(mix))
    m.load_state_dict(tsd)
    y2 = m.realtime_process(_cuda(mix))
    assert torch.equal(y1, y2) and bool(torch.isfinite(y2).all())


def test_dbg_skip_knob_is_refused_by_the_shipped_library(monkeypatch):
    monkeypatch.setenv("SE_DBG_SKIP", "1")
    with pytest.raises(RuntimeError, match="SE_DEBUG_KNOBS"):
        _engine(TINY)


# ---- the persistent training GRU -----------------------------------------------------------------------------------------------
def _ref_gru_segments(ref, x, h0, seg_len):
    """nn.GRU run segment by segment with a detached carry: exactly what realtime_process does (CRN.py:281, 577-586)."""
    outs, h = [], h0[None]
    T = x.shape[1]
    step = seg_len if seg_len > 0 else T
    for t0 in range(0, T, step):
        o, h = ref(x[:, t0:t0 + step], h.detach() if t0 else h)
        outs.append(o)
    return torch.cat(outs, dim=1), h[0]


@pytest.mark.parametrize("In,H,B,T,seg_len", [(64, 512, 4, 42, 21), (64, 512, 17, 30, 10), (48, 512, 32, 21, 7), (40, 128, 8, 25, 0),
                                              (24, 16, 3, 21, 21), (32, 256, 40, 12, 4), (24, 144, 20, 9, 3)])
def test_persistent_gru_layer_vs_autograd(In, H, B, T, seg_len):
    """gru_layer (k_gru_pseq_fwd / _bwd: register-resident W_hh slices, sc1 state exchange) against nn.GRU + autograd run
    segment by segment: outputs, final state, input gradient and all four parameter gradients.  B = 17..32 takes the two-row-tile
    instances, B = 40 two launches (groups of <= 32 streams), H = 144 the per-step fallback (groups of <= 16: the wide-batch
    route ADVICE r02 asked a test for)."""
    from speech_enhancement_mi_amd import train_ops as K
    torch.manual_seed(H + B)
    ref = torch.nn.GRU(In, H, 1, batch_first=True).cuda()
    x = torch.randn(B, T, In, device="cuda") * 0.5
    h0 = torch.randn(B, H, device="cuda") * 0.5
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ps = [p.detach().clone().requires_grad_(True) for p in (ref.weight_ih_l0, ref.weight_hh_l0, ref.bias_ih_l0, ref.bias_hh_l0)]
    out, hT = K.gru_layer(xa, h0, *ps, seg_len=seg_len)
    out_r, hT_r = _ref_gru_segments(ref, xb, h0, seg_len)
    assert _rel(out, out_r) < 1e-5 and _rel(hT, hT_r) < 1e-5
    g = torch.randn_like(out_r)
    gh = torch.randn_like(hT_r)
    (out * g).sum().add((hT * gh).sum()).backward()
    (out_r * g).sum().add((hT_r * gh).sum()).backward()
    K.pseq_check()
    assert _rel(xa.grad, xb.grad) < 2e-5
    for p, q in zip(ps, (ref.weight_ih_l0, ref.weight_hh_l0, ref.bias_ih_l0, ref.bias_hh_l0)):
        assert _rel(p.grad, q.grad) < 1e-4


def test_persistent_gru_is_deterministic_and_batch_independent():
    """Same inputs -> bit-identical outputs across launches (fixed summation order: no atomics in the data path); stream b of
    a 32-stream launch equals the same stream in a 4-stream launch to rounding (different row tile -> same products)."""
    from speech_enhancement_mi_amd import train_ops as K
    torch.manual_seed(3)
    In, H, T = 32, 512, 63
    ps = [torch.randn(3 * H, In, device="cuda") * 0.1, torch.randn(3 * H, H, device="cuda") * 0.05, torch.randn(3 * H, device="cuda") * 0.1,
          torch.randn(3 * H, device="cuda") * 0.1]
    x = torch.randn(32, T, In, device="cuda")
    h0 = torch.randn(32, H, device="cuda") * 0.3
    with torch.no_grad():
        o1, h1 = K.gru_layer(x, h0, *ps)
        o2, h2 = K.gru_layer(x, h0, *ps)
        o4, h4 = K.gru_layer(x[5:9].contiguous(), h0[5:9].contiguous(), *ps)
    K.pseq_check()
    assert torch.equal(o1, o2) and torch.equal(h1, h2)
    assert _rel(o1[5:9], o4) < 1e-6 and _rel(h1[5:9], h4) < 1e-6

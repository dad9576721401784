"""Drop-in for the reference's `FullSubNet` (fullsubnet.py:685-987): same constructor kwargs (config.yaml:153-172) and
state_dict keys (`fb_model.*`, `sb_model.*`), `realtime_process(mixture, source, flag, train)` and the 6-argument `compute_loss`.
Compute runs on the MI355X engine through the fsn_* / se_sig_* C ABI; the torch sub-modules only own the parameters.

  * train=False (what both reference trainers and predict_fullsubnet.py call: train_fullsubnet.py:138,151; predict_fullsubnet.py:75):
    the streaming engine, one forward per 3200-sample window.  Returns `pred` (source=None) or `(pred, None, None, None)`.
  * train=True (fullsubnet.py:921-927): ONE forward over all N*T frames of the chunk (one CumLayerNorm update, the LSTMs run
    through the N*T frames without a per-window seam), then mask / iSTFT / over_add per window; returns the reference's
    4-tuple `(pred_source, pred_crm [N,B,2,F,T], s [N,B,2,F,T], x [N,B,2,F,T])`.  Forward only: the tensors carry no
    autograd graph (no LSTM backward kernels exist; FullSubNet training is outside SURVEY.md 8's rows)."""
from __future__ import annotations

import torch
from torch import nn

from . import engine as _engine


class _Seq(nn.Module):  # SequenceModel parameter holder, fullsubnet.py:209-273
    def __init__(self, input_size, output_size, hidden_size, num_layers):
        super().__init__()
        self.sequence_model = nn.LSTM(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers, batch_first=True,
                                      bidirectional=False)
        self.fc_output_layer = nn.Linear(hidden_size, output_size)


class FullSubNet(nn.Module):
    def __init__(self, num_freqs, look_ahead, sequence_model, fb_num_neighbors, sb_num_neighbors, fb_output_activate_function,
                 sb_output_activate_function, fb_model_hidden_size, sb_model_hidden_size, num_mics, norm_type="offline_laplace_norm",
                 num_groups_in_drop_band=2, num_layers=2, weight_init=True, sample_rate=16000, segment_length=400, win_length=20,
                 hop_length=10, n_fft=320):
        super().__init__()
        if sequence_model != "LSTM" or fb_output_activate_function != "ReLU" or sb_output_activate_function:
            raise NotImplementedError("the engine implements the reference configuration: LSTM, ReLU full-band output, linear sub-band output")
        if fb_num_neighbors != 0 or look_ahead != 0:
            raise NotImplementedError("fb_num_neighbors = 0 and look_ahead = 0 only (config.yaml:154-157)")
        self.fb_model = _Seq(num_freqs * num_mics, num_freqs, fb_model_hidden_size, num_layers)
        self.sb_model = _Seq((sb_num_neighbors * 2 + 1) + (fb_num_neighbors * 2 + 1), 2, sb_model_hidden_size, num_layers)
        self.num_freqs, self.num_mics, self.segment_length = num_freqs, num_mics, segment_length
        self._args = dict(num_freqs=num_freqs, num_mics=num_mics, fb_hidden=fb_model_hidden_size, sb_hidden=sb_model_hidden_size,
                          num_layers=num_layers, sb_neighbors=sb_num_neighbors, fb_neighbors=fb_num_neighbors, look_ahead=look_ahead,
                          sample_rate=sample_rate, segment_length=segment_length, win_length=win_length, hop_length=hop_length, n_fft=n_fft)
        self._eng = None
        self._eng_device = None
        self._versions = None
        self._precision = 0
        self._long = {}  # train=True engines by frames per chunk

    _PRECISIONS = {"fp32": 0, "f32": 0, "bf16x3": 2}

    def set_precision(self, mode):
        """Extension over the reference: "fp32" (default) = fp32-accurate LSTM contractions (6-term split-bf16 MFMA); "bf16x3" =
        3-term split-bf16 (inside the 1e-4 RMS / 0.02 dB parity bar, half the matrix work)."""
        if mode not in self._PRECISIONS:
            raise ValueError(f"precision {mode!r} not in {sorted(self._PRECISIONS)}")
        if self._PRECISIONS[mode] != self._precision:
            self._precision = self._PRECISIONS[mode]
            self._eng, self._long = None, {}
        return self

    def _engine_for(self, t):
        if not t.is_cuda:
            raise RuntimeError("FullSubNet runs on the MI355X engine only (no CPU fallback)")
        dev = t.device.index if t.device.index is not None else torch.cuda.current_device()
        if self._eng is None or self._eng_device != dev:
            self._eng = _engine.FsnEngine(device=dev, precision=self._precision, **self._args)
            self._eng_device = dev
            self._versions = None
            self._long = {}
        versions = tuple(p._version for p in self.parameters())
        if versions != self._versions:
            self._eng.load_state_dict(dict(self.state_dict()))
            self._versions = versions
        return self._eng

    def reset_state(self, batch_size, dtype=None, device=None):
        if self._eng is not None:
            self._eng.reset(batch_size)

    def forward(self, noisy_complex):
        eng = self._engine_for(noisy_complex)
        if eng.batch != noisy_complex.shape[0]:
            eng.reset(noisy_complex.shape[0])
        return eng.forward(noisy_complex.contiguous().float())

    def realtime_process(self, mixture, source=None, flag=False, train=True):
        if train:
            return self._realtime_single_pass(mixture, source, flag)
        eng = self._engine_for(mixture)
        pred = eng.realtime_process(mixture.contiguous().float(), flag=bool(flag))
        return pred if source is None else (pred, None, None, None)

    def _realtime_single_pass(self, mixture, source, flag):
        """train=True (fullsubnet.py:921-927): xf = all N windows' frames side by side [B, 2M, F, N*T] -> ONE forward."""
        import ctypes as C
        from . import train_net as N_, train_ops as K
        if flag:
            raise NotImplementedError("train=True continues a previous chunk only in the reference's autograd loop; the engine path takes flag=False")
        if source is None:
            raise ValueError("train=True needs `source` (the reference concatenates the pad to it, fullsubnet.py:908)")
        eng = self._engine_for(mixture)  # uploads / refreshes the weights
        a = self._args
        dev = mixture.device
        mixture, source = mixture.contiguous().float(), source.contiguous().float()
        B, M, L = mixture.shape
        Ks, P = self.segment_length, self.segment_length // 2
        hop = int(round(a["sample_rate"] / 1000.0 * a["hop_length"]))
        win = int(round(a["sample_rate"] / 1000.0 * a["win_length"]))
        T, F = 1 + Ks // hop, self.num_freqs
        Lp = L + P
        gap = Ks - (P + Lp % Ks) % Ks
        N = 2 * (Lp + gap + P) // Ks
        S = N * B
        lib = K._lib()
        sig = N_._sig(dev, a["n_fft"], win, hop, Ks)
        spec = torch.empty(N, B * M, T, F, 2, device=dev)
        K._chk(lib.se_sig_stft(sig, mixture.data_ptr(), B, M, L, -2 * P, P, N, spec.data_ptr(), K._st()))
        sspec = torch.empty(N, B * source.shape[1], T, F, 2, device=dev)
        K._chk(lib.se_sig_stft(sig, source.data_ptr(), B, source.shape[1], L, -2 * P, P, N, sspec.data_ptr(), K._st()))
        xf = spec.view(N, B, M, T, F, 2).permute(1, 5, 2, 4, 0, 3).reshape(B, 2 * M, F, N * T).contiguous()
        long = self._long.get(N * T)
        if long is None:
            args = dict(a, segment_length=hop * (N * T - 1))
            long = self._long[N * T] = _engine.FsnEngine(device=self._eng_device, precision=self._precision, **args)
            long._versions = None
        if getattr(long, "_versions", None) != self._versions:
            long.load_state_dict(dict(self.state_dict()))
            long._versions = self._versions
        long.reset(B)
        crm_long = long.forward(xf)                                               # [B, 2, F, N*T]
        pred_crm = crm_long.view(B, 2, F, N, T).permute(3, 0, 1, 2, 4).contiguous()  # [N, B, 2, F, T]
        xm = pred_crm.permute(0, 1, 2, 4, 3).reshape(S, 2, T, F).contiguous()
        Y = torch.empty(S, T, F, 2, device=dev)
        K._chk(lib.se_train_mask_fwd(xm.data_ptr(), spec.data_ptr(), Y.data_ptr(), S, M, T, F, K._st()))
        yseg = torch.empty(S, Ks, device=dev)
        K._chk(lib.se_sig_istft(sig, Y.data_ptr(), S, yseg.data_ptr(), K._st()))
        pred = torch.empty(B, L, device=dev)
        K._chk(lib.se_train_ola_fwd(sig, yseg.data_ptr(), pred.data_ptr(), B, L, P, K._st()))
        x0 = spec.view(N, B, M, T, F, 2)[:, :, 0].permute(0, 1, 4, 3, 2).contiguous()    # [N, B, 2, F, T]
        s0 = sspec.view(N, B, -1, T, F, 2)[:, :, 0].permute(0, 1, 4, 3, 2).contiguous()
        return pred, pred_crm, s0, x0

    def compute_loss(self, source, pred_source, xf, sf, cIRM, length):
        """fullsubnet.py:964-986: loss = 0.7 * stoi_loss + 0.3 * (-SI-SNR) of (source, pred_source, length); the spectral arguments
        (xf, sf, cIRM) are accepted and unused, exactly like the reference (its spectral terms are commented out)."""
        from .losses import compute_loss
        return compute_loss(source, pred_source, length)

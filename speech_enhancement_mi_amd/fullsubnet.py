"""Drop-in for the reference's `FullSubNet` (fullsubnet.py:685-987): same constructor kwargs (config.yaml:153-172) and
state_dict keys (`fb_model.*`, `sb_model.*`); inference path `realtime_process(mixture, source, flag, train=False)`.
Compute runs on the MI355X engine through the fsn_* C ABI; the torch sub-modules only own the parameters.
The reference returns `(pred_source, pred_crm, s, x)`; `predict_fullsubnet.py:75` keeps only the first element and this
class returns `None` for the other three (they exist for the training loss, which is out of scope)."""
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

    def _engine_for(self, t):
        if not t.is_cuda:
            raise RuntimeError("FullSubNet runs on the MI355X engine only (no CPU fallback)")
        dev = t.device.index if t.device.index is not None else torch.cuda.current_device()
        if self._eng is None or self._eng_device != dev:
            self._eng = _engine.FsnEngine(device=dev, **self._args)
            self._eng_device = dev
            self._versions = None
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
        # train=True in the reference runs one forward over all N*T frames of the chunk for back-propagation
        # (fullsubnet.py:921-927); both reference trainers and predict call train=False, which is what the engine implements
        if train:
            raise NotImplementedError("FullSubNet.realtime_process(train=True) - one forward over all N*T frames for back-propagation "
                                      "(fullsubnet.py:921-927) - is not built; both reference trainers and predict_fullsubnet.py call "
                                      "train=False (train_fullsubnet.py:138,151; predict_fullsubnet.py:75)")
        eng = self._engine_for(mixture)
        pred = eng.realtime_process(mixture.contiguous().float(), flag=bool(flag))
        return pred if source is None else (pred, None, None, None)

#!/usr/bin/env python3
"""Loss fixtures (tests/golden/loss_golden.npz) from the GENUINE reference loss code, run in this container.

    python tests/golden/make_golden_loss.py        (needs /root/reference; the GPU box never has it)

`utility.stoi_loss`, `utility.cal_si_snr`, `utility.thirdoct`, `utility.removeSilentFrames` and
`TemporalCRN.compute_loss` run from the reference's source files unmodified.  stoi_loss calls two torchaudio==0.7.2
transforms that are absent from the reference tree and from this image, so they are supplied here:
  * torchaudio.transforms.Resample  -> the reference's OWN in-tree Kaldi-style sinc resampler (augment.py:234-545, the
    speechbrain copy of torchaudio.compliance.kaldi.resample_waveform, which is what torchaudio 0.7.2's Resample calls)
  * torchaudio.transforms.Spectrogram -> a restatement over torch.stft (periodic Hann(win_length) centred in n_fft,
    centre reflect padding, |X|^power)
=> "parity unpinned at the torchaudio boundary", exactly like the speechbrain STFT of make_golden.py.
Only DATA (inputs and outputs) is written; no reference source text is copied."""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (shared import placeholders)

from speech_enhancement_mi_amd import synth  # noqa: E402


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    mg._install_import_placeholders()
    # placeholders for the imports at the top of the reference's augment.py (none of them is touched by Resample)
    for name, attrs in (("speechbrain.dataio", []), ("speechbrain.dataio.dataio", ["read_audio"]),
                        ("speechbrain.processing.signal_processing", ["compute_amplitude", "dB_to_amplitude", "convolve1d", "notch_filter", "reverberate"])):
        m = types.ModuleType(name)
        for a in attrs:
            setattr(m, a, lambda *x, **k: (_ for _ in ()).throw(RuntimeError("placeholder")))
        sys.modules[name] = m
    import augment  # the reference's in-tree resampler
    import torchaudio
    import torchaudio.transforms as TT

    class Resample(torch.nn.Module):  # torchaudio.transforms.Resample(orig, new)(waveform[time]) -> [time']
        def __init__(self, orig_freq=16000, new_freq=16000):
            super().__init__()
            self.r = augment.Resample(orig_freq, new_freq)

        def forward(self, x):
            return self.r(x.unsqueeze(0)).squeeze(0)

    class Spectrogram(torch.nn.Module):
        def __init__(self, n_fft=400, win_length=None, hop_length=None, power=2.0):
            super().__init__()
            self.n_fft, self.win, self.hop, self.power = n_fft, win_length or n_fft, hop_length or (win_length or n_fft) // 2, power
            self.window = torch.hann_window(self.win)

        def forward(self, x):
            s = torch.stft(x, self.n_fft, self.hop, self.win, self.window.to(x.device), center=True, pad_mode="reflect",
                           normalized=False, onesided=True, return_complex=True)
            return (s.real ** 2 + s.imag ** 2).pow(0.5 * self.power)

    TT.Resample, TT.Spectrogram = Resample, Spectrogram
    torchaudio.transforms = TT
    import CRN
    import utility

    out = {}
    # ---- inputs: synthetic clean speech + a degraded copy, ragged lengths, one utterance with long pauses -------------
    from loss_inputs import make_loss_inputs
    clean, pred, lens = make_loss_inputs()
    src_t, pred_t, len_t = torch.from_numpy(clean), torch.from_numpy(pred).requires_grad_(True), torch.from_numpy(lens)

    # resampler and spectrogram outputs (the unpinned boundary itself) for the oracle's own check
    r = Resample(16000, 10000)
    out["resample_out"] = mg.t2n(r(torch.from_numpy(clean[0, :4001])))
    out["resample_out_len"] = np.array([r(torch.zeros(n)).shape[-1] for n in (1, 7, 8, 9, 1600, 16001, 24000)], np.int64)
    xs, ys = utility.removeSilentFrames(r(src_t[3]), r(pred_t[3].detach()))
    out["rsf_x"], out["rsf_y"] = mg.t2n(xs), mg.t2n(ys)
    out["spec_out"] = mg.t2n(Spectrogram(512, 256, 128, 2)(xs))[:, ::7]
    out["thirdoct"] = mg.t2n(utility.thirdoct(fs=10000, nfft=512, num_bands=15, min_freq=150))

    # ---- the loss terms, values and the gradient w.r.t. the prediction --------------------------------------------------
    model = CRN.TemporalCRN(**mg.TINY)
    loss, stoi, sisnr = model.compute_loss(src_t, pred_t, len_t)
    loss.backward()
    out["loss"] = np.array([float(loss), float(stoi), float(sisnr)], np.float32)
    out["grad_pred_s5"] = mg.t2n(pred_t.grad)[:, ::5]  # every 5th sample of d loss / d pred
    out["grad_pred_norm"] = np.array([float(pred_t.grad.norm())], np.float32)
    d = -utility.stoi_loss(src_t, pred_t.detach(), len_t, reduction="batch")
    out["stoi_per_utt"] = mg.t2n(d)
    # short input (<= 512 samples after silence removal): the 0.99 branch (utility.py:868-870)
    short = -utility.stoi_loss(src_t[:1, :700], pred_t.detach()[:1, :700], torch.tensor([700]), reduction="batch")
    out["stoi_short"] = mg.t2n(short)
    # fewer than 30 frames (M <= 0 branch, utility.py:882-885)
    few = -utility.stoi_loss(src_t[:1, :5000], pred_t.detach()[:1, :5000], torch.tensor([5000]), reduction="batch")
    out["stoi_few_frames"] = mg.t2n(few)
    np.savez_compressed(os.path.join(HERE, "loss_golden.npz"), **out)
    print("wrote loss_golden.npz", os.path.getsize(os.path.join(HERE, "loss_golden.npz")), "bytes;", {k: v.shape for k, v in out.items()})
    print("loss, stoi, sisnr =", out["loss"], "per utt", out["stoi_per_utt"], "short", out["stoi_short"], "few", out["stoi_few_frames"])


if __name__ == "__main__":
    main()

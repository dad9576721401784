"""Training-loss terms of the reference contract (compute_loss, CRN.py:593-617).

cal_si_snr restates utility.py:207-223 in torch (differentiable, any device) and is pinned by the golden vector
`sisnr_out` (tests/test_host_cpu.py).  stoi_loss (utility.py:821-916) depends on torchaudio==0.7.2
Resample/Spectrogram, which are absent from this image and have no fixture in the reference: parity unpinned,
not restated in this round (SURVEY.md 8f-2 ranks the GPU-resident loss as a "next" item)."""
from __future__ import annotations

import torch


def cal_si_snr(separated, source, length=None, eps=1e-8):
    B = len(separated)
    total = 0.0
    for i in range(B):
        n = separated.shape[-1] if length is None else int(length[i])
        s, r = separated[i, :n], source[i, :n]
        s = s - torch.mean(s, dim=-1, keepdim=True)
        r = r - torch.mean(r, dim=-1, keepdim=True)
        true = torch.sum(s * r, dim=-1, keepdim=True) * r / (torch.norm(r, dim=-1, keepdim=True) ** 2 + eps)
        total = total + 20 * torch.log10(eps + torch.norm(true, dim=-1) / (torch.norm(s - true, dim=-1) + eps))
    return total / B


def stoi_loss(y_true_batch, y_pred_batch, lens, reduction="mean"):
    raise NotImplementedError(
        "stoi_loss needs torchaudio==0.7.2 Resample/Spectrogram semantics (reference utility.py:845-880); those are absent "
        "from this image and unpinned by any reference fixture - not restated in this round (see DESIGN.md, out of scope)")

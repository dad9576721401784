"""CPU restatement (numpy) of the reference's training loss - TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(speech_enhancement_mi_amd/losses.py + csrc/se_loss.hip) never does.

Restates, line by line:
  * utility.stoi_loss            /root/reference/utility.py:821-916
  * utility.thirdoct             /root/reference/utility.py:480-518
  * utility.removeSilentFrames   /root/reference/utility.py:521-571
  * utility.cal_si_snr           /root/reference/utility.py:207-223
  * TemporalCRN.compute_loss     /root/reference/CRN.py:593-617   (0.7 * stoi + 0.3 * (-SI-SNR), NaN -> 0)
and the two third-party transforms stoi_loss calls, which are ABSENT from the reference tree and from this image
(torchaudio==0.7.2, requirements.txt):
  * torchaudio.transforms.Resample(16000, 10000) = torchaudio.compliance.kaldi.resample_waveform (Kaldi LinearResample,
    lowpass_filter_width 6).  Restated from the reference's in-tree copy of the same algorithm,
    /root/reference/augment.py:234-545 (speechbrain's `Resample`, "almost directly from torchaudio.compliance.kaldi").
  * torchaudio.transforms.Spectrogram(n_fft=512, win_length=256, hop_length=128, power=2): torch.stft with a periodic
    Hann window of 256 centred in the 512-point frame, centre reflect padding, |X|^2 (published torchaudio 0.7 semantics).
PARITY UNPINNED AT THE TORCHAUDIO BOUNDARY: the reference holds no fixture for either transform; what is pinned
(tests/golden/loss_golden.npz, made by tests/golden/make_golden_loss.py) is the reference's own stoi_loss / cal_si_snr /
compute_loss code run on top of these two restated transforms (Resample through the reference's augment.Resample class).
"""
from __future__ import annotations

import math

import numpy as np

SMALL = np.finfo("float").eps  # utility.py:478


# ---- Kaldi LinearResample (augment.py:278-545) --------------------------------------------------------------------------
def resample_plan(orig_freq=16000, new_freq=10000, lowpass_filter_width=6):
    """first_indices [P] and weights [P, W] of the polyphase filter (augment.py:478-545); float32 arithmetic like torch's."""
    base = math.gcd(orig_freq, new_freq)
    conv_stride = orig_freq // base          # input samples per unit (8)
    output_samples = new_freq // base        # output samples per unit (5)
    f32 = np.float32
    min_freq = min(orig_freq, new_freq)
    lowpass_cutoff = 0.99 * 0.5 * min_freq
    window_width = lowpass_filter_width / (2.0 * lowpass_cutoff)
    output_t = np.arange(0.0, output_samples, dtype=f32) / f32(new_freq)
    min_t = output_t - f32(window_width)
    max_t = output_t + f32(window_width)
    min_input_index = np.ceil(min_t * f32(orig_freq))
    max_input_index = np.floor(max_t * f32(orig_freq))
    num_indices = max_input_index - min_input_index + 1
    W = int(num_indices.max())
    j = np.arange(W, dtype=f32)
    input_index = min_input_index[:, None] + j[None, :]
    delta_t = (input_index / f32(orig_freq)) - output_t[:, None]
    weights = np.zeros_like(delta_t)
    inside = np.abs(delta_t) < f32(window_width)
    weights[inside] = (0.5 * (1 + np.cos(f32(2 * math.pi * lowpass_cutoff / lowpass_filter_width) * delta_t[inside]))).astype(f32)
    zero = delta_t == 0.0
    nz = ~zero
    weights[nz] *= (np.sin(f32(2 * math.pi * lowpass_cutoff) * delta_t[nz]) / (f32(math.pi) * delta_t[nz])).astype(f32)
    weights[zero] *= f32(2 * lowpass_cutoff)
    weights /= f32(orig_freq)
    return min_input_index.astype(np.int64), weights.astype(f32), conv_stride, output_samples


def resample_num_out(n_in, orig_freq=16000, new_freq=10000):
    """LinearResample::GetNumOutputSamples (augment.py:430-476)."""
    tick = orig_freq * new_freq // math.gcd(orig_freq, new_freq)
    tin, tout = tick // orig_freq, tick // new_freq
    interval = n_in * tin
    if interval <= 0:
        return 0
    last = interval // tout
    if last * tout == interval:
        last -= 1
    return last + 1


def resample(x, orig_freq=16000, new_freq=10000):
    """x [n] float32 -> [n_out]: out[i + P*k] = sum_j w[i, j] * x[first[i] + stride*k + j], zeros outside the signal."""
    x = np.asarray(x, np.float32)
    first, w, stride, P = resample_plan(orig_freq, new_freq)
    n_out = resample_num_out(len(x), orig_freq, new_freq)
    out = np.zeros(n_out, np.float32)
    W = w.shape[1]
    pad = W + stride + int(max(0, -first.min()))
    xp = np.concatenate([np.zeros(pad, np.float32), x, np.zeros(pad + stride * ((n_out + P - 1) // P + 1), np.float32)])
    for i in range(P):
        ks = np.arange((n_out - i + P - 1) // P)
        idx = pad + first[i] + stride * ks[:, None] + np.arange(W)[None, :]
        out[i::P] = (xp[idx] * w[i][None, :]).sum(-1, dtype=np.float32)[: len(out[i::P])]
    return out


# ---- utility.thirdoct (utility.py:480-518) ------------------------------------------------------------------------------
def thirdoct(fs=10000, nfft=512, num_bands=15, min_freq=150):
    f = np.linspace(0, fs, nfft + 1, dtype=np.float32)[: nfft // 2 + 1]
    k = np.arange(num_bands, dtype=np.float64)
    freq_low = min_freq * np.power(2.0, (2 * k - 1) / 6)
    freq_high = min_freq * np.power(2.0, (2 * k + 1) / 6)
    obm = np.zeros((num_bands, len(f)), np.float32)
    for i in range(num_bands):
        fl = int(np.argmin(np.square(f - np.float32(freq_low[i]))))
        fh = int(np.argmin(np.square(f - np.float32(freq_high[i]))))
        obm[i, fl:fh] = 1
    return obm


# ---- utility.removeSilentFrames (utility.py:521-571) --------------------------------------------------------------------
def remove_silent_frames(x, y, dyn_range=40, N=256, K=128):
    x = np.asarray(x, np.float32)
    y = np.asarray(y, np.float32)
    w = np.hanning(256).astype(np.float32)
    n1, n2 = len(x) // N, (len(x) - 128) // N
    if n1 <= 0 or n2 < 0 or not (0 <= n1 - n2 <= 1):
        raise ValueError("signal too short for the frame interleave")  # the reference raises here too (caught by stoi_loss)

    def frames(v):
        V = np.zeros((N, n1 + n2), np.float32)
        V[:, 0::2] = v[: n1 * N].reshape(n1, N).T
        V[:, 1::2] = v[128: n2 * N + 128].reshape(n2, N).T
        return V

    X, Y = frames(x), frames(y)
    energy = 20 * np.log10(np.sqrt((w ** 2) @ (X ** 2)) / np.float32(16.0) + np.float32(SMALL))
    msk = (energy - energy.max() + dyn_range) > 0
    xs, ys = w[:, None] * X[:, msk], w[:, None] * Y[:, msk]

    def ola(v):
        return np.concatenate([v[0:128, 0], (v[0:128, 1:] + v[128:, 0:-1]).T.flatten(), v[128:256, -1]])

    return ola(xs), ola(ys)


# ---- torchaudio.transforms.Spectrogram(512, 256, 128, power=2) ----------------------------------------------------------
def spectrogram_power(x, n_fft=512, win_length=256, hop=128):
    x = np.asarray(x, np.float32)
    n = np.arange(win_length)
    win = np.zeros(n_fft, np.float64)
    left = (n_fft - win_length) // 2
    win[left:left + win_length] = 0.5 - 0.5 * np.cos(2 * np.pi * n / win_length)  # periodic Hann (torch.hann_window)
    xp = np.pad(x.astype(np.float64), n_fft // 2, mode="reflect")
    T = 1 + len(x) // hop
    fr = np.stack([xp[t * hop: t * hop + n_fft] * win for t in range(T)], 1)  # [n_fft, T]
    S = np.fft.rfft(fr, axis=0)
    return (S.real ** 2 + S.imag ** 2).astype(np.float32)  # [257, T]


# ---- utility.stoi_loss (utility.py:821-916) -----------------------------------------------------------------------------
def stoi_per_utterance(y_true, y_pred, length):
    """The D[i] of utility.py:856-911 for one utterance (before the minus sign and the batch mean)."""
    N, J, c = 30, 15.0, np.float32(5.62341325)
    obm = thirdoct()
    t = resample(np.asarray(y_true, np.float32)[: int(length)])
    p = resample(np.asarray(y_pred, np.float32)[: int(length)])
    try:
        st, sp = remove_silent_frames(t, p)
    except Exception:  # utility.py:864-867: bare except keeps the unprocessed signals
        st, sp = t, p
    if st.shape[-1] <= 512:
        return 0.99
    Pt, Pp = spectrogram_power(st), spectrogram_power(sp)
    Ot, Op = np.sqrt(obm @ Pt + np.float32(1e-14)), np.sqrt(obm @ Pp + np.float32(1e-14))
    M = Ot.shape[-1] - (N - 1)
    if M <= 0:
        X, Y, M = Ot, Op, 1
    else:
        X = np.concatenate([Ot[:, m:m + N] for m in range(M)], 0)
        Y = np.concatenate([Op[:, m:m + N] for m in range(M)], 0)
    X, Y = X.astype(np.float64), Y.astype(np.float64)
    alpha = np.linalg.norm(X, axis=-1, keepdims=True) / (np.linalg.norm(Y, axis=-1, keepdims=True) + SMALL)
    y = np.minimum(Y * alpha, X + X * c)
    xn = X - X.mean(-1, keepdims=True)
    xn = xn / (np.linalg.norm(xn, axis=-1, keepdims=True) + SMALL)
    yn = y - y.mean(-1, keepdims=True)
    yn = yn / (np.linalg.norm(yn, axis=-1, keepdims=True) + SMALL)
    return float((xn * yn).sum() / (J * M))


def stoi_loss(y_true_batch, y_pred_batch, lens):
    D = np.array([stoi_per_utterance(y_true_batch[i], y_pred_batch[i], lens[i]) for i in range(len(y_pred_batch))], np.float32)
    return float(-D.mean())


# ---- utility.cal_si_snr (utility.py:207-223) ----------------------------------------------------------------------------
def cal_si_snr(separated, source, length=None, eps=1e-8):
    B = len(separated)
    total = 0.0
    for i in range(B):
        n = separated.shape[-1] if length is None else int(length[i])
        s = np.asarray(separated[i, :n], np.float64)
        r = np.asarray(source[i, :n], np.float64)
        s = s - s.mean()
        r = r - r.mean()
        true = (s * r).sum() * r / (np.linalg.norm(r) ** 2 + eps)
        total += 20 * np.log10(eps + np.linalg.norm(true) / (np.linalg.norm(s - true) + eps))
    return float(total / B)


def compute_loss(source, pred, length):
    """(loss, stoi, sisnr) of TemporalCRN.compute_loss (CRN.py:609-616)."""
    stoi = stoi_loss(source, pred, length)
    sisnr = -cal_si_snr(pred, source, length)
    loss = 0.7 * stoi + 0.3 * sisnr
    if math.isnan(loss):
        return 0.0, 0.0, 0.0
    return loss, stoi, sisnr

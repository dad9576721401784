"""Deterministic synthetic weights and audio for parity tests and benchmarks.

No trained checkpoint and no dataset ship with the reference, so
parity is defined on *identical* pseudo-random weights and synthetic 16 kHz
multi-microphone audio.  Both generators are pure integer hashing (splitmix64
finaliser) so that numpy here, the C oracle and any other host can regenerate
bit-identical tensors without shipping megabytes of fixtures and without
depending on a torch RNG stream.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, Tuple

import numpy as np

_M64 = (1 << 64) - 1
_GOLD = 0x9E3779B97F4A7C15
_C1 = 0xBF58476D1CE4E5B9
_C2 = 0x94D049BB133111EB


def fnv1a64(text: str) -> int:
    h = 0xCBF29CE484222325
    for ch in text.encode("utf-8"):
        h ^= ch
        h = (h * 0x100000001B3) & _M64
    return h


def _mix(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64, copy=True)
    x ^= x >> np.uint64(30)
    x *= np.uint64(_C1)
    x ^= x >> np.uint64(27)
    x *= np.uint64(_C2)
    x ^= x >> np.uint64(31)
    return x


def hash_uniform(stream: int, n: int) -> np.ndarray:
    """n float32 values, uniform in [-1, 1), fully determined by (stream, index)."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        x = idx * np.uint64(_GOLD) + np.uint64(stream & _M64)
        x = _mix(_mix(x) + np.uint64(stream & _M64))
    u24 = (x >> np.uint64(40)).astype(np.float64)  # 24 random bits
    return (u24 / float(1 << 23) - 1.0).astype(np.float32)


def hash_tensor(key: str, shape: Tuple[int, ...], seed: int = 0) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    stream = (fnv1a64(key) ^ ((seed * _C1) & _M64)) & _M64
    return hash_uniform(stream, n).reshape(shape)


def _fan_in(key: str, shape: Tuple[int, ...]) -> int:
    if len(shape) <= 1:
        return 1
    if ".conv.weight" in key and key.startswith("deconvlist"):
        # ConvTranspose2d weight is [Cin, Cout, kh, kw]; every output sums Cin * taps/stride inputs
        return shape[0] * shape[2] * shape[3]
    return int(np.prod(shape[1:]))


def make_state_dict(spec: Iterable[Tuple[str, Tuple[int, ...]]], seed: int = 0) -> Dict[str, np.ndarray]:
    """Hash-generated weights for a list of (checkpoint key, shape).

    Scales follow the spirit of PyTorch's default initialisers (U(-1/sqrt(fan_in), +)) so
    activations stay O(1); norm affines are perturbed away from (1, 0) so that a missing or
    mis-broadcast affine shows up in parity tests.
    """
    out: Dict[str, np.ndarray] = {}
    for name, shape in spec:
        shape = tuple(int(s) for s in shape)
        # the reference registers each conv twice (self.conv and self.net[0], CRN.py:314-316), so its
        # state_dict carries `...net.0.weight` aliases of `...conv.weight`: both get the same tensor.
        key = name.replace(".net.0.", ".conv.")
        u = hash_tensor(key, shape, seed)
        if ".norm." in key or "norm.weight" in key or "norm.bias" in key:
            if key.endswith("weight"):
                t = 1.0 + 0.25 * u
            else:
                t = 0.25 * u
        elif key.endswith("bias"):
            t = 0.1 * u
        else:
            t = u * (1.7 / math.sqrt(max(1, _fan_in(key, shape))))
        out[name] = np.ascontiguousarray(t, dtype=np.float32)
    return out


def crn_param_spec(num_channels, num_freqs, hidden, num_layers=1, num_inputs=3, kernel_size=3, variant=0):
    """(key, shape) list of reference TemporalCRN.state_dict() — CRN.py:428-451 (variant 0), CRN_ELU.py:335-365
    (variant 1) and distillation_crn.py TemporalCRN (variant 2, same keys as 1); checked against the live reference
    modules by tests/golden/make_golden.py (fixtures crn_keys.json, crn_variant_keys.json)."""
    spec = []
    L = len(num_channels)
    c0 = 2 * num_inputs - 1
    if variant:
        for i in range(3):
            p = f"preconvlist.{i}."
            spec += [(p + "conv.weight", (c0, c0, 5, 5)), (p + "conv.bias", (c0,)),
                     (p + "conv_trans.weight", (c0, c0, 1, 1)), (p + "conv_trans.bias", (c0,)),
                     (p + "conv_gated.weight", (c0, c0, 1, 1)), (p + "conv_gated.bias", (c0,)),
                     (p + "net.0.weight", (c0, c0, 5, 5)), (p + "net.0.bias", (c0,)),
                     (p + "norm.weight", (1, c0, 1, 1)), (p + "norm.bias", (1, c0, 1, 1))]
    for i in range(L):
        cin = c0 if i == 0 else num_channels[i - 1]
        cout = num_channels[i]
        p = f"convlist.{i}."
        spec += [(p + "conv.weight", (cout, cin, 5, kernel_size)), (p + "conv.bias", (cout,))]
        if variant:
            spec += [(p + "conv_trans.weight", (cout, cout, 1, 1)), (p + "conv_trans.bias", (cout,)),
                     (p + "conv_gated.weight", (cout, cout, 1, 1)), (p + "conv_gated.bias", (cout,))]
        spec += [(p + "net.0.weight", (cout, cin, 5, kernel_size)), (p + "net.0.bias", (cout,)),
                 (p + "norm.weight", (1, cout, 1, 1)), (p + "norm.bias", (1, cout, 1, 1))]
    for j in range(L):
        i = L - 1 - j  # deconvlist[j] mirrors encoder level i (CRN.py:438-444)
        cin = num_channels[i]
        cout = 2 if i == 0 else num_channels[i - 1]
        p = f"deconvlist.{j}."
        spec += [(p + "conv.weight", (cin, cout, 5, kernel_size)), (p + "conv.bias", (cout,)),
                 (p + "net.0.weight", (cin, cout, 5, kernel_size)), (p + "net.0.bias", (cout,)),
                 (p + "residualmask.weight", (cout, cout, 1, 1)), (p + "residualmask.bias", (cout,)),
                 (p + "residualnorm.weight", (1, cout, 1, 1)), (p + "residualnorm.bias", (1, cout, 1, 1)),
                 (p + "residual.weight", (cout, cout, 1, 1)), (p + "residual.bias", (cout,)),
                 (p + "norm.weight", (1, cout, 1, 1)), (p + "norm.bias", (1, cout, 1, 1))]
    D = (num_freqs // 16 + 1) * num_channels[-1]
    for l in range(num_layers):
        ins = D if l == 0 else hidden
        spec += [(f"gru.sequence_model.weight_ih_l{l}", (3 * hidden, ins)),
                 (f"gru.sequence_model.weight_hh_l{l}", (3 * hidden, hidden)),
                 (f"gru.sequence_model.bias_ih_l{l}", (3 * hidden,)),
                 (f"gru.sequence_model.bias_hh_l{l}", (3 * hidden,))]
    spec += [("gru.fc_output_layer.weight", (D, hidden)), ("gru.fc_output_layer.bias", (D,)),
             ("gru.norm.weight", (1, 1, 1, D)), ("gru.norm.bias", (1, 1, 1, D))]
    return spec


def fsn_param_spec(num_freqs=201, num_mics=3, fb_hidden=512, sb_hidden=384, num_layers=2, sb_neighbors=15, fb_neighbors=0):
    """(key, shape) list of reference FullSubNet.state_dict() (fullsubnet.py:728-746; checked against the live module,
    fixture fsn_keys.json)."""
    spec = []
    for name, ins, hid, out in (("fb_model", num_freqs * num_mics, fb_hidden, num_freqs),
                                ("sb_model", (2 * sb_neighbors + 1) + (2 * fb_neighbors + 1), sb_hidden, 2)):
        for l in range(num_layers):
            i = ins if l == 0 else hid
            spec += [(f"{name}.sequence_model.weight_ih_l{l}", (4 * hid, i)), (f"{name}.sequence_model.weight_hh_l{l}", (4 * hid, hid)),
                     (f"{name}.sequence_model.bias_ih_l{l}", (4 * hid,)), (f"{name}.sequence_model.bias_hh_l{l}", (4 * hid,))]
        spec += [(f"{name}.fc_output_layer.weight", (out, hid)), (f"{name}.fc_output_layer.bias", (out,))]
    return spec


def synth_utterances(batch: int, length: int, num_mics: int = 3, seed: int = 0,
                     sample_rate: int = 16000):
    """Synthetic noisy multi-mic speech-like audio: returns (mix [B,M,L], clean [B,L]).

    Clean: 8 harmonics of f0~U[90,250] Hz under a 3-6 Hz raised-cosine syllabic envelope; noise:
    one-pole low-passed white noise; each mic sees integer-lag/gain variants; SNR~U[-5,25] dB;
    peak normalised to 0.95.  Stands in for data_c.py:210-252 + multichannel.py (gpuRIR).
    """
    mix = np.zeros((batch, num_mics, length), np.float32)
    clean = np.zeros((batch, length), np.float32)
    t = np.arange(length + 8, dtype=np.float64) / sample_rate
    for b in range(batch):
        u = hash_uniform(fnv1a64(f"utt{seed}:{b}"), 64).astype(np.float64) * 0.5 + 0.5  # U[0,1)
        f0 = 90.0 + 160.0 * u[0]
        env_f = 3.0 + 3.0 * u[1]
        s = np.zeros_like(t)
        for h in range(1, 9):
            s += (1.0 / h) * np.sin(2 * np.pi * f0 * h * t + 2 * np.pi * u[1 + h])
        env = 0.5 - 0.5 * np.cos(2 * np.pi * env_f * t + 2 * np.pi * u[10])
        gate = (np.sin(2 * np.pi * (0.8 + u[11]) * t + 2 * np.pi * u[12]) > -0.6).astype(np.float64)
        s = s * env * gate
        w = hash_uniform(fnv1a64(f"noise{seed}:{b}"), length + 8).astype(np.float64)
        n = np.empty_like(w)
        acc = 0.0
        # one-pole low-pass a=0.9 (vectorised via lfilter-equivalent recursion in blocks)
        try:
            from scipy.signal import lfilter
            n = lfilter([1.0], [1.0, -0.9], w)
        except Exception:  # pragma: no cover
            for i in range(len(w)):
                acc = 0.9 * acc + w[i]
                n[i] = acc
        snr_db = -5.0 + 30.0 * u[13]
        ps = np.mean(s ** 2) + 1e-12
        pn = np.mean(n ** 2) + 1e-12
        n = n * math.sqrt(ps / (pn * 10 ** (snr_db / 10)))
        mics = []
        for m in range(num_mics):
            ls = int(u[14 + 2 * m] * 7) % 7
            ln = int(u[15 + 2 * m] * 7) % 7
            gs = 0.7 + 0.3 * u[24 + 2 * m]
            gn = 0.7 + 0.3 * u[25 + 2 * m]
            mics.append(gs * s[ls:ls + length] + gn * n[ln:ln + length])
        mics = np.stack(mics)
        peak = np.max(np.abs(mics)) + 1e-12
        scale = 0.95 / peak
        mix[b] = (mics * scale).astype(np.float32)
        clean[b] = (s[:length] * scale).astype(np.float32)
    return mix, clean


# ---- host restatement of the GPU data generator (csrc/se_synth.hip): the checker of tests/test_synth_gen.py ----

def _image_axis(n: np.ndarray, L: float, xs: float, b0: float, b1: float):
    """Image n of a source at xs in a room [0, L]: position and the product of the wall reflection coefficients."""
    an = np.abs(n)
    even = (an % 2) == 0
    pos = np.where(even, n * L + xs, (n + 1) * L - xs)
    r0 = np.where(even, an // 2, np.where(n > 0, (an - 1) // 2, (an + 1) // 2))
    r1 = np.where(even, an // 2, np.where(n > 0, (an + 1) // 2, (an - 1) // 2))
    return pos, np.power(b0, r0) * np.power(b1, r1)


def image_rir(room, beta, src, mic, nb_img, fs=16000.0, c=343.0, length=4096):
    """Image-source room impulse response of ONE source / microphone pair (what gpuRIR.simulateRIR computes before its
    diffuse-tail model; multichannel.py:83-93): sum over images of prod(beta^reflections) / (4 pi d) at the fractional delay
    d fs / c through a Hann-windowed sinc of Tw = 8 ms."""
    Tw = int(round(8e-3 * fs))
    Tw += Tw & 1
    nx, ny, nz = nb_img
    px, ax = _image_axis(np.arange(nx) - nx // 2, room[0], src[0], beta[0], beta[1])
    py, ay = _image_axis(np.arange(ny) - ny // 2, room[1], src[1], beta[2], beta[3])
    pz, az = _image_axis(np.arange(nz) - nz // 2, room[2], src[2], beta[4], beta[5])
    d = np.sqrt((px[:, None, None] - mic[0]) ** 2 + (py[None, :, None] - mic[1]) ** 2 + (pz[None, None, :] - mic[2]) ** 2).ravel()
    amp = (ax[:, None, None] * ay[None, :, None] * az[None, None, :]).ravel() / (4 * np.pi * np.maximum(d, 1e-3))
    tau = d * fs / c
    h = np.zeros(length, np.float64)
    k0 = np.ceil(tau - Tw / 2).astype(np.int64)
    for j in range(Tw + 1):
        k = k0 + j
        x = k - tau
        ok = (k >= 0) & (k < length) & (x <= Tw / 2)
        w = 0.5 * (1.0 + np.cos(2 * np.pi * x / Tw))
        np.add.at(h, k[ok], (amp * w * np.sinc(x))[ok])
    return h.astype(np.float32)


def _hash32(x):
    x = np.asarray(x, dtype=np.uint64) & 0xFFFFFFFF
    x ^= x >> 16; x = (x * 0x7feb352d) & 0xFFFFFFFF; x ^= x >> 15; x = (x * 0x846ca68b) & 0xFFFFFFFF; x ^= x >> 16
    return x


def diffuse_tail(h: np.ndarray, tdiff: float, t60: float, fs: float, seed: int, rir_index: int) -> np.ndarray:
    """numpy restatement of k_rir_tail (csrc/se_synth.hip): h[n >= Td] <- rms(h[Td-W:Td]) * exp(-6.9078 (n-Td) / (T60 fs)) * logistic noise."""
    h = np.array(h, dtype=np.float32)
    Lr, Td = h.shape[0], int(np.float32(tdiff) * np.float32(fs))
    if Td >= Lr or t60 <= 0:
        return h
    W = min(Td, int(0.010 * fs))
    rms = np.sqrt(np.sum(h[Td - W:Td].astype(np.float64) ** 2) / W) if W > 0 else 0.0
    stream = _hash32((seed & 0xFFFFFFFF) ^ ((rir_index * 0x9e3779b9) & 0xFFFFFFFF))
    n = np.arange(Td, Lr, dtype=np.uint64)
    u = ((_hash32((stream + n) & 0xFFFFFFFF) >> 8).astype(np.float64) + 0.5) / 16777216.0
    xi = np.log(u / (1.0 - u)) * 0.5513289
    h[Td:] = (rms * np.exp(-(6.9078 / (t60 * fs)) * (n - Td).astype(np.float64)) * xi).astype(np.float32)
    return h


def fir_filter(x: np.ndarray, h: np.ndarray) -> np.ndarray:
    """y[n] = sum_k h[k] x[n - k], truncated to len(x) (gpuRIR.simulateTrajectory of a static source)."""
    from scipy.signal import fftconvolve
    return fftconvolve(x.astype(np.float64), h.astype(np.float64))[:len(x)].astype(np.float32)


def mix_noise(y: np.ndarray, snr_db: float, max_amp: float = 0.95):
    """y [S][M][L], the last source is the noise: AddNoise.forward (augment.py:29-77) with per-channel mean-|x| amplitudes,
    then the MAX_AMP guard of data_c.py:249-250.  Returns (mix [M][L], noise [M][L])."""
    clean = y[:-1].astype(np.float64).sum(0)
    nz = y[-1].astype(np.float64)
    f = 1.0 / (10.0 ** (snr_db / 20.0) + 1.0)
    ac = np.abs(clean).mean(-1, keepdims=True)
    an = np.abs(nz).mean(-1, keepdims=True)
    noise = nz * (f * ac / (an + 1e-8))
    mix = (1.0 - f) * clean + noise
    mix = mix / np.maximum(np.abs(mix).max(-1, keepdims=True), 1.0)
    mx = np.abs(mix).max()
    if mx >= max_amp:
        mix = mix * max_amp / (mx + 1e-10)
    return mix.astype(np.float32), noise.astype(np.float32)


def si_sdr(reference: np.ndarray, estimation: np.ndarray) -> np.ndarray:
    """Scale-invariant SDR in dB, the restatable eval metric (reference metrics.py:61-85)."""
    estimation, reference = np.broadcast_arrays(estimation.astype(np.float64), reference.astype(np.float64))
    ref_e = np.sum(reference ** 2, axis=-1, keepdims=True)
    scale = np.sum(reference * estimation, axis=-1, keepdims=True) / ref_e
    proj = scale * reference
    noise = estimation - proj
    return 10 * np.log10(np.sum(proj ** 2, axis=-1) / np.sum(noise ** 2, axis=-1))

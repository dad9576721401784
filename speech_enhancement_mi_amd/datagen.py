"""GPU generator of synthetic multi-microphone training data behind the interface of the reference's
`multichannel.Single2Multi` + `data_c.LibriPartyDataset.dynamic_mix` (multichannel.py:9-103, data_c.py:210-252).

The reference simulates rooms with gpuRIR on the CPU side of the data loader, one utterance at a time; here a whole batch of
rooms is drawn on the host (a few dozen scalars per room, same sampling ranges and formulas) and the heavy parts - the
image-source RIRs, the dry-source convolutions and the SNR-controlled mix - run as HIP kernels (csrc/se_synth.hip) on device
tensors, so a DP-training rank produces its own input on its own GPU.  gpuRIR is not available in this image (un-vendored,
unpinned): the image-source model is restated from its publication and its diffuse-tail model is left out (parity unpinned);
speech_enhancement_mi_amd/synth.py holds the numpy restatement the kernels are tested against.  No CPU fallback."""
import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Sequence, Tuple

import numpy as np
import torch

from .engine import Engine, load_library

SOUND_SPEED = 343.0
MAX_AMP = 0.95  # data_c.py:16


def att2t_sabine(att_db: float, t60: float) -> float:
    """gpuRIR.att2t_SabineEstimator: time for the Sabine decay to lose att_db."""
    return att_db / 60.0 * t60


def t2n(t: float, room: Sequence[float], c: float = SOUND_SPEED):
    """gpuRIR.t2n: images per axis needed to cover reflections up to time t."""
    return [int(math.ceil(2.0 * t * c / r)) for r in room]


@dataclass
class RoomBatch:
    room: np.ndarray    # [R][3]
    beta: np.ndarray    # [R][6]
    src: np.ndarray     # [R][S][3]  (the last source is the noise position)
    mic: np.ndarray     # [R][M][3]
    nb_img: Tuple[int, int, int]
    rir_len: int
    snr_db: np.ndarray = field(default=None)  # [R]
    t60: np.ndarray = field(default=None)     # [R] seconds
    tdiff: np.ndarray = field(default=None)   # [R] seconds: where gpuRIR switches from images to the diffuse tail (multichannel.py:46-52)


class Single2Multi:
    """Same constructor arguments as the reference class (multichannel.py:10-28)."""

    def __init__(self, room_limit, t60_limit, beta_limit, array_limit, mic_limit, source_limit, num_src, num_mic, fs=16000,
                 max_images=(24, 24, 16), max_rir=16384):
        self.room_limit, self.t60_limit, self.beta_limit = room_limit, t60_limit, beta_limit
        self.array_limit, self.mic_limit, self.source_limit = array_limit, mic_limit, source_limit
        self.num_src, self.num_mic, self.fs = num_src, num_mic, fs
        self.max_images, self.max_rir = max_images, max_rir
        self.lib = load_library()

    @staticmethod
    def _uniform(rng, low, high, size=3):
        low, high = np.asarray(low, np.float64), np.asarray(high, np.float64)
        return rng.random(size) * (high - low) + low

    def sample(self, rooms: int, rng: np.random.Generator, snr_low=0.0, snr_high=25.0) -> RoomBatch:
        """multichannel.py:39-80 for `rooms` rooms at once: room size, wall reflections, array and source positions; one image
        count for the batch (the largest room / longest decay, capped by max_images - the diffuse tail is not modelled)."""
        S, M = self.num_src + 1, self.num_mic
        room = np.zeros((rooms, 3)); beta = np.zeros((rooms, 6)); src = np.zeros((rooms, S, 3)); mic = np.zeros((rooms, M, 3))
        nb = np.ones(3, np.int64)
        tmax = 0.1
        t60s, tdiffs = np.zeros(rooms, np.float32), np.full(rooms, 0.1, np.float32)
        for r in range(rooms):
            room[r] = self._uniform(rng, *self.room_limit)
            t60 = rng.random() * (self.t60_limit[1] - self.t60_limit[0]) + self.t60_limit[0]
            beta[r] = self._uniform(rng, *self.beta_limit, size=6)
            if t60 > 0:
                tdiff, tm = att2t_sabine(15, t60), att2t_sabine(60, t60)
                if t60 < 0.15:
                    tdiff = tm
                nb = np.maximum(nb, t2n(tdiff, room[r]))
                tmax = max(tmax, tm)
                t60s[r], tdiffs[r] = t60, tdiff
            array = self._uniform(rng, *self.array_limit) * room[r]
            for m in range(M):
                mic[r, m] = array + self._uniform(rng, *self.mic_limit)
            for s in range(S):
                src[r, s] = self._uniform(rng, *self.source_limit) * room[r]
        nb = tuple(int(min(n + (n & 1), mx)) for n, mx in zip(nb, self.max_images))
        rir_len = int(min(self.max_rir, math.ceil(tmax * self.fs)))
        snr = rng.random(rooms) * (snr_high - snr_low) + snr_low
        return RoomBatch(room.astype(np.float32), beta.astype(np.float32), src.astype(np.float32), mic.astype(np.float32), nb, rir_len,
                         snr.astype(np.float32), t60s, tdiffs)

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(self.lib.se_synth_last_error().decode())

    @staticmethod
    def _upload(rb: RoomBatch, device):
        """All per-room scalars in ONE pinned host buffer and ONE asynchronous copy (six pageable copies = six host stalls per batch)."""
        cached = getattr(rb, "_dev", None)
        if cached is not None and cached[0] == str(device):
            return cached[1]
        parts = [rb.room, rb.beta, rb.src, rb.mic, rb.snr_db if rb.snr_db is not None else np.zeros(len(rb.room), np.float32),
                 rb.tdiff if rb.tdiff is not None else np.zeros(len(rb.room), np.float32), rb.t60 if rb.t60 is not None else np.zeros(len(rb.room), np.float32)]
        flat = np.concatenate([np.ascontiguousarray(a, dtype=np.float32).reshape(-1) for a in parts])
        host = torch.from_numpy(flat).pin_memory()
        dev = host.to(device, non_blocking=True)
        out, off = [], 0
        for a in parts:
            n = int(np.asarray(a).size)
            out.append(dev[off:off + n])
            off += n
        rb._dev = (str(device), (out, host))  # keep the pinned buffer alive until the copy has run
        return rb._dev[1]

    def rir(self, rb: RoomBatch, device="cuda", diffuse=False, seed=0) -> torch.Tensor:
        """[R][S][M][rir_len] image-source RIRs on the GPU.  diffuse=True: beyond each room's Tdiff the response is replaced by the
        stochastic exponentially decaying tail of se_synth_rir_tail (gpuRIR's second stage; PARITY UNPINNED, see csrc/se_synth.hip)."""
        R, S, M = rb.src.shape[0], rb.src.shape[1], rb.mic.shape[1]
        if rb.rir_len > 36 * 1024:
            raise RuntimeError("RIR longer than 36864 samples does not fit the LDS accumulator")
        (dev, _host) = self._upload(rb, device)
        out = torch.empty(R, S, M, rb.rir_len, device=device, dtype=torch.float32)
        self._check(self.lib.se_synth_rir(*[C.c_void_p(t.data_ptr()) for t in dev[:4]], R, S, M, *rb.nb_img, float(self.fs), SOUND_SPEED, rb.rir_len,
                                          C.c_void_p(out.data_ptr()), Engine._stream()))
        if diffuse:
            if rb.tdiff is None or rb.t60 is None:
                raise ValueError("this RoomBatch carries no t60 / tdiff (sample() fills them)")
            td, t6 = dev[5], dev[6]
            self._check(self.lib.se_synth_rir_tail(C.c_void_p(out.data_ptr()), C.c_void_p(td.data_ptr()), C.c_void_p(t6.data_ptr()), R, S, M, rb.rir_len,
                                                   float(self.fs), int(seed) & 0xFFFFFFFF, Engine._stream()))
        return out

    def simulate(self, sources: torch.Tensor, rb: RoomBatch, rir: torch.Tensor = None):
        """sources [R][S][L] on the GPU (S = num_src speech signals + 1 noise signal) -> (mix [R][M][L], reverberant sources
        [R][S][M][L], noise [R][M][L]): Single2Multi.simulate + dynamic_mix's sum / AddNoise / MAX_AMP guard."""
        if not (sources.is_cuda and sources.dtype == torch.float32 and sources.is_contiguous()):
            raise RuntimeError("the generator takes contiguous float32 tensors on the GPU (no CPU fallback)")
        R, S, L = sources.shape
        M = rb.mic.shape[1]
        rir = self.rir(rb, sources.device) if rir is None else rir
        y = torch.empty(R, S, M, L, device=sources.device, dtype=torch.float32)
        st = Engine._stream()
        self._check(self.lib.se_synth_fir(C.c_void_p(sources.data_ptr()), C.c_void_p(rir.data_ptr()), R, S, M, L, rb.rir_len, C.c_void_p(y.data_ptr()), st))
        mix = torch.empty(R, M, L, device=sources.device, dtype=torch.float32)
        noise = torch.empty_like(mix)
        absmax = torch.empty(R, M, device=sources.device, dtype=torch.float32)
        snr = self._upload(rb, sources.device)[0][4]
        self._check(self.lib.se_synth_mix(C.c_void_p(y.data_ptr()), C.c_void_p(snr.data_ptr()), R, S, M, L, MAX_AMP, C.c_void_p(mix.data_ptr()),
                                          C.c_void_p(noise.data_ptr()), C.c_void_p(absmax.data_ptr()), st))
        return mix, y, noise


    def simulate_reference(self, sources, aug_sources=None, noise=False, RIR=None, rng=None, diffuse=False):
        """The reference's call shape (multichannel.py:37-103) for ONE room: `sources` / `aug_sources` = lists of num_src 1-D
        signals -> (multichannel, aug_multichannel[, RIR]) as lists of [M, L] tensors; with `RIR` given, `sources` is one noise
        signal convolved with it (the reference's AddNoise path, augment.py:51-53).  Built on the batched kernels with R = 1; the
        tensors stay on the GPU (the reference returns CPU tensors)."""
        dev = "cuda"
        as_t = lambda v: (v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v))).to(dev, torch.float32).reshape(-1)
        if RIR is not None:
            x = as_t(sources)
            y = torch.empty(1, 1, RIR.shape[-2], x.numel(), device=dev)
            self._check(self.lib.se_synth_fir(C.c_void_p(x.data_ptr()), C.c_void_p(RIR.data_ptr()), 1, 1, RIR.shape[-2], x.numel(), RIR.shape[-1],
                                              C.c_void_p(y.data_ptr()), Engine._stream()))
            return y[0, 0]
        rb = self.sample(1, rng or np.random.default_rng())
        rir = self.rir(rb, dev, diffuse=diffuse)                      # [1][num_src + 1][M][Lr]; the last source position is the noise's
        outs = []
        for group in (sources, aug_sources):
            res = []
            for i, sig in enumerate(group or []):
                x = as_t(sig)
                y = torch.empty(1, 1, self.num_mic, x.numel(), device=dev)
                h = rir[:, i:i + 1].contiguous()
                self._check(self.lib.se_synth_fir(C.c_void_p(x.data_ptr()), C.c_void_p(h.data_ptr()), 1, 1, self.num_mic, x.numel(), rb.rir_len,
                                                  C.c_void_p(y.data_ptr()), Engine._stream()))
                res.append(y[0, 0])
            outs.append(res)
        if noise:
            return outs[0], outs[1], rir[:, self.num_src:self.num_src + 1].contiguous()
        return outs[0], outs[1]


class ChunkChain:
    """data_c.LibriPartyDataset's chunk buffer (data_c.py:60-84, 155-173), restated with its quirks: a mixed utterance is cut into
    chunks of random length l ~ U{16000 .. max_length - 1}; the cursor advances by `start += end` (not `start = end`: after the second
    chunk material is skipped), a remainder shorter than one second ends the cut; get_buffer POPS FROM THE END, so the chunks of one
    utterance are served last-first, the first pop after a refill with flag=False and every further pop with flag=True (the model
    then continues its state across chunks that are not adjacent in time - reproduced as is)."""

    def __init__(self, make_utterance, max_length=60000, rng=None):
        self.make_utterance, self.max_length = make_utterance, int(max_length)
        self.rng = rng or np.random.default_rng()
        self.buffer = []

    def set_buffer(self, mix, source, noise, length):
        start, lens = 0, mix.shape[-1]
        while start < lens:
            l = int(self.rng.integers(16000, self.max_length))
            end = min(lens, start + l)
            if end - start < 16000:
                break
            self.buffer.append((mix[..., start:end], source[..., start:end], noise[..., start:end], end - start))
            start += end

    def __next__(self):
        flag = len(self.buffer) > 0
        while len(self.buffer) == 0:
            self.set_buffer(*self.make_utterance())
        mix, source, noise, length = self.buffer.pop()
        return dict(mix=mix, source=source, noise=noise, length=length, flag=flag)

    def __iter__(self):
        return self

"""Data-parallel training path (BASELINE config 4): utterances sharded across the GPUs of one node,
ONE flat fp32 gradient bucket all-reduced per optimizer step over RCCL/xGMI.

`TrainableCRN` has two interchangeable differentiable forwards of `realtime_process` (CRN.py:560-589):
  * `use_hip_kernels(True)`: every stage forward AND backward on the hand-written kernels (train_net.CRNFunction;
    SURVEY.md 8f-1) - the path bench.py --mode train times;
  * default: a PyTorch autograd restatement of the same arithmetic, CPU-runnable, pinned against the CPU oracle
    (tests/test_training_cpu.py) - the checker the GPU tests compare the kernels with.
Around it: the flat-bucket gradient all-reduce the reference never had (its DDP lines are commented out,
train.py:172-173,252-256) and the optimizer step of the reference trainer (Adam 3e-4, grad-accum 2, clip 5;
train.py:198-204, config.yaml:9,99).  Loss: `compute_loss` = 0.7 * stoi_loss + 0.3 * (-SI-SNR) (CRN.py:609-611; losses.py:
HIP SI-SNR kernels + the batched device-resident STOI restatement, torchaudio boundary unpinned) or the SI-SNR term alone.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn.functional as Fn

from .crn import TemporalCRN
from .crn_elu import TemporalCRN as TemporalCRNELU
from .losses import cal_si_snr

EPS = 1e-8


def _gln(x, w, b):
    """GlobalLayerNorm(time=False), CRN.py:135-149: per-sample stats over all non-batch dims."""
    dims = tuple(range(1, x.dim()))
    mean = x.mean(dims, keepdim=True)
    var = ((x - mean) ** 2).mean(dims, keepdim=True)
    return (x - mean) / (torch.sqrt(var + EPS) + EPS) * w + b


class _TrainableMixin:
    """Differentiable `realtime_process` on top of the drop-in parameter holders (crn.TemporalCRN / crn_elu.TemporalCRN).  Same
    parameters / state_dict as the inference shim, so a checkpoint trained here loads into the HIP engine unchanged."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        c = self._cfg_args
        self._win = int(round(c["sample_rate"] / 1000.0 * c["win_length"]))
        self._hop = int(round(c["sample_rate"] / 1000.0 * c["hop_length"]))
        self._nfft = c["n_fft"]
        self._state = None
        self._hip = False
        self._hip_state_from_torch = False

    def use_hip_kernels(self, flag=True):
        """True: the whole differentiable realtime_process - STFT, features, convolutions, norms, GRU, dense layers, skip gates,
        mask, iSTFT, overlap-add - runs forward AND backward on the hand-written kernels (train_net.CRNFunction: one autograd
        node, no float atomics); False (default): torch ops + autograd, the CPU-runnable checker the tests pin to the oracle."""
        if bool(flag) != self._hip:
            self._state = None  # the two paths keep their carried state in different layouts
        self._hip = bool(flag)
        self._hip_state_from_torch = False
        return self

    # ---- signal glue (utility.py:312-403, CRN.py:505-520) ----
    def _segment(self, x):
        B, M, L = x.shape
        K = self.segment_length
        P = K // 2
        gap = K - (P + L % K) % K
        xp = Fn.pad(x, (P, gap + P))
        n = 2 * (L + gap + P) // K
        idx = (torch.arange(n, device=x.device) * P)[:, None] + torch.arange(K, device=x.device)[None, :]
        return xp[:, :, idx], gap  # [B, M, N, K]

    def _stft(self, seg):  # [..., K] -> [..., F, T] complex
        shp = seg.shape[:-1]
        w = torch.hamming_window(self._win, device=seg.device)
        s = torch.stft(seg.reshape(-1, seg.shape[-1]), self._nfft, self._hop, self._win, w, center=True, pad_mode="constant",
                       normalized=False, onesided=True, return_complex=True)
        return s.reshape(*shp, *s.shape[-2:])

    def _istft(self, spec):  # [..., F, T] complex -> [..., K]
        shp = spec.shape[:-2]
        w = torch.hamming_window(self._win, device=spec.device)
        y = torch.istft(spec.reshape(-1, *spec.shape[-2:]), self._nfft, self._hop, self._win, w, center=True, normalized=False, onesided=True)
        return y.reshape(*shp, y.shape[-1])

    # ---- one segment, CRN.py:454-496 (variant 0) / CRN_ELU.py:367-407 (variant 1) ----
    def _forward_segment(self, X, state):
        """X [B, M, F, T] complex; state = dict(buf=[...], h=tensor|None, pbuf=[...]) (detached, like CRN.py:281,334)."""
        V = self._VARIANT
        actf = Fn.elu if V else torch.relu
        re, im = X.real, X.imag
        ang = torch.atan2(im, re) if V else torch.atan(im / (re + EPS) + EPS)
        mag = torch.sqrt(re ** 2 + im ** 2 + 1e-10)
        x = torch.cat([mag, ang[:, :1] - ang[:, 1:]], dim=1)

        def gated(blk, a):  # CRN_ELU.py:240
            return Fn.conv2d(a, blk.conv_trans.weight, blk.conv_trans.bias) * torch.sigmoid(Fn.conv2d(a, blk.conv_gated.weight, blk.conv_gated.bias))

        new_pbuf = []
        if V:
            for k, blk in enumerate(self.preconvlist):  # x = block(x) + x, CRN_ELU.py:375-376
                fd = 2 ** k
                buf = state["pbuf"][k] if state.get("pbuf") is not None else x.new_zeros(x.shape[0], x.shape[1], x.shape[2], 4)
                y = Fn.conv2d(torch.cat([buf, x], dim=-1), blk.conv.weight, blk.conv.bias, stride=(1, 1), padding=(2 * fd, 0), dilation=(fd, 1))
                new_pbuf.append(x[..., -4:].detach())
                x = _gln(gated(blk, actf(y)), blk.norm.weight, blk.norm.bias) + x
        residuals = [x]
        new_buf = []
        for i, blk in enumerate(self.convlist):
            d = 2 ** i
            P = 2 * d
            buf = state["buf"][i] if state["buf"] is not None else x.new_zeros(x.shape[0], x.shape[1], x.shape[2], P)
            inp = torch.cat([buf, x], dim=-1)
            y = Fn.conv2d(inp, blk.conv.weight, blk.conv.bias, stride=(2, 1), padding=(2, 0), dilation=(1, d))
            new_buf.append(x[..., -P:].detach())
            x = _gln(gated(blk, actf(y)) if V else actf(y), blk.norm.weight, blk.norm.bias)
            residuals.append(x)
        B, C, Fq, T = x.shape
        seq = x.reshape(B, C * Fq, T).permute(0, 2, 1)
        o, h = self.gru.sequence_model(seq, state["h"])
        o = actf(self.gru.fc_output_layer(o))
        o = _gln(o.unsqueeze(1), self.gru.norm.weight, self.gru.norm.bias).squeeze(1)
        x = o.permute(0, 2, 1).reshape(B, C, Fq, T)
        L = len(self.deconvlist)
        for j, blk in enumerate(self.deconvlist):
            d = 2 ** j
            y = Fn.conv_transpose2d(x, blk.conv.weight, blk.conv.bias, stride=(2, 1), padding=(2, 0), dilation=(1, d))[..., -T:]
            y = _gln(actf(y), blk.norm.weight, blk.norm.bias)
            if j < L - 1:
                res = residuals[-2 - j]
                if res.shape[2] > y.shape[2]:
                    y = Fn.pad(y, (0, 0, 0, res.shape[2] - y.shape[2]))
                elif res.shape[2] < y.shape[2]:
                    y = y[:, :, :res.shape[2]]
                m = torch.sigmoid(_gln(Fn.conv2d(res, blk.residualmask.weight, blk.residualmask.bias), blk.residualnorm.weight, blk.residualnorm.bias))
                y = m * actf(Fn.conv2d(res, blk.residual.weight, blk.residual.bias)) + (1.0 - m) * y
            x = y
        m = x.clamp(-9.9, 9.9)  # decompress_cIRM, utility.py:439-442 (the clamp has zero gradient outside, like the reference's masks)
        m = -10.0 * torch.log((10.0 - m) / (10.0 + m))
        Y = torch.complex(m[:, 0] * re[:, 0] - m[:, 1] * im[:, 0], m[:, 1] * re[:, 0] + m[:, 0] * im[:, 0])
        return Y, dict(buf=new_buf, h=h.detach(), pbuf=new_pbuf if V else None)

    def realtime_process_train(self, mixture, flag=False):
        """Differentiable realtime_process (CRN.py:560-589): [B, M, L] -> [B, L]."""
        if self._hip:  # every stage forward and backward on the hand-written kernels, one autograd node (train_net.py)
            from .train_net import realtime_process_fused
            if self._hip_state_from_torch:
                raise RuntimeError("flag=True continuation across a use_hip_kernels() switch is not supported: start with flag=False")
            return realtime_process_fused(self, mixture, flag)
        K = self.segment_length
        P = K // 2
        if not flag:
            mixture = Fn.pad(mixture, (P, 0))
            self._state = dict(buf=None, h=None, pbuf=None)
        seg, gap = self._segment(mixture)  # [B, M, N, K]
        X = self._stft(seg)  # [B, M, N, F, T]
        state = self._state
        outs = []
        for n in range(X.shape[2]):
            Y, state = self._forward_segment(X[:, :, n], state)
            outs.append(self._istft(Y))
        y = torch.stack(outs, dim=1)  # [B, N, K]
        self._state = state
        B, N, _ = y.shape
        s1 = y[:, 0::2].reshape(B, -1)[:, P:]
        s2 = y[:, 1::2].reshape(B, -1)[:, :-P]
        out = (s1 + s2) / 2
        if gap > 0:
            out = out[:, :-gap]
        return out if flag else out[:, P:]


class TrainableCRN(_TrainableMixin, TemporalCRN):
    """CRN.py TemporalCRN (variant 0), trainable."""


class TrainableCRNELU(_TrainableMixin, TemporalCRNELU):
    """CRN_ELU.py TemporalCRN (variant 1) - the model the reference's train.py imports and trains (train.py:16)."""


def si_snr_loss(pred, source, length=None):
    """The SI-SNR term of compute_loss (CRN.py:610): -cal_si_snr(pred, source, length)."""
    return -cal_si_snr(pred, source, length)


class FlatBucket:
    """One contiguous fp32 gradient buffer for all parameters (24.46 MB for the CRN); every p.grad is a view
    into it, so backward accumulates in place and the data-parallel exchange is ONE all-reduce per optimizer step."""

    def __init__(self, params: List[torch.nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self):
        self.flat.zero_()

    def all_reduce_mean(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)  # RCCL over xGMI with backend "nccl"
            self.flat.div_(dist.get_world_size())

    def clip_(self, max_norm: float) -> float:
        """clip_grad_norm_ on the REDUCED gradients, so every rank clips identically (train.py:200)."""
        norm = float(self.flat.norm())
        if norm > max_norm:
            self.flat.mul_(max_norm / (norm + 1e-6))
        return norm


def train_step(model: TrainableCRN, bucket: FlatBucket, optimizer, mixture, source, length=None, accum: int = 1, loss: str = "sisnr", merge=None):
    """One optimizer step of the reference trainer (train.py:195-204) under data parallelism: `accum` micro-batches of local
    utterances, one flat all-reduce, clip 5, Adam.  loss = "full": 0.7 * stoi_loss + 0.3 * (-SI-SNR) (compute_loss,
    CRN.py:609-611); "sisnr": the SI-SNR term alone.

    merge (default: on for the hand-written kernels): gradient accumulation exists to bound memory; every statistic of the model
    is per utterance, so the `accum` micro-batches can share ONE forward / backward sweep (half the dependent GRU steps) while
    the loss is still formed per micro-batch, sum_i loss(micro-batch i) / accum - the same function of the parameters, hence
    the same gradient up to fp32 summation order (tests/test_gpu_round3.py::test_merged_microbatches_give_the_accumulated_gradient).
    merge=False runs the micro-batches one after the other like the reference loop."""
    bucket.zero()
    total = 0.0
    if merge is None:
        merge = model._hip
    srcs = source.chunk(accum)
    lens = [None] * len(srcs) if length is None else list(length.chunk(accum))

    def loss_of(pred, src, ln, slot=0):
        if loss == "full":
            ll = ln if ln is not None else torch.full((pred.shape[0],), pred.shape[-1], dtype=torch.int64, device=pred.device)
            return model.compute_loss(src, pred, ll)[0] / accum
        return si_snr_loss(pred, src, ln) / accum

    if merge:
        pred = model.realtime_process_train(mixture)
        val = None
        for slot, (p_i, src, ln) in enumerate(zip(pred.chunk(accum), srcs, lens)):
            v = loss_of(p_i, src, ln, slot)
            val = v if val is None else val + v
        val.backward()
        total = float(val.detach())
    else:
        for mix, src, ln in zip(mixture.chunk(accum), srcs, lens):
            val = loss_of(model.realtime_process_train(mix), src, ln)
            val.backward()
            total += float(val.detach())
    if model._hip:
        from . import train_ops
        train_ops.pseq_check()  # a persistent GRU launch that gave up its bounded spin must not go unnoticed
    bucket.all_reduce_mean()
    bucket.clip_(5.0)
    optimizer.step()
    return total

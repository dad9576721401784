"""Data-parallel training path (BASELINE config 4): utterances sharded across the GPUs of one node,
ONE flat fp32 gradient bucket all-reduced per optimizer step over RCCL/xGMI.

Scope of this round: the inference hot path is hand-written HIP; the *training* forward/backward here is a PyTorch-ROCm
autograd restatement of the same arithmetic (SURVEY.md 7 step 8 - hand-written backward kernels are ranked "next", 8f-1),
so what this module adds is (a) a differentiable `TrainableCRN` whose forward is pinned against the CPU oracle
(tests/test_training_cpu.py), (b) the flat-bucket gradient all-reduce the reference never had (its DDP lines are
commented out, train.py:172-173,252-256), and (c) the optimizer step of the reference trainer (Adam 3e-4, grad-accum 2,
clip 5; train.py:198-204, config.yaml:9,99).  Loss: the SI-SNR term of compute_loss (CRN.py:609-611); the STOI term needs
torchaudio 0.7.2 semantics that are unpinned here (losses.py) and is left out - stated, not hidden.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn.functional as Fn

from .crn import TemporalCRN
from .losses import cal_si_snr

EPS = 1e-8


def _gln(x, w, b):
    """GlobalLayerNorm(time=False), CRN.py:135-149: per-sample stats over all non-batch dims."""
    dims = tuple(range(1, x.dim()))
    mean = x.mean(dims, keepdim=True)
    var = ((x - mean) ** 2).mean(dims, keepdim=True)
    return (x - mean) / (torch.sqrt(var + EPS) + EPS) * w + b


class TrainableCRN(TemporalCRN):
    """CRN.py TemporalCRN (variant 0) with a differentiable torch forward.  Same parameters / state_dict as the inference
    shim, so a checkpoint trained here loads into the HIP engine unchanged."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        c = self._cfg_args
        self._win = int(round(c["sample_rate"] / 1000.0 * c["win_length"]))
        self._hop = int(round(c["sample_rate"] / 1000.0 * c["hop_length"]))
        self._nfft = c["n_fft"]
        self._state = None
        self._hip = False
        self.batch_segments = True  # HIP path: every layer once over B x N segment streams (False: one segment at a time)

    def use_hip_kernels(self, flag=True):
        """True: convolutions, transposed convolutions, the GRU and the dense layers run forward AND backward on the
        hand-written kernels of csrc/train_ops.inc.h (train_ops.py); False: torch ops + autograd (the checker)."""
        self._hip = bool(flag)
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

    # ---- one segment, CRN.py:454-496 ----
    def _forward_segment(self, X, state):
        """X [B, M, F, T] complex; state = dict(buf=[...], h=tensor|None) (detached, like CRN.py:281,334)."""
        re, im = X.real, X.imag
        ang = torch.atan(im / (re + EPS) + EPS)
        mag = torch.sqrt(re ** 2 + im ** 2 + 1e-10)
        x = torch.cat([mag, ang[:, :1] - ang[:, 1:]], dim=1)
        residuals = [x]
        new_buf = []
        for i, blk in enumerate(self.convlist):
            d = 2 ** i
            P = 2 * d
            buf = state["buf"][i] if state["buf"] is not None else x.new_zeros(x.shape[0], x.shape[1], x.shape[2], P)
            inp = torch.cat([buf, x], dim=-1)
            y = Fn.conv2d(inp, blk.conv.weight, blk.conv.bias, stride=(2, 1), padding=(2, 0), dilation=(1, d))
            new_buf.append(x[..., -P:].detach())
            x = _gln(torch.relu(y), blk.norm.weight, blk.norm.bias)
            residuals.append(x)
        B, C, Fq, T = x.shape
        seq = x.reshape(B, C * Fq, T).permute(0, 2, 1)
        o, h = self.gru.sequence_model(seq, state["h"])
        o = torch.relu(self.gru.fc_output_layer(o))
        o = _gln(o.unsqueeze(1), self.gru.norm.weight, self.gru.norm.bias).squeeze(1)
        x = o.permute(0, 2, 1).reshape(B, C, Fq, T)
        L = len(self.deconvlist)
        for j, blk in enumerate(self.deconvlist):
            d = 2 ** j
            y = Fn.conv_transpose2d(x, blk.conv.weight, blk.conv.bias, stride=(2, 1), padding=(2, 0), dilation=(1, d))[..., -T:]
            y = _gln(torch.relu(y), blk.norm.weight, blk.norm.bias)
            if j < L - 1:
                res = residuals[-2 - j]
                if res.shape[2] > y.shape[2]:
                    y = Fn.pad(y, (0, 0, 0, res.shape[2] - y.shape[2]))
                elif res.shape[2] < y.shape[2]:
                    y = y[:, :, :res.shape[2]]
                m = torch.sigmoid(_gln(Fn.conv2d(res, blk.residualmask.weight, blk.residualmask.bias), blk.residualnorm.weight, blk.residualnorm.bias))
                y = m * torch.relu(Fn.conv2d(res, blk.residual.weight, blk.residual.bias)) + (1.0 - m) * y
            x = y
        m = x.clamp(-9.9, 9.9)  # decompress_cIRM, utility.py:439-442 (the clamp has zero gradient outside, like the reference's masks)
        m = -10.0 * torch.log((10.0 - m) / (10.0 + m))
        Y = torch.complex(m[:, 0] * re[:, 0] - m[:, 1] * im[:, 0], m[:, 1] * re[:, 0] + m[:, 0] * im[:, 0])
        return Y, dict(buf=new_buf, h=h.detach())

    # ---- one segment on the hand-written kernels; activations [B, C, T, F] (F innermost, the engine's layout) ----
    def _forward_segment_hip(self, X, state):
        from . import train_ops as K
        re, im = X.real, X.imag  # [B, M, F, T]
        ang = torch.atan(im / (re + EPS) + EPS)
        mag = torch.sqrt(re ** 2 + im ** 2 + 1e-10)
        x = torch.cat([mag, ang[:, :1] - ang[:, 1:]], dim=1).permute(0, 1, 3, 2).contiguous()  # [B, 5, T, F]
        residuals = [x]
        new_buf = []
        for i, blk in enumerate(self.convlist):
            prev = state["buf"][i] if state["buf"] is not None else None  # the whole previous input: its last 2d columns are the buffer
            y = K.conv_block(x, prev, blk.conv.weight, blk.conv.bias, 2 ** i)
            new_buf.append(x.detach())
            x = _gln(torch.relu(y), blk.norm.weight, blk.norm.bias)
            residuals.append(x)
        B, C, T, Fq = x.shape
        seq = x.permute(0, 2, 1, 3).reshape(B, T, C * Fq)  # feature index c * F + f (CRN.py:476-478)
        g = self.gru.sequence_model
        hs = []
        for l in range(g.num_layers):
            h0 = state["h"][l] if state["h"] is not None else seq.new_zeros(B, g.hidden_size)
            seq, hT = K.gru_layer(seq, h0, getattr(g, f"weight_ih_l{l}"), getattr(g, f"weight_hh_l{l}"), getattr(g, f"bias_ih_l{l}"), getattr(g, f"bias_hh_l{l}"))
            hs.append(hT.detach())
        o = torch.relu(K.linear(seq, self.gru.fc_output_layer.weight, self.gru.fc_output_layer.bias))
        o = _gln(o.unsqueeze(1), self.gru.norm.weight, self.gru.norm.bias).squeeze(1)
        x = o.reshape(B, T, C, Fq).permute(0, 2, 1, 3).contiguous()
        L = len(self.deconvlist)

        def conv1x1(t, mod):  # [B, C, T, F] x [Co, C, 1, 1] through the GEMM kernel
            w = mod.weight.reshape(mod.weight.shape[0], -1)
            return K.linear(t.permute(0, 2, 3, 1), w, mod.bias).permute(0, 3, 1, 2)

        for j, blk in enumerate(self.deconvlist):
            y = K.deconv_block(x, blk.conv.weight, blk.conv.bias, 2 ** j)
            y = _gln(torch.relu(y), blk.norm.weight, blk.norm.bias)
            if j < L - 1:
                res = residuals[-2 - j]
                if res.shape[3] > y.shape[3]:
                    y = Fn.pad(y, (0, res.shape[3] - y.shape[3]))
                elif res.shape[3] < y.shape[3]:
                    y = y[..., :res.shape[3]]
                m = torch.sigmoid(_gln(conv1x1(res, blk.residualmask), blk.residualnorm.weight, blk.residualnorm.bias))
                y = m * torch.relu(conv1x1(res, blk.residual)) + (1.0 - m) * y
            x = y
        m = x.clamp(-9.9, 9.9)
        m = -10.0 * torch.log((10.0 - m) / (10.0 + m))
        mr, mi = m[:, 0].transpose(1, 2), m[:, 1].transpose(1, 2)  # back to [B, F, T]
        Y = torch.complex(mr * re[:, 0] - mi * im[:, 0], mi * re[:, 0] + mr * im[:, 0])
        return Y, dict(buf=new_buf, h=hs)

    # ---- ALL segments of a batch at once on the hand-written kernels ----
    def _forward_all_hip(self, X, state):
        """X [B, M, N, F, T] complex -> Y [B, N, F, T] complex.  The per-segment loop of realtime_process (CRN.py:577-586) is
        sequential only through the GRU state: a convolution's time buffer is the PREVIOUS segment's (detached) input of the
        same block (CRN.py:325-337), every norm is per segment, so each layer runs once over B x N streams with the history
        tensor = the input shifted by one segment.  Same arithmetic as _forward_segment_hip, 34x fewer launches."""
        from . import train_ops as K
        B, M, N, Fq0, T = X.shape
        Xs = X.permute(0, 2, 1, 3, 4).reshape(B * N, M, Fq0, T)
        re, im = Xs.real, Xs.imag
        ang = torch.atan(im / (re + EPS) + EPS)
        mag = torch.sqrt(re ** 2 + im ** 2 + 1e-10)
        x = torch.cat([mag, ang[:, :1] - ang[:, 1:]], dim=1).permute(0, 1, 3, 2).contiguous()  # [B*N, 5, T, F]

        def shifted(t, first):  # history of segment n = input of segment n - 1 (state / zeros for n = 0), no gradient
            tv = t.detach().reshape(B, N, *t.shape[1:])
            head = first[:, None] if first is not None else torch.zeros_like(tv[:, :1])
            return torch.cat([head, tv[:, :-1]], dim=1).reshape(t.shape)

        residuals = [x]
        new_buf = []
        for i, blk in enumerate(self.convlist):
            prev = shifted(x, state["buf"][i] if state["buf"] is not None else None)
            y = K.conv_block(x, prev, blk.conv.weight, blk.conv.bias, 2 ** i)
            new_buf.append(x.detach().reshape(B, N, *x.shape[1:])[:, -1].contiguous())
            x = _gln(torch.relu(y), blk.norm.weight, blk.norm.bias)
            residuals.append(x)
        BN, C, T, Fq = x.shape
        seq = x.permute(0, 2, 1, 3).reshape(B, N * T, C * Fq)
        g = self.gru.sequence_model
        hs = []
        for l in range(g.num_layers):  # the recurrence: ONE pass over the N * T steps of every utterance; the carried state is
            # detached at every segment seam (CRN.py:281), which seg_len = T reproduces in the backward sweep
            h0 = state["h"][l] if state["h"] is not None else seq.new_zeros(B, g.hidden_size)
            seq, hT = K.gru_layer(seq, h0, getattr(g, f"weight_ih_l{l}"), getattr(g, f"weight_hh_l{l}"), getattr(g, f"bias_ih_l{l}"),
                                  getattr(g, f"bias_hh_l{l}"), seg_len=T)
            hs.append(hT.detach())
        o = seq.reshape(BN, T, -1)
        o = torch.relu(K.linear(o, self.gru.fc_output_layer.weight, self.gru.fc_output_layer.bias))
        o = _gln(o.unsqueeze(1), self.gru.norm.weight, self.gru.norm.bias).squeeze(1)
        x = o.reshape(BN, T, C, Fq).permute(0, 2, 1, 3).contiguous()
        L = len(self.deconvlist)

        def conv1x1(t, mod):
            w = mod.weight.reshape(mod.weight.shape[0], -1)
            return K.linear(t.permute(0, 2, 3, 1), w, mod.bias).permute(0, 3, 1, 2)

        for j, blk in enumerate(self.deconvlist):
            y = K.deconv_block(x, blk.conv.weight, blk.conv.bias, 2 ** j)
            y = _gln(torch.relu(y), blk.norm.weight, blk.norm.bias)
            if j < L - 1:
                res = residuals[-2 - j]
                if res.shape[3] > y.shape[3]:
                    y = Fn.pad(y, (0, res.shape[3] - y.shape[3]))
                elif res.shape[3] < y.shape[3]:
                    y = y[..., :res.shape[3]]
                m = torch.sigmoid(_gln(conv1x1(res, blk.residualmask), blk.residualnorm.weight, blk.residualnorm.bias))
                y = m * torch.relu(conv1x1(res, blk.residual)) + (1.0 - m) * y
            x = y
        m = x.clamp(-9.9, 9.9)
        m = -10.0 * torch.log((10.0 - m) / (10.0 + m))
        mr, mi = m[:, 0].transpose(1, 2), m[:, 1].transpose(1, 2)  # [B*N, F, T]
        Y = torch.complex(mr * re[:, 0] - mi * im[:, 0], mi * re[:, 0] + mr * im[:, 0])
        return Y.reshape(B, N, *Y.shape[1:]), dict(buf=new_buf, h=hs)

    def realtime_process_train(self, mixture, flag=False):
        """Differentiable realtime_process (CRN.py:560-589): [B, M, L] -> [B, L]."""
        K = self.segment_length
        P = K // 2
        if not flag:
            mixture = Fn.pad(mixture, (P, 0))
            self._state = dict(buf=None, h=None)
        seg, gap = self._segment(mixture)  # [B, M, N, K]
        X = self._stft(seg)  # [B, M, N, F, T]
        state = self._state
        if self._hip and self.batch_segments:
            Y, state = self._forward_all_hip(X, state)
            y = self._istft(Y)  # [B, N, K]
        else:
            outs = []
            seg_fn = self._forward_segment_hip if self._hip else self._forward_segment
            for n in range(X.shape[2]):
                Y, state = seg_fn(X[:, :, n], state)
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


def train_step(model: TrainableCRN, bucket: FlatBucket, optimizer, mixture, source, length=None, accum: int = 1, loss: str = "sisnr"):
    """One optimizer step of the reference trainer (train.py:195-204) under data parallelism: `accum` micro-batches of local
    utterances, one flat all-reduce, clip 5, Adam.  loss = "full": 0.7 * stoi_loss + 0.3 * (-SI-SNR) (compute_loss,
    CRN.py:609-611); "sisnr": the SI-SNR term alone."""
    bucket.zero()
    total = 0.0
    chunks = mixture.chunk(accum)
    for i, mix in enumerate(chunks):
        src = source.chunk(accum)[i]
        ln = None if length is None else length.chunk(accum)[i]
        pred = model.realtime_process_train(mix)
        if loss == "full":
            lens = ln if ln is not None else torch.full((mix.shape[0],), mix.shape[-1], dtype=torch.int64)
            val = model.compute_loss(src, pred, lens)[0] / accum
        else:
            val = si_snr_loss(pred, src, ln) / accum
        val.backward()
        total += float(val.detach())
    if model._hip:
        from . import train_ops
        train_ops.pseq_check()  # a persistent GRU launch that gave up its bounded spin must not go unnoticed
    bucket.all_reduce_mean()
    bucket.clip_(5.0)
    optimizer.step()
    return total

"""Training loss of the reference contract, device-resident: compute_loss = 0.7 * stoi_loss + 0.3 * (-SI-SNR)
(reference CRN.py:593-617).

* `cal_si_snr` restates utility.py:207-223.  On GPU tensors it runs as ONE fused HIP kernel pair (csrc/se_loss.hip through the
  C ABI `se_loss_sisnr_fwd/bwd`): per-utterance moment reductions in the forward, a closed-form elementwise backward - no
  Python loop over the batch, no host hop.  On CPU tensors (unit tests, CPU training) the same formula runs in torch.
* `stoi_loss` on CUDA tensors runs on the `se_loss_stoi_*` kernels (csrc/se_stoi.hip: 7 launches forward, 4 backward; `_StoiHip`).
  `_stoi_d` restates utility.py:821-916 (+ thirdoct 480-518, removeSilentFrames 521-571) as batched, differentiable torch
  tensor code that stays on the tensors' device: the CPU path and the checker the kernels are tested against (STOI_KERNELS = False).  The reference moves every utterance to the CPU and loops in Python
  (`y_pred_batch.cpu()`, utility.py:845-880); here the whole batch is processed at once with validity masks - the
  data-dependent lengths (silent-frame removal) never come back to the host, so there is no synchronisation in the loss.
  The two torchaudio==0.7.2 transforms it calls are absent from the reference tree and this image and are restated:
  Resample(16000, 10000) = Kaldi LinearResample (the reference's in-tree copy of the algorithm is augment.py:234-545) as a
  5-phase strided convolution; Spectrogram(512, 256, 128, power=2) over torch.fft.  PARITY UNPINNED AT THE TORCHAUDIO
  BOUNDARY; pinned above it by tests/golden/loss_golden.npz (the reference's own stoi_loss / compute_loss code run on these
  restated transforms, values AND the gradient w.r.t. the prediction).
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch
import torch.nn.functional as Fn

SMALL = float(np.finfo("float").eps)  # utility.py:478 `smallVal`


# =====================================================================================================================
# SI-SNR (utility.py:207-223)
# =====================================================================================================================
def _cal_si_snr_torch(separated, source, length=None, eps=1e-8):
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


class _SiSnrHip(torch.autograd.Function):
    """mean over the batch of the per-utterance SI-SNR, forward and backward on hand-written HIP kernels."""

    @staticmethod
    def forward(ctx, separated, source, lens_dev):
        from . import engine
        lib = engine.load_library()
        sep = separated.contiguous().float()
        src = source.contiguous().float()
        B, L = sep.shape
        per = torch.empty(B, dtype=torch.float32, device=sep.device)
        stats = torch.empty(B * 8, dtype=torch.float64, device=sep.device)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = lib.se_loss_sisnr_fwd(C.c_void_p(sep.data_ptr()), C.c_void_p(src.data_ptr()), C.c_void_p(lens_dev.data_ptr()), B, L,
                                   C.c_void_p(per.data_ptr()), C.c_void_p(stats.data_ptr()), st)
        if rc != 0:
            raise RuntimeError(f"se_loss_sisnr_fwd failed ({rc})")
        ctx.save_for_backward(sep, src, lens_dev, stats)
        return per.mean()

    @staticmethod
    def backward(ctx, g):
        from . import engine
        lib = engine.load_library()
        sep, src, lens_dev, stats = ctx.saved_tensors
        B, L = sep.shape
        grad = torch.empty_like(sep)
        gscale = (g.float() / B).reshape(1).contiguous()
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = lib.se_loss_sisnr_bwd(C.c_void_p(sep.data_ptr()), C.c_void_p(src.data_ptr()), C.c_void_p(lens_dev.data_ptr()), B, L,
                                   C.c_void_p(stats.data_ptr()), C.c_void_p(gscale.data_ptr()), C.c_void_p(grad.data_ptr()), st)
        if rc != 0:
            raise RuntimeError(f"se_loss_sisnr_bwd failed ({rc})")
        return grad, None, None


def _lens_tensor(length, B, L, device):
    if length is None:
        return torch.full((B,), L, dtype=torch.int64, device=device)
    t = torch.as_tensor(length).to(device=device, dtype=torch.int64)
    return torch.clamp(t, 0, L)


def cal_si_snr(separated, source, length=None, eps=1e-8):
    """utility.cal_si_snr: mean over the batch of 20 log10(eps + |s_target| / (|s - s_target| + eps)) on the first length[i]
    samples, means removed."""
    if separated.is_cuda and separated.dim() == 2:
        return _SiSnrHip.apply(separated, source, _lens_tensor(length, separated.shape[0], separated.shape[1], separated.device))
    return _cal_si_snr_torch(separated, source, length, eps)


# =====================================================================================================================
# STOI loss (utility.py:821-916)
# =====================================================================================================================
def _resample_plan(orig_freq=16000, new_freq=10000, lowpass_filter_width=6):
    """Kaldi LinearResample polyphase filter (augment.py:478-545), float32 arithmetic like the reference's torch code."""
    base = math.gcd(orig_freq, new_freq)
    stride, phases = orig_freq // base, new_freq // base
    f32 = np.float32
    cutoff = 0.99 * 0.5 * min(orig_freq, new_freq)
    width = lowpass_filter_width / (2.0 * cutoff)
    out_t = np.arange(0.0, phases, dtype=f32) / f32(new_freq)
    lo = np.ceil((out_t - f32(width)) * f32(orig_freq))
    hi = np.floor((out_t + f32(width)) * f32(orig_freq))
    W = int((hi - lo + 1).max())
    idx = lo[:, None] + np.arange(W, dtype=f32)[None, :]
    dt = (idx / f32(orig_freq)) - out_t[:, None]
    w = np.zeros_like(dt)
    inside = np.abs(dt) < f32(width)
    w[inside] = (0.5 * (1 + np.cos(f32(2 * math.pi * cutoff / lowpass_filter_width) * dt[inside]))).astype(f32)
    nz = dt != 0.0
    w[nz] *= (np.sin(f32(2 * math.pi * cutoff) * dt[nz]) / (f32(math.pi) * dt[nz])).astype(f32)
    w[~nz] *= f32(2 * cutoff)
    w /= f32(orig_freq)
    return lo.astype(np.int64), w.astype(f32), stride, phases


_PLAN = {}


def _plan(device):
    key = str(device)
    if key not in _PLAN:
        first, w, stride, phases = _resample_plan()
        n = np.arange(256)
        # thirdoct(fs=10000, nfft=512, num_bands=15, min_freq=150) (utility.py:480-518)
        f = np.linspace(0, 10000, 513, dtype=np.float32)[:257]
        k = np.arange(15, dtype=np.float64)
        lo_f, hi_f = 150 * np.power(2.0, (2 * k - 1) / 6), 150 * np.power(2.0, (2 * k + 1) / 6)
        obm = np.zeros((15, 257), np.float32)
        for i in range(15):
            obm[i, int(np.argmin(np.square(f - np.float32(lo_f[i])))):int(np.argmin(np.square(f - np.float32(hi_f[i]))))] = 1
        win512 = np.zeros(512, np.float32)
        win512[128:384] = (0.5 - 0.5 * np.cos(2 * np.pi * n / 256)).astype(np.float32)  # periodic Hann(256) centred in 512
        lo_bin = [int(np.argmin(np.square(f - np.float32(lo_f[i])))) for i in range(15)]
        hi_bin = [int(np.argmin(np.square(f - np.float32(hi_f[i])))) for i in range(15)]
        _PLAN[key] = dict(first=[int(v) for v in first], w=torch.from_numpy(w).to(device), stride=stride, phases=phases,
                          hann_sym=torch.from_numpy(np.hanning(256).astype(np.float32)).to(device),  # np.hanning(256), utility.py:522
                          win512=torch.from_numpy(win512).to(device), obm=torch.from_numpy(obm).to(device),
                          # tables of the kernel form (csrc/se_stoi.hip)
                          first_dev=torch.tensor([int(v) for v in first], dtype=torch.int32, device=device),
                          hann_per=torch.from_numpy(win512[128:384].copy()).to(device),
                          band_lo=torch.tensor(lo_bin, dtype=torch.int32, device=device), band_hi=torch.tensor(hi_bin, dtype=torch.int32, device=device))
    return _PLAN[key]


def _resample_batch(x, lens, P):
    """x [B, L] (only the first lens[i] samples of row i count) -> y [B, Lo] at 10 kHz and the per-row output counts."""
    B, L = x.shape
    x = x * (torch.arange(L, device=x.device)[None, :] < lens[:, None])
    stride, phases, W = P["stride"], P["phases"], P["w"].shape[1]
    # LinearResample::GetNumOutputSamples with ticks of 1/80000 s: 5 ticks per input, 8 per output sample
    interval = lens * 5
    last = torch.div(interval, 8, rounding_mode="floor")
    n_out = torch.where(interval > 0, torch.where(last * 8 == interval, last, last + 1), torch.zeros_like(last))
    Lo = (L * 5 + 7) // 8
    padl = max(0, -min(P["first"]))
    K = (Lo + phases - 1) // phases
    padr = stride * K + W + max(P["first"]) + 8
    xp = Fn.pad(x, (padl, padr))[:, None, :]
    y = x.new_zeros(B, K * phases)
    # One banded contraction per phase, written as strided views + a matrix-vector product instead of Fn.conv1d: MIOpen's
    # strided 1-D convolution on a sliced input view read past the end of its allocation (GPU memory access fault in the
    # forward or backward of this loss, depending on where the caching allocator had placed the tensor).
    for i in range(phases):
        start = padl + P["first"][i]
        frames = xp[:, 0, start:].unfold(1, W, stride)[:, :K]  # [B, K, W] view: frames[b, k, j] = xp[b, start + k stride + j]
        y[:, i::phases] = torch.matmul(frames.double(), P["w"][i].double()).to(y.dtype)  # (STOI's silent-frame selection is sensitive to the last bits)
    y = y[:, :Lo]
    return y * (torch.arange(Lo, device=x.device)[None, :] < n_out[:, None]), n_out


def _stoi_d(y_true, y_pred, lens):
    """D[i] of utility.py:856-911 for the whole batch, on the inputs' device; differentiable w.r.t. y_pred."""
    dev = y_pred.device
    P = _plan(dev)
    B = y_pred.shape[0]
    t10, n10 = _resample_batch(y_true.float(), lens, P)
    p10, _ = _resample_batch(y_pred.float(), lens, P)
    # ---- removeSilentFrames (utility.py:521-571): 256-sample frames every 128 samples, energy from the CLEAN signal ----
    n1 = torch.div(n10, 256, rounding_mode="floor")
    n2 = torch.div(n10 - 128, 256, rounding_mode="floor")
    nf = torch.clamp(n1 + n2, min=0)
    Lo = t10.shape[1]
    if Lo < 256:
        return torch.full((B,), 0.99, device=dev)
    ft, fp = t10.unfold(1, 256, 128), p10.unfold(1, 256, 128)  # [B, Fu, 256]
    Fu = ft.shape[1]
    valid = torch.arange(Fu, device=dev)[None, :] < nf[:, None]
    w = P["hann_sym"]
    energy = 20 * torch.log10(torch.sqrt(((w ** 2)[None, None, :] * ft.detach() ** 2).sum(-1)) / 16.0 + SMALL)
    emax = torch.where(valid, energy, torch.full_like(energy, -float("inf"))).max(dim=1, keepdim=True).values
    keep = valid & ((energy - emax + 40) > 0)
    nk = keep.sum(1)
    order = torch.argsort((~keep).to(torch.int8), dim=1, stable=True)  # kept frames first, in time order
    kept = (torch.arange(Fu, device=dev)[None, :] < nk[:, None])[..., None]
    gt = torch.gather(ft * w, 1, order[..., None].expand(-1, -1, 256)) * kept
    gp = torch.gather(fp * w, 1, order[..., None].expand(-1, -1, 256)) * kept

    def ola(v):  # [first half of frame 0 | frame k first half + frame k-1 second half | second half of the last]
        return (Fn.pad(v[..., :128], (0, 0, 0, 1)) + Fn.pad(v[..., 128:], (0, 0, 1, 0))).reshape(B, -1)

    st, sp = ola(gt), ola(gp)
    Ls = 128 * (nk + 1)  # samples after silence removal
    # ---- Spectrogram(512, 256, 128, power=2): centre reflect padding at each row's OWN end ----
    n = torch.arange(st.shape[1] + 512, device=dev)[None, :] - 256
    m = n.abs()
    m = torch.where(m >= Ls[:, None], 2 * (Ls[:, None] - 1) - m, m).clamp(0, st.shape[1] - 1)
    T = nk + 2  # 1 + Ls // 128 frames
    Tm = Fu + 2
    tvalid = torch.arange(Tm, device=dev)[None, :] < T[:, None]

    def oct_env(s):
        fr = torch.gather(s, 1, m).unfold(1, 512, 128)[:, :Tm] * P["win512"]
        S = torch.fft.rfft(fr, dim=-1)
        pw = S.real ** 2 + S.imag ** 2  # [B, Tm, 257]
        return torch.sqrt(pw @ P["obm"].T + 1e-14) * tvalid[..., None]  # [B, Tm, 15]

    Ot, Op = oct_env(st), oct_env(sp)
    c = 5.62341325
    if Tm >= 30:  # 30-frame envelope vectors (utility.py:887-892)
        X = Ot.unfold(1, 30, 1).double()  # [B, Mm, 15, 30]
        Y = Op.unfold(1, 30, 1).double()
        Mi = T - 29
        mvalid = (torch.arange(X.shape[1], device=dev)[None, :] < Mi[:, None])[..., None]
        alpha = X.norm(dim=-1, keepdim=True) / (Y.norm(dim=-1, keepdim=True) + SMALL)
        yc = torch.minimum(Y * alpha, X + X * c)
        xn = X - X.mean(-1, keepdim=True)
        xn = xn / (xn.norm(dim=-1, keepdim=True) + SMALL)
        yn = yc - yc.mean(-1, keepdim=True)
        yn = yn / (yn.norm(dim=-1, keepdim=True) + SMALL)
        d_full = ((xn * yn).sum(-1) * mvalid).sum((1, 2)) / (15.0 * Mi.clamp(min=1))
    else:
        d_full = torch.zeros(B, dtype=torch.float64, device=dev)
    # fewer than 30 frames: ONE vector per band spanning all T frames (utility.py:882-885)
    Tc = min(Tm, 29)
    Xs, Ys = Ot[:, :Tc].transpose(1, 2).double(), Op[:, :Tc].transpose(1, 2).double()  # [B, 15, Tc], zero beyond T
    cnt = T.clamp(min=1, max=Tc).double()[:, None, None]
    sv = tvalid[:, None, :Tc]
    alpha = Xs.norm(dim=-1, keepdim=True) / (Ys.norm(dim=-1, keepdim=True) + SMALL)
    yc = torch.minimum(Ys * alpha, Xs + Xs * c)
    xn = (Xs - Xs.sum(-1, keepdim=True) / cnt) * sv
    xn = xn / (xn.norm(dim=-1, keepdim=True) + SMALL)
    yn = (yc - yc.sum(-1, keepdim=True) / cnt) * sv
    yn = yn / (yn.norm(dim=-1, keepdim=True) + SMALL)
    d_few = (xn * yn).sum((1, 2)) / 15.0
    d = torch.where(T >= 30, d_full, d_few).float()
    return torch.where(Ls <= 512, torch.full_like(d, 0.99), d)


STOI_KERNELS = True  # CUDA tensors: csrc/se_stoi.hip (6 + 4 launches); False: the batched torch restatement _stoi_d (~150 ops), the checker


class _StoiHip(torch.autograd.Function):
    """D[i] of utility.py:856-911 on the se_loss_stoi_* kernels; gradient to the prediction only."""

    @staticmethod
    def forward(ctx, y_true, y_pred, lens):
        from . import engine as _engine
        lib = _engine.load_library()
        P = _plan(y_pred.device)
        B, L = y_pred.shape
        yt, yp, ln = y_true.detach().contiguous().float(), y_pred.detach().contiguous().float(), lens.contiguous().to(torch.int64)
        ws = torch.empty(int(lib.se_loss_stoi_ws_floats(B, L)), dtype=torch.float32, device=y_pred.device)
        D = torch.empty(B, dtype=torch.float32, device=y_pred.device)
        st = C.c_void_p(torch.cuda.current_stream(y_pred.device).cuda_stream)
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        tabs = (p(P["w"]), p(P["first_dev"]), int(P["w"].shape[1]), p(P["hann_sym"]), p(P["hann_per"]), p(P["band_lo"]), p(P["band_hi"]))
        if lib.se_loss_stoi_fwd(p(yt), p(yp), p(ln), B, L, *tabs, p(ws), p(D), st):
            raise RuntimeError(lib.se_loss_stoi_last_error().decode())
        ctx.save_for_backward(ln, ws)
        ctx.shape = (B, L)
        ctx.dev = y_pred.device
        return D

    @staticmethod
    def backward(ctx, gD):
        from . import engine as _engine
        lib = _engine.load_library()
        ln, ws = ctx.saved_tensors
        B, L = ctx.shape
        P = _plan(ctx.dev)
        g = gD.contiguous().float()
        dpred = torch.empty(B, L, dtype=torch.float32, device=ctx.dev)
        st = C.c_void_p(torch.cuda.current_stream(ctx.dev).cuda_stream)
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        tabs = (p(P["w"]), p(P["first_dev"]), int(P["w"].shape[1]), p(P["hann_sym"]), p(P["hann_per"]), p(P["band_lo"]), p(P["band_hi"]))
        if lib.se_loss_stoi_bwd(p(g), p(ln), B, L, *tabs, p(ws), p(dpred), st):
            raise RuntimeError(lib.se_loss_stoi_last_error().decode())
        return None, dpred, None


def _stoi_ws_views(ws, B, L):
    """Named views into the kernels' workspace (the layout of csrc/se_stoi.hip:stoi_shape) - for tests and debugging."""
    Lo = (L * 5 + 7) // 8
    Fu = (Lo - 256) // 128 + 1
    Tm = Fu + 2
    off = [0]

    def take(n):
        at = off[0]
        off[0] += (n + 3) & ~3
        return at
    o = {k: take(n) for k, n in (("t10", B * Lo), ("p10", B * Lo), ("n10", B), ("nk", B), ("order", B * Fu), ("rank", B * Fu), ("Ot", B * Tm * 15),
                                 ("Op", B * Tm * 15), ("Sp", B * Tm * 257 * 2), ("dOp", B * Tm * 15), ("dfr", B * Tm * 256), ("dp10", B * Lo))}  # (+ cw, dw behind)
    return dict(nk=ws[o["nk"]:o["nk"] + B].view(torch.int32), n10=ws[o["n10"]:o["n10"] + B].view(torch.int32),
                Ot=ws[o["Ot"]:o["Ot"] + B * Tm * 15].view(B, Tm, 15), Op=ws[o["Op"]:o["Op"] + B * Tm * 15].view(B, Tm, 15),
                dOp=ws[o["dOp"]:o["dOp"] + B * Tm * 15].view(B, Tm, 15), dp10=ws[o["dp10"]:o["dp10"] + B * Lo].view(B, Lo),
                p10=ws[o["p10"]:o["p10"] + B * Lo].view(B, Lo), t10=ws[o["t10"]:o["t10"] + B * Lo].view(B, Lo))


def stoi_loss(y_true_batch, y_pred_batch, lens, reduction="mean"):
    """utility.stoi_loss: -STOI of the enhanced waveform against the clean one (batch mean, or per utterance for
    reduction != "mean").  Stays on the device of `y_pred_batch`; gradients flow to `y_pred_batch`."""
    y_pred_batch = torch.squeeze(y_pred_batch, dim=-1) if y_pred_batch.dim() == 3 else y_pred_batch
    y_true_batch = torch.squeeze(y_true_batch, dim=-1) if y_true_batch.dim() == 3 else y_true_batch
    B, L = y_pred_batch.shape
    lens_t = _lens_tensor(lens, B, L, y_pred_batch.device)
    if STOI_KERNELS and y_pred_batch.is_cuda and (L * 5 + 7) // 8 >= 256:
        D = _StoiHip.apply(y_true_batch.to(y_pred_batch.device), y_pred_batch, lens_t)
    else:
        D = _stoi_d(y_true_batch.to(y_pred_batch.device), y_pred_batch, lens_t)
    return -D.mean() if reduction == "mean" else -D


def compute_loss(source, pred_source, length):
    """TemporalCRN.compute_loss (CRN.py:593-617): (loss, stoi, sisnr) with loss = 0.7 * stoi + 0.3 * sisnr, sisnr = -SI-SNR;
    a NaN loss is replaced by zeros (CRN.py:613-616).  The reference also prints sisnr every call (CRN.py:612); dropped."""
    stoi = stoi_loss(source, pred_source, length)
    sisnr = -cal_si_snr(pred_source, source, length)
    loss = 0.7 * stoi + 0.3 * sisnr
    bad = torch.isnan(loss)
    zero = torch.zeros_like(loss)  # no gradient, like the reference's fill_(0.0)
    return torch.where(bad, zero, loss), torch.where(bad, zero, stoi), torch.where(bad, zero, sisnr)

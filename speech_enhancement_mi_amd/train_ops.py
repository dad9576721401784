"""torch.autograd.Function wrappers over the hand-written training kernels (C ABI `se_train_*`, csrc/train_ops.inc.h).

Replaces, in the timed region of the DP training step (reference train.py:195-204), torch's `conv2d` / `conv_transpose2d`
/ `nn.GRU` / `nn.Linear` forward AND backward (MIOpen / rocBLAS) for the blocks of CRN.py:290-401 and 196-287:

  conv_block(x, xprev, W, b, d)      TemporalConv2d convolution (5x3, stride (2,1), causal dilation d, history rows `xprev`)
  deconv_block(x, W, b, d)           TemporalConvTranspose2d convolution, last T columns kept
  linear(x, W, b)                    x W^T + b
  gru_layer(x, h0, W_ih, W_hh, ...)  one GRU layer over T steps (BPTT in the backward)

Activations are [B, C, T, F] (F innermost, the engine's layout).  Everything is fp32-exact MFMA arithmetic, so the gradients
agree with torch autograd to rounding (tests/test_gpu_round2.py).  There is no CPU path: CPU tensors raise.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import engine as _engine


def _lib():
    return _engine.load_library()


def _p(t):
    return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(rc):
    if rc != 0:
        raise RuntimeError(f"se_train error {rc}: {_lib().se_train_last_error().decode()}")


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("the hand-written training kernels run on the GPU only (no CPU fallback)")


_layout_cache = {}

# Optional per-kernel timing for bench.py's training roofline: PROF = {} switches it on; every launch is bracketed by two
# torch.cuda events on the current stream (the stream the kernels are enqueued on).  profile_summary() folds them.
PROF = None


class _Timed:
    def __init__(self, kernel, flops):
        self.k, self.f = kernel, flops

    def __enter__(self):
        if PROF is not None:
            self.a, self.b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if PROF is not None:
            self.b.record()
            PROF.setdefault(self.k, []).append((self.a, self.b, self.f))


def profile_summary():
    torch.cuda.synchronize()
    out = {}
    for k, recs in (PROF or {}).items():
        ms = sum(a.elapsed_time(b) for a, b, _ in recs)
        out[k] = dict(ms=ms, launches=len(recs), flops=sum(f for _, _, f in recs))
    return out


def _layout(kind, Ci, Co, T, Fi, Fy, d):
    key = (kind, Ci, Co, T, Fi, Fy, d)
    if key not in _layout_cache:
        lay = _engine.TrainConvLayout()
        _chk(_lib().se_train_conv_layout_query(kind, Ci, Co, T, Fi, Fy, d, C.byref(lay)))
        _layout_cache[key] = (lay.ntap, lay.CC, lay.nchunk, lay.CoPad, list(lay.tap_kf)[:lay.ntap], list(lay.tap_kt)[:lay.ntap])
    return _layout_cache[key]


def _arrange(w_co_ci, lay):
    """w_co_ci [Co, Ci, 5, 3] (GEMM rows = first index) -> [nchunk][ntap][CC][CoPad] fp32, zero padded."""
    ntap, CC, nchunk, CoPad, kf, kt = lay
    Co, Ci = w_co_ci.shape[:2]
    taps = w_co_ci[:, :, kf, kt]  # [Co, Ci, ntap]
    out = w_co_ci.new_zeros(nchunk * CC, ntap, CoPad)
    out[:Ci, :, :Co] = taps.permute(1, 2, 0)
    return out.reshape(nchunk, CC, ntap, CoPad).permute(0, 2, 1, 3).contiguous()


def _conv_launch(kind, x, xprev, w_co_ci, bias, y, d, act=0):
    B, Ci, T, Fi = x.shape
    Co, Fy = y.shape[1], y.shape[3]
    lay = _layout(kind, Ci, Co, T, Fi, Fy, d)
    wa = _arrange(w_co_ci, lay)
    FP = Fy if kind == 0 else ((Fy + 1) // 2 if kind == 1 else Fy // 2)
    with _Timed("k_conv_igemm", 2.0 * B * Co * Ci * lay[0] * T * FP):
        _chk(_lib().se_train_conv(kind, _p(x), _p(xprev), _p(wa), _p(bias), _p(y), B, Ci, Co, T, Fi, Fy, d, act, _st()))


def _strided_conv(x, xprev, w, bias, d):  # w [Co, Ci, 5, 3]
    B, Ci, T, Fi = x.shape
    Co, Fo = w.shape[0], (Fi - 1) // 2 + 1
    y = torch.empty(B, Co, T, Fo, device=x.device, dtype=torch.float32)
    _conv_launch(0, x, xprev, w, bias, y, d)
    return y


def _transposed_conv(x, w_ci_co, bias, d, Fy):  # w [Ci, Co, 5, 3] (torch ConvTranspose2d layout)
    B, Ci, T, Fi = x.shape
    Co = w_ci_co.shape[1]
    y = torch.empty(B, Co, T, Fy, device=x.device, dtype=torch.float32)
    wt = w_ci_co.permute(1, 0, 2, 3)  # GEMM rows = output channels
    _conv_launch(1, x, None, wt, bias, y, d)
    _conv_launch(2, x, None, wt, bias, y, d)
    return y


def _wgrad(G, S, Sprev, d):
    B, Ca, T, Fm = G.shape
    Cb, Fs = S.shape[1], S.shape[3]
    out = torch.empty(Ca, Cb, 5, 3, device=G.device, dtype=torch.float32)
    with _Timed("k_corr_wgrad", 2.0 * B * Ca * Cb * 15 * T * Fm):
        _chk(_lib().se_train_conv_wgrad(_p(G), _p(S), _p(Sprev), _p(out), B, Ca, Cb, T, Fm, Fs, d, _st()))
    return out


class _ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, xprev, w, b, d):
        _need_gpu(x, w)
        x = x.contiguous()
        xprev = None if xprev is None else xprev.contiguous()
        y = _strided_conv(x, xprev, w.contiguous(), b.contiguous(), d)
        ctx.save_for_backward(x, xprev if xprev is not None else x.new_empty(0), w)
        ctx.d, ctx.has_prev = d, xprev is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, xprev, w = ctx.saved_tensors
        dy = dy.contiguous()
        d = ctx.d
        zero_b = dy.new_zeros(w.shape[1])
        # d/dx: transposed convolution of dy with the same weight tensor read as [Cin' = Co][Cout' = Ci]
        dx = _transposed_conv(dy, w, zero_b, d, x.shape[3]) if ctx.needs_input_grad[0] else None
        dw = _wgrad(dy, x, xprev if ctx.has_prev else None, d) if ctx.needs_input_grad[2] else None
        db = dy.sum((0, 2, 3)) if ctx.needs_input_grad[3] else None
        return dx, None, dw, db, None


class _DeconvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, d):
        _need_gpu(x, w)
        x = x.contiguous()
        y = _transposed_conv(x, w.contiguous(), b.contiguous(), d, 2 * x.shape[3] - 1)
        ctx.save_for_backward(x, w)
        ctx.d = d
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        d = ctx.d
        dx = None
        if ctx.needs_input_grad[0]:  # strided causal convolution of dy (no history) with the weights read as [Cout' = Ci][Cin' = Co]
            dx = _strided_conv(dy, None, w, dy.new_zeros(w.shape[0]), d)
        dw = _wgrad(x, dy, None, d) if ctx.needs_input_grad[1] else None
        db = dy.sum((0, 2, 3)) if ctx.needs_input_grad[2] else None
        return dx, dw, db, None


def _gemm(A, W, bias=None, act=0):
    """act(A [M, K] @ W [N, K]^T + bias)"""
    A, W = A.contiguous(), W.contiguous()
    M, K = A.shape
    N = W.shape[0]
    if K % 8:  # the kernel walks K in 8-deep blocks: pad with zeros (weight-gradient GEMMs contract over B*T rows)
        pad = 8 - K % 8
        A = torch.nn.functional.pad(A, (0, pad))
        W = torch.nn.functional.pad(W, (0, pad))
        K += pad
    out = torch.empty(M, N, device=A.device, dtype=torch.float32)
    with _Timed("k_gemm_skinny", 2.0 * M * N * K):
        _chk(_lib().se_train_gemm(_p(A), _p(W), _p(bias), _p(out), M, N, K, act, _st()))
    return out


def _gemm_tn(A, B):
    """sum_r A[r, :]^T B[r, :] -> [Na, Nb] (row-major operands; the contraction runs over the rows)"""
    A, B = A.contiguous(), B.contiguous()
    R, Na = A.shape
    Nb = B.shape[1]
    out = torch.empty(Na, Nb, device=A.device, dtype=torch.float32)
    with _Timed("k_gemm_tn_acc", 2.0 * R * Na * Nb):
        _chk(_lib().se_train_gemm_tn(_p(A), _p(B), _p(out), R, Na, Nb, _st()))
    return out


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        _need_gpu(x, w)
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        ctx.save_for_backward(x2, w)
        ctx.shape = x.shape
        return _gemm(x2, w, b.contiguous()).reshape(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1]).contiguous()
        dx = _gemm(dy2, w.t().contiguous()).reshape(ctx.shape) if ctx.needs_input_grad[0] else None
        dw = _gemm_tn(dy2, x2) if ctx.needs_input_grad[1] else None
        db = dy2.sum(0) if ctx.needs_input_grad[2] else None
        return dx, dw, db


_pseq_scratch = {}


def _scratch(dev, B, H, tag=0):
    """Exchange buffer + arrival / timeout words of the persistent GRU launches (stream-ordered reuse; `tag` separates launches that
    run concurrently on different streams)."""
    key = (dev.index, B, H, tag)
    t = _pseq_scratch.get(key)
    if t is None:
        t = _pseq_scratch[key] = torch.zeros(_lib().se_train_gru_pseq_scratch_floats(B, H), device=dev, dtype=torch.float32)
    return t


def pseq_check():
    """Raises if any persistent GRU launch since the last check gave up its bounded spin (one host sync; train_step calls it
    where it reads the loss value anyway)."""
    for key, t in _pseq_scratch.items():
        if int(t[:4].view(torch.int32)[1]) != 0:
            t[:4].zero_()
            raise RuntimeError(f"persistent GRU kernel timed out waiting for its peer workgroups (device {key[0]}, B = {key[1]}, H = {key[2]})")


def _gru_seq_fwd(gi, h0, w_hh, b_hh, out, gates, hT, B, T, H, Tseg, ldN, ldB):
    """All T steps of one layer.  gi / out / gates rows are addressed as row(b, s) = (s // Tseg) * ldN + b * ldB + s % Tseg."""
    lib = _lib()
    if lib.se_train_gru_pseq_supported(B, H) and (B <= 32 or ldN == 0):  # ONE persistent launch (groups of <= 32 streams inside it)
        sc = _scratch(gi.device, B, H)
        with _Timed("k_gru_pseq_fwd", 2.0 * B * 3 * H * H * T):
            _chk(lib.se_train_gru_pseq_fwd(_p(gi), _p(h0), _p(w_hh), _p(b_hh), _p(out), _p(gates), _p(hT), _p(sc), B, T, H, Tseg, ldN, ldB, _st()))
        return
    if Tseg != T or ldB != T:
        raise RuntimeError(f"hidden size {H}: no persistent GRU kernel, and the step-launch kernels take [B][T] rows only")
    for b0 in range(0, B, 16):  # step-launch kernels (round 2): groups of <= 16 streams
        nb = min(16, B - b0)
        scratch = torch.empty(2, nb, H, device=gi.device, dtype=torch.float32)
        with _Timed("k_gru_step", 2.0 * nb * 3 * H * H * T):
            _chk(lib.se_train_gru_seq_fwd(C.c_void_p(gi.data_ptr() + 4 * b0 * T * 3 * H), C.c_void_p(h0.data_ptr() + 4 * b0 * H), _p(w_hh), _p(b_hh),
                                          C.c_void_p(out.data_ptr() + 4 * b0 * T * H), C.c_void_p(gates.data_ptr() + 4 * b0 * T * 4 * H),
                                          C.c_void_p(hT.data_ptr() + 4 * b0 * H), _p(scratch), nb, T, H, _st()))


def _gru_seq_bwd(dout, dhT, gates, out, h0, w_hh_t, dgi, dgh, B, T, H, Tseg, ldN, ldB, seg_len):
    lib = _lib()
    if lib.se_train_gru_pseq_supported(B, H) and (B <= 32 or ldN == 0):
        sc = _scratch(dout.device, B, H)
        with _Timed("k_gru_pseq_bwd", 2.0 * B * 3 * H * H * T):
            _chk(lib.se_train_gru_pseq_bwd(_p(dout), _p(dhT), _p(gates), _p(out), _p(h0), _p(w_hh_t), _p(dgi), _p(dgh), _p(sc), B, T, H, Tseg, ldN, ldB,
                                           seg_len, _st()))
        return
    if Tseg != T or ldB != T:
        raise RuntimeError(f"hidden size {H}: no persistent GRU kernel, and the step-launch kernels take [B][T] rows only")
    for b0 in range(0, B, 16):
        nb = min(16, B - b0)
        scratch = torch.empty(4, nb, H, device=out.device, dtype=torch.float32)
        dh = None if dhT is None else C.c_void_p(dhT.data_ptr() + 4 * b0 * H)
        with _Timed("k_gru_bwd_step", 2.0 * nb * 3 * H * H * T):
            _chk(lib.se_train_gru_seq_bwd(C.c_void_p(dout.data_ptr() + 4 * b0 * T * H), dh, C.c_void_p(gates.data_ptr() + 4 * b0 * T * 4 * H),
                                          C.c_void_p(out.data_ptr() + 4 * b0 * T * H), C.c_void_p(h0.data_ptr() + 4 * b0 * H), _p(w_hh_t),
                                          C.c_void_p(dgi.data_ptr() + 4 * b0 * T * 3 * H), C.c_void_p(dgh.data_ptr() + 4 * b0 * T * 3 * H), _p(scratch),
                                          nb, T, H, seg_len, _st()))


class _GruLayerFn(torch.autograd.Function):
    """One GRU layer, batch_first: x [B, T, In], h0 [B, H] (a constant: the carried state is detached, CRN.py:281) ->
    (out [B, T, H], hT [B, H]).  seg_len > 0: the sequence is a chain of segments of seg_len steps whose carried state is
    detached at every seam, so the backward sweep drops the gradient that would cross a seam (truncated BPTT exactly as the
    per-segment loop of CRN.py:577-586 does it).  The time loop runs inside ONE persistent launch per direction
    (se_train_gru_pseq_*, csrc/gru_pseq.hip.h); hidden sizes that kernel does not cover fall back to one launch per step."""

    @staticmethod
    def forward(ctx, x, h0, w_ih, w_hh, b_ih, b_hh, seg_len):
        _need_gpu(x, w_ih)
        B, T, In = x.shape
        H = w_hh.shape[1]
        x2 = x.reshape(B * T, In).contiguous()
        gi = _gemm(x2, w_ih, b_ih.contiguous())  # [B*T, 3H]
        out = torch.empty(B, T, H, device=x.device, dtype=torch.float32)
        gates = torch.empty(B, T, 4 * H, device=x.device, dtype=torch.float32)
        hT = torch.empty(B, H, device=x.device, dtype=torch.float32)
        h0c, w_hh_c, b_hh_c = h0.contiguous(), w_hh.contiguous(), b_hh.contiguous()
        _gru_seq_fwd(gi, h0c, w_hh_c, b_hh_c, out, gates, hT, B, T, H, T, 0, T)
        ctx.save_for_backward(x2, h0c, w_ih, w_hh_c, out, gates)
        ctx.dims = (B, T, In, H, int(seg_len))
        return out, hT

    @staticmethod
    def backward(ctx, dout, dhT):
        x2, h0, w_ih, w_hh, out, gates = ctx.saved_tensors
        B, T, In, H, seg_len = ctx.dims
        dout = dout.contiguous()
        dgi = torch.empty(B, T, 3 * H, device=out.device, dtype=torch.float32)
        dgh = torch.empty(B, T, 3 * H, device=out.device, dtype=torch.float32)
        w_hh_t = w_hh.t().contiguous()  # [H, 3H]: dh_{t-1} += dgh W_hh with K-contiguous operands
        _gru_seq_bwd(dout, dhT.contiguous() if dhT is not None else None, gates, out, h0, w_hh_t, dgi, dgh, B, T, H, T, 0, T, seg_len)
        dgi2, dgh2 = dgi.reshape(B * T, 3 * H), dgh.reshape(B * T, 3 * H)
        hprev_all = torch.cat([h0[:, None], out[:, :-1]], dim=1).reshape(B * T, H)
        dx = _gemm(dgi2, w_ih.t().contiguous()).reshape(B, T, In) if ctx.needs_input_grad[0] else None
        dw_ih = _gemm_tn(dgi2, x2)
        dw_hh = _gemm_tn(dgh2, hprev_all)
        return dx, None, dw_ih, dw_hh, dgi2.sum(0), dgh2.sum(0), None


def conv_block(x, xprev, w, b, dilation):
    return _ConvFn.apply(x, xprev, w, b, dilation)


def deconv_block(x, w, b, dilation):
    return _DeconvFn.apply(x, w, b, dilation)


def linear(x, w, b):
    return _LinearFn.apply(x, w, b)


def gru_layer(x, h0, w_ih, w_hh, b_ih, b_hh, seg_len=0):
    return _GruLayerFn.apply(x, h0, w_ih, w_hh, b_ih, b_hh, seg_len)

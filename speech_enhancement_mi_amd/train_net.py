"""The whole differentiable `realtime_process` of TemporalCRN (CRN.py:560-589 over 454-496) on hand-written kernels, forward
AND backward, as ONE torch.autograd.Function (reference training step train.py:195-204).

Round 2 composed per-op autograd Functions (conv / GRU / dense) with torch glue for the norms, gates, features, mask, STFT and
iSTFT (about a third of the step, `at::native::*` kernels).  Here every stage is a `se_train_*` / `se_sig_*` launch
(csrc/train_fused.hip.h, gru_pseq.hip.h, train_ops.inc.h); PyTorch only allocates tensors, and autograd sees a single node whose
backward returns the parameter gradients.  No float atomics: gradients are bit-reproducible.

Layout: S = N segments x B utterances, SEGMENT-major.  An encoder block's time history (the reference's `buffer`,
CRN.py:325-337 = the previous segment's detached input) is then the same tensor one slab of B streams earlier, so each layer
runs once over all S streams with `xprev = x - one slab`; slab 0 of every input tensor holds the carried state (zeros after a
reset).  The GRU runs once per layer over the N*T steps of every utterance (one persistent launch), with the BPTT cut at
segment seams exactly where the reference detaches `h` (CRN.py:281).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import engine as _engine
from . import train_ops as K

_sig_cache = {}
_side_streams = {}
PIPELINE_LAYERS = True   # training forward: GRU layers as a wavefront over segments on one HIP stream per layer (False: layer after layer)


def _side_stream(dev, idx):
    key = (dev.index, idx)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=dev)
    return _side_streams[key]


def _sig(dev, n_fft, win, hop, seg):
    key = (dev.index, n_fft, win, hop, seg)
    if key not in _sig_cache:
        h = C.c_void_p()
        K._chk(K._lib().se_sig_create(n_fft, win, hop, seg, dev.index or 0, C.byref(h)))
        _sig_cache[key] = h
    return _sig_cache[key]


def _p(t, off_floats=0):
    return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr() + 4 * off_floats)


def _new(*shape, dev):
    return torch.empty(*shape, device=dev, dtype=torch.float32)


def _run(name, flops, fn, *args):
    with K._Timed(name, flops):
        K._chk(fn(*args))


# ---- thin launch helpers ---------------------------------------------------------------------------------------------------------
def conv_w(kind, x_ptr, xprev_ptr, w, sCo, sCi, bias, y, S, Ci, Co, T, Fi, Fy, d, act=0, Cy=0, cy0=0):
    lib = K._lib()
    n = lib.se_train_conv_ws_floats(kind, Ci, Co, T, Fi, Fy, d)
    if n < 0:
        K._chk(n)
    ws = _new(n, dev=y.device)
    FP = Fy if kind in (0, 3) else ((Fy + 1) // 2 if kind == 1 else Fy // 2)
    ntap = {0: 15, 1: 9, 2: 6, 3: 1}[kind]
    _run("k_conv_igemm", 2.0 * S * Co * Ci * ntap * T * FP, lib.se_train_conv_w, kind, x_ptr, xprev_ptr, _p(w), sCo, sCi, _p(bias), _p(y), _p(ws),
         S, Ci, Co, T, Fi, Fy, d, act, Cy, cy0, K._st())


def wgrad(G, Sx, Sprev_ptr, S, Ca, Cb, T, Fm, Fs, d, ntap):
    """Deterministic weight gradient [Ca][Cb][ntap]: partial tiles per row split + a fixed-order fold."""
    lib = K._lib()
    n = Ca * Cb * ntap
    ws = _new(64 * n, dev=G.device)
    ns = C.c_int(0)
    _run("k_corr_wgrad", 2.0 * S * Ca * Cb * ntap * T * Fm, lib.se_train_conv_wgrad_det, _p(G), _p(Sx) if isinstance(Sx, torch.Tensor) else Sx, Sprev_ptr,
         _p(ws), C.byref(ns), S, Ca, Cb, T, Fm, Fs, d, ntap, K._st())
    out = _new(n, dev=G.device)
    _run("k_colsum", 0.0, lib.se_train_colsum, _p(ws), _p(out), n, None, None, 0, None, None, 0, ns.value, 0, K._st())
    return out


def gemm_tn(A, Bm):
    """sum_r A[r, :]^T B[r, :] -> [Na, Nb], deterministic."""
    lib = K._lib()
    R, Na = A.shape
    Nb = Bm.shape[1]
    ws = _new(64 * Na * Nb, dev=A.device)
    ns = C.c_int(0)
    _run("k_gemm_tn_acc", 2.0 * R * Na * Nb, lib.se_train_gemm_tn_det, _p(A), _p(Bm), _p(ws), C.byref(ns), R, Na, Nb, K._st())
    out = _new(Na, Nb, dev=A.device)
    _run("k_colsum", 0.0, lib.se_train_colsum, _p(ws), _p(out), Na * Nb, None, None, 0, None, None, 0, ns.value, 0, K._st())
    return out


def colsum3(R, *pairs):
    """pairs = (part [R, n], n) ...: returns the column sums (fixed order)."""
    lib = K._lib()
    outs = [_new(n, dev=p.device) for p, n in pairs]
    a = []
    for k in range(3):
        if k < len(pairs):
            a += [_p(pairs[k][0]), _p(outs[k]), pairs[k][1]]
        else:
            a += [None, None, 0]
    _run("k_colsum", 0.0, lib.se_train_colsum, *a, R, 0, K._st())
    return outs


def colsum_tall(x):
    lib = K._lib()
    R, n = x.shape
    ws = _new((R + 63) // 64, n, dev=x.device)
    out = _new(n, dev=x.device)
    _run("k_colsum", 0.0, lib.se_train_colsum_tall, _p(x), R, n, _p(ws), _p(out), 0, K._st())
    return out


def gln_fwd(x, xs, y_ptr, ys, w, b, S, Cc, T, Fi, Fo, mode, act, eps_mode=0):
    stats = _new(S, 2, dev=w.device)
    _run("k_tgln_fwd", 0.0, K._lib().se_train_gln_fwd, _p(x), *xs, y_ptr, *ys, _p(w), _p(b), _p(stats), S, Cc, T, Fi, Fo, mode, act, eps_mode, K._st())
    return stats


def gln_bwd(dy_ptr, ds, x, xs, w, stats, S, Cc, T, Fi, mode, act, eps_mode=0):
    """-> dx (same shape / strides as x), dw, db, dpre (column sums of the [S][NA] slabs)"""
    NA = Cc * Fi if mode else Cc
    dev = x.device
    dx = torch.empty_like(x)
    parts = [_new(S, NA, dev=dev) for _ in range(3)]
    _run("k_tgln_bwd", 0.0, K._lib().se_train_gln_bwd, dy_ptr, *ds, _p(x), *xs, _p(dx), _p(w), _p(stats), _p(parts[0]), _p(parts[1]), _p(parts[2]),
         S, Cc, T, Fi, mode, act, eps_mode, K._st())
    dw, db, dpre = colsum3(S, (parts[0], NA), (parts[1], NA), (parts[2], NA))
    return dx, dw, db, dpre


def transpose(w):
    """[R, C] -> [C, R] contiguous (weights only: tiny)."""
    return w.t().contiguous()


class CRNFunction(torch.autograd.Function):
    """pred = realtime_process(mixture) for the CRN.py (variant 0) and CRN_ELU.py (variant 1, the model train.py:16 trains) networks.
    forward(ctx, model, mixture, flag, *params)."""

    @staticmethod
    def forward(ctx, model, mixture, flag, *params):
        lib = K._lib()
        K._need_gpu(mixture, params[0])
        dev = mixture.device
        mixture = mixture.contiguous()
        B, M, L = mixture.shape
        c = model._cfg_args
        Ks = model.segment_length
        P = Ks // 2
        hop = int(round(c["sample_rate"] / 1000.0 * c["hop_length"]))
        win = int(round(c["sample_rate"] / 1000.0 * c["win_length"]))
        n_fft = c["n_fft"]
        T, F0 = 1 + Ks // hop, n_fft // 2 + 1
        sig = _sig(dev, n_fft, win, hop, Ks)
        Lp = L if flag else L + P
        off0 = -P if flag else -2 * P
        skip = 0 if flag else P
        gap = Ks - (P + Lp % Ks) % Ks
        N = 2 * (Lp + gap + P) // Ks
        S = N * B
        st = K._st
        state = model._state if flag else None
        Lv = len(model.convlist)
        ch = [2 * M - 1] + [blk.conv.weight.shape[0] for blk in model.convlist]
        Fq = [F0]
        for _ in range(Lv):
            Fq.append((Fq[-1] - 1) // 2 + 1)
        sv = {}  # saved for backward

        spec = _new(N, B * M, T, F0, 2, dev=dev)
        _run("k_stft", 0.0, lib.se_sig_stft, sig, _p(mixture), B, M, L, off0, P, N, _p(spec), st())
        V = model._VARIANT            # 0 = CRN.py (ReLU); 1 = CRN_ELU.py (ELU, gated 1x1 pair per block, three 5x5 pre-conv blocks, atan2 phase)
        if V not in (0, 1):
            raise NotImplementedError("the training kernels cover CRN.py and CRN_ELU.py (the student is trained by distillation, out of scope)")
        act = 2 if V else 1

        def first_slab(t, key, idx):   # slab 0 = the carried state of a flag=True continuation, zeros after a reset
            t[0].copy_(state[key][idx]) if state is not None and state.get(key) is not None else t[0].zero_()

        def gate_pair(a_t, blk, Co, Fo, y_ptr, ys):
            """CRN_ELU.py:240-241: conv_trans(a) * sigmoid(conv_gated(a)) -> gLN.  Two 1x1 launches into the halves of tg."""
            tg = _new(S, 2 * Co, T, Fo, dev=dev)
            conv_w(3, _p(a_t), None, blk.conv_trans.weight, Co, 1, blk.conv_trans.bias, tg, S, Co, Co, T, Fo, Fo, 0, 0, 2 * Co, 0)
            conv_w(3, _p(a_t), None, blk.conv_gated.weight, Co, 1, blk.conv_gated.bias, tg, S, Co, Co, T, Fo, Fo, 0, 0, 2 * Co, Co)
            stt = _new(S, 2, dev=dev)
            _run("k_tgate_fwd", 0.0, lib.se_train_gate_fwd, _p(tg), _p(blk.norm.weight), _p(blk.norm.bias), y_ptr, *ys, _p(stt), S, Co, T, Fo, 0, st())
            return tg, stt

        xin = []
        C0 = ch[0]
        x_full = _new(N + 1, B, C0, T, F0, dev=dev)
        slab0 = B * C0 * T * F0
        first_slab(x_full, "pbuf" if V else "buf", 0)
        _run("k_tfeat", 0.0, lib.se_train_feat, _p(spec), _p(x_full, slab0), S, M, T, F0, 1 if V else 0, st())
        pre = []
        if V:  # x = block(x) + x, three times (CRN_ELU.py:375-376)
            cur = x_full
            for k, blk in enumerate(model.preconvlist):
                a_t = _new(S, C0, T, F0, dev=dev)
                _run("k_pre5", 2.0 * S * C0 * C0 * 25 * T * F0, lib.se_train_pre5, 0, _p(cur, slab0), _p(cur), _p(blk.conv.weight), _p(blk.conv.bias), None, _p(a_t),
                     S, C0, T, F0, 2 ** k, 2, st())
                out = _new(S, C0, T, F0, dev=dev)
                tg, stt = gate_pair(a_t, blk, C0, F0, _p(out), (C0 * T * F0, T * F0, F0))
                nxt = _new(N + 1, B, C0, T, F0, dev=dev)
                first_slab(nxt, "pbuf", k + 1) if k < 2 else first_slab(nxt, "buf", 0)
                _run("k_tadd", 0.0, lib.se_train_add3, _p(nxt, slab0), _p(out), _p(cur, slab0), S * C0 * T * F0, st())
                pre.append(dict(cur=cur, a=a_t, tg=tg, st=stt))
                cur = nxt
            x_full = cur
        xin.append(x_full)
        ys, stats_e, enc_tg = [], [], []
        seq = None
        for i, blk in enumerate(model.convlist):
            Ci, Co, Fi, Fo, d = ch[i], ch[i + 1], Fq[i], Fq[i + 1], 2 ** i
            slab = B * Ci * T * Fi
            y = _new(S, Co, T, Fo, dev=dev)   # variant 0: the pre-activation; variant 1: ELU(conv), all the backward needs
            conv_w(0, _p(xin[i], slab), _p(xin[i]), blk.conv.weight, Ci * 15, 15, blk.conv.bias, y, S, Ci, Co, T, Fi, Fo, d, 2 if V else 0)
            ys.append(y)
            if i < Lv - 1:
                nxt = _new(N + 1, B, Co, T, Fo, dev=dev)
                first_slab(nxt, "buf", i + 1)
                y_ptr, ysd = _p(nxt, B * Co * T * Fo), (Co * T * Fo, T * Fo, Fo)
                xin.append(nxt)
            else:  # the last block feeds the GRU: [S][T][D], feature index c * F + f (CRN.py:476-478)
                D = Co * Fo
                seq = _new(S * T, D, dev=dev)
                y_ptr, ysd = _p(seq), (T * D, Fo, D)
            if V:
                tg, stt = gate_pair(y, blk, Co, Fo, y_ptr, ysd)
                enc_tg.append(tg)
                stats_e.append(stt)
            else:
                stats_e.append(gln_fwd(y, (Co * T * Fo, T * Fo, Fo), y_ptr, ysd, blk.norm.weight, blk.norm.bias, S, Co, T, Fo, Fo, 0, 1))
        CL, FL = ch[Lv], Fq[Lv]
        D = CL * FL
        g = model.gru.sequence_model
        H, NL = g.hidden_size, g.num_layers
        R = S * T
        outs, gates, h0s, hTs = [], [], [], []
        gi0 = K._gemm(seq, g.weight_ih_l0, g.bias_ih_l0)
        for l in range(NL):
            h0s.append(state["h"][l] if state is not None and state["h"] is not None else torch.zeros(B, H, device=dev))
            outs.append(_new(R, H, dev=dev)); gates.append(_new(R, 4 * H, dev=dev))
        if NL == 1 or N < 4 or not PIPELINE_LAYERS:
            layer_in = seq
            for l in range(NL):
                gi = gi0 if l == 0 else K._gemm(layer_in, getattr(g, f"weight_ih_l{l}"), getattr(g, f"bias_ih_l{l}"))
                hT = _new(B, H, dev=dev)
                K._gru_seq_fwd(gi, h0s[l], getattr(g, f"weight_hh_l{l}"), getattr(g, f"bias_hh_l{l}"), outs[l], gates[l], hT, B, N * T, H, T, B * T, T)
                hTs.append(hT)
                layer_in = outs[l]
        else:
            # Layer wavefront over segments: the recurrence is sequential in time, but layer l of segment n only needs layer l - 1 of the
            # SAME segment.  Every layer gets its own HIP stream and walks the utterance one segment (T steps, one persistent launch)
            # at a time; layer l's segment n waits for an event of layer l - 1's segment n, so the layers run one segment apart:
            # (N + NL - 1) x T dependent steps instead of NL x N x T.
            cur = torch.cuda.current_stream()
            streams = [cur] + [_side_stream(dev, l) for l in range(1, NL)]
            hall = [_new(N, B, H, dev=dev) for _ in range(NL)]            # the state after every segment (the next segment's h0)
            gis = [gi0] + [_new(R, 3 * H, dev=dev) for _ in range(1, NL)]
            fork = torch.cuda.Event()
            fork.record(cur)
            for l in range(1, NL):
                streams[l].wait_event(fork)
            done = [[None] * N for _ in range(NL)]
            rows = B * T
            for n in range(N):
                for l in range(NL):
                    with torch.cuda.stream(streams[l]):
                        if l > 0:
                            streams[l].wait_event(done[l - 1][n])
                            _run("k_gemm_skinny", 2.0 * rows * 3 * H * H, lib.se_train_gemm, _p(outs[l - 1], n * rows * H), _p(getattr(g, f"weight_ih_l{l}")),
                                 _p(getattr(g, f"bias_ih_l{l}")), _p(gis[l], n * rows * 3 * H), rows, 3 * H, H, 0, K._st())
                        sc = K._scratch(dev, B, H, tag=l)
                        h_in = h0s[l] if n == 0 else hall[l][n - 1]
                        _run("k_gru_pseq_fwd", 2.0 * B * 3 * H * H * T, lib.se_train_gru_pseq_fwd, _p(gis[l], n * rows * 3 * H), _p(h_in),
                             _p(getattr(g, f"weight_hh_l{l}")), _p(getattr(g, f"bias_hh_l{l}")), _p(outs[l], n * rows * H), _p(gates[l], n * rows * 4 * H),
                             _p(hall[l][n]), _p(sc), B, T, H, T, 0, T, K._st())
                        ev = torch.cuda.Event()
                        ev.record(streams[l])
                        done[l][n] = ev
            for l in range(1, NL):
                cur.wait_event(done[l][N - 1])
            hTs = [hall[l][N - 1] for l in range(NL)]
            layer_in = outs[NL - 1]
        fc = model.gru.fc_output_layer
        o_fc = K._gemm(layer_in, fc.weight, fc.bias)  # [R, D] pre-activation
        xd = _new(S, CL, T, FL, dev=dev)
        st_fc = gln_fwd(o_fc, (T * D, FL, D), _p(xd), (CL * T * FL, T * FL, FL), model.gru.norm.weight, model.gru.norm.bias, S, CL, T, FL, FL, 1, act)
        dec = []
        x_in = xd
        Ci, Fi = CL, FL
        for j, blk in enumerate(model.deconvlist):
            Co, d, Fy = blk.conv.weight.shape[1], 2 ** j, 2 * Fi - 1
            yd = _new(S, Co, T, Fy, dev=dev)
            for kind in (1, 2):
                conv_w(kind, _p(x_in), None, blk.conv.weight, 15, Co * 15, blk.conv.bias, yd, S, Ci, Co, T, Fi, Fy, d)
            rec = dict(x_in=x_in, yd=yd, Ci=Ci, Co=Co, Fi=Fi, Fy=Fy, d=d)
            if j < Lv - 1:
                k = Lv - 1 - j  # skip tensor = encoder output x_k (CRN.py:485: residuals[-2-j])
                Cr, Fr = ch[k], Fq[k]
                if Fr < Fy or Cr != Co:
                    raise RuntimeError("decoder / skip geometry outside the reference's (CRN.py:389-392 crop branch is never taken)")
                z = _new(S, Co, T, Fr, dev=dev)
                rec["st"] = gln_fwd(yd, (Co * T * Fy, T * Fy, Fy), _p(z), (Co * T * Fr, T * Fr, Fr), blk.norm.weight, blk.norm.bias, S, Co, T, Fy, Fr, 0, act)
                wuv = _new(2 * Co, Cr, dev=dev)
                buv = _new(2 * Co, dev=dev)
                wuv[:Co].copy_(blk.residual.weight.view(Co, Cr)); wuv[Co:].copy_(blk.residualmask.weight.view(Co, Cr))
                buv[:Co].copy_(blk.residual.bias); buv[Co:].copy_(blk.residualmask.bias)
                uv = _new(S, 2 * Co, T, Fr, dev=dev)
                res_off = B * Cr * T * Fr
                conv_w(3, _p(xin[k], res_off), None, wuv, Cr, 1, buv, uv, S, Cr, 2 * Co, T, Fr, Fr, 0)
                out = _new(S, Co, T, Fr, dev=dev)
                st_uv = _new(S, 2, dev=dev)
                _run("k_tskip_fwd", 0.0, lib.se_train_skip_fwd, _p(uv), _p(z), _p(blk.residualnorm.weight), _p(blk.residualnorm.bias), _p(out), _p(st_uv),
                     S, Co, T, Fr, act, 0, st())
                rec.update(z=z, uv=uv, wuv=wuv, st_uv=st_uv, k=k, Cr=Cr, Fr=Fr)
                x_in, Ci, Fi = out, Co, Fr
            else:
                xl = _new(S, Co, T, Fy, dev=dev)
                rec["st"] = gln_fwd(yd, (Co * T * Fy, T * Fy, Fy), _p(xl), (Co * T * Fy, T * Fy, Fy), blk.norm.weight, blk.norm.bias, S, Co, T, Fy, Fy, 0, act)
                rec["xl"] = xl
                if Co != 2 or Fy != F0:
                    raise RuntimeError("last decoder block must produce the 2-channel mask at full resolution")
            dec.append(rec)
        xl = dec[-1]["xl"]
        Y = _new(S, T, F0, 2, dev=dev)
        _run("k_tmask", 0.0, lib.se_train_mask_fwd, _p(xl), _p(spec), _p(Y), S, M, T, F0, st())
        yseg = _new(S, Ks, dev=dev)
        _run("k_istft", 0.0, lib.se_sig_istft, sig, _p(Y), S, _p(yseg), st())
        Lout = Lp - skip  # == L (flag=False: Lp = L + P, strip P) or L (flag=True)
        pred = _new(B, Lout, dev=dev)
        _run("k_tola", 0.0, lib.se_train_ola_fwd, sig, _p(yseg), _p(pred), B, Lout, skip, st())
        # carried state for a flag=True continuation: the last segment's block inputs and the GRU state (detached by construction)
        model._state = dict(buf=[xin[i][N] for i in range(Lv)], h=hTs, pbuf=[r["cur"][N] for r in pre] if V else None)
        ctx.model = model
        ctx.dims = dict(B=B, M=M, L=Lout, N=N, S=S, T=T, F0=F0, Ks=Ks, skip=skip, ch=ch, Fq=Fq, Lv=Lv, H=H, NL=NL, D=D, CL=CL, FL=FL, n_fft=n_fft, sig=sig)
        ctx.sv = dict(V=V, act=act, pre=pre, enc_tg=enc_tg, spec=spec, xin=xin, ys=ys, stats_e=stats_e, seq=seq, outs=outs, gates=gates, h0s=h0s, o_fc=o_fc, st_fc=st_fc, dec=dec, xl=xl)
        return pred

    @staticmethod
    def backward(ctx, dpred):
        lib = K._lib()
        model, q, sv = ctx.model, ctx.dims, ctx.sv
        B, M, L, N, S, T, F0, Ks, skip = q["B"], q["M"], q["L"], q["N"], q["S"], q["T"], q["F0"], q["Ks"], q["skip"]
        ch, Fq, Lv, H, NL, D, CL, FL, sig = q["ch"], q["Fq"], q["Lv"], q["H"], q["NL"], q["D"], q["CL"], q["FL"], q["sig"]
        dev = dpred.device
        st = K._st
        dpred = dpred.contiguous()
        grads = {}
        V, act = sv["V"], sv["act"]
        zero_bias = torch.zeros(256, device=dev)

        gseg = _new(S, Ks, dev=dev)
        _run("k_tola", 0.0, lib.se_train_ola_bwd, sig, _p(dpred), _p(gseg), B, N, L, skip, st())
        dY = _new(S, T, F0, 2, dev=dev)
        _run("k_stft", 0.0, lib.se_sig_stft, sig, _p(gseg), S, 1, Ks, 0, 0, 1, _p(dY), st())
        dx = _new(S, 2, T, F0, dev=dev)
        _run("k_tmask", 0.0, lib.se_train_mask_bwd, _p(dY), _p(sv["xl"]), _p(sv["spec"]), _p(dx), S, M, T, F0, q["n_fft"], st())
        dres = {}
        dout = dx
        for j in range(Lv - 1, -1, -1):
            blk, rec = model.deconvlist[j], sv["dec"][j]
            Ci, Co, Fi, Fy, d = rec["Ci"], rec["Co"], rec["Fi"], rec["Fy"], rec["d"]
            pre = f"deconvlist.{j}."
            if j < Lv - 1:
                Cr, Fr, k = rec["Cr"], rec["Fr"], rec["k"]
                duv = _new(S, 2 * Co, T, Fr, dev=dev)
                dz = _new(S, Co, T, Fr, dev=dev)
                pw, pb, pbias = _new(S, Co, dev=dev), _new(S, Co, dev=dev), _new(S, 2 * Co, dev=dev)
                _run("k_tskip_bwd", 0.0, lib.se_train_skip_bwd, _p(dout), _p(rec["uv"]), _p(rec["z"]), _p(blk.residualnorm.weight), _p(blk.residualnorm.bias),
                     _p(rec["st_uv"]), _p(duv), _p(dz), _p(pw), _p(pb), _p(pbias), S, Co, T, Fr, act, 0, st())
                dnw, dnb, dbuv = colsum3(S, (pw, Co), (pb, Co), (pbias, 2 * Co))
                grads[pre + "residualnorm.weight"], grads[pre + "residualnorm.bias"] = dnw, dnb
                grads[pre + "residual.bias"], grads[pre + "residualmask.bias"] = dbuv[:Co], dbuv[Co:]
                res_off = B * Cr * T * Fr
                dwuv = wgrad(duv, _p(sv["xin"][k], res_off), None, S, 2 * Co, Cr, T, Fr, Fr, 0, 1).view(2 * Co, Cr)
                grads[pre + "residual.weight"], grads[pre + "residualmask.weight"] = dwuv[:Co], dwuv[Co:]
                dr = _new(S, Cr, T, Fr, dev=dev)
                conv_w(3, _p(duv), None, rec["wuv"], 1, Cr, zero_bias, dr, S, 2 * Co, Cr, T, Fr, Fr, 0)
                dres[k] = dr
                dy_ptr, ds = _p(dz), (Co * T * Fr, T * Fr, Fr)
            else:
                dy_ptr, ds = _p(dout), (Co * T * Fy, T * Fy, Fy)
            dyd, dw, db, dpre = gln_bwd(dy_ptr, ds, rec["yd"], (Co * T * Fy, T * Fy, Fy), blk.norm.weight, rec["st"], S, Co, T, Fy, 0, act)
            grads[pre + "norm.weight"], grads[pre + "norm.bias"], grads[pre + "conv.bias"] = dw, db, dpre
            grads[pre + "conv.weight"] = wgrad(rec["x_in"], dyd, None, S, Ci, Co, T, Fi, Fy, d, 15)
            din = _new(S, Ci, T, Fi, dev=dev)
            conv_w(0, _p(dyd), None, blk.conv.weight, Co * 15, 15, zero_bias, din, S, Co, Ci, T, Fy, Fi, d)
            dout = din
        # bottleneck: gLN(last) + ReLU + fc, then the GRU layers in reverse
        R = S * T
        fc = model.gru.fc_output_layer
        do_fc, dw, db, dpre = gln_bwd(_p(dout), (CL * T * FL, T * FL, FL), sv["o_fc"], (T * D, FL, D), model.gru.norm.weight, sv["st_fc"], S, CL, T, FL, 1, act)
        grads["gru.norm.weight"], grads["gru.norm.bias"], grads["gru.fc_output_layer.bias"] = dw, db, dpre
        top = sv["outs"][NL - 1]
        grads["gru.fc_output_layer.weight"] = gemm_tn(do_fc, top)
        dlayer = K._gemm(do_fc, transpose(fc.weight))  # [R, H]
        g = model.gru.sequence_model
        for l in range(NL - 1, -1, -1):
            w_ih, w_hh = getattr(g, f"weight_ih_l{l}"), getattr(g, f"weight_hh_l{l}")
            out, gt, h0 = sv["outs"][l], sv["gates"][l], sv["h0s"][l]
            dgi, dgh = _new(R, 3 * H, dev=dev), _new(R, 3 * H, dev=dev)
            hp = _new(R, H, dev=dev)
            _run("k_gru_hprev", 0.0, lib.se_train_gru_hprev, _p(out), _p(h0), _p(hp), B, N * T, H, T, B * T, T, st())
            # The carried state is detached at every segment seam (CRN.py:281), so NOTHING flows back across a seam: the N segments of an
            # utterance are independent in the backward sweep.  In the segment-major layout the rows are already [S = N*B][T]: the
            # BPTT runs as S streams of T steps (groups of <= 32 streams per persistent launch) instead of B streams of N*T steps -
            # 21 dependent steps instead of 714.  Each stream's entering state is row 0 of its h_{s-1} block.
            h0s = hp.view(S, T, H)[:, 0].contiguous()
            K._gru_seq_bwd(dlayer, None, gt, out, h0s, transpose(w_hh), dgi, dgh, S, T, H, T, 0, T, 0)
            x_l = sv["seq"] if l == 0 else sv["outs"][l - 1]
            grads[f"gru.sequence_model.weight_ih_l{l}"] = gemm_tn(dgi, x_l)
            grads[f"gru.sequence_model.weight_hh_l{l}"] = gemm_tn(dgh, hp)
            grads[f"gru.sequence_model.bias_ih_l{l}"] = colsum_tall(dgi)
            grads[f"gru.sequence_model.bias_hh_l{l}"] = colsum_tall(dgh)
            dlayer = K._gemm(dgi, transpose(w_ih))  # [R, In]
        def gate_pair_bwd(dy_ptr, ds, a_t, tg, stt, blk, Co, Fo, pre):
            """through gLN + gated pair + ELU: returns dy of the convolution that produced a_t (in place in a fresh tensor)."""
            dtg = _new(S, 2 * Co, T, Fo, dev=dev)
            pw, pb, pbias = _new(S, Co, dev=dev), _new(S, Co, dev=dev), _new(S, 2 * Co, dev=dev)
            _run("k_tgate_bwd", 0.0, lib.se_train_gate_bwd, dy_ptr, *ds, _p(tg), _p(blk.norm.weight), _p(stt), _p(dtg), _p(pw), _p(pb), _p(pbias), S, Co, T, Fo, 0, st())
            dnw, dnb, dbtg = colsum3(S, (pw, Co), (pb, Co), (pbias, 2 * Co))
            grads[pre + "norm.weight"], grads[pre + "norm.bias"] = dnw, dnb
            grads[pre + "conv_trans.bias"], grads[pre + "conv_gated.bias"] = dbtg[:Co], dbtg[Co:]
            dwtg = wgrad(dtg, a_t, None, S, 2 * Co, Co, T, Fo, Fo, 0, 1).view(2 * Co, Co)
            grads[pre + "conv_trans.weight"], grads[pre + "conv_gated.weight"] = dwtg[:Co], dwtg[Co:]
            wst = _new(2 * Co, Co, dev=dev)   # [conv_trans; conv_gated] stacked: d a = W^T dtg in one 2Co-deep contraction
            wst[:Co].copy_(blk.conv_trans.weight.view(Co, Co)); wst[Co:].copy_(blk.conv_gated.weight.view(Co, Co))
            da = _new(S, Co, T, Fo, dev=dev)
            conv_w(3, _p(dtg), None, wst, 1, Co, zero_bias, da, S, 2 * Co, Co, T, Fo, Fo, 0)
            pp = _new(S, Co, dev=dev)
            _run("k_telu_bwd", 0.0, lib.se_train_elu_bwd, _p(da), _p(a_t), _p(pp), S, Co, T, Fo, st())
            grads[pre + "conv.bias"] = colsum3(S, (pp, Co))[0]
            return da

        # encoder, last block first; dlayer = d seq [S][T][D]
        dy_ptr, ds = _p(dlayer), (T * D, FL, D)
        dx0 = None
        for i in range(Lv - 1, -1, -1):
            blk = model.convlist[i]
            Ci, Co, Fi, Fo, d = ch[i], ch[i + 1], Fq[i], Fq[i + 1], 2 ** i
            pre = f"convlist.{i}."
            if V:
                dy = gate_pair_bwd(dy_ptr, ds, sv["ys"][i], sv["enc_tg"][i], sv["stats_e"][i], blk, Co, Fo, pre)
            else:
                dy, dw, db, dpre = gln_bwd(dy_ptr, ds, sv["ys"][i], (Co * T * Fo, T * Fo, Fo), blk.norm.weight, sv["stats_e"][i], S, Co, T, Fo, 0, 1)
                grads[pre + "norm.weight"], grads[pre + "norm.bias"], grads[pre + "conv.bias"] = dw, db, dpre
            slab = B * Ci * T * Fi
            grads[pre + "conv.weight"] = wgrad(dy, _p(sv["xin"][i], slab), _p(sv["xin"][i]), S, Co, Ci, T, Fo, Fi, d, 15)
            if i == 0 and not V:
                break  # the features carry no gradient
            dxi = _new(S, Ci, T, Fi, dev=dev)
            for kind in (1, 2):
                conv_w(kind, _p(dy), None, blk.conv.weight, 15, Ci * 15, zero_bias, dxi, S, Co, Ci, T, Fo, Fi, d)
            if i in dres:
                _run("k_tadd", 0.0, lib.se_train_add, _p(dxi), _p(dres[i]), dxi.numel(), st())
            dy_ptr, ds = _p(dxi), (Ci * T * Fi, T * Fi, Fi)
            dx0 = dxi
        if V:  # the pre-conv chain, last block first: x_{k+1} = block_k(x_k) + x_k
            C0, slab0 = ch[0], B * ch[0] * T * F0
            dnxt = dx0
            for k in range(2, -1, -1):
                blk, r = model.preconvlist[k], sv["pre"][k]
                pre = f"preconvlist.{k}."
                dyk = gate_pair_bwd(_p(dnxt), (C0 * T * F0, T * F0, F0), r["a"], r["tg"], r["st"], blk, C0, F0, pre)
                part = _new(S, C0 * C0 * 25, dev=dev)
                _run("k_pre5", 2.0 * S * C0 * C0 * 25 * T * F0, lib.se_train_pre5, 2, _p(r["cur"], slab0), _p(r["cur"]), _p(blk.conv.weight), None, _p(dyk), _p(part),
                     S, C0, T, F0, 2 ** k, 0, st())
                grads[pre + "conv.weight"] = colsum3(S, (part, C0 * C0 * 25))[0]
                if k == 0:
                    break  # block 0 reads the features
                dcur = _new(S, C0, T, F0, dev=dev)
                _run("k_pre5", 2.0 * S * C0 * C0 * 25 * T * F0, lib.se_train_pre5, 1, None, None, _p(blk.conv.weight), None, _p(dyk), _p(dcur), S, C0, T, F0, 2 ** k, 0, st())
                _run("k_tadd", 0.0, lib.se_train_add, _p(dcur), _p(dnxt), dcur.numel(), st())
                dnxt = dcur
        out = []
        for (name, p) in model.named_parameters():
            gname = name.replace(".net.0.", ".conv.")
            gr = grads.get(gname)
            out.append(None if gr is None else gr.reshape(p.shape))
        ctx.sv = None
        return (None, None, None, *out)


def realtime_process_fused(model, mixture, flag=False):
    params = [p for _, p in model.named_parameters()]
    return CRNFunction.apply(model, mixture, bool(flag), *params)

"""ctypes binding of the C ABI in include/se_engine.h (libse_engine.so, HIP / gfx950).

This is the reference-side stub a maintainer would add: it carries no arithmetic, only
pointer plumbing.  PyTorch is used for device memory and streams; the signatures themselves are plain C.
There is no CPU fallback: if the library or a GPU is missing every call raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libse_engine.so")
SE_MAX_LEVELS = 8

EXPORTS = [
    "se_abi_version", "se_config_size", "fsn_config_size", "se_create", "se_destroy", "se_last_error", "se_load_param", "se_reset", "se_reset_stream", "se_step",
    "se_realtime_process", "se_realtime_process_ragged", "se_stft", "se_istft", "se_forward", "se_read_tap", "se_read_tap_dev", "se_export_state",
    "se_import_state", "se_flops_per_frame", "se_frames_per_segment", "se_profile", "se_profile_read",
    "fsn_create", "fsn_destroy", "fsn_last_error", "fsn_load_param", "fsn_reset", "fsn_forward", "fsn_realtime_process",
    "fsn_read_tap", "fsn_flops_per_frame", "se_loss_sisnr_fwd", "se_loss_sisnr_bwd", "se_loss_stoi_ws_floats", "se_loss_stoi_fwd", "se_loss_stoi_bwd", "se_loss_stoi_last_error",
    "se_train_last_error", "se_train_conv_layout_query", "se_train_conv", "se_train_conv_wgrad", "se_train_gemm", "se_train_gemm_tn", "se_train_gru_step",
    "se_train_gru_bwd_gates", "se_train_gru_seq_fwd", "se_train_gru_seq_bwd", "se_train_gru_pseq_supported", "se_train_gru_pseq_scratch_floats", "se_train_gru_pseq_fwd", "se_train_gru_pseq_bwd",
    "se_sig_create", "se_sig_destroy", "se_sig_stft", "se_sig_istft", "se_train_ola_fwd", "se_train_ola_bwd", "se_train_feat", "se_train_mask_fwd",
    "se_train_mask_bwd", "se_train_gln_fwd", "se_train_gln_bwd", "se_train_colsum", "se_train_colsum_tall", "se_train_skip_fwd", "se_train_skip_bwd",
    "se_train_add", "se_train_add3", "se_train_gate_fwd", "se_train_gate_bwd", "se_train_elu_bwd", "se_train_pre5", "se_train_gru_hprev", "se_train_conv_ws_floats", "se_train_conv_w", "se_train_conv_wgrad_det", "se_train_gemm_tn_det", "se_synth_last_error", "se_synth_rir", "se_synth_rir_tail", "se_synth_fir", "se_synth_mix",
]


class SeConfig(C.Structure):
    _fields_ = [("num_levels", C.c_int32), ("channels", C.c_int32 * SE_MAX_LEVELS), ("num_freqs", C.c_int32),
                ("hidden", C.c_int32), ("num_layers", C.c_int32), ("num_inputs", C.c_int32),
                ("kernel_size", C.c_int32), ("n_fft", C.c_int32), ("win", C.c_int32), ("hop", C.c_int32),
                ("segment_length", C.c_int32), ("variant", C.c_int32), ("precision", C.c_int32)]


class TrainConvLayout(C.Structure):  # se_train_conv_layout
    _fields_ = [("ntap", C.c_int32), ("CC", C.c_int32), ("nchunk", C.c_int32), ("CoPad", C.c_int32), ("FP", C.c_int32),
                ("tap_kf", C.c_int32 * 15), ("tap_kt", C.c_int32 * 15)]


class FsnConfig(C.Structure):
    _fields_ = [("num_freqs", C.c_int32), ("num_mics", C.c_int32), ("fb_hidden", C.c_int32), ("sb_hidden", C.c_int32),
                ("num_layers", C.c_int32), ("sb_neighbors", C.c_int32), ("fb_neighbors", C.c_int32), ("look_ahead", C.c_int32),
                ("n_fft", C.c_int32), ("win", C.c_int32), ("hop", C.c_int32), ("segment_length", C.c_int32),
                ("precision", C.c_int32)]


_lib = None


def load_library():
    """Load libse_engine.so; raises (never falls back) when the HIP extension is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C speech_enhancement_mi_amd/csrc).  The engine has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, fp, i64p = C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)
    L.se_abi_version.restype = C.c_int
    # a struct mirror shorter than the library's se_config would leave its tail fields (precision) reading garbage
    if L.se_config_size() != C.sizeof(SeConfig) or L.fsn_config_size() != C.sizeof(FsnConfig):
        raise RuntimeError(f"libse_engine.so was built with sizeof(se_config) = {L.se_config_size()}, sizeof(fsn_config) = "
                           f"{L.fsn_config_size()}; this binding has {C.sizeof(SeConfig)} / {C.sizeof(FsnConfig)}: rebuild the library")
    L.se_create.argtypes = [C.POINTER(SeConfig), C.c_int, C.POINTER(vp)]
    L.se_destroy.argtypes = [vp]
    L.se_destroy.restype = None
    L.se_last_error.argtypes = [vp]
    L.se_last_error.restype = C.c_char_p
    L.se_load_param.argtypes = [vp, C.c_char_p, fp, i64p, C.c_int]
    L.se_reset.argtypes = [vp, C.c_int]
    L.se_reset_stream.argtypes = [vp, C.c_int, vp]
    L.se_step.argtypes = [vp, fp, fp, vp]
    L.se_realtime_process.argtypes = [vp, fp, C.c_int, C.c_int64, C.c_int, fp, vp]
    L.se_realtime_process_ragged.argtypes = [vp, fp, C.c_int, C.c_int64, i64p, C.c_int, fp, vp]
    L.se_stft.argtypes = [vp, fp, C.c_int, fp, vp]
    L.se_istft.argtypes = [vp, fp, C.c_int, fp, vp]
    L.se_forward.argtypes = [vp, fp, fp, vp]
    L.se_read_tap.argtypes = [vp, C.c_char_p, fp, C.c_int64, i64p, vp]
    L.se_read_tap_dev.argtypes = [vp, C.c_char_p, vp, C.c_int64, i64p, vp]
    L.se_export_state.argtypes = [vp, C.c_char_p, fp, C.c_int64, i64p, vp]
    L.se_import_state.argtypes = [vp, C.c_char_p, fp, C.c_int64, vp]
    L.se_flops_per_frame.argtypes = [vp]
    L.se_flops_per_frame.restype = C.c_double
    L.se_frames_per_segment.argtypes = [vp]
    L.fsn_create.argtypes = [C.POINTER(FsnConfig), C.c_int, C.POINTER(vp)]
    L.fsn_destroy.argtypes = [vp]
    L.fsn_destroy.restype = None
    L.fsn_last_error.argtypes = [vp]
    L.fsn_last_error.restype = C.c_char_p
    L.fsn_load_param.argtypes = [vp, C.c_char_p, fp, i64p, C.c_int]
    L.fsn_reset.argtypes = [vp, C.c_int]
    L.fsn_forward.argtypes = [vp, fp, fp, vp]
    L.fsn_realtime_process.argtypes = [vp, fp, C.c_int, C.c_int64, C.c_int, fp, vp]
    L.fsn_read_tap.argtypes = [vp, C.c_char_p, fp, C.c_int64, i64p, vp]
    L.fsn_flops_per_frame.argtypes = [vp]
    L.fsn_flops_per_frame.restype = C.c_double
    L.se_loss_sisnr_fwd.argtypes = [vp, vp, vp, C.c_int, C.c_int64, vp, vp, vp]
    L.se_loss_stoi_ws_floats.argtypes = [C.c_int, C.c_int64]
    L.se_loss_stoi_ws_floats.restype = C.c_int64
    L.se_loss_stoi_fwd.argtypes = [vp, vp, vp, C.c_int, C.c_int64, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp]
    L.se_loss_stoi_bwd.argtypes = [vp, vp, C.c_int, C.c_int64, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp]
    L.se_loss_stoi_last_error.restype = C.c_char_p
    L.se_loss_sisnr_bwd.argtypes = [vp, vp, vp, C.c_int, C.c_int64, vp, vp, vp, vp]
    i32, i64 = C.c_int, C.c_int64
    L.se_train_last_error.restype = C.c_char_p
    L.se_train_conv_layout_query.argtypes = [i32] * 7 + [C.POINTER(TrainConvLayout)]
    L.se_train_conv.argtypes = [i32, vp, vp, vp, vp, vp] + [i32] * 8 + [vp]
    L.se_train_conv_wgrad.argtypes = [vp, vp, vp, vp] + [i32] * 7 + [vp]
    L.se_train_gemm.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.se_train_gemm_tn.argtypes = [vp, vp, vp, i64, i32, i32, vp]
    L.se_train_gru_step.argtypes = [vp, i64, vp, vp, vp, vp, vp, i64, vp, i64, i32, i32, vp]
    L.se_train_gru_bwd_gates.argtypes = [vp, i64, vp, vp, vp, i64, vp, i64, vp, vp, i64, vp, i32, i32, vp]
    L.se_train_gru_seq_fwd.argtypes = [vp] * 8 + [i32, i32, i32, vp]
    L.se_train_gru_seq_bwd.argtypes = [vp] * 9 + [i32, i32, i32, i32, vp]
    L.se_train_gru_pseq_supported.argtypes = [i32, i32]
    L.se_train_gru_pseq_scratch_floats.argtypes = [i32, i32]
    L.se_train_gru_pseq_fwd.argtypes = [vp] * 8 + [i32, i32, i32, i32, i64, i64, vp]
    L.se_train_gru_pseq_bwd.argtypes = [vp] * 9 + [i32, i32, i32, i32, i64, i64, i32, vp]
    L.se_sig_create.argtypes = [i32, i32, i32, i32, i32, C.POINTER(vp)]
    L.se_sig_destroy.argtypes = [vp]
    L.se_sig_destroy.restype = None
    L.se_sig_stft.argtypes = [vp, vp, i32, i32, i64, i64, i64, i32, vp, vp]
    L.se_sig_istft.argtypes = [vp, vp, i32, vp, vp]
    L.se_train_ola_fwd.argtypes = [vp, vp, vp, i32, i64, i64, vp]
    L.se_train_ola_bwd.argtypes = [vp, vp, vp, i32, i32, i64, i64, vp]
    L.se_train_feat.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp]
    L.se_train_mask_fwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp]
    L.se_train_mask_bwd.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    L.se_train_gln_fwd.argtypes = [vp, i64, i64, i64, vp, i64, i64, i64, vp, vp, vp] + [i32] * 8 + [vp]
    L.se_train_gln_bwd.argtypes = [vp, i64, i64, i64, vp, i64, i64, i64, vp, vp, vp, vp, vp, vp] + [i32] * 7 + [vp]
    L.se_train_colsum.argtypes = [vp, vp, i32, vp, vp, i32, vp, vp, i32, i32, i32, vp]
    L.se_train_colsum_tall.argtypes = [vp, i64, i32, vp, vp, i32, vp]
    L.se_train_skip_fwd.argtypes = [vp] * 6 + [i32] * 6 + [vp]
    L.se_train_skip_bwd.argtypes = [vp] * 11 + [i32] * 6 + [vp]
    L.se_train_add.argtypes = [vp, vp, i64, vp]
    L.se_train_add3.argtypes = [vp, vp, vp, i64, vp]
    L.se_train_gate_fwd.argtypes = [vp, vp, vp, vp, i64, i64, i64, vp] + [i32] * 5 + [vp]
    L.se_train_gate_bwd.argtypes = [vp, i64, i64, i64] + [vp] * 7 + [i32] * 5 + [vp]
    L.se_train_elu_bwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp]
    L.se_train_pre5.argtypes = [i32] + [vp] * 6 + [i32] * 6 + [vp]
    L.se_train_gru_hprev.argtypes = [vp, vp, vp, i32, i32, i32, i32, i64, i64, vp]
    L.se_train_conv_ws_floats.argtypes = [i32] * 7
    L.se_train_conv_w.argtypes = [i32, vp, vp, vp, i64, i64, vp, vp, vp] + [i32] * 10 + [vp]
    L.se_train_conv_wgrad_det.argtypes = [vp, vp, vp, vp, C.POINTER(C.c_int)] + [i32] * 8 + [vp]
    L.se_train_gemm_tn_det.argtypes = [vp, vp, vp, C.POINTER(C.c_int), i64, i32, i32, vp]
    L.se_synth_last_error.restype = C.c_char_p
    L.se_synth_rir.argtypes = [vp, vp, vp, vp] + [i32] * 6 + [C.c_float, C.c_float, i32, vp, vp]
    L.se_synth_rir_tail.argtypes = [vp, vp, vp, i32, i32, i32, i32, C.c_float, C.c_uint32, vp]
    L.se_synth_fir.argtypes = [vp, vp, i32, i32, i32, i64, i32, vp, vp]
    L.se_synth_mix.argtypes = [vp, vp, i32, i32, i32, i64, C.c_float, vp, vp, vp, vp]
    L.se_profile.argtypes = [vp, C.c_int]
    L.se_profile_read.argtypes = [vp, C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_double), i64p, C.POINTER(C.c_double)]
    _lib = L
    return L


def make_config(num_channels, num_freqs, hidden, segment_length, num_layers=1, num_inputs=3, kernel_size=3,
                sample_rate=16000, win_length=25, hop_length=10, n_fft=400, variant=0, precision=0) -> SeConfig:
    cfg = SeConfig()
    if len(num_channels) > SE_MAX_LEVELS:
        raise ValueError("too many levels")
    cfg.num_levels = len(num_channels)
    for i, c in enumerate(num_channels):
        cfg.channels[i] = int(c)
    cfg.num_freqs, cfg.hidden, cfg.num_layers = int(num_freqs), int(hidden), int(num_layers)
    cfg.num_inputs, cfg.kernel_size, cfg.n_fft = int(num_inputs), int(kernel_size), int(n_fft)
    cfg.win = int(round(sample_rate / 1000.0 * win_length))   # speechbrain STFT convention (CRN.py:421-425)
    cfg.hop = int(round(sample_rate / 1000.0 * hop_length))
    cfg.segment_length = int(segment_length)
    cfg.variant = int(variant)
    cfg.precision = int(precision)  # 0 = fp32-accurate (bf16x6), 1 = fp16 MFMA operands (model.half()), 2 = bf16x3 (set_precision)
    return cfg


class Engine:
    """Thin RAII wrapper over one se_engine handle.  All tensor arguments are CUDA(HIP) torch tensors."""

    def __init__(self, cfg: SeConfig, device: int = 0):
        self.lib = load_library()
        self.cfg = cfg
        self.device = int(device)
        h = C.c_void_p()
        rc = self.lib.se_create(C.byref(cfg), self.device, C.byref(h))
        if rc != 0:
            raise RuntimeError(f"se_create failed ({rc}): {self.lib.se_last_error(None).decode()}")
        self._h = h
        self.T = self.lib.se_frames_per_segment(h)
        self.F, self.M, self.K = cfg.num_freqs, cfg.num_inputs, cfg.segment_length
        self.batch = 0

    def close(self):
        if getattr(self, "_h", None):
            self.lib.se_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(f"se_engine error {rc}: {self.lib.se_last_error(self._h).decode()}")

    @staticmethod
    def _stream():
        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    @staticmethod
    def _dev(t, shape=None):
        import torch
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError("the engine takes contiguous float32 tensors on the GPU (no CPU fallback)")
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise RuntimeError(f"expected shape {tuple(shape)}, got {tuple(t.shape)}")
        return C.c_void_p(t.data_ptr())

    def load_state_dict(self, sd: Dict[str, "np.ndarray"]):
        for k, v in sd.items():
            a = np.ascontiguousarray(v.detach().cpu().numpy() if hasattr(v, "detach") else v, dtype=np.float32)
            shp = (C.c_int64 * max(1, a.ndim))(*a.shape)
            self._check(self.lib.se_load_param(self._h, k.encode(), C.c_void_p(a.ctypes.data), shp, a.ndim))

    def reset(self, batch: int):
        self._check(self.lib.se_reset(self._h, int(batch)))
        self.batch = int(batch)

    def reset_stream(self, index: int):
        """Zero the state of ONE stream of the batch (a new caller takes the slot); the others keep streaming."""
        self._check(self.lib.se_reset_stream(self._h, int(index), self._stream()))

    def step(self, wav_in, wav_out=None):
        import torch
        B = self.batch
        if wav_out is None:
            wav_out = torch.empty((B, self.K), dtype=torch.float32, device=wav_in.device)
        self._check(self.lib.se_step(self._h, self._dev(wav_in, (B, self.M, self.K)), self._dev(wav_out, (B, self.K)), self._stream()))
        return wav_out

    def realtime_process(self, mixture, flag=False, out=None, lengths=None):
        """lengths (optional, [B] ints <= L): ragged batch - every stream is processed as if alone with its own length."""
        import torch
        B, M, L = mixture.shape
        if M != self.M:
            raise RuntimeError(f"expected {self.M} microphones, got {M}")
        if out is None:
            out = torch.empty((B, L), dtype=torch.float32, device=mixture.device)
        if lengths is not None:
            ln = [int(v) for v in (lengths.tolist() if hasattr(lengths, "tolist") else lengths)]
            if len(ln) != B:
                raise RuntimeError(f"{len(ln)} lengths for a batch of {B}")
            # Prefix compaction: with non-increasing lengths the engine launches every segment for the streams still running
            # only.  A fresh batch (flag=False) is therefore sorted by length here and un-sorted on the way out; a continuation keeps its
            # slots (the carried state lives in them) and is compacted only if it happens to be sorted.
            order = sorted(range(B), key=lambda i: -ln[i])
            permute = not flag and order != list(range(B))
            if permute:
                idx = torch.tensor(order, dtype=torch.int64, device=mixture.device)
                src, dst = mixture.index_select(0, idx).contiguous(), torch.empty_like(out)
                ln_call = [ln[i] for i in order]
            else:
                src, dst, ln_call = mixture, out, ln
            arr = (C.c_int64 * B)(*ln_call)
            self._check(self.lib.se_realtime_process_ragged(self._h, self._dev(src), B, L, arr, int(bool(flag)), self._dev(dst, (B, L)), self._stream()))
            if permute:
                out.index_copy_(0, idx, dst)
            self.batch = B
            return out
        self._check(self.lib.se_realtime_process(self._h, self._dev(mixture), B, L, int(bool(flag)), self._dev(out, (B, L)), self._stream()))
        self.batch = B
        return out

    def stft(self, seg):
        import torch
        n = seg.shape[0]
        spec = torch.empty((n, self.F, self.T, 2), dtype=torch.float32, device=seg.device)
        self._check(self.lib.se_stft(self._h, self._dev(seg, (n, self.K)), n, self._dev(spec), self._stream()))
        return spec

    def istft(self, spec):
        import torch
        n = spec.shape[0]
        wav = torch.empty((n, self.K), dtype=torch.float32, device=spec.device)
        self._check(self.lib.se_istft(self._h, self._dev(spec, (n, self.F, self.T, 2)), n, self._dev(wav), self._stream()))
        return wav

    def forward(self, x):
        import torch
        B = self.batch
        y = torch.empty((B, self.F, self.T, 2), dtype=torch.float32, device=x.device)
        self._check(self.lib.se_forward(self._h, self._dev(x, (B, self.M, self.F, self.T, 2)), self._dev(y), self._stream()))
        return y

    def _host_read(self, fn, name: str) -> np.ndarray:
        n = C.c_int64(0)
        probe = np.empty(1, np.float32)
        fn(self._h, name.encode(), C.c_void_p(probe.ctypes.data), 0, C.byref(n), self._stream())
        if n.value <= 0:
            self._check(-1)
        out = np.empty(n.value, np.float32)
        self._check(fn(self._h, name.encode(), C.c_void_p(out.ctypes.data), n.value, C.byref(n), self._stream()))
        return out

    def read_tap(self, name: str) -> np.ndarray:
        return self._host_read(self.lib.se_read_tap, name)

    def read_tap_dev(self, name: str, numel: int):
        """A distillation feature tap ("ft0".."ft<L>") as a flat fp32 DEVICE tensor of `numel` elements ([B, C, F, T] memory): no host copy."""
        import torch
        out = torch.empty(int(numel), dtype=torch.float32, device=f"cuda:{self.device}")
        n = C.c_int64(0)
        self._check(self.lib.se_read_tap_dev(self._h, name.encode(), C.c_void_p(out.data_ptr()), int(numel), C.byref(n), self._stream()))
        if n.value != numel:
            raise RuntimeError(f"tap {name}: engine wrote {n.value} elements, caller expected {numel}")
        return out

    def export_state(self, name: str) -> np.ndarray:
        return self._host_read(self.lib.se_export_state, name)

    def import_state(self, name: str, arr: np.ndarray):
        a = np.ascontiguousarray(arr, dtype=np.float32)
        self._check(self.lib.se_import_state(self._h, name.encode(), C.c_void_p(a.ctypes.data), a.size, self._stream()))

    def profile(self, enable: bool):
        self._check(self.lib.se_profile(self._h, int(bool(enable))))

    def profile_read(self):
        """[{kernel, label, ms, launches, flops_per_launch}] accumulated since profile(True)."""
        out = []
        i = 0
        while True:
            k, l = C.create_string_buffer(64), C.create_string_buffer(64)
            ms, n, fl = C.c_double(0), C.c_int64(0), C.c_double(0)
            rc = self.lib.se_profile_read(self._h, i, k, l, 64, C.byref(ms), C.byref(n), C.byref(fl))
            if rc == 1:
                break
            self._check(rc)
            out.append(dict(kernel=k.value.decode(), label=l.value.decode(), ms=ms.value, launches=n.value, flops_per_launch=fl.value))
            i += 1
        return out

    @property
    def flops_per_frame(self) -> float:
        return float(self.lib.se_flops_per_frame(self._h))


class FsnEngine:
    """RAII wrapper over one fsn_engine handle (FullSubNet, fullsubnet.py:685-961)."""

    def __init__(self, num_freqs, num_mics, fb_hidden, sb_hidden, num_layers=2, sb_neighbors=15, fb_neighbors=0, look_ahead=0,
                 sample_rate=16000, segment_length=3200, win_length=25, hop_length=10, n_fft=400, device=0, precision=0):
        self.lib = load_library()
        cfg = FsnConfig(int(num_freqs), int(num_mics), int(fb_hidden), int(sb_hidden), int(num_layers), int(sb_neighbors),
                        int(fb_neighbors), int(look_ahead), int(n_fft), int(round(sample_rate / 1000.0 * win_length)),
                        int(round(sample_rate / 1000.0 * hop_length)), int(segment_length), int(precision))
        self.cfg = cfg
        h = C.c_void_p()
        rc = self.lib.fsn_create(C.byref(cfg), int(device), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"fsn_create failed ({rc}): {self.lib.fsn_last_error(None).decode()}")
        self._h = h
        self.F, self.M, self.K, self.T = cfg.num_freqs, cfg.num_mics, cfg.segment_length, 1 + cfg.segment_length // cfg.hop
        self.batch = 0

    def close(self):
        if getattr(self, "_h", None):
            self.lib.fsn_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(f"fsn_engine error {rc}: {self.lib.fsn_last_error(self._h).decode()}")

    def load_state_dict(self, sd):
        for k, v in sd.items():
            a = np.ascontiguousarray(v.detach().cpu().numpy() if hasattr(v, "detach") else v, dtype=np.float32)
            shp = (C.c_int64 * max(1, a.ndim))(*a.shape)
            self._check(self.lib.fsn_load_param(self._h, k.encode(), C.c_void_p(a.ctypes.data), shp, a.ndim))

    def reset(self, batch):
        self._check(self.lib.fsn_reset(self._h, int(batch)))
        self.batch = int(batch)

    def forward(self, x):
        import torch
        B = self.batch
        crm = torch.empty((B, 2, self.F, self.T), dtype=torch.float32, device=x.device)
        self._check(self.lib.fsn_forward(self._h, Engine._dev(x, (B, 2 * self.M, self.F, self.T)), Engine._dev(crm), Engine._stream()))
        return crm

    def realtime_process(self, mixture, flag=False, out=None):
        import torch
        B, M, L = mixture.shape
        if M != self.M:
            raise RuntimeError(f"expected {self.M} microphones, got {M}")
        if out is None:
            out = torch.empty((B, L), dtype=torch.float32, device=mixture.device)
        self._check(self.lib.fsn_realtime_process(self._h, Engine._dev(mixture), B, L, int(bool(flag)), Engine._dev(out, (B, L)), Engine._stream()))
        self.batch = B
        return out

    def read_tap(self, name):
        n = C.c_int64(0)
        probe = np.empty(1, np.float32)
        self.lib.fsn_read_tap(self._h, name.encode(), C.c_void_p(probe.ctypes.data), 0, C.byref(n), Engine._stream())
        if n.value <= 0:
            self._check(-1)
        out = np.empty(n.value, np.float32)
        self._check(self.lib.fsn_read_tap(self._h, name.encode(), C.c_void_p(out.ctypes.data), n.value, C.byref(n), Engine._stream()))
        return out

    @property
    def flops_per_frame(self):
        return float(self.lib.fsn_flops_per_frame(self._h))

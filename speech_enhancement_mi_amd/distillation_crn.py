"""Drop-in for the student architecture of the reference's distillation_crn.py (`TemporalCRN`, distillation_crn.py:
283-501; the distilled 0.81 M-parameter model is `TemporalCRN(num_channels=[16,32,64,64], hidden=128, ...)`,
distillation_crn.py:524-526).

Like the reference, `forward` returns `(y, [5 feature maps])` and `realtime_process` returns `(pred, [5 tensors of
[N*B, C_k, F_k, T]])` (distillation_crn.py:467-477): the pre-activation outputs of the last encoder convolution, of the
bottleneck's fc layer and of the first three transposed convolutions, which only the distillation *training* loss consumes.
The hot path fuses activations into the producing kernels, so the feature maps are produced by re-running those five kernels
without activation after every segment (engine taps "ft0".."ft4", written to device memory by `se_read_tap_dev`; they never cross the
host): set `return_features = False` for
inference (`predict_distillation.py:84` discards them) to get `(pred, None)` at full streaming speed."""
import torch

from .crn import TemporalCRN as _Base


class TemporalCRN(_Base):
    _VARIANT = 2
    return_features = True

    def _feature_shapes(self):
        c = self._cfg_args["num_channels"]
        L = len(c)
        F = [self.num_freqs]
        for _ in range(L):
            F.append((F[-1] - 1) // 2 + 1)
        return [(c[-1], F[L]), (c[-1], F[L])] + [(c[L - 2 - j], 2 * F[L - j] - 1) for j in range(L - 1)]

    def _features(self, eng, B):
        """The five taps of the segment just processed as DEVICE tensors [B, C*F, T] (se_read_tap_dev: the re-run kernels write
        device memory; nothing crosses the host)."""
        T = eng.T
        return [eng.read_tap_dev(f"ft{k}", B * ch * fr * T).view(B, ch * fr, T) for k, (ch, fr) in enumerate(self._feature_shapes())]

    def _shape_features(self, feats, B):
        return [f.reshape(f.shape[0], ch, fr, -1) for f, (ch, fr) in zip(feats, self._feature_shapes())]

    def forward(self, x):
        y = super().forward(x)
        if not self.return_features:
            return y, None
        feats = self._shape_features(self._features(self._eng, x.shape[0]), x.shape[0])
        return y, [f.to(x.device) for f in feats]

    def realtime_process(self, mixture, flag=False):
        if not self.return_features:
            return super().realtime_process(mixture, flag), None
        # segment by segment (utility.segmentation order, distillation_crn.py:455-471) so that the taps of every segment exist
        eng = self._engine_for(mixture)
        B, M, L = mixture.shape
        K = self.segment_length
        P = K // 2
        x = mixture.contiguous().float()
        if not flag:
            x = torch.nn.functional.pad(x, (P, 0))
            eng.reset(B)
        elif eng.batch != B:
            raise RuntimeError(f"flag=True with batch {B} but the carried state holds {eng.batch} streams")
        Lp = x.shape[-1]
        gap = K - (P + Lp % K) % K
        xp = torch.nn.functional.pad(x, (P, gap + P))
        N = 2 * (Lp + gap + P) // K
        segs, feats = [], None
        for n in range(N):
            segs.append(eng.step(xp[:, :, n * P:n * P + K].contiguous()))
            f = self._features(eng, B)
            feats = [[t] for t in f] if feats is None else [a + [t] for a, t in zip(feats, f)]
        y = torch.stack(segs, dim=1)  # [B, N, K]
        s1 = y[:, 0::2].reshape(B, -1)[:, P:]
        s2 = y[:, 1::2].reshape(B, -1)[:, :-P]
        out = ((s1 + s2) / 2)[:, :Lp]  # utility.over_add: average of the two streams, gap dropped
        if not flag:
            out = out[:, P:]
        ft = self._shape_features([torch.cat(f, dim=0) for f in feats], B)  # [N*B, C, F, T] (distillation_crn.py:473)
        return out, [f.to(mixture.device) for f in ft]

    def get_channel_num(self):  # distillation_crn.py:385-386
        c = self._cfg_args["num_channels"]
        return [c[-1], c[-1], c[2], c[1], c[0]]

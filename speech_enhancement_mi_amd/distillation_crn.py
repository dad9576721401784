"""Drop-in for the student architecture of the reference's distillation_crn.py (`TemporalCRN`, distillation_crn.py:
283-501; the distilled 0.81 M-parameter model is `TemporalCRN(num_channels=[16,32,64,64], hidden=128, ...)`,
distillation_crn.py:524-526).  Inference only: `realtime_process` returns `(pred, None)` - the reference returns the
five pre-activation feature maps in the second slot (distillation_crn.py:467-477), which only the distillation
*training* loss consumes (out of scope); `predict_distillation.py:84` discards them."""
from .crn import TemporalCRN as _Base


class TemporalCRN(_Base):
    _VARIANT = 2

    def realtime_process(self, mixture, flag=False):
        return super().realtime_process(mixture, flag), None

    def forward(self, x):
        return super().forward(x), None

    def get_channel_num(self):  # distillation_crn.py:385-386
        c = self._cfg_args["num_channels"]
        return [c[-1], c[-1], c[2], c[1], c[0]]

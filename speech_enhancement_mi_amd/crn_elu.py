"""Drop-in for the reference's CRN_ELU.py `TemporalCRN` (ELU, gated 1x1 convs, three frequency-dilated 5x5 preconv
blocks, atan2 phase; CRN_ELU.py:314-535): same constructor kwargs, state_dict keys and entry points as the reference
class; compute runs on the MI355X engine (se_config.variant = 1)."""
from .crn import TemporalCRN as _Base


class TemporalCRN(_Base):
    _VARIANT = 1

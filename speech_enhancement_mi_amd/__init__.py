"""MI355X-native streaming speech-enhancement engine: the reference's TemporalCRN model-class contract on
hand-written HIP kernels behind a C ABI (include/se_engine.h).  See DESIGN.md / INTEGRATION.md."""
from .crn import TemporalCRN  # noqa: F401

__all__ = ["TemporalCRN"]

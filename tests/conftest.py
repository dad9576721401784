import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "crn_golden.npz"))


TINY = dict(num_channels=[4, 8, 8, 8], num_freqs=201, hidden=16, segment_length=3200, num_layers=2,
            num_inputs=3, kernel_size=3, dropout=0.0, sample_rate=16000, win_length=25, hop_length=10, n_fft=400)
FULL400 = dict(num_channels=[16, 32, 64, 128], num_freqs=201, hidden=512, segment_length=3200, num_layers=2,
               num_inputs=3, kernel_size=3, dropout=0.0, sample_rate=16000, win_length=25, hop_length=10, n_fft=400)
FULL512 = dict(FULL400, num_freqs=257, n_fft=512)


def spec_of(cfg):
    from speech_enhancement_mi_amd import synth
    return synth.crn_param_spec(cfg["num_channels"], cfg["num_freqs"], cfg["hidden"], cfg["num_layers"],
                                cfg["num_inputs"], cfg["kernel_size"])


def rel_rms(a, b):
    import numpy as np
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / (np.sqrt(np.mean(b ** 2)) + 1e-30))

STUDENT400 = dict(FULL400, num_channels=[16, 32, 64, 64], hidden=128)  # distillation_crn.py:524-525


@pytest.fixture(scope="session")
def vgolden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "crn_variants_golden.npz"))


def spec_of_variant(cfg, variant):
    from speech_enhancement_mi_amd import synth
    return synth.crn_param_spec(cfg["num_channels"], cfg["num_freqs"], cfg["hidden"], cfg["num_layers"],
                                cfg["num_inputs"], cfg["kernel_size"], variant=variant)

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

FSN_FULL = dict(num_freqs=201, look_ahead=0, sequence_model="LSTM", fb_num_neighbors=0, sb_num_neighbors=15,
                fb_output_activate_function="ReLU", sb_output_activate_function=False, fb_model_hidden_size=512,
                sb_model_hidden_size=384, num_mics=3, norm_type="offline_laplace_norm", num_groups_in_drop_band=2, num_layers=2,
                weight_init=False, sample_rate=16000, segment_length=3200, win_length=25, hop_length=10, n_fft=400)  # config.yaml:153-172
FSN_TINY = dict(FSN_FULL, fb_model_hidden_size=16, sb_model_hidden_size=16)


@pytest.fixture(scope="session")
def fgolden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "fsn_golden.npz"))


def fsn_spec(cfg):
    from speech_enhancement_mi_amd import synth
    return synth.fsn_param_spec(cfg["num_freqs"], cfg["num_mics"], cfg["fb_model_hidden_size"], cfg["sb_model_hidden_size"],
                                cfg["num_layers"], cfg["sb_num_neighbors"], cfg["fb_num_neighbors"])

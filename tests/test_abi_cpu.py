"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/se_engine.h declares; no compute is attempted without a GPU."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "se_engine.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:se|fsn)_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    from speech_enhancement_mi_amd import engine
    names = _declared()
    assert "se_step" in names and "se_realtime_process" in names and "fsn_realtime_process" in names and len(names) >= 25
    lib = ctypes.CDLL(engine.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in se_engine.h but not exported by libse_engine.so"
    assert sorted(engine.EXPORTS) == names
    assert lib.se_abi_version() == 4


def test_no_cpu_fallback():
    """Without a GPU the engine must fail loudly, never compute on the host."""
    import torch
    from speech_enhancement_mi_amd import engine
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cfg = engine.make_config([4, 8, 8, 8], 201, 16, 3200, num_layers=2)
    with pytest.raises(RuntimeError, match="no HIP device|no CPU fallback|se_create failed"):
        engine.Engine(cfg)


def test_config_struct_sizes():
    """INTEGRATION.md's ctypes mirror of se_config must have the library's size (a short struct would leave `precision`
    reading whatever follows it): the library exports its own sizeof for the binding to check."""
    from speech_enhancement_mi_amd import engine
    lib = ctypes.CDLL(engine.LIB_PATH)
    assert lib.se_config_size() == ctypes.sizeof(engine.SeConfig)
    assert lib.fsn_config_size() == ctypes.sizeof(engine.FsnConfig)
    # the stub printed in INTEGRATION.md lists the same fields
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for name, _ in engine.SeConfig._fields_:
        assert f'"{name}"' in text, f"INTEGRATION.md SeConfig stub lacks field {name}"

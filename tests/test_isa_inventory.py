"""Build-time guard for the co-execution fault of DESIGN.md 3: kernels whose code contained packed-FP32 VALU instructions with
operand swizzles (v_pk_*_f32 with op_sel / neg_lo / neg_hi, v_pk_mov_b32 - formed by hipcc's SLP vectoriser from (re, im)
pairs) returned wrong values while co-resident with this library's MFMA kernels.  The whole library is therefore built with
-fno-slp-vectorize; this test disassembles the gfx950 code objects inside libse_engine.so and fails if such an instruction
comes back (a plain, modifier-free v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 written on purpose would be allowed)."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

from conftest import ROOT

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump not available")
def test_no_swizzled_packed_fp32_in_device_code():
    lib = os.path.join(ROOT, "speech_enhancement_mi_amd", "libse_engine.so")
    with tempfile.TemporaryDirectory() as d:
        shutil.copy(lib, os.path.join(d, "lib.so"))
        subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=d, capture_output=True, check=True)
        objs = [f for f in os.listdir(d) if f.endswith("gfx950")]
        assert objs, "no gfx950 code object found in libse_engine.so"
        n_mfma = 0
        bad = []
        for f in objs:
            asm = subprocess.run([OBJDUMP, "-d", f], cwd=d, capture_output=True, text=True, check=True).stdout
            n_mfma += len(re.findall(r"\bv_mfma_", asm))
            for line in asm.splitlines():
                m = re.search(r"\b(v_pk_(?:fma|mul|add)_f32|v_pk_mov_b32)\b(.*)", line)
                if not m:
                    continue
                mods = m.group(2)
                swizzled = m.group(1) == "v_pk_mov_b32" or re.search(r"neg_lo|neg_hi|op_sel:\[[01,]*1", mods)
                if swizzled:
                    bad.append(line.strip())
        assert n_mfma > 1000, "expected the hand-written MFMA kernels in the code objects"
        assert not bad, f"{len(bad)} swizzled packed-FP32 instructions, e.g. {bad[:3]}"

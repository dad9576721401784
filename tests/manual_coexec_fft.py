"""Manual reproducer (not collected by pytest) for the open finding in DESIGN.md 3/6: the LDS FFT kernels (k_stft,
k_istft) return wrong frames when their workgroups share CUs with this engine's MFMA kernels launched from ANOTHER
stream.  Two unrelated engine instances: A runs realtime_process single-stream on stream 1, B loops se_stft on stream 2;
B's outputs are compared with its own quiet result.  Expected on an MI355X: iteration 0 clean (allocations serialise the
streams), later iterations 30-40 % bad launches.  Usage:  python tests/manual_coexec_fft.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import FULL512  # noqa: E402
from speech_enhancement_mi_amd import synth  # noqa: E402
from test_gpu_parity import _engine  # noqa: E402

os.environ["SE_PIPELINE"] = "0"
eA, eB = _engine(FULL512, seed=1), _engine(FULL512, seed=2)
mix, _ = synth.synth_utterances(64, 16000, 3, seed=3)
x = torch.from_numpy(mix).cuda()
seg = torch.randn(96, 3200, device="cuda")
ref = eB.stft(seg).clone()
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
eA.realtime_process(x)
torch.cuda.synchronize()
for it in range(4):
    outs = []
    with torch.cuda.stream(s1):
        y = eA.realtime_process(x)
    with torch.cuda.stream(s2):
        for _ in range(200):
            outs.append(eB.stft(seg))
    torch.cuda.synchronize()
    bad = [k for k, o in enumerate(outs) if not torch.equal(o, ref)]
    print(f"iteration {it}: {len(bad)} of {len(outs)} stft launches differ from the quiet result")

#!/usr/bin/env python3
"""How long does the HOST take to enqueue one realtime_process (the call returns when everything is queued) against the time the GPU
takes to finish it?  enqueue ~ total => the stage pipeline is bound by the launch path, not by the kernels."""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from speech_enhancement_mi_amd import engine as E  # noqa: E402


def main():
    dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
    import numpy as np  # noqa: F401
    from speech_enhancement_mi_amd import synth
    cfg = bench.crn_cfg(512, "crn")
    variant = bench.MODELS["crn"][0]
    spec = synth.crn_param_spec(cfg["num_channels"], cfg["num_freqs"], cfg["hidden"], cfg["num_layers"], 3, 3, variant=variant)
    sd = synth.make_state_dict(spec, seed=0)
    B, L = 256, 48000
    eng = E.Engine(E.make_config(cfg["num_channels"], cfg["num_freqs"], cfg["hidden"], 3200, 2, 3, 3, 16000, 25, 10, 512, variant=variant,
                                 precision=bench.PRECISIONS[dtype]), 0)
    eng.load_state_dict(sd)
    mix = torch.randn(B, 3, L, device="cuda") * 0.1
    out = torch.empty(B, L, device="cuda")
    for _ in range(2):
        eng.realtime_process(mix, out=out)
    torch.cuda.synchronize()
    enq, tot = [], []
    for _ in range(5):
        t0 = time.perf_counter()
        eng.realtime_process(mix, out=out)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        enq.append((t1 - t0) * 1e3)
        tot.append((t2 - t0) * 1e3)
    print(f"{dtype}: host enqueue {sum(enq) / 5:.2f} ms, until the GPU is done {sum(tot) / 5:.2f} ms per realtime_process (34 windows)")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Training step (8 x 3 s, merged micro-batches, full loss): how long the HOST needs to enqueue forward / loss + backward against the time
until the GPU has finished them.  enqueue ~ total => the step is bound by the Python / ctypes launch path, not by the kernels."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from speech_enhancement_mi_amd import synth  # noqa: E402
from speech_enhancement_mi_amd.training import FlatBucket, TrainableCRN  # noqa: E402


def main():
    cfg = bench.crn_cfg(400)
    model = TrainableCRN(**cfg)
    spec = synth.crn_param_spec(cfg["num_channels"], cfg["num_freqs"], cfg["hidden"], cfg["num_layers"], 3, 3, variant=0)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(spec, seed=0).items()})
    model = model.cuda().use_hip_kernels(True)
    bucket = FlatBucket(list(model.parameters()))
    mix, clean = synth.synth_utterances(8, 48000, 3, seed=2000)
    mix, clean = torch.from_numpy(mix).cuda(), torch.from_numpy(clean).cuda()
    lens = torch.full((4,), 48000, dtype=torch.int64, device="cuda")
    rows = []
    for it in range(6):
        bucket.zero()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pred = model.realtime_process_train(mix)
        t1 = time.perf_counter()
        val = None
        for p_i, src in zip(pred.chunk(2), clean.chunk(2)):
            v = model.compute_loss(src, p_i, lens)[0] / 2
            val = v if val is None else val + v
        t2 = time.perf_counter()
        val.backward()
        t3 = time.perf_counter()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        rows.append([1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t0)])
    r = np.array(rows[2:]).mean(0)
    print(f"host enqueue: forward {r[0]:.2f} ms, loss {r[1]:.2f} ms, backward {r[2]:.2f} ms (sum {r[:3].sum():.2f}); until the GPU is done {r[3]:.2f} ms")


if __name__ == "__main__":
    main()

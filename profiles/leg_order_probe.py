#!/usr/bin/env python3
"""Why is the FullSubNet secondary leg of the default bench ~14 % slower than `bench.py --model fullsubnet`?  Same function, different
process history: run it alone, after a CRN measurement, and after a CRN measurement + gc."""
import argparse
import gc
import sys

sys.path.insert(0, ".")
import bench  # noqa: E402


def main():
    args = argparse.Namespace(seconds=3.0)
    mode = sys.argv[1]
    if mode in ("after_crn", "after_crn_gc"):
        v, ms, roof, info = bench.measure_crn("crn", 512, 256, "f32", 3.0, 3, 1, 0, 0, 1, "none")
        print("crn", round(v))
        del roof, info
        if mode == "after_crn_gc":
            gc.collect()
            import torch
            torch.cuda.empty_cache()
    if mode == "after_alloc":  # the same amount of device memory allocated and freed through hipMalloc / torch, no kernels
        import torch
        xs = [torch.empty(256 * 1024 * 1024, dtype=torch.float32, device="cuda") for _ in range(8)]
        del xs
        torch.cuda.empty_cache()
    reps = 2 if mode == "twice" else 1
    import contextlib
    import torch
    if mode == "after_crn_newstream":
        v, ms, roof, info = bench.measure_crn("crn", 512, 256, "f32", 3.0, 3, 1, 0, 0, 1, "none")
        del roof, info
        ctx = torch.cuda.stream(torch.cuda.Stream())  # a fresh HSA queue: the profiled CRN step recorded timed events on the old one
    else:
        ctx = contextlib.nullcontext()
    with ctx:
        for i in range(reps):
            r = bench.fullsubnet_measure(args, 0, 0, 1, "none", 256, "f32", 3, 1)
            print(mode, i, "fullsubnet f32", round(r["value"], 1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Condense rocprofv3 output directories into the small CSVs committed under profiles/.

  python profiles/summarize.py stats  <dir-of---kernel-trace---stats-run>  out.csv   # copies the *_kernel_stats.csv
  python profiles/summarize.py pmc    <dir-of---pmc-run> [<dir> ...]       out.csv   # per kernel x counter: mean, launches, sum

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB; bench.py's pmc_traffic() applies the gfx950 FETCH_SIZE x2
correction of MI355X_MICROARCH.md when it reads the summary.  Other counters are summarised in their native unit.

  python profiles/summarize.py hbm-json <pmc_hbm.csv> <workload-key> <source-note> out.json
      per base kernel name (template arguments dropped, launch-weighted): {"fetch_kb", "write_kb"} per launch, stamped with
      the sha of the kernel sources the passes ran on (bench.py refuses a summary whose stamp differs from its own sources)."""
import collections
import csv
import glob
import shutil
import sys


def base_name(k):
    k = k.split("(")[0].strip()
    if k.startswith("void "):
        k = k[5:]
    k = k.split("<")[0]
    return k.split("::")[-1]


def hbm_json(src, workload, note, out):
    import json
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from bench import kernel_source_sha
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(src)):
        if r["counter"] not in ("FETCH_SIZE", "WRITE_SIZE"):
            continue
        a = acc.setdefault(base_name(r["kernel"]), {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
        a[r["counter"]][0] += float(r["sum_KB"])
        a[r["counter"]][1] += int(r["launches"])
    kernels = {k: {"fetch_kb": v["FETCH_SIZE"][0] / max(v["FETCH_SIZE"][1], 1), "write_kb": v["WRITE_SIZE"][0] / max(v["WRITE_SIZE"][1], 1),
                   "launches": v["FETCH_SIZE"][1]} for k, v in acc.items() if k.startswith("k_")}
    json.dump({"src_sha": kernel_source_sha(), "workload": workload, "source": note, "kernels": kernels}, open(out, "w"), indent=1)


def main():
    if sys.argv[1] == "hbm-json":
        return hbm_json(*sys.argv[2:6])
    mode, *dirs, out = sys.argv[1:]
    if mode == "stats":
        src = glob.glob(dirs[0] + "/**/*kernel_stats.csv", recursive=True)
        shutil.copy(src[0], out)
        return
    agg = collections.OrderedDict()
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = (r["Kernel_Name"], r["Counter_Name"])
                a = agg.setdefault(k, [0.0, 0])
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    with open(out, "w", newline="") as fp:
        w = csv.writer(fp)
        w.writerow(["kernel", "counter", "mean_KB_per_launch", "launches", "sum_KB"])
        for (k, c), (s, n) in sorted(agg.items(), key=lambda kv: (kv[0][1], kv[0][0])):
            w.writerow([k, c, s / n, n, s])


if __name__ == "__main__":
    main()

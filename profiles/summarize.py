#!/usr/bin/env python3
"""Condense rocprofv3 output directories into the small CSVs committed under profiles/.

  python profiles/summarize.py stats  <dir-of---kernel-trace---stats-run>  out.csv   # copies the *_kernel_stats.csv
  python profiles/summarize.py pmc    <dir-of---pmc-run> [<dir> ...]       out.csv   # per kernel x counter: mean, launches, sum

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB; bench.py's pmc_traffic() applies the gfx950 FETCH_SIZE x2
correction of MI355X_MICROARCH.md when it reads the summary.  Other counters are summarised in their native unit."""
import collections
import csv
import glob
import shutil
import sys


def main():
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

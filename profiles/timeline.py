#!/usr/bin/env python3
"""Timeline summary of a rocprofv3 --kernel-trace run of the PIPELINED bench (three stage streams): per kernel the mean duration under
overlap, and for the recurrence's step launches the gap between one step's end and the next one's start (the steps are a dependent
chain: a gap is time the chain waits for CU slots or for its launch), plus how much of the wall time has 1 / 2 / 3+ kernels in flight.

  python profiles/timeline.py <dir-with-*_kernel_trace.csv> out.txt"""
import collections
import csv
import glob
import sys


def main():
    src = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(src)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # the timed steps only: from the first pipelined step launch after the warm-up step (1 warm-up + 3 timed steps: the second quarter of the
    # k_gru_step_multi launches onwards) to the last one; bench.py's extra per-launch-event step (serial order) comes later and is left out
    multi = [r for r in rows if "gru_step_multi" in r[2]]
    if multi:
        t_lo, t_hi = multi[len(multi) // 4][0], multi[-1][1]
        rows = [r for r in rows if r[0] >= t_lo and r[1] <= t_hi]
    elif len(sys.argv) > 3:  # other workloads: the window between the first and the last launch of a named kernel's last `frac` of launches
        name, frac = sys.argv[3], float(sys.argv[4]) if len(sys.argv) > 4 else 0.5
        sel = [r for r in rows if name in r[2]]
        t_lo, t_hi = sel[int(len(sel) * (1 - frac))][0], sel[-1][1]
        rows = [r for r in rows if r[0] >= t_lo and r[1] <= t_hi]
    out = open(sys.argv[2], "w")

    def base(k):
        k = k.split("(")[0].replace("void ", "").strip()
        return k.split("::")[-1]

    per = collections.OrderedDict()
    for s, e, k in rows:
        a = per.setdefault(base(k), [0, 0])
        a[0] += 1
        a[1] += e - s
    wall = rows[-1][1] - rows[0][0]
    print(f"window {wall / 1e6:.2f} ms, {len(rows)} launches, sum of durations {sum(v[1] for v in per.values()) / 1e6:.2f} ms", file=out)
    for k, (n, d) in sorted(per.items(), key=lambda x: -x[1][1])[:20]:
        print(f"  {k:44s} {n:6d} launches  mean {d / n / 1e3:8.1f} us  sum {d / 1e6:8.2f} ms ({100 * d / wall:5.1f} % of wall)", file=out)
    # concurrency histogram
    ev = []
    for s, e, _ in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    hist = collections.Counter()
    cur, last = 0, ev[0][0]
    for t, d in ev:
        hist[min(cur, 4)] += t - last
        cur += d
        last = t
    print("kernels in flight: " + ", ".join(f"{k}{'+' if k == 4 else ''}: {100 * v / wall:.1f} %" for k, v in sorted(hist.items())), file=out)
    # the recurrence chain
    steps = [(s, e) for s, e, k in rows if "gru_step" in k]
    if len(steps) > 2:
        gaps = [steps[i + 1][0] - steps[i][1] for i in range(len(steps) - 1)]
        durs = [e - s for s, e in steps]
        gaps_sorted = sorted(gaps)
        print(f"recurrence: {len(steps)} step launches, mean duration {sum(durs) / len(durs) / 1e3:.1f} us, gap to the next step: median "
              f"{gaps_sorted[len(gaps) // 2] / 1e3:.1f} us, mean {sum(gaps) / len(gaps) / 1e3:.1f} us, p90 {gaps_sorted[int(0.9 * len(gaps))] / 1e3:.1f} us; "
              f"chain busy {100 * sum(durs) / wall:.1f} % of wall", file=out)
    out.close()
    print(open(sys.argv[2]).read())


if __name__ == "__main__":
    main()

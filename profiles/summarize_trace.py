#!/usr/bin/env python3
"""Per-kernel statistics of the STEADY-STATE steps of a rocprofv3 --kernel-trace CSV.

rocprofv3 --stats aggregates the whole process, which for bench.py is dominated by MIOpen's one-off
algorithm search during the warm-up steps.  This script keeps only the dispatches between the decode
kernel of step (last - N) and the decode kernel of the last step (N whole steps) and prints the same
columns as the --stats summary.

usage: summarize_trace.py <kernel_trace.csv> <n_steps> [out.csv] [--skip-last K] [--marker KERNEL_SUBSTRING]

--skip-last K leaves out the last K steps (bench.py appends 1 + --exclusive-steps serialised steps after the
timed region when kernel timing is on; pass K = 3 for the default command so the window is the timed region).
Steps are delimited by the teacher-decode launch; with the look-ahead teacher pass that launch sits inside the
previous step's backward, which shifts the window by a fraction of a step but keeps it n whole periods long.
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:<>, ]+?)\(", name)
    return (m.group(1) if m else name)[:96]


def main():
    argv = list(sys.argv)
    skip = 0
    if "--skip-last" in argv:
        i = argv.index("--skip-last")
        skip = int(argv[i + 1])
        del argv[i:i + 2]
    marker = "decode_filter_kernel<true"
    if "--marker" in argv:                       # substring of the kernel name that delimits a step / frame
        i = argv.index("--marker")
        marker = argv[i + 1]
        del argv[i:i + 2]
    sys.argv = argv
    path, n = sys.argv[1], int(sys.argv[2])
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [int(r["Start_Timestamp"]) for r in rows if marker in r["Kernel_Name"]]
    assert len(marks) > n + skip, "need more than %d decode launches, found %d" % (n + skip, len(marks))
    t0, t1 = marks[-n - 1 - skip], marks[-1 - skip]
    agg = defaultdict(lambda: [0, 0, 10 ** 18, 0])
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if t0 <= s < t1:
            a = agg[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += e - s
            a[2] = min(a[2], e - s)
            a[3] = max(a[3], e - s)
    wall = t1 - t0
    busy = sum(a[1] for a in agg.values())
    # kernels of the two compute streams overlap: the union of their intervals is the time the GPU was busy
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if t0 <= int(r["Start_Timestamp"]) < t1)
    union, cs, ce = 0, iv[0][0], iv[0][1]
    for a, b in iv[1:]:
        if a > ce:
            union += ce - cs
            cs, ce = a, b
        else:
            ce = max(ce, b)
    union += ce - cs
    out = [("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.append((k, a[0], a[1], round(a[1] / a[0], 1), round(100.0 * a[1] / busy, 2), a[2], a[3]))
    w = csv.writer(open(sys.argv[3], "w", newline="")) if len(sys.argv) > 3 else None
    print("# window: %d steps, wall %.3f ms/step, sum of kernel durations %.3f ms/step, GPU busy (union of kernel "
          "intervals) %.3f ms/step = %.1f%% of wall, %d launches/step"
          % (n, wall / n / 1e6, busy / n / 1e6, union / n / 1e6, 100.0 * union / wall, len(iv) // n))
    for row in out:
        if w:
            w.writerow(row)
    for row in out[:40]:
        print("%-96s %7s %14s %12s %7s" % row[:5])


if __name__ == "__main__":
    main()

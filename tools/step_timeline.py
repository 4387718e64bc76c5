#!/usr/bin/env python3
"""Per-stream timeline of steady bench steps from a rocprofv3 --kernel-trace CSV.

    step_timeline.py --extract <kernel_trace.csv> <out.csv> [--steps N] [--skip-last K]
        keep the raw dispatch rows of N whole steps (delimited by the teacher-decode launch) before the last K
    step_timeline.py <window.csv> [--marker SUBSTR]
        print, per step of the window and averaged: per-stream busy / idle time and launch counts, how long 0 / 1 /
        2 / 3+ kernels were in flight, how long a kernel that cannot fill the chip ran alone, the longest device-idle
        gaps with the kernels around them, and a backward walk from the step's last kernel over "what did this
        launch wait for" (same-stream predecessor, a join with another stream, or nothing on the device = the host)

Streams are told apart by the Stream_Id column when the trace has one, else by Queue_Id.
"""
import csv
import re
import sys
from collections import defaultdict

MARKER = "decode_filter_kernel<true"
CUS = 256


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:<>, ]+?)\(", name)
    return (m.group(1) if m else name)[:72]


def klass(name):
    s = short(name)
    for key in ("gemm_ks4", "gemm_kernel", "mha_fwd", "mha_bwd", "mha_", "wino_", "bn_", "ln_", "im2col", "col2im",
                "maxpool", "adamw", "ema", "dropout", "colsum", "batch_sum", "transt_loss", "preprocess"):
        if key in s:
            return key.rstrip("_")
    if "at::native" in name or "elementwise" in name:
        return "framework"
    return s[:24]


def load(path):
    rows = list(csv.DictReader(open(path)))
    key = "Stream_Id" if rows and "Stream_Id" in rows[0] else "Queue_Id"
    out = []
    for r in rows:
        wg = 1
        gs = 1
        for ax in ("X", "Y", "Z"):
            g = int(r.get("Grid_Size_" + ax, r.get("Grid_Size", 1)) or 1)
            w = int(r.get("Workgroup_Size_" + ax, r.get("Workgroup_Size", 1)) or 1)
            gs *= g
            wg *= w
            if "Grid_Size_" + ax not in r:
                break
        out.append({"s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"]), "q": r[key], "name": r["Kernel_Name"],
                    "wgs": max(1, gs // max(1, wg)), "wg": wg, "lds": int(r.get("LDS_Block_Size", 0) or 0),
                    "vgpr": int(r.get("VGPR_Count", 0) or 0)})
    out.sort(key=lambda r: r["s"])
    return out, key


def extract(argv):
    src, dst = argv[0], argv[1]
    n = int(argv[argv.index("--steps") + 1]) if "--steps" in argv else 4
    skip = int(argv[argv.index("--skip-last") + 1]) if "--skip-last" in argv else 0
    rows = list(csv.DictReader(open(src)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [int(r["Start_Timestamp"]) for r in rows if MARKER in r["Kernel_Name"]]
    assert len(marks) > n + skip, "not enough steps in the trace"
    t0, t1 = marks[-n - 1 - skip], marks[-1 - skip]
    keep = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1]
    with open(dst, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(keep)
    print("kept %d dispatches of %d steps (%.3f ms/step)" % (len(keep), n, (t1 - t0) / n / 1e6))


def union(iv):
    tot, cs, ce = 0, None, None
    for a, b in sorted(iv):
        if cs is None:
            cs, ce = a, b
        elif a > ce:
            tot += ce - cs
            cs, ce = a, b
        else:
            ce = max(ce, b)
    return tot + (ce - cs if cs is not None else 0)


def full_grid(r):
    """Workgroups a kernel needs to give every CU one: below it the launch leaves CUs idle when it runs alone."""
    return r["wgs"] >= CUS


def analyse(rows, t0, t1, streams, out):
    inside = [r for r in rows if t0 <= r["s"] < t1]
    wall = t1 - t0
    out.append("step wall %.3f ms, %d launches, sum of kernel durations %.3f ms" %
               (wall / 1e6, len(inside), sum(r["e"] - r["s"] for r in inside) / 1e6))
    res = {"wall": wall, "launches": len(inside)}
    for q in streams:
        mine = [r for r in inside if r["q"] == q]
        if not mine:
            continue
        busy = union([(r["s"], min(r["e"], t1)) for r in mine])
        first, last = mine[0]["s"], max(r["e"] for r in mine)
        out.append("  stream %-6s %5d launches  busy %7.3f ms (%4.1f%% of wall)  active span %7.3f..%7.3f ms  idle inside span %7.3f ms"
                   % (q, len(mine), busy / 1e6, 100.0 * busy / wall, (first - t0) / 1e6, (last - t0) / 1e6,
                      (min(last, t1) - first - busy) / 1e6))
        res["busy_" + q] = busy
    # sweep: number of kernels in flight, and "one under-filled kernel alone"
    ev = []
    for i, r in enumerate(inside):
        ev.append((r["s"], 1, i))
        ev.append((min(r["e"], t1), -1, i))
    ev.sort(key=lambda x: (x[0], x[1]))
    live = set()
    hist = defaultdict(int)
    alone_small = 0
    alone_small_by = defaultdict(int)
    streams_live = defaultdict(int)
    prev = t0
    gaps = []
    for t, d, i in ev:
        dt = t - prev
        if dt > 0:
            k = len(live)
            hist[min(k, 3)] += dt
            streams_live[len({inside[j]["q"] for j in live})] += dt
            if k == 1:
                r = inside[next(iter(live))]
                if not full_grid(r):
                    alone_small += dt
                    alone_small_by[klass(r["name"])] += dt
            if k == 0:
                gaps.append((dt, prev, t))
        prev = t
        if d == 1:
            live.add(i)
        else:
            live.discard(i)
    out.append("  kernels in flight: 0: %.3f ms   1: %.3f ms   2: %.3f ms   3+: %.3f ms" %
               tuple(hist[k] / 1e6 for k in range(4)))
    out.append("  streams active:    0: %.3f ms   1: %.3f ms   2: %.3f ms   3: %.3f ms" %
               tuple(streams_live[k] / 1e6 for k in range(4)))
    out.append("  a kernel of < %d workgroups alone on the device: %.3f ms  (%s)" %
               (CUS, alone_small / 1e6, ", ".join("%s %.2f" % (k, v / 1e6) for k, v in
                                                   sorted(alone_small_by.items(), key=lambda kv: -kv[1])[:8])))
    res.update(idle=hist[0], one=hist[1], two=hist[2], three=hist[3], alone_small=alone_small)
    res["alone_small_by"] = dict(alone_small_by)
    gaps.sort(reverse=True)
    res["gaps"] = gaps
    # device-idle gaps by size class
    small = sum(g[0] for g in gaps if g[0] < 5000)
    mid = sum(g[0] for g in gaps if 5000 <= g[0] < 50000)
    big = sum(g[0] for g in gaps if g[0] >= 50000)
    out.append("  device idle: %d gaps; < 5 us: %.3f ms, 5-50 us: %.3f ms, >= 50 us: %.3f ms" %
               (len(gaps), small / 1e6, mid / 1e6, big / 1e6))
    for g, a, b in gaps[:6]:
        before = max((r for r in inside if r["e"] <= a + 1), key=lambda r: r["e"], default=None)
        after = min((r for r in inside if r["s"] >= b - 1), key=lambda r: r["s"], default=None)
        out.append("    idle %7.1f us at %7.3f ms: after %s, before %s" %
                   (g / 1e3, (a - t0) / 1e6, short(before["name"]) if before else "-", short(after["name"]) if after else "-"))
    # backward walk over "what did this launch wait for"
    by_q = defaultdict(list)
    for r in inside:
        by_q[r["q"]].append(r)
    ends = sorted(inside, key=lambda r: r["e"])
    cur = max(inside, key=lambda r: r["e"])
    path_time = defaultdict(int)
    wait_same, wait_join, wait_host = 0, 0, 0
    n_join = 0
    join_on = defaultdict(int)
    steps = 0
    import bisect
    end_keys = [r["e"] for r in ends]
    while cur is not None and steps < 100000:
        steps += 1
        path_time[klass(cur["name"])] += cur["e"] - cur["s"]
        # latest-finishing kernel that ended before cur started
        j = bisect.bisect_right(end_keys, cur["s"] + 200) - 1
        pred = None
        while j >= 0:
            c = ends[j]
            if c is not cur and c["s"] < cur["s"]:
                pred = c
                break
            j -= 1
        if pred is None:
            break
        gap = max(0, cur["s"] - pred["e"])
        if pred["q"] == cur["q"]:
            wait_same += gap
        else:
            # same-stream predecessor of cur
            mine = by_q[cur["q"]]
            k = mine.index(cur)
            sp = mine[k - 1] if k > 0 else None
            if sp is not None and sp["e"] > pred["e"] - 200:
                pred = sp
                gap = max(0, cur["s"] - pred["e"])
                wait_same += gap
            else:
                n_join += 1
                join_on[klass(pred["name"])] += 1
                wait_join += gap
        if gap > 5000:
            wait_host += gap
        cur = pred
    tot = sum(path_time.values())
    out.append("  backward walk from the last kernel (predecessor = the launch that finished last before this one "
               "started): %d hops, kernel time on the walk %.3f ms, of the gaps %.3f ms exceed 5 us (host / join latency); "
               "%d hops cross streams" % (steps, tot / 1e6, wait_host / 1e6, n_join))
    out.append("    walk by class: " + ", ".join("%s %.2f" % (k, v / 1e6) for k, v in
                                                 sorted(path_time.items(), key=lambda kv: -kv[1])[:12]))
    out.append("    joins waited on: " + ", ".join("%s x%d" % kv for kv in sorted(join_on.items(), key=lambda kv: -kv[1])[:8]))
    res["walk"] = tot
    return res


def main():
    argv = sys.argv[1:]
    if argv and argv[0] == "--extract":
        return extract(argv[1:])
    global MARKER
    if "--marker" in argv:
        i = argv.index("--marker")
        MARKER = argv[i + 1]
        del argv[i:i + 2]
    rows, key = load(argv[0])
    marks = [r["s"] for r in rows if MARKER in r["name"]]
    bounds = marks + [max(r["e"] for r in rows)]
    streams = sorted({r["q"] for r in rows}, key=lambda q: -sum(1 for r in rows if r["q"] == q))
    out = ["# per-stream timeline of %d steady steps (streams by %s; times in ms from the step's first teacher-decode launch)"
           % (len(marks), key)]
    agg = defaultdict(float)
    n = 0
    for a, b in zip(bounds[:-1], bounds[1:]):
        if b - a < 1e6:
            continue
        out.append("")
        res = analyse(rows, a, b, streams, out)
        n += 1
        for k, v in res.items():
            if isinstance(v, (int, float)):
                agg[k] += v
    out.append("")
    out.append("# mean over %d steps: wall %.3f ms, launches %d, device idle %.3f ms, one kernel in flight %.3f ms, two %.3f ms, "
               "three or more %.3f ms, under-filled kernel alone %.3f ms"
               % (n, agg["wall"] / n / 1e6, agg["launches"] / n, agg["idle"] / n / 1e6, agg["one"] / n / 1e6,
                  agg["two"] / n / 1e6, agg["three"] / n / 1e6, agg["alone_small"] / n / 1e6))
    # per-class totals over the window, with grid-size classes
    by = defaultdict(lambda: [0, 0, 0])
    for r in rows:
        a = by[klass(r["name"])]
        a[0] += 1
        a[1] += r["e"] - r["s"]
        a[2] += 0 if full_grid(r) else 1
    out.append("# per class over the window (per step): launches, kernel ms, launches with < %d workgroups" % CUS)
    for k, a in sorted(by.items(), key=lambda kv: -kv[1][1])[:24]:
        out.append("  %-24s %6.1f %8.3f %6.1f" % (k, a[0] / n, a[1] / n / 1e6, a[2] / n))
    # how much of each class's kernel time passes with no other kernel in flight (the matrix pipes idle under a
    # memory-bound kernel that runs alone, the memory system under a matrix-bound one)
    ev = []
    for r in rows:
        ev.append((r["s"], 1, klass(r["name"])))
        ev.append((r["e"], -1, klass(r["name"])))
    ev.sort()
    active, alone, beside, last = defaultdict(int), defaultdict(float), defaultdict(float), ev[0][0]
    for t, d, c in ev:
        live = [(k, v) for k, v in active.items() if v]
        tot = sum(v for _, v in live)
        for k, v in live:
            (alone if tot == 1 else beside)[k] += (t - last) * v
        active[c] += d
        last = t
    out.append("# per class (per step): kernel ms with no other kernel in flight / beside at least one other kernel")
    for k in sorted(set(alone) | set(beside), key=lambda k: -(alone[k] + beside[k]))[:16]:
        out.append("  %-24s %8.3f %8.3f" % (k, alone[k] / n / 1e6, beside[k] / n / 1e6))
    print("\n".join(out))


if __name__ == "__main__":
    main()

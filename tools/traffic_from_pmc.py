#!/usr/bin/env python3
"""HBM traffic per launch of the library's kernels from two rocprofv3 --pmc passes over bench.py
(one with FETCH_SIZE, one with WRITE_SIZE; the TCC block cannot hold both at once).

usage: traffic_from_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

Corrections applied (gfx950): FETCH_SIZE and WRITE_SIZE are reported in KiB; FETCH_SIZE tallies the 128-byte
requests of wide coalesced reads at 64 bytes, so it is doubled.  The EMA kernel's launches (a pure stream of
known size: 8 bytes read + 4 written per parameter) are printed as the calibration check.
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:<>, ]+?)\(", name)
    return (m.group(1) if m else name)[:96]


def per_kernel(path, counter):
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = agg[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        if not any(t in k for t in ("gemm_kernel", "gemm_ring_kernel", "gemm_ks4", "mha_", "ema_kernel", "ln_", "bn_", "groupdw", "colsum", "wino_")):
            continue
        nf, f = fetch.get(k, (0, 0.0))
        nw, w = write.get(k, (0, 0.0))
        rd = 2.0 * f * 1024.0 / max(nf, 1)
        wr = w * 1024.0 / max(nw, 1)
        out[k] = {"launches": nf or nw, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                  "hbm_bytes_per_launch": rd + wr}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE over bench.py (separate passes); FETCH_SIZE "
                         "doubled (gfx950 counts 128-byte read requests as 64 bytes); KiB -> bytes",
               "kernels": out}, open(sys.argv[3], "w"), indent=1)
    for k, v in out.items():
        print("%-40s n=%5d  read %10.2f MB  write %10.2f MB" % (k[:40], v["launches"], v["read_bytes_per_launch"] / 1e6,
                                                                v["write_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()

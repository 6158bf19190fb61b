#!/usr/bin/env python3
"""Reduce two rocprofv3 counter-collection CSVs (one FETCH_SIZE pass, one WRITE_SIZE pass) to per-launch HBM-side
traffic per kernel family.

usage: tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv>

rocprofv3 reports both counters in KiB.  On gfx950 FETCH_SIZE tallies the 128-byte requests of wide (16 bytes per
lane) coalesced reads at 64 bytes, so it is doubled; WRITE_SIZE is taken as reported.  Output: JSON, one entry per
kernel family (segk kernels by their function name, everything else lumped), bytes per launch.
"""
import collections
import csv
import json
import re
import sys


def family(name):
    m = re.search(r"segk::(\w+)", name)
    return m.group(1) if m else "other"


def collect(path, counter):
    tot = collections.defaultdict(float)
    n = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        f = family(r["Kernel_Name"])
        tot[f] += float(r["Counter_Value"])
        n[f].add(r["Dispatch_Id"])
    return tot, {k: len(v) for k, v in n.items()}


def main():
    fetch, nf = collect(sys.argv[1], "FETCH_SIZE")
    write, nw = collect(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        lf, lw = max(nf.get(k, 0), 1), max(nw.get(k, 0), 1)
        rd = 2.0 * fetch.get(k, 0.0) * 1024.0 / lf
        wr = write.get(k, 0.0) * 1024.0 / lw
        out[k] = {"launches": nf.get(k, 0), "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                  "hbm_bytes_per_launch": rd + wr}
    # tools/pmc_traffic.sh profiles `bench.py --steps 2 --warmup 1`: three steps
    # the hash of the convolution source these passes profiled: bench.py reports the figures only while it stands
    import hashlib
    import os
    conv = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "semantic-segmentation_amd", "csrc",
                        "conv.hip")
    with open(conv, "rb") as f:
        sha = hashlib.sha1(f.read()).hexdigest()
    json.dump({"unit": "bytes per launch (FETCH_SIZE x2 x1024, WRITE_SIZE x1024)", "bench_steps": 3,
               "conv_hip_sha1": sha, "kernels": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()

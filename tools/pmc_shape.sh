#!/bin/bash
# usage: tools/pmc_shape.sh <substring of a tools/conv_bench.py case>      (GPU box, from the repository root)
# HBM-side fetch of ONE convolution shape: a FETCH_SIZE counter pass over `conv_bench.py --only <case>`, printed per
# dispatch of the segk convolution kernels (KiB as reported x2 for 16-byte-per-lane reads, see tools/pmc_traffic.py).
set -e -o pipefail
only=$1
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_shape
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_shape -o p --output-format csv -- \
    python3 "$repo/tools/conv_bench.py" --only "$only" > /tmp/pmc_shape.out 2>&1 || { tail -5 /tmp/pmc_shape.out; exit 1; }
python3 - "$(find /tmp/pmc_shape -name '*counter_collection.csv' | head -1)" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "segk::conv" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
agg = collections.defaultdict(list)
for r in rows:
    agg[(r["Kernel_Name"][:60], r["Grid_Size"])].append(2.0 * float(r["Counter_Value"]) * 1024 / 1e6)
for k, v in agg.items():
    print("%-62s grid %-9s launches %3d  fetch %8.1f MB/launch (min %.1f max %.1f)" % (k[0], k[1], len(v), sum(v) / len(v), min(v), max(v)))
PY

#!/bin/bash
# usage: tools/pmc_any.sh "<counters>" <conv_bench case substring>     (GPU box, from the repository root)
# One rocprofv3 counter pass over `conv_bench.py --only <case>`: per segk convolution kernel, the counters' averages
# per launch.
set -e -o pipefail
ctr=$1; only=$2
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_any
timeout -s KILL 200 rocprofv3 --pmc $ctr -d /tmp/pmc_any -o p --output-format csv -- \
    python3 "$repo/tools/conv_bench.py" --only "$only" > /tmp/pmc_any.out 2>&1 || { tail -5 /tmp/pmc_any.out; exit 1; }
python3 - "$(find /tmp/pmc_any -name '*counter_collection.csv' | head -1)" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "segk::conv" in r["Kernel_Name"]:
        agg[(r["Kernel_Name"][:60], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    print("%s grid %s" % k)
    for name, v in sorted(c.items()):
        print("    %-28s %16.0f  (launches %d)" % (name, sum(v) / len(v), len(v)))
PY

#!/bin/bash
# usage: tools/prof_conv_bench.sh <tag> [conv_bench.py flags...]     (GPU box, from the repository root)
# Kernel durations of tools/conv_bench.py from a rocprofv3 kernel trace: its own HIP-event timing includes the host's
# ~50 us per call, which hides everything about launches shorter than that.
set -e -o pipefail
tag=$1; shift
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pcb_$tag
timeout -k 10 200 rocprofv3 --kernel-trace -d /tmp/pcb_$tag -o t --output-format csv -- \
  python3 "$repo/tools/conv_bench.py" "$@" > /tmp/pcb_$tag.out 2>&1 || { tail -5 /tmp/pcb_$tag.out; exit 1; }
python3 - "$(find /tmp/pcb_$tag -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "segk::conv" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# consecutive launches of one (kernel, grid) = one conv_bench case (12 launches each: 2 warm-up + 10 timed)
groups = []
for r in rows:
    key = (r["Kernel_Name"][:58], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if groups and groups[-1][0] == key:
        groups[-1][1].append(d)
    else:
        groups.append((key, [d]))
for key, ds in groups:
    # conv_bench issues 12 launches per case; cases with the same kernel and grid follow one another
    for c in range(0, len(ds) - len(ds) % 12 if len(ds) >= 12 else 0, 12):
        part = sorted(ds[c:c + 12])
        print("%-60s grid %-8s y %-3s  median %8.1f us  min %8.1f" % (key[0], key[1], key[2], part[6], part[0]))
PY

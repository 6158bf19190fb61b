#!/bin/bash
# usage: tools/pmc_conv_bench.sh [conv_bench.py flags...]      (GPU box, from the repository root)
# HBM-side fetch and write of every tools/conv_bench.py case: one FETCH_SIZE and one WRITE_SIZE counter pass, printed
# per case in the order of the tool's case list (12 launches each; MB per launch: FETCH x2 as tools/pmc_traffic.py).
set -e -o pipefail
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pcbm_$c
  timeout -s KILL 200 rocprofv3 --pmc $c -d /tmp/pcbm_$c -o p --output-format csv -- \
    python3 "$repo/tools/conv_bench.py" "$@" > /tmp/pcbm_$c.out 2>&1 || { tail -5 /tmp/pcbm_$c.out; exit 1; }
done
python3 - "$(find /tmp/pcbm_FETCH_SIZE -name '*counter_collection.csv' | head -1)" \
          "$(find /tmp/pcbm_WRITE_SIZE -name '*counter_collection.csv' | head -1)" <<'PY'
import csv, sys


def cases(path, counter, scale):
    rows = [r for r in csv.DictReader(open(path)) if "segk::conv" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    groups = []
    for r in rows:
        key = (r["Kernel_Name"][:58], r["Grid_Size"])
        v = scale * float(r["Counter_Value"]) * 1024 / 1e6
        if groups and groups[-1][0] == key:
            groups[-1][1].append(v)
        else:
            groups.append((key, [v]))
    out = []
    for key, vs in groups:
        for c in range(0, len(vs) - len(vs) % 12 if len(vs) >= 12 else 0, 12):
            part = sorted(vs[c:c + 12])
            out.append((key, part[6]))
    return out


f, w = cases(sys.argv[1], "FETCH_SIZE", 2.0), cases(sys.argv[2], "WRITE_SIZE", 1.0)
for (key, fv), (_, wv) in zip(f, w):
    print("%-60s grid %-8s  fetch %8.1f MB  write %8.1f MB" % (key[0], key[1], fv, wv))
PY

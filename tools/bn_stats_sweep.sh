#!/bin/bash
# usage: tools/bn_stats_sweep.sh "<blocks> ..."   (GPU box, experiment build): kernel durations of the batch-norm backward
# statistics pass (tools/bn_stats_probe.py) for several workgroup counts (SEGK_BN_STAT_BLOCKS)
set -e -o pipefail
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
for cfg in ${1:-512}; do
  export SEGK_BN_STAT_BLOCKS=$cfg
  rm -rf /tmp/bns
  timeout -k 10 200 rocprofv3 --kernel-trace -d /tmp/bns -o t --output-format csv -- python3 "$repo/tools/bn_stats_probe.py" > /tmp/bns.out 2>&1 || { tail -5 /tmp/bns.out; exit 1; }
  echo "== $SEGK_BN_STAT_BLOCKS workgroups"
  python3 - "$(find /tmp/bns -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, re, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "bn_bwd_stats_pm" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sets = [int(m) for m in re.findall(r"sets (\d+)", open("/tmp/bns.out").read())]
print("   bwd_stats  bwd_stats+bits   (us, median per shape, operands cold)")
pos = 0
for n in sets:
    pos += n + 1                  # warm-up launches, and the one that made the coefficients' sums... (bn_bwd_local_sums is not called)
    pos -= 1
    blk = rows[pos:pos + 12 * n]
    pos += 12 * n
    meds = []
    for k in range(2):
        ds = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                    for rnd in range(6) for r in blk[(rnd * 2 + k) * n:(rnd * 2 + k + 1) * n])
        meds.append(ds[len(ds) // 2])
    print("  %8.1f  %10.1f" % tuple(meds))
PY
done

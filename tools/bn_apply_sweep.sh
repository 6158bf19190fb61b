#!/bin/bash
# usage: tools/bn_apply_sweep.sh "<wgs:minrows> ..."   (GPU box, experiment build): kernel durations of the batch-norm
# apply passes (tools/bn_apply_probe.py) for several grid rules of pm_apply_grid
set -e -o pipefail
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
for cfg in ${1:-2048:4}; do
  # cfg = wgs:minrows[:bwd slab:bwd wgs:bwd minrows]
  IFS=: read -r a b c d e <<< "$cfg"
  export SEGK_BN_APPLY_WGS=$a SEGK_BN_APPLY_MINROWS=$b SEGK_BN_BAPPLY_SLAB=${c:-0} SEGK_BN_BAPPLY_WGS=${d:-0} SEGK_BN_BAPPLY_MINROWS=${e:-0}
  rm -rf /tmp/bnp
  timeout -k 10 200 rocprofv3 --kernel-trace -d /tmp/bnp -o t --output-format csv -- python3 "$repo/tools/bn_apply_probe.py" > /tmp/bnp.out 2>&1 || { tail -5 /tmp/bnp.out; exit 1; }
  echo "== workgroups $SEGK_BN_APPLY_WGS, at least $SEGK_BN_APPLY_MINROWS pixels per thread; backward: slab $SEGK_BN_BAPPLY_SLAB workgroups $SEGK_BN_BAPPLY_WGS min $SEGK_BN_BAPPLY_MINROWS"
  python3 - "$(find /tmp/bnp -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "bn_apply_pm" in r["Kernel_Name"] or "bn_bwd_apply_pm" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# per shape: a warm-up launch per operand set, then 6 rounds x 4 kinds x sets launches; the probe prints the set counts
import re
sets = [int(m) for m in re.findall(r"sets (\d+)", open("/tmp/bnp.out").read())]
print("   apply  apply+res  bwd_apply  bwd_apply+bits   (us, median per shape, operands cold)")
pos = 0
for n in sets:
    pos += n                      # warm-up launches
    blk = rows[pos:pos + 24 * n]
    pos += 24 * n
    meds = []
    for k in range(4):
        ds = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                    for rnd in range(6) for r in blk[(rnd * 4 + k) * n:(rnd * 4 + k + 1) * n])
        meds.append(ds[len(ds) // 2])
    print("  %6.1f  %8.1f  %9.1f  %9.1f" % tuple(meds))
PY
done

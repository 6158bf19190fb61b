#!/bin/bash
# usage: tools/hbm_cold_probe.sh   (GPU box, repository root): kernel durations of tools/hbm_cold_probe.py
set -e -o pipefail
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/hcp
timeout -k 10 200 rocprofv3 --kernel-trace -d /tmp/hcp -o t --output-format csv -- python3 "$repo/tools/hbm_cold_probe.py" > /tmp/hcp.out 2>&1 || { tail -5 /tmp/hcp.out; exit 1; }
python3 - "$(find /tmp/hcp -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [("upsample_fwd (NCHW)", ["upsample_fwd_nchw"]), ("ce_fwd", ["ce_fwd_kernel"]), ("ce_bwd", ["ce_bwd_kernel"]),
         ("upsample_bwd (NCHW)", ["upsample_bwd_rows", "upsample_bwd_cols"])]
print("%-22s %12s %12s   (us per call, median; same operands again and again / three sets in rotation)" % ("", "same", "rotating"))
for label, keys in names:
    per_call = None
    for k in keys:
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if k in r["Kernel_Name"]]
        if k == "ce_fwd_kernel":
            d = d[1:]             # the call that made lse
        assert len(d) == 18, (k, len(d))
        per_call = d if per_call is None else [a + b for a, b in zip(per_call, d)]
    same, rot = sorted(per_call[3:9]), sorted(per_call[9:])
    print("%-22s %12.1f %12.1f" % (label, same[len(same) // 2], rot[len(rot) // 2]))
PY

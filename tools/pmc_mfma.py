#!/usr/bin/env python3
"""Reduce one rocprofv3 counter-collection CSV (tools/pmc_mfma.sh) to matrix-core utilisation per kernel family.

usage: tools/pmc_mfma.py <counter_collection.csv>

Per dispatch rocprofv3 reports each counter summed over the chip.  For a kernel family (segk kernels by function
name, MIOpen / CK convolution kernels lumped as "vendor_conv", the rest as "other"):
  mfma_busy_frac  = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES): the share of the cycles in which a CU
                    had work during which its matrix pipes were executing.  SQ_VALU_MFMA_BUSY_CYCLES counts the cycles
                    of every MFMA (16 per v_mfma_f32_16x16x32_bf16: flops_per_busy_cycle comes out at exactly 1024);
                    SQ_BUSY_CU_CYCLES is read as plain cycles summed over the CUs (read as quad-cycles it would leave
                    fewer than two resident wavefronts per CU against SQ_WAVE_CYCLES, which these kernels cannot have)
  mfma_util_chip  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs): against the whole chip for
                    the whole duration of the dispatch (idle CUs of a partial last round of workgroups count against it)
  flops_per_busy_cycle: SQ_INSTS_VALU_MFMA_MOPS_* x 512 FLOP / SQ_VALU_MFMA_BUSY_CYCLES (1024 = the dense bf16 rate of
                    one SIMD, 64 = the fp32 rate)
"""
import collections
import csv
import json
import re
import sys


def family(name):
    m = re.search(r"segk::(\w+)", name)
    if m:
        return m.group(1)
    if re.search(r"igemm|ck::|_ZN2ck|naive_conv|Conv|gemm", name):
        return "vendor_conv"
    return "other"


def main():
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for r in csv.DictReader(open(sys.argv[1])):
        f = family(r["Kernel_Name"])
        tot[f][r["Counter_Name"]] += float(r["Counter_Value"])
        n[f].add(r["Dispatch_Id"])
    out = {}
    for f, c in sorted(tot.items()):
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        if busy <= 0:
            continue
        cu_busy = c.get("SQ_BUSY_CU_CYCLES", 0.0)
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        mops = c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) + c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0)
        out[f] = {
            "launches": len(n[f]),
            "SQ_VALU_MFMA_BUSY_CYCLES": busy, "SQ_BUSY_CU_CYCLES": cu_busy, "GRBM_GUI_ACTIVE": gui,
            "SQ_INSTS_VALU_MFMA_MOPS_BF16": c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0),
            "SQ_INSTS_VALU_MFMA_MOPS_F32": c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0),
            "SQ_WAVE_CYCLES": c.get("SQ_WAVE_CYCLES", 0.0), "SQ_WAIT_INST_ANY": c.get("SQ_WAIT_INST_ANY", 0.0),
            "SQ_ACTIVE_INST_VALU": c.get("SQ_ACTIVE_INST_VALU", 0.0),
            "mfma_busy_frac": busy / (4.0 * cu_busy) if cu_busy > 0 else None,
            "wait_any_frac": c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None,
            "waves_per_busy_cu": 4.0 * c.get("SQ_WAVE_CYCLES", 0.0) / cu_busy if cu_busy > 0 else None,
            "mfma_util_chip": busy / (1024.0 * gui / 8.0) if gui > 0 else None,
            "flops_per_busy_cycle": 512.0 * mops / busy,
        }
    json.dump({"source": "rocprofv3 --pmc, one pass (tools/pmc_mfma.sh); bench.py --steps 2 --warmup 1", "kernels": out},
              sys.stdout, indent=1)


if __name__ == "__main__":
    main()

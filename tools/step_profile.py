#!/usr/bin/env python3
"""Summarise ONE steady-state training step from a rocprofv3 --kernel-trace CSV.

The process-wide --stats table also contains warm-up and MIOpen's one-time algorithm search, so this
script cuts the trace between the last two launches of a once-per-step kernel (ce_fwd_kernel) and
prints per-kernel totals for that interval: backward of step k-1 + optimizer + forward of step k.

usage: tools/step_profile.py <kernel_trace.csv> [--top N] [--hist] [--others]
--hist: for the convolution kernels, calls and time per duration class (which launches carry the time: the many
short ones of the 33 x 33 stages or the few long ones of the head).
--others: every kernel that is not this library's (vendor convolutions of the feeder, PyTorch glue), in launch order
with its grid, so that a launch can be matched to the layer it belongs to.
"""
import collections
import csv
import sys


def main():
    path = sys.argv[1]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "ce_fwd_kernel" in r["Kernel_Name"]]
    if len(marks) < 2:
        sys.exit("need at least two steps in the trace")
    seg = rows[marks[-2]:marks[-1]]
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
    agg = collections.defaultdict(lambda: [0, 0])
    busy_end, idle = None, 0
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        agg[r["Kernel_Name"]][0] += e - s
        agg[r["Kernel_Name"]][1] += 1
        if busy_end is not None and s > busy_end:
            idle += s - busy_end
        busy_end = max(busy_end or 0, e)
    tot = sum(v[0] for v in agg.values())
    ours = sum(v[0] for k, v in agg.items() if "segk::" in k)
    print("one steady-state step: wall %.3f ms, kernels %d, sum of kernel time %.3f ms, idle gaps %.3f ms"
          % ((t1 - t0) / 1e6, len(seg), tot / 1e6, idle / 1e6))
    print("segk (HIP head) kernels %.3f ms; everything else (feeder, optimizer, torch glue) %.3f ms"
          % (ours / 1e6, (tot - ours) / 1e6))
    print("%10s %6s %6s %10s  %s" % ("total_ms", "share", "calls", "avg_us", "kernel"))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
        print("%10.3f %5.1f%% %6d %10.1f  %s" % (v[0] / 1e6, 100.0 * v[0] / tot, v[1], v[0] / v[1] / 1e3, k[:130]))
    print("\nsegk kernels only:")
    for k, v in sorted(((k, v) for k, v in agg.items() if "segk::" in k), key=lambda kv: -kv[1][0]):
        print("%10.3f %6d %10.1f  %s" % (v[0] / 1e6, v[1], v[0] / v[1] / 1e3, k[:130]))
    if "--others" in sys.argv:
        print("\nkernels of other libraries, in launch order: start (ms into the step), duration (us), grid, name")
        for r in seg:
            if "segk::" in r["Kernel_Name"]:
                continue
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            grid = "x".join(r.get(k, "?") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
            print("%8.3f %8.1f  %-18s %s" % ((s - t0) / 1e6, (e - s) / 1e3, grid, r["Kernel_Name"][:100]))
    if "--hist" in sys.argv:
        edges = [0, 10, 15, 20, 30, 45, 70, 110, 200, 400, 10 ** 9]
        print("\nconvolution kernels by duration class (us): calls / total ms")
        per = collections.defaultdict(lambda: [[0, 0] for _ in edges[:-1]])
        for r in seg:
            k = r["Kernel_Name"]
            if "segk::conv_" not in k and "wgrad_finish" not in k:
                continue
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            for i in range(len(edges) - 1):
                if edges[i] <= d < edges[i + 1]:
                    per[k][i][0] += 1
                    per[k][i][1] += d
        for k, bins in sorted(per.items(), key=lambda kv: -sum(b[1] for b in kv[1])):
            print(k[:110])
            print("   " + "  ".join("%s-%s: %d / %.2f" % (edges[i], edges[i + 1] if edges[i + 1] < 10 ** 9 else "", b[0], b[1] / 1e3)
                                    for i, b in enumerate(bins) if b[0]))


if __name__ == "__main__":
    main()

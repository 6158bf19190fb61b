#!/bin/bash
# usage: tools/profile_step.sh <tag> [bench.py flags...]   (run on the GPU box from the repository root)
# Traces a short bench run with rocprofv3 and writes the one-step summary to gpurun_out/step_<tag>.txt;
# the raw trace stays in /tmp.
set -e -o pipefail
tag=$1; shift
repo=$(pwd)
mkdir -p "$repo/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o t --output-format csv -- \
  python3 "$repo/bench.py" --steps 3 --warmup 3 --no-cpu-baseline --no-fp32 --no-torch-baseline --no-kernel-timing "$@" > "$repo/gpurun_out/bench_prof_$tag.json"
trace=$(find /tmp/prof_$tag -name '*kernel_trace.csv' | sort | tail -1)
python3 "$repo/tools/step_profile.py" "$trace" --top 45 --hist --others > "$repo/gpurun_out/step_$tag.txt"
stats=$(find /tmp/prof_$tag -name '*kernel_stats.csv' | sort | tail -1)
cp "$stats" "$repo/gpurun_out/kernel_stats_$tag.csv"

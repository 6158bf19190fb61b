#!/bin/bash
# usage: tools/pmc_traffic.sh <tag> [bench.py flags...]      (GPU box, from the repository root)
# HBM-side traffic of the convolution kernels from the L2 fabric counters: two separate rocprofv3 counter
# passes (FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2) over a short bench run, then
# tools/pmc_traffic.py reduces the per-dispatch rows to per-launch averages -> gpurun_out/traffic_<tag>.json
set -e -o pipefail
tag=$1; shift
repo=$(pwd)
mkdir -p "$repo/gpurun_out"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_${tag}_$c
  timeout -s KILL 500 rocprofv3 --pmc $c -d /tmp/pmc_${tag}_$c -o p --output-format csv -- \
    python3 "$repo/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-fp32 --no-torch-baseline --no-kernel-timing --no-miopen-benchmark "$@" \
    > "$repo/gpurun_out/bench_pmc_${tag}_$c.json"
  echo "pass $c done"
done
python3 "$repo/tools/pmc_traffic.py" $(find /tmp/pmc_${tag}_FETCH_SIZE -name '*counter_collection.csv' | head -1) \
  $(find /tmp/pmc_${tag}_WRITE_SIZE -name '*counter_collection.csv' | head -1) > "$repo/gpurun_out/traffic_$tag.json"

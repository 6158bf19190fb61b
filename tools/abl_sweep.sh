#!/bin/bash
# usage: tools/abl_sweep.sh <case substring> <abl values...>     (GPU box, experiment build of the library)
# Kernel durations of one conv_bench case under each SEGK_DMA_ABL timing switch.
set -e -o pipefail
only=$1; shift
for a in "$@"; do
  echo "== SEGK_DMA_ABL=$a"
  SEGK_DMA_ABL=$a CONV_BENCH_NOSTATS=1 tools/prof_conv_bench.sh abl$a --only "$only"
done

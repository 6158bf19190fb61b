#!/bin/bash
# usage: tools/stage_sweep.sh <case substring> <"stages:abl" pairs...>   (GPU box, experiment build of the library)
set -e -o pipefail
only=$1; shift
for sa in "$@"; do
  s=${sa%%:*}; a=${sa##*:}
  echo "== SEGK_CONV_STAGES=$s SEGK_DMA_ABL=$a"
  SEGK_CONV_STAGES=$s SEGK_DMA_ABL=$a CONV_BENCH_NOSTATS=1 tools/prof_conv_bench.sh st${s}_$a --only "$only"
done

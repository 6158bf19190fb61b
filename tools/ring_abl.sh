#!/bin/bash
# usage: tools/ring_abl.sh <form> <case> <abl values...>   (GPU box, experiment build)
set -o pipefail
form=$1; only=$2; shift; shift
for a in "$@"; do
  echo "== form $form abl $a"
  SEGK_CONV_RING=$form SEGK_DMA_ABL=$a CONV_BENCH_NOSTATS=1 tools/prof_conv_bench.sh ra --only "$only" || exit 1
done

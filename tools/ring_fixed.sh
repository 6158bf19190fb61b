#!/bin/bash
# usage: tools/ring_fixed.sh <form> <case>   (GPU box, experiment build): the fixed cost of a ring-tile workgroup
# (timing switches of ConvDev::abl: 8 = no output stores, 16 = no K steps; second field: statistics off)
set -o pipefail
form=$1; only=$2
for cfg in "0:" "8:" "16:" "24:" "0:1" "8:1" "16:1" "24:1"; do
  a=${cfg%%:*}; ns=${cfg##*:}
  echo "== form $form abl $a nostats '$ns'"
  SEGK_CONV_RING=$form SEGK_DMA_ABL=$a CONV_BENCH_NOSTATS=$ns tools/prof_conv_bench.sh rf --only "$only" || exit 1
done

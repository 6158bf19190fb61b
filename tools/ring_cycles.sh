#!/bin/bash
# usage: tools/ring_cycles.sh <form> <case> <abl values...>   (GPU box, experiment build): GPU-busy cycles of the kernel
set -o pipefail
form=$1; only=$2; shift; shift
for a in "$@"; do
  echo "== form $form abl $a"
  SEGK_CONV_RING=$form SEGK_DMA_ABL=$a CONV_BENCH_NOSTATS=1 tools/pmc_any.sh "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "$only" || exit 1
done

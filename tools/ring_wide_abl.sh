#!/bin/bash
# usage: tools/ring_wide_abl.sh [abl values...]   (GPU box, experiment build): kernel durations of the decoder's 3x3
# convolution (256 -> 256) under the timing switches of the ring tile (SEGK_DMA_ABL bit mask, see ConvDev::abl).
set -e -o pipefail
mkdir -p gpurun_out
for abl in ${@:-0 1 2 128 129}; do
  echo "== SEGK_DMA_ABL=$abl (no statistics)"
  SEGK_DMA_ABL=$abl CONV_BENCH_NOSTATS=1 bash tools/prof_conv_bench.sh abl${abl} --only "dec 3x3 256" 2>&1 | grep -v "wgrad" || true
done

#!/bin/bash
# usage: tools/ring_abl_all.sh   (GPU box, experiment build): kernel durations of the forward ring tile on the
# decoder 3x3 (256 -> 256, 129 x 129) and the layer-3 3x3 (33 x 33) under the timing switches of SEGK_DMA_ABL
# (1 = loads without memory access, 2 = loads and barriers only, 128 = fragment reads without MFMAs, 8 = no output stores, 16 = no K loop)
set -e -o pipefail
mkdir -p gpurun_out
for c in "dec 3x3 256" "feeder l3"; do
  for a in 0 1 2 3 128 8 16; do
    echo "== case '$c' SEGK_DMA_ABL=$a"
    SEGK_DMA_ABL=$a CONV_BENCH_NOSTATS=1 bash tools/prof_conv_bench.sh abl --only "$c" | grep -v wgrad
  done
done

#!/bin/bash
# usage: tools/prio_try.sh   (GPU box, experiment build): kernel durations of tools/conv_bench.py with raised issue
# priority for the multiplying (SEGK_DMA_ABL=32) or the loading (64) wavefronts of the forward ring tile
set -e -o pipefail
mkdir -p gpurun_out
for a in 0 32 64; do
  echo "== SEGK_DMA_ABL=$a"
  SEGK_DMA_ABL=$a bash tools/prof_conv_bench.sh prio$a > gpurun_out/conv_kernels_prio$a.txt
  grep -v wgrad gpurun_out/conv_kernels_prio$a.txt | awk '{print $(NF-4), $(NF-3)}' | tr '\n' ' '
  echo
done

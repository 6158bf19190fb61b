#!/bin/bash
# usage: tools/ring_try.sh <forms...>   (GPU box, experiment build): for each form of the ring tile (SEGK_CONV_RING=1..5;
# 0 = the default kernels) the convolution tests, then kernel durations of tools/conv_bench.py
set -o pipefail
for r in "$@"; do
  echo "== SEGK_CONV_RING=$r"
  if [ "$r" != 0 ]; then
    SEGK_CONV_RING=$r timeout -k 10 600 python -u -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv or unit or dgrad" --timeout 120 --timeout-method thread > gpurun_out/ring_tests_$r.log 2>&1 || { tail -15 gpurun_out/ring_tests_$r.log; exit 1; }
    tail -1 gpurun_out/ring_tests_$r.log
  fi
  SEGK_CONV_RING=$r tools/prof_conv_bench.sh ring$r | grep -v wgrad || exit 1
done

#!/bin/bash
# usage: tools/ring_try2.sh   (GPU box, experiment build): the ring forms with four multiplying wavefronts - forward
# tile form 5 (64 x 80 per wavefront, SEGK_CONV_RING=5) and the weight-gradient tile with 128 x 64 per wavefront
# (SEGK_WGRAD_RING=4): their tests, then kernel durations / the sweep table.
set -e -o pipefail
mkdir -p gpurun_out
SEGK_CONV_RING=5 timeout -k 10 400 python -u -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv or unit or dgrad" --timeout 120 --timeout-method thread > gpurun_out/ring5_tests.log 2>&1 || { tail -15 gpurun_out/ring5_tests.log; exit 1; }
tail -1 gpurun_out/ring5_tests.log
SEGK_CONV_RING=5 bash tools/prof_conv_bench.sh ring5 > gpurun_out/conv_kernels_ring5.txt
grep -v wgrad gpurun_out/conv_kernels_ring5.txt
SEGK_WGRAD_RING=4 timeout -k 10 300 python -u -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "wgrad" --timeout 120 --timeout-method thread > gpurun_out/wring4_tests.log 2>&1 || { tail -15 gpurun_out/wring4_tests.log; exit 1; }
tail -1 gpurun_out/wring4_tests.log
SEGK_WGRAD_RING=4 timeout -k 10 200 python3 -u tools/wgrad_bench.py --parts 1 > gpurun_out/wgrad_bench_wring4.txt 2>&1
grep plan gpurun_out/wgrad_bench_wring4.txt

#!/bin/bash
# usage: tools/ring_wide_try.sh   (GPU box): convolution tests on the library as built, then rocprofv3 kernel durations
# of tools/conv_bench.py (the 256-wide ring tile is what the plan selects where output channels fill it).
set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv or unit or dgrad or gemm" --timeout 120 --timeout-method thread > gpurun_out/ringw_tests.log 2>&1 || { tail -25 gpurun_out/ringw_tests.log; exit 1; }
tail -1 gpurun_out/ringw_tests.log
bash tools/prof_conv_bench.sh ringw > gpurun_out/conv_kernels_ringw.txt
cat gpurun_out/conv_kernels_ringw.txt

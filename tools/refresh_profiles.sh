#!/bin/bash
# usage: tools/refresh_profiles.sh <tag>      (GPU box, from the repository root)
# Every measurement behind profiles/<round>_*: the step's kernel trace, the HBM-traffic and matrix-core counter passes,
# kernel durations of the convolution cases, the weight-gradient sweep and the per-unit comparison with PyTorch-ROCm.
# Each step runs under its own time limit; the first failure ends the script.
set -e -o pipefail
tag=$1
mkdir -p gpurun_out
bash tools/profile_step.sh "$tag"
echo "step trace done"
bash tools/pmc_traffic.sh "$tag" --feeder-conv torch --graph off
echo "traffic passes done"
bash tools/pmc_mfma.sh "$tag"
echo "mfma pass done"
bash tools/prof_conv_bench.sh "$tag" > "gpurun_out/conv_kernels_$tag.txt"
echo "conv kernel durations done"
timeout -k 10 200 python3 -u tools/wgrad_bench.py > "gpurun_out/wgrad_bench_$tag.txt" 2>&1
echo "wgrad sweep done"
timeout -k 10 300 python3 -u tools/unit_bench.py > "gpurun_out/unit_bench_$tag.txt" 2>&1
echo "unit bench done"

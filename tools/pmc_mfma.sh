#!/bin/bash
# usage: tools/pmc_mfma.sh <tag> [bench.py flags...]      (GPU box, from the repository root)
# Matrix-core utilisation of the convolution / GEMM kernels from the SQ counters: one rocprofv3 counter pass
# (7 SQ counters + 1 GRBM counter: within the 8 / 2 slots of a pass) over a short bench run, reduced per kernel
# family by tools/pmc_mfma.py -> gpurun_out/mfma_<tag>.json
set -e -o pipefail
tag=$1; shift
repo=$(pwd)
mkdir -p "$repo/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_mfma_$tag
timeout -s KILL 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 \
    SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
    -d /tmp/pmc_mfma_$tag -o p --output-format csv -- \
    python3 "$repo/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-fp32 --no-torch-baseline --no-kernel-timing --no-miopen-benchmark "$@" \
    > "$repo/gpurun_out/bench_pmc_mfma_$tag.json"
python3 "$repo/tools/pmc_mfma.py" $(find /tmp/pmc_mfma_$tag -name '*counter_collection.csv' | head -1) \
    > "$repo/gpurun_out/mfma_$tag.json"

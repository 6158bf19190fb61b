# usage: tools/quick_ab.sh <pytest -k expression or ""> [bench flags...]   (GPU box, repository root)
# Runs the selected GPU tests, then one short bench run, and prints value / ms per step.
set -e
mkdir -p gpurun_out
k=$1; shift
if [ -n "$k" ]; then
  timeout -k 10 400 python -u -m pytest -x -q --timeout 120 --timeout-method thread tests -m gpu -k "$k" > gpurun_out/t_quick.log 2>&1 || { tail -40 gpurun_out/t_quick.log; exit 1; }
  tail -2 gpurun_out/t_quick.log
fi
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32 --no-torch-baseline --no-kernel-timing "$@" > gpurun_out/b_quick.json 2> gpurun_out/b_quick.err || { tail -20 gpurun_out/b_quick.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/b_quick.json").read().strip().splitlines()[-1])
print("bench", d["value"], d["ms_per_step"], d["config"].get("final_loss"))
PY

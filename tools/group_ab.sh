set -e
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_gpu_ops.py -k "wgrad" -m gpu > gpurun_out/t_wgrad.log 2>&1 || { tail -30 gpurun_out/t_wgrad.log; exit 1; }
tail -2 gpurun_out/t_wgrad.log
timeout -k 10 300 python -u -m pytest -x -q --timeout 120 --timeout-method thread tests/test_feeder_conv.py tests/test_graph.py -m gpu > gpurun_out/t_feed.log 2>&1 || { tail -40 gpurun_out/t_feed.log; exit 1; }
tail -2 gpurun_out/t_feed.log
SEGK_GROUP_WGRAD=0 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32 --no-torch-baseline --no-kernel-timing > gpurun_out/b_nogroup.json 2> gpurun_out/b_nogroup.err || { tail -20 gpurun_out/b_nogroup.err; exit 1; }
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32 --no-torch-baseline --no-kernel-timing > gpurun_out/b_group.json 2> gpurun_out/b_group.err || { tail -20 gpurun_out/b_group.err; exit 1; }
SEGK_GROUP_WGRAD=0 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32 --no-torch-baseline --no-kernel-timing > gpurun_out/b_nogroup2.json 2> gpurun_out/b_nogroup2.err
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32 --no-torch-baseline --no-kernel-timing > gpurun_out/b_group2.json 2> gpurun_out/b_group2.err
python - <<'PY'
import json
for n in ("b_nogroup","b_group","b_nogroup2","b_group2"):
    d=json.loads(open("gpurun_out/%s.json"%n).read().strip().splitlines()[-1])
    print(n, d["value"], d["ms_per_step"])
PY

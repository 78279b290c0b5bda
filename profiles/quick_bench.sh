#!/bin/bash
# Short bench run (no CPU baseline, no secondary blocks) + the slowest kernels of the step; optional pytest files after it.
#   bash profiles/quick_bench.sh [test files...]
set -e
mkdir -p gpurun_out/qb
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-secondary > gpurun_out/qb/b.json
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/qb/b.json").read().strip().splitlines()[-1])
print("ms_per_step %.4f  audio-s/s %.0f  chain hbm_frac %.4f" % (d["ms_per_step"], d["value"], d["chain"]["hbm_frac"]))
print({k: v["ms"] for k, v in d["stages"].items()})
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])[:8]:
    print("  %-28s %.4f ms/step (%g launches)" % (k, v["ms_per_step"], v["launches_per_step"]))
PY
if [ $# -gt 0 ]; then
    timeout -k 10 600 python -u -m pytest "$@" -x -q --timeout 120 --timeout-method thread 2>&1 | tail -n 4
fi

#!/bin/bash
# fp64 chain (32 clips per step): per-kernel times
set -e
mkdir -p gpurun_out/qb
timeout -k 10 300 python3 bench.py --dtype f64 --clips 32 --no-cpu-baseline --no-secondary > gpurun_out/qb/b64.json
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/qb/b64.json").read().strip().splitlines()[-1])
print("ms_per_step %.4f  audio-s/s %.0f" % (d["ms_per_step"], d["value"]))
print({k: v["ms"] for k, v in d["stages"].items()})
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])[:10]:
    print("  %-36s %.4f ms/step (%g launches)" % (k[:36], v["ms_per_step"], v["launches_per_step"]))
PY

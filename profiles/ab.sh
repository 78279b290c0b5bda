#!/bin/bash
# A/B of two builds of the library on ONE box (boxes of the pool differ by a few percent):
#   ab/libadn_base.so and ab/libadn_new.so, three alternating rounds of the short bench each.
set -e
mkdir -p gpurun_out/ab
L=audio-denoising_amd/adn_hip/libadn_hip.so
for r in 1 2 3; do
  for v in base new; do
    cp ab/libadn_$v.so $L
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-secondary > gpurun_out/ab/$v$r.json
  done
done
cp ab/libadn_new.so $L
python3 - <<'PY'
import json
for v in ("base", "new"):
    rows = [json.loads(open("gpurun_out/ab/%s%d.json" % (v, r)).read().strip().splitlines()[-1]) for r in (1, 2, 3)]
    print(v, "ms/step", [round(d["ms_per_step"], 3) for d in rows],
          {k: [round(d["kernels"][k]["avg_ms"], 4) for d in rows] for k in list(rows[0]["kernels"])[:4]})
PY

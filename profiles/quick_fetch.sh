#!/bin/bash
# One FETCH_SIZE pass of a short bench run; prints the corrected bytes per launch of the STFT kernels.
set -e
OUT=$PWD/gpurun_out/qf
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o f -- python3 $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null
cd $OLDPWD
python3 - <<'PY'
import csv, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open("gpurun_out/qf/fetch/f_counter_collection.csv")):
    if r["Counter_Name"] == "FETCH_SIZE":
        agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items(), key=lambda kv: -max(kv[1]))[:8]:
    print("%-70s %.3f GB" % (k[-70:], 2 * max(v) * 1024 / 1e9))
PY

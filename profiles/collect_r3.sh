#!/bin/bash
# Measurement set behind the committed profiles/r3_* files (run on the GPU box from the repo root):
#   1. default bench line (with its secondary blocks and the CPU baseline),
#   2. rocprofv3 kernel-trace stats of the same command,
#   3./4. PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass) reduced by pmc_traffic.py,
#   5. a PMC pass of eight SQ counters reduced by pmc_sq.py,
#   6. the mono wavelet shape at BASELINE config 3's full size (1024 x 30 s, db8 level 8): its own line and kernel stats,
#   7. the shape sweep (clip lengths 10 ... 1900 s at the same audio per step).
set -e
export ADN_COMMIT=${1:-unknown}      # commit of the tree the files were collected from (the GPU box has no .git)
OUT=$PWD/gpurun_out/r3
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 400 python3 bench.py > $OUT/bench_default.json
cd /tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $OLDPWD/bench.py --no-cpu-baseline --no-secondary > $OUT/bench_under_rocprof.json
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o f -- python3 $OLDPWD/bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-secondary > /dev/null
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o w -- python3 $OLDPWD/bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-secondary > /dev/null
timeout -s KILL 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/sq -o s -- python3 $OLDPWD/bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-secondary > /dev/null
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3stats -o c3 -- python3 $OLDPWD/profiles/bench_c3.py > $OUT/c3_bench.json
cd $OLDPWD
python3 profiles/pmc_sq.py $OUT/sq/s_counter_collection.csv $OUT/pmc_sq.json > /dev/null
python3 profiles/pmc_traffic.py $OUT/fetch/f_counter_collection.csv $OUT/write/w_counter_collection.csv 64 $OUT/pmc_traffic.json > /dev/null
timeout -k 10 600 python3 profiles/shape_sweep.py 10 20 30 45 60 61.3 70 85 90 120 180 240 360 480 600 900 1200 1900 > $OUT/shape_sweep.txt 2>/dev/null
echo collected

#!/usr/bin/env python
"""CPU baseline leg of bench.py (TEST / MEASUREMENT INFRASTRUCTURE, never on the product path).

Times the numpy restatement of the chain (oracle/adn_oracle.py: phase 1 noise profile, phase 2
wavelet blocks, phase 2b spectral subtraction, phase 2c Wiener) on whole synthetic clips of the
benchmark workload (60 s, 48 kHz, stereo + 5 s noise sample):
  (i)  one thread, clips one after another until the time budget is spent;
  (ii) all host cores of this process's affinity set (capped by --max-procs), one clip chain per
       worker process at a time (file-level parallelism: clips are independent, like the files of
       the reference).
Prints ONE JSON object.  Runs as its own process so that no worker is forked from a process that
holds a GPU context.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

SR, CLIP_S, NOISE_S = 48000, 60, 5


def _one_clip(stream):
    import numpy as np
    import adn_oracle as orc
    import synth
    n, nn = CLIP_S * SR, NOISE_S * SR
    x = synth.stereo_noisy_speech(n, SR, 42, stream)
    noise = synth.white_noise(nn, 42, 100000 + stream)
    t0 = time.perf_counter()
    base, _ = orc.noise_profile(noise)
    y2, _, _ = orc.denoise_signal(x, SR)
    y2b = orc.spectral_subtract(y2, base.astype(np.float32).astype(np.float64))
    orc.wiener(y2b, SR)
    return time.perf_counter() - t0


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=10.0, help="time budget of each leg")
    ap.add_argument("--max-procs", type=int, default=16)
    args = ap.parse_args()
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[var] = "1"
    affinity = len(os.sched_getaffinity(0))
    procs = max(1, min(affinity, args.max_procs))

    # (i) one thread
    done, t0 = 0, time.perf_counter()
    while True:
        _one_clip(done)
        done += 1
        el1 = time.perf_counter() - t0
        if el1 >= args.seconds or done >= 16:
            break
    single = done * CLIP_S / el1
    per_clip = el1 / done

    # (ii) all cores: enough clips that every worker gets the same number of whole clips
    rounds = max(1, int(round(args.seconds / per_clip)))
    n_clips = procs * rounds
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    with ctx.Pool(procs) as pool:
        pool.map(_one_clip, range(n_clips), chunksize=1)
    el2 = time.perf_counter() - t0
    multi = n_clips * CLIP_S / el2

    print(json.dumps({
        "value": multi, "unit": "audio-sec/sec", "cores": procs, "kind": "port",
        "sample": "%d x (60 s 48 kHz stereo clip + 5 s noise) over %d worker processes, fp64 numpy "
                  "restatement of phases 1/2/2b/2c, %.1f s of wall time" % (n_clips, procs, el2),
        "single_thread": {"value": single, "cores": 1,
                          "sample": "%d clips one after another, %.1f s" % (done, el1)},
        "os_cpu_count": os.cpu_count(), "affinity_cpus": affinity, "cpu_model": cpu_model(),
    }))


if __name__ == "__main__":
    main()

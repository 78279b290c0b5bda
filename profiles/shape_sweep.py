"""The chain of bench.py (phases 1 / 2 / 2b / 2c, fp32, stereo clips + 5 s noise) on other batch shapes than the
benchmark's 64 x 60 s: clip lengths 10 ... 120 s at about the same total audio per step.  Prints per shape the
throughput and the stage times per 3840 audio-seconds (the benchmark's step), so that cliffs of the launch
heuristics show.    python profiles/shape_sweep.py [clip_seconds ...]"""
import os, sys, time
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "audio-denoising_amd"))
import bench
from adn_hip import get_engine
e = get_engine(0)
for clip_s in [float(a) for a in sys.argv[1:]] or [10, 20, 30, 45, 60, 61.3, 90, 120]:
    bench.CLIP_S = clip_s
    clips = max(1, int(round(3840 / clip_s)))
    wl = bench.Workload(e, clips, torch.float32, 0)
    wl.step(); wl.step()
    ev = {}
    for _ in range(3):
        wl.step(ev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        wl.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 8
    audio = clips * clip_s
    ms = {k: float(np.mean([a.elapsed_time(b) for a, b in v])) * 3840 / audio for k, v in ev.items()}
    print("clip %6.1f s x %4d: %7.0f audio-s/s   per 3840 audio-s: %s" % (clip_s, clips, audio / dt, {k: round(v, 3) for k, v in ms.items()}), flush=True)
    if os.environ.get("ADN_SWEEP_KERNELS"):
        e.kernel_timing(True)
        for _ in range(3):
            wl.step()
        kt = e.kernel_times()
        e.kernel_timing(False)
        top = sorted(kt.items(), key=lambda kv: -kv[1]["total_ms"])[:12]
        print("      " + ", ".join("%s %.3f" % (k[:22], v["total_ms"] / 3 * 3840 / audio) for k, v in top), flush=True)
    del wl
    torch.cuda.empty_cache()

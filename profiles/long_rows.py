"""The bench chain on clips of 600 s (6 clips per step, the same audio per step as the default bench): the shape whose
power rows (56 247 frames) take the streaming row select.  Run under `rocprofv3 --kernel-trace --stats` for the
per-kernel table in profiles/r3_long_rows_kernel_stats.csv.    python profiles/long_rows.py [clip_seconds [steps]]"""
import os, sys
import torch
REPO = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "audio-denoising_amd"))
import bench
from adn_hip import get_engine
e = get_engine(0)
clip_s = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
bench.CLIP_S = clip_s
wl = bench.Workload(e, max(1, int(round(3840 / clip_s))), torch.float32, 0)
for _ in range(steps):
    wl.step()
torch.cuda.synchronize()

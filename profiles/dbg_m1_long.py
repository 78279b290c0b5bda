"""Diagnostics: the stereo wavelet stage on long blocks with the per-block dump of the single-pass statistics
(option bit 2048).    python profiles/dbg_m1_long.py [clip_seconds] [clips]"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "audio-denoising_amd"))
import bench
from adn_hip import get_engine
e = get_engine(0)
bench.CLIP_S = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
clips = int(sys.argv[2]) if len(sys.argv) > 2 else 2
wl = bench.Workload(e, clips, torch.float32, 0)
e.lib.adn_set_option(e._h, 2, 2048)
wl.phase2()
torch.cuda.synchronize()
e.lib.adn_set_option(e._h, 2, 0)

"""Diagnostics: the Wiener stage's row statistics with and without the bit-by-bit search (option ADN_OPT_ABLATE,
bit 32: the search loop is skipped, results are wrong), to separate load time from search time."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "audio-denoising_amd"))
from adn_hip import get_engine
e = get_engine(0)
SR = 48000
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0          # clip length; about 3840 s of audio per call
B, n = max(1, int(round(3840 / secs))), int(secs * SR)
x = torch.randn(B * n, device="cuda", dtype=torch.float32) * 0.1
for ab in (0, 32):
    e.lib.adn_set_option(e._h, 2, ab)
    e.wiener(x, [n] * B, SR)
    e.kernel_timing(True)
    for _ in range(3):
        e.wiener(x, [n] * B, SR)
    kt = e.kernel_times()
    e.kernel_timing(False)
    print("clip %.0f s x %d, ablate" % (secs, B), ab, {k: round(v["total_ms"] / v["count"], 4) for k, v in kt.items() if "select" in k or "rows" in k})
e.lib.adn_set_option(e._h, 2, 0)

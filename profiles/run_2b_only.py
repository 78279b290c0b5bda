"""Runs only the fused phase-2b kernel (and optionally the Wiener stage) a few times: target
for rocprofv3 --pmc passes."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "audio-denoising_amd"))
from adn_hip import get_engine
e = get_engine(0)
B, n, SR = 32, 60 * 48000, 48000
x = torch.randn(B * n, device="cuda", dtype=torch.float32) * 0.1
noise = torch.rand(1025, device="cuda", dtype=torch.float32) * 0.01
which = sys.argv[1] if len(sys.argv) > 1 else "2b"
for it in range(3):
    if which == "2b":
        e.spectral_subtract(x, [n] * B, noise)
    else:
        e.wiener(x, [n] * B, SR)
torch.cuda.synchronize()
print("done")

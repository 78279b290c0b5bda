import os, sys, torch
sys.path.insert(0, "/root/repo/audio-denoising_amd")
from adn_hip import get_engine
e = get_engine(0)
B, n, SR = 64, 60 * 48000, 48000
x = torch.randn(B * n, device="cuda", dtype=torch.float32) * 0.1
noise = torch.rand(1025, device="cuda", dtype=torch.float32) * 0.01
e.spectral_subtract(x, [n] * B, noise); e.wiener(x, [n] * B, SR)
e.lib.adn_set_option(e._h, 2, 16)
e.spectral_subtract(x, [n] * B, noise)
e.wiener(x, [n] * B, SR)
e.lib.adn_set_option(e._h, 2, 0)
torch.cuda.synchronize()

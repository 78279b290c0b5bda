"""Diagnostics: the same comparison on the numpy-generated test signal, with the per-frame VAD energies' margin."""
import os, sys, torch, numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "audio-denoising_amd")); sys.path.insert(0, os.path.join(R, "oracle")); sys.path.insert(0, os.path.join(R, "tests"))
from adn_hip import get_engine
import adn_oracle as orc, synth
e = get_engine(0)
SR = 48000
n = 7 * SR + 123
xh = synth.noisy_speech(n, SR, 42, 930)
for dt in (torch.float32, torch.float64):
    x = torch.as_tensor(xh, device="cuda").to(dt)
    out = {}
    for bits in (0, 128):
        e.lib.adn_set_option(e._h, 2, bits)
        y, l, g = e.wiener(x, [n], SR, want_gain=True)
        e.lib.adn_set_option(e._h, 2, 0)
        out[bits] = (y.double().cpu().numpy().copy(), g[0].double().cpu().numpy().copy())
    yo, go = orc.wiener(x.double().cpu().numpy(), SR)
    print(dt, "y fold-apart", np.abs(out[0][0] - out[128][0]).max(), "gain fold-apart", np.abs(out[0][1] - out[128][1]).max(),
          "gain fold-oracle", np.abs(out[0][1] - go).max(), "gain apart-oracle", np.abs(out[128][1] - go).max())
    xx = x.double().cpu().numpy(); fl = 960
    nv = -(-n // fl)
    db = np.array([20 * np.log10(np.sqrt(np.mean(xx[i * fl:(i + 1) * fl] ** 2) + 1e-12) + 1e-12) for i in range(nv)])
    print("   VAD frames closest to -40 dB:", np.sort(np.abs(db + 40))[:4])

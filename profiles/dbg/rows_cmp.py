"""Diagnostics: gain of the Wiener stage with the grouped-rows path and with the transposed-rows path."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "audio-denoising_amd"))
from adn_hip import get_engine
e = get_engine(0)
SR = 16000
n = 24 * SR + 77
x = e.synth_noisy_speech(n, SR, stream=920).float()
for vg in (True, False):
    y1, l1, g1 = e.wiener(x, [n], SR, want_gain=True, vad_guided=vg)
    e.lib.adn_set_option(e._h, 2, 64)
    y0, l0, g0 = e.wiener(x, [n], SR, want_gain=True, vad_guided=vg)
    e.lib.adn_set_option(e._h, 2, 0)
    d = (g1 - g0).abs()
    print("vad", vg, "max gain diff", float(d.max()), "bins differing", int((d > 0).sum()), "first", torch.nonzero(d > 0)[:5].flatten().tolist())
S, fr = e.stft_rows(x, [n])
print("frames", fr)
# determinism of each path
for ab in (0, 64):
    e.lib.adn_set_option(e._h, 2, ab)
    gs = [e.wiener(x, [n], SR, want_gain=True, vad_guided=False)[2].clone() for _ in range(3)]
    print("ablate", ab, "repeat diffs", [int(((g - gs[0]).abs() > 0).sum()) for g in gs[1:]])
e.lib.adn_set_option(e._h, 2, 0)
# which frames differ: the P10 of a signal whose frames beyond a cut are silenced
for cut in (100, 400, 700, 740, 746):
    xc = x.clone(); xc[cut * 512 + 2048:] = 0
    a = e.wiener(xc, [n], SR, want_gain=True, vad_guided=False)[2]
    e.lib.adn_set_option(e._h, 2, 64)
    b = e.wiener(xc, [n], SR, want_gain=True, vad_guided=False)[2]
    e.lib.adn_set_option(e._h, 2, 0)
    print("cut", cut, "bins differing", int(((a - b).abs() > 0).sum()))

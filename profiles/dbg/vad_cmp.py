"""Diagnostics: Wiener gain with the voice-activity energies from the rows kernel vs from the separate pass."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "audio-denoising_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
from adn_hip import get_engine
import adn_oracle as orc
e = get_engine(0)
SR = 48000
for n in (7 * SR + 123, 7 * SR, 5 * SR + 77, 336000, 336123):
    x = e.synth_noisy_speech(n, SR, stream=930).float()
    out = {}
    for bits in (0, 128):
        e.lib.adn_set_option(e._h, 2, bits)
        y, l, g = e.wiener(x, [n], SR, want_gain=True)
        e.lib.adn_set_option(e._h, 2, 0)
        out[bits] = g[0].double().cpu().numpy()
    _, g_or = orc.wiener(x.double().cpu().numpy(), SR)
    print(n, "fold vs apart", np.abs(out[0] - out[128]).max(), "fold vs oracle", np.abs(out[0] - g_or).max(), "apart vs oracle", np.abs(out[128] - g_or).max())
for n1, n2 in ((7 * SR + 123, 5 * SR + 77), (7 * SR, 5 * SR + 77), (7 * SR + 124, 5 * SR)):
    x = e.synth_noisy_speech(n1, SR, stream=930).float()
    xs = torch.cat([x, x[:n2]])
    out = {}
    for bits in (0, 128):
        e.lib.adn_set_option(e._h, 2, bits)
        y, l, g = e.wiener(xs, [n1, n2], SR, want_gain=True)
        e.lib.adn_set_option(e._h, 2, 0)
        out[bits] = g.double().cpu().numpy()
    g1 = orc.wiener(xs[:n1].double().cpu().numpy(), SR)[1]
    g2 = orc.wiener(xs[n1:].double().cpu().numpy(), SR)[1]
    print((n1, n2), "fold vs apart per signal", np.abs(out[0] - out[128]).max(axis=1), "fold vs oracle", np.abs(out[0][0] - g1).max(), np.abs(out[0][1] - g2).max(),
          "apart vs oracle", np.abs(out[128][0] - g1).max(), np.abs(out[128][1] - g2).max())

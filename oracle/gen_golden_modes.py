#!/opt/conda/bin/python3.9
"""Golden vectors for the denoiser's optional modes (TEST INFRASTRUCTURE; build container only):

    /opt/conda/bin/python3.9 -W ignore oracle/gen_golden_modes.py

The block arithmetic of denoise.py:77-89 of the reference (wavedec -> mad of the finest band ->
universal threshold -> threshold on every detail band -> waverec), evaluated with PyWavelets
itself, for the two knobs the reference leaves at their defaults:
  * mode='symmetric' instead of 'per' (the extension lib/windowBundle.py:33 uses);
  * a (frames, 2) block transformed along TIME (axis=0 per channel, every channel with its own
    statistics) instead of pywt's default last axis.
Only data is written: inputs, outputs, parameters, library versions.
"""
import json
import os
import sys

import numpy as np
import pywt

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import synth  # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")


def mad(a):
    a = np.ma.array(a).compressed()
    return np.median(np.abs(a - np.median(a)))


def block(x, wavelet, level, mode, thr_mode):
    co = pywt.wavedec(x, wavelet, mode=mode, level=level)
    sigma = mad(co[-1])
    t = sigma * np.sqrt(2 * np.log(len(x)))
    co[1:] = (pywt.threshold(c, value=t, mode=thr_mode) for c in co[1:])
    return pywt.waverec(co, wavelet, mode=mode), sigma, t


def main():
    out, meta = {}, []
    cases = [(4001, "db4", 2, "symmetric", "soft"), (4000, "db8", 3, "symmetric", "soft"),
             (1237, "db8", 4, "symmetric", "hard"), (37, "db4", 1, "symmetric", "soft")]
    for i, (n, w, L, mode, thr) in enumerate(cases):
        x = synth.noisy_speech(n, 16000, 42, 300 + i)
        y, s, t = block(x, w, L, mode, thr)
        out["sym%d_x" % i], out["sym%d_y" % i] = x, y
        meta.append({"n": n, "wavelet": w, "level": L, "mode": mode, "thr": thr, "sigma": float(s), "thresh": float(t)})
    out["sym_meta_json"] = np.array(json.dumps(meta))
    tmeta = []
    for i, (n, w, L) in enumerate([(16001, "db4", 2), (9000, "db8", 8)]):
        x = synth.stereo_noisy_speech(n, 16000, 42, 400 + i)
        ys, sg, th = [], [], []
        for ch in range(2):
            y, s, t = block(np.ascontiguousarray(x[:, ch]), w, L, "per", "soft")
            ys.append(y)
            sg.append(float(s))
            th.append(float(t))
        out["time%d_x" % i], out["time%d_y" % i] = x, np.stack(ys, axis=1)
        tmeta.append({"n": n, "wavelet": w, "level": L, "sigma": sg, "thresh": th})
    out["time_meta_json"] = np.array(json.dumps(tmeta))
    out["versions_json"] = np.array(json.dumps({"pywt": pywt.__version__, "numpy": np.__version__}))
    path = os.path.join(GOLD, "denoise_modes.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()

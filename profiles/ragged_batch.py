"""Spectral subtraction and Wiener stage on a batch of mono files of DIFFERENT lengths (uniform in 20 ... 60 s) against a
batch of equal files with the same total audio (the one-wave-per-run kernels size their runs on the device from all
lengths, k_apply_wg_map) - and with files whose start offsets are 16-byte aligned
against files packed back to back (the kernels' 16-byte loads and 8-byte stores need aligned starts)."""
import os, sys, time
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "audio-denoising_amd"))
from adn_hip import get_engine
e = get_engine(0)
SR = 48000
rng = np.random.default_rng(5)
src = torch.empty(60 * SR, dtype=torch.float64, device="cuda")
e.synth_noisy_speech(60 * SR, SR, seed=42, stream=1, out=src)
src = src.float()
noise = torch.rand(1025, device="cuda") * 0.01


def starts(lens, align):
    """Element offsets of files packed one after another, every start rounded up to `align` elements."""
    off, pos = [], 0
    for n in lens:
        pos = -(-pos // align) * align
        off.append(pos)
        pos += n
    return np.array(off, dtype=np.int64), pos


def run(lens, label, align=1):
    lens = [int(n) for n in lens]
    off, total = starts(lens, align)
    x = torch.zeros(total, dtype=torch.float32, device="cuda")
    for o, n in zip(off, lens):
        x[o:o + n] = src[:n]
    out_lens = [e.istft_out_samples(n, 2048, 512) for n in lens]
    yoff, ytotal = starts(out_lens, align)
    y = torch.empty(ytotal, dtype=torch.float32, device="cuda")
    out_lens2 = [e.istft_out_samples(n, 2048, 512) for n in out_lens]
    yoff2, y2total = starts(out_lens2, align)
    y2 = torch.empty(y2total, dtype=torch.float32, device="cuda")

    def step():
        e.spectral_subtract(x, lens, noise, offsets=off, out=y, out_offsets=yoff)
        e.wiener(y, out_lens, SR, offsets=yoff, out=y2, out_offsets=yoff2)
    step(); step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 8
    audio = sum(lens) / SR
    print("%-28s %3d files %6.0f audio-s: %.3f ms = %.0f K audio-s/s" % (label, len(lens), audio, dt * 1e3, audio / dt / 1e3), flush=True)
    e.kernel_timing(True)
    for _ in range(3):
        step()
    kt = e.kernel_times()
    e.kernel_timing(False)
    print("      " + ", ".join("%s %.3f" % (k[:24], v["total_ms"] / 3) for k, v in sorted(kt.items(), key=lambda kv: -kv[1]["total_ms"])[:7]), flush=True)


ragged = rng.integers(20 * SR, 60 * SR, size=96)
run(ragged, "lengths 20 ... 60 s, packed")
run(ragged, "the same, starts 16-B aligned", align=4)
run([int(np.mean(ragged))] * 96, "equal files (mean), packed")
run([int(np.mean(ragged))] * 96, "the same, starts aligned", align=4)
run([60 * SR] * 64, "64 x 60 s")

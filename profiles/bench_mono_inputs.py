"""Mono wavelet stage (db4 level 2 by default) on inputs that stress the threshold statistics: Gaussian noise,
PCM16-quantised low-level noise (heavy ties), digital silence, half-silent blocks and autocorrelated material
(first-order autoregressive noise, a random walk, the synthetic speech of the other benchmarks).  256 clips of
30 s at 48 kHz in 10 blocks each, fp32 (ADN_BENCH_DTYPE=fp64 changes that).  One JSON line per input: the time
with the statistics in one pass over the samples and with the two-pass route (finest band in memory, two
selections)."""
import json, os, sys, time
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "audio-denoising_amd"))
from adn_hip import get_engine
e = get_engine(0)
DT = torch.float64 if os.environ.get("ADN_BENCH_DTYPE", "fp32") == "fp64" else torch.float32
WAVELET, LEVEL = os.environ.get("ADN_BENCH_WAVELET", "db4"), int(os.environ.get("ADN_BENCH_LEVEL", "2"))
SR, n, clips = 48000, 30 * 48000, int(os.environ.get("ADN_BENCH_CLIPS", "256"))
bs = e.denoise_block_frames(SR, n)
starts = np.arange(0, n, bs, dtype=np.int64)
off = np.concatenate([c * n + starts for c in range(clips)])
lens = np.concatenate([np.minimum(bs, n - starts) for c in range(clips)])
d_off, d_len = e._i64(off), e._i64(lens)
g = torch.Generator(device="cuda").manual_seed(3)


def make(kind):
    x = torch.randn(clips * n, device="cuda", generator=g) * 0.1
    if kind == "pcm_quiet":
        x = torch.round(x * 20.0) / 32768.0
    elif kind == "silence":
        x.zero_()
    elif kind == "half_silent":
        x.view(clips, n)[:, : n // 2] = 0.0
    elif kind == "lowpass":
        x = torch.cumsum(x.view(clips, n), dim=1).reshape(-1) * 0.01
    elif kind == "speech":
        tmp = torch.empty(n, dtype=torch.float64, device="cuda")
        for c in range(16):
            e.synth_noisy_speech(n, SR, stream=c, out=tmp)
            x[c * n:(c + 1) * n] = tmp.float()
        for c in range(16, clips):
            x[c * n:(c + 1) * n] = x[(c % 16) * n:(c % 16 + 1) * n]
    elif kind.startswith("ar"):
        rho = float(kind[2:])
        k = torch.pow(torch.tensor(rho, device="cuda"), torch.arange(255, -1, -1, device="cuda")).view(1, 1, -1)
        xt = torch.nn.functional.conv1d(torch.nn.functional.pad(x.view(clips, 1, n), (255, 0)), k)
        x = (xt * (1.0 - rho * rho) ** 0.5).reshape(-1)
    return x.reshape(-1).contiguous().to(DT)


kinds = sys.argv[1:] or ["gauss", "speech", "pcm_quiet", "silence", "half_silent", "ar0.9", "ar0.99", "lowpass"]
for kind in kinds:
    x = make(kind)
    y = torch.empty_like(x)
    res = {"input": kind, "dtype": str(DT).split(".")[-1], "wavelet": WAVELET, "level": LEVEL}
    for label, ablate in (("one_pass_ms", 0), ("two_pass_ms", 512)):
        e._check(e.lib.adn_set_option(e._h, 2, ablate))
        def run():
            e.wavelet_denoise(x, d_off, d_len, y, d_off, n_ch=1, wavelet=WAVELET, level=LEVEL, max_frames=int(lens.max()))
        run(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            run()
        torch.cuda.synchronize()
        res[label] = round((time.perf_counter() - t0) / 4 * 1e3, 3)
    e._check(e.lib.adn_set_option(e._h, 2, 0))
    print(json.dumps(res), flush=True)

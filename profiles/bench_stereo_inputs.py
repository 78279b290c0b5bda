"""Phase 2 (stereo, db4 level 2) on inputs that stress the threshold statistics: Gaussian noise,
PCM16-quantised low-level noise (+-2 LSB: heavy ties), digital silence, half-silent blocks and
autocorrelated material (first-order autoregressive noise, a random walk).  64 clips of 60 s at 48 kHz, fp32 (ADN_BENCH_DTYPE=fp64, ADN_BENCH_SECONDS=... change that).  Prints one JSON line per
input with the time of the single-pass route and of the plain route (coefficient array + two full
selections)."""
import json, os, sys, time
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "audio-denoising_amd"))
from adn_hip import get_engine
e = get_engine(0)
DT = torch.float64 if os.environ.get("ADN_BENCH_DTYPE", "fp32") == "fp64" else torch.float32
SR, n, clips = 48000, int(os.environ.get("ADN_BENCH_SECONDS", "60")) * 48000, int(os.environ.get("ADN_BENCH_CLIPS", "64"))
bs = e.denoise_block_frames(SR, n)
starts = np.arange(0, n, bs, dtype=np.int64)
off = np.concatenate([c * n + starts for c in range(clips)]) * 2
lens = np.concatenate([np.minimum(bs, n - starts) for c in range(clips)])
d_off, d_len = e._i64(off), e._i64(lens)
g = torch.Generator(device="cuda").manual_seed(3)


def make(kind):
    x = torch.randn(clips * n, 2, device="cuda", generator=g) * 0.1
    if kind == "pcm_quiet":
        x = torch.round(x * 20.0) / 32768.0
    elif kind == "silence":
        x.zero_()
    elif kind == "half_silent":
        x.view(clips, n, 2)[:, : n // 2] = 0.0
    elif kind == "lowpass":
        x = torch.cumsum(x.view(clips, n, 2), dim=1).reshape(-1, 2) * 0.01
    elif kind.startswith("ar"):
        # first-order autoregressive noise, coefficient rho (truncated impulse response)
        rho = float(kind[2:])
        k = torch.pow(torch.tensor(rho, device="cuda"), torch.arange(255, -1, -1, device="cuda")).view(1, 1, -1)
        xt = x.view(clips, n, 2).permute(0, 2, 1).reshape(clips * 2, 1, n)
        xt = torch.nn.functional.conv1d(torch.nn.functional.pad(xt, (255, 0)), k)
        x = (xt.view(clips, 2, n).permute(0, 2, 1) * (1.0 - rho * rho) ** 0.5).reshape(-1, 2)
    return x.reshape(-1).contiguous().to(DT)


kinds = sys.argv[1:] or ["gauss", "pcm_quiet", "silence", "half_silent", "ar0.9", "ar0.99", "lowpass"]
for kind in kinds:
    x = make(kind)
    y = torch.empty_like(x)
    res = {"input": kind, "dtype": str(DT).split(".")[-1], "seconds": n // SR}
    for label, generic in (("single_pass_ms", False), ("plain_ms", True)):
        e.force_generic(generic)
        def run():
            e.wavelet_denoise(x, d_off, d_len, y, d_off, n_ch=2, wavelet="db4", level=2, max_frames=int(lens.max()))
        run(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            run()
        torch.cuda.synchronize()
        res[label] = round((time.perf_counter() - t0) / 4 * 1e3, 3)
    e.force_generic(False)
    print(json.dumps(res), flush=True)

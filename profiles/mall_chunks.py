"""Does the stereo wavelet stage gain from walking the batch in pieces small enough for the 256 MB memory-side cache
(statistics pass and apply pass of a piece back to back, so that the apply pass finds its input there)?
64 clips of 60 s stereo fp32 = 1.47 GB; pieces of 64, 16, 8, 4 clips."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "audio-denoising_amd"))
from adn_hip import get_engine
e = get_engine(0)
SR, n, clips = 48000, 60 * 48000, 64
x = torch.empty((clips, n, 2), dtype=torch.float32, device="cuda")
tmp = torch.empty(n, dtype=torch.float64, device="cuda")
for c in range(8):
    e.synth_noisy_speech(n, SR, seed=42, stream=c, out=tmp)
    x[c, :, 0] = tmp.float()
    x[c, :, 1] = 0.8 * tmp.float()
for c in range(8, clips):
    x[c] = x[c % 8]
x = x.view(-1)
y = torch.empty_like(x)
bs = e.denoise_block_frames(SR, n)
starts = np.arange(0, n, bs, dtype=np.int64)
for piece in (64, 16, 8, 4):
    descs = []
    for c0 in range(0, clips, piece):
        off = np.concatenate([c * n * 2 + starts * 2 for c in range(c0, c0 + piece)])
        descs.append((e._i64(off), e._i64(np.full(off.shape, bs, dtype=np.int64))))
    def run():
        for d_off, d_len in descs:
            e.wavelet_denoise(x, d_off, d_len, y, d_off, n_ch=2, max_frames=bs)
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    print("pieces of %2d clips: %.4f ms per batch" % (piece, (time.perf_counter() - t0) / 10 * 1e3))

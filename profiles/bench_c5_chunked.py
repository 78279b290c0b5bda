"""BASELINE config 5's shape at a size one GPU can also do in one call: ONE long 96 kHz recording, fp64, through
the wavelet stage (10 % blocks, each block cut into time chunks), spectral subtraction and Wiener in K time chunks
(adn_hip/chunked.py, all chunks on this GPU in lock step), compared with the one-call results bit for bit and timed.
    python profiles/bench_c5_chunked.py [minutes=20] [chunks=8]"""
import json, os, sys, time
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "audio-denoising_amd"))
from adn_hip import get_engine, chunked as ck

minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
SR = 96000
e = get_engine(0)
n = int(minutes * 60 * SR) // 40 * 40                      # ten blocks, each a multiple of 2^level
dt = torch.float64
x = torch.empty(n, dtype=dt, device="cuda")
step = 1 << 24
for s in range(0, n, step):
    m = min(step, n - s)
    e.synth_noisy_speech(m, SR, stream=5, start=s, out=x[s:s + m])
noise = e.synth_noisy_speech(5 * SR, SR, stream=6).to(dt)
prof = e.noise_psd(noise, [5 * SR])[0]
red = ck.LocalReducer()
sync = torch.cuda.synchronize
res = {"workload": "one %.0f min 96 kHz mono recording, fp64, %d time chunks per block / file" % (minutes, K), "samples": n}

# ---- phase 2: ten blocks, every block in K chunks
bs = n // 10
sync(); t0 = time.perf_counter()
y_one = e.denoise_signal(x, SR)
sync(); res["wavelet_one_call_ms"] = (time.perf_counter() - t0) * 1e3
sync(); t0 = time.perf_counter()
outs = []
for b in range(10):
    blk = x[b * bs:(b + 1) * bs]
    owned, halo = ck.plan_wavelet_chunks(bs, K, 2, 8)
    bufs = [ck.wavelet_chunk_buffer(blk, a, c, halo) for a, c in owned]
    lens = [c - a for a, c in owned]
    sigma, thresh = ck.wavelet_block_threshold(e, bufs, lens, halo, bs, "db4", red)
    outs += ck.wavelet_block_apply(e, bufs, lens, halo, thresh, "db4", 2)
y_chunks = torch.cat(outs)
sync(); res["wavelet_chunked_ms"] = (time.perf_counter() - t0) * 1e3
res["wavelet_equal"] = bool(torch.equal(y_chunks, y_one[:y_chunks.numel()]))
res["wavelet_max_abs_diff"] = float((y_chunks - y_one[:y_chunks.numel()]).abs().max())
del outs, y_chunks

# ---- phase 2b in K chunks
x2 = y_one
n2 = x2.numel()
sync(); t0 = time.perf_counter()
yb_one, lb = e.spectral_subtract(x2, [n2], prof)
sync(); res["subtract_one_call_ms"] = (time.perf_counter() - t0) * 1e3
hops, frames = ck.plan_stft_chunks(n2, K)
sync(); t0 = time.perf_counter()
pieces = []
for h0, h1 in hops:
    s0, s1, keep = ck.stft_chunk_window(h0, h1, frames, n2)
    pieces.append(ck.spectral_subtract_chunk(e, x2[s0:s1], keep, (h1 - h0) * 512, prof))
yb = torch.cat(pieces)
sync(); res["subtract_chunked_ms"] = (time.perf_counter() - t0) * 1e3
res["subtract_equal"] = bool(torch.equal(yb, yb_one))
del pieces, yb

# ---- phase 2c in K chunks
x3 = yb_one
n3 = x3.numel()
sync(); t0 = time.perf_counter()
yc_one, lc, g_one = e.wiener(x3, [n3], SR, want_gain=True)
sync(); res["wiener_one_call_ms"] = (time.perf_counter() - t0) * 1e3
hops, frames = ck.plan_stft_chunks(n3, K)
fl = int(SR * 20 / 1000.0)
sync(); t0 = time.perf_counter()
stats = []
for i, (h0, h1) in enumerate(hops):
    t_a, t_b = h0, (frames if i == len(hops) - 1 else h1)
    s0 = t_a * 512
    s1 = n3 if t_b == frames else (t_b - 1) * 512 + 2048
    vb = max((t_a * 512 // fl - 5) * fl, 0)
    stats.append(ck.wiener_chunk_stats(e, x3[s0:s1], t_b - t_a, x3[vb:s1], vb, n3, t_a, SR))
gain = ck.wiener_gain_from_chunks(e, stats, frames, dt, red)
del stats
pieces = []
for h0, h1 in hops:
    s0, s1, keep = ck.stft_chunk_window(h0, h1, frames, n3, reach=0)
    pieces.append(ck.wiener_apply_chunk(e, x3[s0:s1], keep, (h1 - h0) * 512, gain))
yc = torch.cat(pieces)
sync(); res["wiener_chunked_ms"] = (time.perf_counter() - t0) * 1e3
res["wiener_gain_max_abs_diff"] = float((gain - g_one[0]).abs().max())
res["wiener_max_abs_diff"] = float((yc - yc_one).abs().max())
res["audio_sec"] = n / SR
res["device_memory_GB"] = torch.cuda.max_memory_allocated() / 1e9 + e.workspace_bytes() / 1e9
print(json.dumps(res))

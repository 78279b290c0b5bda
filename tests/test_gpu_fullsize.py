"""GPU parity at the sizes the benchmark and BASELINE.json's configurations use, through the C ABI:
config 2 (one 60 s 48 kHz stereo clip + 5 s noise through phases 1 / 2 / 2b / 2c, fp32 and fp64),
config 3's shape (a batch of 144 000-sample mono blocks, db8 level 8), a one-GPU rehearsal of
config 4 (files sharded over pretend ranks, one shared noise profile), and the denoiser's
optional modes (symmetric extension, per-channel time axis)."""
import json

import numpy as np
import pytest
import torch

import adn_oracle as orc
import synth
from gpu_util import engine, dev, host, check64, check32

pytestmark = pytest.mark.gpu
SR = 48000


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_config2_full_size_chain(dtype):
    """60 s stereo clip + 5 s noise, every stage against the oracle on the very input the stage
    saw on the GPU (for fp32: those binary32 values, widened), so that errors do not compound."""
    e = engine()
    dt = torch.float32 if dtype == "f32" else torch.float64
    chk = check32 if dtype == "f32" else check64
    n, nn = 60 * SR, 5 * SR
    x = synth.stereo_noisy_speech(n, SR, 42, 7)
    noise = synth.white_noise(nn, 42, 100007)
    xd, nd = dev(x, dt), dev(noise, dt)
    as64 = lambda t: host(t).astype(np.float64)   # noqa: E731

    # phase 1
    base = e.noise_psd(nd, [nn])
    b_or, _ = orc.noise_profile(as64(nd))
    rel = np.max(np.abs(as64(base[0]) - b_or) / b_or)
    assert rel <= (3e-6 if dtype == "f32" else 1e-12), rel

    # phase 2 (stereo: the transform runs across the channel axis, like the reference)
    y2, (sigma, thresh, zeroed) = e.denoise_signal(xd, SR, want_stats=True)
    y2_or, s_or, t_or = orc.denoise_signal(as64(xd), SR)
    chk(as64(y2), y2_or)
    assert np.max(np.abs(host(sigma) - s_or) / s_or) <= (1e-6 if dtype == "f32" else 1e-12)
    # threshold-boundary flips: coefficients whose |c| < t decision differs from the oracle's
    d = (as64(xd)[:, 0] - as64(xd)[:, 1]) / np.sqrt(2.0)
    bs = orc.denoise_block_size(SR, n)
    # per frame: cD2 is identically zero (always below the threshold) and cD1 = +-(L - R) / sqrt(2)
    z_or = np.array([bs + int(np.sum(np.abs(d[s:s + bs]) < t_or[i])) for i, s in enumerate(range(0, n, bs))])
    flips = np.abs(host(zeroed) - z_or)
    print("config 2 %s: zeroed coefficients per block differ from the oracle's by %s (of %d)" % (dtype, flips.tolist(), bs))
    assert flips.max() <= (bs // 2000 if dtype == "f32" else 0)

    # phase 2b on the GPU's own phase-2 output, with the profile as the stage file holds it (float32)
    b32 = as64(base[0]).astype(np.float32).astype(np.float64)
    y2b, l2b = e.spectral_subtract(y2.view(-1), [n], dev(b32, dt), n_ch=2)
    y2b_or = orc.spectral_subtract(as64(y2), b32)
    chk(as64(y2b), y2b_or)
    assert int(l2b[0]) == y2b_or.size == 2877952

    # phase 2c on the GPU's own phase-2b output
    y2c, l2c, gain = e.wiener(y2b, [int(l2b[0])], SR, want_gain=True)
    y2c_or, g_or = orc.wiener(as64(y2b), SR)
    chk(as64(y2c), y2c_or)
    assert int(l2c[0]) == y2c_or.size == 2875904
    assert np.max(np.abs(as64(gain[0]) - g_or)) <= (2e-4 if dtype == "f32" else 1e-10)


def test_wiener_full_size_batch():
    """Two 60 s signals in one call: the second equals its single-signal call bit for bit
    (signals are independent) and the oracle."""
    e = engine()
    n = 60 * SR
    x = e.synth_noisy_speech(2 * n, SR, stream=51).view(2, n)
    y, lens = e.wiener(x.reshape(-1), [n, n], SR)
    L = int(lens[0])
    y1, _ = e.wiener(x[1].contiguous(), [n], SR)
    assert torch.equal(y1, y[L:2 * L])
    want, _ = orc.wiener(host(x[1]), SR)
    check64(host(y1), want)


@pytest.mark.parametrize("wavelet,level", [("db8", 8), ("db4", 2)])
def test_config3_shape_batch(wavelet, level):
    """256 mono blocks of 144 000 samples (30 s clips at 48 kHz, 10 blocks per clip) in one call,
    fp32; a handful of blocks checked against the oracle, all of them for finiteness and for the
    batch-independence property (a block's result does not depend on its neighbours)."""
    e = engine()
    nb, bl = 256, 144000
    gen = 8                                         # distinct blocks; the rest repeat them
    x = torch.empty(nb * bl, dtype=torch.float32, device="cuda:0")
    for i in range(gen):
        x[i * bl:(i + 1) * bl] = e.synth_noisy_speech(bl, SR, stream=600 + i, start=i * bl).float()
    for i in range(gen, nb):
        x[i * bl:(i + 1) * bl] = x[(i % gen) * bl:(i % gen + 1) * bl]
    off = np.arange(nb, dtype=np.int64) * bl
    out = torch.empty_like(x)
    sigma, thresh, zeroed = e.wavelet_denoise(x, e._i64(off), e._i64(np.full(nb, bl)), out, e._i64(off),
                                              wavelet=wavelet, level=level, want_stats=True, max_frames=bl)
    y = host(out).reshape(nb, bl)
    assert np.isfinite(y).all()
    for i in range(gen, nb):                        # copies of a block give copies of its result
        assert np.array_equal(y[i], y[i % gen]), i
    xh = host(x[:gen * bl]).reshape(gen, bl).astype(np.float64)
    tot_flips = 0
    for i in (0, 3, 7):
        want, s_or, t_or, bands = orc.denoise_block(xh[i], wavelet, level, return_bands=True)
        check32(y[i], want)
        assert abs(float(sigma[i]) - s_or) <= 2e-6 * s_or
        z_or = sum(int(np.sum(np.abs(b) < t_or)) for b in bands[1:])
        tot_flips += abs(int(zeroed[i]) - z_or)
    print("config 3 shape %s level %d: threshold-boundary flips in 3 blocks: %d" % (wavelet, level, tot_flips))
    assert tot_flips <= 300


@pytest.mark.parametrize("world", [2, 4, 8])
def test_config4_rehearsal_file_shards(world):
    """Config 4 on one GPU: 16 mono files sharded round-robin over `world` pretend ranks, the noise
    profile learnt once on "rank 0" and handed to every shard through the same function the ranks
    use (its single-process path), phase 2b -> 2c per shard.  The merged outputs equal one call over
    all files bit for bit, and the oracle for sampled files."""
    from adn_hip import dist as adist
    e = engine()
    n_files, n = 16, 6 * SR
    lens = [n - 700 * i for i in range(n_files)]                       # ragged, like real files
    files = [synth.noisy_speech(lens[i], SR, 42, 700 + i).astype(np.float32) for i in range(n_files)]
    noise = synth.white_noise(5 * SR, 42, 100700).astype(np.float32)
    prof = e.noise_psd(dev(noise), [noise.size])[0]
    # one call over all files
    flat = dev(np.concatenate(files))
    yb_all, lb_all = e.spectral_subtract(flat, lens, prof)
    yc_all, lc_all = e.wiener(yb_all, [int(v) for v in lb_all], SR)
    oc = np.concatenate([[0], np.cumsum(lc_all)])
    merged = [None] * n_files
    for rank in range(world):
        idx = adist.shard_indices(n_files, rank, world)
        shard_prof = adist.broadcast_profile(prof if rank == 0 else prof.clone(), 1025, src=0)
        xs = dev(np.concatenate([files[i] for i in idx]))
        yb, lb = e.spectral_subtract(xs, [lens[i] for i in idx], shard_prof)
        yc, lc = e.wiener(yb, [int(v) for v in lb], SR)
        o = np.concatenate([[0], np.cumsum(lc)])
        for j, i in enumerate(idx):
            merged[i] = host(yc[o[j]:o[j + 1]])
    for i in range(n_files):
        assert np.array_equal(merged[i], host(yc_all[oc[i]:oc[i + 1]])), i
    p64 = host(prof).astype(np.float64)
    for i in (0, 5, 15):
        want_b = orc.spectral_subtract(files[i].astype(np.float64), p64)
        yb_i, lb_i = e.spectral_subtract(dev(files[i]), [lens[i]], prof)
        check32(host(yb_i), want_b)
        want_c, _ = orc.wiener(host(yb_i).astype(np.float64), SR)
        check32(merged[i], want_c)


def _check_hard32(got, x64, m, zeroed_gpu):
    """binary32 with a hard threshold: a coefficient whose magnitude sits within rounding of the threshold may
    be kept on one side and zeroed on the other (a threshold-boundary flip), which moves the output by up to
    the threshold round that coefficient.  Such coefficients are counted; every output sample outside their
    reach is held to the binary32 tolerance, the samples inside to twice the threshold."""
    F = 8 if m["wavelet"] == "db4" else 16
    want, s_or, t_or, bands = orc.denoise_block(x64, m["wavelet"], m["level"], "hard", mode="sym", return_bands=True)
    near = np.zeros(want.size, dtype=bool)
    n_near = 0
    for j, band in enumerate(bands[1:]):                       # cD_L ... cD_1
        lev = m["level"] - j
        scale = float(np.max(np.abs(band))) if band.size else 0.0
        idx = np.nonzero(np.abs(np.abs(band) - t_or) <= 4e-6 * max(scale, t_or))[0]
        n_near += idx.size
        for k in idx:                                          # output samples the coefficient reaches
            a = max((int(k) - F) << lev, 0)
            b = min((int(k) + F + 1) << lev, want.size)
            near[a:b] = True
    diff = np.abs(got.astype(np.float64) - want)
    z_or = sum(int(np.sum(np.abs(b) < t_or)) for b in bands[1:])
    print("hard threshold, binary32, %s level %d: %d coefficients within rounding of the threshold, zeroed %d "
          "(oracle %d), samples in their reach %d, largest difference there %.3g, elsewhere %.3g"
          % (m["wavelet"], m["level"], n_near, zeroed_gpu, z_or, int(near.sum()),
             float(diff[near].max()) if near.any() else 0.0, float(diff[~near].max())))
    assert abs(zeroed_gpu - z_or) <= n_near
    assert float(diff[~near].max()) <= 1e-4
    assert not near.any() or float(diff[near].max()) <= 2.0 * t_or + 1e-4


def test_symmetric_extension_blocks(golden):
    """mode='symmetric' through the denoiser entry point against PyWavelets' own results."""
    g = golden("denoise_modes")
    e = engine()
    for i, m in enumerate(json.loads(str(g["sym_meta_json"]))):
        x = g["sym%d_x" % i]
        for dt, chk in ((torch.float64, check64), (torch.float32, check32)):
            xd = dev(x, dt)
            n_out = x.size + (x.size & 1)
            out = torch.empty(n_out, dtype=dt, device="cuda:0")
            sigma, thresh, _ = e.wavelet_denoise(xd, e._i64([0]), e._i64([x.size]), out, e._i64([0]),
                                                 wavelet=m["wavelet"], level=m["level"], thr_mode=m["thr"],
                                                 mode="sym", want_stats=True, max_frames=x.size)
            want = g["sym%d_y" % i] if dt == torch.float64 else \
                orc.denoise_block(host(xd).astype(np.float64), m["wavelet"], m["level"], m["thr"], mode="sym")[0]
            if m["thr"] == "hard" and dt == torch.float32:
                _check_hard32(host(out), host(xd).astype(np.float64), m, int(_[0]))
                continue
            chk(host(out), want)
            if dt == torch.float64:
                assert abs(float(sigma[0]) - m["sigma"]) <= 1e-12 and abs(float(thresh[0]) - m["thresh"]) <= 1e-12


def test_time_axis_stereo_blocks(golden):
    """axis='time': every channel of a stereo block is a mono block of its own (the real DWT of a
    stereo file, which the reference's default axis never gives)."""
    g = golden("denoise_modes")
    e = engine()
    for i, m in enumerate(json.loads(str(g["time_meta_json"]))):
        x = g["time%d_x" % i]
        n = x.shape[0]
        n_out = n + (n & 1)
        for dt, chk in ((torch.float64, check64), (torch.float32, check32)):
            xd = dev(x, dt)
            out = torch.empty(n_out * 2, dtype=dt, device="cuda:0")
            sigma, thresh, zeroed = e.wavelet_denoise(xd.view(-1), e._i64([0]), e._i64([n]), out, e._i64([0]),
                                                      n_ch=2, wavelet=m["wavelet"], level=m["level"], axis="time",
                                                      want_stats=True, max_frames=n)
            y = host(out).reshape(n_out, 2)
            if dt == torch.float64:
                chk(y, g["time%d_y" % i])
                assert np.max(np.abs(host(sigma) - np.array(m["sigma"]))) <= 1e-12
            else:
                xs = host(xd).astype(np.float64)
                want = np.stack([orc.denoise_block(xs[:, ch], m["wavelet"], m["level"])[0] for ch in range(2)], axis=1)
                chk(y, want)
    # whole-file helper: blocks of 10 % of the file, both channels along time
    x = synth.stereo_noisy_speech(40010, 16000, 42, 410)
    y = host(e.denoise_signal(dev(x), 16000, axis="time"))
    want = np.stack([orc.denoise_signal(x[:, ch], 16000)[0] for ch in range(2)], axis=1)
    check64(y, want)


def test_config2_end_to_end_fp32_against_fp64_chain():
    """The whole chain in binary32 on the GPU (every stage fed the GPU's own previous output) against the oracle's
    binary64 chain on the same input file: the compounded error of phases 2 -> 2b -> 2c.  Tolerances of the
    binary32 mode: 1e-4 max-abs, 90 dB SNR of the difference, on the final output."""
    e = engine()
    n, nn = 60 * SR, 5 * SR
    x = synth.stereo_noisy_speech(n, SR, 42, 7).astype(np.float32)
    noise = synth.white_noise(nn, 42, 100007).astype(np.float32)
    xd, nd = dev(x), dev(noise)
    base = e.noise_psd(nd, [nn])
    y2 = e.denoise_signal(xd, SR)
    y2b, l2b = e.spectral_subtract(y2.view(-1), [n], base[0], n_ch=2)
    y2c, l2c = e.wiener(y2b, [int(l2b[0])], SR)
    x64, n64 = x.astype(np.float64), noise.astype(np.float64)
    b_or = orc.noise_profile(n64)[0].astype(np.float32).astype(np.float64)     # the profile file holds binary32
    y2_or = orc.denoise_signal(x64, SR)[0]
    y2b_or = orc.spectral_subtract(y2_or, b_or)
    y2c_or = orc.wiener(y2b_or, SR)[0]
    for name, got, want in (("phase 2", y2, y2_or), ("phase 2b", y2b, y2b_or), ("phase 2c", y2c, y2c_or)):
        g = host(got).astype(np.float64).reshape(want.shape)
        d = g - want
        snr = 10 * np.log10(np.sum(want ** 2) / max(np.sum(d ** 2), 1e-300))
        print("end to end, binary32 vs binary64 chain, %s: max-abs %.3g, SNR of the difference %.1f dB"
              % (name, float(np.max(np.abs(d))), snr))
    check32(host(y2c), y2c_or)


@pytest.mark.timeout(300)
def test_config3_full_size():
    """BASELINE.json config 3 at its stated size: 1024 clips of 30 s at 48 kHz, mono, db8 level 8, ten blocks per
    clip = 10 240 blocks of 144 000 samples (5.9 GB in, 5.9 GB out) in ONE call.  Sixteen distinct blocks, the
    rest copies: every copy's result equals its original's bit for bit (a block's result does not depend on its
    place in the batch), every sample is finite, sampled blocks against the oracle, per-block statistics equal."""
    e = engine()
    nb, bl, gen = 10240, 144000, 16
    x = torch.empty(nb * bl, dtype=torch.float32, device="cuda:0")
    for i in range(gen):
        x[i * bl:(i + 1) * bl] = e.synth_noisy_speech(bl, SR, stream=900 + i, start=i * bl).float()
    xv = x.view(nb // gen, gen * bl)
    xv[1:] = xv[0]
    off = np.arange(nb, dtype=np.int64) * bl
    out = torch.empty_like(x)
    sigma, thresh, zeroed = e.wavelet_denoise(x, e._i64(off), e._i64(np.full(nb, bl)), out, e._i64(off),
                                              wavelet="db8", level=8, want_stats=True, max_frames=bl)
    ov = out.view(nb // gen, gen * bl)
    assert bool(torch.isfinite(out).all())
    for r in range(1, nb // gen, 37):                       # every 37th repetition of the sixteen blocks ...
        assert torch.equal(ov[r], ov[0]), r
    assert torch.equal(ov[nb // gen - 1], ov[0])           # ... and the last one
    sg = sigma.view(nb // gen, gen)
    assert bool((sg == sg[0]).all()) and bool((zeroed.view(nb // gen, gen) == zeroed.view(nb // gen, gen)[0]).all())
    xh = host(x[:gen * bl]).reshape(gen, bl).astype(np.float64)
    y = host(out[:gen * bl]).reshape(gen, bl)
    for i in (0, 5, 15):
        want, s_or, t_or = orc.denoise_block(xh[i], "db8", 8)
        check32(y[i], want)
        assert abs(float(sigma[i]) - s_or) <= 2e-6 * s_or


@pytest.mark.timeout(300)
def test_config4_full_size_file_shards():
    """BASELINE.json config 4 at its stated size on one GPU: 10 h of 48 kHz audio as 1200 mono files of 30 s,
    sharded round-robin over 8 ranks (here: eight calls on one device, one per rank's shard), one noise profile
    learnt once and handed to every shard, phase 2b -> 2c.  Thirteen distinct files, the rest copies: every file's
    result equals that of the file it copies in whatever shard it landed, and the oracle for sampled files."""
    from adn_hip import dist as adist
    e = engine()
    n_files, n, gen, world = 1200, 30 * SR, 13, 8
    files = torch.empty((gen, n), dtype=torch.float32, device="cuda:0")
    for i in range(gen):
        files[i] = e.synth_noisy_speech(n, SR, stream=1300 + i).float()
    noise = dev(synth.white_noise(5 * SR, 42, 101300).astype(np.float32))
    prof = e.noise_psd(noise, [5 * SR])[0]
    first = {}                                              # distinct file -> its first result
    L = None
    for rank in range(world):
        idx = adist.shard_indices(n_files, rank, world)
        shard_prof = adist.broadcast_profile(prof if rank == 0 else prof.clone(), 1025, src=0)
        xs = files[torch.as_tensor(idx % gen, device="cuda:0")].reshape(-1)
        yb, lb = e.spectral_subtract(xs, [n] * len(idx), shard_prof)
        yc, lc = e.wiener(yb, [int(v) for v in lb], SR)
        L = int(lc[0])
        ycv = yc.view(len(idx), L)
        for j, i in enumerate(idx):
            k = int(i % gen)
            if k not in first:
                first[k] = ycv[j].clone()
        for k in sorted(set(int(v) for v in idx % gen)):      # all copies of file k in this shard at once
            rows = torch.as_tensor(np.nonzero(idx % gen == k)[0], device="cuda:0")
            assert bool((ycv[rows] == first[k]).all()), (rank, k)
    p64 = host(prof).astype(np.float64)
    for k in (0, 7):
        xb = host(files[k]).astype(np.float64)
        yb_or = orc.spectral_subtract(xb, p64)
        yb_k, _ = e.spectral_subtract(files[k].contiguous(), [n], prof)
        check32(host(yb_k), yb_or)
        want_c, _ = orc.wiener(host(yb_k).astype(np.float64), SR)
        check32(host(first[k]), want_c)


@pytest.mark.timeout(600)
def test_config5_full_size_time_chunks():
    """BASELINE config 5 at its stated size: ONE 24 h, 96 kHz recording (8 294 400 000 samples, binary64, 66 GB) cut into
    8 time chunks, all on this GPU in lock step (the code path of one chunk per rank, the all-reduce replaced by a sum
    over the list).  Wavelet stage: the first and the last 10 % block (829 440 000 samples each) in 8 chunks against the
    one-call result of the block, bit for bit, the block's sigma against the oracle's MAD of the whole finest band,
    windows round a seam against the oracle.  Spectral subtraction: the whole recording in 8 chunks against the one-call
    result, bit for bit, windows round the seams against the oracle.  (The Wiener stage of this recording cannot be
    rehearsed on one GPU: its power matrix alone is 133 GB in binary64 -- one rank's share is 16.6 GB.)"""
    from adn_hip import chunked as ck
    e = engine()
    sr, K = 96000, 8
    n = 24 * 3600 * sr
    x = torch.empty(n, dtype=torch.float64, device="cuda:0")
    step = 1 << 28
    for s in range(0, n, step):
        m = min(step, n - s)
        e.synth_noisy_speech(m, sr, seed=42, stream=7, start=s, out=x[s:s + m])
    bs = e.denoise_block_frames(sr, n)
    assert bs == 829440000 and n == 10 * bs
    red = ck.LocalReducer()
    g = orc.filters("db4")[1]
    for b in (0, 9):
        blk = x[b * bs:(b + 1) * bs]
        out = torch.empty_like(blk)
        sigma1, thresh1, _ = e.wavelet_denoise(blk, e._i64([0]), e._i64([bs]), out, e._i64([0]), want_stats=True, max_frames=bs)
        owned, halo = ck.plan_wavelet_chunks(bs, K, 2, 8)
        bufs = [ck.wavelet_chunk_buffer(blk, a, c, halo) for a, c in owned]
        lens = [c - a for a, c in owned]
        sigma, thresh = ck.wavelet_block_threshold(e, bufs, lens, halo, bs, "db4", red)
        assert float(sigma[0]) == float(sigma1[0])
        assert abs(float(thresh[0]) - float(thresh1[0])) <= 4e-16 * float(thresh1[0])
        parts = ck.wavelet_block_apply(e, bufs, lens, halo, thresh1.to(torch.float64), "db4", 2)
        pos = 0
        for p in parts:
            assert torch.equal(p, out[pos:pos + p.numel()])
            pos += p.numel()
        assert pos == bs
        del bufs, parts
        if b == 0:
            # the oracle's MAD of the block's whole finest band, formed piece by piece (a coefficient needs the 8
            # samples under its taps: pieces overlap by a margin and keep their interiors)
            xh = host(blk)
            piece, margin, cd = 1 << 26, 64, []
            for s in range(0, bs, piece):
                a, c = max(s - margin, 0), min(s + piece + margin, bs)
                d = orc.dwt_per(xh[a:c], "db4")[1]
                cd.append(d[(s - a) // 2:(s - a) // 2 + (min(s + piece, bs) - s) // 2].copy())
            # the first and last coefficients of the block wrap round it (periodization)
            cd = np.concatenate(cd)
            edge = orc.dwt_per(np.concatenate([xh[-64:], xh[:64]]), "db4")[1]
            cd[:4] = edge[32:36]
            cd[-4:] = edge[28:32]
            assert abs(orc.mad(cd) - float(sigma1[0])) <= 1e-12
            del cd
            # windows round the first seam and at the block's end, with the block's threshold
            t = float(thresh1[0])
            for w0 in (owned[1][0] - 8192, bs - 16384):
                xw = xh[w0:w0 + 16384]
                co = orc.wavedec(xw, "db4", 2)
                yw = orc.waverec([co[0]] + [orc.threshold(c, t, "soft") for c in co[1:]], "db4")
                check64(host(out[w0 + 256:w0 + 16384 - 256]), yw[256:-256])
            del xh
        del out
    torch.cuda.empty_cache()

    # ---- spectral subtraction of the whole recording
    noise = e.synth_noisy_speech(5 * sr, sr, stream=6).to(torch.float64)
    prof = e.noise_psd(noise, [5 * sr])[0]
    y1, l1 = e.spectral_subtract(x, [n], prof)
    assert int(l1[0]) == y1.numel()
    hops, frames = ck.plan_stft_chunks(n, K)
    assert frames > 16000000
    ph = host(prof).astype(np.float64)
    pos = 0
    for i, (h0, h1) in enumerate(hops):
        s0, s1, keep = ck.stft_chunk_window(h0, h1, frames, n)
        piece = ck.spectral_subtract_chunk(e, x[s0:s1], keep, (h1 - h0) * 512, prof)
        assert torch.equal(piece, y1[pos:pos + piece.numel()])
        if i > 0:
            # oracle on a window round the seam: frames well inside the window see the same samples as in the file
            s = h0 * 512
            w0, w1 = s - 20480, s + 20480
            want = orc.spectral_subtract(host(x[w0:w1]).astype(np.float64), ph)
            check64(host(y1[s - 8192:s + 8192]), want[s - 8192 - w0:s + 8192 - w0])
        pos += piece.numel()
        del piece
    assert pos == y1.numel()

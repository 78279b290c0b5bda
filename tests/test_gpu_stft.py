"""GPU parity of the STFT stages (phases 1, 2b, 2c) through the C ABI: HIP kernels vs the
CPU oracle and vs golden vectors produced by running the reference stage scripts.
fp64: 1e-9 max-abs; fp32: 1e-4 max-abs and >= 90 dB SNR of the difference."""
import json

import numpy as np
import pytest
import torch

import adn_oracle as orc
import synth
from gpu_util import engine, dev, host, check64, check32

pytestmark = pytest.mark.gpu
SR = 48000


def _noise_mag(golden):
    return golden("stages")["p1_baseline_col0"].astype(np.float64)


def test_noise_profile_fp64_and_fp32(golden):
    g = golden("stages")
    e = engine()
    noise = synth.white_noise(240000, 42, 1000)
    base_or, T = orc.noise_profile(noise)
    b64 = host(e.noise_psd(dev(noise), [noise.size]))[0]
    check64(b64, base_or, 1e-12)
    ref32 = g["p1_baseline_col0"]
    assert np.max(np.abs(b64.astype(np.float32) - ref32) / ref32) <= 1.2e-7
    b32 = host(e.noise_psd(dev(noise, torch.float32), [noise.size]))[0]
    assert np.max(np.abs(b32 - base_or) / base_or) <= 2e-5
    # a profile learnt with another window length (reference passes win_length as nperseg)
    b1024 = host(e.noise_psd(dev(noise), [noise.size], n_fft=1024, hop=256))[0]
    check64(b1024, orc.noise_profile(noise, n_fft=1024, hop=256, win_length=1024)[0], 1e-12)
    assert np.max(np.abs(b1024.astype(np.float32) - g["p1_1024_baseline_col0"]) / g["p1_1024_baseline_col0"]) <= 1.2e-7
    # stereo noise sample is folded to mono; unpadded variant (center: false)
    st = np.stack([noise[:50000], synth.white_noise(50000, 42, 1002)], axis=1)
    check64(host(e.noise_psd(dev(st).view(-1), [50000], n_ch=2))[0], orc.noise_profile(st)[0], 1e-12)
    check64(host(e.noise_psd(dev(noise[:50000]), [50000], padded=False))[0],
            orc.noise_profile(noise[:50000], padded=False)[0], 1e-12)


def test_spectral_subtract_fp64_vs_reference(golden):
    g = golden("stages")
    e = engine()
    nm = _noise_mag(golden)
    x = synth.noisy_speech(60000, SR, 42, 20)
    y, lens = e.spectral_subtract(dev(x), [x.size], dev(nm))
    assert list(lens) == [g["p2b_y"].size]
    check64(host(y), g["p2b_y"])
    check64(host(y), orc.spectral_subtract(x, nm))


def test_spectral_subtract_fp32(golden):
    g = golden("stages")
    e = engine()
    nm = _noise_mag(golden)
    x32 = synth.noisy_speech(60000, SR, 42, 20).astype(np.float32)
    y, _ = e.spectral_subtract(dev(x32), [x32.size], dev(nm.astype(np.float32)))
    want = orc.spectral_subtract(x32.astype(np.float64), nm)
    print("2b fp32 (max-abs, SNR dB):", check32(host(y), want))
    check32(host(y), g["p2b_y"])


def test_spectral_subtract_edges_and_batch(golden):
    g = golden("stages")
    e = engine()
    nm = _noise_mag(golden)
    xs = [synth.noisy_speech(n, SR, 42, 22) for n in (2048, 2100, 5000)]
    xs.append(synth.noisy_speech(60000, SR, 42, 20))
    flat = np.concatenate(xs)
    y, lens = e.spectral_subtract(dev(flat), [x.size for x in xs], dev(nm))
    y = host(y)
    assert list(lens) == [0, 512, 3072, g["p2b_y"].size]
    pos = 0
    for tag, ln in zip(("n2048", "n2100", "n5000"), lens[:3]):
        check64(y[pos:pos + ln], g["p2b_%s_y" % tag])
        pos += ln
    check64(y[pos:], g["p2b_y"])
    # stereo input is folded to mono
    xst = synth.stereo_noisy_speech(30000, SR, 42, 23)
    ys, _ = e.spectral_subtract(dev(xst).view(-1), [30000], dev(nm), n_ch=2)
    check64(host(ys), g["p2b_stereo_y"])
    # profile with another bin count: the host resizes it with np.interp like the reference
    nm_r = orc.resize_noise(g["p1_1024_baseline_col0"].astype(np.float64), 1025)
    yr, _ = e.spectral_subtract(dev(xs[2]), [5000], dev(nm_r))
    check64(host(yr), g["p2b_resize_y"])
    with pytest.raises(ValueError, match="window is longer than input signal"):
        e.spectral_subtract(dev(xs[0][:1000]), [1000], dev(nm))
    # other transform sizes against the oracle
    for n_fft, hop in ((1024, 256), (512, 128), (2048, 1024), (256, 32), (64, 16)):
        nmo = np.abs(np.sin(np.arange(n_fft // 2 + 1) * 0.01)) * 0.002 + 0.001
        x = synth.noisy_speech(20000, SR, 42, 25)
        yo, _ = e.spectral_subtract(dev(x), [x.size], dev(nmo), n_fft=n_fft, hop=hop)
        check64(host(yo), orc.spectral_subtract(x, nmo, n_fft=n_fft, hop=hop))
        yw, _ = e.wiener(dev(x), [x.size], SR, n_fft=n_fft, hop=hop)
        check64(host(yw), orc.wiener(x, SR, n_fft=n_fft, hop=hop)[0])
    # n_fft = 4096 exceeds the LDS budget in fp64 with time smoothing; fp32 covers it
    nmo = np.abs(np.sin(np.arange(2049) * 0.01)) * 0.002 + 0.001
    x32 = synth.noisy_speech(40000, SR, 42, 25).astype(np.float32)
    yo, _ = e.spectral_subtract(dev(x32), [x32.size], dev(nmo.astype(np.float32)), n_fft=4096, hop=1024)
    check32(host(yo), orc.spectral_subtract(x32.astype(np.float64), nmo, n_fft=4096, hop=1024))
    from adn_hip import AdnError
    with pytest.raises(AdnError, match="does not fit"):
        e.spectral_subtract(dev(x32.astype(np.float64)), [x32.size], dev(nmo), n_fft=4096, hop=1024)
    with pytest.raises(AdnError, match="hop must divide n_fft/2"):
        e.spectral_subtract(dev(x32), [x32.size], dev(nmo[:129].astype(np.float32)), n_fft=256, hop=256)


def test_wiener_fp64_vs_reference(golden):
    g = golden("stages")
    e = engine()
    x = synth.noisy_speech(60000, SR, 42, 20)
    y2b = g["p2b_y"]
    y, lens, gain = e.wiener(dev(y2b), [y2b.size], SR, want_gain=True)
    check64(host(y), g["p2c_y"])
    y_or, gain_or = orc.wiener(y2b, SR)
    check64(host(gain)[0], gain_or, 1e-12)
    check64(host(e.wiener(dev(x), [x.size], SR, vad_guided=False)[0]), g["p2c_novad_y"])
    check64(host(e.wiener(dev(x), [x.size], SR, vad_guided=True)[0]), g["p2c_vad_x_y"])
    xq = 1e-4 * synth.noisy_speech(30000, SR, 42, 21)
    check64(host(e.wiener(dev(xq), [xq.size], SR)[0]), g["p2c_quiet_y"])


def test_wiener_fp32(golden):
    g = golden("stages")
    e = engine()
    x32 = synth.noisy_speech(60000, SR, 42, 20).astype(np.float32)
    y, _ = e.wiener(dev(x32), [x32.size], SR)
    want, _ = orc.wiener(x32.astype(np.float64), SR)
    print("2c fp32 (max-abs, SNR dB):", check32(host(y), want))
    check32(host(y), g["p2c_vad_x_y"])


def test_core_chain_stereo_fp64(golden):
    """phases 1 / 2 / 2b / 2c chained on device against the reference's core_pipeline run."""
    g = golden("pipeline")
    e = engine()
    x = synth.stereo_noisy_speech(40000, SR, 42, 30)
    noise = synth.white_noise(48000, 42, 1001)
    base = e.noise_psd(dev(noise), [noise.size])
    b32 = host(base)[0].astype(np.float32)
    assert np.max(np.abs(b32 - g["baseline_col0"]) / g["baseline_col0"]) <= 1.2e-7
    y2 = e.denoise_signal(dev(x), SR)
    check64(host(y2), g["wavelet_y"])
    # the stage file holds float32 baseline_mag; 2b reads it back as float32 values
    y2b, l2b = e.spectral_subtract(y2.view(-1), [y2.shape[0]], dev(b32.astype(np.float64)), n_ch=2)
    check64(host(y2b), g["ss_y"])
    y2c, _ = e.wiener(y2b, [int(l2b[0])], SR)
    check64(host(y2c), g["wiener_y"])


def test_large_batch_properties():
    """benchmark-sized signals (60 s at 48 kHz): size-independent properties.  With a zero
    noise profile and beta = 1 the 2b stage is the identity up to the box filter, so with
    smoothing 1x1 the output equals the input shifted by n_fft/2 (scipy's istft trim).  (The
    Wiener stage at this size is covered by test_gpu_fullsize.py.)"""
    e = engine()
    n = 60 * SR
    x = e.synth_noisy_speech(2 * n, SR, stream=50).view(2, n)
    zero = torch.zeros(1025, dtype=torch.float64, device="cuda:0")
    y, lens = e.spectral_subtract(x.reshape(-1), [n, n], zero, alpha=0.0, beta=1.0, smooth_time=1, smooth_freq=1)
    L = int(lens[0])
    assert L == (e.stft_num_frames(n, 2048, 512) - 1) * 512
    for s in range(2):
        ref = torch.clamp(x[s, 1024:1024 + L], -1.0, 1.0)
        m = min(L, n - 1024)
        assert float((y[s * L:s * L + m] - ref[:m]).abs().max()) <= 1e-12
    # batch result == single-signal result, bit for bit (independent signals)
    y1, _ = e.spectral_subtract(x[1].contiguous(), [n], zero, alpha=0.0, beta=1.0, smooth_time=1, smooth_freq=1)
    assert torch.equal(y1, y[L:2 * L])


@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
def test_batch_of_mixed_lengths_equals_single_calls(golden, dt):
    """Signals of different lengths in one call: the fused kernels size the runs of hops from all lengths together and
    hand every signal the workgroups its own length needs.  Every signal's output must equal, bit for bit, what a call
    with that signal alone gives (other run boundaries, same arithmetic per hop), for 1, 2b and 2c, and the oracle's
    for a sample of them - shortest signals (no output hop, one hop), lengths around hop multiples, long ones."""
    e = engine()
    nm = _noise_mag(golden)
    rng = np.random.default_rng(77)
    lens_in = [2048, 2100, 2559, 2560, 2561, 3072, 4095, 4096, 300000, 2048 + 512 * 177, 2047 + 512 * 178]
    lens_in += [int(v) for v in rng.integers(2048, 200000, size=29)]
    order = rng.permutation(len(lens_in))
    lens_in = [lens_in[i] for i in order]
    xs = [synth.noisy_speech(n, SR, 42, 100 + i) for i, n in enumerate(lens_in)]
    flat = dev(np.concatenate(xs), dt)
    nmd = dev(nm, dt)
    yb, lb = e.spectral_subtract(flat, lens_in, nmd)
    yc, lc = e.wiener(flat, lens_in, SR)
    pb = e.noise_psd(flat, lens_in)                 # the statistics rows alone (phase 1), padded and unpadded frames
    pu = e.noise_psd(flat, lens_in, padded=False)
    assert list(lb) == list(lc) == [(e.stft_num_frames(n, 2048, 512) - 1) * 512 for n in lens_in]
    pos_in = pos = 0
    for i, n in enumerate(lens_in):
        xi = flat[pos_in:pos_in + n].clone()
        L = int(lb[i])
        y1, _ = e.spectral_subtract(xi, [n], nmd)
        assert torch.equal(y1, yb[pos:pos + L]), (i, n)
        w1, _ = e.wiener(xi, [n], SR)
        assert torch.equal(w1, yc[pos:pos + L]), (i, n)
        assert torch.equal(e.noise_psd(xi, [n])[0], pb[i]), (i, n)
        assert torch.equal(e.noise_psd(xi, [n], padded=False)[0], pu[i]), (i, n)
        if dt == torch.float64 and i % 8 == 0:
            check64(host(yb[pos:pos + L]), orc.spectral_subtract(xs[i], nm))
            check64(host(yc[pos:pos + L]), orc.wiener(xs[i], SR)[0])
        pos_in += n
        pos += L
    assert pos == yb.numel() == yc.numel()


def test_packed_files_with_aligned_starts(golden):
    """Engine.pack_signals: files of odd lengths in one buffer, every start 16-byte aligned; the batched stages on
    it (offsets=...) equal the single calls bit for bit, mono and stereo, both precisions."""
    e = engine()
    for dt, n_ch in ((torch.float32, 1), (torch.float64, 1), (torch.float32, 2)):
        nmd = dev(_noise_mag(golden), dt)
        lens_in = [2049, 30001, 4099, 77777, 5003]
        sigs = [dev(synth.stereo_noisy_speech(n, SR, 42, 200 + i) if n_ch == 2 else synth.noisy_speech(n, SR, 42, 200 + i), dt)
                for i, n in enumerate(lens_in)]
        flat, lens, offs = e.pack_signals(sigs, n_ch=n_ch)
        assert list(lens) == lens_in and all(int(o) * flat.element_size() % 16 == 0 for o in offs)
        assert flat.data_ptr() % 16 == 0
        yb, lb = e.spectral_subtract(flat, lens, nmd, n_ch=n_ch, offsets=offs)
        yc, _ = e.wiener(flat, lens, SR, n_ch=n_ch, offsets=offs)
        pos = 0
        for s, n, L in zip(sigs, lens_in, lb):
            assert torch.equal(e.spectral_subtract(s.reshape(-1), [n], nmd, n_ch=n_ch)[0], yb[pos:pos + int(L)])
            assert torch.equal(e.wiener(s.reshape(-1), [n], SR, n_ch=n_ch)[0], yc[pos:pos + int(L)])
            pos += int(L)


def test_batch_of_1500_short_files_equals_its_halves(golden):
    """More than 1024 signals in one call (the device-side run map walks them in blocks of 1024): the outputs equal,
    bit for bit, those of the two halves of the batch processed on their own, and a sample equals single calls."""
    e = engine()
    nmd = dev(_noise_mag(golden), torch.float32)
    rng = np.random.default_rng(5)
    lens_in = [int(v) for v in rng.integers(2048, 9000, size=1500)]
    flat = e.synth_noisy_speech(sum(lens_in), SR, stream=61).to(torch.float32)
    half = 750
    n_half = sum(lens_in[:half])
    yb, lb = e.spectral_subtract(flat, lens_in, nmd)
    yc, lc = e.wiener(flat, lens_in, SR)
    pb = e.noise_psd(flat, lens_in)
    parts = [(flat[:n_half].clone(), lens_in[:half]), (flat[n_half:].clone(), lens_in[half:])]
    assert torch.equal(yb, torch.cat([e.spectral_subtract(x, l, nmd)[0] for x, l in parts]))
    assert torch.equal(yc, torch.cat([e.wiener(x, l, SR)[0] for x, l in parts]))
    assert torch.equal(pb, torch.cat([e.noise_psd(x, l) for x, l in parts]))
    off_in = np.concatenate([[0], np.cumsum(lens_in)])
    off_out = np.concatenate([[0], np.cumsum(lb)])
    for i in (0, 1023, 1024, 1025, 1499):
        xi = flat[off_in[i]:off_in[i + 1]].clone()
        assert torch.equal(e.spectral_subtract(xi, [lens_in[i]], nmd)[0], yb[off_out[i]:off_out[i + 1]]), i
        assert torch.equal(e.wiener(xi, [lens_in[i]], SR)[0], yc[off_out[i]:off_out[i + 1]]), i


def test_generic_fp32_kernels_match_specialised(golden):
    """binary32 / n_fft 2048 normally runs the register-resident kernels; the shape-generic
    LDS kernels must agree with them (and with the oracle) on the same input."""
    e = engine()
    nm = _noise_mag(golden).astype(np.float32)
    x32 = synth.noisy_speech(60000, SR, 42, 20).astype(np.float32)
    want_b = orc.spectral_subtract(x32.astype(np.float64), nm.astype(np.float64))
    want_c, _ = orc.wiener(x32.astype(np.float64), SR)
    outs = {}
    for generic in (False, True):
        e.force_generic(generic)
        try:
            yb, _ = e.spectral_subtract(dev(x32), [x32.size], dev(nm))
            yc, _ = e.wiener(dev(x32), [x32.size], SR)
            outs[generic] = (host(yb), host(yc))
        finally:
            e.force_generic(False)
        check32(outs[generic][0], want_b)
        check32(outs[generic][1], want_c)
    assert np.max(np.abs(outs[True][0] - outs[False][0])) <= 2e-6
    assert np.max(np.abs(outs[True][1] - outs[False][1])) <= 2e-6


@pytest.mark.parametrize("hop", [512, 256])
@pytest.mark.parametrize("n_ch", [1, 2])
def test_frame_count_edges_all_kernels(hop, n_ch):
    """Ragged batch whose frame counts straddle the kernels' iteration sizes (the register-resident
    fp32 kernels take 8 new frames per iteration and park the newest one; the shape-generic kernels
    3, 4 or 8): every signal must match the oracle, in fp32 (fast kernels), in fp32 with the
    shape-generic kernels forced, and in fp64."""
    e = engine()
    n_fft = 2048
    counts = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15, 16, 17, 18, 23, 24, 25, 26, 31, 32, 33, 34, 47, 48, 49, 97, 200]
    rng = np.random.default_rng(hop + n_ch)
    lens = [(t - 1) * hop + int(rng.integers(0, hop)) for t in counts]
    lens = [max(n, n_fft) for n in lens]
    xs = []
    for i, n in enumerate(lens):
        ch = [synth.noisy_speech(n, SR, 42, 300 + 2 * i + c, speech_gain=1.0 - 0.3 * c) for c in range(n_ch)]
        xs.append(np.stack(ch, axis=1) if n_ch == 2 else ch[0])
    nm = np.abs(np.sin(np.arange(n_fft // 2 + 1) * 0.013)) * 0.004 + 0.002
    mono = [x.mean(axis=1) if n_ch == 2 else x for x in xs]
    want_b = [orc.spectral_subtract(m, nm, n_fft=n_fft, hop=hop) for m in mono]
    want_c = [orc.wiener(m, SR, n_fft=n_fft, hop=hop)[0] for m in mono]
    flat = np.concatenate([x.reshape(-1) for x in xs])
    for dt, generic, chk in ((torch.float32, False, check32), (torch.float32, True, check32),
                             (torch.float64, False, check64)):
        xd = dev(flat, dt)
        e.force_generic(generic)
        try:
            yb, lb = e.spectral_subtract(xd, lens, dev(nm, dt), n_ch=n_ch, n_fft=n_fft, hop=hop)
            yc, lc = e.wiener(xd, lens, SR, n_ch=n_ch, n_fft=n_fft, hop=hop)
        finally:
            e.force_generic(False)
        yb, yc = host(yb), host(yc)
        ob = np.concatenate([[0], np.cumsum(lb)])
        oc = np.concatenate([[0], np.cumsum(lc)])
        for i in range(len(lens)):
            assert lb[i] == len(want_b[i]) and lc[i] == len(want_c[i])
            if dt == torch.float32:
                # the fp32 input differs from the fp64 one by rounding: compare with the oracle on it
                m32 = (xs[i].astype(np.float32).astype(np.float64))
                m32 = m32.mean(axis=1) if n_ch == 2 else m32
                chk(yb[ob[i]:ob[i + 1]], orc.spectral_subtract(m32, nm.astype(np.float32).astype(np.float64),
                                                                 n_fft=n_fft, hop=hop))
            else:
                chk(yb[ob[i]:ob[i + 1]], want_b[i])
                chk(yc[oc[i]:oc[i + 1]], want_c[i])
        if dt == torch.float32:
            # Wiener fp32: P10 / VAD decisions can flip on rounding for tiny signals; compare the batch
            # with single-signal calls (bit-identical) and the long signals with the oracle
            for i in (0, 7, len(lens) - 1):
                x1 = dev(xs[i].reshape(-1), dt)
                e.force_generic(generic)
                try:
                    y1, _ = e.wiener(x1, [lens[i]], SR, n_ch=n_ch, n_fft=n_fft, hop=hop)
                finally:
                    e.force_generic(False)
                assert np.array_equal(host(y1), yc[oc[i]:oc[i + 1]])


@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
def test_wiener_long_rows(dt):
    """A 45 s signal at 48 kHz: the per-bin statistics run over more than 4096 frames, which takes
    the many-threads-per-row variant of the LDS select (with the fused VAD-masked mean)."""
    e = engine()
    n = 45 * SR
    x = synth.noisy_speech(n, SR, 42, 90)
    if dt == torch.float32:
        x = x.astype(np.float32).astype(np.float64)
    want, gain_want = orc.wiener(x, SR)
    y, lens, gain = e.wiener(dev(x, dt), [n], SR, want_gain=True)
    assert e.stft_num_frames(n, 2048, 512) > 4096
    chk = check64 if dt == torch.float64 else check32
    chk(host(y), want)
    tol = 1e-9 if dt == torch.float64 else 2e-4
    assert np.max(np.abs(host(gain)[0].astype(np.float64) - gain_want)) <= tol
    # vad_guided = False: the speech estimate is the per-bin median (second selection, unmasked)
    want2, _ = orc.wiener(x, SR, vad_guided=False)
    y2, _ = e.wiener(dev(x, dt), [n], SR, vad_guided=False)
    chk(host(y2), want2)


def test_statistics_paths_agree():
    """The Wiener stage and the noise profile on the grouped-rows path (rows written by waves that work alone,
    register-resident select, voice-activity energies added up by the rows kernel) against the transposed-rows
    path with its separate voice-activity pass (diagnostic option bits 64 / 128 switch the new pieces off; bit 1024
    lets two waves share every row):
    binary32 bit for bit (both paths transform a frame with the same arithmetic), binary64 to 1e-12 (the
    transposed binary64 rows come from the shape-generic transform)."""
    e = engine()
    SR48 = 48000
    n = 7 * SR48 + 123
    for dt, tol, n_ch in ((torch.float32, 0.0, 1), (torch.float64, 1e-12, 1), (torch.float32, 0.0, 2), (torch.float64, 1e-12, 2)):
        # two signals, the second at an odd offset (its frames take the staged loader); stereo: interleaved channels
        if n_ch == 1:
            x = dev(synth.noisy_speech(n, SR48, 42, 930), dt)
            xs = torch.cat([x, x[: 5 * SR48 + 77]])
        else:
            x = dev(synth.stereo_noisy_speech(n, SR48, 42, 931), dt)
            xs = torch.cat([x.reshape(-1), x[: 5 * SR48 + 77].reshape(-1)])
        lens = [n, 5 * SR48 + 77]
        res = {}
        for name, bits in (("grouped", 0), ("vad_apart", 128), ("transposed", 64), ("shared_rows", 1024)):
            e.lib.adn_set_option(e._h, 2, bits)
            try:
                y, l, g = e.wiener(xs, lens, SR48, n_ch=n_ch, want_gain=True)
                p = e.noise_psd(xs, lens, n_ch=n_ch)
            finally:
                e.lib.adn_set_option(e._h, 2, 0)
            res[name] = (y.clone(), g.clone(), p.clone())
        # rows shared by two waves (bit 1024; what rows of more than 8192 frames take by themselves): the order
        # statistics are the same keys; the voice-activity mean adds its terms in another order
        y, g, p = res["shared_rows"]
        assert torch.equal(p, res["grouped"][2]), str(dt)
        assert float((g - res["grouped"][1]).abs().max()) <= (1e-12 if dt == torch.float64 else 2e-6)
        assert float((y - res["grouped"][0]).abs().max()) <= (1e-12 if dt == torch.float64 else 2e-6)
        for name in ("vad_apart", "transposed"):
            for a, b in zip(res["grouped"], res[name]):
                if tol == 0.0 or name == "vad_apart":
                    assert torch.equal(a, b), (str(dt), name)
                else:
                    assert float((a - b).abs().max()) <= tol, (str(dt), name)


@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
def test_wiener_and_noise_profile_rows_of_7000_frames(dt):
    """Files of 65 to 87 s at 48 kHz have 6145 ... 8192 STFT frames: their power rows still take the grouped layout
    and the register-resident row select (128 keys per lane).  One 75 s mono signal (7030 frames) through the Wiener
    stage, with and without voice-activity guidance, and as a noise sample through the noise profile, against the
    oracle; the gain row to 1e-12 in binary64."""
    e = engine()
    n = 75 * SR
    x = synth.noisy_speech(n, SR, 42, 31)
    npdt = np.float64 if dt == torch.float64 else np.float32
    xin = x.astype(npdt)
    for vad in (True, False):
        y, _, gain = e.wiener(dev(xin, dt), [n], SR, vad_guided=vad, want_gain=True)
        want, gain_or = orc.wiener(xin.astype(np.float64), SR, vad_guided=vad)
        if dt == torch.float64:
            check64(host(y), want)
            check64(host(gain)[0], gain_or, 1e-12)
        else:
            check32(host(y), want)
    base = host(e.noise_psd(dev(xin, dt), [n]))[0]
    base_or = orc.noise_profile(xin.astype(np.float64))[0]
    if dt == torch.float64:
        check64(base, base_or, 1e-12)
    else:
        assert np.max(np.abs(base - base_or)) <= 2e-6 * max(1.0, float(np.max(np.abs(base_or))))


@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
@pytest.mark.parametrize("kernel_bins,update_ms", [(3, 10), (7, 30), (9, 20), (1, 10)])
def test_wiener_other_median_kernels_and_vad_frames(dt, kernel_bins, update_ms):
    """wiener.median_kernel_bins and wiener.update_ms are read from the configuration
    (src/wiener_filter.py:52-53): running medians of 1, 3, 7 and 9 bins, voice-activity frames of 10 ms
    (480 samples: shorter than a hop, the energies take their own pass), 20 and 30 ms, against the oracle."""
    e = engine()
    n = 6 * SR + 321
    x = synth.noisy_speech(n, SR, 42, 1200 + kernel_bins)
    if dt == torch.float32:
        x = x.astype(np.float32).astype(np.float64)
    for vad in (True, False):
        want, g_or = orc.wiener(x, SR, update_ms=update_ms, kernel_bins=kernel_bins, vad_guided=vad)
        y, _, gain = e.wiener(dev(x, dt), [n], SR, update_ms=update_ms, kernel_bins=kernel_bins, vad_guided=vad,
                              want_gain=True)
        (check64 if dt == torch.float64 else check32)(host(y), want)
        assert np.max(np.abs(host(gain)[0].astype(np.float64) - g_or)) <= (1e-10 if dt == torch.float64 else 2e-4)

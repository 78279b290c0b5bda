"""Pins the CPU oracle (oracle/adn_oracle.py) against outputs of the reference itself
(tests/golden/*.npz, produced by oracle/gen_golden.py).  fp64 tolerance 1e-9 max-abs
(in practice <= 1e-12)."""
import json

import numpy as np
import pytest

import adn_oracle as orc
import synth

TOL = 1e-9


def _meta(g, key="meta_json"):
    return json.loads(str(g[key]))


def _close(a, b, tol=TOL):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    m = ~np.isnan(a)
    if m.any():
        assert np.max(np.abs(a[m] - b[m])) <= tol, np.max(np.abs(a[m] - b[m]))


def test_wavedec_waverec_periodization(golden):
    g = golden("dwt_primitives")
    for i, m in enumerate(_meta(g, "per_meta_json")):
        x = g["per%d_x" % i]
        co = orc.wavedec(x, m["wavelet"], m["level"], "per")
        assert len(co) == m["bands"]
        assert [c.shape[-1] for c in co] == orc.band_lengths(m["n"], m["level"], m["wavelet"])
        for b, c in enumerate(co):
            _close(c, g["per%d_c%d" % (i, b)], 1e-12)
        _close(orc.waverec(co, m["wavelet"], "per"), g["per%d_rec" % i], 1e-12)


def test_symmetric_mode_and_wavelet_packet(golden):
    g = golden("dwt_primitives")
    for i, m in enumerate(_meta(g, "sym_meta_json")):
        ca, cd = orc.dwt_sym(g["sym%d_x" % i], m["wavelet"])
        _close(ca, g["sym%d_cA" % i], 1e-12)
        _close(cd, g["sym%d_cD" % i], 1e-12)
        _close(orc.idwt_sym(ca, cd, m["wavelet"]), g["sym%d_rec" % i], 1e-12)
    # WaveletPacket(data, 'db8', 'symmetric', 4) leaves in 'freq' order
    paths = json.loads(str(g["wp_paths_freq_json"]))
    leaves = g["wp_leaves_freq"]
    for p, ref in zip(paths, leaves):
        a = g["wp_x"]
        for ch in p:
            ca, cd = orc.dwt_sym(a, "db8")
            a = ca if ch == "a" else cd
        _close(a, ref, 1e-12)


def test_threshold_semantics(golden):
    g = golden("dwt_primitives")
    c = g["thr_in"]
    _close(orc.threshold(c, 1.0, "soft"), g["thr_soft_1"], 0)
    _close(orc.threshold(c, 1.0, "hard"), g["thr_hard_1"], 0)
    _close(orc.threshold(c, 0.0, "soft"), g["thr_soft_0"], 0)
    assert np.isnan(g["thr_soft_zeros_0"]).all()
    assert np.isnan(orc.threshold(np.zeros(4), 0.0, "soft")).all()


def test_mad(golden):
    g = golden("denoise_files")
    assert abs(orc.mad(g["mad_a"]) - float(g["mad_a_val"])) <= 1e-15
    assert abs(orc.mad(g["mad_b"]) - float(g["mad_b_val"])) <= 1e-15


def _denoise_input(tag):
    if tag == "c1":
        return synth.noisy_speech(160000, 16000, 42, 0)
    if tag == "short11":
        return synth.noisy_speech(20007, 8000, 42, 3)
    if tag == "oddblk":
        return synth.noisy_speech(20011, 8000, 42, 4)
    if tag == "stereo":
        return synth.stereo_noisy_speech(20000, 8000, 42, 5)
    if tag == "quad":
        return np.concatenate([synth.stereo_noisy_speech(5000, 8000, 42, 6),
                               synth.stereo_noisy_speech(5000, 8000, 42, 7)], axis=1)
    if tag == "silence":
        xs = synth.noisy_speech(20000, 8000, 42, 8)
        xs[0:2000] = 0.0
        xs[6000:8000] = 0.0
        return xs
    raise KeyError(tag)


@pytest.mark.parametrize("tag", ["c1", "short11", "oddblk", "stereo", "quad", "silence"])
def test_denoise_files(golden, tag):
    g = golden("denoise_files")
    m = {e["tag"]: e for e in _meta(g)}[tag]
    x = _denoise_input(tag)
    assert synth.bits_checksum(x) == m["x_checksum"], "synthetic generator is not bit-reproducible"
    y, sig, thr = orc.denoise_signal(x, m["sr"])
    assert list(y.shape) == m["out_shape"]
    _close(y, g[tag + "_y"])
    if tag == "silence":
        assert np.isnan(y[:2000]).all() and np.isnan(y[6000:8000]).all()
        assert not np.isnan(y[2000:6000]).any()


def test_block_stats(golden):
    g = golden("block_stats")
    for m in _meta(g):
        x = synth.noisy_speech(m["n"], m["sr"], 42, m["stream"])
        assert synth.bits_checksum(x) == m["x_checksum"]
        clean, sigma, thresh = orc.denoise_block(x, m["wavelet"], m["level"], m["thr_mode"])
        t = m["tag"]
        assert abs(sigma - float(g[t + "_sigma"])) <= 1e-14
        assert abs(thresh - float(g[t + "_thresh"])) <= 1e-13
        _close(clean[::m["stride"]], g[t + "_clean"])
        assert abs(clean.sum() - float(g[t + "_sum"])) <= 1e-8
        assert abs((clean ** 2).sum() - float(g[t + "_sumsq"])) <= 1e-8


def test_noise_profile_stage(golden):
    g = golden("stages")
    noise = synth.white_noise(240000, 42, 1000)
    base, T = orc.noise_profile(noise)
    assert [base.shape[0], T] == list(g["p1_shape"])
    ref = g["p1_baseline_col0"]
    assert ref.dtype == np.float32
    # float32 cast of a float64 value that differs in the last bits: allow one float32 ulp
    assert np.max(np.abs(base.astype(np.float32) - ref) / ref) <= 1.2e-7
    f, t = orc.stft_axes(240000, 48000, 2048, 512)
    _close(f, g["p1_frequencies"], 1e-9)
    _close(t, g["p1_times"], 1e-12)
    base1024, T1024 = orc.noise_profile(noise, n_fft=1024, hop=256, win_length=1024)
    assert [base1024.shape[0], T1024] == list(g["p1_1024_shape"])
    assert np.max(np.abs(base1024.astype(np.float32) - g["p1_1024_baseline_col0"]) / g["p1_1024_baseline_col0"]) <= 1.2e-7


def test_spectral_subtract_and_wiener_stages(golden):
    g = golden("stages")
    sr = 48000
    noise_mag = g["p1_baseline_col0"].astype(np.float64)     # median over a time-constant matrix
    x = synth.noisy_speech(60000, sr, 42, 20)
    y2b = orc.spectral_subtract(x, noise_mag)
    _close(y2b, g["p2b_y"])
    y2c, _ = orc.wiener(y2b, sr)
    _close(y2c, g["p2c_y"])
    _close(orc.wiener(x, sr, vad_guided=False)[0], g["p2c_novad_y"])
    _close(orc.wiener(x, sr, vad_guided=True)[0], g["p2c_vad_x_y"])
    xq = 1e-4 * synth.noisy_speech(30000, sr, 42, 21)
    mask, _ = orc.energy_vad(xq, sr, frame_ms=20)
    assert not mask.any()
    _close(orc.wiener(xq, sr)[0], g["p2c_quiet_y"])


def test_spectral_subtract_edges(golden):
    g = golden("stages")
    sr = 48000
    noise_mag = g["p1_baseline_col0"].astype(np.float64)
    for tag, n in (("n2048", 2048), ("n2100", 2100), ("n5000", 5000)):
        xe = synth.noisy_speech(n, sr, 42, 22)
        y = orc.spectral_subtract(xe, noise_mag)
        assert y.shape == g["p2b_%s_y" % tag].shape
        _close(y, g["p2b_%s_y" % tag])
    assert g["p2b_n2048_y"].shape == (0,)
    xst = synth.stereo_noisy_speech(30000, sr, 42, 23)
    _close(orc.spectral_subtract(xst, noise_mag), g["p2b_stereo_y"])
    # profile learnt at another n_fft -> np.interp resize
    _close(orc.spectral_subtract(synth.noisy_speech(5000, sr, 42, 22),
                                 g["p1_1024_baseline_col0"].astype(np.float64)), g["p2b_resize_y"])
    meta = _meta(g)
    assert "window is longer than input signal" in meta["short_input_error"]
    with pytest.raises(ValueError, match="window is longer"):
        orc.spectral_subtract(synth.noisy_speech(1000, sr, 42, 24), noise_mag)


def test_core_pipeline_chain(golden):
    g = golden("pipeline")
    meta = _meta(g)
    sr = 48000
    x = synth.stereo_noisy_speech(40000, sr, 42, 30)
    noise = synth.white_noise(48000, 42, 1001)
    assert synth.bits_checksum(x) == meta["x_checksum"]
    assert synth.bits_checksum(noise) == meta["noise_checksum"]
    base, T = orc.noise_profile(noise)
    assert [base.shape[0], T] == list(g["baseline_shape"])
    b32 = base.astype(np.float32)
    assert np.max(np.abs(b32 - g["baseline_col0"]) / g["baseline_col0"]) <= 1.2e-7
    y2, _, _ = orc.denoise_signal(x, sr)
    _close(y2, g["wavelet_y"])
    y2b = orc.spectral_subtract(y2, g["baseline_col0"].astype(np.float64))
    _close(y2b, g["ss_y"])
    y2c, _ = orc.wiener(y2b, sr)
    _close(y2c, g["wiener_y"])
    assert meta["outputs"] == ["phase1_noise_profile.npz", "phase2_wavelet.wav",
                               "phase2b_spectral_subtract.wav", "phase2c_wiener.wav"]


def _profiler_input():
    sr = 44100
    n = 3 * sr + 1234
    x = synth.white_noise(n, 42, 40, sigma=0.01)
    burst = synth.noisy_speech(n, sr, 42, 41, sigma=0.0)
    t = np.arange(n) / sr
    gate = ((t < 0.25) | ((t > 0.4) & (t < 0.9)) | ((t > 1.6) & (t < 2.2)) | (t > 2.95)).astype(np.float64)
    return x + 2.0 * gate * (burst + 0.2 * synth.white_noise(n, 42, 42, sigma=1.0))


def test_noise_profiler(golden):
    g = golden("noise_profiler")
    x = _profiler_input()
    assert str(synth.bits_checksum(x)) == str(g["x_checksum"])
    pred, thr, is_noise = orc.noise_profiler_predict(x)
    _close(orc.window_rms(x, 4410), g["rms"], 1e-15)
    assert abs(thr - float(g["threshold"])) <= 1e-15
    assert np.array_equal(is_noise, g["is_noise"])
    assert pred.shape[0] == int(g["pred_len"])
    _close(pred[::5], g["pred_strided"], 0)
    xs = np.stack([x, 0.5 * x[::-1]], axis=1)
    ps, _, _ = orc.noise_profiler_predict(xs)
    assert list(ps.shape) == list(g["stereo_pred_shape"])
    _close(ps[::5], g["stereo_pred_strided"], 0)


def test_analysis_report_matches_reference(golden):
    """src/analyze_audio.py run by the fixture generator vs the restatement: band summaries (dB),
    frame-wise mean noise magnitude, VAD frame count."""
    g = golden("analysis")
    m = json.loads(str(g["meta_json"]))
    ref = json.loads(str(g["report_json"]))
    x = synth.stereo_noisy_speech(m["n"], m["sr"], 42, m["stream"])
    noise = synth.white_noise(m["noise_n"], seed=42, stream=m["noise_stream"])
    assert synth.bits_checksum(x) == int(g["x_bits"]) and synth.bits_checksum(noise) == int(g["noise_bits"])
    got = orc.analysis_report(x, m["sr"], noise, m["sr"], m["bands"], m["n_fft"], m["hop"], m["padded"])
    assert sorted(got.keys()) == m["report_keys"]
    assert got["vad_frames"] == ref["vad_frames"] and got["seed"] == ref["seed"]
    for key in ("child_band_summary_interview_db", "child_band_summary_noise_db"):
        for band, vals in ref[key].items():
            for k, v in vals.items():
                assert abs(got[key][band][k] - v) <= 1e-9, (key, band, k)
    a, b = np.array(got["noise_profile_time_mean_mag"]), np.array(ref["noise_profile_time_mean_mag"])
    assert a.shape == b.shape and np.max(np.abs(a - b)) <= 1e-12


def _hpss_kwargs(hp):
    mb = hp.get("multiband", {})
    return dict(margin_h=hp["margin_harmonic"], margin_p=hp["margin_percussive"], kernel_size=hp["kernel_size"],
                power=hp["power"], mask_type=hp["mask_type"], preserve_transients=hp["preserve_transients"],
                bands=mb.get("bands") if mb.get("enabled", False) else None)


@pytest.mark.parametrize("tag", ["default", "multiband_hard", "binary_scaled"])
def test_hpss_matches_reference(golden, tag):
    """src/hpss_separate.py (median filtering HPSS, masks, transient preservation, normalisation)."""
    g = golden("hpss")
    m = json.loads(str(g["meta_json"]))
    c = m["cases"][tag]
    yh, yp, met = orc.hpss(g["x"], m["sr"], **_hpss_kwargs(c["hpss"]))
    assert yh.shape == g[tag + "_yh"].shape and yp.shape == g[tag + "_yp"].shape
    assert np.max(np.abs(yh - g[tag + "_yh"])) <= 1e-12
    assert np.max(np.abs(yp - g[tag + "_yp"])) <= 1e-12
    for k, v in c["energy"].items():
        assert abs(met[k] - v) <= 1e-12, k


def test_optional_denoiser_modes(golden):
    """Symmetric extension and the per-channel time-axis transform, against PyWavelets itself
    (tests/golden/denoise_modes.npz, written by oracle/gen_golden_modes.py)."""
    g = golden("denoise_modes")
    for i, m in enumerate(_meta(g, "sym_meta_json")):
        y, s, t = orc.denoise_block(g["sym%d_x" % i], m["wavelet"], m["level"], m["thr"], mode="sym")
        _close(y, g["sym%d_y" % i], 1e-12)
        assert abs(s - m["sigma"]) <= 1e-15 and abs(t - m["thresh"]) <= 1e-14
    for i, m in enumerate(_meta(g, "time_meta_json")):
        x = g["time%d_x" % i]
        for ch in range(2):
            y, s, _ = orc.denoise_block(x[:, ch], m["wavelet"], m["level"])
            _close(y, g["time%d_y" % i][:, ch], 1e-12)
            assert abs(s - m["sigma"][ch]) <= 1e-15

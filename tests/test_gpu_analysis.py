"""GPU parity of the phase-1 analysis caller (src/analyze_audio.py): band summaries, frame-wise mean
noise magnitude and the stage command line, against the report the reference produced (golden
fixture) and against the oracle.  fp64: 1e-9; fp32: dB values within 1e-3 dB."""
import json
import os

import numpy as np
import pytest
import torch

import adn_oracle as orc
import synth
from gpu_util import engine, dev, host

pytestmark = pytest.mark.gpu


def _inputs(g):
    m = json.loads(str(g["meta_json"]))
    x = synth.stereo_noisy_speech(m["n"], m["sr"], 42, m["stream"])
    noise = synth.white_noise(m["noise_n"], seed=42, stream=m["noise_stream"])
    return m, x, noise


def test_stage_command_line_matches_reference_report(golden, tmp_path, monkeypatch):
    import yaml
    from adn_hip import stages, wavio
    monkeypatch.setenv("ADN_WAV_SUBTYPE", "DOUBLE")
    monkeypatch.setenv("ADN_DTYPE", "f64")
    g = golden("analysis")
    m, x, noise = _inputs(g)
    ref = json.loads(str(g["report_json"]))
    tmp = str(tmp_path)
    px, pn = os.path.join(tmp, "x.wav"), os.path.join(tmp, "noise.wav")
    wavio.write(px, x, m["sr"])
    wavio.write(pn, noise, m["sr"])
    cfg = {"seed": 42,
           "paths": {"input": px, "noise": pn, "outputs_dir": os.path.join(tmp, "out"),
                     "logs_dir": os.path.join(tmp, "logs"), "reports_dir": os.path.join(tmp, "reports")},
           "analysis": {"samplerate": 44100,
                        "stft": {"n_fft": 2048, "hop_length": 512, "win_length": 2048, "window": "hann", "center": True},
                        "spectrogram": {"cmap": "magma", "dynamic_range_db": 80},
                        "child_bands_hz": m["bands"],
                        "vad": {"enabled": True, "frame_ms": 30, "energy_threshold_db": -40, "hangover_frames": 5}}}
    pc = os.path.join(tmp, "cfg.yaml")
    with open(pc, "w") as fh:
        yaml.safe_dump(cfg, fh)
    stages.analyze_audio_main(["--config", pc])
    got = json.load(open(os.path.join(tmp, "logs", "phase1_analysis.json")))
    assert sorted(got.keys()) == m["report_keys"]
    for k in ("seed", "samplerate_interview", "samplerate_noise", "vad_frames"):
        assert got[k] == ref[k], k
    for key in ("child_band_summary_interview_db", "child_band_summary_noise_db"):
        assert list(got[key].keys()) == list(ref[key].keys())
        for band, vals in ref[key].items():
            for k, v in vals.items():
                assert abs(got[key][band][k] - v) <= 1e-9, (key, band, k, got[key][band][k], v)
    a, b = np.array(got["noise_profile_time_mean_mag"]), np.array(ref["noise_profile_time_mean_mag"])
    assert a.shape == b.shape and np.max(np.abs(a - b)) <= 1e-9
    try:
        import matplotlib  # noqa: F401
        assert sorted(os.listdir(os.path.join(tmp, "reports"))) == m["plots"]
    except ImportError:
        pass


@pytest.mark.parametrize("dt,tol_db,tol_mag", [(torch.float64, 1e-9, 1e-9), (torch.float32, 1e-3, 1e-5)])
def test_band_summary_batch_vs_oracle(dt, tol_db, tol_mag):
    """Two signals of different length in one call (mono), odd band edges, an empty band, a band that
    is long enough for the multi-workgroup selection path, unpadded framing."""
    e = engine()
    sr, n_fft, hop = 48000, 2048, 512
    xs = [synth.noisy_speech(n, sr, 42, 70 + i) for i, n in enumerate((150000, 61234))]
    bands = [(300.0, 600.0), (1000.0, 3000.0), (0.0, 24000.0), (10.0, 20.0), (2343.75, 2343.75)]
    for padded in (True, False):
        x = dev(np.concatenate(xs), dt)
        mean, perc, tmean, mag = e.band_summary(x, [len(v) for v in xs], sr, bands, percentiles=(95.0, 50.0, 0.0, 100.0),
                                                n_fft=n_fft, hop=hop, padded=padded, want_mag=True)
        # (frames beyond a signal's own count are unspecified in d_mag / d_time_mean: only the valid part is widened)
        mean, perc, tmean, mag = host(mean), host(perc), host(tmean), host(mag)
        for s, v in enumerate(xs):
            v = v.astype(np.float32).astype(np.float64) if dt == torch.float32 else v
            Z = orc.stft(v, n_fft, hop, padded=padded)
            f = orc.stft_axes(len(v), sr, n_fft, hop, padded)[0]
            T = Z.shape[1]
            assert np.max(np.abs(mag[s, :, :T].astype(np.float64) - np.abs(Z))) <= tol_mag
            assert np.max(np.abs(tmean[s, :T].astype(np.float64) - np.mean(np.abs(Z), axis=0))) <= tol_mag
            for b, (lo, hi) in enumerate(bands):
                band = np.abs(Z)[(f >= lo) & (f <= hi), :]
                if band.size == 0:
                    assert np.isnan(mean[s, b]) and np.all(np.isnan(perc[:, s, b]))
                    continue
                assert abs(mean[s, b] - np.mean(band)) <= tol_mag
                db = 20 * np.log10(band + 1e-12)
                for p, q in enumerate((95.0, 50.0, 0.0, 100.0)):
                    want = np.percentile(db, q)
                    # near-zero magnitudes amplify fp32 rounding in dB: compare there in magnitude
                    if dt == torch.float32 and want < -80:
                        assert abs(10 ** (perc[p, s, b] / 20) - 10 ** (want / 20)) <= tol_mag
                    else:
                        assert abs(perc[p, s, b] - want) <= tol_db, (padded, s, b, q, perc[p, s, b], want)

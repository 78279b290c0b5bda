#!/opt/conda/bin/python3.9
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Run only in the build container (the reference tree never travels):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg /opt/conda/bin/python3.9 -W ignore oracle/gen_golden.py

What runs: the reference's own ``denoise.AudioDeNoise``, ``lib.noiseProfiler``
and the four stage scripts (``src/noise_profile.py``, ``src/wavelet_denoise.py``,
``src/spectral_subtract.py``, ``src/wiener_filter.py``) plus ``src/core_pipeline.py``,
imported/executed unmodified from /root/reference, with PyWavelets / SciPy /
NumPy as installed here.  ``soundfile`` is absent in every interpreter of this
container, so ``oracle/ref_harness/soundfile.py`` (a lossless float64 stand-in)
is placed first on sys.path.  Direct PyWavelets calls pin the transform
primitives the reference reaches through ``pywt.wavedec/waverec/WaveletPacket``.

Inputs come from ``oracle/synth.py`` (bit-reproducible, no libm) or from a
seeded numpy Generator (then stored in the fixture).  Only data is written to
tests/golden/: inputs, expected outputs, parameters, library versions.
"""
import json
import os
import runpy
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "ref_harness"))
sys.path.insert(1, REF)
sys.path.insert(2, HERE)
os.environ.setdefault("MPLBACKEND", "Agg")

import numpy as np          # noqa: E402
import pywt                 # noqa: E402
import scipy                # noqa: E402
import soundfile as sf      # noqa: E402  (the stand-in)
import yaml                 # noqa: E402

import synth                # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")
VERSIONS = {"pywt": pywt.__version__, "scipy": scipy.__version__, "numpy": np.__version__,
            "python": sys.version.split()[0]}


def save(name, **arrays):
    arrays["versions_json"] = np.array(json.dumps(VERSIONS))
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024.0))


def run_stage(script, argv):
    old = sys.argv
    sys.argv = [script] + argv
    try:
        runpy.run_path(os.path.join(REF, "src", script), run_name="__main__")
    finally:
        sys.argv = old


def base_config(tmp, inp, noise):
    return {
        "paths": {"input": inp, "noise": noise, "outputs_dir": os.path.join(tmp, "out"),
                  "logs_dir": os.path.join(tmp, "logs"), "reports_dir": os.path.join(tmp, "reports")},
        "analysis": {"stft": {"n_fft": 2048, "hop_length": 512, "win_length": 2048,
                              "window": "hann", "center": True}},
        "noise_profile": {"percentile_level": 95, "spectral_floor": 0.1,
                          "smoothing": {"freq_bins": 5, "time_frames": 5}},
        "spectral_subtraction": {"alpha": 2.2, "beta": 0.1, "n_fft": 2048, "hop_length": 512,
                                 "smoothing": {"time": 3, "freq": 5}},
        "wiener": {"update_ms": 20, "median_kernel_bins": 5, "vad_guided": True},
    }


# ----------------------------------------------------------------------------
def gen_dwt_primitives():
    rng = np.random.default_rng(1234)
    out = {}
    cases = [("db4", 2, n) for n in (1, 2, 3, 7, 8, 16, 30, 31, 64, 1001)] + \
            [("db8", 3, n) for n in (1, 5, 9, 37, 64, 1000)] + [("db8", 8, 1125 * 8), ("db8", 8, 2250)]
    meta = []
    for i, (w, lev, n) in enumerate(cases):
        x = rng.standard_normal(n)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            co = pywt.wavedec(x, w, mode="per", level=lev)
            rec = pywt.waverec(co, w, mode="per")
        out["per%d_x" % i] = x
        for b, c in enumerate(co):
            out["per%d_c%d" % (i, b)] = c
        out["per%d_rec" % i] = rec
        meta.append({"wavelet": w, "level": lev, "n": n, "bands": len(co)})
    out["per_meta_json"] = np.array(json.dumps(meta))
    # single-level symmetric analysis/synthesis + wavelet packet leaves (lib/windowBundle.py:33-34)
    smeta = []
    for i, (w, n) in enumerate([("db4", 37), ("db8", 37), ("db8", 4410), ("db8", 9)]):
        x = rng.standard_normal(n)
        ca, cd = pywt.dwt(x, w, "symmetric")
        out["sym%d_x" % i] = x
        out["sym%d_cA" % i] = ca
        out["sym%d_cD" % i] = cd
        out["sym%d_rec" % i] = pywt.idwt(ca, cd, w, "symmetric")
        smeta.append({"wavelet": w, "n": n})
    out["sym_meta_json"] = np.array(json.dumps(smeta))
    x = rng.standard_normal(4410)
    wp = pywt.WaveletPacket(x, "db8", "symmetric", 4)
    out["wp_x"] = x
    nat = [n.path for n in wp.get_level(4, "natural")]
    frq = [n.path for n in wp.get_level(4, "freq")]
    out["wp_paths_natural_json"] = np.array(json.dumps(nat))
    out["wp_paths_freq_json"] = np.array(json.dumps(frq))
    out["wp_leaves_freq"] = np.stack([wp[p].data for p in frq])
    # threshold semantics incl. 0/0
    c = np.array([-2.0, -1.0, -0.5, 0.0, 0.25, 1.0, 3.0])
    with np.errstate(all="ignore"):
        out["thr_in"] = c
        out["thr_soft_1"] = pywt.threshold(c, 1.0, "soft")
        out["thr_hard_1"] = pywt.threshold(c, 1.0, "hard")
        out["thr_soft_0"] = pywt.threshold(c, 0.0, "soft")
        out["thr_soft_zeros_0"] = pywt.threshold(np.zeros(4), 0.0, "soft")
    save("dwt_primitives", **out)


def gen_denoise_files(tmp):
    from denoise import AudioDeNoise, mad
    out = {}
    meta = []

    def one(tag, x, sr, keep_input=False):
        pin = os.path.join(tmp, tag + "_in.wav")
        pout = os.path.join(tmp, tag + "_out.wav")
        sf.write(pin, x, sr)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            AudioDeNoise(pin).deNoise(pout)
        y, sr2 = sf.read(pout)
        assert sr2 == sr
        out[tag + "_y"] = y
        if keep_input:
            out[tag + "_x"] = x
        meta.append({"tag": tag, "sr": sr, "shape": list(x.shape), "out_shape": list(y.shape),
                     "x_checksum": synth.bits_checksum(x)})

    # C1: 10 s 16 kHz mono, db4/per/L2/soft (BASELINE.json config 1)
    one("c1", synth.noisy_speech(160000, 16000, seed=42, stream=0), 16000)
    # 11 blocks: 10 x 2000 + one of 7 frames (odd -> 8 out)
    one("short11", synth.noisy_speech(20007, 8000, seed=42, stream=3), 8000)
    # odd block size 2001 (each block comes back as 2002) + a final block of ONE frame
    one("oddblk", synth.noisy_speech(20011, 8000, seed=42, stream=4), 8000)
    # stereo: the transform runs across the channel axis
    one("stereo", synth.stereo_noisy_speech(20000, 8000, seed=42, stream=5), 8000)
    # 4 channels
    x4 = np.concatenate([synth.stereo_noisy_speech(5000, 8000, seed=42, stream=6),
                         synth.stereo_noisy_speech(5000, 8000, seed=42, stream=7)], axis=1)
    one("quad", x4, 8000)
    # digital silence in blocks 0 and 3 -> NaN output there
    xs = synth.noisy_speech(20000, 8000, seed=42, stream=8)
    xs[0:2000] = 0.0
    xs[6000:8000] = 0.0
    one("silence", xs, 8000)
    # mad() itself
    rng = np.random.default_rng(7)
    a = rng.standard_normal(1001)
    b = rng.standard_normal(1000)
    out["mad_a"], out["mad_b"] = a, b
    out["mad_a_val"], out["mad_b_val"] = np.float64(mad(a)), np.float64(mad(b))
    out["meta_json"] = np.array(json.dumps(meta))
    save("denoise_files", **out)


def gen_block_stats():
    """Per-block sigma / threshold / bands straight from pywt for the block-level tests
    (same calls as denoise.py:77-89), incl. db8 level 8 on a 144000-sample block and hard mode."""
    import warnings
    from denoise import mad
    out = {}
    meta = []
    cases = [("b16000", synth.noisy_speech(16000, 16000, 42, 0), "db4", 2, "soft"),
             ("b16001", synth.noisy_speech(16001, 16000, 42, 11), "db4", 2, "soft"),
             ("b16003", synth.noisy_speech(16003, 16000, 42, 12), "db4", 2, "soft"),
             ("b16000h", synth.noisy_speech(16000, 16000, 42, 13), "db4", 2, "hard"),
             ("b144000", synth.noisy_speech(144000, 48000, 42, 14), "db8", 8, "soft"),
             ("b4096l3", synth.noisy_speech(4096, 48000, 42, 15), "db8", 3, "soft"),
             ("b288000", synth.noisy_speech(288000, 48000, 42, 16), "db4", 2, "soft")]
    for tag, x, w, lev, tm in cases:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            co = pywt.wavedec(x, w, mode="per", level=lev)
            sigma = mad(co[-1])
            thresh = sigma * np.sqrt(2 * np.log(len(x)))
            co[1:] = (pywt.threshold(i, value=thresh, mode=tm) for i in co[1:])
            clean = pywt.waverec(co, w, mode="per")
        stride = 1 if x.size <= 20000 else 9
        out[tag + "_clean"] = clean[::stride]
        out[tag + "_sigma"] = np.float64(sigma)
        out[tag + "_thresh"] = np.float64(thresh)
        out[tag + "_sum"] = np.float64(np.sum(clean))
        out[tag + "_sumsq"] = np.float64(np.sum(clean ** 2))
        meta.append({"tag": tag, "n": int(x.size), "wavelet": w, "level": lev, "thr_mode": tm,
                     "stride": stride, "sr": 16000 if "b16" in tag else 48000,
                     "stream": {"b16000": 0, "b16001": 11, "b16003": 12, "b16000h": 13,
                                "b144000": 14, "b4096l3": 15, "b288000": 16}[tag],
                     "x_checksum": synth.bits_checksum(x)})
    out["meta_json"] = np.array(json.dumps(meta))
    save("block_stats", **out)


def gen_stages(tmp):
    out = {}
    meta = {}
    sr = 48000
    noise = synth.white_noise(240000, seed=42, stream=1000)
    pn = os.path.join(tmp, "noise.wav")
    sf.write(pn, noise, sr)
    x = synth.noisy_speech(60000, sr, seed=42, stream=20)
    px = os.path.join(tmp, "x.wav")
    sf.write(px, x, sr)
    cfg = base_config(tmp, px, pn)
    pc = os.path.join(tmp, "cfg.yaml")
    with open(pc, "w") as fh:
        yaml.safe_dump(cfg, fh)
    odir, ldir = cfg["paths"]["outputs_dir"], cfg["paths"]["logs_dir"]

    # phase 1
    run_stage("noise_profile.py", ["--config", pc])
    with np.load(os.path.join(odir, "phase1_noise_profile.npz"), allow_pickle=True) as z:
        bm = z["baseline_mag"]
        assert bm.dtype == np.float32
        assert float(np.max(np.abs(bm - bm[:, :1]))) == 0.0
        out["p1_baseline_col0"] = bm[:, 0].copy()
        out["p1_shape"] = np.array(bm.shape)
        out["p1_times"] = z["times"]
        out["p1_frequencies"] = z["frequencies"]
        out["p1_samplerate"] = z["samplerate"]
        out["p1_percentile"] = z["percentile"]
        out["p1_spectral_floor"] = z["spectral_floor"]
        meta["p1_keys"] = sorted(z.files)
        meta["p1_dtypes"] = {k: str(z[k].dtype) for k in z.files}
    meta["p1_sidecar"] = json.load(open(os.path.join(ldir, "phase1_noise_profile.json")))

    # phase 2b on the raw input (phase2_wavelet.wav absent -> falls back to paths.input)
    run_stage("spectral_subtract.py", ["--config", pc])
    y2b, _ = sf.read(os.path.join(odir, "phase2b_spectral_subtract.wav"))
    out["p2b_y"] = y2b
    meta["p2b_sidecar"] = json.load(open(os.path.join(ldir, "phase2b_spectral_subtract.json")))

    # phase 2c on the 2b output (vad_guided = True)
    run_stage("wiener_filter.py", ["--config", pc])
    y2c, _ = sf.read(os.path.join(odir, "phase2c_wiener.wav"))
    out["p2c_y"] = y2c
    meta["p2c_sidecar"] = json.load(open(os.path.join(ldir, "phase2c_wiener.json")))

    # phase 2c directly on x, vad_guided false, into another file
    cfg2 = base_config(tmp, px, pn)
    cfg2["wiener"]["vad_guided"] = False
    pc2 = os.path.join(tmp, "cfg2.yaml")
    yaml.safe_dump(cfg2, open(pc2, "w"))
    run_stage("wiener_filter.py", ["--config", pc2, "--input", px, "--output", os.path.join(tmp, "w_novad.wav")])
    out["p2c_novad_y"] = sf.read(os.path.join(tmp, "w_novad.wav"))[0]
    # vad_guided true directly on x
    run_stage("wiener_filter.py", ["--config", pc, "--input", px, "--output", os.path.join(tmp, "w_vad.wav")])
    out["p2c_vad_x_y"] = sf.read(os.path.join(tmp, "w_vad.wav"))[0]
    # very quiet input: no VAD frame fires -> denom = max(0, 1)
    xq = 1e-4 * synth.noisy_speech(30000, sr, seed=42, stream=21)
    pq = os.path.join(tmp, "xq.wav")
    sf.write(pq, xq, sr)
    run_stage("wiener_filter.py", ["--config", pc, "--input", pq, "--output", os.path.join(tmp, "w_quiet.wav")])
    out["p2c_quiet_y"] = sf.read(os.path.join(tmp, "w_quiet.wav"))[0]

    # edge lengths for 2b: N == n_fft (T=1 -> empty output), N slightly longer (T=2), stereo input
    for tag, n in (("n2048", 2048), ("n2100", 2100), ("n5000", 5000)):
        xe = synth.noisy_speech(n, sr, seed=42, stream=22)
        pe = os.path.join(tmp, tag + ".wav")
        sf.write(pe, xe, sr)
        po = os.path.join(tmp, tag + "_2b.wav")
        run_stage("spectral_subtract.py", ["--config", pc, "--input", pe, "--output", po])
        out["p2b_%s_y" % tag] = sf.read(po)[0]
    xst = synth.stereo_noisy_speech(30000, sr, seed=42, stream=23)
    pst = os.path.join(tmp, "st.wav")
    sf.write(pst, xst, sr)
    run_stage("spectral_subtract.py", ["--config", pc, "--input", pst, "--output", os.path.join(tmp, "st_2b.wav")])
    out["p2b_stereo_y"] = sf.read(os.path.join(tmp, "st_2b.wav"))[0]
    # N < n_fft -> the stage must fail
    pshort = os.path.join(tmp, "short.wav")
    sf.write(pshort, synth.noisy_speech(1000, sr, seed=42, stream=24), sr)
    try:
        run_stage("spectral_subtract.py", ["--config", pc, "--input", pshort, "--output", os.path.join(tmp, "s.wav")])
        meta["short_input_error"] = None
    except Exception as e:  # noqa: BLE001
        meta["short_input_error"] = "%s: %s" % (type(e).__name__, e)
    # profile with a different bin count -> np.interp resize path (n_fft 1024 profile, 2048 stage)
    cfg3 = base_config(tmp, px, pn)
    cfg3["analysis"]["stft"].update({"n_fft": 1024, "win_length": 1024, "hop_length": 256})
    pc3 = os.path.join(tmp, "cfg3.yaml")
    yaml.safe_dump(cfg3, open(pc3, "w"))
    run_stage("noise_profile.py", ["--config", pc3, "--output", os.path.join(tmp, "prof1024.npz")])
    with np.load(os.path.join(tmp, "prof1024.npz"), allow_pickle=True) as z:
        out["p1_1024_baseline_col0"] = z["baseline_mag"][:, 0].copy()
        out["p1_1024_shape"] = np.array(z["baseline_mag"].shape)
    run_stage("spectral_subtract.py", ["--config", pc, "--input", os.path.join(tmp, "n5000.wav"),
                                       "--noise_profile", os.path.join(tmp, "prof1024.npz"),
                                       "--output", os.path.join(tmp, "resz.wav")])
    out["p2b_resize_y"] = sf.read(os.path.join(tmp, "resz.wav"))[0]
    out["meta_json"] = np.array(json.dumps(meta))
    save("stages", **out)


def gen_pipeline(tmp):
    """src/core_pipeline.py end to end (4 subprocesses) on a short stereo clip + noise sample."""
    sr = 48000
    x = synth.stereo_noisy_speech(40000, sr, seed=42, stream=30)
    noise = synth.white_noise(48000, seed=42, stream=1001)
    px, pn = os.path.join(tmp, "pipe_in.wav"), os.path.join(tmp, "pipe_noise.wav")
    sf.write(px, x, sr)
    sf.write(pn, noise, sr)
    ptmp = os.path.join(tmp, "pipe")
    os.makedirs(ptmp)
    cfg = base_config(ptmp, "/nonexistent/in.wav", "/nonexistent/noise.wav")
    cfg["paths"]["noise_sample"] = cfg["paths"].pop("noise")
    pc = os.path.join(ptmp, "core.yaml")
    yaml.safe_dump(cfg, open(pc, "w"))
    env = dict(os.environ)
    env["PYTHONPATH"] = os.path.join(HERE, "ref_harness")
    env["PYTHONDONTWRITEBYTECODE"] = "1"
    env["MPLBACKEND"] = "Agg"
    r = subprocess.run([sys.executable, "-W", "ignore", os.path.join(REF, "src", "core_pipeline.py"),
                        "--config", pc, "--input", px, "--noise", pn],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, cwd=ptmp)
    print(r.stdout[-600:])
    assert r.returncode == 0
    odir, ldir = cfg["paths"]["outputs_dir"], cfg["paths"]["logs_dir"]
    out = {}
    out["wavelet_y"] = sf.read(os.path.join(odir, "phase2_wavelet.wav"))[0]
    out["ss_y"] = sf.read(os.path.join(odir, "phase2b_spectral_subtract.wav"))[0]
    out["wiener_y"] = sf.read(os.path.join(odir, "phase2c_wiener.wav"))[0]
    with np.load(os.path.join(odir, "phase1_noise_profile.npz"), allow_pickle=True) as z:
        out["baseline_col0"] = z["baseline_mag"][:, 0].copy()
        out["baseline_shape"] = np.array(z["baseline_mag"].shape)
    meta = {"logs": sorted(os.listdir(ldir)), "outputs": sorted(os.listdir(odir)),
            "x_checksum": synth.bits_checksum(x), "noise_checksum": synth.bits_checksum(noise),
            "wavelet_sidecar": json.load(open(os.path.join(ldir, "phase2_wavelet.json"))),
            "stdout_log_tail": open(os.path.join(ldir, "core_phase2c_wiener_stdout.log")).read()[-200:]}
    out["meta_json"] = np.array(json.dumps(meta))
    save("pipeline", **out)


def gen_noise_profiler(tmp):
    from denoise import AudioDeNoise
    from lib.noiseProfiler import NoiseProfiler
    sr = 44100
    n = 3 * sr + 1234
    x = synth.white_noise(n, seed=42, stream=40, sigma=0.01)
    # loud bursts over 0.4-0.9 s and 1.6-2.2 s, and the first 0.25 s loud too (leading gap)
    burst = synth.noisy_speech(n, sr, seed=42, stream=41, sigma=0.0)
    t = np.arange(n) / sr
    gate = ((t < 0.25) | ((t > 0.4) & (t < 0.9)) | ((t > 1.6) & (t < 2.2)) | (t > 2.95)).astype(np.float64)
    x = x + 2.0 * gate * (burst + 0.2 * synth.white_noise(n, 42, 42, sigma=1.0))
    prof = NoiseProfiler(x)
    thr = prof.extractRMSthresholdFromWindows(prof.percentileLevel)
    rms = np.array([w.getRMS() for w in prof.windows])
    pred = np.array(prof.getNoiseDataPredicted())
    out = {"x_checksum": np.array(str(synth.bits_checksum(x))), "threshold": np.float64(thr), "rms": rms,
           "pred_len": np.int64(pred.shape[0]), "pred_strided": pred[::5].copy(),
           "pred_sum": np.float64(pred.sum())}
    # window classes must be re-derived (cleanUp() dropped the lists): same rule as the reference
    out["is_noise"] = rms < (thr + 0.05 * thr)
    # stereo variant through AudioDeNoise.generateNoiseProfile (reads + overwrites the file)
    xs = np.stack([x, 0.5 * x[::-1]], axis=1)
    p = os.path.join(tmp, "np_stereo.wav")
    sf.write(p, xs, sr)
    AudioDeNoise("unused.wav").generateNoiseProfile(p)
    ys, _ = sf.read(p)
    out["stereo_pred_shape"] = np.array(ys.shape)
    out["stereo_pred_strided"] = ys[::5].copy()
    save("noise_profiler", **out)


def gen_analysis(tmp):
    """src/analyze_audio.py run unmodified (plots go to a temp dir, matplotlib Agg): band summaries,
    frame-wise mean noise magnitude, VAD frame count, for a stereo input and a mono noise sample."""
    sr = 16000
    x = synth.stereo_noisy_speech(48000, sr, 42, 60)
    noise = synth.white_noise(24000, seed=42, stream=1001)
    px, pn = os.path.join(tmp, "ana_x.wav"), os.path.join(tmp, "ana_noise.wav")
    sf.write(px, x, sr)
    sf.write(pn, noise, sr)
    cfg = base_config(os.path.join(tmp, "ana"), px, pn)
    bands = {"crying": [300, 600], "screaming": [1000, 3000], "harmonics": [3000, 4000]}
    cfg["seed"] = 42
    cfg["analysis"].update({"samplerate": 44100, "spectrogram": {"cmap": "magma", "dynamic_range_db": 80},
                            "child_bands_hz": bands,
                            "vad": {"enabled": True, "frame_ms": 30, "energy_threshold_db": -40, "hangover_frames": 5}})
    pc = os.path.join(tmp, "ana_cfg.yaml")
    with open(pc, "w") as fh:
        yaml.safe_dump(cfg, fh)
    run_stage("analyze_audio.py", ["--config", pc])
    report = json.load(open(os.path.join(cfg["paths"]["logs_dir"], "phase1_analysis.json")))
    plots = sorted(os.listdir(cfg["paths"]["reports_dir"]))
    meta = {"sr": sr, "n": 48000, "stream": 60, "noise_n": 24000, "noise_stream": 1001, "bands": bands,
            "n_fft": 2048, "hop": 512, "padded": True, "frame_ms": 30, "plots": plots,
            "report_keys": sorted(report.keys())}
    save("analysis", report_json=np.array(json.dumps(report)), meta_json=np.array(json.dumps(meta)),
         x_bits=np.array(synth.bits_checksum(x)), noise_bits=np.array(synth.bits_checksum(noise)),
         noise_profile_time=np.array(report["noise_profile_time_mean_mag"]))


def gen_hpss(tmp):
    """src/hpss_separate.py run unmodified (scipy.ndimage.median_filter path; librosa is absent):
    default median / soft / transient preservation, and multiband + hard masks without it."""
    sr, n = 16000, 40000
    x = synth.noisy_speech(n, sr, 42, 80)
    # clicks: the transient detector and the percussive mask get something to do
    x = x.copy()
    x[9000:9040] += 0.9
    x[25000:25030] -= 0.8
    px = os.path.join(tmp, "hp_x.wav")
    sf.write(px, x, sr)
    out = {"x": x}
    meta = {"sr": sr, "n": n, "stream": 80, "cases": {}}
    cases = {"default": {"method": "median", "margin_harmonic": 1.0, "margin_percussive": 5.0, "kernel_size": 31,
                         "power": 2.0, "mask_type": "soft", "preserve_transients": True, "reversible": True},
             "multiband_hard": {"method": "median", "margin_harmonic": 1.0, "margin_percussive": 2.0, "kernel_size": 17,
                                "power": 2.0, "mask_type": "hard", "preserve_transients": False, "reversible": True,
                                "multiband": {"enabled": True, "bands": [[0, 250], [250, 2000], [2000, 7000]]}},
             "binary_scaled": {"method": "median", "margin_harmonic": 3.0, "margin_percussive": 1.0, "kernel_size": 31,
                               "power": 1.0, "mask_type": "binary", "preserve_transients": True, "reversible": True}}
    for tag, hp in cases.items():
        d = os.path.join(tmp, "hp_" + tag)
        cfg = base_config(d, px, px)
        cfg["hpss"] = hp
        pc = os.path.join(tmp, "hp_%s.yaml" % tag)
        with open(pc, "w") as fh:
            yaml.safe_dump(cfg, fh)
        run_stage("hpss_separate.py", ["--config", pc])
        odir = cfg["paths"]["outputs_dir"]
        yh, _ = sf.read(os.path.join(odir, "phase3_harmonic.wav"))
        yp, _ = sf.read(os.path.join(odir, "phase3_percussive.wav"))
        yc, _ = sf.read(os.path.join(odir, "phase3_hpss.wav"))
        assert np.array_equal(yc, yh)
        log = json.load(open(os.path.join(cfg["paths"]["logs_dir"], "phase3_hpss.json")))
        out[tag + "_yh"], out[tag + "_yp"] = yh, yp
        meta["cases"][tag] = {"hpss": hp, "log_keys": sorted(log.keys()), "parameters": log["parameters"],
                              "energy": log["metrics"]["energy_distribution"], "outputs": sorted(os.listdir(odir))}
    save("hpss", meta_json=np.array(json.dumps(meta)), **out)


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    if len(sys.argv) > 2 and sys.argv[1] == "--only":
        with tempfile.TemporaryDirectory() as tmp:
            globals()["gen_" + sys.argv[2]](tmp)
        sys.exit(0)
    with tempfile.TemporaryDirectory() as tmp:
        gen_analysis(tmp)
        gen_hpss(tmp)
        gen_dwt_primitives()
        gen_denoise_files(tmp)
        gen_block_stats()
        gen_stages(tmp)
        gen_pipeline(tmp)
        gen_noise_profiler(tmp)
    print("versions:", VERSIONS)

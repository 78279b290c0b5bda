"""Host side of the four core stages and their orchestrator.

Same command lines, output files, sidecar JSON and final "Phase ... complete" lines as
the reference stage scripts (src/noise_profile.py, src/wavelet_denoise.py,
src/spectral_subtract.py, src/wiener_filter.py, src/core_pipeline.py); all arithmetic
runs in the HIP kernels behind ``adn_hip.Engine``.  Host work is file I/O, config
lookup and the few scalar decisions the reference makes on the host.

Environment knobs (defaults reproduce the reference):
  ADN_DTYPE        f64 (default) or f32 - arithmetic precision of the kernels
  ADN_DEVICE       GPU index (default 0)
  ADN_WAV_SUBTYPE  subtype of written WAV files (default PCM_16 like libsndfile)
"""
import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

from . import devfiles, wavio


def _torch_dtype():
    import torch
    name = os.environ.get("ADN_DTYPE", "f64").lower()
    if name not in ("f32", "f64"):
        raise ValueError("ADN_DTYPE must be f32 or f64")
    return torch.float32 if name == "f32" else torch.float64


def _engine():
    from .engine import get_engine
    return get_engine(int(os.environ.get("ADN_DEVICE", "0")))


def _to_device(a):
    import torch
    e = _engine()
    return torch.as_tensor(np.ascontiguousarray(a), device=e.torch_device).to(_torch_dtype())


def _pcm_buffer(e, frames, n_fft, hop):
    """int16 buffer for the fused PCM store of an STFT stage, None when the files are not 16-bit PCM."""
    import torch
    if wavio.default_subtype() != "PCM_16" or frames < n_fft:
        return None
    return torch.empty(e.istft_out_samples(frames, n_fft, hop), dtype=torch.int16, device=e.torch_device)


def _load_config(path):
    import yaml
    with open(path, "r") as fh:
        return yaml.safe_load(fh)


def _mkdirs(*dirs):
    for d in dirs:
        Path(d).mkdir(parents=True, exist_ok=True)


def _dump(path, obj):
    with open(path, "w") as fh:
        json.dump(obj, fh, indent=2)


# ------------------------------------------------------------------ phase 1
def noise_profile_main(argv=None):
    """src/noise_profile.py: learn the per-bin noise magnitude from the noise sample."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--noise", default=None)
    ap.add_argument("--output", default=None, help="Optional path to save learned spectral noise profile (npz)")
    args = ap.parse_args(argv)
    cfg = _load_config(args.config)
    paths = cfg.get("paths", {})
    key = "noise" if "noise" in paths else ("noise_sample" if "noise_sample" in paths else None)
    if key is None:
        raise KeyError("Config missing paths.noise or paths.noise_sample")
    noise_path = args.noise or paths[key]
    outputs_dir, logs_dir, reports_dir = paths["outputs_dir"], paths["logs_dir"], paths["reports_dir"]
    _mkdirs(outputs_dir, logs_dir, reports_dir)

    noise, sr = wavio.read(noise_path)
    n_ch = 1 if noise.ndim == 1 else noise.shape[1]
    st = cfg["analysis"]["stft"]
    npf = cfg["noise_profile"]
    perc, floor = npf["percentile_level"], npf["spectral_floor"]
    tf, ff = npf["smoothing"]["time_frames"], npf["smoothing"]["freq_bins"]
    win, hop, padded = int(st["win_length"]), int(st["hop_length"]), bool(st["center"])
    if str(st["window"]).lower() not in ("hann", "hanning"):
        raise NotImplementedError("only the Hann window of the reference configs is implemented")
    e = _engine()
    x = _to_device(noise).reshape(-1)
    base = e.noise_psd(x, [noise.shape[0]], n_ch=n_ch, n_fft=win, hop=hop, padded=padded, percentile=perc,
                       spectral_floor=floor, freq_bins=ff, time_frames=tf)[0]
    base = base.double().cpu().numpy()
    T = e.stft_num_frames(noise.shape[0], win, hop, padded)
    n_pad = (T - 1) * hop + win if padded else noise.shape[0]
    times = np.arange(win / 2, n_pad - win / 2 + 1, hop) / float(sr)
    freqs = np.fft.rfftfreq(win, 1.0 / sr)
    # the reference stores an F x T matrix that is constant along time
    baseline = np.repeat(base.astype(np.float32)[:, None], T, axis=1)
    out_npz = args.output or os.path.join(outputs_dir, "phase1_noise_profile.npz")
    np.savez_compressed(out_npz, samplerate=sr, frequencies=freqs, times=times, baseline_mag=baseline,
                        percentile=perc, spectral_floor=floor)
    _dump(os.path.join(logs_dir, "phase1_noise_profile.json"),
          {"noise_input": noise_path, "output_npz": out_npz, "percentile_level": perc, "spectral_floor": floor,
           "smoothing": {"time_frames": tf, "freq_bins": ff}})
    print(f"Phase 1 noise profile learned -> {out_npz}")


def _band_report(e, x, sr, st, bands, want_mag):
    """Band summaries of one file on the device -> ({band: {mean_db, p95_db, p50_db}}, time mean, |Z|)."""
    n_ch = 1 if x.ndim == 1 else x.shape[1]
    win, hop, padded = int(st["win_length"]), int(st["hop_length"]), bool(st["center"])
    names = list(bands.keys())
    res = e.band_summary(_to_device(x).reshape(-1), [x.shape[0]], sr, [tuple(bands[k]) for k in names],
                         percentiles=(95.0, 50.0), n_ch=n_ch, n_fft=win, hop=hop, padded=padded, want_mag=want_mag)
    mean, perc, tmean = res[0].cpu().numpy()[0], res[1].cpu().numpy()[:, 0, :], res[2].double().cpu().numpy()[0]
    summary = {}
    for i, name in enumerate(names):
        summary[name] = {"mean_db": float(20 * np.log10(mean[i] + 1e-12)), "p95_db": float(perc[0, i]),
                         "p50_db": float(perc[1, i])}
    mag = res[3].double().cpu().numpy()[0] if want_mag else None
    return summary, tmean, mag


def analyze_audio_main(argv=None):
    """src/analyze_audio.py (phase-1 analysis): band-energy summaries of the interview and the noise
    sample, frame-wise mean noise magnitude, VAD frame count -> logs/phase1_analysis.json; the two
    spectrogram plots are drawn on the host from the device's magnitude matrix when matplotlib is present."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--input", default=None)
    ap.add_argument("--noise", default=None)
    args = ap.parse_args(argv)
    cfg = _load_config(args.config)
    seed = cfg.get("seed", 42)
    paths = cfg.get("paths", {})
    input_path = args.input or cfg["paths"]["input"]
    key = "noise" if "noise" in paths else ("noise_sample" if "noise_sample" in paths else None)
    if key is None:
        raise KeyError("Config missing paths.noise or paths.noise_sample")
    noise_path = args.noise or paths[key]
    out_dir, rep_dir, logs_dir = cfg["paths"]["outputs_dir"], cfg["paths"]["reports_dir"], cfg["paths"]["logs_dir"]
    print(f"[DEBUG] CWD: {os.getcwd()}")
    print(f"[DEBUG] input_path: {input_path}")
    print(f"[DEBUG] noise_path: {noise_path}")
    _mkdirs(out_dir, rep_dir, logs_dir)
    ana = cfg["analysis"]
    sr_cfg, st, spec_cfg = ana.get("samplerate", None), ana["stft"], ana["spectrogram"]
    bands, vad_cfg = ana["child_bands_hz"], ana["vad"]
    if str(st["window"]).lower() not in ("hann", "hanning"):
        raise NotImplementedError("only the Hann window of the reference configs is implemented")
    x, sr_x = wavio.read(input_path)
    n, sr_n = wavio.read(noise_path)
    if sr_cfg and sr_x != sr_cfg:
        print(f"[WARN] Interview SR {sr_x} != config {sr_cfg}. Proceeding with {sr_x}.")
    if sr_x != sr_n:
        print(f"[WARN] Interview SR {sr_x} != noise SR {sr_n}. Proceeding independently.")
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except ImportError:                                    # the report does not depend on the plots
        plt = None
        print("[WARN] matplotlib not importable: spectrogram plots skipped.")
    e = _engine()
    sum_x, _, mag_x = _band_report(e, x, sr_x, st, bands, plt is not None)
    sum_n, tmean_n, mag_n = _band_report(e, n, sr_n, st, bands, plt is not None)
    vad_enabled = vad_cfg.get("enabled", True)
    frame_len = int(sr_x * vad_cfg.get("frame_ms", 30) / 1000)
    frame_len = frame_len if frame_len > 0 else 1
    if plt is not None:
        win, hop, padded = int(st["win_length"]), int(st["hop_length"]), bool(st["center"])
        for mag, sr, length, title, name in ((mag_x, sr_x, x.shape[0], "Interview Spectrogram", "phase1_interview_spectrogram.png"),
                                             (mag_n, sr_n, n.shape[0], "Noise Sample Spectrogram", "phase1_noise_spectrogram.png")):
            T = mag.shape[1]
            n_pad = (T - 1) * hop + win if padded else length
            t = np.arange(win / 2, n_pad - win / 2 + 1, hop) / float(sr)
            f = np.fft.rfftfreq(win, 1.0 / sr)
            fig = plt.figure(figsize=(10, 4))
            s_db = 20 * np.log10(mag + 1e-12)
            vmin = np.max(s_db) - spec_cfg["dynamic_range_db"]
            plt.pcolormesh(t, f, s_db, shading="gouraud", cmap=spec_cfg["cmap"], vmin=vmin, vmax=np.max(s_db))
            plt.colorbar(label="dB")
            plt.title(title)
            plt.xlabel("Time [s]")
            plt.ylabel("Freq [Hz]")
            fig.tight_layout()
            fig.savefig(os.path.join(rep_dir, name), dpi=150)
            plt.close(fig)
    report = {"seed": seed, "samplerate_interview": sr_x, "samplerate_noise": sr_n,
              "child_band_summary_interview_db": sum_x, "child_band_summary_noise_db": sum_n,
              "noise_profile_time_mean_mag": tmean_n.tolist(),
              "vad_frames": int(x.shape[0] / frame_len) if vad_enabled else 0}
    _dump(os.path.join(logs_dir, "phase1_analysis.json"), report)
    print("Phase 1 analysis complete. Plots in reports/plots and JSON in logs/.")


# ------------------------------------------------------------------ phase 2
def wavelet_denoise_main(argv=None):
    """src/wavelet_denoise.py: AudioDeNoise(input).deNoise(output) plus the sidecar."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--input", default=None)
    ap.add_argument("--output", default=None)
    args = ap.parse_args(argv)
    cfg = _load_config(args.config)
    input_path = args.input or cfg["paths"]["input"]
    outputs_dir, logs_dir = cfg["paths"]["outputs_dir"], cfg["paths"]["logs_dir"]
    _mkdirs(outputs_dir, logs_dir)
    out_path = args.output or os.path.join(outputs_dir, "phase2_wavelet.wav")

    from denoise import AudioDeNoise          # the drop-in class next to this package
    AudioDeNoise(inputFile=input_path).deNoise(outputFile=out_path)
    meta = {}
    try:
        inf = wavio.info(out_path)
        meta = {"samplerate": inf.samplerate, "channels": inf.channels, "format": str(inf.format)}
    except Exception as ex:  # noqa: BLE001
        meta["error"] = str(ex)
    _dump(os.path.join(logs_dir, "phase2_wavelet.json"),
          {"input": input_path, "output": out_path,
           "wavelet_config_note": "Uses repository defaults (db4, level=2, VisuShrink soft). "
                                  "Exposed level in config but current denoise.py fixes level.",
           "meta": meta})
    print(f"Phase 2 (Wavelet) complete -> {out_path}")


# ----------------------------------------------------------------- phase 2b
def load_noise_profile(npz_path):
    """Profile npz -> dict with the reference's key set (src/spectral_subtract.py:12-20)."""
    with np.load(npz_path, allow_pickle=False) as d:
        return {"frequencies": d["frequencies"], "baseline_mag": d["baseline_mag"],
                "samplerate": int(d["samplerate"]), "percentile": float(d["percentile"]),
                "spectral_floor": float(d["spectral_floor"])}


def spectral_subtract_main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--input", default=None, help="Input audio (defaults to Phase 2 output if present)")
    ap.add_argument("--noise_profile", default=None, help="npz from noise_profile.py")
    ap.add_argument("--output", default=None)
    args = ap.parse_args(argv)
    cfg = _load_config(args.config)
    outputs_dir, logs_dir = cfg["paths"]["outputs_dir"], cfg["paths"]["logs_dir"]
    _mkdirs(outputs_dir, logs_dir)
    wav2 = os.path.join(outputs_dir, "phase2_wavelet.wav")
    in_path = args.input or (wav2 if os.path.exists(wav2) else cfg["paths"]["input"])
    npz_default = os.path.join(outputs_dir, "phase1_noise_profile.npz")
    noise_npz = args.noise_profile or (npz_default if os.path.exists(npz_default) else None)
    if noise_npz is None:
        raise SystemExit("Noise profile npz not found. Run src/noise_profile.py first.")
    out_path = args.output or os.path.join(outputs_dir, "phase2b_spectral_subtract.wav")

    e = _engine()
    x, sr, n_ch, frames = devfiles.read_wav(in_path, e, _torch_dtype())
    sc = cfg["spectral_subtraction"]
    n_fft, hop = int(sc["n_fft"]), int(sc["hop_length"])
    prof = load_noise_profile(noise_npz)
    # per-bin median over the profile's time axis (float32 values), resized with np.interp
    # when the profile was learnt with another bin count (src/spectral_subtract.py:88-97)
    nf = np.median(prof["baseline_mag"], axis=1).astype(np.float64)
    F = n_fft // 2 + 1
    if nf.shape[0] != F:
        nf = np.interp(np.linspace(0, len(nf) - 1, F), np.arange(len(nf)), nf)
    pcm = _pcm_buffer(e, frames, n_fft, hop)
    y, lens = e.spectral_subtract(x, [frames], _to_device(nf), n_ch=n_ch, n_fft=n_fft,
                                  hop=hop, alpha=sc["alpha"], beta=sc["beta"],
                                  smooth_time=sc["smoothing"]["time"], smooth_freq=sc["smoothing"]["freq"], pcm16=pcm)
    devfiles.write_wav(out_path, y, sr, 1, e, pcm16=pcm)
    _dump(os.path.join(logs_dir, "phase2b_spectral_subtract.json"),
          {"input": in_path, "output": out_path, "noise_profile": noise_npz, "alpha": sc["alpha"],
           "beta": sc["beta"], "n_fft": n_fft, "hop_length": hop, "smoothing": sc["smoothing"]})
    print(f"Phase 2b (Spectral Subtraction) complete -> {out_path}")


# ----------------------------------------------------------------- phase 2c
def wiener_filter_main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--input", default=None, help="Input audio (defaults to spectral_subtract output if present)")
    ap.add_argument("--output", default=None)
    args = ap.parse_args(argv)
    cfg = _load_config(args.config)
    outputs_dir, logs_dir = cfg["paths"]["outputs_dir"], cfg["paths"]["logs_dir"]
    _mkdirs(outputs_dir, logs_dir)
    prefer = [os.path.join(outputs_dir, "phase2b_spectral_subtract.wav"),
              os.path.join(outputs_dir, "phase2_wavelet.wav")]
    in_path = args.input or next((p for p in prefer if os.path.exists(p)), cfg["paths"]["input"])
    out_path = args.output or os.path.join(outputs_dir, "phase2c_wiener.wav")

    e = _engine()
    x, sr, n_ch, frames = devfiles.read_wav(in_path, e, _torch_dtype())
    sc, wc = cfg["spectral_subtraction"], cfg["wiener"]
    n_fft, hop = int(sc["n_fft"]), int(sc["hop_length"])
    update_ms = wc.get("update_ms", 20)
    kernel_bins = wc.get("median_kernel_bins", 5)
    vad_guided = wc.get("vad_guided", True)
    pcm = _pcm_buffer(e, frames, n_fft, hop)
    y, _ = e.wiener(x, [frames], sr, n_ch=n_ch, n_fft=n_fft, hop=hop,
                    update_ms=update_ms, kernel_bins=kernel_bins, vad_guided=vad_guided, pcm16=pcm)
    devfiles.write_wav(out_path, y, sr, 1, e, pcm16=pcm)
    _dump(os.path.join(logs_dir, "phase2c_wiener.json"),
          {"input": in_path, "output": out_path, "n_fft": n_fft, "hop_length": hop,
           "median_kernel_bins": kernel_bins, "vad_guided": vad_guided, "attack_ms": 5, "release_ms": 50})
    print(f"Phase 2c (Wiener) complete -> {out_path}")


# ------------------------------------------------------------- orchestrator
# ------------------------------------------------------------------ phase 3
def hpss_separate_main(argv=None):
    """src/hpss_separate.py: harmonic / percussive separation by median filtering of the spectrogram."""
    ap = argparse.ArgumentParser(description="Phase 3: HPSS Separation")
    ap.add_argument("--config", required=True, help="Path to config file")
    ap.add_argument("--input", default=None, help="Input audio (defaults to Phase 2 output)")
    ap.add_argument("--output_harmonic", default=None)
    ap.add_argument("--output_percussive", default=None)
    ap.add_argument("--output_combined", default=None)
    args = ap.parse_args(argv)
    cfg = _load_config(args.config)
    if not cfg.get("phases", {}).get("phase3", {}).get("enabled", True):
        print("[INFO] Phase 3 (HPSS) is disabled in config, skipping")
        return
    seed = cfg.get("hpss", {}).get("seed", cfg.get("global", {}).get("seed", 42))
    outputs_dir, logs_dir = cfg["paths"]["outputs_dir"], cfg["paths"]["logs_dir"]
    _mkdirs(outputs_dir, logs_dir)
    prefer = [os.path.join(outputs_dir, "phase2c_wiener.wav"), os.path.join(outputs_dir, "phase2b_spectral_subtract.wav"),
              os.path.join(outputs_dir, "phase2_wavelet.wav"), cfg["paths"]["input"]]
    in_path = args.input or next((p for p in prefer if os.path.exists(p)), cfg["paths"]["input"])
    out_harmonic = args.output_harmonic or os.path.join(outputs_dir, "phase3_harmonic.wav")
    out_percussive = args.output_percussive or os.path.join(outputs_dir, "phase3_percussive.wav")
    out_combined = args.output_combined or os.path.join(outputs_dir, "phase3_hpss.wav")
    print(f"[INFO] Loading input from {in_path}")
    e = _engine()
    x, sr, n_ch, frames = devfiles.read_wav(in_path, e, _torch_dtype())
    hp = cfg.get("hpss", {})
    method = hp.get("method", "median")
    margin_h, margin_p = hp.get("margin_harmonic", 1.0), hp.get("margin_percussive", 5.0)
    kernel_size, power = hp.get("kernel_size", 31), hp.get("power", 2.0)
    mask_type = hp.get("mask_type", "soft")
    multiband = hp.get("multiband", {}).get("enabled", False)
    preserve, reversible = hp.get("preserve_transients", True), hp.get("reversible", True)
    t0 = time.time()
    print("[INFO] Computing STFT...")
    print(f"[INFO] Applying HPSS (method={method})...")
    bands = None
    if multiband:
        bands = hp.get("multiband", {}).get("bands", [[0, 250], [250, 2000], [2000, 8000], [8000, 22050]])
        power = 2.0                 # multiband_hpss never forwards `power` (src/hpss_separate.py:103-105)
    elif method == "librosa":
        print("[WARN] librosa not available, using fallback median filtering")
    yh, yp, _, stats = e.hpss(x, [frames], sr, n_ch=n_ch, kernel_size=kernel_size,
                              margin_h=margin_h, margin_p=margin_p, power=power,
                              mask_type=mask_type if mask_type in ("hard", "binary") else "soft", bands=bands,
                              preserve_transients=preserve)
    print("[INFO] Reconstructing audio...")
    if preserve:
        print("[INFO] Preserving transients...")
    st = stats.cpu().numpy()[0]
    print("[INFO] Saving outputs...")
    if reversible:
        devfiles.write_wav(out_harmonic, yh, sr, 1, e)
        devfiles.write_wav(out_percussive, yp, sr, 1, e)
        print(f"  - Harmonic: {out_harmonic}")
        print(f"  - Percussive: {out_percussive}")
    devfiles.write_wav(out_combined, yh, sr, 1, e)
    print(f"  - Combined: {out_combined}")
    dt = time.time() - t0
    energy_harmonic, energy_percussive, energy_total = float(st[0]), float(st[1]), float(st[2])
    _dump(os.path.join(logs_dir, "phase3_hpss.json"), {
        "phase": "3_hpss", "timestamp": time.strftime("%Y-%m-%d %H:%M:%S"), "seed": seed, "input": in_path,
        "outputs": {"harmonic": out_harmonic if reversible else None,
                    "percussive": out_percussive if reversible else None, "combined": out_combined},
        "parameters": {"method": method, "margin_harmonic": margin_h, "margin_percussive": margin_p,
                       "kernel_size": kernel_size, "power": hp.get("power", 2.0), "mask_type": mask_type,
                       "multiband": multiband, "preserve_transients": preserve},
        "metrics": {"processing_time_sec": dt,
                    "energy_distribution": {"harmonic_rms": energy_harmonic, "percussive_rms": energy_percussive,
                                            "total_rms": energy_total,
                                            "harmonic_ratio": float(energy_harmonic / (energy_total + 1e-10))}}})
    print(f"[INFO] Phase 3 complete in {dt:.1f}s")
    print(f"[INFO] Energy distribution: H={energy_harmonic:.3f}, P={energy_percussive:.3f}")


class StageRunner:
    """Runs one stage and leaves the log the orchestrators' users grep: the stage's merged
    stdout / stderr, then `[elapsed_sec]: %.3f` and `[returncode]: %d` (log format of
    src/core_pipeline.py:26-39).  A stage is either a command (one interpreter per stage, like the
    reference) or a callable taking argv (same process: the GPU context, the kernels' tables and
    the device-resident stage files are shared by the stages that follow).  A non-zero outcome
    prints the captured output and ends the run with SystemExit, as the reference does."""

    def __init__(self, log_path):
        self.log_path = Path(log_path)
        self.log_path.parent.mkdir(parents=True, exist_ok=True)

    def _finish(self, what, text, rc, seconds):
        with open(self.log_path, "w", encoding="utf-8") as fh:
            fh.write(f"{text or ''}\n[elapsed_sec]: {seconds:.3f}\n[returncode]: {rc}\n")
        if rc != 0:
            print(text)
            raise SystemExit(f"Step failed: {what} (see {self.log_path})")
        print(f"[OK] ({seconds:.1f}s) -> {self.log_path}")

    def command(self, cmd):
        what = " ".join(cmd)
        print(f"[RUN] {what}")
        start = time.time()
        done = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        self._finish(what, done.stdout, done.returncode, time.time() - start)

    def call(self, label, fn, argv):
        import contextlib
        import io
        import traceback
        what = f"{label} {' '.join(argv)}"
        print(f"[RUN] {what}")
        captured, rc, start = io.StringIO(), 0, time.time()
        with contextlib.redirect_stdout(captured), contextlib.redirect_stderr(captured):
            try:
                fn(argv)
            except SystemExit as ex:
                if ex.code not in (None, 0):
                    rc = ex.code if isinstance(ex.code, int) else 1
                    if not isinstance(ex.code, int):
                        print(ex.code)
            except Exception:  # noqa: BLE001 - a stage failure of any kind ends the run like a non-zero exit
                traceback.print_exc(file=captured)
                rc = 1
        self._finish(what, captured.getvalue(), rc, time.time() - start)


def run_step(cmd, log_path):
    StageRunner(log_path).command(cmd)


def run_step_inprocess(label, fn, argv, log_path):
    StageRunner(log_path).call(label, fn, argv)


def core_pipeline_main(argv=None, src_dir=None):
    """src/core_pipeline.py:42-101.  ``ADN_CHAIN=process`` starts one interpreter per stage like the
    reference; the default runs the four stages in this process (one GPU context, stage outputs
    handed over on the device) with the same files and logs."""
    ap = argparse.ArgumentParser(description="Core DSP pipeline (no analysis, no AI)")
    ap.add_argument("--config", required=True)
    ap.add_argument("--input", required=True, help="Interview/input WAV path")
    ap.add_argument("--noise", required=True, help="Noise sample WAV path")
    args = ap.parse_args(argv)
    cfg = _load_config(args.config)
    paths = cfg.get("paths", {})
    paths["input"], paths["noise"] = args.input, args.noise
    paths.setdefault("logs_dir", "logs")
    paths.setdefault("outputs_dir", "outputs/stages")
    cfg["paths"] = paths
    logs_dir, outputs_dir = Path(paths["logs_dir"]).resolve(), Path(paths["outputs_dir"]).resolve()
    _mkdirs(logs_dir, outputs_dir)
    _dump(logs_dir / "resolved_config_core.json", cfg)
    src = Path(src_dir) if src_dir else Path(__file__).resolve().parents[1] / "src"
    mode = os.environ.get("ADN_CHAIN", "inprocess").lower()
    if mode not in ("inprocess", "process"):
        raise ValueError("ADN_CHAIN must be inprocess or process")
    py = sys.executable
    plan = [("noise_profile.py", noise_profile_main, ["--noise", args.noise], "core_phase1_noise_profile_stdout.log"),
            ("wavelet_denoise.py", wavelet_denoise_main, ["--input", args.input], "core_phase2_wavelet_stdout.log"),
            ("spectral_subtract.py", spectral_subtract_main, [], "core_phase2b_spectral_subtract_stdout.log"),
            ("wiener_filter.py", wiener_filter_main, [], "core_phase2c_wiener_stdout.log")]
    for script, fn, extra, log in plan:
        if mode == "process":
            run_step([py, str(src / script), "--config", args.config] + extra, logs_dir / log)
        else:
            run_step_inprocess(str(src / script), fn, ["--config", args.config] + extra, logs_dir / log)
    print("\n[COMPLETE] Core DSP pipeline finished")
    print(f"Wavelet output: {outputs_dir / 'phase2_wavelet.wav'}")
    print(f"Spectral subtract output: {outputs_dir / 'phase2b_spectral_subtract.wav'}")
    print(f"Wiener output (cleaned): {outputs_dir / 'phase2c_wiener.wav'}")

"""Wall time of the drop-in orchestrator (src/core_pipeline.py) on ONE 60 s, 48 kHz stereo clip +
5 s noise sample (BASELINE.json config 2 as a user runs it: files in, files out), with one interpreter
per stage like the reference (ADN_CHAIN=process) and with the stages chained in one process
(default).  Prints one JSON line."""
import json, os, subprocess, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "audio-denoising_amd"))
sys.path.insert(0, os.path.join(REPO, "oracle"))
import yaml
import synth
from adn_hip import wavio

SR = 48000
tmp = tempfile.mkdtemp(prefix="adn_dropin_")
px, pn = os.path.join(tmp, "in.wav"), os.path.join(tmp, "noise.wav")
wavio.write(px, synth.stereo_noisy_speech(60 * SR, SR, 42, 0), SR)
wavio.write(pn, synth.white_noise(5 * SR, 42, 100000), SR)
out = {"clip": "60 s 48 kHz stereo PCM_16 + 5 s noise", "dtype": os.environ.get("ADN_DTYPE", "f64")}
for mode in ("process", "inprocess"):
    work = os.path.join(tmp, mode)
    os.makedirs(work)
    cfg = {"paths": {"input": px, "noise": pn, "outputs_dir": os.path.join(work, "out"),
                     "logs_dir": os.path.join(work, "logs"), "reports_dir": os.path.join(work, "reports")},
           "analysis": {"stft": {"n_fft": 2048, "hop_length": 512, "win_length": 2048, "window": "hann", "center": True}},
           "noise_profile": {"percentile_level": 95, "spectral_floor": 0.1, "smoothing": {"freq_bins": 5, "time_frames": 5}},
           "spectral_subtraction": {"alpha": 2.2, "beta": 0.1, "n_fft": 2048, "hop_length": 512,
                                    "smoothing": {"time": 3, "freq": 5}},
           "wiener": {"update_ms": 20, "median_kernel_bins": 5, "vad_guided": True}}
    pc = os.path.join(work, "cfg.yaml")
    yaml.safe_dump(cfg, open(pc, "w"))
    env = dict(os.environ, ADN_CHAIN=mode)
    ts = []
    for rep in range(2):                      # second run: warm file cache and kernel code objects
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, os.path.join(REPO, "audio-denoising_amd", "src", "core_pipeline.py"),
                            "--config", pc, "--input", px, "--noise", pn], stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, env=env, cwd=work)
        ts.append(time.perf_counter() - t0)
        assert r.returncode == 0, r.stdout[-2000:]
    logs = os.path.join(work, "logs")
    stage = {}
    for f in sorted(os.listdir(logs)):
        if f.endswith("_stdout.log"):
            for line in open(os.path.join(logs, f)):
                if line.startswith("[elapsed_sec]:"):
                    stage[f.replace("core_", "").replace("_stdout.log", "")] = float(line.split()[1])
    out[mode] = {"wall_s": round(min(ts), 2), "stage_elapsed_s": stage}
print(json.dumps(out))

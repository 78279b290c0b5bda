"""GPU parity of phase 3 (src/hpss_separate.py): median-filtering HPSS through the C ABI and the
stage command line, against the files the reference wrote (golden fixture) and the oracle.
fp64: 1e-9 max-abs.  fp32: soft masks 1e-4 max-abs / 90 dB; hard and binary masks flip where the two
filtered magnitudes tie within fp32 rounding, so fp32 is checked on the soft case only."""
import json
import os

import numpy as np
import pytest
import torch

import adn_oracle as orc
import synth
from gpu_util import engine, dev, host, check64, check32
from test_oracle_golden import _hpss_kwargs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["default", "multiband_hard", "binary_scaled"])
def test_hpss_fp64_vs_reference_files(golden, tag):
    g = golden("hpss")
    m = json.loads(str(g["meta_json"]))
    c = m["cases"][tag]
    kw = _hpss_kwargs(c["hpss"])
    e = engine()
    x = g["x"]
    yh, yp, lens, stats = e.hpss(dev(x, torch.float64), [len(x)], m["sr"], kernel_size=kw["kernel_size"],
                                 margin_h=kw["margin_h"], margin_p=kw["margin_p"], power=kw["power"],
                                 mask_type=kw["mask_type"], bands=kw["bands"],
                                 preserve_transients=kw["preserve_transients"])
    check64(host(yh), g[tag + "_yh"])
    check64(host(yp), g[tag + "_yp"])
    st = host(stats)[0]
    for i, k in enumerate(("harmonic_rms", "percussive_rms", "total_rms")):
        assert abs(st[i] - c["energy"][k]) <= 1e-9, k


def test_hpss_batch_fp32_and_stereo():
    """Two stereo signals of different length in one call, fp32, soft masks, vs the oracle on the
    fp32-rounded input."""
    e = engine()
    sr = 48000
    xs = [synth.stereo_noisy_speech(n, sr, 42, 90 + i) for i, n in enumerate((100000, 70001))]
    for v in xs:
        v[30000:30040, :] += 0.7
    flat = np.concatenate([v.reshape(-1) for v in xs]).astype(np.float32)
    yh, yp, lens, stats = e.hpss(dev(flat, torch.float32), [len(v) for v in xs], sr, n_ch=2)
    yh, yp = host(yh), host(yp)
    pos = 0
    for v, L in zip(xs, lens):
        wh, wp, met = orc.hpss(v.astype(np.float32).astype(np.float64), sr)
        assert len(wh) == L
        check32(yh[pos:pos + L], wh)
        check32(yp[pos:pos + L], wp)
        pos += L
    st = host(stats)
    assert np.all(st[:, 4] >= 0)


def test_hpss_stage_command_line(golden, tmp_path, monkeypatch):
    import yaml
    from adn_hip import stages, wavio
    monkeypatch.setenv("ADN_WAV_SUBTYPE", "DOUBLE")
    monkeypatch.setenv("ADN_DTYPE", "f64")
    g = golden("hpss")
    m = json.loads(str(g["meta_json"]))
    c = m["cases"]["default"]
    tmp = str(tmp_path)
    px = os.path.join(tmp, "x.wav")
    wavio.write(px, g["x"], m["sr"])
    cfg = {"paths": {"input": px, "noise": px, "outputs_dir": os.path.join(tmp, "out"),
                     "logs_dir": os.path.join(tmp, "logs"), "reports_dir": os.path.join(tmp, "reports")},
           "hpss": c["hpss"]}
    pc = os.path.join(tmp, "cfg.yaml")
    with open(pc, "w") as fh:
        yaml.safe_dump(cfg, fh)
    stages.hpss_separate_main(["--config", pc])
    assert sorted(os.listdir(os.path.join(tmp, "out"))) == c["outputs"]
    yh, _ = wavio.read(os.path.join(tmp, "out", "phase3_harmonic.wav"))
    yc, _ = wavio.read(os.path.join(tmp, "out", "phase3_hpss.wav"))
    yp, _ = wavio.read(os.path.join(tmp, "out", "phase3_percussive.wav"))
    check64(yh, g["default_yh"])
    check64(yp, g["default_yp"])
    assert np.array_equal(yc, yh)
    log = json.load(open(os.path.join(tmp, "logs", "phase3_hpss.json")))
    assert sorted(log.keys()) == c["log_keys"]
    assert log["parameters"] == c["parameters"]
    for k, v in c["energy"].items():
        assert abs(log["metrics"]["energy_distribution"][k] - v) <= 1e-9

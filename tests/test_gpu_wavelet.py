"""GPU parity of the wavelet path (phase 2) through the C ABI: HIP kernels vs the CPU
oracle and vs golden vectors produced by the reference.  fp64: 1e-9 max-abs; fp32: 1e-4
max-abs and >= 90 dB SNR of the difference."""
import json

import numpy as np
import pytest
import torch

import adn_oracle as orc
import synth
from gpu_util import engine, dev, host, check64, check32
from test_oracle_golden import _denoise_input

pytestmark = pytest.mark.gpu


def test_native_library_is_the_hip_build():
    import adn_hip
    e = engine()
    assert adn_hip.LIB_PATH.endswith("libadn_hip.so")
    assert e.lib.adn_version() >= 100


def test_synth_generator_bit_exact():
    e = engine()
    for sr, n, stream, gain in ((16000, 160000, 0, 1.0), (48000, 100001, 7, 0.8)):
        got = host(e.synth_noisy_speech(n, sr, seed=42, stream=stream, speech_gain=gain))
        want = synth.noisy_speech(n, sr, 42, stream, speech_gain=gain)
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    got32 = host(e.synth_noisy_speech(5000, 48000, stream=3, dtype=torch.float32))
    assert np.array_equal(got32, synth.noisy_speech(5000, 48000, 42, 3).astype(np.float32))


@pytest.mark.parametrize("tag", ["c1", "short11", "oddblk", "stereo", "quad", "silence"])
def test_denoise_files_fp64_vs_reference(golden, tag):
    g = golden("denoise_files")
    m = {x["tag"]: x for x in json.loads(str(g["meta_json"]))}[tag]
    x = _denoise_input(tag)
    e = engine()
    y, (sigma, thresh, zeroed) = e.denoise_signal(dev(x), m["sr"], want_stats=True)
    y = host(y)
    check64(y, g[tag + "_y"])
    y_or, s_or, t_or = orc.denoise_signal(x, m["sr"])
    check64(y, y_or)
    assert np.max(np.abs(host(sigma) - s_or)) <= 1e-12
    assert np.max(np.abs(host(thresh) - t_or)) <= 1e-12


@pytest.mark.parametrize("tag", ["c1", "short11", "oddblk", "stereo", "silence"])
def test_denoise_files_fp32(golden, tag):
    g = golden("denoise_files")
    m = {x["tag"]: x for x in json.loads(str(g["meta_json"]))}[tag]
    x32 = _denoise_input(tag).astype(np.float32)
    e = engine()
    y = host(e.denoise_signal(dev(x32), m["sr"]))
    assert y.dtype == np.float32
    # the fp32 path sees the fp64 signal rounded to fp32; compare with the oracle on that input
    want, _, _ = orc.denoise_signal(x32.astype(np.float64), m["sr"])
    err, snr = check32(y, want)
    check32(y, g[tag + "_y"])


def test_block_stats_and_flip_count(golden):
    """sigma / threshold per block for db4 L2, db8 L8 (odd length at level 8), hard mode;
    threshold-boundary flips between fp32 and fp64 are counted and reported."""
    g = golden("block_stats")
    e = engine()
    report = []
    for m in json.loads(str(g["meta_json"])):
        x = synth.noisy_speech(m["n"], m["sr"], 42, m["stream"])
        n = m["n"]
        outs = {}
        for dt in (torch.float64, torch.float32):
            xd = dev(x, dt)
            out = torch.empty(2 * ((n + 1) // 2), dtype=dt, device="cuda:0")
            st = e.wavelet_denoise(xd, e._i64([0]), e._i64([n]), out, e._i64([0]), wavelet=m["wavelet"],
                                   level=m["level"], thr_mode=m["thr_mode"], want_stats=True)
            outs[dt] = (host(out), [host(s)[0] for s in st])
        t = m["tag"]
        y64, (sg, th, z64) = outs[torch.float64]
        assert abs(sg - float(g[t + "_sigma"])) <= 1e-12
        assert abs(th - float(g[t + "_thresh"])) <= 1e-12
        check64(y64[::m["stride"]], g[t + "_clean"])
        y32, (sg32, th32, z32) = outs[torch.float32]
        assert abs(sg32 - sg) <= 1e-6 * max(1.0, abs(sg))
        clean_or, _, _, bands = orc.denoise_block(x, m["wavelet"], m["level"], m["thr_mode"], return_bands=True)
        z_or = int(sum(np.sum(np.abs(b) < th) for b in bands[1:]))
        assert z64 == z_or
        flips = abs(int(z32) - int(z64))
        report.append((t, int(z64), flips))
        if m["thr_mode"] == "soft":
            check32(y32, clean_or)
        else:
            # hard thresholding is discontinuous: a flipped coefficient moves samples by ~thresh;
            # require agreement away from flips and a bounded flip count
            assert flips <= max(4, z64 // 20000)
    print("threshold-boundary flips (tag, zeroed_fp64, |zeroed_fp32 - zeroed_fp64|):", report)


def test_wavedec_waverec_primitives(golden):
    g = golden("dwt_primitives")
    e = engine()
    for i, m in enumerate(json.loads(str(g["per_meta_json"]))):
        x = g["per%d_x" % i]
        co = e.wavedec(dev(x), m["wavelet"], m["level"], "per")
        assert len(co) == m["bands"]
        for b, c in enumerate(co):
            check64(host(c), g["per%d_c%d" % (i, b)], 1e-12)
        check64(host(e.waverec(co, m["wavelet"], "per")), g["per%d_rec" % i], 1e-12)
    for i, m in enumerate(json.loads(str(g["sym_meta_json"]))):
        x = g["sym%d_x" % i]
        ca, cd = e.wavedec(dev(x), m["wavelet"], 1, "sym")
        check64(host(ca), g["sym%d_cA" % i], 1e-12)
        check64(host(cd), g["sym%d_cD" % i], 1e-12)
        check64(host(e.waverec([ca, cd], m["wavelet"], "sym")), g["sym%d_rec" % i], 1e-12)
    # wavelet packet leaves = repeated single-level symmetric steps (lib/windowBundle.py:33-34)
    paths = json.loads(str(g["wp_paths_freq_json"]))
    for p, ref in zip(paths[:4] + paths[-2:], list(g["wp_leaves_freq"][:4]) + list(g["wp_leaves_freq"][-2:])):
        a = dev(g["wp_x"])
        for ch in p:
            ca, cd = e.wavedec(a, "db8", 1, "sym")
            a = ca if ch == "a" else cd
        check64(host(a), ref, 1e-12)


def test_round_trip_property_large():
    """Size-independent property at a benchmark-sized block: with a zero threshold the
    analysis/synthesis pair is the identity (perfect reconstruction)."""
    e = engine()
    x = e.synth_noisy_speech(144000 * 8, 48000, stream=5)
    co = e.wavedec(x[:144000], "db8", 8, "per")
    y = e.waverec(co, "db8", "per")
    assert float((y - x[:144000]).abs().max()) <= 1e-11


def test_window_rms(golden):
    from test_oracle_golden import _profiler_input
    g = golden("noise_profiler")
    x = _profiler_input()
    e = engine()
    check64(host(e.window_rms(dev(x), 4410)), g["rms"], 1e-14)
    xs = np.stack([x, 0.5 * x[::-1]], axis=1)
    check64(host(e.window_rms(dev(xs), 4410)), orc.window_rms(xs, 4410), 1e-14)
    r32 = host(e.window_rms(dev(x.astype(np.float32)), 4410))
    assert np.max(np.abs(r32 - g["rms"]) / g["rms"]) <= 1e-6


def test_error_reporting():
    from adn_hip import AdnError
    e = engine()
    x = dev(np.zeros((100, 3)))
    with pytest.raises(AdnError, match="even channel count"):
        e.denoise_signal(x, 100)
    with pytest.raises(ValueError, match="shorter than 10 frames"):
        e.denoise_signal(dev(np.zeros(5)), 8000)


def test_mad_long_segments_exact():
    """Long segments take the sampled-bracket path; a bracket of one key (constant data, heavy ties)
    is resolved from the counts, and the radix passes take whatever a bracket cannot hold; results
    stay exact."""
    e = engine()
    rng = np.random.default_rng(3)
    n = 200001
    cases = {
        "normal_odd": rng.standard_normal(n),
        "normal_even": rng.standard_normal(n - 1),
        "constant": np.full(n, 0.25),
        "zeros": np.zeros(100000),
        "ties": rng.integers(0, 4, n).astype(np.float64),
        "lognormal": np.exp(3.0 * rng.standard_normal(n)),
        "short_big_boundary": rng.standard_normal(40000),
    }
    for name, a in cases.items():
        for dt, tol in ((torch.float64, 0.0), (torch.float32, 0.0)):
            x = a.astype(np.float32 if dt == torch.float32 else np.float64)
            got = float(e.mad(dev(x)))
            want = float(orc.mad(x.astype(np.float64))) if dt == torch.float64 else \
                float(np.median(np.abs(x - np.median(x))))
            assert abs(got - want) <= tol + 1e-7 * abs(want) * (dt == torch.float32), (name, dt, got, want)
    # several segments of different lengths in one call, one of them with a NaN
    parts = [rng.standard_normal(150000), rng.standard_normal(70000), rng.standard_normal(33000)]
    parts[1][123] = np.nan
    flat = np.concatenate(parts)
    got = host(e.mad(dev(flat), [p.size for p in parts]))
    assert np.isnan(got[1])
    assert got[0] == orc.mad(parts[0]) and got[2] == orc.mad(parts[2])


def test_mad_multi_level_bracket_exact():
    """Segments whose first candidate list does not fit LDS are bracketed again on the gathered
    keys (three levels at 3.3 M fp64 keys); a list that turns out short passes through a level."""
    e = engine()
    rng = np.random.default_rng(11)
    big = rng.standard_normal(3_300_001)
    skew = np.exp(2.0 * rng.standard_normal(1_500_000))
    # second segment far shorter than the first: its lists pass through the deeper levels whole
    mixed = [rng.standard_normal(2_000_000), rng.standard_normal(40_000), np.full(300_000, -1.5)]
    for dt in (torch.float64, torch.float32):
        npdt = np.float32 if dt == torch.float32 else np.float64
        for a in (big, skew):
            x = a.astype(npdt)
            want = float(np.median(np.abs(x - np.median(x))))
            assert float(e.mad(dev(x))) == want
        xs = [m.astype(npdt) for m in mixed]
        got = host(e.mad(dev(np.concatenate(xs)), [m.size for m in xs]))
        want = [float(np.median(np.abs(m - np.median(m)))) for m in xs]
        assert list(got) == want


def _batch_denoise(e, xs, dt, wavelet, level, thr_mode):
    """All blocks of `xs` (list of 1-D arrays) in one C-ABI call; returns outputs + stats."""
    lens = np.array([len(x) for x in xs], dtype=np.int64)
    in_off = np.concatenate([[0], np.cumsum(lens)[:-1]])
    out_len = 2 * ((lens + 1) // 2)
    out_off = np.concatenate([[0], np.cumsum(out_len)[:-1]])
    xd = dev(np.concatenate(xs), dt)
    out = torch.full((int(out_len.sum()),), float("nan"), dtype=dt, device="cuda:0")
    st = e.wavelet_denoise(xd, e._i64(in_off), e._i64(lens), out, e._i64(out_off), wavelet=wavelet, level=level,
                           thr_mode=thr_mode, max_frames=int(lens.max()), want_stats=True)
    o = host(out)
    return [o[a:a + l] for a, l in zip(out_off, out_len)], [host(s) for s in st]


@pytest.mark.parametrize("wavelet,level", [("db4", 1), ("db4", 2), ("db4", 3), ("db8", 4), ("db8", 5), ("db8", 8),
                                           ("db4", 7)])
def test_tile_kernels_any_block_length(wavelet, level):
    """Mono blocks of awkward lengths in one batch: shorter than a filter, shorter than one tile,
    several tiles, odd at the first / at deeper levels, a multiple of 2^level.  The fused tile
    kernels (interior and boundary index paths, deep kernel for level > 4) must match the
    oracle, the level-by-level kernels, and count the same zeroed coefficients."""
    e = engine()
    lens = [1, 2, 3, 7, 17, 100, 255, 4095, 4096, 4097, 8191, 9001, 12288, 20000, 20001, 20003, 33333, 40960]
    rng = np.random.default_rng(5)
    xs = [synth.noisy_speech(n, 16000, 42, i) if n > 50 else rng.standard_normal(n) for i, n in enumerate(lens)]
    for thr_mode in ("soft", "hard"):
        ys, st = _batch_denoise(e, xs, torch.float64, wavelet, level, thr_mode)
        e.force_generic(True)
        try:
            ys_g, st_g = _batch_denoise(e, xs, torch.float64, wavelet, level, thr_mode)
        finally:
            e.force_generic(False)
        for x, y, yg in zip(xs, ys, ys_g):
            want, s_or, t_or = orc.denoise_block(x, wavelet, level, thr_mode)
            if np.all(np.isfinite(want)):
                check64(y, want)
                check64(yg, want)
            else:
                assert np.array_equal(np.isnan(y), np.isnan(want))
        for a, b in zip(st, st_g):
            assert np.array_equal(a, b, equal_nan=True), (a, b)
    # fp32: same batch, tolerance of the fp32 path
    ys32, _ = _batch_denoise(e, [x.astype(np.float32) for x in xs], torch.float32, wavelet, level, "soft")
    for x, y in zip(xs, ys32):
        want, _, _ = orc.denoise_block(x.astype(np.float32).astype(np.float64), wavelet, level, "soft")
        if len(x) > 50:
            check32(y, want)


def test_blocks_sharded_like_ranks_match_single_call():
    """The 10 % blocks of one long recording are independent units: processing them in the
    round-robin shards that N ranks would take (adn_hip.dist.shard_indices), each shard writing its
    blocks into the shared output layout, gives bit-identical samples and statistics to one call
    over all blocks.  (This is how a recording too long for one rank's share is split; no
    collective is involved.)"""
    from adn_hip.dist import shard_indices
    e = engine()
    sr, n = 48000, 48000 * 20 + 7
    x = synth.noisy_speech(n, sr, 42, 77)
    bs = e.denoise_block_frames(sr, n)
    starts = np.arange(0, n, bs, dtype=np.int64)
    lens = np.minimum(bs, n - starts)
    out_len = 2 * ((lens + 1) // 2)
    out_off = np.concatenate([[0], np.cumsum(out_len)[:-1]])
    xd = dev(x, torch.float64)

    def run(idx):
        out = torch.full((int(out_len.sum()),), float("nan"), dtype=torch.float64, device="cuda:0")
        st = e.wavelet_denoise(xd, e._i64(starts[idx]), e._i64(lens[idx]), out, e._i64(out_off[idx]), wavelet="db4",
                               level=2, thr_mode="soft", max_frames=int(lens[idx].max()), want_stats=True)
        return host(out), [host(s) for s in st]

    full, st_full = run(np.arange(len(lens)))
    for world in (2, 3, 8):
        merged = np.full_like(full, np.nan)
        stats = [np.full_like(s, np.nan if s.dtype.kind == "f" else -1) for s in st_full]
        for rank in range(world):
            idx = shard_indices(len(lens), rank, world)
            part, st = run(idx)
            for b in idx:
                merged[out_off[b]:out_off[b] + out_len[b]] = part[out_off[b]:out_off[b] + out_len[b]]
            for dst, src in zip(stats, st):
                dst[idx] = src
        assert np.array_equal(merged, full)
        for a, b in zip(stats, st_full):
            assert np.array_equal(a, b)
    want, _, _ = orc.denoise_signal(x, sr)
    check64(full[:len(want)], want)


def _stereo_batch(e, blocks, dt, thr_mode="soft"):
    lens = np.array([b.shape[0] for b in blocks], dtype=np.int64)
    off = np.concatenate([[0], np.cumsum(lens)[:-1]]) * 2
    xd = dev(np.concatenate([b.reshape(-1) for b in blocks]), dt)
    out = torch.full((int(lens.sum()) * 2,), float("nan"), dtype=dt, device="cuda:0")
    st = e.wavelet_denoise(xd, e._i64(off), e._i64(lens), out, e._i64(off), n_ch=2, wavelet="db4", level=2,
                           thr_mode=thr_mode, max_frames=int(lens.max()), want_stats=True)
    o = host(out)
    return [o[a:a + 2 * l].reshape(l, 2) for a, l in zip(off, lens)], [host(s) for s in st]


@pytest.mark.parametrize("dt,longest", [(torch.float32, 290000), (torch.float32, 400000), (torch.float64, 120000),
                                        (torch.float64, 200000)])
def test_stereo_single_pass_statistics_exact(dt, longest):
    """Stereo blocks take their threshold statistics from one pass over the input (sample ->
    brackets -> gather -> exact selection among the gathered keys, verified, exact fallback).
    Sigma, threshold, zero counts and output must EQUAL those of the plain route (coefficient
    array + two full selections) for Gaussian, heavy-tailed, quantised, bimodal, half-silent,
    constant, silent, NaN-carrying, short and one-frame blocks, and match the oracle.  The longer
    size of each dtype makes the candidate lists too long for LDS: they are then selected in memory."""
    e = engine()
    rng = np.random.default_rng(77)
    npdt = np.float32 if dt == torch.float32 else np.float64

    def st(n, kind):
        if kind == "gauss":
            x = 0.1 * rng.standard_normal((n, 2))
        elif kind == "laplace":
            x = 0.05 * rng.laplace(size=(n, 2))
        elif kind == "pcm":
            x = np.round(0.2 * rng.standard_normal((n, 2)) * 512) / 512
        elif kind == "bimodal":
            s = rng.choice([-1.0, 1.0], size=n)
            x = np.stack([s * 0.5 + 1e-3 * rng.standard_normal(n), -s * 0.5 + 1e-3 * rng.standard_normal(n)], axis=1)
        elif kind == "halfsilent":
            x = 0.1 * rng.standard_normal((n, 2))
            x[: n // 2 + 7] = 0.0
        elif kind == "const":
            x = np.tile(np.array([[0.25, -0.125]]), (n, 1))
        elif kind == "zero":
            x = np.zeros((n, 2))
        elif kind == "nan":
            x = 0.1 * rng.standard_normal((n, 2))
            x[n // 3, 1] = np.nan
        elif kind == "offset":
            x = 0.01 * rng.standard_normal((n, 2)) + np.array([[0.7, -0.6]])
        return x.astype(npdt)

    spec = [(longest, "gauss"), (100000, "laplace"), (40000, "gauss"), (32768, "gauss"), (20000, "laplace"),
            (5, "gauss"), (1, "gauss"), (longest - 1, "zero"), (110000, "nan"), (longest, "bimodal"),
            (90000, "const"), (longest - 3, "halfsilent"), (65001, "pcm"), (70000, "offset"), (16385, "gauss")]
    # lists too long for LDS are taken only in batches of at least one block per CU: pad with tiny blocks
    spec += [(2 + i % 3, "gauss") for i in range(250)]
    blocks = [st(n, k) for n, k in spec]
    for thr_mode in ("soft", "hard"):
        ys, s1 = _stereo_batch(e, blocks, dt, thr_mode)
        e.force_generic(True)
        try:
            yg, s2 = _stereo_batch(e, blocks, dt, thr_mode)
        finally:
            e.force_generic(False)
        e.force_exact_fallback(True)                 # every block through the exact per-block fallback
        try:
            yf, s3 = _stereo_batch(e, blocks, dt, thr_mode)
        finally:
            e.force_exact_fallback(False)
        for name, a, b, f in zip(("sigma", "thresh", "zeroed"), s1, s2, s3):
            assert np.array_equal(a, b, equal_nan=True), (name, thr_mode, a, b)
            assert np.array_equal(a, f, equal_nan=True), (name, thr_mode, a, f)
        for (n, k), y, g, f in zip(spec, ys, yg, yf):
            assert np.array_equal(y, g, equal_nan=True), (n, k)
            assert np.array_equal(y, f, equal_nan=True), (n, k)
    # oracle: sigma of every block (binary64 restatement on the same inputs)
    for (n, k), blk, sg in zip(spec, blocks, s1[0]):
        _, s_or, _ = orc.denoise_block(blk.astype(np.float64), "db4", 2, "soft")
        if np.isnan(s_or):
            assert np.isnan(sg), (n, k)
        else:
            # the coefficients are rounded in the block's dtype: a few ulps of the largest sample
            scale = max(float(np.max(np.abs(blk))), 1e-3)
            assert abs(sg - s_or) <= (1e-6 if dt == torch.float32 else 1e-14) * scale, (n, k, sg, s_or)


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_stereo_single_pass_edge_lengths(dt):
    """Block lengths around the single-pass route's internal sizes (sample size 16384, gather tile
    4096): batches of only short blocks (no whole tile at all), exact multiples of a tile, one
    frame more or less, and a batch whose longest block is just past the sample size."""
    e = engine()
    rng = np.random.default_rng(101)
    npdt = np.float32 if dt == torch.float32 else np.float64
    for lens in ([1, 2, 3, 5, 100, 4095], [4096, 8192, 12288, 16384, 4097, 8191], [16385, 16383, 20480, 1, 32768, 40961]):
        blocks = [(0.2 * rng.standard_normal((n, 2))).astype(npdt) for n in lens]
        ys, s1 = _stereo_batch(e, blocks, dt)
        e.force_generic(True)
        try:
            yg, s2 = _stereo_batch(e, blocks, dt)
        finally:
            e.force_generic(False)
        e.force_exact_fallback(True)
        try:
            yf, s3 = _stereo_batch(e, blocks, dt)
        finally:
            e.force_exact_fallback(False)
        for a, b, f in zip(s1, s2, s3):
            assert np.array_equal(a, b, equal_nan=True), (lens, a, b)
            assert np.array_equal(a, f, equal_nan=True), (lens, a, f)
        for n, y, g, f in zip(lens, ys, yg, yf):
            assert np.array_equal(y, g, equal_nan=True), n
            assert np.array_equal(y, f, equal_nan=True), n


@pytest.mark.parametrize("dt,wavelet,level,longest", [(torch.float32, "db4", 2, 144000), (torch.float32, "db8", 3, 400001),
                                                      (torch.float64, "db4", 2, 120001), (torch.float64, "db8", 4, 200000)])
def test_mono_single_pass_statistics_exact(dt, wavelet, level, longest):
    """Mono blocks whose levels all live in the tile kernels take their threshold statistics from one
    pass over the samples (finest-band coefficients formed on the fly: sample -> brackets -> gather ->
    exact selection among the gathered keys, verified, exact fallback).  Sigma, threshold and zero
    counts must EQUAL those of the level-by-level route (coefficient arrays + two full selections)
    and of the exact per-block fallback, for Gaussian, heavy-tailed, quantised, autocorrelated,
    half-silent, constant, silent, NaN-carrying, odd, short and one-sample blocks, with lengths round
    the route's internal sizes (bracket sample 4096 / 16384 coefficients, gather tile 4096
    coefficients = 8192 samples); sigma must match the oracle."""
    e = engine()
    rng = np.random.default_rng(99)
    npdt = np.float32 if dt == torch.float32 else np.float64

    def st(n, kind):
        if kind == "gauss":
            x = 0.1 * rng.standard_normal(n)
        elif kind == "laplace":
            x = 0.05 * rng.laplace(size=n)
        elif kind == "pcm":
            x = np.round(0.2 * rng.standard_normal(n) * 512) / 512
        elif kind == "ar":
            w = 0.05 * rng.standard_normal(n)
            x = np.empty(n)
            acc = 0.0
            for i in range(n):
                acc = 0.9 * acc + w[i]
                x[i] = acc
        elif kind == "halfsilent":
            x = 0.1 * rng.standard_normal(n)
            x[: n // 2 + 7] = 0.0
        elif kind == "const":
            x = np.full(n, 0.25)
        elif kind == "zero":
            x = np.zeros(n)
        elif kind == "nan":
            x = 0.1 * rng.standard_normal(n)
            x[n // 3] = np.nan
        elif kind == "speech":
            x = synth.noisy_speech(n, 48000, 42, n % 97)
        return x.astype(npdt)

    spec = [(longest, "gauss"), (longest - 1, "speech"), (100001, "laplace"), (40000, "ar"), (32768, "gauss"),
            (8192, "gauss"), (8193, "gauss"), (8191, "laplace"), (8194, "gauss"), (16384, "speech"), (16385, "gauss"),
            (5, "gauss"), (1, "gauss"), (2, "gauss"), (longest - 3, "zero"), (110000, "nan"), (90001, "const"),
            (longest - 5, "halfsilent"), (65001, "pcm"), (24577, "gauss"), (9, "zero")]
    blocks = [st(n, k) for n, k in spec]
    for thr_mode in ("soft", "hard"):
        ys, s1 = _batch_denoise(e, blocks, dt, wavelet, level, thr_mode)
        e.force_generic(True)
        try:
            yg, s2 = _batch_denoise(e, blocks, dt, wavelet, level, thr_mode)
        finally:
            e.force_generic(False)
        e.force_exact_fallback(True)                 # every block through the exact per-block fallback
        try:
            yf, s3 = _batch_denoise(e, blocks, dt, wavelet, level, thr_mode)
        finally:
            e.force_exact_fallback(False)
        for name, a, b, f in zip(("sigma", "thresh", "zeroed"), s1, s2, s3):
            assert np.array_equal(a, b, equal_nan=True), (name, thr_mode, a, b)
            assert np.array_equal(a, f, equal_nan=True), (name, thr_mode, a, f)
        for (n, k), y, f in zip(spec, ys, yf):
            assert np.array_equal(y, f, equal_nan=True), (n, k)
    for (n, k), blk, sg in zip(spec, blocks, s1[0]):
        _, s_or, _ = orc.denoise_block(blk.astype(np.float64), wavelet, level, "soft")
        if np.isnan(s_or):
            assert np.isnan(sg), (n, k)
        else:
            scale = max(float(np.max(np.abs(blk))), 1e-3)
            assert abs(sg - s_or) <= (2e-6 if dt == torch.float32 else 1e-14) * scale, (n, k, sg, s_or)

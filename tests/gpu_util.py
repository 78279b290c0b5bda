"""Helpers shared by the GPU parity tests."""
import numpy as np
import torch

TOL64 = 1e-9          # fp64 mode: max-abs error against the reference / oracle
TOL32 = 1e-4          # fp32 mode: max-abs error
SNR32_DB = 90.0       # fp32 mode: SNR of the difference, dB


def engine():
    from adn_hip import get_engine
    return get_engine(0)


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a), device="cuda:0")
    return t.to(dtype) if dtype is not None else t


def host(t):
    return t.detach().cpu().numpy()


def check64(got, want, tol=TOL64):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    assert np.array_equal(np.isnan(got), np.isnan(want)), "NaN positions differ"
    m = ~np.isnan(want)
    err = float(np.max(np.abs(got[m] - want[m]))) if m.any() else 0.0
    assert err <= tol, "fp64 max-abs error %.3e > %.1e" % (err, tol)
    return err


def check32(got, want, tol=TOL32, snr_db=SNR32_DB):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    assert np.array_equal(np.isnan(got), np.isnan(want)), "NaN positions differ"
    m = ~np.isnan(want)
    if not m.any():
        return 0.0, np.inf
    diff = got[m] - want[m]
    err = float(np.max(np.abs(diff)))
    p_sig, p_err = float(np.sum(want[m] ** 2)), float(np.sum(diff ** 2))
    snr = np.inf if p_err == 0 else 10.0 * np.log10(max(p_sig, 1e-300) / p_err)
    assert err <= tol, "fp32 max-abs error %.3e > %.1e" % (err, tol)
    assert snr >= snr_db, "fp32 SNR of the difference %.1f dB < %.1f dB" % (snr, snr_db)
    return err, snr

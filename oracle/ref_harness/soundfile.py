"""Lossless stand-in for python-soundfile, used ONLY by oracle/gen_golden.py.

The reference imports ``soundfile`` at module top (denoise.py:10, src/*.py) but
neither python-soundfile nor libsndfile exists in the build container.  This
module offers the handful of calls the hot path makes and stores audio as
float64 without quantisation, so golden vectors pin the DSP arithmetic and not
a WAV sample format.  The on-disk container is an ``np.savez`` archive written
at the exact path given (whatever its extension).
"""
import numpy as np


def _load(path):
    with np.load(path, allow_pickle=False) as z:
        return np.array(z["data"], dtype=np.float64), int(z["samplerate"])


def _store(path, data, samplerate):
    with open(path, "wb") as fh:
        np.savez(fh, data=np.asarray(data, dtype=np.float64), samplerate=np.int64(samplerate))


class _Info:
    def __init__(self, path):
        data, sr = _load(path)
        self.samplerate = sr
        self.frames = data.shape[0]
        self.channels = 1 if data.ndim == 1 else data.shape[1]
        self.duration = float(self.frames) / sr
        self.format = "WAV"
        self.subtype = "DOUBLE"


def info(path):
    return _Info(path)


def read(path):
    return _load(path)


def write(path, data, samplerate):
    _store(path, np.asarray(data, dtype=np.float64), samplerate)


def blocks(path, blocksize):
    data, _ = _load(path)
    if not blocksize or blocksize <= 0:
        raise ValueError("blocksize must be positive")
    for s in range(0, data.shape[0], blocksize):
        yield data[s:s + blocksize].copy()


class SoundFile:
    def __init__(self, path, mode="r", samplerate=None, channels=None):
        if mode != "w":
            raise NotImplementedError("stand-in supports write mode only")
        self._path, self._sr, self._ch, self._parts = path, samplerate, channels, []

    def write(self, data):
        data = np.asarray(data, dtype=np.float64)
        ch = 1 if data.ndim == 1 else data.shape[1]
        if ch != self._ch:
            raise ValueError("channel count mismatch: %d != %d" % (ch, self._ch))
        self._parts.append(data)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if self._parts:
            data = np.concatenate(self._parts, axis=0)
        else:
            data = np.zeros((0,) if self._ch == 1 else (0, self._ch))
        _store(self._path, data, self._sr)
        return False

"""Minimal WAV reader / writer for the stage files (numpy + struct only).

The reference uses python-soundfile (libsndfile), which this build does not depend
on.  The calls the hot path makes are mirrored here: ``info``, ``read``, ``write``,
``blocks`` and a write-mode ``SoundFile`` context manager (denoise.py:72-76,90,104-108,
src/*.py sf.read / sf.write).  Like libsndfile, a ``.wav`` written without an explicit
subtype is 16-bit PCM; samples are read back as float64.

PCM conversion (parity unpinned: libsndfile is absent from the build container, so the
rule below restates the conversion libsndfile applies to normalised doubles when clipping
is switched on, which python-soundfile does for every file it opens): the sample is scaled
by 2^31, rounded to the nearest integer (ties to even), saturated to the int32 range, and
the top `bits` bits are kept (an arithmetic shift, i.e. rounding towards minus infinity on
the 16-bit grid: int16 = floor(rint(x * 2^31) / 2^16)); NaN is written as 0.  Reading
divides by 2^(bits-1).  ``adn_pcm_encode_*`` applies the same rule on the GPU.
Set ``ADN_WAV_SUBTYPE=DOUBLE`` (or pass ``subtype=``) to keep inter-stage files lossless.
"""
import os
import struct

import numpy as np

_FMT_PCM, _FMT_FLOAT, _FMT_EXT = 1, 3, 0xFFFE
SUBTYPES = {"PCM_16": (_FMT_PCM, 16), "PCM_24": (_FMT_PCM, 24), "PCM_32": (_FMT_PCM, 32),
            "FLOAT": (_FMT_FLOAT, 32), "DOUBLE": (_FMT_FLOAT, 64)}


def default_subtype():
    return os.environ.get("ADN_WAV_SUBTYPE", "PCM_16").upper()


class Info:
    def __init__(self, samplerate, channels, frames, subtype):
        self.samplerate, self.channels, self.frames, self.subtype = samplerate, channels, frames, subtype
        self.duration = float(frames) / samplerate
        self.format = "WAV"

    def __repr__(self):
        return "WAV %d Hz, %d ch, %d frames, %s" % (self.samplerate, self.channels, self.frames, self.subtype)


def _parse(path):
    """(Info, data offset, data bytes, fmt code, bits)."""
    with open(path, "rb") as fh:
        head = fh.read(12)
        if len(head) < 12 or head[:4] not in (b"RIFF", b"RF64") or head[8:12] != b"WAVE":
            raise RuntimeError("%s: not a RIFF/WAVE file" % path)
        fmt = None
        while True:
            ck = fh.read(8)
            if len(ck) < 8:
                raise RuntimeError("%s: no data chunk" % path)
            cid, size = ck[:4], struct.unpack("<I", ck[4:])[0]
            if cid == b"fmt ":
                raw = fh.read(size + (size & 1))
                code, ch, sr, _, _, bits = struct.unpack("<HHIIHH", raw[:16])
                if code == _FMT_EXT and size >= 26:
                    code = struct.unpack("<H", raw[24:26])[0]
                fmt = (code, ch, sr, bits)
            elif cid == b"data":
                if fmt is None:
                    raise RuntimeError("%s: data chunk before fmt chunk" % path)
                off = fh.tell()
                fh.seek(0, 2)
                avail = fh.tell() - off
                size = min(size, avail) if size not in (0, 0xFFFFFFFF) else avail
                code, ch, sr, bits = fmt
                sub = {(_FMT_PCM, 16): "PCM_16", (_FMT_PCM, 24): "PCM_24", (_FMT_PCM, 32): "PCM_32",
                       (_FMT_FLOAT, 32): "FLOAT", (_FMT_FLOAT, 64): "DOUBLE", (_FMT_PCM, 8): "PCM_U8"}.get((code, bits))
                if sub is None:
                    raise RuntimeError("%s: unsupported WAV encoding (format %d, %d bits)" % (path, code, bits))
                frames = size // (ch * bits // 8)
                return Info(sr, ch, frames, sub), off, frames * ch * bits // 8, code, bits
            else:
                fh.seek(size + (size & 1), 1)


def info(path):
    return _parse(path)[0]


def _decode(raw, code, bits):
    if code == _FMT_FLOAT:
        return np.frombuffer(raw, dtype="<f4" if bits == 32 else "<f8").astype(np.float64)
    if bits == 16:
        return np.frombuffer(raw, dtype="<i2").astype(np.float64) / 32768.0
    if bits == 32:
        return np.frombuffer(raw, dtype="<i4").astype(np.float64) / 2147483648.0
    if bits == 24:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v & 0x800000, v - 0x1000000, v)
        return v.astype(np.float64) / 8388608.0
    if bits == 8:
        return (np.frombuffer(raw, dtype=np.uint8).astype(np.float64) - 128.0) / 128.0
    raise RuntimeError("unsupported sample width %d" % bits)


def read(path, start=0, frames=None):
    """(data float64, samplerate); data is (frames,) for mono, (frames, channels) otherwise."""
    inf, off, nbytes, code, bits = _parse(path)
    fb = inf.channels * bits // 8
    frames = inf.frames - start if frames is None else min(frames, inf.frames - start)
    with open(path, "rb") as fh:
        fh.seek(off + start * fb)
        raw = fh.read(max(frames, 0) * fb)
    data = _decode(raw, code, bits)
    if inf.channels > 1:
        data = data.reshape(-1, inf.channels)
    return data, inf.samplerate


def blocks(path, blocksize):
    """Successive blocks of `blocksize` frames; the last one may be shorter (soundfile.blocks)."""
    if not blocksize or blocksize <= 0:
        raise ValueError("blocksize must be a positive number of frames")
    inf = info(path)
    for s in range(0, inf.frames, blocksize):
        yield read(path, s, blocksize)[0]


def _encode(data, subtype):
    code, bits = SUBTYPES[subtype]
    x = np.ascontiguousarray(data, dtype=np.float64)
    if code == _FMT_FLOAT:
        return x.astype("<f4" if bits == 32 else "<f8").tobytes()
    q = pcm_quantize(x, bits)
    if bits == 16:
        return q.astype("<i2").tobytes()
    if bits == 32:
        return q.astype("<i4").tobytes()
    v = q & 0xFFFFFF
    out = np.empty((v.size, 3), dtype=np.uint8)
    fl = v.ravel()
    out[:, 0], out[:, 1], out[:, 2] = fl & 0xFF, (fl >> 8) & 0xFF, (fl >> 16) & 0xFF
    return out.tobytes()


def pcm_quantize(x, bits):
    """Integer sample values (int32 array) of normalised floating-point samples."""
    x = np.clip(np.nan_to_num(np.asarray(x, dtype=np.float64), nan=0.0), -2.0, 2.0)   # saturates anyway
    q = np.clip(np.rint(x * 2147483648.0), -2147483648.0, 2147483647.0).astype(np.int64)
    return (q >> (32 - bits)).astype(np.int32)


def _header(samplerate, channels, subtype, data_bytes):
    code, bits = SUBTYPES[subtype]
    block = channels * bits // 8
    fmt = struct.pack("<HHIIHH", code, channels, samplerate, samplerate * block, block, bits)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt
    if code == _FMT_FLOAT:
        body += b"fact" + struct.pack("<II", 4, data_bytes // block)
    body += b"data" + struct.pack("<I", min(data_bytes, 0xFFFFFFFF))
    return b"RIFF" + struct.pack("<I", min(len(body) + data_bytes, 0xFFFFFFFF)) + body


def write(path, data, samplerate, subtype=None):
    data = np.asarray(data, dtype=np.float64)
    ch = 1 if data.ndim == 1 else data.shape[1]
    with SoundFile(path, "w", samplerate=samplerate, channels=ch, subtype=subtype) as fh:
        fh.write(data)


def write_pcm16(path, q, samplerate, channels=1):
    """16-bit PCM file from integer sample values (interleaved int16, e.g. made on the GPU)."""
    raw = np.ascontiguousarray(q, dtype="<i2").tobytes()
    with open(path, "wb") as fh:
        fh.write(_header(int(samplerate), int(channels), "PCM_16", len(raw)))
        fh.write(raw)


class SoundFile:
    """Write-mode file that accepts successive blocks (denoise.py:75,90)."""

    def __init__(self, path, mode="w", samplerate=None, channels=None, subtype=None):
        if mode != "w":
            raise NotImplementedError("only write mode is provided; use read()/blocks() to read")
        self.subtype = (subtype or default_subtype()).upper()
        if self.subtype not in SUBTYPES:
            raise ValueError("unknown WAV subtype %r" % self.subtype)
        self.samplerate, self.channels = int(samplerate), int(channels)
        self._fh = open(path, "wb")
        self._bytes = 0
        self._fh.write(_header(self.samplerate, self.channels, self.subtype, 0))

    def write(self, data):
        data = np.asarray(data, dtype=np.float64)
        ch = 1 if data.ndim == 1 else data.shape[1]
        if ch != self.channels:
            raise ValueError("block has %d channels, file has %d" % (ch, self.channels))
        raw = _encode(data, self.subtype)
        self._fh.write(raw)
        self._bytes += len(raw)

    def close(self):
        if self._fh is not None:
            if self._bytes & 1:
                self._fh.write(b"\x00")
            self._fh.seek(0)
            self._fh.write(_header(self.samplerate, self.channels, self.subtype, self._bytes))
            self._fh.close()
            self._fh = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

"""Stage files that stay resident on the device.

The reference chains its stages through WAV files (16-bit PCM unless told otherwise): a stage
writes its output, the next one reads it back (src/spectral_subtract.py:66-69,
src/wiener_filter.py:57-62).  The files are still written here, but

* for 16-bit PCM the sample values are produced on the GPU (the stage kernel's own store for
  phases 2b / 2c, ``adn_pcm_encode_*`` otherwise), so 2 bytes per sample cross PCIe instead of 8
  and the host does no conversion;
* the samples a reader of the file would get are kept on the device, keyed by the file's path,
  size and modification time, so a later stage running in the same process skips the file read
  and the upload.  A file changed by anyone else in between is read from disk as usual.

Set ``ADN_DEVICE_FILES=0`` to switch the residency off (the files are identical either way).
At most ``ADN_DEVICE_FILES_MB`` MiB (default 4096) of samples stay resident; larger files are
simply read back from disk by the next stage.
"""
import os
from collections import OrderedDict

import numpy as np

from . import wavio

_MAX_RESIDENT = 4
_resident = OrderedDict()      # realpath -> (tensor, samplerate, channels, (size, mtime_ns))


def _enabled():
    return os.environ.get("ADN_DEVICE_FILES", "1") != "0"


def _stamp(path):
    st = os.stat(path)
    return (st.st_size, st.st_mtime_ns)


def forget(path=None):
    """Drops one resident file (or all of them)."""
    if path is None:
        _resident.clear()
    else:
        _resident.pop(os.path.realpath(path), None)


def read_wav(path, engine, dtype):
    """(flat interleaved device tensor, samplerate, channels, frames) of a WAV file."""
    import torch
    key = os.path.realpath(path)
    hit = _resident.get(key)
    if hit is not None and _enabled():
        t, sr, ch, stamp = hit
        if t.dtype == dtype and t.device == engine.torch_device and os.path.exists(key) and _stamp(key) == stamp:
            _resident.move_to_end(key)
            return t, sr, ch, t.numel() // ch
        _resident.pop(key, None)
    data, sr = wavio.read(path)
    ch = 1 if data.ndim == 1 else data.shape[1]
    t = torch.as_tensor(np.ascontiguousarray(data), device=engine.torch_device).to(dtype).reshape(-1)
    return t, sr, ch, data.shape[0]


def write_wav(path, y, samplerate, channels, engine, pcm16=None):
    """Writes the flat interleaved device signal `y` in the default subtype.  `pcm16`: the int16
    samples when the producing kernel already made them (`y` then holds the read-back values)."""
    subtype = wavio.default_subtype()
    if subtype != "PCM_16":
        data = y.double().cpu().numpy()
        wavio.write(path, data if channels == 1 else data.reshape(-1, channels), samplerate, subtype)
        if subtype == "DOUBLE" and _enabled():
            _remember(path, y, samplerate, channels)
        return
    if pcm16 is None:
        pcm16, y = engine.pcm_encode(y, 16, dequantize=True)
    wavio.write_pcm16(path, pcm16.cpu().numpy(), samplerate, channels)
    if _enabled():
        _remember(path, y, samplerate, channels)


def _budget_bytes():
    return int(float(os.environ.get("ADN_DEVICE_FILES_MB", "4096")) * (1 << 20))


def _remember(path, t, samplerate, channels):
    key = os.path.realpath(path)
    _resident.pop(key, None)
    size = t.numel() * t.element_size()
    if size > _budget_bytes():
        return
    _resident[key] = (t, int(samplerate), int(channels), _stamp(key))
    while len(_resident) > _MAX_RESIDENT or \
            sum(v[0].numel() * v[0].element_size() for v in _resident.values()) > _budget_bytes():
        _resident.popitem(last=False)

"""Drop-in for the reference's ``denoise`` module: same names (``AudioDeNoise``, ``mad``),
same file-in / file-out behaviour, arithmetic on the MI355X through ``adn_hip``.

    >>> AudioDeNoise("input.wav").deNoise("input_denoised.wav")
    >>> AudioDeNoise("input.wav").generateNoiseProfile("input_noise_profile.wav")
"""
import os

import numpy as np

from adn_hip import wavio
from lib.noiseProfiler import NoiseProfiler


def _engine():
    from adn_hip.engine import get_engine
    return get_engine(int(os.environ.get("ADN_DEVICE", "0")))


def _dtype():
    import torch
    return torch.float32 if os.environ.get("ADN_DTYPE", "f64").lower() == "f32" else torch.float64


def mad(arr):
    """Median absolute deviation of the flattened array: median(|a - median(a)|), without
    the 0.6745 normalisation (reference denoise.py:16-23).  Runs the exact radix select on
    the GPU; masked entries of a numpy masked array are dropped first."""
    import torch
    a = np.ma.array(arr).compressed().astype(np.float64)
    e = _engine()
    t = torch.as_tensor(a, device=e.torch_device)
    return float(e.mad(t))


class AudioDeNoise:
    """Wavelet VisuShrink de-noiser (db4, periodization, level 2, soft threshold) that
    processes the file in blocks of 10 % of its length, each block with its own threshold."""

    def __init__(self, inputFile):
        self.__inputFile = inputFile
        self.__noiseProfile = None

    def deNoise(self, outputFile):
        """Reads the input file, de-noises every 10 % block independently on the GPU and
        writes the blocks back to back (same sample rate and channel count).  Blocks of
        odd length come back one sample longer, exactly like the reference."""
        from adn_hip import devfiles
        e = _engine()
        x, rate, channels, frames = devfiles.read_wav(self.__inputFile, e, _dtype())
        y = e.denoise_signal(x if channels == 1 else x.reshape(frames, channels), rate, wavelet="db4", level=2,
                             thr_mode="soft")
        devfiles.write_wav(outputFile, y.reshape(-1), rate, channels, e)

    def generateNoiseProfile(self, noiseFile):
        """Reads `noiseFile`, keeps its quiet 0.1 s windows (filling the gaps with
        neighbouring quiet windows) and overwrites the file with that noise track."""
        data, rate = wavio.read(noiseFile)
        self.__noiseProfile = NoiseProfiler(data)
        noiseSignal = self.__noiseProfile.getNoiseDataPredicted()
        wavio.write(noiseFile, np.asarray(noiseSignal), rate)

    def __del__(self):
        self.__noiseProfile = None

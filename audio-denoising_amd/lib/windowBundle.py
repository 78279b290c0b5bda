"""One analysis window of the noise profiler: raw rows, RMS, optional wavelet packet."""
import math

import numpy


class WaveletPacketLeaves:
    """Leaves of a full wavelet-packet tree (symmetric extension), computed on the GPU by
    repeated single-level analysis.  Supports what the profiler needs from
    pywt.WaveletPacket: ``get_level(level, order)`` paths and ``packet[path].data``."""

    class _Node:
        def __init__(self, path, data):
            self.path, self.data = path, data

    def __init__(self, data, wavelet, mode, maxlevel):
        import torch
        from adn_hip.engine import get_engine
        if mode not in ("symmetric", "sym", "per", "periodization"):
            raise NotImplementedError("extension mode %r" % mode)
        e = get_engine(0)
        self.maxlevel = maxlevel
        nodes = {"": torch.as_tensor(numpy.ascontiguousarray(data, dtype=numpy.float64), device=e.torch_device)}
        frontier = [""]
        for _ in range(maxlevel):
            nxt = []
            for p in frontier:
                ca, cd = e.wavedec(nodes[p], wavelet, 1, "sym" if mode.startswith("sym") else "per")
                nodes[p + "a"], nodes[p + "d"] = ca, cd
                nxt += [p + "a", p + "d"]
            frontier = nxt
        self._nodes = {p: v.cpu().numpy() for p, v in nodes.items()}

    def __getitem__(self, path):
        return self._Node(path, self._nodes[path])

    def get_level(self, level, order="natural"):
        paths = [""]
        for _ in range(level):
            if order == "freq":
                # Gray-code order: ascending frequency of the sub-bands
                paths = [p + c for i, p in enumerate(paths) for c in (("a", "d") if i % 2 == 0 else ("d", "a"))]
            else:
                paths = [p + c for p in paths for c in ("a", "d")]
        return [self[p] for p in paths]


class WindowBundle:
    def __init__(self, data, id):
        self.id = id
        self.data = data
        self.rms = None
        self.waveletPacket = None
        self.noiseWindow = None
        self.denoisedData = []
        self.dbName = None
        self.wlevels = None

    def extractWaveletPacket(self, dbName, wlevels):
        if self.waveletPacket is None:
            self.dbName, self.wlevels = dbName, wlevels
            self.waveletPacket = WaveletPacketLeaves(self.data, dbName, "symmetric", wlevels)
        return self.waveletPacket

    def getWaveletLeafData(self):
        from lib import waveletHelper
        return waveletHelper.waveletLeafData(self.waveletPacket)

    def setDenoisedData(self, denoisedData):
        self.denoisedData = denoisedData

    def getDenoisedData(self):
        return self.denoisedData

    def setNoiseWindow(self, window):
        self.noiseWindow = window

    def isBelowThreshold(self, threshold):
        return self.getRMS() < threshold

    def getData(self):
        return self.data

    def getRMS(self):
        """sqrt(sum(x^2) / rows); set in bulk by the profiler's GPU pass, else computed here."""
        if self.rms is None:
            self.rms = math.sqrt(float(numpy.sum(numpy.power(self.data, 2))) / len(self.data))
        return self.rms

    def getMA(self):
        return float(numpy.sum(numpy.abs(self.data))) / len(self.data)

    def getRMSasArray(self):
        return self.getRMS() * numpy.ones(len(self.data))

    @staticmethod
    def joinDenoisedData(windows):
        return [v for w in windows for v in w.denoisedData]

    @staticmethod
    def joinData(windows):
        return [v for w in windows for v in w.data]

    @staticmethod
    def joinNoiseData(windows):
        return [v for w in windows for v in w.noiseWindow.data]

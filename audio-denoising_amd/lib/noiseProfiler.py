"""RMS-window noise profiler with the reference's class surface (lib/noiseProfiler.py there).

The per-window energies come from one batched GPU pass (adn_window_rms); the window
classification and the gap-filling walk are cheap host logic on window INDICES - the
sample data is only gathered once at the end.
"""
import math

import numpy

from lib import windowBundle
from lib.linkedList import LinkedList


class NoiseProfiler:
    def __init__(self, x, timeWindow=0.1, sampleRate=44100, percentileLevel=95, wlevels=4, dbName='db8'):
        self.x = numpy.asarray(x)
        self.timeWindow = timeWindow
        self.windowSamples = int(timeWindow * sampleRate)
        self.wlevels = wlevels
        self.dbName = dbName
        self.windows = []
        self.noiseWindows = None
        self.signalWindows = None
        self.noiseLinked = LinkedList()
        self.signalLinked = LinkedList()
        self.percentileLevel = percentileLevel
        self.noiseData = None
        self.threshold = None
        self.extractWindows()
        print("Noise profiler finished")

    # -------------------------------------------------------------- windows
    def extractWindows(self):
        """Cut x into windows of windowSamples rows (the last one may be shorter) and
        compute every window's RMS in one GPU pass."""
        import torch
        from adn_hip.engine import get_engine
        n = len(self.x)
        n_win = math.ceil(n / self.windowSamples)
        self.windows = [windowBundle.WindowBundle(self.x[i * self.windowSamples:(i + 1) * self.windowSamples], i)
                        for i in range(n_win)]
        if n_win:
            e = get_engine(0)
            xd = torch.as_tensor(numpy.ascontiguousarray(self.x, dtype=numpy.float64), device=e.torch_device)
            rms = e.window_rms(xd, self.windowSamples).cpu().numpy()
            for w, r in zip(self.windows, rms):
                w.rms = float(r)

    def extractRMSthresholdFromWindows(self, percentileLevel):
        """RMS of the window at index floor(level/100 * n) of the windows sorted by RMS, loudest first."""
        if self.threshold is None:
            ranked = sorted((w.getRMS() for w in self.windows), reverse=True)
            self.threshold = ranked[math.floor(percentileLevel / 100 * len(ranked))]
        return self.threshold

    def extractSignalAndNoiseWindows(self, rmsThreshold):
        """A window is noise when its RMS is below the threshold plus 5 %."""
        if self.noiseWindows is not None and self.signalWindows is not None:
            return
        self.noiseWindows, self.signalWindows = [], []
        limit = rmsThreshold + 0.05 * rmsThreshold
        for w in self.windows:
            quiet = w.getRMS() < limit
            (self.noiseWindows if quiet else self.signalWindows).append(w)
            self.noiseLinked.append(w if quiet else None)
            self.signalLinked.append(None if quiet else w)

    # ------------------------------------------------------------ prediction
    @staticmethod
    def _circular(start, count, attr):
        """count nodes starting at `start`, stepping along `attr` while the neighbour holds a
        window and jumping back to `start` when the run of windows ends."""
        out, cur = [], start
        while True:
            out.append(cur)
            if len(out) == count:
                return out
            nxt = getattr(cur, attr)
            cur = nxt if (nxt is not None and nxt.data is not None) else start

    def _predict(self, node, count):
        prev = node.getPrevWithValidData()
        if prev is not None:
            return self._circular(prev, count, "prev")
        return self._circular(node.getNextWithValidData(), count, "next")

    def getNoiseDataPredicted(self):
        """Noise track of (about) the input's length: noise windows stay, every run of signal
        windows is replaced by as many neighbouring noise windows."""
        self.threshold = self.extractRMSthresholdFromWindows(self.percentileLevel)
        self.extractSignalAndNoiseWindows(self.threshold)
        order, gap, last = [], 0, None
        for node in self.noiseLinked.getAsList():
            if node.data is None:
                gap += 1
                continue
            last = node
            if gap:
                order += self._predict(node, gap)
                gap = 0
            order.append(node)
        if gap:
            order += self._predict(last, gap)
        parts = [n.data.data for n in order]
        self.cleanUp()
        if not parts:
            return []
        # a Python list like the reference's (one element per row: numpy scalars for mono
        # input, channel rows for multi-channel input)
        return list(numpy.concatenate(parts, axis=0))

    def getWindowsRMSasEnvelope(self):
        return numpy.concatenate([w.getRMSasArray() for w in self.windows]) if self.windows else numpy.array([])

    def getDataOrZeroFromPartialWindows(self, allWindows, partialWindows):
        keep = {id(w) for w in partialWindows}
        return numpy.concatenate([w.data if id(w) in keep else numpy.zeros(self.windowSamples) for w in allWindows])

    def extractWavelets(self):
        for w in self.windows:
            w.extractWaveletPacket(self.dbName, self.wlevels)

    def cleanUp(self):
        self.windows = None
        self.noiseData = None
        self.noiseLinked = None
        self.signalLinked = None
        self.signalWindows = None

    def drawOriginalVsNoiseAndSingal(self):
        raise NotImplementedError("plotting helpers of the reference are not part of the GPU hot path")

    plotWavelets = drawOriginalVsNoiseAndSingal

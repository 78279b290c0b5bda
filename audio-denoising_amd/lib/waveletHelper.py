"""Helpers over a wavelet-packet tree (drop-in for the reference's lib/waveletHelper.py).

`waveletPacket` is a lib.windowBundle.WaveletPacketLeaves (the GPU-computed stand-in for
pywt.WaveletPacket: same `maxlevel`, `get_level(level, order)` and `packet[path].data`).
"""


def waveletLeafData(waveletPacket):
    """All leaf bands of the deepest level, in ascending-frequency order, joined into one list
    (lib/waveletHelper.py:5-14 of the reference)."""
    leafData = []
    for node in waveletPacket.get_level(waveletPacket.maxlevel, "freq"):
        leafData.extend(waveletPacket[node.path].data)
    return leafData


def plotWavelets(wavelets):
    """Plot the leaf bands of several packets, one subplot per packet, bands side by side in
    natural order (lib/waveletHelper.py:17-33 of the reference).  Host-side matplotlib only."""
    import matplotlib.pyplot as plt
    plt.figure()
    paths = [node.path for node in wavelets[0].get_level(wavelets[0].maxlevel, "natural")]
    for row, packet in enumerate(wavelets, start=1):
        plt.subplot(len(wavelets), 1, row)
        for band, path in enumerate(paths):
            data = packet[path].data
            plt.plot(range(band * len(data), (band + 1) * len(data)), data)
    plt.show()

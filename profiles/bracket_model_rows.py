"""Model (CPU, numpy) of a bracket method for the per-bin statistics of the Wiener stage, the way the wavelet
stage's single-pass threshold statistics work: a strided sample of the frames gives, per bin, a value bracket
for the 10th percentile; one pass over all frames would count the values below the bracket and keep the ones
inside; the exact order statistic is then selected among those.  The script measures, on a 60 s benchmark clip,
which share of a row's values falls inside a bracket that holds the wanted rank with the usual 4.5 sigma margin,
and how often the bracket misses -- i.e. how much of the |Z|^2 matrix such a pass would still have to write.
    python profiles/bracket_model_rows.py"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "oracle"))
import adn_oracle as orc, synth

SR = 48000
x = synth.noisy_speech(60 * SR, SR, 42, 0)
S = np.abs(orc.stft(x, 2048, 512)) ** 2                      # (1025, 5622)
F, T = S.shape
q = 0.10
r0 = int(np.floor((T - 1) * q))
srt = np.sort(S, axis=1)
print("rows of %d frames, wanted rank %d (10th percentile)" % (T, r0))
for stride in (16, 8, 4, 2):
    sample = np.sort(S[:, ::stride], axis=1)
    m = sample.shape[1]
    delta = int(4.5 * np.sqrt(m * q * (1 - q))) + 3
    rs = int(r0 * m / T)
    lo = sample[:, max(rs - delta, 0)]
    hi = sample[:, min(rs + delta + 1, m - 1)]
    inside = np.sum((S >= lo[:, None]) & (S <= hi[:, None]), axis=1)
    below = np.sum(S < lo[:, None], axis=1)
    miss = np.sum((r0 < below) | (r0 + 1 >= below + inside))
    print("sample of every %2dth frame (%4d values, %4.1f %% more transforms): bracket holds %5.1f %% of a row on average "
          "(largest %5.1f %%), %d of %d rows missed" % (stride, m, 100.0 / stride, 100.0 * inside.mean() / T,
                                                        100.0 * inside.max() / T, miss, F))

"""One recording cut into time chunks (several devices, or one device chunk by chunk).

The reference loads a whole file per stage (src/spectral_subtract.py:75,82, src/wiener_filter.py:66,72)
and denoises it in blocks of 10 % of the file (denoise.py:76).  A recording too long for one device is
cut into time chunks; this module plans the chunks and runs the three stages on them so that the result
equals the whole-file result:

  * everything local in time (filter banks, STFT -> modify -> ISTFT) runs on a chunk plus a halo of
    neighbouring samples through the ordinary entry points, and the halo's output is dropped;
  * statistics that span the block / file are reduced over the chunks: the per-block MAD and the
    per-bin percentile by a distributed radix selection (summed digit histograms), the VAD-masked
    power sums and the frame count by a plain sum.

Every function takes the chunks THIS process holds and a `Reducer`: `LocalReducer` when one process
holds all chunks (they are then walked in lock step, which is also the validation mode for a
recording that is split across devices in production), `DistReducer` when every rank holds some
chunks and the partial results go through `torch.distributed` (RCCL on GPUs, gloo on CPU tensors).
"""
import math

import numpy as np
import torch


# ------------------------------------------------------------------ reducers
class LocalReducer:
    """All chunks live in this process: the reduction is over the list alone."""

    def sum(self, parts):
        out = parts[0].clone()
        for p in parts[1:]:
            out += p
        return out

    def max(self, parts):
        out = parts[0].clone()
        for p in parts[1:]:
            out = torch.maximum(out, p)
        return out


class DistReducer(LocalReducer):
    """Chunks are spread over the ranks of a process group: local reduction, then an all-reduce."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group

    def sum(self, parts):
        out = super().sum(parts)
        self.dist.all_reduce(out, op=self.dist.ReduceOp.SUM, group=self.group)
        return out

    def max(self, parts):
        out = super().max(parts)
        self.dist.all_reduce(out, op=self.dist.ReduceOp.MAX, group=self.group)
        return out


# ------------------------------------------------------------------ chunk plans
def plan_wavelet_chunks(block_len, n_chunks, level, taps):
    """Owned ranges [(start, stop)] of one block, aligned to 2^level, and the halo (samples on both sides,
    a multiple of 2^level that covers analysis + synthesis support over all levels)."""
    step = 1 << level
    if block_len % step:
        raise ValueError("a block is split into chunks only when its length is a multiple of 2^level")
    halo = -(-2 * (taps - 1) * (step - 1) // step) * step
    units = block_len // step
    edges = [units * i // n_chunks * step for i in range(n_chunks + 1)]
    return [(a, b) for a, b in zip(edges[:-1], edges[1:]) if b > a], halo


def wavelet_chunk_buffer(block, start, stop, halo):
    """[halo | owned | halo] of a block held as one tensor, the halo wrapping round the block's ends
    (periodization).  A production holder receives the halo from its neighbours instead."""
    n = block.numel()
    idx = torch.arange(start - halo, stop + halo, device=block.device) % n
    return block[idx]


def plan_stft_chunks(n_samples, n_chunks, n_fft=2048, hop=512):
    """Output-hop ranges [(h0, h1)] that partition the stage's output, and the total frame count."""
    nadd = (-(n_samples - n_fft) % hop) % n_fft
    frames = 1 + (n_samples + nadd - n_fft) // hop
    hops = frames - 1
    edges = [hops * i // n_chunks for i in range(n_chunks + 1)]
    return [(a, b) for a, b in zip(edges[:-1], edges[1:]) if b > a], frames


def stft_chunk_window(h0, h1, frames, n_samples, n_fft=2048, hop=512, reach=1):
    """Samples a holder needs for output hops [h0, h1) and where they sit in the sub-call's output.

    Output hop h is padded hop h + 2 (scipy's istft drops n_fft/2 at both ends) and sums frames h - 1 .. h + 2;
    `reach` = how many neighbouring frames the modification of a frame reads on each side (1 for the 3-frame
    time smoothing of the spectral subtraction, 0 for a gain that is constant in time) -- the sub-call's own
    first and last `reach` frames see its edges instead of their neighbours and must not feed a kept hop,
    unless they are the signal's edges too.
    Returns (sample_start, sample_stop, first kept output sample of the sub-call)."""
    f0 = max(h0 - 1 - reach, 0)
    f1 = min(h1 + 1 + reach, frames - 1)
    s0 = f0 * hop
    s1 = n_samples if f1 == frames - 1 else f1 * hop + n_fft
    return s0, s1, (h0 - f0) * hop


# ------------------------------------------------------------------ distributed order statistics
class OrderStatSearch:
    """k-th smallest value (0-based rank, per segment) of segments whose values are spread over chunks."""

    def __init__(self, engine, ranks, dtype):
        self.e, self.dtype = engine, dtype
        self.key_bits = 32 if dtype == torch.float32 else 64
        dev = engine.torch_device
        self.rank = torch.as_tensor(np.asarray(ranks, dtype=np.int64), device=dev).clone()
        self.prefix = torch.zeros_like(self.rank)
        self.done_bits = 0

    @property
    def finished(self):
        return self.done_bits >= self.key_bits

    def local_hist(self, vals, offs, lens, absdev=False, centers=None, max_len=None):
        hist = torch.empty((self.rank.numel(), 256), dtype=torch.int32, device=self.e.torch_device)
        self.e.select_hist(vals, offs, lens, self.prefix, self.done_bits, hist, absdev=absdev, centers=centers,
                           max_len=max_len)
        return hist

    def resolve(self, hist_total):
        self.e.select_resolve(hist_total.contiguous(), self.key_bits, self.done_bits, self.rank, self.prefix)
        self.done_bits += 8

    def values(self):
        return self.e.select_value(self.prefix, self.dtype)


def order_statistics(engine, parts, ranks_list, dtype, reducer, absdev=False, centers=None):
    """parts: list of (vals, offs, lens[, max_len]) per local chunk (device tensors; offs / lens int64 per segment;
    max_len = the longest segment of the chunk, saves a device round trip).  ranks_list: list of rank arrays, one
    rank per segment each; returns one float64 tensor of values per rank array.  All rank arrays descend together:
    the segments are simply listed once per rank array, so a median costs the rounds of ONE order statistic."""
    R = len(ranks_list)
    n_seg = len(ranks_list[0])
    ranks = np.concatenate([np.asarray(r, dtype=np.int64) for r in ranks_list])
    rep = []
    for part in parts:
        v, o, l = part[0], part[1], part[2]
        ml = part[3] if len(part) > 3 else None
        rep.append((v, o.repeat(R), l.repeat(R), ml))
    cen = None if centers is None else centers.repeat(R)
    s = OrderStatSearch(engine, ranks, dtype)
    while not s.finished:
        hists = [s.local_hist(v, o, l, absdev, cen, ml) for (v, o, l, ml) in rep]
        s.resolve(reducer.sum(hists))
    vals = s.values()
    return [vals[i * n_seg:(i + 1) * n_seg] for i in range(R)]


def _median_ranks(n):
    return (n - 1) // 2, n // 2


def _percentile_ranks(n, q):
    pos = (n - 1) * (q / 100.0)
    r0 = min(int(math.floor(pos)), n - 1)
    return r0, min(r0 + 1, n - 1)


# ------------------------------------------------------------------ phase 2: one block, many chunks
def wavelet_block_threshold(engine, chunk_bufs, owned_lens, halo, block_len, wavelet, reducer):
    """sigma = mad(cD1) of the whole block (denoise.py:16-23, 80) and the universal threshold (denoise.py:83)
    from the chunks this process holds.  Returns (sigma, thresh) as float64 device tensors of one value."""
    e = engine
    dtype = chunk_bufs[0].dtype
    parts = []
    for buf, owned in zip(chunk_bufs, owned_lens):
        cd1 = e.wavedec(buf, wavelet, 1, "per")[1][halo // 2: halo // 2 + owned // 2].contiguous()
        parts.append((cd1, e._i64([0]), e._i64([cd1.numel()]), cd1.numel()))
    n = block_len // 2
    r0, r1 = _median_ranks(n)
    a, b = order_statistics(e, parts, [[r0], [r1]], dtype, reducer)
    med = e.select_combine(a, b, n, None, dtype)
    a, b = order_statistics(e, parts, [[r0], [r1]], dtype, reducer, absdev=True, centers=med)
    sigma = e.select_combine(a, b, n, None, dtype)
    thresh = sigma * math.sqrt(2.0 * math.log(float(block_len)))
    return sigma, thresh


def wavelet_block_apply(engine, chunk_bufs, owned_lens, halo, thresh, wavelet, level, thr_mode="soft"):
    """Shrink every chunk with the block's threshold and keep the owned samples."""
    e = engine
    outs = []
    for buf, owned in zip(chunk_bufs, owned_lens):
        y = torch.empty_like(buf)
        e.wavelet_shrink(buf, e._i64([0]), e._i64([buf.numel()]), y, e._i64([0]), thresh, wavelet=wavelet, level=level,
                         thr_mode=thr_mode, max_frames=buf.numel())
        outs.append(y[halo: halo + owned].clone())
    return outs


# ------------------------------------------------------------------ phases 2b / 2c
def spectral_subtract_chunk(engine, sub, keep_from, n_keep, noise_mag, **kw):
    """Phase 2b on one chunk's samples `sub` (mono, or interleaved with n_ch in kw); returns the chunk's
    n_keep output samples."""
    n_ch = kw.get("n_ch", 1)
    y, _ = engine.spectral_subtract(sub, [sub.numel() // n_ch], noise_mag, **kw)
    return y[keep_from: keep_from + n_keep].clone()


def wiener_chunk_stats(engine, sub, frames_owned, vad_buf, vad_base, n_total, t0, sample_rate, n_fft=2048, hop=512,
                       update_ms=20, vad_guided=True):
    """Statistics inputs of one chunk: the power rows of its owned frames and, with the VAD, the masked sums.
    `sub` starts at frame t0's first sample; `vad_buf` holds the signal from sample `vad_base` on."""
    e = engine
    S, fr = e.stft_rows(sub, [sub.numel()], n_fft=n_fft, hop=hop, padded=True, power=True)
    S = S[0]
    if int(fr[0]) < frames_owned:
        raise ValueError("the chunk's samples give %d frames, %d are owned" % (int(fr[0]), frames_owned))
    sums = count = None
    if vad_guided:
        vad = e.vad_frames(vad_buf, vad_base, n_total, sample_rate, t0, frames_owned, update_ms=update_ms, hop=hop)
        sums, count = e.rows_masked_sum(S, frames_owned, vad)
    F, ts = S.shape
    offs = e._i64(np.arange(F, dtype=np.int64) * ts)
    lens = e._i64(np.full(F, frames_owned, dtype=np.int64))
    return {"S": S, "part": (S.view(-1), offs, lens, frames_owned), "sums": sums, "count": count}


def wiener_gain_from_chunks(engine, stats, frames_total, dtype, reducer, kernel_bins=5, vad_guided=True):
    """The file-wide per-bin gain (src/wiener_filter.py:84-115) from the chunks' statistics."""
    e = engine
    parts = [s["part"] for s in stats]
    F = stats[0]["S"].shape[0]
    r0, r1 = _percentile_ranks(frames_total, 10.0)
    a, b = order_statistics(e, parts, [[r0] * F, [r1] * F], dtype, reducer)
    noise = e.select_combine(a, b, frames_total, 10.0, dtype)
    if vad_guided:
        total = reducer.sum([s["sums"] for s in stats])
        count = reducer.sum([s["count"] for s in stats])
        speech = (total / torch.clamp(count, min=1).to(torch.float64)).to(dtype).to(torch.float64)
    else:
        m0, m1 = _median_ranks(frames_total)
        a, b = order_statistics(e, parts, [[m0] * F, [m1] * F], dtype, reducer)
        speech = e.select_combine(a, b, frames_total, None, dtype)
    return e.wiener_gain(noise.view(1, F), speech.view(1, F), dtype, kernel_bins=kernel_bins)[0]


def wiener_apply_chunk(engine, sub, keep_from, n_keep, gain, n_fft=2048, hop=512):
    y, _ = engine.stft_apply_gain(sub, [sub.numel()], gain, n_fft=n_fft, hop=hop)
    return y[keep_from: keep_from + n_keep].clone()

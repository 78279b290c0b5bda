"""One recording spread over the ranks of a process group, end to end.

Every rank holds one contiguous time range of the recording (aligned to the STFT hop, which is also a
multiple of 2^level of the wavelet stage) and runs the three stages of the core pipeline on it
(denoise.py:59-90, src/spectral_subtract.py:75-109, src/wiener_filter.py:66-129).  The samples a rank
needs from its neighbours (halos; for a 10 % block that wraps, its other end) travel point to point;
the statistics that span a block or the file are reduced over the ranks (adn_hip/chunked.py).  The
result on every rank equals its range of the one-call result: bit for bit for the wavelet stage and
the spectral subtraction, to rounding of the VAD-masked mean's sum order for the Wiener stage.

Block lengths (10 % of the file) must be even, and blocks that straddle two ranks a multiple of
2^level long: the chunks of a shared block are transformed at its own periodization phase.
"""
import math

import numpy as np
import torch
import torch.distributed as dist

from . import chunked as ck

HOP, NFFT = 512, 2048


def partition(n, world, align=HOP):
    """Contiguous ranges [(start, stop)] of n samples over `world` ranks, starts aligned to `align`."""
    units = -(-n // align)
    edges = [min(units * r // world * align, n) for r in range(world)] + [n]
    return [(edges[r], edges[r + 1]) for r in range(world)]


class SpreadSignal:
    """A 1-D signal of which every rank holds one contiguous range (`ranges[rank]`) in `local`."""

    def __init__(self, local, ranges, rank, group=None):
        self.local, self.ranges, self.rank, self.group = local, list(ranges), rank, group
        s, e = self.ranges[rank]
        if local.numel() != e - s:
            raise ValueError("rank %d holds %d samples, its range has %d" % (rank, local.numel(), e - s))
        self.n = self.ranges[-1][1]
        self.world = len(self.ranges)
        # gloo moves host memory only: device tensors are staged through the host for it
        self._stage = local.is_cuda and dist.is_initialized() and dist.get_backend(group) == "gloo"

    def fetch(self, wants_of):
        """wants_of(r) -> list of (a, b): the global sample intervals rank r wants, the same answer on every rank
        (a plan, not a request: nobody has to be asked).  Returns this rank's intervals as tensors."""
        mine = wants_of(self.rank)
        out = [torch.empty(b - a, dtype=self.local.dtype, device=self.local.device) for a, b in mine]
        s_me = self.ranges[self.rank][0]
        ops, landing = [], []
        for q in range(self.world):                              # the same walk on every rank: ordered per pair
            for idx, (a, b) in enumerate(wants_of(q)):
                for o, (s, e) in enumerate(self.ranges):
                    lo, hi = max(a, s), min(b, e)
                    if lo >= hi:
                        continue
                    if q == self.rank and o == self.rank:
                        out[idx][lo - a:hi - a] = self.local[lo - s_me:hi - s_me]
                    elif q == self.rank:
                        dst = out[idx][lo - a:hi - a]
                        buf = torch.empty(hi - lo, dtype=dst.dtype) if self._stage else dst
                        ops.append(dist.P2POp(dist.irecv, buf, o, self.group))
                        if self._stage:
                            landing.append((dst, buf))
                    elif o == self.rank:
                        src = self.local[lo - s_me:hi - s_me]
                        ops.append(dist.P2POp(dist.isend, src.cpu() if self._stage else src.contiguous(), q, self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for dst, buf in landing:
            dst.copy_(buf)
        return out


class SpreadPipeline:
    """The wavelet stage, spectral subtraction and Wiener filter of one mono recording of n samples spread over the
    ranks of `group`.  Every method takes this rank's range of the stage's input and returns its range of the output
    (`ranges_*` tell which)."""

    def __init__(self, engine, n, sample_rate, rank, world, group=None, wavelet="db4", level=2, thr_mode="soft"):
        self.e, self.n, self.sr, self.rank, self.world, self.group = engine, n, sample_rate, rank, world, group
        self.wavelet, self.level, self.thr_mode = wavelet, level, thr_mode
        self.taps = 8 if wavelet == "db4" else 16
        if HOP % (1 << level):
            raise ValueError("level %d: rank ranges are aligned to the hop, which must be a multiple of 2^level" % level)
        self.red = ck.DistReducer(group) if world > 1 else ck.LocalReducer()
        self.ranges_in = partition(n, world)

    # ---------------------------------------------------------------- wavelet stage
    def _blocks(self):
        bs = self.e.denoise_block_frames(self.sr, self.n)
        return [(b0, min(b0 + bs, self.n)) for b0 in range(0, self.n, bs)]

    def wavelet_stage(self, x_local):
        e, step = self.e, 1 << self.level
        s, t = self.ranges_in[self.rank]
        sig = SpreadSignal(x_local, self.ranges_in, self.rank, self.group)
        blocks = self._blocks()
        if any((b1 - b0) & 1 for b0, b1 in blocks):
            raise ValueError("time-chunk mode needs blocks of even length (odd blocks come back one sample longer)")
        halo = -(-2 * (self.taps - 1) * (step - 1) // step) * step

        def owner_ranks(b0, b1):
            return [r for r, (rs, re) in enumerate(self.ranges_in) if max(rs, b0) < min(re, b1)]

        shared = [j for j, (b0, b1) in enumerate(blocks) if len(owner_ranks(b0, b1)) > 1]
        for j in shared:
            b0, b1 = blocks[j]
            if (b1 - b0) % step or b0 % step:
                raise ValueError("block %d straddles ranks but its length / start is not a multiple of 2^level" % j)

        def part_of(r, j):
            rs, re = self.ranges_in[r]
            b0, b1 = blocks[j]
            return max(rs, b0), min(re, b1)

        def wants_of(r):
            """[halo | part | halo] of every shared block rank r holds a part of, the halo wrapping inside the block."""
            out = []
            for j in shared:
                p0, p1 = part_of(r, j)
                if p0 >= p1:
                    continue
                b0, b1 = blocks[j]
                L = b1 - b0
                lo, hi = p0 - halo, p1 + halo                    # in samples, may leave the block on either side
                if halo > L:
                    raise ValueError("block shorter than the halo")
                if lo < b0:
                    out.append((lo + L, b1))
                out.append((max(lo, b0), min(hi, b1)))
                if hi > b1:
                    out.append((b0, hi - L))
            return out

        pieces = sig.fetch(wants_of)
        # reassemble the buffers of my parts (the walk of wants_of, again)
        bufs, parts, k = [], [], 0
        for j in shared:
            p0, p1 = part_of(self.rank, j)
            if p0 >= p1:
                continue
            b0, b1 = blocks[j]
            cnt = 1 + (p0 - halo < b0) + (p1 + halo > b1)
            bufs.append(torch.cat(pieces[k:k + cnt]))
            parts.append((j, p0, p1))
            k += cnt
        # thresholds of the shared blocks: one joint descent over all of them, every rank takes part
        thresh_of = {}
        if shared:
            dtype = x_local.dtype
            cds = []
            for buf, (j, p0, p1) in zip(bufs, parts):
                cd1 = e.wavedec(buf, self.wavelet, 1, "per")[1]
                cds.append((j, cd1[halo // 2: halo // 2 + (p1 - p0) // 2]))
            lens = np.zeros(len(shared), dtype=np.int64)
            offs = np.zeros(len(shared), dtype=np.int64)
            flat, pos = [], 0
            for j, c in cds:
                i = shared.index(j)
                offs[i], lens[i] = pos, c.numel()
                flat.append(c)
                pos += c.numel()
            vals = torch.cat(flat) if flat else torch.zeros(1, dtype=dtype, device=x_local.device)
            part = (vals, e._i64(offs), e._i64(lens), int(lens.max()) if len(lens) else 0)
            n_c = [(blocks[j][1] - blocks[j][0]) // 2 for j in shared]
            r0 = [(m - 1) // 2 for m in n_c]
            r1 = [m // 2 for m in n_c]
            if len(set(n_c)) != 1:
                raise ValueError("shared blocks of different lengths")      # (cannot happen: only whole 10 % blocks are shared)
            a, b = ck.order_statistics(e, [part], [r0, r1], dtype, self.red)
            med = e.select_combine(a, b, n_c[0], None, dtype)
            a, b = ck.order_statistics(e, [part], [r0, r1], dtype, self.red, absdev=True, centers=med)
            sigma = e.select_combine(a, b, n_c[0], None, dtype)
            for i, j in enumerate(shared):
                thresh_of[j] = sigma[i:i + 1] * math.sqrt(2.0 * math.log(float(blocks[j][1] - blocks[j][0])))
        # output: my range, block by block
        y = torch.empty_like(x_local)
        whole = [(b0, b1) for j, (b0, b1) in enumerate(blocks) if j not in shared and s <= b0 and b1 <= t]
        if whole:
            off = np.array([b0 - s for b0, _ in whole], dtype=np.int64)
            ln = np.array([b1 - b0 for b0, b1 in whole], dtype=np.int64)
            e.wavelet_denoise(x_local, e._i64(off), e._i64(ln), y, e._i64(off), wavelet=self.wavelet, level=self.level,
                              thr_mode=self.thr_mode, max_frames=int(ln.max()))
        for buf, (j, p0, p1) in zip(bufs, parts):
            out = torch.empty_like(buf)
            e.wavelet_shrink(buf, e._i64([0]), e._i64([buf.numel()]), out, e._i64([0]), thresh_of[j], wavelet=self.wavelet,
                             level=self.level, thr_mode=self.thr_mode, max_frames=buf.numel())
            y[p0 - s:p1 - s] = out[halo:halo + (p1 - p0)]
        return y

    # ---------------------------------------------------------------- STFT stages
    def _out_ranges(self, ranges, n_in):
        """Ownership of a stage's output: the input's ranges clipped to the output length (whole hops)."""
        hops, frames = ck.plan_stft_chunks(n_in, 1)
        L = (frames - 1) * HOP
        return [(min(a, L), min(b, L)) for a, b in ranges], frames, L

    def spectral_subtract(self, x_local, ranges, noise_mag, **kw):
        """x_local: this rank's range of the stage's input (ranges: every rank's).  Returns (y_local, ranges_out)."""
        e = self.e
        if kw.get("n_fft", NFFT) != NFFT or kw.get("hop", HOP) != HOP:
            raise ValueError("a recording spread over ranks is cut at multiples of hop %d for n_fft %d: other STFT "
                             "shapes are not supported here" % (HOP, NFFT))
        # frames on either side that the time smoothing of a frame reads (src/spectral_subtract.py:39-43:
        # uniform_filter over smooth_time frames): the sub-call's halo must hold them
        reach = (int(kw.get("smooth_time", 3)) - 1) // 2
        n_in = ranges[-1][1]
        out_ranges, frames, _ = self._out_ranges(ranges, n_in)
        sig = SpreadSignal(x_local, ranges, self.rank, self.group)

        def win(r):
            a, b = out_ranges[r]
            return ck.stft_chunk_window(a // HOP, b // HOP, frames, n_in, reach=reach) if b > a else None

        (sub,) = sig.fetch(lambda r: [win(r)[:2]] if win(r) else []) or (None,)
        a, b = out_ranges[self.rank]
        if b <= a:
            return x_local[:0].clone(), out_ranges
        return ck.spectral_subtract_chunk(e, sub, win(self.rank)[2], b - a, noise_mag, **kw), out_ranges

    def wiener(self, x_local, ranges, update_ms=20, kernel_bins=5, vad_guided=True):
        e, dtype = self.e, x_local.dtype
        n_in = ranges[-1][1]
        out_ranges, frames, _ = self._out_ranges(ranges, n_in)
        sig = SpreadSignal(x_local, ranges, self.rank, self.group)
        fl = int(self.sr * update_ms / 1000.0)
        last = max(r for r, (a, b) in enumerate(out_ranges) if b > a)

        def stat_frames(r):                                      # frames whose statistics rank r owns
            a, b = out_ranges[r]
            if b <= a:
                return None
            return a // HOP, (frames if r == last else b // HOP)

        def stat_wants(r):
            f = stat_frames(r)
            if f is None:
                return []
            t0, t1 = f
            s1 = n_in if t1 == frames else (t1 - 1) * HOP + NFFT
            vb = max((t0 * HOP // fl - 5) * fl, 0) if vad_guided else t0 * HOP
            return [(min(vb, t0 * HOP), s1)]

        got = sig.fetch(stat_wants)
        f = stat_frames(self.rank)
        F = NFFT // 2 + 1
        if f is not None:
            t0, t1 = f
            (lo, hi), buf = stat_wants(self.rank)[0], got[0]
            st = ck.wiener_chunk_stats(e, buf[t0 * HOP - lo:], t1 - t0, buf, lo, n_in, t0, self.sr, update_ms=update_ms,
                                       vad_guided=vad_guided)
        else:                                                    # a rank without frames still joins every reduction
            z = torch.zeros(1, dtype=dtype, device=x_local.device)
            zi = e._i64(np.zeros(F, dtype=np.int64))
            st = {"S": z.view(1, 1).expand(F, 1), "part": (z, zi, zi, 0),
                  "sums": torch.zeros(F, dtype=torch.float64, device=x_local.device),
                  "count": torch.zeros(1, dtype=torch.int64, device=x_local.device)}
        gain = ck.wiener_gain_from_chunks(e, [st], frames, dtype, self.red, kernel_bins=kernel_bins, vad_guided=vad_guided)

        def win(r):
            a, b = out_ranges[r]
            return ck.stft_chunk_window(a // HOP, b // HOP, frames, n_in, reach=0) if b > a else None

        (sub,) = sig.fetch(lambda r: [win(r)[:2]] if win(r) else []) or (None,)
        a, b = out_ranges[self.rank]
        if b <= a:
            return x_local[:0].clone(), out_ranges, gain
        return ck.wiener_apply_chunk(e, sub, win(self.rank)[2], b - a, gain), out_ranges, gain

#!/usr/bin/env python
"""Benchmark of the MI355X denoising hot path.

One "step" = one pass of the core chain (phase 1 noise profile, phase 2 wavelet,
phase 2b spectral subtraction, phase 2c Wiener) over one batch of synthetic
clips per GPU: BASELINE.json config 2 (60 s, 48 kHz, stereo clip + 5 s noise sample,
n_fft 2048, hop 512), `--clips` independent clips per rank.  Inputs are generated
on the device before the timed region; every stage reads its input from HBM and
writes its output to HBM like the reference's stage files.

Multi-GPU: clips are independent, so ranks shard them with no data-path
collective (weak scaling: the per-GPU batch is fixed).  One rank per GPU, launched
by torch.distributed.run; RCCL is used only for the barrier / max-time reduction.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "audio-denoising_amd"))

SR = 48000
CLIP_S = 60
NOISE_S = 5
N_FFT, HOP = 2048, 512
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--clips", type=int, default=64, help="clips per GPU per step")
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="process-group backend; gloo lets several ranks rehearse on one GPU")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline time budget")
    return ap.parse_args()


class Workload:
    """Device buffers and descriptors of one rank's batch (built once, outside the timed region)."""

    def __init__(self, eng, clips, dtype, rank):
        self.e, self.B, self.dtype = eng, clips, dtype
        dev = eng.torch_device
        n, nn = CLIP_S * SR, NOISE_S * SR
        self.n, self.nn = n, nn
        # stereo clips, interleaved (frames, 2); generated in fp64 and rounded once
        x = torch.empty((clips, n, 2), dtype=dtype, device=dev)
        tmp = torch.empty(n, dtype=torch.float64, device=dev)
        for c in range(clips):
            gid = rank * clips + c
            eng.synth_noisy_speech(n, SR, seed=42, stream=2 * gid, out=tmp)
            x[c, :, 0] = tmp.to(dtype)
            eng.synth_noisy_speech(n, SR, seed=42, stream=2 * gid + 1, speech_gain=0.8, out=tmp)
            x[c, :, 1] = tmp.to(dtype)
        self.x = x.view(-1)
        noise = torch.empty((clips, nn), dtype=dtype, device=dev)
        tn = torch.empty(nn, dtype=torch.float64, device=dev)
        for c in range(clips):
            eng.synth_noisy_speech(nn, SR, seed=42, stream=100000 + rank * clips + c, speech_gain=0.0, out=tn)
            noise[c] = tn.to(dtype)
        self.noise = noise.view(-1)
        # phase 2 blocks: 10 % of the file per block (denoise.py:76)
        bs = eng.denoise_block_frames(SR, n)
        starts = np.arange(0, n, bs, dtype=np.int64)
        lens = np.minimum(bs, n - starts)
        self.blk_off = np.concatenate([c * n * 2 + starts * 2 for c in range(clips)])
        self.blk_len = np.tile(lens, clips)
        self.blk_max = int(lens.max())
        self.y2 = torch.empty_like(self.x)
        self.l2b = eng.istft_out_samples(n, N_FFT, HOP)
        self.l2c = eng.istft_out_samples(self.l2b, N_FFT, HOP)
        self.y2b = torch.empty(clips * self.l2b, dtype=dtype, device=dev)
        self.y2c = torch.empty(clips * self.l2c, dtype=dtype, device=dev)
        self.clip_off = np.arange(clips, dtype=np.int64) * n * 2
        self.y2b_off = np.arange(clips, dtype=np.int64) * self.l2b
        self.y2c_off = np.arange(clips, dtype=np.int64) * self.l2c
        self.noise_lens = [nn] * clips
        self.d_blk_off, self.d_blk_len = eng._i64(self.blk_off), eng._i64(self.blk_len)

    # each stage is one C-ABI call
    def phase1(self):
        self.base = self.e.noise_psd(self.noise, self.noise_lens, n_fft=N_FFT, hop=HOP)

    def phase2(self):
        self.e.wavelet_denoise(self.x, self.d_blk_off, self.d_blk_len, self.y2, self.d_blk_off, n_ch=2,
                               max_frames=self.blk_max)

    def phase2b(self):
        self.e.spectral_subtract(self.y2, [self.n] * self.B, self.base, n_ch=2, n_fft=N_FFT, hop=HOP,
                                 offsets=self.clip_off, out=self.y2b, out_offsets=self.y2b_off)

    def phase2c(self):
        self.e.wiener(self.y2b, [self.l2b] * self.B, SR, n_fft=N_FFT, hop=HOP, offsets=self.y2b_off,
                      out=self.y2c, out_offsets=self.y2c_off)

    STAGES = ("phase1", "phase2", "phase2b", "phase2c")

    def step(self, events=None):
        for name in self.STAGES:
            if events is not None:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
            getattr(self, name)()
            if events is not None:
                b.record()
                events.setdefault(name, []).append((a, b))

    def algorithmic_bytes(self):
        """Compulsory HBM bytes per step and stage with the stage interface preserved: every stage
        reads its input once and writes its output once; B bytes per value.  Phase 2: 2*B per
        sample-frame-channel; phase 2b and 2c: 2*B per mono sample; phase 1: 1*B per noise sample.
        (Phase 2b actually reads a stereo input, 3*B per frame in total: reported separately as
        `io_bytes`, not used for the roofline fraction.)"""
        bts = 4 if self.dtype == torch.float32 else 8
        return {"phase1": self.B * self.nn * bts,
                "phase2": self.B * self.n * 2 * 2 * bts,
                "phase2b": self.B * self.n * 2 * bts,
                "phase2c": self.B * self.l2b * 2 * bts}

    def io_bytes_2b(self):
        bts = 4 if self.dtype == torch.float32 else 8
        return self.B * (self.n * 2 + self.l2b) * bts


def measured_traffic(kernel_key, clips):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes
    (profiles/r1_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE runs of this
    script, FETCH_SIZE doubled as gfx950 requires), scaled from the profiled batch to this one;
    None when no profile is committed for this kernel."""
    path = os.path.join(REPO, "profiles", "r1_pmc_traffic.json")
    try:
        prof = json.load(open(path))
        k = prof["kernels"][kernel_key]
        return k["hbm_bytes_per_clip"] * clips
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(budget_s):
    """The CPU restatement of the same chain (oracle/adn_oracle.py, numpy, one thread) on a
    bounded sample of the workload: whole 60 s stereo clips + 5 s noise until the budget is spent."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import adn_oracle as orc
    import synth
    n, nn = CLIP_S * SR, NOISE_S * SR
    x = synth.stereo_noisy_speech(n, SR, 42, 0)
    noise = synth.white_noise(nn, 42, 100000)
    done, t0 = 0, time.perf_counter()
    while True:
        base, _ = orc.noise_profile(noise)
        y2, _, _ = orc.denoise_signal(x, SR)
        y2b = orc.spectral_subtract(y2, base.astype(np.float32).astype(np.float64))
        orc.wiener(y2b, SR)
        done += 1
        el = time.perf_counter() - t0
        if el >= budget_s or done >= 16:
            break
    return {"value": done * CLIP_S / el, "unit": "audio-sec/sec", "cores": 1, "kind": "port",
            "sample": "%d x (60 s 48 kHz stereo clip + 5 s noise), fp64 numpy restatement, %.1f s of CPU work"
                      % (done, el)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.backend == "gloo":
        local = local % max(torch.cuda.device_count(), 1)       # rehearsal: ranks may share a GPU
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    else:
        dist = None
    torch.cuda.set_device(local)
    from adn_hip import get_engine
    eng = get_engine(local)
    dtype = torch.float32 if args.dtype == "f32" else torch.float64
    wl = Workload(eng, args.clips, dtype, rank)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        wl.step()
    barrier()
    events = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step(events)
    barrier()
    elapsed = time.perf_counter() - t0
    from adn_hip import dist as adist
    elapsed = adist.max_over_ranks(elapsed, device="cuda" if args.backend == "nccl" else "cpu")

    if rank == 0:
        ms = {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in events.items()}
        alg = wl.algorithmic_bytes()
        stages = {k: {"ms": round(ms[k], 4), "algorithmic_bytes": alg[k],
                      "GBps": round(alg[k] / ms[k] / 1e6, 1),
                      "hbm_frac": round(alg[k] / ms[k] / 1e6 / HBM_PEAK_GBS, 4)} for k in ms}
        # phase 2b is ONE launch of the fused STFT kernel per C-ABI call, so the event pair
        # around the call times exactly that kernel: the dominant kernel of the chain
        dom = "phase2b"
        chain_bytes = sum(alg.values())
        chain_ms = sum(ms.values())
        audio_s = args.clips * world * CLIP_S * args.steps
        out = {
            "metric": "audio-sec/sec (RTF) for 48 kHz denoise; % of HBM roofline",
            "value": audio_s / elapsed,
            "unit": "audio-sec/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": "core_pipeline phases 1/2/2b/2c on 60 s 48 kHz stereo clips + 5 s noise "
                                   "samples, STFT n_fft=2048 hop=512 (BASELINE.json config 2), %d clips per GPU per step"
                                   % args.clips,
                       "clips_per_gpu": args.clips, "sharding": "clips round-robin over ranks, no collective"},
            "roofline": {"bound": "hbm", "kernel": "adn::fast::k_apply_fast<8, SUBTRACT, false> (phase 2b, one launch per step)"
                         if args.dtype == "f32" else "adn::k_stft_apply<double, 3, SUBTRACT> (phase 2b)",
                         "achieved": stages[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": stages[dom]["hbm_frac"],
                         "traffic": measured_traffic("void adn::fast::k_apply_fast<8, 0, false>", args.clips)
                         if args.dtype == "f32" else None,
                         "traffic_unit": "bytes per launch", "io_bytes_per_launch": wl.io_bytes_2b(),
                         "algorithmic_bytes_per_launch": alg[dom], "avg_launch_ms": stages[dom]["ms"]},
            "chain": {"ms": round(chain_ms, 4), "algorithmic_bytes": chain_bytes,
                      "GBps": round(chain_bytes / chain_ms / 1e6, 1),
                      "hbm_frac": round(chain_bytes / chain_ms / 1e6 / HBM_PEAK_GBS, 4)},
            "stages": stages,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""world_size-2 gloo tests of the multi-GPU plumbing (shard assignment, profile broadcast,
max-over-ranks timing) on CPU tensors."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(REPO, "audio-denoising_amd"))
    from adn_hip import dist as adist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        idx = adist.shard_indices(11, rank, world)
        prof = torch.arange(1025, dtype=torch.float64) * 0.5 if rank == 0 else None
        got = adist.broadcast_profile(prof, 1025, src=0, dtype=torch.float64)
        tmax = adist.max_over_ranks(1.0 + rank)
        tsum = adist.sum_over_ranks(float(len(idx)))
        np.savez(os.path.join(out_dir, "r%d.npz" % rank), idx=idx, prof=got.numpy(), tmax=tmax, tsum=tsum)
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_and_profile_broadcast(tmp_path):
    world, port = 2, 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(str(tmp_path / ("r%d.npz" % k))) for k in range(world)]
    all_idx = np.sort(np.concatenate([x["idx"] for x in r]))
    assert np.array_equal(all_idx, np.arange(11))                    # every clip exactly once
    assert abs(len(r[0]["idx"]) - len(r[1]["idx"])) <= 1               # balanced
    for x in r:
        assert np.array_equal(x["prof"], np.arange(1025) * 0.5)       # same profile everywhere
        assert float(x["tmax"]) == 2.0 and float(x["tsum"]) == 11.0


def test_single_process_passthrough():
    sys.path.insert(0, os.path.join(REPO, "audio-denoising_amd"))
    from adn_hip import dist as adist
    assert not adist.is_distributed()
    assert adist.max_over_ranks(3.5) == 3.5
    p = torch.ones(4)
    assert adist.broadcast_profile(p, 4) is p
    assert list(adist.shard_indices(5, 1, 2)) == [1, 3]


def _run_bench(*flags):
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *flags], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    return out, (json.loads(lines[-1]) if lines else None)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no rendezvous in the environment starts two ranks itself
    (host rehearsal: no device work) and prints one line that says n_gpus = 2."""
    out, line = _run_bench("--gpus", "2", "--backend", "gloo", "--rehearse-host", "--steps", "3", "--warmup", "1")
    assert out.returncode == 0, out.stderr[-2000:]
    assert line is not None and line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    assert line["scaling"] == "weak" and "rehearsal" in line["data"]
    assert len([ln for ln in out.stdout.splitlines() if ln.startswith("{")]) == 1     # rank 0 only


def test_bench_single_rank_line_and_world_mismatch():
    out, line = _run_bench("--rehearse-host", "--steps", "2", "--warmup", "0")
    assert out.returncode == 0 and line["n_gpus"] == 1
    # a rendezvous that disagrees with --gpus is an error, not a silent 1-rank run
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    bad = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "4", "--rehearse-host"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE" in bad.stderr

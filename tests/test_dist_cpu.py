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

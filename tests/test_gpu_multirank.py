"""Two real ranks on ONE GPU (gloo; both processes share the device): the benchmark's multi-rank path with device
work, and one recording spread over ranks end to end.  A multi-GPU node runs the same code with one rank per GPU
over RCCL; no such node was available to these tests."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(300)
def test_bench_two_ranks_share_one_gpu():
    """`python bench.py --gpus 2 --backend gloo` as a fresh child process: it starts its own two ranks, both do
    the device work of their shard, rank 0 prints one line with the whole job's rate."""
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo", "--clips", "4",
                          "--steps", "2", "--warmup", "1", "--no-secondary"], env=_clean_env(), stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1
    assert line["value"] > 0 and line["data"] == "synthetic" and line["scaling"] == "weak"
    # whole-job rate: both ranks' clips over the slower rank's time
    assert abs(line["value"] - 2 * 4 * 60 * 2 / (line["ms_per_step"] * 2 / 1e3)) <= 1e-6 * line["value"]
    assert "roofline" in line and line["roofline"]["frac"] > 0


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,seconds,dtype", [(2, 37, "f32"), (3, 31, "f64")])
def test_recording_spread_over_ranks(world, seconds, dtype):
    """One recording spread over `world` ranks (halos exchanged point to point, the block / file statistics
    reduced over the ranks): on every rank the wavelet stage and the spectral subtraction equal the one-call
    results bit for bit, the Wiener output within the tolerance of its reordered voice-activity sum."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(REPO, "profiles", "rehearse_spread.py"),
           str(seconds), dtype]
    out = subprocess.run(cmd, env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    res = json.loads(lines[-1])
    assert len(res["ranks"]) == world
    tol = 1e-12 if dtype == "f64" else 2e-6
    for r in res["ranks"]:
        assert r["wavelet_equal"] and r["subtract_equal"], r
        assert r["wiener_max_abs_diff"] <= tol and r["gain_max_abs_diff"] <= tol, r

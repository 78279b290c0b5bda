"""One recording spread over several ranks, end to end, on ONE GPU: the ranks share the device and talk over gloo
(host-staged point-to-point halos, all-reduced statistics); rank 0 also runs the three stages in one call each on the
whole recording and every rank's range of the results is compared with it.  Start with
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29611 \
        profiles/rehearse_spread.py [seconds=100] [f64|f32]
(on a multi-GPU node the same code runs one rank per GPU over RCCL: backend nccl, no staging)."""
import json, os, sys, time
import numpy as np
import torch
import torch.distributed as dist
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "audio-denoising_amd"))
from adn_hip import Engine, spread

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
dt = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else torch.float64
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
local = int(os.environ.get("LOCAL_RANK", "0"))
if torch.cuda.device_count() >= world:                         # one GPU per rank: RCCL, halos device to device
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    e, backend = Engine(local), "nccl, one GPU per rank"
else:                                                           # fewer GPUs than ranks: share GPU 0, gloo
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    e, backend = Engine(0), "gloo, one GPU"
SR = 48000
n = int(seconds * SR) // 40 * 40                                # ten blocks, each a multiple of 2^level
x = e.synth_noisy_speech(n, SR, stream=77).to(dt)               # every rank can make the whole recording: cheap here
noise = e.synth_noisy_speech(5 * SR, SR, stream=78).to(dt)
prof = e.noise_psd(noise, [5 * SR])[0]
pipe = spread.SpreadPipeline(e, n, SR, rank, world)
s, t = pipe.ranges_in[rank]
torch.cuda.synchronize(); dist.barrier(); t0 = time.perf_counter()
y2 = pipe.wavelet_stage(x[s:t].clone())
y2b, r2b = pipe.spectral_subtract(y2, pipe.ranges_in, prof)
y2c, r2c, gain = pipe.wiener(y2b, r2b)
torch.cuda.synchronize(); dist.barrier(); spread_ms = (time.perf_counter() - t0) * 1e3
# the one-call results (every rank computes them: they are the checker here)
w2 = e.denoise_signal(x, SR)
w2b, l2b = e.spectral_subtract(w2, [w2.numel()], prof)
w2c, l2c, g1 = e.wiener(w2b, [int(l2b[0])], SR, want_gain=True)
a2, b2 = r2b[rank]
a3, b3 = r2c[rank]
res = {"rank": rank, "range": [s, t],
       "wavelet_equal": bool(torch.equal(y2, w2[s:t])), "wavelet_max_abs_diff": float((y2 - w2[s:t]).abs().max()),
       "subtract_equal": bool(torch.equal(y2b, w2b[a2:b2])),
       "wiener_max_abs_diff": float((y2c - w2c[a3:b3]).abs().max()) if b3 > a3 else 0.0,
       "gain_max_abs_diff": float((gain - g1[0]).abs().max())}
out = [None] * world
dist.all_gather_object(out, res)
if rank == 0:
    print(json.dumps({"workload": "one %.0f s 48 kHz mono recording, %s, spread over %d ranks (%s)" % (seconds, str(dt), world, backend),
                      "samples": n, "spread_ms": spread_ms, "ranks": out}))
dist.destroy_process_group()

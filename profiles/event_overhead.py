import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "audio-denoising_amd"))
import bench
from adn_hip import get_engine
e = get_engine(0)
wl = bench.Workload(e, 64, torch.float32, 0)
def timed(n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): wl.step()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
wl.step(); wl.step()
for r in range(3):
    e.kernel_timing(False); a = timed()
    e.kernel_timing(True); b = timed(); e.kernel_times(); e.kernel_timing(False)
    ev = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): wl.step(ev)
    torch.cuda.synchronize(); c = 1e3 * (time.perf_counter() - t0) / 10
    print("no events %.3f  library events %.3f  stage events only %.3f" % (a, b, c), flush=True)

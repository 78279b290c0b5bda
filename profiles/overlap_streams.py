"""Experiment: the bench step as ONE batch on one stream against the same clips as TWO half batches, each with
its own library context and HIP stream, so that the memory-bound kernels of one half can share the chip with the
arithmetic-bound kernels of the other.  Prints ms per step (whole batch) of each schedule."""
import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "audio-denoising_amd"))
import bench
from adn_hip import get_engine
from adn_hip.engine import Engine

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = 10
e0 = get_engine(0)


def timed(fn):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


wl = bench.Workload(e0, clips, torch.float32, 0)
print("one batch of %d clips, one stream: %.3f ms/step" % (clips, timed(wl.step)), flush=True)
del wl
torch.cuda.empty_cache()

for parts in (2, 4):
    engs = [e0] + [Engine(0) for _ in range(parts - 1)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    wls = [bench.Workload(engs[i], clips // parts, torch.float32, i) for i in range(parts)]

    def chain_by_chain():
        for i in range(parts):
            with torch.cuda.stream(streams[i]):
                wls[i].step()

    def stage_by_stage():
        for name in bench.Workload.STAGES:
            for i in range(parts):
                with torch.cuda.stream(streams[i]):
                    getattr(wls[i], name)()

    def skewed(after):
        """Part i+1 starts its chain when part i has finished stage `after`."""
        def run():
            evs = []
            for i in range(parts):
                with torch.cuda.stream(streams[i]):
                    if i > 0:
                        streams[i].wait_event(evs[i - 1])
                    for name in bench.Workload.STAGES:
                        getattr(wls[i], name)()
                        if name == after:
                            ev = torch.cuda.Event()
                            ev.record(streams[i])
                            evs.append(ev)
        return run

    for after in ("phase2", "phase2b"):
        print("%d parts, next part starts after %-8s: %.3f ms/step" % (parts, after, timed(skewed(after))), flush=True)
    print("%d parts, chain after chain : %.3f ms/step" % (parts, timed(chain_by_chain)), flush=True)
    print("%d parts, stage by stage    : %.3f ms/step" % (parts, timed(stage_by_stage)), flush=True)
    del wls, engs
    torch.cuda.empty_cache()

"""Developer probe: step time over a long run in blocks of 50 steps (does the host keep ahead of the GPU?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Runner, PERIOD_S
from thesis_amd.datasets import synthetic
P, B, T = 1024, 1081, 420
log = synthetic.make_log(T + 2, B, period=PERIOD_S)
r = Runner(P, B, 0.05, log)
if len(sys.argv) > 1:
    r.e.set_profiling(True)
for _ in range(10):
    r.step()
r.e.synchronize()
for blk in range(8):
    t0 = time.perf_counter()
    th = 0.0
    for _ in range(50):
        h0 = time.perf_counter(); r.step(); th += time.perf_counter() - h0
    r.e.synchronize()
    dt = time.perf_counter() - t0
    c = r.e.counters()
    print(f"block {blk}: {dt / 50 * 1e3:.3f} ms/step wall, host enqueue {th / 50 * 1e3:.3f} ms/step, fallbacks {c['window_fallbacks']}, tiles {c['tiles_in_use']}", flush=True)

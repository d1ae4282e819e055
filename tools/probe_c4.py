"""Developer probe: a long run at the per-GPU size of BASELINE configs[3] (2048 particles, 1081 beams): step time,
tracking error of particle 0 against the simulated truth, pool use, fallbacks and NDT statistics per block of steps."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Runner, PERIOD_S
from thesis_amd.datasets import synthetic
P = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
T = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
BLK = 500
log = synthetic.make_log(T + 2, 1081, period=PERIOD_S)
truth = log[3]
r = Runner(P, 1081, 0.05, log, ndt=int(os.environ.get("NDT", "1")))
prev = r.e.counters()
for blk in range(T // BLK):
    t0 = time.perf_counter()
    for _ in range(BLK):
        r.step()
    r.e.synchronize()
    dt = time.perf_counter() - t0
    c = r.e.counters()
    pose = r.e.poses()
    w = r.e.weights()
    k = r.frame
    err = np.abs(pose - truth[min(k, len(truth) - 1)])
    err[:, 2] = np.abs(np.angle(np.exp(1j * err[:, 2])))
    print(f"steps {k:5d}: {dt / BLK * 1e3:.3f} ms/step, median |err| xy {np.median(err[:, 0]):.3f} {np.median(err[:, 1]):.3f} m, "
          f"theta {np.median(err[:, 2]):.4f} rad, tiles {c['tiles_in_use']}, fallbacks {c['window_fallbacks'] - prev['window_fallbacks']}, "
          f"ndt evals/run {(c['ndt_evaluations'] - prev['ndt_evaluations']) / max(1, c['ndt_runs'] - prev['ndt_runs']):.2f}, "
          f"accepted {(c['ndt_accepted'] - prev['ndt_accepted']) / max(1, c['ndt_runs'] - prev['ndt_runs']):.2f}, "
          f"weights finite {bool(np.all(np.isfinite(w)))}", flush=True)
    prev = c

"""Quick timing probe of the hot kernels (developer tool, GPU box)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thesis_amd.engine import ParticleEngine
from thesis_amd.datasets import synthetic

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B = 1081
ang, ranges, odo, poses = synthetic.make_log(12, B)
e = ParticleEngine(P, max_beams=B)
e.set_profiling(True)
rng = np.random.Generator(np.random.PCG64(0))
for step in range(8):
    e.set_scan(ranges[step], ang)
    pp = poses[step] + rng.normal(0, 0.01, size=(P, 3))
    t0 = time.perf_counter()
    e.map_update(pp)
    t1 = time.perf_counter()
    c = e.counters()
    g = pp[:, None, :] + rng.normal(0, 0.01, size=(P, 30, 3))
    prs = np.ones((P, 30))
    t2 = time.perf_counter()
    w = e.weight_samples(g, prs)
    t3 = time.perf_counter()
    c2 = e.counters()
    print(f"step {step}: raycast kernel {c['ms_raycast']:.3f} ms (call {1e3*(t1-t0):.2f} ms) cells {c['ray_cells_visited']/P:.0f}/particle "
          f"written {c['cells_written']/P:.0f} slow {c['slow_cells']} | weight kernel {c2['ms_weight']:.3f} ms (call {1e3*(t3-t2):.1f} ms)", flush=True)

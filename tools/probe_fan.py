"""Developer probe: map update only, whole-fan kernel vs the 128x128-window kernel (timing + fallback counters)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thesis_amd import engine
from thesis_amd.datasets import synthetic

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B = int(os.environ.get("PROBE_B", "1081"))
ang = synthetic.beam_angles(B)
a, ranges, odo, poses = synthetic.make_log(12, B, period=0.7)
CS = float(os.environ.get("PROBE_CS", "0.05"))
for mode in os.environ.get("PROBE_KERNELS", "auto,ray,window").split(","):
    if mode != "auto":
        os.environ["RBPF_MAP_KERNEL"] = mode
    else:
        os.environ.pop("RBPF_MAP_KERNEL", None)
    e = engine.ParticleEngine(P, max_beams=B, pool_tiles=2 * P, cell_size=CS)
    e.set_profiling(True)
    rng = np.random.Generator(np.random.PCG64(5))
    for k in range(12):
        e.set_scan(ranges[k], ang)
        pp = poses[k] + rng.normal(0, 0.02, size=(P, 3))
        e.map_update(pp)
    e.synchronize()
    c = e.counters()
    ms = e.kernel_ms("raycast")
    print(mode, "raycast ms", np.round(ms, 3).tolist(), "fallbacks", c["window_fallbacks"], "reasons %x" % c["fallback_reasons"], "slow", c["slow_cells"], "cells", c["ray_cells_visited"], "written", c["cells_written"], "windows", c["map_windows"], "events/pu", round(c["map_events"] / (12 * P), 1), "ev overflows", c["map_event_overflows"], "fb geom/bound/tables", c["fallback_geometry"], c["fallback_bound"], c["fallback_tables"], flush=True)
    st = np.array(list(c["stamps"]), dtype=np.float64)
    if st.sum() > 0:
        print("  stamps kcycles/particle-update:", np.round(st / (12 * P) / 1e3, 1).tolist(), "total", round(st.sum() / (12 * P) / 1e3, 1), flush=True)
    e.close()

"""Developer probe for rocprofv3 --pmc: a few map updates with the whole-fan kernel only."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thesis_amd import engine
from thesis_amd.datasets import synthetic
P, B = 1024, 1081
ang = synthetic.beam_angles(B)
a, ranges, odo, poses = synthetic.make_log(6, B, period=0.7)
e = engine.ParticleEngine(P, max_beams=B, pool_tiles=2 * P)
rng = np.random.Generator(np.random.PCG64(5))
for k in range(6):
    e.set_scan(ranges[k], ang)
    e.map_update(poses[k] + rng.normal(0, 0.005, size=(P, 3)))
e.synchronize()
print(e.counters()["window_fallbacks"])
e.close()

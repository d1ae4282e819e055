"""Diagnostic: per-phase cycle shares of propose_weight_kernel (needs RBPF_STAMPS=propose python -m thesis_amd.build --force)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Runner, PERIOD_S
from thesis_amd.datasets import synthetic
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
log = synthetic.make_log(40, 1081, period=PERIOD_S)
r = Runner(P, 1081, 0.05, log, ndt=0)
for _ in range(5):
    r.step()
r.e.set_profiling(True)
n = 20
for _ in range(n):
    r.step()
c = r.e.counters()
st = np.array(list(c["stamps"]), dtype=np.float64)
names = ["frame (tab, samples)", "beams to LDS", "look-ups", "float64 queue", "moments"]
print("weight ms", r.e.kernel_ms("weight").mean())
for nm, v in zip(names, st):
    print(f"  {nm:24s} {v / st[:5].sum() * 100:6.2f} %  {v / (n * P):10.0f} cycles/particle")

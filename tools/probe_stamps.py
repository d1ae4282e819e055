"""Diagnostic: per-phase cycle shares of the map-update kernel (needs a -DRBPF_STAMPS build)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from bench import Runner, PERIOD_S
from thesis_amd.datasets import synthetic
P = int(sys.argv[1]) if len(sys.argv) > 1 else 10240
log = synthetic.make_log(16, 1081, period=PERIOD_S)
r = Runner(P, 1081, 0.05, log)
for _ in range(3):
    r.step()
r.e.set_profiling(True)
for _ in range(8):
    r.step()
c = r.e.counters()
st = np.array(list(c["stamps"]), dtype=np.float64)
names = ["setup", "P0 clear+lut", "P1b rank+oldv", "P2 walk", "P3 replay", "P4 rmw", "end barrier", "P1a flag+clip"]
print("raycast ms mean", r.e.kernel_ms("raycast").mean())
for n, v in zip(names, st):
    print(f"{n:14s} {v/st.sum()*100:6.2f} %   {v/(8*P):12.0f} cycles/particle")
print("total cycles/particle", st.sum() / (8 * P))

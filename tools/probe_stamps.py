"""Diagnostic: per-phase cycles of a map-update kernel under the bench's step (needs a stamped build:
RBPF_STAMPS=mapray|mapfan python -m thesis_amd.build --force; RBPF_MAP_KERNEL picks the kernel).
usage: probe_stamps.py [particles] [warm steps] [measured steps]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Runner, PERIOD_S
from thesis_amd.datasets import synthetic
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
W = int(sys.argv[2]) if len(sys.argv) > 2 else 3
K = int(sys.argv[3]) if len(sys.argv) > 3 else 8
log = synthetic.make_log(W + K + 4, 1081, period=PERIOD_S)
r = Runner(P, 1081, 0.05, log)
for _ in range(W):
    r.step()
r.e.set_profiling(True)
for _ in range(K):
    r.step()
c = r.e.counters()
st = np.array(list(c["stamps"]), dtype=np.float64)
print("raycast ms mean", r.e.kernel_ms("raycast").mean(), "slow cells/pu", c["slow_cells"] / (K * P), "fallbacks", c["window_fallbacks"], "reasons %x" % c["fallback_reasons"])
print("kcycles per particle-update:", np.round(st / (K * P) / 1e3, 1).tolist(), "total", round(st.sum() / (K * P) / 1e3, 1))

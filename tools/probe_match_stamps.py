"""Diagnostic: per-phase cycle shares of the scan-match kernel (needs a -DRBPF_STAMPS build with the map-update
stamps disabled, see RBPF_STAMPS=match)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Runner, PERIOD_S
from thesis_amd.datasets import synthetic
P = int(sys.argv[1]) if len(sys.argv) > 1 else 10240
log = synthetic.make_log(16, 1081, period=PERIOD_S)
r = Runner(P, 1081, 0.05, log, ndt=0)          # with the NDT stage on, its kernel owns the stamp slots (tools/probe_ndt_stamps.py)
for _ in range(5):
    r.step()
names = ["setup+zero", "field", "dilate+pool", "coarse", "fine", "score+cov", "colmap", "spare"]
for label in ("adj=0 (own map)", "adj=1 (last scan)"):
    r.e.set_profiling(True)
    n = 0
    while n < 2:
        adj = not (r.frame % 5 < 2)
        if adj == (label.startswith("adj=1")):
            n += 1
        else:
            r.e.set_profiling(True)
            n = 0
        r.step()
        if n == 1 and adj != (label.startswith("adj=1")):
            n = 0
    c = r.e.counters(); st = np.array(list(c["stamps"]), dtype=np.float64)
    ms = r.e.kernel_ms("match")
    print(label, "match ms", ms[-2:], "frames", r.frame)
    for nm, v in zip(names, st):
        print(f"  {nm:12s} {v/st.sum()*100:6.2f} %  {v/(len(ms)*P):10.0f} cycles/particle")

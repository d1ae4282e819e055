"""Diagnostic: per-phase cycle shares of the NDT kernel (needs a -DRBPF_STAMPS build of kernels_match.hip:
RBPF_STAMPS=match python -c 'import __graft_entry__ as g; g.build()')."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Runner, PERIOD_S
from thesis_amd.datasets import synthetic
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
log = synthetic.make_log(40, 1081, period=PERIOD_S)
r = Runner(P, 1081, 0.05, log, ndt=1)
for _ in range(5):
    r.step()
r.e.set_profiling(True)
for _ in range(20):
    r.step()
c = r.e.counters()
st = np.array(list(c["stamps"]), dtype=np.float64)
names = ["staging", "beams", "reduce+barrier", "optimiser step+barrier"]
print("evals/run", c["ndt_evaluations"] / c["ndt_runs"], "match+ndt ms", r.e.kernel_ms("match").mean())
for nm, v in zip(names, st):
    print(f"  {nm:24s} {v / st[:4].sum() * 100:6.2f} %  {v / c['ndt_runs']:10.0f} cycles/run")

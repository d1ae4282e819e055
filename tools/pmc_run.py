"""Short run of the 10 240-particle workload for rocprofv3 --pmc passes (developer tool)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Runner, PERIOD_S
from thesis_amd.datasets import synthetic
P = int(sys.argv[1]) if len(sys.argv) > 1 else 10240
log = synthetic.make_log(10, 1081, period=PERIOD_S)
r = Runner(P, 1081, 0.05, log)
for _ in range(6):
    r.step()
r.e.synchronize()

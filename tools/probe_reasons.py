import os, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import Runner, PERIOD_S
from thesis_amd.datasets import synthetic
P = 1024
log = synthetic.make_log(110, 1081, period=PERIOD_S)
r = Runner(P, 1081, 0.05, log)
prev = None
for k in range(105):
    r.step()
    if k % 5 == 4:
        c = r.e.counters()
        cur = (c["window_fallbacks"], c["fallback_reasons"], c["slow_cells"])
        if prev is None or cur != prev:
            fr = c["fallback_reasons"]
            print(k, "fallbacks", c["window_fallbacks"], "reasons", [(fr >> (16 * i)) & 0xFFFF for i in range(4)], "slow", c["slow_cells"], "pose", np.round(log[3][k + 1], 2), flush=True)
        prev = cur

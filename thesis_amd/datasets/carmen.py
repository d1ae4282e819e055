"""CARMEN log reader (ODOM / FLASER records) for the datasets the reference replays.

Mirrors what the reference's adapters extract (IntelLidarData.py:12-20, IntelIMUData.py:9-20): per FLASER
record the first B ranges and a timestamp, per ODOM record (x, y, theta) and a timestamp, both quantised to
the reference's 1e-4 s ticks (`int(10*t)*10`).  Angles: `-pi/2 + i*pi/(B-1)` (IntelLidarData.py:19).
Unlike the reference's per-dataset adapters the beam count is read from the record itself, so the 181-beam
Orebro log and the 361-beam CSAIL log load too (the reference's OberoLidarData expects 360 beams and fails).
"""
from __future__ import annotations

from dataclasses import dataclass
from math import pi
from typing import List

import numpy as np


@dataclass
class CarmenLog:
    angles: np.ndarray        # [B]
    scans: np.ndarray         # [n_scans, B]
    scan_times: np.ndarray    # [n_scans] ticks of 1e-4 s
    odom: np.ndarray          # [n_odom, 3] x, y, theta
    odom_times: np.ndarray    # [n_odom]
    order: np.ndarray         # [n_scans + n_odom] records in file order: >= 0 scan index, < 0 -(odom index) - 1


def load_carmen(path: str, time_scale: int = 10) -> CarmenLog:
    """`time_scale` = 10 reproduces the Intel adapters' `int(10*t)*10`; the Freiburg-style adapters use 1000."""
    scans: List[List[float]] = []
    st: List[int] = []
    odo: List[List[float]] = []
    ot: List[int] = []
    order: List[int] = []
    n_beams = None
    with open(path) as f:
        for line in f:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "FLASER":
                b = int(tok[1])
                if n_beams is None:
                    n_beams = b
                if b != n_beams:
                    continue
                order.append(len(scans))
                scans.append([float(x) for x in tok[2:2 + b]])
                st.append(int(time_scale * float(tok[-3])) * 10)                        # IntelLidarData.py:17
            elif tok[0] == "ODOM":
                order.append(-len(odo) - 1)
                odo.append([float(tok[1]), float(tok[2]), float(tok[3])])
                ot.append(int(time_scale * float(tok[7])) * 10)                         # IntelIMUData.py:18
    if n_beams is None:
        raise ValueError(f"{path}: no FLASER records")
    angles = np.array([-pi / 2 + i * pi / (n_beams - 1) for i in range(n_beams)])
    return CarmenLog(angles, np.array(scans), np.array(st, dtype=np.int64), np.array(odo), np.array(ot, dtype=np.int64),
                     np.array(order, dtype=np.int64))

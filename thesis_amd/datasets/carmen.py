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


def _open_text(path: str):
    """A log as text, gzip-compressed or plain."""
    if path.endswith(".gz"):
        import gzip
        return gzip.open(path, "rt")
    return open(path)


def load_carmen(path: str, time_scale: int = 10) -> CarmenLog:
    """`time_scale` = 10 reproduces the Intel adapters' `int(10*t)*10`; the Freiburg-style adapters use 1000."""
    scans: List[List[float]] = []
    st: List[int] = []
    odo: List[List[float]] = []
    ot: List[int] = []
    order: List[int] = []
    n_beams = None
    with _open_text(path) as f:
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


@dataclass
class VelocityLog:
    """What the Freid101-family adapters hand to main.py (Freid101IMUData.py:9-32, Freid101LidarData.py:12-21)."""
    angles: np.ndarray        # [B]
    scans: np.ndarray         # [n_scans, B]   one scan per unique timestamp (the first record with it)
    scan_times: np.ndarray    # [n_scans] ticks of 1e-4 s, sorted, unique
    odom: np.ndarray          # [n_readings, 3]  (vx, vy, omega) in the global frame; row 0 is (0, 0, 0)
    odom_times: np.ndarray    # [n_readings] sorted, unique


def load_carmen_velocity(path: str, calib_n: int = 5, negate_x: bool = False, relative_time: bool = False) -> VelocityLog:
    """The Freid101 family of adapters (Freid101 / Obero / Aces / Bele / Freid / FreidCorrect / IntelRaw *IMUData.py and
    *LidarData.py): odometry as global-frame velocities from the differences of calibrated poses over unique timestamps.

      times      int(1000 * t) * 10 with t = the record's last field (logger time), Freid101IMUData.py:18;
                 `relative_time`: t minus the first ODOM record's (BeleIMUData.py:18)
      unique     np.unique(times, return_index=True): sorted times, the FIRST record of each (:19)
      calibrate  poses minus the mean of the first `calib_n` poses (:22-23; 5, AcesIMUData.py:22 uses 1)
      velocities 1e4 * diff(calibrated poses at the unique records) / diff(times), per axis; a (0, 0, 0) row in
                 front (:25,31)
      negate_x   FreidIMUData.py:15 / FreidCorrectIMUData.py:15
    Scans: the first `B` fields after the count, time from the last field, unique the same way
    (Freid101LidarData.py:15-18); `B` is read from the record (the reference's OberoLidarData.py:8 says 360 and cannot
    parse the 181-beam data/orebro.log); angles -pi/2 + i*pi/(B-1) (:20)."""
    odo: List[List[float]] = []
    scans: List[List[float]] = []
    st: List[float] = []
    n_beams = None
    with open(path) as f:
        for line in f:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "ODOM":
                odo.append([float(tok[1]), float(tok[2]), float(tok[3]), float(tok[9])])
            elif tok[0] == "FLASER":
                b = int(tok[1])
                if n_beams is None:
                    n_beams = b
                if b != n_beams:
                    continue
                scans.append([float(x) for x in tok[2:2 + b]])
                st.append(float(tok[-1]))
    if n_beams is None or not odo:
        raise ValueError(f"{path}: needs ODOM and FLASER records")
    r = np.array(odo)
    x = -r[:, 0] if negate_x else r[:, 0]
    t0 = r[0, 3] if relative_time else 0.0
    times = np.array([int(1000 * (t - t0)) * 10 for t in r[:, 3]])
    sorted_times, idxs = np.unique(times, return_index=True)
    imu = np.stack([x, r[:, 1], r[:, 2]], axis=1)
    pos = imu - np.mean(imu[0:calib_n], axis=0)
    real_times = np.column_stack((sorted_times, sorted_times, sorted_times))
    vel = 1e4 * np.diff(pos[idxs], axis=0) / np.diff(real_times, axis=0)
    stimes = np.array([int(1000 * t) * 10 for t in st])
    s_sorted, s_idx = np.unique(stimes, return_index=True)
    angles = np.array([-pi / 2 + i * pi / (n_beams - 1) for i in range(n_beams)])
    return VelocityLog(angles, np.array(scans)[s_idx], s_sorted.astype(np.int64), np.vstack(([0.0, 0.0, 0.0], vel)),
                       sorted_times.astype(np.int64))

"""Reader for the reference's default dataset: `data/lidar.mat`, `data/imu.mat`, `data/speed.mat`.

Mirrors what DefaultLidarData.py:11-21 and DefaultIMUData.py:9-25 hand to main.py: 361-beam scans in metres
(`0.01 * (raw & 0x1FFF)`, the upper bits of the SICK words are intensity flags), scan and IMU times in ticks of 1e-4 s
(`t * 1e4`), and IMU rows (speed, omega) with the mean of the first 1000 samples removed from each channel.
The files are MATLAB v5 containers read with `scipy.io.loadmat` (plain data, nothing is executed).
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from math import pi

import numpy as np

POINTS_PER_SCAN = 361          # DefaultLidarData.py:6
NUM_REF_POINTS = 1000          # DefaultIMUData.py:7


@dataclass
class DefaultLog:
    angles: np.ndarray        # [361]  -pi/2 + i*pi/360
    scans: np.ndarray         # [n_scans, 361] metres
    scan_times: np.ndarray    # [n_scans] ticks of 1e-4 s (float64, as the reference keeps them)
    imu: np.ndarray           # [n_imu, 2] speed (m/s), omega (rad/s), baselines removed
    imu_times: np.ndarray     # [n_imu]


def load_default_mat(data_dir: str) -> DefaultLog:
    from scipy.io import loadmat
    lidar = loadmat(os.path.join(data_dir, "lidar"))
    raw = np.asarray(lidar["dataL"]["Scans"][0][0])                     # [361, n] uint16
    scans = (0.01 * (raw.astype(np.int64) & 0x1FFF)).T                     # DefaultLidarData.py:13-15
    scan_times = np.asarray(lidar["dataL"]["times"][0][0][0], dtype=np.float64) * 1e4
    angles = np.array([-pi / 2 + i * pi / 360 for i in range(POINTS_PER_SCAN)])
    imu = loadmat(os.path.join(data_dir, "imu"))
    enc = loadmat(os.path.join(data_dir, "speed"))
    raw_omega = np.asarray(imu["IMU"]["DATAf"][0][0][5])                # stored as single precision
    raw_speed = np.asarray(enc["Vel"]["speeds"][0][0][0])
    # DefaultIMUData.py:14-20: Python's sum() over the first 1000 samples and the subtraction both stay in the
    # stored precision; the promotion to float64 happens in np.vstack
    base_omega = sum(raw_omega[0:NUM_REF_POINTS]) / NUM_REF_POINTS
    base_speed = sum(raw_speed[0:NUM_REF_POINTS]) / NUM_REF_POINTS
    data = np.vstack((raw_speed - base_speed, raw_omega - base_omega)).T.astype(np.float64)
    imu_times = np.asarray(imu["IMU"]["times"][0][0][0], dtype=np.float64) * 1e4
    return DefaultLog(angles, scans, scan_times, data, imu_times)

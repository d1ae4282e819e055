"""Synthetic `room16` lidar/odometry workload.

World: axis-aligned square room, half-width 8 m centred at the origin, plus four
1 m square pillars centred at (+-4, +-4).  Beams: B = 1081 over 270 degrees
(``angle_i = -3pi/4 + i * (3pi/2)/(B-1)``), max range 30 m, range noise
N(0, 0.01^2) from PCG64(seed).  Trajectory: circle of radius 3 m at 0.5 m/s, one
scan every 0.1 s, odometry as global-frame velocities (vx, vy, omega) in the
style of the reference's Freid101 adapter (Freid101IMUData.py:34-41).

Angles follow the reference's convention: sensor frame, x forward
(lidar.py:78-79).
"""
from __future__ import annotations

from math import pi
from typing import Tuple

import numpy as np

ROOM_HALF = 8.0
PILLARS = [(4.0, 4.0), (-4.0, 4.0), (-4.0, -4.0), (4.0, -4.0)]
PILLAR_HALF = 0.5
MAX_RANGE = 30.0


def beam_angles(n_beams: int = 1081, fov: float = 1.5 * pi) -> np.ndarray:
    if n_beams == 1:
        return np.zeros(1)
    return -fov / 2 + np.arange(n_beams, dtype=np.float64) * (fov / (n_beams - 1))


def _ray_box_exit(ox, oy, dx, dy, half):
    """Distance along (dx,dy) from an interior point to the walls of [-half,half]^2."""
    with np.errstate(divide="ignore", invalid="ignore"):
        tx = np.where(dx > 0, (half - ox) / dx, np.where(dx < 0, (-half - ox) / dx, np.inf))
        ty = np.where(dy > 0, (half - oy) / dy, np.where(dy < 0, (-half - oy) / dy, np.inf))
    return np.minimum(tx, ty)


def _ray_box_enter(ox, oy, dx, dy, cx, cy, half):
    """Slab test: entry distance into the box centred (cx,cy), inf if missed."""
    with np.errstate(divide="ignore", invalid="ignore"):
        inv_x = np.where(dx != 0, 1.0 / dx, np.inf)
        inv_y = np.where(dy != 0, 1.0 / dy, np.inf)
    t1x, t2x = (cx - half - ox) * inv_x, (cx + half - ox) * inv_x
    t1y, t2y = (cy - half - oy) * inv_y, (cy + half - oy) * inv_y
    tmin = np.maximum(np.minimum(t1x, t2x), np.minimum(t1y, t2y))
    tmax = np.minimum(np.maximum(t1x, t2x), np.maximum(t1y, t2y))
    hit = (tmax >= np.maximum(tmin, 0.0)) & (tmin > 0.0)
    return np.where(hit, tmin, np.inf)


def cast_scan(pose, angles: np.ndarray, rng: np.random.Generator | None = None,
              noise_sigma: float = 0.01) -> np.ndarray:
    """Ranges [B] seen from ``pose = (x, y, theta)`` inside room16."""
    ox, oy, th = float(pose[0]), float(pose[1]), float(pose[2])
    a = angles + th
    dx, dy = np.cos(a), np.sin(a)
    t = _ray_box_exit(ox, oy, dx, dy, ROOM_HALF)
    for (cx, cy) in PILLARS:
        t = np.minimum(t, _ray_box_enter(ox, oy, dx, dy, cx, cy, PILLAR_HALF))
    t = np.minimum(t, MAX_RANGE)
    if rng is not None and noise_sigma > 0:
        t = t + rng.normal(0.0, noise_sigma, size=t.shape)
    return np.maximum(t, 0.0)


def circle_trajectory(n_steps: int, radius: float = 3.0, speed: float = 0.5,
                      period: float = 0.1) -> np.ndarray:
    """True poses [n_steps+1, 3] on a circle through the origin, heading tangent.

    The robot starts at (0, 0, 0) and circles the centre (0, radius)."""
    w = speed / radius
    k = np.arange(n_steps + 1, dtype=np.float64)
    th = w * period * k
    x = radius * np.sin(th)
    y = radius * (1.0 - np.cos(th))
    return np.stack([x, y, th], axis=1)


def make_log(n_steps: int, n_beams: int = 1081, seed: int = 1234, odo_seed: int = 1235,
             period: float = 0.1, fov: float = 1.5 * pi) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """(angles[B], ranges[n_steps+1, B], odo[n_steps, 3] = (vx, vy, omega), true_poses).

    ``ranges[k]`` is the scan taken at ``true_poses[k]``; ``odo[k]`` moves k -> k+1 over
    ``period`` seconds, with 1 % multiplicative noise (seed ``odo_seed``)."""
    angles = beam_angles(n_beams, fov)
    poses = circle_trajectory(n_steps, period=period)
    rng = np.random.Generator(np.random.PCG64(seed))
    ranges = np.stack([cast_scan(p, angles, rng) for p in poses])
    vel = np.diff(poses, axis=0) / period
    orng = np.random.Generator(np.random.PCG64(odo_seed))
    vel = vel * (1.0 + 0.01 * orng.standard_normal(vel.shape))
    return angles, ranges, vel, poses

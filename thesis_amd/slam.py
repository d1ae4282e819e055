"""The reference's main loop (main.py:138-214) on top of the batched engine.

`ParticleFilter` is the particle list + the three per-event operations of the loop; `run_log` replays a log
with the reference's semantics: merged timestamps, motion gating on particle 0 (0.33 m / pi/9 / first two
updates, main.py:41-43,152-155), scan-match cadence (`frame % 5 < 2`, main.py:156-159), `last_scan` refresh every
fifth frame (main.py:167-168), resample after every accepted scan (main.py:160).  It cold-starts (the committed
reference restores `pickle/1550.state`, which is not in its tree).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from math import pi, sqrt
from typing import List, Optional

import numpy as np

from .engine import ParticleEngine

MAX_UPDATE_COUNT = 2          # main.py:41
ROT_THRESHOLD = pi / 9        # main.py:42
DIST_THRESHOLD = 0.33         # main.py:43


class HybridMapView:
    """Read access to one particle's tiled map with the reference's method names (hybridmap.py:63-327)."""

    def __init__(self, pf: "ParticleFilter", index: int):
        self._pf, self._i = pf, index
        self._cell_size = pf.engine.cfg.cell_size

    def get_odds_at(self, pos):                       # hybridmap.py:85-93
        vals, none = self._pf.engine.get_odds_at(self._i, [[pos[0], pos[1]] if not hasattr(pos, "x") else [pos.x, pos.y]])
        return None if none[0] else float(vals[0])

    def get_pr_at(self, pos):                         # hybridmap.py:74-83
        o = self.get_odds_at(pos)
        if o is None:
            return None
        e = np.exp(o)
        return float(e / (1 + e))

    def get_occupied_points(self):                    # hybridmap.py:303-313 (cell units, as plotted by main.py:170-171)
        xs, ys = [], []
        q, thr = self._pf.engine.cfg.quantum, self._pf.engine.cfg.occupied_threshold
        for (cx, cy), cells in self._pf.engine.tiles(self._i):
            i, j = np.nonzero(cells.astype(np.float64) * q > thr)
            dim = cells.shape[0]
            xs.append(((i - dim / 2) * self._cell_size + cx) / self._cell_size)
            ys.append(((j - dim / 2) * self._cell_size + cy) / self._cell_size)
        return (np.concatenate(xs) if xs else np.empty(0)), (np.concatenate(ys) if ys else np.empty(0))

    def is_occ_at(self, x, y) -> bool:                # gridmap.py:255-260 through the tile that holds (x, y)
        o = self.get_odds_at((x, y))
        return o is not None and o > self._pf.engine.cfg.occupied_threshold

    def get_scan_match(self, guess, pose_range):      # hybridmap.py:210-261 for the scan last given to the filter
        """(pose[3], cov[3][3], score) of the engine seam for this particle's map: the point lists of
        hybridmap.py:213-238, the matcher call with guess 0 and rotation range pi/6 (:244-251), the result offset by
        the guess (:253-255).  NaN covariance and score 0 mean no valid match (matchScanCustom.m:25-28)."""
        from .engine import match_scan
        g = np.asarray([guess.x(), guess.y(), guess.theta()] if hasattr(guess, "theta") else guess, dtype=np.float64)
        curr, ref = self._pf.engine.match_inputs(self._i, g)
        pose, cov, score = match_scan(self._pf.engine, curr, ref, [0.0, 0.0, 0.0], int(1.0 / self._cell_size),
                                      [pose_range[0], pose_range[1], np.pi / 6])
        return pose + g, cov, score

    def get_scan_adj(self, scan_xy, prev_scan_xy, guess, pose_range):   # hybridmap.py:147-191
        """The same seam on raw points: `scan_xy` = the current scan in the sensor frame, `prev_scan_xy` = the previous
        accepted scan in the global frame (main.py:167-168); both translated by -guess.xy, points farther than 11 m
        dropped (:169-171)."""
        from .engine import match_scan
        g = np.asarray([guess.x(), guess.y(), guess.theta()] if hasattr(guess, "theta") else guess, dtype=np.float64)
        c, s_ = np.cos(g[2]), np.sin(g[2])
        sc = np.asarray(scan_xy, dtype=np.float64).reshape(-1, 2)
        curr = np.stack([c * sc[:, 0] - s_ * sc[:, 1], s_ * sc[:, 0] + c * sc[:, 1]], axis=1)      # lidar.py:123 minus guess.xy
        ref = np.asarray(prev_scan_xy, dtype=np.float64).reshape(-1, 2) - g[:2]
        curr = curr[np.sqrt((curr ** 2).sum(axis=1)) < 11.0]
        ref = ref[np.sqrt((ref ** 2).sum(axis=1)) < 11.0]
        pose, cov, score = match_scan(self._pf.engine, curr, ref, [0.0, 0.0, 0.0], int(1.0 / self._cell_size),
                                      [pose_range[0], pose_range[1], np.pi / 6])
        return pose + g, cov, score

    def __str__(self):
        return "Hybrid Map: %d maps" % len(self._pf.engine.tiles(self._i))


class Robot:
    """View of one particle with the reference's read API (robot.py:30-43)."""

    def __init__(self, pf: "ParticleFilter", index: int):
        self._pf, self._i = pf, index
        self._map = HybridMapView(pf, index)

    def get_latest_pose(self):
        return tuple(self._pf.engine.poses()[self._i])

    def weight(self):
        return [float(self._pf.engine.weights()[self._i])]

    def x(self):
        return [p[0] for p in self._pf.trajectory(self._i)]

    def y(self):
        return [p[1] for p in self._pf.trajectory(self._i)]

    def theta(self):
        return [p[2] for p in self._pf.trajectory(self._i)]

    def __str__(self):
        return "Robot at position: Pose: (%s, %s, %s)" % self.get_latest_pose()


class ParticleFilter:
    """NUM_PARTICLES robots (main.py:44,87) as one batched engine."""

    def __init__(self, n_particles: int, angles, motion_model: str = "velocity", *, cell_size: float = 0.05,
                 keep_history: bool = True, seed: int = 42, **engine_options):
        self.angles = np.ascontiguousarray(angles, dtype=np.float64)
        self.engine = ParticleEngine(n_particles, max_beams=len(self.angles), cell_size=cell_size, seed=seed,
                                     pool_tiles=engine_options.pop("pool_tiles", 4 * n_particles + 16), **engine_options)
        self.motion_model = motion_model
        self.particles = [Robot(self, i) for i in range(n_particles)]
        self.keep_history = keep_history
        self._poses: List[np.ndarray] = [self.engine.poses()] if keep_history else []
        self._ancestors: List[Optional[np.ndarray]] = [None] if keep_history else []
        self._urng = np.random.Generator(np.random.PCG64(seed))

    def imu_update(self, data, dt_ticks: float):                                  # main.py:144
        self.engine.imu_update(self.motion_model, data, dt_ticks)
        self._record(None)

    def map_update(self, ranges, last_scan_xy=None, adj: bool = False):          # main.py:157,159
        self.engine.set_scan(ranges, self.angles)
        self.engine.scan_update(adj=adj, last_scan_xy=last_scan_xy if adj else None)
        self._record(None)

    def resample(self, u: Optional[float] = None) -> bool:                        # main.py:160
        did, idx = self.engine.resample(float(self._urng.random()) if u is None else u)
        if did and self.keep_history:
            self._poses.append(self.engine.poses()); self._ancestors.append(idx)
        return did

    def _record(self, anc):
        if self.keep_history:
            self._poses.append(self.engine.poses()); self._ancestors.append(anc)

    def trajectory(self, index: int) -> List[np.ndarray]:
        """Pose history of the particle that is now at `index`, following its ancestry back (robot.py:141-146 copies
        the history lists on every duplication; here they are reconstructed from the per-step ancestor indices)."""
        out, i = [], index
        for poses, anc in zip(reversed(self._poses), reversed(self._ancestors)):
            if anc is not None:
                i = int(anc[i])
                continue
            out.append(poses[i])
        out.reverse()
        return out

    def close(self):
        self.engine.close()


@dataclass
class RunResult:
    frames: int = 0
    accepted: int = 0
    resamples: int = 0
    pose0: List[np.ndarray] = field(default_factory=list)
    # what the loop decided, event by event: ("imu", reading index, dt) and
    # ("scan", scan index, frame, accepted, adj, last_scan refreshed); pose0_before[k] = particle 0 in front of scan event k
    trace: List[tuple] = field(default_factory=list)
    pose0_before: List[np.ndarray] = field(default_factory=list)


def scan_to_global(ranges, angles, pose):
    """Scan.from_global_reference (lidar.py:111-128) as an [B, 2] array."""
    x, y = ranges * np.cos(angles), ranges * np.sin(angles)
    c, s = np.cos(pose[2]), np.sin(pose[2])
    return np.stack([c * x - s * y + pose[0], s * x + c * y + pose[1]], axis=1)


def run_log(pf: ParticleFilter, scans, scan_times, odom, odom_times, max_frames: Optional[int] = None,
            order=None) -> RunResult:
    """main.py:138-214 for a log given as arrays; `odom` rows are the motion model's readings.

    Events are processed in chronological order.  The reference merges the two timestamp arrays with np.unique and
    advances each cursor only on an exact timestamp match (main.py:114,139-148), which stalls for good on a duplicated
    or non-monotonic timestamp (data/intel.txt has both); here every record is consumed exactly once, in file order
    when `order` (CarmenLog.order) is given, else in timestamp order with odometry before a scan of equal time."""
    res = RunResult()
    if order is None:
        keys = np.concatenate((np.stack([odom_times, np.zeros_like(odom_times)], 1), np.stack([scan_times, np.ones_like(scan_times)], 1)))
        ids = np.concatenate((-np.arange(len(odom_times)) - 1, np.arange(len(scan_times))))
        order = ids[np.lexsort((keys[:, 1], keys[:, 0]))]
    prev_ts = odom_times[0] if len(odom_times) else 0
    frame = 0
    last_updated_pose = np.array(pf.particles[0].get_latest_pose())
    last_scan = scan_to_global(scans[0], pf.angles, last_updated_pose)           # main.py:111
    update_count = 0
    for rec in order:
        if rec < 0:                                                              # main.py:139-145
            i = -int(rec) - 1
            dt = max(int(odom_times[i]) - int(prev_ts), 0)
            pf.imu_update(odom[i], float(dt))
            prev_ts = odom_times[i]
            res.trace.append(("imu", i, dt))
            continue
        ranges = scans[int(rec)]                                                 # main.py:147-181
        curr = np.array(pf.particles[0].get_latest_pose())
        dist = sqrt((last_updated_pose[0] - curr[0]) ** 2 + (last_updated_pose[1] - curr[1]) ** 2)
        rot = abs(last_updated_pose[2] - curr[2])
        res.pose0_before.append(curr)
        accepted = update_count < MAX_UPDATE_COUNT or dist >= DIST_THRESHOLD or rot >= ROT_THRESHOLD
        res.trace.append(("scan", int(rec), frame, bool(accepted), bool(accepted and not (frame % 5 < 2)), bool(accepted and frame % 5 == 0)))
        if accepted:
            pf.map_update(ranges, last_scan, adj=not (frame % 5 < 2))
            res.resamples += int(pf.resample())
            res.accepted += 1
            if dist >= DIST_THRESHOLD or rot >= ROT_THRESHOLD:
                update_count = 0
                last_updated_pose = curr
            elif update_count < MAX_UPDATE_COUNT:
                update_count += 1
            if frame % 5 == 0:                                                   # main.py:167-168
                last_scan = scan_to_global(ranges, pf.angles, np.array(pf.particles[0].get_latest_pose()))
        res.pose0.append(np.array(pf.particles[0].get_latest_pose()))
        frame += 1
        res.frames = frame
        if max_frames is not None and frame >= max_frames:
            break
    return res

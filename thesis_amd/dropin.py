"""Per-object facade: the reference's `Robot` / `HybridMap` / `resample` surface over the batched engine.

The lines of the reference's main loop that touch particles run unchanged over these objects:

    particles = [Robot(eng) for _ in range(NUM_PARTICLES)]                       # main.py:87
    [p.imu_update(imu_reading) for p in particles]                               # main.py:144
    curr_pose = particles[0].get_latest_pose()                                   # main.py:152
    weights = [p.map_update(lidar_reading, last_scan, adj) for p in particles]   # main.py:157,159
    particles = resample(particles)                                              # main.py:160
    last_scan = lidar_reading.from_global_reference(particles[0].get_latest_pose())   # main.py:167
    plot_x, plot_y = particles[0]._map.get_occupied_points()                     # main.py:170

How a per-particle call becomes one batched call: the engine advances all particles together, so the FIRST `Robot`
that receives a call of a list comprehension (`imu_update(reading)`, `map_update(scan, last_scan, adj)`) runs it for
the whole population; the calls of the other robots with the same argument objects find their share done.  A round
ends when a robot is called a second time or the arguments change.  Reads (`get_latest_pose`, `weight`, `_map...`)
always see a consistent population.  What the reference's objects take as duck-typed arguments stays duck-typed:
  reading  .get_data() -> 2 or 3 numbers, .dt() -> ticks of 1e-4 s            (models.py:44-77)
  scan     .x(), .y() -> sensor-frame end points                              (lidar.py:82-87)
  pose     .x(), .y(), .theta()                                               (models.py:20-42)
The motion-model family (the adapter callbacks of IMUData.py:9-40 cannot run on the GPU) and the grid are chosen once
with `configure(...)`; the first `Robot(...)` after it opens a new population.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np

from .slam import ParticleFilter, HybridMapView as _MapView

__all__ = ["configure", "Robot", "HybridMap", "resample", "Pose", "Position", "Scan"]


class Position:                                      # models.py:11-18
    def __init__(self, _x, _y):
        self.x, self.y = _x, _y

    def __str__(self):
        return "(" + str(self.x) + ", " + str(self.y) + ")"


class Pose:                                          # models.py:20-42
    def __init__(self, x, y, theta):
        self._x, self._y, self._theta = x, y, theta

    def x(self):
        return self._x

    def y(self):
        return self._y

    def theta(self):
        return self._theta

    def pos(self):
        return Position(self._x, self._y)

    def __iter__(self):
        return iter((self._x, self._y, self._theta))

    def __str__(self):
        return "Pose: (%s, %s, %s)" % (self._x, self._y, self._theta)


class Scan:                                          # lidar.py:70-128, for callers that have no reference Scan at hand
    def __init__(self, ranges, angles, timestamp=0):
        if ranges is not None:
            r, a = np.asarray(ranges, dtype=np.float64), np.asarray(angles, dtype=np.float64)
            self._x, self._y = r * np.cos(a), r * np.sin(a)
        self._timestamp = timestamp

    def x(self):
        return self._x

    def y(self):
        return self._y

    def __len__(self):
        return len(self._x)

    def from_global_reference(self, frame):
        c, s = np.cos(frame.theta()), np.sin(frame.theta())
        out = Scan(None, None, self._timestamp)
        out._x = (c * self._x + (-s) * self._y) + frame.x()
        out._y = (s * self._x + c * self._y) + frame.y()
        return out


class _Session:
    """One population: the Robots created since the last configure(), the engine behind them, the open round."""

    def __init__(self, motion_model, cell_size, n_beams, seed, options):
        self.motion_model, self.cell_size, self.n_beams, self.seed, self.options = motion_model, cell_size, n_beams, seed, options
        self.robots: List["Robot"] = []
        self.pf: Optional[ParticleFilter] = None
        self.round_key = None
        self.round_seen = set()
        self._cache = {}                                 # poses / weights / covs of the whole population, read back once per round

    def state(self, what: str) -> np.ndarray:
        """engine.poses() / weights() / covs(), read back once and kept until the next call that changes them: a list
        comprehension over P Robots costs one device read-back, not P."""
        if what not in self._cache:
            e = self.filter().engine
            self._cache[what] = {"poses": e.poses, "weights": e.weights, "covs": e.covs}[what]()
        return self._cache[what]

    def invalidate(self):
        self._cache = {}

    def filter(self) -> ParticleFilter:
        if self.pf is None:
            if not self.robots:
                raise RuntimeError("no Robot has been created")
            angles = np.zeros(self.n_beams)              # only the length matters: scans arrive as end points
            self.pf = ParticleFilter(len(self.robots), angles, motion_model=self.motion_model, cell_size=self.cell_size,
                                     seed=self.seed, **self.options)
        return self.pf

    def first_of_round(self, robot, key) -> bool:
        """True for the call that has to do the work of the comprehension."""
        if key != self.round_key or robot._i in self.round_seen:
            self.round_key, self.round_seen = key, {robot._i}
            return True
        self.round_seen.add(robot._i)
        return False

    def close(self):
        if self.pf is not None:
            self.pf.close()
            self.pf = None


_session: Optional[_Session] = None


def configure(motion_model: str = "velocity", cell_size: float = 0.05, max_beams: int = 1081, seed: int = 42, **engine_options):
    """Opens a new population.  motion_model: "unicycle" (DefaultIMUData.py:26-54), "absolute" (IntelIMUData.py:23-36),
    "velocity" (Freid101IMUData.py:34-55 and its family; `vel_noise=(...)` for IntelRaw's constants)."""
    global _session
    if _session is not None:
        _session.close()
    _session = _Session(motion_model, cell_size, max_beams, seed, engine_options)
    return _session


class HybridMap(_MapView):
    """hybridmap.py:63-327 for one particle (read access; the map changes through Robot.map_update)."""

    def __init__(self, session: _Session, index: int):
        self._s, self._i = session, index
        self._cell_size = session.cell_size

    @property
    def _size(self) -> float:                         # hybridmap.py:68 map_len_m, from the engine's configuration
        return float(self._s.filter().engine.cfg.tile_len_m)

    @property
    def _pf(self):                                    # the base class's methods go through the live filter
        return self._s.filter()

    def get_cell(self, x: float, y: float) -> Optional[Position]:
        """GridMap.get_cell (gridmap.py:119-128) in the tile that holds (x, y) (hybridmap.py:85-93): the 'get' index
        formula, float64 operations in the reference's order."""
        size = self._size
        dim = self._pf.engine.dim
        half = size / 2
        lx, ly = int(np.floor((x + half) / size)), int(np.floor((y + half) / size))
        while x < lx * size - half: lx -= 1
        while x >= lx * size + half: lx += 1
        while y < ly * size - half: ly -= 1
        while y >= ly * size + half: ly += 1
        rx, ry = x - lx * size, y - ly * size
        if ry < -size / 2 or ry >= size / 2 or rx < -size / 2 or rx >= size / 2:
            return None
        if not any(abs(c[0] - lx * size) < 1e-9 and abs(c[1] - ly * size) < 1e-9 for c, _ in self._pf.engine.tiles(self._i)):
            return None                               # no tile of this map holds the point (hybridmap.py:263-272)
        return Position(int(rx / size * dim + dim / 2), int(ry / size * dim + dim / 2))

    def index_to_distance(self, i: int) -> float:     # gridmap.py:333-334
        dim = self._pf.engine.dim
        return float(i - dim / 2) * self._size / dim

    def get_nearby_occ_points(self, curr_cell: Position, centre=(0.0, 0.0)):
        """GridMap.get_nearby_occ_points (gridmap.py:142-155) on the tile centred `centre`: cells > 1.0 within
        int(1.8 / cell_size) cells of curr_cell, as tile-relative distances."""
        pos_range = int(1.8 / self._cell_size)
        q, thr = self._pf.engine.cfg.quantum, self._pf.engine.cfg.occupied_threshold
        for c, cells in self._pf.engine.tiles(self._i):
            if abs(c[0] - centre[0]) < 1e-9 and abs(c[1] - centre[1]) < 1e-9:
                dim = cells.shape[0]
                x0, y0 = max(0, curr_cell.x - pos_range), max(0, curr_cell.y - pos_range)
                x1, y1 = min(dim, curr_cell.x + pos_range), min(dim, curr_cell.y + pos_range)
                i, j = np.nonzero(cells[x0:x1, y0:y1].astype(np.float64) * q > thr)
                return [[self.index_to_distance(int(a) + x0), self.index_to_distance(int(b) + y0)] for a, b in zip(i, j)]
        return []

    def copy(self):
        """A detached snapshot of the tiles (hybridmap.py:315-320): {centre: float64 log-odds [dim][dim]}."""
        q = self._pf.engine.cfg.quantum
        return {c: cells.astype(np.float64) * q for c, cells in self._pf.engine.tiles(self._i)}

    def __len__(self):
        return len(self._pf.engine.tiles(self._i))


class Robot:
    """robot.py:19-157.  `Robot(eng)`: `eng` is the MATLAB engine handle of the reference; the scan matcher is built in
    (both stages of matchScanCustom.m), so the argument is accepted and ignored."""

    def __init__(self, matlab=None):
        global _session
        if _session is None or _session.pf is not None:
            prev = _session
            _session = _Session(*(("velocity", 0.05, 1081, 42, {}) if prev is None else
                                  (prev.motion_model, prev.cell_size, prev.n_beams, prev.seed, prev.options)))
            if prev is not None:
                prev.close()                              # the population before this one is done with: its engine and tile pool go now
        self._s = _session
        self._i = len(self._s.robots)
        self._s.robots.append(self)
        self._map = HybridMap(self._s, self._i)

    # -- reads (robot.py:30-43) --------------------------------------------------------------------------------
    def get_latest_pose(self) -> Pose:
        p = self._s.state("poses")[self._i]
        return Pose(float(p[0]), float(p[1]), float(p[2]))

    def weight(self):
        return [float(self._s.state("weights")[self._i])]

    @property
    def _weight(self):                                # main.py:47,78 read and append to it
        return self.weight()

    @property
    def _cov(self):
        return self._s.state("covs")[self._i]

    def x(self):
        return [p[0] for p in self._s.filter().trajectory(self._i)]

    def y(self):
        return [p[1] for p in self._s.filter().trajectory(self._i)]

    def theta(self):
        return [p[2] for p in self._s.filter().trajectory(self._i)]

    # -- the three calls of the loop ---------------------------------------------------------------------------
    def imu_update(self, reading) -> Pose:            # robot.py:45-57, main.py:144
        pf = self._s.filter()
        if self._s.first_of_round(self, ("imu", id(reading))):
            pf.imu_update(np.asarray(reading.get_data(), dtype=np.float64), float(reading.dt()))
            self._s.invalidate()
        return self.get_latest_pose()

    def map_update(self, scan, last_scan, adj: bool):  # robot.py:59-115, main.py:157,159
        pf = self._s.filter()
        if self._s.first_of_round(self, ("map", id(scan), id(last_scan), bool(adj))):
            pf.engine.set_scan_xy(scan.x(), scan.y())
            ls = None
            if adj:
                ls = np.stack([np.asarray(last_scan.x(), dtype=np.float64), np.asarray(last_scan.y(), dtype=np.float64)], axis=1)
            pf.engine.scan_update(adj=bool(adj), last_scan_xy=ls)
            pf._record(None)
            self._s.invalidate()

    def copy(self):
        """robot.py:141-149: a detached snapshot (histories, covariance, weight, map) - the copies resampling needs are
        made inside the engine."""
        snap = type("RobotSnapshot", (), {})()
        snap._x, snap._y, snap._theta = self.x(), self.y(), self.theta()
        snap._cov = np.array(self._cov)
        snap._weight = self.weight()
        snap._map = self._map.copy()
        return snap

    def __str__(self):
        return "Robot at position: " + str(self.get_latest_pose())


def resample(particles: List[Robot], u: Optional[float] = None) -> List[Robot]:
    """main.py:46-79 for the whole population (one engine call: trigger, systematic resampling with the uniform `u` -
    np.random.random() as in main.py:59 when not given - slot re-pointing and the tile copies of duplicated ancestors).
    Robot i of the returned list is new particle i; the objects are the same."""
    if not particles:
        return particles
    s = particles[0]._s
    s.round_key = None
    s.filter().resample(float(np.random.random()) if u is None else float(u))
    s.invalidate()
    return particles

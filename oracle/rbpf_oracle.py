"""CPU oracle: loop-faithful numpy/Python restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``thesis_amd/`` may import this module;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it, and only as the checker.

Every function restates one reference function (reference = amansanghvi/Thesis,
cited as ``file:line``).  The arithmetic type and the *order of operations* of
the reference are kept (float64 index math with the reference's two different
cell formulas, ``np.longdouble`` accumulators, Bresenham degenerate cases), so
that the golden vectors captured from the imported reference
(``tests/golden/gen_golden.py``) are reproduced bit for bit.

Pinned by: tests/golden/*.npz (G1..G9 of SURVEY.md section 8c), checked in
``tests/test_oracle_golden.py``.  The scan matcher numerics (MATLAB
``matchScansGrid``/``matchScans``) are closed source: **parity unpinned** for
that stage; see ``oracle/matcher_oracle.py``.
"""
from __future__ import annotations

from math import cos, sin, floor, pi
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

# ---------------------------------------------------------------------------
# constants (reference: gridmap.py:20-24, hybridmap.py:18-20,67-68, robot.py:17)
# ---------------------------------------------------------------------------
LOG_ODDS_OCC = 0.80
LOG_ODDS_NEARBY = 0.20
MAX_ODDS_OCC = 3.0
LOG_ODDS_EMP = -0.30
MIN_ODDS_EMP = -3.0
OCCUPIED_POINT_THRESHOLD = 1.0
VALID_DIST_THRESHOLD = 11.0
NUM_SAMPLE_POINTS = 30
MAX_RAY_M = 15.0  # hybridmap.py:107-108


# ---------------------------------------------------------------------------
# a1  scan geometry (lidar.py:76-80, 111-128)
# ---------------------------------------------------------------------------
def scan_xy(ranges: np.ndarray, angles: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Polar -> cartesian, sensor frame.  lidar.py:78-79 (libm cos/sin per beam)."""
    x = np.array([ranges[i] * cos(angles[i]) for i in range(len(ranges))])
    y = np.array([ranges[i] * sin(angles[i]) for i in range(len(ranges))])
    return x, y


def transform(x: np.ndarray, y: np.ndarray, pose) -> Tuple[np.ndarray, np.ndarray]:
    """Rigid transform of beam endpoints into the global frame.

    lidar.py:111-128: ``np.matmul([[c,-s,tx],[s,c,ty],[0,0,1]], vstack(x,y,1))``.
    ``math.cos`` converts a longdouble theta to float64 first; tx/ty keep their
    dtype, so the product is longdouble when the pose is.
    """
    px, py, pth = pose[0], pose[1], pose[2]
    t_mat = [
        [cos(pth), -sin(pth), px],
        [sin(pth), cos(pth), py],
        [0, 0, 1],
    ]
    cur = np.vstack((x, y, [1 for _ in range(len(x))]))
    res = np.matmul(t_mat, cur)
    return res[0], res[1]


# ---------------------------------------------------------------------------
# a5  Bresenham with the reference's degenerate cases (hybridmap.py:274-301)
# ---------------------------------------------------------------------------
def get_affected_points(x0: int, y0: int, x1: int, y1: int) -> List[Tuple[int, int]]:
    dx = abs(x1 - x0)
    dy = abs(y1 - y0)
    if dx == 0:  # hybridmap.py:278-279 -- empty when y1 < y0
        return [(x0, y) for y in range(y0, y1 + 1)]
    if dy == 0:  # hybridmap.py:280-281 -- empty when x1 < x0
        return [(x, y0) for x in range(x0, x1 + 1)]
    xsign = 1 if x1 - x0 > 0 else -1
    ysign = 1 if y1 - y0 > 0 else -1
    steep = dy > dx
    if steep:
        dx, dy = dy, dx
    D = 2 * dy - dx
    y = 0
    out = []
    for x in range(dx + 1):
        if steep:
            out.append((x0 + xsign * y, y0 + ysign * x))
        else:
            out.append((x0 + xsign * x, y0 + ysign * y))
        if D >= 0:
            y += 1
            D -= 2 * dx
        D += 2 * dy
    return out


# ---------------------------------------------------------------------------
# tile index math (gridmap.py:93-94 "set" formula, gridmap.py:126-127 "get")
# ---------------------------------------------------------------------------
def set_index(rel: float, cell_size: float, dim: int) -> int:
    """gridmap.py:93: ``int(x/self._cell_size + len(self._map)/2.0)``."""
    return int(rel / cell_size + dim / 2.0)


def get_index(rel: float, size_m, dim: int) -> int:
    """gridmap.py:126: ``int(x/self._size * len(self._map) + len(self._map)/2)``."""
    return int(rel / size_m * dim + dim / 2)


def map_centre_1d(v: float, map_len_m) -> int:
    """hybridmap.py:193-200 (one axis of _get_map_centre)."""
    approx = int(round(v / map_len_m))
    c = 0
    for a in range(approx - 1, approx + 2):
        mc = a * map_len_m
        if v < mc + map_len_m / 2 and v >= mc - map_len_m / 2:
            c = mc
            break
    return c


# ---------------------------------------------------------------------------
# a5/a4  per-particle tiled map (hybridmap.py:22-145, gridmap.py:26-128)
# ---------------------------------------------------------------------------
class OracleTile:
    """hybridmap.py:22-45 HybridMapEntry + gridmap.py:26-35 GridMap storage."""

    def __init__(self, cx, cy, map_len_m, cell_size):
        self.cx, self.cy = cx, cy
        self.size = map_len_m
        self.cell_size = cell_size
        self.dim = round(map_len_m / cell_size)  # gridmap.py:31
        self.map = np.zeros((self.dim, self.dim))  # [x][y], gridmap.py:32
        r = map_len_m / 2  # hybridmap.py:31-36
        self.min_x, self.min_y = cx - r, cy - r
        self.max_x, self.max_y = cx + r, cy + r

    def is_in_map(self, x, y) -> bool:  # hybridmap.py:44-45
        return x >= self.min_x and x < self.max_x and y >= self.min_y and y < self.max_y

    def get_cell(self, rx, ry) -> Optional[Tuple[int, int]]:  # gridmap.py:120-128
        if ry < -self.size / 2 or ry >= self.size / 2:
            return None
        elif rx < -self.size / 2 or rx >= self.size / 2:
            return None
        return get_index(rx, self.size, self.dim), get_index(ry, self.size, self.dim)

    def rel_cell(self, rx, ry) -> Tuple[int, int]:  # gridmap.py:130-140
        dec_y = dec_x = 0
        if ry < -self.size / 2:
            dec_y = 1
        elif rx < -self.size / 2:
            dec_x = 1
        return (get_index(rx, self.size, self.dim) - dec_x,
                get_index(ry, self.size, self.dim) - dec_y)

    def index_to_distance(self, i: int) -> float:  # gridmap.py:333-334
        return float(i - self.dim / 2) * self.size / self.dim

    def nearby_occ_points(self, cell) -> List[List[float]]:  # gridmap.py:142-155
        pos_range = int(1.8 / self.cell_size)
        sx = max(0, cell[0] - pos_range)
        sy = max(0, cell[1] - pos_range)
        ex = min(self.dim, cell[0] + pos_range)
        ey = min(self.dim, cell[1] + pos_range)
        if ex <= sx or ey <= sy:
            return []
        sub = self.map[sx:ex, sy:ey]
        xs, ys = np.nonzero(sub > OCCUPIED_POINT_THRESHOLD)  # row-major == loop order
        return [[self.index_to_distance(int(a) + sx), self.index_to_distance(int(b) + sy)]
                for a, b in zip(xs, ys)]

    def copy(self):
        t = OracleTile(self.cx, self.cy, self.size, self.cell_size)
        t.map = self.map.copy()
        return t


class OracleHybridMap:
    """hybridmap.py:63-145.  One independent tile list per particle (the reference's
    class-level ``_maps`` list is a latent aliasing bug)."""

    def __init__(self, cell_size=0.05, map_len_m=40,
                 occ=LOG_ODDS_OCC, near=LOG_ODDS_NEARBY, emp=LOG_ODDS_EMP,
                 max_occ=MAX_ODDS_OCC, min_emp=MIN_ODDS_EMP):
        self.cell_size = cell_size
        self.map_len_m = map_len_m
        self.occ, self.near, self.emp = occ, near, emp
        self.max_occ, self.min_emp = max_occ, min_emp
        self.tiles: List[OracleTile] = [OracleTile(0, 0, map_len_m, cell_size)]
        self.cells_visited = 0  # sum over rays of Bresenham points

    # -- lookup ------------------------------------------------------------
    def tile_with_pos(self, x, y) -> Optional[OracleTile]:  # hybridmap.py:263-272
        for t in self.tiles:
            if t.is_in_map(x, y):
                return t
        return None

    def get_odds_at(self, x, y) -> Optional[float]:  # hybridmap.py:85-93
        for t in self.tiles:
            if t.is_in_map(x, y):
                cell = t.get_cell(x - t.cx, y - t.cy)
                if cell is None:
                    return None
                return t.map[cell[0]][cell[1]]
        return None

    # -- update ------------------------------------------------------------
    def update(self, pose, sx: np.ndarray, sy: np.ndarray) -> "OracleHybridMap":
        """hybridmap.py:95-145.  ``sx, sy`` are the sensor-frame endpoints."""
        cs = self.cell_size
        gx, gy = transform(sx, sy, pose)
        if self.tile_with_pos(pose[0], pose[1]) is None:  # :98-100
            return self
        start = (int(pose[0] / cs), int(pose[1] / cs))  # :102
        for i in range(len(gx)):
            end_is_occ = True
            dist = np.sqrt(sx[i] ** 2 + sy[i] ** 2)  # :105
            end = (int(gx[i] / cs), int(gy[i] / cs))  # :106
            if dist > MAX_RAY_M:  # :107-113
                scale = 15.0 / dist
                end = (int(start[0] + scale * (end[0] - start[0])),
                       int(start[1] + scale * (end[1] - start[1])))
                end_is_occ = False
            pts = get_affected_points(start[0], start[1], end[0], end[1])
            self.cells_visited += len(pts)
            for j, ind in enumerate(pts):
                px, py = ind[0] * cs, ind[1] * cs  # :123
                m = self.tile_with_pos(px, py)
                if m is None:  # :125-133
                    ncx = map_centre_1d(px, self.map_len_m)
                    ncy = map_centre_1d(py, self.map_len_m)
                    m = self.tile_with_pos(ncx, ncy)
                    if m is None:
                        m = OracleTile(ncx, ncy, self.map_len_m, cs)
                        self.tiles.append(m)
                rx, ry = px - m.cx, py - m.cy  # :136
                if end_is_occ and ind[0] == end[0] and ind[1] == end[1]:  # :137
                    a, b = set_index(rx, cs, m.dim), set_index(ry, cs, m.dim)
                    m.map[a][b] = min(m.map[a][b] + self.occ, self.max_occ)
                    if j > 0:  # :139-142
                        nx, ny = pts[j - 1][0] * cs, pts[j - 1][1] * cs
                        if m.is_in_map(nx, ny):
                            a = set_index(nx - m.cx, cs, m.dim)
                            b = set_index(ny - m.cy, cs, m.dim)
                            m.map[a][b] = min(m.map[a][b] + self.near, self.max_occ)
                else:  # :144
                    a, b = set_index(rx, cs, m.dim), set_index(ry, cs, m.dim)
                    m.map[a][b] = max(m.map[a][b] + self.emp, self.min_emp)
        return self

    # -- matcher inputs (a6) -------------------------------------------------
    def scan_match_inputs(self, sx, sy, guess, pose_range):
        """hybridmap.py:210-251 up to the engine call: returns exactly the
        arguments handed to ``matchScanCustom``."""
        gx, gy = transform(sx, sy, guess)
        curr: List[Tuple[float, float]] = []
        ref: List[Tuple[float, float]] = []
        for i in range(len(gx)):
            dist = np.sqrt(sx[i] ** 2 + sy[i] ** 2)
            if dist < VALID_DIST_THRESHOLD and dist > 1e-3:
                x, y = gx[i], gy[i]
                for m in self.tiles:
                    if m.is_in_map(x, y):
                        cell = m.get_cell(x - m.cx, y - m.cy)
                        if cell is None:
                            continue
                        curr.append((m.index_to_distance(cell[0]) + m.cx,
                                     m.index_to_distance(cell[1]) + m.cy))
        for cp in curr:
            for mp in self.tiles:
                cell = mp.rel_cell(cp[0] - mp.cx, cp[1] - mp.cy)
                near = mp.nearby_occ_points(cell)
                ref.extend([(p[0] + mp.cx, p[1] + mp.cy) for p in near])
        curr_adj = [[p[0] - guess[0], p[1] - guess[1]] for p in curr]
        uniq = np.unique(ref, axis=0) if len(ref) else []
        ref_adj = [[p[0] - guess[0], p[1] - guess[1]] for p in uniq]
        valid_ref = [[p[0], p[1]] for p in ref_adj
                     if np.sqrt(p[0] ** 2 + p[1] ** 2) < VALID_DIST_THRESHOLD + 0.5]
        valid_curr = [[p[0], p[1]] for p in curr_adj
                      if np.sqrt(p[0] ** 2 + p[1] ** 2) < VALID_DIST_THRESHOLD]
        return (valid_curr, valid_ref, [0.0, 0.0, 0.0], int(1.0 / self.cell_size),
                [pose_range[0], pose_range[1], np.pi / 6])

    def scan_adj_inputs(self, sx, sy, last_gx, last_gy, guess, pose_range):
        """hybridmap.py:147-181 up to the engine call."""
        gx, gy = transform(sx, sy, guess)
        curr_adj = [[gx[i] - guess[0], gy[i] - guess[1]] for i in range(len(gx))]
        ref_adj = [[last_gx[i] - guess[0], last_gy[i] - guess[1]] for i in range(len(last_gx))]
        valid_ref = [[p[0], p[1]] for p in ref_adj if np.sqrt(p[0] ** 2 + p[1] ** 2) < 11.0]
        valid_curr = [[p[0], p[1]] for p in curr_adj if np.sqrt(p[0] ** 2 + p[1] ** 2) < 11.0]
        return (valid_curr, valid_ref, [0.0, 0.0, 0.0], int(1.0 / self.cell_size),
                [pose_range[0], pose_range[1], np.pi / 6])

    # -- misc --------------------------------------------------------------
    def copy(self) -> "OracleHybridMap":  # hybridmap.py:315-320
        m = OracleHybridMap(self.cell_size, self.map_len_m, self.occ, self.near,
                            self.emp, self.max_occ, self.min_emp)
        m.tiles = [t.copy() for t in self.tiles]
        return m

    def nonzero_cells(self) -> Dict[Tuple[int, int], Tuple[np.ndarray, np.ndarray, np.ndarray]]:
        out = {}
        for t in self.tiles:
            xs, ys = np.nonzero(t.map)
            out[(t.cx, t.cy)] = (xs, ys, t.map[xs, ys])
        return out


# ---------------------------------------------------------------------------
# a4  sample weighting (robot.py:118-139)
# ---------------------------------------------------------------------------
def generate_sample_weight(hmap: OracleHybridMap, guesses, sx, sy, motion_prs) -> np.ndarray:
    w = np.zeros(len(guesses), dtype=np.longdouble)
    for i in range(len(guesses)):
        g = guesses[i]
        obs = np.longdouble(1.0)
        ax, ay = transform(sx, sy, (g[0], g[1], g[2]))
        for j in range(len(ax)):
            dist = np.sqrt(sx[j] ** 2 + sy[j] ** 2)
            if dist < 25 and dist > 0.01:
                o = hmap.get_odds_at(ax[j], ay[j])
                if o is not None:
                    obs += o
        w[i] = obs * motion_prs[i]
    return w


# ---------------------------------------------------------------------------
# a3  pose-range clamp, proposal moments (robot.py:59-115)
# ---------------------------------------------------------------------------
def pose_range_from_cov(cov: np.ndarray) -> np.ndarray:
    """robot.py:62-65."""
    pr = np.sqrt(np.diag(cov)) * 30.0
    pr[2] = max(min(4 * pr[2], pi / 3), pi / 8)
    pr[1] = max(min(4 * pr[1], 0.7), 0.1)
    pr[0] = max(min(4 * pr[0], 0.7), 0.1)
    return pr


def mvn_pdf(x: np.ndarray, mean, cov) -> np.ndarray:
    """scipy.stats.multivariate_normal.pdf restated (robot.py:87): pseudo-inverse
    through the symmetric eigendecomposition, log-pdf then exp (scipy
    ``_multivariate.py`` _PSD + _logpdf)."""
    cov = np.asarray(cov, dtype=float)
    mean = np.asarray(mean, dtype=float)
    s, u = np.linalg.eigh(cov)
    eps = 1e6 * np.finfo(float).eps * np.max(np.abs(s))  # _eigvalsh_to_eps (cond factor 1e6 for f64)
    d = s[s > eps]
    s_pinv = np.array([0 if abs(v) <= eps else 1 / v for v in s])
    U = u * np.sqrt(s_pinv)
    log_pdet = np.sum(np.log(d))
    rank = len(d)
    dev = np.atleast_2d(x) - mean
    maha = np.sum(np.square(dev @ U), axis=-1)
    return np.exp(-0.5 * (rank * np.log(2 * np.pi) + log_pdet + maha))


def proposal_moments(guesses: np.ndarray, ksample_weights: np.ndarray):
    """robot.py:89-108.  Returns (mean[3], sigma[3,3], norm) in longdouble."""
    min_w = min(ksample_weights)
    kw = [k - min_w + 1e-2 for k in ksample_weights]
    mean = np.zeros(3, dtype=np.longdouble)
    sigma = np.zeros((3, 3), dtype=np.longdouble)
    norm = np.longdouble(0.0)
    for i in range(len(guesses)):
        mean = np.add(mean, guesses[i] * kw[i])
        norm = norm + kw[i]
    mean = mean / norm
    for i in range(len(guesses)):
        d = np.add(guesses[i], -mean).reshape(1, 3)
        sigma = sigma + d.T * d * kw[i]
    sigma = np.array(sigma) / norm
    norm = norm + min_w * len(kw)
    return mean, sigma, norm


class OracleRobot:
    """robot.py:19-149 with the engine seam reduced to a callable
    ``matcher(curr, ref, guess0, cells_per_m, pose_range) -> (pose, cov, score)``."""

    def __init__(self, cell_size=0.05, map_len_m=40):
        self.map = OracleHybridMap(cell_size, map_len_m)
        self.weight: List = [1.0]
        self.cov = np.zeros((3, 3), dtype=np.longdouble)
        self.x: List = [0.0]
        self.y: List = [0.0]
        self.theta: List = [0.0]

    def pose(self):
        return (self.x[-1], self.y[-1], self.theta[-1])

    def imu_update(self, model: str, data, dt_ticks):
        """robot.py:45-57 with the dataset callbacks of ``imu_*`` below."""
        prev = self.pose()
        nxt = IMU_MODELS[model][0](prev, data, dt_ticks)
        F = IMU_MODELS[model][1](prev, data, dt_ticks)
        self.cov = np.matmul(np.matmul(F, self.cov), F.transpose())
        self.cov = self.cov + IMU_MODELS[model][2](prev, data, dt_ticks)
        self.x.append(nxt[0]); self.y.append(nxt[1]); self.theta.append(nxt[2])
        return nxt

    def map_update(self, sx, sy, match_result, guesses=None, rng=None):
        """robot.py:59-115.  ``match_result`` = (scan_pose[3], scan_cov[3][3], score)
        as returned by Map.get_scan_match (i.e. already offset by the guess).
        ``guesses`` overrides the RNG draw (robot.py:81)."""
        latest = self.pose()
        scan_pose, scan_cov, _score = match_result
        if np.isnan(np.asarray(scan_cov, dtype=float)).any():  # robot.py:73-78
            self.map.update(latest, sx, sy)
            w = generate_sample_weight(self.map, [[latest[0], latest[1], latest[2]]], sx, sy, [1])
            self.weight.append(w[0] + self.weight[-1])
            return None
        if guesses is None:
            guesses = (rng or np.random).multivariate_normal(scan_pose, np.array(scan_cov), NUM_SAMPLE_POINTS)
        motion_prs = mvn_pdf(guesses, scan_pose, scan_cov) * 10
        kw = generate_sample_weight(self.map, guesses, sx, sy, motion_prs)
        mean, sigma, norm = proposal_moments(guesses, kw)
        self.cov = sigma
        self.x.append(mean[0]); self.y.append(mean[1]); self.theta.append(mean[2])
        self.weight.append(norm + self.weight[-1])
        self.map.update((mean[0], mean[1], mean[2]), sx, sy)
        return dict(guesses=guesses, motion_prs=motion_prs, kw=kw, mean=mean, sigma=sigma, norm=norm)

    def copy(self):  # robot.py:141-149
        r = OracleRobot.__new__(OracleRobot)
        r.weight = list(self.weight)
        r.x, r.y, r.theta = list(self.x), list(self.y), list(self.theta)
        r.cov = np.array(self.cov, copy=True)
        r.map = self.map.copy()
        return r


# ---------------------------------------------------------------------------
# a2  motion models (DefaultIMUData.py:26-54, IntelIMUData.py:23-36,
#     Freid101IMUData.py:34-55).  dt_ticks is Reading.dt() (1 tick = 1e-4 s).
# ---------------------------------------------------------------------------
def _vel_progress(prev, d, dt_ticks):  # Freid101IMUData.py:34-41
    dt = dt_ticks / 1e4
    return (prev[0] + d[0] * dt, prev[1] + d[1] * dt, prev[2] + d[2] * dt)


def _vel_F(prev, d, dt_ticks):  # Freid101IMUData.py:44-46
    return np.diag([1.0, 1.0, 1.0])


def _vel_Q_factory(a0, a1, b0, b1):
    def q(prev, d, dt_ticks):  # Freid101IMUData.py:49-55 / IntelRawIMUData.py:51-55
        dt = dt_ticks / 1e4
        return np.abs(np.diag([
            (a0 + a1 * abs(d[0]) * dt) ** 2,
            (a0 + a1 * abs(d[1]) * dt) ** 2,
            (b0 * pi / 180 + b1 * abs(d[2]) * dt) ** 2]))
    return q


def _uni_progress(prev, d, dt_ticks):  # DefaultIMUData.py:26-32
    dt = dt_ticks / 1e4
    th = prev[2] + dt * d[1]
    return (prev[0] + dt * d[0] * cos(th), prev[1] + dt * d[0] * sin(th), th)


def _uni_F(prev, d, dt_ticks):  # DefaultIMUData.py:35-41
    r = np.diag([1.0, 1.0, 1.0])
    dt = dt_ticks / 1e4
    r[0][2] = dt * d[0] * cos(prev[2])
    r[1][2] = dt * d[0] * sin(prev[2])
    return r


def _uni_Q(prev, d, dt_ticks):  # DefaultIMUData.py:44-54
    dt = dt_ticks / 1e4
    su = np.array([[dt * cos(prev[2]), 0], [dt * sin(prev[2]), 0], [0, dt]], dtype=np.longdouble)
    mag = np.diag([0.05 ** 2, (pi / 180 / 2) ** 2])
    out = np.abs(np.matmul(np.matmul(su, mag), su.transpose()))
    noise = np.abs(np.diag([(0.01) ** 2, (0.01) ** 2, (0.2 * pi / 180) ** 2]))
    return out + noise


def _abs_progress(prev, d, dt_ticks):  # IntelIMUData.py:23-25
    return (d[0], d[1], d[2])


def _abs_F(prev, d, dt_ticks):  # IntelIMUData.py:35-36 (roles swapped in the reference)
    return np.abs(np.diag([(0.01) ** 2, (0.01) ** 2, (0.2 * pi / 180) ** 2]))


def _abs_Q(prev, d, dt_ticks):  # IntelIMUData.py:28-32
    r = np.diag([1.0, 1.0, 1.0])
    r[0][2] = d[0] - prev[0]
    r[1][2] = d[1] - prev[1]
    return r


IMU_MODELS = {
    "velocity_fr101": (_vel_progress, _vel_F, _vel_Q_factory(0.02, 0.01, 0.2, 0.02)),
    "velocity_intelraw": (_vel_progress, _vel_F, _vel_Q_factory(0.002, 0.05, 0.01, 0.05)),
    "unicycle": (_uni_progress, _uni_F, _uni_Q),
    "absolute": (_abs_progress, _abs_F, _abs_Q),
}


# ---------------------------------------------------------------------------
# a8  systematic resampling (main.py:46-79)
# ---------------------------------------------------------------------------
def resample_indices(weights: Sequence, u: float):
    """main.py:46-67.  Returns (did_resample, idx list).  ``u`` replaces the
    single ``np.random.random()`` draw (main.py:59)."""
    w = np.array(list(weights))
    n = len(w)
    if not (max(w) - min(w) > 200):
        return False, list(range(n))
    w[w == -np.inf] = 0
    if min(w) < 0:
        w[w != 0] += abs(min(w))
    slice_ = sum(w) / len(w)
    idx: List[int] = []
    start = u * slice_
    curr = 0.0
    for i in range(n):
        curr += w[i]
        num = floor((curr - start) / slice_) - len(idx) + 1
        idx += [i] * num
    if len(idx) != n:
        raise AssertionError("Incorrect number of resampled weights.")
    return True, idx


def resample(particles: List[OracleRobot], u: float) -> List[OracleRobot]:
    """main.py:46-79 on OracleRobot objects."""
    did, idx = resample_indices([p.weight[-1] for p in particles], u)
    if not did:
        return particles
    out, prev = [], -1
    for i in idx:
        out.append(particles[i].copy() if prev == i else particles[i])
        prev = i
    for p in out:
        p.weight.append(1.0)
    return out

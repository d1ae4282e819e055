"""ParticleEngine -- the batched structure-of-arrays engine behind Robot / HybridMap / resample.

One ParticleEngine owns P particles on one MI355X: poses, covariances, weights and one tiled
int8 occupancy map per particle, all resident in HBM.  Every method is one call into
librbpf_hip.so (include/rbpf_hip.h); numpy arrays cross the boundary as plain pointers.
"""
from __future__ import annotations

import atexit
import ctypes as C
import sys
import weakref
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import _lib
from ._lib import RbpfConfig, RbpfCounters, IMU_UNICYCLE, IMU_ABSOLUTE, IMU_VELOCITY

IMU_MODEL_IDS = {"unicycle": IMU_UNICYCLE, "absolute": IMU_ABSOLUTE, "velocity": IMU_VELOCITY}


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


class RbpfError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"librbpf_hip error {code}: {msg}")
        self.code = code


# Engines still open when the interpreter exits are closed from an atexit hook, i.e. while the HIP runtime (and torch,
# whose stream an engine may have borrowed) is still alive; __del__ never touches the GPU during interpreter shutdown.
_LIVE = weakref.WeakSet()


def _close_all_engines():
    for e in list(_LIVE):
        try:
            e.close()
        except Exception:                                    # noqa: BLE001 - nothing useful can be done at exit
            pass


atexit.register(_close_all_engines)


class ParticleEngine:
    """P particles, their maps and the per-scan update on one GPU."""

    def __init__(self, n_particles: int, *, n_samples: int = 30, max_beams: int = 1081,
                 cell_size: float = 0.05, tile_len_m: int = 40, lattice_radius: int = 3,
                 pool_tiles: int = 0, device: int = 0, seed: int = 42, **overrides):
        self._lib = _lib.load()
        cfg = RbpfConfig()
        self._check(self._lib.rbpf_default_config(C.byref(cfg)), None)
        cfg.n_particles, cfg.n_samples, cfg.max_beams = n_particles, n_samples, max_beams
        cfg.cell_size, cfg.tile_len_m, cfg.lattice_radius = cell_size, tile_len_m, lattice_radius
        cfg.pool_tiles, cfg.device, cfg.seed = pool_tiles, device, seed
        for k, v in overrides.items():
            if k == "vel_noise":
                for i in range(4):
                    cfg.vel_noise[i] = v[i]
            else:
                if not hasattr(cfg, k):
                    raise TypeError(f"unknown engine option {k!r}")
                setattr(cfg, k, v)
        self.cfg = cfg
        self._h = C.c_void_p()
        rc = self._lib.rbpf_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            self._h = C.c_void_p()
            raise RbpfError(rc, (self._lib.rbpf_last_error(None) or b"").decode())
        self.P, self.K = n_particles, n_samples
        d = C.c_int32()
        self._check(self._lib.rbpf_get_dim(self._h, C.byref(d)))
        self.dim = d.value
        self.n_beams = 0
        self._borrowed_stream = False
        _LIVE.add(self)

    # -- plumbing ------------------------------------------------------------------------------------
    def _check(self, rc: int, h="self"):
        if rc != 0:
            hh = self._h if h == "self" else None
            raise RbpfError(rc, (self._lib.rbpf_last_error(hh) or b"").decode())

    def close(self):
        """Releases a borrowed stream first (waits for what the engine queued on it), then destroys the handle."""
        if getattr(self, "_h", None) and self._h.value:
            if self._borrowed_stream:
                self._lib.rbpf_release_stream(self._h)
                self._borrowed_stream = False
            self._lib.rbpf_destroy(self._h)
            self._h = C.c_void_p()
        _LIVE.discard(self)

    def __del__(self):
        if sys is None or sys.is_finalizing():               # interpreter shutdown: the atexit hook has run already
            return
        try:
            self.close()
        except Exception:                                    # noqa: BLE001
            pass

    def set_stream(self, stream_ptr: int):
        """Work on the caller's stream from now on (borrowed: never destroyed by the engine)."""
        self._check(self._lib.rbpf_set_stream(self._h, C.c_void_p(stream_ptr)))
        self._borrowed_stream = True

    def release_stream(self):
        """Give a borrowed stream back; the engine works on a stream of its own again."""
        self._check(self._lib.rbpf_release_stream(self._h))
        self._borrowed_stream = False

    def synchronize(self):
        self._check(self._lib.rbpf_synchronize(self._h))

    def set_profiling(self, on):
        """True / False: timing events around every kernel family / none; a list of family names (KERNELS): only those
        (each record costs a few microseconds of stream time).  Restarts the counters."""
        if isinstance(on, (list, tuple, set)):
            mask = 0
            for k in on:
                mask |= 1 << self.KERNELS[k]
            self._check(self._lib.rbpf_set_profiling_families(self._h, mask))
        else:
            self._check(self._lib.rbpf_set_profiling(self._h, int(bool(on))))

    KERNELS = {"raycast": 0, "weight": 1, "resample": 2, "match": 3, "ndt": 4}

    def kernel_ms(self, which) -> np.ndarray:
        """Per-launch durations (ms, HIP events on the engine's stream) since set_profiling(True)."""
        k = self.KERNELS[which] if isinstance(which, str) else int(which)
        out = np.empty(512)
        n = C.c_int32()
        self._check(self._lib.rbpf_get_kernel_ms(self._h, k, _dp(out), len(out), C.byref(n)))
        return out[:n.value].copy()

    def counters(self) -> Dict[str, float]:
        c = RbpfCounters()
        self._check(self._lib.rbpf_get_counters(self._h, C.byref(c)))
        out = {k: getattr(c, k) for k, _ in RbpfCounters._fields_ if k not in ("reserved", "stamp7")}
        out["stamps"] = list(c.reserved) + [c.stamp7]        # eight phase stamps of a -DRBPF_STAMPS build, else zeros
        return out

    # -- a1 ------------------------------------------------------------------------------------------
    def set_scan(self, ranges, angles):
        r, a = _f64(ranges), _f64(angles)
        if r.shape != a.shape or r.ndim != 1:
            raise ValueError("ranges and angles must be 1-D arrays of equal length")
        self._check(self._lib.rbpf_set_scan(self._h, _dp(r), _dp(a), len(r)))
        self.n_beams = len(r)

    def set_scan_xy(self, x, y):
        """The scan from its sensor-frame end points (what a reference Scan object holds, lidar.py:76-87)."""
        x, y = _f64(x), _f64(y)
        if x.shape != y.shape or x.ndim != 1:
            raise ValueError("x and y must be 1-D arrays of equal length")
        self._check(self._lib.rbpf_set_scan_xy(self._h, _dp(x), _dp(y), len(x)))
        self.n_beams = len(x)

    # -- a2 ------------------------------------------------------------------------------------------
    def imu_update(self, model, data, dt_ticks: float):
        mid = IMU_MODEL_IDS[model] if isinstance(model, str) else int(model)
        d = np.zeros(3)
        d[:len(data)] = np.asarray(data, dtype=np.float64)[:3]
        self._check(self._lib.rbpf_imu_update(self._h, mid, _dp(d), float(dt_ticks)))

    # -- a4 ------------------------------------------------------------------------------------------
    def weight_samples(self, guesses, motion_prs) -> np.ndarray:
        g = _f64(guesses)
        K = g.shape[-2]
        g = g.reshape(self.P, K, 3)
        m = _f64(motion_prs).reshape(self.P, K)
        out = np.empty((self.P, K), dtype=np.float64)
        self._check(self._lib.rbpf_weight_samples(self._h, _dp(g), _dp(m), K, _dp(out)))
        return out

    # -- a5 ------------------------------------------------------------------------------------------
    def map_update(self, poses=None):
        if poses is None:
            self._check(self._lib.rbpf_map_update(self._h, None))
        else:
            p = _f64(poses).reshape(self.P, 3)
            self._check(self._lib.rbpf_map_update(self._h, _dp(p)))

    # -- full step -----------------------------------------------------------------------------------
    def scan_update(self, adj: bool = False, last_scan_xy=None, match_override=None, guesses=None):
        ls = None if last_scan_xy is None else _f64(last_scan_xy).reshape(-1, 2)
        mo = None if match_override is None else _f64(match_override).reshape(self.P, 13)
        gs = None if guesses is None else _f64(guesses).reshape(self.P, self.K, 3)
        self._check(self._lib.rbpf_scan_update(
            self._h, int(adj), None if ls is None else _dp(ls), 0 if ls is None else len(ls),
            None if mo is None else _dp(mo), None if gs is None else _dp(gs)))

    def refresh_last_scan(self, particle: int = 0):
        """main.py:167-168 on the device: the current scan at `particle`'s pose becomes the previous scan that
        scan_update(adj=True, last_scan_xy=None) matches against."""
        self._check(self._lib.rbpf_refresh_last_scan(self._h, int(particle)))

    def scan_update_begin(self, adj: bool = False, last_scan_xy=None, match_override=None, guesses=None):
        """First half of scan_update: matcher, proposal, weighting (robot.py:62-114); follow with scan_update_end()."""
        ls = None if last_scan_xy is None else _f64(last_scan_xy).reshape(-1, 2)
        mo = None if match_override is None else _f64(match_override).reshape(self.P, 13)
        gs = None if guesses is None else _f64(guesses).reshape(self.P, self.K, 3)
        self._check(self._lib.rbpf_scan_update_begin(
            self._h, int(adj), None if ls is None else _dp(ls), 0 if ls is None else len(ls),
            None if mo is None else _dp(mo), None if gs is None else _dp(gs)))

    def scan_update_end(self):
        """Second half: the map update at the new mean pose and the NaN branch (robot.py:115, 73-78)."""
        self._check(self._lib.rbpf_scan_update_end(self._h))

    def match_inputs(self, particle: int, guess, cap_ref: int = 1 << 16):
        """The (curr, ref) point lists HybridMap.get_scan_match would hand to the matcher (hybridmap.py:210-242)."""
        g = _f64(guess).reshape(3)
        curr = np.empty((max(self.n_beams, 1), 2)); ref = np.empty((cap_ref, 2))
        nc, nr = C.c_int32(), C.c_int32()
        self._check(self._lib.rbpf_match_inputs(self._h, particle, _dp(g), _dp(curr), C.byref(nc), _dp(ref), C.byref(nr), cap_ref))
        return curr[:nc.value].copy(), ref[:min(nr.value, cap_ref)].copy()

    def resample(self, u: float = float("nan")) -> Tuple[bool, np.ndarray]:
        idx = np.empty(self.P, dtype=np.int32)
        did = C.c_int32()
        self._check(self._lib.rbpf_resample(self._h, float(u), _ip(idx), C.byref(did)))
        return bool(did.value), idx

    def resample_async(self, u: float = float("nan")):
        """resample without reading the ancestor indices back (no host synchronisation)."""
        self._check(self._lib.rbpf_resample(self._h, float(u), None, None))

    # -- state ---------------------------------------------------------------------------------------
    def poses(self) -> np.ndarray:
        out = np.empty((self.P, 3))
        self._check(self._lib.rbpf_get_poses(self._h, _dp(out)))
        return out

    def covs(self) -> np.ndarray:
        out = np.empty((self.P, 3, 3))
        self._check(self._lib.rbpf_get_covs(self._h, _dp(out)))
        return out

    def weights(self) -> np.ndarray:
        out = np.empty(self.P)
        self._check(self._lib.rbpf_get_weights(self._h, _dp(out)))
        return out

    def set_state(self, poses=None, covs=None, weights=None):
        p = None if poses is None else _f64(np.broadcast_to(poses, (self.P, 3)))
        c = None if covs is None else _f64(np.broadcast_to(covs, (self.P, 3, 3)))
        w = None if weights is None else _f64(np.broadcast_to(weights, (self.P,)))
        self._check(self._lib.rbpf_set_state(self._h, None if p is None else _dp(p),
                                             None if c is None else _dp(c), None if w is None else _dp(w)))

    def tiles(self, particle: int) -> List[Tuple[Tuple[float, float], np.ndarray]]:
        """[(centre_xy, cells int8 [dim, dim] indexed [x][y])] of one particle, lattice order."""
        n = C.c_int32()
        self._check(self._lib.rbpf_get_tile_count(self._h, particle, C.byref(n)))
        out = []
        for k in range(n.value):
            c = np.empty(2)
            cells = np.empty((self.dim, self.dim), dtype=np.int8)
            self._check(self._lib.rbpf_get_tile(self._h, particle, k, _dp(c), cells.ctypes.data_as(C.POINTER(C.c_int8))))
            out.append(((float(c[0]), float(c[1])), cells))
        return out

    def set_tile(self, particle: int, centre, cells: np.ndarray):
        cells = np.ascontiguousarray(cells, dtype=np.int8)
        if cells.shape != (self.dim, self.dim):
            raise ValueError("tile must be [dim, dim] int8")
        self._check(self._lib.rbpf_set_tile(self._h, particle, float(centre[0]), float(centre[1]),
                                            cells.ctypes.data_as(C.POINTER(C.c_int8))))

    # -- checkpoint (a portable replacement of the reference's shelve pickles, main.py:183-210) ----
    CHECKPOINT_VERSION = 1

    def save_checkpoint(self, path: str):
        """Everything needed to continue the run bit-identically, as one compressed .npz: configuration, particle
        state, the position of the random streams and every tile (cropped to its non-zero box)."""
        import json
        su, rd = C.c_uint64(), C.c_uint64()
        self._check(self._lib.rbpf_get_rng_state(self._h, C.byref(su), C.byref(rd)))
        owner, centre, box, chunks = [], [], [], []
        for p in range(self.P):
            for c, cells in self.tiles(p):
                xs, ys = np.nonzero(cells)
                b = (0, 0, 0, 0) if len(xs) == 0 else (int(xs.min()), int(xs.max()) + 1, int(ys.min()), int(ys.max()) + 1)
                owner.append(p); centre.append(c); box.append(b)
                chunks.append(cells[b[0]:b[1], b[2]:b[3]].ravel())
        cfg = {k: (list(getattr(self.cfg, k)) if k == "vel_noise" else getattr(self.cfg, k)) for k, _ in self.cfg._fields_}
        np.savez_compressed(path, version=np.array(self.CHECKPOINT_VERSION), config=np.array(json.dumps(cfg)),
                            poses=self.poses(), covs=self.covs(), weights=self.weights(),
                            rng_state=np.array([su.value, rd.value], dtype=np.uint64),
                            tile_owner=np.array(owner, dtype=np.int32), tile_centre=np.array(centre, dtype=np.float64).reshape(-1, 2),
                            tile_box=np.array(box, dtype=np.int32).reshape(-1, 4),
                            tile_cells=np.concatenate(chunks) if chunks else np.empty(0, dtype=np.int8))

    @classmethod
    def from_checkpoint(cls, path: str, device: int = 0) -> "ParticleEngine":
        import json
        with np.load(path, allow_pickle=False) as d:
            if int(d["version"]) != cls.CHECKPOINT_VERSION:
                raise ValueError("unknown checkpoint version")
            cfg = json.loads(str(d["config"]))
            skip = {"n_particles", "n_samples", "max_beams", "cell_size", "tile_len_m", "lattice_radius", "pool_tiles", "device", "seed"}
            over = {k: (tuple(v) if k == "vel_noise" else v) for k, v in cfg.items() if k not in skip}
            e = cls(cfg["n_particles"], n_samples=cfg["n_samples"], max_beams=cfg["max_beams"], cell_size=cfg["cell_size"],
                    tile_len_m=cfg["tile_len_m"], lattice_radius=cfg["lattice_radius"], pool_tiles=cfg["pool_tiles"],
                    device=device, seed=cfg["seed"], **over)
            e.set_state(d["poses"], d["covs"], d["weights"])
            e._check(e._lib.rbpf_set_rng_state(e._h, int(d["rng_state"][0]), int(d["rng_state"][1])))
            off = 0
            for p, c, b in zip(d["tile_owner"], d["tile_centre"], d["tile_box"]):
                n = int((b[1] - b[0]) * (b[3] - b[2]))
                cells = np.zeros((e.dim, e.dim), dtype=np.int8)
                cells[b[0]:b[1], b[2]:b[3]] = d["tile_cells"][off:off + n].reshape(b[1] - b[0], b[3] - b[2])
                off += n
                e.set_tile(int(p), c, cells)
        return e

    def get_odds_at(self, particle: int, xy) -> Tuple[np.ndarray, np.ndarray]:
        pts = _f64(xy).reshape(-1, 2)
        vals = np.empty(len(pts))
        none = np.empty(len(pts), dtype=np.uint8)
        self._check(self._lib.rbpf_get_odds_at(self._h, particle, _dp(pts), len(pts), _dp(vals),
                                               none.ctypes.data_as(C.POINTER(C.c_uint8))))
        return vals, none.astype(bool)


def match_scan(engine: ParticleEngine, curr_xy, ref_xy, guess, cells_per_m: int, pose_range):
    """Stateless twin of the reference's engine seam ``matchScanCustom(curr, ref, guess, cells_per_m,
    pose_range, nargout=3)`` (hybridmap.py:244-251): returns (pose[3], cov[3,3], score); a failed match
    has NaN covariance and score 0 (matchScanCustom.m:25-28)."""
    cu, rf = _f64(curr_xy).reshape(-1, 2), _f64(ref_xy).reshape(-1, 2)
    g, pr = _f64(guess).reshape(3), _f64(pose_range).reshape(3)
    pose, cov, score = np.empty(3), np.empty((3, 3)), C.c_double()
    engine._check(engine._lib.rbpf_match_scan(engine._h, _dp(cu), len(cu), _dp(rf), len(rf), _dp(g), int(cells_per_m),
                                              _dp(pr), _dp(pose), _dp(cov), C.byref(score)))
    return pose, cov, score.value

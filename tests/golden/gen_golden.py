#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Runs only in the build container (``/root/reference`` is absent on the GPU box).
Nothing of the reference is copied: the script imports its modules (with a
build-owned stub for the absent ``matlab`` package, see ``_stub/``), drives single
functions with seeded inputs and stores inputs + outputs as small ``.npz`` files.

Usage:  python tests/golden/gen_golden.py            (writes tests/golden/G*.npz)

Vector sets:
  G1 get_affected_points        G2 tile index formulas     G3 HybridMap.update
  G4 get_odds_at                G5 _generate_sample_weight G6 Robot.map_update
  G7 Robot.imu_update           G8 main.resample           G9 get_scan_match inputs
  G10 debug.mat (data file: a captured matchScanCustom argument tuple)
  G11 dataset adapters (what the Default* / Intel* loaders hand to main.py)
  G12 the Freid101-family adapters on data/orebro.log (OberoIMUData as is; OberoLidarData with its beam count set to
      the file's 181 - as committed it says 360 and cannot parse the file)
"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("RBPF_REFERENCE", "/root/reference")

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "_stub"))
sys.path.insert(0, REF)
sys.path.insert(0, REPO)
os.chdir(REF)  # adapters open ./data/... relatively

import warnings  # noqa: E402
warnings.filterwarnings("ignore")

with contextlib.redirect_stdout(io.StringIO()):
    import gridmap as ref_gridmap  # noqa: E402
    import hybridmap as ref_hybridmap  # noqa: E402
    import lidar as ref_lidar  # noqa: E402
    import models as ref_models  # noqa: E402
    import robot as ref_robot  # noqa: E402
    import main as ref_main  # noqa: E402
    import DefaultIMUData as ref_default_imu  # noqa: E402
    import Freid101IMUData as ref_fr101_imu  # noqa: E402
    import IntelRawIMUData as ref_intelraw_imu  # noqa: E402
    import IntelIMUData as ref_intel_imu  # noqa: E402
    import IntelLidarData as ref_intel_lidar  # noqa: E402

from thesis_amd.datasets import synthetic  # noqa: E402

Pose, Position, Reading = ref_models.Pose, ref_models.Position, ref_models.Reading
Scan = ref_lidar.Scan


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def fresh_hybrid(cell_size=0.05, map_len_m=40):
    """A HybridMap with its own tile list and chosen cell size."""
    ref_hybridmap.HybridMap._maps.clear()
    hm = ref_hybridmap.HybridMap("eng")
    hm._maps = [ref_hybridmap.HybridMapEntry("eng", Position(0, 0), map_len_m, cell_size)]
    hm._cell_size = cell_size
    hm._map_len_m = map_len_m
    ref_hybridmap.HybridMap._maps.clear()
    return hm


def dump_map(hm):
    """-> centres[T,2], offsets[T+1], xs, ys, vals (non-zero cells per tile)."""
    centres, offs, xs, ys, vals = [], [0], [], [], []
    for e in hm._maps:
        m = e.map()._map
        x, y = np.nonzero(m)
        centres.append([e.centre().x, e.centre().y])
        xs.append(x); ys.append(y); vals.append(m[x, y])
        offs.append(offs[-1] + len(x))
    return dict(centres=np.array(centres, dtype=np.float64), offs=np.array(offs),
                xs=np.concatenate(xs).astype(np.int32), ys=np.concatenate(ys).astype(np.int32),
                vals=np.concatenate(vals))


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote", os.path.relpath(path, REPO), {k: np.asarray(v).shape for k, v in arrs.items()})


def intel_scans():
    d = quiet(ref_intel_lidar.IntelLidarData)
    return d.get_scans(), d.get_angles()


# ---------------------------------------------------------------------------
def g1():
    args, flat, offs = [], [], [0]
    for (x0, y0) in [(0, 0), (3, -2), (-5, 4)]:
        for x1 in range(-9, 10):
            for y1 in range(-9, 10):
                pts = ref_hybridmap.HybridMap.get_affected_points(x0, y0, x1, y1)
                args.append((x0, y0, x1, y1)); flat.extend(pts); offs.append(len(flat))
    for a in [(0, 0, 300, 1), (0, 0, 1, 300), (10, -7, -290, 123), (-3, 5, 211, -298), (7, 7, 7, 7),
              (0, 0, -250, -250), (5, 5, 105, 55), (5, 5, 55, 105), (0, 0, 299, 298)]:
        pts = ref_hybridmap.HybridMap.get_affected_points(*a)
        args.append(a); flat.extend(pts); offs.append(len(flat))
    save("G1_affected_points", args=np.array(args, dtype=np.int32),
         pts=np.array(flat, dtype=np.int32).reshape(-1, 2), offs=np.array(offs, dtype=np.int64))


def g2():
    out = {}
    for cs in (0.1, 0.05, 0.025):
        dim = round(40 / cs)
        for centre in (-40, 0, 40):
            lo = int(round((centre - 20) / cs)) - 3
            hi = int(round((centre + 20) / cs)) + 3
            gi = np.arange(lo, hi + 1)
            set_idx = np.full(len(gi), -9999, dtype=np.int32)
            get_idx = np.full(len(gi), -9999, dtype=np.int32)
            inmap = np.zeros(len(gi), dtype=np.bool_)
            e = ref_hybridmap.HybridMapEntry("eng", Position(centre, 0), 40, cs)
            g = e.map()
            for k, i in enumerate(gi):
                pos = int(i) * cs  # hybridmap.py:123
                inmap[k] = e.is_in_map(Position(pos, 0.0))
                rel = pos - e.centre().x  # hybridmap.py:136
                c = g.get_cell(rel, 0.0)
                if c is not None:
                    get_idx[k] = c.x
                if inmap[k]:
                    g._map[:] = 0
                    g.set_empty_pos(rel, 0.0)
                    nz = np.nonzero(g._map)
                    assert len(nz[0]) == 1
                    set_idx[k] = nz[0][0]
            key = "cs%g_c%d" % (cs, centre)
            out[key + "_gi"] = gi.astype(np.int32)
            out[key + "_set"] = set_idx
            out[key + "_get"] = get_idx
            out[key + "_in"] = inmap
    # _get_map_centre (hybridmap.py:193-208) on lattice points and odd values
    hm = fresh_hybrid()
    xs = np.concatenate([np.arange(-2500, 2500) * 0.05, np.array([-20.0, 20.0, 59.99999, 60.0, -60.0, -100.1, 139.95])])
    cen = np.array([[hm._get_map_centre(float(x), float(-x)).x, hm._get_map_centre(float(x), float(-x)).y] for x in xs])
    out["centre_in"] = xs
    out["centre_out"] = cen.astype(np.float64)
    save("G2_index_math", **out)


def room_scan(pose, n_beams=1081, seed=7, max_range=None):
    ang = synthetic.beam_angles(n_beams)
    rng = np.random.Generator(np.random.PCG64(seed))
    r = synthetic.cast_scan(pose, ang, rng)
    if max_range is not None:
        r = np.minimum(r, max_range)
    return r, ang


def g3_g4_g5():
    iscans, iang = intel_scans()
    cases = {}

    # (a) Intel scans 0..5, cs 0.05, small pose drift
    hm = fresh_hybrid(0.05)
    poses = [(0.0, 0.0, 0.0), (0.02, -0.01, 0.01), (0.11, 0.03, -0.05), (0.3, 0.1, 0.2), (0.31, 0.12, 0.9), (-0.2, -0.4, -2.2)]
    for k, p in enumerate(poses):
        quiet(hm.update, Pose(*p), Scan(iscans[k], iang, 0))
    cases["a"] = dict(cs=0.05, poses=np.array(poses), ranges=iscans[:6], angles=iang, **dump_map(hm))
    hm_a = hm

    # (b) room16, 1081 beams, two updates
    hm = fresh_hybrid(0.05)
    poses = [(0.3, -0.2, 0.4), (0.35, -0.15, 0.45)]
    rr = []
    for k, p in enumerate(poses):
        r, ang = room_scan(p, 1081, seed=10 + k)
        rr.append(r)
        quiet(hm.update, Pose(*p), Scan(r, ang, 0))
    cases["b"] = dict(cs=0.05, poses=np.array(poses), ranges=np.array(rr), angles=ang, **dump_map(hm))
    hm_b, ang_b, rr_b = hm, ang, rr

    # (c) rays longer than 15 m (shortened, end not occupied), 361 beams, plus zero ranges
    hm = fresh_hybrid(0.05)
    ang = synthetic.beam_angles(361, np.pi)
    r = 9.0 + 11.0 * (0.5 + 0.5 * np.sin(np.arange(361) * 0.05))
    r[::37] = 0.0
    r[5] = 15.0
    r[6] = 15.000001
    poses = [(1.0, 2.0, 0.3), (-3.02, 1.51, 2.9)]
    for p in poses:
        quiet(hm.update, Pose(*p), Scan(r, ang, 0))
    cases["c"] = dict(cs=0.05, poses=np.array(poses), ranges=np.array([r, r]), angles=ang, **dump_map(hm))

    # (d) tile crossings: near +x edge, near the -x/-y corner
    hm = fresh_hybrid(0.05)
    ang = synthetic.beam_angles(181, np.pi)
    r = np.full(181, 6.0) + 0.37 * np.cos(np.arange(181) * 0.3)
    poses = [(17.3, 0.2, 0.1), (-18.7, -19.2, -2.4), (19.99, 19.99, 0.8), (-19.975, 3.0, 3.1)]
    for p in poses:
        quiet(hm.update, Pose(*p), Scan(r, ang, 0))
    cases["d"] = dict(cs=0.05, poses=np.array(poses), ranges=np.array([r] * len(poses)), angles=ang, **dump_map(hm))
    hm_d = hm

    # (e) cs 0.1 (config C1) on Intel scans
    hm = fresh_hybrid(0.1)
    poses = [(0.0, 0.0, 0.0), (0.25, 0.1, 0.3), (-0.6, 0.2, 1.7)]
    for k, p in enumerate(poses):
        quiet(hm.update, Pose(*p), Scan(iscans[10 + k], iang, 0))
    cases["e"] = dict(cs=0.1, poses=np.array(poses), ranges=iscans[10:13], angles=iang, **dump_map(hm))

    # (f) cs 0.025 (config C5), 181 beams room
    hm = fresh_hybrid(0.025)
    poses = [(0.5, 0.5, 0.0), (0.52, 0.47, -0.2)]
    rr = []
    for k, p in enumerate(poses):
        r, ang = room_scan(p, 181, seed=20 + k)
        rr.append(r)
        quiet(hm.update, Pose(*p), Scan(r, ang, 0))
    cases["f"] = dict(cs=0.025, poses=np.array(poses), ranges=np.array(rr), angles=ang, **dump_map(hm))

    # (g) start pose outside every tile -> update is a no-op (hybridmap.py:98-100)
    hm = fresh_hybrid(0.05)
    r, ang = room_scan((0, 0, 0), 181, seed=3)
    quiet(hm.update, Pose(25.0, 0.0, 0.0), Scan(r, ang, 0))
    cases["g"] = dict(cs=0.05, poses=np.array([(25.0, 0.0, 0.0)]), ranges=np.array([r]), angles=ang, **dump_map(hm))

    flat = {}
    for c, d in cases.items():
        for k, v in d.items():
            flat["%s_%s" % (c, k)] = v
    save("G3_map_update", **flat)

    # ---- G4: get_odds_at on the (b) and (d) maps
    rng = np.random.Generator(np.random.PCG64(99))
    out = {}
    for name, hmx, lim in (("b", hm_b, 12.0), ("d", hm_d, 70.0)):
        pts = rng.uniform(-lim, lim, size=(400, 2))
        pts[:8] = [[0, 0], [-20.0, 0], [20.0, 0], [19.999999, -20.0], [8.0, 8.0], [-8.0, 8.0], [0.05, 0.05], [-0.05, -0.05]]
        vals = np.zeros(len(pts)); isnone = np.zeros(len(pts), dtype=np.bool_)
        for i, (x, y) in enumerate(pts):
            v = hmx.get_odds_at(Position(float(x), float(y)))
            isnone[i] = v is None
            vals[i] = 0.0 if v is None else v
        out[name + "_pts"], out[name + "_vals"], out[name + "_none"] = pts, vals, isnone
    save("G4_get_odds_at", **out)

    # ---- G5: _generate_sample_weight on the (a) map (B=180) and the (b) map (B=1081)
    out = {}
    rb = ref_robot.Robot(None)
    rng = np.random.Generator(np.random.PCG64(5))
    for name, hmx, ranges, ang, centre in (("a", hm_a, iscans[6], iang, (0.3, 0.1, 0.2)),
                                           ("b", hm_b, rr_b[1], ang_b, (0.35, -0.15, 0.45))):
        rb._map = hmx
        guesses = np.array(centre) + rng.normal(0, [0.03, 0.03, 0.01], size=(30, 3))
        prs = rng.uniform(0.5, 20.0, size=30)
        w = rb._generate_sample_weight(guesses, Scan(ranges, ang, 0), prs)
        out[name + "_guesses"], out[name + "_prs"], out[name + "_ranges"], out[name + "_angles"] = guesses, prs, ranges, ang
        out[name + "_w"] = np.asarray(w, dtype=np.float64)
        out[name + "_w_hi"] = np.asarray(w - np.asarray(w, dtype=np.float64).astype(np.longdouble), dtype=np.float64)
    save("G5_sample_weight", **out)
    return hm_a, hm_b


class FakeEngine:
    """Engine seam double (hybridmap.py:244-251): records the call, returns a fixed result."""

    def __init__(self, pose, cov, score):
        self.ret = (pose, cov, score)
        self.calls = []

    def matchScanCustom(self, curr, ref, guess, cells_per_m, pose_range, nargout=3):
        self.calls.append((curr, ref, guess, cells_per_m, list(pose_range)))
        p, c, s = self.ret
        return [list(p)], [list(r) for r in c], s


def g6_g9():
    iscans, iang = intel_scans()
    out = {}
    # Robot with a map built from two room scans, then one map_update with a fixed matcher result
    for name, nb, seed in (("s", 181, 41), ("l", 1081, 42)):
        eng = FakeEngine([0.004, -0.003, 0.002], [[1e-4, 1e-5, 0], [1e-5, 2e-4, 0], [0, 0, 3e-5]], 321.0)
        ref_hybridmap.HybridMap._maps.clear()
        rb = ref_robot.Robot(eng)
        rb._map = fresh_hybrid(0.05)
        rb._map._matlab = eng
        for e in rb._map._maps:
            e._matlab = eng; e.map()._matlab = eng
        true0 = (0.1, 0.05, 0.02)
        r0, ang = room_scan(true0, nb, seed=seed)
        quiet(rb._map.update, Pose(*true0), Scan(r0, ang, 0))
        quiet(rb._map.update, Pose(*true0), Scan(r0, ang, 0))
        quiet(rb._map.update, Pose(*true0), Scan(r0, ang, 0))
        rb._x, rb._y, rb._theta = [0.0, 0.1], [0.0, 0.05], [0.0, 0.02]
        rb._cov = np.array([[4e-6, 0, 0], [0, 5e-6, 0], [0, 0, 1e-6]], dtype=np.longdouble)
        out[name + "_pre_map_" + "centres"] = dump_map(rb._map)["centres"]
        pre = dump_map(rb._map)
        for k, v in pre.items():
            out["%s_pre_%s" % (name, k)] = v
        r1, _ = room_scan((0.12, 0.06, 0.03), nb, seed=seed + 100)
        np.random.seed(1000 + nb)
        state = np.random.get_state()
        quiet(rb.map_update, Scan(r1, ang, 0), None, False)
        # the guesses the reference drew (replay the RNG)
        np.random.set_state(state)
        scan_pose = [0.004 + 0.1, -0.003 + 0.05, 0.002 + 0.02]
        guesses = np.random.multivariate_normal(scan_pose, np.array(eng.ret[1]), 30)
        import scipy.stats as st
        out[name + "_motion_prs"] = st.multivariate_normal.pdf(guesses, scan_pose, eng.ret[1]) * 10
        out[name + "_guesses"] = guesses
        out[name + "_scan_pose"] = np.array(scan_pose)
        out[name + "_scan_cov"] = np.array(eng.ret[1])
        out[name + "_ranges0"], out[name + "_ranges1"], out[name + "_angles"] = r0, r1, ang
        out[name + "_cov_in"] = np.array([[4e-6, 0, 0], [0, 5e-6, 0], [0, 0, 1e-6]])
        out[name + "_pose_out"] = np.array([rb._x[-1], rb._y[-1], rb._theta[-1]], dtype=np.float64)
        out[name + "_cov_out"] = np.array(rb._cov, dtype=np.float64)
        out[name + "_weight_out"] = np.array(rb._weight, dtype=np.float64)
        post = dump_map(rb._map)
        for k, v in post.items():
            out["%s_post_%s" % (name, k)] = v
        # G9: what the engine was handed
        curr, refp, guess, cpm, prange = eng.calls[0]
        out[name + "_m_curr"] = np.array(curr, dtype=np.float64).reshape(-1, 2)
        out[name + "_m_ref"] = np.array(refp, dtype=np.float64).reshape(-1, 2)
        out[name + "_m_cpm"] = np.array(cpm)
        out[name + "_m_range"] = np.array(prange, dtype=np.float64)
        # adj variant (hybridmap.py:147-191)
        eng.calls.clear()
        last = Scan(r0, ang, 0).from_global_reference(Pose(*true0))
        quiet(rb._map.get_scan_adj, Scan(r1, ang, 0), last, Pose(0.12, 0.06, 0.03), np.array([0.3, 0.2, 0.5]))
        curr, refp, guess, cpm, prange = eng.calls[0]
        out[name + "_adj_curr"] = np.array(curr, dtype=np.float64).reshape(-1, 2)
        out[name + "_adj_ref"] = np.array(refp, dtype=np.float64).reshape(-1, 2)

    # NaN-cov ("BAD SCORE") branch, robot.py:73-78
    eng = FakeEngine([0.0, 0.0, 0.0], [[float("nan")] * 3] * 3, 0.0)
    rb = ref_robot.Robot(eng)
    rb._map = fresh_hybrid(0.05); rb._map._matlab = eng
    r0, ang = room_scan((0, 0, 0), 181, seed=77)
    quiet(rb._map.update, Pose(0, 0, 0), Scan(r0, ang, 0))
    rb._x, rb._y, rb._theta = [0.0, 0.02], [0.0, -0.01], [0.0, 0.005]
    quiet(rb.map_update, Scan(r0, ang, 0), None, False)
    out["nan_ranges"], out["nan_angles"] = r0, ang
    out["nan_pose_in"] = np.array([0.02, -0.01, 0.005])
    out["nan_weight_out"] = np.array(rb._weight, dtype=np.float64)
    out["nan_npose"] = np.array(len(rb._x))
    for k, v in dump_map(rb._map).items():
        out["nan_post_" + k] = v
    save("G6_map_update_G9_match_inputs", **out)


def g7():
    out = {}
    rng = np.random.Generator(np.random.PCG64(17))
    models = {
        "unicycle": (ref_default_imu.DefaultIMUData, lambda: np.array([rng.normal(0.8, 0.3), rng.normal(0.0, 0.3)])),
        "velocity_fr101": (ref_fr101_imu.Freid101IMUData, lambda: rng.normal(0, 0.5, 3)),
        "velocity_intelraw": (ref_intelraw_imu.IntelRawIMUData, lambda: rng.normal(0, 0.5, 3)),
        "absolute": (ref_intel_imu.IntelIMUData, None),
    }
    for name, (cls, gen) in models.items():
        rb = ref_robot.Robot(None)
        rb._weight = [1.0]; rb._cov = np.zeros((3, 3), dtype=np.longdouble)
        rb._x, rb._y, rb._theta = [0.0], [0.0], [0.0]
        data, dts, poses, covs = [], [], [], []
        absp = np.zeros(3)
        for k in range(50):
            if gen is None:
                absp = absp + rng.normal(0, 0.05, 3)
                d = absp.copy()
            else:
                d = gen()
            dt = float(rng.integers(100, 3000))
            rd = Reading(d, 0, cls.progress_pose, cls.get_cov_change_matrix, cls.get_cov_input_uncertainty)
            rd.set_dt(dt)
            rb.imu_update(rd)
            data.append(np.resize(d, 3) if len(d) == 3 else np.array([d[0], d[1], 0.0])); dts.append(dt)
            poses.append([rb._x[-1], rb._y[-1], rb._theta[-1]]); covs.append(np.array(rb._cov, dtype=np.float64))
        out[name + "_data"] = np.array(data); out[name + "_dt"] = np.array(dts)
        out[name + "_poses"] = np.array(poses, dtype=np.float64); out[name + "_covs"] = np.array(covs)
    save("G7_imu_update", **out)


class FakeParticle:
    def __init__(self, w, tag):
        self._weight = [w]; self.tag = tag; self.copied = False

    def weight(self):
        return self._weight

    def copy(self):
        p = FakeParticle(self._weight[-1], self.tag); p.copied = True
        return p


def run_resample(weights, u):
    ps = [FakeParticle(w, i) for i, w in enumerate(weights)]
    orig = np.random.random
    np.random.random = lambda: u
    try:
        new = quiet(ref_main.resample, ps)
    finally:
        np.random.random = orig
    idx = np.array([p.tag for p in new], dtype=np.int32)
    copied = np.array([p.copied for p in new])
    wout = np.array([p._weight[-1] for p in new], dtype=np.float64)
    return idx, copied, wout


def g8():
    out = {}
    known = [10, -250, -100, 300, 5, -np.inf, 0, 42]
    for k, u in enumerate((0.25, 0.999, 0.5)):
        idx, cp, wo = run_resample(known, u)
        out["known%d_u" % k] = np.array(u); out["known%d_idx" % k] = idx; out["known%d_copied" % k] = cp
    out["known_w"] = np.array(known, dtype=np.float64)
    idx, cp, wo = run_resample([1.0, 50.0, 200.0, 120.5], 0.5)  # spread <= 200 -> identity
    out["nores_idx"], out["nores_w"], out["nores_wout"] = idx, np.array([1.0, 50.0, 200.0, 120.5]), wo
    rng = np.random.Generator(np.random.PCG64(2024))
    for P in (64, 1024, 16384):
        for v, gen in enumerate((lambda: rng.normal(0, 300, P), lambda: np.abs(rng.normal(0, 1e6, P)) * (rng.random(P) < 0.1),
                                 lambda: np.exp(rng.normal(5, 3, P)))):
            w = gen().astype(np.float64)
            # the reference's weights are np.longdouble after the first map_update (robot.py:114)
            wl = [np.longdouble(x) for x in w]
            u = float(rng.random())
            idx, cp, wo = run_resample(wl, u)
            out["r%d_%d_w" % (P, v)] = w; out["r%d_%d_u" % (P, v)] = np.array(u); out["r%d_%d_idx" % (P, v)] = idx
    save("G8_resample", **out)


def g10():
    """debug.mat: one captured argument tuple of matchScanCustom (inputs only, no outputs exist)."""
    from scipy.io import loadmat
    d = loadmat(os.path.join(REF, "debug.mat"))
    save("G10_debug_mat_inputs", curr=d["curr"].astype(np.float64), ref=d["ref"].astype(np.float64),
         guess=d["guess"].astype(np.float64).ravel(), resolution=np.array(int(d["resolution"].ravel()[0])),
         rng=d["rng"].astype(np.float64).ravel())


def g11():
    """Dataset adapters: what DefaultLidarData / DefaultIMUData / IntelLidarData / IntelIMUData
    hand to main.py, sampled (first/last records, lengths, sums) so that the build's own loaders can be checked."""
    import DefaultLidarData as ref_default_lidar
    out = {}
    t, sc, ang = quiet(ref_default_lidar.DefaultLidarData().load_and_format)
    out["dl_n"] = np.array(sc.shape); out["dl_times_head"] = np.asarray(t[:5], dtype=np.float64); out["dl_times_tail"] = np.asarray(t[-5:], dtype=np.float64)
    out["dl_scans_head"] = np.asarray(sc[:3], dtype=np.float64); out["dl_scan_sum"] = np.asarray(sc, dtype=np.float64).sum(axis=1)[:50]
    out["dl_angles"] = np.asarray(ang, dtype=np.float64)
    d, ti = quiet(ref_default_imu.DefaultIMUData().load_and_format)
    out["di_n"] = np.array(d.shape); out["di_head"] = np.asarray(d[:20], dtype=np.float64); out["di_tail"] = np.asarray(d[-5:], dtype=np.float64)
    out["di_times_head"] = np.asarray(ti[:20], dtype=np.float64); out["di_times_tail"] = np.asarray(ti[-5:], dtype=np.float64)
    t, sc, ang = quiet(ref_intel_lidar.IntelLidarData().load_and_format)
    out["il_n"] = np.array(np.asarray(sc).shape); out["il_times_head"] = np.asarray(t[:5], dtype=np.float64)
    out["il_scan_sum"] = np.asarray(sc, dtype=np.float64).sum(axis=1)[:50]
    d, ti = quiet(ref_intel_imu.IntelIMUData().load_and_format)
    out["ii_n"] = np.array(np.asarray(d).shape); out["ii_head"] = np.asarray(d[:10], dtype=np.float64); out["ii_times_head"] = np.asarray(ti[:10], dtype=np.float64)
    save("G11_dataset_adapters", **out)


def g12():
    """Freid101-family adapters (Freid101IMUData.py:9-32 shape) on the one log of that family that is in the tree:
    data/orebro.log through OberoIMUData / OberoLidarData.  The lidar adapter's module constant POINTS_PER_SCAN (360,
    OberoLidarData.py:8) is set to the 181 beams the file has before the reference's own parsing code runs; nothing
    else is touched.  Full arrays (237 scans, 238 odometry records): the build's loader must reproduce them exactly."""
    import OberoIMUData as ref_obero_imu
    import OberoLidarData as ref_obero_lidar
    out = {}
    d, ti = quiet(ref_obero_imu.OberoIMUData().load_and_format)
    out["oi_data"] = np.asarray(d, dtype=np.float64); out["oi_times"] = np.asarray(ti, dtype=np.int64)
    ref_obero_lidar.POINTS_PER_SCAN = 181
    t, sc, ang = quiet(ref_obero_lidar.OberoLidarData().load_and_format)
    out["ol_times"] = np.asarray(t, dtype=np.int64); out["ol_scans"] = np.asarray(sc, dtype=np.float64); out["ol_angles"] = np.asarray(ang, dtype=np.float64)
    save("G12_obero_adapters", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g345", "g69", "g7", "g8", "g10", "g11", "g12"]
    if "g1" in which: g1()
    if "g2" in which: g2()
    if "g345" in which: g3_g4_g5()
    if "g69" in which: g6_g9()
    if "g7" in which: g7()
    if "g8" in which: g8()
    if "g10" in which: g10()
    if "g11" in which: g11()
    if "g12" in which: g12()

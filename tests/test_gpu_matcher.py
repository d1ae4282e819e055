"""Scan matcher (own specification; the reference's MATLAB matcher is closed source: PARITY UNPINNED).
What can be tested without reference outputs: the interface, the validity gate, and that known
rigid offsets are recovered to within the search resolution."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rot(th):
    c, s = np.cos(th), np.sin(th)
    return np.array([[c, -s], [s, c]])


@pytest.fixture(scope="module")
def eng():
    from thesis_amd.engine import ParticleEngine
    e = ParticleEngine(4, max_beams=1081, ndt_refine=0)      # the grid stage on its own
    yield e
    e.close()


def test_match_scan_identity_on_debug_mat(golden, eng):
    """debug.mat: the one real argument tuple the reference holds for matchScanCustom (inputs only)."""
    from thesis_amd.engine import match_scan
    g = golden("G10_debug_mat_inputs")
    # matching the reference points against themselves must return the guess
    pose, cov, score = match_scan(eng, g["ref"], g["ref"], [0, 0, 0], int(g["resolution"]), [0.3, 0.3, np.pi / 6])
    assert np.allclose(pose, 0, atol=1e-12) and score > 0.9 * len(g["ref"]) and np.all(np.isfinite(cov))
    assert np.all(np.linalg.eigvalsh(cov) > 0)
    # the recorded call itself runs and yields either a valid pose inside the window or the NaN signal
    pose, cov, score = match_scan(eng, g["curr"], g["ref"], g["guess"], int(g["resolution"]), g["rng"])
    if np.isnan(cov).any():
        assert score == 0.0
    else:
        assert np.all(np.abs(pose[:2] - g["guess"][:2]) < g["rng"][:2])


@pytest.mark.parametrize("off", [(0.10, -0.05, 0.03), (-0.35, 0.20, -0.20), (0.0, 0.55, 0.45), (0.62, -0.62, -0.5)])
def test_match_scan_recovers_known_offset(golden, eng, off):
    from thesis_amd.engine import match_scan
    g = golden("G10_debug_mat_inputs")
    ref = g["ref"]
    dx, dy, dth = off
    # curr = the same points seen from a sensor displaced by `off`:  ref = R(dth) curr + t
    curr = (ref - [dx, dy]) @ rot(dth)            # = R(-dth) (ref - t)
    pose, cov, score = match_scan(eng, curr, ref, [0, 0, 0], 20, [0.7, 0.7, np.pi / 6])
    assert np.all(np.isfinite(cov))
    assert abs(pose[0] - dx) <= 0.051 and abs(pose[1] - dy) <= 0.051
    assert abs(pose[2] - dth) <= 3 * (0.05 / 15.0) + 1e-9      # three fine rotation steps (cell / MaxRange)
    assert score > 0.6 * len(ref)


def test_match_scan_failure_signal(eng):
    from thesis_amd.engine import match_scan
    rng = np.random.Generator(np.random.PCG64(3))
    curr = rng.uniform(-5, 5, size=(100, 2))
    ref = rng.uniform(40, 45, size=(50, 2))        # no overlap inside the window
    pose, cov, score = match_scan(eng, curr, ref, [0, 0, 0], 20, [0.5, 0.5, np.pi / 6])
    assert np.isnan(cov).all() and score == 0.0    # matchScanCustom.m:25-28


def build_room_engine(P, poses0, n_scans=4):
    from thesis_amd.engine import ParticleEngine
    from thesis_amd.datasets import synthetic
    ang = synthetic.beam_angles(1081)
    rng = np.random.Generator(np.random.PCG64(11))
    e = ParticleEngine(P, max_beams=1081)
    for _ in range(n_scans):
        r = synthetic.cast_scan(poses0, ang, rng)
        e.set_scan(r, ang)
        e.map_update(np.broadcast_to(poses0, (P, 3)))
    return e, ang, rng


@pytest.mark.parametrize("adj", [False, True])
def test_scan_update_with_builtin_matcher_pulls_particles_to_truth(adj):
    """Particles start with odometry errors inside the search window; after one full scan update with
    the built-in matcher the posterior poses are within ~1.5 cells / 0.5 degrees of the truth."""
    from thesis_amd.datasets import synthetic
    from oracle import rbpf_oracle as orc
    P = 16
    true0 = np.array([0.5, -0.3, 0.2])
    e, ang, rng = build_room_engine(P, true0)
    true1 = np.array([0.8, -0.1, 0.3])
    r1 = synthetic.cast_scan(true1, ang, rng)
    err = np.stack([rng.uniform(-0.3, 0.3, P), rng.uniform(-0.3, 0.3, P), rng.uniform(-0.15, 0.15, P)], axis=1)
    err[0] = 0
    guess = true1 + err
    cov = np.diag([0.02 ** 2, 0.02 ** 2, 0.01 ** 2])          # -> pose_range clamps to 0.7 m (robot.py:62-65)
    e.set_state(poses=guess, covs=cov, weights=1.0)
    e.set_scan(r1, ang)
    last = None
    if adj:
        r0 = synthetic.cast_scan(true0, ang, rng)
        sx, sy = orc.scan_xy(r0, ang)
        gx, gy = orc.transform(sx, sy, tuple(true0))
        last = np.stack([gx, gy], axis=1)
    e.scan_update(adj=adj, last_scan_xy=last)
    post = e.poses()
    d = post - true1
    before = np.abs(err).max(axis=0)
    assert np.all(np.abs(d[:, :2]) < 0.08), (np.abs(d).max(axis=0), before)
    assert np.all(np.abs(d[:, 2]) < 0.01), (np.abs(d).max(axis=0), before)
    assert np.all(np.isfinite(e.weights())) and np.all(np.isfinite(e.covs()))
    e.close()


def test_builtin_matcher_failure_takes_nan_branch():
    """An empty map gives the matcher nothing to hit: NaN covariance -> robot.py:73-78 (pose kept, map
    updated, weight incremented)."""
    from thesis_amd.engine import ParticleEngine
    from thesis_amd.datasets import synthetic
    ang = synthetic.beam_angles(361, np.pi)
    r = synthetic.cast_scan((0, 0, 0), ang, None)
    e = ParticleEngine(3, max_beams=361)
    e.set_state(poses=[0.1, 0.2, 0.05])
    e.set_scan(r, ang)
    e.scan_update()
    np.testing.assert_array_equal(e.poses(), np.broadcast_to([0.1, 0.2, 0.05], (3, 3)))
    assert np.all(e.weights() != 1.0)
    (_, cells), = e.tiles(0)
    assert np.count_nonzero(cells) > 1000
    e.close()


@pytest.mark.parametrize("cs", [0.05, 0.025])
def test_field_staging_fast_path_equals_bit_by_bit(monkeypatch, cs):
    """(0.025 m maps: a matcher cell is 2 x 2 map cells, the fast path works on 64-column windows.)  The matcher stages the particle's own map as funnel-shifted occupancy words plus the 'one cell lower' columns
    of the reference's index formula; RBPF_MATCH_STAGE=slow reads every bit through the index map instead.  Both
    must give the same match (pose, covariance, score) for particles on the irregular negative side of the map and
    next to a tile edge."""
    from thesis_amd import engine
    from thesis_amd.datasets import synthetic
    B = 1081
    ang = synthetic.beam_angles(B)
    rng = np.random.Generator(np.random.PCG64(4))
    poses = np.array([[-11.0, -12.5, 0.3], [-3.0, -9.9, -1.2], [18.2, -18.7, 2.0], [0.4, 0.3, 0.1], [-15.5, 3.0, 1.0], [19.4, 19.1, -0.6]])
    P = len(poses)
    scans = [5.0 + 2.0 * np.sin(3 * ang) + rng.normal(0, 0.01, B), 4.0 + 2.5 * np.cos(5 * ang) + rng.normal(0, 0.01, B)]
    outs = []
    for mode in ("fast", "slow"):
        if mode == "slow":
            monkeypatch.setenv("RBPF_MATCH_STAGE", "slow")
        else:
            monkeypatch.delenv("RBPF_MATCH_STAGE", raising=False)
        e = engine.ParticleEngine(P, max_beams=B, pool_tiles=64, seed=3, cell_size=cs)
        e.set_state(poses=poses)
        e.set_scan(scans[0], ang)
        e.map_update(poses)
        e.set_scan(scans[1], ang)
        e.map_update(poses)
        e.set_state(poses=poses + [0.08, -0.05, 0.02], covs=np.diag([4e-5, 4e-5, 1e-5]))
        e.set_scan(scans[1], ang)
        e.scan_update(adj=False)
        outs.append((e.poses(), e.covs(), e.weights()))
        e.close()
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)


# ---- NDT refinement stage (matchScanCustom.m:32-50; rbpf_config.ndt_refine) -----------------------------------------
def wall_points(half=5.0, thick=2, cs=0.05):
    """Reference points of a square room whose walls are `thick` cells thick, on the cell-corner lattice the
    reference's point lists live on (gridmap.py:333-334)."""
    k = int(round(half / cs))
    line = np.arange(-k, k + 1)
    pts = []
    for t in range(thick):
        for sgn in (-1, 1):
            pts += [np.stack([np.full_like(line, sgn * (k + t)), line], 1), np.stack([line, np.full_like(line, sgn * (k + t))], 1)]
    return np.unique(np.concatenate(pts), axis=0).astype(np.float64) * cs


@pytest.fixture(scope="module")
def eng_ndt():
    from thesis_amd.engine import ParticleEngine
    e = ParticleEngine(4, max_beams=1081, ndt_refine=2)
    yield e
    e.close()


@pytest.fixture(scope="module")
def eng_grid():
    from thesis_amd.engine import ParticleEngine
    e = ParticleEngine(4, max_beams=1081, ndt_refine=0)
    yield e
    e.close()


@pytest.mark.parametrize("guess", [(0.0, 0.0, 0.0), (0.07, -0.13, 0.02)])     # region origins of both parities
@pytest.mark.parametrize("off", [(0.013, -0.021, 0.0007), (0.31, 0.18, -0.11), (-0.44, 0.27, 0.2)])
def test_ndt_stage_equals_oracle(eng_grid, eng_ndt, off, guess):
    """The HIP NDT stage against oracle/matcher_oracle.py, both started from the correlative optimum: same pose, same
    score, same number of evaluations.  (Against MATLAB's matchScans: parity unpinned.)"""
    from thesis_amd.engine import match_scan
    from oracle import matcher_oracle as mo
    ref = wall_points()
    inner = ref[(np.abs(ref[:, 0]) <= 5.0) & (np.abs(ref[:, 1]) <= 5.0)][::3]     # the scan sees the inner face
    dx, dy, dth = off
    g_abs = np.array(guess)                           # the guess handed to the matcher (its region origin follows it)
    curr = (inner - [dx, dy]) @ rot(dth - g_abs[2])   # the scan as seen from a sensor at (dx, dy, dth), in the guess's rotation
    rng3 = [0.7, 0.7, np.pi / 6]
    p0, cov0, s0 = match_scan(eng_grid, curr, ref, g_abs, 20, rng3)
    before = eng_ndt.counters()
    p1, cov1, s1 = match_scan(eng_ndt, curr, ref, g_abs, 20, rng3)
    after = eng_ndt.counters()
    assert np.all(np.isfinite(cov0)) and np.array_equal(cov0, cov1)               # the covariance stays the grid one
    mcs, N = 0.05, 672                                                            # stateless twin: MaxRange 15 => 672 cells
    occ, ox, oy = mo.rasterise(ref, g_abs, mcs, N, 0.5, 15.0)
    pts = mo.beams_in_cells(curr, mcs)
    X0, Y0 = g_abs[0] / mcs - ox + 0.5, g_abs[1] / mcs - oy + 0.5
    # the grid optimum is a whole number of cells and rotation steps away from the guess
    start = (X0 + np.rint((p0[0] - g_abs[0]) / mcs), Y0 + np.rint((p0[1] - g_abs[1]) / mcs), p0[2])
    pw, score, evals = mo.ndt_refine(occ, pts, start, 2, ox, oy)
    want = np.array([g_abs[0] + (pw[0] - X0) * mcs, g_abs[1] + (pw[1] - Y0) * mcs, pw[2]])
    assert after["ndt_runs"] - before["ndt_runs"] == 1 and after["ndt_accepted"] - before["ndt_accepted"] == 1
    assert after["ndt_evaluations"] - before["ndt_evaluations"] == evals
    assert np.allclose(p1, want, rtol=0, atol=1e-6)                 # metres / radians; float terms, expf vs numpy's exp
    assert abs(s1 - score) <= 1e-6 * score
    # the ascent is monotone and stays within a cell of the constructed offset (the Gaussians sit on the middle of the
    # two-cell walls while the points lie on their inner face: a bias below one cell, not an error of the stage)
    assert score >= -mo.ndt_eval(occ, pts, start, 2, ox, oy, True)[0] - 1e-6 * score
    if not any(guess):
        assert abs(p1[0] - dx) < 0.05 and abs(p1[1] - dy) < 0.05 and abs(p1[2] - dth) < 8e-3


def test_ndt_acceptance_rule(eng_grid, eng_ndt):
    """matchScanCustom.m:38-44: the NDT pose is taken iff it is valid and 2 * ndtScore > gridScore."""
    from thesis_amd.engine import ParticleEngine, match_scan
    ref = wall_points()
    inner = ref[(np.abs(ref[:, 0]) <= 5.0) & (np.abs(ref[:, 1]) <= 5.0)][::3]
    curr = (inner - [0.12, -0.07]) @ rot(0.01)
    e1 = ParticleEngine(1, max_beams=1081, ndt_refine=1)
    try:
        for r in (ref, ref[::7]):            # dense walls: NDT cells hold >= 3 points; thinned: almost none do
            pg, _, sg = match_scan(eng_grid, curr, r, [0, 0, 0], 20, [0.7, 0.7, np.pi / 6])
            pa, _, sa = match_scan(eng_ndt, curr, r, [0, 0, 0], 20, [0.7, 0.7, np.pi / 6])   # always-take mode: the NDT score
            p1, _, s1 = match_scan(e1, curr, r, [0, 0, 0], 20, [0.7, 0.7, np.pi / 6])
            if 2 * sa > sg:
                assert np.array_equal(p1, pa) and s1 == sa
            else:
                assert np.array_equal(p1, pg) and s1 == sg
        assert e1.counters()["ndt_runs"] == 2 and e1.counters()["ndt_accepted"] == 1
    finally:
        e1.close()


def test_ndt_inactive_when_cells_cannot_hold_three_points():
    """0.1 m matcher cells (config C1): an NDT cell of 0.1 m is one matcher cell, no Gaussian can be formed."""
    from thesis_amd.engine import ParticleEngine, match_scan
    ref = wall_points(cs=0.1)
    e = ParticleEngine(1, max_beams=1081, cell_size=0.1, ndt_refine=1)
    try:
        pose, cov, score = match_scan(e, ref[::2], ref, [0, 0, 0], 10, [0.7, 0.7, np.pi / 6])
        assert np.all(np.isfinite(cov)) and e.counters()["ndt_runs"] == 0
    finally:
        e.close()


def test_scan_update_with_ndt_stage_runs_per_particle():
    """The engine path: every particle's match enters the NDT stage, and the posterior is at least as close to the
    truth as the grid stage alone leaves it."""
    from thesis_amd.engine import ParticleEngine
    from thesis_amd.datasets import synthetic
    P = 16
    true0, true1 = np.array([0.5, -0.3, 0.2]), np.array([0.8, -0.1, 0.3])
    ang = synthetic.beam_angles(1081)
    errs = {}
    for mode in (0, 1):
        rng = np.random.Generator(np.random.PCG64(11))
        e = ParticleEngine(P, max_beams=1081, ndt_refine=mode)
        for _ in range(8):
            e.set_scan(synthetic.cast_scan(true0, ang, rng), ang)
            e.map_update(np.broadcast_to(true0, (P, 3)))
        r1 = synthetic.cast_scan(true1, ang, rng)
        err = np.stack([rng.uniform(-0.3, 0.3, P), rng.uniform(-0.3, 0.3, P), rng.uniform(-0.15, 0.15, P)], axis=1)
        e.set_state(poses=true1 + err, covs=np.diag([0.02 ** 2, 0.02 ** 2, 0.01 ** 2]), weights=1.0)
        e.set_scan(r1, ang)
        e.scan_update()
        c = e.counters()
        assert c["ndt_runs"] == (P if mode else 0) and c["ndt_accepted"] <= c["ndt_runs"]
        assert c["ndt_evaluations"] >= c["ndt_runs"]
        errs[mode] = np.abs(e.poses() - true1)
        assert np.all(np.isfinite(e.weights())) and np.all(np.isfinite(e.covs()))
        e.close()
    assert np.all(errs[1][:, :2] < 0.08) and np.all(errs[1][:, 2] < 0.01)
    print("mean |error| grid only", errs[0].mean(axis=0), "with NDT", errs[1].mean(axis=0))

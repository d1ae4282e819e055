"""Parity of the HIP path (through the C ABI) against the oracle and the golden vectors.
GPU only:  python -m pytest tests -m gpu"""
import numpy as np
import pytest

from oracle import rbpf_oracle as orc
from tests.helpers import oracle_map_from_dump, golden_dump_as_dict

pytestmark = pytest.mark.gpu

Q = 0.1


@pytest.fixture(scope="module")
def eng_mod():
    from thesis_amd import engine
    return engine


def select_map_kernel(monkeypatch, name):
    """"auto" = the default chain (event-walk kernel, then windows for what it gives back); "ray" = the global-index kernel
    of round 2 first; "window" = the 128x128-window kernel alone - through RBPF_MAP_KERNEL, which rbpf_create reads."""
    if name == "auto":
        monkeypatch.delenv("RBPF_MAP_KERNEL", raising=False)
    else:
        monkeypatch.setenv("RBPF_MAP_KERNEL", name)


@pytest.fixture(params=["auto", "ray", "window"])
def map_kernel(request, monkeypatch):
    """The default chain (= the event-walk kernel first), the global-index kernel first, the window kernel on its own."""
    select_map_kernel(monkeypatch, request.param)
    return request.param


def to_lattice(vals):
    """float64 log-odds of the reference -> int8 lattice value; asserts the lattice claim."""
    q = np.rint(np.asarray(vals) / Q)
    assert np.max(np.abs(np.asarray(vals) - q * Q), initial=0.0) < 1e-9
    return q.astype(np.int8)


def load_dump_into(engine, particle, dump, dim):
    for (cx, cy), (xs, ys, vals) in dump.items():
        cells = np.zeros((dim, dim), dtype=np.int8)
        cells[xs, ys] = to_lattice(vals)
        engine.set_tile(particle, (cx, cy), cells)


def assert_tiles_equal(engine, particle, dump, dim):
    got = {c: cells for c, cells in engine.tiles(particle)}
    assert set(got.keys()) == set(dump.keys())
    for c, (xs, ys, vals) in dump.items():
        want = np.zeros((dim, dim), dtype=np.int8)
        want[xs, ys] = to_lattice(vals)
        if not np.array_equal(got[c], want):
            bad = np.argwhere(got[c] != want)
            raise AssertionError(f"tile {c}: {len(bad)} cells differ, first {bad[:5].tolist()} "
                                 f"got {got[c][tuple(bad[0])]} want {want[tuple(bad[0])]}")


def oracle_dump(hm):
    out = {}
    for t in hm.tiles:
        xs, ys = np.nonzero(t.map)
        out[(float(t.cx), float(t.cy))] = (xs, ys, t.map[xs, ys])
    return out


# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["b", "d"])
def test_get_odds_at_golden(golden, eng_mod, case):
    g3, g4 = golden("G3_map_update"), golden("G4_get_odds_at")
    e = eng_mod.ParticleEngine(2, max_beams=1081, pool_tiles=16)
    load_dump_into(e, 1, golden_dump_as_dict(g3, case + "_"), e.dim)
    vals, none = e.get_odds_at(1, g4[case + "_pts"])
    assert np.array_equal(none, g4[case + "_none"])
    np.testing.assert_allclose(vals, g4[case + "_vals"], rtol=0, atol=1e-12)
    # particle 0 still has its empty centre tile only
    vals0, none0 = e.get_odds_at(0, g4[case + "_pts"])
    assert np.all(vals0 == 0)
    e.close()


@pytest.mark.parametrize("case", ["a", "b"])
def test_weight_samples_golden(golden, eng_mod, case):
    g3, g5 = golden("G3_map_update"), golden("G5_sample_weight")
    P = 3
    e = eng_mod.ParticleEngine(P, max_beams=1081, pool_tiles=16)
    for p in range(P):
        load_dump_into(e, p, golden_dump_as_dict(g3, case + "_"), e.dim)
    e.set_scan(g5[case + "_ranges"], g5[case + "_angles"])
    guesses = np.broadcast_to(g5[case + "_guesses"], (P, 30, 3))
    prs = np.broadcast_to(g5[case + "_prs"], (P, 30))
    w = e.weight_samples(guesses, prs)
    for p in range(P):
        np.testing.assert_allclose(w[p], g5[case + "_w"], rtol=1e-12, atol=1e-9)   # north-star tolerance is 1e-4
    e.close()


@pytest.mark.parametrize("case", list("abcdefg"))
def test_map_update_golden(golden, eng_mod, case, map_kernel):
    g = golden("G3_map_update")
    cs = float(g[case + "_cs"])
    P = 2
    e = eng_mod.ParticleEngine(P, max_beams=1081, cell_size=cs, pool_tiles=16)
    for pose, r in zip(g[case + "_poses"], g[case + "_ranges"]):
        e.set_scan(r, g[case + "_angles"])
        e.map_update(np.broadcast_to(pose, (P, 3)))
    for p in range(P):
        assert_tiles_equal(e, p, golden_dump_as_dict(g, case + "_"), e.dim)
    e.close()


def test_map_update_random_particles_vs_oracle(eng_mod, map_kernel):
    """Distinct poses per particle, three scans, compared cell by cell with the oracle."""
    from thesis_amd.datasets import synthetic
    rng = np.random.Generator(np.random.PCG64(321))
    P, B = 6, 1081
    ang = synthetic.beam_angles(B)
    e = eng_mod.ParticleEngine(P, max_beams=B, pool_tiles=32)
    maps = [orc.OracleHybridMap(0.05) for _ in range(P)]
    base = np.array([[0.0, 0.0, 0.0], [3.1, -2.2, 1.0], [-5.5, 5.0, -2.0], [6.9, 6.9, 0.7], [-7.2, -0.4, 3.0], [0.01, 7.5, -1.57]])
    for step in range(3):
        true = base[0] + np.array([0.1 * step, 0.05 * step, 0.02 * step])
        r = synthetic.cast_scan(true, ang, rng)
        if step == 1:
            r[100:140] = 0.0        # invalid returns: "occupied" hit on the start cell
            r[500:520] = 29.0       # long rays (> 15 m): shortened, end not occupied
        poses = base + rng.normal(0, 0.02, size=base.shape)
        e.set_scan(r, ang)
        e.map_update(poses)
        sx, sy = orc.scan_xy(r, ang)
        for p in range(P):
            maps[p].update(tuple(float(v) for v in poses[p]), sx, sy)
    for p in range(P):
        assert_tiles_equal(e, p, oracle_dump(maps[p]), e.dim)
    c = e.counters()
    assert c["ray_cells_visited"] > 0 and c["cells_written"] > 0
    if map_kernel == "window":
        assert c["window_fallbacks"] == 0
    e.close()


@pytest.mark.parametrize("kernel", ["auto", "ray"])
@pytest.mark.parametrize("scene", ["near_wall", "one_direction", "tile_corner", "negative_side", "short_rays"])
def test_map_update_first_kernel_hard_cases(eng_mod, monkeypatch, scene, kernel):
    """Inputs chosen against the first kernels' layout limits: cells hit by more rays than an 8-bit field may hold (the
    particle is handed to the window kernel), fans that straddle four tiles, the irregular stretch of the reference's
    index formula on the negative side, many flagged cells and many passes over them.  Cell-exact against the oracle."""
    from thesis_amd.datasets import synthetic
    select_map_kernel(monkeypatch, kernel)
    rng = np.random.Generator(np.random.PCG64(77))
    B = 1081
    ang = synthetic.beam_angles(B)
    if scene == "near_wall":            # 0.3 m in front of a wall: the wall cells collect dozens of hits each
        poses = np.array([[7.7, 0.0, 0.0], [7.72, 0.3, 0.05], [-7.7, -0.2, 3.1]])
        scans = [synthetic.cast_scan(poses[0], ang, rng) for _ in range(2)]
    elif scene == "one_direction":      # every beam along the same line: 1081 hits on every cell of it
        poses = np.array([[0.3, 0.2, 0.4], [-3.0, 1.0, -2.0], [1.0, 1.0, 1.57]])
        scans = [np.full(B, 6.0), np.full(B, 2.5)]
        ang = np.full(B, 0.3)
    elif scene == "tile_corner":        # 40 m tiles meet at (20, 20): the fan covers four tiles
        poses = np.array([[19.9, 19.8, 0.7], [19.5, 19.9, -2.4], [19.99, -19.9, 1.0], [-19.9, 19.97, 0.0]])
        scans = [5.0 + 2.5 * np.sin(4 * ang), 4.0 + 3.0 * np.cos(7 * ang) + rng.normal(0, 0.01, B)]   # smooth walls: few events per cell
    elif scene == "negative_side":      # index-map irregularities all over the fan
        poses = np.array([[-12.3, -15.1, 0.3], [-19.7, -3.3, 2.0], [-9.7, -9.9, -1.0], [-15.0, -19.5, 0.9]])
        scans = [4.5 + 3.0 * np.sin(3 * ang), 3.0 + 2.0 * np.cos(5 * ang) + rng.normal(0, 0.01, B), rng.uniform(0.2, 7.0, B)]
    else:                               # short_rays: every beam ends within 1.5 m, ~2 flagged cells per beam
        poses = np.array([[0.0, 0.0, 0.0], [2.0, -2.0, 1.0], [-2.0, 3.0, -1.0]])
        scans = [rng.uniform(0.3, 1.5, B), rng.uniform(0.05, 0.8, B)]
    P = len(poses)
    e = eng_mod.ParticleEngine(P, max_beams=B, pool_tiles=64)
    maps = [orc.OracleHybridMap(0.05) for _ in range(P)]
    for r in scans:
        e.set_scan(r, ang)
        e.map_update(poses)
        sx, sy = orc.scan_xy(r, ang)
        for p in range(P):
            maps[p].update(tuple(float(v) for v in poses[p]), sx, sy)
    for p in range(P):
        assert_tiles_equal(e, p, oracle_dump(maps[p]), e.dim)
    c = e.counters()
    assert c["ray_cells_visited"] == sum(m.cells_visited for m in maps)
    if scene == "near_wall":
        assert c["window_fallbacks"] == 0                   # dozens of events per wall cell, thousands of passes over flagged cells: still the first kernel
    if scene == "one_direction":
        assert c["window_fallbacks"] == P * len(scans)      # the 8-bit guard must have fired for every particle
    if scene == "tile_corner":          # the four-tile fan stays in the first kernel
        assert c["window_fallbacks"] == 0, "fallback reasons %x" % c["fallback_reasons"]
    if scene == "negative_side":        # only the third scan (independent random ranges: cells with dozens of events) may fall back
        assert c["window_fallbacks"] <= P, "fallback reasons %x" % c["fallback_reasons"]
    e.close()


@pytest.mark.parametrize("fixture", ["R1_nearby_cell_diagonal", "R2_nearby_cell_diagonal_tile_edge"])
def test_map_update_nearby_cell_diagonal_to_the_start_cell(golden, eng_mod, map_kernel, fixture):
    """Regression (found by tools/fuzz_map_update.py, seed 51): a 0.11 m beam whose cell before the end lies at (-1, -1) from
    the start cell - the packed relative cell 0xFFFF | 0xFFFF << 16 looked like "no cell" to the global-index kernel and
    the beam's NEARBY event (hybridmap.py:139-142) was lost.  R2: the same next to a tile edge, second scan of two."""
    d = golden(fixture)
    cs, ang = float(d["cs"]), d["ang"]
    scans = [d["ranges"]] if d["ranges"].ndim == 1 else list(d["ranges"])
    poses = [d["pose"]] if "pose" in d else list(d["poses"])
    e = eng_mod.ParticleEngine(1, max_beams=len(ang), cell_size=cs, pool_tiles=40)
    hm = orc.OracleHybridMap(cs)
    for r, pose in zip(scans, poses):
        e.set_scan(r, ang)
        e.map_update(np.asarray(pose, dtype=np.float64)[None, :])
        sx, sy = orc.scan_xy(r, ang)
        hm.update(tuple(float(v) for v in pose), sx, sy)
    assert_tiles_equal(e, 0, oracle_dump(hm), e.dim)
    e.close()


def test_map_update_ray_cell_count_matches_oracle(eng_mod, map_kernel):
    from thesis_amd.datasets import synthetic
    B = 1081
    ang = synthetic.beam_angles(B)
    r = synthetic.cast_scan((0.2, 0.1, 0.3), ang, np.random.Generator(np.random.PCG64(1)))
    hm = orc.OracleHybridMap(0.05)
    sx, sy = orc.scan_xy(r, ang)
    hm.update((0.2, 0.1, 0.3), sx, sy)
    P = 4
    e = eng_mod.ParticleEngine(P, max_beams=B)
    e.set_scan(r, ang)
    e.map_update(np.broadcast_to([0.2, 0.1, 0.3], (P, 3)))
    c = e.counters()
    assert c["ray_cells_visited"] == P * hm.cells_visited
    n_written = sum(int(np.count_nonzero(t.map)) for t in hm.tiles)   # first update from zero: every written cell is non-zero
    assert c["cells_written"] == P * n_written
    e.close()


@pytest.mark.parametrize("model,mid", [("unicycle", "unicycle"), ("velocity_fr101", "velocity"),
                                       ("velocity_intelraw", "velocity"), ("absolute", "absolute")])
def test_imu_update_golden(golden, eng_mod, model, mid):
    g = golden("G7_imu_update")
    kw = {}
    if model == "velocity_intelraw":
        kw["vel_noise"] = (0.002, 0.05, 0.01, 0.05)      # IntelRawIMUData.py:51-55
    e = eng_mod.ParticleEngine(5, **kw)
    for d, dt, p_ref, c_ref in zip(g[model + "_data"], g[model + "_dt"], g[model + "_poses"], g[model + "_covs"]):
        e.imu_update(mid, d, float(dt))
    poses, covs = e.poses(), e.covs()
    for p in range(5):
        np.testing.assert_allclose(poses[p], g[model + "_poses"][-1], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(covs[p], g[model + "_covs"][-1], rtol=1e-10, atol=1e-18)
    e.close()


def test_map_update_full_size_properties(eng_mod, map_kernel):
    """BASELINE config 2 size (P=1024, B=1081): identical inputs -> identical maps; every tile equals the
    oracle's for that pose; repeating the scan saturates but never leaves [-30, 30]."""
    from thesis_amd.datasets import synthetic
    P, B = 1024, 1081
    ang = synthetic.beam_angles(B)
    r = synthetic.cast_scan((0.0, 0.0, 0.0), ang, np.random.Generator(np.random.PCG64(9)))
    e = eng_mod.ParticleEngine(P, max_beams=B)
    e.set_scan(r, ang)
    for _ in range(3):
        e.map_update(np.zeros((P, 3)))
    hm = orc.OracleHybridMap(0.05)
    sx, sy = orc.scan_xy(r, ang)
    for _ in range(3):
        hm.update((0.0, 0.0, 0.0), sx, sy)
    dump = oracle_dump(hm)
    for p in (0, 1, 511, 1023):
        assert_tiles_equal(e, p, dump, e.dim)
    for _ in range(12):
        e.map_update(np.zeros((P, 3)))
    (_, cells), = e.tiles(777)
    assert cells.min() == -30 and cells.max() == 30
    e.close()


# ---------------------------------------------------------------------------------------------------
# Robot.map_update through rbpf_scan_update with the engine seam doubled (robot.py:59-115)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["s", "l"])
def test_scan_update_golden(golden, eng_mod, case):
    g = golden("G6_map_update_G9_match_inputs")
    P = 3
    e = eng_mod.ParticleEngine(P, max_beams=1081, pool_tiles=16)
    for p in range(P):
        load_dump_into(e, p, golden_dump_as_dict(g, case + "_pre_"), e.dim)
    e.set_state(poses=[0.1, 0.05, 0.02], covs=g[case + "_cov_in"], weights=1.0)
    e.set_scan(g[case + "_ranges1"], g[case + "_angles"])
    match = np.concatenate([g[case + "_scan_pose"], g[case + "_scan_cov"].ravel(), [321.0]])
    e.scan_update(match_override=np.broadcast_to(match, (P, 13)),
                  guesses=np.broadcast_to(g[case + "_guesses"], (P, 30, 3)))
    poses, covs, w = e.poses(), e.covs(), e.weights()
    for p in range(P):
        # north-star tolerance: 1e-4 relative
        np.testing.assert_allclose(poses[p], g[case + "_pose_out"], rtol=1e-7, atol=1e-12)
        np.testing.assert_allclose(covs[p], g[case + "_cov_out"], rtol=1e-5, atol=1e-14)
        np.testing.assert_allclose(w[p], g[case + "_weight_out"][-1], rtol=1e-9)
        assert_tiles_equal(e, p, golden_dump_as_dict(g, case + "_post_"), e.dim)
    e.close()


def test_scan_update_nan_cov_branch_golden(golden, eng_mod):
    g = golden("G6_map_update_G9_match_inputs")
    P = 2
    e = eng_mod.ParticleEngine(P, max_beams=1081)
    e.set_scan(g["nan_ranges"], g["nan_angles"])
    e.map_update(np.zeros((P, 3)))
    e.set_state(poses=g["nan_pose_in"], weights=1.0)
    match = np.concatenate([[0, 0, 0], np.full(9, np.nan), [0.0]])
    e.scan_update(match_override=np.broadcast_to(match, (P, 13)))
    np.testing.assert_allclose(e.weights(), g["nan_weight_out"][-1], rtol=1e-12)
    np.testing.assert_array_equal(e.poses(), np.broadcast_to(g["nan_pose_in"], (P, 3)))   # no pose appended
    for p in range(P):
        assert_tiles_equal(e, p, golden_dump_as_dict(g, "nan_post_"), e.dim)
    e.close()


def test_scan_update_device_sampling_statistics(eng_mod):
    """Philox proposal (no explicit guesses): the weighted mean stays within a few sigma of the matcher
    pose and the covariance has the matcher covariance's scale."""
    from thesis_amd.datasets import synthetic
    P, B = 64, 361
    ang = synthetic.beam_angles(B, np.pi)
    r = synthetic.cast_scan((0, 0, 0), ang, None)
    e = eng_mod.ParticleEngine(P, max_beams=B)
    e.set_scan(r, ang)
    e.map_update(np.zeros((P, 3)))
    cov = np.diag([1e-4, 1e-4, 1e-5])
    match = np.concatenate([[0.01, -0.02, 0.003], cov.ravel(), [100.0]])
    e.scan_update(match_override=np.broadcast_to(match, (P, 13)))
    poses, covs = e.poses(), e.covs()
    assert np.all(np.abs(poses - [0.01, -0.02, 0.003]) < [0.05, 0.05, 0.02])
    assert len({tuple(np.round(p, 12)) for p in poses}) == P        # independent streams per particle
    d = np.array([np.diag(c) for c in covs])
    assert np.all(d > 0) and np.all(d < [1e-3, 1e-3, 1e-4])
    e.close()


# ---------------------------------------------------------------------------------------------------
# resample (main.py:46-79)
# ---------------------------------------------------------------------------------------------------
def test_resample_indices_golden(golden, eng_mod):
    g = golden("G8_resample")
    for k in range(3):
        e = eng_mod.ParticleEngine(8)
        e.set_state(weights=g["known_w"])
        did, idx = e.resample(float(g["known%d_u" % k]))
        assert did and np.array_equal(idx, g["known%d_idx" % k])
        assert np.all(e.weights() == 1.0)                            # main.py:77-78
        e.close()
    e = eng_mod.ParticleEngine(4)
    e.set_state(weights=g["nores_w"])
    did, idx = e.resample(0.5)
    assert not did and np.array_equal(idx, g["nores_idx"])
    np.testing.assert_array_equal(e.weights(), g["nores_w"])         # spread <= 200: untouched
    e.close()
    for P in (64, 1024, 16384):
        e = eng_mod.ParticleEngine(P, max_beams=8)
        for v in range(3):
            e.set_state(weights=g["r%d_%d_w" % (P, v)])
            did, idx = e.resample(float(g["r%d_%d_u" % (P, v)]))
            assert did
            assert np.array_equal(idx, g["r%d_%d_idx" % (P, v)]), (P, v)   # bit-exact ancestor indices
        e.close()


def test_resample_moves_maps_and_state(eng_mod):
    """Maps, poses and covariances follow their ancestors (Robot.copy, robot.py:141-149), across two
    rounds (slot indirection), with tiles allocated and released as the ancestors' tile sets differ."""
    P = 8
    e = eng_mod.ParticleEngine(P, max_beams=8, pool_tiles=24)
    dim = e.dim
    rng = np.random.Generator(np.random.PCG64(5))
    marks = {}
    for p in range(P):
        cells = np.zeros((dim, dim), dtype=np.int8)
        cells[100 + p, 200:260] = p + 1
        cells[300:340, 50 + p] = -(p + 1)
        e.set_tile(p, (0, 0), cells)
        marks[p] = {(0.0, 0.0): cells}
    extra3 = rng.integers(-30, 31, size=(dim, dim)).astype(np.int8)
    extra0 = rng.integers(-30, 31, size=(dim, dim)).astype(np.int8)
    e.set_tile(3, (40, 0), extra3); marks[3][(40.0, 0.0)] = extra3
    e.set_tile(0, (0, 40), extra0); marks[0][(0.0, 40.0)] = extra0
    e.set_tile(5, (-40, -40), extra0); marks[5][(-40.0, -40.0)] = extra0     # particle 5 dies: tile released
    poses = rng.normal(size=(P, 3)); covs = rng.normal(size=(P, 3, 3))
    e.set_state(poses=poses, covs=covs, weights=[10, -250, -100, 300, 5, -np.inf, 0, 42])
    assert e.counters()["tiles_in_use"] == P + 3
    did, idx = e.resample(0.25)
    assert did and idx.tolist() == [0, 0, 3, 3, 3, 4, 4, 7]

    def check(idx_chain):
        np.testing.assert_array_equal(e.poses(), poses[idx_chain])
        np.testing.assert_array_equal(e.covs(), covs[idx_chain])
        total = 0
        for j, anc in enumerate(idx_chain):
            got = dict(e.tiles(j))
            assert set(got) == set(marks[anc]), (j, anc)
            for c in got:
                assert np.array_equal(got[c], marks[anc][c]), (j, anc, c)
            total += len(got)
        assert e.counters()["tiles_in_use"] == total
    check(idx)
    # second round on top of the permuted slots
    e.set_state(weights=[0.0, 500.0, 1.0, 2.0, 900.0, 3.0, 4.0, 0.5])
    did, idx2 = e.resample(0.6)
    assert did
    check(idx[idx2])
    # a map update still lands in the right particle's map after two permutations
    from thesis_amd.datasets import synthetic
    ang = synthetic.beam_angles(8, np.pi)
    e.set_scan(np.full(8, 2.0), ang)
    before = dict(e.tiles(5))
    e.map_update(np.zeros((P, 3)))
    after = dict(e.tiles(5))
    assert not np.array_equal(before[(0.0, 0.0)], after[(0.0, 0.0)])
    e.close()


# ---------------------------------------------------------------------------------------------------
# a6: matcher input lists (hybridmap.py:210-242) against G9
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["s", "l"])
def test_match_inputs_golden(golden, eng_mod, case):
    g = golden("G6_map_update_G9_match_inputs")
    e = eng_mod.ParticleEngine(2, max_beams=1081, pool_tiles=8)
    dump = golden_dump_as_dict(g, case + "_pre_")
    load_dump_into(e, 1, dump, e.dim)
    e.set_scan(g[case + "_ranges1"], g[case + "_angles"])
    curr, ref = e.match_inputs(1, [0.1, 0.05, 0.02])
    assert np.array_equal(curr, g[case + "_m_curr"])                   # same points, same order, bit for bit
    # the reference's `> 1.0` sees float64 rounding noise on cells that are exactly 1.0 on the lattice (DESIGN.md
    # section 2): compare against the golden list without the cells whose reference value is within 1e-9 of 1.0
    want = {tuple(p) for p in g[case + "_m_ref"]}
    got = {tuple(p) for p in ref}
    noisy = set()
    for (cx, cy), (xs, ys, vals) in dump.items():
        for x, y, val in zip(xs, ys, vals):
            if abs(val - 1.0) < 1e-9:
                noisy.add(((float(x) - e.dim / 2) * 40 / e.dim + cx - 0.1, (float(y) - e.dim / 2) * 40 / e.dim + cy - 0.05))
    assert got - want == set() or all(p in noisy for p in got - want)
    assert all(p in noisy for p in want - got)
    order = [tuple(p) for p in ref]
    assert order == sorted(order)                                       # np.unique order: by x, then y
    e.close()


def test_full_size_step_properties_c3(eng_mod):
    """BASELINE config 3 size (4096 particles, 1081 beams, 0.05 m): one full step + resample, size-independent
    properties: finite state, lattice range, non-decreasing ancestors covering [0, P), tile accounting, and the
    per-particle ray-cell / written-cell counters equal to the oracle's for a particle at the same pose."""
    from thesis_amd.datasets import synthetic
    P, B = 4096, 1081
    ang = synthetic.beam_angles(B)
    rng = np.random.Generator(np.random.PCG64(2))
    r0 = synthetic.cast_scan((0.0, 0.0, 0.0), ang, rng)
    e = eng_mod.ParticleEngine(P, max_beams=B, pool_tiles=2 * P)
    e.set_scan(r0, ang)
    e.map_update(np.zeros((P, 3)))
    c = e.counters()
    hm = orc.OracleHybridMap(0.05)
    sx, sy = orc.scan_xy(r0, ang)
    hm.update((0.0, 0.0, 0.0), sx, sy)
    assert c["ray_cells_visited"] == P * hm.cells_visited
    assert c["cells_written"] == P * int(np.count_nonzero(hm.tiles[0].map))
    e.imu_update("velocity", [0.5, 0.0, 0.1], 7000.0)
    r1 = synthetic.cast_scan((0.35, 0.0, 0.07), ang, rng)
    e.set_scan(r1, ang)
    e.scan_update(adj=False)
    did, idx = e.resample(0.4242)
    poses, w = e.poses(), e.weights()
    assert np.all(np.isfinite(poses)) and np.all(np.isfinite(e.covs()))
    assert np.all(np.abs(poses - [0.35, 0.0, 0.07]) < [0.2, 0.2, 0.05])
    assert np.all(np.diff(idx) >= 0) and idx[0] >= 0 and idx[-1] < P
    if did:
        assert np.all(w == 1.0)
    assert e.counters()["tiles_in_use"] == P
    for p in (0, P // 2, P - 1):
        (centre, cells), = e.tiles(p)
        assert centre == (0.0, 0.0) and cells.min() >= -30 and cells.max() <= 30 and np.count_nonzero(cells) > 40000
    e.close()


@pytest.mark.parametrize("B", [1, 2, 4095])
def test_map_update_extreme_beam_counts(eng_mod, B):
    """The smallest scans and the largest the ABI admits (max_beams <= 4095; more rays than either map kernel has
    threads, the whole-fan kernel's tables do not hold them and the window kernel takes over), cell-exact.
    The poses sit off the lattice lines on purpose: with the sensor exactly on a cell boundary and a beam that exactly
    cancels the heading, the end point lands on the boundary up to the last ulp of sin/cos, where the device's and
    numpy's libm may round to different cells (parity notes)."""
    rng = np.random.Generator(np.random.PCG64(B))
    ang = np.linspace(-2.3, 2.3, B) if B > 1 else np.array([0.4])
    r = rng.uniform(0.5, 9.0, B)
    poses = np.array([[0.313, -0.227, 0.5], [-4.011, 2.519, -1.0]])
    e = eng_mod.ParticleEngine(2, max_beams=4095, pool_tiles=16)
    maps = [orc.OracleHybridMap(0.05) for _ in range(2)]
    for k in range(2):
        e.set_scan(r * (1.0 + 0.1 * k), ang)
        e.map_update(poses)
        sx, sy = orc.scan_xy(r * (1.0 + 0.1 * k), ang)
        for p in range(2):
            maps[p].update(tuple(float(v) for v in poses[p]), sx, sy)
    for p in range(2):
        assert_tiles_equal(e, p, oracle_dump(maps[p]), e.dim)
    assert e.counters()["ray_cells_visited"] == sum(m.cells_visited for m in maps)
    e.close()


@pytest.mark.parametrize("B,cs", [(1081, 0.05), (1200, 0.015625)])
def test_map_update_many_long_beams_in_strips(eng_mod, map_kernel, B, cs):
    """Beams of 14.9 m all round: a 600-cell fan at 0.05 m and a 1900-cell one at 1/64 m - several strips of the
    first kernel (at 1/64 m the 1200 rays have more than 65 535 whole 16-step chunks: the global-index kernel's strips then walk
    level by level instead of through 16-bit item prefixes).  Second scan: 11 m.  Cell-exact against the C oracle; no particle reaches
    the window kernel by default."""
    from oracle import c_oracle
    ang = np.linspace(-np.pi, np.pi, B, endpoint=False)
    rng = np.random.Generator(np.random.PCG64(123))
    pose = np.array([[0.313, -0.227, 0.5]])
    e = eng_mod.ParticleEngine(1, max_beams=B, cell_size=cs, pool_tiles=12)
    lib = c_oracle.load()
    m = c_oracle.CMap(lib, cs)
    for r in (14.9 + rng.normal(0, 0.01, B), 11.0 + 0.5 * np.sin(7 * ang)):
        e.set_scan(r, ang)
        e.map_update(pose)
        sx, sy = orc.scan_xy(r, ang)
        m.update(pose[0], sx, sy)
    want = {c: np.rint(t / Q).astype(np.int8) for c, t in m.tiles().items()}
    got = dict(e.tiles(0))
    assert set(got) == set(want)
    for c in want:
        assert np.array_equal(got[c], want[c]), f"tile {c}: {int(np.count_nonzero(got[c] != want[c]))} cells differ"
    c = e.counters()
    assert c["ray_cells_visited"] == lib.orc_map_cells_visited(m.h)
    if map_kernel == "auto":
        assert c["window_fallbacks"] == 0 and c["map_windows"] >= 2 * 2           # at least two strips per scan
    e.close()


@pytest.mark.xfail(strict=True, reason="device sincos vs libm: the last bit decides the cell of an end point that lies exactly on a cell boundary")
def test_map_update_on_a_lattice_line_known_deviation(eng_mod):
    """The documented carve-out of the map-update parity (parity notes), with the inputs that first showed it
    (round 1, test_map_update_extreme_beam_counts[4095] before its poses were moved): sensor exactly on a cell boundary
    (y = -0.2 = -4 cells), heading 0.5, and beam 1602 of linspace(-2.3, 2.3, 4095) at -0.5, so that the end point's y is
    -0.2 up to the last bit of sin and cos.  numpy's libm and the device's sincos round that bit differently and the ray
    runs along adjacent rows (103 cells of row 396: -30 here, -27 in the reference).  A strict expected failure: if the
    device ever agrees with libm here, this test says so."""
    B = 4095
    rng = np.random.Generator(np.random.PCG64(B))
    ang = np.linspace(-2.3, 2.3, B)
    r = rng.uniform(0.5, 9.0, B)
    poses = np.array([[0.3, -0.2, 0.5], [-4.0, 2.5, -1.0]])
    e = eng_mod.ParticleEngine(2, max_beams=4095, pool_tiles=16)
    maps = [orc.OracleHybridMap(0.05) for _ in range(2)]
    try:
        for k in range(2):
            e.set_scan(r * (1.0 + 0.1 * k), ang)
            e.map_update(poses)
            sx, sy = orc.scan_xy(r * (1.0 + 0.1 * k), ang)
            for p in range(2):
                maps[p].update(tuple(float(v) for v in poses[p]), sx, sy)
        for p in range(2):
            assert_tiles_equal(e, p, oracle_dump(maps[p]), e.dim)
    finally:
        e.close()


@pytest.mark.parametrize("cs,B,kernel", [(0.05, 181, "auto"), (0.1, 180, "window"), (0.05, 721, "ray"), (0.025, 181, "ray"), (0.1, 180, "auto"), (0.05, 361, "auto")])
def test_closed_loop_population_equals_oracle(eng_mod, monkeypatch, cs, B, kernel):
    """(Cell sizes of configs C1, C2 and C5, both map-update kernels.)  The whole per-scan cycle of main.py:138-214 over several scans, engine against a population of OracleRobot:
    IMU propagation, Robot.map_update with the engine seam doubled (matcher result and proposal samples injected on both
    sides; one particle takes the NaN-covariance branch once), the map update, and resampling with its deep copies
    whenever the spread trigger fires.  Ancestors must agree exactly at every step, state within the north-star
    tolerance, and every particle's map cell for cell at the end."""
    from thesis_amd.datasets import synthetic
    select_map_kernel(monkeypatch, kernel)
    P, K, T = 6, 30, 9
    ang = synthetic.beam_angles(B, np.pi if B < 400 else 1.5 * np.pi)
    rng = np.random.Generator(np.random.PCG64(2024))
    truth = np.array([0.2, -0.1, 0.05])
    e = eng_mod.ParticleEngine(P, max_beams=B, pool_tiles=8 * P, cell_size=cs)
    robots = [orc.OracleRobot(cs) for _ in range(P)]
    r0 = synthetic.cast_scan(truth, ang, rng)
    sx, sy = orc.scan_xy(r0, ang)
    e.set_scan(r0, ang)
    e.map_update(np.zeros((P, 3)))                               # update_count < 2 branch of main.py:155
    for rb in robots:
        rb.map.update((0.0, 0.0, 0.0), sx, sy)
    resamples = 0
    for t in range(T):
        vel = np.array([0.4, 0.1, 0.08]) * (1 + 0.1 * rng.normal(size=3))
        for _ in range(3):                                       # a few IMU readings per scan
            e.imu_update("velocity", vel, 300.0)
            for rb in robots:
                rb.imu_update("velocity_fr101", vel, 300.0)
        truth = truth + np.array([0.036, 0.009, 0.0072])
        r = synthetic.cast_scan(truth, ang, rng)
        sx, sy = orc.scan_xy(r, ang)
        e.set_scan(r, ang)
        poses_in = e.poses()
        np.testing.assert_allclose(poses_in, [rb.pose() for rb in robots], rtol=1e-9, atol=1e-12)
        match = np.zeros((P, 13))
        guesses = np.zeros((P, K, 3))
        for p in range(P):
            cov = np.diag([2e-4, 3e-4, 2e-5]) * (1 + 0.3 * p)
            cov[0, 1] = cov[1, 0] = 5e-5
            mp = poses_in[p] + rng.normal(0, [0.01, 0.01, 0.003])
            match[p] = np.concatenate([mp, cov.ravel(), [100.0 + p]])
            guesses[p] = rng.multivariate_normal(mp, cov, K)
        if t == 3:
            match[2, 3:12] = np.nan                              # robot.py:73-78
        e.scan_update(match_override=match, guesses=guesses)
        for p, rb in enumerate(robots):
            rb.map_update(sx, sy, (match[p, :3], match[p, 3:12].reshape(3, 3), match[p, 12]), guesses=guesses[p])
        w = e.weights()
        np.testing.assert_allclose(w, [float(rb.weight[-1]) for rb in robots], rtol=1e-9)
        np.testing.assert_allclose(e.poses(), [rb.pose() for rb in robots], rtol=1e-7, atol=1e-10)
        np.testing.assert_allclose(e.covs(), [np.asarray(rb.cov, dtype=np.float64) for rb in robots], rtol=1e-5, atol=1e-13)
        if t in (2, 5, 7):                                       # push the spread past the trigger of main.py:50
            bump = np.zeros(P); bump[(t * 2) % P] = 260.0; bump[(t + 3) % P] = -40.0
            e.set_state(weights=w + bump)
            for p, rb in enumerate(robots):
                rb.weight[-1] = rb.weight[-1] + bump[p]
        u = float(rng.random())
        did, idx = e.resample(u)
        did_ref, idx_ref = orc.resample_indices([rb.weight[-1] for rb in robots], u)
        assert did == did_ref and (not did or list(idx) == list(idx_ref)), (t, idx, idx_ref)
        if did:
            robots = orc.resample(robots, u)
            resamples += 1
    assert resamples >= 3
    for p, rb in enumerate(robots):
        assert_tiles_equal(e, p, oracle_dump(rb.map), e.dim)
    e.close()

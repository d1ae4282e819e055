"""Pin the CPU oracle (oracle/rbpf_oracle.py) against golden vectors captured from
the imported reference (tests/golden/gen_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import rbpf_oracle as orc
from tests.helpers import oracle_map_from_dump, dump_oracle_map, golden_dump_as_dict


def test_g1_affected_points(golden):
    g = golden("G1_affected_points")
    for k, a in enumerate(g["args"]):
        pts = orc.get_affected_points(*[int(v) for v in a])
        ref = g["pts"][g["offs"][k]:g["offs"][k + 1]]
        assert len(pts) == len(ref), a
        if len(pts):
            assert np.array_equal(np.array(pts), ref), a


def test_g2_index_formulas(golden):
    g = golden("G2_index_math")
    n_mismatch_set_vs_get = 0
    for cs in (0.1, 0.05, 0.025):
        dim = round(40 / cs)
        for centre in (-40, 0, 40):
            key = "cs%g_c%d" % (cs, centre)
            tile = orc.OracleTile(centre, 0, 40, cs)
            for i, s_ref, g_ref, in_ref in zip(g[key + "_gi"], g[key + "_set"], g[key + "_get"], g[key + "_in"]):
                pos = int(i) * cs
                assert tile.is_in_map(pos, 0.0) == bool(in_ref)
                rel = pos - centre
                c = tile.get_cell(rel, 0.0)
                assert (c[0] if c is not None else -9999) == g_ref
                if in_ref:
                    assert orc.set_index(rel, cs, dim) == s_ref
                    n_mismatch_set_vs_get += int(s_ref != g_ref)
    assert n_mismatch_set_vs_get > 0  # the two formulas really differ
    for x, c in zip(g["centre_in"], g["centre_out"]):
        assert orc.map_centre_1d(float(x), 40) == c[0]
        assert orc.map_centre_1d(float(-x), 40) == c[1]


@pytest.mark.parametrize("case", list("abcdefg"))
def test_g3_map_update(golden, case):
    g = golden("G3_map_update")
    cs = float(g[case + "_cs"])
    hm = orc.OracleHybridMap(cs)
    for p, r in zip(g[case + "_poses"], g[case + "_ranges"]):
        sx, sy = orc.scan_xy(r, g[case + "_angles"])
        hm.update((float(p[0]), float(p[1]), float(p[2])), sx, sy)
    mine = dump_oracle_map(hm)
    ref = golden_dump_as_dict(g, case + "_")
    assert list(mine.keys()) == list(ref.keys())  # same tiles, same creation order
    for k in ref:
        assert np.array_equal(mine[k][0], ref[k][0]) and np.array_equal(mine[k][1], ref[k][1]), k
        assert np.array_equal(mine[k][2], ref[k][2]), k  # bit-identical float64 log-odds


@pytest.mark.parametrize("case,cs", [("b", 0.05), ("d", 0.05)])
def test_g4_get_odds_at(golden, case, cs):
    g3, g4 = golden("G3_map_update"), golden("G4_get_odds_at")
    hm = oracle_map_from_dump(g3, case + "_", cs)
    for (x, y), v, none in zip(g4[case + "_pts"], g4[case + "_vals"], g4[case + "_none"]):
        o = hm.get_odds_at(float(x), float(y))
        assert (o is None) == bool(none)
        if o is not None:
            assert o == v


@pytest.mark.parametrize("case", ["a", "b"])
def test_g5_sample_weight(golden, case):
    g3, g5 = golden("G3_map_update"), golden("G5_sample_weight")
    hm = oracle_map_from_dump(g3, case + "_", 0.05)
    sx, sy = orc.scan_xy(g5[case + "_ranges"], g5[case + "_angles"])
    w = orc.generate_sample_weight(hm, g5[case + "_guesses"], sx, sy, g5[case + "_prs"])
    ref = g5[case + "_w"].astype(np.longdouble) + g5[case + "_w_hi"].astype(np.longdouble)
    assert np.array_equal(w, ref)  # longdouble accumulators reproduced exactly


@pytest.mark.parametrize("case", ["s", "l"])
def test_g6_map_update_and_g9_match_inputs(golden, case):
    g = golden("G6_map_update_G9_match_inputs")
    rb = orc.OracleRobot(0.05)
    rb.map = oracle_map_from_dump(g, case + "_pre_", 0.05)
    rb.x, rb.y, rb.theta = [0.0, 0.1], [0.0, 0.05], [0.0, 0.02]
    rb.cov = g[case + "_cov_in"].astype(np.longdouble)
    sx, sy = orc.scan_xy(g[case + "_ranges1"], g[case + "_angles"])

    # G9: matcher inputs built from the map (hybridmap.py:210-251)
    pr = orc.pose_range_from_cov(rb.cov)
    curr, ref, guess0, cpm, prange = rb.map.scan_match_inputs(sx, sy, rb.pose(), pr)
    assert np.array_equal(np.array(curr).reshape(-1, 2), g[case + "_m_curr"])
    assert np.array_equal(np.array(ref).reshape(-1, 2), g[case + "_m_ref"])
    assert cpm == int(g[case + "_m_cpm"])
    assert np.array_equal(np.array(prange, dtype=np.float64), g[case + "_m_range"])

    # scipy pdf restatement
    prs = orc.mvn_pdf(g[case + "_guesses"], g[case + "_scan_pose"], g[case + "_scan_cov"]) * 10
    np.testing.assert_allclose(prs, g[case + "_motion_prs"], rtol=1e-12)

    # G6: proposal, moments, weight, map update (robot.py:80-115)
    rb.map_update(sx, sy, (g[case + "_scan_pose"], g[case + "_scan_cov"], 321.0), guesses=g[case + "_guesses"])
    np.testing.assert_allclose(np.array(rb.pose(), dtype=np.float64), g[case + "_pose_out"], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(np.array(rb.cov, dtype=np.float64), g[case + "_cov_out"], rtol=1e-9, atol=1e-18)
    np.testing.assert_allclose(np.array(rb.weight, dtype=np.float64), g[case + "_weight_out"], rtol=1e-12)
    mine, refd = dump_oracle_map(rb.map), golden_dump_as_dict(g, case + "_post_")
    assert list(mine.keys()) == list(refd.keys())
    for k in refd:
        assert np.array_equal(mine[k][0], refd[k][0]) and np.array_equal(mine[k][1], refd[k][1])
        assert np.array_equal(mine[k][2], refd[k][2])

    # adj variant inputs (hybridmap.py:147-181)
    sx0, sy0 = orc.scan_xy(g[case + "_ranges0"], g[case + "_angles"])
    lgx, lgy = orc.transform(sx0, sy0, (0.1, 0.05, 0.02))
    curr, ref, *_ = rb.map.scan_adj_inputs(sx, sy, lgx, lgy, (0.12, 0.06, 0.03), np.array([0.3, 0.2, 0.5]))
    assert np.array_equal(np.array(curr).reshape(-1, 2), g[case + "_adj_curr"])
    assert np.array_equal(np.array(ref).reshape(-1, 2), g[case + "_adj_ref"])


def test_g6_nan_cov_branch(golden):
    g = golden("G6_map_update_G9_match_inputs")
    rb = orc.OracleRobot(0.05)
    sx, sy = orc.scan_xy(g["nan_ranges"], g["nan_angles"])
    rb.map.update((0, 0, 0), sx, sy)
    rb.x, rb.y, rb.theta = [0.0, 0.02], [0.0, -0.01], [0.0, 0.005]
    nan_cov = np.full((3, 3), np.nan)
    rb.map_update(sx, sy, ([0, 0, 0], nan_cov, 0.0))
    assert len(rb.x) == int(g["nan_npose"])  # no pose appended (robot.py:73-78)
    np.testing.assert_allclose(np.array(rb.weight, dtype=np.float64), g["nan_weight_out"], rtol=1e-14)
    mine, refd = dump_oracle_map(rb.map), golden_dump_as_dict(g, "nan_post_")
    for k in refd:
        assert np.array_equal(mine[k][2], refd[k][2])


@pytest.mark.parametrize("model", ["unicycle", "velocity_fr101", "velocity_intelraw", "absolute"])
def test_g7_imu_update(golden, model):
    g = golden("G7_imu_update")
    rb = orc.OracleRobot(0.05)
    for d, dt, p_ref, c_ref in zip(g[model + "_data"], g[model + "_dt"], g[model + "_poses"], g[model + "_covs"]):
        rb.imu_update(model, d, float(dt))
        assert np.array_equal(np.array(rb.pose(), dtype=np.float64), p_ref)
        assert np.array_equal(np.array(rb.cov, dtype=np.float64), c_ref)


def test_g8_resample(golden):
    g = golden("G8_resample")
    for k in range(3):
        did, idx = orc.resample_indices(list(g["known_w"]), float(g["known%d_u" % k]))
        assert did and np.array_equal(np.array(idx), g["known%d_idx" % k])
    # SURVEY quirk 7 known answers
    assert orc.resample_indices([10, -250, -100, 300, 5, -np.inf, 0, 42], 0.25)[1] == [0, 0, 3, 3, 3, 4, 4, 7]
    assert orc.resample_indices([10, -250, -100, 300, 5, -np.inf, 0, 42], 0.999)[1] == [0, 2, 3, 3, 3, 4, 7, 7]
    did, idx = orc.resample_indices(list(g["nores_w"]), 0.5)
    assert not did and np.array_equal(np.array(idx), g["nores_idx"])
    for P in (64, 1024, 16384):
        for v in range(3):
            w = [np.longdouble(x) for x in g["r%d_%d_w" % (P, v)]]
            did, idx = orc.resample_indices(w, float(g["r%d_%d_u" % (P, v)]))
            assert did
            assert np.array_equal(np.array(idx, dtype=np.int32), g["r%d_%d_idx" % (P, v)]), (P, v)

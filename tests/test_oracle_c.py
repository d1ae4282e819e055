"""Pin the C restatement (oracle/rbpf_oracle.c) against the golden vectors captured from the reference."""
import numpy as np
import pytest

from oracle import rbpf_oracle as orc
from oracle import c_oracle
from tests.helpers import golden_dump_as_dict


@pytest.fixture(scope="module")
def lib():
    return c_oracle.load()


def assert_same(tiles, dump, dim):
    assert set(tiles) == set(dump)
    for c, (xs, ys, vals) in dump.items():
        want = np.zeros((dim, dim)); want[xs, ys] = vals
        assert np.array_equal(tiles[c], want), c          # bit-identical float64


@pytest.mark.parametrize("case", list("abcdefg"))
def test_c_map_update_golden(golden, lib, case):
    g = golden("G3_map_update")
    m = c_oracle.CMap(lib, float(g[case + "_cs"]))
    for p, r in zip(g[case + "_poses"], g[case + "_ranges"]):
        sx, sy = orc.scan_xy(r, g[case + "_angles"])
        m.update(p, sx, sy, ld=False)
    assert_same(m.tiles(), golden_dump_as_dict(g, case + "_"), m.dim)


@pytest.mark.parametrize("case", ["a", "b"])
def test_c_sample_weight_golden(golden, lib, case):
    g3, g5 = golden("G3_map_update"), golden("G5_sample_weight")
    m = c_oracle.CMap(lib, 0.05)
    for c, (xs, ys, vals) in golden_dump_as_dict(g3, case + "_").items():
        cells = np.zeros((m.dim, m.dim)); cells[xs, ys] = vals
        m.set_tile(c[0], c[1], cells)
    sx, sy = orc.scan_xy(g5[case + "_ranges"], g5[case + "_angles"])
    w = m.sample_weight(g5[case + "_guesses"], sx, sy, g5[case + "_prs"])
    ref = g5[case + "_w"].astype(np.longdouble) + g5[case + "_w_hi"].astype(np.longdouble)
    assert np.array_equal(w, ref)


@pytest.mark.parametrize("case", ["s", "l"])
def test_c_robot_map_update_golden(golden, lib, case):
    g = golden("G6_map_update_G9_match_inputs")
    m = c_oracle.CMap(lib, 0.05)
    for c, (xs, ys, vals) in golden_dump_as_dict(g, case + "_pre_").items():
        cells = np.zeros((m.dim, m.dim)); cells[xs, ys] = vals
        m.set_tile(c[0], c[1], cells)
    sx, sy = orc.scan_xy(g[case + "_ranges1"], g[case + "_angles"])
    pose, cov, w = m.robot_map_update([0.1, 0.05, 0.02], g[case + "_cov_in"], 1.0, g[case + "_guesses"],
                                      g[case + "_motion_prs"], sx, sy)
    np.testing.assert_allclose(pose.astype(np.float64), g[case + "_pose_out"], rtol=1e-15, atol=1e-18)
    np.testing.assert_allclose(cov.astype(np.float64), g[case + "_cov_out"], rtol=1e-12, atol=1e-22)
    np.testing.assert_allclose(float(w), g[case + "_weight_out"][-1], rtol=1e-15)
    assert_same(m.tiles(), golden_dump_as_dict(g, case + "_post_"), m.dim)

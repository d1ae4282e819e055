"""The NDT refinement restatement (oracle/matcher_oracle.py; matchScanCustom.m:32-37, parity unpinned against MATLAB):
its analytic derivatives are the derivatives of its score, and the damped Newton ascent converges on a known pose."""
import numpy as np

from oracle import matcher_oracle as mo


def room(N=672, half=100, thick=2):
    occ = np.zeros((N, N), dtype=bool)
    c = N // 2
    for t in range(thick):
        occ[c - half - t, c - half:c + half + 1] = True
        occ[c + half + t, c - half:c + half + 1] = True
        occ[c - half:c + half + 1, c - half - t] = True
        occ[c - half:c + half + 1, c + half + t] = True
    ang = np.linspace(-np.pi, np.pi, 720, endpoint=False) + 0.0123
    d = half + 0.5
    r = d / np.maximum(np.abs(np.cos(ang)), np.abs(np.sin(ang)))
    return occ, np.stack([r * np.cos(ang), r * np.sin(ang)], 1), c + 0.5


def test_gradient_and_hessian_are_derivatives_of_the_score():
    occ, pts, c = room()
    p = np.array([c + 0.31, c - 0.22, 0.003])
    m = mo.ndt_eval(occ, pts, p, 2, 17, -5)
    H = np.array([[m[4], m[5], m[6]], [m[5], m[7], m[8]], [m[6], m[8], m[9]]])
    for k, h in enumerate((1e-6, 1e-6, 1e-8)):
        e = np.zeros(3); e[k] = h
        up, dn = mo.ndt_eval(occ, pts, p + e, 2, 17, -5), mo.ndt_eval(occ, pts, p - e, 2, 17, -5)
        assert abs((up[0] - dn[0]) / (2 * h) - m[1 + k]) < 1e-5 * max(1.0, abs(m[1 + k]))
        assert np.allclose((up[1:4] - dn[1:4]) / (2 * h), H[k], rtol=1e-4, atol=1e-3 * np.abs(H).max())


def test_cells_with_fewer_than_three_points_carry_no_gaussian():
    occ = np.zeros((64, 64), dtype=bool)
    occ[10, 10] = occ[10, 11] = True                     # two points in every NDT cell that holds them
    pts = np.array([[0.0, 0.0]])
    assert mo.ndt_eval(occ, pts, (10.5, 10.5, 0.0), 2, 0, 0)[0] == 0.0
    occ[11, 10] = True                                   # three points: the grids that hold all three respond
    assert mo.ndt_eval(occ, pts, (10.5, 10.5, 0.0), 2, 0, 0)[0] < 0.0


def test_collinear_points_get_the_eigenvalue_floor():
    occ = np.zeros((64, 64), dtype=bool)
    occ[8, 8:12] = True                                  # four collinear points inside one 4x4 NDT cell
    ok, qx, qy, B00, B01, B11 = mo._cell_stats(occ, np.array([8]), np.array([8]), 4)
    assert ok[0] and np.isfinite([B00[0], B01[0], B11[0]]).all()
    cov = np.linalg.inv(np.array([[B00[0], B01[0]], [B01[0], B11[0]]]))
    ev = np.linalg.eigvalsh(cov)
    assert np.isclose(ev[0] / ev[1], mo.NDT_EIG_FLOOR, rtol=1e-9)


def test_newton_ascent_recovers_the_pose():
    occ, pts, c = room()
    for start in ((c + 0.6, c - 0.4, 0.004), (c - 0.7, c + 0.7, -0.003), (c, c, 0.0)):
        p, score, evals = mo.ndt_refine(occ, pts, start, 2, 0, 0)
        assert abs(p[0] - c) < 0.05 and abs(p[1] - c) < 0.05 and abs(p[2]) < 1e-3
        assert evals < 60 and score > 0.9 * -mo.ndt_eval(occ, pts, (c, c, 0.0), 2, 0, 0, True)[0]

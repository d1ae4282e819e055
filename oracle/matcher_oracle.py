"""CPU oracle for the NDT refinement stage of the scan matcher (matchScanCustom.m:32-50).

TEST INFRASTRUCTURE ONLY (see oracle/rbpf_oracle.py): nothing in ``thesis_amd/`` imports this.

**Parity unpinned.**  The reference calls MathWorks Navigation Toolbox R2021a ``matchScans`` (closed source, absent
from the reference tree, no recorded outputs).  What is restated here is the *published* algorithm that function
documents -- the Normal Distributions Transform of Biber & Strasser, "The Normal Distributions Transform: A New
Approach to Laser Scan Matching", IROS 2003 -- with the parameters the reference's call site fixes
(``'CellSize', 0.1``, ``'MaxIterations', 500``, matchScanCustom.m:36-37):

* the reference cloud is the set of occupied matcher cells (centres), binned into four overlapping grids of
  ``nc x nc`` matcher cells shifted by half an NDT cell (paper section III);
* a grid cell with at least 3 points carries their mean and sample covariance; the smaller eigenvalue is raised to
  0.001 times the larger one (paper section III, singular-covariance guard);
* score(p) = sum over points and grids of exp(-0.5 d' C^-1 d), d = T(p) x - mean (paper eq. 3);
* Newton's method on -score with the analytic gradient and Hessian (paper section V), made robust with a
  Levenberg-Marquardt damping term (the paper: "H is replaced by H + lambda I" when not positive definite).

The HIP stage (``kernels_match.hip``, ``ndt_*``) follows exactly these steps; this file is its checker.
Coordinates: region-relative matcher-cell units, exactly as the kernel holds them.
"""
from __future__ import annotations

from math import floor

import numpy as np

NDT_CELL_M = 0.1          # matchScanCustom.m:37
NDT_MAX_ITER = 500        # matchScanCustom.m:36
NDT_MIN_POINTS = 3
NDT_EIG_FLOOR = 1e-3
LAM0, LAM_MIN, LAM_MAX = 1e-3, 1e-7, 1e7
TOL_T, TOL_R = 1e-2, 2e-4     # convergence: proposed step below 0.01 cells (0.5 mm at 0.05 m) and 2e-4 rad
LAM_UP, LAM_DOWN = 100.0, 0.1  # damping after a rejected / an accepted trial


def _cell_stats(occ: np.ndarray, u0: np.ndarray, w0: np.ndarray, nc: int):
    """Mean and inverse covariance of the occupied cell centres inside the nc x nc blocks at (u0, w0)."""
    N = occ.shape[0]
    n = np.zeros(len(u0), dtype=np.int64)
    sx = np.zeros_like(n); sy = np.zeros_like(n); sxx = np.zeros_like(n); sxy = np.zeros_like(n); syy = np.zeros_like(n)
    for i in range(nc):
        for j in range(nc):
            uu, ww = u0 + i, w0 + j
            inside = (uu >= 0) & (uu < N) & (ww >= 0) & (ww < N)
            bit = np.zeros(len(u0), dtype=bool)
            bit[inside] = occ[uu[inside], ww[inside]]
            n += bit; sx += bit * i; sy += bit * j; sxx += bit * (i * i); sxy += bit * (i * j); syy += bit * (j * j)
    ok = n >= NDT_MIN_POINTS
    nn = np.where(ok, n, 3).astype(np.float64)
    mx, my = sx / nn, sy / nn
    a = (sxx - sx * mx) / (nn - 1.0)
    b = (sxy - sx * my) / (nn - 1.0)
    c = (syy - sy * my) / (nn - 1.0)
    half_tr = 0.5 * (a + c)
    disc = np.sqrt(0.25 * (a - c) * (a - c) + b * b)
    l1, l2 = half_tr + disc, half_tr - disc
    fix = ok & (l2 < NDT_EIG_FLOOR * l1)
    with np.errstate(divide="ignore", invalid="ignore"):
        k = np.where(fix, (NDT_EIG_FLOOR * l1 - l2) / (l1 - l2), 0.0)
    k = np.where(np.isfinite(k), k, 0.0)
    a2 = a + k * (l1 - a); b2 = b + k * (-b); c2 = c + k * (l1 - c)
    det = a2 * c2 - b2 * b2
    det = np.where(ok, det, 1.0)
    B00, B01, B11 = c2 / det, -b2 / det, a2 / det
    return ok, u0 + 0.5 + mx, w0 + 0.5 + my, B00, B01, B11


def ndt_eval(occ: np.ndarray, pts: np.ndarray, p, nc: int, ox: int, oy: int, single: bool = False):
    """f = -score, gradient (3), Hessian (6: xx xy xt yy yt tt) at p = (tx, ty, theta).

    ``single``: the kernel's hot form for nc == 2 -- the offset from the cell mean is formed from the point's float32
    position inside its matcher cell, everything after it in float32 (sums in float64)."""
    tx, ty, th = p
    sn, cs = np.sin(th), np.cos(th)
    bx, by = pts[:, 0], pts[:, 1]
    rx, ry = cs * bx - sn * by, sn * bx + cs * by
    ex, ey = rx + tx, ry + ty
    u, w = np.floor(ex).astype(np.int64), np.floor(ey).astype(np.int64)
    N = occ.shape[0]
    live = (u >= 0) & (u < N) & (w >= 0) & (w < N)
    m = np.zeros(10)
    h = nc // 2
    for g in range(4):
        gx, gy = (h if g & 1 else 0), (h if g & 2 else 0)
        u0 = u - np.mod(u + ox - gx, nc)
        w0 = w - np.mod(w + oy - gy, nc)
        ok, qx, qy, B00, B01, B11 = _cell_stats(occ, u0, w0, nc)
        ok &= live
        dx, dy = ex - qx, ey - qy
        if single:
            # the kernel's form: position inside the matcher cell (float64 -> float32) plus the cell's offset inside the NDT
            # cell, minus the table's float32 "0.5 + mean"; the terms are float32 (the kernel fuses multiply-adds and uses
            # the hardware exponential: agreement to ~1e-7, not bit for bit)
            f32 = np.float32
            fx, fy = (ex - np.floor(ex)).astype(f32), (ey - np.floor(ey)).astype(f32)
            dx = (fx + (u - u0).astype(f32)) - (qx - u0).astype(f32)
            dy = (fy + (w - w0).astype(f32)) - (qy - w0).astype(f32)
            B00, B01, B11 = B00.astype(f32), B01.astype(f32), B11.astype(f32)
            rx, ry = rx.astype(f32), ry.astype(f32)
            e0, e1 = B00 * dx + B01 * dy, B01 * dx + B11 * dy
            s = np.where(ok, np.exp(f32(-0.5) * (dx * e0 + dy * e1)), f32(0.0))
        else:
            e0, e1 = B00 * dx + B01 * dy, B01 * dx + B11 * dy
            s = np.where(ok, np.exp(-0.5 * (dx * e0 + dy * e1)), 0.0)
        c0, c1, c2 = e0, e1, e0 * (-ry) + e1 * rx
        bj0, bj1 = B00 * (-ry) + B01 * rx, B01 * (-ry) + B11 * rx
        f64 = np.float64
        m[0] -= s.sum(dtype=f64)
        m[1] += (s * c0).sum(dtype=f64); m[2] += (s * c1).sum(dtype=f64); m[3] += (s * c2).sum(dtype=f64)
        m[4] += (s * (-c0 * c0 + B00)).sum(dtype=f64)
        m[5] += (s * (-c0 * c1 + B01)).sum(dtype=f64)
        m[6] += (s * (-c0 * c2 + bj0)).sum(dtype=f64)
        m[7] += (s * (-c1 * c1 + B11)).sum(dtype=f64)
        m[8] += (s * (-c1 * c2 + bj1)).sum(dtype=f64)
        m[9] += (s * (-c2 * c2 + (-ry) * bj0 + rx * bj1 + e0 * (-rx) + e1 * (-ry))).sum(dtype=f64)
        if single:
            rx, ry = rx.astype(f64), ry.astype(f64)
    return m


def _lm_step(m, lam):
    """Solve (H + lam * diag(|H_ii| + 1e-12)) d = -g by Cholesky; None when the damped matrix is not positive definite."""
    g = m[1:4]
    A = np.array([[m[4], m[5], m[6]], [m[5], m[7], m[8]], [m[6], m[8], m[9]]])
    for i in range(3):
        A[i, i] += lam * (abs(A[i, i]) + 1e-12)
    l00 = A[0, 0]
    if not l00 > 0:
        return None
    l00 = np.sqrt(l00)
    l10, l20 = A[1, 0] / l00, A[2, 0] / l00
    d1 = A[1, 1] - l10 * l10
    if not d1 > 0:
        return None
    l11 = np.sqrt(d1)
    l21 = (A[2, 1] - l20 * l10) / l11
    d2 = A[2, 2] - l20 * l20 - l21 * l21
    if not d2 > 0:
        return None
    l22 = np.sqrt(d2)
    y0 = -g[0] / l00
    y1 = (-g[1] - l10 * y0) / l11
    y2 = (-g[2] - l20 * y0 - l21 * y1) / l22
    x2 = y2 / l22
    x1 = (y1 - l21 * x2) / l11
    x0 = (y0 - l10 * x1 - l20 * x2) / l00
    return np.array([x0, x1, x2])


NDT_STRIDE = 1            # points used by the ascent (1: all); the returned score always covers all points


def ndt_refine(occ: np.ndarray, pts_all: np.ndarray, start, nc: int, ox: int, oy: int, max_iter: int = NDT_MAX_ITER,
               single: bool = None, stride: int = NDT_STRIDE):
    """Damped Newton ascent of the NDT score from ``start``; returns (pose, score over all points, evaluations)."""
    single = (nc == 2) if single is None else single
    pts = pts_all[::stride]
    p = np.array(start, dtype=np.float64)
    cur = ndt_eval(occ, pts, p, nc, ox, oy, single)
    evals, lam = 1, LAM0
    while evals <= max_iter:
        d = None
        while d is None and lam <= LAM_MAX:
            d = _lm_step(cur, lam)
            if d is None:
                lam *= 10.0
        if d is None:
            break
        if max(abs(d[0]), abs(d[1])) < TOL_T and abs(d[2]) < TOL_R:
            break
        trial = ndt_eval(occ, pts, p + d, nc, ox, oy, single)
        evals += 1
        if trial[0] < cur[0]:
            p = p + d
            cur = trial
            lam = max(lam * LAM_DOWN, LAM_MIN)
        else:
            lam *= LAM_UP
            if lam > LAM_MAX:
                break
    return p, -ndt_eval(occ, pts_all, p, nc, ox, oy, single)[0], evals


def rasterise(ref_xy: np.ndarray, guess, mcs: float, N: int, cell_off: float, max_range: float):
    """Occupancy of the matcher region from reference points, as the kernel's mode-1 field stage does."""
    ox = int(floor(guess[0] / mcs)) - N // 2
    oy = int(floor(guess[1] / mcs)) - N // 2
    occ = np.zeros((N, N), dtype=bool)
    for rx, ry in np.asarray(ref_xy, dtype=np.float64).reshape(-1, 2):
        dx, dy = rx - guess[0], ry - guess[1]
        if not (np.sqrt(dx * dx + dy * dy) < max_range):
            continue
        u = int(floor(rx / mcs + cell_off)) - ox
        w = int(floor(ry / mcs + cell_off)) - oy
        if 0 <= u < N and 0 <= w < N:
            occ[u, w] = True
    return occ, ox, oy


def beams_in_cells(curr_xy: np.ndarray, mcs: float) -> np.ndarray:
    """Sensor-frame points in matcher-cell units, with the kernel's float32 staging."""
    c = np.asarray(curr_xy, dtype=np.float64).reshape(-1, 2).astype(np.float32)
    inv = np.float32(1.0 / mcs)
    return (c * inv).astype(np.float64)

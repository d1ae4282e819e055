"""The PRODUCT weighting look-ups (propose_weight_kernel's weight_beams, kernels_propose.hip) pinned per sample to the oracle.

rbpf_weight_samples runs the same device function as every scan step: single-precision cell addresses in home-tile
coordinates, the WSAFE guard band, the float64 redo (lookup_cell_home), four lanes per beam with eight samples each.
The cases aim at that code's edges: end points 0.01 .. 2 x WSAFE from cell borders on either axis and on both, the home
tile's rim with a neighbouring tile behind it, end points outside every tile, dim 800 / 1600 / 2048 (WSAFE changes above
1024), K = 1 / 7 / 30 / 32.  Reference: Robot._generate_sample_weight robot.py:118-139, GridMap.get_cell gridmap.py:120-128.
GPU only:  python -m pytest tests -m gpu"""
import numpy as np
import pytest

from oracle import c_oracle

pytestmark = pytest.mark.gpu

Q = 0.1


def _wsafe(dim):
    return 1e-3 if dim <= 1024 else 2e-3


def _build(eng_mod, lib, cs, P, B, rng, centres, **engine_options):
    """An engine and one C-oracle map per particle with the same dense random tiles."""
    e = eng_mod.ParticleEngine(P, max_beams=B, cell_size=cs, pool_tiles=P * (len(centres) + 1) + 2, **engine_options)
    maps = [c_oracle.CMap(lib, cs, **({"tile_len": engine_options["tile_len_m"]} if "tile_len_m" in engine_options else {})) for _ in range(P)]
    for p in range(P):
        for (cx, cy) in centres:
            cells = rng.integers(-30, 31, size=(e.dim, e.dim)).astype(np.int8)
            e.set_tile(p, (cx, cy), cells)
            maps[p].set_tile(cx, cy, cells.astype(np.float64) * Q)
    return e, maps


def _border_points(rng, cs, dim, n, centre, spread_cells):
    """n global end points whose sub-cell position sits 0.01 .. 2 x WSAFE from a cell border on x, on y, or on both."""
    ws = _wsafe(dim)
    mult = rng.choice([0.01, 0.05, 0.2, 0.5, 0.9, 1.1, 2.0], size=(n, 2))
    side = rng.integers(0, 2, size=(n, 2))                       # just above the lower border / just below the upper one
    frac = np.where(side == 0, mult * ws, 1.0 - mult * ws)
    which = rng.integers(0, 3, size=n)                            # 0: x at a border, 1: y, 2: both
    free = rng.uniform(0.1, 0.9, size=(n, 2))
    frac[:, 1] = np.where(which == 0, free[:, 1], frac[:, 1])
    frac[:, 0] = np.where(which == 1, free[:, 0], frac[:, 0])
    cell = np.floor(np.asarray(centre) / cs) + rng.integers(-spread_cells, spread_cells + 1, size=(n, 2))
    return (cell + frac) * cs


def _sensor_frame(points, pose):
    c, s = np.cos(pose[2]), np.sin(pose[2])
    d = points - np.asarray(pose[:2])
    return c * d[:, 0] + s * d[:, 1], -s * d[:, 0] + c * d[:, 1]


def _guesses(rng, pose, cs, K, P):
    """Sample 0 is the pose itself; half of the others are the pose moved by whole cells (the end points keep their
    sub-cell position), the rest are free perturbations."""
    g = np.tile(np.asarray(pose, dtype=np.float64), (P, K, 1))
    for p in range(P):
        for k in range(1, K):
            if k % 2:
                g[p, k, :2] += rng.integers(-3, 4, size=2) * cs
            else:
                g[p, k] += rng.normal(0, [0.04, 0.04, 0.02])
    return g


def _check(e, maps, guesses, sx, sy, prs):
    w = e.weight_samples(guesses, prs)
    worst = 0.0
    for p in range(len(maps)):
        want = np.asarray(maps[p].sample_weight(guesses[p], sx, sy, prs[p]), dtype=np.float64)
        worst = max(worst, float(np.max(np.abs(w[p] - want) / np.maximum(1.0, np.abs(want)))))
    return worst


@pytest.mark.parametrize("K", [1, 7, 30, 32])
@pytest.mark.parametrize("cs", [0.05, 0.025, 40.0 / 2048])
def test_product_lookups_at_cell_borders(cs, K):
    from thesis_amd import engine as eng_mod
    lib = c_oracle.load()
    rng = np.random.Generator(np.random.PCG64(int(1000 * cs) + K))
    P, B = 2, 1000
    try:
        e, maps = _build(eng_mod, lib, cs, P, B, rng, [(0, 0)])
    except eng_mod.RbpfError as err:                 # (a cell size the matcher's LDS layout cannot hold is refused by rbpf_create)
        pytest.skip(f"rbpf_create refuses cell_size {cs}: {err}")
    pose = (rng.uniform(-3, 3), rng.uniform(-3, 3), rng.uniform(-np.pi, np.pi))
    pts = _border_points(rng, cs, e.dim, B, pose[:2], int(6.0 / cs))
    sx, sy = _sensor_frame(pts, pose)
    e.set_scan_xy(sx, sy)
    g = _guesses(rng, pose, cs, K, P)
    prs = rng.uniform(0.5, 2.0, size=(P, K))
    assert _check(e, maps, g, sx, sy, prs) < 1e-12
    e.close()


def test_product_lookups_on_the_home_tile_rim_and_beyond():
    """The pose sits 0.6 m from the home tile's edge: a third of the end points fall into the neighbouring tile, some beyond
    every tile (None -> 0, hybridmap.py:85-93), and a fan of long beams ends up to 24.9 m away."""
    from thesis_amd import engine as eng_mod
    lib = c_oracle.load()
    rng = np.random.Generator(np.random.PCG64(77))
    cs, P, B, K = 0.05, 2, 1081, 30
    e, maps = _build(eng_mod, lib, cs, P, B, rng, [(0, 0), (40, 0)])
    pose = (19.4, rng.uniform(-2, 2), rng.uniform(-np.pi, np.pi))
    near = _border_points(rng, cs, e.dim, 700, pose[:2], int(3.0 / cs))
    ang = rng.uniform(-np.pi, np.pi, size=B - 700)
    rad = rng.uniform(5.0, 24.9, size=B - 700)
    far = np.asarray(pose[:2]) + np.column_stack([rad * np.cos(ang), rad * np.sin(ang)])
    pts = np.vstack([near, far])
    sx, sy = _sensor_frame(pts, pose)
    e.set_scan_xy(sx, sy)
    g = _guesses(rng, pose, cs, K, P)
    prs = rng.uniform(0.5, 2.0, size=(P, K))
    assert _check(e, maps, g, sx, sy, prs) < 1e-12
    e.close()


def test_product_lookups_with_more_beams_than_one_lds_chunk():
    """2000 beams: the kernel stages the beams that count in chunks of 1536 (two chunks, the second one partly padding);
    every fifth beam is outside the weighting's range (robot.py:130) and is not in the list at all."""
    from thesis_amd import engine as eng_mod
    lib = c_oracle.load()
    rng = np.random.Generator(np.random.PCG64(2000))
    cs, P, B, K = 0.05, 2, 2000, 30
    e, maps = _build(eng_mod, lib, cs, P, B, rng, [(0, 0)])
    pose = (rng.uniform(-3, 3), rng.uniform(-3, 3), rng.uniform(-np.pi, np.pi))
    pts = _border_points(rng, cs, e.dim, B, pose[:2], int(6.0 / cs))
    ang = rng.uniform(-np.pi, np.pi, size=B // 5)
    pts[::5] = np.asarray(pose[:2]) + 30.0 * np.column_stack([np.cos(ang), np.sin(ang)])      # beyond weight_max_range (25 m)
    sx, sy = _sensor_frame(pts, pose)
    e.set_scan_xy(sx, sy)
    g = _guesses(rng, pose, cs, K, P)
    prs = rng.uniform(0.5, 2.0, size=(P, K))
    assert _check(e, maps, g, sx, sy, prs) < 1e-12
    e.close()


def test_product_lookups_with_beams_outside_the_error_budget():
    """8 m tiles (dim 160): a beam whose |x| + |y| exceeds 1.5 tile lengths is outside the premise of the single-precision
    addresses - the host stores it as NaN in the weighting's beam list and every one of its look-ups goes the float64 way
    (more of them than the queue holds: the rest are done on the spot).  Nine tiles round the pose, end points up to 24 m away."""
    from thesis_amd import engine as eng_mod
    lib = c_oracle.load()
    rng = np.random.Generator(np.random.PCG64(10))
    cs, P, B, K = 0.05, 2, 1081, 30
    centres = [(8 * i, 8 * j) for i in (-1, 0, 1) for j in (-1, 0, 1)]
    e, maps = _build(eng_mod, lib, cs, P, B, rng, centres, tile_len_m=8, max_ray_m=7.0)   # (rays must be shorter than a tile: the map update is not run here)
    assert e.dim == 160
    pose = (rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(-np.pi, np.pi))
    near = _border_points(rng, cs, e.dim, 500, pose[:2], int(3.0 / cs))
    ang = rng.uniform(-np.pi, np.pi, size=B - 500)
    rad = rng.uniform(4.0, 24.0, size=B - 500)
    far = np.asarray(pose[:2]) + np.column_stack([rad * np.cos(ang), rad * np.sin(ang)])
    pts = np.vstack([near, far])
    sx, sy = _sensor_frame(pts, pose)
    assert np.sum(np.abs(sx) + np.abs(sy) > 12.0) > 100            # beams outside the budget exist
    e.set_scan_xy(sx, sy)
    g = _guesses(rng, pose, cs, K, P)
    prs = rng.uniform(0.5, 2.0, size=(P, K))
    assert _check(e, maps, g, sx, sy, prs) < 1e-12
    e.close()


def test_guard_band_is_what_keeps_the_single_precision_addresses_exact(monkeypatch):
    """With the guard band forced to 0 (RBPF_WSAFE, a test knob read by rbpf_create) the single-precision addresses are
    taken for every look-up, and end points 1e-5 .. 2e-4 cell from a border land in the wrong cell (the pose sits in the
    tile's far corner, where a single-precision cell coordinate resolves 6e-5): the same case that passes
    above must then FAIL.  The float64 entry of round 1 (RBPF_WEIGHT_ENTRY=f64) agrees with the oracle as before."""
    from thesis_amd import engine as eng_mod
    lib = c_oracle.load()
    cs, P, B, K = 0.05, 2, 1000, 30
    results = {}
    for label, env in [("product", {}), ("no guard", {"RBPF_WSAFE": "0"}), ("f64 entry", {"RBPF_WEIGHT_ENTRY": "f64"})]:
        for k in ("RBPF_WSAFE", "RBPF_WEIGHT_ENTRY"):
            monkeypatch.delenv(k, raising=False)
        for k, val in env.items():
            monkeypatch.setenv(k, val)
        rng = np.random.Generator(np.random.PCG64(5))
        e, maps = _build(eng_mod, lib, cs, P, B, rng, [(0, 0)])
        pose = (15.2345, 14.777, 0.9)
        pts = _border_points(rng, cs, e.dim, B, pose[:2], int(6.0 / cs))
        sx, sy = _sensor_frame(pts, pose)
        e.set_scan_xy(sx, sy)
        g = _guesses(rng, pose, cs, K, P)
        prs = rng.uniform(0.5, 2.0, size=(P, K))
        results[label] = _check(e, maps, g, sx, sy, prs)
        e.close()
    assert results["product"] < 1e-12 and results["f64 entry"] < 1e-12
    assert results["no guard"] > 1e-6, "without the guard band the look-ups should leave the reference's cells"

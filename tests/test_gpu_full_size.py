"""BASELINE configs[3] and [4] at their per-GPU sizes, and how often a real log leaves the fast map kernels.

configs[3]: 16 384 particles over 8 GPUs = 2 048 per GPU, 1081 beams, 0.05 m, a long run.
configs[4]: 65 536 particles over 8 GPUs = 8 192 per GPU, 181 beams (data/orebro.log), 0.025 m: 21 GB of tiles.
At these sizes the oracle cannot follow every particle, so the tests pin (a) a few particles cell for cell against the
oracle over the first scans, with the poses the engine itself chose read back, (b) the counters the oracle also keeps
(ray cells, written cells), (c) size-independent properties of the whole population over the rest of the run.
The numbers DESIGN.md quotes (fallback fractions, windows per particle) are written to gpurun_out/full_size.json."""
import json
import os

import numpy as np
import pytest

from oracle import c_oracle, rbpf_oracle as orc

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
REPORT = os.path.join(os.path.dirname(HERE), "gpurun_out", "full_size.json")


def _report(key, value):
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        d = json.load(open(REPORT)) if os.path.exists(REPORT) else {}
        d[key] = value
        json.dump(d, open(REPORT, "w"), indent=1, sort_keys=True)
    except OSError:
        pass


def _oracle_dump(cmap):
    return {c: np.rint(t / 0.1).astype(np.int8) for c, t in cmap.tiles().items()}


def _run_config(P, B, cs, fov, steps_checked, steps_total, probe):
    """The bench's step (bench.Runner: IMU, scan update with both matcher stages, resample) at full size."""
    from bench import Runner, PERIOD_S
    from thesis_amd.datasets import synthetic
    log = synthetic.make_log(steps_total + 3, B, period=PERIOD_S, fov=fov)
    angles, ranges, odo, truth = log
    r = Runner(P, B, cs, log)
    e = r.e
    # (a) the first scans without resampling: the probed particles' maps must equal the oracle's, fed with the poses the
    #     engine used (its own mean poses, read back after each update)
    lib = c_oracle.load()                                                        # the C restatement (pinned to G3 bit for bit)
    maps = {p: c_oracle.CMap(lib, cs) for p in probe}
    sx, sy = orc.scan_xy(ranges[0], angles)
    for p in probe:
        maps[p].update((0.0, 0.0, 0.0), sx, sy)
    c0 = e.counters()
    assert c0["ray_cells_visited"] == P * lib.orc_map_cells_visited(maps[probe[0]].h)   # all particles start at the origin
    for k in range(steps_checked):
        e.imu_update("velocity", odo[k], PERIOD_S * 1e4)
        e.set_scan(ranges[k + 1], angles)
        e.scan_update(adj=False)
        poses = e.poses()
        sx, sy = orc.scan_xy(ranges[k + 1], angles)
        for p in probe:
            maps[p].update(tuple(float(x) for x in poses[p]), sx, sy)
    for p in probe:
        want = _oracle_dump(maps[p])
        got = dict(e.tiles(p))
        assert set(got) == set(want), (p, sorted(got), sorted(want))
        for c in want:
            assert np.array_equal(got[c], want[c]), f"particle {p} tile {c}: {int(np.count_nonzero(got[c] != want[c]))} cells differ"
    # (c) the rest of the run with the reference's cadence and resampling
    r.frame = steps_checked
    e.set_profiling(True)                                                        # restarts the counters
    n = steps_total - steps_checked
    for _ in range(n):
        r.step()
    c = e.counters()
    poses, w, cov = e.poses(), e.weights(), e.covs()
    assert np.all(np.isfinite(poses)) and np.all(np.isfinite(w)) and np.all(np.isfinite(cov))
    err = np.linalg.norm(np.median(poses[:, :2], axis=0) - truth[steps_total, :2])
    assert err < 0.5, f"the particle cloud left the simulated truth by {err:.2f} m"
    for p in probe:
        for _, cells in e.tiles(p):
            assert cells.min() >= -30 and cells.max() <= 30
    ms = e.kernel_ms("raycast")
    out = {"particles": P, "beams": B, "cell_size": cs, "steps": n,
           "window_fallbacks_per_particle_step": c["window_fallbacks"] / (P * n),
           "fast_kernel_give_backs_per_particle_step": (c["fallback_geometry"] + c["fallback_bound"] + c["fallback_tables"]) / (P * n),
           "ray_kernel_windows_per_particle_step": c["map_windows"] / (P * n),
           "tiles_in_use": c["tiles_in_use"], "map_update_ms_mean": float(ms[1:].mean()) if len(ms) > 1 else None,
           "median_position_error_m": float(err)}
    e.close()
    return out


def test_config4_share_2048_particles_1081_beams_long_run():
    out = _run_config(2048, 1081, 0.05, 1.5 * np.pi, steps_checked=4, steps_total=300, probe=(0, 1023, 2047))
    _report("configs[3] share", out)
    assert out["window_fallbacks_per_particle_step"] == 0.0
    assert out["tiles_in_use"] == 2048                       # the 16 m room stays inside the first 40 m tile


def test_config5_share_8192_particles_181_beams_0025m():
    out = _run_config(8192, 181, 0.025, np.pi, steps_checked=3, steps_total=24, probe=(0, 4095, 8191))
    _report("configs[4] share", out)
    assert out["window_fallbacks_per_particle_step"] == 0.0  # every fan is 640 cells wide: strips of the global-index kernel
    assert out["ray_kernel_windows_per_particle_step"] > 1.0
    assert out["tiles_in_use"] == 8192


@pytest.mark.parametrize("cs", [0.05, 0.1])
def test_intel_head_fallback_fraction(cs):
    """The first 70 scans of data/intel.txt through the reference's loop (rays up to the 15 m cap of hybridmap.py:107 make
    fans of up to 600 cells at 0.05 m): how many particle-updates the global-index kernel hands to the 128x128-window
    kernel (none), and how many strips it needs."""
    from thesis_amd.datasets.carmen import load_carmen
    from thesis_amd.slam import ParticleFilter, run_log
    log = load_carmen(os.path.join(HERE, "golden", "intel_head.log"))
    pf = ParticleFilter(64, log.angles, motion_model="absolute", cell_size=cs)
    pf.engine.set_profiling(True)
    res = run_log(pf, log.scans, log.scan_times, log.odom, log.odom_times, max_frames=70, order=log.order)
    c = pf.engine.counters()
    n = 64 * max(1, c["scan_updates"])
    out = {"accepted_scans": res.accepted, "map_updates": c["scan_updates"],
           "window_fallbacks_per_particle_update": c["window_fallbacks"] / n,
           "fast_kernel_give_backs_per_particle_update": (c["fallback_geometry"] + c["fallback_bound"] + c["fallback_tables"]) / n,
           "ray_kernel_windows_per_particle_update": c["map_windows"] / n}
    _report(f"intel_head cs={cs}", out)
    pf.close()
    if cs == 0.05:
        assert out["window_fallbacks_per_particle_update"] <= 0.02


def test_fans_larger_than_the_lds_window_run_in_strips(monkeypatch):
    """15 m rays all round at 0.05 m: 600-cell fans, three times what the event-walk kernel's LDS window holds: the fan is
    processed in strips of rows inside the one launch (any partition of the ray steps may be written back on its own).
    Cell-exact against the C oracle for a few particles, no particle reaches the 128x128-window kernel."""
    monkeypatch.delenv("RBPF_MAP_KERNEL", raising=False)
    from thesis_amd.engine import ParticleEngine
    P, B = 96, 721
    ang = np.linspace(-np.pi, np.pi, B, endpoint=False)
    rng = np.random.Generator(np.random.PCG64(5))
    e = ParticleEngine(P, max_beams=B, pool_tiles=8 * P)
    lib = c_oracle.load()
    probe = (0, 17, 95)
    maps = {p: c_oracle.CMap(lib, 0.05) for p in probe}
    poses = np.column_stack([rng.uniform(-1, 1, P), rng.uniform(-1, 1, P), rng.uniform(-3, 3, P)])
    n_scans = 4
    for k in range(n_scans):
        r = 13.0 + 1.5 * np.sin(5 * ang + k) + rng.normal(0, 0.01, B)          # 11.5 .. 14.5 m, under the 15 m cap
        e.set_scan(r, ang)
        e.map_update(poses)
        sx, sy = orc.scan_xy(r, ang)
        for p in probe:
            maps[p].update(poses[p], sx, sy)
        poses = poses + [0.2, -0.1, 0.05]
    c = e.counters()
    for p in probe:
        want = _oracle_dump(maps[p])
        got = dict(e.tiles(p))
        assert set(got) == set(want)
        for cc in want:
            assert np.array_equal(got[cc], want[cc]), f"particle {p} tile {cc}"
    assert c["window_fallbacks"] == 0 and c["fallback_geometry"] == 0 and c["fallback_bound"] == 0
    assert c["map_windows"] >= 2 * P * n_scans                               # several strips per particle and scan
    e.close()


def test_config2_map_update_of_a_particle_depends_on_its_pose_alone():
    """BASELINE configs[2] at full size (4096 particles x 1081 beams, 0.05 m), a property no oracle run is needed for: the
    map a particle gets depends on ITS pose and the scans only (hybridmap.py:95-145 is a method of one map).  Two engines
    get the same 4096 poses in opposite order for three scans; particle i of the first must hold the tiles of particle
    4095 - i of the second, cell for cell - 4096 workgroups share the tile pool, its allocator and the LDS-window chain,
    and nothing of that may leak between particles.  64 particles are read back and compared, eight of them also against
    the C oracle; the counters of both runs (ray cells, cells written) must agree exactly.  The sample weights of a fourth
    scan (4096 x 30 samples, the product's look-ups) are mirrored too, bit for bit, and the oracle's for the eight."""
    from thesis_amd.engine import ParticleEngine
    from thesis_amd.datasets import synthetic
    P, B, cs = 4096, 1081, 0.05
    angles, ranges, _, truth = synthetic.make_log(4, B, period=0.1)
    rng = np.random.Generator(np.random.PCG64(4096))
    poses = [np.asarray(truth[k]) + rng.normal(0, [0.15, 0.15, 0.03], size=(P, 3)) for k in range(3)]
    runs = []
    for flip in (False, True):
        e = ParticleEngine(P, max_beams=B, cell_size=cs, pool_tiles=P + 64)
        for k in range(3):
            e.set_scan(ranges[k], angles)
            e.map_update(poses[k][::-1].copy() if flip else poses[k])
        runs.append(e)
    a, b = runs
    # the sample weights of the next scan (robot.py:118-139) on those maps, K = 30 samples round every particle's last pose
    K = 30
    guesses = poses[2][:, None, :] + rng.normal(0, [0.04, 0.04, 0.015], size=(P, K, 3))
    prs = rng.uniform(0.5, 2.0, size=(P, K))
    a.set_scan(ranges[3], angles); b.set_scan(ranges[3], angles)
    wa = a.weight_samples(guesses, prs)
    wb = b.weight_samples(guesses[::-1].copy(), prs[::-1].copy())
    assert np.array_equal(wa, wb[::-1]) and np.all(np.isfinite(wa))
    ca, cb = a.counters(), b.counters()
    assert ca["ray_cells_visited"] == cb["ray_cells_visited"] and ca["cells_written"] == cb["cells_written"] > 0
    assert ca["window_fallbacks"] == cb["window_fallbacks"]
    picks = rng.choice(P, size=64, replace=False)
    lib = c_oracle.load()
    for n, i in enumerate(picks):
        ta = {c: t for c, t in a.tiles(int(i))}
        tb = {c: t for c, t in b.tiles(int(P - 1 - i))}
        assert ta.keys() == tb.keys() and sum(int(np.count_nonzero(t)) for t in ta.values()) > 20000      # (a 16 m room at 0.05 m)
        for c in ta:
            assert np.array_equal(ta[c], tb[c]), f"particle {i}: tile {c} differs between the two orders"
        if n < 8:                                            # and they are the reference's cells
            m = c_oracle.CMap(lib, cs)
            x, y = ranges[0] * np.cos(angles), ranges[0] * np.sin(angles)
            for k in range(3):
                x, y = ranges[k] * np.cos(angles), ranges[k] * np.sin(angles)
                m.update(poses[k][i], x, y)
            want = _oracle_dump(m)
            got = {c: t for c, t in ta.items() if np.any(t)}
            assert set(got) == {c for c, t in want.items() if np.any(t)}
            for c in got:
                assert np.array_equal(got[c], want[c])
            x3, y3 = ranges[3] * np.cos(angles), ranges[3] * np.sin(angles)
            ww = np.asarray(m.sample_weight(guesses[i], x3, y3, prs[i]), dtype=np.float64)
            assert np.max(np.abs(wa[i] - ww) / np.maximum(1.0, np.abs(ww))) < 1e-12
    a.close(); b.close()


def test_config2_full_step_is_deterministic_at_full_size():
    """BASELINE configs[2] (4096 x 1081, both matcher stages, resampling): two runs of the bench's loop over 12 scans end in
    the same poses, covariances and weights bit for bit, and particle 0's map is the same - LDS atomics, work queues and the
    order in which 4096 workgroups take pool tiles may not show in any result (integer map arithmetic, fixed-order float
    reductions, proposal streams keyed by particle and step)."""
    from bench import Runner, PERIOD_S
    from thesis_amd.datasets import synthetic
    log = synthetic.make_log(15, 1081, period=PERIOD_S)
    out = []
    for _ in range(2):
        r = Runner(4096, 1081, 0.05, log)
        for _k in range(12):
            r.step()
        out.append((r.e.poses().copy(), r.e.covs().copy(), r.e.weights().copy(), dict(r.e.tiles(0)), r.e.counters()))
        r.e.close()
    (p0, c0, w0, t0, k0), (p1, c1, w1, t1, k1) = out
    assert np.array_equal(p0, p1) and np.array_equal(c0, c1, equal_nan=True) and np.array_equal(w0, w1)
    assert t0.keys() == t1.keys() and all(np.array_equal(t0[c], t1[c]) for c in t0)
    for key in ("ray_cells_visited", "cells_written", "ndt_runs", "ndt_evaluations", "ndt_accepted", "match_shared", "window_fallbacks"):
        assert k0[key] == k1[key], key
    assert np.all(np.isfinite(p0)) and np.all(np.isfinite(w0))

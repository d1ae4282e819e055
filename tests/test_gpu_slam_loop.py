"""The reference's main loop (main.py:138-214) over the engine: BASELINE config C1 (Intel log, 64 particles,
0.1 m grid, native 180 beams) on data/intel.txt (data fixtures: tests/golden/intel_head.log = its first scans,
tests/golden/intel.txt.gz = the whole log, 910 scans, compressed)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_intel_replay_c1():
    from thesis_amd.datasets.carmen import load_carmen
    from thesis_amd.slam import ParticleFilter, run_log
    log = load_carmen(os.path.join(HERE, "golden", "intel_head.log"))
    assert log.scans.shape[1] == 180 and len(log.scans) >= 70
    pf = ParticleFilter(64, log.angles, motion_model="absolute", cell_size=0.1)
    res = run_log(pf, log.scans, log.scan_times, log.odom, log.odom_times, max_frames=70, order=log.order)
    assert res.frames == 70 and res.accepted >= 8
    poses = pf.engine.poses()
    assert np.all(np.isfinite(poses)) and np.all(np.isfinite(pf.engine.weights()))
    # particles stay near the odometry track (the absolute model passes ODOM poses through, IntelIMUData.py:23-25)
    pos = int(np.nonzero(log.order == 69)[0][0])               # record of the 70th scan in the file
    last_odom = -int(log.order[:pos][log.order[:pos] < 0][-1]) - 1
    assert np.all(np.linalg.norm(poses[:, :2] - log.odom[last_odom, :2], axis=1) < 1.5)
    xs, ys = pf.particles[0]._map.get_occupied_points()
    assert len(xs) > 300                                   # walls have been mapped
    traj = pf.trajectory(0)
    assert len(traj) > 70 and np.all(np.isfinite(np.array(traj)))
    assert pf.particles[0]._map.get_odds_at((0.0, 0.0)) is not None
    pf.close()


def test_intel_replay_c1_whole_log():
    """BASELINE configs[0] in full: all 910 scans of data/intel.txt (14 541 ODOM records) through run_log with 64
    particles at 0.1 m cells (dim 400: the first map kernel's last 32-cell group of a tile row is partial), the
    absolute-pose model of IntelIMUData.py:23-36.  The reference itself cannot replay this file (its cursors stall at the
    first duplicated 0.1 s tick, tests/test_host_logic.py); checked here: every record is consumed, the state stays
    finite, the particles stay on the odometry track, the map grows over several tiles and no particle leaves the first
    map kernel."""
    from thesis_amd.datasets.carmen import load_carmen
    from thesis_amd.slam import ParticleFilter, run_log
    log = load_carmen(os.path.join(HERE, "golden", "intel.txt.gz"))
    assert log.scans.shape == (910, 180) and len(log.odom) == 14541
    pf = ParticleFilter(64, log.angles, motion_model="absolute", cell_size=0.1, keep_history=False, pool_tiles=64 * 20)
    res = run_log(pf, log.scans, log.scan_times, log.odom, log.odom_times, order=log.order)
    assert res.frames == 910 and sum(1 for ev in res.trace if ev[0] == "imu") == 14541 and res.accepted >= 300
    poses = pf.engine.poses()
    assert np.all(np.isfinite(poses)) and np.all(np.isfinite(pf.engine.weights()))
    assert np.all(np.linalg.norm(poses[:, :2] - log.odom[-1, :2], axis=1) < 3.0)
    c = pf.engine.counters()
    assert c["window_fallbacks"] <= 0.01 * 64 * res.accepted, (c["window_fallbacks"], c["fallback_geometry"], c["fallback_bound"])
    tiles = pf.engine.tiles(0)
    assert len(tiles) >= 2 and sum(int(np.count_nonzero(t)) for _, t in tiles) > 20000
    xs, ys = pf.particles[0]._map.get_occupied_points()
    assert len(xs) > 2000
    pf.close()


def test_synthetic_room_loop_tracks_truth():
    """Velocity-model odometry with 1 % noise + built-in matcher: particle 0 stays within 0.15 m of the truth."""
    from thesis_amd.datasets import synthetic
    from thesis_amd.slam import ParticleFilter, run_log
    period = 0.7
    angles, ranges, odo, truth = synthetic.make_log(40, 361, period=period)
    scan_times = (np.arange(41) * period * 1e4).astype(np.int64)
    odom_times = scan_times[1:] - 1                       # the reading that moves k -> k+1 arrives just before scan k+1
    pf = ParticleFilter(128, angles, motion_model="velocity", cell_size=0.05)
    # readings carry dt = time since the previous reading
    res = run_log(pf, ranges, scan_times, odo, odom_times)
    assert res.accepted >= 30
    p0 = np.array(pf.particles[0].get_latest_pose())
    assert np.linalg.norm(p0[:2] - truth[-1, :2]) < 0.15 and abs(p0[2] - truth[-1, 2]) < 0.05
    pf.close()


def test_checkpoint_continues_bit_identically(tmp_path):
    """a run restored from the portable checkpoint continues exactly like the original
    (poses, covariances, weights, ancestors, every map cell) - including the device-side random streams."""
    from thesis_amd import engine
    from thesis_amd.datasets import synthetic
    P, B = 24, 1081
    ang, ranges, odo, poses = synthetic.make_log(6, B, period=0.7)
    a = engine.ParticleEngine(P, max_beams=B, pool_tiles=4 * P, seed=7)
    a.set_scan(ranges[0], ang)
    a.map_update(np.zeros((P, 3)))

    def step(e, k):
        e.imu_update("velocity", odo[k], 7000.0)
        e.set_scan(ranges[k + 1], ang)
        e.scan_update(adj=False)
        return e.resample(float("nan"))           # internal uniform: part of the checkpointed stream state

    for k in range(3):
        step(a, k)
    path = str(tmp_path / "run.npz")
    a.save_checkpoint(path)
    b = engine.ParticleEngine.from_checkpoint(path)
    for k in range(3, 5):
        da, ia = step(a, k)
        db, ib = step(b, k)
        assert da == db and np.array_equal(ia, ib)
    assert np.array_equal(a.poses(), b.poses()) and np.array_equal(a.covs(), b.covs()) and np.array_equal(a.weights(), b.weights())
    for p in range(P):
        ta, tb = a.tiles(p), b.tiles(p)
        assert [c for c, _ in ta] == [c for c, _ in tb]
        for (_, ca), (_, cb) in zip(ta, tb):
            assert np.array_equal(ca, cb)
    a.close(); b.close()


def test_long_run_first_kernel_equals_window_kernel(monkeypatch):
    """120 steps of the full pipeline (matcher, proposal, map update, resample) on two engines that differ only in the
    map-update kernel (event-walk kernel vs 128x128 windows): states after every step and all maps at the end are identical.
    The first kernel's rare paths (fallback to windows, full event list, NaN branch) all leave the result unchanged."""
    from thesis_amd import engine
    from thesis_amd.datasets import synthetic
    P, B, T = 192, 1081, 120
    ang, ranges, odo, _ = synthetic.make_log(T, B, period=0.7)
    monkeypatch.delenv("RBPF_MAP_KERNEL", raising=False)
    a = engine.ParticleEngine(P, max_beams=B, pool_tiles=4 * P, seed=5)
    monkeypatch.setenv("RBPF_MAP_KERNEL", "window")
    b = engine.ParticleEngine(P, max_beams=B, pool_tiles=4 * P, seed=5)
    urng = np.random.Generator(np.random.PCG64(12))
    for e in (a, b):
        e.set_scan(ranges[0], ang)
        e.map_update(np.zeros((P, 3)))
        e.refresh_last_scan(0)
    for k in range(T):
        u = float(urng.random())
        adj = not (k % 5 < 2)
        res = []
        for e in (a, b):
            e.imu_update("velocity", odo[k], 7000.0)
            e.set_scan(ranges[k + 1], ang)
            e.scan_update(adj=adj)
            res.append(e.resample(u))
            if k % 5 == 0:
                e.refresh_last_scan(0)
        assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1]), f"ancestors differ at step {k}"
        if k % 10 == 9:
            assert np.array_equal(a.poses(), b.poses()) and np.array_equal(a.weights(), b.weights()), f"state differs at step {k}"
    ca, cb = a.counters(), b.counters()
    assert cb["window_fallbacks"] == 0 and ca["ray_cells_visited"] == cb["ray_cells_visited"] and ca["cells_written"] == cb["cells_written"]
    for p in range(0, P, 7):
        ta, tb = a.tiles(p), b.tiles(p)
        assert [c for c, _ in ta] == [c for c, _ in tb]
        for (_, x), (_, y) in zip(ta, tb):
            assert np.array_equal(x, y)
    a.close(); b.close()


def test_map_view_seam_methods():
    """HybridMapView.get_scan_match / get_scan_adj / is_occ_at: the reference's per-map methods (hybridmap.py:147-261)
    on one particle's map, through rbpf_match_inputs + rbpf_match_scan."""
    from thesis_amd.slam import ParticleFilter
    from thesis_amd.datasets import synthetic
    from oracle import rbpf_oracle as orc
    ang = synthetic.beam_angles(1081)
    rng = np.random.Generator(np.random.PCG64(8))
    truth = np.array([0.6, -0.2, 0.25])
    pf = ParticleFilter(2, ang, motion_model="velocity", cell_size=0.05)
    try:
        for _ in range(5):
            pf.engine.set_scan(synthetic.cast_scan(truth, ang, rng), ang)
            pf.engine.map_update(np.broadcast_to(truth, (2, 3)))
        view = pf.particles[1]._map
        # a wall cell of the 16 m room is occupied, the room's interior is not
        assert view.is_occ_at(8.02, 0.5) and not view.is_occ_at(2.0, 1.0) and not view.is_occ_at(300.0, 0.0)
        r = synthetic.cast_scan(truth, ang, rng)
        pf.engine.set_scan(r, ang)
        guess = truth + [0.12, -0.09, 0.03]
        pose, cov, score = view.get_scan_match(guess, [0.4, 0.4, np.pi / 6])
        assert np.all(np.isfinite(cov)) and score > 0
        assert np.all(np.abs(pose[:2] - truth[:2]) < 0.08) and abs(pose[2] - truth[2]) < 0.01
        sx, sy = orc.scan_xy(r, ang)
        gx, gy = orc.transform(sx, sy, tuple(truth))
        pose, cov, score = view.get_scan_adj(np.stack([sx, sy], 1), np.stack([gx, gy], 1), guess, [0.4, 0.4, np.pi / 6])
        assert np.all(np.isfinite(cov)) and np.all(np.abs(pose[:2] - truth[:2]) < 0.08) and abs(pose[2] - truth[2]) < 0.01
    finally:
        pf.close()


def _run_loop(steps, P=48, seed=5):
    from bench import Runner, PERIOD_S
    from thesis_amd.datasets import synthetic
    log = synthetic.make_log(steps + 3, 1081, period=PERIOD_S)
    r = Runner(P, 1081, 0.05, log)
    for _ in range(steps):
        r.step()
    out = (r.e.poses(), r.e.covs(), r.e.weights(), [dict(r.e.tiles(p)) for p in (0, P // 2, P - 1)], r.e.counters())
    r.e.close()
    return out


@pytest.mark.parametrize("P,steps", [(48, 25), (4608, 6)])            # the fused resample kernel, and the multi-kernel path above 4096 particles
def test_sharing_the_match_of_exact_duplicates_changes_nothing(monkeypatch, P, steps):
    """After a resample the copies of one ancestor have the same pose, covariance and map until the next proposal
    draws their samples, so their scan matches are identical: the matcher runs once per group and the proposal reads
    the representative's result.  RBPF_MATCH_DEDUP=0 runs every particle; both must give the same bits everywhere."""
    monkeypatch.delenv("RBPF_MATCH_DEDUP", raising=False)
    a = _run_loop(steps, P)
    monkeypatch.setenv("RBPF_MATCH_DEDUP", "0")
    b = _run_loop(steps, P)
    assert a[4]["match_shared"] > 0 and b[4]["match_shared"] == 0
    for x, y in zip(a[:3], b[:3]):
        assert np.array_equal(x, y)
    for ta, tb in zip(a[3], b[3]):
        assert set(ta) == set(tb)
        for c in ta:
            assert np.array_equal(ta[c], tb[c])


def test_occupied_points_and_is_occ_at_equal_the_oracle_modulo_threshold_noise():
    """HybridMapView.get_occupied_points / is_occ_at against the oracle's restatement of hybridmap.py:303-313 /
    gridmap.py:255-260 on maps built by the same updates.  The engine stores cells on the 0.1 lattice; the reference's
    float64 cells carry rounding noise, so a cell whose exact value IS the threshold (1.0 = 10 quanta) reads as occupied
    in the reference when its noise is positive (0.8 + 0.8 - 0.3 - 0.3 = 1.0000000000000002) and never here: the
    comparison holds for every other cell, and the number of exactly-threshold cells is reported."""
    from thesis_amd.slam import ParticleFilter
    from thesis_amd.datasets import synthetic
    from oracle import rbpf_oracle as orc, loop_oracle
    ang = synthetic.beam_angles(1081)
    rng = np.random.Generator(np.random.PCG64(18))
    pf = ParticleFilter(2, ang, motion_model="velocity", cell_size=0.05)
    hm = orc.OracleHybridMap(0.05)
    try:
        poses = [np.array([0.3 * k, 0.1 * k, 0.05 * k]) for k in range(6)]
        for pose in poses:
            r = synthetic.cast_scan(pose, ang, rng)
            pf.engine.set_scan(r, ang)
            pf.engine.map_update(np.broadcast_to(pose, (2, 3)))
            sx, sy = orc.scan_xy(r, ang)
            hm.update(tuple(float(v) for v in pose), sx, sy)
        view = pf.particles[1]._map
        gx, gy = view.get_occupied_points()
        tiles = [((t.cx, t.cy), t.map) for t in hm.tiles]
        ox, oy = loop_oracle.occupied_points(tiles, 0.05)
        got = set(zip(np.rint(gx * 2).astype(int).tolist(), np.rint(gy * 2).astype(int).tolist()))     # half-cell units: exact keys
        want = set(zip(np.rint(np.asarray(ox) * 2).astype(int).tolist(), np.rint(np.asarray(oy) * 2).astype(int).tolist()))
        # cells that sit exactly on the threshold in exact arithmetic
        edge = set()
        n_edge = 0
        for (cx, cy), m in tiles:
            dim = m.shape[0]
            i, j = np.nonzero(np.abs(m - 1.0) < 1e-9)
            n_edge += len(i)
            ex = ((i - dim / 2) * 0.05 + cx) / 0.05; ey = ((j - dim / 2) * 0.05 + cy) / 0.05
            edge |= set(zip(np.rint(ex * 2).astype(int).tolist(), np.rint(ey * 2).astype(int).tolist()))
        assert got - edge == want - edge and got <= want | edge and len(want) > 1000
        # is_occ_at on a sample of occupied and free cells away from the threshold
        for (cx, cy), m in tiles:
            dim = m.shape[0]
            occ = np.argwhere(m > 1.05)[:40]; free = np.argwhere((m < 0.95) & (m != 0))[:40]
            for (i, j), expect in [(c, True) for c in occ] + [(c, False) for c in free]:
                x, y = (i - dim / 2 + 0.5) * 0.05 + cx, (j - dim / 2 + 0.5) * 0.05 + cy
                assert view.is_occ_at(x, y) == expect == (hm.get_odds_at(x, y) > 1.0)
        try:
            import json
            rep = os.path.join(os.path.dirname(HERE), "gpurun_out", "threshold_cells.json")
            os.makedirs(os.path.dirname(rep), exist_ok=True)
            json.dump({"occupied_cells_oracle": len(want), "exactly_threshold_cells": n_edge,
                       "occupied_only_in_reference": len(want - got)}, open(rep, "w"))
        except OSError:
            pass
    finally:
        pf.close()


def test_orebro_replay_all_scans_at_0025m():
    """BASELINE configs[4]'s log: data/orebro.log (a data fixture, tests/golden/orebro.log), all 237 scans, 0.025 m grid.
    The file is a corrected log: ODOM / FLASER pairs in file order carrying poses, its time fields hold no usable
    clock (the reference's Obero adapters sort them into one scan at time 0, tests/golden/G12) - so the replay takes
    the records in file order with the absolute-pose motion model (IntelIMUData.py:23-25 shape)."""
    from thesis_amd.datasets.carmen import load_carmen
    from thesis_amd.slam import ParticleFilter, run_log
    log = load_carmen(os.path.join(HERE, "golden", "orebro.log"))
    assert log.scans.shape == (237, 181)
    pf = ParticleFilter(128, log.angles, motion_model="absolute", cell_size=0.025, pool_tiles=128 * 24)   # the path crosses several 40 m tiles
    pf.engine.set_profiling(True)
    res = run_log(pf, log.scans, log.scan_times, log.odom, log.odom_times, order=log.order)
    assert res.frames == 237 and res.accepted >= 200
    poses = pf.engine.poses()
    assert np.all(np.isfinite(poses)) and np.all(np.isfinite(pf.engine.weights()))
    assert np.all(np.linalg.norm(poses[:, :2] - log.odom[-1, :2], axis=1) < 2.0)
    c = pf.engine.counters()
    assert c["window_fallbacks"] == 0
    xs, ys = pf.particles[0]._map.get_occupied_points()
    assert len(xs) > 300                                   # walls seen at least twice (one hit is +0.8, the threshold 1.0)
    pf.close()


def test_dropin_objects_run_the_reference_loop_body():
    """thesis_amd.dropin: the statements of main.py:138-168 that touch particles - the two list comprehensions over
    Robot objects, resample(particles), get_latest_pose(), from_global_reference, _map.get_occupied_points() - run over
    the facade and give exactly what run_log gives over the batched ParticleFilter with the same seeds."""
    from math import pi, sqrt
    from thesis_amd import dropin
    from thesis_amd.datasets import synthetic
    from thesis_amd.slam import ParticleFilter, run_log
    NUM_PARTICLES, B, T = 24, 361, 14
    angles, ranges, odo, truth = synthetic.make_log(T, B, period=0.7)
    scan_times = (np.arange(T + 1) * 7000).astype(np.int64)
    odom_times = scan_times[1:] - 1

    class Reading:                                    # models.py:44-77, as much as Robot.imu_update uses
        def __init__(self, data, ts):
            self._data, self._ts, self._dt = data, ts, 0.0

        def get_data(self): return self._data
        def timestamp(self): return self._ts
        def set_dt(self, dt): self._dt = dt
        def dt(self): return self._dt

    dropin.configure(motion_model="velocity", cell_size=0.05, max_beams=B, seed=42)
    particles = [dropin.Robot(None) for _ in range(NUM_PARTICLES)]                    # main.py:87
    urng = np.random.Generator(np.random.PCG64(42))                                 # the uniforms ParticleFilter(seed=42).resample() draws
    imu = [Reading(odo[k], int(odom_times[k])) for k in range(T)]
    lidar = [dropin.Scan(ranges[k], angles, int(scan_times[k])) for k in range(T + 1)]
    prev_timestamp = imu[0].timestamp()                                               # main.py:102
    imu_idx = lidar_idx = 0
    plotFrameNumber = 1550                                                            # main.py:110
    last_updated_pose = particles[0].get_latest_pose()
    last_scan = lidar[0].from_global_reference(last_updated_pose)
    update_count = 0
    times = np.unique(np.concatenate((odom_times, scan_times)))                       # main.py:114
    for t in times:                                                                   # main.py:138-168
        imu_reading = imu[imu_idx]
        if imu_reading.timestamp() == t:
            imu_idx = min(imu_idx + 1, len(imu) - 1)
            dt = imu_reading.timestamp() - prev_timestamp
            imu_reading.set_dt(dt)
            [p.imu_update(imu_reading) for p in particles]
            prev_timestamp = imu_reading.timestamp()
        if int(scan_times[lidar_idx]) == t:
            lidar_reading = lidar[lidar_idx]
            lidar_idx = min(lidar_idx + 1, len(lidar) - 1)
            curr_pose = particles[0].get_latest_pose()
            dist = sqrt((last_updated_pose.x() - curr_pose.x()) ** 2 + (last_updated_pose.y() - curr_pose.y()) ** 2)
            rot = abs(last_updated_pose.theta() - curr_pose.theta())
            if update_count < 2 or (dist >= 0.33 or rot >= pi / 9):
                if plotFrameNumber % 5 < 2:
                    [p.map_update(lidar_reading, last_scan, False) for p in particles]
                else:
                    [p.map_update(lidar_reading, last_scan, True) for p in particles]
                particles = dropin.resample(particles, u=float(urng.random()))
                if dist >= 0.33 or rot >= pi / 9:
                    update_count = 0
                    last_updated_pose = curr_pose
                elif update_count < 2:
                    update_count += 1
                if plotFrameNumber % 5 == 0:
                    last_scan = lidar_reading.from_global_reference(particles[0].get_latest_pose())
            plotFrameNumber += 1
    plot_x, plot_y = particles[0]._map.get_occupied_points()
    eng_a = particles[0]._s.filter().engine
    # the same run through the batched loop
    pf = ParticleFilter(NUM_PARTICLES, angles, motion_model="velocity", cell_size=0.05, seed=42)
    res = run_log(pf, ranges, scan_times, odo, odom_times)
    assert res.accepted >= 10
    try:
        np.testing.assert_array_equal(eng_a.poses(), pf.engine.poses())
        np.testing.assert_array_equal(eng_a.weights(), pf.engine.weights())
        np.testing.assert_array_equal(eng_a.covs(), pf.engine.covs())
        for p in (0, NUM_PARTICLES - 1):
            for (ca, ta), (cb, tb) in zip(eng_a.tiles(p), pf.engine.tiles(p)):
                assert ca == cb and np.array_equal(ta, tb)
        bx, by = pf.particles[0]._map.get_occupied_points()
        assert np.array_equal(plot_x, bx) and np.array_equal(plot_y, by) and len(plot_x) > 500
        # the rest of the per-object surface
        r0 = particles[0]
        assert len(r0.x()) == len(r0.y()) == len(r0.theta()) > T and isinstance(r0.weight(), list)
        m = r0._map
        cell = m.get_cell(1.0, -2.0)
        assert (cell.x, cell.y) == (int(1.0 / 40 * 800 + 400), int(-2.0 / 40 * 800 + 400)) and m.get_cell(500.0, 0.0) is None
        assert m.index_to_distance(400) == 0.0 and abs(m.index_to_distance(401) - 0.05) < 1e-12
        near = m.get_nearby_occ_points(m.get_cell(7.9, 0.0))
        assert len(near) > 10 and all(abs(x) < 10 and abs(y) < 10 for x, y in near)
        snap = r0.copy()
        assert snap._map.keys() == {(0.0, 0.0)} and snap._x == r0.x()
        assert str(r0).startswith("Robot at position: Pose: (")
    finally:
        pf.close()
        dropin.configure()

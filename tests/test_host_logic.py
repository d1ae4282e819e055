"""Host-side logic on CPU: CARMEN reader, scan geometry helper, synthetic workload, loop gating constants."""
import os

import numpy as np

from oracle import rbpf_oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))


def test_carmen_reader_intel_head():
    from thesis_amd.datasets.carmen import load_carmen
    log = load_carmen(os.path.join(HERE, "golden", "intel_head.log"))
    assert log.scans.shape == (72, 180) and log.odom.shape[1] == 3          # IntelLidarData.py:8 POINTS_PER_SCAN = 180
    np.testing.assert_allclose(log.angles[[0, -1]], [-np.pi / 2, np.pi / 2])  # IntelLidarData.py:19
    # first FLASER record of data/intel.txt: ranges 1.09 1.08 1.08 ..., time field 32.9068 -> int(10*t)*10
    np.testing.assert_allclose(log.scans[0, :3], [1.09, 1.08, 1.08])
    assert log.scan_times[0] == int(10 * 32.9068) * 10
    assert len(log.order) == len(log.scans) + len(log.odom)
    assert sorted(log.order[log.order >= 0].tolist()) == list(range(72))
    assert sorted((-log.order[log.order < 0] - 1).tolist()) == list(range(len(log.odom)))


def test_scan_to_global_matches_reference_transform():
    from thesis_amd.slam import scan_to_global
    rng = np.random.Generator(np.random.PCG64(0))
    r = rng.uniform(0.5, 9.0, 181); a = np.linspace(-np.pi / 2, np.pi / 2, 181)
    pose = (1.25, -0.5, 0.7)
    xy = scan_to_global(r, a, np.array(pose))
    sx, sy = orc.scan_xy(r, a)
    gx, gy = orc.transform(sx, sy, pose)                                     # lidar.py:111-128
    np.testing.assert_allclose(xy[:, 0], gx, rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(xy[:, 1], gy, rtol=1e-13, atol=1e-13)


def test_synthetic_room16_geometry():
    from thesis_amd.datasets import synthetic
    ang = synthetic.beam_angles(1081)
    assert len(ang) == 1081 and np.isclose(ang[0], -0.75 * np.pi) and np.isclose(ang[-1], 0.75 * np.pi)
    r = synthetic.cast_scan((0.0, 0.0, 0.0), ang, None)
    assert np.isclose(r[540], 8.0)                                           # straight ahead: the wall at x = 8
    assert r.min() >= 3.5 / np.cos(np.pi / 4) - 1e-9                         # nearest pillar corner (3.5, 3.5)
    a2, ranges, odo, poses = synthetic.make_log(5, 1081, period=0.7)
    assert ranges.shape == (6, 1081) and odo.shape == (5, 3) and poses.shape == (6, 3)
    np.testing.assert_allclose(np.linalg.norm(np.diff(poses[:, :2], axis=0), axis=1), 0.35, atol=2e-3)   # >= main.py:43 threshold


def test_loop_constants_match_reference():
    from thesis_amd import slam
    assert (slam.MAX_UPDATE_COUNT, slam.DIST_THRESHOLD) == (2, 0.33) and np.isclose(slam.ROT_THRESHOLD, np.pi / 9)   # main.py:41-43

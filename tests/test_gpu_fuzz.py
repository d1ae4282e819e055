"""Seeded slices of the developer fuzz tools (tools/fuzz_*.py) inside the GPU suite, and the driver loop's decisions on the
real engine against the loop oracle.  Fixed seeds: a failure reproduces; the tools dump the inputs of a failing case under
gpurun_out/.  GPU only:  python -m pytest tests -m gpu"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))


@pytest.mark.parametrize("seed,kernels", [(101, ["auto"]), (102, ["auto", "ray", "window"]), (103, ["auto", "auto", "window"]), (104, ["auto", "ray"])])
def test_fuzz_slice_map_update_and_sample_weights(seed, kernels, monkeypatch):
    """300 random cases per seed (1200 in all): cell sizes 0.025 / 0.05 / 0.1, 1 to 1500 beams, four fields of view, ranges
    including > 15 m and ~0, poses at tile edges and on the irregular negative side, one to three scans, every map
    kernel selection - every map cell and every sample weight against the C oracle (hybridmap.py:95-145, robot.py:118-139)."""
    monkeypatch.delenv("FUZZ_KERNEL", raising=False)
    import fuzz_map_update
    assert fuzz_map_update.run(300, seed, kernels=kernels, verbose=False) == 0


def test_fuzz_slice_resample_ancestors():
    """3000 random weight vectors (flat, peaked, with -inf and zeros; 1 to 5000 particles): ancestors bit for bit (main.py:46-79)."""
    import fuzz_resample
    assert fuzz_resample.run(3000, 201, verbose=False) == 0


def test_fuzz_slice_ndt_stage_equals_its_oracle():
    """800 random polygonal rooms, offsets and guesses: the HIP NDT stage reproduces oracle/matcher_oracle.py (pose, score,
    evaluation count).  The oracle is the builder's statement of NDT (matchScanCustom.m:32-44): parity with MATLAB unpinned."""
    import fuzz_ndt
    ran, worst_p, worst_s, diff_ev = fuzz_ndt.run(800, 301, verbose=False)
    assert ran >= 240 and diff_ev == 0 and worst_p < 1e-6 and worst_s < 1e-6


def test_fuzz_slice_match_field_staging():
    """400 random poses round tile edges, corners and the irregular negative side, cell sizes 0.05 / 0.025: the matcher's
    funnel-shift field staging gives the bits of the bit-by-bit form (hybridmap.py:210-242 is where the field comes from)."""
    import fuzz_match_staging
    assert fuzz_match_staging.run(400, 401, verbose=False) == 0


def test_run_log_decisions_on_the_real_engine_equal_the_reference_loop():
    """main.py:138-168 on the GPU: a ParticleFilter (the real engine: matcher, proposal, map update, resampling) is driven
    through 60 scans of the synthetic room by thesis_amd.slam.run_log; its decisions - which record at which merged time
    with which dt, which scans pass the motion gate of particle 0, which are matched against the previous scan, when
    last_scan is refreshed - must equal oracle/loop_oracle.py replaying the same timestamps with the poses particle 0
    really had.  (tests/test_host_logic.py does the same with a fake filter on the CPU.)"""
    from oracle import loop_oracle
    from thesis_amd.datasets import synthetic
    from thesis_amd.slam import ParticleFilter, run_log
    n = 60
    angles, ranges, odo, truth = synthetic.make_log(n, 361, period=0.31)          # 0.155 m per scan: the gate rejects some
    scan_times = (np.arange(n + 1) * 3100).astype(np.int64)
    odom_times = scan_times[1:] - 7
    pf = ParticleFilter(64, angles, "velocity", keep_history=False, seed=7)
    res = run_log(pf, ranges, scan_times, odo, odom_times)
    want = loop_oracle.replay_decisions(odom_times, scan_times, lambda k: res.pose0_before[k], start_frame=0)
    assert res.trace == want
    scans = [ev for ev in want if ev[0] == "scan"]
    assert len(scans) == n + 1 and any(not ev[3] for ev in scans) and any(ev[3] and ev[4] for ev in scans) and any(ev[5] for ev in scans)
    assert res.accepted == sum(1 for ev in scans if ev[3])
    # the run itself is sane: finite state, particle 0 near the simulated truth
    poses = pf.engine.poses()
    assert np.all(np.isfinite(poses)) and np.all(np.isfinite(pf.engine.weights()))
    assert np.hypot(*(np.median(poses[:, :2], axis=0) - truth[-1][:2])) < 0.6
    pf.close()

"""Sharded particles on the GPU: two ranks sharing the one GPU of the test box (gloo transport, host-staged)
must reproduce a single engine holding all particles bit for bit: proposal streams are keyed by the global
particle id, ancestors come from the same kernel, migrated maps travel as packed tiles."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _scenario(scans=6):
    from thesis_amd.datasets import synthetic
    return synthetic.make_log(scans, 361, period=0.7)


def _run_steps(engine, shard, log, steps, p_total):
    angles, ranges, odo, _ = log
    rng = np.random.Generator(np.random.PCG64(9))
    engine.set_scan(ranges[0], angles)
    engine.map_update(np.zeros((engine.P, 3)))
    out = []
    for k in range(steps):
        engine.imu_update("velocity", odo[k], 7000.0)
        engine.set_scan(ranges[k + 1], angles)
        engine.scan_update(adj=False)
        u = float(rng.random())
        if shard is None:
            did, idx = engine.resample(u)
        else:
            did, idx = shard.resample(u)
        out.append((did, None if idx is None else np.array(idx)))
    return out


def _worker(rank, world, p_local, port, steps, q):
    keep = _fault_log(f"rank{rank}")                        # noqa: F841
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from thesis_amd.engine import ParticleEngine
    from thesis_amd.sharding import ShardedResampler
    e = ParticleEngine(p_local, max_beams=361, pool_tiles=4 * p_local, seed=42)
    sr = ShardedResampler(rank, world, p_local, device=0, dist=dist)
    sr.attach(e)
    hist = _run_steps(e, sr, _scenario(steps + 3), steps, world * p_local)
    gids = np.nonzero(sr.owner == rank)[0]
    order = np.argsort(sr.local_of[gids])
    gids = gids[order]
    tiles = [{c: cells.copy() for c, cells in e.tiles(i)} for i in range(p_local)]
    q.put((rank, gids, e.poses(), e.covs(), e.weights(), tiles, hist, sr.stats))
    dist.barrier()
    sr.close()                               # the borrowed torch stream goes back before the engine is destroyed
    e.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,p_local,steps", [(2, 24, 3), (3, 40, 12)])
def test_ranks_on_one_gpu_equal_one_engine(world, p_local, steps):
    import torch.multiprocessing as mp
    from thesis_amd.engine import ParticleEngine
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, p_local, port, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    one = ParticleEngine(world * p_local, max_beams=361, pool_tiles=4 * world * p_local, seed=42)
    hist = _run_steps(one, None, _scenario(steps + 3), steps, world * p_local)
    poses, covs, w = one.poses(), one.covs(), one.weights()
    assert any(h[0] for h in hist)                                   # the scenario does resample
    moved = sum(r[7]["moved"] for r in results) // world
    for rank, gids, p_r, c_r, w_r, tiles_r, hist_r, stats in results:
        for k in range(steps):
            assert hist_r[k][0] == hist[k][0]
            if hist[k][0]:
                assert np.array_equal(hist_r[k][1], hist[k][1])      # same ancestors as the single engine
        np.testing.assert_array_equal(p_r, poses[gids])
        np.testing.assert_array_equal(c_r, covs[gids])
        np.testing.assert_array_equal(w_r, w[gids])
        for i, g in enumerate(gids):
            ref = dict(one.tiles(int(g)))
            assert set(ref) == set(tiles_r[i])
            for c in ref:
                assert np.array_equal(ref[c], tiles_r[i][c]), (rank, i, g, c)
    one.close()
    assert moved >= 0 and (steps < 10 or moved > 0)           # the longer run does migrate particles


class _OneRankDist:
    """torch.distributed stand-in for a single rank: the collectives are identities, the backend reads as RCCL so that
    the device path (and the early, overlapped resample) is the one under test."""
    class ReduceOp:
        MAX = "max"

    def get_backend(self):
        return "nccl"

    def all_reduce(self, t, op=None):
        return None

    def broadcast(self, t, src=0):
        return None

    def all_to_all_single(self, recv, send, in_splits, out_splits):
        recv.copy_(send)


def _fault_log(tag):
    """A fatal signal in a child (also one at interpreter exit) leaves its Python stack in gpurun_out/; a child that
    ends normally removes the (empty) file from an atexit hook, i.e. after the engines' own teardown hook."""
    import atexit
    import faulthandler
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, f"child_fault_{tag}_{os.getpid()}.log")
        f = open(path, "w")
        faulthandler.enable(f, all_threads=True)

        def tidy():
            try:
                if os.path.getsize(path) == 0:
                    os.remove(path)
            except OSError:
                pass
        atexit.register(tidy)          # registered before thesis_amd.engine is imported: runs after its hook (LIFO)
        return f
    except OSError:
        faulthandler.enable()
        return None


def _early_vs_plain_guarded(nan_particle, q):
    """The child process: whatever goes wrong comes back to the parent as text; the process then ends the ordinary
    way (interpreter teardown included), and the parent checks its exit code."""
    import traceback
    keep = _fault_log("early")                              # noqa: F841 - the file stays open until the process ends
    try:
        _early_vs_plain(nan_particle, q)
    except BaseException:                                   # noqa: BLE001 - the parent prints it
        q.put("error in child:\n" + traceback.format_exc())
        raise


def _early_vs_plain(nan_particle, q):
    """scan_update_begin -> resample_begin -> scan_update_end -> resample_finish (weights exported, all-reduced and
    turned into ancestors on a side stream while the map update runs) gives exactly the state of scan_update +
    resample on a plain engine; with a particle on the NaN-covariance branch (robot.py:73-78) the early result is
    discarded and the late path runs."""
    import torch
    torch.cuda.init()                      # torch's device context first, as under torchrun (bench.py)
    from thesis_amd import engine as eng, sharding
    from thesis_amd.datasets import synthetic
    P, B = 48, 1081
    ang, ranges, odo, _ = synthetic.make_log(5, B, period=0.7)
    a = eng.ParticleEngine(P, max_beams=B, pool_tiles=4 * P, seed=11)
    b = eng.ParticleEngine(P, max_beams=B, pool_tiles=4 * P, seed=11)
    rs = sharding.ShardedResampler(0, 1, P, device=0, dist=_OneRankDist())
    rs.attach(a)
    for e in (a, b):
        e.set_scan(ranges[0], ang)
        e.map_update(np.zeros((P, 3)))
    urng = np.random.Generator(np.random.PCG64(3))
    for k in range(4):
        u = float(urng.random())
        mo = None
        if nan_particle and k == 2:          # the matcher's failure signal for one particle: NaN covariance, score 0
            mo = np.zeros((P, 13)); mo[:, :3] = [0.35 * (k + 1), 0.0, 0.0]; mo[:, 3] = mo[:, 7] = 1e-4; mo[:, 11] = 1e-5; mo[:, 12] = 100.0
            mo[5, 3:12] = np.nan; mo[5, 12] = 0.0
        for e in (a, b):
            e.imu_update("velocity", odo[k], 7000.0)
            e.set_scan(ranges[k + 1], ang)
        a.scan_update_begin(adj=False, match_override=mo)
        rs.resample_begin(u)
        a.scan_update_end()
        did_a, idx_a = rs.resample_finish()
        b.scan_update(adj=False, match_override=mo)
        did_b, idx_b = b.resample(u)
        assert did_a == did_b
        if did_a:
            assert np.array_equal(idx_a, idx_b)
    assert rs.stats.get("late", 0) == (1 if nan_particle else 0)
    assert np.array_equal(a.poses(), b.poses()) and np.array_equal(a.weights(), b.weights()) and np.array_equal(a.covs(), b.covs())
    for p in (0, 5, P - 1):
        for (ca, ta), (cb, tb) in zip(a.tiles(p), b.tiles(p)):
            assert ca == cb and np.array_equal(ta, tb)
    rs.close()                              # drops the early tensors, gives torch's stream back
    a.close(); b.close()
    q.put("ok")


@pytest.mark.parametrize("nan_particle", [False, True])
def test_early_overlapped_resample_equals_the_plain_engine(nan_particle):
    """Runs in a fresh process: torch's device context is created there, next to the engine's."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pr = ctx.Process(target=_early_vs_plain_guarded, args=(nan_particle, q))
    pr.start()
    try:
        msg = q.get(timeout=300)
    except Exception:                                       # noqa: BLE001
        msg = "no message from the child within 300 s"
    pr.join(60)
    assert msg == "ok" and pr.exitcode == 0, (msg, pr.exitcode)


def test_engine_leaves_current_device_and_borrowed_stream_alone():
    """rbpf_create and every later call restore the caller's current device; a borrowed torch stream is released
    (not destroyed) by close(), and torch can go on using it."""
    import torch
    from thesis_amd.engine import ParticleEngine
    from thesis_amd.datasets import synthetic
    torch.cuda.init()
    dev0 = torch.cuda.current_device()
    side = torch.cuda.Stream()
    e = ParticleEngine(8, max_beams=181, pool_tiles=32)
    assert torch.cuda.current_device() == dev0
    ang = synthetic.beam_angles(181, np.pi)
    r = synthetic.cast_scan((0.0, 0.0, 0.0), ang, np.random.Generator(np.random.PCG64(0)))
    e.set_stream(side.cuda_stream)
    e.set_scan(r, ang)
    e.map_update(np.zeros((8, 3)))
    assert torch.cuda.current_device() == dev0
    e.release_stream()
    e.map_update(np.zeros((8, 3)))                       # works on a stream of its own again
    e.set_stream(side.cuda_stream)
    e.close()                                            # releases the stream first
    with torch.cuda.stream(side):                        # the stream is still torch's and still works
        t = torch.ones(1024, device="cuda") * 2
    side.synchronize()
    assert float(t.sum().item()) == 2048.0

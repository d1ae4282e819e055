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


def _scenario():
    from thesis_amd.datasets import synthetic
    return synthetic.make_log(6, 361, period=0.7)


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
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from thesis_amd.engine import ParticleEngine
    from thesis_amd.sharding import ShardedResampler
    e = ParticleEngine(p_local, max_beams=361, pool_tiles=4 * p_local, seed=42)
    sr = ShardedResampler(rank, world, p_local, device=0, dist=dist)
    sr.attach(e)
    hist = _run_steps(e, sr, _scenario(), steps, world * p_local)
    gids = np.nonzero(sr.owner == rank)[0]
    order = np.argsort(sr.local_of[gids])
    gids = gids[order]
    tiles = [{c: cells.copy() for c, cells in e.tiles(i)} for i in range(p_local)]
    q.put((rank, gids, e.poses(), e.covs(), e.weights(), tiles, hist, sr.stats))
    dist.barrier()
    e.close()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_one_engine():
    import torch.multiprocessing as mp
    from thesis_amd.engine import ParticleEngine
    world, p_local, steps = 2, 24, 3
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
    hist = _run_steps(one, None, _scenario(), steps, world * p_local)
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
    assert moved >= 0

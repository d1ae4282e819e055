"""The N > 1 path on CPU: migration plan properties and a world_size-2 gloo run of ShardedResampler against a
numpy particle store (test double), checked against the oracle's single-process resample (main.py:46-79)."""
import os
import socket

import numpy as np
import pytest

from oracle import rbpf_oracle as orc
from thesis_amd.sharding import plan_migration


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("world,p_local,seed", [(2, 8, 0), (4, 16, 1), (8, 32, 2), (8, 4, 3)])
def test_plan_migration_properties(world, p_local, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    n = world * p_local
    # a previous, arbitrary placement
    perm = rng.permutation(n)
    owner = np.empty(n, dtype=np.int32); local_of = np.empty(n, dtype=np.int32)
    owner[perm] = np.arange(n) // p_local
    local_of[perm] = np.arange(n) % p_local
    w = np.exp(rng.normal(0, 3, n)) * 1000
    did, idx = orc.resample_indices(list(w), float(rng.random()))
    assert did
    idx = np.array(idx)
    plan = plan_migration(idx, owner, local_of, world, p_local)
    assert np.array_equal(np.bincount(plan.dest, minlength=world), np.full(world, p_local))
    n_src = np.bincount(owner[idx], minlength=world)
    assert plan.n_move == int(np.maximum(n_src - p_local, 0).sum())          # only the surplus migrates
    seen = np.zeros(n, dtype=bool)
    for r in range(world):
        gid, src = plan.new_gid[r], plan.new_src[r]
        assert len(gid) == p_local and not seen[gid].any()
        seen[gid] = True
        kept = src >= 0
        assert np.all(np.diff(src[kept]) >= 0) and not kept[np.argmax(~kept):].any() if (~kept).any() else True
        assert np.array_equal(local_of[idx[gid[kept]]], src[kept]) and np.all(owner[idx[gid[kept]]] == r)
        arrivals = gid[~kept]
        expected = np.concatenate([gid_q for gid_q in ([np.sort(np.nonzero((plan.src_rank == q) & (plan.dest == r))[0])
                                                         for q in range(world) if q != r])]) if (~kept).any() else arrivals
        assert np.array_equal(arrivals, expected)                            # arrival order = all-to-all order
        for q in range(world):
            if q != r:
                js = np.sort(np.nonzero((plan.src_rank == q) & (plan.dest == r))[0])
                assert np.array_equal(plan.send[q][r], local_of[idx[js]])
    assert seen.all()


class NumpyShard:
    """Test double of EngineShard: particles are (13 doubles, 48-byte map tag) records in numpy."""
    meta_width = 4

    def __init__(self, states, tags):
        import torch
        self.torch = torch
        self.states, self.tags = [s.copy() for s in states], [t.copy() for t in tags]
        self.gids = None

    def set_global_ids(self, ids):
        self.gids = np.array(ids)

    def weights_global(self, n):
        t = self.torch.zeros(n, dtype=self.torch.float64)
        t[self.torch.from_numpy(self.gids.astype(np.int64))] = self.torch.tensor([s[12] for s in self.states], dtype=self.torch.float64)
        return t

    def indices(self, w, u):
        did, idx = orc.resample_indices(list(w.numpy()), u)
        return did, np.array(idx, dtype=np.int32)

    def pack(self, local_idx):
        n = len(local_idx)
        meta = np.zeros((n, 4), dtype=np.int32)
        chunks = []
        for i, li in enumerate(local_idx):
            rec = np.zeros(176, dtype=np.uint8)
            rec[:104] = np.frombuffer(self.states[li].tobytes(), dtype=np.uint8)
            rec[128:176] = self.tags[li]
            meta[i] = [1, 176 // 16, 0, 0]
            chunks.append(rec)
        pay = self.torch.from_numpy(np.concatenate(chunks)) if chunks else self.torch.empty(0, dtype=self.torch.uint8)
        return meta, pay

    def empty_payload(self, n):
        return self.torch.empty(n, dtype=self.torch.uint8)

    def apply_local(self, new_src, new_gid):
        ns, nt = [], []
        for s in new_src:
            if s >= 0:
                st = self.states[s].copy(); st[12] = 1.0
                ns.append(st); nt.append(self.tags[s].copy())
            else:
                ns.append(None); nt.append(None)
        self.states, self.tags, self.gids = ns, nt, np.array(new_gid)

    def unpack(self, local_idx, meta, payload):
        buf = payload.numpy()
        off = 0
        for i, li in enumerate(local_idx):
            st = np.frombuffer(buf[off:off + 104].tobytes(), dtype=np.float64).copy(); st[12] = 1.0
            self.states[li] = st; self.tags[li] = buf[off + 128:off + 176].copy()
            off += int(meta[i][1]) * 16

    def pose(self, i):
        return self.states[i][:3]


class NumpyShardRaw(NumpyShard):
    """The same double with the three-step migration protocol of EngineShard (records exchanged before the pack:
    rbpf_gather_pack_meta / rbpf_meta_from_raw / rbpf_pack_particles_raw), which ShardedResampler prefers."""
    raw_width = 5

    def gather_pack_meta(self, local_idx):
        rec = np.array([[int(li) + 1000, 176 // 16, 7, 8, 9] for li in local_idx], dtype=np.int32).reshape(-1)
        return self.torch.from_numpy(rec) if len(rec) else self.torch.empty(0, dtype=self.torch.int32)

    def meta_from_raw(self, raw, n):
        raw = np.asarray(raw, dtype=np.int32).reshape(n, 5)
        assert n == 0 or (np.all(raw[:, 2:] == [7, 8, 9]) and np.all(raw[:, 0] >= 1000))
        meta = np.zeros((n, 4), dtype=np.int32)
        meta[:, 0] = 1
        meta[:, 1] = raw[:, 1]
        return meta

    def pack_raw(self, local_idx, raw, nbytes):
        assert [int(x) - 1000 for x in np.asarray(raw).reshape(-1, 5)[:, 0]] == [int(x) for x in local_idx]
        _, pay = self.pack(local_idx)
        assert pay.numel() == nbytes
        return pay


def _worker(rank, world, p_local, port, rounds, out_q, flavour="pack"):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from thesis_amd.sharding import ShardedResampler
    n = world * p_local
    rng = np.random.Generator(np.random.PCG64(42))
    states = rng.normal(size=(n, 13)); tags = rng.integers(0, 255, size=(n, 48)).astype(np.uint8)
    sl = slice(rank * p_local, (rank + 1) * p_local)
    shard = (NumpyShardRaw if flavour == "raw" else NumpyShard)(list(states[sl]), list(tags[sl]))
    sr = ShardedResampler(rank, world, p_local)
    sr.attach(shard)
    history = []
    for k in range(rounds):
        wts = np.exp(rng.normal(0, 3, n)) * 1000           # same on every rank: new weights per global id
        for i, g in enumerate(shard.gids):
            shard.states[i][12] = wts[g]
        did, idx = sr.resample(0.37 + 0.1 * k)
        history.append((did, None if idx is None else idx.copy()))
        p0 = sr.pose_of_particle0()
    out_q.put((rank, shard.gids.copy(), np.array(shard.states), np.array(shard.tags), history, p0, sr.stats))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,p_local,flavour", [(2, 8, "pack"), (2, 33, "raw"), (4, 12, "raw"), (8, 5, "raw"), (8, 5, "pack")])
def test_sharded_resample_gloo_matches_single_process(world, p_local, flavour):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    rounds = 3
    procs = [ctx.Process(target=_worker, args=(r, world, p_local, port, rounds, q, flavour)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process expectation
    n = world * p_local
    rng = np.random.Generator(np.random.PCG64(42))
    states = rng.normal(size=(n, 13)); tags = rng.integers(0, 255, size=(n, 48)).astype(np.uint8)
    for k in range(rounds):
        wts = np.exp(rng.normal(0, 3, n)) * 1000
        did, idx = orc.resample_indices(list(wts), 0.37 + 0.1 * k)
        assert did
        idx = np.array(idx)
        states, tags = states[idx].copy(), tags[idx].copy()
        states[:, 12] = 1.0
        for res in results:
            assert res[4][k][0] and np.array_equal(res[4][k][1], idx)          # same ancestors on every rank
    got_s, got_t = np.empty_like(states), np.empty_like(tags)
    seen = np.zeros(n, dtype=bool)
    for rank, gids, st, tg, hist, p0, stats in results:
        got_s[gids] = st; got_t[gids] = tg; seen[gids] = True
        np.testing.assert_array_equal(p0, states[0, :3])
    assert seen.all()
    np.testing.assert_array_equal(got_s, states)
    np.testing.assert_array_equal(got_t, tags)


def _plan_reference(idx, owner, local_of, world, p_local):
    """The migration rule written out with explicit loops (the documented rule of plan_migration)."""
    idx = np.asarray(idx, dtype=np.int64)
    n = len(idx)
    src_rank = owner[idx]
    dest = np.full(n, -1, dtype=np.int32)
    counts, surplus = np.zeros(world, dtype=np.int64), []
    for r in range(world):
        js = np.nonzero(src_rank == r)[0]
        dest[js[:p_local]] = r
        counts[r] = len(js[:p_local])
        surplus.append(js[p_local:])
    pool, k = np.concatenate(surplus), 0
    for d in range(world):
        need = int(p_local - counts[d])
        dest[pool[k:k + need]] = d
        k += need
    new_gid, new_src, send = [], [], [[None] * world for _ in range(world)]
    for r in range(world):
        js = np.nonzero(dest == r)[0]
        kept = js[src_rank[js] == r]
        anc = local_of[idx[kept]]
        o = np.lexsort((kept, anc))
        arr = js[src_rank[js] != r]
        arr = arr[np.lexsort((arr, src_rank[arr]))]
        new_gid.append(np.concatenate([kept[o], arr])); new_src.append(np.concatenate([anc[o], np.full(len(arr), -1)]))
        for d in range(world):
            send[r][d] = local_of[idx[np.nonzero((src_rank == r) & (dest == d) & (r != d))[0]]]
    return dest, new_gid, new_src, send


@pytest.mark.parametrize("world,p_local,seed", [(2, 8, 0), (4, 16, 1), (8, 64, 2), (8, 1024, 3), (3, 5, 4)])
def test_plan_migration_matches_the_written_out_rule(world, p_local, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    n = world * p_local
    perm = rng.permutation(n)                                    # a scrambled ownership, as after earlier resamples
    owner = (perm // p_local).astype(np.int32)
    local_of = (perm % p_local).astype(np.int32)
    for trial in range(4):
        w = rng.random(n) ** (1 + 3 * trial)
        idx = np.sort(rng.choice(n, size=n, p=w / w.sum()))      # non-decreasing ancestors, like systematic resampling
        plan = plan_migration(idx, owner, local_of, world, p_local)
        dest, gid, src, send = _plan_reference(idx, owner, local_of, world, p_local)
        assert np.array_equal(plan.dest, dest) and plan.n_move == int((dest != owner[idx]).sum())
        for r in range(world):
            assert np.array_equal(plan.new_gid[r], gid[r]) and np.array_equal(plan.new_src[r], src[r])
            for d in range(world):
                assert np.array_equal(plan.send[r][d], send[r][d])

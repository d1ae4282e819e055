"""Particles sharded over the GPUs of one node: the global resample step (main.py:46-79).

Everything between two resamples is independent per particle (main.py:144,157-159 are maps over the particle
list), so each rank runs its own ParticleEngine on its own particles with no communication.  `resample` couples
them:

  1. every rank scatters its weights into a zeroed vector at the particles' global ids; ONE all-reduce (RCCL over
     xGMI) gives every rank the full weight vector;
  2. every rank computes the same ancestor indices from it (same kernel, same uniform u);
  3. `plan_migration` (pure numpy, identical on every rank) keeps each new particle on the rank that holds its
     ancestor while that rank has room; only the surplus migrates (state + the written boxes of its tiles), with
     two all-to-alls (fixed-width metadata, then payload);
  4. locally, first occurrences keep their map slot and duplicates are tile copies, exactly as on one GPU.

Which rank holds a particle never changes a result: proposal streams are keyed by the global particle id.
The module is transport-agnostic: with the "gloo" backend (CPU tests) tensors are staged through the host.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import os
import time

import numpy as np


@dataclass
class Plan:
    dest: np.ndarray          # [n] rank that holds new global particle j
    src_rank: np.ndarray      # [n] rank that holds its ancestor
    new_gid: List[np.ndarray]     # per rank: global ids in the rank's new local order
    new_src: List[np.ndarray]     # per rank: old local index of the ancestor, -1 for arrivals (last)
    send: List[List[np.ndarray]]  # send[r][d]: old local indices rank r packs for rank d (in global-id order)
    n_move: int


def plan_migration(idx: np.ndarray, owner: np.ndarray, local_of: np.ndarray, world: int, p_local: int,
                   rank: Optional[int] = None) -> Plan:
    """idx[j] = old global id continued by new global particle j; owner/local_of = rank and local index of
    every old global particle.  Deterministic; every rank computes the same plan.

    Rule: a new particle stays on its ancestor's rank while that rank has room (first come in global order); the
    surplus, taken rank by rank, fills the ranks that are short, lowest rank first.  Inside a rank the survivors
    come first, ordered by their ancestor's old local index, then the arrivals, ordered by source rank; ties by
    global id.

    `rank` = r restricts the per-rank parts (new_gid, new_src, send rows) to what rank r itself needs: its own new
    order, what it sends, and how many particles each rank sends to it (send[q][r] keeps only the right length).
    This is what runs on every rank at every resample; it needs `local_of` only for the particles rank r owns."""
    idx = np.asarray(idx, dtype=np.int64)
    n = len(idx)
    src_rank = np.asarray(owner)[idx].astype(np.int16)
    by_rank = np.argsort(src_rank, kind="stable")                # grouped by ancestor rank, global order inside
    cnt = np.bincount(src_rank, minlength=world)
    first = np.concatenate(([0], np.cumsum(cnt)[:-1]))
    pos = np.empty(n, dtype=np.int64)
    pos[by_rank] = np.arange(n) - first[src_rank[by_rank]]       # position among the children of the same rank
    keep = pos < p_local
    dest = np.where(keep, src_rank, -1).astype(np.int16)
    pool = by_rank[~keep[by_rank]]                               # surplus: rank by rank, global order inside
    need = p_local - np.minimum(cnt, p_local)
    assert int(need.sum()) == len(pool)
    dest[pool] = np.repeat(np.arange(world), need)
    movers = np.nonzero(dest != src_rank)[0]                     # global order
    pair = src_rank[movers].astype(np.int64) * world + dest[movers]
    n_pair = np.bincount(pair, minlength=world * world)
    empty = np.empty(0, dtype=np.int32)
    new_gid, new_src = [empty] * world, [empty] * world
    send = [[empty] * world for _ in range(world)]
    local_of = np.asarray(local_of)
    for r in (range(world) if rank is None else (rank,)):
        js = np.nonzero(dest == r)[0]
        arr = src_rank[js] != r
        kept, came = js[~arr], js[arr]
        anc = local_of[idx[kept]]
        ko = np.lexsort((kept, anc))
        co = np.lexsort((came, src_rank[came]))
        new_gid[r] = np.concatenate([kept[ko], came[co]]).astype(np.int32)
        new_src[r] = np.concatenate([anc[ko], np.full(len(came), -1)]).astype(np.int32)
        mine = movers[src_rank[movers] == r]                     # what rank r sends, by destination then global id
        o = np.argsort(dest[mine], kind="stable")
        parts = np.split(local_of[idx[mine[o]]].astype(np.int32), np.cumsum(n_pair[r * world:(r + 1) * world])[:-1])
        send[r] = list(parts)
    if rank is not None:                                          # lengths of what arrives at `rank`
        for q in range(world):
            if q != rank:
                send[q][rank] = np.zeros(int(n_pair[q * world + rank]), dtype=np.int32)
    return Plan(dest.astype(np.int32), src_rank.astype(np.int32), new_gid, new_src, send, int(len(movers)))


class EngineShard:
    """Adapter between ShardedResampler and one rank's ParticleEngine (device tensors through torch)."""

    def __init__(self, engine, device):
        import torch
        self.torch = torch
        self.e = engine
        self.device = torch.device("cuda", device)
        self.meta_width = int(engine._lib.rbpf_pack_meta_width(engine._h))
        self._buf = None
        # the engine works on torch's current stream: its kernels, the collectives and the copies are then ordered
        # without host synchronisation (the only one left per resample is reading the ancestor indices back)
        engine.synchronize()
        self.stream = torch.cuda.current_stream(self.device)
        engine.set_stream(self.stream.cuda_stream)

    def close(self):
        """Detach from the engine: the borrowed torch stream goes back (everything the engine queued on it has finished)
        and the tensors this adapter holds are dropped; the engine itself stays usable on a stream of its own."""
        if self.e is not None and getattr(self.e, "_h", None) and self.e._h.value:
            self.e.release_stream()
        self._buf = None
        self.stream = None
        self.e = None

    def set_global_ids(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        self.e._check(self.e._lib.rbpf_set_global_ids(self.e._h, ids.ctypes.data_as(_I32)))

    def weights_global(self, n_global):
        t = self.torch.empty(n_global, dtype=self.torch.float64, device=self.device)
        self.e._check(self.e._lib.rbpf_export_weights(self.e._h, _vp(t.data_ptr()), n_global))
        return t

    def weights_global_early(self, n_global):
        """weights_global right after the weighting kernel of scan_update_begin, plus the NaN-branch flag element."""
        t = self.torch.empty(n_global + 1, dtype=self.torch.float64, device=self.device)
        self.e._check(self.e._lib.rbpf_export_weights_early(self.e._h, _vp(t.data_ptr()), n_global, _vp(self.stream.cuda_stream)))
        return t

    def indices_early(self, wglobal, u):
        """Queue the ancestor computation and its read-back (no waiting)."""
        self._early_n = wglobal.numel() - 1
        self.e._check(self.e._lib.rbpf_resample_indices_global_early(self.e._h, _vp(wglobal.data_ptr()), self._early_n, float(u),
                                                                     _vp(self.stream.cuda_stream)))

    def indices_wait(self):
        """(did, idx, any_bad) of the queued computation; waits for its read-back only."""
        idx = np.empty(self._early_n, dtype=np.int32)
        did, bad = _c.c_int32(), _c.c_double()
        self.e._check(self.e._lib.rbpf_resample_indices_global_wait(self.e._h, idx.ctypes.data_as(_I32), _c.byref(did), _c.byref(bad)))
        return bool(did.value), idx, bad.value != 0.0

    def indices(self, wglobal, u):
        n = wglobal.numel()
        idx = np.empty(n, dtype=np.int32)
        did = _c.c_int32()
        self.e._check(self.e._lib.rbpf_resample_indices_global(self.e._h, _vp(wglobal.data_ptr()), n, float(u),
                                                               idx.ctypes.data_as(_I32), _c.byref(did)))
        return bool(did.value), idx

    def pack(self, local_idx):
        local_idx = np.ascontiguousarray(local_idx, dtype=np.int32)
        n = len(local_idx)
        meta = np.zeros((n, self.meta_width), dtype=np.int32)
        if n == 0:
            return meta, self.torch.empty(0, dtype=self.torch.uint8, device=self.device)
        cap = n * int(self.e._lib.rbpf_packed_particle_bytes(self.e._h))
        # worst-case capacity can be large: grow on demand from the actual bounding boxes instead
        cap = min(cap, max(1 << 22, n * (1 << 20)))
        while True:
            buf = self.torch.empty(cap, dtype=self.torch.uint8, device=self.device)
            nbytes = _c.c_int64()
            rc = self.e._lib.rbpf_pack_particles(self.e._h, local_idx.ctypes.data_as(_I32), n, _vp(buf.data_ptr()), cap,
                                                 meta.ctypes.data_as(_I32), _c.byref(nbytes))
            if rc == -2 and cap < n * int(self.e._lib.rbpf_packed_particle_bytes(self.e._h)):
                cap *= 4
                continue
            self.e._check(rc)
            return meta, buf[:nbytes.value]

    # the same in three steps with a single host wait (rbpf_gather_pack_meta / rbpf_meta_from_raw / rbpf_pack_particles_raw)
    @property
    def raw_width(self):
        return int(self.e._lib.rbpf_pack_raw_width(self.e._h))

    def gather_pack_meta(self, local_idx):
        """Tile boxes of the departing particles as a device int32 tensor [n * raw_width]; nothing is waited for."""
        local_idx = np.ascontiguousarray(local_idx, dtype=np.int32)
        raw = self.torch.empty(len(local_idx) * self.raw_width, dtype=self.torch.int32, device=self.device)
        if len(local_idx):
            self.e._check(self.e._lib.rbpf_gather_pack_meta(self.e._h, local_idx.ctypes.data_as(_I32), len(local_idx), _vp(raw.data_ptr())))
        return raw

    def meta_from_raw(self, raw, n):
        """(meta rows for rbpf_unpack_particles, payload bytes per particle) from gathered records on the host."""
        raw = np.ascontiguousarray(raw, dtype=np.int32)
        meta = np.zeros((n, self.meta_width), dtype=np.int32)
        nbytes = _c.c_int64()
        if n:
            self.e._check(self.e._lib.rbpf_meta_from_raw(self.e._h, raw.ctypes.data_as(_I32), n, meta.ctypes.data_as(_I32), _c.byref(nbytes)))
        return meta

    def pack_raw(self, local_idx, raw, nbytes):
        """Pack with the records already on the host; the buffer is filled in stream order."""
        local_idx = np.ascontiguousarray(local_idx, dtype=np.int32)
        raw = np.ascontiguousarray(raw, dtype=np.int32)
        buf = self.torch.empty(int(nbytes), dtype=self.torch.uint8, device=self.device)
        if len(local_idx):
            out = _c.c_int64()
            self.e._check(self.e._lib.rbpf_pack_particles_raw(self.e._h, local_idx.ctypes.data_as(_I32), len(local_idx),
                                                              raw.ctypes.data_as(_I32), _vp(buf.data_ptr()), int(nbytes), _c.byref(out)))
            assert out.value == int(nbytes)
        return buf

    def apply_local(self, new_src, new_gid):
        s = np.ascontiguousarray(new_src, dtype=np.int32)
        g = np.ascontiguousarray(new_gid, dtype=np.int32)
        self.e._check(self.e._lib.rbpf_apply_resample_local(self.e._h, s.ctypes.data_as(_I32), g.ctypes.data_as(_I32)))

    def unpack(self, local_idx, meta, payload):
        local_idx = np.ascontiguousarray(local_idx, dtype=np.int32)
        if len(local_idx) == 0:
            return
        meta = np.ascontiguousarray(meta, dtype=np.int32)
        if not payload.is_cuda:                            # host-staged transport: the copy up is synchronous
            payload = payload.to(self.device)
        payload = payload.contiguous()                     # on the engine's stream already: stream order is enough
        self.e._check(self.e._lib.rbpf_unpack_particles(self.e._h, local_idx.ctypes.data_as(_I32), len(local_idx),
                                                        _vp(payload.data_ptr()), meta.ctypes.data_as(_I32)))

    def pose(self, local_index):
        return self.e.poses()[local_index]

    def empty_payload(self, nbytes):
        return self.torch.empty(nbytes, dtype=self.torch.uint8, device=self.device)


import ctypes as _c  # noqa: E402
_I32 = _c.POINTER(_c.c_int32)


def _vp(ptr):
    return _c.c_void_p(ptr)


class ShardedResampler:
    """Global resample over `world` ranks, `p_local` particles each.  `shard` is an EngineShard (or any object
    with its methods); collectives go through torch.distributed (backend "nccl" = RCCL on ROCm)."""

    def __init__(self, rank: int, world: int, p_local: int, device: Optional[int] = None, dist=None):
        if dist is None:
            import torch.distributed as dist
        import torch
        self.torch, self.dist = torch, dist
        self.rank, self.world, self.p_local = rank, world, p_local
        self.n_global = world * p_local
        self.device = device
        self.host_staged = dist.get_backend() == "gloo"
        g = np.arange(self.n_global)
        self.owner = (g // p_local).astype(np.int32)       # rank of every global particle
        self.local_of = (g % p_local).astype(np.int32)     # its index inside that rank's engine
        self.shard = None
        self.stats = {"resamples": 0, "moved": 0, "bytes_sent": 0}
        self.timing = {} if os.environ.get("RBPF_SHARD_TIMING") else None   # developer aid: host seconds per phase

    def attach(self, engine_or_shard):
        self.shard = engine_or_shard if hasattr(engine_or_shard, "weights_global") else EngineShard(engine_or_shard, self.device or 0)
        self.shard.set_global_ids(np.arange(self.rank * self.p_local, (self.rank + 1) * self.p_local))

    def close(self):
        """Drop the device tensors of an unfinished early resample and detach the shard (call before engine.close()
        and before torch.distributed.destroy_process_group())."""
        self._early = None
        if self.shard is not None and hasattr(self.shard, "close"):
            self.shard.close()
        self.shard = None

    # -- transport helpers ---------------------------------------------------------------------------------------------
    def _all_reduce(self, t):
        if self.host_staged and t.is_cuda:
            c = t.cpu()
            self.dist.all_reduce(c)
            t.copy_(c)
        else:
            self.dist.all_reduce(t)
        return t

    def _all_to_all(self, send, out_splits, in_splits, dtype):
        torch = self.torch
        dev = send.device
        if self.host_staged and send.is_cuda:
            send = send.cpu()
        recv = torch.empty(int(sum(in_splits)), dtype=dtype, device=send.device)
        self.dist.all_to_all_single(recv, send.contiguous(), [int(x) for x in in_splits], [int(x) for x in out_splits])
        return recv.to(dev) if recv.device != dev else recv

    def _all_to_all_begin(self, send, out_splits, in_splits, dtype):
        """all_to_all that runs beside what is queued on the current stream afterwards (device transport only):
        returns (recv, finish); finish() makes the current stream wait for the exchange."""
        if self.host_staged or not send.is_cuda:
            recv = self._all_to_all(send, out_splits, in_splits, dtype)
            return recv, (lambda: None)
        recv = self.torch.empty(int(sum(in_splits)), dtype=dtype, device=send.device)
        work = self.dist.all_to_all_single(recv, send.contiguous(), [int(x) for x in in_splits], [int(x) for x in out_splits],
                                           async_op=True)
        return recv, work.wait

    # -- the step --------------------------------------------------------------------------------------------------------
    def _book(self, plan):
        """Ownership of every new global particle (replicated); local indices only of this rank's own particles."""
        self.owner[:] = plan.dest
        mine = plan.new_gid[self.rank]
        self.local_of[mine] = np.arange(len(mine), dtype=np.int32)

    def _tick(self, name, t0):
        if self.timing is not None:
            self.timing[name] = self.timing.get(name, 0.0) + time.perf_counter() - t0
        return time.perf_counter()

    def resample_begin(self, u: float):
        """Call between scan_update_begin and scan_update_end: the weight export, the collective, the ancestor
        computation and its read-back are queued BEFORE the map update, so that the host gets the ancestors, plans the
        migration and queues the tile copies while the GPU runs the map update.  Falls back to nothing
        (resample_finish does all the work) for shards without the early calls or a host-staged transport."""
        self._early, self._u = None, u
        sh = self.shard
        if self.host_staged or not hasattr(sh, "weights_global_early"):
            return
        t0 = time.perf_counter()
        wl = sh.weights_global_early(self.n_global)
        self.dist.all_reduce(wl)                                             # the one collective on the weights
        sh.indices_early(wl, u)
        self._early = wl
        self._tick("early_queue", t0)

    def resample_finish(self) -> Tuple[bool, Optional[np.ndarray]]:
        """Call after scan_update_end.  Uses the early result unless a particle took the NaN-covariance branch (its
        weight changed after the map update, robot.py:73-78); then the whole resample runs late, as resample()."""
        if getattr(self, "_early", None) is None:
            return self.resample(self._u)
        t0 = time.perf_counter()
        did, idx, any_bad = self.shard.indices_wait()
        self._early = None
        t0 = self._tick("indices_wait", t0)
        if any_bad:                                                          # the same on every rank (part of the collective)
            self.stats["late"] = self.stats.get("late", 0) + 1
            return self.resample(self._u)
        return self._apply(did, idx)

    def resample(self, u: float) -> Tuple[bool, Optional[np.ndarray]]:
        sh = self.shard
        t0 = time.perf_counter()
        wl = sh.weights_global(self.n_global)
        t0 = self._tick("export", t0)
        w = self._all_reduce(wl)                                             # the one collective on the weights
        t0 = self._tick("all_reduce", t0)
        did, idx = sh.indices(w, u)
        t0 = self._tick("indices", t0)
        return self._apply(did, idx)

    def _apply(self, did, idx) -> Tuple[bool, Optional[np.ndarray]]:
        sh, torch = self.shard, self.torch
        t0 = time.perf_counter()
        if not did:
            return False, None
        plan = plan_migration(idx, self.owner, self.local_of, self.world, self.p_local, rank=self.rank)
        t0 = self._tick("plan", t0)
        r, W = self.rank, sh.meta_width
        if plan.n_move == 0:                                                 # every rank sees the same plan: no exchange
            sh.apply_local(plan.new_src[r], plan.new_gid[r])
            t0 = self._tick("apply_local", t0)
            self._book(plan)
            self.stats["resamples"] += 1
            return True, idx
        if hasattr(sh, "gather_pack_meta"):
            # What leaves this rank, ordered by destination.  One host wait for the whole migration: the tile boxes of the
            # departing particles are gathered on the device, exchanged between the ranks while still there (the counts per
            # pair of ranks follow from the plan, which every rank holds), and read back together with the incoming ones.
            n_out = [len(plan.send[r][d]) for d in range(self.world)]
            n_in = [len(plan.send[q][r]) for q in range(self.world)]
            leaving = np.concatenate([np.asarray(plan.send[r][d], dtype=np.int32) for d in range(self.world)]) if sum(n_out) else np.zeros(0, dtype=np.int32)
            RW = sh.raw_width
            raw_out = sh.gather_pack_meta(leaving)
            raw_in = self._all_to_all(raw_out, [n * RW for n in n_out], [n * RW for n in n_in], torch.int32)
            both = torch.cat([raw_out, raw_in]).cpu().numpy()                    # the wait
            raw_out_h, raw_in_h = both[:sum(n_out) * RW], both[sum(n_out) * RW:]
            meta_out, meta_in = sh.meta_from_raw(raw_out_h, sum(n_out)), sh.meta_from_raw(raw_in_h, sum(n_in))
            ends_o, ends_i = np.cumsum(n_out), np.cumsum(n_in)
            b_out = [int(meta_out[e - n:e, 1].astype(np.int64).sum()) * 16 for n, e in zip(n_out, ends_o)]
            b_in = [int(meta_in[e - n:e, 1].astype(np.int64).sum()) * 16 for n, e in zip(n_in, ends_i)]
            pay_send = sh.pack_raw(leaving, raw_out_h, sum(b_out))
            # the departing particles are packed (stream order): the payload exchange starts, and beside it the local
            # part - slot pairing, state permutation, tile copies of duplicated ancestors - may reuse their slots
            if sum(b_out) + sum(b_in) > 0 or self.world > 1:
                pay_recv, finish = self._all_to_all_begin(pay_send, b_out, b_in, torch.uint8)
            else:
                pay_recv, finish = sh.empty_payload(0), (lambda: None)
            sh.apply_local(plan.new_src[r], plan.new_gid[r])
            finish()
        else:                                                                # shards with the one-call pack only (tests)
            # pack what leaves this rank in ONE call, ordered by destination (one gather of the metadata, one kernel);
            # the per-destination byte counts follow from the metadata rows (16-byte units in column 1)
            n_out = [len(plan.send[r][d]) for d in range(self.world)]
            leaving = np.concatenate([np.asarray(plan.send[r][d], dtype=np.int32) for d in range(self.world)]) if sum(n_out) else np.zeros(0, dtype=np.int32)
            meta_all, pay_all = sh.pack(leaving)
            ends = np.cumsum(n_out)
            b_out = [int(meta_all[e - n:e, 1].astype(np.int64).sum()) * 16 for n, e in zip(n_out, ends)]
            metas, payloads = [meta_all], [pay_all]
            # the departing particles are packed (stream order): the local part - slot pairing, state permutation, tile
            # copies of duplicated ancestors - can run now, while the host exchanges the metadata
            sh.apply_local(plan.new_src[r], plan.new_gid[r])
            n_in = [len(plan.send[q][r]) for q in range(self.world)]
            if sum(n_out) + sum(n_in) > 0 or self.world > 1:
                dev = payloads[0].device
                meta_send = torch.from_numpy(np.concatenate(metas).reshape(-1)).to(dev)
                meta_recv = self._all_to_all(meta_send, [n * W for n in n_out], [n * W for n in n_in], torch.int32)
                meta_in = meta_recv.cpu().numpy().reshape(-1, W)
                b_in, k = [], 0
                for q in range(self.world):
                    b_in.append(int(meta_in[k:k + n_in[q], 1].astype(np.int64).sum()) * 16)
                    k += n_in[q]
                pay_send = torch.cat(payloads) if sum(b_out) else sh.empty_payload(0)
                pay_recv = self._all_to_all(pay_send, b_out, b_in, torch.uint8)
            else:
                meta_in, pay_recv = np.zeros((0, W), dtype=np.int32), sh.empty_payload(0)
        arrivals = np.nonzero(plan.new_src[r] < 0)[0].astype(np.int32)
        sh.unpack(arrivals, meta_in, pay_recv)
        self._book(plan)
        self.stats["resamples"] += 1
        self.stats["moved"] += plan.n_move
        self.stats["bytes_sent"] += sum(b_out)
        return True, idx

    def refresh_last_scan(self):
        """main.py:167-168 for a sharded filter, on the device: the rank that holds global particle 0 transforms the
        current scan to that particle's pose and broadcasts the points; every rank keeps them as the previous scan
        for scan_update(adj=True).  No pose is read back, nothing waits on the host."""
        torch, e = self.torch, self.shard.e
        src = int(self.owner[0])
        buf = torch.empty(2 * int(e.cfg.max_beams), dtype=torch.float64, device=self.shard.device)
        n = _c.c_int32(e.n_beams)
        if self.rank == src:
            e.refresh_last_scan(int(self.local_of[0]))
            e._check(e._lib.rbpf_export_last_scan(e._h, _vp(buf.data_ptr()), _c.byref(n)))
        if self.world > 1:
            self.dist.broadcast(buf, src=src)
        if self.rank != src:
            e._check(e._lib.rbpf_import_last_scan(e._h, _vp(buf.data_ptr()), e.n_beams))

    def pose_of_particle0(self) -> np.ndarray:
        """main.py:152,167: particles[0].get_latest_pose(), from whichever rank holds global particle 0."""
        torch = self.torch
        src = int(self.owner[0])
        t = torch.zeros(3, dtype=torch.float64)
        if self.rank == src:
            t = torch.from_numpy(np.asarray(self.shard.pose(int(self.local_of[0])), dtype=np.float64).copy())
        if not self.host_staged:
            t = t.to(torch.device("cuda", self.device or 0))
        self.dist.broadcast(t, src=src)
        return t.cpu().numpy()

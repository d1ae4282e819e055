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

import numpy as np


@dataclass
class Plan:
    dest: np.ndarray          # [n] rank that holds new global particle j
    src_rank: np.ndarray      # [n] rank that holds its ancestor
    new_gid: List[np.ndarray]     # per rank: global ids in the rank's new local order
    new_src: List[np.ndarray]     # per rank: old local index of the ancestor, -1 for arrivals (last)
    send: List[List[np.ndarray]]  # send[r][d]: old local indices rank r packs for rank d (in global-id order)
    n_move: int


def plan_migration(idx: np.ndarray, owner: np.ndarray, local_of: np.ndarray, world: int, p_local: int) -> Plan:
    """idx[j] = old global id continued by new global particle j; owner/local_of = rank and local index of
    every old global particle.  Deterministic; every rank computes the same plan."""
    idx = np.asarray(idx, dtype=np.int64)
    n = len(idx)
    src_rank = owner[idx]
    dest = np.full(n, -1, dtype=np.int32)
    counts = np.zeros(world, dtype=np.int64)
    surplus = []
    for r in range(world):
        js = np.nonzero(src_rank == r)[0]
        keep = js[:p_local]
        dest[keep] = r
        counts[r] = len(keep)
        surplus.append(js[p_local:])
    pool = np.concatenate(surplus) if surplus else np.empty(0, dtype=np.int64)
    k = 0
    for d in range(world):
        need = int(p_local - counts[d])
        if need > 0:
            dest[pool[k:k + need]] = d
            k += need
    assert k == len(pool) and (dest >= 0).all()
    new_gid, new_src, send = [], [], [[np.empty(0, dtype=np.int32) for _ in range(world)] for _ in range(world)]
    for r in range(world):
        js = np.nonzero(dest == r)[0]
        kept = js[src_rank[js] == r]
        anc_local = local_of[idx[kept]]
        order = np.lexsort((kept, anc_local))                    # by ancestor's local index, then global id
        kept, anc_local = kept[order], anc_local[order]
        arr = js[src_rank[js] != r]
        arr = arr[np.lexsort((arr, src_rank[arr]))]              # by source rank, then global id
        new_gid.append(np.concatenate([kept, arr]).astype(np.int32))
        new_src.append(np.concatenate([anc_local, np.full(len(arr), -1)]).astype(np.int32))
    for r in range(world):
        for d in range(world):
            if d == r:
                continue
            js = np.nonzero((src_rank == r) & (dest == d))[0]
            send[r][d] = local_of[idx[js]].astype(np.int32)
    return Plan(dest, src_rank.astype(np.int32), new_gid, new_src, send, int((dest != src_rank).sum()))


class EngineShard:
    """Adapter between ShardedResampler and one rank's ParticleEngine (device tensors through torch)."""

    def __init__(self, engine, device):
        import torch
        self.torch = torch
        self.e = engine
        self.device = torch.device("cuda", device)
        self.meta_width = int(engine._lib.rbpf_pack_meta_width(engine._h))
        self._buf = None

    def set_global_ids(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        self.e._check(self.e._lib.rbpf_set_global_ids(self.e._h, ids.ctypes.data_as(_I32)))

    def weights_global(self, n_global):
        t = self.torch.empty(n_global, dtype=self.torch.float64, device=self.device)
        self.e._check(self.e._lib.rbpf_export_weights(self.e._h, _vp(t.data_ptr()), n_global))
        return t

    def indices(self, wglobal, u):
        n = wglobal.numel()
        idx = np.empty(n, dtype=np.int32)
        did = _c.c_int32()
        self.torch.cuda.synchronize(self.device)
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

    def apply_local(self, new_src, new_gid):
        s = np.ascontiguousarray(new_src, dtype=np.int32)
        g = np.ascontiguousarray(new_gid, dtype=np.int32)
        self.e._check(self.e._lib.rbpf_apply_resample_local(self.e._h, s.ctypes.data_as(_I32), g.ctypes.data_as(_I32)))

    def unpack(self, local_idx, meta, payload):
        local_idx = np.ascontiguousarray(local_idx, dtype=np.int32)
        if len(local_idx) == 0:
            return
        meta = np.ascontiguousarray(meta, dtype=np.int32)
        payload = payload.to(self.device).contiguous()
        self.torch.cuda.synchronize(self.device)
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

    def attach(self, engine_or_shard):
        self.shard = engine_or_shard if hasattr(engine_or_shard, "weights_global") else EngineShard(engine_or_shard, self.device or 0)
        self.shard.set_global_ids(np.arange(self.rank * self.p_local, (self.rank + 1) * self.p_local))

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

    # -- the step --------------------------------------------------------------------------------------------------------
    def resample(self, u: float) -> Tuple[bool, Optional[np.ndarray]]:
        sh, torch = self.shard, self.torch
        w = self._all_reduce(sh.weights_global(self.n_global))              # the one collective on the weights
        did, idx = sh.indices(w, u)
        if not did:
            return False, None
        plan = plan_migration(idx, self.owner, self.local_of, self.world, self.p_local)
        r, W = self.rank, sh.meta_width
        if plan.n_move == 0:                                                 # every rank sees the same plan: no exchange
            sh.apply_local(plan.new_src[r], plan.new_gid[r])
            for q in range(self.world):
                self.owner[plan.new_gid[q]] = q
                self.local_of[plan.new_gid[q]] = np.arange(len(plan.new_gid[q]), dtype=np.int32)
            self.stats["resamples"] += 1
            return True, idx
        # pack what leaves this rank, destination by destination
        metas, payloads, n_out, b_out = [], [], [], []
        for d in range(self.world):
            li = plan.send[r][d]
            m, p = sh.pack(li) if len(li) else (np.zeros((0, W), dtype=np.int32), sh.empty_payload(0))
            metas.append(m); payloads.append(p); n_out.append(len(li)); b_out.append(int(p.numel()))
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
        sh.apply_local(plan.new_src[r], plan.new_gid[r])
        arrivals = np.nonzero(plan.new_src[r] < 0)[0].astype(np.int32)
        sh.unpack(arrivals, meta_in, pay_recv)
        # replicated bookkeeping
        for q in range(self.world):
            self.owner[plan.new_gid[q]] = q
            self.local_of[plan.new_gid[q]] = np.arange(len(plan.new_gid[q]), dtype=np.int32)
        self.stats["resamples"] += 1
        self.stats["moved"] += plan.n_move
        self.stats["bytes_sent"] += sum(b_out)
        return True, idx

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

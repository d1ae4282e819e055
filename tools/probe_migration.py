"""Developer probe (GPU box): what the sharded resample would move.  One engine holds the particles of `world` virtual ranks
(p_local each) and runs the bench's loop; after every resample the migration rule of thesis_amd/sharding.plan_migration (a
new particle stays on its ancestor's rank while there is room, the surplus fills the ranks that are short) is applied to the
ancestors: migrating particles per step, bytes per migrating particle (the engine's own tile copies give the size of a
packed tile), size of the all-reduced weight vector.  usage: probe_migration.py <world> <p_local> [steps]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import PERIOD_S
from thesis_amd.engine import ParticleEngine
from thesis_amd.datasets import synthetic

world, p_local = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
P = world * p_local
angles, ranges, odo, truth = synthetic.make_log(steps + 2, 1081, period=PERIOD_S)
e = ParticleEngine(P, max_beams=1081, pool_tiles=2 * P + 64, seed=42)
e.set_scan(ranges[0], angles)
e.map_update(np.zeros((P, 3)))
e.refresh_last_scan(0)
urng = np.random.Generator(np.random.PCG64(777))
owner = np.arange(P) // p_local
movers, resamples, per_pair_max = [], 0, 0
for k in range(steps):
    e.imu_update("velocity", odo[k], PERIOD_S * 1e4)
    e.set_scan(ranges[k + 1], angles)
    e.scan_update(adj=not (k % 5 < 2))
    did, idx = e.resample(float(urng.random()))
    if k % 5 == 0:
        e.refresh_last_scan(0)
    if not did:
        movers.append(0)
        continue
    resamples += 1
    src = owner[np.asarray(idx, dtype=np.int64)]
    by_rank = np.argsort(src, kind="stable")
    cnt = np.bincount(src, minlength=world)
    first = np.concatenate(([0], np.cumsum(cnt)[:-1]))
    pos = np.empty(P, dtype=np.int64)
    pos[by_rank] = np.arange(P) - first[src[by_rank]]
    keep = pos < p_local
    dest = np.where(keep, src, -1)
    pool = by_rank[~keep[by_rank]]
    need = p_local - np.minimum(cnt, p_local)
    dest[pool] = np.repeat(np.arange(world), need)
    mv = dest != src
    movers.append(int(mv.sum()))
    if mv.any():
        per_pair_max = max(per_pair_max, int(np.bincount(src[mv] * world + dest[mv], minlength=world * world).max()))
    owner = dest
c = e.counters()
copy_bytes = c["bytes_copied"] / max(c["resample_copies"], 1) / 2          # read + write per copy -> one packed tile
out = {"world": world, "particles_per_rank": p_local, "steps": steps, "resamples": resamples,
       "migrating_particles_per_step_mean": float(np.mean(movers)), "migrating_particles_per_step_max": int(np.max(movers)),
       "migrating_per_rank_per_step_mean": float(np.mean(movers)) / world, "largest_pair_in_one_step": per_pair_max,
       "bytes_per_migrating_particle": float(copy_bytes + 104), "migration_bytes_per_step_mean_total": float(np.mean(movers) * (copy_bytes + 104)),
       "allreduce_bytes": P * 8 + 8, "duplicates_per_step_in_rank_copies": c["resample_copies"] / steps}
print(json.dumps(out))
e.close()

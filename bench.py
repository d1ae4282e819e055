#!/usr/bin/env python3
"""bench.py -- particle-updates/s of the RBPF-SLAM particle-update hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N = 1: run directly)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one accepted lidar scan of the reference's main loop (main.py:138-214) over all particles:
IMU propagation (robot.py:45-57), Robot.map_update for every particle (robot.py:59-115: scan match,
30-sample proposal, weighting, moments, ray-cast map update) and resample (main.py:46-79), with the
reference's scan-match cadence (main.py:156-159) and last_scan refresh (main.py:167-168).

Workload (BASELINE.json configs[2]): 4096 particles per GPU, 1081 beams over 270
degrees, 0.05 m cells, synthetic `room16` world (data/fr101.log is not in the reference's tree).  configs[1]
(1024 particles) and the north-star size (10 240) are measured in the same run and reported beside it.  Weak scaling:
every rank holds 4096 particles; the per-particle weights are all-reduced over RCCL before the global resampling.
Inputs (the scan log) are generated before the timed region; one scan (17 KB) is uploaded per step.

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # RCCL needs dmabuf IPC on this driver

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
PERIOD_S = 0.7                 # one accepted scan every 0.35 m of travel (main.py:42 DIST_THRESHOLD = 0.33)


NDT_DEFAULT = 1


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--particles", type=int, default=4096, help="particles per GPU")
    ap.add_argument("--beams", type=int, default=1081)
    ap.add_argument("--cell-size", type=float, default=0.05)
    ap.add_argument("--ndt", type=int, default=NDT_DEFAULT, choices=(0, 1),
                    help="second matcher stage (matchScanCustom.m:32-50, NDT refinement)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-target-run", action="store_true", help="skip the extra 1024- and 10 240-particle measurements")
    return ap.parse_args()


class Runner:
    """The reference's event loop body for accepted scans, on one rank."""

    def __init__(self, P, B, cell_size, log, rank=0, world=1, shard=None, device=0, ndt=NDT_DEFAULT):
        from thesis_amd.engine import ParticleEngine
        self.P, self.B, self.rank, self.world = P, B, rank, world
        self.angles, self.ranges, self.odo, self.true_poses = log
        # the seed is global: proposal streams are keyed by (seed, step, global particle id)
        self.e = ParticleEngine(P, max_beams=B, cell_size=cell_size, pool_tiles=2 * P + 64, seed=42, device=device,
                                ndt_refine=ndt)
        self.shard = shard
        if shard is not None:
            shard.attach(self.e)
        self.frame = 0
        self.last_scan_xy = None
        self.urng = np.random.Generator(np.random.PCG64(777))   # same stream on every rank
        # cold start: the first scan goes into every map at the origin (update_count < 2 branch, main.py:155)
        self.e.set_scan(self.ranges[0], self.angles)
        self.e.map_update(np.zeros((P, 3)))
        self.host_last_scan = shard is not None and shard.host_staged      # gloo rehearsals keep the host path
        self._refresh_last_scan(0, np.zeros(3))

    def _refresh_last_scan(self, k, pose0=None):
        # main.py:167-168: last_scan = scan.from_global_reference(particles[0].get_latest_pose()); on the device (the
        # current scan is scan k): nothing is read back, the host keeps running ahead of the GPU
        if not self.host_last_scan:
            self.last_scan_xy = None
            if self.shard is None:
                self.e.refresh_last_scan(0)
            else:
                self.shard.refresh_last_scan()
            return
        if pose0 is None:
            pose0 = self.shard.pose_of_particle0()
        r, a = self.ranges[k], self.angles
        c, s = np.cos(pose0[2]), np.sin(pose0[2])
        x, y = r * np.cos(a), r * np.sin(a)
        self.last_scan_xy = np.stack([c * x - s * y + pose0[0], s * x + c * y + pose0[1]], axis=1)

    def step(self):
        k = self.frame
        e = self.e
        e.imu_update("velocity", self.odo[k], PERIOD_S * 1e4)                 # main.py:139-145
        e.set_scan(self.ranges[k + 1], self.angles)
        adj = not (k % 5 < 2)                                                 # main.py:156-159
        u = float(self.urng.random())
        if self.shard is None:
            e.scan_update(adj=adj, last_scan_xy=self.last_scan_xy if adj else None)
            e.resample_async(u)                                               # main.py:160
        else:   # sharded: the global resample starts as soon as the weights exist and overlaps the map update
            e.scan_update_begin(adj=adj, last_scan_xy=self.last_scan_xy if adj else None)
            self.shard.resample_begin(u)
            e.scan_update_end()
            self.shard.resample_finish()
        if k % 5 == 0:                                                        # main.py:167
            self._refresh_last_scan(k + 1)
        self.frame += 1


# the dominant kernel of every timed family (the map update's other kernel - 128x128 windows - exits at once when the first
# gave nothing back)
KERNEL_NAMES = {"raycast": "rbpf::map_update_ev_kernel", "match": "rbpf::match_kernel", "ndt": "rbpf::ndt_kernel",
                "weight": "rbpf::propose_weight_kernel", "resample": "rbpf::resample_copy_kernel"}


def pmc_traffic(kernel, particles):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/*pmc_traffic.json: separate
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of tools/pmc_run.py on the same workload; FETCH_SIZE
    doubled as MI355X_MICROARCH.md section HBM prescribes for 16-byte-per-lane reads on gfx950).  None when no pass
    was recorded for this particle count."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "*pmc_traffic.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1])).get(str(particles), {}).get(KERNEL_NAMES.get(kernel, ""), None)
    except Exception:
        return None
    return None if d is None else 2.0 * d["fetch_raw_bytes"] + d["write_bytes"]


def copy_peak_gbs(torch, nbytes=1 << 30, reps=10):
    """Measured HBM peak of a plain device-to-device copy (read + write bytes / time), SURVEY section 8(d)."""
    a = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    b = torch.empty_like(a)
    for _ in range(2):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def cpu_baseline(log, B, cell_size, seconds=12.0):
    """Reference-equivalent CPU path (oracle/rbpf_oracle.c, pinned bit-exact to the reference's outputs):
    Robot.map_update without the MATLAB scan matcher (as BASELINE.md section 2), all host cores."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import c_oracle, rbpf_oracle as orc
    lib = c_oracle.load()
    angles, ranges, odo, poses = log
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))                 # a one-GPU box's CPU share
    n_steps = 3
    sx = np.empty((n_steps, B)); sy = np.empty((n_steps, B))
    for s in range(n_steps):
        sx[s], sy[s] = orc.scan_xy(ranges[s], angles)
    rng = np.random.Generator(np.random.PCG64(5))
    K = 30

    def work(n_particles):
        g = (poses[:n_steps, None, None, :] + rng.normal(0, 0.01, size=(n_steps, n_particles, K, 3))).copy()
        prs = np.ones((n_steps, n_particles, K))
        t0 = time.perf_counter()
        lib.orc_bench_particle_updates(n_particles, n_steps, c_oracle.dp(sx), c_oracle.dp(sy), B, c_oracle.dp(g),
                                       c_oracle.dp(prs), K, cell_size)
        return time.perf_counter() - t0

    t1 = work(4)                                   # calibrate: 4 particles x 3 steps on one core
    per_pu = t1 / (4 * n_steps)
    n_each = max(1, min(int(seconds / (per_pu * n_steps)), 2000))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:          # ctypes releases the GIL: real parallelism
        list(ex.map(work, [n_each] * cores))
    wall = time.perf_counter() - t0
    total = cores * n_each * n_steps
    # the loop-faithful Python restatement (the form the reference itself has), one thread, a few particle-updates
    py_n = 2
    hm = orc.OracleHybridMap(cell_size)
    hm.update(tuple(poses[0]), sx[0], sy[0])
    g = poses[1] + rng.normal(0, 0.01, size=(py_n, K, 3))
    t0 = time.perf_counter()
    for i in range(py_n):
        w = orc.generate_sample_weight(hm, g[i], sx[1], sy[1], np.ones(K))
        mean, _, _ = orc.proposal_moments(g[i], w)
        hm.update(tuple(float(x) for x in mean), sx[1], sy[1])
    py_rate = py_n / (time.perf_counter() - t0)
    return {"value": total / wall, "unit": "particle-updates/s", "cores": cores, "kind": "port", "matcher_included": False,
            "sample": f"{total} particle-updates ({cores} threads x {n_each} particles x {n_steps} scans, B={B}, "
                      f"cs={cell_size}): C restatement of Robot.map_update (weighting + moments + ray-cast), "
                      f"scan matcher excluded (MATLAB, not timeable); single core: {1.0 / per_pu:.1f}/s",
            "python_port_single_thread": {"value": py_rate, "unit": "particle-updates/s", "cores": 1,
                                          "sample": f"{py_n} particle-updates of oracle/rbpf_oracle.py (the reference's own "
                                                    "loop structure: Robot._generate_sample_weight + moments + HybridMap.update)"}}


def _finite(x):
    """JSON has no NaN/Infinity: non-finite floats become null."""
    if isinstance(x, dict):
        return {k: _finite(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_finite(v) for v in x]
    if isinstance(x, float) and not np.isfinite(x):
        return None
    return x


def main():
    args = parse()
    # stdout carries exactly one line, the JSON: whatever libraries print there (RCCL's version banner on the first
    # collective, for one) is sent to stderr instead
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    force_dist = bool(os.environ.get("RBPF_FORCE_DIST"))     # rehearse the RCCL path with a single rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from thesis_amd.datasets import synthetic
    n_scans = args.steps + args.warmup + 2 + 20            # + the untimed per-kernel pass
    log = synthetic.make_log(n_scans, args.beams, period=PERIOD_S)

    shard = None
    if dist is not None:
        # RCCL sets up its peer-to-peer connections on first use, pair by pair: touch every pair (and the collectives the
        # step uses) once before anything is timed, so that no connection is built inside the timed region
        wt = torch.ones(1024 * world, dtype=torch.uint8, device="cuda")
        wr = torch.empty_like(wt)
        dist.all_to_all_single(wr, wt)
        wi = torch.ones(256 * world, dtype=torch.int32, device="cuda")
        dist.all_to_all_single(torch.empty_like(wi), wi)
        wd = torch.ones(64, dtype=torch.float64, device="cuda")
        dist.all_reduce(wd)
        dist.broadcast(wd, src=0)
        torch.cuda.synchronize()
        from thesis_amd.sharding import ShardedResampler
        shard = ShardedResampler(rank, world, args.particles, device=local_rank, dist=dist)
    run = Runner(args.particles, args.beams, args.cell_size, log, rank, world, shard, device=local_rank, ndt=args.ndt)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Timing events cost a few microseconds of stream time each.  The warm-up steps bracket every kernel family (that
    # is where the per-kernel table comes from and how the dominant kernel is found); the timed region brackets only
    # the dominant one, whose live duration the roofline needs.
    FAMILIES = ("raycast", "weight", "resample", "match", "ndt")
    assert set(KERNEL_NAMES) == set(FAMILIES), "every timed kernel family needs its dominant kernel's symbol"
    run.e.set_profiling(True)
    for _ in range(args.warmup):
        run.step()
    warm_ms = {k: run.e.kernel_ms(k) for k in FAMILIES} if args.warmup else {}
    warm_mean = {k: (float(v[1:].mean()) if len(v) > 1 else float(v.mean()) if len(v) else 0.0) for k, v in warm_ms.items()}
    dominant = max(FAMILIES, key=lambda k: warm_mean.get(k, 0.0)) if args.warmup else None
    run.e.set_profiling([dominant] if dominant else True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run.step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        moved = torch.tensor([float(shard.stats["moved"]), float(shard.stats["bytes_sent"])], dtype=torch.float64, device="cuda")
        dist.all_reduce(moved)
    if run.shard is not None and run.shard.timing is not None:
        print("shard timing (host s over the run):", {k: round(v, 4) for k, v in run.shard.timing.items()}, file=sys.stderr)
    if rank != 0:
        if run.shard is not None:
            run.shard.close()
        run.e.close()
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    P_total = args.particles * world
    value = P_total * args.steps / elapsed
    # ---- roofline of the dominant kernel, HIP events recorded inside the timed region ------------------
    kms = {k: run.e.kernel_ms(k) for k in FAMILIES}
    c = run.e.counters()
    if dominant is not None and dist is None:
        # the other kernels' durations: twenty more steps, untimed, with every family bracketed
        run.e.set_profiling(True)
        for _ in range(min(20, len(run.ranges) - run.frame - 2)):
            run.step()
        post = {k: run.e.kernel_ms(k) for k in FAMILIES}
        warm_mean = {k: (float(v.mean()) if len(v) else warm_mean.get(k, 0.0)) for k, v in post.items()}
    mean_ms = {k: (float(v.mean()) if len(v) else warm_mean.get(k, 0.0)) for k, v in kms.items()}
    n_upd = max(1, args.steps)
    W_per_particle = c["cells_written"] / (n_upd * args.particles)          # |W|, unique cells written per particle-update
    cells_per_particle = c["ray_cells_visited"] / (n_upd * args.particles)
    if dominant is None:
        dominant = max(FAMILIES, key=lambda k: mean_ms[k])
    # algorithmic bytes with 4-byte cells.  Per particle-update:
    #   ray-cast kernel   4|W| read + 4|W| write
    #   weighting kernel  4|R_w|, R_w = cells under the K*B sample endpoints (<= K*B, ~B distinct)
    #   match kernel      4 * region cells (the occupancy region staged once per particle)
    alg_bytes = {"raycast": 8.0 * W_per_particle, "weight": 4.0 * args.beams * 2,
                 "match": 4.0 * 480 * 480, "ndt": 4.0 * 480 * 480, "resample": 0.0}
    if c["resample_copies"]:
        alg_bytes["resample"] = 4.0 * c["bytes_copied"] / (n_upd * args.particles)
    ach = alg_bytes[dominant] * args.particles / (mean_ms[dominant] * 1e-3) / 1e9 if mean_ms[dominant] > 0 else 0.0
    traffic = pmc_traffic(dominant, args.particles)
    # the whole step against the same peak: SURVEY 8(d)'s bytes per particle-update (4-byte cells: cells read by matcher,
    # weighting and ray-cast, cells written, 2 x 104 B of state) over the step time
    bytes_pu = 8.0 * W_per_particle + 4.0 * args.beams * 2 + 208.0
    step_ach = bytes_pu * args.particles * world / (elapsed / args.steps) / 1e9
    n_fb = max(1, args.steps) * args.particles
    roofline = {"bound": "hbm", "kernel": dominant, "kernel_symbol": KERNEL_NAMES.get(dominant), "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                # the same kernel in the bytes it actually moves (int8 cells; PMC FETCH/WRITE passes, profiles/)
                "frac_stored_bytes": (traffic / (mean_ms[dominant] * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic and mean_ms[dominant] > 0 else None,
                "step_frac": step_ach / (HBM_PEAK_GBS * world),
                "step_bytes_per_particle_update": bytes_pu,
                "window_fallbacks_per_particle_step": c["window_fallbacks"] / n_fb,
                "fast_kernel_give_backs_per_particle_step": (int(c["fallback_geometry"]) + int(c["fallback_bound"]) + int(c["fallback_tables"])) / n_fb,
                "global_index_kernel_windows_per_particle_step": c["map_windows"] / n_fb,
                "algorithmic_bytes_per_particle_update": alg_bytes[dominant],
                "cell_bytes_algorithmic": 4, "cell_bytes_stored": 1,
                "achieved_stored_bytes_GBs": ach / 4.0,
                "kernel_ms_mean": mean_ms,
                "kernel_ms_source": f"{dominant}: HIP events in the timed region (raycast: the events ride on the first map kernel's dispatch "
                                    "and take its own start and end); the others: HIP events round the family's launches in untimed steps "
                                    "(after the timed region on one GPU, the warm-up steps otherwise)",
                "slow_cells_per_step": c["slow_cells"] / n_upd,
                "ndt": {"enabled": bool(args.ndt), "runs_per_step": c["ndt_runs"] / n_upd,
                        "evaluations_per_run": c["ndt_evaluations"] / max(1, c["ndt_runs"]),
                        "accepted_fraction": c["ndt_accepted"] / max(1, c["ndt_runs"])},
                "match_shared_per_step": c["match_shared"] / n_upd,       # exact duplicates that took their representative's match
                "window_fallback_particles_per_step": c["window_fallbacks"] / n_upd,
                "unique_cells_written_per_particle": W_per_particle,
                "ray_cells_per_particle": cells_per_particle}
    out = {"metric": "particle-updates/sec (particles x scans/s) @1081 beams", "value": value,
           "unit": "particle-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "int8 map cells (lattice log-odds), f64 poses/weights", "data": "synthetic",
           "config": {"workload": ("BASELINE configs[2]" if (args.particles, args.beams, args.cell_size) == (4096, 1081, 0.05)
                                   else "BASELINE configs[1]" if (args.particles, args.beams, args.cell_size) == (1024, 1081, 0.05)
                                   else "non-default size") +
                                  f": room16 synthetic, {args.particles} particles/GPU, "
                                  f"{args.beams} beams, {args.cell_size} m grid, K=30 samples, "
                                  + ("both matcher stages, " if args.ndt else "grid matcher stage only, ") +
                                  "resample every step",
                      "particles_per_gpu": args.particles, "beams": args.beams, "cell_size": args.cell_size,
                      "parallelism": f"particles sharded x{world}"},
           "roofline": roofline}
    if dist is not None:
        out["collective"] = {"weights": "one RCCL all-reduce of %d float64 per step" % P_total,
                             "migrated_particles_per_step": float(moved[0].item()) / (2 * max(1, args.steps + args.warmup)),
                             "migrated_bytes_per_step": float(moved[1].item()) / max(1, args.steps + args.warmup)}
    if run.shard is not None:
        run.shard.close()
    run.e.close()
    if world == 1:
        roofline["peak_measured_copy"] = copy_peak_gbs(torch)        # after the timed region, on an idle device
        roofline["frac_of_measured_copy"] = ach / roofline["peak_measured_copy"]
    if not args.no_target_run and world == 1:
        # the same step at configs[1] (1024 particles) and at the north-star size (>= 10k particles x 1081 beams on one
        # GPU): reported beside `value`, not in it
        for name, Pn in (("config1_1024", 1024), ("target_10k", 10240)):
            if Pn == args.particles:
                continue
            big = Runner(Pn, args.beams, args.cell_size, log, ndt=args.ndt)
            for _ in range(3):
                big.step()
            big.e.set_profiling(True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            nb = min(args.steps, 40)
            for _ in range(nb):
                big.step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            km = {k: (float(x.mean()) if len(x) else 0.0)
                  for k, x in ((k, big.e.kernel_ms(k)) for k in ("raycast", "weight", "resample", "match", "ndt"))}
            cb = big.e.counters()
            Wb = cb["cells_written"] / (nb * Pn)
            out[name] = {"particles": Pn, "value": Pn * nb / dt, "ms_per_step": 1e3 * dt / nb, "steps": nb,
                         "kernel_ms_mean": km,
                         "raycast_frac_of_hbm_peak": 8.0 * Wb * Pn / (km["raycast"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "step_frac": (8.0 * Wb + 4.0 * args.beams * 2 + 208.0) * Pn / (dt / nb) / 1e9 / HBM_PEAK_GBS,
                         "window_fallbacks_per_particle_step": cb["window_fallbacks"] / (nb * Pn)}
            big.e.close()
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(log, args.beams, args.cell_size)
    sys.stdout.flush()
    os.write(real_stdout, (json.dumps(_finite(out), allow_nan=False) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

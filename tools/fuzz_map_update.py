"""Developer fuzz: random scans / poses / beam counts / cell sizes, map update + sample weighting of the HIP path
against the C oracle (oracle/rbpf_oracle.c), cell for cell, through both map-update kernels.  Not a test: run on a GPU
box when the kernels change (`python tools/fuzz_map_update.py [cases] [seed]`); prints the first mismatch."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c_oracle, rbpf_oracle as orc

def run(N=120, SEED=1, kernels=None, verbose=True):
  """N random cases; returns the number that mismatch (0 = every map cell and every sample weight equals the oracle's)."""
  lib = c_oracle.load()
  rng = np.random.Generator(np.random.PCG64(SEED))
  bad = 0
  saved_env = os.environ.get("RBPF_MAP_KERNEL")
  for case in range(N):
      cs = float(rng.choice([0.05, 0.05, 0.1, 0.025]))
      B = int(rng.choice([1, 7, 64, 180, 361, 721, 1081, 1081, 1500]))
      kernel = os.environ.get("FUZZ_KERNEL") or str(rng.choice(kernels or ["auto", "auto", "auto", "ray", "window"]))
      if kernel != "auto":
          os.environ["RBPF_MAP_KERNEL"] = kernel
      else:
          os.environ.pop("RBPF_MAP_KERNEL", None)
      from thesis_amd import engine
      fov = float(rng.choice([np.pi / 3, np.pi, 1.5 * np.pi, 2 * np.pi * (1 - 1 / max(B, 2))]))
      ang = (-fov / 2 + np.arange(B) * (fov / max(B - 1, 1))) if B > 1 else np.array([rng.uniform(-3, 3)])
      P = 3
      e = engine.ParticleEngine(P, max_beams=B, cell_size=cs, pool_tiles=40)
      maps = [c_oracle.CMap(lib, cs) for _ in range(P)]
      # the first tile is centred (0,0) and others exist only once a ray entered them: half of the cases start inside it
      # (anywhere, negative side included), a fifth at its corners and edges, the rest anywhere (mostly no-ops, hybridmap.py:98-100)
      pick = rng.random()
      if pick < 0.5:
          centre = rng.uniform(-19.5, 19.5, size=2)
      elif pick < 0.7:
          centre = np.array([rng.choice([-20, 20, 0]), rng.choice([-20, 20, 0])]) + rng.uniform(-0.4, 0.4, 2)
      elif pick < 0.85:
          centre = rng.uniform(-55, 55, size=2)
      else:
          centre = np.array([rng.choice([-20, 20, 0, 60, -60]), rng.choice([-20, 20, 0])]) + rng.uniform(-0.3, 0.3, 2)
      ok = True
      hist = []
      for scan in range(int(rng.integers(1, 4))):
          style = rng.random()
          if style < 0.4:
              r = rng.uniform(0.3, 12.0) + rng.normal(0, 0.3, B).cumsum() * 0.05 + rng.normal(0, 0.02, B)
          elif style < 0.7:
              r = rng.uniform(0.0, 30.0, B)                       # incl. rays longer than 15 m and near-zero ranges
          elif style < 0.85:
              r = np.full(B, rng.uniform(0.5, 9.0)) + rng.normal(0, 0.005, B)
          else:                                                   # a smooth wall with a fifth of the beams ending next to the sensor
              r = rng.uniform(2.0, 14.0) + rng.normal(0, 0.3, B).cumsum() * 0.05
              short = rng.random(B) < 0.2
              r = np.where(short, rng.uniform(0.0, 0.5, B), r)
          r = np.abs(r)
          poses = np.column_stack([centre[0] + rng.normal(0, 0.4, P), centre[1] + rng.normal(0, 0.4, P), rng.uniform(-np.pi, np.pi, P)])
          e.set_scan(r, ang)
          e.map_update(poses)
          hist.append((r.copy(), poses.copy()))
          sx, sy = orc.scan_xy(r, ang)
          for p in range(P):
              maps[p].update(poses[p], sx, sy)
      for p in range(P):
          want = maps[p].tiles()
          got = {c: t for c, t in e.tiles(p)}
          if set(got) != set(want):
              ok = False; print("case", case, "tiles differ", sorted(got), sorted(want)); break
          for c in want:
              wq = np.rint(want[c] / 0.1).astype(np.int8)
              if not np.array_equal(got[c], wq):
                  d = np.argwhere(got[c] != wq)
                  ok = False; print("case", case, "cs", cs, "B", B, kernel, "particle", p, "tile", c, len(d), "cells differ, first", d[0], got[c][tuple(d[0])], wq[tuple(d[0])]); break
          if not ok:
              break
      if ok:
          # sample weighting on the final maps
          K = 30
          g = poses[:, None, :] + rng.normal(0, [0.03, 0.03, 0.01], size=(P, K, 3))
          prs = rng.uniform(0.5, 2.0, size=(P, K))
          w = e.weight_samples(g, prs)
          for p in range(P):
              wr = np.asarray(maps[p].sample_weight(g[p], sx, sy, prs[p]), dtype=np.float64)
              if not np.allclose(w[p], wr, rtol=1e-9, atol=1e-9):
                  ok = False; print("case", case, "weights differ", p, np.abs(w[p] - wr).max()); break
      bad += not ok
      if not ok:   # keep the inputs of a failing case (gpurun_out/ travels back)
          os.makedirs("gpurun_out", exist_ok=True)
          np.savez(f"gpurun_out/fuzz_fail_{SEED}_{case}.npz", cs=cs, B=B, ang=ang, kernel=kernel,
                   ranges=np.array([h[0] for h in hist]), poses=np.array([h[1] for h in hist]))
      c = e.counters()
      e.close()
      if verbose and case % 20 == 0:
          print("case", case, "ok so far, bad", bad, "cs", cs, "B", B, kernel, "fallbacks", c["window_fallbacks"], "reasons %x" % c["fallback_reasons"], "windows", c["map_windows"], "events", c["map_events"], "ev overflows", c["map_event_overflows"], flush=True)
  if saved_env is None: os.environ.pop("RBPF_MAP_KERNEL", None)
  else: os.environ["RBPF_MAP_KERNEL"] = saved_env
  if verbose: print("done", N, "cases, mismatching:", bad)
  return bad


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 120, int(sys.argv[2]) if len(sys.argv) > 2 else 1)

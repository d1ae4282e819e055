"""Developer fuzz: the HIP NDT stage against oracle/matcher_oracle.py through the stateless twin rbpf_match_scan, on
random polygonal rooms, offsets and guesses: final pose, score and the number of evaluations."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thesis_amd.engine import ParticleEngine, match_scan
from oracle import matcher_oracle as mo

def run(N=40, SEED=1, verbose=True):
    """N random rooms; returns (cases that entered the NDT stage, worst pose difference, worst relative score difference, cases whose evaluation counts differ)."""
    rng = np.random.Generator(np.random.PCG64(SEED))
    e0 = ParticleEngine(1, max_beams=1081, ndt_refine=0)
    e2 = ParticleEngine(1, max_beams=1081, ndt_refine=2)


    def rot(th):
        c, s = np.cos(th), np.sin(th)
        return np.array([[c, -s], [s, c]])


    worst_p, worst_s, diff_ev, ran = 0.0, 0.0, 0, 0
    for case in range(N):
        # a random closed polygon rasterised on the 0.05 m lattice, walls 1-3 cells thick
        nv = int(rng.integers(4, 9))
        angs = np.sort(rng.uniform(0, 2 * np.pi, nv))
        rad = rng.uniform(3.0, 9.0, nv)
        V = np.stack([rad * np.cos(angs), rad * np.sin(angs)], 1)
        pts = []
        for a, b in zip(V, np.roll(V, -1, axis=0)):
            n = int(np.hypot(*(b - a)) / 0.02) + 2
            pts.append(a + (b - a) * np.linspace(0, 1, n)[:, None])
        wall = np.concatenate(pts)
        cells = np.unique(np.rint(wall / 0.05).astype(int), axis=0)
        thick = int(rng.integers(1, 4))
        ref = np.unique(np.concatenate([cells + [i, j] for i in range(thick) for j in range(thick)]), axis=0) * 0.05
        sub = ref[rng.permutation(len(ref))[:min(len(ref), 1000)]]
        off = np.array([rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), rng.uniform(-0.3, 0.3)])
        g = np.array([rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), rng.uniform(-0.05, 0.05)]) if rng.random() < 0.5 else np.zeros(3)
        curr = (sub - off[:2]) @ rot(off[2] - g[2])
        rng3 = [0.7, 0.7, np.pi / 6]
        p0, cov0, s0 = match_scan(e0, curr, ref, g, 20, rng3)
        if np.isnan(cov0).any():
            continue
        b = e2.counters()
        p1, cov1, s1 = match_scan(e2, curr, ref, g, 20, rng3)
        a = e2.counters()
        if a["ndt_runs"] == b["ndt_runs"]:
            continue
        mcs, Nn = 0.05, 672
        occ, ox, oy = mo.rasterise(ref, g, mcs, Nn, 0.5, 15.0)
        pp = mo.beams_in_cells(curr, mcs)
        X0, Y0 = g[0] / mcs - ox + 0.5, g[1] / mcs - oy + 0.5
        start = (X0 + np.rint((p0[0] - g[0]) / mcs), Y0 + np.rint((p0[1] - g[1]) / mcs), p0[2])
        pw, score, evals = mo.ndt_refine(occ, pp, start, 2, ox, oy)
        want = np.array([g[0] + (pw[0] - X0) * mcs, g[1] + (pw[1] - Y0) * mcs, pw[2]])
        ran += 1
        accepted = a["ndt_accepted"] > b["ndt_accepted"]
        if accepted:
            worst_p = max(worst_p, float(np.abs(p1 - want).max()))
            worst_s = max(worst_s, abs(s1 - score) / max(score, 1e-9))
        if a["ndt_evaluations"] - b["ndt_evaluations"] != evals:
            diff_ev += 1
            print("case", case, "evaluations", a["ndt_evaluations"] - b["ndt_evaluations"], "oracle", evals, "pose diff", np.abs(p1 - want).max() if accepted else None)
    e0.close(); e2.close()
    if verbose:
        print("ran", ran, "worst |pose diff|", worst_p, "worst rel score diff", worst_s, "evaluation counts differing", diff_ev)

    return ran, worst_p, worst_s, diff_ev


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1)

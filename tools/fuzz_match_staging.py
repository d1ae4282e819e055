"""Developer fuzz: the matcher's fast field staging (funnel-shifted occupancy words + defect columns, 32- and 64-column
forms) against the bit-by-bit form (RBPF_MATCH_STAGE=slow) on random poses around tile edges / corners / the irregular
negative side, cell sizes 0.05 and 0.025: poses, covariances and weights after one scan update must be the same bits."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thesis_amd import engine
from thesis_amd.datasets import synthetic

def run(N=40, SEED=1, verbose=True):
    """N random poses round tile edges; returns the number of cases where the fast field staging and the bit-by-bit form differ."""
    rng = np.random.Generator(np.random.PCG64(SEED))
    B = 721
    ang = synthetic.beam_angles(B)
    bad = 0
    for case in range(N):
        cs = float(rng.choice([0.05, 0.025]))
        P = 6
        anchors = np.array([[20, 20], [-20, 20], [20, -20], [-20, -20], [0, 0], [60, -20], [-60, 60], [-20, 0], [0, -20]], dtype=float)
        poses = np.column_stack([anchors[rng.integers(0, len(anchors), P)] + rng.normal(0, 1.5, (P, 2)), rng.uniform(-np.pi, np.pi, P)])
        scans = [rng.uniform(2.0, 7.0) + 1.5 * np.sin(rng.integers(2, 7) * ang + rng.uniform(0, 6)) + rng.normal(0, 0.01, B) for _ in range(3)]
        outs = []
        for mode in ("fast", "slow"):
            if mode == "slow":
                os.environ["RBPF_MATCH_STAGE"] = "slow"
            else:
                os.environ.pop("RBPF_MATCH_STAGE", None)
            e = engine.ParticleEngine(P, max_beams=B, pool_tiles=80, seed=3, cell_size=cs)
            e.set_state(poses=poses)
            for s in scans[:2]:
                e.set_scan(s, ang); e.map_update(poses)
            e.set_state(poses=poses + [0.06, -0.04, 0.015], covs=np.diag([4e-5, 4e-5, 1e-5]))
            e.set_scan(scans[2], ang)
            e.scan_update(adj=False)
            outs.append((e.poses(), e.covs(), e.weights()))
            e.close()
        same = all(np.array_equal(a, b) for a, b in zip(outs[0], outs[1]))
        bad += not same
        if not same:
            print("case", case, "cs", cs, "poses", poses.tolist())
    os.environ.pop("RBPF_MATCH_STAGE", None)
    if verbose:
        print("done", N, "cases, mismatching:", bad)

    return bad


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1)

"""A/B check of a kernel change (developer tool, GPU box): the bench loop's state after every step, hashed.

    python tools/ab_hash.py thesis_amd/librbpf_hip.so thesis_amd/librbpf_hip_prev.so

Each library is loaded in its own process; for three configurations (0.05 m / 0.025 m / 0.1 m cells) the loop of
bench.Runner runs 40 steps and the poses, covariances and weights of all particles are hashed after each one.  The whole step
is deterministic (integer map update, keyed proposal streams, fixed-order reductions), so two builds that compute the
same thing print the same digests; the first differing step is reported."""
import hashlib
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [(256, 1081, 0.05, 40), (128, 181, 0.025, 30), (128, 361, 0.1, 30), (64, 1500, 0.05, 12)]


def child(lib_path):
    sys.path.insert(0, REPO)
    import numpy as np
    import thesis_amd._lib as _lib
    _lib.LIB_PATH = os.path.abspath(lib_path)
    from bench import Runner, PERIOD_S
    from thesis_amd.datasets import synthetic
    for (P, B, cs, steps) in CASES:
        log = synthetic.make_log(steps + 2, B, period=PERIOD_S)
        r = Runner(P, B, cs, log)
        for k in range(steps):
            r.step()
            h = hashlib.sha256()
            for arr in (r.e.poses(), r.e.covs(), r.e.weights()):
                h.update(np.ascontiguousarray(arr).tobytes())
            print(f"{P} {B} {cs} {k} {h.hexdigest()[:16]}", flush=True)
        c = r.e.counters()
        print(f"{P} {B} {cs} counters ndt_runs {c['ndt_runs']} ndt_evals {c['ndt_evaluations']} accepted {c['ndt_accepted']}", flush=True)
        r.e.close()


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--child":
        return child(sys.argv[2])
    outs = []
    for lib in sys.argv[1:3]:
        res = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib], capture_output=True, text=True, timeout=900)
        if res.returncode != 0:
            print(res.stderr[-2000:])
            return 2
        outs.append([ln for ln in res.stdout.splitlines() if ln and ln[0].isdigit()])
    a, b = outs
    bad = [(x, y) for x, y in zip(a, b) if x != y]
    print(f"{len(a)} / {len(b)} lines, {len(bad)} differ")
    for x, y in bad[:6]:
        print("  A", x)
        print("  B", y)
    return 1 if bad or len(a) != len(b) else 0


if __name__ == "__main__":
    sys.exit(main())

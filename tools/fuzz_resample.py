"""Developer fuzz: systematic resampling (main.py:46-79) of the HIP path against the oracle's, ancestors bit for bit, on
random weight vectors (flat, peaked, with -inf and zeros, below and above the spread trigger) and particle counts."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import rbpf_oracle as orc
from thesis_amd import engine

def run(N=200, SEED=1, verbose=True):
    """N random weight vectors; returns the number whose ancestors differ from the oracle's."""
    rng = np.random.Generator(np.random.PCG64(SEED))
    bad = 0
    engines = {}
    for case in range(N):
        P = int(rng.choice([1, 2, 3, 17, 64, 255, 1024, 1025, 4096, 5000]))
        if P not in engines:
            engines[P] = engine.ParticleEngine(P, max_beams=8, pool_tiles=2 * P + 8)
        e = engines[P]
        style = rng.random()
        if style < 0.3:
            w = 1000 + rng.normal(0, rng.choice([1, 50, 300]), P)
        elif style < 0.6:
            w = np.exp(rng.normal(0, 3, P)) * rng.choice([1, 100, 1e4])
        elif style < 0.8:
            w = rng.uniform(-500, 500, P)
        else:
            w = 1000 + rng.normal(0, 200, P)
            w[rng.random(P) < 0.1] = -np.inf
            w[rng.random(P) < 0.1] = 0.0
        u = float(rng.random())
        e.set_state(weights=w)
        try:
            did_ref, idx_ref = orc.resample_indices([np.longdouble(x) for x in w], u)
            ref_asserts = False
        except (AssertionError, ValueError, ZeroDivisionError, OverflowError):   # main.py:63-67 fails (e.g. every weight -inf or 0: slice 0, NaN)
            ref_asserts = True
        try:
            did, idx = e.resample(u)
        except engine.RbpfError as ex:                           # RBPF_ESTATE is the engine's form of that assertion
            if not ref_asserts:
                bad += 1
                print("case", case, "P", P, "style", style, "engine error where the reference has none:", ex)
            e.close()
            del engines[P]
            continue
        if ref_asserts:
            bad += 1
            print("case", case, "P", P, "style", style, "the reference asserts, the engine does not")
            continue
        if bool(did) != bool(did_ref) or (did and list(idx) != list(idx_ref)):
            bad += 1
            print("case", case, "P", P, "style", style, "did", did, did_ref, "first diff", next((i for i, (a, b) in enumerate(zip(idx, idx_ref)) if a != b), None))
    for e in engines.values():
        e.close()
    if verbose:
        print("done", N, "cases, mismatching:", bad)

    return bad


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 1)

"""Fuzz: several feature streams (ghmm_estep_streams / ghmm_score_streams, TF:1406-1409 ...) against
the oracle on seeded random shapes: 2-4 streams with their own mixtures and coefficient counts.
usage: fuzz_streams.py [n_seeds]"""
import sys
import numpy as np
sys.path.insert(0, "tests")
from _load import load_pkg
import test_gpu_parity as T

G = load_pkg().ghmm
ctx = G.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
bad = 0
for seed in range(n):
    rng = np.random.default_rng(41000 + seed)
    P = int(rng.integers(2, 5)); N = int(rng.integers(1, 17))
    lens = [int(x) for x in rng.integers(N, N + 100, size=int(rng.integers(1, 6)))]
    hms, Xs = [], []
    for p in range(P):
        M = int(rng.choice([1, 2, 3, 4, 8, 16])); D = int(rng.choice([3, 5, 9, 13, 36, 39, 40]))
        hm, X, lens_a = T.synth_case(G, N, M, D, lens, first=17 * p + seed, seed=seed, perturb=float(rng.choice([0.02, 0.1])))
        if p:
            hm.A[:] = hms[0].A
        hms.append(hm); Xs.append(X)
    try:
        T._estep_streams_vs_oracle(G, ctx, hms, Xs, lens_a, f"seed {seed} P={P} N={N} shapes={[(h.M, h.D) for h in hms]} lens={lens}")
    except AssertionError as e:
        bad += 1
        print(str(e)[:260])
print(f"{n} multi-stream shapes against the oracle, {bad} disagreements")

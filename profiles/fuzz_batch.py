"""Fuzz: ghmm_score_batch (the recogniser's vocabulary loop RF:326-374 in two launches) against the
oracle's forward score of every (model, utterance) pair, on seeded random vocabularies.
usage: fuzz_batch.py [n_seeds]"""
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
    rng = np.random.default_rng(51000 + seed)
    K = int(rng.integers(1, 7)); M = int(rng.choice([1, 2, 3, 8, 16])); D = int(rng.choice([5, 9, 13, 36, 39, 40]))
    lens = [int(x) for x in rng.integers(20, 120, size=int(rng.integers(1, 6)))]
    hms = []
    for k in range(K):
        N = int(rng.integers(1, 13))
        hm, X, lens_a = T.synth_case(G, N, M, D, lens, first=(seed if k == 0 else 100 + k), seed=seed + k,
                                     perturb=float(rng.choice([0.02, 0.1, 0.3])), dense_A=bool(rng.integers(0, 2)))
        if k == 0:
            X0 = X
        hms.append(hm)
    models = [ctx.model(h) for h in hms]
    corpus = ctx.corpus(X0, lens_a)
    try:
        got = ctx.score_batch(models, corpus)
        off = np.concatenate([[0], np.cumsum(lens_a)])
        for k, hm in enumerate(hms):
            ref = np.array([T.O.score(hm, X0[off[u]:off[u + 1]]) for u in range(len(lens_a))])
            T.assert_close(np.asarray(got).reshape(K, -1)[k], ref, rtol=1e-9,
                           what=f"seed {seed} K={K} model {k} N={hm.N} M={M} D={D} lens={lens}")
    except AssertionError as e:
        bad += 1
        print(str(e)[:260])
    finally:
        for o in models + [corpus]:
            o.close()
print(f"{n} vocabularies against the oracle, {bad} disagreements")

"""Fuzz: grid-shaping options (GHMM_OPT_CUS, GHMM_OPT_PARTIALS) must not change the statistics:
seeded random shapes and option values against the oracle.
usage: fuzz_options.py [n_seeds]"""
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
    rng = np.random.default_rng(71000 + seed)
    N, M, D = T.fuzz_shape(rng, bool(rng.integers(0, 2)))
    lens = [int(x) for x in rng.integers(N, N + 150, size=int(rng.integers(1, 9)))]
    hm, X, lens = T.synth_case(G, N, M, D, lens, seed=seed, perturb=0.1)
    ref, _ = T.O.estep(hm, X, lens, dumps=False)
    model, corpus = ctx.model(hm), ctx.corpus(X, lens)
    stats = ctx.stats(N, M, D)
    try:
        for _ in range(3):
            cus = int(rng.choice([0, 1, 2, 7, 33, 64, 200])); parts = int(rng.choice([0, 1, 3, 17, 300]))
            ctx.set_option(G.OPT_CUS, cus); ctx.set_option(G.OPT_PARTIALS, parts)
            ctx.estep(model, corpus, stats)
            T.assert_close(stats.download(), ref, what=f"seed {seed} N={N} M={M} D={D} lens={list(map(int, lens))} cus={cus} partials={parts}")
    except AssertionError as e:
        bad += 1
        print(str(e)[:260])
    finally:
        ctx.set_option(G.OPT_CUS, 0); ctx.set_option(G.OPT_PARTIALS, 0)
        for o in (model, corpus, stats):
            o.close()
print(f"{n} shapes x 3 option settings against the oracle, {bad} disagreements")

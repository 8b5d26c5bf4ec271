"""Fuzz: GHMM_OPT_ROBUST (per-frame max-normalised densities, an extension) on harsh shapes — models
far from their data, where the reference's linear-domain densities underflow.  Where the oracle
(linear domain) stays finite the robust statistics must agree with it; where it does not, they
must still be finite.
usage: fuzz_robust.py [n_seeds]"""
import sys
import numpy as np
sys.path.insert(0, "tests")
from _load import load_pkg
import test_gpu_parity as T

G = load_pkg().ghmm
ctx = G.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = agree = survived = 0
for seed in range(n):
    rng = np.random.default_rng(9000 + seed)
    N, M, D = T.fuzz_shape(rng, False)
    lens = [int(x) for x in rng.integers(N, N + 120, size=int(rng.integers(1, 7)))]
    dense, delta = bool(rng.integers(0, 2)), int(rng.integers(0, 4))
    hm, X, lens = T.synth_case(G, N, M, D, lens, dense_A=dense, seed=seed, perturb=float(rng.choice([0.6, 1.0])))
    k = float(rng.choice([1.0, 3.0, 9.0])); hm.inv_var *= k; hm.det /= k ** D
    ref, _ = T.O.estep(hm, X, lens, delta=delta, dumps=False)
    model, corpus = ctx.model(hm), ctx.corpus(X, lens)
    stats = ctx.stats(N, M, D)
    ctx.set_option(G.OPT_DELTA, delta); ctx.set_option(G.OPT_ROBUST, 1)
    tag = f"seed {seed} N={N} M={M} D={D} lens={list(map(int, lens))} dense={dense} delta={delta}"
    try:
        ctx.estep(model, corpus, stats)
        got = stats.download()
        if np.all(np.isfinite(ref)):
            T.assert_close(got, ref, rtol=1e-7, what=tag + ": robust statistics against the linear-domain oracle")
            agree += 1
        else:
            assert np.all(np.isfinite(got)), tag + ": robust statistics not finite"
            survived += 1
    except AssertionError as e:
        bad += 1
        print(str(e)[:260])
    finally:
        ctx.set_option(G.OPT_DELTA, 1); ctx.set_option(G.OPT_ROBUST, 0)
        for o in (model, corpus, stats):
            o.close()
print(f"{n} harsh shapes in robust mode: {agree} agree with the finite oracle, {survived} finite where the oracle is not, {bad} failures")

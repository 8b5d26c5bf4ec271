"""Fuzz: the product's E-step + M-step (default tier) against the oracle (oracle/ghmm_oracle.c, the
CPU restatement pinned to the reference) on seeded random shapes.  usage: fuzz_oracle.py [n_seeds]"""
import sys
import numpy as np
sys.path.insert(0, "tests")
from _load import load_pkg
import test_gpu_parity as T
import oracle_lib as O

G = load_pkg().ghmm
ctx = G.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0
for seed in range(n):
    rng = np.random.default_rng(9000 + seed)
    N, M, D = int(rng.integers(1, 21)), int(rng.integers(1, 12)), int(rng.integers(1, 45))
    lens = [int(x) for x in rng.integers(N, N + 120, size=int(rng.integers(1, 7)))]  # every utterance can reach the last state
    dense, delta = bool(rng.integers(0, 2)), int(rng.integers(0, 4))
    hm, X, lens = T.synth_case(G, N, M, D, lens, dense_A=dense, seed=seed, perturb=float(rng.choice([0.02, 0.1, 0.3])))
    ref_stats, ref = O.estep(hm, X, lens, delta=delta)
    model, corpus = ctx.model(hm), ctx.corpus(X, lens)
    F = corpus.frames
    ctx.set_option(G.OPT_DELTA, delta)
    stats = ctx.stats(N, M, D)
    ctx.estep(model, corpus, stats)
    got = dict(stats=stats.download(), beta=ctx.fetch(G.BUF_BETA, (F, N)), alpha=ctx.fetch(G.BUF_ALPHA, (F, N)),
               ll=ctx.fetch(G.BUF_LOGLIK, (len(lens),)))
    exp = dict(stats=ref_stats, beta=ref["beta"], alpha=ref["alpha"], ll=ref["loglik"])
    ctx.mstep(model, stats)
    new, ref_new = model.get(), O.mstep(hm, ref_stats)
    checks = [(k, got[k], exp[k], 1e-8) for k in ("ll", "alpha", "beta", "stats")]
    checks += [("mstep." + nm, a, b, 1e-7) for nm, a, b in zip(("A", "c", "mean", "inv_var", "det"), new.arrays(), ref_new.arrays())]
    for k, a, b, tol in checks:
        try:
            T.assert_close(a, b, rtol=tol, what=k)
        except AssertionError as e:
            bad += 1
            print(f"seed {seed}: N={N} M={M} D={D} lens={list(map(int, lens))} dense={dense} delta={delta}: {str(e)[:160]}")
    for o in (model, corpus, stats):
        o.close()
ctx.set_option(G.OPT_DELTA, 1)
print(f"{n} shapes against the oracle, {bad} disagreements")

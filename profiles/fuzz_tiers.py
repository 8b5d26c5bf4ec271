"""Fuzz: default tier (matrix cores, paired scans) against the vector-ALU / reference-order tier
on seeded random shapes; prints every disagreement beyond 1e-9.  usage: fuzz_tiers.py [n_seeds]"""
import sys
import numpy as np
sys.path.insert(0, "tests")
from _load import load_pkg
import test_gpu_parity as T

G = load_pkg().ghmm
ctx = G.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = 0
for seed in range(n):
    rng = np.random.default_rng(5000 + seed)
    N, M, D = int(rng.integers(1, 25)), int(rng.integers(1, 12)), int(rng.integers(1, 45))
    lens = [int(x) for x in rng.integers(1, 140, size=int(rng.integers(1, 8)))]
    if seed % 5 == 0:
        lens.insert(int(rng.integers(0, len(lens) + 1)), 0)
    dense, delta, robust = bool(rng.integers(0, 2)), int(rng.integers(0, 4)), int(rng.integers(0, 2))
    hm, X, lens = T.synth_case(G, N, M, D, lens, dense_A=dense, seed=seed, perturb=float(rng.choice([0.02, 0.1, 0.3])))
    model, corpus = ctx.model(hm), ctx.corpus(X, lens)
    F = corpus.frames
    out = {}
    ctx.set_option(G.OPT_DELTA, delta)
    ctx.set_option(G.OPT_ROBUST, robust)
    for tier in (1, 0):
        ctx.set_option(G.OPT_KERNELS, tier)
        stats = ctx.stats(N, M, D)
        ctx.estep(model, corpus, stats)
        out[tier] = dict(stats=stats.download(), gamma=ctx.fetch(G.BUF_GAMMA, (F, N)),
                         beta=ctx.fetch(G.BUF_BETA, (F, N)), ll=ctx.fetch(G.BUF_LOGLIK, (len(lens),)))
        stats.close()
    # the default tier with its recursions in separate launches (GHMM_OPT_FUSED_SCAN 2): per-frame results bit for bit
    ctx.set_option(G.OPT_FUSED_SCAN, 2)
    stats = ctx.stats(N, M, D)
    ctx.estep(model, corpus, stats)
    sep = dict(stats=stats.download(), gamma=ctx.fetch(G.BUF_GAMMA, (F, N)),
               beta=ctx.fetch(G.BUF_BETA, (F, N)), ll=ctx.fetch(G.BUF_LOGLIK, (len(lens),)))
    stats.close()
    ctx.set_option(G.OPT_FUSED_SCAN, 0)
    for k in ("ll", "gamma", "beta", "stats"):
        # (log P and the statistics are sums over an utterance's chunks: fewer chunks in the one launch
        # when the utterances are short)
        same = np.array_equal(out[0][k], sep[k], equal_nan=True)
        if not same and k in ("ll", "stats") and np.array_equal(np.isnan(out[0][k]), np.isnan(sep[k])):
            try:
                T.assert_close(out[0][k], sep[k], rtol=1e-12, what=k)
                same = True
            except AssertionError:
                pass
        if not same:
            bad += 1
            print(f"seed {seed}: N={N} M={M} D={D} lens={list(lens)} dense={dense} delta={delta} robust={robust}: {k}: one launch and separate launches differ")
    for k in ("ll", "gamma", "beta", "stats"):
        try:
            T.assert_close(out[0][k], out[1][k], rtol=1e-9, what=k)
        except AssertionError as e:
            bad += 1
            print(f"seed {seed}: N={N} M={M} D={D} lens={list(lens)} dense={dense} delta={delta} robust={robust}: {k}: {str(e)[:200]}")
    model.close(); corpus.close()
ctx.set_option(G.OPT_KERNELS, 0); ctx.set_option(G.OPT_DELTA, 1); ctx.set_option(G.OPT_ROBUST, 0)
print(f"{n} shapes, {bad} disagreements")

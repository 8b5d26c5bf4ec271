"""beta^ of utterances shorter than the model (profiles/fuzz_oracle.py N short): ours against the oracle's."""
import sys, numpy as np
sys.path.insert(0, "tests")
from _load import load_pkg
import test_gpu_parity as T
G = load_pkg().ghmm
O = T.O
ctx = G.Context(0)
wide = "wide" in sys.argv
for seed in [int(a) for a in sys.argv[1:] if a.isdigit()]:
    rng = np.random.default_rng((19000 if wide else 9000) + seed)
    N, M, D = T.fuzz_shape(rng, wide)
    lens = [int(x) for x in rng.integers(N, N + 120, size=int(rng.integers(1, 7)))]
    lens = [int(x) for x in rng.integers(1, N + 31, size=len(lens) + 2)]
    dense, delta = bool(rng.integers(0, 2)), int(rng.integers(0, 4))
    hm, X, lens = T.synth_case(G, N, M, D, lens, dense_A=dense, seed=seed, perturb=float(rng.choice([0.02, 0.1, 0.3])))
    ref_stats, ref = O.estep(hm, X, lens, delta=delta)
    model, corpus = ctx.model(hm), ctx.corpus(X, lens)
    F = corpus.frames
    ctx.set_option(G.OPT_DELTA, delta)
    stats = ctx.stats(N, M, D)
    ctx.estep(model, corpus, stats)
    got_b = ctx.fetch(G.BUF_BETA, (F, N)); got_s = stats.download()
    print(f"seed {seed}: N={N} M={M} D={D} lens={list(map(int, lens))} dense={dense}")
    off = np.concatenate([[0], np.cumsum(lens)])
    for u, Tn in enumerate(lens):
        g, r = got_b[off[u]:off[u+1]], ref["beta"][off[u]:off[u+1]]
        with np.errstate(all="ignore"):
            sc = np.abs(r).max(axis=1, keepdims=True); sc[sc == 0] = 1
            err = np.abs(g - r) / sc
        err[np.isnan(err)] = np.inf
        print(f"  utt {u} T={Tn} {'(shorter than the model)' if Tn < N else ''}: beta^ worst {err.max():.2e}; ref max {np.abs(r).max():.3e} finite {np.isfinite(r).all()}; got max {np.abs(g[np.isfinite(g)]).max() if np.isfinite(g).any() else float('nan'):.3e} finite {np.isfinite(g).all()}")
    with np.errstate(all="ignore"):
        rel = np.abs(got_s - ref_stats) / np.maximum(np.abs(ref_stats).max(), 1e-300)
    print("  statistics: worst error relative to the largest entry", np.nanmax(rel), "NaN got/ref", int(np.isnan(got_s).sum()), int(np.isnan(ref_stats).sum()))
    ctx.set_option(G.OPT_DELTA, 1)
    for o in (model, corpus, stats): o.close()

import sys, numpy as np
sys.path.insert(0, "tests")
from _load import load_pkg
import test_gpu_parity as T
G = load_pkg().ghmm
O = T.O
ctx = G.Context(0)
for seed in (433, 555):
    rng = np.random.default_rng(19000 + seed)
    N, M, D = T.fuzz_shape(rng, True)
    lens = [int(x) for x in rng.integers(N, N + 120, size=int(rng.integers(1, 7)))]
    dense, delta = bool(rng.integers(0, 2)), int(rng.integers(0, 4))
    hm, X, lens = T.synth_case(G, N, M, D, lens, dense_A=dense, seed=seed, perturb=float(rng.choice([0.02, 0.1, 0.3])))
    ref_stats, ref = O.estep(hm, X, lens, delta=delta)
    model, corpus = ctx.model(hm), ctx.corpus(X, lens)
    ctx.set_option(G.OPT_DELTA, delta)
    stats = ctx.stats(N, M, D)
    ctx.estep(model, corpus, stats)
    got = stats.download()
    ll = ctx.fetch(G.BUF_LOGLIK, (len(lens),))
    print("seed", seed, N, M, D, list(lens), "dense", dense, "delta", delta)
    print(" loglik ref", ref["loglik"], "\n loglik got", ll)
    print(" NaN in ref stats", int(np.isnan(ref_stats).sum()), "of", ref_stats.size, "; NaN in got", int(np.isnan(got).sum()))
    b = ref["b"]; off = np.concatenate([[0], np.cumsum(lens)])
    for u in range(len(lens)):
        bu = b[off[u]:off[u+1]]
        dead = np.where((bu == 0).all(axis=1))[0]
        print("  utt", u, "frames with every b = 0:", dead[:5], "min over frames of max_i b:", bu.max(axis=1).min())
    ctx.set_option(G.OPT_DELTA, 1)
    idx = np.where(np.isnan(got))[0]
    G_ = N * M
    o_c = N * N + 2 * N; o_mu = o_c + G_; o_var = o_mu + G_ * D
    def where(i):
        if i < N * N: return ("num_a", i // N, i % N)
        if i < o_c: return ("den", i - N * N)
        if i < o_mu: return ("num_c", i - o_c)
        if i < o_var: return ("num_mu", (i - o_mu) // D, (i - o_mu) % D)
        if i < o_var + G_ * D: return ("num_var", (i - o_var) // D, (i - o_var) % D)
        return ("tail", i)
    print("  NaN at", [where(int(i)) for i in idx[:6]], "...", [where(int(i)) for i in idx[-3:]])
    for tier in (1, 2):
        ctx.set_option(G.OPT_KERNELS, tier); ctx.set_option(G.OPT_DELTA, delta)
        ctx.estep(model, corpus, stats); g2 = stats.download()
        print("  tier", tier, "NaNs", int(np.isnan(g2).sum()))
    ctx.set_option(G.OPT_KERNELS, 0); ctx.set_option(G.OPT_DELTA, 1)
    print("  inv_var range", hm.inv_var.min(), hm.inv_var.max(), " mean range", hm.mean.min(), hm.mean.max())

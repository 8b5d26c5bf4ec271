import sys, numpy as np
sys.path.insert(0, "tests")
from _load import load_pkg
G = load_pkg().ghmm
N, M, D, U, T = 10, 8, 39, 1000, 300
mean, std = G.synth_truth(N, M, D)
lens = np.full(U, T, dtype=np.int32)
X = G.synth_utterances(mean, std, lens)
hm = G.synth_start_model(mean, std, 0.05)
ctx = G.Context(0)
model, corpus = ctx.model(hm), ctx.corpus(X, lens)
stats = ctx.stats(N, M, D)
for it in range(3):
    ctx.estep(model, corpus, stats)
    g = ctx.fetch(G.BUF_GAMMA, (U * T, N))
    post = ctx.fetch(G.BUF_POST, (U * T, N, M))
    w = g[:, :, None] * post
    print("iteration", it, "gamma: ==0 %.3f  <1e-290 %.3f  <1e-100 %.3f  <1e-30 %.3f  <1e-16 %.3f" % tuple(
        (g < th).mean() if th else (g == 0).mean() for th in (0, 1e-290, 1e-100, 1e-30, 1e-16)),
        "| w: ==0 %.3f <1e-290 %.3f <1e-30 %.3f" % ((w == 0).mean(), (w < 1e-290).mean(), (w < 1e-30).mean()),
        "| post ==0 %.3f <1e-290 %.3f" % ((post == 0).mean(), (post < 1e-290).mean()))
    ctx.mstep(model, stats)

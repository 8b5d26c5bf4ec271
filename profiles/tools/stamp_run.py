import sys, numpy as np
sys.path.insert(0, "tests")
from _load import load_pkg
pkg = load_pkg(); G, em = pkg.ghmm, pkg.em
N, M, D, U, T = 10, 8, 39, 1000, 300
mean, std = G.synth_truth(N, M, D)
lens = np.full(U, T, dtype=np.int32)
X = G.synth_utterances(mean, std, lens)
ctx = G.Context(0)
corpus = ctx.corpus(X, lens)
model = ctx.model(G.synth_start_model(mean, std, 0.05))
stats = ctx.stats(N, M, D)
for _ in range(3):
    ctx.estep(model, corpus, stats); ctx.sync()

import sys, time, numpy as np
sys.path.insert(0, "tests")
from _load import load_pkg
G = load_pkg().ghmm
N, M, D, U, T = 10, 8, 39, 1000, 300
mean, std = G.synth_truth(N, M, D)
lens = np.full(U, T, dtype=np.int32)
X = G.synth_utterances(mean, std, lens)
hm = G.HostModel.init_from(X, lens, N, M)
ctx = G.Context(0)
corpus = ctx.corpus(X, lens)
model = ctx.model(hm)
for parts in (0, 512, 1024, 2048, 4096, 0):
    ctx.set_option(G.OPT_PARTIALS, parts)
    model.init_from(corpus, fetch=False); ctx.sync()
    t = time.perf_counter()
    for _ in range(3): model.init_from(corpus, fetch=False)
    ctx.sync(); td = (time.perf_counter() - t) / 3
    dm = model.init_from(corpus)
    err = max(np.abs(a - b).max() / np.abs(b).max() for a, b in zip(dm.arrays(), hm.arrays()))
    print(f"partials {parts}: device init {td*1e3:.2f} ms, max rel diff vs host {err:.2e}")
ctx.set_option(G.OPT_PARTIALS, 0)

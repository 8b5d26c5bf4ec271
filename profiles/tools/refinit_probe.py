import sys, time, numpy as np
sys.path.insert(0, "tests")
from _load import load_pkg
pkg = load_pkg(); G, em = pkg.ghmm, pkg.em
N, M, D, U, T = 10, 8, 39, 1000, 300
mean, std = G.synth_truth(N, M, D)
lens = np.full(U, T, dtype=np.int32)
X = G.synth_utterances(mean, std, lens)
ctx = G.Context(0)
corpus = ctx.corpus(X, lens)
for mode in (0, 1, 2, 0):
    ctx.set_option(G.OPT_VEC_STATS, mode)
    model = ctx.model(G.synth_start_model(mean, std, 0.05))
    model.init_from(corpus, fetch=False)
    be = em.HipBackend(G, ctx, model, corpus); drv = em.EMDriver(be)
    for _ in range(2): drv.step()
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(8): drv.step()
    ctx.sync(); dt = (time.perf_counter() - t0) / 8
    print("VEC_STATS", mode, "ms per iteration 3-10:", round(dt * 1e3, 4))
    be.stats.close(); model.close()

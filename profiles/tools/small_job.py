# A whole training job at the reference's own scale: one word model, a few hundred short utterances
import sys, time, numpy as np
sys.path.insert(0, "tests")
from _load import load_pkg
pkg = load_pkg(); G, em = pkg.ghmm, pkg.em
ctx = G.Context(0)
for (N, M, D, U, T) in [(6, 4, 13, 300, 60), (6, 4, 13, 2000, 60), (10, 8, 39, 300, 60), (10, 8, 39, 1000, 300)]:
    mean, std = G.synth_truth(N, M, D)
    lens = np.full(U, T, dtype=np.int32)
    X = G.synth_utterances(mean, std, lens)
    start = G.synth_start_model(mean, std, 0.05)
    for rep in range(3):
        ctx.sync(); t0 = time.perf_counter()
        corpus = ctx.corpus(X, lens)
        t1 = time.perf_counter()
        model = ctx.model(start)
        model.init_from(corpus, fetch=False)
        ctx.sync(); t2 = time.perf_counter()
        be = em.HipBackend(G, ctx, model, corpus); drv = em.EMDriver(be)
        prev = None; it = 0
        while it < 100:
            drv.step(); it += 1
            lp = be.loglik()
            if prev is not None and abs((lp - prev) / lp) <= 1e-3: break
            prev = lp
        ctx.sync(); t3 = time.perf_counter()
        if rep == 2:
            print(f"{N}x{M} D={D}, {U} x {T} frames: upload {1e3*(t1-t0):.2f} ms, initial model {1e3*(t2-t1):.2f} ms, "
                  f"{it} EM iterations {1e3*(t3-t2):.2f} ms ({1e3*(t3-t2)/it:.3f} each, log P read every iteration), job {1e3*(t3-t0):.2f} ms")
        be.stats.close(); model.close(); corpus.close()

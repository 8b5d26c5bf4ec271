# How well do two independent E-steps on two streams (two contexts, half the corpus each) overlap?
# Best case for a pipelined iteration (emission of one half beside the scans of the other).
import sys, time, numpy as np
sys.path.insert(0, "tests")
from _load import load_pkg
pkg = load_pkg(); G = pkg.ghmm
N, M, D, U, T = 10, 8, 39, 1000, 300
mean, std = G.synth_truth(N, M, D)
start = G.synth_start_model(mean, std, 0.05)
lens = np.full(U, T, dtype=np.int32)
X = G.synth_utterances(mean, std, lens)
def run(ctxs, corpora, models, stats, steps):
    for c in ctxs: c.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        for c, co, m, s in zip(ctxs, corpora, models, stats):
            c.estep(m, co, s)
    for c in ctxs: c.sync()
    return 1e3 * (time.perf_counter() - t0) / steps
for parts in (1, 2, 4):
    ctxs = [G.Context(0) for _ in range(parts)]
    per = U // parts
    corpora = [c.corpus(X[k * per * T:(k + 1) * per * T], lens[k * per:(k + 1) * per]) for k, c in enumerate(ctxs)]
    models = [c.model(start) for c in ctxs]
    stats = [c.stats(N, M, D) for c in ctxs]
    run(ctxs, corpora, models, stats, 100)
    print(parts, "contexts x", per, "utterances: E-step of the whole corpus", round(run(ctxs, corpora, models, stats, 300), 4), "ms")
    one = run(ctxs[:1], corpora[:1], models[:1], stats[:1], 300)
    print("   one of them alone:", round(one, 4), "ms")
    for o in stats + models + corpora: o.close()
    for c in ctxs: c.close()

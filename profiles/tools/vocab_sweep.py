# The recogniser's vocabulary loop (ghmm_score_batch) on isolated-word shapes: W word models x U utterances
import sys, time, numpy as np
sys.path.insert(0, "tests")
from _load import load_pkg
pkg = load_pkg(); G = pkg.ghmm
ctx = G.Context(0)
cases = [(10, 6, 4, 13, 2000, 40), (50, 6, 4, 13, 2000, 40), (50, 6, 4, 39, 2000, 40), (20, 10, 8, 39, 2000, 100), (100, 5, 2, 13, 5000, 30)]
for (W, N, M, D, U, T) in cases:
    hms = []
    for k in range(W):
        mean, std = G.synth_truth(N, M, D, seed=100 + k) if "seed" in G.synth_truth.__code__.co_varnames else G.synth_truth(N, M, D)
        hms.append(G.synth_start_model(mean, std, 0.05 + 0.001 * k))
    lens = np.full(U, T, dtype=np.int32)
    X = G.synth_utterances(mean, std, lens)
    corpus = ctx.corpus(X, lens)
    models = [ctx.model(h) for h in hms]
    ctx.score_batch(models, corpus)
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(5): ctx.score_batch(models, corpus)
    ctx.sync(); wall = 1e3 * (time.perf_counter() - t0) / 5
    ctx.set_option(G.OPT_TIMING, 1); ctx.kernel_times_reset()
    for _ in range(3): ctx.score_batch(models, corpus)
    kt = {k: round(1e3 * ms / 3, 1) for k, (ms, n) in ctx.kernel_times().items() if n}
    ctx.set_option(G.OPT_TIMING, 0)
    print(f"W={W} words of {N}x{M} D={D}, {U} utterances x {T} frames: {wall:.3f} ms per vocabulary pass "
          f"({U * T * W / wall / 1e6:.1f} G frame-models/s), kernels us {kt}")
    for m_ in models: m_.close()
    corpus.close()

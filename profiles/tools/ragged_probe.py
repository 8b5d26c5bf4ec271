# fixed-length against ragged corpus, same model: wall per iteration and per-kernel event times
import sys, time, numpy as np
sys.path.insert(0, "tests")
from _load import load_pkg
pkg = load_pkg(); G, em = pkg.ghmm, pkg.em
N, M, D, U, T = 10, 8, 39, 1000, 300
mean, std = G.synth_truth(N, M, D)
start = G.synth_start_model(mean, std, 0.05)
ctx = G.Context(0)
rng = np.random.default_rng(20260104)
cases = {"fixed": np.full(U, T, dtype=np.int32), "ragged": rng.integers(100, 501, size=U).astype(np.int32)}
if len(sys.argv) > 1:
    cases = {k: v for k, v in cases.items() if k in sys.argv[1:]}
for name, lens in cases.items():
    X = G.synth_utterances(mean, std, lens)
    corpus = ctx.corpus(X, lens)
    model = ctx.model(start)
    be = em.HipBackend(G, ctx, model, corpus); drv = em.EMDriver(be)
    for _ in range(50): drv.step()
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(100): drv.step()
    ctx.sync(); wall = 1e3 * (time.perf_counter() - t0) / 100
    ctx.set_option(G.OPT_TIMING, 1); ctx.kernel_times_reset()
    for _ in range(20): drv.step()
    kt = {k: round(1e3 * ms / 20, 1) for k, (ms, n) in ctx.kernel_times().items() if n}
    ctx.set_option(G.OPT_TIMING, 0)
    fr = int(lens.sum())
    print(name, "frames", fr, "ms/iteration", round(wall, 4), "ns/frame", round(1e6 * wall / fr, 4), "kernels us", kt)
    be.stats.close(); model.close(); corpus.close()

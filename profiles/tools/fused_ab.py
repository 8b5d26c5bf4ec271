# GHMM_OPT_FUSED_SCAN: the one-launch scans + combine against the separate launches, same process
import sys, time, numpy as np
sys.path.insert(0, "tests")
from _load import load_pkg
pkg = load_pkg(); G, em = pkg.ghmm, pkg.em
N, M, D = 10, 8, 39
mean, std = G.synth_truth(N, M, D)
start = G.synth_start_model(mean, std, 0.05)
ctx = G.Context(0)
rng = np.random.default_rng(20260104)
cases = {"fixed 1000 x 300": np.full(1000, 300, dtype=np.int32),
         "ragged 1000 x U[100,500]": rng.integers(100, 501, size=1000).astype(np.int32),
         "fixed 500 x 300": np.full(500, 300, dtype=np.int32),
         "fixed 2000 x 300": np.full(2000, 300, dtype=np.int32),
         "fixed 4000 x 300": np.full(4000, 300, dtype=np.int32),
         "fixed 12500 x 300": np.full(12500, 300, dtype=np.int32),
         "ragged 12500 x U[100,500]": rng.integers(100, 501, size=12500).astype(np.int32)}
if len(sys.argv) > 1:
    cases = {k: v for k, v in cases.items() if any(a in k for a in sys.argv[1:])}
for name, lens in cases.items():
    X = G.synth_utterances(mean, std, lens)
    corpus = ctx.corpus(X, lens)
    for mode in (2, 1, 2, 1):
        ctx.set_option(G.OPT_FUSED_SCAN, mode)
        model = ctx.model(start)
        be = em.HipBackend(G, ctx, model, corpus); drv = em.EMDriver(be)
        for _ in range(40): drv.step()
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(100): drv.step()
        ctx.sync(); wall = 1e3 * (time.perf_counter() - t0) / 100
        ctx.set_option(G.OPT_TIMING, 1); ctx.kernel_times_reset()
        for _ in range(20): drv.step()
        kt = {k: round(1e3 * ms / 20, 1) for k, (ms, n) in ctx.kernel_times().items() if n}
        ctx.set_option(G.OPT_TIMING, 0)
        print(name, "fused" if mode == 1 else "separate", "ms/iteration", round(wall, 4), "scan kernels us", kt.get("forward"), kt.get("backward"),
              "loglik", be.loglik())
        be.stats.close(); model.close()
    corpus.close()
ctx.set_option(G.OPT_FUSED_SCAN, 0)

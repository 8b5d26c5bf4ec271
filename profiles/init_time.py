"""Initial-model construction at BASELINE configs[1] size: host C vs device passes."""
import sys, time, numpy as np
sys.path.insert(0, "tests")
from _load import load_pkg
G = load_pkg().ghmm
N, M, D, U, T = 10, 8, 39, 1000, 300
mean, std = G.synth_truth(N, M, D)
lens = np.full(U, T, dtype=np.int32)
X = G.synth_utterances(mean, std, lens)
t = time.perf_counter(); hm = G.HostModel.init_from(X, lens, N, M); th = time.perf_counter() - t
ctx = G.Context(0)
corpus = ctx.corpus(X, lens)
model = ctx.model(hm)
model.init_from(corpus)
t = time.perf_counter(); dm = model.init_from(corpus); td = time.perf_counter() - t
err = max(np.abs(a - b).max() / np.abs(b).max() for a, b in zip(dm.arrays(), hm.arrays()))
print(f"host init {th*1e3:.1f} ms, device init {td*1e3:.1f} ms, max rel diff {err:.2e}")
ctx.set_option(G.OPT_TIMING, 1)
ctx.kernel_times_reset()
t = time.perf_counter(); model.init_from(corpus); td = time.perf_counter() - t
print(f"with kernel timers: {td*1e3:.1f} ms;", {k: (round(v[0], 3), v[1]) for k, v in ctx.kernel_times().items() if v[1]})

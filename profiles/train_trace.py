"""configs[1] corpus, EM from the +-15 % perturbed ground truth until the reference's convergence
rule stops it (or 60 iterations): log P per frame after every E-step, wall time per iteration."""
import sys, time, numpy as np
sys.path.insert(0, "tests")
from _load import load_pkg
G = load_pkg().ghmm
N, M, D, U, T = 10, 8, 39, 1000, 300
mean, std = G.synth_truth(N, M, D)
lens = np.full(U, T, dtype=np.int32)
X = G.synth_utterances(mean, std, lens)
hm = G.synth_start_model(mean, std, 0.15)
ctx = G.Context(0)
model, corpus = ctx.model(hm), ctx.corpus(X, lens)
stats = ctx.stats(N, M, D)
old, trace = 1.0, []
t0 = time.perf_counter()
for it in range(60):
    ctx.estep(model, corpus, stats)
    p = stats.loglik()[0]   # 16 bytes (ghmm_stats_loglik), not the whole vector
    trace.append(p / (U * T))
    var = abs((old - p) / old)
    if not (var > 1e-3):
        break
    old = p
    ctx.mstep(model, stats)
dt = time.perf_counter() - t0
print("iterations", len(trace), "wall", round(dt * 1e3, 1), "ms;", "log P / frame:", " ".join(f"{x:.4f}" for x in trace))
print("finite model:", all(np.isfinite(a).all() for a in model.get().arrays()))

# the reference's own starting point (creating_initial_model, TF:732-1317, on the device): its
# variance-floored "needle" components send the first iterations through the direct-form kernels
ctx.set_option(G.OPT_TIMING, 1)
t0 = time.perf_counter()
hm0 = model.init_from(corpus)
t_init = time.perf_counter() - t0
trace, times = [], []
for it in range(12):
    ctx.kernel_times_reset()
    t0 = time.perf_counter()
    ctx.estep(model, corpus, stats)
    p = stats.loglik()[0]   # 16 bytes (ghmm_stats_loglik), not the whole vector
    ctx.mstep(model, stats)
    times.append((time.perf_counter() - t0) * 1e3)
    trace.append(p / (U * T))
    if it in (0, 6):
        print("  iteration", it, "kernels (ms):", {k: round(v[0], 3) for k, v in ctx.kernel_times().items() if v[1]})
print("from the reference's initial model (%.1f ms): log P / frame" % (t_init * 1e3), " ".join(f"{x:.3f}" for x in trace))
print("ms per iteration (incl. the 16-byte download of log P):", " ".join(f"{x:.2f}" for x in times))

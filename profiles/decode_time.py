import sys, time, numpy as np
sys.path.insert(0, "tests")
from _load import load_pkg
G = load_pkg().ghmm
N, M, D, U, T = 10, 8, 39, 10000, 300
mean, std = G.synth_truth(N, M, D)
lens = np.full(U, T, dtype=np.int32)
X = G.synth_utterances(mean, std, lens)
hm = G.synth_start_model(mean, std, 0.05)
ctx = G.Context(0)
model, corpus = ctx.model(hm), ctx.corpus(X, lens)
ctx.set_option(G.OPT_TIMING, 1)
for name, fn in (("viterbi", lambda: ctx.viterbi(model, corpus)), ("score", lambda: ctx.score(model, corpus))):
    fn(); fn(); ctx.kernel_times_reset()     # (the first calls size the workspace and the host buffers)
    R = 3
    t = time.perf_counter()
    for _ in range(R):
        fn()
    dt = (time.perf_counter() - t) / R
    print(name, f"{dt*1e3:.2f} ms wall (incl. D2H), {U*T/dt/1e6:.1f} Mframes/s", {k: round(v[0] / R, 3) for k, v in ctx.kernel_times().items() if v[1]})

"""Times the two directions of k_scan_pair separately (row API: ghmm_forward runs the forward
direction alone, ghmm_backward then runs the other direction + k_combine) and together
(ghmm_estep), with the library's HIP-event timers.  configs[1] shape."""
import sys, numpy as np
sys.path.insert(0, "tests")
from _load import load_pkg
G = load_pkg().ghmm
N, M, D, U, T = 10, 8, 39, 1000, 300
mean, std = G.synth_truth(N, M, D)
lens = np.full(U, T, dtype=np.int32)
X = G.synth_utterances(mean, std, lens)
hm = G.synth_start_model(mean, std, 0.05)
ctx = G.Context(0)
model, corpus = ctx.model(hm), ctx.corpus(X, lens)
stats = ctx.stats(N, M, D)
ctx.set_option(G.OPT_TIMING, 1)
ctx.emission(model, corpus, True)
for rep in range(3):
    ctx.kernel_times_reset()
    for _ in range(10):
        ctx.forward(model, corpus)
    tf = ctx.kernel_times()
    ctx.kernel_times_reset()
    for _ in range(10):
        ctx.backward(model, corpus)   # the other direction ran once: k_combine alone from the 2nd call on
    tb = ctx.kernel_times()
    ctx.kernel_times_reset()
    for _ in range(10):
        ctx.forward(model, corpus)
        ctx.backward(model, corpus)   # own backward scan + k_combine every time
    tfb = ctx.kernel_times()
    ctx.kernel_times_reset()
    for _ in range(10):
        ctx.estep(model, corpus, stats)
    te = ctx.kernel_times()
    print("forward alone", {k: round(v[0] / 10, 4) for k, v in tf.items() if v[1]},
          "| combine(beta)", {k: round(v[0] / 10, 4) for k, v in tb.items() if v[1]},
          "| forward, then own backward + combine(beta)", {k: round(v[0] / 10, 4) for k, v in tfb.items() if v[1]},
          "| estep", {k: round(v[0] / 10, 4) for k, v in te.items() if v[1]})

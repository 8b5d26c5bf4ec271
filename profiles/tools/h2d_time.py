"""Wall time of ghmm_corpus_create (frames host -> device) at configs[1] and configs[2] sizes."""
import sys, time, numpy as np
sys.path.insert(0, "tests")
from _load import load_pkg
G = load_pkg().ghmm
ctx = G.Context(0)
mean, std = G.synth_truth(10, 8, 39)
for U in (1000, 10000):
    lens = np.full(U, 300, dtype=np.int32)
    X = G.synth_utterances(mean, std, lens)
    for rep in range(3):
        t = time.perf_counter()
        c = ctx.corpus(X, lens)
        ctx.sync()
        dt = time.perf_counter() - t
        c.close()
    print(f"{U} utterances, {X.nbytes/1e6:.0f} MB: corpus_create {dt*1e3:.2f} ms = {X.nbytes/dt/1e9:.1f} GB/s")

import sys, numpy as np
sys.path.insert(0, "tests")
from _load import load_pkg
import test_gpu_parity as T
G = load_pkg().ghmm
O = T.O
ctx = G.Context(0)
seed = 124
rng = np.random.default_rng(29000 + seed)
N, M, D = T.fuzz_shape(rng, "huge")
lens = [int(x) for x in rng.integers(N, N + 120, size=int(rng.integers(1, 7)))]
lens = [int(x) for x in rng.integers(1, N + 31, size=len(lens) + 2)]
dense, delta = bool(rng.integers(0, 2)), int(rng.integers(0, 4))
hm, X, lens = T.synth_case(G, N, M, D, lens, dense_A=dense, seed=seed, perturb=float(rng.choice([0.02, 0.1, 0.3])))
ref_stats, ref = O.estep(hm, X, lens, delta=delta)
model, corpus = ctx.model(hm), ctx.corpus(X, lens)
ctx.set_option(G.OPT_DELTA, delta)
stats = ctx.stats(N, M, D)
ctx.estep(model, corpus, stats)
F = corpus.frames
beta = ctx.fetch(G.BUF_BETA, (F, N))
rb = ref["beta"]
with np.errstate(all="ignore"):
    rel = np.abs(beta - rb) / np.maximum(np.abs(rb), 1e-300)
rel[~np.isfinite(rel)] = 0
# per-frame scale as assert_frames does
idx = np.unravel_index(np.argsort(rel, axis=None)[-5:], rel.shape)
off = np.concatenate([[0], np.cumsum(lens)])
for f, i in zip(*idx):
    u = np.searchsorted(off, f, side="right") - 1
    print("frame", f, "utt", u, "t", f - off[u], "T", lens[u], "state", i, "got", beta[f, i], "ref", rb[f, i], "rel", rel[f, i],
          "frame max", np.nanmax(np.abs(rb[f])), "b next", ref["b"][min(f + 1, F - 1), i:i + 2], "beta next", rb[min(f + 1, F - 1), i:i + 2], "A", hm.A[i, i:i+2])

"""Fuzz: whole training runs — the reference's initial model (creating_initial_model, on the host,
bit-exact) followed by K fixed EM iterations on the GPU — against the oracle's, on seeded random
corpora.  From that model components collapse onto single frames within a few iterations
(variances at the 1e-5 floor), which is where the statistics classes, the per-tile offsets and the
direct-form band of the emission kernel come into play.
usage: fuzz_train.py [n_seeds] [iterations = 6] [big | huge]   (big: 16 / 32 mixtures, several chunks of
Gaussians; huge: 17 .. 200 states, the 32- and 64-lane groups and the one-wave-per-utterance kernels)"""
import sys
import numpy as np
sys.path.insert(0, "tests")
from _load import load_pkg
import test_gpu_parity as T

G = load_pkg().ghmm
O = T.O
ctx = G.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
big = "big" in sys.argv[3:]
bad = skipped = floored_cases = 0
for seed in range(n):
    rng = np.random.default_rng(61000 + seed)
    N = int(rng.integers(2, 11)); M = int(rng.choice([1, 2, 3, 4, 8])); D = int(rng.choice([5, 9, 13, 36, 39, 40]))
    if big:
        N = int(rng.integers(4, 17)); M = int(rng.choice([16, 32])); D = int(rng.choice([36, 39]))
    if "huge" in sys.argv[3:]:
        N = int(rng.choice([17, 20, 32, 33, 64, 65, 100, 200])); M = int(rng.choice([1, 2, 4])); D = int(rng.choice([5, 9, 13]))
    lens = np.asarray([int(x) for x in rng.integers(2 * N + 10, 2 * N + 90, size=int(rng.integers(3, 14)))], dtype=np.int32)
    if big:   # enough frames for every cell of the initial codebook
        lens = np.asarray([int(x) for x in rng.integers(3 * N * M // 4, N * M + 60, size=int(rng.integers(6, 14)))], dtype=np.int32)
    mean, std = G.synth_truth(N, M, D)
    X = G.synth_utterances(mean, std, lens, first_utt=seed)
    hm0 = G.HostModel.init_from(X, lens, N, M)
    ref_hm, it, _, trace = O.train(hm0, X, lens, max_iter=K, fixed_iter=True)
    if not (np.all(np.isfinite(trace)) and all(np.all(np.isfinite(a)) for a in ref_hm.arrays())):
        skipped += 1
        continue
    model, corpus = ctx.model(hm0), ctx.corpus(X, lens)
    stats = ctx.stats(N, M, D)
    tag = f"seed {seed} N={N} M={M} D={D} utterances={len(lens)} frames={int(lens.sum())}"
    try:
        got = []
        for _ in range(K):
            ctx.estep(model, corpus, stats)
            got.append(stats.download()[-2])
            ctx.mstep(model, stats)
        iv = np.asarray(model.get().arrays()[3]).reshape(N * M, D)
        floored_cases += int((iv > 9.0e4).all(axis=1).any())
        T.assert_close(got, trace, rtol=1e-8, what=tag + ": loglik trace")
        for name, a, b in zip(("A", "c", "mean", "inv_var", "det"), model.get().arrays(), ref_hm.arrays()):
            T.assert_close(a, b, rtol=1e-6, what=tag + ": model." + name)
    except AssertionError as e:
        bad += 1
        print(str(e)[:300])
    finally:
        for o in (model, corpus, stats):
            o.close()
print(f"{n} training runs of {K} iterations against the oracle: {bad} disagreements, {skipped} skipped "
      f"(reference not finite), {floored_cases} ended with a collapsed component")

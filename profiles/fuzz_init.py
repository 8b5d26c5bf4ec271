"""Fuzz: ghmm_model_init (creating_initial_model TF:732-1317 with its passes on the device) against
the host C version (bit-exact against the reference) on seeded random shapes and corpora.
usage: fuzz_init.py [n_seeds] [huge]     (huge: 17 .. 200 states)"""
import sys
import numpy as np
sys.path.insert(0, "tests")
from _load import load_pkg
import test_gpu_parity as T

G = load_pkg().ghmm
ctx = G.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
bad = 0
for seed in range(n):
    rng = np.random.default_rng(31000 + seed)
    N = int(rng.integers(1, 13)); M = int(rng.choice([1, 2, 3, 4, 5, 8, 11, 16])); D = int(rng.choice([2, 9, 13, 26, 36, 39, 40]))
    if "huge" in sys.argv[2:]:
        N = int(rng.choice([17, 20, 33, 64, 65, 100, 200])); M = int(rng.choice([1, 2, 4])); D = int(rng.choice([2, 9, 13]))
    lens = [int(x) for x in rng.integers(max(N, 2 * M), 2 * N * M + 160, size=int(rng.integers(2, 9)))]
    mean, std = G.synth_truth(N, M, D)
    lens = np.asarray(lens, dtype=np.int32)
    X = G.synth_utterances(mean, std, lens, first_utt=seed)
    host = G.HostModel.init_from(X, lens, N, M)
    corpus = ctx.corpus(X, lens)
    model = ctx.model(host)
    try:
        got = model.init_from(corpus)
        for nm, a, b in zip(("A", "c", "mean", "inv_var", "det"), got.arrays(), host.arrays()):
            T.assert_close(a, b, rtol=1e-9, floor=0.0, what=f"seed {seed} N={N} M={M} D={D} lens={list(map(int, lens))}: init.{nm}")
    except AssertionError as e:
        bad += 1
        print(str(e)[:260])
    finally:
        model.close(); corpus.close()
print(f"{n} corpora, device initial model against the host's: {bad} disagreements")

"""Fuzz: Viterbi state sequences and forward scores of the product against the oracle's (oracle/ghmm_oracle.c) on
seeded random shapes; a path that differs is reported with the score gap of the two paths
(0 or ~1e-13 relative = a genuine tie broken by emission rounding).  usage: fuzz_viterbi.py [n]"""
import sys
import numpy as np
sys.path.insert(0, "tests")
from _load import load_pkg
import test_gpu_parity as T
import oracle_lib as O

G = load_pkg().ghmm
ctx = G.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
bad = utts = 0
for seed in range(n):
    rng = np.random.default_rng(7000 + seed)
    N, M, D = int(rng.integers(1, 21)), int(rng.integers(1, 12)), int(rng.integers(1, 45))
    lens = [int(x) for x in rng.integers(1, 150, size=int(rng.integers(1, 6)))]
    dense = bool(rng.integers(0, 2))
    hm, X, lens = T.synth_case(G, N, M, D, lens, dense_A=dense, seed=seed, perturb=float(rng.choice([0.02, 0.1, 0.3])))
    model, corpus = ctx.model(hm), ctx.corpus(X, lens)
    path, score = ctx.viterbi(model, corpus)
    fwd = ctx.score(model, corpus)   # calc_alpha + calc_probability alone (the recogniser's path)
    o = 0
    for u, Tn in enumerate(lens):
        p, s = O.viterbi(hm, X[o:o + Tn])
        utts += 1
        same = np.array_equal(path[o:o + Tn], p)
        sc_ok = (score[u] == s) if not np.isfinite(s) else abs(score[u] - s) <= 1e-10 * abs(s)
        fs = O.score(hm, X[o:o + Tn])
        f_ok = (np.isnan(fs) and np.isnan(fwd[u])) or fwd[u] == fs or abs(fwd[u] - fs) <= 1e-10 * abs(fs)
        if not f_ok:
            bad += 1
            print(f"seed {seed} utt {u}: N={N} M={M} D={D} T={Tn}: forward score {fwd[u]!r} vs {fs!r}")
        if not (same and sc_ok):
            bad += 1
            print(f"seed {seed} utt {u}: N={N} M={M} D={D} T={Tn} dense={dense}: path same={same} score {score[u]!r} vs {s!r}")
        o += Tn
    model.close(); corpus.close()
print(f"{n} shapes, {utts} utterances, {bad} differ")

"""BASELINE configs[4] (2 000 tied states x 16 mix, 39-d, 1 M frames), emission only: parity of a
slice against the oracle (oracle/ghmm_oracle.c orc_emission = calc_symbol_probab / calc_gaus,
TF:1749-1841) and the kernel's rate over the full 1 M frames (the same workload as bench.py's
"config5" extra and tests/test_gpu_parity.py::test_config5_*).
usage: python profiles/config5_emission.py [frames]   -> profiles/r2_config5.txt"""
import sys
import time

import numpy as np

sys.path.insert(0, "tests")
from _load import load_pkg
import oracle_lib as O
import test_gpu_parity as T

G = load_pkg().ghmm
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
hm, centre = T._config5_model(G)
N, M, D = hm.N, hm.M, hm.D
X = T._config5_frames(centre, F, seed=2)
ctx = G.Context(0)
model = ctx.model(hm)
small = ctx.corpus(X[:96], [96])
ctx.emission(model, small, False)
got = ctx.fetch(G.BUF_B, (96, N))
ref = O.emission(hm, X[:96])
err = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-300)
print(f"parity, 96 frames x {N * M} Gaussians against the oracle: max relative error of b {err.max():.3e}")
corpus = ctx.corpus(X, [F])
ctx.set_option(G.OPT_TIMING, 1)
ctx.emission(model, corpus, False)
ctx.sync()
for rep in range(3):
    ctx.kernel_times_reset()
    t = time.perf_counter()
    ctx.emission(model, corpus, False)
    ctx.sync()
    dt = time.perf_counter() - t
    ms = ctx.kernel_times()["emission"][0]
    flop = 2.0 * F * 80 * N * M
    print(f"{F} frames x {N} states x {M} mix: kernel {ms:.2f} ms (wall {dt * 1e3:.2f}), "
          f"{flop / (ms * 1e-3) / 1e12:.2f} TFLOP/s f64 on the matrix pipe = "
          f"{flop / (ms * 1e-3) / 1e12 / 78.6:.3f} of the 78.6 TFLOP/s peak; output {8 * F * N / 1e9:.1f} GB")

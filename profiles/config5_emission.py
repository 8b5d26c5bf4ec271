"""BASELINE configs[4] shape (2000 tied states x 16 mix, 39-d), emission only: parity of a
small slice against the oracle and the kernel's rate on a larger one.
usage: python profiles/config5_emission.py [frames]"""
import ctypes
import sys
import time

import numpy as np

sys.path.insert(0, "tests")
from _load import load_pkg
import oracle_lib as O

G = load_pkg().ghmm
N, M, D = 2000, 16, 39
rng = np.random.default_rng(5)
mean = rng.normal(0, 2, (N, M, D))
std = rng.uniform(0.5, 1.5, (N, M, D))
hm = G.HostModel(np.eye(N), np.full((N, M), 1.0 / M), mean, 1.0 / std**2, np.prod(std**2, axis=2))
F = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
X = rng.normal(0, 2.2, (F, D))
ctx = G.Context(0)
model = ctx.model(hm)
small = ctx.corpus(X[:48], [48])
ctx.emission(model, small, False)
got = ctx.fetch(G.BUF_B, (48, N))
b = np.zeros((48, N))
L = O.lib()
L.orc_emission.restype = None
L.orc_emission.argtypes = [ctypes.c_int] * 4 + [ctypes.POINTER(ctypes.c_double)] * 7
L.orc_emission(N, M, D, 48, O._d(X[:48].copy()), O._d(hm.c), O._d(hm.mean), O._d(hm.inv_var),
               O._d(hm.det), O._d(b), None)
err = np.abs(got - b) / np.maximum(np.abs(b), 1e-300)
print("parity 48 frames x 32000 Gaussians: max rel err", err.max())
corpus = ctx.corpus(X, [F])
ctx.set_option(G.OPT_TIMING, 1)
ctx.emission(model, corpus, False)
ctx.sync()
ctx.kernel_times_reset()
t = time.perf_counter()
ctx.emission(model, corpus, False)
ctx.sync()
dt = time.perf_counter() - t
flop = 2.0 * F * 80 * N * M
print(f"{F} frames: {dt*1e3:.1f} ms, {F/dt/1e6:.2f} Mframes/s, {flop/dt/1e12:.1f} TFLOP/s f64 (MFMA form)",
      ctx.kernel_times()["emission"])

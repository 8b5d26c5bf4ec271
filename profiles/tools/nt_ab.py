"""Posterior stores of the emission kernel, non-temporal against plain (GHMM_OPT_NT_POST 1 / 2), on the
benchmark corpus (10x8, 1 000 x 300 frames) and on one GPU's share of configs[3] (64 mixtures,
12 500 x 300): EM iteration wall time and per-kernel HIP-event times."""
import sys, time, numpy as np
sys.path.insert(0, "tests")
from _load import load_pkg
pkg = load_pkg(); G, em = pkg.ghmm, pkg.em
import torch
ctx = G.Context(0)
for (M, U, steps) in ((8, 1000, 200), (64, 12500, 4)):
    N, D, T = 10, 39, 300
    mean, std = G.synth_truth(N, M, D)
    lens = np.full(U, T, dtype=np.int32)
    X = torch.from_numpy(G.synth_utterances(mean, std, lens, threads=8)).to("cuda:0")
    corpus = ctx.corpus_from_device(X.data_ptr(), lens, D)
    for rep in range(2):
        for mode, name in ((2, "plain"), (1, "nt")):
            ctx.set_option(G.OPT_NT_POST, mode)
            model = ctx.model(G.synth_start_model(mean, std, 0.05))
            be = em.HipBackend(G, ctx, model, corpus); drv = em.EMDriver(be)
            for _ in range(max(2, steps // 2)): drv.step()
            ctx.sync(); t0 = time.perf_counter()
            for _ in range(steps): drv.step()
            ctx.sync(); dt = (time.perf_counter() - t0) / steps
            ctx.set_option(G.OPT_TIMING, 1); ctx.kernel_times_reset()
            for _ in range(steps): drv.step()
            kt = {k: round(v[0] / steps, 4) for k, v in ctx.kernel_times().items() if v[1]}
            ctx.set_option(G.OPT_TIMING, 0)
            print(f"{M} mixtures, {name:5s}: {dt * 1e3:.4f} ms per iteration  {kt}")
            be.stats.close(); model.close()
    corpus.close(); del X
ctx.set_option(G.OPT_NT_POST, 0)

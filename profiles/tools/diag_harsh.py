"""Details of harsh-fuzz disagreements (profiles/fuzz_oracle.py N harsh): where the worst entry of
alpha^ / beta^ / the statistics sits and what the densities look like there."""
import sys, numpy as np
sys.path.insert(0, "tests")
from _load import load_pkg
import test_gpu_parity as T
G = load_pkg().ghmm
O = T.O
ctx = G.Context(0)
wide = "wide" in sys.argv
for seed in [int(a) for a in sys.argv[1:] if a.isdigit()]:
    rng = np.random.default_rng((19000 if wide else 9000) + seed)
    N, M, D = T.fuzz_shape(rng, wide)
    lens = [int(x) for x in rng.integers(N, N + 120, size=int(rng.integers(1, 7)))]
    dense, delta = bool(rng.integers(0, 2)), int(rng.integers(0, 4))
    hm, X, lens = T.synth_case(G, N, M, D, lens, dense_A=dense, seed=seed, perturb=float(rng.choice([0.6, 1.0])))
    k = float(rng.choice([1.0, 3.0, 9.0])); hm.inv_var *= k; hm.det /= k ** D
    ref_stats, ref = O.estep(hm, X, lens, delta=delta)
    model, corpus = ctx.model(hm), ctx.corpus(X, lens)
    F = corpus.frames
    ctx.set_option(G.OPT_DELTA, delta)
    stats = ctx.stats(N, M, D)
    ctx.estep(model, corpus, stats)
    print(f"seed {seed}: N={N} M={M} D={D} lens={list(map(int, lens))} dense={dense} delta={delta} sharpen={k}")
    for name, buf in (("b", G.BUF_B), ("alpha", G.BUF_ALPHA), ("beta", G.BUF_BETA)):
        got = ctx.fetch(buf, (F, N)); r = ref[name]
        scale = np.abs(r).max(axis=1, keepdims=True); scale[scale == 0] = 1
        err = np.abs(got - r) / scale
        err[~np.isfinite(err)] = np.inf if np.any(np.isnan(got) != np.isnan(r)) else 0
        t, i = np.unravel_index(np.nanargmax(err), err.shape)
        print(f"  {name}: worst per-frame relative error {err[t, i]:.3e} at frame {t} state {i}: got {got[t, i]!r} ref {r[t, i]!r}; frame's b: max {ref['b'][t].max():.3e} min {ref['b'][t].min():.3e}")
    got = stats.download(); sp = G.split_stats(got, N, M, D); rp = G.split_stats(ref_stats, N, M, D)
    for nm in ("num_a", "den_a", "den_c", "num_c", "num_mu", "num_var"):
        a, b = sp[nm], rp[nm]
        with np.errstate(all="ignore"):
            rel = np.abs(a - b) / np.maximum(np.abs(b), 1e-300)
        rel[~np.isfinite(rel)] = 0
        j = np.unravel_index(np.argmax(rel), rel.shape)
        print(f"  {nm}: worst relative error {rel[j]:.3e} at {j}: got {a[j]!r} ref {b[j]!r} (array max {np.abs(b).max():.3e})")
    ctx.set_option(G.OPT_DELTA, 1)
    for o in (model, corpus, stats): o.close()
    # (appended) M-step comparison of the same case
    model, corpus = ctx.model(hm), ctx.corpus(X, lens)
    ctx.set_option(G.OPT_DELTA, delta)
    stats = ctx.stats(N, M, D)
    ctx.estep(model, corpus, stats); got = stats.download()
    ctx.mstep(model, stats)
    new, ref_new = model.get(), O.mstep(hm, ref_stats)
    sp = G.split_stats(got, N, M, D); rp = G.split_stats(ref_stats, N, M, D)
    for nm, a, b in zip(("A", "c", "mean", "inv_var", "det"), new.arrays(), ref_new.arrays()):
        with np.errstate(all="ignore"):
            rel = np.abs(a - b) / np.maximum(np.abs(b), 1e-300)
        rel[~np.isfinite(rel)] = 0
        j = np.unravel_index(np.argmax(rel), rel.shape)
        msg = f"  mstep.{nm}: worst relative error {rel[j]:.3e} at {j}: got {a[j]!r} ref {b[j]!r}"
        if nm in ("mean", "inv_var"):
            i, k, d = (int(v) for v in j)
            msg += f" | num_c got {sp['num_c'][i, k]!r} ref {rp['num_c'][i, k]!r}; num_mu {sp['num_mu'][i, k, d]!r} / {rp['num_mu'][i, k, d]!r}; num_var {sp['num_var'][i, k, d]!r} / {rp['num_var'][i, k, d]!r}; old mean {hm.mean[i, k, d]!r}"
        print(msg)
    ctx.set_option(G.OPT_DELTA, 1)

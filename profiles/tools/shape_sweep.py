# Iteration time and kernel times over model shapes inside the reference's range, to spot outliers:
#   per Gaussian-frame cost of emission and statistics, per state-frame cost of the recursions
import sys, time, numpy as np
sys.path.insert(0, "tests")
from _load import load_pkg
pkg = load_pkg(); G, em = pkg.ghmm, pkg.em
ctx = G.Context(0)
U, T = 1000, 300
shapes = [(10, 8, 39), (6, 8, 39), (5, 4, 39), (10, 1, 39), (10, 2, 39), (10, 3, 39), (10, 4, 39), (10, 5, 39), (10, 6, 39),
          (10, 16, 39), (10, 32, 39), (20, 4, 39), (20, 1, 39), (10, 8, 13), (10, 8, 26), (10, 8, 12), (8, 8, 40), (10, 8, 45)]
corp = [(U, T)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:] if not a.startswith("U")]
    corp = [tuple(int(x) for x in a[1:].split("x")) for a in sys.argv[1:] if a.startswith("U")] or corp
for (U, T) in corp:
  for (N, M, D) in shapes:
      mean, std = G.synth_truth(N, M, D)
      lens = np.full(U, T, dtype=np.int32)
      X = G.synth_utterances(mean, std, lens)
      corpus = ctx.corpus(X, lens)
      model = ctx.model(G.synth_start_model(mean, std, 0.05))
      be = em.HipBackend(G, ctx, model, corpus); drv = em.EMDriver(be)
      for _ in range(30): drv.step()
      ctx.sync(); t0 = time.perf_counter()
      for _ in range(50): drv.step()
      ctx.sync(); wall = 1e3 * (time.perf_counter() - t0) / 50
      ctx.set_option(G.OPT_TIMING, 1); ctx.kernel_times_reset()
      for _ in range(10): drv.step()
      kt = {k: 1e3 * ms / 10 for k, (ms, n) in ctx.kernel_times().items() if n}
      ctx.set_option(G.OPT_TIMING, 0)
      gf = U * T * N * M
      print(f"U={U} T={T} {N:3d}x{M:2d} D={D:2d}: {wall:.4f} ms | emission {kt.get('emission',0):6.1f} us ({1e3*kt.get('emission',0)/gf*1e3:.3f} ps/Gf) "
            f"recursions {kt.get('forward',0)+kt.get('backward',0):6.1f} statistics {kt.get('mixstats',0):6.1f} ({1e3*kt.get('mixstats',0)/gf*1e3:.3f} ps/Gf) "
            f"reduce {kt.get('reduce',0):5.1f} mstep {kt.get('mstep',0):5.1f}")
      be.stats.close(); model.close(); corpus.close()

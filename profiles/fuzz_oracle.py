"""Fuzz: the product's E-step + M-step (default tier) against the oracle (oracle/ghmm_oracle.c, the
CPU restatement pinned to the reference) on seeded random shapes — the body is
tests/test_gpu_parity.py:fuzz_estep_case (20 seeds of it run in the -m gpu suite).
usage: fuzz_oracle.py [n_seeds] [wide] [harsh] [short] [subnormal]   (wide: shapes up to 64 states x 64 mixtures x 64
coefficients; harsh: models far from their data with sharpened Gaussians — cases in which the
reference itself leaves the finite numbers are skipped and counted)"""
import sys
sys.path.insert(0, "tests")
from _load import load_pkg
import test_gpu_parity as T

G = load_pkg().ghmm
ctx = G.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
wide = "huge" if "huge" in sys.argv[2:] else "wide" in sys.argv[2:]   # huge: 65 .. 255 states (ghmm_wide.hpp)
harsh = "harsh" in sys.argv[2:]
short = "short" in sys.argv[2:]
subnormal = "subnormal" in sys.argv[2:]   # statistics below 1e-300 compared absolutely, their quotients not at all
bad = skipped = reordered = 0
which = []
for seed in range(n):
    try:
        if T.fuzz_estep_case(G, ctx, seed, wide=wide, harsh=harsh, short=short, subnormal=subnormal):
            reordered += 1
            which.append(seed)
    except T.FuzzSkip:
        skipped += 1
    except AssertionError as e:
        bad += 1
        print(str(e)[:240])
print(f"{n} {'huge ' if wide == 'huge' else 'wide ' if wide else ''}{'harsh ' if harsh else ''}{'short ' if short else ''}shapes against the oracle, {bad} disagreements" + (f", {skipped} skipped (reference not finite)" if harsh or short else "")
      + f"; {reordered} agreeing shapes had utterances taken again in the reference's order")

if which and (harsh or "list" in sys.argv[2:]):
    print("seeds with utterances taken again:", which)

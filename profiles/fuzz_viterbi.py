"""Fuzz: Viterbi state sequences and forward scores of the product against the oracle's
(oracle/ghmm_oracle.c) on seeded random shapes — the body is
tests/test_gpu_parity.py:fuzz_viterbi_case (20 seeds of it run in the -m gpu suite).
usage: fuzz_viterbi.py [n] [wide] [harsh]"""
import sys
sys.path.insert(0, "tests")
from _load import load_pkg
import test_gpu_parity as T

G = load_pkg().ghmm
ctx = G.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
wide = "huge" if "huge" in sys.argv[2:] else "wide" in sys.argv[2:]   # huge: 65 .. 255 states (ghmm_wide.hpp)
harsh = "harsh" in sys.argv[2:]
bad = utts = 0
for seed in range(n):
    try:
        utts += T.fuzz_viterbi_case(G, ctx, seed, wide=wide, harsh=harsh)
    except AssertionError as e:
        bad += 1
        print(str(e)[:240])
print(f"{n} {'huge ' if wide == 'huge' else 'wide ' if wide else ''}shapes, {utts} utterances, {bad} differ")

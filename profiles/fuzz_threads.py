"""Stress: two host threads, each with its own context (own stream, same GPU), run the oracle fuzz
body over different seeds at the same time (no global mutable state).
usage: fuzz_threads.py [seeds per thread]"""
import sys, threading
sys.path.insert(0, "tests")
from _load import load_pkg
import test_gpu_parity as T

G = load_pkg().ghmm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = []


def work(k):
    ctx = G.Context(0)
    for seed in range(k * n, (k + 1) * n):
        try:
            T.fuzz_estep_case(G, ctx, seed, wide=bool(seed & 1))
            T.fuzz_viterbi_case(G, ctx, seed)
        except AssertionError as e:
            bad.append(str(e)[:200])
    ctx.close()


ts = [threading.Thread(target=work, args=(k,)) for k in range(3)]
for t in ts: t.start()
for t in ts: t.join()
for b in bad: print(b)
print(f"3 threads x {n} seeds (E-step + M-step and Viterbi bodies) on three contexts at once: {len(bad)} disagreements")

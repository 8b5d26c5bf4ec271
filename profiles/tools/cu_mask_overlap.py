"""Feasibility of hiding the forward / backward scans behind the emission kernel: two contexts on
two CU-masked streams (hipExtStreamCreateWithCUMask) of one GPU — the emission kernel on most of
the device, the scans on a few reserved compute units — alone and together from two host threads.
usage: cu_mask_overlap.py [reserved CUs = 16] [layout = stride|low]"""
import ctypes, sys, threading, time
import numpy as np
sys.path.insert(0, "tests")
from _load import load_pkg
G = load_pkg().ghmm
hip = ctypes.CDLL("libamdhip64.so")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 16
layout = sys.argv[2] if len(sys.argv) > 2 else "stride"
CUS = 256


def masked_stream(bits):
    words = (ctypes.c_uint32 * (CUS // 32))()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), CUS // 32, words)
    assert rc == 0, f"hipExtStreamCreateWithCUMask -> {rc}"
    return st


if layout == "stride":            # every (256/R)-th bit
    side_bits = [i * (CUS // R) for i in range(R)]
elif layout == "low":             # the first R bits
    side_bits = list(range(R))
else:                             # "high": the last R bits
    side_bits = list(range(CUS - R, CUS))
main_bits = [b for b in range(CUS) if b not in set(side_bits)]
s_main, s_side = masked_stream(main_bits), masked_stream(side_bits)

N, M, D, U, T = 10, 8, 39, 1000, 300
mean, std = G.synth_truth(N, M, D)
lens = np.full(U, T, dtype=np.int32)
X = G.synth_utterances(mean, std, lens)
hm = G.synth_start_model(mean, std, 0.05)
A = G.Context(0, stream=s_main.value)
B = G.Context(0, stream=s_side.value)
A.set_option(G.OPT_CUS, CUS - R)
mA, cA = A.model(hm), A.corpus(X, lens)
mB, cB = B.model(hm), B.corpus(X, lens)
B.emission(mB, cB, False)
B.forward(mB, cB)
B.sync()
REP = 200


def run_a(out):
    A.sync()
    t = time.perf_counter()
    for _ in range(REP):
        A.emission(mA, cA, True)
    A.sync()
    out["emission"] = (time.perf_counter() - t) / REP * 1e6


def run_b(out):
    B.sync()
    t = time.perf_counter()
    for _ in range(REP):
        B.forward(mB, cB)
    B.sync()
    out["scan"] = (time.perf_counter() - t) / REP * 1e6


for rep in range(2):
    alone = {}
    run_a(alone)
    run_b(alone)
    both = {}
    ta, tb = threading.Thread(target=run_a, args=(both,)), threading.Thread(target=run_b, args=(both,))
    ta.start(); tb.start(); ta.join(); tb.join()
    print(f"reserved {R} CUs ({layout}): alone emission {alone['emission']:.1f} us, scan {alone['scan']:.1f} us | "
          f"together emission {both['emission']:.1f} us, scan {both['scan']:.1f} us")

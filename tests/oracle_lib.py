"""ctypes face of oracle/libghmm_oracle.so (TEST INFRASTRUCTURE: the CPU
restatement of the reference path).  Imported by tests/, smoke() and bench.py's
cpu_baseline leg only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "libghmm_oracle.so")
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"),
                                   os.path.join(ROOT, "oracle", "libghmm_oracle.so")])
        L = C.CDLL(LIB)
        L.orc_stats_len.restype = C.c_size_t
        L.orc_stats_len.argtypes = [C.c_int] * 3
        L.orc_estep.restype = C.c_int
        L.orc_estep.argtypes = [C.c_int] * 4 + [_dp] * 6 + [_ip, C.c_int] + [_dp] * 7
        L.orc_estep_mt.restype = C.c_int
        L.orc_estep_mt.argtypes = [C.c_int] * 5 + [_dp] * 6 + [_ip, C.c_int, _dp]
        L.orc_mstep.restype = None
        L.orc_mstep.argtypes = [C.c_int] * 3 + [_dp] * 6
        L.orc_train.restype = C.c_int
        L.orc_train.argtypes = ([C.c_int] * 4 + [C.c_double, C.c_int, C.c_int] + [_dp] * 6 +
                                [_ip, C.c_int, _dp, _dp])
        L.orc_score.restype = C.c_double
        L.orc_score.argtypes = [C.c_int] * 4 + [_dp] * 6
        L.orc_viterbi.restype = C.c_double
        L.orc_viterbi.argtypes = [C.c_int] * 4 + [_dp] * 6 + [_ip]
        L.orc_viterbi_lattice.restype = C.c_double
        L.orc_viterbi_lattice.argtypes = [C.c_int, C.c_int, _dp, _dp, _ip]
        L.orc_log_emission.restype = None
        L.orc_log_emission.argtypes = [C.c_int] * 4 + [_dp] * 6
        L.orc_emission.restype = None
        L.orc_emission.argtypes = [C.c_int] * 4 + [_dp] * 7
        _pp = C.POINTER(_dp)
        L.orc_estep_streams.restype = C.c_int
        L.orc_estep_streams.argtypes = ([C.c_int, C.c_int, _ip, _ip, C.c_int, _dp] + [_pp] * 5 +
                                        [_ip, C.c_int, _pp, _dp, _dp])
        L.orc_train_streams.restype = C.c_int
        L.orc_train_streams.argtypes = ([C.c_int, C.c_int, _ip, _ip, C.c_int, C.c_double, C.c_int, C.c_int,
                                         _dp] + [_pp] * 5 + [_ip, C.c_int, _dp, _dp])
        L.orc_score_streams.restype = C.c_double
        L.orc_score_streams.argtypes = [C.c_int, C.c_int, _ip, _ip, C.c_int, _dp] + [_pp] * 5
        L.orc_sort_scores.restype = None
        L.orc_sort_scores.argtypes = [C.c_int, _dp, _ip]
        _lib = L
    return _lib


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _marr(hm):
    return [_d(x) for x in (hm.A, hm.c, hm.mean, hm.inv_var, hm.det)]


def stats_len(N, M, D):
    return lib().orc_stats_len(N, M, D)


def estep(hm, X, lens, delta=1, dumps=True):
    """One E-step (TF:244-321).  Returns (stats, dict of per-frame arrays)."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    N, M, D = hm.N, hm.M, hm.D
    F, U = int(lens.sum()), len(lens)
    stats = np.zeros(stats_len(N, M, D))
    out = {}
    if dumps:
        out = dict(b=np.zeros((F, N)), post=np.zeros((F, N, M)), alpha=np.zeros((F, N)),
                   beta=np.zeros((F, N)), scale=np.zeros(F), loglik=np.zeros(U))
    rc = lib().orc_estep(N, M, D, delta, *_marr(hm), _d(X), lens.ctypes.data_as(_ip), U,
                         _d(stats), _d(out.get("b")), _d(out.get("post")), _d(out.get("alpha")),
                         _d(out.get("beta")), _d(out.get("scale")), _d(out.get("loglik")))
    assert rc == 0
    return stats, out


def estep_mt(hm, X, lens, threads, delta=1):
    """orc_estep over `threads` host threads (utterance blocks); returns the statistics vector."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    stats = np.zeros(stats_len(hm.N, hm.M, hm.D))
    rc = lib().orc_estep_mt(threads, hm.N, hm.M, hm.D, delta, *_marr(hm), _d(X), lens.ctypes.data_as(_ip),
                            len(lens), _d(stats))
    assert rc == 0
    return stats


def mstep(hm, stats):
    """In-place M-step (TF:332-346) on a copy of hm; returns the new model."""
    new = hm.copy()
    stats = np.ascontiguousarray(stats, dtype=np.float64)
    lib().orc_mstep(hm.N, hm.M, hm.D, _d(stats), *_marr(new))
    return new


def train(hm, X, lens, delta=1, threshold=1e-3, max_iter=0, fixed_iter=False):
    """EM driver (TF:238-358).  Returns (model, iterations, mean loglik, trace)."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    new = hm.copy()
    mean_ll = C.c_double()
    trace = np.zeros(max(max_iter, 1) if max_iter > 0 else 1000)
    it = lib().orc_train(hm.N, hm.M, hm.D, delta, threshold, max_iter, int(fixed_iter),
                         *_marr(new), _d(X), lens.ctypes.data_as(_ip), len(lens),
                         C.byref(mean_ll), _d(trace))
    assert it > 0
    return new, it, mean_ll.value, trace[:it].copy()


def score(hm, X):
    X = np.ascontiguousarray(X, dtype=np.float64)
    return lib().orc_score(hm.N, hm.M, hm.D, X.shape[0], *_marr(hm), _d(X))


def viterbi(hm, X):
    X = np.ascontiguousarray(X, dtype=np.float64)
    path = np.zeros(X.shape[0], dtype=np.int32)
    s = lib().orc_viterbi(hm.N, hm.M, hm.D, X.shape[0], *_marr(hm), _d(X),
                          path.ctypes.data_as(_ip))
    return path, s


def log_emission(hm, X):
    X = np.ascontiguousarray(X, dtype=np.float64)
    out = np.zeros((X.shape[0], hm.N))
    lib().orc_log_emission(hm.N, hm.M, hm.D, X.shape[0], _d(X), _d(hm.c), _d(hm.mean),
                           _d(hm.inv_var), _d(hm.det), _d(out))
    return out


def emission(hm, X, want_post=False):
    """calc_symbol_probab over the frames of X (TF:1749-1841): b[T][N] (and post[T][N][M])."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    b = np.zeros((X.shape[0], hm.N))
    post = np.zeros((X.shape[0], hm.N, hm.M)) if want_post else None
    lib().orc_emission(hm.N, hm.M, hm.D, X.shape[0], _d(X), _d(hm.c), _d(hm.mean), _d(hm.inv_var),
                       _d(hm.det), _d(b), _d(post))
    return (b, post) if want_post else b


def viterbi_lattice(A, logb):
    A = np.ascontiguousarray(A, dtype=np.float64)
    logb = np.ascontiguousarray(logb, dtype=np.float64)
    T, N = logb.shape
    path = np.zeros(T, dtype=np.int32)
    s = lib().orc_viterbi_lattice(N, T, _d(A), _d(logb), path.ctypes.data_as(_ip))
    return path, s


def sort_scores(scores):
    scores = np.ascontiguousarray(scores, dtype=np.float64)
    idx = np.zeros(len(scores), dtype=np.int32)
    lib().orc_sort_scores(len(scores), _d(scores), idx.ctypes.data_as(_ip))
    return idx


# ---------------------------------------------------------- several feature streams

def _ptrs(arrays):
    """C array of double* over numpy arrays (kept alive by the caller)."""
    return (_dp * len(arrays))(*[_d(a) for a in arrays])


def _stream_args(hms, Xs):
    P = len(hms)
    M = np.array([h.M for h in hms], dtype=np.int32)
    D = np.array([h.D for h in hms], dtype=np.int32)
    Xs = [np.ascontiguousarray(x, dtype=np.float64) for x in Xs]
    return P, M, D, Xs


def estep_streams(hms, Xs, lens, delta=1):
    """E-step of a P-stream model (hms[p] = stream p as a HostModel, A from hms[0]), TF:272-321
    with param_number = P.  Returns ([stats_p], product b[F][N], loglik[U])."""
    P, M, D, Xs = _stream_args(hms, Xs)
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    N, F, U = hms[0].N, int(lens.sum()), len(lens)
    stats = [np.zeros(stats_len(N, h.M, h.D)) for h in hms]
    b, ll = np.zeros((F, N)), np.zeros(U)
    rc = lib().orc_estep_streams(P, N, M.ctypes.data_as(_ip), D.ctypes.data_as(_ip), delta, _d(hms[0].A),
                                 _ptrs([h.c for h in hms]), _ptrs([h.mean for h in hms]),
                                 _ptrs([h.inv_var for h in hms]), _ptrs([h.det for h in hms]),
                                 _ptrs(Xs), lens.ctypes.data_as(_ip), U, _ptrs(stats), _d(b), _d(ll))
    assert rc == 0
    return stats, b, ll


def train_streams(hms, Xs, lens, delta=1, threshold=1e-3, max_iter=0, fixed_iter=False):
    """EM driver for P streams.  Returns ([new HostModel per stream], iterations, mean loglik)."""
    P, M, D, Xs = _stream_args(hms, Xs)
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    new = [h.copy() for h in hms]
    A = new[0].A
    mean_ll = C.c_double()
    it = lib().orc_train_streams(P, hms[0].N, M.ctypes.data_as(_ip), D.ctypes.data_as(_ip), delta,
                                 threshold, max_iter, int(fixed_iter), _d(A),
                                 _ptrs([h.c for h in new]), _ptrs([h.mean for h in new]),
                                 _ptrs([h.inv_var for h in new]), _ptrs([h.det for h in new]),
                                 _ptrs(Xs), lens.ctypes.data_as(_ip), len(lens), C.byref(mean_ll), None)
    assert it > 0
    for h in new[1:]:
        h.A[:] = A
    return new, it, mean_ll.value


def score_streams(hms, Xs):
    P, M, D, Xs = _stream_args(hms, Xs)
    return lib().orc_score_streams(P, hms[0].N, M.ctypes.data_as(_ip), D.ctypes.data_as(_ip),
                                   Xs[0].shape[0], _d(hms[0].A), _ptrs([h.c for h in hms]),
                                   _ptrs([h.mean for h in hms]), _ptrs([h.inv_var for h in hms]),
                                   _ptrs([h.det for h in hms]), _ptrs(Xs))

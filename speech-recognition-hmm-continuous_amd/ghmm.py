"""ctypes binding of include/ghmm.h — the thin Python face of the C ABI.

Used by bench.py, __graft_entry__.py and tests/.  Nothing is computed here: every
call goes straight into libghmm_hip.so (HIP kernels) or, for the file-format /
synthetic-data helpers only, libghmm_host.so.  There is no CPU fallback for the
hot path: opening a context without the HIP library or without a gfx950 device
raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GHMM_HIP_LIB: a measurement build of the library (profiles/tools/lab.sh); default = the product
HIP_LIB = os.environ.get("GHMM_HIP_LIB") or os.path.join(_HERE, "libghmm_hip.so")
HOST_LIB = os.path.join(_HERE, "libghmm_host.so")

OK = 0
ERR_ARG, ERR_ALLOC, ERR_HIP, ERR_NODEVICE, ERR_UNSUPPORTED, ERR_IO, ERR_FORMAT = range(1, 8)
(OPT_DELTA, OPT_ROBUST, OPT_KERNELS, OPT_TIMING, OPT_PARTIALS, OPT_CUS, OPT_REFORDER_COUNT, OPT_VEC_STATS,
 OPT_NT_POST, OPT_FUSED_SCAN) = range(1, 11)
(K_EMISSION, K_FORWARD, K_BACKWARD, K_MIXSTATS, K_REDUCE, K_MSTEP, K_VITERBI, K_PREPARE,
 K_COUNT) = range(9)
(BUF_B, BUF_POST, BUF_ALPHA, BUF_BETA, BUF_SCALE, BUF_GAMMA, BUF_LOGLIK, BUF_LOGNORM) = range(8)
SYNTH_SEED = 20260104
MAX_WORD = 256

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_vp = C.c_void_p


class GhmmError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        super().__init__(f"ghmm error {code}: {detail}")


class HostModelStruct(C.Structure):
    _fields_ = [("word", C.c_char * MAX_WORD), ("N", C.c_int), ("M", C.c_int), ("D", C.c_int),
                ("A", _dp), ("c", _dp), ("mean", _dp), ("inv_var", _dp), ("det", _dp)]


# every exported symbol of include/ghmm.h: name -> (restype, argtypes, needs_hip)
SYMBOLS = {
    "ghmm_strerror": (C.c_char_p, [C.c_int], False),
    "ghmm_last_error": (C.c_char_p, [], False),
    "ghmm_version": (C.c_int, [], False),
    "ghmm_ctx_create": (C.c_int, [C.c_int, _vp, C.POINTER(_vp)], True),
    "ghmm_ctx_destroy": (None, [_vp], True),
    "ghmm_ctx_sync": (C.c_int, [_vp], True),
    "ghmm_ctx_set_option": (C.c_int, [_vp, C.c_int, C.c_int64], True),
    "ghmm_ctx_get_option": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int64)], True),
    "ghmm_ctx_kernel_time": (C.c_int, [_vp, C.c_int, _dp, C.POINTER(C.c_int64)], True),
    "ghmm_ctx_kernel_time_reset": (C.c_int, [_vp], True),
    "ghmm_kernel_name": (C.c_char_p, [C.c_int], True),
    "ghmm_model_create": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)], True),
    "ghmm_model_destroy": (None, [_vp, _vp], True),
    "ghmm_model_set": (C.c_int, [_vp, _vp, _dp, _dp, _dp, _dp, _dp], True),
    "ghmm_model_get": (C.c_int, [_vp, _vp, _dp, _dp, _dp, _dp, _dp], True),
    "ghmm_model_init": (C.c_int, [_vp, _vp, _vp], True),
    "ghmm_model_init_comm": (C.c_int, [_vp, _vp, _vp, _vp], True),
    "ghmm_comm_unique_id": (C.c_int, [_vp], True),
    "ghmm_comm_create": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.POINTER(_vp)], True),
    "ghmm_comm_create_file": (C.c_int, [_vp, C.c_char_p, C.c_int, C.c_int, C.c_double,
                                        C.POINTER(_vp)], True),
    "ghmm_comm_destroy": (None, [_vp], True),
    "ghmm_comm_rank": (C.c_int, [_vp], True),
    "ghmm_comm_size": (C.c_int, [_vp], True),
    "ghmm_stats_allreduce": (C.c_int, [_vp, _vp, _vp], True),
    "ghmm_model_dims": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                  C.POINTER(C.c_int)], True),
    "ghmm_corpus_create": (C.c_int, [_vp, _dp, _ip, C.c_int, C.c_int, C.POINTER(_vp)], True),
    "ghmm_corpus_wrap": (C.c_int, [_vp, _vp, _ip, C.c_int, C.c_int, C.POINTER(_vp)], True),
    "ghmm_corpus_destroy": (None, [_vp, _vp], True),
    "ghmm_corpus_frames": (C.c_int64, [_vp], True),
    "ghmm_corpus_utterances": (C.c_int, [_vp], True),
    "ghmm_stats_len": (C.c_size_t, [C.c_int, C.c_int, C.c_int], True),
    "ghmm_stats_create": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)], True),
    "ghmm_stats_wrap": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, C.POINTER(_vp)], True),
    "ghmm_stats_destroy": (None, [_vp, _vp], True),
    "ghmm_stats_device_ptr": (_vp, [_vp], True),
    "ghmm_stats_download": (C.c_int, [_vp, _vp, _dp], True),
    "ghmm_stats_upload": (C.c_int, [_vp, _vp, _dp], True),
    "ghmm_stats_loglik": (C.c_int, [_vp, _vp, _dp], True),
    "ghmm_emission": (C.c_int, [_vp, _vp, _vp, C.c_int], True),
    "ghmm_forward": (C.c_int, [_vp, _vp, _vp], True),
    "ghmm_backward": (C.c_int, [_vp, _vp, _vp], True),
    "ghmm_accumulate": (C.c_int, [_vp, _vp, _vp, _vp], True),
    "ghmm_fetch": (C.c_int, [_vp, C.c_int, _dp, C.c_size_t], True),
    "ghmm_fetch_range": (C.c_int, [_vp, C.c_int, C.c_size_t, _dp, C.c_size_t], True),
    "ghmm_estep": (C.c_int, [_vp, _vp, _vp, _vp], True),
    "ghmm_mstep": (C.c_int, [_vp, _vp, _vp], True),
    "ghmm_score": (C.c_int, [_vp, _vp, _vp, _dp], True),
    "ghmm_score_batch": (C.c_int, [_vp, C.POINTER(_vp), C.c_int, _vp, _dp], True),
    "ghmm_estep_streams": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.c_int, C.POINTER(_vp)], True),
    "ghmm_score_streams": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.c_int, _dp], True),
    "ghmm_viterbi": (C.c_int, [_vp, _vp, _vp, _ip, _dp], True),
    "ghmm_perfil_read": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                   C.POINTER(_dp)], False),
    "ghmm_perfil_write": (C.c_int, [C.c_char_p, C.c_int, C.c_int, _dp], False),
    "ghmm_perfil_stat": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)], False),
    "ghmm_shard_balanced": (C.c_int, [_ip, C.c_int, C.c_int, C.c_int, _ip, C.POINTER(C.c_int)],
                            False),
    "ghmm_rendezvous_file": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_double, _vp], False),
    "ghmm_free": (None, [_vp], False),
    "ghmm_host_model_alloc": (C.c_int, [C.POINTER(HostModelStruct), C.c_int, C.c_int, C.c_int],
                              False),
    "ghmm_host_model_free": (None, [C.POINTER(HostModelStruct)], False),
    "ghmm_hmm_read": (C.c_int, [C.c_char_p, C.POINTER(HostModelStruct)], False),
    "ghmm_hmm_write": (C.c_int, [C.c_char_p, C.POINTER(HostModelStruct), C.c_int], False),
    "ghmm_hmm_read_streams": (C.c_int, [C.c_char_p, C.POINTER(HostModelStruct), C.c_int,
                                        C.POINTER(C.c_int)], False),
    "ghmm_hmm_write_streams": (C.c_int, [C.c_char_p, C.POINTER(HostModelStruct), C.c_int, C.c_int],
                               False),
    "ghmm_init_model": (C.c_int, [_dp, _ip, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.POINTER(HostModelStruct)], False),
    "ghmm_synth_truth": (C.c_int, [C.c_uint64, C.c_int, C.c_int, C.c_int, _dp, _dp], False),
    "ghmm_synth_utterances": (C.c_int, [C.c_uint64, C.c_int, C.c_int, C.c_int, _dp, _dp,
                                        C.c_int64, C.c_int, _ip, _dp], False),
    "ghmm_synth_start_model": (C.c_int, [C.c_uint64, C.c_int, C.c_int, C.c_int, _dp, _dp,
                                         C.c_double, _dp, _dp, _dp, _dp, _dp], False),
}

_libs = {}


def _bind(lib, hip):
    for name, (res, args, needs_hip) in SYMBOLS.items():
        if needs_hip and not hip:
            continue
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


def hip_lib():
    """The C-ABI library with the HIP kernels.  Raises if it has not been built."""
    if "hip" not in _libs:
        if not os.path.exists(HIP_LIB):
            raise GhmmError(ERR_NODEVICE, f"{HIP_LIB} is missing — run __graft_entry__.build(); "
                                          "the hot path has no CPU fallback")
        _libs["hip"] = _bind(C.CDLL(HIP_LIB), True)
    return _libs["hip"]


def host_lib():
    """Host-only helpers (file formats, init model, synthetic corpora)."""
    if "host" not in _libs:
        if os.path.exists(HIP_LIB):
            _libs["host"] = hip_lib()
        else:
            _libs["host"] = _bind(C.CDLL(HOST_LIB), False)
    return _libs["host"]


def _check(rc, lib):
    if rc != OK:
        raise GhmmError(rc, (lib.ghmm_last_error() or b"").decode() or
                        lib.ghmm_strerror(rc).decode())


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


class HostModel:
    """A model in host memory: the flat struct-of-arrays of include/ghmm.h."""

    def __init__(self, A, c, mean, inv_var, det, word=""):
        self.A = _f64(A)
        self.N = self.A.shape[0]
        self.c = _f64(c).reshape(self.N, -1)
        self.M = self.c.shape[1]
        self.mean = _f64(mean).reshape(self.N, self.M, -1)
        self.D = self.mean.shape[2]
        self.inv_var = _f64(inv_var).reshape(self.N, self.M, self.D)
        self.det = _f64(det).reshape(self.N, self.M)
        self.word = word

    def copy(self):
        return HostModel(self.A.copy(), self.c.copy(), self.mean.copy(), self.inv_var.copy(),
                         self.det.copy(), self.word)

    def arrays(self):
        return self.A, self.c, self.mean, self.inv_var, self.det

    def _struct(self):
        s = HostModelStruct()
        s.word = self.word.encode()[:MAX_WORD - 1]
        s.N, s.M, s.D = self.N, self.M, self.D
        s.A, s.c, s.mean, s.inv_var, s.det = (_d(x) for x in self.arrays())
        return s

    @staticmethod
    def _from_struct(s, lib):
        N, M, D = s.N, s.M, s.D
        G = N * M
        hm = HostModel(np.ctypeslib.as_array(s.A, (N, N)).copy(),
                       np.ctypeslib.as_array(s.c, (N, M)).copy(),
                       np.ctypeslib.as_array(s.mean, (N, M, D)).copy(),
                       np.ctypeslib.as_array(s.inv_var, (N, M, D)).copy(),
                       np.ctypeslib.as_array(s.det, (G,)).copy(), s.word.decode())
        lib.ghmm_host_model_free(C.byref(s))
        return hm

    @staticmethod
    def read(path):
        lib = host_lib()
        s = HostModelStruct()
        _check(lib.ghmm_hmm_read(os.fsencode(path), C.byref(s)), lib)
        return HostModel._from_struct(s, lib)

    def write(self, path, len_bytes=8):
        lib = host_lib()
        s = self._struct()
        _check(lib.ghmm_hmm_write(os.fsencode(path), C.byref(s), len_bytes), lib)

    @staticmethod
    def read_streams(path, max_streams=8):
        """A model of several feature streams (param_number P): one HostModel per stream."""
        lib = host_lib()
        arr = (HostModelStruct * max_streams)()
        n = C.c_int()
        _check(lib.ghmm_hmm_read_streams(os.fsencode(path), arr, max_streams, C.byref(n)), lib)
        return [HostModel._from_struct(arr[p], lib) for p in range(n.value)]

    @staticmethod
    def write_streams(path, hms, len_bytes=8):
        lib = host_lib()
        arr = (HostModelStruct * len(hms))(*[h._struct() for h in hms])
        _check(lib.ghmm_hmm_write_streams(os.fsencode(path), arr, len(hms), len_bytes), lib)

    @staticmethod
    def init_from(X, lens, N, M):
        """creating_initial_model (TF:732) on in-memory utterances."""
        lib = host_lib()
        X = _f64(X)
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        s = HostModelStruct()
        _check(lib.ghmm_init_model(_d(X), lens.ctypes.data_as(_ip), len(lens), N, M, X.shape[1],
                                   C.byref(s)), lib)
        return HostModel._from_struct(s, lib)


def perfil_read(path):
    lib = host_lib()
    D, T, p = C.c_int(), C.c_int(), _dp()
    _check(lib.ghmm_perfil_read(os.fsencode(path), C.byref(D), C.byref(T), C.byref(p)), lib)
    X = np.ctypeslib.as_array(p, (T.value, D.value)).copy() if T.value else np.zeros((0, D.value))
    lib.ghmm_free(p)
    return X


def perfil_stat(path):
    """(D, T) of a .perfil file from its header and size."""
    lib = host_lib()
    D, T = C.c_int(), C.c_int()
    _check(lib.ghmm_perfil_stat(os.fsencode(path), C.byref(D), C.byref(T)), lib)
    return D.value, T.value


def shard_balanced(lens, rank, world):
    """ghmm_shard_balanced: this rank's utterance indices (ascending), length-balanced."""
    lib = host_lib()
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    idx = np.empty((len(lens) + world - 1) // world + 1, dtype=np.int32)
    n = C.c_int()
    _check(lib.ghmm_shard_balanced(lens.ctypes.data_as(_ip), len(lens), rank, world,
                                   idx.ctypes.data_as(_ip), C.byref(n)), lib)
    return idx[:n.value].copy()


def perfil_write(path, X):
    lib = host_lib()
    X = _f64(X)
    _check(lib.ghmm_perfil_write(os.fsencode(path), X.shape[1], X.shape[0], _d(X)), lib)


def synth_truth(N, M, D, seed=SYNTH_SEED):
    lib = host_lib()
    mean = np.empty((N, M, D))
    std = np.empty((N, M, D))
    _check(lib.ghmm_synth_truth(seed, N, M, D, _d(mean), _d(std)), lib)
    return mean, std


def synth_utterances(mean, std, lens, first_utt=0, seed=SYNTH_SEED, threads=1):
    """Utterance u depends only on (seed, first_utt + u) (ghmm_synth.c), so `threads` > 1 deals
    contiguous blocks of utterances to host threads (the C call releases the GIL): same bytes."""
    lib = host_lib()
    N, M, D = mean.shape
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    X = np.empty((int(lens.sum()), D))
    U = len(lens)
    threads = max(1, min(int(threads), U))

    def block(lo, hi, f0):
        ln = np.ascontiguousarray(lens[lo:hi])
        _check(lib.ghmm_synth_utterances(seed, N, M, D, _d(mean), _d(std), first_utt + lo, hi - lo,
                                         ln.ctypes.data_as(_ip), _d(X[f0:])), lib)

    if threads == 1:
        block(0, U, 0)
        return X
    import threading
    off = np.concatenate([[0], np.cumsum(lens, dtype=np.int64)])
    cuts = [U * k // threads for k in range(threads + 1)]
    errs = []

    def run(lo, hi):
        try:
            block(lo, hi, int(off[lo]))
        except BaseException as e:   # noqa: BLE001 - re-raised below on the caller's thread
            errs.append(e)

    ts = [threading.Thread(target=run, args=(cuts[k], cuts[k + 1])) for k in range(threads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errs:
        raise errs[0]
    return X


def synth_start_model(mean, std, perturb=0.05, seed=SYNTH_SEED):
    lib = host_lib()
    N, M, D = mean.shape
    A = np.empty((N, N)); c = np.empty((N, M)); mu = np.empty((N, M, D))
    iv = np.empty((N, M, D)); det = np.empty((N, M))
    _check(lib.ghmm_synth_start_model(seed, N, M, D, _d(mean), _d(std), perturb, _d(A), _d(c),
                                      _d(mu), _d(iv), _d(det)), lib)
    return HostModel(A, c, mu, iv, det, "synth")


def rendezvous_file(path, rank, world, timeout_s=120.0, id_bytes=None):
    """ghmm_rendezvous_file: rank 0 passes the 128-byte id, the other ranks get it back."""
    lib = host_lib()
    buf = C.create_string_buffer(id_bytes if id_bytes is not None else b"", 128)
    _check(lib.ghmm_rendezvous_file(os.fsencode(path), rank, world, float(timeout_s), buf), lib)
    return buf.raw


def stats_len(N, M, D):
    return N * N + 2 * N + N * M * (2 * D + 1) + 2


def split_stats(v, N, M, D):
    """Views into a flat statistics vector (layout of include/ghmm.h)."""
    G = N * M
    o = 0
    out = {}
    for name, n, shape in (("num_a", N * N, (N, N)), ("den_a", N, (N,)), ("den_c", N, (N,)),
                           ("num_c", G, (N, M)), ("num_mu", G * D, (N, M, D)),
                           ("num_var", G * D, (N, M, D)), ("loglik", 1, ()), ("n_utt", 1, ())):
        out[name] = v[o:o + n].reshape(shape)
        o += n
    return out


class Context:
    """One GPU, one stream.  `stream` = a hipStream_t as int (e.g. torch's
    torch.cuda.current_stream().cuda_stream) or None for a private stream."""

    def __init__(self, device=0, stream=None):
        self.lib = hip_lib()
        h = _vp()
        _check(self.lib.ghmm_ctx_create(device, _vp(stream) if stream else None, C.byref(h)),
               self.lib)
        self.h = h
        self._children = []

    def close(self):
        if self.h:
            for ch in self._children:
                ch.close()
            self.lib.ghmm_ctx_destroy(self.h)
            self.h = None

    def sync(self):
        _check(self.lib.ghmm_ctx_sync(self.h), self.lib)

    def set_option(self, opt, value):
        _check(self.lib.ghmm_ctx_set_option(self.h, opt, int(value)), self.lib)

    def get_option(self, opt):
        v = C.c_int64()
        _check(self.lib.ghmm_ctx_get_option(self.h, opt, C.byref(v)), self.lib)
        return v.value

    def kernel_times(self):
        out = {}
        for k in range(K_COUNT):
            ms, n = C.c_double(), C.c_int64()
            _check(self.lib.ghmm_ctx_kernel_time(self.h, k, C.byref(ms), C.byref(n)), self.lib)
            out[self.lib.ghmm_kernel_name(k).decode()] = (ms.value, n.value)
        return out

    def kernel_times_reset(self):
        _check(self.lib.ghmm_ctx_kernel_time_reset(self.h), self.lib)

    # ---- objects
    def model(self, hm):
        return Model(self, hm)

    def corpus(self, X, lens):
        return Corpus(self, X, lens)

    def corpus_from_device(self, dev_ptr, lens, D):
        return Corpus(self, None, lens, dev_ptr=dev_ptr, D=D)

    def stats(self, N, M, D, dev_ptr=None):
        return Stats(self, N, M, D, dev_ptr)

    # ---- the path, row by row
    def emission(self, model, corpus, want_post=True):
        _check(self.lib.ghmm_emission(self.h, model.h, corpus.h, int(want_post)), self.lib)

    def forward(self, model, corpus):
        _check(self.lib.ghmm_forward(self.h, model.h, corpus.h), self.lib)

    def backward(self, model, corpus):
        _check(self.lib.ghmm_backward(self.h, model.h, corpus.h), self.lib)

    def accumulate(self, model, corpus, stats):
        _check(self.lib.ghmm_accumulate(self.h, model.h, corpus.h, stats.h), self.lib)

    def fetch(self, which, shape):
        out = np.empty(shape, dtype=np.float64)
        _check(self.lib.ghmm_fetch(self.h, which, _d(out), out.size), self.lib)
        return out

    def fetch_range(self, which, first, shape):
        """doubles [first, first + prod(shape)) of a workspace buffer"""
        out = np.empty(shape, dtype=np.float64)
        _check(self.lib.ghmm_fetch_range(self.h, which, int(first), _d(out), out.size), self.lib)
        return out

    # ---- several GPUs
    def comm(self, rank=0, world=1, id_bytes=None, path=None, timeout_s=120.0):
        return Comm(self, rank, world, id_bytes, path, timeout_s)

    def stats_allreduce(self, stats, comm):
        _check(self.lib.ghmm_stats_allreduce(self.h, stats.h, comm.h), self.lib)

    # ---- fused
    def estep(self, model, corpus, stats):
        _check(self.lib.ghmm_estep(self.h, model.h, corpus.h, stats.h), self.lib)

    def mstep(self, model, stats):
        _check(self.lib.ghmm_mstep(self.h, model.h, stats.h), self.lib)

    def score(self, model, corpus):
        out = np.empty(corpus.n_utt, dtype=np.float64)
        _check(self.lib.ghmm_score(self.h, model.h, corpus.h, _d(out)), self.lib)
        return out

    def score_batch(self, models, corpus):
        """RF:326-374 for a whole vocabulary: out[k, u] = log P(utterance u | model k)."""
        arr = (_vp * len(models))(*[m.h for m in models])
        out = np.empty((len(models), corpus.n_utt), dtype=np.float64)
        _check(self.lib.ghmm_score_batch(self.h, arr, len(models), corpus.h, _d(out)), self.lib)
        return out

    # ---- several feature streams (param_number P > 1)
    def estep_streams(self, models, corpora, stats):
        P = len(models)
        pm = (_vp * P)(*[m.h for m in models])
        pc = (_vp * P)(*[c.h for c in corpora])
        ps = (_vp * P)(*[s.h for s in stats])
        _check(self.lib.ghmm_estep_streams(self.h, pm, pc, P, ps), self.lib)

    def score_streams(self, models, corpora):
        P = len(models)
        pm = (_vp * P)(*[m.h for m in models])
        pc = (_vp * P)(*[c.h for c in corpora])
        out = np.empty(corpora[0].n_utt, dtype=np.float64)
        _check(self.lib.ghmm_score_streams(self.h, pm, pc, P, _d(out)), self.lib)
        return out

    def viterbi(self, model, corpus):
        path = np.empty(corpus.frames, dtype=np.int32)
        score = np.empty(corpus.n_utt, dtype=np.float64)
        _check(self.lib.ghmm_viterbi(self.h, model.h, corpus.h, path.ctypes.data_as(_ip),
                                     _d(score)), self.lib)
        return path, score


class Model:
    def __init__(self, ctx, hm):
        self.ctx, self.N, self.M, self.D = ctx, hm.N, hm.M, hm.D
        h = _vp()
        _check(ctx.lib.ghmm_model_create(ctx.h, hm.N, hm.M, hm.D, C.byref(h)), ctx.lib)
        self.h = h
        ctx._children.append(self)
        self.set(hm)

    def set(self, hm):
        _check(self.ctx.lib.ghmm_model_set(self.ctx.h, self.h, *(_d(x) for x in hm.arrays())),
               self.ctx.lib)

    def init_from(self, corpus, comm=None, fetch=True):
        """creating_initial_model (TF:732) on the device; returns the model as HostModel
        (fetch=False: leaves it on the device, nothing is downloaded).
        `comm`: the corpus is one rank's shard, the k-means sums are all-reduced."""
        _check(self.ctx.lib.ghmm_model_init_comm(self.ctx.h, self.h, corpus.h,
                                                 comm.h if comm is not None else None),
               self.ctx.lib)
        return self.get() if fetch else None

    def get(self):
        N, M, D = self.N, self.M, self.D
        A = np.empty((N, N)); c = np.empty((N, M)); mu = np.empty((N, M, D))
        iv = np.empty((N, M, D)); det = np.empty((N, M))
        _check(self.ctx.lib.ghmm_model_get(self.ctx.h, self.h, _d(A), _d(c), _d(mu), _d(iv),
                                           _d(det)), self.ctx.lib)
        return HostModel(A, c, mu, iv, det)

    def close(self):
        if self.h:
            self.ctx.lib.ghmm_model_destroy(self.ctx.h, self.h)
            self.h = None


class Corpus:
    def __init__(self, ctx, X, lens, dev_ptr=None, D=None):
        self.ctx = ctx
        self.lens = np.ascontiguousarray(lens, dtype=np.int32)
        self.n_utt = len(self.lens)
        self.frames = int(self.lens.sum())
        h = _vp()
        lp = self.lens.ctypes.data_as(_ip)
        if dev_ptr is None:
            X = _f64(X)
            assert X.shape[0] == self.frames
            self.D = X.shape[1]
            _check(ctx.lib.ghmm_corpus_create(ctx.h, _d(X), lp, self.n_utt, self.D, C.byref(h)),
                   ctx.lib)
        else:
            self.D = D
            _check(ctx.lib.ghmm_corpus_wrap(ctx.h, _vp(dev_ptr), lp, self.n_utt, D, C.byref(h)),
                   ctx.lib)
        self.h = h
        ctx._children.append(self)

    def close(self):
        if self.h:
            self.ctx.lib.ghmm_corpus_destroy(self.ctx.h, self.h)
            self.h = None


class Comm:
    """One rank of an RCCL communicator (ghmm_comm_*): the statistics all-reduce in C."""

    def __init__(self, ctx, rank, world, id_bytes=None, path=None, timeout_s=120.0):
        self.ctx = ctx
        h = _vp()
        if path is not None:
            _check(ctx.lib.ghmm_comm_create_file(ctx.h, os.fsencode(path), rank, world, timeout_s,
                                                 C.byref(h)), ctx.lib)
        else:
            if id_bytes is None:
                buf = C.create_string_buffer(128)
                _check(ctx.lib.ghmm_comm_unique_id(buf), ctx.lib)
                id_bytes = buf.raw
            buf = C.create_string_buffer(id_bytes, 128)
            _check(ctx.lib.ghmm_comm_create(ctx.h, buf, rank, world, C.byref(h)), ctx.lib)
        self.h = h
        ctx._children.append(self)

    @property
    def rank(self):
        return self.ctx.lib.ghmm_comm_rank(self.h)

    @property
    def size(self):
        return self.ctx.lib.ghmm_comm_size(self.h)

    def close(self):
        if self.h:
            self.ctx.lib.ghmm_comm_destroy(self.h)
            self.h = None


class Stats:
    def __init__(self, ctx, N, M, D, dev_ptr=None):
        self.ctx, self.N, self.M, self.D = ctx, N, M, D
        self.n = stats_len(N, M, D)
        assert ctx.lib.ghmm_stats_len(N, M, D) == self.n
        h = _vp()
        if dev_ptr is None:
            _check(ctx.lib.ghmm_stats_create(ctx.h, N, M, D, C.byref(h)), ctx.lib)
        else:
            _check(ctx.lib.ghmm_stats_wrap(ctx.h, N, M, D, _vp(dev_ptr), C.byref(h)), ctx.lib)
        self.h = h
        ctx._children.append(self)

    def download(self):
        out = np.empty(self.n, dtype=np.float64)
        _check(self.ctx.lib.ghmm_stats_download(self.ctx.h, self.h, _d(out)), self.ctx.lib)
        return out

    def loglik(self):
        """(sum of log P, utterance count): the 16 bytes the stopping rule reads (TF:318-325)"""
        out = np.empty(2, dtype=np.float64)
        _check(self.ctx.lib.ghmm_stats_loglik(self.ctx.h, self.h, _d(out)), self.ctx.lib)
        return float(out[0]), float(out[1])

    def upload(self, v):
        v = _f64(v)
        assert v.size == self.n
        _check(self.ctx.lib.ghmm_stats_upload(self.ctx.h, self.h, _d(v)), self.ctx.lib)

    def close(self):
        if self.h:
            self.ctx.lib.ghmm_stats_destroy(self.ctx.h, self.h)
            self.h = None

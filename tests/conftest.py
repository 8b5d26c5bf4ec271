import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _load import ghmm as _ghmm  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def G():
    return _ghmm()


class Case:
    """One function-level golden dump of the real reference (tests/golden/*.npz)."""

    def __init__(self, G, name):
        d = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.name, self.d = name, d
        self.model0 = G.HostModel(*(d["model0." + k] for k in ("A", "c", "mean", "inv_var", "det")))
        self.model1 = G.HostModel(*(d["model1." + k] for k in ("A", "c", "mean", "inv_var", "det")))
        self.U = len([k for k in d.files if k.endswith(".X")])
        self.lens = np.array([d[f"u{u}.X"].shape[0] for u in range(self.U)], dtype=np.int32)
        self.X = np.concatenate([d[f"u{u}.X"] for u in range(self.U)])
        self.N, self.M, self.D = self.model0.N, self.model0.M, self.model0.D

    def frames(self, key):
        """per-frame reference array `key` of all utterances, back to back"""
        return np.concatenate([self.d[f"u{u}.{key}"] for u in range(self.U)])

    def logliks(self):
        return np.array([self.d[f"u{u}.loglik"][0] for u in range(self.U)])

    def stats(self):
        return np.concatenate([self.d["stats." + k].ravel() for k in
                               ("num_a", "den_a", "den_c", "num_c", "num_mu", "num_var",
                                "loglik", "n_utt")])


CASES = ["bundled186_m1", "bundled13_m3", "synth39_m8", "synth39_m8_refinit", "synth39_m64"]


@pytest.fixture(scope="session", params=CASES)
def case(request, G):
    return Case(G, request.param)


@pytest.fixture(scope="session")
def load_case(G):
    return lambda name: Case(G, name)

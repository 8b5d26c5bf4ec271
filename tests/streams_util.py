"""Second feature stream for the several-streams (param_number = 2) tests: the reference ships
one stream of 9-d frames per utterance; the second one is derived from it deterministically —
the two-frame difference of the first five coefficients, x[t+1] - x[t-1] (clamped at the ends).
Used by tests/golden/make_golden_streams.py (which feeds it to the REAL reference) and by the
tests (which feed the same frames to the oracle and the GPU)."""
import numpy as np


def second_stream(X, D2=5):
    X = np.asarray(X, dtype=np.float64)
    T = X.shape[0]
    nxt = X[np.minimum(np.arange(T) + 1, T - 1), :D2]
    prv = X[np.maximum(np.arange(T) - 1, 0), :D2]
    return np.ascontiguousarray(nxt - prv)

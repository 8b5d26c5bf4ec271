"""CPU-side pins (no GPU): the oracle (oracle/ghmm_oracle.c) against golden dumps of
the REAL reference (tests/golden/, produced by make_golden.py from oracle/_ref),
and the Viterbi definition against brute-force path enumeration."""
import itertools
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from conftest import GOLDEN


def test_oracle_estep_is_bit_exact_vs_reference(case):
    """b, post, alpha^, beta^, c_t, log P and every accumulator: identical bits to what
    the reference's own functions produced (TF:1380-1841 driven by ref_harness.c)."""
    stats, out = O.estep(case.model0, case.X, case.lens)
    for key in ("b", "post", "alpha", "beta", "scale"):
        ref = case.frames(key)
        assert np.array_equal(out[key].reshape(ref.shape), ref, equal_nan=True), key
    assert np.array_equal(out["loglik"], case.logliks())
    assert np.array_equal(stats, case.stats())


def test_oracle_mstep_is_bit_exact_vs_reference(case):
    """updating_transition_probab / updating_mix_param / calc_det / inv_matrix, TF:332-346"""
    new = O.mstep(case.model0, case.stats())
    for a, b in zip(new.arrays(), case.model1.arrays()):
        assert np.array_equal(a, b, equal_nan=True)


def cfmt(x):
    """printf("%f") as glibc prints it (python drops the sign of a NaN)"""
    if np.isnan(x):
        return "-nan" if np.signbit(x) else "nan"
    return f"{x:f}"


@pytest.fixture(scope="module")
def whole():
    return json.load(open(os.path.join(GOLDEN, "whole_program.json")))


def test_oracle_training_loop_matches_reference_programs(G, whole):
    """EM driver TF:238-358 from the reference-identical initial model: the iteration
    count and the printed mean log-likelihood of the 13 single-utterance runs and the
    13-utterance 3-mixture run of the real executable."""
    models = np.load(os.path.join(GOLDEN, "train13_m1_models.npz"))
    for word, exp in whole["train13_m1"].items():
        X = G.perfil_read(os.path.join(GOLDEN, "perfil", f"mean_{word}.perfil"))
        hm0 = G.HostModel.init_from(X, [len(X)], 6, 1)
        hm, it, mp, _ = O.train(hm0, X, [len(X)])
        assert it == exp["iterations"], word
        assert f"{mp:f}" == f"{exp['mean_probability']:f}", word
        for k, v in zip(("A", "c", "mean", "inv_var", "det"), hm.arrays()):
            assert np.array_equal(v, models[f"{word}.{k}"].reshape(v.shape)), (word, k)
    Xs = [G.perfil_read(os.path.join(GOLDEN, "perfil", fn)) for fn in whole["mean_list"]]
    lens = [len(x) for x in Xs]
    X = np.concatenate(Xs)
    hm, it, mp, _ = O.train(G.HostModel.init_from(X, lens, 6, 3), X, lens)
    exp = whole["train_all13_m3"]
    assert it == exp["iterations"] and f"{mp:f}" == f"{exp['mean_probability']:f}"
    for k, v in zip(("A", "c", "mean", "inv_var", "det"), hm.arrays()):
        assert np.array_equal(v, np.array(exp["model"][k]).reshape(v.shape)), k


def test_oracle_training_synth39(G, whole):
    exp = whole["train_synth39_m8"]
    mean, std = G.synth_truth(10, 8, 39)
    X = G.synth_utterances(mean, std, exp["lens"], first_utt=exp["first_utt"])
    hm, it, mp, _ = O.train(G.HostModel.init_from(X, exp["lens"], 10, 8), X, exp["lens"])
    assert it == exp["iterations"] and f"{mp:f}" == f"{exp['mean_probability']:f}"
    ref = np.load(os.path.join(GOLDEN, "train_synth39_m8_model.npz"))
    for k, v in zip(("A", "c", "mean", "inv_var", "det"), hm.arrays()):
        assert np.array_equal(v, ref[k].reshape(v.shape), equal_nan=True), k


def test_oracle_recognition_matches_reference_program(G, whole):
    """RF:326-388 on the 13 bundled utterances x 13 diagonal models: every printed score
    (finite, -inf and nan alike) and the bubble-sorted ranking of the real executable."""
    models = np.load(os.path.join(GOLDEN, "train13_m1_models.npz"))
    words = whole["words"]
    hms = [G.HostModel(*(models[f"{w}.{k}"] for k in ("A", "c", "mean", "inv_var", "det")))
           for w in words]
    kinds = set()
    for blk, fn in zip(whole["recog13_m1"]["blocks"], whole["mean_list"]):
        X = G.perfil_read(os.path.join(GOLDEN, "perfil", fn))
        scores = np.array([O.score(hm, X) for hm in hms])
        order = O.sort_scores(scores)
        got = [[words[i], cfmt(scores[i])] for i in order]
        assert got == blk["ranking"], blk["spoken"]
        kinds |= {("nan" if np.isnan(s) else "inf" if np.isinf(s) else "finite") for s in scores}
    assert kinds == {"nan", "inf", "finite"}  # the reference's numerical artefacts are covered


def test_viterbi_lattice_against_brute_force():
    """The reference has no Viterbi (parity unpinned by it): the oracle's definition is
    pinned by enumerating every state path on small lattices."""
    rng = np.random.default_rng(7)
    for N, T in [(2, 1), (2, 5), (3, 6), (4, 5), (3, 7)]:
        for trial in range(6):
            A = rng.random((N, N))
            if trial % 2 == 0:  # left-to-right band like the reference's models
                A = np.triu(A) - np.triu(A, 2)
            A /= A.sum(1, keepdims=True)
            logb = np.log(rng.random((T, N)))
            with np.errstate(divide="ignore"):
                la = np.log(A)
            best, best_path = -np.inf, None
            for p in itertools.product(range(N), repeat=T):
                if p[0] != 0 or p[-1] != N - 1:
                    continue
                s = logb[0, p[0]]
                for t in range(1, T):
                    s = (s + la[p[t - 1], p[t]]) + logb[t, p[t]]
                if s > best:
                    best, best_path = s, p
            path, score = O.viterbi_lattice(A, logb)
            if best_path is None:
                assert score == -np.inf
            else:
                assert list(path) == list(best_path)
                assert score == pytest.approx(best, rel=1e-14)


def test_viterbi_ties_take_lowest_predecessor():
    A = np.array([[0.5, 0.5, 0.0], [0.0, 0.5, 0.5], [0.0, 0.0, 0.5]])  # every step costs log .5
    logb = np.zeros((6, 3))
    path, _ = O.viterbi_lattice(A, logb)
    # all paths through the band score the same: the lowest predecessor wins at every
    # back-pointer, i.e. the path leaves state 0 as late as possible
    assert list(path) == [0, 0, 0, 0, 1, 2]


def test_log_emission_consistent_with_linear(load_case):
    c = load_case("synth39_m8")
    lb = O.log_emission(c.model0, c.X)
    assert np.allclose(lb, np.log(c.frames("b")), rtol=1e-12, atol=0)


# ------------------------------------------------------------- several feature streams

@pytest.fixture(scope="module")
def streams():
    return json.load(open(os.path.join(GOLDEN, "streams_p2.json")))


def stream_data(G, streams, idx):
    """The two streams of the bundled utterances `idx`: the 9-d frames and their 5-d
    differences (tests/streams_util.py — what make_golden_streams.py fed the real reference)."""
    from streams_util import second_stream
    Xs = [G.perfil_read(os.path.join(GOLDEN, "perfil", streams["mean_list"][k])) for k in idx]
    lens = [len(x) for x in Xs]
    return [np.concatenate(Xs), np.concatenate([second_stream(x, streams["D2"]) for x in Xs])], lens


def golden_stream_models(G, rec):
    m = rec["model"]
    return [G.HostModel(m["A"], s["c"], s["mean"], s["inv_var"], s["det"]) for s in m["streams"]]


def test_oracle_streams_match_the_reference_trainer(G, streams):
    """param_number = 2 (TF:1406-1409, 1501-1504, 1607-1610; calc_symbol_probab / calc_mix_param per
    stream TF:278-315): the real trainer's iteration count, printed mean log-likelihood and every
    double of the model it wrote, from the reference-identical initial model of each stream."""
    exp = streams["train_all13_p2"]
    Xs, lens = stream_data(G, streams, range(13))
    hm0 = [G.HostModel.init_from(Xs[p], lens, 6, exp["model"]["M"][p]) for p in range(2)]
    new, it, mp = O.train_streams(hm0, Xs, lens)
    assert it == exp["iterations"]
    assert f"{mp:f}" == f"{exp['mean_probability']:f}"
    for got, ref in zip(new, golden_stream_models(G, exp)):
        for a, b in zip(got.arrays(), ref.arrays()):
            assert np.array_equal(a, b.reshape(a.shape))
    by_word = {fn[5:-7]: k for k, fn in enumerate(streams["mean_list"])}
    for w in streams["words"][:4]:
        exp = streams["train13_p2"][w]
        Xw, lw = stream_data(G, streams, [by_word[w]])
        new, it, mp = O.train_streams([G.HostModel.init_from(Xw[p], lw, 6, 1) for p in range(2)], Xw, lw)
        assert it == exp["iterations"] and f"{mp:f}" == f"{exp['mean_probability']:f}", w
        for got, ref in zip(new, golden_stream_models(G, exp)):
            for a, b in zip(got.arrays(), ref.arrays()):
                assert np.array_equal(a, b.reshape(a.shape)), w


def test_oracle_streams_match_the_reference_recogniser(G, streams):
    """RF:349-366 with two streams: the recogniser's printed scores of 13 utterances x 13
    two-stream models, -inf / nan artefacts included, and its NaN-blind ranking."""
    words = streams["words"]
    models = [golden_stream_models(G, streams["train13_p2"][w]) for w in words]
    for u, blk in enumerate(streams["recog13_p2"]["blocks"]):
        Xu, _ = stream_data(G, streams, [u])
        scores = np.array([O.score_streams(m, Xu) for m in models])
        order = O.sort_scores(scores)
        assert [words[i] for i in order] == [w for w, _ in blk["ranking"]], blk["spoken"]
        for i, (w, txt) in zip(order, blk["ranking"]):
            assert cfmt(scores[i]).lstrip("-") == txt.lstrip("-") or cfmt(scores[i]) == txt, (blk["spoken"], w)

"""Parity of the HIP path (through the C ABI) with the reference — GPU box only.

Three kinds of check:
  * against golden dumps of the REAL reference (tests/golden/*.npz, *.json),
  * against the oracle (oracle/ghmm_oracle.c, itself bit-exact vs the reference) on
    seeded synthetic inputs the goldens do not cover,
  * size-independent properties at BASELINE's full sizes.

Tolerance: north_star asks for log-likelihoods and re-estimated parameters within
1e-5 relative.  The assertions below use RTOL = 1e-8 on every intermediate (the GPU
differs from the reference only by FMA contraction, summation order and device
exp/log), so a pass here is three orders of magnitude inside the bar.
"""
import json
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from conftest import GOLDEN
from _load import PKG_DIR

pytestmark = pytest.mark.gpu

RTOL = 1e-8          # asserted
NORTH_STAR_RTOL = 1e-5  # the bar


def assert_close(got, ref, rtol=RTOL, floor=1e-13, what="", rows=False):
    """|got-ref| <= rtol*|ref| + floor*scale; non-finite entries must agree in kind.
    scale = max|ref| over the whole array, or (rows=True: per-frame quantities b, alpha^, beta^,
    gamma, post, whose frames span tens of decades) over the entry's own frame, so that an
    entry is only excused when it is 13 decades below the largest value OF ITS FRAME."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    if rows and ref.ndim >= 2:
        r2 = np.abs(ref.reshape(ref.shape[0], -1))
        r2 = np.where(np.isfinite(r2), r2, 0.0)
        rmax = r2.max(axis=1, keepdims=True)
        # (a frame whose reference entries are all 0 — e.g. beta^ of an utterance shorter than the
        # model, underflown in the reference's scaling — has no scale of its own: the array's)
        rmax = np.where(rmax > 0.0, rmax, r2.max() if r2.size else 0.0)
        scale = np.broadcast_to(rmax, r2.shape).ravel()
    else:
        scale = None
    got, ref = got.ravel(), ref.ravel()
    assert got.shape == ref.shape, what
    fin = np.isfinite(ref)
    assert np.array_equal(np.isnan(got), np.isnan(ref)), f"{what}: NaN pattern differs"
    assert np.array_equal(got[~fin & ~np.isnan(ref)], ref[~fin & ~np.isnan(ref)]), f"{what}: inf differs"
    if fin.any():
        sc = np.abs(ref[fin]).max() if scale is None else scale[fin]
        err = np.abs(got[fin] - ref[fin])
        tol = rtol * np.abs(ref[fin]) + floor * sc
        worst = (err / np.maximum(tol, 1e-320)).max()
        assert worst <= 1.0, f"{what}: worst error {worst:.3g} x tolerance"


def assert_frames(got, ref, what, rtol=RTOL):
    """Per-frame arrays: every entry against the largest of its own frame (assert_close, rows)."""
    ref = np.asarray(ref)
    got = np.asarray(got).reshape(ref.shape)
    assert_close(got, ref, rtol=rtol, what=what, rows=True)


@pytest.fixture(scope="module")
def ctx(G):
    c = G.Context(0)
    yield c
    c.close()


# ------------------------------------------------------------------ goldens

def test_rows_against_reference_dumps(G, ctx, case):
    """Every §8(a) row on its own, against what the reference's functions produced:
    b, post (TF:1749-1841), alpha^, c_t, log P (TF:1380-1443, 1536-1553), beta^
    (TF:1463-1516), gamma, and all accumulators (TF:1577-1727)."""
    model, corpus = ctx.model(case.model0), ctx.corpus(case.X, case.lens)
    F, N, M, D = corpus.frames, case.N, case.M, case.D
    ctx.emission(model, corpus, True)
    assert_frames(ctx.fetch(G.BUF_B, (F, N)), case.frames("b"), "b")
    assert_frames(ctx.fetch(G.BUF_POST, (F, N * M)), case.frames("post").reshape(F, -1), "post")
    ctx.forward(model, corpus)
    assert_frames(ctx.fetch(G.BUF_ALPHA, (F, N)), case.frames("alpha"), "alpha")
    assert_close(ctx.fetch(G.BUF_SCALE, (F,)), case.frames("scale"), floor=0.0, what="scale")
    assert_close(ctx.fetch(G.BUF_LOGLIK, (case.U,)), case.logliks(), what="loglik")
    ctx.backward(model, corpus)
    beta, scale = case.frames("beta"), case.frames("scale")
    assert_frames(ctx.fetch(G.BUF_BETA, (F, N)), beta, "beta")
    gamma_ref = case.frames("alpha") * beta / scale[:, None]
    assert_frames(ctx.fetch(G.BUF_GAMMA, (F, N)), gamma_ref, "gamma")
    stats = ctx.stats(N, M, D)
    ctx.accumulate(model, corpus, stats)
    got, ref = G.split_stats(stats.download(), N, M, D), G.split_stats(case.stats(), N, M, D)
    for k in ref:
        assert_close(got[k], ref[k], what="stats." + k)
    for o in (model, corpus, stats):
        o.close()


def test_fused_estep_and_mstep_against_reference_dumps(G, ctx, case):
    model, corpus = ctx.model(case.model0), ctx.corpus(case.X, case.lens)
    stats = ctx.stats(case.N, case.M, case.D)
    ctx.estep(model, corpus, stats)
    got = G.split_stats(stats.download(), case.N, case.M, case.D)
    ref = G.split_stats(case.stats(), case.N, case.M, case.D)
    for k in ref:
        assert_close(got[k], ref[k], what="stats." + k)
    # M-step from the reference's own accumulators: isolates TF:332-346
    stats.upload(case.stats())
    ctx.mstep(model, stats)
    new = model.get()
    for name, a, b in zip(("A", "c", "mean", "inv_var", "det"), new.arrays(), case.model1.arrays()):
        assert_close(a, b, rtol=1e-12, floor=0.0, what="model1." + name)
    for o in (model, corpus, stats):
        o.close()


@pytest.mark.parametrize("name", ["bundled186_m1", "bundled13_m3", "synth39_m8_refinit"])
def test_device_initial_model_against_reference(G, ctx, load_case, name):
    """ghmm_model_init = creating_initial_model TF:732-1317 with the distance/accumulation
    passes on the GPU: the reference's own initial model (needle components, det = 1e-195,
    included) within 1e-9 — sums are taken in a different order, nothing else differs."""
    case = load_case(name)
    corpus = ctx.corpus(case.X, case.lens)
    model = ctx.model(case.model0)            # shapes only; overwritten
    got = model.init_from(corpus)
    for nm, a, b in zip(("A", "c", "mean", "inv_var", "det"), got.arrays(), case.model0.arrays()):
        assert_close(a, b, rtol=1e-9, floor=0.0, what=f"{name} init.{nm}")
    host = G.HostModel.init_from(case.X, case.lens, case.N, case.M)
    for a, b in zip(host.arrays(), case.model0.arrays()):
        assert np.array_equal(a, b, equal_nan=True)      # the host version is bit-exact
    model.close()
    corpus.close()


@pytest.fixture(scope="module")
def whole():
    return json.load(open(os.path.join(GOLDEN, "whole_program.json")))


def gpu_train(G, ctx, hm0, X, lens, threshold=1e-3, max_iter=200):
    """EM driver TF:238-358 on the GPU (what train_main.c does)."""
    em = __import__("ghmm_amd").em
    model, corpus = ctx.model(hm0), ctx.corpus(X, lens)
    backend = em.HipBackend(G, ctx, model, corpus)
    it, p = em.EMDriver(backend).train(threshold, max_iter)
    hm = model.get()
    for o in (model, corpus, backend.stats):
        o.close()
    return hm, it, p / len(lens)


def test_training_loop_matches_reference_programs(G, ctx, whole):
    """Iteration counts and mean log-likelihoods of the real trainer executable."""
    models = np.load(os.path.join(GOLDEN, "train13_m1_models.npz"))
    for word, exp in whole["train13_m1"].items():
        X = G.perfil_read(os.path.join(GOLDEN, "perfil", f"mean_{word}.perfil"))
        hm, it, mp = gpu_train(G, ctx, G.HostModel.init_from(X, [len(X)], 6, 1), X, [len(X)])
        assert it == exp["iterations"], word
        assert mp == pytest.approx(exp["mean_probability"], rel=1e-9, abs=1e-6), word
        for k, v in zip(("A", "c", "mean", "inv_var", "det"), hm.arrays()):
            assert_close(v, models[f"{word}.{k}"], rtol=1e-7, what=f"{word}.{k}")
    Xs = [G.perfil_read(os.path.join(GOLDEN, "perfil", fn)) for fn in whole["mean_list"]]
    lens = [len(x) for x in Xs]
    X = np.concatenate(Xs)
    hm, it, mp = gpu_train(G, ctx, G.HostModel.init_from(X, lens, 6, 3), X, lens)
    exp = whole["train_all13_m3"]
    assert it == exp["iterations"]
    assert mp == pytest.approx(exp["mean_probability"], rel=1e-9, abs=1e-6)
    for k, v in zip(("A", "c", "mean", "inv_var", "det"), hm.arrays()):
        assert_close(v, np.array(exp["model"][k]), rtol=1e-6, what="all13." + k)


def test_training_synth39_matches_reference_program(G, ctx, whole):
    exp = whole["train_synth39_m8"]
    mean, std = G.synth_truth(10, 8, 39)
    X = G.synth_utterances(mean, std, exp["lens"], first_utt=exp["first_utt"])
    hm, it, mp = gpu_train(G, ctx, G.HostModel.init_from(X, exp["lens"], 10, 8), X, exp["lens"])
    assert it == exp["iterations"]
    assert mp == pytest.approx(exp["mean_probability"], rel=1e-9, abs=1e-6)
    ref = np.load(os.path.join(GOLDEN, "train_synth39_m8_model.npz"))
    for k, v in zip(("A", "c", "mean", "inv_var", "det"), hm.arrays()):
        assert_close(v, ref[k], rtol=1e-6, what="synth39." + k)


def cfmt(x):
    if np.isnan(x):
        return "-nan" if np.signbit(x) else "nan"
    return f"{x:f}"


def test_recognition_scores_match_reference_program(G, ctx, whole):
    """RF:326-374: forward scores of 13 utterances x 13 models, the reference's printed
    values (finite ones within 1e-9, -inf / nan in the same places) and its ranking."""
    models = np.load(os.path.join(GOLDEN, "train13_m1_models.npz"))
    words = whole["words"]
    hms = [G.HostModel(*(models[f"{w}.{k}"] for k in ("A", "c", "mean", "inv_var", "det")))
           for w in words]
    Xs = [G.perfil_read(os.path.join(GOLDEN, "perfil", fn)) for fn in whole["mean_list"]]
    corpus = ctx.corpus(np.concatenate(Xs), [len(x) for x in Xs])
    scores = np.zeros((len(words), len(Xs)))
    for k, hm in enumerate(hms):
        m = ctx.model(hm)
        scores[k] = ctx.score(m, corpus)
        m.close()
    corpus.close()
    for u, blk in enumerate(whole["recog13_m1"]["blocks"]):
        order = O.sort_scores(scores[:, u])
        assert [words[i] for i in order] == [w for w, _ in blk["ranking"]], blk["spoken"]
        for i, (w, txt) in zip(order, blk["ranking"]):
            if "nan" in txt or "inf" in txt:
                assert cfmt(scores[i, u]).lstrip("-") == txt.lstrip("-"), (blk["spoken"], w)
            else:
                assert scores[i, u] == pytest.approx(float(txt), rel=1e-9, abs=2e-6), (blk["spoken"], w)


# ------------------------------------------------------------- command lines

def _write_lists(tmp, files, name):
    p = os.path.join(tmp, name)
    with open(p, "w") as f:
        f.write("\n".join(files) + "\n")
    return p


def test_train_command_line(G, whole, tmp_path):
    """bin/hmm-continuous-train-fs with the reference's argv (train/test/Run Arguments.txt)."""
    exe = os.path.join(PKG_DIR, "bin", "hmm-continuous-train-fs")
    word = "vc_186_f_03_ap_0225"
    lst = _write_lists(str(tmp_path), [os.path.join(GOLDEN, "perfil", f"mean_{word}.perfil")], "parameters.txt")
    out = os.path.join(str(tmp_path), "result.mean.hmm")
    p = subprocess.run([exe, word, "6", "1", "1", lst, out], stdout=subprocess.PIPE)
    assert p.returncode == 0, p.stdout.decode()
    # report name: strtok(".") semantics (TF:205-207) -> cut at the FIRST dot of the file name
    first_dot = out.index(".", 1)
    report = open(out[:first_dot] + ".txt").read().split("\n")
    exp = whole["train13_m1"][word]
    assert report[0].startswith("Continuous HMM created using forward backward algorithm (diagonal")
    assert report[2] == f"word: {word} " and report[3] == "number of states: 6 "
    assert report[9] == f"mean probability: {exp['mean_probability']:f} "
    assert report[10] == f"number of iterations: {exp['iterations']} "
    hm = G.HostModel.read(out)
    models = np.load(os.path.join(GOLDEN, "train13_m1_models.npz"))
    assert hm.word == word
    for k, v in zip(("A", "c", "mean", "inv_var", "det"), hm.arrays()):
        assert_close(v, models[f"{word}.{k}"], rtol=1e-7, what=k)
    # fewer than 7 arguments: usage text and exit status 1 (TF:179-191)
    p = subprocess.run([exe, "w", "6", "1"], stdout=subprocess.PIPE)
    assert p.returncode == 1 and p.stdout.startswith(b"Usage: hmm_continuous_fs")


def test_recognition_command_line(G, whole, tmp_path):
    """bin/recognition-continuous-test-fs with the reference's argv: the report must be
    the reference's report line for line (dates and CPU times aside)."""
    exe = os.path.join(PKG_DIR, "bin", "recognition-continuous-test-fs")
    tmp = str(tmp_path)
    models = np.load(os.path.join(GOLDEN, "train13_m1_models.npz"))
    paths = []
    for w in whole["words"]:
        hm = G.HostModel(*(models[f"{w}.{k}"] for k in ("A", "c", "mean", "inv_var", "det")), word=w)
        paths.append(os.path.join(tmp, w + ".hmm"))
        hm.write(paths[-1], 8)
    ml = _write_lists(tmp, paths, "models.txt")
    fl = _write_lists(tmp, [os.path.join(GOLDEN, "perfil", fn) for fn in whole["mean_list"]], "mean_list.txt")
    wl = _write_lists(tmp, whole["words"], "words.txt")
    out = os.path.join(tmp, "hmm-result.txt")
    p = subprocess.run([exe, "1", ml, "1", fl, wl, out], stdout=subprocess.PIPE)
    assert p.returncode == 0, p.stdout.decode()
    got = [l for l in open(out).read().split("\n")
           if not l.startswith("Date and time") and "recognition time" not in l
           and not l.startswith("Model name")]
    assert got == whole["recog13_m1"]["report"]


# ------------------------------------------------------------ oracle, seeded

def synth_case(G, N, M, D, lens, perturb=0.05, first=0, dense_A=False, seed=3):
    mean, std = G.synth_truth(N, M, D)
    X = G.synth_utterances(mean, std, lens, first_utt=first)
    hm = G.synth_start_model(mean, std, perturb)
    if dense_A:
        rng = np.random.default_rng(seed)
        A = rng.random((N, N)) + 0.05
        hm.A[:] = A / A.sum(1, keepdims=True)
    return hm, X, np.asarray(lens, dtype=np.int32)


@pytest.mark.parametrize("N,M,D,lens,dense", [
    (10, 8, 39, [300, 211, 128, 77, 64, 5, 1, 2, 33, 500], False),   # ragged, T < N, T = 1
    (10, 8, 39, [90, 120, 65], True),                                # dense A: general recursion
    (3, 2, 5, [40, 17, 64, 65, 63, 1], False),                       # tiny model, tile edges
    (16, 4, 13, [70, 80], True),                                     # N = group width
    (20, 2, 9, [60, 45, 81], True),                                  # N > 16: one wave per utterance
    (6, 3, 40, [64, 128], False),                                    # even D (LDS row padding)
    (7, 3, 39, [100, 61, 16, 15, 17], False),                        # D=39 fast kernels, M padded 3 -> 4
    (5, 5, 39, [80, 48], True),                                      # M padded 5 -> 8, dense A
    (3, 16, 39, [90, 33], False),                                    # one state per tile
    (2, 32, 39, [70, 64], False),                                    # a state spans two tiles
    (1, 1, 1, [10, 1, 3], False),                                    # the smallest model
    (2, 5, 3, [33, 20], True),
    (33, 2, 4, [50, 70], True),                                      # N > 16: one wave per utterance
    (10, 8, 39, [5000, 2999], False),                                # long utterances (reference cap: 500)
    (2, 96, 80, [40, 33], False),                                    # a state too large for the M-step's LDS staging
])
def test_estep_against_oracle(G, ctx, N, M, D, lens, dense):
    hm, X, lens = synth_case(G, N, M, D, lens, dense_A=dense)
    ref_stats, ref = O.estep(hm, X, lens)
    model, corpus = ctx.model(hm), ctx.corpus(X, lens)
    stats = ctx.stats(N, M, D)
    ctx.estep(model, corpus, stats)
    F = corpus.frames
    assert_frames(ctx.fetch(G.BUF_B, (F, N)), ref["b"], "b")
    assert_frames(ctx.fetch(G.BUF_POST, (F, N * M)), ref["post"].reshape(F, -1), "post")
    assert_frames(ctx.fetch(G.BUF_ALPHA, (F, N)), ref["alpha"], "alpha")
    assert_frames(ctx.fetch(G.BUF_BETA, (F, N)), ref["beta"], "beta")
    assert_close(ctx.fetch(G.BUF_SCALE, (F,)), ref["scale"], floor=0.0, what="scale")
    assert_close(ctx.fetch(G.BUF_LOGLIK, (len(lens),)), ref["loglik"], what="loglik")
    got, refs = G.split_stats(stats.download(), N, M, D), G.split_stats(ref_stats, N, M, D)
    for k in refs:
        assert_close(got[k], refs[k], what="stats." + k)
    ctx.mstep(model, stats)
    new, ref_new = model.get(), O.mstep(hm, ref_stats)
    for name, a, b in zip(("A", "c", "mean", "inv_var", "det"), new.arrays(), ref_new.arrays()):
        assert_close(a, b, rtol=1e-7, what="mstep." + name)
    for o in (model, corpus, stats):
        o.close()


@pytest.mark.parametrize("tier", [1, 2])
def test_both_kernel_tiers(G, ctx, tier, load_case):
    """GHMM_OPT_KERNELS: 1 = vector-ALU kernels only, 2 = matrix-core kernels; both must
    reproduce the reference dumps (well-conditioned and needle-component models) and agree
    with each other far inside the tolerance."""
    ctx.set_option(G.OPT_KERNELS, tier)
    try:
        for name in ("synth39_m8", "synth39_m8_refinit", "bundled13_m3", "synth39_m64"):
            case = load_case(name)
            model, corpus = ctx.model(case.model0), ctx.corpus(case.X, case.lens)
            stats = ctx.stats(case.N, case.M, case.D)
            ctx.estep(model, corpus, stats)
            F = corpus.frames
            assert_frames(ctx.fetch(G.BUF_B, (F, case.N)), case.frames("b"), f"{name} b")
            assert_frames(ctx.fetch(G.BUF_POST, (F, case.N * case.M)),
                          case.frames("post").reshape(F, -1), f"{name} post")
            got = G.split_stats(stats.download(), case.N, case.M, case.D)
            ref = G.split_stats(case.stats(), case.N, case.M, case.D)
            for k in ref:
                assert_close(got[k], ref[k], what=f"{name} stats.{k}")
            for o in (model, corpus, stats):
                o.close()
    finally:
        ctx.set_option(G.OPT_KERNELS, 0)


@pytest.mark.parametrize("N,M,D,lens,dense,delta", [
    (10, 8, 39, [300, 211, 77, 5, 1, 2, 33], False, 1),   # ragged, T < N (kappa = 0), T = 1
    (6, 2, 7, [50, 60, 3], True, 1),                      # dense A: general recursion
    (6, 2, 7, [50, 60, 9], True, 3),                      # wider transition band
    (20, 2, 9, [60, 45], True, 1),                        # N > 16: one wave per utterance
])
def test_paired_scans_equal_the_reference_order(G, ctx, N, M, D, lens, dense, delta):
    """The default tier runs the backward recursion with its own normaliser beside the forward
    pass and forms gamma / xi / beta^ from the scaling identity sum_i alpha^ beta^ = c_t kappa
    (ghmm_pair.hpp); GHMM_OPT_KERNELS = 1 keeps the reference's order (calc_beta scaled by
    calc_alpha's c_t, TF:1463-1516).  Same gamma, beta^, log P and statistics."""
    hm, X, lens = synth_case(G, N, M, D, lens, dense_A=dense)
    model, corpus = ctx.model(hm), ctx.corpus(X, lens)
    F = corpus.frames
    out = {}
    try:
        ctx.set_option(G.OPT_DELTA, delta)
        for tier in (1, 0):
            ctx.set_option(G.OPT_KERNELS, tier)
            stats = ctx.stats(N, M, D)
            ctx.estep(model, corpus, stats)
            out[tier] = dict(stats=stats.download(), beta=ctx.fetch(G.BUF_BETA, (F, N)),
                             gamma=ctx.fetch(G.BUF_GAMMA, (F, N)), ll=ctx.fetch(G.BUF_LOGLIK, (len(lens),)))
            stats.close()
    finally:
        ctx.set_option(G.OPT_KERNELS, 0)
        ctx.set_option(G.OPT_DELTA, 1)
    for k in ("gamma", "beta", "ll", "stats"):
        assert_close(out[0][k], out[1][k], rtol=1e-9, what=k)
    for o in (model, corpus):
        o.close()


@pytest.mark.parametrize("seed", list(range(12)))
def test_tiers_agree_on_random_shapes(G, ctx, seed):
    """Seeded random shapes (states, mixtures, coefficients, ragged lengths down to one frame,
    band-diagonal or dense A, transition band 0..3, linear or robust emission): the default tier
    (matrix cores, paired scans) and the vector-ALU / reference-order tier give the same E-step."""
    rng = np.random.default_rng(1000 + seed)
    N, M, D = int(rng.integers(1, 21)), int(rng.integers(1, 10)), int(rng.integers(1, 41))
    lens = [int(x) for x in rng.integers(1, 90, size=int(rng.integers(1, 7)))]
    if seed % 4 == 0:
        lens.append(0)
    dense, delta, robust = bool(rng.integers(0, 2)), int(rng.integers(0, 4)), int(rng.integers(0, 2))
    hm, X, lens = synth_case(G, N, M, D, lens, dense_A=dense, seed=seed)
    model, corpus = ctx.model(hm), ctx.corpus(X, lens)
    F = corpus.frames
    out = {}
    try:
        ctx.set_option(G.OPT_DELTA, delta)
        ctx.set_option(G.OPT_ROBUST, robust)
        for tier in (1, 0):
            ctx.set_option(G.OPT_KERNELS, tier)
            stats = ctx.stats(N, M, D)
            ctx.estep(model, corpus, stats)
            out[tier] = dict(stats=stats.download(), gamma=ctx.fetch(G.BUF_GAMMA, (F, N)),
                             beta=ctx.fetch(G.BUF_BETA, (F, N)), ll=ctx.fetch(G.BUF_LOGLIK, (len(lens),)))
            stats.close()
    finally:
        ctx.set_option(G.OPT_KERNELS, 0)
        ctx.set_option(G.OPT_DELTA, 1)
        ctx.set_option(G.OPT_ROBUST, 0)
    what = f"N={N} M={M} D={D} lens={list(lens)} dense={dense} delta={delta} robust={robust}"
    for k in ("ll", "gamma", "beta", "stats"):
        assert_close(out[0][k], out[1][k], rtol=1e-9, what=f"{k} ({what})")
    for o in (model, corpus):
        o.close()


def test_beta_on_demand_follows_the_model(G, ctx):
    """ghmm_estep leaves beta^ out; ghmm_fetch forms it from the E-step's alpha^ / W while the model
    is unchanged, and refuses once ghmm_mstep has replaced the parameters they belong to."""
    hm, X, lens = synth_case(G, 5, 2, 6, [40, 25])
    model, corpus = ctx.model(hm), ctx.corpus(X, lens)
    stats = ctx.stats(5, 2, 6)
    _, ref = O.estep(hm, X, lens)
    ctx.estep(model, corpus, stats)
    assert_frames(ctx.fetch(G.BUF_BETA, (corpus.frames, 5)), ref["beta"], "beta after estep")
    ctx.estep(model, corpus, stats)
    ctx.mstep(model, stats)
    with pytest.raises(G.GhmmError):
        ctx.fetch(G.BUF_BETA, (corpus.frames, 5))
    ctx.emission(model, corpus, True)      # the row API on the new model brings it back
    ctx.forward(model, corpus)
    ctx.backward(model, corpus)
    assert np.isfinite(ctx.fetch(G.BUF_BETA, (corpus.frames, 5))).all()
    for o in (model, corpus, stats):
        o.close()


def test_ten_em_iterations_track_the_oracle(G, ctx):
    """Fixed iteration count (the benchmark mode): the per-iteration log-likelihood and
    the final model stay within 1e-7 of the oracle over 10 E+M steps."""
    hm, X, lens = synth_case(G, 10, 8, 39, [120] * 24, perturb=0.15)
    ref_hm, it, _, trace = O.train(hm, X, lens, max_iter=10, fixed_iter=True)
    model, corpus = ctx.model(hm), ctx.corpus(X, lens)
    stats = ctx.stats(10, 8, 39)
    got = []
    for _ in range(10):
        ctx.estep(model, corpus, stats)
        got.append(stats.download()[-2])
        ctx.mstep(model, stats)
    assert_close(got, trace, rtol=1e-9, what="loglik trace")
    # (no monotonicity check: the reference re-estimates variances around the OLD mean,
    # TF:1720-1722, so its iteration is not exact EM and may dip)
    for name, a, b in zip(("A", "c", "mean", "inv_var", "det"), model.get().arrays(), ref_hm.arrays()):
        assert_close(a, b, rtol=1e-6, what="model." + name)
    for o in (model, corpus, stats):
        o.close()


def test_em_from_the_references_initial_model_tracks_the_oracle(G, ctx):
    """EM from creating_initial_model's result (TF:732-1317): within a few iterations some
    components collapse onto single frames (variances at the 1e-5 floor, conditioning ~1e7 around
    the data's centre), which is where the expanded Mahalanobis form needs its per-tile offsets
    and, for two such components in one tile, the direct form.  Log-likelihood trace and final
    model against the oracle over 8 iterations."""
    N, M, D = 10, 8, 39
    mean, std = G.synth_truth(N, M, D)
    lens = np.full(48, 90, dtype=np.int32)
    X = G.synth_utterances(mean, std, lens)
    hm = G.synth_start_model(mean, std, 0.05)
    model, corpus = ctx.model(hm), ctx.corpus(X, lens)
    hm0 = model.init_from(corpus)
    ref_hm, it, _, trace = O.train(hm0, X, lens, max_iter=8, fixed_iter=True)
    stats = ctx.stats(N, M, D)
    got = []
    for _ in range(8):
        ctx.estep(model, corpus, stats)
        got.append(stats.download()[-2])
        ctx.mstep(model, stats)
    iv = np.asarray(model.get().arrays()[3]).reshape(N * M, D)
    floored = int((iv > 9.0e4).all(axis=1).sum())
    assert floored >= 1, "the case is meant to contain collapsed components"
    assert_close(got, trace, rtol=1e-8, what="loglik trace")
    for name, a, b in zip(("A", "c", "mean", "inv_var", "det"), model.get().arrays(), ref_hm.arrays()):
        assert_close(a, b, rtol=1e-6, what="model." + name)
    for o in (model, corpus, stats):
        o.close()


def test_delta_option_widens_the_transition_band(G, ctx):
    hm, X, lens = synth_case(G, 6, 2, 7, [50, 60], dense_A=True)
    model, corpus = ctx.model(hm), ctx.corpus(X, lens)
    stats = ctx.stats(6, 2, 7)
    try:
        for delta in (0, 1, 3):
            ctx.set_option(G.OPT_DELTA, delta)
            ctx.estep(model, corpus, stats)
            ref, _ = O.estep(hm, X, lens, delta=delta, dumps=False)
            assert_close(stats.download(), ref, what=f"delta={delta}")
    finally:
        ctx.set_option(G.OPT_DELTA, 1)
    for o in (model, corpus, stats):
        o.close()


def test_grids_sized_for_part_of_the_device(G, ctx):
    """GHMM_OPT_CUS (a caller whose stream is CU-masked): the one-block-per-CU kernels sized for
    a few compute units give the oracle's sums like the full-device grids do."""
    hm, X, lens = synth_case(G, 10, 8, 39, [300, 211, 128, 77, 64, 500, 333, 90])
    ref, _ = O.estep(hm, X, lens, dumps=False)
    model, corpus = ctx.model(hm), ctx.corpus(X, lens)
    stats = ctx.stats(10, 8, 39)
    try:
        for cus in (3, 64, 0):
            ctx.set_option(G.OPT_CUS, cus)
            ctx.estep(model, corpus, stats)
            assert_close(stats.download(), ref, what=f"cus={cus}")
        assert ctx.get_option(G.OPT_CUS) >= 64   # 0 = the device's own count
    finally:
        ctx.set_option(G.OPT_CUS, 0)
    for o in (model, corpus, stats):
        o.close()


def test_viterbi_paths_identical_to_oracle(G, ctx):
    """State sequences bit-identical, scores within 1e-10 (index work: exact)."""
    for N, M, D, lens, dense in [(10, 8, 39, [300, 150, 64, 10, 1], False),
                                 (5, 3, 12, [40, 80, 33], True), (20, 2, 9, [70], True)]:
        hm, X, lens = synth_case(G, N, M, D, lens, dense_A=dense)
        model, corpus = ctx.model(hm), ctx.corpus(X, lens)
        path, score = ctx.viterbi(model, corpus)
        o = 0
        for u, T in enumerate(lens):
            p, s = O.viterbi(hm, X[o:o + T])
            assert np.array_equal(path[o:o + T], p), (N, u)
            if np.isfinite(s):
                assert score[u] == pytest.approx(s, rel=1e-10)
            else:
                assert score[u] == s
            o += T
        model.close()
        corpus.close()


def test_score_equals_training_loglik_and_oracle(G, ctx):
    hm, X, lens = synth_case(G, 10, 8, 39, [200, 100, 50])
    model, corpus = ctx.model(hm), ctx.corpus(X, lens)
    s = ctx.score(model, corpus)
    o = 0
    for u, T in enumerate(lens):
        assert s[u] == pytest.approx(O.score(hm, X[o:o + T]), rel=1e-10)
        o += T
    model.close()
    corpus.close()


def test_score_batch_is_the_vocabulary_loop(G, ctx, whole):
    """ghmm_score_batch (one emission launch over all word models' Gaussians + one forward
    launch over every (model, utterance) pair) = ghmm_score model by model, on the
    reference's 13 x 13 recognition case (NaN / -inf artefacts included) and on models
    with different state counts."""
    models = np.load(os.path.join(GOLDEN, "train13_m1_models.npz"))
    hms = [G.HostModel(*(models[f"{w}.{k}"] for k in ("A", "c", "mean", "inv_var", "det")))
           for w in whole["words"]]
    Xs = [G.perfil_read(os.path.join(GOLDEN, "perfil", fn)) for fn in whole["mean_list"]]
    corpus = ctx.corpus(np.concatenate(Xs), [len(x) for x in Xs])
    dms = [ctx.model(hm) for hm in hms]
    batch = ctx.score_batch(dms, corpus)
    single = np.stack([ctx.score(m, corpus) for m in dms])
    assert_close(batch, single, rtol=1e-12, what="13x13 batch vs single")
    for m in dms:
        m.close()
    corpus.close()
    cases = [synth_case(G, n, 4, 13, [70, 33, 90, 64, 1], dense_A=(n == 5)) for n in (5, 9, 16)]
    X, lens = cases[0][1], cases[0][2]
    corpus = ctx.corpus(X, lens)
    dms = [ctx.model(c[0]) for c in cases]
    batch = ctx.score_batch(dms, corpus)
    o = 0
    for u, T in enumerate(lens):
        for k, c in enumerate(cases):
            assert batch[k, u] == pytest.approx(O.score(c[0], X[o:o + T]), rel=1e-10)
        o += T
    for m in dms:
        m.close()
    corpus.close()


def test_robust_mode(G, ctx):
    """GHMM_OPT_ROBUST (per-frame max-normalised densities): same statistics where the
    reference is finite, finite scores where whole frames underflow in the reference's
    linear domain (its scale becomes 1/0 and the score NaN)."""
    hm, X, lens = synth_case(G, 10, 8, 39, [100, 80])
    model, corpus = ctx.model(hm), ctx.corpus(X, lens)
    stats = ctx.stats(10, 8, 39)
    ref, _ = O.estep(hm, X, lens, dumps=False)
    ctx.set_option(G.OPT_ROBUST, 1)
    try:
        ctx.estep(model, corpus, stats)
        assert_close(stats.download(), ref, what="robust stats")
        # the same data under a model whose variances are 49x too small: every linear
        # density is exp(-0.5 * 39 * 49) = 0, the reference's scale becomes 1/0 -> NaN
        sharp = hm.copy()
        sharp.inv_var *= 49.0
        sharp.det /= 49.0 ** 39
        m2 = ctx.model(sharp)
        assert np.isnan(O.score(sharp, X[:100]))
        assert np.all(np.isfinite(ctx.score(m2, corpus)))
        ctx.set_option(G.OPT_ROBUST, 0)
        assert np.all(np.isnan(ctx.score(m2, corpus)))
        m2.close()
    finally:
        ctx.set_option(G.OPT_ROBUST, 0)
    for o in (model, corpus, stats):
        o.close()


def test_empty_and_degenerate_inputs(G, ctx):
    hm, X, lens = synth_case(G, 4, 2, 6, [30, 0, 12])
    model, corpus = ctx.model(hm), ctx.corpus(X, lens)
    stats = ctx.stats(4, 2, 6)
    ctx.estep(model, corpus, stats)
    ref, _ = O.estep(hm, X, lens, dumps=False)   # a zero-length utterance adds nothing
    assert_close(stats.download(), ref, what="zero-length utterance")
    empty = ctx.corpus(np.zeros((0, 6)), np.zeros(0, dtype=np.int32))
    ctx.estep(model, empty, stats)
    assert np.all(stats.download() == 0.0)
    assert ctx.score(model, empty).shape == (0,)
    with pytest.raises(G.GhmmError):   # dimension mismatch is an error, not a crash
        ctx.estep(model, ctx.corpus(np.zeros((4, 5)), [4]), stats)
    with pytest.raises(G.GhmmError):
        ctx.fetch(G.BUF_B, (7,))
    for o in (model, corpus, stats, empty):
        o.close()


# ------------------------------------------------- full BASELINE sizes: properties

def test_baseline_config2_properties(G, ctx):
    """39-d, 10x8, 1 000 utterances x 300 frames (BASELINE configs[1]): identities that
    hold at any size — posteriors sum to one, occupancies sum to the frame count, the
    statistics are additive over utterance shards (what the all-reduce relies on),
    and a 1 000-frame sample agrees with the oracle."""
    N, M, D, U, T = 10, 8, 39, 1000, 300
    mean, std = G.synth_truth(N, M, D)
    lens = np.full(U, T, dtype=np.int32)
    X = G.synth_utterances(mean, std, lens)
    hm = G.synth_start_model(mean, std, 0.05)
    model, corpus = ctx.model(hm), ctx.corpus(X, lens)
    stats = ctx.stats(N, M, D)
    ctx.estep(model, corpus, stats)
    full = stats.download()
    s = G.split_stats(full, N, M, D)
    F = U * T
    gamma = ctx.fetch(G.BUF_GAMMA, (F, N))
    post = ctx.fetch(G.BUF_POST, (F, N, M))
    assert np.allclose(gamma.sum(1), 1.0, rtol=0, atol=1e-12)
    assert np.allclose(post.sum(2), 1.0, rtol=0, atol=1e-12)
    assert s["den_c"].sum() == pytest.approx(F, rel=1e-12)
    assert np.allclose(s["num_c"].sum(1), s["den_c"], rtol=1e-12)
    assert np.allclose(s["num_a"].sum(1), s["den_a"], rtol=1e-10)
    assert s["n_utt"] == U
    # shard additivity
    acc = np.zeros_like(full)
    for lo, hi in ((0, 400), (400, 1000)):
        c = ctx.corpus(X[lo * T:hi * T], lens[lo:hi])
        ctx.estep(model, c, stats)
        acc += stats.download()
        c.close()
    assert_close(acc, full, rtol=1e-11, what="shard additivity")
    # oracle on the first 4 utterances
    c = ctx.corpus(X[:4 * T], lens[:4])
    ctx.estep(model, c, stats)
    ref, _ = O.estep(hm, X[:4 * T], lens[:4], dumps=False)
    assert_close(stats.download(), ref, what="sample vs oracle")
    # the decode config (configs[2], scaled to this corpus): every path is a valid
    # left-to-right walk from state 0 to state N-1
    path, score = ctx.viterbi(model, corpus)
    p = path.reshape(U, T)
    assert np.all(p[:, 0] == 0) and np.all(p[:, -1] == N - 1)
    step = np.diff(p, axis=1)
    assert np.all((step == 0) | (step == 1))
    assert np.all(score <= ctx.score(model, corpus) + 1e-9)   # best path <= all paths
    for o in (model, corpus, stats, c):
        o.close()


def test_config4_shape_64_mixtures(G, ctx):
    """10 states x 64 mixtures (BASELINE configs[3] model) on a slice the oracle can replay."""
    hm, X, lens = synth_case(G, 10, 64, 39, [150, 90, 200, 64])
    model, corpus = ctx.model(hm), ctx.corpus(X, lens)
    stats = ctx.stats(10, 64, 39)
    ctx.estep(model, corpus, stats)
    ref, _ = O.estep(hm, X, lens, dumps=False)
    assert_close(stats.download(), ref, what="64-mixture stats")
    for o in (model, corpus, stats):
        o.close()


# ------------------------------------------------------------- two ranks, one GPU

def _rank_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from _load import load_pkg
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = load_pkg()
    G, em = pkg.ghmm, pkg.em
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream(device=0)
    torch.cuda.set_stream(stream)
    hm, X, lens = synth_case(G, 10, 8, 39, [90, 120, 65, 77, 101, 64, 88])
    idx = em.shard_balanced(lens, rank, world)     # length-balanced shards, as bench.py / the C trainer
    off = np.concatenate([[0], np.cumsum(lens)])
    ctx = G.Context(0, stream=stream.cuda_stream)
    model = ctx.model(hm)
    corpus = ctx.corpus(np.concatenate([X[off[u]:off[u + 1]] for u in idx]), lens[idx])
    be = em.HipBackend(G, ctx, model, corpus, torch=torch)
    drv = em.EMDriver(be, dist)
    trace = []
    for _ in range(4):
        drv.step()
        trace.append(be.loglik())
    q.put((rank, [a.copy() for a in model.get().arrays()], trace))
    ctx.close()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_one_rank(G, ctx):
    """The bench's N > 1 path (EMDriver + HipBackend + a torch tensor aliasing the
    statistics, all on one explicit stream) with two processes sharing this GPU and a
    gloo all-reduce standing in for RCCL: same model as one rank on the whole corpus."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    hm, X, lens = synth_case(G, 10, 8, 39, [90, 120, 65, 77, 101, 64, 88])
    model, corpus = ctx.model(hm), ctx.corpus(X, lens)
    stats = ctx.stats(10, 8, 39)
    trace = []
    for _ in range(4):
        ctx.estep(model, corpus, stats)
        trace.append(stats.download()[-2])
        ctx.mstep(model, stats)
    one = model.get().arrays()
    for rank, arrays, tr in res:
        assert_close(tr, trace, rtol=1e-11, what=f"rank {rank} loglik trace")
        for a, b in zip(arrays, one):
            assert_close(a, b, rtol=1e-9, what=f"rank {rank} model")
    for a, b in zip(res[0][1], res[1][1]):
        assert np.array_equal(a, b)   # identical statistics -> identical redundant M-steps
    for o in (model, corpus, stats):
        o.close()


def test_collapsed_components_sharing_a_tile(G, ctx):
    """Variance-floored "needle" components (every variance at 1e-5, det = 1e-195: what EM from the
    reference's initial model produces within two iterations), three of them in ONE tile
    of 16 Gaussians and two more elsewhere, each sitting exactly on a frame of the corpus.  The
    emission kernel takes them through the expanded form and re-evaluates them in the reference's
    direct form only next to their frames; their statistics come from the matrix-core sums, checked
    against the floor.  Everything against the oracle, then two more EM iterations."""
    hm, X, lens = synth_case(G, 10, 8, 39, [300, 211, 128, 77], perturb=0.05)
    # each needle on a frame that its state certainly occupies (else its num_c is 0 and the
    # reference's own M-step divides 0 by 0)
    _, d0 = O.estep(hm, X, lens)
    occ = d0["alpha"] * d0["beta"] / d0["scale"][:, None]
    used = set()
    for (i, j) in ((0, 0), (0, 3), (1, 2), (4, 7), (9, 1)):
        order = np.argsort(-occ[:, i])
        f = next(int(t) for t in order if int(t) not in used)
        used.add(f)
        hm.mean[i, j] = X[f]
        hm.inv_var[i, j] = 1.0 / 1e-5
        hm.det[i, j] = 1e-5 ** 39
    model, corpus = ctx.model(hm), ctx.corpus(X, lens)
    stats = ctx.stats(10, 8, 39)
    F = corpus.frames
    ref_stats, ref = O.estep(hm, X, lens)
    ctx.estep(model, corpus, stats)
    assert np.isfinite(ref["loglik"]).all()
    assert_frames(ctx.fetch(G.BUF_B, (F, 10)), ref["b"], "b")
    assert_frames(ctx.fetch(G.BUF_POST, (F, 80)), ref["post"].reshape(F, -1), "post")
    assert_close(ctx.fetch(G.BUF_LOGLIK, (len(lens),)), ref["loglik"], what="loglik")
    got, refs = G.split_stats(stats.download(), 10, 8, 39), G.split_stats(ref_stats, 10, 8, 39)
    for k in refs:
        assert_close(got[k], refs[k], what="stats." + k)
    cur = hm
    for it in range(3):
        ctx.mstep(model, stats)
        cur = O.mstep(cur, ref_stats)
        for nm, a, b in zip(("A", "c", "mean", "inv_var", "det"), model.get().arrays(), cur.arrays()):
            assert_close(a, b, rtol=1e-7, what=f"iteration {it} model.{nm}")
        ref_stats, _ = O.estep(cur, X, lens, dumps=False)
        if not np.isfinite(ref_stats).all():
            assert it > 0          # parity is defined where the reference stays finite
            break
        ctx.estep(model, corpus, stats)
        assert_close(stats.download(), ref_stats, rtol=1e-7, what=f"iteration {it} statistics")
    # the Viterbi lattice takes the same kernel with log b
    path, score = ctx.viterbi(model, corpus)
    o = 0
    hm_now = model.get()
    for T in lens:
        p_ref, s_ref = O.viterbi(hm_now, X[o:o + T])
        assert np.array_equal(path[o:o + T], p_ref)
        o += T
    for o_ in (model, corpus, stats):
        o_.close()


# ------------------------------------------------ fuzz against the oracle (seeded shapes)

def fuzz_shape(rng, wide):
    """(states, mixtures, coefficients) of one fuzz case.  wide: up to the library's 64 states,
    64 mixtures and 64 coefficients, and half of the cases on the 36..40-coefficient shapes the
    scheduled emission kernel serves."""
    if not wide:
        return int(rng.integers(1, 21)), int(rng.integers(1, 12)), int(rng.integers(1, 45))
    if wide == "huge":   # more states than a wave has lanes (ghmm_wide.hpp), small Gaussians
        return (int(rng.choice([65, 80, 100, 128, 129, 200, 255])), int(rng.choice([1, 2, 3, 4])),
                int(rng.choice([1, 3, 8, 13])))
    N = int(rng.choice([1, 2, 3, 5, 10, 16, 17, 31, 32, 48, 64]))
    M = int(rng.choice([1, 2, 3, 4, 7, 8, 16, 24, 32, 33, 64]))
    D = int(rng.integers(36, 41)) if rng.integers(0, 2) else int(rng.choice([1, 8, 13, 26, 45, 52, 64]))
    return N, M, D


class FuzzSkip(Exception):
    """harsh fuzz case in which the reference itself has left the finite numbers"""


def fuzz_estep_case(G, ctx, seed, wide=False, harsh=False, short=False, subnormal=False):
    """E-step + M-step of the default tier against the oracle on one seeded random shape
    (1-20 states, 1-11 mixtures, 1-44 coefficients, dense or band-diagonal A, band 0..3; wide:
    fuzz_shape's larger shapes).  profiles/fuzz_oracle.py runs the same body over hundreds of
    seeds.  Returns the number of utterances the gamma / xi pass took again in the reference's
    order of operations (GHMM_OPT_REFORDER_COUNT).

    subnormal: the shape is known to hold statistics that are themselves SUBNORMAL numbers
    (1e-316 .. 5e-324: a handful of bits, different in any two implementations, the reference's
    own -O0 and -O2 builds included).  Statistics below 1e-300 are then compared absolutely
    (to 1e-300), and the M-step's quotients of such statistics are not compared."""
    rng = np.random.default_rng((29000 if wide == "huge" else 19000 if wide else 9000) + seed)
    N, M, D = fuzz_shape(rng, wide)
    # every utterance can reach the last state (short: utterances of 1 .. N + 30 frames, some
    # shorter than the model — no path into the last state, gamma = xi = 0 as in the reference)
    lens = [int(x) for x in rng.integers(N, N + 120, size=int(rng.integers(1, 7)))]
    if short:
        lens = [int(x) for x in rng.integers(1, N + 31, size=len(lens) + 2)]
    dense, delta = bool(rng.integers(0, 2)), int(rng.integers(0, 4))
    hm, X, lens = synth_case(G, N, M, D, lens, dense_A=dense, seed=seed,
                             perturb=float(rng.choice([0.6, 1.0] if harsh else [0.02, 0.1, 0.3])))
    if harsh:
        # a model far from its data, with sharpened Gaussians: densities that underflow, subnormal
        # state sums, posteriors split between distant components (profiles/fuzz_oracle.py harsh)
        k = float(rng.choice([1.0, 3.0, 9.0]))
        hm.inv_var *= k
        hm.det /= k ** D
    ref_stats, ref = O.estep(hm, X, lens, delta=delta)
    if harsh and not (np.all(np.isfinite(ref["loglik"])) and np.all(np.isfinite(ref_stats))):
        raise FuzzSkip()   # the reference's own NaN cascade: the documented deviations apply
    ref_nan = bool(np.any(np.isnan(ref_stats)))
    # (short, NaN in the reference: beta^ of a too-short utterance overflowed in the reference's
    # scaling and inf * 0 spread; such an utterance is taken in the reference's own order of
    # operations here as well, and the statistics must be NaN in the same places)
    model, corpus = ctx.model(hm), ctx.corpus(X, lens)
    F = corpus.frames
    ctx.set_option(G.OPT_DELTA, delta)
    try:
        stats = ctx.stats(N, M, D)
        ctx.estep(model, corpus, stats)
        nre = ctx.get_option(G.OPT_REFORDER_COUNT)   # utterances taken again in the reference's order
        tag = (f"seed {seed} N={N} M={M} D={D} lens={list(map(int, lens))} dense={dense} delta={delta} "
               f"reordered={nre}: ")
        assert_close(ctx.fetch(G.BUF_LOGLIK, (len(lens),)), ref["loglik"], what=tag + "loglik")
        assert_frames(ctx.fetch(G.BUF_B, (F, N)), ref["b"], tag + "b")
        # posteriors (TF:1773-1778): against the oracle where the state's density is a normal
        # number; where it is subnormal (its 1/b is beyond the largest double, and the few bits a
        # subnormal density keeps differ between exp implementations) they must still be shares
        post = ctx.fetch(G.BUF_POST, (F, N * M)).reshape(F, N, M)
        normal = (ref["b"] >= 1e-280)[:, :, None]
        assert_frames(np.where(normal, post, 0.0).reshape(F, -1),
                      np.where(normal, ref["post"].reshape(F, N, M), 0.0).reshape(F, -1), tag + "post")
        assert np.all(np.isfinite(post)) and post.min() >= 0.0 and post.max() <= 1.0 + 1e-12, tag + "post range"
        assert_frames(ctx.fetch(G.BUF_ALPHA, (F, N)), ref["alpha"], tag + "alpha")
        assert_frames(ctx.fetch(G.BUF_BETA, (F, N)), ref["beta"], tag + "beta",
                      rtol=subnormal if isinstance(subnormal, float) else RTOL)
        got = stats.download()
        rt = subnormal if isinstance(subnormal, float) else RTOL
        if subnormal:
            tiny = (np.abs(ref_stats) < 1e-300) & (np.abs(got - ref_stats) <= 1e-300)
            got = np.where(tiny, ref_stats, got)
        assert_close(got, ref_stats, rtol=rt, what=tag + "stats")
        ctx.mstep(model, stats)
        new, ref_new = model.get(), O.mstep(hm, ref_stats)
        rs = G.split_stats(ref_stats, N, M, D)
        for nm, a, b in zip(("A", "c", "mean", "inv_var", "det"), new.arrays(), ref_new.arrays()):
            if subnormal:
                # parameters that are quotients of subnormal statistics: not compared
                if nm == "A":
                    weak = (np.abs(rs["num_a"]) < 1e-290) & (rs["num_a"] != 0)
                    weak |= (np.abs(rs["den_a"]) < 1e-290)[:, None]
                else:
                    weak = np.abs(rs["num_c"]) < 1e-290
                    if nm in ("mean", "inv_var"):
                        weak = np.broadcast_to(weak[:, :, None], a.shape)
                a, b = np.where(weak, 0.0, a), np.where(weak, 0.0, b)
            if ref_nan:
                continue   # (quotients of NaN statistics: the E-step's NaN pattern is what is pinned)
            assert_close(a, b, rtol=max(1e-7, 10 * rt), what=tag + "mstep." + nm)
    finally:
        ctx.set_option(G.OPT_DELTA, 1)
        for o in (model, corpus):
            o.close()
    return nre


def fuzz_viterbi_case(G, ctx, seed, wide=False, harsh=False):
    """Viterbi state sequences (bit-identical) and forward scores against the oracle on one
    seeded random shape; profiles/fuzz_viterbi.py runs it over more seeds.  Returns the number
    of utterances checked."""
    rng = np.random.default_rng((27000 if wide == "huge" else 17000 if wide else 7000) + seed)
    N, M, D = fuzz_shape(rng, wide)
    lens = [int(x) for x in rng.integers(1, 150, size=int(rng.integers(1, 6)))]
    dense = bool(rng.integers(0, 2))
    hm, X, lens = synth_case(G, N, M, D, lens, dense_A=dense, seed=seed,
                             perturb=float(rng.choice([0.6, 1.0] if harsh else [0.02, 0.1, 0.3])))
    if harsh:   # as in fuzz_estep_case: sharpened Gaussians far from the data
        k = float(rng.choice([1.0, 3.0, 9.0]))
        hm.inv_var *= k
        hm.det /= k ** D
    model, corpus = ctx.model(hm), ctx.corpus(X, lens)
    try:
        path, score = ctx.viterbi(model, corpus)
        fwd = ctx.score(model, corpus)   # calc_alpha + calc_probability alone (the recogniser's path)
        o = 0
        for u, Tn in enumerate(lens):
            tag = f"seed {seed} utt {u}: N={N} M={M} D={D} T={Tn} dense={dense}"
            p, sc = O.viterbi(hm, X[o:o + Tn])
            assert np.array_equal(path[o:o + Tn], p), tag + ": path differs"
            assert (score[u] == sc) if not np.isfinite(sc) else abs(score[u] - sc) <= 1e-10 * abs(sc), \
                tag + f": score {score[u]!r} vs {sc!r}"
            fs = O.score(hm, X[o:o + Tn])
            assert ((np.isnan(fs) and np.isnan(fwd[u])) or fwd[u] == fs
                    or abs(fwd[u] - fs) <= 1e-10 * abs(fs)), tag + f": forward score {fwd[u]!r} vs {fs!r}"
            o += Tn
    finally:
        model.close()
        corpus.close()
    return len(lens)


@pytest.mark.parametrize("seed", list(range(20)))
def test_fuzz_estep_against_oracle(G, ctx, seed):
    fuzz_estep_case(G, ctx, seed)


@pytest.mark.parametrize("seed", list(range(20)))
def test_fuzz_viterbi_against_oracle(G, ctx, seed):
    fuzz_viterbi_case(G, ctx, seed)


# Models of 65 .. 255 states: one wave per utterance, states strided over its lanes (ghmm_wide.hpp)
@pytest.mark.parametrize("seed", list(range(10)))
def test_fuzz_estep_more_states_than_lanes(G, ctx, seed):
    fuzz_estep_case(G, ctx, seed, wide="huge")


@pytest.mark.parametrize("seed", list(range(4)))
def test_fuzz_estep_more_states_than_lanes_short(G, ctx, seed):
    fuzz_estep_case(G, ctx, seed, wide="huge", short=True)


@pytest.mark.parametrize("seed", list(range(8)))
def test_fuzz_viterbi_more_states_than_lanes(G, ctx, seed):
    fuzz_viterbi_case(G, ctx, seed, wide="huge")


def test_vocabulary_with_a_model_of_more_states_than_lanes(G, ctx):
    """ghmm_score_batch falls back to the word-by-word loop when a model has more than 64 states."""
    hms = []
    X = lens = None
    for k, N in enumerate((5, 70, 12)):
        hm, Xk, lk = synth_case(G, N, 2, 6, [90, 75, 130], seed=40 + k)
        hms.append(hm)
        if k == 1:
            X, lens = Xk, lk
    models = [ctx.model(hm) for hm in hms]
    corpus = ctx.corpus(X, lens)
    try:
        got = ctx.score_batch(models, corpus)
        o = 0
        for u, Tn in enumerate(lens):
            for k, hm in enumerate(hms):
                ref = O.score(hm, X[o:o + Tn])
                assert (np.isnan(ref) and np.isnan(got[k, u])) or got[k, u] == ref or \
                    abs(got[k, u] - ref) <= 1e-10 * abs(ref), (k, u, got[k, u], ref)
            o += Tn
    finally:
        for m_ in models:
            m_.close()
        corpus.close()


def test_more_states_than_the_wide_kernels_take_is_refused(G, ctx):
    hm, X, lens = synth_case(G, 513, 1, 2, [520])
    model, corpus = ctx.model(hm), ctx.corpus(X, lens)
    try:
        with pytest.raises(G.GhmmError) as e:
            ctx.score(model, corpus)
        assert "states" in str(e.value)
    finally:
        model.close()
        corpus.close()


# (433 and 555: single-mixture models whose far states have SUBNORMAL densities — the posterior
# 1/b must not overflow there; found by a 600-seed sweep of profiles/fuzz_oracle.py)
@pytest.mark.parametrize("seed", list(range(12)) + [433, 555])
def test_fuzz_wide_shapes_against_oracle(G, ctx, seed):
    """The same two bodies on shapes up to 64 states x 64 mixtures x 64 coefficients."""
    fuzz_estep_case(G, ctx, seed, wide=True)
    fuzz_viterbi_case(G, ctx, seed, wide=True)


def test_long_utterances_against_oracle(G, ctx):
    """Utterances of thousands of frames (the reference caps them at 500, TF:44; the C ABI does
    not): recursions, statistics and Viterbi paths against the oracle."""
    hm, X, lens = synth_case(G, 10, 8, 39, [5000, 1203, 17, 2500])
    ref_stats, ref = O.estep(hm, X, lens)
    model, corpus = ctx.model(hm), ctx.corpus(X, lens)
    stats = ctx.stats(10, 8, 39)
    ctx.estep(model, corpus, stats)
    F = corpus.frames
    assert_close(ctx.fetch(G.BUF_LOGLIK, (len(lens),)), ref["loglik"], what="loglik")
    assert_frames(ctx.fetch(G.BUF_ALPHA, (F, 10)), ref["alpha"], "alpha")
    assert_frames(ctx.fetch(G.BUF_BETA, (F, 10)), ref["beta"], "beta")
    assert_close(stats.download(), ref_stats, what="stats")
    path, _ = ctx.viterbi(model, corpus)
    o = 0
    for u, Tn in enumerate(lens):
        p, _ = O.viterbi(hm, X[o:o + Tn])
        assert np.array_equal(path[o:o + Tn], p), f"utterance {u}: path differs"
        o += Tn
    for obj in (model, corpus, stats):
        obj.close()


@pytest.mark.parametrize("seed", [68, 253, 437] + list(range(12)))
def test_fuzz_short_utterances_against_oracle(G, ctx, seed):
    """Utterances of 1 .. N + 30 frames, some shorter than the model: no path into the last state.
    Round 2 gave such an utterance gamma = xi = 0 where the reference's beta^ may overflow and
    spread NaN (4 of 300 tier-fuzz shapes, 8 + 22 skipped seeds of the short fuzzer); now it is
    taken in the reference's own order of operations (k_backward_fix, dense inner loop as at
    TF:1493-1510), and beta^, gamma and the statistics are NaN exactly where the reference's are:
    nothing is skipped any more.  68 / 253 / 437: beta^ ~ 1e208 .. 1e304 in the reference."""
    fuzz_estep_case(G, ctx, seed, short=True)


def test_short_utterances_follow_the_reference_into_nan(G, ctx):
    """At least one seed of the short fuzzer whose reference statistics ARE NaN (beta^ of a
    too-short utterance overflows in the reference's scaling, inf * 0): the default tier must
    reproduce the pattern, and must have taken utterances again to do so."""
    hit = 0
    for seed in range(120):
        rng = np.random.default_rng(9000 + seed)
        N, M, D = fuzz_shape(rng, False)
        n = int(rng.integers(1, 7))
        lens = [int(x) for x in rng.integers(N, N + 120, size=n)]
        lens = [int(x) for x in rng.integers(1, N + 31, size=len(lens) + 2)]
        dense, delta = bool(rng.integers(0, 2)), int(rng.integers(0, 4))
        hm, X, lens = synth_case(G, N, M, D, lens, dense_A=dense, seed=seed,
                                 perturb=float(rng.choice([0.02, 0.1, 0.3])))
        ref_stats, _ = O.estep(hm, X, lens, delta=delta, dumps=False)
        if not np.any(np.isnan(ref_stats)):
            continue
        assert fuzz_estep_case(G, ctx, seed, short=True) > 0
        hit += 1
        if hit == 3:
            break
    assert hit >= 1


# (seed, wide, subnormal): every harsh-fuzz shape that disagreed with the reference in round 2
# or in round 3's sweeps (profiles/r3_fuzz.txt), none selected to pass.
#   subnormal = True: the shape holds statistics that are themselves subnormal numbers (1e-316 ..
#   5e-324 — a Gaussian or a transition whose WHOLE occupancy is a handful of bits): compared to
#   1e-300 absolutely, their M-step quotients not at all (noise in the reference as well);
#   subnormal = 1e-6 (seed 561): the reference's own beta^_53(15) is a subnormal 1e-310 and the
#   value it feeds, beta^_52(15) = 4.15e-285, the largest of its frame, carries that rounding
#   (2e-7) into a statistics vector whose largest entry is 1e-288; checked to 1e-6 (bar: 1e-5).
HARSH_SEEDS = [(12, False, False), (37, True, False), (1, False, False), (5, False, False), (6, False, False),
               (7, False, False), (0, True, False), (1, True, False), (5, True, False), (6, True, False),
               # statistics that are subnormal numbers
               (21, False, True), (180, False, True), (284, False, True), (390, False, True),
               (2, True, True), (8, True, True), (140, True, True), (167, True, True), (170, True, True),
               (245, True, True), (276, True, True), (350, True, True), (390, True, True), (397, True, True),
               (561, False, 1e-6),
               # utterances taken again in the reference's order (forward and backward mass more
               # than 200 decades apart, or D_t underflown): round 2's class (3)
               (8, False, False), (10, False, False), (66, False, False), (74, False, False),
               (322, False, True), (367, False, True), (368, False, True), (575, False, True),
               (40, True, False), (48, True, False), (268, True, True), (371, True, True), (380, True, True)]


@pytest.mark.parametrize("seed,wide,subnormal", HARSH_SEEDS)
def test_fuzz_harsh_models_against_oracle(G, ctx, seed, wide, subnormal):
    """Models far from their data with sharpened Gaussians (densities near the underflow, subnormal
    state sums).  12 and wide 37: a first frame whose densities are ~1e-294 — 1/b there must be
    as exact as anywhere (the raw hardware reciprocal is good to 1e-8 only).  Cases in which the
    reference itself leaves the finite numbers are not comparable (skipped).  Round 2's third class
    of disagreement — a backward recursion normalised to a row sum of 1 lost components more than
    308 decades below the row's largest, which the reference's beta^ keeps — is closed: rows are
    scaled to 2^680 by exact powers of two, and an utterance whose rho_t exceeds that is taken
    again in the reference's order (ghmm_pair.hpp, RANGE)."""
    try:
        fuzz_estep_case(G, ctx, seed, wide=wide, harsh=True, subnormal=subnormal)
    except FuzzSkip:
        pytest.skip("the reference's own statistics are not finite for this seed")


def test_class2_gaussians_both_exact_paths(G, ctx):
    """Gaussians that the expanded sums cannot carry although their variances are not at the
    floor (stats_class 2: one coefficient with sigma = 0.01 sitting on a frame its state
    occupies): their statistics come from the vector-ALU kernel when the host launches it
    (GHMM_OPT_VEC_STATS 1, and what auto does right after ghmm_model_set) and from the exact
    recomputation inside k_reduce_all when it does not (2: what auto does when such a Gaussian
    appears in a model that had none for a few iterations).  Both against the oracle, then EM
    iterations in auto mode against the oracle's."""
    hm, X, lens = synth_case(G, 10, 8, 39, [300, 211, 128, 77], perturb=0.05)
    _, d0 = O.estep(hm, X, lens)
    occ = d0["alpha"] * d0["beta"] / d0["scale"][:, None]
    for (i, j, d) in ((4, 3, 7), (0, 0, 38), (9, 7, 0)):
        f = int(np.argmax(occ[:, i] * d0["post"].reshape(-1, 10, 8)[:, i, j]))
        old_var = 1.0 / hm.inv_var[i, j, d]
        hm.mean[i, j, d] = X[f, d]
        hm.inv_var[i, j, d] = 1.0 / 1e-4
        hm.det[i, j] *= 1e-4 / old_var
    ref, _ = O.estep(hm, X, lens, dumps=False)
    corpus = ctx.corpus(X, lens)
    stats = ctx.stats(10, 8, 39)
    try:
        for mode in (1, 2, 0):
            ctx.set_option(G.OPT_VEC_STATS, mode)
            model = ctx.model(hm)
            ctx.estep(model, corpus, stats)
            assert_close(stats.download(), ref, what=f"class-2 statistics, GHMM_OPT_VEC_STATS {mode}")
            if mode == 0:
                cur = hm
                for it in range(6):   # past the window in which the kernel is launched unconditionally
                    ctx.estep(model, corpus, stats)
                    r, _ = O.estep(cur, X, lens, dumps=False)
                    assert_close(stats.download(), r, what=f"iteration {it}")
                    ctx.mstep(model, stats)
                    cur = O.mstep(cur, r)
            model.close()
    finally:
        ctx.set_option(G.OPT_VEC_STATS, 0)
        corpus.close(); stats.close()


@pytest.mark.parametrize("shape", [
    (10, 8, 39, [300, 211, 128, 77, 1, 5, 0, 300]),     # 16-lane groups, T = 1, T < N, an empty utterance
    (20, 2, 13, [90, 45, 19, 300, 7]),                  # 32-lane groups
    (40, 1, 5, [120, 39, 200]),                         # 64-lane groups
    (3, 2, 5, [17] * 70),                               # more utterances than one block's groups
])
def test_one_launch_recursions_equal_the_separate_launches(G, ctx, shape):
    """k_scan_combine (both scans + the gamma / xi pass in one launch, what ghmm_estep uses on a
    band-diagonal A) against k_scan_pair + k_combine (GHMM_OPT_FUSED_SCAN = 2): the same
    operations per frame, so gamma, alpha^ and beta^ (on demand after either) bit for bit; the
    statistics and log P are sums over an utterance's chunks, of which the one launch makes fewer
    when the utterances are short."""
    N, M, D, lens = shape
    hm, X, lens = synth_case(G, N, M, D, lens, seed=N)
    corpus = ctx.corpus(X, lens)
    F = corpus.frames
    out = {}
    try:
        for mode in (2, 0):
            ctx.set_option(G.OPT_FUSED_SCAN, mode)
            model, stats = ctx.model(hm), ctx.stats(N, M, D)
            ctx.estep(model, corpus, stats)
            out[mode] = (stats.download(), ctx.fetch(G.BUF_GAMMA, (F, N)), ctx.fetch(G.BUF_ALPHA, (F, N)),
                         ctx.fetch(G.BUF_LOGLIK, (len(lens),)), ctx.fetch(G.BUF_BETA, (F, N)))
            model.close(); stats.close()
        for a, b, nm in zip(out[2], out[0], ("statistics", "gamma", "alpha", "loglik", "beta")):
            if nm in ("statistics", "loglik"):   # sums over an utterance's chunks: 8 there, 2 .. 8 here
                assert np.array_equal(np.isnan(a), np.isnan(b)), nm
                assert_close(b, a, rtol=1e-12, what=nm)
            else:
                assert np.array_equal(a, b, equal_nan=True), nm
        ref, _ = O.estep(hm, X, lens, dumps=False)
        assert_close(out[0][0], ref, what="statistics against the oracle")
    finally:
        ctx.set_option(G.OPT_FUSED_SCAN, 0)
        corpus.close()


def test_no_utterance_of_a_fitting_model_is_taken_again(G, ctx):
    """On data the model fits (the benchmark's generator, ragged lengths) the gamma / xi pass
    must not hand anything to the reference-order kernel: the counter stays 0."""
    hm, X, lens = synth_case(G, 10, 8, 39, [300, 211, 128, 77, 500, 100, 345, 64], perturb=0.05)
    model, corpus = ctx.model(hm), ctx.corpus(X, lens)
    stats = ctx.stats(10, 8, 39)
    for _ in range(3):
        ctx.estep(model, corpus, stats)
        assert ctx.get_option(G.OPT_REFORDER_COUNT) == 0
        ctx.mstep(model, stats)
    for o in (model, corpus, stats):
        o.close()


# ------------------------------------------------------------------- boundary

def test_two_contexts_two_host_threads(G, ctx):
    """SURVEY §8(b) threading row: one context per host thread, no global mutable state.  Two
    contexts (own streams; on a one-GPU box both on device 0) run E-steps concurrently from
    two threads — kernels that need the raised LDS limit included — and each must match the
    oracle; the contexts are created on the threads that use them."""
    import threading
    cases = [synth_case(G, 10, 8, 39, [120, 77, 64, 90]), synth_case(G, 7, 3, 39, [100, 61, 16], first=9)]
    refs = [O.estep(hm, X, lens, dumps=False)[0] for hm, X, lens in cases]
    out, errs = [None, None], []

    def work(k):
        try:
            hm, X, lens = cases[k]
            c = G.Context(0)
            try:
                model, corpus = c.model(hm), c.corpus(X, lens)
                stats = c.stats(hm.N, hm.M, hm.D)
                for _ in range(5):
                    c.estep(model, corpus, stats)
                out[k] = stats.download()
                path, _ = c.viterbi(model, corpus)
                assert path.shape == (int(np.sum(lens)),)
            finally:
                c.close()
        except Exception as e:  # noqa: BLE001
            errs.append((k, repr(e)))

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for k in range(2):
        assert_close(out[k], refs[k], what=f"thread {k} statistics")


def test_stale_workspace_is_refused(G, ctx):
    """beta^ on demand is rebuilt from the E-step's alpha^ / W: once another call has rewritten
    the workspace (a score of another corpus) ghmm_fetch must refuse, not return garbage;
    ghmm_forward must refuse the emission densities of another model of the same shape."""
    hm, X, lens = synth_case(G, 5, 2, 6, [40, 25])
    hm2, X2, lens2 = synth_case(G, 5, 2, 6, [40, 25], perturb=0.2, first=4)
    model, corpus, other = ctx.model(hm), ctx.corpus(X, lens), ctx.corpus(X2, lens2)
    model2 = ctx.model(hm2)
    stats = ctx.stats(5, 2, 6)
    ctx.estep(model, corpus, stats)
    ctx.score(model, other)
    with pytest.raises(G.GhmmError):
        ctx.fetch(G.BUF_BETA, (corpus.frames, 5))
    ctx.emission(model, corpus, True)
    with pytest.raises(G.GhmmError):
        ctx.forward(model2, corpus)        # b belongs to `model`
    with pytest.raises(G.GhmmError):
        ctx.forward(model, other)          # ... and to `corpus`
    ctx.forward(model, corpus)
    for o in (model, model2, corpus, other, stats):
        o.close()


def test_mstep_reads_the_transition_band_only(G, ctx):
    """ghmm_mstep takes num_a inside i <= j <= i + delta (the only entries TF:1601 ever
    accumulates); whatever else an uploaded vector holds there is ignored, so a band-diagonal
    A stays band-diagonal and the band-only combine pass stays valid."""
    hm, X, lens = synth_case(G, 6, 2, 5, [50, 41, 33])
    model, corpus = ctx.model(hm), ctx.corpus(X, lens)
    stats = ctx.stats(6, 2, 5)
    ctx.estep(model, corpus, stats)
    v = stats.download()
    ref_new = O.mstep(hm, v)
    dirty = v.copy()
    na = dirty[:36].reshape(6, 6)
    na[np.tril_indices(6, -1)] = 0.37       # below the diagonal
    na[np.triu_indices(6, 2)] = 0.11        # beyond i + 1
    stats.upload(dirty)
    ctx.mstep(model, stats)
    new = model.get()
    for nm, a, b in zip(("A", "c", "mean", "inv_var", "det"), new.arrays(), ref_new.arrays()):
        assert_close(a, b, rtol=1e-12, floor=0.0, what="mstep." + nm)
    # and the next E-step (band-only combine) agrees with the oracle on that model
    ctx.estep(model, corpus, stats)
    ref2, _ = O.estep(ref_new, X, lens, dumps=False)
    assert_close(stats.download(), ref2, what="E-step after the masked M-step")
    for o in (model, corpus, stats):
        o.close()


def test_length_order_and_balanced_shards(G, ctx):
    """Utterances are packed four to a wave in length order (longest first): results must not
    depend on it.  A corpus and its shuffled copy give the same per-utterance log P and the
    same statistics; the length-balanced shards of em.shard_balanced add up to the whole."""
    em = __import__("ghmm_amd").em
    rng = np.random.default_rng(5)
    lens = rng.integers(100, 501, size=37).astype(np.int32)    # T ~ U[100, 500]
    hm, X, lens = synth_case(G, 10, 8, 39, lens)
    off = np.concatenate([[0], np.cumsum(lens)])
    model, corpus = ctx.model(hm), ctx.corpus(X, lens)
    stats = ctx.stats(10, 8, 39)
    ctx.estep(model, corpus, stats)
    full = stats.download()
    ll = ctx.fetch(G.BUF_LOGLIK, (len(lens),))
    ref, dump = O.estep(hm, X, lens)
    assert_close(full, ref, what="ragged corpus vs oracle")
    assert_close(ll, dump["loglik"], what="ragged loglik")
    perm = rng.permutation(len(lens))
    Xp = np.concatenate([X[off[u]:off[u + 1]] for u in perm])
    cp = ctx.corpus(Xp, lens[perm])
    ctx.estep(model, cp, stats)
    assert_close(stats.download(), full, rtol=1e-11, what="shuffled corpus")
    assert_close(ctx.fetch(G.BUF_LOGLIK, (len(lens),)), ll[perm], rtol=1e-13, what="shuffled loglik")
    path, score = ctx.viterbi(model, cp)
    path0, score0 = ctx.viterbi(model, corpus)
    assert np.array_equal(score, score0[perm])
    offp = np.concatenate([[0], np.cumsum(lens[perm])])
    for k, u in enumerate(perm):
        assert np.array_equal(path[offp[k]:offp[k + 1]], path0[off[u]:off[u + 1]])
    acc = np.zeros_like(full)
    frames = []
    for r in range(3):
        idx = em.shard_balanced(lens, r, 3)
        assert list(idx) == list(G.shard_balanced(lens, r, 3))
        c = ctx.corpus(np.concatenate([X[off[u]:off[u + 1]] for u in idx]), lens[idx])
        ctx.estep(model, c, stats)
        acc += stats.download()
        frames.append(int(lens[idx].sum()))
        c.close()
    assert max(frames) - min(frames) <= lens.max()
    assert_close(acc, full, rtol=1e-11, what="balanced shards add up")
    for o in (model, corpus, cp, stats):
        o.close()


# ------------------------------------------------------- the C-side collective (RCCL)

def test_rccl_allreduce_one_rank_communicator(G, ctx, tmp_path):
    """ghmm_comm_* / ghmm_stats_allreduce on a real RCCL communicator of ONE rank (all a
    one-GPU box can hold: RCCL refuses two ranks on one device): the sum over ranks of a vector
    is the vector; ghmm_model_init_comm over it equals ghmm_model_init; the id-through-a-file
    rendezvous works and removes its file."""
    hm, X, lens = synth_case(G, 10, 8, 39, [90, 120, 65, 77])
    model, corpus = ctx.model(hm), ctx.corpus(X, lens)
    stats = ctx.stats(10, 8, 39)
    comm = ctx.comm(0, 1)
    assert (comm.rank, comm.size) == (0, 1)
    ctx.estep(model, corpus, stats)
    before = stats.download()
    ctx.stats_allreduce(stats, comm)
    assert np.array_equal(stats.download(), before)
    ctx.mstep(model, stats)                       # E-step -> all-reduce -> M-step on one stream
    ref_new = O.mstep(hm, O.estep(hm, X, lens, dumps=False)[0])
    for nm, a, b in zip(("A", "c", "mean", "inv_var", "det"), model.get().arrays(), ref_new.arrays()):
        assert_close(a, b, rtol=1e-7, what="mstep after all-reduce: " + nm)
    a = model.init_from(corpus)
    b = model.init_from(corpus, comm)
    for x, y in zip(a.arrays(), b.arrays()):
        assert np.array_equal(x, y)
    comm.close()
    idfile = os.path.join(str(tmp_path), "comm.id")
    c2 = ctx.comm(0, 1, path=idfile)
    assert not os.path.exists(idfile)
    ctx.stats_allreduce(stats, c2)
    c2.close()
    for o in (model, corpus, stats):
        o.close()


def test_train_command_line_multi_rank_mode(G, whole, tmp_path):
    """The C trainer's rank / world mode (GHMM_WORLD, GHMM_RANK, GHMM_COMM_ID, GHMM_DEVICE) over a
    one-rank RCCL communicator: same report and model as the plain run."""
    exe = os.path.join(PKG_DIR, "bin", "hmm-continuous-train-fs")
    files = [os.path.join(GOLDEN, "perfil", fn) for fn in whole["mean_list"]]
    lst = _write_lists(str(tmp_path), files, "parameters.txt")
    outs = []
    for tag, env in (("plain", {}), ("comm", {"GHMM_WORLD": "1", "GHMM_RANK": "0", "GHMM_DEVICE": "0",
                                              "GHMM_COMM_ID": os.path.join(str(tmp_path), "id")})):
        out = os.path.join(str(tmp_path), tag + "_all13.hmm")
        p = subprocess.run([exe, "all13", "6", "1", "3", lst, out], stdout=subprocess.PIPE,
                           env={**os.environ, **env})
        assert p.returncode == 0, p.stdout.decode()
        rep = [l for l in open(out[:-4] + ".txt").read().split("\n")
               if not l.startswith(("model file", "starting time", "ending time", "cpu time"))]
        outs.append((rep, G.HostModel.read(out)))
    assert outs[0][0] == outs[1][0]
    exp = whole["train_all13_m3"]
    assert f"number of iterations: {exp['iterations']} " in outs[1][0]
    assert "number of exemplars in training sequence: 13 " in outs[1][0]
    for a, b in zip(outs[0][1].arrays(), outs[1][1].arrays()):
        assert np.array_equal(a, b)
    # a rank layout without a rendezvous path is refused
    p = subprocess.run([exe, "all13", "6", "1", "3", lst, os.path.join(str(tmp_path), "x.hmm")],
                       stdout=subprocess.PIPE, env={**os.environ, "GHMM_WORLD": "2", "GHMM_RANK": "1"})
    assert p.returncode == 1 and b"GHMM_COMM_ID" in p.stdout


# ------------------------- BASELINE configs[4]: 2 000 tied states x 16 mixtures, 1 M frames

def _config5_model(G, N=2000, M=16, D=39, seed=11):
    """A triphone-scale codebook: N x M Gaussians around N state centres (numpy, seeded)."""
    rng = np.random.default_rng(seed)
    centre = rng.normal(0.0, 2.0, size=(N, 1, D))
    mean = centre + rng.normal(0.0, 0.7, size=(N, M, D))
    var = rng.uniform(0.5, 1.5, size=(N, M, D)) ** 2
    c = rng.uniform(0.5, 1.5, size=(N, M))
    c /= c.sum(1, keepdims=True)
    A = np.zeros((N, N))
    return G.HostModel(A, c, mean, 1.0 / var, var.prod(axis=2)), centre[:, 0, :]


def _config5_frames(centre, F, seed, chunk=1 << 16):
    """Frames near randomly chosen state centres, generated in chunks (1 M x 39 doubles)."""
    rng = np.random.default_rng(seed)
    X = np.empty((F, centre.shape[1]))
    for lo in range(0, F, chunk):
        n = min(chunk, F - lo)
        X[lo:lo + n] = centre[rng.integers(0, centre.shape[0], n)] + rng.normal(0.0, 1.2, (n, centre.shape[1]))
    return X


def test_config5_emission_slice_against_oracle(G, ctx):
    """calc_symbol_probab / calc_gaus (TF:1749-1841) at N = 2 000, M = 16, D = 39 (emission
    only: the reference's stack arrays cannot hold this model as a program):
    b of a 96-frame slice (one ragged tile) against the oracle's orc_emission, entry by entry
    relative to each frame's largest density."""
    hm, centre = _config5_model(G)
    X = _config5_frames(centre, 96 + 7, seed=1)
    lens = np.array([len(X)], dtype=np.int32)
    model, corpus = ctx.model(hm), ctx.corpus(X, lens)
    ctx.emission(model, corpus, False)
    got = ctx.fetch(G.BUF_B, (len(X), hm.N))
    ref = O.emission(hm, X)
    assert np.isfinite(ref).all() and (ref.max(axis=1) > 0).all()
    assert_frames(got, ref, "config 5 b")
    for o in (model, corpus):
        o.close()


def test_config5_full_size_properties(G, ctx):
    """The full configuration: 1 M frames x 2 000 states (b = 16 GB in HBM).  Properties that
    hold at any size — every b finite and >= 0, every frame with a positive density — checked
    over the whole output in blocks, and blocks of frames equal to what a SEPARATE launch over
    just those frames computes (which the slice test ties to the oracle)."""
    F, N = 1_000_000, 2000
    hm, centre = _config5_model(G)
    X = _config5_frames(centre, F, seed=2)
    model = ctx.model(hm)
    corpus = ctx.corpus(X, np.array([F], dtype=np.int32))
    ctx.emission(model, corpus, False)
    keep = {}
    blk = 50_000
    mins, maxs = [], []
    for lo in range(0, F, blk):
        b = ctx.fetch_range(G.BUF_B, lo * N, (blk, N))
        assert np.isfinite(b).all() and (b >= 0.0).all(), f"frames {lo}..{lo + blk}"
        rowmax = b.max(axis=1)
        assert (rowmax > 0.0).all(), f"a frame without any density in {lo}..{lo + blk}"
        mins.append(rowmax.min()); maxs.append(rowmax.max())
        for f0 in (0, 499_904, 999_904):      # first tile, a middle tile, the last 96 frames
            if lo <= f0 < lo + blk:
                keep[f0] = b[f0 - lo:f0 - lo + 96].copy()
    corpus.close()
    for f0, bfull in keep.items():
        c = ctx.corpus(X[f0:f0 + 96], np.array([96], dtype=np.int32))
        ctx.emission(model, c, False)
        assert np.array_equal(ctx.fetch(G.BUF_B, (96, N)), bfull), f"frames {f0}.. differ between launches"
        ref = O.emission(hm, X[f0:f0 + 96])
        assert_frames(bfull, ref, f"config 5 full run, frames {f0}..")
        c.close()
    model.close()


# ------------------------------------------------ several feature streams (param_number P > 1)

@pytest.fixture(scope="module")
def streams():
    return json.load(open(os.path.join(GOLDEN, "streams_p2.json")))


def _stream_data(G, streams, idx):
    from streams_util import second_stream
    Xs = [G.perfil_read(os.path.join(GOLDEN, "perfil", streams["mean_list"][k])) for k in idx]
    lens = np.array([len(x) for x in Xs], dtype=np.int32)
    return [np.concatenate(Xs), np.concatenate([second_stream(x, streams["D2"]) for x in Xs])], lens


def _golden_streams(G, rec, word=""):
    m = rec["model"]
    return [G.HostModel(m["A"], s["c"], s["mean"], s["inv_var"], s["det"], word=word) for s in m["streams"]]


def _estep_streams_vs_oracle(G, ctx, hms, Xs, lens, tag):
    models = [ctx.model(h) for h in hms]
    corpora = [ctx.corpus(x, lens) for x in Xs]
    stats = [ctx.stats(h.N, h.M, h.D) for h in hms]
    ctx.estep_streams(models, corpora, stats)
    ref_stats, ref_b, ref_ll = O.estep_streams(hms, Xs, lens)
    F, N = int(np.sum(lens)), hms[0].N
    assert_frames(ctx.fetch(G.BUF_B, (F, N)), ref_b, tag + " product b")
    assert_close(ctx.fetch(G.BUF_LOGLIK, (len(lens),)), ref_ll, what=tag + " loglik")
    for p, (s, r) in enumerate(zip(stats, ref_stats)):
        got, ref = G.split_stats(s.download(), N, hms[p].M, hms[p].D), G.split_stats(r, N, hms[p].M, hms[p].D)
        for k in ref:
            assert_close(got[k], ref[k], what=f"{tag} stream {p} stats.{k}")
    # M-step per stream (TF:332-346): every stream's call writes the same A
    for p in range(len(hms)):
        ctx.mstep(models[p], stats[p])
        new, ref_new = models[p].get(), O.mstep(hms[p], ref_stats[p])
        for nm, a, b in zip(("A", "c", "mean", "inv_var", "det"), new.arrays(), ref_new.arrays()):
            assert_close(a, b, rtol=1e-7, what=f"{tag} stream {p} mstep.{nm}")
    assert_close(ctx.score_streams(models, corpora),
                 [O.score_streams([m.get() for m in models], [x[o:o + t] for x in Xs])
                  for o, t in zip(np.concatenate([[0], np.cumsum(lens)[:-1]]), lens)],
                 rtol=1e-9, what=tag + " score after the M-step")
    for o in models + corpora + stats:
        o.close()


def test_streams_estep_against_oracle(G, ctx, streams):
    """Two feature streams on the bundled utterances (9-d + 5-d, 3 + 2 mixtures: the vector-ALU
    and generic kernels) and on synthetic 39-d + 13-d streams with 8 + 4 mixtures (the matrix-core
    kernels): product b, log P, every stream's statistics and M-step against the oracle, which
    reproduces the real two-stream trainer bit for bit (tests/test_oracle.py)."""
    Xs, lens = _stream_data(G, streams, range(13))
    _estep_streams_vs_oracle(G, ctx, _golden_streams(G, streams["train_all13_p2"]), Xs, lens, "bundled")
    hm1, X1, lens = synth_case(G, 10, 8, 39, [120, 77, 64, 90, 33])
    hm2, X2, _ = synth_case(G, 10, 4, 13, [120, 77, 64, 90, 33], first=50)
    hm2.A[:] = hm1.A
    _estep_streams_vs_oracle(G, ctx, [hm1, hm2], [X1, X2], lens, "synthetic")


def test_streams_training_matches_reference_program(G, ctx, streams):
    """The EM loop over two streams on the GPU from the reference-identical initial models: the
    real trainer's iteration count, printed mean log-likelihood and written model."""
    exp = streams["train_all13_p2"]
    Xs, lens = _stream_data(G, streams, range(13))
    hms = [G.HostModel.init_from(Xs[p], lens, 6, exp["model"]["M"][p]) for p in range(2)]
    models = [ctx.model(h) for h in hms]
    corpora = [ctx.corpus(x, lens) for x in Xs]
    stats = [ctx.stats(h.N, h.M, h.D) for h in hms]
    old, it = 1.0, 0
    while True:
        it += 1
        ctx.estep_streams(models, corpora, stats)
        p = float(stats[0].download()[-2])
        var = abs((old - p) / old)
        if not var > 1e-3 or it > 100:
            break
        old = p
        for m, s in zip(models, stats):
            ctx.mstep(m, s)
    assert it == exp["iterations"]
    assert p / len(lens) == pytest.approx(exp["mean_probability"], rel=1e-9, abs=1e-6)
    for got, ref in zip(models, _golden_streams(G, exp)):
        for nm, a, b in zip(("A", "c", "mean", "inv_var", "det"), got.get().arrays(), ref.arrays()):
            assert_close(a, b, rtol=1e-6, what="all13 P=2 " + nm)
    for o in models + corpora + stats:
        o.close()


def test_streams_command_lines(G, streams, tmp_path):
    """Both executables with param_number = 2 (argv as the reference takes it: the mixtures, then
    the lists, stream by stream): the trainer's report and model against the real trainer's, the
    recogniser's report line for line against the real recogniser's."""
    from streams_util import second_stream
    tmp = str(tmp_path)
    s1, s2 = [], []
    for fn in streams["mean_list"]:
        X = G.perfil_read(os.path.join(GOLDEN, "perfil", fn))
        p2 = os.path.join(tmp, "d_" + fn)
        G.perfil_write(p2, second_stream(X, streams["D2"]))
        s1.append(os.path.join(GOLDEN, "perfil", fn))
        s2.append(p2)
    l1, l2 = _write_lists(tmp, s1, "list1.txt"), _write_lists(tmp, s2, "list2.txt")
    exe = os.path.join(PKG_DIR, "bin", "hmm-continuous-train-fs")
    out = os.path.join(tmp, "all13p2.hmm")
    p = subprocess.run([exe, "all13p2", "6", "2", "3", "2", l1, l2, out], stdout=subprocess.PIPE,
                       env={**os.environ, "GHMM_HOST_INIT": "1"})
    assert p.returncode == 0, p.stdout.decode()
    exp = streams["train_all13_p2"]
    rep = open(out[:-4] + ".txt").read().split("\n")
    assert rep[4] == "number of parameters: 2 "
    assert rep[5:9] == ["number of mixtures 1: 3 ", "number of mixtures 2: 2 ", f"parameter 1: {l1} ",
                        f"parameter 2: {l2} "]
    assert rep[11] == f"mean probability: {exp['mean_probability']:f} "
    assert rep[12] == f"number of iterations: {exp['iterations']} "
    got = G.HostModel.read_streams(out)
    assert [h.word for h in got] == ["all13p2"] * 2
    for g, r in zip(got, _golden_streams(G, exp)):
        for nm, a, b in zip(("A", "c", "mean", "inv_var", "det"), g.arrays(), r.arrays()):
            assert_close(a, b, rtol=1e-6, what="trainer P=2 " + nm)
    # the recogniser on the real trainer's 13 two-stream word models
    paths = []
    for w in streams["words"]:
        paths.append(os.path.join(tmp, w + ".hmm"))
        G.HostModel.write_streams(paths[-1], _golden_streams(G, streams["train13_p2"][w], word=w))
    ml = _write_lists(tmp, paths, "models.txt")
    wl = _write_lists(tmp, streams["words"], "words.txt")
    exe = os.path.join(PKG_DIR, "bin", "recognition-continuous-test-fs")
    rout = os.path.join(tmp, "report.txt")
    p = subprocess.run([exe, "1", ml, "1", l1, l2, wl, rout], stdout=subprocess.PIPE)
    assert p.returncode == 0, p.stdout.decode()
    got = [l for l in open(rout).read().split("\n")
           if not l.startswith("Date and time") and "recognition time" not in l
           and not l.startswith("Model name")]
    assert got == streams["recog13_p2"]["report"]


# ------------------------------------ the scheduled emission kernel, every instantiation

@pytest.mark.parametrize("M,D", [(1, 39), (2, 38), (3, 37), (4, 36), (6, 39), (8, 36), (12, 38), (16, 37),
                                 (24, 39), (32, 36), (48, 38), (64, 37)])
def test_emission_kernel_instantiations(G, ctx, M, D):
    """k_emission_sched<20, MP, OUT> for every mixture padding MP = 1..64 (M padded up: 3 -> 4, 6 -> 8,
    12 -> 16, 24 -> 32, 48 -> 64) and every coefficient count it serves (36..39), in its three
    output modes: b + posteriors (E-step), b (score) and log b (Viterbi), on a corpus whose last
    frame tile is ragged, against the oracle."""
    N = 5 if M >= 32 else 7
    hm, X, lens = synth_case(G, N, M, D, [77, 130, 41], perturb=0.1)
    ref_stats, ref = O.estep(hm, X, lens)
    model, corpus = ctx.model(hm), ctx.corpus(X, lens)
    stats = ctx.stats(N, M, D)
    ctx.estep(model, corpus, stats)
    F = corpus.frames
    tag = f"M={M} D={D}: "
    assert_frames(ctx.fetch(G.BUF_B, (F, N)), ref["b"], tag + "b")
    assert_frames(ctx.fetch(G.BUF_POST, (F, N * M)), ref["post"].reshape(F, -1), tag + "post")
    assert_close(stats.download(), ref_stats, what=tag + "stats")
    assert_close(ctx.score(model, corpus), ref["loglik"], what=tag + "score")
    path, score = ctx.viterbi(model, corpus)
    o = 0
    for T in lens:
        p_ref, s_ref = O.viterbi(hm, X[o:o + T])
        assert np.array_equal(path[o:o + T], p_ref), tag + "Viterbi path"
        o += T
    for o_ in (model, corpus, stats):
        o_.close()


def test_emission_frames_on_an_odd_8_byte_boundary(G, ctx):
    """ghmm_corpus_wrap of a device pointer that is 8- but not 16-byte aligned: the emission
    kernel's 8-byte tile loads and the statistics kernel's unstaged operands take over."""
    import torch
    hm, X, lens = synth_case(G, 10, 8, 39, [120, 77, 64])
    buf = torch.zeros(X.size + 1, dtype=torch.float64, device="cuda:0")
    buf[1:] = torch.from_numpy(X.ravel()).to("cuda:0")
    torch.cuda.synchronize()
    assert (buf.data_ptr() + 8) % 16 == 8
    model = ctx.model(hm)
    corpus = ctx.corpus_from_device(buf.data_ptr() + 8, lens, 39)
    stats = ctx.stats(10, 8, 39)
    ctx.estep(model, corpus, stats)
    ref, _ = O.estep(hm, X, lens, dumps=False)
    assert_close(stats.download(), ref, what="statistics from unaligned frames")
    for o_ in (model, corpus, stats):
        o_.close()


# ------------------------------------------------- bench.py --gpus 2, self-launched, one GPU

@pytest.mark.gpu
def test_bench_self_launch_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2 ...` exactly as the driver types it, no launcher environment:
    bench.py starts its two ranks itself (launch.py), here both on device 0 with gloo standing in
    for RCCL (a one-GPU box cannot hold a two-rank RCCL communicator), and rank 0 prints the one
    JSON line with n_gpus = ranks = 2, the all-reduce timed, and BASELINE configs[3]'s shape
    (64 mixtures, fixed 10 iterations, scaled down to 40 utterances per rank) as extras.config4."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(GHMM_FORCE_DEVICE="0", GHMM_DIST_BACKEND="gloo")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3",
                        "--warmup", "1", "--spinup", "0", "--utts", "64", "--frames", "120",
                        "--config4", "on", "--config4-utts", "40"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks"] == 2 and rec["dist_backend"] == "gloo"
    assert rec["steps"] == 3 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    assert rec["config"]["frames_per_step"] == 2 * 64 * 120
    assert rec["value"] > 0 and rec["value_cold"] > 0 and rec["allreduce_ms"] > 0
    c4 = rec["extras"]["config4"]
    assert c4["iterations"] == 10 and c4["checks_ok"] is True
    assert np.isfinite(rec["loglik_per_frame"])


def test_more_models_than_flag_slots(G, ctx):
    """A vocabulary of more than 64 models alive on one context (the context keeps one page of
    pinned memory with 64 slots for the models' host-visible statistics flags): the models
    beyond it work without one — E-step and M-step against the oracle on the 70th."""
    hm, X, lens = synth_case(G, 6, 2, 9, [40, 33, 52])
    models = [ctx.model(hm) for _ in range(70)]
    corpus = ctx.corpus(X, lens)
    stats = ctx.stats(6, 2, 9)
    ref, _ = O.estep(hm, X, lens, dumps=False)
    for m in (models[0], models[69]):
        ctx.estep(m, corpus, stats)
        assert_close(stats.download(), ref, what="statistics")
        ctx.mstep(m, stats)
        for a, b in zip(m.get().arrays(), O.mstep(hm, ref).arrays()):
            assert_close(a, b, rtol=1e-7, what="model")
    for o in models + [corpus, stats]:
        o.close()


def test_the_widest_model_the_recursions_take(G, ctx):
    """512 states (ghmm_wide.hpp's limit: 64 KB of LDS in the backward pass) against the oracle."""
    hm, X, lens = synth_case(G, 512, 1, 3, [530, 40], seed=77)
    ref, _ = O.estep(hm, X, lens, dumps=False)
    model, corpus = ctx.model(hm), ctx.corpus(X, lens)
    stats = ctx.stats(512, 1, 3)
    try:
        ctx.estep(model, corpus, stats)
        got = stats.download()
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        assert_close(got, ref, what="statistics at 512 states")
    finally:
        for o_ in (model, corpus, stats):
            o_.close()

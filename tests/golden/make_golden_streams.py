#!/usr/bin/env python3
"""Golden fixture for several feature streams (param_number = 2) from the REAL reference, as
shipped (its capacities allow 6 streams of <= 9 coefficients, <= 3 mixtures): streams_p2.json.

Run in the build container only (needs /root/reference and oracle/_ref):
    sh oracle/build_ref.sh && python tests/golden/make_golden_streams.py

  train_all13_p2   hmm-continuous-train-fs all13p2 6 2 3 2 list1 list2 out.hmm over the 13 bundled
                   utterances: stream 1 = the bundled 9-d frames, stream 2 = tests/streams_util.py's
                   5-d differences; mean probability, iterations, the written model
  recog13_p2       13 one-utterance word models (6 states, 1 + 1 mixtures) trained the same way,
                   then recognition-continuous-test-fs 1 models.txt 1 list1 list2 words.txt out.txt:
                   the printed scores per spoken word
Nothing of the reference's source is stored: only inputs and outputs."""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _load import ghmm as _ghmm  # noqa: E402
from streams_util import second_stream  # noqa: E402

G = _ghmm()
REF = os.environ.get("GHMM_REFERENCE", "/root/reference")
BIN = os.path.join(ROOT, "oracle", "_ref")
WORDS = [l.strip() for l in open(os.path.join(REF, "test/test/words.txt")) if l.strip()]
MEAN_LIST = [os.path.basename(l.strip()) for l in
             open(os.path.join(REF, "test/test/perfil_data/mean_list.txt")) if l.strip()]


def read_hmm_streams(path):
    """Parser of the reference's .hmm layout for P streams (TF:2043-2146), 8-byte length."""
    raw = open(path, "rb").read()
    o = 0
    (n,) = struct.unpack_from("<Q", raw, o); o += 8
    word = raw[o:o + n].decode(); o += n
    N, P = struct.unpack_from("<ii", raw, o); o += 8
    M = list(struct.unpack_from(f"<{P}i", raw, o)); o += 4 * P
    D = list(struct.unpack_from(f"<{P}i", raw, o)); o += 4 * P
    A = np.frombuffer(raw, "<f8", N * N, o).reshape(N, N).copy(); o += 8 * N * N
    streams = []
    for p in range(P):
        c = np.zeros((N, M[p])); mean = np.zeros((N, M[p], D[p])); iv = np.zeros((N, M[p], D[p]))
        det = np.zeros((N, M[p]))
        for i in range(N):
            c[i] = np.frombuffer(raw, "<f8", M[p], o); o += 8 * M[p]
            for k in range(M[p]):
                mean[i, k] = np.frombuffer(raw, "<f8", D[p], o); o += 8 * D[p]
                (det[i, k],) = struct.unpack_from("<d", raw, o); o += 8
                iv[i, k] = np.frombuffer(raw, "<f8", D[p], o); o += 8 * D[p]
        streams.append({"c": c.tolist(), "mean": mean.tolist(), "inv_var": iv.tolist(), "det": det.tolist()})
    assert o == len(raw), (o, len(raw))
    return {"word": word, "N": N, "P": P, "M": M, "D": D, "A": A.tolist(), "streams": streams}


def run(cmd, cwd):
    p = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return p.returncode, p.stdout.decode(errors="replace")


def train(tmp, word, lists, N, Ms):
    out = os.path.join(tmp, f"{word}.hmm")
    rc, txt = run([os.path.join(BIN, "hmm-continuous-train-fs"), word, str(N), str(len(lists))] +
                  [str(m) for m in Ms] + lists + [out], tmp)
    assert rc == 0, txt
    rep = open(os.path.join(tmp, f"{word}.txt")).read()
    mp = float(re.search(r"mean probability: (\S+)", rep).group(1))
    it = int(re.search(r"number of iterations: (\d+)", rep).group(1))
    return mp, it, read_hmm_streams(out), out


def main():
    pdir = os.path.join(HERE, "perfil")
    out = {"words": WORDS, "mean_list": MEAN_LIST, "D2": 5}
    with tempfile.TemporaryDirectory() as tmp:
        s1, s2 = [], []
        for fn in MEAN_LIST:
            X = G.perfil_read(os.path.join(pdir, fn))
            p2 = os.path.join(tmp, "d_" + fn)
            G.perfil_write(p2, second_stream(X))
            s1.append(os.path.join(pdir, fn))
            s2.append(p2)

        def lists(tag, idx):
            a, b = os.path.join(tmp, f"{tag}_1.txt"), os.path.join(tmp, f"{tag}_2.txt")
            open(a, "w").write("\n".join(s1[i] for i in idx) + "\n")
            open(b, "w").write("\n".join(s2[i] for i in idx) + "\n")
            return [a, b]

        mp, it, model, _ = train(tmp, "all13p2", lists("all", range(13)), 6, [3, 2])
        out["train_all13_p2"] = {"mean_probability": mp, "iterations": it, "model": model}
        print(f"train all13 P=2 (3 + 2 mixtures): {mp:.6f} in {it}")
        # one model per word from its own utterance, then the recogniser over all 13 utterances
        by_word = {os.path.basename(f)[5:-7]: k for k, f in enumerate(MEAN_LIST)}
        paths, per_word = [], {}
        for w in WORDS:
            mp, it, model, path = train(tmp, w, lists(w, [by_word[w]]), 6, [1, 1])
            per_word[w] = {"mean_probability": mp, "iterations": it, "model": model}
            paths.append(path)
            print(f"train {w} P=2: {mp:.6f} in {it}")
        open(os.path.join(tmp, "models.txt"), "w").write("\n".join(paths) + "\n")
        open(os.path.join(tmp, "words.txt"), "w").write("\n".join(WORDS) + "\n")
        rc, txt = run([os.path.join(BIN, "recognition-continuous-test-fs"), "1", "models.txt", "1"] +
                      lists("rec", range(13)) + ["words.txt", "report.txt"], tmp)
        assert rc == 0, txt
        report = open(os.path.join(tmp, "report.txt")).read()
    blocks, cur = [], None
    for line in txt.replace("\r", "").split("\n"):
        m = re.match(r"Spoken word: (\S+)", line)
        if m:
            cur = {"spoken": m.group(1), "ranking": []}
            blocks.append(cur)
            continue
        m = re.match(r"(\S+) :  (\S+) $", line)
        if m and cur is not None:
            cur["ranking"].append([m.group(1), m.group(2)])
    out["train13_p2"] = per_word
    out["recog13_p2"] = {"blocks": blocks,
                         "report": [l for l in report.split("\n")
                                    if not l.startswith("Date and time") and "recognition time" not in l
                                    and not l.startswith("Model name")]}
    with open(os.path.join(HERE, "streams_p2.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("done:", len(blocks), "recognition blocks")


if __name__ == "__main__":
    main()

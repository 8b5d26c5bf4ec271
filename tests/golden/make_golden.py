#!/usr/bin/env python3
"""Generates the golden fixtures in this directory from the REAL reference.

Run in the build container only (needs /root/reference):
    sh oracle/build_ref.sh && python tests/golden/make_golden.py

Inputs : the 13 bundled .perfil files (copied here verbatim as data fixtures under
         perfil/), and synthetic utterances from the product's deterministic
         generator (ghmm_synth.c, seed 20260104).
Outputs: *.npz  — function-level dumps of the reference's own functions driven by
                  oracle/ref_harness.c (model before, b, post, alpha^, beta^, c_t,
                  log P per utterance, every accumulator, model after one M-step)
         whole_program.json — what the reference executables print/write when run
                  as whole programs (mean log-likelihood, iteration count, final
                  model, recognition scores and report)
Nothing of the reference's source is stored: only inputs and outputs.
"""
import json
import os
import re
import resource
import shutil
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _load import ghmm as _ghmm  # noqa: E402

G = _ghmm()
REF = os.environ.get("GHMM_REFERENCE", "/root/reference")
BIN = os.path.join(ROOT, "oracle", "_ref")
WORDS = [l.strip() for l in open(os.path.join(REF, "test/test/words.txt")) if l.strip()]
MEAN_LIST = [os.path.basename(l.strip()) for l in
             open(os.path.join(REF, "test/test/perfil_data/mean_list.txt")) if l.strip()]


def big_stack():
    resource.setrlimit(resource.RLIMIT_STACK, (resource.RLIM_INFINITY, resource.RLIM_INFINITY))


def run(cmd, cwd):
    p = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       preexec_fn=big_stack)
    return p.returncode, p.stdout.decode(errors="replace")


def parse_dump(path):
    out = {}
    with open(path, "rb") as f:
        while True:
            hdr = f.read(32)
            if len(hdr) < 32:
                break
            name = hdr.split(b"\0")[0].decode()
            nd, = struct.unpack("i", f.read(4))
            dims = struct.unpack("4q", f.read(32))[:nd]
            n = int(np.prod(dims))
            out[name] = np.frombuffer(f.read(8 * n), dtype=np.float64).reshape(dims).copy()
    return out


def harness_case(name, files, N, M, init_model=None):
    """files: list of .perfil paths; init_model: HostModel or None (reference init)."""
    with tempfile.TemporaryDirectory() as tmp:
        with open(os.path.join(tmp, "list.txt"), "w") as f:
            f.write("\n".join(files) + "\n")
        init = "-"
        if init_model is not None:
            init = os.path.join(tmp, "init.hmm")
            init_model.write(init, 8)
        rc, out = run([os.path.join(BIN, "ref_harness"), "list.txt", str(N), str(M), init,
                       "dump.bin"], tmp)
        assert rc == 0, out
        d = parse_dump(os.path.join(tmp, "dump.bin"))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    print(f"{name}: {len(d)} arrays, loglik={float(d['stats.loglik'][0]):.6f}")
    return d


def parse_report(txt):
    mp = float(re.search(r"mean probability: (\S+)", txt).group(1))
    it = int(re.search(r"number of iterations: (\d+)", txt).group(1))
    return mp, it


def train_program(files, word, N, M, big=False):
    exe = "hmm-continuous-train-fs" + ("-big" if big else "")
    with tempfile.TemporaryDirectory() as tmp:
        with open(os.path.join(tmp, "list.txt"), "w") as f:
            f.write("\n".join(files) + "\n")
        rc, out = run([os.path.join(BIN, exe), word, str(N), "1", str(M), "list.txt",
                       "out.hmm"], tmp)
        assert rc == 0, out
        mp, it = parse_report(open(os.path.join(tmp, "out.txt")).read())
        hm = G.HostModel.read(os.path.join(tmp, "out.hmm"))
        raw = open(os.path.join(tmp, "out.hmm"), "rb").read()
    return mp, it, hm, raw


def model_json(hm):
    return {k: np.asarray(v).tolist() for k, v in
            zip(("A", "c", "mean", "inv_var", "det"), hm.arrays())}


def main():
    assert os.path.isdir(REF), "reference not present"
    assert os.path.exists(os.path.join(BIN, "ref_harness")), "run oracle/build_ref.sh first"
    # ---- data fixtures: the bundled utterances
    pdir = os.path.join(HERE, "perfil")
    os.makedirs(pdir, exist_ok=True)
    for fn in MEAN_LIST:
        shutil.copyfile(os.path.join(REF, "train/test/perfil_data", fn), os.path.join(pdir, fn))
    bundled = [os.path.join(pdir, fn) for fn in MEAN_LIST]
    w186 = os.path.join(pdir, "mean_vc_186_f_03_ap_0225.perfil")

    # ---- function-level dumps
    harness_case("bundled186_m1", [w186], 6, 1)
    harness_case("bundled13_m3", bundled, 6, 3)

    def synth_files(tmp, N, M, D, lens, first=0):
        mean, std = G.synth_truth(N, M, D)
        X = G.synth_utterances(mean, std, lens, first_utt=first)
        files, o = [], 0
        for u, T in enumerate(lens):
            p = os.path.join(tmp, f"s{u:03d}.perfil")
            G.perfil_write(p, X[o:o + T])
            files.append(p)
            o += T
        return mean, std, files

    with tempfile.TemporaryDirectory() as tmp:
        mean, std, files = synth_files(tmp, 10, 8, 39, [120, 97, 64, 110])
        start = G.synth_start_model(mean, std, 0.05)
        harness_case("synth39_m8", files, 10, 8, start)
        harness_case("synth39_m8_refinit", files, 10, 8, None)
    with tempfile.TemporaryDirectory() as tmp:
        mean, std, files = synth_files(tmp, 10, 64, 39, [60, 50])
        start = G.synth_start_model(mean, std, 0.05)
        harness_case("synth39_m64", files, 10, 64, start)

    # ---- whole programs
    wp = {"train13_m1": {}, "words": WORDS, "mean_list": MEAN_LIST}
    models = {}
    with tempfile.TemporaryDirectory() as mdir:
        for word in WORDS:
            fn = os.path.join(pdir, f"mean_{word}.perfil")
            mp, it, hm, raw = train_program([fn], word, 6, 1)
            wp["train13_m1"][word] = {"mean_probability": mp, "iterations": it}
            models[word] = hm
            hm.write(os.path.join(mdir, f"{word}.hmm"), 8)
            print(f"train {word}: {mp:.6f} in {it}")
        mp, it, hm, _ = train_program(bundled, "all13", 6, 3)
        wp["train_all13_m3"] = {"mean_probability": mp, "iterations": it, "model": model_json(hm)}
        print(f"train all13 m3: {mp:.6f} in {it}")

        # recognition with the 13 diagonal models just trained, reference argv:
        #   1 models.txt 1 mean_list.txt words.txt out.txt   (test/test/Run Arguments.txt)
        with open(os.path.join(mdir, "models.txt"), "w") as f:
            f.write("\n".join(os.path.join(mdir, f"{w}.hmm") for w in WORDS) + "\n")
        with open(os.path.join(mdir, "mean_list.txt"), "w") as f:
            f.write("\n".join(bundled) + "\n")
        with open(os.path.join(mdir, "words.txt"), "w") as f:
            f.write("\n".join(WORDS) + "\n")
        rc, out = run([os.path.join(BIN, "recognition-continuous-test-fs"), "1", "models.txt", "1",
                       "mean_list.txt", "words.txt", "report.txt"], mdir)
        assert rc == 0, out
        report = open(os.path.join(mdir, "report.txt")).read()
    # stdout of writing_result (RF:1084): "<word> :  <score>" 13 rows per spoken word
    blocks, cur = [], None
    for line in out.replace("\r", "").split("\n"):
        m = re.match(r"Spoken word: (\S+)", line)
        if m:
            cur = {"spoken": m.group(1), "ranking": []}
            blocks.append(cur)
            continue
        m = re.match(r"(\S+) :  (\S+) $", line)
        if m and cur is not None:
            cur["ranking"].append([m.group(1), m.group(2)])
    wp["recog13_m1"] = {
        "blocks": blocks,
        "report": [l for l in report.split("\n")
                   if not l.startswith("Date and time") and "recognition time" not in l
                   and not l.startswith("Model name")],
    }
    np.savez_compressed(os.path.join(HERE, "train13_m1_models.npz"),
                        **{f"{w}.{k}": v for w, hm in models.items()
                           for k, v in zip(("A", "c", "mean", "inv_var", "det"), hm.arrays())})

    # synthetic 39-d whole-program training (raised capacities), reference init
    with tempfile.TemporaryDirectory() as tmp:
        lens = [80 + (37 * u) % 61 for u in range(20)]
        mean, std, files = synth_files(tmp, 10, 8, 39, lens, first=1000)
        mp, it, hm, _ = train_program(files, "synth", 10, 8, big=True)
        wp["train_synth39_m8"] = {"lens": lens, "first_utt": 1000, "mean_probability": mp,
                                  "iterations": it}
        np.savez_compressed(os.path.join(HERE, "train_synth39_m8_model.npz"),
                            **dict(zip(("A", "c", "mean", "inv_var", "det"), hm.arrays())))
        print(f"train synth39 m8: {mp:.6f} in {it}")
    with open(os.path.join(HERE, "whole_program.json"), "w") as f:
        json.dump(wp, f, indent=1)
    print("done")


if __name__ == "__main__":
    main()

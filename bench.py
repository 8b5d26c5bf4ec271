#!/usr/bin/env python3
"""bench.py — Baum-Welch frames/sec on MI355X (BASELINE.json's metric).

One "step" = one full EM iteration (E-step over every utterance + all-reduce of the
sufficient statistics when N > 1 + M-step) of a 39-d, 10-state x 8-mixture
diagonal GMM-HMM over 1 000 synthetic utterances x 300 frames PER GPU
(BASELINE.json configs[1]; weak scaling: every rank holds its own 1 000 utterances
of one conceptual corpus, no data-path collective besides the statistics sum).
Frames are resident in HBM before the timed region starts.  Before the W warmup steps the same
step runs --spinup times (default 400 = 0.1 s, untimed, reported as "clock_spinup_steps"): an idle
GPU runs its first ~25 steps 10 % slower, and W = 5 steps are 1.3 ms.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task description) with two extra
objects: "roofline" (the emission kernel, algorithmic bytes / HIP-event duration
measured on the kernel's own stream) and "cpu_baseline" (the oracle, i.e. a port of
the reference's C path, timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _load import load_pkg  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 measured copy rate
F64_PEAK_TFLOPS = 78.6  # MI355X FP64 vector = matrix peak


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--spinup", type=int, default=400,
                    help="EM steps run (untimed) before the warmup steps so that the GPU is at its operating "
                         "clocks: after idling, the first ~25 steps (6 ms) run 10 %% slower (0 = none)")
    ap.add_argument("--utts", type=int, default=1000, help="utterances per GPU")
    ap.add_argument("--frames", type=int, default=300, help="frames per utterance")
    ap.add_argument("--states", type=int, default=10)
    ap.add_argument("--mix", type=int, default=8)
    ap.add_argument("--dim", type=int, default=39)
    ap.add_argument("--kernels", type=int, default=0, help="0 auto, 1 vector-ALU, 2 MFMA")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the SURVEY §8(d) side measurements (decode, 64 mixtures, 2 000 states, "
                         "ragged lengths, EM from the reference's initial model)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    pkg = load_pkg()
    G, em = pkg.ghmm, pkg.em
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "GHMM_FORCE_DEVICE" in os.environ:   # rehearsal of the N > 1 path on a 1-GPU box (gloo)
        local = int(os.environ["GHMM_FORCE_DEVICE"])
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the GMM-HMM path has no CPU fallback")
    torch.cuda.set_device(local)
    if world > 1:
        backend = os.environ.get("GHMM_DIST_BACKEND", "nccl")  # "nccl" IS RCCL on ROCm
        kw = {"device_id": torch.device(f"cuda:{local}")} if backend == "nccl" else {}
        dist.init_process_group(backend, **kw)
    # ONE explicit (non-default) stream carries every kernel of the library AND the
    # all-reduce: torch's default stream has handle 0, which the C ABI would read as "make
    # your own stream", and then nothing would order the collective against the E-step.
    stream = torch.cuda.Stream(device=local)
    torch.cuda.set_stream(stream)

    N, M, D, U, T = args.states, args.mix, args.dim, args.utts, args.frames
    # this rank's utterances of the conceptual corpus [rank*U, (rank+1)*U)
    mean, std = G.synth_truth(N, M, D)
    lens = np.full(U, T, dtype=np.int32)
    X = G.synth_utterances(mean, std, lens, first_utt=rank * U)
    start = G.synth_start_model(mean, std, 0.05)
    Xd = torch.from_numpy(X).to(f"cuda:{local}")          # frames resident in HBM
    assert stream.cuda_stream != 0
    ctx = G.Context(local, stream=stream.cuda_stream)
    ctx.set_option(G.OPT_KERNELS, args.kernels)
    model = ctx.model(start)
    corpus = ctx.corpus_from_device(Xd.data_ptr(), lens, D)
    backend = em.HipBackend(G, ctx, model, corpus, torch=torch)
    driver = em.EMDriver(backend, dist if world > 1 else None)
    frames_rank = int(lens.sum())

    def barrier():
        if world > 1:
            dist.barrier()

    # Clock spin-up, then the contract's W warmup steps, then the K timed steps.  Measured on one
    # box (steps 20): warmup 5 alone 0.275 ms per step, warmup 50 0.249, warmup 200 0.246 — an idle
    # MI355X needs some tens of milliseconds of work to reach its operating clocks, and 5 steps
    # are 1.3 ms.  The spin-up is the same EM step, a fixed count on every rank (the all-reduce
    # keeps the ranks in lockstep), reported in the JSON line as "clock_spinup_steps".
    for _ in range(max(0, args.spinup)):
        driver.step()
    for _ in range(args.warmup):
        driver.step()
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        driver.step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loglik = backend.loglik()

    # second pass of the same K steps with every kernel bracketed by HIP events on the
    # stream the kernels run on (kept out of the timed region above: two event records
    # per launch would perturb a ~100 us step)
    ctx.set_option(G.OPT_TIMING, 1)
    ctx.kernel_times_reset()
    for _ in range(args.steps):
        driver.step()
    kt = ctx.kernel_times()
    ctx.set_option(G.OPT_TIMING, 0)
    # per EM step: total HIP-event time of each kernel class / steps (a class may hold
    # several launches per step, e.g. "reduce")
    kavg = {k: (ms / args.steps if n else None) for k, (ms, n) in kt.items()}

    if rank == 0:
        Gn = N * M
        value = world * frames_rank * args.steps / elapsed
        # emission kernel, posteriors materialised (the training path): per frame it reads
        # the frame (8D) and writes b (8N) and post (8G) — SURVEY.md §8(d) "emission +
        # posteriors materialised" = 1 032 B at 10x8
        emis_ms = kavg.get("emission")
        bytes_per_frame = 8 * (D + N + Gn)
        flops_per_frame = Gn * (3 * D + 2)
        roofline = None
        if emis_ms:
            ach = bytes_per_frame * frames_rank / (emis_ms * 1e-3) / 1e9
            roofline = {
                "kernel": "k_emission", "bound": "hbm", "achieved": round(ach, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                "traffic": emission_traffic(frames_rank, N, M, D),
                "avg_kernel_ms": round(emis_ms, 5),
                "bytes_per_frame": bytes_per_frame, "frames_per_launch": frames_rank,
                "f64_tflops": round(flops_per_frame * frames_rank / (emis_ms * 1e-3) / 1e12, 3),
                "f64_frac": round(flops_per_frame * frames_rank / (emis_ms * 1e-3) / 1e12
                                  / F64_PEAK_TFLOPS, 4),
            }
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(G, start, X, lens, args.cpu_seconds)
        extras = None
        if world == 1 and not args.no_extras and (N, M, D, U, T) == (10, 8, 39, 1000, 300):
            extras = side_measurements(G, em, ctx, torch, local, mean, std, start, corpus, kavg)
        out = {
            "metric": "frames/sec Baum-Welch (39-d MFCC, 10 states x 8 mix)",
            "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"Baum-Welch EM iteration, {D}-d, {N} states x {M} mix diag GMM-HMM, "
                                   f"{U} synthetic utterances x {T} frames per GPU (BASELINE configs[1])",
                       "utterances_per_gpu": U, "frames_per_utterance": T,
                       "frames_per_step": world * frames_rank,
                       "parallelism": f"utterance-sharded x{world}, 1 all-reduce of "
                                      f"{G.stats_len(N, M, D)} f64 per iteration"},
            "roofline": roofline, "cpu_baseline": cpu,
            "kernel_ms": {k: (round(v, 5) if v else None) for k, v in kavg.items()},
            "clock_spinup_steps": max(0, args.spinup),
            "extras": extras,
            "loglik_per_frame": round(loglik / (world * frames_rank), 6),
        }
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


def _timed_steps(ctx, G, fn, steps, warmup=2):
    """wall ms per call of fn() (stream drained on both sides) and per-kernel HIP-event ms."""
    for _ in range(warmup):
        fn()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    ctx.sync()
    wall = 1e3 * (time.perf_counter() - t0) / steps
    ctx.set_option(G.OPT_TIMING, 1)
    ctx.kernel_times_reset()
    for _ in range(steps):
        fn()
    kt = {k: ms / steps for k, (ms, n) in ctx.kernel_times().items() if n}
    ctx.set_option(G.OPT_TIMING, 0)
    return wall, kt


def side_measurements(G, em, ctx, torch, dev, mean, std, start, corpus, kavg):
    """What SURVEY §8(d) lists besides the headline, each bounded to a few seconds (N = 1 only):
    EM from the reference's own initial model (collapsed components), ragged utterance lengths,
    decode over configs[2], the per-GPU share of configs[3] (64 mixtures), and the 2 000-state
    emission of configs[4].  Times are HIP-event sums of the kernels (device) and host wall
    clock around the C-ABI call (wall, includes the device-to-host copy of the results)."""
    N, M, D = 10, 8, 39
    out = {}
    rnd = lambda v, k=4: round(float(v), k)  # noqa: E731

    # (1) EM from ghmm_model_init's model on the bench corpus: iterations 3-10, the regime the
    # real trainer runs in (variance-floored components collapse onto single frames)
    model = ctx.model(start)
    model.init_from(corpus)      # (the first call also sizes its workspace and loads its kernels)
    ctx.sync()
    t0 = time.perf_counter()
    model.init_from(corpus)
    ctx.sync()
    init_ms = 1e3 * (time.perf_counter() - t0)
    be = em.HipBackend(G, ctx, model, corpus)
    drv = em.EMDriver(be)
    for _ in range(2):
        drv.step()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(8):
        drv.step()
    ctx.sync()
    out["refinit_ms_per_step"] = rnd(1e3 * (time.perf_counter() - t0) / 8)
    out["refinit"] = {"initial_model_ms": rnd(init_ms, 2), "iterations_timed": "3-10",
                      "loglik_per_frame": rnd(be.loglik() / corpus.frames, 6)}
    be.stats.close()
    model.close()

    # (2) ragged lengths, T ~ U[100, 500], same utterance count (frames within 1 % of the fixed run)
    rng = np.random.default_rng(20260104)
    lens = rng.integers(100, 501, size=1000).astype(np.int32)
    Xr = torch.from_numpy(G.synth_utterances(mean, std, lens)).to(f"cuda:{dev}")
    cr = ctx.corpus_from_device(Xr.data_ptr(), lens, D)
    model = ctx.model(start)
    be = em.HipBackend(G, ctx, model, cr)
    drv = em.EMDriver(be)
    wall, kt = _timed_steps(ctx, G, drv.step, 10)
    fr = int(lens.sum())
    out["ragged"] = {"workload": "T ~ U[100, 500], 1 000 utterances", "frames": fr,
                     "ms_per_step": rnd(wall), "frames_per_s": rnd(fr / (wall * 1e-3), 1),
                     "scan_ms": rnd(kt.get("forward", 0) + kt.get("backward", 0)),
                     "fixed_length_scan_ms": rnd((kavg.get("forward") or 0) + (kavg.get("backward") or 0))}
    be.stats.close(); model.close(); cr.close()
    del Xr

    # (3) decode, BASELINE configs[2]: 10 000 utterances x 300 frames, forward score + Viterbi
    lens = np.full(10000, 300, dtype=np.int32)
    Xh = G.synth_utterances(mean, std, lens, first_utt=100000)
    Xd = torch.from_numpy(Xh).to(f"cuda:{dev}")
    del Xh
    cd = ctx.corpus_from_device(Xd.data_ptr(), lens, D)
    model = ctx.model(start)
    fr = int(lens.sum())
    wall_f, kf = _timed_steps(ctx, G, lambda: ctx.score(model, cd), 3, warmup=1)
    wall_v, kv = _timed_steps(ctx, G, lambda: ctx.viterbi(model, cd), 3, warmup=1)
    dev_f = kf.get("emission", 0) + kf.get("forward", 0)
    dev_v = kv.get("emission", 0) + kv.get("viterbi", 0)
    out["decode"] = {"workload": "10 000 utterances x 300 frames (BASELINE configs[2])", "frames": fr,
                     "forward_device_ms": rnd(dev_f), "forward_wall_ms": rnd(wall_f),
                     "forward_frames_per_s_device": rnd(fr / (dev_f * 1e-3), 1),
                     "forward_frames_per_s_wall": rnd(fr / (wall_f * 1e-3), 1),
                     "viterbi_device_ms": rnd(dev_v), "viterbi_wall_ms": rnd(wall_v),
                     "viterbi_frames_per_s_device": rnd(fr / (dev_v * 1e-3), 1),
                     "viterbi_frames_per_s_wall": rnd(fr / (wall_v * 1e-3), 1),
                     "path_bytes_device_to_host": fr}
    model.close(); cd.close()
    del Xd

    # (4) 64 mixtures per state, one GPU's share of BASELINE configs[3]: 12 500 x 300 frames
    M64 = 64
    mean64, std64 = G.synth_truth(N, M64, D)
    lens = np.full(12500, 300, dtype=np.int32)
    Xh = G.synth_utterances(mean64, std64, lens)
    Xd = torch.from_numpy(Xh).to(f"cuda:{dev}")
    del Xh
    c64 = ctx.corpus_from_device(Xd.data_ptr(), lens, D)
    model = ctx.model(G.synth_start_model(mean64, std64, 0.05))
    be = em.HipBackend(G, ctx, model, c64)
    drv = em.EMDriver(be)
    wall, kt = _timed_steps(ctx, G, drv.step, 3, warmup=1)
    fr = int(lens.sum())
    out["m64"] = {"workload": "10 states x 64 mix, 12 500 utterances x 300 frames = one GPU's share of "
                              "BASELINE configs[3]", "frames": fr, "ms_per_step": rnd(wall, 3),
                  "frames_per_s": rnd(fr / (wall * 1e-3), 1),
                  "kernel_ms": {k: rnd(v, 3) for k, v in kt.items()}}
    be.stats.close(); model.close(); c64.close()
    del Xd
    torch.cuda.empty_cache()

    # (5) BASELINE configs[4]: emission only, 2 000 tied states x 16 mixtures, 1 M frames
    out["config5"] = config5_emission(G, ctx, torch, dev)
    return out


def config5_emission(G, ctx, torch, dev, F=1_000_000, N=2000, M=16, D=39):
    """b only (the recogniser's calc_symbol_probab, RF:860-889) over one [F][39] matrix; the
    matrix-core work is the expanded Mahalanobis form, 2 * F * 80 * G flop."""
    rng = np.random.default_rng(11)
    centre = rng.normal(0.0, 2.0, size=(N, 1, D))
    mean = centre + rng.normal(0.0, 0.7, size=(N, M, D))
    var = rng.uniform(0.5, 1.5, size=(N, M, D)) ** 2
    c = rng.uniform(0.5, 1.5, size=(N, M))
    c /= c.sum(1, keepdims=True)
    hm = G.HostModel(np.zeros((N, N)), c, mean, 1.0 / var, var.prod(axis=2))
    g = torch.Generator(device=f"cuda:{dev}")
    g.manual_seed(2)
    cen = torch.from_numpy(centre[:, 0, :]).to(f"cuda:{dev}")
    pick = torch.randint(0, N, (F,), generator=g, device=f"cuda:{dev}")
    X = cen[pick] + 1.2 * torch.randn((F, D), generator=g, device=f"cuda:{dev}", dtype=torch.float64)
    torch.cuda.synchronize()
    model = ctx.model(hm)
    corpus = ctx.corpus_from_device(X.data_ptr(), np.array([F], dtype=np.int32), D)
    wall, kt = _timed_steps(ctx, G, lambda: ctx.emission(model, corpus, False), 2, warmup=1)
    ms = kt.get("emission", wall)
    flop = 2.0 * F * 80 * N * M
    tf = flop / (ms * 1e-3) / 1e12
    # a look at the result: one block of frames, finite and positive somewhere in every frame
    b = ctx.fetch_range(G.BUF_B, 0, (4096, N))
    ok = bool(np.isfinite(b).all() and (b.max(axis=1) > 0).all())
    model.close(); corpus.close()
    del X
    torch.cuda.empty_cache()
    return {"workload": "emission b only, 2 000 states x 16 mix, 39-d, 1 000 000 frames (BASELINE configs[4])",
            "ms": round(ms, 3), "matrix_tflops": round(tf, 2), "f64_peak_tflops": F64_PEAK_TFLOPS,
            "mfma_frac": round(tf / F64_PEAK_TFLOPS, 4), "output_bytes": 8 * F * N,
            "first_4096_frames_finite_positive": ok}


def emission_traffic(frames, N, M, D):
    """HBM bytes per emission launch from the committed rocprofv3 PMC pass of this same
    workload (profiles/emission_traffic.json; counters cannot be read from inside the
    process).  None when the shape on the command line is not the profiled one."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "emission_traffic.json")))
    except OSError:
        return None
    if t.get("frames_per_launch") != frames or (N, M, D) != (10, 8, 39):
        return None
    return t["traffic_bytes_per_launch"]


def cpu_baseline(G, start, X, lens, budget_s):
    """The oracle (a port of the reference's single-threaded C path, oracle/ghmm_oracle.c)
    on this box's host: whole EM iterations over a bounded sample of the same corpus."""
    import oracle_lib as O
    D = X.shape[1]
    T = int(lens[0])
    # calibrate on 20 utterances, then size the sample for ~budget_s of CPU work
    n0 = min(20, len(lens))
    t = time.perf_counter()
    O.train(start, X[:n0 * T], lens[:n0], max_iter=1, fixed_iter=True)
    per_utt = (time.perf_counter() - t) / n0
    n = int(max(n0, min(len(lens), budget_s / per_utt / 2)))
    iters = int(max(1, min(10, budget_s / (per_utt * n))))
    t = time.perf_counter()
    O.train(start, X[:n * T], lens[:n], max_iter=iters, fixed_iter=True)
    dt = time.perf_counter() - t
    out = {"value": round(n * T * iters / dt, 1), "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": f"{iters} EM iterations over the first {n} utterances ({n * T} frames) of the "
                     f"benchmark corpus, oracle/ghmm_oracle.c (gcc -O2), {dt:.1f} s"}
    # SURVEY §8(d)'s optional second figure: the same E-step with the utterances dealt to every
    # host core this process may use (the reference itself is single-threaded)
    # (threads = the CPUs this process may run on, at most 16: a one-GPU box's CPU share)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, 16)
    if cores > 1:
        nn, reps, dt = len(lens), 0, 0.0
        t = time.perf_counter()
        while dt < 3.0 and reps < 50:
            O.estep_mt(start, X[:nn * T], lens[:nn], cores)
            reps += 1
            dt = time.perf_counter() - t
        out["all_cores"] = {"value": round(reps * nn * T / dt, 1), "unit": "frames/s", "cores": cores,
                            "sample": f"{reps} E-steps over {nn} utterances on {cores} threads, {dt:.1f} s"}
    return out


if __name__ == "__main__":
    main()

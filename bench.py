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
    python bench.py --gpus N --steps K --warmup W        # starts its N ranks itself (launch.py)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task description) with two extra
objects: "roofline" (the emission kernel, algorithmic bytes / HIP-event duration
measured on the kernel's own stream) and "cpu_baseline" (the oracle, i.e. a port of
the reference's C path, timed on this box's host cores on a bounded sample).  Further keys:
"ms_per_step_cold" / "value_cold" (the same W + K steps before the clock spin-up), "job_ms" (a
whole training job: upload + initial model + EM under the reference's stopping rule),
"ranks" / "allreduce_ms" (N > 1: ranks as the process group sees them, HIP-event time of the
statistics all-reduce per step), extras.config4 at 8 GPUs (BASELINE configs[3]); every extra
carries "checks_ok" (invariants of its results, evaluated outside the timed regions).
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
    ap.add_argument("--fused-scan", type=int, default=0,
                    help="GHMM_OPT_FUSED_SCAN: 0 / 1 the recursions in one launch, 2 in separate launches (A/B)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the SURVEY §8(d) side measurements (decode, 64 mixtures, 2 000 states, "
                         "ragged lengths, EM from the reference's initial model)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--config4", choices=("auto", "on", "off"), default="auto",
                    help="N > 1 only: BASELINE configs[3] (64 mixtures, 12 500 utterances per GPU, fixed 10 "
                         "iterations) as extras.config4; auto = at 8 GPUs")
    ap.add_argument("--config4-utts", type=int, default=12500, help="utterances per GPU of --config4")
    args = ap.parse_args()

    pkg = load_pkg()
    if args.gpus > 1 and not pkg.launch.under_launcher():
        # `python bench.py --gpus N` as the driver types it: this process becomes the launcher of
        # N child ranks (one per GPU) BEFORE anything here touches the GPU, passes rank 0's JSON
        # line through and exits with the children's return code (launch.py).
        raise SystemExit(pkg.launch.self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus))

    import torch
    import torch.distributed as dist

    G, em = pkg.ghmm, pkg.em
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "GHMM_FORCE_DEVICE" in os.environ:   # rehearsal of the N > 1 path on a 1-GPU box (gloo)
        local = int(os.environ["GHMM_FORCE_DEVICE"])
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher started "
                         f"{world} ranks; use --nproc-per-node {args.gpus}")
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
    ctx.set_option(G.OPT_FUSED_SCAN, args.fused_scan)
    model = ctx.model(start)
    corpus = ctx.corpus_from_device(Xd.data_ptr(), lens, D)
    backend = em.HipBackend(G, ctx, model, corpus, torch=torch)
    driver = em.EMDriver(backend, dist if world > 1 else None)
    frames_rank = int(lens.sum())

    def barrier():
        if world > 1:
            dist.barrier()

    def timed(steps):
        """EXACTLY `steps` EM iterations bracketed by a barrier + device synchronisation on both
        sides; the MAX over ranks of the wall time (s)."""
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            driver.step()
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local}")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # (1) COLD: the contract's W warmup steps and K timed steps on a GPU that has just been
    # idle (what a reference-style training job of ~10 iterations sees): "ms_per_step_cold".
    for _ in range(args.warmup):
        driver.step()
    elapsed_cold = timed(args.steps)
    # (2) Clock spin-up, then W warmup steps, then the K timed steps: the headline.  Measured on
    # one box (steps 20): warmup 5 alone 0.275 ms per step, warmup 50 0.249, warmup 200 0.246 — an
    # idle MI355X needs some tens of milliseconds of work to reach its operating clocks, and 5
    # steps are 1.3 ms.  The spin-up is the same EM step, a fixed count on every rank (the
    # all-reduce keeps the ranks in lockstep), reported as "clock_spinup_steps".
    for _ in range(max(0, args.spinup)):
        driver.step()
    for _ in range(args.warmup):
        driver.step()
    elapsed = timed(args.steps)
    loglik = backend.loglik()

    # (3) INSTRUMENTED pass of the same K steps: every kernel bracketed by HIP events on the
    # stream the kernels run on (kept out of the timed region above: two event records per
    # launch perturb a ~100 us step — this pass runs ~6 % slower, so "kernel_ms" sums to more
    # than "ms_per_step"), and the all-reduce between two events on that same stream.
    ctx.set_option(G.OPT_TIMING, 1)
    ctx.kernel_times_reset()
    ar_events = []
    for _ in range(args.steps):
        backend.estep()
        if world > 1:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            dist.all_reduce(backend.stats_tensor(), op=dist.ReduceOp.SUM)
            e1.record(stream)
            ar_events.append((e0, e1))
        backend.mstep()
    kt = ctx.kernel_times()
    ctx.set_option(G.OPT_TIMING, 0)
    torch.cuda.synchronize()
    allreduce_ms = (sum(a.elapsed_time(b) for a, b in ar_events) / len(ar_events)) if ar_events else None
    # per EM step: total HIP-event time of each kernel class / steps (a class may hold
    # several launches per step, e.g. "reduce")
    kavg = {k: (ms / args.steps if n else None) for k, (ms, n) in kt.items()}

    # (4) N > 1: BASELINE configs[3] — 64 mixtures, 12 500 utterances x 300 frames PER GPU (100 000
    # over 8 GPUs), fixed 10 EM iterations with the statistics all-reduce (every rank takes part)
    config4 = None
    want4 = args.config4 == "on" or (args.config4 == "auto" and world == 8)
    if world > 1 and want4:
        try:
            config4 = config4_sharded(G, em, ctx, torch, dist, local, rank, world, stream,
                                      args.config4_utts, barrier)
        except Exception as e:   # noqa: BLE001 - the headline line must survive a failing extra
            config4 = {"error": f"{type(e).__name__}: {e}"[:300], "checks_ok": False}

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
            extras = side_measurements(G, em, ctx, torch, local, mean, std, start, corpus, kavg, X)
        if config4 is not None:
            extras = dict(extras or {}, config4=config4)
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
            "kernel_ms_source": "second, instrumented pass of the same K steps (two HIP events per launch; "
                                "it runs a few % slower than the timed pass, so the sum exceeds ms_per_step); "
                                "'forward' = k_scan_combine when 'backward' is null: both scans and the gamma / xi "
                                "pass of the E-step in one launch",
            "clock_spinup_steps": max(0, args.spinup),
            # the same W + K steps BEFORE the spin-up, on a GPU coming out of idle
            "ms_per_step_cold": round(1e3 * elapsed_cold / args.steps, 4),
            "value_cold": round(world * frames_rank * args.steps / elapsed_cold, 1),
            "ranks": dist.get_world_size() if world > 1 else 1,
            "dist_backend": (dist.get_backend() if world > 1 else None),
            "allreduce_ms": (round(allreduce_ms, 5) if allreduce_ms is not None else None),
            "extras": extras,
            "loglik_per_frame": round(loglik / (world * frames_rank), 6),
        }
        if extras and "job" in extras:
            out["job_ms"] = extras["job"]["job_ms"]
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


def config4_sharded(G, em, ctx, torch, dist, dev, rank, world, stream, utts, barrier,
                    N=10, M=64, D=39, T=300, iters=10):
    """BASELINE configs[3]: 39-d, 10 states x 64 mix, `utts` utterances x 300 frames on EVERY
    rank (12 500 x 8 GPUs = 100 000), a fixed 10 EM iterations (SURVEY §8(d) config 4, §8(e):
    "fix the iteration count") with one all-reduce of 50 682 f64 each.  Upload excluded (frames
    resident); barrier + synchronise on both sides; MAX over ranks."""
    mean, std = G.synth_truth(N, M, D)
    lens = np.full(utts, T, dtype=np.int32)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    Xh = G.synth_utterances(mean, std, lens, first_utt=rank * utts, threads=max(1, min(8, cores // world)))
    Xd = torch.from_numpy(Xh).to(f"cuda:{dev}")
    del Xh
    corpus = ctx.corpus_from_device(Xd.data_ptr(), lens, D)
    model = ctx.model(G.synth_start_model(mean, std, 0.05))
    be = em.HipBackend(G, ctx, model, corpus, torch=torch)
    drv = em.EMDriver(be, dist)
    drv.step()                       # sizes the workspace (19 GB of posteriors), untimed
    model.set(G.synth_start_model(mean, std, 0.05))
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(iters):
        drv.step()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{dev}")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    st = be.stats.download()
    fr = int(lens.sum())
    sp = G.split_stats(st, N, M, D)
    # every frame of every rank is counted once in the occupancies (sum over states of
    # den_c = total frames of all ranks when every utterance has a path), n_utt = all utterances
    ok = bool(np.isfinite(st).all() and abs(sp["den_c"].sum() - world * fr) < 1e-6 * world * fr
              and int(round(float(sp["n_utt"]))) == world * utts)
    be.stats.close(); model.close(); corpus.close()
    del Xd
    torch.cuda.empty_cache()
    return {"workload": f"10 states x 64 mix, {utts} utterances x {T} frames per GPU x {world} GPUs "
                        f"(BASELINE configs[3]), fixed {iters} EM iterations, 1 all-reduce of "
                        f"{G.stats_len(N, M, D)} f64 each",
            "iterations": iters, "total_ms": round(1e3 * dt, 3), "ms_per_step": round(1e3 * dt / iters, 3),
            "frames_per_s": round(world * fr * iters / dt, 1),
            "loglik_per_frame": round(float(sp["loglik"]) / (world * fr), 6), "checks_ok": ok}


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


def _stats_checks(G, st, N, M, D, frames, n_utt):
    """What any E-step's statistics must satisfy whatever the data (TF:1642-1664, 1691-1727):
    finite; every frame's occupancy sums to 1 over the states, so sum(den_c) = frames; num_c
    summed over a state's mixtures = den_c of that state; den_a = den_c minus the last frames'
    occupancy (<= den_c); n_utt counted (TF:320)."""
    sp = G.split_stats(st, N, M, D)
    tol = 1e-9 * max(1, frames)
    return bool(np.isfinite(st).all()
                and abs(sp["den_c"].sum() - frames) < tol
                and np.abs(sp["num_c"].sum(axis=1) - sp["den_c"]).max() < tol
                and (sp["den_a"] <= sp["den_c"] + tol).all()
                and int(round(float(sp["n_utt"]))) == n_utt)


def side_measurements(G, em, ctx, torch, dev, mean, std, start, corpus, kavg, Xhost):
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
    st = be.stats.download()
    out["refinit"]["checks_ok"] = _stats_checks(G, st, N, M, D, corpus.frames, corpus.n_utt)
    be.stats.close()
    model.close()

    # (1b) a whole training job as a user of hmm-continuous-train-fs waits for it (file reading
    # excluded): frames from pageable host memory -> HBM, the initial model on the device
    # (TF:732-1317), then EM under the reference's stopping rule (TF:325-358).  Second of two
    # runs (the first sizes the workspace of the new corpus).
    job = None
    for _ in range(2):
        ctx.sync()
        t0 = time.perf_counter()
        cj = ctx.corpus(Xhost, corpus.lens)
        ctx.sync()
        t1 = time.perf_counter()
        mj = ctx.model(start)
        mj.init_from(cj, fetch=False)
        ctx.sync()
        t2 = time.perf_counter()
        bj = em.HipBackend(G, ctx, mj, cj)
        iters, lp = em.EMDriver(bj).train(threshold=1e-3, max_iter=100)
        ctx.sync()
        t3 = time.perf_counter()
        job = {"workload": "upload + ghmm_model_init + EM until the reference's stopping rule "
                           "(relative change of log P <= 1e-3), BASELINE configs[1] corpus",
               "job_ms": rnd(1e3 * (t3 - t0), 3), "upload_ms": rnd(1e3 * (t1 - t0), 3),
               "initial_model_ms": rnd(1e3 * (t2 - t1), 3), "em_ms": rnd(1e3 * (t3 - t2), 3),
               "iterations": iters, "loglik_per_frame": rnd(lp / cj.frames, 6),
               "checks_ok": bool(np.isfinite(lp) and 2 <= iters < 100)}
        bj.stats.close(); mj.close(); cj.close()
    out["job"] = job

    # (2) ragged lengths, T ~ U[100, 500], same utterance count (frames within 1 % of the fixed run)
    rng = np.random.default_rng(20260104)
    lens = rng.integers(100, 501, size=1000).astype(np.int32)
    Xr = torch.from_numpy(G.synth_utterances(mean, std, lens)).to(f"cuda:{dev}")
    cr = ctx.corpus_from_device(Xr.data_ptr(), lens, D)
    model = ctx.model(start)
    be = em.HipBackend(G, ctx, model, cr)
    drv = em.EMDriver(be)
    # measured like the fixed-length corpus beside it: 30 untimed iterations, 60 timed (ten
    # iterations straight after the change of corpus read 12 % slower than the kernels are)
    wall, kt = _timed_steps(ctx, G, drv.step, 60, warmup=30)
    fr = int(lens.sum())
    mf = ctx.model(start)
    bf = em.HipBackend(G, ctx, mf, corpus)
    wall_f, _ = _timed_steps(ctx, G, em.EMDriver(bf).step, 60, warmup=30)
    bf.stats.close(); mf.close()
    out["ragged"] = {"workload": "T ~ U[100, 500], 1 000 utterances", "frames": fr,
                     "ms_per_step": rnd(wall), "frames_per_s": rnd(fr / (wall * 1e-3), 1),
                     "fixed_length_ms_per_step": rnd(wall_f),
                     "per_frame_vs_fixed_length": rnd((wall / fr) / (wall_f / corpus.frames), 3),
                     "scan_ms": rnd(kt.get("forward", 0) + kt.get("backward", 0)),
                     "fixed_length_scan_ms": rnd((kavg.get("forward") or 0) + (kavg.get("backward") or 0))}
    out["ragged"]["checks_ok"] = _stats_checks(G, be.stats.download(), N, M, D, fr, len(lens))
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
    # outside the timed regions: every path starts in state 0, ends in state N - 1 (the
    # reference's one-hot start RF:249-251 and final-state termination TF:1487), moves by 0 or 1
    # state per frame (left-to-right model), and the best path's score cannot exceed the sum over
    # all paths (the forward score, RF:820-836)
    fwd = ctx.score(model, cd)
    path, vsc = ctx.viterbi(model, cd)
    P = path.reshape(len(lens), 300)
    step = np.diff(P, axis=1)
    out["decode"]["checks_ok"] = bool(
        (P[:, 0] == 0).all() and (P[:, -1] == N - 1).all() and ((step == 0) | (step == 1)).all()
        and np.isfinite(fwd).all() and np.isfinite(vsc).all()
        and (vsc <= fwd + 1e-9 * np.abs(fwd)).all())
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
    # outside the timed region: the statistics' invariants, and the posteriors of 1 000 sampled
    # frames: g_ij / b_i sums to 1 over a state's mixtures wherever b_i > 0 (TF:1770-1778)
    ok = _stats_checks(G, be.stats.download(), N, M64, D, fr, len(lens))
    rs = np.random.default_rng(5)
    for f in rs.integers(0, fr, size=1000):
        pr = ctx.fetch_range(G.BUF_POST, int(f) * N * M64, (N, M64)).sum(axis=1)
        bb = ctx.fetch_range(G.BUF_B, int(f) * N, (N,))
        ok = ok and bool(np.all(np.abs(pr[bb > 0] - 1.0) < 1e-9) and np.all(pr[bb == 0] == 0))
    out["m64"]["checks_ok"] = ok
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
            "first_4096_frames_finite_positive": ok, "checks_ok": ok}


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

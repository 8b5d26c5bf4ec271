#!/usr/bin/env python3
"""bench.py — Baum-Welch frames/sec on MI355X (BASELINE.json's metric).

One "step" = one full EM iteration (E-step over every utterance + all-reduce of the
sufficient statistics when N > 1 + M-step) of a 39-d, 10-state x 8-mixture
diagonal GMM-HMM over 1 000 synthetic utterances x 300 frames PER GPU
(BASELINE.json configs[1]; weak scaling: every rank holds its own 1 000 utterances
of one conceptual corpus, no data-path collective besides the statistics sum).
Frames are resident in HBM before the timed region starts.

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
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--utts", type=int, default=1000, help="utterances per GPU")
    ap.add_argument("--frames", type=int, default=300, help="frames per utterance")
    ap.add_argument("--states", type=int, default=10)
    ap.add_argument("--mix", type=int, default=8)
    ap.add_argument("--dim", type=int, default=39)
    ap.add_argument("--kernels", type=int, default=0, help="0 auto, 1 vector-ALU, 2 MFMA")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
            "loglik_per_frame": round(loglik / (world * frames_rank), 6),
        }
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


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
    return {"value": round(n * T * iters / dt, 1), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{iters} EM iterations over the first {n} utterances ({n * T} frames) of the "
                      f"benchmark corpus, oracle/ghmm_oracle.c (gcc -O2), {dt:.1f} s"}


if __name__ == "__main__":
    main()

"""Self-launch of the one-process-per-GPU ranks (utterances shard over the
GPUs of one node, one all-reduce of the statistics per EM iteration).

`python bench.py --gpus N` with no launcher environment has to start its N ranks itself.  That
is done HERE and only with child processes: the parent never touches the GPU (no HIP call, no
`torch.cuda.is_available()`), it starts `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 --master-port P <script> <args>` as a child, passes
the child's stdout through and exits with its return code.  Nothing is ever exec'ed from a
process that has initialised the GPU.

The reference has no parallelism (single-threaded C, TF:238-358); what is sharded is its
accumulator loop over utterances (TF:1614, 1618, 1660, 1716-1722, 318-320).
"""
import os
import socket
import subprocess
import sys

LAUNCH_ENV = ("RANK", "WORLD_SIZE", "LOCAL_RANK")


def under_launcher(env=None):
    """True when a launcher (torch.distributed.run) has already placed this process as a rank."""
    env = os.environ if env is None else env
    return all(k in env for k in LAUNCH_ENV)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_command(script, argv, nproc, port, python=None):
    """The driver's own command form for N > 1 (task contract), as an argv list."""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
            f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), script] + list(argv)


def self_launch(script, argv, nproc, env=None, timeout=None, stdout=None):
    """Start `nproc` ranks of `script argv` as children of this (GPU-free) process and wait.

    The children's stdout is passed through line by line (rank 0 prints the ONE JSON line),
    stderr is inherited.  Returns the launcher's return code: non-zero if any rank failed
    (torch.distributed.run tears the other ranks down and reports the failure)."""
    e = dict(os.environ if env is None else env)
    for k in LAUNCH_ENV + ("MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK"):
        e.pop(k, None)
    # the host driver only supports dmabuf IPC: RCCL between processes needs this
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e.setdefault("OMP_NUM_THREADS", "1")
    out = sys.stdout if stdout is None else stdout
    cmd = launch_command(script, argv, nproc, free_port())
    p = subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE, text=True, bufsize=1)
    try:
        for line in p.stdout:
            out.write(line)
            out.flush()
        return p.wait(timeout=timeout)
    except BaseException:
        # our own child, by PID: never by pattern
        p.kill()
        p.wait()
        raise

"""The N > 1 path on CPU: two gloo ranks, utterances sharded by rank, one all-reduce
of the flat statistics vector per EM iteration, redundant M-step — driven by the
same EMDriver bench.py uses, with the oracle standing in for the HIP backend (tests
may call the oracle; the product never does)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
from _load import load_pkg


class OracleBackend:
    def __init__(self, hm, X, lens):
        self.hm, self.X, self.lens = hm, X, lens
        self.t = torch.zeros(O.stats_len(hm.N, hm.M, hm.D), dtype=torch.float64)

    def estep(self):
        s, _ = O.estep(self.hm, self.X, self.lens, dumps=False)
        self.t.copy_(torch.from_numpy(s))

    def stats_tensor(self):
        return self.t

    def mstep(self):
        self.hm = O.mstep(self.hm, self.t.numpy())

    def loglik(self):
        return float(self.t[-2])


def _problem(G):
    N, M, D = 5, 2, 7
    mean, std = G.synth_truth(N, M, D)
    lens = np.array([40, 55, 32, 61, 47, 38, 52], dtype=np.int32)
    X = G.synth_utterances(mean, std, lens)
    return G.synth_start_model(mean, std, 0.1), X, lens


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = load_pkg()
    G, em = pkg.ghmm, pkg.em
    hm, X, lens = _problem(G)
    # length-balanced shards, the split bench.py and the C trainer use
    idx = em.shard_balanced(lens, rank, world)
    off = np.concatenate([[0], np.cumsum(lens)])
    be = OracleBackend(hm, np.concatenate([X[off[u]:off[u + 1]] for u in idx]), lens[idx])
    drv = em.EMDriver(be, dist)
    for _ in range(3):
        drv.step()
    it, p = drv.train(threshold=1e-3, max_iter=50)
    q.put((rank, [a.copy() for a in be.hm.arrays()], it, p, float(be.t[-1])))
    dist.destroy_process_group()


def test_two_ranks_equal_one(G):
    pkg = load_pkg()
    em = pkg.em
    assert [em.shard_range(7, r, 2) for r in range(2)] == [(0, 4), (4, 7)]
    assert [em.shard_range(8, r, 3) for r in range(3)] == [(0, 3), (3, 6), (6, 8)]
    assert [em.shard_balanced([40, 55, 32, 61, 47, 38, 52], r, 2) for r in range(2)] == \
        [[0, 2, 3, 6], [1, 4, 5]]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process, whole corpus
    hm, X, lens = _problem(G)
    be = OracleBackend(hm, X, lens)
    drv = em.EMDriver(be)
    for _ in range(3):
        drv.step()
    it, p = drv.train(threshold=1e-3, max_iter=50)
    for rank, arrays, it_r, p_r, n_utt in res:
        assert it_r == it and n_utt == len(lens)
        assert abs(p_r - p) <= 1e-12 * abs(p)
        for a, b in zip(arrays, be.hm.arrays()):
            assert np.allclose(a, b, rtol=1e-10, atol=0)
    # both ranks hold the same model bit for bit (same reduced statistics, same M-step)
    for a, b in zip(res[0][1], res[1][1]):
        assert np.array_equal(a, b)


# ------------------------------------------------------------------ self-launch of the ranks

_RANK_SCRIPT = """\
import json, os, sys
r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
fail = len(sys.argv) > 2 and int(sys.argv[2]) == r
if r == 0:
    print(json.dumps({"n_gpus": w, "argv": sys.argv[1:], "local": os.environ["LOCAL_RANK"],
                      "master": os.environ["MASTER_ADDR"]}), flush=True)
sys.exit(3 if fail else 0)
"""


def test_self_launch_starts_the_ranks(tmp_path):
    """`python bench.py --gpus N` without a launcher environment: launch.py
    starts N child ranks through torch.distributed.run on 127.0.0.1, passes rank 0's one line
    through and returns the children's code; a failing rank makes it non-zero; an inherited
    launcher environment does not leak into the children."""
    import io
    import json
    L = load_pkg().launch
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    assert not L.under_launcher({})
    assert L.under_launcher({"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
    cmd = L.launch_command("bench.py", ["--gpus", "4"], 4, 1234, python="python3")
    assert cmd[:3] == ["python3", "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-3:] == ["bench.py", "--gpus", "4"]
    out = io.StringIO()
    env = dict(os.environ, RANK="7", WORLD_SIZE="9", LOCAL_RANK="7", MASTER_PORT="1")
    rc = L.self_launch(str(script), ["--gpus"], 3, env=env, stdout=out)
    assert rc == 0
    lines = [ln for ln in out.getvalue().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec == {"n_gpus": 3, "argv": ["--gpus"], "local": "0", "master": "127.0.0.1"}
    out = io.StringIO()
    assert L.self_launch(str(script), ["x", "1"], 2, stdout=out) != 0


def test_bench_self_launches_when_gpus_exceeds_one(tmp_path):
    """bench.py itself: with --gpus 2 and no launcher environment it must become the launcher
    (and not raise "launch with torch.distributed.run" as in round 2).  There is no GPU here, so
    the ranks stop at bench.py's own "needs an MI355X" exit: what is checked is that they were
    started as 2 ranks (that message, rc != 0, no round-2 message)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = ""       # the same outcome on a box that has a GPU
    env["CUDA_VISIBLE_DEVICES"] = ""
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1",
                        "--warmup", "0", "--spinup", "0"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert p.returncode != 0
    assert "needs an MI355X" in p.stderr and "launch with torch.distributed.run" not in p.stderr

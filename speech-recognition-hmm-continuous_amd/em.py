"""Host-side Baum-Welch driver over the C ABI: utterance sharding and the one
all-reduce of sufficient statistics per EM iteration.

The reference has no parallelism of any kind; what is mirrored here is its EM loop
(TF:238-358): zero the accumulators, E-step over every utterance, M-step.  Across
ranks the accumulators are plain sums over utterances (TF:1614, 1618, 1660,
1716-1722, TF:318-320), so each rank runs the E-step on its own utterances, the flat
statistics vector is summed over ranks (RCCL all-reduce on GPUs, gloo in CPU tests)
and every rank applies the same M-step redundantly — no broadcast.
"""
import math


def shard_range(n_utt, rank, world):
    """Contiguous block of utterances [lo, hi) of `rank`; sizes differ by at most one."""
    q, r = divmod(n_utt, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_balanced(lens, rank, world):
    """Length-balanced shard (sort by T, deal in turn): utterances by
    decreasing length (ties: lower index first), rank r takes every world-th one starting
    at r; returned in ascending index order.  Frames per rank differ by at most the longest
    utterance.  Same result as the C ABI's ghmm_shard_balanced (what the C trainer uses)."""
    order = sorted(range(len(lens)), key=lambda u: (-int(lens[u]), u))
    return sorted(order[rank::world])


class EMDriver:
    """One EM iteration = backend.estep -> all_reduce(stats) -> backend.mstep.

    `backend` provides estep() (fills the statistics buffer), stats_tensor() (a
    torch tensor aliasing that buffer, or None when world == 1), mstep() and
    loglik() (sum of log P over ALL ranks' utterances after the all-reduce)."""

    def __init__(self, backend, dist=None):
        self.backend = backend
        self.dist = dist if (dist is not None and dist.is_initialized()
                             and dist.get_world_size() > 1) else None

    def step(self):
        self.backend.estep()
        if self.dist is not None:
            self.dist.all_reduce(self.backend.stats_tensor(), op=self.dist.ReduceOp.SUM)
        self.backend.mstep()

    def train(self, threshold=1e-3, max_iter=1000):
        """The reference's convergence rule (TF:325-358): old = 1.0, M-step only while
        the relative change of the total log-likelihood exceeds `threshold`."""
        old, it = 1.0, 0
        while True:
            it += 1
            self.backend.estep()
            if self.dist is not None:
                self.dist.all_reduce(self.backend.stats_tensor(), op=self.dist.ReduceOp.SUM)
            p = self.backend.loglik()
            var = abs((old - p) / old)
            if var > threshold and not math.isnan(var):
                old = p
                self.backend.mstep()
            if not (var > threshold) or it >= max_iter:
                return it, p


class HipBackend:
    """The product path: HIP kernels through the C ABI, statistics in a torch CUDA
    tensor so that torch.distributed (RCCL) reduces them in place."""

    def __init__(self, G, ctx, model, corpus, torch=None):
        self.ctx, self.model, self.corpus = ctx, model, corpus
        n = G.stats_len(model.N, model.M, model.D)
        self._t = None
        if torch is not None:
            self._t = torch.zeros(n, dtype=torch.float64, device=f"cuda:{torch.cuda.current_device()}")
            self.stats = ctx.stats(model.N, model.M, model.D, dev_ptr=self._t.data_ptr())
        else:
            self.stats = ctx.stats(model.N, model.M, model.D)

    def estep(self):
        self.ctx.estep(self.model, self.corpus, self.stats)

    def mstep(self):
        self.ctx.mstep(self.model, self.stats)

    def stats_tensor(self):
        return self._t

    def loglik(self):
        return self.stats.loglik()[0]

"""Row-sharded FISTA over torch.distributed (backend "nccl" = RCCL over xGMI on ROCm), one process per GPU.

The reference is single-process NumPy; this module is new.  The gradient decomposes over row blocks,
    A^T (A y - b) = sum_p A_p^T (A_p y - b_p),
so each rank runs the single-pass kernel on its shard and ONE all-reduce(SUM) of n+1 floats
([partial gradient ; partial ||r||^2], 64 KiB at n = 16384) is the only exchange per iteration.  It is
latency-bound (tens of microseconds against ~1 ms of streaming per rank), so a plain RCCL all-reduce is used.
alpha2*y and the prox are applied after the reduction (adding alpha2*y per shard would count it P times).
x_k, x_{k-1} and the momentum scalars are replicated; every rank applies the identical update to the identical
reduced gradient, so the replicas stay bit-identical without any further traffic.

The class is written against a small "engine" interface (grad / gbuf / update / x) so the collective
choreography is tested on CPU with gloo and a stand-in engine (tests/test_distributed_cpu.py); the product
engine below drives the HIP kernels and has no CPU fallback.
"""
import math

import torch.distributed as dist

from . import _core


def shard_rows(m, world, rank):
    """Contiguous row range [lo, hi) of rank `rank`; the first m % world ranks get one extra row."""
    base, extra = divmod(int(m), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class HipShardEngine:
    """The product engine: this rank's rows on this rank's GPU."""

    def __init__(self, A_shard, b_shard, dtype=None):
        self.prob = _core.Problem(A_shard, b_shard, dtype)
        self.n = self.prob.n_dev             # device length: gbuf[n_dev] carries the partial ||r||^2
        self.st = _core.Fista(self.prob)
        self.gbuf = self.prob.gbuf            # torch tensor (n + 4 floats) the kernels write / read

    def reset(self, **kw):
        self.st.reset(**kw)

    def grad(self):
        self.st.grad()

    def update(self):
        self.st.update()

    def x(self):
        return self.st.x_tensor()

    def status(self):
        return self.st.status()


class ShardedFista:
    """FISTA / FISTA-delta / ISTA with A row-sharded over the ranks of `group`."""

    def __init__(self, engine, group=None):
        self.engine = engine
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def step(self):
        e = self.engine
        e.grad()                                                  # partial A_p^T (A_p y - b_p), partial rr
        if self.world > 1:
            dist.all_reduce(e.gbuf[: e.n + 1], op=dist.ReduceOp.SUM, group=self.group)
        e.update()                                                # identical on every rank

    def run(self, iters):
        for _ in range(int(iters)):
            self.step()

    def x(self):
        return self.engine.x()


class HipVecOps:
    """n-vector norm / scale on the device (fos_vec_stats / fos_vec_axpby)."""

    @staticmethod
    def norm(v):
        from .operators import vec_stats
        return math.sqrt(vec_stats(None, None, v)[2])

    @staticmethod
    def scale(v, a):
        from .operators import vec_axpby
        return vec_axpby(a, v, 0.0, None)


def sharded_lipschitz(matvec, n, v0, n_iter=100, tol=1e-6, group=None, ops=HipVecOps):
    """Power iteration (iterative_solvers.py:45-60) with w = sum_p A_p^T (A_p v) all-reduced.

    `matvec(v) -> w_partial` works on tensors of the engine's device; v0 must be the SAME on all ranks (draw it
    from the same seeded global NumPy stream on every rank, as the single-process path does).  `ops` supplies
    the n-vector norm/scale (HIP kernels in the product; the CPU gloo test injects a stand-in)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    v = ops.scale(v0, 1.0 / ops.norm(v0))
    prev, L = 0.0, None
    for _ in range(n_iter):
        w = matvec(v)
        if world > 1:
            dist.all_reduce(w, op=dist.ReduceOp.SUM, group=group)
        L = ops.norm(w)
        v = ops.scale(w, 1.0 / L)
        if abs(L - prev) < tol:
            break
        prev = L
    return L

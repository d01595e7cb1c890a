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
import ctypes as C
import math

import torch
import torch.distributed as dist

from . import _core, _lib


def shard_rows(m, world, rank):
    """Contiguous row range [lo, hi) of rank `rank`; the first m % world ranks get one extra row."""
    base, extra = divmod(int(m), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class Comm:
    """Communicator under the C ABI (fos_comm_*: RCCL over xGMI): attached to a Problem it makes every row-sum of that
    problem a sum over the ranks, on the problem's stream - sharded runs then enqueue like single-GPU ones.

    ``Comm(group)`` is collective over the ranks of a torch.distributed group (any backend: it only carries the 128-byte
    id from rank 0 to the others); the current device must be this rank's GPU.  ``Comm.solo()`` is a one-rank
    communicator that needs no process group (the all-reduce path on a single GPU: tests, rehearsals)."""

    def __init__(self, group=None, _solo=False, transport="rccl", cap_bytes=1 << 18):
        """transport "rccl" (default) or "mesh": the one-shot full-mesh kernel (fos_comm_mesh_*: every rank writes its
        vector into an IPC-mapped inbox on every peer and sums the rows in rank order - one launch, one link latency,
        for messages up to cap_bytes; needs all ranks on one node, at most 8)."""
        self.lib = _lib.load()
        _core.require_gpu()
        if _solo:
            world, rank = 1, 0
        else:
            world, rank = dist.get_world_size(group), dist.get_rank(group)
        self.kind = transport
        h = C.c_void_p()
        if transport == "mesh":
            mine = C.create_string_buffer(128)
            _lib.check(self.lib.fos_comm_mesh_create(C.byref(h), world, rank, int(cap_bytes), mine), "fos_comm_mesh_create")
            handles = [None] * world
            if world > 1:
                dist.all_gather_object(handles, mine.raw, group=group)
            else:
                handles[0] = mine.raw
            _lib.check(self.lib.fos_comm_mesh_connect(h, b"".join(handles)), "fos_comm_mesh_connect")
            if world > 1:
                dist.barrier(group=group)            # every inbox is mapped everywhere before the first push
            self.h, self.world, self.rank = h, world, rank
            return
        if transport != "rccl":
            raise ValueError("transport must be 'rccl' or 'mesh'")
        ids = [None]
        if rank == 0:
            buf = C.create_string_buffer(128)
            _lib.check(self.lib.fos_comm_unique_id(buf), "fos_comm_unique_id")
            ids[0] = buf.raw
        if world > 1:
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(ids, src=src, group=group)
        _lib.check(self.lib.fos_comm_create(C.byref(h), ids[0], world, rank), "fos_comm_create")
        self.h, self.world, self.rank = h, world, rank

    @classmethod
    def solo(cls, transport="rccl"):
        return cls(_solo=True, transport=transport)

    def transport(self):
        if self.kind == "mesh":
            fine, cap = C.c_int32(), C.c_int64()
            _lib.check(self.lib.fos_comm_mesh_info(self.h, C.byref(fine), C.byref(cap)), "fos_comm_mesh_info")
            return (f"mesh: one-shot full-mesh kernel over IPC-mapped {'fine-grained' if fine.value else 'COARSE-GRAINED'} "
                    f"inboxes of {cap.value} bytes per source")
        return self.lib.fos_comm_transport().decode()

    def check(self):
        """Raise if a mesh all-reduce timed out waiting for a peer (synchronises); no-op for RCCL."""
        _lib.check(self.lib.fos_comm_check(self.h, _core.stream_ptr()), "fos_comm_check")

    def allreduce(self, t):
        """In-place sum of a float32 / float64 device tensor over the ranks, on the current stream."""
        assert t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, torch.float64)
        _lib.check(self.lib.fos_comm_allreduce(self.h, _core.ptr(t), t.numel(), int(t.dtype == torch.float64),
                                               _core.stream_ptr()), "fos_comm_allreduce")
        return t

    def __del__(self):
        h = getattr(self, "h", None)
        if h:
            try:
                self.lib.fos_comm_destroy(h)
            except Exception:
                pass
            self.h = None


class HipShardEngine:
    """The product engine: this rank's rows on this rank's GPU.

    ``comm``: a `Comm` -> the all-reduce happens under the C ABI, on the problem's stream (`run(k)` enqueues k whole
    iterations).  Without it the engine exposes the split step (grad / gbuf / update) and `ShardedFista` moves gbuf
    through torch.distributed between the two halves (any backend; what the gloo tests use).
    Columns are always padded to the kernel granularity here (pad=True), so every rank derives the same device length
    n_dev from n alone, whatever its local row count or pointer alignment - the all-reduce count must agree."""

    def __init__(self, A_shard, b_shard, dtype=None, comm=None, group=None):
        self.prob = _core.Problem(A_shard, b_shard, dtype, pad=True)
        self.n = self.prob.n_dev             # device length: gbuf[n_dev] carries the partial ||r||^2
        self.comm = comm
        if comm is not None:
            self.prob.set_comm(comm)
        if group is not None and dist.is_initialized() and dist.get_world_size(group) > 1:
            probe = torch.tensor([self.n, -self.n], dtype=torch.int64,
                                 device=self.prob.device if dist.get_backend(group) == "nccl" else "cpu")
            dist.all_reduce(probe, op=dist.ReduceOp.MAX, group=group)
            if int(probe[0]) != self.n or int(-probe[1]) != self.n:
                raise ValueError("row shards disagree on the device length of the column dimension")
        self.st = _core.Fista(self.prob)
        self.gbuf = self.prob.gbuf            # torch tensor (n + 4 floats) the kernels write / read

    def reset(self, **kw):
        self.st.reset(**kw)

    def run(self, iters):
        """comm attached: `iters` whole sharded iterations, enqueue-only (fos_fista_run)."""
        self.st.run(iters)

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
        if getattr(e, "comm", None) is not None:                  # exchange under the C ABI, on the kernels' stream
            e.run(1)
            return
        e.grad()                                                  # partial A_p^T (A_p y - b_p), partial rr
        if self.world > 1:
            dist.all_reduce(e.gbuf[: e.n + 1], op=dist.ReduceOp.SUM, group=self.group)
        e.update()                                                # identical on every rank

    def run(self, iters):
        if getattr(self.engine, "comm", None) is not None:
            self.engine.run(int(iters))
            return
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

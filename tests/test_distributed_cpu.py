"""CPU, world_size 2, gloo: the row-sharded FISTA choreography of fastoptsolver_amd.distributed with a stand-in
engine built on the oracle (the HIP engine needs a GPU).  Checks: every rank ends with bit-identical iterates,
and they equal the unsharded oracle; sharded power iteration equals the unsharded one."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from fastoptsolver_amd import distributed as fd   # noqa: E402
from oracle import fos_oracle as orc              # noqa: E402
from tests import _data                           # noqa: E402


class OracleShardEngine:
    """Same interface as HipShardEngine; arithmetic from the oracle (test infrastructure only)."""

    def __init__(self, A, b, tau, a1, a2):
        self.A, self.b = A, b
        self.n = A.shape[1]
        self.tau, self.a1, self.a2 = tau, a1, a2
        self.xk = np.zeros(self.n)
        self.xp = np.zeros(self.n)
        self.t, self.beta = 1.0, 0.0
        self.gbuf = torch.zeros(self.n + 4, dtype=torch.float64)

    def _y(self):
        return self.xk + self.beta * (self.xk - self.xp)

    def grad(self):
        g, rr = orc.gram_gradient(self.A, self._y(), self.b, 0.0)     # WITHOUT alpha2*y, like fos_fista_grad
        self.gbuf[: self.n] = torch.from_numpy(g)
        self.gbuf[self.n] = rr

    def update(self):
        y = self._y()
        g = self.gbuf[: self.n].numpy() + (self.a2 * y if self.a2 > 0 else 0.0)
        v = y - self.tau * g
        xn = orc.prox_l1(v, self.tau * self.a1) if self.a1 > 0 else v
        t_new = 0.5 * (1.0 + np.sqrt(1.0 + 4.0 * self.t ** 2))
        self.beta = (self.t - 1.0) / t_new
        self.t = t_new
        self.xp, self.xk = self.xk, xn

    def x(self):
        return self.xk


class NumpyVecOps:
    @staticmethod
    def norm(v):
        return float(torch.linalg.norm(v))

    @staticmethod
    def scale(v, a):
        return v * a


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    A, b, _ = _data.synth(1001, 48, 9)                       # ragged: 1001 rows over 2 ranks
    lam = float(np.max(np.abs(A.T @ b)))
    a1, a2 = 0.05 * lam, 0.5
    lo, hi = fd.shard_rows(A.shape[0], world, rank)
    As = torch.from_numpy(A[lo:hi])
    np.random.seed(0)
    v0 = torch.from_numpy(np.random.randn(48))
    L = fd.sharded_lipschitz(lambda v: As.T @ (As @ v), 48, v0, ops=NumpyVecOps)
    eng = OracleShardEngine(A[lo:hi], b[lo:hi], 1.0 / (L + a2), a1, a2)
    solver = fd.ShardedFista(eng)
    solver.run(40)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), x=solver.x(), L=L, rr=float(eng.gbuf[48]))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_rows_cover_everything():
    for m in (1, 7, 1000, 65536, 2 ** 20):
        for world in (1, 2, 3, 4, 8):
            edges = [fd.shard_rows(m, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == m
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_two_rank_gloo_sharded_fista(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["x"], r1["x"]), "replicated iterates drifted apart"
    assert float(r0["L"]) == float(r1["L"])
    A, b, _ = _data.synth(1001, 48, 9)
    lam = float(np.max(np.abs(A.T @ b)))
    np.random.seed(0)
    v0 = np.random.randn(48)
    L_ref = orc.estimate_lipschitz(A, v0=v0)
    assert float(r0["L"]) == pytest.approx(L_ref, rel=1e-12)
    x_ref = orc.fista(A, b, "elasticnet", 0.05 * lam, 0.5, max_iter=40, L=L_ref)
    assert _data.rel(r0["x"], x_ref) < 1e-10


# --------------------------------------------------------------------------------------------------
# row-sharded L-BFGS: one all-reduce per fg, the real LBFGSSolver.fit with stand-in primitives
# --------------------------------------------------------------------------------------------------
class _NoTimer:
    def start(self):
        return None

    def stop(self, ev, count=1):
        pass

    def flush(self):
        pass


class OracleLbfgsOps:
    """Same interface as fastoptsolver_amd.lbfgs._HipOps; float64 torch CPU tensors, arithmetic from the oracle."""

    def __init__(self, A, b):
        self.A, self.b, self.n = A, b, A.shape[1]
        self.rr = torch.zeros(1, dtype=torch.float64)

    def timer(self, sink):
        return _NoTimer()

    def comm_buffer(self):
        return torch.zeros(self.n + 1, dtype=torch.float64)

    def new_x(self):
        return torch.zeros(self.n, dtype=torch.float64)

    def new_g(self):
        return torch.empty(self.n, dtype=torch.float64)

    def new_history(self, cap):
        return torch.zeros(cap, self.n, dtype=torch.float64), torch.zeros(cap, self.n, dtype=torch.float64)

    def grad(self, x, a2, g):
        gg, rr = orc.gram_gradient(self.A, x.numpy(), self.b, a2)
        g.copy_(torch.from_numpy(gg))
        self.rr[0] = rr

    def stats(self, x, g, d):
        z = torch.zeros(self.n, dtype=torch.float64)
        xx, dd_ = (x if x is not None else z), (d if d is not None else z)
        return [float(xx @ xx), float(g @ dd_), float(dd_ @ dd_), float(g.abs().max()), float(xx.abs().sum()),
                float(self.rr[0])]

    def direction(self, g, S, Y, hist, head):
        cap = S.shape[0]
        idx = [(head + i) % cap for i in range(hist)]
        return torch.from_numpy(orc.two_loop_direction(g.numpy(), [S[i].numpy() for i in idx], [Y[i].numpy() for i in idx]))

    def step_to(self, x_old, stp, d):
        return x_old + stp * d

    def store_pair(self, S, Y, slot, stp, d, g, g_old):
        S[slot] = stp * d
        Y[slot] = g - g_old

    def to_caller(self, x):
        return x.numpy().copy()


def _lbfgs_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fastoptsolver_amd.lbfgs import LBFGSSolver
    A, b, _ = _data.synth(1001, 48, 9)
    lo, hi = fd.shard_rows(A.shape[0], world, rank)
    s = LBFGSSolver("elasticnet", 3.0, 0.7).fit(None, None, group=dist.group.WORLD, ops=OracleLbfgsOps(A[lo:hi], b[lo:hi]))
    np.savez(os.path.join(out_dir, f"lbfgs{rank}.npz"), x=s.x_, f=s.final_obj_, nit=s.nit_, nfev=s.nfev_,
             hist=np.asarray(s.history_))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_sharded_lbfgs(tmp_path):
    world = 2
    mp.spawn(_lbfgs_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "lbfgs0.npz"), np.load(tmp_path / "lbfgs1.npz")
    assert np.array_equal(r0["x"], r1["x"]) and int(r0["nfev"]) == int(r1["nfev"]), "replicas drifted apart"
    A, b, _ = _data.synth(1001, 48, 9)
    ref = orc.LBFGSSolver("elasticnet", 3.0, 0.7).fit(A, b)            # unsharded oracle (alpha2*x counted once)
    assert _data.rel(r0["x"], ref.x_) < 1e-8
    assert float(r0["f"]) == pytest.approx(ref.final_obj_, rel=1e-10)
    assert abs(int(r0["nit"]) - len(ref.history_)) <= 1
    k = min(len(r0["hist"]), len(ref.history_))
    assert np.allclose(r0["hist"][:k], ref.history_[:k], rtol=1e-9)    # callback objective incl. alpha1*||x||_1

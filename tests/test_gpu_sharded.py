"""The product's sharded path on the GPU (fastoptsolver_amd.distributed.HipShardEngine / ShardedFista, and
LBFGSSolver.fit(group=...)): the split step in one process, and two ranks sharing the box's single GPU with gloo as
the transport (the 8-GPU RCCL run is the driver's; the kernels, buffers and choreography are the same)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import fos_oracle as orc  # noqa: E402
from tests import _data  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 1e-5
SHAPE = (2001, 512, 5)          # rows (ragged over 2 ranks), columns, seed


def _weights(A, b):
    lam = float(np.max(np.abs(A.T @ b)))
    return 0.05 * lam, 0.5


def test_split_step_single_process_matches_oracle():
    """grad() -> (no exchange) -> update(): the choreography of the sharded run with world size 1."""
    from fastoptsolver_amd import distributed as fd
    A, b, _ = _data.synth(*SHAPE)
    a1, a2 = _weights(A, b)
    L = float(np.linalg.norm(A, 2) ** 2)
    eng = fd.HipShardEngine(A.astype(np.float32), b.astype(np.float32))
    eng.reset(tau=1.0 / (L + a2), alpha1=a1, alpha2=a2)
    fd.ShardedFista(eng).run(40)
    x_ref = orc.fista(A, b, "elasticnet", a1, a2, max_iter=40, L=L)
    assert _data.rel(eng.x().cpu().numpy(), x_ref) < TOL
    assert int(eng.status().k) == 40


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import fastoptsolver_amd as fos
    from fastoptsolver_amd import distributed as fd
    dev = torch.device("cuda", 0)
    A, b, _ = _data.synth(*SHAPE)
    a1, a2 = _weights(A, b)
    lo, hi = fd.shard_rows(A.shape[0], world, rank)
    As = torch.from_numpy(A[lo:hi].astype(np.float32)).to(dev)
    bs = torch.from_numpy(b[lo:hi].astype(np.float32)).to(dev)
    eng = fd.HipShardEngine(As, bs)
    matvec = fos.prepare(As, None)
    np.random.seed(0)
    v0 = torch.from_numpy(np.random.randn(A.shape[1]).astype(np.float32)).to(dev)
    L = fd.sharded_lipschitz(lambda v: matvec.gemv_pair(v, 0.0), A.shape[1], v0)
    eng.reset(tau=1.0 / (L + a2), alpha1=a1, alpha2=a2)
    fd.ShardedFista(eng).run(40)
    s = fos.LBFGSSolver("ridge", 0.0, a2).fit(As, bs, group=dist.group.WORLD)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), x=eng.x().cpu().numpy(), L=L, xl=np.asarray(s.x_.cpu() if torch.is_tensor(s.x_) else s.x_),
             fl=s.final_obj_, nfev=s.nfev_)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_share_the_gpu(tmp_path):
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    assert np.array_equal(r0["x"], r1["x"]) and np.array_equal(r0["xl"], r1["xl"]), "replicas drifted apart"
    assert float(r0["L"]) == float(r1["L"]) and int(r0["nfev"]) == int(r1["nfev"])
    A, b, _ = _data.synth(*SHAPE)
    a1, a2 = _weights(A, b)
    np.random.seed(0)
    L_ref = orc.estimate_lipschitz(A, v0=np.random.randn(A.shape[1]))
    assert float(r0["L"]) == pytest.approx(L_ref, rel=TOL)
    x_ref = orc.fista(A, b, "elasticnet", a1, a2, max_iter=40, L=float(r0["L"]))
    assert _data.rel(r0["x"], x_ref) < TOL
    ref = orc.LBFGSSolver("ridge", 0.0, a2).fit(A, b)
    assert _data.rel(r0["xl"], ref.x_) < TOL and int(r0["nfev"]) == ref.nfev_
    assert float(r0["fl"]) == pytest.approx(ref.final_obj_, rel=1e-6)

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


# --------------------------------------------------------------------------------------------------
# RCCL under the C ABI: a one-rank communicator forces every exchange through ncclAllReduce on the kernels' stream
# --------------------------------------------------------------------------------------------------
def test_rccl_communicator_on_the_kernels_stream_matches_oracle():
    """Comm.solo() = ncclCommInitRank with one rank: the sharded code path end to end on one GPU - fos_fista_run
    enqueues K2 -> slab reduce -> ncclAllReduce(n + 1 floats) -> prox/momentum per iteration with no host step; the
    Armijo trials, the history residual, the power iteration and the L-BFGS fg all-reduce their sums the same way.
    Everything must equal the unsharded oracle (a sum over one rank is the identity)."""
    import fastoptsolver_amd as fos
    from fastoptsolver_amd import distributed as fd
    comm = fd.Comm.solo()
    assert comm.transport().startswith("rccl"), comm.transport()
    t = torch.arange(5, dtype=torch.float64, device="cuda")
    assert torch.equal(comm.allreduce(t.clone()), t)
    A, b, _ = _data.synth(*SHAPE)
    a1, a2 = _weights(A, b)
    np.random.seed(0)
    v0 = np.random.randn(A.shape[1])
    L_ref = orc.estimate_lipschitz(A, v0=v0)
    A32, b32 = A.astype(np.float32), b.astype(np.float32)
    # (1) enqueue-only sharded run through the engine
    eng = fd.HipShardEngine(A32, b32, comm=comm)
    assert eng.prob.plan()["resident"] == 0
    eng.reset(tau=1.0 / (L_ref + a2), alpha1=a1, alpha2=a2)
    fd.ShardedFista(eng).run(40)
    x_ref = orc.fista(A, b, "elasticnet", a1, a2, max_iter=40, L=L_ref)
    assert _data.rel(eng.x().cpu().numpy(), x_ref) < TOL and int(eng.status().k) == 40
    # (2) the reference's front-end with every flag, on a shard problem with the communicator attached
    np.random.seed(0)
    assert fos.estimate_lipschitz(eng.prob) == pytest.approx(L_ref, rel=TOL)
    for kw in (dict(backtracking=True, t_init_factor=2.0), dict(adaptive_restart=True), dict(tol=1e-3), dict(tol_ratio=0.9)):
        x, h = fos.fista(A32, b32, "elasticnet", a1, a2, max_iter=60, L=L_ref, return_history=True, comm=comm, **kw)
        x_ref, h_ref = orc.fista(A, b, "elasticnet", a1, a2, max_iter=60, L=L_ref, return_history=True, **kw)
        assert len(h["obj"]) == len(h_ref["obj"]), kw
        assert _data.rel(x, x_ref) < TOL and np.allclose(h["obj"], h_ref["obj"], rtol=TOL), kw
    x = fos.fista_delta(A32, b32, "lasso", a1, 0.0, 3.0, max_iter=50, L=L_ref, comm=comm)
    assert _data.rel(x, orc.fista_delta(A, b, "lasso", a1, 0.0, 3.0, max_iter=50, L=L_ref)) < TOL
    # (3) L-BFGS: fg's n + 1 doubles through the communicator
    s = fos.LBFGSSolver("ridge", 0.0, a2).fit(A32, b32, comm=comm)
    ref = orc.LBFGSSolver("ridge", 0.0, a2).fit(A32.astype(np.float64), b32.astype(np.float64))
    assert (s.nit_, s.nfev_) == (ref.nit_, ref.nfev_) and _data.rel(s.x_, ref.x_) < TOL
    # (4) bf16 shard (config 5 in miniature) through the same path
    A16 = torch.as_tensor(A32).to(torch.bfloat16).cuda()
    e16 = fd.HipShardEngine(A16, b32, comm=comm)
    e16.reset(tau=1.0 / (L_ref + 10.0), alpha1=a1, alpha2=10.0)
    e16.run(30)
    x_ref = orc.fista(A16.to(torch.float64).cpu().numpy(), b32.astype(np.float64), "elasticnet", a1, 10.0, max_iter=30,
                      L=L_ref)
    assert _data.rel(e16.x().cpu().numpy(), x_ref) < TOL


# --------------------------------------------------------------------------------------------------
# two ranks, split form over gloo: every flag of the reference's loop on a row-sharded problem, and config 5 in miniature
# --------------------------------------------------------------------------------------------------
CASES = [dict(), dict(backtracking=True, t_init_factor=2.0), dict(adaptive_restart=True), dict(tol=1e-3),
         dict(tol_ratio=0.9)]


def _flags_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import fastoptsolver_amd as fos
    from fastoptsolver_amd import distributed as fd
    A, b, _ = _data.synth(*SHAPE)
    a1, a2 = _weights(A, b)
    lo, hi = fd.shard_rows(A.shape[0], world, rank)
    As, bs = A[lo:hi].astype(np.float32), b[lo:hi].astype(np.float32)
    out = {}
    np.random.seed(0)
    out["L"] = fos.estimate_lipschitz(fos.prepare(As, None), group=dist.group.WORLD)
    L = out["L"]
    for i, kw in enumerate(CASES):
        x, h = fos.fista(As, bs, "elasticnet", a1, a2, max_iter=60, L=L, return_history=True, group=dist.group.WORLD, **kw)
        out[f"x{i}"], out[f"obj{i}"] = np.asarray(x), np.asarray(h["obj"])
        out[f"ls{i}"] = np.asarray(fos.get_metrics()["ls_iters_total"])
    x, h = fos.fista_delta(As, bs, "lasso", a1, 0.0, 3.0, max_iter=40, L=L, return_history=True, group=dist.group.WORLD)
    out["xd"], out["objd"] = np.asarray(x), np.asarray(h["obj"])
    # config 5 in miniature: bf16 storage, l1 + l2, row-sharded
    A16 = torch.as_tensor(As).to(torch.bfloat16).cuda()
    x5 = fos.fista(A16, bs, "elasticnet", a1, 10.0, max_iter=40, L=L, group=dist.group.WORLD)
    out["x5"] = x5.float().cpu().numpy() if torch.is_tensor(x5) else np.asarray(x5)
    eng = fd.HipShardEngine(A16, bs, group=dist.group.WORLD)
    eng.reset(tau=1.0 / (L + 10.0), alpha1=a1, alpha2=10.0)
    fd.ShardedFista(eng, dist.group.WORLD).run(40)
    out["x5e"] = eng.x().cpu().numpy()
    np.savez(os.path.join(out_dir, f"f{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_ranks_every_flag_and_config5_in_miniature(tmp_path):
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_flags_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "f0.npz"), np.load(tmp_path / "f1.npz")
    for k in r0.files:
        assert np.array_equal(r0[k], r1[k]), f"replicas drifted apart: {k}"
    A, b, _ = _data.synth(*SHAPE)
    a1, a2 = _weights(A, b)
    np.random.seed(0)
    L_ref = orc.estimate_lipschitz(A, v0=np.random.randn(A.shape[1]))
    L = float(r0["L"])
    assert L == pytest.approx(L_ref, rel=TOL)
    for i, kw in enumerate(CASES):
        x_ref, h_ref = orc.fista(A, b, "elasticnet", a1, a2, max_iter=60, L=L, return_history=True, **kw)
        assert len(r0[f"obj{i}"]) == len(h_ref["obj"]), kw              # same stopping iteration as the unsharded run
        assert _data.rel(r0[f"x{i}"], x_ref) < TOL and np.allclose(r0[f"obj{i}"], h_ref["obj"], rtol=TOL), kw
        if kw.get("backtracking"):       # split form backtracks on the fp64 gradient too (round 3: n + 1 doubles travel)
            _, met = orc.fista(A, b, "elasticnet", a1, a2, max_iter=60, L=L, return_metrics=True, **kw)
            assert abs(int(r0[f"ls{i}"]) - met["ls_iters_total"]) <= 8, (kw, int(r0[f"ls{i}"]), met["ls_iters_total"])
    x_ref, h_ref = orc.fista_delta(A, b, "lasso", a1, 0.0, 3.0, max_iter=40, L=L, return_history=True)
    assert _data.rel(r0["xd"], x_ref) < TOL and np.allclose(r0["objd"], h_ref["obj"], rtol=TOL)
    Aq = torch.as_tensor(A.astype(np.float32)).to(torch.bfloat16).to(torch.float64).numpy()
    x5_ref = orc.fista(Aq, b.astype(np.float32).astype(np.float64), "elasticnet", a1, 10.0, max_iter=40, L=L)
    assert _data.rel(r0["x5"], x5_ref) < TOL and _data.rel(r0["x5e"], x5_ref) < TOL


# --------------------------------------------------------------------------------------------------
# one-shot full-mesh all-reduce kernel (fos_comm_mesh_*): two processes, IPC-mapped inboxes, one GPU
# --------------------------------------------------------------------------------------------------
def _mesh_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import fastoptsolver_amd as fos
    from fastoptsolver_amd import distributed as fd
    comm = fd.Comm(dist.group.WORLD, transport="mesh")
    out = {}
    # (1) the kernel alone: floats and doubles, several sizes, many rounds back to back (the two inbox sets alternate)
    ok = True
    for count, dt in ((1, torch.float32), (513, torch.float32), (16385, torch.float32), (8193, torch.float64), (32768, torch.float64)):
        for rnd in range(6):
            g = torch.Generator(device="cuda").manual_seed(100 * rnd + count)
            both = [torch.randn(count, device="cuda", dtype=dt, generator=g) for _ in range(world)]   # same on every rank
            mine = both[rank].clone()
            comm.allreduce(mine)
            want = both[0].clone()
            for r in range(1, world):
                want += both[r]                                                  # rank order, like the kernel
            ok = ok and bool(torch.equal(mine, want))
    comm.check()
    out["kernel_ok"] = np.asarray(ok)
    # (2) the whole sharded stack on it: enqueue-only FISTA, every flag, L-BFGS
    A, b, _ = _data.synth(*SHAPE)
    a1, a2 = _weights(A, b)
    lo, hi = fd.shard_rows(A.shape[0], world, rank)
    As, bs = A[lo:hi].astype(np.float32), b[lo:hi].astype(np.float32)
    eng = fd.HipShardEngine(As, bs, comm=comm)
    np.random.seed(0)
    L = fos.estimate_lipschitz(eng.prob)
    out["L"] = np.asarray(L)
    eng.reset(tau=1.0 / (L + a2), alpha1=a1, alpha2=a2)
    fd.ShardedFista(eng).run(40)
    out["x"] = eng.x().cpu().numpy()
    x, h = fos.fista(As, bs, "elasticnet", a1, a2, max_iter=40, L=L, return_history=True, backtracking=True,
                     t_init_factor=2.0, comm=comm)
    out["xbt"], out["objbt"] = np.asarray(x), np.asarray(h["obj"])
    # (2b) a device-side stop inside ONE enqueue-only run of 300 iterations: ~295 no-op iterations whose all-reduces still
    # run.  The gradient buffer must keep the last active iteration's sums (an in-place all-reduce of a stale SUM would
    # double it per iteration: inf after 128) - compared with a run of exactly that many iterations.
    eng2 = fd.HipShardEngine(As, bs, comm=comm)
    eng2.reset(tau=1.0 / (L + a2), alpha1=a1, alpha2=a2)
    eng2.run(5)
    step5 = float(eng2.status().this_step)
    eng2.reset(tau=1.0 / (L + a2), alpha1=a1, alpha2=a2, tol_step=1.0001 * step5)
    eng2.run(300)
    st2 = eng2.status()
    out["stop_k"], out["stop_flag"] = np.asarray(int(st2.k)), np.asarray(int(st2.stopped))
    out["gbuf_stopped"] = eng2.gbuf[: eng2.n + 1].cpu().numpy()
    eng3 = fd.HipShardEngine(As, bs, comm=comm)
    eng3.reset(tau=1.0 / (L + a2), alpha1=a1, alpha2=a2)
    eng3.run(int(st2.k))
    out["gbuf_exact"] = eng3.gbuf[: eng3.n + 1].cpu().numpy()
    out["x_stopped"], out["x_exact"] = eng2.x().cpu().numpy(), eng3.x().cpu().numpy()
    del eng2, eng3
    s = fos.LBFGSSolver("ridge", 0.0, a2).fit(As, bs, comm=comm)
    out["xl"], out["nfev"] = np.asarray(s.x_), np.asarray(s.nfev_)
    comm.check()
    # (3) the multi-lambda pass on a row-sharded problem: 16 gradients, one all-reduce of 16 x n floats per iteration
    big = fd.Comm(dist.group.WORLD, transport="mesh", cap_bytes=8 << 20)
    lam = float(np.max(np.abs(A.T @ b)))
    alphas = [(lam * 0.4 * 0.7 ** i, 0.5 if i % 3 == 1 else 0.0) for i in range(6)]
    xs = fos.fista_path(As, bs, alphas, max_iter=25, L=L, comm=big)
    out["xpath"] = np.stack([np.asarray(x) for x in xs])
    big.check()
    np.savez(os.path.join(out_dir, f"m{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


def _mesh4_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fastoptsolver_amd import distributed as fd
    comm = fd.Comm(dist.group.WORLD, transport="mesh", cap_bytes=1 << 20)
    ok = True
    for rnd in range(8):
        for count, dt in ((7, torch.float32), (16385, torch.float32), (40000, torch.float64)):
            g = torch.Generator(device="cuda").manual_seed(31 * rnd + count)
            every = [torch.randn(count, device="cuda", dtype=dt, generator=g) for _ in range(world)]
            mine = every[rank].clone()
            comm.allreduce(mine)
            want = every[0].clone()
            for r in range(1, world):
                want += every[r]
            ok = ok and bool(torch.equal(mine, want))
    comm.check()
    np.savez(os.path.join(out_dir, f"q{rank}.npz"), ok=np.asarray(ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_one_shot_mesh_allreduce_four_processes(tmp_path):
    """Four ranks (four processes sharing the GPU): inbox rows, flag rows and the rank-order sum for P > 2."""
    import torch.multiprocessing as mp
    world = 4
    mp.spawn(_mesh4_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(bool(np.load(tmp_path / f"q{r}.npz")["ok"]) for r in range(world))


@pytest.mark.timeout(600)
def test_one_shot_mesh_allreduce_two_processes(tmp_path):
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_mesh_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "m0.npz"), np.load(tmp_path / "m1.npz")
    assert bool(r0["kernel_ok"]) and bool(r1["kernel_ok"])
    for k in r0.files:
        assert np.array_equal(r0[k], r1[k]), f"replicas drifted apart: {k}"
    A, b, _ = _data.synth(*SHAPE)
    a1, a2 = _weights(A, b)
    np.random.seed(0)
    L_ref = orc.estimate_lipschitz(A, v0=np.random.randn(A.shape[1]))
    L = float(r0["L"])
    assert L == pytest.approx(L_ref, rel=TOL)
    assert _data.rel(r0["x"], orc.fista(A, b, "elasticnet", a1, a2, max_iter=40, L=L)) < TOL
    x_ref, h_ref = orc.fista(A, b, "elasticnet", a1, a2, max_iter=40, L=L, return_history=True, backtracking=True,
                             t_init_factor=2.0)
    assert _data.rel(r0["xbt"], x_ref) < TOL and np.allclose(r0["objbt"], h_ref["obj"], rtol=TOL)
    assert int(r0["stop_flag"]) == 1 and 1 <= int(r0["stop_k"]) <= 6, (int(r0["stop_flag"]), int(r0["stop_k"]))   # FOS_STOP_STEP
    assert np.isfinite(r0["gbuf_stopped"]).all() and np.array_equal(r0["gbuf_stopped"], r0["gbuf_exact"])
    assert np.array_equal(r0["x_stopped"], r0["x_exact"])
    ref = orc.LBFGSSolver("ridge", 0.0, a2).fit(A, b)
    assert _data.rel(r0["xl"], ref.x_) < TOL and int(r0["nfev"]) == ref.nfev_
    lam = float(np.max(np.abs(A.T @ b)))
    for i in range(6):
        a1p, a2p = lam * 0.4 * 0.7 ** i, (0.5 if i % 3 == 1 else 0.0)
        assert _data.rel(r0["xpath"][i], orc.fista(A, b, "elasticnet", a1p, a2p, max_iter=25, L=L)) < TOL, i


# --------------------------------------------------------------------------------------------------
# column sharding (very wide A): x partitioned, one m-vector exchange per iteration
# --------------------------------------------------------------------------------------------------
COL_SHAPE = (600, 2048, 13)
COL_CASES = [dict(), dict(adaptive_restart=True), dict(tol=0.5), dict(tol_ratio=0.9), dict(return_history=True),
             dict(return_history=True, adaptive_restart=True, tol_ratio=0.97),
             dict(backtracking=True, t_init_factor=2.0), dict(backtracking=True, t_init_factor=1.0, return_history=True)]


COL_PATH_WEIGHTS = 6
COL_PATH_CTL = dict(adaptive_restart=True, tol_ratio=0.9)


def _cols_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import fastoptsolver_amd as fos
    from fastoptsolver_amd import distributed as fd
    comm = fd.Comm(dist.group.WORLD, transport="mesh", cap_bytes=1 << 20)
    A, b, _ = _data.synth(*COL_SHAPE)
    m, n = A.shape
    a1, a2 = _weights(A, b)
    lo, hi = fd.shard_rows(n, world, rank)                  # the same contiguous split, applied to the columns
    Ad = torch.as_tensor(A.astype(np.float32)).cuda()[:, lo:hi]          # strided view: lda = n
    prob = fos.prepare(Ad, b.astype(np.float32), pad=False)
    out = {"lohi": np.asarray([lo, hi])}
    np.random.seed(0)
    x = fos.fista(prob, None, "elasticnet", a1, a2, max_iter=5, comm=comm, cols=(lo, hi, n))      # L by power iteration
    out["x_l"] = x.cpu().numpy()
    np.random.seed(0)
    L = orc.estimate_lipschitz(A, v0=np.random.randn(n))
    for i, kw in enumerate(COL_CASES):
        res = fos.fista(prob, None, "elasticnet", a1, a2, max_iter=40, L=L, comm=comm, cols=(lo, hi, n), **kw)
        if kw.get("return_history"):
            xk, h = res
            out[f"obj{i}"] = np.asarray(h["obj"])
            out[f"xmid{i}"] = h["x"][len(h["x"]) // 2].cpu().numpy()
        else:
            xk = res
        out[f"x{i}"] = xk.cpu().numpy()
        out[f"ngrad{i}"] = np.asarray(fos.get_metrics()["grad_num_calls"])
        if kw.get("backtracking"):
            out[f"ls{i}"] = np.asarray(fos.get_metrics()["ls_iters_total"])
    xd = fos.fista_delta(prob, None, "lasso", a1, 0.0, 3.0, max_iter=30, L=L, comm=comm, cols=(lo, hi, n))
    out["xd"] = xd.cpu().numpy()
    # the fp64-accumulating pass (L-BFGS fg) column-sharded: this rank's block of the gradient, the global ||r||^2
    from fastoptsolver_amd import _core, _lib
    xg = np.random.default_rng(5).standard_normal(n) * (1.0 + 1e-9)
    xd64 = torch.as_tensor(xg[lo:hi], dtype=torch.float64, device="cuda")
    out64 = torch.zeros(hi - lo + 1, dtype=torch.float64, device="cuda")
    _lib.check(_lib.load().fos_gemv_pair_dd(prob.h, _core.ptr(xd64), 0.3, _core.ptr(out64)))
    out["dd"] = out64.cpu().numpy()
    # L-BFGS with the iterate partitioned: fg exchanges the m-vector, the direction the Gram matrices, the scalars one vector
    sl = fos.LBFGSSolver("ridge", 0.0, a2).fit(prob, None, comm=comm, cols=(lo, hi, n))
    out["xl"], out["lb_counts"] = sl.x_.cpu().numpy(), np.asarray([sl.nit_, sl.nfev_])
    out["lb_hist"] = np.asarray(sl.history_)
    # the multi-lambda lockstep column-sharded: per panel ONE exchange of the 16 residual columns between the two products
    lam = float(np.max(np.abs(A.T @ b)))
    alphas = [(lam * 0.4 * 0.7 ** i, 0.5 if i % 3 == 1 else 0.0) for i in range(COL_PATH_WEIGHTS)]
    xs = fos.fista_path(prob, None, alphas, max_iter=25, L=L, comm=comm, cols=(lo, hi, n))
    out["xpath"] = np.stack([x.cpu().numpy() for x in xs])
    xs, info = fos.fista_path(prob, None, alphas, max_iter=60, L=L, comm=comm, cols=(lo, hi, n), return_info=True,
                              **COL_PATH_CTL)
    out["xpath_ctl"], out["path_info"] = np.stack([x.cpu().numpy() for x in xs]), np.asarray(info)
    # ... and through a communicator whose inbox rows hold one 512-row panel only (two panels, the second one ragged)
    small = fd.Comm(dist.group.WORLD, transport="mesh", cap_bytes=40 << 10)
    prob_s = fos.prepare(Ad, b.astype(np.float32), pad=False)
    xs = fos.fista_path(prob_s, None, alphas[:3], max_iter=25, L=L, comm=small, cols=(lo, hi, n))
    out["xpath_small"] = np.stack([x.cpu().numpy() for x in xs])
    small.check()
    del prob_s
    refused = 0
    for call in (lambda: prob.replan(no_colblock=True), lambda: prob.power_iter(np.ones(hi - lo, np.float32), 3)):
        try:
            call()
        except _lib.FosError:
            refused += 1
    out["refused"] = np.asarray(refused)
    comm.check()
    np.savez(os.path.join(out_dir, f"c{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_column_sharding_two_processes(tmp_path):
    """A split by COLUMNS over two ranks (each: all rows, half the columns, the whole b): the concatenated blocks of x
    equal the unsharded oracle for the plain loop, adaptive restart, the three stopping rules (same stopping iteration),
    the history objective, backtracking (round 3: 16 candidates' m-vectors in one all-reduce; same shrink counts),
    FISTA-delta, the fp64 `fg` pass, L-BFGS, the multi-lambda lockstep (plain, device-controlled, two panels), and with L
    from the column-sharded power iteration."""
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_cols_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"c{k}.npz") for k in range(world)]
    A, b, _ = _data.synth(*COL_SHAPE)
    a1, a2 = _weights(A, b)
    n = A.shape[1]
    cat = lambda key: np.concatenate([r[k][key] for k in range(world)])       # noqa: E731
    assert all(int(rk["refused"]) == 2 for rk in r)          # replan / local power iteration refuse partial sums
    xg = np.random.default_rng(5).standard_normal(n) * (1.0 + 1e-9)
    A32 = A.astype(np.float32).astype(np.float64)
    g_ref, rr_ref = orc.gram_gradient(A32, xg, b.astype(np.float32).astype(np.float64), 0.3)
    assert _data.rel(np.concatenate([rk["dd"][:-1] for rk in r]), g_ref) < 1e-12
    assert all(float(rk["dd"][-1]) == pytest.approx(rr_ref, rel=1e-12) for rk in r)
    ref_l = orc.LBFGSSolver("ridge", 0.0, a2).fit(A32, b.astype(np.float32).astype(np.float64))
    assert _data.rel(cat("xl"), ref_l.x_) < TOL
    assert all(list(rk["lb_counts"]) == [ref_l.nit_, ref_l.nfev_] for rk in r), (r[0]["lb_counts"], ref_l.nit_, ref_l.nfev_)
    assert np.array_equal(r[0]["lb_hist"], r[1]["lb_hist"]) and np.allclose(r[0]["lb_hist"], ref_l.history_, rtol=1e-7)
    np.random.seed(0)
    L = orc.estimate_lipschitz(A, v0=np.random.randn(n))
    assert _data.rel(cat("x_l"), orc.fista(A, b, "elasticnet", a1, a2, max_iter=5, L=L)) < TOL
    for i, kw in enumerate(COL_CASES):
        ref = orc.fista(A, b, "elasticnet", a1, a2, max_iter=40, L=L, **kw)
        _, met = orc.fista(A, b, "elasticnet", a1, a2, max_iter=40, L=L, return_metrics=True,
                           **{k: v for k, v in kw.items() if k != "return_history"})
        x_ref = ref[0] if kw.get("return_history") else ref
        assert cat(f"x{i}").shape == (n,) and _data.rel(cat(f"x{i}"), x_ref) < TOL, kw
        assert int(r[0][f"ngrad{i}"]) == int(r[1][f"ngrad{i}"]) == met["grad_num_calls"], kw
        if kw.get("backtracking"):             # the ranks take identical Armijo decisions - the reference's, up to the
            ls = int(r[0][f"ls{i}"])           # length of its ~50-halving step-underflow event (fp64 noise regime,
            assert ls == int(r[1][f"ls{i}"]), kw                      # DESIGN.md 5; iterates and gradient counts are exact)
            assert abs(ls - met["ls_iters_total"]) <= 8, (kw, ls, met["ls_iters_total"])
        if kw.get("return_history"):
            h_ref = ref[1]
            assert np.array_equal(r[0][f"obj{i}"], r[1][f"obj{i}"]) and len(r[0][f"obj{i}"]) == len(h_ref["obj"]), kw
            assert np.allclose(r[0][f"obj{i}"], h_ref["obj"], rtol=TOL), kw
            assert _data.rel(cat(f"xmid{i}"), h_ref["x"][len(h_ref["x"]) // 2]) < TOL, kw
    assert _data.rel(cat("xd"), orc.fista_delta(A, b, "lasso", a1, 0.0, 3.0, max_iter=30, L=L)) < TOL
    # the lockstep over the column blocks: every weight equals its own unsharded oracle run - plain, with restart and the
    # ratio stop decided per weight (same stopping iteration on both ranks), and through the two-panel communicator
    lam = float(np.max(np.abs(A.T @ b)))
    alphas = [(lam * 0.4 * 0.7 ** i, 0.5 if i % 3 == 1 else 0.0) for i in range(COL_PATH_WEIGHTS)]
    assert np.array_equal(r[0]["path_info"], r[1]["path_info"])
    stops = 0
    for j, (p1, p2) in enumerate(alphas):
        x_ref = orc.fista(A, b, "elasticnet", p1, p2, max_iter=25, L=L)
        got = np.concatenate([r[k]["xpath"][j] for k in range(world)])
        assert _data.rel(got, x_ref) < TOL, j
        if j < 3:
            assert _data.rel(np.concatenate([r[k]["xpath_small"][j] for k in range(world)]), x_ref) < TOL, j
        x_ref, h_ref = orc.fista(A, b, "elasticnet", p1, p2, max_iter=60, L=L, return_history=True, **COL_PATH_CTL)
        got = np.concatenate([r[k]["xpath_ctl"][j] for k in range(world)])
        assert _data.rel(got, x_ref) < TOL, j
        assert int(r[0]["path_info"][j][0]) == len(h_ref["obj"]), (j, r[0]["path_info"][j], len(h_ref["obj"]))
        stops += int(r[0]["path_info"][j][1] != 0)
    assert stops >= 1                                       # the case exercises a masked column

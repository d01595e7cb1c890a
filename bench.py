#!/usr/bin/env python3
"""Headline benchmark: FISTA iterations/second on dense A (BASELINE.json metric), MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg4|cfg5]

Every N runs the SAME workload, cfg4: Lasso, A in R^{2^20 x 16384} fp32 - BASELINE's "iters/sec reported at 1/2/4/8
GPUs" configuration.  Rows are sharded over the N ranks (launched by `python -m torch.distributed.run --nproc-per-node
N ... bench.py --gpus N`), ONE RCCL all-reduce of n+1 floats per iteration; total work is fixed -> "scaling":
"strong"; 64 GiB fits one MI355X, so N = 1 is a real point of the series and `value` is comparable across N.
The >= 70 % single-GPU roofline target is quoted on cfg2 (A in R^{65536 x 8192} fp32): the default N = 1 run measures
that configuration too and reports it as "target_ref" (value, roofline, cpu_baseline of its own).
`--workload cfg2|cfg4|cfg5` selects one workload explicitly.

A "step" is one full FISTA iteration (gradient A^T(Ay-b) in a single pass over A, prox, momentum), inputs
resident in HBM, nothing skipped.  One JSON line is printed by rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: m, n, A dtype, alpha1 as a fraction of ||A^T b||_inf, alpha2, reg label   
    "cfg2": dict(m=65536, n=8192, dtype="f32", a1_frac=0.10, a2=0.0, reg="lasso"),
    "cfg4": dict(m=2 ** 20, n=16384, dtype="f32", a1_frac=0.10, a2=0.0, reg="lasso"),
    "cfg5": dict(m=2 ** 20, n=16384, dtype="bf16", a1_frac=0.05, a2=10.0, reg="elasticnet"),
}
BLOCK = 8192          # rows per generator block: shard boundaries of N in {1,2,4,8} fall on block edges
HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s float4-copy measured)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def make_shard(cfg, lo, hi, device, seed=0):
    """Rows [lo, hi) of the synthetic problem: A_ij ~ N(0,1), b = A x_true + 0.1 N(0,1); independent of N."""
    n = cfg["n"]
    rng = np.random.default_rng(seed)
    x_true = np.zeros(n, dtype=np.float32)
    nz = max(1, int(round(0.05 * n)))
    idx = rng.choice(n, size=nz, replace=False)
    x_true[idx] = rng.standard_normal(nz).astype(np.float32)
    xt = torch.from_numpy(x_true).to(device)
    tdt = torch.bfloat16 if cfg["dtype"] == "bf16" else torch.float32
    A = torch.empty(hi - lo, n, dtype=tdt, device=device)
    b = torch.empty(hi - lo, dtype=torch.float32, device=device)
    g = torch.Generator(device=device)
    for r0 in range(lo, hi, BLOCK):
        r1 = min(r0 + BLOCK, hi)
        g.manual_seed(1_000_003 * (seed + 1) + r0 // BLOCK)
        blk = torch.randn(r1 - r0, n, dtype=torch.float32, device=device, generator=g)
        noise = torch.randn(r1 - r0, dtype=torch.float32, device=device, generator=g)
        if tdt == torch.bfloat16:
            blk16 = blk.to(torch.bfloat16)
            A[r0 - lo:r1 - lo] = blk16
            blk = blk16.to(torch.float32)          # b is generated from the matrix the solver will see
        else:
            A[r0 - lo:r1 - lo] = blk
        b[r0 - lo:r1 - lo] = blk @ xt + 0.1 * noise
        del blk, noise
    return A, b


def bytes_per_iter(m_local, n, dtype):
    """B_iter = m*n*s_A + 4m + 16n  (ONE read of A, b, and the four n-vectors)."""
    s = 2 if dtype == "bf16" else 4
    return m_local * n * s + 4 * m_local + 16 * n


def cpu_threads():
    try:
        from threadpoolctl import threadpool_info
        blas = [i["num_threads"] for i in threadpool_info() if i.get("user_api") == "blas"]
        if blas:
            return int(max(blas))
    except Exception:
        pass
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1


def cpu_topology():
    """Host cores as BASELINE.md section 4 asks for them: logical CPUs this process may run on, the physical cores behind
    them (distinct thread-sibling sets), and the BLAS thread count NumPy will actually use."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except Exception:
        allowed = list(range(os.cpu_count() or 1))
    cores = set()
    for c in allowed:
        try:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") as fh:
                cores.add(fh.read().strip())
        except Exception:
            cores.add(str(c))
    blas = None
    try:
        from threadpoolctl import threadpool_info
        info = [i for i in threadpool_info() if i.get("user_api") == "blas"]
        if info:
            blas = dict(threads=int(max(i["num_threads"] for i in info)), library=str(info[0].get("internal_api")),
                        version=str(info[0].get("version")))
    except Exception:
        pass
    return dict(logical=len(allowed), physical=len(cores), os_cpu_count=os.cpu_count(), blas=blas)


SAMPLE_ROWS = 65536     # rows of the CPU-baseline / parity sample (the whole of cfg2; the first 65536 rows of cfg4 / cfg5)


def cpu_baseline(cfg, A_dev, b_dev, a1, a2, L, x_gpu_k, k_check, budget_s=25.0):
    """The oracle (NumPy fp64, all host cores through BLAS) on a bounded sample of the same workload.  x_gpu_k: the
    device's iterate after k_check iterations on the SAME sample (same rows, same L - the SAMPLE's own Lipschitz
    estimate, so that tau is the step the reference would take on it and the iterate really moves - alpha1, alpha2);
    the parity figure of the bench line is its relative distance to the oracle's iterate k_check."""
    from oracle import fos_oracle as orc
    m, n = cfg["m"], cfg["n"]
    rows = min(m, SAMPLE_ROWS)
    t0 = time.perf_counter()
    A64 = A_dev[:rows].to(torch.float32).cpu().numpy().astype(np.float64)   # reference-native dtype
    b64 = b_dev[:rows].cpu().numpy().astype(np.float64)
    log(f"[cpu_baseline] host copy of {rows}x{n} fp64 ({A64.nbytes / 2**30:.1f} GiB) in {time.perf_counter() - t0:.1f}s")
    prob = orc.FistaProblem(A64, b64, a1, a2)
    st = prob.init_state(L)
    parity = None
    prob.step(st)
    prob.step(st)                                   # 2 warm-up iterations
    iters, t0 = 0, time.perf_counter()
    while True:
        prob.step(st)
        iters += 1
        if st.k == k_check and x_gpu_k is not None:
            parity = float(np.linalg.norm(x_gpu_k - st.x) / max(np.linalg.norm(st.x), 1e-300))
        el = time.perf_counter() - t0
        if ((el > budget_s and iters >= 5) or iters >= 200) and (x_gpu_k is None or st.k >= k_check):
            break
    its = iters / el
    scale = rows / m
    sample = (f"oracle FistaProblem.step (NumPy fp64, BLAS threads) on {'the full' if rows == m else 'the first'} "
              f"{rows}x{n} rows of this workload's A in fp64, 2 warm-up + {iters} timed iterations")
    if rows != m:
        sample += f"; value = measured {its:.2f} it/s x {scale:.4f} (linear extrapolation to m = {m})"
    topo = cpu_topology()
    return dict(value=its * scale, unit="it/s", cores=cpu_threads(), kind="port", sample=sample,
                physical_cores=topo["physical"], logical_cpus=topo["logical"], blas=topo["blas"]), parity


def run_workload(name, args, rank, world, device, steps, warmup, want_cpu, dist, repeats=1):
    import ctypes as C
    import fastoptsolver_amd as fos
    from fastoptsolver_amd import _lib, distributed as fd
    from fastoptsolver_amd.operators import vec_stats
    cfg = WORKLOADS[name]
    m, n = cfg["m"], cfg["n"]
    lo, hi = fd.shard_rows(m, world, rank)
    t0 = time.perf_counter()
    A, b = make_shard(cfg, lo, hi, device)
    torch.cuda.synchronize()
    log(f"[rank {rank}] {name}: rows [{lo},{hi}) x {n} {cfg['dtype']} generated in {time.perf_counter() - t0:.1f}s")

    # N > 1 over RCCL: the communicator lives under the C ABI and every row-sum of this rank's shard is all-reduced on
    # the kernels' own stream - K2 -> slab reduce -> ncclAllReduce(n + 1 floats) -> prox + momentum, `k` iterations
    # enqueued without a host step (the gloo rehearsal keeps the split form driven from Python).
    comm = None
    transport = os.environ.get("FOS_COMM_TRANSPORT", "rccl")    # "mesh": the one-shot full-mesh kernel (any backend)
    if world > 1 and (args.backend == "nccl" or transport == "mesh"):
        try:
            comm = fd.Comm(dist.group.WORLD, transport=transport)
            ok = 1.0
        except Exception as exc:                 # e.g. no loadable RCCL for dlopen: fall back to the split form, loudly
            log(f"[rank {rank}] fos_comm unavailable ({exc}); falling back to torch.distributed all-reduce")
            ok = 0.0
        flag = torch.tensor([ok], device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)          # all ranks take the same path
        if float(flag.item()) < 1.0:
            comm = None
    args.comm_mode = ("c-abi/" + transport) if comm is not None else ("torch" if world > 1 else "none")
    args.comm_info = None
    if comm is not None:
        # what the communicator itself reports (not what the launcher asked for) + the cost of the iteration's one
        # collective alone: all-reduce of n + 1 floats, 100 back-to-back calls on the stream, max over ranks
        nr, rk = C.c_int(), C.c_int()
        _lib.check(comm.lib.fos_comm_info(comm.h, C.byref(nr), C.byref(rk)), "fos_comm_info")
        probe = torch.zeros(cfg["n"] + 4, dtype=torch.float32, device=device)
        for _ in range(10):
            comm.allreduce(probe[: cfg["n"] + 1])
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            comm.allreduce(probe[: cfg["n"] + 1])
        e1.record()
        e1.synchronize()
        t_ar = torch.tensor([e0.elapsed_time(e1) * 10.0], dtype=torch.float64, device=device)      # us per call
        dist.all_reduce(t_ar, op=dist.ReduceOp.MAX)
        args.comm_info = dict(ranks_seen_by_communicator=int(nr.value), this_rank=int(rk.value), transport=comm.transport(),
                              allreduce_elements=cfg["n"] + 1, allreduce_us_standalone=float(t_ar.item()))
        comm.check()
    eng = fd.HipShardEngine(A, b, comm=comm, group=dist.group.WORLD if world > 1 else None)
    matvec_prob = fos.prepare(A, None)                       # same A, b = 0: power iteration / A^T b
    if comm is not None:
        matvec_prob.set_comm(comm)
    if args.interleave != "auto":
        eng.prob.replan(interleave=args.interleave == "on")
        matvec_prob.replan(interleave=args.interleave == "on")
    if args.geometry:
        th, ch, rw, wg = (int(v) for v in args.geometry.split("x"))
        eng.prob.tune(th, ch, rw, wg)
        matvec_prob.tune(th, ch, rw, wg)

    # alpha1 = frac * ||A^T b||_inf  (A^T b = -grad at y = 0)
    atb = eng.prob.gemv_pair(torch.zeros(n, device=device), 0.0)          # summed over the ranks when comm is attached
    if world > 1 and comm is None:
        dist.all_reduce(atb)
    a1 = cfg["a1_frac"] * vec_stats(None, atb, None)[3]
    a2 = cfg["a2"]

    # L by the reference's power iteration (100 GEMV pairs), timed separately
    np.random.seed(0)
    v0 = torch.from_numpy(np.random.randn(n).astype(np.float32)).to(device)
    t0 = time.perf_counter()
    if comm is not None:
        L = matvec_prob.power_iter(v0)[0]                    # fos_power_iter: w all-reduced on the stream
    else:
        L = fd.sharded_lipschitz(lambda v: matvec_prob.gemv_pair(v, 0.0), n, v0)
    torch.cuda.synchronize()
    lip_s = time.perf_counter() - t0
    tau = 1.0 / (L + (a2 if a2 > 0 else 0.0))
    eng.reset(tau=tau, alpha1=a1, alpha2=a2)
    solver = fd.ShardedFista(eng)

    def do_steps(k):
        if world == 1:
            eng.st.run(k)          # fused device-driven path: K2 -> (slab reduce + prox + momentum) -> finalize
        else:
            solver.run(k)          # comm: enqueue-only under the C ABI; gloo rehearsal: split form from Python

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    do_steps(warmup)
    x_check, k_check, a1_chk, L_chk, x_moved = None, max(warmup, 10), a1, L, None
    if want_cpu and rank == 0 and world == 1:
        # Parity sample, outside the timed region: the same plan family on the rows the CPU baseline copies (all of
        # cfg2; the first 65536 rows of cfg4 / cfg5), k_check >= 10 iterations from x = 0.  On a row sample alpha1 and L
        # keep their MEANING: the same fraction of the sample's ||A^T b||_inf, and the sample's own power-iteration L
        # (with the full problem's L the step would be 16x too short and the iterate would barely move).
        rows = min(m, SAMPLE_ROWS)
        sub = eng if rows == m and warmup == k_check else fd.HipShardEngine(A[:rows], b[:rows])
        if sub is not eng:
            if rows != m:
                atb_s = sub.prob.gemv_pair(torch.zeros(n, device=device), 0.0)
                a1_chk = cfg["a1_frac"] * vec_stats(None, atb_s, None)[3]
                L_chk = fos.prepare(A[:rows], None).power_iter(v0)[0]
            if args.interleave != "auto":
                sub.prob.replan(interleave=args.interleave == "on")
            sub.reset(tau=1.0 / (L_chk + (a2 if a2 > 0 else 0.0)), alpha1=a1_chk, alpha2=a2)
            sub.st.run(k_check)
        x_check = sub.x().cpu().numpy()
        x_moved = float(np.linalg.norm(x_check))
        assert x_moved > 0.0, "parity sample did not move: vacuous check"
        del sub
    # HIP events around the launches of the dominant kernel: all of them for short runs, every 8th otherwise (the
    # markers cost ~1 % when they bracket every launch of a long run)
    eng.prob.profile(1 if steps <= 32 else 8)
    eng.prob.profile_read()
    # `repeats` timed regions of EXACTLY `steps` iterations each, every one bracketed by barrier + synchronize on both sides
    # and taken as the MAX over ranks; the reported time is their median ("median of 5 runs").
    runs = []
    for _ in range(repeats):
        fence()
        t0 = time.perf_counter()
        do_steps(steps)
        fence()
        el = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([el], dtype=torch.float64, device=device)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
        runs.append(el)
    elapsed = float(np.median(runs))
    k_ms, k_launches = eng.prob.profile_read()
    eng.prob.profile(0)
    # beside the nominal 8 TB/s, what THIS box delivers to a kernel that only loads (the same A, after the timed
    # regions): fos_stream_read_probe, 8 non-temporal 16-byte loads in flight per thread, HIP events around the launches.
    from fastoptsolver_amd import _core
    read_gbps, read_us = _core.stream_read_probe(A, launches=max(4, min(40, int(2e11 // max(A.numel() * A.element_size(), 1)))))
    st = eng.status()
    assert int(st.k) == warmup + steps * repeats and st.stopped == 0, (int(st.k), st.stopped)
    assert math.isfinite(st.this_step) and st.this_step > 0.0, "iterate did not move: invalid run"
    replicas_identical = None
    if world > 1:
        # every rank applied the identical update to the identical reduced gradient: the replicated iterates must agree
        # bit for bit (a transport that delivered different sums to different ranks would show here)
        xr = eng.x()
        probe = torch.stack([xr.sum(), xr.abs().sum(), (xr * xr).sum()]).to(torch.float64)
        hi_, lo_ = probe.clone(), probe.clone()
        dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
        replicas_identical = bool(torch.equal(hi_, lo_))
        assert replicas_identical, "replicated iterates differ between ranks: invalid run"

    b_iter = bytes_per_iter(hi - lo, n, cfg["dtype"])
    kern_us = k_ms * 1e3 / max(k_launches, 1)
    res = dict(
        workload=name, ms_per_step=elapsed * 1e3 / steps, value=steps / elapsed, steps=steps, warmup=warmup, repeats=repeats,
        ms_per_step_runs=[e * 1e3 / steps for e in runs],
        kernel_us=kern_us, kernel_launches=int(k_launches), bytes_iter_per_gpu=b_iter,
        achieved_gbps=b_iter / (kern_us * 1e-6) / 1e9 if k_launches else None,
        step_gbps=b_iter / (elapsed / steps) / 1e9, lipschitz_s=lip_s, L=L, alpha1=a1, alpha2=a2,
        read_gbps=read_gbps, read_us=read_us,
        plan=eng.prob.plan(), m=m, n=n, rows_per_gpu=hi - lo, dtype=cfg["dtype"], reg=cfg["reg"],
        final_step_norm=st.this_step, replicas_identical=replicas_identical)
    cpu, parity = None, None
    if want_cpu and rank == 0 and world == 1:
        cpu, parity = cpu_baseline(cfg, A, b, a1_chk, a2, L_chk, x_check, k_check)
        assert parity is not None and parity < 1e-5, f"GPU iterate {k_check} differs from the oracle by {parity}: invalid run"
    res["cpu_baseline"], res["parity_rel_err"], res["parity_k"] = cpu, parity, k_check
    res["parity_rows"] = min(m, SAMPLE_ROWS)
    res["parity_x_norm"] = x_moved
    del solver, eng, matvec_prob, A, b, comm
    torch.cuda.empty_cache()
    return res


def lbfgs_reference(device):
    """BASELINE config 3: LBFGSSolver("ridge", 0, 1).fit on the cfg2 matrix (65536 x 8192 fp32), fp64 end to end under the
    C ABI (fos_lbfgs_minimize).  Warm fit timed on the wall clock; fg = one fp64-accumulating pass over A, priced on
    B_fg = m*n*4 + 4m + 8n with the HIP-event mean of the passes.  Parity of this exact run (SciPy's nit / nfev,
    every iterate to 1e-5) is tests/test_gpu_parity.py::test_cfg3_full_size_lbfgs."""
    import fastoptsolver_amd as fos
    cfg = WORKLOADS["cfg2"]
    A, b = make_shard(cfg, 0, cfg["m"], device)
    prob = fos.prepare(A, b)
    fos.LBFGSSolver("ridge", 0.0, 1.0).fit(prob, None)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    s = fos.LBFGSSolver("ridge", 0.0, 1.0).fit(prob, None)
    torch.cuda.synchronize(device)
    wall = time.perf_counter() - t0
    met = fos.get_metrics()
    b_fg = cfg["m"] * cfg["n"] * 4 + 4 * cfg["m"] + 8 * cfg["n"]
    out = dict(workload="cfg3: LBFGSSolver('ridge', 0, 1.0).fit, A 65536x8192 f32 on 1 GPU, fp64 iterate / gradient / line search",
               nit=int(s.nit_), nfev=int(s.nfev_), task=s.task_, fit_ms=wall * 1e3, iterations_per_second=s.nit_ / wall,
               fg_device_mean_us=met["grad_time_mean"] * 1e6, fg_algorithmic_bytes=b_fg,
               fg_roofline_frac=b_fg / met["grad_time_mean"] / (HBM_PEAK_GBPS * 1e9), final_obj=float(s.final_obj_))
    del prob, A, b
    torch.cuda.empty_cache()
    return out


def mfma_reference(device):
    """north_star's "MFMA-busy counters": the two places this build uses the matrix cores - 16 Armijo candidates per pass
    (fos_residual_batch) and 16 regularisation weights in lockstep (fos_fista_run_multi) - timed here with HIP events at
    cfg2's shape, fp32 and bf16 storage; the SQ_VALU_MFMA_BUSY_CYCLES figures beside them are the committed rocprofv3 --pmc
    passes of the same kernels (profiles/r01_linesearch_mfma.md, profiles/r02_multilambda.md), not counters of this run."""
    import fastoptsolver_amd as fos
    from fastoptsolver_amd import _core
    cfg = WORKLOADS["cfg2"]
    m, n = cfg["m"], cfg["n"]
    A32, b = make_shard(cfg, 0, m, device)
    pmc = {"f32": dict(line_search_mfma_util=0.322, path_mfma_util=(0.317, 0.344),
                       instruction="v_mfma_f32_16x16x4_f32", source="profiles/r01_linesearch_mfma.md, profiles/r02_multilambda.md"),
           "bf16": dict(line_search_mfma_util=0.115, path_mfma_util=(0.127, 0.124),
                        instruction="v_mfma_f32_16x16x32_bf16 (candidates / residuals as three bf16 terms)",
                        source="profiles/r01_linesearch_mfma.md, profiles/r02_multilambda.md")}
    out = {}
    for kind in ("f32", "bf16"):
        A = A32 if kind == "f32" else A32.to(torch.bfloat16)
        esz = 4 if kind == "f32" else 2
        prob = fos.prepare(A, b)
        leg = {}
        # (1) 16 candidates per pass: ||A dlt_j||^2, j < 16, from one read of A
        X = torch.randn(n, 16, device=device)
        prob.residual_batch(X)
        prob.profile(1)
        prob.profile_read()
        for _ in range(30):
            prob.residual_batch(X)
        ms, cnt = prob.profile_read()
        prob.profile(0)
        us = ms * 1e3 / max(cnt, 1)
        terms = 3 if kind == "bf16" else 1
        leg["line_search_16_candidates"] = dict(
            kernel="residual_batch_mfma_kernel" if kind == "f32" else "residual_batch_mfma_bf16_kernel", kernel_us=us,
            hbm_frac=m * n * esz / (us * 1e-6) / (HBM_PEAK_GBPS * 1e9), mfma_tflops=2.0 * m * n * 16 * terms / (us * 1e-6) / 1e12,
            mfma_util_committed_pmc=pmc[kind]["line_search_mfma_util"])
        # (2) 16 weights in lockstep: two GEMM-shaped products per iteration
        lam = float((A32.T @ b).abs().max())
        L = 4.0 * m
        hs = []
        for i in range(16):
            st = _core.Fista(prob)
            st.reset(1.0 / L, 0.2 * lam * 0.8 ** i, 0.0)
            hs.append(st)
        if _core.run_multi(hs, 3):
            torch.cuda.synchronize()
            best = 1e30
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _core.run_multi(hs, 40)
                e1.record()
                e1.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / 40)
            reads = 1 if prob.plan()["cluster"] else 2        # the planner's one-read cluster form (fp32, 4096 / 8192 columns)
            leg["path_16_weights"] = dict(us_per_iteration=best, us_per_weight_iteration=best / 16,
                                          form="one-read cluster pass" if reads == 1 else "two products (A read twice)",
                                          a_reads_per_iteration=reads,
                                          hbm_frac_on_those_reads=reads * m * n * esz / (best * 1e-6) / (HBM_PEAK_GBPS * 1e9),
                                          mfma_util_committed_pmc=pmc[kind]["path_mfma_util"])
        leg["instruction"], leg["pmc_source"] = pmc[kind]["instruction"], pmc[kind]["source"]
        out[kind] = leg
        del hs, prob
    del A32, b
    torch.cuda.empty_cache()
    out["workload"] = f"cfg2 shape ({m}x{n}), 1 GPU; kernel_us by HIP events on the launch stream (fos_problem_profile)"
    return out


def fused_reference(device):
    """north_star's literal wording of the step - "GEMV pair with CDNA4 MFMA and LDS-staged A tiles, fused in one launch with
    the soft-threshold prox and the FISTA momentum update so the iterate x stays resident" - as its own measurement:
    fos_fista_run_fused (csrc/fused_step.hpp: ONE persistent launch per run, 4-row panels staged through LDS, the row dots on
    v_mfma_f32_4x4x1_16B_f32, prox + momentum by the workgroup that owns the columns, its slice of x_k, x_{k-1} in LDS) at
    cfg2, next to the default two-launch VALU step on the same problem.  Opt-in (FOS_PLAN_FUSED_MFMA): the default is faster."""
    import fastoptsolver_amd as fos
    from fastoptsolver_amd import _core
    from fastoptsolver_amd.operators import vec_stats
    cfg = WORKLOADS["cfg2"]
    m, n = cfg["m"], cfg["n"]
    A, b = make_shard(cfg, 0, m, device)
    prob = fos.prepare(A, b)
    atb = prob.gemv_pair(torch.zeros(n, device=device), 0.0)
    a1 = cfg["a1_frac"] * vec_stats(None, atb, None)[3]
    np.random.seed(0)
    v0 = torch.from_numpy(np.random.randn(n).astype(np.float32)).to(device)
    L = fos.prepare(A, None).power_iter(v0)[0]
    out = {}
    xs = {}
    for name in ("two_launch_valu", "one_launch_mfma_lds"):
        st = _core.Fista(prob)
        st.reset(1.0 / L, a1, 0.0)
        run = st.run if name == "two_launch_valu" else st.run_fused
        if run(10) is False:
            return dict(error="fos_fista_run_fused does not serve this shape")
        torch.cuda.synchronize()
        times = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(200)
            e1.record()
            e1.synchronize()
            times.append(e0.elapsed_time(e1) * 1e3 / 200)
        us = float(np.median(times))
        xs[name] = st.x_tensor()
        out[name] = dict(us_per_iteration=us, iterations_per_second=1e6 / us,
                         whole_step_frac=bytes_per_iter(m, n, "f32") / (us * 1e-6) / (HBM_PEAK_GBPS * 1e9), launches_per_run=None)
        del st
    out["two_launch_valu"]["launches_per_run"] = "2 per iteration"
    out["one_launch_mfma_lds"]["launches_per_run"] = "1 per call (200 iterations each here), 2 grid-wide barriers per iteration"
    out["iterate_rel_diff_after_1010_iterations"] = float((xs["one_launch_mfma_lds"] - xs["two_launch_valu"]).norm() /
                                                        xs["two_launch_valu"].norm())
    out["workload"] = f"cfg2 ({m}x{n} f32, lasso), 1 GPU; median of 5 x 200 iterations by HIP events"
    del prob, A, b
    torch.cuda.empty_cache()
    return out


def load_traffic(workload):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc pass (profiles/)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as fh:
            return json.load(fh).get(workload, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def roofline_obj(res, traffic_key):
    """`roofline` of one measured workload: algorithmic bytes per launch / HIP-event average of the dominant kernel."""
    return {
        "bound": "hbm",
        "achieved": res["achieved_gbps"],
        "peak": HBM_PEAK_GBPS,
        "unit": "GB/s",
        "frac": (res["achieved_gbps"] or 0.0) / HBM_PEAK_GBPS,
        "traffic": load_traffic(traffic_key) if traffic_key else None,   # PMC pass exists for the 1-GPU shapes only
        "traffic_source": ("profiles/pmc_traffic.json: FETCH_SIZE x 2 + WRITE_SIZE per launch from a committed rocprofv3 "
                           "--pmc pass of this workload, NOT a counter of this run") if traffic_key else None,
        "kernel": "fos::gemv_pair_kernel (single pass: r = A y - b and g += A^T r from the same registers)",
        "kernel_avg_us": res["kernel_us"],
        "kernel_launches_timed": res["kernel_launches"],
        "algorithmic_bytes_per_launch": res["bytes_iter_per_gpu"],
        "whole_step_frac": res["step_gbps"] / HBM_PEAK_GBPS,
        # context, not the denominator of `frac`: a read-only pass over the same A on this box, this run
        "stream_read_measured": {"gbps": res["read_gbps"], "us_per_pass_over_A": res["read_us"],
                                 "kernel_over_stream_read": (res["achieved_gbps"] or 0.0) / res["read_gbps"],
                                 "what": "fos_stream_read_probe: loads + adds only, same buffer, HIP events; the fastest of three access orders (a contiguous range per workgroup / one window swept by all in 8 KiB / in 64 KiB pieces)"},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed regions of exactly --steps iterations each; the median is reported (1 = a single region)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-target-ref", "--no-scale-ref", dest="no_target_ref", action="store_true",
                    help="skip the cfg2 (65536 x 8192) single-GPU measurement that the default N=1 run adds")
    ap.add_argument("--geometry", type=str, default="", help="THREADSxCHUNKSxROWSxWORKGROUPS override (tuning)")
    ap.add_argument("--interleave", choices=("auto", "on", "off"), default="auto",
                    help="row order of the streaming pass: planner default / round-robin rows / contiguous blocks (A/B)")
    ap.add_argument("--rows", type=int, default=0, help="override m (rehearsal / tuning only; reported in config)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with: python -m torch.distributed.run --nnodes=1 "
                     "--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # REHEARSAL ONLY (FOS_BENCH_BACKEND=gloo): lets N ranks share one GPU on a 1-GPU box to exercise the N>1 code
    # path (sharding, barriers, all-reduce, JSON); its numbers are meaningless and are labelled as such.
    backend = os.environ.get("FOS_BENCH_BACKEND", "nccl")
    args.backend = backend
    dev_index = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)

    # ONE series for every N: BASELINE's "iters/sec reported at 1/2/4/8 GPUs on A in R^{2^20 x 16384}" (cfg4, total work
    # fixed -> strong scaling; 64 GiB fits one MI355X, so N=1 is a real point of the series).  The single-GPU roofline
    # target is quoted on cfg2 (65536 x 8192): the default N=1 run measures it as well and reports it as `target_ref`
    # with its own roofline and cpu_baseline objects.
    name = args.workload or "cfg4"
    if args.rows:
        WORKLOADS[name] = dict(WORKLOADS[name], m=int(args.rows))
    default_run = world == 1 and name == "cfg4" and not args.no_target_ref and args.workload is None and not args.rows

    def settle(seconds):
        # Returning VRAM to the driver (empty_cache at the end of a workload) leaves the device busy for a while: the same
        # allocation measured 306.6 us per cfg2 pass before and 316.8 us right after 64 GiB were freed, 306.6 us again after
        # 5 s idle (profiles/r03_row_order.md, tools/order_probe.py).  Every workload of this run is therefore measured on
        # a quiet device: small ones first, a pause after each free.
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        time.sleep(seconds)

    # The riders of the default N=1 run go FIRST (2 GiB each), the 64 GiB headline last.
    target_ref = None
    settle(4.0)            # a process that ended just before this one (a test suite, the previous N) is still being released
    if default_run:
        try:
            r2 = run_workload("cfg2", args, rank, world, device, steps=200, warmup=10,
                              want_cpu=not args.no_cpu_baseline, dist=dist, repeats=args.repeats)
            c2 = WORKLOADS["cfg2"]
            target_ref = dict(
                workload=f"cfg2: {c2['reg']} FISTA, A {c2['m']}x{c2['n']} f32 on 1 GPU - the configuration the "
                         ">=70 % single-GPU roofline target is quoted on",
                value=r2["value"], unit="it/s", ms_per_step=r2["ms_per_step"], steps=r2["steps"], warmup=r2["warmup"],
                repeats=r2["repeats"], ms_per_step_runs=r2["ms_per_step_runs"],
                roofline=roofline_obj(r2, "cfg2"), cpu_baseline=r2["cpu_baseline"],
                parity_rel_err_vs_cpu_at_warmup_iterate=r2["parity_rel_err"],
                parity={"rel_err": r2["parity_rel_err"], "iterate": r2["parity_k"], "rows": r2["parity_rows"],
                        "x_norm": r2["parity_x_norm"]},
                kernel_plan=r2["plan"], lipschitz_power_iteration_s=r2["lipschitz_s"])
        except Exception as exc:      # must not take the headline measurement down with it
            target_ref = dict(error=str(exc)[:200])
        settle(2.0)

    # BASELINE config 3 (L-BFGS, ridge, on the cfg2 matrix) rides along in the default N=1 run as `lbfgs_ref`
    lbfgs_ref = None
    if default_run:
        try:
            lbfgs_ref = lbfgs_reference(device)
        except Exception as exc:
            lbfgs_ref = dict(error=str(exc)[:200])
        settle(3.0)

    # BASELINE's MFMA evidence (16 Armijo candidates / 16 weights per pass on the matrix cores) as driver-timed numbers
    mfma_ref = None
    if default_run:
        try:
            mfma_ref = mfma_reference(device)
        except Exception as exc:
            mfma_ref = dict(error=str(exc)[:200])
        settle(3.0)

    # north_star's literal one-launch MFMA / LDS-staged step, measured next to the default (opt-in plan)
    fused_ref = None
    if default_run:
        try:
            fused_ref = fused_reference(device)
        except Exception as exc:
            fused_ref = dict(error=str(exc)[:200])
        settle(3.0)

    res = run_workload(name, args, rank, world, device, args.steps, args.warmup,
                       want_cpu=not args.no_cpu_baseline, dist=dist, repeats=args.repeats)

    if rank == 0:
        cfg = WORKLOADS[name]
        out = {
            "metric": "fista_iterations_per_second",
            "value": res["value"],
            "unit": "it/s",
            "n_gpus": world,
            "steps": res["steps"],
            "warmup": res["warmup"],
            "ms_per_step": res["ms_per_step"],
            "repeats": res["repeats"],
            "ms_per_step_runs": res["ms_per_step_runs"],
            "timing": "median over `repeats` timed regions of exactly `steps` iterations each (barrier + synchronize on both "
                      "sides, max over ranks)",
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32" if cfg["dtype"] == "f32" else "bf16-in/f32-acc",
            "data": "synthetic",
            "config": {
                "workload": f"{name}: {cfg['reg']} FISTA, A {cfg['m']}x{cfg['n']} {cfg['dtype']}, "
                            f"rows sharded over {world} GPU(s), alpha1={res['alpha1']:.4g}, alpha2={res['alpha2']}",
                "m": cfg["m"], "n": cfg["n"], "rows_per_gpu": res["rows_per_gpu"],
                "sharding": f"rows/{world}" if world > 1 else "none",
                "collective": ("all-reduce(SUM) of n+1 fp32 per iteration, enqueued by libfos_hip.so on the kernels' stream ("
                               + getattr(args, "comm_mode", "") + ")"
                               if getattr(args, "comm_mode", "").startswith("c-abi") else
                               ("torch.distributed all-reduce of n+1 fp32 per iteration (split form)" if world > 1 else "none")),
                "kernel_plan": res["plan"],
                "iterate_state": "fp64 on device; y rounded once to fp32 for the single pass over A",
                "backend": backend if world > 1 else None,
                "rehearsal": bool(args.rows) or (world > 1 and backend != "nccl"),
                "replicas_identical": res["replicas_identical"],
                "communicator": getattr(args, "comm_info", None),
            },
            "roofline": roofline_obj(res, name if world == 1 else None),
            "cpu_baseline": res["cpu_baseline"],
            "parity_rel_err_vs_cpu_at_warmup_iterate": res["parity_rel_err"],
            "parity": {"rel_err": res["parity_rel_err"], "iterate": res["parity_k"], "rows": res["parity_rows"],
                       "x_norm": res["parity_x_norm"],
                       "what": "||x_gpu - x_oracle|| / ||x_oracle|| after `iterate` iterations from x = 0 on the first "
                               "`rows` rows of this workload with the SAMPLE's own power-iteration L and its own alpha1 "
                               "fraction (tau = 1/L_sample: the iterate moves as on a real problem); the run aborts above "
                               "1e-5.  The full 2^20-row problem is checked at size by tests/test_gpu_parity.py::"
                               "test_cfg4_full_size_properties"},
            "lipschitz_power_iteration_s": res["lipschitz_s"],
            "target_ref": target_ref,
            "lbfgs_ref": lbfgs_ref,
            "mfma_ref": mfma_ref,
            "fused_ref": fused_ref,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

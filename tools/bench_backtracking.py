#!/usr/bin/env python3
"""Backtracking FISTA, ms per iteration: decided on the device (fos_fista_run_backtracking, enqueue-only) vs driven by the
host (one synchronising batch per iteration), at cfg2 and on mid-size problems where the host round trip dominates."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastoptsolver_amd as fos
from fastoptsolver_amd import iterative_solvers as its

torch.cuda.set_device(0)
out = {}
for m, n in ((65536, 8192), (16384, 2048), (4096, 1024), (20000, 256)):
    A = torch.randn(m, n, device="cuda")
    b = torch.randn(m, device="cuda")
    prob = fos.prepare(A, b)
    lam = float((A.T @ b).abs().max())
    L = float(m + n + 2.0 * (m * n) ** 0.5)
    res = {}
    for name, kw in (("device-driven", {}), ("device-driven with history", dict(return_history=True))):
        fos.fista(prob, None, "lasso", 0.1 * lam, 0.0, max_iter=5, L=L, backtracking=True, t_init_factor=2.0, **kw)
        iters = 60
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fos.fista(prob, None, "lasso", 0.1 * lam, 0.0, max_iter=iters, L=L, backtracking=True, t_init_factor=2.0, check_every=16, **kw)
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) * 1e3 / iters
    res["shrinks"] = sum(its.ls_call_iters)
    out[f"{m}x{n}"] = res
    print(f"{m}x{n}: {res}", flush=True)
    del prob, A, b
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "bench_backtracking.json"), "w"), indent=1)

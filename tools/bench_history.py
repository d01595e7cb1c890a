#!/usr/bin/env python3
"""return_history=True at cfg2: per-iteration wall time (the reference pays 3 passes over A per iteration here,
this path pays 1)."""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastoptsolver_amd as fos
from bench import make_shard, WORKLOADS
torch.cuda.set_device(0)
cfg = WORKLOADS["cfg2"]
A, b = make_shard(cfg, 0, cfg["m"], torch.device("cuda", 0))
prob = fos.prepare(A, b)
lam = float((A.T @ b).abs().max())
np.random.seed(0)
L = fos.estimate_lipschitz(prob)
res = {}
for name, kw in (("plain", dict()), ("history", dict(return_history=True)), ("history+armijo", dict(return_history=True, backtracking=True))):
    fos.fista(prob, None, "lasso", 0.1 * lam, 0.0, max_iter=5, L=L, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = fos.fista(prob, None, "lasso", 0.1 * lam, 0.0, max_iter=100, L=L, **kw)
    torch.cuda.synchronize(); wall = time.perf_counter() - t0
    res[name] = wall * 1e3 / 100
    print(name, f"{res[name]:.3f} ms/iter", flush=True)
json.dump(res, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "bench_history.json"), "w"))

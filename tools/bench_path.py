#!/usr/bin/env python3
"""Regularisation path at cfg2: 4 weights in lockstep (one pass over A per iteration) vs one by one."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastoptsolver_amd as fos
from fastoptsolver_amd import _core
from bench import make_shard, WORKLOADS
torch.cuda.set_device(0)
cfg = WORKLOADS["cfg2"]
A, b = make_shard(cfg, 0, cfg["m"], torch.device("cuda", 0))
prob = fos.prepare(A, b)
lam = float((A.T @ b).abs().max())
np.random.seed(0)
L = fos.estimate_lipschitz(prob)
alphas = [(0.2 * lam, 0.0), (0.1 * lam, 0.0), (0.05 * lam, 0.0), (0.025 * lam, 0.0)]
out = {}
for nl in (1, 2, 3, 4):
    hs = []
    for a1, a2 in alphas[:nl]:
        st = _core.Fista(prob); st.reset(1.0 / L, a1, a2); hs.append(st)
    run = (lambda k: _core.run_multi(hs, k)) if nl > 1 else (lambda k: hs[0].run(k))
    run(5); torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(100); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 100)
    out[nl] = dict(us_per_iteration=best, us_per_lambda_iteration=best / nl,
                   a_bytes_frac_of_8TBps=cfg["m"] * cfg["n"] * 4 / (best * 1e-6) / 8e12)
    print(nl, out[nl], flush=True)
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "bench_path.json"), "w"), indent=1)

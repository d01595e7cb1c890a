#!/usr/bin/env python3
"""Backtracking FISTA at cfg2: cost per iteration with the MFMA-batched line search vs one candidate per pass,
plus the bare kernel time of the batched residual pass (roofline of the matrix-core kernel)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastoptsolver_amd as fos
from fastoptsolver_amd import iterative_solvers as its
from bench import make_shard, WORKLOADS

torch.cuda.set_device(0)
DT = "bf16" if "--dtype=bf16" in sys.argv else "f32"          # bf16: A stored in bf16, 3-term bf16 MFMA batch kernel
cfg = dict(WORKLOADS["cfg2"])
A, b = make_shard(cfg, 0, cfg["m"], torch.device("cuda", 0))
if DT == "bf16":
    A = A.to(torch.bfloat16)
prob = fos.prepare(A, b)
lam = float((A.float().T @ b).abs().max())
np.random.seed(0)
L = fos.estimate_lipschitz(prob)
out = {}
for batch in (True, False):
    for t_init in (1.0, 2.0):
        its.reset_metrics()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = its._drive(prob, A, mode=0, prox_kind=0, alpha1=0.1 * lam, alpha2=0.0, tau=t_init / L, backtracking=True,
                        eta=0.5, max_iter=60, grad_tol_check=True, batch_trials=batch)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        out[f"batch={batch},t_init={t_init}"] = dict(wall_ms_per_iter=wall * 1e3 / 60, shrinks=sum(its.ls_call_iters),
                                                     max_shrinks=max(its.ls_call_iters), ls_ms_mean=1e3 * float(np.mean(its.ls_call_times)))
        print(f"batch={batch} t_init={t_init}: {out[f'batch={batch},t_init={t_init}']}", flush=True)
# bare kernel time of the batched pass
X = torch.randn(cfg["n"], 16, device="cuda")
prob.residual_batch(X)
prob.profile(1); prob.profile_read()
for _ in range(20):
    prob.residual_batch(X)
ms, cnt = prob.profile_read()
us = ms * 1e3 / cnt
byts = cfg["m"] * cfg["n"] * (2 if DT == "bf16" else 4)
out["residual_batch_mfma_kernel"] = dict(us=us, gbps=byts / (us * 1e-6) / 1e9, frac_hbm=byts / (us * 1e-6) / 8e12,
                                         tflops=2.0 * cfg["m"] * cfg["n"] * 16 * (3 if DT == "bf16" else 1) / (us * 1e-6) / 1e12)
print(out["residual_batch_mfma_kernel"], flush=True)
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", f"bench_linesearch_{DT}.json"), "w"), indent=1)

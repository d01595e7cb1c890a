#!/usr/bin/env python3
"""Regularisation path: 1-4 weights in lockstep (multi-vector VALU pass, one read of A per iteration) and 8 / 16 weights
(matrix-core pass, two GEMM-shaped products per iteration) vs one by one - cfg2 (65536 x 8192 fp32) and the bf16 shard of
config 5 (131072 x 16384).  Per product: HIP-event kernel time of the A-pass kernels (fos_problem_profile)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastoptsolver_amd as fos
from fastoptsolver_amd import _core
from bench import make_shard, WORKLOADS
torch.cuda.set_device(0)
out = {}
for name, cfg in (("cfg2", WORKLOADS["cfg2"]), ("cfg5_shard", dict(WORKLOADS["cfg5"], m=131072))):
    A, b = make_shard(cfg, 0, cfg["m"], torch.device("cuda", 0))
    prob = fos.prepare(A, b)
    lam = float((A.float().T @ b).abs().max()) if cfg["m"] * cfg["n"] < 2 ** 30 else 1e5
    L = 4.0 * cfg["m"]
    alphas = [(0.2 * lam * 0.8 ** i, cfg["a2"]) for i in range(16)]
    esz = 2 if cfg["dtype"] == "bf16" else 4
    res = {}
    for nl in (1, 2, 4, 8, 16):
        hs = []
        for a1, a2 in alphas[:nl]:
            st = _core.Fista(prob); st.reset(1.0 / L, a1, a2); hs.append(st)
        run = (lambda k: _core.run_multi(hs, k)) if nl > 1 else (lambda k: hs[0].run(k))
        if nl > 1 and not _core.run_multi(hs, 2):
            print(name, nl, "no multi kernel"); continue
        run(3); torch.cuda.synchronize()
        best = 1e9
        iters = 40
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(iters); e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
        prob.profile(1); prob.profile_read(); run(10); ms, cnt = prob.profile_read(); prob.profile(0)
        res[nl] = dict(us_per_iteration=best, us_per_lambda_iteration=best / nl, a_pass_us=ms * 1e3 / max(cnt, 1),
                       a_bytes_over_time_frac_of_8TBps=cfg["m"] * cfg["n"] * esz / (best * 1e-6) / 8e12)
        print(name, nl, res[nl], flush=True)
        del hs
    out[name] = res
    del prob, A, b
    torch.cuda.empty_cache()
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "bench_path.json"), "w"), indent=1)

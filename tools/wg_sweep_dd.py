#!/usr/bin/env python3
"""Workgroup count of the streaming fp64-accumulating pass (L-BFGS fg: pass + fp64 slab sum) against the planner's choice
(fos_problem_tune_dd)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastoptsolver_amd as fos
from fastoptsolver_amd import _core, _lib
torch.cuda.set_device(0)
shapes = [(2796032, 192), (2097152, 256), (1677568, 320), (1398016, 384), (1048576, 512), (838656, 640), (698880, 768), (524288, 1024),
          (419328, 1280), (262144, 2048), (209664, 2560), (131072, 4096), (65536, 8192), (200000, 256), (100000, 1024)]
for m, n in shapes:
    A = torch.randn(m, n, device="cuda"); b = torch.randn(m, device="cuda")
    prob = fos.prepare(A, b)
    x64 = torch.randn(n, dtype=torch.float64, device="cuda"); out = torch.zeros(n + 1, dtype=torch.float64, device="cuda")
    run = lambda k: [_lib.check(prob.lib.fos_gemv_pair_dd(prob.h, _core.ptr(x64), 0.5, _core.ptr(out))) for _ in range(k)]
    res = []
    for wg in (0, 256, 512, 768, 1024, 1536, 2048, 4096):
        _lib.check(prob.lib.fos_problem_tune_dd(prob.h, wg))
        run(5); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(30); e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / 30)
        res.append((wg, best))
    base = res[0][1]
    print(f"{m}x{n}: planner {base:.1f} us = {m*n*4/base/8e4:.1f} % | " + "  ".join(f"{wg}: {t:.1f}" for wg, t in res[1:]), flush=True)
    del prob, A, b
    torch.cuda.empty_cache()

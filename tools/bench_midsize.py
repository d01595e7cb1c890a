#!/usr/bin/env python3
"""Per-iteration time of the plain FISTA loop on small and mid-size problems (the regime bound by launch latency and the
update kernel, not by HBM): microseconds per iteration, the plan, and the NumPy oracle's time for the same step."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastoptsolver_amd as fos
from fastoptsolver_amd import _core

torch.cuda.set_device(0)
shapes = [(1000, 5), (20000, 5), (100000, 5), (10000, 100), (50000, 200), (5000, 500), (20000, 1000), (8000, 3000), (3000, 5000),
          (2000, 10000), (100000, 64), (200000, 256), (30000, 2048)]
for m, n in shapes:
    rng = np.random.default_rng(m + n)
    A = rng.standard_normal((m, n)).astype(np.float32)
    b = rng.standard_normal(m).astype(np.float32)
    prob = fos.prepare(torch.as_tensor(A).cuda(), b)
    st = _core.Fista(prob); st.reset(1e-9, 1.0, 0.0); st.run(20); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); st.run(200); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 5.0)
    A64 = A.astype(np.float64); y = rng.standard_normal(n); b64 = b.astype(np.float64)
    t0 = time.perf_counter()
    reps = max(3, int(2e8 // (m * n)))
    for _ in range(reps):
        g = A64.T @ (A64 @ y - b64)
    cpu = (time.perf_counter() - t0) / reps * 1e6
    pl = prob.plan()
    kind = "resident" if pl["resident"] else ("tall" if pl["tall"] else f"{pl['threads']}x{pl['chunks']}")
    print(f"{m}x{n}: {best:8.1f} us / iteration  [{kind}, {pl['workgroups']} wg]   NumPy fp64 gradient alone {cpu:9.1f} us  ({cpu / best:.0f}x)", flush=True)

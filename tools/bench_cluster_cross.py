#!/usr/bin/env python3
"""16 weights in lockstep: the two-product matrix-core pass against the one-read cluster pass (FOS_PLAN_CLUSTER) by shape -
where does the one-read form win?  us per iteration (all 16 weights), best of 3 x 30 iterations, same process."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastoptsolver_amd as fos
from fastoptsolver_amd import _core

torch.cuda.set_device(0)
SHAPES = [(65536, 8192), (32768, 8192), (262144, 8192), (131072, 4096), (524288, 4096), (65536, 6144), (131072, 16384), (65536, 3072)]
for m, n in SHAPES:
    A = torch.randn(m, n, device="cuda")
    b = torch.randn(m, device="cuda")
    row = {}
    for cluster in (False, True, False, True):
        prob = fos.prepare(A, b)
        prob.replan(cluster=cluster)
        hs = []
        for i in range(16):
            st = _core.Fista(prob)
            st.reset(1.0 / (4.0 * m), 10.0 * 0.8 ** i, 0.0)
            hs.append(st)
        if not _core.run_multi(hs, 3):
            row[cluster] = None
            continue
        planned = prob.plan()["cluster"]
        torch.cuda.synchronize()
        best = 1e30
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); _core.run_multi(hs, 30); e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / 30)
        row.setdefault(cluster, []).append((best, planned))
        del hs, prob
    fmt = lambda v: "n/a" if not v else " / ".join(f"{t:.0f} us (cluster planned {p})" for t, p in v)
    print(f"{m}x{n}: two-product {fmt(row.get(False))}   one-read {fmt(row.get(True))}", flush=True)
    del A, b
    torch.cuda.empty_cache()

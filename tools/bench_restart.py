#!/usr/bin/env python3
"""Per-iteration time of the device-controlled loop (adaptive restart on: momentum and stop rules decided on the device every
iteration) against the plain loop, by size: what the scalar bookkeeping costs."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastoptsolver_amd as fos
from fastoptsolver_amd import _core

torch.cuda.set_device(0)
for m, n in [(65536, 8192), (16384, 4096), (16384, 2048), (4096, 1024), (2048, 512)]:
    g = torch.Generator(device="cuda").manual_seed(m + n)
    A = torch.randn(m, n, device="cuda", generator=g)
    b = torch.randn(m, device="cuda", generator=g)
    prob = fos.prepare(A, b)
    prob.replan(no_resident=True)
    res = {}
    for name, kw in (("plain", {}), ("adaptive restart", dict(adaptive_restart=True, restart_threshold=1.0))):
        st = _core.Fista(prob); st.reset(1e-9, 1.0, 0.0, **kw)
        st.run(20); torch.cuda.synchronize()
        best = 1e30
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); st.run(200); e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) * 5.0)
        res[name] = best
    print(f"{m}x{n}: plain {res['plain']:.1f} us / iteration, with adaptive restart {res['adaptive restart']:.1f} us "
          f"(+{res['adaptive restart'] - res['plain']:.1f})", flush=True)

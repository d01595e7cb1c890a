#!/usr/bin/env python3
"""north_star's literal step (one persistent launch, LDS-staged panels, row dots on v_mfma_f32_4x4x1_16B_f32, resident
iterate: fos_fista_run_fused) against the default two-launch VALU step: parity and microseconds per iteration."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastoptsolver_amd as fos
from fastoptsolver_amd import _core

torch.cuda.set_device(0)
shapes = [(65536, 8192), (131072, 4096), (32768, 2048), (65536, 6144)]
if len(sys.argv) > 1:
    shapes = shapes[: int(sys.argv[1])]
for m, n in shapes:
    g = torch.Generator(device="cuda").manual_seed(m + n)
    A = torch.randn(m, n, device="cuda", generator=g)
    xt = torch.zeros(n, device="cuda"); xt[::20] = 1.0
    b = A @ xt + 0.1 * torch.randn(m, device="cuda", generator=g)
    prob = fos.prepare(A, b)
    lam = float((A.T @ b).abs().max())
    L = float(fos.estimate_lipschitz(prob, n_iter=30))
    res = {}
    for mode, a2 in (("lasso", 0.0), ("enet", 0.5)):
        ref = _core.Fista(prob); ref.reset(1.0 / (L + a2), 0.1 * lam, a2); ref.run(25)
        fz = _core.Fista(prob); fz.reset(1.0 / (L + a2), 0.1 * lam, a2)
        ok = fz.run_fused(10) and fz.run_fused(15)           # two calls: the state carries over
        if not ok:
            print(f"{m}x{n}: not served"); break
        xr, xf = ref.x_tensor(), fz.x_tensor()
        sr, sf = ref.status(), fz.status()
        res[mode] = (float((xr - xf).norm() / xr.norm()), int(sf.k), sr.this_step, sf.this_step)
    if not res:
        continue
    def timed(fn, iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(3); torch.cuda.synchronize()
        best = 1e30
        for _ in range(3):
            e0.record(); fn(iters); e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
        return best
    st = _core.Fista(prob); st.reset(1.0 / L, 0.1 * lam, 0.0)
    t2 = timed(lambda k: st.run(k), 100)
    st2 = _core.Fista(prob); st2.reset(1.0 / L, 0.1 * lam, 0.0)
    t1 = timed(lambda k: st2.run_fused(k), 100)
    byt = m * n * 4 + 4 * m + 16 * n
    print(f"{m}x{n}: fused vs two-launch iterate rel diff {res['lasso'][0]:.2e} (lasso) {res['enet'][0]:.2e} (elastic net), k {res['lasso'][1]}, "
          f"step norms {res['lasso'][2]:.6e} / {res['lasso'][3]:.6e}; two-launch {t2:.1f} us = {byt / t2 / 8e4:.1f} % of 8 TB/s, "
          f"one persistent launch {t1:.1f} us = {byt / t1 / 8e4:.1f} %", flush=True)

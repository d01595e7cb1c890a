#!/usr/bin/env python3
"""Where a persistent-step iteration spends its time: per-workgroup wall-clock stamps of the last iteration
(fos_problem_set_fused_stamps) - phase A, wait at barrier 1, phase B, wait at barrier 2 - min / median / max over workgroups."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastoptsolver_amd as fos
from fastoptsolver_amd import _core, _lib

torch.cuda.set_device(0)
shapes = [(65536, 8192), (131072, 4096), (32768, 2048)]
for m, n in shapes:
    g = torch.Generator(device="cuda").manual_seed(m + n)
    A = torch.randn(m, n, device="cuda", generator=g)
    b = torch.randn(m, device="cuda", generator=g)
    prob = fos.prepare(A, b)
    ncu = prob.plan()["cus"]
    stamps = torch.zeros(ncu * 8, dtype=torch.int64, device="cuda")
    _lib.check(prob.lib.fos_problem_set_fused_stamps(prob.h, _core.ptr(stamps)))
    st = _core.Fista(prob); st.reset(1e-9, 1.0, 0.0)
    st.run_fused(5)
    rows = []
    for _ in range(5):
        st.run_fused(20)
        torch.cuda.synchronize()
        s = stamps.cpu().numpy().reshape(ncu, 8).astype(np.float64) / 100.0       # us
        t0 = s[:, 0].min()
        rows.append(dict(start_spread=s[:, 0].max() - t0, phase_a=s[:, 1] - s[:, 0], wait1=s[:, 2] - s[:, 1], phase_b=s[:, 3] - s[:, 2],
                         wait2=s[:, 4] - s[:, 3], total=s[:, 4].max() - t0, a_end_spread=s[:, 1].max() - s[:, 1].min(),
                         b1_release=s[:, 2].min() - s[:, 1].max(), b2_release=s[:, 4].min() - s[:, 3].max()))
    r = rows[-1]
    f = lambda v: f"min {np.min(v):7.1f} med {np.median(v):7.1f} max {np.max(v):7.1f}"
    print(f"{m}x{n}: iteration {r['total']:.1f} us (first start -> last release); start spread {r['start_spread']:.1f} us")
    print(f"   phase A            {f(r['phase_a'])}   (end spread over workgroups {r['a_end_spread']:.1f} us)")
    print(f"   wait at barrier 1  {f(r['wait1'])}   (last arrival -> first release {r['b1_release']:.1f} us)")
    print(f"   phase B            {f(r['phase_b'])}")
    print(f"   wait at barrier 2  {f(r['wait2'])}   (last arrival -> first release {r['b2_release']:.1f} us)", flush=True)
    pa = np.stack([q["phase_a"] for q in rows])                  # [run][workgroup]
    by_xcd = [float(np.median(pa[-1][x::8])) for x in range(8)]
    slow = np.argsort(-pa[-1])[:12]
    again = [int(np.sum(np.isin(np.argsort(-pa[k])[:32], np.argsort(-pa[-1])[:32]))) for k in range(len(rows) - 1)]
    print(f"   phase A median by workgroup %% 8 (XCD): {' '.join(f'{v:.0f}' for v in by_xcd)}")
    print(f"   slowest workgroups of the last run: {' '.join(f'{int(i)}:{pa[-1][i]:.0f}' for i in slow)}")
    print(f"   of its 32 slowest workgroups, how many were among the 32 slowest of the earlier runs: {again}", flush=True)
    _lib.check(prob.lib.fos_problem_set_fused_stamps(prob.h, None))
    del st, prob, A, b
    torch.cuda.empty_cache()

#!/usr/bin/env python3
"""The opt-in chip-resident loop (fos_fista_run_chip: A in the LDS of up to all CUs, one grid barrier per iteration) against
the default two-launch loop on tall-skinny shapes: microseconds per iteration and the iterates' agreement."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastoptsolver_amd as fos
from fastoptsolver_amd import _core

torch.cuda.set_device(0)
for m, n in [(9000, 5), (20000, 5), (50000, 5), (100000, 5), (300000, 5), (1000000, 5), (20000, 16), (100000, 16), (500000, 16), (100000, 8)]:
    A = torch.randn(m, n, device="cuda"); b = torch.randn(m, device="cuda")
    prob = fos.prepare(A, b)
    prob.replan(no_resident=True, chip_resident=False)       # run() = the two-launch loop
    L = float((A.double() ** 2).sum())
    res = {}
    for name in ("two-launch", "chip"):
        st = _core.Fista(prob); st.reset(1.0 / L, 1.0, 0.0)
        run = st.run if name == "two-launch" else st.run_chip
        if run(20) is False:
            res[name] = None; continue
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(500); e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) * 2.0)
        res[name] = (best, st.x_tensor().clone())
    if res["chip"] is None:
        print(f"{m}x{n}: not served"); continue
    # the same with adaptive restart (device-decided momentum): the two-launch loop adds a bookkeeping launch per iteration
    rr = {}
    for name in ("two-launch", "chip"):
        st = _core.Fista(prob); st.reset(1.0 / L, 1.0, 0.0, adaptive_restart=True)
        run = st.run if name == "two-launch" else st.run_chip
        run(20); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(500); e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) * 2.0)
        rr[name] = best
    d = float((res["chip"][1] - res["two-launch"][1]).norm() / res["two-launch"][1].norm())
    print(f"{m}x{n}: two-launch {res['two-launch'][0]:.2f} us / iteration, chip-resident {res['chip'][0]:.2f} us  ({res['two-launch'][0] / res['chip'][0]:.2f}x), "
          f"iterates differ by {d:.1e} after 1520 iterations; with adaptive restart {rr['two-launch']:.2f} -> {rr['chip']:.2f} us ({rr['two-launch'] / rr['chip']:.2f}x)", flush=True)

#!/usr/bin/env python3
"""Tall-skinny regression shapes (m >> n, the reference's own domain at larger m): per-iteration time and the share of
the HBM roofline the streaming kernels reach when a row is only a few dozen bytes long."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastoptsolver_amd as fos
from fastoptsolver_amd import _core, _lib

def run(m, n, iters=200):
    A = torch.randn(m, n, device="cuda"); b = torch.randn(m, device="cuda")
    prob = fos.prepare(A, b)
    st = _core.Fista(prob)
    st.reset(1e-9, 1.0, 0.0)
    st.run(5); torch.cuda.synchronize()
    t0 = time.perf_counter(); st.run(iters); torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / iters * 1e6
    print(f"{m:>9d} x {n:<3d} path {prob.plan()['path']} resident {prob.plan()['resident']}  {us:9.1f} us/iteration  "
          f"{m * n * 4 / us / 1e3:8.1f} GB/s ({m * n * 4 / us / 1e3 / 80:.1f} % of 8 TB/s)", flush=True)

for m, n in ((200_000, 5), (1_000_000, 5), (1_000_000, 8), (4_000_000, 16), (2_000_000, 64)):
    run(m, n)

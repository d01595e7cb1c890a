#!/usr/bin/env python3
"""Per-iteration wall time of the plain device-driven run on small (launch-bound) problems."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastoptsolver_amd as fos
from fastoptsolver_amd import _core, _lib
from fastoptsolver_amd.easy_boston_data import generate_correlated_boston_like_data

def run(name, A, b, iters=2000):
    prob = fos.prepare(A, b)
    st = _core.Fista(prob)
    L = float(np.linalg.norm(np.asarray(A, dtype=np.float64), 2) ** 2)
    st.reset(1.0 / L, 1.0, 0.0, mode=_lib.MODE_FISTA, prox_kind=_lib.PROX_L1)
    st.run(50); torch.cuda.synchronize()
    t0 = time.perf_counter(); st.run(iters); t1 = time.perf_counter(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name:28s} path {prob.plan()['path']}  {dt / iters * 1e6:7.2f} us/iteration  (host enqueue returned after "
          f"{(t1 - t0) / iters * 1e6:6.2f} us/iteration)  plan {prob.plan()}", flush=True)

A, b, _ = generate_correlated_boston_like_data()
run("boston 1000x5 (resident)", A, b)
rng = np.random.default_rng(0)
for m, n in ((1024, 256), (4096, 1024), (16384, 2048)):
    A = rng.standard_normal((m, n)).astype(np.float32); b = rng.standard_normal(m).astype(np.float32)
    run(f"gaussian {m}x{n}", A, b)

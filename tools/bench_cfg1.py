#!/usr/bin/env python3
"""BASELINE config 1: Boston-housing Lasso (1000 x 5, the reference's own CPU case) through fista() on the GPU, next to
the CPU oracle on the same inputs: wall time of the whole call (500 iterations incl. the power iteration) and parity."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastoptsolver_amd as fos
from fastoptsolver_amd.easy_boston_data import generate_correlated_boston_like_data
from oracle import fos_oracle as orc

A, b, _ = generate_correlated_boston_like_data()
out = {}
for name, fn in (("gpu", lambda: fos.fista(A, b, "lasso", 1.0, 0.0, max_iter=500)),
                 ("cpu_oracle", lambda: orc.fista(A, b, "lasso", 1.0, 0.0, max_iter=500))):
    np.random.seed(0); fn()                      # warm-up
    ts = []
    for _ in range(5):
        np.random.seed(0)
        torch.cuda.synchronize(); t0 = time.perf_counter(); x = fn(); torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    out[name] = dict(wall_ms_median=1e3 * float(np.median(ts)), x=[float(v) for v in x])
out["rel_err_gpu_vs_cpu"] = float(np.linalg.norm(np.array(out["gpu"]["x"]) - np.array(out["cpu_oracle"]["x"])) /
                                  np.linalg.norm(out["cpu_oracle"]["x"]))
# the reference's own answer for this call (np.random.seed(0) before it)
out["reference_known_answer"] = [0.45313803, 0.32043257, 0.10464122, -0.14684924, 0.47167702]
out["rel_err_gpu_vs_reference_known_answer"] = float(
    np.linalg.norm(np.array(out["gpu"]["x"]) - np.array(out["reference_known_answer"])) /
    np.linalg.norm(out["reference_known_answer"]))
print(json.dumps(out))
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "bench_cfg1.json"), "w"), indent=1)

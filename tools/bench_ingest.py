#!/usr/bin/env python3
"""What the boundary costs when it is handed HOST arrays (the reference's callers pass float64 ndarrays): seconds to bind
A for each kind of input, and the whole fista(A, b, ...) call from ndarrays (upload + power iteration + 500 iterations +
read-back) against the resident loop."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastoptsolver_amd as fos

m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (65536, 8192)
torch.cuda.set_device(0)
rng = np.random.default_rng(0)
A32 = rng.standard_normal((m, n), dtype=np.float32)
b = rng.standard_normal(m)
torch.cuda.synchronize()


def timed(fn, reps=3):
    best = 1e30
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
        del out
    return best


gib = m * n * 4 / 2 ** 30
t = timed(lambda: fos.prepare(A32, b))
print(f"{m}x{n}: bind float32 ndarray ({gib:.1f} GiB): {t:.3f} s = {gib * 1.0737 / t:.1f} GB/s of fp32", flush=True)
A64 = A32.astype(np.float64)
t = timed(lambda: fos.prepare(A64, b))
print(f"{m}x{n}: bind float64 ndarray ({2 * gib:.1f} GiB on the host): {t:.3f} s = {gib * 1.0737 / t:.1f} GB/s of fp32", flush=True)
t = timed(lambda: fos.prepare(A64, b, dtype="bf16"))
print(f"{m}x{n}: bind float64 ndarray as bf16 storage: {t:.3f} s", flush=True)
At = torch.from_numpy(A32)
t = timed(lambda: fos.prepare(At, b))
print(f"{m}x{n}: bind float32 CPU tensor: {t:.3f} s", flush=True)
Af = np.asfortranarray(A32[:, : n // 2])
t = timed(lambda: fos.prepare(Af, b))
print(f"{m}x{n // 2}: bind Fortran-ordered float32 ndarray: {t:.3f} s", flush=True)
lam = float(np.max(np.abs(A32.T @ b.astype(np.float32))))
for name, arr in (("float32", A32), ("float64", A64)):
    np.random.seed(0)
    t = timed(lambda: fos.fista(arr, b, "lasso", 0.1 * lam, 0.0, max_iter=500), reps=2)
    print(f"{m}x{n}: fista(max_iter=500) from {name} ndarrays, whole call: {t:.3f} s = {500 / t:.0f} it/s PCIe-inclusive", flush=True)
prob = fos.prepare(A32, b)
np.random.seed(0)
t = timed(lambda: fos.fista(prob, None, "lasso", 0.1 * lam, 0.0, max_iter=500), reps=2)
print(f"{m}x{n}: the same call on the prepared problem (A resident): {t:.3f} s = {500 / t:.0f} it/s", flush=True)

#!/usr/bin/env python3
"""Where the host time of a warm cfg3 LBFGSSolver.fit goes: the native call alone against the whole fit (wall clock)."""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastoptsolver_amd as fos
from fastoptsolver_amd import _core, _lib, lbfgs
from bench import make_shard, WORKLOADS

torch.cuda.set_device(0)
cfg = WORKLOADS["cfg2"]
A, b = make_shard(cfg, 0, cfg["m"], torch.device("cuda", 0))
prob = fos.prepare(A, b)
for _ in range(2):
    fos.LBFGSSolver("ridge", 0.0, 1.0).fit(prob, None)
torch.cuda.synchronize()
lib = _lib.load()


def sync_t():
    torch.cuda.synchronize()
    return time.perf_counter()


for rep in range(3):
    t0 = sync_t()
    s = fos.LBFGSSolver("ridge", 0.0, 1.0).fit(prob, None)
    t1 = sync_t()
    # the native call alone, same arguments as _fit_native builds
    n, max_iter = prob.n_dev, 500
    x = torch.zeros(n, dtype=torch.float64, device="cuda")
    hist = (C.c_double * (2 * max_iter))()
    iterates = torch.empty(max_iter, n, dtype=torch.float64, device="cuda")
    cap = 21 * max_iter + 2
    fg_ms = (C.c_float * cap)()
    res = _lib.LbfgsResult()
    t2 = sync_t()
    with prob.ctx():
        _lib.check(lib.fos_lbfgs_minimize(prob.h, 1.0, max_iter, 1e-6, _core.ptr(x), hist, _core.ptr(iterates), fg_ms, cap, C.byref(res)))
    t3 = time.perf_counter()
    t4 = sync_t()
    with prob.ctx():
        _lib.check(lib.fos_lbfgs_minimize(prob.h, 1.0, max_iter, 1e-6, _core.ptr(x.zero_()), hist, None, None, 0, C.byref(res)))
    t5 = sync_t()
    fg_total = sum(fg_ms[i] for i in range(res.nfev))
    print(f"rep {rep}: whole fit {1e3 * (t1 - t0):.3f} ms | setup of the arguments {1e3 * (t2 - t1):.3f} ms | native call returns after "
          f"{1e3 * (t3 - t2):.3f} ms, drained {1e3 * (t4 - t2):.3f} ms | without iterates / timing {1e3 * (t5 - t4):.3f} ms | "
          f"sum of fg device times {fg_total:.3f} ms over {res.nfev} evaluations, nit {res.nit}", flush=True)

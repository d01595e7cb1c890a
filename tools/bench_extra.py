#!/usr/bin/env python3
"""Secondary measurements for DESIGN.md / profiles (not the headline bench):
  cfg3  L-BFGS (ridge, alpha2 = 1) on the cfg2 matrix: iterations, fg evaluations, fg/s, us per two-loop call
  cfg5' bf16 A / fp32 accumulate elastic-net step at 131072 x 16384 (one 8-GPU shard of cfg5)
  pcie  fista() end to end when the boundary is handed HOST ndarrays (upload + power iteration + 500 iterations)
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastoptsolver_amd as fos  # noqa: E402
from fastoptsolver_amd import _core, _lib  # noqa: E402
from bench import make_shard, WORKLOADS, bytes_per_iter  # noqa: E402


def timed_us(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    out = {}
    lib = _lib.load()

    # ---- cfg3: L-BFGS on the cfg2 matrix --------------------------------------------------------------
    cfg = WORKLOADS["cfg2"]
    A, b = make_shard(cfg, 0, cfg["m"], dev)
    prob = fos.prepare(A, b)
    fos.LBFGSSolver("ridge", 0.0, 1.0).fit(prob, None)          # warm-up (first launches, workspace allocation)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s = fos.LBFGSSolver("ridge", 0.0, 1.0).fit(prob, None)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    met = fos.get_metrics()
    n = cfg["n"]
    S = torch.randn(10, n, device=dev, dtype=torch.float64)
    Y = S + 0.1 * torch.randn(10, n, device=dev, dtype=torch.float64)
    g = torch.randn(n, device=dev, dtype=torch.float64)
    d = torch.empty(n, device=dev, dtype=torch.float64)
    two_loop = timed_us(lambda: lib.fos_lbfgs_two_loop_dd(_core.ptr(g), _core.ptr(S), _core.ptr(Y), 10, 0, 10, n,
                                                          _core.ptr(d), _core.stream_ptr()), 200)
    b_fg = cfg["m"] * n * 4 + 4 * cfg["m"] + 8 * n
    out["cfg3_lbfgs"] = dict(nit=s.nit_, nfev=s.nfev_, task=s.task_, wall_s=wall, it_per_s=s.nit_ / wall,
                             fg_per_s_wall=s.nfev_ / wall, fg_device_mean_us=met["grad_time_mean"] * 1e6,
                             fg_gbps=b_fg / met["grad_time_mean"] / 1e9, two_loop_us_hist10=two_loop,
                             final_obj=s.final_obj_)
    print(json.dumps(out["cfg3_lbfgs"]), flush=True)

    # ---- pcie-inclusive: host ndarrays handed over ------------------------------------------------------
    A_np = A.cpu().numpy()
    b_np = b.cpu().numpy()
    lam = float((A.T @ b).abs().max())
    del prob
    np.random.seed(0)
    t0 = time.perf_counter()
    x = fos.fista(A_np, b_np, "lasso", 0.1 * lam, 0.0, max_iter=500)
    wall = time.perf_counter() - t0
    out["pcie_inclusive_cfg2"] = dict(wall_s=wall, iters=500, it_per_s=500 / wall,
                                      note="fp32 host ndarray in; includes H2D copy of 2 GiB, 100-step power iteration, "
                                           "500 iterations, D2H of x")
    print(json.dumps(out["pcie_inclusive_cfg2"]), flush=True)
    del A, b, A_np, b_np
    torch.cuda.empty_cache()

    # ---- cfg5 shard: bf16 elastic-net step ----------------------------------------------------------------
    cfg = dict(WORKLOADS["cfg5"], m=131072)
    A, b = make_shard(cfg, 0, cfg["m"], dev)
    prob = fos.prepare(A, b)
    st = _core.Fista(prob)
    st.reset(1e-7, 100.0, 10.0)
    st.run(5)
    torch.cuda.synchronize()
    best = min(timed_us(lambda: st.run(50), 1) / 50 for _ in range(3))
    bi = bytes_per_iter(cfg["m"], cfg["n"], "bf16")
    prob.profile(1)
    prob.profile_read()
    st.run(50)
    ms, cnt = prob.profile_read()
    out["cfg5_shard_bf16"] = dict(m=cfg["m"], n=cfg["n"], step_us=best, step_gbps=bi / (best * 1e-6) / 1e9,
                                  step_frac=bi / (best * 1e-6) / 8e12, kernel_us=ms * 1e3 / cnt,
                                  kernel_frac=bi / (ms * 1e-3 / cnt) / 8e12, plan=prob.plan())
    print(json.dumps(out["cfg5_shard_bf16"]), flush=True)
    json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out",
                                     "bench_extra.json"), "w"), indent=1)


if __name__ == "__main__":
    main()

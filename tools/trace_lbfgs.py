#!/usr/bin/env python3
"""One warm-up and one timed L-BFGS fit on the cfg2 matrix (config 3) - run under `rocprofv3 --kernel-trace` to see
the launch sequence of an iteration and the idle gaps between its kernels (tools/trace_gaps.py summarises the CSV)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastoptsolver_amd as fos
from bench import make_shard, WORKLOADS

torch.cuda.set_device(0)
cfg = WORKLOADS["cfg2"]
A, b = make_shard(cfg, 0, cfg["m"], torch.device("cuda", 0))
prob = fos.prepare(A, b)
fos.LBFGSSolver("ridge", 0.0, 1.0).fit(prob, None)
torch.cuda.synchronize()
t0 = time.perf_counter()
s = fos.LBFGSSolver("ridge", 0.0, 1.0).fit(prob, None)
torch.cuda.synchronize()
print("fit wall ms", (time.perf_counter() - t0) * 1e3, "nit", s.nit_, "nfev", s.nfev_)

#!/usr/bin/env python3
"""Which feature makes the process die at exit under rocprofv3?  usage: exit_probe.py plain|cluster|twoproduct|lbfgs|mesh"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastoptsolver_amd as fos
from fastoptsolver_amd import _core
what = sys.argv[1]
torch.cuda.set_device(0)
m, n = 16384, 4096
A = torch.randn(m, n, device="cuda")
b = torch.randn(m, device="cuda")
prob = fos.prepare(A, b)
if what == "plain":
    st = _core.Fista(prob); st.reset(1e-6, 1.0, 0.0); st.run(10)
elif what in ("cluster", "twoproduct"):
    prob.replan(cluster=(what == "cluster"))
    hs = []
    for i in range(8):
        st = _core.Fista(prob); st.reset(1e-6, 1.0 + i, 0.0); hs.append(st)
    print("multi", _core.run_multi(hs, 5), prob.plan())
elif what == "lbfgs":
    s = fos.LBFGSSolver("ridge", 0.0, 1.0).fit(prob, None)
    print("nit", s.nit_)
elif what == "mesh":
    from fastoptsolver_amd import distributed as fd
    c = fd.Comm.solo(transport="mesh")
    t = torch.ones(1000, device="cuda"); c.allreduce(t); print(c.transport())
torch.cuda.synchronize()
print("done", what, flush=True)

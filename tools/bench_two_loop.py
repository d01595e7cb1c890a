#!/usr/bin/env python3
"""L-BFGS direction d = -H g: the one-workgroup two-loop kernel (fos_lbfgs_two_loop_dd) and the whole-chip form
(fos_lbfgs_direction_dd: Gram matrix + coefficient recursion), microseconds per call against history length and n
(HIP events over 200 back-to-back calls)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastoptsolver_amd import _core, _lib

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
lib = _lib.load()
out = {}
for n in (4096, 8192, 16384, 65536):
    S = torch.randn(20, n, device=dev, dtype=torch.float64)
    Y = S + 0.1 * torch.randn(20, n, device=dev, dtype=torch.float64)
    g = torch.randn(n, device=dev, dtype=torch.float64)
    d = torch.empty(n, device=dev, dtype=torch.float64)
    nwork = lib.fos_lbfgs_direction_work(n)
    work = torch.empty(nwork, device=dev, dtype=torch.float64)
    gd = torch.zeros(2, device=dev, dtype=torch.float64)
    for hist in (0, 1, 2, 5, 10, 20):
        def one_wg():
            _lib.check(lib.fos_lbfgs_two_loop_dd(_core.ptr(g), _core.ptr(S), _core.ptr(Y), hist, 0, 20, n, _core.ptr(d),
                                                 _core.stream_ptr()), "two_loop")

        def whole_chip():
            _lib.check(lib.fos_lbfgs_direction_dd(_core.ptr(g), _core.ptr(S), _core.ptr(Y), hist, 0, 20, n, _core.ptr(d),
                                                  _core.ptr(gd), _core.ptr(work), nwork, _core.stream_ptr()), "direction")
        for name, call in (("two_loop", one_wg), ("direction", whole_chip)):
            if name == "direction" and hist > 10:
                continue
            for _ in range(20):
                call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                call()
            e1.record(); e1.synchronize()
            out[f"{name}_n{n}_hist{hist}"] = e0.elapsed_time(e1) * 1e3 / 200
            print(name, n, hist, round(out[f"{name}_n{n}_hist{hist}"], 2), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/bench_two_loop.json", "w"), indent=1)

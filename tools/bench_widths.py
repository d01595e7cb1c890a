#!/usr/bin/env python3
"""The default FISTA step at widths that are not a power of two: microseconds per iteration, % of 8 TB/s on B_iter, and the
geometry the planner picked (threads x chunks: lanes beyond n idle).  python tools/bench_widths.py [f32|bf16]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastoptsolver_amd as fos
from fastoptsolver_amd import _core

kind = sys.argv[1] if len(sys.argv) > 1 else "f32"
dd = len(sys.argv) > 2 and sys.argv[2] == "dd"          # the fp64-accumulating pass (L-BFGS fg) instead of the FISTA step
esz = 4 if kind == "f32" else 2
torch.cuda.set_device(0)
widths = [2048, 2560, 3072, 3584, 4096, 5120, 6144, 7168, 8192, 10240, 12288, 14336, 16384]
if kind == "bf16":
    widths += [20480, 24576, 28672, 32768]
if "narrow" in sys.argv:
    widths = [72, 96, 128, 192, 256, 320, 384, 512, 640, 768, 1024, 1280, 1536, 1792, 2048]
for n in widths:
    m = (1 << 31) // (n * 4) // 256 * 256              # ~2 GiB of fp32 (1 GiB of bf16)
    g = torch.Generator(device="cuda").manual_seed(n)
    A = torch.randn(m, n, device="cuda", generator=g)
    if kind == "bf16":
        A = A.to(torch.bfloat16)
    b = torch.randn(m, device="cuda", generator=g)
    prob = fos.prepare(A, b)
    if os.environ.get("FOS_BENCH_IL"):
        prob.replan(interleave=os.environ["FOS_BENCH_IL"] == "1")
    if dd:
        from fastoptsolver_amd import _lib
        x64 = torch.randn(n, dtype=torch.float64, device="cuda")
        out64 = torch.zeros(n + 1, dtype=torch.float64, device="cuda")
        run = lambda k: [_lib.check(prob.lib.fos_gemv_pair_dd(prob.h, _core.ptr(x64), 0.5, _core.ptr(out64))) for _ in range(k)]
        st = None
    else:
        st = _core.Fista(prob); st.reset(1e-9, 1.0, 0.0)
        run = st.run
    run(10); torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(50); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 20.0)
    byt = m * n * esz + 4 * m + (8 if dd else 16) * n
    pl = prob.plan()
    cap = pl["threads"] * pl["chunks"] * (16 // esz)
    print(f"{kind}{' fg(fp64)' if dd else ''} {m}x{n}: {best:7.1f} us = {byt / best / 8e4:5.1f} % of 8 TB/s   plan {pl['threads']}x{pl['chunks']} rows {pl['rows']} "
          f"(capacity {cap}, {100.0 * n / cap:.0f} % of the lanes busy), {pl['workgroups']} wg", flush=True)
    del st, prob, A, b
    torch.cuda.empty_cache()

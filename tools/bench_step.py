#!/usr/bin/env python3
"""Per-iteration device time of the fused FISTA step vs the bare A pass (tuning aid, not the official bench)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastoptsolver_amd as fos  # noqa: E402
from fastoptsolver_amd import _core  # noqa: E402


def timed(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn(iters)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=65536)
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--geoms", type=str, default="")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    g = torch.Generator(device="cuda").manual_seed(0)
    A = torch.randn(args.m, args.n, device="cuda", generator=g)
    if args.bf16:
        A = A.to(torch.bfloat16)
    b = torch.randn(args.m, device="cuda", generator=g)
    prob = fos.prepare(A, b)
    esz = 2 if args.bf16 else 4
    bytes_iter = args.m * args.n * esz + 4 * args.m + 16 * args.n
    geoms = [None] + [tuple(int(v) for v in s.split("x")) for s in args.geoms.split(",") if s]
    y = torch.randn(args.n, device="cuda", generator=g)
    out = torch.empty(args.n, device="cuda")
    for geom in geoms:
        if geom is not None:
            prob.tune(*geom)
        st = _core.Fista(prob)
        st.reset(1e-6, 1.0, 0.0)
        st.run(10)
        torch.cuda.synchronize()
        res = []
        for rep in range(3):
            t_full = timed(lambda k: st.run(k), args.iters)
            t_pass = timed(lambda k: [prob.gemv_pair(y, 0.0, out=out) for _ in range(k)], args.iters)
            t_split = timed(lambda k: [(st.grad(), st.update()) for _ in range(k)], args.iters)
            res.append((t_full, t_pass, t_split))
        t_full, t_pass, t_split = (min(r[i] for r in res) for i in range(3))
        t0 = time.perf_counter()
        st.run(args.iters)
        host_us = (time.perf_counter() - t0) * 1e6 / args.iters
        torch.cuda.synchronize()
        print(f"plan {prob.plan()}  fused step {t_full:.1f} us ({bytes_iter / (t_full * 1e-6) / 1e9:.0f} GB/s, "
              f"{bytes_iter / (t_full * 1e-6) / 8e12 * 100:.1f}% of 8 TB/s)  gemv_pair call {t_pass:.1f} us  "
              f"grad+update split {t_split:.1f} us  host enqueue {host_us:.1f} us/iter", flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Workgroup-count sweep of the narrow / tall plans on the WHOLE step (A pass + update, which sums one slab per workgroup)
and on the A pass alone: fos_problem_tune with 256 ... 4096 workgroups against the planner's choice.  python tools/wg_sweep.py [dd]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastoptsolver_amd as fos
from fastoptsolver_amd import _core, _lib
torch.cuda.set_device(0)
shapes = [(838656, 640), (698880, 768), (419328, 1280), (349440, 1536), (209664, 2560), (149760, 3584), (74752, 7168), (37376, 14336)] if "mid" in sys.argv else [(4194304, 128), (5592320, 96), (7456512, 72), (8000000, 64), (16000000, 32), (12000000, 40)] if "big" in sys.argv else [(4000000, 32), (2000000, 64), (1000000, 128), (2000000, 96), (8000000, 5), (4000000, 16), (32000000, 5), (300000, 64), (200000, 100), (500000, 8),
          (2796032, 160), (2097152, 224), (1398016, 320), (1398016, 448)]
for m, n in shapes:
    A = torch.randn(m, n, device="cuda"); b = torch.randn(m, device="cuda")
    prob = fos.prepare(A, b)
    pl = prob.plan()
    base = pl["workgroups"]
    cands = sorted(set([256, 512, 768, 1024, 1536, 2048, 3072, 4096, base])) if 'mid' not in sys.argv else sorted(set([128, 256, 384, 512, 768, 1024, 2048, base]))
    for wg in cands:
        try:
            prob.tune(pl["threads"], pl["chunks"], pl["rows"], wg)
        except Exception as e:
            print("tune failed", wg, str(e)[:80]); continue
        st = _core.Fista(prob); st.reset(1e-9, 1.0, 0.0); st.run(10); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); st.run(50); e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) * 20)
        prob.profile(1); prob.profile_read()
        for _ in range(20):
            st.run(1)
        ms, cnt = prob.profile_read(); prob.profile(0)
        tag = " <- planner" if prob.plan()["workgroups"] == base and wg == base else ""
        print(f"{m}x{n} [{pl['threads']}x{pl['chunks']} tall={pl['tall']}] wg {prob.plan()['workgroups']}: whole step {best:.1f} us = {(m*n*4)/best/8e4:.1f} %; A pass alone {ms*1e3/cnt:.1f} us{tag}", flush=True)
        del st
    del prob, A, b
    torch.cuda.empty_cache()

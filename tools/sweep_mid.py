import os, sys, torch
sys.path.insert(0, "/root/repo")
import fastoptsolver_amd as fos
from fastoptsolver_amd import _core
torch.cuda.set_device(0)
for m, n, geos in ((20000, 256, [(64, 1, 4, w) for w in (2500, 1250, 625, 312, 156)]),
                   (4096, 512, [(64, 2, 4, w) for w in (512, 256, 128, 64)] + [(256, 1, 4, w) for w in (512, 256, 128, 64)]),
                   (4096, 1024, [(256, 1, 4, w) for w in (512, 256, 128, 64)]),
                   (16384, 2048, [(256, 2, 4, w) for w in (512, 256, 128)]),
                   (100000, 128, [(64, 1, 4, w) for w in (4096, 2048, 1024, 512, 256)])):
    A = torch.randn(m, n, device="cuda"); b = torch.randn(m, device="cuda")
    prob = fos.prepare(A, b)
    print(m, n, "default plan", prob.plan()["threads"], prob.plan()["chunks"], prob.plan()["workgroups"])
    for th, k, r, w in geos:
        try:
            prob.tune(th, k, r, w)
        except Exception as e:
            print("  ", th, k, r, w, "unsupported"); continue
        st = _core.Fista(prob); st.reset(1e-9, 1.0, 0.0); st.run(10); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); st.run(200); e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / 200)
        print(f"   {th}x{k}x{r} wg {prob.plan()['workgroups']:5d}: {best:7.2f} us per iteration", flush=True)

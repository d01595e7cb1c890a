#!/usr/bin/env python3
"""Per-iteration cost of the sharded code path on one GPU: the config-4 shard (131072 x 16384 fp32) through the fused
single-GPU loop, through the C-ABI sharded loop with a ONE-rank RCCL communicator (every kernel and the ncclAllReduce of
the real path, no peers: what is left out is the link latency of the 64 KiB exchange), and with the mesh transport."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastoptsolver_amd import distributed as fd
from bench import make_shard, WORKLOADS

torch.cuda.set_device(0)
cfg = dict(WORKLOADS["cfg4"], m=131072)
A, b = make_shard(cfg, 0, cfg["m"], torch.device("cuda", 0))
out = {}
for name, comm in (("fused single-GPU loop", None), ("sharded loop, 1-rank RCCL", fd.Comm.solo()),
                   ("sharded loop, 1-rank mesh kernel", fd.Comm.solo("mesh"))):
    eng = fd.HipShardEngine(A, b, comm=comm)
    eng.reset(tau=1e-9, alpha1=1.0, alpha2=0.0)
    run = eng.run if comm is not None else eng.st.run
    run(5)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(50); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 50)
    out[name] = best
    print(f"{name:36s} {best:8.1f} us per iteration", flush=True)
    del eng
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "bench_sharded_overhead.json"), "w"), indent=1)

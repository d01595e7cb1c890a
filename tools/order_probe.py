#!/usr/bin/env python3
"""Why does cfg2 measure ~301 us per pass alone and ~325 us in bench.py after cfg4?  One process:
 (a) cfg2 problem on a fresh device;  (b) the SAME allocation re-measured after 64 GiB were allocated, written and freed;
 (c) a NEW cfg2 allocation made after that;  (d) the same after torch.cuda.empty_cache() + a new allocation again;
 each with contiguous blocks and with interleaved rows."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastoptsolver_amd as fos
from fastoptsolver_amd import _core
from bench import make_shard, WORKLOADS

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
cfg = WORKLOADS["cfg2"]


def measure(tag, A, b):
    for il in (False, True):
        prob = fos.prepare(A, b)
        prob.replan(interleave=il)
        st = _core.Fista(prob)
        st.reset(1e-6, 1.0, 0.0)
        st.run(10)
        torch.cuda.synchronize()
        out = []
        for rep in range(3):
            prob.profile(1); prob.profile_read()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); st.run(200); e1.record(); e1.synchronize()
            ms, cnt = prob.profile_read(); prob.profile(0)
            out.append((ms * 1e3 / cnt, e0.elapsed_time(e1) * 1e3 / 200))
        print(f"{tag:48s} il={int(il)} A@{A.data_ptr():#x}: kernel us " + " ".join(f"{k:.1f}" for k, _ in out) +
              "  step us " + " ".join(f"{s:.1f}" for _, s in out), flush=True)
        del st, prob


A, b = make_shard(cfg, 0, cfg["m"], dev)
measure("(a) fresh device", A, b)
big = torch.empty(64 * 2**30 // 4, dtype=torch.float32, device=dev)
big.normal_()
torch.cuda.synchronize()
measure("(b) same allocation, 64 GiB live", A, b)
del big
torch.cuda.empty_cache()
measure("(b2) same allocation, 64 GiB freed", A, b)
A2, b2 = make_shard(cfg, 0, cfg["m"], dev)
measure("(c) new allocation after the 64 GiB were freed", A2, b2)
del A, b
torch.cuda.empty_cache()
A3, b3 = make_shard(cfg, 0, cfg["m"], dev)
measure("(d) another new allocation", A3, b3)
measure("(c') the (c) allocation again", A2, b2)
# the bench's own sequence: cfg4 problem created, run, destroyed; then cfg2
del A2, b2, A3, b3
torch.cuda.empty_cache()
c4 = WORKLOADS["cfg4"]
A4, b4 = make_shard(c4, 0, c4["m"], dev)
p4 = fos.prepare(A4, b4); s4 = _core.Fista(p4); s4.reset(1e-7, 1.0, 0.0); s4.run(50); torch.cuda.synchronize()
del s4, p4, A4, b4
torch.cuda.empty_cache()
A5, b5 = make_shard(cfg, 0, cfg["m"], dev)
measure("(e) cfg2 after a cfg4 problem lived and died", A5, b5)
time.sleep(5.0)
measure("(e') same after 5 s idle", A5, b5)

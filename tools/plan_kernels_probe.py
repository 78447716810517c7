"""Builds the HIP plan of one batch shape a few times; run under `rocprofv3 --kernel-trace --stats` to
see every kernel of a plan build, rocPRIM's sorts and scans included.
usage: python tools/plan_kernels_probe.py <graphs> [reps]     (graphs of 10k hits / 100k segments)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_fpga_amd import HitGraphBatch, synth, _lib
from gnn_fpga_amd.plan_hip import HipSellPlan

G = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
lim = _lib.plan_limits(3, 8)
b = HitGraphBatch.from_graphs([synth.layered_graph(10000, 100000, 3, seed=s) for s in range(G)]).cuda()
HipSellPlan(b, lim)
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    HipSellPlan(b, lim)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("c3 x %d: plan build ms %s (min %.3f)" % (G, " ".join("%.3f" % t for t in ts), min(ts)))

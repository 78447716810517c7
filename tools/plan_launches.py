"""Launch sequence of one plan build (HIP builder) with per-launch times: tools/plan_launches.py [G]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_fpga_amd import HitGraphBatch, synth, _lib
from gnn_fpga_amd.plan_hip import HipSellPlan
G = int(sys.argv[1]) if len(sys.argv) > 1 else 1
b = HitGraphBatch.from_graphs([synth.layered_graph(10000, 100000, 3, seed=s) for s in range(G)]).cuda()
lim = _lib.plan_limits(3, 8)
for _ in range(3): HipSellPlan(b, lim)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): HipSellPlan(b, lim)
torch.cuda.synchronize(); print("plan build %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
with _lib.profile(512) as prof:
    HipSellPlan(b, lim)
tot = 0.0
for k, v in prof.records:
    print("  %-22s %7.1f us" % (k, v * 1e3)); tot += v
print("named launches: %d, %.3f ms of kernel time" % (len(prof.records), tot))

"""Execution-plan build time on the GPU for the benchmark batch (second build: kernels loaded)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_fpga_amd import HitGraphBatch, synth, _lib
from gnn_fpga_amd.plan_device import DeviceSellPlan

G = int(sys.argv[1]) if len(sys.argv) > 1 else 256
b = HitGraphBatch.from_graphs([synth.layered_graph(10000, 100000, 3, seed=s) for s in range(G)]).cuda()
lim = _lib.plan_limits(3, 8)
for i in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    p = DeviceSellPlan(b, lim)
    torch.cuda.synchronize(); print("build %d: %.3f s (%d segments, %d tiles)" % (i, time.perf_counter() - t, b.n_segments, p.n_tiles))
torch.cuda.synchronize(); t = time.perf_counter()
b._ensure_csr()
torch.cuda.synchronize(); print("two CSRs: %.3f s" % (time.perf_counter() - t))

"""Where the wall clock of a plan build on a NEVER-SEEN batch goes beyond its kernels (bench.py `plan_ms.warm`): the two
library calls (stage 1 incl. the sizes read-back, stage 2), and the rest (uploads, allocations, Python).
usage: python tools/plan_host_probe.py [graphs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_fpga_amd import HitGraphBatch, synth, _lib
from gnn_fpga_amd.model import SegmentClassifier
from gnn_fpga_amd import plan_hip

G = int(sys.argv[1]) if len(sys.argv) > 1 else 256
graphs = [synth.layered_graph(10000, 100000, 3, seed=s) for s in range(G)]
dev = torch.device("cuda:0")
model = SegmentClassifier(3, 8, 3).to(dev).eval()
batch = HitGraphBatch.from_graphs(graphs).to(dev)
with torch.no_grad():
    model(batch); model(batch)
T = {}
def wrap(name):
    f = getattr(_lib, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); T[name] = T.get(name, 0.0) + time.perf_counter() - t0
        return r
    setattr(_lib, name, g)
wrap("plan_build_sizes"); wrap("plan_build_fill"); wrap("plan_build_workspace_bytes")
for rep in range(4):
    fresh = HitGraphBatch.from_graphs(graphs).to(dev)
    with torch.no_grad():
        for _ in range(20):
            model(batch)
    torch.cuda.synchronize()
    T.clear()
    t0 = time.perf_counter()
    fresh.build_plan(8)
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print("build %d: total %.3f ms; stage 1 call + read-back %.3f, stage 2 call %.3f, rest (uploads, allocations, Python, the extra synchronisations of this probe) %.3f"
          % (rep, tot * 1e3, T["plan_build_sizes"] * 1e3, T["plan_build_fill"] * 1e3,
             (tot - T["plan_build_sizes"] - T["plan_build_fill"]) * 1e3))

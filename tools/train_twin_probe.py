"""Experiment: training step on a level-ordered copy of the batch (hits renumbered in plan order)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gnn_fpga_amd import HitGraphBatch, synth
from gnn_fpga_amd.model import SegmentClassifier
from gnn_fpga_amd.loss import BCELoss

G = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
graphs = [synth.layered_graph(10000, 100000, 3, seed=s) for s in range(G)]
batch = HitGraphBatch.from_graphs(graphs).to(dev)
plan = batch.build_plan(8)
perm = plan.perm.long()
order = perm[perm >= 0]
rank = torch.empty(batch.n_hits, dtype=torch.long, device=dev)
rank[order] = torch.arange(batch.n_hits, device=dev)
src, dst = batch.src.long(), batch.dst.long()
twin = HitGraphBatch.__new__(HitGraphBatch)
twin.__dict__.update(batch.__dict__)
twin.X = batch.X[order].contiguous()
twin.src = torch.where(src >= 0, rank[src.clamp_min(0)], src).to(torch.int32)
twin.dst = torch.where(dst >= 0, rank[dst.clamp_min(0)], dst).to(torch.int32)
twin._csr = None; twin.plan = None; twin._gstruct = None; twin._src_host = twin._dst_host = None; twin._event = None
y = (torch.rand(batch.n_segments, device=dev) < 0.3).float()
torch.manual_seed(0)
m = SegmentClassifier(input_dim=3, hidden_dim=8, n_iters=3).to(dev).train()
bce = BCELoss()
def timeit(b, n=20):
    def step():
        m.zero_grad(); loss = bce(m(b), y); loss.backward(); return loss
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): l = step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, float(l), [p.grad.clone() for p in m.parameters()]
ta, la, ga = timeit(batch)
tb, lb, gb = timeit(twin)
print("original order %.3f ms (loss %.6f)   level order %.3f ms (loss %.6f)   max grad diff %.2e"
      % (ta, la, tb, lb, max(float((a - b).abs().max()) for a, b in zip(ga, gb))))

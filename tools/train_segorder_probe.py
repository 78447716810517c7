"""Experiment: training step on the level-ordered twin with its SEGMENTS reordered too (sorted by end
hit, then start hit: the in-lists' scores become contiguous, the edge pass gathers nearly sequentially)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_fpga_amd import HitGraphBatch, synth
from gnn_fpga_amd.model import SegmentClassifier
from gnn_fpga_amd.loss import BCELoss

G = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
graphs = [synth.layered_graph(10000, 100000, 3, seed=s) for s in range(G)]
batch = HitGraphBatch.from_graphs(graphs).to(dev)
twin = batch.level_ordered(8)
src, dst = twin.src.long(), twin.dst.long()
key = torch.where(src >= 0, dst * (twin.n_hits + 1) + src, torch.full_like(src, 2 ** 62))
order = torch.argsort(key, stable=True)
t2 = HitGraphBatch.__new__(HitGraphBatch)
t2.__dict__.update(twin.__dict__)
t2.src = twin.src[order].contiguous()
t2.dst = twin.dst[order].contiguous()
t2._csr = None; t2._gstruct = None; t2._src_host = t2._dst_host = None; t2._twin = t2; t2.plan = None
y = (torch.rand(batch.n_segments, device=dev) < 0.3).float()
y2 = y[order].contiguous()
torch.manual_seed(0)
m = SegmentClassifier(input_dim=3, hidden_dim=8, n_iters=3).to(dev).train()
m.level_order_training = False           # the batches below are used as given
bce = BCELoss()
def timeit(b, yy, n=30):
    def step():
        m.zero_grad(); loss = bce(m(b), yy); loss.backward(); return loss
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): l = step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, float(l), [p.grad.clone() for p in m.parameters()]
ta, la, ga = timeit(twin, y)
tb, lb, gb = timeit(t2, y2)
print("twin (caller's segment order) %.3f ms (loss %.6f)   twin + segments by end hit %.3f ms (loss %.6f)   max grad diff %.2e"
      % (ta, la, tb, lb, max(float((a - b).abs().max()) for a, b in zip(ga, gb))))

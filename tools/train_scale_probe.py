"""Training step at tracking-detector scale (c3 graphs: 10k hits / 100k segments, F=3 D=8 T=3):
HIP forward (keeps e_t, H_t) + fused BCE + HIP backward + Adam, G graphs per batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_fpga_amd import HitGraphBatch, synth
from gnn_fpga_amd.model import SegmentClassifier
from gnn_fpga_amd.loss import BCELoss

G = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
graphs = [synth.layered_graph(10000, 100000, 3, seed=s, sort_hits_by_layer=bool(int(os.environ.get("SORTED", "0")))) for s in range(G)]
batch = HitGraphBatch.from_graphs(graphs).to(dev)
y = (torch.rand(batch.n_segments, device=dev) < 0.3).float()
torch.manual_seed(0)
m = SegmentClassifier(input_dim=3, hidden_dim=8, n_iters=3).to(dev).train()
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
bce = BCELoss()

def fwd_only():
    with torch.no_grad():
        return m(batch)

def step():
    opt.zero_grad(set_to_none=False)
    loss = bce(m(batch), y)
    loss.backward()
    opt.step()
    return loss

def timeit(fn, n):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

tf = timeit(fwd_only, steps)
ts = timeit(step, steps)

from gnn_fpga_amd import shard
bucket = shard.GradBucket(m.parameters())
def direct():                       # no autograd graph: the backward adds straight into the bucket's views
    bucket.zero()
    loss = bucket.step(m, batch, y)
    opt.step()
    return loss
td = timeit(direct, steps)
print("c3 x %d graphs (%d segments): inference forward %.3f ms (%.3g seg/s); training step %.3f ms (%.3g seg/s), "
      "GradBucket.step %.3f ms (%.3g seg/s), loss %.4f"
      % (G, batch.n_segments, tf * 1e3, batch.n_segments / tf, ts * 1e3, batch.n_segments / ts,
         td * 1e3, batch.n_segments / td, float(step())))

"""Training on NEVER-SEEN detector-size batches: the step in the caller's hit order (what `level_order_training = "auto"`
runs the first time it sees a batch) against the step on the level-ordered twin (second time on), and what building the
twin costs - i.e. whether a never-repeated batch should get its twin at once.
usage: python tools/twin_probe.py [graphs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_fpga_amd import HitGraphBatch, synth, shard
from gnn_fpga_amd.model import SegmentClassifier

G = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
graphs = [synth.layered_graph(10000, 100000, 3, seed=s) for s in range(G)]
torch.manual_seed(0)
m = SegmentClassifier(input_dim=3, hidden_dim=8, n_iters=3).to(dev).train()
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
bucket = shard.GradBucket(m.parameters())


def step(batch, y):
    bucket.zero()
    loss = bucket.step(m, batch, y)
    opt.step()
    return loss


def sync_time(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, r


for policy in (False, True):
    m.level_order_training = policy
    ts = []
    for rep in range(5):
        b = HitGraphBatch.from_graphs(graphs).to(dev)
        y = (torch.rand(b.n_segments, device=dev) < 0.3).float()
        t1, _ = sync_time(lambda: step(b, y))          # first step on a never-seen batch (twin / plan built inside)
        t2, _ = sync_time(lambda: step(b, y))
        t3, _ = sync_time(lambda: step(b, y))
        ts.append((t1, t2, t3))
    ts = ts[1:]
    print("level_order_training = %-5s  c3 x %d: first step on a never-seen batch %.3f ms, second %.3f, third %.3f (means of 4)"
          % (policy, G, sum(t[0] for t in ts) / 4, sum(t[1] for t in ts) / 4, sum(t[2] for t in ts) / 4))
b = HitGraphBatch.from_graphs(graphs).to(dev)
t_plan, _ = sync_time(lambda: b.build_plan(8))
t_twin, _ = sync_time(lambda: b.level_ordered(8))
print("plan %.3f ms, twin on top of it %.3f ms" % (t_plan, t_twin))

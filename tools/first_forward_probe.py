"""What the FIRST forward on a freshly planned batch costs beyond a steady-state one (bench.py `fresh_batch_forward_ms`
against `ms_per_step`): its launches by HIP events and the synchronised wall clock of the first three forwards.
usage: python tools/first_forward_probe.py [graphs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_fpga_amd import HitGraphBatch, synth, _lib
from gnn_fpga_amd.model import SegmentClassifier

G = int(sys.argv[1]) if len(sys.argv) > 1 else 256
graphs = [synth.layered_graph(10000, 100000, 3, seed=s) for s in range(G)]
dev = torch.device("cuda:0")
model = SegmentClassifier(3, 8, 3).to(dev).eval()
batch = HitGraphBatch.from_graphs(graphs).to(dev)
batch.build_plan(8)
with torch.no_grad():
    for _ in range(30):
        model(batch)
    for rep in range(3):
        fresh = HitGraphBatch.from_graphs(graphs).to(dev)
        fresh.build_plan(8)
        for _ in range(20):
            model(batch)
        torch.cuda.synchronize()
        ts = []
        for k in range(3):
            t0 = time.perf_counter()
            model(fresh)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        print("fresh plan %d: forward 1 / 2 / 3 = %.3f / %.3f / %.3f ms" % (rep, ts[0], ts[1], ts[2]))
    fresh = HitGraphBatch.from_graphs(graphs).to(dev)
    fresh.build_plan(8)
    torch.cuda.synchronize()
    with _lib.profile(64) as prof:
        model(fresh)
    print("first forward launches:", ", ".join("%s %.3f" % kv for kv in prof.records))
    with _lib.profile(64) as prof:
        model(fresh)
    print("second forward launches:", ", ".join("%s %.3f" % kv for kv in prof.records))

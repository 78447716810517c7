"""Batch assembly: HitGraphBatch.from_graphs on the host + upload, against batcher.GraphStore (the dataset resident in
HBM, batches assembled on the device), then plan + forward of each - a stream of never-seen batches end to end.
usage: python tools/store_probe.py [graphs in the dataset] [batch size]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_fpga_amd import HitGraphBatch, synth, GraphStore
from gnn_fpga_amd.model import SegmentClassifier

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
BS = int(sys.argv[2]) if len(sys.argv) > 2 else 256
graphs = [synth.layered_graph(10000, 100000, 3, seed=s) for s in range(N)]
dev = torch.device("cuda:0")
model = SegmentClassifier(3, 8, 3).to(dev).eval()
model.use_plan = True
t0 = time.perf_counter()
store = GraphStore(graphs, device=dev)
torch.cuda.synchronize()
t_store = time.perf_counter() - t0
E = BS * 100000
with torch.no_grad():
    for rep in range(2):                                # (second pass: host allocator and code objects warm)
        t_host = t_up = t_dev = t_fwd = 0.0
        for j in range(0, N, BS):
            t0 = time.perf_counter()
            b = HitGraphBatch.from_graphs(graphs[j:j + BS])
            t1 = time.perf_counter()
            b = b.to(dev)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            model(b)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            g, _ = store.batch(j, BS, "flat")
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            model(g)
            torch.cuda.synchronize()
            t5 = time.perf_counter()
            t_host += t1 - t0; t_up += t2 - t1; t_fwd += (t3 - t2 + t5 - t4) / 2; t_dev += t4 - t3
        nb = (N + BS - 1) // BS
print("dataset of %d detector graphs resident in HBM (built and uploaded once: %.2f s); per batch of %d (%.1f M segments): "
      "from_graphs on the host %.1f ms + upload %.1f ms, GraphStore.batch on the device %.2f ms; plan + forward %.2f ms -> "
      "%.2e segments/s end to end from the store, %.2e through the host"
      % (N, t_store, BS, E / 1e6, t_host / nb * 1e3, t_up / nb * 1e3, t_dev / nb * 1e3, t_fwd / nb * 1e3,
         E / ((t_dev + t_fwd) / nb), E / ((t_host + t_up + t_fwd) / nb)))

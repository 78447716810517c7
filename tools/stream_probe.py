"""A STREAM of never-repeated batches (what inference on real events is): per batch a plan build and one forward.
Serial (plan, then forward, one stream) against overlapped (the forward of batch k on one HIP stream while the plan of
batch k + 1 is built on another; the host blocks only in the plan's size read-back).  Index arrays are device-resident
before the clock starts (host-side batch assembly is not part of either number).
usage: python tools/stream_probe.py [graphs per batch] [batches]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_fpga_amd import HitGraphBatch, synth
from gnn_fpga_amd.model import SegmentClassifier

G = int(sys.argv[1]) if len(sys.argv) > 1 else 256
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 8
graphs = [synth.layered_graph(10000, 100000, 3, seed=s) for s in range(G)]
dev = torch.device("cuda:0")
model = SegmentClassifier(3, 8, 3).to(dev).eval()
model.use_plan = True
E = sum(len(g.src) for g in graphs)


def fresh(k):
    return [HitGraphBatch.from_graphs(graphs).to(dev) for _ in range(k)]


with torch.no_grad():
    w = fresh(2)
    for b in w:
        b.build_plan(8); model(b)
    torch.cuda.synchronize()
    # serial
    bs = fresh(NB)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = []
    for b in bs:
        b.build_plan(8)
        outs.append(model(b))
    torch.cuda.synchronize()
    t_ser = (time.perf_counter() - t0) / NB
    ref = [o.clone() for o in outs]
    del bs, outs
    # overlapped: forward(k) on F while plan(k + 1) on P
    bs = fresh(NB)
    F, P = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = []
    with torch.cuda.stream(P):
        bs[0].build_plan(8)
    P.synchronize()
    for k in range(NB):
        with torch.cuda.stream(F):
            outs.append(model(bs[k]))
        if k + 1 < NB:
            with torch.cuda.stream(P):
                bs[k + 1].build_plan(8)
        F.synchronize(); P.synchronize()
    t_ovl = (time.perf_counter() - t0) / NB
    ok = all(torch.equal(a, b) for a, b in zip(ref, outs))
print("c3 x %d, %d never-seen batches: serial %.3f ms per batch (%.2e segments/s), plan of the next batch under the "
      "forward of this one %.3f ms (%.2e segments/s); scores identical: %s"
      % (G, NB, t_ser * 1e3, E / t_ser, t_ovl * 1e3, E / t_ovl, ok))

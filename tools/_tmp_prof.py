import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_fpga_amd import HitGraphBatch, synth, _lib
from gnn_fpga_amd.model import SegmentClassifier
for name, graphs, F in (("muon", [synth.muon_graph(3)], 11), ("c3", [synth.layered_graph(10000, 100000, 3, seed=0)], 3)):
    m = SegmentClassifier(input_dim=F, hidden_dim=8, n_iters=3).cuda().eval()
    bs = [HitGraphBatch.from_graphs(graphs).cuda() for _ in range(60)]
    with torch.no_grad():
        for b in bs[:10]:
            m(b)
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        pr.enable()
        for b in bs[10:]:
            m(b)
            torch.cuda.synchronize()
        pr.disable()
    print("=====", name)
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)

"""Per-kernel times of one shape's forward on c3-size graphs: tools/shape_profile.py F D T [G]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_fpga_amd import HitGraphBatch, synth, _lib
from gnn_fpga_amd.model import SegmentClassifier
F, D, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
G = int(sys.argv[4]) if len(sys.argv) > 4 else 32
b = HitGraphBatch.from_graphs([synth.layered_graph(10000, 100000, F, seed=s) for s in range(G)]).cuda()
m = SegmentClassifier(input_dim=F, hidden_dim=D, n_iters=T).cuda().eval()
m.use_events = False
with torch.no_grad():
    for _ in range(5): m(b)
    with _lib.profile(256) as prof:
        m(b)
for k, v in prof.records: print("  %-14s %8.1f us" % (k, v * 1e3))
p = b.plan
print("tiles %d, lds tiles %d, tile_hits_max %s, iter_lds_in/out %s/%s, padding %.1f %%" % (p.n_tiles, p.n_lds_tiles, getattr(p, "tile_hits_max", "?"), getattr(p, "iter_lds_in", "?"), getattr(p, "iter_lds_out", "?"), 100 * p.padding))

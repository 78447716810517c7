"""How wide are the neighbour id ranges of 256 consecutive hits of the level-ordered twin?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gnn_fpga_amd import HitGraphBatch, synth
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
b = HitGraphBatch.from_graphs([synth.layered_graph(10000, 100000, 3, seed=s) for s in range(G)]).cuda()
t = b.level_ordered(8)
src, dst = t.src.cpu().numpy().astype(np.int64), t.dst.cpu().numpy().astype(np.int64)
ok = src >= 0
src, dst = src[ok], dst[ok]
n = t.n_hits
nb = (n + 255) // 256
for name, own, far in (("out (far = end hits)", src, dst), ("in (far = start hits)", dst, src)):
    blk = own // 256
    lo = np.full(nb, 1 << 60); hi = np.full(nb, -1)
    np.minimum.at(lo, blk, far); np.maximum.at(hi, blk, far)
    w = (hi - lo + 1)[hi >= 0]
    print("%-24s blocks %d  width: median %d  p90 %d  max %d  fit 1450: %.1f %%" % (name, len(w), np.median(w), np.percentile(w, 90), w.max(), 100 * np.mean(w <= 1450)))

"""Looking for slow corners: the same batch through the one-launch event kernels and the tiled pipeline /
per-pass training kernels, for graph sizes between the muon events and the detector graphs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_fpga_amd import HitGraphBatch, synth
from gnn_fpga_amd.model import SegmentClassifier
from gnn_fpga_amd.loss import BCELoss

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

CASES = ((3, 8, 3, 300, 2000, 256), (3, 8, 3, 1000, 8000, 64), (3, 8, 3, 2500, 20000, 32),
         (11, 8, 3, 500, 3000, 128), (3, 16, 3, 10000, 100000, 8), (3, 32, 3, 1000, 8000, 64))
if len(sys.argv) > 1 and sys.argv[1] == "small":
    CASES = ((3, 8, 3, 40, 200, 512), (3, 8, 3, 80, 500, 512), (3, 8, 3, 150, 1000, 256), (3, 8, 3, 150, 1000, 32),
             (3, 8, 3, 300, 2000, 32), (3, 8, 3, 300, 2000, 4), (11, 8, 3, 100, 400, 512), (11, 8, 3, 200, 1200, 128))
for F, D, T, nh, ns, G in CASES:
    b = HitGraphBatch.from_graphs([synth.layered_graph(nh, ns, F, seed=s) for s in range(G)]).cuda()
    y = (torch.rand(b.n_segments, device="cuda") < 0.3).float()
    m = SegmentClassifier(input_dim=F, hidden_dim=D, n_iters=T).cuda()
    bce = BCELoss()
    res = {}
    for ev in (True, False):
        m.use_events = ev
        m.eval()
        with torch.no_grad():
            res["fwd", ev] = t(lambda: m(b))
        m.train()
        def step():
            m.zero_grad(); bce(m(b), y).backward()
        res["train", ev] = t(step, 10)
    print("F=%2d D=%2d  %3d x (%5d hits, %6d segs): forward events %.3f / tiled %.3f ms   train step events %.3f / per-pass %.3f ms"
          % (F, D, G, nh, ns, res["fwd", True], res["fwd", False], res["train", True], res["train", False]))

"""Training step on a batch of the reference's toy graphs (gnn/MPNN_Seg_Toy2D.ipynb: 40 hits, 144 segments, F=2, D=32, T=10)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_fpga_amd import HitGraphBatch, synth
from gnn_fpga_amd.model import SegmentClassifier
from gnn_fpga_amd.loss import BCELoss
for G in (1, 32, 256):
    graphs = [synth.toy2d_graph(seed=s) if hasattr(synth, "toy2d_graph") else synth.layered_graph(40, 144, 2, seed=s) for s in range(G)]
    b = HitGraphBatch.from_graphs(graphs).cuda()
    y = b.y.cuda() if b.y is not None else (torch.rand(b.n_segments, device="cuda") < 0.3).float()
    m = SegmentClassifier(input_dim=2, hidden_dim=32, n_iters=10).cuda().train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    bce = BCELoss()
    for ev in (True, False):
        m.use_events = ev
        def step():
            opt.zero_grad(set_to_none=False); bce(m(b), y).backward(); opt.step()
        for _ in range(3): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        m.eval()
        with torch.no_grad():
            for _ in range(3): m(b)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(50): m(b)
            torch.cuda.synchronize(); df = (time.perf_counter() - t0) / 50
        m.train()
        print("%3d toy graphs, use_events=%s: training step %.3f ms, forward %.3f ms" % (G, ev, dt * 1e3, df * 1e3))

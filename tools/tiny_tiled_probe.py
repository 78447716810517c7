import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_fpga_amd import HitGraphBatch, synth, _lib
from gnn_fpga_amd.model import SegmentClassifier
for G, nh, ns in ((512, 40, 200), (512, 80, 500), (2048, 40, 200)):
    b = HitGraphBatch.from_graphs([synth.layered_graph(nh, ns, 3, seed=s) for s in range(G)]).cuda()
    m = SegmentClassifier(input_dim=3, hidden_dim=8, n_iters=3).cuda().eval()
    m.use_events = False
    with torch.no_grad():
        for _ in range(5): m(b)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): m(b)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        with _lib.profile(64) as prof:
            m(b)
    p = b.plan
    print("%d x (%d, %d): %.3f ms  " % (G, nh, ns, dt * 1e3), [(k, round(v * 1e3, 1)) for k, v in prof.records],
          "tiles", p.n_tiles, "lds tiles", p.n_lds_tiles, "n_pad", p.n_pad, "padding %.2f" % p.padding)

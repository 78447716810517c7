"""c5-shape timing (one 50k-hit / 500k-segment graph, F=3, D=64, T=6, fp32); GNN_ABLATE applies."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_fpga_amd import HitGraphBatch, synth, _lib
from gnn_fpga_amd.model import SegmentClassifier
G = int(sys.argv[1]) if len(sys.argv) > 1 else 1
torch.manual_seed(0)
m = SegmentClassifier(input_dim=3, hidden_dim=64, n_iters=6).cuda().eval()
b = HitGraphBatch.from_graphs([synth.layered_graph(50000, 500000, 3, seed=s) for s in range(G)]).cuda()
with torch.no_grad():
    for _ in range(3): m(b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): m(b)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    with _lib.profile(64) as prof:
        m(b)
per = {}
for k, v in prof.records: per.setdefault(k, []).append(v)
print("ablate=%s G=%d  %.3f ms/forward  %.3g seg/s  kernels: %s" % (os.environ.get("GNN_ABLATE", "0"), G, dt * 1e3, b.n_segments / dt,
      {k: (len(v), round(sum(v) / len(v), 4)) for k, v in per.items()}))

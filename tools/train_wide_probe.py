"""Training step at the mu200 shape (D = 64, T = 6; gnn/MPNN_Seg_ACTS_mu200.ipynb trains this model)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_fpga_amd import HitGraphBatch, synth, _lib
from gnn_fpga_amd.model import SegmentClassifier
from gnn_fpga_amd.loss import BCELoss
D = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 6
G = int(sys.argv[3]) if len(sys.argv) > 3 else 1
b = HitGraphBatch.from_graphs([synth.layered_graph(50000, 500000, 3, seed=s) for s in range(G)]).cuda()
y = (torch.rand(b.n_segments, device="cuda") < 0.3).float()
torch.manual_seed(0)
m = SegmentClassifier(input_dim=3, hidden_dim=D, n_iters=T).cuda().train()
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
bce = BCELoss()
def step():
    opt.zero_grad(set_to_none=False); loss = bce(m(b), y); loss.backward(); opt.step(); return loss
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): l = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
with torch.no_grad():
    m.eval()
    for _ in range(3): m(b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): m(b)
    torch.cuda.synchronize(); df = (time.perf_counter() - t0) / 10
    m.train()
print("D=%d T=%d, %d x (50k hits, 500k segments): training step %.2f ms (%.3g seg/s), inference forward %.2f ms; loss %.4f"
      % (D, T, G, dt * 1e3, b.n_segments / dt, df * 1e3, float(l)))
with _lib.profile(512) as prof:
    step()
per = {}
for k, v in prof.records: per[k] = per.get(k, 0.0) + v
print("  " + ", ".join("%s %.2f" % kv for kv in sorted(per.items(), key=lambda kv: -kv[1])[:10]))

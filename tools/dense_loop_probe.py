"""The UNCHANGED reference training loop's input contract on the drop-in: every step hands over fresh dense
[X, Ri, Ro] CUDA tensors (gnn/estimator.py:49-60 after np_to_torch(...).cuda()), B muon-schema graphs zero-padded
like merge_graphs (gnn/trainSegmentClassifier.py:66-95).  Per step: dense -> index (gnn_dense_to_index), segment
lists (gnn_csr_build; GNN_CSR_BUILDER=torch: stable torch sorts, what rounds 1-2 did), one-launch forward /
backward kernels, BCE, Adam."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gnn_fpga_amd import synth
from gnn_fpga_amd.model import SegmentClassifier

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
graphs = [synth.muon_graph(s) for s in range(B)]
Nmax = max(g.X.shape[0] for g in graphs)
Emax = max(g.src.shape[0] for g in graphs)
dense = [synth.to_dense(g, Nmax, Emax) for g in graphs]
X, Ri, Ro = (torch.from_numpy(np.stack([d[i] for d in dense])).cuda() for i in range(3))
y = torch.from_numpy(np.stack([np.pad(g.y, (0, Emax - g.y.shape[0])) for g in graphs])).cuda()
torch.manual_seed(0)
m = SegmentClassifier(input_dim=11, hidden_dim=8, n_iters=3).cuda().train()
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
loss_func = torch.nn.BCELoss()

def step():
    m.zero_grad()
    opt.zero_grad()
    out = m([X.clone(), Ri.clone(), Ro.clone()])          # fresh tensors, a new batch object every step
    loss = loss_func(out, y)
    loss.backward()
    opt.step()
    return loss

for builder in ("hip", "torch"):
    os.environ["GNN_CSR_BUILDER"] = builder
    for _ in range(10):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100):
        l = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
    print("reference-style step on fresh dense [X, Ri, Ro] (%d muon graphs, N_max %d, E_max %d), segment lists by %-5s: "
          "%.0f us per step, loss %.4f" % (B, Nmax, Emax, builder, dt * 1e6, float(l)))

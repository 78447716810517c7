"""k_event_bwd time against the number of iterations (set-up + flush vs per-iteration stages):
run under `rocprofv3 --kernel-trace --stats` once per T (argv[1])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_fpga_amd import HitGraphBatch, synth
from gnn_fpga_amd.model import SegmentClassifier
from gnn_fpga_amd.loss import BCELoss

T = int(sys.argv[1])
b = HitGraphBatch.from_graphs([synth.muon_graph(s) for s in range(512)]).cuda()
y = (torch.arange(b.n_segments, device="cuda") % 2 == 0).float()
m = SegmentClassifier(input_dim=11, hidden_dim=8, n_iters=T).cuda().train()
for _ in range(30):
    m.zero_grad()
    BCELoss()(m(b), y).backward()
torch.cuda.synchronize()

import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gnn_fpga_amd import HitGraphBatch, synth, _lib
from gnn_fpga_amd.model import SegmentClassifier
for F, D, T in ((3, 64, 6), (3, 32, 3), (2, 32, 10)):
    torch.manual_seed(D + T)
    graphs = [synth.layered_graph(700, 4000, F, seed=60 + i) for i in range(3)]
    m = SegmentClassifier(input_dim=F, hidden_dim=D, n_iters=T).cuda().eval(); m.use_events = False
    b = HitGraphBatch.from_graphs(graphs).cuda()
    with torch.no_grad():
        e32 = m(b); m.mlp_bf16 = True; e16 = m(b)
    print("F%d D%d T%d  max |bf16 - fp32| = %.2e  mean %.2e" % (F, D, T, (e16 - e32).abs().max().item(), (e16 - e32).abs().mean().item()))
# c5 timing
torch.manual_seed(0)
m = SegmentClassifier(input_dim=3, hidden_dim=64, n_iters=6).cuda().eval()
b = HitGraphBatch.from_graphs([synth.layered_graph(50000, 500000, 3, seed=0)]).cuda()
for bf in (False, True):
    m.mlp_bf16 = bf
    with torch.no_grad():
        for _ in range(3): e = m(b)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): m(b)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        with _lib.profile(64) as prof: m(b)
    per = {}
    for k, v in prof.records: per.setdefault(k, []).append(v)
    print("c5 bf16=%s  %.3f ms  %s" % (bf, dt * 1e3, {k: (len(v), round(sum(v) / len(v), 4)) for k, v in per.items()}))
    if bf: print("   max diff vs fp32 %.2e" % (e - e_prev).abs().max().item())
    e_prev = e

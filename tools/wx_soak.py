"""Soak of the role-split wide kernel (k_iter_wx: LDS ring + counters instead of barriers) against the barrier
kernel it replaces (k_iter_w, GNN_WIDE_LOCKSTEP=1): random batch shapes (ragged, tiny, deep, many graphs), hidden_dim
16 / 32 / 64, fp32 and bf16 records, every forward repeated; scores must be bit-identical.  A lost, doubled or
never-finished slot hand-off shows up here as a mismatch or as a forward that takes seconds (bounded polls).
usage: python tools/wx_soak.py [trials]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gnn_fpga_amd import HitGraphBatch, synth
from gnn_fpga_amd.model import SegmentClassifier

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(0)
os.environ["GNN_WIDE_ROLES"] = "1"
worst_t = 0.0
for trial in range(trials):
    F, D = [(2, 16), (3, 16), (2, 32), (3, 32), (3, 64)][int(rng.integers(0, 5))]
    T = int(rng.integers(1, 5))
    bf16 = bool(D >= 32 and rng.random() < 0.4)
    G = int(rng.integers(1, 40))
    graphs = []
    for g in range(G):
        n = int(rng.choice([7, 40, 300, 2500, 12000], p=[0.1, 0.2, 0.3, 0.3, 0.1]))
        L = int(rng.integers(2, 14))
        e = int(n * rng.uniform(0.5, 12))
        graphs.append(synth.layered_graph(max(n, L), max(e, 1), F, n_layers=L, seed=int(rng.integers(1 << 30))))
    torch.manual_seed(trial)
    m = SegmentClassifier(input_dim=F, hidden_dim=D, n_iters=T).cuda().eval()
    m.use_events = False
    m.mlp_bf16 = bf16
    b = HitGraphBatch.from_graphs(graphs).cuda()
    with torch.no_grad():
        os.environ.pop("GNN_WIDE_LOCKSTEP", None)
        m(b); torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = [m(b).clone() for _ in range(4)]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 4
        os.environ["GNN_WIDE_LOCKSTEP"] = "1"
        ref = m(b).clone()
        torch.cuda.synchronize()
        os.environ.pop("GNN_WIDE_LOCKSTEP", None)
    ok = all(torch.equal(o, ref) for o in outs)
    worst_t = max(worst_t, dt)
    print("trial %3d F=%d D=%2d T=%d bf16=%d graphs=%3d hits=%7d segs=%8d  %.3f ms  %s"
          % (trial, F, D, T, bf16, G, b.n_hits, b.n_segments, dt * 1e3, "ok" if ok else "MISMATCH"), flush=True)
    if not ok or dt > 0.5:
        sys.exit(1)
print("all %d trials bit-identical; slowest forward %.3f ms" % (trials, worst_t * 1e3))

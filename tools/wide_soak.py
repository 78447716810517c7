"""One-off soak of the wide training kernels: random batch shapes, new path against the per-pass / one-lane
kernels it replaces (forward tensors 1e-6, gradients 1e-5 of the largest entry)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gnn_fpga_amd import HitGraphBatch, synth, _lib
from gnn_fpga_amd.model import SegmentClassifier
rng = np.random.default_rng(0)
worst_f = worst_g = 0.0
for trial in range(24):
    F, D = [(3, 64), (2, 32), (3, 32)][trial % 3]
    T = int(rng.integers(1, 4))
    graphs = []
    for g in range(int(rng.integers(1, 6))):
        nh = int(rng.integers(10, 3000)); ns = int(rng.integers(nh // 2 + 1, 9 * nh))
        graphs.append(synth.layered_graph(nh, ns, F, n_layers=int(rng.integers(2, 12)), seed=1000 * trial + g))
    b = HitGraphBatch.from_graphs(graphs)
    src, dst = b.src.numpy().copy(), b.dst.numpy().copy()
    if trial % 2:
        pad = rng.random(len(src)) < 0.05
        src[pad] = -1; dst[pad] = -1
    b = HitGraphBatch(b.X.numpy(), src, dst, hit_ptr=b.hit_ptr, seg_ptr=b.seg_ptr).cuda()
    torch.manual_seed(trial)
    m = SegmentClassifier(input_dim=F, hidden_dim=D, n_iters=T).cuda()
    w = [t.detach().contiguous() for t in m.state_dict().values()]
    os.environ.pop("GNN_NODE_ONE_LANE", None); os.environ.pop("GNN_BWD_WIDE_PER_PASS", None)
    fw = _lib.segclf_forward_train(b, w, F, D, T)
    go = torch.randn(b.n_segments, device="cuda") / max(b.n_segments, 1)
    gw = _lib.segclf_backward(b, w, F, D, T, fw[0], fw[1], go, Q_all=fw[2])
    os.environ["GNN_NODE_ONE_LANE"] = "1"; os.environ["GNN_BWD_WIDE_PER_PASS"] = "1"
    fo = _lib.segclf_forward_train(b, w, F, D, T)
    gopp = _lib.segclf_backward(b, w, F, D, T, fo[0], fo[1], go, Q_all=fo[2])
    df = max(float((a - c).abs().max()) for a, c in zip(fw, fo))
    dg = max(float((a - c).abs().max() / (c.abs().max() + 1e-30)) for a, c in zip(gw, gopp))
    worst_f, worst_g = max(worst_f, df), max(worst_g, dg)
    print("trial %2d F=%d D=%d T=%d hits %6d segs %7d graphs %d: forward diff %.2e, gradient rel diff %.2e" % (trial, F, D, T, b.n_hits, b.n_segments, len(graphs), df, dg))
print("worst forward %.2e, worst gradient %.2e" % (worst_f, worst_g))
assert worst_f < 2e-6 and worst_g < 2e-5

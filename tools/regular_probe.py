"""How much do degree irregularity, SELL padding and group rounding cost?  Same c3 sizes, but every
hit has exactly DEG in- and out-segments (DEG random permutations per layer pair): no padding, no
rounding when DEG % 4 == 0, perfectly balanced phases.  Compare segments/s with bench.py's graphs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gnn_fpga_amd import HitGraphBatch, synth, _lib
from gnn_fpga_amd.model import SegmentClassifier

def regular_graph(n_layers, per_layer, deg, seed):
    rng = np.random.default_rng(seed)
    n = n_layers * per_layer
    X = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    src, dst = [], []
    for l in range(n_layers - 1):
        for _ in range(deg):
            perm = rng.permutation(per_layer)
            src.append(l * per_layer + np.arange(per_layer))
            dst.append((l + 1) * per_layer + perm)
    src = np.concatenate(src).astype(np.int32); dst = np.concatenate(dst).astype(np.int32)
    return synth.HitGraph(X, src, dst, np.zeros(len(src), np.float32))

G = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for deg in [int(a) for a in sys.argv[2:]] or (12, 10, 8):
    graphs = [regular_graph(10, 1000, deg, s) for s in range(G)]
    b = HitGraphBatch.from_graphs(graphs).cuda()
    torch.manual_seed(0)
    m = SegmentClassifier(input_dim=3, hidden_dim=8, n_iters=3).cuda().eval()
    with torch.no_grad():
        for _ in range(20): m(b)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): m(b)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
        with _lib.profile(64) as prof:
            for _ in range(5): m(b)
    per = {}
    for k, v in prof.records: per.setdefault(k, []).append(v)
    print("regular degree %2d: %d segments  %.3f ms/step  %.3g segments/s  padding %.1f%%  kernels %s"
          % (deg, b.n_segments, dt * 1e3, b.n_segments / dt, 100 * b.plan.padding,
             {k: round(sum(v) / len(v), 4) for k, v in per.items()}))

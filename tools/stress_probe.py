"""One-off robustness runs beyond the test suite's sizes (MI355X, a few minutes):
 (1) 512 c3 graphs in one batch (5.1M hits / 51M segments): spot-check graphs against the C oracle;
 (2) ONE graph of 1M hits / 10M segments (levels of 100k hits: windows exceed LDS, every tile runs
     the general kernel's global-gather mode) against the C oracle;
 (3) 20k tiny graphs of 1-3 hits."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gnn_fpga_amd import HitGraphBatch, synth
from gnn_fpga_amd.model import SegmentClassifier
from oracle import index_c

torch.manual_seed(0)
m = SegmentClassifier(input_dim=3, hidden_dim=8, n_iters=3).cuda().eval()
params = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}


def check(name, graphs, sample):
    t0 = time.time()
    b = HitGraphBatch.from_graphs(graphs).cuda()
    with torch.no_grad():
        e = m(b)
        torch.cuda.synchronize()
        t1 = time.time()
        for _ in range(3): m(b)
        torch.cuda.synchronize()
        dt = (time.time() - t1) / 3
    es = b.split_scores(e.cpu().numpy())
    worst = 0.0
    for i in sample:
        g = graphs[i]
        ref = index_c.segment_classifier(g.X, g.src, g.dst, params, 3)
        if ref.size: worst = max(worst, float(np.abs(es[i] - ref).max()))
    plan = b.plan
    print("%-34s hits %9d segs %10d  setup %.1f s  forward %.3f ms  max |HIP - oracle| %.2e  lds tiles %s/%s"
          % (name, b.n_hits, b.n_segments, t1 - t0, dt * 1e3, worst,
             getattr(plan, "n_lds_tiles", "-"), getattr(plan, "n_tiles", "-")))
    assert worst < 1e-5


check("512 c3 graphs", [synth.layered_graph(10000, 100000, 3, seed=s) for s in range(512)], [0, 255, 511])
check("one 1M-hit / 10M-segment graph", [synth.layered_graph(1000000, 10000000, 3, seed=7)], [0])
rng = np.random.default_rng(3)
tiny = []
for s in range(20000):
    n = int(rng.integers(1, 4))
    e = int(rng.integers(0, 4))
    X = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    tiny.append(synth.HitGraph(X, rng.integers(0, n, e).astype(np.int32), rng.integers(0, n, e).astype(np.int32),
                               np.zeros(e, np.float32)))
check("20000 graphs of 1-3 hits", tiny, list(range(0, 20000, 997)))
print("stress ok")

"""A never-seen batch end to end - device-resident (X, src, dst) in, scores out, synchronised wall clock - on the
three routes: default first forward (use_plan = "auto": gnn_csr_build + per-module kernels), fused pipeline incl.
its plan, per-module kernels on torch-sorted lists (what rounds 1-2 did); and what the first-forward route's
launches cost one by one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_fpga_amd import HitGraphBatch, synth, _lib
from gnn_fpga_amd.model import SegmentClassifier


def fresh(host, X, src, dst):
    b = HitGraphBatch.__new__(HitGraphBatch)          # fresh batch object over resident arrays
    b.__dict__.update(host.__dict__)
    b.X, b.src, b.dst, b._csr, b.plan, b._event = X, src, dst, None, None, None
    b._src_host = b._dst_host = None
    b._gstruct = None
    return b


def one_shot(name, graphs, F, D, T, reps=10, events=False):
    torch.manual_seed(0)
    m = SegmentClassifier(input_dim=F, hidden_dim=D, n_iters=T).cuda().eval()
    m.use_events = events
    host = HitGraphBatch.from_graphs(graphs)
    X, src, dst = host.X.cuda(), host.src.cuda(), host.dst.cuda()
    out = {}
    for route, plan, builder in (("first", "auto", "hip"), ("plan", True, "hip"), ("torch-csr", False, "torch")):
        m.use_plan = plan
        os.environ["GNN_CSR_BUILDER"] = builder
        ts = []
        for _ in range(reps + 2):
            b = fresh(host, X, src, dst)
            if events:
                b._event = None
            torch.cuda.synchronize(); t0 = time.perf_counter()
            with torch.no_grad():
                m(b)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        out[route] = sorted(ts[2:])[len(ts[2:]) // 2] * 1e3
    os.environ["GNN_CSR_BUILDER"] = "hip"
    m.use_plan = "auto"
    b = fresh(host, X, src, dst)
    with torch.no_grad(), _lib.profile(128) as prof:
        m(b)
    per = {}
    for k, v in prof.records:
        per[k] = per.get(k, 0.0) + v
    print("%-24s segs %9d  first forward %.3f ms (%.3g seg/s) | plan + fused %.3f | torch sorts + per-module %.3f | "
          "first-forward kernels (ms): %s"
          % (name, host.n_segments, out["first"], host.n_segments / out["first"] * 1e3, out["plan"], out["torch-csr"],
             ", ".join("%s %.3f" % kv for kv in sorted(per.items(), key=lambda kv: -kv[1]))))


one_shot("c3 single graph", [synth.layered_graph(10000, 100000, 3, seed=0)], 3, 8, 3)
one_shot("c2 one muon graph", [synth.muon_graph(3)], 11, 8, 3, events=True)
one_shot("c5 single graph", [synth.layered_graph(50000, 500000, 3, seed=0)], 3, 64, 6)
one_shot("c3 x 32", [synth.layered_graph(10000, 100000, 3, seed=s) for s in range(32)], 3, 8, 3, 5)
one_shot("c3 x 256", [synth.layered_graph(10000, 100000, 3, seed=s) for s in range(256)], 3, 8, 3, 3)
# wide shapes: where the per-module kernels stop answering a first forward faster than plan + fused pipeline
one_shot("c5 x 2 (D = 64)", [synth.layered_graph(50000, 500000, 3, seed=s) for s in range(2)], 3, 64, 6, 5)
one_shot("c3 x 8, D = 32", [synth.layered_graph(10000, 100000, 3, seed=s) for s in range(8)], 3, 32, 3, 5)
one_shot("c3 x 16, D = 32", [synth.layered_graph(10000, 100000, 3, seed=s) for s in range(16)], 3, 32, 3, 5)
one_shot("c3 x 16, D = 16", [synth.layered_graph(10000, 100000, 3, seed=s) for s in range(16)], 3, 16, 3, 5)
one_shot("c3 x 32, D = 16", [synth.layered_graph(10000, 100000, 3, seed=s) for s in range(32)], 3, 16, 3, 5)

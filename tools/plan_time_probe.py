"""Plan build times on the GPU: HIP builder (csrc/plan_build.hip) vs the torch-op builder."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_fpga_amd import HitGraphBatch, synth, _lib
from gnn_fpga_amd.plan_hip import HipSellPlan
from gnn_fpga_amd.plan_device import DeviceSellPlan

def probe(name, graphs, F, D, reps=5):
    lim = _lib.plan_limits(F, D)
    b = HitGraphBatch.from_graphs(graphs).cuda()
    out = []
    for cls in (HipSellPlan, DeviceSellPlan):
        cls(b, lim); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            cls(b, lim)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / reps * 1e3)
    with _lib.profile(512) as prof:
        HipSellPlan(b, lim)
    per = {}
    for k, v in prof.records: per[k] = per.get(k, 0.0) + v
    top = sorted(per.items(), key=lambda kv: -kv[1])[:8]
    print("%-28s hits %8d segs %9d   HIP %8.3f ms   torch %8.1f ms   kernels (own, ms): %s"
          % (name, b.n_hits, b.n_segments, out[0], out[1], ", ".join("%s %.3f" % kv for kv in top)))

probe("c3 single graph", [synth.layered_graph(10000, 100000, 3, seed=0)], 3, 8, 20)
probe("c4 512 muon graphs", [synth.muon_graph(s) for s in range(512)], 11, 8, 20)
probe("c3 x 32", [synth.layered_graph(10000, 100000, 3, seed=s) for s in range(32)], 3, 8, 5)
probe("c5 x 8", [synth.layered_graph(50000, 500000, 3, seed=s) for s in range(8)], 3, 64, 5)
probe("c3 x 256", [synth.layered_graph(10000, 100000, 3, seed=s) for s in range(256)], 3, 8, 3)


def one_shot(name, graphs, F, D, T, reps=10):
    """A never-seen batch end to end (what inference on a stream of events pays): fused pipeline
    (HIP-built plan + forward) against the per-module CSR route (two GPU sorts + per-module kernels)."""
    from gnn_fpga_amd.model import SegmentClassifier
    torch.manual_seed(0)
    m = SegmentClassifier(input_dim=F, hidden_dim=D, n_iters=T).cuda().eval()
    m.use_events = False
    host = HitGraphBatch.from_graphs(graphs)
    X, src, dst = host.X.cuda(), host.src.cuda(), host.dst.cuda()
    out = {}
    for plan in (True, False):
        m.use_plan = plan
        ts = []
        for _ in range(reps + 2):
            b = HitGraphBatch.__new__(HitGraphBatch)          # fresh batch object over resident arrays
            b.__dict__.update(host.__dict__)
            b.X, b.src, b.dst, b._csr, b.plan, b._event = X, src, dst, None, None, None
            b._src_host = b._dst_host = None
            b._gstruct = None
            torch.cuda.synchronize(); t0 = time.perf_counter()
            with torch.no_grad():
                m(b)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        out[plan] = sorted(ts[2:])[len(ts[2:]) // 2] * 1e3
    print("%-28s one-shot forward on a fresh batch: fused pipeline incl. plan %.3f ms, per-module CSR route incl. "
          "CSR build %.3f ms" % (name, out[True], out[False]))


one_shot("c3 single graph", [synth.layered_graph(10000, 100000, 3, seed=0)], 3, 8, 3)
one_shot("c3 x 32", [synth.layered_graph(10000, 100000, 3, seed=s) for s in range(32)], 3, 8, 3, 5)
one_shot("c3 x 256", [synth.layered_graph(10000, 100000, 3, seed=s) for s in range(256)], 3, 8, 3, 3)

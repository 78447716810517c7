"""Forward time of the fused pipeline for other supported shapes on c3-size graphs: tools/shape_probe.py [G]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_fpga_amd import HitGraphBatch, synth
from gnn_fpga_amd.model import SegmentClassifier
G = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for F, D, T in ((3, 4, 3), (3, 8, 3), (3, 16, 3), (11, 8, 3), (11, 16, 3), (3, 32, 3), (3, 64, 3)):
    graphs = [synth.layered_graph(10000, 100000, F, seed=s) for s in range(G)]
    b = HitGraphBatch.from_graphs(graphs).cuda()
    torch.manual_seed(0)
    m = SegmentClassifier(input_dim=F, hidden_dim=D, n_iters=T).cuda().eval()
    m.use_events = False
    with torch.no_grad():
        for _ in range(5): m(b)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): m(b)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("F=%2d D=%2d T=%d  c3 x %d: %.3f ms  %.3g segments/s" % (F, D, T, G, dt * 1e3, b.n_segments / dt))
    if os.environ.get("SHAPE_PROBE_KERNELS"):            # per-launch medians (bench.event_medians)
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        with torch.no_grad():
            print("      " + "  ".join("%s %.3f" % kv for kv in bench.event_medians(lambda: m(b), 10)))

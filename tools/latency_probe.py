"""Single-launch latencies (reported separately from the batch roofline number, SURVEY hard part 1):
one c3 graph, and 512 faithful-size muon graphs (c2/c4 shape) as one block-diagonal batch."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_fpga_amd import HitGraphBatch, synth
from gnn_fpga_amd.model import SegmentClassifier

def probe(name, graphs, F, D, T, reps=200, events=True, bf16=False):
    torch.manual_seed(0)
    m = SegmentClassifier(input_dim=F, hidden_dim=D, n_iters=T).cuda().eval()
    m.use_events = events          # False: force the tiled pipeline (one-launch small-event path off)
    m.mlp_bf16 = bf16              # True: hit update on the matrix cores (bf16 operands), D = 32 / 64
    b = HitGraphBatch.from_graphs(graphs).cuda()
    if not events:
        b.build_plan(D)
    with torch.no_grad():
        for _ in range(10): m(b)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): m(b)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        cg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg): m(b)
        for _ in range(10): cg.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): cg.replay()
        torch.cuda.synchronize(); dg = (time.perf_counter() - t0) / reps
    print("%-40s hits %8d segs %9d  eager %8.1f us  graph-replay %8.1f us  (%.3g / %.3g segs/s)"
          % (name, b.n_hits, b.n_segments, dt * 1e6, dg * 1e6, b.n_segments / dt, b.n_segments / dg))

probe("c3 single graph (10k/100k, F3 D8 T3)", [synth.layered_graph(10000, 100000, 3, seed=0)], 3, 8, 3)
probe("c2 one muon graph (F11 D8 T3)", [synth.muon_graph(0)], 11, 8, 3)
probe("c2 one muon graph, tiled pipeline", [synth.muon_graph(0)], 11, 8, 3, events=False)
probe("c4 512 muon graphs (F11 D8 T3)", [synth.muon_graph(s) for s in range(512)], 11, 8, 3)
probe("c4 512 muon graphs, tiled pipeline", [synth.muon_graph(s) for s in range(512)], 11, 8, 3, events=False)
probe("c4/8: 64 muon graphs (one GPU's share)", [synth.muon_graph(s) for s in range(64)], 11, 8, 3)
probe("16384 muon graphs", [synth.muon_graph(s) for s in range(16384)], 11, 8, 3, reps=50)
probe("16384 muon graphs, tiled pipeline", [synth.muon_graph(s) for s in range(16384)], 11, 8, 3, reps=50, events=False)
probe("toy2d 40 hits/144 segs (F2 D32 T10)", [synth.toy2d_graph(seed=0)], 2, 32, 10) if hasattr(synth, "toy2d_graph") else None
probe("c2-scale 2k hits/10k segs (F11 D8 T3)", [synth.layered_graph(2000, 10000, 11, seed=0)], 11, 8, 3)
probe("c1-scale 1k/5k (F2 D32 T10)", [synth.layered_graph(1000, 5000, 2, seed=0)], 2, 32, 10)
probe("c5 fp32 50k/500k (F3 D64 T6)", [synth.layered_graph(50000, 500000, 3, seed=0)], 3, 64, 6, reps=5)
probe("c5 bf16 matrix-core hit update (F3 D64 T6)", [synth.layered_graph(50000, 500000, 3, seed=0)], 3, 64, 6, reps=5, bf16=True)
probe("c5 x 8 graphs, bf16 hit update", [synth.layered_graph(50000, 500000, 3, seed=s) for s in range(8)], 3, 64, 6, reps=3, bf16=True)
probe("c5 x 8 graphs, fp32", [synth.layered_graph(50000, 500000, 3, seed=s) for s in range(8)], 3, 64, 6, reps=3)

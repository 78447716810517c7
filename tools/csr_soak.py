"""Random batches through gnn_csr_build (both size classes: per-segment atomics, LDS-private counting) against the host
builder (hitgraph._csr_by: a stable sort), entry for entry, twice per batch (the arrays must not depend on the order
the atomics arrive in); and through the fused training forward against the per-module one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gnn_fpga_amd import HitGraphBatch, synth, _lib
from gnn_fpga_amd.hitgraph import _csr_by
from gnn_fpga_amd.model import SegmentClassifier

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(2026)
t0 = time.time()
n_big = 0
for t in range(trials):
    kind = t % 4
    if kind == 0:      # many small ragged graphs
        graphs = [synth.layered_graph(int(rng.integers(3, 400)), int(rng.integers(0, 3000)), 3,
                                      n_layers=int(rng.integers(2, 4)), seed=int(rng.integers(1 << 30)))
                  for _ in range(int(rng.integers(1, 200)))]
    elif kind == 1:    # a few detector-size graphs
        graphs = [synth.layered_graph(int(rng.integers(2000, 20000)), int(rng.integers(10000, 200000)), 3,
                                      seed=int(rng.integers(1 << 30))) for _ in range(int(rng.integers(1, 6)))]
    elif kind == 2:    # >= 1 M segments: the LDS-private kernels, narrow ranges
        graphs = [synth.layered_graph(int(rng.integers(5000, 15000)), int(rng.integers(60000, 140000)), 3,
                                      seed=int(rng.integers(1 << 30))) for _ in range(int(rng.integers(11, 16)))]
    else:              # >= 1 M segments in one graph wider than a workgroup's hit range: global claims
        graphs = [synth.layered_graph(int(rng.integers(40000, 300000)), int(rng.integers(1050000, 1500000)), 2,
                                      n_layers=int(rng.integers(3, 20)), seed=int(rng.integers(1 << 30)))]
    b = HitGraphBatch.from_graphs(graphs)
    src, dst = b.src.numpy().copy(), b.dst.numpy().copy()
    if src.size and rng.random() < 0.7:
        k = rng.random(src.size) < rng.random() * 0.2
        src[k] = -1
        dst[k] = -1
    if src.size and rng.random() < 0.3:          # shuffled segment order
        o = rng.permutation(src.size)
        src, dst = src[o], dst[o]
    n = b.n_hits
    want = _csr_by(dst, src, n) + _csr_by(src, dst, n)
    sd, dd = torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda()
    n_big += src.size >= (1 << 20)
    for rep in range(2):
        got = _lib.csr_build(sd, dd, n)
        assert int(got[6].item()) == 0
        nv = int(want[0][-1])
        for name, w, g in zip(HitGraphBatch._CSR_NAMES, want, got[:6]):
            g = g.cpu().numpy()
            if name.endswith("ptr"):
                assert np.array_equal(w, g), (t, name)
            else:
                assert g.shape[0] == src.size and np.array_equal(w, g[:nv]) and np.all(g[nv:] == -1), (t, name)
print("gnn_csr_build: %d random batches (%d of >= 1 M segments), two builds each, all six arrays equal to the host "
      "builder's; %.0f s" % (trials, n_big, time.time() - t0))

# fused training forward vs per-module training forward on random detector-size batches
t0 = time.time()
worst = 0.0
BOUND = 1e-5      # other summation orders: a hit with 100+ segments moves its hidden layer by a few 1e-6; scores agree at 1e-7
for t in range(max(trials // 6, 4)):
    F, D = [(3, 8), (3, 4), (11, 8), (2, 8), (11, 16)][t % 5]
    T = int(rng.integers(0, 4))
    graphs = [synth.layered_graph(int(rng.integers(500, 6000)), int(rng.integers(2000, 50000)), F,
                                  n_layers=int(rng.integers(3, 12)), seed=int(rng.integers(1 << 30)))
              for _ in range(int(rng.integers(2, 10)))]
    b = HitGraphBatch.from_graphs(graphs)
    src, dst = b.src.numpy().copy(), b.dst.numpy().copy()
    k = rng.random(src.size) < 0.05
    src[k] = -1
    dst[k] = -1
    b = HitGraphBatch(b.X.numpy(), src, dst, hit_ptr=b.hit_ptr, seg_ptr=b.seg_ptr).cuda()
    twin = b.level_ordered(D)
    assert twin is not b
    torch.manual_seed(t)
    m = SegmentClassifier(input_dim=F, hidden_dim=D, n_iters=T).cuda()
    w = [x.detach().contiguous() for x in m.effective_weights()]
    fused = _lib.segclf_forward_train_fused(twin, w, F, D, T)
    assert fused is not None
    e_f, H_f, Q_f, out_f = fused
    e_p, H_p, Q_p = _lib.segclf_forward_train(twin, w, F, D, T)
    valid = twin.src >= 0
    ds = ((e_f[:T, valid] - e_p[:T, valid]).abs().max().item() if valid.any() and T else 0.0,
          (H_f - H_p).abs().max().item(), (Q_f - Q_p).abs().max().item() if T else 0.0,
          (out_f - e_p[T].index_select(0, twin.seg_rank)).abs().max().item())
    d = max(ds)
    worst = max(worst, d)
    if os.environ.get("SOAK_VERBOSE") or d >= BOUND:
        per_t = [(H_f[k] - H_p[k]).abs().max().item() for k in range(T + 1)]
        print("batch %d F=%d D=%d T=%d: e_t %.2e  H %.2e  Q %.2e  final %.2e; H per pass %s; max |W| %.2f"
              % (t, F, D, T, ds[0], ds[1], ds[2], ds[3], ["%.1e" % v for v in per_t], max(x.abs().max().item() for x in w)))
    assert d < BOUND, (t, F, D, T, d)
print("fused training forward: %d random batches, every kept tensor within %.1e of the per-module forward's "
      "(bound 1e-5); %.0f s" % (max(trials // 6, 4), worst, time.time() - t0))

"""GPU parity for the data formats either side of the hot path, through the
C ABI: a graph file written by the reference's own `save_graph` -> `from_npz` -> HIP scores, and the
reference batch generator's batches -> `gnn_fpga_amd.batch_generator` -> HIP scores [B, E_max] and
the reference's BCELoss over all B x E_max entries.  Expected values: oracle/gen_golden.py (the
imported reference)."""
import os

import numpy as np
import pytest
import torch

import gnn_fpga_amd
from gnn_fpga_amd import HitGraphBatch, synth
from golden_util import REF_WRITTEN
from test_batcher_host import _fx, _sparse_graphs

pytestmark = pytest.mark.gpu
TOL = 1e-5          # north_star: edge scores within 1e-5 of the CPU reference


def _model(d, F, dev="cuda"):
    from gnn_fpga_amd.model import SegmentClassifier
    params = {k[2:]: torch.from_numpy(v) for k, v in d.items() if k.startswith("p.")}
    D = params["input_network.0.weight"].shape[0]
    m = SegmentClassifier(input_dim=F, hidden_dim=D, n_iters=int(d["n_iters"]))
    m.load_state_dict(params)
    return m.to(dev)


@pytest.mark.parametrize("kind", ["sector", "muon"])
@pytest.mark.parametrize("route", ["events", "plan", "modules"])
def test_npz_written_by_the_reference_to_scores(hip, kind, route):
    d = _fx("refnpz_" + kind)
    b = HitGraphBatch.from_npz(os.path.join(REF_WRITTEN, str(d["filename"]))).cuda()
    m = _model(d, b.n_features).eval()
    m.use_events, m.use_plan = route == "events", route != "modules"
    with torch.no_grad():
        e = m(b)
    assert e.shape == (b.n_segments,)
    assert np.abs(e.cpu().numpy() - d["scores"]).max() < TOL


@pytest.mark.parametrize("name", ["batchgen_sector_b2", "batchgen_muon_b4"])
def test_batch_generator_batches_score_like_the_reference(hip, name):
    from gnn_fpga_amd.loss import BCELoss
    d = _fx(name)
    graphs = _sparse_graphs(d)
    gen = gnn_fpga_amd.batch_generator(graphs, n_samples=int(d["n_samples"]),
                                       batch_size=int(d["batch_size"]), device="cuda")
    m = _model(d, graphs[0].X.shape[1]).eval()
    for b in range(int(d["n_batches"])):
        batch, y = next(gen)
        assert y.is_cuda and batch.X.is_cuda
        for events in (True, False):
            m.use_events = events
            with torch.no_grad():
                out = m(batch)
            assert tuple(out.shape) == d["b%d.scores" % b].shape                # [B, E_max]
            assert np.abs(out.cpu().numpy() - d["b%d.scores" % b]).max() < TOL
            # the reference's loss: BCELoss mean over ALL B x E_max entries, padded ones included
            loss = BCELoss()(out, y)
            assert abs(loss.item() - float(d["b%d.loss" % b])) < 1e-6
    # flat layout: the same real-segment scores without any padding
    gen = gnn_fpga_amd.batch_generator(graphs, n_samples=int(d["n_samples"]),
                                       batch_size=int(d["batch_size"]), device="cuda", layout="flat")
    batch, y = next(gen)
    with torch.no_grad():
        out = m(batch)
    ref = d["b0.scores"]
    counts = np.diff(batch.seg_ptr)
    want = np.concatenate([ref[i, :c] for i, c in enumerate(counts)])
    assert np.abs(out.cpu().numpy() - want).max() < TOL


def test_graph_store_on_the_device_scores_like_host_built_batches(hip):
    """A dataset resident in HBM (batcher.GraphStore on cuda): batches assembled on the device are the arrays
    HitGraphBatch.from_graphs(...).cuda() holds, so every route gives the same bits - detector-size graphs through
    first forward (per-module kernels on gnn_csr_build's lists) and planned forward, muon graphs through k_event."""
    from gnn_fpga_amd.batcher import GraphStore
    from gnn_fpga_amd.model import SegmentClassifier
    for F, graphs, bs in ((3, [synth.layered_graph(3000 + 100 * s, 30000, 3, seed=s) for s in range(6)], 3),
                          (11, [synth.muon_graph(s) for s in range(40)], 16)):
        store = GraphStore(graphs, device="cuda")
        torch.manual_seed(0)
        m = SegmentClassifier(input_dim=F, hidden_dim=8, n_iters=3).cuda().eval()
        for layout in ("flat", "padded"):
            for j in range(0, len(graphs), bs):
                got, y = store.batch(j, bs, layout)
                ref = HitGraphBatch.from_graphs(graphs[j:j + bs], pad_segments=layout == "padded").cuda()
                assert got.X.is_cuda and torch.equal(got.src, ref.src) and torch.equal(got.dst, ref.dst)
                assert torch.equal(got.X, ref.X) and torch.equal(y.reshape(-1), ref.y)
                with torch.no_grad():
                    a1, b1 = m(got), m(ref)            # first forward of each
                    a2, b2 = m(got), m(ref)            # second (planned / event route again)
                assert torch.equal(a1, b1) and torch.equal(a2, b2)


def test_host_built_batches_can_be_put_together_in_pinned_memory(hip):
    """from_graphs(pin_memory=True) (the default from 1 M segments on when a GPU is there): the same arrays, in
    page-locked memory, so that `.cuda()` runs at the link's rate."""
    graphs = [synth.layered_graph(400 + 10 * s, 3000, 3, seed=s) for s in range(5)]
    for pad in (False, True):
        a = HitGraphBatch.from_graphs(graphs, pad_segments=pad, pin_memory=False)
        b = HitGraphBatch.from_graphs(graphs, pad_segments=pad, pin_memory=True)
        assert b.src.is_pinned() and b.dst.is_pinned() and b.X.is_pinned() and b.y.is_pinned() and not a.src.is_pinned()
        for k in ("X", "src", "dst", "y"):
            assert torch.equal(getattr(a, k), getattr(b, k)), k
        bc = b.cuda()
        assert torch.equal(bc.src.cpu(), a.src) and torch.equal(bc.X.cpu(), a.X)


def test_dense_inputs_are_converted_on_the_device(hip):
    """The reference's dense [B,N,E] contract on CUDA tensors: one HIP kernel (gnn_dense_to_index)
    gives the index form the host adapter gives, refuses what it refuses, and - with validation off -
    synchronises nothing."""
    from gnn_fpga_amd import synth
    graphs = [synth.muon_graph(s) for s in (3, 4, 5, 6)]
    Nmax = max(g.X.shape[0] for g in graphs)
    Emax = max(g.src.shape[0] for g in graphs)
    dense = [synth.to_dense(g, Nmax, Emax) for g in graphs]
    X, Ri, Ro = (torch.from_numpy(np.stack([d[i] for d in dense])) for i in range(3))
    host = HitGraphBatch.from_dense(X, Ri, Ro)
    dev = HitGraphBatch.from_dense(X.cuda(), Ri.cuda(), Ro.cuda())
    assert dev.src.is_cuda and dev.dense_shape == host.dense_shape == (4, Nmax, Emax)
    for k in ("X", "src", "dst"):
        assert torch.equal(getattr(host, k), getattr(dev, k).cpu()), k
    assert np.array_equal(host.hit_ptr, dev.hit_ptr) and np.array_equal(host.seg_ptr, dev.seg_ptr)
    for name in HitGraphBatch._CSR_NAMES:                       # segment lists built on the device from it
        a, c = getattr(host, name), getattr(dev, name).cpu()    # (gnn_csr_build keeps [E] entries: the lists, then -1)
        assert torch.equal(a, c[:a.numel()]) and bool((c[a.numel():] == -1).all()), name
    lay = dev.event_layout()
    assert lay.max_hits == Nmax and lay.max_segments == Emax
    bad = Ri.clone()
    bad[0, 1, 0] = 1.0
    bad[0, 2, 0] = 1.0                                            # two end hits in one column
    with pytest.raises(ValueError):
        HitGraphBatch.from_dense(X.cuda(), bad.cuda(), Ro.cuda())
    half = Ro.clone()
    half[1, :, 0] = 0.0                                           # a column set in Ri only
    with pytest.raises(ValueError):
        HitGraphBatch.from_dense(X.cuda(), Ri.cuda(), half.cuda())
    HitGraphBatch.validate_dense = False
    try:
        b = HitGraphBatch.from_dense(X.cuda(), Ri.cuda(), half.cuda())   # asynchronous: column comes out padded
        assert int(b.src[Emax].item()) == -1 and int(b.dst[Emax].item()) == -1
    finally:
        HitGraphBatch.validate_dense = True


def test_event_route_crossover(hip):
    """Which kernels `SegmentClassifier.forward` launches on either side of the event-route thresholds
    (`_lib.events_preferred`, `n_graphs <= 1024` in model.py), and that the route taken is not the slow
    one: on each shape both routes are timed (HIP-graph replays) and the chosen one may be at most
    1.3x the other's time - the cliff measured past the thresholds was 1.7x (tools/cliff_probe.py), so
    a kernel change that moves the crossover fails here instead of silently costing 1.7x."""
    import time
    from gnn_fpga_amd.model import SegmentClassifier
    torch.manual_seed(0)

    def route_and_times(graphs, F, D):
        m = SegmentClassifier(input_dim=F, hidden_dim=D, n_iters=3).cuda().eval()
        b = HitGraphBatch.from_graphs(graphs).cuda()
        with torch.no_grad():
            m(b)
            with hip.profile(64) as prof:
                m(b)
        names = {k for k, _ in prof.records}
        times = {}
        for ev in (True, False):
            m.use_events = ev
            with torch.no_grad():
                for _ in range(3):
                    m(b)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    m(b)
                for _ in range(5):
                    g.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(50):
                    g.replay()
                torch.cuda.synchronize()
                times[ev] = (time.perf_counter() - t0) / 50
        return names, times

    # below the segment threshold: one launch, one workgroup per graph
    names, t = route_and_times([synth.layered_graph(150, 1000, 3, seed=s) for s in range(256)], 3, 8)
    assert names == {"k_event"}
    assert t[True] <= 1.3 * t[False], t
    # above it: the tiled pipeline
    names, t = route_and_times([synth.layered_graph(300, 2000, 3, seed=s) for s in range(256)], 3, 8)
    assert "k_event" not in names and ("k_iter2" in names or "k_iter" in names)
    # (forcing use_events = True changes nothing above the threshold: both timings are the tiled route)
    # more than 1024 small graphs: the tiled pipeline's throughput wins
    names, _ = route_and_times([synth.muon_graph(s) for s in range(1100)], 11, 8)
    assert "k_event" not in names
    names, t = route_and_times([synth.muon_graph(s) for s in range(512)], 11, 8)
    assert names == {"k_event"}
    assert t[True] <= 1.3 * t[False], t
    # wide hidden layers never take it
    names, _ = route_and_times([synth.bipartite_graph([4] * 10, 2, seed=1)], 2, 32)
    assert "k_event" not in names


def test_role_split_threshold(hip, monkeypatch):
    """hidden_dim 16 / 32 / 64 batches of >= 32768 padded hits run the role-split kernel `k_iter_wx`, smaller ones
    the barrier kernel `k_iter_w` (`kRoleSplitMinHits`, csrc/sell_pipeline.hip).  On a batch well below and one well
    above the threshold both kernels are timed (HIP-graph replays): the one chosen may be at most 1.15x the other's
    time, so a kernel change that moves the crossover is noticed."""
    import time
    from gnn_fpga_amd.model import SegmentClassifier
    torch.manual_seed(0)

    def times(graphs, D):
        m = SegmentClassifier(input_dim=3, hidden_dim=D, n_iters=3).cuda().eval()
        m.use_events = False
        b = HitGraphBatch.from_graphs(graphs).cuda()
        out = {}
        for env in ("GNN_WIDE_ROLES", "GNN_WIDE_LOCKSTEP", None):
            monkeypatch.delenv("GNN_WIDE_ROLES", raising=False)
            monkeypatch.delenv("GNN_WIDE_LOCKSTEP", raising=False)
            if env:
                monkeypatch.setenv(env, "1")
            with torch.no_grad():
                for _ in range(3):
                    m(b)
                if env is None:
                    with hip.profile(64) as prof:
                        m(b)
                    out["default"] = {k for k, _ in prof.records}
                    continue
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    m(b)
                for _ in range(5):
                    g.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(30):
                    g.replay()
                torch.cuda.synchronize()
                out[env] = (time.perf_counter() - t0) / 30
        return out

    small = times([synth.layered_graph(4000, 30000, 3, seed=s) for s in range(2)], 32)          # 8 k hits
    assert "k_iter_w" in small["default"] and "k_iter_wx" not in small["default"]
    assert small["GNN_WIDE_LOCKSTEP"] <= 1.15 * small["GNN_WIDE_ROLES"], small
    large = times([synth.layered_graph(10000, 100000, 3, seed=s) for s in range(24)], 64)      # 240 k hits
    assert "k_iter_wx" in large["default"]
    assert large["GNN_WIDE_ROLES"] <= 1.15 * large["GNN_WIDE_LOCKSTEP"], large

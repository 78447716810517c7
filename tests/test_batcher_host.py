"""Host-side checks (no GPU) of the two data formats either side of the hot path:

* graph files written by the REFERENCE'S OWN writers (`graph.save_graph`, gnn/graph.py:179-181;
  `Muon_graph.save_graph`, gnn/Muon_graph.py:198-205, with pt / eta) - tests/golden/ref_written/,
  produced by oracle/gen_golden.py - through `HitGraphBatch.from_npz`;
* the batches of the reference's `batch_generator` (gnn/trainSegmentClassifier.py:97-111), captured
  by oracle/gen_golden.py through the imported `graph.graph_from_sparse` - `batchgen_*.npz` -
  against `gnn_fpga_amd.batch_generator`: composition order, [B, E_max] targets, padding.
"""
import collections
import os

import numpy as np
import pytest
import torch

import gnn_fpga_amd
from gnn_fpga_amd import HitGraphBatch, synth
from golden_util import GOLDEN, REF_WRITTEN
from oracle import index_c

SparseGraph = collections.namedtuple("SparseGraph", ["X", "Ri_rows", "Ri_cols", "Ro_rows", "Ro_cols", "y"])


def _fx(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as f:
        return {k: f[k] for k in f.files}


def _sparse_graphs(d):
    return [SparseGraph(*[d["s%d.%s" % (i, k)] for k in SparseGraph._fields])
            for i in range(int(d["n_graphs"]))]


@pytest.mark.parametrize("kind", ["sector", "muon"])
def test_file_written_by_the_reference_loads_as_csr(kind):
    d = _fx("refnpz_" + kind)
    b = HitGraphBatch.from_npz(os.path.join(REF_WRITTEN, str(d["filename"])))
    Ri, Ro = d["Ri"], d["Ro"]                     # the reference's graph_from_sparse of the same file
    n, e = Ri.shape
    assert (b.n_hits, b.n_segments) == (n, e)
    # dense <-> index round trip (gnn/GraphConstructionDev_mu200.ipynb cells 41-44)
    Ri_reco = np.zeros_like(Ri)
    Ro_reco = np.zeros_like(Ro)
    Ri_reco[b.dst.numpy(), np.arange(e)] = 1
    Ro_reco[b.src.numpy(), np.arange(e)] = 1
    assert (Ri_reco == Ri).all() and (Ro_reco == Ro).all()
    # the file's order IS the CSR order (nonzero() is row-major): rowptr by bincount, no sort
    in_ptr, in_eid = b.in_ptr.numpy(), b.in_eid.numpy()
    for hit in range(n):
        assert sorted(in_eid[in_ptr[hit]:in_ptr[hit + 1]]) == list(np.flatnonzero(Ri[hit]))
    assert np.array_equal(b.in_nbr.numpy(), b.src.numpy()[in_eid])
    assert np.array_equal(b.out_nbr.numpy(), b.dst.numpy()[b.out_eid.numpy()])
    if kind == "muon":
        assert b.n_features == 11 and abs(b.pt - 23.5) < 1e-6 and abs(b.eta + 1.7) < 1e-6
    else:
        assert b.pt is None and b.eta is None
    # and the oracle on what was loaded reproduces the reference model's scores on that file
    params = {k[2:]: v for k, v in d.items() if k.startswith("p.")}
    ref = index_c.segment_classifier(b.X.numpy(), b.src.numpy(), b.dst.numpy(), params, int(d["n_iters"]))
    assert np.abs(ref - d["scores"]).max() < 1e-5


@pytest.mark.parametrize("name", ["batchgen_sector_b2", "batchgen_muon_b4"])
@pytest.mark.parametrize("layout", ["padded", "flat"])
def test_batch_generator_yields_the_reference_batches(name, layout):
    d = _fx(name)
    graphs = _sparse_graphs(d)
    n_samples, bs, nb = int(d["n_samples"]), int(d["batch_size"]), int(d["n_batches"])
    gen = gnn_fpga_amd.batch_generator(graphs, n_samples=n_samples, batch_size=bs, layout=layout)
    params = {k[2:]: v for k, v in d.items() if k.startswith("p.")}
    seen = []
    for epoch in range(2):                        # the generator loops over epochs for ever
        for b in range(nb):
            batch, y = next(gen)
            ref_y, ref_scores = d["b%d.y" % b], d["b%d.scores" % b]
            members = graphs[b * bs:(b + 1) * bs]              # graphs[j:j+batch_size], :103-104
            assert batch.n_graphs == len(members)
            assert np.array_equal(batch.X.numpy(), np.concatenate([g.X for g in members]))
            if layout == "padded":
                assert tuple(y.shape) == ref_y.shape and y.dtype == torch.float32
                assert np.array_equal(y.numpy(), ref_y)
                assert batch.dense_shape[0] == ref_y.shape[0] and batch.dense_shape[2] == ref_y.shape[1]
                want = ref_scores.reshape(-1)
            else:
                counts = [g.y.shape[0] for g in members]
                assert np.array_equal(batch.to_padded(y).numpy(), ref_y)
                assert torch.equal(batch.from_padded(batch.to_padded(y), counts), y)
                want = np.concatenate([ref_scores[i, :c] for i, c in enumerate(counts)])
            # the oracle on this batch's index form gives the reference's scores, padded columns too
            # (the C oracle scores a padded column like the reference: sigmoid(W2 tanh(b1) + b2))
            e = index_c.segment_classifier(batch.X.numpy(), batch.src.numpy(), batch.dst.numpy(),
                                           params, int(d["n_iters"]))
            assert np.abs(e - want).max() < 1e-5
            seen.append(id(batch))
    assert seen[:nb] == seen[nb:]                 # epochs reuse the batches (plans / CSRs with them)


@pytest.mark.parametrize("name", ["batchgen_sector_b2", "batchgen_muon_b4"])
@pytest.mark.parametrize("layout", ["padded", "flat"])
def test_graph_store_assembles_the_reference_batches(name, layout):
    """batcher.GraphStore: the dataset concatenated once (here on the CPU), every batch assembled from slices + one
    offset add where it lives - the same batches, entry for entry, as merge_graphs on the list (and therefore the
    reference's: the fixtures' targets), also through batch_generator."""
    from gnn_fpga_amd.batcher import GraphStore, merge_graphs
    d = _fx(name)
    graphs = _sparse_graphs(d)
    n_samples, bs, nb = int(d["n_samples"]), int(d["batch_size"]), int(d["n_batches"])
    store = GraphStore(graphs)
    gen = gnn_fpga_amd.batch_generator(store, n_samples=n_samples, batch_size=bs, layout=layout)
    for b in range(nb):
        got, y = next(gen)
        ref, yr = merge_graphs(graphs[b * bs:(b + 1) * bs], layout)
        for k in ("X", "src", "dst", "y"):
            assert torch.equal(getattr(got, k), getattr(ref, k)), k
        assert torch.equal(y, yr) and got.dense_shape == ref.dense_shape
        assert np.array_equal(got.hit_ptr, ref.hit_ptr) and np.array_equal(got.seg_ptr, ref.seg_ptr)
        assert (got.n_hits, got.n_segments, got.n_graphs) == (ref.n_hits, ref.n_segments, ref.n_graphs)
        if layout == "padded":
            assert np.array_equal(y.numpy(), d["b%d.y" % b])
        assert all(torch.equal(a, c) for a, c in zip(got._ensure_csr(), ref._ensure_csr()))
    # ragged graphs, any window, graphs without segments; malformed graphs are refused when the store is built
    gs = [synth.layered_graph(n, e, 3, n_layers=2, seed=s) for s, (n, e) in enumerate([(5, 9), (40, 0), (2, 1), (33, 80), (7, 7)])]
    store = GraphStore(gs)
    for j, k in ((0, 5), (1, 3), (4, 1), (3, 9)):
        got, y = store.batch(j, k, layout)
        ref, yr = merge_graphs(gs[j:j + k], layout)
        assert torch.equal(got.src, ref.src) and torch.equal(got.dst, ref.dst) and torch.equal(y, yr)
    bad = synth.HitGraph(gs[0].X, gs[0].src.copy(), gs[0].dst.copy(), gs[0].y)
    bad.dst[2] = 5
    with pytest.raises(ValueError):
        GraphStore([gs[3], bad])


def test_graph_store_reads_the_files_the_reference_wrote():
    """GraphStore.from_npz on the two graph files written by the reference's own save_graph (tracking and muon
    schema): each graph comes back as the batch HitGraphBatch.from_npz makes of its file."""
    from gnn_fpga_amd.batcher import GraphStore
    files = [os.path.join(REF_WRITTEN, "graph000000.npz"), os.path.join(REF_WRITTEN, "graph_muon_000000.npz")]
    for fn in files:                                   # (the two files have different feature counts: one store each)
        store = GraphStore.from_npz([fn])
        ref = HitGraphBatch.from_npz(fn)
        got, y = store.batch(0, 1, "flat")
        assert torch.equal(got.X, ref.X) and torch.equal(got.src, ref.src) and torch.equal(got.dst, ref.dst)
        assert torch.equal(y, ref.y) and store.pt[0] == ref.pt and store.eta[0] == ref.eta
        assert all(torch.equal(a, c) for a, c in zip(got._ensure_csr(), ref._ensure_csr()))


def test_batch_generator_cache_is_bounded_by_bytes():
    """The generator keeps batches (and what hangs off them: CSRs, plans, level-ordered twins) only
    while their estimated bytes fit `max_cached_bytes`; beyond that, and with cache=False, it holds
    nothing between uses - like the reference's generator (gnn/trainSegmentClassifier.py:97-111)."""
    from gnn_fpga_amd.batcher import _batch_bytes, merge_graphs
    graphs = [synth.layered_graph(200, 900, 3, seed=s) for s in range(6)]
    one = _batch_bytes(*merge_graphs(graphs[:2]))
    assert one > 2 * (2 * 900 * 8 + 2 * 200 * 12)               # twin and plans are counted
    for kw, reused in (({}, [True, True, True]), ({"cache": False}, [False] * 3),
                       ({"max_cached_bytes": int(2.5 * one)}, [True, True, False])):
        gen = gnn_fpga_amd.batch_generator(graphs, n_samples=6, batch_size=2, **kw)
        first = [next(gen)[0] for _ in range(3)]
        second = [next(gen)[0] for _ in range(3)]
        assert [a is b for a, b in zip(first, second)] == reused, kw
        assert all(np.array_equal(a.src.numpy(), b.src.numpy()) for a, b in zip(first, second))


def test_padded_batch_from_hit_graphs_matches_the_dense_adapter():
    graphs = [synth.muon_graph(s) for s in (3, 4, 5, 6)]
    b = HitGraphBatch.from_graphs(graphs, pad_segments=True)
    Nmax = max(g.X.shape[0] for g in graphs)
    Emax = max(g.src.shape[0] for g in graphs)
    assert b.dense_shape == (4, Nmax, Emax) and b.n_segments == 4 * Emax
    for i, g in enumerate(graphs):
        s = b.src.numpy()[i * Emax:(i + 1) * Emax]
        assert np.array_equal(s[:g.src.shape[0]] - b.hit_ptr[i], g.src) and np.all(s[g.src.shape[0]:] == -1)
    lay = b.event_layout()
    assert lay is not None and lay.max_segments == Emax

"""Synthetic hit graphs with the shapes of the reference's data products.

No dataset, ROOT file or TrackML CSV is reachable from this build (no network,
`uproot`/`trackml` absent), so inputs are generated with the *schema and
statistics* the reference's data-prep scripts emit:

* `layered_graph`   - TrackML/ACTS barrel style: hits assigned to
  `n_layers` detector layers, every segment joins a hit in layer l to a hit in
  layer l+1 (adjacent-layer rule, reference gnn/prepareGraphs.py:153-155), segments
  emitted grouped by layer pair (as the pd.concat over layer pairs does,
  reference gnn/graph.py:80-93).
* `bipartite_graph` - toy-2D / muon style: complete bipartite segments between
  consecutive occupied layers (reference gnn/Muon_graph.py:60-83 keeps every pair;
  gnn/MPNN_Seg_Toy2D.ipynb: 4 tracks x 10 layers = 40 hits, 144 segments).

Index convention (reference gnn/graph.py:128-135): segment j starts at hit
`src[j]` (the row with Ro[n, j] = 1) and ends at hit `dst[j]` (Ri[n, j] = 1).

Everything is numpy on the host; `numpy.random.default_rng(seed)` only.
"""
from collections import namedtuple

import numpy as np

# X float32 [N, F]; src, dst int32 [E]; y float32 [E] (segment truth label)
HitGraph = namedtuple("HitGraph", ["X", "src", "dst", "y"])

MUON_FEATURES = ("vh_sim_z", "vh_sim_theta", "vh_sim_phi", "vh_sim_r", "vh_bend",
                 "vh_sim_tp1", "vh_sim_tp2", "vh_station", "vh_ring", "vh_type",
                 "vh_layer")  # reference gnn/prepareMuonGraphs.py:169-170 (F = 11)


def layered_graph(n_hits, n_segments, n_features=3, n_layers=10, seed=0,
                  sort_hits_by_layer=False):
    """TrackML-shaped random layered graph: SURVEY.md 8(d) `synth_graph(N,E,F,L,seed)`."""
    if n_hits < n_layers:
        raise ValueError("need at least one hit per layer")
    rng = np.random.default_rng(seed)
    layer = rng.integers(0, n_layers, size=n_hits)
    layer[rng.permutation(n_hits)[:n_layers]] = np.arange(n_layers)  # no empty layer
    if sort_hits_by_layer:
        layer = np.sort(layer)
    X = rng.uniform(-1.0, 1.0, size=(n_hits, n_features)).astype(np.float32)
    hits_of = [np.flatnonzero(layer == l) for l in range(n_layers)]
    n_pairs = n_layers - 1
    per_pair = np.full(n_pairs, n_segments // n_pairs, dtype=np.int64)
    per_pair[: n_segments % n_pairs] += 1
    src = np.empty(n_segments, dtype=np.int32)
    dst = np.empty(n_segments, dtype=np.int32)
    o = 0
    for l in range(n_pairs):
        k = int(per_pair[l])
        src[o:o + k] = rng.choice(hits_of[l], size=k)
        dst[o:o + k] = rng.choice(hits_of[l + 1], size=k)
        o += k
    y = (rng.random(n_segments) < 0.2).astype(np.float32)
    return HitGraph(X, src, dst, y)


def bipartite_graph(hits_per_layer, n_features=2, seed=0):
    """Complete bipartite segments between consecutive layers.

    `hits_per_layer` is a sequence of hit counts, one per occupied layer; hits are
    numbered layer by layer. Toy-2D: [4]*10 -> 40 hits, 144 segments.
    """
    rng = np.random.default_rng(seed)
    hits_per_layer = [int(h) for h in hits_per_layer]
    n_hits = sum(hits_per_layer)
    X = rng.uniform(-1.0, 1.0, size=(n_hits, n_features)).astype(np.float32)
    first = np.concatenate([[0], np.cumsum(hits_per_layer)])
    src, dst = [], []
    for l in range(len(hits_per_layer) - 1):
        a = np.arange(first[l], first[l + 1])
        b = np.arange(first[l + 1], first[l + 2])
        aa, bb = np.meshgrid(a, b, indexing="ij")
        src.append(aa.ravel())
        dst.append(bb.ravel())
    src = np.concatenate(src).astype(np.int32)
    dst = np.concatenate(dst).astype(np.int32)
    y = (rng.random(src.shape[0]) < 0.25).astype(np.float32)
    return HitGraph(X, src, dst, y)


def toy2d_graph(seed=0):
    """gnn/MPNN_Seg_Toy2D.ipynb shape: 40 hits, 144 segments, F = 2."""
    return bipartite_graph([4] * 10, n_features=2, seed=seed)


def muon_graph(seed=0):
    """prepareMuonGraphs.py output shape: tens of hits over a few signed layers, F = 11."""
    rng = np.random.default_rng(1000 + seed)
    n_layers = int(rng.integers(4, 9))
    hits = rng.integers(1, 6, size=n_layers)
    g = bipartite_graph(hits, n_features=len(MUON_FEATURES), seed=seed)
    return g


def to_dense(graph, n_hits_pad=None, n_segments_pad=None, dtype=np.float32):
    """Index form -> the dense one-hot incidence matrices the reference consumes.

    Same fill rule as reference gnn/graph.py:28-35 (`graph_from_sparse`):
    Ri[dst[j], j] = 1, Ro[src[j], j] = 1; padded rows/columns stay zero
    (reference gnn/trainSegmentClassifier.py:83-93).
    """
    n, e = graph.X.shape[0], graph.src.shape[0]
    N = n if n_hits_pad is None else n_hits_pad
    E = e if n_segments_pad is None else n_segments_pad
    X = np.zeros((N, graph.X.shape[1]), dtype=np.float32)
    X[:n] = graph.X
    Ri = np.zeros((N, E), dtype=dtype)
    Ro = np.zeros((N, E), dtype=dtype)
    j = np.arange(e)
    Ri[graph.dst, j] = 1
    Ro[graph.src, j] = 1
    return X, Ri, Ro

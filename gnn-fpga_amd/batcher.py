"""Index-form counterpart of the reference's batch generator.

Reference (gnn/trainSegmentClassifier.py:66-111): `batch_generator(graphs, n_samples, batch_size,
train)` loops for ever over `graphs[j:j+batch_size]`, j = 0, batch_size, ... < n_samples; every
batch is densified (`graph_from_sparse`, gnn/graph.py:28-35: two [N, E] uint8 matrices per graph),
zero-padded to the batch's largest graph (`merge_graphs`, :66-95), cast to float32 (:38-44) and
handed over as `([X, Ri, Ro], y)` with y of shape [B, E_max].

Here the same batches come out in the same order as `(HitGraphBatch, y)`:

* `layout="padded"` (default, the reference's contract): every graph's segment list is padded to
  E_max with `src = dst = -1` columns, so the model returns scores [B, E_max] and `y` is
  [B, E_max] float32 exactly like the reference's target - padded entries included, which is what
  its BCELoss averages over.  O(E) integers instead of O(N E) matrix entries.
* `layout="flat"`: no padding at all; scores and `y` are [E_total]; `HitGraphBatch.to_padded` /
  `from_padded` map between the two.

Graphs may be the reference's `SparseGraph` tuples (fields X, Ri_rows, Ri_cols, Ro_rows, Ro_cols,
y - what `load_graphs(filenames, SparseGraph)` returns, gnn/graph.py:188-194) or `HitGraph`s.
"""
import numpy as np
import torch

from .hitgraph import HitGraphBatch
from .synth import HitGraph


def as_hit_graph(g):
    """SparseGraph-like (Ri_rows/Ri_cols/Ro_rows/Ro_cols) or HitGraph-like (src/dst) -> HitGraph."""
    if hasattr(g, "src"):
        return g
    e = int(np.asarray(g.Ri_rows).shape[0])
    if int(np.asarray(g.Ro_rows).shape[0]) != e:
        raise ValueError("Ri and Ro must describe the same segments")
    dst = np.full(e, -1, dtype=np.int64)
    src = np.full(e, -1, dtype=np.int64)
    dst[np.asarray(g.Ri_cols, dtype=np.int64)] = np.asarray(g.Ri_rows, dtype=np.int64)
    src[np.asarray(g.Ro_cols, dtype=np.int64)] = np.asarray(g.Ro_rows, dtype=np.int64)
    if e and (src.min() < 0 or dst.min() < 0):
        raise ValueError("every segment needs exactly one start and one end hit")
    y = getattr(g, "y", None)
    return HitGraph(np.asarray(g.X, dtype=np.float32), src.astype(np.int32), dst.astype(np.int32),
                    None if y is None else np.asarray(y, dtype=np.float32))


def merge_graphs(graphs, layout="padded"):
    """One batch from a list of graphs, composition order = list order (reference
    gnn/trainSegmentClassifier.py:66-95).  Returns (HitGraphBatch, y) with y float32 of shape
    [B, E_max] (padded) or [E_total] (flat), or None when a graph has no labels."""
    gs = [as_hit_graph(g) for g in graphs]
    if layout == "flat":
        b = HitGraphBatch.from_graphs(gs)
        return b, (None if b.y is None else b.y.clone())
    if layout != "padded":
        raise ValueError("layout must be 'padded' or 'flat'")
    b = HitGraphBatch.from_graphs(gs, pad_segments=True)
    return b, (None if b.y is None else b.y.view(b.dense_shape[0], b.dense_shape[2]).clone())


def _batch_bytes(b, y):
    """Bytes a cached batch pins at most: its own arrays, its two CSRs and the level-ordered twin the
    training path may add (a second copy of all of them), plus the two execution plans (about the size
    of the segment lists again)."""
    own = b.X.numel() * 4 + 2 * b.n_segments * 4 + (0 if y is None else y.numel() * 4)
    csr = 2 * (b.n_hits + 1) * 4 + 4 * b.n_segments * 4
    plan = 6 * b.n_segments * 4 + 64 * b.n_hits
    return 2 * (own + csr + plan)


def batch_generator(graphs, n_samples=1, batch_size=1, train=True, device=None, layout="padded",
                    cache=True, max_cached_bytes=8 << 30):
    """Endless generator of `(HitGraphBatch, y)` in the reference's order
    (gnn/trainSegmentClassifier.py:97-111).  `train` is accepted for signature compatibility (the
    reference uses it for the long-gone `volatile` flag only).  `device` moves the batches (and y)
    there.  With `cache=True` batches are built once and reused over epochs - their CSRs, execution
    plans and level-ordered twins with them - while the estimated bytes they pin (`_batch_bytes`: twin
    and plans included) stay under `max_cached_bytes` (default 8 GiB of the 288); what does not fit is
    rebuilt each epoch and holds nothing between uses, like every batch of the reference's generator
    (`cache=False`)."""
    del train
    idxs = np.arange(0, n_samples, batch_size)
    kept, kept_bytes = {}, 0
    while True:
        for j in idxs:
            item = kept.get(int(j))
            if item is None:
                b, y = merge_graphs(graphs[j:j + batch_size], layout)
                if device is not None:
                    b = b.to(device)
                    y = None if y is None else y.to(device)
                item = (b, y)
                if cache:
                    need = _batch_bytes(b, y)
                    if kept_bytes + need <= max_cached_bytes:
                        kept[int(j)] = item
                        kept_bytes += need
            yield item

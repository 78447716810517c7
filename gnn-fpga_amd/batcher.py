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

`GraphStore(graphs, device)` keeps a whole dataset resident on the device and assembles the same batches there;
`batch_generator` takes one in place of the list.

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


class GraphStore:
    """A DATASET of hit graphs resident where the model runs: every graph's features, endpoints (ids local to the
    graph) and labels concatenated once into four arrays on `device`, and any run of graphs handed out as a
    `HitGraphBatch` assembled THERE - slices, one offset add, nothing else; no host pass over the segments, nothing
    across PCIe per batch.  What the reference does per batch on the host (`graph_from_sparse` densifies, `merge_graphs`
    zero-pads, `np_to_torch(...).cuda()` uploads: gnn/graph.py:28-35, gnn/trainSegmentClassifier.py:66-111) and what
    `HitGraphBatch.from_graphs` still does per batch in numpy (23-50 ms per 64 detector graphs + their upload, against
    the 3.7 ms the GPU needs to plan and score 256 of them) happens once per dataset: a tracking dataset of a few
    thousand events is tens of GB in index form - it fits the 288 GB of one MI355X many times over.

    `batch(j, batch_size, layout)` = the reference's `graphs[j:j + batch_size]` in list order (`layout` as in
    `merge_graphs`); endpoints are checked once, graph by graph, when the store is built (works on CPU tensors too:
    that is how the CPU tests hold it equal to `HitGraphBatch.from_graphs`)."""

    def __init__(self, graphs, device=None):
        gs = [as_hit_graph(g) for g in graphs]
        self.n_graphs = len(gs)
        hp = np.zeros(self.n_graphs + 1, dtype=np.int64)
        sp = np.zeros(self.n_graphs + 1, dtype=np.int64)
        for i, g in enumerate(gs):
            hp[i + 1] = hp[i] + g.X.shape[0]
            sp[i + 1] = sp[i] + np.asarray(g.src).shape[0]
        self.hit_ptr, self.seg_ptr = hp, sp
        src = np.empty(int(sp[-1]), dtype=np.int32)
        dst = np.empty(int(sp[-1]), dtype=np.int32)
        for i, g in enumerate(gs):                       # local ids, checked while the graph is in cache
            a, b = np.asarray(g.src), np.asarray(g.dst)
            if a.ndim != 1 or a.shape != b.shape:
                raise ValueError("expected src [E], dst [E]")
            if a.shape[0] == 0:
                continue
            if int(a.max()) >= g.X.shape[0] or int(b.max()) >= g.X.shape[0]:
                raise ValueError("segment endpoint out of range")
            if (int(a.min()) < 0 or int(b.min()) < 0) and np.any((a < 0) != (b < 0)):
                raise ValueError("a padded segment must have src = dst = -1")
            src[sp[i]:sp[i + 1]] = a
            dst[sp[i]:sp[i + 1]] = b
        ys = [getattr(g, "y", None) for g in gs]
        t = torch.from_numpy
        dev = torch.device("cpu") if device is None else torch.device(device)
        self.device = dev
        self.X = t(np.concatenate([np.asarray(g.X, dtype=np.float32) for g in gs])).to(dev)
        self.src, self.dst = t(src).to(dev), t(dst).to(dev)
        self.y = None if any(v is None for v in ys) else \
            t(np.concatenate([np.asarray(v, dtype=np.float32) for v in ys])).to(dev)

    @classmethod
    def from_npz(cls, filenames, device=None):
        """The store of the graph files the reference's `save_graph` / `save_graphs` wrote (gnn/graph.py:179-194; the
        muon writer's `pt` / `eta` are kept in `self.pt` / `self.eta`): what `load_graphs(filenames, SparseGraph)`
        (gnn/graph.py:188-194) reads per training run, read once and kept on the device."""
        from collections import namedtuple
        SG = namedtuple("SparseGraph", ["X", "Ri_rows", "Ri_cols", "Ro_rows", "Ro_cols", "y"])
        gs, pt, eta = [], [], []
        for fn in filenames:
            with np.load(fn) as f:                        # allow_pickle stays False
                gs.append(SG(f["X"], f["Ri_rows"], f["Ri_cols"], f["Ro_rows"], f["Ro_cols"],
                             f["y"] if "y" in f.files else None))
                pt.append(float(f["pt"]) if "pt" in f.files else None)
                eta.append(float(f["eta"]) if "eta" in f.files else None)
        store = cls(gs, device)
        store.pt, store.eta = pt, eta
        return store

    def batch(self, j, batch_size=1, layout="padded"):
        """(HitGraphBatch, y) of graphs j ... j + batch_size - 1, like `merge_graphs(graphs[j:j + batch_size], layout)`."""
        if layout not in ("padded", "flat"):
            raise ValueError("layout must be 'padded' or 'flat'")
        j1 = min(j + batch_size, self.n_graphs)
        B = j1 - j
        hp = self.hit_ptr[j:j1 + 1] - self.hit_ptr[j]
        counts = np.diff(self.seg_ptr[j:j1 + 1])
        e0, e1 = int(self.seg_ptr[j]), int(self.seg_ptr[j1])
        dev = self.device
        src, dst = self.src[e0:e1], self.dst[e0:e1]
        y = None if self.y is None else self.y[e0:e1]
        cnt_t = torch.from_numpy(counts).to(dev)
        # the offset of a segment's graph inside the batch, one entry per segment (a few hundred bytes cross over;
        # output_size: no read-back of the total)
        off = torch.repeat_interleave(torch.from_numpy(hp[:-1].astype(np.int32)).to(dev), cnt_t, output_size=e1 - e0)
        src = torch.where(src >= 0, src + off, src)
        dst = torch.where(dst >= 0, dst + off, dst)
        dense_shape = None
        sp = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        if layout == "padded":
            e_max = int(counts.max(initial=0))
            n_max = int(np.diff(hp).max(initial=0))
            if B * e_max != e1 - e0:                     # ragged: every graph's segments into its row of [B, E_max]
                row = torch.repeat_interleave(torch.arange(B, device=dev), cnt_t, output_size=e1 - e0)
                col = torch.arange(e1 - e0, device=dev) - torch.repeat_interleave(
                    torch.from_numpy(sp[:-1]).to(dev), cnt_t, output_size=e1 - e0)
                at = row * e_max + col
                full = lambda v, fill, dt: torch.full((B * e_max,), fill, dtype=dt, device=dev).index_copy_(0, at, v)  # noqa: E731
                src, dst = full(src, -1, torch.int32), full(dst, -1, torch.int32)
                y = None if y is None else full(y, 0.0, torch.float32)
            sp = np.arange(B + 1, dtype=np.int64) * e_max
            dense_shape = (B, n_max, e_max)
        b = HitGraphBatch._from_device_arrays(self.X[int(self.hit_ptr[j]):int(self.hit_ptr[j1])], src.contiguous(),
                                              dst.contiguous(), y, hp, sp, dense_shape)
        if y is not None and dense_shape is not None:
            y = b.y.view(dense_shape[0], dense_shape[2])
        elif y is not None:
            y = b.y
        return b, (None if y is None else y.clone())


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
                if isinstance(graphs, GraphStore):        # assembled where the dataset lives
                    b, y = graphs.batch(int(j), min(batch_size, n_samples - int(j)), layout)
                else:
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

"""Index-form hit graphs: the HBM layout the HIP kernels consume.

The reference keeps the hit<->segment association as two dense one-hot matrices
`Ri`, `Ro` of shape [B, N, E] (reference gnn/graph.py:28-35,
gnn/trainSegmentClassifier.py:66-95) and uses `bmm` against them as gather and
scatter-add (reference gnn/model.py:71-72,114-119).  Here the same association is
stored once, in index form, for a *block-diagonal batch* of G graphs:

    X        float32 [N, F]     hit features, graphs concatenated (global hit ids)
    src,dst  int32   [E]        start / end hit of every segment; -1 = padded column
    in_ptr   int32   [N+1]      CSR over segments ENDING at each hit   (rows of Ri)
    in_eid   int32   [E_valid]  segment ids, grouped by end hit, ascending id (built on the GPU: [E], the
                                entries from in_ptr[N] on are -1)
    in_nbr   int32   [E_valid]  = src[in_eid]   (the hit at the other end)
    out_ptr / out_eid / out_nbr the same for segments STARTING at each hit (rows of Ro),
                                out_nbr = dst[out_eid]

On disk the reference already stores exactly this ordering: `Ri.nonzero()` is
row-major (reference gnn/graph.py:23-26), so `Ri_rows` is sorted by hit and
`Ri_cols` is `in_eid`; `from_sparse_arrays` therefore needs a bincount, not a sort.

Host-side numpy builds the plan once per batch; `to(device)` uploads it.
"""
import os

import numpy as np
import torch

_I32 = np.int32


def _csr_by(key, other, n_hits):
    """Group valid segments by `key` hit; stable => ascending segment id per hit."""
    valid = np.flatnonzero(key >= 0)
    k = key[valid]
    from .plan import stable_argsort
    order = stable_argsort(k)
    eid = valid[order].astype(_I32)
    ptr = np.zeros(n_hits + 1, dtype=np.int64)
    np.cumsum(np.bincount(k, minlength=n_hits), out=ptr[1:])
    if ptr[-1] >= 2 ** 31:
        raise ValueError("segment count exceeds int32 index range")
    return ptr.astype(_I32), eid, other[eid].astype(_I32)


def _csr_by_device(key, other, n_hits):
    """_csr_by with torch ops on the tensors' own device (stable sort: the same arrays)."""
    valid = torch.nonzero(key >= 0).reshape(-1)
    k = key[valid].to(torch.int64)
    order = torch.sort(k, stable=True).indices
    eid = valid[order]
    ptr = torch.zeros(n_hits + 1, dtype=torch.int64, device=key.device)
    torch.cumsum(torch.bincount(k, minlength=n_hits), 0, out=ptr[1:])
    if int(ptr[-1]) >= 2 ** 31:
        raise ValueError("segment count exceeds int32 index range")
    return ptr.to(torch.int32), eid.to(torch.int32), other[eid].to(torch.int32)


class _EventLayout:
    """Per-graph offsets of a block-diagonal batch (HitGraphBatch.event_layout): max_hits, max_segments and - as
    device tensors, uploaded at first use (one copy for both; a single-graph forward never needs them) -
    hit_ptr / seg_ptr [G+1] int32."""

    def __init__(self, hp, sp, device=None, sizes=None):
        self._hp, self._sp = hp, sp
        if sizes is None:
            sizes = (int(np.diff(np.asarray(hp)).max(initial=0)), int(np.diff(np.asarray(sp)).max(initial=0)))
        self.max_hits, self.max_segments = sizes
        self._device, self._both = device, None

    def ptrs(self, device=None):
        device = self._device if device is None else device
        if self._both is None or self._both.device != torch.device(device):
            self._both = torch.from_numpy(np.stack([np.asarray(self._hp), np.asarray(self._sp)]).astype(_I32)).to(device)
        return self._both[0], self._both[1]

    hit_ptr = property(lambda self: self.ptrs()[0])
    seg_ptr = property(lambda self: self.ptrs()[1])


class HitGraphBatch:
    """A block-diagonal batch of hit graphs in index form (see module docstring)."""

    _TENSORS = ("X", "src", "dst", "y")        # (+ the six CSR arrays once they exist)

    def __init__(self, X, src, dst, y=None, hit_ptr=None, seg_ptr=None,
                 dense_shape=None, csr=None, _checked=False):
        X = np.ascontiguousarray(X, dtype=np.float32)
        src = np.ascontiguousarray(src, dtype=_I32)
        dst = np.ascontiguousarray(dst, dtype=_I32)
        if X.ndim != 2 or src.ndim != 1 or src.shape != dst.shape:
            raise ValueError("expected X [N,F], src [E], dst [E]")
        n = X.shape[0]
        if not _checked:                             # (from_graphs checks graph by graph, while a graph is in cache)
            if np.any((src ^ dst) < 0):              # signs differ: exactly one end negative
                raise ValueError("a padded segment must have src = dst = -1")
            if src.size and (src.max(initial=-1) >= n or dst.max(initial=-1) >= n):
                raise ValueError("segment endpoint out of range")
        self.n_hits, self.n_features = n, X.shape[1]
        self.n_segments = src.shape[0]
        self.hit_ptr = np.asarray([0, n] if hit_ptr is None else hit_ptr, dtype=np.int64)
        self.seg_ptr = np.asarray([0, self.n_segments] if seg_ptr is None else seg_ptr,
                                  dtype=np.int64)
        self.n_graphs = len(self.hit_ptr) - 1
        self.dense_shape = dense_shape  # (B, N_max, E_max) when built from a padded batch
        t = torch.from_numpy
        self.X, self.src, self.dst = t(X), t(src), t(dst)
        # The two CSRs serve the per-module kernels, the small-event kernel and the backward; the
        # tiled pipeline works from its own plan.  They are therefore built on first use (two
        # stable sorts of E keys: 15 s for 25.6 M segments) unless the caller brought them.
        self._csr = None if csr is None else tuple(t(np.ascontiguousarray(a, dtype=_I32)) for a in csr)
        self._src_host = src if csr is None else None
        self._dst_host = dst if csr is None else None
        self.y = None if y is None else t(np.ascontiguousarray(y, dtype=np.float32))
        self.plan = None

    _CSR_NAMES = ("in_ptr", "in_eid", "in_nbr", "out_ptr", "out_eid", "out_nbr")

    # Device-built lists: False = trust the endpoints (they were checked where the batch was made: on the host in
    # the constructor, by gnn_dense_to_index's flags for dense input; the builder skips malformed segments either
    # way), True = read the builder's status word back - 4 bytes and one synchronisation per batch.
    validate_csr = False

    def _ensure_csr(self):
        if self._csr is None:
            n = self.n_hits
            if self.X.is_cuda and os.environ.get("GNN_CSR_BUILDER", "hip") == "hip":
                # gnn_csr_build (csrc/csr_build.hip): counting sort + rank, a handful of launches, no read-back;
                # eid / nbr arrays keep all n_segments entries (the lists, then -1)
                from . import _lib
                parts = _lib.csr_build(self.src, self.dst, n)
                if self.validate_csr and int(parts[6].item()) & 1:
                    raise ValueError("segment endpoint out of range, or a segment with exactly one negative end")
                self._csr, self._csr_status = parts[:6], parts[6]
            elif self.X.is_cuda:     # torch ops (stable sorts on the GPU; GNN_CSR_BUILDER=torch): A / B runs
                self._csr = _csr_by_device(self.dst, self.src, n) + _csr_by_device(self.src, self.dst, n)
            else:
                src, dst = self._src_host, self._dst_host
                parts = _csr_by(dst, src, n) + _csr_by(src, dst, n)
                self._csr = tuple(torch.from_numpy(a) for a in parts)
                self._src_host = self._dst_host = None
            # (device-built lists: the host copies of the endpoints are NOT dropped here - handing 2 x 12.8 MB back to
            # the OS took 5 ms of the first forward / training step of a 3.2 M-segment batch born on the host, ten
            # times the list build itself; `to(device)` lets go of large ones where the upload is paid anyway)
        return self._csr

    in_ptr = property(lambda self: self._ensure_csr()[0])
    in_eid = property(lambda self: self._ensure_csr()[1])
    in_nbr = property(lambda self: self._ensure_csr()[2])
    out_ptr = property(lambda self: self._ensure_csr()[3])
    out_eid = property(lambda self: self._ensure_csr()[4])
    out_nbr = property(lambda self: self._ensure_csr()[5])

    def build_plan(self, hidden_dim, limits=None):
        """Tiles + windows + SELL-16 execution plan of the fused kernels, built once for the
        kernel shape (input_dim = n_features, hidden_dim): by HIP kernels when the batch lives on the
        GPU (plan_hip.py / csrc/plan_build.hip; GNN_PLAN_BUILDER=torch selects the torch-op builder
        plan_device.py, which also takes the batches outside the kernels' bounds), else on the host
        with numpy (plan.py, the specification); all three build the same plan, array for array."""
        if self.plan is None or self.plan.hidden_dim != hidden_dim:
            from . import _lib
            lim = _lib.plan_limits(self.n_features, hidden_dim)
            lim.update(limits or {})       # tests / experiments: e.g. iter_records=0 -> global mode
            builder = os.environ.get("GNN_PLAN_BUILDER", "hip" if self.X.is_cuda else "host")
            if os.environ.get("GNN_HOST_PLAN") or not self.X.is_cuda:
                builder = "host"
            plan = None
            if builder == "hip":          # HIP kernels (csrc/plan_build.hip): milliseconds
                from .plan_hip import HipSellPlan, PlanBuilderUnsupported
                try:
                    plan = HipSellPlan(self, lim)
                except PlanBuilderUnsupported:
                    builder = "torch"     # outside the kernels' static bounds, or empty
            if plan is None and builder == "torch":
                from .plan_device import DeviceSellPlan
                plan = DeviceSellPlan(self, lim)
            if plan is None:
                from .plan import SellPlan
                plan = SellPlan(self, lim)
            self.plan = plan
            self.plan.hidden_dim = hidden_dim
            self.plan.to(self.X.device)
        return self.plan

    def with_features(self, X):
        """The same graphs with other hit features (the same segments read out again: another calibration, another
        feature scaling): a new batch object that SHARES the index arrays, the segment lists and - through
        `SellPlan.with_features` - the structure of the execution plan; nothing is sorted or planned again."""
        X = torch.as_tensor(X).to(torch.float32)
        if tuple(X.shape) != tuple(self.X.shape):
            raise ValueError("expected X of shape %s" % (tuple(self.X.shape),))
        X = X.to(self.X.device).contiguous()
        b = HitGraphBatch.__new__(HitGraphBatch)
        b.__dict__.update(self.__dict__)
        b.X = X
        b._gstruct = b._gstruct_raw = None       # cached C structs hold the old X pointer
        b._twin = None
        b._forwards = getattr(self, "_forwards", 0)
        if self.plan is not None:
            b.plan = self.plan.with_features(X)
            b.plan.hidden_dim = self.plan.hidden_dim
        return b

    def level_ordered(self, hidden_dim=8):
        """The same batch in PLAN SPACE: hits numbered by the execution plan's padded ids - (graph, detector
        level), tiles, degree; `plan.n_pad` hits, the padding dummies without segments and with X = 0 - and
        segments sorted by end hit (stable: a hit's incoming segments keep the caller's order, which is the order
        of its list in the plan).  In that order the neighbours of consecutive hits lie in narrow id ranges, which
        makes the training kernels' record gathers L2-local (c3 x 32 training step 1.68 -> 1.46 ms), the scores of
        a hit's incoming segments are contiguous, and - round 3 - the training forward can run the fused tile
        kernels on the ORIGINAL batch's plan (`_fused`: gnn_segclf_forward_train_plan writes e_t / H_t / Q_t
        straight in this batch's numbering).  Scores are handed back per segment in the caller's order
        (`seg_order` / `seg_rank`) and weight gradients do not depend on numbering, so the twin is a drop-in for the
        training forward / backward.  Built once per batch (one plan + GPU sorts for its CSRs) and cached; returns
        self when there is nothing to gain (CPU batch, no segments, no plan for this shape)."""
        twin = getattr(self, "_twin", None)
        if twin is not None:
            return twin
        self._twin = self
        if not self.X.is_cuda or self.n_hits == 0 or self.n_segments == 0:
            return self
        try:
            plan = self.build_plan(hidden_dim)
        except Exception:                      # no fused kernels for this shape: keep the caller's order
            return self
        dev = self.X.device
        n_pad = int(plan.n_pad)
        perm = plan.perm.to(torch.int64)                          # padded id -> caller's hit id, -1 = dummy
        new_ids = torch.nonzero(perm >= 0).reshape(-1)
        rank = torch.empty(self.n_hits, dtype=torch.int64, device=dev)
        rank[perm[new_ids]] = new_ids                             # caller's hit id -> padded id
        src, dst = self.src.to(torch.int64), self.dst.to(torch.int64)
        t = HitGraphBatch.__new__(HitGraphBatch)
        t.__dict__.update({k: v for k, v in self.__dict__.items() if not k.startswith("_")})
        t.n_hits = n_pad
        t.X = plan.X[:n_pad].contiguous()                         # the plan's renumbered rows (dummies: zeros)
        ts = torch.where(src >= 0, rank[src.clamp_min(0)], src)
        td = torch.where(dst >= 0, rank[dst.clamp_min(0)], dst)
        # segments sorted by end hit, padded ones last, ties in the caller's order: seg_order[k] = caller's index
        # of the twin's segment k; the autograd function hands scores back (and takes their gradient) in the
        # caller's order
        key = torch.where(ts >= 0, td, torch.full_like(td, 2 ** 62))
        seg_order = torch.argsort(key, stable=True)
        t.src = ts[seg_order].to(torch.int32).contiguous()
        t.dst = td[seg_order].to(torch.int32).contiguous()
        t.seg_order = seg_order
        t.seg_rank = torch.empty_like(seg_order)
        t.seg_rank[seg_order] = torch.arange(self.n_segments, dtype=torch.int64, device=dev)
        if getattr(self, "y", None) is not None and torch.is_tensor(self.y) and self.y.numel() == self.n_segments:
            t.y = self.y.to(dev)[seg_order]
        t._csr = None
        t._src_host = t._dst_host = None
        t._gstruct = None
        t._event = (None,)                     # detector-size graphs: never the one-launch kernels
        t.plan = None
        t._fused = plan                        # the plan whose padded ids ARE this batch's hit ids
        t._fused_dim = hidden_dim
        t._twin = t
        self._twin = t
        return t

    def event_layout(self):
        """Per-graph layout for the one-workgroup-per-graph kernel (`_lib.segclf_forward_events`):
        device copies of hit_ptr / seg_ptr and the largest graph's size, or None when the batch is
        not block-diagonal in the sense that kernel needs (every segment of graph i inside
        [seg_ptr[i], seg_ptr[i+1]) joins two hits of graph i; padded segments are fine).
        Checked once, on the host."""
        if getattr(self, "_event", None) is None and self.n_graphs == 1:
            # one graph (a single event through the model): nothing to check beyond what the constructor checked,
            # and no numpy passes on the way to the first launch
            hp, sp = self.hit_ptr, self.seg_ptr
            from . import _lib
            ok = (hp[0] == 0 and sp[0] == 0 and hp[1] == self.n_hits and sp[1] == self.n_segments and
                  self.n_hits < 2 ** 31 and self.n_segments <= _lib.EVENTS_MAX_SEGMENTS)     # (larger: never a layout)
            self._event = (_EventLayout(hp, sp, sizes=(self.n_hits, self.n_segments)) if ok else None,)
        if getattr(self, "_event", None) is None:
            hp, sp = self.hit_ptr, self.seg_ptr
            ok = (hp[0] == 0 and sp[0] == 0 and hp[-1] == self.n_hits and
                  sp[-1] == self.n_segments and np.all(np.diff(hp) >= 0) and np.all(np.diff(sp) >= 0)
                  and self.n_hits < 2 ** 31 and self.n_segments < 2 ** 31)
            if ok:
                # graphs too large for the one-workgroup-per-graph kernels never take them: skip the
                # endpoint check, which copies src / dst to the host (50 us and a synchronisation on the
                # first forward of every detector-size batch)
                from . import _lib
                if int(np.diff(sp).max(initial=0)) > _lib.EVENTS_MAX_SEGMENTS:
                    ok = False
            if ok and self.n_segments and self.n_graphs > 1:
                # (one graph: its endpoints were range-checked where the batch was made - nothing left to check;
                # several: the host copies the constructor kept, else one copy back from the device)
                src = self._src_host if self._src_host is not None else self.src.cpu().numpy()
                dst = self._dst_host if self._dst_host is not None else self.dst.cpu().numpy()
                gseg = np.repeat(np.arange(self.n_graphs), np.diff(sp))
                lo, hi = hp[:-1][gseg], hp[1:][gseg]
                pad = src < 0
                ok = bool(np.all(pad | ((src >= lo) & (src < hi) & (dst >= lo) & (dst < hi))))
            self._event = (_EventLayout(hp, sp) if ok else None,)
        lay = self._event[0]
        if lay is not None:
            lay._device = self.X.device          # where hit_ptr / seg_ptr are uploaded when somebody asks for them
        return lay

    # -- constructors ------------------------------------------------------------------
    @classmethod
    def from_graphs(cls, graphs, pad_segments=False, pin_memory=None):
        """Index-form batcher: block-diagonal concatenation.

        Replaces the zero-padding `merge_graphs` (reference
        gnn/trainSegmentClassifier.py:66-95): batch composition order is the list
        order (`graphs[j:j+batch_size]`, :103-104), features cast to float32 (:38-44).
        `pad_segments=False`: no padding at all (scores [E_total]).
        `pad_segments=True`: every graph's segment list is padded to the batch's E_max with
        `src = dst = -1` columns and `dense_shape = (B, N_max, E_max)` is set, so the model
        returns the reference's [B, E_max] layout (padded columns score sigmoid(W2 tanh(b1) + b2)
        and carry target 0, exactly like merge_graphs' zero columns); hits are never padded.
        """
        B = len(graphs)
        hit_ptr = np.zeros(B + 1, dtype=np.int64)
        seg_ptr = np.zeros(B + 1, dtype=np.int64)
        e_max = max((int(np.asarray(g.src).shape[0]) for g in graphs), default=0)
        for i, g in enumerate(graphs):
            hit_ptr[i + 1] = hit_ptr[i] + g.X.shape[0]
            seg_ptr[i + 1] = seg_ptr[i] + (e_max if pad_segments else g.src.shape[0])
        # Large batches are put together in PINNED host memory when a GPU is there to receive them (pin_memory=None:
        # from 1 M segments on; torch's caching host allocator hands the same blocks out again batch after batch): the
        # upload of 256 detector graphs then runs at the link's rate, 21 -> 6 ms.
        E_all = int(seg_ptr[-1])
        pin = bool(pin_memory) if pin_memory is not None else (E_all >= (1 << 20) and torch.cuda.is_available())

        def host(shape, dtype, fill=None):
            if pin:
                a = torch.empty(shape, dtype={np.float32: torch.float32, _I32: torch.int32}[dtype], pin_memory=True).numpy()
                if fill is not None:
                    a.fill(fill)
                return a
            return np.empty(shape, dtype=dtype) if fill is None else np.full(shape, fill, dtype=dtype)

        n_feat = int(np.asarray(graphs[0].X).shape[1]) if graphs else 0
        X = host((int(hit_ptr[-1]), n_feat), np.float32)
        for i, g in enumerate(graphs):
            X[hit_ptr[i]:hit_ptr[i + 1]] = np.asarray(g.X, dtype=np.float32)
        ys = [getattr(g, "y", None) for g in graphs]
        have_y = not any(v is None for v in ys)

        if int(hit_ptr[-1]) >= 2 ** 31 or int(seg_ptr[-1]) >= 2 ** 31:
            raise ValueError("batch outside the int32 index range")

        # every graph's endpoints + its hit offset, written straight into its slice of ONE int32 array per end (the
        # first version went through int64 copies, np.where and a concatenate: 25 ms per million segments - more than
        # the GPU needs to plan AND score the batch), and checked graph by graph while the graph is in cache (the
        # constructor's checks are whole-batch passes)
        E_tot = int(seg_ptr[-1])
        src = host(E_tot, _I32, -1 if pad_segments else None)
        dst = host(E_tot, _I32, -1 if pad_segments else None)
        for i, g in enumerate(graphs):
            a, b = np.asarray(g.src), np.asarray(g.dst)
            if a.ndim != 1 or a.shape != b.shape:
                raise ValueError("expected src [E], dst [E]")
            e, o, off, n_g = a.shape[0], int(seg_ptr[i]), int(hit_ptr[i]), int(g.X.shape[0])
            if e == 0:
                continue
            amin, bmin = int(a.min()), int(b.min())
            if int(a.max()) >= n_g or int(b.max()) >= n_g:
                raise ValueError("segment endpoint out of range")
            np.add(a, off, out=src[o:o + e], casting="unsafe")
            np.add(b, off, out=dst[o:o + e], casting="unsafe")
            if amin < 0 or bmin < 0:                     # padded segments of the graph itself stay -1
                pa = a < 0
                if np.any(pa != (b < 0)):
                    raise ValueError("a padded segment must have src = dst = -1")
                src[o:o + e][pa] = -1
                dst[o:o + e][pa] = -1
        y = None
        if have_y:
            y = host(E_tot, np.float32, 0.0 if pad_segments else None)
            for i, v in enumerate(ys):
                v = np.asarray(v, dtype=np.float32)
                y[int(seg_ptr[i]):int(seg_ptr[i]) + v.shape[0]] = v
        dense_shape = None
        if pad_segments:
            n_max = max((int(g.X.shape[0]) for g in graphs), default=0)
            dense_shape = (B, n_max, e_max)
        return cls(X, src, dst, y=y, hit_ptr=hit_ptr, seg_ptr=seg_ptr, dense_shape=dense_shape, _checked=True)

    def to_padded(self, flat, fill=0.0):
        """[E_total] values in this batch's segment order -> [B, E_max] (the reference's target /
        score layout, gnn/trainSegmentClassifier.py:86,110); `fill` in the padded entries."""
        if self.dense_shape is not None:
            return flat.reshape(self.dense_shape[0], self.dense_shape[2])
        counts = np.diff(self.seg_ptr)
        B, e_max = self.n_graphs, int(counts.max(initial=0))
        out = flat.new_full((B, e_max), fill) if torch.is_tensor(flat) else np.full((B, e_max), fill,
                                                                                 dtype=np.asarray(flat).dtype)
        for i in range(B):
            out[i, :counts[i]] = flat[self.seg_ptr[i]:self.seg_ptr[i + 1]]
        return out

    def from_padded(self, padded, counts=None):
        """[B, E_max] -> flat values of the real segments, graph by graph (`counts`: segments per
        graph; default: this batch's own when it is unpadded)."""
        if counts is None:
            if self.dense_shape is not None:
                raise ValueError("a padded batch does not know its graphs' true segment counts")
            counts = np.diff(self.seg_ptr)
        parts = [padded[i, :int(c)] for i, c in enumerate(counts)]
        return torch.cat(parts) if torch.is_tensor(padded) else np.concatenate(parts)

    @classmethod
    def from_sparse_arrays(cls, X, Ri_rows, Ri_cols, Ro_rows, Ro_cols, y=None):
        """From the reference's on-disk `SparseGraph` arrays (gnn/graph.py:20-26,179-194).

        `Ri_rows` (end hits) and `Ro_rows` (start hits) are sorted ascending because
        `nonzero()` is row-major, so they are CSR order already: rowptr is a bincount.
        Unsorted input (hand-made files) is handled by a stable sort.
        """
        X = np.asarray(X, dtype=np.float32)
        n = X.shape[0]
        e = int(np.asarray(Ri_rows).shape[0])
        if np.asarray(Ro_rows).shape[0] != e:
            raise ValueError("Ri and Ro must describe the same segments")

        def one(rows, cols):
            rows = np.asarray(rows, dtype=np.int64)
            cols = np.asarray(cols, dtype=np.int64)
            if rows.size and np.any(np.diff(rows) < 0):
                o = np.lexsort((cols, rows))
                rows, cols = rows[o], cols[o]
            ptr = np.zeros(n + 1, dtype=np.int64)
            np.cumsum(np.bincount(rows, minlength=n), out=ptr[1:])
            end = np.full(e, -1, dtype=np.int64)
            end[cols] = rows
            return ptr.astype(_I32), cols.astype(_I32), end

        in_ptr, in_eid, dst = one(Ri_rows, Ri_cols)
        out_ptr, out_eid, src = one(Ro_rows, Ro_cols)
        if np.any(src < 0) or np.any(dst < 0):
            raise ValueError("every segment needs exactly one start and one end hit")
        csr = (in_ptr, in_eid, src[in_eid].astype(_I32),
               out_ptr, out_eid, dst[out_eid].astype(_I32))
        return cls(X, src, dst, y=y, csr=csr)

    @classmethod
    def from_npz(cls, filename):
        """Read one graph file written by the reference's `save_graph` (gnn/graph.py:179-194) or its
        muon variant (gnn/Muon_graph.py:198-217, `SparseGraphProp`: + `pt`, `eta`)."""
        with np.load(filename) as f:  # allow_pickle stays False
            b = cls.from_sparse_arrays(f["X"], f["Ri_rows"], f["Ri_cols"],
                                       f["Ro_rows"], f["Ro_cols"],
                                       y=f["y"] if "y" in f.files else None)
            # the muon writer adds the generated muon's pt and eta (gnn/Muon_graph.py:198-205)
            b.pt = float(f["pt"]) if "pt" in f.files else None
            b.eta = float(f["eta"]) if "eta" in f.files else None
            return b

    @classmethod
    def from_dense(cls, X, Ri, Ro, y=None):
        """Compatibility adapter for the reference's dense inputs [B,N,F], [B,N,E], [B,N,E].

        Column j of Ri/Ro holds a single 1 at the end/start hit (gnn/graph.py:132-135);
        an all-zero column is a padded segment (gnn/trainSegmentClassifier.py:83-93)
        and becomes src = dst = -1.  O(B*N*E) reads: kept off the benchmarked path.
        """
        as_t = lambda a: a.detach() if torch.is_tensor(a) else torch.from_numpy(np.asarray(a))
        Xt, Rit, Rot = as_t(X), as_t(Ri), as_t(Ro)
        if Xt.dim() == 2:
            Xt, Rit, Rot = Xt[None], Rit[None], Rot[None]
        B, N, _ = Xt.shape
        E = Rit.shape[2]
        if tuple(Rit.shape) != (B, N, E) or tuple(Rot.shape) != (B, N, E):
            raise ValueError("expected X [B,N,F], Ri [B,N,E], Ro [B,N,E]")
        if Rit.is_cuda and Rot.is_cuda and Xt.is_cuda:
            return cls._from_dense_device(Xt, Rit, Rot, y)
        Xn = Xt.cpu().numpy()
        off = (np.arange(B, dtype=np.int64) * N)[:, None]

        def ends(R):   # the O(B*N*E) reduction runs where the matrix lives; [B,E] comes back
            nz = R != 0
            cnt = nz.sum(dim=1).cpu().numpy()
            if np.any(cnt > 1):
                raise ValueError("incidence matrix column with more than one hit")
            idx = nz.to(torch.uint8).argmax(dim=1).cpu().numpy().astype(np.int64) + off
            return np.where(cnt == 1, idx, -1).reshape(-1)

        dst, src = ends(Rit), ends(Rot)
        if np.any((src < 0) != (dst < 0)):
            raise ValueError("a segment column must be set in both Ri and Ro or in neither")
        yy = None if y is None else (y.detach().cpu().numpy() if torch.is_tensor(y) else
                                     np.asarray(y)).reshape(-1)
        return cls(Xn.reshape(B * N, -1), src, dst, y=yy,
                   hit_ptr=np.arange(B + 1) * N, seg_ptr=np.arange(B + 1) * E,
                   dense_shape=(B, N, E))

    validate_dense = True      # read the conversion kernel's error flag back (4 bytes, one sync per batch)

    @classmethod
    def _from_dense_device(cls, Xt, Rit, Rot, y=None):
        """from_dense for matrices that already live on the GPU (what an unchanged estimator.py loop
        hands over after `np_to_torch(...).cuda()`): one HIP kernel (`gnn_dense_to_index`), nothing
        crosses PCIe except - with `validate_dense` - the 4-byte error flag."""
        from . import _lib
        B, N, F = Xt.shape
        E = Rit.shape[2]
        f32 = lambda t: t.to(torch.float32).contiguous()          # noqa: E731
        src, dst, flags = _lib.dense_to_index(f32(Rit), f32(Rot))
        if cls.validate_dense:
            fl = int(flags.item())
            if fl & 1:
                raise ValueError("incidence matrix column with more than one hit")
            if fl & 2:
                raise ValueError("a segment column must be set in both Ri and Ro or in neither")
        self = cls.__new__(cls)
        self.n_hits, self.n_features, self.n_segments = B * N, F, B * E
        self.hit_ptr = np.arange(B + 1, dtype=np.int64) * N
        self.seg_ptr = np.arange(B + 1, dtype=np.int64) * E
        self.n_graphs = B
        self.dense_shape = (B, N, E)
        self.X = f32(Xt).reshape(B * N, F)
        self.src, self.dst = src, dst
        self._csr = None
        self._src_host = self._dst_host = None
        self.y = None if y is None else (y.detach() if torch.is_tensor(y) else torch.from_numpy(np.asarray(y))
                                         ).to(torch.float32).reshape(-1).to(Xt.device)
        self.plan = None
        # block-diagonal by construction: no host check of the endpoints
        self._event = (_EventLayout(self.hit_ptr, self.seg_ptr, Xt.device),)
        return self

    @classmethod
    def _from_device_arrays(cls, X, src, dst, y, hit_ptr, seg_ptr, dense_shape=None):
        """A batch from arrays that already live where they will be used (batcher.GraphStore): torch tensors X
        [N, F] float32, src / dst [E] int32 in batch numbering (-1 = padded), y [E] or None; hit_ptr / seg_ptr host
        arrays.  The endpoints were checked when the store was built: block-diagonal by construction."""
        self = cls.__new__(cls)
        self.n_hits, self.n_features, self.n_segments = int(X.shape[0]), int(X.shape[1]), int(src.shape[0])
        self.hit_ptr = np.asarray(hit_ptr, dtype=np.int64)
        self.seg_ptr = np.asarray(seg_ptr, dtype=np.int64)
        self.n_graphs = len(self.hit_ptr) - 1
        self.dense_shape = dense_shape
        self.X = X.to(torch.float32).contiguous()
        self.src, self.dst = src.to(torch.int32).contiguous(), dst.to(torch.int32).contiguous()
        self._csr = None
        self._src_host = self._dst_host = None
        self.y = None if y is None else y.to(torch.float32).contiguous()
        self.plan = None
        if X.is_cuda:
            self._event = (_EventLayout(self.hit_ptr, self.seg_ptr, X.device),)
        else:                        # a CPU batch builds its lists on the host from these
            self._src_host, self._dst_host = self.src.numpy(), self.dst.numpy()
        return self

    # -- device movement -----------------------------------------------------------------
    def to(self, device):
        for k in self._TENSORS:
            v = getattr(self, k)
            if v is not None:
                setattr(self, k, v.to(device))
        if torch.device(device).type == "cuda" and self.n_segments >= (1 << 20):
            # the host copies served the host CSR builder and the event layout's endpoint check (graphs of this size
            # never take the event kernels): freed with the upload, not inside the first forward
            self._src_host = self._dst_host = None
        if self._csr is not None:
            self._csr = tuple(a.to(device) for a in self._csr)
        if self.plan is not None:
            self.plan.to(device)
        self._gstruct = None         # cached C struct of raw device pointers (_lib.cached_graph_struct)
        self._gstruct_raw = None
        self._twin = None            # the level-ordered twin lives on the old device
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda", torch.cuda.current_device() if device is None
                                    else device))

    @property
    def device(self):
        return self.X.device

    def split_scores(self, e):
        """Per-graph views of a flat score vector [E]."""
        return [e[self.seg_ptr[i]:self.seg_ptr[i + 1]] for i in range(self.n_graphs)]

"""Execution plan for the fused message-passing kernels: relabelled hits + SELL-16 lists.

Built once per batch on the host (like the CSR of `hitgraph.py`), reused by every
iteration and every epoch.  Nothing the caller sees changes: scores are per segment,
in the caller's segment order; hit numbering is internal to the plan.

1. Relabel.  Inside each graph, hits are renumbered by (in-degree, out-degree)
   descending.  A wavefront processes a *slice* of 16 consecutive hits (4 lanes per hit),
   so after the sort all 16 hits of a slice have (nearly) the same number of segments:
   no divergence in the neighbour loop and almost no padding.
2. SELL-16 ("sliced ELLPACK", slice height 16).  For slice s the neighbour list is stored
   column-major: entry k of hit i of the slice sits at `off[s] + 16*k + i`, so the 16
   hits of a wavefront read 64 contiguous bytes per step.  Lists are padded to the
   slice's longest list with the NULL hit id `n_hits`, whose stored record is
   [P = b1, R = 0]: a padded step adds exactly 0.
   Within a hit, neighbours keep ascending segment id (fixed summation order).
3. `src`/`dst` of every segment in the new numbering (caller's segment order) for the
   final edge pass; padded segments (-1) map to the NULL hit.

Arrays (all int32 / float32, `to(device)` uploads them):
    X        [n_hits+1, F]  relabelled features, last row zero (NULL hit)
    src,dst  [n_segments]   relabelled endpoints, NULL hit for padded segments
    in_off   [n_slices+1]   SELL offsets of the lists of segments ENDING at a hit
    in_nbr   [in_off[-1]]   start hit of each such segment (or NULL)
    out_off / out_nbr       the same for segments STARTING at a hit (entries = end hits)
    perm     [n_hits]       new id -> caller's hit id (tests / traces only)
"""
import numpy as np
import torch

SLICE = 16


def _sell(key_new, other_new, n_hits, n_slices):
    """SELL-16 lists: for every hit (new ids) the `other_new` endpoint of each valid segment
    whose `key_new` endpoint is that hit, ascending segment id."""
    valid = np.flatnonzero(key_new < n_hits)
    k = key_new[valid]
    order = np.argsort(k, kind="stable")
    k = k[order]
    oth = other_new[valid][order]
    deg = np.bincount(k, minlength=n_slices * SLICE)
    ptr = np.zeros(deg.shape[0] + 1, dtype=np.int64)
    np.cumsum(deg, out=ptr[1:])
    pos = np.arange(k.shape[0], dtype=np.int64) - ptr[k]          # rank inside the hit's list
    slen = deg.reshape(n_slices, SLICE).max(axis=1).astype(np.int64)
    off = np.zeros(n_slices + 1, dtype=np.int64)
    np.cumsum(slen * SLICE, out=off[1:])
    if off[-1] >= 2 ** 31:
        raise ValueError("SELL list exceeds int32 index range")
    nbr = np.full(int(off[-1]), n_hits, dtype=np.int32)           # NULL-padded
    nbr[off[k // SLICE] + pos * SLICE + (k % SLICE)] = oth
    return off.astype(np.int32), nbr


class SellPlan:
    def __init__(self, batch):
        """batch: a HitGraphBatch (host tensors)."""
        src = batch.src.cpu().numpy().astype(np.int64)
        dst = batch.dst.cpu().numpy().astype(np.int64)
        X = batch.X.cpu().numpy()
        n = batch.n_hits
        ok = src >= 0
        deg_in = np.bincount(dst[ok], minlength=n)
        deg_out = np.bincount(src[ok], minlength=n)
        gid = np.zeros(n, dtype=np.int64)
        if batch.n_graphs > 1:
            gid[batch.hit_ptr[1:-1]] = 1
            gid = np.cumsum(gid)
        perm = np.lexsort((-deg_out, -deg_in, gid))                # new id -> old id
        inv = np.empty(n + 1, dtype=np.int64)
        inv[perm] = np.arange(n)
        inv[n] = n                                                 # -1 (padded) -> NULL
        self.n_hits, self.n_segments = n, batch.n_segments
        self.n_features = batch.n_features
        self.n_slices = (n + SLICE - 1) // SLICE
        src_new = inv[np.where(ok, src, n)]
        dst_new = inv[np.where(ok, dst, n)]
        in_off, in_nbr = _sell(dst_new, src_new, n, self.n_slices)
        out_off, out_nbr = _sell(src_new, dst_new, n, self.n_slices)
        Xp = np.zeros((n + 1, X.shape[1]), dtype=np.float32)
        Xp[:n] = X[perm]
        t = torch.from_numpy
        self.X = t(Xp)
        self.src, self.dst = t(src_new.astype(np.int32)), t(dst_new.astype(np.int32))
        self.in_off, self.in_nbr = t(in_off), t(in_nbr)
        self.out_off, self.out_nbr = t(out_off), t(out_nbr)
        self.perm = t(perm.astype(np.int32))
        self.padding = (int(in_off[-1]) + int(out_off[-1])) / max(1, 2 * int(ok.sum())) - 1.0

    _TENSORS = ("X", "src", "dst", "in_off", "in_nbr", "out_off", "out_nbr", "perm")

    def to(self, device):
        for k in self._TENSORS:
            setattr(self, k, getattr(self, k).to(device))
        return self

    @property
    def device(self):
        return self.X.device

"""Execution plan for the fused message-passing kernels: tiles, windows, SELL-16 lists.

Built once per batch on the host (like the CSR of `hitgraph.py`), reused by every
iteration and every epoch.  Nothing the caller sees changes: scores are per segment, in
the caller's segment order; hit numbering is internal to the plan.

1. Levels.  Every hit gets a topological level (longest path from a hit without incoming
   segments; for a layered detector graph this recovers the detector layer).  Hits are
   ordered by (graph, level): all segments then join hits of neighbouring id ranges.
2. Tiles.  Consecutive (graph, level) units are merged greedily into tiles of at most
   `tile_hits` hits (one workgroup each); a unit larger than that is split.  A tile's
   neighbour records therefore lie in two contiguous id windows - the start hits of its
   incoming segments (`in` window) and the end hits of its outgoing segments (`out`
   window).  If both windows fit the LDS budget the tile runs in LDS mode: the kernel
   stages the windows with coalesced reads and gathers from LDS; lists then hold
   window-relative indices.  Otherwise (irregular graph) the tile gathers from global
   memory and lists hold absolute ids.
3. Inside a tile hits are sorted by (in-degree, out-degree) descending and the tile is
   padded to a multiple of 16 hits, so each 16-hit slice (one wavefront, 4 lanes per hit)
   has near-uniform list lengths: SELL-16 ("sliced ELLPACK") with ~3 % padding.  Entry k of
   hit i of slice s sits at `off[s] + 16*k + i`; padded entries point at the window's NULL
   record (R/S half zero: adds exactly 0).  Neighbours keep ascending segment id.
4. Final edge pass: segments stay in the caller's order, cut into chunks of `chunk_segments`;
   per chunk the src / dst windows are computed the same way (LDS or global mode).

All arrays int32 / float32; `to(device)` uploads them.  `n_pad` = padded hit count; the
NULL hit has id `n_pad`.
"""
import numpy as np
import torch

SLICE = 16
TILE_DESC = 8    # ints per tile:  slice_begin, slice_end, in_lo, in_cnt, out_lo, out_cnt, mode, sched_base
CHUNK_DESC = 8   # ints per chunk: seg_begin, seg_end, src_lo, src_cnt, dst_lo, dst_cnt, mode, 0


def stable_argsort(a):
    """np.argsort(a, kind="stable") for large arrays of non-negative ids that fit int32: torch's
    CPU sort is a parallel radix sort there (10x numpy's merge sort at 25 M keys)."""
    a = np.asarray(a)
    if a.size >= (1 << 16) and a.size and a.min(initial=0) >= 0 and a.max(initial=0) < 2 ** 31:
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).argsort(stable=True).numpy()
    return np.argsort(a, kind="stable")


def _run_starts(sorted_keys):
    """(unique values, index of their first occurrence) of an already SORTED array."""
    if sorted_keys.size == 0:
        return sorted_keys[:0], np.zeros(0, dtype=np.int64)
    start = np.flatnonzero(np.r_[True, sorted_keys[1:] != sorted_keys[:-1]])
    return sorted_keys[start], start


def topological_levels(src, dst, n, max_iter=64):
    """level[n] = longest path (in segments) from a hit with no incoming segment.
    Graphs with cycles stop at `max_iter` (any labelling is valid, only locality suffers)."""
    level = np.zeros(n, dtype=np.int32)
    if src.size == 0:
        return level
    order = stable_argsort(dst)
    s, d = src[order], dst[order]
    uniq, start = _run_starts(d)
    for _ in range(max_iter):
        cand = np.maximum.reduceat(level[s] + 1, start)
        new = level.copy()
        new[uniq] = np.maximum(new[uniq], cand)
        if np.array_equal(new, level):
            break
        level = new
    # a hit without incoming segments would sit at level 0 whatever its layer and stretch the
    # windows of that tile over the whole graph: place it one level below its nearest end hit
    no_in = np.ones(n, dtype=bool)
    no_in[uniq] = False
    so = stable_argsort(src)
    us, st = _run_starts(src[so])
    down = np.minimum.reduceat(level[dst[so]], st) - 1
    fix = no_in[us]
    level[us[fix]] = np.maximum(down[fix], 0)
    return level


def _node_minmax(key, other, n):
    """Per hit: min / max of `other` over segments whose `key` endpoint is the hit."""
    order = stable_argsort(key)
    k, o = key[order], other[order]
    lo = np.full(n, np.iinfo(np.int64).max, dtype=np.int64)
    hi = np.full(n, -1, dtype=np.int64)
    if k.size:
        uniq, start = _run_starts(k)
        lo[uniq] = np.minimum.reduceat(o, start)
        hi[uniq] = np.maximum.reduceat(o, start)
    return lo, hi


def _range_minmax(lo, hi, bounds):
    """min(lo) / max(hi) over consecutive ranges [bounds[i], bounds[i+1])."""
    rl = np.minimum.reduceat(lo, bounds[:-1])
    rh = np.maximum.reduceat(hi, bounds[:-1])
    return rl, rh


def _sell(key_new, other_rel, n_pad, null_of_slice):
    """SELL-16 lists over padded hit ids; `other_rel` is the stored value per segment,
    `null_of_slice[s]` the padding value of slice s."""
    n_slices = n_pad // SLICE
    order = stable_argsort(key_new)
    k = key_new[order]
    oth = other_rel[order]
    deg = np.bincount(k, minlength=n_pad)
    ptr = np.zeros(n_pad + 1, dtype=np.int64)
    np.cumsum(deg, out=ptr[1:])
    pos = np.arange(k.shape[0], dtype=np.int64) - ptr[k]
    slen = deg.reshape(n_slices, SLICE).max(axis=1).astype(np.int64) if n_slices else \
        np.zeros(0, np.int64)
    off = np.zeros(n_slices + 1, dtype=np.int64)
    np.cumsum(slen * SLICE, out=off[1:])
    if off[-1] >= 2 ** 31:
        raise ValueError("SELL list exceeds int32 index range")
    nbr = np.repeat(null_of_slice.astype(np.int32), (slen * SLICE).astype(np.int64))
    nbr[off[k // SLICE] + pos * SLICE + (k % SLICE)] = oth
    # the kernel reads lists in chunks of 4 steps: up to 3 steps past the end of the last list
    nbr = np.concatenate([nbr, np.zeros(4 * SLICE, dtype=np.int32)])
    return off.astype(np.int32), nbr


def _pack16(off, nbr, null_of_slice):
    """16-bit packed copy of SELL-16 lists for the phase-split kernel (k_iter2).

    Steps are padded per slice to a multiple of 8 with the slice's NULL entry and consecutive
    step pairs share one 32-bit word: word p of hit i of a slice holds steps 2p (low half) and
    2p+1 (high half) at `off16[s] + 16*p + i`.  Lane q of a quad loads word 4*sc+q of
    super-chunk sc, i.e. one load covers 8 list steps.  Entries are window-relative (< 65536)."""
    n_slices = off.shape[0] - 1
    L = np.diff(off.astype(np.int64)) // SLICE                 # steps per slice
    steps8 = (L + 7) // 8 * 8
    poff = np.zeros(n_slices + 1, dtype=np.int64)
    np.cumsum(steps8 * SLICE, out=poff[1:])
    pad = np.repeat((null_of_slice.astype(np.int64) & 0xFFFF), steps8 * SLICE)
    real = int(off[-1])
    if real:
        s_of = np.repeat(np.arange(n_slices), L * SLICE)
        idx = np.arange(real, dtype=np.int64)
        pad[idx - off.astype(np.int64)[s_of] + poff[s_of]] = nbr[:real].astype(np.int64) & 0xFFFF
    pr = pad.reshape(-1, 2, SLICE)
    words = (pr[:, 0, :] | (pr[:, 1, :] << 16)).reshape(-1)
    words = np.concatenate([words, np.zeros(64, dtype=np.int64)])
    return (poff // 2).astype(np.int32), words.astype(np.uint32).view(np.int32)


def _chunk_bounds(key, ok, E, CH):
    """Chunk boundaries of the final edge pass.  Segments arrive grouped (the reference emits
    them per layer pair, gnn/graph.py:80-93): a run of equal (graph, start level) key has all its
    start hits in one level and its end hits in the next, i.e. the narrowest possible windows.
    Runs of at least CH/4 segments keep their own boundaries; everything is then cut into equal
    parts of at most CH segments."""
    if E == 0:
        return np.zeros(1, np.int64)
    k = key.copy()
    if not ok.all():                                   # padded segments join the run before them
        idx = np.where(ok, np.arange(E), 0)
        np.maximum.accumulate(idx, out=idx)
        k = key[idx]
    rb = np.flatnonzero(np.r_[True, k[1:] != k[:-1]])  # run starts
    rsz = np.diff(np.r_[rb, E])
    big = rsz >= max(1, CH // 4)
    marks = np.unique(np.r_[0, rb[big], (rb + rsz)[big], E])
    out = [np.zeros(1, np.int64)]
    for a, b in zip(marks[:-1].tolist(), marks[1:].tolist()):
        parts = -(-(b - a) // CH)
        out.append(a + (np.arange(1, parts + 1, dtype=np.int64) * (b - a)) // parts)
    return np.concatenate(out)


class SellPlan:
    def __init__(self, batch, limits):
        """batch: HitGraphBatch; limits: dict(tile_hits, iter_records, chunk_segments,
        edge_records) from the library (`_lib.plan_limits(F, D)`): LDS budgets in records."""
        src = batch.src.cpu().numpy().astype(np.int64)
        dst = batch.dst.cpu().numpy().astype(np.int64)
        X = batch.X.cpu().numpy()
        n, E = batch.n_hits, batch.n_segments
        tile_hits = int(limits["tile_hits"])
        # small batches: shrink the tiles so that there are a few hundred workgroups to spread
        # over the CUs (a 10k-hit graph would otherwise be 10 workgroups); windows stay whole
        # levels, so this costs staging traffic that only matters once the chip is full anyway
        # (wide shapes - no LDS windows, one workgroup per CU, a 70-100 KB weight table to stage per
        # workgroup - want one tile per CU rather than two)
        want = 512 if int(limits["iter_records"]) > 0 else 256
        if n < want * tile_hits:
            tile_hits = max(64, ((n + want - 1) // want + SLICE - 1) // SLICE * SLICE)
        ok = src >= 0
        vs, vd = src[ok], dst[ok]
        deg_in = np.bincount(vd, minlength=n)
        deg_out = np.bincount(vs, minlength=n)
        gid = np.zeros(n, dtype=np.int64)
        if batch.n_graphs > 1:
            np.add.at(gid, batch.hit_ptr[1:-1][batch.hit_ptr[1:-1] < n], 1)
            gid = np.cumsum(gid)
        level = topological_levels(vs, vd, n)

        # -- (graph, level) units -> tiles ------------------------------------------------
        # units are degree-sorted inside, so a unit that is cut into several tiles yields tiles of
        # near-uniform list lengths
        base = np.lexsort((-deg_out, -deg_in, level, gid))    # position -> old id
        ukey = gid[base] * (int(level.max(initial=0)) + 1) + level[base]
        ustart = np.flatnonzero(np.r_[True, ukey[1:] != ukey[:-1]]) if n else np.zeros(0, np.int64)
        usize = np.diff(np.r_[ustart, n])
        tile_bounds = [0]                                     # in base positions
        cur = 0
        for st, sz in zip(ustart.tolist(), usize.tolist()):
            if sz > tile_hits:                                # split a big unit
                if cur:
                    tile_bounds.append(st)
                    cur = 0
                for a in range(st + tile_hits, st + sz, tile_hits):
                    tile_bounds.append(a)
                tile_bounds.append(st + sz)
                continue
            if cur + sz > tile_hits:
                tile_bounds.append(st)
                cur = 0
            cur += sz
        if n and tile_bounds[-1] != n:
            tile_bounds.append(n)
        tile_bounds = np.asarray(tile_bounds, dtype=np.int64)
        n_tiles = len(tile_bounds) - 1
        tsize = np.diff(tile_bounds)
        tile_of_pos = np.repeat(np.arange(n_tiles), tsize)
        # degree sort inside each tile.  The kernels walk a list in groups of 4 steps, so a slice
        # costs ceil(max in / 4) + ceil(max out / 4) groups: hits are binned by ceil(in / 4) first
        # (in-degrees inside one bin cost the same) and sorted by out-degree inside the bins, which
        # are 4x longer than single-degree runs - 8 % fewer groups than a plain (in, out) sort
        order = np.lexsort((-deg_out[base], -((deg_in[base] + 3) // 4), tile_of_pos))
        old_of_rank = base[order]                             # tile-major, degree-sorted
        tpad = (tsize + SLICE - 1) // SLICE * SLICE
        tpad_off = np.zeros(n_tiles + 1, dtype=np.int64)
        np.cumsum(tpad, out=tpad_off[1:])
        n_pad = int(tpad_off[-1])
        rank_in_tile = np.arange(n, dtype=np.int64) - np.repeat(tile_bounds[:-1], tsize)
        new_of_rank = np.repeat(tpad_off[:-1], tsize) + rank_in_tile
        inv = np.empty(n + 1, dtype=np.int64)                 # old id -> new (padded) id
        inv[old_of_rank] = new_of_rank
        inv[n] = n_pad                                        # padded segment -> NULL hit
        perm = np.full(n_pad, -1, dtype=np.int64)             # new id -> old id (-1 = dummy)
        perm[new_of_rank] = old_of_rank

        self.n_hits, self.n_pad, self.n_segments = n, n_pad, E
        self.n_features = batch.n_features
        self.n_slices = n_pad // SLICE
        self.n_tiles = n_tiles
        src_new = inv[np.where(ok, src, n)]
        dst_new = inv[np.where(ok, dst, n)]
        vs_new, vd_new = src_new[ok], dst_new[ok]

        # -- per-tile windows -------------------------------------------------------------
        tb = tpad_off                                         # tile bounds in new ids
        in_lo_n, in_hi_n = _node_minmax(vd_new, vs_new, n_pad)
        out_lo_n, out_hi_n = _node_minmax(vs_new, vd_new, n_pad)
        if n_tiles:
            in_lo, in_hi = _range_minmax(in_lo_n, in_hi_n, tb)
            out_lo, out_hi = _range_minmax(out_lo_n, out_hi_n, tb)
        else:
            in_lo = in_hi = out_lo = out_hi = np.zeros(0, np.int64)
        in_cnt = np.where(in_hi >= 0, in_hi - in_lo + 1, 0)
        out_cnt = np.where(out_hi >= 0, out_hi - out_lo + 1, 0)
        in_lo = np.where(in_cnt > 0, in_lo, 0)
        out_lo = np.where(out_cnt > 0, out_lo, 0)
        lds_mode = (in_cnt + out_cnt + 2) <= int(limits["iter_records"])
        tile_of_new = np.repeat(np.arange(n_tiles), tpad)
        slice_tile = tile_of_new[::SLICE] if n_pad else np.zeros(0, np.int64)

        def lists(key_new, other_new, lo, cnt):
            t = tile_of_new[key_new]
            rel = np.where(lds_mode[t], other_new - lo[t], other_new)
            null = np.where(lds_mode[slice_tile], cnt[slice_tile], n_pad)
            off, nbr = _sell(key_new, rel.astype(np.int32), n_pad, null)
            return (off, nbr) + _pack16(off, nbr, null)

        in_off, in_nbr, in_off16, in_nbr16 = lists(vd_new, vs_new, in_lo, in_cnt)
        out_off, out_nbr, out_off16, out_nbr16 = lists(vs_new, vd_new, out_lo, out_cnt)
        tiles = np.zeros((n_tiles, TILE_DESC), dtype=np.int32)
        tiles[:, 0] = tb[:-1] // SLICE
        tiles[:, 1] = tb[1:] // SLICE
        tiles[:, 2], tiles[:, 3] = in_lo, in_cnt
        tiles[:, 4], tiles[:, 5] = out_lo, out_cnt
        tiles[:, 6] = lds_mode
        # per-phase schedules of the phase-split kernel: which slice a wavefront (16 per tile)
        # takes in round r.  Slices are dealt in snake order of decreasing cost, so the waves of
        # a tile reach each phase barrier together and a partial last round holds the lightest
        # slices.  sched[base + 16*r + wave] = slice id or -1; base = tiles[:, 7].
        nsl = (tiles[:, 1] - tiles[:, 0]).astype(np.int64)
        rounds = (nsl + 15) // 16
        sbase = np.zeros(n_tiles + 1, dtype=np.int64)
        np.cumsum(rounds * 16, out=sbase[1:])
        tiles[:, 7] = sbase[:-1]
        tile_of_slice = np.repeat(np.arange(n_tiles), nsl)

        def schedule(cost):
            order = np.lexsort((-cost, tile_of_slice))           # tile-major, heaviest first
            rank = np.arange(order.shape[0]) - np.repeat(np.cumsum(nsl) - nsl, nsl)
            r, c = rank // 16, rank % 16
            wave = np.where(r % 2 == 0, c, 15 - c)
            out = np.full(int(sbase[-1]) + 16, -1, dtype=np.int32)
            out[sbase[tile_of_slice[order]] + r * 16 + wave] = order
            return out

        steps_in = np.diff(in_off.astype(np.int64)) // SLICE
        steps_out = np.diff(out_off.astype(np.int64)) // SLICE

        def schedule_two_phase(cost_a, cost_b):
            """One slice -> wave assignment that balances BOTH phases (k_iter2 keeps a slice's
            partial sum in the registers of the wave that owns it, and a barrier ends each phase,
            so a tile costs max_w A_w + max_w B_w).  Slices are taken round by round in (A, B)
            order - the 16 slices of a round have near-equal in-phase cost thanks to the binned
            sort - and inside a round the heaviest goes to the free wave that raises
            max A + max B least.  Vectorised over tiles; 6 % less barrier wait than the snake."""
            out = np.full(int(sbase[-1]) + 16, -1, dtype=np.int32)
            if not len(nsl) or nsl.max(initial=0) == 0:
                return out
            first = np.cumsum(nsl) - nsl
            by_ab = np.lexsort((-cost_b, -cost_a, tile_of_slice))            # tile-major ranks
            rnd = (np.arange(by_ab.shape[0]) - np.repeat(first, nsl)) // 16
            # inside each (tile, round): heaviest total first
            by_tot = by_ab[np.lexsort((-(cost_a + cost_b)[by_ab], rnd, tile_of_slice[by_ab]))]
            A = np.zeros((n_tiles, 16))
            B = np.zeros((n_tiles, 16))
            for r in range(int(rounds.max())):
                used = np.zeros((n_tiles, 16), dtype=bool)
                for j in range(16):
                    act = np.flatnonzero(nsl > r * 16 + j)
                    if not act.size:
                        break
                    sl = by_tot[first[act] + r * 16 + j]
                    a, b = cost_a[sl][:, None], cost_b[sl][:, None]
                    Aa, Ba = A[act], B[act]
                    score = (np.maximum(Aa + a, Aa.max(1, keepdims=True)) +
                             np.maximum(Ba + b, Ba.max(1, keepdims=True)) + 1e-3 * (Aa + Ba))
                    score[used[act]] = np.inf
                    w = score.argmin(1)
                    out[sbase[act] + r * 16 + w] = sl
                    A[act, w] += a[:, 0]
                    B[act, w] += b[:, 0]
                    used[act, w] = True
            return out

        # cost in the kernel's own unit, groups of 4 list steps; the hit update + record stores of
        # a slice cost about 2.6 groups (measured ratio of tail to sweep time)
        groups_in, groups_out = (steps_in + 3) // 4, (steps_out + 3) // 4
        sched_a = schedule_two_phase(groups_in.astype(np.float64), groups_out + 2.6)
        # sched_b: out-phase-only balance (kept for kernels that decouple the phases)
        sched_b = schedule(groups_out + 3)

        # -- final edge pass: chunks of the caller's segment order --------------------------
        CH = int(limits["chunk_segments"])
        if E < 512 * CH:          # small batches: more, smaller chunks (same reasoning as the tiles)
            CH = min(CH, max(1024, (E + 511) // 512))
        cb = _chunk_bounds(np.where(ok, gid[np.where(ok, src, 0)] * (int(level.max(initial=0)) + 1)
                                    + level[np.where(ok, src, 0)], -1), ok, E, CH)
        n_chunks = len(cb) - 1
        big = np.iinfo(np.int64).max
        s_lo_e = np.where(ok, src_new, big)
        s_hi_e = np.where(ok, src_new, -1)
        d_lo_e = np.where(ok, dst_new, big)
        d_hi_e = np.where(ok, dst_new, -1)
        if n_chunks:
            s_lo, s_hi = _range_minmax(s_lo_e, s_hi_e, cb)
            d_lo, d_hi = _range_minmax(d_lo_e, d_hi_e, cb)
        else:
            s_lo = s_hi = d_lo = d_hi = np.zeros(0, np.int64)
        s_cnt = np.where(s_hi >= 0, s_hi - s_lo + 1, 0)
        d_cnt = np.where(d_hi >= 0, d_hi - d_lo + 1, 0)
        s_lo = np.where(s_cnt > 0, s_lo, 0)
        d_lo = np.where(d_cnt > 0, d_lo, 0)
        c_lds = (s_cnt + d_cnt + 2) <= int(limits["edge_records"])
        chunk_of_seg = np.repeat(np.arange(n_chunks), np.diff(cb))
        c = chunk_of_seg
        src_st = np.where(c_lds[c], np.where(ok, src_new - s_lo[c], s_cnt[c]), src_new)
        dst_st = np.where(c_lds[c], np.where(ok, dst_new - d_lo[c], d_cnt[c]), dst_new)
        chunks = np.zeros((n_chunks, CHUNK_DESC), dtype=np.int32)
        chunks[:, 0], chunks[:, 1] = cb[:-1], cb[1:]
        chunks[:, 2], chunks[:, 3] = s_lo, s_cnt
        chunks[:, 4], chunks[:, 5] = d_lo, d_cnt
        chunks[:, 6] = c_lds
        self.n_chunks = n_chunks
        # LDS actually needed by the launches (records / rows incl. the two NULL slots)
        self.iter_lds_records = int((in_cnt + out_cnt + 2)[lds_mode].max(initial=0))
        self.edge_lds_rows = int((s_cnt + d_cnt + 2)[c_lds].max(initial=0))
        self.n_lds_tiles = int(lds_mode.sum())
        self.iter_lds_in = int(in_cnt[lds_mode].max(initial=0))
        self.iter_lds_out = int(out_cnt[lds_mode].max(initial=0))
        self.tile_hits_max = int(tpad.max(initial=0))
        self.max_list_steps = int(max(np.diff(in_off).max(initial=0), np.diff(out_off).max(initial=0))) // SLICE

        # + 64 zero rows: the first-iteration kernel DMAs X windows in whole 256-byte pieces
        Xp = np.zeros((n_pad + 1 + 64, X.shape[1]), dtype=np.float32)
        Xp[new_of_rank] = X[old_of_rank]
        t = torch.from_numpy
        self.X = t(Xp)
        self.x_absmax = t(np.abs(Xp).max(axis=0).astype(np.float32)) if n else t(np.zeros(X.shape[1], np.float32))
        self.src, self.dst = t(src_st.astype(np.int32)), t(dst_st.astype(np.int32))
        # LDS-mode chunks: both window-relative endpoints in one word (dst << 16 | src)
        sd = np.where(c_lds[c], (dst_st.astype(np.int64) << 16) | (src_st.astype(np.int64) & 0xFFFF), 0) if E else np.zeros(0, np.int64)
        self.sd16 = t(sd.astype(np.uint32).view(np.int32))
        self.in_off, self.in_nbr = t(in_off), t(in_nbr)
        self.out_off, self.out_nbr = t(out_off), t(out_nbr)
        self.in_off16, self.in_nbr16 = t(in_off16), t(in_nbr16)
        self.out_off16, self.out_nbr16 = t(out_off16), t(out_nbr16)
        self.tiles, self.chunks = t(tiles.reshape(-1)), t(chunks.reshape(-1))
        self.sched_a, self.sched_b = t(sched_a), t(sched_b)
        self.perm = t(perm.astype(np.int32))
        nv = max(1, int(ok.sum()))
        self.padding = (int(in_off[-1]) + int(out_off[-1])) / (2 * nv) - 1.0
        self.lds_tile_fraction = float(lds_mode.mean()) if n_tiles else 1.0
        self.lds_chunk_fraction = float(c_lds.mean()) if n_chunks else 1.0
        # host copies for tests
        self.src_abs, self.dst_abs = src_new, dst_new
        self.level = level

    _TENSORS = ("X", "src", "dst", "in_off", "in_nbr", "out_off", "out_nbr", "in_off16", "in_nbr16",
                "out_off16", "out_nbr16", "tiles", "chunks", "sched_a", "sched_b", "perm",
                "x_absmax", "sd16")

    def to(self, device):
        for k in self._TENSORS:
            setattr(self, k, getattr(self, k).to(device))
        # cached C structs hold raw device pointers of the tensors just replaced
        self._struct = None
        self._ws_need = None
        self._xp = None
        return self

    def with_features(self, X):
        """The same plan for OTHER hit features of the same graphs (a re-calibrated or re-scaled read-out of the same
        segments): every structure array is shared, only the renumbered feature rows and their per-feature range
        (the exp-product bound's input) are made anew - a handful of small torch ops instead of a plan build (5.7 ms
        at c3 x 256).  X: [n_hits, F] in the caller's hit numbering, on this plan's device."""
        import copy
        X = torch.as_tensor(X).to(torch.float32)
        if tuple(X.shape) != (self.n_hits, self.X.shape[1]):
            raise ValueError("expected X of shape (%d, %d)" % (self.n_hits, self.X.shape[1]))
        if X.device != self.X.device:
            raise ValueError("X is on %s, the plan on %s" % (X.device, self.X.device))
        new = copy.copy(self)
        Xp = torch.zeros_like(self.X)
        where = getattr(self, "_feature_rows", None)        # (structure: found once, shared by the copies)
        if where is None or where[0].device != X.device:
            perm = self.perm.to(torch.int64)                # padded id -> caller's hit id, -1 = dummy
            slots = torch.nonzero(perm >= 0).reshape(-1)
            where = self._feature_rows = (slots, perm[slots])
        Xp[where[0]] = X[where[1]]
        new.X = Xp
        new.x_absmax = Xp.abs().max(dim=0).values if self.n_hits else torch.zeros_like(self.x_absmax)
        new._struct = None               # cached C struct (raw pointers), exp-product decision: of the old features
        new._xp = None
        return new

    @property
    def device(self):
        return self.X.device

"""The execution plan of plan.py built by HIP kernels (csrc/plan_build.hip) where the batch lives.

Same plan as `plan.SellPlan` (numpy, the specification) and `plan_device.DeviceSellPlan` (torch
ops), array for array; what changes is the cost: a few dozen kernel launches and ONE host read-back
(the array sizes) instead of ~300 torch ops with a dozen synchronisations - milliseconds instead of
0.2 s for the 25.6 M-segment benchmark batch, so that a stream of never-repeated batches (inference
on real events) no longer pays hundreds of forwards per plan.

Batches outside the builder's static bounds (`sizes.status != 0`: a hit with >= 65536 segments,
more than n/16 + 1024 tiles, ...) and empty batches raise `PlanBuilderUnsupported`; the caller
(`HitGraphBatch.build_plan`) then uses the torch builder.
"""
import os

import numpy as np
import torch

from . import _lib
from .plan import SLICE, SellPlan


class PlanBuilderUnsupported(RuntimeError):
    pass


class HipSellPlan(SellPlan):
    def __init__(self, batch, limits, debug=False, graph_local=None):   # noqa: C901
        dev = batch.X.device
        n, E, G = batch.n_hits, batch.n_segments, batch.n_graphs
        if n <= 0 or E <= 0 or G <= 0:
            raise PlanBuilderUnsupported("empty batch")
        tile_hits = int(limits["tile_hits"])
        want = 512 if int(limits["iter_records"]) > 0 else 256          # plan.py: small batches
        if n < want * tile_hits:
            tile_hits = max(64, ((n + want - 1) // want + SLICE - 1) // SLICE * SLICE)
        CH = int(limits["chunk_segments"])
        if E < 512 * CH:
            CH = min(CH, max(1024, (E + 511) // 512))
        F = batch.n_features
        if F > 16 or n >= 2 ** 30 or E >= 2 ** 30 or G >= 2 ** 24:
            raise PlanBuilderUnsupported("batch outside the builder's index ranges")
        src = batch.src.to(dev).to(torch.int32).contiguous()
        dst = batch.dst.to(dev).to(torch.int32).contiguous()
        X = batch.X.to(dev).to(torch.float32).contiguous()
        hp = np.ascontiguousarray(batch.hit_ptr, dtype=np.int64)
        sp = np.ascontiguousarray(batch.seg_ptr, dtype=np.int64)
        # the graph-local stage 1 (one workgroup per graph, LDS tables) when the batch names its graphs' segment
        # ranges and the largest graph fits; the kernels check the layout themselves (status 128 -> the global form)
        local = graph_local if graph_local is not None else os.environ.get("GNN_PLAN_GRAPH_LOCAL", "1") != "0"
        if graph_local is None and E < 16 * int(np.diff(sp).max()) and int(np.diff(sp).max()) > 8192:
            # one workgroup per graph: a graph-local build takes as long as its largest graph does (1 ms for a 100 k-segment
            # detector graph), the global form as long as all segments together (0.75 ms + 0.02 ms per such graph):
            # 1.29 / 0.76 ms at 4 detector graphs, 0.98 / 0.86 at 8, 0.92 / 0.92 at 16, 0.95 / 1.07 at 24, 1.03 / 1.41 at 32
            local = False
        local = bool(local and len(sp) == G + 1 and int(sp[0]) == 0 and int(sp[-1]) == E
                     and int(np.diff(hp).max()) <= _lib.PLAN_GRAPH_CAP_HITS)
        ptrs = torch.from_numpy(np.concatenate([hp, sp]) if local else hp).to(dev)       # one upload
        hit_ptr = ptrs[:G + 1]
        ws = torch.empty(_lib.plan_build_workspace_bytes(n, E, CH), dtype=torch.uint8, device=dev)
        sz = None
        if local:
            sz = _lib.plan_build_sizes(src, dst, hit_ptr, n, E, G, tile_hits, int(limits["iter_records"]), CH,
                                       int(limits["edge_records"]), ws, seg_ptr=ptrs[G + 1:],
                                       max_graph_hits=int(np.diff(hp).max()), max_graph_segments=int(np.diff(sp).max()))
            if sz.status & _lib.PLAN_STATUS_FAST_MISS:
                local, sz = False, None
        if sz is None:
            sz = _lib.plan_build_sizes(src, dst, hit_ptr, n, E, G, tile_hits, int(limits["iter_records"]), CH,
                                       int(limits["edge_records"]), ws)
        self.graph_local = local
        self.list_mode = int(sz.list_mode) if local else 0    # 1: neighbour lists built per tile in LDS
        if sz.status & 64:      # ST_ENDPOINT: the numpy builder's ValueError, not a fallback
            raise ValueError("segment endpoint out of range, or a segment with exactly one padded end "
                             "(a padded segment must have src = dst = -1)")
        if sz.status:
            raise PlanBuilderUnsupported("plan builder status %d" % sz.status)
        if max(sz.in_total, sz.out_total) + 64 >= 2 ** 31:
            raise ValueError("SELL list exceeds int32 index range")
        i32 = lambda k: torch.empty(int(k), dtype=torch.int32, device=dev)     # noqa: E731
        a = {
            "X": torch.empty((sz.n_pad + 1 + 64, F), dtype=torch.float32, device=dev),
            "x_absmax": torch.empty(F, dtype=torch.float32, device=dev),
            "src": i32(E), "dst": i32(E), "sd16": i32(E),
            "in_off": i32(sz.n_slices + 1), "in_nbr": i32(sz.in_total + 4 * SLICE),
            "out_off": i32(sz.n_slices + 1), "out_nbr": i32(sz.out_total + 4 * SLICE),
            "in_off16": i32(sz.n_slices + 1), "in_nbr16": i32(sz.in16_words + 64),
            "out_off16": i32(sz.n_slices + 1), "out_nbr16": i32(sz.out16_words + 64),
            "tiles": i32(8 * sz.n_tiles), "chunks": i32(8 * sz.n_chunks),
            "sched_a": i32(sz.n_sched), "sched_b": i32(sz.n_sched), "perm": i32(sz.n_pad),
        }
        if debug:
            a["src_abs"], a["dst_abs"], a["level"] = i32(E), i32(E), i32(n)
        _lib.plan_build_fill(X, src, dst, n, E, CH, sz, ws, a)
        for k in self._TENSORS:
            setattr(self, k, a[k])
        self.n_hits, self.n_pad, self.n_segments = n, int(sz.n_pad), E
        self.n_features = F
        self.n_slices, self.n_tiles, self.n_chunks = int(sz.n_slices), int(sz.n_tiles), int(sz.n_chunks)
        self.iter_lds_records, self.edge_lds_rows = int(sz.iter_lds_records), int(sz.edge_lds_rows)
        self.n_lds_tiles = int(sz.n_lds_tiles)
        self.iter_lds_in, self.iter_lds_out = int(sz.iter_lds_in), int(sz.iter_lds_out)
        self.tile_hits_max, self.max_list_steps = int(sz.tile_hits_max), int(sz.max_list_steps)
        nv = max(1, int(sz.n_valid))
        self.padding = (int(sz.in_total) + int(sz.out_total)) / (2 * nv) - 1.0
        self.lds_tile_fraction = float(np.float64(sz.n_lds_tiles) / np.float64(sz.n_tiles)) if sz.n_tiles else 1.0
        self.lds_chunk_fraction = float(np.float64(sz.n_lds_chunks) / np.float64(sz.n_chunks)) if sz.n_chunks else 1.0
        self._dbg = (a.get("src_abs"), a.get("dst_abs"), a.get("level"))

    # host copies for tests (debug=True)
    src_abs = property(lambda self: self._dbg[0].cpu().numpy().astype(np.int64))
    dst_abs = property(lambda self: self._dbg[1].cpu().numpy().astype(np.int64))
    level = property(lambda self: self._dbg[2].cpu().numpy().astype(np.int32))

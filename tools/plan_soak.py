"""Random batches through the HIP plan builder's three stage-1 forms - graph-local with per-tile lists, graph-local with
scattered pairs, global - against the numpy specification plan.SellPlan, EVERY array and scalar equal.  What the fixed
shapes of tests/test_plan_hip.py do not reach: random graph counts and sizes (1 hit ... a few thousand), layer counts,
skip-layer segments (level = longest walk, not the layer), multi-edges, graphs without segments, now and then a graph
with cycles (the layout check sends the batch to the global form), shuffled segment
order, padded segments at random places and in blocks, random tile sizes, both kernel shape families.
usage: python tools/plan_soak.py [trials] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from gnn_fpga_amd import HitGraphBatch, synth, _lib
from gnn_fpga_amd.plan import SellPlan
from gnn_fpga_amd.plan_hip import HipSellPlan
from test_plan import _same_plan

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def random_dag(n, e, F, L, rng, skip):
    """hits on L layers, segments from a lower to a higher layer (adjacent ones unless `skip`), grouped by start layer"""
    layer = rng.integers(0, L, n)
    X = rng.uniform(-1, 1, (n, F)).astype(np.float32)
    by = [np.flatnonzero(layer == l) for l in range(L)]
    src, dst = [], []
    for _ in range(e):
        a = int(rng.integers(0, L - 1))
        b = a + 1 if not skip else int(rng.integers(a + 1, L))
        if len(by[a]) == 0 or len(by[b]) == 0:
            continue
        src.append((a, int(rng.choice(by[a]))))
        dst.append(int(rng.choice(by[b])))
    order = np.argsort([s[0] for s in src], kind="stable") if src else np.zeros(0, np.int64)
    s = np.asarray([src[i][1] for i in order], np.int32)
    d = np.asarray([dst[i] for i in order], np.int32)
    return synth.HitGraph(X, s, d, np.zeros(len(s), np.float32))


modes = {"tile": 0, "scatter": 0, "global": 0}
t0 = time.time()
for t in range(trials):
    F, D = (3, 8) if rng.random() < 0.7 else ((11, 8) if rng.random() < 0.5 else (3, 64))
    G = int(rng.choice([1, 2, 3, 5, 9, 17, 40]))
    graphs = []
    for g in range(G):
        n = int(rng.choice([1, 2, 7, 40, 300, 1500, 4000]))
        L = int(rng.integers(2, 13))
        e = 0 if rng.random() < 0.1 else int(rng.integers(0, 12 * n + 1))
        if n < 2:
            e = 0
        cyclic = n >= 2 and e > 0 and rng.random() < 0.05     # any segments at all: cycles, self loops -> status 128
        gr = synth.HitGraph(rng.uniform(-1, 1, (n, F)).astype(np.float32), rng.integers(0, n, e).astype(np.int32),
                            rng.integers(0, n, e).astype(np.int32), np.zeros(e, np.float32)) if cyclic else \
            random_dag(n, e, F, min(L, max(2, n)), rng, skip=rng.random() < 0.3) if n >= 2 else \
            synth.HitGraph(rng.uniform(-1, 1, (n, F)).astype(np.float32), np.zeros(0, np.int32), np.zeros(0, np.int32),
                           np.zeros(0, np.float32))
        if len(gr.src) and rng.random() < 0.15:          # shuffled segment order
            o = rng.permutation(len(gr.src))
            gr = synth.HitGraph(gr.X, gr.src[o], gr.dst[o], gr.y[o])
        graphs.append(gr)
    b = HitGraphBatch.from_graphs(graphs, pad_segments=bool(rng.random() < 0.2))
    if b.n_segments == 0:
        continue
    src, dst = b.src.numpy().copy(), b.dst.numpy().copy()
    if rng.random() < 0.3:                               # pads at random places / in a block
        m = rng.random(len(src)) < 0.1
        if rng.random() < 0.5:
            a = int(rng.integers(0, len(src)))
            m[a:a + int(rng.integers(1, 300))] = True
        src[m] = -1
        dst[m] = -1
    if (src >= 0).sum() == 0:
        continue
    b = HitGraphBatch(b.X.numpy(), src, dst, hit_ptr=b.hit_ptr, seg_ptr=b.seg_ptr)
    lim = _lib.plan_limits(F, D)
    if rng.random() < 0.4:
        lim["tile_hits"] = int(rng.choice([64, 256, 1280])) if D == 8 else int(rng.choice([64, 256]))
    if rng.random() < 0.15:
        lim.update({"iter_records": 0, "edge_records": 0})
    host = SellPlan(b, lim)
    bd = b.cuda()
    dev = HipSellPlan(bd, lim, debug=True, graph_local=True)
    _same_plan(host, dev)
    modes["tile" if dev.graph_local and dev.list_mode else ("scatter" if dev.graph_local else "global")] += 1
    if dev.graph_local:
        os.environ["GNN_PLAN_SCATTER_LISTS"] = "1"
        scat = HipSellPlan(bd, lim, debug=True, graph_local=True)
        del os.environ["GNN_PLAN_SCATTER_LISTS"]
        assert scat.graph_local and scat.list_mode == 0
        _same_plan(host, scat)
        glob = HipSellPlan(bd, lim, debug=True, graph_local=False)
        _same_plan(host, glob)
    again = HipSellPlan(bd, lim, debug=True, graph_local=True)      # the same arrays in a second build
    _same_plan(dev, again)
print("plan_soak: %d random batches, every plan array of every form equal to plan.py; default form chosen: %s; %.0f s"
      % (trials, modes, time.time() - t0))

"""The HIP plan builder (csrc/plan_build.hip, gnn_plan_build_sizes / gnn_plan_build_fill) against the
numpy specification plan.SellPlan: EVERY array and scalar equal (integer work: bit-exact), on layered
detector batches, ragged / tiny graphs, padded segments, forced global-gather mode, wide hidden
layers, one graph whose levels exceed a tile, muon-size graphs, graphs with cycles / self loops
(the 64-sweep cap of the level relaxation), shuffled segment order (narrow and wide hit ranges per
segment block), and BASELINE's c3 graph."""
import numpy as np
import pytest
import torch

from gnn_fpga_amd import HitGraphBatch, synth
from test_plan import _same_plan

pytestmark = pytest.mark.gpu


def _random_graph(n, e, F, seed):
    rng = np.random.default_rng(seed)
    X = rng.uniform(-1, 1, (n, F)).astype(np.float32)
    return synth.HitGraph(X, rng.integers(0, n, e).astype(np.int32), rng.integers(0, n, e).astype(np.int32),
                          np.zeros(e, np.float32))


CASES = ["layered", "ragged", "padded", "global", "wide", "one_graph_big_levels", "muon", "cyclic",
         "shuffled", "big_shuffled", "c3", "many_c3", "leading_pads", "tiny_tiles", "many_muon", "mu200_size"]


@pytest.mark.parametrize("case", CASES)
def test_hip_builder_builds_the_same_plan(hip, case, monkeypatch):
    from gnn_fpga_amd.plan import SellPlan
    from gnn_fpga_amd.plan_hip import HipSellPlan
    F, D, lim_over = 3, 8, {}
    pads = None
    if case == "layered":
        graphs = [synth.layered_graph(700, 4000, 3, seed=s) for s in range(5)]
    elif case == "ragged":
        graphs = [synth.layered_graph(n, e, 3, n_layers=L, seed=s)
                  for s, (n, e, L) in enumerate([(40, 90, 10), (3, 2, 2), (300, 2500, 10), (17, 16, 3), (2, 1, 2)])]
    elif case == "padded":
        graphs = [synth.layered_graph(200, 900, 3, seed=s) for s in range(3)]
        pads = slice(0, None, 7)
    elif case == "leading_pads":
        graphs = [synth.layered_graph(300, 2000, 3, seed=s) for s in range(2)]
        pads = slice(0, 37)
    elif case == "global":
        graphs = [synth.layered_graph(700, 4000, 3, seed=s) for s in range(3)]
        lim_over = {"iter_records": 0, "edge_records": 0}
    elif case == "wide":
        F, D = 3, 64
        graphs = [synth.layered_graph(500, 3000, 3, seed=s) for s in range(4)]
    elif case == "one_graph_big_levels":
        graphs = [synth.layered_graph(20000, 60000, 3, n_layers=4, seed=3)]
    elif case == "muon":
        F = 11
        graphs = [synth.muon_graph(s) for s in range(700)]
    elif case == "many_muon":   # > 8192 (graph, level) units: the sequential tile cut (pb_cut_tiles), not the parallel one
        F = 11
        graphs = [synth.muon_graph(s) for s in range(2600)]
    elif case == "mu200_size":  # graphs between 16384 and 19456 hits: graph-local with 32-bit sort keys only
        graphs = [synth.layered_graph(18000, 70000, 3, seed=9), synth.layered_graph(15000, 58000, 3, seed=10)]
    elif case == "cyclic":
        graphs = [_random_graph(400, 3000, 3, 1), _random_graph(37, 90, 3, 2),
                  synth.layered_graph(600, 5000, 3, seed=3), _random_graph(5, 40, 3, 4)]
    elif case == "shuffled":
        g = synth.layered_graph(3000, 30000, 3, seed=5)
        o = np.random.default_rng(0).permutation(30000)
        graphs = [synth.HitGraph(g.X, g.src[o], g.dst[o], g.y[o]), synth.layered_graph(2000, 15000, 3, seed=6)]
    elif case == "big_shuffled":
        # a 16 k-segment block of pb_degrees spans > 16384 hits: its global-atomics path, next to a
        # graph whose blocks stay narrow (LDS counts), and a block with padded segments only
        g = synth.layered_graph(40000, 120000, 3, seed=7)
        o = np.random.default_rng(1).permutation(120000)
        graphs = [synth.HitGraph(g.X, g.src[o], g.dst[o], g.y[o]), synth.layered_graph(5000, 40000, 3, seed=8)]
        pads = slice(120000 + 16384, 120000 + 2 * 16384)
    elif case == "c3":
        graphs = [synth.layered_graph(10000, 100000, 3, seed=0)]
    elif case == "many_c3":
        graphs = [synth.layered_graph(10000, 100000, 3, seed=s) for s in range(12)]
    else:   # tiny_tiles: force the smallest tiles (many tiles, many units per workgroup chunk)
        graphs = [synth.layered_graph(900, 5000, 3, n_layers=30, seed=s) for s in range(6)]
        lim_over = {"tile_hits": 64}
    b = HitGraphBatch.from_graphs(graphs)
    if pads is not None:     # zero-padded segments (src = dst = -1)
        src, dst = b.src.numpy().copy(), b.dst.numpy().copy()
        src[pads] = -1
        dst[pads] = -1
        b = HitGraphBatch(b.X.numpy(), src, dst, hit_ptr=b.hit_ptr, seg_ptr=b.seg_ptr)
    lim = hip.plan_limits(F, D)
    lim.update(lim_over)
    host = SellPlan(b, lim)
    # (a batch of fewer than sixteen graphs the size of its largest takes the global form by default, unless they are
    # small: ask for the graph-local one)
    dev = HipSellPlan(b.cuda(), lim, debug=True, graph_local=True)
    assert dev.X.is_cuda
    _same_plan(host, dev)
    # which stage 1 built it: the graph-local form (one workgroup per graph, LDS tables) wherever a batch's graphs
    # fit its tables and its layout checks hold; random graphs have cycles (levels above the 64-sweep cap -> status
    # 128 -> the global form), 20 k / 40 k-hit graphs exceed the tables
    assert dev.graph_local == (case not in ("one_graph_big_levels", "cyclic", "big_shuffled")), case
    if dev.graph_local:     # and the global form of the same batch, still array for array
        glob = HipSellPlan(b.cuda(), lim, debug=True, graph_local=False)
        assert not glob.graph_local
        _same_plan(host, glob)
        # neighbour lists: per tile in LDS when the segments of a tile's lists lie together (the reference's layer-pair
        # order), by scattered pairs + a sort per list otherwise (a graph's segments shuffled) - and on request
        # (one c3 graph: 64-hit tiles cut every level into 16 - its layer-pair block would be read 16 times)
        # (likewise two mu200-size graphs: 80-hit tiles)
        assert dev.list_mode == (0 if case in ("shuffled", "c3", "mu200_size") else 1), (case, dev.list_mode)
        if dev.list_mode:
            monkeypatch.setenv("GNN_PLAN_SCATTER_LISTS", "1")
            scat = HipSellPlan(b.cuda(), lim, debug=True, graph_local=True)
            monkeypatch.delenv("GNN_PLAN_SCATTER_LISTS")
            assert scat.graph_local and scat.list_mode == 0
            _same_plan(host, scat)


def test_graph_local_builder_is_chosen_where_it_pays(hip):
    """One workgroup per graph: the graph-local stage 1 takes as long as its LARGEST graph does, the global form as long
    as all segments together - the default takes the graph-local form from sixteen graphs of the largest one's size on,
    and always for small graphs (muon events)."""
    from gnn_fpga_amd.plan_hip import HipSellPlan
    lim = hip.plan_limits(3, 8)
    few = HitGraphBatch.from_graphs([synth.layered_graph(3000, 30000, 3, seed=s) for s in range(5)]).cuda()
    many = HitGraphBatch.from_graphs([synth.layered_graph(3000, 30000, 3, seed=s) for s in range(16)]).cuda()
    small = HitGraphBatch.from_graphs([synth.layered_graph(300, 2000, 3, seed=s) for s in range(3)]).cuda()
    assert not HipSellPlan(few, lim).graph_local
    assert HipSellPlan(many, lim).graph_local
    assert HipSellPlan(small, lim).graph_local
    assert HipSellPlan(few, lim, graph_local=True).graph_local


def test_graph_local_builder_checks_the_layout_it_is_told(hip):
    """seg_ptr / hit_ptr that do not describe the batch (a segment joining hits of another graph's range, segment
    ranges that leave segments out) must not change the plan: the graph-local kernels report status 128 and the
    global form builds it."""
    from gnn_fpga_amd.plan import SellPlan
    from gnn_fpga_amd.plan_hip import HipSellPlan
    graphs = [synth.layered_graph(600, 4000, 3, seed=s) for s in range(4)]
    b = HitGraphBatch.from_graphs(graphs)
    lim = hip.plan_limits(3, 8)
    # (a) the same arrays described as ONE graph's hits but four segment ranges: still consistent -> local
    # (b) a segment of graph 1 rewired to a hit of graph 2
    src, dst = b.src.numpy().copy(), b.dst.numpy().copy()
    j = int(b.seg_ptr[1]) + 5
    dst[j] = int(b.hit_ptr[2]) + 3
    crossed = HitGraphBatch(b.X.numpy(), src, dst, hit_ptr=b.hit_ptr, seg_ptr=b.seg_ptr)
    dev = HipSellPlan(crossed.cuda(), lim, debug=True)
    assert not dev.graph_local
    _same_plan(SellPlan(crossed, lim), dev)
    # (c) segment ranges shifted by one: graph 0 claims a segment of graph 1
    sp = b.seg_ptr.copy()
    sp[1] += 1
    shifted = HitGraphBatch(b.X.numpy(), b.src.numpy(), b.dst.numpy(), hit_ptr=b.hit_ptr, seg_ptr=sp)
    dev = HipSellPlan(shifted.cuda(), lim, debug=True)
    assert not dev.graph_local
    _same_plan(SellPlan(shifted, lim), dev)


def test_hip_builder_is_the_default_and_declines_what_it_cannot_hold(hip, monkeypatch):
    from gnn_fpga_amd.plan_device import DeviceSellPlan
    from gnn_fpga_amd.plan_hip import HipSellPlan, PlanBuilderUnsupported
    g = [synth.layered_graph(800, 5000, 3, seed=s) for s in range(3)]
    assert isinstance(HitGraphBatch.from_graphs(g).cuda().build_plan(8), HipSellPlan)
    monkeypatch.setenv("GNN_PLAN_BUILDER", "torch")
    assert isinstance(HitGraphBatch.from_graphs(g).cuda().build_plan(8), DeviceSellPlan)
    monkeypatch.delenv("GNN_PLAN_BUILDER")
    empty = HitGraphBatch.from_graphs([synth.layered_graph(50, 0, 3, seed=1)]).cuda()
    with pytest.raises(PlanBuilderUnsupported):
        HipSellPlan(empty, hip.plan_limits(3, 8))
    assert isinstance(empty.build_plan(8), DeviceSellPlan)          # falls back by itself
    # a hub with >= 65536 segments is outside the 16-bit degree keys of the hit sorts
    n, e = 70000, 70000
    X = np.zeros((n, 3), np.float32)
    hub = synth.HitGraph(X, np.arange(1, e + 1, dtype=np.int32) % n, np.zeros(e, np.int32), np.zeros(e, np.float32))
    with pytest.raises(PlanBuilderUnsupported):
        HipSellPlan(HitGraphBatch.from_graphs([hub]).cuda(), hip.plan_limits(3, 8))


def test_hip_builder_refuses_malformed_endpoints(hip):
    """A malformed index-form batch on the device (an endpoint >= n_hits, a segment with exactly one
    padded end) must come back as the ValueError the numpy builder's path gives - not as out-of-range
    global atomics / stores of the builder kernels (status bit ST_ENDPOINT, set by pb_degrees; every
    kernel that indexes per-hit arrays before the status read-back skips such segments)."""
    from gnn_fpga_amd.plan_hip import HipSellPlan
    g = synth.layered_graph(5000, 40000, 3, seed=2)
    for bad_src, bad_dst in (((123, 5000 + 7), None), (None, (40, 2 ** 30)), ((77, -1), None), (None, (9, -5))):
        b = HitGraphBatch.from_graphs([g]).cuda()
        src, dst = b.src.clone(), b.dst.clone()            # corrupt the DEVICE arrays: no host check sees them
        if bad_src:
            src[bad_src[0]] = bad_src[1]
        if bad_dst:
            dst[bad_dst[0]] = bad_dst[1]
        b.src, b.dst = src, dst
        with pytest.raises(ValueError):
            HipSellPlan(b, hip.plan_limits(3, 8))
    ok = HitGraphBatch.from_graphs([g]).cuda()
    HipSellPlan(ok, hip.plan_limits(3, 8))                  # the device is still healthy


def test_forward_on_a_hip_built_plan_matches_the_oracle(hip):
    from gnn_fpga_amd.model import SegmentClassifier
    from gnn_fpga_amd.plan_hip import HipSellPlan
    from oracle import index_c
    torch.manual_seed(0)
    graphs = [synth.layered_graph(3000, 30000, 3, seed=s) for s in range(6)]
    m = SegmentClassifier(input_dim=3, hidden_dim=8, n_iters=3).cuda().eval()
    m.use_events = False
    b = HitGraphBatch.from_graphs(graphs).cuda()
    with torch.no_grad():
        e = m(b)
    assert isinstance(b.plan, HipSellPlan)
    params = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    for g, eg in zip(graphs, b.split_scores(e.cpu().numpy())):
        assert np.abs(eg - index_c.segment_classifier(g.X, g.src, g.dst, params, 3)).max() < 1e-5

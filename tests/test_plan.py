"""Host-side checks of the execution plan (gnn-fpga_amd/plan.py): a numpy emulation of what the
fused kernels do with the plan arrays (tiles, windows, SELL-16 lists, chunks) must reproduce
the oracle.  Runs without a GPU (the limits come from the library's host-side query)."""
import numpy as np
import torch
import pytest

from golden_util import Fixture
from gnn_fpga_amd import HitGraphBatch, _lib, synth
from gnn_fpga_amd.plan import SLICE, SellPlan
from oracle import index_c


def emulate(plan, params, n_iters):
    """float64 emulation of csrc/sell_pipeline.hip driven by the plan arrays only."""
    p = {k: np.asarray(v, np.float64) for k, v in params.items()}
    D, F = p["input_network.0.weight"].shape
    C = D + F
    W1, b1 = p["edge_network.network.0.weight"], p["edge_network.network.0.bias"]
    W2, b2 = p["edge_network.network.2.weight"][0], p["edge_network.network.2.bias"][0]
    W3, b3 = p["node_network.network.0.weight"], p["node_network.network.0.bias"]
    W4, b4 = p["node_network.network.2.weight"], p["node_network.network.2.bias"]
    X = plan.X.numpy().astype(np.float64)
    Np = plan.n_pad
    sig = lambda z: 1 / (1 + np.exp(-z))

    def records(Hn):           # rows 0..Np-1 real/dummy, row Np = NULL
        H = np.concatenate([Hn, X[:Np]], axis=1)
        P = np.vstack([H @ W1[:, :C].T + b1, b1[None]])
        R = np.vstack([H @ W3[:, :C].T, np.zeros((1, D))])
        Q = np.vstack([H @ W1[:, C:].T, np.zeros((1, D))])
        S = np.vstack([H @ W3[:, C:2 * C].T, np.zeros((1, D))])
        U = H @ W3[:, 2 * C:].T + b3
        return P, R, Q, S, U

    P, R, Q, S, U = records(np.tanh(X[:Np] @ p["input_network.0.weight"].T +
                                    p["input_network.0.bias"]))
    tiles = plan.tiles.numpy().reshape(-1, 8)
    in_off, in_nbr = plan.in_off.numpy(), plan.in_nbr.numpy()
    out_off, out_nbr = plan.out_off.numpy(), plan.out_nbr.numpy()
    for _ in range(n_iters):
        acc = U.copy()
        for (s0, s1, in_lo, in_cnt, out_lo, out_cnt, mode, _z) in tiles:
            for sl in range(s0, s1):
                for i in range(SLICE):
                    n = sl * SLICE + i
                    for off, nbr, A, B, own, lo, cnt in (
                            (in_off, in_nbr, P, R, Q[n], in_lo, in_cnt),
                            (out_off, out_nbr, Q, S, P[n], out_lo, out_cnt)):
                        for k in range((off[sl + 1] - off[sl]) // SLICE):
                            ent = nbr[off[sl] + k * SLICE + i]
                            if mode:      # window-relative, NULL = cnt
                                assert 0 <= ent <= cnt
                                idx = Np if ent == cnt else lo + ent
                            else:
                                idx = ent
                            e = sig(W2 @ np.tanh(A[idx] + own) + b2)
                            acc[n] += e * B[idx]
        Hn = np.tanh(np.tanh(acc) @ W4.T + b4)
        P, R, Q, S, U = records(Hn)
    chunks = plan.chunks.numpy().reshape(-1, 8)
    src, dst = plan.src.numpy(), plan.dst.numpy()
    out = np.zeros(plan.n_segments)
    for (e0, e1, s_lo, s_cnt, d_lo, d_cnt, mode, _z) in chunks:
        for j in range(e0, e1):
            s, d = src[j], dst[j]
            if mode:
                s = Np if s == s_cnt else s_lo + s
                d = Np if d == d_cnt else d_lo + d
            out[j] = sig(W2 @ np.tanh(P[s] + Q[d]) + b2)
    return out


LIMITS = [None,                                                     # the library's real budgets
          dict(tile_hits=64, iter_records=40, chunk_segments=50, edge_records=60),   # mixed modes
          dict(tile_hits=32, iter_records=0, chunk_segments=64, edge_records=0)]     # all global


@pytest.mark.parametrize("name", ["sector_s0", "muon_s1", "ragged_isolated_s7",
                                  "ragged_one_segment", "toy2d_s0"])
@pytest.mark.parametrize("lim", LIMITS)
def test_plan_emulation_matches_golden(name, lim):
    fx = Fixture(name)
    batch = HitGraphBatch.from_graphs([fx.graph])
    plan = SellPlan(batch, lim or _lib.plan_limits(fx.F, fx.D))
    e = emulate(plan, fx.effective_params(), fx.n_iters)
    assert np.abs(e - fx.scores).max() < 2e-6


def test_plan_on_padded_ragged_batch():
    rng = np.random.default_rng(3)
    graphs = [synth.layered_graph(int(rng.integers(20, 90)), int(rng.integers(10, 300)), 3,
                                  seed=50 + i) for i in range(5)]
    fx = Fixture("sector_s0")
    b = HitGraphBatch.from_graphs(graphs)
    # append padded segments (src = dst = -1), as a dense zero-padded batch produces them
    src = np.concatenate([b.src.numpy(), -np.ones(7, np.int32)])
    dst = np.concatenate([b.dst.numpy(), -np.ones(7, np.int32)])
    bp = HitGraphBatch(b.X.numpy(), src, dst, hit_ptr=b.hit_ptr)
    for lim in LIMITS[:2]:
        plan = SellPlan(bp, lim or _lib.plan_limits(3, 8))
        e = emulate(plan, fx.params, 2)
        ref = index_c.segment_classifier(b.X.numpy(), src, dst, fx.params, 2)
        assert np.abs(e - ref).max() < 2e-6


def test_plan_structure_at_c3_shape():
    """Layered 10k-hit graphs: every tile and (almost) every chunk runs in LDS mode, list
    padding stays small, levels recover the layers."""
    graphs = [synth.layered_graph(10000, 100000, 3, seed=s) for s in range(3)]
    b = HitGraphBatch.from_graphs(graphs)
    plan = SellPlan(b, _lib.plan_limits(3, 8))
    assert plan.n_pad % SLICE == 0 and plan.n_pad >= b.n_hits
    assert plan.level.max() == 9
    assert plan.lds_tile_fraction == 1.0
    assert plan.lds_chunk_fraction > 0.85
    assert plan.padding < 0.15
    perm = plan.perm.numpy()
    real = perm[perm >= 0]
    assert np.array_equal(np.sort(real), np.arange(b.n_hits))
    # endpoints in absolute padded ids reproduce the caller's segments
    assert np.array_equal(perm[plan.src_abs], b.src.numpy())
    assert np.array_equal(perm[plan.dst_abs], b.dst.numpy())


def test_plan_handles_cycles_and_empty():
    # a graph with a cycle: levels do not converge, plan must still be valid
    X = np.random.default_rng(0).uniform(-1, 1, (6, 3)).astype(np.float32)
    src = np.array([0, 1, 2, 3, 4, 0], np.int32)
    dst = np.array([1, 2, 0, 4, 5, 3], np.int32)
    fx = Fixture("sector_s0")
    b = HitGraphBatch(X, src, dst)
    plan = SellPlan(b, _lib.plan_limits(3, 8))
    e = emulate(plan, fx.params, 3)
    ref = index_c.segment_classifier(X, src, dst, fx.params, 3)
    assert np.abs(e - ref).max() < 2e-6
    empty = HitGraphBatch(X, np.zeros(0, np.int32), np.zeros(0, np.int32))
    pe = SellPlan(empty, _lib.plan_limits(3, 8))
    assert pe.n_segments == 0 and pe.n_chunks == 0 and emulate(pe, fx.params, 2).shape == (0,)


def test_packed16_lists_decode_to_the_sell_lists():
    """plan._pack16: word p of hit i of slice s = steps 2p | 2p+1 << 16, steps padded to 8 with
    the tile's NULL entry."""
    graphs = [synth.layered_graph(300, 1500, 3, seed=s) for s in range(3)]
    b = HitGraphBatch.from_graphs(graphs)
    plan = SellPlan(b, _lib.plan_limits(3, 8))
    assert plan.lds_tile_fraction == 1.0
    tiles = plan.tiles.numpy().reshape(-1, 8)
    for off, nbr, off16, nbr16, col in ((plan.in_off, plan.in_nbr, plan.in_off16, plan.in_nbr16, 3),
                                        (plan.out_off, plan.out_nbr, plan.out_off16, plan.out_nbr16, 5)):
        off, nbr, off16 = off.numpy(), nbr.numpy(), off16.numpy()
        w = nbr16.numpy().view(np.uint32)
        for (s0, s1, *_rest) in tiles:
            null = tiles[np.searchsorted(tiles[:, 0], s0, side="right") - 1][col]
            for sl in range(s0, s1):
                L = (off[sl + 1] - off[sl]) // SLICE
                n8 = (L + 7) // 8 * 8
                assert (off16[sl + 1] - off16[sl]) == n8 // 2 * SLICE
                for step in range(n8):
                    word = w[off16[sl] + (step // 2) * SLICE + np.arange(SLICE)]
                    got = (word >> (16 * (step % 2))) & 0xFFFF
                    want = nbr[off[sl] + step * SLICE + np.arange(SLICE)] if step < L else null
                    assert np.all(got == want)
    assert plan.max_list_steps == max(np.diff(plan.in_off.numpy()).max(),
                                      np.diff(plan.out_off.numpy()).max()) // SLICE


def _same_plan(a, b):
    for k in a._TENSORS:
        x, y = getattr(a, k), getattr(b, k)
        assert x.dtype == y.dtype and x.shape == y.shape, k
        assert torch.equal(x.cpu(), y.cpu()), k
    for k in ("n_hits", "n_pad", "n_segments", "n_features", "n_slices", "n_tiles", "n_chunks",
              "iter_lds_records", "edge_lds_rows", "n_lds_tiles", "iter_lds_in", "iter_lds_out",
              "tile_hits_max", "max_list_steps", "padding", "lds_tile_fraction", "lds_chunk_fraction"):
        assert getattr(a, k) == getattr(b, k), k
    assert np.array_equal(a.src_abs, b.src_abs) and np.array_equal(a.dst_abs, b.dst_abs)
    assert np.array_equal(a.level, b.level)


@pytest.mark.parametrize("case", ["layered", "ragged", "padded", "global", "wide", "empty", "one_graph_big_levels"])
def test_device_builder_builds_the_same_plan(case):
    """plan_device.DeviceSellPlan (torch ops, runs where the batch lives) against plan.SellPlan
    (numpy), array for array."""
    from gnn_fpga_amd.plan_device import DeviceSellPlan
    F, D, lim_over = 3, 8, {}
    if case == "layered":
        graphs = [synth.layered_graph(700, 4000, 3, seed=s) for s in range(5)]
    elif case == "ragged":
        graphs = [synth.layered_graph(n, e, 3, n_layers=L, seed=s)
                  for s, (n, e, L) in enumerate([(40, 90, 10), (3, 2, 2), (300, 2500, 10), (17, 16, 3), (2, 1, 2)])]
    elif case == "padded":
        graphs = [synth.layered_graph(200, 900, 3, seed=s) for s in range(3)]
    elif case == "global":
        graphs = [synth.layered_graph(700, 4000, 3, seed=s) for s in range(3)]
        lim_over = {"iter_records": 0, "edge_records": 0}
    elif case == "wide":
        F, D = 3, 64
        graphs = [synth.layered_graph(500, 3000, 3, seed=s) for s in range(4)]
    elif case == "empty":
        graphs = [synth.layered_graph(50, 0, 3, seed=1)]
    else:
        graphs = [synth.layered_graph(20000, 60000, 3, n_layers=4, seed=3)]
    b = HitGraphBatch.from_graphs(graphs)
    if case == "padded":     # zero-padded segments (src = dst = -1) scattered through the list
        src, dst = b.src.numpy().copy(), b.dst.numpy().copy()
        src[::7] = -1
        dst[::7] = -1
        b = HitGraphBatch(b.X.numpy(), src, dst, hit_ptr=b.hit_ptr, seg_ptr=b.seg_ptr)
    lim = _lib.plan_limits(F, D)
    lim.update(lim_over)
    _same_plan(SellPlan(b, lim), DeviceSellPlan(b, lim))


def test_plan_structure_is_reused_for_other_features():
    """HitGraphBatch.with_features / SellPlan.with_features: the same graphs with other hit features share every
    structure array of the plan (and the index arrays of the batch); the feature rows and their range come out as a
    plan built from scratch has them."""
    graphs = [synth.layered_graph(300, 1500, 3, seed=s) for s in range(3)] + [synth.layered_graph(7, 5, 3, n_layers=2, seed=9)]
    b = HitGraphBatch.from_graphs(graphs)
    lim = dict(tile_hits=64, iter_records=400, chunk_segments=500, edge_records=600)
    b.build_plan(8, lim)
    X2 = torch.randn_like(b.X) * 3.0
    b2 = b.with_features(X2)
    ref = HitGraphBatch(X2.numpy(), b.src.numpy(), b.dst.numpy(), hit_ptr=b.hit_ptr, seg_ptr=b.seg_ptr)
    ref.build_plan(8, lim)
    for k in ref.plan._TENSORS:
        assert torch.equal(getattr(ref.plan, k), getattr(b2.plan, k)), k
    assert b2.plan is not b.plan and b2.plan.in_nbr is b.plan.in_nbr and b2.src is b.src and b2.plan.hidden_dim == 8
    assert torch.equal(b.plan.X[:b.plan.n_pad][b.plan.perm >= 0], b.X[b.plan.perm[b.plan.perm >= 0].long()])   # untouched
    with pytest.raises(ValueError):
        b.with_features(torch.zeros(5, 3))

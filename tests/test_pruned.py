"""SURVEY 8(f) N4 - pruned / masked inference specialisation.  Masks that kill WHOLE units
(reference gnn/model.py:14-33; the pattern of gnn/MPNN_Seg_ACTS_maskedlinear.ipynb cell 34, rows 3 and
4 of the first edge layer) let a narrower network compute the same function: `compact_dead_units`
builds it, the kernels of the smaller hidden_dim run it.  Expected scores: the reference model with
those masks (oracle/gen_golden.py `pruned`)."""
import numpy as np
import pytest
import torch

from gnn_fpga_amd import HitGraphBatch
from gnn_fpga_amd.model import SegmentClassifier, compact_dead_units
from golden_util import Fixture, PRUNED
from oracle import index_c
from oracle.dense_torch import KEYS

TOL = 1e-5          # north_star: edge scores within 1e-5 of the CPU reference
EXPECT = {"pruned_units_d8_s0": 4, "pruned_units_d16_s0": 8, "pruned_units_muon_s0": 4,
          "pruned_edge_only_d8_s0": None}


def _masked_model(fx):
    me = [torch.from_numpy(fx.masks["edge_network.network.0.weight"]),
          torch.from_numpy(fx.masks["edge_network.network.2.weight"])]
    mn = [torch.from_numpy(fx.masks["node_network.network.0.weight"]),
          torch.from_numpy(fx.masks["node_network.network.2.weight"])]
    m = SegmentClassifier(input_dim=fx.F, hidden_dim=fx.D, n_iters=fx.n_iters, masks_e=me, masks_n=mn)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in fx.params.items()})
    return m


@pytest.mark.parametrize("name", PRUNED)
def test_compacted_weights_compute_the_reference_function(name):
    """CPU: the oracle on the compacted (narrower) weights reproduces the reference's scores."""
    fx = Fixture(name)
    eff = fx.effective_params()
    hit = compact_dead_units([torch.from_numpy(eff[k]) for k in KEYS], fx.F, fx.D,
                             [d for d in (4, 8, 16, 32) if d < fx.D])
    if EXPECT[name] is None:
        assert hit is None                 # two dead edge units out of eight: nothing narrower fits
        return
    w, Dn, info = hit
    assert Dn == EXPECT[name] == info["hidden_dim"]
    assert max(info["hit_features"], info["edge_units"], info["node_units"]) <= Dn
    assert w[0].shape == (Dn, fx.F) and w[2].shape == (Dn, 2 * (Dn + fx.F)) and w[8].shape == (Dn, Dn)
    e = index_c.segment_classifier(fx.graph.X, fx.graph.src, fx.graph.dst,
                                   {k: v.numpy() for k, v in zip(KEYS, w)}, fx.n_iters)
    assert np.abs(e - fx.scores).max() < TOL


def test_nothing_is_compacted_without_dead_units():
    fx = Fixture("sector_masked_s0")       # element-wise random masks: every unit still alive
    eff = fx.effective_params()
    assert compact_dead_units([torch.from_numpy(eff[k]) for k in KEYS], fx.F, fx.D, [4]) is None


@pytest.mark.gpu
@pytest.mark.parametrize("name", PRUNED)
@pytest.mark.parametrize("route", ["events", "plan"])
def test_pruned_model_runs_the_narrower_kernels(hip, name, route):
    fx = Fixture(name)
    m = _masked_model(fx).cuda().eval()
    m.use_events = route == "events"
    b = HitGraphBatch.from_graphs([fx.graph]).cuda()
    with torch.no_grad():
        e = m(b)
    assert np.abs(e.cpu().numpy() - fx.scores).max() < TOL
    info = m.pruned_info()
    if EXPECT[name] is None:
        assert info is None
    else:
        assert info["hidden_dim"] == EXPECT[name] and m._w_cache[2].D == EXPECT[name]
        if route == "plan":
            assert b.plan.hidden_dim == EXPECT[name]          # the plan of the narrower kernels
    # the same model with the specialisation off: full-width kernels, same scores
    m.prune_dead_units = False
    m.invalidate()
    b2 = HitGraphBatch.from_graphs([fx.graph]).cuda()
    with torch.no_grad():
        e_full = m(b2)
    assert m.pruned_info() is None and m._w_cache[2].D == fx.D
    assert np.abs(e_full.cpu().numpy() - fx.scores).max() < TOL
    assert (e_full - e).abs().max().item() < TOL
    # training on a pruned model is untouched by the specialisation (autograd path, full width)
    m.train()
    out = m(HitGraphBatch.from_graphs([fx.graph]).cuda())
    out.sum().backward()
    for k, p in m.named_parameters():
        if k in fx.masks:
            assert np.all(p.grad.cpu().numpy()[fx.masks[k] == 0] == 0)


def test_params_struct_rejects_tensors_of_another_width():
    """The kernels index the ten tensors as (F, D) says: handing over tensors compacted to another
    width (W1' [8, 22] read as [16, 38]) must raise before any pointer reaches a kernel."""
    from gnn_fpga_amd import _lib
    fx = Fixture("pruned_units_d16_s0")
    eff = fx.effective_params()
    w, Dn, _ = compact_dead_units([torch.from_numpy(eff[k]) for k in KEYS], fx.F, fx.D, [4, 8])
    with pytest.raises(_lib.GnnHipError, match="hidden_dim=%d needs shape" % fx.D):
        _lib.params_struct(w, fx.F, fx.D)                   # compacted tensors, full-width D
    with pytest.raises(_lib.GnnHipError, match="ten weight tensors"):
        _lib.params_struct(w[:9], fx.F, Dn)


@pytest.mark.gpu
def test_exp_product_bound_is_taken_at_the_width_the_kernels_run(hip):
    """The exp-product decision of a pruned model is made on the compacted weights at THEIR width: the
    bound equals the formula 2 log2(e) max_row(sum |W1 row| (1 or max|X_k|) + |b1_row|) evaluated in
    numpy on the tensors the kernels consume.  A dead edge unit with a large bias (W1 row masked,
    b1 = 30) is folded into b2 by the compaction: the narrower network's bound stays <= 60 (flag on)
    while the full-width network's exceeds it (flag off) - and both score like the oracle."""
    from gnn_fpga_amd import _lib
    fx = Fixture("pruned_units_d16_s0")
    m = _masked_model(fx).cuda().eval()
    m.use_events = False
    dead = int(np.flatnonzero(~fx.masks["edge_network.network.0.weight"].any(axis=1))[0])
    with torch.no_grad():
        m.edge_network.network[0].bias[dead] = 30.0
    params = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    for k, msk in fx.masks.items():
        params[k] = params[k] * msk
    ref = index_c.segment_classifier(fx.graph.X, fx.graph.src, fx.graph.dst, params, fx.n_iters)

    def bound_np(w, F):
        W1, b1 = w[2].cpu().numpy().astype(np.float64), w[3].cpu().numpy().astype(np.float64)
        D = W1.shape[0]
        C = F + D
        xmax = np.abs(fx.graph.X).max(axis=0)
        scale = np.concatenate([np.ones(D), xmax])
        p = np.abs(W1[:, :C]) @ scale + np.abs(b1)
        q = np.abs(W1[:, C:]) @ scale
        return 2.8853900817779268 * max(p.max(), q.max())

    for prune, want_flag in ((True, _lib.GNN_FLAG_EXP_PRODUCT), (False, 0)):
        m.prune_dead_units = prune
        m.invalidate()
        b = HitGraphBatch.from_graphs([fx.graph]).cuda()
        with torch.no_grad():
            e = m(b)
        w, _, D_run = m._cached_weights()
        assert D_run == (8 if prune else fx.D) and w[2].shape[0] == D_run
        got = _lib.exp_product_bound(w, fx.F, D_run, b.plan.x_absmax)
        assert abs(got - bound_np(w, fx.F)) < 1e-3 * max(1.0, got)
        assert (got <= 60.0) == bool(want_flag)
        assert m._xp_cache[1] == want_flag
        assert np.abs(e.cpu().numpy() - ref).max() < TOL

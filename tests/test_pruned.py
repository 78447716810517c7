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

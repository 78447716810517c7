"""Pins every oracle restatement against outputs of the reference itself (tests/golden/)."""
import numpy as np
import pytest
import torch

from golden_util import BATCHES, SINGLE, Fixture
from gnn_fpga_amd import synth
from oracle import dense_torch, index_c, index_numpy


def _t(d):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in d.items()} if d else None


@pytest.mark.parametrize("name", SINGLE)
def test_dense_torch_is_bitwise_the_reference(name):
    fx = Fixture(name)
    X, Ri, Ro = (torch.from_numpy(a)[None] for a in synth.to_dense(fx.graph))
    tr = {}
    with torch.no_grad():
        e = dense_torch.segment_classifier(X, Ri, Ro, _t(fx.params), fx.n_iters, _t(fx.masks), tr)
    # same ATen ops in the same order -> allow only last-bit noise from threading
    np.testing.assert_allclose(e[0].numpy(), fx.scores, rtol=0, atol=1e-7)
    for t in range(fx.n_iters + 1):
        np.testing.assert_allclose(tr["e"][t][0].numpy(), fx.e_trace[t], rtol=0, atol=1e-7)
        np.testing.assert_allclose(tr["H"][t][0].numpy(), fx.H_trace[t], rtol=0, atol=1e-7)


@pytest.mark.parametrize("name", SINGLE)
@pytest.mark.parametrize("f64", [False, True])
def test_index_c_matches_reference(name, f64):
    fx = Fixture(name)
    tr = {}
    e = index_c.segment_classifier(fx.graph.X, fx.graph.src, fx.graph.dst, fx.params,
                                   fx.n_iters, fx.masks, f64=f64, trace=tr)
    tol = 2e-6   # fp32 summation-order noise through T+1 passes; north_star bound is 1e-5
    assert np.abs(e - fx.scores).max() < tol
    for t in range(fx.n_iters + 1):
        assert np.abs(tr["e"][t] - fx.e_trace[t]).max() < tol
        assert np.abs(tr["H"][t] - fx.H_trace[t]).max() < 5e-6


@pytest.mark.parametrize("name", [n for n in SINGLE if "scale" not in n and "reduced" not in n])
def test_index_numpy_matches_reference(name):
    fx = Fixture(name)
    e = index_numpy.segment_classifier(fx.graph.X, fx.graph.src, fx.graph.dst, fx.params,
                                       fx.n_iters, fx.masks)
    assert np.abs(e - fx.scores).max() < 2e-6


@pytest.mark.parametrize("name", BATCHES)
def test_padded_batch_semantics(name):
    """Row P of SURVEY 8(a): padded columns score sigmoid(W2 tanh(b1) + b2); real segments
    are unaffected by padding."""
    fx = Fixture(name)
    p = fx.params
    b1 = p["edge_network.network.0.bias"]
    w2 = p["edge_network.network.2.weight"][0]
    b2 = p["edge_network.network.2.bias"][0]
    e_pad = 1.0 / (1.0 + np.exp(-(np.dot(w2, np.tanh(b1)) + b2)))
    for i, g in enumerate(fx.graphs):
        E = g.src.shape[0]
        e = index_c.segment_classifier(g.X, g.src, g.dst, p, fx.n_iters)
        assert np.abs(e - fx.scores[i, :E]).max() < 2e-6
        assert np.allclose(fx.scores[i, E:], e_pad, atol=1e-6)
        # the oracle's own padded form (src = dst = -1) gives the same constant
        src = np.concatenate([g.src, -np.ones(3, np.int32)])
        dst = np.concatenate([g.dst, -np.ones(3, np.int32)])
        ep = index_c.segment_classifier(g.X, src, dst, p, fx.n_iters)
        assert np.abs(ep[:E] - e).max() == 0.0
        assert np.allclose(ep[E:], e_pad, atol=1e-6)


def test_c3_full_size_against_reference():
    """Config 3 (N=10k, E=100k, F=3, D=8, T=3): the dense reference output was captured in
    the build container; the index-form C oracle reproduces it."""
    fx = Fixture("c3_full_s0")
    e = index_c.segment_classifier(fx.graph.X, fx.graph.src, fx.graph.dst, fx.params, fx.n_iters)
    assert np.abs(e - fx.scores).max() < 5e-6


def test_parameter_counts():
    """Known-answer facts from the reference notebooks."""
    def count(F, D):
        C = D + F
        return (F * D + D) + (2 * C * D + D) + (D + 1) + (3 * C * D + D) + (D * D + D)
    assert [count(3, 8), count(11, 8), count(3, 4), count(2, 32), count(3, 32), count(3, 64)] == \
        [569, 953, 189, 6689, 6881, 26049]
    fx = Fixture("sector_s0")
    assert sum(v.size for v in fx.params.values()) == 569 and len(fx.params) == 10

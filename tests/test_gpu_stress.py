"""Sizes beyond the parity tests' (GPU): many detector graphs in one batch, one graph whose levels
exceed the LDS windows (every tile in the general kernel's global-gather mode), thousands of tiny
graphs - each spot-checked against the C oracle.  (BASELINE's full batch: 256 graphs of
10k hits / 100k segments; a 1 M-hit / 10 M-segment graph; 20 000 graphs of 1-3 hits.)"""
import numpy as np
import pytest
import torch

from gnn_fpga_amd import HitGraphBatch, synth
from oracle import index_c

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model():
    from gnn_fpga_amd.model import SegmentClassifier
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    torch.manual_seed(0)
    m = SegmentClassifier(input_dim=3, hidden_dim=8, n_iters=3).cuda().eval()
    return m, {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}


def _check(model, graphs, sample):
    m, params = model
    b = HitGraphBatch.from_graphs(graphs).cuda()
    with torch.no_grad():
        e = m(b)
    es = b.split_scores(e.cpu().numpy())
    worst = 0.0
    for i in sample:
        g = graphs[i]
        ref = index_c.segment_classifier(g.X, g.src, g.dst, params, 3)
        if ref.size:
            worst = max(worst, float(np.abs(es[i] - ref).max()))
    assert worst < 1e-5
    return b


def test_many_detector_graphs_in_one_batch(model):
    graphs = [synth.layered_graph(10000, 100000, 3, seed=s) for s in range(256)]
    b = _check(model, graphs, [0, 127, 255])
    assert b.plan.n_lds_tiles == b.plan.n_tiles          # the phase-split LDS-window kernel ran


def test_one_graph_with_levels_wider_than_the_lds_windows(model):
    b = _check(model, [synth.layered_graph(1000000, 10000000, 3, seed=7)], [0])
    assert b.plan.n_lds_tiles == 0                        # levels of 100k hits: global-gather tiles


def test_thousands_of_tiny_graphs(model):
    rng = np.random.default_rng(3)
    tiny = []
    for _ in range(20000):
        n = int(rng.integers(1, 4))
        e = int(rng.integers(0, 4))
        X = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
        tiny.append(synth.HitGraph(X, rng.integers(0, n, e).astype(np.int32),
                                   rng.integers(0, n, e).astype(np.int32), np.zeros(e, np.float32)))
    _check(model, tiny, list(range(0, 20000, 997)))

"""GPU parity: the HIP path (through the C ABI) against reference-generated golden vectors
and against the C oracle on seeded inputs.  Tolerance: north_star's 1e-5 absolute on
edge scores (fp32); the kernels actually land near 1e-6."""
import numpy as np
import pytest
import torch

from golden_util import assert_grad_close, BATCHES, SINGLE, Fixture
from gnn_fpga_amd import HitGraphBatch, synth
from oracle import index_c

pytestmark = pytest.mark.gpu

TOL = 1e-5          # north_star: edge scores within 1e-5 of the CPU reference
TOL_H = 2e-5        # intermediate hit features (not a north_star quantity; tracked for drift)


def _weights(fx, dev):
    from oracle.dense_torch import KEYS
    p = fx.effective_params()
    return [torch.from_numpy(np.ascontiguousarray(p[k], dtype=np.float32)).to(dev) for k in KEYS]


def _supported(hip, fx):
    if not hip.shape_supported(fx.F, fx.D):
        pytest.skip("no kernel instantiation for F=%d D=%d yet" % (fx.F, fx.D))


@pytest.mark.parametrize("name", SINGLE)
def test_forward_matches_reference_golden(hip, name):
    fx = Fixture(name)
    _supported(hip, fx)
    dev = torch.device("cuda:0")
    batch = HitGraphBatch.from_graphs([fx.graph]).to(dev)
    e, et, Ht = hip.segclf_forward(batch, _weights(fx, dev), fx.F, fx.D, fx.n_iters, trace=True)
    torch.cuda.synchronize()
    assert np.abs(e.cpu().numpy() - fx.scores).max() < TOL
    for t in range(fx.n_iters + 1):
        assert np.abs(et[t].cpu().numpy() - fx.e_trace[t]).max() < TOL, "e_trace[%d]" % t
        assert np.abs(Ht[t].cpu().numpy() - fx.H_trace[t]).max() < TOL_H, "H_trace[%d]" % t
    # the untraced call (different buffer routing) gives bit-identical scores
    e2 = hip.segclf_forward(batch, _weights(fx, dev), fx.F, fx.D, fx.n_iters)
    assert torch.equal(e, e2)


@pytest.mark.parametrize("name", SINGLE + ["c3_full_s0"])
def test_fused_plan_forward_matches_reference_golden(hip, name):
    """The fast path: relabel + SELL-16 plan, one fused kernel per iteration."""
    fx = Fixture(name)
    if not hip.plan_shape_supported(fx.F, fx.D):
        pytest.skip("no fused kernel for F=%d D=%d" % (fx.F, fx.D))
    dev = torch.device("cuda:0")
    batch = HitGraphBatch.from_graphs([fx.graph]).to(dev)
    plan = batch.build_plan(fx.D)
    w = _weights(fx, dev)
    e = hip.segclf_forward_plan(plan, w, fx.F, fx.D, fx.n_iters)
    torch.cuda.synchronize()
    assert np.abs(e.cpu().numpy() - fx.scores).max() < TOL
    # exp-product mode (2^P' * 2^Q' instead of 2^(P'+Q')) where its range bound is proven
    if hip.exp_product_bound(w, fx.F, fx.D, plan.x_absmax) <= 60.0:
        ex = hip.segclf_forward_plan(plan, w, fx.F, fx.D, fx.n_iters,
                                     flags=hip.GNN_FLAG_EXP_PRODUCT)
        assert np.abs(ex.cpu().numpy() - fx.scores).max() < TOL
    # n_iters = 0 (input network + one edge pass) against the first traced edge pass
    if fx.e_trace is not None:
        e0 = hip.segclf_forward_plan(plan, _weights(fx, dev), fx.F, fx.D, 0)
        assert np.abs(e0.cpu().numpy() - fx.e_trace[0]).max() < TOL


def test_c3_full_size_golden(hip):
    fx = Fixture("c3_full_s0")
    dev = torch.device("cuda:0")
    batch = HitGraphBatch.from_graphs([fx.graph]).to(dev)
    e = hip.segclf_forward(batch, _weights(fx, dev), fx.F, fx.D, fx.n_iters)
    assert np.abs(e.cpu().numpy() - fx.scores).max() < TOL


@pytest.mark.parametrize("name", BATCHES)
def test_padded_batch_dense_dropin(hip, name):
    """The reference's own calling convention: dense zero-padded [B,N,E] one-hot matrices."""
    from gnn_fpga_amd.model import SegmentClassifier
    fx = Fixture(name)
    dev = torch.device("cuda:0")
    B = len(fx.graphs)
    Nmax = max(g.X.shape[0] for g in fx.graphs)
    Emax = fx.scores.shape[1]
    dense = [synth.to_dense(g, Nmax, Emax) for g in fx.graphs]
    X, Ri, Ro = (torch.from_numpy(np.stack([d[i] for d in dense])).to(dev) for i in range(3))
    m = SegmentClassifier(input_dim=fx.F, hidden_dim=fx.D, n_iters=fx.n_iters)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in fx.params.items()})
    m.cuda().eval()
    with torch.no_grad():
        out = m([X, Ri, Ro])
    assert out.shape == (B, Emax)
    assert np.abs(out.cpu().numpy() - fx.scores).max() < TOL


@pytest.mark.parametrize("F,D,T", [(3, 8, 3), (11, 8, 3), (2, 16, 2), (3, 4, 5), (3, 16, 1)])
def test_megabatch_against_oracle(hip, F, D, T):
    """Block-diagonal batch of ragged graphs, incl. an empty-segment graph, vs the C oracle."""
    rng = np.random.default_rng(5)
    graphs = [synth.layered_graph(int(rng.integers(20, 400)), int(rng.integers(30, 2500)), F,
                                  seed=100 + i) for i in range(7)]
    g0 = graphs[0]
    graphs.append(synth.HitGraph(g0.X[:12], np.zeros(0, np.int32), np.zeros(0, np.int32),
                                 np.zeros(0, np.float32)))      # hits but no segments
    torch.manual_seed(F * 100 + D)
    from gnn_fpga_amd.model import SegmentClassifier
    m = SegmentClassifier(input_dim=F, hidden_dim=D, n_iters=T).cuda().eval()
    params = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    batch = HitGraphBatch.from_graphs(graphs).cuda()
    m.use_events = False                 # (the one-launch small-event path has its own tests)
    for use_plan in (True, False):       # fused pipeline, then per-module CSR kernels
        m.use_plan = use_plan
        with torch.no_grad():
            e = m(batch).cpu().numpy()
        for g, eg in zip(graphs, batch.split_scores(e)):
            ref = index_c.segment_classifier(g.X, g.src, g.dst, params, T)
            assert eg.shape == ref.shape
            if ref.size:
                assert np.abs(eg - ref).max() < TOL


def test_submodules_like_the_notebooks(hip):
    """model.input_network / edge_network / node_network called directly
    (reference gnn/MPNN_Seg_ACTS_maskedlinear.ipynb cell 42,46)."""
    from gnn_fpga_amd.model import SegmentClassifier
    fx = Fixture("sector_masked_s0")
    dev = torch.device("cuda:0")
    me = [torch.from_numpy(fx.masks["edge_network.network.0.weight"]),
          torch.from_numpy(fx.masks["edge_network.network.2.weight"])]
    mn = [torch.from_numpy(fx.masks["node_network.network.0.weight"]),
          torch.from_numpy(fx.masks["node_network.network.2.weight"])]
    m = SegmentClassifier(input_dim=fx.F, hidden_dim=fx.D, n_iters=fx.n_iters,
                          masks_e=me, masks_n=mn)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in fx.params.items()})
    m.cuda().eval()
    X, Ri, Ro = (torch.from_numpy(a)[None].to(dev) for a in synth.to_dense(fx.graph))
    with torch.no_grad():
        H0 = torch.from_numpy(fx.H_trace[0])[None].to(dev)
        e0 = m.edge_network(H0, Ri, Ro)
        assert np.abs(e0[0].cpu().numpy() - fx.e_trace[0]).max() < TOL
        H1 = m.node_network(H0, e0, Ri, Ro)
        assert H1.shape == (1, fx.graph.X.shape[0], fx.D)
        assert np.abs(H1[0].cpu().numpy() - fx.H_trace[1][:, :fx.D]).max() < TOL_H
        out = m([X, Ri, Ro])
        assert np.abs(out[0].cpu().numpy() - fx.scores).max() < TOL


@pytest.mark.parametrize("F,D", [(3, 8), (11, 8), (2, 32)])
def test_submodules_are_differentiable_like_the_reference(hip, F, D):
    """model.edge_network(H, Ri, Ro) / model.node_network(H, e, Ri, Ro) are ordinary autograd modules
    in the reference (gnn/model.py:69-81,113-125): gradients w.r.t. the hit features, the scores and
    the masked weights through the HIP backward (gnn_edge_bwd / gnn_node_bwd) against autograd through
    the dense oracle, on a padded batch of two graphs."""
    from gnn_fpga_amd.model import SegmentClassifier
    from oracle import dense_torch
    torch.manual_seed(F * 10 + D)
    C = F + D
    graphs = [synth.layered_graph(60, 200, F, seed=1), synth.layered_graph(45, 120, F, seed=2)]
    Nmax, Emax = 60, 200
    dense = [synth.to_dense(g, Nmax, Emax) for g in graphs]
    Ri, Ro = (torch.from_numpy(np.stack([d[i] for d in dense])) for i in (1, 2))
    me = [(torch.rand(D, 2 * C) < 0.8).float(), torch.ones(1, D)]
    mn = [(torch.rand(D, 3 * C) < 0.8).float(), (torch.rand(D, D) < 0.9).float()]
    m = SegmentClassifier(input_dim=F, hidden_dim=D, n_iters=1, masks_e=me, masks_n=mn).cuda().train()
    params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    masks = {"edge_network.network.0.weight": me[0], "edge_network.network.2.weight": me[1],
             "node_network.network.0.weight": mn[0], "node_network.network.2.weight": mn[1]}
    H0 = torch.randn(2, Nmax, C) * 0.5
    we, wh = torch.randn(2, Emax), torch.randn(2, Nmax, D)
    # reference formulation on the CPU
    Hc = H0.clone().requires_grad_(True)
    e_c = dense_torch.edge_network(Hc, Ri, Ro, params, masks)
    Hn_c = dense_torch.node_network(Hc, e_c, Ri, Ro, params, masks)
    ((e_c * we).sum() + (Hn_c * wh).sum()).backward()
    # HIP modules
    Hg = H0.cuda().requires_grad_(True)
    e_g = m.edge_network(Hg, Ri.cuda(), Ro.cuda())
    assert e_g.requires_grad and e_g.shape == (2, Emax)
    Hn_g = m.node_network(Hg, e_g, Ri.cuda(), Ro.cuda())
    assert Hn_g.requires_grad and Hn_g.shape == (2, Nmax, D)
    assert np.abs(e_g.detach().cpu().numpy() - e_c.detach().numpy()).max() < TOL
    assert np.abs(Hn_g.detach().cpu().numpy() - Hn_c.detach().numpy()).max() < TOL_H
    ((e_g * we.cuda()).sum() + (Hn_g * wh.cuda()).sum()).backward()

    def close(a, r, what):
        assert_grad_close(a, r, "submodule " + what)

    close(Hg.grad.cpu().numpy(), Hc.grad.numpy(), "dL/dH")
    for k, p_ in m.named_parameters():
        if k.startswith("input_network"):
            assert p_.grad is None
            continue
        close(p_.grad.cpu().numpy(), params[k].grad.numpy(), k)
        if k in masks:
            assert np.all(p_.grad.cpu().numpy()[masks[k].numpy() == 0] == 0)
    # and nothing is detached silently in inference either: no grad -> plain tensors
    with torch.no_grad():
        assert not m.edge_network(Hg, Ri.cuda(), Ro.cuda()).requires_grad


def test_full_size_properties(hip):
    """Config-3 size (10k hits / 100k segments) x 8 graphs: size-independent properties -
    block-diagonal independence (each graph's scores equal its stand-alone run bit for bit),
    padding invariance, determinism across runs."""
    from gnn_fpga_amd.model import SegmentClassifier
    torch.manual_seed(0)
    m = SegmentClassifier(input_dim=3, hidden_dim=8, n_iters=3).cuda().eval()
    graphs = [synth.layered_graph(10000, 100000, 3, seed=s) for s in range(8)]
    batch = HitGraphBatch.from_graphs(graphs).cuda()
    with torch.no_grad():
        e = m(batch)
        e_again = m(batch)
        assert torch.equal(e, e_again)
        assert bool(((e > 0) & (e < 1)).all())
        single = m(HitGraphBatch.from_graphs([graphs[3]]).cuda())
        assert torch.equal(batch.split_scores(e)[3], single)
        # pad graph 3 with 1000 fake segments: real scores unchanged, pads score the constant
        g = graphs[3]
        pad = -np.ones(1000, np.int32)
        gp = HitGraphBatch(g.X, np.concatenate([g.src, pad]), np.concatenate([g.dst, pad])).cuda()
        ep = m(gp)
        assert torch.equal(ep[:100000], single)
        w = {k: v.detach().cpu().double() for k, v in m.state_dict().items()}
        e_pad = torch.sigmoid(w["edge_network.network.2.weight"][0] @
                              torch.tanh(w["edge_network.network.0.bias"]) +
                              w["edge_network.network.2.bias"][0]).item()
        assert abs(ep[100000:].cpu().numpy() - e_pad).max() < 1e-6
    params = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    ref = index_c.segment_classifier(g.X, g.src, g.dst, params, 3)
    assert np.abs(single.cpu().numpy() - ref).max() < TOL


def test_cpu_tensors_fail_loudly(hip):
    from gnn_fpga_amd.model import SegmentClassifier
    m = SegmentClassifier(input_dim=3, hidden_dim=8, n_iters=1).eval()
    g = synth.layered_graph(30, 50, 3, seed=0)
    with torch.no_grad(), pytest.raises(RuntimeError):
        m(HitGraphBatch.from_graphs([g]))


@pytest.mark.parametrize("name", BATCHES)
def test_training_step_matches_reference(hip, name):
    """One training step as gnn/estimator.py:49-60 runs it (BCELoss mean over all B x E_max
    entries, padded ones included): loss and all ten gradients against the values captured from
    the reference's own Estimator.training_step."""
    from gnn_fpga_amd.model import SegmentClassifier
    fx = Fixture(name)
    dev = torch.device("cuda:0")
    Nmax = max(g.X.shape[0] for g in fx.graphs)
    Emax = fx.scores.shape[1]
    dense = [synth.to_dense(g, Nmax, Emax) for g in fx.graphs]
    X, Ri, Ro = (torch.from_numpy(np.stack([d[i] for d in dense])).to(dev) for i in range(3))
    m = SegmentClassifier(input_dim=fx.F, hidden_dim=fx.D, n_iters=fx.n_iters)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in fx.params.items()})
    m.cuda().train()
    m.zero_grad()
    out = m([X, Ri, Ro])
    assert out.requires_grad and out.shape == (len(fx.graphs), Emax)
    loss = torch.nn.BCELoss()(out, torch.from_numpy(fx.y).to(dev))
    loss.backward()
    assert abs(loss.item() - fx.loss) < 1e-6
    for k, p in m.named_parameters():
        ref = fx.grads[k]
        assert_grad_close(p.grad, ref, "reference training step " + k)


def test_masked_training_gradients(hip):
    """Masks: gradients of masked weights vanish where mask = 0 (W*mask in autograd,
    gnn/model.py:30), and the whole gradient matches autograd through the dense oracle."""
    from gnn_fpga_amd.model import SegmentClassifier
    from oracle import dense_torch
    fx = Fixture("sector_masked_s1")
    dev = torch.device("cuda:0")
    me = [torch.from_numpy(fx.masks["edge_network.network.0.weight"]),
          torch.from_numpy(fx.masks["edge_network.network.2.weight"])]
    mn = [torch.from_numpy(fx.masks["node_network.network.0.weight"]),
          torch.from_numpy(fx.masks["node_network.network.2.weight"])]
    m = SegmentClassifier(input_dim=fx.F, hidden_dim=fx.D, n_iters=fx.n_iters, masks_e=me, masks_n=mn)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in fx.params.items()})
    m.cuda().train()
    batch = HitGraphBatch.from_graphs([fx.graph]).cuda()
    y = (torch.arange(batch.n_segments) % 3 == 0).float()
    loss = torch.nn.BCELoss()(m(batch), y.to(dev))
    loss.backward()
    # oracle: autograd through the dense restatement on CPU
    params = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in fx.params.items()}
    masks = {k: torch.from_numpy(v) for k, v in fx.masks.items()}
    Xd, Ri, Ro = (torch.from_numpy(a)[None] for a in synth.to_dense(fx.graph))
    ref = torch.nn.BCELoss()(dense_torch.segment_classifier(Xd, Ri, Ro, params, fx.n_iters, masks)[0], y)
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-6
    for k, p in m.named_parameters():
        g = p.grad.cpu().numpy()
        r = params[k].grad.numpy()
        assert_grad_close(g, r, "masked " + k)
        if k in fx.masks:
            assert np.all(g[fx.masks[k] == 0] == 0)


def test_reference_training_step_with_l1_and_masks(hip):
    """The reference's training step WITH its L1 branch (gnn/estimator.py:49-60, `l1 > 0`): the loss
    adds l1 * sum |W| over `layer.weight` of every layer of node_network.network and
    edge_network.network that has one (:54-56; the raw weights, not W * mask), then loss.backward() and
    optimizer.step().  Run statement for statement on the drop-in with masks set, against autograd
    through the dense oracle with the same statements on CPU: loss, all ten gradients (the L1 term's
    sign(W) included, zero where the mask zeroed W), and the weights after one SGD step."""
    from gnn_fpga_amd.model import SegmentClassifier
    from oracle import dense_torch
    fx = Fixture("sector_masked_s2")
    l1, lr = 1e-3, 0.05
    me = [torch.from_numpy(fx.masks["edge_network.network.0.weight"]),
          torch.from_numpy(fx.masks["edge_network.network.2.weight"])]
    mn = [torch.from_numpy(fx.masks["node_network.network.0.weight"]),
          torch.from_numpy(fx.masks["node_network.network.2.weight"])]
    model = SegmentClassifier(input_dim=fx.F, hidden_dim=fx.D, n_iters=fx.n_iters, masks_e=me, masks_n=mn)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in fx.params.items()})
    model.cuda().train()
    for layer in (model.edge_network.network[0], model.edge_network.network[2],
                  model.node_network.network[0], model.node_network.network[2]):
        layer.set_mask(layer.mask)              # (as the constructor did before load_state_dict: zero W where masked)
    start = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    optimizer = torch.optim.SGD(model.parameters(), lr=lr)
    loss_func = torch.nn.BCELoss()
    batch = HitGraphBatch.from_graphs([fx.graph]).cuda()
    targets = (torch.arange(batch.n_segments) % 4 == 0).float()

    def l1_penalty(arr):                        # gnn/estimator.py:46-47
        return torch.abs(arr).sum()

    # --- gnn/estimator.py:49-60, on the drop-in
    model.zero_grad()
    optimizer.zero_grad()
    outputs = model(batch)
    node_weights = [layer.weight for layer in model.node_network.network if hasattr(layer, 'weight')]
    edge_weights = [layer.weight for layer in model.edge_network.network if hasattr(layer, 'weight')]
    assert len(node_weights) == 2 and len(edge_weights) == 2
    l1_regularization = l1 * sum([l1_penalty(arr) for arr in node_weights]) + l1 * sum([l1_penalty(arr) for arr in edge_weights])
    loss = loss_func(outputs, targets.cuda()) + l1_regularization
    loss.backward()
    grads = {k: p.grad.detach().cpu().numpy().copy() for k, p in model.named_parameters()}
    optimizer.step()

    # --- the same statements through the dense oracle (CPU autograd)
    params = {k: v.clone().requires_grad_(True) for k, v in start.items()}
    masks = {k: torch.from_numpy(v) for k, v in fx.masks.items()}
    Xd, Ri, Ro = (torch.from_numpy(a)[None] for a in synth.to_dense(fx.graph))
    ref_out = dense_torch.segment_classifier(Xd, Ri, Ro, params, fx.n_iters, masks)[0]
    ref_l1 = (l1 * sum(l1_penalty(params[k]) for k in ("node_network.network.0.weight", "node_network.network.2.weight")) +
              l1 * sum(l1_penalty(params[k]) for k in ("edge_network.network.0.weight", "edge_network.network.2.weight")))
    ref = loss_func(ref_out, targets) + ref_l1
    ref.backward()
    assert float(ref_l1) > 1e-3                  # the branch is live
    assert abs(loss.item() - ref.item()) < 2e-6
    for k, p in model.named_parameters():
        r = params[k].grad.numpy()
        assert_grad_close(grads[k], r, "l1 step " + k)
        if k in fx.masks:
            assert np.all(grads[k][fx.masks[k] == 0] == 0)           # |w|' = 0 at the zeroed weights, W * mask elsewhere
        stepped = start[k].numpy() - lr * r
        assert np.abs(p.detach().cpu().numpy() - stepped).max() < 1e-6 + 2e-6 * np.abs(stepped).max(), k


def test_exp_product_bound_and_fallback(hip):
    """The model takes the exp-product fast path only while max|P'|,|Q'| <= 60 is proven; with
    huge first-layer weights it must fall back to the exact path and still match the oracle."""
    from gnn_fpga_amd.model import SegmentClassifier
    torch.manual_seed(3)
    g = synth.layered_graph(300, 1500, 3, seed=9)
    batch = HitGraphBatch.from_graphs([g]).cuda()
    m = SegmentClassifier(input_dim=3, hidden_dim=8, n_iters=2).cuda().eval()
    m.use_events = False                                           # this is about the tiled pipeline
    with torch.no_grad():
        e = m(batch)
        assert m._xp_cache[1] == hip.GNN_FLAG_EXP_PRODUCT          # default init: tiny bound
        m.edge_network.network[0].weight.mul_(40.0)                # pre-activations of +-100
        e_big = m(batch)
        assert m._xp_cache[1] == 0                                 # in-place update seen, exact path
    params = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    ref = index_c.segment_classifier(g.X, g.src, g.dst, params, 2)
    assert np.abs(e_big.cpu().numpy() - ref).max() < TOL
    assert not torch.equal(e, e_big)


def test_exp_product_decision_belongs_to_the_batch_not_to_a_recycled_id(hip):
    """An inference loop `model(HitGraphBatch.from_graphs(g).to(dev))` frees every plan after its
    forward, and CPython hands the freed object's id() to the next one: the exp-product decision
    (taken from the batch's |X| range) must not travel with the id.  A small-|X| batch, freed, then
    a batch whose features make 2^P' overflow: the exact kernels must run and match the oracle."""
    from gnn_fpga_amd.model import SegmentClassifier
    torch.manual_seed(3)
    m = SegmentClassifier(input_dim=3, hidden_dim=8, n_iters=2).cuda().eval()
    m.use_events = False
    params = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    small = synth.layered_graph(300, 1500, 3, seed=9)
    big = synth.HitGraph(small.X * 400.0, small.src, small.dst, small.y)     # |X| up to 400
    seen = []
    with torch.no_grad():
        for _ in range(4):                       # ids of freed plans get reused within a few rounds
            for g, want in ((small, hip.GNN_FLAG_EXP_PRODUCT), (big, 0)):
                e = m(HitGraphBatch.from_graphs([g]).cuda())
                seen.append(m._xp_cache[1])
                assert m._xp_cache[1] == want
                ref = index_c.segment_classifier(g.X, g.src, g.dst, params, 2)
                assert np.abs(e.cpu().numpy() - ref).max() < TOL
    assert seen == [hip.GNN_FLAG_EXP_PRODUCT, 0] * 4


def test_batch_moved_between_devices_rebuilds_its_cached_structs(hip):
    """cuda -> cpu -> cuda: the cached C structs hold raw device pointers of tensors that no longer
    exist; HitGraphBatch.to() / SellPlan.to() must drop them."""
    from gnn_fpga_amd.model import SegmentClassifier
    torch.manual_seed(2)
    m = SegmentClassifier(input_dim=3, hidden_dim=8, n_iters=2).cuda().eval()
    graphs = [synth.layered_graph(500, 3000, 3, seed=70 + i) for i in range(3)]
    for events in (False, True):
        m.use_events = events
        b = HitGraphBatch.from_graphs(graphs if not events else
                                      [synth.layered_graph(40, 120, 3, seed=5)]).cuda()
        with torch.no_grad():
            e1 = m(b).clone()
            b.to("cpu")
            junk = [torch.empty(1 << 20, device="cuda") for _ in range(8)]   # reuse the freed blocks
            for j in junk:
                j.fill_(-1.0)
            b.to("cuda")
            e2 = m(b)
        assert torch.equal(e1, e2)


def _random_graph(n, e, F, seed, self_loops=True):
    """Not layered: arbitrary endpoints, multi-edges, cycles and (optionally) self loops."""
    rng = np.random.default_rng(seed)
    X = rng.uniform(-1, 1, (n, F)).astype(np.float32)
    src = rng.integers(0, n, e).astype(np.int32)
    dst = rng.integers(0, n, e).astype(np.int32)
    if not self_loops:
        dst = np.where(dst == src, (dst + 1) % n, dst).astype(np.int32)
    return synth.HitGraph(X, src, dst, np.zeros(e, np.float32))


@pytest.mark.parametrize("F,D,T", [(3, 8, 3), (2, 4, 2), (11, 8, 2), (3, 16, 2)])
def test_irregular_graphs_and_forced_global_mode(hip, F, D, T):
    """Graphs without any layer structure (cycles, multi-edges, self loops) and plans forced into
    global-gather mode: the general kernels (k_iter / k_edge without LDS windows) against the
    C oracle, then the same batch through whatever the default plan picks."""
    from gnn_fpga_amd.model import SegmentClassifier
    torch.manual_seed(7)
    graphs = [_random_graph(400, 3000, F, 1), _random_graph(37, 90, F, 2),
              synth.layered_graph(600, 5000, F, seed=3), _random_graph(5, 40, F, 4)]
    m = SegmentClassifier(input_dim=F, hidden_dim=D, n_iters=T).cuda().eval()
    m.use_events = False
    params = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    refs = [index_c.segment_classifier(g.X, g.src, g.dst, params, T) for g in graphs]
    for limits in ({"iter_records": 0, "edge_records": 0}, None):
        batch = HitGraphBatch.from_graphs(graphs).cuda()
        plan = batch.build_plan(D, limits)
        if limits:
            assert plan.n_lds_tiles == 0 and plan.lds_chunk_fraction == 0.0
        with torch.no_grad():
            e = m(batch).cpu().numpy()
        for eg, ref in zip(batch.split_scores(e), refs):
            assert np.abs(eg - ref).max() < TOL


def test_all_padded_and_degenerate_batches(hip):
    """Only padded segments; hits without any segment; a single hit."""
    from gnn_fpga_amd.model import SegmentClassifier
    torch.manual_seed(1)
    m = SegmentClassifier(input_dim=3, hidden_dim=8, n_iters=2).cuda().eval()
    w = {k: v.detach().cpu().double() for k, v in m.state_dict().items()}
    e_pad = torch.sigmoid(w["edge_network.network.2.weight"][0] @
                          torch.tanh(w["edge_network.network.0.bias"]) +
                          w["edge_network.network.2.bias"][0]).item()
    X = np.random.default_rng(0).uniform(-1, 1, (9, 3)).astype(np.float32)
    pad = -np.ones(6, np.int32)
    with torch.no_grad():
        e = m(HitGraphBatch(X, pad, pad).cuda())
        assert e.shape == (6,) and np.abs(e.cpu().numpy() - e_pad).max() < 1e-6
        e0 = m(HitGraphBatch(X, np.zeros(0, np.int32), np.zeros(0, np.int32)).cuda())
        assert e0.shape == (0,)
        e1 = m(HitGraphBatch(X[:1], np.zeros(1, np.int32), np.zeros(1, np.int32)).cuda())   # self loop
    params = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    ref = index_c.segment_classifier(X[:1], np.zeros(1, np.int32), np.zeros(1, np.int32), params, 2)
    assert np.abs(e1.cpu().numpy() - ref).max() < TOL


def test_training_loop_like_estimator_fit_gen(hip):
    """The loop of gnn/estimator.py:98-104 (zero_grad, forward, BCELoss, backward, Adam step) on
    the HIP forward/backward, with the flat gradient all-reduce helper in its single-rank form:
    the loss must fall and must track the same loop run through the dense oracle on CPU."""
    from gnn_fpga_amd import shard
    from gnn_fpga_amd.model import SegmentClassifier
    from oracle import dense_torch
    torch.manual_seed(5)
    g = synth.layered_graph(150, 600, 3, seed=11)
    y = torch.from_numpy(g.y)
    m = SegmentClassifier(input_dim=3, hidden_dim=8, n_iters=2).cuda().train()
    ref = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    opt = torch.optim.Adam(m.parameters(), lr=0.02)
    opt_ref = torch.optim.Adam(ref.values(), lr=0.02)
    batch = HitGraphBatch.from_graphs([g]).cuda()
    Xd, Ri, Ro = (torch.from_numpy(a)[None] for a in synth.to_dense(g))
    losses, losses_ref = [], []
    for _ in range(25):
        m.zero_grad()
        out = m(batch)
        loss_sum = torch.nn.functional.binary_cross_entropy(out, y.cuda(), reduction="sum")
        loss_sum.backward()
        losses.append(shard.allreduce_step(m.parameters(), loss_sum.item(), out.numel()))
        opt.step()
        opt_ref.zero_grad()
        lr = torch.nn.BCELoss()(dense_torch.segment_classifier(Xd, Ri, Ro, ref, 2)[0], y)
        lr.backward()
        opt_ref.step()
        losses_ref.append(lr.item())
    assert losses[-1] < 0.8 * losses[0]
    assert np.abs(np.array(losses) - np.array(losses_ref)).max() < 2e-4


# ---- small events: whole forward in one launch, one workgroup per graph -------------------------
def _muon_batch(n_graphs, seed=0):
    return [synth.muon_graph(seed=seed + i) for i in range(n_graphs)]


@pytest.mark.parametrize("F,D,T,kind", [(11, 8, 3, "muon"), (3, 8, 3, "layered"), (2, 4, 2, "layered"),
                                        (3, 16, 2, "layered"), (11, 16, 1, "muon")])
def test_small_event_kernel_is_bit_identical_to_the_module_kernels(hip, F, D, T, kind):
    """k_event (one workgroup per graph, H and scores resident in LDS) runs the same arithmetic
    in the same order as k_input / k_edge / k_node: scores must be EQUAL, and within 1e-5 of the
    C oracle; covers ragged sizes, a graph without segments and a graph without hits."""
    from gnn_fpga_amd.model import SegmentClassifier
    rng = np.random.default_rng(11)
    if kind == "muon":
        graphs = _muon_batch(37, seed=3)
    else:
        graphs = [synth.layered_graph(int(rng.integers(4, 150)), int(rng.integers(1, 600)), F,
                                      seed=40 + i) for i in range(23)]
    g0 = graphs[0]
    graphs.insert(5, synth.HitGraph(g0.X[:7], np.zeros(0, np.int32), np.zeros(0, np.int32),
                                    np.zeros(0, np.float32)))
    graphs.insert(9, synth.HitGraph(g0.X[:0], np.zeros(0, np.int32), np.zeros(0, np.int32),
                                    np.zeros(0, np.float32)))
    torch.manual_seed(F * 10 + D)
    m = SegmentClassifier(input_dim=F, hidden_dim=D, n_iters=T).cuda().eval()
    batch = HitGraphBatch.from_graphs(graphs).cuda()
    lay = batch.event_layout()
    assert lay is not None and hip.events_supported(F, D, lay.max_hits, lay.max_segments)
    with torch.no_grad():
        m.use_events = True
        e_ev = m(batch)
        m.use_events, m.use_plan = False, False
        e_csr = m(batch)
    torch.cuda.synchronize()
    assert torch.equal(e_ev, e_csr)
    params = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    for g, eg in zip(graphs, batch.split_scores(e_ev.cpu().numpy())):
        ref = index_c.segment_classifier(g.X, g.src, g.dst, params, T)
        assert eg.shape == ref.shape
        if ref.size:
            assert np.abs(eg - ref).max() < TOL


@pytest.mark.parametrize("name", BATCHES)
def test_small_event_kernel_on_the_reference_padded_batches(hip, name):
    """Zero-padded dense batches (gnn/trainSegmentClassifier.py:66-95) take the one-launch path
    too: padded segments score sigmoid(W2 tanh(b1) + b2), like the reference."""
    from gnn_fpga_amd.model import SegmentClassifier
    fx = Fixture(name)
    dev = torch.device("cuda:0")
    Nmax = max(g.X.shape[0] for g in fx.graphs)
    Emax = fx.scores.shape[1]
    dense = [synth.to_dense(g, Nmax, Emax) for g in fx.graphs]
    X, Ri, Ro = (torch.from_numpy(np.stack([d[i] for d in dense])).to(dev) for i in range(3))
    m = SegmentClassifier(input_dim=fx.F, hidden_dim=fx.D, n_iters=fx.n_iters)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in fx.params.items()})
    m.cuda().eval()
    batch = HitGraphBatch.from_dense(X, Ri, Ro).to(dev)
    lay = batch.event_layout()
    assert lay is not None and hip.events_supported(fx.F, fx.D, lay.max_hits, lay.max_segments)
    with torch.no_grad():
        out = m(batch)
    assert np.abs(out.cpu().numpy() - fx.scores).max() < TOL


def test_small_event_path_declines_what_it_cannot_hold(hip):
    """Large graphs and batches that are not block-diagonal fall through to the tiled pipeline."""
    from gnn_fpga_amd.model import SegmentClassifier
    big = synth.layered_graph(3000, 20000, 3, seed=1)
    b = HitGraphBatch.from_graphs([big]).cuda()
    assert b.event_layout() is None                       # beyond EVENTS_MAX_SEGMENTS: not even checked on the host
    assert not hip.events_supported(3, 8, 3000, 20000)    # ... and it would not fit one workgroup's LDS anyway
    # segments crossing graph boundaries: legal for the global kernels, not for one-graph-per-workgroup
    g = synth.layered_graph(60, 100, 3, seed=2)
    cross = HitGraphBatch(g.X, g.src, g.dst, hit_ptr=[0, 30, 60], seg_ptr=[0, 50, 100]).cuda()
    assert cross.event_layout() is None
    torch.manual_seed(0)
    m = SegmentClassifier(input_dim=3, hidden_dim=8, n_iters=2).cuda().eval()
    params = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    with torch.no_grad():
        e = m(cross).cpu().numpy()
    assert np.abs(e - index_c.segment_classifier(g.X, g.src, g.dst, params, 2)).max() < TOL


@pytest.mark.parametrize("F,D,T", [(2, 32, 3), (3, 64, 2), (3, 32, 2)])
def test_wide_hidden_dims_train_and_submodules(hip, F, D, T):
    """hidden_dim 32 (the reference's toy and ACTS notebooks) and 64 (mu200 notebook): the
    per-module kernels, the small-event kernel and the backward kernels against the dense oracle
    (forward) and autograd through it (all ten gradients)."""
    from gnn_fpga_amd.model import SegmentClassifier
    from oracle import dense_torch
    g = synth.layered_graph(90, 260, F, seed=21)
    torch.manual_seed(D)
    m = SegmentClassifier(input_dim=F, hidden_dim=D, n_iters=T).cuda()
    params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    batch = HitGraphBatch.from_graphs([g]).cuda()
    Xd, Ri, Ro = (torch.from_numpy(a)[None] for a in synth.to_dense(g))
    y = (torch.arange(batch.n_segments) % 2 == 0).float()
    ref_e = dense_torch.segment_classifier(Xd, Ri, Ro, params, T)[0]
    ref = torch.nn.BCELoss()(ref_e, y)
    ref.backward()
    m.eval()
    with torch.no_grad():
        for events, plan in ((True, True), (False, True), (False, False)):
            m.use_events, m.use_plan = events, plan
            e = m(batch)
            assert np.abs(e.cpu().numpy() - ref_e.detach().numpy()).max() < TOL, (events, plan)
    m.train()
    loss = torch.nn.BCELoss()(m(batch), y.cuda())
    loss.backward()
    assert abs(loss.item() - ref.item()) < 1e-6
    for k, p in m.named_parameters():
        gk, r = p.grad.cpu().numpy(), params[k].grad.numpy()
        assert_grad_close(gk, r, "dense oracle " + k)


@pytest.mark.parametrize("reduction", ["mean", "sum"])
def test_fused_bce_loss_matches_torch(hip, reduction):
    """gnn_fpga_amd.loss.BCELoss (one HIP pass: value + gradient) against torch.nn.BCELoss on the
    same device tensors (fp32 reference of the same op): value, gradient, torch's clamps at
    scores of exactly 0 and 1, empty input, 2-D [B, E] shape like the reference's batches."""
    from gnn_fpga_amd.loss import BCELoss
    torch.manual_seed(2)
    dev = torch.device("cuda:0")
    for shape in ((100003,), (4, 250), (1,)):
        e = torch.rand(shape, device=dev)
        y = (torch.rand(shape, device=dev) < 0.3).float()
        flat = e.view(-1)
        if flat.numel() > 10:
            flat[0], flat[1], flat[2], flat[3] = 0.0, 1.0, 0.0, 1.0      # exact 0 / 1 scores
            y.view(-1)[0:4] = torch.tensor([0.0, 1.0, 1.0, 0.0], device=dev)
        e1 = e.clone().requires_grad_(True)
        e2 = e.clone().requires_grad_(True)
        l1 = BCELoss(reduction)(e1, y)
        l2 = torch.nn.BCELoss(reduction=reduction)(e2, y)
        (3.0 * l1).backward()
        (3.0 * l2).backward()
        assert l1.shape == l2.shape == ()
        assert abs(l1.item() - l2.item()) <= 2e-6 * max(1.0, abs(l2.item()))
        assert torch.allclose(e1.grad, e2.grad, rtol=1e-5, atol=1e-12)
    with pytest.raises(Exception):
        BCELoss()(torch.rand(4), torch.rand(4))                           # CPU tensors: no fallback


def test_training_step_with_fused_loss_matches_reference(hip):
    """Estimator.training_step (gnn/estimator.py:51-59) with both the model and the loss on the HIP
    path: loss value and all ten gradients against the reference-generated fixture."""
    from gnn_fpga_amd.loss import BCELoss
    from gnn_fpga_amd.model import SegmentClassifier
    fx = Fixture(BATCHES[0])
    dev = torch.device("cuda:0")
    Nmax = max(g.X.shape[0] for g in fx.graphs)
    Emax = fx.scores.shape[1]
    dense = [synth.to_dense(g, Nmax, Emax) for g in fx.graphs]
    X, Ri, Ro = (torch.from_numpy(np.stack([d[i] for d in dense])).to(dev) for i in range(3))
    y = torch.from_numpy(fx.y).to(dev)
    m = SegmentClassifier(input_dim=fx.F, hidden_dim=fx.D, n_iters=fx.n_iters)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in fx.params.items()})
    m.cuda().train()
    m.zero_grad()
    loss = BCELoss()(m([X, Ri, Ro]), y)
    loss.backward()
    assert abs(loss.item() - fx.loss) < 1e-6
    for k, p in m.named_parameters():
        g, r = p.grad.cpu().numpy(), fx.grads[k]
        assert_grad_close(g, r, "fixture " + k)


@pytest.mark.parametrize("F,D", [(3, 8), (11, 8), (2, 32)])
def test_training_gradients_are_additive_over_graphs(hip, F, D):
    """Grids beyond 8 workgroups take the XCD-contiguous item order (xcd_block, the split segment
    ranges of k_edge_bwd) and spread their gradient atomics over the per-XCD replicas; small ones
    do not.  With a sum-reduced loss the gradient of a 40-graph batch (47 / 235 workgroups) must
    equal the sum of the 40 single-graph gradients (2 / 6 workgroups each), and its training-mode
    scores the single-graph scores."""
    from gnn_fpga_amd.loss import BCELoss
    from gnn_fpga_amd.model import SegmentClassifier
    torch.manual_seed(F * D)
    graphs = [synth.layered_graph(300, 1500, F, seed=200 + i) for i in range(40)]
    m = SegmentClassifier(input_dim=F, hidden_dim=D, n_iters=2).cuda().train()
    bce = BCELoss(reduction="sum")

    def grads_of(gs):
        b = HitGraphBatch.from_graphs(gs).cuda()
        y = (torch.arange(b.n_segments, device="cuda") % 3 == 0).float()
        m.zero_grad()
        e = m(b)
        bce(e, y).backward()
        return e.detach().cpu().numpy(), [p.grad.detach().cpu().double().numpy().copy() for p in m.parameters()]

    e_all, g_all = grads_of(graphs)
    # labels are a function of the position in the batch: rebuild them per graph the same way
    off, e_parts, g_sum = 0, [], None
    for g in graphs:
        b = HitGraphBatch.from_graphs([g]).cuda()
        y = ((torch.arange(b.n_segments, device="cuda") + off) % 3 == 0).float()
        m.zero_grad()
        e = m(b)
        bce(e, y).backward()
        e_parts.append(e.detach().cpu().numpy())
        gs = [p.grad.detach().cpu().double().numpy() for p in m.parameters()]
        g_sum = gs if g_sum is None else [a + c for a, c in zip(g_sum, gs)]
        off += b.n_segments
    assert np.abs(e_all - np.concatenate(e_parts)).max() < 1e-6
    for (k, _), a, r in zip(m.named_parameters(), g_all, g_sum):
        assert_grad_close(a, r, "additivity " + k)


@pytest.mark.parametrize("kind", ["detector graphs (per-pass kernels)", "muon events (one-launch kernels)"])
def test_training_gradients_are_bit_reproducible(hip, kind):
    """deterministic by default.  Every weight-gradient sum runs in a fixed order (one row
    of partial sums per workgroup, rows folded in row order), so two backward passes over the same
    batch give bit-identical gradients - with cross-workgroup float atomics they differed in the last
    bits from run to run."""
    from gnn_fpga_amd.loss import BCELoss
    from gnn_fpga_amd.model import SegmentClassifier
    torch.manual_seed(1)
    if kind.startswith("detector"):
        graphs = [synth.layered_graph(10000, 100000, 3, seed=80 + s) for s in range(6)]
        F = 3
    else:
        graphs = [synth.muon_graph(s) for s in range(512)]
        F = 11
    b = HitGraphBatch.from_graphs(graphs).cuda()
    y = b.y.cuda()
    m = SegmentClassifier(input_dim=F, hidden_dim=8, n_iters=3).cuda().train()

    def three_runs(policy, batch):
        m.level_order_training = policy
        runs = []
        for _ in range(3):
            m.zero_grad()
            loss = BCELoss()(m(batch), y)
            loss.backward()
            runs.append([loss.detach().clone()] + [p.grad.detach().clone() for p in m.parameters()])
        return runs

    fixed = {}
    for policy in (False, True):               # in the caller's order / on the level-ordered twin
        runs = fixed[policy] = three_runs(policy, b)
        for other in runs[1:]:
            for a, c in zip(runs[0], other):
                assert torch.equal(a, c)
        assert all(float(g.abs().max()) > 0 for g in runs[0][1:])
    # the default, "auto": the first step on a batch object runs in the caller's order, every later
    # one on the twin (detector-size batches only) - each bit-identical to the fixed policy's runs
    b2 = HitGraphBatch.from_graphs(graphs).cuda()
    auto = three_runs("auto", b2)
    for a, c in zip(auto[0], fixed[False][0]):
        assert torch.equal(a, c)
    for run in auto[1:]:
        for a, c in zip(run, fixed[True][0]):
            assert torch.equal(a, c)


@pytest.mark.parametrize("kind", ["muon events", "detector graphs", "masked"])
def test_direct_training_step_equals_the_autograd_step(hip, kind):
    """GradBucket.step (no autograd graph, backward adds straight into the bucket) against the
    autograd loop of gnn/estimator.py:49-60 on the same kernels: loss and all ten gradients bit for bit."""
    from gnn_fpga_amd import shard
    from gnn_fpga_amd.loss import BCELoss
    from gnn_fpga_amd.model import SegmentClassifier
    torch.manual_seed(4)
    masks = {}
    if kind == "muon events":
        graphs, F = [synth.muon_graph(s) for s in range(64)], 11
    elif kind == "detector graphs":
        graphs, F = [synth.layered_graph(3000, 20000, 3, seed=s) for s in range(4)], 3
    else:
        graphs, F = [synth.layered_graph(200, 900, 3, seed=s) for s in range(3)], 3
        C = F + 8
        masks = dict(masks_e=[(torch.rand(8, 2 * C) < 0.7).float(), torch.ones(1, 8)],
                     masks_n=[(torch.rand(8, 3 * C) < 0.7).float(), (torch.rand(8, 8) < 0.8).float()])
    b = HitGraphBatch.from_graphs(graphs).cuda()
    y = b.y.cuda()
    m = SegmentClassifier(input_dim=F, hidden_dim=8, n_iters=3, **masks).cuda().train()
    bucket = shard.GradBucket(m.parameters())
    bucket.zero()
    loss = BCELoss(reduction="sum")(m(b), y)
    loss.backward()
    mean_a = bucket.allreduce(loss.detach(), y.numel()).clone()
    grads_a = [p.grad.clone() for p in m.parameters()]
    mean_d = bucket.step(m, b, y)
    assert torch.equal(mean_a, mean_d)
    for (k, p), ga in zip(m.named_parameters(), grads_a):
        assert torch.equal(p.grad, ga), k
    assert not any(p.grad.grad_fn is not None for p in m.parameters())


@pytest.mark.parametrize("D", [8, 32])
def test_training_on_the_level_ordered_twin(hip, D):
    """Detector-size batches train on their level-ordered twin (hits renumbered in plan order: the
    gathers of the training kernels become L2-local): same loss and gradients as in the caller's
    order (golden_util.GRAD_REL of the largest entry: the sums run in another order), bit-reproducible.  D = 32: the
    16-lanes-per-hit kernels of the wide shapes on the twin."""
    from gnn_fpga_amd.loss import BCELoss
    from gnn_fpga_amd.model import SegmentClassifier
    torch.manual_seed(6)
    graphs = [synth.layered_graph(10000, 100000, 3, seed=90 + s) for s in range(4 if D == 8 else 2)]
    b = HitGraphBatch.from_graphs(graphs)
    src, dst = b.src.numpy().copy(), b.dst.numpy().copy()
    src[5::97] = -1                              # padded segments anywhere in the caller's order
    dst[5::97] = -1
    b = HitGraphBatch(b.X.numpy(), src, dst, y=b.y.numpy(), hit_ptr=b.hit_ptr, seg_ptr=b.seg_ptr).cuda()
    y = b.y.cuda()
    m = SegmentClassifier(input_dim=3, hidden_dim=D, n_iters=3 if D == 8 else 2).cuda().train()

    def grads(level_order):
        m.level_order_training = level_order
        m.zero_grad()
        out = m(b)
        loss = BCELoss()(out, y)
        loss.backward()
        return out.detach().clone(), loss.detach().clone(), [p.grad.detach().clone() for p in m.parameters()]

    assert m.level_order_training == "auto"          # the default: from the second use of a batch object
    b2 = HitGraphBatch(b.X.cpu().numpy(), src, dst, y=b.y.cpu().numpy(), hit_ptr=b.hit_ptr, seg_ptr=b.seg_ptr).cuda()
    m.zero_grad(); BCELoss()(m(b2), y).backward()
    assert getattr(b2, "_twin", None) is None
    m.zero_grad(); BCELoss()(m(b2), y).backward()
    assert getattr(b2, "_twin", None) is not None and b2._twin is not b2
    e0, l0, g0 = grads(False)
    twin = b.level_ordered(D)
    assert twin is not b and twin.n_hits == b.plan.n_pad >= b.n_hits       # plan space: padded hit ids
    # the twin's segments are sorted by end hit, padded ones last; seg_order / seg_rank are inverse
    td = twin.dst.long()
    n_valid = int((td >= 0).sum())
    assert bool((td[:n_valid] >= 0).all()) and bool((td[1:n_valid] >= td[:n_valid - 1]).all())
    assert torch.equal(twin.seg_order[twin.seg_rank], torch.arange(b.n_segments, device="cuda"))
    e1, l1, g1 = grads(True)
    e2, l2, g2 = grads(True)
    assert (e0 - e1).abs().max().item() < 1e-6                       # scores stay in the caller's segment order
    assert abs(float(l0) - float(l1)) < 1e-6
    for (k, _), a, c, d in zip(m.named_parameters(), g0, g1, g2):
        assert torch.equal(c, d), k
        assert_grad_close(c, a, "level-ordered twin " + k)


@pytest.mark.parametrize("F,D,T", [(3, 8, 3), (3, 4, 2), (11, 8, 2), (11, 16, 1), (2, 8, 0)])
def test_fused_training_forward_keeps_what_the_per_module_one_keeps(hip, F, D, T, monkeypatch):
    """gnn_segclf_forward_train_plan (ABI 4): the training forward of a plan-space batch on the fused tile kernels
    of the plan it was made from, against gnn_segclf_forward_train (per-module kernels) on the same batch: the
    scores of every pass e_t (valid segments), the hit rows H_t, the node networks' hidden layers Q_t - 2e-6
    (other summation orders; the dummies of the padding included: they are hits without segments) - the final
    scores in the caller's order, and the gradients the backward makes of either set.  Ragged graphs, padded
    segments, a graph smaller than a slice."""
    from gnn_fpga_amd import _lib
    from gnn_fpga_amd.model import SegmentClassifier
    torch.manual_seed(4 * D + T)
    graphs = [synth.layered_graph(2500, 21000, F, seed=60 + i) for i in range(3)] + \
             [synth.layered_graph(9, 11, F, n_layers=3, seed=5), synth.layered_graph(400, 700, F, n_layers=5, seed=6)]
    b = HitGraphBatch.from_graphs(graphs)
    src, dst = b.src.numpy().copy(), b.dst.numpy().copy()
    src[3::41] = -1
    dst[3::41] = -1
    b = HitGraphBatch(b.X.numpy(), src, dst, hit_ptr=b.hit_ptr, seg_ptr=b.seg_ptr).cuda()
    twin = b.level_ordered(D)
    assert twin is not b and twin._fused is b.plan and twin.n_hits == b.plan.n_pad
    m = SegmentClassifier(input_dim=F, hidden_dim=D, n_iters=T).cuda()
    w = [t.detach().contiguous() for t in m.effective_weights()]
    fused = _lib.segclf_forward_train_fused(twin, w, F, D, T)
    assert fused is not None
    e_f, H_f, Q_f, out_f = fused
    e_p, H_p, Q_p = _lib.segclf_forward_train(twin, w, F, D, T)
    valid = (twin.src >= 0)
    assert (e_f[:, valid] - e_p[:, valid]).abs().max().item() < 2e-6
    assert (H_f - H_p).abs().max().item() < 2e-6
    if T:
        assert (Q_f - Q_p).abs().max().item() < 2e-6
    # final scores in the caller's order: the twin's, gathered back
    assert (out_f - e_p[T].index_select(0, twin.seg_rank)).abs().max().item() < 2e-6
    # row T of e_all (k_edge_tw, the twin's order) holds the same bits as e_out (k_edge, the caller's order), padded
    # segments included; without e_out (a loss taken in the twin's order) it is still written
    assert torch.equal(e_f[T], out_f.index_select(0, twin.seg_order))
    e_f2, _, _, none = _lib.segclf_forward_train_fused(twin, w, F, D, T, want_out=False)
    assert none is None and torch.equal(e_f2[T], e_f[T])
    go = torch.rand(b.n_segments, device="cuda")
    g_f = _lib.segclf_backward(twin, w, F, D, T, e_f, H_f, go, Q_all=Q_f)
    g_p = _lib.segclf_backward(twin, w, F, D, T, e_p, H_p, go, Q_all=Q_p)
    for k, (a, r) in enumerate(zip(g_f, g_p)):
        assert_grad_close(a, r, "fused training forward tensor %d" % k)
    # the autograd route takes it for detector-size batches (and GNN_NO_FUSED_TRAIN=1 keeps the per-module one)
    m.train()
    m.level_order_training = True
    with hip.profile(128) as prof:
        out = m(b)
    assert "k_iter" in {k for k, _ in prof.records} or b.n_hits < 20000
    monkeypatch.setenv("GNN_NO_FUSED_TRAIN", "1")
    out2 = m(b)
    assert (out - out2).abs().max().item() < 2e-6


@pytest.mark.parametrize("F,D,T", [(3, 8, 3), (11, 16, 2), (3, 4, 2), (11, 8, 1)])
def test_backward_with_and_without_the_kept_hidden_layers(hip, F, D, T):
    """gnn_segclf_forward_train keeps the node networks' hidden layers (Q_all): the backward then runs
    its four-lanes-per-hit kernels (k_hit_bwd4 / k_seg_bwd4 / k_seg_fin) and needs no second walk over
    the segment lists.  A caller that did not keep them (Q_all = NULL) gets the same gradients from the
    one-lane kernels, which rebuild the sums (1e-5 of the largest entry: other summation orders).
    Ragged lists (lengths not multiples of the quad's 4 entries per step), hits without segments and
    padded segments included; the four-lane route is bit-reproducible."""
    from gnn_fpga_amd.model import SegmentClassifier
    torch.manual_seed(8)
    g = synth.layered_graph(3000, 24000, F, seed=17)
    b = HitGraphBatch.from_graphs([g, synth.layered_graph(40, 31, F, seed=18)])
    src, dst = b.src.numpy().copy(), b.dst.numpy().copy()
    src[5::11] = -1
    dst[5::11] = -1
    b = HitGraphBatch(b.X.numpy(), src, dst, hit_ptr=b.hit_ptr, seg_ptr=b.seg_ptr).cuda()
    m = SegmentClassifier(input_dim=F, hidden_dim=D, n_iters=T).cuda()
    w = [t.detach().contiguous() for t in m.state_dict().values()]
    e_all, H_all, Q_all = hip.segclf_forward_train(b, w, F, D, T)
    assert Q_all.shape == (T, b.n_hits, D) and float(Q_all.abs().max()) <= 1.0
    go = torch.randn(b.n_segments, device="cuda") / b.n_segments
    with_q = hip.segclf_backward(b, w, F, D, T, e_all, H_all, go, Q_all=Q_all)
    again = hip.segclf_backward(b, w, F, D, T, e_all, H_all, go, Q_all=Q_all)
    without = hip.segclf_backward(b, w, F, D, T, e_all, H_all, go)
    for a, a2, c in zip(with_q, again, without):
        assert torch.equal(a, a2)
        assert float(c.abs().max()) > 0
        assert (a - c).abs().max().item() <= 1e-9 + 1e-5 * c.abs().max().item()


TOL_BF16 = 2e-3     # bf16 records and matrix-core operands (GNN_FLAG_BF16_MLP): stated separately from the
                    # fp32 path's 1e-5 ("1e-5 does not apply to bf16"); measured max 6e-4, mean 8e-5


@pytest.mark.parametrize("F,D,T", [(3, 64, 6), (3, 32, 3), (2, 32, 2), (3, 64, 1)])
def test_bf16_matrix_core_hit_update(hip, F, D, T):
    """Opt-in matrix-core path for hidden_dim 32 / 64 (v_mfma_f32_16x16x32_bf16, fp32 accumulate):
    deterministic, really a different path, and within TOL_BF16 of both the fp32 kernels and the
    C oracle (a wrong lane map or k order in the packed weight fragments gives errors of order
    0.1, not 1e-3; tools/mfma_layout_check.hip pins the instruction's maps with exact integers)."""
    from gnn_fpga_amd.model import SegmentClassifier
    torch.manual_seed(D + T)
    graphs = [synth.layered_graph(700, 4000, F, seed=60 + i) for i in range(3)]
    m = SegmentClassifier(input_dim=F, hidden_dim=D, n_iters=T).cuda().eval()
    m.use_events = False
    params = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    batch = HitGraphBatch.from_graphs(graphs).cuda()
    with torch.no_grad():
        e32 = m(batch)
        m.mlp_bf16 = True
        e16 = m(batch)
        e16_again = m(batch)
    torch.cuda.synchronize()
    assert torch.equal(e16, e16_again)                         # deterministic
    assert not torch.equal(e16, e32)                           # the flag really selects another path
    d = (e16 - e32).abs().max().item()
    assert d < TOL_BF16, d
    for g, eg in zip(graphs, batch.split_scores(e16.cpu().numpy())):
        ref = index_c.segment_classifier(g.X, g.src, g.dst, params, T)
        assert np.abs(eg - ref).max() < TOL_BF16


@pytest.mark.parametrize("F,D,T", [(3, 64, 2), (2, 32, 3), (3, 32, 1)])
def test_wide_node_pass_walk_and_matrix_core_mlp(hip, F, D, T, monkeypatch):
    """hidden_dim 32 / 64 at detector size: the node pass as a 16-lanes-per-hit list walk (k_node_walkW)
    plus the MLP of 256 hits as exact fp32 matrix-core products (k_node_mlpW) against the one-lane
    k_node it replaces (GNN_NODE_ONE_LANE): every kept tensor of the training forward within 1e-6
    (k-ordered fp32 fma chains from the bias, like k_node's loops), scores within 1e-5 of the C oracle;
    deterministic; ragged lists, hits without segments, padded segments, a tiny graph in the batch."""
    from gnn_fpga_amd.model import SegmentClassifier
    torch.manual_seed(7 * D + T)
    graphs = [synth.layered_graph(2600, 22000, F, seed=41), synth.layered_graph(7, 6, F, n_layers=3, seed=42)]
    b = HitGraphBatch.from_graphs(graphs)
    src, dst = b.src.numpy().copy(), b.dst.numpy().copy()
    src[3::17] = -1
    dst[3::17] = -1
    b = HitGraphBatch(b.X.numpy(), src, dst, hit_ptr=b.hit_ptr, seg_ptr=b.seg_ptr).cuda()
    m = SegmentClassifier(input_dim=F, hidden_dim=D, n_iters=T).cuda()
    params = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    w = [t.detach().contiguous() for t in m.state_dict().values()]
    monkeypatch.delenv("GNN_NODE_ONE_LANE", raising=False)
    wide = hip.segclf_forward_train(b, w, F, D, T)
    again = hip.segclf_forward_train(b, w, F, D, T)
    monkeypatch.setenv("GNN_NODE_ONE_LANE", "1")
    one = hip.segclf_forward_train(b, w, F, D, T)
    for a, a2, c in zip(wide, again, one):
        assert torch.equal(a, a2)
        assert (a - c).abs().max().item() < 1e-6
    valid = src >= 0
    ref = index_c.segment_classifier(b.X.cpu().numpy(), src[valid], dst[valid], params, T)
    assert np.abs(wide[0][T].cpu().numpy()[valid] - ref).max() < TOL


@pytest.mark.parametrize("F,D,T", [(3, 64, 2), (2, 32, 3), (3, 32, 1)])
def test_wide_backward_on_sixteen_lanes_per_hit(hip, F, D, T, monkeypatch):
    """hidden_dim 32 / 64 (the reference's toy, ACTS and mu200 models): the pull-form backward with 16
    lanes per hit in the list walks (k_hit_bwdW / k_seg_bwdW / k_seg_finW) against the per-pass kernels
    it replaces (GNN_BWD_WIDE_PER_PASS) - all ten gradients within 1e-5 of the largest entry (other
    summation orders), ragged lists, a tiny graph, padded segments; bit-reproducible."""
    from gnn_fpga_amd.model import SegmentClassifier
    torch.manual_seed(5 * D + T)
    b = HitGraphBatch.from_graphs([synth.layered_graph(2500, 21000, F, seed=31), synth.layered_graph(9, 11, F, n_layers=3, seed=32)])
    src, dst = b.src.numpy().copy(), b.dst.numpy().copy()
    src[7::13] = -1
    dst[7::13] = -1
    b = HitGraphBatch(b.X.numpy(), src, dst, hit_ptr=b.hit_ptr, seg_ptr=b.seg_ptr).cuda()
    m = SegmentClassifier(input_dim=F, hidden_dim=D, n_iters=T).cuda()
    w = [t.detach().contiguous() for t in m.state_dict().values()]
    e_all, H_all, Q_all = hip.segclf_forward_train(b, w, F, D, T)
    go = torch.randn(b.n_segments, device="cuda") / b.n_segments
    monkeypatch.delenv("GNN_BWD_WIDE_PER_PASS", raising=False)
    wide = hip.segclf_backward(b, w, F, D, T, e_all, H_all, go, Q_all=Q_all)
    again = hip.segclf_backward(b, w, F, D, T, e_all, H_all, go, Q_all=Q_all)
    monkeypatch.setenv("GNN_BWD_WIDE_PER_PASS", "1")
    per_pass = hip.segclf_backward(b, w, F, D, T, e_all, H_all, go, Q_all=Q_all)
    for a, a2, c in zip(wide, again, per_pass):
        assert torch.equal(a, a2)
        assert float(c.abs().max()) > 0
        assert (a - c).abs().max().item() <= 1e-9 + 1e-5 * c.abs().max().item()


@pytest.mark.parametrize("F,D,T", [(3, 64, 3), (2, 32, 4), (3, 32, 2), (3, 16, 3), (2, 16, 2)])
def test_exact_wide_path_and_its_fallback(hip, F, D, T, monkeypatch):
    """hidden_dim 16 (F <= 4) / 32 / 64 in fp32: the 16-lanes-per-hit kernel with the hit update on
    v_mfma_f32_16x16x4_f32 (a chain of fp32 fmas: nothing is rounded) is the default; the general
    4-lanes-per-hit kernel stays as the fallback for tables beyond 32-bit record offsets
    (GNN_NO_WIDE_EXACT forces it).  Both within the fp32 tolerance of the C oracle, both
    deterministic, and really two different code paths (other summation orders)."""
    from gnn_fpga_amd.model import SegmentClassifier
    torch.manual_seed(3 * D + T)
    graphs = [synth.layered_graph(900, 6000, F, seed=70 + i) for i in range(3)] + [synth.layered_graph(5, 4, F, n_layers=2, seed=9)]
    m = SegmentClassifier(input_dim=F, hidden_dim=D, n_iters=T).cuda().eval()
    m.use_events = False
    params = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    batch = HitGraphBatch.from_graphs(graphs).cuda()
    with torch.no_grad():
        monkeypatch.delenv("GNN_NO_WIDE_EXACT", raising=False)
        wide, wide_again = m(batch), m(batch)
        monkeypatch.setenv("GNN_NO_WIDE_EXACT", "1")
        general = m(batch)
    torch.cuda.synchronize()
    assert torch.equal(wide, wide_again)
    assert not torch.equal(wide, general)
    assert (wide - general).abs().max().item() < TOL
    for g, ew, eg in zip(graphs, batch.split_scores(wide.cpu().numpy()), batch.split_scores(general.cpu().numpy())):
        ref = index_c.segment_classifier(g.X, g.src, g.dst, params, T)
        assert np.abs(ew - ref).max() < TOL and np.abs(eg - ref).max() < TOL


@pytest.mark.parametrize("F,D,T,bf16", [(3, 64, 3, False), (3, 64, 2, True), (2, 32, 3, False), (3, 32, 2, True), (3, 16, 3, False)])
def test_role_split_wide_kernel_equals_the_barrier_kernel(hip, F, D, T, bf16, monkeypatch):
    """k_iter_wx (sweep waves + matrix-core waves, LDS ring and counters instead of workgroup barriers:
    the default for hidden_dim 16 / 32 / 64) against k_iter_w (every round behind barriers;
    GNN_WIDE_LOCKSTEP=1): the same sweeps in the same order and the same k-ordered products, so the
    scores must be BIT-identical - on ragged graphs, graphs smaller than a slice, many more slices than
    ring slots, repeated (a lost or doubled slot hand-off would show), exact fp32 and bf16 records."""
    from gnn_fpga_amd.model import SegmentClassifier
    torch.manual_seed(11 * D + T)
    graphs = ([synth.layered_graph(2500, 20000, F, seed=40 + i) for i in range(5)] +
              [synth.layered_graph(7, 9, F, n_layers=3, seed=3), synth.layered_graph(300, 2900, F, n_layers=4, seed=4)])
    m = SegmentClassifier(input_dim=F, hidden_dim=D, n_iters=T).cuda().eval()
    m.use_events = False
    m.mlp_bf16 = bf16
    batch = HitGraphBatch.from_graphs(graphs).cuda()
    with torch.no_grad():
        monkeypatch.delenv("GNN_WIDE_LOCKSTEP", raising=False)
        monkeypatch.setenv("GNN_WIDE_ROLES", "1")          # (batches below 32k hits take the barrier kernel by default)
        with hip.profile(64) as prof:
            a = m(batch).clone()
        assert "k_iter_wx" in {k for k, _ in prof.records}
        again = [m(batch).clone() for _ in range(3)]
        monkeypatch.delenv("GNN_WIDE_ROLES")
        with hip.profile(64) as prof:
            b = m(batch).clone()                               # 13 k hits: the default here is the barrier kernel
        assert "k_iter_w" in {k for k, _ in prof.records} and "k_iter_wx" not in {k for k, _ in prof.records}
        monkeypatch.setenv("GNN_WIDE_LOCKSTEP", "1")
        assert torch.equal(b, m(batch))
    torch.cuda.synchronize()
    assert all(torch.equal(a, x) for x in again)
    assert torch.equal(a, b)


@pytest.mark.parametrize("lim_over", [{}, {"iter_records": 0, "edge_records": 0}])
def test_plan_built_on_the_gpu_equals_the_host_plan(hip, lim_over):
    """HitGraphBatch.build_plan on a CUDA batch runs plan_device.DeviceSellPlan (torch sorts and
    scatters on the GPU): every array and scalar must equal the numpy builder's (plan.SellPlan) -
    the CPU suite checks the same code on CPU tensors, this checks the CUDA sort / scatter / argmin
    semantics it relies on (stability, first minimum)."""
    from test_plan import _same_plan
    from gnn_fpga_amd import _lib
    from gnn_fpga_amd.plan import SellPlan
    from gnn_fpga_amd.plan_device import DeviceSellPlan
    graphs = [synth.layered_graph(3000, 30000, 3, seed=300 + s) for s in range(12)]
    graphs.append(synth.layered_graph(2, 1, 3, n_layers=2, seed=1))
    b = HitGraphBatch.from_graphs(graphs)
    src, dst = b.src.numpy().copy(), b.dst.numpy().copy()
    src[5::11] = -1
    dst[5::11] = -1
    b = HitGraphBatch(b.X.numpy(), src, dst, hit_ptr=b.hit_ptr, seg_ptr=b.seg_ptr)
    lim = _lib.plan_limits(3, 8)
    lim.update(lim_over)
    host = SellPlan(b, lim)
    dev = DeviceSellPlan(b.cuda(), lim)
    assert dev.X.is_cuda
    _same_plan(host, dev)
    import os
    os.environ["GNN_PLAN_BUILDER"] = "torch"
    try:
        assert isinstance(b.cuda().build_plan(8), DeviceSellPlan)
    finally:
        del os.environ["GNN_PLAN_BUILDER"]


def _host_and_device_twins(graphs, knock_out=13):
    b = HitGraphBatch.from_graphs(graphs)
    src, dst = b.src.numpy().copy(), b.dst.numpy().copy()
    if knock_out:
        src[3::knock_out] = -1
        dst[3::knock_out] = -1
    host = HitGraphBatch(b.X.numpy(), src, dst, hit_ptr=b.hit_ptr, seg_ptr=b.seg_ptr)
    dev = HitGraphBatch(b.X.numpy(), src, dst, hit_ptr=b.hit_ptr, seg_ptr=b.seg_ptr).cuda()
    return host, dev


@pytest.mark.parametrize("builder", ["hip", "torch"])
@pytest.mark.parametrize("case", ["ragged", "one_graph", "many_hits", "no_segments", "all_padded", "hub", "big_batch",
                                  "big_wide"])
def test_csr_built_on_the_gpu_equals_the_host_csr(hip, builder, case, monkeypatch):
    """The two segment lists of the per-module / training kernels are built where the batch lives - by
    gnn_csr_build (counting sort + rank: csrc/csr_build.hip) or, GNN_CSR_BUILDER=torch, by stable torch sorts - and
    must be the host version's arrays entry for entry (= the reference's on-disk order, `Ri.nonzero()` row-major,
    gnn/graph.py:20-26): padded segments (src = dst = -1) left out of both; the HIP builder keeps n_segments
    entries per array, -1 past the lists.  Cases: ragged graphs, one graph, more hits than one scan workgroup takes
    (block sums path), a batch without segments, only padded segments, one hit with 3000 segments, and batches of
    more than a million segments (LDS-private counting) with narrow and with wide endpoint ranges."""
    monkeypatch.setenv("GNN_CSR_BUILDER", builder)
    if case == "ragged":
        graphs = [synth.layered_graph(2000, 15000, 3, seed=400 + s) for s in range(6)] + \
                 [synth.layered_graph(5, 4, 3, n_layers=2, seed=1), synth.layered_graph(30, 0, 3, n_layers=3, seed=2)]
        host, dev = _host_and_device_twins(graphs)
    elif case == "one_graph":
        host, dev = _host_and_device_twins([synth.layered_graph(10000, 100000, 3, seed=7)], knock_out=0)
    elif case == "many_hits":
        host, dev = _host_and_device_twins([synth.layered_graph(300000, 400000, 2, seed=8)], knock_out=17)
    elif case == "big_batch":        # >= 1 M segments: LDS-private counting per 16 k segments (k_csr_*_wg), narrow ranges
        host, dev = _host_and_device_twins([synth.layered_graph(9000 + 100 * s, 90000 + 1500 * s, 3, seed=70 + s)
                                            for s in range(13)], knock_out=29)
    elif case == "big_wide":         # the same kernels where a workgroup's endpoints span > 16 k hits: global claims
        host, dev = _host_and_device_twins([synth.layered_graph(200000, 1100000, 2, seed=9)], knock_out=31)
    elif case == "no_segments":
        host, dev = _host_and_device_twins([synth.layered_graph(40, 0, 3, n_layers=3, seed=3)], knock_out=0)
    elif case == "all_padded":
        g = synth.layered_graph(50, 60, 3, n_layers=4, seed=4)
        X = np.asarray(g.X, dtype=np.float32)
        pad = np.full(60, -1, dtype=np.int32)
        host, dev = HitGraphBatch(X, pad, pad), HitGraphBatch(X, pad, pad).cuda()
    else:
        rng = np.random.default_rng(5)
        X = rng.standard_normal((4000, 3)).astype(np.float32)
        src = np.concatenate([np.full(3000, 17), rng.integers(0, 4000, 5000)]).astype(np.int32)
        dst = np.concatenate([rng.integers(0, 4000, 3000), np.full(2500, 99), rng.integers(0, 4000, 2500)]).astype(np.int32)
        o = rng.permutation(8000)
        host, dev = HitGraphBatch(X, src[o], dst[o]), HitGraphBatch(X, src[o], dst[o]).cuda()
    n_valid = int(host.in_ptr[-1])
    for name in HitGraphBatch._CSR_NAMES:
        a, c = getattr(host, name), getattr(dev, name)
        assert c.is_cuda and a.dtype == c.dtype, name
        c = c.cpu()
        if name.endswith("ptr") or builder == "torch":
            assert torch.equal(a, c), name
        else:
            assert c.numel() == host.n_segments and torch.equal(a, c[:n_valid]) and bool((c[n_valid:] == -1).all()), name


def test_csr_builder_reports_malformed_endpoints(hip):
    """gnn_csr_build skips a segment with an end outside [0, n_hits) or with exactly one negative end (no
    out-of-range atomic or store) and says so in its status word; HitGraphBatch.validate_csr = True turns that into
    the ValueError the host constructor raises for the same arrays."""
    from gnn_fpga_amd import _lib
    g = synth.layered_graph(500, 3000, 3, seed=11)
    b = HitGraphBatch.from_graphs([g]).cuda()
    good = _lib.csr_build(b.src, b.dst, b.n_hits)
    assert int(good[6].item()) == 0
    src, dst = b.src.clone(), b.dst.clone()
    src[5] = 500                # one past the last hit
    dst[9] = -1                 # exactly one end negative
    src[11] = dst[11] = -1      # a padded segment: fine
    bad = _lib.csr_build(src, dst, b.n_hits)
    assert int(bad[6].item()) & 1
    assert int(bad[0][-1]) == int(bad[3][-1]) == 3000 - 3            # the three segments are in neither list
    eids = set(bad[1][:2997].tolist())
    assert not eids & {5, 9, 11} and len(eids) == 2997
    b2 = HitGraphBatch.from_graphs([g]).cuda()
    b2.src, b2.validate_csr = src, True
    with pytest.raises(ValueError):
        b2.in_ptr


def test_first_forward_of_a_never_seen_batch_needs_no_plan(hip):
    """model.use_plan = "auto" (default): the FIRST inference forward of a never-seen detector-size batch runs the
    per-module kernels on gnn_csr_build's lists (no plan: trigger-style use, gnn/Inference.ipynb cell 3), the second
    forward of the same batch builds the plan and runs the fused tile kernels, a batch that brings a plan runs them
    at once; all three agree within the score tolerance."""
    from gnn_fpga_amd.model import SegmentClassifier
    torch.manual_seed(3)
    m = SegmentClassifier(input_dim=3, hidden_dim=8, n_iters=3).cuda().eval()
    import conftest
    assert conftest.DEFAULT_USE_PLAN == "auto"          # what a user gets (the test session runs with True)
    m.use_plan = "auto"
    g = synth.layered_graph(10000, 100000, 3, seed=21)
    b = HitGraphBatch.from_graphs([g]).cuda()
    with torch.no_grad():
        with hip.profile(64) as p1:
            e1 = m(b)
        names1 = {k for k, _ in p1.records}
        assert b.plan is None and "k_node" in names1 and "k_csr_rank" in names1 and not any("k_iter" in k for k in names1)
        with hip.profile(64) as p2:
            e2 = m(b)
        names2 = {k for k, _ in p2.records}
        assert b.plan is not None and any("k_iter" in k for k in names2) and "k_node" not in names2
        b3 = HitGraphBatch.from_graphs([g]).cuda()
        b3.build_plan(8)
        with hip.profile(64) as p3:
            e3 = m(b3)
        assert any("k_iter" in k for k, _ in p3.records)
    assert (e1 - e2).abs().max().item() < 1e-5 and torch.equal(e2, e3)
    m.use_plan = True
    b4 = HitGraphBatch.from_graphs([g]).cuda()
    with torch.no_grad():
        assert torch.equal(m(b4), e2) and b4.plan is not None


@pytest.mark.parametrize("F,D,T", [(11, 8, 3), (3, 8, 2), (2, 16, 1), (3, 4, 0)])
def test_one_launch_backward_for_small_events(hip, F, D, T):
    """gnn_segclf_backward_events (one workgroup per graph, everything in LDS) against the per-pass
    backward kernels on the same saved tensors: all ten gradients, padded segments, isolated hits,
    graphs without segments."""
    from gnn_fpga_amd import _lib
    from gnn_fpga_amd.loss import BCELoss
    from gnn_fpga_amd.model import SegmentClassifier
    torch.manual_seed(F + D + T)
    if F == 11:
        graphs = [synth.muon_graph(s) for s in range(40)]
    elif T == 2:        # close to what one workgroup's LDS holds (111 of 128 KB)
        graphs = [synth.layered_graph(150, 600, F, n_layers=6, seed=520 + i) for i in range(6)]
        graphs.append(synth.layered_graph(7, 9, F, n_layers=3, seed=9))
    else:
        graphs = [synth.layered_graph(n, e, F, n_layers=L, seed=500 + i)
                  for i, (n, e, L) in enumerate([(60, 200, 6), (3, 2, 2), (40, 0, 4), (25, 90, 5), (2, 1, 2)])]
    b = HitGraphBatch.from_graphs(graphs)
    src, dst = b.src.numpy().copy(), b.dst.numpy().copy()
    src[1::9] = -1
    dst[1::9] = -1
    b = HitGraphBatch(b.X.numpy(), src, dst, hit_ptr=b.hit_ptr, seg_ptr=b.seg_ptr).cuda()
    lay = b.event_layout()
    assert lay is not None and _lib.events_backward_supported(F, D, lay.max_hits, lay.max_segments)
    y = (torch.arange(b.n_segments, device="cuda") % 2 == 0).float()
    m = SegmentClassifier(input_dim=F, hidden_dim=D, n_iters=T).cuda().train()
    out, calls = {}, []
    real = _lib.segclf_backward_events
    _lib.segclf_backward_events = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        for events in (True, False):
            m.use_events = events
            m.zero_grad()
            BCELoss()(m(b), y).backward()
            out[events] = [p.grad.detach().cpu().double().numpy().copy() for p in m.parameters()]
    finally:
        _lib.segclf_backward_events = real
    assert len(calls) == 1                      # the one-launch path ran for use_events = True only
    for (k, _), a, r in zip(m.named_parameters(), out[True], out[False]):
        assert_grad_close(a, r, "one-launch vs per-pass " + k)
    big = HitGraphBatch.from_graphs([synth.layered_graph(20000, 100000, 3, seed=1)])
    assert big.event_layout() is None          # beyond EVENTS_MAX_SEGMENTS: no layout (and no host endpoint check)
    assert not _lib.events_backward_supported(3, 8, 20000, 100000)


def test_one_launch_training_forward_keeps_the_same_tensors(hip):
    """gnn_segclf_forward_train_events: the small-event kernel also stores every pass's scores and
    hit rows; they must equal the per-pass training forward's bit for bit (padding columns too)."""
    from gnn_fpga_amd import _lib
    from gnn_fpga_amd.model import SegmentClassifier
    torch.manual_seed(5)
    graphs = [synth.muon_graph(s) for s in range(30)]
    b = HitGraphBatch.from_graphs(graphs)
    src, dst = b.src.numpy().copy(), b.dst.numpy().copy()
    src[2::7] = -1
    dst[2::7] = -1
    b = HitGraphBatch(b.X.numpy(), src, dst, hit_ptr=b.hit_ptr, seg_ptr=b.seg_ptr).cuda()
    m = SegmentClassifier(input_dim=11, hidden_dim=8, n_iters=3).cuda()
    w = [t.detach().contiguous() for t in m.state_dict().values()]
    e1, H1, Q1 = _lib.segclf_forward_train(b, w, 11, 8, 3)
    e2, H2, Q2 = _lib.segclf_forward_train(b, w, 11, 8, 3, layout=b.event_layout())
    assert Q1.shape == (3, b.n_hits, 8) and Q2.numel() == 0
    assert torch.equal(e1, e2) and torch.equal(H1, H2)


def test_model_deep_copies_after_a_forward(hip):
    """gnn/estimator_maskedlinear.py:83 deep-copies the model it holds: after forwards on every route (caches hold
    a workspace, a packed-parameter struct of device pointers, an exp-product decision) the copy must come out with
    its own parameters, no caches, and the same scores; editing the copy leaves the original alone."""
    import copy
    from gnn_fpga_amd.model import SegmentClassifier
    torch.manual_seed(5)
    m = SegmentClassifier(input_dim=3, hidden_dim=8, n_iters=2).cuda().eval()
    b = HitGraphBatch.from_graphs([synth.layered_graph(3000, 20000, 3, seed=31)]).cuda()
    with torch.no_grad():
        e = m(b)
        assert m._w_cache is not None and m._workspace is not None
        c = copy.deepcopy(m)
        assert c._w_cache is None and c._workspace is None
        assert all(a.data_ptr() != p.data_ptr() for a, p in zip(c.parameters(), m.parameters()))
        assert torch.equal(c(b), e)
        for p in c.parameters():
            p.mul_(0.5)
        assert not torch.equal(c(b), e) and torch.equal(m(b), e)


@pytest.mark.parametrize("route", ["events", "plan", "modules"])
def test_replaced_parameters_are_seen_like_in_the_pruning_notebooks(hip, route):
    """gnn/MPNN_Seg_ACTS_maskedlinear.ipynb cell 34 / MPNN_Seg_ACTS.ipynb cell 33 prune by REPLACING the layers'
    Parameter objects (`model.edge_network.network[0].weight = torch.nn.Parameter(weight * mask)`, all four layers)
    between two evaluations: the inference caches (packed weights, exp-product decision) must follow - scores of the
    second forward against the oracle on the pruned weights, on every route."""
    from gnn_fpga_amd.model import SegmentClassifier
    torch.manual_seed(11)
    if route == "events":
        graphs = [synth.layered_graph(60, 220, 3, n_layers=5, seed=80 + s) for s in range(6)]
    else:
        graphs = [synth.layered_graph(1500, 9000, 3, seed=80 + s) for s in range(2)]
    b = HitGraphBatch.from_graphs(graphs).cuda()
    m = SegmentClassifier(input_dim=3, hidden_dim=8, n_iters=2).cuda().eval()
    m.use_events, m.use_plan = route == "events", route != "modules"
    with torch.no_grad():
        e0 = m(b)
        for net in (m.edge_network, m.node_network):
            for i in (0, 2):
                w = net.network[i].weight
                keep = (w.abs() > 0.5 * w.abs().mean()).float()      # |W| > threshold, the notebooks' rule (cell 21)
                net.network[i].weight = torch.nn.Parameter(w * keep)
        e1 = m(b)
    assert not torch.equal(e0, e1)
    params = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    for g, got in zip(graphs, b.split_scores(e1.cpu().numpy())):
        ref = index_c.segment_classifier(g.X, g.src, g.dst, params, 2)
        assert np.abs(got - ref).max() < TOL


@pytest.mark.parametrize("F,D,T", [(11, 8, 3), (3, 8, 2), (2, 16, 1), (3, 4, 3)])
def test_event_kernel_builds_the_lists_of_a_never_seen_batch_itself(hip, F, D, T):
    """gnn_segclf_forward_events with all six list pointers NULL (what the model passes for a batch whose lists nobody
    has asked for): `k_event` builds them in LDS from (src, dst) - counts, scan, arrival-order slots, rank - and the
    scores must be the SAME BITS as with gnn_csr_build's lists (same lists, same summation order), twice in a row
    (no dependence on the order the LDS atomics arrive in); one graph also without offset arrays.  Muon events, graphs
    close to the kernel's LDS limit, padded segments, isolated hits, a graph without segments, a 300-segment hub."""
    from gnn_fpga_amd import _lib
    from gnn_fpga_amd.model import SegmentClassifier
    torch.manual_seed(F * D + T)
    rng = np.random.default_rng(F + D)
    if F == 11:
        sets = [[synth.muon_graph(s) for s in range(64)], [synth.muon_graph(7)]]
    else:
        hub_src = np.concatenate([np.full(300, 5), rng.integers(0, 90, 200)]).astype(np.int32)
        hub_dst = np.concatenate([rng.integers(0, 90, 300), np.full(150, 17), rng.integers(0, 90, 50)]).astype(np.int32)
        hub = synth.HitGraph(rng.standard_normal((90, F)).astype(np.float32), hub_src, hub_dst,
                             np.zeros(500, np.float32))
        sets = [[synth.layered_graph(n, e, F, n_layers=L, seed=600 + i)
                 for i, (n, e, L) in enumerate([(60, 200, 6), (3, 2, 2), (40, 0, 4), (150, 1100, 5), (2, 1, 2)])] + [hub],
                [synth.layered_graph(140, 900, F, n_layers=6, seed=77)]]
    m = SegmentClassifier(input_dim=F, hidden_dim=D, n_iters=T).cuda().eval()
    w = [t.detach().contiguous() for t in m.effective_weights()]
    for graphs in sets:
        b = HitGraphBatch.from_graphs(graphs)
        src, dst = b.src.numpy().copy(), b.dst.numpy().copy()
        src[2::7] = -1
        dst[2::7] = -1
        mk = lambda: HitGraphBatch(b.X.numpy(), src, dst, hit_ptr=b.hit_ptr, seg_ptr=b.seg_ptr).cuda()  # noqa: E731
        fresh, listed = mk(), mk()
        lay = fresh.event_layout()
        assert _lib.events_preferred(F, D, lay)
        listed.in_ptr                                        # lists by gnn_csr_build
        assert fresh._csr is None and listed._csr is not None
        with hip.profile(16) as prof:
            e_raw = _lib.segclf_forward_events(fresh, lay, w, F, D, T)
        assert [k for k, _ in prof.records] == ["k_event"] and fresh._csr is None      # one launch, nothing else
        e_raw2 = _lib.segclf_forward_events(fresh, lay, w, F, D, T)
        e_listed = _lib.segclf_forward_events(listed, listed.event_layout(), w, F, D, T)
        assert torch.equal(e_raw, e_listed) and torch.equal(e_raw, e_raw2)
        # and through the model: the first forward of a never-seen batch of small graphs is that one launch
        with torch.no_grad(), hip.profile(16) as prof:
            e_model = m(mk())
        assert [k for k, _ in prof.records] == ["k_event"] and torch.equal(e_model.reshape(-1), e_raw)


def test_new_features_on_a_planned_batch_need_no_new_plan(hip):
    """HitGraphBatch.with_features: other hit features on the same graphs reuse the plan's structure (VERDICT r2 item
    4) - scores are the bits a from-scratch batch gives, no plan-builder kernel runs, the exp-product decision is
    taken again for the new feature range (|X| x 400 makes the proven bound fail: exact kernels)."""
    from gnn_fpga_amd.model import SegmentClassifier
    torch.manual_seed(2)
    graphs = [synth.layered_graph(4000, 30000, 3, seed=40 + s) for s in range(3)]
    b = HitGraphBatch.from_graphs(graphs).cuda()
    m = SegmentClassifier(input_dim=3, hidden_dim=8, n_iters=3).cuda().eval()
    m.use_events = False
    with torch.no_grad():
        m(b)
        assert m._xp_cache[1] == hip.GNN_FLAG_EXP_PRODUCT
        for scale, want in ((0.5, hip.GNN_FLAG_EXP_PRODUCT), (400.0, 0)):
            X2 = (torch.randn_like(b.X) * scale).contiguous()
            with hip.profile(256) as prof:
                b2 = b.with_features(X2)
                e2 = m(b2)
            assert not any(k.startswith("pb_") for k, _ in prof.records)          # no plan build
            assert m._xp_cache[1] == want
            ref = HitGraphBatch(X2.cpu().numpy(), b.src.cpu().numpy(), b.dst.cpu().numpy(),
                                hit_ptr=b.hit_ptr, seg_ptr=b.seg_ptr).cuda()
            assert torch.equal(e2, m(ref))

"""The reference's own training harness holding the drop-in (build container only: imports
/root/reference/gnn/estimator.py UNMODIFIED; skipped where the reference is absent, i.e. on the GPU
box).  gnn/estimator.py:22-47 (constructor: optimizer over model.parameters()), :54-55 (the weight
lists its L1 penalty walks), :62-78 (checkpoint save / load of state_dict + optimizer).  The forward
itself needs a GPU (no CPU path): on CPU tensors the step must fail loudly, not fall back."""
import contextlib
import io
import os
import sys

import pytest
import torch

REF = "/root/reference/gnn"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")


@pytest.fixture()
def ref():
    sys.dont_write_bytecode = True            # never write __pycache__ into the read-only reference
    sys.path.insert(0, REF)
    try:
        import estimator as ref_estimator
        import model as ref_model
        yield ref_estimator, ref_model
    finally:
        sys.path.remove(REF)


def test_reference_estimator_holds_the_dropin(ref, tmp_path):
    ref_estimator, ref_model = ref
    from gnn_fpga_amd import HitGraphBatch, synth
    from gnn_fpga_amd._lib import GnnHipError
    from gnn_fpga_amd.model import SegmentClassifier
    torch.manual_seed(0)
    m = SegmentClassifier(input_dim=3, hidden_dim=8, n_iters=2)
    with contextlib.redirect_stdout(io.StringIO()) as out:      # the constructor prints the model
        est = ref_estimator.Estimator(m, torch.nn.BCELoss(), opt="Adam", cuda=False, l1=1e-4)
    assert "Parameters: 569" in out.getvalue()                  # the notebooks' printed count (F=3, D=8)
    assert sum(len(g["params"]) for g in est.optimizer.param_groups) == 10
    # the lists training_step builds for its L1 penalty (estimator.py:54-55)
    node_w = [l.weight for l in est.model.node_network.network if hasattr(l, "weight")]
    edge_w = [l.weight for l in est.model.edge_network.network if hasattr(l, "weight")]
    assert [tuple(w.shape) for w in node_w] == [(8, 33), (8, 8)]
    assert [tuple(w.shape) for w in edge_w] == [(8, 22), (1, 8)]
    # same state_dict keys / shapes as the reference's own model
    torch.manual_seed(0)
    r = ref_model.SegmentClassifier(input_dim=3, hidden_dim=8, n_iters=2,
                                    masks_n=[torch.ones(8, 33), torch.ones(8, 8)])
    assert {k: tuple(v.shape) for k, v in r.state_dict().items()} == \
           {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(r.state_dict())                           # a reference checkpoint loads
    # checkpoint round trip through the reference's own save / load (estimator.py:62-78,128-135)
    fn = str(tmp_path / "ckpt" / "checkpoint.pt")
    est.save_checkpoint({"epoch": 1, "state_dict": est.model.state_dict(), "best_valid_loss": 0.5,
                         "valid_losses": [0.5], "train_losses": [0.6],
                         "optimizer": est.optimizer.state_dict()}, True, filename=fn)
    assert os.path.exists(str(tmp_path / "ckpt" / "model_best.pt"))
    with torch.no_grad():
        for p in m.parameters():
            p.add_(1.0)
    est.load_checkpoint(fn)
    for k, v in r.state_dict().items():
        assert torch.equal(m.state_dict()[k], v), k
    assert est.train_losses == [0.6] and est.valid_losses == [0.5]
    # no CPU path: the step raises instead of silently computing somewhere else
    g = synth.layered_graph(30, 60, 3, seed=0)
    X, Ri, Ro = (torch.from_numpy(a)[None] for a in synth.to_dense(g))
    with pytest.raises(GnnHipError):
        est.training_step([X, Ri, Ro], torch.from_numpy(g.y)[None])
    with pytest.raises(GnnHipError):
        est.model(HitGraphBatch.from_graphs([g]))


def test_reference_pruning_estimator_deep_copies_and_reloads_the_dropin(tmp_path):
    """gnn/estimator_maskedlinear.py:82-101 `load_weights` (the pruning notebooks' retraining step) deep-copies the
    model, loads a checkpoint into the copy and assigns `weight.data = W * mask.data` layer by layer.  The drop-in must
    survive that AFTER it has run a forward - its caches then hold a ctypes struct of device pointers, which neither
    pickles nor deep-copies (simulated here: the build container has no GPU) - and must see the replaced `.data`."""
    import copy
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    try:
        import estimator_maskedlinear as ref_pruning
    finally:
        sys.path.remove(REF)
    from gnn_fpga_amd import _lib
    from gnn_fpga_amd.model import SegmentClassifier
    torch.manual_seed(1)
    D, C = 8, 11
    me = [(torch.rand(D, 2 * C) > 0.4).float(), torch.ones(1, D)]
    mn = [(torch.rand(D, 3 * C) > 0.4).float(), (torch.rand(D, D) > 0.4).float()]
    m = SegmentClassifier(input_dim=3, hidden_dim=D, n_iters=2, masks_e=me, masks_n=mn)
    with contextlib.redirect_stdout(io.StringIO()):
        est = ref_pruning.Estimator(m, torch.nn.BCELoss(), opt="Adam", cuda=False)
    # what a forward on the GPU leaves behind
    key = m._param_key()
    m._w_cache = (key, [p.detach() for p in m.parameters()], _lib.GnnParams(), D, None)
    m._xp_cache = (key, 0)
    c = copy.deepcopy(m)
    assert c._w_cache is None and c._xp_cache is None and m._w_cache is not None
    assert all(torch.equal(a, b) and a.data_ptr() != b.data_ptr() for a, b in zip(m.parameters(), c.parameters()))
    # a dense checkpoint (trained without masks), reloaded into the masked model through the reference's own code
    torch.manual_seed(2)
    dense = SegmentClassifier(input_dim=3, hidden_dim=D, n_iters=2)
    fn = str(tmp_path / "dense.pt")
    torch.save({"state_dict": dense.state_dict()}, fn)
    with contextlib.redirect_stdout(io.StringIO()):
        est.load_weights(fn)
    sd = dense.state_dict()
    assert torch.equal(m.edge_network.network[0].weight.data, sd["edge_network.network.0.weight"] * me[0])
    assert torch.equal(m.node_network.network[0].weight.data, sd["node_network.network.0.weight"] * mn[0])
    assert torch.equal(m.node_network.network[2].weight.data, sd["node_network.network.2.weight"] * mn[1])
    assert m._param_key() != key                      # the replaced .data is seen: the cached weights are rebuilt
    # torch.save(model) / torch.load of the whole module, as notebooks do, drops the caches too
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    again = torch.load(buf, weights_only=False)       # (a file this test wrote itself)
    assert again._w_cache is None and all(torch.equal(a, b) for a, b in zip(m.parameters(), again.parameters()))

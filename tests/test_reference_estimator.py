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

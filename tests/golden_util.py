"""Helpers to read tests/golden/*.npz (written by oracle/gen_golden.py from the reference)."""
import glob
import os

import numpy as np

from gnn_fpga_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Gradient parity bound: |HIP - reference| <= GRAD_ABS + GRAD_REL * max|reference| per tensor.
# Measured over the whole -m gpu suite on MI355X (GNN_TEST_RECORD=<file> writes every comparison's
# error and max|reference| there; 217 comparisons): largest error / max|reference| 2.0e-6 (fp32 sums
# in another order than autograd's; tiny gradients: 1.1e-6 beyond an absolute 1e-8).  The bound is
# 2.5x that - rounds 1-2 allowed 1e-4.
GRAD_REL = 5e-6
GRAD_ABS = 1e-8


def assert_grad_close(got, ref, what, rel=GRAD_REL, abs_=GRAD_ABS):
    """got, ref: arrays (numpy or torch) of one gradient tensor."""
    g = np.asarray(got.detach().cpu().double().numpy() if hasattr(got, "detach") else got, dtype=np.float64)
    r = np.asarray(ref.detach().cpu().double().numpy() if hasattr(ref, "detach") else ref, dtype=np.float64)
    err = float(np.abs(g - r).max()) if g.size else 0.0
    scale = float(np.abs(r).max()) if r.size else 0.0
    rec = os.environ.get("GNN_TEST_RECORD")
    if rec:
        with open(rec, "a") as f:
            f.write("%s\t%.3e\t%.3e\t%.3e\n" % (what, err, scale, err / scale if scale else 0.0))
    assert err < abs_ + rel * scale, (what, err, scale)

_ALL = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz")))
# refnpz_*: scores for the files in ref_written/ (written by the reference's own save_graph);
# batchgen_*: the reference batch generator's batches - both have their own tests
SINGLE = [n for n in _ALL if "batch" not in n and "c3_full" not in n and not n.startswith("refnpz_")]
BATCHES = [n for n in _ALL if "batch" in n and not n.startswith("batchgen_")]
PRUNED = [n for n in _ALL if n.startswith("pruned_")]
REF_WRITTEN = os.path.join(GOLDEN, "ref_written")


class Fixture:
    def __init__(self, name):
        with np.load(os.path.join(GOLDEN, name + ".npz")) as f:
            d = {k: f[k] for k in f.files}
        self.name = name
        self.raw = d
        self.params = {k[2:]: v for k, v in d.items() if k.startswith("p.")}
        m = {k[2:]: v for k, v in d.items() if k.startswith("m.")}
        self.masks = m or None
        self.n_iters = int(d["n_iters"])
        self.scores = d["scores"]
        self.e_trace = d.get("e_trace")
        self.H_trace = d.get("H_trace")
        self.D = self.params["input_network.0.weight"].shape[0]
        self.F = self.params["input_network.0.weight"].shape[1]
        if "X" in d:
            self.graph = synth.HitGraph(d["X"], d["src"], d["dst"], None)
        elif "gen" in d:   # c3_full: inputs regenerated from the recorded seed
            self.graph = synth.layered_graph(10000, 100000, 3, seed=0)
        if "n_graphs" in d:
            self.graphs = [synth.HitGraph(d["g%d.X" % i], d["g%d.src" % i], d["g%d.dst" % i], None)
                           for i in range(int(d["n_graphs"]))]
            self.y = d["y"]
            self.loss = float(d["loss"])
            self.grads = {k[2:]: v for k, v in d.items() if k.startswith("g.")}

    def effective_params(self):
        """W*mask where a mask exists (reference gnn/model.py:30)."""
        p = dict(self.params)
        if self.masks:
            for k, m in self.masks.items():
                p[k] = p[k] * m
        return p

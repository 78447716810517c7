"""Helpers to read tests/golden/*.npz (written by oracle/gen_golden.py from the reference)."""
import glob
import os

import numpy as np

from gnn_fpga_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

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

"""ORACLE (test infrastructure): generate tests/golden/*.npz FROM THE REFERENCE ITSELF.

Run in the build container only (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [--with-c3-full]

It imports /root/reference/gnn/model.py and estimator.py unmodified, runs
`SegmentClassifier.forward` (gnn/model.py:140-156) on seeded synthetic inputs,
captures per-iteration edge scores and hit features with forward hooks (no reference
modification), and writes inputs + state_dict + expected outputs as small .npz files.
Fixtures are data only: arrays, no source text.

torch version at generation time is recorded in every file (`torch_version`).
"""
import argparse
import os
import sys

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference/gnn")

import numpy as np  # noqa: E402
import torch  # noqa: E402

import model as ref_model  # noqa: E402  (the reference)
import estimator as ref_estimator  # noqa: E402  (the reference)
from gnn_fpga_amd import synth  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")


def make_reference(F, D, T, seed, masked=False, scale=1.0):
    """Build the reference model; all-ones node masks stand in for 'no mask'
    (gnn/model.py:100 requires masks_n)."""
    torch.manual_seed(seed)
    C = D + F
    if masked:
        g = torch.Generator().manual_seed(10_000 + seed)
        masks_e = [(torch.rand(D, 2 * C, generator=g) < 0.6).float(),
                   (torch.rand(1, D, generator=g) < 0.8).float()]
        masks_n = [(torch.rand(D, 3 * C, generator=g) < 0.6).float(),
                   (torch.rand(D, D, generator=g) < 0.7).float()]
    else:
        masks_e = None
        masks_n = [torch.ones(D, 3 * C), torch.ones(D, D)]
    m = ref_model.SegmentClassifier(input_dim=F, hidden_dim=D, n_iters=T,
                                    masks_e=masks_e, masks_n=masks_n)
    if scale != 1.0:
        with torch.no_grad():
            for p in m.parameters():
                p.mul_(scale)
    return m, (masks_e, masks_n) if masked else None


def run_traced(m, X, Ri, Ro):
    es, hs = [], []
    h1 = m.edge_network.register_forward_hook(lambda mod, i, o: es.append(o.detach().clone()))
    h2 = m.node_network.register_forward_hook(lambda mod, i, o: hs.append(o.detach().clone()))
    h0 = m.input_network.register_forward_hook(lambda mod, i, o: hs.append(o.detach().clone()))
    with torch.no_grad():
        out = m([X, Ri, Ro])
    for h in (h0, h1, h2):
        h.remove()
    H = [torch.cat([h, X], dim=-1) for h in hs]       # model.py:146,154
    return out, es, H


def pack_model(m, masks):
    d = {"p." + k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
    if masks is not None:
        me, mn = masks
        d["m.edge_network.network.0.weight"] = me[0].numpy()
        d["m.edge_network.network.2.weight"] = me[1].numpy()
        d["m.node_network.network.0.weight"] = mn[0].numpy()
        d["m.node_network.network.2.weight"] = mn[1].numpy()
    return d


def save(name, **arrays):
    arrays["torch_version"] = np.array(torch.__version__)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-28s %8.1f kB" % (name, os.path.getsize(path) / 1e3))


def single(name, graph, D, T, seed, masked=False, scale=1.0, keep_trace=True):
    F = graph.X.shape[1]
    m, masks = make_reference(F, D, T, seed, masked, scale)
    X, Ri, Ro = synth.to_dense(graph)
    X, Ri, Ro = (torch.from_numpy(a)[None] for a in (X, Ri, Ro))
    out, es, H = run_traced(m, X, Ri, Ro)
    assert len(es) == T + 1 and len(H) == T + 1
    arrays = dict(X=graph.X, src=graph.src, dst=graph.dst, n_iters=np.int32(T),
                  scores=out[0].numpy(), **pack_model(m, masks))
    if keep_trace:
        arrays["e_trace"] = np.stack([e[0].numpy() for e in es])
        arrays["H_trace"] = np.stack([h[0].numpy() for h in H])
    save(name, **arrays)


def padded_batch(name, graphs, D, T, seed):
    """Zero-padded batch as merge_graphs builds it (gnn/trainSegmentClassifier.py:66-95)
    plus one Estimator.training_step (gnn/estimator.py:49-60): loss and all 10 gradients."""
    F = graphs[0].X.shape[1]
    B = len(graphs)
    Nmax = max(g.X.shape[0] for g in graphs)
    Emax = max(g.src.shape[0] for g in graphs)
    dense = [synth.to_dense(g, Nmax, Emax) for g in graphs]
    X = torch.from_numpy(np.stack([d[0] for d in dense]))
    Ri = torch.from_numpy(np.stack([d[1] for d in dense]))
    Ro = torch.from_numpy(np.stack([d[2] for d in dense]))
    y = np.zeros((B, Emax), dtype=np.float32)
    for i, g in enumerate(graphs):
        y[i, :g.y.shape[0]] = g.y
    m, masks = make_reference(F, D, T, seed)
    out, es, H = run_traced(m, X, Ri, Ro)
    arrays = dict(n_graphs=np.int32(B), n_iters=np.int32(T), y=y, scores=out.numpy(),
                  e_trace=np.stack([e.numpy() for e in es]), **pack_model(m, masks))
    for i, g in enumerate(graphs):
        arrays["g%d.X" % i], arrays["g%d.src" % i], arrays["g%d.dst" % i] = g.X, g.src, g.dst
    # one SGD training step through the reference Estimator (deterministic; lr irrelevant
    # for the captured loss/gradients, which are taken before the parameter update)
    grads = {}
    hooks = [p.register_hook(lambda gr, k=k: grads.__setitem__(k, gr.detach().clone()))
             for k, p in m.named_parameters()]
    est = ref_estimator.Estimator(m, torch.nn.BCELoss(), opt="SGD", cuda=False, l1=0.)
    loss = est.training_step([X, Ri, Ro], torch.from_numpy(y))
    for h in hooks:
        h.remove()
    arrays["loss"] = np.float32(loss.item())
    for k, v in grads.items():
        arrays["g." + k] = v.numpy()
    save(name, **arrays)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--with-c3-full", action="store_true",
                    help="also run the dense reference at N=10k,E=100k (about 25 s, 16 GB)")
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):   # Estimator prints the model
        pass
    for s in (0, 1, 2):
        # c1 faithful toy: 40 hits / 144 segments, F=2, D=32, T=10 (MPNN_Seg_Toy2D)
        single("toy2d_s%d" % s, synth.toy2d_graph(s), D=32, T=10, seed=s)
        # c2 faithful muon schema: F=11, D=8, T=3
        single("muon_s%d" % s, synth.muon_graph(s), D=8, T=3, seed=s)
        # ACTS phi-sector like: N=100, E=250, F=3, D=8, T=4, with and without masks
        single("sector_s%d" % s, synth.layered_graph(100, 250, 3, seed=s), D=8, T=4, seed=s)
        single("sector_masked_s%d" % s, synth.layered_graph(100, 250, 3, seed=s), D=8, T=4,
               seed=s, masked=True)
    # c1 scale: N=1000, E=5000, F=2, D=32, T=10
    single("c1_scale_s0", synth.layered_graph(1000, 5000, 2, seed=0), D=32, T=10, seed=0)
    # c2 scale (BASELINE "~2k hits"): N=2000, E=10000, F=11, D=8, T=3
    single("c2_scale_s0", synth.layered_graph(2000, 10000, 11, seed=0), D=8, T=3, seed=0)
    # c3 reduced: N=2000, E=20000, F=3, D=8, T=3, full traces
    single("c3_reduced_s0", synth.layered_graph(2000, 20000, 3, seed=0), D=8, T=3, seed=0)
    # stress: weights x3 (saturating tanh amplifies summation-order differences)
    single("sector_w3_s0", synth.layered_graph(100, 250, 3, seed=0), D=8, T=4, seed=0, scale=3.0)
    # other hidden sizes the notebooks use: D=4 (Inference.ipynb), D=16, D=64 (mu200)
    single("sector_d4_s0", synth.layered_graph(100, 250, 3, seed=0), D=4, T=1, seed=0)
    single("sector_d16_s0", synth.layered_graph(100, 250, 3, seed=0), D=16, T=2, seed=0)
    single("sector_d64_s0", synth.layered_graph(200, 800, 3, seed=0), D=64, T=6, seed=0)
    # empty / ragged: a graph with isolated hits and one with a single segment
    iso = synth.layered_graph(50, 20, 3, seed=7)
    single("ragged_isolated_s7", iso, D=8, T=3, seed=7)
    one = synth.HitGraph(iso.X[:3], np.array([0], np.int32), np.array([2], np.int32),
                         np.array([1], np.float32))
    single("ragged_one_segment", one, D=8, T=3, seed=3)
    # zero-padded batch of 4 muon graphs + reference training step (loss, gradients)
    with contextlib.redirect_stdout(io.StringIO()):
        padded_batch("muon_batch4_train", [synth.muon_graph(s) for s in (3, 4, 5, 6)],
                     D=8, T=3, seed=11)
        padded_batch("sector_batch3_train",
                     [synth.layered_graph(60 + 20 * i, 150 + 40 * i, 3, seed=20 + i)
                      for i in range(3)], D=8, T=2, seed=12)
    print("muon_batch4_train, sector_batch3_train written")
    if args.with_c3_full:
        g = synth.layered_graph(10000, 100000, 3, seed=0)
        m, masks = make_reference(3, 8, 3, 0)
        X, Ri, Ro = (torch.from_numpy(a)[None] for a in synth.to_dense(g))
        with torch.no_grad():
            out = m([X, Ri, Ro])
        # inputs are regenerated from the seed by the test: layered_graph(10000,100000,3,seed=0)
        save("c3_full_s0", n_iters=np.int32(3), scores=out[0].numpy(),
             gen=np.array("layered_graph(10000,100000,3,seed=0)"), **pack_model(m, masks))


if __name__ == "__main__":
    main()

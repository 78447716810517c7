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


def pruned(name, graph, D, T, seed, dead_e, dead_q, dead_h):
    """Masks that remove WHOLE units (the pattern of the reference's own pruned model: rows 3, 4 of
    the first edge layer are zero in gnn/MPNN_Seg_ACTS_maskedlinear.ipynb cell 34):
      dead_e  edge-network hidden units: half of them lose their ROW of the first edge layer (the
              unit becomes the constant tanh(b1_i)), the other half their entry of the second
      dead_q  node-network hidden units, same two ways (row of layer 0 / column of layer 2)
      dead_h  hit features H'_k never read by any layer (their columns of both first layers)
    plus the random element-wise masks of `make_reference(masked=True)` on what is left."""
    F = graph.X.shape[1]
    C = D + F
    g = torch.Generator().manual_seed(20_000 + seed)
    me = [(torch.rand(D, 2 * C, generator=g) < 0.8).float(), torch.ones(1, D)]
    mn = [(torch.rand(D, 3 * C, generator=g) < 0.8).float(), (torch.rand(D, D, generator=g) < 0.9).float()]
    for n, i in enumerate(dead_e):
        if n % 2 == 0:
            me[0][i, :] = 0
        else:
            me[1][0, i] = 0
    for n, i in enumerate(dead_q):
        if n % 2 == 0:
            mn[0][i, :] = 0
        else:
            mn[1][:, i] = 0
    for k in dead_h:
        me[0][:, k] = 0
        me[0][:, C + k] = 0
        mn[0][:, k] = 0
        mn[0][:, C + k] = 0
        mn[0][:, 2 * C + k] = 0
    torch.manual_seed(seed)
    m = ref_model.SegmentClassifier(input_dim=F, hidden_dim=D, n_iters=T, masks_e=me, masks_n=mn)
    X, Ri, Ro = (torch.from_numpy(a)[None] for a in synth.to_dense(graph))
    out, es, H = run_traced(m, X, Ri, Ro)
    save(name, X=graph.X, src=graph.src, dst=graph.dst, n_iters=np.int32(T), scores=out[0].numpy(),
         e_trace=np.stack([e[0].numpy() for e in es]), H_trace=np.stack([h[0].numpy() for h in H]),
         dead_e=np.asarray(dead_e, np.int32), dead_q=np.asarray(dead_q, np.int32),
         dead_h=np.asarray(dead_h, np.int32), **pack_model(m, (me, mn)))


def ref_written_files():
    """Graph files written BY THE REFERENCE'S OWN WRITERS, and what its model scores on them.

    tests/golden/ref_written/graph000000.npz       gnn/graph.py:179-181  save_graph((SparseGraph, segments), f)
    tests/golden/ref_written/graph_muon_000000.npz gnn/Muon_graph.py:198-205 save_graph(graph, particle, f): + pt, eta
    The SparseGraph comes from the reference's make_sparse_graph (Ri.nonzero(), gnn/graph.py:23-26) on
    dense matrices filled by its rule (gnn/graph.py:132-135); expected scores: the reference model on
    graph_from_sparse(load_graph(f, SparseGraph)) (gnn/graph.py:28-35,188-194)."""
    import collections
    import graph as ref_graph
    import Muon_graph as ref_muon
    d = os.path.join(OUT, "ref_written")
    os.makedirs(d, exist_ok=True)
    for kind, g, seed in (("sector", synth.layered_graph(120, 400, 3, seed=31), 31),
                          ("muon", synth.muon_graph(7), 32)):
        X, Ri, Ro = synth.to_dense(g, dtype=np.uint8)
        if kind == "sector":
            sp = ref_graph.make_sparse_graph(X, Ri, Ro, g.y)
            fn = os.path.join(d, "graph000000")
            ref_graph.save_graph((sp, None), fn)
            back = ref_graph.load_graph(fn + ".npz", ref_graph.SparseGraph)
            dense = ref_graph.graph_from_sparse(back)
        else:
            sp = ref_muon.make_sparse_graph(X, Ri, Ro, g.y)
            fn = os.path.join(d, "graph_muon_000000")
            particle = collections.namedtuple("P", ["vp_pt", "vp_eta"])(np.float32(23.5), np.float32(-1.7))
            ref_muon.save_graph((sp, None), particle, fn)
            back = ref_muon.load_graph(fn + ".npz", ref_muon.SparseGraphProp)
            dense = ref_muon.graph_from_sparse_prop(back)
        F = g.X.shape[1]
        m, masks = make_reference(F, 8, 3, seed)
        Xt, Rit, Rot = (torch.from_numpy(np.asarray(a, dtype=np.float32))[None]
                        for a in (dense.X, dense.Ri, dense.Ro))
        with torch.no_grad():
            out = m([Xt, Rit, Rot])
        save("refnpz_%s" % kind, filename=np.array(os.path.basename(fn) + ".npz"), n_iters=np.int32(3),
             scores=out[0].numpy(), Ri=np.asarray(dense.Ri), Ro=np.asarray(dense.Ro), **pack_model(m, masks))
        print("   wrote", fn + ".npz", sorted(np.load(fn + ".npz").files))


def batch_generator_fixture(name, graphs, n_samples, batch_size, D, T, seed):
    """What the reference's batch_generator hands the model, batch by batch
    (gnn/trainSegmentClassifier.py:97-111): graphs[j:j+batch_size] densified by the IMPORTED
    graph.graph_from_sparse (gnn/graph.py:28-35), merged by merge_graphs' rule (:66-95, restated
    here line by line - the CLI module itself does not import: SURVEY 2 row 5), cast to float32
    (:38-44); the reference model's scores [B, E_max] and BCELoss per batch."""
    import graph as ref_graph
    sparse = []
    for g in graphs:
        X, Ri, Ro = synth.to_dense(g, dtype=np.uint8)
        sparse.append(ref_graph.make_sparse_graph(X, Ri, Ro, g.y))

    def merge_graphs(gs):                                      # trainSegmentClassifier.py:66-95
        if len(gs) == 1:
            g = gs[0]
            return g.X[None], g.Ri[None], g.Ro[None], g.y[None]
        n_nodes = np.array([g.X.shape[0] for g in gs])
        n_edges = np.array([g.y.shape[0] for g in gs])
        bX = np.zeros((len(gs), n_nodes.max(), gs[0].X.shape[1]), dtype=np.float32)
        bRi = np.zeros((len(gs), n_nodes.max(), n_edges.max()), dtype=np.uint8)
        bRo = np.zeros((len(gs), n_nodes.max(), n_edges.max()), dtype=np.uint8)
        by = np.zeros((len(gs), n_edges.max()), dtype=np.uint8)
        for i, g in enumerate(gs):
            bX[i, :n_nodes[i]] = g.X
            bRi[i, :n_nodes[i], :n_edges[i]] = g.Ri
            bRo[i, :n_nodes[i], :n_edges[i]] = g.Ro
            by[i, :n_edges[i]] = g.y
        return bX, bRi, bRo, by

    F = graphs[0].X.shape[1]
    m, masks = make_reference(F, D, T, seed)
    arrays = dict(n_graphs=np.int32(len(graphs)), n_samples=np.int32(n_samples),
                  batch_size=np.int32(batch_size), n_iters=np.int32(T), **pack_model(m, masks))
    for i, sp in enumerate(sparse):
        for k, v in sp._asdict().items():
            arrays["s%d.%s" % (i, k)] = np.asarray(v)
    for b, j in enumerate(np.arange(0, n_samples, batch_size)):     # :99-103
        bg = [ref_graph.graph_from_sparse(g) for g in sparse[j:j + batch_size]]
        bX, bRi, bRo, by = merge_graphs(bg)
        to_t = lambda a: torch.from_numpy(a.astype(np.float32))     # :38-44
        with torch.no_grad():
            out = m([to_t(bX), to_t(bRi), to_t(bRo)])
            loss = torch.nn.BCELoss()(out, to_t(by))
        arrays["b%d.scores" % b] = out.numpy()
        arrays["b%d.y" % b] = by.astype(np.float32)
        arrays["b%d.loss" % b] = np.float32(loss.item())
    arrays["n_batches"] = np.int32(b + 1)
    save(name, **arrays)


def round2():
    """Fixtures added in round 2: whole-unit pruning, files written by the
    reference's own writers, the batch generator's batches."""
    pruned("pruned_units_d8_s0", synth.layered_graph(100, 250, 3, seed=0), D=8, T=4, seed=40,
           dead_e=[1, 3, 4, 6], dead_q=[0, 2, 5, 7], dead_h=[2, 3, 5, 6])
    pruned("pruned_units_d16_s0", synth.layered_graph(150, 500, 3, seed=1), D=16, T=2, seed=41,
           dead_e=list(range(0, 16, 2)) + [1, 3], dead_q=list(range(1, 16, 2)) + [0],
           dead_h=list(range(4, 13)))
    pruned("pruned_units_muon_s0", synth.muon_graph(2), D=8, T=3, seed=42,
           dead_e=[0, 2, 5, 7], dead_q=[1, 3, 4, 6], dead_h=[0, 1, 6, 7])
    pruned("pruned_edge_only_d8_s0", synth.layered_graph(100, 250, 3, seed=2), D=8, T=3, seed=43,
           dead_e=[3, 4], dead_q=[], dead_h=[])      # the notebook's own pattern: nothing to shrink to
    ref_written_files()
    batch_generator_fixture("batchgen_sector_b2", [synth.layered_graph(50 + 15 * i, 120 + 35 * i, 3, seed=50 + i)
                                                   for i in range(5)], n_samples=5, batch_size=2, D=8, T=3, seed=13)
    batch_generator_fixture("batchgen_muon_b4", [synth.muon_graph(20 + s) for s in range(8)],
                            n_samples=8, batch_size=4, D=8, T=3, seed=14)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--with-c3-full", action="store_true",
                    help="also run the dense reference at N=10k,E=100k (about 25 s, 16 GB)")
    ap.add_argument("--only-round2", action="store_true",
                    help="write only the fixtures added in round 2 (the others are unchanged)")
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    if args.only_round2:
        return round2()
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
    round2()
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

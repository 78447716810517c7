"""CPU-side checks: the C-ABI library loads and exports every symbol include/gnn_hip.h
declares; host logic (index-form batch, dense adapter, npz loader, module tree)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from golden_util import Fixture
from gnn_fpga_amd import HitGraphBatch, synth

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(REPO, "include", "gnn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gnn_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from gnn_fpga_amd import _lib
    names = _declared_functions()
    assert len(names) >= 10
    assert sorted(_lib.SIGNATURES) == names          # binding covers the whole header
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n
    loaded = _lib.load()                              # no GPU needed for these calls
    assert loaded.gnn_abi_version() == _lib.GNN_ABI_VERSION
    assert loaded.gnn_shape_supported(3, 8) == 1
    assert loaded.gnn_shape_supported(5, 7) == 0
    assert loaded.gnn_h_stride(3, 8) == 12
    assert loaded.gnn_forward_workspace_bytes(10, 20, 3, 8) > 0


def test_no_cpu_fallback_in_product():
    """The product path must not import the oracle or run on CPU tensors."""
    pkg = os.path.join(REPO, "gnn-fpga_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn
    from gnn_fpga_amd.model import SegmentClassifier
    m = SegmentClassifier(input_dim=3, hidden_dim=8, n_iters=1).eval()
    g = synth.layered_graph(30, 50, 3, seed=0)
    with torch.no_grad(), pytest.raises(RuntimeError):
        m(HitGraphBatch.from_graphs([g]))


def test_module_tree_matches_reference_state_dict():
    from gnn_fpga_amd.model import SegmentClassifier
    fx = Fixture("sector_s0")
    m = SegmentClassifier(input_dim=3, hidden_dim=8, n_iters=4)
    assert list(m.state_dict().keys()) == list(fx.params.keys())
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == fx.params[k].shape
    assert sum(p.numel() for p in m.parameters()) == 569
    assert list(m.named_buffers()) == []
    m.load_state_dict({k: torch.from_numpy(v) for k, v in fx.params.items()})
    # what gnn/estimator.py:54-55 iterates
    assert len([l.weight for l in m.node_network.network if hasattr(l, "weight")]) == 2
    assert len([l.weight for l in m.edge_network.network if hasattr(l, "weight")]) == 2


def test_masked_linear_semantics():
    from gnn_fpga_amd.model import MaskedLinear
    torch.manual_seed(0)
    l = MaskedLinear(6, 4)
    assert l.mask_flag is False
    w0 = l.weight.detach().clone()
    mask = (torch.rand(4, 6) < 0.5).float()
    l.set_mask(mask)
    assert l.mask_flag is True
    assert torch.equal(l.weight.detach(), w0 * mask)          # gnn/model.py:21
    x = torch.randn(3, 6)
    assert torch.allclose(l(x), torch.nn.functional.linear(x, w0 * mask, l.bias))
    assert "mask" not in l.state_dict()


def test_csr_layout_and_on_disk_order():
    g = synth.layered_graph(200, 900, 3, seed=4)
    b = HitGraphBatch.from_graphs([g])
    in_ptr, in_eid, in_nbr = b.in_ptr.numpy(), b.in_eid.numpy(), b.in_nbr.numpy()
    out_ptr, out_eid, out_nbr = b.out_ptr.numpy(), b.out_eid.numpy(), b.out_nbr.numpy()
    for n in range(200):
        seg = in_eid[in_ptr[n]:in_ptr[n + 1]]
        assert np.all(g.dst[seg] == n) and np.all(np.diff(seg) > 0)
        assert np.all(in_nbr[in_ptr[n]:in_ptr[n + 1]] == g.src[seg])
        seg = out_eid[out_ptr[n]:out_ptr[n + 1]]
        assert np.all(g.src[seg] == n) and np.all(np.diff(seg) > 0)
        assert np.all(out_nbr[out_ptr[n]:out_ptr[n + 1]] == g.dst[seg])
    # the reference's on-disk SparseGraph arrays (nonzero() of the dense matrices,
    # gnn/graph.py:23-26) are this CSR order already
    X, Ri, Ro = synth.to_dense(g)
    Ri_rows, Ri_cols = Ri.nonzero()
    Ro_rows, Ro_cols = Ro.nonzero()
    assert np.array_equal(Ri_cols, in_eid) and np.array_equal(Ro_cols, out_eid)
    b2 = HitGraphBatch.from_sparse_arrays(X, Ri_rows, Ri_cols, Ro_rows, Ro_cols)
    for k in ("src", "dst", "in_ptr", "in_eid", "in_nbr", "out_ptr", "out_eid", "out_nbr"):
        assert torch.equal(getattr(b, k), getattr(b2, k)), k


def test_dense_round_trip_and_padding(tmp_path):
    """Dense <-> index round trip (gnn/GraphConstructionDev_mu200.ipynb cell 41-44) and the
    zero-padded batch of gnn/trainSegmentClassifier.py:66-95."""
    gs = [synth.muon_graph(s) for s in range(3)]
    Nmax = max(g.X.shape[0] for g in gs)
    Emax = max(g.src.shape[0] for g in gs)
    dense = [synth.to_dense(g, Nmax, Emax) for g in gs]
    X, Ri, Ro = (torch.from_numpy(np.stack([d[i] for d in dense])) for i in range(3))
    b = HitGraphBatch.from_dense(X, Ri, Ro)
    assert b.dense_shape == (3, Nmax, Emax)
    src = b.src.numpy().reshape(3, Emax)
    dst = b.dst.numpy().reshape(3, Emax)
    for i, g in enumerate(gs):
        E = g.src.shape[0]
        assert np.array_equal(src[i, :E], g.src + i * Nmax)
        assert np.array_equal(dst[i, :E], g.dst + i * Nmax)
        assert np.all(src[i, E:] == -1) and np.all(dst[i, E:] == -1)
    assert int(b.in_ptr[-1]) == sum(g.src.shape[0] for g in gs)
    # npz written with the reference's key schema (gnn/graph.py:179-181)
    g = gs[0]
    Xd, Rid, Rod = synth.to_dense(g)
    rr, rc = Rid.nonzero()
    orr, oc = Rod.nonzero()
    fn = str(tmp_path / "event000000.npz")
    np.savez(fn, X=Xd, Ri_rows=rr, Ri_cols=rc, Ro_rows=orr, Ro_cols=oc, y=g.y)
    bl = HitGraphBatch.from_npz(fn)
    assert np.array_equal(bl.src.numpy(), g.src) and np.array_equal(bl.dst.numpy(), g.dst)


def test_bad_inputs_raise():
    g = synth.layered_graph(30, 50, 3, seed=0)
    with pytest.raises(ValueError):
        HitGraphBatch(g.X, g.src + 100, g.dst)
    with pytest.raises(ValueError):
        HitGraphBatch(g.X, np.where(np.arange(50) == 0, -1, g.src), g.dst)
    X, Ri, Ro = synth.to_dense(g)
    Ri[0, 0] = Ri[1, 0] = 1
    with pytest.raises(ValueError):
        HitGraphBatch.from_dense(X, Ri, Ro)


def test_pipelined_kernels_never_spill():
    """Cfg::pipelined shapes keep asm-loaded registers in flight across a slice; that is only
    legal with zero scratch (a spill would save a register before its load has landed).  The
    compiler's resource remarks of the real build are kept in build/sell_pipeline.remarks."""
    import subprocess
    path = os.path.join(REPO, "build", "sell_pipeline.remarks")
    if not os.path.exists(path):
        pytest.skip("no build remarks (library was built elsewhere)")
    txt = open(path).read()
    blocks = re.split(r"remark: Function Name: ", txt)[1:]
    seen = 0
    for b in blocks:
        name = subprocess.run(["c++filt", b.split()[0]], capture_output=True, text=True).stdout
        m = re.search(r"(k_iter2?)<(\d+), (\d+), (true|false), (true|false)(, (true|false))?>", name)
        if not m:
            continue
        F, D = int(m.group(2)), int(m.group(3))
        if m.group(1) == "k_iter":
            pipelined = (D <= 8 and F <= 3) or D == 4             # mirrors Cfg::pipelined
            if m.group(7) == "true":                              # the training variant (TR) waits for its prefetch
                pipelined = False                                 # at once: nothing asm-loaded is in flight across work
        else:
            pipelined = D <= 8                                    # mirrors Cfg::iter2
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
        agprs = int(re.search(r"AGPRs: (\d+)", b).group(1))
        if pipelined:
            seen += 1
            assert agprs == 0, "%s<%d,%d> is pipelined but parks values in AGPRs" % (m.group(1), F, D)
            assert scratch == 0, "%s<%d,%d> is pipelined but spills %d bytes" % (m.group(1), F, D, scratch)
    assert seen >= 20 + 27          # k_iter: 5 shapes x 4 variants; k_iter2: 6 shapes x 4 + 3 fused-first


def test_event_layout_checks_block_diagonality():
    """HitGraphBatch.event_layout: per-graph offsets for the one-workgroup-per-graph kernel, and
    None when a segment crosses a graph boundary (that kernel indexes LDS by local hit id)."""
    graphs = [synth.muon_graph(seed=i) for i in range(5)]
    b = HitGraphBatch.from_graphs(graphs)
    lay = b.event_layout()
    assert lay is not None and lay is b.event_layout()          # cached
    assert lay.hit_ptr.dtype == torch.int32 and lay.hit_ptr.tolist() == b.hit_ptr.tolist()
    assert lay.seg_ptr.tolist() == b.seg_ptr.tolist()
    assert lay.max_hits == max(g.X.shape[0] for g in graphs)
    assert lay.max_segments == max(g.src.shape[0] for g in graphs)
    g = synth.layered_graph(60, 100, 3, seed=2)
    assert HitGraphBatch(g.X, g.src, g.dst).event_layout() is not None      # one graph: trivially fine
    assert HitGraphBatch(g.X, g.src, g.dst, hit_ptr=[0, 30, 60],
                         seg_ptr=[0, 50, 100]).event_layout() is None
    # padded segments (-1) are allowed anywhere
    X, Ri, Ro = synth.to_dense(graphs[0], 40, 200)
    pb = HitGraphBatch.from_dense(torch.from_numpy(X)[None], torch.from_numpy(Ri)[None],
                                  torch.from_numpy(Ro)[None])
    assert pb.event_layout() is not None and pb.event_layout().max_segments == 200


def test_c_abi_rejects_bad_arguments_without_a_gpu():
    """Error behaviour of the C ABI (include/gnn_hip.h: 0 on success, GNN_ERR_* otherwise,
    gnn_last_error() explains): every entry point validates its arguments before it touches the
    device, so this runs on the CPU-only build box."""
    from gnn_fpga_amd import _lib
    lib = _lib.load()
    G, P = _lib.GnnGraph(), _lib.GnnParams()
    P.F, P.D = 3, 8
    G.n_hits, G.n_segments = 5, 7                     # sizes without arrays
    BAD, UNS = _lib.GNN_ERR_BADARG, _lib.GNN_ERR_UNSUPPORTED
    err = lambda: lib.gnn_last_error().decode()
    assert lib.gnn_segclf_forward(None, None, 1, None, None, None, None, 0, None) == BAD and "bad argument" in err()
    assert lib.gnn_segclf_forward(ctypes.byref(G), ctypes.byref(P), -1, None, None, None, None, 0, None) == BAD
    assert lib.gnn_input_fwd(None, None, None, None, 4, 3, 8, 12, None) == BAD
    assert lib.gnn_segclf_forward_events(ctypes.byref(G), ctypes.byref(P), None, None, 2, 5, 7, 1, None, None) == BAD
    assert "offsets" in err()
    assert lib.gnn_bce_loss(None, None, 5, 1.0, None, None, None, None) == BAD
    assert lib.gnn_segclf_forward_plan(None, None, 1, None, None, 0, None) != 0
    # shapes without kernels are reported as unsupported, not run
    P.F, P.D = 5, 7
    assert lib.gnn_events_supported(5, 7, 10, 10) == 0
    assert lib.gnn_events_supported(3, 8, 10 ** 6, 10 ** 7) == 0          # does not fit one workgroup
    assert lib.gnn_events_supported(11, 8, 40, 150) == 1
    assert lib.gnn_plan_shape_supported(5, 7) == 0 and lib.gnn_plan_shape_supported(3, 64) == 1
    lim = (ctypes.c_int32 * 4)()
    assert lib.gnn_plan_limits(5, 7, lim) == UNS and "input_dim=5" in err()
    assert lib.gnn_plan_limits(3, 8, lim) == 0 and lim[0] >= 16 and lim[2] >= 1024
    assert lib.gnn_forward_workspace_bytes(-1, 5, 3, 8) == 0 or True       # size queries never fail hard
    assert lib.gnn_plan_workspace_bytes(10, 20, 5, 7) == 0


def test_csr_is_built_on_first_use_only():
    """The two CSRs are not needed by the tiled pipeline: HitGraphBatch builds them when a consumer
    (per-module kernels, small-event kernel, backward) first asks, also after a device move."""
    g = synth.layered_graph(200, 900, 3, seed=4)
    b = HitGraphBatch.from_graphs([g, g])
    assert b._csr is None
    b.build_plan(8, dict(tile_hits=64, iter_records=40, chunk_segments=50, edge_records=60))
    assert b._csr is None                                   # the plan does not touch them
    b.to("cpu")
    ip, ie, inb = b.in_ptr, b.in_eid, b.in_nbr
    assert b._csr is not None and b._src_host is None
    assert ip.dtype == torch.int32 and ip.shape == (401,) and int(ip[-1]) == 1800
    src, dst = b.src.numpy(), b.dst.numpy()
    assert np.array_equal(inb.numpy(), src[ie.numpy()])
    assert np.all(np.diff(dst[ie.numpy()]) >= 0)            # grouped by end hit
    assert np.array_equal(b.out_nbr.numpy(), dst[b.out_eid.numpy()])


def test_calls_refuse_tensors_of_another_device():
    """_lib._on makes the tensors' device current for a call and refuses structs / tensors that
    live elsewhere (a launch on device 0 with device-1 pointers is a GPU fault, not an exception)."""
    import torch
    from gnn_fpga_amd import _lib
    with pytest.raises(_lib.GnnHipError):
        _lib._on(torch.zeros(3))                                   # CPU tensor: no CPU path
    g = _lib.GnnGraph()
    g._device = torch.device("cuda", 1)
    with pytest.raises(_lib.GnnHipError):
        _lib._on(torch.device("cuda", 0), g)                       # struct built for another device
    _lib._on(torch.device("cuda", 1), g)                           # same device: accepted (not entered)
    # while a call on cuda:1 is in progress, a cuda:0 / CPU tensor is refused by _dev()
    _lib._cur_dev = torch.device("cuda", 1)
    try:
        with pytest.raises(_lib.GnnHipError):
            _lib._dev(torch.zeros(3), torch.float32, "x")
    finally:
        _lib._cur_dev = None


def test_event_route_decision_is_pinned():
    """`_lib.events_preferred` (which batches take the one-workgroup-per-graph kernels) on both sides of
    each threshold: <= 1200 segments per graph (tools/cliff_probe.py: 256 x (150, 1000) one-launch 0.050
    ms vs tiled 0.066; 256 x (300, 2000) 0.119 vs 0.069), hidden_dim <= 16 (one D = 32 toy graph 0.123 vs
    0.089), the graph must fit the kernel's LDS.  The GPU twin of this test
    (test_gpu_formats.py::test_event_route_crossover) checks the kernels that ran and their times."""
    from gnn_fpga_amd import _lib
    from gnn_fpga_amd.hitgraph import _EventLayout

    def lay(h, s):
        return _EventLayout([0, h], [0, s])

    assert _lib.EVENTS_MAX_SEGMENTS == 1200
    assert _lib.events_preferred(3, 8, lay(150, 1000))
    assert _lib.events_preferred(3, 8, lay(150, 1200))
    assert not _lib.events_preferred(3, 8, lay(150, 1201))
    assert not _lib.events_preferred(3, 8, lay(300, 2000))
    assert _lib.events_preferred(11, 8, lay(40, 150))                 # the reference's muon graphs
    assert _lib.events_preferred(3, 16, lay(100, 250))
    assert not _lib.events_preferred(2, 32, lay(40, 144))             # wide hidden layers: tiled
    assert not _lib.events_preferred(3, 8, None)                      # not block-diagonal
    assert not _lib.events_preferred(3, 8, lay(40000, 1000))          # does not fit the LDS of one workgroup

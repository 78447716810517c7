"""ctypes binding of libgnn_hip.so (C ABI in include/gnn_hip.h).

This is the only compute path of the package: there is no CPU or eager-PyTorch
fallback.  If the library is missing or a tensor is not on a ROCm device the call
raises.  torch is used for device memory and the stream handle only.
"""
import ctypes
import functools
import os

import torch  # imported first: its bundled libamdhip64.so.7 is the one HIP runtime of the process

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgnn_hip.so")

GNN_ABI_VERSION = 5
GNN_ERR_UNSUPPORTED = -10001
GNN_ERR_BADARG = -10002
GNN_ERR_WORKSPACE = -10003
GNN_FLAG_EXP_PRODUCT = 1
GNN_FLAG_BF16_MLP = 2

_f = ctypes.c_void_p          # device pointers travel as integers
_i32, _i64, _sz = ctypes.c_int32, ctypes.c_int64, ctypes.c_size_t


class GnnParams(ctypes.Structure):
    _fields_ = [(n, _f) for n in ("Win", "bin", "W1", "b1", "W2", "b2", "W3", "b3", "W4", "b4")] + \
               [("F", _i32), ("D", _i32), ("flags", _i32)]


class GnnGraph(ctypes.Structure):
    _fields_ = [(n, _f) for n in ("X", "src", "dst", "in_ptr", "in_eid", "in_nbr",
                                  "out_ptr", "out_eid", "out_nbr")] + \
               [("n_hits", _i64), ("n_segments", _i64)]


class GnnGrads(ctypes.Structure):
    _fields_ = [(n, _f) for n in ("Win", "bin", "W1", "b1", "W2", "b2", "W3", "b3", "W4", "b4")]


class GnnPlan(ctypes.Structure):
    _fields_ = [(n, _f) for n in ("X", "src", "dst", "in_off", "in_nbr", "out_off", "out_nbr",
                                  "tiles", "chunks", "in_off16", "in_nbr16", "out_off16",
                                  "out_nbr16", "sched_a", "sched_b", "sd16")] + \
               [("n_pad", _i64), ("n_segments", _i64), ("n_tiles", _i64), ("n_chunks", _i64),
                ("iter_lds_records", _i64), ("edge_lds_rows", _i64), ("n_lds_tiles", _i64),
                ("iter_lds_in", _i64), ("iter_lds_out", _i64), ("tile_hits_max", _i64),
                ("max_list_steps", _i64)]


class GnnPlanSizes(ctypes.Structure):
    _fields_ = [(n, _i64) for n in ("n_pad", "n_tiles", "n_slices", "n_chunks", "in_total", "out_total",
                                    "in16_words", "out16_words", "n_sched", "iter_lds_records",
                                    "edge_lds_rows", "n_lds_tiles", "n_lds_chunks", "iter_lds_in",
                                    "iter_lds_out", "tile_hits_max", "max_list_steps", "n_valid",
                                    "max_level", "status", "list_mode")]


class GnnPlanOut(ctypes.Structure):
    _fields_ = [(n, _f) for n in ("X", "x_absmax", "src", "dst", "sd16", "in_off", "in_nbr", "out_off",
                                  "out_nbr", "in_off16", "in_nbr16", "out_off16", "out_nbr16", "tiles",
                                  "chunks", "sched_a", "sched_b", "perm", "src_abs", "dst_abs", "level")]


# name -> (restype, argtypes); must list every function include/gnn_hip.h declares
SIGNATURES = {
    "gnn_abi_version": (ctypes.c_int, []),
    "gnn_last_error": (ctypes.c_char_p, []),
    "gnn_shape_supported": (ctypes.c_int, [_i32, _i32]),
    "gnn_h_stride": (_i32, [_i32, _i32]),
    "gnn_input_fwd": (ctypes.c_int, [_f, _f, _f, _f, _i64, _i32, _i32, _i32, _f]),
    "gnn_edge_fwd": (ctypes.c_int, [_f, _i32, _f, _f, _f, _f, _f, _f, _f, _f, _i64, _i64,
                                    _i32, _i32, _f]),
    "gnn_node_fwd": (ctypes.c_int, [_f, _i32, _f, ctypes.POINTER(GnnGraph), _f, _f, _f, _f, _f,
                                    _i32, _i32, _f]),
    "gnn_forward_workspace_bytes": (_sz, [_i64, _i64, _i32, _i32]),
    "gnn_segclf_forward": (ctypes.c_int, [ctypes.POINTER(GnnGraph), ctypes.POINTER(GnnParams),
                                          _i32, _f, _f, _f, _f, _sz, _f]),
    "gnn_events_supported": (ctypes.c_int, [_i32, _i32, _i64, _i64]),
    "gnn_segclf_forward_events": (ctypes.c_int, [ctypes.POINTER(GnnGraph), ctypes.POINTER(GnnParams),
                                                 _f, _f, _i64, _i32, _i32, _i32, _f, _f]),
    "gnn_segclf_forward_train_events": (ctypes.c_int, [ctypes.POINTER(GnnGraph), ctypes.POINTER(GnnParams),
                                                       _f, _f, _i64, _i32, _i32, _i32, _f, _f, _f]),
    "gnn_events_backward_supported": (ctypes.c_int, [_i32, _i32, _i64, _i64]),
    "gnn_backward_events_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "gnn_segclf_backward_events": (ctypes.c_int, [ctypes.POINTER(GnnGraph), ctypes.POINTER(GnnParams),
                                                  _f, _f, _i64, _i32, _i32, _i32, _f, _f, _f,
                                                  ctypes.POINTER(GnnGrads), _f, _sz, _f]),
    "gnn_segclf_forward_train": (ctypes.c_int, [ctypes.POINTER(GnnGraph),
                                                ctypes.POINTER(GnnParams), _i32, _f, _f, _f, _f, _sz, _f]),
    "gnn_backward_workspace_bytes": (_sz, [_i64, _i64, _i32, _i32]),
    "gnn_segclf_backward": (ctypes.c_int, [ctypes.POINTER(GnnGraph), ctypes.POINTER(GnnParams),
                                           _i32, _f, _f, _f, _f, ctypes.POINTER(GnnGrads), _f, _sz, _f]),
    "gnn_bce_loss": (ctypes.c_int, [_f, _f, _i64, ctypes.c_float, _f, _f, _f, _f]),
    "gnn_dense_to_index": (ctypes.c_int, [_f, _f, _i64, _i64, _i64, _f, _f, _f, _f]),
    "gnn_edge_bwd": (ctypes.c_int, [_f, _i32, ctypes.POINTER(GnnGraph), ctypes.POINTER(GnnParams), _f, _f, _f,
                                    ctypes.POINTER(GnnGrads), _f, _sz, _f]),
    "gnn_node_bwd": (ctypes.c_int, [_f, _i32, _f, _f, ctypes.POINTER(GnnGraph), ctypes.POINTER(GnnParams), _f, _f,
                                    _f, ctypes.POINTER(GnnGrads), _f, _sz, _f]),
    "gnn_plan_workspace_bytes": (_sz, [_i64, _i64, _i32, _i32]),
    "gnn_segclf_forward_plan": (ctypes.c_int, [ctypes.POINTER(GnnPlan), ctypes.POINTER(GnnParams),
                                               _i32, _f, _f, _sz, _f]),
    "gnn_segclf_forward_train_plan": (ctypes.c_int, [ctypes.POINTER(GnnPlan), ctypes.POINTER(GnnParams), _i32, _f, _f, _f,
                                                     _f, _f, _f, _f, _f, _sz, _f]),
    "gnn_plan_shape_supported": (ctypes.c_int, [_i32, _i32]),
    "gnn_plan_limits": (ctypes.c_int, [_i32, _i32, ctypes.POINTER(_i32)]),
    "gnn_exp_product_bound": (ctypes.c_int, [ctypes.POINTER(GnnParams), _f, _f, _f]),
    "gnn_csr_build_workspace_bytes": (_sz, [_i64, _i64]),
    "gnn_csr_build": (ctypes.c_int, [_f, _f, _i64, _i64, _f, _f, _f, _f, _f, _f, _f, _f, _sz, _f]),
    "gnn_plan_build_workspace_bytes": (_sz, [_i64, _i64, _i32]),
    "gnn_plan_build_sizes": (ctypes.c_int, [_f, _f, _f, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _f, _sz,
                                            _f, _f]),
    "gnn_plan_build_sizes_graphs": (ctypes.c_int, [_f, _f, _f, _f, _i64, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _i32,
                                                   _f, _sz, _f, _f]),
    "gnn_plan_build_fill": (ctypes.c_int, [_f, _i32, _f, _f, _i64, _i64, _i32, ctypes.POINTER(GnnPlanSizes),
                                           _f, _sz, ctypes.POINTER(GnnPlanOut), _f]),
    "gnn_profile_begin": (ctypes.c_int, [_i32]),
    "gnn_profile_end": (ctypes.c_int, [ctypes.POINTER(ctypes.c_char_p),
                                       ctypes.POINTER(ctypes.c_float), _i32]),
}

_lib = None


class GnnHipError(RuntimeError):
    pass


def load():
    """Load libgnn_hip.so once; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GnnHipError(
            "HIP library %s is missing - build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C gnn-fpga_amd/csrc`; there is no CPU fallback" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)           # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    if lib.gnn_abi_version() != GNN_ABI_VERSION:
        raise GnnHipError("libgnn_hip.so ABI %d != binding ABI %d"
                          % (lib.gnn_abi_version(), GNN_ABI_VERSION))
    with open("/proc/self/maps") as m:
        runtimes = {ln.split()[-1] for ln in m if "libamdhip64" in ln}
    if len(runtimes) > 1:
        raise GnnHipError("two HIP runtimes mapped (%s): kernels and torch streams would not "
                          "share a context" % ", ".join(sorted(runtimes)))
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise GnnHipError("libgnn_hip: %s (code %d)" % (load().gnn_last_error().decode(), rc))


_cur_dev = None     # device of the call in progress (set by `_on`); _dev() checks tensors against it


def _dev(t, dtype, what):
    if not torch.is_tensor(t) or not t.is_cuda:
        raise GnnHipError("%s must be a tensor on a ROCm device (no CPU path exists)" % what)
    if t.dtype != dtype or not t.is_contiguous():
        raise GnnHipError("%s must be contiguous %s" % (what, dtype))
    if _cur_dev is not None and t.device != _cur_dev:
        raise GnnHipError("%s is on %s but this call runs on %s: every tensor of one call must live "
                          "on one device" % (what, t.device, _cur_dev))
    return t.data_ptr()


class _on:
    """`with _on(tensor_or_device) as stream:` - makes that device CURRENT for the library call (the
    kernels launch on the current device; a stream of another device would be an invalid handle and
    another device's pointers an illegal address), hands out ITS current stream, and lets _dev()
    refuse tensors that live elsewhere.  Structs built earlier carry `_device` and are checked too."""

    def __init__(self, where, *structs):
        dev = where.device if torch.is_tensor(where) else torch.device(where)
        if dev.type != "cuda":
            raise GnnHipError("tensors must be on a ROCm device (no CPU path exists); got %s" % dev)
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        for st in structs:
            sd = getattr(st, "_device", None)
            if st is not None and sd is not None and sd != dev:
                raise GnnHipError("%s was built for %s but this call runs on %s"
                                  % (type(st).__name__, sd, dev))
        self.dev = dev
        self.ctx = torch.cuda.device(dev)

    def __enter__(self):
        global _cur_dev
        self.ctx.__enter__()
        self.prev, _cur_dev = _cur_dev, self.dev
        return torch.cuda.current_stream(self.dev).cuda_stream

    def __exit__(self, *exc):
        global _cur_dev
        _cur_dev = self.prev
        return self.ctx.__exit__(*exc)


def shape_supported(F, D):
    return bool(load().gnn_shape_supported(F, D))


def h_stride(F, D):
    s = load().gnn_h_stride(F, D)
    if s == 0:
        raise GnnHipError("no HIP kernel for input_dim=%d hidden_dim=%d" % (F, D))
    return s


def graph_struct(batch):
    i32 = torch.int32
    g = GnnGraph()
    g.X = _dev(batch.X, torch.float32, "X")
    for k in ("src", "dst", "in_ptr", "in_eid", "in_nbr", "out_ptr", "out_eid", "out_nbr"):
        setattr(g, k, _dev(getattr(batch, k), i32, k))
    g.n_hits, g.n_segments = batch.n_hits, batch.n_segments
    g._device = batch.X.device
    devs = {getattr(batch, k).device for k in ("src", "dst", "in_ptr", "in_eid", "in_nbr",
                                               "out_ptr", "out_eid", "out_nbr")}
    if devs != {g._device}:
        raise GnnHipError("the arrays of a batch must live on one device, got %s" % sorted(map(str, devs)))
    return g


def raw_graph_struct(batch):
    """gnn_graph_t of a batch WITHOUT its segment lists (all six pointers NULL): what gnn_segclf_forward_events takes
    for a never-seen batch - it builds the lists in LDS.  Cached on the batch like cached_graph_struct."""
    dev = batch.X.device
    g = getattr(batch, "_gstruct_raw", None)
    if g is None or g._device != dev:
        g = GnnGraph()
        g.X = _dev(batch.X, torch.float32, "X")
        g.src, g.dst = _dev(batch.src, torch.int32, "src"), _dev(batch.dst, torch.int32, "dst")
        if batch.src.device != dev or batch.dst.device != dev:
            raise GnnHipError("the arrays of a batch must live on one device")
        g.n_hits, g.n_segments = batch.n_hits, batch.n_segments
        g._device = dev
        batch._gstruct_raw = g
    return g


def cached_graph_struct(batch):
    """graph_struct(batch), built once per (batch, device): the tensors of a batch are never
    replaced in place, so their device pointers are stable while the batch lives."""
    dev = batch.X.device
    g = getattr(batch, "_gstruct", None)
    if g is None or batch._gstruct_dev != dev:
        g = batch._gstruct = graph_struct(batch)
        batch._gstruct_dev = dev
    return g


def params_struct(weights, F, D, flags=0):
    """weights: the ten effective (masked) tensors in state_dict order."""
    p = GnnParams()
    C = F + D
    shapes = ((D, F), (D,), (D, 2 * C), (D,), (1, D), (1,), (D, 3 * C), (D,), (D, D), (D,))
    if len(weights) != 10:
        raise GnnHipError("expected the ten weight tensors in state_dict order, got %d" % len(weights))
    for name, w, shp in zip(("Win", "bin", "W1", "b1", "W2", "b2", "W3", "b3", "W4", "b4"), weights, shapes):
        n = 1
        for d in shp:
            n *= d
        if torch.is_tensor(w) and w.numel() != n:
            # the kernels index the tensors as [D, ...] rows of (F, D): a mismatch is an out-of-bounds read
            raise GnnHipError("%s has %d elements, but input_dim=%d hidden_dim=%d needs shape %s"
                              % (name, w.numel(), F, D, shp))
        setattr(p, name, _dev(w, torch.float32, name))
    p.F, p.D, p.flags = F, D, flags
    devs = {w.device for w in weights}
    if len(devs) != 1:
        raise GnnHipError("the weight tensors must live on one device, got %s" % sorted(map(str, devs)))
    p._device = devs.pop()
    return p


def exp_product_bound(weights, F, D, x_absmax):
    """max |P'|, |Q'| bound (python float; synchronises).  <= 60 permits GNN_FLAG_EXP_PRODUCT."""
    out = torch.empty(1, dtype=torch.float32, device=x_absmax.device)
    p = params_struct(weights, F, D)
    with _on(x_absmax, p) as st:
        _check(load().gnn_exp_product_bound(ctypes.byref(p), _dev(x_absmax, torch.float32, "x_absmax"),
                                            out.data_ptr(), st))
    return float(out.item())


def input_fwd(X, Win, bin_):
    """[n_hits, F] -> H [n_hits, ldh] = [tanh(Win X + bin) | X | 0]."""
    n, F = X.shape
    D = Win.shape[0]
    ldh = h_stride(F, D)
    H = torch.empty((n, ldh), dtype=torch.float32, device=X.device)
    with _on(X) as st:
        _check(load().gnn_input_fwd(_dev(X, torch.float32, "X"), _dev(Win, torch.float32, "Win"),
                                    _dev(bin_, torch.float32, "bin"), H.data_ptr(), n, F, D, ldh,
                                    st))
    return H


def edge_fwd(H, src, dst, W1, b1, W2, b2, F, D):
    """H [n_hits, ldh] (ldh >= C), src/dst int32 [n_segments] -> e [n_segments]."""
    n, ldh = H.shape
    E = src.shape[0]
    e = torch.empty(E, dtype=torch.float32, device=H.device)
    pq = torch.empty((max(n, 1), 2 * D), dtype=torch.float32, device=H.device)
    with _on(H) as st:
        _check(load().gnn_edge_fwd(_dev(H, torch.float32, "H"), ldh, _dev(src, torch.int32, "src"),
                                   _dev(dst, torch.int32, "dst"), _dev(W1, torch.float32, "W1"),
                                   _dev(b1, torch.float32, "b1"), _dev(W2, torch.float32, "W2"),
                                   _dev(b2, torch.float32, "b2"), e.data_ptr(), pq.data_ptr(),
                                   n, E, F, D, st))
    return e


def node_fwd(H, e, batch, W3, b3, W4, b4, F, D):
    """H [n_hits, ldh], e [n_segments] -> Hnext [n_hits, ldh] = [H' | X | 0]."""
    n, ldh = H.shape
    Hn = torch.zeros_like(H)
    g = graph_struct(batch)
    with _on(H, g) as st:
        _check(load().gnn_node_fwd(_dev(H, torch.float32, "H"), ldh, _dev(e, torch.float32, "e"),
                                   ctypes.byref(g), _dev(W3, torch.float32, "W3"),
                                   _dev(b3, torch.float32, "b3"), _dev(W4, torch.float32, "W4"),
                                   _dev(b4, torch.float32, "b4"), Hn.data_ptr(), F, D, st))
    return Hn


def workspace_bytes(n_hits, n_segments, F, D):
    return int(load().gnn_forward_workspace_bytes(n_hits, n_segments, F, D))


def segclf_forward(batch, weights, F, D, n_iters, out=None, workspace=None, trace=False):
    """Whole SegmentClassifier forward on an index-form batch.

    Returns scores [n_segments] (and, with trace=True, e_trace [(T+1), E] and
    H_trace [(T+1), N, C])."""
    dev = batch.X.device
    E, N = batch.n_segments, batch.n_hits
    if not shape_supported(F, D):
        raise GnnHipError("no HIP kernel for input_dim=%d hidden_dim=%d" % (F, D))
    need = workspace_bytes(N, E, F, D)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=dev)
    if out is None:
        out = torch.empty(E, dtype=torch.float32, device=dev)
    et = Ht = None
    if trace:
        et = torch.empty((n_iters + 1, E), dtype=torch.float32, device=dev)
        Ht = torch.empty((n_iters + 1, N, F + D), dtype=torch.float32, device=dev)
    g = graph_struct(batch)
    p = params_struct(weights, F, D)
    with _on(batch.X, g, p) as st:
        _check(load().gnn_segclf_forward(ctypes.byref(g), ctypes.byref(p), n_iters,
                                         _dev(out, torch.float32, "out"),
                                         et.data_ptr() if trace else None,
                                         Ht.data_ptr() if trace else None,
                                         workspace.data_ptr(), workspace.numel(), st))
    return (out, et, Ht) if trace else out


@functools.lru_cache(maxsize=None)
def events_supported(F, D, max_hits, max_segments):
    """True if graphs of at most that size fit the one-workgroup-per-graph kernel."""
    return bool(load().gnn_events_supported(F, D, max_hits, max_segments))


# Beyond this many segments per graph the one-workgroup-per-graph kernels lose to the tiled pipeline /
# per-pass kernels even when a graph fits their LDS (tools/cliff_probe.py: 256 x (300 hits, 2000 segments)
# 0.119 ms against 0.069; 256 x (150, 1000) 0.050 against 0.066)
EVENTS_MAX_SEGMENTS = 1200


def events_preferred(F, D, layout, backward=False):
    """Should a batch with this event layout take the one-launch kernels?  (They must be able to - LDS -
    and the graphs must be small enough to be worth a workgroup each.)"""
    if layout is None or layout.max_segments > EVENTS_MAX_SEGMENTS or D > 16:
        return False                 # (wide hidden layers: one toy graph 0.123 ms in one workgroup, 0.089 ms tiled)
    if not events_supported(F, D, layout.max_hits, layout.max_segments):
        return False
    return (not backward) or events_backward_supported(F, D, layout.max_hits, layout.max_segments)


def segclf_forward_events(batch, layout, weights, F, D, n_iters, out=None, params=None):
    """Whole forward in one launch, one workgroup per graph (small events); `layout` is
    `batch.event_layout()`.  Returns scores [n_segments], bit-identical to segclf_forward."""
    dev = batch.X.device
    if out is None:
        out = torch.empty(batch.n_segments, dtype=torch.float32, device=dev)
    # a batch nobody has asked the segment lists of (a never-seen event): the kernel builds them in LDS itself;
    # one graph: no offset arrays either - nothing is prepared or uploaded for a single fresh event
    g = cached_graph_struct(batch) if getattr(batch, "_csr", None) is not None else raw_graph_struct(batch)
    p = params if params is not None else params_struct(weights, F, D)
    with _on(batch.X, g, p) as st:
        if batch.n_graphs == 1:
            hp = sp = None
        else:
            hp, sp = layout.ptrs(dev)
            hp, sp = _dev(hp, torch.int32, "hit_ptr"), _dev(sp, torch.int32, "seg_ptr")
        _check(load().gnn_segclf_forward_events(
            ctypes.byref(g), ctypes.byref(p), hp, sp, batch.n_graphs, layout.max_hits,
            layout.max_segments, n_iters, _dev(out, torch.float32, "out"), st))
    return out


def segclf_forward_train(batch, weights, F, D, n_iters, layout=None):
    """Training forward: returns (e_all [(T+1), E], H_all [(T+1), N, ldh], Q_all [T, N, D]); scores =
    e_all[-1].  Q_all (the node networks' hidden layers) is empty on the one-launch route, whose
    backward keeps everything in LDS.
    `layout` (batch.event_layout() of a batch of small graphs): one launch for the whole forward."""
    dev = batch.X.device
    E, N = batch.n_segments, batch.n_hits
    ldh = h_stride(F, D)
    e_all = torch.empty((n_iters + 1, E), dtype=torch.float32, device=dev)
    H_all = torch.empty((n_iters + 1, N, ldh), dtype=torch.float32, device=dev)
    if layout is not None:
        g = cached_graph_struct(batch)
        p = params_struct(weights, F, D)
        with _on(batch.X, g, p) as st:
            _check(load().gnn_segclf_forward_train_events(
                ctypes.byref(g), ctypes.byref(p), _dev(layout.hit_ptr, torch.int32, "hit_ptr"),
                _dev(layout.seg_ptr, torch.int32, "seg_ptr"), batch.n_graphs, layout.max_hits,
                layout.max_segments, n_iters, _dev(e_all, torch.float32, "e_all"),
                _dev(H_all, torch.float32, "H_all"), st))
        return e_all, H_all, torch.empty((0, N, D), dtype=torch.float32, device=dev)
    Q_all = torch.empty((n_iters, N, D), dtype=torch.float32, device=dev)
    ws = torch.empty(workspace_bytes(N, E, F, D), dtype=torch.uint8, device=dev)
    g = cached_graph_struct(batch)
    p = params_struct(weights, F, D)
    with _on(batch.X, g, p) as st:
        _check(load().gnn_segclf_forward_train(ctypes.byref(g), ctypes.byref(p), n_iters,
                                               e_all.data_ptr(), H_all.data_ptr(), Q_all.data_ptr(),
                                               ws.data_ptr(), ws.numel(), st))
    return e_all, H_all, Q_all


def segclf_forward_train_fused(batch, weights, F, D, n_iters, want_out=True):
    """The training forward of a PLAN-SPACE batch (HitGraphBatch.level_ordered) on the fused tile kernels of the
    plan it was made from.  Returns (e_all, H_all, Q_all, e_out) - the first three as segclf_forward_train gives
    them for this batch (row T of e_all by k_edge_tw, in this batch's segment order), e_out the final scores in the
    CALLER's segment order (None with want_out=False: a loss taken in this batch's order needs no second copy) - or
    None when this batch or shape has no fused training forward (GNN_NO_FUSED_TRAIN=1 also says no: A / B runs)."""
    plan = getattr(batch, "_fused", None)
    if plan is None or getattr(batch, "_fused_dim", None) != D or os.environ.get("GNN_NO_FUSED_TRAIN"):
        return None
    if plan.n_pad != batch.n_hits or plan.n_segments != batch.n_segments or not plan_shape_supported(F, D):
        return None
    n_valid = getattr(batch, "_n_valid", None)
    if n_valid is None:
        n_valid = batch._n_valid = int((batch.src >= 0).sum().item())          # once per batch
    return segclf_forward_train_plan(plan, batch.in_ptr, n_valid, weights, F, D, n_iters, tw_src=batch.src,
                                     tw_dst=batch.dst, want_out=want_out)


def segclf_backward(batch, weights, F, D, n_iters, e_all, H_all, grad_out, into=None, Q_all=None):
    """Gradients of the ten (effective) weight tensors, in state_dict order (`into`: ten tensors the
    gradients are ADDED into instead of a fresh zero buffer; `Q_all`: the hidden layers kept by
    segclf_forward_train - without them the node passes are walked a second time)."""
    if Q_all is not None and Q_all.numel() != n_iters * batch.n_hits * D:
        Q_all = None
    dev = batch.X.device
    if into is not None:
        grads, gs = list(into), _grads_into(into)
    else:       # one zero-filled buffer, ten views (one memset launch instead of ten)
        grads, gs = _grad_views(weights, dev)
    need = int(load().gnn_backward_workspace_bytes(batch.n_hits, batch.n_segments, F, D))
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    g = cached_graph_struct(batch)
    p = params_struct(weights, F, D)
    with _on(batch.X, g, p) as st:
        _check(load().gnn_segclf_backward(ctypes.byref(g), ctypes.byref(p), n_iters,
                                          _dev(e_all, torch.float32, "e_all"),
                                          _dev(H_all, torch.float32, "H_all"),
                                          _dev(Q_all, torch.float32, "Q_all") if Q_all is not None and Q_all.numel() else None,
                                          _dev(grad_out, torch.float32, "grad_out"),
                                          ctypes.byref(gs), ws.data_ptr(), ws.numel(), st))
    return grads


def dense_to_index(Ri, Ro):
    """Dense [B, N, E] float32 incidence matrices on the device -> (src, dst int32 [B*E], flags int32 [1]);
    asynchronous - the caller decides whether to read the flags back."""
    B, N, E = Ri.shape
    src = torch.empty(B * E, dtype=torch.int32, device=Ri.device)
    dst = torch.empty(B * E, dtype=torch.int32, device=Ri.device)
    flags = torch.empty(1, dtype=torch.int32, device=Ri.device)
    with _on(Ri) as st:
        _check(load().gnn_dense_to_index(_dev(Ri, torch.float32, "Ri"), _dev(Ro, torch.float32, "Ro"), B, N, E,
                                         src.data_ptr(), dst.data_ptr(), flags.data_ptr(), st))
    return src, dst, flags


def csr_build(src, dst, n_hits):
    """The two segment lists of a batch on the device (gnn_csr_build): (in_ptr, in_eid, in_nbr, out_ptr, out_eid,
    out_nbr, status) - eid / nbr arrays of n_segments entries (the lists, then -1), status int32 [1] on the device
    (bit 0: malformed endpoints).  Asynchronous, no read-back - the caller decides whether to look at the status."""
    dev, i32 = src.device, torch.int32
    E = int(src.numel())
    ptrs = torch.empty((2, n_hits + 1), dtype=i32, device=dev)
    lists = torch.empty((4, max(E, 1)), dtype=i32, device=dev)
    status = torch.empty(1, dtype=i32, device=dev)
    need = int(load().gnn_csr_build_workspace_bytes(n_hits, E))
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    with _on(src) as st:
        _check(load().gnn_csr_build(_dev(src, i32, "src"), _dev(dst, i32, "dst"), n_hits, E, ptrs[0].data_ptr(),
                                    lists[0].data_ptr(), lists[1].data_ptr(), ptrs[1].data_ptr(), lists[2].data_ptr(),
                                    lists[3].data_ptr(), status.data_ptr(), ws.data_ptr(), ws.numel(), st))
    return ptrs[0], lists[0][:E], lists[1][:E], ptrs[1], lists[2][:E], lists[3][:E], status


def _grad_views(weights, dev):
    """One zero-filled flat buffer, ten views shaped like the weights (state_dict order)."""
    flat = torch.zeros(sum(w.numel() for w in weights), dtype=torch.float32, device=dev)
    grads, o = [], 0
    for w in weights:
        grads.append(flat[o:o + w.numel()].view_as(w))
        o += w.numel()
    gs = GnnGrads()
    for name, t in zip(("Win", "bin", "W1", "b1", "W2", "b2", "W3", "b3", "W4", "b4"), grads):
        setattr(gs, name, t.data_ptr())
    return grads, gs


def edge_bwd(H, batch, weights, F, D, e, grad_e):
    """Backward of EdgeNetwork.forward: (grad_H [n_hits, ldh], [gW1, gb1, gW2, gb2]).
    `weights`: the ten effective tensors (only the edge network's four are read)."""
    dev = H.device
    grads, gs = _grad_views(weights, dev)
    gH = torch.zeros_like(H)
    ws = torch.empty(int(load().gnn_backward_workspace_bytes(batch.n_hits, batch.n_segments, F, D)),
                     dtype=torch.uint8, device=dev)
    g = cached_graph_struct(batch)
    p = params_struct(weights, F, D)
    with _on(H, g, p) as st:
        _check(load().gnn_edge_bwd(_dev(H, torch.float32, "H"), H.shape[1], ctypes.byref(g), ctypes.byref(p),
                                   _dev(e, torch.float32, "e"), _dev(grad_e, torch.float32, "grad_e"),
                                   _dev(gH, torch.float32, "grad_H"), ctypes.byref(gs), ws.data_ptr(), ws.numel(), st))
    return gH, grads[2:6]


def node_bwd(H, e, Hn, batch, weights, F, D, grad_Hn):
    """Backward of NodeNetwork.forward: (grad_H [n_hits, ldh], grad_e [n_segments], [gW3, gb3, gW4, gb4])."""
    dev = H.device
    grads, gs = _grad_views(weights, dev)
    gH = torch.empty_like(H)
    ge = torch.empty(batch.n_segments, dtype=torch.float32, device=dev)
    ws = torch.empty(int(load().gnn_backward_workspace_bytes(batch.n_hits, batch.n_segments, F, D)),
                     dtype=torch.uint8, device=dev)
    g = cached_graph_struct(batch)
    p = params_struct(weights, F, D)
    with _on(H, g, p) as st:
        _check(load().gnn_node_bwd(_dev(H, torch.float32, "H"), H.shape[1], _dev(e, torch.float32, "e"),
                                   _dev(Hn, torch.float32, "Hnext"), ctypes.byref(g), ctypes.byref(p),
                                   _dev(grad_Hn, torch.float32, "grad_Hnext"), _dev(gH, torch.float32, "grad_H"),
                                   _dev(ge, torch.float32, "grad_e"), ctypes.byref(gs), ws.data_ptr(), ws.numel(), st))
    return gH, ge, grads[6:10]


@functools.lru_cache(maxsize=None)
def events_backward_supported(F, D, max_hits, max_segments):
    """True if graphs of at most that size fit the one-launch backward (one workgroup per graph)."""
    return bool(load().gnn_events_backward_supported(F, D, max_hits, max_segments))


def _grads_into(into):
    """GnnGrads over ten caller-owned tensors (the backward ADDS into them)."""
    gs = GnnGrads()
    for name, t in zip(("Win", "bin", "W1", "b1", "W2", "b2", "W3", "b3", "W4", "b4"), into):
        setattr(gs, name, _dev(t, torch.float32, "grad " + name))
    return gs


def segclf_backward_events(batch, layout, weights, F, D, n_iters, e_all, H_all, grad_out, into=None):
    """segclf_backward for a batch of small graphs in ONE launch (`layout` = batch.event_layout()).
    `into`: ten tensors the gradients are ADDED into (e.g. the views of a GradBucket) instead of a
    fresh zero buffer."""
    dev = batch.X.device
    if into is not None:
        grads, gs = list(into), _grads_into(into)
    else:
        grads, gs = _grad_views(weights, dev)
    ws = torch.empty(int(load().gnn_backward_events_workspace_bytes(batch.n_graphs, F, D)), dtype=torch.uint8,
                     device=dev)
    g = cached_graph_struct(batch)
    p = params_struct(weights, F, D)
    with _on(batch.X, g, p) as st:
        _check(load().gnn_segclf_backward_events(
            ctypes.byref(g), ctypes.byref(p), _dev(layout.hit_ptr, torch.int32, "hit_ptr"),
            _dev(layout.seg_ptr, torch.int32, "seg_ptr"), batch.n_graphs, layout.max_hits, layout.max_segments,
            n_iters, _dev(e_all, torch.float32, "e_all"), _dev(H_all, torch.float32, "H_all"),
            _dev(grad_out, torch.float32, "grad_out"), ctypes.byref(gs), ws.data_ptr(), ws.numel(), st))
    return grads


def bce_loss(e, y, scale, want_grad=True):
    """(loss [1], dLoss/de [n] or None): nn.BCELoss value and gradient in one pass (HIP)."""
    n = e.numel()
    dev = e.device
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    grad = torch.empty(n, dtype=torch.float32, device=dev) if want_grad else None
    ws = torch.empty(1024, dtype=torch.float32, device=dev)         # GNN_BCE_WORKSPACE_BYTES
    with _on(e) as st:
        _check(load().gnn_bce_loss(_dev(e, torch.float32, "scores"), _dev(y, torch.float32, "targets"), n,
                                   float(scale), loss.data_ptr(), grad.data_ptr() if want_grad else None,
                                   ws.data_ptr(), st))
    return loss, grad


def plan_shape_supported(F, D):
    return bool(load().gnn_plan_shape_supported(F, D))


def plan_limits(F, D):
    """Tile / chunk sizes and LDS window budgets the plan builder must respect (no GPU needed)."""
    out = (_i32 * 4)()
    _check(load().gnn_plan_limits(F, D, out))
    return {"tile_hits": out[0], "iter_records": out[1], "chunk_segments": out[2],
            "edge_records": out[3]}


def plan_struct(plan):
    g = GnnPlan()
    g.X = _dev(plan.X, torch.float32, "plan.X")
    for k in ("src", "dst", "in_off", "in_nbr", "out_off", "out_nbr", "tiles", "chunks",
              "in_off16", "in_nbr16", "out_off16", "out_nbr16", "sched_a", "sched_b", "sd16"):
        setattr(g, k, _dev(getattr(plan, k), torch.int32, "plan." + k))
    g.n_pad, g.n_segments = plan.n_pad, plan.n_segments
    g.n_tiles, g.n_chunks = plan.n_tiles, plan.n_chunks
    g.iter_lds_records, g.edge_lds_rows = plan.iter_lds_records, plan.edge_lds_rows
    g.n_lds_tiles, g.tile_hits_max = plan.n_lds_tiles, plan.tile_hits_max
    g.iter_lds_in, g.iter_lds_out = plan.iter_lds_in, plan.iter_lds_out
    g.max_list_steps = plan.max_list_steps
    g._device = plan.X.device
    return g


def plan_workspace_bytes(n_hits, n_segments, F, D):
    return int(load().gnn_plan_workspace_bytes(n_hits, n_segments, F, D))


def segclf_forward_plan(plan, weights, F, D, n_iters, out=None, workspace=None, flags=0,
                        params=None):
    """Whole SegmentClassifier forward on a planned batch (fused pipeline) -> scores [E].
    `params`: a GnnParams built earlier from the same `weights` (skips re-validation)."""
    dev = plan.X.device
    need = getattr(plan, "_ws_need", None)
    if need is None or plan._ws_need_shape != (F, D):
        if not plan_shape_supported(F, D):
            raise GnnHipError("no fused HIP kernel for input_dim=%d hidden_dim=%d" % (F, D))
        need = plan._ws_need = plan_workspace_bytes(plan.n_pad, plan.n_segments, F, D)
        plan._ws_need_shape = (F, D)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=dev)
    if out is None:
        out = torch.empty(plan.n_segments, dtype=torch.float32, device=dev)
    g = getattr(plan, "_struct", None)
    if g is None or plan._struct_dev != dev:       # device pointers are stable while the plan lives
        g = plan._struct = plan_struct(plan)
        plan._struct_dev = dev
    p = params if params is not None else params_struct(weights, F, D, flags)
    p.flags = flags
    with _on(plan.X, g, p) as st:
        _check(load().gnn_segclf_forward_plan(ctypes.byref(g), ctypes.byref(p), n_iters,
                                              _dev(out, torch.float32, "out"),
                                              workspace.data_ptr(), workspace.numel(), st))
    return out


def segclf_forward_train_plan(plan, seg_ptr, n_segments_valid, weights, F, D, n_iters, flags=0, workspace=None,
                              tw_src=None, tw_dst=None, want_out=True):
    """The training forward on a planned batch (fused tile kernels; gnn_segclf_forward_train_plan).
    `seg_ptr` int32 [n_pad + 1]: CSR pointer over end hits of the plan-space batch the backward runs on;
    `tw_src` / `tw_dst` int32 [E]: that batch's segment endpoints (plan hit ids, -1 = padded).
    Returns (e_all [(T + 1), E] in that batch's segment order - row T filled when tw_src / tw_dst are given, else
    the caller's to fill from e_out -, H_all [(T + 1), n_pad, ldh], Q_all [T, n_pad, D], e_out [E] in the plan's
    segment order or None with want_out=False) or None when the shape has no fused training forward."""
    dev = plan.X.device
    E, Np = plan.n_segments, plan.n_pad
    ldh = h_stride(F, D)
    need = plan_workspace_bytes(Np, E, F, D)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=dev)
    if tw_src is None and not want_out:
        raise GnnHipError("segclf_forward_train_plan: the final scores are wanted in at least one order")
    # (padded segments are in no hit's list: their entries of rows 0 .. T-1 are never written nor read by the
    # backward's walks; zeros keep them defined)
    e_all = (torch.zeros if n_segments_valid < E else torch.empty)((n_iters + 1, E), dtype=torch.float32, device=dev)
    H_all = torch.empty((n_iters + 1, Np, ldh), dtype=torch.float32, device=dev)
    Q_all = torch.empty((n_iters, Np, D), dtype=torch.float32, device=dev)
    e_out = torch.empty(E, dtype=torch.float32, device=dev) if want_out else None
    g = getattr(plan, "_struct", None)
    if g is None or plan._struct_dev != dev:
        g = plan._struct = plan_struct(plan)
        plan._struct_dev = dev
    p = params_struct(weights, F, D, flags)
    with _on(plan.X, g, p) as st:
        rc = load().gnn_segclf_forward_train_plan(ctypes.byref(g), ctypes.byref(p), n_iters,
                                                  _dev(seg_ptr, torch.int32, "seg_ptr"),
                                                  None if tw_src is None else _dev(tw_src, torch.int32, "tw_src"),
                                                  None if tw_dst is None else _dev(tw_dst, torch.int32, "tw_dst"),
                                                  e_all.data_ptr(), H_all.data_ptr(),
                                                  Q_all.data_ptr(), None if e_out is None else e_out.data_ptr(),
                                                  workspace.data_ptr(), workspace.numel(), st)
    if rc == GNN_ERR_UNSUPPORTED:
        return None
    _check(rc)
    return e_all, H_all, Q_all, e_out


def plan_build_workspace_bytes(n_hits, n_segments, chunk_segments):
    return int(load().gnn_plan_build_workspace_bytes(n_hits, n_segments, chunk_segments))


PLAN_GRAPH_CAP_HITS = 19456      # csrc/plan_build.hip kGraphCapHits: LDS tables and sort keys of the graph-local stage 1
PLAN_STATUS_FAST_MISS = 128


def plan_build_sizes(src, dst, hit_ptr, n_hits, n_segments, n_graphs, tile_hits, iter_records,
                     chunk_segments, edge_records, workspace, seg_ptr=None, max_graph_hits=0, max_graph_segments=0):
    """Stage 1 of the GPU plan builder (csrc/plan_build.hip).  Returns a GnnPlanSizes read back from
    the device - the ONE host synchronisation of a plan build.  With `seg_ptr` (device int64 [G+1]) the
    graph-local form runs (gnn_plan_build_sizes_graphs); status bit PLAN_STATUS_FAST_MISS = call again without."""
    sizes = torch.zeros(ctypes.sizeof(GnnPlanSizes) // 8, dtype=torch.int64, device=src.device)
    with _on(src) as st:
        if seg_ptr is not None:
            _check(load().gnn_plan_build_sizes_graphs(
                _dev(src, torch.int32, "src"), _dev(dst, torch.int32, "dst"), _dev(hit_ptr, torch.int64, "hit_ptr"),
                _dev(seg_ptr, torch.int64, "seg_ptr"), int(max_graph_hits), int(max_graph_segments),
                n_hits, n_segments, n_graphs, tile_hits, iter_records, chunk_segments, edge_records,
                workspace.data_ptr(), workspace.numel(), sizes.data_ptr(), st))
        else:
            _check(load().gnn_plan_build_sizes(
                _dev(src, torch.int32, "src"), _dev(dst, torch.int32, "dst"), _dev(hit_ptr, torch.int64, "hit_ptr"),
                n_hits, n_segments, n_graphs, tile_hits, iter_records, chunk_segments, edge_records,
                workspace.data_ptr(), workspace.numel(), sizes.data_ptr(), st))
    host = sizes.cpu()
    out = GnnPlanSizes()
    ctypes.memmove(ctypes.byref(out), host.data_ptr(), ctypes.sizeof(GnnPlanSizes))
    return out


def plan_build_fill(X, src, dst, n_hits, n_segments, chunk_segments, sizes, workspace, arrays):
    """Stage 2: `arrays` maps the GnnPlanOut field names to the tensors to fill (src_abs, dst_abs,
    level optional)."""
    out = GnnPlanOut()
    with _on(X) as st:
        for name, _ in GnnPlanOut._fields_:
            t = arrays.get(name)
            if t is not None:
                setattr(out, name, _dev(t, torch.float32 if name in ("X", "x_absmax") else torch.int32, name))
        _check(load().gnn_plan_build_fill(
            _dev(X, torch.float32, "X"), X.shape[1], _dev(src, torch.int32, "src"),
            _dev(dst, torch.int32, "dst"), n_hits, n_segments, chunk_segments, ctypes.byref(sizes),
            workspace.data_ptr(), workspace.numel(), ctypes.byref(out), st))


class profile:
    """Context manager: per-kernel HIP-event timings of everything launched inside.

    with _lib.profile(64) as prof: ...; prof.records -> [(kernel_name, ms), ...]"""

    def __init__(self, capacity=256):
        self.capacity = capacity
        self.records = []

    def __enter__(self):
        _check(load().gnn_profile_begin(self.capacity))
        return self

    def __exit__(self, *exc):
        names = (ctypes.c_char_p * self.capacity)()
        ms = (ctypes.c_float * self.capacity)()
        n = load().gnn_profile_end(names, ms, self.capacity)
        if n < 0:
            _check(n)
        self.records = [(names[i].decode(), float(ms[i])) for i in range(min(n, self.capacity))]
        return False

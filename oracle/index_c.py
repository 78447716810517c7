"""ORACLE (test infrastructure): ctypes wrapper over oracle/index_c/segclf_oracle.c.

Build with `make -C oracle` (also done by `__graft_entry__.build()`).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_segclf.so")
_lib = None

_ORDER = ("input_network.0.weight", "input_network.0.bias",
          "edge_network.network.0.weight", "edge_network.network.0.bias",
          "edge_network.network.2.weight", "edge_network.network.2.bias",
          "node_network.network.0.weight", "node_network.network.0.bias",
          "node_network.network.2.weight", "node_network.network.2.bias")


def build():
    subprocess.check_call(["make", "-C", _HERE, "--no-print-directory"],
                          stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int)
        _lib.segclf_oracle_forward.restype = ctypes.c_int
        _lib.segclf_oracle_forward.argtypes = (
            [fp, ctypes.c_long, ctypes.c_int, ip, ip, ctypes.c_long] + [fp] * 10 +
            [ctypes.c_int, ctypes.c_int, fp, fp, fp, ctypes.c_int, ctypes.c_int])
        _lib.segclf_oracle_max_threads.restype = ctypes.c_int
    return _lib


def max_threads():
    return int(lib().segclf_oracle_max_threads())


def segment_classifier(X, src, dst, params, n_iters, masks=None, f64=False, trace=None,
                       n_threads=0):
    """Edge scores [E] float32.  `params`: the ten state_dict arrays; masks applied here
    as W*mask (reference gnn/model.py:30)."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    src = np.ascontiguousarray(src, dtype=np.int32)
    dst = np.ascontiguousarray(dst, dtype=np.int32)
    n, F = X.shape
    E = src.shape[0]
    arrs = []
    for k in _ORDER:
        a = np.asarray(params[k], dtype=np.float32)
        if masks is not None and k in masks:
            a = a * np.asarray(masks[k], dtype=np.float32)
        arrs.append(np.ascontiguousarray(a))
    D = arrs[0].shape[0]
    C = D + F
    e = np.empty(E, dtype=np.float32)
    et = np.empty((n_iters + 1, E), dtype=np.float32) if trace is not None else None
    Ht = np.empty((n_iters + 1, n, C), dtype=np.float32) if trace is not None else None
    fp = ctypes.POINTER(ctypes.c_float)
    ip = ctypes.POINTER(ctypes.c_int)
    p = lambda a: a.ctypes.data_as(fp) if a is not None else None
    rc = lib().segclf_oracle_forward(
        p(X), n, F, src.ctypes.data_as(ip), dst.ctypes.data_as(ip), E,
        *[p(a) for a in arrs], D, n_iters, p(e), p(et), p(Ht), int(n_threads), int(bool(f64)))
    if rc != 0:
        raise RuntimeError("segclf_oracle_forward failed: %d" % rc)
    if trace is not None:
        trace["e"] = [et[t] for t in range(n_iters + 1)]
        trace["H"] = [Ht[t] for t in range(n_iters + 1)]
    return e

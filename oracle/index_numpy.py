"""ORACLE (test infrastructure): index-form restatement of the reference forward in numpy.

Same mathematics as reference gnn/model.py:140-156 with the dense one-hot `bmm`s
replaced by what they compute:

    bmm(Ro^T, H)[j] = H[src[j]]      bmm(Ri^T, H)[j] = H[dst[j]]        model.py:71-72,114-115
    mi[n] = sum_{j: dst[j]=n} e_j * H[src[j]]                           model.py:117-118
    mo[n] = sum_{j: src[j]=n} e_j * H[dst[j]]                           model.py:116,119
    padded segment (src = dst = -1): gathered rows are 0, contributes nothing
                                     (gnn/trainSegmentClassifier.py:83-93)

`dtype=np.float64` gives a higher-precision "truth" for error budgeting.
"""
import numpy as np


def _w(params, masks, key, dtype):
    w = np.asarray(params[key], dtype=dtype)
    if masks is not None and key in masks:
        w = w * np.asarray(masks[key], dtype=dtype)     # model.py:30
    return w


def _gather(H, idx):
    out = H[np.maximum(idx, 0)]
    out[idx < 0] = 0
    return out


def edge_scores(H, src, dst, params, masks=None, dtype=np.float32):
    W1 = _w(params, masks, "edge_network.network.0.weight", dtype)
    W2 = _w(params, masks, "edge_network.network.2.weight", dtype)
    b1 = np.asarray(params["edge_network.network.0.bias"], dtype=dtype)
    b2 = np.asarray(params["edge_network.network.2.bias"], dtype=dtype)
    B = np.concatenate([_gather(H, src), _gather(H, dst)], axis=1)   # out first, then in (:73)
    a = np.tanh(B @ W1.T + b1)
    z = a @ W2.T + b2
    return (1.0 / (1.0 + np.exp(-z)))[:, 0].astype(dtype)


def node_update(H, e, src, dst, params, masks=None, dtype=np.float32):
    W3 = _w(params, masks, "node_network.network.0.weight", dtype)
    W4 = _w(params, masks, "node_network.network.2.weight", dtype)
    b3 = np.asarray(params["node_network.network.0.bias"], dtype=dtype)
    b4 = np.asarray(params["node_network.network.2.bias"], dtype=dtype)
    n = H.shape[0]
    ok = src >= 0
    s, d, w = src[ok], dst[ok], e[ok].astype(dtype)
    mi = np.zeros_like(H)
    mo = np.zeros_like(H)
    np.add.at(mi, d, w[:, None] * H[s])
    np.add.at(mo, s, w[:, None] * H[d])
    M = np.concatenate([mi, mo, H], axis=1)                          # model.py:120
    return np.tanh(np.tanh(M @ W3.T + b3) @ W4.T + b4)


def segment_classifier(X, src, dst, params, n_iters, masks=None, dtype=np.float32, trace=None):
    X = np.asarray(X, dtype=dtype)
    Win = np.asarray(params["input_network.0.weight"], dtype=dtype)
    bin_ = np.asarray(params["input_network.0.bias"], dtype=dtype)
    H = np.concatenate([np.tanh(X @ Win.T + bin_), X], axis=1)       # model.py:144-146
    if trace is not None:
        trace["e"], trace["H"] = [], [H]
    for _ in range(n_iters):
        e = edge_scores(H, src, dst, params, masks, dtype)
        H = np.concatenate([node_update(H, e, src, dst, params, masks, dtype), X], axis=1)
        if trace is not None:
            trace["e"].append(e)
            trace["H"].append(H)
    e = edge_scores(H, src, dst, params, masks, dtype)
    if trace is not None:
        trace["e"].append(e)
    return e
